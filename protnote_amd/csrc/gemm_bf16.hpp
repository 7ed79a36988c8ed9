// Single-product bf16 NT GEMM of the "bf16 backward" (pn_set_backward_math(1)): dh = dz W of the hidden layers' backward,
// operands rounded to bf16 (round to nearest even), ONE v_mfma_f32_32x32x16_bf16 per product, f32 accumulation - the
// arithmetic class of the reference's autocast backward (ProtNoteTrainer.py:728-738).
//
// Same tile (256 x 256, 8 waves of 64 x 128), LDS images, fragment reads, XCD order and epilogue as the NP = 1 instantiation
// of gemm_nt_bf16x3_kernel (its general fallback; same products in the same order: bit-identical results).  What is
// different is the pipeline depth.  A single-product slab is 16 MFMAs per wave - 1024 matrix-pipe cycles per SIMD, about
// half a microsecond - so the bf16x3 schedule (operands of slab s+1 fetched half a slab before they are needed) leaves a
// load less time to land than the memory system takes: measured on the NP = 1 instantiation, matrix pipe 0.45 busy, the
// vector memory unit stalled 62 % of the time (profiles/r04_pmc_bf16_backward.json).  Here every operand is fetched TWO
// slabs ahead:
//   * the register-staged A operand lives in two register sets that alternate by slab parity - the set that held slab s+1
//     is refilled with slab s+3 as soon as its conversion has gone to the LDS;
//   * the weight plane (pre-rounded once per launch, LDS-DMA) rotates through three LDS buffers;
//   * addresses: uniform part in SGPRs, per-lane part loop-invariant - no vector instruction is spent on an address in the
//     slab loop; every wait is a counted vmcnt placed by hand (the loads are issued from inline asm).
// (The same treatment of the TN weight-gradient kernel - two register sets per operand, buffer loads - does not fit: next
//  to 128 accumulators hipcc spills 140-600 registers into the slab loop in every variant tried; its NP = 1 instantiation
//  stays.)
#pragma once
#include "gemm_bf16x3.hpp"
#include "gemm_tn_fast.hpp"

namespace pn {

// s_waitcnt vmcnt(VM) with the awaited registers as operands: no consumer can be scheduled above it
template <int VM>
__device__ __forceinline__ void wait_vm4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(VM) : "memory");
}
// ---------------------------------------------------------------------------------------------------------------------
// C[M][N] = A[M][K] * W[N][K]^T,  A plain f32 (the materialised dz), W as ONE pre-rounded bf16 plane [N][K] in MFMA k-group
// order (k_split_planes' hi plane, p.w_hi), E_STORE.  K % 32 == 0, N % 256 == 0; a 256-row tile of A spans < 4 GB.
// LDS: A0 | A1 (256 rows x 144 B, the bf16x3 image with only the hi slots used) | B0 | B1 | B2 (256 rows x 64 B, granule g of
// row r at position g ^ ((r >> 2) & 3)): 120 KiB.
// ---------------------------------------------------------------------------------------------------------------------
template <int EK>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16_kernel(const GemmParams p) {
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 4;
  constexpr int BM = 256, BN = 256, BK = 32, LDK = BK + 4;
  constexpr unsigned ABYTES = BM * LDK * 4u;  // 36864
  constexpr unsigned BBYTES = BN * 64u;       // 16384

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  int tile_m, tile_n;
  if (!tile_coords<BM, BN>(p, tile_m, tile_n)) return;
  const int row0 = tile_m * BM;
  const int col0 = tile_n * BN;
  const int nslab = p.Kseg / BK;
  const unsigned lds0 = lds_addr(smem);

  // ---- weight plane by LDS-DMA: wave w issues chunks c = 2 w + q (q < 2) of a stage: tile rows 16 c .. + 15; lane l: row +
  //      l / 4, LDS granule position l % 4 <- source granule (l % 4) ^ ((row >> 2) & 3)
  const uint16_t* w_tile = p.w_hi + (long)col0 * p.Kseg;
  unsigned boff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = 16 * (2 * wave + q) + (lane >> 2);
    const int g = (lane & 3) ^ ((r >> 2) & 3);
    boff[q] = (unsigned)((long)r * p.Kseg + 8 * g) * 2u;
  }
  auto issue_b = [&](int s, unsigned bbuf_bytes) {  // bbuf_bytes: LDS byte address of the destination buffer (uniform)
    const float* src = reinterpret_cast<const float*>(w_tile + (long)s * BK);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      glds16s(src, boff[q], __builtin_amdgcn_readfirstlane(bbuf_bytes + (unsigned)(2 * wave + q) * 1024u));
  };

  // ---- A through registers: thread = (row r_in + 128 q, k-group kv): k {4 kv .. + 3} and {16 + 4 kv .. + 3} of the slab
  const int kv = tid & 3;
  const int r_in = tid >> 2;
  const float* a_tile = p.A + (long)row0 * p.lda;
  unsigned aoff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int r = row0 + r_in + 128 * q;
    if (r > p.M - 1) r = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    aoff[q] = (unsigned)((long)(r - row0) * p.lda + 4 * kv) * 4u;
  }
  f32x4 ra[2][2][2];  // [register set][q][half]
  auto fetch_a = [&](auto set_c, int s) {
    constexpr int S = decltype(set_c)::value;
    const float* src = a_tile + s * BK;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      gload4_s(ra[S][q][0], src, aoff[q]);
      gload4_s(ra[S][q][1], src + 16, aoff[q]);
    }
  };
  auto wait_a = [&](auto set_c, auto vm_c) {
    constexpr int S = decltype(set_c)::value, VM = decltype(vm_c)::value;
    wait_vm4<VM>(ra[S][0][0], ra[S][0][1], ra[S][1][0], ra[S][1][1]);
  };
  const unsigned awr = (unsigned)(r_in * LDK + 8 * kv) * 4u;  // LDS byte offset of this thread's granule inside an A buffer
  auto commit_a = [&](auto set_c, auto buf_c) {
    constexpr int S = decltype(set_c)::value, BUF = decltype(buf_c)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bf16x8 v = round8(make_float4(ra[S][q][0].x, ra[S][q][0].y, ra[S][q][0].z, ra[S][q][0].w),
                              make_float4(ra[S][q][1].x, ra[S][q][1].y, ra[S][q][1].z, ra[S][q][1].w));
      *reinterpret_cast<PN_LDS bf16x8*>(lds0 + BUF * ABYTES + awr + (unsigned)(128 * q * LDK) * 4u) = v;
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_row = lane & 31;
  const int frag_g = lane >> 5;
  const unsigned fa_addr = lds0 + (unsigned)((wm * WM * 32 + frag_row) * LDK + frag_g * 8) * 4u;
  // B fragment: row r of a buffer at 64 r bytes, k-group g at granule position g ^ ((r >> 2) & 3)
  unsigned fb_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    fb_off[ks] = (unsigned)((wn * WN * 32 + frag_row) * 64 + 16 * ((2 * ks + frag_g) ^ ((frag_row >> 2) & 3)));
  auto compute = [&](auto abuf_c, unsigned bbuf_bytes, auto ks_c) {
    constexpr int ABUF = decltype(abuf_c)::value, KS = decltype(ks_c)::value;
    bf16x8 a[WM], b[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
      a[i] = *reinterpret_cast<const PN_LDS bf16x8*>(fa_addr + (ABUF * ABYTES + (unsigned)(i * 32 * LDK + 16 * KS) * 4u));
    const unsigned fb = bbuf_bytes + fb_off[KS];
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const PN_LDS bf16x8*>(fb + (unsigned)(j * 32 * 64));
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };
  auto weave = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
#pragma unroll
    for (int i = 0; i < WM * WN; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  const unsigned b_base = lds0 + 2u * ABYTES;
  const int last = nslab - 1;
  auto clamps = [&](int s) { return s < last ? s : last; };
  // prologue: A(0) -> set 0 -> A buffer 0, A(1) -> set 1, A(2) -> set 0; W(0) -> B0, W(1) -> B1
  issue_b(0, b_base);
  issue_b(clamps(1), b_base + BBYTES);
  fetch_a(I0{}, 0);
  fetch_a(I1{}, clamps(1));
  wait_a(I0{}, integral_constant<int, 4>{});  // A(0) landed (and, the counter being in order, both weight stages)
  commit_a(I0{}, I0{});
  fetch_a(I0{}, clamps(2));
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");  // A(1) landed too; A(2) stays in flight
  __builtin_amdgcn_s_barrier();
  // Slab s out of A buffer CUR and weight buffer s % 3.  It issues W(s+2) at its top (that buffer held W(s-1): last read in
  // slab s-1, barrier passed) and A(s+3) after k-step 0, into the register set whose A(s+1) has just gone to the LDS.  The
  // vmcnt counter is in order; oldest first, the operations in flight are
  //   top    (after W(s+2) went out): A(s+1) x 4 [slab s-2] | W(s+1) x 2, A(s+2) x 4 [slab s-1] | W(s+2) x 2
  //                                   -> "at most 8 outstanding" = A(s+1) has landed
  //   bottom (after A(s+3) went out): W(s+1) x 2 | A(s+2) x 4 | W(s+2) x 2, A(s+3) x 4
  //                                   -> "at most 10 outstanding" = W(s+1) has landed (next slab's weight tile)
  // so a weight tile has two slabs to land and an A slab one and a half.
  unsigned bcur = 0;  // s % 3
  auto slab = [&](int s, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    const unsigned bbuf = b_base + bcur * BBYTES;
    const unsigned bnn = bcur >= 1 ? bcur - 1 : 2;  // (s + 2) % 3
    __builtin_amdgcn_sched_barrier(0);
    issue_b(clamps(s + 2), b_base + bnn * BBYTES);
    wait_a(N{}, integral_constant<int, 8>{});
    compute(C{}, bbuf, I0{});
    commit_a(N{}, N{});
    weave();
    __builtin_amdgcn_sched_barrier(0);
    fetch_a(N{}, clamps(s + 3));
    __builtin_amdgcn_sched_barrier(0);
    compute(C{}, bbuf, I1{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bcur = bcur == 2 ? 0 : bcur + 1;
  };
  int s = 0;
  for (; s + 1 < nslab; s += 2) {
    slab(s, I0{});
    slab(s + 1, I1{});
  }
  if (s < nslab) slab(s, I0{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}
constexpr int NT_BF16_LDS_BYTES = 2 * 256 * 36 * 4 + 3 * 256 * 64;

}  // namespace pn
