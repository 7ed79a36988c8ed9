// Single-product bf16 GEMMs of the "bf16 backward" (pn_set_backward_math(1)): the pair-grid GEMMs of the hidden layers'
// backward, operands rounded to bf16 (round to nearest even), ONE v_mfma_f32_32x32x16_bf16 per product, f32 accumulation -
// the arithmetic class of the reference's autocast backward (ProtNoteTrainer.py:728-738).  Two kernels, each with the
// NP = 1 instantiation of its bf16x3 counterpart (gemm_bf16x3.hpp) as general fallback and bit-identical to it:
//
//   gemm_nt_bf16_kernel    dh = dz W.  Same tile (256 x 256, 8 waves of 64 x 128), LDS images, fragment reads, XCD order and
//     epilogue as the bf16x3 kernel; what is different is the pipeline depth.  A single-product slab is 16 MFMAs per wave -
//     1024 matrix-pipe cycles per SIMD, about half a microsecond - so the bf16x3 schedule (operands of slab s+1 fetched half
//     a slab before they are needed) leaves a load less time to land than the memory system takes: measured on the NP = 1
//     instantiation, matrix pipe 0.45 busy, the vector memory unit stalled 62 % of the time
//     (profiles/r04_pmc_bf16_backward.json).  Here the register-staged A operand lives in two register sets that alternate
//     by slab parity (the set that held slab s+1 is refilled with slab s+3 as soon as its conversion has gone to the LDS),
//     the weight plane (pre-rounded once per launch, LDS-DMA) rotates through three LDS buffers, addresses are SGPR bases +
//     loop-invariant per-lane offsets, and every wait is a counted vmcnt placed by hand (the loads are issued from inline
//     asm).
//   gemm_tn_bf16tr_kernel  dW = dz^T h.  Operands staged the way they lie in memory (16-byte loads along a row) and
//     transposed by the LDS read path (ds_read_b64_tr_b16) instead of in registers: a quarter of the load instructions.
#pragma once
#include <type_traits>

#include "gemm_bf16x3.hpp"
#include "gemm_tn_fast.hpp"

#ifndef PN_TN_SYNC_SLABS_B16
#define PN_TN_SYNC_SLABS_B16 PN_TN_SYNC_SLABS  // pacing interval of the single-product dW kernel (a slab here is 16 MFMAs per wave)
#endif

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

namespace pn {

// ---------------------------------------------------------------------------------------------------------------------
// Single-product TN contraction (weight gradients of the bf16 backward) with LDS TRANSPOSE reads:
//     dW[m][n] = sum_r dz[r][m] * h[r][n],  h = relu(s z + t) (TB_AFFINE_RELU) or relu(A'[r % B] + B'[r / B]) (TB_PAIRSUM_RELU).
// Both operands are K-major in HBM (the contraction index r is the row), the MFMA wants 8 consecutive k per lane.  The bf16x3
// TN kernel transposes in registers: a thread owns ONE column and loads eight rows of it as single dwords - 32 dword loads
// per thread and slab, and the vector memory unit, not the matrix pipe, sets the pace once a slab is only 16 MFMAs long
// (profiles/r04_pmc_bf16_backward.json: matrix pipe 0.35 busy, TCP stalled 54 % of the time).  gfx950 has the transpose in
// the LDS read path: ds_read_b64_tr_b16 hands lane i of a 16-lane group the i-th 16-bit column of the 4 x 16 block whose
// four rows the group's lanes point at (lane 4 j + i / 4 supplies the address of row j, 4 columns each; probed on the
// device: tools/tr_probe.hip).  So the operands are staged the way they lie in memory - 16-byte loads along a row (a
// wave-load = one whole 1 KiB tile row), converted, written K-major as [32 k-rows][256 columns] bf16 with a 576-byte row
// stride (the four rows of a read then fall on four different 64-byte bank segments: conflict-free) - and a fragment
// (32 columns x 16 k, 8 k per lane) is two transpose reads.  8 + 8 (+1) load instructions per thread and slab instead of 32.
// Preconditions (launcher): M, N multiples of 256, every split a whole number of 32-row slabs, pairB % 32 == 0, 8 rows of an
// operand span < 2 GB.  Same products in the same order as the bf16x3 kernel's NP = 1 instantiation (natural k-groups in
// both): bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x4 lds_read_tr4(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<PN_LDS s16x4*>(addr));
}
__device__ __forceinline__ uint32_t round2(float a, float b) {
  const f32x2 x = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
}
__device__ __forceinline__ void lds_write_bf16x4(unsigned addr, float a, float b, float c, float d) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<PN_LDS u32x2*>(addr) = u32x2{round2(a, b), round2(c, d)};
}

// ABF16: the A operand (dz) is ALREADY bf16 in memory (bwd_bf16_dz.hpp: written in place by k_dz_apply_bf16, row stride
// p.lda FLOATS, i.e. 4 p.lda bytes): a thread stages 16 bytes = 8 columns of rows 2 wave + lane / 32 (+ 16) as they are.
// SYNC: the 32 workgroups of a region task pace each other every PN_TN_SYNC_SLABS slabs (gemm_tn_fast.hpp: arrival counters
// in the caller's workspace, bounded wait, an optimisation and never a dependency) - for the kind whose two operands both
// stream from HBM, so that the task's panels stay in the XCD's L2 while all 32 need them.
// M16 (round 6): the products are issued as v_mfma_f32_16x16x32_bf16 - the shape this package can feed (gemm_bf16_m16.hpp: the
// register-only loop sustains 2.12 PFLOP/s on it, 1.86 on 32 x 32 x 16).  A slab of 32 k-rows is then ONE k-step: a fragment is
// 16 columns x 32 k, lane l takes column l % 16 and k = 8 (l / 16) .. + 7 as two transpose reads of the k-rows 8 (l / 16) .. + 3
// and + 4 .. + 7 - so the two 16-lane groups a ds_read_b64_tr_b16 services together point at k-rows r and r + 8 of the SAME
// columns, which a uniform row stride puts on the same banks (8 x 576 = 18 x 256): the k-rows with bit 3 set are therefore
// stored 32 bytes to the right, into the row's padding (writers and readers add the same constant; rows 0-3 then cover the
// 32-byte slots 0, 2, 4, 6 of the bank window and rows 8-11 the slots 1, 3, 5, 7).  Same staging, pacing, task order and k
// order per accumulator; 24 transpose reads and 32 MFMAs of 16 Kflop per slab where the 32 x 32 x 16 form has 24 and 16 of 32.
template <int TB, bool ABF16 = false, bool SYNC = false, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_bf16tr_kernel(const TnParams p) {
  static_assert(TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU, "operand kind not built for the transpose-read TN kernel");
  constexpr int BM = 256, BN = 256, BK = 32, NQ = 4;
  constexpr unsigned ROWB = 576u;         // LDS bytes of one k-row: 256 bf16 + 64 bytes of padding
  constexpr unsigned SH8 = M16 ? 32u : 0u;  // byte shift of the k-rows with bit 3 set (8-15, 24-31)
  constexpr unsigned TILEB = BK * ROWB;   // one operand buffer; LDS: A0 | A1 | B0 | B1

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = p.N / BN, ntm = p.M / BM;
  int tile_m, tile_n;
  int split = blockIdx.y;
  int* sync_ctr = nullptr;  // this task's four arrival counters (whole-split tasks only: equal work for all 32 workgroups)
  if (p.task_ns > 0) {  // 32-workgroup region tasks (gemm_tn.hpp)
    if (!tn_task_coords(p.task_ns, tile_m, tile_n, split)) return;
#ifndef PN_TN_TASKS_PLAIN
    // Region tasks are taken TRANSPOSED for the kind whose two operands both stream from HBM: 8 dz x 4 activation panels per
    // 32-workgroup task instead of 4 x 8 - the dz panel is bf16 here (bwd_bf16_dz.hpp), half the bytes of an f32 activation
    // panel, so the task fetches 8 x 1 + 4 x 2 = 16 units instead of 4 x 1 + 8 x 2 = 20.  The 12 x 12 tile grid is symmetric, so
    // the transpose is a swap.  [measured, round 5, A/B on rebuilt binaries whose source hash covers the flag:
    // 162.6 / 162.0 -> 156.6 ms per launch, profiles/r05_tn_tasks_ab.json; -DPN_TN_TASKS_PLAIN=1 builds the old order.]
    if (TB == TB_AFFINE_RELU) {
      const int t_ = tile_m;
      tile_m = tile_n;
      tile_n = t_;
    }
#endif
    const int T = ((int)blockIdx.x >> 8) * 8 + ((int)blockIdx.x & 7);
    if (SYNC && p.task_sync != nullptr && T < p.task_ns * 4) sync_ctr = p.task_sync + 4 * T;
  } else if (PN_XCD && TB == TB_PAIRSUM_RELU && (ntm % 4 == 0) && (ntn % 2 == 0)) {
    const int rm = ntm / 4, rn = ntn / 2;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    tile_m = (xcd >> 1) * rm + w / rn;
    tile_n = (xcd & 1) * rn + w % rn;
  } else if (PN_XCD && (ntm % 2 == 0) && (ntn % 4 == 0)) {
    const int rm = ntm / 2, rn = ntn / 4;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    tile_m = (xcd >> 2) * rm + w / rn;
    tile_n = (xcd & 3) * rn + w % rn;
  } else {
    tile_n = blockIdx.x % ntn;
    tile_m = blockIdx.x / ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long r_begin = (long)split * p.rows_per_split;
  long r_end = r_begin + p.rows_per_split;
  if (r_end > p.R) r_end = p.R;
  const int nsl = r_end > r_begin ? (int)((r_end - r_begin) / BK) : 0;  // whole slabs (launcher's precondition)
  const unsigned lds0 = lds_addr(smem);

  // ---- staging: thread (wave, lane) loads rows wave + 8 q (q < 4), columns 4 lane .. + 3 of a slab, for both operands
  const unsigned a_lane = ABF16 ? (unsigned)((long)(2 * wave + (lane >> 5)) * p.lda * 4 + 16 * (lane & 31))
                                : (unsigned)((long)wave * p.lda + 4 * lane) * 4u;
  const unsigned b_lane = (unsigned)((long)wave * p.ldb + 4 * lane) * 4u;
  const unsigned b2_lane = 16u * lane;
  f32x4 ra[NQ], rb[NQ], rb2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bs = {0.f, 0.f, 0.f, 0.f}, bt = bs;
  if constexpr (TB == TB_AFFINE_RELU) {
    const float4 s4 = ld4(p.b_s + n0 + 4 * lane), t4 = ld4(p.b_t + n0 + 4 * lane);
    bs = f32x4{s4.x, s4.y, s4.z, s4.w};
    bt = f32x4{t4.x, t4.y, t4.z, t4.w};
  }
  int pf_i = 0, pf_j = 0, pf_t = 0;  // pair decode of the slab to fetch next (a slab lies inside one label)
  if constexpr (TB == TB_PAIRSUM_RELU) {
    pf_j = (int)(r_begin / p.pairB);
    pf_i = (int)(r_begin - (long)pf_j * p.pairB);
  }
  auto fetch_a = [&](int t) {
    if constexpr (ABF16) {  // (m0 bf16 columns = m0 / 2 floats into the row)
      const float* src = p.A + (r_begin + (long)t * BK) * p.lda + m0 / 2;
      ra[0] = bload4(src, a_lane);
      ra[1] = bload4(src + 16L * p.lda, a_lane);
    } else {
      const float* src = p.A + (r_begin + (long)t * BK) * p.lda + m0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) ra[q] = bload4(src + (long)(8 * q) * p.lda, a_lane);
    }
  };
  auto fetch_b = [&](int t) {  // t = the previous call's t or that + 1
    if constexpr (TB == TB_PAIRSUM_RELU) {
      const bool adv = t != pf_t;
      pf_t = t;
      const int ni = pf_i + (adv ? BK : 0);
      const bool wrap = ni >= p.pairB;
      pf_i = wrap ? ni - p.pairB : ni;
      pf_j = wrap ? pf_j + 1 : pf_j;
      const float* src = p.B + (long)pf_i * p.ldb + n0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = bload4(src + (long)(8 * q) * p.ldb, b_lane);
      rb2 = bload4(p.B2 + (long)pf_j * p.ldb2 + n0, b2_lane);
    } else {
      const float* src = p.B + (r_begin + (long)t * BK) * p.ldb + n0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) rb[q] = bload4(src + (long)(8 * q) * p.ldb, b_lane);
    }
  };
  auto pin_a = [&]() {
    if constexpr (ABF16) asm volatile("" : "+v"(ra[0]), "+v"(ra[1]));
    else asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]));
  };
  auto pin_b = [&]() { asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb2)); };
  unsigned wr_addr = lds0 + (unsigned)wave * ROWB + 8u * lane;  // this thread's 4 columns in tile row `wave` of A buffer 0
  asm volatile("" : "+v"(wr_addr));
  // ABF16: 8 columns of row 2 wave + lane / 32 (and of that + 16: the same bit 3)
  unsigned wr16_addr = lds0 + (unsigned)(2 * wave + (lane >> 5)) * ROWB + 16u * (lane & 31) + ((wave & 4) ? SH8 : 0u);
  asm volatile("" : "+v"(wr16_addr));
  auto commit_a = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    if constexpr (ABF16) {
      *reinterpret_cast<PN_LDS f32x4*>(wr16_addr + BUF * TILEB) = ra[0];
      *reinterpret_cast<PN_LDS f32x4*>(wr16_addr + (BUF * TILEB + 16u * ROWB)) = ra[1];
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        lds_write_bf16x4(wr_addr + (BUF * TILEB + (unsigned)(8 * q) * ROWB + (q & 1) * SH8), ra[q].x, ra[q].y, ra[q].z, ra[q].w);
    }
  };
  auto commit_b = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f32x2 lo, hi;
      if constexpr (TB == TB_AFFINE_RELU) {
        lo = pk_fma(rb[q].xy, bs.xy, bt.xy);
        hi = pk_fma(rb[q].zw, bs.zw, bt.zw);
      } else {
        lo = pk_add(rb[q].xy, rb2.xy);
        hi = pk_add(rb[q].zw, rb2.zw);
      }
      lds_write_bf16x4(wr_addr + ((2 + BUF) * TILEB + (unsigned)(8 * q) * ROWB + (q & 1) * SH8), relu_raw(lo.x), relu_raw(lo.y),
                       relu_raw(hi.x), relu_raw(hi.y));
    }
  };

  // accumulators: 2 x 4 tiles of 32 x 32 (f32x16), or - M16 - 4 x 8 tiles of 16 x 16 (f32x4): 128 registers either way
  typedef typename std::conditional<M16, f32x4, f32x16>::type acc_t;
  constexpr int FI = M16 ? 4 : 2, FJ = M16 ? 8 : 4, AE = M16 ? 4 : 16;
  acc_t acc[FI][FJ];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FJ; ++j)
#pragma unroll
      for (int e = 0; e < AE; ++e) acc[i][j][e] = 0.f;

  // fragment = 32 columns x 16 k of a k-step: lane l takes column l % 32 and k = 8 (l / 32) .. + 7 as two transpose reads of
  // 4 k each.  Inside its 16-lane group lane i points at k-row (i / 4) of the quad and at columns 16 ((l / 16) % 2) + 4 (i % 4)
  // of the 32; the hardware returns column i of that 4 x 16 block.
  // M16: fragment = 16 columns x 32 k: lane group l / 16 takes the k-rows 8 (l / 16) .. + 7, lane i of it points at k-row
  // 8 (l / 16) + i / 4 (+ 4 for the second read) and at columns 4 (i % 4) of the 16.
  const int li = lane & 15, lg = lane >> 4;
  unsigned fa_addr = M16 ? lds0 + (unsigned)(8 * lg + (li >> 2)) * ROWB + (unsigned)(lg & 1) * SH8 + (unsigned)(wm * 64 + 4 * (li & 3)) * 2u
                         : lds0 + (unsigned)(8 * (lg >> 1) + (li >> 2)) * ROWB + (unsigned)(wm * 64 + 16 * (lg & 1) + 4 * (li & 3)) * 2u;
  unsigned fb_addr = M16 ? lds0 + 2u * TILEB + (unsigned)(8 * lg + (li >> 2)) * ROWB + (unsigned)(lg & 1) * SH8 +
                               (unsigned)(wn * 128 + 4 * (li & 3)) * 2u
                         : lds0 + 2u * TILEB + (unsigned)(8 * (lg >> 1) + (li >> 2)) * ROWB +
                               (unsigned)(wn * 128 + 16 * (lg & 1) + 4 * (li & 3)) * 2u;
  asm volatile("" : "+v"(fa_addr), "+v"(fb_addr));
  auto compute = [&](auto buf_c, auto ks_c) {
    constexpr int BUF = decltype(buf_c)::value, KS = decltype(ks_c)::value;
    bf16x8 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const s16x4 lo = lds_read_tr4(fa_addr + (BUF * TILEB + (unsigned)(16 * KS) * ROWB + i * 64u));
      const s16x4 hi = lds_read_tr4(fa_addr + (BUF * TILEB + (unsigned)(16 * KS + 4) * ROWB + i * 64u));
      a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const s16x4 lo = lds_read_tr4(fb_addr + (BUF * TILEB + (unsigned)(16 * KS) * ROWB + j * 64u));
      const s16x4 hi = lds_read_tr4(fb_addr + (BUF * TILEB + (unsigned)(16 * KS + 4) * ROWB + j * 64u));
      b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    if constexpr (!M16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  // M16: the slab's 24 transpose reads, then its 32 MFMAs in two halves (accumulator rows 0-1, 2-3)
  bf16x8 a16[4], b16[8];
  auto read16 = [&](auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const s16x4 lo = lds_read_tr4(fa_addr + (BUF * TILEB + i * 32u));
      const s16x4 hi = lds_read_tr4(fa_addr + (BUF * TILEB + 4u * ROWB + i * 32u));
      a16[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const s16x4 lo = lds_read_tr4(fb_addr + (BUF * TILEB + j * 32u));
      const s16x4 hi = lds_read_tr4(fb_addr + (BUF * TILEB + 4u * ROWB + j * 32u));
      b16[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  };
  auto mma16 = [&](auto h_c) {
    constexpr int H = decltype(h_c)::value;
    if constexpr (M16) {
#pragma unroll
      for (int i = 2 * H; i < 2 * H + 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[i], b16[j], acc[i][j], 0, 0, 0);
    }
  };
  // eight MFMAs of a k-step with half a slab's conversion + LDS writes woven between them
  auto weave = [&](auto nvalu_c) {
    constexpr int NV = decltype(nvalu_c)::value;
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);  // the k-step's transpose reads first
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      if (i % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  // M16: sixteen MFMAs (half a slab) with the same work woven between them; NR transpose reads first
  auto weave16 = [&](auto nread_c, auto nvalu_c) {
    constexpr int NR = decltype(nread_c)::value, NV = decltype(nvalu_c)::value;
    if constexpr (NR > 0) __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  if (nsl > 0) {
    const int last = nsl - 1;
    fetch_a(0);
    fetch_b(0);
    pin_a();
    pin_b();
    commit_a(I0{});
    commit_b(I0{});
    fetch_a(last < 1 ? last : 1);
    fetch_b(last < 1 ? last : 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // slab t out of buffer CUR: A(t+1) converted and written under k-step 0, then A(t+2) fetched; B(t+1) under k-step 1,
    // then B(t+2) fetched (past the end the last slab is fetched again; nobody reads it).
    // (Measured and dropped: a second register set per operand, slab t+3 fetched as soon as slab t+1 has gone to the LDS -
    //  230 registers, no spill, bit-identical; dW = dz^T relu(bn(z)) 172 -> 188 ms per launch, the pair-sum kind 158 -> 160 ms:
    //  with both operands streaming from HBM the extra slab in flight costs more in the caches than the latency it hides.)
    auto checkpoint = [&](int t) {  // (gemm_tn_fast.hpp: epoch e reports into slot e % 4 and waits for epoch e - 1)
      if (sync_ctr != nullptr && (t & (PN_TN_SYNC_SLABS_B16 - 1)) == 0 && wave == 0 && lane == 0) {
        const int e = t / PN_TN_SYNC_SLABS_B16;
        __hip_atomic_fetch_add(sync_ctr + (e & 3), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e > 1) {
          const int* c = sync_ctr + ((e - 1) & 3);
          const int want = 32 * (((e - 1) >> 2) + 1);
          for (int spin = 0; spin < 1024 && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want; ++spin)
            __builtin_amdgcn_s_sleep(8);
        }
      }
    };
    auto slab = [&](int t, auto cur_c) {
      constexpr int CUR = decltype(cur_c)::value;
      using C = integral_constant<int, CUR>;
      using N = integral_constant<int, CUR ^ 1>;
      const int t2 = t + 2 < last ? t + 2 : last;
      __builtin_amdgcn_sched_barrier(0);
      pin_a();
      if constexpr (SYNC && CUR == 0) checkpoint(t);
      if constexpr (M16) {
        read16(C{});
        mma16(I0{});
        commit_a(N{});
        weave16(integral_constant<int, 24>{}, integral_constant<int, 1>{});
      } else {
        compute(C{}, I0{});
        commit_a(N{});
        weave(integral_constant<int, 1>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      fetch_a(t2);
      __builtin_amdgcn_sched_barrier(0);
      pin_b();
      if constexpr (M16) {
        mma16(I1{});
        commit_b(N{});
        weave16(integral_constant<int, 0>{}, integral_constant<int, 2>{});
      } else {
        compute(C{}, I1{});
        commit_b(N{});
        weave(integral_constant<int, 3>{});
      }
      __builtin_amdgcn_sched_barrier(0);
      fetch_b(t2);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    int t = 0;
    for (; t + 2 <= last; t += 2) {
      slab(t, I0{});
      slab(t + 1, I1{});
    }
    if (t < last) slab(t, I0{});
    // the last slab, out of buffer last % 2 (a run-time buffer offset: an if / else over the two compile-time buffers makes
    // hipcc keep two copies of the 128 accumulators across the merge and spill ~250 registers)
    fa_addr += (unsigned)(last & 1) * TILEB;
    fb_addr += (unsigned)(last & 1) * TILEB;
    if constexpr (M16) {
      read16(I0{});
      mma16(I0{});
      mma16(I1{});
    } else {
      compute(I0{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      compute(I0{}, I1{});
    }
  }

  float* out = p.Cpart + (long)split * p.M * p.ldc;
  if constexpr (M16) {  // accumulator (i, j), lane l: column l % 16, rows 4 (l / 16) + e
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = n0 + (wn * 8 + j) * 16 + li;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + (wm * 4 + i) * 16 + 4 * lg + e;
          out[(long)m * p.ldc + n] = acc[i][j][e];
        }
      }
  } else {
    const int hl = lane >> 5, cl = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + (wn * 4 + j) * 32 + cl;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + (wm * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
          out[(long)m * p.ldc + n] = acc[i][j][e];
        }
      }
  }
}
constexpr int TN_BF16TR_LDS_BYTES = 4 * 32 * 576;

}  // namespace pn
