// f32-MFMA "NT" GEMM for gfx950 with LDS-DMA operand staging (global_load_lds_dwordx4): the 256x256-tile kernel of the
// pair-grid GEMMs, C[M,N] = gen(A)[M,K] * W[N,K]^T, same operand generators / epilogues / tile order as
// gemm_engine.hpp.  What changes is how operands reach the LDS:
//   * W (always a plain matrix) goes global -> LDS directly, no VGPR round trip, no ds_write pass;
//   * A_PLAIN (the dh = dz * W GEMMs of the backward) stages A the same way: the main loop is then nothing but
//     fragment reads + MFMAs, one barrier per slab;
//   * generated A operands (relu(s*z+t), relu(A'[i]+B'[j])) keep the register path, written under the second half
//     of the slab's MFMAs.
// An LDS-DMA wave-instruction writes 64 lanes x 16 B = 1 KiB of CONTIGUOUS LDS, so rows cannot be padded: the LDS
// image is [row][32 floats] (128 B) with the 16-byte granule g of row r stored at position g ^ ((r >> 1) & 7).  The
// swizzle is applied on the per-lane SOURCE address of the DMA (and on the ds_write of the register path) and undone by
// the fragment reads; every 16-lane group of a ds_read_b128 (rows r..r+15 at one k-granule) then covers 16 distinct
// 16-byte slots - conflict-free, like the padded image of gemm_engine.hpp.
// Pipeline per slab s (BK = 32, 128 MFMAs per wave = ~16k cycles per SIMD): issue the DMA (and the register loads) of
// slab s+1 into the other buffer, compute slab s, then s_waitcnt vmcnt(0) + s_barrier: every load has a whole slab
// (or, for the register operand, half of one) to land.  The DMA is issued from inline asm (the compiler does not know
// about it, so it cannot put a conservative vmcnt(0) in front of the fragment reads); its completion is ordered for the
// readers by the explicit vmcnt(0) of the issuing wave followed by the barrier (MI355X_MICROARCH.md, LDS-DMA item 7).
// Restrictions: one K segment, K % 32 == 0, N % 256 == 0 (the pair-grid shapes).
#pragma once
#include "gemm_engine.hpp"

namespace pn {

// one LDS-DMA wave-instruction: lane l copies 16 bytes from its own global address to LDS[lds_base + 16 l]
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_base_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base_uniform)
      : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}

template <int AK, int EK, bool DROP = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_dma_kernel(const GemmParams p) {
  static_assert(!DROP || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "dropout applies to the hidden activations");
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 4, BK = 32;
  constexpr int BM = 256, BN = 256;
  constexpr bool A_DMA = (AK == A_PLAIN);
  static_assert(AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "operand kind not built for the DMA kernel");
  constexpr int TILE = BM * BK;         // floats per operand stage (32 KiB)
  constexpr int STAGE = 2 * TILE;       // A then B

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

  // ---- DMA source addresses: wave w, instruction q covers tile rows 8 (4 w + q) .. + 7; lane l: row + l / 8,
  //      LDS granule position l % 8, i.e. source granule (l % 8) ^ ((row >> 1) & 7)
  const float* bsrc[4];
  const float* asrc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (4 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    bsrc[q] = p.W + (long)(col0 + r) * p.ldw + 4 * g;
    int ra_ = row0 + r;
    if (ra_ > p.M - 1) ra_ = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    asrc[q] = A_DMA ? p.A + (long)ra_ * p.lda + 4 * g : nullptr;
  }
  const unsigned lds0 = lds_addr(smem);
  auto issue_b = [&](int s, int buf) {
    const unsigned base = lds0 + (unsigned)(buf * STAGE + TILE) * 4u + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(bsrc[q] + s * BK, __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto issue_a = [&](int s, int buf) {
    const unsigned base = lds0 + (unsigned)(buf * STAGE) * 4u + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(asrc[q] + s * BK, __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };

  // ---- register path of a generated A operand: thread = (row r_in + 64 q, granule kv), 4 rows per thread
  constexpr int KV = 8, RPP = 512 / KV, NQA = BM / RPP;
  const int kv = tid % KV;
  const int r_in = tid / KV;
  const float* arow[NQA];
  const float* arow2[NQA];
  int aoff[NQA];
  uint32_t a_key[NQA];  // DROP: per-row key of the dropout hash (gemm_engine.hpp)
  int a_col = 0;
  if constexpr (!A_DMA) {
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      const int rl = r_in + q * RPP;
      int r = row0 + rl;
      if (r > p.M - 1) r = p.M - 1;
      a_key[q] = DROP ? drop_rowkey(p.drop_seed, (uint32_t)r) : 0u;
      if constexpr (AK == A_PAIRSUM_RELU) {
        const int j = r / p.pairB;
        const int i = r - j * p.pairB;
        arow[q] = p.A + (long)i * p.lda + 4 * kv;
        arow2[q] = p.A2 + (long)j * p.lda2 + 4 * kv;
      } else {
        arow[q] = p.A + (long)r * p.lda + 4 * kv;
        arow2[q] = nullptr;
      }
      aoff[q] = rl * BK + 4 * (kv ^ ((rl >> 1) & 7));
    }
  }
  float4 ra[NQA], ra2[NQA];
  float4 rsc = make_float4(0, 0, 0, 0), rsh = rsc;
  auto fetch_a = [&](int s) {
    const int c = s * BK;
    if constexpr (DROP) a_col = c + 4 * kv;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      ra[q] = ld4(arow[q] + c);
      if constexpr (AK == A_PAIRSUM_RELU) ra2[q] = ld4(arow2[q] + c);
    }
    if constexpr (AK == A_AFFINE_RELU) {
      rsc = ld4(p.a_scale + c + 4 * kv);
      rsh = ld4(p.a_shift + c + 4 * kv);
    }
  };
  auto pin_a = [&]() {
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      pin4(ra[q]);
      if constexpr (AK == A_PAIRSUM_RELU) pin4(ra2[q]);
    }
  };
  auto commit_a = [&](int buf) {
    float* As = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      float4 v = ra[q];
      if constexpr (AK == A_AFFINE_RELU) {
        v.x = relu(fmaf(v.x, rsc.x, rsh.x));
        v.y = relu(fmaf(v.y, rsc.y, rsh.y));
        v.z = relu(fmaf(v.z, rsc.z, rsh.z));
        v.w = relu(fmaf(v.w, rsc.w, rsh.w));
      } else if constexpr (AK == A_PAIRSUM_RELU) {
        v.x = relu(v.x + ra2[q].x);
        v.y = relu(v.y + ra2[q].y);
        v.z = relu(v.z + ra2[q].z);
        v.w = relu(v.w + ra2[q].w);
      }
      if constexpr (DROP) v = drop4(v, a_key[q], (uint32_t)a_col, p.drop_thresh, p.drop_scale);
      *reinterpret_cast<float4*>(As + aoff[q]) = v;
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane l takes row l % 32 of its wave tile and k-granule 2 kk + l / 32 of k-step kk, stored at
  // granule position (2 kk + l / 32) ^ ((row >> 1) & 7); (row >> 1) & 7 == (l >> 1) & 7 for every tile of the wave
  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int sw = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = 4 * ((2 * kk + fh) ^ sw);
  const int a_base = (wm * WM * 32 + frow) * BK;
  const int b_base = TILE + (wn * WN * 32 + frow) * BK;

  auto compute = [&](int buf, auto kk0_c, auto kk1_c) {
    constexpr int KK0 = decltype(kk0_c)::value, KK1 = decltype(kk1_c)::value;
    const float* As = smem + buf * STAGE + a_base;
    const float* Bs = smem + buf * STAGE + b_base;
#pragma unroll
    for (int kk = KK0; kk < KK1; ++kk) {
      float4 a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * BK + fo[kk]);
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * BK + fo[kk]);
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };

  using std::integral_constant;
  // ---- prologue: slab 0 into buffer 0
  issue_b(0, 0);
  if constexpr (A_DMA) {
    issue_a(0, 0);
  } else {
    fetch_a(0);
    pin_a();
    commit_a(0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  auto read_frag = [&](int buf, int kk, float4 (&a)[WM], float4 (&b)[WN]) {
    const float* As = smem + buf * STAGE + a_base + fo[kk];
    const float* Bs = smem + buf * STAGE + b_base + fo[kk];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * BK);
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * BK);
  };
  auto mma = [&](const float4 (&a)[WM], const float4 (&b)[WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
      }
  };

  if constexpr (A_DMA) {
    // Rotated loop: the last k-step's 32 MFMAs of slab s are issued AFTER the barrier that publishes slab s+1, behind
    // the fragment reads of slab s+1's first k-step - the matrix pipe has work while those reads are in flight, so
    // no wave starts a slab waiting on the LDS.  (The fragment registers alternate between two sets.)  Measured on
    // M = 524288, K = 6144: 144.0 -> 145.1 TFLOP/s; a static s_setprio for either half of the workgroup: 0 %.
    // Ablations of this loop (same shape): without the DMA 150.4, without DMA and barrier 151.8 (the MFMA + fragment-read
    // ceiling), without the barrier only 140.0 (desynchronised waves); spreading the 8 DMA instructions over the four
    // k-steps instead of issuing them in one burst: 139.3.  The LDS-side cost of the operand stream (3.6 %) is what is left.
    float4 fa[WM], fb[WN], ga[WM], gb[WN];
    read_frag(0, 0, fa, fb);
    for (int s = 0; s < nslab; ++s) {
      const int cur = s & 1;
      const int nxt = s + 1 < nslab ? s + 1 : s;  // branch-free: the last slab re-stages itself into the idle buffer
      issue_b(nxt, cur ^ 1);  // the other buffer was last read in slab s-1, which ended with a barrier
      issue_a(nxt, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(cur, 1, ga, gb);
      mma(fa, fb);
      read_frag(cur, 2, fa, fb);
      mma(ga, gb);
      read_frag(cur, 3, ga, gb);
      mma(fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      read_frag(cur ^ 1, 0, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      mma(ga, gb);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    // branch-free: the last slab re-stages itself into the idle buffer (nobody reads it) instead of taking a
    // different path - a conditional fetch makes hipcc wait for the loads right where they are issued
    const int nxt = s + 1 < nslab ? s + 1 : s;
    issue_b(nxt, cur ^ 1);  // the other buffer was last read in slab s-1, which ended with a barrier
    fetch_a(nxt);
    __builtin_amdgcn_sched_barrier(0);
    compute(cur, integral_constant<int, 0>{}, integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    pin_a();  // the register operand has had half a slab (~8k cycles) to land
    compute(cur, integral_constant<int, 2>{}, integral_constant<int, 4>{});
    commit_a(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    // DMA of slab s+1 complete for this wave, its LDS writes and this wave's fragment reads drained; then the barrier
    // makes every wave's share visible (and frees buffer `cur` for the DMA of slab s+2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  }

  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

constexpr int GEMM_DMA_LDS_BYTES = 2 * 2 * 256 * 32 * (int)sizeof(float);  // 128 KiB

}  // namespace pn
