// Encoder convolutions (MaskedConv1D, reference protein_encoders.py:8-17,39-46) as an all-LDS-DMA implicit GEMM for
// gfx950: out[p][n] = sum_tap sum_c H[p + (tap - ntap/2) * dil][c] * W[n][tap][c], 256x192 output tiles (550 -> 3 x 192,
// 1100 -> 6 x 192), the slab loop of gemm_dma.hpp's A_PLAIN kernel (fragment reads + MFMAs + one barrier per slab, both
// operands by global_load_lds_dwordx4).  What makes the tap gather DMA-able is the operand layout, prepared by one
// streaming pass per convolution (k_conv_stage_act):
//   * H holds the convolution's INPUT ACTIVATION relu(bn(x)) already masked (rows t >= len are zero), K padded with
//     zero columns to a multiple of 32, and every sequence followed by G = (ntap/2) * dil zero guard rows (G more in
//     front of the first sequence): row(b, t) = b * (L + G) + t.  A tap shift is then a plain row offset - no bounds
//     test, no select, no per-lane address arithmetic in the loop; a shifted row outside [0, len) reads zeros.
//   * the packed weights [Cout][tap][ld4(Cin)] are re-laid as [round192(Cout)][tap][round32(Cin)] (zero padded), so the
//     B operand of slab s starts at column 32 s of its row.
// Products and their order are those of the register-staged engine (gemm_engine.hpp, A_CONV): the padding contributes
// exact zeros, so the result is bit-identical (tests/test_hip_parity.py::test_encoder_conv_dma_bit_identical).
#pragma once
#include "gemm_dma.hpp"

namespace pn {

struct ConvDmaParams {
  GemmParams g;     // M, N, Nstore, epilogue fields, lens, L; g.A = H row of (b = 0, t = 0), g.lda = ld of H (= Kpad);
                    // g.W = re-laid weights, g.ldw = ntap * Kpad; g.nseg = ntap; g.Kseg = Kpad; g.dil
  int Lp;           // row pitch of one sequence in H (L + G)
};

template <int WN>
__global__ __launch_bounds__(512, 2) void gemm_conv_dma_kernel(const ConvDmaParams cp) {
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, BK = 32;
  constexpr int BM = 256, BN = WAVES_N * WN * 32;
  constexpr int TILE_A = BM * BK, TILE_B = BN * BK;  // floats per operand stage
  constexpr int STAGE = TILE_A + TILE_B;
  constexpr int QB = BN / 64;  // DMA instructions per wave for the B tile (8 rows each)
  const GemmParams& p = cp.g;

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
  const int spt = p.Kseg / BK;  // slabs per tap
  const int nslab = p.nseg * spt;

  // ---- DMA source addresses (gemm_dma.hpp): instruction q of wave w covers 8 tile rows; lane l: row + l / 8, LDS
  //      granule l % 8 holds source granule (l % 8) ^ ((row >> 1) & 7)
  const float* asrc[4];
  const float* bsrc[QB];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (4 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    int pr = row0 + r;
    if (pr > p.M - 1) pr = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    const int b = pr / p.L;
    const int t = pr - b * p.L;
    asrc[q] = p.A + ((long)b * cp.Lp + t) * p.lda + 4 * g;
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int r = 8 * (QB * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    bsrc[q] = p.W + (long)(col0 + r) * p.ldw + 4 * g;  // rows up to round192(Cout) exist (zero padded)
  }
  const unsigned lds0 = lds_addr(smem);
  auto issue = [&](long aoff, int boff, int buf) {
    const unsigned abase = lds0 + (unsigned)(buf * STAGE) * 4u + (unsigned)wave * 4096u;
    const unsigned bbase = lds0 + (unsigned)(buf * STAGE + TILE_A) * 4u + (unsigned)wave * (QB * 1024u);
#pragma unroll
    for (int q = 0; q < QB; ++q) glds16(bsrc[q] + boff, __builtin_amdgcn_readfirstlane(bbase + q * 1024u));
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(asrc[q] + aoff, __builtin_amdgcn_readfirstlane(abase + q * 1024u));
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int sw = (lane >> 1) & 7;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = 4 * ((2 * kk + fh) ^ sw);
  const int a_base = (wm * WM * 32 + frow) * BK;
  const int b_base = TILE_A + (wn * WN * 32 + frow) * BK;

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

  // slab s = (tap, column block c): A offset = (tap - ntap/2) * dil rows + 32 c, B offset = 32 s (taps are contiguous
  // in the re-laid weight rows).  Carried incrementally in scalar registers; the slab after the last re-stages the
  // last one into the idle buffer (branch-free loop, as in gemm_dma.hpp).
  const long tap_step = (long)p.dil * p.lda;
  long a_tap = -(long)(p.nseg / 2) * tap_step;  // row offset of the current tap
  int c = 0, s_b = 0;
  auto advance = [&](int s) {  // move (a_tap, c, s_b) to slab s + 1 unless s is the last slab
    const bool more = s + 1 < nslab;
    const bool wrap = c + BK == p.Kseg;
    const int c1 = wrap ? 0 : c + BK;
    const long t1 = wrap ? a_tap + tap_step : a_tap;
    c = more ? c1 : c;
    a_tap = more ? t1 : a_tap;
    s_b = more ? s_b + BK : s_b;
  };

  issue(a_tap + c, s_b, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float4 fa[WM], fb[WN], ga[WM], gb[WN];
  read_frag(0, 0, fa, fb);
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    advance(s);
    issue(a_tap + c, s_b, cur ^ 1);
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

  gemm_epilogue<E_CONV, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

template <int WN>
constexpr int conv_dma_lds_bytes() {
  return 2 * (256 + 2 * WN * 32) * 32 * (int)sizeof(float);
}

// H[G + b * Lp + t][c] = (t < len[b] && c < C) ? act(x[b * L + t][c]) : 0 over all G + B * Lp rows and Kpad columns;
// act = relu(s * x + t) (BatchNorm fold + ReLU in front of the convolution), or the identity when s is null.
__global__ void k_conv_stage_act(const float* __restrict__ x, long ldx, const float* __restrict__ s,
                                 const float* __restrict__ t, const int* __restrict__ lens, float* __restrict__ H, int Kpad,
                                 int B, int L, int Lp, int G, int C) {
  const int kq = Kpad >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = ((long)G + (long)B * Lp) * kq;
  if (i >= total) return;
  const long row = i / kq;
  const int c = (int)(i - row * kq) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const long rr = row - G;
  if (rr >= 0) {
    const int b = (int)(rr / Lp);
    const int tt = (int)(rr - (long)b * Lp);
    if (tt < L && tt < lens[b] && c < C) {  // C % 4 == 0 is not required: ldx = ld4(C) and x's pad lanes are zero
      v = ld4(x + ((long)b * L + tt) * ldx + c);
      if (s != nullptr) {
        const float4 sc = ld4(s + c), sh = ld4(t + c);
        v.x = relu(fmaf(v.x, sc.x, sh.x));
        v.y = relu(fmaf(v.y, sc.y, sh.y));
        v.z = relu(fmaf(v.z, sc.z, sh.z));
        v.w = relu(fmaf(v.w, sc.w, sh.w));
        // lanes c >= C of the last quad: s, t are zero-padded to ld4(C) by the fold kernels -> relu(0) = 0
      }
    }
  }
  *reinterpret_cast<float4*>(H + row * Kpad + c) = v;
}

// packed [Cout][ntap][ldp] -> [Cpad][ntap][Kpad], zero padded
__global__ void k_conv_relay_weight(const float* __restrict__ w, int Cout, int ntap, int ldp, float* __restrict__ out,
                                    int Cpad, int Kpad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)Cpad * ntap * Kpad;
  if (i >= total) return;
  const int c = (int)(i % Kpad);
  const long rt = i / Kpad;
  const int tap = (int)(rt % ntap);
  const int n = (int)(rt / ntap);
  out[i] = (n < Cout && c < ldp) ? w[((long)n * ntap + tap) * ldp + c] : 0.f;
}

}  // namespace pn
