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

// (slab loop in the style of gemm_nt_dma_kernel: SGPR-base DMA, loop unrolled over the two LDS buffers so that every ds
//  offset is an immediate, the weight tile's DMA after the first k-step - no vector instruction but MFMAs in the loop)
template <int WN>
__global__ __launch_bounds__(512, 2) void gemm_conv_dma_kernel(const ConvDmaParams cp) {
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, BK = 32;
  constexpr int BM = 256, BN = WAVES_N * WN * 32;
  constexpr unsigned TILEA = BM * BK * 4u, TILEB = BN * BK * 4u;  // bytes per operand buffer; LDS: A0 | A1 | B0 | B1
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
  const unsigned lds0 = lds_addr(smem);

  // ---- DMA sources (gemm_dma.hpp): instruction q of wave w covers 8 tile rows; lane l: row + l / 8, LDS granule l % 8
  //      holds source granule (l % 8) ^ ((row >> 1) & 7).  Per-lane byte offsets from the tile's first activation row.
  const int b0 = row0 / p.L;
  const long hrow0 = (long)b0 * cp.Lp + (row0 - b0 * p.L);  // H row of the tile's first output row
  const float* a_tile = p.A + hrow0 * p.lda;
  const float* w_tile = p.W + (long)col0 * p.ldw;
  unsigned aoff[4], boff[QB];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (4 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    int pr = row0 + r;
    if (pr > p.M - 1) pr = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    const int b = pr / p.L;
    const int t = pr - b * p.L;
    aoff[q] = (unsigned)((((long)b * cp.Lp + t) - hrow0) * p.lda + 4 * g) * 4u;
  }
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    const int r = 8 * (QB * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    boff[q] = (unsigned)((long)r * p.ldw + 4 * g) * 4u;  // rows up to round192(Cout) exist (zero padded)
  }
  auto issue_a = [&](long aoff_s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = a_tile + aoff_s;  // (a negative tap offset reaches into the guard rows in front: allocated, zero)
    const unsigned base = lds0 + BUF * TILEA + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(src, aoff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto issue_b = [&](int boff_s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = w_tile + boff_s;
    const unsigned base = lds0 + 2u * TILEA + BUF * TILEB + (unsigned)wave * (QB * 1024u);
#pragma unroll
    for (int q = 0; q < QB; ++q) glds16s(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
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
  unsigned fa_addr[4], fb_addr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int fo = 4 * ((2 * kk + fh) ^ sw);
    fa_addr[kk] = lds0 + (unsigned)((wm * WM * 32 + frow) * BK + fo) * 4u;
    fb_addr[kk] = lds0 + 2u * TILEA + (unsigned)((wn * WN * 32 + frow) * BK + fo) * 4u;
    asm volatile("" : "+v"(fa_addr[kk]), "+v"(fb_addr[kk]));
  }
  auto read_frag = [&](auto buf_c, auto kk_c, float4 (&a)[WM], float4 (&b)[WN]) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = lds_read4(fa_addr[KK] + (BUF * TILEA + i * 32 * BK * 4));
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = lds_read4(fb_addr[KK] + (BUF * TILEB + j * 32 * BK * 4));
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

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using I2 = integral_constant<int, 2>;
  using I3 = integral_constant<int, 3>;
  issue_a(a_tap + c, I0{});
  issue_b(s_b, I0{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  float4 fa[WM], fb[WN], ga[WM], gb[WN];
  auto slab = [&](int s, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    advance(s);
    issue_a(a_tap + c, N{});
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I1{}, ga, gb);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    issue_b(s_b, N{});
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I2{}, fa, fb);
    mma(ga, gb);
    read_frag(C{}, I3{}, ga, gb);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frag(N{}, I0{}, fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    mma(ga, gb);
  };
  read_frag(I0{}, I0{}, fa, fb);
  int s = 0;
  for (; s + 1 < nslab; s += 2) {
    slab(s, I0{});
    slab(s + 1, I1{});
  }
  if (s < nslab) slab(s, I0{});
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
