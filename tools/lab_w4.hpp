// Lab variants (experiment record, not product) of the one-product bf16 NT GEMM: four waves of 128 x 128, placements of the DMA
// pieces, the 16 x 16 x 32 MFMA with three epilogue layouts.  The adopted one lives in csrc/gemm_bf16_m16.hpp; tools/lab_bf16_nt.hip
// times all of them against the shipped kernels.  Everything here is in namespace pn::lab.
//
// gemm_nt_bf16dma_kernel's GEMM (C = A W^T, both operands bf16 in HBM and staged by LDS-DMA, BK = 64, the same LDS image:
// A0 | A1 | B0 | B1, 256 rows x 128 B each, granule g of row r at position g ^ ((r >> 1) & 7)) on FOUR waves, one per SIMD:
// a wave owns 128 x 128 of the 256 x 256 tile (4 x 4 accumulators of 32 x 32 = 256 registers, the AGPR half of the file), so a
// k-step of 16 takes 8 fragment reads for 16 MFMAs where the eight-wave kernel's 64 x 128 wave tiles take 6 for 8 - a third less
// LDS read traffic per product - and nothing competes with the wave for its SIMD's matrix pipe.
//
// Slab loop, rotated at the barrier (phase = 16 MFMAs = 4 groups of one accumulator row):
//   barrier(t): slab t has landed in buffer t % 2; every wave has finished reading the other buffer
//   P0  read (t, k-step 0) -> F      MFMAs of (t - 1, k-step 3) from G      NP0 DMA pieces of slab t + 1 -> the other buffer
//   P1  read (t, 1) -> G             MFMAs of (t, 0) from F                 NP1 pieces
//   P2  read (t, 2) -> F             MFMAs of (t, 1) from G                 NP2 pieces
//   P3  read (t, 3) -> G             MFMAs of (t, 2) from F                 NP3 pieces (0: the last pieces get a phase to land)
//   s_waitcnt vmcnt(0) lgkmcnt(0); barrier(t + 1)
// A group is [DMA pieces] | [fragment reads] | [4 MFMAs], fenced so the reads are issued ahead of the group's MFMAs; the reads of a
// phase all sit in its first two groups (>= 8 MFMAs ahead of their first use), the DMA pieces lean to the last two.
// Same products in the same order per accumulator as gemm_nt_bf16dma_kernel: bit-identical outputs for the epilogues whose
// reduction shape does not depend on the wave grid (E_ROWDOT, E_STORE_H16, plain E_STORE; the BatchNorm column partials of
// E_STORE sum 2 wave rows of 128 where that kernel sums 4 of 64).
#pragma once
#include "bwd_bf16_dz.hpp"

namespace pn {
namespace lab {

// one LDS-DMA piece like glds16s, M0 declared clobbered instead of saved and restored (nothing else in these kernels needs it)
__device__ __forceinline__ void glds16n(const float* sbase_uniform, unsigned voff_bytes, unsigned lds_base_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff_bytes), "s"(sbase_uniform), "s"(lds_base_uniform)
               : "memory", "m0");
}

// SCHED 0: groups of [pieces | 4 reads | 4 MFMAs];  1: one auxiliary op per MFMA gap - reads in gaps 0..7, pieces from gap 8;
// 2: pieces in the even gaps from 0, reads in the odd gaps;  3: reads in gaps 0..7 and pieces in the same gaps
template <int EK, int NP0 = 6, int NP1 = 6, int NP2 = 4, int NP3 = 0, int SCHED = 0, bool M0N = false>
__global__ __launch_bounds__(256, 1) void gemm_nt_bf16dma_w4_kernel(const GemmParams p) {
  static_assert(NP0 + NP1 + NP2 + NP3 == 16, "a slab is 16 DMA pieces per wave");
  constexpr int WAVES_M = 2, WAVES_N = 2, WM = 4, WN = 4;
  constexpr int BM = 256, BN = 256;
  constexpr unsigned SLABB = 128u;          // bytes of one tile row per slab (64 bf16)
  constexpr unsigned TILEB = 256u * SLABB;  // 32 KiB per operand buffer

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;
  const int wn = wave & 1;

  int tile_m, tile_n;
  if (!tile_coords<BM, BN>(p, tile_m, tile_n)) return;
  const int row0 = tile_m * BM;
  const int col0 = tile_n * BN;
  const int nslab = p.Kseg / 64;
  const unsigned lds0 = lds_addr(smem);

  // DMA: wave w, piece q (0..7 of either operand) covers tile rows 8 (8 w + q) .. + 7; lane l: row + l / 8, LDS granule
  // position l % 8 holds source granule (l % 8) ^ ((row >> 1) & 7)
  const char* w_tile = reinterpret_cast<const char*>(p.w_hi) + (long)col0 * p.Kseg * 2;
  const char* a_tile = reinterpret_cast<const char*>(p.A) + (long)row0 * p.lda * 4;
  unsigned boff[8], aoff[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = 8 * (8 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    boff[q] = (unsigned)((long)r * p.Kseg * 2 + 16 * g);
    int ra_ = row0 + r;
    if (ra_ > p.M - 1) ra_ = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    aoff[q] = (unsigned)((long)(ra_ - row0) * p.lda * 4 + 16 * g);
  }
  const unsigned dma_base = lds0 + (unsigned)wave * 8192u;
  // piece P of the slab whose operand rows start at (a_src, w_src), into buffer BUF: 0..7 = A, 8..15 = W
  auto piece = [&](const float* a_src, const float* w_src, auto buf_c, auto p_c) {
    constexpr int BUF = decltype(buf_c)::value, P = decltype(p_c)::value;
    if constexpr (P < 8) {
      if constexpr (M0N) glds16n(a_src, aoff[P], __builtin_amdgcn_readfirstlane(dma_base + BUF * TILEB + P * 1024u));
      else glds16s(a_src, aoff[P], __builtin_amdgcn_readfirstlane(dma_base + BUF * TILEB + P * 1024u));
    } else {
      if constexpr (M0N) glds16n(w_src, boff[P - 8], __builtin_amdgcn_readfirstlane(dma_base + (2 + BUF) * TILEB + (P - 8) * 1024u));
      else glds16s(w_src, boff[P - 8], __builtin_amdgcn_readfirstlane(dma_base + (2 + BUF) * TILEB + (P - 8) * 1024u));
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane l takes row l % 32 of a 32-row fragment and the 8 k of k-step kk's half l / 32 = granule 2 kk + l / 32
  // at position (2 kk + l / 32) ^ ((row >> 1) & 7); (row >> 1) & 7 == (l >> 1) & 7 for every fragment (row offsets % 32 == 0)
  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int sw = (lane >> 1) & 7;
  unsigned fa_addr[4], fb_addr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const unsigned fo = 16u * (unsigned)((2 * kk + fh) ^ sw);
    fa_addr[kk] = lds0 + (unsigned)(wm * 128 + frow) * SLABB + fo;
    fb_addr[kk] = lds0 + 2u * TILEB + (unsigned)(wn * 128 + frow) * SLABB + fo;
    asm volatile("" : "+v"(fa_addr[kk]), "+v"(fb_addr[kk]));
  }
  auto rd_a = [&](auto buf_c, auto kk_c, auto i_c, bf16x8 (&a)[WM]) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value, I = decltype(i_c)::value;
    a[I] = *reinterpret_cast<const PN_LDS bf16x8*>(fa_addr[KK] + (BUF * TILEB + I * 32 * SLABB));
  };
  auto rd_b = [&](auto buf_c, auto kk_c, auto j_c, bf16x8 (&b)[WN]) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value, J = decltype(j_c)::value;
    b[J] = *reinterpret_cast<const PN_LDS bf16x8*>(fb_addr[KK] + (BUF * TILEB + J * 32 * SLABB));
  };
  auto mma_row = [&](auto i_c, const bf16x8 (&a)[WM], const bf16x8 (&b)[WN]) {
    constexpr int I = decltype(i_c)::value;
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[I], b[j], acc[I][j], 0, 0, 0);
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using I2 = integral_constant<int, 2>;
  using I3 = integral_constant<int, 3>;
#define PN_IC(n) integral_constant<int, (n)>{}
#define PN_FENCE() __builtin_amdgcn_sched_barrier(0)

  // one phase: fragments of (RB, KK) into (ra, rb); 16 MFMAs from (ma, mb) when MMA; DMA pieces [P0, P0 + NP) of the next slab into
  // buffer DB, split 1 / 1 / rest over the groups so that the groups that carry fragment reads carry at most one piece
  auto phase = [&](auto rb_c, auto kk_c, bf16x8 (&ra)[WM], bf16x8 (&rb)[WN], auto mma_c, const bf16x8 (&ma)[WM], const bf16x8 (&mb)[WN],
                   const float* a_src, const float* w_src, auto db_c, auto p0_c, auto np_c) {
    using RB = decltype(rb_c);
    using KK = decltype(kk_c);
    using DB = decltype(db_c);
    constexpr bool MMA = decltype(mma_c)::value != 0;
    constexpr int P0 = decltype(p0_c)::value, NP = decltype(np_c)::value;
    // the groups that carry fragment reads (0 and 1) carry at most one piece each
    constexpr int N0 = NP >= 1 ? 1 : 0, N1 = NP >= 2 ? 1 : 0, N2 = (NP - N0 - N1 + 1) / 2, N3 = NP - N0 - N1 - N2;
    static_assert(SCHED != 0 || (N2 <= 4 && N3 <= 4), "at most 4 pieces per group");
    auto pieces = [&](auto q0_c, auto n_c) {
      constexpr int Q0 = decltype(q0_c)::value, NN = decltype(n_c)::value;
      if constexpr (NN > 0) piece(a_src, w_src, DB{}, PN_IC(Q0));
      if constexpr (NN > 1) piece(a_src, w_src, DB{}, PN_IC(Q0 + 1));
      if constexpr (NN > 2) piece(a_src, w_src, DB{}, PN_IC(Q0 + 2));
      if constexpr (NN > 3) piece(a_src, w_src, DB{}, PN_IC(Q0 + 3));
    };
    if constexpr (SCHED != 0) {
      // read k of the phase, in the order the next phase's MFMAs need them: a0 b0 b1 b2 b3 a1 a2 a3
      auto rd = [&](auto k_c) {
        constexpr int K = decltype(k_c)::value;
        if constexpr (K == 0) rd_a(RB{}, KK{}, I0{}, ra);
        else if constexpr (K <= 4) rd_b(RB{}, KK{}, PN_IC(K - 1), rb);
        else rd_a(RB{}, KK{}, PN_IC(K - 4), ra);
      };
      auto slot = [&](auto s_c) {
        constexpr int S = decltype(s_c)::value;
        constexpr int RK = SCHED == 2 ? ((S & 1) ? S / 2 : -1) : (S < 8 ? S : -1);                      // which read rides in gap S
        constexpr int PK = SCHED == 1 ? S - 8 : (SCHED == 2 ? ((S & 1) ? -1 : S / 2) : S);              // which piece
        PN_FENCE();
        if constexpr (PK >= 0 && PK < NP) piece(a_src, w_src, DB{}, PN_IC(P0 + PK));
        if constexpr (RK >= 0) rd(PN_IC(RK));
        PN_FENCE();
        if constexpr (MMA) acc[S / 4][S % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma[S / 4], mb[S % 4], acc[S / 4][S % 4], 0, 0, 0);
      };
      slot(PN_IC(0)); slot(PN_IC(1)); slot(PN_IC(2)); slot(PN_IC(3)); slot(PN_IC(4)); slot(PN_IC(5)); slot(PN_IC(6)); slot(PN_IC(7));
      slot(PN_IC(8)); slot(PN_IC(9)); slot(PN_IC(10)); slot(PN_IC(11)); slot(PN_IC(12)); slot(PN_IC(13)); slot(PN_IC(14)); slot(PN_IC(15));
      PN_FENCE();
      return;
    }
    // group 0
    pieces(PN_IC(P0), PN_IC(N0));
    PN_FENCE();
    rd_a(RB{}, KK{}, I0{}, ra);
    rd_b(RB{}, KK{}, I0{}, rb);
    rd_b(RB{}, KK{}, I1{}, rb);
    rd_b(RB{}, KK{}, I2{}, rb);
    PN_FENCE();
    if constexpr (MMA) mma_row(I0{}, ma, mb);
    PN_FENCE();
    // group 1
    pieces(PN_IC(P0 + N0), PN_IC(N1));
    PN_FENCE();
    rd_b(RB{}, KK{}, I3{}, rb);
    rd_a(RB{}, KK{}, I1{}, ra);
    rd_a(RB{}, KK{}, I2{}, ra);
    rd_a(RB{}, KK{}, I3{}, ra);
    PN_FENCE();
    if constexpr (MMA) mma_row(I1{}, ma, mb);
    PN_FENCE();
    // group 2
    pieces(PN_IC(P0 + N0 + N1), PN_IC(N2));
    PN_FENCE();
    if constexpr (MMA) mma_row(I2{}, ma, mb);
    PN_FENCE();
    // group 3
    pieces(PN_IC(P0 + N0 + N1 + N2), PN_IC(N3));
    PN_FENCE();
    if constexpr (MMA) mma_row(I3{}, ma, mb);
    PN_FENCE();
  };

  // prologue: slab 0 into buffer 0
  {
    const float* a_src = reinterpret_cast<const float*>(a_tile);
    const float* w_src = reinterpret_cast<const float*>(w_tile);
    piece(a_src, w_src, I0{}, PN_IC(0));
    piece(a_src, w_src, I0{}, PN_IC(8));
    piece(a_src, w_src, I0{}, PN_IC(1));
    piece(a_src, w_src, I0{}, PN_IC(9));
    piece(a_src, w_src, I0{}, PN_IC(2));
    piece(a_src, w_src, I0{}, PN_IC(10));
    piece(a_src, w_src, I0{}, PN_IC(3));
    piece(a_src, w_src, I0{}, PN_IC(11));
    piece(a_src, w_src, I0{}, PN_IC(4));
    piece(a_src, w_src, I0{}, PN_IC(12));
    piece(a_src, w_src, I0{}, PN_IC(5));
    piece(a_src, w_src, I0{}, PN_IC(13));
    piece(a_src, w_src, I0{}, PN_IC(6));
    piece(a_src, w_src, I0{}, PN_IC(14));
    piece(a_src, w_src, I0{}, PN_IC(7));
    piece(a_src, w_src, I0{}, PN_IC(15));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 fa[WM], fb[WN], ga[WM], gb[WN];
  // slab t out of buffer CUR (entered behind barrier(t)); FIRST: no MFMAs of a previous slab are pending in G
  auto slab = [&](int t, auto cur_c, auto first_c) {
    constexpr int CUR = decltype(cur_c)::value;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    using PEND = integral_constant<int, decltype(first_c)::value ? 0 : 1>;
    const int nxt = t + 1 < nslab ? t + 1 : t;  // past the end the last slab is staged again into the idle buffer
    const float* a_src = reinterpret_cast<const float*>(a_tile + (long)nxt * SLABB);
    const float* w_src = reinterpret_cast<const float*>(w_tile + (long)nxt * SLABB);
    phase(C{}, I0{}, fa, fb, PEND{}, ga, gb, a_src, w_src, N{}, PN_IC(0), PN_IC(NP0));
    phase(C{}, I1{}, ga, gb, I1{}, fa, fb, a_src, w_src, N{}, PN_IC(NP0), PN_IC(NP1));
    phase(C{}, I2{}, fa, fb, I1{}, ga, gb, a_src, w_src, N{}, PN_IC(NP0 + NP1), PN_IC(NP2));
    phase(C{}, I3{}, ga, gb, I1{}, fa, fb, a_src, w_src, N{}, PN_IC(NP0 + NP1 + NP2), PN_IC(NP3));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  slab(0, I0{}, I1{});
  int t = 1;
  for (; t + 1 < nslab; t += 2) {
    slab(t, I1{}, I0{});
    slab(t + 1, I0{}, I0{});
  }
  if (t < nslab) slab(t, I1{}, I0{});
  // the last slab's k-step 3
  mma_row(I0{}, ga, gb);
  mma_row(I1{}, ga, gb);
  mma_row(I2{}, ga, gb);
  mma_row(I3{}, ga, gb);
#undef PN_IC
#undef PN_FENCE
  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

// the shipped eight-wave kernel with lab switches.  VAR bit 0: M0 clobbered instead of saved / restored;  bit 1: DMA order A x 4 behind
// k-step 0, W x 4 behind k-step 1, nothing later;  bit 2: fragment reads fenced ahead of their k-step's MFMAs;  bit 3: diagnostic (below)
template <int EK, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16dma_v_kernel(const GemmParams p) {
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 4;
  constexpr int BM = 256, BN = 256;
  constexpr unsigned SLABB = 128u;        // bytes of one tile row per slab (64 bf16)
  constexpr unsigned TILEB = 256u * SLABB;  // 32 KiB per operand buffer

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
  const int nslab = p.Kseg / 64;
  const unsigned lds0 = lds_addr(smem);

  // DMA sources: wave w, instruction q covers tile rows 8 (4 w + q) .. + 7; lane l: row + l / 8, LDS granule position l % 8
  // holds source granule (l % 8) ^ ((row >> 1) & 7).  Byte offsets relative to the tile origins.
  const char* w_tile = reinterpret_cast<const char*>(p.w_hi) + (long)col0 * p.Kseg * 2;
  // VAR bit 3 (diagnostic, wrong results): every row tile streams row tile 0's operand rows - the A operand stays in the L2
  const char* a_tile = reinterpret_cast<const char*>(p.A) + ((VAR & 8) ? 0L : (long)row0 * p.lda * 4);
  unsigned boff[4], aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (4 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    boff[q] = (unsigned)((long)r * p.Kseg * 2 + 16 * g);
    int ra_ = row0 + r;
    if (ra_ > p.M - 1) ra_ = p.M - 1;  // clamp: duplicate rows are discarded by the epilogue
    aoff[q] = (unsigned)((long)(ra_ - row0) * p.lda * 4 + 16 * g);
  }
  auto issue_a = [&](int s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(a_tile + (long)s * SLABB);
    const unsigned base = lds0 + BUF * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (VAR & 1) glds16n(src, aoff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
      else glds16s(src, aoff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
    }
  };
  auto issue_b = [&](int s, auto buf_c, int q0 = 0, int q1 = 4) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(w_tile + (long)s * SLABB);
    const unsigned base = lds0 + (2 + BUF) * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q >= q0 && q < q1) {
        if constexpr (VAR & 1) glds16n(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
        else glds16s(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
      }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane l takes row l % 32 of its wave tile and the 8 k of k-step kk's half l / 32 = granule 2 kk + l / 32,
  // stored at granule position (2 kk + l / 32) ^ ((row >> 1) & 7); (row >> 1) & 7 == (l >> 1) & 7 for every tile of the wave
  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int sw = (lane >> 1) & 7;
  unsigned fa_addr[4], fb_addr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const unsigned fo = 16u * (unsigned)((2 * kk + fh) ^ sw);
    fa_addr[kk] = lds0 + (unsigned)(wm * WM * 32 + frow) * SLABB + fo;
    fb_addr[kk] = lds0 + 2u * TILEB + (unsigned)(wn * WN * 32 + frow) * SLABB + fo;
    asm volatile("" : "+v"(fa_addr[kk]), "+v"(fb_addr[kk]));
  }
  auto read_frag = [&](auto buf_c, auto kk_c, bf16x8 (&a)[WM], bf16x8 (&b)[WN]) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const PN_LDS bf16x8*>(fa_addr[KK] + (BUF * TILEB + i * 32 * SLABB));
#pragma unroll
    for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const PN_LDS bf16x8*>(fb_addr[KK] + (BUF * TILEB + j * 32 * SLABB));
  };
  auto mma = [&](const bf16x8 (&a)[WM], const bf16x8 (&b)[WN]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  using I2 = integral_constant<int, 2>;
  using I3 = integral_constant<int, 3>;
  issue_b(0, I0{});
  issue_a(0, I0{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One slab out of buffer CUR; the DMA of slab s+1 goes into the other buffer (last read in slab s-1) - the operand that
  // streams from HBM first, the weight tile in two halves behind the first two k-steps - and has the whole slab to land.
  // Rotated like gemm_nt_dma_kernel: the last k-step's MFMAs are issued after the barrier, behind the first fragment reads of
  // the next slab.  Past the end the last slab is staged again into the idle buffer (branch-free; nobody reads it).
  bf16x8 fa[WM], fb[WN], ga[WM], gb[WN];
  auto slab = [&](int s, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    const int nxt = s + 1 < nslab ? s + 1 : s;
    constexpr bool EARLY = (VAR & 2) != 0, RF = (VAR & 4) != 0;
    issue_a(nxt, N{});
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I1{}, ga, gb);
    if constexpr (RF) __builtin_amdgcn_sched_barrier(0);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (EARLY) issue_b(nxt, N{}, 0, 4); else issue_b(nxt, N{}, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I2{}, fa, fb);
    if constexpr (RF) __builtin_amdgcn_sched_barrier(0);
    mma(ga, gb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!EARLY) issue_b(nxt, N{}, 2, 4);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I3{}, ga, gb);
    if constexpr (RF) __builtin_amdgcn_sched_barrier(0);
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
  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_bf16.  Why: under this package's power limit the register-only MFMA loop sustains
// 2.12 PFLOP/s on 16 x 16 x 32 and 1.86 on 32 x 32 x 16 (tools/mfma_power_probe.hip, round 6) - the deeper dot product moves
// half the accumulator bytes per flop - and every variant of the 32 x 32 x 16 kernel, 4 or 8 waves, any DMA placement, ends
// at the same 1.25 PFLOP/s while the same kernel on a zero operand runs 1.49: the rate is set by watts, not by issue slots.
// Geometry unchanged (eight waves of 64 x 128, the same LDS image and DMA pieces); a wave tile is 4 x 8 fragments of 16 x 16,
// a slab (BK = 64) two k-steps of 32: 12 fragment reads (ds_read_b128: lane l = row l % 16, granule 4 kk + l / 16) for 32 MFMAs.
// Accumulator of fragment (i, j): lane l holds column l % 16, rows 4 (l / 16) + e, e = 0..3.
// Loop, rotated at the barrier:   barrier(s) | read (s, 0) -> F | DMA of slab s + 1 | MFMAs (s - 1, 1) from G |
//                                 read (s, 1) -> G | MFMAs (s, 0) from F | wait, barrier(s + 1)
// VAR bit 0: the weight pieces go behind the first MFMA phase instead of ahead of it;  bit 1: swapped operand roles (epilogue16s);  bit 2: non-temporal stores;  bit 5: staggered start (below);  bit 6: E_STORE_H16 of the swapped layout stores fragment pairs (v_permlane16_swap).
// ---------------------------------------------------------------------------------------------------------------------
// 16-byte / 8-byte global stores, optionally non-temporal (the output is not read again by this kernel)
template <bool NTS>
__device__ __forceinline__ void st16(float* ptr, float a, float b, float c, float d) {
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  const f32x4_ v = {a, b, c, d};
  if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_*>(ptr));
  else *reinterpret_cast<f32x4_*>(ptr) = v;
}
template <bool NTS>
__device__ __forceinline__ void st8(uint16_t* ptr, uint32_t lo, uint32_t hi) {
  typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ v = {lo, hi};
  if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<u32x2_*>(ptr));
  else *reinterpret_cast<u32x2_*>(ptr) = v;
}

// 4 x 4 transpose across the four lanes of a quad (lane & 3 = a): in, lane a holds x[b] = V[b][a]; out, x[b] = V[a][b].
// Two butterfly steps of quad-permute DPP moves (lane ^ 1: [1,0,3,2] = 0xB1, lane ^ 2: [2,3,0,1] = 0x4E) and selects.
__device__ __forceinline__ float dpp_quad(float v, bool xor2) {
  const int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, xor2 ? __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)
                                        : __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ void quad_transpose(float (&x)[4], int a) {
  const bool o1 = a & 1, o2 = a & 2;
  float r0 = dpp_quad(o1 ? x[0] : x[1], false), r1 = dpp_quad(o1 ? x[2] : x[3], false);
  if (o1) { x[0] = r0; x[2] = r1; } else { x[1] = r0; x[3] = r1; }
  r0 = dpp_quad(o2 ? x[0] : x[2], true);
  r1 = dpp_quad(o2 ? x[1] : x[3], true);
  if (o2) { x[0] = r0; x[1] = r1; } else { x[2] = r0; x[3] = r1; }
}

// Epilogue of the 16 x 16 x 32 kernels: accumulator of fragment (i, j), lane l: column l % 16, rows 4 (l / 16) + e.  Stores go
// through quad_transpose over four adjacent fragments, after which lane (c = (l % 16) / 4, a = l % 4) holds columns
// 16 (4 J + a) + 4 c .. + 3 of its row: one 16-byte (f32) / 8-byte (bf16) store per lane, 256 / 128 contiguous bytes per row and
// instruction.  N % 256 == 0 (launcher), so there is no column bound to check.
template <int EK, int WAVES_M, int WAVES_N, int FM, int FN, bool NTS = false>
__device__ __forceinline__ void gemm_epilogue16(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0, int tile_n, float* smem) {
  static_assert(EK == E_STORE || EK == E_ROWDOT || EK == E_STORE_H16, "epilogues of the bf16 h-operand GEMMs");
  static_assert(FN % 4 == 0, "stores transpose four fragments at a time");
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BN = WAVES_N * FN * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int rq = lane >> 4;  // which 4-row group of a 16-row fragment
  const int cl = lane & 15;
  const int qa = lane & 3, qc = cl >> 2;
  const bool want_stats = (EK == E_STORE) && (p.col_part != nullptr);
  const bool store_act = (EK == E_STORE) && (p.e_scale != nullptr);
  float* red = smem;  // [WAVES_M][2][BN] column partials (LDS is free after the final barrier of the main loop)

  if constexpr (EK == E_ROWDOT) {
    float rowacc[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) rowacc[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = col0 + (wn * FN + j) * 16 + cl;
      const float es = p.e_scale[col], et = p.e_shift[col], ew = p.e_w[col];
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) rowacc[i][e] += relu(fmaf(acc[i][j][e], es, et)) * ew;
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = rowacc[i][e];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        const int row = row0 + (wm * FM + i) * 16 + 4 * rq + e;
        if (cl == 0 && row < p.M) p.rowdot_out[(long)(tile_n * WAVES_N + wn) * p.M + row] = v;
      }
    }
    return;
  } else {
    // per-column epilogue terms of this lane's 16-column slice of every fragment, applied in place; column statistics
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = col0 + (wn * FN + j) * 16 + cl;
      float bj = 0.f, es = 1.f, et = 0.f;
      if constexpr (EK == E_STORE) bj = p.bias ? p.bias[col] : 0.f;
      const bool act = (EK == E_STORE_H16) || store_act;
      if (act) {
        es = p.e_scale[col];
        et = p.e_shift[col];
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e];
          if constexpr (EK == E_STORE) v += bj;
          if (act) v = relu(fmaf(v, es, et));
          acc[i][j][e] = v;
          const int row = row0 + (wm * FM + i) * 16 + 4 * rq + e;
          if (row < p.M) {
            s1 += v;
            s2 += v * v;
          }
        }
      }
      if (want_stats) {
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (rq == 0) {
          red[(wm * 2 + 0) * BN + (wn * FN + j) * 16 + cl] = s1;
          red[(wm * 2 + 1) * BN + (wn * FN + j) * 16 + cl] = s2;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + (wm * FM + i) * 16 + 4 * rq + e;
#pragma unroll
        for (int J = 0; J < FN / 4; ++J) {
          float x[4] = {acc[i][4 * J][e], acc[i][4 * J + 1][e], acc[i][4 * J + 2][e], acc[i][4 * J + 3][e]};
          quad_transpose(x, qa);
          const int col = col0 + (wn * FN + 4 * J + qa) * 16 + 4 * qc;
          if (row < p.M) {
            if constexpr (EK == E_STORE) {
              st16<NTS>(p.C + (long)row * p.ldc + col, x[0], x[1], x[2], x[3]);
            } else {
              typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
              st8<NTS>(reinterpret_cast<uint16_t*>(p.C) + (long)row * p.ldc + col, round2(x[0], x[1]), round2(x[2], x[3]));
            }
          }
        }
      }
    }
    if (want_stats) {
      __syncthreads();
      const long tile_m = row0 / (WAVES_M * FM * 16);
      for (int i = tid; i < 2 * BN; i += NT) {
        const int which = i / BN, c = i - which * BN;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) a += red[(w * 2 + which) * BN + c];
        const int col = col0 + c;
        p.col_part[(tile_m * 2 + which) * p.N + col] = a;
      }
    }
  }
}

// Epilogue for the SWAPPED operand roles (the MFMA is given the weight fragment as its row operand): accumulator of fragment
// (i, j), lane l: ROW l % 16 of the 16-row fragment i, COLUMNS 4 (l / 16) + e of fragment j - four consecutive columns per lane, so
// a fragment is one 16-byte store per lane (16 rows x 64 contiguous bytes per instruction) and needs no cross-lane move; a row's
// dot product is an in-lane sum and two shuffles; a column's BatchNorm partial is a sum over the 16 lanes of a row group.
template <int EK, int WAVES_M, int WAVES_N, int FM, int FN, bool NTS = false, bool PAIR16 = false, bool QUAD16 = false>
__device__ __forceinline__ void gemm_epilogue16s(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0, int tile_n, float* smem) {
  static_assert(EK == E_STORE || EK == E_ROWDOT || EK == E_STORE_H16, "epilogues of the bf16 h-operand GEMMs");
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BN = WAVES_N * FN * 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int cq = lane >> 4;  // which 4-column group of a 16-column fragment
  const int rl = lane & 15;
  const bool want_stats = (EK == E_STORE) && (p.col_part != nullptr);
  const bool store_act = (EK == E_STORE) && (p.e_scale != nullptr);
  float* red = smem;
  const int colw = col0 + wn * FN * 16 + 4 * cq;  // + 16 j + e
  if constexpr (EK == E_ROWDOT) {
    float rowacc[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) rowacc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const float4 es = ld4(p.e_scale + colw + 16 * j), et = ld4(p.e_shift + colw + 16 * j), ew = ld4(p.e_w + colw + 16 * j);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        rowacc[i] += relu(fmaf(acc[i][j][0], es.x, et.x)) * ew.x;
        rowacc[i] += relu(fmaf(acc[i][j][1], es.y, et.y)) * ew.y;
        rowacc[i] += relu(fmaf(acc[i][j][2], es.z, et.z)) * ew.z;
        rowacc[i] += relu(fmaf(acc[i][j][3], es.w, et.w)) * ew.w;
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float v = rowacc[i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int row = row0 + (wm * FM + i) * 16 + rl;
      if (cq == 0 && row < p.M) p.rowdot_out[(long)(tile_n * WAVES_N + wn) * p.M + row] = v;
    }
    return;
  } else if constexpr (EK == E_STORE_H16 && QUAD16) {
    // bf16 store, FOUR fragments at a time: the permlane16 exchange of the pair variant below, then v_permlane32_swap between the two
    // pairs - lane row cq then holds all 16 columns of fragment j + cq: 32 contiguous bytes per lane (two 16-byte stores), 128 per row
#pragma unroll
    for (int j = 0; j < FN; j += 4) {
      float4 es[4], et[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        es[f] = ld4(p.e_scale + colw + 16 * (j + f));
        et[f] = ld4(p.e_shift + colw + 16 * (j + f));
      }
      const int col = col0 + wn * FN * 16 + 16 * (j + ((cq & 1) | ((cq & 2)))) ;  // fragment j + cq
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = row0 + (wm * FM + i) * 16 + rl;
        uint32_t x[4][2];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          x[f][0] = round2(relu(fmaf(acc[i][j + f][0], es[f].x, et[f].x)), relu(fmaf(acc[i][j + f][1], es[f].y, et[f].y)));
          x[f][1] = round2(relu(fmaf(acc[i][j + f][2], es[f].z, et[f].z)), relu(fmaf(acc[i][j + f][3], es[f].w, et[f].w)));
        }
        // step 1 (rows of 16 lanes): pair (0, 1) -> P = {p0, p1, p2, p3}, pair (2, 3) -> Q
        const auto a0 = __builtin_amdgcn_permlane16_swap(x[0][0], x[1][0], false, false);
        const auto a1 = __builtin_amdgcn_permlane16_swap(x[0][1], x[1][1], false, false);
        const auto b0 = __builtin_amdgcn_permlane16_swap(x[2][0], x[3][0], false, false);
        const auto b1 = __builtin_amdgcn_permlane16_swap(x[2][1], x[3][1], false, false);
        // P = (a0[0], a1[0], a0[1], a1[1]): rows 0 / 2 hold columns 0-7 / 8-15 of fragment j, rows 1 / 3 those of fragment j + 1; Q likewise
        const auto c0 = __builtin_amdgcn_permlane32_swap(a0[0], b0[0], false, false);
        const auto c1 = __builtin_amdgcn_permlane32_swap(a1[0], b1[0], false, false);
        const auto c2 = __builtin_amdgcn_permlane32_swap(a0[1], b0[1], false, false);
        const auto c3 = __builtin_amdgcn_permlane32_swap(a1[1], b1[1], false, false);
        // lanes 0-31 (rows 0, 1): P' = P (columns 0-7 of fragment j / j + 1), Q' = P of rows 2, 3 (their columns 8-15);
        // lanes 32-63 (rows 2, 3): P' = Q of rows 0, 1 (columns 0-7 of fragment j + 2 / j + 3), Q' = Q (their columns 8-15)
        if (row < p.M) {
          typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
          uint16_t* dst = reinterpret_cast<uint16_t*>(p.C) + (long)row * p.ldc + col;
          *reinterpret_cast<u32x4_*>(dst) = u32x4_{c0[0], c1[0], c2[0], c3[0]};
          *reinterpret_cast<u32x4_*>(dst + 8) = u32x4_{c0[1], c1[1], c2[1], c3[1]};
        }
      }
    }
  } else if constexpr (EK == E_STORE_H16 && PAIR16) {
    // bf16 store, two fragments at a time: v_permlane16_swap exchanges the odd 16-lane rows of fragment j's packed columns with the
    // even rows of fragment j + 1's, after which a lane of an even row holds columns 4 cq .. + 7 of fragment j and a lane of an odd
    // row columns 4 (cq - 1) .. + 7 of fragment j + 1: one 16-byte store, 64 contiguous bytes per row and instruction
#pragma unroll
    for (int j = 0; j < FN; j += 2) {
      const float4 es0 = ld4(p.e_scale + colw + 16 * j), et0 = ld4(p.e_shift + colw + 16 * j);
      const float4 es1 = ld4(p.e_scale + colw + 16 * j + 16), et1 = ld4(p.e_shift + colw + 16 * j + 16);
      const int col = col0 + wn * FN * 16 + 16 * (j + (cq & 1)) + 4 * (cq & ~1);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = row0 + (wm * FM + i) * 16 + rl;
        const uint32_t x0 = round2(relu(fmaf(acc[i][j][0], es0.x, et0.x)), relu(fmaf(acc[i][j][1], es0.y, et0.y)));
        const uint32_t x1 = round2(relu(fmaf(acc[i][j][2], es0.z, et0.z)), relu(fmaf(acc[i][j][3], es0.w, et0.w)));
        const uint32_t y0 = round2(relu(fmaf(acc[i][j + 1][0], es1.x, et1.x)), relu(fmaf(acc[i][j + 1][1], es1.y, et1.y)));
        const uint32_t y1 = round2(relu(fmaf(acc[i][j + 1][2], es1.z, et1.z)), relu(fmaf(acc[i][j + 1][3], es1.w, et1.w)));
        const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        if (row < p.M) {
          typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
          *reinterpret_cast<u32x4_*>(reinterpret_cast<uint16_t*>(p.C) + (long)row * p.ldc + col) = u32x4_{s0[0], s1[0], s0[1], s1[1]};
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float4 bj = make_float4(0.f, 0.f, 0.f, 0.f), es = make_float4(1.f, 1.f, 1.f, 1.f), et = bj;
      if constexpr (EK == E_STORE)
        if (p.bias) bj = ld4(p.bias + colw + 16 * j);
      const bool act = (EK == E_STORE_H16) || store_act;
      if (act) {
        es = ld4(p.e_scale + colw + 16 * j);
        et = ld4(p.e_shift + colw + 16 * j);
      }
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = row0 + (wm * FM + i) * 16 + rl;
        float v[4] = {acc[i][j][0] + bj.x, acc[i][j][1] + bj.y, acc[i][j][2] + bj.z, acc[i][j][3] + bj.w};
        if (act) {
          v[0] = relu(fmaf(v[0], es.x, et.x));
          v[1] = relu(fmaf(v[1], es.y, et.y));
          v[2] = relu(fmaf(v[2], es.z, et.z));
          v[3] = relu(fmaf(v[3], es.w, et.w));
        }
        if (row < p.M) {
          if constexpr (EK == E_STORE) {
            st16<NTS>(p.C + (long)row * p.ldc + colw + 16 * j, v[0], v[1], v[2], v[3]);
          } else {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            st8<NTS>(reinterpret_cast<uint16_t*>(p.C) + (long)row * p.ldc + colw + 16 * j, round2(v[0], v[1]), round2(v[2], v[3]));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s1[e] += v[e];
            s2[e] += v[e] * v[e];
          }
        }
      }
      if (want_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int m = 1; m < 16; m <<= 1) {
            s1[e] += __shfl_xor(s1[e], m);
            s2[e] += __shfl_xor(s2[e], m);
          }
        }
        if (rl == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            red[(wm * 2 + 0) * BN + (wn * FN + j) * 16 + 4 * cq + e] = s1[e];
            red[(wm * 2 + 1) * BN + (wn * FN + j) * 16 + 4 * cq + e] = s2[e];
          }
        }
      }
    }
    if (want_stats) {
      __syncthreads();
      const long tile_m = row0 / (WAVES_M * FM * 16);
      for (int i = tid; i < 2 * BN; i += NT) {
        const int which = i / BN, c = i - which * BN;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) a += red[(w * 2 + which) * BN + c];
        p.col_part[(tile_m * 2 + which) * p.N + col0 + c] = a;
      }
    }
  }
}

template <int EK, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16dma_m16_kernel(const GemmParams p) {
  constexpr int WAVES_M = 4, WAVES_N = 2, FM = 4, FN = 8;
  constexpr int BM = 256, BN = 256;
  constexpr unsigned SLABB = 128u;
  constexpr unsigned TILEB = 256u * SLABB;

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
  const int nslab = p.Kseg / 64;
  const unsigned lds0 = lds_addr(smem);
  // VAR bit 5 (lab): stagger the first wave of workgroups (one per CU) over p.L phases of p.dil shader cycles each, so that the
  // CUs - which otherwise run their equal tiles in lockstep and all store at the same moment - spread their epilogues over a tile time
  if constexpr ((VAR & 32) != 0) {
    const int bid = blockIdx.x;
    if (bid < 256 && p.L > 1) {
      const long target = (long)((bid >> 3) % p.L) * p.dil;
      const long t0 = (long)__builtin_amdgcn_s_memtime();
      while ((long)__builtin_amdgcn_s_memtime() - t0 < target) __builtin_amdgcn_s_sleep(32);
    }
  }

  const char* w_tile = reinterpret_cast<const char*>(p.w_hi) + (long)col0 * p.Kseg * 2;
  const char* a_tile = reinterpret_cast<const char*>(p.A) + (long)row0 * p.lda * 4;
  unsigned boff[4], aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 8 * (4 * wave + q) + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    boff[q] = (unsigned)((long)r * p.Kseg * 2 + 16 * g);
    int ra_ = row0 + r;
    if (ra_ > p.M - 1) ra_ = p.M - 1;
    aoff[q] = (unsigned)((long)(ra_ - row0) * p.lda * 4 + 16 * g);
  }
  auto issue_a = [&](int s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(a_tile + (long)s * SLABB);
    const unsigned base = lds0 + BUF * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16n(src, aoff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto issue_b = [&](int s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(w_tile + (long)s * SLABB);
    const unsigned base = lds0 + (2 + BUF) * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16n(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane l takes row l % 16 of a 16-row fragment and granule 4 kk + l / 16 of the slab, stored at position
  // (4 kk + l / 16) ^ ((row >> 1) & 7); fragment rows start at multiples of 16, so (row >> 1) & 7 == (l >> 1) & 7
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int sw = (lane >> 1) & 7;
  unsigned fa_addr[2], fb_addr[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const unsigned fo = 16u * (unsigned)((4 * kk + fq) ^ sw);
    fa_addr[kk] = lds0 + (unsigned)(wm * FM * 16 + frow) * SLABB + fo;
    fb_addr[kk] = lds0 + 2u * TILEB + (unsigned)(wn * FN * 16 + frow) * SLABB + fo;
    asm volatile("" : "+v"(fa_addr[kk]), "+v"(fb_addr[kk]));
  }
  auto read_frag = [&](auto buf_c, auto kk_c, bf16x8 (&a)[FM], bf16x8 (&b)[FN]) {
    constexpr int BUF = decltype(buf_c)::value, KK = decltype(kk_c)::value;
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const PN_LDS bf16x8*>(fa_addr[KK] + (BUF * TILEB + i * 16 * SLABB));
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const PN_LDS bf16x8*>(fb_addr[KK] + (BUF * TILEB + j * 16 * SLABB));
  };
  auto mma = [&](const bf16x8 (&a)[FM], const bf16x8 (&b)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr ((VAR & 2) != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
  };

  auto issue_a_part = [&](int s, auto buf_c, int q0, int q1) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(a_tile + (long)s * SLABB);
    const unsigned base = lds0 + BUF * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q >= q0 && q < q1) glds16n(src, aoff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto issue_b_part = [&](int s, auto buf_c, int q0, int q1) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(w_tile + (long)s * SLABB);
    const unsigned base = lds0 + (2 + BUF) * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q >= q0 && q < q1) glds16n(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto mma_half = [&](const bf16x8 (&a)[FM], const bf16x8 (&b)[FN], auto h_c) {
    constexpr int H = decltype(h_c)::value;
#pragma unroll
    for (int i = 2 * H; i < 2 * H + 2; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if constexpr ((VAR & 2) != 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
  };
  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  issue_b(0, I0{});
  issue_a(0, I0{});
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 fa[FM], fb[FN], ga[FM], gb[FN];
  auto slab = [&](int s, auto cur_c, auto pend_c) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool PEND = decltype(pend_c)::value != 0;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    const int nxt = s + 1 < nslab ? s + 1 : s;  // past the end the last slab is staged again into the idle buffer
    if constexpr ((VAR & 24) != 0) {
      // finer placement: the DMA pieces in pairs between half phases of 16 MFMAs.  bit 3: A01 | 16 | A23 | 16 | W01 | read G | 16 |
      // W23 | 16;  bit 4: A01 | 16 | A23 | 16 | W0123 | read G | 32
      read_frag(C{}, I0{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      issue_a_part(nxt, N{}, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PEND) mma_half(ga, gb, I0{});
      __builtin_amdgcn_sched_barrier(0);
      issue_a_part(nxt, N{}, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PEND) mma_half(ga, gb, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((VAR & 8) != 0) issue_b_part(nxt, N{}, 0, 2); else issue_b_part(nxt, N{}, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(C{}, I1{}, ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(fa, fb, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((VAR & 8) != 0) issue_b_part(nxt, N{}, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(fa, fb, I1{});
      __builtin_amdgcn_sched_barrier(0);
    } else {
      read_frag(C{}, I0{}, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(nxt, N{});
      if constexpr (!(VAR & 1)) issue_b(nxt, N{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (PEND) mma(ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((VAR & 1) != 0) issue_b(nxt, N{});
      __builtin_amdgcn_sched_barrier(0);
      read_frag(C{}, I1{}, ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      mma(fa, fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  slab(0, I0{}, I0{});
  int s = 1;
  for (; s + 1 < nslab; s += 2) {
    slab(s, I1{}, I1{});
    slab(s + 1, I0{}, I1{});
  }
  if (s < nslab) slab(s, I1{}, I1{});
  mma(ga, gb);
  if constexpr ((VAR & 2) != 0) gemm_epilogue16s<EK, WAVES_M, WAVES_N, FM, FN, (VAR & 4) != 0, (VAR & 64) != 0, (VAR & 128) != 0>(p, acc, row0, col0, tile_n, smem);
  else gemm_epilogue16<EK, WAVES_M, WAVES_N, FM, FN, (VAR & 4) != 0>(p, acc, row0, col0, tile_n, smem);
}

}  // namespace lab
}  // namespace pn
