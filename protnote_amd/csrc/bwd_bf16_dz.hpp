// bf16 backward with dz STORED as bf16 (pn_set_bwd_deep bit 2): k_dz_apply's counterpart writes the rounded dz in place - into
// the first half of every f32 row it came from, so row r of the bf16 matrix starts where row r of the f32 matrix started
// (row stride unchanged, no extra memory) - and the two GEMMs that consume it take a bf16 operand:
//   * dh = dz W        gemm_nt_bf16dma_kernel: BOTH operands by LDS-DMA (no vector instruction touches an operand), BK = 64;
//   * dW = dz^T h      gemm_tn_bf16tr_kernel<TB, true>: the dz tile goes global -> registers -> LDS as it is (no conversion).
// The bytes of the dz operand halve on every path (HBM, fabric, L2 -> L1), and k_dz_apply writes half of what it wrote.
// (Round 6: both GEMMs issue their products as v_mfma_f32_16x16x32_bf16 by default - gemm_bf16_m16.hpp's gemm_nt_bf16m16_kernel and
// the M16 form of gemm_tn_bf16tr_kernel, bit-identical accumulators; pn_set_bf16_mfma16(0) selects the 32 x 32 x 16 forms below.)
// Same bf16 values as the staging-time rounding of gemm_bf16.hpp (round to nearest even, applied to the same f32 dz).
#pragma once
#include "gemm_bf16.hpp"
#include "train_kernels.hpp"

namespace pn {

// In-place compaction: dz row r (C floats at out + r * ldo) becomes C bf16 at the same row start.  A thread owns 4 columns,
// a workgroup of C / 4 threads owns whole rows, ROWS of them per iteration: every thread first loads its part of all ROWS rows
// (z and, for inner layers, the incoming gradient, which is the buffer being overwritten), THEN the workgroup synchronises,
// THEN it stores - a store may land on bytes another thread of the SAME row has just read, never on another row.
// grid: (1, row blocks); block: C / 4 threads (C <= 4096, C % 256 == 0).
template <int ROWG, int ROWS>
__global__ __launch_bounds__(1024) void k_dz_apply_bf16(const DzParams P) {
  const int c = threadIdx.x * 4;
  const float4 s = ld4(P.s + c), t = ld4(P.t + c), cs = ld4(P.cs + c), pp = ld4(P.p + c), q = ld4(P.q + c);
  const long r0 = (long)blockIdx.y * P.rows_per_block;
  long r1 = r0 + P.rows_per_block;
  if (r1 > P.R) r1 = P.R;
  for (long rb = r0; rb < r1; rb += ROWS) {
    float4 z[ROWS], g[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const long r = rb + i < r1 ? rb + i : r1 - 1;  // (clamped re-read past the end; not stored)
      z[i] = ld4(P.Z + r * P.ldz + c);
      if constexpr (ROWG) {
        const float gr = P.gvec[r];
        g[i] = make_float4(gr, gr, gr, gr);
      } else {
        g[i] = ld4(P.G + r * P.ldg + c);
      }
    }
    __syncthreads();  // (waits for every thread's loads: all of these rows are in registers now)
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (rb + i < r1) {
        float4 o;
        o.x = (fmaf(z[i].x, s.x, t.x) > 0.f ? g[i].x * cs.x : 0.f) + fmaf(q.x, z[i].x, pp.x);
        o.y = (fmaf(z[i].y, s.y, t.y) > 0.f ? g[i].y * cs.y : 0.f) + fmaf(q.y, z[i].y, pp.y);
        o.z = (fmaf(z[i].z, s.z, t.z) > 0.f ? g[i].z * cs.z : 0.f) + fmaf(q.z, z[i].z, pp.z);
        o.w = (fmaf(z[i].w, s.w, t.w) > 0.f ? g[i].w * cs.w : 0.f) + fmaf(q.w, z[i].w, pp.w);
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(P.out + (rb + i) * P.ldo) + c) = u32x2{round2(o.x, o.y), round2(o.z, o.w)};
      }
    }
    // (the next iteration's loads touch other rows; its barrier also orders these stores before the stores after it)
  }
}

// W [N][ldw] f32 -> ONE bf16 plane [N][K] in natural k order (the weight operand of gemm_nt_bf16dma_kernel)
__global__ void k_round_plane(const float* __restrict__ W, long ldw, int N, int K, uint16_t* __restrict__ out) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= (long)N * K) return;
  const long n = i / K;
  const int k = (int)(i - n * K);
  const float4 v = ld4(W + n * ldw + k);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u32x2*>(out + i) = u32x2{round2(v.x, v.y), round2(v.z, v.w)};
}

// ---------------------------------------------------------------------------------------------------------------------
// C[M][N] = A[M][K] * W[N][K]^T with BOTH operands bf16 in HBM and staged by LDS-DMA: A = the in-place bf16 dz (row stride
// p.lda FLOATS = 4 p.lda bytes, K bf16 per row), W = k_round_plane's plane (p.w_hi, row stride K bf16).  BK = 64: a tile row
// of a slab is 128 bytes, the LDS image [256 rows][8 granules of 16 B] with granule g of row r at position g ^ ((r >> 1) & 7)
// - the geometry, DMA lane mapping, fragment addressing and rotated slab loop of gemm_nt_dma_kernel's all-DMA path
// (gemm_dma.hpp), with ONE v_mfma_f32_32x32x16_bf16 per fragment pair where that kernel issues four f32 MFMAs.
// K % 64 == 0, N % 256 == 0, a 256-row tile of either operand spans < 4 GB.  LDS: A0 | A1 | B0 | B1, 32 KiB each.
// ---------------------------------------------------------------------------------------------------------------------
template <int EK>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16dma_kernel(const GemmParams p) {
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
  const char* a_tile = reinterpret_cast<const char*>(p.A) + (long)row0 * p.lda * 4;
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
    for (int q = 0; q < 4; ++q) glds16s(src, aoff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };
  auto issue_b = [&](int s, auto buf_c, int q0 = 0, int q1 = 4) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(w_tile + (long)s * SLABB);
    const unsigned base = lds0 + (2 + BUF) * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q >= q0 && q < q1) glds16s(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
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
    issue_a(nxt, N{});
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I1{}, ga, gb);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    issue_b(nxt, N{}, 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I2{}, fa, fb);
    mma(ga, gb);
    __builtin_amdgcn_sched_barrier(0);
    issue_b(nxt, N{}, 2, 4);
    __builtin_amdgcn_sched_barrier(0);
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
  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}
constexpr int NT_BF16DMA_LDS_BYTES = 4 * 256 * 128;

}  // namespace pn
