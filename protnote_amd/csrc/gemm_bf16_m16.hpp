// The one-product bf16 NT GEMM of the AMP-class modes (C = H W^T: the hidden pair-grid GEMMs of forward_math = bf16 and
// dh = dz W of backward_math = bf16; both operands bf16 in HBM and staged by LDS-DMA) on v_mfma_f32_16x16x32_bf16.
//
// Why this shape (round 6, tools/lab_bf16_nt.hip + tools/mfma_power_probe.hip, profiles/r06_bf16_gemm_lab.txt): bf16 MFMA issue on
// this package is set by watts, not by issue slots.  gemm_nt_bf16dma_kernel (32 x 32 x 16, bwd_bf16_dz.hpp) ran 1.25 PFLOP/s in
// every variant tried - four waves of 128 x 128 or eight of 64 x 128, six placements of the DMA pieces, fragment reads fenced or
// not: all within 2 % - 1.24 with its streamed operand pinned in the L2, and 1.49 on an all-zero operand; the register-only
// MFMA loop sustains 1.86 PFLOP/s on 32 x 32 x 16 and 2.12 on 16 x 16 x 32 (the deeper dot product moves half the accumulator
// bytes per flop), and the vendor library's kernel for this shape (256 x 256 x 64 tile, 16 x 16 x 32, 1.46 PFLOP/s:
// profiles/r06_bf16_gemm_yardstick.json) uses the latter.  Same tile, LDS image, DMA pieces and XCD order as
// gemm_nt_bf16dma_kernel; the main loop alone gains 12 % (E_ROWDOT 1.25 -> 1.41 PFLOP/s), the f32-storing epilogues 4 % (their
// ~10 us of stores per tile stay exposed: one workgroup per CU), the bf16-storing one 12 % (16-byte stores of fragment pairs).
//
// Results: every accumulator is bit-identical to the 32 x 32 x 16 kernel's (z and the bf16 h compare equal on the device;
// measured, not assumed: tools/lab_bf16_nt.hip on 262 144 x 3072 x 3072, tests/test_hip_fwd_bf16.py::test_mfma16_matches_mfma32);
// the row dots and the BatchNorm column partials reduce in another order (last-ulp differences).  pn_set_bf16_mfma16(0) selects
// the 32 x 32 x 16 kernel (A/B, the tests).
#pragma once
#include "bwd_bf16_dz.hpp"

namespace pn {

// 16-byte global store
__device__ __forceinline__ void st16(float* ptr, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(ptr) = make_float4(a, b, c, d);
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
// 16 (4 J + a) + 4 c .. + 3 of its row: one 16-byte store per lane, 256 contiguous bytes per row and instruction.  (E_STORE_H16 goes
// through the swapped roles and gemm_epilogue_m16s_h16 below.)  N % 256 == 0 (launcher), so there is no column bound to check.
template <int EK, int WAVES_M, int WAVES_N, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_m16(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0, int tile_n, float* smem) {
  static_assert(EK == E_STORE || EK == E_ROWDOT, "E_STORE_H16 takes the swapped roles");
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
      const bool act = store_act;
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
            st16(p.C + (long)row * p.ldc + col, x[0], x[1], x[2], x[3]);
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

// Epilogue of E_STORE_H16 for the SWAPPED operand roles (the MFMA is given the weight fragment as its row operand): accumulator of
// fragment (i, j), lane l: ROW l % 16 of the 16-row fragment i, COLUMNS 4 (l / 16) + e of fragment j - four consecutive columns per
// lane without a cross-lane move.  As bf16 that is 8 bytes per lane and 32 contiguous bytes per row, which the store path handles
// badly (~5 B per clock and CU: 15 us per tile).  So two fragments go together: v_permlane16_swap exchanges the odd 16-lane rows of
// fragment j's packed columns with the even rows of fragment j + 1's, after which a lane of an even row holds columns
// 4 cq .. + 7 of fragment j and a lane of an odd row columns 4 (cq - 1) .. + 7 of fragment j + 1: ONE 16-byte store, 64 contiguous
// bytes per row and instruction - 4.33 -> 3.97 ms per 262 144-row launch, 0.23 ms above the storeless row-dot kernel (four
// fragments per store group, 128 contiguous bytes: the same).  Same values in the same places: bit-identical output.
template <int WAVES_M, int WAVES_N, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_m16s_h16(const GemmParams& p, f32x4 (&acc)[FM][FN], int row0, int col0) {
  static_assert(FN % 2 == 0, "fragments are stored in pairs");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int cq = lane >> 4;  // which 4-column group of a 16-column fragment
  const int rl = lane & 15;
  const int colw = col0 + wn * FN * 16 + 4 * cq;  // + 16 j + e: this lane's columns before the exchange
#pragma unroll
  for (int j = 0; j < FN; j += 2) {
    const float4 es0 = ld4(p.e_scale + colw + 16 * j), et0 = ld4(p.e_shift + colw + 16 * j);
    const float4 es1 = ld4(p.e_scale + colw + 16 * j + 16), et1 = ld4(p.e_shift + colw + 16 * j + 16);
    const int col = col0 + wn * FN * 16 + 16 * (j + (cq & 1)) + 4 * (cq & ~1);  // after the exchange
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = row0 + (wm * FM + i) * 16 + rl;
      const uint32_t x0 = round2(relu(fmaf(acc[i][j][0], es0.x, et0.x)), relu(fmaf(acc[i][j][1], es0.y, et0.y)));
      const uint32_t x1 = round2(relu(fmaf(acc[i][j][2], es0.z, et0.z)), relu(fmaf(acc[i][j][3], es0.w, et0.w)));
      const uint32_t y0 = round2(relu(fmaf(acc[i][j + 1][0], es1.x, et1.x)), relu(fmaf(acc[i][j + 1][1], es1.y, et1.y)));
      const uint32_t y1 = round2(relu(fmaf(acc[i][j + 1][2], es1.z, et1.z)), relu(fmaf(acc[i][j + 1][3], es1.w, et1.w)));
      const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);  // (every lane takes part: no divergence above)
      const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      if (row < p.M) {
        typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u32x4_*>(reinterpret_cast<uint16_t*>(p.C) + (long)row * p.ldc + col) = u32x4_{s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
}

// Geometry: eight waves of 64 x 128 (4 x 8 fragments of 16 x 16), a slab (BK = 64) is two k-steps of 32: 12 fragment reads
// (ds_read_b128: lane l = row l % 16 of the fragment, granule 4 kk + l / 16 of the slab row) for 32 MFMAs.
// Loop, rotated at the barrier:   barrier(s) | read (s, 0) -> F | A pieces of slab s + 1 | 32 MFMAs of (s - 1, 1) from G |
//                                 W pieces of slab s + 1 | read (s, 1) -> G | 32 MFMAs of (s, 0) from F | wait, barrier(s + 1)
// (of six placements of the eight DMA pieces this one was the fastest: the streamed operand first, a whole slab to land).
// SWAP (E_STORE_H16): the weight fragment is the MFMA's row operand - the accumulator is then transposed (gemm_epilogue_m16s_h16:
// four consecutive columns per lane); measured slower for the f32 store (64-byte row segments against the quad transpose's 256).
template <int EK>
__global__ __launch_bounds__(512, 2) void gemm_nt_bf16m16_kernel(const GemmParams p) {
  constexpr int WAVES_M = 4, WAVES_N = 2, FM = 4, FN = 8;
  constexpr int BM = 256, BN = 256;
  constexpr bool SWAP = (EK == E_STORE_H16);
  constexpr unsigned SLABB = 128u;          // bytes of one tile row per slab (64 bf16)
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

  // DMA sources as in gemm_nt_bf16dma_kernel: wave w, piece q covers tile rows 8 (4 w + q) .. + 7; lane l: row + l / 8, LDS
  // granule position l % 8 holds source granule (l % 8) ^ ((row >> 1) & 7)
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
  auto issue_b = [&](int s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const float* src = reinterpret_cast<const float*>(w_tile + (long)s * SLABB);
    const unsigned base = lds0 + (2 + BUF) * TILEB + (unsigned)wave * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16s(src, boff[q], __builtin_amdgcn_readfirstlane(base + q * 1024u));
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragment reads: lane l takes row l % 16 of a 16-row fragment and granule 4 kk + l / 16 of the slab, stored at position
  // (4 kk + l / 16) ^ ((row >> 1) & 7); fragment rows start at multiples of 16, so (row >> 1) & 7 == (l >> 1) & 7.  Conflict-free:
  // the 16 lanes a ds_read_b128 services per cycle hit 16 different (row parity, position) slots of the 256-byte bank window.
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
        if constexpr (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
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
  // slab s out of buffer CUR, entered behind barrier(s); PEND: the previous slab's second k-step is still to be issued from G
  auto slab = [&](int s, auto cur_c, auto pend_c) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr bool PEND = decltype(pend_c)::value != 0;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    const int nxt = s + 1 < nslab ? s + 1 : s;  // past the end the last slab is staged again into the idle buffer
    read_frag(C{}, I0{}, fa, fb);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(nxt, N{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PEND) mma(ga, gb);
    __builtin_amdgcn_sched_barrier(0);
    issue_b(nxt, N{});
    __builtin_amdgcn_sched_barrier(0);
    read_frag(C{}, I1{}, ga, gb);
    __builtin_amdgcn_sched_barrier(0);
    mma(fa, fb);
    __builtin_amdgcn_sched_barrier(0);
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
  if constexpr (SWAP) gemm_epilogue_m16s_h16<WAVES_M, WAVES_N, FM, FN>(p, acc, row0, col0);
  else gemm_epilogue_m16<EK, WAVES_M, WAVES_N, FM, FN>(p, acc, row0, col0, tile_n, smem);
}

}  // namespace pn
