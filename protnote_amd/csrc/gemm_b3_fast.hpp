// bf16x3 NT kernel of the pair-grid GEMMs, rewritten like gemm_nt_dma_kernel so that the slab loop spends vector
// instructions only on what cannot be avoided - the operand transform and the hi / lo split of the generated A operand.
// PMC on gemm_nt_bf16x3_kernel<.., BDMA>: 3.5 VALU instructions per MFMA, matrix pipe 62-70 % busy; a 32x32x16 bf16 MFMA
// holds the pipe for 32 cycles and every other VALU instruction of the SIMD's two waves takes ~6 of them (measured on the
// f32 kernels, gemm_dma.hpp), i.e. the bf16x3 kernels were VALU-issue bound before they were power bound.  Here:
//   * addresses: SGPR bases + loop-invariant lane offsets (buffer loads for A, SGPR-base LDS-DMA for the W planes);
//   * the loop is unrolled over the two LDS stages (A0 | A1 | B0 | B1): every ds offset is an immediate;
//   * relu(s z + t) / relu(a + b) with packed f32 ops; the split costs 5 instead of 6 instructions per pair
//     (v_cvt_pk_bf16_f32, shift, and, ONE packed subtract, v_cvt_pk_bf16_f32).
// Same LDS images, same MFMA order and the same split arithmetic as gemm_nt_bf16x3_kernel<.., BDMA>: bit-identical.
#pragma once
#include "gemm_bf16x3.hpp"
#include "gemm_tn_fast.hpp"

namespace pn {

__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// (x0, x1) -> packed bf16 hi pair and lo pair (hi = RNE(x), lo = RNE(x - hi); x - hi is exact in f32)
__device__ __forceinline__ void split_pair(f32x2 x, uint32_t& hi, uint32_t& lo) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2_));
  const f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(pk_sub(x, hf), bf16x2_));
}
__device__ __forceinline__ void pin_f4(f32x4& v) { asm volatile("" : "+v"(v)); }
typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_write_u4(unsigned addr, u32x4_ v) { *reinterpret_cast<PN_LDS u32x4_*>(addr) = v; }
__device__ __forceinline__ bf16x8 lds_read_b8(unsigned addr) { return *reinterpret_cast<const PN_LDS bf16x8*>(addr); }

template <int AK, int EK>
__global__ __launch_bounds__(512, 2) void gemm_nt_b3_fast_kernel(const GemmParams p) {
  static_assert(AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "operand kind not built for bf16x3");
  constexpr int WAVES_M = 4, WAVES_N = 2, WM = 2, WN = 4, BK = 32, BM = 256, BN = 256;
  constexpr unsigned AROW = 144u;             // bytes of an A image row: 4 x (hi8 | lo8) + 16 pad
  constexpr unsigned ATILE = BM * AROW;       // 36864
  constexpr unsigned BPLANE = BN * 64u;       // 16384: one plane of the B image (64-byte rows)
  constexpr unsigned BTILE = 2 * BPLANE;      // 32768
  constexpr unsigned BBASE = 2 * ATILE;       // LDS: A0 | A1 | B0 | B1
  constexpr int NQA = 2;                      // tile rows per thread (r_in + 128 q), 8 k-values each

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

  // ---- W planes by LDS-DMA: wave w issues chunks c = 4 w + q: plane c / 16, tile rows 16 (c % 16) .. + 15; lane l:
  //      row + l / 4, LDS granule position l % 4 <- source granule (l % 4) ^ ((row >> 2) & 3)
  //      ((row >> 2) & 3 == (l >> 4) & 3 for every chunk, so ONE lane offset serves all four: the chunk's rows are a
  //      uniform 16 * (c % 16) * K further on - scalar address arithmetic)
  const unsigned boff = (unsigned)((long)(lane >> 2) * p.Kseg + 8 * ((lane & 3) ^ ((lane >> 4) & 3))) * 2u;
  const uint16_t* wh_tile = p.w_hi + (long)col0 * p.Kseg;
  const uint16_t* wl_tile = p.w_lo + (long)col0 * p.Kseg;
  auto issue_b = [&](int s, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * wave + q;  // (uniform)
      const uint16_t* src = ((c >> 4) ? wl_tile : wh_tile) + (long)(16 * (c & 15)) * p.Kseg + s * BK;
      glds16s(reinterpret_cast<const float*>(src), boff,
              __builtin_amdgcn_readfirstlane(lds0 + BBASE + BUF * BTILE + (unsigned)(c >> 4) * BPLANE + (unsigned)(c & 15) * 1024u));
    }
  };

  // ---- A through registers: thread (r_in = tid / 4, kv = tid % 4) owns k {4 kv .. + 3} and {16 + 4 kv .. + 3} of rows
  //      r_in and r_in + 128
  const int kv = tid & 3;
  const int r_in = tid >> 2;
  const float* a_tile = p.A + (long)row0 * p.lda;
  unsigned aoff[NQA], aoff2[NQA];
#pragma unroll
  for (int q = 0; q < NQA; ++q) {
    const int rl = r_in + q * 128;
    int r = row0 + rl;
    if (r > p.M - 1) r = p.M - 1;  // clamp: duplicates are discarded by the epilogue
    if constexpr (AK == A_PAIRSUM_RELU) {
      const int j = r / p.pairB;
      const int i = r - j * p.pairB;
      aoff[q] = (unsigned)((long)i * p.lda + 4 * kv) * 4u;
      aoff2[q] = (unsigned)((long)j * p.lda2 + 4 * kv) * 4u;
    } else {
      aoff[q] = (unsigned)((long)(r - row0) * p.lda + 4 * kv) * 4u;
      aoff2[q] = 0u;
    }
  }
  unsigned awr = lds0 + (unsigned)r_in * AROW + 32u * kv;  // (row r_in + 128 q: + q * 128 * AROW, an immediate)
  asm volatile("" : "+v"(awr));
  const unsigned st_off = 16u * kv;
  // TWO register sets: the slab is short (48 MFMAs per wave, ~1.5 us), half of it is not enough to hide an HBM access, so
  // the operand of slab s+3 is requested as soon as the set that held slab s+1 has been consumed - 1.5 slabs of lead
  f32x4 ra[2][NQA][2], ra2[2][NQA][2], rsc[2], rsh[2];
  auto fetch_a = [&](int s, auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const int c = s * BK;
    const float* src = (AK == A_PAIRSUM_RELU ? p.A : a_tile) + c;
#pragma unroll
    for (int q = 0; q < NQA; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[S][q][h] = bload4(src + 16 * h, aoff[q]);
        if constexpr (AK == A_PAIRSUM_RELU) ra2[S][q][h] = bload4(p.A2 + c + 16 * h, aoff2[q]);
      }
  };
  // the BatchNorm scale / shift quads of a slab (8 distinct 16-byte quads per vector, hot in the vector L1): ONE register
  // set, requested half a slab ahead
  auto fetch_st = [&](int s) {
    if constexpr (AK == A_AFFINE_RELU) {
      const int c = s * BK;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        rsc[h] = bload4(p.a_scale + c + 16 * h, st_off);
        rsh[h] = bload4(p.a_shift + c + 16 * h, st_off);
      }
    }
  };
  auto pin_a = [&](auto set_c) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int q = 0; q < NQA; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pin_f4(ra[S][q][h]);
        if constexpr (AK == A_PAIRSUM_RELU) pin_f4(ra2[S][q][h]);
      }
    if constexpr (AK == A_AFFINE_RELU) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pin_f4(rsc[h]);
        pin_f4(rsh[h]);
      }
    }
  };
  auto commit_a = [&](auto buf_c, auto set_c) {
    constexpr int BUF = decltype(buf_c)::value, S = decltype(set_c)::value;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      u32x4_ hi, lo;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x2 v0 = ra[S][q][h].xy, v1 = ra[S][q][h].zw;
        if constexpr (AK == A_AFFINE_RELU) {
          v0 = pk_fma(v0, rsc[h].xy, rsh[h].xy);
          v1 = pk_fma(v1, rsc[h].zw, rsh[h].zw);
        } else if constexpr (AK == A_PAIRSUM_RELU) {
          v0 = pk_add(v0, ra2[S][q][h].xy);
          v1 = pk_add(v1, ra2[S][q][h].zw);
        }
        if constexpr (AK != A_PLAIN) {
          v0 = f32x2{relu_raw(v0.x), relu_raw(v0.y)};
          v1 = f32x2{relu_raw(v1.x), relu_raw(v1.y)};
        }
        uint32_t h0, l0, h1, l1;
        split_pair(v0, h0, l0);
        split_pair(v1, h1, l1);
        hi[2 * h] = h0; hi[2 * h + 1] = h1;
        lo[2 * h] = l0; lo[2 * h + 1] = l1;
      }
      lds_write_u4(awr + (BUF * ATILE + q * 128 * AROW), hi);
      lds_write_u4(awr + (BUF * ATILE + q * 128 * AROW + 16u), lo);
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment (row = lane % 32, k-group = 2 ks + lane / 32): one 16-byte read per plane
  const int frag_row = lane & 31;
  const int frag_g = lane >> 5;
  unsigned fa_addr = lds0 + (unsigned)(wm * WM * 32 + frag_row) * AROW + 32u * frag_g;
  unsigned fb_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    fb_addr[ks] = lds0 + BBASE + (unsigned)(wn * WN * 32 + frag_row) * 64u + 16u * ((2 * ks + frag_g) ^ ((frag_row >> 2) & 3));
  asm volatile("" : "+v"(fa_addr), "+v"(fb_addr[0]), "+v"(fb_addr[1]));

  bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
  auto read_frags = [&](auto buf_c, auto ks_c) {
    constexpr int BUF = decltype(buf_c)::value, KS = decltype(ks_c)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      ah[i] = lds_read_b8(fa_addr + (BUF * ATILE + i * 32 * AROW + KS * 64u));
      al[i] = lds_read_b8(fa_addr + (BUF * ATILE + i * 32 * AROW + KS * 64u + 16u));
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      bh[j] = lds_read_b8(fb_addr[KS] + (BUF * BTILE + j * 2048u));
      bl[j] = lds_read_b8(fb_addr[KS] + (BUF * BTILE + j * 2048u + BPLANE));
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };
  auto compute = [&](auto buf_c, auto ks_c) {
    read_frags(buf_c, ks_c);
    mma();
  };
  // staging work woven between the MFMAs of a k-step (NV vector instructions and, every third MFMA, one LDS write)
  auto weave = [&](auto nvalu_c) {
    constexpr int NV = decltype(nvalu_c)::value;
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (WM + WN), 0);
#pragma unroll
    for (int i = 0; i < WM * WN * 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      if (i % 6 == 5) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 4 LDS writes per region
    }
  };

  using std::integral_constant;
  using I0 = integral_constant<int, 0>;
  using I1 = integral_constant<int, 1>;
  const int lasts = nslab - 1;
  issue_b(0, I0{});
  fetch_a(0, I0{});
  fetch_st(0);
  pin_a(I0{});
  commit_a(I0{}, I0{});
  fetch_st(lasts < 1 ? lasts : 1);
  fetch_a(lasts < 1 ? lasts : 1, I1{});  // A(1) -> set 1, A(2) -> set 0 (indices past the end re-read the last slab)
  fetch_a(lasts < 2 ? lasts : 2, I0{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the B DMA of slab 0; the two A sets land with it - once per tile)
  __builtin_amdgcn_s_barrier();
  // slab s (not the last) out of stage CUR: A(s+1) - register set N, requested 1.5 slabs ago - is transformed, split and
  // written under k-step 0's MFMAs while the DMA of the W planes of slab s+1 lands (W is L2-resident: half a slab is
  // enough); then A(s+3) is requested into the set just freed and k-step 1 runs.
  auto slab = [&](int s, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    using C = integral_constant<int, CUR>;
    using N = integral_constant<int, CUR ^ 1>;
    __builtin_amdgcn_sched_barrier(0);
    pin_a(N{});  // (first: hipcc's counted waits for the A registers must not see the DMA as younger traffic)
    read_frags(C{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    issue_b(s + 1, N{});  // the idle stage was last read in slab s-1 (barrier passed)
    __builtin_amdgcn_sched_barrier(0);
    mma();
    commit_a(N{}, N{});
#ifdef PN_B3_WEAVE
    weave(integral_constant<int, (AK == A_PLAIN ? 2 : 3)>{});
#endif
    __builtin_amdgcn_sched_barrier(0);
    fetch_st(s + 2 < lasts ? s + 2 : lasts);  // (before A(s+3): vmcnt is in order, the next slab must be able to wait
    fetch_a(s + 3 < lasts ? s + 3 : lasts, N{});  //  for these quads without waiting for the younger A(s+3))
    __builtin_amdgcn_sched_barrier(0);
    compute(C{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    // this wave's share of the B(s+1) DMA must have landed before the barrier publishes it.  vmcnt counts in order: the
    // only requests younger than the DMA are the NA loads of A(s+3) just issued, so "at most NA outstanding" means the DMA
    // (and the older A(s+2) set) is complete while A(s+3) stays in flight across the barrier.
    constexpr int NA = NQA * 2 * (AK == A_PAIRSUM_RELU ? 2 : 1) + (AK == A_AFFINE_RELU ? 4 : 0);
    // (the builtin, not inline asm: the compiler's own wait-count bookkeeping then knows that no LDS operation is pending
    //  at the loop edge - otherwise it answers the unknown with a full lgkmcnt(0) in front of the next slab's first MFMA
    //  instead of counted waits as the fragments arrive)
    __builtin_amdgcn_s_waitcnt((NA & 0xF) | (0x7 << 4) | (0 << 8) | ((NA >> 4) << 14));
    __builtin_amdgcn_s_barrier();
  };
  int s = 0;
  for (; s + 2 < nslab; s += 2) {
    slab(s, I0{});
    slab(s + 1, I1{});
  }
  if (s + 1 < nslab) {  // one more staged slab, the last one then sits in stage 1
    slab(s, I0{});
    compute(I1{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    compute(I1{}, I1{});
  } else {
    compute(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    compute(I0{}, I1{});
  }
  __syncthreads();

  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

constexpr int B3_FAST_LDS_BYTES = 2 * (256 * 144 + 2 * 256 * 64);  // 139264

}  // namespace pn
