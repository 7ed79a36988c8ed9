// "bf16x3" NT GEMM for gfx950: f32 operands are split on the fly into hi = bf16(x), lo = bf16(x - hi) and every
// product is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on v_mfma_f32_32x32x16_bf16 with f32 accumulation
// (the dropped a_lo*b_lo term and the split residuals are ~2^-17 relative, i.e. ~1e-5 per product instead of f32's
// 6e-8; 3 MFMAs on the 16x faster bf16 pipe = 5.3x the f32-MFMA rate).  Opt-in (pn_set_math_mode(1)): the default
// path stays the exact-f32 engine of gemm_engine.hpp.  Same operand generators / epilogues / tile order as there.
//
// Restrictions (the pair-grid GEMMs of the hot path meet them): one K segment, K % 32 == 0, N % BN == 0.
// LDS row = 32 k-values as 4 groups of [hi k0..7 (16 B)][lo k0..7 (16 B)] + 16 B pad = 144 B: a thread owns 8
// consecutive k of a row (two float4 global loads), so both planes are written with ds_write_b128 and a fragment
// (row = lane % 32, k-group = 2*kstep + lane / 32) is one ds_read_b128 per plane - no register shuffling, and the
// 144 B stride keeps every 16-lane group on 16 distinct 16-byte slots for reads and writes alike.
#pragma once
#include "gemm_engine.hpp"
#include "gemm_tn.hpp"

namespace pn {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HiLo {
  bf16x4 hi, lo;
};

__device__ __forceinline__ HiLo split4(float4 v) {
  const f32x4 x = {v.x, v.y, v.z, v.w};
  HiLo r;
  r.hi = __builtin_convertvector(x, bf16x4);                                  // round to nearest even
  r.lo = __builtin_convertvector(x - __builtin_convertvector(r.hi, f32x4), bf16x4);  // x - hi is exact in f32
  return r;
}

template <int AK, int EK, int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, PN_MINW) void gemm_nt_bf16x3_kernel(const GemmParams p) {
  constexpr int BK = 32;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 32;
  constexpr int BN = WAVES_N * WN * 32;
  constexpr int LDK = BK + 4;   // row stride in 4-byte units (144 B)
  constexpr int KV = BK / 8;    // threads per tile row (8 k-values each)
  constexpr int RPP = NT / KV;  // tile rows covered per pass of the workgroup
  constexpr int NQA = BM / RPP;
  constexpr int NQB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/thread mismatch");
  static_assert(AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "operand kind not built for bf16x3");
  constexpr int STAGE = (BM + BN) * LDK;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  int tile_m, tile_n;
  if (!tile_coords<BM, BN>(p, tile_m, tile_n)) return;
  const int row0 = tile_m * BM;
  const int col0 = tile_n * BN;

  const int kv = tid % KV;
  const int r_in = tid / KV;

  const float* arow[NQA];
  const float* arow2[NQA];
#pragma unroll
  for (int q = 0; q < NQA; ++q) {
    int r = row0 + r_in + q * RPP;
    if (r > p.M - 1) r = p.M - 1;  // clamp: duplicates are discarded by the epilogue
    if constexpr (AK == A_PAIRSUM_RELU) {
      const int j = r / p.pairB;
      const int i = r - j * p.pairB;
      arow[q] = p.A + (long)i * p.lda + 8 * kv;
      arow2[q] = p.A2 + (long)j * p.lda2 + 8 * kv;
    } else {
      arow[q] = p.A + (long)r * p.lda + 8 * kv;
      arow2[q] = nullptr;
    }
  }
  const float* brow[NQB];
#pragma unroll
  for (int q = 0; q < NQB; ++q) brow[q] = p.W + (long)(col0 + r_in + q * RPP) * p.ldw + 8 * kv;

  const int nslab = p.Kseg / BK;

  float4 ra[NQA][2], ra2[NQA][2], rb[NQB][2];
  float4 rsc[2], rsh[2];

  auto fetch = [&](int s) {
    const int c = s * BK;
#pragma unroll
    for (int q = 0; q < NQA; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[q][h] = ld4(arow[q] + c + 4 * h);
        if constexpr (AK == A_PAIRSUM_RELU) ra2[q][h] = ld4(arow2[q] + c + 4 * h);
      }
    if constexpr (AK == A_AFFINE_RELU) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        rsc[h] = ld4(p.a_scale + c + 8 * kv + 4 * h);
        rsh[h] = ld4(p.a_shift + c + 8 * kv + 4 * h);
      }
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) rb[q][h] = ld4(brow[q] + c + 4 * h);
  };

  auto pin_fetched = [&]() {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pin4(ra[q][h]);
        if constexpr (AK == A_PAIRSUM_RELU) pin4(ra2[q][h]);
      }
#pragma unroll
    for (int q = 0; q < NQB; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) pin4(rb[q][h]);
  };

  auto store8 = [](float* dst, float4 v0, float4 v1) {
    const HiLo a = split4(v0), b = split4(v1);
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = a.hi[e];
      hi[4 + e] = b.hi[e];
      lo[e] = a.lo[e];
      lo[4 + e] = b.lo[e];
    }
    *reinterpret_cast<bf16x8*>(dst) = hi;
    *reinterpret_cast<bf16x8*>(dst + 4) = lo;
  };

  auto commit = [&](int buf) {
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * LDK;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      float4 v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[h] = ra[q][h];
        if constexpr (AK == A_AFFINE_RELU) {
          v[h].x = relu(fmaf(v[h].x, rsc[h].x, rsh[h].x));
          v[h].y = relu(fmaf(v[h].y, rsc[h].y, rsh[h].y));
          v[h].z = relu(fmaf(v[h].z, rsc[h].z, rsh[h].z));
          v[h].w = relu(fmaf(v[h].w, rsc[h].w, rsh[h].w));
        } else if constexpr (AK == A_PAIRSUM_RELU) {
          v[h].x = relu(v[h].x + ra2[q][h].x);
          v[h].y = relu(v[h].y + ra2[q][h].y);
          v[h].z = relu(v[h].z + ra2[q][h].z);
          v[h].w = relu(v[h].w + ra2[q][h].w);
        }
      }
      store8(As + (r_in + q * RPP) * LDK + 8 * kv, v[0], v[1]);
    }
#pragma unroll
    for (int q = 0; q < NQB; ++q) store8(Bs + (r_in + q * RPP) * LDK + 8 * kv, rb[q][0], rb[q][1]);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_row = lane & 31;
  const int frag_g = lane >> 5;  // which 8-k group of the 16-k step

  auto compute = [&](int buf, auto ks_c) {
    constexpr int KS = decltype(ks_c)::value;
    const float* As = smem + buf * STAGE + (wm * WM * 32 + frag_row) * LDK + (2 * KS + frag_g) * 8;
    const float* Bs = smem + buf * STAGE + BM * LDK + (wn * WN * 32 + frag_row) * LDK + (2 * KS + frag_g) * 8;
    bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK);
      al[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK + 4);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK);
      bl[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK + 4);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };

  using std::integral_constant;
  fetch(0);
  pin_fetched();
  commit(0);
  __syncthreads();
  for (int s = 0; s + 1 < nslab; ++s) {
    const int cur = s & 1;
    fetch(s + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(cur, integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    pin_fetched();
    compute(cur, integral_constant<int, 1>{});
    commit(cur ^ 1);
    __syncthreads();
  }
  compute((nslab - 1) & 1, integral_constant<int, 0>{});
  compute((nslab - 1) & 1, integral_constant<int, 1>{});
  __syncthreads();

  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16x3 "TN" contraction over rows (weight gradients):  Cpart[split][m][n] = sum_r A[r][m] * Bgen[r][n].
// Both operands are K-major in HBM, the MFMA wants 8 consecutive k per lane: a thread owns ONE column and loads
// 8 consecutive rows of it (a wave-load = 64 consecutive floats of one row), so the k-packing is a register-local
// transpose and the LDS image is the same [col][4 x (hi8 | lo8)] as in the NT kernel above - the fragment reads and
// the MFMA block are identical.  256x256 output tile, 8 waves (4 x 2, each 64 x 128), split-K partial tiles summed
// in order by k_splitk_reduce.  Restrictions: M % 256 == 0, N % 256 == 0, TA_PLAIN, pairB % 8 == 0 for PAIRSUM.
// ---------------------------------------------------------------------------------------------------------------
template <int TB>
__global__ __launch_bounds__(512, PN_MINW) void gemm_tn_bf16x3_kernel(const TnParams p) {
  constexpr int BM = 256, BN = 256, BK = 32, LDK = BK + 4;
  constexpr int WM = 2, WN = 4;
  constexpr int STAGE = (BM + BN) * LDK;
  static_assert(TB == TB_PLAIN || TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU, "operand kind not built for bf16x3");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = p.N / BN, ntm = p.M / BM;
  int tile_m, tile_n;
  if (PN_XCD && (ntm % 2 == 0) && (ntn % 4 == 0)) {  // same XCD regions as gemm_tn_kernel
    const int rm = ntm / 2, rn = ntn / 4;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    tile_m = (xcd >> 2) * rm + w / rn;
    tile_n = (xcd & 3) * rn + w % rn;
  } else {
    tile_n = blockIdx.x % ntn;
    tile_m = blockIdx.x / ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int split = blockIdx.y;
  const long r_begin = (long)split * p.rows_per_split;
  long r_end = r_begin + p.rows_per_split;
  if (r_end > p.R) r_end = p.R;

  const int col = tid & 255;  // the column of both tiles this thread stages
  const int kg = tid >> 8;    // it stages k-groups kg and kg + 2 (8 rows each)
  float bs = 0.f, bt = 0.f;
  if constexpr (TB == TB_AFFINE_RELU) {
    bs = p.b_s[n0 + col];
    bt = p.b_t[n0 + col];
  }

  float ra[2][8], rb[2][8], rb2[2];
  unsigned pi0 = 0, pj0 = 0;  // pair decode (i = r % B, j = r / B) of this thread's first row, carried along
  const unsigned pB = (unsigned)p.pairB;
  if constexpr (TB == TB_PAIRSUM_RELU) {
    const unsigned ru = (unsigned)(r_begin + 8 * kg);
    pj0 = ru / pB;
    pi0 = ru - pj0 * pB;
  }

  auto fetch = [&](long k0, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;
    const float* Ak = p.A + k0 * p.lda + m0 + col;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int rr0 = 8 * (kg + 2 * g);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        long ro = rr0 + e;
        if (MASKED && k0 + ro >= r_end) ro = r_begin - k0;
        ra[g][e] = Ak[ro * p.lda];
      }
      if constexpr (TB == TB_PAIRSUM_RELU) {
        unsigned i = pi0 + 16 * g, j = pj0;
        while (i >= pB) {
          i -= pB;
          ++j;
        }
        if (MASKED && k0 + rr0 >= r_end) i = 0, j = 0;
        const float* Bi = p.B + (long)i * p.ldb + n0 + col;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          long eo = e;
          if (MASKED && k0 + rr0 + e >= r_end) eo = 0;
          rb[g][e] = Bi[eo * p.ldb];
        }
        rb2[g] = p.B2[(long)j * p.ldb2 + n0 + col];
      } else {
        const float* Bk = p.B + k0 * p.ldb + n0 + col;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          long ro = rr0 + e;
          if (MASKED && k0 + ro >= r_end) ro = r_begin - k0;
          rb[g][e] = Bk[ro * p.ldb];
        }
      }
    }
    if constexpr (TB == TB_PAIRSUM_RELU) {
      pi0 += BK;
      while (pi0 >= pB) {
        pi0 -= pB;
        ++pj0;
      }
    }
  };

  auto pin_fetched = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(ra[g][e]), "+v"(rb[g][e]));
      if constexpr (TB == TB_PAIRSUM_RELU) asm volatile("" : "+v"(rb2[g]));
    }
  };

  auto store8 = [](float* dst, const float (&v)[8]) {
    const HiLo a = split4(make_float4(v[0], v[1], v[2], v[3])), b = split4(make_float4(v[4], v[5], v[6], v[7]));
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = a.hi[e];
      hi[4 + e] = b.hi[e];
      lo[e] = a.lo[e];
      lo[4 + e] = b.lo[e];
    }
    *reinterpret_cast<bf16x8*>(dst) = hi;
    *reinterpret_cast<bf16x8*>(dst + 4) = lo;
  };

  auto commit = [&](int buf, long k0, auto masked_c) {
    constexpr bool MASKED = decltype(masked_c)::value;
    float* As = smem + buf * STAGE;
    float* Bs = As + BM * LDK;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float a[8], b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] = ra[g][e];
        b[e] = rb[g][e];
        if constexpr (TB == TB_AFFINE_RELU) b[e] = relu(fmaf(b[e], bs, bt));
        if constexpr (TB == TB_PAIRSUM_RELU) b[e] = relu(b[e] + rb2[g]);
        if (MASKED && k0 + 8 * (kg + 2 * g) + e >= r_end) a[e] = 0.f, b[e] = 0.f;
      }
      store8(As + col * LDK + 8 * (kg + 2 * g), a);
      store8(Bs + col * LDK + 8 * (kg + 2 * g), b);
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
  auto compute = [&](int buf, auto ks_c) {
    constexpr int KS = decltype(ks_c)::value;
    const float* As = smem + buf * STAGE + (wm * WM * 32 + frag_row) * LDK + (2 * KS + frag_g) * 8;
    const float* Bs = smem + buf * STAGE + BM * LDK + (wn * WN * 32 + frag_row) * LDK + (2 * KS + frag_g) * 8;
    bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK);
      al[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK + 4);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK);
      bl[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK + 4);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };

  using std::integral_constant;
  using std::false_type;
  using std::true_type;
  if (r_begin < r_end) {
    const bool first_full = r_begin + BK <= r_end;
    if (first_full) {
      fetch(r_begin, false_type{});
      pin_fetched();
      commit(0, r_begin, false_type{});
    } else {
      fetch(r_begin, true_type{});
      pin_fetched();
      commit(0, r_begin, true_type{});
    }
    __syncthreads();
    int cur = 0;
    long k0 = r_begin;
    for (; k0 + 2 * BK <= r_end; k0 += BK) {  // the next slab is a full one
      fetch(k0 + BK, false_type{});
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      pin_fetched();
      compute(cur, integral_constant<int, 1>{});
      commit(cur ^ 1, k0 + BK, false_type{});
      __syncthreads();
      cur ^= 1;
    }
    if (k0 + BK < r_end) {  // ragged last slab
      fetch(k0 + BK, true_type{});
      compute(cur, integral_constant<int, 0>{});
      compute(cur, integral_constant<int, 1>{});
      pin_fetched();
      commit(cur ^ 1, k0 + BK, true_type{});
      __syncthreads();
      cur ^= 1;
    }
    compute(cur, integral_constant<int, 0>{});
    compute(cur, integral_constant<int, 1>{});
  }

  float* out = p.Cpart + (long)split * p.M * p.ldc;
  const int hl = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + (wn * WN + j) * 32 + cl;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
        out[(long)m * p.ldc + n] = acc[i][j][e];
      }
    }
}

}  // namespace pn
