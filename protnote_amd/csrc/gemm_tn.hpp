// f32-MFMA "TN" GEMM engine for gfx950: weight-gradient contraction over the (huge) row dimension,
//     Cpart[split][m][n] = sum_{r in split}  Agen[r][m] * Bgen[r][n]
// Both operands are row-major [R][cols] in HBM, i.e. K-major for this contraction; their tiles are staged
// global -> registers (generator applied: BN/ReLU backward for A, BN+ReLU forward recompute for B) ->
// LDS [BK][cols+4] and read as single dwords (lane l: k = l>>5, col = l&31 -> 32 consecutive banks, conflict
// free).  Split-K over row ranges; partial tiles are written (not atomically added) to Cpart and summed in
// split order by k_splitk_reduce, so weight gradients are bit-reproducible.
// Grid: x = output tiles (n fastest), y = row split, so all tiles of one row range are co-scheduled and the
// streamed dz / activation rows are shared through L2 / Infinity Cache.
#pragma once
#include "gemm_engine.hpp"

namespace pn {

enum { TA_PLAIN = 0, TA_DZ_ELEM = 1, TA_DZ_ROWG = 2 };
enum { TB_PLAIN = 0, TB_AFFINE_RELU = 1, TB_PAIRSUM_RELU = 2, TB_PAIRPROD = 3, TB_CONVTAP = 4 };
// 3: B[r % pairB] * B2[r / pairB];  4: one tap of the masked dilated-conv input (weight gradient of MaskedConv1D):
//    row p = (b, t) reads B[p + shift] when 0 <= t + shift < len[b] (else 0), then relu(b_s*x + b_t) if b_s != null

struct TnParams {
  long R;               // contraction extent (rows)
  long rows_per_split;  // multiple of BK
  int M, N;             // output tile space: m indexes A columns, n indexes B columns
  // ---- A generator ----
  const float* A;  // TA_PLAIN: the matrix; TA_DZ_*: z (pre-activation)
  long lda;
  const float* G;  // TA_DZ_ELEM: upstream gradient matrix
  long ldg;
  const float* gvec;  // TA_DZ_ROWG: per-row upstream scalar
  const float* m_s;   // per-m vectors of the dz generator (see gemm_engine.hpp A_DZ_*)
  const float* m_t;
  const float* m_cs;
  const float* m_p;
  const float* m_q;
  // ---- B generator ----
  const float* B;  // TB_PLAIN: X; TB_AFFINE_RELU: relu(b_s*Y + b_t); TB_PAIRSUM_RELU: relu(B[r % pairB] + B2[r / pairB])
  long ldb;
  const float* b_s;
  const float* b_t;
  const float* B2;
  long ldb2;
  int pairB;
  const int* lens;  // TB_CONVTAP: int32 sequence lengths
  int L;            //             positions per sequence
  int shift;        //             (tap - k/2) * dilation
  // ---- inter-layer dropout of the B operand (DROP kernels; same hash as GemmParams::drop_*): B[r][n] *= keep ? scale : 0
  uint32_t drop_seed, drop_thresh;
  float drop_scale;
  // ---- output ----
  float* Cpart;  // [nsplit][M][ldc]
  long ldc;
  long row_base;       // generic kernel: first row of split 0 (a tail launch over rows [row_base, R) of a bigger contraction)
  int* task_sync;  // optional [tasks][4] arrival counters (zeroed by the launcher): keeps a task's workgroups in step
  int task_ns;   // > 0: 1-D grid of 32-workgroup region tasks over a 12 x 12 tile grid with task_ns row splits (see kernel)
};

// one LDS-DMA wave-instruction (global_load_lds_dwordx4): lane l copies 16 bytes from its own global address to
// LDS[lds_base + 16 l]; issued from inline asm, so its completion is the kernel's own business (s_waitcnt vmcnt)
__device__ __forceinline__ void tn_glds16(const float* gsrc, unsigned lds_base_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base_uniform)
      : "memory");
}

// Region tasks (12 x 12 tiles = the 3072 x 3072 weight gradients; 1-D grid).  Workgroup id -> XCD is id % 8 and an XCD
// has 32 CUs, one 256x256 workgroup each: ids {256 g + 8 i + x, i < 32} are the 32 workgroups XCD x runs TOGETHER in
// round g, and they finish together (same work), so round g+1's 32 start together too.  Each such task is one region of
// ONE row split whose operand panels the 32 workgroups stream in step through that XCD's 4 MB L2:
//   kind 0/1/3: 4 x 8 tiles (4 dz panels + 8 activation panels), kind 2: 8 x 4, and the 4 x 4 corners of two splits
//   share a task.  Per split 4 * 12 + 8 = 56 panel reads for 144 tiles (2.3x the 24 of a perfect cache) - and the
// sharing no longer depends on which workgroups happen to be co-resident (with one 6 x 3 region per XCD and split,
// 18 + 14 workgroups of two splits shared an XCD and drifted apart: fetch 0.7 or 1.6 TB per launch, measured
// 0.73 -> 0.49 TB and L2 hit rate 0.70 -> 0.80, profiles/r02_tn_region_tasks_ab.json).  false = padding workgroup.
__device__ __forceinline__ bool tn_task_coords(int task_ns, int& tile_m, int& tile_n, int& split) {
  const int lid = blockIdx.x, xcd = lid & 7, slot = (lid & 255) >> 3;
  const int T = (lid >> 8) * 8 + xcd, nfull = task_ns * 4;
  if (T < nfull) {
    split = T >> 2;
    const int kind = T & 3;
    if (kind == 2) {
      tile_m = slot >> 2;
      tile_n = 8 + (slot & 3);
    } else {
      tile_m = (kind == 0 ? 0 : (kind == 1 ? 4 : 8)) + (slot >> 3);
      tile_n = slot & 7;
    }
    return true;
  }
  split = 2 * (T - nfull) + (slot >> 4);
  if (split >= task_ns) return false;
  tile_m = 8 + ((slot & 15) >> 2);
  tile_n = 8 + (slot & 3);
  return true;
}
inline unsigned tn_task_grid(int ns) { return (unsigned)((ns * 4 + (ns + 1) / 2 + 7) / 8 * 256); }

// BIG = false: 128x128 output tile, 4 waves (2x2, each 64x64), 2 workgroups/CU.
// BIG = true : 256x256 output tile, 8 waves (4x2, each 64x128), 1 workgroup/CU - half the operand traffic per flop.
// ADMA (BIG, TA_PLAIN, every split a whole number of slabs): the plain A operand (the materialised dz) goes global ->
// LDS by LDS-DMA, one 1 KiB tile row per wave-instruction, instead of through registers; the generated B operand keeps
// the register path.  A(t+1) is issued at the top of slab t into the idle buffer and waited for (together with the
// B(t+1) registers) in the middle of the slab; the barrier at the end of the slab publishes it.
template <int TA, int TB, bool BIG, bool ADMA = false, bool DROP = false>
__global__ __launch_bounds__(BIG ? 512 : 256, 2) void gemm_tn_kernel(const TnParams p) {
  static_assert(!ADMA || (BIG && TA == TA_PLAIN), "LDS-DMA staging is for the plain A operand of the big tile");
  static_assert(!DROP || TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU, "dropout applies to the hidden activations");
  constexpr int BM = BIG ? 256 : 128, BN = BIG ? 256 : 128, BK = 32;
  constexpr int NGN = BIG ? 2 : 1;        // 64-column groups per wave along n
  constexpr int C4 = BM / 4;              // float4 per tile row (BM == BN)
  // (an LDS-DMA row must be contiguous: no padding - the 8-byte fragment reads and the 16-byte row-contiguous
  //  writes are conflict-free either way)
  constexpr int LDM = ADMA ? BM : BM + 4, LDN = ADMA ? BN : BN + 4;
  constexpr int STAGE = BK * (LDM + LDN);
  constexpr int NQ = 4;  // (BK rows * C4 float4 per row) / threads

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves (small) or 4x2 waves (BIG)

  const int ntn = (p.N + BN - 1) / BN;
  const int ntm = (p.M + BM - 1) / BM;
  int tile_m, tile_n;
  int split = blockIdx.y;
  if (p.task_ns > 0) {
    if (!tn_task_coords(p.task_ns, tile_m, tile_n, split)) return;
  } else if (PN_XCD && (ntm % 2 == 0) && (ntn % 4 == 0)) {
    // XCD-aware order: workgroup x of a split runs on XCD x % 8; give each XCD one (ntm/2) x (ntn/4) region of
    // the tile grid so it streams 1/2 of dz and 1/4 of the activations through its L2 instead of all of dz
    // and 1/8 of the activations (24x24 tiles: 18 instead of 27 distinct operand tiles per slab and XCD).
    const int rm = ntm / 2, rn = ntn / 4;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    tile_m = (xcd >> 2) * rm + w / rn;
    tile_n = (xcd & 3) * rn + w % rn;
  } else {
    tile_n = blockIdx.x % ntn;
    tile_m = blockIdx.x / ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long r_begin = p.row_base + (long)split * p.rows_per_split;
  long r_end = r_begin + p.rows_per_split;
  if (r_end > p.R) r_end = p.R;

  const int c4 = tid & (C4 - 1);  // float4 column within the tile row
  const int rr = tid / C4;        // 0..7: tile row (k) within a pass
  const int am = m0 + 4 * c4;
  const int bn = n0 + 4 * c4;
  const bool a_ok = am < p.M;  // M, N are multiples of 4
  const bool b_ok = bn < p.N;

  // per-column constants of the generators never change along K: keep them in registers
  float4 ms = make_float4(0, 0, 0, 0), mt = ms, mcs = ms, mp = ms, mq = ms, bs = ms, bt = ms;
  if constexpr (TA != TA_PLAIN) {
    if (a_ok) {
      ms = ld4(p.m_s + am);
      mt = ld4(p.m_t + am);
      mcs = ld4(p.m_cs + am);
      mp = ld4(p.m_p + am);
      mq = ld4(p.m_q + am);
    }
  }
  if constexpr (TB == TB_AFFINE_RELU || TB == TB_CONVTAP) {
    if (b_ok && p.b_s != nullptr) {
      bs = ld4(p.b_s + bn);
      bt = ld4(p.b_t + bn);
    }
  }

  float4 ra[NQ], rg[NQ], rb[NQ], rb2[NQ];
  unsigned rowok = 0, rowok_b = 0, tapok = 0;
  uint32_t b_key[NQ];  // DROP: per-row keys of the rows being staged

  // branch-free fetch: rows past the split end are clamped to the split's first row and zeroed by selects in
  // commit().  The pair-grid decode (i = r % B, j = r / B) of the thread's first row is carried incrementally
  // from slab to slab (one division per workgroup lifetime instead of four per slab).
  const int amc = a_ok ? am : 0, bnc = b_ok ? bn : 0;
  const unsigned pB = (unsigned)p.pairB;
  unsigned pi0 = 0, pj0 = 0;
  if constexpr (TB == TB_PAIRSUM_RELU || TB == TB_PAIRPROD) {
    const unsigned ru = (unsigned)(r_begin + rr);
    pj0 = ru / pB;
    pi0 = ru - pj0 * pB;
  }
  auto fetch_a = [&](long k0) {
    rowok = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      long r = k0 + rr + 8 * q;
      const bool ok = r < r_end;
      rowok |= (ok ? 1u : 0u) << q;
      r = ok ? r : r_begin;
      ra[q] = ld4(p.A + r * p.lda + amc);
      if constexpr (TA == TA_DZ_ELEM) rg[q] = ld4(p.G + r * p.ldg + amc);
      if constexpr (TA == TA_DZ_ROWG) {
        const float g = p.gvec[r];
        rg[q] = make_float4(g, g, g, g);
      }
    }
  };
  auto fetch_b = [&](long k0) {
    rowok_b = 0;
    tapok = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      long r = k0 + rr + 8 * q;
      const bool ok = r < r_end;
      rowok_b |= (ok ? 1u : 0u) << q;
      r = ok ? r : r_begin;
      if constexpr (DROP) b_key[q] = drop_rowkey(p.drop_seed, (uint32_t)r);
      if constexpr (TB == TB_PAIRSUM_RELU || TB == TB_PAIRPROD) {
        unsigned i = pi0 + 8 * q, j = pj0;
        while (i >= pB) {  // at most one iteration when B >= 24
          i -= pB;
          ++j;
        }
        i = ok ? i : 0;
        j = ok ? j : 0;
        rb[q] = ld4(p.B + (long)i * p.ldb + bnc);
        rb2[q] = ld4(p.B2 + (long)j * p.ldb2 + bnc);
      } else if constexpr (TB == TB_CONVTAP) {
        const int b = (int)(r / p.L);
        const int t = (int)(r - (long)b * p.L) + p.shift;
        const bool tap_ok = ok && t >= 0 && t < p.lens[b];
        tapok |= (tap_ok ? 1u : 0u) << q;
        rb[q] = ld4(p.B + (tap_ok ? r + p.shift : r_begin) * p.ldb + bnc);
      } else {
        rb[q] = ld4(p.B + r * p.ldb + bnc);
      }
    }
    if constexpr (TB == TB_PAIRSUM_RELU || TB == TB_PAIRPROD) {  // advance the carried decode by one slab
      pi0 += BK;
      while (pi0 >= pB) {
        pi0 -= pB;
        ++pj0;
      }
    }
  };

  auto sel4 = [](bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };

  auto commit_a = [&](int buf) {
    float* As = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      pin4(ra[q]);
      if constexpr (TA != TA_PLAIN) pin4(rg[q]);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = (rowok >> q) & 1u;
      float4 a = ra[q];
      if constexpr (TA != TA_PLAIN) {
        const float4 g = rg[q];
        a.x = (fmaf(a.x, ms.x, mt.x) > 0.f ? g.x * mcs.x : 0.f) + fmaf(mq.x, a.x, mp.x);
        a.y = (fmaf(a.y, ms.y, mt.y) > 0.f ? g.y * mcs.y : 0.f) + fmaf(mq.y, a.y, mp.y);
        a.z = (fmaf(a.z, ms.z, mt.z) > 0.f ? g.z * mcs.z : 0.f) + fmaf(mq.z, a.z, mp.z);
        a.w = (fmaf(a.w, ms.w, mt.w) > 0.f ? g.w * mcs.w : 0.f) + fmaf(mq.w, a.w, mp.w);
      }
      *reinterpret_cast<float4*>(As + (rr + 8 * q) * LDM + 4 * c4) = sel4(ok && a_ok, a);
    }
  };
  auto commit_b = [&](int buf) {
    float* Bs = smem + buf * STAGE + BK * LDM;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      pin4(rb[q]);
      if constexpr (TB == TB_PAIRSUM_RELU || TB == TB_PAIRPROD) pin4(rb2[q]);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const bool ok = (rowok_b >> q) & 1u;
      float4 b = rb[q];
      bool bok = ok && b_ok;
      if constexpr (TB == TB_CONVTAP) {
        bok = b_ok && ((tapok >> q) & 1u);
        if (p.b_s != nullptr) {
          b.x = relu(fmaf(b.x, bs.x, bt.x));
          b.y = relu(fmaf(b.y, bs.y, bt.y));
          b.z = relu(fmaf(b.z, bs.z, bt.z));
          b.w = relu(fmaf(b.w, bs.w, bt.w));
        }
      }
      if constexpr (TB == TB_AFFINE_RELU) {
        b.x = relu(fmaf(b.x, bs.x, bt.x));
        b.y = relu(fmaf(b.y, bs.y, bt.y));
        b.z = relu(fmaf(b.z, bs.z, bt.z));
        b.w = relu(fmaf(b.w, bs.w, bt.w));
      } else if constexpr (TB == TB_PAIRPROD) {
        b.x *= rb2[q].x;
        b.y *= rb2[q].y;
        b.z *= rb2[q].z;
        b.w *= rb2[q].w;
      } else if constexpr (TB == TB_PAIRSUM_RELU) {
        b.x = relu(b.x + rb2[q].x);
        b.y = relu(b.y + rb2[q].y);
        b.z = relu(b.z + rb2[q].z);
        b.w = relu(b.w + rb2[q].w);
      }
      if constexpr (DROP) b = drop4(b, b_key[q], (uint32_t)bn, p.drop_thresh, p.drop_scale);
      *reinterpret_cast<float4*>(Bs + (rr + 8 * q) * LDN + 4 * c4) = sel4(bok, b);
    }
  };

  f32x16 acc[2][2 * NGN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2 * NGN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int fcol = lane & 31;
  const int fk = lane >> 5;

  // Fragment reads are 8-byte: lane l takes columns (2l, 2l+1) of k-row fk, i.e. MFMA tile 0 of a wave owns the
  // even and tile 1 the odd columns of its 64-wide strip (un-permuted in the epilogue).  The fragments of
  // k-pair kk+1 are read before the MFMAs of k-pair kk so the LDS latency sits under the matrix pipe.
  // k-pairs [KK0, KK1) of the slab
  auto compute = [&](int buf, auto kk0_c, auto kk1_c) {
    constexpr int KK0 = decltype(kk0_c)::value, KK1 = decltype(kk1_c)::value;
    const float* As = smem + buf * STAGE + fk * LDM + wm * 64 + 2 * fcol;
    const float* Bs = smem + buf * STAGE + BK * LDM + fk * LDN + wn * 64 * NGN + 2 * fcol;
    float2 a = *reinterpret_cast<const float2*>(As + 2 * KK0 * LDM);
    float2 b[NGN];
#pragma unroll
    for (int g = 0; g < NGN; ++g) b[g] = *reinterpret_cast<const float2*>(Bs + 2 * KK0 * LDN + g * 64);
#pragma unroll
    for (int kk = KK0; kk < KK1; ++kk) {
      float2 na = a, nb[NGN];
#pragma unroll
      for (int g = 0; g < NGN; ++g) nb[g] = b[g];
      if (kk + 1 < KK1) {
        na = *reinterpret_cast<const float2*>(As + (2 * kk + 2) * LDM);
#pragma unroll
        for (int g = 0; g < NGN; ++g) nb[g] = *reinterpret_cast<const float2*>(Bs + (2 * kk + 2) * LDN + g * 64);
      }
#pragma unroll
      for (int g = 0; g < NGN; ++g) {
        acc[0][2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[g].x, acc[0][2 * g], 0, 0, 0);
        acc[0][2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[g].y, acc[0][2 * g + 1], 0, 0, 0);
        acc[1][2 * g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[g].x, acc[1][2 * g], 0, 0, 0);
        acc[1][2 * g + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[g].y, acc[1][2 * g + 1], 0, 0, 0);
      }
      a = na;
#pragma unroll
      for (int g = 0; g < NGN; ++g) b[g] = nb[g];
    }
  };

  // Two-region pipeline (see gemm_engine.hpp): A(t+1) is staged under the first 8 k-pairs of slab t, B(t+1) under
  // the last 8; the loads of slab t+2 go out as soon as their registers are free (rows past r_end are clamped and
  // masked, so over-fetching one slab at the end is harmless).
  using std::integral_constant;
  // ADMA: wave w stages tile rows w, w + 8, w + 16, w + 24 of a slab; lane l the columns 4 l .. 4 l + 3
  const unsigned tn_lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)smem;
  auto issue_a_dma = [&](long k0_, int buf) {
    const int w = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const long r = k0_ + w + 8 * q;
      tn_glds16(p.A + r * p.lda + m0 + 4 * lane,
                __builtin_amdgcn_readfirstlane(tn_lds0 + (unsigned)(buf * STAGE + (w + 8 * q) * LDM) * 4u));
    }
  };
  if (r_begin < r_end) {
    if constexpr (ADMA) {
      issue_a_dma(r_begin, 0);
      fetch_b(r_begin);
      commit_b(0);
      fetch_b(r_begin + BK);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      fetch_a(r_begin);
      fetch_b(r_begin);
      commit_a(0);
      commit_b(0);
      fetch_a(r_begin + BK);
      fetch_b(r_begin + BK);
      __syncthreads();
    }
    int cur = 0;
    long k0 = r_begin;
    if constexpr (!ADMA) {
      for (; k0 + BK < r_end; k0 += BK) {
        __builtin_amdgcn_sched_barrier(0);
        compute(cur, integral_constant<int, 0>{}, integral_constant<int, BK / 4>{});
        commit_a(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch_a(k0 + 2 * BK);
        __builtin_amdgcn_sched_barrier(0);
        compute(cur, integral_constant<int, BK / 4>{}, integral_constant<int, BK / 2>{});
        commit_b(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch_b(k0 + 2 * BK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
      }
    } else {
      for (; k0 + BK < r_end; k0 += BK) {
        issue_a_dma(k0 + BK, cur ^ 1);  // the idle buffer was last read in the previous slab (barrier passed)
        __builtin_amdgcn_sched_barrier(0);
        compute(cur, integral_constant<int, 0>{}, integral_constant<int, BK / 4>{});
        __builtin_amdgcn_sched_barrier(0);
        // everything in flight is at least half a slab (~8k cycles) old: the B(t+1) registers and this wave's share of
        // the A(t+1) DMA.  The explicit wait (the DMA is invisible to hipcc) precedes the barrier that publishes it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        compute(cur, integral_constant<int, BK / 4>{}, integral_constant<int, BK / 2>{});
        commit_b(cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        fetch_b(k0 + 2 * BK);  // stays in flight across the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
      }
    }
    compute(cur, integral_constant<int, 0>{}, integral_constant<int, BK / 2>{});
  }

  float* out = p.Cpart + (long)split * p.M * p.ldc;
  const int hl = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int g = 0; g < NGN; ++g) {
      const int n = n0 + wn * 64 * NGN + g * 64 + 2 * fcol;  // columns n, n+1 <- tiles 2g, 2g+1
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * hl) + i;
        if (m < p.M && n < p.N)  // N is a multiple of 4 and n is even: n + 1 < N as well
          *reinterpret_cast<float2*>(out + (long)m * p.ldc + n) = make_float2(acc[i][2 * g][e], acc[i][2 * g + 1][e]);
      }
    }
  }
}

constexpr int TN_LDS_BYTES = 2 * 32 * (132 + 132) * (int)sizeof(float);
constexpr int TN_LDS_BYTES_BIG = 2 * 32 * (260 + 260) * (int)sizeof(float);

// dst[m][n] (ld = ldd) = sum_s part[s][m][n] (ld = ldp)
__global__ void k_splitk_reduce(const float* __restrict__ part, int nsplit, int M, int N, long ldp,
                                float* __restrict__ dst, long ldd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (long)m * N);
  float a = 0.f;
  for (int s = 0; s < nsplit; ++s) a += part[((long)s * M + m) * ldp + n];
  dst[(long)m * ldd + n] = a;
}

}  // namespace pn
