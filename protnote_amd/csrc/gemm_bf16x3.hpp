// "bf16x3" NT GEMM for gfx950: f32 operands are split on the fly into hi = bf16(x), lo = bf16(x - hi) and every
// product is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on v_mfma_f32_32x32x16_bf16 with f32 accumulation
// (the dropped a_lo*b_lo term and the split residuals are ~2^-17 relative, i.e. ~1e-5 per product instead of f32's
// 6e-8; 3 MFMAs on the 16x faster bf16 pipe = 5.3x the f32-MFMA rate).  Opt-in (pn_set_math_mode(1)): the default
// path stays the exact-f32 engine of gemm_engine.hpp.  Same operand generators / epilogues / tile order as there.
//
// Restrictions (the pair-grid GEMMs of the hot path meet them): one K segment, K % 32 == 0, N % BN == 0.
// LDS row = 32 k-values as 4 groups of [hi x 8 (16 B)][lo x 8 (16 B)] + 16 B pad = 144 B: thread kv of a row owns the
// k-values {4kv..4kv+3} and {16+4kv..16+4kv+3} of the slab (two float4 loads; the four threads of a row read one
// contiguous 64-byte half line per load instruction; which 8 k share an MFMA k-group is free as long as both operands
// agree), so both planes are written with ds_write_b128 and a fragment
// (row = lane % 32, k-group = 2*kstep + lane / 32) is one ds_read_b128 per plane - no register shuffling, and the
// 144 B stride keeps every 16-lane group on 16 distinct 16-byte slots for reads and writes alike.
#pragma once
#include "gemm_engine.hpp"
#include "gemm_tn.hpp"

#ifndef PN_B3_VALU
#define PN_B3_VALU 4
#endif

namespace pn {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HiLo {
  bf16x4 hi, lo;
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> packed bf16 hi pair and lo pair: 6 VALU ops (cvt_pk, shift, and, 2 subs, cvt_pk)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const f32x2 x = {x0, x1};
  const bf16x2 h = __builtin_convertvector(x, bf16x2);  // v_cvt_pk_bf16_f32, round to nearest even
  hi = __builtin_bit_cast(uint32_t, h);
  // (a packed subtract - v_pk_add_f32 with neg modifiers from inline asm - was measured here and is SLOWER, -6 % on every
  //  bf16x3 kernel: the sched_group_barrier weave cannot classify inline asm, and the pair constraint costs register moves)
  const float r0 = x0 - __uint_as_float(hi << 16);          // exact in f32
  const float r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  const f32x2 r = {r0, r1};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
}

__device__ __forceinline__ HiLo split4(float4 v) {
  uint32_t h0, l0, h1, l1;
  split2(v.x, v.y, h0, l0);
  split2(v.z, v.w, h1, l1);
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  HiLo r;
  r.hi = __builtin_bit_cast(bf16x4, (u32x2){h0, h1});
  r.lo = __builtin_bit_cast(bf16x4, (u32x2){l0, l1});
  return r;
}

// eight f32 -> eight bf16 (round to nearest even): 4 x v_cvt_pk_bf16_f32
__device__ __forceinline__ bf16x8 round8(float4 v0, float4 v1) {
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const f32x8 x = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  return __builtin_convertvector(x, bf16x8);
}

// GEN = false: the pair-grid fast path (one K segment, K % 32 == 0, N % BN == 0: no masks anywhere).
// GEN = true : segmented K with a ragged tail, ragged N, the dilated-conv tap gather (A_CONV) - the encoder's convs
//              and the row MLPs; invalid operand quads are zeroed by selects, loads stay unconditional.
// BDMA (pair-grid fast path only): the weight operand arrives PRE-SPLIT as two bf16 planes [N][K] (p.w_hi / p.w_lo,
//              written once per launch by k_split_planes) and goes global -> LDS by LDS-DMA (global_load_lds_dwordx4,
//              16 tile rows of one plane per wave-instruction): no VGPR round trip, no split, no ds_write for half of
//              the staged data.  Its LDS image is [plane][row][4 granules of 8 k] (64 B rows, unpadded - a DMA
//              destination is contiguous) with granule g of row r at position g ^ ((r >> 2) & 3): conflict-free for the
//              16-lane groups of ds_read_b128.  W is L2-resident (3 MB per column tile), so one region of lead time is
//              enough: B(s+1) is issued at the top of slab s and waited for (vmcnt(0), the only VMEM in flight at that
//              point) between the A commit and the A prefetch; the barrier at the end of the slab publishes it.
// NP = 3: the three split products above.  NP = 1 ("bf16 backward", pn_set_backward_math(1)): ONE product of the bf16-rounded
//              operands, f32 accumulation - the arithmetic class of the reference's autocast backward (ProtNoteTrainer.py:728-738:
//              the Linear layers' gradient GEMMs run in half precision).  Same LDS image (the lo slots stay unwritten and
//              unread), a third of the MFMAs, half of the fragment reads and LDS writes, only the hi plane is DMA-staged.
template <int AK, int EK, int WAVES_M, int WAVES_N, int WM, int WN, bool GEN = false, bool BDMA = false, int NP = 3>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, (WAVES_M * WAVES_N) / 4) void gemm_nt_bf16x3_kernel(const GemmParams p) {
  static_assert(!BDMA || (!GEN && WAVES_M * WAVES_N == 8 && WAVES_N * WN * 32 == 256), "BDMA: 256-column pair-grid tiles");
  static_assert(NP == 3 || (NP == 1 && BDMA), "the single-product variant is built for the DMA-staged pair-grid kernel");
  constexpr int BK = 32;
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 32;
  constexpr int BN = WAVES_N * WN * 32;
  constexpr int LDK = BK + 4;   // row stride in 4-byte units (144 B)
  constexpr int KV = BK / 8;    // threads per tile row (8 k-values each)
  constexpr int RPP = NT / KV;  // tile rows covered per pass of the workgroup
  constexpr int NQA = BM / RPP;
  constexpr int NQB = (BN + RPP - 1) / RPP;  // BN = 192 (conv tile): the second pass covers rows 128..191 only
  static_assert(BM % RPP == 0, "tile/thread mismatch");
  static_assert(AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU || AK == A_CONV,
                "operand kind not built for bf16x3");
  static_assert(AK != A_CONV || GEN, "the conv gather needs the general path");
  constexpr int BPLANE = BN * 16;                              // floats per B plane of a stage (BN rows x 64 B)
  constexpr int STAGE = BDMA ? BM * LDK + 2 * BPLANE : (BM + BN) * LDK;

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

  // ---- BDMA: wave w issues chunks c = 4 w + q (q < 4) of a stage: plane c / 16, tile rows 16 (c % 16) .. + 15;
  //      lane l: row + l / 4, LDS granule position l % 4 <- source granule (l % 4) ^ ((row >> 2) & 3)
  //      (NP = 1: the 16 chunks of the hi plane only, c = 2 w + q, q < 2)
  constexpr int NDQ = NP == 3 ? 4 : 2;
  const uint16_t* bsrc[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned bdst[4] = {0, 0, 0, 0};
  if constexpr (BDMA) {
    const int w = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int q = 0; q < NDQ; ++q) {
      const int c = NDQ * w + q;
      const int r = 16 * (c & 15) + (lane >> 2);
      const int g = (lane & 3) ^ ((r >> 2) & 3);
      bsrc[q] = ((c >> 4) ? p.w_lo : p.w_hi) + (long)(col0 + r) * p.Kseg + 8 * g;
      bdst[q] = (unsigned)(BM * LDK + (c >> 4) * BPLANE + 16 * (c & 15) * 16) * 4u;
    }
  }
  const unsigned lds0_b = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)smem;
  auto issue_b_dma = [&](int s_, int buf) {
#pragma unroll
    for (int q = 0; q < NDQ; ++q)
      tn_glds16(reinterpret_cast<const float*>(bsrc[q] + s_ * BK),
                __builtin_amdgcn_readfirstlane(lds0_b + (unsigned)(buf * STAGE) * 4u + bdst[q]));
  };

  const int kv = tid % KV;
  const int r_in = tid / KV;

  const float* arow[NQA];
  const float* arow2[NQA];
  int a_t[NQA], a_len[NQA];
#pragma unroll
  for (int q = 0; q < NQA; ++q) {
    int r = row0 + r_in + q * RPP;
    if (r > p.M - 1) r = p.M - 1;  // clamp: duplicates are discarded by the epilogue
    if constexpr (AK == A_PAIRSUM_RELU) {
      const int j = r / p.pairB;
      const int i = r - j * p.pairB;
      arow[q] = p.A + (long)i * p.lda + 4 * kv;
      arow2[q] = p.A2 + (long)j * p.lda2 + 4 * kv;
    } else {
      arow[q] = p.A + (long)r * p.lda + 4 * kv;
      arow2[q] = nullptr;
    }
    a_t[q] = 0;
    a_len[q] = 0;
    if constexpr (AK == A_CONV) {
      const int b = r / p.L;
      a_t[q] = r - b * p.L;
      a_len[q] = p.lens[b];
    }
  }
  const float* brow[NQB];
  bool bvalid[NQB];
#pragma unroll
  for (int q = 0; q < NQB; ++q) {
    const int n = col0 + r_in + q * RPP;
    bvalid[q] = (r_in + q * RPP < BN) && (!GEN || n < p.N);
    brow[q] = p.W + (long)(bvalid[q] ? n : 0) * p.ldw + 4 * kv;
  }

  const int spt = GEN ? (p.Kseg + BK - 1) / BK : p.Kseg / BK;  // slabs per segment
  const int nslab = GEN ? p.nseg * spt : spt;
  const bool conv_affine = (AK == A_CONV) && (p.a_scale != nullptr);

  float4 ra[NQA][2], ra2[NQA][2], rb[NQB][2];
  float4 rsc[2], rsh[2];
  unsigned avalid = 0xffffffffu, bmask = 0xffffffffu;  // bit 2q+h: quad h of pass q is live (GEN only)

  auto fetch_a = [&](int s) {
    if constexpr (!GEN) {
      const int c = s * BK;
#pragma unroll
      for (int q = 0; q < NQA; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          ra[q][h] = ld4(arow[q] + c + 16 * h);
          if constexpr (AK == A_PAIRSUM_RELU) ra2[q][h] = ld4(arow2[q] + c + 16 * h);
        }
      if constexpr (AK == A_AFFINE_RELU) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          rsc[h] = ld4(p.a_scale + c + 4 * kv + 16 * h);
          rsh[h] = ld4(p.a_shift + c + 4 * kv + 16 * h);
        }
      }
    } else {
      const int seg = s / spt;
      const int c0 = (s - seg * spt) * BK;  // column of this slab within the segment (thread adds 4 kv + 16 h)
      bool kok[2];
      int cc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        kok[h] = c0 + 4 * kv + 16 * h < p.Kseg;
        cc[h] = kok[h] ? c0 + 16 * h : -4 * kv;  // clamped to the row start (arow already holds + 4 kv)
      }
      avalid = 0;
      if constexpr (AK == A_CONV) {
        const int sh = (seg - p.nseg / 2) * p.dil;
#pragma unroll
        for (int q = 0; q < NQA; ++q) {
          const int tt = a_t[q] + sh;
          const bool ok = (a_t[q] < a_len[q]) && (tt >= 0) && (tt < a_len[q]);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            ra[q][h] = ld4(arow[q] + (long)(ok ? sh : 0) * p.lda + cc[h]);
            avalid |= ((ok && kok[h]) ? 1u : 0u) << (2 * q + h);
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < NQA; ++q)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            ra[q][h] = ld4(arow[q] + cc[h]);
            if constexpr (AK == A_PAIRSUM_RELU) ra2[q][h] = ld4(arow2[q] + cc[h]);
            avalid |= (kok[h] ? 1u : 0u) << (2 * q + h);
          }
      }
      if (AK == A_AFFINE_RELU || conv_affine) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          rsc[h] = ld4(p.a_scale + 4 * kv + cc[h]);
          rsh[h] = ld4(p.a_shift + 4 * kv + cc[h]);
        }
      }
    }
  };
  auto fetch_b = [&](int s) {
    if constexpr (!GEN) {
      const int c = s * BK;
#pragma unroll
      for (int q = 0; q < NQB; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) rb[q][h] = ld4(brow[q] + c + 16 * h);
    } else {
      const int seg = s / spt;
      const int c0 = (s - seg * spt) * BK;
      bmask = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool kok = c0 + 4 * kv + 16 * h < p.Kseg;
        const int cc = kok ? c0 + 16 * h : -4 * kv;
#pragma unroll
        for (int q = 0; q < NQB; ++q) {
          rb[q][h] = ld4(brow[q] + (long)seg * p.Kseg + cc);
          bmask |= ((kok && bvalid[q]) ? 1u : 0u) << (2 * q + h);
        }
      }
    }
  };
  auto pin_a = [&]() {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pin4(ra[q][h]);
        if constexpr (AK == A_PAIRSUM_RELU) pin4(ra2[q][h]);
      }
  };
  auto pin_b = [&]() {
#pragma unroll
    for (int q = 0; q < NQB; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) pin4(rb[q][h]);
  };

  auto store8 = [](float* dst, float4 v0, float4 v1) {
    if constexpr (NP == 1) {
      *reinterpret_cast<bf16x8*>(dst) = round8(v0, v1);
      return;
    }
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

  auto sel4 = [](bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };
  auto commit_a = [&](int buf) {
    float* As = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      float4 v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v[h] = ra[q][h];
        if (AK == A_AFFINE_RELU || (AK == A_CONV && conv_affine)) {
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
        if constexpr (GEN) v[h] = sel4((avalid >> (2 * q + h)) & 1u, v[h]);
      }
      store8(As + (r_in + q * RPP) * LDK + 8 * kv, v[0], v[1]);
    }
  };
  auto commit_b = [&](int buf) {
    float* Bs = smem + buf * STAGE + BM * LDK;
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      float4 v0 = rb[q][0], v1 = rb[q][1];
      if constexpr (GEN) {
        v0 = sel4((bmask >> (2 * q)) & 1u, v0);
        v1 = sel4((bmask >> (2 * q + 1)) & 1u, v1);
      }
      if (BN % RPP == 0 || r_in + q * RPP < BN) store8(Bs + (r_in + q * RPP) * LDK + 8 * kv, v0, v1);
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
  const int frag_g = lane >> 5;  // which 8-k group of the 16-k step

  auto compute = [&](int buf, auto ks_c) {
    constexpr int KS = decltype(ks_c)::value;
    const float* As = smem + buf * STAGE + (wm * WM * 32 + frag_row) * LDK + (2 * KS + frag_g) * 8;
    bf16x8 ah[WM], al[WM], bh[WN], bl[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK);
      if constexpr (NP == 3) al[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK + 4);
    }
    if constexpr (BDMA) {
      // row r of plane P: P + 16 r floats; k-group g at granule position g ^ ((r >> 2) & 3), (r >> 2) & 3 is the
      // lane's own (frag_row >> 2) & 3 for every tile of the wave
      const float* Bs = smem + buf * STAGE + BM * LDK + (wn * WN * 32 + frag_row) * 16 +
                        4 * ((2 * KS + frag_g) ^ ((frag_row >> 2) & 3));
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * 16);
        if constexpr (NP == 3) bl[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * 16 + BPLANE);
      }
    } else {
      const float* Bs = smem + buf * STAGE + BM * LDK + (wn * WN * 32 + frag_row) * LDK + (2 * KS + frag_g) * 8;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK);
        bl[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        if constexpr (NP == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };

  // Main loop.  A 32x32x16 bf16 MFMA occupies the matrix pipe for 32 cycles, in which the SIMD can issue ~5 other
  // instructions of the same wave; PMC on the first version (MFMA block, then staging block): matrix pipe busy 57 %
  // + VALU issue 37 % of the time = ~never both.  So the staging work is woven between the MFMAs at instruction level
  // (sched_group_barrier): per slab s
  //   region 1: wait A(s+1) | 24 MFMAs of k-step 0  x  generator + bf16 split + LDS writes of A(s+1) | issue A(s+2)
  //   region 2: wait B(s+1) | 24 MFMAs of k-step 1  x  split + LDS writes of B(s+1)               | issue B(s+2)
  // every global load has a full slab to land.
  using std::integral_constant;
  auto weave = [&](auto nvalu_c) {
    constexpr int NV = decltype(nvalu_c)::value;
    if constexpr (NP == 1) {  // 8 MFMAs per k-step; the region stages 2 x 8 elements per thread: ~3 VALU per MFMA, 2 LDS writes
      __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
#pragma unroll
      for (int i = 0; i < WM * WN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
        if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      return;
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (WM + WN), 0);  // all fragment reads of the k-step first
#pragma unroll
    for (int i = 0; i < WM * WN * 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // VALU of the staging path
      if (i % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // LDS writes: 8 per region
    }
  };
  if constexpr (BDMA) {
    issue_b_dma(0, 0);
    fetch_a(0);
    pin_a();
    commit_a(0);
    fetch_a(nslab > 1 ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s + 1 < nslab; ++s) {
      const int cur = s & 1;
      const int nxt = s + 2 < nslab ? s + 2 : s + 1;
      __builtin_amdgcn_sched_barrier(0);
      pin_a();  // (first: hipcc's counted waits for the A registers must not see the DMA as younger traffic)
      issue_b_dma(s + 1, cur ^ 1);  // the idle buffer was last read in slab s-1 (barrier passed)
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, integral_constant<int, 0>{});
      commit_a(cur ^ 1);
      weave(integral_constant<int, (AK == A_PLAIN ? PN_B3_VALU : PN_B3_VALU + 1)>{});
      __builtin_amdgcn_sched_barrier(0);
      // only the DMA of B(s+1) is in flight here (the A registers were consumed above): wait for this wave's share,
      // then prefetch A(s+2) so that it stays in flight across the barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      fetch_a(nxt);
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    compute((nslab - 1) & 1, integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    compute((nslab - 1) & 1, integral_constant<int, 1>{});
    __syncthreads();
  } else {
  fetch_a(0);
  fetch_b(0);
  pin_a();
  pin_b();
  commit_a(0);
  commit_b(0);
  fetch_a(nslab > 1 ? 1 : 0);
  fetch_b(nslab > 1 ? 1 : 0);
  __syncthreads();
  for (int s = 0; s + 1 < nslab; ++s) {
    const int cur = s & 1;
    const int nxt = s + 2 < nslab ? s + 2 : s + 1;  // (the last one is a harmless re-read)
    __builtin_amdgcn_sched_barrier(0);
    pin_a();
    compute(cur, integral_constant<int, 0>{});
    commit_a(cur ^ 1);
    weave(integral_constant<int, (AK == A_PLAIN ? PN_B3_VALU : PN_B3_VALU + 1)>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch_a(nxt);
    __builtin_amdgcn_sched_barrier(0);
    pin_b();
    compute(cur, integral_constant<int, 1>{});
    commit_b(cur ^ 1);
    weave(integral_constant<int, PN_B3_VALU>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch_b(nxt);
    __syncthreads();
  }
  compute((nslab - 1) & 1, integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  compute((nslab - 1) & 1, integral_constant<int, 1>{});
  __syncthreads();
  }

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
// NP = 1: one product of the bf16-rounded operands (see the NT kernel).
template <int TB, int NP = 3>
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
  int split = blockIdx.y;
  if (p.task_ns > 0) {  // 32-workgroup region tasks, see tn_task_coords (gemm_tn.hpp)
    if (!tn_task_coords(p.task_ns, tile_m, tile_n, split)) return;
  } else if (PN_XCD && TB == TB_PAIRSUM_RELU && (ntm % 4 == 0) && (ntn % 2 == 0)) {
    // the B operand comes from two small tables: only dz streams, so give each XCD as few dz column panels as
    // possible (4 x 2 regions: 3 of the 12 panels instead of 6)
    const int rm = ntm / 4, rn = ntn / 2;
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    tile_m = (xcd >> 1) * rm + w / rn;
    tile_n = (xcd & 1) * rn + w % rn;
  } else if (PN_XCD && (ntm % 2 == 0) && (ntn % 4 == 0)) {  // same XCD regions as gemm_tn_kernel
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

  const int col = tid & 255;  // the column of both tiles this thread stages
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 8);  // it stages k-groups kg and kg + 2 (wave-uniform)
  float bs = 0.f, bt = 0.f;
  if constexpr (TB == TB_AFFINE_RELU) {
    bs = p.b_s[n0 + col];
    bt = p.b_t[n0 + col];
  }

  float ra[2][8], rb[2][8], rb2[2];
  const unsigned pB = (unsigned)p.pairB;
  const long lda = p.lda, ldb = p.ldb;

  // SAFE = false: slab known to lie inside [r_begin, r_end).  SAFE = true: rows >= r_end are read from the last valid
  // row instead (and zeroed by commit).
  // Addresses = wave-uniform row base (scalar registers) + the thread's column as a 32-bit offset.
  auto fetch_a = [&](long k0, auto safe_c) {
    constexpr bool SAFE = decltype(safe_c)::value;
    const float* Ak = p.A + k0 * lda + m0;
    const int lim = (int)(r_end - 1 - k0);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int ro = 8 * (kg + 2 * g) + e;
        if (SAFE) ro = ro < lim ? ro : lim;
        const float* rowp = Ak + (long)ro * lda;  // uniform
        ra[g][e] = rowp[(unsigned)col];
      }
  };
  auto fetch_b = [&](long k0, auto safe_c) {
    constexpr bool SAFE = decltype(safe_c)::value;
    const int lim = (int)(r_end - 1 - k0);
    if constexpr (TB == TB_PAIRSUM_RELU) {
      // rows 8 * (kg + 2g) .. + 7 of the slab: pairB % 8 == 0 and k0 % 8 == 0, so the 8 rows share j = r / pairB
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int rr0 = 8 * (kg + 2 * g);
        if (SAFE) rr0 = rr0 < lim ? rr0 : (lim & ~7);
        const unsigned ru = (unsigned)(k0 + rr0);
        const unsigned j = ru / pB;
        const unsigned i = ru - j * pB;
        const float* Bi = p.B + (long)i * ldb + n0;  // uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          int eo = e;
          if (SAFE) eo = (rr0 + e) < lim ? e : (lim - rr0 > 0 ? lim - rr0 : 0);
          const float* rowp = Bi + (long)eo * ldb;
          rb[g][e] = rowp[(unsigned)col];
        }
        const float* B2j = p.B2 + (long)j * p.ldb2 + n0;
        rb2[g] = B2j[(unsigned)col];
      }
    } else {
      const float* Bk = p.B + k0 * ldb + n0;
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          int ro = 8 * (kg + 2 * g) + e;
          if (SAFE) ro = ro < lim ? ro : lim;
          const float* rowp = Bk + (long)ro * ldb;
          rb[g][e] = rowp[(unsigned)col];
        }
    }
  };

  auto pin_a = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(ra[g][e]));
  };
  auto pin_b = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(rb[g][e]));
      if constexpr (TB == TB_PAIRSUM_RELU) asm volatile("" : "+v"(rb2[g]));
    }
  };

  auto store8 = [](float* dst, const float (&v)[8]) {
    if constexpr (NP == 1) {
      *reinterpret_cast<bf16x8*>(dst) = round8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
      return;
    }
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

  auto commit_a = [&](int buf, long k0, auto safe_c) {
    constexpr bool SAFE = decltype(safe_c)::value;
    float* As = smem + buf * STAGE;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] = ra[g][e];
        if (SAFE && k0 + 8 * (kg + 2 * g) + e >= r_end) a[e] = 0.f;
      }
      store8(As + col * LDK + 8 * (kg + 2 * g), a);
    }
  };
  auto commit_b = [&](int buf, long k0, auto safe_c) {
    constexpr bool SAFE = decltype(safe_c)::value;
    float* Bs = smem + buf * STAGE + BM * LDK;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        b[e] = rb[g][e];
        if constexpr (TB == TB_AFFINE_RELU) b[e] = relu(fmaf(b[e], bs, bt));
        if constexpr (TB == TB_PAIRSUM_RELU) b[e] = relu(b[e] + rb2[g]);
        if (SAFE && k0 + 8 * (kg + 2 * g) + e >= r_end) b[e] = 0.f;
      }
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
      if constexpr (NP == 3) al[i] = *reinterpret_cast<const bf16x8*>(As + i * 32 * LDK + 4);
    }
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      bh[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK);
      if constexpr (NP == 3) bl[j] = *reinterpret_cast<const bf16x8*>(Bs + j * 32 * LDK + 4);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        if constexpr (NP == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  };

  using std::integral_constant;
  using std::false_type;
  using std::true_type;
  auto weave = [&](auto nvalu_c) {
    constexpr int NV = decltype(nvalu_c)::value;
    if constexpr (NP == 1) {  // 8 MFMAs per k-step against the staging of 16 elements: 3 x the vector work per MFMA, 2 LDS writes
      __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
#pragma unroll
      for (int i = 0; i < WM * WN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3 * NV, 0);
        if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
      return;
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (WM + WN), 0);
#pragma unroll
    for (int i = 0; i < WM * WN * 3; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      if (i % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
  };
  // slab t covers rows [r_begin + 32 t, +32); slabs 0 .. nf-1 are full, slab nf (if any) is the ragged tail.
  // Same two-region pipeline as the NT kernel: region 1 = k-step 0 x staging of A(t+1), then issue A(t+2);
  // region 2 = k-step 1 x staging of B(t+1), then issue B(t+2).
  if (r_begin < r_end) {
    const long span = r_end - r_begin;
    const int nf = (int)(span / BK);
    const int ns = nf + ((span % BK) ? 1 : 0);
    auto K0 = [&](int t) { return r_begin + (long)t * BK; };
    auto body = [&](int t, auto safe_c) {  // computes slab t, stages slab t+1, issues the loads of slab t+2
      const int cur = t & 1;
      const int t2 = t + 2 < ns ? t + 2 : t + 1;
      __builtin_amdgcn_sched_barrier(0);
      pin_a();
      compute(cur, integral_constant<int, 0>{});
      commit_a(cur ^ 1, K0(t + 1), safe_c);
      weave(integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      fetch_a(K0(t2), safe_c);
      __builtin_amdgcn_sched_barrier(0);
      pin_b();
      compute(cur, integral_constant<int, 1>{});
      commit_b(cur ^ 1, K0(t + 1), safe_c);
      weave(integral_constant<int, (TB == TB_PLAIN ? 2 : 3)>{});
      __builtin_amdgcn_sched_barrier(0);
      fetch_b(K0(t2), safe_c);
      __syncthreads();
    };
    fetch_a(K0(0), true_type{});
    fetch_b(K0(0), true_type{});
    pin_a();
    pin_b();
    commit_a(0, K0(0), true_type{});
    commit_b(0, K0(0), true_type{});
    fetch_a(K0(ns > 1 ? 1 : 0), true_type{});
    fetch_b(K0(ns > 1 ? 1 : 0), true_type{});
    __syncthreads();
    int t = 0;
    for (; t + 2 < nf; ++t) body(t, false_type{});  // slabs t+1 and t+2 are full ones
    for (; t + 1 < ns; ++t) body(t, true_type{});
    compute((ns - 1) & 1, integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    compute((ns - 1) & 1, integral_constant<int, 1>{});
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

// W [N][ldw] f32 -> hi / lo bf16 planes [N][K] (row stride K), the pre-split weight operand of the BDMA kernels.
// Inside every 32-k block the planes are stored in MFMA k-group order: the register-staged A operand puts
// k = {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3} into k-group g (see the file header), so plane position 8 g + e holds
// k = 4 g + e (e < 4) or 16 + 4 g + e - 4 (e >= 4) - both operands must agree on which 8 k meet in one MFMA lane.
__global__ void k_split_planes(const float* __restrict__ W, long ldw, int N, int K, uint16_t* __restrict__ hi,
                               uint16_t* __restrict__ lo) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;  // output position (4 consecutive plane slots)
  if (i >= (long)N * K) return;
  const long n = i / K;
  const int pos = (int)(i - n * K);
  const int blk = pos >> 5, in = pos & 31, g = in >> 3, half = (in >> 2) & 1;
  const HiLo s = split4(ld4(W + n * ldw + blk * 32 + half * 16 + 4 * g));
  *reinterpret_cast<bf16x4*>(hi + i) = s.hi;
  *reinterpret_cast<bf16x4*>(lo + i) = s.lo;
}

}  // namespace pn
