// f32-MFMA "NT" GEMM engine for gfx950 (CDNA4):  C[M,N] = opA(A)[M,K] * W[N,K]^T  with fused operand
// generators and epilogues.  One workgroup = WAVES_M x WAVES_N wave64s, each owning WM x WN tiles of
// v_mfma_f32_32x32x2_f32 (exact f32: bitwise a k-ordered fmaf chain).  Operand tiles are staged
// global -> registers (transform applied here: BN affine / ReLU / padding mask / pair broadcast-sum)
// -> LDS [row][BK+4] -> ds_read_b128 fragments.  The LDS row stride BK+4 floats keeps every 16-lane
// ds_read_b128 group on 16 distinct 16-byte slots (conflict-free) and every ds_write_b128 aligned.
//
// K is "segmented": K = nseg segments of Kseg floats (Kseg % 4 == 0).  A plain GEMM has nseg = 1.
// The dilated conv (reference protein_encoders.py:8-17,39-46) is the implicit GEMM with one segment per
// tap: segment s reads activation row p + (s - nseg/2)*dil, zero outside [0, len[b]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifndef PN_MINW
#define PN_MINW 2
#endif
#ifndef PN_BK
#define PN_BK 32
#endif
#ifndef PN_XCD
#define PN_XCD 1
#endif

namespace pn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// A_DZ_*: BatchNorm+ReLU backward fused into the operand load.  With u = s*z + t the forward activation
// was relu(u); given the upstream gradient g of that activation,
//     dz = (u > 0 ? g * cs[k] : 0) + p[k] + q[k] * z
// where cs/p/q are per-column vectors prepared by k_bn_bwd_finalize (cs = s, or s*w_out for the last
// hidden layer whose upstream gradient is the rank-1 dlogit[r] * w_out[k]).
//   A_DZ_ELEM: g = G[r][k] (stored gradient matrix);  A_DZ_ROWG: g = gvec[r] (one scalar per row).
//   A_PAIRPROD: A[r % pairB][k] * A2[r / pairB][k]  (the P (.) L block of concatenation_prod, ProtNote.py:139-150)
enum { A_PLAIN = 0, A_AFFINE_RELU = 1, A_PAIRSUM_RELU = 2, A_CONV = 3, A_DZ_ELEM = 4, A_DZ_ROWG = 5, A_PAIRPROD = 6 };
// E_PAIRADD: E_STORE plus the separable part of the first pair layer, out = acc + X1[r % pairB][n] + X2[r / pairB][n]
// E_STORE_H16: relu(acc * e_scale[n] + e_shift[n]) stored as bf16 (C is a uint16 matrix, ldc in bf16 elements): the eval-mode
//              producer of the AMP-class forward writes the next layer's operand in the form its all-DMA consumer stages
enum { E_STORE = 0, E_CONV = 1, E_ROWDOT = 2, E_SCALE_RC = 3, E_PAIRADD = 4, E_STORE_H16 = 5 };

struct GemmParams {
  int M, N;          // output rows / true output columns
  int Nstore;        // columns written (>= N; columns in [N, Nstore) are written as 0 - padded layouts)
  int nseg, Kseg;    // segmented K
  // ---- A operand ----
  const float* A;
  long lda;
  const float* a_scale;  // per-k affine (A_AFFINE_RELU, optional for A_CONV): relu(x*scale+shift)
  const float* a_shift;
  const float* A2;       // A_PAIRSUM_RELU: relu(A[r % pairB] + A2[r / pairB]);  A_DZ_ELEM: G (ld = lda2)
  long lda2;
  int pairB;
  const int* lens;       // A_CONV / E_CONV: int32 sequence lengths [M / L]
  int L;
  int dil;
  // ---- B operand (weights, [N][ldw], K-contiguous) ----
  const float* W;
  long ldw;
  // ---- epilogue ----
  float* C;
  long ldc;
  const float* bias;     // [N] or null
  const float* resid;    // E_CONV: residual added on valid rows, or null
  long ldr;
  double* col_sum;       // optional per-column sum / sum of squares of the stored values (train-mode BN): written
  double* col_sumsq;     // (not accumulated) by the launcher from the per-row-tile partials below, in a fixed order
  float* col_part;       // [row tiles][2][N] f32 partials, one slot per workgroup - no atomics, bit-reproducible
  double* col_red;       // [64][2][N] scratch of the two-level reduction
  const float* e_scale;  // E_ROWDOT: sum_n relu(acc*e_scale[n]+e_shift[n]) * e_w[n];  E_STORE (eval, optional): the
                         // stored value is relu(acc*e_scale[n]+e_shift[n]) - the next layer's BN+ReLU applied by the
                         // producer, so that the next GEMM takes a plain (LDS-DMA staged) operand
  const float* e_shift;
  const float* e_w;
  float* rowdot_out;     // [n_col_tiles * WAVES_N][M] deterministic partials
  const float* row_scale;  // E_SCALE_RC: acc * row_scale[m] * col_scale[n] * alpha
  const float* col_scale;
  float alpha;
  // ---- A_DZ_* (A = z matrix, a_scale/a_shift = s,t of the forward BN fold) ----
  const float* gvec;   // A_DZ_ROWG: per-row upstream scalar
  const float* dz_cs;  // per-k vectors
  const float* dz_p;
  const float* dz_q;
  const float* padd1;  // E_PAIRADD
  long ldp1;
  const float* padd2;
  long ldp2;
  // ---- inter-layer dropout of the A operand (training with OUTPUT_MLP_DROPOUT > 0; DROP kernels only):
  //      A[r][k] *= keep(drop_seed, r, k) ? drop_scale : 0, a counter-based hash of (seed, row, column) - the
  //      backward regenerates the same mask from the same seed (nothing is stored)
  uint32_t drop_seed, drop_thresh;  // thresh = round(p * 2^32); 0 = no dropout
  float drop_scale;                 // 1 / (1 - p)
  uint16_t* wsplit;      // optional scratch, 2 * N * Kseg bf16: bf16x3 mode pre-splits W into hi / lo planes there and
  const uint16_t* w_hi;  // stages them by LDS-DMA (set by the launcher from wsplit)
  const uint16_t* w_lo;
  int xcd_br, xcd_bc;  // > 0: XCD-aware order in br x bc tile blocks (set by the launcher when the grid suits it)
  const int* run_if;   // optional device flag: the whole launch is a no-op while *run_if == 0 (the general convolution
                       // behind the one-hot gather kernel of conv1 runs only when the input turned out not to be one-hot)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.f); }

// ---- dropout masks: a pure function of (seed, row, column), so forward, backward and the tests (pn_dropout_mask)
// all see the same Bernoulli(1 - p) draw.  lowbias32 (a well-mixed 32-bit integer hash) of the row with the seed
// gives a per-row key; hashing key + column * golden-ratio gives the element's uniform 32-bit draw.
__host__ __device__ __forceinline__ uint32_t pn_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_rowkey(uint32_t seed, uint32_t row) {
  return pn_lowbias32(row ^ seed) + seed;
}
__host__ __device__ __forceinline__ bool drop_keep(uint32_t rowkey, uint32_t col, uint32_t thresh) {
  return pn_lowbias32(rowkey + col * 0x9E3779B1U) >= thresh;
}
__device__ __forceinline__ float4 drop4(float4 v, uint32_t rowkey, uint32_t col, uint32_t thresh, float scale) {
  v.x = drop_keep(rowkey, col, thresh) ? v.x * scale : 0.f;
  v.y = drop_keep(rowkey, col + 1, thresh) ? v.y * scale : 0.f;
  v.z = drop_keep(rowkey, col + 2, thresh) ? v.z * scale : 0.f;
  v.w = drop_keep(rowkey, col + 3, thresh) ? v.w * scale : 0.f;
  return v;
}

// Make a fetched register quad opaque at this point of the program: the transform math that consumes it
// cannot be hoisted above (DAG linearisation otherwise floats it in front of the MFMA block, dragging the
// s_waitcnt vmcnt with it and exposing the global-load latency).
__device__ __forceinline__ void pin4(float4& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// Output-tile coordinates of this workgroup; false = padding workgroup of the XCD-ordered grid (returns at once).
// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only), and each XCD
// has its own 4 MB L2.  Give every XCD whole br x bc blocks of output tiles, one block (= the workgroups resident
// on its 32 CUs) at a time: inside a block each A row-panel is shared by bc and each W column-panel by br
// workgroups through that L2, and a row-panel is needed by ntn/bc XCDs instead of all 8.
template <int BM, int BN>
__device__ __forceinline__ bool tile_coords(const GemmParams& p, int& tile_m, int& tile_n) {
  if (p.run_if != nullptr && *p.run_if == 0) return false;
  const int bid = blockIdx.x;
  const int ntn = (p.Nstore + BN - 1) / BN;
  if (PN_XCD && p.xcd_bc > 0) {
    const int ntm = (p.M + BM - 1) / BM;
    const int br = p.xcd_br, bc = p.xcd_bc, bsz = br * bc;
    const int nbn = ntn / bc;
    const int nbm = (ntm + br - 1) / br;
    const int xw = bid >> 3;
    const int gb = (xw / bsz) * 8 + (bid & 7);
    if (gb >= nbm * nbn) return false;
    const int r = xw % bsz;
    tile_m = (gb / nbn) * br + r / bc;
    tile_n = (gb % nbn) * bc + r % bc;
    return tile_m < ntm && tile_n < ntn;
  }
  tile_n = bid % ntn;
  tile_m = bid / ntn;
  return true;
}

// Epilogue shared by the f32 and the bf16x3 kernels: acc[i][j] is the 32x32 MFMA accumulator of wave tile (i, j)
// (lane l: column l % 32, rows (e & 3) + 8 * (e >> 2) + 4 * (l / 32)).  LDS must be free (barrier passed).
template <int EK, int WAVES_M, int WAVES_N, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[WM][WN], int row0, int col0, int tile_n,
                                              float* smem) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BN = WAVES_N * WN * 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int hl = lane >> 5;  // which 4-row group of each 8
  const int cl = lane & 31;
  const bool want_stats = (EK == E_STORE || EK == E_CONV || EK == E_PAIRADD) && (p.col_part != nullptr);
  // [WAVES_M][2][BN] column partials, one slot per (wave row, column): plain stores, summed in wave-row order below
  float* red = smem;  // (LDS is free after the final barrier of the main loop)

  // E_CONV: which of this lane's rows are real residues (row < M and position < len of its sequence) - decided once
  // per row here, not per element inside the column loop (a lens[] load and an integer division behind a divergent
  // branch per element made the epilogue a chain of ~100 dependent global loads)
  unsigned live_mask[WM];
  if constexpr (EK == E_CONV) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      live_mask[i] = 0u;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
        const int rc = row < p.M ? row : p.M - 1;
        const int b = rc / p.L;
        const bool live = (row < p.M) && (rc - b * p.L) < p.lens[b];
        live_mask[i] |= (live ? 1u : 0u) << e;
      }
    }
  }

#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int col = col0 + (wn * WN + j) * 32 + cl;
    const bool cok = col < p.N;
    float s1 = 0.f, s2 = 0.f;
    float bj = 0.f, es = 0.f, et = 0.f, ew = 0.f, cs = 0.f;
    if constexpr (EK == E_STORE || EK == E_CONV || EK == E_PAIRADD) {
      bj = (cok && p.bias) ? p.bias[col] : 0.f;
    }
    if constexpr (EK == E_ROWDOT) {
      if (cok) {
        es = p.e_scale[col];
        et = p.e_shift[col];
        ew = p.e_w[col];
      }
    }
    if constexpr (EK == E_STORE_H16) {
      if (cok) {
        es = p.e_scale[col];
        et = p.e_shift[col];
      }
    }
    const bool store_act = (EK == E_STORE) && (p.e_scale != nullptr);
    if constexpr (EK == E_STORE) {
      if (store_act && cok) {
        es = p.e_scale[col];
        et = p.e_shift[col];
      }
    }
    if constexpr (EK == E_SCALE_RC) cs = cok ? p.col_scale[col] * p.alpha : 0.f;
    if constexpr (EK == E_CONV) {
      // residual: every load of this column strip issued back to back from clamped (always valid) addresses
      const int cc = cok ? col : 0;
      const bool has_res = p.resid != nullptr;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        float res[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
          const int rc = row < p.M ? row : p.M - 1;
          res[e] = has_res ? p.resid[(long)rc * p.ldr + cc] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
          const bool live = (live_mask[i] >> e) & 1u;
          float v = acc[i][j][e];
          v = (live && cok) ? (v + bj) + res[e] : 0.f;
          if (row < p.M && col < p.Nstore) p.C[(long)row * p.ldc + col] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if constexpr (EK != E_CONV) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
        const bool rok = row < p.M;
        float v = acc[i][j][e];
        if constexpr (EK == E_PAIRADD) {
          if (rok && cok) {
            const int pj = row / p.pairB;
            const int pi = row - pj * p.pairB;
            v += bj + p.padd1[(long)pi * p.ldp1 + col] + p.padd2[(long)pj * p.ldp2 + col];
            p.C[(long)row * p.ldc + col] = v;
            s1 += v;
            s2 += v * v;
          }
        } else if constexpr (EK == E_STORE) {
          v = cok ? v + bj : 0.f;
          if (store_act) v = cok ? relu(fmaf(v, es, et)) : 0.f;
          if (rok && col < p.Nstore) p.C[(long)row * p.ldc + col] = v;
          if (rok) {
            s1 += v;
            s2 += v * v;
          }
        } else if constexpr (EK == E_STORE_H16) {
          if (rok && cok) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
            const bf16x2_ hv = __builtin_convertvector(f32x2_{relu(fmaf(v, es, et)), 0.f}, bf16x2_);  // v_cvt_pk_bf16_f32: RNE
            reinterpret_cast<__bf16*>(p.C)[(long)row * p.ldc + col] = hv[0];
          }
        } else if constexpr (EK == E_SCALE_RC) {
          if (rok && cok) p.C[(long)row * p.ldc + col] = v * p.row_scale[row] * cs;
        } else if constexpr (EK == E_ROWDOT) {
          acc[i][j][e] = cok ? relu(fmaf(v, es, et)) * ew : 0.f;
        }
      }
    }
    }
    if (want_stats) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (hl == 0) {
        red[(wm * 2 + 0) * BN + (wn * WN + j) * 32 + cl] = s1;
        red[(wm * 2 + 1) * BN + (wn * WN + j) * 32 + cl] = s2;
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    const long tile_m = row0 / (WAVES_M * WM * 32);
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, c = i - which * BN;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) a += red[(w * 2 + which) * BN + c];
      const int col = col0 + c;
      if (col < p.N) p.col_part[(tile_m * 2 + which) * p.N + col] = a;
    }
  }
  if constexpr (EK == E_ROWDOT) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) v += acc[i][j][e];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        const int row = row0 + (wm * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
        if (cl == 0 && row < p.M) p.rowdot_out[(long)(tile_n * WAVES_N + wn) * p.M + row] = v;
      }
    }
  }
}

template <int AK, int EK, int WAVES_M, int WAVES_N, int WM, int WN, int BK, bool DROP = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, PN_MINW) void gemm_nt_kernel(const GemmParams p) {
  static_assert(!DROP || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU, "dropout applies to the hidden activations");
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int BM = WAVES_M * WM * 32;
  constexpr int BN = WAVES_N * WN * 32;
  constexpr int LDK = BK + 4;
  constexpr int KV = BK / 4;    // float4 granules per tile row
  constexpr int RPP = NT / KV;  // tile rows covered per pass of the workgroup
  constexpr int NQA = BM / RPP;
  constexpr int NQB = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile/thread mismatch");
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

  // ---------------- per-thread operand row state ----------------
  const float* arow[NQA];
  const float* arow2[NQA];
  int a_t[NQA], a_len[NQA];
  float a_g[NQA];
  uint32_t a_key[NQA];  // DROP: per-row key of the dropout hash
  int a_col = 0;        // DROP: column of the quad being staged
#pragma unroll
  for (int q = 0; q < NQA; ++q) {
    int r = row0 + r_in + q * RPP;
    if (r > p.M - 1) r = p.M - 1;  // clamp: duplicates are discarded by the epilogue
    a_key[q] = DROP ? drop_rowkey(p.drop_seed, (uint32_t)r) : 0u;
    if constexpr (AK == A_PAIRSUM_RELU || AK == A_PAIRPROD) {
      const int j = r / p.pairB;
      const int i = r - j * p.pairB;
      arow[q] = p.A + (long)i * p.lda;
      arow2[q] = p.A2 + (long)j * p.lda2;
    } else {
      arow[q] = p.A + (long)r * p.lda;
      arow2[q] = nullptr;
    }
    if constexpr (AK == A_CONV) {
      const int b = r / p.L;
      a_t[q] = r - b * p.L;
      a_len[q] = p.lens[b];
    } else {
      a_t[q] = 0;
      a_len[q] = 0;
    }
    if constexpr (AK == A_DZ_ELEM) arow2[q] = p.A2 + (long)r * p.lda2;
    a_g[q] = 0.f;
    if constexpr (AK == A_DZ_ROWG) a_g[q] = p.gvec[r];
  }
  const float* brow[NQB];
  bool bvalid[NQB];
#pragma unroll
  for (int q = 0; q < NQB; ++q) {
    const int n = col0 + r_in + q * RPP;
    bvalid[q] = n < p.N;
    brow[q] = p.W + (long)(bvalid[q] ? n : 0) * p.ldw;
  }

  const int spt = (p.Kseg + BK - 1) / BK;  // slabs per segment
  const int nslab = p.nseg * spt;
  const bool conv_affine = (AK == A_CONV) && (p.a_scale != nullptr);

  float4 ra[NQA], ra2[NQA], rb[NQB];
  float4 rsc = make_float4(0, 0, 0, 0), rsh = make_float4(0, 0, 0, 0);
  float4 rcs = make_float4(0, 0, 0, 0), rp = make_float4(0, 0, 0, 0), rq = make_float4(0, 0, 0, 0);
  unsigned avalid = 0, bmask = 0;

  // fetch(): branch-free - every load is issued unconditionally from a clamped (always valid) address and
  // invalid lanes are zeroed by selects in commit(), so the loads of slab s+1 stay in flight under the
  // MFMAs of slab s (a predicated load would put an s_waitcnt vmcnt(0) in front of the MFMA block).
  auto fetch_a = [&](int s) {
    const int seg = s / spt;
    const int c = (s - seg * spt) * BK + 4 * kv;
    const bool kok = c < p.Kseg;
    const int cc = kok ? c : 0;
    if constexpr (DROP) a_col = cc;
    avalid = 0;
    if constexpr (AK == A_CONV) {
      const int sh = (seg - p.nseg / 2) * p.dil;
#pragma unroll
      for (int q = 0; q < NQA; ++q) {
        const int tt = a_t[q] + sh;
        const bool ok = kok && (a_t[q] < a_len[q]) && (tt >= 0) && (tt < a_len[q]);
        ra[q] = ld4(arow[q] + (long)(ok ? sh : 0) * p.lda + cc);
        avalid |= (ok ? 1u : 0u) << q;
      }
      if (conv_affine) {
        rsc = ld4(p.a_scale + cc);
        rsh = ld4(p.a_shift + cc);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQA; ++q) {
        ra[q] = ld4(arow[q] + cc);
        if constexpr (AK == A_PAIRSUM_RELU || AK == A_DZ_ELEM || AK == A_PAIRPROD) ra2[q] = ld4(arow2[q] + cc);
      }
      avalid = kok ? 0xffffffffu : 0u;
      if constexpr (AK == A_AFFINE_RELU || AK == A_DZ_ELEM || AK == A_DZ_ROWG) {
        rsc = ld4(p.a_scale + cc);
        rsh = ld4(p.a_shift + cc);
      }
      if constexpr (AK == A_DZ_ELEM || AK == A_DZ_ROWG) {
        rcs = ld4(p.dz_cs + cc);
        rp = ld4(p.dz_p + cc);
        rq = ld4(p.dz_q + cc);
      }
    }
  };
  auto fetch_b = [&](int s) {
    const int seg = s / spt;
    const int c = (s - seg * spt) * BK + 4 * kv;
    const bool kok = c < p.Kseg;
    const int cc = kok ? c : 0;
    bmask = 0;
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      rb[q] = ld4(brow[q] + (long)seg * p.Kseg + cc);
      bmask |= ((kok && bvalid[q]) ? 1u : 0u) << q;
    }
  };

  auto fetch = [&](int s) {
    fetch_a(s);
    fetch_b(s);
  };

  auto pin_a = [&]() {
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      pin4(ra[q]);
      if constexpr (AK == A_PAIRSUM_RELU || AK == A_DZ_ELEM || AK == A_PAIRPROD) pin4(ra2[q]);
    }
  };
  auto pin_b = [&]() {
#pragma unroll
    for (int q = 0; q < NQB; ++q) pin4(rb[q]);
  };
  auto pin_fetched = [&]() {
    pin_a();
    pin_b();
  };

  auto sel4 = [](bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };

  auto commit_a = [&](int buf) {
    float* As = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < NQA; ++q) {
      float4 v = ra[q];
      const bool ok = (avalid >> q) & 1u;
      if constexpr (AK == A_AFFINE_RELU) {
        v.x = relu(fmaf(v.x, rsc.x, rsh.x));
        v.y = relu(fmaf(v.y, rsc.y, rsh.y));
        v.z = relu(fmaf(v.z, rsc.z, rsh.z));
        v.w = relu(fmaf(v.w, rsc.w, rsh.w));
      } else if constexpr (AK == A_PAIRPROD) {
        v.x *= ra2[q].x;
        v.y *= ra2[q].y;
        v.z *= ra2[q].z;
        v.w *= ra2[q].w;
      } else if constexpr (AK == A_PAIRSUM_RELU) {
        v.x = relu(v.x + ra2[q].x);
        v.y = relu(v.y + ra2[q].y);
        v.z = relu(v.z + ra2[q].z);
        v.w = relu(v.w + ra2[q].w);
      } else if constexpr (AK == A_DZ_ELEM || AK == A_DZ_ROWG) {
        float4 g;
        if constexpr (AK == A_DZ_ELEM) g = ra2[q];
        else g = make_float4(a_g[q], a_g[q], a_g[q], a_g[q]);
        v.x = (fmaf(v.x, rsc.x, rsh.x) > 0.f ? g.x * rcs.x : 0.f) + fmaf(rq.x, v.x, rp.x);
        v.y = (fmaf(v.y, rsc.y, rsh.y) > 0.f ? g.y * rcs.y : 0.f) + fmaf(rq.y, v.y, rp.y);
        v.z = (fmaf(v.z, rsc.z, rsh.z) > 0.f ? g.z * rcs.z : 0.f) + fmaf(rq.z, v.z, rp.z);
        v.w = (fmaf(v.w, rsc.w, rsh.w) > 0.f ? g.w * rcs.w : 0.f) + fmaf(rq.w, v.w, rp.w);
      } else if constexpr (AK == A_CONV) {
        if (conv_affine) {
          v.x = relu(fmaf(v.x, rsc.x, rsh.x));
          v.y = relu(fmaf(v.y, rsc.y, rsh.y));
          v.z = relu(fmaf(v.z, rsc.z, rsh.z));
          v.w = relu(fmaf(v.w, rsc.w, rsh.w));
        }
      }
      if constexpr (DROP) v = drop4(v, a_key[q], (uint32_t)a_col, p.drop_thresh, p.drop_scale);
      *reinterpret_cast<float4*>(As + (r_in + q * RPP) * LDK + 4 * kv) = sel4(ok, v);
    }
  };
  auto commit_b = [&](int buf) {
    float* Bs = smem + buf * STAGE + BM * LDK;
#pragma unroll
    for (int q = 0; q < NQB; ++q) {
      *reinterpret_cast<float4*>(Bs + (r_in + q * RPP) * LDK + 4 * kv) = sel4((bmask >> q) & 1u, rb[q]);
    }
  };

  auto commit = [&](int buf) {
    commit_a(buf);
    commit_b(buf);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  auto compute = [&](int buf, auto kk0_c, auto kk1_c) {
    constexpr int KK0 = decltype(kk0_c)::value, KK1 = decltype(kk1_c)::value;
    const float* As = smem + buf * STAGE + (wm * WM * 32 + frag_row) * LDK + frag_k;
    const float* Bs = smem + buf * STAGE + BM * LDK + (wn * WN * 32 + frag_row) * LDK + frag_k;
#pragma unroll
    for (int kk = KK0; kk < KK1; ++kk) {
      float4 a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const float4*>(As + i * 32 * LDK + kk * 8);
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDK + kk * 8);
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

  // ---------------- main loop: register-prefetch double buffering, one barrier per slab ----------------
  using std::integral_constant;

  fetch(0);
  pin_fetched();
  commit(0);
  fetch(nslab > 1 ? 1 : 0);
  __syncthreads();
  // two-region pipeline (as in gemm_bf16x3.hpp): A(s+1) staged under the first half of the MFMAs, B(s+1) under the
  // second; the loads of slab s+2 are issued as soon as their registers are free - a full slab to land.  (Ablations
  // of this loop - MFMA + fragment reads only, loads only, no LDS writes, the earlier single-region loop - are
  // recorded with their numbers in DESIGN.md section 4.1; the switches are no longer compiled in.)
  for (int s = 0; s + 1 < nslab; ++s) {
    const int cur = s & 1;
    const int nxt = s + 2 < nslab ? s + 2 : s + 1;
    __builtin_amdgcn_sched_barrier(0);
    pin_a();
    compute(cur, integral_constant<int, 0>{}, integral_constant<int, BK / 16>{});
    commit_a(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_a(nxt);
    __builtin_amdgcn_sched_barrier(0);
    pin_b();
    compute(cur, integral_constant<int, BK / 16>{}, integral_constant<int, BK / 8>{});
    commit_b(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    fetch_b(nxt);
    // raw barrier: LDS traffic drained, but the global loads just issued stay in flight across it (a
    // __syncthreads() here makes hipcc wait vmcnt(0) at the top of the next slab)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  compute((nslab - 1) & 1, integral_constant<int, 0>{}, integral_constant<int, BK / 8>{});
  __syncthreads();

  // ---------------- epilogue ----------------
  gemm_epilogue<EK, WAVES_M, WAVES_N, WM, WN>(p, acc, row0, col0, tile_n, smem);
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int BK>
struct GemmCfg {
  static constexpr int BM = WAVES_M * WM * 32;
  static constexpr int BN = WAVES_N * WN * 32;
  static constexpr int NT = WAVES_M * WAVES_N * 64;
  static constexpr int LDS_BYTES = 2 * (BM + BN) * (BK + 4) * (int)sizeof(float);
};

}  // namespace pn
