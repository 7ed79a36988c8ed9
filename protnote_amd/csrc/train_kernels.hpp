// HBM-bound passes of the training path (gfx950): BatchNorm-backward column statistics over stored
// pre-activations, the layer-1 pair-grid reductions, the logits row-dot, loss + metrics, clip + Adam.
// All are streaming kernels: 16-byte coalesced loads (one wave = 1 KiB of one row), per-thread register
// accumulation.  Cross-workgroup sums never use floating-point atomics: every workgroup writes its partial to its own
// slot and k_part_reduce / k_part_final (or k_scalar_final) add the slots in a fixed order, so a training step is
// bit-reproducible run to run.
#pragma once
#include "gemm_engine.hpp"

namespace pn {

// ------------------------------------------------------------------------------------------------
// Fixed-order reduction of per-workgroup partials.  part[nparts][width] (float or double) ->
//   level 1 (k_part_reduce): out[chunk][width] = sum of `per_chunk` consecutive partial rows, sequentially, in f64
//   level 2 (k_part_final):  dst_{c / seg}[c % seg] = sum over the chunks, sequentially
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_part_reduce(const T* __restrict__ part, long nparts, int width, long per_chunk,
                                                     double* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  const long p0 = (long)blockIdx.y * per_chunk;
  long p1 = p0 + per_chunk;
  if (p1 > nparts) p1 = nparts;
  double a = 0;
#pragma unroll 8
  for (long q = p0; q < p1; ++q) a += (double)part[q * width + c];
  out[(long)blockIdx.y * width + c] = a;
}

__global__ void k_part_final(const double* __restrict__ in, int nchunk, int width, int seg, double* d0, double* d1,
                             double* d2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= width) return;
  double a = 0;
  for (int k = 0; k < nchunk; ++k) a += in[(long)k * width + c];
  const int which = c / seg;
  double* dst = which == 0 ? d0 : (which == 1 ? d1 : d2);
  dst[c - which * seg] = a;
}

// one workgroup: out[0] = sum part[0..n) in a fixed order (strided per-thread sums, then a fixed LDS tree)
__global__ __launch_bounds__(256) void k_scalar_final(const double* __restrict__ part, int n, double* out) {
  __shared__ double sh[256];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// sum of a 256-thread workgroup's per-thread doubles in a fixed order -> valid in thread 0
__device__ __forceinline__ double block_sum_256(double a) {
  __shared__ double wsum[4];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
  __syncthreads();
  return ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
}

// ------------------------------------------------------------------------------------------------
// BN-backward statistics for one layer.  With u = s*z + t, mask = u > 0, xhat = (z - mean)*invstd:
//   S1[c] = sum_r du,  S2[c] = sum_r du*xhat,   du = g*mask   (g: matrix G[r][c], or gvec[r]*w[c])
// and for the row-scalar form also dw[c] = sum_r gvec[r]*relu(u) (gradient of the output neuron).
// ZK = 1: z is not stored but regenerated on the pair grid, z[r] = A[r % pairB] + B2[r / pairB].
// grid: x = column slabs of 1024 (256 threads x float4), y = row chunks.
// ------------------------------------------------------------------------------------------------
struct StatsParams {
  long R;
  int C;
  long rows_per_block;
  const float* Z;
  long ldz;
  const float* G;
  long ldg;
  const float* gvec;
  const float* w;  // row-scalar form: output-neuron weight
  const float *s, *t, *mean, *invstd;
  const float* A;  // pair-generated z
  long lda;
  const float* B2;
  long ldb2;
  int pairB;
  double* part;  // [gridDim.y][NST][C] per-workgroup partials (NST = 3 for the row-scalar form: S1, S2, dw; else 2)
};

template <int ROWG, int ZK>
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const StatsParams p) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= p.C) return;
  const float4 s = ld4(p.s + c), t = ld4(p.t + c), mu = ld4(p.mean + c), is = ld4(p.invstd + c);
  float4 w = make_float4(1, 1, 1, 1);
  if constexpr (ROWG) w = ld4(p.w + c);
  const long r0 = (long)blockIdx.y * p.rows_per_block;
  long r1 = r0 + p.rows_per_block;
  if (r1 > p.R) r1 = p.R;
  double d1[4] = {0, 0, 0, 0}, d2[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
  for (long rb = r0; rb < r1; rb += 128) {
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, aw[4] = {0, 0, 0, 0};
    const long re = (rb + 128 < r1) ? rb + 128 : r1;
#pragma unroll 4
    for (long r = rb; r < re; ++r) {
      float4 z;
      if constexpr (ZK) {
        const long j = r / p.pairB;
        const long i = r - j * p.pairB;
        const float4 za = ld4(p.A + i * p.lda + c), zb = ld4(p.B2 + j * p.ldb2 + c);
        z = make_float4(za.x + zb.x, za.y + zb.y, za.z + zb.z, za.w + zb.w);
      } else {
        z = ld4(p.Z + r * p.ldz + c);
      }
      float4 g;
      if constexpr (ROWG) {
        const float gr = p.gvec[r];
        g = make_float4(gr * w.x, gr * w.y, gr * w.z, gr * w.w);
        aw[0] += gr * relu(fmaf(z.x, s.x, t.x));
        aw[1] += gr * relu(fmaf(z.y, s.y, t.y));
        aw[2] += gr * relu(fmaf(z.z, s.z, t.z));
        aw[3] += gr * relu(fmaf(z.w, s.w, t.w));
      } else {
        g = ld4(p.G + r * p.ldg + c);
      }
      const float u0 = fmaf(z.x, s.x, t.x) > 0.f ? g.x : 0.f;
      const float u1 = fmaf(z.y, s.y, t.y) > 0.f ? g.y : 0.f;
      const float u2 = fmaf(z.z, s.z, t.z) > 0.f ? g.z : 0.f;
      const float u3 = fmaf(z.w, s.w, t.w) > 0.f ? g.w : 0.f;
      a1[0] += u0;
      a1[1] += u1;
      a1[2] += u2;
      a1[3] += u3;
      a2[0] = fmaf(u0, (z.x - mu.x) * is.x, a2[0]);
      a2[1] = fmaf(u1, (z.y - mu.y) * is.y, a2[1]);
      a2[2] = fmaf(u2, (z.z - mu.z) * is.z, a2[2]);
      a2[3] = fmaf(u3, (z.w - mu.w) * is.w, a2[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d1[k] += a1[k];
      d2[k] += a2[k];
      dw[k] += aw[k];
    }
  }
  constexpr int NST = ROWG ? 3 : 2;
  double* o = p.part + (long)blockIdx.y * NST * p.C + c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k] = d1[k];
    o[p.C + k] = d2[k];
    if constexpr (ROWG) o[2 * p.C + k] = dw[k];
  }
}

// Materialise dz = (s*z+t > 0 ? g*cs : 0) + p + q*z for a whole layer in one streaming pass (g = G[r][c] or
// gvec[r]); `out` may alias Z or G (pure element-wise).  Both backward GEMMs of the layer then read a plain
// operand instead of regenerating dz per tile (it is consumed 24x by dW and 24x by dh).
struct DzParams {
  long R;
  int C;
  long rows_per_block;
  const float* Z;
  long ldz;
  const float* G;
  long ldg;
  const float* gvec;
  const float *s, *t, *cs, *p, *q;
  float* out;
  long ldo;
  const int* lens;     // optional: rows are positions (b, t); dz is zeroed for t >= lens[b] (MaskedConv1D output mask)
  int L;
  const float* addto;  // optional: out = addto + dz (residual branch of the encoder blocks), ld = ldo
};

template <int ROWG>
__global__ __launch_bounds__(256) void k_dz_apply(const DzParams P) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= P.C) return;
  const float4 s = ld4(P.s + c), t = ld4(P.t + c), cs = ld4(P.cs + c), pp = ld4(P.p + c), q = ld4(P.q + c);
  const long r0 = (long)blockIdx.y * P.rows_per_block;
  long r1 = r0 + P.rows_per_block;
  if (r1 > P.R) r1 = P.R;
#pragma unroll 4
  for (long r = r0; r < r1; ++r) {
    const float4 z = ld4(P.Z + r * P.ldz + c);
    float4 g;
    if constexpr (ROWG) {
      const float gr = P.gvec[r];
      g = make_float4(gr, gr, gr, gr);
    } else {
      g = ld4(P.G + r * P.ldg + c);
    }
    float4 o;
    o.x = (fmaf(z.x, s.x, t.x) > 0.f ? g.x * cs.x : 0.f) + fmaf(q.x, z.x, pp.x);
    o.y = (fmaf(z.y, s.y, t.y) > 0.f ? g.y * cs.y : 0.f) + fmaf(q.y, z.y, pp.y);
    o.z = (fmaf(z.z, s.z, t.z) > 0.f ? g.z * cs.z : 0.f) + fmaf(q.z, z.z, pp.z);
    o.w = (fmaf(z.w, s.w, t.w) > 0.f ? g.w * cs.w : 0.f) + fmaf(q.w, z.w, pp.w);
    if (P.lens != nullptr) {
      const long b = r / P.L;
      if ((int)(r - b * P.L) >= P.lens[b]) o = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (P.addto != nullptr) {
      const float4 a = ld4(P.addto + r * P.ldo + c);
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    *reinterpret_cast<float4*>(P.out + r * P.ldo + c) = o;
  }
}

// ---------------- inter-layer dropout (OUTPUT_MLP_DROPOUT > 0 in training; reference ProtNote.py:63-81,369-371) -------
// The mask is the counter-based hash of gemm_engine.hpp (drop_rowkey / drop_keep): nothing is stored, every consumer
// regenerates it from (seed, row, column).
// MODE 0: X[r][c] *= mask(r, c) * scale in place        (upstream gradient of a dropped activation; output dropout)
// MODE 1: out[r][c] = relu(s[c] * X[r][c] + t[c]) * mask * scale   (materialised dropped activation of a row MLP)
// MODE 2: out[r][c] = mask(r, c) ? 1 : 0                (pn_dropout_mask: what the tests hand to the oracle)
template <int MODE>
__global__ __launch_bounds__(256) void k_dropout(const float* __restrict__ X, long ldx, float* __restrict__ out, long ldo,
                                                 long R, int C, const float* __restrict__ s, const float* __restrict__ t,
                                                 uint32_t seed, uint32_t thresh, float scale, long rows_per_block) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  float4 sv = make_float4(1, 1, 1, 1), tv = make_float4(0, 0, 0, 0);
  if constexpr (MODE == 1) {
    sv = ld4(s + c);
    tv = ld4(t + c);
  }
  const long r0 = (long)blockIdx.y * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > R) r1 = R;
#pragma unroll 4
  for (long r = r0; r < r1; ++r) {
    const uint32_t key = drop_rowkey(seed, (uint32_t)r);
    float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (MODE != 2) v = ld4(X + r * ldx + c);
    if constexpr (MODE == 1) {
      v.x = relu(fmaf(v.x, sv.x, tv.x));
      v.y = relu(fmaf(v.y, sv.y, tv.y));
      v.z = relu(fmaf(v.z, sv.z, tv.z));
      v.w = relu(fmaf(v.w, sv.w, tv.w));
    }
    *reinterpret_cast<float4*>(out + r * ldo + c) = drop4(v, key, (uint32_t)c, thresh, MODE == 2 ? 1.f : scale);
  }
}

// ---------------- encoder backward helpers (TRAIN_SEQUENCE_ENCODER) ----------------
// gradient of the masked mean-pool (protein_encoders.py:114-117): g[p][c] = t < len ? demb[b][c] / len : 0
__global__ void k_pool_bwd(const float* __restrict__ demb, int ld_emb, const int* __restrict__ lens, int L, int C,
                           int ld, long P, float* __restrict__ g) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * ld) return;
  const long p = i / ld;
  const int c = (int)(i - p * ld);
  const int b = (int)(p / L), t = (int)(p - (long)b * L);
  const int len = lens[b];
  g[i] = (c < C && t < len) ? demb[(long)b * ld_emb + c] / (float)len : 0.f;
}

// column sums of X[P][ld] (bias gradients): part[blockIdx.y][c] = sum over this block's rows; grid (C/256, row chunks)
__global__ void k_colsum(const float* __restrict__ X, long ld, long P, int C, long rows_per_block, double* part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  double a = 0;
  for (long r = r0; r < r1; ++r) a += X[r * ld + c];
  part[(long)blockIdx.y * C + c] = a;
}

// forward-packed conv weight [Cout][k][ld4(Cin)] -> data-gradient weight [Cin][k][ld4(Cout)] with the taps reversed:
// dX[p][ci] = sum_seg sum_co dY[p + (seg - k/2) dil][co] * W[co][ci][k-1-seg]
__global__ void k_conv_w_dgrad(const float* __restrict__ packed, int Cout, int Cin, int k, int ldci, int ldco,
                               float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)Cin * k * ldco;
  if (i >= total) return;
  const int co = (int)(i % ldco);
  const int seg = (int)((i / ldco) % k);
  const int ci = (int)(i / ((long)ldco * k));
  out[i] = (co < Cout) ? packed[((long)co * k + (k - 1 - seg)) * ldci + ci] : 0.f;
}

// packed weight gradient [Mpad][k][ldci] -> torch Conv1d layout [Cout][Cin][k]
__global__ void k_unpack_conv_grad(const float* __restrict__ packed, int Cout, int Cin, int k, int ldci,
                                   float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)Cout * Cin * k;
  if (i >= total) return;
  const int tap = (int)(i % k);
  const int ci = (int)((i / k) % Cin);
  const int co = (int)(i / ((long)k * Cin));
  out[i] = packed[((long)co * k + tap) * ldci + ci];
}

// From S1,S2: dgamma = S2, dbeta = S1 and the per-column vectors of the dz generator
//   dz = (mask ? g*cs : 0) + p + q*z,   cs = s*(w or 1),  q = -s*invstd*S2/R,  p = -s*S1/R - q*mean
// (s = gamma*invstd).  Without BatchNorm (gamma == nullptr): cs = (w or 1), p = q = 0, dbias = S1.
__global__ void k_bn_bwd_finalize(const double* S1, const double* S2, const double* dwacc, double count,
                                  const double* count_dev, int C, const float* gamma, const float* s, const float* mean,
                                  const float* invstd, const float* w, float* cs, float* pv, float* qv, float* dgamma,
                                  float* dbeta, float* dw_out, int fixed_stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (count_dev != nullptr) count = *count_dev;  // SYNC_BN: the rows of all ranks (see sync_sum2)
  const float wc = w ? w[c] : 1.f;
  if (gamma != nullptr) {
    const float sc = s[c];
    const double c1 = S1[c] / count, c2 = S2[c] / count;
    // fixed_stats: eval-mode BatchNorm (running statistics are constants) - no batch-statistics terms in dz
    const float q = fixed_stats ? 0.f : (float)(-(double)sc * (double)invstd[c] * c2);
    cs[c] = sc * wc;
    qv[c] = q;
    pv[c] = fixed_stats ? 0.f : (float)(-(double)sc * c1 - (double)q * (double)mean[c]);
    if (dgamma) dgamma[c] = (float)S2[c];
    if (dbeta) dbeta[c] = (float)S1[c];
  } else {
    cs[c] = wc;
    qv[c] = 0.f;
    pv[c] = 0.f;
    if (dbeta) dbeta[c] = (float)S1[c];  // gradient of the Linear bias
  }
  if (dw_out && dwacc) dw_out[c] = (float)dwacc[c];
}

// Layer without BatchNorm (OUTPUT_MLP_BATCHNORM: False): the "fold" is u = z + bias, nothing to normalise or track.
__global__ void k_fold_nobn(const float* bias, int C, int ld, float* s, float* t, float* mean_out, float* invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  s[c] = c < C ? 1.f : 0.f;
  t[c] = (c < C && bias) ? bias[c] : 0.f;
  mean_out[c] = 0.f;
  invstd_out[c] = 1.f;
}

// BatchNorm (train) fold for the separable first pair layer: z1[i,j] = A[i] + Bm[j] over the full B x NL grid:
// mean = mean_i(A) + mean_j(Bm), biased var = var_i(A) + var_j(Bm) (cross term vanishes on a full grid).
__global__ void k_bn_fold_pair(pn_bn bn, const double* sumA, const double* sqA, double nA, const double* sumB,
                               const double* sqB, double nB, float eps, float momentum, int C, float* s, float* t,
                               float* mean_out, float* invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double ma = sumA[c] / nA, mb = sumB[c] / nB;
  double va = sqA[c] / nA - ma * ma, vb = sqB[c] / nB - mb * mb;
  if (va < 0) va = 0;
  if (vb < 0) vb = 0;
  const double mean = ma + mb, var = va + vb, count = nA * nB;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = bn.weight[c] * invstd;
  s[c] = sc;
  t[c] = bn.bias[c] - (float)mean * sc;
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const double unb = count > 1 ? var * (count / (count - 1.0)) : var;
  bn.running_mean[c] = (1.f - momentum) * bn.running_mean[c] + momentum * (float)mean;
  bn.running_var[c] = (1.f - momentum) * bn.running_var[c] + momentum * (float)unb;
}

// SYNC_BN: column sum / sum of squares of z1[i,j] = A[i] + Bm[j] over THIS rank's nA x nB grid, from the table sums
__global__ void k_pair_grid_sums(const double* sumA, const double* sqA, double nA, const double* sumB, const double* sqB,
                                 double nB, int C, double* sum, double* sumsq) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  sum[c] = nB * sumA[c] + nA * sumB[c];
  sumsq[c] = nB * sqA[c] + 2.0 * sumA[c] * sumB[c] + nA * sqB[c];
}

// ------------------------------------------------------------------------------------------------
// layer-1 reductions over the pair grid (z1 = A[i] + Bm[j] regenerated from the two small tables):
//   MODE 0: one label per workgroup row   grid (C/1024, NL)   rows of one label are contiguous
//   MODE 1: one protein per workgroup row grid (C/1024, B)    stride-B rows, 1 KiB per wave per row
// ------------------------------------------------------------------------------------------------
struct PairRedParams {
  int B, NL, C;
  const float* DH;  // [NL*B][ldh], row r = j*B + i
  long ldh;
  const float* A;
  long lda;
  const float* Bm;
  long ldb;
  const float *s, *t, *cs, *p, *q;
  float* out;
  long ldo;
  // RANK1 kernels (OUTPUT_MLP_NUM_LAYERS: 1 - the hidden layer IS the top layer): the upstream gradient is the rank-1
  // dl[r] * w_out[c], never a matrix: gvec = dl over the label-major pair grid [NL*B], w = w_out [C]; dwpart receives the
  // partial rows of dw_out[c] = sum_r dl[r] relu(s z1 + t) (one row per label chunk / per label, summed by k_colsum_rows)
  const float* gvec;
  const float* w;
  float* dwpart;
};

// ------------------------------------------------------------------------------------------------
// Layer-1 backward from two small tables of the upstream gradient.  With du = (s*z1+t > 0 ? g : 0):
//   M0[j][c] = sum_i du   (rows of one label are contiguous)
//   M1[i][c] = sum_j du
// k_pair_mask_reduce_fused produces both in ONE pass over the gradient (B <= 256); k_pair_mask_reduce<0|1> are the
// two-pass form for larger batches.
// everything BatchNorm-backward needs follows from these two small tables, because z1 = A[i] + Bm[j] is separable:
//   S1 = sum du = sum_j M0[j],   sum du*z1 = sum_i A[i] M1[i] + sum_j Bm[j] M0[j],   S2 = invstd (sum du*z1 - mean S1)
//   dBm[j] = sum_i dz1 = cs M0[j] + B p + q (sum_i A[i] + B Bm[j]),   dA[i] = cs M1[i] + NL p + q (NL A[i] + sum_j Bm[j])
// (dz1 = cs du + p + q z1 with cs, p, q as in k_bn_bwd_finalize).  The separate statistics pass over the 101 GB
// gradient is gone; the small-table work is k_pair_colsums -> k_pair_bn0_finalize -> k_pair_apply.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool RANK1 = false>
__global__ __launch_bounds__(256) void k_pair_mask_reduce(const PairRedParams P) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= P.C) return;
  const float4 s = ld4(P.s + c), t = ld4(P.t + c);
  const int fixed = blockIdx.y;
  const int n = MODE == 0 ? P.B : P.NL;
  const float4 zf = MODE == 0 ? ld4(P.Bm + (long)fixed * P.ldb + c) : ld4(P.A + (long)fixed * P.lda + c);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;  // RANK1, MODE 0: this label's share of dw_out
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  int cnt = 0;
#pragma unroll 4
  for (int k = 0; k < n; ++k) {
    const long r = MODE == 0 ? (long)fixed * P.B + k : (long)k * P.B + fixed;
    const float4 zv = MODE == 0 ? ld4(P.A + (long)k * P.lda + c) : ld4(P.Bm + (long)k * P.ldb + c);
    float4 g;
    if (RANK1) {
      const float gv = P.gvec[r];
      g = make_float4(gv, gv, gv, gv);
    } else {
      g = ld4(P.DH + r * P.ldh + c);
    }
    const float h0 = fmaf(zf.x + zv.x, s.x, t.x), h1 = fmaf(zf.y + zv.y, s.y, t.y);
    const float h2 = fmaf(zf.z + zv.z, s.z, t.z), h3 = fmaf(zf.w + zv.w, s.w, t.w);
    a0 += h0 > 0.f ? g.x : 0.f;
    a1 += h1 > 0.f ? g.y : 0.f;
    a2 += h2 > 0.f ? g.z : 0.f;
    a3 += h3 > 0.f ? g.w : 0.f;
    if (RANK1 && MODE == 0) {
      w0 = fmaf(fmaxf(h0, 0.f), g.x, w0);
      w1 = fmaf(fmaxf(h1, 0.f), g.x, w1);
      w2 = fmaf(fmaxf(h2, 0.f), g.x, w2);
      w3 = fmaf(fmaxf(h3, 0.f), g.x, w3);
    }
    if (++cnt == 256) {
      d0 += a0; d1 += a1; d2 += a2; d3 += a3;
      a0 = a1 = a2 = a3 = 0.f;
      if (RANK1 && MODE == 0) {
        e0 += w0; e1 += w1; e2 += w2; e3 += w3;
        w0 = w1 = w2 = w3 = 0.f;
      }
      cnt = 0;
    }
  }
  d0 += a0; d1 += a1; d2 += a2; d3 += a3;
  if (RANK1) {  // du = mask * dl * w_out: the column factor leaves the sums
    const float4 wv = ld4(P.w + c);
    d0 *= wv.x; d1 *= wv.y; d2 *= wv.z; d3 *= wv.w;
    if (MODE == 0) {
      e0 += w0; e1 += w1; e2 += w2; e3 += w3;
      *reinterpret_cast<float4*>(P.dwpart + (long)fixed * P.C + c) = make_float4((float)e0, (float)e1, (float)e2, (float)e3);
    }
  }
  *reinterpret_cast<float4*>(P.out + (long)fixed * P.ldo + c) =
      make_float4((float)d0, (float)d1, (float)d2, (float)d3);
}

// Both reductions in ONE pass over the gradient (round 3; B <= 256).  A workgroup of 512 threads = 16 protein groups x 32
// column quads owns a 128-column strip and a chunk of labels: thread (ig, cq) keeps A[i][c..c+3] and the M1 accumulators of
// its 16 proteins i = ig + 16 k in registers, streams the label's rows (a wave reads 2 rows x 512 B), and the per-label sum
// over proteins M0[j] is closed over the 16 protein groups through the LDS, two labels per barrier pair.  M1 leaves as one
// partial per label chunk ([chunk][B][C] f32, at most 256 labels each - the f32 run length of the two-pass kernels), added
// in chunk order by k_pair_m1_reduce: deterministic, no atomics.  Reads the 101 GB gradient once instead of twice.
constexpr int PMR_JB = 2;   // labels per barrier pair
constexpr int PMR_IG = 16;  // protein groups: 512 threads = 16 groups x 32 column quads, 16 proteins per thread
template <bool RANK1 = false>
__global__ __launch_bounds__(PMR_IG * 32) void k_pair_mask_reduce_fused(const PairRedParams P, float* __restrict__ m1part,
                                                                        int labels_per_chunk) {
  constexpr int RPT = 256 / PMR_IG;  // proteins per thread
  __shared__ float4 red[PMR_JB][PMR_IG][32];
  const int tid = threadIdx.x;
  const int cq = tid & 31, ig = tid >> 5;
  const int c_raw = blockIdx.x * 128 + cq * 4;
  const bool live = c_raw < P.C;
  const int c = live ? c_raw : 0;  // dead column quads compute on column 0 and store nothing (they must meet the barriers)
  const int j0 = blockIdx.y * labels_per_chunk;
  int j1 = j0 + labels_per_chunk;
  if (j1 > P.NL) j1 = P.NL;
  const float4 s = ld4(P.s + c), t = ld4(P.t + c);
  float4 a[RPT], m1[RPT];
  float4 dwo = make_float4(0.f, 0.f, 0.f, 0.f);  // RANK1: this thread's share of dw_out over its proteins and label chunk
  float4 wq = make_float4(1.f, 1.f, 1.f, 1.f);
  if (RANK1) wq = ld4(P.w + c);
  unsigned roff[RPT];  // element offset of protein i's row inside one label's block of the gradient (< 2^31: B * ldh)
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int i = ig + PMR_IG * k;
    // proteins past the batch: NaN pre-activation -> the "> 0" test below is false whatever the sign of s (branch-free)
    const float qn = __builtin_nanf("");
    a[k] = i < P.B ? ld4(P.A + (long)i * P.lda + c) : make_float4(qn, qn, qn, qn);
    m1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    roff[k] = (unsigned)((long)(i < P.B ? i : 0) * P.ldh + c);
  }
  for (int j = j0; j < j1; j += PMR_JB) {
#pragma unroll 1
    for (int jj = 0; jj < PMR_JB; ++jj) {
      float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j + jj < j1) {
        const float4 zb = ld4(P.Bm + (long)(j + jj) * P.ldb + c);
        if (RANK1) {
          // upstream gradient dl[r] * w_out[c]: one scalar per (protein, label), shared by the 32 column quads of a group
          const float* dlrow = P.gvec + (long)(j + jj) * P.B;  // uniform
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            const int i = ig + PMR_IG * k;
            const float gk = i < P.B ? dlrow[i] : 0.f;
            const float4 av = a[k];
            const float hx = fmaf(av.x + zb.x, s.x, t.x), hy = fmaf(av.y + zb.y, s.y, t.y);
            const float hz = fmaf(av.z + zb.z, s.z, t.z), hw = fmaf(av.w + zb.w, s.w, t.w);
            const float dx = hx > 0.f ? gk : 0.f, dy = hy > 0.f ? gk : 0.f;
            const float dz = hz > 0.f ? gk : 0.f, dw = hw > 0.f ? gk : 0.f;
            m0.x += dx; m0.y += dy; m0.z += dz; m0.w += dw;
            float4& acc = m1[k];
            acc.x += dx; acc.y += dy; acc.z += dz; acc.w += dw;
            dwo.x = fmaf(fmaxf(hx, 0.f), gk, dwo.x);  // (fmaxf(NaN, 0) = 0: the proteins past the batch add nothing)
            dwo.y = fmaf(fmaxf(hy, 0.f), gk, dwo.y);
            dwo.z = fmaf(fmaxf(hz, 0.f), gk, dwo.z);
            dwo.w = fmaf(fmaxf(hw, 0.f), gk, dwo.w);
          }
          m0.x *= wq.x; m0.y *= wq.y; m0.z *= wq.z; m0.w *= wq.w;
        } else {
        const float* base = P.DH + (long)(j + jj) * P.B * P.ldh;  // uniform
#pragma unroll
        for (int k0 = 0; k0 < RPT; k0 += 8) {  // eight 16-byte loads in flight per thread
          float4 g[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = ld4(base + roff[k0 + k]);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float4 av = a[k0 + k];
            const float dx = fmaf(av.x + zb.x, s.x, t.x) > 0.f ? g[k].x : 0.f;
            const float dy = fmaf(av.y + zb.y, s.y, t.y) > 0.f ? g[k].y : 0.f;
            const float dz = fmaf(av.z + zb.z, s.z, t.z) > 0.f ? g[k].z : 0.f;
            const float dw = fmaf(av.w + zb.w, s.w, t.w) > 0.f ? g[k].w : 0.f;
            m0.x += dx; m0.y += dy; m0.z += dz; m0.w += dw;
            float4& acc = m1[k0 + k];
            acc.x += dx; acc.y += dy; acc.z += dz; acc.w += dw;
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the next batch of loads behind this batch's arithmetic (registers)
        }
        }
      }
      red[jj][ig][cq] = m0;
    }
    __syncthreads();
    if (tid < PMR_JB * 32) {  // fixed-order sum over the protein groups
      const int jj = tid >> 5, q = tid & 31;
      const int cc = blockIdx.x * 128 + q * 4;
      if (j + jj < j1 && cc < P.C) {
        float4 v = red[jj][0][q];
#pragma unroll
        for (int g2 = 1; g2 < PMR_IG; ++g2) {
          const float4 w = red[jj][g2][q];
          v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        *reinterpret_cast<float4*>(P.out + (long)(j + jj) * P.ldo + cc) = v;
      }
    }
    __syncthreads();
  }
  if (live) {
    float* dst = m1part + ((long)blockIdx.y * P.B) * P.C + c;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int i = ig + PMR_IG * k;
      if (i < P.B) {
        float4 v = m1[k];
        if (RANK1) { v.x *= wq.x; v.y *= wq.y; v.z *= wq.z; v.w *= wq.w; }
        *reinterpret_cast<float4*>(dst + (long)i * P.C) = v;
      }
    }
  }
  if (RANK1) {  // this label chunk's row of the dw_out partials: fixed-order sum over the protein groups
    red[0][ig][cq] = dwo;
    __syncthreads();
    if (tid < 32) {
      const int cc = blockIdx.x * 128 + tid * 4;
      if (cc < P.C) {
        float4 v = red[0][0][tid];
#pragma unroll
        for (int g2 = 1; g2 < PMR_IG; ++g2) {
          const float4 w2 = red[0][g2][tid];
          v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w;
        }
        *reinterpret_cast<float4*>(P.dwpart + (long)blockIdx.y * P.C + cc) = v;
      }
    }
  }
}

// out[c] = sum over rows of part[row][c], f64, row order (dw_out of the one-hidden-layer head from its partial rows)
__global__ void k_colsum_rows(const float* __restrict__ part, long nrows, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0;
  for (long r = 0; r < nrows; ++r) a += (double)part[r * C + c];
  out[c] = (float)a;
}

// OUTPUT_MLP_NUM_LAYERS: 1 - the whole output MLP over the pair grid in one pass, no pair-grid GEMM:
//   out[j*B + i] = b + sum_c w[c] * relu(Ap[i][c] + Bp[j][c])        (Ap = s*A1 + t, Bp = s*B1: BatchNorm folded in)
// 64 proteins x 64 labels per workgroup, columns staged through the LDS 32 at a time (transposed: a thread reads the 4
// proteins / 4 labels of its 4 x 4 pairs as one 16-byte LDS read each); every pair sums its columns in index order.
__global__ __launch_bounds__(256) void k_pairsum_rowdot(const float* __restrict__ Ap, long lda, const float* __restrict__ Bp,
                                                        long ldb, int B, int NL, int C, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out) {
  constexpr int T = 64, KC = 32;
  __shared__ float As[KC][T + 4];
  __shared__ float Bs[KC][T + 4];
  __shared__ float Ws[KC];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.x * T, j0 = blockIdx.y * T;
  float acc[4][4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) acc[jj][ii] = 0.f;
  for (int c0 = 0; c0 < C; c0 += KC) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = tid + 256 * q, row = idx >> 3, cq = (idx & 7) * 4;
      const bool cin = c0 + cq < C;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 av = (cin && i0 + row < B) ? ld4(Ap + (long)(i0 + row) * lda + c0 + cq) : z4;
      const float4 bv = (cin && j0 + row < NL) ? ld4(Bp + (long)(j0 + row) * ldb + c0 + cq) : z4;
      As[cq][row] = av.x; As[cq + 1][row] = av.y; As[cq + 2][row] = av.z; As[cq + 3][row] = av.w;
      Bs[cq][row] = bv.x; Bs[cq + 1][row] = bv.y; Bs[cq + 2][row] = bv.z; Bs[cq + 3][row] = bv.w;
    }
    if (tid < KC) Ws[tid] = c0 + tid < C ? w[c0 + tid] : 0.f;
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < KC; ++c) {
      const float4 av = *reinterpret_cast<const float4*>(&As[c][tx * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[c][ty * 4]);
      const float wv = Ws[c];
      const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) acc[jj][ii] = fmaf(fmaxf(a4[ii] + b4[jj], 0.f), wv, acc[jj][ii]);
    }
    __syncthreads();
  }
  const float bias = b[0];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int j = j0 + ty * 4 + jj;
    if (j >= NL) continue;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int i = i0 + tx * 4 + ii;
      if (i < B) out[(long)j * B + i] = acc[jj][ii] + bias;
    }
  }
}

// hidden[r = j*B + i][c] = relu(Ap[i][c] + Bp[j][c]): the penultimate activations of the one-hidden-layer head
// (save_embeddings, small evaluation subsets only)
__global__ void k_pairsum_relu_rows(const float* __restrict__ Ap, long lda, const float* __restrict__ Bp, long ldb, int B,
                                    long R, int C, float* __restrict__ out, long ldo) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= R * C) return;
  const long r = e / C;
  const int c = (int)(e - r * C);
  const long j = r / B;
  const int i = (int)(r - j * B);
  out[r * ldo + c] = fmaxf(Ap[(long)i * lda + c] + Bp[j * ldb + c], 0.f);
}

// M1[i][c] = sum over the label chunks of k_pair_mask_reduce_fused's partials, in chunk order (f64)
__global__ void k_pair_m1_reduce(const float* __restrict__ part, int nchunk, long BC, int C, float* __restrict__ out, long ldo) {
  const long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e >= BC) return;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  for (int k = 0; k < nchunk; ++k) {
    const float4 v = ld4(part + (long)k * BC + e);
    d0 += v.x; d1 += v.y; d2 += v.z; d3 += v.w;
  }
  const long i = e / C;
  const int c = (int)(e - i * C);
  *reinterpret_cast<float4*>(out + i * ldo + c) = make_float4((float)d0, (float)d1, (float)d2, (float)d3);
}

// per column and label chunk: sum_j M0[j], sum_j Bm[j] M0[j], sum_j Bm[j]  -> out[chunk][3][C] (f64)
__global__ __launch_bounds__(256) void k_pair_colsums(const float* __restrict__ M0, long ldm, const float* __restrict__ Bm,
                                                      long ldb, int NL, int C, int per_chunk, double* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int j0 = blockIdx.y * per_chunk;
  int j1 = j0 + per_chunk;
  if (j1 > NL) j1 = NL;
  double s1 = 0, sbm = 0, sb = 0;
#pragma unroll 8
  for (int j = j0; j < j1; ++j) {
    const double m = M0[(long)j * ldm + c], b = Bm[(long)j * ldb + c];
    s1 += m;
    sbm += b * m;
    sb += b;
  }
  double* o = out + (long)blockIdx.y * 3 * C + c;
  o[0] = s1;
  o[C] = sbm;
  o[2 * C] = sb;
}

// S1, S2 of the separable layer from the chunk sums and the [B][C] tables; then cs, p, q, dgamma, dbeta exactly as
// k_bn_bwd_finalize, plus sum_i A[i] and sum_j Bm[j] for k_pair_apply
__global__ void k_pair_bn0_finalize(const double* __restrict__ chunks, int nchunk, const float* __restrict__ A, long lda,
                                    const float* __restrict__ M1, long ldm1, int B, int NL, int C,
                                    const float* gamma, const float* s, const float* mean, const float* invstd,
                                    float* cs, float* pv, float* qv, float* dgamma, float* dbeta, double* sumA,
                                    double* sumB, double* s12_out, int fixed_stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double S1 = 0, T = 0, sb = 0;
  for (int k = 0; k < nchunk; ++k) {
    S1 += chunks[(long)k * 3 * C + c];
    T += chunks[(long)k * 3 * C + C + c];
    sb += chunks[(long)k * 3 * C + 2 * C + c];
  }
  double sa = 0;
  for (int i = 0; i < B; ++i) {
    const double a = A[(long)i * lda + c];
    sa += a;
    T += a * (double)M1[(long)i * ldm1 + c];
  }
  sumA[c] = sa;
  sumB[c] = sb;
  const double count = (double)B * (double)NL;
  if (gamma != nullptr) {
    const double S2 = (double)invstd[c] * (T - (double)mean[c] * S1);
    if (s12_out) {  // SYNC_BN: this rank's S1 / S2, to be summed over the ranks
      s12_out[c] = S1;
      s12_out[C + c] = S2;
    }
    const float sc = s[c];
    const float q = fixed_stats ? 0.f : (float)(-(double)sc * (double)invstd[c] * (S2 / count));
    cs[c] = sc;
    qv[c] = q;
    pv[c] = fixed_stats ? 0.f : (float)(-(double)sc * (S1 / count) - (double)q * (double)mean[c]);
    if (dgamma) dgamma[c] = (float)S2;
    if (dbeta) dbeta[c] = (float)S1;
  } else {  // no BatchNorm: dz1 = du, the slot carries the Linear-bias gradient
    cs[c] = 1.f;
    qv[c] = 0.f;
    pv[c] = 0.f;
    if (dbeta) dbeta[c] = (float)S1;
  }
}

// in place: X[r][c] = cs M[r][c] + n_other p + q (sum_other + n_other Z[r][c]);  (M0, Bm, sum_i A, B) or (M1, A, sum_j Bm, NL)
__global__ void k_pair_apply(float* __restrict__ M, long ldm, const float* __restrict__ Z, long ldz, long rows, int C,
                             const float* __restrict__ cs, const float* __restrict__ pv, const float* __restrict__ qv,
                             const double* __restrict__ sum_other, double n_other) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  const double v = (double)cs[c] * (double)M[r * ldm + c] + n_other * (double)pv[c] +
                   (double)qv[c] * (sum_other[c] + n_other * (double)Z[r * ldz + c]);
  M[r * ldm + c] = (float)v;
}

// plain reductions of a stored pair-grid matrix X[r = j*B + i][c] (concatenation_prod backward):
//   MODE 0: out[j][c] (+)= sum_i X[r][c] * (mul ? mul[i][c] : 1)     grid (C/1024, NL)
//   MODE 1: out[i][c] (+)= sum_j X[r][c] * (mul ? mul[j][c] : 1)     grid (C/1024, B)
// `accumulate` adds to what out already holds (the dense-GEMM part of the same gradient).
template <int MODE>
__global__ __launch_bounds__(256) void k_pair_sum(const float* __restrict__ X, long ldx, int B, int NL, int C,
                                                   const float* __restrict__ mul, long ldm, float* __restrict__ out,
                                                   long ldo, int accumulate) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  const int fixed = blockIdx.y;
  const int n = MODE == 0 ? B : NL;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int cnt = 0;
#pragma unroll 4
  for (int k = 0; k < n; ++k) {
    const long r = MODE == 0 ? (long)fixed * B + k : (long)k * B + fixed;
    float4 x = ld4(X + r * ldx + c);
    if (mul) {
      const float4 m = ld4(mul + (long)k * ldm + c);
      x.x *= m.x; x.y *= m.y; x.z *= m.z; x.w *= m.w;
    }
    a0 += x.x; a1 += x.y; a2 += x.z; a3 += x.w;
    if (++cnt == 256) {
      d0 += a0; d1 += a1; d2 += a2; d3 += a3;
      a0 = a1 = a2 = a3 = 0.f;
      cnt = 0;
    }
  }
  d0 += a0; d1 += a1; d2 += a2; d3 += a3;
  float4* o = reinterpret_cast<float4*>(out + (long)fixed * ldo + c);
  float4 v = make_float4((float)d0, (float)d1, (float)d2, (float)d3);
  if (accumulate) {
    const float4 old = *o;
    v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
  }
  *o = v;
}

// logits of the pair grid from the stored last pre-activation: out[r] = b + sum_c relu(s*z[r][c]+t) * w[c];
// one wave per row.
__global__ __launch_bounds__(256) void k_rowdot_rows(const float* __restrict__ Z, long ldz, long R, int C,
                                                      const float* __restrict__ s, const float* __restrict__ t,
                                                      const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ out) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= R) return;
  float a = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const float4 z = ld4(Z + r * ldz + c), sv = ld4(s + c), tv = ld4(t + c), wv = ld4(w + c);
    a = fmaf(relu(fmaf(z.x, sv.x, tv.x)), wv.x, a);
    a = fmaf(relu(fmaf(z.y, sv.y, tv.y)), wv.y, a);
    a = fmaf(relu(fmaf(z.z, sv.z, tv.z)), wv.z, a);
    a = fmaf(relu(fmaf(z.w, sv.w, tv.w)), wv.w, a);
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) out[r] = a + b[0];
}

// The same for C <= 3072 with the three per-column vectors held in registers (round 3): the kernel above re-reads s, t, w
// (36 KB) from the L1 for every 12 KB row - three of its four load instructions - and starts a 4-row workgroup per 48 KB.
// Here a wave keeps its 12 column quads of s, t, w (144 registers), walks `rows_per_wave` rows with the whole next row in
// flight, and only the row itself is loaded.  Same products in the same order: bit-identical logits.
// [measured] 22.6 -> 16.3 ms for the 101 GB of z3 (6.2 TB/s).
__global__ __launch_bounds__(256) void k_rowdot_rows_reg(const float* __restrict__ Z, long ldz, long R, int C,
                                                          const float* __restrict__ s, const float* __restrict__ t,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          float* __restrict__ out, int rows_per_wave) {
  constexpr int NQ = 12;
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long r0 = wave * rows_per_wave;
  if (r0 >= R) return;
  long r1 = r0 + rows_per_wave;
  if (r1 > R) r1 = R;
  float4 sv[NQ], tv[NQ], wv[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) {
    const int c = lane * 4 + 256 * k;
    const bool in = c < C;  // columns past C: weight 0 (and the row loads are clamped to column 0)
    sv[k] = in ? ld4(s + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    tv[k] = in ? ld4(t + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    wv[k] = in ? ld4(w + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bias = b[0];
  float4 z[NQ];
#pragma unroll
  for (int k = 0; k < NQ; ++k) z[k] = ld4(Z + r0 * ldz + (lane * 4 + 256 * k < C ? lane * 4 + 256 * k : 0));
  for (long r = r0; r < r1; ++r) {
    float a = 0.f;
    const long rn = r + 1 < r1 ? r + 1 : r;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const float4 zc = z[k];
      z[k] = ld4(Z + rn * ldz + (lane * 4 + 256 * k < C ? lane * 4 + 256 * k : 0));  // the next row, behind this one's math
      if (lane * 4 + 256 * k < C) {
        a = fmaf(relu(fmaf(zc.x, sv[k].x, tv[k].x)), wv[k].x, a);
        a = fmaf(relu(fmaf(zc.y, sv[k].y, tv[k].y)), wv[k].y, a);
        a = fmaf(relu(fmaf(zc.z, sv[k].z, tv[k].z)), wv[k].z, a);
        a = fmaf(relu(fmaf(zc.w, sv[k].w, tv[k].w)), wv[k].w, a);
      }
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) out[r] = a + bias;
  }
}

// dst[c][r] (ld = ldd) = src[r][c] (ld = lds); 32x32 LDS tiles
__global__ void k_transpose(const float* __restrict__ src, long lds_, int rows, int cols, float* __restrict__ dst,
                            long ldd) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < rows && c < cols) ? src[(long)r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * ldd + r] = tile[tx][k];
  }
}

// f64 sum of a float vector: part[blockIdx.x] = this workgroup's share (256 threads; k_scalar_final adds them up)
__global__ __launch_bounds__(256) void k_sum(const float* __restrict__ x, long n, double* part) {
  double a = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += x[i];
  a = block_sum_256(a);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

__global__ void k_d2f(const double* in, float* out, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)(in[i] * (double)scale);
}

// ------------------------------------------------------------------------------------------------
// loss forward + dL/dlogit + per-label TP/FN/FP in one pass over logits [B][N]
//   (reference utils/losses.py:190-213,275-276 and ProtNoteTrainer.py:61-83)
// kind 0: BCEWithLogits(pos_weight) mean; kind 1: focal (gamma, alpha < 0 = off, label smoothing) mean.
// targets: f32 or i64 multihots.  grad is d(mean loss)/dlogit * grad_scale.
// grid: x over columns (256 per block), y over row chunks; per-thread column accumulation.
// ------------------------------------------------------------------------------------------------
struct LossParams {
  const float* logits;
  const float* tf;     // float targets or null
  const int64_t* ti;   // int64 targets or null
  const uint8_t* tu;   // uint8 targets or null (1 B per pair: the algorithmic size of a multihot)
  int B, N;
  int kind;
  float pos_weight, gamma, alpha, smoothing;
  float threshold;     // on the probability
  float grad_scale;    // 1/(B*N)
  float* dlogits;      // [B][N] or null
  double* loss_part;   // [gridDim.y * gridDim.x] per-workgroup loss sums
  float *tp, *fn, *fp; // [N] accumulators (+=) or null
  int rows_per_block;
  const float* row_w;   // [B] element weight of row i (WeightedBCE / CBLoss, losses.py:214-241) or null
  const float* posneg;  // [2] = (weight of positives, weight of negatives) (BatchWeightedBCE) or null
};

// one multihot target as float from whichever array the caller passed (exactly one is non-null)
__device__ __forceinline__ float load_target(const float* tf, const int64_t* ti, const uint8_t* tu, long idx) {
  return tf ? tf[idx] : (ti ? (float)ti[idx] : (float)tu[idx]);
}

// number of positive targets -> (w_pos, w_neg) of BatchWeightedBCE (losses.py:131-139)
__global__ void k_posneg_weights(const double* npos_in, double numel, double eps, float* out) {
  const double num_pos = npos_in[0] + eps;
  const double num_neg = numel - num_pos + eps;
  const double total = num_pos + num_neg;
  out[0] = (float)((1.0 / num_pos) * (total / 2.0));
  out[1] = (float)((1.0 / num_neg) * (total / 2.0));
}

// row_w[i] = sum_j label_weights[j] * target[i][j];  npos += sum of targets  (one wave per row)
__global__ __launch_bounds__(256) void k_target_weights(const float* tf, const int64_t* ti, const uint8_t* tu, int B, int N,
                                                        const float* label_weights, float* row_w, double* npos) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  float w = 0.f, c = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float y = load_target(tf, ti, tu, (long)i * N + j);
    c += y;
    if (label_weights) w += label_weights[j] * y;
  }
  for (int o = 32; o > 0; o >>= 1) {
    w += __shfl_xor(w, o);
    c += __shfl_xor(c, o);
  }
  if (lane == 0) {
    if (row_w) row_w[i] = w;
    if (npos) atomicAdd(npos, (double)c);
  }
}

// RGDBCE as the reference computes it (mean loss m re-weighted by exp(min(m, T) / (T + 1)), factor detached)
__global__ void k_rgd_scale(const double* loss_sum, double inv_count, float temperature, float* dlogits, long n,
                            float* loss_out) {
  const float m = (float)(loss_sum[0] * inv_count);
  const float f = expf(fminf(m, temperature) / (temperature + 1.f));
  const long stride = (long)gridDim.x * blockDim.x;
  if (dlogits)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dlogits[i] *= f;
  if (blockIdx.x == 0 && threadIdx.x == 0) loss_out[0] = m * f;
}

__device__ __forceinline__ float softplusf(float x) {  // log(1+exp(x)), stable
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(256) void k_loss(const LossParams p) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * p.rows_per_block;
  int i1 = i0 + p.rows_per_block;
  if (i1 > p.B) i1 = p.B;
  double lsum = 0;
  float tp = 0.f, fn = 0.f, fp = 0.f;
  if (j < p.N) {
    // (the pass is latency-bound - 8 row blocks of 32 dependent iterations: the loads of 8 rows go out back to back)
    constexpr int U = 8;
    for (int ib = i0; ib < i1; ib += U) {
      float xs[U], ys[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int iu = ib + u < i1 ? ib + u : i1 - 1;
        const long idx = (long)iu * p.N + j;
        xs[u] = p.logits[idx];
        ys[u] = load_target(p.tf, p.ti, p.tu, idx);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
      const int i = ib + u;
      if (i >= i1) break;
      const long idx = (long)i * p.N + j;
      const float x = xs[u];
      const float y = ys[u];
      float l, g;
      const float sp_pos = softplusf(x);   // -log(1-sigmoid)
      const float sp_neg = softplusf(-x);  // -log(sigmoid)
      const float sig = 1.f / (1.f + expf(-x));
      if (p.kind == 0) {
        // torch BCEWithLogits: pw*y*softplus(-x) + (1-y)*softplus(x)
        l = p.pos_weight * y * sp_neg + (1.f - y) * sp_pos;
        g = (1.f - y) * sig - p.pos_weight * y * (1.f - sig);
      } else {
        const float ys = p.smoothing > 0.f ? y * (1.f - p.smoothing) + (1.f - y) * p.smoothing : y;
        const float bce = ys * sp_neg + (1.f - ys) * sp_pos;
        const float dbce = sig - ys;
        const float pt = expf(-bce);
        const float om = 1.f - pt;
        float mod, dmod;  // (1-pt)^gamma and its derivative wrt bce: gamma*(1-pt)^(gamma-1)*pt
        if (p.gamma == 2.f) {
          mod = om * om;
          dmod = 2.f * om * pt;
        } else if (p.gamma == 0.f) {
          mod = 1.f;
          dmod = 0.f;
        } else {
          mod = powf(om, p.gamma);
          dmod = om > 0.f ? p.gamma * powf(om, p.gamma - 1.f) * pt : 0.f;
        }
        l = mod * bce;
        g = (dmod * bce + mod) * dbce;
        if (p.alpha >= 0.f) {
          const float at = p.alpha * ys + (1.f - p.alpha) * (1.f - ys);
          l *= at;
          g *= at;
        }
      }
      if (p.row_w || p.posneg) {
        const float wgt = p.row_w ? p.row_w[i] : (y * p.posneg[0] + (1.f - y) * p.posneg[1]);
        l *= wgt;
        g *= wgt;
      }
      lsum += (double)l;
      if (p.dlogits) p.dlogits[idx] = g * p.grad_scale;
      if (p.tp) {
        const float pred = sig >= p.threshold ? 1.f : 0.f;
        tp += pred * y;
        fn += (1.f - pred) * y;
        fp += pred * (1.f - y);
      }
      }
    }
    if (p.tp) {
      atomicAdd(&p.tp[j], tp);
      atomicAdd(&p.fn[j], fn);
      atomicAdd(&p.fp[j], fp);
    }
  }
  lsum = block_sum_256(lsum);
  if (threadIdx.x == 0) p.loss_part[(long)blockIdx.y * gridDim.x + blockIdx.x] = lsum;
}

// SupCon (reference utils/losses.py:7-56; marked "not currently using" there): for each protein row the mean
// log-softmax (over the label axis) of its positive labels; loss = -mean over rows.  One workgroup per row:
//   lp_ij = x_ij - max_i - log sum_j exp(x_ij - max_i),  m_i = sum_j y_ij lp_ij / n_i  (0/0 -> 0 by nan_to_num),
//   dL/dx_ij = -(1/B) (y_ij / n_i - softmax_ij).
// Quirk kept: a row WITHOUT positives contributes 0 to the loss but NaN to the gradient - the reference's nan_to_num
// zeroes the forward value while autograd still multiplies 0 by 1/n_i = inf.  row_loss[i] is summed in a fixed order.
__global__ __launch_bounds__(256) void k_supcon(const float* __restrict__ logits, const float* __restrict__ tf,
                                                const int64_t* __restrict__ ti, int B, int N, float* __restrict__ dlogits,
                                                double* __restrict__ row_loss) {
  __shared__ float sh[256];
  __shared__ double shd[256];
  const int i = blockIdx.x;
  const float* x = logits + (long)i * N;
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < N; j += 256) mx = fmaxf(mx, x[j]);
  sh[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  mx = sh[0];
  __syncthreads();
  double se = 0, syx = 0, n = 0;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float y = tf ? tf[(long)i * N + j] : (float)ti[(long)i * N + j];
    se += (double)expf(x[j] - mx);
    syx += (double)(y * (x[j] - mx));
    n += (double)y;
  }
  double* acc[3] = {&se, &syx, &n};
  double tot[3];
  for (int k = 0; k < 3; ++k) {
    shd[threadIdx.x] = *acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o];
      __syncthreads();
    }
    tot[k] = shd[0];
    __syncthreads();
  }
  const float lse = logf((float)tot[0]);
  const float npos = (float)tot[2];
  if (threadIdx.x == 0) row_loss[i] = npos > 0.f ? -((double)((float)tot[1] / npos - lse)) : 0.0;
  if (dlogits) {
    const float invB = 1.f / (float)B;
    for (int j = threadIdx.x; j < N; j += 256) {
      const float y = tf ? tf[(long)i * N + j] : (float)ti[(long)i * N + j];
      const float sm = expf(x[j] - mx - lse);
      dlogits[(long)i * N + j] = npos > 0.f ? -invB * (y / npos - sm) : NAN;
    }
  }
}

// calculate_tp_fn_fp on probabilities (ProtNoteTrainer.py:61-83): counts are integers held in f32
__global__ __launch_bounds__(256) void k_tp_fn_fp(const float* __restrict__ probs, const float* tf, const int64_t* ti,
                                                   const uint8_t* tu, int B, int N, float threshold, float* tp, float* fn,
                                                   float* fp, int rows_per_block) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  const int i0 = blockIdx.y * rows_per_block;
  int i1 = i0 + rows_per_block;
  if (i1 > B) i1 = B;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = i0; i < i1; ++i) {
    const long idx = (long)i * N + j;
    const float y = load_target(tf, ti, tu, idx);
    const float pred = probs[idx] >= threshold ? 1.f : 0.f;
    a += pred * y;
    b += (1.f - pred) * y;
    c += pred * (1.f - y);
  }
  atomicAdd(&tp[j], a);
  atomicAdd(&fn[j], b);
  atomicAdd(&fp[j], c);
}

// ------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam (ProtNoteTrainer.py:745-755) on flat f32 buffers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, long n, double* part) {
  double a = 0;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = ld4(g + i);
      a += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (long k = i; k < n; ++k) a += (double)g[k] * g[k];
    }
  }
  a = block_sum_256(a);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}

// coef = min(max_norm / (sqrt(sumsq) + 1e-6), 1)  (torch.nn.utils.clip_grad_norm_); max_norm <= 0: no clipping
__global__ void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       long n, const double* sumsq, float max_norm, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, float weight_decay, float* norm_out) {
  float coef = 1.f;
  const float norm = sumsq ? (float)sqrt(*sumsq) : 0.f;
  if (sumsq && max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gi = g[i] * coef;
    float wi = w[i];
    if (weight_decay != 0.f) wi *= (1.f - lr * weight_decay);  // AdamW decoupled decay
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = wi - (lr / bc1) * (mi / denom);
  }
}

// torch.optim.SGD(lr, momentum, weight_decay) after the same clipping (ProtNoteTrainer.py:238-243 builds it with the
// defaults momentum = 0, dampening = 0, nesterov = False): g' = coef g + wd w; buf = g' on the first step, else
// momentum buf + g'; w -= lr (buf | g').
__global__ void k_sgd(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ buf, long n,
                      const double* sumsq, float max_norm, float lr, float momentum, float weight_decay, int first,
                      float* norm_out) {
  float coef = 1.f;
  const float norm = sumsq ? (float)sqrt(*sumsq) : 0.f;
  if (sumsq && max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float wi = w[i];
    float gi = g[i] * coef;
    if (weight_decay != 0.f) gi = gi + weight_decay * wi;
    if (buf != nullptr) {
      gi = first ? gi : momentum * buf[i] + gi;
      buf[i] = gi;
    }
    w[i] = wi - lr * gi;
  }
}

}  // namespace pn
