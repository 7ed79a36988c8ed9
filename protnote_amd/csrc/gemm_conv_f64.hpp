// Encoder convolutions with float64 accumulation on the matrix cores (v_mfma_f64_16x16x4_f64) - the forward of a
// TRAINABLE encoder only (pn_encoder_fwd_train: TRAIN_SEQUENCE_ENCODER: True, reference ProtNote.py:248-256).
//
// Why: a convolution output of the reference width sums K = 9 x 1100 = 9900 products.  v_mfma_f32_32x32x2_f32 does that in
// one k-ordered f32 chain (~3e-6 relative error), stock torch / MIOpen on this GPU is in the same class, the reference's
// f32 CPU kernels (16+ interleaved chains and a tree) reach ~1e-7.  The forward does not care (embeddings 1e-4 inside the
// bound either way) but the BACKWARD does: each of the ~2e7 BatchNorm+ReLU pre-activations that lands within that error
// of zero flips its mask against the float64 ground truth, and one flip is a ~5e-4 relative step in conv1's gradient
// (measured, tools/encoder_grad_error.py: CPU f32 2.7e-6, MIOpen f32 3.1e-3, f32-MFMA 3.3e-3 on conv1.weight).  With the
// sum carried in float64 and rounded to f32 once, the stored pre-activations are the correctly rounded ones: the masks
// are those of the ground truth up to half an ulp, and the gradient error falls into the CPU's class.
//
// Operands are the staged images of gemm_conv_dma.hpp (k_conv_stage_act: relu(bn(x)) masked, K padded to 32, guard rows
// between sequences, so a tap shift is a row offset; k_conv_relay_weight with rows padded to 64): f32 in HBM and LDS,
// widened to f64 per fragment.  128 x 64 tiles, 4 waves of 64 x 32 (4 x 2 MFMA tiles, 64 accumulator registers).
#pragma once
#include "gemm_conv_dma.hpp"

namespace pn {

struct ConvF64Params {
  const float* H;   // staged activation, row of (b = 0, t = 0)
  long ldh;         // = Kpad
  int Lp;           // row pitch of a sequence in H
  const float* W;   // re-laid weights [round64(Cout)][ntap * Kpad]
  long ldw;
  int M, N, Nstore; // output rows B*L, true Cout, columns written (pad lanes as 0)
  int ntap, Kpad, dil, L;
  const int* lens;
  const float* bias;
  const float* resid;
  long ldr;
  float* C;
  long ldc;
};

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gemm_conv_f64_kernel(const ConvF64Params p) {
  constexpr int BM = 128, BN = 64, BK = 32, LD = BK + 1;
  __shared__ float As[BM * LD];
  __shared__ float Bs[BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.Nstore + BN - 1) / BN;
  const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;  // column tiles of one row panel run together (L2)
  const int row0 = tile_m * BM, col0 = tile_n * BN;

  const int lr = tid >> 3, kq = (tid & 7) * 4;
  long arow[4], brow[2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    int pr = row0 + lr + 32 * it;
    if (pr > p.M - 1) pr = p.M - 1;  // duplicates are discarded by the epilogue
    const int b = pr / p.L;
    arow[it] = ((long)b * p.Lp + (pr - b * p.L)) * p.ldh + kq;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) brow[it] = (long)(col0 + lr + 32 * it) * p.ldw + kq;  // rows < round64(Cout) exist

  const int spt = p.Kpad / BK, nslab = p.ntap * spt;
  const long tap_step = (long)p.dil * p.ldh;
  float4 ra[4], rb[2];
  auto fetch = [&](int s) {
    const int tap = s / spt;
    const long aoff = (long)(tap - p.ntap / 2) * tap_step + (long)(s - tap * spt) * BK;  // guard rows: zeros
    const long boff = (long)s * BK;
#pragma unroll
    for (int it = 0; it < 4; ++it) ra[it] = ld4(p.H + arow[it] + aoff);
#pragma unroll
    for (int it = 0; it < 2; ++it) rb[it] = ld4(p.W + brow[it] + boff);
  };
  auto commit = [&]() {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      float* d = As + (lr + 32 * it) * LD + kq;
      d[0] = ra[it].x; d[1] = ra[it].y; d[2] = ra[it].z; d[3] = ra[it].w;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float* d = Bs + (lr + 32 * it) * LD + kq;
      d[0] = rb[it].x; d[1] = rb[it].y; d[2] = rb[it].z; d[3] = rb[it].w;
    }
  };

  f64x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  // A fragment: row = lane & 15, k = lane >> 4; B fragment: column = lane & 15, k = lane >> 4 (one f64 per lane)
  const float* fa = As + (wm * 64 + (lane & 15)) * LD + (lane >> 4);
  const float* fb = Bs + (wn * 32 + (lane & 15)) * LD + (lane >> 4);

  fetch(0);
  commit();
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const bool more = s + 1 < nslab;
    if (more) fetch(s + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = (double)fa[i * 16 * LD + kk * 4];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = (double)fb[j * 16 * LD + kk * 4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) commit();
    __syncthreads();
  }

  // epilogue (E_CONV of gemm_engine.hpp): bias, [0, len) mask, residual; C/D layout of the f64 MFMA: column = lane & 15,
  // row = (lane >> 4) + 4 * reg
  const bool has_res = p.resid != nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = col0 + wn * 32 + j * 16 + (lane & 15);
    const bool cok = col < p.N;
    const double bj = (cok && p.bias) ? (double)p.bias[col] : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = row0 + wm * 64 + i * 16 + (lane >> 4) + 4 * e;
        if (row < p.M && col < p.Nstore) {
          const int b = row / p.L;
          const bool live = (row - b * p.L) < p.lens[b];
          float v = 0.f;
          if (live && cok) {
            v = (float)(acc[i][j][e] + bj);  // ONE rounding of the exact sum
            if (has_res) v += p.resid[(long)row * p.ldr + col];
          }
          p.C[(long)row * p.ldc + col] = v;
        }
      }
    }
  }
}

// per-column sum / sum of squares of the rows [r0, r0 + rows_per_block) of X -> part[block][2][C] (f64, one slot per
// workgroup: summed afterwards in a fixed order, reduce_parts)
__global__ void k_col_stats(const float* __restrict__ X, long ldx, long R, int C, long rows_per_block,
                            double* __restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
  double s1 = 0.0, s2 = 0.0;
  for (long r = r0; r < r1; ++r) {
    const double v = (double)X[r * ldx + c];
    s1 += v;
    s2 += v * v;
  }
  part[((long)blockIdx.y * 2 + 0) * C + c] = s1;
  part[((long)blockIdx.y * 2 + 1) * C + c] = s2;
}

}  // namespace pn
