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

// ------------------------------------------------------------------------------------------------------------------
// conv1 on ONE-HOT input (MaskedConv1D(20 -> 1100, k = 9), protein_encoders.py:84-91,110): with exactly one 1.0
// per residue the convolution is a gather-sum, y[p][c] = b[c] + sum_tap W[c][aa[p + tap - k/2]][tap] - 9 adds per output
// instead of 180 multiply-adds, bound by writing the [B*L, 1100] activation (4 400 B per residue).  The f32-MFMA chain
// adds the same terms in the same (tap) order and every other product is an exact 0, so the result is BIT-IDENTICAL to
// the general convolution.  k_onehot_ids recognises one-hot input (and raises *flag otherwise: soft / augmented inputs
// take the general kernel, which is launched behind this one with GemmParams::run_if = flag).
// ------------------------------------------------------------------------------------------------------------------
// ids[p] = index of the 1.0 of residue p, -1 for an all-zero column or a pad position; *flag |= 1 when a live residue is
// anything else.  onehots [B][Cin][L].
__global__ void k_onehot_ids(const float* __restrict__ x, const int* __restrict__ lens, signed char* __restrict__ ids,
                             int* __restrict__ flag, int B, int Cin, int L) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)B * L) return;
  const int b = (int)(p / L), t = (int)(p - (long)b * L);
  int id = -1;
  if (t < lens[b]) {
    const float* src = x + (long)b * Cin * L + t;
    int ones = 0;
    bool bad = false;
    for (int c = 0; c < Cin; ++c) {
      const float v = src[(long)c * L];
      if (v == 1.f) {
        id = c;
        ++ones;
      } else if (v != 0.f) {
        bad = true;
      }
    }
    if (bad || ones > 1) atomicOr(flag, 1);
  }
  ids[p] = (signed char)id;
}

// packed conv weight [C][ntap][ldi] -> Wt [ntap][Cin][ldc] (output channel fastest), zero pad lanes
__global__ void k_conv1_relay(const float* __restrict__ w, int C, int ntap, int Cin, int ldi, float* __restrict__ out,
                              int ldc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)ntap * Cin * ldc) return;
  const int c = (int)(i % ldc);
  const int a = (int)((i / ldc) % Cin);
  const int tap = (int)(i / ((long)ldc * Cin));
  out[i] = c < C ? w[((long)c * ntap + tap) * ldi + a] : 0.f;
}

// One workgroup (512 threads): `tiles` row tiles of BM residues x one 64-channel slice whose weights sit in LDS as
// [tap][1 + Cin][64] - row 0 of every tap is all zeros, so "no residue here" (outside the sequence, an all-zero column)
// is just another row and the inner loop has no select: it adds +0.  Thread (rl = tid / 16, cl = tid % 16): residues rl,
// rl + 32, ... of a tile, channels 4 cl .. 4 cl + 3 of the slice.  Per tile, each row's NTAP weight-row indices (one byte
// each) and its live flag are staged in LDS once (16 bytes per row: one ds_read_b128), so a tap costs a byte extract, an
// address add, a 16-byte LDS read and four adds.  col_part (train-mode BatchNorm statistics of the output, optional):
// [row tile][2][C] f32 partials, the layout and tile height of the general convolution's epilogue, so the same
// fixed-order reduction finishes them.
template <int NTAP>
__global__ __launch_bounds__(512) void k_conv1_gather(const signed char* __restrict__ ids, const int* __restrict__ lens,
                                                      const float* __restrict__ Wt, const float* __restrict__ bias,
                                                      float* __restrict__ out, int P, int L, int C, int ldc, int Cin,
                                                      const int* __restrict__ flag, float* __restrict__ col_part,
                                                      int BM, int tiles) {
  static_assert(NTAP <= 12, "row indices of one residue are packed into 12 bytes");
  if (*flag != 0) return;
  constexpr int CS = 64, RL = 32, NT = 512;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int R1 = Cin + 1;
  float* Ws = smem;                               // [NTAP * (1 + Cin)][CS]
  float* red = Ws + (size_t)NTAP * R1 * CS;       // [RL][2][CS]
  unsigned* ridx = (unsigned*)(red + RL * 2 * CS);  // [BM][4]: bytes 0 .. NTAP-1 = weight row of each tap, byte 15 = live
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  // (blockIdx.x = channel slice: the slices of one row tile run at the same time, so a 4.4 KB activation row is written
  //  by its 18 workgroups together instead of in 18 visits far apart)
  const int c0 = blockIdx.x * CS, col = c0 + 4 * cl;
  const bool cin_range = col < ldc;
  for (int i = tid; i < NTAP * R1 * (CS / 4); i += NT) {
    const int row = i / (CS / 4), q = i - row * (CS / 4);
    const int tap = row / R1, a1 = row - tap * R1;  // a1 = 0: the zero row
    const int cc = c0 + 4 * q;
    const float4 v = (a1 > 0 && cc < ldc) ? ld4(Wt + ((long)tap * Cin + (a1 - 1)) * ldc + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(Ws + row * CS + 4 * q) = v;
  }
  float4 bj = make_float4(0.f, 0.f, 0.f, 0.f);
  if (cin_range && bias != nullptr) {
    bj.x = col + 0 < C ? bias[col + 0] : 0.f;
    bj.y = col + 1 < C ? bias[col + 1] : 0.f;
    bj.z = col + 2 < C ? bias[col + 2] : 0.f;
    bj.w = col + 3 < C ? bias[col + 3] : 0.f;
  }
  const bool c0ok = col + 0 < C, c1ok = col + 1 < C, c2ok = col + 2 < C, c3ok = col + 3 < C;
  constexpr int half = NTAP / 2;
  const int ntile = (P + BM - 1) / BM;
  const unsigned ws_lane = lds_addr(Ws) + 16u * cl;
  for (int tl = 0; tl < tiles; ++tl) {
    const int tile = blockIdx.y * tiles + tl;
    if (tile >= ntile) break;
    const int row0 = tile * BM;
    __syncthreads();
    for (int k = tid; k < BM; k += NT) {
      const int p = row0 + k;
      unsigned w4[4] = {0u, 0u, 0u, 0u};
      if (p < P) {
        const int b = p / L, t = p - b * L;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
          const int tt = t + tap - half;
          const int id = ((unsigned)tt < (unsigned)L) ? (int)ids[p + tap - half] : -1;  // (pads carry -1)
          w4[tap >> 2] |= (unsigned)(tap * R1 + id + 1) << (8 * (tap & 3));
        }
        if (t < lens[b]) w4[3] |= 0x80000000u;
      } else {
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) w4[tap >> 2] |= (unsigned)(tap * R1) << (8 * (tap & 3));
      }
      *reinterpret_cast<uint4*>(ridx + 4 * k) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    __syncthreads();
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (int r = rl; r < BM; r += RL) {
      if (row0 + r >= P) break;
      const uint4 rx = *reinterpret_cast<const uint4*>(ridx + 4 * r);
      const unsigned w4[4] = {rx.x, rx.y, rx.z, rx.w};
      const bool live = (rx.w >> 31) != 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int tap = 0; tap < NTAP; ++tap) {
        const unsigned row = (w4[tap >> 2] >> (8 * (tap & 3))) & (tap == 15 ? 0x7fu : 0xffu);
        const float4 wv = lds_read4(ws_lane + row * (CS * 4u));
        // the terms of the k-ordered fmaf chain in its order (acc = fma(1, w, acc)); the zero row adds +0
        acc.x += wv.x; acc.y += wv.y; acc.z += wv.z; acc.w += wv.w;
      }
      float4 v;  // (acc + bias) on live rows, 0 elsewhere and in the pad lanes (E_CONV)
      v.x = (live && c0ok) ? acc.x + bj.x : 0.f;
      v.y = (live && c1ok) ? acc.y + bj.y : 0.f;
      v.z = (live && c2ok) ? acc.z + bj.z : 0.f;
      v.w = (live && c3ok) ? acc.w + bj.w : 0.f;
      if (cin_range) *reinterpret_cast<float4*>(out + (long)(row0 + r) * ldc + col) = v;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
    }
    if (col_part != nullptr) {
      *reinterpret_cast<float4*>(red + (rl * 2 + 0) * CS + 4 * cl) = s1;
      *reinterpret_cast<float4*>(red + (rl * 2 + 1) * CS + 4 * cl) = s2;
      __syncthreads();
      if (tid < 2 * CS) {
        const int which = tid / CS, c = tid - which * CS;
        float a = 0.f;
        for (int w = 0; w < RL; ++w) a += red[(w * 2 + which) * CS + c];
        if (c0 + c < C) col_part[((long)tile * 2 + which) * C + c0 + c] = a;
      }
    }
  }
}

}  // namespace pn
