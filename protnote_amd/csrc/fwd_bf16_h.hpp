// AMP-class forward (pn_set_forward_math(1)) with the activation operand MATERIALISED as bf16, the forward-side twin of
// bwd_bf16_dz.hpp.  What bounded the single-product kernels that round their A operand while staging was the L2 -> L1 path and
// the vector work per slab (docs/experiments.md 4.6a: an f32 operand is 48 KB per 256 x 256 x 32 slab for 1024 matrix-pipe
// cycles); the backward got past it by storing dz as bf16 and taking BOTH operands by LDS-DMA (gemm_nt_bf16dma_kernel,
// 1.1 PFLOP/s against 0.85-0.95 for the register-staged kinds).  The forward does the same per chunk of pair rows:
//   h_{l-1}[r][:] = relu(A'[r % B] + B'[r / B])     (layer 2: from the two folded layer-1 tables, L2-resident)
//                 = relu(s * z_{l-1}[r] + t)         (deeper layers: one read of the stored f32 pre-activation)
// is written ONCE as bf16 (k_make_h_bf16: 2 B per element, a chunk at a time into a workspace buffer - nothing of pair-grid
// size is allocated), then z_l = h_{l-1} W_l^T runs on gemm_nt_bf16m16_kernel<EK> (gemm_bf16_m16.hpp; pn_set_bf16_mfma16(0): gemm_nt_bf16dma_kernel<EK>,
// the same accumulators) with the usual epilogues (E_STORE + BatchNorm
// column partials in training; E_STORE_H16 = the NEXT layer's relu(bn(.)) written straight as bf16, and E_ROWDOT, in eval).
// Same bf16 values in the same products as the staging-time rounding; k is paired in natural order inside a 16-k MFMA step
// (the register-staged kernels pair {4g..4g+3, 16+4g..}), so the two routes agree to f32 summation order.
#pragma once
#include "bwd_bf16_dz.hpp"

namespace pn {

struct MakeHParams {
  long r0, rows;  // pair rows [r0, r0 + rows) of the grid -> out rows [0, rows)
  int C;          // hidden width (C % 8 == 0)
  int pairB;      // KIND 0: rows r = j * pairB + i
  const float* A;  // KIND 0: A' [pairB][lda];  KIND 1 / 2: z [R][lda] (row r0 + k)
  long lda;
  const float* A2;  // KIND 0: B' [labels][lda2]
  long lda2;
  const float* s;  // KIND 1: per-column scale / shift of the BatchNorm fold
  const float* t;
  uint16_t* out;  // [rows][C] bf16, dense
};

// KIND 0: relu(A'[i] + B'[j]);  1: relu(s z + t);  2: round(z) (an activation that is already relu(bn(.)))
// grid.x = row blocks, block = C / 8 threads (<= 1024): thread c8 owns columns 8 c8 .. + 7 of every row of its block.
template <int KIND>
__global__ __launch_bounds__(1024) void k_make_h_bf16(const MakeHParams P, int rows_per_block) {
  const int c = threadIdx.x * 8;
  if (c >= P.C) return;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, t0 = s0, t1 = s0;
  if constexpr (KIND == 1) {
    s0 = ld4(P.s + c); s1 = ld4(P.s + c + 4);
    t0 = ld4(P.t + c); t1 = ld4(P.t + c + 4);
  }
  const long k0 = (long)blockIdx.x * rows_per_block;
  long k1 = k0 + rows_per_block;
  if (k1 > P.rows) k1 = P.rows;
  for (long k = k0; k < k1; ++k) {
    const long r = P.r0 + k;
    float4 v0, v1;
    if constexpr (KIND == 0) {
      const long j = r / P.pairB;
      const long i = r - j * P.pairB;
      const float* a = P.A + i * P.lda + c;
      const float* b = P.A2 + j * P.lda2 + c;
      const float4 a0 = ld4(a), a1 = ld4(a + 4), b0 = ld4(b), b1 = ld4(b + 4);
      v0 = make_float4(relu(a0.x + b0.x), relu(a0.y + b0.y), relu(a0.z + b0.z), relu(a0.w + b0.w));
      v1 = make_float4(relu(a1.x + b1.x), relu(a1.y + b1.y), relu(a1.z + b1.z), relu(a1.w + b1.w));
    } else {
      const float* z = P.A + r * P.lda + c;
      v0 = ld4(z);
      v1 = ld4(z + 4);
      if constexpr (KIND == 1) {
        v0 = make_float4(relu(fmaf(v0.x, s0.x, t0.x)), relu(fmaf(v0.y, s0.y, t0.y)), relu(fmaf(v0.z, s0.z, t0.z)),
                         relu(fmaf(v0.w, s0.w, t0.w)));
        v1 = make_float4(relu(fmaf(v1.x, s1.x, t1.x)), relu(fmaf(v1.y, s1.y, t1.y)), relu(fmaf(v1.z, s1.z, t1.z)),
                         relu(fmaf(v1.w, s1.w, t1.w)));
      }
    }
    *reinterpret_cast<bf16x8*>(P.out + k * P.C + c) = round8(v0, v1);
  }
}

}  // namespace pn
