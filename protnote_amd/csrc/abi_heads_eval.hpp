// Part of the translation unit protnote_hip.hip (#included there, after the launchers and small kernels; not a
// stand-alone header: it uses the static helpers defined above its #include):
// row MLPs and pair head in eval mode, label noise, ensembling, similarity head, generic GEMM entry, TN launchers.
// ------------------------------------------------------------------------------------------------
// row MLP (W_p / W_l), eval
// ------------------------------------------------------------------------------------------------
extern "C" size_t pn_mlp_rows_ws_bytes(const pn_mlp* m, int rows) {
  size_t b = 0;
  int hmax = 0;
  for (int i = 0; i + 1 < m->nlayers; ++i) hmax = m->dims[i + 1] > hmax ? m->dims[i + 1] : hmax;
  b += 2 * al256((size_t)rows * hmax * sizeof(float));  // ping-pong hidden activations
  b += 2 * al256((size_t)hmax * sizeof(float));         // s, t
  return b;
}

extern "C" int pn_mlp_rows_fwd_eval(const pn_mlp* m, const float* x, int ldx, int rows, float* y, void* ws,
                                    size_t ws_bytes, void* stream) {
  PN_OK(math_field_check(m->math_mode, "pn_mlp_rows_fwd_eval"));
  MathScope math_scope(m->math_mode);
  hipStream_t st = (hipStream_t)stream;
  if (m->nlayers < 1 || m->nlayers > PN_MAX_LAYERS) return fail("mlp: bad layer count %d", m->nlayers);
  for (int i = 0; i <= m->nlayers; ++i)
    if (i < m->nlayers && m->dims[i] % 4 != 0) return fail("mlp: dims[%d]=%d not a multiple of 4", i, m->dims[i]);
  if (ldx % 4 != 0) return fail("mlp: ldx %% 4 != 0");
  int hmax = 0;
  for (int i = 0; i + 1 < m->nlayers; ++i) hmax = m->dims[i + 1] > hmax ? m->dims[i + 1] : hmax;
  Bump bp(ws, ws_bytes);
  float* buf[2];
  buf[0] = bp.take<float>((size_t)rows * hmax);
  buf[1] = bp.take<float>((size_t)rows * hmax);
  float* s = bp.take<float>(hmax);
  float* t = bp.take<float>(hmax);
  if (!bp.ok) return fail("mlp: workspace too small");
  const float* in = x;
  long ldin = ldx;
  for (int i = 0; i < m->nlayers; ++i) {
    const bool last = (i + 1 == m->nlayers);
    GemmParams p = gp_zero();
    p.M = rows;
    p.N = m->dims[i + 1];
    p.Nstore = p.N;
    p.Kseg = m->dims[i];
    p.A = in;
    p.lda = ldin;
    p.W = m->w[i];
    p.ldw = m->dims[i];
    float* out = last ? y : buf[i & 1];
    p.C = out;
    p.ldc = p.N;
    // Linear bias: with a BN behind it the bias is applied by the fold; for the last layer add directly
    p.bias = last ? m->bias[i] : nullptr;
    if (i == 0) {
      PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, pick_variant(p.N), st)));
    } else {
      p.a_scale = s;
      p.a_shift = t;
      PN_OK((launch_gemm<A_AFFINE_RELU, E_STORE>(p, pick_variant(p.N), st)));
    }
    if (!last) {
      // fold BN_i (or identity + bias) for the next layer's operand load
      if (m->bn[i].weight != nullptr && m->bias[i] != nullptr)
        return fail("mlp: Linear bias together with BatchNorm is not supported");
      hipLaunchKernelGGL(k_bn_fold_eval, dim3(nblk(p.N, 256)), dim3(256), 0, st, m->bn[i], m->bias[i], m->bn_eps,
                         p.N, p.N, s, t);
      HIP_OK(hipGetLastError());
    }
    in = out;
    ldin = p.N;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// pair head, eval
// ------------------------------------------------------------------------------------------------
struct PairWs {
  float *A1, *B1, *weff, *z[2], *partials, *s[PN_MAX_LAYERS], *t[PN_MAX_LAYERS];
  uint16_t* wsplit;  // bf16x3 mode: hi / lo planes of the layer's weight (h x h x 4 bytes)
  uint16_t* hbuf[2];  // forward_math = bf16: the chunk's activation operand materialised as bf16 (fwd_bf16_h.hpp), ping-pong
  int nparts;
};

static bool pair_carve(const pn_pairhead* hd, int B, int NL, int chunk, Bump& bp, PairWs& w) {
  const int h = hd->h;
  const long crow = (long)chunk * B;
  w.A1 = bp.take<float>((size_t)B * h);
  w.B1 = bp.take<float>((size_t)NL * h);
  w.weff = hd->fusion == 1 ? bp.take<float>((size_t)h * 2 * hd->d) : nullptr;
  // stored chunk activations: layers 2..n-1 (ping-pong), plus z1 itself for concatenation_prod
  const int nstored = (hd->nlayers > 2 ? hd->nlayers - 2 : 0) + (hd->fusion == 2 ? 1 : 0);
  const int nz = nstored >= 2 ? 2 : nstored;
  w.z[0] = nz >= 1 ? bp.take<float>((size_t)crow * h) : nullptr;
  w.z[1] = nz >= 2 ? bp.take<float>((size_t)crow * h) : nullptr;
  w.nparts = rowdot_nparts(h);
  w.partials = bp.take<float>((size_t)w.nparts * crow);
  for (int i = 0; i < hd->nlayers; ++i) {
    w.s[i] = bp.take<float>(h);
    w.t[i] = bp.take<float>(h);
  }
  w.wsplit = (uint16_t*)bp.take<float>((size_t)h * h);
  // (carved LAST and by the descriptor alone: the fields above sit where they always sat)
  w.hbuf[0] = w.hbuf[1] = nullptr;
  if (fwd_bf16_requested(hd) && fwd_staged_shape(h) && hd->nlayers > 1) {
    w.hbuf[0] = (uint16_t*)bp.take<float>((size_t)crow * h / 2);
    if (hd->nlayers > 2) w.hbuf[1] = (uint16_t*)bp.take<float>((size_t)crow * h / 2);
  }
  return bp.ok;
}

static int clamp_chunk(int chunk, int NL) {
  if (chunk <= 0 || chunk > NL) chunk = NL;
  return chunk;
}

extern "C" size_t pn_pairhead_eval_ws_bytes(const pn_pairhead* hd, int B, int NL, int label_chunk) {
  Bump bp(nullptr, (size_t)-1);
  PairWs w;
  pair_carve(hd, B, NL, clamp_chunk(label_chunk, NL), bp, w);
  return bp.off;
}

extern "C" int pn_pairhead_fwd_eval(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                                    float* logits_pairs, int label_chunk, void* ws, size_t ws_bytes,
                                    void* stream) {
  PN_OK(math_field_check(hd->math_mode, "pn_pairhead_fwd_eval"));
  MathScope math_scope(hd->math_mode);
  bool fwd_bf16 = false;
  PN_OK(fwd_math_of(hd, &fwd_bf16));
  hipStream_t st = (hipStream_t)stream;
  const int h = hd->h, d = hd->d;
  if (hd->nlayers < 1 || hd->nlayers > PN_MAX_LAYERS) return fail("pairhead: nlayers=%d unsupported (need 1..%d)", hd->nlayers, PN_MAX_LAYERS);
  if (hd->fusion < 0 || hd->fusion > 2) return fail("pairhead: fusion %d not implemented", hd->fusion);
  if (d % 4 || h % 4) return fail("pairhead: d and h must be multiples of 4");
  const int chunk = clamp_chunk(label_chunk, NL);
  if ((long)chunk * B > 0x7fffffffL) return fail("pairhead: chunk too large");
  Bump bp(ws, ws_bytes);
  PairWs w;
  if (!pair_carve(hd, B, NL, chunk, bp, w)) return fail("pairhead: workspace too small");

  // layer 1, separable: A1 = P_e W1a^T, B1 = L_e W1b^T
  const float* w1 = hd->w[0];
  long ldw1 = hd->in_dim;
  if (hd->fusion == 1) {
    hipLaunchKernelGGL(k_diff_weight, dim3(nblk((long)h * 2 * d, 256)), dim3(256), 0, st, hd->w[0], w.weff, h, d);
    HIP_OK(hipGetLastError());
    w1 = w.weff;
    ldw1 = 2 * d;
  }
  {
    GemmParams p = gp_zero();
    p.M = B; p.N = h; p.Nstore = h; p.Kseg = d;
    p.A = P_e; p.lda = d; p.W = w1; p.ldw = ldw1; p.C = w.A1; p.ldc = h;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
    p.M = NL; p.A = L_e; p.W = w1 + d; p.C = w.B1;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
  }
  if (hd->fusion == 2 && hd->nlayers - 1 > 2 && w.z[1] == nullptr) return fail("pairhead: internal z buffers");
  for (int i = 0; i < hd->nlayers; ++i) {
    if (hd->bn[i].weight != nullptr && hd->bias[i] != nullptr)
      return fail("pairhead: Linear bias together with BatchNorm is not supported");
    hipLaunchKernelGGL(k_bn_fold_eval, dim3(nblk(h, 256)), dim3(256), 0, st, hd->bn[i], hd->bias[i], hd->bn_eps, h,
                       h, w.s[i], w.t[i]);
  }
  const bool prod = hd->fusion == 2;
  if (!prod) {
    // A' = s1*A1 + t1, B' = s1*B1  =>  h1[i,j] = relu(A'[i] + B'[j])
    hipLaunchKernelGGL(k_affine_rows, dim3(nblk((long)B * h, 256)), dim3(256), 0, st, w.A1, (long)h, w.A1, (long)h,
                       (long)B, h, w.s[0], w.t[0]);
    hipLaunchKernelGGL(k_affine_rows, dim3(nblk((long)NL * h, 256)), dim3(256), 0, st, w.B1, (long)h, w.B1, (long)h,
                       (long)NL, h, w.s[0], (const float*)nullptr);
    HIP_OK(hipGetLastError());
  }

  if (hd->nlayers == 1 && !prod) {
    // OUTPUT_MLP_NUM_LAYERS: 1 (get_mlp, ProtNote.py:337-378: one hidden layer + the output neuron).  The hidden layer is
    // the separable one, so there is no pair-grid GEMM at all: logit[i,j] = w_out . relu(A'[i] + B'[j]) + b_out in one pass
    ProfScope ps(ST_PAIR1_FWD, 3.0 * (double)B * (double)NL * (double)h, st);
    hipLaunchKernelGGL(k_pairsum_rowdot, dim3(nblk(B, 64), nblk(NL, 64)), dim3(256), 0, st, (const float*)w.A1, (long)h,
                       (const float*)w.B1, (long)h, B, NL, h, hd->w_out, hd->b_out, logits_pairs);
    HIP_OK(hipGetLastError());
    return 0;
  }
  for (int j0 = 0; j0 < NL; j0 += chunk) {
    const int nj = (NL - j0 < chunk) ? NL - j0 : chunk;
    const long rows = (long)nj * B;
    const float* in = nullptr;
    bool in_act = false;  // `in` holds post-activation values (its producer applied BN + ReLU)
    int zsel = 0;
    if (prod) {
      // concatenation_prod: z1 = A1[i] + B1[j] + (P_e[i] (.) L_e[j]) W1c^T  (not separable: one more pair GEMM)
      GemmParams p = gp_zero();
      p.M = (int)rows; p.N = h; p.Nstore = h; p.Kseg = d;
      p.A = P_e; p.lda = d; p.A2 = L_e + (long)j0 * d; p.lda2 = d; p.pairB = B;
      p.W = hd->w[0] + 2 * d; p.ldw = hd->in_dim;
      p.padd1 = w.A1; p.ldp1 = h; p.padd2 = w.B1 + (long)j0 * h; p.ldp2 = h;
      p.C = w.z[zsel]; p.ldc = h;
      PN_OK((launch_gemm<A_PAIRPROD, E_PAIRADD>(p, 0, st)));
      in = w.z[zsel];
      zsel ^= 1;
    }
    if (fwd_staged_on(fwd_bf16, h) && w.hbuf[0] != nullptr && hd->nlayers > 1) {  // (one hidden layer: no hidden pair-grid GEMM)
      // AMP-class forward, materialised operand (fwd_bf16_h.hpp): h_{li-1} of this chunk as bf16 -> all-DMA GEMM; a hidden
      // layer's epilogue writes the next operand directly (E_STORE_H16: relu(bn(z)) rounded once), the last one the row-dot
      int hsel = 0;
      bool have_h = false;
      for (int li = 1; li < hd->nlayers; ++li) {
        const bool last = (li + 1 == hd->nlayers);
        if (li == 1 && !prod)
          PN_OK(make_h(0, (long)j0 * B, rows, h, w.A1, h, w.B1, h, B, nullptr, nullptr, w.hbuf[hsel], st));
        else if (!have_h)  // concatenation_prod: the stored raw z1 of this chunk through its fold
          PN_OK(make_h(1, 0, rows, h, in, h, nullptr, 0, 1, w.s[li - 1], w.t[li - 1], w.hbuf[hsel], st));
        hipLaunchKernelGGL(k_round_plane, dim3(nblk((long)h * h / 4, 256)), dim3(256), 0, st, hd->w[li], (long)h, h, h, w.wsplit);
        GemmParams p = gp_zero();
        p.M = (int)rows; p.N = h; p.Nstore = h; p.Kseg = h;
        p.A = (const float*)w.hbuf[hsel]; p.lda = h / 2; p.w_hi = w.wsplit;
        p.e_scale = w.s[li]; p.e_shift = w.t[li];
        const int src = (li == 1 && !prod) ? 2 : (have_h ? 0 : 1);
        if (last) {
          p.e_w = hd->w_out; p.rowdot_out = w.partials;
          PN_OK((launch_gemm_h16<E_ROWDOT>(p, src, st)));
        } else {
          if (w.hbuf[hsel ^ 1] == nullptr) return fail("pairhead: internal h buffers");
          p.C = (float*)w.hbuf[hsel ^ 1]; p.ldc = h;
          PN_OK((launch_gemm_h16<E_STORE_H16>(p, src, st)));
          hsel ^= 1;
          have_h = true;
        }
      }
      hipLaunchKernelGGL(k_rowdot_reduce, dim3(nblk(rows, 256)), dim3(256), 0, st, w.partials, w.nparts, rows, hd->b_out,
                         logits_pairs + (long)j0 * B);
      HIP_OK(hipGetLastError());
      continue;
    }
    FwdBf16Scope fwd_scope(fwd_bf16);  // the hidden layers' pair-grid GEMMs below (the layer-1 GEMM of _prod above is not one)
    for (int li = 1; li < hd->nlayers; ++li) {
      const bool last = (li + 1 == hd->nlayers);
      const bool from_pairs = (li == 1) && !prod;
      GemmParams p = gp_zero();
      p.M = (int)rows; p.N = h; p.Nstore = h; p.Kseg = h;
      p.W = hd->w[li]; p.ldw = h; p.wsplit = w.wsplit;
      // Eval mode knows every BatchNorm fold up front, so a stored layer's BN + ReLU is applied by its PRODUCER (E_STORE
      // with e_scale / e_shift): the consumer then reads a plain operand, which the 256-tile kernels stage by LDS-DMA
      // (an all-DMA slab loop instead of a register-staged, generated A operand).  Same fmaf + max on the same values
      // as the operand-side fold: bit-identical logits.
      const bool in_is_act = in_act;
      if (from_pairs) {
        p.A = w.A1; p.lda = h; p.A2 = w.B1 + (long)j0 * h; p.lda2 = h; p.pairB = B;
      } else {
        p.A = in; p.lda = h;
        if (!in_is_act) { p.a_scale = w.s[li - 1]; p.a_shift = w.t[li - 1]; }
      }
      if (last) {
        p.e_scale = w.s[li]; p.e_shift = w.t[li]; p.e_w = hd->w_out; p.rowdot_out = w.partials;
        if (from_pairs) PN_OK((launch_gemm<A_PAIRSUM_RELU, E_ROWDOT>(p, 0, st)));
        else if (in_is_act) PN_OK((launch_gemm<A_PLAIN, E_ROWDOT>(p, 0, st)));
        else PN_OK((launch_gemm<A_AFFINE_RELU, E_ROWDOT>(p, 0, st)));
      } else {
        float* out = w.z[zsel];
        zsel ^= 1;
        p.C = out; p.ldc = h;
        p.e_scale = w.s[li]; p.e_shift = w.t[li];  // store relu(bn(z_li)) instead of z_li
        if (from_pairs) PN_OK((launch_gemm<A_PAIRSUM_RELU, E_STORE>(p, 0, st)));
        else if (in_is_act) PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
        else PN_OK((launch_gemm<A_AFFINE_RELU, E_STORE>(p, 0, st)));
        in = out;
        in_act = true;
      }
    }
    if (hd->nlayers == 1) {  // concatenation_prod with one hidden layer: the stored z1 of this chunk -> logits
      hipLaunchKernelGGL(k_rowdot_rows, dim3(nblk(rows, 4)), dim3(256), 0, st, in, (long)h, rows, h, (const float*)w.s[0],
                         (const float*)w.t[0], hd->w_out, hd->b_out, logits_pairs + (long)j0 * B);
      HIP_OK(hipGetLastError());
      continue;
    }
    hipLaunchKernelGGL(k_rowdot_reduce, dim3(nblk(rows, 256)), dim3(256), 0, st, w.partials, w.nparts, rows,
                       hd->b_out, logits_pairs + (long)j0 * B);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

__global__ void k_label_noise(const float* __restrict__ x, const float* __restrict__ u, float scale,
                              float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] + (2.f * u[i] - 1.f) * scale;
}

extern "C" int pn_label_noise(const float* L_f, const float* u, float scale, float* out, long n, void* stream) {
  hipLaunchKernelGGL(k_label_noise, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, L_f, u, scale, out, n);
  HIP_OK(hipGetLastError());
  return 0;
}

// The same noise with u drawn INSIDE the kernel: a counter hash of (seed, row, column) like the dropout masks
// (gemm_engine.hpp: drop_rowkey / pn_lowbias32), top 24 bits -> u in [0, 1) on the float grid torch's own uniform uses.  No
// [rows][cols] tensor of uniforms is written and read back (131 MB each way at the bench size); pn_uniform hands a test the
// very same draw.  LABEL_NOISE_STREAM keeps the sequence apart from the dropout streams of the same seed.
enum { LABEL_NOISE_STREAM = 400 };
__device__ __forceinline__ float noise_uniform(uint32_t rowkey, uint32_t col) {
  return (float)(pn_lowbias32(rowkey + col * 0x9E3779B1U) >> 8) * (1.f / 16777216.f);
}
__global__ void k_label_noise_seeded(const float* __restrict__ x, uint32_t seed, float scale, float* __restrict__ out, long rows,
                                     int cols, int just_u) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const uint32_t c = (uint32_t)(i - r * cols);
  const float u = noise_uniform(drop_rowkey(seed, (uint32_t)r), c);
  out[i] = just_u ? u : x[i] + (2.f * u - 1.f) * scale;
}
static uint32_t noise_seed(unsigned seed) { return seed ^ ((uint32_t)LABEL_NOISE_STREAM * 0x9E3779B9u); }

extern "C" int pn_label_noise_seeded(const float* L_f, unsigned seed, float scale, float* out, long rows, int cols,
                                     void* stream) {
  if (rows < 0 || cols <= 0 || rows > 0xffffffffL) return fail("label_noise: bad shape %ld x %d", rows, cols);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_label_noise_seeded, dim3(nblk(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, L_f, noise_seed(seed),
                     scale, out, rows, cols, 0);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_uniform(unsigned seed, long rows, int cols, float* out, void* stream) {
  if (rows < 0 || cols <= 0 || rows > 0xffffffffL) return fail("uniform: bad shape %ld x %d", rows, cols);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_label_noise_seeded, dim3(nblk(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                     noise_seed(seed), 0.f, out, rows, cols, 1);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_ensemble_logit(const float* logits_pairs, int B, int NL, int ndesc, int protein_major, float* out,
                                 void* stream) {
  if (ndesc < 1 || NL % ndesc != 0) return fail("ensemble: NL=%d not divisible by ndesc=%d", NL, ndesc);
  const long n = (long)B * (NL / ndesc);
  hipLaunchKernelGGL(k_ensemble, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, logits_pairs, B, NL, ndesc,
                     protein_major, out);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_ensemble_logit_bwd(const float* logits, const float* dout, int B, int NL, int ndesc, float* dlogits,
                                     void* stream) {
  if (ndesc < 1 || NL % ndesc != 0) return fail("ensemble bwd: NL=%d not divisible by ndesc=%d", NL, ndesc);
  hipLaunchKernelGGL(k_ensemble_bwd, dim3(nblk((long)B * (NL / ndesc), 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     dout, B, NL, ndesc, dlogits);
  HIP_OK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// similarity head
// ------------------------------------------------------------------------------------------------
extern "C" size_t pn_similarity_ws_bytes(int B, int NL) {
  return al256((size_t)B * sizeof(float)) + al256((size_t)NL * sizeof(float));
}

extern "C" int pn_similarity_fwd(const float* P_e, const float* L_e, int B, int NL, int d, float temperature,
                                 float* logits, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (d % 4) return fail("similarity: d %% 4 != 0");
  Bump bp(ws, ws_bytes);
  float* rs = bp.take<float>(B);
  float* cs = bp.take<float>(NL);
  if (!bp.ok) return fail("similarity: workspace too small");
  hipLaunchKernelGGL(k_rownorm_inv, dim3(nblk(B, 4)), dim3(256), 0, st, P_e, (long)d, B, d, rs);
  hipLaunchKernelGGL(k_rownorm_inv, dim3(nblk(NL, 4)), dim3(256), 0, st, L_e, (long)d, NL, d, cs);
  HIP_OK(hipGetLastError());
  GemmParams p = gp_zero();
  p.M = B; p.N = NL; p.Nstore = NL; p.Kseg = d;
  p.A = P_e; p.lda = d; p.W = L_e; p.ldw = d; p.C = logits; p.ldc = NL;
  p.row_scale = rs; p.col_scale = cs; p.alpha = 1.f / temperature;
  return launch_gemm<A_PLAIN, E_SCALE_RC>(p, 0, st);
}

// ------------------------------------------------------------------------------------------------
// generic GEMM entry (tests / building block)
// ------------------------------------------------------------------------------------------------
extern "C" size_t pn_gemm_nt_stats_ws_bytes(int M, int N) {
  Bump bp(nullptr, (size_t)-1);
  ColScr c;
  colscr_carve(bp, M, N, c);
  return bp.off + 256;
}

extern "C" int pn_gemm_nt(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N,
                          int K, const float* bias, const float* a_scale, const float* a_shift, double* col_sum,
                          double* col_sumsq, int tile_variant, void* ws, size_t ws_bytes, void* stream) {
  if (K % 4 || lda % 4 || ldw % 4) return fail("gemm_nt: K, lda, ldw must be multiples of 4");
  GemmParams p = gp_zero();
  p.M = M; p.N = N; p.Nstore = N; p.Kseg = K;
  p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.C = C; p.ldc = ldc; p.bias = bias;
  if (col_sum != nullptr) {
    if (col_sumsq == nullptr) return fail("gemm_nt: col_sum without col_sumsq");
    Bump bp(ws, ws_bytes);
    ColScr c;
    if (ws == nullptr || !colscr_carve(bp, M, N, c)) return fail("gemm_nt: column statistics need pn_gemm_nt_stats_ws_bytes(M, N) of workspace");
    p.col_sum = col_sum; p.col_sumsq = col_sumsq; p.col_part = c.part; p.col_red = c.red;
  }
  const int v = tile_variant < 0 ? pick_variant(N) : tile_variant;
  if (a_scale) {
    p.a_scale = a_scale; p.a_shift = a_shift;
    return launch_gemm<A_AFFINE_RELU, E_STORE>(p, v, (hipStream_t)stream);
  }
  return launch_gemm<A_PLAIN, E_STORE>(p, v, (hipStream_t)stream);
}

// ================================================================================================
//                                      TRAINING PATH
// ================================================================================================
static int transpose_into(const float* src, long lds_, int rows, int cols, float* dst, long ldd, hipStream_t st) {
  hipLaunchKernelGGL(k_transpose, dim3(nblk(cols, 32), nblk(rows, 32)), dim3(256), 0, st, src, lds_, rows, cols, dst,
                     ldd);
  HIP_OK(hipGetLastError());
  return 0;
}

// choose the row split of a TN contraction: enough workgroups to fill the chip several times over,
// bounded by the partial-tile scratch the caller provided.
static int tn_pick_split(long R, int M, int N, size_t part_cap_floats, int tile, int resident_per_cu) {
  const long tiles = (long)((M + tile - 1) / tile) * ((N + tile - 1) / tile);
  const long slabs = (R + 31) / 32;
  const long target = 9L * 256 * resident_per_cu;  // ~9 full waves of resident workgroups
  long ns = (target + tiles - 1) / tiles;
  if (ns > slabs / 8) ns = slabs / 8;
  if (ns < 1) ns = 1;
  const long cap = (long)(part_cap_floats / ((size_t)M * N));
  if (ns > cap) ns = cap;
  if (ns < 1) ns = 1;
  return (int)ns;
}


template <int TA, int TB, bool BIG, bool ADMA = false, bool DROP = false>
static int launch_tn_cfg(TnParams p, float* dst, long ldd, float* part, size_t part_cap_floats, hipStream_t st) {
  auto kern = gemm_tn_kernel<TA, TB, BIG, ADMA, DROP>;
  constexpr int TILE = BIG ? 256 : 128;
  constexpr int LDS = BIG ? TN_LDS_BYTES_BIG : TN_LDS_BYTES;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done[dev] = true;
  }
  int ns = tn_pick_split(p.R, p.M, p.N, part_cap_floats, TILE, BIG ? 1 : 2);
  if (ns == 1) {
    p.Cpart = dst;
    p.ldc = ldd;
    p.rows_per_split = (p.R + 31) / 32 * 32;
  } else {
    if (part == nullptr) return fail("gemm_tn: no partial buffer");
    p.Cpart = part;
    p.ldc = p.N;
    long rps = (p.R + ns - 1) / ns;
    p.rows_per_split = (rps + 31) / 32 * 32;
    ns = (int)((p.R + p.rows_per_split - 1) / p.rows_per_split);
  }
  const unsigned tiles = (unsigned)(((p.M + TILE - 1) / TILE) * ((p.N + TILE - 1) / TILE));
  dim3 grid(tiles, (unsigned)ns);
  p.task_ns = 0;
  if (BIG && p.M == 3072 && p.N == 3072 && ns >= 2) {  // 12 x 12 tiles: 32-workgroup region tasks
    p.task_ns = ns;
    grid = dim3(tn_task_grid(ns), 1);
  }
  {
    ProfScope ps(100 + TA * 10 + TB, 2.0 * (double)p.R * (double)p.M * (double)p.N, st);
    hipLaunchKernelGGL(kern, grid, dim3(BIG ? 512 : 256), LDS, st, p);
  }
  HIP_OK(hipGetLastError());
  if (ns > 1) {
    hipLaunchKernelGGL(k_splitk_reduce, dim3(nblk((long)p.M * p.N, 256)), dim3(256), 0, st, (const float*)part, ns,
                       p.M, p.N, (long)p.N, dst, ldd);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

// the specialised f32 kernel of the big weight gradients (gemm_tn_fast.hpp); preconditions checked by launch_tn
static const int TN_SYNC_INTS = 4096;  // arrival counters of the paced TN kernel: [splits][4 regions][4 rotating]

// A contraction whose row count is not a multiple of 32 (a ragged last batch: B = 100 x 32 102 labels ...) runs its first
// R - R % 32 rows here and the last R % 32 rows as one more split-K partial on the small generic kernel (row_base), summed
// by the same fixed-order reduce.
// (Tried in round 4 and removed: the pair-sum operand for batch sizes that are not multiples of 32 with a scalar per-row
//  pair decode - 4 more B' row registers per thread push the slab loop into scratch spills, 129 instead of the generic
//  kernel's 133 TFLOP/s at B = 100 / 250; profiles/r04_shape_sweep.json.)
template <int TB>
static int launch_tn_fast(TnParams p, float* dst, long ldd, float* part, size_t part_cap_floats, hipStream_t st) {
  // pacing only for the kind whose second operand streams from HBM too (the pair-sum kind's tables are L2-resident: 0.22 TB)
  constexpr bool SYNC = TB == TB_AFFINE_RELU;
  auto kern = gemm_tn_fast_kernel<TB, SYNC>;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TN_FAST_LDS_BYTES));
    attr_done[dev] = true;
  }
  const long R_all = p.R, tail = p.R % 32;
  if (tail != 0) {
    if (part == nullptr || part_cap_floats < 2 * (size_t)p.M * p.N) return fail("gemm_tn: no partial buffer for the row tail");
    p.R = R_all - tail;
    part_cap_floats -= (size_t)p.M * p.N;  // the tail's slot
  }
  int ns = tn_pick_split(p.R, p.M, p.N, part_cap_floats, 256, 1);
  if (ns == 1 && tail == 0) {
    p.Cpart = dst;
    p.ldc = ldd;
    p.rows_per_split = (p.R + 31) / 32 * 32;
  } else {
    if (part == nullptr) return fail("gemm_tn: no partial buffer");
    p.Cpart = part;
    p.ldc = p.N;
    long rps = (p.R + ns - 1) / ns;
    p.rows_per_split = (rps + 31) / 32 * 32;
    ns = (int)((p.R + p.rows_per_split - 1) / p.rows_per_split);
  }
  const unsigned tiles = (unsigned)((p.M / 256) * (p.N / 256));
  dim3 grid(tiles, (unsigned)ns);
  p.task_ns = 0;
  int* sync_ws = p.task_sync;  // TN_SYNC_INTS ints of the CALLER's workspace (or NULL: unpaced)
  p.task_sync = nullptr;
  if (p.M == 3072 && p.N == 3072 && ns >= 2) {
    p.task_ns = ns;
    grid = dim3(tn_task_grid(ns), 1);
    // arrival counters of the region tasks (gemm_tn_fast.hpp): pacing only, they carry no result.  They live in the
    // workspace of the call that launches the kernel, so two streams (two workspaces) never share them
    if (SYNC && sync_ws != nullptr && ns * 4 * 4 <= TN_SYNC_INTS) {
      HIP_OK(hipMemsetAsync(sync_ws, 0, (size_t)ns * 4 * 4 * sizeof(int), st));
      p.task_sync = sync_ws;
    }
  }
  {
    ProfScope ps(100 + TA_PLAIN * 10 + TB, 2.0 * (double)R_all * (double)p.M * (double)p.N, st);
    hipLaunchKernelGGL(kern, grid, dim3(512), TN_FAST_LDS_BYTES, st, p);
    if (tail != 0) {  // rows [R - tail, R): one more partial, from the 128-tile generic kernel
      auto tk = gemm_tn_kernel<TA_PLAIN, TB, false>;
      static bool tattr[64] = {false};
      if (dev < 64 && !tattr[dev]) {
        HIP_OK(hipFuncSetAttribute((const void*)tk, hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS_BYTES));
        tattr[dev] = true;
      }
      TnParams t = p;
      t.R = R_all; t.row_base = R_all - tail; t.rows_per_split = 32;
      t.Cpart = part + (size_t)ns * p.M * p.N; t.ldc = p.N; t.task_ns = 0; t.task_sync = nullptr;
      hipLaunchKernelGGL(tk, dim3((unsigned)((p.M / 128) * (p.N / 128)), 1), dim3(256), TN_LDS_BYTES, st, t);
    }
  }
  HIP_OK(hipGetLastError());
  if (ns > 1 || tail != 0) {
    hipLaunchKernelGGL(k_splitk_reduce, dim3(nblk((long)p.M * p.N, 256)), dim3(256), 0, st, (const float*)part,
                       ns + (tail != 0 ? 1 : 0), p.M, p.N, (long)p.N, dst, ldd);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

// TR (NP = 1 only): the transpose-read kernel of gemm_bf16.hpp (16-byte row loads, K-major LDS image, ds_read_b64_tr_b16);
// ABF16: its A operand is the in-place bf16 dz of bwd_bf16_dz.hpp
template <int TB, int NP = 3, bool TR = false, bool ABF16 = false>
static int launch_tn_bf16x3(TnParams p, float* dst, long ldd, float* part, size_t part_cap_floats, hipStream_t st) {
  void (*kern)(TnParams) = nullptr;
  constexpr bool SYNC = TR && TB == TB_AFFINE_RELU;  // pacing for the kind whose two operands both stream from HBM
  void (*kern_alt)(TnParams) = nullptr;  // TR: the 32 x 32 x 16 form (pn_set_bf16_mfma16(0))
  if constexpr (TR) {
    kern = gemm_tn_bf16tr_kernel<TB, ABF16, SYNC, true>;
    kern_alt = gemm_tn_bf16tr_kernel<TB, ABF16, SYNC, false>;
  } else {
    kern = gemm_tn_bf16x3_kernel<TB, NP>;
  }
  constexpr int LDS = TR ? TN_BF16TR_LDS_BYTES : 2 * 512 * 36 * (int)sizeof(float);
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    if (kern_alt != nullptr) HIP_OK(hipFuncSetAttribute((const void*)kern_alt, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done[dev] = true;
  }
  if (kern_alt != nullptr && !g_bf16_m16.load()) kern = kern_alt;
  int ns = tn_pick_split(p.R, p.M, p.N, part_cap_floats, 256, 1);
  if (ns == 1) {
    p.Cpart = dst;
    p.ldc = ldd;
    p.rows_per_split = (p.R + 31) / 32 * 32;
  } else {
    if (part == nullptr) return fail("gemm_tn: no partial buffer");
    p.Cpart = part;
    p.ldc = p.N;
    long rps = (p.R + ns - 1) / ns;
    p.rows_per_split = (rps + 31) / 32 * 32;
    ns = (int)((p.R + p.rows_per_split - 1) / p.rows_per_split);
  }
  const unsigned tiles = (unsigned)((p.M / 256) * (p.N / 256));
  dim3 grid(tiles, (unsigned)ns);
  p.task_ns = 0;
  int* sync_ws = p.task_sync;  // TN_SYNC_INTS ints of the CALLER's workspace (or NULL: unpaced)
  p.task_sync = nullptr;
  if (p.M == 3072 && p.N == 3072 && ns >= 2) {
    p.task_ns = ns;
    grid = dim3(tn_task_grid(ns), 1);
    if (SYNC && sync_ws != nullptr && ns * 4 * 4 <= TN_SYNC_INTS) {  // arrival counters of the region tasks: pacing only
      HIP_OK(hipMemsetAsync(sync_ws, 0, (size_t)ns * 4 * 4 * sizeof(int), st));
      p.task_sync = sync_ws;
    }
  }
  {
    ProfScope ps((NP == 3 ? 1100 : 1600) + TB, 2.0 * (double)p.R * (double)p.M * (double)p.N, st);
    hipLaunchKernelGGL(kern, grid, dim3(512), LDS, st, p);
  }
  HIP_OK(hipGetLastError());
  if (ns > 1) {
    hipLaunchKernelGGL(k_splitk_reduce, dim3(nblk((long)p.M * p.N, 256)), dim3(256), 0, st, (const float*)part, ns,
                       p.M, p.N, (long)p.N, dst, ldd);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

template <int TA, int TB>
static int launch_tn(TnParams p, float* dst, long ldd, float* part, size_t part_cap_floats, hipStream_t st) {
  if (p.M % 4 || p.N % 4) return fail("gemm_tn: M and N must be multiples of 4");
  if (p.R <= 0) return fail("gemm_tn: empty contraction");
  if (p.drop_thresh != 0) {  // dropped hidden activations as the B operand (training): f32 kernels, mask in the loader
    if constexpr (TA == TA_PLAIN && (TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU)) {
      if (PN_BIG && use_f32_dma() && p.M % 256 == 0 && p.N % 256 == 0 && p.R >= 65536 && p.R % 32 == 0 && p.lda % 4 == 0)
        return launch_tn_cfg<TA, TB, true, true, true>(p, dst, ldd, part, part_cap_floats, st);
      return launch_tn_cfg<TA, TB, false, false, true>(p, dst, ldd, part, part_cap_floats, st);
    } else {
      return fail("gemm_tn: dropout is not defined for operand kinds %d x %d", TA, TB);
    }
  }
  if constexpr (TA == TA_PLAIN && (TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU)) {  // pn_set_backward_math(1)
    if (tl_bwd_bf16 && tl_dz_bf16)  // (preconditions checked by pn_pairhead_bwd before it wrote dz as bf16)
      return launch_tn_bf16x3<TB, 1, true, true>(p, dst, ldd, part, part_cap_floats, st);
    if (tl_bwd_bf16 && PN_BIG && p.M % 256 == 0 && p.N % 256 == 0 && p.R >= 65536 &&
        (TB != TB_PAIRSUM_RELU || p.pairB % 8 == 0)) {
      // the transpose-read kernel: whole 32-row slabs only, a slab inside one label, 16-byte aligned rows, 32-bit row offsets
      if ((g_bwd_deep & 2) && p.R % 32 == 0 && (TB != TB_PAIRSUM_RELU || (p.pairB % 32 == 0 && p.ldb2 % 4 == 0)) &&
          p.lda % 4 == 0 && p.ldb % 4 == 0 && (long)8 * p.lda * 4 < (1L << 31) && (long)8 * p.ldb * 4 < (1L << 31) &&
          (TB != TB_AFFINE_RELU || p.b_s != nullptr))
        return launch_tn_bf16x3<TB, 1, true>(p, dst, ldd, part, part_cap_floats, st);
      return launch_tn_bf16x3<TB, 1>(p, dst, ldd, part, part_cap_floats, st);
    }
  }
  if constexpr (TA == TA_PLAIN && (TB == TB_PLAIN || TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU)) {
    if (cur_math() == 1 && PN_BIG && p.M % 256 == 0 && p.N % 256 == 0 && p.R >= 65536 &&
        (TB != TB_PAIRSUM_RELU || p.pairB % 8 == 0))
      return launch_tn_bf16x3<TB>(p, dst, ldd, part, part_cap_floats, st);
  }
  // 256x256 tiles for the big weight gradients (M, N multiples of 256 and a long contraction); a plain A operand (the
  // materialised dz) is staged by LDS-DMA when every split is a whole number of 32-row slabs
  if constexpr (TA == TA_PLAIN && (TB == TB_PLAIN || TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU)) {
    if (PN_BIG && use_f32_dma() && p.M % 256 == 0 && p.N % 256 == 0 && p.R >= 16384 && p.lda % 4 == 0) {
      if constexpr (TB == TB_AFFINE_RELU || TB == TB_PAIRSUM_RELU) {
        // the low-VALU kernel: 32-bit per-lane offsets, and for the pair sum a slab inside one label (B % 32 == 0); a row
        // count that is not a multiple of 32 leaves its tail to one extra partial
        const bool fits = (long)8 * p.ldb * 4 < (1L << 31) && p.ldb % 4 == 0 && (TB != TB_AFFINE_RELU || p.b_s != nullptr);
        const bool tail_ok = p.R % 32 == 0 || (part != nullptr && part_cap_floats >= 3 * (size_t)p.M * p.N);
        const bool pair_ok = TB != TB_PAIRSUM_RELU || (p.pairB % 32 == 0 && p.ldb2 % 4 == 0);
        if (fits && tail_ok && pair_ok) return launch_tn_fast<TB>(p, dst, ldd, part, part_cap_floats, st);
      }
      if (p.R % 32 == 0) return launch_tn_cfg<TA, TB, true, true>(p, dst, ldd, part, part_cap_floats, st);
    }
  }
  if (PN_BIG && p.M % 256 == 0 && p.N % 256 == 0 && p.R >= 16384)
    return launch_tn_cfg<TA, TB, true>(p, dst, ldd, part, part_cap_floats, st);
  return launch_tn_cfg<TA, TB, false>(p, dst, ldd, part, part_cap_floats, st);
}

static TnParams tn_zero() {
  TnParams p;
  memset(&p, 0, sizeof(p));
  p.pairB = 1;
  return p;
}

static const size_t TN_PART_FLOATS_MAX = (size_t)16 * 3072 * 3072;

// BatchNorm-backward vectors of the dz generator from this rank's S1 = sum du, S2 = sum du * xhat.  With SYNC_BN the
// first launch takes dgamma / dbeta (and dw_out) from the LOCAL sums, then S1 / S2 are summed over the ranks and a second
// launch overwrites cs / p / q with the global ones (count * world rows).
static int bwd_finalize(hipStream_t st, const double* S1, const double* S2, const double* dwacc, double count, int C,
                        const float* gamma, const float* s, const float* mean, const float* invstd, const float* w,
                        float* cs, float* pv, float* qv, float* dgamma, float* dbeta, float* dw_out) {
  hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(nblk(C, 256)), dim3(256), 0, st, S1, S2, dwacc, count,
                     (const double*)nullptr, C, gamma, s, mean, invstd, w, cs, pv, qv, dgamma, dbeta, dw_out,
                     tl_bn_running ? 1 : 0);
  HIP_OK(hipGetLastError());
  if (sync_bn_on() && gamma != nullptr && !tl_bn_running) {
    const double* gcount = nullptr;
    PN_OK(sync_sum2(const_cast<double*>(S1), const_cast<double*>(S2), C, count, &gcount, st));
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(nblk(C, 256)), dim3(256), 0, st, S1, S2, (const double*)nullptr,
                       count, gcount, C, gamma, s, mean, invstd, w, cs, pv, qv, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, 0);
    HIP_OK(hipGetLastError());
  }
  return 0;
}
