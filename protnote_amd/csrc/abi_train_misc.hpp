// Part of the translation unit protnote_hip.hip (#included there, after the launchers and small kernels; not a
// stand-alone header: it uses the static helpers defined above its #include):
// loss + metrics, optimiser, layout helpers, similarity backward, save_embeddings, attention pooling, batch assembly, encoder backward.
// ------------------------------------------------------------------------------------------------
// loss + metrics, optimiser, layout helpers
// ------------------------------------------------------------------------------------------------
extern "C" size_t pn_loss_ws_bytes(int B, int N) {
  return al256(512 + (size_t)B * sizeof(float)) + al256((size_t)nblk(N, 256) * nblk(B, 32) * sizeof(double));
}

// targets as one typed pointer -> the three typed pointers of the kernels (exactly one non-null)
struct TargetPtrs {
  const float* f;
  const int64_t* i;
  const uint8_t* u;
};
static int typed_targets(const void* targets, int kind, const char* who, TargetPtrs* t) {
  t->f = nullptr; t->i = nullptr; t->u = nullptr;
  if (targets == nullptr) return fail("%s: targets are NULL", who);
  if (kind == PN_LABEL_F32) t->f = (const float*)targets;
  else if (kind == PN_LABEL_I64) t->i = (const int64_t*)targets;
  else if (kind == PN_LABEL_U8) t->u = (const uint8_t*)targets;
  else return fail("%s: target_kind %d (PN_LABEL_F32 = 0, PN_LABEL_I64 = 1, PN_LABEL_U8 = 2)", who, kind);
  return 0;
}

extern "C" int pn_loss_fwd_bwd_t(const float* logits, const void* targets, int target_kind, int B, int N, int kind,
                                 float pos_weight, float gamma, float alpha, float smoothing, float threshold, float* loss_out,
                                 float* dlogits, float* tp, float* fn, float* fp, int weight_mode, const float* label_weights,
                                 float rgd_temperature, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  TargetPtrs tg;
  PN_OK(typed_targets(targets, target_kind, "loss", &tg));
  const float* targets_f32 = tg.f;
  const int64_t* targets_i64 = tg.i;
  if (ws_bytes < pn_loss_ws_bytes(B, N)) return fail("loss: workspace too small (need pn_loss_ws_bytes(B, N))");
  if (weight_mode < 0 || weight_mode > 2) return fail("loss: weight_mode must be 0, 1 (batch) or 2 (label weights)");
  if (weight_mode == 2 && label_weights == nullptr) return fail("loss: weight_mode 2 needs label_weights");
  double* acc = (double*)ws;               // [0] loss sum, [1] number of positives (integer-valued: order-free)
  float* posneg = (float*)((char*)ws + 256);
  float* row_w = (float*)((char*)ws + 512);
  const dim3 lgrid(nblk(N, 256), nblk(B, 32));
  double* lpart = (double*)((char*)ws + al256(512 + (size_t)B * sizeof(float)));  // one loss partial per workgroup
  HIP_OK(hipMemsetAsync(acc, 0, 2 * sizeof(double), st));
  LossParams p;
  memset(&p, 0, sizeof(p));
  p.logits = logits; p.tf = targets_f32; p.ti = targets_i64; p.tu = tg.u; p.B = B; p.N = N; p.kind = kind;
  p.pos_weight = pos_weight; p.gamma = gamma; p.alpha = alpha; p.smoothing = smoothing; p.threshold = threshold;
  p.grad_scale = 1.f / ((float)B * (float)N);
  p.dlogits = dlogits; p.loss_part = lpart; p.tp = tp; p.fn = fn; p.fp = fp;
  p.rows_per_block = 32;
  if (weight_mode != 0) {
    hipLaunchKernelGGL(k_target_weights, dim3(nblk(B, 4)), dim3(256), 0, st, targets_f32, targets_i64, tg.u, B, N,
                       weight_mode == 2 ? label_weights : (const float*)nullptr,
                       weight_mode == 2 ? row_w : (float*)nullptr, acc + 1);
    if (weight_mode == 1) {
      hipLaunchKernelGGL(k_posneg_weights, dim3(1), dim3(1), 0, st, (const double*)(acc + 1), (double)B * (double)N,
                         1e-10, posneg);
      p.posneg = posneg;
    } else {
      p.row_w = row_w;
    }
  }
  {  // K13/K14: 4 B logit + 1 B target (algorithmic: a multihot; PN_LABEL_U8 reads exactly that) read, 4 B gradient written
    ProfScope ps(ST_LOSS, (double)B * (double)N * (dlogits ? 9.0 : 5.0), st);
    hipLaunchKernelGGL(k_loss, lgrid, dim3(256), 0, st, p);
  }
  hipLaunchKernelGGL(k_scalar_final, dim3(1), dim3(256), 0, st, (const double*)lpart, (int)(lgrid.x * lgrid.y), acc);
  if (rgd_temperature >= 0.f)
    hipLaunchKernelGGL(k_rgd_scale, dim3(dlogits ? 1024 : 1), dim3(256), 0, st, (const double*)acc,
                       1.0 / ((double)B * (double)N), rgd_temperature, dlogits, (long)B * N, loss_out);
  else
    hipLaunchKernelGGL(k_d2f, dim3(1), dim3(64), 0, st, (const double*)acc, loss_out, 1,
                       1.f / ((float)B * (float)N));
  HIP_OK(hipGetLastError());
  return 0;
}

// the two-pointer form (exactly one of targets_f32 / targets_i64 non-NULL)
extern "C" int pn_loss_fwd_bwd(const float* logits, const float* targets_f32, const int64_t* targets_i64, int B,
                               int N, int kind, float pos_weight, float gamma, float alpha, float smoothing,
                               float threshold, float* loss_out, float* dlogits, float* tp, float* fn, float* fp,
                               int weight_mode, const float* label_weights, float rgd_temperature, void* ws,
                               size_t ws_bytes, void* stream) {
  if ((targets_f32 == nullptr) == (targets_i64 == nullptr)) return fail("loss: pass exactly one target array");
  return pn_loss_fwd_bwd_t(logits, targets_f32 ? (const void*)targets_f32 : (const void*)targets_i64,
                           targets_f32 ? PN_LABEL_F32 : PN_LABEL_I64, B, N, kind, pos_weight, gamma, alpha, smoothing, threshold,
                           loss_out, dlogits, tp, fn, fp, weight_mode, label_weights, rgd_temperature, ws, ws_bytes, stream);
}

extern "C" size_t pn_supcon_ws_bytes(int B) { return al256((size_t)B * sizeof(double)) + 256; }

// LOSS_FN: SupCon (reference utils/losses.py:7-56).  loss_out [1]; dlogits [B][N] or NULL.
extern "C" int pn_supcon_fwd_bwd(const float* logits, const float* targets_f32, const int64_t* targets_i64, int B, int N,
                                 float* loss_out, float* dlogits, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((targets_f32 == nullptr) == (targets_i64 == nullptr)) return fail("supcon: pass exactly one target array");
  if (ws_bytes < pn_supcon_ws_bytes(B)) return fail("supcon: workspace too small");
  double* acc = (double*)ws;
  double* rows = (double*)((char*)ws + 256);
  hipLaunchKernelGGL(k_supcon, dim3(B), dim3(256), 0, st, logits, targets_f32, targets_i64, B, N, dlogits, rows);
  hipLaunchKernelGGL(k_scalar_final, dim3(1), dim3(256), 0, st, (const double*)rows, B, acc);
  hipLaunchKernelGGL(k_d2f, dim3(1), dim3(64), 0, st, (const double*)acc, loss_out, 1, 1.f / (float)B);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_tp_fn_fp_t(const float* probs, const void* targets, int target_kind, int B, int N, float threshold,
                             float* tp, float* fn, float* fp, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  TargetPtrs tg;
  PN_OK(typed_targets(targets, target_kind, "tp_fn_fp", &tg));
  HIP_OK(hipMemsetAsync(tp, 0, N * sizeof(float), st));
  HIP_OK(hipMemsetAsync(fn, 0, N * sizeof(float), st));
  HIP_OK(hipMemsetAsync(fp, 0, N * sizeof(float), st));
  hipLaunchKernelGGL(k_tp_fn_fp, dim3(nblk(N, 256), nblk(B, 64)), dim3(256), 0, st, probs, tg.f, tg.i, tg.u, B, N, threshold, tp,
                     fn, fp, 64);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_tp_fn_fp(const float* probs, const float* targets_f32, const int64_t* targets_i64, int B, int N,
                           float threshold, float* tp, float* fn, float* fp, void* stream) {
  if ((targets_f32 == nullptr) == (targets_i64 == nullptr)) return fail("tp_fn_fp: pass exactly one target array");
  return pn_tp_fn_fp_t(probs, targets_f32 ? (const void*)targets_f32 : (const void*)targets_i64,
                       targets_f32 ? PN_LABEL_F32 : PN_LABEL_I64, B, N, threshold, tp, fn, fp, stream);
}

extern "C" int pn_clip_adam_step(float* w, const float* g, float* m, float* v, long n, float max_norm, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step, float* norm_out,
                                 void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (ws_bytes < PN_ADAM_WS_BYTES) return fail("adam: workspace too small (need %d bytes)", PN_ADAM_WS_BYTES);
  if (step < 1) return fail("adam: step must be >= 1");
  double* acc = (double*)ws;  // [0] sum of squares, [32..) one partial per workgroup of k_sumsq
  ProfScope ps(ST_CLIP_OPT, 28.0 * (double)n, st);  // K16: p, g, m, v read (16 B) + p, m, v written (12 B) per parameter
  hipLaunchKernelGGL(k_sumsq, dim3(2048), dim3(256), 0, st, g, n, acc + 32);
  hipLaunchKernelGGL(k_scalar_final, dim3(1), dim3(256), 0, st, (const double*)(acc + 32), 2048, acc);
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adam, dim3(2048), dim3(256), 0, st, w, g, m, v, n, (const double*)acc, max_norm, lr, beta1,
                     beta2, eps, bc1, bc2s, weight_decay, norm_out);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_clip_sgd_step(float* w, const float* g, float* momentum_buf, long n, float max_norm, float lr,
                                float momentum, float weight_decay, int step, float* norm_out, void* ws,
                                size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (ws_bytes < PN_ADAM_WS_BYTES) return fail("sgd: workspace too small (need %d bytes)", PN_ADAM_WS_BYTES);
  if (step < 1) return fail("sgd: step must be >= 1");
  if (momentum != 0.f && momentum_buf == nullptr) return fail("sgd: momentum %g needs a momentum buffer", momentum);
  double* acc = (double*)ws;
  ProfScope ps(ST_CLIP_OPT, (momentum != 0.f ? 20.0 : 12.0) * (double)n, st);  // p, g (, buf) read + p (, buf) written
  hipLaunchKernelGGL(k_sumsq, dim3(2048), dim3(256), 0, st, g, n, acc + 32);
  hipLaunchKernelGGL(k_scalar_final, dim3(1), dim3(256), 0, st, (const double*)(acc + 32), 2048, acc);
  hipLaunchKernelGGL(k_sgd, dim3(2048), dim3(256), 0, st, w, g, momentum != 0.f ? momentum_buf : (float*)nullptr, n,
                     (const double*)acc, max_norm, lr, momentum, weight_decay, step == 1 ? 1 : 0, norm_out);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_transpose(const float* src, long ld_src, int rows, int cols, float* dst, long ld_dst, void* stream) {
  return transpose_into(src, ld_src, rows, cols, dst, ld_dst, (hipStream_t)stream);
}

extern "C" int pn_gemm_tn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, long R, int M,
                          int N, void* ws, size_t ws_bytes, void* stream) {
  TnParams tp = tn_zero();
  tp.R = R; tp.M = M; tp.N = N; tp.A = A; tp.lda = lda; tp.B = Bm; tp.ldb = ldb;
  return launch_tn<TA_PLAIN, TB_PLAIN>(tp, C, ldc, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// similarity head, training (ProtNote.py:281-284 + autograd):  logits = (P^ L^T) / T,  X^ = X / max(|X|, eps)
// ------------------------------------------------------------------------------------------------
// xhat[r][:] = x[r][:] * rs[r]
__global__ void k_scale_rows(const float* __restrict__ x, const float* __restrict__ rs, float* __restrict__ out,
                             long rows, int d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * d) return;
  out[i] = x[i] * rs[i / d];
}

// dx = rs * (dxhat - xhat * <xhat, dxhat>)   (rows with |x| < eps: normalisation is x/eps, dx = dxhat/eps)
__global__ void k_normalize_bwd(const float* __restrict__ xhat, const float* __restrict__ dxhat,
                                const float* __restrict__ rs, float alpha, float* __restrict__ dx, int rows, int d) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* xh = xhat + (long)r * d;
  const float* dh = dxhat + (long)r * d;
  float dot = 0.f;
  for (int c = lane; c < d; c += 64) dot = fmaf(xh[c], dh[c], dot);
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  const float s = rs[r];
  const bool clamped = s >= 1e12f;  // |x| <= 1e-12: F.normalize divides by eps, a constant
  for (int c = lane; c < d; c += 64) dx[(long)r * d + c] = alpha * s * (clamped ? dh[c] : dh[c] - xh[c] * dot);
}

struct SimWs {
  float *rs, *cs, *Ph, *Lh, *dPh, *dLh, *T1, *PhT, *part;
  size_t part_floats;
};
static bool sim_carve(int B, int NL, int d, Bump& bp, SimWs& w) {
  const int Bp = ld4(B);
  w.rs = bp.take<float>(B);
  w.cs = bp.take<float>(NL);
  w.Ph = bp.take<float>((size_t)B * d);
  w.Lh = bp.take<float>((size_t)NL * d);
  w.dPh = bp.take<float>((size_t)Bp * d);
  w.dLh = bp.take<float>((size_t)NL * d);
  w.T1 = bp.take<float>((size_t)NL * Bp);   // dlogits^T, zero-padded columns
  w.PhT = bp.take<float>((size_t)d * Bp);   // P^^T, zero-padded columns
  // split-K partial tiles of dP^ = dlogits P-side contraction over the NL label rows: its [Bp x d] output is only a
  // handful of tiles, so the rows are split ~128 ways to fill the chip (one workgroup per tile ran at 4 TFLOP/s)
  w.part_floats = (size_t)128 * Bp * d < TN_PART_FLOATS_MAX ? (size_t)128 * Bp * d : TN_PART_FLOATS_MAX;
  w.part = bp.take<float>(w.part_floats);
  return bp.ok;
}

extern "C" size_t pn_similarity_train_ws_bytes(int B, int NL, int d) {
  Bump bp(nullptr, (size_t)-1);
  SimWs w;
  sim_carve(B, NL, d, bp, w);
  return bp.off + 256;
}

// dP_e, dL_e from dlogits [B][NL] (any NL: rows of dlogits need not be 16-byte aligned - everything goes
// through the zero-padded transpose T1 = dlogits^T [NL][ld4(B)]); the normalisations are recomputed.
extern "C" int pn_similarity_bwd(const float* P_e, const float* L_e, int B, int NL, int d, float temperature,
                                 const float* dlogits, float* dP_e, float* dL_e, void* ws, size_t ws_bytes,
                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (d % 4) return fail("similarity bwd: d must be a multiple of 4");
  Bump bp(ws, ws_bytes);
  SimWs w;
  if (!sim_carve(B, NL, d, bp, w)) return fail("similarity bwd: workspace too small");
  const int Bp = ld4(B);
  hipLaunchKernelGGL(k_rownorm_inv, dim3(nblk(B, 4)), dim3(256), 0, st, P_e, (long)d, B, d, w.rs);
  hipLaunchKernelGGL(k_rownorm_inv, dim3(nblk(NL, 4)), dim3(256), 0, st, L_e, (long)d, NL, d, w.cs);
  hipLaunchKernelGGL(k_scale_rows, dim3(nblk((long)B * d, 256)), dim3(256), 0, st, P_e, (const float*)w.rs, w.Ph,
                     (long)B, d);
  hipLaunchKernelGGL(k_scale_rows, dim3(nblk((long)NL * d, 256)), dim3(256), 0, st, L_e, (const float*)w.cs, w.Lh,
                     (long)NL, d);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemsetAsync(w.T1, 0, (size_t)NL * Bp * sizeof(float), st));
  HIP_OK(hipMemsetAsync(w.PhT, 0, (size_t)d * Bp * sizeof(float), st));
  PN_OK(transpose_into(dlogits, NL, B, NL, w.T1, Bp, st));  // T1[j][i] = dl[i][j]
  PN_OK(transpose_into(w.Ph, d, B, d, w.PhT, Bp, st));       // PhT[k][i] = P^[i][k]
  const float alpha = 1.f / temperature;
  // both backward contractions of the cosine head under one timing kind (900: 2 x 2 B NL d FLOP)
  ProfScope ps_sim(900, 4.0 * (double)B * (double)NL * (double)d, st);
  // dP^[i][k] = sum_j T1[j][i] L^[j][k]   (contraction over the NL rows)
  {
    TnParams tp = tn_zero();
    tp.R = NL; tp.M = Bp; tp.N = d; tp.A = w.T1; tp.lda = Bp; tp.B = w.Lh; tp.ldb = d;
    PN_OK((launch_tn<TA_PLAIN, TB_PLAIN>(tp, w.dPh, d, w.part, w.part_floats, st)));
  }
  // dL^[j][k] = sum_i T1[j][i] P^[i][k]
  {
    GemmParams p = gp_zero();
    p.M = NL; p.N = d; p.Nstore = d; p.Kseg = Bp;
    p.A = w.T1; p.lda = Bp; p.W = w.PhT; p.ldw = Bp; p.C = w.dLh; p.ldc = d;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
  }
  hipLaunchKernelGGL(k_normalize_bwd, dim3(nblk(B, 4)), dim3(256), 0, st, (const float*)w.Ph, (const float*)w.dPh,
                     (const float*)w.rs, alpha, dP_e, B, d);
  hipLaunchKernelGGL(k_normalize_bwd, dim3(nblk(NL, 4)), dim3(256), 0, st, (const float*)w.Lh, (const float*)w.dLh,
                     (const float*)w.cs, alpha, dL_e, NL, d);
  HIP_OK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// save_embeddings path (ProtNote.py:294-302): logits AND the penultimate activations of the output MLP
// ------------------------------------------------------------------------------------------------
__global__ void k_affine_relu_rows(const float* __restrict__ z, long ldz, float* __restrict__ out, long ldo, long rows,
                                   int cols, const float* __restrict__ s, const float* __restrict__ t) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  out[r * ldo + c] = fmaxf(fmaf(z[r * ldz + c], s[c], t[c]), 0.f);
}

extern "C" size_t pn_pairhead_hidden_ws_bytes(const pn_pairhead* hd, int B, int NL) {
  const size_t rh = (size_t)B * NL * hd->h * sizeof(float);
  return pn_pairhead_eval_ws_bytes(hd, B, NL, NL) + 2 * al256(rh) + 4096;
}

// hidden_pairs[r][h] (r = j*B + i) = relu(bn(z_last)), logits_pairs[r] = hidden . w_out + b_out; eval-mode BN.
// Meant for the small subsets the reference saves embeddings for (the full [B*NL, h] tensor is materialised).
extern "C" int pn_pairhead_fwd_eval_hidden(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                                           float* logits_pairs, float* hidden_pairs, void* ws, size_t ws_bytes,
                                           void* stream) {
  PN_OK(math_field_check(hd->math_mode, "pn_pairhead_fwd_eval_hidden"));
  MathScope math_scope(hd->math_mode);
  bool fwd_bf16 = false;
  PN_OK(fwd_math_of(hd, &fwd_bf16));
  struct NoStage {  // this entry point reads the f32 activations of the chunk back: the register-staged single-product route
    bool prev;
    NoStage() : prev(tl_fwd_nostage) { tl_fwd_nostage = true; }
    ~NoStage() { tl_fwd_nostage = prev; }
  } no_stage;
  hipStream_t st = (hipStream_t)stream;
  const int h = hd->h;
  const long R = (long)B * NL;
  if (hd->nlayers < 1) return fail("pairhead hidden: nlayers < 1 unsupported");
  if (hd->nlayers == 1) {
    // one hidden layer: the penultimate activation is relu(bn(z1)) itself - relu(A'[i] + B'[j]) from the two folded tables,
    // or (concatenation_prod) the stored z1 of the single chunk through its fold
    const size_t base1 = pn_pairhead_eval_ws_bytes(hd, B, NL, NL);
    if (ws_bytes < base1) return fail("pairhead hidden: workspace too small");
    PN_OK(pn_pairhead_fwd_eval(hd, P_e, L_e, B, NL, logits_pairs, NL, ws, base1, stream));
    Bump bp1(ws, base1);
    PairWs w1;
    if (!pair_carve(hd, B, NL, NL, bp1, w1)) return fail("pairhead hidden: workspace carve failed");
    if (hd->fusion == 2)
      hipLaunchKernelGGL(k_affine_relu_rows, dim3(nblk(R * h, 256)), dim3(256), 0, st, (const float*)w1.z[0], (long)h,
                         hidden_pairs, (long)h, R, h, (const float*)w1.s[0], (const float*)w1.t[0]);
    else
      hipLaunchKernelGGL(k_pairsum_relu_rows, dim3(nblk(R * h, 256)), dim3(256), 0, st, (const float*)w1.A1, (long)h,
                         (const float*)w1.B1, (long)h, B, R, h, hidden_pairs, (long)h);
    HIP_OK(hipGetLastError());
    return 0;
  }
  // Run the eval head with one layer fewer and a unit "output neuron" trick is not possible (the row-dot epilogue
  // never stores), so: layers 1..n-2 through the normal path into a scratch z, the last hidden layer stored too.
  const size_t base = pn_pairhead_eval_ws_bytes(hd, B, NL, NL);
  if (ws_bytes < base + 2 * al256((size_t)R * h * sizeof(float))) return fail("pairhead hidden: workspace too small");
  float* zlast = (float*)((char*)ws + al256(base));
  // 1) logits (also prepares A', B', the folded BN vectors and, for n >= 3, z_{n-2} of the single chunk in ws)
  PN_OK(pn_pairhead_fwd_eval(hd, P_e, L_e, B, NL, logits_pairs, NL, ws, base, stream));
  // 2) recompute the last hidden pre-activation with a storing epilogue
  Bump bp(ws, base);
  PairWs w;
  if (!pair_carve(hd, B, NL, NL, bp, w)) return fail("pairhead hidden: workspace carve failed");
  const int li = hd->nlayers - 1;
  const bool prod = hd->fusion == 2;
  GemmParams p = gp_zero();
  p.M = (int)R; p.N = h; p.Nstore = h; p.Kseg = h;
  p.W = hd->w[li]; p.ldw = h; p.C = zlast; p.ldc = h;
  FwdBf16Scope fwd_scope(fwd_bf16);  // the same arithmetic as the logits of step 1
  if (fwd_bf16) p.wsplit = w.wsplit;
  if (li == 1 && !prod) {
    p.A = w.A1; p.lda = h; p.A2 = w.B1; p.lda2 = h; p.pairB = B;
    PN_OK((launch_gemm<A_PAIRSUM_RELU, E_STORE>(p, 0, st)));
  } else {
    // the stored input of the last layer: ping-pong position after (li - 1) [+1 for prod] stores
    const int nstores = (li - 1) + (prod ? 1 : 0);
    p.A = w.z[(nstores - 1) & 1]; p.lda = h;
    if (prod && li == 1) {  // z1 of concatenation_prod is stored raw (E_PAIRADD)
      p.a_scale = w.s[li - 1]; p.a_shift = w.t[li - 1];
      PN_OK((launch_gemm<A_AFFINE_RELU, E_STORE>(p, 0, st)));
    } else {  // pn_pairhead_fwd_eval stored relu(bn(z_{li-1})) (producer-side activation)
      PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
    }
  }
  hipLaunchKernelGGL(k_affine_relu_rows, dim3(nblk(R * h, 256)), dim3(256), 0, st, (const float*)zlast, (long)h,
                     hidden_pairs, (long)h, R, h, (const float*)w.s[li], (const float*)w.t[li]);
  HIP_OK(hipGetLastError());
  return 0;
}

// save_embeddings on the activation-storing path (ProtNote.py:292-302 under model.train(), or eval mode with autograd on):
// the penultimate activations relu(bn(z_last)) [NL*B][h] of the forward whose activations `save` holds - read back from
// the store (the last hidden pre-activation and its BatchNorm fold), nothing is recomputed.  Call after
// pn_pairhead_fwd_train and before pn_pairhead_bwd (the backward consumes the store in place).
extern "C" int pn_pairhead_train_hidden(const pn_pairhead* hd, int B, int NL, int label_chunk, const void* save,
                                        size_t save_bytes, float* hidden_pairs, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  PN_OK(pair_check(hd, B, NL));
  const int h = hd->h, n = hd->nlayers;
  const long R = (long)B * NL, S = pair_chunk_rows(B, NL, label_chunk);
  Bump bs(const_cast<void*>(save), save_bytes);
  PairSave sv;
  if (!pair_save_carve(hd, B, NL, S, bs, sv)) return fail("pairhead train hidden: save buffer too small");
  if (n == 1 && hd->fusion != 2)
    hipLaunchKernelGGL(k_pairsum_relu_rows, dim3(nblk(R * h, 256)), dim3(256), 0, st, (const float*)sv.Ap, (long)h,
                       (const float*)sv.Bp, (long)h, B, R, h, hidden_pairs, (long)h);
  else
    hipLaunchKernelGGL(k_affine_relu_rows, dim3(nblk(R * h, 256)), dim3(256), 0, st,
                       (const float*)(sv.zbuf[n - 1] + (size_t)S * h), (long)h, hidden_pairs, (long)h, R, h,
                       (const float*)sv.s[n - 1], (const float*)sv.t[n - 1]);
  HIP_OK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// additive attention pooling over label tokens (ProtNote.py:154-166), inference:
//   out[n][:] = sum_t softmax_t(mask ? w.h[n][t] + b : -inf) * h[n][t][:]
// one workgroup per label
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_additive_attention(const float* __restrict__ hid, const int64_t* __restrict__ mask,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            int T, int d, float* __restrict__ out) {
  extern __shared__ float sc[];  // [T] scores
  const int n = blockIdx.x;
  const float* hn = hid + (long)n * T * d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < T; t += 4) {
    float a = 0.f;
    for (int c = lane; c < d; c += 64) a = fmaf(hn[(long)t * d + c], w[c], a);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) sc[t] = mask[(long)n * T + t] != 0 ? a + b[0] : -INFINITY;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(sc[t] - mx);
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = fmaf(expf(sc[t] - mx) / den, hn[(long)t * d + c], acc);
    out[(long)n * d + c] = acc;
  }
}

extern "C" int pn_additive_attention(const float* hidden, const int64_t* attention_mask, const float* w,
                                     const float* b, int N, int T, int d, float* out, void* stream) {
  if (T <= 0 || T > 8192) return fail("additive_attention: unsupported token count %d", T);
  hipLaunchKernelGGL(k_additive_attention, dim3(N), dim3(256), T * sizeof(float), (hipStream_t)stream, hidden,
                     attention_mask, w, b, T, d, out);
  HIP_OK(hipGetLastError());
  return 0;
}

// Backward of the pooling wrt the scorer (training with LABEL_EMBEDDING_POOLING_METHOD: all, ProtNote.py:89-91,154-166):
// with a = softmax(s), out = sum_t a_t h_t and upstream gradient g = d out:
//   da_t = g . h_t,   ds_t = a_t (da_t - sum_u a_u da_u),   dw = sum_{n,t} ds_t h_t,   db = sum_{n,t} ds_t
// One workgroup per label writes its [d] partial of dw (slot d: its db partial); the partials are added in a fixed
// order afterwards.  The token embeddings are inputs (frozen label encoder): no gradient wrt h is produced.
__global__ __launch_bounds__(256) void k_additive_attention_bwd(const float* __restrict__ hid,
                                                                const int64_t* __restrict__ mask,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                const float* __restrict__ dout, int T, int d, int ldp,
                                                                float* __restrict__ part) {
  extern __shared__ float sc[];  // [T] scores -> ds, [T] da
  float* da = sc + T;
  const int n = blockIdx.x;
  const float* hn = hid + (long)n * T * d;
  const float* gn = dout + (long)n * d;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < T; t += 4) {
    float a = 0.f, g = 0.f;
    for (int c = lane; c < d; c += 64) {
      const float hv = hn[(long)t * d + c];
      a = fmaf(hv, w[c], a);
      g = fmaf(hv, gn[c], g);
    }
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o);
      g += __shfl_xor(g, o);
    }
    if (lane == 0) {
      sc[t] = mask[(long)n * T + t] != 0 ? a + b[0] : -INFINITY;
      da[t] = g;
    }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
  float den = 0.f, dot = 0.f;
  for (int t = 0; t < T; ++t) den += expf(sc[t] - mx);
  for (int t = 0; t < T; ++t) dot = fmaf(expf(sc[t] - mx) / den, da[t], dot);
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += 256) sc[t] = (expf(sc[t] - mx) / den) * (da[t] - dot);  // ds_t (0 where masked)
  __syncthreads();
  float* pn_ = part + (long)n * ldp;
  for (int c = threadIdx.x; c < d; c += 256) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = fmaf(sc[t], hn[(long)t * d + c], acc);
    pn_[c] = acc;
  }
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += sc[t];
    pn_[d] = acc;
    for (int c = d + 1; c < ldp; ++c) pn_[c] = 0.f;
  }
}

extern "C" size_t pn_additive_attention_bwd_ws_bytes(int N, int d) {
  const int ldp = ld4(d + 1);
  return al256((size_t)N * ldp * sizeof(float)) + al256((size_t)RED_CHUNKS * ldp * sizeof(double)) +
         al256((size_t)ldp * sizeof(double)) + 256;
}

extern "C" int pn_additive_attention_bwd(const float* hidden, const int64_t* attention_mask, const float* w,
                                         const float* b, const float* dout, int N, int T, int d, float* dw, float* db,
                                         void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (T <= 0 || T > 8192) return fail("additive_attention bwd: unsupported token count %d", T);
  if (N <= 0) return fail("additive_attention bwd: no labels");
  const int ldp = ld4(d + 1);
  Bump bp(ws, ws_bytes);
  float* part = bp.take<float>((size_t)N * ldp);
  double* red = bp.take<double>((size_t)RED_CHUNKS * ldp);
  double* tot = bp.take<double>(ldp);
  if (!bp.ok) return fail("additive_attention bwd: workspace too small");
  hipLaunchKernelGGL(k_additive_attention_bwd, dim3(N), dim3(256), 2 * T * sizeof(float), st, hidden, attention_mask, w,
                     b, dout, T, d, ldp, part);
  HIP_OK(hipGetLastError());
  PN_OK(reduce_parts<float>(part, N, ldp, ldp, tot, nullptr, nullptr, red, st));
  hipLaunchKernelGGL(k_d2f, dim3(nblk(d, 256)), dim3(256), 0, st, (const double*)tot, dw, d, 1.f);
  hipLaunchKernelGGL(k_d2f, dim3(1), dim3(64), 0, st, (const double*)(tot + d), db, 1, 1.f);
  HIP_OK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// device-side batch assembly: ragged uint8 residue ids -> padded f32 one-hots [B][A][Lmax] + lengths.
// The host ships B*L bytes instead of B*A*L*4 (80x less PCIe traffic than the reference's collated one-hots).
// ------------------------------------------------------------------------------------------------
__global__ void k_onehot_batch(const uint8_t* __restrict__ ids, const int64_t* __restrict__ offsets, int B, int A,
                               int Lmax, float* __restrict__ out, int64_t* __restrict__ lengths) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * A * Lmax;
  if (i >= total) return;
  const int t = (int)(i % Lmax);
  const int a = (int)((i / Lmax) % A);
  const int b = (int)(i / ((long)Lmax * A));
  const int64_t off = offsets[b];
  const int len = (int)(offsets[b + 1] - off);
  out[i] = (t < len && ids[off + t] == a) ? 1.f : 0.f;
  if (t == 0 && a == 0) lengths[b] = len;
}

extern "C" int pn_onehot_batch(const uint8_t* ids, const int64_t* offsets, int B, int A, int Lmax, float* onehots,
                               int64_t* lengths, void* stream) {
  if (B <= 0 || A <= 0 || Lmax <= 0) return fail("onehot_batch: empty batch");
  hipLaunchKernelGGL(k_onehot_batch, dim3(nblk((long)B * A * Lmax, 256)), dim3(256), 0, (hipStream_t)stream, ids,
                     offsets, B, A, Lmax, onehots, lengths);
  HIP_OK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// encoder backward (TRAIN_SEQUENCE_ENCODER: True; reference ProtNote.py:248-256 + autograd through
// protein_encoders.py:8-118).  Gradients live on valid positions only: every conv input and output is masked, so a
// padded position's gradient can reach neither a parameter nor a valid position (BN statistics see du = 0 there).
// ------------------------------------------------------------------------------------------------
struct EncBwdWs {
  float *g, *T1, *T2, *cs, *p, *q, *WbT, *WtA, *dWpk, *part;
  double *S1, *S2, *col;
  size_t part_floats;
  StatScr statscr;
};
static const long ENC_STATS_ROWS = 1024;

static bool enc_bwd_carve(const pn_encoder* e, int B, int L, Bump& bp, EncBwdWs& w) {
  const long P = (long)B * L;
  const int ldc = ld4(e->C), ldb = ld4(e->Cb), ldi = ld4(e->Cin);
  w.g = bp.take<float>(P * ldc);
  w.T1 = bp.take<float>(P * ldb);
  w.T2 = bp.take<float>(P * ldc);
  w.cs = bp.take<float>(ldc);
  w.p = bp.take<float>(ldc);
  w.q = bp.take<float>(ldc);
  w.WbT = bp.take<float>((size_t)ldb * ldc);
  w.WtA = bp.take<float>((size_t)e->C * e->ksize * ldb);
  const size_t pk_a = (size_t)ldb * e->ksize * ldc, pk_1 = (size_t)ldc * e->ksize * ldi, pk_b = (size_t)ldc * ldb;
  size_t pk = pk_a > pk_1 ? pk_a : pk_1;
  if (pk_b > pk) pk = pk_b;
  w.dWpk = bp.take<float>(pk);
  w.part_floats = (size_t)8 * ldc * ldc;
  w.part = bp.take<float>(w.part_floats);
  w.S1 = bp.take<double>(ldc);
  w.S2 = bp.take<double>(ldc);
  w.col = bp.take<double>(ldc);
  statscr_carve(bp, P, ENC_STATS_ROWS, ldc, w.statscr);
  return bp.ok;
}

extern "C" size_t pn_encoder_bwd_ws_bytes(const pn_encoder* enc, int B, int L) {
  Bump bp(nullptr, (size_t)-1);
  EncBwdWs w;
  enc_bwd_carve(enc, B, L, bp, w);
  return bp.off + 256;
}

extern "C" int pn_encoder_bwd(const pn_encoder* e, int B, int L, const float* demb, int ld_demb,
                              const pn_encoder_grads* gr, void* save, size_t save_bytes, void* ws, size_t ws_bytes,
                              void* stream) {
  PN_OK(math_field_check(e->math_mode, "pn_encoder_bwd"));
  MathScope math_scope(e->math_mode);
  hipStream_t st = (hipStream_t)stream;
  if (e->nblocks > PN_MAX_BLOCKS) return fail("encoder bwd: too many blocks");
  BnMode bn_mode(e->bn_use_running != 0);
  Bump bs(save, save_bytes), bw(ws, ws_bytes);
  EncSave sv;
  EncBwdWs w;
  if (!enc_save_carve(e, B, L, bs, sv)) return fail("encoder bwd: save buffer too small");
  if (!enc_bwd_carve(e, B, L, bw, w)) return fail("encoder bwd: workspace too small");
  const long P = (long)B * L;
  const int C = e->C, Cb = e->Cb, k = e->ksize;
  const int ldc = ld4(C), ldb = ld4(Cb), ldi = ld4(e->Cin);

  auto colsum_to = [&](const float* X, int ld, int cols, float* dst) -> int {  // bias gradient
    hipLaunchKernelGGL(k_colsum, dim3(nblk(cols, 256), nblk(P, 2048)), dim3(256), 0, st, X, (long)ld, P, cols, 2048L,
                       w.statscr.part);
    PN_OK(reduce_parts<double>(w.statscr.part, nblk(P, 2048), cols, cols, w.col, nullptr, nullptr, w.statscr.red, st));
    hipLaunchKernelGGL(k_d2f, dim3(nblk(cols, 256)), dim3(256), 0, st, (const double*)w.col, dst, cols, 1.f);
    HIP_OK(hipGetLastError());
    return 0;
  };
  // BN + ReLU backward of `G` (gradient wrt mask * relu(bn(Zin))) -> out = [addto +] mask_pad * dz
  auto bn_relu_bwd = [&](const float* Zin, int ld, int cols, const pn_bn& bn, const float* s, const float* t,
                         const float* mean, const float* invstd, const float* G, float* dgamma, float* dbeta,
                         float* out, const float* addto) -> int {
    HIP_OK(hipMemsetAsync(w.S1, 0, (size_t)ldc * sizeof(double), st));
    HIP_OK(hipMemsetAsync(w.S2, 0, (size_t)ldc * sizeof(double), st));
    HIP_OK(hipMemsetAsync(w.cs, 0, (size_t)ldc * sizeof(float), st));
    HIP_OK(hipMemsetAsync(w.p, 0, (size_t)ldc * sizeof(float), st));
    HIP_OK(hipMemsetAsync(w.q, 0, (size_t)ldc * sizeof(float), st));
    StatsParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.R = P; sp.C = ld; sp.rows_per_block = ENC_STATS_ROWS; sp.pairB = 1;
    sp.Z = Zin; sp.ldz = ld; sp.G = G; sp.ldg = ld; sp.s = s; sp.t = t; sp.mean = mean; sp.invstd = invstd;
    sp.part = w.statscr.part;
    hipLaunchKernelGGL((k_bn_bwd_stats<0, 0>), dim3(nblk(ld, 1024), nblk(P, ENC_STATS_ROWS)), dim3(256), 0, st, sp);
    PN_OK(reduce_parts<double>(w.statscr.part, nblk(P, ENC_STATS_ROWS), 2 * ld, ld, w.S1, w.S2, nullptr, w.statscr.red,
                               st));
    PN_OK(bwd_finalize(st, (const double*)w.S1,
                       (const double*)w.S2, (const double*)nullptr, (double)P, cols, bn.weight, s, mean, invstd,
                       (const float*)nullptr, w.cs, w.p, w.q, dgamma, dbeta, (float*)nullptr));
    DzParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.R = P; dp.C = ld; dp.rows_per_block = 512;
    dp.Z = Zin; dp.ldz = ld; dp.G = G; dp.ldg = ld; dp.s = s; dp.t = t; dp.cs = w.cs; dp.p = w.p; dp.q = w.q;
    dp.out = out; dp.ldo = ld; dp.lens = sv.lens32; dp.L = L; dp.addto = addto;
    hipLaunchKernelGGL((k_dz_apply<0>), dim3(nblk(ld, 1024), nblk(P, 512)), dim3(256), 0, st, dp);
    HIP_OK(hipGetLastError());
    return 0;
  };
  // weight gradient of one MaskedConv1D: dW[co][tap][ci] = sum_p dY[p][co] * in_act[p + shift(tap)][ci]
  auto conv_wgrad = [&](const float* dY, int ld_y, int Cout, const float* in, int ld_in, int Cin, int ntap, int dil,
                        const float* s, const float* t, float* dst_torch) -> int {
    for (int tap = 0; tap < ntap; ++tap) {
      TnParams tp = tn_zero();
      tp.R = P; tp.M = ld_y; tp.N = ld_in; tp.A = dY; tp.lda = ld_y;
      tp.B = in; tp.ldb = ld_in; tp.b_s = s; tp.b_t = t; tp.lens = sv.lens32; tp.L = L;
      tp.shift = (tap - ntap / 2) * dil;
      PN_OK((launch_tn<TA_PLAIN, TB_CONVTAP>(tp, w.dWpk + (size_t)tap * ld_in, (long)ntap * ld_in, w.part,
                                              w.part_floats, st)));
    }
    hipLaunchKernelGGL(k_unpack_conv_grad, dim3(nblk((long)Cout * Cin * ntap, 256)), dim3(256), 0, st,
                       (const float*)w.dWpk, Cout, Cin, ntap, ld_in, dst_torch);
    HIP_OK(hipGetLastError());
    return 0;
  };
  auto conv_nt = [&](const float* in, int ld_in, const float* wpk, int Cout, int ld_out, float* out, int ntap,
                     int dil) -> int {  // masked conv without bias / affine (data gradients)
    GemmParams p = gp_zero();
    p.M = (int)P; p.N = Cout; p.Nstore = ld_out; p.nseg = ntap; p.Kseg = ld_in;
    p.A = in; p.lda = ld_in; p.lens = sv.lens32; p.L = L; p.dil = dil;
    p.W = wpk; p.ldw = (long)ntap * ld_in; p.C = out; p.ldc = ld_out; p.ldr = ld_out;
    return launch_gemm<A_CONV, E_CONV>(p, pick_variant(ld_out), st);
  };

  // d(pool): gradient wrt the last block output
  hipLaunchKernelGGL(k_pool_bwd, dim3(nblk(P * ldc, 256)), dim3(256), 0, st, demb, ld_demb, (const int*)sv.lens32, L,
                     C, ldc, P, w.g);
  HIP_OK(hipGetLastError());

  int dil = 1;
  for (int i = 1; i < e->nblocks; ++i) dil *= e->dil_base;
  for (int i = e->nblocks - 1; i >= 0; --i) {
    const pn_res_block& bk = e->blk[i];
    const pn_res_block_grads& gb = gr->blk[i];
    // ---- masked_conv2 (1x1, Cb -> C): y = conv(b_act) + bias, X[i+1] = mask*y + X[i]; dy = g
    PN_OK(colsum_to(w.g, ldc, C, gb.conv_b_b));
    PN_OK(conv_wgrad(w.g, ldc, C, sv.Z[i], ldb, Cb, 1, 1, sv.s2[i], sv.t2[i], gb.conv_b_w));
    HIP_OK(hipMemsetAsync(w.WbT, 0, (size_t)ldb * ldc * sizeof(float), st));
    PN_OK(transpose_into(bk.conv_b_w, ldb, C, ldb, w.WbT, ldc, st));  // [Cb(pad)][ldc]
    PN_OK(conv_nt(w.g, ldc, w.WbT, Cb, ldb, w.T1, 1, 1));               // d b_act  [P][ldb]
    // ---- bn_activation_2 -> dz (masked conv_a output gradient), in place over T1
    PN_OK(bn_relu_bwd(sv.Z[i], ldb, Cb, bk.bn2, sv.s2[i], sv.t2[i], sv.m2[i], sv.i2[i], w.T1, gb.bn2_w, gb.bn2_b,
                      w.T1, nullptr));
    // ---- masked_conv1 (k taps, dilated, C -> Cb)
    PN_OK(colsum_to(w.T1, ldb, Cb, gb.conv_a_b));
    PN_OK(conv_wgrad(w.T1, ldb, Cb, sv.X[i], ldc, C, k, dil, sv.s1[i], sv.t1[i], gb.conv_a_w));
    hipLaunchKernelGGL(k_conv_w_dgrad, dim3(nblk((long)C * k * ldb, 256)), dim3(256), 0, st, bk.conv_a_w, Cb, C, k,
                       ldc, ldb, w.WtA);
    HIP_OK(hipGetLastError());
    PN_OK(conv_nt(w.T1, ldb, w.WtA, C, ldc, w.T2, k, dil));           // d a_act  [P][ldc]
    // ---- bn_activation_1 + residual: g <- g + mask * dz1
    PN_OK(bn_relu_bwd(sv.X[i], ldc, C, bk.bn1, sv.s1[i], sv.t1[i], sv.m1[i], sv.i1[i], w.T2, gb.bn1_w, gb.bn1_b, w.g,
                      w.g));
    dil /= e->dil_base;
  }
  // ---- conv1 (Cin -> C, no BN in front): dy = g
  PN_OK(colsum_to(w.g, ldc, C, gr->conv1_b));
  PN_OK(conv_wgrad(w.g, ldc, C, sv.x0, ldi, e->Cin, k, 1, nullptr, nullptr, gr->conv1_w));
  return 0;
}
