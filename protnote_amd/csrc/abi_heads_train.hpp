// Part of the translation unit protnote_hip.hip (#included there, after the launchers and small kernels; not a
// stand-alone header: it uses the static helpers defined above its #include):
// row MLPs and pair head: training forward + backward.
// ------------------------------------------------------------------------------------------------
// row MLP (W_p / W_l), train forward + backward
// ------------------------------------------------------------------------------------------------
struct MlpSave {
  float* Y[PN_MAX_LAYERS];
  float* H[PN_MAX_LAYERS];  // dropout > 0: the dropped activations relu(bn(Y_l)) * mask, materialised (small tensors)
  float *s[PN_MAX_LAYERS], *t[PN_MAX_LAYERS], *mean[PN_MAX_LAYERS], *invstd[PN_MAX_LAYERS];
};

static bool mlp_save_carve(const pn_mlp* m, int rows, Bump& bp, MlpSave& s) {
  for (int l = 0; l + 1 < m->nlayers; ++l) {
    const int h = m->dims[l + 1];
    s.Y[l] = bp.take<float>((size_t)rows * h);
    s.H[l] = m->dropout_p > 0.f ? bp.take<float>((size_t)rows * h) : nullptr;
    s.s[l] = bp.take<float>(h);
    s.t[l] = bp.take<float>(h);
    s.mean[l] = bp.take<float>(h);
    s.invstd[l] = bp.take<float>(h);
  }
  return bp.ok;
}

struct MlpTrainWs {
  double *S1, *S2;  // also forward column sum / sumsq
  float *cs, *p, *q, *G[2], *WT, *part;
  size_t part_floats;
  ColScr colscr;
  StatScr statscr;
  int* tnsync;  // pacing counters of the big weight-gradient kernel (launch_tn_fast)
};
static const long MLP_STATS_ROWS = 1024;

static bool mlp_train_ws_carve(const pn_mlp* m, int rows, Bump& bp, MlpTrainWs& w) {
  int hmax = 0;
  size_t wmax = 0;
  for (int l = 0; l < m->nlayers; ++l) {
    if (l + 1 < m->nlayers && m->dims[l + 1] > hmax) hmax = m->dims[l + 1];
    const size_t e = (size_t)m->dims[l] * m->dims[l + 1];
    if (e > wmax) wmax = e;
  }
  if (m->dropout_p > 0.f && m->dims[m->nlayers] > hmax) hmax = m->dims[m->nlayers];  // dy * mask scratch
  if (hmax == 0) hmax = 4;
  w.S1 = bp.take<double>(hmax);
  w.S2 = bp.take<double>(hmax);
  w.cs = bp.take<float>(hmax);
  w.p = bp.take<float>(hmax);
  w.q = bp.take<float>(hmax);
  w.G[0] = bp.take<float>((size_t)rows * hmax);
  w.G[1] = bp.take<float>((size_t)rows * hmax);
  w.WT = bp.take<float>(wmax);
  w.part_floats = wmax * 8 < TN_PART_FLOATS_MAX ? wmax * 8 : TN_PART_FLOATS_MAX;
  w.part = bp.take<float>(w.part_floats);
  colscr_carve(bp, rows, hmax, w.colscr);
  statscr_carve(bp, rows, MLP_STATS_ROWS, hmax, w.statscr);
  w.tnsync = bp.take<int>(TN_SYNC_INTS);
  return bp.ok;
}

extern "C" size_t pn_mlp_rows_train_save_bytes(const pn_mlp* m, int rows) {
  Bump bp(nullptr, (size_t)-1);
  MlpSave s;
  mlp_save_carve(m, rows, bp, s);
  return bp.off + 256;
}

extern "C" size_t pn_mlp_rows_train_ws_bytes(const pn_mlp* m, int rows) {
  Bump bp(nullptr, (size_t)-1);
  MlpTrainWs w;
  mlp_train_ws_carve(m, rows, bp, w);
  return bp.off + 256;
}

static int mlp_check(const pn_mlp* m, int ldx) {
  if (m->nlayers < 1 || m->nlayers > PN_MAX_LAYERS) return fail("mlp: bad layer count %d", m->nlayers);
  if (m->dropout_p < 0.f || m->dropout_p >= 1.f) return fail("mlp: dropout_p %g outside [0, 1)", m->dropout_p);
  for (int i = 0; i <= m->nlayers; ++i)
    if (m->dims[i] % 4 != 0) return fail("mlp: dims[%d]=%d not a multiple of 4", i, m->dims[i]);
  if (ldx % 4 != 0) return fail("mlp: ldx %% 4 != 0");
  for (int i = 0; i + 1 < m->nlayers; ++i) {
    if (m->bn[i].weight == nullptr) return fail("mlp train: layer %d has no BatchNorm (unsupported)", i);
    if (m->bias[i] != nullptr) return fail("mlp train: Linear bias with BatchNorm unsupported");
  }
  if (m->bias[m->nlayers - 1] != nullptr) return fail("mlp train: bias on the last Linear unsupported");
  return 0;
}

extern "C" int pn_mlp_rows_fwd_train(const pn_mlp* m, const float* x, int ldx, int rows, float* y, void* save,
                                     size_t save_bytes, void* ws, size_t ws_bytes, void* stream) {
  PN_OK(math_field_check(m->math_mode, "pn_mlp_rows_fwd_train"));
  MathScope math_scope(m->math_mode);
  hipStream_t st = (hipStream_t)stream;
  PN_OK(mlp_check(m, ldx));
  BnMode bn_mode(m->bn_use_running != 0);
  Bump bs(save, save_bytes), bw(ws, ws_bytes);
  MlpSave sv;
  MlpTrainWs w;
  if (!mlp_save_carve(m, rows, bs, sv)) return fail("mlp train: save buffer too small");
  if (!mlp_train_ws_carve(m, rows, bw, w)) return fail("mlp train: workspace too small");
  const float* in = x;
  long ldin = ldx;
  const bool drop = m->dropout_p > 0.f;
  bool in_is_act = false;  // `in` already holds activations (dropout path) instead of pre-activations
  for (int l = 0; l < m->nlayers; ++l) {
    const bool last = (l + 1 == m->nlayers);
    const int N = m->dims[l + 1];
    GemmParams p = gp_zero();
    p.M = rows; p.N = N; p.Nstore = N; p.Kseg = m->dims[l];
    p.A = in; p.lda = ldin; p.W = m->w[l]; p.ldw = m->dims[l];
    p.C = last ? y : sv.Y[l]; p.ldc = N;
    if (!last) {
      p.col_sum = w.S1; p.col_sumsq = w.S2; p.col_part = w.colscr.part; p.col_red = w.colscr.red;
    }
    if (l == 0 || in_is_act) {
      PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, pick_variant(N), st)));
    } else {
      p.a_scale = sv.s[l - 1]; p.a_shift = sv.t[l - 1];
      PN_OK((launch_gemm<A_AFFINE_RELU, E_STORE>(p, pick_variant(N), st)));
    }
    if (!last) {
      PN_OK(fold_train(st, m->bn[l], (const double*)w.S1,
                         (const double*)w.S2, (double)rows, m->bn_eps, m->bn_momentum, N, N, sv.s[l], sv.t[l],
                         sv.mean[l], sv.invstd[l]));
      HIP_OK(hipGetLastError());
      if (drop) {  // H_l = relu(bn(Y_l)) * mask / (1 - p), the next layer's plain operand
        PN_OK(launch_dropout<1>(sv.Y[l], N, sv.H[l], N, rows, N, sv.s[l], sv.t[l],
                                drop_spec(m->dropout_p, m->dropout_seed, m->dropout_stream + l), st));
        in = sv.H[l];
        in_is_act = true;
      } else {
        in = sv.Y[l];
      }
      ldin = N;
    } else if (drop) {  // Dropout after the last Linear (torchvision.ops.MLP)
      PN_OK(launch_dropout<0>(y, N, y, N, rows, N, nullptr, nullptr,
                              drop_spec(m->dropout_p, m->dropout_seed, m->dropout_stream + DROP_STREAM_OUT), st));
    }
  }
  return 0;
}

// pn_set_mlp_materialize(0): the row-MLP backward regenerates dY in the operand loaders for every row count
// (the path the small-size oracle tests pin) - an A/B switch for tests and measurements
static std::atomic<int> g_mlp_mat{1};
extern "C" int pn_set_mlp_materialize(int on) {
  g_mlp_mat = on ? 1 : 0;
  return 0;
}

extern "C" int pn_mlp_rows_bwd(const pn_mlp* m, const float* x, int ldx, int rows, const float* dy,
                               const pn_mlp_grads* gr, float* dx, void* save, size_t save_bytes, void* ws,
                               size_t ws_bytes, void* stream) {
  PN_OK(math_field_check(m->math_mode, "pn_mlp_rows_bwd"));
  MathScope math_scope(m->math_mode);
  hipStream_t st = (hipStream_t)stream;
  PN_OK(mlp_check(m, ldx));
  BnMode bn_mode(m->bn_use_running != 0);
  Bump bs(save, save_bytes), bw(ws, ws_bytes);
  MlpSave sv;
  MlpTrainWs w;
  if (!mlp_save_carve(m, rows, bs, sv)) return fail("mlp bwd: save buffer too small");
  if (!mlp_train_ws_carve(m, rows, bw, w)) return fail("mlp bwd: workspace too small");
  const int n = m->nlayers;
  const float* G = dy;  // gradient wrt the OUTPUT of layer l's Linear ... see below
  long ldg = m->dims[n];
  int gsel = 0;
  const bool drop = m->dropout_p > 0.f;
  if (drop) {  // Dropout after the last Linear: dY = dy * mask (into scratch: dy is the caller's)
    PN_OK(launch_dropout<0>(dy, ldg, w.G[gsel], ldg, rows, m->dims[n], nullptr, nullptr,
                            drop_spec(m->dropout_p, m->dropout_seed, m->dropout_stream + DROP_STREAM_OUT), st));
    G = w.G[gsel];
    gsel ^= 1;
  }
  for (int l = n - 1; l >= 0; --l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    const bool last = (l == n - 1);
    // For the last layer G = dY (plain).  For hidden layers G = d(relu(bn(Y_l))) and dY_l is generated.
    if (!last && drop)  // G is the gradient wrt the DROPPED activation: through the mask first (in place, our scratch)
      PN_OK(launch_dropout<0>(G, ldg, const_cast<float*>(G), ldg, rows, N, nullptr, nullptr,
                              drop_spec(m->dropout_p, m->dropout_seed, m->dropout_stream + l), st));
    if (!last) {
      StatsParams sp;
      memset(&sp, 0, sizeof(sp));
      sp.R = rows; sp.C = N; sp.rows_per_block = MLP_STATS_ROWS;
      sp.Z = sv.Y[l]; sp.ldz = N; sp.G = G; sp.ldg = ldg;
      sp.s = sv.s[l]; sp.t = sv.t[l]; sp.mean = sv.mean[l]; sp.invstd = sv.invstd[l];
      sp.part = w.statscr.part; sp.pairB = 1;
      hipLaunchKernelGGL((k_bn_bwd_stats<0, 0>), dim3(nblk(N, 1024), nblk(rows, MLP_STATS_ROWS)), dim3(256), 0, st, sp);
      PN_OK(reduce_parts<double>(w.statscr.part, nblk(rows, MLP_STATS_ROWS), 2 * N, N, w.S1, w.S2, nullptr,
                                 w.statscr.red, st));
      PN_OK(bwd_finalize(st, (const double*)w.S1,
                         (const double*)w.S2, (const double*)nullptr, (double)rows, N, m->bn[l].weight,
                         (const float*)sv.s[l], (const float*)sv.mean[l], (const float*)sv.invstd[l],
                         (const float*)nullptr, w.cs, w.p, w.q, gr->dgamma[l], gr->dbeta[l], (float*)nullptr));
      HIP_OK(hipGetLastError());
    }
    // Big row counts (W_l over the label table): dY_l is materialised once, in place over the incoming gradient (our
    // scratch), and both GEMMs take it as a plain operand - the 256-tile LDS-DMA NT kernel and the big TN tiles - instead of
    // regenerating it in the operand loaders of the 128-tile engine (0.70 of peak).  Same dz arithmetic (k_dz_apply).
    const bool mat = !last && g_mlp_mat && rows >= g_dma_min_rows && cur_math() == 0 && use_f32_dma();
    if (mat) {
      DzParams dp;
      memset(&dp, 0, sizeof(dp));
      dp.R = rows; dp.C = N; dp.rows_per_block = 512;
      dp.Z = sv.Y[l]; dp.ldz = N; dp.G = G; dp.ldg = ldg; dp.s = sv.s[l]; dp.t = sv.t[l]; dp.cs = w.cs; dp.p = w.p; dp.q = w.q;
      dp.out = const_cast<float*>(G); dp.ldo = ldg;
      hipLaunchKernelGGL((k_dz_apply<0>), dim3(nblk(N, 1024), nblk(rows, 512)), dim3(256), 0, st, dp);
      HIP_OK(hipGetLastError());
    }
    // dW_l[N][K] = dY_l^T X_l   (gr->dw[l] == NULL: frozen weight, no gradient GEMM; the data gradient still flows)
    TnParams tp = tn_zero();
    tp.R = rows; tp.M = N; tp.N = K;
    if (last || mat) {
      tp.A = G; tp.lda = ldg;
    } else {
      tp.A = sv.Y[l]; tp.lda = N; tp.G = G; tp.ldg = ldg;
      tp.m_s = sv.s[l]; tp.m_t = sv.t[l]; tp.m_cs = w.cs; tp.m_p = w.p; tp.m_q = w.q;
    }
    if (gr->dw[l] == nullptr) {
    } else if (l == 0 || drop) {  // plain B operand: the input rows, or the materialised dropped activation H_{l-1}
      tp.B = l == 0 ? x : sv.H[l - 1]; tp.ldb = l == 0 ? ldx : K;
      if (last || mat) PN_OK((launch_tn<TA_PLAIN, TB_PLAIN>(tp, gr->dw[l], K, w.part, w.part_floats, st)));
      else PN_OK((launch_tn<TA_DZ_ELEM, TB_PLAIN>(tp, gr->dw[l], K, w.part, w.part_floats, st)));
    } else {
      tp.B = sv.Y[l - 1]; tp.ldb = K; tp.b_s = sv.s[l - 1]; tp.b_t = sv.t[l - 1]; tp.task_sync = w.tnsync;
      if (last || mat) PN_OK((launch_tn<TA_PLAIN, TB_AFFINE_RELU>(tp, gr->dw[l], K, w.part, w.part_floats, st)));
      else PN_OK((launch_tn<TA_DZ_ELEM, TB_AFFINE_RELU>(tp, gr->dw[l], K, w.part, w.part_floats, st)));
    }
    // dX_l[rows][K] = dY_l W_l   (NT engine against W_l^T); nothing below this layer wants a gradient -> done
    bool below = dx != nullptr;
    for (int k = 0; k < l; ++k) below = below || gr->dw[k] || gr->dgamma[k] || gr->dbeta[k];
    if (!below) break;
    if (l > 0 || dx != nullptr) {
      PN_OK(transpose_into(m->w[l], K, N, K, w.WT, N, st));  // WT[K][N]
      GemmParams p = gp_zero();
      p.M = rows; p.N = K; p.Nstore = K; p.Kseg = N;
      p.W = w.WT; p.ldw = N;
      float* out = (l == 0) ? dx : w.G[gsel];
      p.C = out; p.ldc = K;
      if (last || mat) {
        p.A = G; p.lda = ldg;
        PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, pick_variant(K), st)));
      } else {
        p.A = sv.Y[l]; p.lda = N; p.A2 = G; p.lda2 = ldg;
        p.a_scale = sv.s[l]; p.a_shift = sv.t[l]; p.dz_cs = w.cs; p.dz_p = w.p; p.dz_q = w.q;
        PN_OK((launch_gemm<A_DZ_ELEM, E_STORE>(p, pick_variant(K), st)));
      }
      G = out;
      ldg = K;
      gsel ^= 1;
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// pair head, train forward + backward
// ------------------------------------------------------------------------------------------------
struct PairSave {
  float *A1, *B1, *Ap, *Bp;
  float *s[PN_MAX_LAYERS], *t[PN_MAX_LAYERS], *mean[PN_MAX_LAYERS], *invstd[PN_MAX_LAYERS];
  float* zbuf[PN_MAX_LAYERS];  // l >= 1: (R + S) rows x h; z_l lives at row offset S
  float* dq;  // concatenation_prod with ONE hidden layer: dQ [R][d] (deeper heads put it over the then-dead z1 buffer)
};

static bool pair_save_carve(const pn_pairhead* hd, int B, int NL, long S, Bump& bp, PairSave& s) {
  const int h = hd->h;
  const long R = (long)B * NL;
  s.A1 = bp.take<float>((size_t)B * h);
  s.B1 = bp.take<float>((size_t)NL * h);
  s.Ap = bp.take<float>((size_t)B * h);
  s.Bp = bp.take<float>((size_t)NL * h);
  for (int l = 0; l < hd->nlayers; ++l) {
    s.s[l] = bp.take<float>(h);
    s.t[l] = bp.take<float>(h);
    s.mean[l] = bp.take<float>(h);
    s.invstd[l] = bp.take<float>(h);
  }
  s.zbuf[0] = hd->fusion == 2 ? bp.take<float>((size_t)(R + S) * h) : nullptr;  // concatenation_prod stores z1 too
  for (int l = 1; l < hd->nlayers; ++l) s.zbuf[l] = bp.take<float>((size_t)(R + S) * h);
  s.dq = (hd->fusion == 2 && hd->nlayers == 1) ? bp.take<float>((size_t)R * hd->d) : nullptr;
  return bp.ok;
}

struct PairTrainWs {
  double *sumA, *sqA, *sumB, *sqB, *S1, *S2, *dwacc, *scal, *s12;
  float *cs, *p, *q, *WT, *weff, *dweff, *part, *dA1, *dB1;
  float* m1part;  // B <= 256: per-label-chunk partials of M1 (k_pair_mask_reduce_fused), [m1_chunks][B][h]
  int m1_chunks;
  float* dwpart;  // one hidden layer: partial rows of dw_out, [m1_chunks][h] (B <= 256) or [NL][h]
  long dwpart_rows;
  size_t part_floats;
  ColScr colscr;
  StatScr statscr;
  uint16_t* wsplit;  // bf16x3 mode: hi / lo planes of the weight operand of the current pair-grid GEMM
  int* tnsync;       // pacing counters of the big weight-gradient kernel (launch_tn_fast)
  uint16_t* hbf;     // forward_math = bf16: one chunk (FWD_H_ROWS pair rows) of the activation operand as bf16
};
static const long PAIR_STATS_ROWS = 4096;
static const int SUM_BLOCKS = 1024;

static bool pair_train_ws_carve(const pn_pairhead* hd, int B, int NL, Bump& bp, PairTrainWs& w) {
  const int h = hd->h, d = hd->d;
  w.sumA = bp.take<double>(h);
  w.sqA = bp.take<double>(h);
  w.sumB = bp.take<double>(h);
  w.sqB = bp.take<double>(h);
  w.S1 = bp.take<double>(h);
  w.S2 = bp.take<double>(h);
  w.dwacc = bp.take<double>(h);
  w.scal = bp.take<double>(4 + SUM_BLOCKS);  // [0] result, [4..) per-workgroup partials of k_sum
  w.s12 = bp.take<double>(2 * (size_t)h);    // SYNC_BN: the separable first layer's S1 | S2 on their way to the all-reduce
  w.cs = bp.take<float>(h);
  w.p = bp.take<float>(h);
  w.q = bp.take<float>(h);
  const size_t wt = (size_t)h * (h > 2 * d ? h : 2 * d);
  w.WT = bp.take<float>(wt);
  w.weff = hd->fusion == 1 ? bp.take<float>((size_t)h * 2 * d) : nullptr;
  w.dweff = hd->fusion == 1 ? bp.take<float>((size_t)h * 2 * d) : nullptr;
  w.part_floats = (size_t)16 * h * h < TN_PART_FLOATS_MAX ? (size_t)16 * h * h : TN_PART_FLOATS_MAX;
  w.part = bp.take<float>(w.part_floats);
  w.dA1 = bp.take<float>((size_t)B * h);
  w.dB1 = bp.take<float>((size_t)NL * h);
  // label chunks of k_pair_mask_reduce_fused: about 256 labels each (the f32 run length of the two-pass kernels), at most
  // 128 chunks (the partial buffer is chunks x B x h floats: 0.4 GB at the bench size)
  w.m1_chunks = (NL + 255) / 256;
  if (w.m1_chunks < 1) w.m1_chunks = 1;
  if (w.m1_chunks > 128) w.m1_chunks = 128;
  w.m1part = (B <= 256 && hd->fusion != 2) ? bp.take<float>((size_t)w.m1_chunks * B * h) : nullptr;
  w.dwpart_rows = w.m1part != nullptr ? w.m1_chunks : NL;
  w.dwpart = (hd->nlayers == 1 && hd->fusion != 2) ? bp.take<float>((size_t)w.dwpart_rows * h) : nullptr;
  colscr_carve(bp, (long)B * NL, h, w.colscr);
  statscr_carve(bp, (long)B * NL, PAIR_STATS_ROWS, h, w.statscr);
  w.wsplit = (uint16_t*)bp.take<float>((size_t)h * h);
  w.tnsync = bp.take<int>(TN_SYNC_INTS);
  w.hbf = nullptr;  // (last, and by the descriptor alone: forward and backward carve the same layout)
  if (fwd_bf16_requested(hd) && fwd_staged_shape(h) && hd->nlayers > 1) {
    const long R = (long)B * NL;
    w.hbf = (uint16_t*)bp.take<float>((size_t)(R < FWD_H_ROWS ? R : FWD_H_ROWS) * h / 2);
  }
  return bp.ok;
}

static long pair_chunk_rows(int B, int NL, int label_chunk) {
  long c = label_chunk <= 0 ? 256 : label_chunk;
  if (c > NL) c = NL;
  return c * (long)B;
}

extern "C" size_t pn_pairhead_train_save_bytes(const pn_pairhead* hd, int B, int NL, int label_chunk) {
  Bump bp(nullptr, (size_t)-1);
  PairSave s;
  pair_save_carve(hd, B, NL, pair_chunk_rows(B, NL, label_chunk), bp, s);
  return bp.off + 256;
}

extern "C" size_t pn_pairhead_train_ws_bytes(const pn_pairhead* hd, int B, int NL) {
  Bump bp(nullptr, (size_t)-1);
  PairTrainWs w;
  pair_train_ws_carve(hd, B, NL, bp, w);
  return bp.off + 256;
}

static int pair_check(const pn_pairhead* hd, int B, int NL) {
  if (hd->nlayers < 1 || hd->nlayers > PN_MAX_LAYERS) return fail("pairhead: nlayers=%d unsupported (need 1..%d)", hd->nlayers, PN_MAX_LAYERS);
  if (hd->fusion < 0 || hd->fusion > 2) return fail("pairhead: fusion %d not implemented", hd->fusion);
  if (hd->d % 4 || hd->h % 4) return fail("pairhead: d and h must be multiples of 4");
  if ((long)B * NL > 0x7fffffffL) return fail("pairhead: pair grid too large");
  if (hd->dropout_p < 0.f || hd->dropout_p >= 1.f) return fail("pairhead: dropout_p %g outside [0, 1)", hd->dropout_p);
  for (int l = 0; l < hd->nlayers; ++l) {
    if (hd->bn[l].weight != nullptr && hd->bias[l] != nullptr)
      return fail("pairhead: Linear bias together with BatchNorm is not supported");
  }
  return 0;
}

__global__ void k_diff_weight_grad(const float* __restrict__ dweff, float* __restrict__ dw, int h, int d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)h * d) return;
  const int n = (int)(i / d), k = (int)(i - (long)n * d);
  const float ga = dweff[(long)n * 2 * d + k], gb = dweff[(long)n * 2 * d + d + k];
  float* row = dw + (long)n * 3 * d;
  row[k] = ga;
  row[d + k] = gb;
  row[2 * d + k] = ga - gb;
}

extern "C" int pn_pairhead_fwd_train(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                                     float* logits_pairs, int label_chunk, void* save, size_t save_bytes, void* ws,
                                     size_t ws_bytes, void* stream) {
  PN_OK(math_field_check(hd->math_mode, "pn_pairhead_fwd_train"));
  MathScope math_scope(hd->math_mode);
  bool fwd_bf16 = false;
  PN_OK(fwd_math_of(hd, &fwd_bf16));
  hipStream_t st = (hipStream_t)stream;
  PN_OK(pair_check(hd, B, NL));
  BnMode bn_mode(hd->bn_use_running != 0);
  const int h = hd->h, d = hd->d, n = hd->nlayers;
  const long R = (long)B * NL, S = pair_chunk_rows(B, NL, label_chunk);
  Bump bs(save, save_bytes), bw(ws, ws_bytes);
  PairSave sv;
  PairTrainWs w;
  if (!pair_save_carve(hd, B, NL, S, bs, sv)) return fail("pairhead train: save buffer too small");
  // OUTPUT_MLP_BATCHNORM: False -> layer l is Linear(bias) + ReLU: s = 1, t = bias, no statistics
  auto fold_nobn = [&](int l) {
    hipLaunchKernelGGL(k_fold_nobn, dim3(nblk(h, 256)), dim3(256), 0, st, hd->bias[l], h, h, sv.s[l], sv.t[l], sv.mean[l],
                       sv.invstd[l]);
  };
  if (!pair_train_ws_carve(hd, B, NL, bw, w)) return fail("pairhead train: workspace too small");

  const float* w1 = hd->w[0];
  long ldw1 = hd->in_dim;
  if (hd->fusion == 1) {
    hipLaunchKernelGGL(k_diff_weight, dim3(nblk((long)h * 2 * d, 256)), dim3(256), 0, st, hd->w[0], w.weff, h, d);
    w1 = w.weff;
    ldw1 = 2 * d;
  }
  const bool prod = hd->fusion == 2;
  if (prod) ldw1 = hd->in_dim;
  {
    GemmParams p = gp_zero();
    p.M = B; p.N = h; p.Nstore = h; p.Kseg = d;
    p.A = P_e; p.lda = d; p.W = w1; p.ldw = ldw1; p.C = sv.A1; p.ldc = h;
    p.col_sum = w.sumA; p.col_sumsq = w.sqA; p.col_part = w.colscr.part; p.col_red = w.colscr.red;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
    p.M = NL; p.A = L_e; p.W = w1 + d; p.C = sv.B1; p.col_sum = w.sumB; p.col_sumsq = w.sqB;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
  }
  if (!prod) {
    if (hd->bn[0].weight == nullptr) fold_nobn(0);
    else if (sync_bn_on() || tl_bn_running) {
      // (eval-mode BatchNorm in a differentiable forward: fold_train takes the running statistics and ignores the sums)
      // SYNC_BN: this rank's grid sums (sum = NL sumA + B sumB, sumsq = NL sqA + 2 sumA sumB + B sqB), added over the
      // ranks, folded like any other BatchNorm over world * B * NL rows (the ranks' tables differ, so the global grid
      // is not a product grid and the var_i(A) + var_j(Bm) shortcut does not apply)
      hipLaunchKernelGGL(k_pair_grid_sums, dim3(nblk(h, 256)), dim3(256), 0, st, (const double*)w.sumA,
                         (const double*)w.sqA, (double)B, (const double*)w.sumB, (const double*)w.sqB, (double)NL, h, w.S1,
                         w.S2);
      PN_OK(fold_train(st, hd->bn[0], (const double*)w.S1, (const double*)w.S2, (double)B * (double)NL, hd->bn_eps,
                       hd->bn_momentum, h, h, sv.s[0], sv.t[0], sv.mean[0], sv.invstd[0]));
    } else
    hipLaunchKernelGGL(k_bn_fold_pair, dim3(nblk(h, 256)), dim3(256), 0, st, hd->bn[0], (const double*)w.sumA,
                       (const double*)w.sqA, (double)B, (const double*)w.sumB, (const double*)w.sqB, (double)NL,
                       hd->bn_eps, hd->bn_momentum, h, sv.s[0], sv.t[0], sv.mean[0], sv.invstd[0]);
    hipLaunchKernelGGL(k_affine_rows, dim3(nblk((long)B * h, 256)), dim3(256), 0, st, (const float*)sv.A1, (long)h,
                       sv.Ap, (long)h, (long)B, h, (const float*)sv.s[0], (const float*)sv.t[0]);
    hipLaunchKernelGGL(k_affine_rows, dim3(nblk((long)NL * h, 256)), dim3(256), 0, st, (const float*)sv.B1, (long)h,
                       sv.Bp, (long)h, (long)NL, h, (const float*)sv.s[0], (const float*)nullptr);
    HIP_OK(hipGetLastError());
  } else {
    // concatenation_prod: z1 = A1[i] + B1[j] + (P_e[i] (.) L_e[j]) W1c^T is not separable -> one more pair GEMM
    // whose output is stored, with BatchNorm statistics taken directly over the grid
    GemmParams p = gp_zero();
    p.M = (int)R; p.N = h; p.Nstore = h; p.Kseg = d;
    p.A = P_e; p.lda = d; p.A2 = L_e; p.lda2 = d; p.pairB = B;
    p.W = hd->w[0] + 2 * d; p.ldw = hd->in_dim;
    p.padd1 = sv.A1; p.ldp1 = h; p.padd2 = sv.B1; p.ldp2 = h;
    p.C = sv.zbuf[0] + (size_t)S * h; p.ldc = h; p.col_sum = w.S1; p.col_sumsq = w.S2;
    p.col_part = w.colscr.part; p.col_red = w.colscr.red;
    PN_OK((launch_gemm<A_PAIRPROD, E_PAIRADD>(p, 0, st)));
    if (hd->bn[0].weight == nullptr) fold_nobn(0);
    else
    PN_OK(fold_train(st, hd->bn[0], (const double*)w.S1,
                       (const double*)w.S2, (double)R, hd->bn_eps, hd->bn_momentum, h, h, sv.s[0], sv.t[0],
                       sv.mean[0], sv.invstd[0]));
    HIP_OK(hipGetLastError());
  }

  for (int l = 1; l < n; ++l) {
    float* z = sv.zbuf[l] + (size_t)S * h;
    GemmParams p = gp_zero();
    p.M = (int)R; p.N = h; p.Nstore = h; p.Kseg = h;
    p.W = hd->w[l]; p.ldw = h; p.C = z; p.ldc = h; p.col_sum = w.S1; p.col_sumsq = w.S2;
    p.col_part = w.colscr.part; p.col_red = w.colscr.red; p.wsplit = w.wsplit;
    {  // the input h_{l-1} of this layer went through Dropout (get_mlp: after every hidden ReLU but the last)
      const DropSpec ds = drop_spec(hd->dropout_p, hd->dropout_seed, DROP_STREAM_PAIR + (l - 1));
      p.drop_seed = ds.seed; p.drop_thresh = ds.thresh; p.drop_scale = ds.scale;
    }
    if (fwd_staged_on(fwd_bf16, h) && w.hbf != nullptr) {
      // AMP-class forward, materialised operand (fwd_bf16_h.hpp): per chunk of FWD_H_ROWS pair rows h_{l-1} is written once
      // as bf16 and z_l = h_{l-1} W_l^T runs all-DMA; every chunk leaves its BatchNorm column partials in its own slots of
      // the partial buffer (row tiles numbered across the chunks) and ONE fixed-order reduction adds them
      hipLaunchKernelGGL(k_round_plane, dim3(nblk((long)h * h / 4, 256)), dim3(256), 0, st, hd->w[l], (long)h, h, h, w.wsplit);
      long tiles = 0;
      for (long r0 = 0; r0 < R; r0 += FWD_H_ROWS) {
        const long rows = R - r0 < FWD_H_ROWS ? R - r0 : FWD_H_ROWS;
        if (l == 1 && !prod)
          PN_OK(make_h(0, r0, rows, h, sv.Ap, h, sv.Bp, h, B, nullptr, nullptr, w.hbf, st));
        else
          PN_OK(make_h(1, r0, rows, h, sv.zbuf[l - 1] + (size_t)S * h, h, nullptr, 0, 1, sv.s[l - 1], sv.t[l - 1], w.hbf, st));
        GemmParams q = gp_zero();
        q.M = (int)rows; q.N = h; q.Nstore = h; q.Kseg = h;
        q.A = (const float*)w.hbf; q.lda = h / 2; q.w_hi = w.wsplit;
        q.C = z + (size_t)r0 * h; q.ldc = h;
        q.col_part = w.colscr.part + (size_t)tiles * 2 * h;
        PN_OK((launch_gemm_h16<E_STORE>(q, (l == 1 && !prod) ? 2 : 1, st)));
        tiles += (rows + 255) / 256;
      }
      PN_OK(reduce_parts<float>(w.colscr.part, tiles, 2 * h, h, w.S1, w.S2, nullptr, w.colscr.red, st));
    } else {
      FwdBf16Scope fwd_scope(fwd_bf16);
      if (l == 1 && !prod) {
        p.A = sv.Ap; p.lda = h; p.A2 = sv.Bp; p.lda2 = h; p.pairB = B;
        PN_OK((launch_gemm<A_PAIRSUM_RELU, E_STORE>(p, 0, st)));
      } else {
        p.A = sv.zbuf[l - 1] + (size_t)S * h; p.lda = h; p.a_scale = sv.s[l - 1]; p.a_shift = sv.t[l - 1];
        PN_OK((launch_gemm<A_AFFINE_RELU, E_STORE>(p, 0, st)));
      }
    }
    if (hd->bn[l].weight == nullptr) fold_nobn(l);
    else
    PN_OK(fold_train(st, hd->bn[l], (const double*)w.S1,
                       (const double*)w.S2, (double)R, hd->bn_eps, hd->bn_momentum, h, h, sv.s[l], sv.t[l],
                       sv.mean[l], sv.invstd[l]));
    HIP_OK(hipGetLastError());
  }
  if (n == 1 && !prod) {
    // OUTPUT_MLP_NUM_LAYERS: 1: the separable layer is the only hidden layer - its BatchNorm statistics came in closed form
    // from the two tables above, and the logits are one fused pair-sum -> ReLU -> row-dot pass (no pair-grid GEMM, nothing
    // stored over the grid)
    ProfScope ps(ST_PAIR1_FWD, 3.0 * (double)R * (double)h, st);
    hipLaunchKernelGGL(k_pairsum_rowdot, dim3(nblk(B, 64), nblk(NL, 64)), dim3(256), 0, st, (const float*)sv.Ap, (long)h,
                       (const float*)sv.Bp, (long)h, B, NL, h, hd->w_out, hd->b_out, logits_pairs);
    HIP_OK(hipGetLastError());
    return 0;
  }
  if (h <= 3072) {
    const int rpw = 32;  // rows per wave: 128 rows (1.5 MB) per workgroup
    ProfScope ps(ST_ROWDOT, (double)R * (4.0 * h + 4.0), st);
    hipLaunchKernelGGL(k_rowdot_rows_reg, dim3(nblk(R, 4 * rpw)), dim3(256), 0, st,
                       (const float*)(sv.zbuf[n - 1] + (size_t)S * h), (long)h, R, h, (const float*)sv.s[n - 1],
                       (const float*)sv.t[n - 1], hd->w_out, hd->b_out, logits_pairs, rpw);
  } else
  hipLaunchKernelGGL(k_rowdot_rows, dim3(nblk(R, 4)), dim3(256), 0, st,
                     (const float*)(sv.zbuf[n - 1] + (size_t)S * h), (long)h, R, h, (const float*)sv.s[n - 1],
                     (const float*)sv.t[n - 1], hd->w_out, hd->b_out, logits_pairs);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_pairhead_bwd(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                               const float* dl_pairs, const pn_pairhead_grads* gr, float* dP_e, float* dL_e,
                               int label_chunk, void* save, size_t save_bytes, void* ws, size_t ws_bytes,
                               void* stream) {
  PN_OK(math_field_check(hd->math_mode, "pn_pairhead_bwd"));
  MathScope math_scope(hd->math_mode);
  hipStream_t st = (hipStream_t)stream;
  PN_OK(pair_check(hd, B, NL));
  BnMode bn_mode(hd->bn_use_running != 0);
  if (NL > 65535 || B > 65535)  // the layer-1 reductions put one label / protein per gridDim.y entry
    return fail("pairhead bwd: at most 65535 labels and 65535 proteins per step (got %d x %d)", B, NL);
  const int h = hd->h, d = hd->d, n = hd->nlayers;
  const long R = (long)B * NL, S = pair_chunk_rows(B, NL, label_chunk);
  Bump bs(save, save_bytes), bw(ws, ws_bytes);
  PairSave sv;
  PairTrainWs w;
  if (!pair_save_carve(hd, B, NL, S, bs, sv)) return fail("pairhead bwd: save buffer too small");
  if (!pair_train_ws_carve(hd, B, NL, bw, w)) return fail("pairhead bwd: workspace too small");

  // d b_out = sum_r dl[r]
  // (a NULL destination in `gr` = that parameter is frozen, e.g. TRAIN_PROJECTION_HEAD: False freezes output_layer.*,
  //  ProtNoteTrainer.py:221-222: its gradient is not computed - for a weight that is one whole pair-grid TN GEMM less -
  //  while the data gradient dh still flows through the layer)
  if (gr->db_out != nullptr) {
    hipLaunchKernelGGL(k_sum, dim3(SUM_BLOCKS), dim3(256), 0, st, dl_pairs, R, w.scal + 4);
    hipLaunchKernelGGL(k_scalar_final, dim3(1), dim3(256), 0, st, (const double*)(w.scal + 4), SUM_BLOCKS, w.scal);
    hipLaunchKernelGGL(k_d2f, dim3(1), dim3(64), 0, st, (const double*)w.scal, gr->db_out, 1, 1.f);
    HIP_OK(hipGetLastError());
  }

  const float* G = nullptr;  // gradient wrt relu(bn(z_l)) for the layer being processed (rows [0,R) of a zbuf)
  const long stats_rows = PAIR_STATS_ROWS;
  // pn_set_backward_math(1): the two pair-grid GEMMs of every hidden layer below run on one bf16 product (the dropped
  // layers keep the f32 kernels that carry the mask code)
  if (hd->backward_math < 0 || hd->backward_math > 2)
    return fail("pairhead bwd: backward_math %d (0 = library default, 1 = as the forward, 2 = bf16)", hd->backward_math);
  const int bwd_math = hd->backward_math == 0 ? g_bwd_math.load(std::memory_order_relaxed) : hd->backward_math - 1;
  BwdBf16Scope bwd_scope(bwd_math == 1 && hd->dropout_p == 0.f);
  for (int l = n - 1; l >= 1; --l) {
    const bool top = (l == n - 1);
    float* z = sv.zbuf[l] + (size_t)S * h;
    StatsParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.R = R; sp.C = h; sp.rows_per_block = stats_rows; sp.pairB = 1;
    sp.Z = z; sp.ldz = h;
    sp.s = sv.s[l]; sp.t = sv.t[l]; sp.mean = sv.mean[l]; sp.invstd = sv.invstd[l];
    sp.part = w.statscr.part;
    const dim3 sg(nblk(h, 1024), nblk(R, stats_rows));
    if (top) {
      sp.gvec = dl_pairs; sp.w = hd->w_out;
      {
        ProfScope ps(ST_BN_BWD_STATS, (double)R * (4.0 * h + 4.0), st);  // reads z (+ one dl per row)
        hipLaunchKernelGGL((k_bn_bwd_stats<1, 0>), sg, dim3(256), 0, st, sp);
      }
      PN_OK(reduce_parts<double>(w.statscr.part, sg.y, 3 * h, h, w.S1, w.S2, w.dwacc, w.statscr.red, st));
    } else {
      sp.G = G; sp.ldg = h;
      {
        ProfScope ps(ST_BN_BWD_STATS, (double)R * 8.0 * h, st);  // reads z and the incoming gradient
        hipLaunchKernelGGL((k_bn_bwd_stats<0, 0>), sg, dim3(256), 0, st, sp);
      }
      PN_OK(reduce_parts<double>(w.statscr.part, sg.y, 2 * h, h, w.S1, w.S2, nullptr, w.statscr.red, st));
    }
    PN_OK(bwd_finalize(st, (const double*)w.S1,
                       (const double*)w.S2, (const double*)(top ? w.dwacc : nullptr), (double)R, h,
                       hd->bn[l].weight, (const float*)sv.s[l], (const float*)sv.mean[l],
                       (const float*)sv.invstd[l], top ? hd->w_out : (const float*)nullptr, w.cs, w.p, w.q,
                       gr->dgamma[l], gr->dbeta[l], top ? gr->dw_out : (float*)nullptr));
    HIP_OK(hipGetLastError());

    // dz_l materialised once, in place: over z_l itself for the top layer (its upstream gradient is the rank-1
    // dl * w_out), over the incoming gradient buffer for inner layers.
    // bf16 backward with pn_set_bwd_deep bit 2: dz is written ROUNDED, into the first half of each of those rows
    // (bwd_bf16_dz.hpp), when both GEMMs below can take it that way: whole 32-row slabs, hidden width a multiple of 256, and -
    // for the layer whose activation is the pair sum - a slab inside one label.
    const bool dz_bf16 = tl_bwd_bf16 && (g_bwd_deep & 4) && h % 256 == 0 && h <= 4096 && R % 32 == 0 && R >= 65536 &&
                         (l != 1 || hd->fusion == 2 || B % 32 == 0) && (long)32 * h * 4 < (1L << 31);
    float* dz;
    {
      DzParams dp;
      memset(&dp, 0, sizeof(dp));
      dp.R = R; dp.C = h; dp.rows_per_block = dz_bf16 ? 64 : 512;
      dp.Z = z; dp.ldz = h; dp.s = sv.s[l]; dp.t = sv.t[l]; dp.cs = w.cs; dp.p = w.p; dp.q = w.q; dp.ldo = h;
      const dim3 dg(nblk(h, 1024), nblk(R, 512));
      const dim3 dgb(1, nblk(R, 64));
      if (top) {
        dp.gvec = dl_pairs; dp.out = z; dz = z;
        ProfScope ps(ST_DZ_APPLY, (double)R * ((dz_bf16 ? 6.0 : 8.0) * h + 4.0), st);  // z read, dz written over it
        if (dz_bf16) hipLaunchKernelGGL((k_dz_apply_bf16<1, 4>), dgb, dim3(h / 4), 0, st, dp);
        else hipLaunchKernelGGL((k_dz_apply<1>), dg, dim3(256), 0, st, dp);
      } else {
        dp.G = G; dp.ldg = h; dp.out = const_cast<float*>(G); dz = const_cast<float*>(G);
        ProfScope ps(ST_DZ_APPLY, (double)R * (dz_bf16 ? 10.0 : 12.0) * h, st);  // z and G read, dz written over G
        if (dz_bf16) hipLaunchKernelGGL((k_dz_apply_bf16<0, 4>), dgb, dim3(h / 4), 0, st, dp);
        else hipLaunchKernelGGL((k_dz_apply<0>), dg, dim3(256), 0, st, dp);
      }
      HIP_OK(hipGetLastError());
    }
    tl_dz_bf16 = dz_bf16;  // read by the launchers of the two GEMMs below; cleared at the end of the layer

    // dW_l = dz_l^T h_{l-1}   (h_{l-1} = dropped activation: the B loader regenerates the mask)
    TnParams tp = tn_zero();
    tp.R = R; tp.M = h; tp.N = h;
    tp.A = dz; tp.lda = h;
    const DropSpec ds_in = drop_spec(hd->dropout_p, hd->dropout_seed, DROP_STREAM_PAIR + (l - 1));
    tp.drop_seed = ds_in.seed; tp.drop_thresh = ds_in.thresh; tp.drop_scale = ds_in.scale;
    if (gr->dw[l] == nullptr) {
      // frozen weight: no dW GEMM
    } else if (l == 1 && hd->fusion != 2) {
      tp.B = sv.Ap; tp.ldb = h; tp.B2 = sv.Bp; tp.ldb2 = h; tp.pairB = B;
      PN_OK((launch_tn<TA_PLAIN, TB_PAIRSUM_RELU>(tp, gr->dw[l], h, w.part, w.part_floats, st)));
    } else {
      tp.B = sv.zbuf[l - 1] + (size_t)S * h; tp.ldb = h; tp.b_s = sv.s[l - 1]; tp.b_t = sv.t[l - 1];
      tp.task_sync = w.tnsync;
      PN_OK((launch_tn<TA_PLAIN, TB_AFFINE_RELU>(tp, gr->dw[l], h, w.part, w.part_floats, st)));
    }

    // dh_{l-1} = dz_l W_l.  Inner layers: dz_l sits in the other buffer, so the result goes straight over the
    // (now dead) z_l.  Top layer: dz_l sits in z_l's own buffer at row offset S, so the result is written chunk
    // by chunk over the part already consumed (ring with one chunk of slack).
    PN_OK(transpose_into(hd->w[l], h, h, h, w.WT, h, st));
    const long step = top ? S : R;
    for (long r0 = 0; r0 < R; r0 += step) {
      const long rows = (R - r0 < step) ? R - r0 : step;
      GemmParams p = gp_zero();
      p.M = (int)rows; p.N = h; p.Nstore = h; p.Kseg = h;
      p.A = dz + (size_t)r0 * h; p.lda = h;
      p.W = w.WT; p.ldw = h; p.wsplit = w.wsplit;
      p.C = sv.zbuf[l] + (size_t)r0 * h; p.ldc = h;
      PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
    }
    tl_dz_bf16 = false;
    G = sv.zbuf[l];
    if (hd->dropout_p > 0.f)  // G = gradient wrt the DROPPED h_{l-1}: through the mask (one streaming pass, in place)
      PN_OK(launch_dropout<0>(G, h, sv.zbuf[l], h, R, h, nullptr, nullptr, ds_in, st));
  }

  // ---- layer 0: upstream gradient G = dh_0 over the pair grid ----
  tl_bwd_bf16 = false;  // (the scope object restores the caller's value on return)
  const bool prod = hd->fusion == 2;
  float* dQ = nullptr;  // concatenation_prod: gradient wrt the P (.) L block, [R][d]
  if (!prod) {
    // separable: z1[i,j] = A1[i] + B1[j] is regenerated, never stored.  Two passes over G give M0 = sum_i du and
    // M1 = sum_j du; the statistics, dgamma / dbeta and both table gradients follow from them (train_kernels.hpp)
    PairRedParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.B = B; rp.NL = NL; rp.C = h; rp.DH = G; rp.ldh = h;
    rp.A = sv.A1; rp.lda = h; rp.Bm = sv.B1; rp.ldb = h;
    rp.s = sv.s[0]; rp.t = sv.t[0];
    rp.out = w.dB1; rp.ldo = h;
    if (n == 1) {
      // OUTPUT_MLP_NUM_LAYERS: 1: the separable layer is the top layer.  Its upstream gradient is the rank-1 dl[r] * w_out[c],
      // generated inside the same masked reductions (nothing [R][h] exists in this configuration); the same pass leaves the
      // partial rows of dw_out[c] = sum_r dl[r] relu(bn(z1))[r][c]
      rp.DH = nullptr; rp.gvec = dl_pairs; rp.w = hd->w_out; rp.dwpart = w.dwpart;
      long dw_rows;
      if (w.m1part != nullptr) {
        const int per = (NL + w.m1_chunks - 1) / w.m1_chunks;
        const int nch = (NL + per - 1) / per;
        {
          ProfScope ps(ST_PAIR1_BWD, 8.0 * (double)R * (double)h, st);
          hipLaunchKernelGGL((k_pair_mask_reduce_fused<true>), dim3(nblk(h, 128), nch), dim3(PMR_IG * 32), 0, st, rp, w.m1part,
                             per);
        }
        hipLaunchKernelGGL(k_pair_m1_reduce, dim3(nblk((long)B * h / 4, 256)), dim3(256), 0, st, (const float*)w.m1part, nch,
                           (long)B * h, h, w.dA1, (long)h);
        dw_rows = nch;
      } else {
        hipLaunchKernelGGL((k_pair_mask_reduce<0, true>), dim3(nblk(h, 1024), NL), dim3(256), 0, st, rp);
        rp.out = w.dA1;
        hipLaunchKernelGGL((k_pair_mask_reduce<1, true>), dim3(nblk(h, 1024), B), dim3(256), 0, st, rp);
        dw_rows = NL;
      }
      if (gr->dw_out != nullptr)
        hipLaunchKernelGGL(k_colsum_rows, dim3(nblk(h, 256)), dim3(256), 0, st, (const float*)w.dwpart, dw_rows, h, gr->dw_out);
      HIP_OK(hipGetLastError());
    } else if (w.m1part != nullptr) {  // B <= 256: both tables from one pass over the gradient
      const int per = (NL + w.m1_chunks - 1) / w.m1_chunks;
      const int nch = (NL + per - 1) / per;
      {
        ProfScope ps(ST_PAIR_MASK_REDUCE, (double)R * 4.0 * h, st);  // one read of the 101 GB gradient
        hipLaunchKernelGGL((k_pair_mask_reduce_fused<false>), dim3(nblk(h, 128), nch), dim3(PMR_IG * 32), 0, st, rp, w.m1part, per);
      }
      hipLaunchKernelGGL(k_pair_m1_reduce, dim3(nblk((long)B * h / 4, 256)), dim3(256), 0, st, (const float*)w.m1part, nch,
                         (long)B * h, h, w.dA1, (long)h);
    } else {
      hipLaunchKernelGGL((k_pair_mask_reduce<0>), dim3(nblk(h, 1024), NL), dim3(256), 0, st, rp);
      rp.out = w.dA1;
      hipLaunchKernelGGL((k_pair_mask_reduce<1>), dim3(nblk(h, 1024), B), dim3(256), 0, st, rp);
    }
    const int per_chunk = (NL + RED_CHUNKS - 1) / RED_CHUNKS;
    const int nchunk = (NL + per_chunk - 1) / per_chunk;
    hipLaunchKernelGGL(k_pair_colsums, dim3(nblk(h, 256), nchunk), dim3(256), 0, st, (const float*)w.dB1, (long)h,
                       (const float*)sv.B1, (long)h, NL, h, per_chunk, w.statscr.red);
    hipLaunchKernelGGL(k_pair_bn0_finalize, dim3(nblk(h, 256)), dim3(256), 0, st, (const double*)w.statscr.red, nchunk,
                       (const float*)sv.A1, (long)h, (const float*)w.dA1, (long)h, B, NL, h, hd->bn[0].weight,
                       (const float*)sv.s[0], (const float*)sv.mean[0], (const float*)sv.invstd[0], w.cs, w.p, w.q,
                       gr->dgamma[0], gr->dbeta[0], w.S1, w.S2, sync_bn_on() ? w.s12 : (double*)nullptr,
                       tl_bn_running ? 1 : 0);
    if (sync_bn_on() && hd->bn[0].weight != nullptr && !tl_bn_running) {  // global S1 / S2 -> cs, p, q (dgamma / dbeta stay local)
      double* s12 = w.s12;  // workspace, not the staging buffer: sync_sum2 stages through that itself
      const double* gcount = nullptr;
      PN_OK(sync_sum2(s12, s12 + h, h, (double)B * (double)NL, &gcount, st));
      hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(nblk(h, 256)), dim3(256), 0, st, (const double*)s12, (const double*)(s12 + h),
                         (const double*)nullptr, (double)B * (double)NL, gcount, h, hd->bn[0].weight,
                         (const float*)sv.s[0], (const float*)sv.mean[0], (const float*)sv.invstd[0], (const float*)nullptr,
                         w.cs, w.p, w.q, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0);
      HIP_OK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_pair_apply, dim3(nblk((long)NL * h, 256)), dim3(256), 0, st, w.dB1, (long)h,
                       (const float*)sv.B1, (long)h, (long)NL, h, (const float*)w.cs, (const float*)w.p,
                       (const float*)w.q, (const double*)w.S1, (double)B);
    hipLaunchKernelGGL(k_pair_apply, dim3(nblk((long)B * h, 256)), dim3(256), 0, st, w.dA1, (long)h,
                       (const float*)sv.A1, (long)h, (long)B, h, (const float*)w.cs, (const float*)w.p,
                       (const float*)w.q, (const double*)w.S2, (double)NL);
    HIP_OK(hipGetLastError());
  } else {
    // concatenation_prod: z1 is stored; dz1 is materialised over G, then summed / contracted
    float* z0 = sv.zbuf[0] + (size_t)S * h;
    const bool top0 = (n == 1);  // one hidden layer: layer 0 is the top layer, its upstream gradient is dl (x) w_out
    StatsParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.R = R; sp.C = h; sp.rows_per_block = stats_rows; sp.pairB = 1;
    sp.Z = z0; sp.ldz = h;
    if (top0) { sp.gvec = dl_pairs; sp.w = hd->w_out; }
    else { sp.G = G; sp.ldg = h; }
    sp.s = sv.s[0]; sp.t = sv.t[0]; sp.mean = sv.mean[0]; sp.invstd = sv.invstd[0];
    sp.part = w.statscr.part;
    if (top0) {
      hipLaunchKernelGGL((k_bn_bwd_stats<1, 0>), dim3(nblk(h, 1024), nblk(R, stats_rows)), dim3(256), 0, st, sp);
      PN_OK(reduce_parts<double>(w.statscr.part, nblk(R, stats_rows), 3 * h, h, w.S1, w.S2, w.dwacc, w.statscr.red, st));
    } else {
      hipLaunchKernelGGL((k_bn_bwd_stats<0, 0>), dim3(nblk(h, 1024), nblk(R, stats_rows)), dim3(256), 0, st, sp);
      PN_OK(reduce_parts<double>(w.statscr.part, nblk(R, stats_rows), 2 * h, h, w.S1, w.S2, nullptr, w.statscr.red, st));
    }
    PN_OK(bwd_finalize(st, (const double*)w.S1,
                       (const double*)w.S2, (const double*)(top0 ? w.dwacc : nullptr), (double)R, h, hd->bn[0].weight,
                       (const float*)sv.s[0], (const float*)sv.mean[0], (const float*)sv.invstd[0],
                       top0 ? hd->w_out : (const float*)nullptr, w.cs, w.p, w.q, gr->dgamma[0], gr->dbeta[0],
                       top0 ? gr->dw_out : (float*)nullptr));
    DzParams dp;
    memset(&dp, 0, sizeof(dp));
    dp.R = R; dp.C = h; dp.rows_per_block = 512;
    dp.Z = z0; dp.ldz = h; dp.s = sv.s[0]; dp.t = sv.t[0]; dp.cs = w.cs; dp.p = w.p; dp.q = w.q;
    float* dz0 = top0 ? z0 : const_cast<float*>(G);  // top layer: dz over z1 itself (as the deeper heads' top layer)
    dp.out = dz0; dp.ldo = h;
    if (top0) {
      dp.gvec = dl_pairs;
      hipLaunchKernelGGL((k_dz_apply<1>), dim3(nblk(h, 1024), nblk(R, 512)), dim3(256), 0, st, dp);
    } else {
      dp.G = G; dp.ldg = h;
      hipLaunchKernelGGL((k_dz_apply<0>), dim3(nblk(h, 1024), nblk(R, 512)), dim3(256), 0, st, dp);
    }
    hipLaunchKernelGGL((k_pair_sum<0>), dim3(nblk(h, 1024), NL), dim3(256), 0, st, (const float*)dz0, (long)h, B, NL, h,
                       (const float*)nullptr, 0L, w.dB1, (long)h, 0);
    hipLaunchKernelGGL((k_pair_sum<1>), dim3(nblk(h, 1024), B), dim3(256), 0, st, (const float*)dz0, (long)h, B, NL, h,
                       (const float*)nullptr, 0L, w.dA1, (long)h, 0);
    HIP_OK(hipGetLastError());
    // dW1c[n][k] = sum_r dz1[r][n] * P_e[i][k] * L_e[j][k]
    if (gr->dw[0] != nullptr) {
      TnParams tp = tn_zero();
      tp.R = R; tp.M = h; tp.N = d; tp.A = dz0; tp.lda = h;
      tp.B = P_e; tp.ldb = d; tp.B2 = L_e; tp.ldb2 = d; tp.pairB = B;
      PN_OK((launch_tn<TA_PLAIN, TB_PAIRPROD>(tp, gr->dw[0] + 2 * d, hd->in_dim, w.part, w.part_floats, st)));
    }
    // dQ = dz1 W1c  ([R][d], over the dead z1 buffer; with one hidden layer dz1 lives IN that buffer -> its own block)
    dQ = top0 ? sv.dq : sv.zbuf[0];
    PN_OK(transpose_into(hd->w[0] + 2 * d, hd->in_dim, h, d, w.WT, h, st));  // WT[d][h]
    GemmParams p = gp_zero();
    p.M = (int)R; p.N = d; p.Nstore = d; p.Kseg = h;
    p.A = dz0; p.lda = h; p.W = w.WT; p.ldw = h; p.C = dQ; p.ldc = d;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
  }
  // dW_0: [h][in_dim];  concatenation: [dA1^T P_e | dB1^T L_e]
  float* dwa = hd->fusion == 1 ? w.dweff : gr->dw[0];
  const long ldd = hd->fusion == 1 ? 2 * d : hd->in_dim;
  if (gr->dw[0] != nullptr) {
    TnParams tp = tn_zero();
    tp.R = B; tp.M = h; tp.N = d; tp.A = w.dA1; tp.lda = h; tp.B = P_e; tp.ldb = d;
    PN_OK((launch_tn<TA_PLAIN, TB_PLAIN>(tp, dwa, ldd, w.part, w.part_floats, st)));
    tp.R = NL; tp.A = w.dB1; tp.B = L_e;
    PN_OK((launch_tn<TA_PLAIN, TB_PLAIN>(tp, dwa + d, ldd, w.part, w.part_floats, st)));
  }
  const float* w1 = hd->w[0];
  long ldw1 = hd->in_dim;
  if (hd->fusion == 1) {
    if (gr->dw[0] != nullptr)
      hipLaunchKernelGGL(k_diff_weight_grad, dim3(nblk((long)h * d, 256)), dim3(256), 0, st, (const float*)w.dweff,
                         gr->dw[0], h, d);
    hipLaunchKernelGGL(k_diff_weight, dim3(nblk((long)h * 2 * d, 256)), dim3(256), 0, st, hd->w[0], w.weff, h, d);
    HIP_OK(hipGetLastError());
    w1 = w.weff;
    ldw1 = 2 * d;
  }
  // dP_e = dA1 W1a, dL_e = dB1 W1b
  for (int side = 0; side < 2; ++side) {
    float* out = side == 0 ? dP_e : dL_e;
    if (out == nullptr) continue;
    PN_OK(transpose_into(w1 + (side ? d : 0), ldw1, h, d, w.WT, h, st));  // WT[d][h]
    GemmParams p = gp_zero();
    p.M = side == 0 ? B : NL; p.N = d; p.Nstore = d; p.Kseg = h;
    p.A = side == 0 ? w.dA1 : w.dB1; p.lda = h; p.W = w.WT; p.ldw = h; p.C = out; p.ldc = d;
    PN_OK((launch_gemm<A_PLAIN, E_STORE>(p, 0, st)));
  }
  if (prod) {
    // dP_e[i] += sum_j dQ[i,j] (.) L_e[j],   dL_e[j] += sum_i dQ[i,j] (.) P_e[i]
    if (dP_e) hipLaunchKernelGGL((k_pair_sum<1>), dim3(nblk(d, 1024), B), dim3(256), 0, st, (const float*)dQ, (long)d,
                                 B, NL, d, L_e, (long)d, dP_e, (long)d, 1);
    if (dL_e) hipLaunchKernelGGL((k_pair_sum<0>), dim3(nblk(d, 1024), NL), dim3(256), 0, st, (const float*)dQ, (long)d,
                                 B, NL, d, P_e, (long)d, dL_e, (long)d, 1);
    HIP_OK(hipGetLastError());
  }
  return 0;
}
