// Part of the translation unit protnote_hip.hip (#included there, after the launchers and small kernels; not a
// stand-alone header: it uses the static helpers defined above its #include):
// ProteInfer encoder entry points (pn_encoder_fwd / _fwd_ids / _fwd_train) and the stand-alone MaskedConv1D / Residual.
// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------
extern "C" int pn_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int k, void* stream) {
  const int ld = ld4(Cin);
  const long total = (long)Cout * k * ld;
  hipLaunchKernelGGL(k_pack_conv, dim3(nblk(total, 256)), dim3(256), 0, (hipStream_t)stream, w, packed, Cout,
                     Cin, k, ld);
  HIP_OK(hipGetLastError());
  return 0;
}

struct EncWs {
  int* lens32;
  float *x0, *xa, *xb, *z, *s1, *t1, *s2, *t2;
  double *sum_x, *sq_x, *sum_z, *sq_z;
  ColScr cs;  // per-tile partials of the train-mode BatchNorm statistics
  float *H, *Wr;  // LDS-DMA convolution path (gemm_conv_dma.hpp): staged activation with guard rows, re-laid weights
  StatScr st64;   // f64-accumulating convolutions (gemm_conv_f64.hpp): column statistics of their output
  signed char* ids;  // conv1 as a gather-sum over one-hot input: residue ids, the "not one-hot" flag, re-laid weights
  int* oh_flag;
  float* W1t;
};
static const int CONV1_GATHER_CS = 64;
static bool conv1_gather_shape(const pn_encoder* e) {  // the weight slice [ksize * Cin][64] must fit the LDS next to the scratch
  return e->ksize == 9 && e->ksize * (e->Cin + 1) <= 255;  // (weight-row indices are bytes; the kernel is built for k = 9)
}
static size_t conv1_gather_lds(const pn_encoder* e, int BM) {
  return ((size_t)e->ksize * (e->Cin + 1) * CONV1_GATHER_CS + 32 * 2 * CONV1_GATHER_CS) * sizeof(float) + 16 * (size_t)BM + 16;
}
static const long ENC_COLSTAT_ROWS = 512;

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// the all-DMA convolution kernel serves the wide layers of a big enough batch (conv1 with its 20 input channels and
// toy models keep the register-staged engine)
static bool conv_dma_shape(int ld_in, int ld_out, long P) { return PN_BIG && ld_in >= 256 && ld_out >= 512 && P >= 4096; }

static bool enc_carve(const pn_encoder* e, int B, int L, Bump& bp, EncWs& w) {
  const long P = (long)B * L;
  const int ldc = ld4(e->C), ldb = ld4(e->Cb), ldi = ld4(e->Cin);
  w.lens32 = bp.take<int>(B);
  w.x0 = bp.take<float>(P * ldi);
  w.xa = bp.take<float>(P * ldc);
  w.xb = bp.take<float>(P * ldc);
  w.z = bp.take<float>(P * ldb);
  w.s1 = bp.take<float>(ldc);
  w.t1 = bp.take<float>(ldc);
  w.s2 = bp.take<float>(ldb);
  w.t2 = bp.take<float>(ldb);
  w.sum_x = bp.take<double>(2 * (size_t)ldc);
  w.sq_x = w.sum_x ? w.sum_x + ldc : nullptr;
  w.sum_z = bp.take<double>(2 * (size_t)ldb);
  w.sq_z = w.sum_z ? w.sum_z + ldb : nullptr;
  colscr_carve(bp, P, ldc, w.cs);
  w.H = w.Wr = nullptr;
  statscr_carve(bp, P, ENC_COLSTAT_ROWS, ldc, w.st64);
  w.ids = (signed char*)bp.take<char>((size_t)P);
  w.oh_flag = bp.take<int>(64);
  w.W1t = bp.take<float>((size_t)e->ksize * e->Cin * ldc);
  {  // (staged operands: the all-DMA f32 kernels at the big shapes, the f64-accumulating kernels at every shape)
    long dil = 1;
    for (int i = 1; i < e->nblocks; ++i) dil *= e->dil_base;
    const long G = (long)(e->ksize / 2) * dil;  // widest guard band
    const size_t ha = (size_t)(G + (long)B * (L + G)) * round_up(ldc, 32), hb = (size_t)P * round_up(ldb, 32);
    const size_t wa = (size_t)round_up(e->Cb, 192) * e->ksize * round_up(ldc, 32), wb = (size_t)round_up(e->C, 192) * round_up(ldb, 32);
    w.H = bp.take<float>(ha > hb ? ha : hb);
    w.Wr = bp.take<float>(wa > wb ? wa : wb);
  }
  return bp.ok;
}

extern "C" size_t pn_encoder_ws_bytes(const pn_encoder* enc, int B, int L) {
  Bump bp(nullptr, (size_t)-1);
  EncWs w;
  enc_carve(enc, B, L, bp, w);
  return bp.off;
}

// activations and BN statistics kept from a training forward for the encoder backward (TRAIN_SEQUENCE_ENCODER)
struct EncSave {
  int* lens32;
  float* x0;                         // [P][ld4(Cin)] masked channels-last input
  float* X[PN_MAX_BLOCKS + 1];       // X[0] = conv1 output, X[i+1] = block i output, each [P][ld4(C)]
  float* Z[PN_MAX_BLOCKS];           // conv_a outputs [P][ld4(Cb)]
  float *s1[PN_MAX_BLOCKS], *t1[PN_MAX_BLOCKS], *m1[PN_MAX_BLOCKS], *i1[PN_MAX_BLOCKS];
  float *s2[PN_MAX_BLOCKS], *t2[PN_MAX_BLOCKS], *m2[PN_MAX_BLOCKS], *i2[PN_MAX_BLOCKS];
};

static bool enc_save_carve(const pn_encoder* e, int B, int L, Bump& bp, EncSave& sv) {
  const long P = (long)B * L;
  const int ldc = ld4(e->C), ldb = ld4(e->Cb), ldi = ld4(e->Cin);
  sv.lens32 = bp.take<int>(B);
  sv.x0 = bp.take<float>(P * ldi);
  for (int i = 0; i <= e->nblocks; ++i) sv.X[i] = bp.take<float>(P * ldc);
  for (int i = 0; i < e->nblocks; ++i) {
    sv.Z[i] = bp.take<float>(P * ldb);
    sv.s1[i] = bp.take<float>(ldc); sv.t1[i] = bp.take<float>(ldc);
    sv.m1[i] = bp.take<float>(ldc); sv.i1[i] = bp.take<float>(ldc);
    sv.s2[i] = bp.take<float>(ldb); sv.t2[i] = bp.take<float>(ldb);
    sv.m2[i] = bp.take<float>(ldb); sv.i2[i] = bp.take<float>(ldb);
  }
  return bp.ok;
}

extern "C" size_t pn_encoder_train_save_bytes(const pn_encoder* enc, int B, int L) {
  Bump bp(nullptr, (size_t)-1);
  EncSave sv;
  enc_save_carve(enc, B, L, bp, sv);
  return bp.off + 256;
}

// ragged residue ids (back to back, uint8) + offsets [B+1] -> padded ids [B*L] int8 (-1 = pad, or a residue outside the
// alphabet: an all-zero one-hot column) and int32 lengths: what k_onehot_ids derives from one-hots, without the one-hots
__global__ void k_ids_pad(const uint8_t* __restrict__ flat, const int64_t* __restrict__ offsets, int B, int L, int Cin,
                          signed char* __restrict__ ids, int* __restrict__ lens32) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)B * L) return;
  const int b = (int)(p / L), t = (int)(p - (long)b * L);
  const int64_t off = offsets[b];
  long len = (long)(offsets[b + 1] - off);
  if (len > L) len = L;
  int id = -1;
  if (t < len) {
    const int v = (int)flat[off + t];
    id = v < Cin ? v : -1;
  }
  ids[p] = (signed char)id;
  if (t == 0) lens32[b] = (int)len;
}

static int encoder_forward(const pn_encoder* e, const float* onehots, const int64_t* lens, int B, int L, float* emb,
                           int ld_emb, int training, EncWs& w, EncSave* sv, hipStream_t st,
                           const uint8_t* flat_ids = nullptr, const int64_t* id_offsets = nullptr) {
  const long P = (long)B * L;
  if (P > 0x7fffffffL) return fail("encoder: B*L too large");
  const int ldc = ld4(e->C), ldb = ld4(e->Cb), ldi = ld4(e->Cin);
  const float bn_eps = 1e-3f, bn_mom = 0.01f;  // protein_encoders.py:36,48
  int* lens32 = sv ? sv->lens32 : w.lens32;
  float* x0 = sv ? sv->x0 : w.x0;

  const bool from_ids = flat_ids != nullptr;  // pn_encoder_fwd_ids: conv1 is the gather-sum, no one-hot tensor exists
  if (from_ids) {
    if (sv != nullptr || !g_conv1_gather || w.ids == nullptr || !conv1_gather_shape(e))
      return fail("encoder (ids): needs the gather form of conv1 (kernel_size 9, alphabet <= 27, pn_set_conv1_gather on, "
                  "frozen encoder); pass one-hots to pn_encoder_fwd otherwise");
    hipLaunchKernelGGL(k_ids_pad, dim3(nblk(P, 256)), dim3(256), 0, st, flat_ids, id_offsets, B, L, e->Cin, w.ids, lens32);
  } else {
    hipLaunchKernelGGL(k_lens32, dim3(nblk(B, 256)), dim3(256), 0, st, lens, lens32, B);
  }
  HIP_OK(hipGetLastError());

  const int* conv_run_if = nullptr;  // set around conv1: the general kernel is a no-op while the flag is 0
  auto conv = [&](const float* in, int ld_in, const float* wpk, const float* bias, int Cout, int ld_out, float* out,
                  int ntap, int dil, const float* s, const float* t, const float* resid, double* csum,
                  double* csq) -> int {
    GemmParams p = gp_zero();
    p.M = (int)P; p.N = Cout; p.Nstore = ld_out; p.nseg = ntap; p.Kseg = ld_in;
    p.A = in; p.lda = ld_in; p.a_scale = s; p.a_shift = t; p.lens = lens32; p.L = L; p.dil = dil;
    p.W = wpk; p.ldw = (long)ntap * ld_in; p.C = out; p.ldc = ld_out; p.bias = bias; p.resid = resid; p.ldr = ld_out;
    p.col_sum = csum; p.col_sumsq = csq;
    if (csum) { p.col_part = w.cs.part; p.col_red = w.cs.red; }
    p.run_if = conv_run_if;
    if (sv != nullptr && s != nullptr && cur_math() == 0 && g_enc_f64 && w.H != nullptr) {
      // trainable encoder (pn_encoder_fwd_train): the two wide convolutions of a block accumulate in float64 so that
      // the stored pre-activations - and with them the ReLU masks the backward multiplies by - are the correctly
      // rounded ones (gemm_conv_f64.hpp).  conv1 (K = 9 x 20) keeps the f32 kernel.
      const int Kpad = round_up(ld_in, 32), G = (ntap / 2) * dil, Lp = L + G, Cpad = round_up(Cout, 64);
      hipLaunchKernelGGL(k_conv_relay_weight, dim3(nblk((long)Cpad * ntap * Kpad, 256)), dim3(256), 0, st, wpk, Cout, ntap,
                         ld_in, w.Wr, Cpad, Kpad);
      hipLaunchKernelGGL(k_conv_stage_act, dim3(nblk(((long)G + (long)B * Lp) * (Kpad / 4), 256)), dim3(256), 0, st, in,
                         (long)ld_in, s, t, (const int*)lens32, w.H, Kpad, B, L, Lp, G, ld_in);
      HIP_OK(hipGetLastError());
      ConvF64Params cp;
      cp.H = w.H + (long)G * Kpad; cp.ldh = Kpad; cp.Lp = Lp; cp.W = w.Wr; cp.ldw = (long)ntap * Kpad;
      cp.M = (int)P; cp.N = Cout; cp.Nstore = ld_out; cp.ntap = ntap; cp.Kpad = Kpad; cp.dil = dil; cp.L = L;
      cp.lens = lens32; cp.bias = bias; cp.resid = resid; cp.ldr = ld_out; cp.C = out; cp.ldc = ld_out;
      {
        ProfScope ps(32, 2.0 * (double)P * (double)Cout * (double)ntap * (double)ld_in, st);
        hipLaunchKernelGGL(gemm_conv_f64_kernel, dim3(nblk(P, 128) * nblk(ld_out, 64)), dim3(256), 0, st, cp);
      }
      HIP_OK(hipGetLastError());
      if (csum) {
        const unsigned nrb = nblk(P, ENC_COLSTAT_ROWS);
        hipLaunchKernelGGL(k_col_stats, dim3(nblk(Cout, 256), nrb), dim3(256), 0, st, (const float*)out, (long)ld_out, P,
                           Cout, ENC_COLSTAT_ROWS, w.st64.part);
        PN_OK(reduce_parts<double>(w.st64.part, nrb, 2 * Cout, Cout, csum, csq, nullptr, w.st64.red, st));
      }
      return 0;
    }
    const bool big = PN_BIG && ld_out >= 512 && P >= 4096;
    if (cur_math() == 0 && use_f32_dma() && s != nullptr && w.H != nullptr && conv_dma_shape(ld_in, ld_out, P)) {
      // f32 default: stage relu(bn(in)) once (masked, K padded to 32, guard rows between sequences), re-lay the weights,
      // then the all-LDS-DMA kernel - bit-identical to the register-staged tap gather below
      const int Kpad = round_up(ld_in, 32), G = (ntap / 2) * dil, Lp = L + G, Cpad = round_up(Cout, 192);
      hipLaunchKernelGGL(k_conv_relay_weight, dim3(nblk((long)Cpad * ntap * Kpad, 256)), dim3(256), 0, st, wpk, Cout, ntap,
                         ld_in, w.Wr, Cpad, Kpad);
      {  // reads the activation once, writes its staged image (guard rows and K padding included)
        ProfScope ps(ST_CONV_STAGE, 4.0 * ((double)P * ld_in + ((double)G + (double)B * Lp) * Kpad), st);
        hipLaunchKernelGGL(k_conv_stage_act, dim3(nblk(((long)G + (long)B * Lp) * (Kpad / 4), 256)), dim3(256), 0, st, in,
                           (long)ld_in, s, t, (const int*)lens32, w.H, Kpad, B, L, Lp, G, ld_in);
      }
      HIP_OK(hipGetLastError());
      p.A = w.H + (long)G * Kpad; p.lda = Kpad; p.a_scale = nullptr; p.a_shift = nullptr;
      p.W = w.Wr; p.ldw = (long)ntap * Kpad; p.Kseg = Kpad;
      return launch_conv_dma(p, Lp, ld_in, st);
    }
    return launch_gemm<A_CONV, E_CONV>(p, big ? 3 : pick_variant(ld_out), st);
  };

  // conv1: MaskedConv1D(Cin -> C, k, dil 1), no BN/ReLU in front (protein_encoders.py:84-91,110)
  float* x = sv ? sv->X[0] : w.xa;
  float* xn = w.xb;
  {  // K2: 4 B x Cin read + 4 B x C written per residue
    ProfScope ps(ST_CONV1, (double)P * 4.0 * (e->Cin + e->C), st);
    // One-hot input (what the reference's collator produces): a gather-sum, bit-identical to the general convolution
    // (gemm_conv_f64.hpp, k_conv1_gather); the general kernel is queued behind it and runs only if k_onehot_ids found a
    // residue that is not one-hot (the flag lives on the device: no host round trip).
    const bool gather = g_conv1_gather && w.ids != nullptr && conv1_gather_shape(e);
    if (gather) {
      const bool big = PN_BIG && ldc >= 512 && P >= 4096;
      const int BM = big ? 256 : 128;  // row-tile height of the general kernel's statistics partials
      const int tiles = P >= 65536 ? 4 : 1;
      const size_t lds = conv1_gather_lds(e, BM);
      static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
      int dev = 0;
      HIP_OK(hipGetDevice(&dev));
      if (dev < 64 && !attr_done[dev]) {
        HIP_OK(hipFuncSetAttribute((const void*)k_conv1_gather<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_done[dev] = true;
      }
      HIP_OK(hipMemsetAsync(w.oh_flag, 0, sizeof(int), st));
      if (!from_ids)
        hipLaunchKernelGGL(k_onehot_ids, dim3(nblk(P, 256)), dim3(256), 0, st, onehots, (const int*)lens32, w.ids, w.oh_flag, B,
                           e->Cin, L);
      hipLaunchKernelGGL(k_conv1_relay, dim3(nblk((long)e->ksize * e->Cin * ldc, 256)), dim3(256), 0, st, e->conv1_w, e->C,
                         e->ksize, e->Cin, ldi, w.W1t, ldc);
      hipLaunchKernelGGL(k_conv1_gather<9>, dim3(nblk(ldc, CONV1_GATHER_CS), nblk(nblk(P, BM), tiles)), dim3(512), lds, st,
                         (const signed char*)w.ids, (const int*)lens32, (const float*)w.W1t, e->conv1_b, x, (int)P, L, e->C, ldc,
                         e->Cin, (const int*)w.oh_flag, training ? w.cs.part : (float*)nullptr, BM, tiles);
      HIP_OK(hipGetLastError());
      conv_run_if = w.oh_flag;
    }
    if (!from_ids) {  // (from ids the input IS one-hot by construction: the general kernel has nothing to do)
      // channels-last copy of the input: operand of the general kernel, and of the conv1 weight gradient (sv)
      hipLaunchKernelGGL(k_ncl_to_nlc, dim3(nblk(P, 256)), dim3(256), 0, st, onehots, (const int*)lens32, x0, B, e->Cin,
                         L, ldi, (gather && sv == nullptr) ? (const int*)w.oh_flag : (const int*)nullptr);
      HIP_OK(hipGetLastError());
      PN_OK(conv(x0, ldi, e->conv1_w, e->conv1_b, e->C, ldc, x, e->ksize, 1, nullptr, nullptr, nullptr,
                 training ? w.sum_x : nullptr, training ? w.sq_x : nullptr));
    } else if (training) {  // column statistics of conv1's output from the gather kernel's per-tile partials (as conv() does)
      GemmParams p = gp_zero();
      p.M = (int)P; p.N = e->C; p.col_sum = w.sum_x; p.col_sumsq = w.sq_x; p.col_part = w.cs.part; p.col_red = w.cs.red;
      const bool big = PN_BIG && ldc >= 512 && P >= 4096;
      PN_OK(finish_col_stats(p, (P + (big ? 256 : 128) - 1) / (big ? 256 : 128), st));
    }
    conv_run_if = nullptr;
  }

  int dil = 1;
  for (int i = 0; i < e->nblocks; ++i) {
    const pn_res_block& bk = e->blk[i];
    float* s1 = sv ? sv->s1[i] : w.s1;
    float* t1 = sv ? sv->t1[i] : w.t1;
    float* s2 = sv ? sv->s2[i] : w.s2;
    float* t2 = sv ? sv->t2[i] : w.t2;
    float* z = sv ? sv->Z[i] : w.z;
    if (sv) xn = sv->X[i + 1];
    // bn_activation_1 folded into conv_a's operand load
    if (training) {
      PN_OK(fold_train(st, bk.bn1, (const double*)w.sum_x,
                         (const double*)w.sq_x, (double)P, bn_eps, bn_mom, e->C, ldc, s1, t1,
                         sv ? sv->m1[i] : (float*)nullptr, sv ? sv->i1[i] : (float*)nullptr));
    } else {
      hipLaunchKernelGGL(k_bn_fold_eval, dim3(nblk(ldc, 256)), dim3(256), 0, st, bk.bn1, (const float*)nullptr,
                         bn_eps, e->C, ldc, s1, t1);
    }
    PN_OK(conv(x, ldc, bk.conv_a_w, bk.conv_a_b, e->Cb, ldb, z, e->ksize, dil, s1, t1, nullptr,
               training ? w.sum_z : nullptr, training ? w.sq_z : nullptr));
    if (training) {
      PN_OK(fold_train(st, bk.bn2, (const double*)w.sum_z,
                         (const double*)w.sq_z, (double)P, bn_eps, bn_mom, e->Cb, ldb, s2, t2,
                         sv ? sv->m2[i] : (float*)nullptr, sv ? sv->i2[i] : (float*)nullptr));
    } else {
      hipLaunchKernelGGL(k_bn_fold_eval, dim3(nblk(ldb, 256)), dim3(256), 0, st, bk.bn2, (const float*)nullptr,
                         bn_eps, e->Cb, ldb, s2, t2);
    }
    const bool need_stats = training && (i + 1 < e->nblocks);
    PN_OK(conv(z, ldb, bk.conv_b_w, bk.conv_b_b, e->C, ldc, xn, 1, 1, s2, t2, x, need_stats ? w.sum_x : nullptr,
               need_stats ? w.sq_x : nullptr));
    if (sv) {
      x = xn;
    } else {
      float* tmp = x;
      x = xn;
      xn = tmp;
    }
    dil *= e->dil_base;
  }
  {  // K6: 4 B x C read per residue
    ProfScope ps(ST_POOL, (double)P * 4.0 * e->C, st);
    hipLaunchKernelGGL(k_pool, dim3(nblk(e->C, 256), B), dim3(256), 0, st, (const float*)x, (const int*)lens32, emb, L,
                       e->C, ldc, ld_emb);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_encoder_fwd(const pn_encoder* e, const float* onehots, const int64_t* lens, int B, int L,
                              float* emb, int ld_emb, int training, void* ws, size_t ws_bytes, void* stream) {
  PN_OK(math_field_check(e->math_mode, "pn_encoder_fwd"));
  MathScope math_scope(e->math_mode);
  if (e->nblocks > PN_MAX_BLOCKS) return fail("encoder: too many blocks (%d)", e->nblocks);
  if (B <= 0 || L <= 0) return fail("encoder: empty batch");
  Bump bp(ws, ws_bytes);
  EncWs w;
  if (!enc_carve(e, B, L, bp, w)) return fail("encoder: workspace too small (%zu given)", ws_bytes);
  return encoder_forward(e, onehots, lens, B, L, emb, ld_emb, training, w, nullptr, (hipStream_t)stream);
}

// The same forward from residue ids: `ids` = the batch's residue indices back to back (uint8), `offsets` [B+1] i64 - the input
// of pn_onehot_batch, i.e. what collate_to_device already holds on the device.  Equivalent to pn_onehot_batch followed by
// pn_encoder_fwd (sequences longer than L are cut to L; an id >= Cin is an all-zero column), bit for bit, without the
// [B][Cin][L] f32 one-hot tensor and the two passes that re-derive the ids from it.
extern "C" int pn_encoder_fwd_ids(const pn_encoder* e, const uint8_t* ids, const int64_t* offsets, int B, int L, float* emb,
                                  int ld_emb, int training, void* ws, size_t ws_bytes, void* stream) {
  PN_OK(math_field_check(e->math_mode, "pn_encoder_fwd_ids"));
  MathScope math_scope(e->math_mode);
  if (e->nblocks > PN_MAX_BLOCKS) return fail("encoder: too many blocks (%d)", e->nblocks);
  if (B <= 0 || L <= 0) return fail("encoder: empty batch");
  if (ids == nullptr || offsets == nullptr) return fail("encoder (ids): ids / offsets are NULL");
  Bump bp(ws, ws_bytes);
  EncWs w;
  if (!enc_carve(e, B, L, bp, w)) return fail("encoder: workspace too small (%zu given)", ws_bytes);
  return encoder_forward(e, nullptr, nullptr, B, L, emb, ld_emb, training, w, nullptr, (hipStream_t)stream, ids, offsets);
}

// training forward that keeps what the backward needs (block inputs, conv_a outputs, BN batch statistics)
extern "C" int pn_encoder_fwd_train(const pn_encoder* e, const float* onehots, const int64_t* lens, int B, int L,
                                    float* emb, int ld_emb, void* save, size_t save_bytes, void* ws, size_t ws_bytes,
                                    void* stream) {
  PN_OK(math_field_check(e->math_mode, "pn_encoder_fwd_train"));
  MathScope math_scope(e->math_mode);
  if (e->nblocks > PN_MAX_BLOCKS) return fail("encoder: too many blocks (%d)", e->nblocks);
  if (B <= 0 || L <= 0) return fail("encoder: empty batch");
  Bump bp(ws, ws_bytes), bs(save, save_bytes);
  EncWs w;
  EncSave sv;
  if (!enc_carve(e, B, L, bp, w)) return fail("encoder: workspace too small (%zu given)", ws_bytes);
  if (!enc_save_carve(e, B, L, bs, sv)) return fail("encoder: save buffer too small (%zu given)", save_bytes);
  BnMode bn_mode(e->bn_use_running != 0);
  return encoder_forward(e, onehots, lens, B, L, emb, ld_emb, 1, w, &sv, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// MaskedConv1D / Residual called stand-alone (protein_encoders.py:8-17, :23-67): the public classes under ProteInfer.
// ProteInfer itself never takes this route (pn_encoder_fwd fuses them and stays channels-last); these entry points keep the
// reference's [B][C][L] layout on both sides and its stand-alone semantics, which differ from the fused pipeline exactly
// where the input's PAD positions hold something: Residual normalises the RAW input (train-mode statistics include the pads)
// and adds it back unmasked, so its output carries the input's pad values.
// ------------------------------------------------------------------------------------------------
__global__ void k_fill_int(int* out, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}
// channels-last [B*L][ld] -> [B][C][L]; positions t >= len[b] take pad_src[b][c][t] (the raw input) or 0
__global__ void k_nlc_to_ncl(const float* __restrict__ y, int ld, const int* __restrict__ lens, const float* __restrict__ pad_src,
                             float* __restrict__ out, int B, int C, int L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * L) return;
  const int t = (int)(i % L);
  const int c = (int)((i / L) % C);
  const int b = (int)(i / ((long)L * C));
  out[i] = t < lens[b] ? y[((long)b * L + t) * ld + c] : (pad_src != nullptr ? pad_src[i] : 0.f);
}

struct PieceWs {
  int *lens32, *lens_full;
  float *xin, *z, *y, *s1, *t1, *s2, *t2;
  double *sum_a, *sq_a, *sum_b, *sq_b;
  ColScr cs;
  StatScr st;
};
static const long PIECE_STAT_ROWS = 256;
static bool piece_carve(int B, int L, int Ca, int Cb, Bump& bp, PieceWs& w) {
  const long P = (long)B * L;
  const int lda = ld4(Ca), ldb = ld4(Cb), ldm = lda > ldb ? lda : ldb;
  w.lens32 = bp.take<int>(B);
  w.lens_full = bp.take<int>(B);
  w.xin = bp.take<float>((size_t)P * lda);
  w.z = bp.take<float>((size_t)P * ldb);
  w.y = bp.take<float>((size_t)P * lda);
  w.s1 = bp.take<float>(lda); w.t1 = bp.take<float>(lda);
  w.s2 = bp.take<float>(ldb); w.t2 = bp.take<float>(ldb);
  w.sum_a = bp.take<double>(lda); w.sq_a = bp.take<double>(lda);
  w.sum_b = bp.take<double>(ldb); w.sq_b = bp.take<double>(ldb);
  colscr_carve(bp, P, ldm, w.cs);
  statscr_carve(bp, P, PIECE_STAT_ROWS, ldm, w.st);
  return bp.ok;
}
static int piece_conv(const float* in, int ld_in, const float* wpk, const float* bias, int Cout, int ld_out, float* out, int ntap,
                      int dil, const float* s, const float* t, const float* resid, double* csum, double* csq, const int* lens32,
                      int B, int L, PieceWs& w, hipStream_t st) {
  const long P = (long)B * L;
  GemmParams p = gp_zero();
  p.M = (int)P; p.N = Cout; p.Nstore = ld_out; p.nseg = ntap; p.Kseg = ld_in;
  p.A = in; p.lda = ld_in; p.a_scale = s; p.a_shift = t; p.lens = lens32; p.L = L; p.dil = dil;
  p.W = wpk; p.ldw = (long)ntap * ld_in; p.C = out; p.ldc = ld_out; p.bias = bias; p.resid = resid; p.ldr = ld_out;
  p.col_sum = csum; p.col_sumsq = csq;
  if (csum) { p.col_part = w.cs.part; p.col_red = w.cs.red; }
  const bool big = PN_BIG && ld_out >= 512 && P >= 4096;
  return launch_gemm<A_CONV, E_CONV>(p, big ? 3 : pick_variant(ld_out), st);
}

extern "C" size_t pn_masked_conv1d_ws_bytes(int B, int L, int Cin, int Cout) {
  Bump bp(nullptr, (size_t)-1);
  PieceWs w;
  piece_carve(B, L, Cin, Cout, bp, w);
  return bp.off;
}

extern "C" int pn_masked_conv1d_fwd(const float* x, const int64_t* lens, const float* w_packed, const float* bias, int B, int Cin,
                                    int Cout, int L, int ksize, int dilation, float* out, void* ws, size_t ws_bytes,
                                    void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B <= 0 || L <= 0 || Cin <= 0 || Cout <= 0) return fail("masked_conv1d: empty input");
  if (ksize < 1 || ksize % 2 != 1) return fail("masked_conv1d: kernel_size %d must be odd (padding='same')", ksize);
  if (dilation < 1) return fail("masked_conv1d: dilation %d", dilation);
  if ((long)B * L > 0x7fffffffL) return fail("masked_conv1d: B*L too large");
  Bump bp(ws, ws_bytes);
  PieceWs w;
  if (!piece_carve(B, L, Cin, Cout, bp, w)) return fail("masked_conv1d: workspace too small (%zu given)", ws_bytes);
  const long P = (long)B * L;
  const int ldi = ld4(Cin), ldo = ld4(Cout);
  hipLaunchKernelGGL(k_lens32, dim3(nblk(B, 256)), dim3(256), 0, st, lens, w.lens32, B);
  hipLaunchKernelGGL(k_ncl_to_nlc, dim3(nblk(P, 256)), dim3(256), 0, st, x, (const int*)w.lens32, w.xin, B, Cin, L, ldi,
                     (const int*)nullptr);  // masks the input (protein_encoders.py:14)
  HIP_OK(hipGetLastError());
  PN_OK(piece_conv(w.xin, ldi, w_packed, bias, Cout, ldo, w.z, ksize, dilation, nullptr, nullptr, nullptr, nullptr, nullptr,
                   w.lens32, B, L, w, st));
  hipLaunchKernelGGL(k_nlc_to_ncl, dim3(nblk((long)B * Cout * L, 256)), dim3(256), 0, st, (const float*)w.z, ldo,
                     (const int*)w.lens32, (const float*)nullptr, out, B, Cout, L);  // ... and the output (:16)
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" size_t pn_residual_ws_bytes(int B, int L, int C, int Cb) {
  Bump bp(nullptr, (size_t)-1);
  PieceWs w;
  piece_carve(B, L, C, Cb, bp, w);
  return bp.off;
}

extern "C" int pn_residual_fwd(const pn_res_block* blk, int C, int Cb, int ksize, int dilation, const float* x,
                               const int64_t* lens, int B, int L, float* out, int training, void* ws, size_t ws_bytes,
                               void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B <= 0 || L <= 0 || C <= 0 || Cb <= 0) return fail("residual: empty input");
  if (ksize < 1 || ksize % 2 != 1) return fail("residual: kernel_size %d must be odd (padding='same')", ksize);
  if (dilation < 1) return fail("residual: dilation %d", dilation);
  if ((long)B * L > 0x7fffffffL) return fail("residual: B*L too large");
  Bump bp(ws, ws_bytes);
  PieceWs w;
  if (!piece_carve(B, L, C, Cb, bp, w)) return fail("residual: workspace too small (%zu given)", ws_bytes);
  const long P = (long)B * L;
  const int ldc = ld4(C), ldb = ld4(Cb);
  const float bn_eps = 1e-3f, bn_mom = 0.01f;  // protein_encoders.py:36,48
  hipLaunchKernelGGL(k_lens32, dim3(nblk(B, 256)), dim3(256), 0, st, lens, w.lens32, B);
  hipLaunchKernelGGL(k_fill_int, dim3(nblk(B, 256)), dim3(256), 0, st, w.lens_full, B, L);
  // the RAW input, channels-last: bn_activation_1 sees it unmasked (:62), pads included
  hipLaunchKernelGGL(k_ncl_to_nlc, dim3(nblk(P, 256)), dim3(256), 0, st, x, (const int*)w.lens_full, w.xin, B, C, L, ldc,
                     (const int*)nullptr);
  HIP_OK(hipGetLastError());
  if (training) {
    const unsigned nrb = nblk(P, PIECE_STAT_ROWS);
    hipLaunchKernelGGL(k_col_stats, dim3(nblk(C, 256), nrb), dim3(256), 0, st, (const float*)w.xin, (long)ldc, P, C,
                       PIECE_STAT_ROWS, w.st.part);
    PN_OK(reduce_parts<double>(w.st.part, nrb, 2 * C, C, w.sum_a, w.sq_a, nullptr, w.st.red, st));
    PN_OK(fold_train(st, blk->bn1, (const double*)w.sum_a, (const double*)w.sq_a, (double)P, bn_eps, bn_mom, C, ldc, w.s1, w.t1,
                     nullptr, nullptr));
  } else {
    hipLaunchKernelGGL(k_bn_fold_eval, dim3(nblk(ldc, 256)), dim3(256), 0, st, blk->bn1, (const float*)nullptr, bn_eps, C, ldc,
                       w.s1, w.t1);
  }
  // masked_conv1 on relu(bn1(x)): the tap gather masks it (positions >= len read as 0, output rows >= len are 0)
  PN_OK(piece_conv(w.xin, ldc, blk->conv_a_w, blk->conv_a_b, Cb, ldb, w.z, ksize, dilation, w.s1, w.t1, nullptr,
                   training ? w.sum_b : nullptr, training ? w.sq_b : nullptr, w.lens32, B, L, w, st));
  if (training) {
    PN_OK(fold_train(st, blk->bn2, (const double*)w.sum_b, (const double*)w.sq_b, (double)P, bn_eps, bn_mom, Cb, ldb, w.s2, w.t2,
                     nullptr, nullptr));
  } else {
    hipLaunchKernelGGL(k_bn_fold_eval, dim3(nblk(ldb, 256)), dim3(256), 0, st, blk->bn2, (const float*)nullptr, bn_eps, Cb, ldb,
                       w.s2, w.t2);
  }
  // masked_conv2 (1 x 1) + x on the live rows; the pad rows of `out + x` (:66) are x itself
  PN_OK(piece_conv(w.z, ldb, blk->conv_b_w, blk->conv_b_b, C, ldc, w.y, 1, 1, w.s2, w.t2, w.xin, nullptr, nullptr, w.lens32, B, L,
                   w, st));
  hipLaunchKernelGGL(k_nlc_to_ncl, dim3(nblk((long)B * C * L, 256)), dim3(256), 0, st, (const float*)w.y, ldc,
                     (const int*)w.lens32, x, out, B, C, L);
  HIP_OK(hipGetLastError());
  return 0;
}
