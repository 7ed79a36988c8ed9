/* protnote_hip.h - C ABI of libprotnote_hip.so, the MI355X (gfx950) implementation of ProtNote's
 * forward/training hot path.
 *
 * The reference (microsoft/protnote) is 100 % Python and has no FFI of its own; the boundary its callers
 * use is the nn.Module API (bin/main.py:383-452, protnote/models/ProtNoteTrainer.py:288,729).  Every
 * entry point below therefore names the reference *function* it replaces (file:line under
 * /root/reference); the ctypes binding a maintainer would add is protnote_amd/_lib.py (see
 * INTEGRATION.md).
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; pn_last_error() returns the message of
 *    the calling thread's last failure.
 *  - all pointers are DEVICE pointers (HBM) unless the name ends in _host; f32 unless typed otherwise.
 *  - no hidden allocation, no ownership transfer: scratch comes from the caller (`ws`, size from the
 *    matching *_ws_bytes query); launches go to `stream` (a hipStream_t passed as void*), nothing
 *    synchronises the device.  The library never calls hipMalloc: even the arrival counters that pace
 *    the big weight-gradient kernel live in the `ws` of the call that launches it.
 *  - thread-safe per stream: calls that use DIFFERENT streams, workspaces and save buffers may run
 *    concurrently from different host threads (tests/test_hip_train.py::
 *    test_two_models_on_two_streams_concurrently_equal_serial).  The pn_set_* switches are process-wide.
 *  - collectives are NOT part of this ABI, deliberately (SURVEY 2.3 / 8b lists pn_comm_init / allreduce /
 *    broadcast as an option): data-parallel exchange is torch.distributed over RCCL in the Python host
 *    layer (protnote_amd/utils/distributed.py - one flat gradient all-reduce, one BN-buffer broadcast,
 *    one [3, N_L] count all-reduce), exactly where the reference has it (DDP, bin/main.py:452); the one
 *    place the C side needs a cross-rank sum inside a call (SYNC_BN) takes it as a callback, pn_set_sync_bn.
 *  - internal activations are channels-last [B*L, ld4(C)] with ld4(C) = C rounded up to a multiple of 4
 *    and zero pad lanes; conv weights must be packed by pn_pack_conv_weight first.
 */
#ifndef PROTNOTE_HIP_H
#define PROTNOTE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_MAX_BLOCKS 16
#define PN_MAX_LAYERS 8
/* element types of a multihot label / target array (pn_loss_fwd_bwd_t, pn_tp_fn_fp_t, the metrics entry points) */
#define PN_LABEL_F32 0
#define PN_LABEL_I64 1
#define PN_LABEL_U8 2

const char* pn_last_error(void);
int pn_version(void);
/* 16 hex digits: sha256 over the sources this binary was built from (csrc/*.hip, *.hpp, *.cpp and this header;
 * protnote_amd/build.py csrc_hash()).  The binding refuses a binary whose hash differs from the sources beside it. */
const char* pn_build_hash(void);

/* torch.nn.BatchNorm1d state: weight, bias, running_mean, running_var (each [C]) */
typedef struct pn_bn {
  const float* weight;
  const float* bias;
  float* running_mean;
  float* running_var;
} pn_bn;

/* ---- ProteInfer encoder: protnote/models/protein_encoders.py:70-153 ---- */
typedef struct pn_res_block { /* Residual, protein_encoders.py:23-67 */
  pn_bn bn1;
  const float* conv_a_w; /* packed [Cb][ksize][ld4(C)]  (masked_conv1) */
  const float* conv_a_b; /* [Cb] */
  pn_bn bn2;
  const float* conv_b_w; /* packed [C][1][ld4(Cb)]      (masked_conv2) */
  const float* conv_b_b; /* [C] */
} pn_res_block;

typedef struct pn_encoder {
  int Cin, C, Cb, ksize, nblocks, dil_base;
  const float* conv1_w; /* packed [C][ksize][ld4(Cin)] */
  const float* conv1_b; /* [C] */
  pn_res_block blk[PN_MAX_BLOCKS];
  /* pn_encoder_fwd_train / pn_encoder_bwd only: != 0 = BatchNorm in EVAL mode inside a differentiable forward (see
   * pn_mlp.bn_use_running); pn_encoder_fwd takes its mode from the `training` argument */
  int bn_use_running;
  /* arithmetic of this call's big GEMMs: 0 = the library default (pn_set_math_mode), 1 = f32 MFMA, 2 = bf16x3.  Per call
   * and per calling thread: two host threads may drive two models in different modes at the same time. */
  int math_mode;
} pn_encoder;

/* torch Conv1d weight [Cout][Cin][k] -> packed [Cout][k][ld4(Cin)] (zero pad lanes). */
int pn_pack_conv_weight(const float* w, float* packed, int Cout, int Cin, int k, void* stream);

size_t pn_encoder_ws_bytes(const pn_encoder* enc, int B, int L);

/* ProteInfer.get_embeddings (protein_encoders.py:109-118): onehots [B][Cin][L] f32, lens [B] i64 ->
 * emb [B][ld_emb] (first C columns).  training != 0 reproduces train-mode BatchNorm (batch statistics
 * over all B*L positions incl. zeroed pads, running-stat update, momentum 0.01, eps 1e-3) exactly as the
 * "frozen" encoder behaves under model.train() (ProtNoteTrainer.py:844). */
int pn_encoder_fwd(const pn_encoder* enc, const float* onehots, const int64_t* lens, int B, int L,
                   float* emb, int ld_emb, int training, void* ws, size_t ws_bytes, void* stream);

/* MaskedConv1D.forward (protein_encoders.py:8-17) called stand-alone: x [B][Cin][L] f32, lens [B] i64 -> out [B][Cout][L];
 * input and output zeroed at positions >= len.  w_packed from pn_pack_conv_weight.  (ProteInfer never calls this: pn_encoder_fwd
 * runs the convolutions fused and channels-last.) */
size_t pn_masked_conv1d_ws_bytes(int B, int L, int Cin, int Cout);
int pn_masked_conv1d_fwd(const float* x, const int64_t* lens, const float* w_packed, const float* bias, int B, int Cin, int Cout,
                         int L, int ksize, int dilation, float* out, void* ws, size_t ws_bytes, void* stream);
/* Residual.forward (protein_encoders.py:61-67) called stand-alone: x [B][C][L] -> out [B][C][L] =
 * masked_conv2(relu(bn2(masked_conv1(relu(bn1(x)))))) + x.  As in the reference bn1 sees the RAW input (train-mode statistics
 * run over all B*L positions, pads included) and the residual adds it back unmasked: positions >= len of `out` hold x.
 * training != 0: batch statistics, running statistics updated (momentum 0.01, eps 1e-3). */
size_t pn_residual_ws_bytes(int B, int L, int C, int Cb);
int pn_residual_fwd(const pn_res_block* blk, int C, int Cb, int ksize, int dilation, const float* x, const int64_t* lens, int B,
                    int L, float* out, int training, void* ws, size_t ws_bytes, void* stream);

/* ProteInfer.get_embeddings from residue IDS: ids = the batch's residue indices back to back (uint8), offsets [B+1] i64 (the
 * input of pn_onehot_batch - what the device-side collator holds).  Bit-identical to pn_onehot_batch + pn_encoder_fwd without
 * the [B][Cin][L] one-hot tensor; needs the gather form of conv1 (kernel_size 9, Cin <= 27) and a frozen encoder. */
int pn_encoder_fwd_ids(const pn_encoder* enc, const uint8_t* ids, const int64_t* offsets, int B, int L, float* emb, int ld_emb,
                       int training, void* ws, size_t ws_bytes, void* stream);

/* ---- row MLPs W_p / W_l: torchvision.ops.MLP built at protnote/models/ProtNote.py:63-81 ---- */
typedef struct pn_mlp {
  int nlayers;              /* number of Linear layers */
  int dims[PN_MAX_LAYERS + 1]; /* dims[0]=in, dims[i+1]=out of layer i */
  const float* w[PN_MAX_LAYERS];    /* [dims[i+1]][dims[i]] */
  const float* bias[PN_MAX_LAYERS]; /* or NULL */
  pn_bn bn[PN_MAX_LAYERS];          /* BN after layer i for i < nlayers-1 (weight==NULL: no BN) */
  float bn_eps, bn_momentum;
  /* training only: Dropout(p) after every hidden ReLU AND after the last Linear (torchvision.ops.MLP, reference
   * ProtNote.py:63-81 with dropout=OUTPUT_MLP_DROPOUT).  Masks are a counter-based hash of (dropout_seed, layer, row,
   * column): the caller draws a fresh seed per forward and passes the SAME seed to the backward. */
  float dropout_p;
  unsigned dropout_seed;
  int dropout_stream; /* base of this stack's mask streams: 100 for W_p, 200 for W_l (pn_dropout_mask) */
  /* *_fwd_train / *_bwd only.  0 (default): train-mode BatchNorm - batch statistics, running statistics updated.
   * != 0: the module is in eval() but autograd is on (reference ProtNote.forward puts no mode restriction on either,
   * ProtNote.py:243-309): normalise with the RUNNING statistics, update nothing; the backward treats them as
   * constants (dz = relu' * g * gamma/sigma; dgamma / dbeta as usual). */
  int bn_use_running;
  int math_mode; /* as pn_encoder.math_mode */
} pn_mlp;

size_t pn_mlp_rows_ws_bytes(const pn_mlp* m, int rows);

/* y[rows][dims[n]] = MLP(x[rows][ldx]); eval-mode BN only in this entry point.
 * Replaces self.W_p(P_f) / self.W_l(L_f), ProtNote.py:270-271. */
int pn_mlp_rows_fwd_eval(const pn_mlp* m, const float* x, int ldx, int rows, float* y, void* ws,
                         size_t ws_bytes, void* stream);

/* ---- pair head: _get_joint_embeddings + output_layer, ProtNote.py:112-152,286-293,337-378 ---- */
typedef struct pn_pairhead {
  int d;        /* latent dim: P_e [B][d], L_e [NL][d] */
  int in_dim;   /* 2d (concatenation) or 3d (concatenation_diff / _prod) */
  int fusion;   /* 0 concatenation, 1 concatenation_diff, 2 concatenation_prod */
  int nlayers;  /* hidden layers (>= 1; OUTPUT_MLP_NUM_LAYERS: 1 has no pair-grid GEMM at all) */
  int h;        /* hidden width */
  const float* w[PN_MAX_LAYERS];    /* w[0]: [h][in_dim]; w[i>0]: [h][h] */
  const float* bias[PN_MAX_LAYERS]; /* hidden-layer bias when there is no BN, else NULL */
  pn_bn bn[PN_MAX_LAYERS];
  const float* w_out; /* [h] */
  const float* b_out; /* [1] */
  float bn_eps, bn_momentum;
  /* training only: Dropout(p) after the ReLU of every hidden layer except the last (get_mlp, ProtNote.py:369-371);
   * generated inside the operand loaders of the next GEMM and regenerated in the backward from the same seed */
  float dropout_p;
  unsigned dropout_seed;
  int bn_use_running; /* as in pn_mlp: eval-mode BatchNorm inside pn_pairhead_fwd_train / pn_pairhead_bwd */
  int math_mode;      /* as pn_encoder.math_mode */
  /* pn_pairhead_bwd: arithmetic of the hidden layers' backward pair-grid GEMMs: 0 = the library default
   * (pn_set_backward_math), 1 = as the forward, 2 = one bf16 product with f32 accumulation (AMP class) */
  int backward_math;
  /* pn_pairhead_fwd_*: arithmetic of the hidden layers' FORWARD pair-grid GEMMs (z_l = h_{l-1} W_l^T, l >= 1): 0 = the library
   * default (pn_set_forward_math), 1 = as math_mode, 2 = one bf16 product with f32 accumulation (AMP class, see
   * pn_set_forward_math).  pn_pairhead_bwd ignores it (the backward regenerates h from the stored f32 z).  The workspace
   * queries (pn_pairhead_eval_ws_bytes / _train_ws_bytes) read it too - mode 2 carves one chunk of bf16 operand - so query and
   * call must see the same value (with 0: the same process default). */
  int forward_math;
} pn_pairhead;

size_t pn_pairhead_eval_ws_bytes(const pn_pairhead* hd, int B, int NL, int label_chunk);

/* logits_pairs[j*B + i] (label-major pair grid) for all B x NL pairs, eval-mode BN.
 * The [B*NL, 2d] joint tensor of the reference is never materialised: layer 1 is separable,
 * z1[i,j] = P_e[i] W1a^T + L_e[j] W1b^T. */
int pn_pairhead_fwd_eval(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                         float* logits_pairs, int label_chunk, void* ws, size_t ws_bytes, void* stream);

/* save_embeddings path (ProtNote.py:294-302): also returns the penultimate activations relu(bn(z_last)) of the
 * output MLP, hidden_pairs [NL*B][h] on the label-major pair grid (small evaluation subsets only). */
size_t pn_pairhead_hidden_ws_bytes(const pn_pairhead* hd, int B, int NL);
int pn_pairhead_fwd_eval_hidden(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                                float* logits_pairs, float* hidden_pairs, void* ws, size_t ws_bytes, void* stream);

/* ProtNote.additive_attention (ProtNote.py:154-166), inference: hidden [N][T][d] token embeddings,
 * attention_mask [N][T] i64, raw_attn_scorer weight [d] / bias [1] -> out [N][d]. */
int pn_additive_attention(const float* hidden, const int64_t* attention_mask, const float* w, const float* b, int N,
                          int T, int d, float* out, void* stream);

/* Backward of the pooling wrt raw_attn_scorer (training with LABEL_EMBEDDING_POOLING_METHOD: all; autograd of
 * ProtNote.py:154-166 with frozen token embeddings): dout [N][d] = gradient wrt the pooled embeddings;
 * dw [d], db [1] are written.  Per-label partials + fixed-order reduction (bit-reproducible). */
size_t pn_additive_attention_bwd_ws_bytes(int N, int d);
int pn_additive_attention_bwd(const float* hidden, const int64_t* attention_mask, const float* w, const float* b,
                              const float* dout, int N, int T, int d, float* dw, float* db, void* ws, size_t ws_bytes,
                              void* stream);

/* ProtNote.py:308-322: pair logits for NL = n_out*ndesc description rows (consecutive rows = one label)
 * -> out[B][n_out];  ndesc == 1: plain re-layout;  else logit(mean_d sigmoid(x), eps=1e-7).
 * protein_major = 0: input is the label-major pair grid x[j*B + i]; 1: input is x[i*NL + j]. */
int pn_ensemble_logit(const float* logits_pairs, int B, int NL, int ndesc, int protein_major, float* out,
                      void* stream);

/* Backward of the ensembling for a differentiated eval-mode forward (autograd through ProtNote.py:313-322):
 * logits [B][NL] protein-major description-row logits, dout [B][NL/ndesc] -> dlogits [B][NL]. */
int pn_ensemble_logit_bwd(const float* logits, const float* dout, int B, int NL, int ndesc, float* dlogits,
                          void* stream);

/* ProtNote.py:219-240: out = L_f + (2u - 1) * scale, scale = alpha / sqrt(d); u ~ U[0,1) from the caller. */
int pn_label_noise(const float* L_f, const float* u, float scale, float* out, long n, void* stream);
/* ... with u drawn inside the kernel from a counter hash of (seed, row, column) - no tensor of uniforms exists - and the hook
 * that hands a test (or the oracle) the very same u [rows][cols] (24-bit uniforms in [0, 1), like torch's float uniform). */
int pn_label_noise_seeded(const float* L_f, unsigned seed, float scale, float* out, long rows, int cols, void* stream);
int pn_uniform(unsigned seed, long rows, int cols, float* out, void* stream);

/* ---- similarity head, ProtNote.py:281-284 ---- */
size_t pn_similarity_ws_bytes(int B, int NL);
int pn_similarity_fwd(const float* P_e, const float* L_e, int B, int NL, int d, float temperature,
                      float* logits /* [B][NL] */, void* ws, size_t ws_bytes, void* stream);

/* ================================ training path ================================ */

/* gradient destinations (device pointers, each the shape of the matching parameter).  NULL = the parameter is frozen
 * (requires_grad False - e.g. TRAIN_PROJECTION_HEAD: False freezes output_layer.*, ProtNoteTrainer.py:221-222): its
 * gradient is not computed at all - for a weight that is one whole TN GEMM less - while the data gradient still flows
 * through the layer to whatever is trainable below it. */
typedef struct pn_mlp_grads {
  float* dw[PN_MAX_LAYERS];
  float* dgamma[PN_MAX_LAYERS];
  float* dbeta[PN_MAX_LAYERS];
} pn_mlp_grads;

typedef struct pn_pairhead_grads {
  float* dw[PN_MAX_LAYERS];
  float* dgamma[PN_MAX_LAYERS];
  float* dbeta[PN_MAX_LAYERS]; /* layer without BatchNorm (OUTPUT_MLP_BATCHNORM: False): gradient of its Linear bias */
  float* dw_out; /* [h] */
  float* db_out; /* [1] */
} pn_pairhead_grads;

/* ProteInfer encoder with TRAIN_SEQUENCE_ENCODER: True (ProtNote.py:248-256): training forward that keeps the
 * block inputs / conv outputs / BN batch statistics in `save`, and the backward from d(embeddings) to every encoder
 * parameter (gradient tensors in torch layout: conv weights [Cout][Cin][k]). */
typedef struct pn_res_block_grads {
  float *bn1_w, *bn1_b, *conv_a_w, *conv_a_b, *bn2_w, *bn2_b, *conv_b_w, *conv_b_b;
} pn_res_block_grads;
typedef struct pn_encoder_grads {
  float* conv1_w;
  float* conv1_b;
  pn_res_block_grads blk[PN_MAX_BLOCKS];
} pn_encoder_grads;
size_t pn_encoder_train_save_bytes(const pn_encoder* enc, int B, int L);
size_t pn_encoder_bwd_ws_bytes(const pn_encoder* enc, int B, int L);
int pn_encoder_fwd_train(const pn_encoder* enc, const float* onehots, const int64_t* lens, int B, int L, float* emb,
                         int ld_emb, void* save, size_t save_bytes, void* ws, size_t ws_bytes, void* stream);
int pn_encoder_bwd(const pn_encoder* enc, int B, int L, const float* demb, int ld_demb, const pn_encoder_grads* gr,
                   void* save, size_t save_bytes, void* ws, size_t ws_bytes, void* stream);

/* W_p / W_l with train-mode BatchNorm1d (batch statistics over `rows`, running-stat update) - the
 * training-time self.W_p(P_f) / self.W_l(L_f) of ProtNote.py:270-271 - and its backward.  `save` carries
 * the pre-activations and BN statistics from forward to backward (size: *_train_save_bytes). */
size_t pn_mlp_rows_train_save_bytes(const pn_mlp* m, int rows);
size_t pn_mlp_rows_train_ws_bytes(const pn_mlp* m, int rows);
int pn_mlp_rows_fwd_train(const pn_mlp* m, const float* x, int ldx, int rows, float* y, void* save,
                          size_t save_bytes, void* ws, size_t ws_bytes, void* stream);
int pn_mlp_rows_bwd(const pn_mlp* m, const float* x, int ldx, int rows, const float* dy, const pn_mlp_grads* gr,
                    float* dx /* or NULL */, void* save, size_t save_bytes, void* ws, size_t ws_bytes, void* stream);

/* Training-mode pair head (ProtNote.py:286-293 under model.train(): BatchNorm statistics over the whole
 * B x NL pair grid).  Forward stores the h-wide pre-activations of hidden layers 2..n in `save`
 * ((nlayers-1) * (B*NL + chunk) * h floats); backward walks them in label chunks and reuses the consumed
 * part of each buffer for the propagated gradient.  logits / dl are on the label-major pair grid [j*B+i]. */
size_t pn_pairhead_train_save_bytes(const pn_pairhead* hd, int B, int NL, int label_chunk);
size_t pn_pairhead_train_ws_bytes(const pn_pairhead* hd, int B, int NL);
int pn_pairhead_fwd_train(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL,
                          float* logits_pairs, int label_chunk, void* save, size_t save_bytes, void* ws,
                          size_t ws_bytes, void* stream);
int pn_pairhead_bwd(const pn_pairhead* hd, const float* P_e, const float* L_e, int B, int NL, const float* dl_pairs,
                    const pn_pairhead_grads* gr, float* dP_e, float* dL_e, int label_chunk, void* save,
                    size_t save_bytes, void* ws, size_t ws_bytes, void* stream);
/* save_embeddings=True on this path (ProtNote.py:292-302 in train mode; the trainer passes the flag through,
 * ProtNoteTrainer.py:288): hidden_pairs [NL*B][h] = relu(bn(z_last)), the penultimate activations of the forward whose
 * activations `save` holds (label-major pair grid).  Call between pn_pairhead_fwd_train and pn_pairhead_bwd. */
int pn_pairhead_train_hidden(const pn_pairhead* hd, int B, int NL, int label_chunk, const void* save, size_t save_bytes,
                             float* hidden_pairs, void* stream);

/* Backward of the similarity head (ProtNote.py:281-284 under autograd): dlogits [B][NL] -> dP_e [B][d],
 * dL_e [NL][d]; the L2 normalisations are recomputed. */
size_t pn_similarity_train_ws_bytes(int B, int NL, int d);
int pn_similarity_bwd(const float* P_e, const float* L_e, int B, int NL, int d, float temperature,
                      const float* dlogits, float* dP_e, float* dL_e, void* ws, size_t ws_bytes, void* stream);

/* Loss forward + d(mean loss)/dlogit + per-label TP/FN/FP in one pass over logits [B][N]
 * (utils/losses.py:190-213 FocalLoss, :275-276 BCEWithLogits(pos_weight); ProtNoteTrainer.py:61-83).
 * kind 0 = BCE, 1 = focal.  Exactly one of targets_f32 / targets_i64 is non-NULL.  tp/fn/fp (each [N], f32
 * holding integer counts) are ACCUMULATED into when non-NULL.
 * Element weights (the reference's other BCE variants, kind 0): weight_mode 0 = none; 1 = BatchWeightedBCE
 * (losses.py:124-146: positives and negatives of the batch weigh total/2 each); 2 = WeightedBCE / CBLoss
 * (losses.py:78-121, 214-241: row i weighs sum_j label_weights[j] * target[i][j]; label_weights [N] on the device).
 * rgd_temperature >= 0: RGDBCE (losses.py:58-75: mean loss m times exp(min(m, T) / (T + 1)), factor detached).
 * ws: >= pn_loss_ws_bytes(B, N) (row weights + one f64 loss partial per workgroup: the mean is summed in a fixed
 * order, no floating-point atomics - the whole train step is bit-reproducible run to run). */
size_t pn_loss_ws_bytes(int B, int N);
int pn_loss_fwd_bwd(const float* logits, const float* targets_f32, const int64_t* targets_i64, int B, int N,
                    int kind, float pos_weight, float gamma, float alpha, float smoothing, float threshold,
                    float* loss_out, float* dlogits, float* tp, float* fn, float* fp, int weight_mode,
                    const float* label_weights, float rgd_temperature, void* ws, size_t ws_bytes, void* stream);

/* The same pass with the targets as ONE typed pointer: target_kind = PN_LABEL_F32 | PN_LABEL_I64 | PN_LABEL_U8 (defined with the
 * metrics below).  A multihot is 1 B of information per pair; the reference's collator hands int64 (collators.py:136-137, row
 * (a)14 - still the default of the Python twin), PN_LABEL_U8 lets a caller that keeps its multihots as bytes
 * (collate_to_device(multihot_dtype=torch.uint8)) pay the algorithmic 1 B instead of 8.  Same arithmetic, bit for bit. */
int pn_loss_fwd_bwd_t(const float* logits, const void* targets, int target_kind, int B, int N, int kind, float pos_weight,
                      float gamma, float alpha, float smoothing, float threshold, float* loss_out, float* dlogits, float* tp,
                      float* fn, float* fp, int weight_mode, const float* label_weights, float rgd_temperature, void* ws,
                      size_t ws_bytes, void* stream);

/* LOSS_FN: SupCon (utils/losses.py:7-56, one_way_supcon over the label axis; the reference marks it unused and never
 * reads its temperature): loss = -mean_i [ sum_j y_ij log_softmax_j(x_i) / n_i ], rows without positives count 0 in the
 * loss and - as in the reference's autograd - NaN in the gradient.  ws >= pn_supcon_ws_bytes(B). */
size_t pn_supcon_ws_bytes(int B);
int pn_supcon_fwd_bwd(const float* logits, const float* targets_f32, const int64_t* targets_i64, int B, int N,
                      float* loss_out, float* dlogits, void* ws, size_t ws_bytes, void* stream);

/* calculate_tp_fn_fp (ProtNoteTrainer.py:61-83) on probabilities; outputs are overwritten. */
int pn_tp_fn_fp(const float* probs, const float* targets_f32, const int64_t* targets_i64, int B, int N,
                float threshold, float* tp, float* fn, float* fp, void* stream);

/* ... with typed targets (see pn_loss_fwd_bwd_t) */
int pn_tp_fn_fp_t(const float* probs, const void* targets, int target_kind, int B, int N, float threshold, float* tp, float* fn,
                  float* fp, void* stream);

/* clip_grad_norm_(max_norm) + Adam / AdamW step on flat f32 buffers (ProtNoteTrainer.py:745-755);
 * max_norm <= 0 disables clipping; norm_out (optional, [1]) receives the total gradient norm.
 * ws: >= PN_ADAM_WS_BYTES (per-workgroup partials of the squared norm, added in a fixed order). */
#define PN_ADAM_WS_BYTES 32768
int pn_clip_adam_step(float* w, const float* g, float* m, float* v, long n, float max_norm, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int step, float* norm_out, void* ws,
                      size_t ws_bytes, void* stream);

/* OPTIMIZER: SGD (ProtNoteTrainer.py:238-243: torch.optim.SGD(lr, weight_decay), momentum 0) after the same
 * clip_grad_norm_: g' = coef g + weight_decay w; with momentum != 0 (not used by the reference; torch semantics,
 * dampening 0, no Nesterov) momentum_buf [n] holds the velocity (step == 1 initialises it), else it may be NULL.
 * ws: >= PN_ADAM_WS_BYTES. */
int pn_clip_sgd_step(float* w, const float* g, float* momentum_buf, long n, float max_norm, float lr, float momentum,
                     float weight_decay, int step, float* norm_out, void* ws, size_t ws_bytes, void* stream);

/* dst[c][r] = src[r][c] (pair-grid <-> [B][N] re-layout of logits / dlogits) */
int pn_transpose(const float* src, long ld_src, int rows, int cols, float* dst, long ld_dst, void* stream);

/* C[M][N] = A[R][M]^T B[R][N]  (split-K f32-MFMA contraction over rows; unit tests / building block) */
int pn_gemm_tn(const float* A, long lda, const float* Bm, long ldb, float* C, long ldc, long R, int M, int N,
               void* ws, size_t ws_bytes, void* stream);

/* Device-side batch assembly (input layout of collate_variable_sequence_length, collators.py:123-133):
 * ids = the batch's residue indices back to back (uint8), offsets [B+1] i64 -> onehots [B][A][Lmax] f32
 * (zero-padded) and lengths [B] i64. */
int pn_onehot_batch(const uint8_t* ids, const int64_t* offsets, int B, int A, int Lmax, float* onehots,
                    int64_t* lengths, void* stream);

/* ---- evaluation metrics on the device -----------------------------------------------------
 * Replaces the per-batch D2H + CPU torcheval BinaryAUPRC / MultilabelAUPRC of ProtNoteTrainer.py:477-485,540-543
 * (ESTIMATE_MAP False: exact) and the on-device Binary/MultilabelBinnedAUPRC(threshold=50) (ESTIMATE_MAP True),
 * and torchmetrics AveragePrecision of utils/evaluation.py:148-169.  Label element types: */

/* Exact AP.  The accumulator is label-major and stays in HBM for the whole evaluation: keys [N_L][cap] u32
 * (order-preserving image of the f32 score), hits [N_L][cap] u8.  pn_ap_append transposes one batch
 * (scores [B][ld_s] f32, labels [B][ld_y] of label_kind) into columns [n0, n0+B). */
int pn_ap_append(const float* scores, int ld_s, const void* labels, int label_kind, int ld_y, int B, int N_L,
                 uint32_t* keys, uint8_t* hits, long long cap, long long n0, void* stream);
/* AP over the first n columns: per label -> ap [N_L] f64 (NaN where a label has no positive), npos [N_L] i64
 * (optional); if micro_ap != NULL also the AP over all N_L*n pairs -> micro_ap [1], micro_npos [1] (optional).
 * AP = sum over distinct score thresholds (ties share one) of (recall step) x precision, f64, deterministic.
 * The accumulator is not modified. */
size_t pn_ap_ws_bytes(int N_L, long long n, long long cap, int micro);
int pn_ap_compute(const uint32_t* keys, const uint8_t* hits, int N_L, long long n, long long cap, double* ap,
                  long long* npos, double* micro_ap, long long* micro_npos, void* ws, size_t ws_bytes, void* stream);

/* Multi-GPU micro AP without gathering the evaluation set on every GPU (the reference's sync_and_compute ships every
 * rank's scores to every rank, ProtNoteTrainer.py:655-657): after a sample-sort exchange this GPU holds ALL pairs of the
 * evaluation whose key falls into one key range; tp_before / k_before = positives / pairs held by the higher-ranked
 * ranges.  partial [1] f64 = sum over this range's tie groups of (TP_g - TP_{g-1}) * TP_g / k_g with the global TP_g, k_g
 * (not normalised), npos [1] i64 = positives of the range; micro AP = sum(partial) / sum(npos) over the GPUs.
 * keys / hits [m] are not modified. */
size_t pn_ap_partial_ws_bytes(long long m);
int pn_ap_partial(const uint32_t* keys, const uint8_t* hits, long long m, long long tp_before, long long k_before,
                  double* partial, long long* npos, void* ws, size_t ws_bytes, void* stream);

/* Binned AUPRC.  Streaming state: pos_hist / all_hist [(N_L+1)][T+1] u64 (zero-initialised by the caller; row N_L
 * = all labels pooled, written by pn_binned_auprc when with_micro), bin(p) = #{k : p >= thresholds[k]},
 * thresholds [T] ascending f32 on the device (T <= 120).
 * pn_binned_auprc: out [N_L (+1 if with_micro)] f64, area = sum_k (recall_k - recall_{k+1}) precision_k with
 * precision = 1 where nothing is predicted and a final (1, 0) point; NaN where a label has no positive. */
int pn_binned_hist_update(const float* scores, int ld_s, const void* labels, int label_kind, int ld_y, int B, int N_L,
                          const float* thresholds, int T, unsigned long long* pos_hist, unsigned long long* all_hist,
                          void* stream);
int pn_binned_auprc(unsigned long long* pos_hist, unsigned long long* all_hist, int N_L, int T, double* out,
                    long long* npos, int with_micro, void* stream);

/* Arithmetic of the pair-grid GEMMs (process-wide): 0 = exact f32 MFMA (default; results are a k-ordered fmaf
 * chain), 1 = "bf16x3": operands split into bf16 hi + lo on the fly, a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the bf16
 * matrix pipe with f32 accumulation - ~1e-5 relative error per product instead of 6e-8, several times faster.
 * The encoder, the row MLPs (W_p, W_l) and all reductions stay f32 in either mode. */
int pn_set_math_mode(int mode);

/* SYNC_BN: True (reference bin/main.py:449-450, nn.SyncBatchNorm.convert_sync_batchnorm): train-mode BatchNorm of the
 * encoder, the projection heads and the pair head takes its statistics over the batches of ALL ranks, in the forward
 * (sum, sum of squares) and in the backward (sum du, sum du*xhat; dgamma / dbeta stay rank-local, as in torch's
 * SyncBatchNorm).  The library calls hook(n, user) whenever the first n doubles of `stage` (device memory) must be
 * summed over the ranks, in place and ordered on the stream the library was called on; it returns 0 on success.
 * `world` = number of ranks.  The ranks' batches may differ in shape (each collator pads to its own batch maximum,
 * collators.py:40; ragged last batch): every reduction carries the rank's row count as one more double and the global
 * count is the sum of the local ones, as torch.nn.SyncBatchNorm gathers them.  hook == NULL switches back to per-rank
 * statistics (the default, SYNC_BN: False). */
/* Row-MLP backward over >= 16384 rows (W_l over the label table): 1 (default) materialises dY once and runs the 256-tile
 * kernels, 0 regenerates it in the operand loaders of the 128-tile engine (A/B switch; no reference counterpart). */
int pn_set_mlp_materialize(int on);

typedef int (*pn_sync_hook)(long n_doubles, void* user);
int pn_set_sync_bn(pn_sync_hook hook, void* user, double* stage, long stage_doubles, int world);
int pn_get_math_mode(void);

/* Arithmetic of the BACKWARD pair-grid GEMMs of the output MLP's hidden layers (dW_l = dz_l^T h_{l-1}, dh_{l-1} = dz_l W_l):
 * 0 (default) = whatever pn_set_math_mode selected for the forward; 1 = one product of the bf16-rounded operands with
 * f32 accumulation (v_mfma_f32_32x32x16_bf16) - the arithmetic class of the reference's own training run, whose Linear
 * gradient GEMMs execute in half precision under torch.autocast (ProtNoteTrainer.py:728-738).  The forward (logits are
 * bit-identical to mode 0), the BatchNorm backward, every reduction, the layer-1 factorisation and the row MLPs are not
 * affected; layers with OUTPUT_MLP_DROPOUT > 0 keep the f32 kernels. */
int pn_set_backward_math(int mode);
int pn_get_backward_math(void);

/* Arithmetic of the FORWARD pair-grid GEMMs of the output MLP's hidden layers (z_l = h_{l-1} W_l^T, l >= 1; the separable first
 * layer has no pair-grid GEMM): 0 (default) = whatever math_mode selects; 1 = one product of the bf16-rounded operands with f32
 * accumulation (v_mfma_f32_32x32x16_bf16) - the arithmetic class of the reference's own GPU run, whose forward executes under
 * torch.autocast as well (ProtNoteTrainer.py:287 eval, :728-729 train).  What stays as math_mode says: the layer-1 tables, the
 * extra pair GEMM of concatenation_prod, W_p / W_l, the encoder; what stays f32 in every mode: BatchNorm statistics (f64
 * partials), the STORED pre-activations z_l (so the backward is untouched), the row-dot of the output neuron, the loss.  Opt-in:
 * rounding the h x h weights to bf16 alone moves O(1) logits by ~1e-2 (profiles/r05_fp16_weight_probe.json), so this mode
 * cannot meet the 1e-3 logit bound the f32 / bf16x3 modes are held to; it is held to torch's own autocast(bfloat16) run of
 * the oracle instead (tests/test_hip_fwd_bf16.py).  Layers with OUTPUT_MLP_DROPOUT > 0 keep the math_mode kernels (they carry
 * the mask code); shapes the single-product kernel does not cover (hidden width not a multiple of 256) fall back likewise. */
int pn_set_forward_math(int mode);
int pn_get_forward_math(void);
/* Route of mode 1: 1 (default) = the activation operand h_{l-1} is written once per chunk as bf16 (k_make_h_bf16) and both
 * operands of z_l = h_{l-1} W_l^T go by LDS-DMA (gemm_nt_bf16dma_kernel; needs a hidden width that is a multiple of 256);
 * 0 = the single-product instantiation of the bf16x3 kernel rounds the operand while staging it through registers.  Same bf16
 * values in the same products, another k order inside a 16-k MFMA step: the routes agree to f32 summation order.  A/B switch. */
int pn_set_fwd_staged(int on);
/* MFMA shape of the all-LDS-DMA one-product bf16 NT GEMMs (the staged route above and dh = dz W with dz stored as bf16):
 * 1 (default) = v_mfma_f32_16x16x32_bf16 (gemm_bf16_m16.hpp - the shape this package can feed: +12 % in the main loop),
 * 0 = v_mfma_f32_32x32x16_bf16 (bwd_bf16_dz.hpp).  Every accumulator is bit-identical between the two; row dots and BatchNorm
 * column partials reduce in another order (last-ulp differences).  A/B switch. */
int pn_set_bf16_mfma16(int on);
/* Kernels of mode 1, a bit mask (default 7).  Bit 0: dh = dz W on the deep-pipelined single-product kernel (gemm_bf16.hpp:
 * every operand fetched two slabs ahead) instead of the single-product instantiation of the bf16x3 kernel.  Bit 1: dW = dz^T h
 * on the transpose-read kernel (16-byte row loads, K-major LDS image, ds_read_b64_tr_b16) instead of the single-product
 * instantiation of the bf16x3 TN kernel.  Masks 0 / 1 / 3 issue the same products in the same order: bit-identical.  Bit 2:
 * dz is STORED as bf16 (bwd_bf16_dz.hpp: written in place into the first half of each f32 row; both operands of dh = dz W
 * then go by LDS-DMA, the dW kernel stages dz as it is) wherever a layer's shapes allow - the same bf16 values in the same
 * products, another k order inside a 16-k MFMA step of the dh GEMM (agrees to f32 summation order).  The switch exists for
 * A/B timing and for the test that holds the variants to each other. */
int pn_set_bwd_deep(int mask);

/* Operand staging of the f32 pair-grid GEMMs: 1 (default) = LDS-DMA (global_load_lds, gemm_dma.hpp),
 * 0 = the register-staged engine (gemm_engine.hpp).  Same arithmetic in the same order: results are bit-identical;
 * the switch exists for A/B timing and for the test that asserts exactly that. */
int pn_set_f32_dma(int on);
/* Same switch for the bf16x3 pair-grid GEMMs (weight operand pre-split into bf16 hi / lo planes and staged by LDS-DMA). */
int pn_set_b3_dma(int on);

/* pn_encoder_fwd_train (the forward of a TRAINABLE encoder): 1 (default) = its two wide convolutions per block accumulate
 * in float64 on the matrix cores (v_mfma_f64_16x16x4_f64) and round to f32 once, so that the stored pre-activations and
 * the ReLU masks of the backward are the correctly rounded ones (gradient error class of the reference's f32 CPU run
 * instead of that of a K = 9900 f32 chain); 0 = the f32-MFMA kernels of pn_encoder_fwd (A/B measurements). */
int pn_set_encoder_f64(int on);

/* conv1 of the encoder (MaskedConv1D(20 -> C, k), protein_encoders.py:84-91): 1 (default) = when the input is one-hot
 * (checked on the device per call) a gather-sum kernel bound by the activation write, bit-identical to the general
 * convolution, which only runs for inputs that are not one-hot; 0 = always the general convolution (A/B, tests). */
int pn_set_conv1_gather(int on);

/* The dropout keep-mask the kernels generate for (seed, stream, rows x cols): out[r][c] = 1.0 (kept) or 0.0, so a
 * test can hand the oracle the very same masks.  stream: row MLP hidden layer l -> base + l, its output -> base + 99
 * (base 100 for W_p, 200 for W_l); pair-head hidden layer l -> 300 + l.  A pair-grid row is r = j * B + i. */
int pn_dropout_mask(unsigned seed, int stream, float p, long rows, int cols, float* out, void* stream_);

/* ---- measurement hook (bench.py `roofline`): between begin and end every GEMM-engine launch is bracketed
 * by hipEvents on its own stream.  pn_prof_end aggregates per kernel kind
 * (kind = family*100 + operand_kind*10 + epilogue_kind; family 0 = NT engine, 1 = TN engine); the caller
 * must have synchronised the device first.  flops = 2*M*N*K of each launch (arithmetic actually issued).
 * Kinds >= 2000 are the HBM-bound streaming stages (bench.py `stages`); for them `total_flops` carries the
 * ALGORITHMIC BYTES of the launches (what the pass must read + write once): 2001 conv1 from one-hots (K2),
 * 2002 masked mean-pool (K6), 2003 loss + dlogits + TP/FN/FP (K13/K14), 2004 clip + Adam/SGD (K16),
 * 2005 dz in place, 2006 BatchNorm-backward statistics, 2007 layer-1 masked reduction, 2008 row-dot logits,
 * 2009 convolution operand staging, 2010 the AMP-class forward's activation operand written as bf16.  GEMM kinds carry the
 * arithmetic as an offset: + 1000 bf16x3, + 1500 one bf16 product (NT) / + 1600 (TN), 1700 + 10 * source + epilogue = one bf16
 * product on a materialised bf16 operand (source 0 = a bf16 activation, 1 = relu(bn(z)), 2 = pair sum).  Kinds >= 3000 are VALU-bound stages; `total_flops` carries their algorithmic vector
 * instructions per lane-element: 3001 the one-hidden-layer head's forward (3 per pair and hidden column), 3002 its backward
 * masked reductions (8).  Returns the number of kinds written. */
int pn_prof_begin(void);
int pn_prof_end(int max_kinds, int* kinds, long* counts, double* total_ms, double* total_flops);

/* ---- generic f32-MFMA GEMM (unit tests / building block): C[M][N] = relu?(A*s+t)[M][K] W[N][K]^T + bias.
 * col_sum / col_sumsq (optional, [N] f64): per-column sum and sum of squares of the stored values, WRITTEN (not
 * accumulated); they need ws >= pn_gemm_nt_stats_ws_bytes(M, N): every workgroup stores its column partials in its
 * own slot and a fixed-order reduction adds them (train-mode BatchNorm statistics without atomics). */
size_t pn_gemm_nt_stats_ws_bytes(int M, int N);
int pn_gemm_nt(const float* A, long lda, const float* W, long ldw, float* C, long ldc, int M, int N, int K,
               const float* bias, const float* a_scale, const float* a_shift, double* col_sum,
               double* col_sumsq, int tile_variant, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
