// libprotnote_hip.so - MI355X (gfx950) kernels + C ABI for the ProtNote hot path.  See include/protnote_hip.h.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/protnote_hip.h"
#include "common.hpp"
#include "gemm_engine.hpp"
#include "gemm_bf16x3.hpp"
#include "gemm_bf16.hpp"
#include "gemm_dma.hpp"
#include "gemm_conv_dma.hpp"
#include "gemm_conv_f64.hpp"
#include "gemm_tn_fast.hpp"
#include "train_kernels.hpp"
#include "bwd_bf16_dz.hpp"
#include "fwd_bf16_h.hpp"

using namespace pn;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

#define HIP_OK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                      __FILE__, __LINE__);                                   \
  } while (0)
#define PN_OK(expr)         \
  do {                      \
    int _r = (expr);        \
    if (_r != 0) return _r; \
  } while (0)

int pn::fail_msg(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

extern "C" const char* pn_last_error(void) { return g_err; }
extern "C" int pn_version(void) { return 1; }

static inline int ld4(int c) { return (c + 3) & ~3; }
static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Bump {
  char* base;
  size_t off, cap;
  bool ok;
  Bump(void* p, size_t cap_) : base((char*)p), off(0), cap(cap_), ok(true) {}
  template <class T>
  T* take(size_t n) {
    size_t b = al256(n * sizeof(T));
    if (off + b > cap) {
      ok = false;
      return nullptr;
    }
    T* r = (T*)(base + off);
    off += b;
    return r;
  }
};

// ------------------------------------------------------------------------------------------------
// optional per-launch timing of the GEMM kernels (hipEvents on the launch stream; bench.py roofline)
// ------------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <vector>
struct ProfRec {
  int kind;  // family*100 + operand kind*10 + epilogue kind  (family 0: NT engine, 1: TN engine)
  double flops;
  hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;  // launches may come from several host threads (one stream each)

// HBM-bound streaming passes are timed the same way under kinds >= 2000; their `flops` field carries the pass's
// ALGORITHMIC BYTES per launch (SURVEY 8d / DESIGN 4.3: what the pass must read + write once), so bench.py can put
// every stage next to the 8 TB/s HBM roofline.
enum {
  ST_CONV1 = 2001,        // K2: one-hots -> conv1 output (NCL->NLC re-layout + the 20-channel conv)
  ST_POOL = 2002,         // K6: masked mean-pool
  ST_LOSS = 2003,         // K13/K14: loss + dlogits + TP/FN/FP in one pass
  ST_CLIP_OPT = 2004,     // K16: sum of squares + clip + Adam / SGD
  ST_DZ_APPLY = 2005,     // BatchNorm/ReLU backward applied in place
  ST_BN_BWD_STATS = 2006, // sum du, sum du*xhat over the stored pre-activations
  ST_PAIR_MASK_REDUCE = 2007,  // layer-1 backward: M0 / M1 tables from one pass over the gradient
  ST_ROWDOT = 2008,       // logits from the stored top pre-activation
  ST_CONV_STAGE = 2009,   // relu(bn(x)) staged once per convolution for the all-DMA conv kernel
  ST_MAKE_H = 2010,       // forward_math = bf16: the activation operand of a pair-grid GEMM written once as bf16 (fwd_bf16_h.hpp)
  // VALU-bound stages (kinds >= 3000): the `flops` field carries the pass's ALGORITHMIC VECTOR OPERATIONS per lane-element.
  // bench.py prices them against the vector unit's issue rate, 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T issue slots/s:
  // an add or fma has a packed f32 form (two operations per slot -> 78.6 T/s), fmaxf has none on gfx9 (one per slot), so the
  // forward's add + max + fma mix peaks at 3 operations per 2 slots = 59 T/s; the backward is priced at the packed rate
  ST_PAIR1_FWD = 3001,    // OUTPUT_MLP_NUM_LAYERS: 1 forward: add, max, fma per pair and hidden column
  ST_PAIR1_BWD = 3002     // ... backward masked reductions on the rank-1 gradient: 8 per pair and hidden column
};

extern "C" int pn_prof_begin(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  g_prof.clear();
  g_prof_on = true;
  return 0;
}

// Stops recording and aggregates per kernel kind.  Caller must have synchronised the stream(s).
extern "C" int pn_prof_end(int max_kinds, int* kinds, long* counts, double* total_ms, double* total_flops) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = false;
  int n = 0;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
    int k = -1;
    for (int i = 0; i < n; ++i)
      if (kinds[i] == r.kind) k = i;
    if (k < 0) {
      if (n >= max_kinds) continue;
      k = n++;
      kinds[k] = r.kind;
      counts[k] = 0;
      total_ms[k] = 0;
      total_flops[k] = 0;
    }
    counts[k] += 1;
    total_ms[k] += ms;
    total_flops[k] += r.flops;
    hipEventDestroy(r.e0);
    hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return n;
}

struct ProfScope {
  bool on;
  ProfRec r;
  hipStream_t st;
  ProfScope(int kind, double flops, hipStream_t s) : on(g_prof_on), st(s) {
    if (on) {
      r.kind = kind;
      r.flops = flops;
      hipEventCreate(&r.e0);
      hipEventCreate(&r.e1);
      hipEventRecord(r.e0, st);
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(r.e1, st);
      std::lock_guard<std::mutex> lk(g_prof_mu);
      if (g_prof.size() < 100000) g_prof.push_back(r);
      else { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// fixed-order reduction of per-workgroup partials (train_kernels.hpp): no floating-point atomics anywhere
// ------------------------------------------------------------------------------------------------
static inline unsigned nblk(long n, int t) { return (unsigned)((n + t - 1) / t); }
static const int RED_CHUNKS = 64;

template <class T>
static int reduce_parts(const T* part, long nparts, int width, int seg, double* d0, double* d1, double* d2,
                        double* red /* [RED_CHUNKS][width] */, hipStream_t st) {
  if (nparts <= 0) return fail("reduce_parts: no partials");
  long per = (nparts + RED_CHUNKS - 1) / RED_CHUNKS;
  const int nchunk = (int)((nparts + per - 1) / per);
  hipLaunchKernelGGL((k_part_reduce<T>), dim3(nblk(width, 256), nchunk), dim3(256), 0, st, part, nparts, width, per, red);
  hipLaunchKernelGGL(k_part_final, dim3(nblk(width, 256)), dim3(256), 0, st, (const double*)red, nchunk, width, seg, d0,
                     d1, d2);
  HIP_OK(hipGetLastError());
  return 0;
}

// scratch of the column statistics a GEMM epilogue produces (GemmParams::col_part / col_red)
struct ColScr {
  float* part;
  double* red;
};
static bool colscr_carve(Bump& bp, long M, int N, ColScr& c) {
  c.part = bp.take<float>((size_t)((M + 127) / 128) * 2 * (size_t)N);  // smallest row tile is 128
  c.red = bp.take<double>((size_t)RED_CHUNKS * 2 * (size_t)N);
  return bp.ok;
}
// scratch of k_bn_bwd_stats / k_colsum: [row chunks][3][C] partials + the level-1 output
struct StatScr {
  double* part;
  double* red;
};
static bool statscr_carve(Bump& bp, long R, long rows_per_block, int C, StatScr& s) {
  s.part = bp.take<double>((size_t)((R + rows_per_block - 1) / rows_per_block) * 3 * (size_t)C);
  s.red = bp.take<double>((size_t)RED_CHUNKS * 3 * (size_t)C);
  return bp.ok;
}

// column sum / sum of squares of what the GEMM just stored, from its per-row-tile partials
static int finish_col_stats(const GemmParams& p, long tm, hipStream_t st) {
  if (p.col_sum == nullptr) return 0;
  if (p.col_part == nullptr || p.col_red == nullptr) return fail("gemm: column statistics need col_part / col_red scratch");
  return reduce_parts<float>(p.col_part, tm, 2 * p.N, p.N, p.col_sum, p.col_sumsq, nullptr, p.col_red, st);
}

// ------------------------------------------------------------------------------------------------
// inter-layer dropout (training): mask = hash(seed ^ stream, row, column), see gemm_engine.hpp
// ------------------------------------------------------------------------------------------------
struct DropSpec {
  uint32_t seed, thresh;
  float scale;
};
static DropSpec drop_spec(float p, unsigned base_seed, int stream) {
  DropSpec d = {0u, 0u, 1.f};
  if (p <= 0.f) return d;
  d.seed = base_seed ^ ((uint32_t)stream * 0x9E3779B9u);
  double th = (double)p * 4294967296.0 + 0.5;
  d.thresh = th >= 4294967295.0 ? 0xffffffffu : (uint32_t)th;
  if (d.thresh == 0) d.thresh = 1;  // p > 0 must keep the "dropout on" meaning of a non-zero threshold
  d.scale = 1.f / (1.f - p);
  return d;
}
enum { DROP_STREAM_WP = 100, DROP_STREAM_WL = 200, DROP_STREAM_PAIR = 300, DROP_STREAM_OUT = 99 };

template <int MODE>
static int launch_dropout(const float* X, long ldx, float* out, long ldo, long R, int C, const float* s, const float* t,
                          const DropSpec& d, hipStream_t st) {
  if (C % 4) return fail("dropout: width %d not a multiple of 4", C);
  const long rpb = 256;
  hipLaunchKernelGGL((k_dropout<MODE>), dim3(nblk(C, 1024), nblk(R, rpb)), dim3(256), 0, st, X, ldx, out, ldo, R, C, s, t,
                     d.seed, d.thresh, d.scale, rpb);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_dropout_mask(unsigned seed, int stream, float p, long rows, int cols, float* out, void* stream_) {
  if (p <= 0.f || p >= 1.f) return fail("dropout_mask: p must be in (0, 1)");
  return launch_dropout<2>(nullptr, 0, out, cols, rows, cols, nullptr, nullptr, drop_spec(p, seed, stream),
                           (hipStream_t)stream_);
}

// ------------------------------------------------------------------------------------------------
// SYNC_BN (reference bin/main.py:449-450, nn.SyncBatchNorm.convert_sync_batchnorm): train-mode BatchNorm statistics over
// the batches of ALL ranks.  The statistics of a layer are reduced inside one C call, between the GEMM that accumulates
// them and the fold that consumes them, so the cross-rank sum is a callback: the host registers `hook(n, user)`, which
// must sum the first n doubles of the staging buffer over the ranks in place, ordered on the launch stream
// (protnote_amd.utils.distributed.enable_sync_batchnorm: torch.distributed.all_reduce = RCCL).  Forward: per-column
// [sum, sum of squares, row count] travel (the ranks' batch shapes may differ: see sync_sum2).  Backward: [sum du, sum du*xhat] travel for the dz generator's p / q
// vectors, while dgamma / dbeta stay the LOCAL sums (the gradient all-reduce averages them like every other
// gradient) - torch's SyncBatchNorm backward.
// ------------------------------------------------------------------------------------------------
typedef int (*pn_sync_hook_t)(long n_doubles, void* user);
static pn_sync_hook_t g_sync_hook = nullptr;
static void* g_sync_user = nullptr;
static double* g_sync_stage = nullptr;
static long g_sync_cap = 0;
static int g_sync_world = 1;

extern "C" int pn_set_sync_bn(pn_sync_hook_t hook, void* user, double* stage, long stage_doubles, int world) {
  if (hook == nullptr) {
    g_sync_hook = nullptr; g_sync_user = nullptr; g_sync_stage = nullptr; g_sync_cap = 0; g_sync_world = 1;
    return 0;
  }
  if (stage == nullptr || stage_doubles < 4096 || world < 1) return fail("pn_set_sync_bn: need a staging buffer of >= 4096 doubles and world >= 1");
  g_sync_hook = hook; g_sync_user = user; g_sync_stage = stage; g_sync_cap = stage_doubles; g_sync_world = world;
  return 0;
}
static bool sync_bn_on() { return g_sync_hook != nullptr && g_sync_world > 1; }
__global__ void k_set_double(double* p, double v) { *p = v; }
// a[0..n) and b[0..n) (device, f64) become their sums over the ranks, and so does this rank's row count: the ranks'
// batches differ in shape (each collator pads to its own batch maximum, collators.py:40, and the last batch of an epoch
// is ragged), so the global count is the SUM of the local ones (torch.nn.SyncBatchNorm all-gathers the counts), not
// local * world.  Returns (through count_dev) the device address of the global count, valid until the next call; the
// kernels that consume the statistics read it from there.  nullptr when SYNC_BN is off.
static int sync_sum2(double* a, double* b, long n, double local_count, const double** count_dev, hipStream_t st) {
  *count_dev = nullptr;
  if (!sync_bn_on()) return 0;
  if (2 * n + 2 > g_sync_cap) return fail("sync_bn: %ld statistics exceed the staging buffer (%ld doubles)", 2 * n + 2, g_sync_cap);
  HIP_OK(hipMemcpyAsync(g_sync_stage, a, n * sizeof(double), hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(g_sync_stage + n, b, n * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_set_double, dim3(1), dim3(1), 0, st, g_sync_stage + 2 * n, local_count);
  if (g_sync_hook(2 * n + 1, g_sync_user) != 0) return fail("sync_bn: the all-reduce callback failed");
  HIP_OK(hipMemcpyAsync(a, g_sync_stage, n * sizeof(double), hipMemcpyDeviceToDevice, st));
  HIP_OK(hipMemcpyAsync(b, g_sync_stage + n, n * sizeof(double), hipMemcpyDeviceToDevice, st));
  double* keep = g_sync_stage + g_sync_cap - 1;  // outside every staged range: survives until the next call
  HIP_OK(hipMemcpyAsync(keep, g_sync_stage + 2 * n, sizeof(double), hipMemcpyDeviceToDevice, st));
  *count_dev = keep;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GEMM launch
// ------------------------------------------------------------------------------------------------
template <int AK, int EK, int WAVES_M, int WAVES_N, int WM, int WN, int BK, bool DROP = false>
static int launch_gemm_cfg(const GemmParams& p, hipStream_t st) {
  using Cfg = GemmCfg<WAVES_M, WAVES_N, WM, WN, BK>;
  auto kern = gemm_nt_kernel<AK, EK, WAVES_M, WAVES_N, WM, WN, BK, DROP>;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                               Cfg::LDS_BYTES));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  if (p.Kseg % 4 != 0) return fail("gemm: K segment %d not a multiple of 4", p.Kseg);
  const long tm = (p.M + Cfg::BM - 1) / Cfg::BM;
  const long tn = (p.Nstore + Cfg::BN - 1) / Cfg::BN;
  if (tm * tn > 0x7fffffffL) return fail("gemm: grid too large");
  GemmParams pp = p;
  // XCD-aware block order for tall grids over h = 3072 columns: 64 (128x128 tiles, 2 workgroups/CU) or
  // 32 (256x256 tiles, 1 workgroup/CU) workgroups are resident per XCD
  const int resident = (Cfg::BM * Cfg::BN >= 256 * 256) ? 32 : 64;
  pp.xcd_bc = (PN_XCD && tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  // pair-sum operand (A from two small tables): W is the only streamed operand - one column tile per XCD block keeps
  // its 3 MB panel in that XCD's L2 (fabric fetch per launch 1.48 TB -> see profiles/hbm_traffic.json)
  if (AK == A_PAIRSUM_RELU && resident == 32 && pp.xcd_bc && tm >= 64) pp.xcd_bc = 1;
  pp.xcd_br = pp.xcd_bc ? resident / pp.xcd_bc : 0;
  long grid = tm * tn;
  if (pp.xcd_bc) {
    const long nblk = ((tm + pp.xcd_br - 1) / pp.xcd_br) * (tn / pp.xcd_bc);
    grid = ((nblk + 7) / 8) * 8 * resident;
  }
  if (grid > 0x7fffffffL) return fail("gemm: grid too large");
  {
    ProfScope ps(AK * 10 + EK, 2.0 * (double)p.M * (double)p.N * (double)p.nseg * (double)p.Kseg, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, pp);
  }
  HIP_OK(hipGetLastError());
  return finish_col_stats(p, tm, st);
}

// Arithmetic of the pair-grid GEMMs: 0 = exact f32 MFMA (default), 1 = bf16x3 split (gemm_bf16x3.hpp).
// Per CALL: every descriptor (pn_encoder / pn_mlp / pn_pairhead) carries `math_mode` (and pn_pairhead `backward_math`);
// the entry point that receives it installs the value for the duration of the call on the CALLING THREAD (MathScope), and
// the launchers read cur_math().  Two host threads driving two models on two streams can therefore run different modes.
// pn_set_math_mode / pn_set_backward_math only set the process DEFAULT that a descriptor field of 0 falls back to.
static std::atomic<int> g_math_mode{0};
extern "C" int pn_set_math_mode(int mode) {
  if (mode != 0 && mode != 1) return fail("pn_set_math_mode: 0 (f32) or 1 (bf16x3)");
  g_math_mode = mode;
  return 0;
}
extern "C" int pn_get_math_mode(void) { return g_math_mode; }
static thread_local int tl_math = -1;  // -1: no descriptor in scope -> the process default
static inline int cur_math() { return tl_math >= 0 ? tl_math : g_math_mode.load(std::memory_order_relaxed); }
struct MathScope {  // descriptor field: 0 = process default, 1 = f32, 2 = bf16x3
  int prev;
  explicit MathScope(int field) : prev(tl_math) {
    if (field == 1 || field == 2) tl_math = field - 1;
  }
  ~MathScope() { tl_math = prev; }
};
static int math_field_check(int field, const char* what) {
  if (field < 0 || field > 2) return fail("%s: math_mode %d (0 = library default, 1 = f32, 2 = bf16x3)", what, field);
  return 0;
}

// Arithmetic of the BACKWARD pair-grid GEMMs of the hidden layers (dW_l = dz_l^T h_{l-1} and dh_{l-1} = dz_l W_l, l >= 1):
// 0 = the forward's mode (default), 1 = ONE product of the bf16-rounded operands with f32 accumulation (the NP = 1 kernels
// of gemm_bf16x3.hpp) - the arithmetic class of the reference's autocast backward (ProtNoteTrainer.py:728-738).  The
// forward, every reduction, the BatchNorm backward and the row MLPs keep the forward's mode: logits are bit-identical.
// Per call through pn_pairhead.backward_math (0 = this process default, 1 = as the forward, 2 = bf16).
static std::atomic<int> g_bwd_math{0};
extern "C" int pn_set_backward_math(int mode) {
  if (mode != 0 && mode != 1) return fail("pn_set_backward_math: 0 (as the forward) or 1 (bf16, one product)");
  g_bwd_math = mode;
  return 0;
}
extern "C" int pn_get_backward_math(void) { return g_bwd_math; }
// set by pn_pairhead_bwd around the GEMMs it applies to; read by launch_gemm / launch_tn
static thread_local bool tl_bwd_bf16 = false;
// ... and the dz operand of those GEMMs is already bf16 in memory (bwd_bf16_dz.hpp; decided per layer by pn_pairhead_bwd)
static thread_local bool tl_dz_bf16 = false;
struct BwdBf16Scope {
  bool prev;
  explicit BwdBf16Scope(bool on) : prev(tl_bwd_bf16) { tl_bwd_bf16 = on; }
  ~BwdBf16Scope() {  // (also on the error returns of pn_pairhead_bwd: no flag outlives the call)
    tl_bwd_bf16 = prev;
    tl_dz_bf16 = false;
  }
};

// Arithmetic of the FORWARD pair-grid GEMMs of the hidden layers (z_l = h_{l-1} W_l^T, l >= 1): 0 = math_mode's kernels
// (default), 1 = ONE product of the bf16-rounded operands with f32 accumulation - the class of the reference's autocast forward
// (ProtNoteTrainer.py:287,728-729).  Per call through pn_pairhead.forward_math (0 = this process default, 1 = as math_mode,
// 2 = bf16).  The stored pre-activations, the BatchNorm statistics and the row-dot stay f32: the backward does not change.
static std::atomic<int> g_fwd_math{0};
extern "C" int pn_set_forward_math(int mode) {
  if (mode != 0 && mode != 1) return fail("pn_set_forward_math: 0 (as math_mode) or 1 (bf16, one product)");
  g_fwd_math = mode;
  return 0;
}
extern "C" int pn_get_forward_math(void) { return g_fwd_math; }
// set by the pn_pairhead_fwd_* entry points around the hidden layers' GEMMs only; read by launch_gemm
static thread_local bool tl_fwd_bf16 = false;
struct FwdBf16Scope {
  bool prev;
  explicit FwdBf16Scope(bool on) : prev(tl_fwd_bf16) { tl_fwd_bf16 = on; }
  ~FwdBf16Scope() { tl_fwd_bf16 = prev; }
};
static bool fwd_bf16_requested(const pn_pairhead* hd) {  // (workspace sizing: the same rule as fwd_math_of, no error path)
  const int m = hd->forward_math == 0 ? g_fwd_math.load(std::memory_order_relaxed) : hd->forward_math - 1;
  return m == 1 && hd->dropout_p == 0.f;
}
static int fwd_math_of(const pn_pairhead* hd, bool* bf16) {
  if (hd->forward_math < 0 || hd->forward_math > 2)
    return fail("pairhead: forward_math %d (0 = library default, 1 = as math_mode, 2 = bf16)", hd->forward_math);
  const int m = hd->forward_math == 0 ? g_fwd_math.load(std::memory_order_relaxed) : hd->forward_math - 1;
  *bf16 = (m == 1) && hd->dropout_p == 0.f;
  return 0;
}

// bf16x3 pair-grid GEMMs with the weight operand pre-split and staged by LDS-DMA; pn_set_b3_dma(0) keeps the register
// path (the bit-identity test compares the two)
static std::atomic<int> g_b3_dma{1};
static bool use_b3_dma() { return g_b3_dma == 1; }
extern "C" int pn_set_b3_dma(int on) {
  g_b3_dma = on ? 1 : 0;
  return 0;
}

template <int AK, int EK, int WAVES_N, int WN, bool GEN, bool BDMA = false, int NP = 3>
static int launch_gemm_bf16x3(const GemmParams& p, hipStream_t st) {
  using Cfg = GemmCfg<4, WAVES_N, 2, WN, 32>;
  if constexpr (!GEN && !BDMA && WAVES_N * WN * 32 == 256) {
    if (p.wsplit != nullptr && use_b3_dma() && p.Nstore == p.N)
      return launch_gemm_bf16x3<AK, EK, WAVES_N, WN, GEN, true>(p, st);
  }
  auto kern = gemm_nt_bf16x3_kernel<AK, EK, 4, WAVES_N, 2, WN, GEN, BDMA, NP>;
  constexpr int LDS_BYTES = BDMA ? 2 * (256 * 36 + 2 * 256 * 16) * (int)sizeof(float) : Cfg::LDS_BYTES;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  if (p.Kseg % 4 != 0) return fail("gemm: K segment %d not a multiple of 4", p.Kseg);
  const long tm = (p.M + Cfg::BM - 1) / Cfg::BM;
  const long tn = (p.Nstore + Cfg::BN - 1) / Cfg::BN;
  GemmParams pp = p;
  pp.xcd_bc = (PN_XCD && tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  // pair-sum operand: A comes from two small tables, so the only streamed operand is W - give each XCD ONE column
  // tile at a time (32 row tiles x 1): its 3 MB W panel stays in that XCD's L2 instead of 4 panels thrashing it
  // (fabric fetch per launch 0.8 TB -> W once per block; measured +1.4 % on the eval pair head)
  if (!GEN && AK == A_PAIRSUM_RELU && pp.xcd_bc && tm >= 64) pp.xcd_bc = 1;
  pp.xcd_br = pp.xcd_bc ? 32 / pp.xcd_bc : 0;
  long grid = tm * tn;
  if (pp.xcd_bc) {
    const long nblk = ((tm + pp.xcd_br - 1) / pp.xcd_br) * (tn / pp.xcd_bc);
    grid = ((nblk + 7) / 8) * 8 * 32;
  }
  if (grid > 0x7fffffffL) return fail("gemm: grid too large");
  if constexpr (BDMA) {  // W -> hi / lo bf16 planes [N][K], once per launch (37 MB for 3072 x 3072: ~20 us)
    pp.w_hi = p.wsplit;
    pp.w_lo = p.wsplit + (size_t)p.N * p.Kseg;
    hipLaunchKernelGGL(k_split_planes, dim3(nblk((long)p.N * p.Kseg / 4, 256)), dim3(256), 0, st, p.W, p.ldw, p.N, p.Kseg,
                       p.wsplit, p.wsplit + (size_t)p.N * p.Kseg);
  }
  {
    ProfScope ps((NP == 3 ? 1000 : 1500) + AK * 10 + EK, 2.0 * (double)p.M * (double)p.N * (double)p.nseg * (double)p.Kseg, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NT), LDS_BYTES, st, pp);
  }
  HIP_OK(hipGetLastError());
  return finish_col_stats(p, tm, st);
}

// dh = dz W of the bf16 backward on the deep-pipelined single-product kernel (gemm_bf16.hpp); preconditions checked by
// launch_gemm.  pn_set_bwd_deep(0) keeps the NP = 1 instantiation of the bf16x3 kernel (same products in the same order:
// the bit-identity test compares the two)
static std::atomic<int> g_bwd_deep{7};  // bit 0: the deep-pipelined dh kernel, bit 1: the transpose-read dW kernel, bit 2: dz stored as bf16
extern "C" int pn_set_bwd_deep(int mask) {
  g_bwd_deep = mask & 7;
  return 0;
}
static int launch_gemm_bf16_single(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_nt_bf16_kernel<E_STORE>;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NT_BF16_LDS_BYTES));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  const long tm = (p.M + 255) / 256, tn = p.N / 256;
  GemmParams pp = p;
  pp.xcd_bc = (PN_XCD && tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  pp.xcd_br = pp.xcd_bc ? 32 / pp.xcd_bc : 0;
  long grid = tm * tn;
  if (pp.xcd_bc) {
    const long nblk_ = ((tm + pp.xcd_br - 1) / pp.xcd_br) * (tn / pp.xcd_bc);
    grid = ((nblk_ + 7) / 8) * 8 * 32;
  }
  if (grid > 0x7fffffffL) return fail("gemm: grid too large");
  pp.w_hi = p.wsplit;  // W rounded to one bf16 plane [N][K], once per launch (k_split_planes' hi plane; its lo plane is unused)
  pp.w_lo = p.wsplit + (size_t)p.N * p.Kseg;
  hipLaunchKernelGGL(k_split_planes, dim3(nblk((long)p.N * p.Kseg / 4, 256)), dim3(256), 0, st, p.W, p.ldw, p.N, p.Kseg,
                     p.wsplit, p.wsplit + (size_t)p.N * p.Kseg);
  {
    ProfScope ps(1500 + A_PLAIN * 10 + E_STORE, 2.0 * (double)p.M * (double)p.N * (double)p.Kseg, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), NT_BF16_LDS_BYTES, st, pp);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// dh = dz W with dz stored as bf16 (bwd_bf16_dz.hpp): both operands by LDS-DMA
static int launch_gemm_bf16dma(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_nt_bf16dma_kernel<E_STORE>;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NT_BF16DMA_LDS_BYTES));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  if (p.wsplit == nullptr || p.Kseg % 64 != 0 || p.N % 256 != 0 || p.Nstore != p.N || (long)256 * p.lda * 4 >= (1L << 32) ||
      (long)256 * p.Kseg * 2 >= (1L << 32))
    return fail("gemm (bf16 dz): shape %d x %d x %d not supported", p.M, p.N, p.Kseg);
  const long tm = (p.M + 255) / 256, tn = p.N / 256;
  GemmParams pp = p;
  pp.xcd_bc = (PN_XCD && tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  pp.xcd_br = pp.xcd_bc ? 32 / pp.xcd_bc : 0;
  long grid = tm * tn;
  if (pp.xcd_bc) {
    const long nblk_ = ((tm + pp.xcd_br - 1) / pp.xcd_br) * (tn / pp.xcd_bc);
    grid = ((nblk_ + 7) / 8) * 8 * 32;
  }
  if (grid > 0x7fffffffL) return fail("gemm: grid too large");
  pp.w_hi = p.wsplit;
  hipLaunchKernelGGL(k_round_plane, dim3(nblk((long)p.N * p.Kseg / 4, 256)), dim3(256), 0, st, p.W, p.ldw, p.N, p.Kseg, p.wsplit);
  {
    ProfScope ps(1500 + A_PLAIN * 10 + E_STORE, 2.0 * (double)p.M * (double)p.N * (double)p.Kseg, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), NT_BF16DMA_LDS_BYTES, st, pp);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

#ifndef PN_BIG
#define PN_BIG 1
#endif
// ---- AMP-class forward with the activation operand materialised as bf16 (fwd_bf16_h.hpp) ----
// pn_set_fwd_staged(0) keeps the register-staged single-product kernels (the NP = 1 instantiations of gemm_bf16x3.hpp) for the
// forward: same bf16 values in the same products, another k order inside a 16-k MFMA step (A/B timing, and the test that holds
// the two routes to each other).
static std::atomic<int> g_fwd_staged{1};
extern "C" int pn_set_fwd_staged(int on) {
  g_fwd_staged = on ? 1 : 0;
  return 0;
}
static thread_local bool tl_fwd_nostage = false;  // pn_pairhead_fwd_eval_hidden reads f32 activations back: register-staged route
static const long FWD_H_ROWS = 262144;            // pair rows per materialised chunk (a multiple of the 256-row tile)
static bool fwd_staged_shape(int h) { return PN_BIG && h % 256 == 0 && h >= 256 && h <= 8192; }
static bool fwd_staged_on(bool fwd_bf16, int h) { return fwd_bf16 && g_fwd_staged == 1 && !tl_fwd_nostage && fwd_staged_shape(h); }

// h (bf16, [rows][C]) for pair rows [r0, r0 + rows): kind 0 = relu(A'[i] + B'[j]), 1 = relu(s z + t), 2 = round(z)
static int make_h(int kind, long r0, long rows, int C, const float* A, long lda, const float* A2, long lda2, int pairB,
                  const float* s, const float* t, uint16_t* out, hipStream_t st) {
  if (rows <= 0) return 0;
  MakeHParams P;
  memset(&P, 0, sizeof(P));
  P.r0 = r0; P.rows = rows; P.C = C; P.pairB = pairB > 0 ? pairB : 1; P.A = A; P.lda = lda; P.A2 = A2; P.lda2 = lda2;
  P.s = s; P.t = t; P.out = out;
  const int rpb = 16;
  const dim3 grid(nblk(rows, rpb)), block(C / 8);
  // algorithmic bytes: 2 B written per element; kinds 1 / 2 also read the 4 B pre-activation (kind 0 reads L2-resident tables)
  ProfScope ps(ST_MAKE_H, (double)rows * (double)C * (kind == 0 ? 2.0 : 6.0), st);
  if (kind == 0) hipLaunchKernelGGL((k_make_h_bf16<0>), grid, block, 0, st, P, rpb);
  else if (kind == 1) hipLaunchKernelGGL((k_make_h_bf16<1>), grid, block, 0, st, P, rpb);
  else hipLaunchKernelGGL((k_make_h_bf16<2>), grid, block, 0, st, P, rpb);
  HIP_OK(hipGetLastError());
  return 0;
}

// C = H W^T, H = bf16 [M][K] dense (make_h / an E_STORE_H16 producer), W = the pre-rounded plane p.w_hi (k_round_plane):
// gemm_nt_bf16dma_kernel with epilogue EK.  src_kind (0 plain, 1 bn_relu, 2 pairsum) only labels the profile kind.
template <int EK>
static int launch_gemm_h16(const GemmParams& p, int src_kind, hipStream_t st) {
  auto kern = gemm_nt_bf16dma_kernel<EK>;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NT_BF16DMA_LDS_BYTES));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  if (p.w_hi == nullptr || p.Kseg % 64 != 0 || p.N % 256 != 0 || p.Nstore != p.N || (long)256 * p.lda * 4 >= (1L << 32) ||
      (long)256 * p.Kseg * 2 >= (1L << 32))
    return fail("gemm (bf16 h): shape %d x %d x %d not supported", p.M, p.N, p.Kseg);
  const long tm = (p.M + 255) / 256, tn = p.N / 256;
  GemmParams pp = p;
  pp.xcd_bc = (PN_XCD && tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  pp.xcd_br = pp.xcd_bc ? 32 / pp.xcd_bc : 0;
  long grid = tm * tn;
  if (pp.xcd_bc) {
    const long nblk_ = ((tm + pp.xcd_br - 1) / pp.xcd_br) * (tn / pp.xcd_bc);
    grid = ((nblk_ + 7) / 8) * 8 * 32;
  }
  if (grid > 0x7fffffffL) return fail("gemm: grid too large");
  {
    // (1700 + ...: 1500 + kind covers the single-product NT kinds 0..59 AND the TN kinds 100..122 = 1600..1622)
    ProfScope ps(1700 + src_kind * 10 + EK, 2.0 * (double)p.M * (double)p.N * (double)p.Kseg, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), NT_BF16DMA_LDS_BYTES, st, pp);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// f32 pair-grid GEMMs with LDS-DMA operand staging (gemm_dma.hpp); pn_set_f32_dma(0) selects the register-staged engine
// (the bit-identity tests compare the two)
// float64 accumulation in the forward convolutions of a trainable encoder (gemm_conv_f64.hpp): on by default; the switch
// exists for the A/B measurement (tools/encoder_grad_error.py) and the test that shows what it buys
static std::atomic<int> g_enc_f64{1};
extern "C" int pn_set_encoder_f64(int on) {
  g_enc_f64 = on ? 1 : 0;
  return 0;
}

// conv1 on one-hot input as a gather-sum (k_conv1_gather); 0 = always the general convolution (A/B, bit-identity test)
static std::atomic<int> g_conv1_gather{1};
extern "C" int pn_set_conv1_gather(int on) {
  g_conv1_gather = on ? 1 : 0;
  return 0;
}

static std::atomic<int> g_f32_dma{1};
static bool use_f32_dma() { return g_f32_dma == 1; }

extern "C" int pn_set_f32_dma(int on) {
  g_f32_dma = on ? 1 : 0;
  return 0;
}

// smallest M for which an NT GEMM takes the 256-tile LDS-DMA kernel
static const int g_dma_min_rows = 16384;

template <int AK, int EK, bool DROP = false>
static int launch_gemm_dma(const GemmParams& p, hipStream_t st) {
  auto kern = gemm_nt_dma_kernel<AK, EK, DROP>;
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_DMA_LDS_BYTES));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  const long tm = (p.M + 255) / 256, tn = p.N / 256;
  GemmParams pp = p;
  pp.xcd_bc = (PN_XCD && tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  if (AK == A_PAIRSUM_RELU && pp.xcd_bc && tm >= 64) pp.xcd_bc = 1;  // W panel resident in one XCD's L2 (see launch_gemm_cfg)
  pp.xcd_br = pp.xcd_bc ? 32 / pp.xcd_bc : 0;
  long grid = tm * tn;
  if (pp.xcd_bc) {
    const long nblk_ = ((tm + pp.xcd_br - 1) / pp.xcd_br) * (tn / pp.xcd_bc);
    grid = ((nblk_ + 7) / 8) * 8 * 32;
  }
  if (grid > 0x7fffffffL) return fail("gemm: grid too large");
  {
    ProfScope ps(AK * 10 + EK, 2.0 * (double)p.M * (double)p.N * (double)p.Kseg, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), GEMM_DMA_LDS_BYTES, st, pp);
  }
  HIP_OK(hipGetLastError());
  return finish_col_stats(p, tm, st);
}

// Encoder convolution as an all-LDS-DMA implicit GEMM (gemm_conv_dma.hpp): p.A / p.lda describe the staged activation H,
// p.W / p.ldw the re-laid weights, p.Kseg = Kpad (a multiple of 32).  One column tile per XCD block: the 32 row tiles an
// XCD runs at a time stream the same weight panel (7.6 MB for 192 x 9 x 1100) through its L2 in step.
static int launch_conv_dma(const GemmParams& p, int Lp, int ktrue, hipStream_t st) {
  constexpr int WN = 3;
  auto kern = gemm_conv_dma_kernel<WN>;
  constexpr int LDS = conv_dma_lds_bytes<WN>();
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done[dev] = true;
  }
  if (p.M <= 0 || p.Nstore <= 0) return 0;
  if (p.Kseg % 32 != 0) return fail("conv_dma: K segment %d not a multiple of 32", p.Kseg);
  const long tm = (p.M + 255) / 256, tn = (p.Nstore + 64 * WN - 1) / (64 * WN);
  ConvDmaParams cp;
  cp.g = p;
  cp.Lp = Lp;
  cp.g.xcd_bc = (PN_XCD && tm >= 64) ? 1 : 0;
  cp.g.xcd_br = cp.g.xcd_bc ? 32 : 0;
  long grid = tm * tn;
  if (cp.g.xcd_bc) {
    const long nblk_ = ((tm + 31) / 32) * tn;
    grid = ((nblk_ + 7) / 8) * 8 * 32;
  }
  if (grid > 0x7fffffffL) return fail("conv_dma: grid too large");
  {
    ProfScope ps(A_CONV * 10 + E_CONV, 2.0 * (double)p.M * (double)p.N * (double)p.nseg * (double)ktrue, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), LDS, st, cp);
  }
  HIP_OK(hipGetLastError());
  return finish_col_stats(p, tm, st);
}

static int rowdot_nparts(int n);
// variant 0: 128x128 tile (2x2 waves of 64x64); variant 1: 128x64 tile (4x1 waves of 32x64);
// variant 2: 256x256 tile (4x2 waves of 64x128, one workgroup per CU) - half the operand traffic per flop
#ifndef PN_BIG
#define PN_BIG 1
#endif
// the LDS-DMA kernels address their operands as SGPR base + 32-bit per-lane byte offset: a 256-row tile of either
// operand (and, for the pair sum, each of the two tables from its origin) must span less than 4 GB
template <int AK>
static bool dma_offsets_fit(const GemmParams& p) {
  const long lim = 1L << 32;
  if ((long)256 * p.lda * 4 >= lim || (long)256 * p.ldw * 4 >= lim) return false;
  if (AK == A_PAIRSUM_RELU)
    return p.pairB > 0 && p.lda2 % 4 == 0 && (long)p.pairB * p.lda * 4 < lim && ((long)p.M / p.pairB + 1) * p.lda2 * 4 < lim;
  return true;
}

template <int AK, int EK>
static int launch_gemm(const GemmParams& p, int variant, hipStream_t st) {
  if (p.drop_thresh != 0) {  // dropped hidden activations (training): f32 engines with the mask in the A loader
    if constexpr ((AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU) && EK == E_STORE) {
      if (PN_BIG && use_f32_dma() && variant == 0 && p.M >= 65536 && p.nseg == 1 && p.Kseg % 32 == 0 && p.N % 256 == 0 &&
          p.Nstore == p.N && p.lda % 4 == 0 && p.ldw % 4 == 0 && dma_offsets_fit<AK>(p))
        return launch_gemm_dma<AK, EK, true>(p, st);
      return launch_gemm_cfg<AK, EK, 2, 2, 2, 2, PN_BK, true>(p, st);
    } else {
      return fail("gemm: dropout is not defined for operand kind %d / epilogue %d", AK, EK);
    }
  }
  if (tl_fwd_bf16 && PN_BIG) {  // forward_math = bf16: z_l = h_{l-1} W_l^T on one bf16 product (weight plane by LDS-DMA)
    if constexpr ((AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU) && (EK == E_STORE || EK == E_ROWDOT)) {
      if (variant == 0 && p.nseg == 1 && p.Kseg % 32 == 0 && p.N % 256 == 0 && p.Nstore == p.N && p.wsplit != nullptr &&
          p.lda % 4 == 0 && (AK != A_PAIRSUM_RELU || p.lda2 % 4 == 0))
        return launch_gemm_bf16x3<AK, EK, 2, 4, false, true, 1>(p, st);
    }
  }
  if (tl_bwd_bf16 && PN_BIG) {  // pn_set_backward_math(1): dh = dz W on one bf16 product (weight plane by LDS-DMA)
    if constexpr (AK == A_PLAIN && EK == E_STORE) {
      if (tl_dz_bf16) return launch_gemm_bf16dma(p, st);
      if (variant == 0 && p.M >= 4096 && p.nseg == 1 && p.Kseg % 32 == 0 && p.N % 256 == 0 && p.Nstore == p.N &&
          p.wsplit != nullptr && p.bias == nullptr && p.e_scale == nullptr && p.col_part == nullptr) {
        if ((g_bwd_deep & 1) && p.Kseg >= 96 && (long)256 * p.lda * 4 < (1L << 32) && p.lda % 4 == 0)
          return launch_gemm_bf16_single(p, st);
        return launch_gemm_bf16x3<AK, EK, 2, 4, false, true, 1>(p, st);
      }
    }
  }
  if (cur_math() == 1 && PN_BIG) {  // opt-in bf16x3 arithmetic (gemm_bf16x3.hpp)
    if constexpr ((AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU) && (EK == E_STORE || EK == E_ROWDOT)) {
      // pair-grid shapes: no masks needed
      if (variant == 0 && p.M >= 65536 && p.nseg == 1 && p.Kseg % 32 == 0 && p.N % 256 == 0 && p.Nstore == p.N)
        return launch_gemm_bf16x3<AK, EK, 2, 4, false>(p, st);
    }
    if constexpr ((AK == A_PLAIN || AK == A_AFFINE_RELU) && EK == E_STORE) {  // row MLPs over the label set (W_l)
      if (variant == 0 && p.M >= 8192 && p.N >= 512) return launch_gemm_bf16x3<AK, EK, 2, 4, true>(p, st);
    }
    if constexpr (AK == A_CONV && EK == E_CONV) {  // encoder convolutions, 256x192 tiles
      if (variant == 3) return launch_gemm_bf16x3<AK, EK, 2, 3, true>(p, st);
    }
  }
  if constexpr ((AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU) && (EK == E_STORE || EK == E_ROWDOT)) {
    // (row MLPs over the label table, M = N_L = 32102, take it too: 126 x 12 tiles = 5.9 rounds of 256 workgroups)
    if (PN_BIG && use_f32_dma() && (variant == 0 || EK == E_ROWDOT) && p.M >= g_dma_min_rows && p.nseg == 1 && p.Kseg % 32 == 0 &&
        p.N % 256 == 0 && p.Nstore == p.N && p.lda % 4 == 0 && p.ldw % 4 == 0 && dma_offsets_fit<AK>(p))
      return launch_gemm_dma<AK, EK>(p, st);
  }
  if constexpr ((EK == E_STORE && (AK == A_PLAIN || AK == A_AFFINE_RELU || AK == A_PAIRSUM_RELU)) ||
                (EK == E_PAIRADD && AK == A_PAIRPROD)) {
    if (PN_BIG && variant == 0 && p.N % 256 == 0 && p.M >= 65536)
      return launch_gemm_cfg<AK, EK, 4, 2, 2, 4, 32>(p, st);
  }
  if constexpr (EK == E_CONV) {  // variant 3: 256x192 tile (4x2 waves of 64x96): 550 -> 3 x 192 = 576, 1100 -> 6 x 192
    if (PN_BIG && variant == 3) return launch_gemm_cfg<AK, EK, 4, 2, 2, 3, 32>(p, st);
  }
  if constexpr (EK == E_ROWDOT) {  // partial-slab count depends on the tile: decided by N alone (rowdot_nparts)
    if (PN_BIG && p.N % 256 == 0) return launch_gemm_cfg<AK, EK, 4, 2, 2, 4, 32>(p, st);
  }
  if (variant == 1) return launch_gemm_cfg<AK, EK, 4, 1, 1, 2, PN_BK>(p, st);
  return launch_gemm_cfg<AK, EK, 2, 2, 2, 2, PN_BK>(p, st);
}

static int rowdot_nparts(int n) {  // column tiles x WAVES_N partial slabs written by the E_ROWDOT epilogue
  return (PN_BIG && n % 256 == 0) ? (n / 256) * 2 : ((n + 127) / 128) * 2;
}

static int pick_variant(int n) {
  const int n128 = (n + 127) / 128 * 128, n64 = (n + 63) / 64 * 64;
  return (n128 * 100 > n64 * 108) ? 1 : 0;
}

static GemmParams gp_zero() {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.nseg = 1;
  p.alpha = 1.f;
  p.pairB = 1;
  p.L = 1;
  return p;
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_lens32(const int64_t* lens, int* out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = (int)lens[i];
}

// [B][Cin][L] f32 -> channels-last [B*L][ld], padding (t >= len) and pad lanes zeroed
// (first half of MaskedConv1D.forward, protein_encoders.py:14)
__global__ void k_ncl_to_nlc(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ out,
                             int B, int Cin, int L, int ld, const int* __restrict__ run_if) {
  if (run_if != nullptr && *run_if == 0) return;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)B * L) return;
  const int b = (int)(p / L), t = (int)(p - (long)b * L);
  const bool live = t < lens[b];
  const float* src = x + (long)b * Cin * L + t;
  float* dst = out + p * ld;
  for (int c = 0; c < ld; ++c) dst[c] = (live && c < Cin) ? src[(long)c * L] : 0.f;
}

__global__ void k_pack_conv(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int k,
                            int ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)Cout * k * ld;
  if (i >= total) return;
  const int c = (int)(i % ld);
  const int tap = (int)((i / ld) % k);
  const int co = (int)(i / ((long)ld * k));
  packed[i] = (c < Cin) ? w[((long)co * Cin + c) * k + tap] : 0.f;
}

// eval-mode BatchNorm folded to y = x*s + t; pad lanes (c >= C) get s = t = 0
__global__ void k_bn_fold_eval(pn_bn bn, const float* lin_bias, float eps, int C, int ld, float* s, float* t) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  float sc = 0.f, sh = 0.f;
  if (c < C) {
    if (bn.weight != nullptr) {
      sc = bn.weight[c] / sqrtf(bn.running_var[c] + eps);
      sh = bn.bias[c] - bn.running_mean[c] * sc;
    } else {  // no BatchNorm: identity scale, optional Linear bias
      sc = 1.f;
      sh = lin_bias ? lin_bias[c] : 0.f;
    }
  }
  s[c] = sc;
  t[c] = sh;
}

// train-mode BatchNorm from accumulated column sums: batch mean / biased variance for normalisation,
// running stats updated with momentum and the unbiased variance (torch.nn.BatchNorm1d semantics).
__global__ void k_bn_fold_train(pn_bn bn, const double* sum, const double* sumsq, double count,
                                const double* count_dev, float eps, float momentum, int C, int ld, float* s, float* t,
                                float* mean_out, float* invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  if (count_dev != nullptr) count = *count_dev;  // SYNC_BN: the rows of all ranks
  float sc = 0.f, sh = 0.f;
  if (c < C) {
    const double mean = sum[c] / count;
    double var = sumsq[c] / count - mean * mean;
    if (var < 0) var = 0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    sc = bn.weight[c] * invstd;
    sh = bn.bias[c] - (float)mean * sc;
    const double unb = count > 1 ? var * (count / (count - 1.0)) : var;
    bn.running_mean[c] = (1.f - momentum) * bn.running_mean[c] + momentum * (float)mean;
    bn.running_var[c] = (1.f - momentum) * bn.running_var[c] + momentum * (float)unb;
    if (mean_out) mean_out[c] = (float)mean;
    if (invstd_out) invstd_out[c] = invstd;
  } else {  // pad lane: keep every saved vector finite
    if (mean_out) mean_out[c] = 0.f;
    if (invstd_out) invstd_out[c] = 0.f;
  }
  s[c] = sc;
  t[c] = sh;
}

// fold of a train-mode BatchNorm from this rank's column sums; with SYNC_BN the sums of all ranks (see pn_set_sync_bn)
// BatchNorm in EVAL mode inside a differentiable forward (model.eval() with autograd on: reference ProtNote.forward has
// no mode restriction, ProtNote.py:243-309): the fold comes from the running statistics, nothing is updated, and the
// saved mean / invstd are the running ones, so the backward's xhat, dgamma, dbeta follow - only the batch-statistics
// terms (p, q of the dz generator) vanish.  Selected per call by the descriptor's bn_use_running field; the exported
// functions run in one host thread each, so a thread-local carries it to the fold / finalise helpers.
static thread_local bool tl_bn_running = false;
struct BnMode {
  bool prev;
  explicit BnMode(bool on) : prev(tl_bn_running) { tl_bn_running = on; }
  ~BnMode() { tl_bn_running = prev; }
};

__global__ void k_bn_fold_running(pn_bn bn, float eps, int C, int ld, float* s, float* t, float* mean_out,
                                  float* invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ld) return;
  float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
  if (c < C) {
    mu = bn.running_mean[c];
    is = 1.f / sqrtf(bn.running_var[c] + eps);
    sc = bn.weight[c] * is;
    sh = bn.bias[c] - mu * sc;
  }
  s[c] = sc;
  t[c] = sh;
  if (mean_out) mean_out[c] = mu;
  if (invstd_out) invstd_out[c] = is;
}

static int fold_train(hipStream_t st, pn_bn bn, const double* sum, const double* sumsq, double count, float eps,
                      float momentum, int C, int ld, float* s, float* t, float* mean_out, float* invstd_out) {
  if (tl_bn_running) {
    hipLaunchKernelGGL(k_bn_fold_running, dim3(nblk(ld, 256)), dim3(256), 0, st, bn, eps, C, ld, s, t, mean_out,
                       invstd_out);
    HIP_OK(hipGetLastError());
    return 0;
  }
  const double* gcount = nullptr;
  PN_OK(sync_sum2(const_cast<double*>(sum), const_cast<double*>(sumsq), C, count, &gcount, st));
  hipLaunchKernelGGL(k_bn_fold_train, dim3(nblk(ld, 256)), dim3(256), 0, st, bn, sum, sumsq, count, gcount, eps,
                     momentum, C, ld, s, t, mean_out, invstd_out);
  HIP_OK(hipGetLastError());
  return 0;
}

// masked mean over positions (protein_encoders.py:114-117); x is channels-last and already zero at pads
__global__ void k_pool(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ emb, int L,
                       int C, int ldx, int ld_emb) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int len = lens[b];
  const float* src = x + (long)b * L * ldx + c;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int t = 0;
  for (; t + 3 < len; t += 4) {
    a0 += src[(long)t * ldx];
    a1 += src[(long)(t + 1) * ldx];
    a2 += src[(long)(t + 2) * ldx];
    a3 += src[(long)(t + 3) * ldx];
  }
  for (; t < len; ++t) a0 += src[(long)t * ldx];
  emb[(long)b * ld_emb + c] = ((a0 + a1) + (a2 + a3)) / (float)len;
}

// out[r][c] = in[r][c]*s[c] + (t ? t[c] : 0)
__global__ void k_affine_rows(const float* __restrict__ in, long ldi, float* __restrict__ out, long ldo, long rows,
                              int cols, const float* __restrict__ s, const float* __restrict__ t) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const long r = i / cols;
  const int c = (int)(i - r * cols);
  out[r * ldo + c] = fmaf(in[r * ldi + c], s[c], t ? t[c] : 0.f);
}

// concatenation_diff: W1 = [W1a | W1b | W1c] acting on [P, L, P-L]  ->  effective [W1a+W1c | W1b-W1c]
__global__ void k_diff_weight(const float* __restrict__ w, float* __restrict__ out, int h, int d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)h * 2 * d) return;
  const int n = (int)(i / (2 * d));
  const int k = (int)(i - (long)n * 2 * d);
  const float* row = w + (long)n * 3 * d;
  out[i] = (k < d) ? row[k] + row[2 * d + k] : row[k] - row[2 * d + (k - d)];
}

__global__ void k_rowdot_reduce(const float* __restrict__ partials, int nparts, long M, const float* __restrict__ b,
                                float* __restrict__ out) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  float a = 0.f;
  for (int q = 0; q < nparts; ++q) a += partials[(long)q * M + r];
  out[r] = a + (b ? b[0] : 0.f);
}

// pair logits (label-major: r = j*B + i) -> out[i][n_out]
__global__ void k_ensemble(const float* __restrict__ pairs, int B, int NL, int ndesc, int pmajor,
                           float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nout = NL / ndesc;
  if (idx >= (long)B * nout) return;
  const int i = (int)(idx % B);
  const int jo = (int)(idx / B);
  float v;
  if (ndesc == 1) {
    v = pmajor ? pairs[(long)i * NL + jo] : pairs[(long)jo * B + i];
  } else {
    float acc = 0.f;
    for (int d = 0; d < ndesc; ++d) {
      const long j = (long)jo * ndesc + d;
      const float x = pmajor ? pairs[(long)i * NL + j] : pairs[j * B + i];
      acc += 1.f / (1.f + expf(-x));
    }
    float pm = acc / (float)ndesc;
    const float eps = 1e-7f;  // torch.special.logit(eps=1e-7) clamps to [eps, 1-eps]
    pm = fminf(fmaxf(pm, eps), 1.f - eps);
    v = logf(pm / (1.f - pm));
  }
  out[(long)i * nout + jo] = v;
}

// backward of the ensembling (autograd of ProtNote.py:313-322 when an eval-mode forward is differentiated): x [B][NL]
// protein-major logits of the description rows, dout [B][NL / ndesc] -> dx [B][NL];
// d logit(clamp(pm)) / dx_k = [eps <= pm <= 1 - eps] / (pm (1 - pm)) * sigma'(x_k) / ndesc
__global__ void k_ensemble_bwd(const float* __restrict__ x, const float* __restrict__ dout, int B, int NL, int ndesc,
                               float* __restrict__ dx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nout = NL / ndesc;
  if (idx >= (long)B * nout) return;
  const int i = (int)(idx / nout), jo = (int)(idx % nout);
  const float* xr = x + (long)i * NL + (long)jo * ndesc;
  float* dr = dx + (long)i * NL + (long)jo * ndesc;
  const float g = dout[(long)i * nout + jo];
  if (ndesc == 1) {
    dr[0] = g;
    return;
  }
  float acc = 0.f;
  for (int d = 0; d < ndesc; ++d) acc += 1.f / (1.f + expf(-xr[d]));
  const float pm = acc / (float)ndesc, eps = 1e-7f;
  const float outer = (pm < eps || pm > 1.f - eps) ? 0.f : g / (pm * (1.f - pm)) / (float)ndesc;
  for (int d = 0; d < ndesc; ++d) {
    const float sg = 1.f / (1.f + expf(-xr[d]));
    dr[d] = outer * sg * (1.f - sg);
  }
}

// 1 / max(||x_r||_2, 1e-12)  (F.normalize, ProtNote.py:282-283); one wave per row
__global__ void k_rownorm_inv(const float* __restrict__ x, long ld, int rows, int d, float* __restrict__ out) {
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float a = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = x[(long)r * ld + c];
    a = fmaf(v, v, a);
  }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) out[r] = 1.f / fmaxf(sqrtf(a), 1e-12f);
}

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
  if constexpr (TR) kern = gemm_tn_bf16tr_kernel<TB, ABF16, SYNC>;
  else kern = gemm_tn_bf16x3_kernel<TB, NP>;
  constexpr int LDS = TR ? TN_BF16TR_LDS_BYTES : 2 * 512 * 36 * (int)sizeof(float);
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the call
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done[dev] = true;
  }
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
