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
#include "gemm_bf16_m16.hpp"

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

// The all-DMA one-product bf16 NT GEMMs run on v_mfma_f32_16x16x32_bf16 (gemm_bf16_m16.hpp: the power-efficient shape on this
// package); pn_set_bf16_mfma16(0) selects the 32 x 32 x 16 kernel of bwd_bf16_dz.hpp (same accumulators bit for bit; A/B, tests)
static std::atomic<int> g_bf16_m16{1};
extern "C" int pn_set_bf16_mfma16(int on) {
  g_bf16_m16 = on ? 1 : 0;
  return 0;
}
// both kernels of an epilogue get their dynamic-LDS attribute once per device; returns the one to launch
template <int EK>
static int bf16dma_kernel(void (**out)(GemmParams)) {
  static std::atomic<bool> attr_done[64];  // zero-initialised; hipFuncSetAttribute is idempotent, the flag only saves the calls
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 64 && !attr_done[dev]) {
    HIP_OK(hipFuncSetAttribute((const void*)gemm_nt_bf16dma_kernel<EK>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_BF16DMA_LDS_BYTES));
    HIP_OK(hipFuncSetAttribute((const void*)gemm_nt_bf16m16_kernel<EK>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_BF16DMA_LDS_BYTES));
    attr_done[dev] = true;
  }
  *out = g_bf16_m16.load() ? gemm_nt_bf16m16_kernel<EK> : gemm_nt_bf16dma_kernel<EK>;
  return 0;
}

// dh = dz W with dz stored as bf16 (bwd_bf16_dz.hpp): both operands by LDS-DMA
static int launch_gemm_bf16dma(const GemmParams& p, hipStream_t st) {
  void (*kern)(GemmParams) = nullptr;
  if (int rc = bf16dma_kernel<E_STORE>(&kern)) return rc;
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
// gemm_nt_bf16m16_kernel (pn_set_bf16_mfma16(0): gemm_nt_bf16dma_kernel) with epilogue EK.  src_kind (0 plain, 1 bn_relu, 2 pairsum) only labels the profile kind.
template <int EK>
static int launch_gemm_h16(const GemmParams& p, int src_kind, hipStream_t st) {
  void (*kern)(GemmParams) = nullptr;
  if (int rc = bf16dma_kernel<EK>(&kern)) return rc;
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

// The entry points, in the order they were written in this translation unit (each part uses the static helpers above it):
#include "abi_encoder.hpp"
#include "abi_heads_eval.hpp"
#include "abi_heads_train.hpp"
#include "abi_train_misc.hpp"
