// Evaluation metrics on the device: exact average precision per label and over all pairs, and
// the binned-threshold AUPRC estimate, over score matrices that stay resident in HBM for the whole evaluation.
//
// Reference call sites: ProtNoteTrainer.py:477-485 (BinaryAUPRC / MultilabelAUPRC on the CPU when ESTIMATE_MAP is
// False - every batch is copied D2H at :540-543 - or Binary/MultilabelBinnedAUPRC(threshold=50) on the device when
// True), utils/evaluation.py:148-169 (torchmetrics AveragePrecision).  torcheval 0.0.7 / torchmetrics 1.2.0 are not
// part of the reference tree; their published definition (precision-recall curve over the distinct score thresholds,
// Riemann sum of precision over recall) is what is restated here and pinned by oracle/ + sklearn in the tests.
//
// Layout: the accumulator is label-major, keys[N_L][cap] (u32, order-preserving image of the f32 score) and
// hits[N_L][cap] (u8), so that one label's scores are one contiguous segment; a batch [B, N_L] is transposed into
// column block [n0, n0+B) as it arrives (pn_ap_append).  The only library primitive is rocPRIM's (segmented) radix
// sort; key building, tie-grouped precision scan, binned histograms are the kernels below.  All HBM-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "../../include/protnote_hip.h"
#include "common.hpp"

namespace {

using pn::fail_msg;

#define HIP_OK(expr)                                                                             \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return fail_msg("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                          __FILE__, __LINE__);                                   \
  } while (0)

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// f32 -> u32 such that unsigned order == float order (-0 < +0 is the one distinction floats do not make; scores are
// sigmoid outputs or logits, and -0.0 is mapped onto +0.0 first so ties stay ties).
__device__ __forceinline__ uint32_t order_key(float s) {
  if (s == 0.f) s = 0.f;
  uint32_t u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint8_t label_hit(const void* y, int kind, long idx) {
  switch (kind) {
    case PN_LABEL_F32: return ((const float*)y)[idx] > 0.5f;
    case PN_LABEL_I64: return ((const long long*)y)[idx] != 0;
    default: return ((const uint8_t*)y)[idx] != 0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// append: [B rows, N_L labels] batch -> label-major accumulator columns [n0, n0+B).  64x64 tile through LDS so both
// the read (along labels) and the write (along proteins) are contiguous.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ap_append(const float* __restrict__ scores, int ld_s, const void* __restrict__ labels,
                                                   int kind, int ld_y, int B, int NL, uint32_t* __restrict__ keys,
                                                   uint8_t* __restrict__ hits, long cap, long n0) {
  __shared__ uint32_t tk[64][65];
  __shared__ uint8_t th[64][68];
  const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int i = i0 + r, j = j0 + tx;
    if (i < B && j < NL) {
      tk[r][tx] = order_key(scores[(long)i * ld_s + j]);
      th[r][tx] = label_hit(labels, kind, (long)i * ld_y + j);
    }
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int j = j0 + r, i = i0 + tx;
    if (i < B && j < NL) {
      keys[(long)j * cap + n0 + i] = tk[tx][r];
      hits[(long)j * cap + n0 + i] = th[tx][r];
    }
  }
}

// rows of length n at stride cap -> stride n
__global__ void k_ap_compact(const uint32_t* __restrict__ k_in, const uint8_t* __restrict__ h_in, uint32_t* __restrict__ k_out,
                             uint8_t* __restrict__ h_out, long n, long cap) {
  const long j = blockIdx.y;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    k_out[j * n + i] = k_in[j * cap + i];
    h_out[j * n + i] = h_in[j * cap + i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Tie-grouped average precision of descending-sorted segments.
//   AP = (1/npos) * sum over tie groups g of (TP_g - TP_{g-1}) * TP_g / k_g,   k_g = rank of the group's last element.
// A segment is cut into chunks (one workgroup each).  Pass A summarises a chunk (its positives; the positives up to
// its last group end), pass B carries (TP before the chunk, TP at the last group end before the chunk) sequentially
// over the few thousand chunk summaries, pass C evaluates the group ends of each chunk, pass D sums the per-chunk
// partials in order (deterministic, f64).
// ---------------------------------------------------------------------------------------------------------
constexpr int AP_T = 256, AP_I = 16, AP_TILE = AP_T * AP_I;

struct ApSeg {
  const uint32_t* keys;
  const uint8_t* hits;
  long long stride;  // between segments
  long long n;       // elements per segment
  long long chunk;   // elements per chunk (multiple of AP_TILE)
  int nchunks;
  long long tp0, k0;  // rank offsets of a key-range shard of a larger ranking (pn_ap_partial): positives / elements
                      // that rank before this segment's first element; 0 for a whole ranking
};

__device__ __forceinline__ long long block_sum(long long v, long long* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  long long t = 0;
  for (int i = 0; i < AP_T / 64; ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ long long block_max(long long v, long long* sh) {
  for (int o = 32; o > 0; o >>= 1) {
    long long u = __shfl_down(v, o);
    v = u > v ? u : v;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  long long t = sh[0];
  for (int i = 1; i < AP_T / 64; ++i) t = sh[i] > t ? sh[i] : t;
  return t;
}

__global__ __launch_bounds__(AP_T) void k_ap_summary(ApSeg sg, long long* __restrict__ c_tp, long long* __restrict__ c_tp_to_end,
                                                     int* __restrict__ c_has_end) {
  __shared__ long long sh[AP_T / 64];
  const int c = blockIdx.x;
  const long long seg = blockIdx.y;
  const uint32_t* k = sg.keys + seg * sg.stride;
  const uint8_t* h = sg.hits + seg * sg.stride;
  const long long lo = (long long)c * sg.chunk, hi = min(lo + sg.chunk, sg.n);
  long long last_end = -1, tp = 0;
  for (long long t0 = lo; t0 < hi; t0 += AP_TILE) {
    const long long b = t0 + (long long)threadIdx.x * AP_I;
#pragma unroll
    for (int e = 0; e < AP_I; ++e) {
      const long long i = b + e;
      if (i < hi) {
        tp += h[i];
        if (i == sg.n - 1 || k[i] != k[i + 1]) last_end = i;
      }
    }
  }
  const long long tot = block_sum(tp, sh);
  const long long le = block_max(last_end, sh);
  long long upto = 0;
  if (le >= 0) {  // second walk (L2-resident): positives at indices <= le
    for (long long t0 = lo; t0 < hi && t0 <= le; t0 += AP_TILE) {
      const long long b = t0 + (long long)threadIdx.x * AP_I;
#pragma unroll
      for (int e = 0; e < AP_I; ++e) {
        const long long i = b + e;
        if (i < hi && i <= le) upto += h[i];
      }
    }
  }
  upto = block_sum(upto, sh);
  if (threadIdx.x == 0) {
    const long long o = seg * sg.nchunks + c;
    c_tp[o] = tot;
    c_tp_to_end[o] = upto;
    c_has_end[o] = le >= 0;
  }
}

// one thread per segment: exclusive prefix of positives, and TP at the last group end before each chunk
__global__ void k_ap_carry(int nseg, int nchunks, const long long* __restrict__ c_tp, const long long* __restrict__ c_tp_to_end,
                           const int* __restrict__ c_has_end, long long* __restrict__ c_tp_before,
                           long long* __restrict__ c_prev_end_tp, long long* __restrict__ npos) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nseg) return;
  long long run = 0, prev_end = 0;
  for (int c = 0; c < nchunks; ++c) {
    const long long o = (long long)seg * nchunks + c;
    c_tp_before[o] = run;
    c_prev_end_tp[o] = prev_end;
    if (c_has_end[o]) prev_end = run + c_tp_to_end[o];
    run += c_tp[o];
  }
  npos[seg] = run;
}

__global__ __launch_bounds__(AP_T) void k_ap_apply(ApSeg sg, const long long* __restrict__ c_tp_before,
                                                   const long long* __restrict__ c_prev_end_tp, double* __restrict__ c_part) {
  __shared__ long long sh[AP_T / 64];
  __shared__ long long s_sum[AP_T], s_end[AP_T];
  __shared__ double s_acc[AP_T / 64];
  const int c = blockIdx.x;
  const long long seg = blockIdx.y;
  const uint32_t* k = sg.keys + seg * sg.stride;
  const uint8_t* h = sg.hits + seg * sg.stride;
  const long long lo = (long long)c * sg.chunk, hi = min(lo + sg.chunk, sg.n);
  long long tp_run = c_tp_before[seg * sg.nchunks + c];        // TP before the current tile
  long long end_run = c_prev_end_tp[seg * sg.nchunks + c];     // TP at the last group end before the current tile
  double acc = 0.0;
  for (long long t0 = lo; t0 < hi; t0 += AP_TILE) {
    const long long b = t0 + (long long)threadIdx.x * AP_I;
    uint32_t kk[AP_I + 1];
    uint8_t hh[AP_I];
    int cnt = 0, to_end = -1;  // positives in my items; positives up to my last group end (-1: none)
#pragma unroll
    for (int e = 0; e <= AP_I; ++e) kk[e] = (b + e < sg.n) ? k[b + e] : 0u;
#pragma unroll
    for (int e = 0; e < AP_I; ++e) {
      const long long i = b + e;
      hh[e] = (i < hi) ? h[i] : 0;
      cnt += hh[e];
      if (i < hi && (i == sg.n - 1 || kk[e] != kk[e + 1])) to_end = cnt;
    }
    // block exclusive scan of cnt (sum) and of "TP at my last group end" (max) - 256 entries, Hillis-Steele in LDS
    __syncthreads();
    s_sum[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 1; o < AP_T; o <<= 1) {
      long long v = threadIdx.x >= o ? s_sum[threadIdx.x - o] : 0;
      __syncthreads();
      s_sum[threadIdx.x] += v;
      __syncthreads();
    }
    const long long incl = s_sum[threadIdx.x];
    const long long my_tp0 = tp_run + incl - cnt;  // TP before my first item
    s_end[threadIdx.x] = to_end >= 0 ? my_tp0 + to_end : -1;
    __syncthreads();
    for (int o = 1; o < AP_T; o <<= 1) {
      long long v = threadIdx.x >= o ? s_end[threadIdx.x - o] : -1;
      __syncthreads();
      if (v > s_end[threadIdx.x]) s_end[threadIdx.x] = v;
      __syncthreads();
    }
    long long prev = threadIdx.x ? s_end[threadIdx.x - 1] : -1;
    if (prev < end_run) prev = end_run;  // TP is non-decreasing, so "latest" == "largest"
    long long tp = my_tp0;
#pragma unroll
    for (int e = 0; e < AP_I; ++e) {
      const long long i = b + e;
      tp += hh[e];
      if (i < hi && (i == sg.n - 1 || kk[e] != kk[e + 1])) {
        if (tp > prev) acc += (double)(tp - prev) * (double)(tp + sg.tp0) / (double)(i + 1 + sg.k0);
        prev = tp;
      }
    }
    const long long tile_tp = s_sum[AP_T - 1], tile_end = s_end[AP_T - 1];
    tp_run += tile_tp;
    if (tile_end > end_run) end_run = tile_end;
  }
  (void)sh;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int i = 0; i < AP_T / 64; ++i) t += s_acc[i];
    c_part[seg * sg.nchunks + c] = t;
  }
}

__global__ void k_ap_final(int nseg, int nchunks, const double* __restrict__ c_part, const long long* __restrict__ npos,
                           double* __restrict__ ap, int raw) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= nseg) return;
  double t = 0;
  for (int c = 0; c < nchunks; ++c) t += c_part[(long long)seg * nchunks + c];
  if (raw) ap[seg] = t;  // un-normalised partial of a sharded ranking
  else ap[seg] = npos[seg] > 0 ? t / (double)npos[seg] : __longlong_as_double(0x7ff8000000000000LL);
}

struct SegOff {  // offset of segment j: begin = j*stride, end = j*stride + n
  unsigned stride, add;
  __host__ __device__ unsigned operator()(unsigned j) const { return j * stride + add; }
};

long long chunk_for(long long n) {
  long long ch = (n + 8191) / 8192;  // at most 8192 chunks per segment
  ch = (ch + AP_TILE - 1) / AP_TILE * AP_TILE;
  return ch < AP_TILE ? AP_TILE : ch;
}

struct ApScratch {
  long long *c_tp, *c_tp_to_end, *c_tp_before, *c_prev_end;
  int* c_has_end;
  double* c_part;
  long long* npos;
};

size_t scratch_bytes(long long nseg, int nchunks) {
  const size_t e = (size_t)nseg * nchunks;
  return 4 * al256(e * 8) + al256(e * 4) + al256(e * 8) + al256((size_t)nseg * 8);
}

char* carve(char* p, ApScratch& s, long long nseg, int nchunks) {
  const size_t e = (size_t)nseg * nchunks;
  s.c_tp = (long long*)p; p += al256(e * 8);
  s.c_tp_to_end = (long long*)p; p += al256(e * 8);
  s.c_tp_before = (long long*)p; p += al256(e * 8);
  s.c_prev_end = (long long*)p; p += al256(e * 8);
  s.c_has_end = (int*)p; p += al256(e * 4);
  s.c_part = (double*)p; p += al256(e * 8);
  s.npos = (long long*)p; p += al256((size_t)nseg * 8);
  return p;
}

int ap_of_sorted(const uint32_t* keys, const uint8_t* hits, long long nseg, long long n, long long stride, char* scratch,
                 double* ap, long long* npos_out, hipStream_t st, long long tp0 = 0, long long k0 = 0, int raw = 0) {
  ApSeg sg{keys, hits, stride, n, chunk_for(n), 0, tp0, k0};
  sg.nchunks = (int)((n + sg.chunk - 1) / sg.chunk);
  ApScratch s;
  carve(scratch, s, nseg, sg.nchunks);
  for (long long s0 = 0; s0 < nseg; s0 += 65535) {  // gridDim.y limit
    const int ns = (int)((nseg - s0) < 65535 ? (nseg - s0) : 65535);
    ApSeg g = sg;
    g.keys += s0 * stride;
    g.hits += s0 * stride;
    const long long o = s0 * sg.nchunks;
    hipLaunchKernelGGL(k_ap_summary, dim3(sg.nchunks, ns), dim3(AP_T), 0, st, g, s.c_tp + o, s.c_tp_to_end + o, s.c_has_end + o);
  }
  hipLaunchKernelGGL(k_ap_carry, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, st, (int)nseg, sg.nchunks, s.c_tp, s.c_tp_to_end,
                     s.c_has_end, s.c_tp_before, s.c_prev_end, s.npos);
  for (long long s0 = 0; s0 < nseg; s0 += 65535) {
    const int ns = (int)((nseg - s0) < 65535 ? (nseg - s0) : 65535);
    ApSeg g = sg;
    g.keys += s0 * stride;
    g.hits += s0 * stride;
    const long long o = s0 * sg.nchunks;
    hipLaunchKernelGGL(k_ap_apply, dim3(sg.nchunks, ns), dim3(AP_T), 0, st, g, s.c_tp_before + o, s.c_prev_end + o, s.c_part + o);
  }
  hipLaunchKernelGGL(k_ap_final, dim3((unsigned)((nseg + 63) / 64)), dim3(64), 0, st, (int)nseg, sg.nchunks, s.c_part, s.npos, ap, raw);
  if (npos_out) HIP_OK(hipMemcpyAsync(npos_out, s.npos, (size_t)nseg * 8, hipMemcpyDeviceToDevice, st));
  HIP_OK(hipGetLastError());
  return 0;
}

// labels per rocPRIM segmented-sort call (its sizes and offsets are 32-bit)
long long labels_per_sort(long long cap) {
  long long m = 0x7fffffffLL / (cap > 0 ? cap : 1);
  return m < 1 ? 0 : m;
}

struct ApPlan {
  size_t sorted_keys, sorted_hits, compact_keys, compact_hits, tmp, scratch, total;
  size_t tmp_bytes;
};

int ap_plan(int NL, long long n, long long cap, int micro, ApPlan& p) {
  if (NL <= 0 || n <= 0 || cap < n) return fail_msg("average precision: need N_L > 0, 0 < n <= cap");
  if (labels_per_sort(cap) == 0) return fail_msg("average precision: cap %lld exceeds the 2^31 segment limit", cap);
  const size_t e = (size_t)NL * (size_t)cap;
  size_t off = 0;
  p.sorted_keys = off; off += al256(e * 4);
  p.sorted_hits = off; off += al256(e);
  const bool need_compact = micro && cap != n;
  p.compact_keys = off; off += need_compact ? al256((size_t)NL * n * 4) : 0;
  p.compact_hits = off; off += need_compact ? al256((size_t)NL * n) : 0;
  size_t t_seg = 0, t_flat = 0;
  const long long lps = labels_per_sort(cap);
  const unsigned nl0 = (unsigned)(NL < lps ? NL : lps);
  auto b = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), SegOff{(unsigned)cap, 0u});
  auto en = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), SegOff{(unsigned)cap, (unsigned)n});
  if (rocprim::segmented_radix_sort_pairs_desc(nullptr, t_seg, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                               (const uint8_t*)nullptr, (uint8_t*)nullptr, (unsigned)(nl0 * cap), nl0, b, en, 0, 32,
                                               (hipStream_t)0) != hipSuccess)
    return fail_msg("rocprim segmented sort: temp-storage query failed");
  if (micro && rocprim::radix_sort_pairs_desc(nullptr, t_flat, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                              (const uint8_t*)nullptr, (uint8_t*)nullptr, (size_t)NL * (size_t)n, 0, 32,
                                              (hipStream_t)0) != hipSuccess)
    return fail_msg("rocprim radix sort: temp-storage query failed");
  p.tmp_bytes = t_seg > t_flat ? t_seg : t_flat;
  p.tmp = off; off += al256(p.tmp_bytes ? p.tmp_bytes : 1);
  p.scratch = off;
  const long long ch_l = chunk_for(n), ch_m = chunk_for((long long)NL * n);
  size_t sc = scratch_bytes(NL, (int)((n + ch_l - 1) / ch_l));
  if (micro) {
    const size_t sm = scratch_bytes(1, (int)(((long long)NL * n + ch_m - 1) / ch_m));
    sc = sc > sm ? sc : sm;
  }
  off += sc;
  p.total = off + 256;
  return 0;
}

}  // namespace

extern "C" int pn_ap_append(const float* scores, int ld_s, const void* labels, int label_kind, int ld_y, int B, int N_L,
                            uint32_t* keys, uint8_t* hits, long long cap, long long n0, void* stream) {
  if (label_kind < 0 || label_kind > 2) return fail_msg("pn_ap_append: label_kind must be PN_LABEL_F32/I64/U8");
  if (n0 < 0 || n0 + B > cap) return fail_msg("pn_ap_append: rows [%lld, %lld) exceed capacity %lld", n0, n0 + B, cap);
  if (B <= 0 || N_L <= 0) return 0;
  hipLaunchKernelGGL(k_ap_append, dim3((N_L + 63) / 64, (B + 63) / 64), dim3(256), 0, (hipStream_t)stream, scores, ld_s, labels,
                     label_kind, ld_y, B, N_L, keys, hits, (long)cap, (long)n0);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" size_t pn_ap_ws_bytes(int N_L, long long n, long long cap, int micro) {
  ApPlan p;
  if (ap_plan(N_L, n, cap, micro, p)) return 0;
  return p.total;
}

extern "C" int pn_ap_compute(const uint32_t* keys, const uint8_t* hits, int N_L, long long n, long long cap, double* ap,
                             long long* npos, double* micro_ap, long long* micro_npos, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ApPlan p;
  const int micro = micro_ap != nullptr;
  if (ap_plan(N_L, n, cap, micro, p)) return 1;
  if (ws_bytes < p.total) return fail_msg("pn_ap_compute: workspace %zu < %zu bytes", ws_bytes, p.total);
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  uint32_t* sk = (uint32_t*)(base + p.sorted_keys);
  uint8_t* shh = (uint8_t*)(base + p.sorted_hits);
  void* tmp = base + p.tmp;
  // per-label: sort every label's n scores (descending) with its hit flags, in calls of < 2^31 elements
  const long long lps = labels_per_sort(cap);
  for (long long j0 = 0; j0 < N_L; j0 += lps) {
    const unsigned nl = (unsigned)((N_L - j0) < lps ? (N_L - j0) : lps);
    auto b = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), SegOff{(unsigned)cap, 0u});
    auto en = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), SegOff{(unsigned)cap, (unsigned)n});
    size_t tb = p.tmp_bytes;
    HIP_OK(rocprim::segmented_radix_sort_pairs_desc(tmp, tb, keys + j0 * cap, sk + j0 * cap, hits + j0 * cap, shh + j0 * cap,
                                                    (unsigned)(nl * cap), nl, b, en, 0, 32, st));
  }
  if (ap_of_sorted(sk, shh, N_L, n, cap, base + p.scratch, ap, npos, st)) return 1;
  if (micro) {
    const uint32_t* kin = keys;
    const uint8_t* hin = hits;
    if (cap != n) {
      uint32_t* ck = (uint32_t*)(base + p.compact_keys);
      uint8_t* chh = (uint8_t*)(base + p.compact_hits);
      for (int j0 = 0; j0 < N_L; j0 += 65535) {  // gridDim.y limit
        const int nl = N_L - j0 < 65535 ? N_L - j0 : 65535;
        hipLaunchKernelGGL(k_ap_compact, dim3((unsigned)((n + 1023) / 1024 < 64 ? (n + 1023) / 1024 : 64), nl), dim3(256), 0, st,
                           keys + (size_t)j0 * cap, hits + (size_t)j0 * cap, ck + (size_t)j0 * n, chh + (size_t)j0 * n, (long)n,
                           (long)cap);
      }
      kin = ck;
      hin = chh;
    }
    size_t tb = p.tmp_bytes;
    const size_t total = (size_t)N_L * (size_t)n;
    HIP_OK(rocprim::radix_sort_pairs_desc(tmp, tb, kin, sk, hin, shh, total, 0, 32, st));
    if (ap_of_sorted(sk, shh, 1, (long long)total, 0, base + p.scratch, micro_ap, micro_npos, st)) return 1;
  }
  return 0;
}

// Sharded ranking (multi-GPU micro AP): this rank holds ALL pairs of the whole evaluation whose key lies in one key
// range (a sample-sort bucket); tp_before / k_before = positives / pairs of the higher-ranked buckets.  Writes the
// un-normalised partial sum over this bucket's tie groups of (TP_g - TP_{g-1}) * TP_g / k_g with the GLOBAL TP_g and
// k_g, and the bucket's positives; micro AP = (sum of partials over ranks) / (sum of positives).
extern "C" size_t pn_ap_partial_ws_bytes(long long m) {
  if (m <= 0) return 256;
  size_t t_flat = 0;
  if (rocprim::radix_sort_pairs_desc(nullptr, t_flat, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint8_t*)nullptr,
                                     (uint8_t*)nullptr, (size_t)m, 0, 32, (hipStream_t)0) != hipSuccess)
    return 0;
  const long long ch = chunk_for(m);
  return al256((size_t)m * 4) + al256((size_t)m) + al256(t_flat ? t_flat : 1) + scratch_bytes(1, (int)((m + ch - 1) / ch)) + 512;
}

extern "C" int pn_ap_partial(const uint32_t* keys, const uint8_t* hits, long long m, long long tp_before, long long k_before,
                             double* partial, long long* npos, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (m < 0 || tp_before < 0 || k_before < 0) return fail_msg("pn_ap_partial: negative size / offset");
  if (m == 0) {
    HIP_OK(hipMemsetAsync(partial, 0, sizeof(double), st));
    HIP_OK(hipMemsetAsync(npos, 0, sizeof(long long), st));
    return 0;
  }
  const size_t need = pn_ap_partial_ws_bytes(m);
  if (need == 0) return fail_msg("rocprim radix sort: temp-storage query failed");
  if (ws_bytes < need) return fail_msg("pn_ap_partial: workspace %zu < %zu bytes", ws_bytes, need);
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  uint32_t* sk = (uint32_t*)base;
  uint8_t* shh = (uint8_t*)(base + al256((size_t)m * 4));
  size_t t_flat = 0;
  rocprim::radix_sort_pairs_desc(nullptr, t_flat, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint8_t*)nullptr,
                                 (uint8_t*)nullptr, (size_t)m, 0, 32, (hipStream_t)0);
  char* tmp = base + al256((size_t)m * 4) + al256((size_t)m);
  size_t tb = t_flat;
  HIP_OK(rocprim::radix_sort_pairs_desc(tmp, tb, keys, sk, hits, shh, (size_t)m, 0, 32, st));
  return ap_of_sorted(sk, shh, 1, m, 0, tmp + al256(t_flat ? t_flat : 1), partial, npos, st, tp_before, k_before, 1);
}

// ---------------------------------------------------------------------------------------------------------
// Binned AUPRC (ESTIMATE_MAP: True, ProtNoteTrainer.py:481-485: threshold=50 -> linspace(0,1,50)).
// Streaming state = per-label histograms over the T+1 intervals the thresholds cut [..]: bin(p) = #{k : p >= thr_k}.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int BIN_LABELS = 64;

__global__ __launch_bounds__(256) void k_binned_hist(const float* __restrict__ scores, int ld_s, const void* __restrict__ labels,
                                                     int kind, int ld_y, int B, int NL, const float* __restrict__ thr, int T,
                                                     unsigned long long* __restrict__ pos_hist,
                                                     unsigned long long* __restrict__ all_hist) {
  extern __shared__ unsigned sm[];
  float* sthr = (float*)sm;                 // [T]
  unsigned* hp = sm + T;                    // [BIN_LABELS][T+1]
  unsigned* ha = hp + BIN_LABELS * (T + 1);
  for (int i = threadIdx.x; i < T; i += 256) sthr[i] = thr[i];
  for (int i = threadIdx.x; i < 2 * BIN_LABELS * (T + 1); i += 256) hp[i] = 0;
  __syncthreads();
  const int j0 = blockIdx.x * BIN_LABELS, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = j0 + tx;
  if (j < NL) {
    for (int i = blockIdx.y * 4 + ty; i < B; i += gridDim.y * 4) {
      const float p = scores[(long)i * ld_s + j];
      int lo = 0, hi = T;  // first k with thr[k] > p  ==  #{k: thr[k] <= p}
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sthr[mid] <= p) lo = mid + 1; else hi = mid;
      }
      atomicAdd(&ha[tx * (T + 1) + lo], 1u);
      if (label_hit(labels, kind, (long)i * ld_y + j)) atomicAdd(&hp[tx * (T + 1) + lo], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < BIN_LABELS * (T + 1); i += 256) {
    const int l = i / (T + 1), b = i % (T + 1);
    if (j0 + l < NL) {
      if (ha[i]) atomicAdd(&all_hist[(long)(j0 + l) * (T + 1) + b], (unsigned long long)ha[i]);
      if (hp[i]) atomicAdd(&pos_hist[(long)(j0 + l) * (T + 1) + b], (unsigned long long)hp[i]);
    }
  }
}

// row NL of both histograms := sum of the label rows (one workgroup per bin; fixed order, integer -> deterministic)
__global__ __launch_bounds__(256) void k_hist_pool(unsigned long long* __restrict__ pos_hist, unsigned long long* __restrict__ all_hist,
                                                   int NL, int T) {
  __shared__ unsigned long long sp[256], sa[256];
  const int b = blockIdx.x;
  unsigned long long p = 0, a = 0;
  for (int l = threadIdx.x; l < NL; l += 256) {
    p += pos_hist[(long)l * (T + 1) + b];
    a += all_hist[(long)l * (T + 1) + b];
  }
  sp[threadIdx.x] = p;
  sa[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sp[threadIdx.x] += sp[threadIdx.x + o];
      sa[threadIdx.x] += sa[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pos_hist[(long)NL * (T + 1) + b] = sp[0];
    all_hist[(long)NL * (T + 1) + b] = sa[0];
  }
}

// AUPRC from histograms [(NL+1)][T+1]; row NL is the pooled (micro) histogram (filled by k_hist_pool).
// tp_k = #{pos, p >= thr_k}, precision_k = tp_k / cnt_k (1 when cnt_k = 0), recall_k = tp_k / npos, a final
// (precision 1, recall 0) point is appended, area = sum_k (recall_k - recall_{k+1}) * precision_k.
__global__ void k_binned_auprc(const unsigned long long* __restrict__ pos_hist, const unsigned long long* __restrict__ all_hist,
                               int NL, int T, double* __restrict__ out, long long* __restrict__ npos_out, int with_micro) {
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= NL + (with_micro ? 1 : 0)) return;
  auto at = [&](const unsigned long long* h, int b) -> unsigned long long { return h[(long)seg * (T + 1) + b]; };
  unsigned long long npos = 0;
  for (int b = 0; b <= T; ++b) npos += at(pos_hist, b);
  // walk thresholds from the highest down, accumulating suffix sums; area needs recall_{k+1} (the next higher one)
  unsigned long long tp = 0, cnt = 0;
  double area = 0, r_next = 0;
  for (int k = T - 1; k >= 0; --k) {
    tp += at(pos_hist, k + 1);
    cnt += at(all_hist, k + 1);
    const double prec = cnt ? (double)tp / (double)cnt : 1.0;
    const double rec = npos ? (double)tp / (double)npos : __longlong_as_double(0x7ff8000000000000LL);
    area += (rec - r_next) * prec;
    r_next = rec;
  }
  out[seg] = area;
  if (npos_out) npos_out[seg] = (long long)npos;
}
}  // namespace

extern "C" int pn_binned_hist_update(const float* scores, int ld_s, const void* labels, int label_kind, int ld_y, int B, int N_L,
                                     const float* thresholds, int T, unsigned long long* pos_hist, unsigned long long* all_hist,
                                     void* stream) {
  if (label_kind < 0 || label_kind > 2) return fail_msg("pn_binned_hist_update: label_kind must be PN_LABEL_F32/I64/U8");
  if (T < 1 || T > 120) return fail_msg("pn_binned_hist_update: 1 <= T <= 120 thresholds (LDS histogram)");
  if (B <= 0 || N_L <= 0) return 0;
  const size_t lds = (size_t)T * 4 + 2 * (size_t)BIN_LABELS * (T + 1) * 4;
  int gy = B / 256;  // >= 256 rows per workgroup amortise zeroing and flushing its 64 x (T+1) histogram
  gy = gy < 1 ? 1 : (gy > 64 ? 64 : gy);
  hipLaunchKernelGGL(k_binned_hist, dim3((N_L + BIN_LABELS - 1) / BIN_LABELS, gy), dim3(256), lds, (hipStream_t)stream, scores, ld_s,
                     labels, label_kind, ld_y, B, N_L, thresholds, T, pos_hist, all_hist);
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int pn_binned_auprc(unsigned long long* pos_hist, unsigned long long* all_hist, int N_L, int T, double* out,
                               long long* npos, int with_micro, void* stream) {
  if (N_L <= 0 || T < 1) return fail_msg("pn_binned_auprc: need N_L > 0, T >= 1");
  const int n = N_L + (with_micro ? 1 : 0);
  if (with_micro) hipLaunchKernelGGL(k_hist_pool, dim3(T + 1), dim3(256), 0, (hipStream_t)stream, pos_hist, all_hist, N_L, T);
  hipLaunchKernelGGL(k_binned_auprc, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pos_hist, all_hist, N_L, T, out, npos,
                     with_micro);
  HIP_OK(hipGetLastError());
  return 0;
}
