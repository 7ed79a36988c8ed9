// Lab (not product): variants of the all-LDS-DMA one-product bf16 NT GEMM at the AMP-class forward / dh shapes, timed with HIP
// events on random bf16 operands and checked against the 32 x 32 x 16 kernel (gemm_nt_bf16dma_kernel, csrc/bwd_bf16_dz.hpp): bit
// for bit where the reduction shape is the same, to f32 rounding otherwise.  Also the kernel-level check of the adopted
// 16 x 16 x 32 kernel (csrc/gemm_bf16_m16.hpp, rows "PRODUCT").  The vendor library's best kernel for this shape (tools/bf16_gemm_yardstick.py: 1.46 PFLOP/s, a 256 x 256 x 64 macro
// tile on FOUR waves) is the yardstick the variants are read against.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -Iprotnote_amd/csrc -Itools tools/lab_bf16_nt.hip -o tools/lab_bf16_nt.bin
//   tools/lab_bf16_nt.bin [M ...]        (default 524288 262144)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "protnote_hip.h"
#include "bwd_bf16_dz.hpp"
#include "gemm_bf16_m16.hpp"
#include "lab_w4.hpp"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// normal(0, sd) rounded to bf16 (round to nearest even), from a counter hash
__global__ void k_fill_bf16(uint16_t* out, long n, uint32_t seed, float sd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = mix((uint32_t)i * 2u + seed), b = mix((uint32_t)i * 2u + 1u + seed * 31u);
  const float u1 = ((a >> 8) + 1) * (1.0f / 16777217.0f), u2 = (b >> 8) * (1.0f / 16777216.0f);
  const float v = sd * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  uint32_t bits = __float_as_uint(v);
  bits += 0x7fffu + ((bits >> 16) & 1u);
  out[i] = (uint16_t)(bits >> 16);
}
__global__ void k_fill_f32(float* out, long n, uint32_t seed, float lo, float hi) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = lo + (hi - lo) * ((mix((uint32_t)i + seed) >> 8) * (1.0f / 16777216.0f));
}
// order-independent 64-bit digest of a buffer of 32-bit words (sum of hashed (index, word))
__global__ void k_digest(const uint32_t* w, long n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc += (unsigned long long)mix(w[i] ^ mix((uint32_t)i)) * 0x9E3779B97F4A7C15ull + (unsigned long long)w[i];
  atomicAdd(out, acc);
}

// max |a - b| and max |a| of two f32 arrays (BF = 1: arrays of bf16), as the bit patterns of non-negative floats
template <int BF>
__global__ void k_maxdiff(const void* a_, const void* b_, long n, unsigned* out) {
  float md = 0.f, ma = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a, b;
    if (BF) {
      a = __uint_as_float((uint32_t) reinterpret_cast<const uint16_t*>(a_)[i] << 16);
      b = __uint_as_float((uint32_t) reinterpret_cast<const uint16_t*>(b_)[i] << 16);
    } else {
      a = reinterpret_cast<const float*>(a_)[i];
      b = reinterpret_cast<const float*>(b_)[i];
    }
    const float d = fabsf(a - b);
    md = d > md || d != d ? (d != d ? 3e38f : d) : md;
    ma = fmaxf(ma, fabsf(a));
  }
  atomicMax(out, __float_as_uint(md));
  atomicMax(out + 1, __float_as_uint(ma));
}

struct Bufs {
  uint16_t *A, *W, *C16;
  float *es, *et, *ew, *rowdot;
  unsigned long long* dg;
};

static void maxdiff(const char* what, const void* a, const void* b, long n, bool bf, unsigned long long* scratch, int* bad, double tol) {
  unsigned* o = reinterpret_cast<unsigned*>(scratch);
  CK(hipMemset(o, 0, 8));
  if (bf) hipLaunchKernelGGL(k_maxdiff<1>, dim3(4096), dim3(256), 0, 0, a, b, n, o);
  else hipLaunchKernelGGL(k_maxdiff<0>, dim3(4096), dim3(256), 0, 0, a, b, n, o);
  unsigned h[2];
  CK(hipMemcpy(h, o, 8, hipMemcpyDeviceToHost));
  float md, ma;
  memcpy(&md, &h[0], 4);
  memcpy(&ma, &h[1], 4);
  const bool ok = md <= tol * ma;
  printf("    %-40s max |diff| %.3e at max |ref| %.3e  (%.2e relative) %s\n", what, md, ma, md / (ma > 0 ? ma : 1), ok ? "ok" : "TOO LARGE");
  *bad += !ok;
}

static unsigned long long digest(const void* p, long words, unsigned long long* dg) {
  CK(hipMemset(dg, 0, 8));
  hipLaunchKernelGGL(k_digest, dim3(4096), dim3(256), 0, 0, (const uint32_t*)p, words, dg);
  unsigned long long h = 0;
  CK(hipMemcpy(&h, dg, 8, hipMemcpyDeviceToHost));
  return h;
}

template <typename K>
static double run(const char* name, K kern, int threads, int lds, pn::GemmParams p, int reps, double* out_ms) {
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const long tm = (p.M + 255) / 256, tn = p.N / 256;
  p.xcd_bc = (tm >= 16) ? ((tn % 8 == 0) ? 8 : ((tn % 4 == 0) ? 4 : 0)) : 0;
  p.xcd_br = p.xcd_bc ? 32 / p.xcd_bc : 0;
  long grid = tm * tn;
  if (p.xcd_bc) {
    const long nblk = ((tm + p.xcd_br - 1) / p.xcd_br) * (tn / p.xcd_bc);
    grid = ((nblk + 7) / 8) * 8 * 32;
  }
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, 0, p);
  CK(hipDeviceSynchronize());
  std::vector<hipEvent_t> ev(reps + 1);
  for (auto& e : ev) CK(hipEventCreate(&e));
  CK(hipEventRecord(ev[0], 0));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, 0, p);
    CK(hipEventRecord(ev[i + 1], 0));
  }
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> ms(reps);
  for (int i = 0; i < reps; ++i) CK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  for (auto& e : ev) CK(hipEventDestroy(e));
  std::sort(ms.begin(), ms.end());
  const double med = ms[reps / 2];
  const double tf = 2.0 * p.M * (double)p.N * p.Kseg / (med * 1e-3) / 1e12;
  printf("  %-44s median %8.3f ms  (min %.3f max %.3f)  %8.1f TFLOP/s\n", name, med, ms[0], ms[reps - 1], tf);
  *out_ms = med;
  return tf;
}

int main(int argc, char** argv) {
  std::vector<long> Ms;
  for (int i = 1; i < argc; ++i) Ms.push_back(atol(argv[i]));
  if (Ms.empty()) Ms = {524288, 262144};
  const int h = 3072;
  const int reps = 20;
  long Mmax = 0;
  for (long m : Ms) Mmax = std::max(Mmax, m);
  Bufs b;
  CK(hipMalloc(&b.A, (size_t)Mmax * h * 2));
  CK(hipMalloc(&b.W, (size_t)h * h * 2));
  CK(hipMalloc(&b.C16, (size_t)Mmax * h * 2));
  CK(hipMalloc(&b.es, h * 4));
  CK(hipMalloc(&b.et, h * 4));
  CK(hipMalloc(&b.ew, h * 4));
  CK(hipMalloc(&b.rowdot, (size_t)(h / 256) * 2 * Mmax * 4));
  CK(hipMalloc(&b.dg, 8));
  const long nA = Mmax * h;
  hipLaunchKernelGGL(k_fill_bf16, dim3((unsigned)((nA + 255) / 256)), dim3(256), 0, 0, b.A, nA, 11u, 1.0f);
  hipLaunchKernelGGL(k_fill_bf16, dim3((unsigned)(((long)h * h + 255) / 256)), dim3(256), 0, 0, b.W, (long)h * h, 23u, 0.02f);
  hipLaunchKernelGGL(k_fill_f32, dim3((h + 255) / 256), dim3(256), 0, 0, b.es, (long)h, 5u, 0.5f, 1.5f);
  hipLaunchKernelGGL(k_fill_f32, dim3((h + 255) / 256), dim3(256), 0, 0, b.et, (long)h, 6u, -0.5f, 0.5f);
  hipLaunchKernelGGL(k_fill_f32, dim3((h + 255) / 256), dim3(256), 0, 0, b.ew, (long)h, 7u, -1.f, 1.f);
  CK(hipDeviceSynchronize());

  int bad = 0;
  float *C32, *C32ref, *cpart, *cpart_ref, *rd_ref;
  uint16_t* C16ref;
  CK(hipMalloc(&C32, (size_t)Mmax * h * 4));
  CK(hipMalloc(&C32ref, (size_t)Mmax * h * 4));
  CK(hipMalloc(&C16ref, (size_t)Mmax * h * 2));
  CK(hipMalloc(&cpart, (size_t)((Mmax + 255) / 256) * 2 * h * 4));
  CK(hipMalloc(&cpart_ref, (size_t)((Mmax + 255) / 256) * 2 * h * 4));
  CK(hipMalloc(&rd_ref, (size_t)(h / 256) * 2 * Mmax * 4));
  for (long M : Ms) {
    printf("M = %ld, N = K = %d (%.2f TFLOP per launch)\n", M, h, 2.0 * M * h * (double)h / 1e12);
    pn::GemmParams p = {};
    p.M = (int)M;
    p.N = p.Nstore = h;
    p.nseg = 1;
    p.Kseg = h;
    p.A = reinterpret_cast<const float*>(b.A);
    p.lda = h / 2;  // row stride in floats of a dense bf16 [M][K]
    p.w_hi = b.W;
    p.e_scale = b.es;
    p.e_shift = b.et;
    p.e_w = b.ew;
    p.rowdot_out = b.rowdot;
    p.C = reinterpret_cast<float*>(b.C16);
    p.ldc = h;
    double ms;
    const long rd_words = (long)(h / 256) * 2 * M, c_words = M * h / 2, cp_words = ((M + 255) / 256) * 2 * h;
    // ---- E_ROWDOT
    CK(hipMemset(b.rowdot, 0xff, (size_t)rd_words * 4));
    run("shipped  <E_ROWDOT>   8 waves, 32x32x16", pn::gemm_nt_bf16dma_kernel<pn::E_ROWDOT>, 512, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);
    const unsigned long long d0 = digest(b.rowdot, rd_words, b.dg);
    CK(hipMemcpy(rd_ref, b.rowdot, (size_t)rd_words * 4, hipMemcpyDeviceToDevice));
#define VARIANT(label, ...)                                                                  \
  do {                                                                                       \
    CK(hipMemset(b.rowdot, 0xff, (size_t)rd_words * 4));                                     \
    run(label, __VA_ARGS__, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);                         \
    const bool same = d0 == digest(b.rowdot, rd_words, b.dg);                                \
    if (!same) printf("  ^^^ DIFFERENT from the shipped kernel\n");                          \
    bad += !same;                                                                            \
  } while (0)
#define VARIANT16(label, ...)                                                                \
  do {                                                                                       \
    CK(hipMemset(b.rowdot, 0xff, (size_t)rd_words * 4));                                     \
    run(label, __VA_ARGS__, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);                         \
    maxdiff("row dots vs shipped", rd_ref, b.rowdot, rd_words, false, b.dg, &bad, 2e-5);     \
  } while (0)
    VARIANT("w4 6/6/4/0 fine: pieces even, reads odd", pn::lab::gemm_nt_bf16dma_w4_kernel<pn::E_ROWDOT, 6, 6, 4, 0, 2, true>, 256);
    VARIANT("w8 m0 + early + fenced", pn::lab::gemm_nt_bf16dma_v_kernel<pn::E_ROWDOT, 7>, 512);
    VARIANT16("PRODUCT gemm_nt_bf16m16_kernel<E_ROWDOT>", pn::gemm_nt_bf16m16_kernel<pn::E_ROWDOT>, 512);
    VARIANT16("m16 <E_ROWDOT> 16x16x32, DMA ahead of phase A", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 0>, 512);
    VARIANT16("m16 <E_ROWDOT> 16x16x32, W pieces behind phase A", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 1>, 512);
    VARIANT16("m16s <E_ROWDOT> swapped roles", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 3>, 512);
    VARIANT16("m16 <E_ROWDOT> A01|16|A23|16|W01|rd|16|W23|16", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 8>, 512);
    VARIANT16("m16 <E_ROWDOT> A01|16|A23|16|W0123|rd|32", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 16>, 512);
    // ---- E_STORE_H16
    CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
    run("shipped  <E_STORE_H16>", pn::gemm_nt_bf16dma_kernel<pn::E_STORE_H16>, 512, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);
    CK(hipMemcpy(C16ref, b.C16, (size_t)c_words * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
    run("PRODUCT gemm_nt_bf16m16_kernel<E_STORE_H16>", pn::gemm_nt_bf16m16_kernel<pn::E_STORE_H16>, 512, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);
    maxdiff("bf16 h vs shipped: must be 0", C16ref, b.C16, M * h, true, b.dg, &bad, 0.0);
    CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
    run("m16      <E_STORE_H16>", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE_H16, 1>, 512, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);
    maxdiff("bf16 h vs shipped (one bf16 ulp = 7.8e-3 rel.)", C16ref, b.C16, M * h, true, b.dg, &bad, 8e-3);
    CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
    run("m16s     <E_STORE_H16> swapped roles", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE_H16, 3>, 512, pn::NT_BF16DMA_LDS_BYTES, p, reps, &ms);
    maxdiff("bf16 h vs shipped", C16ref, b.C16, M * h, true, b.dg, &bad, 8e-3);
    // ---- E_STORE: f32 store + BatchNorm column partials
    pn::GemmParams ps = p;
    ps.e_scale = ps.e_shift = nullptr;
    ps.C = C32;
    ps.col_part = cpart;
    CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
    run("shipped  <E_STORE> f32 + column partials", pn::gemm_nt_bf16dma_kernel<pn::E_STORE>, 512, pn::NT_BF16DMA_LDS_BYTES, ps, reps, &ms);
    CK(hipMemcpy(C32ref, C32, (size_t)M * h * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(cpart_ref, cpart, (size_t)cp_words * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
    CK(hipMemset(cpart, 0xff, (size_t)cp_words * 4));
    run("PRODUCT gemm_nt_bf16m16_kernel<E_STORE> + partials", pn::gemm_nt_bf16m16_kernel<pn::E_STORE>, 512, pn::NT_BF16DMA_LDS_BYTES, ps, reps, &ms);
    maxdiff("z vs shipped: must be 0", C32ref, C32, M * h, false, b.dg, &bad, 0.0);
    maxdiff("column partials vs shipped", cpart_ref, cpart, cp_words, false, b.dg, &bad, 2e-5);
    CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
    CK(hipMemset(cpart, 0xff, (size_t)cp_words * 4));
    run("m16      <E_STORE> f32 + column partials", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 1>, 512, pn::NT_BF16DMA_LDS_BYTES, ps, reps, &ms);
    maxdiff("z vs shipped", C32ref, C32, M * h, false, b.dg, &bad, 2e-5);
    maxdiff("column partials vs shipped", cpart_ref, cpart, cp_words, false, b.dg, &bad, 2e-5);
    CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
    CK(hipMemset(cpart, 0xff, (size_t)cp_words * 4));
    run("m16s     <E_STORE> f32 + column partials, swapped", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 3>, 512, pn::NT_BF16DMA_LDS_BYTES, ps, reps, &ms);
    maxdiff("z vs shipped", C32ref, C32, M * h, false, b.dg, &bad, 2e-5);
    maxdiff("column partials vs shipped", cpart_ref, cpart, cp_words, false, b.dg, &bad, 2e-5);
    {
      pn::GemmParams pq = ps;
      pq.col_part = nullptr;
      run("shipped  <E_STORE> f32 plain (the dh GEMM)", pn::gemm_nt_bf16dma_kernel<pn::E_STORE>, 512, pn::NT_BF16DMA_LDS_BYTES, pq, reps, &ms);
      run("m16      <E_STORE> f32 plain", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 1>, 512, pn::NT_BF16DMA_LDS_BYTES, pq, reps, &ms);
      CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
      run("m16s     <E_STORE> f32 plain, swapped", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 3>, 512, pn::NT_BF16DMA_LDS_BYTES, pq, reps, &ms);
      maxdiff("z vs shipped", C32ref, C32, M * h, false, b.dg, &bad, 2e-5);
      CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
      run("m16      <E_STORE> f32 plain, non-temporal stores", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 5>, 512, pn::NT_BF16DMA_LDS_BYTES, pq, reps, &ms);
      maxdiff("z vs shipped", C32ref, C32, M * h, false, b.dg, &bad, 2e-5);
      CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
      run("m16s     <E_STORE> f32 plain, swapped, non-temporal", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 7>, 512, pn::NT_BF16DMA_LDS_BYTES, pq, reps, &ms);
      maxdiff("z vs shipped", C32ref, C32, M * h, false, b.dg, &bad, 2e-5);
      for (int P : {2, 4, 8}) {
        for (int tile_cyc : {120000, 170000}) {
          pn::GemmParams pg = pq;
          pg.L = P;
          pg.dil = tile_cyc / P;
          char lab[96];
          snprintf(lab, sizeof lab, "m16 <E_STORE> plain, staggered %d x %d cyc", P, tile_cyc / P);
          CK(hipMemset(C32, 0xff, (size_t)M * h * 4));
          run(lab, pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE, 33>, 512, pn::NT_BF16DMA_LDS_BYTES, pg, reps, &ms);
          maxdiff("z vs shipped", C32ref, C32, M * h, false, b.dg, &bad, 0.0);
        }
      }
      {
        pn::GemmParams pg = p;  // row dots: no stores - what does the stagger itself cost?
        pg.L = 4;
        pg.dil = 170000 / 4;
        run("m16 <E_ROWDOT>, staggered 4 x 42500 cyc", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 33>, 512, pn::NT_BF16DMA_LDS_BYTES, pg, reps, &ms);
      }
      pn::GemmParams ph = p;
      CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
      run("m16s     <E_STORE_H16> swapped, non-temporal", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE_H16, 7>, 512, pn::NT_BF16DMA_LDS_BYTES, ph, reps, &ms);
      maxdiff("bf16 h vs shipped", C16ref, b.C16, M * h, true, b.dg, &bad, 8e-3);
      CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
      run("m16s     <E_STORE_H16> swapped, fragment pairs (permlane16_swap)", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE_H16, 67>, 512, pn::NT_BF16DMA_LDS_BYTES, ph, reps, &ms);
      maxdiff("bf16 h vs shipped: must be 0", C16ref, b.C16, M * h, true, b.dg, &bad, 0.0);
      CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
      run("m16s     <E_STORE_H16> swapped, fragment quads (permlane16 + 32 swaps)", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE_H16, 131>, 512, pn::NT_BF16DMA_LDS_BYTES, ph, reps, &ms);
      maxdiff("bf16 h vs shipped: must be 0", C16ref, b.C16, M * h, true, b.dg, &bad, 0.0);
      CK(hipMemset(b.C16, 0xff, (size_t)c_words * 4));
      run("m16      <E_STORE_H16> non-temporal", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_STORE_H16, 5>, 512, pn::NT_BF16DMA_LDS_BYTES, ph, reps, &ms);
      maxdiff("bf16 h vs shipped", C16ref, b.C16, M * h, true, b.dg, &bad, 8e-3);
    }
  }
  // a ragged M (row clamp) at small size
  {
    const long M = 256 * 37 + 77;
    pn::GemmParams p = {};
    p.M = (int)M;
    p.N = p.Nstore = h;
    p.nseg = 1;
    p.Kseg = h;
    p.A = reinterpret_cast<const float*>(b.A);
    p.lda = h / 2;
    p.w_hi = b.W;
    p.e_scale = b.es;
    p.e_shift = b.et;
    p.e_w = b.ew;
    p.rowdot_out = b.rowdot;
    p.C = reinterpret_cast<float*>(b.C16);
    p.ldc = h;
    double ms;
    const long rd_words = (long)(h / 256) * 2 * M;
    printf("M = %ld (ragged)\n", M);
    CK(hipMemset(b.rowdot, 0xff, (size_t)rd_words * 4));
    run("shipped  <E_ROWDOT>", pn::gemm_nt_bf16dma_kernel<pn::E_ROWDOT>, 512, pn::NT_BF16DMA_LDS_BYTES, p, 3, &ms);
    const unsigned long long d0 = digest(b.rowdot, rd_words, b.dg);
    CK(hipMemcpy(rd_ref, b.rowdot, (size_t)rd_words * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(b.rowdot, 0xff, (size_t)rd_words * 4));
    run("w4       <E_ROWDOT>", pn::lab::gemm_nt_bf16dma_w4_kernel<pn::E_ROWDOT>, 256, pn::NT_BF16DMA_LDS_BYTES, p, 3, &ms);
    const unsigned long long d1 = digest(b.rowdot, rd_words, b.dg);
    printf("  rowdot digests %016llx %016llx %s\n", d0, d1, d0 == d1 ? "identical" : "DIFFERENT");
    bad += d0 != d1;
    CK(hipMemset(b.rowdot, 0xff, (size_t)rd_words * 4));
    run("m16      <E_ROWDOT>", pn::lab::gemm_nt_bf16dma_m16_kernel<pn::E_ROWDOT, 0>, 512, pn::NT_BF16DMA_LDS_BYTES, p, 3, &ms);
    maxdiff("row dots vs shipped", rd_ref, b.rowdot, rd_words, false, b.dg, &bad, 2e-5);
  }
  printf(bad ? "LAB: %d mismatches\n" : "LAB: all variants agree with the shipped kernel\n", bad);
  return bad ? 2 : 0;
}
