// What the MI355X package can power: a register-only MFMA loop (no LDS, no global memory in the loop) on all 256 CUs,
// two waves per SIMD like the GEMM kernels, operands = random finite bit patterns held in registers.  Prints the sustained
// matrix rate; run it under tools/power_trace.py to see socket power, shader clock and throttler residency next to it:
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_power_probe.hip -o /tmp/mfma_probe
//   python tools/power_trace.py --out gpurun_out/r03_power_probe_bf16.json -- /tmp/mfma_probe bf16 8
//   python tools/power_trace.py --out gpurun_out/r03_power_probe_f32.json  -- /tmp/mfma_probe f32 8
//   /tmp/mfma_probe bf16_16 8     (round 6: the same loop on v_mfma_f32_16x16x32_bf16)
//
// This is an UPPER bound for any real kernel (operands never change, so the datapath toggles less than with streamed data,
// and nothing else on the chip draws power): DESIGN.md 4.6 uses it to place the bf16x3 GEMMs (3 bf16 MFMAs per product)
// against what 1400 W can feed rather than against the dense-issue peak.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

// 2 x 4 accumulator tiles per wave (the 64 x 128 wave tile of the GEMM kernels): 8 independent MFMA chains
template <bool BF16>
__global__ __launch_bounds__(512, 2) void k_probe(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  uint32_t r[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) r[k] = seed[(threadIdx.x * 24 + k) & 4095];
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (BF16) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      bf16x8 a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = __builtin_bit_cast(bf16x8, (u32x4){r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]});
#pragma unroll
      for (int j = 0; j < 4; ++j)
        b[j] = __builtin_bit_cast(bf16x8, (u32x4){r[8 + 4 * j], r[9 + 4 * j], r[10 + 4 * j], r[11 + 4 * j]});
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r[i + 2 * rep]), __uint_as_float(r[8 + j + 4 * rep]),
                                                             acc[i][j], 0, 0, 0);
    }
    // keep the operands "live" for the compiler without changing them (no VALU work in the loop)
#pragma unroll
    for (int k = 0; k < 24; ++k) asm volatile("" : "+v"(r[k]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 123.456f) out[blockIdx.x * 64 + lane] = s;  // never true: keeps the accumulators alive
}

// the same register-only loop on v_mfma_f32_16x16x32_bf16: 4 x 8 accumulator tiles of 16 x 16 (the same 64 x 128 wave tile and the
// same 128 accumulator registers), 32 independent chains; one pass = 32 MFMAs over k = 32 = the flops of 16 MFMAs of 32 x 32 x 16
__global__ __launch_bounds__(512, 2) void k_probe16(const uint32_t* __restrict__ seed, float* __restrict__ out, int iters) {
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  uint32_t r[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) r[k] = seed[(threadIdx.x * 48 + k) & 4095];
  f32x4_ acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
    bf16x8 a[4], b[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(bf16x8, (u32x4){r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]});
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = __builtin_bit_cast(bf16x8, (u32x4){r[16 + 4 * j], r[17 + 4 * j], r[18 + 4 * j], r[19 + 4 * j]});
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 48; ++k) asm volatile("" : "+v"(r[k]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  if (s == 123.456f) out[blockIdx.x * 64 + lane] = s;
}

int main(int argc, char** argv) {
  const bool bf16 = argc < 2 || strcmp(argv[1], "f32") != 0;
  const bool m16 = argc > 1 && strcmp(argv[1], "bf16_16") == 0;  // v_mfma_f32_16x16x32_bf16 (64 MFMAs of 16 Kflop per pass = the same flops)
  const double seconds = argc > 2 ? atof(argv[2]) : 8.0;
  // operands: bf16 / f32 values with exponents around 1.0 and random mantissas (finite, no denormals)
  std::vector<uint32_t> h(4096);
  uint32_t x = 0x9E3779B9u;
  for (auto& v : h) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    if (bf16) v = ((0x3F00u | (x & 0xFFu) | ((x >> 8) & 0x8000u)) << 16) | (0x3F00u | ((x >> 16) & 0xFFu) | ((x >> 9) & 0x8000u));
    else v = 0x3F000000u | (x & 0x807FFFFFu);
  }
  uint32_t* d_seed;
  float* d_out;
  CHECK(hipMalloc((void**)&d_seed, h.size() * 4));
  CHECK(hipMalloc((void**)&d_out, 1 << 20));
  CHECK(hipMemcpy(d_seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int grid = cus * 8;  // several workgroups per CU slot over the launch: 512 threads, 2 waves per SIMD resident
  const int iters = 200000;
  const double flop_per_launch = (double)grid * 8 /*waves*/ * iters * 32.0 /*MFMAs*/ * (bf16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (m16) hipLaunchKernelGGL(k_probe16, dim3(grid), dim3(512), 0, 0, d_seed, d_out, iters);
    else if (bf16) hipLaunchKernelGGL(k_probe<true>, dim3(grid), dim3(512), 0, 0, d_seed, d_out, iters);
    else hipLaunchKernelGGL(k_probe<false>, dim3(grid), dim3(512), 0, 0, d_seed, d_out, iters);
  };
  launch();
  CHECK(hipDeviceSynchronize());
  double total_ms = 0, best = 0, last = 0;
  int n = 0;
  while (total_ms < seconds * 1e3) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms;
    ++n;
    last = flop_per_launch / (ms * 1e-3) / 1e12;
    if (last > best) best = last;
  }
  const double mean = flop_per_launch * n / (total_ms * 1e-3) / 1e12;
  printf("{\"probe\": \"%s\", \"cus\": %d, \"launches\": %d, \"seconds\": %.2f, \"tflops_mean\": %.1f, \"tflops_first_best\": %.1f, "
         "\"tflops_last\": %.1f}\n",
         m16 ? "v_mfma_f32_16x16x32_bf16" : (bf16 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32"), cus, n, total_ms * 1e-3, mean, best, last);
  return 0;
}
