// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit elements does lane l receive, as a function of the addresses the lanes
// of its 16-lane group supply?  LDS holds u16 value = element index; pattern p selects the per-lane byte address.
//   hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void k(int pattern, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint16_t*)lds;
  unsigned addr;
  if (pattern == 0) addr = base + 8u * l;                                  // contiguous 8 B per lane
  else if (pattern == 1) addr = base + 8u * (l & 15) + 1024u * (l >> 4);   // each 16-lane group its own 128-B block, 1 KB apart
  else if (pattern == 2) addr = base + 512u * ((l & 15) >> 2) + 8u * (l & 3) + 32u * (l >> 4);  // K-major rows 512 B apart
  else addr = base + 8u * (15 - (l & 15)) + 128u * (l >> 4);               // reversed lane order inside a group
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
  uint16_t* d;
  hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int p = 0; p < 4; ++p) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  }
  return 0;
}
