// Random 64-byte gathers (four 16-byte loads per lane, as k_accum reads affine points) over tables of 1..32 GiB:
// does a precomputed-multiples table of tens of GB keep the gather rate of a 1 GiB table?  One JSON line per span.
//   hipcc -O3 --offload-arch=gfx950 tools/gather_span.hip -o build/gather_span
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_gather64(const uint4* __restrict__ rec, uint64_t nrec_mask, uint32_t* __restrict__ out, uint32_t seed) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t h = (uint64_t)(i + seed) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  const uint4* p = rec + (size_t)(h & nrec_mask) * 4;
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  out[i] = a.x ^ b.y ^ c.z ^ d.w;
}
int main() {
  const uint32_t lanes = 1u << 26;  // 4 GiB gathered per launch
  uint32_t* out;
  if (hipMalloc(&out, (size_t)lanes * 4) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int lg = 30; lg <= 35; lg++) {
    uint4* rec;
    const size_t bytes = (size_t)1 << lg;
    if (hipMalloc(&rec, bytes) != hipSuccess) { printf("{\"span_GiB\": %zu, \"error\": \"alloc\"}\n", bytes >> 30); return 1; }
    hipMemset(rec, 1, bytes);
    const uint64_t mask = (bytes / 64) - 1;
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_gather64, dim3(lanes / 256), dim3(256), 0, 0, rec, mask, out, (uint32_t)r * lanes);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    printf("{\"span_GiB\": %zu, \"gathered_GiB\": 4, \"ms\": %.3f, \"GBps\": %.1f}\n", bytes >> 30, best, (double)lanes * 64 / (best * 1e-3) / 1e9);
    fflush(stdout);
    hipFree(rec);
  }
  return 0;
}
