// Calibration of rocprofv3's FETCH_SIZE for k_accum's access pattern: every lane reads one random 64-byte record with
// four 16-byte loads (no reuse: 2^24 lanes over 2^26 records = 4 GiB).  Algorithmic bytes: 2^24 x 64 = 1 GiB.
//   hipcc -O3 --offload-arch=gfx950 tools/calib_gather.hip -o build/calib_gather
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./build/calib_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_gather64(const uint4* __restrict__ rec, uint32_t nrec_mask, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = i * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const uint4* p = rec + (size_t)(h & nrec_mask) * 4;
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  out[i] = a.x ^ b.y ^ c.z ^ d.w;
}
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
int main() {
  const uint32_t nrec = 1u << 26, lanes = 1u << 24;
  uint4* rec; uint32_t* out;
  if (hipMalloc(&rec, (size_t)nrec * 64) != hipSuccess || hipMalloc(&out, (size_t)lanes * 4) != hipSuccess) return 1;
  hipMemset(rec, 1, (size_t)nrec * 64);
  hipLaunchKernelGGL(k_gather64, dim3(lanes / 256), dim3(256), 0, 0, rec, nrec - 1, out);
  // streaming reference: 1 GiB read + 1 GiB written with 16-byte accesses
  hipLaunchKernelGGL(k_stream, dim3((1u << 26) / 256), dim3(256), 0, 0, rec, rec + ((size_t)1 << 27), (size_t)1 << 26);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
