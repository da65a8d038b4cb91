// Integer-multiply issue-rate microbenchmark for gfx950: measures the chip-wide rate of
// v_mad_u64_u32 (32x32+64 -> 64 multiply-accumulate) and v_mul_lo/hi_u32 from dependent-free chains.
// The MSM roofline is priced against the measured v_mad_u64_u32 rate.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 4096
#define CHAINS 8

__global__ void __launch_bounds__(256) k_mad64(uint64_t* out, uint32_t a, uint32_t b) {
  uint64_t acc[CHAINS];
  uint32_t x = a + threadIdx.x, y = b ^ blockIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = (uint64_t)x * (uint32_t)(y + c) + acc[c];
    x = (uint32_t)acc[0];
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s ^= acc[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mullo(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[CHAINS];
  uint32_t y = b ^ blockIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + a;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = acc[c] * (y + c);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s ^= acc[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_mulhi(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[CHAINS];
  uint32_t y = b ^ blockIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + a;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = __umulhi(acc[c] | 0x80000000u, (y + c) | 0x80000000u);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s ^= acc[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_add(uint32_t* out, uint32_t a, uint32_t b) {
  uint32_t acc[CHAINS];
  uint32_t y = b ^ blockIdx.x;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = c + threadIdx.x + a;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc[c] = (acc[c] ^ y) + (acc[(c + 1) % CHAINS]);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s ^= acc[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K, class T>
static double run(K kern, T* buf, int blocks, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, 3u, 5u);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf, 3u, 5u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ITERS * CHAINS;
  double rate = ops / (best * 1e-3);
  printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"ops_per_s\": %.4e}\n", name, best, rate);
  return rate;
}

int main() {
  int blocks = 256 * 32;
  void* buf;
  hipMalloc(&buf, (size_t)blocks * 256 * 8);
  run(k_mad64, (uint64_t*)buf, blocks, "v_mad_u64_u32");
  run(k_mullo, (uint32_t*)buf, blocks, "v_mul_lo_u32");
  run(k_mulhi, (uint32_t*)buf, blocks, "v_mul_hi_u32");
  run(k_add, (uint32_t*)buf, blocks, "v_add_u32(+xor)");
  return 0;
}
