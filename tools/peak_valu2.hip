// Issue rate of the simple VALU instructions the reduced-radix arithmetic leans on (companion of peak_valu.hip):
// eight independent chains, 32 instructions per loop body, one JSON line per instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/peak_valu2.hip -o build/peak_valu2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 2048
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R32(X) R8(X) R8(X) R8(X) R8(X)
#define OPS32 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define KERNEL32(NAME, BODY)                                                     \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* o) {                 \
    uint32_t a[8];                                                               \
    for (int c = 0; c < 8; c++) a[c] = threadIdx.x + c;                          \
    uint32_t s = o[threadIdx.x & 7] | 1, s2 = s * 3;                             \
    for (int i = 0; i < ITERS; i++) asm volatile(R32(BODY) : OPS32 : "v"(s), "v"(s2) : "vcc"); \
    uint32_t r = 0;                                                              \
    for (int c = 0; c < 8; c++) r ^= a[c];                                       \
    o[blockIdx.x * 256 + threadIdx.x] = r;                                       \
  }
#define KERNEL64(NAME, BODY)                                                     \
  __global__ void __launch_bounds__(256) k_##NAME(uint32_t* o) {                 \
    uint64_t a[8];                                                               \
    for (int c = 0; c < 8; c++) a[c] = threadIdx.x + c;                          \
    uint32_t s = o[threadIdx.x & 7] | 1, s2 = s * 3;                             \
    for (int i = 0; i < ITERS; i++) asm volatile(R32(BODY) : OPS32 : "v"(s), "v"(s2) : "vcc"); \
    uint64_t r = 0;                                                              \
    for (int c = 0; c < 8; c++) r ^= a[c];                                       \
    o[blockIdx.x * 256 + threadIdx.x] = (uint32_t)r;                             \
  }
#define B_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define B_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define B_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define B_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define B_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define B_OR(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define B_ALIGN(i) "v_alignbit_b32 %" #i ", %8, %" #i ", 29\n"
#define B_BFI(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define B_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define B_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define B_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define B_ADDLIT(i) "v_add_u32 %" #i ", 0x12345678, %" #i "\n"
#define B_LSHR64(i) "v_lshrrev_b64 %" #i ", 29, %" #i "\n"
#define B_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n"
#define B_MADLIT(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
KERNEL32(v_and_b32, B_AND)
KERNEL32(v_lshrrev_b32, B_LSHR)
KERNEL32(v_lshlrev_b32, B_LSHL)
KERNEL32(v_sub_u32, B_SUB)
KERNEL32(v_xor_b32, B_XOR)
KERNEL32(v_or_b32, B_OR)
KERNEL32(v_alignbit_b32, B_ALIGN)
KERNEL32(v_bfi_b32, B_BFI)
KERNEL32(v_and_or_b32, B_ANDOR)
KERNEL32(v_or3_b32, B_OR3)
KERNEL32(v_lshl_add_u32, B_LSHLADD)
KERNEL32(v_add_u32_literal, B_ADDLIT)
KERNEL64(v_lshrrev_b64, B_LSHR64)
KERNEL64(v_lshlrev_b64, B_LSHL64)
KERNEL64(v_mad_u64_u32, B_MADLIT)
template <class K>
static void run(const char* name, K kern, uint32_t* buf, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms, best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, buf);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ITERS * 32;
  printf("{\"instr\": \"%s\", \"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_at_2.4GHz\": %.2f}\n", name, ops / (best * 1e-3),
         1024.0 * 64 * 2.4e9 / (ops / (best * 1e-3)));
}
#define RUN(NAME) run(#NAME, k_##NAME, buf, blocks)
int main() {
  int blocks = 256 * 32;
  uint32_t* buf;
  hipMalloc(&buf, (size_t)blocks * 256 * 4);
  hipMemset(buf, 1, (size_t)blocks * 256 * 4);
  RUN(v_and_b32); RUN(v_lshrrev_b32); RUN(v_lshlrev_b32); RUN(v_sub_u32); RUN(v_xor_b32); RUN(v_or_b32); RUN(v_alignbit_b32);
  RUN(v_bfi_b32); RUN(v_and_or_b32); RUN(v_or3_b32); RUN(v_lshl_add_u32); RUN(v_add_u32_literal); RUN(v_lshrrev_b64);
  RUN(v_lshlrev_b64); RUN(v_mad_u64_u32);
  return 0;
}
