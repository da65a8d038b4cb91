// Standalone C-ABI exerciser (no Python): device-resident vs host scalars must give the same MSM result.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/zeth_gpu.h"
#define CK(x) do { int s_ = (x); if (s_) { printf("FAIL %s -> %s\n", #x, zg_strerror(s_)); return 1; } } while (0)
int main(int argc, char** argv) {
  size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1024;
  zg_ctx* ctx; CK(zg_ctx_create(0, &ctx));
  uint8_t t0[32] = {1, 2, 3}, d[32] = {9, 8, 7, 6};
  zg_bases* b; CK(zg_bases_generate(ctx, 0, 1, n, t0, d, &b));
  void* ds; CK(zg_dev_malloc(ctx, n * 32, &ds));
  CK(zg_scalars_random(ctx, 0, ds, n, 42));
  std::vector<uint8_t> hs(n * 32);
  CK(zg_dev_download(ctx, hs.data(), ds, n * 32));
  uint8_t r1[64], r2[64], r3[64];
  CK(zg_msm(b, 0, ds, n, 0, 1, 0, r1));
  CK(zg_msm(b, 0, hs.data(), n, 0, 0, 0, r2));
  CK(zg_msm(b, 0, ds, n, 0, 1, 0, r3));
  printf("n=%zu dev==host:%d dev==dev2:%d  r1[0..3]=%02x%02x%02x%02x r2=%02x%02x%02x%02x\n", n, !memcmp(r1, r2, 64), !memcmp(r1, r3, 64),
         r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3]);
  return memcmp(r1, r2, 64) != 0;
}
