// Fixed-base table mode of the MSM: the bases of a proving key never change, and 288 GB of HBM hold D = ceil(bits / c)
// precomputed multiples T_j[i] = 2^(j c) P_i of every base.  Then sum_i s_i P_i = sum_i sum_j d_ij T_j[i] with the signed
// c-bit digits d_ij of s_i, and ALL n * D terms fall into one set of 2^(c-1) buckets: the bucket reduction is paid once
// instead of once per window, so c can grow to 22 (12 digits per 254-bit scalar where the windowed form needs 16
// or 17) -- a quarter fewer mixed additions in the dominant kernel for 12x the base memory.
//
// Sorting n * D (entry, bucket) pairs on 21 bucket bits is a three-level counting sort.  Bucket b (weight b + 1) belongs
// to group b mod W under key b / W: the top digit of a scalar takes few values (12 bits at c = 22), so the lowest buckets
// hold n / 4096 entries each on top of their share -- the low bits spread them evenly over the groups.
//   k_table_count / k_table_starts / k_table_scatter   by bucket group (<= 128 groups of 2^15 buckets)
//   k_scatter_hi_table, k_scatter_lo                    inside a group, as the windowed form sorts a window (msm.cuh)
// Every level bins a tile in LDS and copies it out in runs, so global stores stay line-sized.
#pragma once
#include "msm.cuh"

namespace zg {

static constexpr uint32_t TABLE_TILE = 8192;  // entries per tile of the group-level pass (tile scalars = TABLE_TILE / D)

// entries per bucket group and (optionally) the number of scalars other than 0 and 1
template <class Fr>
__global__ void __launch_bounds__(256) k_table_count(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int D,
                                                      AddConst K, uint32_t gmask /* W - 1 */, uint32_t* __restrict__ gcnt /* [W] */,
                                                      uint32_t* __restrict__ dense_count) {
  __shared__ uint32_t h[128], nd;
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  if (threadIdx.x == 0) nd = 0;
  __syncthreads();
  uint32_t my_dense = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    bool dense;
    recode_scalar<Fr>(scalars, i, mont, c, D, K, &dense, [&](int, int d) {
      if (d != 0) atomicAdd(&h[(uint32_t)((d < 0 ? -d : d) - 1) & gmask], 1u);
    });
    my_dense += dense ? 1u : 0u;
  }
  if (dense_count && my_dense) atomicAdd(&nd, my_dense);
  __syncthreads();
  if (threadIdx.x < 128 && h[threadIdx.x]) atomicAdd(&gcnt[threadIdx.x], h[threadIdx.x]);
  if (dense_count && threadIdx.x == 0 && nd) atomicAdd(dense_count + (blockIdx.x & 63u), nd);
}

static __global__ void __launch_bounds__(64) k_table_starts(const uint32_t* __restrict__ gcnt, uint32_t W,
                                                            uint32_t* __restrict__ gstart /* [W + 1] */,
                                                            uint32_t* __restrict__ gcursor /* [W] */) {
  ZG_WAVE_PRIO(3);
  if (threadIdx.x != 0) return;
  uint32_t run = 0;
  for (uint32_t w = 0; w < W; w++) {
    gstart[w] = run;
    gcursor[w] = run;
    run += gcnt[w];
  }
  gstart[W] = run;
}

// Two-ended layout (tables of one or two bucket groups, c <= 17: the prover's queries up to 2^19 points).  The level-0
// partition needs every group's start, i.e. a counting pass over the scalars before it (k_table_count, k_table_starts).
// With at most two groups it does not: group 0 fills the entry array from its front and group 1 from its back, each
// through one cursor, and the cursors ARE the groups' end and start afterwards -- gse[0..1] = starts {0, cursor of group
// 1}, gse[2..3] = ends {cursor of group 0, E}.  Two launches fewer at the head of every such evaluation (~75 us of the
// H chain inside a proof).  `gse` is null in the general layout, where gcursor[] holds the counted starts.
__device__ __forceinline__ uint32_t table_claim(uint32_t* __restrict__ gcursor, uint32_t* __restrict__ gse, uint32_t g, uint32_t v) {
  if (!gse) return atomicAdd(&gcursor[g], v);
  return g == 0 ? atomicAdd(&gse[2], v) : atomicSub(&gse[1], v) - v;
}

// One tile of scalars per workgroup, one scalar per lane: bin the tile's entries by bucket group in LDS, claim a run
// per group, copy out.  Entry = (j * N + first + i) | sign << 31 (index into the table), key = bucket within the group.
template <class Fr>
__global__ void __launch_bounds__(1024) k_table_scatter(const uint32_t* __restrict__ scalars, size_t n, int mont, int c, int D,
                                                         AddConst K, int gbits, uint32_t W, uint32_t tile_scalars,
                                                         uint32_t table_n, uint32_t first, uint32_t* __restrict__ gcursor,
                                                         uint32_t* __restrict__ gse, uint32_t* __restrict__ fent, uint16_t* __restrict__ fkey) {
  __shared__ uint32_t h[128], lstart[128], gbase[128], wsum;
  __shared__ uint32_t buf_ent[TABLE_TILE];
  __shared__ uint16_t buf_key[TABLE_TILE];
  __shared__ uint8_t buf_bin[TABLE_TILE];
  const size_t i = (size_t)blockIdx.x * tile_scalars + threadIdx.x;
  const bool live = threadIdx.x < tile_scalars && i < n;
  const uint32_t gmask = W - 1u;
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  bool dense;
  if (live)
    recode_scalar<Fr>(scalars, i, mont, c, D, K, &dense, [&](int, int d) {
      if (d != 0) atomicAdd(&h[(uint32_t)((d < 0 ? -d : d) - 1) & gmask], 1u);
    });
  __syncthreads();
  uint32_t v = 0, incl = 0;
  if (threadIdx.x < 128) {
    v = threadIdx.x < W ? h[threadIdx.x] : 0;
    incl = wave_inclusive_scan(v);
    if (threadIdx.x == 63) wsum = incl;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const uint32_t excl = incl - v + (threadIdx.x >= 64 ? wsum : 0);
    lstart[threadIdx.x] = excl;
    gbase[threadIdx.x] = v ? table_claim(gcursor, gse, threadIdx.x, v) : 0;
    h[threadIdx.x] = excl;
  }
  __syncthreads();
  if (live)
    recode_scalar<Fr>(scalars, i, mont, c, D, K, &dense, [&](int j, int d) {
      if (d != 0) {
        const uint32_t b = (uint32_t)((d < 0 ? -d : d) - 1);
        const uint32_t pos = atomicAdd(&h[b & gmask], 1u);
        buf_ent[pos] = ((uint32_t)j * table_n + first + (uint32_t)i) | (d < 0 ? 0x80000000u : 0u);
        buf_key[pos] = (uint16_t)(b >> gbits);
        buf_bin[pos] = (uint8_t)(b & gmask);
      }
    });
  __syncthreads();
  const uint32_t total = h[W - 1];  // the last group's running cursor ends at the tile's entry count
  for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
    const uint32_t bin = buf_bin[q];
    const uint32_t g = gbase[bin] + (q - lstart[bin]);
    fent[g] = buf_ent[q];
    fkey[g] = buf_key[q];
  }
}

// The same for a digit width known at compile time (the widths msm_table_pick_c chooses): the scalar is recoded ONCE, its
// digits and their ranks within their groups stay in registers, and every entry costs one LDS atomic instead of two
// (2^24 points, c = 22: 0.77 -> see DESIGN 4.2).
template <class Fr, int C>
__global__ void __launch_bounds__(1024) k_table_scatter_static(const uint32_t* __restrict__ scalars, size_t n, int mont, AddConst K,
                                                                int gbits, uint32_t W, uint32_t tile_scalars, uint32_t table_n,
                                                                uint32_t first, uint32_t* __restrict__ gcursor, uint32_t* __restrict__ gse,
                                                                uint32_t* __restrict__ fent, uint16_t* __restrict__ fkey) {
  constexpr int D = msm_table_digits(Fr::Params::BITS, C);
  __shared__ uint32_t h[128], lstart[128], gbase[128], wsum;
  __shared__ uint32_t buf_ent[TABLE_TILE];
  __shared__ uint16_t buf_key[TABLE_TILE];
  __shared__ uint8_t buf_bin[TABLE_TILE];
  const size_t i = (size_t)blockIdx.x * tile_scalars + threadIdx.x;
  const bool live = threadIdx.x < tile_scalars && i < n;
  const uint32_t gmask = W - 1u;
  if (threadIdx.x < 128) h[threadIdx.x] = 0;
  __syncthreads();
  int dg[D];
  uint32_t rk[D];
  if (live) {
    bool dense;
    recode_scalar_static<Fr, C, D>(scalars, i, mont, K, &dense, dg);
#pragma unroll
    for (int j = 0; j < D; j++)
      if (dg[j] != 0) rk[j] = atomicAdd(&h[(uint32_t)((dg[j] < 0 ? -dg[j] : dg[j]) - 1) & gmask], 1u);
  }
  __syncthreads();
  uint32_t v = 0, incl = 0;
  if (threadIdx.x < 128) {
    v = threadIdx.x < W ? h[threadIdx.x] : 0;
    incl = wave_inclusive_scan(v);
    if (threadIdx.x == 63) wsum = incl;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const uint32_t excl = incl - v + (threadIdx.x >= 64 ? wsum : 0);
    lstart[threadIdx.x] = excl;
    gbase[threadIdx.x] = v ? table_claim(gcursor, gse, threadIdx.x, v) : 0;
  }
  __syncthreads();
  if (live) {
#pragma unroll
    for (int j = 0; j < D; j++)
      if (dg[j] != 0) {
        const uint32_t b = (uint32_t)((dg[j] < 0 ? -dg[j] : dg[j]) - 1);
        const uint32_t pos = lstart[b & gmask] + rk[j];
        buf_ent[pos] = ((uint32_t)j * table_n + first + (uint32_t)i) | (dg[j] < 0 ? 0x80000000u : 0u);
        buf_key[pos] = (uint16_t)(b >> gbits);
        buf_bin[pos] = (uint8_t)(b & gmask);
      }
  }
  __syncthreads();
  const uint32_t total = lstart[W - 1] + h[W - 1];
  for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
    const uint32_t bin = buf_bin[q];
    const uint32_t g = gbase[bin] + (q - lstart[bin]);
    fent[g] = buf_ent[q];
    fkey[g] = buf_key[q];
  }
}

// bucket counts of a group from its key list (k_hist's counterpart); grid (chunks, W)
static __global__ void __launch_bounds__(1024) k_hist_table(const uint16_t* __restrict__ fkey, const uint32_t* __restrict__ gstart,
                                                            const uint32_t* __restrict__ gend, uint32_t nb, uint32_t* __restrict__ cnt) {
  extern __shared__ uint32_t lh[];
  const uint32_t w = blockIdx.y;
  const uint32_t start = gstart[w], end = gend[w];
  if (start == end) return;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  const uint32_t per = (end - start + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = start + blockIdx.x * per < end ? start + blockIdx.x * per : end;
  const uint32_t hi = lo + per < end ? lo + per : end;
  // the bulk eight keys (16 bytes) per lane and load, two loads in flight; the unaligned ends one key per lane
  const uint32_t lo8 = (lo + 7u) & ~7u, hi8 = hi & ~7u;
  if (lo8 < hi8) {
    for (uint32_t q = lo + threadIdx.x; q < lo8; q += blockDim.x) atomicAdd(&lh[fkey[q]], 1u);
    const uint4* v = reinterpret_cast<const uint4*>(fkey + lo8);
    const uint32_t nv = (hi8 - lo8) >> 3;
    auto count = [&](const uint4& k) {
      atomicAdd(&lh[k.x & 0xFFFFu], 1u);
      atomicAdd(&lh[k.x >> 16], 1u);
      atomicAdd(&lh[k.y & 0xFFFFu], 1u);
      atomicAdd(&lh[k.y >> 16], 1u);
      atomicAdd(&lh[k.z & 0xFFFFu], 1u);
      atomicAdd(&lh[k.z >> 16], 1u);
      atomicAdd(&lh[k.w & 0xFFFFu], 1u);
      atomicAdd(&lh[k.w >> 16], 1u);
    };
    uint32_t i = threadIdx.x;
    for (; i + blockDim.x < nv; i += 2 * blockDim.x) {
      const uint4 k0 = v[i], k1 = v[i + blockDim.x];
      count(k0);
      count(k1);
    }
    if (i < nv) count(v[i]);
    for (uint32_t q = hi8 + threadIdx.x; q < hi; q += blockDim.x) atomicAdd(&lh[fkey[q]], 1u);
  } else {
    for (uint32_t q = lo + threadIdx.x; q < hi; q += blockDim.x) atomicAdd(&lh[fkey[q]], 1u);
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
    const uint32_t v = lh[b];
    if (v) atomicAdd(&cnt[(size_t)w * nb + b], v);
  }
}

// single-level finish for groups of fewer than 512 buckets (small tables): k_scatter's counterpart; grid (chunks, W)
static __global__ void __launch_bounds__(1024) k_scatter_table(const uint32_t* __restrict__ fent, const uint16_t* __restrict__ fkey,
                                                               const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gend,
                                                               uint32_t nb, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t lh[];
  const uint32_t w = blockIdx.y;
  const uint32_t start = gstart[w], end = gend[w];
  if (start == end) return;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  const uint32_t per = (end - start + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = start + blockIdx.x * per < end ? start + blockIdx.x * per : end;
  const uint32_t hi = lo + per < end ? lo + per : end;
  for (uint32_t q = lo + threadIdx.x; q < hi; q += blockDim.x) atomicAdd(&lh[fkey[q]], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
    const uint32_t v = lh[b];
    if (v) lh[b] = atomicAdd(&cursor[(size_t)w * nb + b], v);
  }
  __syncthreads();
  for (uint32_t q = lo + threadIdx.x; q < hi; q += blockDim.x) sorted[atomicAdd(&lh[fkey[q]], 1u)] = fent[q];
}

// k_scatter_hi's counterpart: tiles of a group's (entry, key) list split over the group's partitions of 256 buckets;
// grid (tile slots, W), tiles taken grid-stride so that a group holding most entries is still covered
static __global__ void __launch_bounds__(SORT_WG) k_scatter_hi_table(const uint32_t* __restrict__ fent, const uint16_t* __restrict__ fkey,
                                                                  const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ gend,
                                                                  uint32_t nhi, uint32_t* __restrict__ cursor1, uint32_t* __restrict__ mid_ent,
                                                                  uint8_t* __restrict__ mid_lo) {
  __shared__ uint32_t h[128], lstart[128], gbase[128], wsum;
  __shared__ uint32_t buf_ent[SORT_TILE1T];
  __shared__ uint8_t buf_lo[SORT_TILE1T], buf_bin[SORT_TILE1T];
  const uint32_t w = blockIdx.y;
  const uint32_t start = gstart[w], end = gend[w];
  // Tiles start on multiples of SORT_PER entries (the first one below `start`, its leading entries masked), so that a lane
  // takes consecutive entries, eight at a time with one 16-byte load of keys and two of entries: with one narrow load per entry the
  // kernel waited on load latency, not bandwidth.  Which lane ranks which entry does not matter to a counting sort.
  for (uint32_t lo = (start & ~(SORT_PER - 1u)) + blockIdx.x * SORT_TILE1T; lo < end; lo += gridDim.x * SORT_TILE1T) {
    const uint32_t hi = lo + SORT_TILE1T < end ? lo + SORT_TILE1T : end;
    if (threadIdx.x < 128) h[threadIdx.x] = 0;
    __syncthreads();
    // one pass: entries, keys and ranks within their bins stay in registers (one LDS atomic per entry)
    constexpr uint32_t PER = SORT_PER;
    uint32_t my_ent[PER], my_key[PER], my_rank[PER];
    const uint32_t q0 = lo + threadIdx.x * PER;
#pragma unroll
    for (uint32_t c = 0; c < PER; c += 8) {
      if (q0 + c < hi) {  // reads up to seven entries past `end`: inside the workspace (every array is followed by another)
        const uint4 k = *reinterpret_cast<const uint4*>(fkey + q0 + c);
        const uint4 e0 = *reinterpret_cast<const uint4*>(fent + q0 + c);
        const uint4 e1 = *reinterpret_cast<const uint4*>(fent + q0 + c + 4);
        my_key[c + 0] = k.x & 0xFFFFu, my_key[c + 1] = k.x >> 16, my_key[c + 2] = k.y & 0xFFFFu, my_key[c + 3] = k.y >> 16;
        my_key[c + 4] = k.z & 0xFFFFu, my_key[c + 5] = k.z >> 16, my_key[c + 6] = k.w & 0xFFFFu, my_key[c + 7] = k.w >> 16;
        my_ent[c + 0] = e0.x, my_ent[c + 1] = e0.y, my_ent[c + 2] = e0.z, my_ent[c + 3] = e0.w;
        my_ent[c + 4] = e1.x, my_ent[c + 5] = e1.y, my_ent[c + 6] = e1.z, my_ent[c + 7] = e1.w;
      }
    }
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
      const uint32_t q = q0 + r;
      if (q >= start && q < hi)
        my_rank[r] = atomicAdd(&h[my_key[r] >> 8], 1u);
      else
        my_key[r] = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t v = 0, incl = 0;
    if (threadIdx.x < 128) {
      v = threadIdx.x < nhi ? h[threadIdx.x] : 0;
      incl = wave_inclusive_scan(v);
      if (threadIdx.x == 63) wsum = incl;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const uint32_t excl = incl - v + (threadIdx.x >= 64 ? wsum : 0);
      lstart[threadIdx.x] = excl;
      gbase[threadIdx.x] = v ? atomicAdd(&cursor1[(size_t)w * nhi + threadIdx.x], v) : 0;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < PER; r++) {
      if (my_key[r] != 0xFFFFFFFFu) {
        const uint32_t pos = lstart[my_key[r] >> 8] + my_rank[r];
        buf_ent[pos] = my_ent[r];
        buf_lo[pos] = (uint8_t)(my_key[r] & 255u);
        buf_bin[pos] = (uint8_t)(my_key[r] >> 8);
      }
    }
    __syncthreads();
    const uint32_t total = hi - (lo > start ? lo : start);
    for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
      const uint32_t bin = buf_bin[q];
      const uint32_t g = gbase[bin] + (q - lstart[bin]);
      mid_ent[g] = buf_ent[q];
      mid_lo[g] = buf_lo[q];
    }
    __syncthreads();
  }
}

// next level of the table: out[i] = 2^c * prev[i] (XYZZ; normalised to affine by k_normalize afterwards)
template <class F>
__global__ void __launch_bounds__(256) k_table_next(const Affine<F>* __restrict__ prev, size_t n, int c, XYZZ<F>* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  XYZZ<F> p = XYZZ<F>::from_affine(prev[i]);
  for (int k = 0; k < c; k++) p = xyzz_dbl(p);
  out[i] = p;
}

}  // namespace zg
