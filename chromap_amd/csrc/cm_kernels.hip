// cm_kernels.hip -- __global__ wrappers (one thread per item) around the stage functions of
// cm_stages.h, the index-probe kernel, prefix scans and the launch helpers.  gfx950 only.
#include <hip/hip_runtime.h>

#include "cm_kernels.h"
#include "cm_stages.h"

#define CM_BLOCK 256

#define CM_ITEM_KERNEL(kname, fn)                                            \
  __global__ __launch_bounds__(CM_BLOCK) void kname(CmDev d, uint32_t n) {   \
    const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;                  \
    if (i < n) fn(d, i);                                                     \
  }

CM_ITEM_KERNEL(k_s0_prep, cm_s0_prep)
CM_ITEM_KERNEL(k_s1_minimizers, cm_s1_minimizers)
CM_ITEM_KERNEL(k_s1b_compact, cm_s1b_compact)
CM_ITEM_KERNEL(k_s3a_count, cm_s3a_count)
CM_ITEM_KERNEL(k_s3b_candidates, cm_s3b_candidates)
CM_ITEM_KERNEL(k_s4a_rescue_count, cm_s4a_rescue_count)
CM_ITEM_KERNEL(k_s4b_rescue_merge, cm_s4b_rescue_merge)
CM_ITEM_KERNEL(k_s4c_reduce, cm_s4c_reduce)
CM_ITEM_KERNEL(k_s5_verify, cm_s5_verify)
CM_ITEM_KERNEL(k_s6a_pair, cm_s6a_pair)
CM_ITEM_KERNEL(k_s6c_multi, cm_s6c_multi)

__global__ __launch_bounds__(64) void k_s6b_sample(CmDev d, uint32_t n_chunks) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n_chunks) return;
  CmMt g;
  cm_s6b_sample(d, i, g);
}

// slot capacity of each read: one (hash,pos) per k-mer position at most
__global__ __launch_bounds__(CM_BLOCK) void k_slot_cap(CmDev d, uint32_t n_reads, uint32_t *cap) {
  const uint32_t r = blockIdx.x * CM_BLOCK + threadIdx.x;
  if (r >= n_reads) return;
  const uint32_t len = d.rlen[r];
  cap[r] = len >= (uint32_t)d.p.k ? len - (uint32_t)d.p.k + 1 : 0;
}

// ---------------------------------------------------------------------------------------
// The index-probe kernel (graded roofline kernel): one minimizer per thread, dependent
// 16-byte bucket gathers from the HBM-resident table; memory-level parallelism comes from
// the number of resident waves.  Probe steps are reduced per wave before the atomic.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(CM_BLOCK) void k_probe(const uint64_t *__restrict__ bkt, uint32_t bmask,
                                                     const uint64_t *__restrict__ hash, uint64_t *__restrict__ val,
                                                     uint8_t *__restrict__ kind, uint32_t n,
                                                     unsigned long long *__restrict__ counters) {
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  uint32_t steps = 0, hit = 0;
  if (i < n) {
    uint64_t v;
    uint8_t kd;
    steps = cm_probe(bkt, bmask, hash[i], &v, &kd);
    val[i] = v;
    kind[i] = kd;
    hit = kd != CM_PR_MISS;
  }
  if (counters) {
    // wave64 reduction
    for (int off = 32; off > 0; off >>= 1) {
      steps += __shfl_down(steps, off, 64);
      hit += __shfl_down(hit, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && steps) {
      atomicAdd(&counters[0], (unsigned long long)steps);
      atomicAdd(&counters[1], (unsigned long long)hit);
    }
  }
}

// per-pair counters of Chromap::OutputMappingStatistics (chromap.h:1057-1058, 1118-1137)
__global__ __launch_bounds__(CM_BLOCK) void k_stats(CmDev d, uint32_t n) {
  const uint32_t pair = blockIdx.x * CM_BLOCK + threadIdx.x;
  unsigned long long cand = 0, mappings = 0, mapped = 0, uniq = 0, multi = 0, resc = 0, occ = 0, nrec = 0;
  if (pair < n) {
    const uint32_t r1 = 2 * pair, r2 = r1 + 1;
    if (d.alive[pair]) {
      cand = d.fcp[r1] + d.fcn[r1] + d.fcp[r2] + d.fcn[r2];
      const uint32_t nd1 = d.ndp[r1] + d.ndn[r1], nd2 = d.ndp[r2] + d.ndn[r2];
      if (nd1 > 0 && nd2 > 0) {
        const int nb = d.pe_nbest[pair];
        if (nb == 1) uniq = 2;
        mappings = 2ull * (unsigned long long)(nb < d.p.max_best ? nb : d.p.max_best);
        if (nb > 0) mapped = 2;
        if (nb > 1 && nb <= d.p.drop_rep) multi = 1;
      }
    }
    resc = (unsigned long long)d.aug[r1] + d.aug[r2];
    occ = (unsigned long long)d.hit_tot[r1] + d.hit_tot[r2];
    nrec = d.rec_ok[pair];
  }
  __shared__ unsigned long long sh[8];
  if (threadIdx.x < 8) sh[threadIdx.x] = 0;
  __syncthreads();
  if (cand) atomicAdd(&sh[0], cand);
  if (mappings) atomicAdd(&sh[1], mappings);
  if (mapped) atomicAdd(&sh[2], mapped);
  if (uniq) atomicAdd(&sh[3], uniq);
  if (multi) atomicAdd(&sh[4], multi);
  if (resc) atomicAdd(&sh[5], resc);
  if (occ) atomicAdd(&sh[6], occ);
  if (nrec) atomicAdd(&sh[7], nrec);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (sh[0]) atomicAdd(&d.stats[CM_ST_CAND], sh[0]);
    if (sh[1]) atomicAdd(&d.stats[CM_ST_MAPPINGS], sh[1]);
    if (sh[2]) atomicAdd(&d.stats[CM_ST_MAPPED], sh[2]);
    if (sh[3]) atomicAdd(&d.stats[CM_ST_UNIQ], sh[3]);
    if (sh[4]) atomicAdd(&d.stats[CM_ST_MULTI], sh[4]);
    if (sh[5]) atomicAdd(&d.stats[CM_ST_RESCUED], sh[5]);
    if (sh[6]) atomicAdd(&d.stats[CM_ST_OCC], sh[6]);
    if (sh[7]) atomicAdd(&d.stats[CM_ST_RECORDS], sh[7]);
  }
}

// ---------------------------------------------------------------------------------------
// exclusive prefix sum of uint32 (n inputs -> n+1 outputs), three-phase, recursive
// ---------------------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_TILE (CM_BLOCK * SCAN_ITEMS)

__global__ __launch_bounds__(CM_BLOCK) void k_scan_tile(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                         uint32_t n, uint32_t *__restrict__ tile_sum) {
  __shared__ uint32_t wsum[CM_BLOCK / 64];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = base + j < n ? in[base + j] : 0;
    s += v[j];
  }
  // inclusive scan of s across the wave
  uint32_t inc = s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int j = 0; j < wv; ++j) woff += wsum[j];
  uint32_t run = woff + inc - s;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == CM_BLOCK - 1) tile_sum[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(CM_BLOCK) void k_scan_add(uint32_t *__restrict__ out, uint32_t n,
                                                        const uint32_t *__restrict__ tile_off) {
  const uint32_t add = tile_off[blockIdx.x];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j)
    if (base + j < n) out[base + j] += add;
}

__global__ void k_scan_total(const uint32_t *in, uint32_t *out, uint32_t n, const uint32_t *tile_off,
                             const uint32_t *tile_sum, uint32_t n_tiles) {
  // out[n] = total
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    (void)in;
    out[n] = n == 0 ? 0 : tile_off[n_tiles - 1] + tile_sum[n_tiles - 1];
  }
}

// tmp must hold at least cm_scan_tmp_words(n) uint32
size_t cm_scan_tmp_words(uint32_t n) {
  size_t words = 0;
  uint32_t m = n;
  while (true) {
    const uint32_t tiles = (m + SCAN_TILE - 1) / SCAN_TILE;
    words += 2 * (size_t)(tiles + 1) + 2;
    if (tiles <= 1) break;
    m = tiles;
  }
  return words + 16;
}

void cm_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *tmp, hipStream_t s) {
  if (n == 0) { (void)hipMemsetAsync(out, 0, sizeof(uint32_t), s); return; }
  const uint32_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t *tile_sum = tmp;            // [tiles]
  uint32_t *tile_off = tmp + tiles + 1; // [tiles+1]
  hipLaunchKernelGGL(k_scan_tile, dim3(tiles), dim3(CM_BLOCK), 0, s, in, out, n, tile_sum);
  if (tiles > 1) {
    cm_scan_u32(tile_sum, tile_off, tiles, tmp + 2 * (size_t)(tiles + 1) + 2, s);
    hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(CM_BLOCK), 0, s, out, n, tile_off);
  } else {
    (void)hipMemsetAsync(tile_off, 0, sizeof(uint32_t), s);
  }
  hipLaunchKernelGGL(k_scan_total, dim3(1), dim3(64), 0, s, in, out, n, tile_off, tile_sum, tiles);
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n) { return dim3((n + CM_BLOCK - 1) / CM_BLOCK); }

#define CM_LAUNCH(kname)                                                               \
  void cm_launch_##kname(const CmDev &d, uint32_t n, hipStream_t s) {                  \
    if (n) hipLaunchKernelGGL(kname, grid_for(n), dim3(CM_BLOCK), 0, s, d, n);         \
  }
CM_LAUNCH(k_s0_prep)
CM_LAUNCH(k_s1_minimizers)
CM_LAUNCH(k_s1b_compact)
CM_LAUNCH(k_s3a_count)
CM_LAUNCH(k_s3b_candidates)
CM_LAUNCH(k_s4a_rescue_count)
CM_LAUNCH(k_s4b_rescue_merge)
CM_LAUNCH(k_s4c_reduce)
CM_LAUNCH(k_s5_verify)
CM_LAUNCH(k_s6a_pair)
CM_LAUNCH(k_s6c_multi)
CM_LAUNCH(k_stats)

void cm_launch_k_s6b_sample(const CmDev &d, uint32_t n_chunks, hipStream_t s) {
  if (n_chunks) hipLaunchKernelGGL(k_s6b_sample, dim3((n_chunks + 63) / 64), dim3(64), 0, s, d, n_chunks);
}
void cm_launch_k_slot_cap(const CmDev &d, uint32_t n_reads, uint32_t *cap, hipStream_t s) {
  if (n_reads) hipLaunchKernelGGL(k_slot_cap, grid_for(n_reads), dim3(CM_BLOCK), 0, s, d, n_reads, cap);
}
void cm_launch_k_probe(const uint64_t *bkt, uint32_t bmask, const uint64_t *hash, uint64_t *val, uint8_t *kind,
                       uint32_t n, unsigned long long *counters, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_probe, grid_for(n), dim3(CM_BLOCK), 0, s, bkt, bmask, hash, val, kind, n, counters);
}
