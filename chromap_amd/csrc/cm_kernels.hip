// cm_kernels.hip -- __global__ wrappers (one thread per item) around the stage functions of
// cm_stages.h, the index-probe kernel, prefix scans and the launch helpers.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <rocprim/rocprim.hpp>

#include "cm_kernels.h"
#include "cm_stages.h"
#include "cm_coop.h"

#define CM_BLOCK 256

// One item per lane.  `perm` (d.perm_reads / d.perm_pairs, or nullptr) lists the items with the few heavy ones -- reads
// from repeats, whose candidate lists are a hundred times longer -- at the end: a wave then holds either light items or
// heavy items, instead of 63 light lanes waiting for one heavy lane in nearly every wave.
#define CM_ITEM_KERNEL(kname, fn, perm)                                      \
  __global__ __launch_bounds__(CM_BLOCK) void kname(CmDev d, uint32_t n) {   \
    if (d.abort && *d.abort) return;                                         \
    const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;                  \
    if (i < n) fn(d, d.perm ? d.perm[i] : i);                                \
  }



// ---------------------------------------------------------------------------------------
// Read staging: a block owns PB consecutive pairs; the bytes of their mate-0 reads are one
// contiguous range of rb0 (likewise rb1), copied to LDS with coalesced 16-byte loads so that
// every read byte leaves HBM once (per-thread strided byte loads thrash L1/L2: rocprofv3
// FETCH_SIZE showed 14x-40x the read bytes for the first version of these kernels).
// LDS layout: [mate-0 range][mate-1 range], each starting at the 16-byte-aligned address
// below its first byte.  Returns this thread's read pointer in LDS.
// ---------------------------------------------------------------------------------------
extern __shared__ __align__(16) uint8_t cm_lds[];
static inline dim3 grid_for_n(uint32_t n) { return dim3((n + CM_BLOCK - 1) / CM_BLOCK); }

__device__ __forceinline__ void cm_stage_range(uint8_t *dst, const uint8_t *src, uint64_t a0, uint64_t g1) {
  for (uint64_t off = a0 + (uint64_t)threadIdx.x * 16; off < g1; off += (uint64_t)blockDim.x * 16)
    *reinterpret_cast<uint4 *>(dst + (off - a0)) = *reinterpret_cast<const uint4 *>(src + off);
}

// stages the reads of pairs [p0, p1) and returns the LDS pointers of pair `pair`'s reads
struct CmStaged { const uint8_t *m0, *m1; };
__device__ __forceinline__ CmStaged cm_stage_pairs(const CmDev &d, uint32_t p0, uint32_t p1, uint32_t pair, uint32_t lds_half) {
  const uint64_t g0a = d.ro0[p0], g0b = d.ro0[p1], g1a = d.ro1[p0], g1b = d.ro1[p1];
  const uint64_t a0 = g0a & ~15ull, a1 = g1a & ~15ull;
  uint8_t *l0 = cm_lds, *l1 = cm_lds + lds_half;
  cm_stage_range(l0, d.rb0, a0, g0b);
  cm_stage_range(l1, d.rb1, a1, g1b);
  __syncthreads();
  CmStaged s;
  s.m0 = l0 + (d.ro0[pair < p1 ? pair : p0] - a0);
  s.m1 = l1 + (d.ro1[pair < p1 ? pair : p0] - a1);
  return s;
}

// S0 + S1 count, fused: threads [0,PB) trim their pair, then all 2*PB threads (one per read)
// run the minimizer state machine in counting mode.  blockDim.x = 2*PB.
__global__ void k_prep_count(CmDev d, uint32_t n_pairs, uint32_t lds_half) {
  const uint32_t PB = blockDim.x >> 1;
  const uint32_t p0 = blockIdx.x * PB, p1 = p0 + PB < n_pairs ? p0 + PB : n_pairs;
  const uint32_t t = threadIdx.x, lp = t < PB ? t : t - PB, pair = p0 + lp;
  const CmStaged s = cm_stage_pairs(d, p0, p1, pair, lds_half);
  if (t < PB && pair < p1) cm_s0_prep_ptr(d, pair, s.m0, s.m1);
  __syncthreads();  // rlen of both mates is read below by other threads of this block
  if (pair < p1) cm_s1_count(d, 2 * pair + (t < PB ? 0 : 1), t < PB ? s.m0 : s.m1);
}

// S1 fill: same staging, minimizers written directly to their dense positions
__global__ void k_mm_fill(CmDev d, uint32_t pair_lo, uint32_t n_pairs /* end of this launch's pair range */, uint32_t lds_half) {
  const uint32_t PB = blockDim.x >> 1;
  const uint32_t p0 = pair_lo + blockIdx.x * PB, p1 = p0 + PB < n_pairs ? p0 + PB : n_pairs;
  const uint32_t t = threadIdx.x, lp = t < PB ? t : t - PB, pair = p0 + lp;
  const CmStaged s = cm_stage_pairs(d, p0, p1, pair, lds_half);
  if (pair < p1) cm_s1_fill(d, 2 * pair + (t < PB ? 0 : 1), t < PB ? s.m0 : s.m1);
}

// S0 + S1 in ONE pass (k = 17..26, w = 7): the minimizer state machine runs once per read; its
// emissions are staged in LDS ([entry][thread], one packed u64 = hash | (pos<<1|strand) << 2k),
// the block scans its counts, reserves a range of the dense arrays with one atomic and copies
// the staged entries out.  mm_off is therefore not monotone in the read index across blocks --
// every consumer addresses a read's list through (mm_off[r], mm_cnt[r]).  A read with more than
// `stg` minimizers (never seen: stg = L/4 + 4 against an expected (L-16)/4) recomputes straight
// into its range.  Replaces k_prep_count + scan + k_mm_fill (two passes of ~150 integer ops per base).
// GSTAGE (reads longer than 69 bases): the emissions are staged in the block's tile of a global buffer instead -- the same
// [entry][thread] layout, so the lanes of a wave fill whole lines -- because stg x threads x 8 bytes of LDS would leave one
// block per CU to a VALU-bound kernel; the tile is copied out by OUTPUT position (which read a dense slot belongs to is a
// search in the block's 256 offsets), so the dense arrays are written in whole lines too.  Before, such reads were hashed
// twice (k_prep_count, k_mm_fill) around a scan and two host waits.
// E6 (shared-memory staging only): an emission is staged as 32 + 16 bits (hash | position-strand << 2k fits 48 bits for k <= 20 and
// reads of up to 127 bases) -- 24 instead of 32 KB of emissions per block of 256 lanes: four blocks per CU instead of three for this
// VALU-bound kernel (three Hash64 per base are the reference's own: the minimizer hash is the hash of the smaller strand's hash)
template <bool GSTAGE, bool E6>
__global__ void k_prep_mm(CmDev d, uint32_t pair_lo, uint32_t n_pairs /* end of this launch's pair range */, uint32_t lds_half, uint32_t stg,
                          uint32_t mm_cap, unsigned long long *cursor, uint64_t *gstage) {
  const uint32_t T = blockDim.x, PB = T >> 1;
  const uint32_t p0 = pair_lo + blockIdx.x * PB, p1 = p0 + PB < n_pairs ? p0 + PB : n_pairs;
  const uint32_t t = threadIdx.x, lp = t < PB ? t : t - PB, pair = p0 + lp;
  const CmStaged s = cm_stage_pairs(d, p0, p1, pair, lds_half);
  if (t < PB && pair < p1) cm_s0_prep_ptr(d, pair, s.m0, s.m1);
  __syncthreads();
  uint64_t *sh_e = GSTAGE ? gstage + (size_t)blockIdx.x * stg * T : reinterpret_cast<uint64_t *>(cm_lds + 2 * lds_half);
  uint32_t *sh_lo = reinterpret_cast<uint32_t *>(cm_lds + 2 * lds_half);  // E6: the emissions' low words, then their bits 32-47
  uint16_t *sh_hi = reinterpret_cast<uint16_t *>(sh_lo + (size_t)stg * T);
  uint32_t *sh_w = reinterpret_cast<uint32_t *>(cm_lds + 2 * lds_half + (GSTAGE ? 0 : (size_t)stg * T * (E6 ? 6 : 8)));  // wave totals [8], base lo/hi [2]
  uint32_t *sh_off = sh_w + 16, *sh_cnt = sh_off + T + 1;  // GSTAGE: the reads' offsets in the block's range [T + 1], their counts [T]
  const bool valid = pair < p1;
  const uint32_t r = 2 * pair + (t < PB ? 0 : 1);
  const uint8_t *seq = t < PB ? s.m0 : s.m1;
  const int k = d.p.k;
  const uint32_t hb = 2 * (uint32_t)k;
  uint32_t cnt = 0, len = 0;
  if (valid) {
    len = d.rlen[r];
    cnt = cm_minimizers_w7(seq, len, k, [&](uint32_t n, uint64_t h, uint32_t p) {
      const uint64_t v = h | ((uint64_t)p << hb);
      if (n < stg) { if (E6) { sh_lo[(size_t)n * T + t] = (uint32_t)v; sh_hi[(size_t)n * T + t] = (uint16_t)(v >> 32); } else sh_e[(size_t)n * T + t] = v; }
    });
  }
  // block exclusive scan of cnt
  const uint32_t lane = t & 63, wave = t >> 6;
  uint32_t incl = cnt;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const uint32_t v = __shfl_up(incl, dlt, 64);
    if (lane >= (uint32_t)dlt) incl += v;
  }
  if (lane == 63 || t == T - 1) sh_w[wave] = incl;
  __syncthreads();
  const uint32_t n_waves = (T + 63) >> 6;
  if (t == 0) {
    uint32_t tot = 0;
    for (uint32_t wv = 0; wv < n_waves; ++wv) { const uint32_t x = sh_w[wv]; sh_w[wv] = tot; tot += x; }
    const unsigned long long base = tot ? atomicAdd(cursor, (unsigned long long)tot) : 0ull;
    sh_w[8] = (uint32_t)base;
    sh_w[9] = (uint32_t)(base >> 32);
    sh_w[10] = tot;
  }
  __syncthreads();
  const unsigned long long base = (unsigned long long)sh_w[8] | ((unsigned long long)sh_w[9] << 32);
  const uint32_t loc = sh_w[wave] + (incl - cnt);
  const unsigned long long off64 = base + loc;
  const uint64_t hmask = (1ull << hb) - 1;
  if (valid) {
    d.mm_cnt[r] = cnt;
    d.mm_off[r] = (uint32_t)off64;
  }
  if (GSTAGE) {
    sh_off[t] = loc;
    sh_cnt[t] = cnt;
    if (t == T - 1) sh_off[T] = loc + cnt;
    __syncthreads();  // (also: the tile's entries, written by other lanes, are visible)
    const uint32_t tot = sh_w[10];
    for (uint32_t o = t; o < tot; o += T) {
      uint32_t lo = 0, hi = T;  // the last read whose range starts at or before o (reads without minimizers share their neighbour's start)
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sh_off[mid] <= o) lo = mid; else hi = mid;
      }
      if (sh_cnt[lo] > stg || base + o >= mm_cap) continue;  // (a read with more emissions than the tile holds writes its own range below)
      const uint64_t v = sh_e[(size_t)(o - sh_off[lo]) * T + lo];
      d.mm_hash[base + o] = v & hmask;
      d.mm_ps[base + o] = (uint32_t)(v >> hb);
    }
    if (valid && cnt > stg && off64 + cnt <= mm_cap) cm_minimizers_window<7>(seq, len, k, d.mm_hash + (uint32_t)off64, d.mm_ps + (uint32_t)off64, cnt);
    return;
  }
  if (!valid) return;
  if (cnt == 0 || off64 + cnt > mm_cap) return;  // overflow of the dense arrays: the host sees cursor > mm_cap and reruns
  const uint32_t off = (uint32_t)off64;
  if (cnt <= stg) {
    for (uint32_t e = 0; e < cnt; ++e) {
      const uint64_t v = E6 ? ((uint64_t)sh_lo[(size_t)e * T + t] | ((uint64_t)sh_hi[(size_t)e * T + t] << 32)) : sh_e[(size_t)e * T + t];
      d.mm_hash[off + e] = v & hmask;
      d.mm_ps[off + e] = (uint32_t)(v >> hb);
    }
  } else {
    cm_minimizers_window<7>(seq, len, k, d.mm_hash + off, d.mm_ps + off, cnt);
  }
}

// ---------------------------------------------------------------------------------------
// S0 + S1 with one lane per K-MER POSITION (w = 7, odd k <= 26, reads up to 69 bases): k_prep_mm above keeps one lane
// per read -- a sequential 7-entry window per base, lanes idle while shorter (trimmed) reads of their wave are done,
// half the block idle during trimming, byte loads from LDS at a 50-byte stride (bank conflicts 0.32 of the LDS cycles).
// Here a block
//   stages its pairs' bytes in LDS (as before) and packs them to 2 bits per base (16 bases per word, SWAR, one mask
//   of non-ACGT bytes per word);
//   trims (one lane per pair, cm_s0_prep_ptr);
//   lays the k-mer positions of its reads end to end (block scan of len - k + 1) and takes them in tiles of whole
//   reads: every lane extracts the k-mer of one position from the packed words, hashes it (cm_mmf_hash: the three
//   Hash64 of minimizer_generator.cc:47-57), stores the hash in LDS; two sliding extrema over the read's hashes
//   (window minima, then the maximum of the window minima that contain the position) flag the minimizers -- the
//   closed form of cm_minimizers_w7_oddk, cm_stages.h; a block scan of the flags gives every minimizer its slot in
//   the dense arrays (one atomic per tile reserves the range), in read and position order;
//   redoes the few reads the closed form does not cover with the sequential code (non-ACGT bases, fewer than
//   7 k-mers, first-window tie), from global memory.
// ---------------------------------------------------------------------------------------
struct CmFlatRead { uint32_t at; uint16_t cnt; uint8_t fb; uint8_t mate; };

__global__ __launch_bounds__(CM_BLOCK) void k_prep_flat(CmDev d, uint32_t pair_lo, uint32_t n_pairs /* end of this launch's pair range */,
                                                        uint32_t lds_half, uint32_t nt_max, uint32_t tile_reads, uint32_t mm_cap,
                                                        unsigned long long *cursor) {
  constexpr uint32_t T = CM_BLOCK, PB = T / 2;
  const uint32_t p0 = pair_lo + blockIdx.x * PB, p1 = p0 + PB < n_pairs ? p0 + PB : n_pairs;
  const uint32_t t = threadIdx.x, lp = t < PB ? t : t - PB, pair = p0 + lp, mate = t < PB ? 0u : 1u;
  const int k = d.p.k;
  // ---- LDS carve-up
  const uint32_t region0 = 2 * lds_half > 16 * nt_max ? 2 * lds_half : 16 * nt_max;
  const uint32_t nw = lds_half / 16 + 1;
  uint64_t *H = reinterpret_cast<uint64_t *>(cm_lds), *M = H + nt_max;
  uint32_t *pk0 = reinterpret_cast<uint32_t *>(cm_lds + region0), *pk1 = pk0 + nw + 3;
  uint16_t *bd0 = reinterpret_cast<uint16_t *>(pk1 + nw + 3), *bd1 = bd0 + nw + 1;
  CmFlatRead *ri = reinterpret_cast<CmFlatRead *>(reinterpret_cast<uint8_t *>(bd1 + nw + 1) + ((8 - ((uintptr_t)(2 * (nw + 1) * 2)) % 8) % 8));
  uint32_t *pos_off = reinterpret_cast<uint32_t *>(ri + T);
  uint16_t *meta = reinterpret_cast<uint16_t *>(pos_off + T + 1 + 1);
  uint16_t *pref = meta + nt_max + (nt_max & 1);
  uint32_t *misc = reinterpret_cast<uint32_t *>(pref + nt_max + 2 + (nt_max & 1));
  // ---- stage + trim
  const uint64_t g0a = d.ro0[p0], g0b = d.ro0[p1], g1a = d.ro1[p0], g1b = d.ro1[p1];
  const uint64_t a0 = g0a & ~15ull, a1 = g1a & ~15ull;
  uint8_t *l0 = cm_lds, *l1 = cm_lds + lds_half;
  cm_stage_range(l0, d.rb0, a0, g0b);
  cm_stage_range(l1, d.rb1, a1, g1b);
  __syncthreads();
  const bool valid = pair < p1;
  const uint32_t off_m0 = valid ? (uint32_t)(d.ro0[pair] - a0) : 0u, off_m1 = valid ? (uint32_t)(d.ro1[pair] - a1) : 0u;
  // pack: 16 staged bytes -> one word of codes + a mask of bytes that are no base letter
  {
    const uint32_t n0 = (uint32_t)(g0b - a0), n1 = (uint32_t)(g1b - a1);
    const uint32_t w0 = (n0 + 15) / 16, w1 = (n1 + 15) / 16;
    for (uint32_t w = t; w < w0 + w1; w += T) {
      const bool second = w >= w0;
      const uint32_t wi = second ? w - w0 : w;
      const uint4 q = *reinterpret_cast<const uint4 *>((second ? l1 : l0) + 16 * wi);
      uint32_t b0, b1, b2, b3;
      const uint32_t word = cm_mmf_pack4(q.x, &b0) | (cm_mmf_pack4(q.y, &b1) << 8) | (cm_mmf_pack4(q.z, &b2) << 16) | (cm_mmf_pack4(q.w, &b3) << 24);
      (second ? pk1 : pk0)[wi] = word;
      (second ? bd1 : bd0)[wi] = (uint16_t)(b0 | (b1 << 4) | (b2 << 8) | (b3 << 12));
    }
    if (t < 3) { pk0[w0 + t] = 0; pk1[w1 + t] = 0; }
    if (t == 3) { bd0[w0] = 0; bd1[w1] = 0; }
  }
  if (t < PB && valid) cm_s0_prep_ptr(d, pair, l0 + off_m0, l1 + off_m1);
  __syncthreads();
  // ---- one read per lane: k-mer count, or the sequential route
  const uint32_t r = 2 * pair + mate;
  uint32_t len = 0, cnt = 0, fb = 0;
  const uint32_t at = mate ? off_m1 : off_m0;
  if (valid) {
    len = d.rlen[r];
    const uint32_t m = len >= (uint32_t)k ? len - (uint32_t)k + 1 : 0;
    bool bad = false;
    if (m) {
      const uint16_t *bd = mate ? bd1 : bd0;
      const uint32_t wa = at >> 4, wb = (at + len - 1) >> 4;
      for (uint32_t w = wa; w <= wb; ++w) {
        uint32_t mk = bd[w];
        if (w == wa) mk &= 0xFFFFu << (at & 15u);
        if (w == wb) mk &= 0xFFFFu >> (15u - ((at + len - 1) & 15u));
        bad = bad || mk != 0;
      }
    }
    if (m && (bad || m < 7)) fb = 1; else cnt = m;
  }
  ri[t].at = at; ri[t].cnt = (uint16_t)cnt; ri[t].fb = (uint8_t)fb; ri[t].mate = (uint8_t)mate;
  {  // block exclusive scan of cnt -> pos_off[0..T]
    const uint32_t lane = t & 63, wave = t >> 6;
    uint32_t incl = cnt;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint32_t v = __shfl_up(incl, dlt, 64);
      if (lane >= (uint32_t)dlt) incl += v;
    }
    if (lane == 63) misc[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t q = 0; q < wave; ++q) wbase += misc[q];
    pos_off[t] = wbase + incl - cnt;
    if (t == T - 1) pos_off[T] = wbase + incl;
  }
  __syncthreads();
  // ---- tiles of whole reads
  for (uint32_t r0 = 0; r0 < T; r0 += tile_reads) {
    const uint32_t r1 = r0 + tile_reads < T ? r0 + tile_reads : T;
    const uint32_t P0 = pos_off[r0], NT = pos_off[r1] - P0;
    const float per_pos = NT ? (float)(r1 - r0) / (float)NT : 0.0f;  // reads per position: first guess of a position's read
    // (a) hash of every position
    for (uint32_t p = t; p < NT; p += T) {
      uint32_t lr = r0 + (uint32_t)((float)p * per_pos);
      lr = lr < r1 ? lr : r1 - 1;
      while (pos_off[lr + 1] - P0 <= p) ++lr;
      while (pos_off[lr] - P0 > p) --lr;
      const uint32_t i = p - (pos_off[lr] - P0);
      uint32_t sd;
      H[p] = cm_mmf_hash(cm_mmf_kmer(ri[lr].mate ? pk1 : pk0, ri[lr].at + i, k), k, &sd);
      meta[p] = (uint16_t)(((lr - r0) << 7) | (i << 1) | sd);  // read in tile (<= 127) | k-mer index (<= 63) | strand
    }
    __syncthreads();
    // (b) window minima; the first window also decides whether the read needs the sequential route
    for (uint32_t p = t; p < NT; p += T) {
      const uint32_t mt = meta[p], lr = r0 + (mt >> 7), i = (mt >> 1) & 63u, m = ri[lr].cnt;
      if (i + 7 <= m) {
        uint64_t x = H[p];
#pragma unroll
        for (int q = 1; q < 6; ++q) x = H[p + q] < x ? H[p + q] : x;
        const uint64_t h6 = H[p + 6];
        if (i == 0 && h6 == x) ri[lr].fb = 2;
        M[p] = h6 < x ? h6 : x;
      }
    }
    __syncthreads();
    // (c) minimizer <=> own hash equals the largest minimum of the windows that hold it
    for (uint32_t p = t; p < NT; p += T) {
      const uint32_t mt = meta[p], lr = r0 + (mt >> 7), i = (mt >> 1) & 63u, m = ri[lr].cnt;
      const uint32_t j0 = i >= 6 ? i - 6 : 0, j1 = i + 7 <= m ? i : m - 7, b = p - i;
      uint64_t x = 0;
      for (uint32_t j = j0; j <= j1; ++j) x = M[b + j] > x ? M[b + j] : x;
      const bool fl = x == H[p] && ri[lr].fb == 0;
      meta[p] = (uint16_t)(mt | (fl ? 0x8000u : 0u));
    }
    __syncthreads();
    // (d) slots: exclusive scan of the flags in position order, one reservation per tile
    {
      const uint32_t chunk = (NT + T - 1) / T, c0 = t * chunk < NT ? t * chunk : NT, c1 = c0 + chunk < NT ? c0 + chunk : NT;
      uint32_t sum = 0;
      for (uint32_t p = c0; p < c1; ++p) sum += meta[p] >> 15;
      const uint32_t lane = t & 63, wave = t >> 6;
      uint32_t incl = sum;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const uint32_t v = __shfl_up(incl, dlt, 64);
        if (lane >= (uint32_t)dlt) incl += v;
      }
      if (lane == 63) misc[8 + wave] = incl;
      __syncthreads();
      uint32_t run = incl - sum;
      for (uint32_t q = 0; q < wave; ++q) run += misc[8 + q];
      for (uint32_t p = c0; p < c1; ++p) { pref[p] = (uint16_t)run; run += meta[p] >> 15; }
      if (t == T - 1) {
        pref[NT] = (uint16_t)run;
        const unsigned long long base = run ? atomicAdd(cursor, (unsigned long long)run) : 0ull;
        misc[16] = (uint32_t)base; misc[17] = (uint32_t)(base >> 32);
      }
    }
    __syncthreads();
    const unsigned long long base = (unsigned long long)misc[16] | ((unsigned long long)misc[17] << 32);
    for (uint32_t p = t; p < NT; p += T) {
      const uint32_t mt = meta[p];
      if (mt >> 15) {
        const unsigned long long o = base + pref[p];
        if (o < mm_cap) { d.mm_hash[o] = H[p]; d.mm_ps[o] = ((((mt >> 1) & 63u) + (uint32_t)k - 1u) << 1) | (mt & 1u); }
      }
    }
    if (t >= r0 && t < r1 && valid && ri[t].fb == 0) {  // this lane's read lies in the tile
      const uint32_t a = pos_off[t] - P0, b = pos_off[t + 1] - P0;
      d.mm_cnt[r] = (uint32_t)pref[b] - (uint32_t)pref[a];
      d.mm_off[r] = (uint32_t)(base + pref[a]);
    }
    __syncthreads();
  }
  // ---- the reads the closed form does not cover: sequentially, from global memory (the staged bytes are gone)
  if (valid && ri[t].fb) {
    const uint8_t *seq = cm_read_ptr(d, r);
    const uint32_t c = cm_minimizers_window<7>(seq, len, k, nullptr, nullptr, 0);
    const unsigned long long off = c ? atomicAdd(cursor, (unsigned long long)c) : 0ull;
    d.mm_cnt[r] = c;
    d.mm_off[r] = (uint32_t)off;
    if (c && off + c <= mm_cap) cm_minimizers_window<7>(seq, len, k, d.mm_hash + off, d.mm_ps + off, c);
  }
}

// S3a: hit counts per read; reads whose hit list will not fit a lane's LDS slots in k_s3b_candidates are listed by
// size class for the cooperative kernel below
__global__ __launch_bounds__(CM_BLOCK) void k_s3a_count(CmDev d, uint32_t n) {
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  uint32_t tot = 0;
  if (i < n) {
    cm_s3a_count(d, i);
    tot = d.hit_tot[i];
  }
  // one atomic per BLOCK and class: millions of lanes adding to one counter serialise at ~10 ns each, and so do the waves of a
  // batch from a repeat-bearing genome, where nearly every wave holds a read of some class (k_s3a_count 1.4 ms against 0.45 ms
  // on the uniform genome with one atomic per wave).  The order inside a list is of no consequence.
  // (class 21: the lists of the wave class that fit a quarter of its work area -- four reads per CU where the full-size area has one)
  const uint32_t cls = tot <= d.s3b_cap ? 5u /* (none: the lane's own slots) */ : tot <= d.hv_mid ? CM_L_HIT_G16 : tot <= d.hv_sub ? CM_L_HIT_WAVE_SMALL : tot <= d.hv_max[0] ? CM_L_HIT_WAVE : tot <= d.hv_max[1] ? CM_L_HIT_B256A
                       : tot <= d.hv_max[2] ? CM_L_HIT_B256B : tot <= d.hv_max[3] ? CM_L_HIT_B512 : tot <= d.hv_big ? CM_L_HIT_B1024 : CM_L_HIT_SLAB;
  __shared__ uint32_t sh_cnt[8], sh_base[8];
  if (threadIdx.x < 8) sh_cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t ids[8] = {CM_L_HIT_WAVE, CM_L_HIT_B256A, CM_L_HIT_B256B, CM_L_HIT_SLAB, CM_L_HIT_G16, CM_L_HIT_B512, CM_L_HIT_WAVE_SMALL, CM_L_HIT_B1024};
  uint32_t slot = 0, mine = 8;
#pragma unroll
  for (uint32_t q = 0; q < 8; ++q) {
    const unsigned long long m = __ballot(cls == ids[q]);
    if (m == 0) continue;
    uint32_t base = 0;
    if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(&sh_cnt[q], (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1, 64);
    if (cls == ids[q]) { slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); mine = q; }
  }
  __syncthreads();
  if (threadIdx.x < 8 && sh_cnt[threadIdx.x]) sh_base[threadIdx.x] = atomicAdd(&d.hv_cnt[ids[threadIdx.x]], sh_cnt[threadIdx.x]);
  __syncthreads();
  if (mine < 8) d.hv_list[(size_t)ids[mine] * d.hv_stride + sh_base[mine] + slot] = i;
}
// S3b with the per-read hit list staged in LDS ([entry][thread] layout: 16 x 8-byte entries and
// 16 count bytes per thread = 36 KB per block).  The first version sorted every list in its
// global segment; per-thread read-modify-write of small segments thrashed L2 (rocprofv3: 10 GB
// of HBM traffic per launch for 0.55 GB of hits).
// Geometry from the longest read of the batch: cap (hits per lane staged in LDS) and threads per block are
// chosen so that cap * threads * 9 bytes stays at 36 KB -- 16 x 256 for 50-base reads (7.7 minimizers, ~9 hits),
// 48 x 64 for 150-base reads (~34 minimizers); longer lists go to k_s3b_heavy.
__global__ __launch_bounds__(CM_BLOCK) void k_s3b_candidates(CmDev d, uint32_t n, uint32_t cap) {
  uint64_t *sh_h = reinterpret_cast<uint64_t *>(cm_lds);
  uint8_t *sh_c = cm_lds + (size_t)cap * blockDim.x * 8;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && d.hit_tot[i] <= cap) cm_s3b_candidates_lds(d, i, sh_h + threadIdx.x, sh_c + threadIdx.x, cap, blockDim.x);
}

// ---------------------------------------------------------------------------------------
// S3b for long hit lists (reads from repeats: hundreds to thousands of hits).  One lane sorting such a list
// in its global segment took ~28 ms and held its whole wave for that long (a genome with 2 % of its bases in
// 600-copy repeat families: 452 ms per 4 M-pair batch in this stage).  Here a GROUP of lanes -- a wave for
// lists up to 1024 hits, a block beyond -- works on one read:
//   expand   the occurrence runs of the read's minimizers are copied by the group's lanes into LDS (slots handed
//            out by an LDS counter; the order is irrelevant), strand in bit 63 of the key so that one sort yields
//            the + list followed by the - list (a sequence id needs bit 31 free: checked on the host);
//   sort     bitonic network in LDS over the next power of two (padding = all ones);
//   cluster  the sweep is sequential only inside a LOCAL cluster (cm_sweep_cluster, cm_stages.h): every lane
//            takes the local clusters whose first hit it owns, counts their candidates, the group scans the
//            counts, the lanes sweep again and write the candidates to the read's global segment in list order.
// Results are those of cm_s3b_core, element for element.
// ---------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ void cm_group_sync() {
  if (G <= 64) {
    // a wave or a part of one: its LDS operations execute in order; make them complete and keep the compiler from moving accesses across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else {
    __syncthreads();
  }
}

// n_list_dev (or nullptr): the list's length when only the device knows it (reads the merge-sort kernel declined); the
// grid then strides over the list
template <int G>
__global__ __launch_bounds__(CM_BLOCK) void k_s3b_heavy(CmDev d, const uint32_t *__restrict__ list, uint32_t n_list, uint32_t P,
                                                        const uint32_t *__restrict__ n_list_dev) {
  constexpr int GPB = CM_BLOCK / G;  // groups per block
  const uint32_t grp = threadIdx.x / G, t = threadIdx.x % G;
  if (n_list_dev) n_list = *n_list_dev;
  uint64_t *S = reinterpret_cast<uint64_t *>(cm_lds) + (size_t)grp * P;
  uint16_t *oc = reinterpret_cast<uint16_t *>(cm_lds + (size_t)GPB * P * 8) + (size_t)grp * P;
  uint32_t *ctr = reinterpret_cast<uint32_t *>(cm_lds + (size_t)GPB * P * 10) + (size_t)grp * (G + 8);
  for (uint32_t gid = blockIdx.x * GPB + grp; gid < n_list; gid += gridDim.x * GPB) {  // whole group (a wave, or the block)
  const uint32_t r = list[gid];
  const uint32_t tot = d.hit_tot[r];
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  const uint32_t maxf = d.round2[r] ? (uint32_t)d.p.f1 : (uint32_t)d.p.f0;
  const uint64_t SB = 1ull << 63;
  uint32_t P2 = 2;
  while (P2 < tot) P2 <<= 1;
  if (t == 0) { ctr[0] = 0; ctr[1] = 0; }
  for (uint32_t i = tot + t; i < P2; i += G) S[i] = ~0ull;
  cm_group_sync<G>();
  // ---- expand
  uint32_t my_pos = 0;
  if (G < 64) {
    // short lists of many minimizers (2 x 150 reads: ~37 hits of ~34 minimizers): a lane per minimizer, its few occurrences too
    for (uint32_t mi = t; mi < n; mi += G) {
      const uint8_t kind = d.pr_kind[b + mi];
      if (kind == CM_PR_MISS) continue;
      const uint64_t val = d.pr_val[b + mi];
      const uint32_t ps = d.mm_ps[b + mi];
      bool same;
      if (kind == CM_PR_SINGLE) {
        const uint64_t cp = cm_cand_from_hit(val, ps, d.p.k, &same);
        S[atomicAdd(&ctr[0], 1u)] = same ? cp : (cp | SB);
        my_pos += same ? 1u : 0u;
        continue;
      }
      const uint32_t nocc = (uint32_t)val;
      if (nocc >= maxf) continue;
      const uint64_t *o = d.occ + (uint32_t)(val >> 32);
      for (uint32_t oi = 0; oi < nocc; ++oi) {
        const uint64_t cp = cm_cand_from_hit(o[oi], ps, d.p.k, &same);
        S[atomicAdd(&ctr[0], 1u)] = same ? cp : (cp | SB);
        my_pos += same ? 1u : 0u;
      }
    }
  } else
  for (uint32_t mi = 0; mi < n; ++mi) {
    const uint8_t kind = d.pr_kind[b + mi];
    if (kind == CM_PR_MISS) continue;
    const uint64_t val = d.pr_val[b + mi];
    const uint32_t ps = d.mm_ps[b + mi];
    bool same;
    if (kind == CM_PR_SINGLE) {
      if (t == 0) {
        const uint64_t cp = cm_cand_from_hit(val, ps, d.p.k, &same);
        S[atomicAdd(&ctr[0], 1u)] = same ? cp : (cp | SB);
        my_pos += same ? 1u : 0u;
      }
      continue;
    }
    const uint32_t nocc = (uint32_t)val;
    if (nocc >= maxf) continue;
    const uint64_t *o = d.occ + (uint32_t)(val >> 32);
    for (uint32_t oi = t; oi < nocc; oi += G) {
      const uint64_t cp = cm_cand_from_hit(o[oi], ps, d.p.k, &same);
      S[atomicAdd(&ctr[0], 1u)] = same ? cp : (cp | SB);
      my_pos += same ? 1u : 0u;
    }
  }
  if (my_pos) atomicAdd(&ctr[1], my_pos);
  cm_group_sync<G>();
  const uint32_t np = ctr[1], nn = tot - np;
  // ---- sort (ascending; keys with bit 63 -- the - strand -- follow the + strand's)
  for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1) {
    for (uint32_t j = k2 >> 1, lj = 31u - (uint32_t)__clz(k2 >> 1); j > 0; j >>= 1, --lj) {
      for (uint32_t i = t; i < (P2 >> 1); i += G) {
        const uint32_t a = ((i >> lj) << (lj + 1)) | (i & (j - 1)), c = a + j;  // j = 1 << lj
        const uint64_t x = S[a], y = S[c];
        const bool up = (a & k2) == 0;
        if ((x > y) == up) { S[a] = y; S[c] = x; }
      }
      cm_group_sync<G>();
    }
  }
  // ---- cluster: count pass
  const bool use_high = d.round2[r] && np > 0 && nn > 0;
  int req = (int)n - (int)d.rep_cnt[r];
  req = req > 1 ? req : 1;
  req = req > d.p.min_seeds ? d.p.min_seeds : req;
  if (use_high) req = d.p.min_seeds;
  const int e = d.p.e;
  for (uint32_t i = t; i < tot; i += G) {
    uint32_t c = 0;
    if (i == 0 || cm_sweep_local_break(S[i - 1], S[i], e)) {
      uint32_t end = i + 1;
      while (end < tot && !cm_sweep_local_break(S[end - 1], S[end], e)) ++end;
      c = cm_sweep_cluster(S, 1, i, end, e, req, n, nullptr, nullptr);
    }
    oc[i] = (uint16_t)c;
  }
  cm_group_sync<G>();
  // exclusive scan of oc[0..tot): per-lane chunk sums, lane 0 scans the G sums, lanes rewrite their chunks
  const uint32_t chunk = (tot + G - 1) / G, c0 = t * chunk, c1 = c0 + chunk < tot ? c0 + chunk : tot;
  {
    uint32_t sum = 0;
    for (uint32_t i = c0; i < c1; ++i) sum += oc[i];
    ctr[8 + t] = sum;
  }
  cm_group_sync<G>();
  if (t == 0) {
    uint32_t run = 0;
    for (uint32_t q = 0; q < (uint32_t)G; ++q) { const uint32_t x = ctr[8 + q]; ctr[8 + q] = run; run += x; }
    ctr[2] = run;  // all candidates
  }
  cm_group_sync<G>();
  {
    uint32_t run = ctr[8 + t];
    for (uint32_t i = c0; i < c1; ++i) { const uint32_t x = oc[i]; oc[i] = (uint16_t)run; run += x; }
  }
  cm_group_sync<G>();
  const uint32_t total = ctr[2];
  const uint32_t ncp = np < tot ? (np > 0 ? oc[np] : 0u) : total, ncn = total - ncp;
  // ---- cluster: write pass (candidates of the + list at h[0..ncp), of the - list at h[np..np+ncn))
  uint64_t *h = d.hbuf + d.hit_off[r];
  uint8_t *hc = d.hcnt + d.hit_off[r];
  for (uint32_t i = t; i < tot; i += G) {
    if (i == 0 || cm_sweep_local_break(S[i - 1], S[i], e)) {
      uint32_t end = i + 1;
      while (end < tot && !cm_sweep_local_break(S[end - 1], S[end], e)) ++end;
      const uint32_t off = oc[i];
      const uint32_t at = i < np ? off : np + (off - ncp);
      cm_sweep_cluster(S, 1, i, end, e, req, n, h + at, hc + at, ~SB);
    }
  }
  if (t == 0) { d.n_pos_hit[r] = np; d.ncp[r] = ncp; d.ncn[r] = ncn; }
  cm_group_sync<G>();  // the next read of this group reuses S / oc / ctr
  }
}

// ---------------------------------------------------------------------------------------
// The group type of cm_coop.h on the device: G = 16 (a quarter wave), 64 (a wave) or CM_BLOCK lanes.  rank() is a
// ballot, scan() a shuffle scan inside the wave part plus, for a block, the wave totals through LDS (xw).
// ---------------------------------------------------------------------------------------
// Waves per SIMD the register allocation of k_s3b_coop aims at (second launch-bounds argument): at 76-85 registers a lane only five waves
// fit a SIMD, i.e. two blocks of 512 lanes per CU whatever shared memory allows; 64 registers (6-13 of them spilled) bought 8 % of the
// kernel's time.  Measured and NOT kept (round 5): the same for k_s5c_coop (146 -> 80 registers: 30 % slower), k_s4c_coop, k_s4b_coop
// (slower), k_s4a/4b_rescue_wave (119 -> 80: no change -- beyond 16 waves per CU its searches are not short of waves)
#define CM_WPE_S3B 8
template <int G_>
struct CmDevGroup {
  static constexpr int G = G_;
  static constexpr int W = G_ < 64 ? G_ : 64;
  uint32_t t;
  uint32_t *xw;  // LDS, 256 bytes of this group: the wave parts' totals (unused when G <= 64)
  __device__ __forceinline__ void sync() { cm_group_sync<G_>(); }
  __device__ __forceinline__ unsigned long long ballot(bool p) {  // of the wave part
    unsigned long long m = __ballot(p);
    if (W < 64) m = (m >> ((threadIdx.x & 63u) / W * W)) & ((1ull << W) - 1ull);
    return m;
  }
  __device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t lane) { return __shfl(v, (int)lane, W); }  // lane: the same for the wave part
  __device__ __forceinline__ uint32_t rank(bool p, uint32_t *total) {
    unsigned long long m = __ballot(p);
    if (W < 64) m = (m >> ((threadIdx.x & 63u) / W * W)) & ((1ull << W) - 1ull);
    *total = (uint32_t)__popcll(m);
    return (uint32_t)__popcll(m & ((1ull << (t % W)) - 1ull));
  }
  __device__ __forceinline__ uint32_t scan(uint32_t v, uint32_t *total) {
    const uint32_t wl = t % W;
    uint32_t incl = v;
#pragma unroll
    for (int dlt = 1; dlt < W; dlt <<= 1) {
      const uint32_t x = __shfl_up(incl, dlt, W);
      if (wl >= (uint32_t)dlt) incl += x;
    }
    if (G <= 64) {
      *total = __shfl(incl, W - 1, W);
      return incl - v;
    }
    const uint32_t wv = t / W;
    if (wl == W - 1) xw[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < G / W; ++q) { const uint32_t x = xw[q]; base += (uint32_t)q < wv ? x : 0u; tot += x; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
  }
  __device__ __forceinline__ uint32_t scanmax(uint32_t v, uint32_t *total) {
    const uint32_t wl = t % W;
    uint32_t incl = v;
#pragma unroll
    for (int dlt = 1; dlt < W; dlt <<= 1) {
      const uint32_t x = __shfl_up(incl, dlt, W);
      if (wl >= (uint32_t)dlt && x > incl) incl = x;
    }
    uint32_t excl = __shfl_up(incl, 1, W);
    if (wl == 0) excl = 0;
    if (G <= 64) {
      *total = __shfl(incl, W - 1, W);
      return excl;
    }
    const uint32_t wv = t / W;
    if (wl == W - 1) xw[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < G / W; ++q) { const uint32_t x = xw[q]; if ((uint32_t)q < wv && x > base) base = x; if (x > tot) tot = x; }
    __syncthreads();
    *total = tot;
    return excl > base ? excl : base;
  }
  __device__ __forceinline__ uint64_t max64(uint64_t v) {
#pragma unroll
    for (int dlt = W / 2; dlt > 0; dlt >>= 1) {
      const uint64_t x = __shfl_xor(v, dlt, W);
      v = x > v ? x : v;
    }
    if (G <= 64) return v;
    uint64_t *xq = reinterpret_cast<uint64_t *>(xw);
    if (t % W == 0) xq[t / W] = v;
    __syncthreads();
    uint64_t mx = 0;
#pragma unroll
    for (int q = 0; q < G / W; ++q) mx = xq[q] > mx ? xq[q] : mx;
    __syncthreads();
    return mx;
  }
  __device__ __forceinline__ uint64_t min64(uint64_t v) { return ~max64(~v); }
  __device__ __forceinline__ uint32_t sum(uint32_t v) { uint32_t tot; (void)scan(v, &tot); return tot; }
};
// per-group LDS: the cooperative work area, then the group's cross-wave words
#define CM_XW_BYTES 256
__host__ __device__ inline size_t cm_coop_group_bytes(uint32_t P, uint32_t MM, uint32_t RB, bool own_oc, bool key32 = false) { return ((cm_coop_mem_bytes(P, MM, RB, own_oc, key32) + 15) & ~(size_t)15) + CM_XW_BYTES; }

// S3b for long hit lists, merge-sort form (cm_coop_s3b): blockDim.x / G groups per block, one listed read each.  What the
// function declines (more occurrence runs than its tables hold) is appended to fb_list for the bitonic kernel above.
// use_slab: the launch has at most d.coop_slab_blocks blocks of one group each, block b works on slab b (lists longer than P)
// K32: 32-bit hit keys (global coordinates, d.goff), 11 instead of 19 bytes of shared memory per hit -- not with use_slab
template <int G, bool K32>
__global__ __launch_bounds__(G < CM_BLOCK ? CM_BLOCK : G, CM_WPE_S3B) void k_s3b_coop(CmDev d, const uint32_t *__restrict__ list, uint32_t n_list, uint32_t P, uint32_t MM, uint32_t RB,
                                                                      uint32_t *__restrict__ fb_list, uint32_t *__restrict__ fb_cnt, uint32_t use_slab) {
  const uint32_t gpb = blockDim.x / G, grp = threadIdx.x / G;
  const size_t gb = cm_coop_group_bytes(P, MM, RB, false, K32);
  uint8_t *base = cm_lds + (size_t)grp * gb;
  CmCoopMem m = cm_coop_mem_at(base, P, MM, RB, false, K32);
  if (use_slab) cm_coop_slab_at(m, d.coop_slab + (size_t)blockIdx.x * cm_coop_slab_bytes(d.coop_slab_cap), d.coop_slab_cap);
  CmDevGroup<G> g;
  g.t = threadIdx.x % G;
  g.xw = reinterpret_cast<uint32_t *>(base + gb - CM_XW_BYTES);
  for (uint32_t gid = blockIdx.x * gpb + grp; gid < n_list; gid += gridDim.x * gpb) {  // a whole group
    const uint32_t r = list[gid];
    bool done;
    if constexpr (K32) done = cm_coop_s3b_k32(d, r, g, m);
    else done = use_slab ? cm_coop_s3b<true>(d, r, g, m) : cm_coop_s3b<false>(d, r, g, m);
    if (!done && g.t == 0) fb_list[atomicAdd(fb_cnt, 1u)] = r;
    g.sync();  // the work area is reused
  }
}

// lists too long for the groups' LDS: one lane each, in the read's global segment (rare: > 8192 hits)
__global__ __launch_bounds__(64) void k_s3b_serial(CmDev d, const uint32_t *__restrict__ list, uint32_t n_list, const uint32_t *__restrict__ n_list_dev) {
  if (n_list_dev) n_list = *n_list_dev;
  for (uint32_t i = blockIdx.x * 64 + threadIdx.x; i < n_list; i += gridDim.x * 64) cm_s3b_candidates(d, list[i]);
}

// every lane with `pred` appends `value` to a device list: one atomic per wave (called by all lanes of the wave)
__device__ __forceinline__ void cm_wave_append(uint32_t *__restrict__ list, uint32_t *__restrict__ counter, bool pred, uint32_t value) {
  const unsigned long long m = __ballot(pred);
  if (m == 0) return;
  const uint32_t lane = threadIdx.x & 63u, leader = (uint32_t)(__ffsll((long long)m) - 1);
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, (int)leader, 64);
  if (pred) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// Dynamic distribution of a device work list over the waves of a launch (round 6; cmgpu_set_option "coop" bit 16, NOT the default).
// The list kernels stride (item j to wave j % waves): with items whose cost spans two orders of magnitude (a rescue search over 4 or over
// 300 windows) the waves' lifetimes average 43 % of the launch's duration (SQ_WAVE_CYCLES / waves against the dispatch's duration,
// profile 2: k_s4a_rescue_wave<false> 4.5 of 10.5 ms) -- the launch waits for its unluckiest wave.  With the option a wave takes
// CM_WQ_CHUNK consecutive items with ONE atomic on the list's cursor and comes back for more.  Measured (gpurun_out/r06g, A/B on one
// box): one lane, profile 2, S4a 45.5 -> 38.8 ms -- but with three lanes, where another lane's kernels fill the tail anyway, profile 2
// 20.85 -> 21.1 M pairs/s and the planted-repeat workload 156.5 -> 147.8: same-address atomics retire at ~90 per microsecond, and a
// list of 200 k cheap items is 50 k atomics = 0.55 ms on a kernel of 1 ms.  Kept as an option for single-lane callers.
// The cursors are words 32.. of hv_cnt (zeroed with the lists' counters before S3a); every launch that drains a list has its own.
#define CM_WQ_CHUNK 4u
enum { CM_WQ_S4A_SMALL = 32, CM_WQ_S4A_BIG, CM_WQ_S4B_SMALL, CM_WQ_S4B_BIG, CM_WQ_S5_SORT_SMALL, CM_WQ_S5_SORT_WAVE, CM_WQ_S5_SORT_BLOCK };
struct CmWaveQueue {
  uint32_t *cur;
  uint32_t cnt, j, jend, stride;
  // dynamic == 0 (the default, CmDev::wq_dynamic): item j to wave j % waves
  __device__ __forceinline__ CmWaveQueue(uint32_t *cursor, uint32_t n, uint32_t dynamic = 0) : cur(cursor), cnt(n), j(0), jend(0), stride(0) {
    if (!dynamic) {
      const uint32_t wpb = blockDim.x >> 6;
      stride = gridDim.x * wpb;
      j = blockIdx.x * wpb + (threadIdx.x >> 6);
      jend = 0xffffffffu;
    }
  }
  // the next item's index in the list (all lanes of the wave get it), false when the list is drained
  __device__ __forceinline__ bool next(uint32_t *item) {
    if (stride) {
      if (j >= cnt) return false;
      *item = j;
      j += stride;
      return true;
    }
    if (j == jend) {
      uint32_t b = 0;
      if ((threadIdx.x & 63u) == 0) b = atomicAdd(cur, CM_WQ_CHUNK);
      b = __shfl(b, 0, 64);
      if (b >= cnt) return false;
      j = b;
      jend = b + CM_WQ_CHUNK < cnt ? b + CM_WQ_CHUNK : cnt;
    }
    *item = j++;
    return true;
  }
};

// A read whose mate has CM_RS_WAVE candidates or more on a strand: a WAVE per read (cm_coop_rescue: the windows of the mate's best
// candidates once per direction, a lane per (minimizer, window) pair for the bounds, the search chain replayed on indices) -- with a
// lane per minimizer a search over ~300 windows took ~8 ms, the duration of the whole list kernel on the mosaic genome.
#define CM_RS_WAVE 4u  // (round 5: 24 -> 4.  The wave's searches leave their hits in the pool, so its fill pass copies where the list kernels' lanes and
                       //  16-lane groups search a second time: profile 1 51.2 -> 53.1, repeat workload 145.8 -> 148.9 M pairs/s; 8 and 4 measure alike)
__device__ __forceinline__ bool cm_rescue_is_wave(const CmDev &d, uint32_t r, uint32_t coop) {
  const uint32_t o = r ^ 1u;
  return coop && (d.ncp[o] >= CM_RS_WAVE || d.ncn[o] >= CM_RS_WAVE);
}
// S4a / S4b: few reads (pairs whose mate has to be rescued, ~7 % here) run the long occurrence-run
// searches.  Spread over all waves they keep every wave busy for one search's latency at 4-5 active lanes;
// packed per block (r01h) one wave per block still ran them at a quarter of its lanes.  Every lane now does the
// cheap part of its own read and the reads with a search go to ONE device list (a wave-aggregated append),
// which k_s4a_rescue_list / k_s4b_rescue_list work through with full waves.
// The list is kept in CM_RS_SEGS segments, each with a counter on a cache line of its own: same-address device atomics
// retire at ~90 per microsecond (one per wave of an 8 M-read batch on ONE counter measured 1.4 ms); a block appends with
// one atomic to segment blockIdx % CM_RS_SEGS.  Segment g starts at rs_list + g * seg_cap.
__global__ __launch_bounds__(CM_BLOCK) void k_s4a_rescue_count(CmDev d, uint32_t n, uint32_t seg_cap, uint32_t coop) {
  __shared__ uint32_t sh_cnt, sh_base;
  if (threadIdx.x == 0) sh_cnt = 0;
  __syncthreads();
  const uint32_t i0 = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t i = i0 < n && d.perm_reads ? d.perm_reads[i0] : i0;
  const bool aug0 = i0 < n && cm_s4a_decide(d, i);
  // a read whose mate has many candidates goes to list 23 instead: a wave each (k_s4a_rescue_wave / k_s4b_rescue_wave)
  const bool wv = aug0 && cm_rescue_is_wave(d, i, coop);
  cm_wave_append(d.hv_list + (size_t)CM_L_SEARCH_WAVE * d.hv_stride, d.hv_cnt + CM_L_SEARCH_WAVE, wv, i);
  const bool aug = aug0 && !wv;
  // slot inside the block: wave-aggregated LDS atomic
  const unsigned long long m = __ballot(aug);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t wbase = 0;
  if (m) {
    const uint32_t leader = (uint32_t)(__ffsll((long long)m) - 1);
    if (lane == leader) wbase = atomicAdd(&sh_cnt, (uint32_t)__popcll(m));
    wbase = __shfl(wbase, (int)leader, 64);
  }
  __syncthreads();
  const uint32_t seg = blockIdx.x % CM_RS_SEGS;
  if (threadIdx.x == 0 && sh_cnt) sh_base = atomicAdd(&d.rs_cnt[seg * 16], sh_cnt);
  __syncthreads();
  if (aug) d.rs_list[(uint64_t)seg * seg_cap + sh_base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}
// the searches of the listed reads: blockIdx.y = segment, the x blocks stride over it (its length is only known on the device)
// A read whose mate has many candidates (a read pair from a repeat family) walks hundreds of windows per minimizer: one lane
// doing that kept its list kernel alive for milliseconds.  Such reads (CM_RS_HEAVY mate candidates or more on a strand) are left
// to the second half of the kernel, where a GROUP of 16 lanes takes one read, a lane per minimizer (cm_rescue_minimizer: the
// minimizers are independent, the windows inside one are not).
#define CM_RS_HEAVY 8u
#define CM_RS_G 16
#define CM_RS_MAXMM 192  // minimizers of a read the group path holds counts for in LDS (longer reads: the one-lane path)
__device__ __forceinline__ bool cm_rescue_is_heavy(const CmDev &d, uint32_t r, uint32_t coop) {
  const uint32_t o = r ^ 1u;
  return (d.ncp[o] >= CM_RS_HEAVY || d.ncn[o] >= CM_RS_HEAVY) && d.mm_cnt[r] <= CM_RS_MAXMM && !cm_rescue_is_wave(d, r, coop);
}
// best count among the mate candidates and how many have it, over the group's lanes (all lanes return the same)
__device__ __forceinline__ void cm_group_best(const uint8_t *mc, uint32_t mn, uint32_t t, int *max_count, int *best_num) {
  int lmax = 0, lnum = 0;
  for (uint32_t i = t; i < mn; i += CM_RS_G) {
    const int c = mc[i];
    if (c > lmax) { lmax = c; lnum = 1; } else if (c == lmax) ++lnum;
  }
  for (int off = CM_RS_G / 2; off > 0; off >>= 1) {
    const int om = __shfl_xor(lmax, off, CM_RS_G), on = __shfl_xor(lnum, off, CM_RS_G);
    if (om > lmax) { lmax = om; lnum = on; } else if (om == lmax) lnum += on;
  }
  *max_count = lmax;
  *best_num = lnum;
}
// counting pass of one direction by a group; returns max_count or its negation (bail-out); *cnt = hits, *rl = repetitive length
__device__ __forceinline__ int cm_group_rescue_count(const CmDev &d, uint32_t r, int strand, const uint64_t *mp, const uint8_t *mc, uint32_t mn,
                                                     uint32_t t, uint32_t *cnt, uint32_t *rl) {
  int max_count, best_num;
  cm_group_best(mc, mn, t, &max_count, &best_num);
  *cnt = 0;
  if (cm_rescue_bails(d, max_count, best_num, mn)) {
    if (d.prof && t == 0) atomicAdd(&d.prof[26], 1ull);
    return -max_count;
  }
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  if (d.prof && t == 0) {  // measurement aid (tools/coop_profile.py): what the heavy rescue searches look like
    unsigned long long so = 0;
    for (uint32_t mi = 0; mi < n; ++mi) if (d.pr_kind[b + mi] == CM_PR_MULTI) so += (uint32_t)d.pr_val[b + mi];
    atomicAdd(&d.prof[16], 1ull); atomicAdd(&d.prof[17], (unsigned long long)best_num); atomicAdd(&d.prof[18], (unsigned long long)mn);
    atomicAdd(&d.prof[19], so); atomicAdd(&d.prof[20], (unsigned long long)n);
    atomicAdd(&d.prof[best_num < 4 ? 22 : best_num < 16 ? 23 : best_num < 64 ? 24 : 25], 1ull);
  }
  uint32_t lc = 0;
  for (uint32_t mi = t; mi < n; mi += CM_RS_G)
    lc += cm_rescue_minimizer(d, strand, mp, mc, mn, max_count, d.pr_kind[b + mi], d.pr_val[b + mi], d.mm_ps[b + mi], nullptr, nullptr);
  for (int off = CM_RS_G / 2; off > 0; off >>= 1) lc += __shfl_xor(lc, off, CM_RS_G);
  uint32_t rep_len = 0;
  if (t == 0) {
    uint32_t prev_rep = ~0u;
    for (uint32_t mi = 0; mi < n; ++mi) cm_rescue_rep(d, d.pr_kind[b + mi], d.pr_val[b + mi], d.mm_ps[b + mi], &rep_len, &prev_rep);
  }
  *rl = __shfl(rep_len, 0, CM_RS_G);
  *cnt = lc;
  if (d.prof && t == 0) atomicAdd(&d.prof[21], (unsigned long long)lc);
  return max_count;
}
__global__ __launch_bounds__(64) void k_s4a_rescue_list(CmDev d, uint32_t seg_cap, uint32_t coop) {
  const uint32_t cnt = d.rs_cnt[blockIdx.y * 16];
  const uint32_t *list = d.rs_list + (uint64_t)blockIdx.y * seg_cap;
  // reads with few mate candidates: a lane each
  const long long t1 = d.prof ? clock64() : 0;
  for (uint32_t j = blockIdx.x * 64 + threadIdx.x; j < cnt; j += gridDim.x * 64) {
    const uint32_t r = list[j];
    if (!cm_rescue_is_heavy(d, r, coop)) cm_s4a_rescue(d, r);
  }
  const long long t2 = d.prof ? clock64() : 0;
  if (d.prof && threadIdx.x == 0) { const unsigned long long dt = (unsigned long long)(t2 - t1); atomicAdd(&d.prof[29], dt); atomicMax(&d.prof[30], dt); }
  // the others: a group of 16 lanes each (cm_s4a_rescue, its two searches shared out over the minimizers)
  const uint32_t t = threadIdx.x % CM_RS_G, grp = threadIdx.x / CM_RS_G, gpb = 64 / CM_RS_G;
  for (uint32_t j = blockIdx.x * gpb + grp; j < cnt; j += gridDim.x * gpb) {
    const uint32_t r = list[j];
    if (!cm_rescue_is_heavy(d, r, coop)) continue;  // uniform in the group
    const uint32_t o = r ^ 1u;
    uint32_t cntn = 0, cntp = 0, rl = 0, rl_val = 0;
    int res_neg = 0, res_pos = 0;
    bool set_rl = false;
    if (d.ncp[o] > 0) {
      res_neg = cm_group_rescue_count(d, r, 1, cm_c0_pos(d, o), cm_c0_pcnt(d, o), d.ncp[o], t, &cntn, &rl);
      if (res_neg >= 0) { set_rl = true; rl_val = rl; }
    }
    if (d.ncn[o] > 0) {
      res_pos = cm_group_rescue_count(d, r, 0, cm_c0_neg(d, o), cm_c0_ncnt(d, o), d.ncn[o], t, &cntp, &rl);
      if (res_pos >= 0) { set_rl = true; rl_val = rl; }
    }
    if (t == 0) {
      d.aug[r] = 1;
      d.res_neg[r] = res_neg; d.res_pos[r] = res_pos;
      d.resc_n[r] = cntn; d.resc_p[r] = cntp;
      if (set_rl) d.rep_len[r] = rl_val;
      d.m_tot[r] = d.ncp[r] + d.ncn[r] + cntn + cntp;
    }
  }
  if (d.prof && threadIdx.x == 0) { const unsigned long long dt = (unsigned long long)(clock64() - t2); atomicAdd(&d.prof[31], dt); atomicMax(&d.prof[15], dt); }
}
// coop: a read without rescue hits but with a long candidate list (a read from a repeat whose mate is one too) only has that list
// copied -- hundreds of entries by one lane; such reads join list 6, where a wave copies them (cm_coop_rescue_merge)
#define CM_S4B_COPY_MIN 64u
__global__ __launch_bounds__(CM_BLOCK) void k_s4b_rescue_merge(CmDev d, uint32_t n, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t i0 = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t i = i0 < n ? (d.perm_reads ? d.perm_reads[i0] : i0) : 0u;
  const bool mine = i0 < n && !(d.aug[i] && d.resc_n[i] + d.resc_p[i] > 0);  // the others: k_s4b_rescue_list
  const bool to_wave = mine && coop && d.m_tot[i] > CM_S4B_COPY_MIN;
  if (mine && !to_wave) cm_s4b_rescue_merge(d, i);
  if (to_wave) { d.mcp[i] = 0; d.mcn[i] = 0; }  // (until list 6's wave has copied the lists)
  if (coop) cm_wave_append(d.hv_list + (size_t)CM_L_RS_WAVE * d.hv_stride, d.hv_cnt + CM_L_RS_WAVE, to_wave, i);
}
// fill pass of one direction by a group: per-minimizer counts to LDS, lane 0 turns them into offsets (minimizer order = the
// order cm_rescue writes in), the lanes write their minimizers' hits there
__device__ __forceinline__ void cm_group_rescue_fill(const CmDev &d, uint32_t r, int strand, const uint64_t *mp, const uint8_t *mc, uint32_t mn,
                                                     uint32_t t, uint32_t *lds_cnt, uint64_t *out) {
  int max_count, best_num;
  cm_group_best(mc, mn, t, &max_count, &best_num);
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  for (uint32_t mi = t; mi < n; mi += CM_RS_G)
    lds_cnt[mi] = cm_rescue_minimizer(d, strand, mp, mc, mn, max_count, d.pr_kind[b + mi], d.pr_val[b + mi], d.mm_ps[b + mi], nullptr, nullptr);
  cm_group_sync<CM_RS_G>();
  if (t == 0) {
    uint32_t run = 0;
    for (uint32_t mi = 0; mi < n; ++mi) { const uint32_t x = lds_cnt[mi]; lds_cnt[mi] = run; run += x; }
  }
  cm_group_sync<CM_RS_G>();
  for (uint32_t mi = t; mi < n; mi += CM_RS_G)
    (void)cm_rescue_minimizer(d, strand, mp, mc, mn, max_count, d.pr_kind[b + mi], d.pr_val[b + mi], d.mm_ps[b + mi], out + lds_cnt[mi], nullptr);
  cm_group_sync<CM_RS_G>();
}
// Reads with many rescue hits (more than CM_RS_COOP_MIN on a strand) only get their hits written here; sorting, clustering and
// merging them is the work of a group of lanes (k_s4b_coop), by size class: list 6 up to hv_max[0] hits (a wave each), 7 / 8 / 11
// up to hv_max[1] / [2] / [3] (a block of 256 / 512 / 1024 lanes each).  coop == 0: everything by one lane, as before.
#define CM_RS_COOP_MIN 32u
#define CM_S4B_PMAX 7680u  // rescue hits the largest shared work area holds (20 bytes per hit: 8192 would pass the CU's 160 KB)
__host__ __device__ inline uint32_t cm_s4b_pmax(const CmDev &d) { return d.rs_big; }
__device__ __forceinline__ uint32_t cm_rescue_coop_class(const CmDev &d, uint32_t r, uint32_t coop) {
  const uint32_t big = d.resc_p[r] > d.resc_n[r] ? d.resc_p[r] : d.resc_n[r];
  if (!coop || big <= CM_RS_COOP_MIN || d.hv_max[0] == 0) return 0;
  return big <= d.hv_max[0] ? CM_L_RS_WAVE : big <= d.hv_max[1] ? CM_L_RS_B256A : big <= d.hv_max[2] ? CM_L_RS_B256B : big <= d.rs_max3 ? CM_L_RS_B512 : big <= d.rs_big ? CM_L_RS_B1024
         : (d.coop_slab && big <= d.coop_slab_cap) ? CM_L_RS_SLAB : 0u;
}
__global__ __launch_bounds__(64) void k_s4b_rescue_list(CmDev d, uint32_t seg_cap, uint32_t coop) {
  if (d.abort && *d.abort) return;
  __shared__ uint32_t sh_cnt[64 / CM_RS_G][CM_RS_MAXMM];
  const uint32_t cnt = d.rs_cnt[blockIdx.y * 16];
  const uint32_t *list = d.rs_list + (uint64_t)blockIdx.y * seg_cap;
  const long long tl0 = d.prof ? clock64() : 0;
  for (uint32_t j0 = blockIdx.x * 64; j0 < cnt; j0 += gridDim.x * 64) {  // whole waves (the appends below are wave-wide)
    const uint32_t j = j0 + threadIdx.x;
    const uint32_t r = j < cnt ? list[j] : 0u;
    const bool mine = j < cnt && d.resc_n[r] + d.resc_p[r] > 0 && !cm_rescue_is_heavy(d, r, coop);
    const uint32_t cls = mine ? cm_rescue_coop_class(d, r, coop) : 0u;
    if (mine) cm_s4b_rescue_merge(d, r, cls ? CM_S4B_FILL_ONLY : CM_S4B_ALL);
    for (uint32_t c = 6; c <= 8; ++c) cm_wave_append(d.hv_list + (size_t)c * d.hv_stride, d.hv_cnt + c, cls == c, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_RS_B512 * d.hv_stride, d.hv_cnt + CM_L_RS_B512, cls == CM_L_RS_B512, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_RS_B1024 * d.hv_stride, d.hv_cnt + CM_L_RS_B1024, cls == CM_L_RS_B1024, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_RS_SLAB * d.hv_stride, d.hv_cnt + CM_L_RS_SLAB, cls == CM_L_RS_SLAB, r);
  }
  const long long tg0 = d.prof ? clock64() : 0;
  if (d.prof && threadIdx.x == 0) { const unsigned long long dt = (unsigned long long)(tg0 - tl0); atomicAdd(&d.prof[34], dt); atomicMax(&d.prof[35], dt); }
  const uint32_t t = threadIdx.x % CM_RS_G, grp = threadIdx.x / CM_RS_G, gpb = 64 / CM_RS_G;
  for (uint32_t j0 = blockIdx.x * gpb; j0 < cnt; j0 += gridDim.x * gpb) {
    const uint32_t j = j0 + grp;
    const uint32_t r = j < cnt ? list[j] : 0u;
    const bool mine = j < cnt && cm_rescue_is_heavy(d, r, coop) && d.resc_n[r] + d.resc_p[r] > 0;  // uniform in the group
    uint32_t cls = 0;
    if (mine) {
      const uint32_t o = r ^ 1u;
      const uint32_t ncp = d.ncp[r], ncn = d.ncn[r], rp = d.resc_p[r], rn = d.resc_n[r];
      uint64_t *P = d.mbuf + d.m_off[r];
      uint64_t *N = P + ncp + rp;
      // same conditions and destinations as cm_s4b_rescue_merge
      if (d.ncp[o] > 0 && d.res_neg[r] >= 0 && rn > 0)
        cm_group_rescue_fill(d, r, 1, cm_c0_pos(d, o), cm_c0_pcnt(d, o), d.ncp[o], t, sh_cnt[grp], N + ncn);
      if (d.ncn[o] > 0 && d.res_pos[r] >= 0 && rp > 0)
        cm_group_rescue_fill(d, r, 0, cm_c0_neg(d, o), cm_c0_ncnt(d, o), d.ncn[o], t, sh_cnt[grp], P + ncp);
      if (t == 0) { d.mcp[r] = 0; d.mcn[r] = 0; }  // (until the hits are sorted and merged: here, or by the class's group)
      __threadfence_block();
      cls = cm_rescue_coop_class(d, r, coop);
      if (t == 0 && !cls) cm_s4b_rescue_merge(d, r, CM_S4B_PREFILLED);  // sort, cluster, merge of the hits the group wrote
    }
    for (uint32_t c = 6; c <= 8; ++c) cm_wave_append(d.hv_list + (size_t)c * d.hv_stride, d.hv_cnt + c, t == 0 && cls == c, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_RS_B512 * d.hv_stride, d.hv_cnt + CM_L_RS_B512, t == 0 && cls == CM_L_RS_B512, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_RS_B1024 * d.hv_stride, d.hv_cnt + CM_L_RS_B1024, t == 0 && cls == CM_L_RS_B1024, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_RS_SLAB * d.hv_stride, d.hv_cnt + CM_L_RS_SLAB, t == 0 && cls == CM_L_RS_SLAB, r);
  }
  if (d.prof && threadIdx.x == 0) { const unsigned long long dt = (unsigned long long)(clock64() - tg0); atomicAdd(&d.prof[36], dt); atomicMax(&d.prof[37], dt); }
}
// The reads of list 23 (their mate has CM_RS_WAVE candidates or more on a strand): a wave per read.  Kernels of their own so that the
// list kernels above keep their shared memory free (three lanes' kernels share the CUs); the grid strides over the device-side list
// and leaves at once when it is empty.
// SMALL: tables for searches with up to CM_RESCUE_WMAX_S best mate candidates (4.5 KB: 32 waves per CU instead of 15, cm_coop.h) over
// list 23; the reads with a longer search are appended to list 31, which the launch with the full tables (SMALL = false) works through.
template <bool SMALL>
__global__ __launch_bounds__(64) void k_s4a_rescue_wave(CmDev d) {
  __shared__ __attribute__((aligned(16))) uint8_t rmem[SMALL ? CM_RESCUE_MEM_BYTES_S : CM_RESCUE_MEM_BYTES];
  const uint32_t li = SMALL ? CM_L_SEARCH_WAVE : CM_L_SEARCH_WAVE_BIG;
  const uint32_t cnt = d.hv_cnt[li];
  if (blockIdx.x >= cnt) return;
  const uint32_t *list = d.hv_list + (size_t)li * d.hv_stride;
  const CmCoopRescueMem m = SMALL ? cm_coop_rescue_mem_at(rmem, CM_RESCUE_WMAX_S, CM_RESCUE_PAIRS_S) : cm_coop_rescue_mem_at(rmem);
  CmDevGroup<64> g;
  g.t = threadIdx.x;
  g.xw = nullptr;
  cm_coop_rescue_mem_reset(g, m);
  const long long t0 = d.prof ? clock64() : 0;
  CmWaveQueue wq(d.hv_cnt + (SMALL ? CM_WQ_S4A_SMALL : CM_WQ_S4A_BIG), cnt, d.wq_dynamic);
  for (uint32_t j; wq.next(&j);) {
    const uint32_t r = list[j];
    if (SMALL && d.prof) {  // measurement aid (tools/coop_profile.py): the best mate candidates of the wave kernel's searches
      const uint32_t o = r ^ 1u;
      const uint32_t a = cm_coop_rescue_best_num(g, cm_c0_pcnt(d, o), d.ncp[o]), b = cm_coop_rescue_best_num(g, cm_c0_ncnt(d, o), d.ncn[o]);
      const uint32_t mx = a > b ? a : b;
      if (threadIdx.x == 0) atomicAdd(&d.prof[mx < 16 ? 40 : mx < 32 ? 41 : mx < 64 ? 42 : mx < 128 ? 43 : mx < 200 ? 44 : mx < 300 ? 45 : 46], 1ull);
    }
    if (SMALL && !cm_coop_rescue_fits(d, r, g, CM_RESCUE_WMAX_S)) {
      if (threadIdx.x == 0) d.hv_list[(size_t)CM_L_SEARCH_WAVE_BIG * d.hv_stride + atomicAdd(d.hv_cnt + CM_L_SEARCH_WAVE_BIG, 1u)] = r;
      continue;
    }
    const long long ti = d.prof ? clock64() : 0;
    cm_coop_s4a_rescue(d, r, g, m);
    g.sync();
    if (d.prof && threadIdx.x == 0) atomicMax(&d.prof[22], (unsigned long long)(clock64() - ti));  // the longest single search pair
  }
  cm_coop_rescue_mem_flush(d, g, m);
  if (d.prof && threadIdx.x == 0) { const unsigned long long dt = (unsigned long long)(clock64() - t0); atomicAdd(&d.prof[27], dt); atomicMax(&d.prof[28], dt); }
}
// the fill pass: the hits written (or copied out of the pool the counting pass left them in), then sorted / merged by the size class's
// group (k_s4b_coop) -- or, a short list, by lane 0 here
template <bool SMALL>
__global__ __launch_bounds__(64) void k_s4b_rescue_wave(CmDev d, uint32_t coop) {
  if (d.abort && *d.abort) return;
  __shared__ __attribute__((aligned(16))) uint8_t rmem[SMALL ? CM_RESCUE_MEM_BYTES_S : CM_RESCUE_MEM_BYTES];
  const uint32_t li = SMALL ? CM_L_SEARCH_WAVE : CM_L_SEARCH_WAVE_BIG;
  const uint32_t cnt = d.hv_cnt[li];
  if (blockIdx.x >= cnt) return;
  const uint32_t *list = d.hv_list + (size_t)li * d.hv_stride;
  const CmCoopRescueMem m = SMALL ? cm_coop_rescue_mem_at(rmem, CM_RESCUE_WMAX_S, CM_RESCUE_PAIRS_S) : cm_coop_rescue_mem_at(rmem);
  CmDevGroup<64> g;
  g.t = threadIdx.x;
  g.xw = nullptr;
  const long long t0 = d.prof ? clock64() : 0;
  CmWaveQueue wq(d.hv_cnt + (SMALL ? CM_WQ_S4B_SMALL : CM_WQ_S4B_BIG), cnt, d.wq_dynamic);
  for (uint32_t j; wq.next(&j);) {
    const uint32_t r = list[j];
    if (d.resc_n[r] + d.resc_p[r] == 0) continue;  // nothing found: k_s4b_rescue_merge copies the read's own candidates
    if (SMALL && !cm_coop_rescue_fits(d, r, g, CM_RESCUE_WMAX_S)) continue;  // (list 31: the launch with the full tables)
    cm_coop_s4b_fill(d, r, g, m);
    g.sync();
    const uint32_t cls = cm_rescue_coop_class(d, r, coop);
    if (threadIdx.x == 0) {
      if (cls) d.hv_list[(size_t)cls * d.hv_stride + atomicAdd(d.hv_cnt + cls, 1u)] = r;
      else cm_s4b_rescue_merge(d, r, CM_S4B_PREFILLED);
    }
  }
  if (d.prof && threadIdx.x == 0) { const unsigned long long dt = (unsigned long long)(clock64() - t0); atomicAdd(&d.prof[32], dt); atomicMax(&d.prof[33], dt); }
}
// S4b for the reads listed above: a group per read sorts its rescue hits, clusters them and merges them with the read's
// candidates (cm_coop_rescue_merge).  The list's length is only known on the device: the grid strides over it.
template <int G>
__global__ __launch_bounds__(G < CM_BLOCK ? CM_BLOCK : G) void k_s4b_coop(CmDev d, const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list_dev, uint32_t P, uint32_t RB,
                                                                      uint32_t use_slab) {
  if (d.abort && *d.abort) return;
  const uint32_t gpb = blockDim.x / G, grp = threadIdx.x / G;
  const uint32_t n_list = *n_list_dev;
  const size_t gb = cm_coop_group_bytes(P, 1, RB, true);
  uint8_t *base = cm_lds + (size_t)grp * gb;
  CmCoopMem m = cm_coop_mem_at(base, P, 1, RB, true);
  if (use_slab) cm_coop_slab_at(m, d.coop_slab + (size_t)blockIdx.x * cm_coop_slab_bytes(d.coop_slab_cap), d.coop_slab_cap);
  CmDevGroup<G> g;
  g.t = threadIdx.x % G;
  g.xw = reinterpret_cast<uint32_t *>(base + gb - CM_XW_BYTES);
  for (uint32_t gid = blockIdx.x * gpb + grp; gid < n_list; gid += gridDim.x * gpb) {
    if (use_slab) cm_coop_rescue_merge<true>(d, list[gid], g, m); else cm_coop_rescue_merge<false>(d, list[gid], g, m);
    g.sync();  // the work area is reused
  }
}
// S4c; long filtered candidate lists are queued for k_sort_lists.  coop: a pair with a merged candidate list longer than
// CM_S4C_COOP_MIN entries only gets the part before the filter here and goes to list 9 for k_s4c_coop (a wave per pair).
#define CM_S4C_COOP_MIN 48u
#define CM_S5C_COOP_MIN 48u  // candidates of a read above which S5 (sorting the lists, alignments, acceptance) is a wave's work
#define CM_S5C_P_SMALL 512u   // most reads of the wave class: work arrays of this size (more waves per CU)
#define CM_S5C_P_WAVE 2048u   // candidates of a strand the wave's work arrays hold
#define CM_S5C_P_BLOCK 16384u // ... a block's (longer lists: the acceptance loop by one lane)
__device__ __forceinline__ void cm_s4c_queue_sort(const CmDev &d, uint32_t pair, bool live, uint32_t coop) {
  for (uint32_t q = 0; q < 4; ++q) {  // (read, strand) lists of the pair
    const uint32_t r = 2 * pair + (q >> 1);
    uint32_t cnt = live ? ((q & 1u) ? d.fcn[r] : d.fcp[r]) : 0u;
    if ((coop & 8u) && !d.p.split && live && d.fcp[r] + d.fcn[r] > CM_S5C_COOP_MIN) cnt = 0;  // the S5 waves sort this read's lists themselves
    cm_wave_append(d.srt_list, &d.srt_cnt[0], cnt > CM_SORT_SERIAL_MAX && cnt <= CM_SORT_WAVE_MAX, (r << 1) | (q & 1u));
  }
}
#define CM_S4C_P_SMALL 256u   // most pairs with long lists: a wave each (the classes above take a block of 256 lanes per pair)
#define CM_S4C_P_WAVE 1024u   // entries of a candidate list the work arrays of a wave hold
#define CM_S4C_P_BLOCK 4096u  // ... of a block
#define CM_S4C_P_BIG 15360u   // ... of a block of 1024 lanes that leaves the position lists in global memory (a lane walking such a
                              // pair's two-pointer loop at global latency was the whole 16 ms of k_s4c_reduce on the mosaic genome)
__global__ __launch_bounds__(CM_BLOCK) void k_s4c_reduce(CmDev d, uint32_t n, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t pair = i < n ? (d.perm_pairs ? d.perm_pairs[i] : i) : 0u;
  uint32_t cls = 0;
  if (i < n && cm_s4c_pre(d, pair)) {
    const uint32_t r1 = 2 * pair, r2 = r1 + 1;
    uint32_t big = d.mcp[r1] > d.mcn[r1] ? d.mcp[r1] : d.mcn[r1];
    big = d.mcp[r2] > big ? d.mcp[r2] : big;
    big = d.mcn[r2] > big ? d.mcn[r2] : big;
    if ((coop & 4u) && big > CM_S4C_COOP_MIN) cls = big <= CM_S4C_P_SMALL ? CM_L_PF_WAVE : big <= CM_S4C_P_WAVE ? CM_L_PF_BLOCK : big <= CM_S4C_P_BLOCK ? CM_L_PF_BLOCK_BIG : big <= d.s4c_pbig ? CM_L_PF_HUGE : 0u;
    if (!cls) { cm_s4c_filter(d, pair); cm_s4c_post(d, pair); }
  }
  if (coop & 4u) {
    cm_wave_append(d.hv_list + CM_L_PF_BLOCK * (size_t)d.hv_stride, d.hv_cnt + CM_L_PF_BLOCK, cls == CM_L_PF_BLOCK, pair);
    cm_wave_append(d.hv_list + CM_L_PF_WAVE * (size_t)d.hv_stride, d.hv_cnt + CM_L_PF_WAVE, cls == CM_L_PF_WAVE, pair);
    cm_wave_append(d.hv_list + CM_L_PF_BLOCK_BIG * (size_t)d.hv_stride, d.hv_cnt + CM_L_PF_BLOCK_BIG, cls == CM_L_PF_BLOCK_BIG, pair);
    cm_wave_append(d.hv_list + CM_L_PF_HUGE * (size_t)d.hv_stride, d.hv_cnt + CM_L_PF_HUGE, cls == CM_L_PF_HUGE, pair);
  }
  if (!d.perm_pairs) return;  // the queue is only served in a batch with heavy reads
  cm_s4c_queue_sort(d, pair, i < n && !cls && d.alive[pair], coop);
}
// the pairs of list 9 / 14: the filter's two directions by a wave / a block each (cm_coop_s4c); the list's length is on the device
template <int G, bool STAGED>
__global__ __launch_bounds__(G < CM_BLOCK ? CM_BLOCK : G) void k_s4c_coop(CmDev d, uint32_t P, uint32_t lid, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t gpb = blockDim.x / G, grp = threadIdx.x / G;
  const uint32_t n_list = d.hv_cnt[lid];
  const uint32_t *list = d.hv_list + (size_t)lid * d.hv_stride;
  const size_t gb = ((cm_coop_pair_mem_bytes(P, STAGED) + 15) & ~(size_t)15) + CM_XW_BYTES;
  uint8_t *base = cm_lds + (size_t)grp * gb;
  const CmCoopPairMem m = cm_coop_pair_mem_at(base, P, STAGED);
  CmDevGroup<G> g;
  g.t = threadIdx.x % G;
  g.xw = reinterpret_cast<uint32_t *>(base + gb - CM_XW_BYTES);
  for (uint32_t j0 = blockIdx.x * gpb; j0 < n_list; j0 += gridDim.x * gpb) {
    const uint32_t j = j0 + grp;
    const uint32_t pair = j < n_list ? list[j] : 0u;
    if (j < n_list) cm_coop_s4c<STAGED>(d, pair, g, m);
    g.sync();
    if (d.perm_pairs) {  // long filtered lists: queued for the sorting waves
      const bool live = j < n_list && d.alive[pair];
      for (uint32_t q = 0; q < 4; ++q) {
        const uint32_t r = 2 * pair + (q >> 1);
        uint32_t cnt = live ? ((q & 1u) ? d.fcn[r] : d.fcp[r]) : 0u;
        if ((coop & 8u) && live && d.fcp[r] + d.fcn[r] > CM_S5C_COOP_MIN) cnt = 0;  // sorted by the S5 waves
        if (g.t == 0 && cnt > CM_SORT_SERIAL_MAX && cnt <= CM_SORT_WAVE_MAX) d.srt_list[atomicAdd(&d.srt_cnt[0], 1u)] = (r << 1) | (q & 1u);
      }
    }
  }
}
// S5a.  coop: a read with more than CM_S5C_COOP_MIN candidates is left to a wave -- alignments and acceptance loop (k_s5c_coop, list 12)
__global__ __launch_bounds__(CM_BLOCK) void k_s5a_prepare(CmDev d, uint32_t n, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t r = i < n ? (d.perm_reads ? d.perm_reads[i] : i) : 0u;
  const bool to_wave = i < n && cm_s5a_prepare(d, r, coop ? CM_S5C_COOP_MIN : 0u);
  // list 12: a wave per read; list 22: a block per read -- a strand's list is longer than the wave's work arrays
  const bool big = to_wave && (d.fcp[r] > CM_S5C_P_WAVE || d.fcn[r] > CM_S5C_P_WAVE);
  const bool small = to_wave && d.fcp[r] <= CM_S5C_P_SMALL && d.fcn[r] <= CM_S5C_P_SMALL;  // list 28: a quarter of the work arrays
  if (coop) {
    cm_wave_append(d.hv_list + (size_t)CM_L_S5_SMALL * d.hv_stride, d.hv_cnt + CM_L_S5_SMALL, small, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_S5_WAVE * d.hv_stride, d.hv_cnt + CM_L_S5_WAVE, to_wave && !big && !small, r);
    cm_wave_append(d.hv_list + (size_t)CM_L_S5_BLOCK * d.hv_stride, d.hv_cnt + CM_L_S5_BLOCK, big, r);
  }
}
// S5c; long draft-mapping lists are queued for k_sort_lists (S6a sorts them by position; split alignment keeps emission order)
__global__ __launch_bounds__(CM_BLOCK) void k_s5c_finalize(CmDev d, uint32_t n, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t r = i < n ? (d.perm_reads ? d.perm_reads[i] : i) : 0u;
  const uint32_t cmin = coop ? CM_S5C_COOP_MIN : 0u;
  if (i < n) cm_s5c_finalize(d, r, cmin);
  if (!d.perm_reads || d.p.split || d.p.single) return;  // the queue is only served in a batch with heavy reads
  const bool live = i < n && !(cmin && d.nv[r] > cmin) && d.alive[r >> 1];  // (a wave's reads: sorted there)
  const uint32_t a = live ? d.ndp[r] : 0u, b = live ? d.ndn[r] : 0u;
  cm_wave_append(d.srt_list, &d.srt_cnt[0], a > CM_SORT_SERIAL_MAX && a <= CM_SORT_WAVE_MAX, r << 1);
  cm_wave_append(d.srt_list, &d.srt_cnt[0], b > CM_SORT_SERIAL_MAX && b <= CM_SORT_WAVE_MAX, (r << 1) | 1u);
}
// S5b of the reads in list 12: a wave per read, its lanes over the read's candidates (cm_coop_s5b)
#define CM_SORT_NB 64u  // counts below this go through the groups' counting sort of a candidate list
#define CM_SORT_STAGE 512u  // candidates of a list the sorting wave stages in shared memory (32 + 18 KB per block of four waves)
// the candidate lists of the reads in list 12 (k_s5a_prepare left them unsorted): a wave each (cm_coop_s5_sort)
__global__ __launch_bounds__(CM_BLOCK, 8) void k_s5_sort_coop(CmDev d, uint32_t lid) {
  if (d.abort && *d.abort) return;
  // (a wave keeps the sort's bins in its lanes: no histogram array -- 18 KB per block, eight blocks per CU)
  __shared__ uint64_t stage_p[(CM_BLOCK / 64) * CM_SORT_STAGE];  // a list of up to CM_SORT_STAGE candidates is staged here once
  __shared__ uint8_t stage_c[(CM_BLOCK / 64) * CM_SORT_STAGE];
  const uint32_t grp = threadIdx.x / 64;
  const uint32_t n_list = d.hv_cnt[lid];
  const uint32_t *list = d.hv_list + (size_t)lid * d.hv_stride;
  CmDevGroup<64> g;
  g.t = threadIdx.x % 64;
  g.xw = nullptr;
  const long long t0 = d.prof ? clock64() : 0;
  uint32_t mine = 0;
  CmWaveQueue wq(d.hv_cnt + (lid == CM_L_S5_SMALL ? CM_WQ_S5_SORT_SMALL : lid == CM_L_S5_WAVE ? CM_WQ_S5_SORT_WAVE : CM_WQ_S5_SORT_BLOCK), n_list, d.wq_dynamic);
  for (uint32_t j; wq.next(&j); ++mine)
    cm_coop_s5_sort(d, list[j], g, nullptr, CM_SORT_NB, stage_p + (size_t)grp * CM_SORT_STAGE, stage_c + (size_t)grp * CM_SORT_STAGE, CM_SORT_STAGE);
  if (d.prof && g.t == 0 && mine) {  // measurement aid (tools/coop_profile.py): a sorting wave's cycles and reads
    const unsigned long long dt = (unsigned long long)(clock64() - t0);
    atomicAdd(&d.prof[38], dt); atomicMax(&d.prof[39], dt); atomicAdd(&d.prof[6], (unsigned long long)mine);
  }
}
#define CM_S5C_SORT_P 1024u  // draft mappings a wave sorts in shared memory (longer lists: in global memory)
#define CM_S5C_SORT_RB 130u
template <int G>
__global__ __launch_bounds__(CM_BLOCK) void k_s5c_coop(CmDev d, uint32_t P, uint32_t lid) {
  if (d.abort && *d.abort) return;
  const uint32_t gpb = blockDim.x / G, grp = threadIdx.x / G;
  const uint32_t n_list = d.hv_cnt[lid];
  const uint32_t *list = d.hv_list + (size_t)lid * d.hv_stride;
  const size_t b1 = cm_coop_ver_mem_bytes(P), b2 = cm_coop_sort_mem_bytes(CM_S5C_SORT_P, CM_S5C_SORT_RB);
  const size_t gb = (((b1 > b2 ? b1 : b2) + 15) & ~(size_t)15) + CM_XW_BYTES;  // the sort's buffers overlay the acceptance loop's arrays
  uint8_t *base = cm_lds + (size_t)grp * gb;
  const CmCoopVerMem m = cm_coop_ver_mem_at(base, P);
  const CmCoopSortMem sm = cm_coop_sort_mem_at(base, CM_S5C_SORT_P, CM_S5C_SORT_RB);
  CmDevGroup<G> g;
  g.t = threadIdx.x % G;
  g.xw = reinterpret_cast<uint32_t *>(base + gb - CM_XW_BYTES);
  for (uint32_t j = blockIdx.x * gpb + grp; j < n_list; j += gridDim.x * gpb) {
    cm_coop_s5c(d, list[j], g, m, sm);
    g.sync();
  }
}

// ---------------------------------------------------------------------------------------
// Long candidate / draft-mapping lists (reads from repeats: hundreds of entries) sorted by a wave each, in LDS, before
// the per-read stage that needs them in order -- a lane heap-sorting 300 entries in global memory took milliseconds and
// there are ~10^5 such reads in a batch.  MODE 0: candidates of (read, strand) by Candidate::operator< (count descending,
// position ascending; the (count, position) pairs of a list are distinct); MODE 1: draft mappings by position (the order
// among equal positions does not matter, mapping_metadata.h:70-78).  Persistent waves pull items off the queue.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(CM_BLOCK) void k_sort_lists(CmDev d) {
  if (d.abort && *d.abort) return;
  const uint32_t wv = threadIdx.x >> 6, t = threadIdx.x & 63;
  uint64_t *K = reinterpret_cast<uint64_t *>(cm_lds) + (size_t)wv * CM_SORT_WAVE_MAX;
  uint16_t *V = reinterpret_cast<uint16_t *>(cm_lds + (size_t)(CM_BLOCK / 64) * CM_SORT_WAVE_MAX * 8) + (size_t)wv * CM_SORT_WAVE_MAX;
  const uint32_t n_items = d.srt_cnt[0];
  for (;;) {
    uint32_t item = 0;
    if (t == 0) item = atomicAdd(&d.srt_cnt[1], 1u);
    item = __shfl(item, 0, 64);
    if (item >= n_items) return;
    const uint32_t code = d.srt_list[item], r = code >> 1, strand = code & 1u;
    const uint32_t base = d.m_off[r] + (strand ? d.ncp[r] + d.resc_p[r] : 0);
    const uint32_t n = MODE == 0 ? (strand ? d.fcn[r] : d.fcp[r]) : (strand ? d.ndn[r] : d.ndp[r]);
    uint64_t *gk = (MODE == 0 ? d.fbuf : d.dpos) + base;
    uint32_t P2 = 2;
    while (P2 < n) P2 <<= 1;
    for (uint32_t i = t; i < P2; i += 64) {
      if (i < n) {
        K[i] = gk[i];
        V[i] = MODE == 0 ? (uint16_t)(255u - d.fcnt[base + i]) : (uint16_t)d.derr[base + i];
      } else {
        K[i] = ~0ull;
        V[i] = 0xffffu;  // padding sorts last in both modes (MODE 0: behind every count)
      }
    }
    cm_group_sync<64>();
    for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1) {
      for (uint32_t j = k2 >> 1, lj = 31u - (uint32_t)__clz(k2 >> 1); j > 0; j >>= 1, --lj) {
        for (uint32_t i = t; i < (P2 >> 1); i += 64) {
          const uint32_t a = ((i >> lj) << (lj + 1)) | (i & (j - 1)), c = a + j;  // j = 1 << lj
          const uint64_t xa = K[a], xc = K[c];
          const uint16_t va = V[a], vc = V[c];
          // "a sorts after c"
          const bool gt = MODE == 0 ? (va != vc ? va > vc : xa > xc) : (xa != xc ? xa > xc : va > vc);
          const bool up = (a & k2) == 0;
          if (gt == up && (xa != xc || va != vc)) { K[a] = xc; K[c] = xa; V[a] = vc; V[c] = va; }
        }
        cm_group_sync<64>();
      }
    }
    for (uint32_t i = t; i < n; i += 64) {
      gk[i] = K[i];
      if (MODE == 0) d.fcnt[base + i] = (uint8_t)(255u - V[i]); else d.derr[base + i] = (int16_t)V[i];
    }
    cm_group_sync<64>();
  }
}
void cm_launch_k_sort_lists(const CmDev &d, int mode, hipStream_t s) {
  const size_t lds = (size_t)(CM_BLOCK / 64) * CM_SORT_WAVE_MAX * 10;
  if (mode == 0) hipLaunchKernelGGL(k_sort_lists<0>, dim3(1024), dim3(CM_BLOCK), lds, s, d);
  else hipLaunchKernelGGL(k_sort_lists<1>, dim3(1024), dim3(CM_BLOCK), lds, s, d);
}
// the number of work items is v_off[n_reads] (device side); the grid covers an upper bound, surplus blocks leave
__global__ __launch_bounds__(CM_BLOCK) void k_s5b_verify(CmDev d, uint32_t n_reads) {
  if (d.abort && *d.abort) return;
  const uint32_t n_items = d.v_off[n_reads];
  for (uint32_t j = blockIdx.x * CM_BLOCK + threadIdx.x; j < n_items; j += gridDim.x * CM_BLOCK) cm_s5b_verify_item(d, j, n_reads);
}
// --SAM has its own instantiations: the alignment's register window must not cost the BED path occupancy
// S6a.  coop: a pair with a draft-mapping list longer than CM_S6A_COOP_MIN goes to list 13, where a wave runs its two sweeps with
// the second list of each direction staged in shared memory -- or, when read 2 has a list longer than CM_S6A_P_WAVE, to list 18,
// where a block does (lists up to CM_S6A_P_BLOCK staged, longer ones read where they are)
#define CM_S6A_COOP_MIN 48u
#define CM_S6A_P_SMALL 256u
#define CM_S6A_P_WAVE 1024u
#define CM_S6A_P_BLOCK 8192u
__device__ __forceinline__ uint32_t cm_s6_class(const CmDev &d, uint32_t pair, uint32_t small_list, uint32_t wave_list, uint32_t block_list) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  uint32_t big2 = d.ndp[r2] > d.ndn[r2] ? d.ndp[r2] : d.ndn[r2], big = big2;
  big = d.ndp[r1] > big ? d.ndp[r1] : big;
  big = d.ndn[r1] > big ? d.ndn[r1] : big;
  if (big <= CM_S6A_COOP_MIN) return 0;
  return big2 <= CM_S6A_P_SMALL ? small_list : big2 <= CM_S6A_P_WAVE ? wave_list : block_list;
}
__global__ __launch_bounds__(CM_BLOCK) void k_s6a_pair(CmDev d, uint32_t n, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t pair = i < n ? (d.perm_pairs ? d.perm_pairs[i] : i) : 0u;
  uint32_t cls = 0;
  if (i < n && cm_s6a_pre<false>(d, pair)) {
    cls = coop ? cm_s6_class(d, pair, CM_L_S6A_SMALL, CM_L_S6A_WAVE, CM_L_S6A_BLOCK) : 0u;
    if (!cls) cm_s6a_sweeps<false>(d, pair);
  }
  if (coop) {
    cm_wave_append(d.hv_list + (size_t)CM_L_S6A_SMALL * d.hv_stride, d.hv_cnt + CM_L_S6A_SMALL, cls == CM_L_S6A_SMALL, pair);
    cm_wave_append(d.hv_list + (size_t)CM_L_S6A_WAVE * d.hv_stride, d.hv_cnt + CM_L_S6A_WAVE, cls == CM_L_S6A_WAVE, pair);
    cm_wave_append(d.hv_list + (size_t)CM_L_S6A_BLOCK * d.hv_stride, d.hv_cnt + CM_L_S6A_BLOCK, cls == CM_L_S6A_BLOCK, pair);
  }
}
template <int G>
__global__ __launch_bounds__(CM_BLOCK, 6) void k_s6a_coop(CmDev d, uint32_t P, uint32_t lid) {
  if (d.abort && *d.abort) return;
  const uint32_t gpb = blockDim.x / G, grp = threadIdx.x / G;
  const uint32_t n_list = d.hv_cnt[lid];
  const uint32_t *list = d.hv_list + (size_t)lid * d.hv_stride;
  const size_t gb = ((cm_coop_pe_mem_bytes(P) + 15) & ~(size_t)15) + CM_XW_BYTES;
  uint8_t *base = cm_lds + (size_t)grp * gb;
  const CmCoopPeMem m = cm_coop_pe_mem_at(base, P);
  CmDevGroup<G> g;
  g.t = threadIdx.x % G;
  g.xw = reinterpret_cast<uint32_t *>(base + gb - CM_XW_BYTES);
  for (uint32_t j = blockIdx.x * gpb + grp; j < n_list; j += gridDim.x * gpb) cm_coop_s6a<false>(d, list[j], g, m);
}
// S6c.  coop: a multi-mapped pair with a draft-mapping list longer than CM_S6A_COOP_MIN goes to list 17 (a wave) / 20 (a block),
// where the group finds the sampled pairings (cm_coop_s6c) -- one lane repeating both pairing sweeps over lists of hundreds of
// entries held its wave for the whole kernel (k_s6c_multi 1.6 ms for 28 k multi-mapped pairs of the repeat workload)
__global__ __launch_bounds__(CM_BLOCK) void k_s6c_multi(CmDev d, uint32_t n, uint32_t coop) {
  if (d.abort && *d.abort) return;
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  const uint32_t pair = i < n ? (d.perm_pairs ? d.perm_pairs[i] : i) : 0u;
  uint32_t cls = 0;
  if (i < n) {
    if (coop && !d.p.single && !d.p.split && d.pe_nbest[pair] > 1) cls = cm_s6_class(d, pair, CM_L_S6C_SMALL, CM_L_S6C_WAVE, CM_L_S6C_BLOCK);
    if (!cls) cm_s6c_multi<false>(d, pair);
  }
  if (coop) {
    cm_wave_append(d.hv_list + (size_t)CM_L_S6C_SMALL * d.hv_stride, d.hv_cnt + CM_L_S6C_SMALL, cls == CM_L_S6C_SMALL, pair);
    cm_wave_append(d.hv_list + (size_t)CM_L_S6C_WAVE * d.hv_stride, d.hv_cnt + CM_L_S6C_WAVE, cls == CM_L_S6C_WAVE, pair);
    cm_wave_append(d.hv_list + (size_t)CM_L_S6C_BLOCK * d.hv_stride, d.hv_cnt + CM_L_S6C_BLOCK, cls == CM_L_S6C_BLOCK, pair);
  }
}
template <int G>
__global__ __launch_bounds__(CM_BLOCK) void k_s6c_coop(CmDev d, uint32_t P, uint32_t lid) {
  if (d.abort && *d.abort) return;
  const uint32_t gpb = blockDim.x / G, grp = threadIdx.x / G;
  const uint32_t n_list = d.hv_cnt[lid];
  const uint32_t *list = d.hv_list + (size_t)lid * d.hv_stride;
  const size_t gb = ((cm_coop_pe_mem_bytes(P) + 15) & ~(size_t)15) + CM_XW_BYTES;
  uint8_t *base = cm_lds + (size_t)grp * gb;
  const CmCoopPeMem m = cm_coop_pe_mem_at(base, P);
  CmDevGroup<G> g;
  g.t = threadIdx.x % G;
  g.xw = reinterpret_cast<uint32_t *>(base + gb - CM_XW_BYTES);
  for (uint32_t j = blockIdx.x * gpb + grp; j < n_list; j += gridDim.x * gpb) cm_coop_s6c<false>(d, list[j], g, m);
}
CM_ITEM_KERNEL(k_s6a_pair_sam, cm_s6a_pair<true>, perm_pairs)
CM_ITEM_KERNEL(k_s6c_multi_sam, cm_s6c_multi<true>, perm_pairs)

// S6b, one WAVE per taskloop chunk: the 64 lanes scan the chunk's n_best values coalesced
// and ballot the multi-mappers; lane 0 then walks them in pair order with the chunk's
// mt19937 (state in LDS).
__global__ __launch_bounds__(64) void k_s6b_sample(CmDev d, uint32_t n_chunks) {
  if (d.abort && *d.abort) return;
  const uint32_t chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  __shared__ CmMt g;
  uint32_t lo, hi;
  cm_chunk_range(d.n_pairs, (uint32_t)d.p.ref_batch, (uint32_t)d.p.grain, chunk, &lo, &hi);
  bool seeded = false;
  for (uint32_t base = lo; base < hi; base += 64) {
    const uint32_t pair = base + threadIdx.x;
    int nb = 0;
    if (pair < hi) nb = d.pe_nbest[pair];
    const bool multi = nb > 1 && (d.p.single || nb <= d.p.drop_rep);
    unsigned long long m = __ballot(multi);
    if (m == 0) continue;
    if (threadIdx.x == 0) {
      while (m) {
        if (!seeded || d.p.single) { cm_mt_seed(g, 11); seeded = true; }  // single-end: fresh generator per read
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t pr = base + (uint32_t)l;
        cm_reservoir(d, pr, d.pe_nbest[pr], g);
      }
    }
    seeded = __shfl((int)seeded, 0, 64) != 0;
  }
}

// K6: barcode correction; per-block reduction of the two counters, one atomic pair per block
__global__ __launch_bounds__(CM_BLOCK) void k_s0b_barcode(CmDev d, uint32_t n) {
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  uint32_t a = 0, b = 0;
  if (i < n) cm_s0b_barcode(d, i, &a, &b);
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  __shared__ uint32_t sa[CM_BLOCK / 64], sb[CM_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int j = 1; j < CM_BLOCK / 64; ++j) { a += sa[j]; b += sb[j]; }
    if (a) atomicAdd(&d.stats[CM_ST_BC_INWL], (unsigned long long)a);
    if (b) atomicAdd(&d.stats[CM_ST_BC_CORR], (unsigned long long)b);
  }
}

// ComputeBarcodeAbundance (chromap.cc:492-548) for barcodes [lo, hi): whitelist hits without N
// increment the entry's count; *num_sample += hits
__global__ __launch_bounds__(CM_BLOCK) void k_bc_abundance(const uint8_t *__restrict__ bcb, const uint32_t *__restrict__ bco,
                                                            uint32_t lo, uint32_t hi, uint64_t *__restrict__ wl,
                                                            uint32_t wl_mask, unsigned long long *__restrict__ num_sample) {
  const uint32_t i = lo + blockIdx.x * CM_BLOCK + threadIdx.x;
  uint32_t hit = 0;
  if (i < hi) {
    const uint8_t *s = bcb + bco[i];
    const uint32_t l = bco[i + 1] - bco[i];
    bool has_n = false;
    for (uint32_t j = 0; j < l; ++j) has_n |= s[j] == 'N';
    if (!has_n) {
      const uint64_t key = cm_seed_from_sequence(s, l);
      const uint64_t x = key * 0x9E3779B97F4A7C15ull;
      uint32_t b = (uint32_t)(x >> 32) & wl_mask;
      for (;;) {
        const uint64_t k = wl[2 * (uint64_t)b];
        if (k == ~0ull) break;
        if (k == key) { atomicAdd(reinterpret_cast<unsigned long long *>(wl + 2 * (uint64_t)b + 1), 1ull); hit = 1; break; }
        b = (b + 1) & wl_mask;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) hit += __shfl_down(hit, off, 64);
  if ((threadIdx.x & 63) == 0 && hit) atomicAdd(num_sample, (unsigned long long)hit);
}

// ---------------------------------------------------------------------------------------
// The index-probe kernel (graded roofline kernel).  A lookup is a chain of dependent 16-byte bucket
// gathers from the HBM-resident table (kh_get's triangular probing, 1.7 buckets per lookup on the
// GRCh38-sized table); what bounds it is how many of them the memory system has in flight.  With one
// lookup per lane a wave issues 64 gathers, then waits on the few lanes that need a second, third, ...
// bucket while its other lanes idle.  Here every lane carries U independent lookups:
//   * the U first buckets are requested back to back (U gathers in flight per lane);
//   * bucket i+1 -- the second probe step -- is requested together with bucket i whenever both lie in one
//     64-byte sector (i mod 4 != 3): no extra HBM traffic, and 3 of 4 second steps no longer cost a
//     dependent round trip;
//   * the remaining steps of the U lookups advance together, one gather per unfinished lookup per round.
// A block covers U*256 consecutive minimizers, lane t takes t, t+256, ...: loads and stores stay coalesced.
// Results (hit / miss, value, number of buckets visited) are kh_get's, bucket for bucket.
// ---------------------------------------------------------------------------------------
#define CM_PROBE_U 1
template <int U, bool PAIR>
__device__ __forceinline__ void cm_probe_lanes(const uint64_t *__restrict__ bkt, uint32_t bmask, const uint64_t *__restrict__ hash,
                                               uint64_t *__restrict__ val, uint8_t *__restrict__ kind, unsigned long long lo,
                                               unsigned long long hi, unsigned long long tile0, uint32_t *steps_out, uint32_t *hits_out) {
  uint64_t h[U], v[U];
  uint32_t idx[U], first[U], step[U];
  uint8_t kd[U];
  bool live[U];
  uint32_t visited = 0, hits = 0;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const unsigned long long i = tile0 + (unsigned long long)j * CM_BLOCK + threadIdx.x;
    live[j] = i >= lo && i < hi;
    h[j] = live[j] ? hash[i] : 0;
    idx[j] = (uint32_t)h[j] & bmask;
    first[j] = idx[j];
    step[j] = 0;
    v[j] = 0;
    kd[j] = CM_PR_MISS;
  }
  // round 0: bucket i and, when it shares the sector, bucket i+1 (= probe step 1)
  {
    ulonglong2 a[U], b[U];
    bool two[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      two[j] = PAIR && live[j] && (idx[j] & 3u) != 3u && idx[j] != bmask;
      if (live[j]) a[j] = *reinterpret_cast<const ulonglong2 *>(bkt + 2 * (uint64_t)idx[j]);
      if (two[j]) b[j] = *reinterpret_cast<const ulonglong2 *>(bkt + 2 * (uint64_t)idx[j] + 2);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (!live[j]) continue;
      ++visited;
      if (a[j].x == CM_EMPTY_KEY) { live[j] = false; continue; }
      if (a[j].x != CM_DELETED_KEY && (a[j].x >> 1) == h[j]) { v[j] = a[j].y; kd[j] = (a[j].x & 1) ? CM_PR_SINGLE : CM_PR_MULTI; ++hits; live[j] = false; continue; }
      idx[j] = (idx[j] + (++step[j])) & bmask;  // step 1: bucket i+1
      if (idx[j] == first[j]) { live[j] = false; continue; }
      if (two[j]) {
        ++visited;
        if (b[j].x == CM_EMPTY_KEY) { live[j] = false; continue; }
        if (b[j].x != CM_DELETED_KEY && (b[j].x >> 1) == h[j]) { v[j] = b[j].y; kd[j] = (b[j].x & 1) ? CM_PR_SINGLE : CM_PR_MULTI; ++hits; live[j] = false; continue; }
        idx[j] = (idx[j] + (++step[j])) & bmask;
        if (idx[j] == first[j]) live[j] = false;
      }
    }
  }
  // further rounds: one gather per unfinished lookup
  bool any = false;
#pragma unroll
  for (int j = 0; j < U; ++j) any = any || live[j];
  while (any) {
    ulonglong2 a[U];
#pragma unroll
    for (int j = 0; j < U; ++j)
      if (live[j]) a[j] = *reinterpret_cast<const ulonglong2 *>(bkt + 2 * (uint64_t)idx[j]);
    any = false;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (!live[j]) continue;
      ++visited;
      if (a[j].x == CM_EMPTY_KEY) { live[j] = false; continue; }
      if (a[j].x != CM_DELETED_KEY && (a[j].x >> 1) == h[j]) { v[j] = a[j].y; kd[j] = (a[j].x & 1) ? CM_PR_SINGLE : CM_PR_MULTI; ++hits; live[j] = false; continue; }
      idx[j] = (idx[j] + (++step[j])) & bmask;
      if (idx[j] == first[j]) { live[j] = false; continue; }
      any = true;
    }
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const unsigned long long i = tile0 + (unsigned long long)j * CM_BLOCK + threadIdx.x;
    if (i >= lo && i < hi) { val[i] = v[j]; kind[i] = kd[j]; }
  }
  *steps_out = visited;
  *hits_out = hits;
}

// per-block (probe steps, hits): wave reduction, then one plain store per block (a single device-scope
// counter would serialise ~10 ns per atomic)
__device__ __forceinline__ void cm_probe_account(uint32_t steps, uint32_t hit, uint2 *__restrict__ block_partials) {
  __shared__ uint32_t sh_s[CM_BLOCK / 64], sh_h[CM_BLOCK / 64];
  for (int off = 32; off > 0; off >>= 1) {
    steps += __shfl_down(steps, off, 64);
    hit += __shfl_down(hit, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { sh_s[threadIdx.x >> 6] = steps; sh_h[threadIdx.x >> 6] = hit; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, c = 0;
    for (int j = 0; j < CM_BLOCK / 64; ++j) { a += sh_s[j]; c += sh_h[j]; }
    block_partials[blockIdx.x] = make_uint2(a, c);
  }
}

template <int U, bool PAIR>
__global__ __launch_bounds__(CM_BLOCK) void k_probe(const uint64_t *__restrict__ bkt, uint32_t bmask,
                                                     const uint64_t *__restrict__ hash, uint64_t *__restrict__ val,
                                                     uint8_t *__restrict__ kind, uint32_t n,
                                                     uint2 *__restrict__ block_partials) {
  uint32_t steps = 0, hit = 0;
  cm_probe_lanes<U, PAIR>(bkt, bmask, hash, val, kind, 0ull, (unsigned long long)n, (unsigned long long)blockIdx.x * (U * CM_BLOCK), &steps, &hit);
  if (block_partials) cm_probe_account(steps, hit, block_partials);
}

// k_probe over the minimizers [range[0], range[1]) -- the range is only known on the device (it is where
// the chunk's minimizer launch moved the cursor), so the grid covers an upper bound and surplus blocks leave.
template <int U, bool PAIR>
__global__ __launch_bounds__(CM_BLOCK) void k_probe_range(const uint64_t *__restrict__ bkt, uint32_t bmask,
                                                           const uint64_t *__restrict__ hash, uint64_t *__restrict__ val,
                                                           uint8_t *__restrict__ kind, const unsigned long long *__restrict__ range,
                                                           uint32_t cap, uint2 *__restrict__ block_partials) {
  const unsigned long long lo = range[0];
  unsigned long long hi = range[1];
  if (hi > cap) hi = cap;  // overflow of the dense arrays: the host reruns with larger ones
  uint32_t steps = 0, hit = 0;
  const unsigned long long tile0 = lo + (unsigned long long)blockIdx.x * (U * CM_BLOCK);
  if (tile0 < hi) cm_probe_lanes<U, PAIR>(bkt, bmask, hash, val, kind, lo, hi, tile0, &steps, &hit);
  cm_probe_account(steps, hit, block_partials);
}

// sums k_probe's per-block partials into counters[0] (steps) and counters[1] (hits)
__global__ __launch_bounds__(CM_BLOCK) void k_probe_reduce(const uint2 *__restrict__ partials, uint32_t n_blocks,
                                                            unsigned long long *__restrict__ counters) {
  unsigned long long a = 0, c = 0;
  for (uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x; i < n_blocks; i += gridDim.x * CM_BLOCK) { a += partials[i].x; c += partials[i].y; }
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    c += __shfl_down(c, off, 64);
  }
  __shared__ unsigned long long sa[CM_BLOCK / 64], sc[CM_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int j = 1; j < CM_BLOCK / 64; ++j) { a += sa[j]; c += sc[j]; }
    if (a) atomicAdd(&counters[0], a);
    if (c) atomicAdd(&counters[1], c);
  }
}

// per-pair counters of Chromap::OutputMappingStatistics (chromap.h:1057-1058, 1118-1137);
// block-level reduction in LDS, one row of 8 partials per block, summed by k_stats_reduce
#define CM_NSTAT 8
#define CM_STATS_BLOCKS 2048u
__global__ __launch_bounds__(CM_BLOCK) void k_stats(CmDev d, uint32_t n, unsigned long long *__restrict__ partials) {
  unsigned long long v[CM_NSTAT] = {0, 0, 0, 0, 0, 0, 0, 0};
  // the grid strides over the pairs: at most CM_STATS_BLOCKS rows of partials for k_stats_reduce's single block
  for (uint32_t pair = blockIdx.x * CM_BLOCK + threadIdx.x; pair < n; pair += gridDim.x * CM_BLOCK) {
    const uint32_t r1 = 2 * pair, r2 = r1 + 1;
    if (d.alive[pair]) {
      v[0] += d.fcp[r1] + d.fcn[r1] + d.fcp[r2] + d.fcn[r2];
      const uint32_t nd1 = d.ndp[r1] + d.ndn[r1], nd2 = d.ndp[r2] + d.ndn[r2];
      const unsigned long long per = d.p.single ? 1ull : 2ull;  // reads per item
      if (nd1 > 0 && (d.p.single || nd2 > 0)) {
        const int nb = d.pe_nbest[pair];
        if (nb == 1) v[3] += per;
        v[1] += per * (unsigned long long)(nb < d.p.max_best ? nb : d.p.max_best);
        if (nb > 0) v[2] += per;
        if (nb > 1 && (d.p.single || nb <= d.p.drop_rep)) v[4] += 1;
      }
    }
    v[5] += (unsigned long long)d.aug[r1] + d.aug[r2];
    v[6] += (unsigned long long)d.hit_tot[r1] + d.hit_tot[r2];
    for (uint32_t t = 0; t < (uint32_t)d.p.max_best; ++t) v[7] += d.rec_ok[(uint64_t)pair * (uint32_t)d.p.max_best + t];
  }
  __shared__ unsigned long long sh[CM_BLOCK / 64][CM_NSTAT];
#pragma unroll
  for (int k = 0; k < CM_NSTAT; ++k) {
    unsigned long long x = v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][k] = x;
  }
  __syncthreads();
  if (threadIdx.x < CM_NSTAT) {
    unsigned long long x = 0;
    for (int j = 0; j < CM_BLOCK / 64; ++j) x += sh[j][threadIdx.x];
    partials[(uint64_t)blockIdx.x * CM_NSTAT + threadIdx.x] = x;
  }
}

__global__ __launch_bounds__(CM_BLOCK) void k_stats_reduce(const unsigned long long *__restrict__ partials,
                                                            uint32_t n_blocks, unsigned long long *__restrict__ stats) {
  // thread t handles statistic t % 8 over rows t/8, t/8 + 32, ...
  const int k = threadIdx.x % CM_NSTAT, lane_row = threadIdx.x / CM_NSTAT;
  unsigned long long x = 0;
  for (uint32_t r = lane_row; r < n_blocks; r += CM_BLOCK / CM_NSTAT) x += partials[(uint64_t)r * CM_NSTAT + k];
  __shared__ unsigned long long sh[CM_BLOCK];
  sh[threadIdx.x] = x;
  __syncthreads();
  if (threadIdx.x < CM_NSTAT) {
    unsigned long long t = 0;
    for (int j = 0; j < CM_BLOCK / CM_NSTAT; ++j) t += sh[j * CM_NSTAT + threadIdx.x];
    const int slot[CM_NSTAT] = {CM_ST_CAND, CM_ST_MAPPINGS, CM_ST_MAPPED, CM_ST_UNIQ, CM_ST_MULTI, CM_ST_RESCUED, CM_ST_OCC, CM_ST_RECORDS};
    stats[slot[threadIdx.x]] += t;
  }
}

// ---------------------------------------------------------------------------------------
// exclusive prefix sum of uint32: n inputs -> n+1 outputs (out[n] = total).  rocPRIM's
// single-pass decoupled look-back scan over n+1 elements with in[n] forced to 0 (every
// counted array is allocated with one spare element).
// ---------------------------------------------------------------------------------------
size_t cm_scan_tmp_words(uint32_t n) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)n + 1,
                                rocprim::plus<uint32_t>(), hipStream_t(0));
  return bytes / 4 + 64;
}

void cm_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *tmp, hipStream_t s) {
  (void)hipMemsetAsync(const_cast<uint32_t *>(in) + n, 0, sizeof(uint32_t), s);
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s);
  (void)rocprim::exclusive_scan((void *)tmp, bytes, in, out, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), s);
}

// heavy-last order of reads and pairs: flags (a read is heavy when its hit list went to the cooperative kernel, a pair when
// either read is), exclusive scans of the flags, scatter: light items keep their order in front, heavy items follow
__global__ __launch_bounds__(CM_BLOCK) void k_hv_flags(CmDev d, uint32_t n_pairs, uint32_t *__restrict__ fr, uint32_t *__restrict__ fp) {
  const uint32_t p = blockIdx.x * CM_BLOCK + threadIdx.x;
  if (p >= n_pairs) return;
  const uint32_t lim = d.hv_mid > d.s3b_cap ? d.hv_mid : d.s3b_cap;
  const uint32_t a = d.hit_tot[2 * p] > lim ? 1u : 0u, b = d.hit_tot[2 * p + 1] > lim ? 1u : 0u;
  fr[2 * p] = a; fr[2 * p + 1] = b;
  fp[p] = a | b;
}
// scan[i] = heavy items before i, scan[n] = all heavy items
__global__ __launch_bounds__(CM_BLOCK) void k_hv_scatter(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ scan, uint32_t n,
                                                          uint32_t *__restrict__ perm) {
  const uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint32_t heavy_before = scan[i], n_light = n - scan[n];
  perm[flag[i] ? n_light + heavy_before : i - heavy_before] = i;
}
// fr / sr: 2 n_pairs + 1 words each, fp / sp: n_pairs + 1 words each (scratch)
void cm_build_heavy_last(const CmDev &d, uint32_t n_pairs, uint32_t *fr, uint32_t *sr, uint32_t *fp, uint32_t *sp, uint32_t *perm_reads,
                         uint32_t *perm_pairs, uint32_t *scan_tmp, hipStream_t s) {
  if (!n_pairs) return;
  hipLaunchKernelGGL(k_hv_flags, grid_for_n(n_pairs), dim3(CM_BLOCK), 0, s, d, n_pairs, fr, fp);
  cm_scan_u32(fr, sr, 2 * n_pairs, scan_tmp, s);
  cm_scan_u32(fp, sp, n_pairs, scan_tmp, s);
  hipLaunchKernelGGL(k_hv_scatter, grid_for_n(2 * n_pairs), dim3(CM_BLOCK), 0, s, (const uint32_t *)fr, (const uint32_t *)sr, 2 * n_pairs, perm_reads);
  hipLaunchKernelGGL(k_hv_scatter, grid_for_n(n_pairs), dim3(CM_BLOCK), 0, s, (const uint32_t *)fp, (const uint32_t *)sp, n_pairs, perm_pairs);
}

// dst[0] = src[0]: the cursor value after a chunk's minimizer launch (a device-to-device hipMemcpyAsync of 8 bytes runs as a
// ~35 us blit kernel, eight of them sat between the chunks' minimizer and probe launches)
__global__ void k_copy_u64(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst) { dst[0] = src[0]; }
// *flag = 1 when *total > cap (the arrays of this batch were sized ahead of its totals)
__global__ void k_check_cap(const unsigned long long *__restrict__ total, unsigned long long cap, unsigned long long *__restrict__ flag) {
  if (*total > cap) *flag = 1ull;
}
void cm_launch_k_check_cap(const unsigned long long *total, unsigned long long cap, unsigned long long *flag, hipStream_t s) {
  hipLaunchKernelGGL(k_check_cap, dim3(1), dim3(1), 0, s, total, cap, flag);
}
// the reference bytes as interleaved bit-plane records (CmDev::ref_pl): one lane per record of 32 bases, one 16-byte store
__global__ __launch_bounds__(CM_BLOCK) void k_pack_ref(const uint8_t *__restrict__ ref, uint64_t n_bytes, CmPlRec *__restrict__ pl) {
  const uint64_t w = (uint64_t)blockIdx.x * CM_BLOCK + threadIdx.x;
  if (w * 32 >= n_bytes) return;
  const uint64_t left = n_bytes - w * 32;
  CmPlRec r;
  cm_pack_planes32(ref + w * 32, left < 32 ? (uint32_t)left : 32u, &r.p0, &r.p1, &r.pn, &r.pc);
  pl[w] = r;
}
void cm_launch_k_pack_ref(const uint8_t *ref, uint64_t n_bytes, CmPlRec *pl, hipStream_t s) {
  const uint64_t n = (n_bytes + 31) / 32;
  if (!n) return;
  hipLaunchKernelGGL(k_pack_ref, dim3((unsigned)((n + CM_BLOCK - 1) / CM_BLOCK)), dim3(CM_BLOCK), 0, s, ref, n_bytes, pl);
}
// the batch's reads as bit planes, both orientations (CmDev::read_pl): one lane per read.  The words per plane are a template
// parameter of the KERNEL (W = 0: any number, the read-back form): with the choice inside the kernel its registers were those of the
// largest case -- 17.8 ms per 4 M pairs of 50-base reads next to 1.3 ms, on the second stream under S3 / S4 (round 4, when the
// cases for 150-base reads were added).
template <int W>
__global__ __launch_bounds__(CM_BLOCK) void k_pack_reads(CmDev d, uint32_t n_reads) {
  const uint32_t r = blockIdx.x * CM_BLOCK + threadIdx.x;
  if (r >= n_reads) return;
  if (W == 0) cm_pack_read_planes_any(d, r); else cm_pack_read_planes_w<(W > 0 ? W : 1)>(d, r);
}
void cm_launch_k_pack_reads(const CmDev &d, uint32_t n_reads, hipStream_t s) {
  if (!n_reads) return;
#define CM_PACK_CASE(W_) case W_: hipLaunchKernelGGL(k_pack_reads<W_>, grid_for_n(n_reads), dim3(CM_BLOCK), 0, s, d, n_reads); break;
  switch (d.read_pl_w) {
    CM_PACK_CASE(1) CM_PACK_CASE(2) CM_PACK_CASE(3) CM_PACK_CASE(4) CM_PACK_CASE(5) CM_PACK_CASE(6) CM_PACK_CASE(7) CM_PACK_CASE(8)
    default: hipLaunchKernelGGL(k_pack_reads<0>, grid_for_n(n_reads), dim3(CM_BLOCK), 0, s, d, n_reads);
  }
#undef CM_PACK_CASE
}
void cm_launch_k_copy_u64(const unsigned long long *src, unsigned long long *dst, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_u64, dim3(1), dim3(1), 0, s, src, dst);
}

// *out += sum of in[0..n) in 64 bits (the u32 prefix sums above wrap silently; the callers size and bound the
// dense arrays from this total)
__global__ __launch_bounds__(CM_BLOCK) void k_sum_u32(const uint32_t *__restrict__ in, uint32_t n, unsigned long long *__restrict__ out) {
  unsigned long long a = 0;
  for (uint32_t i = blockIdx.x * CM_BLOCK + threadIdx.x; i < n; i += gridDim.x * CM_BLOCK) a += in[i];
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  __shared__ unsigned long long sa[CM_BLOCK / 64];
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int j = 1; j < CM_BLOCK / 64; ++j) a += sa[j];
    if (a) atomicAdd(out, a);
  }
}
void cm_launch_k_sum_u32(const uint32_t *in, uint32_t n, unsigned long long *out, hipStream_t s) {
  if (!n) return;
  uint32_t blocks = (n + CM_BLOCK * 8 - 1) / (CM_BLOCK * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_sum_u32, dim3(blocks), dim3(CM_BLOCK), 0, s, in, n, out);
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n) { return dim3((n + CM_BLOCK - 1) / CM_BLOCK); }

#define CM_LAUNCH(kname)                                                               \
  void cm_launch_##kname(const CmDev &d, uint32_t n, hipStream_t s) {                  \
    if (n) hipLaunchKernelGGL(kname, grid_for(n), dim3(CM_BLOCK), 0, s, d, n);         \
  }
void cm_launch_k_s3a_count(const CmDev &d, uint32_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_s3a_count, grid_for(n), dim3(CM_BLOCK), 0, s, d, n);
}
uint32_t cm_s3b_lane_cap(uint32_t max_read_len) {
  uint32_t cap = max_read_len / 3;
  return cap < 16 ? 16 : (cap > 64 ? 64 : cap);
}
// the cooperative kernel's size classes: hits per wave-group, per block, per block with a large LDS allocation
// Round 4: finer classes (a wave up to 256 / 512 hits, 256 lanes up to 1024 / 2048, 512 up to 4096, 1024 up to 8192 with the large
// allocation): the stages are chains of shared-memory round trips, a group's work area is what limits the waves a CU holds, and a
// work area of the list's own size class was worth 20 % of the stage on the mosaic genome.
void cm_s3b_heavy_classes(uint32_t *hv_max, uint32_t *hv_big) {
  // HIP function attributes belong to the device the call is made on (one process may drive several: chromap-amd --gpus N),
  // so the large-LDS opt-in is made -- and remembered -- per device; lane threads of one device may race here, hence atomics
  static std::atomic<int> big_ok[64];  // 0 unknown, 1 granted, 2 refused
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int slot = dev >= 0 && dev < 64 ? dev : 0;
  int st = big_ok[slot].load(std::memory_order_acquire);
  if (st == 0 || dev != slot) {
    st = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_s3b_heavy<CM_BLOCK>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 10 + (CM_BLOCK + 8) * 4) == hipSuccess ? 1 : 2;
    (void)hipGetLastError();
    if (dev == slot) big_ok[slot].store(st, std::memory_order_release);
  }
  hv_max[0] = 512; hv_max[1] = 1024; hv_max[2] = 2048; hv_max[3] = 4096;
  *hv_big = st == 1 ? 8192 : 0;
}
// Speculative launch set (CmDev::cls_mask): the kernels of a long-list class are launched when the class had items in the context's
// previous range (or always: the first range, a re-run).  An empty launch is not free -- a class kernel asks for up to 150 KB of shared
// memory and waits for a CU that has it, under three lanes' other kernels: 0.2-0.5 ms each, 8 % of a step of the hic workload
// (profiles/r04p_hic_kernel_stats.csv: k_s4c_coop<1024, false> 0.49 ms with no pair to work on).  The host compares the lists' counts
// with the mask at the end of the range and maps the range again with every class on if an unlaunched one had items.
static inline bool cm_cls_on(const CmDev &d, uint32_t list) { return (d.cls_mask >> list) & 1ull; }
// the largest dynamic LDS allocation a kernel of this device may ask for, opted in once per device and kernel
template <class K>
static bool cm_lds_optin(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  if (bytes > 160 * 1024) return false;
  // function attributes belong to the device the call is made on; remember the largest size granted (or refused) per device
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, std::pair<size_t, size_t>> seen;  // -> (granted up to, refused from)
  int dev = 0;
  (void)hipGetDevice(&dev);
  const std::pair<int, const void *> key(dev, reinterpret_cast<const void *>(kernel));
  std::lock_guard<std::mutex> lk(mu);
  auto it = seen.find(key);
  if (it != seen.end()) {
    if (bytes <= it->second.first) return true;
    if (it->second.second && bytes >= it->second.second) return false;
  }
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)hipGetLastError();
  std::pair<size_t, size_t> &v = seen[key];
  if (e == hipSuccess) { if (bytes > v.first) v.first = bytes; } else if (!v.second || bytes < v.second) v.second = bytes;
  return e == hipSuccess;
}
// n_cls[c]: reads of list c (k_s3a_count; 11 entries).  coop: lists 0, 1, 2, 10 go through the merge-sort kernel (tables sized for
// reads of up to max_read_len bases) with a wave / 256 / 512 / 1024 lanes per read; what it declines -- hv_cnt[CM_L_HIT_DECLINED] reads at list 5 --
// is taken by the bitonic kernel.
static inline uint32_t cm_coop_mm(const CmDev &d, uint32_t max_read_len) {
  // a read of L bases has at most L - k + 1 minimizers; the tables hold that many (capped)
  uint32_t MM = max_read_len > (uint32_t)d.p.k ? max_read_len - (uint32_t)d.p.k + 1 : 1;
  return MM > 256 ? 256 : MM;
}
void cm_launch_k_s3b_heavy(const CmDev &d, const uint32_t *n_cls, hipStream_t s, bool coop, uint32_t max_read_len) {
  auto pow2 = [](uint32_t x) { uint32_t p = 2; while (p < x) p <<= 1; return p; };  // the sort network's size: a power of two
  auto lst = [&](uint32_t c) { return (const uint32_t *)(d.hv_list + (size_t)c * d.hv_stride); };
  uint32_t rest[CM_HV_LISTS];
  for (int c = 0; c < CM_HV_LISTS; ++c) rest[c] = n_cls[c];
  if (coop && d.hv_max[0]) {
    const uint32_t MM = cm_coop_mm(d, max_read_len);
    const uint32_t RB = d.coop_rb ? d.coop_rb : 3 * MM + 2;  // a run per minimizer and strand, the + list's table padded to a power of two (cm_coop_s3b_expand)
    uint32_t *fb_list = d.hv_list + CM_L_HIT_DECLINED * (size_t)d.hv_stride, *fb_cnt = d.hv_cnt + CM_L_HIT_DECLINED;
    bool any_coop = false;
    const bool k32 = d.goff != nullptr;  // (the reference fits 32-bit hit keys: every class's work area shrinks from 19 to 11 bytes per hit)
    // (a block per read: blocks that stride over the list -- 2048 / 8192 of them -- were measured 3-17 % slower, the reads differ too much in size)
#define CM_S3B_LAUNCH(G_, GRID_, BLOCK_, LDS_, ...)                                                                         \
    do { if (k32) hipLaunchKernelGGL((k_s3b_coop<G_, true>), GRID_, BLOCK_, LDS_, s, __VA_ARGS__);                          \
         else hipLaunchKernelGGL((k_s3b_coop<G_, false>), GRID_, BLOCK_, LDS_, s, __VA_ARGS__); } while (0)
#define CM_S3B_OPTIN(G_, LDS_) (k32 ? cm_lds_optin(&k_s3b_coop<G_, true>, LDS_) : cm_lds_optin(&k_s3b_coop<G_, false>, LDS_))
    if (n_cls[CM_L_HIT_WAVE]) {  // a wave per read, two reads per block
      const size_t lds = 2 * cm_coop_group_bytes(d.hv_max[0], MM, RB, false, k32);
      if (CM_S3B_OPTIN(64, lds)) {
        CM_S3B_LAUNCH(64, dim3((n_cls[CM_L_HIT_WAVE] + 1) / 2), dim3(128), lds, d, lst(CM_L_HIT_WAVE), n_cls[CM_L_HIT_WAVE], d.hv_max[0], MM, RB, fb_list, fb_cnt, 0u);
        rest[CM_L_HIT_WAVE] = 0; any_coop = true;
      }
    }
    if (n_cls[CM_L_HIT_WAVE_SMALL] && d.hv_sub) {  // the wave class's short lists: four reads per block of 256 lanes, a quarter of the work area each
      const size_t lds = 4 * cm_coop_group_bytes(d.hv_sub, MM, RB, false, k32);
      if (CM_S3B_OPTIN(64, lds)) {
        CM_S3B_LAUNCH(64, dim3((n_cls[CM_L_HIT_WAVE_SMALL] + 3) / 4), dim3(256), lds, d, lst(CM_L_HIT_WAVE_SMALL), n_cls[CM_L_HIT_WAVE_SMALL], d.hv_sub, MM, RB, fb_list, fb_cnt, 0u);
        rest[CM_L_HIT_WAVE_SMALL] = 0; any_coop = true;
      }
    }
#define CM_S3B_COOP_CLASS(C_, Q_, G_)                                                                                                              \
    if (n_cls[C_] && d.hv_max[Q_] > d.hv_max[Q_ - 1]) {                                                                                            \
      const size_t lds = cm_coop_group_bytes(d.hv_max[Q_], MM, RB, false, k32);                                                                    \
      if (CM_S3B_OPTIN(G_, lds)) {                                                                                                                 \
        CM_S3B_LAUNCH(G_, dim3(n_cls[C_]), dim3(G_), lds, d, lst(C_), n_cls[C_], d.hv_max[Q_], MM, RB, fb_list, fb_cnt, 0u);                      \
        rest[C_] = 0; any_coop = true;                                                                                                             \
      }                                                                                                                                            \
    }
    CM_S3B_COOP_CLASS(CM_L_HIT_B256A, 1, 256)
    CM_S3B_COOP_CLASS(CM_L_HIT_B256B, 2, 256)   // (512 lanes for 1024 .. 2048 hits and 1024 for 2048 .. 4096 were measured slower: more lanes
    CM_S3B_COOP_CLASS(CM_L_HIT_B512, 3, 512)  //  per hit cost more in barriers than the shorter chunks save; ~8 hits per lane it is)
#undef CM_S3B_COOP_CLASS
    if (n_cls[CM_L_HIT_B1024] && d.hv_big > d.hv_max[3]) {  // the largest work area: one block per CU
      const size_t lds = cm_coop_group_bytes(d.hv_big, MM, RB, false, k32);
      if (CM_S3B_OPTIN(1024, lds)) {
        CM_S3B_LAUNCH(1024, dim3(n_cls[CM_L_HIT_B1024]), dim3(1024), lds, d, lst(CM_L_HIT_B1024), n_cls[CM_L_HIT_B1024], d.hv_big, MM, RB, fb_list, fb_cnt, 0u);
        rest[CM_L_HIT_B1024] = 0; any_coop = true;
      }
    }
    if (n_cls[CM_L_HIT_SLAB] && d.coop_slab && d.hv_max[3]) {  // lists beyond the largest class: 1024 lanes on a slab of global memory each
      const size_t lds = cm_coop_group_bytes(d.hv_max[3], MM, RB, false);
      if (cm_lds_optin(&k_s3b_coop<1024, false>, lds)) {
        const uint32_t blocks = n_cls[CM_L_HIT_SLAB] < d.coop_slab_blocks ? n_cls[CM_L_HIT_SLAB] : d.coop_slab_blocks;
        uint32_t *sl = d.hv_list + CM_L_HIT_SERIAL * (size_t)d.hv_stride, *sc = d.hv_cnt + CM_L_HIT_SERIAL;  // what even that declines: one lane each
        hipLaunchKernelGGL((k_s3b_coop<1024, false>), dim3(blocks), dim3(1024), lds, s, d, lst(CM_L_HIT_SLAB), n_cls[CM_L_HIT_SLAB], d.hv_max[3], MM, RB, sl, sc, 1u);
        hipLaunchKernelGGL(k_s3b_serial, dim3(64), dim3(64), 0, s, d, (const uint32_t *)sl, 0u, (const uint32_t *)sc);
        rest[CM_L_HIT_SLAB] = 0;
      }
    }
    if (any_coop) {  // the declined reads: up to hv_big hits, a block each, the grid strides over the device-side list
      const uint32_t P = pow2(d.hv_big > d.hv_max[3] ? d.hv_big : d.hv_max[3]);
      hipLaunchKernelGGL(k_s3b_heavy<CM_BLOCK>, dim3(512), dim3(CM_BLOCK), (size_t)P * 10 + (CM_BLOCK + 8) * 4, s, d, (const uint32_t *)fb_list, 0u, P, (const uint32_t *)fb_cnt);
    }
  }
  if (rest[CM_L_HIT_WAVE_SMALL]) {
    const uint32_t P = pow2(d.hv_max[0]), gpb = CM_BLOCK / 64;
    hipLaunchKernelGGL(k_s3b_heavy<64>, dim3((rest[CM_L_HIT_WAVE_SMALL] + gpb - 1) / gpb), dim3(CM_BLOCK), (size_t)gpb * P * 10 + gpb * (64 + 8) * 4, s, d, lst(CM_L_HIT_WAVE_SMALL), rest[CM_L_HIT_WAVE_SMALL], P, (const uint32_t *)nullptr);
  }
  if (rest[CM_L_HIT_WAVE]) {
    const uint32_t P = pow2(d.hv_max[0]), gpb = CM_BLOCK / 64;
    hipLaunchKernelGGL(k_s3b_heavy<64>, dim3((rest[CM_L_HIT_WAVE] + gpb - 1) / gpb), dim3(CM_BLOCK), (size_t)gpb * P * 10 + gpb * (64 + 8) * 4, s, d, lst(CM_L_HIT_WAVE), rest[CM_L_HIT_WAVE], P, (const uint32_t *)nullptr);
  }
  const uint32_t blk[3][2] = {{1, 1}, {2, 2}, {10, 3}};  // list, size class
  for (int q = 0; q < 3; ++q) {
    const uint32_t c = blk[q][0];
    if (!rest[c]) continue;
    const uint32_t P = pow2(d.hv_max[blk[q][1]]);
    hipLaunchKernelGGL(k_s3b_heavy<CM_BLOCK>, dim3(rest[c]), dim3(CM_BLOCK), (size_t)P * 10 + (CM_BLOCK + 8) * 4, s, d, lst(c), rest[c], P, (const uint32_t *)nullptr);
  }
  if (rest[CM_L_HIT_B1024]) {
    const uint32_t P = pow2(d.hv_big);
    hipLaunchKernelGGL(k_s3b_heavy<CM_BLOCK>, dim3(rest[CM_L_HIT_B1024]), dim3(CM_BLOCK), (size_t)P * 10 + (CM_BLOCK + 8) * 4, s, d, lst(CM_L_HIT_B1024), rest[CM_L_HIT_B1024], P, (const uint32_t *)nullptr);
  }
  if (rest[CM_L_HIT_SLAB]) hipLaunchKernelGGL(k_s3b_serial, dim3((rest[CM_L_HIT_SLAB] + 63) / 64), dim3(64), 0, s, d, lst(CM_L_HIT_SLAB), rest[CM_L_HIT_SLAB], (const uint32_t *)nullptr);
  if (rest[CM_L_HIT_G16]) {  // groups of 16 lanes, 16 reads per block
    const uint32_t P = pow2(d.hv_mid), gpb = CM_BLOCK / 16;
    hipLaunchKernelGGL(k_s3b_heavy<16>, dim3((rest[CM_L_HIT_G16] + gpb - 1) / gpb), dim3(CM_BLOCK), (size_t)gpb * P * 10 + gpb * (16 + 8) * 4, s, d, lst(CM_L_HIT_G16), rest[CM_L_HIT_G16], P, (const uint32_t *)nullptr);
  }
}
void cm_launch_k_s3b_candidates(const CmDev &d, uint32_t n, uint32_t max_read_len, hipStream_t s) {
  if (!n) return;
  const uint32_t cap = d.s3b_cap;  // cm_s3b_lane_cap(max_read_len) unless overridden (cmgpu_set_option "s3b_lane_cap")
  (void)max_read_len;
  uint32_t threads = 256;
  while (threads > 64 && (size_t)cap * threads * 9 > 36 * 1024 + 1024) threads >>= 1;
  hipLaunchKernelGGL(k_s3b_candidates, dim3((n + threads - 1) / threads), dim3(threads), (size_t)cap * threads * 9, s, d, n, cap);
}
// capacity of one list segment: the reads of every CM_RS_SEGS-th block
uint32_t cm_rescue_seg_cap(uint32_t n_reads) { return ((n_reads + CM_BLOCK - 1) / CM_BLOCK / CM_RS_SEGS + 1) * CM_BLOCK; }
void cm_launch_k_s4a_rescue_count(const CmDev &d, uint32_t n, hipStream_t s, bool coop) {
  if (n) hipLaunchKernelGGL(k_s4a_rescue_count, grid_for(n), dim3(CM_BLOCK), 0, s, d, n, cm_rescue_seg_cap(n), coop ? 1u : 0u);
}
// list kernels: enough waves for every listed read of a typical batch to get a lane at once, grid-stride beyond that
static inline uint32_t rescue_wave_blocks(uint32_t n_reads) {  // waves for list 23: a few per CU and more for large batches
  // (round 6: n / 256 .. n / 1024 measure alike, within 1 % on profile 2 -- a lane of a 4 M-pair batch is at the cap anyway)
  uint32_t b = n_reads / 512 + 64;
  return b > 8192u ? 8192u : b;
}
static inline dim3 rescue_list_grid(uint32_t n_reads) {
  uint32_t b = n_reads / 64 / CM_RS_SEGS / 4 + 1;  // a quarter of the reads listed: one pass
  if (b > 256u) b = 256u;
  return dim3(b, CM_RS_SEGS);
}
void cm_launch_k_s4a_rescue_list(const CmDev &d, uint32_t n_reads, hipStream_t s, bool coop) {
  if (!n_reads) return;
  hipLaunchKernelGGL(k_s4a_rescue_list, rescue_list_grid(n_reads), dim3(64), 0, s, d, cm_rescue_seg_cap(n_reads), coop ? 1u : 0u);
  if ((coop) && cm_cls_on(d, CM_L_SEARCH_WAVE)) {
    hipLaunchKernelGGL(k_s4a_rescue_wave<true>, dim3(rescue_wave_blocks(n_reads)), dim3(64), 0, s, d);
    hipLaunchKernelGGL(k_s4a_rescue_wave<false>, dim3(rescue_wave_blocks(n_reads)), dim3(64), 0, s, d);  // (what the small tables do not hold; leaves at once when there is none)
  }
}
// the per-read part (reads without rescue hits) and, coop: the reads whose long lists a wave copies
static bool cm_s4b_coop_ready(const CmDev &d, uint32_t RB, size_t *lds) {
  if (!d.hv_max[0]) return false;
  lds[0] = 2 * cm_coop_group_bytes(d.hv_max[0], 1, RB, true);
  lds[1] = cm_coop_group_bytes(d.hv_max[1], 1, RB, true);
  lds[2] = cm_coop_group_bytes(d.hv_max[2], 1, RB, true);
  lds[3] = cm_coop_group_bytes(d.rs_max3, 1, RB, true);
  lds[4] = cm_coop_group_bytes(d.rs_big > d.rs_max3 ? d.rs_big : d.rs_max3, 1, RB, true);
  const bool ok = cm_lds_optin(&k_s4b_coop<64>, lds[0]) && cm_lds_optin(&k_s4b_coop<256>, lds[1] > lds[2] ? lds[1] : lds[2]) && cm_lds_optin(&k_s4b_coop<512>, lds[3]) &&
                  cm_lds_optin(&k_s4b_coop<1024>, lds[4]);
  if (!ok) {  // not an error (the one-lane forms take over), but never silently: it costs a factor on repeat-rich input
    static std::atomic<int> told{0};
    if (!told.exchange(1)) fprintf(stderr, "chromap_amd: the cooperative rescue kernels do not fit this device's shared memory (%zu / %zu / %zu / %zu bytes); using the one-lane forms\n", lds[0], lds[1], lds[2], lds[4]);
  }
  return ok;
}
static inline uint32_t cm_s4b_rb(const CmDev &d, uint32_t max_read_len) {
  return d.coop_rb ? d.coop_rb : 2 * cm_coop_mm(d, max_read_len) + 2;  // ascending runs the sorter's tables hold: one per minimizer unless a diagonal wraps
}
void cm_launch_k_s4b_rescue_merge(const CmDev &d, uint32_t n, hipStream_t s, bool coop, uint32_t max_read_len) {
  if (!n) return;
  size_t lds[5];
  const bool all = coop && cm_s4b_coop_ready(d, cm_s4b_rb(d, max_read_len), lds);
  hipLaunchKernelGGL(k_s4b_rescue_merge, grid_for(n), dim3(CM_BLOCK), 0, s, d, n, all ? 1u : 0u);
}
// coop: reads with many rescue hits are only filled by the list kernel and finished by groups of lanes (k_s4b_coop)
void cm_launch_k_s4b_rescue_list(const CmDev &d, uint32_t n_reads, hipStream_t s, bool coop, uint32_t max_read_len) {
  if (!n_reads) return;
  const uint32_t RB = cm_s4b_rb(d, max_read_len);
  size_t lds[5];
  const bool all = coop && cm_s4b_coop_ready(d, RB, lds);
  hipLaunchKernelGGL(k_s4b_rescue_list, rescue_list_grid(n_reads), dim3(64), 0, s, d, cm_rescue_seg_cap(n_reads), all ? 1u : 0u);
  // list 23's reads were counted by waves whenever the option is on: they are filled by waves too (all == false: the groups that
  // would sort long lists do not fit this device -- every list is then finished by the wave's lane 0)
  if ((coop) && cm_cls_on(d, CM_L_SEARCH_WAVE)) {
    hipLaunchKernelGGL(k_s4b_rescue_wave<true>, dim3(rescue_wave_blocks(n_reads)), dim3(64), 0, s, d, all ? 1u : 0u);
    hipLaunchKernelGGL(k_s4b_rescue_wave<false>, dim3(rescue_wave_blocks(n_reads)), dim3(64), 0, s, d, all ? 1u : 0u);
  }
  if (!all) return;
  uint32_t blocks = n_reads / 2048 + 64;  // the listed reads are a few per cent of the batch; surplus blocks leave at once
  if (blocks > 2048) blocks = 2048;
  auto lst = [&](uint32_t c) { return (const uint32_t *)(d.hv_list + (size_t)c * d.hv_stride); };
  if (cm_cls_on(d, CM_L_RS_WAVE)) hipLaunchKernelGGL(k_s4b_coop<64>, dim3(blocks), dim3(128), lds[0], s, d, lst(CM_L_RS_WAVE), (const uint32_t *)(d.hv_cnt + CM_L_RS_WAVE), d.hv_max[0], RB, 0u);
  if ((d.hv_max[1] > d.hv_max[0]) && cm_cls_on(d, CM_L_RS_B256A)) hipLaunchKernelGGL(k_s4b_coop<256>, dim3(blocks), dim3(256), lds[1], s, d, lst(CM_L_RS_B256A), (const uint32_t *)(d.hv_cnt + CM_L_RS_B256A), d.hv_max[1], RB, 0u);
  if ((d.hv_max[2] > d.hv_max[1]) && cm_cls_on(d, CM_L_RS_B256B)) hipLaunchKernelGGL(k_s4b_coop<256>, dim3(blocks), dim3(256), lds[2], s, d, lst(CM_L_RS_B256B), (const uint32_t *)(d.hv_cnt + CM_L_RS_B256B), d.hv_max[2], RB, 0u);
  if ((d.rs_max3 > d.hv_max[2]) && cm_cls_on(d, CM_L_RS_B512)) hipLaunchKernelGGL(k_s4b_coop<512>, dim3(blocks > 512 ? 512 : blocks), dim3(512), lds[3], s, d, lst(CM_L_RS_B512), (const uint32_t *)(d.hv_cnt + CM_L_RS_B512), d.rs_max3, RB, 0u);
  if ((d.rs_big > d.rs_max3) && cm_cls_on(d, CM_L_RS_B1024)) hipLaunchKernelGGL(k_s4b_coop<1024>, dim3(blocks > 256 ? 256 : blocks), dim3(1024), lds[4], s, d, lst(CM_L_RS_B1024), (const uint32_t *)(d.hv_cnt + CM_L_RS_B1024), d.rs_big, RB, 0u);
  if (d.coop_slab && cm_cls_on(d, CM_L_RS_SLAB))  // lists beyond the largest class: on the blocks' slabs of global memory
    hipLaunchKernelGGL(k_s4b_coop<1024>, dim3(d.coop_slab_blocks), dim3(1024), lds[4], s, d, lst(CM_L_RS_SLAB), (const uint32_t *)(d.hv_cnt + CM_L_RS_SLAB), d.rs_big > d.rs_max3 ? d.rs_big : d.rs_max3, RB, 1u);
}
// coop: the cmgpu_set_option "coop" bit mask (bit 2: pairs with long lists to groups; bit 3: the S5 waves sort the heavy reads' lists)
void cm_launch_k_s4c_reduce(const CmDev &d, uint32_t n, hipStream_t s, uint32_t coop) {
  if (!n) return;
  const size_t gw = ((cm_coop_pair_mem_bytes(CM_S4C_P_WAVE) + 15) & ~(size_t)15) + CM_XW_BYTES;
  const size_t gbk = ((cm_coop_pair_mem_bytes(CM_S4C_P_BLOCK) + 15) & ~(size_t)15) + CM_XW_BYTES;
  // the lists beyond that: as many entries as the shared memory a block may have holds at 10 bytes each
  uint32_t pbig = CM_S4C_P_BIG;
  while (pbig > CM_S4C_P_BLOCK && !cm_lds_optin(&k_s4c_coop<1024, false>, ((cm_coop_pair_mem_bytes(pbig, false) + 15) & ~(size_t)15) + CM_XW_BYTES)) pbig >>= 1;
  if (!cm_lds_optin(&k_s4c_coop<CM_BLOCK, true>, gbk)) coop &= ~4u;
  CmDev d2 = d;
  d2.s4c_pbig = pbig > CM_S4C_P_BLOCK ? pbig : 0;
  hipLaunchKernelGGL(k_s4c_reduce, grid_for(n), dim3(CM_BLOCK), 0, s, d2, n, coop);
  if (!(coop & 4u)) return;
  uint32_t blocks = n / 2048 + 64;
  if (blocks > 4096) blocks = 4096;
  const size_t gs = ((cm_coop_pair_mem_bytes(CM_S4C_P_SMALL) + 15) & ~(size_t)15) + CM_XW_BYTES;
  if (cm_cls_on(d, CM_L_PF_WAVE)) hipLaunchKernelGGL((k_s4c_coop<64, true>), dim3(blocks), dim3(CM_BLOCK), (CM_BLOCK / 64) * gs, s, d, CM_S4C_P_SMALL, CM_L_PF_WAVE, coop);  // a wave per pair
  if (cm_cls_on(d, CM_L_PF_BLOCK)) hipLaunchKernelGGL((k_s4c_coop<CM_BLOCK, true>), dim3(blocks), dim3(CM_BLOCK), gw, s, d, CM_S4C_P_WAVE, CM_L_PF_BLOCK, coop);
  if (cm_cls_on(d, CM_L_PF_BLOCK_BIG)) hipLaunchKernelGGL((k_s4c_coop<CM_BLOCK, true>), dim3(256), dim3(CM_BLOCK), gbk, s, d, CM_S4C_P_BLOCK, CM_L_PF_BLOCK_BIG, coop);
  if (d2.s4c_pbig && cm_cls_on(d, CM_L_PF_HUGE))
    hipLaunchKernelGGL((k_s4c_coop<1024, false>), dim3(128), dim3(1024), ((cm_coop_pair_mem_bytes(pbig, false) + 15) & ~(size_t)15) + CM_XW_BYTES, s, d, pbig, 19u, coop);
}
void cm_launch_k_s5a_prepare(const CmDev &d, uint32_t n, hipStream_t s, bool coop) {
  if (!n) return;
  hipLaunchKernelGGL(k_s5a_prepare, grid_for(n), dim3(CM_BLOCK), 0, s, d, n, coop ? 1u : 0u);
  if (coop) {  // the lists it left unsorted: a wave per read
    if (cm_cls_on(d, CM_L_S5_SMALL)) hipLaunchKernelGGL(k_s5_sort_coop, dim3(4096), dim3(CM_BLOCK), 0, s, d, CM_L_S5_SMALL);
    if (cm_cls_on(d, CM_L_S5_WAVE)) hipLaunchKernelGGL(k_s5_sort_coop, dim3(2048), dim3(CM_BLOCK), 0, s, d, CM_L_S5_WAVE);
    if (cm_cls_on(d, CM_L_S5_BLOCK)) hipLaunchKernelGGL(k_s5_sort_coop, dim3(64), dim3(CM_BLOCK), 0, s, d, CM_L_S5_BLOCK);
  }
}
void cm_launch_k_s5c_finalize(const CmDev &d, uint32_t n, hipStream_t s, bool coop) {
  if (!n) return;
  hipLaunchKernelGGL(k_s5c_finalize, grid_for(n), dim3(CM_BLOCK), 0, s, d, n, coop ? 1u : 0u);
  if (!coop) return;
  auto gbytes = [](uint32_t P) {
    const size_t b1 = cm_coop_ver_mem_bytes(P), b2 = cm_coop_sort_mem_bytes(CM_S5C_SORT_P, CM_S5C_SORT_RB);
    return (((b1 > b2 ? b1 : b2) + 15) & ~(size_t)15) + CM_XW_BYTES;
  };
  uint32_t blocks = n / 4096 + 64;
  if (blocks > 2048) blocks = 2048;
  if (cm_cls_on(d, CM_L_S5_SMALL)) hipLaunchKernelGGL(k_s5c_coop<64>, dim3(blocks), dim3(CM_BLOCK), (CM_BLOCK / 64) * gbytes(CM_S5C_P_SMALL), s, d, CM_S5C_P_SMALL, CM_L_S5_SMALL);
  if (cm_cls_on(d, CM_L_S5_WAVE)) hipLaunchKernelGGL(k_s5c_coop<64>, dim3(blocks), dim3(192), 3 * gbytes(CM_S5C_P_WAVE), s, d, CM_S5C_P_WAVE, CM_L_S5_WAVE);  // three waves per block: under the 64 KB a launch gets without asking
  // the reads with a longer list: a block each (what even its work arrays cannot hold: lane 0's acceptance loop, the group's sort)
  uint32_t pb = CM_S5C_P_BLOCK;
  while (pb > CM_S5C_P_WAVE && !cm_lds_optin(&k_s5c_coop<CM_BLOCK>, gbytes(pb))) pb >>= 1;
  if (cm_cls_on(d, CM_L_S5_BLOCK)) hipLaunchKernelGGL(k_s5c_coop<CM_BLOCK>, dim3(128), dim3(CM_BLOCK), gbytes(pb), s, d, pb, CM_L_S5_BLOCK);
}
CM_LAUNCH(k_s6a_pair_sam)
CM_LAUNCH(k_s6c_multi_sam)
// max_items: an upper bound of the item count (the capacity of the candidate arrays)
void cm_launch_k_s5b_verify(const CmDev &d, uint32_t max_items, uint32_t n_reads, hipStream_t s) {
  if (!max_items) return;
  uint32_t blocks = (max_items + CM_BLOCK - 1) / CM_BLOCK;
  if (blocks > 65536) blocks = 65536;  // grid-stride beyond
  hipLaunchKernelGGL(k_s5b_verify, dim3(blocks), dim3(CM_BLOCK), 0, s, d, n_reads);
}
// the pairing stages' groups: a wave per pair with the second list staged (4 waves x 10 KB per block), a block per pair for the
// few pairs whose read 2 has a list beyond CM_S6A_P_WAVE (80 KB)
static inline size_t cm_s6_group_bytes(uint32_t P) { return ((cm_coop_pe_mem_bytes(P) + 15) & ~(size_t)15) + CM_XW_BYTES; }
void cm_launch_k_s6a_pair(const CmDev &d, uint32_t n, hipStream_t s, bool coop) {
  if (!n) return;
  const size_t lw = (CM_BLOCK / 64) * cm_s6_group_bytes(CM_S6A_P_WAVE), lb = cm_s6_group_bytes(CM_S6A_P_BLOCK);
  if (coop && !(cm_lds_optin(&k_s6a_coop<64>, lw) && cm_lds_optin(&k_s6a_coop<CM_BLOCK>, lb))) coop = false;
  hipLaunchKernelGGL(k_s6a_pair, grid_for(n), dim3(CM_BLOCK), 0, s, d, n, coop ? 1u : 0u);
  if (!coop) return;
  uint32_t blocks = n / 4096 + 64;
  if (blocks > 2048) blocks = 2048;
  if (cm_cls_on(d, CM_L_S6A_SMALL)) hipLaunchKernelGGL(k_s6a_coop<64>, dim3(blocks), dim3(CM_BLOCK), (CM_BLOCK / 64) * cm_s6_group_bytes(CM_S6A_P_SMALL), s, d, CM_S6A_P_SMALL, CM_L_S6A_SMALL);
  if (cm_cls_on(d, CM_L_S6A_WAVE)) hipLaunchKernelGGL(k_s6a_coop<64>, dim3(blocks), dim3(CM_BLOCK), lw, s, d, CM_S6A_P_WAVE, CM_L_S6A_WAVE);
  if (cm_cls_on(d, CM_L_S6A_BLOCK)) hipLaunchKernelGGL(k_s6a_coop<CM_BLOCK>, dim3(256), dim3(CM_BLOCK), lb, s, d, CM_S6A_P_BLOCK, CM_L_S6A_BLOCK);
}
void cm_launch_k_s6c_multi(const CmDev &d, uint32_t n, hipStream_t s, bool coop) {
  if (!n) return;
  const size_t lw = (CM_BLOCK / 64) * cm_s6_group_bytes(CM_S6A_P_WAVE), lb = cm_s6_group_bytes(CM_S6A_P_BLOCK);
  if (coop && !(cm_lds_optin(&k_s6c_coop<64>, lw) && cm_lds_optin(&k_s6c_coop<CM_BLOCK>, lb))) coop = false;
  hipLaunchKernelGGL(k_s6c_multi, grid_for(n), dim3(CM_BLOCK), 0, s, d, n, coop ? 1u : 0u);
  if (!coop) return;
  uint32_t blocks = n / 4096 + 64;
  if (blocks > 2048) blocks = 2048;
  if (cm_cls_on(d, CM_L_S6C_SMALL)) hipLaunchKernelGGL(k_s6c_coop<64>, dim3(blocks), dim3(CM_BLOCK), (CM_BLOCK / 64) * cm_s6_group_bytes(CM_S6A_P_SMALL), s, d, CM_S6A_P_SMALL, CM_L_S6C_SMALL);
  if (cm_cls_on(d, CM_L_S6C_WAVE)) hipLaunchKernelGGL(k_s6c_coop<64>, dim3(blocks), dim3(CM_BLOCK), lw, s, d, CM_S6A_P_WAVE, CM_L_S6C_WAVE);
  if (cm_cls_on(d, CM_L_S6C_BLOCK)) hipLaunchKernelGGL(k_s6c_coop<CM_BLOCK>, dim3(256), dim3(CM_BLOCK), lb, s, d, CM_S6A_P_BLOCK, CM_L_S6C_BLOCK);
}

// threads per block / LDS bytes for the read-staging kernels, from the longest read of the batch
static inline void staging_geometry(uint32_t max_read_len, uint32_t *threads, uint32_t *lds_half) {
  uint32_t pb = 128;  // pairs per block
  while (pb > 16 && (uint64_t)pb * max_read_len + 64 > 24 * 1024) pb >>= 1;
  *threads = 2 * pb;
  *lds_half = (uint32_t)(((uint64_t)pb * max_read_len + 64 + 15) & ~15ull);
}
// fused trim + minimizers: geometry (threads per block, LDS) from the longest read; false when the
// configuration needs the two-pass kernels: other k / w, or reads longer than 69 bases -- their emissions
// would leave room for 128 or 64 lanes per block only, and at that occupancy the two-pass kernels are
// as fast (2 x 100) or faster (2 x 150: 10.7 ms against 12.7 ms for 2 M pairs)
static bool prep_mm_geometry(const CmDev &d, uint32_t max_read_len, uint32_t *threads, uint32_t *half, uint32_t *stg, size_t *lds, bool *gstage = nullptr, bool *e6_out = nullptr) {
  if (d.p.w != 7 || 2 * d.p.k + 12 > 64) return false;
  *stg = max_read_len / 4 + 4;
  const bool g = max_read_len > 69;  // the emissions staged in global memory (k_prep_mm<true>)
  const bool e6 = !g && 2 * d.p.k + 8 <= 48;  // (position << 1 | strand of a read of up to 69 bases: 8 bits)
  if (gstage) *gstage = g;
  if (e6_out) *e6_out = e6;
  uint32_t t = 256;
  for (;; t >>= 1) {
    *half = (uint32_t)(((uint64_t)(t / 2) * max_read_len + 64 + 15) & ~15ull);
    *lds = 2 * (size_t)*half + (g ? (size_t)(2 * t + 1) * 4 : (size_t)*stg * t * (e6 ? 6 : 8)) + 64;
    if (*lds <= (g ? 48 : 60) * 1024) break;
    if (t == 64) return false;
  }
  *threads = t;
  return true;
}
// position-parallel kernel: odd k (the closed form), k <= 26 (k-mer + shift inside 96 packed bits), reads up to 69 bases (6-bit k-mer index)
static bool prep_flat_geometry(const CmDev &d, uint32_t max_read_len, uint32_t tile_reads, uint32_t *half, uint32_t *nt_max, size_t *lds) {
  if (d.p.w != 7 || !(d.p.k & 1) || d.p.k > 26 || max_read_len > 69 || max_read_len < (uint32_t)d.p.k + 6) return false;
  *half = (uint32_t)(((uint64_t)(CM_BLOCK / 2) * max_read_len + 64 + 15) & ~15ull);
  *nt_max = tile_reads * (max_read_len - (uint32_t)d.p.k + 1);
  const size_t region0 = 2 * (size_t)*half > 16 * (size_t)*nt_max ? 2 * (size_t)*half : 16 * (size_t)*nt_max;
  const size_t nw = *half / 16 + 1;
  *lds = region0 + 2 * (nw + 3) * 4 + 2 * (nw + 1) * 2 + 8 + CM_BLOCK * sizeof(CmFlatRead) + (CM_BLOCK + 2) * 4 + ((size_t)*nt_max + 2) * 2 * 2 + 8 + 32 * 4;
  return *lds <= 64 * 1024;
}
bool cm_prep_flat_supported(const CmDev &d, uint32_t max_read_len, uint32_t tile_reads) {
  uint32_t h, n;
  size_t l;
  return prep_flat_geometry(d, max_read_len, tile_reads, &h, &n, &l);
}
void cm_launch_k_prep_flat(const CmDev &d, uint32_t pair_lo, uint32_t pair_hi, uint32_t max_read_len, uint32_t tile_reads, uint32_t mm_cap,
                           unsigned long long *cursor, hipStream_t s) {
  uint32_t half, nt_max;
  size_t lds;
  if (pair_hi <= pair_lo || !prep_flat_geometry(d, max_read_len, tile_reads, &half, &nt_max, &lds)) return;
  const uint32_t pb = CM_BLOCK / 2;
  hipLaunchKernelGGL(k_prep_flat, dim3((pair_hi - pair_lo + pb - 1) / pb), dim3(CM_BLOCK), lds, s, d, pair_lo, pair_hi, half, nt_max, tile_reads, mm_cap, cursor);
}
bool cm_prep_mm_supported(const CmDev &d, uint32_t max_read_len) {
  uint32_t t, h, g;
  size_t l;
  return prep_mm_geometry(d, max_read_len, &t, &h, &g, &l);
}
// pairs per block of k_prep_mm for this read length (chunk boundaries are multiples of it)
uint32_t cm_prep_mm_pairs_per_block(const CmDev &d, uint32_t max_read_len) {
  uint32_t threads, half, stg;
  size_t lds;
  return prep_mm_geometry(d, max_read_len, &threads, &half, &stg, &lds) ? threads / 2 : 0;
}
// bytes of the global staging buffer a launch over `pairs` pairs needs (0: the emissions are staged in LDS)
size_t cm_prep_mm_stage_bytes(const CmDev &d, uint32_t max_read_len, uint32_t pairs) {
  uint32_t threads, half, stg;
  size_t lds;
  bool g = false;
  if (!prep_mm_geometry(d, max_read_len, &threads, &half, &stg, &lds, &g) || !g) return 0;
  const uint32_t pb = threads / 2;
  return (size_t)((pairs + pb - 1) / pb) * stg * threads * 8;
}
// pairs [pair_lo, pair_hi)
void cm_launch_k_prep_mm(const CmDev &d, uint32_t pair_lo, uint32_t pair_hi, uint32_t max_read_len, uint32_t mm_cap,
                         unsigned long long *cursor, hipStream_t s, void *gstage) {
  uint32_t threads, half, stg;
  size_t lds;
  bool g = false, e6 = false;
  if (pair_hi <= pair_lo || !prep_mm_geometry(d, max_read_len, &threads, &half, &stg, &lds, &g, &e6)) return;
  const uint32_t pb = threads / 2;
  if (g) hipLaunchKernelGGL((k_prep_mm<true, false>), dim3((pair_hi - pair_lo + pb - 1) / pb), dim3(threads), lds, s, d, pair_lo, pair_hi, half, stg, mm_cap, cursor, (uint64_t *)gstage);
  else if (e6) hipLaunchKernelGGL((k_prep_mm<false, true>), dim3((pair_hi - pair_lo + pb - 1) / pb), dim3(threads), lds, s, d, pair_lo, pair_hi, half, stg, mm_cap, cursor, (uint64_t *)nullptr);
  else hipLaunchKernelGGL((k_prep_mm<false, false>), dim3((pair_hi - pair_lo + pb - 1) / pb), dim3(threads), lds, s, d, pair_lo, pair_hi, half, stg, mm_cap, cursor, (uint64_t *)nullptr);
}
// probe of the minimizers [range[0], range[1]) (device-side range), at most max_entries of them
static inline int probe_variant_norm(int variant) {
  if (variant == 0) variant = CM_PROBE_U;
  const int u = variant & 15;
  return ((u == 2 || u == 4 || u == 8) ? u : 1) | (variant & 16);
}
uint32_t cm_probe_range_blocks(uint64_t max_entries, int variant) {
  const uint64_t per = (uint64_t)(probe_variant_norm(variant) & 15) * CM_BLOCK;
  return (uint32_t)((max_entries + per - 1) / per);
}
void cm_launch_k_probe_range(const CmDev &d, const unsigned long long *range, uint64_t max_entries, uint32_t cap, void *partials, hipStream_t s, int variant) {
  const uint32_t blocks = cm_probe_range_blocks(max_entries, variant);
  if (!blocks) return;
  variant = probe_variant_norm(variant);
  const int u = variant & 15;
  const bool pair = (variant & 16) != 0;
#define CM_PROBE_CASE(U_, P_) hipLaunchKernelGGL((k_probe_range<U_, P_>), dim3(blocks), dim3(CM_BLOCK), 0, s, d.bkt, d.bmask, d.mm_hash, d.pr_val, d.pr_kind, range, cap, (uint2 *)partials)
  if (u == 1) { if (pair) CM_PROBE_CASE(1, true); else CM_PROBE_CASE(1, false); }
  else if (u == 2) { if (pair) CM_PROBE_CASE(2, true); else CM_PROBE_CASE(2, false); }
  else if (u == 8) { if (pair) CM_PROBE_CASE(8, true); else CM_PROBE_CASE(8, false); }
  else { if (pair) CM_PROBE_CASE(4, true); else CM_PROBE_CASE(4, false); }
#undef CM_PROBE_CASE
}
void cm_launch_k_probe_reduce(const void *partials, uint32_t blocks, unsigned long long *counters, hipStream_t s) {
  if (blocks) hipLaunchKernelGGL(k_probe_reduce, dim3(blocks / (CM_BLOCK * 8) + 1), dim3(CM_BLOCK), 0, s, (const uint2 *)partials, blocks, counters);
}
void cm_launch_k_prep_count(const CmDev &d, uint32_t n_pairs, uint32_t max_read_len, hipStream_t s) {
  if (!n_pairs) return;
  uint32_t threads, half;
  staging_geometry(max_read_len, &threads, &half);
  const uint32_t pb = threads / 2;
  hipLaunchKernelGGL(k_prep_count, dim3((n_pairs + pb - 1) / pb), dim3(threads), 2 * half, s, d, n_pairs, half);
}
// pairs [pair_lo, pair_hi)
void cm_launch_k_mm_fill(const CmDev &d, uint32_t pair_lo, uint32_t pair_hi, uint32_t max_read_len, hipStream_t s) {
  if (pair_hi <= pair_lo) return;
  uint32_t threads, half;
  staging_geometry(max_read_len, &threads, &half);
  const uint32_t pb = threads / 2;
  hipLaunchKernelGGL(k_mm_fill, dim3((pair_hi - pair_lo + pb - 1) / pb), dim3(threads), 2 * half, s, d, pair_lo, pair_hi, half);
}
// minimizer ranges of pair chunks from the scanned offsets: marks[ch] = mm_off[2 * lo[ch]] (mm_off has 2 n + 1 entries)
struct CmChunkLo { uint32_t lo[CM_MM_CHUNKS + 1]; };
__global__ void k_mm_marks(const uint32_t *__restrict__ mm_off, CmChunkLo cl, uint32_t n_marks, unsigned long long *__restrict__ marks) {
  if (threadIdx.x < n_marks) marks[threadIdx.x] = mm_off[2 * (size_t)cl.lo[threadIdx.x]];
}
void cm_launch_k_mm_marks(const uint32_t *mm_off, const uint32_t *lo, uint32_t n_marks, unsigned long long *marks, hipStream_t s) {
  CmChunkLo cl;
  for (uint32_t i = 0; i < n_marks && i <= CM_MM_CHUNKS; ++i) cl.lo[i] = lo[i];
  hipLaunchKernelGGL(k_mm_marks, dim3(1), dim3(64), 0, s, mm_off, cl, n_marks, marks);
}
void cm_launch_k_s0b_barcode(const CmDev &d, uint32_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_s0b_barcode, grid_for(n), dim3(CM_BLOCK), 0, s, d, n);
}
void cm_launch_k_bc_abundance(const uint8_t *bcb, const uint32_t *bco, uint32_t lo, uint32_t hi, uint64_t *wl, uint32_t wl_mask,
                              unsigned long long *num_sample, hipStream_t s) {
  if (hi > lo) hipLaunchKernelGGL(k_bc_abundance, grid_for(hi - lo), dim3(CM_BLOCK), 0, s, bcb, bco, lo, hi, wl, wl_mask, num_sample);
}
void cm_launch_k_s6b_sample(const CmDev &d, uint32_t n_chunks, hipStream_t s) {
  if (n_chunks) hipLaunchKernelGGL(k_s6b_sample, dim3(n_chunks), dim3(64), 0, s, d, n_chunks);
}
// partials: at least cm_stats_partial_words(n) unsigned long long
size_t cm_stats_partial_words(uint32_t n) { return (size_t)((n + CM_BLOCK - 1) / CM_BLOCK) * CM_NSTAT + CM_NSTAT; }
void cm_launch_k_stats(const CmDev &d, uint32_t n, unsigned long long *partials, hipStream_t s) {
  if (!n) return;
  uint32_t blocks = (n + CM_BLOCK - 1) / CM_BLOCK;
  if (blocks > CM_STATS_BLOCKS) blocks = CM_STATS_BLOCKS;
  hipLaunchKernelGGL(k_stats, dim3(blocks), dim3(CM_BLOCK), 0, s, d, n, partials);
  hipLaunchKernelGGL(k_stats_reduce, dim3(1), dim3(CM_BLOCK), 0, s, (const unsigned long long *)partials, blocks, d.stats);
}
// partials: one uint2 per block (cm_probe_partial_words(n) uint2), or nullptr to skip the accounting;
// counters[0] += probe steps, counters[1] += hits.  variant: lookups per lane (1, 2, 4, 8), + 16 to request the
// second probe step together with the first when both share a sector; 0 = the pipeline's setting.
size_t cm_probe_partial_words(uint32_t n) { return (size_t)((n + CM_BLOCK - 1) / CM_BLOCK) + 1; }
void cm_launch_k_probe(const uint64_t *bkt, uint32_t bmask, const uint64_t *hash, uint64_t *val, uint8_t *kind,
                       uint32_t n, void *partials, unsigned long long *counters, hipStream_t s, int variant) {
  if (!n) return;
  variant = probe_variant_norm(variant);
  const int u = variant & 15;
  const bool pair = (variant & 16) != 0;
  const uint32_t blocks = (n + u * CM_BLOCK - 1) / (u * CM_BLOCK);
#define CM_PROBE_CASE(U_, P_) hipLaunchKernelGGL((k_probe<U_, P_>), dim3(blocks), dim3(CM_BLOCK), 0, s, bkt, bmask, hash, val, kind, n, (uint2 *)partials)
  if (u == 1) { if (pair) CM_PROBE_CASE(1, true); else CM_PROBE_CASE(1, false); }
  else if (u == 2) { if (pair) CM_PROBE_CASE(2, true); else CM_PROBE_CASE(2, false); }
  else if (u == 8) { if (pair) CM_PROBE_CASE(8, true); else CM_PROBE_CASE(8, false); }
  else { if (pair) CM_PROBE_CASE(4, true); else CM_PROBE_CASE(4, false); }
#undef CM_PROBE_CASE
  if (partials && counters)
    hipLaunchKernelGGL(k_probe_reduce, dim3(blocks / (CM_BLOCK * 8) + 1), dim3(CM_BLOCK), 0, s, (const uint2 *)partials, blocks, counters);
}
