// cm_synth.hip -- synthetic genome, index construction and read generation on the device.
//
// Stands in for Index::Construct (index.cc:12-89) when a GRCh38-sized index is needed on a
// box that only has the GPU: collect the reference's minimizers (same state machine as
// MinimizerGenerator::GenerateMinimizers, run per chunk with a warm-up so that every
// emission is identical to a sequential pass), sort by (hash, hit) (index.cc:26), derive
// singleton / multi-occurrence entries and the occurrence table (index.cc:41-78), and insert
// the keys into an open-addressing table with khash's hash, triangular probing and sizing
// rule (khash.h:232-245, 310-318).  Lookup results are identical to an index built by
// the reference; only the bucket placement (insertion order) differs.
#include <hip/hip_runtime.h>
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <string>
#include <vector>

#include "../../include/chromap_amd.h"
#include "cm_ctx.h"
#include "cm_kernels.h"
#include "cm_stages.h"

#define SY_BLOCK 256
#define SY_CHUNK 2048u   // reference positions per thread
#define SY_WARM 96u      // warm-up positions in front of a chunk (>= 2w + k)

#define SYCHECK(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      cm_set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                \
      return CMGPU_EHIP;                                                                    \
    }                                                                                       \
  } while (0)

CM_HD uint64_t sy_mix(uint64_t x) {  // splitmix64 finalizer: counter-based RNG
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// base at global position g of the synthetic genome
CM_HD uint8_t sy_base(uint64_t seed, uint64_t g) {
  const uint64_t h = sy_mix(seed ^ ((g >> 5) * 0xD6E8FEB86659FD93ull));
  return (uint8_t)("ACGT"[(h >> (2 * (g & 31))) & 3]);
}

// Planted repeat families ("3 kb element x 600 copies, 2 % divergence" so that the seed-frequency
// thresholds, multi-mapper sampling and mate rescue fire): the genome is cut into n_families * copies equal slots,
// slot ci carries one copy of family ci % n_families at a pseudo-random offset, in either orientation, every base
// of the copy replaced by a random one with probability `divergence`.  O(1) per position, no tables.
struct SyRepeats {
  uint32_t n_families, copies, element_len, div_thresh;  // div_thresh = divergence * 2^32
  uint64_t slot;                                         // slot size in bases (0 = no repeats)
  uint32_t profile;                                      // 1, 2: the mosaic below instead
};
// one base of a planted element copy: family `fam` of the element kind `salt`, position `within` of `len`, orientation, and
// a per-base replacement with probability div_thresh / 2^32 (hc: the copy's hash)
CM_HD uint8_t sy_copy_base(uint64_t seed, uint64_t salt, uint32_t fam, uint32_t within, uint32_t len, bool rev, uint32_t div_thresh, uint64_t hc) {
  const uint32_t at = rev ? len - 1 - within : within;
  uint8_t b = sy_base(seed ^ salt ^ ((uint64_t)(fam + 1) << 40), at);
  if (rev) b = cm_negchar(b);
  const uint64_t hm = sy_mix(hc ^ ((uint64_t)within * 0x9E3779B97F4A7C15ull));
  if ((uint32_t)hm < div_thresh) b = (uint8_t)("ACGT"[(hm >> 32) & 3]);
  return b;
}
// Profile 1, a repeat landscape of the kind a mammalian genome has (what the verdict of round 2 asked the benchmark to face:
// >= 20 % of the bases in families of 10 .. 10^4 copies at 1 .. 15 % divergence, plus satellite runs).  The genome is tiled
// in 64-kb tiles; a tile's hash makes it
//   SINE-like   (20 % of the tiles): 300-base elements, one per 512-base cell at a random offset, 128 families -- on 3.1 Gb
//               about 9 500 copies per family -- every copy with its own divergence between 5 and 15 %;
//   LINE-like   (12 %): 3 000-base elements, one per 4 096-base cell, 256 families (~360 copies each), divergence 1 .. 5 %;
//   satellite   (2 %): the whole tile a tandem array of one of 64 units of 171 bases, 2 % of its bases replaced;
//   unique      (66 %).
// 11.7 % + 8.8 % + 2 % = 22.5 % of the bases are repeat-derived.  O(1) per position, no tables.
// Profile 2 (round 4: GRCh38 is about half repeat-derived): the same element kinds with 40 % SINE-like, 28 % LINE-like and 3 %
// satellite tiles -- 23.4 % + 20.5 % + 3 % = 47 % of the bases repeat-derived, ~19 000 copies per SINE-like family and ~830 per
// LINE-like family on 3.1 Gb.
#define SY_TILE 65536ull
CM_HD uint8_t sy_mosaic_base(uint64_t seed, uint64_t g, uint32_t profile) {
  const uint32_t t_sine = profile == 2 ? 40u : 20u, t_line = profile == 2 ? 68u : 32u, t_sat = profile == 2 ? 71u : 34u;
  const uint64_t tile = g / SY_TILE, in_tile = g % SY_TILE;
  const uint64_t th = sy_mix(seed ^ 0x7113D00DCAFEull ^ (tile * 0xA0761D6478BD642Full));
  const uint32_t kind = (uint32_t)(th % 100);
  if (kind < t_line) {
    const bool sine = kind < t_sine;
    const uint32_t cell_len = sine ? 512u : 4096u, elem = sine ? 300u : 3000u, n_fam = sine ? 128u : 256u;
    const uint64_t cell = tile * (SY_TILE / cell_len) + in_tile / cell_len;
    const uint64_t hc = sy_mix(seed ^ (sine ? 0x51AEull : 0x11AEull) ^ (cell * 0xC2B2AE3D27D4EB4Full));
    const uint32_t start = (uint32_t)(hc % (cell_len - elem)), at = (uint32_t)(in_tile % cell_len);
    if (at >= start && at < start + elem) {
      const uint32_t fam = (uint32_t)((hc >> 20) % n_fam);
      const bool rev = ((hc >> 40) & 1) != 0;
      // divergence of this copy: 5 .. 15 % (SINE-like) or 1 .. 5 %
      const uint64_t f = (hc >> 44) & 1023u;
      const uint32_t div = sine ? (uint32_t)(214748365ull + f * 419430ull) : (uint32_t)(42949673ull + f * 167772ull);
      return sy_copy_base(seed, sine ? 0x5EED51AEull : 0x5EED11AEull, fam, at - start, elem, rev, div, hc);
    }
    return sy_base(seed, g);
  }
  if (kind < t_sat) {
    const uint32_t fam = (uint32_t)((th >> 8) % 64);
    uint8_t b = sy_base(seed ^ 0x5A7E111EEull ^ ((uint64_t)(fam + 1) << 40), in_tile % 171u);
    const uint64_t hm = sy_mix(th ^ (in_tile * 0x9E3779B97F4A7C15ull));
    if ((uint32_t)hm < 85899346u) b = (uint8_t)("ACGT"[(hm >> 32) & 3]);  // 2 %
    return b;
  }
  return sy_base(seed, g);
}
CM_HD uint8_t sy_genome_base(uint64_t seed, uint64_t g, const SyRepeats &rp) {
  if (rp.profile) return sy_mosaic_base(seed, g, rp.profile);
  if (rp.slot) {
    const uint64_t ci = g / rp.slot;
    if (ci < (uint64_t)rp.n_families * rp.copies) {
      const uint64_t hc = sy_mix(seed ^ 0xA5A5A5A5F00DULL ^ (ci * 0xC2B2AE3D27D4EB4Full));
      const uint64_t start = ci * rp.slot + hc % (rp.slot - rp.element_len);
      if (g >= start && g < start + rp.element_len) {
        uint32_t within = (uint32_t)(g - start);
        const bool rev = ((hc >> 40) & 1) != 0;
        if (rev) within = rp.element_len - 1 - within;
        const uint32_t fam = (uint32_t)(ci % rp.n_families);
        uint8_t b = sy_base(seed ^ 0x5EEDFA11ull ^ ((uint64_t)(fam + 1) << 40), within);
        if (rev) b = cm_negchar(b);
        const uint64_t hm = sy_mix(hc ^ ((uint64_t)(g - start) * 0x9E3779B97F4A7C15ull));
        if ((uint32_t)hm < rp.div_thresh) b = (uint8_t)("ACGT"[(hm >> 32) & 3]);
        return b;
      }
    }
  }
  return sy_base(seed, g);
}

__global__ __launch_bounds__(SY_BLOCK) void k_sy_genome(uint8_t *ref, uint64_t ref_off, uint64_t gstart, uint32_t len,
                                                         uint64_t seed, SyRepeats rp) {
  const uint64_t i = (uint64_t)blockIdx.x * SY_BLOCK + threadIdx.x;
  if (i >= len) return;
  ref[ref_off + i] = sy_genome_base(seed, gstart + i, rp);
}

// ---------------------------------------------------------------------------------------
// reference minimizers, one chunk per thread.  mode 0: count, mode 1: fill.
// ---------------------------------------------------------------------------------------
struct SyChunk { uint32_t seq, start; };

__global__ __launch_bounds__(SY_BLOCK) void k_sy_ref_minimizers(const uint8_t *__restrict__ ref,
                                                                 const uint64_t *__restrict__ ref_off,
                                                                 const uint32_t *__restrict__ ref_len,
                                                                 const SyChunk *__restrict__ chunks, uint32_t n_chunks,
                                                                 int k, int w, int mode, uint32_t *__restrict__ cnt,
                                                                 const uint32_t *__restrict__ off,
                                                                 uint64_t *__restrict__ out_hash,
                                                                 uint64_t *__restrict__ out_hit) {
  const uint32_t c = blockIdx.x * SY_BLOCK + threadIdx.x;
  if (c >= n_chunks) return;
  const uint32_t rid = chunks[c].seq, s = chunks[c].start;
  const uint32_t n = cm_ref_chunk_minimizers(ref + ref_off[rid], ref_len[rid], rid, s, SY_CHUNK, SY_WARM, k, w,
                                             mode == 1 ? out_hash + off[c] : nullptr, mode == 1 ? out_hit + off[c] : nullptr);
  if (mode == 0) cnt[c] = n;
}

// entry i belongs to a multi-occurrence run
__global__ __launch_bounds__(SY_BLOCK) void k_sy_flag_multi(const uint64_t *__restrict__ hash, uint32_t n,
                                                             uint32_t *__restrict__ multi, uint32_t *__restrict__ is_start) {
  const uint32_t i = blockIdx.x * SY_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = hash[i];
  const bool same_prev = i > 0 && hash[i - 1] == h;
  const bool same_next = i + 1 < n && hash[i + 1] == h;
  multi[i] = (same_prev || same_next) ? 1u : 0u;
  is_start[i] = same_prev ? 0u : 1u;
}

// insert run starts into the table, scatter multi-occurrence hits into the occurrence table
__global__ __launch_bounds__(SY_BLOCK) void k_sy_insert(const uint64_t *__restrict__ hash, const uint64_t *__restrict__ hit,
                                                         uint32_t n, const uint32_t *__restrict__ multi,
                                                         const uint32_t *__restrict__ opos, uint64_t *__restrict__ bkt,
                                                         uint32_t bmask, uint64_t *__restrict__ occ,
                                                         unsigned long long *__restrict__ err) {
  const uint32_t i = blockIdx.x * SY_BLOCK + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = hash[i];
  const bool m = multi[i] != 0;
  if (m) occ[opos[i]] = hit[i];
  if (i > 0 && hash[i - 1] == h) return;  // not a run start
  uint64_t key, val;
  if (!m) {
    key = (h << 1) | 1ull;       // IsSingletonLookupKey (index_utils.h:56-58)
    val = hit[i];
  } else {
    uint32_t run = 1;
    while (i + run < n && hash[i + run] == h) ++run;
    key = h << 1;
    val = ((uint64_t)opos[i] << 32) | run;  // GenerateEntryValueInLookupTable (index_utils.h:28-31)
  }
  uint32_t b = (uint32_t)h & bmask, step = 0;
  const uint32_t first = b;
  for (;;) {
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(bkt + 2 * (uint64_t)b);
    const unsigned long long old = atomicCAS(slot, (unsigned long long)CM_EMPTY_KEY, (unsigned long long)key);
    if (old == (unsigned long long)CM_EMPTY_KEY) { bkt[2 * (uint64_t)b + 1] = val; return; }
    b = (b + (++step)) & bmask;
    if (b == first) { atomicAdd(err, 1ull); return; }
  }
}

__global__ __launch_bounds__(SY_BLOCK) void k_sy_fill_empty(uint64_t *bkt, uint64_t n_words) {
  const uint64_t i = (uint64_t)blockIdx.x * SY_BLOCK + threadIdx.x;
  if (i < n_words) bkt[i] = (i & 1) ? 0ull : CM_EMPTY_KEY;
}

// khash sizing: kh_put grows the table while n_occupied >= upper_bound = (uint32)(nb*0.77+0.5)
// before an insertion (khash.h:310-318); the final n_buckets is the smallest power of two
// >= 4 whose upper bound exceeds n_keys - 1.
static uint32_t sy_buckets_for(uint64_t n_keys) {
  uint64_t nb = 4;
  while (n_keys > 0 && (uint64_t)((double)nb * 0.77 + 0.5) <= n_keys - 1) nb <<= 1;
  return nb > 0x80000000ull ? 0 : (uint32_t)nb;
}

static int sy_build_index(cmgpu_ctx *c) {
  hipStream_t s = c->stream;
  const int k = c->p.k, w = c->p.w;
  // ---- chunk list
  std::vector<SyChunk> chunks;
  for (uint32_t r = 0; r < c->n_seq; ++r)
    for (uint64_t st = 0; st < c->h_ref_len[r]; st += SY_CHUNK) chunks.push_back({r, (uint32_t)st});
  const uint32_t nch = (uint32_t)chunks.size();
  if (nch == 0) { cm_set_error(c, "empty reference"); return CMGPU_EINVAL; }
  DevBuf d_chunks, d_cnt, d_off, d_tmp;
  if (d_chunks.ensure((size_t)nch * sizeof(SyChunk)) || d_cnt.ensure(((size_t)nch + 1) * 4) || d_off.ensure(((size_t)nch + 1) * 4) ||
      d_tmp.ensure(cm_scan_tmp_words(nch) * 4)) { cm_set_error(c, "out of device memory (index build)"); return CMGPU_ENOMEM; }
  SYCHECK(c, hipMemcpyAsync(d_chunks.p, chunks.data(), (size_t)nch * sizeof(SyChunk), hipMemcpyHostToDevice, s));
  const dim3 g((nch + SY_BLOCK - 1) / SY_BLOCK), b(SY_BLOCK);
  hipLaunchKernelGGL(k_sy_ref_minimizers, g, b, 0, s, (const uint8_t *)c->ref.p, (const uint64_t *)c->ref_off.p,
                     (const uint32_t *)c->ref_len.p, (const SyChunk *)d_chunks.p, nch, k, w, 0, (uint32_t *)d_cnt.p,
                     (const uint32_t *)nullptr, (uint64_t *)nullptr, (uint64_t *)nullptr);
  cm_scan_u32((const uint32_t *)d_cnt.p, (uint32_t *)d_off.p, nch, (uint32_t *)d_tmp.p, s);
  uint32_t n_mm = 0;
  SYCHECK(c, hipMemcpyAsync(&n_mm, (uint32_t *)d_off.p + nch, 4, hipMemcpyDeviceToHost, s));
  SYCHECK(c, cm_stream_sync(s));
  if (n_mm == 0) { cm_set_error(c, "reference has no minimizers"); return CMGPU_EINVAL; }
  if (n_mm > 0x7fffffffu) { cm_set_error(c, "more than INT_MAX minimizers (index.cc:33)"); return CMGPU_ECAPACITY; }
  DevBuf h0, t0, h1, t1;
  if (h0.ensure((size_t)n_mm * 8) || t0.ensure((size_t)n_mm * 8) || h1.ensure((size_t)n_mm * 8) || t1.ensure((size_t)n_mm * 8)) {
    cm_set_error(c, "out of device memory (minimizer arrays)"); return CMGPU_ENOMEM;
  }
  hipLaunchKernelGGL(k_sy_ref_minimizers, g, b, 0, s, (const uint8_t *)c->ref.p, (const uint64_t *)c->ref_off.p,
                     (const uint32_t *)c->ref_len.p, (const SyChunk *)d_chunks.p, nch, k, w, 1, (uint32_t *)d_cnt.p,
                     (const uint32_t *)d_off.p, (uint64_t *)h0.p, (uint64_t *)t0.p);
  // ---- sort by (hash, hit): LSD -- stable sort by hit, then stable sort by hash
  size_t tb = 0;
  SYCHECK(c, rocprim::radix_sort_pairs(nullptr, tb, (uint64_t *)t0.p, (uint64_t *)t1.p, (uint64_t *)h0.p, (uint64_t *)h1.p,
                                       (size_t)n_mm, 0, 64, s));
  DevBuf sort_tmp;
  if (sort_tmp.ensure(tb + 256)) { cm_set_error(c, "out of device memory (sort)"); return CMGPU_ENOMEM; }
  SYCHECK(c, rocprim::radix_sort_pairs(sort_tmp.p, tb, (uint64_t *)t0.p, (uint64_t *)t1.p, (uint64_t *)h0.p, (uint64_t *)h1.p,
                                       (size_t)n_mm, 0, 64, s));
  size_t tb2 = tb;
  SYCHECK(c, rocprim::radix_sort_pairs(nullptr, tb2, (uint64_t *)h1.p, (uint64_t *)h0.p, (uint64_t *)t1.p, (uint64_t *)t0.p,
                                       (size_t)n_mm, 0, 2 * k, s));
  if (sort_tmp.ensure(tb2 + 256)) { cm_set_error(c, "out of device memory (sort)"); return CMGPU_ENOMEM; }
  SYCHECK(c, rocprim::radix_sort_pairs(sort_tmp.p, tb2, (uint64_t *)h1.p, (uint64_t *)h0.p, (uint64_t *)t1.p, (uint64_t *)t0.p,
                                       (size_t)n_mm, 0, 2 * k, s));
  SYCHECK(c, cm_stream_sync(s));
  sort_tmp.release(); h1.release(); t1.release();
  // ---- singleton / multi flags, occurrence offsets, key count
  DevBuf multi, starts, opos, spos, tmp2;
  if (multi.ensure(((size_t)n_mm + 1) * 4) || starts.ensure(((size_t)n_mm + 1) * 4) || opos.ensure(((size_t)n_mm + 1) * 4) ||
      spos.ensure(((size_t)n_mm + 1) * 4) || tmp2.ensure(cm_scan_tmp_words(n_mm) * 4)) {
    cm_set_error(c, "out of device memory (index flags)"); return CMGPU_ENOMEM;
  }
  const dim3 gm((n_mm + SY_BLOCK - 1) / SY_BLOCK);
  hipLaunchKernelGGL(k_sy_flag_multi, gm, b, 0, s, (const uint64_t *)h0.p, n_mm, (uint32_t *)multi.p, (uint32_t *)starts.p);
  cm_scan_u32((const uint32_t *)multi.p, (uint32_t *)opos.p, n_mm, (uint32_t *)tmp2.p, s);
  cm_scan_u32((const uint32_t *)starts.p, (uint32_t *)spos.p, n_mm, (uint32_t *)tmp2.p, s);
  uint32_t n_occ = 0, n_keys = 0;
  SYCHECK(c, hipMemcpyAsync(&n_occ, (uint32_t *)opos.p + n_mm, 4, hipMemcpyDeviceToHost, s));
  SYCHECK(c, hipMemcpyAsync(&n_keys, (uint32_t *)spos.p + n_mm, 4, hipMemcpyDeviceToHost, s));
  SYCHECK(c, cm_stream_sync(s));
  spos.release(); starts.release();
  const uint32_t nb = sy_buckets_for(n_keys);
  if (nb == 0) { cm_set_error(c, "too many distinct minimizers for a 32-bit khash"); return CMGPU_ECAPACITY; }
  if (c->bkt.ensure((size_t)nb * 16) || c->occ.ensure((size_t)(n_occ ? n_occ : 1) * 8)) {
    cm_set_error(c, "out of device memory (index table)"); return CMGPU_ENOMEM;
  }
  const uint64_t words = (uint64_t)nb * 2;
  hipLaunchKernelGGL(k_sy_fill_empty, dim3((unsigned)((words + SY_BLOCK - 1) / SY_BLOCK)), b, 0, s, (uint64_t *)c->bkt.p, words);
  SYCHECK(c, hipMemsetAsync(c->stats.p, 0, CM_ST_N * 8, s));
  hipLaunchKernelGGL(k_sy_insert, gm, b, 0, s, (const uint64_t *)h0.p, (const uint64_t *)t0.p, n_mm, (const uint32_t *)multi.p,
                     (const uint32_t *)opos.p, (uint64_t *)c->bkt.p, nb - 1, (uint64_t *)c->occ.p,
                     (unsigned long long *)c->stats.p);
  unsigned long long err = 0;
  SYCHECK(c, hipMemcpyAsync(&err, c->stats.p, 8, hipMemcpyDeviceToHost, s));
  SYCHECK(c, cm_stream_sync(s));
  if (err) { cm_set_error(c, "hash table overflow during index build"); return CMGPU_ECAPACITY; }
  c->bmask = nb - 1;
  c->n_occ = n_occ;
  c->synth_n_minimizers = n_mm;
  c->synth_n_keys = n_keys;
  d_chunks.release(); d_cnt.release(); d_off.release(); d_tmp.release(); h0.release(); t0.release();
  multi.release(); opos.release(); tmp2.release();
  return CMGPU_OK;
}

extern "C" int cmgpu_create_synthetic(uint64_t total_bases, uint32_t n_sequences, uint64_t seed, int32_t kmer_size,
                                      int32_t window_size, const cmgpu_params *params, int device_id, cmgpu_ctx **out) {
  return cmgpu_create_synthetic_repeats(total_bases, n_sequences, seed, kmer_size, window_size, params, device_id, 0, 0, 0, 0.0, out);
}

extern "C" int cmgpu_create_synthetic_profile(uint64_t total_bases, uint32_t n_sequences, uint64_t seed, int32_t kmer_size, int32_t window_size,
                                              const cmgpu_params *params, int device_id, uint32_t profile, cmgpu_ctx **out) {
  return cmgpu_create_synthetic_repeats(total_bases, n_sequences, seed, kmer_size, window_size, params, device_id, 0xffffffffu, profile, 0, 0.0, out);
}

extern "C" int cmgpu_create_synthetic_repeats(uint64_t total_bases, uint32_t n_sequences, uint64_t seed, int32_t kmer_size,
                                              int32_t window_size, const cmgpu_params *params, int device_id, uint32_t n_families,
                                              uint32_t copies, uint32_t element_len, double divergence, cmgpu_ctx **out) {
  if (!params || !out || n_sequences == 0 || total_bases < n_sequences * 1000ull) { cm_set_error(nullptr, "bad argument"); return CMGPU_EINVAL; }
  SyRepeats rp = {0, 0, 0, 0, 0, 0};
  if (n_families == 0xffffffffu) {  // cmgpu_create_synthetic_profile
    rp.profile = copies;
    if (rp.profile != 1 && rp.profile != 2) { cm_set_error(nullptr, "unknown synthetic genome profile"); return CMGPU_EINVAL; }
  } else if (n_families && copies && element_len) {
    rp.n_families = n_families; rp.copies = copies; rp.element_len = element_len;
    rp.div_thresh = (uint32_t)(divergence * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (divergence < 0 ? 0 : divergence * 4294967296.0));
    rp.slot = total_bases / ((uint64_t)n_families * copies);
    if (rp.slot < 2ull * element_len) { cm_set_error(nullptr, "repeat copies do not fit the genome (slot < 2 x element length)"); return CMGPU_EINVAL; }
  }
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { cm_set_error(nullptr, "no HIP device available (this library has no CPU path)"); return CMGPU_ENODEVICE; }
  if (device_id < 0 || device_id >= n || hipSetDevice(device_id) != hipSuccess) { cm_set_error(nullptr, "bad device"); return CMGPU_EINVAL; }
  (void)hipGetLastError();
  cmgpu_ctx *c = new cmgpu_ctx();
  int rc = cm_ctx_init_common(c, params, kmer_size, window_size, device_id);
  if (rc) { cm_set_error(nullptr, c->err); cmgpu_destroy(c); return rc; }
  // chromosome lengths: linear spread, largest three times the smallest (as tools/gen_synth.py)
  c->n_seq = n_sequences;
  c->h_ref_len.resize(n_sequences);
  c->goff_tried = false; c->goff.release();  // (the 32-bit key offsets follow the reference's lengths)
  c->h_ref_off.resize(n_sequences);
  double wsum = 0;
  for (uint32_t i = 0; i < n_sequences; ++i) wsum += n_sequences > 1 ? 3.0 - 2.0 * i / (n_sequences - 1) : 1.0;
  uint64_t tot = 64;
  std::vector<uint64_t> gstart(n_sequences);
  uint64_t gs = 0;
  for (uint32_t i = 0; i < n_sequences; ++i) {
    const double wi = n_sequences > 1 ? 3.0 - 2.0 * i / (n_sequences - 1) : 1.0;
    uint64_t len = (uint64_t)(wi / wsum * (double)total_bases);
    if (len < 1000) len = 1000;
    if (len > 0xfffffff0ull) { cm_set_error(nullptr, "sequence longer than 2^32"); cmgpu_destroy(c); return CMGPU_EINVAL; }
    c->h_ref_len[i] = (uint32_t)len;
    c->h_ref_off[i] = tot;
    gstart[i] = gs;
    gs += len;
    tot += len + 64;
    tot = (tot + 15) & ~15ull;
  }
  c->ref_bytes = tot;
  if (c->ref.ensure(tot) || c->ref_off.ensure((size_t)n_sequences * 8) || c->ref_len.ensure((size_t)n_sequences * 4)) {
    cm_set_error(nullptr, "out of device memory (reference)"); cmgpu_destroy(c); return CMGPU_ENOMEM;
  }
  hipError_t e = hipMemsetAsync(c->ref.p, 0, tot, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->ref_off.p, c->h_ref_off.data(), (size_t)n_sequences * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->ref_len.p, c->h_ref_len.data(), (size_t)n_sequences * 4, hipMemcpyHostToDevice, c->stream);
  for (uint32_t i = 0; e == hipSuccess && i < n_sequences; ++i) {
    const uint32_t len = c->h_ref_len[i];
    hipLaunchKernelGGL(k_sy_genome, dim3((len + SY_BLOCK - 1) / SY_BLOCK), dim3(SY_BLOCK), 0, c->stream, (uint8_t *)c->ref.p,
                       c->h_ref_off[i], gstart[i], len, seed, rp);
  }
  if (e == hipSuccess) e = cm_stream_sync(c->stream);
  if (e != hipSuccess) { cm_set_error(nullptr, std::string("genome generation: ") + hipGetErrorString(e)); cmgpu_destroy(c); return CMGPU_EHIP; }
  rc = sy_build_index(c);
  if (rc) { cm_set_error(nullptr, c->err); cmgpu_destroy(c); return rc; }
  if (cm_build_fast_table(c, 1) != CMGPU_OK) { c->bkt_fast.release(); c->fmask = 0; c->err.clear(); }  // (see cmgpu_create)
  *out = c;
  return CMGPU_OK;
}

// Index::Construct (index.cc:12-89) on the device for a caller-provided reference: the same
// builder as the synthetic genome's.  Lookups are identical to an index built by the
// reference (same keys, values, occurrence order); only the bucket an entry lands in may
// differ where khash's resize history would have placed it elsewhere along its probe path.
extern "C" int cmgpu_create_from_reference(const cmgpu_ref_view *ref, int32_t kmer_size, int32_t window_size,
                                           const cmgpu_params *params, int device_id, cmgpu_ctx **out) {
  if (!ref || !params || !out || ref->n_sequences == 0) { cm_set_error(nullptr, "bad argument"); return CMGPU_EINVAL; }
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { cm_set_error(nullptr, "no HIP device available (this library has no CPU path)"); return CMGPU_ENODEVICE; }
  if (device_id < 0 || device_id >= n || hipSetDevice(device_id) != hipSuccess) { cm_set_error(nullptr, "bad device"); return CMGPU_EINVAL; }
  (void)hipGetLastError();
  cmgpu_ctx *c = new cmgpu_ctx();
  int rc = cm_ctx_init_common(c, params, kmer_size, window_size, device_id);
  if (rc == CMGPU_OK) rc = cm_upload_reference(c, ref);
  if (rc == CMGPU_OK) rc = sy_build_index(c);
  if (rc) { cm_set_error(nullptr, c->err); cmgpu_destroy(c); return rc; }
  if (cm_build_fast_table(c, 1) != CMGPU_OK) { c->bkt_fast.release(); c->fmask = 0; c->err.clear(); }  // (see cmgpu_create)
  *out = c;
  return CMGPU_OK;
}

// Index::Save (index.cc:91-121) of the resident index: kmer size, window size, key count,
// kh_save's {n_buckets, size, n_occupied, upper_bound, flags, keys, vals} (khash.h:222-233),
// occurrence count and table.  Empty buckets are written as zero key/value.
extern "C" int cmgpu_save_index_file(cmgpu_ctx *c, const char *path) {
  if (!c || !path) return CMGPU_EINVAL;
  const uint32_t nb = c->bmask + 1;
  std::vector<uint64_t> bk((size_t)nb * 2), occ(c->n_occ);
  int rc = cmgpu_export_index(c, bk.data(), occ.data());
  if (rc) return rc;
  const size_t fw = nb < 16 ? 1 : nb >> 4;
  std::vector<uint32_t> flags(fw, 0xaaaaaaaau);  // every bucket "empty" (khash.h:155-161)
  std::vector<uint64_t> keys(nb, 0), vals(nb, 0);
  uint32_t nk = 0;
  for (uint32_t i = 0; i < nb; ++i) {
    if (bk[2 * (size_t)i] == CM_EMPTY_KEY) continue;
    keys[i] = bk[2 * (size_t)i];
    vals[i] = bk[2 * (size_t)i + 1];
    flags[i >> 4] &= ~(3u << ((i & 15) << 1));
    ++nk;
  }
  FILE *f = fopen(path, "wb");
  if (!f) { cm_set_error(c, std::string("cannot open ") + path); return CMGPU_EIO; }
  const int32_t k = c->p.k, w = c->p.w;
  const uint32_t hdr[4] = {nb, nk, nk, (uint32_t)((double)nb * 0.77 + 0.5)};
  bool ok = fwrite(&k, 4, 1, f) == 1 && fwrite(&w, 4, 1, f) == 1 && fwrite(&nk, 4, 1, f) == 1 && fwrite(hdr, 4, 4, f) == 4 &&
            fwrite(flags.data(), 4, fw, f) == fw && fwrite(keys.data(), 8, nb, f) == nb && fwrite(vals.data(), 8, nb, f) == nb &&
            fwrite(&c->n_occ, 4, 1, f) == 1 && (c->n_occ == 0 || fwrite(occ.data(), 8, c->n_occ, f) == c->n_occ);
  ok = fclose(f) == 0 && ok;
  if (!ok) { cm_set_error(c, std::string("short write to ") + path); return CMGPU_EIO; }
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// synthetic read pairs drawn from the resident reference: fragment start
// uniform, length uniform in [frag_min, frag_max), R1 = fragment head, R2 = head of the
// reverse complement, adapter read-through when the fragment is shorter than the read,
// substitutions with probability sub_rate, mates swapped with p = 1/2.
// ---------------------------------------------------------------------------------------
#define SY_ADAPTER_LEN 66
__constant__ char sy_adapter1[SY_ADAPTER_LEN + 1] = "CTGTCTCTTATACACATCTCCGAGCCCACGAGACTAAGGCGAATCTCGTATGCCGTCTTCTGCTTG";
__constant__ char sy_adapter2[SY_ADAPTER_LEN + 1] = "CTGTCTCTTATACACATCTGACGCTGCCGACGAGTGTAGATCTCGGTGGTCGCCGTATCATTAAAA";

__global__ __launch_bounds__(SY_BLOCK) void k_sy_reads(const uint8_t *__restrict__ ref, const uint64_t *__restrict__ ref_off,
                                                        const uint32_t *__restrict__ ref_len, uint32_t n_seq,
                                                        uint64_t total_len, uint32_t n_pairs, uint32_t L, uint32_t fmin,
                                                        uint32_t fmax, uint32_t sub_thresh, uint32_t indel_thresh, uint64_t seed,
                                                        uint8_t *__restrict__ r1, uint8_t *__restrict__ r2,
                                                        uint32_t *__restrict__ o1, uint32_t *__restrict__ o2) {
  const uint32_t i = blockIdx.x * SY_BLOCK + threadIdx.x;
  if (i > n_pairs) return;
  if (i == n_pairs) { o1[i] = i * L; o2[i] = i * L; return; }
  o1[i] = i * L;
  o2[i] = i * L;
  uint64_t ctr = sy_mix(seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull));
  uint64_t g = sy_mix(ctr) % total_len;
  uint32_t rid = 0;
  while (rid + 1 < n_seq && g >= ref_len[rid]) { g -= ref_len[rid]; ++rid; }
  const uint32_t len = ref_len[rid];
  uint32_t fl = fmin + (uint32_t)(sy_mix(ctr + 1) % (uint64_t)(fmax > fmin ? fmax - fmin : 1));
  if (fl > len) fl = len;
  uint32_t st = (uint32_t)g;
  if (st + fl > len) st = len - fl;
  const uint8_t *frag = ref + ref_off[rid] + st;
  const bool swap = (sy_mix(ctr + 2) & 1) != 0;
  uint8_t *a = (swap ? r2 : r1) + (uint64_t)i * L;  // receives the forward head
  uint8_t *b = (swap ? r1 : r2) + (uint64_t)i * L;  // receives the reverse-complement head
  uint64_t rs = ctr + 3;
  // fa / fb: next fragment base of the forward / reverse-complement read; a 1-base insertion emits a random base
  // without consuming one, a 1-base deletion skips one ("+0.1 % 1-bp indels for config 5")
  uint32_t fa = 0, fb = 0;
  for (uint32_t j = 0; j < L; ++j) {
    const uint64_t h1 = sy_mix(rs + 2 * j), h2 = sy_mix(rs + 2 * j + 1);
    uint8_t x, y;
    if (indel_thresh) {
      const uint64_t g1 = sy_mix(rs + 0x10000 + 2 * j), g2 = sy_mix(rs + 0x10001 + 2 * j);
      const bool ev1 = (uint32_t)g1 < indel_thresh, ev2 = (uint32_t)g2 < indel_thresh;
      const bool ins1 = ev1 && ((g1 >> 40) & 1), ins2 = ev2 && ((g2 >> 40) & 1);
      if (ev1 && !ins1) ++fa;
      if (ev2 && !ins2) ++fb;
      x = ins1 ? (uint8_t)("ACGT"[(g1 >> 32) & 3]) : (fa < fl ? frag[fa] : (uint8_t)sy_adapter1[(fa - fl) % SY_ADAPTER_LEN]);
      y = ins2 ? (uint8_t)("ACGT"[(g2 >> 32) & 3]) : (fb < fl ? cm_negchar(frag[fl - 1 - fb]) : (uint8_t)sy_adapter2[(fb - fl) % SY_ADAPTER_LEN]);
      if (!ins1) ++fa;
      if (!ins2) ++fb;
    } else {
      x = j < fl ? frag[j] : (uint8_t)sy_adapter1[(j - fl) % SY_ADAPTER_LEN];
      y = j < fl ? cm_negchar(frag[fl - 1 - j]) : (uint8_t)sy_adapter2[(j - fl) % SY_ADAPTER_LEN];
    }
    if ((uint32_t)(h1 & 0xffffffffu) < sub_thresh) x = (uint8_t)("ACGT"[(h1 >> 32) & 3]);
    if ((uint32_t)(h2 & 0xffffffffu) < sub_thresh) y = (uint8_t)("ACGT"[(h2 >> 32) & 3]);
    a[j] = x;
    b[j] = y;
  }
}

// Hi-C shaped pairs (BASELINE config 5): the two mates come from two INDEPENDENT loci -- 60 % of the pairs within 1 Mb on one
// sequence (cis contacts), the others anywhere in the genome -- each in either orientation, and chim_thresh / 2^32 of the pairs
// carry a ligation junction inside one read: its first `cut` bases (25 .. L - 25) from its own locus, the rest the reverse
// complement of the partner's fragment, which is what split alignment (draft_mapping_generator.cc:410-487) exists for.
// Same substitutions / 1-base indels as k_sy_reads.
__global__ __launch_bounds__(SY_BLOCK) void k_sy_reads_hic(const uint8_t *__restrict__ ref, const uint64_t *__restrict__ ref_off,
                                                            const uint32_t *__restrict__ ref_len, uint32_t n_seq, uint64_t total_len,
                                                            uint32_t n_pairs, uint32_t L, uint32_t sub_thresh, uint32_t indel_thresh,
                                                            uint32_t chim_thresh, uint64_t seed, uint8_t *__restrict__ r1,
                                                            uint8_t *__restrict__ r2, uint32_t *__restrict__ o1, uint32_t *__restrict__ o2) {
  const uint32_t i = blockIdx.x * SY_BLOCK + threadIdx.x;
  if (i > n_pairs) return;
  o1[i] = i * L;
  o2[i] = i * L;
  if (i == n_pairs) return;
  const uint32_t F = 2 * L;  // fragment kept at either locus
  const uint64_t ctr = sy_mix(seed ^ 0x41C0FFEEull ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull));
  // locus A
  uint64_t g = sy_mix(ctr) % total_len;
  uint32_t ra = 0;
  while (ra + 1 < n_seq && g >= ref_len[ra]) { g -= ref_len[ra]; ++ra; }
  const uint32_t la = ref_len[ra];
  uint32_t sa = (uint32_t)g;
  if (sa + F > la) sa = la > F ? la - F : 0;
  // locus B: cis within 1 Mb, or anywhere
  uint32_t rb = ra, sb;
  const uint64_t hb = sy_mix(ctr + 1);
  if ((hb & 1023u) < 614u) {
    const int64_t d = (int64_t)((hb >> 10) % 2000001ull) - 1000000;
    int64_t p = (int64_t)sa + d;
    if (p < 0) p = 0;
    if (p + (int64_t)F > (int64_t)la) p = la > F ? (int64_t)la - F : 0;
    sb = (uint32_t)p;
  } else {
    uint64_t g2 = sy_mix(ctr + 2) % total_len;
    rb = 0;
    while (rb + 1 < n_seq && g2 >= ref_len[rb]) { g2 -= ref_len[rb]; ++rb; }
    sb = (uint32_t)g2;
    if (sb + F > ref_len[rb]) sb = ref_len[rb] > F ? ref_len[rb] - F : 0;
  }
  const uint8_t *fa_ = ref + ref_off[ra] + sa, *fb_ = ref + ref_off[rb] + sb;
  const bool oa = (hb >> 40) & 1, ob = (hb >> 41) & 1;  // orientation of either fragment
  auto SA = [&](uint32_t j) -> uint8_t { return oa ? cm_negchar(fa_[F - 1 - j]) : fa_[j]; };
  auto SB = [&](uint32_t j) -> uint8_t { return ob ? cm_negchar(fb_[F - 1 - j]) : fb_[j]; };
  const uint64_t hc = sy_mix(ctr + 3);
  const bool chim = (uint32_t)hc < chim_thresh && L >= 52;
  const uint32_t cut = chim ? 25u + (uint32_t)((hc >> 32) % (uint64_t)(L - 50 + 1)) : L;
  const bool chim_on_a = ((hc >> 62) & 1) != 0;
  const bool swap = (sy_mix(ctr + 4) & 1) != 0;
  uint8_t *a = (swap ? r2 : r1) + (uint64_t)i * L;
  uint8_t *b = (swap ? r1 : r2) + (uint64_t)i * L;
  const uint64_t rs = ctr + 5;
  uint32_t ia = 0, ib = 0;  // next clean base of either read (1-base indels as in k_sy_reads)
  for (uint32_t j = 0; j < L; ++j) {
    const uint64_t h1 = sy_mix(rs + 2 * j), h2 = sy_mix(rs + 2 * j + 1);
    bool ins1 = false, ins2 = false;
    uint64_t g1 = 0, g2 = 0;
    if (indel_thresh) {
      g1 = sy_mix(rs + 0x10000 + 2 * j); g2 = sy_mix(rs + 0x10001 + 2 * j);
      const bool ev1 = (uint32_t)g1 < indel_thresh, ev2 = (uint32_t)g2 < indel_thresh;
      ins1 = ev1 && ((g1 >> 40) & 1); ins2 = ev2 && ((g2 >> 40) & 1);
      if (ev1 && !ins1) ++ia;
      if (ev2 && !ins2) ++ib;
    }
    const uint32_t ka = ia < F ? ia : F - 1, kb = ib < F ? ib : F - 1;
    // clean base k of read a / b: own fragment up to the junction, then the partner's fragment read backwards on the other strand
    uint8_t x = (chim && chim_on_a && ka >= cut) ? cm_negchar(SB(F - 1 - (ka - cut))) : SA(ka);
    uint8_t y = (chim && !chim_on_a && kb >= cut) ? cm_negchar(SA(F - 1 - (kb - cut))) : SB(kb);
    if (ins1) x = (uint8_t)("ACGT"[(g1 >> 32) & 3]); else ++ia;
    if (ins2) y = (uint8_t)("ACGT"[(g2 >> 32) & 3]); else ++ib;
    if ((uint32_t)(h1 & 0xffffffffu) < sub_thresh) x = (uint8_t)("ACGT"[(h1 >> 32) & 3]);
    if ((uint32_t)(h2 & 0xffffffffu) < sub_thresh) y = (uint8_t)("ACGT"[(h2 >> 32) & 3]);
    a[j] = x;
    b[j] = y;
  }
}

extern "C" int cmgpu_generate_resident_batch_hic(cmgpu_ctx *c, uint32_t n_pairs, uint32_t read_length, double sub_rate, double indel_rate,
                                                 double chimeric_fraction, uint64_t seed) {
  if (!c || read_length < 30 || read_length > 250 || (uint64_t)n_pairs * read_length > 0xfffffff0ull) { cm_set_error(c, "bad argument"); return CMGPU_EINVAL; }
  SYCHECK(c, cm_enter(c));
  for (uint32_t i = 0; i < c->n_seq; ++i)
    if (c->h_ref_len[i] < 2 * read_length) { cm_set_error(c, "a sequence is shorter than two read lengths"); return CMGPU_EINVAL; }
  c->n_pairs = n_pairs;
  c->first_read_id = 0;
  c->bases0 = c->bases1 = (size_t)n_pairs * read_length;
  c->max_read_len = read_length;
  c->has_barcodes = false;
  c->single = false;
  if (c->rb0.ensure(c->bases0 + 16) || c->rb1.ensure(c->bases1 + 16) || c->ro0.ensure(((size_t)n_pairs + 1) * 4) ||
      c->ro1.ensure(((size_t)n_pairs + 1) * 4)) { cm_set_error(c, "out of device memory (reads)"); return CMGPU_ENOMEM; }
  uint64_t total = 0;
  for (uint32_t i = 0; i < c->n_seq; ++i) total += c->h_ref_len[i];
  auto thresh = [](double r) { return (uint32_t)(r * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (r < 0 ? 0 : r * 4294967296.0)); };
  hipLaunchKernelGGL(k_sy_reads_hic, dim3((n_pairs + 1 + SY_BLOCK - 1) / SY_BLOCK), dim3(SY_BLOCK), 0, c->stream,
                     (const uint8_t *)c->ref.p, (const uint64_t *)c->ref_off.p, (const uint32_t *)c->ref_len.p, c->n_seq, total,
                     n_pairs, read_length, thresh(sub_rate), thresh(indel_rate), thresh(chimeric_fraction), seed, (uint8_t *)c->rb0.p,
                     (uint8_t *)c->rb1.p, (uint32_t *)c->ro0.p, (uint32_t *)c->ro1.p);
  SYCHECK(c, cm_stream_sync(c->stream));
  return CMGPU_OK;
}

extern "C" int cmgpu_generate_resident_batch(cmgpu_ctx *c, uint32_t n_pairs, uint32_t read_length, uint32_t frag_min,
                                             uint32_t frag_max, double sub_rate, uint64_t seed) {
  return cmgpu_generate_resident_batch_indels(c, n_pairs, read_length, frag_min, frag_max, sub_rate, 0.0, seed);
}

extern "C" int cmgpu_generate_resident_batch_indels(cmgpu_ctx *c, uint32_t n_pairs, uint32_t read_length, uint32_t frag_min,
                                                    uint32_t frag_max, double sub_rate, double indel_rate, uint64_t seed) {
  if (!c || read_length == 0 || read_length > 250 || frag_min == 0 || (uint64_t)n_pairs * read_length > 0xfffffff0ull) {
    cm_set_error(c, "bad argument");
    return CMGPU_EINVAL;
  }
  SYCHECK(c, cm_enter(c));
  c->n_pairs = n_pairs;
  c->first_read_id = 0;
  c->bases0 = c->bases1 = (size_t)n_pairs * read_length;
  c->max_read_len = read_length;
  c->has_barcodes = false;
  c->single = false;
  if (c->rb0.ensure(c->bases0 + 16) || c->rb1.ensure(c->bases1 + 16) || c->ro0.ensure(((size_t)n_pairs + 1) * 4) ||
      c->ro1.ensure(((size_t)n_pairs + 1) * 4)) { cm_set_error(c, "out of device memory (reads)"); return CMGPU_ENOMEM; }
  uint64_t total = 0;
  for (uint32_t i = 0; i < c->n_seq; ++i) total += c->h_ref_len[i];
  const uint32_t thr = (uint32_t)(sub_rate * 4294967296.0 > 4294967295.0 ? 4294967295.0 : sub_rate * 4294967296.0);
  const uint32_t ithr = (uint32_t)(indel_rate * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (indel_rate < 0 ? 0 : indel_rate * 4294967296.0));
  hipLaunchKernelGGL(k_sy_reads, dim3((n_pairs + 1 + SY_BLOCK - 1) / SY_BLOCK), dim3(SY_BLOCK), 0, c->stream,
                     (const uint8_t *)c->ref.p, (const uint64_t *)c->ref_off.p, (const uint32_t *)c->ref_len.p, c->n_seq, total,
                     n_pairs, read_length, frag_min, frag_max, thr, ithr, seed, (uint8_t *)c->rb0.p, (uint8_t *)c->rb1.p,
                     (uint32_t *)c->ro0.p, (uint32_t *)c->ro1.p);
  SYCHECK(c, cm_stream_sync(c->stream));
  return CMGPU_OK;
}

// the device code of this translation unit is loaded by the HIP runtime at the first launch of one of its kernels (milliseconds to tens of
// milliseconds for the larger ones): context creation launches this empty kernel so that a job's first batch does not pay for it (cm_api.hip: cm_load_device_code)
__global__ void k_touch_synth() {}
void cm_touch_synth(hipStream_t s) { hipLaunchKernelGGL(k_touch_synth, dim3(1), dim3(1), 0, s); }
