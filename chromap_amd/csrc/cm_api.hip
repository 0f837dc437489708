// cm_api.hip -- C ABI (include/chromap_amd.h): context, HBM residency of index and
// reference, per-batch orchestration of the stage kernels.  No CPU path: every entry
// point that computes needs a HIP device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/chromap_amd.h"
#include "cm_ctx.h"
#include "cm_kernels.h"
#include "cm_coop.h"
#include <chrono>
#include <mutex>
#include "cm_mapq_tables.h"

static thread_local std::string g_last_error;  // errors without a ctx (creation), per calling thread

#define HIPCHECK(ctx, call)                                                                      \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      cm_set_error(ctx, std::string(#call) + ": " + hipGetErrorString(e_));                     \
      return CMGPU_EHIP;                                                                         \
    }                                                                                            \
  } while (0)

// environment knobs (include/chromap_amd_debug.h): read once
static bool cm_debug_pool() { static const bool on = getenv("CM_DEBUG_POOL") != nullptr; return on; }
static thread_local std::string t_last_error;  // the calling thread's own last message (calls on one ctx from several threads: cmgpu_fastq_scan*)
void cm_set_error(cmgpu_ctx *ctx, const std::string &msg) {
  static std::mutex mu;  // (the scans of a batch's files may fail side by side)
  std::lock_guard<std::mutex> lk(mu);
  if (ctx) ctx->err = msg;
  g_last_error = msg;
  t_last_error = msg;
}
extern "C" const char *cmgpu_last_error_thread(void) { return t_last_error.c_str(); }

int DevBuf::ensure(size_t bytes) {
  if (bytes <= cap) return 0;
  if (!owned) return -1;
  // CM_ALLOC_TRACE=1: every growth with its size and the time hipFree / hipMalloc took, on stderr (what a job's first batch spends on memory)
  static const bool trace = getenv("CM_ALLOC_TRACE") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  const size_t had = cap;
  if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
  size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&p, want);
  if (trace)
    fprintf(stderr, "[alloc] %10.3f MB (had %10.3f) %8.3f ms\n", (double)want / 1e6, (double)had / 1e6,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  if (e != hipSuccess) {
    (void)hipGetLastError();  // the retry below decides
    want = bytes;
    e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return -1; }
  }
  cap = want;
  return 0;
}
void DevBuf::release() {
  if (p && owned) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}

// tuning knobs for measurements (defaults are what the measurements chose)
static int build_fast_table(cmgpu_ctx *c, int shift);
// the arrays indexed through m_off (merged / filtered candidates, draft mappings, alignment results): room for n_m entries
static int cm_ensure_candidate_arrays(cmgpu_ctx *c, uint64_t n_m) {
  return c->mbuf.ensure((size_t)n_m * 8 + 8) || c->mcnt.ensure((size_t)n_m + 4) || c->fbuf.ensure((size_t)n_m * 8 + 8) ||
         c->fcnt.ensure((size_t)n_m + 4) || c->dpos.ensure((size_t)n_m * 8 + 8) || c->derr.ensure((size_t)n_m * 2 + 4) ||
         c->dsplit.ensure((size_t)n_m * 4 + 4) || c->v_err.ensure((size_t)n_m * 2 + 4) || c->v_end.ensure((size_t)n_m * 2 + 4);
}
extern "C" int cmgpu_set_option(cmgpu_ctx *c, const char *name, int64_t value) {
  if (!c || !name) return CMGPU_EINVAL;
  const std::string n(name);
  if (n == "probe_lookups_per_lane") {
    if (value != 1 && value != 2 && value != 4 && value != 8) { cm_set_error(c, "probe_lookups_per_lane: 1, 2, 4 or 8"); return CMGPU_EINVAL; }
    c->opt_probe_variant = (int)value | (c->opt_probe_variant & 16);
  } else if (n == "probe_pair_prefetch") {
    c->opt_probe_variant = (c->opt_probe_variant & 15) | (value ? 16 : 0);
  } else if (n == "h2d_copy_blocks") {  // 0: hipMemcpyAsync; else blocks of the copy kernel that reads page-locked host memory
    if (value < 0 || value > 4096) { cm_set_error(c, "h2d_copy_blocks: 0..4096"); return CMGPU_EINVAL; }
    c->opt_h2d_kernel = (int)value;
  } else if (n == "d2h_copy_blocks") {  // the record download the same way (0: hipMemcpyAsync)
    if (value < 0 || value > 4096) { cm_set_error(c, "d2h_copy_blocks: 0..4096"); return CMGPU_EINVAL; }
    c->opt_d2h_kernel = (int)value;
  } else if (n == "mm_chunks") {
    if (value < 1 || value > CM_MM_CHUNKS) { cm_set_error(c, "mm_chunks: 1.." + std::to_string(CM_MM_CHUNKS)); return CMGPU_EINVAL; }
    c->opt_mm_chunks = (int)value;
  } else if (n == "prep_kernel") {
    c->opt_prep_kernel = (int)value;
  } else if (n == "prep_tile_reads") {
    if (value < 8 || value > 128) { cm_set_error(c, "prep_tile_reads: 8..128"); return CMGPU_EINVAL; }
    c->opt_prep_tile_reads = (int)value;
  } else if (n == "heavy_wave_max" || n == "heavy_block_max" || n == "heavy_big_max") {  // tests: force the size classes
    c->opt_heavy_max[n == "heavy_wave_max" ? 0 : n == "heavy_block_max" ? 1 : 2] = (int)value;
  } else if (n == "s3b_lane_cap") {  // hits a lane clusters in its own LDS slots; longer lists go to a wave / block each
    if (value < 0 || value > 64) { cm_set_error(c, "s3b_lane_cap: 0 (by read length) or 1..64"); return CMGPU_EINVAL; }
    c->opt_s3b_cap = (int)value;
  } else if (n == "exchange_overlap") {
    c->opt_exchange_overlap = value ? 1 : 0;
  } else if (n == "lanes") {
    if (value < 1 || value > 8) { cm_set_error(c, "lanes: 1..8"); return CMGPU_EINVAL; }
    c->opt_lanes = (int)value;
  } else if (n == "first_read_id") {  // read id of the resident batch's first pair (device-generated batches start at 0)
    if (value < 0 || value > 0xffffffffll) { cm_set_error(c, "first_read_id: 0..2^32-1"); return CMGPU_EINVAL; }
    c->first_read_id = (uint32_t)value;
  } else if (n == "heavy_mid_max") {  // longest hit list a group of 16 lanes takes (default 64; -1: no such class)
    c->opt_heavy_mid = (int)value;
  } else if (n == "heavy_last") {
    c->opt_heavy_last = (int)value;
  } else if (n == "probe_table_shift") {  // 0: probe the file's table; 1 / 2: a device copy with 2 / 4 times the buckets
    if (value < 0 || value > 4) { cm_set_error(c, "probe_table_shift: 0..4"); return CMGPU_EINVAL; }
    return cm_build_fast_table(c, (int)value);
  } else if (n == "long_read_fused") {  // 0: reads longer than 69 bases take the two-pass minimizer kernels (count, scan, fill)
    c->opt_long_fused = value ? 1 : 0;
  } else if (n == "verify_planes") {  // 0: k_s5b_verify aligns on the reference / read bytes (the round-2 form) instead of their bit planes
    c->opt_planes = value ? 1 : 0;  // (the planes stay where they are: contexts made by cmgpu_create_shared may hold views of them)
  } else if (n == "speculative_sizes") {  // 0: every batch waits for the total of its candidate lists before sizing their arrays
    c->opt_spec = value ? 1 : 0;
    if (value < 0) { c->cls_seen = 0; memset(c->cls_age, 0, sizeof(c->cls_age)); }  // tests: an empty speculative launch set -- the next range finds classes with items and is mapped again
  } else if (n == "debug_candidate_capacity") {  // tests: pretend the previous batch left this much room (forces the re-run path)
    // (the arrays are made to hold what is claimed: the device-side check trusts m_cap)
    if (value > 0 && cm_ensure_candidate_arrays(c, (uint64_t)value)) { cm_set_error(c, "out of device memory (candidates)"); return CMGPU_ENOMEM; }
    c->pred_m_ok = value > 0; c->m_cap = (uint64_t)value; c->pred_n = 0xffffffffu;  // (any batch size)
  } else if (n == "coop_profile") {  // measurement aid: per-phase cycle sums of k_s3b_coop (cmgpu_get_option coop_profile_0 .. _15)
    if (value) { if (c->coop_prof.ensure(64 * 8)) return CMGPU_ENOMEM; HIPCHECK(c, hipMemset(c->coop_prof.p, 0, 64 * 8)); } else c->coop_prof.release();
  } else if (n == "coop_run_table") {  // tests: a small table makes the cooperative sorters decline reads (their fallback paths)
    c->opt_coop_rb = (int)value;
  } else if (n == "coop") {  // bit mask of the stages whose long lists go to groups of lanes (cm_coop.h)
    c->opt_coop = (int)value;
  } else if (n == "item_limit") {  // forces the sub-batch path (tests): largest dense intermediate the pipeline may allocate
    c->opt_item_limit = value > 0 ? (uint64_t)value : 0xfffffff0ull;
  } else {
    cm_set_error(c, "unknown option " + n);
    return CMGPU_EINVAL;
  }
  return CMGPU_OK;
}

extern "C" int cmgpu_get_option(const cmgpu_ctx *c, const char *name, int64_t *value) {
  if (!c || !name || !value) return CMGPU_EINVAL;
  const std::string n(name);
  if (n == "probe_lookups_per_lane") *value = c->opt_probe_variant & 15;
  else if (n == "probe_pair_prefetch") *value = (c->opt_probe_variant & 16) ? 1 : 0;
  else if (n == "mm_chunks") *value = c->opt_mm_chunks;
  else if (n == "h2d_copy_blocks") *value = c->opt_h2d_kernel;
  else if (n == "d2h_copy_blocks") *value = c->opt_d2h_kernel;
  else if (n == "prep_kernel") *value = c->opt_prep_kernel;
  else if (n == "prep_tile_reads") *value = c->opt_prep_tile_reads;
  else if (n == "item_limit") *value = (int64_t)c->opt_item_limit;
  else if (n == "lanes") *value = c->opt_lanes;
  else if (n == "coop") *value = c->opt_coop;
  else if (n.rfind("coop_profile_", 0) == 0) {
    const int k = atoi(n.c_str() + 13);
    unsigned long long v = 0;
    if (k < 0 || k > 63 || !c->coop_prof.p || hipMemcpy(&v, (const unsigned long long *)c->coop_prof.p + k, 8, hipMemcpyDeviceToHost) != hipSuccess) return CMGPU_EINVAL;
    *value = (int64_t)v;
  }
  else if (n == "probe_table_buckets") *value = c->fmask ? (int64_t)c->fmask + 1 : (int64_t)c->bmask + 1;
  else return CMGPU_EINVAL;
  return CMGPU_OK;
}

extern "C" const char *cmgpu_last_error(const cmgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

// ---------------------------------------------------------------------------------------
// index re-pack: khash (flags, keys, vals) -> 16-byte {key,val} buckets at the same bucket
// index; empty -> CM_EMPTY_KEY, deleted -> CM_DELETED_KEY (khash.h:165-173)
// ---------------------------------------------------------------------------------------
__global__ void k_repack(const uint32_t *__restrict__ flags, const uint64_t *__restrict__ keys,
                         const uint64_t *__restrict__ vals, uint64_t *__restrict__ bkt, uint32_t i0, uint32_t n) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t i = i0 + j;
  const uint32_t f = (flags[i >> 4] >> ((i & 0xfU) << 1)) & 3u;
  uint64_t k = keys[j], v = vals[j];
  if (f & 2u) { k = CM_EMPTY_KEY; v = 0; }
  else if (f & 1u) { k = CM_DELETED_KEY; v = 0; }
  bkt[2 * (uint64_t)i] = k;
  bkt[2 * (uint64_t)i + 1] = v;
}

// the table re-hashed into `mask + 1` buckets (a power of two, larger): every occupied bucket of the file layout is inserted at
// the first empty bucket of khash's own probe sequence (hash & mask, then += 1, 2, 3, ...), so kh_get's walk finds the same keys
// with the same values and stops at an empty bucket for the others; only the number of buckets it visits changes
__global__ __launch_bounds__(256) void k_rehash(const uint64_t *__restrict__ src, uint64_t n_src, uint64_t *__restrict__ dst, uint32_t mask) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_src) return;
  const uint64_t key = src[2 * i], val = src[2 * i + 1];
  if (key == CM_EMPTY_KEY || key == CM_DELETED_KEY) return;
  uint32_t b = (uint32_t)(key >> 1) & mask, step = 0;
  for (;;) {
    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(dst + 2 * (uint64_t)b), (unsigned long long)CM_EMPTY_KEY, (unsigned long long)key);
    if (old == CM_EMPTY_KEY) { dst[2 * (uint64_t)b + 1] = val; return; }
    b = (b + (++step)) & mask;
  }
}
int cm_build_fast_table(cmgpu_ctx *c, int shift) {
  HIPCHECK(c, cm_enter(c));
  HIPCHECK(c, cm_stream_sync(c->stream));
  // contexts made by cmgpu_create_shared view this table (the lanes refresh their views, lane_prepare; a caller's own children do not)
  if (c->shared_children > (int)c->lanes.size()) { cm_set_error(c, "probe_table_shift: set it before cmgpu_create_shared (contexts that share this index view the table)"); return CMGPU_EINVAL; }
  c->bkt_fast.release();
  c->fmask = 0;
  if (shift <= 0) return CMGPU_OK;
  if (!c->bkt.p || !c->bkt.owned) { cm_set_error(c, "probe_table_shift: set it on the context that owns the index"); return CMGPU_EINVAL; }
  const uint64_t nb = (uint64_t)c->bmask + 1;
  uint64_t nf = nb << shift;
  if (nf > (1ull << 32)) nf = 1ull << 32;  // khash starts at the low 32 bits of the hash
  if (nf <= nb) return CMGPU_OK;
  if (c->bkt_fast.ensure((size_t)nf * 16)) { cm_set_error(c, "out of device memory (re-hashed index table)"); return CMGPU_ENOMEM; }
  HIPCHECK(c, hipMemsetAsync(c->bkt_fast.p, 0xff, (size_t)nf * 16, c->stream));
  hipLaunchKernelGGL(k_rehash, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, (const uint64_t *)c->bkt.p, nb, (uint64_t *)c->bkt_fast.p,
                     (uint32_t)(nf - 1));
  HIPCHECK(c, cm_stream_sync(c->stream));
  c->fmask = (uint32_t)(nf - 1);
  return CMGPU_OK;
}

static int build_mapq_tables(cmgpu_ctx *c) {
  // mapping_generator.h:924-925, 963-966: mapq_coef_fraction = (int)log(50) = 3
  std::vector<double> coef;
  std::vector<uint32_t> brk;
  cm_build_len_coef(coef);
  cm_build_nsec_break(brk);
  c->n_break = (int)brk.size();
  if (c->len_coef.ensure(coef.size() * 8) || c->nsec_break.ensure(brk.size() * 4)) return CMGPU_ENOMEM;
  HIPCHECK(c, hipMemcpy(c->len_coef.p, coef.data(), coef.size() * 8, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(c->nsec_break.p, brk.data(), brk.size() * 4, hipMemcpyHostToDevice));
  return CMGPU_OK;
}

static void cm_load_device_code(hipStream_t s);
int cm_ctx_init_common(cmgpu_ctx *c, const cmgpu_params *params, int kmer, int window, int device_id) {
  c->device = device_id;
  c->hp = *params;
  if (params->max_num_best_mappings < 1 || params->max_num_best_mappings > CM_MAX_BEST) {
    cm_set_error(c, "max_num_best_mappings must be in [1, " + std::to_string(CM_MAX_BEST) + "]"); return CMGPU_EINVAL;
  }
  if (params->max_num_best_mappings > 1 && params->output_format == 1) {  // one SAM slot per read on the device
    cm_set_error(c, "--SAM with more than one best mapping per read is outside this build"); return CMGPU_EINVAL;
  }
  if (kmer < 1 || kmer > 28 || window < 1 || window > CM_MAX_W_HOST) { cm_set_error(c, "unsupported k/w"); return CMGPU_EINVAL; }
  if (params->error_threshold < 1 || params->error_threshold > 15) { cm_set_error(c, "error_threshold must be 1..15"); return CMGPU_EINVAL; }
  CmParams &p = c->p;
  p.e = params->error_threshold;
  p.min_seeds = params->min_num_seeds;
  p.f0 = params->max_seed_frequency0;
  p.f1 = params->max_seed_frequency1;
  p.max_insert = params->max_insert_size;
  p.min_read_len = params->min_read_length;
  p.max_best = params->max_num_best_mappings;
  p.drop_rep = params->drop_repetitive_reads;
  p.trim = params->trim_adapters;
  p.split = params->split_alignment ? 1 : 0;
  p.bc_err = params->bc_error_threshold;
  p.bc_keep = params->output_mappings_not_in_whitelist ? 1 : 0;
  p.bc_prob = params->bc_probability_threshold;
  if (p.bc_err < 0 || p.bc_err > 2) { cm_set_error(c, "bc_error_threshold must be 0, 1 or 2"); return CMGPU_EINVAL; }
  p.k = kmer;
  p.w = window;
  p.lanes = p.e < 8 ? 8 : (p.e < 16 ? 4 : 0);  // GetNumVPULanes (mapping_parameters.h:80-88)
  if (p.split) p.lanes = 0;                    // split alignment cannot use the lane-grouped loop (draft_mapping_generator.cc:31)
  p.ref_batch = params->read_batch_size > 0 ? params->read_batch_size : 500000;
  p.grain = params->taskloop_grain_size > 0 ? params->taskloop_grain_size : 5000;
  p.sam = params->output_format == CMGPU_FORMAT_SAM ? 1 : 0;
  p.pairs_out = params->output_format == CMGPU_FORMAT_PAIRS && !params->split_alignment ? 1 : 0;  // (split alignment writes pairs records anyway)
  if (params->output_format != 0 && params->output_format != CMGPU_FORMAT_SAM && params->output_format != CMGPU_FORMAT_PAIRS) { cm_set_error(c, "unknown output_format"); return CMGPU_EINVAL; }
  HIPCHECK(c, hipStreamCreate(&c->stream));
  HIPCHECK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));  // same priority as `stream`: a probe stream of higher or lower
                                                                             // priority measured 3-5 % slower end to end
  // the FASTQ streams' HIP streams, made HERE, right behind the two mapping streams, when the process asks for it (CM_FQ_EARLY=1; chromap-amd sets
  // it): the runtime spreads streams of one priority over its hardware queues as they are made; made lazily by the first scans, read 1's and
  // read 2's streams both landed on queue 4 and their inflate kernels ran one behind the other (k_bgzf_tokens 9.2 + 9.2 ms; made here: queues 2
  // and 4, side by side, BGZF -> BED 0.18 -> 0.17 s).  Not the default for library callers: three more streams ahead of the lanes' shift THEIR queues
  { const char *e = getenv("CM_FQ_EARLY"); if (e && e[0] != '0')
    // (a priority of their own for these streams, highest or lowest: measured on a 32 M-pair job, no difference)
    for (int m = 0; m < 3; ++m) if (hipStreamCreateWithFlags(&c->fq[m].hs, hipStreamNonBlocking) != hipSuccess) { c->fq[m].hs = nullptr; (void)hipGetLastError(); }
  }
  for (hipEvent_t &e : c->chunk_ev) HIPCHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (c->stats.ensure(CM_ST_N * 8)) return CMGPU_ENOMEM;
  HIPCHECK(c, hipMemset(c->stats.p, 0, CM_ST_N * 8));
  for (int i = 0; i < CM_MAX_EVENTS; ++i) HIPCHECK(c, hipEventCreate(&c->ev[i]));
  cm_load_device_code(c->stream);
  return build_mapq_tables(c);
}

// Every translation unit's device code, loaded now (the runtime defers it to the first launch: round 6 found 13 ms of a job's first FASTQ
// scan inside the first prefix sum -- cm_kernels' code object being loaded -- and as much again in the first mapping and post-processing calls)
void cm_touch_post(hipStream_t s);
void cm_touch_ingest(hipStream_t s);
void cm_touch_exchange(hipStream_t s);
void cm_touch_synth(hipStream_t s);
__global__ void k_touch_api() {}
static void cm_load_device_code(hipStream_t s) {
  static std::once_flag once[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::call_once(once[dev], [s]() {
    hipLaunchKernelGGL(k_touch_api, dim3(1), dim3(1), 0, s);
    cm_touch_post(s); cm_touch_ingest(s); cm_touch_exchange(s); cm_touch_synth(s);
    // (cm_kernels.hip, the largest: through one of its launchers -- a prefix sum of four numbers -- the file itself stays as the
    //  committed profiles' source hash has it)
    DevBuf tiny;
    if (tiny.ensure((8 + 8 + cm_scan_tmp_words(4)) * 4) == 0) {
      (void)hipMemsetAsync(tiny.p, 0, 64, s);
      cm_scan_u32((const uint32_t *)tiny.p, (uint32_t *)tiny.p + 8, 4, (uint32_t *)tiny.p + 16, s);
    }
    (void)hipStreamSynchronize(s);
    tiny.release();
    (void)hipGetLastError();
  });
}

static int select_device(int device_id);
static int select_device(int device_id) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    cm_set_error(nullptr, "no HIP device available (this library has no CPU path)");
    return CMGPU_ENODEVICE;
  }
  if (device_id < 0 || device_id >= n) { cm_set_error(nullptr, "device_id out of range"); return CMGPU_EINVAL; }
  if (hipSetDevice(device_id) != hipSuccess) { cm_set_error(nullptr, "hipSetDevice failed"); return CMGPU_EHIP; }
  // (CM_BLOCKING_SYNC=1, before the device's first use by the process: host threads that wait for the device sleep instead of spinning --
  //  a process with a CPU quota and inflating threads of its own has better use for the processors; 0.1-0.2 ms more per wait)
  static const bool blocking = getenv("CM_BLOCKING_SYNC") != nullptr;
  if (blocking) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  (void)hipGetLastError();  // clean slate: cm_stream_sync reports this call's launches only
  return CMGPU_OK;
}

// reference -> HBM: raw bytes, 64 zero bytes after every sequence (so windows read past an end see 0)
int cm_upload_reference(cmgpu_ctx *c, const cmgpu_ref_view *ref) {
  c->n_seq = ref->n_sequences;
  c->h_ref_off.resize(c->n_seq);
  c->h_ref_len.assign(ref->lengths, ref->lengths + c->n_seq);
  c->goff_tried = false; c->goff.release();  // (the 32-bit key offsets follow the reference's lengths)
  uint64_t tot = 64;
  for (uint32_t i = 0; i < c->n_seq; ++i) {
    c->h_ref_off[i] = tot;
    tot += (uint64_t)ref->lengths[i] + 64;
    tot = (tot + 15) & ~15ull;
  }
  c->ref_bytes = tot;
  c->ref_planes.release(); c->ref_pl_words = 0;
  if (c->ref.ensure(tot) || c->ref_off.ensure((size_t)(c->n_seq ? c->n_seq : 1) * 8) || c->ref_len.ensure((size_t)(c->n_seq ? c->n_seq : 1) * 4)) {
    cm_set_error(c, "out of device memory (reference)"); return CMGPU_ENOMEM;
  }
  hipError_t e = hipMemset(c->ref.p, 0, tot);
  for (uint32_t i = 0; e == hipSuccess && i < c->n_seq; ++i)
    e = hipMemcpy((uint8_t *)c->ref.p + c->h_ref_off[i], ref->sequences[i], ref->lengths[i], hipMemcpyHostToDevice);
  if (e == hipSuccess && c->n_seq) e = hipMemcpy(c->ref_off.p, c->h_ref_off.data(), (size_t)c->n_seq * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && c->n_seq) e = hipMemcpy(c->ref_len.p, c->h_ref_len.data(), (size_t)c->n_seq * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) { cm_set_error(c, std::string("reference upload: ") + hipGetErrorString(e)); return CMGPU_EHIP; }
  return CMGPU_OK;
}

// the reference bytes -> interleaved bit-plane records (CmDev::ref_pl: 16 bytes per 32 bases, half a byte per base next to the
// bytes themselves), CM_PL_LEAD zero records in front of record 0
int cm_build_ref_planes(cmgpu_ctx *c) {
  const uint64_t words = (c->ref_bytes + 31) / 32 + 4;
  const size_t bytes = (size_t)(words + CM_PL_LEAD) * sizeof(CmPlRec);
  if (c->ref_planes.ensure(bytes)) { cm_set_error(c, "out of device memory (reference bit planes)"); return CMGPU_ENOMEM; }
  HIPCHECK(c, hipMemsetAsync(c->ref_planes.p, 0, bytes, c->stream));
  cm_launch_k_pack_ref((const uint8_t *)c->ref.p, c->ref_bytes, (CmPlRec *)c->ref_planes.p + CM_PL_LEAD, c->stream);
  HIPCHECK(c, cm_stream_sync(c->stream));  // the lanes read them from streams of their own
  c->ref_pl_words = words;
  return CMGPU_OK;
}

extern "C" int cmgpu_create(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                            int device_id, cmgpu_ctx **out) {
  if (!index || !ref || !params || !out) { cm_set_error(nullptr, "null argument"); return CMGPU_EINVAL; }
  *out = nullptr;
  int rc = select_device(device_id);
  if (rc) return rc;
  const uint32_t nb = index->n_buckets;
  if (nb == 0 || (nb & (nb - 1))) { cm_set_error(nullptr, "n_buckets must be a power of two"); return CMGPU_EINVAL; }
  cmgpu_ctx *c = new cmgpu_ctx();
  rc = cm_ctx_init_common(c, params, index->kmer_size, index->window_size, device_id);
  if (rc) { g_last_error = c->err; cmgpu_destroy(c); return rc; }
  // ---- index -> HBM, re-packed in slabs so host staging stays small
  if (c->bkt.ensure((size_t)nb * 16)) { cm_set_error(nullptr, "out of device memory (index)"); cmgpu_destroy(c); return CMGPU_ENOMEM; }
  {
    const size_t fwords = nb < 16 ? 1 : nb >> 4;
    DevBuf dflags, dk, dv;
    const uint32_t slab = 1u << 24;
    if (dflags.ensure(fwords * 4) || dk.ensure((size_t)(nb < slab ? nb : slab) * 8) || dv.ensure((size_t)(nb < slab ? nb : slab) * 8)) {
      cm_set_error(nullptr, "out of device memory (index staging)"); cmgpu_destroy(c); return CMGPU_ENOMEM;
    }
    hipError_t e = hipMemcpy(dflags.p, index->flags, fwords * 4, hipMemcpyHostToDevice);
    for (uint32_t i0 = 0; e == hipSuccess && i0 < nb; i0 += slab) {
      const uint32_t m = nb - i0 < slab ? nb - i0 : slab;
      e = hipMemcpy(dk.p, index->keys + i0, (size_t)m * 8, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = hipMemcpy(dv.p, index->vals + i0, (size_t)m * 8, hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_repack, dim3((m + 255) / 256), dim3(256), 0, 0, (const uint32_t *)dflags.p,
                           (const uint64_t *)dk.p, (const uint64_t *)dv.p, (uint64_t *)c->bkt.p, i0, m);
        e = hipDeviceSynchronize();
      }
      if (i0 + m == nb) break;
    }
    dflags.release(); dk.release(); dv.release();
    if (e != hipSuccess) { cm_set_error(nullptr, std::string("index upload: ") + hipGetErrorString(e)); cmgpu_destroy(c); return CMGPU_EHIP; }
  }
  c->bmask = nb - 1;
  c->n_occ = index->n_occurrences;
  if (c->occ.ensure((size_t)(c->n_occ ? c->n_occ : 1) * 8)) { cm_set_error(nullptr, "out of device memory (occurrences)"); cmgpu_destroy(c); return CMGPU_ENOMEM; }
  if (c->n_occ && hipMemcpy(c->occ.p, index->occurrences, (size_t)c->n_occ * 8, hipMemcpyHostToDevice) != hipSuccess) {
    cm_set_error(nullptr, "occurrence upload failed"); cmgpu_destroy(c); return CMGPU_EHIP;
  }
  rc = cm_upload_reference(c, ref);
  if (rc) { cm_set_error(nullptr, c->err); cmgpu_destroy(c); return rc; }
  // the pipeline probes a copy of the table re-hashed into twice the buckets (same keys, values, hash and probe sequence: identical
  // lookups, 1.18 instead of 1.69 buckets visited per lookup; + 16 bytes per bucket of HBM) unless memory is short; the file's table
  // stays resident for export / save.  cmgpu_set_option "probe_table_shift" 0 releases the copy.
  if (cm_build_fast_table(c, 1) != CMGPU_OK) { c->bkt_fast.release(); c->fmask = 0; c->err.clear(); }
  *out = c;
  return CMGPU_OK;
}

// A second context over the same resident index + reference (views, no copy): two host threads can
// then keep two batches in flight on one GPU -- the kernels of one batch's latency-bound stages share the
// CUs with the other's VALU-bound stages.  The parent must outlive its children.
extern "C" int cmgpu_create_shared(const cmgpu_ctx *parent, cmgpu_ctx **out) {
  if (!parent || !out) { cm_set_error(nullptr, "null argument"); return CMGPU_EINVAL; }
  *out = nullptr;
  int rc = select_device(parent->device);
  if (rc) return rc;
  cmgpu_ctx *c = new cmgpu_ctx();
  rc = cm_ctx_init_common(c, &parent->hp, parent->p.k, parent->p.w, parent->device);
  if (rc) { g_last_error = c->err; cmgpu_destroy(c); return rc; }
  auto view = [](DevBuf &dst, const DevBuf &src) { dst.p = src.p; dst.cap = src.cap; dst.owned = false; };
  view(c->bkt, parent->bkt); view(c->occ, parent->occ); view(c->ref, parent->ref);
  view(c->ref_off, parent->ref_off); view(c->ref_len, parent->ref_len);
  c->bmask = parent->bmask; c->n_occ = parent->n_occ; c->n_seq = parent->n_seq; c->ref_bytes = parent->ref_bytes;
  c->opt_planes = parent->opt_planes;
  // the reference's bit planes are built on the parent now, so that every child views them instead of building its own copy
  // (half a byte per reference base each); the re-hashed probe table likewise is the parent's
  if (!parent->ref_pl_words && parent->opt_planes && parent->ref_bytes && parent->ref_planes.owned) {
    cmgpu_ctx *pp = const_cast<cmgpu_ctx *>(parent);
    if (cm_build_ref_planes(pp) != CMGPU_OK) { pp->ref_planes.release(); pp->ref_pl_words = 0; pp->err.clear(); }
    (void)select_device(parent->device);
  }
  if (parent->ref_pl_words) { view(c->ref_planes, parent->ref_planes); c->ref_pl_words = parent->ref_pl_words; }  // (else: its own, on its first call)
  if (parent->fmask) { view(c->bkt_fast, parent->bkt_fast); c->fmask = parent->fmask; }
  c->h_ref_off = parent->h_ref_off; c->h_ref_len = parent->h_ref_len;
  c->goff_tried = false; c->goff.release();
  c->synth_n_minimizers = parent->synth_n_minimizers; c->synth_n_keys = parent->synth_n_keys;
  // --chr-order / --pairs-natural-chr-order of the parent apply to the child too (records carry ranks)
  if (parent->has_rank) {
    view(c->rid_rank, parent->rid_rank); view(c->ref_off_r, parent->ref_off_r); view(c->ref_len_r, parent->ref_len_r);
    c->has_rank = true;
    c->h_rank = parent->h_rank;
  }
  if (parent->has_pairs_rank) { view(c->pairs_rank, parent->pairs_rank); c->has_pairs_rank = true; }
  c->shared_parent = const_cast<cmgpu_ctx *>(parent);
  c->shared_parent->shared_children += 1;
  *out = c;
  return CMGPU_OK;
}

// --chr-order: Chromap::GenerateCustomRidRanks + SequenceBatch::ReorderSequences (chromap.cc:867-913,
// chromap.h:654-659).  rank[i] = place of reference sequence i in the output order.
extern "C" int cmgpu_set_chr_order(cmgpu_ctx *c, const uint32_t *rank, uint32_t n) {
  if (!c || !rank) return CMGPU_EINVAL;
  if (n != c->n_seq) { cm_set_error(c, "rank table size differs from the number of reference sequences"); return CMGPU_EINVAL; }
  std::vector<uint64_t> off(n);
  std::vector<uint32_t> len(n);
  std::vector<uint8_t> seen(n, 0);
  for (uint32_t i = 0; i < n; ++i) {
    if (rank[i] >= n || seen[rank[i]]) { cm_set_error(c, "rank table is not a permutation"); return CMGPU_EINVAL; }
    seen[rank[i]] = 1;
    off[rank[i]] = c->h_ref_off[i];
    len[rank[i]] = c->h_ref_len[i];
  }
  HIPCHECK(c, cm_enter(c));
  if (c->rid_rank.ensure((size_t)n * 4) || c->ref_off_r.ensure((size_t)n * 8) || c->ref_len_r.ensure((size_t)n * 4)) {
    cm_set_error(c, "out of device memory (chromosome order)"); return CMGPU_ENOMEM;
  }
  HIPCHECK(c, hipMemcpy(c->rid_rank.p, rank, (size_t)n * 4, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(c->ref_off_r.p, off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(c->ref_len_r.p, len.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  c->has_rank = true;
  c->h_rank.assign(rank, rank + n);
  return CMGPU_OK;
}

extern "C" int cmgpu_set_pairs_chr_order(cmgpu_ctx *c, const uint32_t *rank, uint32_t n) {
  if (!c || !rank) return CMGPU_EINVAL;
  if (n != c->n_seq) { cm_set_error(c, "rank table size differs from the number of reference sequences"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  if (c->pairs_rank.ensure((size_t)n * 4)) { cm_set_error(c, "out of device memory (pairs order)"); return CMGPU_ENOMEM; }
  HIPCHECK(c, hipMemcpy(c->pairs_rank.p, rank, (size_t)n * 4, hipMemcpyHostToDevice));
  c->has_pairs_rank = true;
  return CMGPU_OK;
}

extern "C" int cmgpu_destroy(cmgpu_ctx *c) {
  if (!c) return CMGPU_OK;
  if (c->in_flight) { c->worker.join(); c->in_flight = false; }
  for (cmgpu_ctx *l : c->lanes) cmgpu_destroy(l);
  c->lanes.clear();
  if (c->shared_parent) c->shared_parent->shared_children -= 1;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  cm_exchange_release(c);
  for (DevBuf *b : c->all_bufs()) b->release();
  for (CmFqStream &f : c->fq) if (f.hs) (void)hipStreamDestroy(f.hs);
  for (int i = 0; i < CM_MAX_EVENTS; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  for (hipEvent_t e : c->chunk_ev) if (e) (void)hipEventDestroy(e);
  if (c->h_maxlen) (void)hipHostFree(c->h_maxlen);
  for (hipEvent_t e : c->ev_h2d) if (e) (void)hipEventDestroy(e);
  if (c->stream_h2d) (void)hipStreamDestroy(c->stream_h2d);
  for (hipEvent_t e : c->ev_d2h) if (e) (void)hipEventDestroy(e);
  if (c->ev_comp) (void)hipEventDestroy(c->ev_comp);
  if (c->stream_d2h) (void)hipStreamDestroy(c->stream_d2h);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream_pack) (void)hipStreamDestroy(c->stream_pack);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// batch upload / residency
// ---------------------------------------------------------------------------------------
int cm_ensure_slot_scratch(cmgpu_ctx *c, uint64_t slots) {
  if (slots + 1 > 0xffffffffull) { cm_set_error(c, "too many record slots in one batch"); return CMGPU_EINVAL; }
  if (c->scratch_a.ensure((slots + 1) * 4) || c->scratch_b.ensure((slots + 1) * 4) ||
      c->scan_tmp.ensure(cm_scan_tmp_words((uint32_t)slots + 1) * 4)) {
    cm_set_error(c, "out of device memory (record compaction)"); return CMGPU_ENOMEM;
  }
  return CMGPU_OK;
}

static int ensure_pair_arrays(cmgpu_ctx *c, uint32_t n) {
  const size_t n2 = 2 * (size_t)n;
#define ENS(buf, bytes) if (c->buf.ensure(bytes)) { cm_set_error(c, "out of device memory (" #buf ")"); return CMGPU_ENOMEM; }
  ENS(rlen, n2 * 4) ENS(scratch_a, (n2 + 1) * 4) ENS(scratch_b, (n2 + 1) * 4) ENS(mm_cnt, (n2 + 1) * 4) ENS(mm_off, (n2 + 1) * 4)
  ENS(hit_tot, (n2 + 1) * 4) ENS(hit_off, (n2 + 1) * 4) ENS(round2, n2) ENS(rep_cnt, n2 * 4) ENS(rep_len, n2 * 4)
  ENS(n_pos_hit, n2 * 4) ENS(ncp, n2 * 4) ENS(ncn, n2 * 4) ENS(aug, n2) ENS(res_neg, n2 * 4) ENS(res_pos, n2 * 4)
  ENS(resc_n, n2 * 4) ENS(resc_p, n2 * 4) ENS(m_tot, (n2 + 1) * 4) ENS(m_off, (n2 + 1) * 4) ENS(mcp, n2 * 4) ENS(mcn, n2 * 4)
  ENS(nv, (n2 + 1) * 4) ENS(v_off, (n2 + 1) * 4)
  ENS(force0, n) ENS(fcp, n2 * 4) ENS(fcn, n2 * 4) ENS(alive, n) ENS(ndp, n2 * 4) ENS(ndn, n2 * 4)
  ENS(min_err, n2 * 4) ENS(second_err, n2 * 4) ENS(n_best, n2 * 4) ENS(n_second, n2 * 4)
  ENS(pe_min, (size_t)n * 4) ENS(pe_second, (size_t)n * 4) ENS(pe_nbest, (size_t)n * 4) ENS(pe_nsecond, (size_t)n * 4)
  ENS(pe_first, (size_t)n * 4) ENS(pe_i1, (size_t)n * 4) ENS(pe_i2, (size_t)n * 4) ENS(pe_choice, (size_t)n * 4 * cm_rec_per_pair(c))
  ENS(scan_tmp, cm_scan_tmp_words((uint32_t)n2 + 1) * 4)
  ENS(srt_cnt, 64) ENS(srt_list, (2 * n2 + 2) * 4) ENS(coop_slab, (size_t)CM_SLAB_BLOCKS * cm_coop_slab_bytes(CM_SLAB_CAP)) ENS(hv_cnt, 256) ENS(hv_list, CM_HV_LISTS * (n2 + 1) * 4) ENS(perm_reads, (n2 + 1) * 4) ENS(perm_pairs, ((size_t)n + 1) * 4) ENS(hv_tmp, (n2 + 1 + n + 1) * 4) ENS(rs_list, ((size_t)cm_rescue_seg_cap((uint32_t)n2) * CM_RS_SEGS + 1) * 4) ENS(rs_cnt, CM_RS_SEGS * 64)
#undef ENS
  return CMGPU_OK;
}

// longest read of a batch from its offset arrays (n + 1 entries each; o2 may be null), on the device: a host loop over
// 4 M offsets costs milliseconds of the upload path
__global__ __launch_bounds__(256) void k_max_len(const uint32_t *__restrict__ o1, const uint32_t *__restrict__ o2, uint32_t n, uint32_t *__restrict__ out) {
  uint32_t m = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t a = o1[i + 1] - o1[i], b = o2 ? o2[i + 1] - o2[i] : 0u;
    m = a > m ? a : m;
    m = b > m ? b : m;
  }
  for (int off = 32; off > 0; off >>= 1) { const uint32_t v = __shfl_down(m, off, 64); m = v > m ? v : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
static int launch_max_len(cmgpu_ctx *c, const DevBuf &o1, const DevBuf *o2, uint32_t n, hipStream_t s, int word = 0) {
  if (c->maxlen_dev.ensure(16)) { cm_set_error(c, "out of device memory"); return CMGPU_ENOMEM; }
  if (!c->h_maxlen) HIPCHECK(c, hipHostMalloc((void **)&c->h_maxlen, 16, hipHostMallocDefault));
  uint32_t *dev = (uint32_t *)c->maxlen_dev.p + word;
  HIPCHECK(c, hipMemsetAsync(dev, 0, 4, s));
  if (n) {
    uint32_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_max_len, dim3(blocks), dim3(256), 0, s, (const uint32_t *)o1.p, o2 ? (const uint32_t *)o2->p : (const uint32_t *)nullptr, n, dev);
  }
  HIPCHECK(c, hipMemcpyAsync(c->h_maxlen + word, dev, 4, hipMemcpyDeviceToHost, s));
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// page-locked host memory for the caller's batch buffers and record arrays: copies from / to it run at the link's rate
// and asynchronously (pageable memory goes through the driver's staging buffer at a fraction of it)
// ---------------------------------------------------------------------------------------
extern "C" void *cmgpu_host_alloc(uint64_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
extern "C" void cmgpu_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int cmgpu_host_register(void *p, uint64_t bytes) {
  if (!p || !bytes) return CMGPU_EINVAL;
  if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); cm_set_error(nullptr, "hipHostRegister failed"); return CMGPU_EHIP; }
  return CMGPU_OK;
}
extern "C" int cmgpu_host_unregister(void *p) {
  if (!p) return CMGPU_EINVAL;
  if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return CMGPU_EHIP; }
  return CMGPU_OK;
}

// Pipelined host-buffer entry -- the load-next-batch task beside the mapping taskloop (chromap.h:871-877): the upload of
// batch c+1 runs on a copy stream into the last parking slot while batch c is mapped.
//   cmgpu_submit_pairs(b0); for (c = 0; ...) { cmgpu_submit_pairs(b[c+1]); cmgpu_map_submitted(out[c], ...); }
// Up to two batches may be submitted and not yet mapped (parking slots 6 and 7 take turns).
//
// Upload of one buffer: page-locked memory the device can address (cmgpu_host_alloc / cmgpu_host_register) is read by a copy
// KERNEL straight over the link -- one SDMA engine moved 22-32 GB/s here (less while the mapping kernels load HBM), a few
// waves with 16-byte loads in flight fill the link; anything else goes through hipMemcpyAsync.
typedef unsigned int cm_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_host_copy(const cm_u32x4 *__restrict__ src, cm_u32x4 *__restrict__ dst, uint64_t n16,
                                                  const uint8_t *__restrict__ src_tail, uint8_t *__restrict__ dst_tail, uint32_t tail) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {  // four independent 16-byte loads per lane in flight
    const cm_u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
    const cm_u32x4 e = __builtin_nontemporal_load(src + i + 2 * stride), f = __builtin_nontemporal_load(src + i + 3 * stride);
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = e; dst[i + 3 * stride] = f;
  }
  for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
  if (blockIdx.x == 0 && threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
static int host_to_device(cmgpu_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return CMGPU_OK;
  void *dev_view = nullptr;
  if (c->opt_h2d_kernel && bytes >= 65536 && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0 &&
      hipHostGetDevicePointer(&dev_view, const_cast<void *>(src), 0) == hipSuccess && dev_view) {
    const uint64_t n16 = bytes >> 4;
    const uint32_t tail = (uint32_t)(bytes & 15u);
    hipLaunchKernelGGL(k_host_copy, dim3((unsigned)c->opt_h2d_kernel), dim3(256), 0, s, (const cm_u32x4 *)dev_view, (cm_u32x4 *)dst, n16,
                       (const uint8_t *)dev_view + (n16 << 4), (uint8_t *)dst + (n16 << 4), tail);
    return CMGPU_OK;
  }
  (void)hipGetLastError();  // hipHostGetDevicePointer on pageable memory
  HIPCHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
  return CMGPU_OK;
}
// the other direction (records): the same kernel writing to page-locked host memory when asked to (d2h_copy_blocks)
static int device_to_host(cmgpu_ctx *c, void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return CMGPU_OK;
  void *dev_view = nullptr;
  if (c->opt_d2h_kernel && bytes >= 65536 && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0 &&
      hipHostGetDevicePointer(&dev_view, dst, 0) == hipSuccess && dev_view) {
    const uint64_t n16 = bytes >> 4;
    const uint32_t tail = (uint32_t)(bytes & 15u);
    hipLaunchKernelGGL(k_host_copy, dim3((unsigned)c->opt_d2h_kernel), dim3(256), 0, s, (const cm_u32x4 *)src, (cm_u32x4 *)dev_view, n16,
                       (const uint8_t *)src + (n16 << 4), (uint8_t *)dev_view + (n16 << 4), tail);
    return CMGPU_OK;
  }
  (void)hipGetLastError();
  HIPCHECK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
  return CMGPU_OK;
}

extern "C" int cmgpu_submit_pairs(cmgpu_ctx *c, const cmgpu_batch *in) {
  if (!c || !in) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  if (c->sub_count >= 2) { cm_set_error(c, "two submitted batches are waiting (cmgpu_map_submitted)"); return CMGPU_EINVAL; }
  const uint32_t n = in->n_pairs;
  if (n > 0x3fffffffu) { cm_set_error(c, "batch too large"); return CMGPU_EINVAL; }
  if (!c->stream_h2d) {
    // a priority of its own: the runtime spreads streams of one priority over a few hardware queues (4 by default), and
    // with the lanes' streams alive the upload stream landed behind the mapping kernels' queue -- 11.5 -> 19.1 ms per
    // 4 M-pair batch; streams of another priority get queues of their own
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    HIPCHECK(c, hipStreamCreateWithPriority(&c->stream_h2d, hipStreamNonBlocking, prio_greatest));
    for (hipEvent_t &e : c->ev_h2d) HIPCHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int which = (int)(c->sub_total & 1u);
  CmBatchSlot &sl = c->slots[CM_BATCH_SLOTS - 1 - which];
  sl.n_pairs = n;
  sl.first_read_id = in->first_read_id;
  sl.bases0 = n ? in->read1_offsets[n] : 0;
  sl.bases1 = n ? in->read2_offsets[n] : 0;
  if (sl.rb0.ensure(sl.bases0 + 16) || sl.rb1.ensure(sl.bases1 + 16) || sl.ro0.ensure(((size_t)n + 1) * 4) || sl.ro1.ensure(((size_t)n + 1) * 4)) {
    cm_set_error(c, "out of device memory (reads)"); return CMGPU_ENOMEM;
  }
  hipStream_t s = c->stream_h2d;
  int rc0 = host_to_device(c, sl.ro0.p, in->read1_offsets, ((size_t)n + 1) * 4, s);
  if (!rc0) rc0 = host_to_device(c, sl.ro1.p, in->read2_offsets, ((size_t)n + 1) * 4, s);
  if (!rc0) rc0 = host_to_device(c, sl.rb0.p, in->read1_bases, sl.bases0, s);
  if (!rc0) rc0 = host_to_device(c, sl.rb1.p, in->read2_bases, sl.bases1, s);
  if (rc0) return rc0;
  int rc = launch_max_len(c, sl.ro0, &sl.ro1, n, s, 1 + which);
  if (rc) return rc;
  HIPCHECK(c, hipEventRecord(c->ev_h2d[which], s));
  ++c->sub_total;
  ++c->sub_count;
  return CMGPU_OK;
}

static int download_dense(cmgpu_ctx *c, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out);
extern "C" int cmgpu_map_submitted(cmgpu_ctx *c, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats) {
  if (!c) return CMGPU_EINVAL;
  if (c->sub_count == 0) { cm_set_error(c, "no submitted batch (cmgpu_submit_pairs)"); return CMGPU_EINVAL; }
  if (c->has_barcodes || c->single) { c->has_barcodes = false; c->single = false; }
  HIPCHECK(c, cm_enter(c));
  const int which = (int)((c->sub_total - c->sub_count) & 1u);  // the oldest submitted batch
  HIPCHECK(c, hipEventSynchronize(c->ev_h2d[which]));
  --c->sub_count;
  int rc = cmgpu_swap_resident_batch(c, CM_BATCH_SLOTS - 1 - which);
  if (rc) return rc;
  c->max_read_len = c->h_maxlen[1 + which] ? c->h_maxlen[1 + which] : 1;
  uint64_t k = 0;
  rc = cmgpu_map_resident(c, &k, stats);
  if (rc) return rc;
  if (n_out) *n_out = k;
  if (!out) return CMGPU_OK;
  return download_dense(c, out, out_capacity, n_out);
}

extern "C" int cmgpu_upload_batch(cmgpu_ctx *c, const cmgpu_batch *in) {
  if (!c || !in) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  const uint32_t n = in->n_pairs;
  if (n > 0x3fffffffu) { cm_set_error(c, "batch too large"); return CMGPU_EINVAL; }
  c->n_pairs = n;
  c->has_barcodes = false;
  c->single = false;
  c->first_read_id = in->first_read_id;
  c->bases0 = n ? in->read1_offsets[n] : 0;
  c->bases1 = n ? in->read2_offsets[n] : 0;
  if (c->rb0.ensure(c->bases0 + 16) || c->rb1.ensure(c->bases1 + 16) || c->ro0.ensure(((size_t)n + 1) * 4) || c->ro1.ensure(((size_t)n + 1) * 4)) {
    cm_set_error(c, "out of device memory (reads)");
    return CMGPU_ENOMEM;
  }
  HIPCHECK(c, hipMemcpyAsync(c->rb0.p, in->read1_bases, c->bases0, hipMemcpyHostToDevice, c->stream));
  HIPCHECK(c, hipMemcpyAsync(c->rb1.p, in->read2_bases, c->bases1, hipMemcpyHostToDevice, c->stream));
  HIPCHECK(c, hipMemcpyAsync(c->ro0.p, in->read1_offsets, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHECK(c, hipMemcpyAsync(c->ro1.p, in->read2_offsets, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  const int rc = launch_max_len(c, c->ro0, &c->ro1, n, c->stream);
  if (rc) return rc;
  HIPCHECK(c, cm_stream_sync(c->stream));
  c->max_read_len = *c->h_maxlen ? *c->h_maxlen : 1;
  return CMGPU_OK;
}

// The resident batch (bulk paired-end reads) changes places with the batch parked in `slot` (either may be empty).
extern "C" int cmgpu_swap_resident_batch(cmgpu_ctx *c, int slot) {
  if (!c || slot < 0 || slot >= CM_BATCH_SLOTS) return CMGPU_EINVAL;
  if (c->in_flight) { cm_set_error(c, "a batch is in flight"); return CMGPU_EINVAL; }
  if (c->has_barcodes || c->single) { cm_set_error(c, "only bulk paired-end batches can be parked"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  HIPCHECK(c, cm_stream_sync(c->stream));
  CmBatchSlot &sl = c->slots[slot];
  std::swap(c->rb0, sl.rb0); std::swap(c->rb1, sl.rb1); std::swap(c->ro0, sl.ro0); std::swap(c->ro1, sl.ro1);
  std::swap(c->n_pairs, sl.n_pairs); std::swap(c->first_read_id, sl.first_read_id); std::swap(c->max_read_len, sl.max_read_len);
  std::swap(c->bases0, sl.bases0); std::swap(c->bases1, sl.bases1);
  c->n_records = 0;
  return CMGPU_OK;
}

void cm_fill_dev_range(cmgpu_ctx *c, CmDev &d, uint32_t lo, uint32_t hi);
void cm_fill_dev(cmgpu_ctx *c, CmDev &d) { cm_fill_dev_range(c, d, 0, c->n_pairs); }

// the stage kernels' view of pairs [lo, hi) of the resident batch: per-pair inputs and outputs are offset views,
// the intermediates (indexed from 0) are reused by every sub-batch
void cm_fill_dev_range(cmgpu_ctx *c, CmDev &d, uint32_t lo, uint32_t hi) {
  memset(&d, 0, sizeof(d));
  d.bkt = (const uint64_t *)(c->fmask ? c->bkt_fast.p : c->bkt.p); d.bmask = c->fmask ? c->fmask : c->bmask; d.occ = (const uint64_t *)c->occ.p; d.n_occ = c->n_occ;
  d.ref = (const uint8_t *)c->ref.p; d.ref_off = (const uint64_t *)c->ref_off.p; d.ref_len = (const uint32_t *)c->ref_len.p;
  if (!c->goff_tried) {  // the sequences end to end in index order (candidate rids are re-ranked only behind the pair filter), gaps between
    c->goff_tried = true;
    std::vector<uint32_t> go(c->n_seq + 1);
    uint64_t acc = 0;
    const bool lens_ok = c->n_seq && c->h_ref_len.size() == c->n_seq;  // (checked BEFORE the loop reads h_ref_len[i])
    for (uint32_t i = 0; lens_ok && i < c->n_seq; ++i) { go[i] = (uint32_t)acc; acc += (uint64_t)c->h_ref_len[i] + CM_GOFF_GAP; if (acc >= 0xffff0000ull) break; }
    if (lens_ok && acc < 0xffff0000ull && !getenv("CM_NO_KEY32")) {
      go[c->n_seq] = (uint32_t)acc;
      if (c->goff.ensure(go.size() * 4) == 0 && hipMemcpy(c->goff.p, go.data(), go.size() * 4, hipMemcpyHostToDevice) != hipSuccess) c->goff.release();
    }
  }
  d.goff = (const uint32_t *)c->goff.p;
  d.n_seq = c->n_seq;
  d.ref_pl = c->ref_pl_words && c->opt_planes ? (const CmPlRec *)c->ref_planes.p + CM_PL_LEAD : nullptr; d.ref_pl_words = c->ref_pl_words;
  d.p = c->p;
  d.p.single = c->single ? 1 : 0;
  d.mq.len_coef = (const double *)c->len_coef.p; d.mq.nsec_break = (const uint32_t *)c->nsec_break.p; d.mq.n_break = c->n_break;
  d.n_pairs = hi - lo; d.first_read_id = c->first_read_id + lo;
  d.rb0 = (const uint8_t *)c->rb0.p; d.rb1 = (const uint8_t *)c->rb1.p;
  d.ro0 = (const uint32_t *)c->ro0.p + lo; d.ro1 = (const uint32_t *)c->ro1.p + lo;
#define PTR(f, T) d.f = (T *)c->f.p;
  PTR(rlen, uint32_t) PTR(mm_cnt, uint32_t)
  PTR(mm_off, uint32_t) PTR(mm_hash, uint64_t) PTR(mm_ps, uint32_t) PTR(pr_val, uint64_t) PTR(pr_kind, uint8_t)
  PTR(hit_tot, uint32_t) PTR(hit_off, uint32_t) PTR(round2, uint8_t) PTR(rep_cnt, uint32_t) PTR(rep_len, uint32_t)
  PTR(hbuf, uint64_t) PTR(hcnt, uint8_t) PTR(n_pos_hit, uint32_t) PTR(ncp, uint32_t) PTR(ncn, uint32_t)
  PTR(aug, uint8_t) PTR(res_neg, int32_t) PTR(res_pos, int32_t) PTR(resc_n, uint32_t) PTR(resc_p, uint32_t)
  PTR(m_tot, uint32_t) PTR(m_off, uint32_t) PTR(mbuf, uint64_t) PTR(mcnt, uint8_t) PTR(mcp, uint32_t) PTR(mcn, uint32_t)
  PTR(force0, uint8_t) PTR(fbuf, uint64_t) PTR(fcnt, uint8_t) PTR(fcp, uint32_t) PTR(fcn, uint32_t) PTR(alive, uint8_t)
  PTR(dpos, uint64_t) PTR(derr, int16_t) PTR(dsplit, uint32_t) PTR(nv, uint32_t) PTR(v_off, uint32_t) PTR(v_err, int16_t) PTR(v_end, int16_t) PTR(ndp, uint32_t) PTR(ndn, uint32_t)
  PTR(min_err, int32_t) PTR(second_err, int32_t) PTR(n_best, int32_t) PTR(n_second, int32_t)
  PTR(pe_min, int32_t) PTR(pe_second, int32_t) PTR(pe_nbest, int32_t) PTR(pe_nsecond, int32_t)
  PTR(pe_first, uint32_t) PTR(pe_i1, uint32_t) PTR(pe_i2, uint32_t) PTR(pe_choice, uint32_t)
#undef PTR
  d.rec = (uint8_t *)c->rec.p + (size_t)lo * 24 * cm_rec_per_pair(c);
  d.rec_ok = (uint8_t *)c->rec_ok.p + (size_t)lo * cm_rec_per_pair(c);
  d.stats = (unsigned long long *)c->stats.p;
  d.srt_cnt = (uint32_t *)c->srt_cnt.p;
  d.srt_list = (uint32_t *)c->srt_list.p;
  d.hv_cnt = (uint32_t *)c->hv_cnt.p;
  d.hv_list = (uint32_t *)c->hv_list.p;
  d.rs_cnt = (uint32_t *)c->rs_cnt.p;
  d.rs_list = (uint32_t *)c->rs_list.p;
  d.hv_stride = 2 * (hi - lo) + 1;
  d.perm_reads = c->use_perm ? (const uint32_t *)c->perm_reads.p : nullptr;
  d.perm_pairs = c->use_perm ? (const uint32_t *)c->perm_pairs.p : nullptr;
  d.coop_slab = (uint8_t *)c->coop_slab.p; d.coop_slab_cap = CM_SLAB_CAP; d.coop_slab_blocks = CM_SLAB_BLOCKS;
  d.cls_mask = c->cls_all || !c->opt_spec ? ~0ull : c->cls_seen;
  d.prof = (unsigned long long *)c->coop_prof.p;
  d.rs_pool = (uint64_t *)c->rs_pool.p; d.rs_pool_cap = c->rs_pool.p ? c->rs_pool_cap : 0u; d.rs_pool_off = (uint32_t *)c->rs_pool_off.p;
  d.abort = (const unsigned long long *)c->stats.p + CM_ST_ABORT;
  d.coop_rb = c->opt_coop_rb > 0 ? (uint32_t)c->opt_coop_rb : 0u;
  d.wq_dynamic = ((uint32_t)c->opt_coop >> 16) & 1u;
  d.s3b_cap = c->opt_s3b_cap > 0 ? (uint32_t)c->opt_s3b_cap : cm_s3b_lane_cap(c->max_read_len);
  if (c->n_seq < 0x80000000u) {  // the cooperative kernel keeps the strand in bit 31 of the sequence id
    cm_s3b_heavy_classes(d.hv_max, &d.hv_big);
    // tests force the classes: heavy_wave_max caps the wave class, heavy_block_max the two middle classes, heavy_big_max the largest
    if (c->opt_heavy_max[0] > 0 && (uint32_t)c->opt_heavy_max[0] < d.hv_max[0]) d.hv_max[0] = (uint32_t)c->opt_heavy_max[0];
    for (int q = 1; q <= 2; ++q) if (c->opt_heavy_max[1] > 0 && (uint32_t)c->opt_heavy_max[1] < d.hv_max[q]) d.hv_max[q] = (uint32_t)c->opt_heavy_max[1];
    if (c->opt_heavy_max[2] > 0 && (uint32_t)c->opt_heavy_max[2] < d.hv_max[3]) d.hv_max[3] = (uint32_t)c->opt_heavy_max[2];
    for (int q = 1; q < 4; ++q) if (d.hv_max[q] < d.hv_max[q - 1]) d.hv_max[q] = d.hv_max[q - 1];
    if (c->opt_heavy_max[2] > 0 && (uint32_t)c->opt_heavy_max[2] < d.hv_big) d.hv_big = (uint32_t)c->opt_heavy_max[2];
    const uint32_t hv_big_area = d.hv_big;  // (what a CU's shared memory holds once: the rescue lists' largest class is sized from it)
    // 32-bit hit keys: 11 bytes per hit -- 7 040 hits leave room for TWO blocks of 1 024 lanes per CU (k_s3b_coop<1024, true> on profile 2:
    // 53.6 -> 43.0 ms of S3b per step against one block with 8 192; the few longer lists go to the slab launch)
    if (c->goff.p && d.hv_big > 7040u) d.hv_big = 7040u;
    if (d.hv_big <= d.hv_max[3]) d.hv_big = 0;
    if (c->opt_heavy_max[0] < 0) { d.hv_max[0] = d.hv_max[1] = d.hv_max[2] = d.hv_max[3] = 0; d.hv_big = 0; }  // everything long goes to the one-lane path
    // the rescue lists' classes: the hit lists' up to 2048, then as many 20-byte entries as fit a CU's shared memory twice / once
    d.rs_max3 = d.hv_max[3] < 3968u ? d.hv_max[3] : 3968u;
    // (round 6, measured and NOT kept: 32-bit keys in k_s4b_coop as in k_s3b_coop -- 12 instead of 20 bytes per entry, classes of 6 400 / 3 968
    //  entries at two / three blocks per CU.  The stage got SLOWER, profile 2 46.5 -> 49.7 ms and profile 1 9.7 -> 18.3 ms per step: unlike the
    //  hit lists, the rescue lists are merged with the read's own candidates and written back as sequence << 32 | position, so every hit
    //  pays a lookup of its sequence's offset on the way in and every candidate a search of the offset table on the way out, and the
    //  groups' time is not the number of reads in flight here)
    d.rs_big = d.hv_big ? (hv_big_area < 7680u ? hv_big_area : 7680u) : 0u;
    if (d.rs_big <= d.rs_max3) d.rs_big = 0;
    // the longest list a 16-lane group takes: 64 hits; 96 for reads of 100 bases and more (round 6, `tools/gpu_mid_knob.sh`: 2 x 150 hic reads
    // carry ~37 hits each with a tail to ~100 -- 64 / 80 / 96 / 112: 102.6 / 101.8 / 110.3 / 101.3 M pairs/s, twice; at 50 bases 96 is neutral on the
    // headline and the mammalian-like genomes and loses 4 % on the planted repeats).  The classes decide who works on a list, never the result
    d.hv_mid = c->opt_heavy_mid < 0 ? 0u : (c->opt_heavy_mid > 0 ? (uint32_t)c->opt_heavy_mid : (c->max_read_len >= 100 ? 96u : 64u));
    if (d.hv_mid > 256) d.hv_mid = 256;
    if (d.hv_max[0] == 0) d.hv_mid = 0;
    d.hv_sub = d.hv_max[0] >= 512 ? 256u : 0u;  // (the tests' small size classes: no sub-class)
    // with the 16-lane groups taking the lists up to hv_mid, a lane keeps the short ones only (16 hits: 256-thread blocks)
    if (d.hv_mid && c->opt_s3b_cap <= 0 && d.s3b_cap > 16) d.s3b_cap = 16;
  } else { d.hv_max[0] = d.hv_max[1] = d.hv_max[2] = d.hv_max[3] = 0; d.hv_mid = 0; d.hv_big = 0; d.rs_max3 = 0; d.rs_big = 0; }
  if (c->has_rank) {  // stages from verification on address the reference by rank
    d.rid_rank = (const uint32_t *)c->rid_rank.p;
    d.ref_off = (const uint64_t *)c->ref_off_r.p;
    d.ref_len = (const uint32_t *)c->ref_len_r.p;
  }
  if (c->has_pairs_rank) d.pairs_rank = (const uint32_t *)c->pairs_rank.p;
  if (c->has_barcodes) {
    d.bcb = (const uint8_t *)c->bcb.p; d.bcq = (const uint8_t *)c->bcq.p; d.bco = (const uint32_t *)c->bco.p + lo;
    d.wl = (const uint64_t *)c->wl.p; d.wl_mask = c->wl_mask; d.wl_num_sample = (double)c->wl_num_sample;
    d.pow10_tab = (const double *)c->pow10_tab.p;
    d.bc_key = c->bc_key.p ? (uint64_t *)c->bc_key.p + lo : nullptr; d.bc_ok = c->bc_ok.p ? (uint8_t *)c->bc_ok.p + lo : nullptr;
  }
}

static inline void mark(cmgpu_ctx *c, const char *name) {
  if (c->n_ev < CM_MAX_EVENTS) {
    (void)hipEventRecord(c->ev[c->n_ev], c->stream);
    c->ev_name[c->n_ev] = name;
    ++c->n_ev;
  }
}

// exclusive scan of per-read counts (u32 offsets) together with their exact 64-bit total: the offsets are only
// meaningful when the total fits the dense arrays' 32-bit indexing, which the caller checks on *total
static int scan_with_total(cmgpu_ctx *c, const uint32_t *in, uint32_t *out, uint32_t n, unsigned long long *total) {
  hipStream_t s = c->stream;
  unsigned long long *acc = (unsigned long long *)c->stats.p + CM_ST_TOTAL;  // spare counter slot
  HIPCHECK(c, hipMemsetAsync(acc, 0, 8, s));
  cm_launch_k_sum_u32(in, n, acc, s);
  cm_scan_u32(in, out, n, (uint32_t *)c->scan_tmp.p, s);
  HIPCHECK(c, hipMemcpyAsync(total, acc, 8, hipMemcpyDeviceToHost, s));
  HIPCHECK(c, cm_stream_sync(s));
  return CMGPU_OK;
}

#define CM_RC_SPLIT (-100)  // internal: a dense intermediate of this pair range exceeds the 32-bit item limit

// The pipeline on pairs [rlo, rhi) of the resident batch.  Four small device->host reads size the
// variable-length intermediates (minimizers, hits, candidate capacity, verification items).
static int map_range(cmgpu_ctx *c, uint32_t rlo, uint32_t rhi, uint64_t *k_out, cmgpu_stats *stats, bool allow_spec = true) {
  const uint32_t n = rhi - rlo, n2 = 2 * n;
  const uint64_t limit = c->opt_item_limit;
  c->n_ev = 0;
  c->use_perm = false;
  *k_out = 0;
  int rc = ensure_pair_arrays(c, n);
  if (rc) return rc;
  hipStream_t s = c->stream;
  CmDev d;
  HIPCHECK(c, hipMemsetAsync(c->stats.p, 0, CM_ST_N * 8, s));
  cm_fill_dev_range(c, d, rlo, rhi);
  mark(c, "begin");
  if (c->has_barcodes) {  // K6: barcode correction (chromap.h:896-909)
    cm_launch_k_s0b_barcode(d, n, s);
    mark(c, "s0b_barcode");
  }
  uint32_t n_mm = 0;
  bool s3a_done = false;
  uint32_t n_heavy[CM_HV_LISTS] = {};  // lists 0..2, 10 by size, 3: one lane each, 4: the short lists of the 16-lane groups
  unsigned long long hits_total = 0;
  const bool flat = c->opt_prep_kernel == 1 && cm_prep_flat_supported(d, c->max_read_len, (uint32_t)c->opt_prep_tile_reads);
  if (flat || (cm_prep_mm_supported(d, c->max_read_len) && (c->max_read_len <= 69 || c->opt_long_fused))) {
    // S0 + S1 fused: one pass of the minimizer state machine, block-level reservation of the dense arrays
    uint64_t cap = (uint64_t)n2 * (c->max_read_len / 4 + 3);
    const uint64_t bound = (uint64_t)c->bases0 + c->bases1 + 1;  // one emission per k-mer position at most
    if (cap > bound) cap = bound;
    // The pairs go through in chunks: chunk c's minimizers (K0 + K1, VALU-bound) are followed on a second
    // stream by their index probe (K2, latency-bound gather), which runs next to chunk c+1's minimizer
    // pass.  A chunk's minimizers are the cursor range its launch covered (copied to mm_marks on the device).
    const uint32_t ppb = flat ? 128u : cm_prep_mm_pairs_per_block(d, c->max_read_len);
    const uint32_t n_chunks = n >= (1u << 20) ? (uint32_t)c->opt_mm_chunks : (n >= (1u << 17) ? 2 : 1);
    const uint64_t per_read_bound = c->max_read_len > (uint32_t)c->p.k ? c->max_read_len - (uint32_t)c->p.k + 1 : 1;
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (cap > limit) return CM_RC_SPLIT;
      // per chunk: probe grid = what the chunk can emit at most, but never more than the arrays hold
      uint32_t lo[CM_MM_CHUNKS + 1];
      uint64_t max_entries[CM_MM_CHUNKS];
      uint32_t part_off[CM_MM_CHUNKS + 1];
      lo[0] = 0; part_off[0] = 0;
      for (uint32_t ch = 0; ch < n_chunks; ++ch) {
        uint64_t hi = (uint64_t)n * (ch + 1) / n_chunks;
        hi = ch + 1 == n_chunks ? n : hi / ppb * ppb;
        lo[ch + 1] = (uint32_t)hi;
        uint64_t me = 2ull * (lo[ch + 1] - lo[ch]) * (attempt == 0 ? (uint64_t)(c->max_read_len / 4 + 3) : per_read_bound);
        if (me > cap) me = cap;
        max_entries[ch] = me;
        part_off[ch + 1] = part_off[ch] + cm_probe_range_blocks(me, c->opt_probe_variant);
      }
      uint32_t chunk_pairs = 0;
      for (uint32_t ch = 0; ch < n_chunks; ++ch) chunk_pairs = lo[ch + 1] - lo[ch] > chunk_pairs ? lo[ch + 1] - lo[ch] : chunk_pairs;
      if (c->mm_hash.ensure((size_t)cap * 8 + 8) || c->mm_ps.ensure((size_t)cap * 4 + 4) || c->pr_val.ensure((size_t)cap * 8 + 8) ||
          (!flat && c->mm_stage.ensure(cm_prep_mm_stage_bytes(d, c->max_read_len, chunk_pairs) + 8)) ||
          c->pr_kind.ensure((size_t)cap + 4) || c->mm_cursor.ensure(8) || c->mm_marks.ensure((CM_MM_CHUNKS + 1) * 8) ||
          c->partials.ensure(((size_t)part_off[n_chunks] + 1) * 8 + cm_stats_partial_words(n) * 8)) {
        cm_set_error(c, "out of device memory (minimizers)"); return CMGPU_ENOMEM;
      }
      cm_fill_dev_range(c, d, rlo, rhi);
      HIPCHECK(c, hipMemsetAsync(c->mm_cursor.p, 0, 8, s));
      HIPCHECK(c, hipMemsetAsync(c->mm_marks.p, 0, 8, s));
      HIPCHECK(c, hipMemsetAsync(d.stats + CM_ST_PROBE_STEPS, 0, 2 * 8, s));
      unsigned long long *marks = (unsigned long long *)c->mm_marks.p;
      for (uint32_t ch = 0; ch < n_chunks; ++ch) {
        if (flat) cm_launch_k_prep_flat(d, lo[ch], lo[ch + 1], c->max_read_len, (uint32_t)c->opt_prep_tile_reads, (uint32_t)cap, (unsigned long long *)c->mm_cursor.p, s);
        else cm_launch_k_prep_mm(d, lo[ch], lo[ch + 1], c->max_read_len, (uint32_t)cap, (unsigned long long *)c->mm_cursor.p, s, c->mm_stage.p);
        cm_launch_k_copy_u64((const unsigned long long *)c->mm_cursor.p, marks + ch + 1, s);
        HIPCHECK(c, hipEventRecord(c->chunk_ev[ch], s));
        HIPCHECK(c, hipStreamWaitEvent(c->stream2, c->chunk_ev[ch], 0));
        cm_launch_k_probe_range(d, marks + ch, max_entries[ch], (uint32_t)cap, (uint2 *)c->partials.p + part_off[ch], c->stream2, c->opt_probe_variant);
      }
      HIPCHECK(c, hipEventRecord(c->chunk_ev[CM_MM_CHUNKS], c->stream2));
      HIPCHECK(c, hipStreamWaitEvent(s, c->chunk_ev[CM_MM_CHUNKS], 0));
      cm_launch_k_probe_reduce(c->partials.p, part_off[n_chunks], d.stats + CM_ST_PROBE_STEPS, s);
      // S3a (hit counts, size classes) and the scan of the counts follow at once: their totals come back with the minimizer
      // marks in ONE read (a read's minimizer range is checked against the arrays' capacity, so an overflow leaves them idle)
      d.mm_cap = (uint32_t)cap;
      HIPCHECK(c, hipMemsetAsync(c->hv_cnt.p, 0, 256, s));
      cm_launch_k_s3a_count(d, n2, s);
      {
        unsigned long long *acc = (unsigned long long *)c->stats.p + CM_ST_TOTAL;
        HIPCHECK(c, hipMemsetAsync(acc, 0, 8, s));
        cm_launch_k_sum_u32(d.hit_tot, n2, acc, s);
        cm_scan_u32(d.hit_tot, d.hit_off, n2, (uint32_t *)c->scan_tmp.p, s);
        HIPCHECK(c, hipMemcpyAsync(&hits_total, acc, 8, hipMemcpyDeviceToHost, s));
        HIPCHECK(c, hipMemcpyAsync(n_heavy, c->hv_cnt.p, sizeof(n_heavy), hipMemcpyDeviceToHost, s));
      }
      s3a_done = true;
      // one read-back: the marks (cursor after every chunk; the last one is the total)
      unsigned long long hm[CM_MM_CHUNKS + 1];
      HIPCHECK(c, hipMemcpyAsync(hm, c->mm_marks.p, ((size_t)n_chunks + 1) * 8, hipMemcpyDeviceToHost, s));  // (not the null stream: lanes run side by side)
      HIPCHECK(c, cm_stream_sync(s));
      const unsigned long long tot = hm[n_chunks];
      bool grid_short = false;  // a chunk emitted more than its probe grid covers (cannot happen on attempt 1)
      if (attempt == 0 && tot <= cap)
        for (uint32_t ch = 0; ch < n_chunks; ++ch) grid_short = grid_short || hm[ch + 1] - hm[ch] > max_entries[ch];
      if (tot <= cap && !grid_short) { n_mm = (uint32_t)tot; break; }
      if (attempt == 1) { cm_set_error(c, "minimizer arrays overflowed twice"); return CMGPU_ECAPACITY; }
      cap = bound;  // rerun with the worst-case sizes
    }
    mark(c, "s0_s1_s2_trim_minimizers_probe");
  } else {
    // S0 + S1a: length filter, adapter trimming, minimizer counts (reads staged through LDS)
    cm_launch_k_prep_count(d, n, c->max_read_len, s);
    unsigned long long mm_total = 0;
    if ((rc = scan_with_total(c, d.mm_cnt, d.mm_off, n2, &mm_total))) return rc;
    if (mm_total > limit) return CM_RC_SPLIT;
    n_mm = (uint32_t)mm_total;
    mark(c, "s0_s1a_trim_count");
    if (c->mm_hash.ensure((size_t)n_mm * 8 + 8) || c->mm_ps.ensure((size_t)n_mm * 4 + 4) || c->pr_val.ensure((size_t)n_mm * 8 + 8) ||
        c->pr_kind.ensure((size_t)n_mm + 4)) { cm_set_error(c, "out of device memory (minimizers)"); return CMGPU_ENOMEM; }
    cm_fill_dev_range(c, d, rlo, rhi);
    // S1b + S2: the minimizers are written to their dense positions chunk by chunk, and the index probe of a chunk (second
    // stream) runs under the next chunk's fill pass -- the offsets are known, so a chunk's minimizer range is
    // [mm_off[2 lo], mm_off[2 hi]) (k_mm_marks; read back once to size the probe grids)
    const uint32_t n_chunks = n >= (1u << 20) ? (uint32_t)c->opt_mm_chunks : (n >= (1u << 17) ? 2 : 1);
    uint32_t lo[CM_MM_CHUNKS + 1];
    for (uint32_t ch = 0; ch <= n_chunks; ++ch) lo[ch] = (uint32_t)((uint64_t)n * ch / n_chunks);
    if (c->mm_marks.ensure((CM_MM_CHUNKS + 1) * 8)) { cm_set_error(c, "out of device memory (minimizers)"); return CMGPU_ENOMEM; }
    unsigned long long *marks = (unsigned long long *)c->mm_marks.p;
    unsigned long long hm[CM_MM_CHUNKS + 1];
    cm_launch_k_mm_marks(d.mm_off, lo, n_chunks + 1, marks, s);
    HIPCHECK(c, hipMemcpyAsync(hm, marks, ((size_t)n_chunks + 1) * 8, hipMemcpyDeviceToHost, s));
    HIPCHECK(c, cm_stream_sync(s));
    uint32_t part_off[CM_MM_CHUNKS + 1];
    part_off[0] = 0;
    for (uint32_t ch = 0; ch < n_chunks; ++ch) part_off[ch + 1] = part_off[ch] + cm_probe_range_blocks(hm[ch + 1] - hm[ch], c->opt_probe_variant);
    if (c->partials.ensure(((size_t)part_off[n_chunks] + 1) * 8 + cm_stats_partial_words(n) * 8)) { cm_set_error(c, "out of device memory (partials)"); return CMGPU_ENOMEM; }
    HIPCHECK(c, hipMemsetAsync(d.stats + CM_ST_PROBE_STEPS, 0, 2 * 8, s));
    for (uint32_t ch = 0; ch < n_chunks; ++ch) {
      cm_launch_k_mm_fill(d, lo[ch], lo[ch + 1], c->max_read_len, s);
      HIPCHECK(c, hipEventRecord(c->chunk_ev[ch], s));
      HIPCHECK(c, hipStreamWaitEvent(c->stream2, c->chunk_ev[ch], 0));
      cm_launch_k_probe_range(d, marks + ch, hm[ch + 1] - hm[ch], n_mm, (uint2 *)c->partials.p + part_off[ch], c->stream2, c->opt_probe_variant);
    }
    HIPCHECK(c, hipEventRecord(c->chunk_ev[CM_MM_CHUNKS], c->stream2));
    HIPCHECK(c, hipStreamWaitEvent(s, c->chunk_ev[CM_MM_CHUNKS], 0));
    cm_launch_k_probe_reduce(c->partials.p, part_off[n_chunks], d.stats + CM_ST_PROBE_STEPS, s);
    mark(c, "s1b_s2_minimizers_probe");
  }
  // the alignments of S5b run on bit planes: this range's reads, both orientations, packed on the second stream (idle from here
  // on) under S3 and S4 -- the trimmed lengths are final
  bool planes = c->ref_pl_words != 0 && c->opt_planes;  // (verify_planes 0 keeps the planes for the children but does not use them: no packed reads either)
  if (planes && c->read_planes.ensure((size_t)n2 * cm_read_pl_stride((c->max_read_len + 31) / 32) * 4 + 16)) planes = false;  // (no room: this range on bytes)
  if (planes) {
    const uint32_t pw = (c->max_read_len + 31) / 32;
    // on a stream of the highest priority: at the mapping streams' priority the kernel only got the slots three lanes' S3 / S4 kernels
    // left over and S5 waited for it (2 x 150: 4 ms of a 21 ms step once the empty class launches no longer padded S4)
    if (!c->stream_pack) {
      int prio_least = 0, prio_greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
      HIPCHECK(c, hipStreamCreateWithPriority(&c->stream_pack, hipStreamNonBlocking, prio_greatest));
    }
    HIPCHECK(c, hipEventRecord(c->chunk_ev[1], s));
    HIPCHECK(c, hipStreamWaitEvent(c->stream_pack, c->chunk_ev[1], 0));
    d.read_pl = (uint32_t *)c->read_planes.p; d.read_pl_w = pw;
    cm_launch_k_pack_reads(d, n2, c->stream_pack);
    HIPCHECK(c, hipEventRecord(c->chunk_ev[0], c->stream_pack));
  }
  // S3: hit counts -> offsets -> candidates
  if (!s3a_done) {
    HIPCHECK(c, hipMemsetAsync(c->hv_cnt.p, 0, 256, s));
    cm_launch_k_s3a_count(d, n2, s);
    HIPCHECK(c, hipMemcpyAsync(n_heavy, c->hv_cnt.p, sizeof(n_heavy), hipMemcpyDeviceToHost, s));
    if ((rc = scan_with_total(c, d.hit_tot, d.hit_off, n2, &hits_total))) return rc;
  }
  if (hits_total > limit) return CM_RC_SPLIT;  // 2 x 150 reads on a repeat-rich genome: ~500 hits per read x 8 M reads wraps 2^32
  const uint32_t n_hits = (uint32_t)hits_total;
  if (c->hbuf.ensure((size_t)n_hits * 8 + 8) || c->hcnt.ensure((size_t)n_hits + 4)) { cm_set_error(c, "out of device memory (hits)"); return CMGPU_ENOMEM; }
  cm_fill_dev_range(c, d, rlo, rhi);
  mark(c, "s3a_count");
  cm_launch_k_s3b_candidates(d, n2, c->max_read_len, s);
  cm_launch_k_s3b_heavy(d, n_heavy, s, (c->opt_coop & 1) != 0, c->max_read_len);  // reads with long hit lists: a wave or a block each
  // with more than a handful of such reads the later per-read / per-pair stages take them last, in waves of their own
  // (lists of class 0 -- up to heavy_wave_max hits, a wave each here -- cost the later per-lane stages little; a uniform genome
  // still has a few thousand of them per batch, and the permutation's scans and scatters cost more than they save there)
  c->use_perm = (uint64_t)n_heavy[CM_L_HIT_B256B] + n_heavy[CM_L_HIT_SLAB] + n_heavy[CM_L_HIT_B512] + n_heavy[CM_L_HIT_B1024] > n2 / 65536 || (uint64_t)n_heavy[CM_L_HIT_WAVE] + n_heavy[CM_L_HIT_WAVE_SMALL] + n_heavy[CM_L_HIT_B256A] > n2 / 256;  // (lists 21, 0, 1: up to 1024 hits)
  if (c->opt_heavy_last) c->use_perm = c->opt_heavy_last > 0;
  if (c->use_perm) {
    uint32_t *tmp = (uint32_t *)c->hv_tmp.p;
    cm_build_heavy_last(d, n, (uint32_t *)c->scratch_a.p, tmp, (uint32_t *)c->scratch_b.p, tmp + n2 + 1, (uint32_t *)c->perm_reads.p,
                        (uint32_t *)c->perm_pairs.p, (uint32_t *)c->scan_tmp.p, s);
  }
  cm_fill_dev_range(c, d, rlo, rhi);
  mark(c, "s3b_candidates");
  // S4: mate rescue, merge, paired-end filter
  HIPCHECK(c, hipMemsetAsync(c->rs_cnt.p, 0, CM_RS_SEGS * 64, s));
  if (c->opt_coop & 2) {  // the pool of rescue hits found while counting: sized from what the previous range asked for (+ 25 %)
    // (+ a grant per wave that can take one, cm_coop_pool_take: what the waves leave unused of their last grants)
    // slack: half a grant for every wave of the rescue-wave launch (rescue_wave_blocks: n / 512 + 64, at most 8192) -- but only once the
    // waves have taken grants at all (a range whose searches never asked leaves rs_pool_want at 0: no 268 MB per lane for nothing)
    uint64_t waves = (uint64_t)n2 / 512 + 64;
    if (waves > 8192) waves = 8192;
    const uint64_t slack = c->rs_pool_want ? waves * CM_POOL_GRANT / 2 : 0;
    uint64_t want = c->rs_pool_want + c->rs_pool_want / 4 + slack;
    // (a range that ran out of pool undercounts what it would have used -- the pieces behind the refusal are only estimated -- and
    // growing the pool is a hipFree + hipMalloc of gigabytes, 0.3-0.5 s: grow once, generously)
    if (c->rs_pool_want > c->rs_pool_cap) want = 2 * c->rs_pool_want + slack;
    if (want < (1u << 22)) want = 1u << 22;  // (32 MB to begin with: a first grant per wave of a small batch; the pool then follows the demand)
    if (want > 0xfffffff0ull) want = 0xfffffff0ull;
    const auto dbg_t0 = std::chrono::steady_clock::now();
    if (c->rs_pool.ensure((size_t)want * 8) == 0 && c->rs_pool_off.ensure((size_t)n2 * 2 * 4 + 16) == 0) c->rs_pool_cap = (uint32_t)(c->rs_pool.cap / 8 > 0xfffffff0ull ? 0xfffffff0ull : c->rs_pool.cap / 8);
    else { c->rs_pool.release(); c->rs_pool_cap = 0; }  // (no pool: the fill pass searches again)
    if (cm_debug_pool()) fprintf(stderr, "pool ensure want %llu: %.3f ms\n", (unsigned long long)want, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - dbg_t0).count());
    cm_fill_dev_range(c, d, rlo, rhi);
  }
  cm_launch_k_s4a_rescue_count(d, n2, s, (c->opt_coop & 2) != 0);  // decision per read + the packed list of reads that supplement
  cm_launch_k_s4a_rescue_list(d, n2, s, (c->opt_coop & 2) != 0);   // their searches: a lane, a group of 16 lanes or a wave per read
  // The candidate arrays.  A batch of the size of the previous one reuses that batch's arrays (sized with 25 % to spare) without
  // waiting for its own total: the device compares the total with the capacity and, should it ever pass it, raises d.abort --
  // every later kernel then leaves at once and the range is mapped again with exact sizes (the `spec` test below).
  unsigned long long *acc = (unsigned long long *)c->stats.p + CM_ST_TOTAL;
  const bool spec = allow_spec && c->opt_spec && c->pred_m_ok && (c->pred_n == n || c->pred_n == 0xffffffffu) && c->m_cap > 0 && c->m_cap <= limit;
  unsigned long long m_total = 0;
  uint32_t n_m;
  if (spec) {
    HIPCHECK(c, hipMemsetAsync(acc, 0, 8, s));
    cm_launch_k_sum_u32(d.m_tot, n2, acc, s);
    cm_scan_u32(d.m_tot, d.m_off, n2, (uint32_t *)c->scan_tmp.p, s);
    cm_launch_k_check_cap(acc, c->m_cap, (unsigned long long *)c->stats.p + CM_ST_ABORT, s);
    n_m = (uint32_t)c->m_cap;
  } else {
    if ((rc = scan_with_total(c, d.m_tot, d.m_off, n2, &m_total))) return rc;
    if (m_total > limit) return CM_RC_SPLIT;
    uint64_t want = m_total + m_total / 4 + 1024;  // room for the next batch
    if (want > limit) want = limit;
    if (want < m_total) want = m_total;
    n_m = (uint32_t)want;
    if (cm_ensure_candidate_arrays(c, n_m)) {
      cm_set_error(c, "out of device memory (candidates)");
      return CMGPU_ENOMEM;
    }
    c->m_cap = n_m;
  }
  if (cm_debug_pool()) fprintf(stderr, "spec %d m_cap %llu m_total %llu\n", (int)spec, (unsigned long long)c->m_cap, (unsigned long long)m_total);
  cm_fill_dev_range(c, d, rlo, rhi);
  mark(c, "s4a_rescue_count");
  cm_launch_k_s4b_rescue_merge(d, n2, s, (c->opt_coop & 2) != 0, c->max_read_len);
  cm_launch_k_s4b_rescue_list(d, n2, s, (c->opt_coop & 2) != 0, c->max_read_len);
  mark(c, "s4b_rescue_merge");
  HIPCHECK(c, hipMemsetAsync(c->srt_cnt.p, 0, 8, s));
  cm_launch_k_s4c_reduce(d, n, s, (uint32_t)c->opt_coop & (c->p.split ? ~8u : ~0u));
  if (c->use_perm) cm_launch_k_sort_lists(d, 0, s);  // long candidate lists: a wave each, before S5a wants them in order
  mark(c, "s4c_pair_filter");
  // S5: verification -- (a) shortcut / sort + work-item counts, (b) one banded alignment per
  // candidate, (c) the sequential acceptance loop per read
  if (planes) {  // (split alignments are verified in S5a already)
    HIPCHECK(c, hipStreamWaitEvent(s, c->chunk_ev[0], 0));  // k_pack_reads (second stream) is done
    d.read_pl = (uint32_t *)c->read_planes.p; d.read_pl_w = (c->max_read_len + 31) / 32;
  }
  cm_launch_k_s5a_prepare(d, n2, s, (c->opt_coop & 8) != 0 && !c->p.split);
  cm_scan_u32(d.nv, d.v_off, n2, (uint32_t *)c->scan_tmp.p, s);  // the items' number stays on the device: never above n_m
  mark(c, "s5a_prepare");
  cm_launch_k_s5b_verify(d, n_m, n2, s);
  mark(c, "s5b_verify");
  HIPCHECK(c, hipMemsetAsync(c->srt_cnt.p, 0, 8, s));
  cm_launch_k_s5c_finalize(d, n2, s, (c->opt_coop & 8) != 0 && !c->p.split);
  if (c->use_perm) cm_launch_k_sort_lists(d, 1, s);  // long draft-mapping lists, before S6a pairs them
  mark(c, "s5c_accept");
  // S6: best pair, sampling of multi-mappers, records
  if (c->p.sam) {  // per-slot record / CIGAR / MD pools and the backtrack cells of one alignment per pair
    // the slot pools hold the whole batch (cmgpu_map_resident sized them); this range's slots start at slot0
    const uint64_t slots = c->single ? n : n2, slot0 = c->single ? rlo : 2ull * rlo;
    const uint32_t md_cap = c->sam_md_cap;
    const uint32_t zw = (2 * c->p.e + 2 <= 18) ? 5 : 8;
    if (c->sam_z.ensure((size_t)c->max_read_len * zw * 4 * n + 16)) { cm_set_error(c, "out of device memory (SAM buffers)"); return CMGPU_ENOMEM; }
    HIPCHECK(c, hipMemsetAsync((uint8_t *)c->sam_rec.p + slot0 * 40, 0, slots * 40, s));
    d.sam_rec = (uint8_t *)c->sam_rec.p + slot0 * 40; d.sam_cigar = (uint32_t *)c->sam_cigar.p + slot0 * CM_SAM_CIGAR_CAP;
    d.sam_md = (uint8_t *)c->sam_md.p + slot0 * md_cap;
    d.sam_z = (uint32_t *)c->sam_z.p; d.sam_md_cap = md_cap;
  }
  if (c->p.sam) cm_launch_k_s6a_pair_sam(d, n, s); else cm_launch_k_s6a_pair(d, n, s, (c->opt_coop & 16) != 0);
  mark(c, "s6a_pairing");
  const uint32_t n_chunks = cm_num_chunks_host(n, (uint32_t)c->p.ref_batch, (uint32_t)c->p.grain);
  cm_launch_k_s6b_sample(d, n_chunks, s);
  if (c->p.sam) cm_launch_k_s6c_multi_sam(d, n, s); else cm_launch_k_s6c_multi(d, n, s, (c->opt_coop & 16) != 0);
  mark(c, "s6bc_multimappers");
  cm_launch_k_stats(d, n, (unsigned long long *)c->partials.p, s);
  unsigned long long hst[CM_ST_N];
  uint32_t h_cls[CM_HV_LISTS];
  HIPCHECK(c, hipMemcpyAsync(hst, c->stats.p, sizeof(hst), hipMemcpyDeviceToHost, s));
  HIPCHECK(c, hipMemcpyAsync(h_cls, c->hv_cnt.p, sizeof(h_cls), hipMemcpyDeviceToHost, s));
  HIPCHECK(c, cm_stream_sync(s));
  mark(c, "stats");
  HIPCHECK(c, cm_stream_sync(s));
  {  // speculative launch set: a class with items whose kernels were not launched -> the range again with every class on
    unsigned long long seen = 0;
    for (uint32_t l = 6; l < CM_HV_LISTS; ++l) if (h_cls[l]) seen |= 1ull << (l == 31 ? 23 : l);  // (list 31's kernels are launched with list 23's)
    const unsigned long long launched = c->cls_all || !c->opt_spec ? ~0ull : c->cls_seen;
    if (!(spec && hst[CM_ST_ABORT]) && (seen & ~launched)) {
      c->cls_all = true;
      const int rc2 = map_range(c, rlo, rhi, k_out, stats, allow_spec);
      c->cls_all = false;
      return rc2;
    }
    if (!(spec && hst[CM_ST_ABORT])) {
      unsigned long long keep = 0;
      for (uint32_t l = 0; l < 64; ++l) {
        if ((seen >> l) & 1ull) c->cls_age[l] = 32; else if (c->cls_age[l]) --c->cls_age[l];
        if (c->cls_age[l]) keep |= 1ull << l;
      }
      c->cls_seen = keep;
    }
  }
  if (spec && hst[CM_ST_ABORT]) {  // this batch needs more room than the previous one left: again, with its own totals
    c->pred_m_ok = false;
    return map_range(c, rlo, rhi, k_out, stats, false);
  }
  c->pred_m_ok = true;
  c->pred_n = n;
  c->rs_pool_want = hst[CM_ST_POOL];
  if (cm_debug_pool()) fprintf(stderr, "pool: cap %u entries, asked %llu\n", c->rs_pool_cap, (unsigned long long)hst[CM_ST_POOL]);
  if (hst[CM_ST_ERR]) { cm_set_error(c, "internal device error flag " + std::to_string((unsigned long long)hst[CM_ST_ERR])); return CMGPU_ECAPACITY; }
  *k_out = hst[CM_ST_RECORDS];
  c->last_range_lo = rlo; c->last_range_hi = rhi;
  c->last_n_mm = n_mm; c->last_n_hits = n_hits; c->last_n_cand_cap = n_m;
  if (stats) {
    stats->num_candidates += hst[CM_ST_CAND];
    stats->num_mappings += hst[CM_ST_MAPPINGS];
    stats->num_mapped_reads += hst[CM_ST_MAPPED];
    stats->num_uniquely_mapped_reads += hst[CM_ST_UNIQ];
    stats->num_minimizers += n_mm;
    stats->probe_steps += hst[CM_ST_PROBE_STEPS];
    stats->occurrences_read += hst[CM_ST_OCC];
    stats->num_pairs_rescued += hst[CM_ST_RESCUED];
    stats->num_multi_mappers += hst[CM_ST_MULTI];
    stats->num_barcode_in_whitelist += hst[CM_ST_BC_INWL];
    stats->num_corrected_barcode += hst[CM_ST_BC_CORR];
  }
  return CMGPU_OK;
}

// pairs [lo, hi): as one range, or -- when a dense intermediate would pass the item limit -- as two halves cut on a
// reference-batch boundary (the multi-mapper sampling is defined per reference batch, so the records do not change)
static int map_split(cmgpu_ctx *c, uint32_t lo, uint32_t hi, uint64_t *k_total, cmgpu_stats *stats) {
  uint64_t k = 0;
  int rc = map_range(c, lo, hi, &k, stats);
  const uint32_t rb = (uint32_t)c->p.ref_batch, n = hi - lo;
  // (round 6, tried and removed: halving a range that got "out of device memory" like one that passes the item limit -- the buffers of a
  //  context only grow, the other lanes keep theirs, and a 16 M-pair call on the profile-2 genome failed just the same: DESIGN.md 9-4)
  if (rc != CM_RC_SPLIT) { *k_total += k; return rc; }
  if (n <= rb) {
    cm_set_error(c, "a reference batch of " + std::to_string(n) + " pairs needs more than " + std::to_string((unsigned long long)c->opt_item_limit) +
                        " entries in one intermediate array (hits / candidates)");
    return CMGPU_ECAPACITY;
  }
  uint32_t half = (n / 2) / rb * rb;
  if (half == 0) half = rb;
  rc = map_split(c, lo, lo + half, k_total, stats);
  if (rc) return rc;
  return map_split(c, lo + half, hi, k_total, stats);
}

// A further lane of `c`: a context of its own (streams, events, intermediates, counters) over c's index and reference,
// which sees c's resident batch and writes into c's record arrays.  The views are refreshed before every use -- c's
// buffers move when they grow or when parked batches change places.
static int lane_prepare(cmgpu_ctx *c, size_t i) {
  while (c->lanes.size() <= i) {
    cmgpu_ctx *l = nullptr;
    const int rc = cmgpu_create_shared(c, &l);
    if (rc) { cm_set_error(c, std::string("lane context: ") + cmgpu_last_error(nullptr)); return rc; }
    c->lanes.push_back(l);
  }
  cmgpu_ctx *l = c->lanes[i];
  auto view = [](DevBuf &dst, const DevBuf &src) { dst.p = src.p; dst.cap = src.cap; dst.owned = false; };
  view(l->bkt_fast, c->bkt_fast); l->fmask = c->fmask;
  view(l->ref_planes, c->ref_planes); l->ref_pl_words = c->ref_pl_words; l->opt_planes = c->opt_planes; l->opt_long_fused = c->opt_long_fused;
  view(l->rb0, c->rb0); view(l->rb1, c->rb1); view(l->ro0, c->ro0); view(l->ro1, c->ro1);
  view(l->rec, c->rec); view(l->rec_ok, c->rec_ok);
  view(l->bcb, c->bcb); view(l->bcq, c->bcq); view(l->bco, c->bco); view(l->bc_key, c->bc_key); view(l->bc_ok, c->bc_ok);
  view(l->wl, c->wl); view(l->pow10_tab, c->pow10_tab);
  view(l->sam_rec, c->sam_rec); view(l->sam_cigar, c->sam_cigar); view(l->sam_md, c->sam_md);
  l->wl_mask = c->wl_mask; l->wl_size = c->wl_size; l->bc_len = c->bc_len; l->wl_num_sample = c->wl_num_sample;
  l->n_pairs = c->n_pairs; l->first_read_id = c->first_read_id; l->bases0 = c->bases0; l->bases1 = c->bases1;
  l->max_read_len = c->max_read_len; l->has_barcodes = c->has_barcodes; l->single = c->single;
  l->sam_slots = c->sam_slots; l->sam_md_cap = c->sam_md_cap;
  l->opt_probe_variant = c->opt_probe_variant; l->opt_mm_chunks = c->opt_mm_chunks; l->opt_prep_kernel = c->opt_prep_kernel;
  l->opt_s3b_cap = c->opt_s3b_cap; l->opt_prep_tile_reads = c->opt_prep_tile_reads; l->opt_item_limit = c->opt_item_limit; l->opt_heavy_last = c->opt_heavy_last; l->opt_heavy_mid = c->opt_heavy_mid; l->opt_coop = c->opt_coop; l->opt_coop_rb = c->opt_coop_rb; l->opt_spec = c->opt_spec;
  for (int q = 0; q < 3; ++q) l->opt_heavy_max[q] = c->opt_heavy_max[q];
  return CMGPU_OK;
}

extern "C" int cmgpu_map_resident(cmgpu_ctx *c, uint64_t *n_out, cmgpu_stats *stats) {
  if (!c) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  const uint32_t n = c->n_pairs;
  c->n_ev = 0;
  c->n_records = 0;
  c->batch_exchanged = false;
  if (n_out) *n_out = 0;
  if (n == 0) return CMGPU_OK;
  if (c->max_read_len > CM_MAX_READ_LEN) {  // LDS staging of a block's reads (cm_kernels.hip: staging_geometry)
    cm_set_error(c, "reads longer than " + std::to_string(CM_MAX_READ_LEN) + " bases are not supported");
    return CMGPU_EINVAL;
  }
  if (cm_rec_slots(c) > 0xfffffff0ull) {  // record slots are addressed with 32 bits (compaction scans, store indices)
    cm_set_error(c, "batch too large: n_pairs * max_num_best_mappings must stay below 2^32");
    return CMGPU_EINVAL;
  }
  // per-pair outputs of the whole batch; the intermediates are sized per range
  if (c->rec.ensure(cm_rec_slots(c) * 24) || c->rec_ok.ensure(cm_rec_slots(c))) { cm_set_error(c, "out of device memory (records)"); return CMGPU_ENOMEM; }
  if (c->has_barcodes && (c->bc_key.ensure((size_t)n * 8) || c->bc_ok.ensure(n))) { cm_set_error(c, "out of device memory (barcodes)"); return CMGPU_ENOMEM; }
  if (c->p.sam) {  // per-slot record / CIGAR / MD pools
    const uint64_t slots = c->single ? n : 2ull * n;
    const uint32_t md_cap = 2 * c->max_read_len + 16;
    if (c->sam_rec.ensure(slots * 40 + 16) || c->sam_cigar.ensure(slots * CM_SAM_CIGAR_CAP * 4 + 16) || c->sam_md.ensure(slots * md_cap + 16)) {
      cm_set_error(c, "out of device memory (SAM buffers)"); return CMGPU_ENOMEM;
    }
    c->sam_slots = slots;
    c->sam_md_cap = md_cap;
  }
  if (c->opt_planes && !c->ref_pl_words && c->ref_bytes) {  // once per reference: its bit planes (k_s5b_verify)
    const int rc = cm_build_ref_planes(c);
    if (rc == CMGPU_ENOMEM) {  // no room for them: the alignments run on the bytes (same results, slower)
      c->ref_planes.release(); c->ref_pl_words = 0; c->opt_planes = 0;
      fprintf(stderr, "chromap_amd: no device memory for the reference bit planes, verifying on bytes\n");
    } else if (rc) return rc;
  }
  // Lanes: the batch cut on reference-batch boundaries into up to opt_lanes ranges that are mapped side by side, each on
  // its own streams with its own intermediates -- the latency-bound stages of one range fill the gaps of the others'.
  const uint32_t rb = (uint32_t)c->p.ref_batch;
  uint32_t L = (uint32_t)(c->opt_lanes < 1 ? 1 : c->opt_lanes);
  const uint32_t n_rb = (n + rb - 1) / rb;
  if (L > n_rb) L = n_rb;
  if (n < (1u << 20)) L = 1;
  uint64_t k = 0;
  if (L <= 1) {
    const int rc = map_split(c, 0, n, &k, stats);
    if (rc) return rc;
  } else {
    std::vector<uint32_t> cut(L + 1, 0);
    for (uint32_t i = 1; i < L; ++i) cut[i] = (uint32_t)(((uint64_t)n_rb * i / L) * rb);
    cut[L] = n;
    for (uint32_t i = 1; i < L; ++i) { const int rc = lane_prepare(c, i - 1); if (rc) return rc; }
    std::vector<int> rcs(L, CMGPU_OK);
    std::vector<uint64_t> ks(L, 0);
    std::vector<cmgpu_stats> sts(L);
    for (cmgpu_stats &x : sts) memset(&x, 0, sizeof(x));
    std::vector<std::thread> th;
    for (uint32_t i = 1; i < L; ++i)
      th.emplace_back([&, i]() {
        cmgpu_ctx *l = c->lanes[i - 1];
        (void)hipSetDevice(l->device);
        rcs[i] = map_split(l, cut[i], cut[i + 1], &ks[i], &sts[i]);
      });
    rcs[0] = map_split(c, cut[0], cut[1], &ks[0], &sts[0]);
    for (std::thread &t : th) t.join();
    for (uint32_t i = 0; i < L; ++i) {
      if (rcs[i]) { if (i) cm_set_error(c, c->lanes[i - 1]->err); return rcs[i]; }
      k += ks[i];
      if (stats) {
        uint64_t *dst = reinterpret_cast<uint64_t *>(stats);
        const uint64_t *src = reinterpret_cast<const uint64_t *>(&sts[i]);
        for (size_t q = 0; q < sizeof(cmgpu_stats) / 8; ++q) dst[q] += src[q];
      }
    }
    c->last_range_lo = 0; c->last_range_hi = cut[1];
  }
  c->n_records = k;
  if (n_out) *n_out = k;
  return CMGPU_OK;
}

// the batch's records, compacted on the device, in one copy (the order of the pairs is kept)
__global__ void k_rec_flag2(const uint8_t *ok, uint32_t *flag, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = ok[i];
}
__global__ void k_rec_compact2(const uint8_t *__restrict__ rec, const uint8_t *__restrict__ ok, const uint32_t *__restrict__ pos,
                               uint8_t *__restrict__ dst, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !ok[i]) return;
  const uint64_t *s = reinterpret_cast<const uint64_t *>(rec + (uint64_t)i * 24);
  uint64_t *d = reinterpret_cast<uint64_t *>(dst + (uint64_t)pos[i] * 24);
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}
// the resident batch's records, compacted into `dst` on the mapping stream (pair order kept); nothing is waited for
static int compact_records(cmgpu_ctx *c, DevBuf &dst) {
  const uint32_t n = (uint32_t)cm_rec_slots(c);
  if (n == 0) return CMGPU_OK;
  { const int rc = cm_ensure_slot_scratch(c, n); if (rc) return rc; }
  if (dst.ensure((size_t)n * 24 + 16)) { cm_set_error(c, "out of device memory (records)"); return CMGPU_ENOMEM; }
  uint32_t *flag = (uint32_t *)c->scratch_a.p, *pos = (uint32_t *)c->scratch_b.p;  // free between batches
  hipStream_t s = c->stream;
  hipLaunchKernelGGL(k_rec_flag2, dim3((n + 255) / 256), dim3(256), 0, s, (const uint8_t *)c->rec_ok.p, flag, n);
  cm_scan_u32(flag, pos, n, (uint32_t *)c->scan_tmp.p, s);
  hipLaunchKernelGGL(k_rec_compact2, dim3((n + 255) / 256), dim3(256), 0, s, (const uint8_t *)c->rec.p, (const uint8_t *)c->rec_ok.p,
                     (const uint32_t *)pos, (uint8_t *)dst.p, n);
  return CMGPU_OK;
}
static int download_dense(cmgpu_ctx *c, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out) {
  if (n_out) *n_out = 0;
  if (cm_rec_slots(c) == 0) return CMGPU_OK;
  if (c->pend_count && c->stream_d2h) HIPCHECK(c, hipStreamSynchronize(c->stream_d2h));  // (a pending asynchronous download reads rec_dense)
  { const int rc = compact_records(c, c->rec_dense); if (rc) return rc; }
  hipStream_t s = c->stream;
  const uint64_t k = c->n_records;  // counted by the mapping call
  if (k > out_capacity) { cm_set_error(c, "record buffer too small"); return CMGPU_ECAPACITY; }
  if (k) { const int rc = device_to_host(c, out, c->rec_dense.p, (size_t)k * 24, s); if (rc) return rc; }
  HIPCHECK(c, cm_stream_sync(s));
  if (n_out) *n_out = k;
  return CMGPU_OK;
}

// cmgpu_map_submitted with the record download left running: the oldest submitted batch is mapped, its records are compacted, and
// their copy to `out` (page-locked memory, or the copy is not asynchronous) is queued on a copy stream of its own -- it runs under
// the NEXT batch's kernels.  cmgpu_records_wait returns when the oldest pending download is complete.
extern "C" int cmgpu_map_submitted_async(cmgpu_ctx *c, cmgpu_record *out, uint64_t out_capacity, cmgpu_stats *stats) {
  if (!c || !out) return CMGPU_EINVAL;
  if (c->sub_count == 0) { cm_set_error(c, "no submitted batch (cmgpu_submit_pairs)"); return CMGPU_EINVAL; }
  if (c->pend_count >= 2) { cm_set_error(c, "two record downloads are pending (cmgpu_records_wait)"); return CMGPU_EINVAL; }
  uint64_t k = 0;
  int rc = cmgpu_map_submitted(c, nullptr, 0, &k, stats);
  if (rc) return rc;
  if (k > out_capacity) { cm_set_error(c, "record buffer too small"); return CMGPU_ECAPACITY; }
  if (!c->stream_d2h) {
    int prio_least = 0, prio_greatest = 0;
    HIPCHECK(c, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIPCHECK(c, hipStreamCreateWithPriority(&c->stream_d2h, hipStreamNonBlocking, prio_greatest));
    HIPCHECK(c, hipEventCreateWithFlags(&c->ev_comp, hipEventDisableTiming));
    for (hipEvent_t &e : c->ev_d2h) HIPCHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const uint32_t slot = c->pend_total & 1u;
  DevBuf &dst = slot ? c->rec_dense_b : c->rec_dense;
  if (k) {
    rc = compact_records(c, dst);
    if (rc) return rc;
    HIPCHECK(c, hipEventRecord(c->ev_comp, c->stream));
    HIPCHECK(c, hipStreamWaitEvent(c->stream_d2h, c->ev_comp, 0));
    HIPCHECK(c, hipMemcpyAsync(out, dst.p, (size_t)k * 24, hipMemcpyDeviceToHost, c->stream_d2h));
    // the next mapping call's lanes write the per-pair record arrays from streams of their own: the compaction has to be through
    HIPCHECK(c, hipEventSynchronize(c->ev_comp));
  }
  HIPCHECK(c, hipEventRecord(c->ev_d2h[slot], c->stream_d2h));
  c->pend_k[slot] = k;
  ++c->pend_total; ++c->pend_count;
  return CMGPU_OK;
}
extern "C" int cmgpu_records_wait(cmgpu_ctx *c, uint64_t *n_out) {
  if (!c) return CMGPU_EINVAL;
  if (c->pend_count == 0) { cm_set_error(c, "no record download pending (cmgpu_map_submitted_async)"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  const uint32_t slot = (c->pend_total - c->pend_count) & 1u;
  HIPCHECK(c, hipEventSynchronize(c->ev_d2h[slot]));
  --c->pend_count;
  if (n_out) *n_out = c->pend_k[slot];
  return CMGPU_OK;
}

extern "C" int cmgpu_download_records(cmgpu_ctx *c, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out) {
  if (!c || !out || !n_out) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  return download_dense(c, out, out_capacity, n_out);
}

extern "C" int cmgpu_map_pairs(cmgpu_ctx *c, const cmgpu_batch *in, cmgpu_record *out, uint64_t out_capacity,
                               uint64_t *n_out, cmgpu_stats *stats) {
  int rc = cmgpu_upload_batch(c, in);
  if (rc) return rc;
  uint64_t k = 0;
  rc = cmgpu_map_resident(c, &k, stats);
  if (rc) return rc;
  if (!out) { if (n_out) *n_out = k; return CMGPU_OK; }  // records stay resident (cmgpu_store_append_resident)
  return cmgpu_download_records(c, out, out_capacity, n_out);
}

// ---------------------------------------------------------------------------------------
// one batch in flight: the caller's thread parses / packs the next batch while a worker thread
// of the library uploads and maps this one -- the role of the "load the next batch" task next to
// the mapping taskloop (chromap.h:871-877).  The ctx stays single-caller: between submit and
// wait no other call on this ctx is allowed; the batch's host buffers must stay valid.
// ---------------------------------------------------------------------------------------
extern "C" int cmgpu_map_pairs_async(cmgpu_ctx *c, const cmgpu_batch *in, cmgpu_stats *stats) {
  if (!c || !in) return CMGPU_EINVAL;
  if (c->in_flight) { cm_set_error(c, "a batch is already in flight (call cmgpu_wait first)"); return CMGPU_EINVAL; }
  const cmgpu_batch batch = *in;
  c->in_flight = true;
  c->async_rc = 0;
  c->async_n = 0;
  c->worker = std::thread([c, batch, stats]() {
    uint64_t k = 0;
    c->async_rc = cmgpu_map_pairs(c, &batch, nullptr, 0, &k, stats);
    c->async_n = k;
  });
  return CMGPU_OK;
}

extern "C" int cmgpu_wait(cmgpu_ctx *c, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out) {
  if (!c) return CMGPU_EINVAL;
  if (!c->in_flight) { cm_set_error(c, "no batch in flight"); return CMGPU_EINVAL; }
  c->worker.join();
  c->in_flight = false;
  if (n_out) *n_out = c->async_n;
  if (c->async_rc) return c->async_rc;
  if (!out) return CMGPU_OK;  // records stay resident (cmgpu_store_append_resident)
  return cmgpu_download_records(c, out, out_capacity, n_out);
}

extern "C" int cmgpu_sam_layout(const cmgpu_ctx *c, uint64_t *n_slots, uint32_t *md_cap) {
  if (!c) return CMGPU_EINVAL;
  if (n_slots) *n_slots = c->p.sam ? c->sam_slots : 0;
  if (md_cap) *md_cap = c->sam_md_cap;
  return CMGPU_OK;
}

extern "C" int cmgpu_download_sam(cmgpu_ctx *c, cmgpu_sam_record *records, uint32_t *cigar_pool, char *md_pool) {
  if (!c || !records || !cigar_pool || !md_pool) return CMGPU_EINVAL;
  if (!c->p.sam) { cm_set_error(c, "the ctx was not created with output_format = CMGPU_FORMAT_SAM"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  const uint64_t ns = c->sam_slots;
  if (ns == 0) return CMGPU_OK;
  HIPCHECK(c, hipMemcpy(records, c->sam_rec.p, ns * 40, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(cigar_pool, c->sam_cigar.p, ns * CM_SAM_CIGAR_CAP * 4, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(md_pool, c->sam_md.p, ns * c->sam_md_cap, hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < ns; ++i)
    if (records[i].valid == 2) { cm_set_error(c, "an alignment needs more than CMGPU_SAM_CIGAR_CAP CIGAR operations"); return CMGPU_ECAPACITY; }
  return CMGPU_OK;
}

extern "C" int cmgpu_download_barcode_keys(cmgpu_ctx *c, uint64_t *keys) {
  if (!c || !keys) return CMGPU_EINVAL;
  if (!c->has_barcodes) { cm_set_error(c, "the last batch had no barcodes"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  if (c->n_pairs) HIPCHECK(c, hipMemcpy(keys, c->bc_key.p, (size_t)c->n_pairs * 8, hipMemcpyDeviceToHost));
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// stage-level view of the last mapped batch for parity tests on the gfx950 build: per pair what the oracle's trace
// (oracle/chromap_oracle.h: ora_trace, same layout) records after trimming, minimizers, candidate generation +
// pair filter, verification and pairing -- so that a kernel change that breaks parity fails at the stage that broke.
// ---------------------------------------------------------------------------------------
extern "C" int cmgpu_debug_trace(cmgpu_ctx *c, cmgpu_trace *out, uint64_t capacity) {
  if (!c || !out) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  const uint32_t n = c->n_pairs;
  if (capacity < n) { cm_set_error(c, "trace buffer too small"); return CMGPU_ECAPACITY; }
  if (c->single) { cm_set_error(c, "the trace is defined for paired-end batches"); return CMGPU_EINVAL; }
  if (c->last_range_lo != 0 || c->last_range_hi != n) { cm_set_error(c, "the batch was mapped in sub-batches: only the last one's intermediates are resident"); return CMGPU_EINVAL; }
  memset(out, 0, (size_t)n * sizeof(cmgpu_trace));
  if (n == 0) return CMGPU_OK;
  const size_t n2 = 2 * (size_t)n;
  std::vector<uint32_t> rlen(n2), mm(n2), mcp(n2), mcn(n2), fcp(n2), fcn(n2), ndp(n2), ndn(n2), rep(n2);
  std::vector<int32_t> me(n2), nb(n2), se(n2), ns(n2), pmin(n), pnb(n), psec(n), pns(n);
  std::vector<uint8_t> f0(n);
#define DL(v, buf) HIPCHECK(c, hipMemcpy(v.data(), c->buf.p, v.size() * sizeof(v[0]), hipMemcpyDeviceToHost));
  DL(rlen, rlen) DL(mm, mm_cnt) DL(mcp, mcp) DL(mcn, mcn) DL(fcp, fcp) DL(fcn, fcn) DL(ndp, ndp) DL(ndn, ndn) DL(rep, rep_len)
  DL(me, min_err) DL(nb, n_best) DL(se, second_err) DL(ns, n_second) DL(pmin, pe_min) DL(pnb, pe_nbest) DL(psec, pe_second) DL(pns, pe_nsecond)
  DL(f0, force0)
#undef DL
  for (uint32_t i = 0; i < n; ++i) {
    cmgpu_trace &t = out[i];
    const size_t a = 2 * (size_t)i, b = a + 1;
    if (rlen[a] == 0 && rlen[b] == 0) continue;  // dropped by the length filter: nothing is traced (the entry stays zero)
    t.len1 = rlen[a]; t.len2 = rlen[b]; t.n_mm1 = mm[a]; t.n_mm2 = mm[b]; t.force_mapq = -1;
    if (mm[a] == 0 || mm[b] == 0) continue;
    uint32_t nc1 = mcp[a] + mcn[a], nc2 = mcp[b] + mcn[b];
    if (nc1 > 0 && nc2 > 0 && !c->p.split) { nc1 = fcp[a] + fcn[a]; nc2 = fcp[b] + fcn[b]; }
    t.n_cand1 = nc1; t.n_cand2 = nc2; t.rep1 = rep[a]; t.rep2 = rep[b];
    if (!(nc1 > 0 && nc2 > 0)) continue;
    t.n_draft1 = ndp[a] + ndn[a]; t.n_draft2 = ndp[b] + ndn[b];
    t.min_err1 = me[a]; t.min_err2 = me[b]; t.nbest1 = nb[a]; t.nbest2 = nb[b];
    t.second1 = se[a]; t.second2 = se[b]; t.nsecond1 = ns[a]; t.nsecond2 = ns[b];
    if (!(t.n_draft1 > 0 && t.n_draft2 > 0)) continue;
    t.min_sum = pmin[i]; t.nbest = pnb[i]; t.second_sum = psec[i]; t.nsecond = pns[i];
    t.force_mapq = c->p.split ? -1 : (f0[i] ? 0 : -1);
  }
  return CMGPU_OK;
}

// a per-read (2 n entries) or per-pair (n entries) 32-bit array of the last mapped batch by name: list lengths for the
// distribution tools (tools/list_hist.py) -- measurement aid, declared in include/chromap_amd_debug.h
extern "C" int cmgpu_debug_array(cmgpu_ctx *c, const char *name, uint32_t *out, uint64_t capacity, uint64_t *n_out) {
  if (!c || !name || !out) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  const std::string nm(name);
  struct { const char *n; DevBuf *b; int per_pair; } tab[] = {
    {"rlen", &c->rlen, 0}, {"mm_cnt", &c->mm_cnt, 0}, {"hit_tot", &c->hit_tot, 0}, {"ncp", &c->ncp, 0}, {"ncn", &c->ncn, 0},
    {"resc_p", &c->resc_p, 0}, {"resc_n", &c->resc_n, 0}, {"mcp", &c->mcp, 0}, {"mcn", &c->mcn, 0}, {"fcp", &c->fcp, 0},
    {"fcn", &c->fcn, 0}, {"ndp", &c->ndp, 0}, {"ndn", &c->ndn, 0}, {"nv", &c->nv, 0}, {"pe_nbest", &c->pe_nbest, 1}};
  for (auto &t : tab)
    if (nm == t.n) {
      const uint64_t n = t.per_pair ? c->n_pairs : 2ull * c->n_pairs;
      if (n_out) *n_out = n;
      if (capacity < n || !t.b->p || t.b->cap < n * 4) { cm_set_error(c, "debug array: buffer too small or array not resident"); return CMGPU_ECAPACITY; }
      HIPCHECK(c, hipMemcpy(out, t.b->p, n * 4, hipMemcpyDeviceToHost));
      return CMGPU_OK;
    }
  cm_set_error(c, "debug array: unknown name " + nm);
  return CMGPU_EINVAL;
}

// one read's minimizers of the last mapped batch: (hash, position << 1 | strand) in emission order
extern "C" int cmgpu_debug_minimizers(cmgpu_ctx *c, uint32_t read, uint64_t *hash_out, uint32_t *ps_out, uint32_t capacity, uint32_t *n_out) {
  if (!c || !n_out || read >= 2 * c->n_pairs) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  if (c->last_range_lo != 0 || c->last_range_hi != c->n_pairs) { cm_set_error(c, "the batch was mapped in sub-batches"); return CMGPU_EINVAL; }
  uint32_t cnt = 0, off = 0;
  HIPCHECK(c, hipMemcpy(&cnt, (const uint32_t *)c->mm_cnt.p + read, 4, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(&off, (const uint32_t *)c->mm_off.p + read, 4, hipMemcpyDeviceToHost));
  *n_out = cnt;
  if (cnt > capacity) { cm_set_error(c, "minimizer buffer too small"); return CMGPU_ECAPACITY; }
  if (cnt && hash_out) HIPCHECK(c, hipMemcpy(hash_out, (const uint64_t *)c->mm_hash.p + off, (size_t)cnt * 8, hipMemcpyDeviceToHost));
  if (cnt && ps_out) HIPCHECK(c, hipMemcpy(ps_out, (const uint32_t *)c->mm_ps.p + off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  return CMGPU_OK;
}
// all of them at once: counts / offsets per read (2 n_pairs each) and the dense arrays (n_total entries)
extern "C" int cmgpu_debug_minimizers_all(cmgpu_ctx *c, uint32_t *cnt_out, uint32_t *off_out, uint64_t *hash_out, uint32_t *ps_out, uint64_t capacity,
                                          uint64_t *n_total) {
  if (!c || !cnt_out || !off_out || !n_total) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  if (c->last_range_lo != 0 || c->last_range_hi != c->n_pairs) { cm_set_error(c, "the batch was mapped in sub-batches"); return CMGPU_EINVAL; }
  const size_t n2 = 2 * (size_t)c->n_pairs;
  *n_total = c->last_n_mm;
  if (n2 == 0) return CMGPU_OK;
  HIPCHECK(c, hipMemcpy(cnt_out, c->mm_cnt.p, n2 * 4, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(off_out, c->mm_off.p, n2 * 4, hipMemcpyDeviceToHost));
  if (c->last_n_mm > capacity) { cm_set_error(c, "minimizer buffer too small"); return CMGPU_ECAPACITY; }
  if (c->last_n_mm && hash_out) HIPCHECK(c, hipMemcpy(hash_out, c->mm_hash.p, c->last_n_mm * 8, hipMemcpyDeviceToHost));
  if (c->last_n_mm && ps_out) HIPCHECK(c, hipMemcpy(ps_out, c->mm_ps.p, c->last_n_mm * 4, hipMemcpyDeviceToHost));
  return CMGPU_OK;
}

extern "C" int cmgpu_last_timings(const cmgpu_ctx *c, const char **names, float *ms, int cap) {
  if (!c) return 0;
  int k = 0;
  for (int i = 1; i < c->n_ev && k < cap; ++i) {
    float t = 0;
    if (hipEventElapsedTime(&t, c->ev[i - 1], c->ev[i]) != hipSuccess) t = -1;
    names[k] = c->ev_name[i];
    ms[k] = t;
    ++k;
  }
  return k;
}

extern "C" int cmgpu_download_batch(cmgpu_ctx *c, char *r1, uint32_t *o1, char *r2, uint32_t *o2) {
  if (!c) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  const uint32_t n = c->n_pairs;
  if (r1) HIPCHECK(c, hipMemcpy(r1, c->rb0.p, c->bases0, hipMemcpyDeviceToHost));
  if (r2) HIPCHECK(c, hipMemcpy(r2, c->rb1.p, c->bases1, hipMemcpyDeviceToHost));
  if (o1) HIPCHECK(c, hipMemcpy(o1, c->ro0.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
  if (o2) HIPCHECK(c, hipMemcpy(o2, c->ro1.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// kernel-only measurement of the index probe
// ---------------------------------------------------------------------------------------
static int probe_bench_impl(cmgpu_ctx *c, const uint64_t *hashes, uint64_t n, int repeat, int variant, double *avg_ms,
                            uint64_t *probe_steps, uint64_t *hits, uint64_t *occurrences);
extern "C" int cmgpu_probe_bench(cmgpu_ctx *c, const uint64_t *hashes, uint64_t n, int repeat, double *avg_ms,
                                 uint64_t *probe_steps, uint64_t *hits, uint64_t *occurrences) {
  return probe_bench_impl(c, hashes, n, repeat, 0, avg_ms, probe_steps, hits, occurrences);
}
extern "C" int cmgpu_probe_bench_variant(cmgpu_ctx *c, uint64_t n, int repeat, int lookups_per_lane, int pair_prefetch, double *avg_ms,
                                         uint64_t *probe_steps, uint64_t *hits) {
  if (lookups_per_lane != 1 && lookups_per_lane != 2 && lookups_per_lane != 4 && lookups_per_lane != 8) return CMGPU_EINVAL;
  // pair_prefetch bit 0: second probe step requested with the first; bit 1: the file's table even when a re-hashed one is resident
  return probe_bench_impl(c, nullptr, n, repeat, lookups_per_lane | ((pair_prefetch & 1) ? 16 : 0) | ((pair_prefetch & 2) ? 32 : 0), avg_ms, probe_steps, hits, nullptr);
}
static int probe_bench_impl(cmgpu_ctx *c, const uint64_t *hashes, uint64_t n, int repeat, int variant, double *avg_ms,
                            uint64_t *probe_steps, uint64_t *hits, uint64_t *occurrences) {
  if (!c || n == 0 || n > 0xffffff00ull || repeat < 1) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  if (hashes) {
    if (c->mm_hash.ensure(n * 8)) { cm_set_error(c, "out of device memory (probe hashes)"); return CMGPU_ENOMEM; }
    HIPCHECK(c, hipMemcpy(c->mm_hash.p, hashes, n * 8, hipMemcpyHostToDevice));
  } else if (c->mm_hash.cap < n * 8) {
    cm_set_error(c, "no resident hashes");
    return CMGPU_EINVAL;
  }
  if (c->pr_val.ensure(n * 8) || c->pr_kind.ensure(n)) { cm_set_error(c, "out of device memory (probe results)"); return CMGPU_ENOMEM; }
  hipStream_t s = c->stream;
  unsigned long long *ctr = (unsigned long long *)c->stats.p;
  HIPCHECK(c, hipMemsetAsync(ctr, 0, CM_ST_N * 8, s));
  // warm-up + counted launch
  if (c->partials.ensure(cm_probe_partial_words((uint32_t)n) * 8)) { cm_set_error(c, "out of device memory (partials)"); return CMGPU_ENOMEM; }
  const bool file_layout = (variant & 32) != 0 || !c->fmask;  // + 32: the file's table even when a re-hashed one is resident
  variant &= ~32;
  const uint64_t *tab = (const uint64_t *)(file_layout ? c->bkt.p : c->bkt_fast.p);
  const uint32_t tmask = file_layout ? c->bmask : c->fmask;
  cm_launch_k_probe(tab, tmask, (const uint64_t *)c->mm_hash.p, (uint64_t *)c->pr_val.p,
                    (uint8_t *)c->pr_kind.p, (uint32_t)n, c->partials.p, ctr + CM_ST_PROBE_STEPS, s, variant);
  unsigned long long h[CM_ST_N];
  HIPCHECK(c, hipMemcpyAsync(h, ctr, sizeof(h), hipMemcpyDeviceToHost, s));
  HIPCHECK(c, cm_stream_sync(s));
  HIPCHECK(c, hipEventRecord(c->ev[0], s));
  for (int i = 0; i < repeat; ++i)
    cm_launch_k_probe(tab, tmask, (const uint64_t *)c->mm_hash.p, (uint64_t *)c->pr_val.p,
                      (uint8_t *)c->pr_kind.p, (uint32_t)n, nullptr, nullptr, s, variant);
  HIPCHECK(c, hipEventRecord(c->ev[1], s));
  HIPCHECK(c, hipEventSynchronize(c->ev[1]));
  float ms = 0;
  HIPCHECK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  if (avg_ms) *avg_ms = ms / repeat;
  if (probe_steps) *probe_steps = h[CM_ST_PROBE_STEPS];
  if (hits) *hits = h[CM_ST_PROBE_HITS];
  if (occurrences) *occurrences = 0;
  return CMGPU_OK;
}

extern "C" int cmgpu_reference_lengths(cmgpu_ctx *c, uint32_t *lengths, uint32_t capacity, uint32_t *n_sequences) {
  if (!c) return CMGPU_EINVAL;
  if (n_sequences) *n_sequences = c->n_seq;
  for (uint32_t i = 0; lengths && i < c->n_seq && i < capacity; ++i) lengths[i] = c->h_ref_len[i];
  return CMGPU_OK;
}

extern "C" int cmgpu_export_reference(cmgpu_ctx *c, uint32_t seq, char *out, uint32_t capacity) {
  if (!c || seq >= c->n_seq || !out || capacity < c->h_ref_len[seq]) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  HIPCHECK(c, hipMemcpy(out, (const uint8_t *)c->ref.p + c->h_ref_off[seq], c->h_ref_len[seq], hipMemcpyDeviceToHost));
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// exports used by bench.py / tests to hand the device-built index to the CPU baseline
// ---------------------------------------------------------------------------------------
extern "C" int cmgpu_index_info(cmgpu_ctx *c, int32_t *kmer_size, int32_t *window_size, uint32_t *n_buckets,
                                uint32_t *n_occurrences, uint64_t *n_minimizers, uint64_t *n_keys) {
  if (!c) return CMGPU_EINVAL;
  if (kmer_size) *kmer_size = c->p.k;
  if (window_size) *window_size = c->p.w;
  if (n_buckets) *n_buckets = c->bmask + 1;
  if (n_occurrences) *n_occurrences = c->n_occ;
  if (n_minimizers) *n_minimizers = c->synth_n_minimizers;
  if (n_keys) *n_keys = c->synth_n_keys;
  return CMGPU_OK;
}

// buckets_out: 2*n_buckets uint64 ({key,val} interleaved, CM_EMPTY_KEY = all ones marks an
// empty bucket); occurrences_out: n_occurrences uint64
extern "C" int cmgpu_export_index(cmgpu_ctx *c, uint64_t *buckets_out, uint64_t *occurrences_out) {
  if (!c || !buckets_out) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  HIPCHECK(c, hipMemcpy(buckets_out, c->bkt.p, ((size_t)c->bmask + 1) * 16, hipMemcpyDeviceToHost));
  if (occurrences_out && c->n_occ) HIPCHECK(c, hipMemcpy(occurrences_out, c->occ.p, (size_t)c->n_occ * 8, hipMemcpyDeviceToHost));
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// dense copy of the resident records into a caller-provided DEVICE buffer (e.g. a torch
// tensor's data_ptr) for the multi-GPU record exchange
// ---------------------------------------------------------------------------------------
__global__ void k_rec_flag(const uint8_t *ok, uint32_t *flag, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flag[i] = ok[i];
}
__global__ void k_rec_compact(const uint8_t *__restrict__ rec, const uint8_t *__restrict__ ok,
                              const uint32_t *__restrict__ pos, uint8_t *__restrict__ dst, uint32_t n, uint64_t cap) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !ok[i]) return;
  const uint32_t o = pos[i];
  if (o >= cap) return;
  const uint64_t *s = reinterpret_cast<const uint64_t *>(rec + (uint64_t)i * 24);
  uint64_t *d = reinterpret_cast<uint64_t *>(dst + (uint64_t)o * 24);
  d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

extern "C" int cmgpu_records_to_device(cmgpu_ctx *c, void *device_dst, uint64_t capacity, uint64_t *n_out) {
  if (!c || !device_dst || !n_out) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  const uint32_t n = (uint32_t)cm_rec_slots(c);
  *n_out = 0;
  if (n == 0) return CMGPU_OK;
  { const int rc = cm_ensure_slot_scratch(c, n); if (rc) return rc; }
  uint32_t *flag = (uint32_t *)c->scratch_a.p, *pos = (uint32_t *)c->scratch_b.p;  // free between batches
  hipLaunchKernelGGL(k_rec_flag, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint8_t *)c->rec_ok.p, flag, n);
  cm_scan_u32(flag, pos, n, (uint32_t *)c->scan_tmp.p, c->stream);
  hipLaunchKernelGGL(k_rec_compact, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint8_t *)c->rec.p,
                     (const uint8_t *)c->rec_ok.p, (const uint32_t *)pos, (uint8_t *)device_dst, n, capacity);
  uint32_t k = 0;
  HIPCHECK(c, hipMemcpyAsync(&k, pos + n, 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(c, cm_stream_sync(c->stream));
  *n_out = k;
  if (k > capacity) { cm_set_error(c, "device record buffer too small"); return CMGPU_ECAPACITY; }
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// HBM random-gather microbenchmark on the resident table (the measured ceiling the
// probe kernel is compared with): n independent accesses at pseudo-random buckets, LOADS of them
// requested back to back per lane before any is used.  WIDE = false: one 16-byte bucket per access;
// WIDE = true: the whole 64-byte sector around it (four 16-byte loads), i.e. every fetched byte used.
// ---------------------------------------------------------------------------------------
template <int LOADS, bool WIDE>
__global__ __launch_bounds__(256) void k_gather(const uint64_t *__restrict__ bkt, uint32_t bmask, uint64_t n,
                                                uint64_t seed, unsigned long long *__restrict__ sink) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t acc = 0;
  ulonglong2 kv[LOADS * (WIDE ? 4 : 1)];
#pragma unroll
  for (int j = 0; j < LOADS; ++j) {
    const uint64_t i = t * LOADS + j;
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    const uint32_t b = (uint32_t)x & bmask;
    if (WIDE) {
      const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(bkt + 2 * (uint64_t)(b & ~3u));
#pragma unroll
      for (int q = 0; q < 4; ++q) kv[4 * j + q] = i < n ? p[q] : make_ulonglong2(0, 0);
    } else {
      kv[j] = i < n ? *reinterpret_cast<const ulonglong2 *>(bkt + 2 * (uint64_t)b) : make_ulonglong2(0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < LOADS * (WIDE ? 4 : 1); ++j) acc ^= kv[j].x + kv[j].y;
  if (acc == 0x123456789abcdefull) atomicAdd(sink, 1ull);  // keeps the loads alive
}

template <int LOADS, bool WIDE>
static void gather_launch(cmgpu_ctx *c, uint64_t n, uint64_t seed) {
  const unsigned blocks = (unsigned)((n + 256ull * LOADS - 1) / (256ull * LOADS));
  hipLaunchKernelGGL((k_gather<LOADS, WIDE>), dim3(blocks), dim3(256), 0, c->stream, (const uint64_t *)c->bkt.p, c->bmask, n, seed,
                     (unsigned long long *)c->stats.p + CM_ST_TOTAL);
}
static bool gather_dispatch(cmgpu_ctx *c, uint64_t n, uint64_t seed, int loads, int wide) {
#define CM_G(L_) if (loads == L_) { if (wide) gather_launch<L_, true>(c, n, seed); else gather_launch<L_, false>(c, n, seed); return true; }
  CM_G(1) CM_G(2) CM_G(4) CM_G(8)
  if (loads == 16 && !wide) { gather_launch<16, false>(c, n, seed); return true; }
#undef CM_G
  return false;
}

// loads_per_lane 1 / 2 / 4 / 8 / 16 (16 only with 16-byte accesses); access_bytes 16 or 64
extern "C" int cmgpu_gather_sweep(cmgpu_ctx *c, uint64_t n, int repeat, int loads_per_lane, int access_bytes, double *avg_ms) {
  if (!c || n == 0 || repeat < 1 || !avg_ms || (access_bytes != 16 && access_bytes != 64)) return CMGPU_EINVAL;
  HIPCHECK(c, cm_enter(c));
  hipStream_t s = c->stream;
  if (!gather_dispatch(c, n, 1ull, loads_per_lane, access_bytes == 64)) { cm_set_error(c, "unsupported gather shape"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_stream_sync(s));
  HIPCHECK(c, hipEventRecord(c->ev[0], s));
  for (int i = 0; i < repeat; ++i) gather_dispatch(c, n, (uint64_t)(i + 2) * 7919ull, loads_per_lane, access_bytes == 64);
  HIPCHECK(c, hipEventRecord(c->ev[1], s));
  HIPCHECK(c, hipEventSynchronize(c->ev[1]));
  float ms = 0;
  HIPCHECK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  *avg_ms = ms / repeat;
  return CMGPU_OK;
}

extern "C" int cmgpu_gather_bench(cmgpu_ctx *c, uint64_t n, int repeat, double *avg_ms) {
  return cmgpu_gather_sweep(c, n, repeat, 4, 16, avg_ms);
}

// ---------------------------------------------------------------------------------------
// single-cell barcodes (K6)
// ---------------------------------------------------------------------------------------
extern "C" int cmgpu_set_whitelist(cmgpu_ctx *c, const uint64_t *keys, uint32_t n_keys, uint32_t barcode_length) {
  if (!c || !keys || n_keys == 0 || barcode_length == 0 || barcode_length > 32) { cm_set_error(c, "bad whitelist"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  uint32_t nb = 16;
  while (nb < 2ull * n_keys + 16) nb <<= 1;
  std::vector<uint64_t> tab((size_t)nb * 2);
  for (uint32_t i = 0; i < nb; ++i) { tab[2 * (size_t)i] = ~0ull; tab[2 * (size_t)i + 1] = 0; }
  uint32_t size = 0;
  for (uint32_t i = 0; i < n_keys; ++i) {
    const uint64_t x = keys[i] * 0x9E3779B97F4A7C15ull;
    uint32_t b = (uint32_t)(x >> 32) & (nb - 1);
    while (tab[2 * (size_t)b] != ~0ull && tab[2 * (size_t)b] != keys[i]) b = (b + 1) & (nb - 1);
    if (tab[2 * (size_t)b] == ~0ull) { tab[2 * (size_t)b] = keys[i]; ++size; }
  }
  // pow(10, -q/10) for q = 0..80, libm on the host (chromap.cc:639-640, 688-689)
  std::vector<double> pw(81);
  for (int q = 0; q <= 80; ++q) pw[q] = pow(10.0, ((-q) / 10.0));
  if (c->wl.ensure(tab.size() * 8) || c->pow10_tab.ensure(pw.size() * 8) || c->wl_num.ensure(8)) { cm_set_error(c, "out of device memory (whitelist)"); return CMGPU_ENOMEM; }
  HIPCHECK(c, hipMemcpy(c->wl.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(c->pow10_tab.p, pw.data(), pw.size() * 8, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemset(c->wl_num.p, 0, 8));
  c->wl_mask = nb - 1;
  c->wl_size = size;
  c->bc_len = barcode_length;
  c->wl_num_sample = 0;
  return CMGPU_OK;
}

// abundance counting over device-resident barcodes, in reference batches, until the sample is
// large enough (chromap.cc:494-540: stops after the batch that reaches 20 M whitelisted barcodes)
static int bc_abundance_run(cmgpu_ctx *c, const uint8_t *dbases, const uint32_t *doffs, uint32_t n, bool *done) {
  const uint64_t max_sample = 20000000ull;   // initial_num_sample_barcodes_ (chromap.h:211)
  const uint32_t batch = (uint32_t)(c->p.ref_batch > 0 ? c->p.ref_batch : 500000);
  unsigned long long ns = c->wl_num_sample;
  int rc = CMGPU_OK;
  *done = ns >= max_sample;
  for (uint32_t b0 = 0; b0 < n && !*done; b0 += batch) {
    const uint32_t bn = n - b0 < batch ? n - b0 : batch;
    cm_launch_k_bc_abundance(dbases, doffs, b0, b0 + bn, (uint64_t *)c->wl.p, c->wl_mask, (unsigned long long *)c->wl_num.p, c->stream);
    hipError_t e = hipMemcpyAsync(&ns, c->wl_num.p, 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = cm_stream_sync(c->stream);
    if (e != hipSuccess) { cm_set_error(c, std::string("barcode abundance: ") + hipGetErrorString(e)); rc = CMGPU_EHIP; break; }
    if (!c->skip_barcode_check && ns * 20 < bn) {  // chromap.cc:523-533
      cm_set_error(c, "Less than 5% barcodes can be found or corrected based on the barcode whitelist.");
      rc = CMGPU_EINVAL;
      break;
    }
    if (ns >= max_sample) *done = true;
  }
  c->wl_num_sample = ns;
  return rc;
}

// whitelist table with the abundances of the pre-pass, copied from another context (the contexts of a multi-GPU run:
// the pre-pass streams the barcode file through one GPU, every GPU corrects barcodes against the same counts)
extern "C" int cmgpu_copy_whitelist(cmgpu_ctx *dst, cmgpu_ctx *src) {
  if (!dst || !src || src->wl_size == 0) return CMGPU_EINVAL;
  const size_t bytes = ((size_t)src->wl_mask + 1) * 16;
  std::vector<uint64_t> tab(bytes / 8);
  std::vector<double> pw(81);
  HIPCHECK(src, cm_enter(src));
  HIPCHECK(src, hipMemcpy(tab.data(), src->wl.p, bytes, hipMemcpyDeviceToHost));
  HIPCHECK(src, hipMemcpy(pw.data(), src->pow10_tab.p, pw.size() * 8, hipMemcpyDeviceToHost));
  HIPCHECK(dst, cm_enter(dst));
  if (dst->wl.ensure(bytes) || dst->pow10_tab.ensure(pw.size() * 8) || dst->wl_num.ensure(8)) { cm_set_error(dst, "out of device memory (whitelist)"); return CMGPU_ENOMEM; }
  HIPCHECK(dst, hipMemcpy(dst->wl.p, tab.data(), bytes, hipMemcpyHostToDevice));
  HIPCHECK(dst, hipMemcpy(dst->pow10_tab.p, pw.data(), pw.size() * 8, hipMemcpyHostToDevice));
  const unsigned long long ns = src->wl_num_sample;
  HIPCHECK(dst, hipMemcpy(dst->wl_num.p, &ns, 8, hipMemcpyHostToDevice));
  dst->wl_mask = src->wl_mask; dst->wl_size = src->wl_size; dst->bc_len = src->bc_len; dst->wl_num_sample = src->wl_num_sample;
  dst->skip_barcode_check = src->skip_barcode_check;
  return CMGPU_OK;
}

extern "C" int cmgpu_set_barcode_check(cmgpu_ctx *c, int enabled) {
  if (!c) return CMGPU_EINVAL;
  c->skip_barcode_check = !enabled;
  return CMGPU_OK;
}

extern "C" int cmgpu_compute_barcode_abundance(cmgpu_ctx *c, const char *bases, const uint32_t *offsets, uint32_t n,
                                               uint64_t *num_sample_barcodes) {
  if (!c || !bases || !offsets || c->wl_size == 0) { cm_set_error(c, "no whitelist set"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  DevBuf db, dofs;
  const size_t nbytes = n ? offsets[n] : 0;
  if (db.ensure(nbytes + 16) || dofs.ensure(((size_t)n + 1) * 4)) { cm_set_error(c, "out of device memory (barcodes)"); return CMGPU_ENOMEM; }
  HIPCHECK(c, hipMemcpy(db.p, bases, nbytes, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(dofs.p, offsets, ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
  bool done = false;
  const int rc = bc_abundance_run(c, (const uint8_t *)db.p, (const uint32_t *)dofs.p, n, &done);
  db.release(); dofs.release();
  if (num_sample_barcodes) *num_sample_barcodes = c->wl_num_sample;
  return rc;
}

// same over the barcodes last taken from FASTQ stream 2 (cmgpu_fastq_take); call per chunk until *done
extern "C" int cmgpu_barcode_abundance_resident(cmgpu_ctx *c, uint64_t *num_sample_barcodes, int *done) {
  if (!c || !done || c->wl_size == 0) { cm_set_error(c, "no whitelist set"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  bool d = false;
  const int rc = bc_abundance_run(c, (const uint8_t *)c->st_bcb.p, (const uint32_t *)c->st_bco.p, c->fq[2].taken, &d);  // (taken, not committed: the staging buffers)
  *done = d ? 1 : 0;
  if (num_sample_barcodes) *num_sample_barcodes = c->wl_num_sample;
  return rc;
}

extern "C" int cmgpu_map_pairs_barcoded(cmgpu_ctx *c, const cmgpu_batch *in, const cmgpu_barcode_batch *bc, cmgpu_record_bc *out,
                                        uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats) {
  if (!c || !in || !bc || !n_out) return CMGPU_EINVAL;
  // no whitelist at all: every barcode is kept as read (chromap.h:897-903); with one, the abundance pre-pass must have run
  if (c->wl_size != 0 && c->wl_num_sample == 0) { cm_set_error(c, "barcode abundance not computed (cmgpu_compute_barcode_abundance)"); return CMGPU_EINVAL; }
  int rc = cmgpu_upload_batch(c, in);
  if (rc) return rc;
  const uint32_t n = in->n_pairs;
  *n_out = 0;
  if (n == 0) return CMGPU_OK;
  const size_t nbytes = bc->offsets[n];
  if (c->bcb.ensure(nbytes + 16) || c->bcq.ensure(nbytes + 16) || c->bco.ensure(((size_t)n + 1) * 4)) { cm_set_error(c, "out of device memory (barcodes)"); return CMGPU_ENOMEM; }
  HIPCHECK(c, hipMemcpy(c->bcb.p, bc->bases, nbytes, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(c->bcq.p, bc->qualities, nbytes, hipMemcpyHostToDevice));
  HIPCHECK(c, hipMemcpy(c->bco.p, bc->offsets, ((size_t)n + 1) * 4, hipMemcpyHostToDevice));
  c->has_barcodes = true;
  uint64_t k = 0;
  rc = cmgpu_map_resident(c, &k, stats);
  if (rc) return rc;
  if (!out) { *n_out = k; return CMGPU_OK; }
  const uint32_t K = cm_rec_per_pair(c);
  const size_t ns = (size_t)n * K;  // K record slots per pair, one key per pair
  std::vector<cmgpu_record> rec(ns);
  std::vector<uint8_t> ok(ns);
  std::vector<uint64_t> keys(n);
  HIPCHECK(c, hipMemcpy(rec.data(), c->rec.p, ns * 24, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(ok.data(), c->rec_ok.p, ns, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(keys.data(), c->bc_key.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  uint64_t o = 0;
  for (size_t i = 0; i < ns; ++i) {
    if (!ok[i]) continue;
    if (o >= out_capacity) { cm_set_error(c, "record buffer too small"); return CMGPU_ECAPACITY; }
    out[o].r = rec[i];
    out[o].barcode = keys[i / K];
    ++o;
  }
  *n_out = o;
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// single-end reads: the taskloop body of Chromap::MapSingleEndReads (chromap.h:385-472).
// The batch is held as pairs whose second mate is empty.
// ---------------------------------------------------------------------------------------
static int map_single_impl(cmgpu_ctx *c, const cmgpu_single_batch *in, const cmgpu_barcode_batch *bc, cmgpu_record *out,
                           cmgpu_record_bc *out_bc, uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats);
extern "C" int cmgpu_map_single(cmgpu_ctx *c, const cmgpu_single_batch *in, cmgpu_record *out, uint64_t out_capacity,
                                uint64_t *n_out, cmgpu_stats *stats) {
  return map_single_impl(c, in, nullptr, out, nullptr, out_capacity, n_out, stats);
}
// single-end reads with cell barcodes (MappingWithBarcode, bed_mapping.h:11-56): out may be NULL (records stay resident)
extern "C" int cmgpu_map_single_barcoded(cmgpu_ctx *c, const cmgpu_single_batch *in, const cmgpu_barcode_batch *bc, cmgpu_record_bc *out,
                                         uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats) {
  if (!bc) return CMGPU_EINVAL;
  return map_single_impl(c, in, bc, nullptr, out, out_capacity, n_out, stats);
}
static int map_single_impl(cmgpu_ctx *c, const cmgpu_single_batch *in, const cmgpu_barcode_batch *bc, cmgpu_record *out,
                           cmgpu_record_bc *out_bc, uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats) {
  if (!c || !in || !n_out) return CMGPU_EINVAL;
  if (bc && c->wl_size != 0 && c->wl_num_sample == 0) { cm_set_error(c, "barcode abundance not computed (cmgpu_compute_barcode_abundance)"); return CMGPU_EINVAL; }
  if (c->p.split) { cm_set_error(c, "single-end split alignment is not supported"); return CMGPU_EINVAL; }
  HIPCHECK(c, cm_enter(c));
  const uint32_t n = in->n_reads;
  c->n_pairs = n;
  c->has_barcodes = false;
  c->single = true;
  c->first_read_id = in->first_read_id;
  c->bases0 = n ? in->offsets[n] : 0;
  c->bases1 = 0;
  uint32_t mx = 1;
  for (uint32_t i = 0; i < n; ++i) { const uint32_t l = in->offsets[i + 1] - in->offsets[i]; mx = l > mx ? l : mx; }
  c->max_read_len = mx;
  *n_out = 0;
  if (n == 0) return CMGPU_OK;
  if (c->rb0.ensure(c->bases0 + 16) || c->rb1.ensure(16) || c->ro0.ensure(((size_t)n + 1) * 4) || c->ro1.ensure(((size_t)n + 1) * 4)) {
    cm_set_error(c, "out of device memory (reads)");
    return CMGPU_ENOMEM;
  }
  HIPCHECK(c, hipMemcpyAsync(c->rb0.p, in->bases, c->bases0, hipMemcpyHostToDevice, c->stream));
  HIPCHECK(c, hipMemcpyAsync(c->ro0.p, in->offsets, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHECK(c, hipMemsetAsync(c->ro1.p, 0, ((size_t)n + 1) * 4, c->stream));
  if (bc) {
    const size_t nbytes = bc->offsets[n];
    if (c->bcb.ensure(nbytes + 16) || c->bcq.ensure(nbytes + 16) || c->bco.ensure(((size_t)n + 1) * 4)) { cm_set_error(c, "out of device memory (barcodes)"); return CMGPU_ENOMEM; }
    HIPCHECK(c, hipMemcpyAsync(c->bcb.p, bc->bases, nbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->bcq.p, bc->qualities, nbytes, hipMemcpyHostToDevice, c->stream));
    HIPCHECK(c, hipMemcpyAsync(c->bco.p, bc->offsets, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, c->stream));
    c->has_barcodes = true;
  }
  HIPCHECK(c, cm_stream_sync(c->stream));
  uint64_t k = 0;
  int rc = cmgpu_map_resident(c, &k, stats);
  if (rc) return rc;
  if (!out && !out_bc) { *n_out = k; return CMGPU_OK; }
  if (!bc) return cmgpu_download_records(c, out, out_capacity, n_out);
  const uint32_t K = cm_rec_per_pair(c);
  const size_t ns = (size_t)n * K;  // K record slots per pair, one key per pair
  std::vector<cmgpu_record> rec(ns);
  std::vector<uint8_t> ok(ns);
  std::vector<uint64_t> keys(n);
  HIPCHECK(c, hipMemcpy(rec.data(), c->rec.p, ns * 24, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(ok.data(), c->rec_ok.p, ns, hipMemcpyDeviceToHost));
  HIPCHECK(c, hipMemcpy(keys.data(), c->bc_key.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  uint64_t o = 0;
  for (size_t i = 0; i < ns; ++i) {
    if (!ok[i]) continue;
    if (o >= out_capacity) { cm_set_error(c, "record buffer too small"); return CMGPU_ECAPACITY; }
    out_bc[o].r = rec[i];
    out_bc[o].barcode = keys[i / K];
    ++o;
  }
  *n_out = o;
  return CMGPU_OK;
}

// ---------------------------------------------------------------------------------------
// cmgpu_warm_up: what a process can do before it has an index to make a context with (chromap-amd calls it on a thread of its own while the
// main thread reads the reference and the index file).  The runtime initialised on the device, the library's device code loaded, and ONE DRY
// RUN of the whole path on a made-up 64 kb reference: 2 000 pairs from FASTQ text through scan / take / commit, mapping, the record store and
// the BED text.  What a process otherwise pays inside its first batch (round 6, BGZF -> BED of 8 M pairs: the first 4 M-pair batch took
// 33-64 ms to map, the second 7): the first stream of a priority makes the runtime create a hardware queue (14.5 ms in front of the first
// k_pack_reads), the first kernel of a queue that needs scratch memory has it allocated (15 ms in front of the first k_s3b_candidates), the
// first sort builds its plan.  Hardware queues, their scratch and the loaded code stay with the process when the dry run's context goes.
// Errors are ignored: a warm-up that fails leaves the process as it would have been without it.
// ---------------------------------------------------------------------------------------
static void cm_dry_run(int device_id) {
  const uint32_t L = 65536, n_pairs = 2000, rl = 50;
  std::string seq(L, 'A');
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (uint32_t i = 0; i < L; ++i) { x = x * 6364136223846793005ull + 1442695040888963407ull; seq[i] = "ACGT"[(x >> 60) & 3]; }
  const char *names[1] = {"warm"};
  const char *seqs[1] = {seq.data()};
  const uint32_t lens[1] = {L};
  cmgpu_ref_view rv;
  rv.n_sequences = 1; rv.names = names; rv.sequences = seqs; rv.lengths = lens;
  cmgpu_params p;
  cmgpu_default_params(&p);
  p.max_insert_size = 2000; p.remove_pcr_duplicates = 1; p.low_memory_mode = 1; p.tn5_shift = 1; p.mapq_threshold = 30; p.trim_adapters = 1;  // (--preset atac)
  cmgpu_ctx *ctx = nullptr;
  if (cmgpu_create_from_reference(&rv, 17, 7, &p, device_id, &ctx) != CMGPU_OK || !ctx) return;
  std::string t1, t2;
  t1.reserve(n_pairs * (2 * rl + 16)); t2.reserve(n_pairs * (2 * rl + 16));
  for (uint32_t i = 0; i < n_pairs; ++i) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    const uint32_t at = (uint32_t)((x >> 33) % (L - 400)), frag = 120 + (uint32_t)((x >> 20) & 127);
    std::string r1 = seq.substr(at, rl), r2(rl, 'A');
    for (uint32_t k = 0; k < rl; ++k) { const char ch = seq[at + frag - 1 - k]; r2[k] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : 'A'; }
    if ((i & 7) == 0) r1[rl / 2] = r1[rl / 2] == 'A' ? 'C' : 'A';  // (some with a mismatch: the alignment kernels see work)
    t1 += "@w\n"; t1 += r1; t1 += "\n+\n"; t1.append(rl, 'I'); t1 += "\n";
    t2 += "@w\n"; t2 += r2; t2 += "\n+\n"; t2.append(rl, 'I'); t2 += "\n";
  }
  uint32_t c1 = 0, c2 = 0;
  uint64_t used = 0, k = 0, nl = 0, nb = 0;
  cmgpu_stats st;
  memset(&st, 0, sizeof(st));
  const bool ok = cmgpu_fastq_scan(ctx, 0, t1.data(), t1.size(), 1, &c1) == CMGPU_OK && cmgpu_fastq_scan(ctx, 1, t2.data(), t2.size(), 1, &c2) == CMGPU_OK &&
                  c1 == n_pairs && c2 == n_pairs && cmgpu_fastq_take(ctx, 0, n_pairs, &used) == CMGPU_OK && cmgpu_fastq_take(ctx, 1, n_pairs, &used) == CMGPU_OK &&
                  cmgpu_fastq_commit(ctx, n_pairs, 0, 1, 0) == CMGPU_OK && cmgpu_map_resident(ctx, &k, &st) == CMGPU_OK &&
                  cmgpu_store_append_resident(ctx, nullptr) == CMGPU_OK &&
                  cmgpu_store_format(ctx, CMGPU_TEXT_BED_PE, names, 1, &p, 0, &nl, &nb) == CMGPU_OK;
  (void)ok;
  (void)cmgpu_destroy(ctx);
  (void)hipGetLastError();
}

extern "C" int cmgpu_warm_up(int device_id) {
  const int rc = select_device(device_id);
  if (rc != CMGPU_OK) return rc;
  cm_load_device_code(nullptr);
  static std::once_flag dry[64];
  if (device_id >= 0 && device_id < 64 && !getenv("CM_NO_DRY_RUN")) std::call_once(dry[device_id], [device_id]() { cm_dry_run(device_id); });
  (void)hipSetDevice(device_id);
  return CMGPU_OK;
}
