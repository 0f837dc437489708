// cm_ctx.h -- the context behind the C ABI: device buffers and per-batch state
#ifndef CM_CTX_H_
#define CM_CTX_H_
#include <hip/hip_runtime.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/chromap_amd.h"
#include "../../include/chromap_amd_debug.h"
#include "cm_types.h"

#define CM_MAX_EVENTS 32
#define CM_MAX_W_HOST 32

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  bool owned = true;  // false: a view of another context's buffer (cmgpu_create_shared); never freed or regrown here
  int ensure(size_t bytes);  // grows (contents are NOT preserved); 0 on success
  void release();
};

// one FASTQ text stream being parsed on the device (cm_ingest.hip)
struct CmFqStream {
  DevBuf text, cnt, off, nl, keep, pos, recidx, len, bad;
  DevBuf scan_tmp;            // scratch of this stream's prefix sums (the files' scans may run side by side, each on its own HIP stream)
  hipStream_t hs = nullptr;   // created on first use (fq_hs, cm_ingest.hip), destroyed with the context
  DevBuf red_tmp;             // scratch of the take's reduction (kept: a hipFree per take would wait for the whole device)
  DevBuf text2, comp, btab, toks, ntok;  // BGZF inflated on the device: second text buffer (the retained rest moves through it), compressed blocks,
                                         // their table, the match tokens of the first pass and their number per block
  bool dev_mode = false;      // the text lives on the device between calls (cmgpu_fastq_scan_bgzf)
  uint64_t dev_len = 0;       // bytes of text kept in front of the next blocks
  uint64_t n_bytes = 0;
  uint32_t n_nl = 0, n_raw = 0, n_rec = 0, taken = 0, taken_bases = 0, taken_max_len = 0;
  bool final_chunk = false;
  // --read-format (SequenceEffectiveRange, sequence_effective_range.h): up to 4 [start, end] ranges (end -1 = to the
  // last base) concatenated, then reverse-complemented when strand is '-'
  int n_ranges = 0;
  int rng_start[4] = {0, 0, 0, 0}, rng_end[4] = {-1, -1, -1, -1};
  bool minus = false;
};

// multi-GPU record exchange (cm_exchange.hip): this context's place in a group of `world` contexts, one per GPU
struct CmExchange {
  int transport = 0;        // 0: none, 1: RCCL (library-owned communicator on the mapping stream), 2: caller-provided callbacks
  int rank = 0, world = 0;
  void *comm = nullptr;     // ncclComm_t
  cmgpu_exchange_transport ext = {};
  std::vector<uint8_t> h_owner;   // owner rank of every rid (length-weighted, contiguous rid ranges)
  DevBuf owner, send, counts /* counts[64], cursors[64], matrix[64*64] */, stage;
  unsigned long long *h_matrix = nullptr;  // pinned, world*world
  hipStream_t stream = nullptr;            // the payload (grouped send / recv) runs here, under the next batch's kernels
  hipEvent_t ev_part = nullptr, ev_payload = nullptr;
  bool payload_pending = false;
  uint64_t sent_total = 0, recv_total = 0, steps = 0;
  uint64_t owned_by[64] = {};  // records every rank's store has received so far (column sums of the count matrices)
};

// a parked resident batch (cmgpu_swap_resident_batch): the measurement rotates several distinct batches
#define CM_BATCH_SLOTS 8
struct CmBatchSlot {
  DevBuf rb0, rb1, ro0, ro1;
  uint32_t n_pairs = 0, first_read_id = 0, max_read_len = 1;
  size_t bases0 = 0, bases1 = 0;
};

struct cmgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;     // index probe of chunk c runs here next to the minimizer pass of chunk c+1
  hipStream_t stream_pack = nullptr; // k_pack_reads under S3 / S4, highest priority (a fraction of a millisecond of work that S5 waits for)
  hipEvent_t chunk_ev[CM_MM_CHUNKS + 1] = {};  // minimizers of chunk c written / all probes done
  std::string err;
  cmgpu_params hp;
  CmParams p;
  // index + reference
  DevBuf bkt, occ, ref, ref_off, ref_len, len_coef, nsec_break;
  uint32_t bmask = 0, n_occ = 0, n_seq = 0;
  // the same table re-hashed on the device into 2^shift times as many buckets (cmgpu_set_option "probe_table_shift"): the
  // pipeline probes it instead -- same keys, values, hash and probe sequence, fewer buckets visited per lookup
  DevBuf bkt_fast, coop_prof;
  uint32_t fmask = 0;
  // the reference as bit planes (CmDev::ref_pl; built on the first mapping call, views in the lanes) and the resident batch's
  // reads as bit planes (CmDev::read_pl, per range): what k_s5b_verify aligns on.  cmgpu_set_option "verify_planes" 0: the byte form
  DevBuf ref_planes, read_planes;
  // reads longer than 69 bases: k_prep_mm's emissions staged in global memory (one tile per block of a launch); cmgpu_set_option
  // "long_read_fused" 0: the two-pass kernels with their scan and host waits (the round-2 form)
  DevBuf mm_stage;
  int opt_long_fused = 1;
  uint64_t ref_pl_words = 0;
  int opt_planes = 1;
  int n_break = 0;
  uint64_t ref_bytes = 0;
  std::vector<uint64_t> h_ref_off;
  std::vector<uint32_t> h_ref_len;
  // resident batch
  uint32_t n_pairs = 0, first_read_id = 0;
  size_t bases0 = 0, bases1 = 0;
  uint32_t max_read_len = 1;
  DevBuf rb0, rb1, ro0, ro1;
  // the batch being TAKEN from the FASTQ streams (cmgpu_fastq_take writes here, cmgpu_fastq_commit swaps these with the resident
  // batch's buffers): the next batch is parsed and gathered while the resident one is being mapped on another host thread
  DevBuf st_rb0, st_rb1, st_ro0, st_ro1, st_bcb, st_bcq, st_bco;
  // per-read / per-pair arrays (names match CmDev)
  DevBuf rlen, scratch_a, scratch_b /* 2n+1 u32 each, free between batches */, mm_cnt, mm_off, mm_hash, mm_ps, pr_val, pr_kind;
  DevBuf hit_tot, hit_off, round2, rep_cnt, rep_len, hbuf, hcnt, n_pos_hit, ncp, ncn;
  DevBuf aug, res_neg, res_pos, resc_n, resc_p, m_tot, m_off, mbuf, mcnt, mcp, mcn, force0;
  DevBuf fbuf, fcnt, fcp, fcn, alive, dpos, derr, dsplit, nv, v_off, v_err, v_end, ndp, ndn, min_err, second_err, n_best, n_second;
  DevBuf pe_min, pe_second, pe_nbest, pe_nsecond, pe_first, pe_i1, pe_i2, pe_choice, rec, rec_ok;
  DevBuf scan_tmp, stats, partials;
  // single-cell barcodes
  DevBuf wl, pow10_tab, bcb, bcq, bco, bc_key, bc_ok, wl_num;
  uint32_t wl_mask = 0, wl_size = 0, bc_len = 0;
  uint64_t wl_num_sample = 0;
  bool skip_barcode_check = false;  // --skip-barcode-check
  bool has_barcodes = false;  // resident batch carries barcodes
  bool single = false;        // resident batch is single-end
  // device-side record store + rendered text (cm_post.hip)
  DevBuf store, store_bc, text;
  uint64_t store_n = 0, store_cap = 0, text_bytes = 0, text_lines = 0;
  bool store_has_bc = false;
  CmFqStream fq[3];  // read 1, read 2, barcode
  // --SAM outputs of the last batch (cm_stages.h: cm_ref_start_end_sam)
  DevBuf sam_rec, sam_cigar, sam_md, sam_z;
  DevBuf pairs_rank;
  bool has_pairs_rank = false;
  DevBuf rid_rank, ref_off_r, ref_len_r;  // --chr-order: rank per rid, reference offset / length arrays reordered by rank
  bool has_rank = false;
  DevBuf mm_cursor;  // k_prep_mm: next free entry of the dense minimizer arrays
  DevBuf mm_marks;   // cursor after every chunk of pairs: the range of minimizers a chunk's probe launch covers
  DevBuf part_cnt;  // cmgpu_records_partition: per-owner counts and cursors
  unsigned long long cls_seen = ~0ull;  // long-list classes whose kernels the next range launches: those that had items in one of the last 32 ranges (all: nothing known yet)
  uint8_t cls_age[64] = {};             // ranges a class stays in the set without items (a class with a handful of items per batch comes and goes)
  bool cls_all = false;                 // this range is a re-run with every class on
  DevBuf rs_pool, rs_pool_off;  // CmDev::rs_pool
  DevBuf goff;                  // CmDev::goff (built on the first mapping call; stays empty when the reference does not fit 32 bits)
  bool goff_tried = false;
  uint32_t rs_pool_cap = 0;
  uint64_t rs_pool_want = 0;   // entries the previous range asked of the pool
  DevBuf coop_slab, hv_cnt, hv_list, perm_reads, perm_pairs, hv_tmp, srt_cnt, srt_list, rs_list, rs_cnt;
  bool use_perm = false;  // the current range has enough heavy reads to process them apart (cm_build_heavy_last)  // reads with long hit lists by size class (k_s3a_count -> k_s3b_heavy)
  CmExchange ex;
  bool batch_exchanged = false;  // the resident batch's records have been through cmgpu_exchange_step
  CmBatchSlot slots[CM_BATCH_SLOTS];
  // pipelined host-buffer entry (cmgpu_submit_pairs / cmgpu_map_submitted): the next batch is uploaded on a copy stream into the
  // last parking slot while the current one is mapped
  hipStream_t stream_h2d = nullptr;
  int opt_d2h_kernel = 0;
  int opt_h2d_kernel = 0;   // blocks of k_host_copy for uploads from page-locked memory; 0 = hipMemcpyAsync (measured faster: the copy
                            // kernel's waves slow the mapping kernels more than the copy engine's lower rate costs)
  hipEvent_t ev_h2d[2] = {nullptr, nullptr};
  uint32_t sub_total = 0, sub_count = 0;  // batches submitted so far / submitted and not yet mapped (<= 2: parking slots 6 and 7 take turns)
  DevBuf rec_dense, maxlen_dev;
  // cmgpu_map_submitted_async / cmgpu_records_wait: the compacted records of a batch go to the host on a copy stream of their
  // own while the next batch is mapped; two downloads may be pending (their device buffers take turns)
  hipStream_t stream_d2h = nullptr;
  hipEvent_t ev_comp = nullptr, ev_d2h[2] = {nullptr, nullptr};
  DevBuf rec_dense_b;
  uint64_t pend_k[2] = {0, 0};
  uint32_t pend_total = 0, pend_count = 0;
  uint32_t *h_maxlen = nullptr;  // pinned: longest read of the submitted batch, computed on the device
  // cmgpu_set_option
  int opt_probe_variant = 1;       // lookups per lane | 16: second probe step requested with the first (measured: more requests in
                                   // flight per lane only slow the probe down -- the table's random-access rate is the bound)
  int opt_mm_chunks = CM_MM_CHUNKS;
  int opt_prep_tile_reads = 32;    // reads per tile of the position-parallel minimizer kernel
  int opt_prep_kernel = 0;         // 0: lane-per-read minimizer kernels; 1: the position-parallel kernel where it applies (k_prep_flat:
                                   // half the instructions, but 7 barriers + one global reservation per tile -- measured slower)
  uint64_t opt_item_limit = 0xfffffff0ull;
  int opt_exchange_overlap = 0;      // exchange payload on a stream of its own (under the next batch's kernels)
  int opt_s3b_cap = 0;               // 0: by read length (cm_s3b_lane_cap)
  int opt_lanes = 1;                 // sub-batches of one cmgpu_map_* call mapped side by side (own streams and intermediates each)
  std::vector<cmgpu_ctx *> lanes;    // the further lanes' contexts (views of this context's index, reference, batch and record arrays)
  int shared_children = 0;           // contexts made by cmgpu_create_shared that view this one's buffers (the lanes among them refresh their views)
  cmgpu_ctx *shared_parent = nullptr;
  int opt_heavy_mid = 0;             // 0: 64 hits (96 for reads of 100 bases and more); -1: no 16-lane class
  int opt_heavy_max[3] = {0, 0, 0};  // size classes of the cooperative hit-list kernel (0: the kernel's own)
  int opt_heavy_last = 0;            // heavy-last processing order: 0 auto, 1 always, -1 never
  // candidate arrays sized from the previous batch of the same size (+ 25 %): no host wait for their total; the device checks it
  bool pred_m_ok = false;
  uint32_t pred_n = 0;
  uint64_t m_cap = 0;
  int opt_spec = 1;                  // 0: every batch waits for its totals
  int opt_coop_rb = 0;               // tests: run-table size of the cooperative sorters (0: their own)
  int opt_coop = 0xff;               // cooperative (group per item, cm_coop.h) forms of the stages for long lists: bit 0 S3b hit lists,
                                     // 1 S4b rescue hits, 2 S4c pair filter, 3 S5c acceptance, 4 S6 pairing; 0: the round-2 kernels
  std::vector<uint32_t> h_rank;  // --chr-order: rank of every index rid (host copy of rid_rank)
  uint64_t sam_slots = 0;
  uint32_t sam_md_cap = 0;
  // one batch in flight (cmgpu_map_pairs_async / cmgpu_wait)
  std::thread worker;
  bool in_flight = false;
  int async_rc = 0;
  uint64_t async_n = 0;
  uint64_t n_records = 0;
  uint64_t last_n_mm = 0, last_n_hits = 0, last_n_cand_cap = 0;
  uint32_t last_range_lo = 0, last_range_hi = 0;  // pair range whose intermediates are resident (cmgpu_debug_*)
  uint64_t synth_n_minimizers = 0, synth_n_keys = 0;
  // timing
  hipEvent_t ev[CM_MAX_EVENTS] = {};
  const char *ev_name[CM_MAX_EVENTS] = {};
  int n_ev = 0;

  std::vector<DevBuf *> all_bufs() {
    std::vector<DevBuf *> v = core_bufs();
    for (CmFqStream &f : fq) for (DevBuf *b : {&f.text, &f.cnt, &f.off, &f.nl, &f.keep, &f.pos, &f.recidx, &f.len, &f.bad, &f.text2, &f.comp, &f.btab, &f.toks, &f.ntok, &f.scan_tmp, &f.red_tmp}) v.push_back(b);
    for (DevBuf *b : {&st_rb0, &st_rb1, &st_ro0, &st_ro1, &st_bcb, &st_bcq, &st_bco}) v.push_back(b);
    for (CmBatchSlot &sl : slots) for (DevBuf *b : {&sl.rb0, &sl.rb1, &sl.ro0, &sl.ro1}) v.push_back(b);
    return v;
  }
  std::vector<DevBuf *> core_bufs() {
    return {&bkt, &occ, &ref, &ref_off, &ref_len, &len_coef, &nsec_break, &rb0, &rb1, &ro0, &ro1, &rlen, &scratch_a,
            &scratch_b, &mm_cnt, &mm_off, &mm_hash, &mm_ps, &pr_val, &pr_kind, &hit_tot,
            &hit_off, &round2, &rep_cnt, &rep_len, &hbuf, &hcnt, &n_pos_hit, &ncp, &ncn, &aug, &res_neg, &res_pos,
            &resc_n, &resc_p, &m_tot, &m_off, &mbuf, &mcnt, &mcp, &mcn, &force0, &fbuf, &fcnt, &fcp, &fcn, &alive,
            &dpos, &derr, &dsplit, &nv, &v_off, &v_err, &v_end, &ndp, &ndn, &min_err, &second_err, &n_best, &n_second, &pe_min, &pe_second, &pe_nbest,
            &pe_nsecond, &pe_first, &pe_i1, &pe_i2, &pe_choice, &rec, &rec_ok, &scan_tmp, &stats, &partials, &wl, &pow10_tab, &bcb, &bcq, &bco, &bc_key, &bc_ok, &wl_num, &store, &store_bc, &text, &sam_rec, &sam_cigar, &sam_md, &sam_z, &part_cnt, &mm_cursor, &mm_marks, &rid_rank, &ref_off_r, &ref_len_r, &pairs_rank,
            &ex.owner, &ex.send, &ex.counts, &ex.stage, &rec_dense, &rec_dense_b, &maxlen_dev, &bkt_fast, &ref_planes, &read_planes, &mm_stage, &coop_prof, &rs_pool, &rs_pool_off, &goff, &coop_slab, &hv_cnt, &hv_list, &perm_reads, &perm_pairs, &hv_tmp, &srt_cnt, &srt_list, &rs_list, &rs_cnt};
  }
};

void cm_set_error(cmgpu_ctx *ctx, const std::string &msg);

// Kernel launches report a bad configuration (LDS size, grid) only through hipGetLastError; every
// synchronisation point of the library therefore checks it too, and every entry point starts from a
// clean slate so that only this call's own launches are seen.
static inline hipError_t cm_enter(const cmgpu_ctx *c) {
  (void)hipGetLastError();
  return hipSetDevice(c->device);
}
static inline hipError_t cm_stream_sync(hipStream_t s) {
  const hipError_t e = hipStreamSynchronize(s);
  const hipError_t l = hipGetLastError();
  return e != hipSuccess ? e : l;
}
int cm_ctx_init_common(cmgpu_ctx *c, const cmgpu_params *params, int kmer, int window, int device_id);
void cm_fill_dev(cmgpu_ctx *c, CmDev &d);
int cm_upload_reference(cmgpu_ctx *c, const cmgpu_ref_view *ref);
int cm_build_ref_planes(cmgpu_ctx *c);
// the probe table re-hashed into 2^shift times the buckets (0: released)
int cm_build_fast_table(cmgpu_ctx *c, int shift);
// cm_post.hip: room for `need` records in the device-side store (with_bc: the parallel barcode array too)
int cm_store_reserve(cmgpu_ctx *c, uint64_t need, bool with_bc);
// record slots of the resident batch: max_num_best_mappings per pair (cm_emit_record); flag / position scratch for a
// compaction over them (scratch_a, scratch_b, scan_tmp sized for `slots` entries)
static inline uint32_t cm_rec_per_pair(const cmgpu_ctx *c) { return (uint32_t)(c->p.max_best > 0 ? c->p.max_best : 1); }
static inline bool cm_pairs_records(const cmgpu_ctx *c) { return c->p.split || c->p.pairs_out; }  // the records are cmgpu_pairs_record entries
static inline uint64_t cm_rec_slots(const cmgpu_ctx *c) { return (uint64_t)c->n_pairs * cm_rec_per_pair(c); }
int cm_ensure_slot_scratch(cmgpu_ctx *c, uint64_t slots);
// cm_post.hip: n 32-byte {record, barcode} entries -> the store's record / barcode arrays at position store_n
void cm_store_split_bc(cmgpu_ctx *c, const void *in32, uint64_t n, hipStream_t s);
// cm_exchange.hip
void cm_exchange_release(cmgpu_ctx *c);
// waits for an exchange payload still in flight (it lands in the record store): before anything reads or moves the store
int cm_exchange_quiesce(cmgpu_ctx *c);
// owner rank of every sequence for `world` ranks: contiguous rid ranges of (nearly) equal total length
std::vector<uint8_t> cm_owner_table(const cmgpu_ctx *c, uint32_t world);

static inline uint32_t cm_num_chunks_host(uint32_t n, uint32_t ref_batch, uint32_t grain) {
  uint32_t tot = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += ref_batch) {
    const uint32_t bn = n - b0 < ref_batch ? n - b0 : ref_batch;
    const uint32_t T = bn / grain;
    tot += T <= 1 ? 1 : T;
  }
  return tot;
}
#endif
