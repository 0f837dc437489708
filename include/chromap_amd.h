/*
 * chromap_amd.h -- C ABI of the MI355X (gfx950) implementation of Chromap's per-read
 * mapping hot path.
 *
 * The reference (haowenz/chromap v0.3.3, /root/reference) has no plugin or FFI seam; the
 * replacement point is the body of the per-batch OpenMP taskloop in
 * Chromap::MapPairedEndReads (src/chromap.h:892-1143) together with the one-time loads in
 * front of it (src/chromap.h:641-670).  Each entry point below names the reference
 * code it stands in for.  Plain pointers and sizes only; no C++ or torch types.
 *
 * All functions return 0 on success and a negative CMGPU_E* code on failure; the text of
 * the last failure is available from cmgpu_last_error().  The reference's convention is
 * to abort the process (ExitWithMessage -> exit(-1), src/utils.h:71-74); the host driver
 * turns a non-zero status into exactly that.  There is no CPU fallback: without a HIP
 * device cmgpu_create fails with CMGPU_ENODEVICE.
 */
#ifndef CHROMAP_AMD_H_
#define CHROMAP_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMGPU_OK 0
#define CMGPU_EINVAL (-1)
#define CMGPU_ENODEVICE (-2)
#define CMGPU_EHIP (-3)
#define CMGPU_ENOMEM (-4)
#define CMGPU_ECAPACITY (-5)
#define CMGPU_EIO (-6)
#define CMGPU_EFORMAT (-7)

#define CMGPU_FORMAT_SAM 1
#define CMGPU_FORMAT_PAIRS 2 /* --pairs WITHOUT split alignment: cmgpu_map_pairs leaves cmgpu_pairs_record entries (MapPairedEndReads<PairsMapping>, src/chromap_driver.cc:748-751) */

/* The minimizer index exactly as Index::Load leaves it in host memory
 * (src/index.cc:132-169, kh_load src/khash.h:358-373): khash open-addressing arrays with
 * 2-bit flags, keys = minimizer_hash<<1 | is_singleton, values, and the occurrence
 * table.  The library re-packs it for HBM; the caller keeps ownership. */
typedef struct cmgpu_index_view {
  int32_t kmer_size;   /* Index::kmer_size_, read from the index file */
  int32_t window_size; /* Index::window_size_ */
  uint32_t n_buckets;  /* kh_n_buckets, a power of two */
  const uint32_t *flags; /* n_buckets/16 words (at least 1) */
  const uint64_t *keys;  /* n_buckets */
  const uint64_t *vals;  /* n_buckets */
  uint32_t n_occurrences;
  const uint64_t *occurrences;
} cmgpu_index_view;

/* The reference sequences as SequenceBatch::LoadAllSequences holds them
 * (src/sequence_batch.cc:84-120): raw bytes, case and N preserved, file order. */
typedef struct cmgpu_ref_view {
  uint32_t n_sequences;
  const char *const *names;     /* used only by cmgpu_write_bed_pe */
  const char *const *sequences; /* n_sequences pointers, lengths[] bytes each */
  const uint32_t *lengths;
} cmgpu_ref_view;

/* The fields of MappingParameters the path reads (src/mapping_parameters.h:18-89). */
typedef struct cmgpu_params {
  int32_t error_threshold;        /* -e */
  int32_t min_num_seeds;          /* -s */
  int32_t max_seed_frequency0;    /* -f first value */
  int32_t max_seed_frequency1;    /* -f second value */
  int32_t max_insert_size;        /* -l */
  int32_t min_read_length;        /* --min-read-length */
  int32_t max_num_best_mappings;  /* -n: up to this many records per read / pair (1..8192), reservoir-sampled when there are more
                                   * best mappings (mapping_generator.h:121-139,199-214); not with CMGPU_FORMAT_SAM */
  int32_t drop_repetitive_reads;  /* --drop-repetitive-reads */
  int32_t trim_adapters;          /* --trim-adapters */
  int32_t split_alignment;        /* --split-alignment (records are then cmgpu_pairs_record) */
  int32_t mapq_threshold;         /* -q; used by cmgpu_write_bed_pe only */
  int32_t remove_pcr_duplicates;  /* used by the writers / cmgpu_store_format* only */
  int32_t tn5_shift;              /* used by cmgpu_write_bed_pe only */
  int32_t low_memory_mode;        /* used by the writers / cmgpu_store_format* only */
  int32_t read_batch_size;        /* Chromap::read_batch_size_ = 500000 (chromap.h:182) */
  int32_t taskloop_grain_size;    /* 5000 (chromap.h:887): scope of the reservoir RNG */
  int32_t bc_error_threshold;     /* --bc-error-threshold (0, 1 or 2 on the device) */
  int32_t output_mappings_not_in_whitelist; /* --output-mappings-not-in-whitelist */
  int32_t output_format;          /* 0: BED records (pairs records with split_alignment); CMGPU_FORMAT_SAM: --SAM (alignment coordinates, CIGAR,
                                   * NM, MD); CMGPU_FORMAT_PAIRS: pairs records from the ordinary (non-split) pairing */
  int32_t dedup_at_bulk_level;    /* single-cell BED, low-memory flavour: --remove-pcr-duplicates-at-bulk-level (the reference's
                                   * default without --preset atac); applied by cmgpu_store_format only */
  double bc_probability_threshold; /* --bc-probability-threshold */
} cmgpu_params;

/* A batch of read pairs as SequenceBatch holds them after parsing (src/chromap.cc:93-174):
 * concatenated ASCII bases (any case, N allowed) with n+1 offsets per mate.  read_id of
 * pair i is first_read_id + i (src/sequence_batch.cc:38-39). */
typedef struct cmgpu_batch {
  uint32_t n_pairs;
  uint32_t first_read_id;
  const char *read1_bases;
  const uint32_t *read1_offsets; /* n_pairs + 1 */
  const char *read2_bases;
  const uint32_t *read2_offsets; /* n_pairs + 1 */
} cmgpu_batch;

/* Constructor arguments of PairedEndMappingWithoutBarcode (src/bed_mapping.h:191-206) plus
 * rid, which the reference keeps implicitly as the index of the per-chromosome vector
 * (src/mapping_generator.cc:116). 24 bytes. */
typedef struct cmgpu_record {
  uint32_t read_id;
  uint32_t rid;
  uint32_t fragment_start;
  uint16_t fragment_length;
  uint8_t mapq;      /* 6-bit field in the reference */
  uint8_t direction; /* 1: read1 maps to the + strand */
  uint8_t is_unique;
  uint8_t num_dups;  /* 1 */
  uint16_t positive_alignment_length;
  uint16_t negative_alignment_length;
} cmgpu_record;

/* With params.split_alignment (--preset hic) the record is the constructor argument list of
 * PairsMapping (src/pairs_mapping.h:25-38) without read name and barcode, already flipped so
 * that (rid1,pos1) <= (rid2,pos2) (src/mapping_generator.cc:169-210); pos = ref start for a +
 * read, ref end for a - read, 0-based.  Same 24-byte slot as cmgpu_record: the `out` buffers of
 * cmgpu_map_pairs / cmgpu_download_records then hold cmgpu_pairs_record entries. */
typedef struct cmgpu_pairs_record {
  uint32_t read_id;
  uint32_t rid1, rid2;
  uint32_t pos1, pos2;
  uint8_t strand1, strand2; /* 1 = positive */
  uint8_t mapq;
  uint8_t is_unique;
} cmgpu_pairs_record;

/* Single-cell data: PairedEndMappingWithBarcode (src/bed_mapping.h:128-144) = the bulk record
 * plus the 2-bit-packed (corrected) cell barcode (GenerateSeedFromSequence, src/utils.h:111-129). */
typedef struct cmgpu_record_bc {
  cmgpu_record r;
  uint64_t barcode;
} cmgpu_record_bc;

/* Cell barcodes of a batch as the barcode SequenceBatch holds them: bases + qualities
 * (same offsets), n_pairs+1 offsets. */
typedef struct cmgpu_barcode_batch {
  const char *bases;
  const char *qualities;
  const uint32_t *offsets;
} cmgpu_barcode_batch;

/* Counters of Chromap::OutputMappingStatistics (src/chromap.cc:808-823) plus the
 * quantities SURVEY.md 8(d) defines the index-probe kernel's algorithmic bytes from. */
typedef struct cmgpu_stats {
  uint64_t num_candidates;
  uint64_t num_mappings;
  uint64_t num_mapped_reads;
  uint64_t num_uniquely_mapped_reads;
  uint64_t num_minimizers;      /* lookups issued by the probe kernel */
  uint64_t probe_steps;         /* hash buckets visited by the probe kernel (16 B each) */
  uint64_t occurrences_read;    /* occurrence-table entries expanded (8 B each) */
  uint64_t num_pairs_rescued;   /* reads that went through mate rescue */
  uint64_t num_multi_mappers;   /* pairs resolved by reservoir sampling */
  uint64_t num_barcode_in_whitelist; /* Chromap::OutputBarcodeStatistics (chromap.cc:801-806) */
  uint64_t num_corrected_barcode;
  uint64_t reserved[5];
} cmgpu_stats;

typedef struct cmgpu_ctx cmgpu_ctx;

/* Fills *p with the reference's defaults (src/mapping_parameters.h:19-61). */
void cmgpu_default_params(cmgpu_params *p);
/* Applies a CLI preset ("atac", "chip", "hic") the way chromap_driver.cc:247-275 does. */
int cmgpu_apply_preset(cmgpu_params *p, const char *preset);

/* Replaces: reference.LoadAllSequences() + index.Load() + construction of the stateless
 * worker objects (src/chromap.h:641-670, 763-778).  Uploads index and reference to the
 * HBM of device_id once. */
int cmgpu_create(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                 int device_id, cmgpu_ctx **out);

/* Replaces: Index::Construct (src/index.cc:12-89) -- builds the minimizer index of `ref` on
 * the device (chunked minimizer pass, radix sort by (hash, hit), khash-sized open-addressing
 * table) and keeps it resident together with the reference; the ctx maps like one made by
 * cmgpu_create.  cmgpu_save_index_file replaces Index::Save (src/index.cc:91-121): the file
 * loads in the reference (kh_load) and answers every lookup identically. */
int cmgpu_create_from_reference(const cmgpu_ref_view *ref, int32_t kmer_size, int32_t window_size,
                                const cmgpu_params *params, int device_id, cmgpu_ctx **out);
int cmgpu_save_index_file(cmgpu_ctx *ctx, const char *path);

/* --chr-order (Chromap::GenerateCustomRidRanks + SequenceBatch::ReorderSequences + RerankCandidatesRid,
 * src/chromap.cc:867-923, src/chromap.h:654-659, 1060-1074): rank[i] = place of reference sequence i in the
 * output order.  Records then carry ranks in their rid fields; pass names / lengths in rank order to the writers. */
int cmgpu_set_chr_order(cmgpu_ctx *ctx, const uint32_t *rank, uint32_t n_sequences);
/* --pairs-natural-chr-order (src/chromap.h:660-664, src/mapping_generator.cc:193-203): rank, over the sequences as the
 * records number them, that decides which end of a pair is written first; cmgpu_write_pairs_ranked orders the header by it. */
int cmgpu_set_pairs_chr_order(cmgpu_ctx *ctx, const uint32_t *rank, uint32_t n_sequences);
/* Before there is an index to make a context with: initialises the HIP runtime on the device, loads the library's device
 * code and maps 2 000 made-up pairs against a made-up 64 kb reference, from FASTQ text to BED text, in a context of its own
 * that is destroyed again -- hardware queues, scratch memory and sort plans a process otherwise pays for inside its first
 * batch (tens of milliseconds).  Optional; once per device and process; thread-safe; errors of the dry run are ignored. */
int cmgpu_warm_up(int device_id);

/* A further context over the SAME resident index and reference (no copy; the parent must outlive it).
 * Contexts are single-caller, so this is how one GPU keeps several batches in flight: one host
 * thread and one context per batch; their kernels share the compute units. */
int cmgpu_create_shared(const cmgpu_ctx *parent, cmgpu_ctx **out);
int cmgpu_destroy(cmgpu_ctx *ctx);
/* ctx may be NULL (returns the last error of a failed cmgpu_create*). */
const char *cmgpu_last_error(const cmgpu_ctx *ctx);
/* the message of the last failed call made BY THE CALLING THREAD (the per-file scans of a batch -- cmgpu_fastq_scan / _scan_bgzf -- may run
 * side by side on one ctx; cmgpu_last_error(ctx) then holds whichever failed last) */
const char *cmgpu_last_error_thread(void);

/* Replaces the taskloop body src/chromap.h:892-1143 for one batch: adapter trimming,
 * minimizers, index probe, candidate generation, mate rescue, pair filter, verification,
 * best-pair selection, coordinates and MAPQ.  `in` holds HOST pointers; records are
 * written to the caller's host buffer `out` (capacity in records; n_pairs *
 * max_num_best_mappings always suffices; a pair's records are adjacent, best-mapping index increasing).  Record order is unspecified (a total-order sort follows in the reference:
 * src/mapping_processor.h:117-159).  stats may be NULL; counters are ACCUMULATED.
 * A batch must start on a read_batch_size boundary of the input file for the reservoir
 * sampling of multi-mappers to reproduce the reference. */
/* out may be NULL (capacity 0): the records then stay resident for cmgpu_store_append_resident /
 * cmgpu_records_to_device and only their count is returned (same for the other cmgpu_map_* calls). */
int cmgpu_map_pairs(cmgpu_ctx *ctx, const cmgpu_batch *in, cmgpu_record *out, uint64_t out_capacity,
                    uint64_t *n_out, cmgpu_stats *stats);

/* One batch in flight: cmgpu_map_pairs_async returns at once and a worker thread of the library
 * uploads and maps the batch while the caller parses the next one (the reference overlaps its
 * "load next batch" task with the mapping taskloop the same way, src/chromap.h:871-877).
 * cmgpu_wait joins it; out may be NULL to leave the records resident.  Between the two calls the
 * ctx must not be used and the batch's buffers (and *stats) must stay valid. */
int cmgpu_map_pairs_async(cmgpu_ctx *ctx, const cmgpu_batch *in, cmgpu_stats *stats);
int cmgpu_wait(cmgpu_ctx *ctx, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out);

/* ---- the host-buffer boundary at the link's rate --------------------------------------------------------
 * Page-locked host memory for batch buffers and record arrays (cmgpu_host_alloc, or cmgpu_host_register on the caller's
 * own allocation), and a pipelined entry: cmgpu_submit_pairs starts the upload of the NEXT batch on a copy stream and
 * returns; cmgpu_map_submitted waits for that upload, maps the batch and downloads its records compacted (pair order
 * kept) -- so the upload of batch c+1 runs under the kernels of batch c, like the reference's load-next-batch task
 * beside its mapping taskloop (src/chromap.h:871-877):
 *     cmgpu_submit_pairs(ctx, &b[0]);
 *     for (c = 0; c < n; ++c) { if (c + 1 < n) cmgpu_submit_pairs(ctx, &b[c + 1]); cmgpu_map_submitted(ctx, out[c], cap, &k, &st); }
 * Up to two batches may be submitted and not yet mapped.  A submitted batch's buffers must stay valid until its
 * cmgpu_map_submitted returns; parking slots 6 and 7 of cmgpu_swap_resident_batch serve the uploads. */
void *cmgpu_host_alloc(uint64_t bytes);
void cmgpu_host_free(void *p);
int cmgpu_host_register(void *p, uint64_t bytes);
int cmgpu_host_unregister(void *p);
int cmgpu_submit_pairs(cmgpu_ctx *ctx, const cmgpu_batch *in);
int cmgpu_map_submitted(cmgpu_ctx *ctx, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats);
/* The same with the record download left running: cmgpu_map_submitted_async maps the oldest submitted batch, compacts its
 * records and queues their copy to `out` (page-locked: cmgpu_host_alloc / cmgpu_host_register) on a copy stream of its own, so
 * the copy runs under the NEXT batch's kernels -- the reference hands a finished batch's mappings to its output task the same
 * way while the next taskloop runs (src/chromap.h:871-877, mapping_writer.h:166-376).  cmgpu_records_wait returns when the
 * oldest pending download is complete and gives its record count; `out` must not be read before.  Up to two downloads may be
 * pending:
 *     cmgpu_submit_pairs(ctx, &b[0]);
 *     for (c = 0; c < n; ++c) {
 *       if (c + 1 < n) cmgpu_submit_pairs(ctx, &b[c + 1]);
 *       cmgpu_map_submitted_async(ctx, out[c & 1], cap, &st);
 *       if (c > 0) { cmgpu_records_wait(ctx, &k); consume(out[(c - 1) & 1], k); }
 *     }
 *     cmgpu_records_wait(ctx, &k); consume(out[(n - 1) & 1], k); */
int cmgpu_map_submitted_async(cmgpu_ctx *ctx, cmgpu_record *out, uint64_t out_capacity, cmgpu_stats *stats);
int cmgpu_records_wait(cmgpu_ctx *ctx, uint64_t *n_out);

/* Single-end reads: replaces the taskloop body of Chromap::MapSingleEndReads
 * (src/chromap.h:385-472) for bulk data; records are MappingWithoutBarcode's constructor
 * arguments (src/bed_mapping.h:67-83) stored in the cmgpu_record layout: alignment-length
 * fields 0, fragment = the read's alignment, direction 1 = + strand. */
typedef struct cmgpu_single_batch {
  uint32_t n_reads;
  uint32_t first_read_id;
  const char *bases;
  const uint32_t *offsets; /* n_reads + 1 */
} cmgpu_single_batch;
int cmgpu_map_single(cmgpu_ctx *ctx, const cmgpu_single_batch *in, cmgpu_record *out, uint64_t out_capacity,
                     uint64_t *n_out, cmgpu_stats *stats);
/* BED for single-end bulk records: sort (src/bed_mapping.h:90-95), duplicate removal on the
 * start position (:96-99), MAPQ filter, Tn5 shift (:104-110), line src/mapping_writer.cc:44-52 */
int64_t cmgpu_write_bed_se(const char *const *names, uint32_t n_sequences, const cmgpu_params *params,
                           cmgpu_record *records, uint64_t n_records, const char *out_path);

/* ---- single-cell barcodes (K6) --------------------------------------------------------
 * cmgpu_set_whitelist: Chromap::LoadBarcodeWhitelist (src/chromap.cc:388-490): keys =
 * GenerateSeedFromSequence of every whitelist line (cmgpu_load_whitelist_file produces them).
 * cmgpu_compute_barcode_abundance: Chromap::ComputeBarcodeAbundance (src/chromap.cc:492-548)
 * over ALL barcodes of the input (host pointers), batch structure and stop rule included;
 * fails with CMGPU_EINVAL when fewer than 5% of the first batch are whitelisted (the reference
 * exits there).  cmgpu_map_pairs_barcoded: the taskloop body with CorrectBarcodeAt in front
 * (src/chromap.h:896-909, src/chromap.cc:572-799). */
int cmgpu_load_whitelist_file(const char *path, uint32_t barcode_length, uint64_t **keys_out, uint32_t *n_out);
int cmgpu_set_whitelist(cmgpu_ctx *ctx, const uint64_t *keys, uint32_t n_keys, uint32_t barcode_length);
/* the whitelist with the abundances of the pre-pass, copied to another context (one context per GPU: the pre-pass
 * runs on one of them) */
int cmgpu_copy_whitelist(cmgpu_ctx *dst, cmgpu_ctx *src);
/* single-end reads with cell barcodes: MappingWithBarcode (src/bed_mapping.h:11-56), src/chromap.h:385-472 */
int cmgpu_map_single_barcoded(cmgpu_ctx *ctx, const cmgpu_single_batch *in, const cmgpu_barcode_batch *barcodes,
                              cmgpu_record_bc *out, uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats);
/* enabled = 0: --skip-barcode-check (src/chromap.cc:523), the 5% test of the abundance pre-pass is not applied */
int cmgpu_set_barcode_check(cmgpu_ctx *ctx, int enabled);
int cmgpu_compute_barcode_abundance(cmgpu_ctx *ctx, const char *barcode_bases, const uint32_t *barcode_offsets,
                                    uint32_t n_barcodes, uint64_t *num_sample_barcodes);
int cmgpu_map_pairs_barcoded(cmgpu_ctx *ctx, const cmgpu_batch *in, const cmgpu_barcode_batch *barcodes,
                             cmgpu_record_bc *out, uint64_t out_capacity, uint64_t *n_out, cmgpu_stats *stats);
/* BED with the barcode column and cell-level duplicate removal (src/mapping_writer.cc:119-131,
 * src/bed_mapping.h:145-159; remove_pcr_duplicates_at_bulk_level == false as --preset atac sets). */
int64_t cmgpu_write_bed_pe_bc(const char *const *names, uint32_t n_sequences, const cmgpu_params *params,
                              cmgpu_record_bc *records, uint64_t n_records, uint32_t barcode_length,
                              const char *out_path);

/* Device-resident variant used for throughput measurement: inputs are uploaded once,
 * mapping runs on HBM-resident data and leaves the records in HBM. */
int cmgpu_upload_batch(cmgpu_ctx *ctx, const cmgpu_batch *in);
int cmgpu_map_resident(cmgpu_ctx *ctx, uint64_t *n_out, cmgpu_stats *stats);
int cmgpu_download_records(cmgpu_ctx *ctx, cmgpu_record *out, uint64_t out_capacity, uint64_t *n_out);
/* Size / content of the resident index (two-step: query sizes, then export).  buckets_out
 * receives 2*n_buckets uint64 ({key,val} per bucket, key == all ones marks an empty bucket),
 * occurrences_out n_occurrences uint64.  Used to give the CPU baseline the same index. */
int cmgpu_index_info(cmgpu_ctx *ctx, int32_t *kmer_size, int32_t *window_size, uint32_t *n_buckets,
                     uint32_t *n_occurrences, uint64_t *n_minimizers, uint64_t *n_keys);
int cmgpu_export_index(cmgpu_ctx *ctx, uint64_t *buckets_out, uint64_t *occurrences_out);

/* Dense copy of the resident batch's records into a caller-provided DEVICE buffer
 * (capacity in records) -- the send buffer of the multi-GPU record exchange. */
int cmgpu_records_to_device(cmgpu_ctx *ctx, void *device_dst, uint64_t capacity, uint64_t *n_out);
/* Same, grouped by the rank that owns the record's chromosome (cmgpu_exchange_owner_table):
 * counts[r] records for rank r, in rank order -- the send buffer and split sizes of an all-to-all. */
int cmgpu_records_partition(cmgpu_ctx *ctx, uint32_t world, void *device_dst, uint64_t capacity, uint64_t *counts);

/* ---- multi-GPU record exchange ----------------------------------------------
 * Read batches shard across GPUs with no collective on the mapping path; what the reference does
 * once per run on one host -- sort every chromosome's records and drop PCR duplicates
 * (MappingProcessor / MappingWriter::OutputTempMappings..., src/mapping_processor.h:100-202,
 * src/mapping_writer.h:166-376, fed by src/chromap.h:1305-1355) -- is done per chromosome OWNER:
 * after every batch each context sends its records to the rank that owns their chromosome (one
 * all-to-all, every record crosses xGMI once) and appends what it receives to its device-side record
 * store; cmgpu_store_format on every rank then yields that rank's section of the output, sections
 * concatenate in rank order (the output is chromosome-major).
 *   owner table   contiguous rid ranges (in the --chr-order rank space when one is set) such that the largest
 *                 owner's total sequence length is minimal: greedy in-order packing into bins of the smallest
 *                 size for which `world` bins suffice
 *   transport     RCCL, called by the library on the context's mapping stream (communicator made from a
 *                 ncclUniqueId the host distributes, or by cmgpu_exchange_init_all for the contexts of one
 *                 process), or two callbacks of the host's own communicator (MPI, a test harness).
 * cmgpu_exchange_step is collective: every rank calls it once per round, with an empty resident batch
 * when it has nothing to contribute. */
#define CMGPU_UNIQUE_ID_BYTES 128
typedef struct cmgpu_exchange_transport {
  void *user;
  /* mine[n] -> matrix[world * n], row r = rank r's `mine`.  Called twice per step: with n = world + 1 (records this
   * rank sends to each rank, then its record size) and, when world > 1, with n = 1 (a status word: a rank that cannot
   * take its share makes every rank leave the step before any payload moves) */
  int (*allgather_counts)(void *user, const uint64_t *mine, uint64_t *matrix, uint32_t n);
  /* send_dev: records grouped by destination rank (send_counts[world]); recv_dev: room for sum(recv_counts)
   * records, to be filled grouped by source rank; both are DEVICE pointers, record_bytes = 24 (32 with barcodes) */
  int (*alltoallv)(void *user, const void *send_dev, const uint64_t *send_counts, void *recv_dev,
                   const uint64_t *recv_counts, uint32_t world, uint32_t record_bytes);
} cmgpu_exchange_transport;
/* ncclGetUniqueId (call on one rank, hand the bytes to the others) */
int cmgpu_exchange_unique_id(void *id_out);
/* ncclCommInitRank on the context's device; collective over the `world` contexts */
int cmgpu_exchange_init(cmgpu_ctx *ctx, const void *unique_id, int rank, int world);
/* the contexts of ONE process, ctxs[i] becomes rank i (ncclCommInitAll over their devices) */
int cmgpu_exchange_init_all(cmgpu_ctx *const *ctxs, int n);
/* caller-provided transport instead of RCCL */
int cmgpu_exchange_init_external(cmgpu_ctx *ctx, const cmgpu_exchange_transport *t, int rank, int world);
/* owner rank of every sequence for `world` ranks (no exchange needs to be initialised) */
int cmgpu_exchange_owner_table(const cmgpu_ctx *ctx, uint32_t world, uint8_t *owner_out, uint32_t n_sequences);
/* partition the resident batch's records by owner -> exchange -> append the received records to this
 * context's store.  sent_per_rank: world entries or NULL. */
int cmgpu_exchange_step(cmgpu_ctx *ctx, uint64_t *sent_per_rank, uint64_t *n_received);
int cmgpu_exchange_info(const cmgpu_ctx *ctx, int *rank, int *world, uint64_t *records_sent, uint64_t *records_received);
/* The communication plan of one cmgpu_exchange_step round for `rank`, given the world x world count matrix (row r = the
 * records rank r sends to every rank) that the round's first all-gather hands to every rank: the collectives, the local copy
 * and the grouped sends / receives the step posts, in its order.  A pure host function (no device, no RCCL): multi-process
 * callers and tests replay it for all ranks to see that a round's posts meet (every send has its receive, every rank calls
 * the same collectives -- also a rank without records).  capacity < *n_ops: CMGPU_ECAPACITY (ops == NULL: just the count). */
typedef struct cmgpu_exchange_op {
  uint32_t kind;    /* CMGPU_EX_* */
  uint32_t peer;    /* send: destination, receive: source, copy: the rank itself */
  uint64_t records; /* records moved (all-gathers: 64-bit words contributed per rank) */
} cmgpu_exchange_op;
enum { CMGPU_EX_ALLGATHER_COUNTS = 0, CMGPU_EX_ALLGATHER_STATUS = 1, CMGPU_EX_COPY_SELF = 2, CMGPU_EX_SEND = 3, CMGPU_EX_RECV = 4 };
int cmgpu_exchange_plan(uint32_t rank, uint32_t world, const uint64_t *matrix, cmgpu_exchange_op *ops, uint32_t capacity, uint32_t *n_ops);
int cmgpu_exchange_finalize(cmgpu_ctx *ctx);
/* host <-> device copy through the library's HIP runtime, for a host-staged transport; kind 1: host -> device, 2: device -> host */
int cmgpu_memcpy(cmgpu_ctx *ctx, void *dst, const void *src, uint64_t bytes, int kind);

/* ---- --SAM --------------------------------------------------------------
 * With params.output_format == CMGPU_FORMAT_SAM the reported mappings are aligned with the
 * banded affine-gap DP the reference uses for SAM (ksw_semi_global3, src/ksw.cc:505-626, called
 * from src/mapping_generator.h:723-761, 807-854): start / end come from that alignment (MAPQ is
 * computed from them), and every reported read carries CIGAR, NM and MD
 * (GenerateNMAndMDTag, src/alignment.cc:85-139).  cmgpu_sam_record = the arguments of SAMMapping's
 * constructor (src/sam_mapping.h:151-190, src/mapping_generator.cc:43-57, 84-108) except the
 * strings the host already holds (name, sequence, quality).  Slots: 2*i / 2*i+1 for pair i
 * (read 1 / read 2), i for single-end read i; valid == 0: the read has no record.
 * CIGAR words (BAM encoding len<<4|op) and MD text sit in fixed-size slots of the pools. */
#define CMGPU_SAM_CIGAR_CAP 64
typedef struct cmgpu_sam_record {
  uint32_t read_id;
  uint32_t rid;
  uint32_t pos;    /* 0-based start on the reference */
  uint32_t mpos;
  int32_t mrid;    /* -1: no mate */
  int32_t tlen;
  uint32_t nm;
  uint16_t flag;
  uint16_t n_cigar;
  uint16_t md_len;
  uint8_t mapq;
  uint8_t strand;  /* 1 = + (SAMMapping::is_rev_ stores exactly this) */
  uint8_t is_unique;
  uint8_t valid;
  uint16_t length_after_trim; /* bases of the read that were mapped (adapter trimming) */
} cmgpu_sam_record;
/* slots and MD bytes per slot of the last mapped batch */
int cmgpu_sam_layout(const cmgpu_ctx *ctx, uint64_t *n_slots, uint32_t *md_cap);
/* copies records, n_slots * CMGPU_SAM_CIGAR_CAP cigar words and n_slots * md_cap MD bytes to the host */
int cmgpu_download_sam(cmgpu_ctx *ctx, cmgpu_sam_record *records, uint32_t *cigar_pool, char *md_pool);
/* SAM text (src/mapping_writer.cc:312-356): @SQ header unless append, records sorted by
 * SAMMapping::operator< (src/sam_mapping.h:193-199), duplicate removal on operator== in the
 * low-memory or in-memory flavour, MAPQ filter.  names / bases / quals / offsets: the batch(es) the
 * slots refer to (mate-2 arrays NULL for single-end data). */
int64_t cmgpu_write_sam(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *params,
                        const cmgpu_sam_record *records, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                        const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                        const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                        const char *quals2, const uint32_t *offsets2, const char *out_path);
/* single-cell data: the (corrected) barcode keys of the batch last mapped with cmgpu_map_pairs_barcoded /
 * cmgpu_map_single_barcoded, one per pair / read (what the reference keeps in SAMMapping::cell_barcode_) */
int cmgpu_download_barcode_keys(cmgpu_ctx *ctx, uint64_t *keys);
/* cmgpu_write_sam for single-cell data: cell_barcode_ takes part in operator< / operator==
 * (src/sam_mapping.h:201-212) and every line ends with CB:Z:<barcode> (src/mapping_writer.cc:350-354).
 * barcode_keys: one per pair / read, in slot order */
int64_t cmgpu_write_sam_barcoded(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *params,
                                 const cmgpu_sam_record *records, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                                 const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                                 const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                                 const char *quals2, const uint32_t *offsets2, const uint64_t *barcode_keys, uint32_t barcode_length,
                                 const char *out_path);
/* the same with --barcode-translate (src/barcode_translator.h:43-101, src/mapping_writer.cc:350-354): the CB:Z: value goes through the
 * translation table, whose (inflated) text the caller hands over -- lines "to<TAB or ,>from".  A barcode that is not in the
 * table: CMGPU_EFORMAT (the reference exits with "Barcode does not exist in the translation table."). */
int64_t cmgpu_write_sam_barcoded_translated(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *params,
                                            const cmgpu_sam_record *records, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                                            const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                                            const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                                            const char *quals2, const uint32_t *offsets2, const uint64_t *barcode_keys, uint32_t barcode_length,
                                            const char *translate_table, uint64_t translate_table_bytes, const char *out_path);

/* ---- device-side post-processing -------------------------------------
 * Replaces, for BED output: MappingProcessor::SortOutputMappings / RemovePCRDuplicate
 * (src/mapping_processor.h:100-202), the low-memory merge's duplicate handling
 * (src/mapping_writer.h:166-376), ApplyTn5ShiftOnMappings, the MAPQ filter and
 * MappingWriter::AppendMapping (src/mapping_writer.cc:44-52, 72-83, 119-131).
 * Batches append their records to a store in HBM; one call sorts (radix sort on the
 * operator< key), removes duplicates (low_memory_mode: first record with the maximal MAPQ of an
 * operator== run; otherwise RemovePCRDuplicate's last record of the run, Tn5 shift applied
 * before the sort as src/chromap.h:1323-1331 does), filters, and renders the text in HBM.
 * The bytes equal cmgpu_write_bed_* / the reference's output file. */
#define CMGPU_TEXT_BED_PE 0
#define CMGPU_TEXT_BED_SE 1
#define CMGPU_TEXT_BED_PE_BC 2
#define CMGPU_TEXT_TAGALIGN_PE 3    /* --TagAlign, paired-end bulk: two lines per fragment (src/mapping_writer.cc:84-117) */
#define CMGPU_TEXT_TAGALIGN_PE_BC 4
#define CMGPU_TEXT_BED_SE_BC 5      /* single-end single-cell BED: MappingWithBarcode (src/bed_mapping.h:11-56, src/mapping_writer.cc:6-25) */
#define CMGPU_TEXT_TAGALIGN_SE_BC 6 /* --TagAlign, single-end single-cell: chr start end N mapq strand (src/mapping_writer.cc:26-34) */
int cmgpu_store_clear(cmgpu_ctx *ctx);
/* room for n_records ahead of time (the store otherwise doubles on demand: allocation + copy + synchronous free) */
int cmgpu_store_reserve(cmgpu_ctx *ctx, uint64_t n_records, int barcoded);
/* appends the records of the last cmgpu_map_* call (still resident); n_total = store size */
int cmgpu_store_append_resident(cmgpu_ctx *ctx, uint64_t *n_total);
/* appends n records from a host or device array; barcoded = 0: cmgpu_record entries,
 * barcoded = 1: cmgpu_record_bc entries -- e.g. the receive buffer of the multi-GPU exchange */
int cmgpu_store_append(cmgpu_ctx *ctx, const void *records, uint64_t n, int on_device, int barcoded);
int cmgpu_store_format(cmgpu_ctx *ctx, int kind, const char *const *names, uint32_t n_sequences, const cmgpu_params *params,
                       uint32_t barcode_length, uint64_t *n_lines, uint64_t *n_bytes);
/* --preset hic: the store holds cmgpu_pairs_record entries (cmgpu_store_append_resident after a split-alignment batch);
 * sorted as MappingWriter<PairsMapping> sorts them (src/pairs_mapping.h:41-47), MAPQ filter, one text line per record
 * (src/mapping_writer.cc:400-423).  read_names: the names of reads read_id_base .. read_id_base + n_read_names - 1
 * concatenated, read_name_offsets[n_read_names + 1] into it.  The header lines: cmgpu_write_pairs_header.
 * The bytes equal cmgpu_write_pairs' lines. */
int cmgpu_store_format_pairs(cmgpu_ctx *ctx, const char *const *names, uint32_t n_sequences, const cmgpu_params *params,
                             const char *read_names, const uint64_t *read_name_offsets, uint32_t n_read_names, uint32_t read_id_base,
                             uint64_t *n_lines, uint64_t *n_bytes);
int cmgpu_write_pairs_header(const char *const *names, const uint32_t *lengths, uint32_t n_sequences, const uint32_t *pairs_rank,
                             const char *out_path);
int cmgpu_store_text(cmgpu_ctx *ctx, char *out, uint64_t capacity);
int cmgpu_store_write_text(cmgpu_ctx *ctx, const char *path, int append);
int cmgpu_store_info(const cmgpu_ctx *ctx, uint64_t *n_records, uint64_t *text_bytes, uint64_t *text_lines);

/* ---- FASTQ ingest on the device ---------------------------------------
 * Replaces, for 4-line FASTQ text, kseq_read + SequenceBatch::LoadOneSequenceAndSaveAt
 * (src/sequence_batch.cc:22-62): the host hands over raw (inflated) file bytes chunk by chunk;
 * lines, records, the skip of empty sequences and the SoA batch are built in HBM.
 * stream: 0 = read 1, 1 = read 2, 2 = cell barcodes (bases + qualities).
 *   cmgpu_fastq_scan   uploads one chunk (< 4 GiB) and counts its complete, non-empty records;
 *                      final_chunk != 0: the text ends the file (a missing last newline is fine).
 *                      CMGPU_EFORMAT: not 4-line FASTQ (multi-line records need a host parser).
 *                      Scans (this call and cmgpu_fastq_scan_bgzf) of DIFFERENT streams of one context may be
 *                      made from different host threads at the same time: each stream has a HIP stream and
 *                      scratch of its own; every other call on a context is one thread at a time.
 *   cmgpu_fastq_take   gathers the first n of them as this stream's part of the NEXT batch (staging buffers) and
 *                      returns how many bytes of the chunk they (and skipped records) cover --
 *                      the host resubmits the rest in front of the next chunk.
 *   cmgpu_fastq_commit declares the batch (n records taken from every participating stream): the staging buffers
 *                      become the resident batch; cmgpu_map_resident then maps it.
 * Overlap: scans and takes touch only the streams' own buffers, HIP streams and the staging buffers, so ONE thread may
 * scan and take the next batch while ANOTHER is inside cmgpu_map_resident / cmgpu_store_append_resident with the
 * committed one (chromap-amd does; tests/test_gpu_ingest.py::test_next_batch_taken_while_the_last_is_mapped).
 * cmgpu_fastq_commit itself must not run beside a mapping call of the same context. */
/* --read-format for one stream (SequenceEffectiveRange, src/sequence_effective_range.h, src/chromap.cc:825-866):
 * up to four [start, end] base ranges (0-based, inclusive, end -1 = last base) are concatenated, then the result is
 * reverse-complemented (bases) / reversed (qualities) when strand is '-'.  Applies to the following cmgpu_fastq_take calls. */
int cmgpu_fastq_set_format(cmgpu_ctx *ctx, int stream, int n_ranges, const int32_t *starts, const int32_t *ends, char strand);
int cmgpu_fastq_scan(cmgpu_ctx *ctx, int stream, const char *text, uint64_t n_bytes, int final_chunk, uint32_t *n_records);
/* BGZF input (bgzip: gzip members of at most 64 KiB with their size in a 'BC' extra field) inflated ON THE DEVICE: `blocks` holds
 * whole COMPRESSED blocks as they stand in the file (the end-of-file marker block may be among them); a lane per block decodes
 * its DEFLATE codes, a wave per block copies the matches inside the block's place in the stream's text and checks its CRC-32 --
 * what zlib's gzread does for the reference, one stream per file (src/sequence_batch.cc:22-62).  The text stays on the device: what cmgpu_fastq_take does not
 * consume is kept in front of the blocks of the next call (the host does not resubmit anything; a call after which nothing is
 * taken simply extends the text).  CMGPU_EFORMAT: not BGZF, a truncated or a damaged block (cmgpu_last_error says which), or not
 * 4-line FASTQ.  A call of cmgpu_fastq_scan on the stream returns it to host-resubmitted text. */
int cmgpu_fastq_scan_bgzf(cmgpu_ctx *ctx, int stream, const void *blocks, uint64_t n_bytes, int final_chunk, uint32_t *n_records);
int cmgpu_fastq_take(cmgpu_ctx *ctx, int stream, uint32_t n, uint64_t *bytes_consumed);
int cmgpu_fastq_commit(cmgpu_ctx *ctx, uint32_t n, uint32_t first_read_id, int paired, int barcoded);
/* cmgpu_compute_barcode_abundance over the barcodes last taken from stream 2; feed the barcode
 * file in whole reference batches and stop when *done is set (20 M sampled, src/chromap.h:211). */
int cmgpu_barcode_abundance_resident(cmgpu_ctx *ctx, uint64_t *num_sample_barcodes, int *done);

/* Host post-processing that defines the final BED bytes: sort by (rid, operator<),
 * PCR-duplicate removal as in the low-memory merge, MAPQ filter, Tn5 shift, text
 * formatting (src/mapping_writer.h:166-376, src/mapping_writer.cc:72-83). Sorts `records`
 * in place.  names: n_sequences reference names. */
int64_t cmgpu_write_bed_pe(const char *const *names, uint32_t n_sequences, const cmgpu_params *params,
                           cmgpu_record *records, uint64_t n_records, const char *out_path);

/* pairs output of --preset hic (src/mapping_writer.cc:381-420): sort by PairsMapping::operator<
 * (src/pairs_mapping.h:40-43), MAPQ filter, header + one line per record.  read_names[read_id -
 * read_id_base] is the name of read 1 of the pair (names never go to the device). */
int64_t cmgpu_write_pairs(const char *const *names, const uint32_t *lengths, uint32_t n_sequences,
                          const cmgpu_params *params, cmgpu_pairs_record *records, uint64_t n_records,
                          const char *const *read_names, uint32_t read_id_base, const char *out_path);
int64_t cmgpu_write_pairs_ranked(const char *const *names, const uint32_t *lengths, uint32_t n_sequences,
                                 const cmgpu_params *params, cmgpu_pairs_record *records, uint64_t n_records,
                                 const char *const *read_names, uint32_t read_id_base, const uint32_t *pairs_rank,
                                 const char *out_path);

/* Host loaders mirroring Index::Load and SequenceBatch::LoadAllSequences for callers
 * that do not have the reference's own objects (the CLI, Python).  Free with
 * cmgpu_free_host_index / cmgpu_free_host_ref. */
int cmgpu_load_index_file(const char *path, cmgpu_index_view *out);
void cmgpu_free_host_index(cmgpu_index_view *v);
int cmgpu_load_reference_fasta(const char *path, cmgpu_ref_view *out);
void cmgpu_free_host_ref(cmgpu_ref_view *v);

#ifdef __cplusplus
}
#endif
#endif /* CHROMAP_AMD_H_ */
