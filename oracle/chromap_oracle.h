/*
 * chromap_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A from-scratch, sequential C restatement of the reference's per-read mapping hot
 * path (haowenz/chromap v0.3.3-r521, /root/reference/src).  It exists to check the
 * HIP path and to serve as the "port" CPU baseline in bench.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (chromap_amd/) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks this code against
 * outputs of the reference itself (oracle/_ref/chromap, built unchanged from
 * /root/reference by oracle/Makefile) committed under tests/golden/.
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef CHROMAP_ORACLE_H_
#define CHROMAP_ORACLE_H_
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- index (index.cc:91-169, khash.h:358-386) ---------------------------- */
typedef struct ora_index {
  int k, w;
  uint32_t n_keys;
  uint32_t n_buckets, size, n_occupied, upper_bound;
  uint32_t *flags;
  uint64_t *keys, *vals;
  uint32_t n_occ;
  uint64_t *occ;
} ora_index;

/* ---- reference sequences (sequence_batch.cc:84-120) ---------------------- */
typedef struct ora_ref {
  uint32_t n_seq;
  char **name;
  char **seq; /* raw bytes, case preserved, NUL terminated */
  uint32_t *len;
} ora_ref;

/* subset of MappingParameters used on the path (mapping_parameters.h:18-89) */
typedef struct ora_params {
  int error_threshold;       /* -e, 8 */
  int min_num_seeds;         /* -s, 2 */
  int max_seed_freq0;        /* -f, 500 */
  int max_seed_freq1;        /* 1000 */
  int max_insert_size;       /* -l, 1000 (atac/chip 2000) */
  int min_read_length;       /* 30 */
  int max_num_best_mappings; /* 1 */
  int drop_repetitive_reads; /* 500000 */
  int trim_adapters;
  int split_alignment;
  int mapq_threshold;        /* -q, 30 */
  int remove_pcr_duplicates;
  int tn5_shift;
  int low_mem;
  int bc_error_threshold;               /* --bc-error-threshold, 1 */
  int output_mappings_not_in_whitelist; /* --output-mappings-not-in-whitelist */
  int output_format;                    /* 0: BED / pairs records, 1: --SAM (ksw alignment, CIGAR, NM, MD), 2: --TagAlign text from the BED writers */
  int dedup_at_bulk_level;              /* single-cell data: --remove-pcr-duplicates-at-bulk-level (ora_write_bed_pe_bc_bulk) */
  double bc_probability_threshold;      /* --bc-probability-threshold, 0.9 */
} ora_params;

/* constructor arguments of PairedEndMappingWithoutBarcode (bed_mapping.h:191-206)
 * plus rid (implicit vector index in the reference, mapping_generator.cc:116). */
typedef struct ora_record {
  uint32_t read_id;
  uint32_t rid;
  uint32_t fragment_start;
  uint16_t fragment_length;
  uint8_t mapq; /* already reduced to 6 bits */
  uint8_t direction;
  uint8_t is_unique;
  uint8_t num_dups;
  uint16_t pos_aln_len;
  uint16_t neg_aln_len;
} ora_record;

/* constructor arguments of PairsMapping (pairs_mapping.h:25-38) without name and barcode;
 * written instead of ora_record (same 24-byte slot) when split_alignment is set */
typedef struct ora_pairs_record {
  uint32_t read_id, rid1, rid2, pos1, pos2;
  uint8_t strand1, strand2; /* 1 = positive */
  uint8_t mapq, is_unique;
} ora_pairs_record;

/* counters of Chromap::OutputMappingStatistics (chromap.cc:808-823) */
typedef struct ora_stats {
  uint64_t num_candidates, num_mappings, num_mapped_reads, num_uniquely_mapped_reads;
  uint64_t probe_steps;   /* khash buckets visited (16 B each) */
  uint64_t occ_reads;     /* occurrence-table entries read (8 B each) */
  uint64_t lookups;       /* kh_get calls */
  uint64_t num_minimizers;
  uint64_t num_verifications; /* banded alignments run */
  uint64_t num_shortcut;      /* reads resolved by the all-minimizer shortcut */
  uint64_t num_rescue;        /* mate-rescue strand searches */
  uint64_t num_trimmed;
} ora_stats;

/* PairedEndMappingWithBarcode constructor arguments (bed_mapping.h:128-144) + rid */
typedef struct ora_record_bc {
  ora_record r;
  uint64_t barcode;
} ora_record_bc;

/* barcode whitelist with abundance (chromap.cc:388-548): khash k64_seq in the reference, any
 * map here */
typedef struct ora_whitelist ora_whitelist;
ora_whitelist *ora_whitelist_load(const char *path, uint32_t barcode_length);
void ora_whitelist_free(ora_whitelist *w);
/* ComputeBarcodeAbundance (chromap.cc:492-548) over the n barcodes of the input (read batches
 * of 500000, stops once 20000000 whitelisted barcodes were seen). returns -1 when fewer than
 * 5% of the first batch are whitelisted (the reference exits), else num_sample_barcodes_. */
long ora_whitelist_abundance(ora_whitelist *w, const char *bc, const uint32_t *bc_off, uint32_t n);
uint32_t ora_whitelist_size(const ora_whitelist *w);
/* export for the device table: keys[i], counts[i], i < size */
void ora_whitelist_export(const ora_whitelist *w, uint64_t *keys, uint32_t *counts);
uint64_t ora_seed_from_sequence(const char *seq, uint32_t seq_len, uint32_t start, uint32_t seed_len); /* utils.h:111-129 */
/* Chromap::CorrectBarcodeAt (chromap.cc:572-799); bc is modified in place. returns whitelisted */
int ora_correct_barcode(const ora_params *p, const ora_whitelist *w, char *bc, const char *qual, uint32_t len,
                        uint64_t *num_in_whitelist, uint64_t *num_corrected);

typedef struct ora_ctx ora_ctx;

/* --SAM: what SAMMapping's constructor receives (sam_mapping.h:151-190; mapping_generator.cc:43-57,
 * 84-108) without the strings the host already holds (name, sequence, quality).  One slot per
 * read: slot 2*i / 2*i+1 for pair i (read 1 / read 2), slot i for single-end read i; valid = 0
 * when the read produced no record.  cigar and MD live in fixed-size slots of the pools passed to
 * ora_map_*_sam: cigar_pool[slot * ORA_SAM_CIGAR_CAP ...], md_pool[slot * md_cap ...]. */
#define ORA_SAM_CIGAR_CAP 64
typedef struct ora_sam_record {
  uint32_t read_id;
  uint32_t rid;
  uint32_t pos;      /* 0-based ref_start_position */
  uint32_t mpos;
  int32_t mrid;      /* -1: no mate */
  int32_t tlen;
  uint32_t nm;
  uint16_t flag;
  uint16_t n_cigar;
  uint16_t md_len;
  uint8_t mapq;
  uint8_t strand;    /* 1 = + (SAMMapping::is_rev_ holds exactly this) */
  uint8_t is_unique;
  uint8_t valid;
  uint16_t reserved;
} ora_sam_record;

long ora_map_pairs_sam(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1, const uint32_t *r1_off,
                       const char *r2, const uint32_t *r2_off, ora_sam_record *out /* 2n */, uint32_t *cigar_pool,
                       char *md_pool, uint32_t md_cap, ora_stats *stats);
long ora_map_single_sam(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r, const uint32_t *r_off,
                        ora_sam_record *out /* n */, uint32_t *cigar_pool, char *md_pool, uint32_t md_cap, ora_stats *stats);
/* SAM text (mapping_writer.cc:312-356): @SQ header, records sorted by SAMMapping::operator<
 * (sam_mapping.h:193-199), duplicate removal on operator== when remove_pcr_duplicates, MAPQ filter.
 * names/bases/quals: the batch (mate 2 arrays NULL for single-end); n_slots = 2n or n. */
long ora_write_sam(const ora_ref *ref, const ora_params *p, const ora_sam_record *rec, long n_slots, int paired,
                   const uint32_t *cigar_pool, const char *md_pool, uint32_t md_cap, const char *const *names1,
                   const char *const *names2, const char *b1, const char *q1, const uint32_t *o1, const char *b2,
                   const char *q2, const uint32_t *o2, const uint32_t *len_after_trim /* per slot */, const char *out_path);
/* single-cell --SAM: barcode in the sort / duplicate keys and as CB:Z (sam_mapping.h:201-212, mapping_writer.cc:350-354) */
long ora_write_sam_bc(const ora_ref *ref, const ora_params *p, const ora_sam_record *rec, long n_slots, int paired,
                      const uint32_t *cigar_pool, const char *md_pool, uint32_t md_cap, const char *const *names1,
                      const char *const *names2, const char *b1, const char *q1, const uint32_t *o1, const char *b2,
                      const char *q2, const uint32_t *o2, const uint64_t *barcode_keys, uint32_t barcode_length, const char *out_path);
long ora_map_pairs_bc_sam(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1, const uint32_t *r1_off,
                          const char *r2, const uint32_t *r2_off, char *bc, const char *bc_qual, const uint32_t *bc_off,
                          const ora_whitelist *w, ora_sam_record *out, uint32_t *cigar_pool, char *md_pool, uint32_t md_cap,
                          uint64_t *keys_per_pair, ora_stats *stats);
/* single-end reads with cell barcodes (MappingWithBarcode, bed_mapping.h:10-56; chromap.h:385-472) */
long ora_map_single_bc(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r, const uint32_t *r_off,
                       char *bc, const char *bc_qual, const uint32_t *bc_off, const ora_whitelist *w, ora_record_bc *out,
                       ora_stats *stats, uint64_t *num_in_whitelist, uint64_t *num_corrected);
/* BED (tagalign = 0) or TagAlign text with the reference's low-memory merge or in-memory duplicate removal,
 * cell-level or bulk-level (p->dedup_at_bulk_level) */
long ora_write_se_bc(const ora_ref *ref, const ora_params *p, ora_record_bc *rec, long n, uint32_t barcode_length,
                     const ora_whitelist *w, int tagalign, const char *out_path);
int ora_ksw_semi_global3(int qlen, const char *query, int tlen, const char *target, int w, uint32_t *cigar, int cigar_cap,
                         int *n_cigar, int *start, int *end);

void ora_default_params(ora_params *p);
void ora_preset(ora_params *p, const char *preset); /* chromap_driver.cc:247-275 */

uint64_t ora_hash64(uint64_t key, uint64_t mask); /* utils.h:76-85 */

/* minimizer_generator.cc:7-139.  out arrays need capacity >= len. returns count. */
int ora_minimizers(const char *seq, uint32_t len, uint32_t seq_index, int k, int w,
                   uint64_t *out_hash, uint64_t *out_hit);

int ora_index_load(const char *path, ora_index *idx);
int ora_index_save(const char *path, const ora_index *idx);
/* index.cc:12-89 with khash.h:246-350 put/resize replayed, so the file is byte-identical */
int ora_index_build(const ora_ref *ref, int k, int w, ora_index *idx);
void ora_index_free(ora_index *idx);
/* plumbing for bench.py: index from the device layout exported by cmgpu_export_index */
int ora_index_from_buckets(const uint64_t *buckets, uint32_t n_buckets, const uint64_t *occ, uint32_t n_occ,
                           int k, int w, ora_index *idx);
/* khash.h:232-245 with hash/eq of index_utils.h:13-17. returns bucket or n_buckets; *steps += visited */
uint32_t ora_kh_get(const ora_index *idx, uint64_t key, uint64_t *steps);

int ora_ref_load(const char *fasta_path, ora_ref *ref);
void ora_ref_free(ora_ref *ref);

/* alignment.cc:141-192 */
int ora_banded_align(int e, const char *pattern, const char *text, int read_length,
                     int *mapping_end_position);
/* alignment.cc:656-718 */
void ora_banded_traceback(int e, int min_num_errors, const char *pattern, const char *text,
                          int read_length, int *mapping_start_position);

ora_ctx *ora_create(const ora_index *idx, const ora_ref *ref, const ora_params *p);
void ora_destroy(ora_ctx *c);
/* --chr-order: rank[i] = position of reference sequence i in the output order (every value 0..n-1 once) */
int ora_set_chr_order(ora_ctx *c, const uint32_t *rank, uint32_t n);
/* the reference as the mapping stages and the writers see it (reordered after ora_set_chr_order) */
const ora_ref *ora_ctx_ref(const ora_ctx *c);
/* --pairs-natural-chr-order: rank (over the possibly reordered reference) that decides which end of a pair comes first */
int ora_set_pairs_chr_order(ora_ctx *c, const uint32_t *rank, uint32_t n);
long ora_write_pairs_ranked(const ora_ref *ref, const ora_params *p, ora_pairs_record *rec, long n, const char *const *read_names,
                            const uint32_t *pairs_rank, const char *out_path);

/* Body of the taskloop chromap.h:892-1143 for the n pairs of one input file (read
 * batches of 500000 pairs, taskloop tasks of ~5000 pairs, each task owning a fresh
 * std::mt19937(11) -- see make_chunks() in the .c file).  Reads are concatenated ASCII
 * with n+1 offsets.  Returns number of records written (capacity must be
 * >= n * max_num_best_mappings).  stats are accumulated. */
long ora_map_pairs(ora_ctx *c, uint32_t n, uint32_t first_read_id, const char *r1,
                   const uint32_t *r1_off, const char *r2, const uint32_t *r2_off,
                   ora_record *out, ora_stats *stats);
/* Same work with the tasks spread over OpenMP threads (cpu_baseline leg); results are
 * identical to the sequential call, as in the reference at any -t. */
long ora_map_pairs_mt(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id,
                      const char *r1, const uint32_t *r1_off, const char *r2,
                      const uint32_t *r2_off, ora_record *out, ora_stats *stats);

/* Single-cell variant: barcode correction in front of every pair (chromap.h:897-909), records
 * carry the (corrected) barcode key.  bc is corrected in place. */
long ora_map_pairs_bc(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1,
                      const uint32_t *r1_off, const char *r2, const uint32_t *r2_off, char *bc,
                      const char *bc_qual, const uint32_t *bc_off, const ora_whitelist *w,
                      ora_record_bc *out, ora_stats *stats, uint64_t *num_in_whitelist, uint64_t *num_corrected);
/* BED for PairedEndMappingWithBarcode (mapping_writer.cc:119-131), cell-level dedup */
long ora_write_bed_pe_bc(const ora_ref *ref, const ora_params *p, ora_record_bc *rec, long n, uint32_t barcode_length,
                         const char *out_path);
/* FASTQ with qualities: returns n, allocates bases, quals (same offsets) and off */
/* same with duplicate removal at bulk level (--remove-pcr-duplicates-at-bulk-level, the default without --preset atac) */
long ora_write_bed_pe_bc_bulk(const ora_ref *ref, const ora_params *p, ora_record_bc *rec, long n, uint32_t barcode_length,
                              const ora_whitelist *w, const char *out_path);
long ora_read_fastq_qual(const char *path, char **bases, char **quals, uint32_t **off);

/* single-end reads (chromap.h:385-472): bulk records, positive/negative_alignment_length 0 */
long ora_map_single(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r, const uint32_t *r_off,
                    ora_record *out, ora_stats *stats);
long ora_write_bed_se(const ora_ref *ref, const ora_params *p, ora_record *rec, long n, const char *out_path);

/* per-pair trace for stage-level comparisons with the HIP path */
typedef struct ora_trace {
  uint32_t len1, len2;           /* read lengths after trimming */
  uint32_t n_mm1, n_mm2;         /* minimizers */
  uint32_t n_cand1, n_cand2;     /* candidates entering verification */
  uint32_t n_draft1, n_draft2;   /* draft mappings */
  int32_t min_err1, min_err2, nbest1, nbest2, second1, second2, nsecond1, nsecond2;
  uint32_t rep1, rep2;           /* repetitive_seed_length */
  int32_t min_sum, nbest, second_sum, nsecond;
  int32_t force_mapq;
} ora_trace;
void ora_set_trace(ora_ctx *c, ora_trace *trace /* n entries, or NULL */);

/* mapping_writer.h:166-376 (low-mem) / chromap.h:1322-1355 (in-memory) + writer
 * mapping_writer.cc:72-117 (PE bulk BED).  Sorts records in place. Returns #lines. */
long ora_write_bed_pe(const ora_ref *ref, const ora_params *p, ora_record *rec, long n,
                      const char *out_path);

/* pairs output for --preset hic (mapping_writer.cc:381-420); read_names[read_id] */
long ora_write_pairs(const ora_ref *ref, const ora_params *p, ora_pairs_record *rec, long n,
                     const char *const *read_names, const char *out_path);

/* FASTQ/FASTA reader (kseq.h semantics: name up to whitespace, multi-line ok) used by
 * tests to build the SoA batches. Returns number of records, allocates bases and off. */
long ora_read_fastx(const char *path, char **bases, uint32_t **off);

#ifdef __cplusplus
}
#endif
#endif
