/* chromap_amd_debug.h -- measurement, test and fixture hooks of libchromap_amd.so.
 *
 * NOT part of the drop-in boundary (include/chromap_amd.h holds SURVEY.md 8(b)'s contract: create / map / wait / destroy,
 * the writers and the post-processing a maintainer binds).  What is declared here exists for bench.py, the parity tests and
 * the profiling tools: synthetic genomes and read batches generated on the device, the kernel-only probe measurement,
 * measurement knobs, stage-level views of the last mapped batch and exports of the resident index / reference for the
 * oracle.  Plain C like the boundary header.
 *
 * Environment variables the library / CLI read (measurement and test aids, no effect on results):
 *   CM_NO_KEY32     the cooperative hit-list stage keeps the reference's 64-bit keys although the reference would fit 32-bit
 *                   global coordinates (the path of references beyond 4 Gbases; tests/test_gpu_stages.py)
 *   CM_DEBUG_POOL   per mapped range: size of and demand on the rescue-hit pool, time of its (re)allocation, on stderr
 *   CM_CLI_TIMES    chromap-amd: seconds of every scan / take + commit / map / store call on stderr
 */
#ifndef CHROMAP_AMD_DEBUG_H_
#define CHROMAP_AMD_DEBUG_H_

#include "chromap_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Bench/test helper standing in for Index::Construct (src/index.cc:12-89) on a synthetic
 * genome: n_sequences chromosomes of uniform random bases (xorshift seeded with seed) are
 * generated and indexed on the device (functionally identical lookup results; bucket
 * placement differs from khash's).  total_bases may be GRCh38-sized (3.1e9). */
int cmgpu_create_synthetic(uint64_t total_bases, uint32_t n_sequences, uint64_t seed, int32_t kmer_size,
                           int32_t window_size, const cmgpu_params *params, int device_id, cmgpu_ctx **out);

/* The same with planted repeat families: n_families elements of element_len random bases, `copies`
 * copies each, spread over the genome in either orientation, every copy's bases replaced with probability
 * `divergence` -- so that frequent seeds, multi-mappers and mate rescue occur as on a real genome. */
int cmgpu_create_synthetic_repeats(uint64_t total_bases, uint32_t n_sequences, uint64_t seed, int32_t kmer_size,
                                   int32_t window_size, const cmgpu_params *params, int device_id, uint32_t n_families,
                                   uint32_t copies, uint32_t element_len, double divergence, cmgpu_ctx **out);

/* The same with a whole repeat landscape instead of one family class.  profile 1: 64-kb tiles, 20 % of them packed with
 * SINE-like 300-base elements (128 families, ~10^4 copies each on 3.1 Gb, 5-15 % divergence per copy), 12 % with LINE-like
 * 3-kb elements (256 families, ~360 copies each, 1-5 %), 2 % satellite arrays (171-base units, 2 % replaced): 22.5 % of the
 * bases are repeat-derived (bench.py's third workload).  profile 2: the same kinds with 40 % / 28 % / 3 % of the tiles: 47 % of
 * the bases repeat-derived (~19 000 copies per SINE-like family, ~830 per LINE-like family): bench.py's fourth genome. */
int cmgpu_create_synthetic_profile(uint64_t total_bases, uint32_t n_sequences, uint64_t seed, int32_t kmer_size,
                                   int32_t window_size, const cmgpu_params *params, int device_id, uint32_t profile,
                                   cmgpu_ctx **out);

/* Generates n_pairs synthetic read pairs from the ctx's reference directly in HBM
 * (fragment length uniform in [frag_min, frag_max), substitution rate sub_rate, R1/R2
 * swapped with p = 0.5) and makes them the resident batch. */
int cmgpu_generate_resident_batch(cmgpu_ctx *ctx, uint32_t n_pairs, uint32_t read_length, uint32_t frag_min,
                                  uint32_t frag_max, double sub_rate, uint64_t seed);
/* The same with 1-base insertions / deletions at rate indel_rate per base (half each). */
int cmgpu_generate_resident_batch_indels(cmgpu_ctx *ctx, uint32_t n_pairs, uint32_t read_length, uint32_t frag_min,
                                         uint32_t frag_max, double sub_rate, double indel_rate, uint64_t seed);
/* Hi-C shaped pairs (BASELINE config 5): the mates come from two independent loci (60 % within 1 Mb on one sequence, the
 * others anywhere), either orientation each; `chimeric_fraction` of the pairs carry a ligation junction inside one read
 * (25 .. L - 25 bases of its own locus, then the partner's fragment on the other strand) -- the input split alignment
 * (draft_mapping_generator.cc:410-487, alignment.cc:197-376) is for. */
int cmgpu_generate_resident_batch_hic(cmgpu_ctx *ctx, uint32_t n_pairs, uint32_t read_length, double sub_rate, double indel_rate,
                                      double chimeric_fraction, uint64_t seed);
/* The resident batch changes places with the one parked in `slot` (0..7; either may be empty): several distinct
 * batches stay in HBM and take turns (the measurement must not map one batch over and over). */
int cmgpu_swap_resident_batch(cmgpu_ctx *ctx, int slot);
/* Copies the resident batch back to host SoA buffers (for checking against the oracle). */
int cmgpu_download_batch(cmgpu_ctx *ctx, char *read1_bases, uint32_t *read1_offsets, char *read2_bases,
                         uint32_t *read2_offsets);

/* Kernel-level entry for the graded index-probe kernel: looks up n minimizer hashes
 * (device-resident after the call) `repeat` times; returns average kernel time in ms
 * measured with HIP events on the launch stream, and the probe-step / hit counts. */
int cmgpu_probe_bench(cmgpu_ctx *ctx, const uint64_t *hashes, uint64_t n, int repeat, double *avg_ms,
                      uint64_t *probe_steps, uint64_t *hits, uint64_t *occurrences);

/* The same kernel in its other shapes, on the hashes left resident by the last mapped batch:
 * lookups_per_lane 1 / 2 / 4 / 8 independent lookups interleaved per lane, pair_prefetch != 0: the second
 * probe step is requested with the first when both buckets share a 64-byte sector. */
int cmgpu_probe_bench_variant(cmgpu_ctx *ctx, uint64_t n, int repeat, int lookups_per_lane, int pair_prefetch,
                              double *avg_ms, uint64_t *probe_steps, uint64_t *hits);

/* HBM random-gather microbenchmark on the resident table: n independent 16-byte loads at
 * pseudo-random buckets, average kernel time over `repeat` launches (HIP events). */
int cmgpu_gather_bench(cmgpu_ctx *ctx, uint64_t n, int repeat, double *avg_ms);
/* its sweep: loads_per_lane (1, 2, 4, 8, 16) requests in flight per lane, access_bytes 16 (one bucket) or
 * 64 (the whole aligned sector, every fetched byte used) -- the best shape is the ceiling k_probe is held to */
int cmgpu_gather_sweep(cmgpu_ctx *ctx, uint64_t n, int repeat, int loads_per_lane, int access_bytes, double *avg_ms);

/* Measurement knobs (the defaults are the measured best): "probe_lookups_per_lane" 1/2/4/8,
 * "probe_pair_prefetch" 0/1, "mm_chunks" 1..8, "prep_kernel" 0/1, "item_limit" (largest dense
 * intermediate array, in entries; batches that need more are mapped in sub-batches), "heavy_wave_max" /
 * "heavy_block_max" / "heavy_big_max" (size classes of the cooperative kernel for long hit lists; -1 on the first:
 * one-lane path), "heavy_last" (reads with long hit lists processed in waves of their own: 0 auto, 1 always, -1 never),
 * "lanes" 1..8 (ranges of a batch mapped side by side), "h2d_copy_blocks" / "d2h_copy_blocks" (blocks of the copy kernel
 * that moves page-locked host memory over the link instead of the copy engine; 0 = hipMemcpyAsync),
 * "first_read_id" (read id of the resident batch's first pair; device-generated batches start at 0),
 * "probe_table_shift" 0..4 (the pipeline probes a device copy of the index table re-hashed into 2^shift times as many
 * buckets: same lookups, fewer buckets visited; 0 = the file's table), "coop" (bit mask of the stages whose long lists go to
 * groups of lanes: 1 hit lists, 2 rescue hits, 4 pair filter, 8 acceptance, 16 pairing; 0 = the one-lane / bitonic forms),
 * "speculative_sizes" 0/1 (candidate arrays sized from the previous batch, checked on the device, one re-run when too small; the
 * kernels of the long-list classes launched for the classes the previous range used, one re-run when another class has items; -1: on,
 * starting from an empty launch set -- the tests' way to force that re-run),
 * "verify_planes" 0/1 (alignments of the verification on bit planes of the reference and the reads instead of their bytes),
 * "long_read_fused" 0/1 (reads longer than 69 bases: trimming + minimizers in one pass instead of count / scan / fill).
 * Every setting gives the same records (tests/test_gpu_parity.py runs the fuzz data under each). */
int cmgpu_set_option(cmgpu_ctx *ctx, const char *name, int64_t value);
int cmgpu_get_option(const cmgpu_ctx *ctx, const char *name, int64_t *value);

/* ---- stage-level view of the last mapped batch (parity tests of the gfx950 build against the oracle's trace) ----
 * cmgpu_trace has the layout of oracle/chromap_oracle.h: ora_trace: per pair the read lengths after trimming
 * (chromap.cc:176-289), minimizer counts (minimizer_generator.cc:7-139), candidates entering verification
 * (candidate_processor.cc:12-263), draft mappings and error bookkeeping (draft_mapping_generator.cc:9-557),
 * pairing (mapping_generator.h:160-253).  Paired-end batches mapped in one piece only. */
typedef struct cmgpu_trace {
  uint32_t len1, len2, n_mm1, n_mm2, n_cand1, n_cand2, n_draft1, n_draft2;
  int32_t min_err1, min_err2, nbest1, nbest2, second1, second2, nsecond1, nsecond2;
  uint32_t rep1, rep2;
  int32_t min_sum, nbest, second_sum, nsecond, force_mapq;
} cmgpu_trace;
int cmgpu_debug_trace(cmgpu_ctx *ctx, cmgpu_trace *out, uint64_t capacity);
/* minimizers of one read (index 2 * pair + mate) / of all reads of the last mapped batch: hash and position << 1 | strand */
int cmgpu_debug_minimizers(cmgpu_ctx *ctx, uint32_t read, uint64_t *hash_out, uint32_t *ps_out, uint32_t capacity, uint32_t *n_out);
int cmgpu_debug_minimizers_all(cmgpu_ctx *ctx, uint32_t *cnt_out, uint32_t *off_out, uint64_t *hash_out, uint32_t *ps_out,
                               uint64_t capacity, uint64_t *n_total);

/* Per-stage timing of the last cmgpu_map_* call (HIP events on the launch stream).
 * names/ms arrays of capacity cap; returns number of stages. */
int cmgpu_last_timings(const cmgpu_ctx *ctx, const char **names, float *ms, int cap);

/* Export of the synthetic reference / index for checking against the oracle (small sizes). */
int cmgpu_export_reference(cmgpu_ctx *ctx, uint32_t seq, char *out, uint32_t capacity);
int cmgpu_reference_lengths(cmgpu_ctx *ctx, uint32_t *lengths, uint32_t capacity, uint32_t *n_sequences);


/* A per-read (2 n entries: "rlen", "mm_cnt", "hit_tot", "ncp", "ncn", "resc_p", "resc_n", "mcp", "mcn", "fcp", "fcn", "ndp",
 * "ndn", "nv") or per-pair (n entries: "pe_nbest") 32-bit array of the last mapped batch -- the list lengths the stages saw
 * (tools/list_hist.py).  *n_out = entries. */
int cmgpu_debug_array(cmgpu_ctx *ctx, const char *name, uint32_t *out, uint64_t capacity, uint64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* CHROMAP_AMD_DEBUG_H_ */
