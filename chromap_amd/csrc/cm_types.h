// cm_types.h -- plain structs shared by host code, HIP kernels and the host-side
// emulation harness under tests/hostemu (which compiles cm_stages.h with g++ to check the
// per-item stage logic on a machine without a GPU; it is test infrastructure, the
// library itself has no CPU path).
#ifndef CM_TYPES_H_
#define CM_TYPES_H_

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CM_HD __host__ __device__ __forceinline__
#define CM_D __device__ __forceinline__
#else
#define CM_HD inline
#define CM_D inline
#endif

#define CM_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull   // bucket never written (khash "empty" flag)
#define CM_DELETED_KEY 0xFFFFFFFFFFFFFFFEull // khash "deleted" flag (not produced by Index::Construct)

// probe result kinds
#define CM_PR_MISS 0
#define CM_PR_SINGLE 1
#define CM_PR_MULTI 2
#define CM_V_INVALID 0x7fff
#define CM_MM_CHUNKS 8            // chunks of a batch whose index probe overlaps the next chunk's minimizer pass
// Device work lists (CmDev::hv_list + id * hv_stride, their lengths at hv_cnt[id]): the reads / pairs whose lists are too long for the
// one-lane-per-item kernel of a stage, by stage and size class.  P: entries of the class's shared work area; LDS: bytes per block.
//   id  name                  items                                  kernel (lanes per item)             P / LDS per block
//    0  CM_L_HIT_WAVE         reads, hits <= hv_max[0] (512)         k_s3b_coop<64> (two reads a block)   512 / 2 x 6 KB (32-bit keys; 10 KB with 64-bit keys)
//   21  CM_L_HIT_WAVE_SMALL   reads, hits <= hv_sub (256)            k_s3b_coop<64> (four reads a block)  256 / 4 x 3 KB
//    4  CM_L_HIT_G16          reads, hits <= hv_mid (64)             k_s3b_heavy<16> (16 lanes, bitonic)  64 / 10 B per slot
//    1  CM_L_HIT_B256A        reads, hits <= hv_max[1] (1024)        k_s3b_coop<256>                      1024 / 12 KB
//    2  CM_L_HIT_B256B        reads, hits <= hv_max[2] (2048)        k_s3b_coop<256>                      2048 / 23 KB
//   10  CM_L_HIT_B512         reads, hits <= hv_max[3] (4096)        k_s3b_coop<512>                      4096 / 46 KB
//   25  CM_L_HIT_B1024        reads, hits <= hv_big (7040; 8192 with 64-bit keys)  k_s3b_coop<1024>          7040 / 79 KB: two blocks per CU
//    3  CM_L_HIT_SLAB         reads with more hits                   k_s3b_coop<1024, false>, use_slab    global slab (19 B per hit)
//    5  CM_L_HIT_DECLINED     reads k_s3b_coop declined              k_s3b_heavy<CM_BLOCK> (bitonic)      pow2(hv_big) x 10 B
//   16  CM_L_HIT_SERIAL       reads the slab launch declined         k_s3b_serial (a lane)                --
//   23  CM_L_SEARCH_WAVE      reads whose mate has >= 24 candidates  k_s4a/4b_rescue_wave<true> (64)      64 windows / 4.9 KB
//   31  CM_L_SEARCH_WAVE_BIG  ... with more than 64 best candidates  k_s4a/4b_rescue_wave<false> (64)     304 windows / 11 KB
//    6  CM_L_RS_WAVE          reads, rescue hits <= hv_max[0]        k_s4b_coop<64> (two reads a block)   512 / 2 x 11 KB
//  7/8  CM_L_RS_B256A / B     reads, rescue hits <= hv_max[1] / [2]  k_s4b_coop<256>                      1024 / 21 KB, 2048 / 41 KB
//   11  CM_L_RS_B512          reads, rescue hits <= rs_max3 (3968)   k_s4b_coop<512>                      3968 / 79 KB
//   26  CM_L_RS_B1024         reads, rescue hits <= rs_big (7680)    k_s4b_coop<1024>                     7680 / 153 KB
//   15  CM_L_RS_SLAB          reads with more rescue hits            k_s4b_coop<1024>, use_slab           global slab
//   27  CM_L_PF_WAVE          pairs, candidate lists <= 256          k_s4c_coop<64, true>                 CM_S4C_P_SMALL
//    9  CM_L_PF_BLOCK         pairs, lists <= 1024                   k_s4c_coop<256, true>                CM_S4C_P_WAVE
//   14  CM_L_PF_BLOCK_BIG     pairs, lists <= 4096                   k_s4c_coop<256, true>                CM_S4C_P_BLOCK
//   19  CM_L_PF_HUGE          pairs, lists <= s4c_pbig (15360)       k_s4c_coop<1024, false>              lists stay in global memory, 10 B per entry
//   28  CM_L_S5_SMALL         reads, candidates <= CM_S5C_P_SMALL    k_s5_sort_coop, k_s5c_coop<64>       a quarter of the wave class's arrays
//   12  CM_L_S5_WAVE          reads, candidates <= CM_S5C_P_WAVE     k_s5_sort_coop, k_s5c_coop<64>       CM_S5C_P_WAVE
//   22  CM_L_S5_BLOCK         reads with more candidates             k_s5_sort_coop, k_s5c_coop<256>      16384
//   29  CM_L_S6A_SMALL        pairs, draft mappings <= 256           k_s6a_coop<64>                       CM_S6A_P_SMALL
//   13  CM_L_S6A_WAVE         pairs, draft mappings <= 1024          k_s6a_coop<64>                       CM_S6A_P_WAVE
//   18  CM_L_S6A_BLOCK        pairs with more                        k_s6a_coop<256>                      CM_S6A_P_BLOCK
//   30 / 17 / 20  CM_L_S6C_SMALL / WAVE / BLOCK   multi-mapped pairs, the same classes   k_s6c_coop<64> / <64> / <256>
// (24: free.)  A class's kernels are launched when the class had items lately (CmDev::cls_mask, cm_kernels.hip).
enum CmList : uint32_t {
  CM_L_HIT_WAVE = 0, CM_L_HIT_B256A = 1, CM_L_HIT_B256B = 2, CM_L_HIT_SLAB = 3, CM_L_HIT_G16 = 4, CM_L_HIT_DECLINED = 5, CM_L_RS_WAVE = 6, CM_L_RS_B256A = 7,
  CM_L_RS_B256B = 8, CM_L_PF_BLOCK = 9, CM_L_HIT_B512 = 10, CM_L_RS_B512 = 11, CM_L_S5_WAVE = 12, CM_L_S6A_WAVE = 13, CM_L_PF_BLOCK_BIG = 14, CM_L_RS_SLAB = 15,
  CM_L_HIT_SERIAL = 16, CM_L_S6C_WAVE = 17, CM_L_S6A_BLOCK = 18, CM_L_PF_HUGE = 19, CM_L_S6C_BLOCK = 20, CM_L_HIT_WAVE_SMALL = 21, CM_L_S5_BLOCK = 22,
  CM_L_SEARCH_WAVE = 23, CM_L_HIT_B1024 = 25, CM_L_RS_B1024 = 26, CM_L_PF_WAVE = 27, CM_L_S5_SMALL = 28, CM_L_S6A_SMALL = 29, CM_L_S6C_SMALL = 30,
  CM_L_SEARCH_WAVE_BIG = 31
};
#define CM_HV_LISTS 32
#define CM_RS_SEGS 64           // rescue list segments (one counter each, on its own cache line)
#define CM_MAX_BEST 8192        // upper bound on max_num_best_mappings (-n): a 500 000-pair batch then has 2^32 record slots, which are addressed with 32 bits
#define CM_SORT_SERIAL_MAX 24   // cm_sort_cand / cm_sort_draft: insertion sort up to here, heap sort beyond
#define CM_SORT_WAVE_MAX 1024   // longest list a wave sorts in LDS (k_sort_lists)

// Subset of MappingParameters used on the device.
#define CM_SAM_CIGAR_CAP 64
#define CM_GOFF_GAP 2048u       // > CM_MAX_READ_LEN + the largest error threshold: no hit of one sequence lies within e of another's
#define CM_MAX_READ_LEN 1400   // 16 pairs of reads per block must fit the LDS staging area (2 x 24 KB)
struct CmParams {
  int32_t e;             // error_threshold
  int32_t min_seeds;     // min_num_seeds_required_for_mapping
  int32_t f0, f1;        // max_seed_frequencies
  int32_t max_insert;    // max_insert_size
  int32_t min_read_len;  // min_read_length
  int32_t max_best;      // max_num_best_mappings: record slots (and reservoir entries) per pair, 1..CM_MAX_BEST
  int32_t drop_rep;      // drop_repetitive_reads
  int32_t trim;          // trim_adapters
  int32_t split;         // split_alignment (--preset hic)
  int32_t single;        // single-end batch: mate 1 of every pair is empty (MapSingleEndReads, chromap.h:385-472)
  int32_t bc_err;        // barcode_correction_error_threshold (0, 1 or 2)
  int32_t bc_keep;       // output_mappings_not_in_whitelist
  double bc_prob;        // barcode_correction_probability_threshold
  int32_t k, w;          // from the index file
  int32_t lanes;         // GetNumVPULanes(): 8 if e<8, 4 if e<16, else 0
  int32_t ref_batch;     // 500000
  int32_t grain;         // 5000
  int32_t pairs_out;     // mapping_output_format == PAIRS without split alignment: the pairing's record is a cmgpu_pairs_record
  int32_t sam;           // mapping_output_format == SAM: coordinates from ksw_semi_global3, CIGAR / NM / MD kept
};

// MAPQ tables computed on the host with libm so the device reproduces the reference's
// double arithmetic without calling log() (mapping_generator.h:920-1192).
struct CmMapqTables {
  const double *len_coef;     // [65536]: alen < 50 ? 1.0 : 3 / log(alen)
  const uint32_t *nsec_break; // [n_break]: smallest n with (int)(4.343*log(n+1)+0.499) >= v+1... see cm_api
  int32_t n_break;
};

// one record of the reference's interleaved bit planes (CmDev::ref_pl)
struct alignas(16) CmPlRec { uint32_t p0, p1, pn, pc; };
#define CM_PL_LEAD 4

// All device pointers a stage needs.  Per-read arrays are indexed r = 2*pair + mate.
struct CmDev {
  // ---- index in HBM: bucket i = {key, val} at bkt[2i], bkt[2i+1] (16-B aligned)
  const uint64_t *bkt;
  uint32_t bmask;
  const uint64_t *occ;
  uint32_t n_occ;
  // ---- reference in HBM: raw bytes, sequences separated by >= 64 zero bytes
  const uint8_t *ref;
  const uint64_t *ref_off;
  const uint32_t *ref_len;
  uint32_t n_seq;
  // goff[i]: start of sequence i (index order) when the sequences are laid end to end with CM_GOFF_GAP positions between them, n_seq + 1
  // entries -- the 32-bit hit keys of the cooperative hit-list stage (CmKeyOps, cm_stages.h); nullptr: the reference does not fit 32 bits
  const uint32_t *goff;
  // ---- the same bytes as bit planes (cm_pack_planes32, cm_stages.h), interleaved: record w = the four plane words of bases
  //      32 w .. 32 w + 31 of `ref` -- bit i of p0 / p1 = the two bits of base i's CharToUint8 code, pn = none of ACGTacgt,
  //      pc = a lower-case letter -- so that an alignment window (66 .. 100 bases) is ONE run of 48 .. 64 bytes (1-2 sectors)
  //      instead of one run per plane.  CM_PL_LEAD zero records stand in front of record 0 (the backward windows of the split
  //      alignment start up to two words below their first base).  ref_pl_words = records.  nullptr: not built
  const CmPlRec *ref_pl;
  uint64_t ref_pl_words;
  // ---- the batch's reads as bit planes, forward and reverse complement (cm_pack_read_planes): read r, orientation o, plane q at
  //      read_pl + r * cm_read_pl_stride(read_pl_w) + (o * 3 + q) * read_pl_w, read_pl_w words of 32 bases each.  nullptr: not packed
  uint32_t *read_pl;
  uint32_t read_pl_w;
  CmParams p;
  CmMapqTables mq;
  // ---- batch
  uint32_t n_pairs;
  uint32_t first_read_id;
  const uint8_t *rb0, *rb1;    // bases of mate 0 / mate 1
  const uint32_t *ro0, *ro1;   // offsets (n_pairs+1)
  // ---- single-cell barcodes (nullptr for bulk data)
  const uint8_t *bcb, *bcq;    // barcode bases / qualities
  const uint32_t *bco;         // offsets (n_pairs+1)
  const uint64_t *wl;          // whitelist table: bucket i = {key, count} at wl[2i], wl[2i+1]; empty key = all ones
  uint32_t wl_mask;
  double wl_num_sample;        // num_sample_barcodes_ as double
  const double *pow10_tab;     // [81]: pow(10, -q/10)
  uint64_t *bc_key;            // [n] (corrected) barcode key
  uint8_t *bc_ok;              // [n] CorrectBarcodeAt's return value
  // ---- per read
  uint32_t *rlen;       // length after trimming; 0 when the pair was dropped
  uint32_t *mm_cnt;     // [2n]
  uint32_t *mm_off;     // [2n+1] dense prefix
  uint64_t *mm_hash;    // dense
  uint32_t *mm_ps;
  uint64_t *pr_val;     // dense probe result value
  uint8_t *pr_kind;
  // ---- hits / first candidates
  uint32_t *hit_tot;    // [2n] total hits of the round used
  uint32_t *hit_off;    // [2n+1]
  uint8_t *round2;      // [2n] 1 when the high-frequency round was used
  uint32_t *rep_cnt;    // [2n]
  uint32_t *rep_len;    // [2n]
  uint64_t *hbuf;       // hits, then candidates in place
  uint8_t *hcnt;        // candidate counts (same indexing as hbuf)
  uint32_t *n_pos_hit;  // [2n] boundary P between + and - sub-lists
  uint32_t *ncp, *ncn;  // [2n] candidates after GenerateCandidates
  // ---- reads whose hit list does not fit a lane's LDS slots, by size class (k_s3a_count fills, k_s3b_heavy consumes):
  //      hv_cnt[c] reads of class c listed at hv_list + c * hv_stride; class 0: <= hv_max[0] hits (a wave each),
  //      1: <= hv_max[1] (a block each), 2: <= hv_max[2] (a block, large LDS), 3: longer (one lane, in global memory),
  //      4: <= hv_mid (a group of 16 lanes each, four reads per wave)
  const uint32_t *perm_reads, *perm_pairs;  // heavy-last processing order of reads / pairs, or nullptr (identity)
  uint32_t *hv_cnt, *hv_list;
  // reads that supplement their candidates from the mate (S4a/S4b), packed in CM_RS_SEGS list segments:
  // rs_cnt[16 * g] reads at rs_list + g * cm_rescue_seg_cap(n_reads)
  uint32_t *rs_cnt, *rs_list;
  // lists longer than CM_SORT_SERIAL_MAX (candidates after the pair filter, draft mappings) that a wave sorts before the per-read
  // stage looks at them: srt_cnt[0] items (read << 1 | strand) at srt_list, srt_cnt[1] the work cursor
  uint32_t *srt_cnt, *srt_list;
  // hv_big: hit lists of hv_max[3] < hits <= hv_big go to list 25 (a block of 1024 lanes with the largest work area); rs_max3 / rs_big:
  // the rescue lists' two largest classes (lists 11 / 26: their entries take 20 bytes of the work area, not 19) -- 0: no such class
  uint32_t hv_big, rs_max3, rs_big;
  unsigned long long cls_mask;  // bit l: the kernels of long-list class (list) l are launched for this range (speculative launch set, cm_kernels.hip)
  uint32_t s4c_pbig;  // entries of a candidate list the pair filter's largest class takes (k_s4c_coop<1024, false>; 0: no such class)
  uint32_t hv_stride, s3b_cap, hv_max[4];  // hv_max: hit-list size classes -- a wave, a block of 256 / 512 / 1024 lanes (lists 0, 1, 2, 10)
  uint8_t *coop_slab;      // global work memory of the groups that take lists longer than their shared memory holds:
  uint32_t coop_slab_cap;  // coop_slab_blocks slabs of cm_coop_slab_bytes(coop_slab_cap) bytes, one per block of those launches
  uint32_t coop_slab_blocks;
  const unsigned long long *abort;  // nonzero: the candidate arrays were sized from the previous batch and this batch needs more --
                                    // every stage from S4b on leaves at once, the host maps the range again with exact sizes
  // ---- the rescue hits a wave found while it COUNTED them (k_s4a_rescue_list), kept for the fill pass: a bump-allocated pool.
  //      rs_pool_off[2 r + strand]: where read r's hits of that direction start (0xffffffff: not kept -- no room, or a search whose
  //      tables did not fit one round: k_s4b_rescue_list then searches again).  nullptr: no pool
  uint64_t *rs_pool;
  uint32_t rs_pool_cap;
  uint32_t *rs_pool_off;
  uint32_t wq_dynamic;       // cmgpu_set_option "coop" bit 16: the wave-per-item list kernels take chunks of their lists from a cursor instead of striding (CmWaveQueue)
  unsigned long long *prof;  // measurement aid (cmgpu_set_option "coop_profile"): shader-clock cycles per phase of k_s3b_coop, summed over groups
  uint32_t mm_cap;  // capacity of the dense minimizer arrays (0: not checked): S3a leaves a read whose range passes it idle
  uint32_t coop_rb; // tests: run-table size of the cooperative sorters (0: two per minimizer of the longest read)
  uint32_t hv_sub;  // class 21: lists of hv_mid < hits <= hv_sub go to waves with a quarter of the wave class's work area (0: no such class)
  uint32_t hv_mid;  // class 4: lists of s3b_cap < hits <= hv_mid go to groups of 16 lanes (k_s3b_heavy<16>); 0 = no such class
  // ---- rescue / merged candidates
  uint8_t *aug;         // [2n] augment flag
  int32_t *res_neg;     // [2n] result of the search on the - strand (driven by mate + candidates)
  int32_t *res_pos;     // [2n]
  uint32_t *resc_n;     // [2n] rescue hits on - strand
  uint32_t *resc_p;     // [2n] rescue hits on + strand
  uint32_t *m_tot;      // [2n] capacity of merged lists (ncp+resc_p + ncn+resc_n)
  uint32_t *m_off;      // [2n+1]
  uint64_t *mbuf;       // merged candidates: + list at m_off[r], - list at m_off[r]+ncp+resc_p
  uint8_t *mcnt;
  uint32_t *mcp, *mcn;  // [2n] merged candidate counts
  uint8_t *force0;      // [n] SupplementCandidates returned 1
  // ---- filtered candidates (same offsets/capacities as merged)
  uint64_t *fbuf;
  uint8_t *fcnt;
  uint32_t *fcp, *fcn;  // [2n]
  uint8_t *alive;       // [n] pair still in play
  // ---- draft mappings (same offsets as filtered candidates)
  uint64_t *dpos;
  int16_t *derr;
  uint32_t *dsplit;     // split-alignment only: (actual_errors<<24 | gap_beginning<<16 | read_mapping_length)
  uint32_t *ndp, *ndn;  // [2n]
  // ---- verification work items (one per candidate of the reads that need banded alignment)
  uint32_t *nv;         // [2n] candidates to verify (0: shortcut / dead / split handled in place)
  uint32_t *v_off;      // [2n+1]
  int16_t *v_err;       // per candidate (candidate offsets): edit distance, e+1 = rejected, CM_V_INVALID = invalid position
  int16_t *v_end;       // per candidate: mapping end position in the window
  int32_t *min_err, *second_err, *n_best, *n_second; // [2n]
  // ---- pair level
  int32_t *pe_min, *pe_second, *pe_nbest, *pe_nsecond; // [n]
  uint32_t *pe_first;   // [n] packed first best: dir<<31 | ... see cm_stages
  uint32_t *pe_i1, *pe_i2; // [n]
  uint32_t *pe_choice;  // [n * max_best] chosen best indices of a multi-mapper, increasing
  // ---- output
  uint8_t *rec;         // n records of 24 bytes (cmgpu_record layout)
  uint8_t *rec_ok;      // [n * max_best]
  // ---- --chr-order: rank of every index rid, or nullptr.  When set, ref_off / ref_len above are the arrays
  //      REORDERED by rank: every stage from verification on works in rank space (cm_s4c_reduce re-ranks).
  const uint32_t *rid_rank;
  const uint32_t *pairs_rank;  // --pairs-natural-chr-order: rank deciding the order of a pair's two ends, or nullptr
  // ---- --SAM (per slot: 2*pair + mate): 40-byte cmgpu_sam_record, CM_SAM_CIGAR_CAP cigar words, sam_md_cap MD bytes;
  //      sam_z: backtrack cells of the pair being aligned, word (row*ZW + q) * n_pairs + pair
  uint8_t *sam_rec;
  uint32_t *sam_cigar;
  uint8_t *sam_md;
  uint32_t *sam_z;
  uint32_t sam_md_cap;
  unsigned long long *stats; // device counters (see CM_ST_*)
};

#define CM_ST_CAND 0
#define CM_ST_MAPPINGS 1
#define CM_ST_MAPPED 2
#define CM_ST_UNIQ 3
#define CM_ST_MINIMIZERS 4
#define CM_ST_PROBE_STEPS 5
#define CM_ST_PROBE_HITS 6
#define CM_ST_OCC 7
#define CM_ST_RESCUED 8
#define CM_ST_MULTI 9
#define CM_ST_ERR 10
#define CM_ST_RECORDS 11
#define CM_ST_BC_INWL 12
#define CM_ST_BC_CORR 13
#define CM_ST_TOTAL 14   // scratch: the 64-bit total next to a 32-bit scan
#define CM_ST_ABORT 15   // CmDev::abort
#define CM_ST_POOL 16   // cursor of the rescue-hit pool (CmDev::rs_pool): entries asked for so far
#define CM_ST_N 17

#endif
