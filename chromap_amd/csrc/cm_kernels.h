// cm_kernels.h -- launch helpers implemented in cm_kernels.hip
#ifndef CM_KERNELS_H_
#define CM_KERNELS_H_
#include <hip/hip_runtime.h>

#include "cm_types.h"

#define CM_SLAB_CAP 65535u   // entries of a slab (the cooperative sweep's offsets are 16-bit)
#define CM_SLAB_BLOCKS 128u  // blocks of a launch that works on slabs
#define CM_DECL_LAUNCH(kname) void cm_launch_##kname(const CmDev &d, uint32_t n, hipStream_t s);
void cm_launch_k_prep_count(const CmDev &d, uint32_t n_pairs, uint32_t max_read_len, hipStream_t s);
void cm_launch_k_mm_fill(const CmDev &d, uint32_t pair_lo, uint32_t pair_hi, uint32_t max_read_len, hipStream_t s);
void cm_launch_k_mm_marks(const uint32_t *mm_off, const uint32_t *lo, uint32_t n_marks, unsigned long long *marks, hipStream_t s);
bool cm_prep_mm_supported(const CmDev &d, uint32_t max_read_len);
bool cm_prep_flat_supported(const CmDev &d, uint32_t max_read_len, uint32_t tile_reads);
void cm_launch_k_prep_flat(const CmDev &d, uint32_t pair_lo, uint32_t pair_hi, uint32_t max_read_len, uint32_t tile_reads, uint32_t mm_cap,
                           unsigned long long *cursor, hipStream_t s);
uint32_t cm_prep_mm_pairs_per_block(const CmDev &d, uint32_t max_read_len);
size_t cm_prep_mm_stage_bytes(const CmDev &d, uint32_t max_read_len, uint32_t pairs);
void cm_launch_k_prep_mm(const CmDev &d, uint32_t pair_lo, uint32_t pair_hi, uint32_t max_read_len, uint32_t mm_cap,
                         unsigned long long *cursor, hipStream_t s, void *gstage);
uint32_t cm_probe_range_blocks(uint64_t max_entries, int variant);
void cm_launch_k_probe_range(const CmDev &d, const unsigned long long *range, uint64_t max_entries, uint32_t cap, void *partials, hipStream_t s, int variant);
void cm_launch_k_probe_reduce(const void *partials, uint32_t blocks, unsigned long long *counters, hipStream_t s);
CM_DECL_LAUNCH(k_s3a_count)
void cm_launch_k_sort_lists(const CmDev &d, int mode, hipStream_t s);
uint32_t cm_s3b_lane_cap(uint32_t max_read_len);
void cm_s3b_heavy_classes(uint32_t *hv_max, uint32_t *hv_big);
void cm_launch_k_s3b_heavy(const CmDev &d, const uint32_t *n_cls, hipStream_t s, bool coop, uint32_t max_read_len);
void cm_launch_k_s3b_candidates(const CmDev &d, uint32_t n, uint32_t max_read_len, hipStream_t s);
void cm_launch_k_s4a_rescue_count(const CmDev &d, uint32_t n, hipStream_t s, bool coop);
void cm_launch_k_s4b_rescue_merge(const CmDev &d, uint32_t n, hipStream_t s, bool coop, uint32_t max_read_len);
uint32_t cm_rescue_seg_cap(uint32_t n_reads);
void cm_launch_k_s4a_rescue_list(const CmDev &d, uint32_t n_reads, hipStream_t s, bool coop);
void cm_launch_k_s4b_rescue_list(const CmDev &d, uint32_t n_reads, hipStream_t s, bool coop, uint32_t max_read_len);
void cm_launch_k_s4c_reduce(const CmDev &d, uint32_t n, hipStream_t s, uint32_t coop);
void cm_launch_k_s5a_prepare(const CmDev &d, uint32_t n, hipStream_t s, bool coop);
void cm_launch_k_s5c_finalize(const CmDev &d, uint32_t n, hipStream_t s, bool coop);
void cm_launch_k_s5b_verify(const CmDev &d, uint32_t max_items, uint32_t n_reads, hipStream_t s);
void cm_launch_k_s6a_pair(const CmDev &d, uint32_t n, hipStream_t s, bool coop);
void cm_launch_k_s6c_multi(const CmDev &d, uint32_t n, hipStream_t s, bool coop);
CM_DECL_LAUNCH(k_s6a_pair_sam)
CM_DECL_LAUNCH(k_s6c_multi_sam)
void cm_launch_k_s6b_sample(const CmDev &d, uint32_t n_chunks, hipStream_t s);
void cm_launch_k_s0b_barcode(const CmDev &d, uint32_t n, hipStream_t s);
void cm_launch_k_bc_abundance(const uint8_t *bcb, const uint32_t *bco, uint32_t lo, uint32_t hi, uint64_t *wl, uint32_t wl_mask,
                              unsigned long long *num_sample, hipStream_t s);
size_t cm_probe_partial_words(uint32_t n);
void cm_launch_k_probe(const uint64_t *bkt, uint32_t bmask, const uint64_t *hash, uint64_t *val, uint8_t *kind,
                       uint32_t n, void *partials, unsigned long long *counters, hipStream_t s, int variant = 0);
size_t cm_stats_partial_words(uint32_t n);
void cm_launch_k_stats(const CmDev &d, uint32_t n, unsigned long long *partials, hipStream_t s);

void cm_build_heavy_last(const CmDev &d, uint32_t n_pairs, uint32_t *fr, uint32_t *sr, uint32_t *fp, uint32_t *sp, uint32_t *perm_reads,
                         uint32_t *perm_pairs, uint32_t *scan_tmp, hipStream_t s);
void cm_launch_k_pack_ref(const uint8_t *ref, uint64_t n_bytes, CmPlRec *pl, hipStream_t s);
void cm_launch_k_pack_reads(const CmDev &d, uint32_t n_reads, hipStream_t s);
void cm_launch_k_check_cap(const unsigned long long *total, unsigned long long cap, unsigned long long *flag, hipStream_t s);
void cm_launch_k_copy_u64(const unsigned long long *src, unsigned long long *dst, hipStream_t s);
void cm_launch_k_sum_u32(const uint32_t *in, uint32_t n, unsigned long long *out, hipStream_t s);
size_t cm_scan_tmp_words(uint32_t n);
// out[0..n] = exclusive prefix sums of in[0..n), out[n] = total
void cm_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t *tmp, hipStream_t s);

#endif
