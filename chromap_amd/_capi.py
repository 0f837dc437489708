"""ctypes declarations of include/chromap_amd.h and loader of the in-tree HIP library.

The library is libchromap_amd.so next to this file (built by `make -C chromap_amd/csrc`
or __graft_entry__.build()).  There is no fallback: a missing library is an ImportError-like
failure at load time, a missing GPU makes cmgpu_create fail with CMGPU_ENODEVICE.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libchromap_amd.so")


class IndexView(C.Structure):
    _fields_ = [("kmer_size", C.c_int32), ("window_size", C.c_int32), ("n_buckets", C.c_uint32),
                ("flags", C.POINTER(C.c_uint32)), ("keys", C.POINTER(C.c_uint64)), ("vals", C.POINTER(C.c_uint64)),
                ("n_occurrences", C.c_uint32), ("occurrences", C.POINTER(C.c_uint64))]


class RefView(C.Structure):
    _fields_ = [("n_sequences", C.c_uint32), ("names", C.POINTER(C.c_char_p)),
                ("sequences", C.POINTER(C.c_void_p)), ("lengths", C.POINTER(C.c_uint32))]


PARAM_FIELDS = ("error_threshold", "min_num_seeds", "max_seed_frequency0", "max_seed_frequency1", "max_insert_size",
                "min_read_length", "max_num_best_mappings", "drop_repetitive_reads", "trim_adapters",
                "split_alignment", "mapq_threshold", "remove_pcr_duplicates", "tn5_shift", "low_memory_mode",
                "read_batch_size", "taskloop_grain_size", "bc_error_threshold", "output_mappings_not_in_whitelist",
                "output_format", "dedup_at_bulk_level", "bc_probability_threshold")


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in PARAM_FIELDS[:-1]] + [("bc_probability_threshold", C.c_double)]


class Batch(C.Structure):
    _fields_ = [("n_pairs", C.c_uint32), ("first_read_id", C.c_uint32), ("read1_bases", C.c_void_p),
                ("read1_offsets", C.c_void_p), ("read2_bases", C.c_void_p), ("read2_offsets", C.c_void_p)]


class SingleBatch(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("first_read_id", C.c_uint32), ("bases", C.c_void_p), ("offsets", C.c_void_p)]


class Record(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("rid", C.c_uint32), ("fragment_start", C.c_uint32),
                ("fragment_length", C.c_uint16), ("mapq", C.c_uint8), ("direction", C.c_uint8),
                ("is_unique", C.c_uint8), ("num_dups", C.c_uint8), ("positive_alignment_length", C.c_uint16),
                ("negative_alignment_length", C.c_uint16)]


class PairsRecord(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("rid1", C.c_uint32), ("rid2", C.c_uint32), ("pos1", C.c_uint32),
                ("pos2", C.c_uint32), ("strand1", C.c_uint8), ("strand2", C.c_uint8), ("mapq", C.c_uint8),
                ("is_unique", C.c_uint8)]


STAT_FIELDS = ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads", "num_minimizers",
               "probe_steps", "occurrences_read", "num_pairs_rescued", "num_multi_mappers", "num_barcode_in_whitelist",
               "num_corrected_barcode")


class RecordBc(C.Structure):
    _fields_ = [("r", Record), ("barcode", C.c_uint64)]


class BarcodeBatch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("qualities", C.c_void_p), ("offsets", C.c_void_p)]


class SamRecord(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("rid", C.c_uint32), ("pos", C.c_uint32), ("mpos", C.c_uint32), ("mrid", C.c_int32),
                ("tlen", C.c_int32), ("nm", C.c_uint32), ("flag", C.c_uint16), ("n_cigar", C.c_uint16), ("md_len", C.c_uint16),
                ("mapq", C.c_uint8), ("strand", C.c_uint8), ("is_unique", C.c_uint8), ("valid", C.c_uint8),
                ("length_after_trim", C.c_uint16)]


SAM_CIGAR_CAP = 64
FORMAT_SAM = 1


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS] + [("reserved", C.c_uint64 * 5)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in STAT_FIELDS}


assert C.sizeof(Record) == 24 == C.sizeof(PairsRecord)
assert C.sizeof(RecordBc) == 32

# every symbol include/chromap_amd.h (the boundary) and include/chromap_amd_debug.h (measurement / test hooks) declare
SYMBOLS = ("cmgpu_default_params", "cmgpu_apply_preset", "cmgpu_create", "cmgpu_create_synthetic", "cmgpu_create_from_reference",
           "cmgpu_save_index_file", "cmgpu_create_shared", "cmgpu_set_chr_order", "cmgpu_set_pairs_chr_order", "cmgpu_write_pairs_ranked", "cmgpu_destroy",
           "cmgpu_last_error", "cmgpu_last_error_thread", "cmgpu_map_pairs", "cmgpu_map_pairs_async", "cmgpu_wait", "cmgpu_upload_batch", "cmgpu_map_resident",
           "cmgpu_download_records", "cmgpu_generate_resident_batch", "cmgpu_download_batch", "cmgpu_probe_bench", "cmgpu_gather_bench",
           "cmgpu_last_timings", "cmgpu_index_info", "cmgpu_export_index", "cmgpu_records_to_device", "cmgpu_records_partition",
           "cmgpu_export_reference", "cmgpu_reference_lengths", "cmgpu_write_bed_pe", "cmgpu_write_pairs", "cmgpu_map_single", "cmgpu_write_bed_se", "cmgpu_load_whitelist_file", "cmgpu_set_whitelist", "cmgpu_store_format_pairs", "cmgpu_write_pairs_header",
           "cmgpu_compute_barcode_abundance", "cmgpu_map_pairs_barcoded", "cmgpu_map_single_barcoded", "cmgpu_write_bed_pe_bc",
           "cmgpu_store_clear", "cmgpu_store_reserve", "cmgpu_store_append_resident", "cmgpu_store_append", "cmgpu_store_format",
           "cmgpu_store_text", "cmgpu_store_write_text", "cmgpu_store_info",
           "cmgpu_sam_layout", "cmgpu_download_sam", "cmgpu_write_sam", "cmgpu_download_barcode_keys", "cmgpu_set_barcode_check", "cmgpu_write_sam_barcoded", "cmgpu_write_sam_barcoded_translated",
           "cmgpu_warm_up", "cmgpu_fastq_set_format", "cmgpu_fastq_scan", "cmgpu_fastq_scan_bgzf", "cmgpu_fastq_take", "cmgpu_fastq_commit", "cmgpu_barcode_abundance_resident",
           "cmgpu_load_index_file", "cmgpu_free_host_index", "cmgpu_load_reference_fasta", "cmgpu_free_host_ref",
           "cmgpu_create_synthetic_repeats", "cmgpu_create_synthetic_profile", "cmgpu_generate_resident_batch_indels", "cmgpu_generate_resident_batch_hic", "cmgpu_probe_bench_variant", "cmgpu_gather_sweep", "cmgpu_set_option", "cmgpu_get_option", "cmgpu_swap_resident_batch",
           "cmgpu_exchange_unique_id", "cmgpu_exchange_init", "cmgpu_exchange_init_all", "cmgpu_exchange_init_external",
           "cmgpu_exchange_owner_table", "cmgpu_exchange_step", "cmgpu_exchange_info", "cmgpu_exchange_plan", "cmgpu_exchange_finalize", "cmgpu_memcpy", "cmgpu_copy_whitelist",
           "cmgpu_host_alloc", "cmgpu_host_free", "cmgpu_host_register", "cmgpu_host_unregister", "cmgpu_submit_pairs", "cmgpu_map_submitted", "cmgpu_map_submitted_async", "cmgpu_records_wait",
           "cmgpu_debug_trace", "cmgpu_debug_minimizers", "cmgpu_debug_minimizers_all", "cmgpu_debug_array")

UNIQUE_ID_BYTES = 128
ALLGATHER_COUNTS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32)


class ExchangeTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather_counts", ALLGATHER_COUNTS_FN), ("alltoallv", ALLTOALLV_FN)]



TEXT_BED_PE, TEXT_BED_SE, TEXT_BED_PE_BC, TEXT_TAGALIGN_PE, TEXT_TAGALIGN_PE_BC, TEXT_BED_SE_BC, TEXT_TAGALIGN_SE_BC = 0, 1, 2, 3, 4, 5, 6

_LIB = None


def declare(L):
    """attach argtypes/restypes (shared with tests/hostemu, which exports a subset)"""
    def sig(name, res, args):
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
    P = C.POINTER
    sig("cmgpu_default_params", None, [P(Params)])
    sig("cmgpu_apply_preset", C.c_int, [P(Params), C.c_char_p])
    sig("cmgpu_create", C.c_int, [P(IndexView), P(RefView), P(Params), C.c_int, P(C.c_void_p)])
    sig("cmgpu_create_synthetic", C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, P(Params),
                                            C.c_int, P(C.c_void_p)])
    sig("cmgpu_create_from_reference", C.c_int, [P(RefView), C.c_int32, C.c_int32, P(Params), C.c_int, P(C.c_void_p)])
    sig("cmgpu_save_index_file", C.c_int, [C.c_void_p, C.c_char_p])
    sig("cmgpu_set_chr_order", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32])
    sig("cmgpu_set_pairs_chr_order", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32])
    sig("cmgpu_write_pairs_ranked", C.c_int64, [P(C.c_char_p), P(C.c_uint32), C.c_uint32, P(Params), C.c_void_p, C.c_uint64,
                                                P(C.c_char_p), C.c_uint32, C.c_void_p, C.c_char_p])
    sig("cmgpu_create_shared", C.c_int, [C.c_void_p, P(C.c_void_p)])
    sig("cmgpu_destroy", C.c_int, [C.c_void_p])
    sig("cmgpu_last_error", C.c_char_p, [C.c_void_p])
    sig("cmgpu_last_error_thread", C.c_char_p, [])
    sig("cmgpu_map_pairs", C.c_int, [C.c_void_p, P(Batch), C.c_void_p, C.c_uint64, P(C.c_uint64), P(Stats)])
    sig("cmgpu_map_pairs_async", C.c_int, [C.c_void_p, P(Batch), P(Stats)])
    sig("cmgpu_wait", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64)])
    sig("cmgpu_upload_batch", C.c_int, [C.c_void_p, P(Batch)])
    sig("cmgpu_map_resident", C.c_int, [C.c_void_p, P(C.c_uint64), P(Stats)])
    sig("cmgpu_download_records", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64)])
    sig("cmgpu_generate_resident_batch", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                                   C.c_double, C.c_uint64])
    sig("cmgpu_download_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
    sig("cmgpu_probe_bench", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, P(C.c_double), P(C.c_uint64),
                                       P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_gather_bench", C.c_int, [C.c_void_p, C.c_uint64, C.c_int, P(C.c_double)])
    sig("cmgpu_last_timings", C.c_int, [C.c_void_p, P(C.c_char_p), P(C.c_float), C.c_int])
    sig("cmgpu_index_info", C.c_int, [C.c_void_p, P(C.c_int32), P(C.c_int32), P(C.c_uint32), P(C.c_uint32),
                                      P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_export_index", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p])
    sig("cmgpu_records_to_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64)])
    sig("cmgpu_export_reference", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32])
    sig("cmgpu_reference_lengths", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, P(C.c_uint32)])
    sig("cmgpu_write_bed_pe", C.c_int64, [P(C.c_char_p), C.c_uint32, P(Params), C.c_void_p, C.c_uint64, C.c_char_p])
    sig("cmgpu_write_pairs", C.c_int64, [P(C.c_char_p), P(C.c_uint32), C.c_uint32, P(Params), C.c_void_p, C.c_uint64,
                                         P(C.c_char_p), C.c_uint32, C.c_char_p])
    sig("cmgpu_map_single", C.c_int, [C.c_void_p, P(SingleBatch), C.c_void_p, C.c_uint64, P(C.c_uint64), P(Stats)])
    sig("cmgpu_records_partition", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, P(C.c_uint64)])
    sig("cmgpu_store_clear", C.c_int, [C.c_void_p])
    sig("cmgpu_store_reserve", C.c_int, [C.c_void_p, C.c_uint64, C.c_int])
    sig("cmgpu_store_append_resident", C.c_int, [C.c_void_p, P(C.c_uint64)])
    sig("cmgpu_store_append", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int])
    sig("cmgpu_store_format", C.c_int, [C.c_void_p, C.c_int, P(C.c_char_p), C.c_uint32, P(Params), C.c_uint32,
                                        P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_store_text", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64])
    sig("cmgpu_store_write_text", C.c_int, [C.c_void_p, C.c_char_p, C.c_int])
    sig("cmgpu_store_info", C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_sam_layout", C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_uint32)])
    sig("cmgpu_download_sam", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
    sig("cmgpu_write_sam", C.c_int64, [P(C.c_char_p), C.c_void_p, C.c_uint32, P(Params), C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_uint32, P(C.c_char_p), P(C.c_char_p)] + [C.c_void_p] * 6 + [C.c_char_p])
    sig("cmgpu_download_barcode_keys", C.c_int, [C.c_void_p, C.c_void_p])
    sig("cmgpu_set_barcode_check", C.c_int, [C.c_void_p, C.c_int])
    sig("cmgpu_write_sam_barcoded", C.c_int64, [P(C.c_char_p), C.c_void_p, C.c_uint32, P(Params), C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_uint32, P(C.c_char_p), P(C.c_char_p)] + [C.c_void_p] * 6 +
        [C.c_void_p, C.c_uint32, C.c_char_p])
    sig("cmgpu_write_sam_barcoded_translated", C.c_int64, [P(C.c_char_p), C.c_void_p, C.c_uint32, P(Params), C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                                           C.c_void_p, C.c_uint32, P(C.c_char_p), P(C.c_char_p)] + [C.c_void_p] * 6 +
        [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint64, C.c_char_p])
    sig("cmgpu_fastq_set_format", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_char])
    sig("cmgpu_fastq_scan", C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_uint64, C.c_int, P(C.c_uint32)])
    sig("cmgpu_fastq_scan_bgzf", C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_uint64, C.c_int, P(C.c_uint32)])
    sig("cmgpu_warm_up", C.c_int, [C.c_int])
    sig("cmgpu_fastq_take", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, P(C.c_uint64)])
    sig("cmgpu_barcode_abundance_resident", C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_int)])
    sig("cmgpu_fastq_commit", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int])
    sig("cmgpu_write_bed_se", C.c_int64, [P(C.c_char_p), C.c_uint32, P(Params), C.c_void_p, C.c_uint64, C.c_char_p])
    sig("cmgpu_load_whitelist_file", C.c_int, [C.c_char_p, C.c_uint32, P(C.c_void_p), P(C.c_uint32)])
    sig("cmgpu_set_whitelist", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32])
    sig("cmgpu_store_format_pairs", C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)])
    sig("cmgpu_write_pairs_header", C.c_int, [C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p])
    sig("cmgpu_compute_barcode_abundance", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, P(C.c_uint64)])
    sig("cmgpu_map_single_barcoded", C.c_int, [C.c_void_p, P(SingleBatch), P(BarcodeBatch), C.c_void_p, C.c_uint64, P(C.c_uint64), P(Stats)])
    sig("cmgpu_map_pairs_barcoded", C.c_int, [C.c_void_p, P(Batch), P(BarcodeBatch), C.c_void_p, C.c_uint64, P(C.c_uint64),
                                               P(Stats)])
    sig("cmgpu_write_bed_pe_bc", C.c_int64, [P(C.c_char_p), C.c_uint32, P(Params), C.c_void_p, C.c_uint64, C.c_uint32,
                                             C.c_char_p])
    sig("cmgpu_probe_bench_variant", C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, P(C.c_double), P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_create_synthetic_profile", C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, P(Params), C.c_int, C.c_uint32,
                                                    P(C.c_void_p)])
    sig("cmgpu_create_synthetic_repeats", C.c_int, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int32, C.c_int32, P(Params), C.c_int,
                                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, P(C.c_void_p)])
    sig("cmgpu_generate_resident_batch_hic", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint64])
    sig("cmgpu_generate_resident_batch_indels", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                                          C.c_double, C.c_uint64])
    sig("cmgpu_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64])
    sig("cmgpu_get_option", C.c_int, [C.c_void_p, C.c_char_p, P(C.c_int64)])
    sig("cmgpu_gather_sweep", C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, P(C.c_double)])
    sig("cmgpu_swap_resident_batch", C.c_int, [C.c_void_p, C.c_int])
    sig("cmgpu_exchange_unique_id", C.c_int, [C.c_void_p])
    sig("cmgpu_exchange_init", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int])
    sig("cmgpu_exchange_init_all", C.c_int, [P(C.c_void_p), C.c_int])
    sig("cmgpu_exchange_init_external", C.c_int, [C.c_void_p, P(ExchangeTransport), C.c_int, C.c_int])
    sig("cmgpu_exchange_owner_table", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32])
    sig("cmgpu_exchange_step", C.c_int, [C.c_void_p, P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_exchange_info", C.c_int, [C.c_void_p, P(C.c_int), P(C.c_int), P(C.c_uint64), P(C.c_uint64)])
    sig("cmgpu_exchange_plan", C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, P(C.c_uint32)])
    sig("cmgpu_exchange_finalize", C.c_int, [C.c_void_p])
    sig("cmgpu_host_alloc", C.c_void_p, [C.c_uint64])
    sig("cmgpu_host_free", None, [C.c_void_p])
    sig("cmgpu_host_register", C.c_int, [C.c_void_p, C.c_uint64])
    sig("cmgpu_host_unregister", C.c_int, [C.c_void_p])
    sig("cmgpu_submit_pairs", C.c_int, [C.c_void_p, P(Batch)])
    sig("cmgpu_map_submitted", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64), P(Stats)])
    sig("cmgpu_map_submitted_async", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, P(Stats)])
    sig("cmgpu_records_wait", C.c_int, [C.c_void_p, P(C.c_uint64)])
    sig("cmgpu_debug_trace", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64])
    sig("cmgpu_debug_array", C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)])
    sig("cmgpu_debug_minimizers", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, P(C.c_uint32)])
    sig("cmgpu_debug_minimizers_all", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, P(C.c_uint64)])
    sig("cmgpu_copy_whitelist", C.c_int, [C.c_void_p, C.c_void_p])
    sig("cmgpu_memcpy", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int])
    sig("cmgpu_load_index_file", C.c_int, [C.c_char_p, P(IndexView)])
    sig("cmgpu_free_host_index", None, [P(IndexView)])
    sig("cmgpu_load_reference_fasta", C.c_int, [C.c_char_p, P(RefView)])
    sig("cmgpu_free_host_ref", None, [P(RefView)])
    return L


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C chromap_amd/csrc` (hipcc, gfx950). "
                               "chromap_amd has no CPU fallback." % LIB_PATH)
        _LIB = declare(C.CDLL(LIB_PATH))
    return _LIB


def default_params(preset=None, **overrides):
    L = lib()
    p = Params()
    L.cmgpu_default_params(C.byref(p))
    if preset:
        if L.cmgpu_apply_preset(C.byref(p), preset.encode()) != 0:
            raise ValueError("unknown preset %r" % preset)
    for k, v in overrides.items():
        if k not in PARAM_FIELDS:
            raise KeyError(k)
        setattr(p, k, v)
    return p
