"""ctypes binding of oracle/liboracle.so (test infrastructure only)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class OraIndex(C.Structure):
    _fields_ = [("k", C.c_int), ("w", C.c_int), ("n_keys", C.c_uint32), ("n_buckets", C.c_uint32),
                ("size", C.c_uint32), ("n_occupied", C.c_uint32), ("upper_bound", C.c_uint32),
                ("flags", C.POINTER(C.c_uint32)), ("keys", C.POINTER(C.c_uint64)),
                ("vals", C.POINTER(C.c_uint64)), ("n_occ", C.c_uint32), ("occ", C.POINTER(C.c_uint64))]


class OraRef(C.Structure):
    _fields_ = [("n_seq", C.c_uint32), ("name", C.POINTER(C.c_char_p)), ("seq", C.POINTER(C.c_void_p)),
                ("len", C.POINTER(C.c_uint32))]


class OraParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "error_threshold", "min_num_seeds", "max_seed_freq0", "max_seed_freq1", "max_insert_size",
        "min_read_length", "max_num_best_mappings", "drop_repetitive_reads", "trim_adapters",
        "split_alignment", "mapq_threshold", "remove_pcr_duplicates", "tn5_shift", "low_mem", "bc_error_threshold",
        "output_mappings_not_in_whitelist", "output_format", "dedup_at_bulk_level")] + [("bc_probability_threshold", C.c_double)]


PARAM_NAMES = {n for n, _ in OraParams._fields_}
# the C ABI's names (include/chromap_amd.h) for the same parameters
ORACLE_NAMES = {"max_seed_frequency0": "max_seed_freq0", "max_seed_frequency1": "max_seed_freq1", "low_memory_mode": "low_mem"}


class OraRecord(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("rid", C.c_uint32), ("fragment_start", C.c_uint32),
                ("fragment_length", C.c_uint16), ("mapq", C.c_uint8), ("direction", C.c_uint8),
                ("is_unique", C.c_uint8), ("num_dups", C.c_uint8), ("pos_aln_len", C.c_uint16),
                ("neg_aln_len", C.c_uint16)]


class OraStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads", "probe_steps",
        "occ_reads", "lookups", "num_minimizers", "num_verifications", "num_shortcut", "num_rescue",
        "num_trimmed")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class OraTrace(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("len1", "len2", "n_mm1", "n_mm2", "n_cand1", "n_cand2",
                                          "n_draft1", "n_draft2")] + \
               [(n, C.c_int32) for n in ("min_err1", "min_err2", "nbest1", "nbest2", "second1", "second2",
                                         "nsecond1", "nsecond2")] + \
               [("rep1", C.c_uint32), ("rep2", C.c_uint32)] + \
               [(n, C.c_int32) for n in ("min_sum", "nbest", "second_sum", "nsecond", "force_mapq")]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "chromap_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.ora_hash64.restype = C.c_uint64
    L.ora_hash64.argtypes = [C.c_uint64, C.c_uint64]
    L.ora_minimizers.restype = C.c_int
    L.ora_minimizers.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ora_index_load.argtypes = [C.c_char_p, C.POINTER(OraIndex)]
    L.ora_index_save.argtypes = [C.c_char_p, C.POINTER(OraIndex)]
    L.ora_index_build.argtypes = [C.POINTER(OraRef), C.c_int, C.c_int, C.POINTER(OraIndex)]
    L.ora_index_free.argtypes = [C.POINTER(OraIndex)]
    L.ora_index_from_buckets.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int,
                                         C.POINTER(OraIndex)]
    L.ora_kh_get.restype = C.c_uint32
    L.ora_kh_get.argtypes = [C.POINTER(OraIndex), C.c_uint64, C.POINTER(C.c_uint64)]
    L.ora_ref_load.argtypes = [C.c_char_p, C.POINTER(OraRef)]
    L.ora_ref_free.argtypes = [C.POINTER(OraRef)]
    L.ora_banded_align.restype = C.c_int
    L.ora_banded_align.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.ora_banded_traceback.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.ora_create.restype = C.c_void_p
    L.ora_create.argtypes = [C.POINTER(OraIndex), C.POINTER(OraRef), C.POINTER(OraParams)]
    L.ora_destroy.argtypes = [C.c_void_p]
    L.ora_set_trace.argtypes = [C.c_void_p, C.c_void_p]
    L.ora_default_params.argtypes = [C.POINTER(OraParams)]
    L.ora_preset.argtypes = [C.POINTER(OraParams), C.c_char_p]
    for name in ("ora_map_pairs",):
        f = getattr(L, name)
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p, C.POINTER(OraStats)]
    L.ora_map_pairs_mt.restype = C.c_long
    L.ora_map_pairs_mt.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OraStats)]
    L.ora_write_bed_pe.restype = C.c_long
    L.ora_write_bed_pe.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_char_p]
    L.ora_read_fastx.restype = C.c_long
    L.ora_read_fastx.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    _LIB = L
    return L


def params(preset=None, **kw):
    L = lib()
    p = OraParams()
    L.ora_default_params(C.byref(p))
    if preset:
        L.ora_preset(C.byref(p), preset.encode())
    for k, v in kw.items():
        if k == "chr_order":
            p._chr_order = list(v)  # applied by Oracle.__init__ (ora_set_chr_order)
        elif k == "pairs_order":
            p._pairs_order = list(v)
        else:
            k = ORACLE_NAMES.get(k, k)
            if k not in PARAM_NAMES:  # a ctypes structure would take any attribute name silently
                raise KeyError(k)
            setattr(p, k, v)
    return p


def read_fastx(path):
    """returns (bases: bytes-like numpy u8, offsets: numpy u32[n+1])"""
    import numpy as np
    L = lib()
    b = C.c_void_p()
    o = C.c_void_p()
    n = L.ora_read_fastx(path.encode(), C.byref(b), C.byref(o))
    if n < 0:
        raise IOError(path)
    off = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint32)), shape=(n + 1,)).copy()
    tot = int(off[-1])
    bases = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), shape=(max(tot, 1),)).copy()[:tot]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(b)
    libc.free(o)
    return bases, off


class Oracle:
    def __init__(self, index_path, ref_path, p):
        L = lib()
        self.L = L
        self.idx = OraIndex()
        self.ref = OraRef()
        if L.ora_ref_load(ref_path.encode(), C.byref(self.ref)) != 0:
            raise IOError(ref_path)
        if index_path is None:
            if L.ora_index_build(C.byref(self.ref), 17, 7, C.byref(self.idx)) != 0:
                raise RuntimeError("index build failed")
        elif L.ora_index_load(index_path.encode(), C.byref(self.idx)) != 0:
            raise IOError(index_path)
        self.p = p
        self.ctx = L.ora_create(C.byref(self.idx), C.byref(self.ref), C.byref(p))
        order = getattr(p, "_chr_order", None)
        if order:
            import datasets
            names = [self.ref.name[i] for i in range(self.ref.n_seq)]
            ranks = datasets.chr_order_ranks(order, names)
            arr = (C.c_uint32 * len(ranks))(*ranks)
            L.ora_set_chr_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
            assert L.ora_set_chr_order(self.ctx, arr, len(ranks)) == 0
            L.ora_ctx_ref.restype = C.POINTER(OraRef)
            L.ora_ctx_ref.argtypes = [C.c_void_p]
            self._ref_loaded = self.ref              # owns the sequences
            self.ref = L.ora_ctx_ref(self.ctx).contents  # the reordered view the writers print from
        self.pairs_rank = None
        porder = getattr(p, "_pairs_order", None)
        if porder:
            import datasets
            names = [self.ref.name[i] for i in range(self.ref.n_seq)]
            self.pairs_rank = datasets.chr_order_ranks(porder, names)
            arr = (C.c_uint32 * len(self.pairs_rank))(*self.pairs_rank)
            L.ora_set_pairs_chr_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
            assert L.ora_set_pairs_chr_order(self.ctx, arr, len(self.pairs_rank)) == 0

    def map_pairs(self, b1, o1, b2, o2, first_read_id=0, threads=0, trace=False):
        import numpy as np
        n = len(o1) - 1
        assert len(o2) - 1 == n
        rec = (OraRecord * max(1, n * max(1, self.p.max_num_best_mappings)))()
        st = OraStats()
        tr = None
        if trace:
            tr = (OraTrace * max(1, n))()
            self.L.ora_set_trace(self.ctx, C.cast(tr, C.c_void_p))
        b1 = np.ascontiguousarray(b1)
        b2 = np.ascontiguousarray(b2)
        o1 = np.ascontiguousarray(o1, dtype=np.uint32)
        o2 = np.ascontiguousarray(o2, dtype=np.uint32)
        if threads and threads > 0:
            k = self.L.ora_map_pairs_mt(self.ctx, threads, n, first_read_id, b1.ctypes.data, o1.ctypes.data,
                                        b2.ctypes.data, o2.ctypes.data, C.cast(rec, C.c_void_p), C.byref(st))
        else:
            k = self.L.ora_map_pairs(self.ctx, n, first_read_id, b1.ctypes.data, o1.ctypes.data,
                                     b2.ctypes.data, o2.ctypes.data, C.cast(rec, C.c_void_p), C.byref(st))
        self.L.ora_set_trace(self.ctx, None)
        return rec, k, st, tr

    def write_bed(self, rec, k, path, p=None):
        p = p or self.p
        return self.L.ora_write_bed_pe(C.byref(self.ref), C.byref(p), C.cast(rec, C.c_void_p), k, path.encode())

    def close(self):
        if self.ctx:
            self.L.ora_destroy(self.ctx)
            self.ctx = None
            self.L.ora_index_free(C.byref(self.idx))
            self.L.ora_ref_free(C.byref(getattr(self, "_ref_loaded", self.ref)))


class OraPairsRecord(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("rid1", C.c_uint32), ("rid2", C.c_uint32), ("pos1", C.c_uint32),
                ("pos2", C.c_uint32), ("strand1", C.c_uint8), ("strand2", C.c_uint8), ("mapq", C.c_uint8),
                ("is_unique", C.c_uint8)]


assert C.sizeof(OraPairsRecord) == 24 == C.sizeof(OraRecord)


def read_names(path):
    """names (up to first whitespace) of a FASTA/FASTQ file's records, kseq style"""
    out = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln[:1] == b"@":
            out.append(ln[1:].split()[0])
            i += 4
        elif ln[:1] == b">":
            out.append(ln[1:].split()[0])
            i += 1
        else:
            i += 1
    return out


def write_pairs(oracle, rec, k, names, path):
    L = oracle.L
    L.ora_write_pairs.restype = C.c_long
    L.ora_write_pairs.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long,
                                  C.POINTER(C.c_char_p), C.c_char_p]
    arr = (C.c_char_p * len(names))(*names)
    if oracle.pairs_rank:
        L.ora_write_pairs_ranked.restype = C.c_long
        L.ora_write_pairs_ranked.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.POINTER(C.c_char_p),
                                             C.c_void_p, C.c_char_p]
        pr = (C.c_uint32 * len(oracle.pairs_rank))(*oracle.pairs_rank)
        return L.ora_write_pairs_ranked(C.byref(oracle.ref), C.byref(oracle.p), C.cast(rec, C.c_void_p), k, arr, pr, path.encode())
    return L.ora_write_pairs(C.byref(oracle.ref), C.byref(oracle.p), C.cast(rec, C.c_void_p), k, arr, path.encode())


class OraRecordBc(C.Structure):
    _fields_ = [("r", OraRecord), ("barcode", C.c_uint64)]


assert C.sizeof(OraRecordBc) == 32


def read_fastq_qual(path):
    import numpy as np
    L = lib()
    L.ora_read_fastq_qual.restype = C.c_long
    L.ora_read_fastq_qual.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    b = C.c_void_p(); q = C.c_void_p(); o = C.c_void_p()
    n = L.ora_read_fastq_qual(path.encode(), C.byref(b), C.byref(q), C.byref(o))
    if n < 0:
        raise IOError(path)
    off = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint32)), shape=(n + 1,)).copy()
    tot = int(off[-1])
    bases = np.ctypeslib.as_array(C.cast(b, C.POINTER(C.c_uint8)), shape=(max(tot, 1),)).copy()[:tot]
    quals = np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint8)), shape=(max(tot, 1),)).copy()[:tot]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for x in (b, q, o):
        libc.free(x)
    return bases, quals, off


class Whitelist:
    def __init__(self, path, barcode_length):
        L = lib()
        self.L = L
        L.ora_whitelist_load.restype = C.c_void_p
        L.ora_whitelist_load.argtypes = [C.c_char_p, C.c_uint32]
        L.ora_whitelist_free.argtypes = [C.c_void_p]
        L.ora_whitelist_abundance.restype = C.c_long
        L.ora_whitelist_abundance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.ora_whitelist_size.restype = C.c_uint32
        L.ora_whitelist_size.argtypes = [C.c_void_p]
        L.ora_whitelist_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.h = L.ora_whitelist_load(path.encode(), barcode_length)
        if not self.h:
            raise IOError(path)
        self.barcode_length = barcode_length

    def abundance(self, bc, bc_off):
        import numpy as np
        bc = np.ascontiguousarray(bc)
        bc_off = np.ascontiguousarray(bc_off, dtype=np.uint32)
        return self.L.ora_whitelist_abundance(self.h, bc.ctypes.data, bc_off.ctypes.data, len(bc_off) - 1)

    def export(self):
        import numpy as np
        n = self.L.ora_whitelist_size(self.h)
        k = np.zeros(n, np.uint64)
        c = np.zeros(n, np.uint32)
        self.L.ora_whitelist_export(self.h, k.ctypes.data, c.ctypes.data)
        return k, c


def map_pairs_bc(oracle, b1, o1, b2, o2, bc, bcq, bco, wl, threads=1):
    """returns (records OraRecordBc[], k, stats, n_in_whitelist, n_corrected); bc is corrected in place"""
    import numpy as np
    L = oracle.L
    L.ora_map_pairs_bc.restype = C.c_long
    L.ora_map_pairs_bc.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 7 + \
                                  [C.c_void_p, C.c_void_p, C.POINTER(OraStats), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    n = len(o1) - 1
    rec = (OraRecordBc * max(1, n))()
    st = OraStats()
    a = C.c_uint64(0)
    b = C.c_uint64(0)
    arrs = [np.ascontiguousarray(x) for x in (b1, o1, b2, o2, bc, bcq, bco)]
    k = L.ora_map_pairs_bc(oracle.ctx, threads, n, 0, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                           arrs[3].ctypes.data, arrs[4].ctypes.data, arrs[5].ctypes.data, arrs[6].ctypes.data, wl.h,
                           C.cast(rec, C.c_void_p), C.byref(st), C.byref(a), C.byref(b))
    if arrs[4] is not bc:
        bc[:] = arrs[4]
    return rec, k, st, int(a.value), int(b.value)


def write_bed_bc(oracle, rec, k, barcode_length, path):
    L = oracle.L
    L.ora_write_bed_pe_bc.restype = C.c_long
    L.ora_write_bed_pe_bc.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_uint32, C.c_char_p]
    return L.ora_write_bed_pe_bc(C.byref(oracle.ref), C.byref(oracle.p), C.cast(rec, C.c_void_p), k, barcode_length,
                                 path.encode())


def write_bed_bc_bulk(oracle, rec, k, barcode_length, wl, path):
    L = oracle.L
    L.ora_write_bed_pe_bc_bulk.restype = C.c_long
    L.ora_write_bed_pe_bc_bulk.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_uint32, C.c_void_p,
                                           C.c_char_p]
    return L.ora_write_bed_pe_bc_bulk(C.byref(oracle.ref), C.byref(oracle.p), C.cast(rec, C.c_void_p), k, barcode_length, wl.h,
                                      path.encode())


def map_single(oracle, b, off, threads=1):
    import numpy as np
    L = oracle.L
    L.ora_map_single.restype = C.c_long
    L.ora_map_single.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(OraStats)]
    n = len(off) - 1
    rec = (OraRecord * max(1, n * max(1, oracle.p.max_num_best_mappings)))()
    st = OraStats()
    b = np.ascontiguousarray(b)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    k = L.ora_map_single(oracle.ctx, threads, n, 0, b.ctypes.data, off.ctypes.data, C.cast(rec, C.c_void_p), C.byref(st))
    return rec, k, st


def write_bed_se(oracle, rec, k, path):
    L = oracle.L
    L.ora_write_bed_se.restype = C.c_long
    L.ora_write_bed_se.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_char_p]
    return L.ora_write_bed_se(C.byref(oracle.ref), C.byref(oracle.p), C.cast(rec, C.c_void_p), k, path.encode())


# ---- --SAM ---------------------------------------------------------------------------------
SAM_CIGAR_CAP = 64


class OraSamRecord(C.Structure):
    _fields_ = [("read_id", C.c_uint32), ("rid", C.c_uint32), ("pos", C.c_uint32), ("mpos", C.c_uint32),
                ("mrid", C.c_int32), ("tlen", C.c_int32), ("nm", C.c_uint32), ("flag", C.c_uint16),
                ("n_cigar", C.c_uint16), ("md_len", C.c_uint16), ("mapq", C.c_uint8), ("strand", C.c_uint8),
                ("is_unique", C.c_uint8), ("valid", C.c_uint8), ("reserved", C.c_uint16)]


class SamResult:
    """records + cigar / MD pools of one batch (slot layout of ora_sam_record)"""
    def __init__(self, n_slots, md_cap):
        import numpy as np
        self.rec = (OraSamRecord * max(1, n_slots))()
        self.cigar = np.zeros(max(1, n_slots) * SAM_CIGAR_CAP, np.uint32)
        self.md = np.zeros(max(1, n_slots) * md_cap, np.uint8)
        self.md_cap = md_cap
        self.n_slots = n_slots

    def tuples(self):
        out = []
        for i in range(self.n_slots):
            r = self.rec[i]
            if not r.valid:
                continue
            cg = tuple(int(x) for x in self.cigar[i * SAM_CIGAR_CAP:i * SAM_CIGAR_CAP + r.n_cigar])
            md = self.md[i * self.md_cap:i * self.md_cap + r.md_len].tobytes()
            out.append((i, r.read_id, r.rid, r.pos, r.mpos, r.mrid, r.tlen, r.nm, r.flag, r.mapq, r.strand, r.is_unique, cg, md,
                        r.reserved))
        return out


def map_pairs_sam(oracle, b1, o1, b2, o2, threads=1):
    import numpy as np
    L = oracle.L
    n = len(o1) - 1
    md_cap = 2 * int(max(np.diff(o1).max(initial=1), np.diff(o2).max(initial=1))) + 16
    res = SamResult(2 * n, md_cap)
    st = OraStats()
    L.ora_map_pairs_sam.restype = C.c_long
    L.ora_map_pairs_sam.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 7 + [C.c_uint32, C.POINTER(OraStats)]
    arrs = [np.ascontiguousarray(x) for x in (b1, o1, b2, o2)]
    k = L.ora_map_pairs_sam(oracle.ctx, threads, n, 0, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                            arrs[3].ctypes.data, C.cast(res.rec, C.c_void_p), res.cigar.ctypes.data, res.md.ctypes.data, md_cap,
                            C.byref(st))
    return res, k, st


def map_single_sam(oracle, b, off, threads=1):
    import numpy as np
    L = oracle.L
    n = len(off) - 1
    md_cap = 2 * int(np.diff(off).max(initial=1)) + 16
    res = SamResult(n, md_cap)
    st = OraStats()
    L.ora_map_single_sam.restype = C.c_long
    L.ora_map_single_sam.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint32, C.POINTER(OraStats)]
    arrs = [np.ascontiguousarray(x) for x in (b, off)]
    k = L.ora_map_single_sam(oracle.ctx, threads, n, 0, arrs[0].ctypes.data, arrs[1].ctypes.data, C.cast(res.rec, C.c_void_p),
                             res.cigar.ctypes.data, res.md.ctypes.data, md_cap, C.byref(st))
    return res, k, st


def write_sam(oracle, res, paired, names1, names2, b1, q1, o1, b2, q2, o2, path):
    import numpy as np
    L = oracle.L
    L.ora_write_sam.restype = C.c_long
    L.ora_write_sam.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)] + [C.c_void_p] * 7 + [C.c_char_p]
    n1 = (C.c_char_p * len(names1))(*names1)
    n2 = (C.c_char_p * max(1, len(names2 or [])))(*(names2 or [b""]))
    keep = [np.ascontiguousarray(x) if x is not None else None for x in (b1, q1, o1, b2, q2, o2)]
    ptr = [k.ctypes.data if k is not None else None for k in keep]
    return L.ora_write_sam(C.byref(oracle.ref), C.byref(oracle.p), C.cast(res.rec, C.c_void_p), res.n_slots, int(paired),
                           res.cigar.ctypes.data, res.md.ctypes.data, res.md_cap, n1, n2, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4],
                           ptr[5], None, path.encode())


def map_pairs_bc_sam(oracle, b1, o1, b2, o2, bc, bcq, bco, wl, threads=1):
    """single-cell --SAM: (SamResult, per-pair barcode keys, stats)"""
    import numpy as np
    L = oracle.L
    n = len(o1) - 1
    md_cap = 2 * int(max(np.diff(o1).max(initial=1), np.diff(o2).max(initial=1))) + 16
    res = SamResult(2 * n, md_cap)
    st = OraStats()
    keys = np.zeros(max(1, n), np.uint64)
    L.ora_map_pairs_bc_sam.restype = C.c_long
    L.ora_map_pairs_bc_sam.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 11 + [C.c_uint32, C.c_void_p,
                                                                                                       C.POINTER(OraStats)]
    arrs = [np.ascontiguousarray(x) for x in (b1, o1, b2, o2, bc, bcq, bco)]
    L.ora_map_pairs_bc_sam(oracle.ctx, threads, n, 0, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                           arrs[4].ctypes.data, arrs[5].ctypes.data, arrs[6].ctypes.data, wl.h, C.cast(res.rec, C.c_void_p),
                           res.cigar.ctypes.data, res.md.ctypes.data, md_cap, keys.ctypes.data, C.byref(st))
    return res, keys[:n], st


def write_sam_bc(oracle, res, paired, names1, names2, b1, q1, o1, b2, q2, o2, keys, barcode_length, path):
    import numpy as np
    L = oracle.L
    L.ora_write_sam_bc.restype = C.c_long
    L.ora_write_sam_bc.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)] + [C.c_void_p] * 7 + [C.c_uint32, C.c_char_p]
    n1 = (C.c_char_p * len(names1))(*names1)
    n2 = (C.c_char_p * max(1, len(names2 or [])))(*(names2 or [b""]))
    keep = [np.ascontiguousarray(x) if x is not None else None for x in (b1, q1, o1, b2, q2, o2)]
    ptr = [k.ctypes.data if k is not None else None for k in keep]
    kk = np.ascontiguousarray(keys, dtype=np.uint64)
    return L.ora_write_sam_bc(C.byref(oracle.ref), C.byref(oracle.p), C.cast(res.rec, C.c_void_p), res.n_slots, int(paired),
                              res.cigar.ctypes.data, res.md.ctypes.data, res.md_cap, n1, n2, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4],
                              ptr[5], kk.ctypes.data, barcode_length, path.encode())


def map_single_bc(oracle, b, off, bc, bcq, bco, wl, threads=1):
    """single-end reads with cell barcodes: (records OraRecordBc[], k, stats, n_in_whitelist, n_corrected)"""
    import numpy as np
    L = oracle.L
    L.ora_map_single_bc.restype = C.c_long
    L.ora_map_single_bc.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 5 + \
                                   [C.c_void_p, C.c_void_p, C.POINTER(OraStats), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    n = len(off) - 1
    rec = (OraRecordBc * max(1, n))()
    st = OraStats()
    a, b2 = C.c_uint64(0), C.c_uint64(0)
    arrs = [np.ascontiguousarray(x) for x in (b, off, bc, bcq, bco)]
    k = L.ora_map_single_bc(oracle.ctx, threads, n, 0, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                            arrs[3].ctypes.data, arrs[4].ctypes.data, wl.h, C.cast(rec, C.c_void_p), C.byref(st), C.byref(a), C.byref(b2))
    return rec, k, st, int(a.value), int(b2.value)


def write_se_bc(oracle, rec, k, barcode_length, wl, tagalign, path):
    L = oracle.L
    L.ora_write_se_bc.restype = C.c_long
    L.ora_write_se_bc.argtypes = [C.POINTER(OraRef), C.POINTER(OraParams), C.c_void_p, C.c_long, C.c_uint32, C.c_void_p, C.c_int,
                                  C.c_char_p]
    return L.ora_write_se_bc(C.byref(oracle.ref), C.byref(oracle.p), C.cast(rec, C.c_void_p), k, barcode_length, wl.h, int(tagalign),
                             path.encode())
