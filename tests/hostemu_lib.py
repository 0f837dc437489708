"""Loads tests/hostemu/libhostemu.so: the product's stage functions (cm_stages.h) compiled
for the host and driven by plain loops.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, ROOT)
from chromap_amd import _capi  # noqa: E402

_L = None


def lib():
    global _L
    if _L is None:
        so = os.path.join(ROOT, "tests", "hostemu", "libhostemu.so")
        srcs = [os.path.join(ROOT, "tests", "hostemu", "hostemu.cpp")] + \
               [os.path.join(ROOT, "tests", "hostemu", "emu_group.h")] + \
               [os.path.join(ROOT, "chromap_amd", "csrc", f) for f in ("cm_stages.h", "cm_coop.h", "cm_types.h", "cm_host.cpp",
                                                                       "cm_mapq_tables.h", "cm_inflate.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            # -fno-strict-aliasing: the stage functions read and write their byte / word arrays through wider types (aligned 8-
            # and 16-byte accesses), which the device compiler takes as written
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing",
                                   "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-pthread", "-o", so, srcs[0],
                                   os.path.join(ROOT, "chromap_amd", "csrc", "cm_host.cpp")])
        _L = _capi.declare(C.CDLL(so))
        P = C.POINTER
        _L.hostemu_map_pairs.restype = C.c_int
        _L.hostemu_map_pairs.argtypes = [P(_capi.IndexView), P(_capi.RefView), P(_capi.Params), P(_capi.Batch),
                                         C.c_void_p, P(C.c_uint64), P(_capi.Stats), C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
    return _L


# the oracle's names (oracle/chromap_oracle.h) for the same parameters
CAPI_NAMES = {"max_seed_freq0": "max_seed_frequency0", "max_seed_freq1": "max_seed_frequency1", "low_mem": "low_memory_mode"}


def params(preset=None, **kw):
    L = lib()
    p = _capi.Params()
    L.cmgpu_default_params(C.byref(p))
    if preset:
        assert L.cmgpu_apply_preset(C.byref(p), preset.encode()) == 0
    for k, v in kw.items():
        if k == "chr_order":
            p._chr_order = list(v)  # applied by HostEmu.__init__
        elif k == "pairs_order":
            p._pairs_order = list(v)
        else:
            k = CAPI_NAMES.get(k, k)
            if k not in _capi.PARAM_FIELDS:  # a ctypes structure would take any attribute name silently
                raise KeyError(k)
            setattr(p, k, v)
    return p


class HostEmu:
    def __init__(self, index_path, ref_path, p):
        self.L = lib()
        self.idx = _capi.IndexView()
        self.ref = _capi.RefView()
        assert self.L.cmgpu_load_index_file(index_path.encode(), C.byref(self.idx)) == 0
        assert self.L.cmgpu_load_reference_fasta(ref_path.encode(), C.byref(self.ref)) == 0
        self.p = p
        self.names = [self.ref.names[i] for i in range(self.ref.n_sequences)]
        self.rank = None
        order = getattr(p, "_chr_order", None)
        if order:
            import datasets
            self.rank = datasets.chr_order_ranks(order, self.names)
            names = [None] * len(self.rank)
            for i, r in enumerate(self.rank):
                names[r] = self.names[i]
            self.names = names
        arr = (C.c_uint32 * max(1, len(self.rank or [])))(*(self.rank or [0]))
        self.L.hostemu_set_chr_order.argtypes = [C.c_void_p, C.c_uint32]
        self.L.hostemu_set_chr_order(arr, len(self.rank or []))
        self.pairs_rank = None
        porder = getattr(p, "_pairs_order", None)
        if porder:
            import datasets
            self.pairs_rank = datasets.chr_order_ranks(porder, self.names)
        arr2 = (C.c_uint32 * max(1, len(self.pairs_rank or [])))(*(self.pairs_rank or [0]))
        self.L.hostemu_set_pairs_chr_order.argtypes = [C.c_void_p, C.c_uint32]
        self.L.hostemu_set_pairs_chr_order(arr2, len(self.pairs_rank or []))

    def map_pairs(self, b1, o1, b2, o2, first_read_id=0):
        n = len(o1) - 1
        keep = [np.ascontiguousarray(b1, dtype=np.uint8), np.ascontiguousarray(o1, dtype=np.uint32),
                np.ascontiguousarray(b2, dtype=np.uint8), np.ascontiguousarray(o2, dtype=np.uint32)]
        bt = _capi.Batch(n, first_read_id, keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data,
                         keep[3].ctypes.data)
        rec = (_capi.Record * max(1, n * max(1, self.p.max_num_best_mappings)))()
        k = C.c_uint64(0)
        st = _capi.Stats()
        dbg = {"mm_cnt": np.zeros(2 * n, np.uint32), "ncand": np.zeros(2 * n, np.uint32),
               "ndraft": np.zeros(2 * n, np.uint32), "nbest": np.zeros(n, np.int32)}
        rc = self.L.hostemu_map_pairs(C.byref(self.idx), C.byref(self.ref), C.byref(self.p), C.byref(bt),
                                      C.cast(rec, C.c_void_p), C.byref(k), C.byref(st), dbg["mm_cnt"].ctypes.data,
                                      dbg["ncand"].ctypes.data, dbg["ndraft"].ctypes.data, dbg["nbest"].ctypes.data)
        assert rc == 0, rc
        return rec, int(k.value), st, dbg

    def map_single(self, b, off):
        n = len(off) - 1
        keep = [np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint32)]
        bt = _capi.SingleBatch(n, 0, keep[0].ctypes.data, keep[1].ctypes.data)
        rec = (_capi.Record * max(1, n * max(1, self.p.max_num_best_mappings)))()
        k = C.c_uint64(0)
        st = _capi.Stats()
        f = self.L.hostemu_map_single
        f.restype = C.c_int
        f.argtypes = [C.POINTER(_capi.IndexView), C.POINTER(_capi.RefView), C.POINTER(_capi.Params),
                      C.POINTER(_capi.SingleBatch), C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(_capi.Stats)]
        assert f(C.byref(self.idx), C.byref(self.ref), C.byref(self.p), C.byref(bt), C.cast(rec, C.c_void_p), C.byref(k),
                 C.byref(st)) == 0
        return rec, int(k.value), st

    def write_bed_se(self, rec, n, path):
        names = (C.c_char_p * len(self.names))(*self.names)
        return self.L.cmgpu_write_bed_se(names, len(self.names), C.byref(self.p), C.cast(rec, C.c_void_p), n, path.encode())

    def map_pairs_bc(self, b1, o1, b2, o2, bc, bcq, bco, wl_keys):
        n = len(o1) - 1
        keep = [np.ascontiguousarray(x) for x in (b1, o1, b2, o2, bc, bcq, bco, wl_keys)]
        bt = _capi.Batch(n, 0, keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data)
        bb = _capi.BarcodeBatch(keep[4].ctypes.data, keep[5].ctypes.data, keep[6].ctypes.data)
        rec = (_capi.RecordBc * max(1, n))()
        k = C.c_uint64(0)
        st = _capi.Stats()
        f = self.L.hostemu_map_pairs_bc
        f.restype = C.c_int
        f.argtypes = [C.POINTER(_capi.IndexView), C.POINTER(_capi.RefView), C.POINTER(_capi.Params), C.POINTER(_capi.Batch),
                      C.POINTER(_capi.BarcodeBatch), C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64),
                      C.POINTER(_capi.Stats)]
        rc = f(C.byref(self.idx), C.byref(self.ref), C.byref(self.p), C.byref(bt), C.byref(bb), keep[7].ctypes.data,
               len(keep[7]), C.cast(rec, C.c_void_p), C.byref(k), C.byref(st))
        assert rc == 0, rc
        return rec, int(k.value), st

    def map_single_bc(self, b, off, bc, bcq, bco, wl_keys):
        n = len(off) - 1
        keep = [np.ascontiguousarray(x) for x in (b, off, bc, bcq, bco, wl_keys)]
        bt = _capi.SingleBatch(n, 0, keep[0].ctypes.data, keep[1].ctypes.data)
        bb = _capi.BarcodeBatch(keep[2].ctypes.data, keep[3].ctypes.data, keep[4].ctypes.data)
        rec = (_capi.RecordBc * max(1, n))()
        k = C.c_uint64(0)
        st = _capi.Stats()
        f = self.L.hostemu_map_single_bc
        f.restype = C.c_int
        f.argtypes = [C.POINTER(_capi.IndexView), C.POINTER(_capi.RefView), C.POINTER(_capi.Params), C.POINTER(_capi.SingleBatch),
                      C.POINTER(_capi.BarcodeBatch), C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64),
                      C.POINTER(_capi.Stats)]
        rc = f(C.byref(self.idx), C.byref(self.ref), C.byref(self.p), C.byref(bt), C.byref(bb), keep[5].ctypes.data,
               len(keep[5]), C.cast(rec, C.c_void_p), C.byref(k), C.byref(st))
        assert rc == 0, rc
        return rec, int(k.value), st

    def write_bed_bc(self, rec, n, barcode_length, path):
        names = (C.c_char_p * len(self.names))(*self.names)
        return self.L.cmgpu_write_bed_pe_bc(names, len(self.names), C.byref(self.p), C.cast(rec, C.c_void_p), n,
                                            barcode_length, path.encode())

    def write_pairs(self, rec, n, read_names, path):
        names = (C.c_char_p * len(self.names))(*self.names)
        ll = [self.ref.lengths[i] for i in range(len(self.names))]
        if self.rank:
            ll2 = [0] * len(ll)
            for i, r in enumerate(self.rank):
                ll2[r] = ll[i]
            ll = ll2
        lens = (C.c_uint32 * len(self.names))(*ll)
        rn = (C.c_char_p * len(read_names))(*read_names)
        pr = (C.c_uint32 * len(self.pairs_rank))(*self.pairs_rank) if self.pairs_rank else None
        return self.L.cmgpu_write_pairs_ranked(names, lens, len(self.names), C.byref(self.p), C.cast(rec, C.c_void_p), n, rn, 0, pr,
                                        path.encode())

    def write_bed(self, rec, n, path):
        names = (C.c_char_p * len(self.names))(*self.names)
        return self.L.cmgpu_write_bed_pe(names, len(self.names), C.byref(self.p), C.cast(rec, C.c_void_p), n,
                                         path.encode())


class SamOut:
    """slot-addressed SAM records + pools (cmgpu_sam_record layout)"""
    def __init__(self, n_slots, md_cap):
        self.rec = (_capi.SamRecord * max(1, n_slots))()
        self.cigar = np.zeros(max(1, n_slots) * _capi.SAM_CIGAR_CAP, np.uint32)
        self.md = np.zeros(max(1, n_slots) * md_cap, np.uint8)
        self.md_cap = md_cap
        self.n_slots = n_slots

    def tuples(self):
        out = []
        for i in range(self.n_slots):
            r = self.rec[i]
            if not r.valid:
                continue
            cg = tuple(int(x) for x in self.cigar[i * _capi.SAM_CIGAR_CAP:i * _capi.SAM_CIGAR_CAP + r.n_cigar])
            md = self.md[i * self.md_cap:i * self.md_cap + r.md_len].tobytes()
            out.append((i, r.read_id, r.rid, r.pos, r.mpos, r.mrid, r.tlen, r.nm, r.flag, r.mapq, r.strand, r.is_unique, cg, md,
                        r.length_after_trim))
        return out


def write_sam(L, ref, p, so, paired, names1, names2, b1, q1, o1, b2, q2, o2, path):
    """the product's host SAM writer (cm_host.cpp) over slot-addressed records"""
    nseq = ref.n_sequences
    rn = (C.c_char_p * nseq)(*[ref.names[i] for i in range(nseq)])
    n1 = (C.c_char_p * len(names1))(*names1)
    n2 = (C.c_char_p * max(1, len(names2 or [])))(*(names2 or [b""]))
    keep = [np.ascontiguousarray(x) if x is not None else None for x in (b1, q1, o1, b2, q2, o2)]
    ptr = [k.ctypes.data if k is not None else None for k in keep]
    return L.cmgpu_write_sam(rn, C.cast(ref.lengths, C.c_void_p), nseq, C.byref(p), C.cast(so.rec, C.c_void_p), so.n_slots, int(paired),
                             so.cigar.ctypes.data, so.md.ctypes.data, so.md_cap, n1, n2, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4],
                             ptr[5], path.encode())


def map_sam(h, b1, o1, b2=None, o2=None):
    """HostEmu h: stage functions in --SAM mode; returns (SamOut, Stats)"""
    P = C.POINTER
    n = len(o1) - 1
    paired = b2 is not None
    mx = int(np.diff(o1).max(initial=1))
    if paired:
        mx = max(mx, int(np.diff(o2).max(initial=1)))
    so = SamOut(2 * n if paired else n, 2 * mx + 16)
    st = _capi.Stats()
    keep = [np.ascontiguousarray(x) for x in ((b1, o1, b2, o2) if paired else (b1, o1))]
    if paired:
        bt = _capi.Batch(n, 0, keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data)
        f = h.L.hostemu_map_pairs_sam
        f.argtypes = [P(_capi.IndexView), P(_capi.RefView), P(_capi.Params), P(_capi.Batch), C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_uint32, P(_capi.Stats)]
    else:
        bt = _capi.SingleBatch(n, 0, keep[0].ctypes.data, keep[1].ctypes.data)
        f = h.L.hostemu_map_single_sam
        f.argtypes = [P(_capi.IndexView), P(_capi.RefView), P(_capi.Params), P(_capi.SingleBatch), C.c_void_p, C.c_void_p,
                      C.c_void_p, C.c_uint32, P(_capi.Stats)]
    f.restype = C.c_int
    rc = f(C.byref(h.idx), C.byref(h.ref), C.byref(h.p), C.byref(bt), C.cast(so.rec, C.c_void_p), so.cigar.ctypes.data,
           so.md.ctypes.data, so.md_cap, C.byref(st))
    assert rc == 0, rc
    return so, st


def map_bc_sam(h, b1, o1, b2, o2, bc, bcq, bco, wl_keys):
    """HostEmu h: stage functions in --SAM mode on single-cell data; returns (SamOut, per-pair barcode keys, Stats)"""
    P = C.POINTER
    n = len(o1) - 1
    mx = max(int(np.diff(o1).max(initial=1)), int(np.diff(o2).max(initial=1)))
    so = SamOut(2 * n, 2 * mx + 16)
    st = _capi.Stats()
    keep = [np.ascontiguousarray(x) for x in (b1, o1, b2, o2, bc, bcq, bco, wl_keys)]
    bt = _capi.Batch(n, 0, keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data)
    bb = _capi.BarcodeBatch(keep[4].ctypes.data, keep[5].ctypes.data, keep[6].ctypes.data)
    keys = np.zeros(max(1, n), np.uint64)
    f = h.L.hostemu_map_pairs_bc_sam
    f.restype = C.c_int
    f.argtypes = [P(_capi.IndexView), P(_capi.RefView), P(_capi.Params), P(_capi.Batch), P(_capi.BarcodeBatch), C.c_void_p, C.c_uint32,
                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, P(_capi.Stats)]
    rc = f(C.byref(h.idx), C.byref(h.ref), C.byref(h.p), C.byref(bt), C.byref(bb), keep[7].ctypes.data, len(keep[7]),
           C.cast(so.rec, C.c_void_p), so.cigar.ctypes.data, so.md.ctypes.data, so.md_cap, keys.ctypes.data, C.byref(st))
    assert rc == 0, rc
    return so, keys[:n], st


def write_sam_bc(L, ref, p, so, names1, names2, b1, q1, o1, b2, q2, o2, keys, barcode_length, path):
    """the product's host SAM writer for single-cell data (CB:Z tag, barcode in the sort key)"""
    _capi.declare(L)
    nseq = ref.n_sequences
    rn = (C.c_char_p * nseq)(*[ref.names[i] for i in range(nseq)])
    n1 = (C.c_char_p * len(names1))(*names1)
    n2 = (C.c_char_p * len(names2))(*names2)
    keep = [np.ascontiguousarray(x) for x in (b1, q1, o1, b2, q2, o2)]
    kk = np.ascontiguousarray(keys, dtype=np.uint64)
    return L.cmgpu_write_sam_barcoded(rn, C.cast(ref.lengths, C.c_void_p), nseq, C.byref(p), C.cast(so.rec, C.c_void_p), so.n_slots, 1,
                                      so.cigar.ctypes.data, so.md.ctypes.data, so.md_cap, n1, n2, keep[0].ctypes.data,
                                      keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data, keep[4].ctypes.data,
                                      keep[5].ctypes.data, kk.ctypes.data, barcode_length, path.encode())
