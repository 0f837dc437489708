"""Host-side mirror of the reference's mapping entry point for paired-end bulk data
(Chromap::MapPairedEndReads, src/chromap.h:636-1409) on top of the C ABI."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import SamRecord, SingleBatch, BarcodeBatch, Batch, IndexView, Params, Record, RecordBc, RefView, Stats


class ChromapError(RuntimeError):
    pass


def read_fastx(path):
    """FASTA/FASTQ -> (bases uint8[total], offsets uint32[n+1], names list).  Plain text,
    kseq record semantics (name up to first whitespace; zero-length records skipped)."""
    names, chunks, lens = [], [], []
    with open(path, "rb") as f:
        data = f.read()
    lines = data.split(b"\n")
    i, nl = 0, len(lines)
    while i < nl:
        ln = lines[i].rstrip(b"\r")
        if not ln:
            i += 1
            continue
        if ln[:1] == b"@":  # FASTQ, sequence may span lines until '+'
            name = ln[1:].split()[0] if len(ln) > 1 else b""
            i += 1
            seq = []
            while i < nl and not lines[i].startswith(b"+"):
                seq.append(lines[i].rstrip(b"\r"))
                i += 1
            s = b"".join(seq)
            i += 1  # '+'
            q = 0
            while i < nl and q < len(s):
                q += len(lines[i].rstrip(b"\r"))
                i += 1
        elif ln[:1] == b">":
            name = ln[1:].split()[0] if len(ln) > 1 else b""
            i += 1
            seq = []
            while i < nl and lines[i][:1] not in (b">", b"@"):
                seq.append(lines[i].rstrip(b"\r"))
                i += 1
            s = b"".join(seq)
        else:
            i += 1
            continue
        if len(s) == 0:
            continue
        names.append(name)
        chunks.append(s)
        lens.append(len(s))
    off = np.zeros(len(lens) + 1, dtype=np.uint32)
    if lens:
        off[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64)).astype(np.uint32)
    bases = np.frombuffer(b"".join(chunks), dtype=np.uint8).copy() if chunks else np.zeros(0, dtype=np.uint8)
    return bases, off, names


def read_fastq_pairs(path1, path2):
    b1, o1, _ = read_fastx(path1)
    b2, o2, _ = read_fastx(path2)
    if len(o1) != len(o2):
        raise ChromapError("Numbers of reads and barcodes don't match!")  # chromap.cc:110
    return b1, o1, b2, o2


class ChromapGPU:
    """One context per GPU: index + reference resident in HBM, batches mapped by HIP kernels."""

    def __init__(self, index_path=None, ref_path=None, params=None, device=0, preset=None, synthetic=None,
                 build_index=None, shared_from=None, **overrides):
        self.L = _capi.lib()
        chr_order = overrides.pop("chr_order", None)
        pairs_order = overrides.pop("pairs_order", None)
        self.params = params if params is not None else _capi.default_params(preset, **overrides)
        self.ctx = C.c_void_p()
        self._idx = None
        self._ref = None
        self.names = None
        if shared_from is not None:
            # a further context over shared_from's resident index + reference (keep shared_from alive)
            self.params = shared_from.params
            self._check(self.L.cmgpu_create_shared(shared_from.ctx, C.byref(self.ctx)), None)
            self.names = shared_from.names
            self._parent = shared_from
            self._inherit = (shared_from.rank, shared_from.pairs_rank)
        elif synthetic is not None:
            total, nseq, seed = synthetic[:3]
            rep = synthetic[3] if len(synthetic) > 3 and synthetic[3] else (0, 0, 0, 0.0)
            if isinstance(rep, str):  # "profile:1" -- a whole repeat landscape (cmgpu_create_synthetic_profile)
                rc = self.L.cmgpu_create_synthetic_profile(total, nseq, seed, 17, 7, C.byref(self.params), device, int(rep.split(":")[1]),
                                                           C.byref(self.ctx))
            else:
                fam, copies, elen, div = rep
                rc = self.L.cmgpu_create_synthetic_repeats(total, nseq, seed, 17, 7, C.byref(self.params), device, fam, copies, elen,
                                                           float(div), C.byref(self.ctx))
            self._check(rc, None)
            self.names = [b"chr%d" % (i + 1) for i in range(nseq)]
        elif index_path is None:
            # Index::Construct on the device (kmer/window from `build_index=(k, w)`, default 17/7)
            self._ref = RefView()
            if self.L.cmgpu_load_reference_fasta(ref_path.encode(), C.byref(self._ref)) != 0:
                raise ChromapError("cannot read reference %s" % ref_path)
            k, w = build_index if build_index else (17, 7)
            rc = self.L.cmgpu_create_from_reference(C.byref(self._ref), k, w, C.byref(self.params), device,
                                                    C.byref(self.ctx))
            self._check(rc, None)
            self.names = [self._ref.names[i] for i in range(self._ref.n_sequences)]
        else:
            self._idx = IndexView()
            self._ref = RefView()
            if self.L.cmgpu_load_index_file(index_path.encode(), C.byref(self._idx)) != 0:
                raise ChromapError("cannot read index file %s" % index_path)
            if self.L.cmgpu_load_reference_fasta(ref_path.encode(), C.byref(self._ref)) != 0:
                raise ChromapError("cannot read reference %s" % ref_path)
            rc = self.L.cmgpu_create(C.byref(self._idx), C.byref(self._ref), C.byref(self.params), device,
                                     C.byref(self.ctx))
            self._check(rc, None)
            self.names = [self._ref.names[i] for i in range(self._ref.n_sequences)]
        self.stats = Stats()
        self.rank = None
        self.pairs_rank = None
        if shared_from is not None:  # the library hands the parent's chromosome orders to the child as well
            self.rank, self.pairs_rank = self._inherit
        self._ex_keep = None
        if chr_order:
            self.set_chr_order(chr_order)
        if pairs_order:
            self.set_pairs_chr_order(pairs_order)

    def _ranks(self, order):
        pos = {(n if isinstance(n, bytes) else n.encode()): i for i, n in enumerate(order)}
        ranks = [pos.get(n, -1) for n in self.names]
        k = len(pos)
        for i, r in enumerate(ranks):
            if r < 0:
                ranks[i] = k
                k += 1
        if k != len(self.names):
            raise ChromapError("ERROR: unknown chromsome names found in chromosome order file.")
        return ranks

    def set_pairs_chr_order(self, order):
        """--pairs-natural-chr-order: which end of a pair is written first, and the header order (call after set_chr_order)"""
        self.pairs_rank = self._ranks(order)
        arr = (C.c_uint32 * len(self.pairs_rank))(*self.pairs_rank)
        self._check(self.L.cmgpu_set_pairs_chr_order(self.ctx, arr, len(self.pairs_rank)), self.ctx)

    def set_chr_order(self, order):
        """--chr-order: names in output order (unlisted sequences follow in reference order); from now on
        records carry ranks as rid and self.names is in rank order"""
        pos = {(n if isinstance(n, bytes) else n.encode()): i for i, n in enumerate(order)}
        ranks = [pos.get(n, -1) for n in self.names]
        k = len(pos)
        for i, r in enumerate(ranks):
            if r < 0:
                ranks[i] = k
                k += 1
        if k != len(self.names):
            raise ChromapError("ERROR: unknown chromsome names found in chromosome order file.")
        arr = (C.c_uint32 * len(ranks))(*ranks)
        self._check(self.L.cmgpu_set_chr_order(self.ctx, arr, len(ranks)), self.ctx)
        names = [None] * len(ranks)
        for i, r in enumerate(ranks):
            names[r] = self.names[i]
        self.names = names
        self.rank = ranks

    def reference_lengths(self):
        """lengths in the order of self.names"""
        nseq = C.c_uint32(0)
        self.L.cmgpu_reference_lengths(self.ctx, None, 0, C.byref(nseq))
        lens = (C.c_uint32 * nseq.value)()
        self.L.cmgpu_reference_lengths(self.ctx, lens, nseq.value, C.byref(nseq))
        out = list(lens)
        if self.rank:
            out = [0] * len(lens)
            for i, r in enumerate(self.rank):
                out[r] = lens[i]
        return (C.c_uint32 * len(out))(*out)

    def save_index(self, path):
        """Index::Save of the resident index (loads in the reference's kh_load)"""
        self._check(self.L.cmgpu_save_index_file(self.ctx, path.encode()), self.ctx)

    def _check(self, rc, ctx):
        if rc != 0:
            msg = self.L.cmgpu_last_error(ctx)
            raise ChromapError("chromap_amd error %d: %s" % (rc, (msg or b"").decode()))

    def _batch(self, b1, o1, b2, o2, first_read_id):
        self._keep = [np.ascontiguousarray(b1, dtype=np.uint8), np.ascontiguousarray(o1, dtype=np.uint32),
                      np.ascontiguousarray(b2, dtype=np.uint8), np.ascontiguousarray(o2, dtype=np.uint32)]
        k = self._keep
        return Batch(len(k[1]) - 1, first_read_id, k[0].ctypes.data, k[1].ctypes.data, k[2].ctypes.data,
                     k[3].ctypes.data)

    def _slots(self, n):
        """record capacity of a batch of n pairs / reads: max_num_best_mappings records each (-n)"""
        return max(1, n * max(1, int(self.params.max_num_best_mappings)))

    def map_pairs(self, b1, o1, b2, o2, first_read_id=0):
        """returns a ctypes array of Record (length = number of mapped pairs)"""
        bt = self._batch(b1, o1, b2, o2, first_read_id)
        cap = self._slots(bt.n_pairs)
        rec = (Record * cap)()
        n = C.c_uint64(0)
        rc = self.L.cmgpu_map_pairs(self.ctx, C.byref(bt), C.cast(rec, C.c_void_p), cap, C.byref(n),
                                    C.byref(self.stats))
        self._check(rc, self.ctx)
        return rec, int(n.value)

    # ---- single-end
    def map_pairs_async(self, b1, o1, b2, o2, first_read_id=0):
        """submits the batch and returns; call wait() for the records (one batch in flight)"""
        self._abatch = self._batch(b1, o1, b2, o2, first_read_id)
        self._check(self.L.cmgpu_map_pairs_async(self.ctx, C.byref(self._abatch), C.byref(self.stats)), self.ctx)

    def wait(self):
        n = self._slots(self._abatch.n_pairs)
        rec = (Record * n)()
        k = C.c_uint64(0)
        self._check(self.L.cmgpu_wait(self.ctx, C.cast(rec, C.c_void_p), n, C.byref(k)), self.ctx)
        return rec, int(k.value)

    # ---- page-locked host memory and the pipelined host-buffer entry
    def host_array(self, n, dtype):
        """numpy array of n items in page-locked host memory (cmgpu_host_alloc); kept alive by the mapper"""
        import numpy as np
        dt = np.dtype(dtype)
        nbytes = max(1, n * dt.itemsize)
        ptr = self.L.cmgpu_host_alloc(nbytes)
        if not ptr:
            raise ChromapError("cmgpu_host_alloc failed")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(ptr)
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dt, count=n)

    def submit_pairs(self, b1, o1, b2, o2, first_read_id=0):
        """starts the upload of a batch on the copy stream (cmgpu_submit_pairs); the arrays must stay alive until map_submitted"""
        bt = self._batch(b1, o1, b2, o2, first_read_id)
        self._sub_keep = getattr(self, "_sub_keep", [])
        self._sub_keep.append((bt, b1, o1, b2, o2))
        self._check(self.L.cmgpu_submit_pairs(self.ctx, C.byref(bt)), self.ctx)

    def map_submitted(self, out=None, capacity=0, stats=None):
        """maps the oldest submitted batch; out: Record array (or pinned numpy uint8 array of capacity * 24 bytes) or None"""
        n = C.c_uint64(0)
        st = stats if stats is not None else self.stats
        ptr = None if out is None else (C.c_void_p(out.ctypes.data) if hasattr(out, "ctypes") else C.cast(out, C.c_void_p))
        self._check(self.L.cmgpu_map_submitted(self.ctx, ptr, capacity, C.byref(n), C.byref(st)), self.ctx)
        self._sub_keep.pop(0)
        return int(n.value)

    def map_submitted_async(self, out, capacity, stats=None):
        """maps the oldest submitted batch and leaves the download of its records to `out` (pinned) running; records_wait()"""
        st = stats if stats is not None else self.stats
        ptr = C.c_void_p(out.ctypes.data) if hasattr(out, "ctypes") else C.cast(out, C.c_void_p)
        self._check(self.L.cmgpu_map_submitted_async(self.ctx, ptr, capacity, C.byref(st)), self.ctx)
        self._sub_keep.pop(0)

    def records_wait(self):
        """the oldest pending record download is complete: its record count"""
        n = C.c_uint64(0)
        self._check(self.L.cmgpu_records_wait(self.ctx, C.byref(n)), self.ctx)
        return int(n.value)

    def map_pairs_pipelined(self, b1, o1, b2, o2, repeats=4):
        """throughput of the host-buffer boundary with page-locked buffers and the upload of batch c+1 under the kernels of
        batch c (bench.py: pcie_inclusive.pipelined)"""
        import time
        import numpy as np
        n = len(o1) - 1
        bufs = []
        for _ in range(2):  # two sets of pinned input buffers, one pinned record array each
            pb1, pb2 = self.host_array(len(b1), np.uint8), self.host_array(len(b2), np.uint8)
            po1, po2 = self.host_array(n + 1, np.uint32), self.host_array(n + 1, np.uint32)
            pb1[:] = b1; pb2[:] = b2; po1[:] = o1; po2[:] = o2
            bufs.append((pb1, po1, pb2, po2, self.host_array(n * 24, np.uint8)))
        self.submit_pairs(*bufs[0][:4])
        self.submit_pairs(*bufs[1][:4])
        k0 = self.map_submitted(bufs[0][4], n, Stats())   # warm-up of both buffer sets
        self.submit_pairs(*bufs[0][:4])
        self.map_submitted(bufs[1][4], n, Stats())
        t0 = time.perf_counter()
        for i in range(repeats):
            self.submit_pairs(*bufs[(i + 1) & 1][:4])
            k = self.map_submitted(bufs[i & 1][4], n, Stats())
            assert k == k0
        dt = time.perf_counter() - t0
        self.map_submitted(bufs[repeats & 1][4], n, Stats())  # drain
        out = {"M pairs/s": round(n * repeats / dt / 1e6, 2), "ms_per_batch": round(dt / repeats * 1e3, 2), "records": int(k0),
               "note": "page-locked host buffers, cmgpu_submit_pairs of batch c+1 before cmgpu_map_submitted of batch c (upload under "
                       "the kernels), records compacted on the device and downloaded in one copy"}
        # the same with the record download of batch c under the kernels of batch c+1 (cmgpu_map_submitted_async / cmgpu_records_wait)
        ref = bytes(bufs[0][4][:k0 * 24])  # (every batch here is the same reads: the synchronous path's records)
        for b in bufs:
            b[4][:] = 0
        self.submit_pairs(*bufs[0][:4])
        self.submit_pairs(*bufs[1][:4])
        self.map_submitted_async(bufs[0][4], n, Stats())
        t0 = time.perf_counter()
        for i in range(1, repeats + 1):
            self.submit_pairs(*bufs[(i + 1) & 1][:4])
            self.map_submitted_async(bufs[i & 1][4], n, Stats())
            k = self.records_wait()
            assert k == k0
        dt = time.perf_counter() - t0
        same = bytes(bufs[(repeats - 1) & 1][4][:k0 * 24]) == ref
        self.records_wait()
        self.map_submitted(bufs[(repeats + 1) & 1][4], n, Stats())  # drain the last submitted batch
        out["download_overlapped"] = {"M pairs/s": round(n * repeats / dt / 1e6, 2), "ms_per_batch": round(dt / repeats * 1e3, 2),
                                      "records_identical_to_synchronous_path": bool(same),
                                      "note": "cmgpu_map_submitted_async + cmgpu_records_wait: the records of batch c travel under the kernels of batch c+1"}
        return out

    def map_single(self, b, off, first_read_id=0):
        self._keep = [np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint32)]
        n = len(self._keep[1]) - 1
        bt = SingleBatch(n, first_read_id, self._keep[0].ctypes.data, self._keep[1].ctypes.data)
        rec = (Record * self._slots(n))()
        k = C.c_uint64(0)
        self._check(self.L.cmgpu_map_single(self.ctx, C.byref(bt), C.cast(rec, C.c_void_p), self._slots(n), C.byref(k), C.byref(self.stats)),
                    self.ctx)
        return rec, int(k.value)

    def map_single_barcoded(self, b, off, bc, bcq, bco, first_read_id=0):
        """single-end reads with cell barcodes (MappingWithBarcode); records also stay resident for the store"""
        self._keep = [np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint32)]
        n = len(self._keep[1]) - 1
        bt = SingleBatch(n, first_read_id, self._keep[0].ctypes.data, self._keep[1].ctypes.data)
        kb = [np.ascontiguousarray(bc, dtype=np.uint8), np.ascontiguousarray(bcq, dtype=np.uint8),
              np.ascontiguousarray(bco, dtype=np.uint32)]
        bb = BarcodeBatch(kb[0].ctypes.data, kb[1].ctypes.data, kb[2].ctypes.data)
        rec = (RecordBc * self._slots(n))()
        k = C.c_uint64(0)
        rc = self.L.cmgpu_map_single_barcoded(self.ctx, C.byref(bt), C.byref(bb), C.cast(rec, C.c_void_p), self._slots(n), C.byref(k),
                                              C.byref(self.stats))
        self._check(rc, self.ctx)
        return rec, int(k.value)

    def write_bed_se(self, rec, n, path, params=None):
        p = params if params is not None else self.params
        names = (C.c_char_p * len(self.names))(*self.names)
        k = self.L.cmgpu_write_bed_se(names, len(self.names), C.byref(p), C.cast(rec, C.c_void_p), n, path.encode())
        if k < 0:
            raise ChromapError("cannot write %s" % path)
        return int(k)

    # ---- single-cell barcodes
    def set_whitelist_file(self, path, barcode_length):
        keys = C.c_void_p()
        n = C.c_uint32(0)
        if self.L.cmgpu_load_whitelist_file(path.encode(), barcode_length, C.byref(keys), C.byref(n)) != 0:
            raise ChromapError("cannot read whitelist %s" % path)
        self._check(self.L.cmgpu_set_whitelist(self.ctx, keys, n.value, barcode_length), self.ctx)
        C.CDLL(None).free(keys)
        self.barcode_length = barcode_length

    def compute_barcode_abundance(self, bc, bco):
        bc = np.ascontiguousarray(bc, dtype=np.uint8)
        bco = np.ascontiguousarray(bco, dtype=np.uint32)
        ns = C.c_uint64(0)
        self._check(self.L.cmgpu_compute_barcode_abundance(self.ctx, bc.ctypes.data, bco.ctypes.data, len(bco) - 1,
                                                           C.byref(ns)), self.ctx)
        return int(ns.value)

    def map_pairs_barcoded(self, b1, o1, b2, o2, bc, bcq, bco, first_read_id=0):
        bt = self._batch(b1, o1, b2, o2, first_read_id)
        kb = [np.ascontiguousarray(bc, dtype=np.uint8), np.ascontiguousarray(bcq, dtype=np.uint8),
              np.ascontiguousarray(bco, dtype=np.uint32)]
        bb = BarcodeBatch(kb[0].ctypes.data, kb[1].ctypes.data, kb[2].ctypes.data)
        rec = (RecordBc * self._slots(bt.n_pairs))()
        n = C.c_uint64(0)
        rc = self.L.cmgpu_map_pairs_barcoded(self.ctx, C.byref(bt), C.byref(bb), C.cast(rec, C.c_void_p), self._slots(bt.n_pairs),
                                             C.byref(n), C.byref(self.stats))
        self._check(rc, self.ctx)
        return rec, int(n.value)

    def write_bed_bc(self, rec, n, path, params=None):
        p = params if params is not None else self.params
        names = (C.c_char_p * len(self.names))(*self.names)
        k = self.L.cmgpu_write_bed_pe_bc(names, len(self.names), C.byref(p), C.cast(rec, C.c_void_p), n, self.barcode_length,
                                         path.encode())
        if k < 0:
            raise ChromapError("cannot write %s" % path)
        return int(k)

    def upload(self, b1, o1, b2, o2, first_read_id=0):
        bt = self._batch(b1, o1, b2, o2, first_read_id)
        self._check(self.L.cmgpu_upload_batch(self.ctx, C.byref(bt)), self.ctx)

    def generate_resident(self, n_pairs, read_length=50, frag_min=100, frag_max=600, sub_rate=0.01, seed=1, indel_rate=0.0, hic=None):
        """synthetic pairs on the device; hic = fraction of pairs with a ligation junction inside a read: Hi-C shaped pairs
        (mates from independent loci) instead of fragments"""
        if hic is not None:
            self._check(self.L.cmgpu_generate_resident_batch_hic(self.ctx, n_pairs, read_length, sub_rate, indel_rate, float(hic), seed), self.ctx)
        else:
            self._check(self.L.cmgpu_generate_resident_batch_indels(self.ctx, n_pairs, read_length, frag_min, frag_max, sub_rate,
                                                                    indel_rate, seed), self.ctx)
        self._n_resident = n_pairs

    def swap_resident(self, slot):
        """the resident batch changes places with the one parked in `slot` (0..7)"""
        self._check(self.L.cmgpu_swap_resident_batch(self.ctx, slot), self.ctx)

    def get_option(self, name):
        v = C.c_int64(0)
        self._check(self.L.cmgpu_get_option(self.ctx, name.encode(), C.byref(v)), self.ctx)
        return int(v.value)

    def set_option(self, name, value):
        self._check(self.L.cmgpu_set_option(self.ctx, name.encode(), int(value)), self.ctx)

    def map_resident(self, stats=None):
        n = C.c_uint64(0)
        st = stats if stats is not None else self.stats
        self._check(self.L.cmgpu_map_resident(self.ctx, C.byref(n), C.byref(st)), self.ctx)
        return int(n.value)

    def download_records(self, capacity):
        rec = (Record * max(1, capacity))()
        n = C.c_uint64(0)
        self._check(self.L.cmgpu_download_records(self.ctx, C.cast(rec, C.c_void_p), capacity, C.byref(n)), self.ctx)
        return rec, int(n.value)

    def timings(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        k = self.L.cmgpu_last_timings(self.ctx, names, ms, 32)
        return [(names[i].decode(), float(ms[i])) for i in range(k)]

    # ---- --SAM (create the mapper with output_format=_capi.FORMAT_SAM)
    def download_sam(self):
        """SAM records of the last mapped batch: (records, cigar_pool, md_pool, md_cap, n_slots)"""
        ns, cap = C.c_uint64(0), C.c_uint32(0)
        self._check(self.L.cmgpu_sam_layout(self.ctx, C.byref(ns), C.byref(cap)), self.ctx)
        n = int(ns.value)
        rec = (SamRecord * max(1, n))()
        cigar = np.zeros(max(1, n) * _capi.SAM_CIGAR_CAP, np.uint32)
        md = np.zeros(max(1, n) * max(1, cap.value), np.uint8)
        self._check(self.L.cmgpu_download_sam(self.ctx, C.cast(rec, C.c_void_p), cigar.ctypes.data, md.ctypes.data), self.ctx)
        return rec, cigar, md, int(cap.value), n

    def download_barcode_keys(self, n):
        """(corrected) barcode keys of the last barcoded batch, one per pair / read"""
        keys = np.zeros(max(1, n), np.uint64)
        self._check(self.L.cmgpu_download_barcode_keys(self.ctx, keys.ctypes.data), self.ctx)
        return keys[:n]

    def write_sam(self, sam, paired, names1, names2, b1, q1, o1, b2, q2, o2, path, params=None, barcode_keys=None, barcode_length=0):
        rec, cigar, md, md_cap, n = sam
        p = params if params is not None else self.params
        rn = (C.c_char_p * len(self.names))(*self.names)
        lens = self.reference_lengths()
        n1 = (C.c_char_p * len(names1))(*names1)
        n2 = (C.c_char_p * max(1, len(names2 or [])))(*(names2 or [b""]))
        keep = [np.ascontiguousarray(x) if x is not None else None for x in (b1, q1, o1, b2, q2, o2)]
        ptr = [k.ctypes.data if k is not None else None for k in keep]
        if barcode_keys is not None:
            bk = np.ascontiguousarray(barcode_keys, dtype=np.uint64)
            k = self.L.cmgpu_write_sam_barcoded(rn, C.cast(lens, C.c_void_p), len(self.names), C.byref(p), C.cast(rec, C.c_void_p), n,
                                                int(paired), cigar.ctypes.data, md.ctypes.data, md_cap, n1, n2, ptr[0], ptr[1], ptr[2],
                                                ptr[3], ptr[4], ptr[5], bk.ctypes.data, int(barcode_length), path.encode())
        else:
            k = self.L.cmgpu_write_sam(rn, C.cast(lens, C.c_void_p), len(self.names), C.byref(p), C.cast(rec, C.c_void_p), n, int(paired),
                                       cigar.ctypes.data, md.ctypes.data, md_cap, n1, n2, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], ptr[5],
                                       path.encode())
        if k < 0:
            raise ChromapError("cannot write %s" % path)
        return int(k)

    # ---- FASTQ ingest on the device
    def fastq_scan(self, stream, text, final=True, bgzf=False):
        """text: bytes of a FASTQ chunk; returns the number of complete non-empty records in it"""
        n = C.c_uint32(0)
        f = self.L.cmgpu_fastq_scan_bgzf if bgzf else self.L.cmgpu_fastq_scan  # bgzf: whole compressed BGZF blocks, inflated on the device
        self._check(f(self.ctx, stream, text, len(text), int(final), C.byref(n)), self.ctx)
        return n.value

    def fastq_take(self, stream, n):
        used = C.c_uint64(0)
        self._check(self.L.cmgpu_fastq_take(self.ctx, stream, n, C.byref(used)), self.ctx)
        return used.value

    def fastq_commit(self, n, first_read_id=0, paired=True, barcoded=False):
        self._check(self.L.cmgpu_fastq_commit(self.ctx, n, first_read_id, int(paired), int(barcoded)), self.ctx)
        self._n_resident = n

    def download_batch(self, n):
        """resident batch back on the host: (b1, o1, b2, o2)"""
        o1 = np.zeros(n + 1, np.uint32)
        o2 = np.zeros(n + 1, np.uint32)
        self._check(self.L.cmgpu_download_batch(self.ctx, None, o1.ctypes.data, None, o2.ctypes.data), self.ctx)
        b1 = np.zeros(max(1, int(o1[n])), np.uint8)
        b2 = np.zeros(max(1, int(o2[n])), np.uint8)
        self._check(self.L.cmgpu_download_batch(self.ctx, b1.ctypes.data, None, b2.ctypes.data, None), self.ctx)
        return b1[:int(o1[n])], o1, b2[:int(o2[n])], o2

    # ---- device-side post-processing: record store -> sorted / deduplicated BED text in HBM
    def store_clear(self):
        self._check(self.L.cmgpu_store_clear(self.ctx), self.ctx)

    def store_reserve(self, n_records, barcoded=False):
        self._check(self.L.cmgpu_store_reserve(self.ctx, n_records, int(barcoded)), self.ctx)

    def store_append_resident(self):
        """appends the records of the last map_* call (still resident); returns the store size"""
        n = C.c_uint64(0)
        self._check(self.L.cmgpu_store_append_resident(self.ctx, C.byref(n)), self.ctx)
        return n.value

    def store_append(self, records, n, on_device=False, barcoded=False):
        """records: ctypes array / address of Record (or RecordBc) entries, host or device"""
        ptr = records if isinstance(records, int) else C.cast(records, C.c_void_p)
        self._check(self.L.cmgpu_store_append(self.ctx, ptr, n, int(on_device), int(barcoded)), self.ctx)

    def store_format(self, kind=_capi.TEXT_BED_PE, params=None, barcode_length=0):
        """sort + duplicate removal + MAPQ filter + Tn5 shift + BED text, all in HBM; returns (lines, bytes)"""
        p = params if params is not None else self.params
        names = (C.c_char_p * len(self.names))(*self.names)
        nl, nb = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.cmgpu_store_format(self.ctx, kind, names, len(self.names), C.byref(p), barcode_length,
                                              C.byref(nl), C.byref(nb)), self.ctx)
        return nl.value, nb.value

    def store_text(self):
        nb = C.c_uint64(0)
        self.L.cmgpu_store_info(self.ctx, None, C.byref(nb), None)
        buf = C.create_string_buffer(max(1, nb.value))
        self._check(self.L.cmgpu_store_text(self.ctx, buf, nb.value), self.ctx)
        return buf.raw[:nb.value]

    def store_write_text(self, path, append=False):
        self._check(self.L.cmgpu_store_write_text(self.ctx, path.encode(), int(append)), self.ctx)

    def write_bed(self, rec, n, path, params=None):
        p = params if params is not None else self.params
        names = (C.c_char_p * len(self.names))(*self.names)
        k = self.L.cmgpu_write_bed_pe(names, len(self.names), C.byref(p), C.cast(rec, C.c_void_p), n, path.encode())
        if k < 0:
            raise ChromapError("cannot write %s" % path)
        return int(k)

    def write_pairs(self, rec, n, read_names, path, read_id_base=0, params=None):
        """rec: Record array returned by map_pairs with split_alignment set (holds PairsRecord entries)"""
        p = params if params is not None else self.params
        names = (C.c_char_p * len(self.names))(*self.names)
        lens = self.reference_lengths()
        rn = (C.c_char_p * len(read_names))(*read_names)
        pr = (C.c_uint32 * len(self.pairs_rank))(*self.pairs_rank) if self.pairs_rank else None
        k = self.L.cmgpu_write_pairs_ranked(names, lens, len(self.names), C.byref(p), C.cast(rec, C.c_void_p), n, rn,
                                            read_id_base, pr, path.encode())
        if k < 0:
            raise ChromapError("cannot write %s" % path)
        return int(k)

    def store_format_pairs(self, read_names, read_id_base=0, params=None):
        """pairs text of the stored pairs records on the device (sort, MAPQ filter, lines); returns (lines, bytes);
        the text is fetched with store_text() / store_write_text(); the header lines: write_pairs_header()"""
        p = params if params is not None else self.params
        names = (C.c_char_p * len(self.names))(*self.names)
        blob = b"".join(read_names)
        off = np.zeros(len(read_names) + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in read_names], dtype=np.uint64)
        nl, nb = C.c_uint64(0), C.c_uint64(0)
        self._check(self.L.cmgpu_store_format_pairs(self.ctx, names, len(self.names), C.byref(p), blob, off.ctypes.data, len(read_names),
                                                    read_id_base, C.byref(nl), C.byref(nb)), self.ctx)
        return int(nl.value), int(nb.value)

    def write_pairs_header(self, path):
        names = (C.c_char_p * len(self.names))(*self.names)
        pr = (C.c_uint32 * len(self.pairs_rank))(*self.pairs_rank) if self.pairs_rank else None
        if self.L.cmgpu_write_pairs_header(names, self.reference_lengths(), len(self.names), pr, path.encode()) != 0:
            raise ChromapError("cannot write %s" % path)

    # ---- multi-GPU record exchange (include/chromap_amd.h: cmgpu_exchange_*)
    def exchange_unique_id(self):
        buf = C.create_string_buffer(_capi.UNIQUE_ID_BYTES)
        self._check(self.L.cmgpu_exchange_unique_id(buf), None)
        return buf.raw

    def exchange_init(self, unique_id, rank, world):
        """RCCL communicator of this context's device (collective over the `world` contexts)"""
        self._check(self.L.cmgpu_exchange_init(self.ctx, C.c_char_p(unique_id), rank, world), self.ctx)

    def exchange_init_external(self, transport, rank, world):
        """transport: object with allgather_counts(mine: list) -> list of rows, and
        alltoallv(send_dev, send_counts, recv_dev, recv_counts, record_bytes) working on DEVICE addresses"""
        def _ag(user, mine, matrix, n):
            try:
                rows = transport.allgather_counts([int(mine[i]) for i in range(n)])
                for r, row in enumerate(rows):
                    for i in range(n):
                        matrix[r * n + i] = int(row[i])
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self._ex_error = e
                return -1

        def _a2a(user, send_dev, send_counts, recv_dev, recv_counts, w, rb):
            try:
                transport.alltoallv(send_dev, [int(send_counts[i]) for i in range(w)], recv_dev,
                                    [int(recv_counts[i]) for i in range(w)], rb)
                return 0
            except Exception as e:
                self._ex_error = e
                return -1

        t = _capi.ExchangeTransport(None, _capi.ALLGATHER_COUNTS_FN(_ag), _capi.ALLTOALLV_FN(_a2a))
        self._ex_keep = t  # the callbacks must outlive the exchange
        self._ex_error = None
        self._check(self.L.cmgpu_exchange_init_external(self.ctx, C.byref(t), rank, world), self.ctx)

    def exchange_owner_table(self, world):
        out = (C.c_uint8 * len(self.names))()
        self._check(self.L.cmgpu_exchange_owner_table(self.ctx, world, out, len(self.names)), self.ctx)
        return list(out)

    def exchange_step(self, world=None):
        """records of the resident batch -> their chromosomes' owners; returns (sent per rank, received)"""
        w = C.c_int(0)
        self.L.cmgpu_exchange_info(self.ctx, None, C.byref(w), None, None)
        sent = (C.c_uint64 * max(1, w.value))()
        nr = C.c_uint64(0)
        self._check(self.L.cmgpu_exchange_step(self.ctx, sent, C.byref(nr)), self.ctx)
        return [int(x) for x in sent[:w.value]], int(nr.value)

    def exchange_info(self):
        r, w, a, b = C.c_int(0), C.c_int(0), C.c_uint64(0), C.c_uint64(0)
        self.L.cmgpu_exchange_info(self.ctx, C.byref(r), C.byref(w), C.byref(a), C.byref(b))
        return {"rank": r.value, "world": w.value, "records_sent": int(a.value), "records_received": int(b.value)}

    def exchange_finalize(self):
        self._check(self.L.cmgpu_exchange_finalize(self.ctx), self.ctx)

    def close(self):
        if self.ctx:
            self.L.cmgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()
        for ptr in getattr(self, "_pinned", []):
            self.L.cmgpu_host_free(ptr)
        self._pinned = []
        if self._idx is not None:
            self.L.cmgpu_free_host_index(C.byref(self._idx))
            self._idx = None
        if self._ref is not None:
            self.L.cmgpu_free_host_ref(C.byref(self._ref))
            self._ref = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
