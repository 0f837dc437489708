"""CPU-side checks of the C ABI library: it loads, exports every symbol the header
declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from chromap_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _capi.lib(), _capi


def test_exports_every_declared_symbol():
    L, capi = _lib()
    hdr = open(os.path.join(ROOT, "include", "chromap_amd.h")).read()
    declared = set(re.findall(r"\b(cmgpu_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
        # every entry point has its signature declared to ctypes (a missing one silently truncates pointers to 32 bits)
        assert getattr(L, s).argtypes is not None, "no ctypes signature for %s in chromap_amd/_capi.py" % s


def test_presets_and_defaults():
    L, capi = _lib()
    p = capi.default_params()
    assert (p.error_threshold, p.min_num_seeds, p.max_seed_frequency0, p.max_seed_frequency1) == (8, 2, 500, 1000)
    assert (p.max_insert_size, p.min_read_length, p.mapq_threshold) == (1000, 30, 30)
    a = capi.default_params("atac")
    assert (a.max_insert_size, a.trim_adapters, a.remove_pcr_duplicates, a.tn5_shift, a.low_memory_mode) == (2000, 1, 1, 1, 1)
    c = capi.default_params("chip")
    assert (c.max_insert_size, c.trim_adapters, c.remove_pcr_duplicates, c.tn5_shift) == (2000, 0, 1, 0)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L, capi = _lib()
    idx = capi.IndexView()
    ref = capi.RefView()
    idx.n_buckets = 4
    p = capi.default_params()
    ctx = C.c_void_p()
    rc = L.cmgpu_create(C.byref(idx), C.byref(ref), C.byref(p), 0, C.byref(ctx))
    assert rc == -2  # CMGPU_ENODEVICE
    assert b"no HIP device" in L.cmgpu_last_error(None)
