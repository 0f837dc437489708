"""Chunked reference-minimizer collection used by the device index builder
(cm_ref_chunk_minimizers in cm_stages.h) must emit exactly what the sequential state
machine emits (oracle: ora_minimizers == minimizer_generator.cc:7-139)."""
import ctypes as C

import numpy as np
import pytest

import datasets
import hostemu_lib as he
import oracle_lib as ol
from chromap_amd import _capi


@pytest.mark.parametrize("case,chunk,warm", [("s4_atac_q0", 2048, 96), ("s4_atac_q0", 257, 40), ("s1_atac", 2048, 96),
                                               ("toy_chip", 1000, 31)])
def test_chunked_reference_minimizers(case, chunk, warm):
    fa, _, _ = datasets.case_inputs(case)
    L = he.lib()
    L.hostemu_ref_minimizers.restype = C.c_long
    L.hostemu_ref_minimizers.argtypes = [C.POINTER(_capi.RefView), C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_long]
    ref = _capi.RefView()
    assert L.cmgpu_load_reference_fasta(fa.encode(), C.byref(ref)) == 0
    total = sum(ref.lengths[i] for i in range(ref.n_sequences))
    h = np.zeros(total, np.uint64)
    t = np.zeros(total, np.uint64)
    n = L.hostemu_ref_minimizers(C.byref(ref), 17, 7, chunk, warm, h.ctypes.data, t.ctypes.data, total)
    assert n > 0
    O = ol.lib()
    oh_all, ot_all = [], []
    for r in range(ref.n_sequences):
        ln = ref.lengths[r]
        oh = np.zeros(ln + 1, np.uint64)
        ot = np.zeros(ln + 1, np.uint64)
        seq = C.string_at(ref.sequences[r], ln)
        c = O.ora_minimizers(seq, ln, r, 17, 7, oh.ctypes.data, ot.ctypes.data)
        oh_all.append(oh[:c])
        ot_all.append(ot[:c])
    oh = np.concatenate(oh_all)
    ot = np.concatenate(ot_all)
    assert n == len(oh)
    # same multiset of (hash, hit); the builder sorts by (hash, hit) afterwards
    a = np.lexsort((t[:n], h[:n]))
    b = np.lexsort((ot, oh))
    assert np.array_equal(h[:n][a], oh[b]) and np.array_equal(t[:n][a], ot[b])
    L.cmgpu_free_host_ref(C.byref(ref))
