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
    # the boundary header and the header of the measurement / test hooks (no entry point is declared in both)
    hdr = open(os.path.join(ROOT, "include", "chromap_amd.h")).read()
    dbg = open(os.path.join(ROOT, "include", "chromap_amd_debug.h")).read()
    boundary = set(re.findall(r"\b(cmgpu_[a-z_0-9]+)\s*\(", hdr))
    hooks = set(re.findall(r"\b(cmgpu_[a-z_0-9]+)\s*\(", dbg))
    assert not (boundary & hooks)
    assert not [s for s in boundary if "debug" in s or "bench" in s or "synthetic" in s or s.endswith("_option")]
    declared = boundary | hooks
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


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """the drop-in boundary is a C ABI: include/chromap_amd.h compiles as C99 (-pedantic), a C program links against the
    library, and without a device cmgpu_create* fails with a message instead of falling back to anything"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "chromap_amd")
    if not os.path.exists(os.path.join(lib_dir, "libchromap_amd.so")):
        pytest.skip("library not built")
    src = tmp_path / "t.c"
    src.write_text('#include "chromap_amd.h"\n#include "chromap_amd_debug.h"\n#include <stdio.h>\n'
                   'int main(void) {\n  cmgpu_params p;\n  cmgpu_default_params(&p);\n'
                   '  if (cmgpu_apply_preset(&p, "atac") != 0) return 2;\n  cmgpu_ctx *ctx = 0;\n'
                   '  int rc = cmgpu_create_synthetic(1000000, 2, 1, 17, 7, &p, 0, &ctx);\n'
                   '  printf("%d %d %s\\n", p.max_insert_size, rc, cmgpu_last_error(0));\n  return 0;\n}\n')
    exe = str(tmp_path / "t")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(root, "include"), str(src), "-o", exe,
                        "-L" + lib_dir, "-lchromap_amd", "-Wl,-rpath," + lib_dir], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    out = subprocess.run([exe], stdout=subprocess.PIPE).stdout.decode().split(None, 2)
    assert out[0] == "2000"
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except ImportError:
        has_gpu = False
    if not has_gpu:
        assert int(out[1]) != 0 and "HIP" in out[2]
