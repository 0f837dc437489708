"""The cooperative stage functions (chromap_amd/csrc/cm_coop.h: a group of lanes per read with a long hit / candidate
list) run on the CPU with one OS thread per lane (tests/hostemu/emu_group.h) and compared with the oracle on the
repeat-rich fuzz data: same records, same counters.  Group sizes 16 / 64 / 256, and table sizes small enough that the
'decline' path (more runs than the tables hold -> the one-lane definition) is taken too."""
import ctypes as C

import pytest

import fuzz_data
import hostemu_lib as he
from test_hostemu_fuzz import run_case


def _set(L, G, thr, P, MM, RB, reverse=0):
    L.hostemu_set_coop.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.hostemu_set_coop(G, thr, P, MM, RB)
    L.hostemu_set_coop_order(int(reverse))


def _items(L):
    a = (C.c_ulonglong * 8)()
    L.hostemu_coop_items(a)
    return list(a)


# (group size, list length above which a read goes to the group, hits the work area holds, minimizer table, run table)
# + whether the emulated lanes take their turns in descending order
GEOMETRIES = [(64, 8, 8192, 64, 130, 0), (256, 16, 8192, 64, 130, 1), (16, 8, 2048, 64, 130, 1), (64, 8, 700, 5, 11, 1)]


@pytest.mark.parametrize("geo", GEOMETRIES, ids=["G%d_P%d_MM%d" % (g[0], g[2], g[3]) for g in GEOMETRIES])
@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS[:4], ids=[str(c[0]) for c in fuzz_data.CONFIGS[:4]])
def test_cooperative_stages_equal_oracle(cfg, geo, tmp_path):
    L = he.lib()

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        h = he.HostEmu(idx, fa, he.params(preset, **gkw))
        _set(L, *geo)
        try:
            rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
            items = _items(L)
        finally:
            _set(L, 0, 0, 0, 0, 0, 0)
        factory.items = items
        return rec, k, st.as_dict()
    run_case(factory, cfg, tmp_path)
    done, declined = factory.items[0], factory.items[1]
    assert done > 0, "no read went through the cooperative hit-list stage"
    if geo[3] < 10:
        assert declined > 0, "the decline path was not taken"
