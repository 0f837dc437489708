"""Device-built synthetic genome + index + reads (cm_synth.hip) against the oracle: the
reference bytes and the read batch are downloaded, the oracle builds its own index from
them with the reference's algorithm and maps the same reads; records must be identical."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _export_fasta(g, path):
    L = g.L
    n = C.c_uint32(0)
    L.cmgpu_reference_lengths(g.ctx, None, 0, C.byref(n))
    lens = (C.c_uint32 * n.value)()
    L.cmgpu_reference_lengths(g.ctx, lens, n.value, C.byref(n))
    with open(path, "wb") as f:
        for i in range(n.value):
            buf = C.create_string_buffer(lens[i])
            assert L.cmgpu_export_reference(g.ctx, i, buf, lens[i]) == 0
            f.write(b">chr%d\n" % (i + 1))
            f.write(buf.raw[: lens[i]])
            f.write(b"\n")
    return [int(x) for x in lens]


def _tuples(rec, k, gpu):
    out = []
    for i in range(k):
        r = rec[i]
        if gpu:
            out.append((r.read_id, r.rid, r.fragment_start, r.fragment_length, r.mapq, r.direction, r.is_unique,
                        r.positive_alignment_length, r.negative_alignment_length))
        else:
            out.append((r.read_id, r.rid, r.fragment_start, r.fragment_length, r.mapq, r.direction, r.is_unique,
                        r.pos_aln_len, r.neg_aln_len))
    return sorted(out)


@pytest.mark.parametrize("preset,readlen,fmin,fmax", [("atac", 50, 30, 600), ("chip", 100, 200, 700)])
def test_synthetic_index_and_reads_match_oracle(preset, readlen, fmin, fmax, tmp_path):
    from chromap_amd import ChromapGPU
    g = ChromapGPU(synthetic=(3_000_000, 5, 4242), preset=preset)
    fa = str(tmp_path / "syn.fa")
    lens = _export_fasta(g, fa)
    assert len(lens) == 5 and abs(sum(lens) - 3_000_000) < 100
    n = 30000
    g.generate_resident(n, read_length=readlen, frag_min=fmin, frag_max=fmax, sub_rate=0.01, seed=7)
    b1 = np.zeros(n * readlen, np.uint8)
    b2 = np.zeros(n * readlen, np.uint8)
    o1 = np.zeros(n + 1, np.uint32)
    o2 = np.zeros(n + 1, np.uint32)
    assert g.L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
    assert o1[-1] == n * readlen and set(np.unique(b1)) <= set(b"ACGT")
    k = g.map_resident()
    rec, k2 = g.download_records(n)
    assert k2 == k
    o = ol.Oracle(None, fa, ol.params(preset))
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    assert ok == k and k > 0.8 * n
    assert _tuples(rec, k, True) == _tuples(orec, ok, False)
    s = g.stats.as_dict()
    od = ost.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == od[key], key
    # size-independent property: records are one per read_id, coordinates inside chromosomes
    ids = [rec[i].read_id for i in range(k)]
    assert len(set(ids)) == k
    for i in range(k):
        assert rec[i].fragment_start + rec[i].fragment_length <= lens[rec[i].rid]
    g.close()
    o.close()


def test_mosaic_genome_profile_matches_oracle(tmp_path):
    """the repeat-landscape genome of bench.py's third workload (cmgpu_create_synthetic_profile): about a fifth of the bases
    repeat-derived; every record of 20 000 pairs equal to the oracle's, and the repeats do what they are there for --
    multi-mapped pairs and reads with long candidate lists"""
    from chromap_amd import ChromapGPU
    g = ChromapGPU(synthetic=(12_000_000, 3, 99, "profile:1"), preset="atac")
    fa = str(tmp_path / "mosaic.fa")
    lens = _export_fasta(g, fa)
    n, readlen = 20000, 50
    g.generate_resident(n, read_length=readlen, frag_min=30, frag_max=600, sub_rate=0.01, seed=11)
    b1, o1, b2, o2 = g.download_batch(n)
    k = g.map_resident()
    rec, k2 = g.download_records(n)
    o = ol.Oracle(None, fa, ol.params("atac"))
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    assert ok == k
    assert _tuples(rec, k, True) == _tuples(orec, ok, False)
    s, od = g.stats.as_dict(), ost.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == od[key], key
    assert s["num_mapped_reads"] - s["num_uniquely_mapped_reads"] > 0  # repeats make multi-mappers
    assert s["num_candidates"] > s["num_mapped_reads"]
    print("mosaic: mapped %d unique %d candidates %d" % (s["num_mapped_reads"], s["num_uniquely_mapped_reads"], s["num_candidates"]))
    g.close()
    o.close()


def test_hic_shaped_pairs_match_oracle(tmp_path):
    """Hi-C shaped synthetic pairs (mates from independent loci, a third of the pairs with a ligation junction inside a read)
    under --preset hic: every pairs record equal to the oracle's; the chimeric reads really take the split-alignment branch
    (a junction read maps although half of it belongs elsewhere)"""
    from chromap_amd import ChromapGPU, _capi
    g = ChromapGPU(synthetic=(6_000_000, 4, 4243), preset="hic")
    fa = str(tmp_path / "syn.fa")
    _export_fasta(g, fa)
    n, readlen = 20000, 150
    g.generate_resident(n, read_length=readlen, sub_rate=0.01, indel_rate=0.001, seed=5, hic=0.35)
    b1, o1, b2, o2 = g.download_batch(n)
    k = g.map_resident()
    rec, k2 = g.download_records(n)
    assert k2 == k
    o = ol.Oracle(None, fa, ol.params("hic"))
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    assert ok == k and k > 0.8 * n
    pg = C.cast(rec, C.POINTER(_capi.PairsRecord))
    po = C.cast(orec, C.POINTER(ol.OraPairsRecord))
    tg = sorted((pg[i].read_id, pg[i].rid1, pg[i].rid2, pg[i].pos1, pg[i].pos2, pg[i].strand1, pg[i].strand2, pg[i].mapq, pg[i].is_unique) for i in range(k))
    to = sorted((po[i].read_id, po[i].rid1, po[i].rid2, po[i].pos1, po[i].pos2, po[i].strand1, po[i].strand2, po[i].mapq, po[i].is_unique) for i in range(ok))
    assert tg == to
    # mates from independent loci: many pairs span sequences or lie far apart
    far = sum(1 for t in tg if t[1] != t[2] or abs(int(t[3]) - int(t[4])) > 5000)
    assert far > 0.5 * k
    g.close()
    o.close()
