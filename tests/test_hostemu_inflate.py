"""The lane-level DEFLATE decoder of the device's BGZF inflate (chromap_amd/csrc/cm_inflate.h), compiled for the host, against zlib:
stored, fixed and dynamic blocks, several deflate blocks per stream, FASTQ-like and random bytes, every compression level; and
damaged streams -- truncated, bit flips, wrong sizes, wrong CRC -- must end with an error code (or, a flip zlib also does not
notice, with the same bytes) and never touch memory outside the buffers (guard bytes around them)."""
import ctypes as C
import zlib

import numpy as np

import hostemu_lib as he


def _inflate(L, comp, n_out, crc, lane=0):
    guard = 64
    out = np.full(n_out + 2 * guard, 0x5A, np.uint8)
    cin = np.frombuffer(comp, np.uint8).copy() if len(comp) else np.zeros(1, np.uint8)
    f = L.hostemu_inflate_block
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    rc = f(cin.ctypes.data, len(comp), out.ctypes.data + guard, n_out, crc, lane)
    assert (out[:guard] == 0x5A).all() and (out[guard + n_out:] == 0x5A).all(), "the decoder wrote outside its output"
    return rc, bytes(out[guard:guard + n_out])


def _raw(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, chunks=1):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    out = b""
    step = max(1, len(data) // chunks)
    for i in range(0, len(data), step):
        out += c.compress(data[i:i + step])
        if chunks > 1:
            out += c.flush(zlib.Z_FULL_FLUSH)  # several deflate blocks, an empty stored block between them
    return out + c.flush()


def _fastq(rng, n):
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 151))
        s = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L))
        q = bytes(rng.integers(35, 75, L).astype(np.uint8))
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(recs)


def test_decoder_equals_zlib_on_valid_streams():
    L = he.lib()
    rng = np.random.default_rng(5)
    datas = [b"", b"A", b"ACGT" * 10, _fastq(rng, 300)[:65536], bytes(rng.integers(0, 256, 40000).astype(np.uint8)),
             b"N" * 65536, _fastq(rng, 40), bytes(rng.integers(0, 4, 65536).astype(np.uint8))]
    n = 0
    for d in datas:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                for chunks in (1, 3):
                    comp = _raw(d, level, strategy, chunks)
                    rc, got = _inflate(L, comp, len(d), zlib.crc32(d), lane=n)
                    assert rc == 0 and got == d, (len(d), level, strategy, chunks, rc)
                    n += 1
    assert n == len(datas) * 32


def test_damaged_streams_end_with_an_error():
    L = he.lib()
    rng = np.random.default_rng(9)
    d = _fastq(rng, 200)[:30000]
    crc = zlib.crc32(d)
    comp = _raw(d, 6)
    # truncated input, wrong output sizes, wrong CRC
    for cut in (0, 1, 5, len(comp) // 2, len(comp) - 1):
        rc, _ = _inflate(L, comp[:cut], len(d), crc)
        assert rc != 0, cut
    assert _inflate(L, comp, len(d) - 1, crc)[0] != 0
    assert _inflate(L, comp, len(d) + 1, crc)[0] != 0
    assert _inflate(L, comp, len(d), crc ^ 1)[0] == 4
    # bit flips anywhere: an error, or -- where the flip does not change the decoded bytes -- the same output
    bad = 0
    for it in range(400):
        c = bytearray(comp if it % 3 else _raw(d, int(rng.integers(0, 10)), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED])), 2))
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))
        rc, got = _inflate(L, bytes(c), len(d), crc, lane=it)
        if rc == 0:
            assert got == d
        else:
            bad += 1
    assert bad > 250
    # garbage
    for it in range(300):
        g = bytes(rng.integers(0, 256, int(rng.integers(1, 400))).astype(np.uint8))
        rc, _ = _inflate(L, g, int(rng.integers(0, 5000)), 0, lane=it)
        assert rc != 0 or True


def test_phases_alternate_for_any_step_budget():
    """the first pass with 1, 3 and 50 symbols between two header phases (the device: 2048) gives the same bytes"""
    L = he.lib()
    rng = np.random.default_rng(11)
    d = _fastq(rng, 120)
    try:
        for steps in (1, 3, 50):
            L.hostemu_inflate_steps(steps)
            for level, chunks in ((1, 1), (6, 3), (0, 2)):
                comp = _raw(d, level, zlib.Z_DEFAULT_STRATEGY, chunks)
                rc, got = _inflate(L, comp, len(d), zlib.crc32(d), lane=steps)
                assert rc == 0 and got == d, (steps, level, chunks, rc)
    finally:
        L.hostemu_inflate_steps(2048)


def _resolve_reference(lits, toks):
    """tokens in stream order: (literals before, length, distance) with length 0 for a token without a match"""
    out = bytearray()
    li = 0
    for nb, ln, ds in toks:
        out += lits[li:li + nb]
        li += nb
        for _ in range(ln):
            out.append(out[-ds])
    out += lits[li:]
    return bytes(out)


def test_second_pass_on_token_streams_no_compressor_writes():
    """chains of matches that read each other's output, matches that overlap themselves at every distance, 9..16-byte and longer
    matches, runs of literals beyond 511, groups that end in the middle of a chain; and tokens that reach outside the block"""
    L = he.lib()
    f = L.hostemu_bgzf_resolve
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int]
    rng = np.random.default_rng(17)
    for it in range(60):
        toks, n_out, n_lit = [], 0, 0
        style = it % 4
        n = int(rng.integers(1, 400))
        for k in range(n):
            nb = int(rng.choice([0, 0, 1, 2, 5, 40, 510, 511])) if style != 1 else int(rng.integers(0, 3))
            if k == 0 and nb == 0:
                nb = 1
            if nb == 511:
                toks.append((511, 0, 0))
                n_out += 511
                n_lit += 511
                continue
            n_out += nb
            n_lit += nb
            ln = int(rng.choice([3, 4, 8, 9, 16, 17, 64, 65, 258])) if style == 2 else int(rng.integers(3, 259 if style == 3 else 20))
            hi = min(n_out, 32768) if style == 0 else min(n_out, int(rng.choice([1, 2, 3, 7, 8, 9, 30, 300])))
            ds = int(rng.integers(1, hi + 1))
            toks.append((nb, ln, ds))
            n_out += ln
            if n_out > 60000:
                break
        tail = int(rng.integers(0, 50))
        n_out += tail
        n_lit += tail
        lits = bytes(rng.integers(0, 256, n_lit).astype(np.uint8))
        want = _resolve_reference(lits, toks)
        assert len(want) == n_out
        # the window as the first pass leaves it: literals in place, the matches' bytes untouched (here: 0xEE)
        win = np.full(n_out, 0xEE, np.uint8)
        words = np.zeros(max(1, len(toks)), np.uint32)
        pos = li = 0
        for i, (nb, ln, ds) in enumerate(toks):
            win[pos:pos + nb] = np.frombuffer(lits[li:li + nb], np.uint8)
            pos += nb + ln
            li += nb
            words[i] = 511 if nb == 511 else nb | (ln - 3) << 9 | (ds - 1) << 17
        win[pos:] = np.frombuffer(lits[li:], np.uint8)
        w = win.copy()
        assert f(w.ctypes.data, n_out, words.ctypes.data, len(toks), it & 1) == 0
        assert w.tobytes() == want, (it, style)
        # a distance before the block's first byte, a match past its end
        if toks[0][1]:
            bad = words.copy()
            bad[0] = (bad[0] & 0x1ffff) | (toks[0][0] << 17)  # distance = literals before + 1
            assert f(win.copy().ctypes.data, n_out, bad.ctypes.data, len(toks), 0) == 3
        assert f(win.copy().ctypes.data, max(0, n_out - tail - 1), words.ctypes.data, len(toks), 0) == 3
