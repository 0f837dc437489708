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


# ---- hand-made dynamic blocks: code shapes no compressor picks (the limit-compare decode's corner cases) ----
class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, n):  # n bits of v, least significant first
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, n):  # a Huffman code word: most significant bit first
        self.put(int(format(c, "0%db" % n)[::-1], 2), n)

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
        return bytes(self.out)


def _canonical(lens):
    codes, code = {}, 0
    for ln in range(1, 16):
        for s, l in enumerate(lens):
            if l == ln:
                codes[s] = (code, ln)
                code += 1
        code <<= 1
    return codes


_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXT = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
_DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
_DEXT = [0, 0, 0, 0] + [e for e in range(1, 14) for _ in (0, 1)]


def _dynamic_block(ll, dl, items, last=True):
    """one dynamic block: ll / dl the code lengths (286 / 30 entries), items: ('L', byte) or ('M', length code index, extra, distance
    code, extra); the code-length code: the 16 lengths 0..15 at four bits each (complete), no run lengths"""
    w = _Bits()
    w.put(1 if last else 0, 1)
    w.put(2, 2)
    w.put(286 - 257, 5)
    w.put(30 - 1, 5)
    w.put(19 - 4, 4)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    for s in order:
        w.put(4 if s < 16 else 0, 3)
    cl = _canonical([4] * 16 + [0, 0, 0])
    for l in list(ll) + list(dl):
        w.code(*cl[l])
    lc, dc = _canonical(ll), _canonical(dl)
    for it in items:
        if it[0] == "L":
            w.code(*lc[it[1]])
        else:
            _, li, lx, di, dx = it
            w.code(*lc[257 + li])
            w.put(lx, _LEXT[li])
            w.code(*dc[di])
            w.put(dx, _DEXT[di])
    w.code(*lc[256])
    return w


def test_first_pass_on_code_shapes_no_compressor_picks():
    """codes of every depth up to 15, a code of two words, a lone end-of-block word, a lone distance word (incomplete, as zlib
    allows), no distance code at all; random literals and matches over them -- zlib must accept the stream and the bytes must be its"""
    L = he.lib()
    rng = np.random.default_rng(23)
    staircase = list(range(1, 15)) + [15, 15]  # complete: 2^-1 + ... + 2^-14 + 2 * 2^-15
    n_checked = 0
    for shape in range(8):
        for rep in range(6):
            ll, dl = [0] * 286, [0] * 30
            if shape == 0:    # the deepest code there is: literals and length codes on a staircase
                syms = [256] + list(rng.choice(256, 9, replace=False)) + [257 + int(x) for x in rng.choice(29, 6, replace=False)]
                for s, l in zip(rng.permutation(syms), staircase):
                    ll[int(s)] = l
                for s, l in zip(rng.permutation(30)[:16], staircase):
                    dl[int(s)] = l
            elif shape == 1:  # every symbol: 8 / 9 bits for the 286 (complete: 226 x 2^-8 + 60 x 2^-9), five bits for 30 distances + padding words
                ll = [8] * 226 + [9] * 60
                ll = [int(x) for x in rng.permutation(ll)]
                dl = [5] * 28 + [4] * 2
            elif shape == 2:  # two words: one literal and the end of the block; no distance code
                ll[256] = 1
                ll[int(rng.integers(0, 256))] = 1
            elif shape == 3:  # the end-of-block word alone (incomplete code of one word): an empty block
                ll[256] = 1
            elif shape == 4:  # one distance word (incomplete), one length code
                ll[256], ll[65], ll[257 + int(rng.integers(0, 29))] = 2, 1, 2
                dl[int(rng.integers(0, 30))] = 1
            elif shape == 5:  # 15-bit words only where the stream uses them most
                syms = [int(x) for x in rng.choice(256, 13, replace=False)] + [256, 257, 260]
                for s, l in zip(syms, [15, 15] + list(range(14, 0, -1))):
                    ll[s] = l
                dl[0], dl[29] = 1, 1
            elif shape == 6:  # deep distance code, shallow literal code
                for s in list(rng.choice(256, 5, replace=False)) + [256, 257 + 28, 257]:
                    ll[int(s)] = 3
                for s, l in zip(rng.permutation(30)[:16], staircase):
                    dl[int(s)] = l
            else:             # lengths 7 and 8 mixed with one deep pair
                ll = [0] * 286
                pool = [int(x) for x in rng.permutation(286)]
                if 256 not in pool[:130]:
                    pool[0] = 256
                for s in pool[:126]:
                    ll[s] = 7
                for s in pool[126:128]:
                    ll[s] = 8
                for s in pool[128:130]:
                    ll[s] = 9
                # 126/128 + 2/256 + 2/512 = 0.996: two words of 9 bits missing -> fill with one of 8
                ll[pool[130] if pool[130] != 256 else pool[131]] = 8
                dl = [5] * 28 + [4] * 2
            lits = [s for s in range(256) if ll[s]]
            lens = [s - 257 for s in range(257, 286) if ll[s]]
            dists = [s for s in range(30) if dl[s]]
            items, out = [], bytearray()
            for _ in range(int(rng.integers(0, 400)) if (lits or lens) else 0):
                if lits and (not lens or not dists or not out or rng.random() < 0.6):
                    b = int(rng.choice(lits))
                    items.append(("L", b))
                    out.append(b)
                    continue
                if not (lens and dists and out):
                    continue
                li, di = int(rng.choice(lens)), int(rng.choice(dists))
                lx, dx = int(rng.integers(0, 1 << _LEXT[li])), int(rng.integers(0, 1 << _DEXT[di]))
                ln, ds = _LBASE[li] + lx, _DBASE[di] + dx
                if ds > len(out) or len(out) + ln > 65000:
                    continue
                items.append(("M", li, lx, di, dx))
                for _ in range(ln):
                    out.append(out[-ds])
            comp = _dynamic_block(ll, dl, items).done()
            want = bytes(out)
            assert zlib.decompressobj(-15).decompress(comp) == want, ("the hand-made stream is not valid", shape, rep)
            rc, got = _inflate(L, comp, len(want), zlib.crc32(want), lane=shape * 8 + rep)
            assert rc == 0 and got == want, (shape, rep, rc)
            n_checked += 1
            # one bit more or less of output is an error, and so is the stream cut short
            if len(want):
                assert _inflate(L, comp, len(want) - 1, zlib.crc32(want))[0] != 0
            assert _inflate(L, comp[:-1], len(want), zlib.crc32(want))[0] != 0 or len(comp) < 2
    assert n_checked == 48
