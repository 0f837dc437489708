// cm_inflate.h -- the DEFLATE streams (RFC 1951) of BGZF blocks (SAM spec 4.1: gzip members of at most 64 KiB that carry their
// compressed size in a 'BC' extra field) inflated on the device, so that the host only moves COMPRESSED file bytes (SURVEY.md
// 8(f)-2; the reference reads every input through one zlib gzread per file, sequence_batch.cc:22-62 / kseq.h).  Two passes:
//
//  1. cm_inflate_tokens: ONE LANE per block decodes the Huffman codes -- the part of DEFLATE that is serial per stream, so tens of
//     thousands of streams run side by side.  Literal bytes go straight to their place in the text; a match is NOT copied (a lane
//     copying bytes through global memory waits a memory round trip per byte) but written as a 32-bit token.  Canonical Huffman
//     codes: the code-length code of a dynamic header bit by bit over its per-length counts, the literal / length and distance
//     codes by comparing the stream's next bits against one limit per length (cm_inf_decode_lim); one symbol table per code in
//     the workgroup's shared memory (`sym`: entry i of this lane at sym[i * stride]; `len8`: the code lengths while a dynamic
//     header is read; `delta`: per length, a code word's place in the table); a 64-bit bit buffer refilled a 32-bit word at a
//     time, the next word already on its way.  The lanes of a wave alternate between two phases -- block headers (long, rare),
//     then up to `max_steps` symbols -- so that a lane that reaches a header does not make the 63 others sit through it at a
//     random time each.
//  2. cm_bgzf_resolve + cm_bgzf_crc: ONE WAVE per block with the block's text in LDS: the tokens' output places by a prefix sum
//     64 at a time, then the group's matches side by side -- a lane per short match -- in as many rounds as the longest chain of
//     matches that read each other's output has links; the CRC-32 of the text from 4 slices per lane, combined with the shifts
//     of zlib's crc32_combine.
//
// Every loop is bounded by the input or the output size: a damaged stream ends with an error code, never with an access outside
// [in, in + n_in), [out, out + n_out) or the block's token array.
// The same text compiles for the host (tests/hostemu: both passes against zlib on random and damaged streams).
#ifndef CM_INFLATE_H_
#define CM_INFLATE_H_

#include <stdint.h>

#include "cm_types.h"

#define CM_INF_OK 0
#define CM_INF_EINPUT 1   // the stream asks for bytes beyond its end
#define CM_INF_EOUTPUT 2  // more output than the block's ISIZE, or less
#define CM_INF_ECODE 3    // invalid block type, code lengths, code or distance
#define CM_INF_ECRC 4     // CRC-32 of the output differs from the trailer's
#define CM_INF_SYMS 320u  // symbol-table entries per lane: 288 literal / length + 32 distance
#define CM_INF_LENS 320u  // code lengths of a dynamic header: up to 286 + 30

struct CmInfBits {
  const uint8_t *in;
  uint32_t n_in, ip;  // ip: the next input byte not yet in the bit buffer
  uint64_t bb;
  int bc;
  int err;
  uint32_t nw;        // the 4 bytes at ip, loaded ahead of their use (valid while ip + 4 <= n_in)
};
CM_HD void cm_inf_prime(CmInfBits &s) {
  if (s.ip + 4u <= s.n_in) __builtin_memcpy(&s.nw, s.in + s.ip, 4);
}
// a word or, at the end of the input, as many whole bytes as the buffer takes (the stream's last code words may be shorter than
// what a decoder would like to see: the buffer is zero beyond the input, and taking more bits than it really holds is the error)
CM_HD void cm_inf_fill(CmInfBits &s) {
  if (s.bc <= 32 && s.ip + 4u <= s.n_in) {
    s.bb |= (uint64_t)s.nw << s.bc;
    s.bc += 32;
    s.ip += 4;
    cm_inf_prime(s);
    return;
  }
  while (s.bc <= 56 && s.ip < s.n_in) {
    s.bb |= (uint64_t)s.in[s.ip++] << s.bc;
    s.bc += 8;
  }
  cm_inf_prime(s);
}
CM_HD uint32_t cm_inf_bits(CmInfBits &s, int n) {
  if (s.bc < n) cm_inf_fill(s);
  if (s.bc < n) { s.err = CM_INF_EINPUT; return 0; }
  const uint32_t v = (uint32_t)(s.bb & ((1ull << n) - 1ull));
  s.bb >>= n;
  s.bc -= n;
  return v;
}
// per-length code counts of one canonical code, packed: count of length l (1..15) in 16 bits
struct CmInfCnt { uint16_t c[16]; };

// symbol of the next code word: the counts say how many codes each length has; the first code of a length is (first code of the
// previous length + its count) << 1 (RFC 1951 3.2.2).  Returns -1 for a code word no symbol has.
CM_HD int cm_inf_decode(CmInfBits &s, const CmInfCnt &cnt, const uint16_t *sym, uint32_t stride) {
  int code = 0, first = 0, index = 0;
  if (s.bc < 15) cm_inf_fill(s);
  uint32_t bits = (uint32_t)s.bb;
#pragma unroll
  for (int len = 1; len <= 15; ++len) {
    code |= (int)(bits & 1u);
    bits >>= 1;
    const int count = cnt.c[len];
    if (code - count < first) {
      if (s.bc < len) { s.err = CM_INF_EINPUT; return -1; }
      s.bb >>= len;
      s.bc -= len;
      return sym[(uint32_t)(index + (code - first)) * stride];
    }
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}
// counts + symbol table of a canonical code from its lengths (len8[i * stride], n symbols).  Returns 0, or -1 for an
// over-subscribed set of lengths (an incomplete one is accepted: its unused code words decode to -1).
CM_HD int cm_inf_build(CmInfCnt &cnt, uint16_t *sym, const uint8_t *len8, uint32_t n, uint32_t stride) {
#pragma unroll
  for (int l = 0; l < 16; ++l) cnt.c[l] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t l = len8[i * stride] & 15u;
#pragma unroll
    for (int q = 0; q < 16; ++q) cnt.c[q] = (uint16_t)(cnt.c[q] + ((uint32_t)q == l ? 1u : 0u));
  }
  int left = 1;
#pragma unroll
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= cnt.c[l];
    if (left < 0) return -1;
  }
  uint16_t offs[16];
  offs[1] = 0;
#pragma unroll
  for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + cnt.c[l]);
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t l = len8[i * stride] & 15u;
    if (l == 0) continue;
    uint32_t at = 0;
#pragma unroll
    for (int q = 1; q < 16; ++q)
      if ((uint32_t)q == l) { at = offs[q]; offs[q] = (uint16_t)(offs[q] + 1); }
    sym[at * stride] = (uint16_t)i;
  }
  cnt.c[0] = 0;
  return 0;
}

// The literal / length and the distance code are walked differently from the (short, rare) code-length code above: the next 31
// bits of the stream, first bit on top, compared against one limit per length -- canonical code words of length l, left-justified,
// lie in [limit of l - 1, limit of l) -- so the length is 1 + the number of limits at or below the window: 14 compares without a
// branch instead of a loop the lanes leave one by one.  delta (16 entries of this lane, shared memory): what is added to a code
// word of a length to get its place in the symbol table.
struct CmInfLim { uint32_t lim[16]; };
CM_HD uint32_t cm_inf_rev32(uint32_t x) {
#if defined(__clang__)
  return __builtin_bitreverse32(x);
#else
  x = (x >> 16) | (x << 16);
  x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
  x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
  x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
  return ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
#endif
}
CM_HD void cm_inf_limits(CmInfLim &L, const CmInfCnt &cnt, int16_t *delta, uint32_t stride) {
  uint32_t first = 0, offs = 0;
  L.lim[0] = 0;
#pragma unroll
  for (int l = 1; l <= 15; ++l) {
    const uint32_t c = cnt.c[l];
    delta[(uint32_t)l * stride] = (int16_t)((int)offs - (int)first);
    L.lim[l] = (first + c) << (31 - l);
    offs += c;
    first = (first + c) << 1;
  }
}
CM_HD int cm_inf_decode_lim(CmInfBits &s, const CmInfLim &L, const uint16_t *sym, const int16_t *delta, uint32_t stride) {
  if (s.bc < 15) cm_inf_fill(s);
  const uint32_t v = cm_inf_rev32((uint32_t)s.bb) >> 1;
  uint32_t len = 1;
#pragma unroll
  for (int l = 1; l <= 14; ++l) len += v >= L.lim[l] ? 1u : 0u;
  if (v >= L.lim[15]) return -1;  // a code word no symbol has
  if ((uint32_t)s.bc < len) { s.err = CM_INF_EINPUT; return -1; }
  const int idx = (int)(v >> (31u - len)) + (int)delta[len * stride];
  s.bb >>= len;
  s.bc -= (int)len;
  return sym[(uint32_t)idx * stride];
}

// crc_tab: the 256 entries of the reflected CRC-32 (polynomial 0xEDB88320)
CM_HD uint32_t cm_crc32_entry(uint32_t i) {
  uint32_t c = i;
  for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
  return c;
}

// ---- pass 1: one lane, the codes of one stream ----
// token: bits 0..8 the literals since the token before (0..510; 511: 511 literals and no match), 9..16 the match's length - 3,
// 17..31 its distance - 1.  A block of isize bytes has at most cm_inf_tok_cap(isize) of them.
#define CM_INF_TOK_SKIP 511u
CM_HD uint32_t cm_inf_tok_cap(uint32_t isize) { return isize / 3u + isize / 511u + 2u; }
#define CM_INF_PH_HEADER 0
#define CM_INF_PH_SYMBOLS 1
#define CM_INF_PH_DONE 2
struct CmInfLane {
  CmInfBits s;
  CmInfLim ll, dl;  // the limits of the literal / length and of the distance code
  uint32_t op, lit, ntok;
  int phase, last, rc;
};
CM_HD void cm_inf_fail(CmInfLane &L, int rc) { L.rc = rc; L.phase = CM_INF_PH_DONE; }
CM_HD void cm_inf_push(CmInfLane &L, uint32_t *tok, uint32_t cap, uint32_t v) {
  if (L.ntok >= cap) { cm_inf_fail(L, CM_INF_EOUTPUT); return; }
  tok[L.ntok++] = v;
}
CM_HD void cm_inf_literal(CmInfLane &L, uint8_t *out, uint32_t *tok, uint32_t cap, uint8_t b) {
  out[L.op++] = b;
  if (++L.lit == CM_INF_TOK_SKIP) { cm_inf_push(L, tok, cap, CM_INF_TOK_SKIP); L.lit = 0; }
}
// the header of the next deflate block: a stored block is copied whole (literals), the fixed or the transmitted codes are built
CM_HD void cm_inf_header(CmInfLane &L, uint8_t *out, uint32_t n_out, uint32_t *tok, uint32_t cap, uint16_t *sym, uint8_t *len8, int16_t *delta, uint32_t stride) {
  // the order of the code-length code's lengths as 5-bit fields: no table a lane would have to index in private memory
  const uint64_t clord_lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
  const uint64_t clord_hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
  CmInfBits &s = L.s;
  uint16_t *dsym = sym + 288u * stride;
  CmInfCnt lc, dc;
  L.last = (int)cm_inf_bits(s, 1);
  const uint32_t type = cm_inf_bits(s, 2);
  if (s.err) return cm_inf_fail(L, s.err);
  if (type == 0) {  // stored: the rest of the current byte is skipped, LEN, ~LEN, LEN bytes
    const int drop = s.bc & 7;
    s.bb >>= drop;
    s.bc -= drop;
    const uint32_t len = cm_inf_bits(s, 16), nlen = cm_inf_bits(s, 16);
    if (s.err) return cm_inf_fail(L, s.err);
    if (len != (~nlen & 0xffffu)) return cm_inf_fail(L, CM_INF_ECODE);
    // the whole bytes still in the bit buffer go back to the input
    s.ip -= (uint32_t)(s.bc >> 3);
    s.bb = 0;
    s.bc = 0;
    if (len > s.n_in - s.ip) return cm_inf_fail(L, CM_INF_EINPUT);
    if (len > n_out - L.op) return cm_inf_fail(L, CM_INF_EOUTPUT);
    for (uint32_t i = 0; i < len && L.phase != CM_INF_PH_DONE; ++i) cm_inf_literal(L, out, tok, cap, s.in[s.ip + i]);
    s.ip += len;
    cm_inf_prime(s);
    if (L.phase != CM_INF_PH_DONE && L.last) L.phase = CM_INF_PH_DONE;
    return;
  }
  if (type == 3) return cm_inf_fail(L, CM_INF_ECODE);
  if (type == 1) {  // the fixed codes
    for (uint32_t i = 0; i < 288; ++i) len8[i * stride] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
    (void)cm_inf_build(lc, sym, len8, 288, stride);
    for (uint32_t i = 0; i < 30; ++i) len8[i * stride] = 5;
    (void)cm_inf_build(dc, dsym, len8, 30, stride);
  } else {
    const uint32_t nlen = cm_inf_bits(s, 5) + 257, ndist = cm_inf_bits(s, 5) + 1, ncode = cm_inf_bits(s, 4) + 4;
    if (s.err) return cm_inf_fail(L, s.err);
    if (nlen > 286 || ndist > 30) return cm_inf_fail(L, CM_INF_ECODE);
    for (uint32_t i = 0; i < 19; ++i) len8[i * stride] = 0;
    for (uint32_t i = 0; i < ncode; ++i) {
      const uint32_t which = (uint32_t)((i < 12 ? clord_lo >> (5 * i) : clord_hi >> (5 * (i - 12))) & 31ull);
      len8[which * stride] = (uint8_t)cm_inf_bits(s, 3);
    }
    if (s.err) return cm_inf_fail(L, s.err);
    CmInfCnt cc;
    // (the code-length code's table shares the distance entries: it is dead before the distance table is built)
    if (cm_inf_build(cc, dsym, len8, 19, stride) != 0) return cm_inf_fail(L, CM_INF_ECODE);
    uint32_t i = 0;
    while (i < nlen + ndist) {
      const int c = cm_inf_decode(s, cc, dsym, stride);
      if (s.err) return cm_inf_fail(L, s.err);
      if (c < 0) return cm_inf_fail(L, CM_INF_ECODE);
      if (c < 16) { len8[i * stride] = (uint8_t)c; ++i; continue; }
      uint32_t rep, val = 0;
      if (c == 16) {
        if (i == 0) return cm_inf_fail(L, CM_INF_ECODE);
        val = len8[(i - 1) * stride];
        rep = 3 + cm_inf_bits(s, 2);
      } else if (c == 17) rep = 3 + cm_inf_bits(s, 3);
      else rep = 11 + cm_inf_bits(s, 7);
      if (s.err) return cm_inf_fail(L, s.err);
      if (i + rep > nlen + ndist) return cm_inf_fail(L, CM_INF_ECODE);
      for (uint32_t k = 0; k < rep; ++k) len8[(i + k) * stride] = (uint8_t)val;
      i += rep;
    }
    if (len8[256 * stride] == 0) return cm_inf_fail(L, CM_INF_ECODE);  // no end-of-block code
    if (cm_inf_build(lc, sym, len8, nlen, stride) != 0) return cm_inf_fail(L, CM_INF_ECODE);
    if (cm_inf_build(dc, dsym, len8 + (size_t)nlen * stride, ndist, stride) != 0) return cm_inf_fail(L, CM_INF_ECODE);
  }
  cm_inf_limits(L.ll, lc, delta, stride);
  cm_inf_limits(L.dl, dc, delta + 16u * stride, stride);
  L.phase = CM_INF_PH_SYMBOLS;
}
// up to max_steps literals / matches of the current deflate block; its end-of-block code ends the phase
CM_HD void cm_inf_symbols(CmInfLane &L, uint8_t *out, uint32_t n_out, uint32_t *tok, uint32_t cap, const uint16_t *sym, const int16_t *delta, uint32_t stride, uint32_t max_steps) {
  CmInfBits &s = L.s;
  const uint16_t *dsym = sym + 288u * stride;
  for (uint32_t step = 0; step < max_steps && L.phase == CM_INF_PH_SYMBOLS; ++step) {
    const int c = cm_inf_decode_lim(s, L.ll, sym, delta, stride);
    if (s.err) return cm_inf_fail(L, s.err);
    if (c < 0) return cm_inf_fail(L, CM_INF_ECODE);
    if (c < 256) {
      if (L.op >= n_out) return cm_inf_fail(L, CM_INF_EOUTPUT);
      cm_inf_literal(L, out, tok, cap, (uint8_t)c);
      continue;
    }
    if (c == 256) { L.phase = L.last ? CM_INF_PH_DONE : CM_INF_PH_HEADER; return; }
    const uint32_t li = (uint32_t)c - 257;
    if (li >= 29) return cm_inf_fail(L, CM_INF_ECODE);
    // lengths 3..10 one code each, then four codes per extra bit, 258 on its own; distances 1..4, then two codes per extra bit
    // (RFC 1951 3.2.5 in closed form)
    const uint32_t le = li < 8 || li == 28 ? 0u : (li - 4) >> 2;
    const uint32_t len = (li == 28 ? 258u : li < 8 ? 3u + li : 3u + ((4u + (li & 3u)) << le)) + (le ? cm_inf_bits(s, (int)le) : 0u);
    const int dcode = cm_inf_decode_lim(s, L.dl, dsym, delta + 16u * stride, stride);
    if (s.err) return cm_inf_fail(L, s.err);
    if (dcode < 0 || dcode >= 30) return cm_inf_fail(L, CM_INF_ECODE);
    const uint32_t de = dcode < 4 ? 0u : ((uint32_t)dcode >> 1) - 1u;
    const uint32_t dist = (dcode < 4 ? 1u + (uint32_t)dcode : 1u + ((2u + ((uint32_t)dcode & 1u)) << de)) + (de ? cm_inf_bits(s, (int)de) : 0u);
    if (s.err) return cm_inf_fail(L, s.err);
    if (dist > L.op) return cm_inf_fail(L, CM_INF_ECODE);  // (a BGZF block has no history before its first byte)
    if (len > n_out - L.op) return cm_inf_fail(L, CM_INF_EOUTPUT);
    cm_inf_push(L, tok, cap, L.lit | (len - 3u) << 9 | (dist - 1u) << 17);
    L.lit = 0;
    L.op += len;
  }
}
// whether any lane of the wave (on the host: this lane) is in a state
#ifdef __HIP_DEVICE_COMPILE__
#define CM_INF_ANY(p) (__any(p) != 0)
#else
#define CM_INF_ANY(p) (p)
#endif
// The codes of in[0 .. n_in): the literal bytes to their places in out[0 .. n_out), the matches to tok[0 .. *n_tok).  Exactly n_out
// bytes of output or an error.  sym: CM_INF_SYMS entries of this lane (literal / length table, then 32 distance entries), len8:
// CM_INF_LENS entries, delta: 32 entries.  active: false for a lane without a block (it only keeps the wave's phases company).
CM_HD int cm_inflate_tokens(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, uint32_t *tok, uint32_t *n_tok, bool active,
                            uint16_t *sym, uint8_t *len8, int16_t *delta, uint32_t stride, uint32_t max_steps) {
  CmInfLane L;
  L.s.in = in; L.s.n_in = n_in; L.s.ip = 0; L.s.bb = 0; L.s.bc = 0; L.s.err = 0; L.s.nw = 0;
  L.op = 0; L.lit = 0; L.ntok = 0; L.last = 0; L.rc = CM_INF_OK;
  L.phase = active ? CM_INF_PH_HEADER : CM_INF_PH_DONE;
  const uint32_t cap = cm_inf_tok_cap(n_out);
  if (active) cm_inf_prime(L.s);
  while (CM_INF_ANY(L.phase != CM_INF_PH_DONE)) {
    if (L.phase == CM_INF_PH_HEADER) cm_inf_header(L, out, n_out, tok, cap, sym, len8, delta, stride);
    if (CM_INF_ANY(L.phase == CM_INF_PH_SYMBOLS)) {
      if (L.phase == CM_INF_PH_SYMBOLS) cm_inf_symbols(L, out, n_out, tok, cap, sym, delta, stride, max_steps);
    }
  }
  if (L.rc == CM_INF_OK && L.op != n_out) L.rc = CM_INF_EOUTPUT;
  *n_tok = L.ntok;
  return L.rc;
}

// ---- pass 2: one wave (a group of cm_coop.h's kind: t, sync(), scan(), rank(), ballot(), bcast(); G <= 64), one block's text in `win` ----
// The matches of tok[0 .. n_tok) copied inside win[0 .. isize) (the literals are in place), G tokens at a time: their places by a
// prefix sum; a match waits for the matches of its group whose output its source overlaps (found by bisection over the group's
// output starts and ends -- everything before the group is complete), so most of a group is copied side by side, a lane per match of
// up to 8 bytes (its source bytes read into a register, written back repeated if the match overlaps its own output) or up to 16
// that do not overlap themselves, the lanes together on a longer one.  What is left is the chains of matches that feed on each other: a round per link.
// ends: 2 G words of the group's shared memory; win: isize + 16 bytes.  Returns CM_INF_OK or CM_INF_ECODE (a token that reaches outside the block: pass 1
// writes none).
template <class GT>
CM_HD int cm_bgzf_resolve(GT &g, uint8_t *win, const uint32_t *tok, uint32_t n_tok, uint32_t isize, uint32_t *ends) {
  const uint32_t G = (uint32_t)GT::G;
  const uint64_t all = G >= 64 ? ~0ull : (1ull << (G & 63u)) - 1ull;
  uint32_t pos = 0;
  uint32_t v = g.t < n_tok ? tok[g.t] : 0u;
  for (uint32_t base = 0; base < n_tok; base += G) {
    const uint32_t i = base + g.t;
    const uint32_t lb = v & 511u;
    const bool skip = lb == CM_INF_TOK_SKIP;
    const uint32_t len = i < n_tok && !skip ? ((v >> 9) & 255u) + 3u : 0u, lits = i < n_tok ? lb : 0u, dist = (v >> 17) + 1u;
    uint32_t total, n_bad;
    const uint32_t dst = pos + g.scan(lits + len, &total) + lits;
    (void)g.rank(len != 0 && (dist > dst || dst > isize || len > isize - dst), &n_bad);
    if (n_bad != 0 || total > isize - pos) return CM_INF_ECODE;
    const uint32_t src = dst - dist;
    ends[g.t] = i < n_tok ? dst + len : 0xffffffffu;  // (ascending; a token without a match starts and ends where its literals end)
    ends[G + g.t] = i < n_tok ? dst : 0xffffffffu;
    g.sync();
    // (the next group's tokens: asked for here, after this group's have arrived, on their way while this group is copied)
    const uint32_t v_next = i + G < n_tok ? tok[i + G] : 0u;
    // a source that ends before the group's first byte waits for nothing; another one for the group's matches [a, b) whose output
    // overlaps it: a tokens end at or before its first byte, b tokens start before its end (two bisections side by side)
    uint32_t a = g.t, b = g.t;
    const uint32_t src_end = src + (dist < len ? dist : len);
    if (len != 0 && src_end > pos) {
      a = 0;
      b = 0;
      for (uint32_t step = G >> 1; step; step >>= 1) {
        const uint32_t x = ends[a + step - 1u], y = ends[G + b + step - 1u];
        if (x <= src) a += step;
        if (y < src_end) b += step;
      }
      if (b > g.t) b = g.t;
      if (a > b) a = b;
    }
    const uint64_t dep = ((1ull << b) - 1ull) & ~((1ull << a) - 1ull);
    bool done = len == 0;
    for (;;) {
      const uint64_t fin = g.ballot(done);
      if (fin == all) break;
      const bool ready = !done && (~fin & dep) == 0;  // (the first unfinished match always is)
      const bool wide = ready && (len > 16 || (len > 8 && dist < len));
      if (ready && !wide) {
        // (8 source bytes whatever the length, 8 stores whatever the length -- the ones past the match go to the pad behind the
        // text: loads that wait for each other and branches cost more than bytes nobody uses; win has 16 bytes of room behind isize)
        const uint32_t ns = dist < len ? dist : len;
        uint64_t p = 0;
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) p |= (uint64_t)win[src + k] << (8u * k);
        p &= ns >= 8 ? ~0ull : (1ull << (8u * ns)) - 1ull;
        if (dist < len) {  // the first `dist` bytes again and again (distance <= 7 here: three doublings fill the 8 bytes)
          uint32_t sh = 8u * dist;
          p |= p << sh;
          sh <<= 1;
          p |= sh < 64 ? p << sh : 0ull;
          sh <<= 1;
          p |= sh < 64 ? p << sh : 0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) win[k < len ? dst + k : isize + 8u] = (uint8_t)(p >> (8u * k));
      }
      if (g.ballot(ready && !wide && len > 8) != 0) {  // bytes 8..15 of the matches of 9..16 bytes that do not overlap themselves
        if (ready && !wide && len > 8) {
          uint64_t q = 0;
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) q |= (uint64_t)win[src + 8u + k] << (8u * k);
#pragma unroll
          for (uint32_t k = 0; k < 8; ++k) win[8u + k < len ? dst + 8u + k : isize + 8u] = (uint8_t)(q >> (8u * k));
        }
      }
      for (uint64_t w = g.ballot(wide); w; w &= w - 1ull) {
        const uint32_t t = (uint32_t)__builtin_ctzll(w);
        const uint32_t d = g.bcast(dst, t), l = g.bcast(len, t), ds = g.bcast(dist, t);
        if (ds >= l) {
          for (uint32_t j = g.t; j < l; j += G) win[d + j] = win[d - ds + j];
        } else {
          for (uint32_t j = g.t; j < l; j += G) win[d + j] = win[d - ds + (ds == 1 ? 0u : j % ds)];
        }
      }
      done = done || ready;
      g.sync();  // this round's bytes before the next round reads them
    }
    pos += total;
    v = v_next;
  }
  return CM_INF_OK;
}

// multiplication modulo the CRC-32 polynomial, and x^(n * 2^k) (the arithmetic of zlib's crc32_combine); x2n[i] = x^(2^i)
CM_HD uint32_t cm_crc_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (uint32_t m = 1u << 31; m; m >>= 1) {
    if (a & m) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
  }
  return p;
}
struct CmCrcX2n { uint32_t v[32]; };
CM_HD void cm_crc_x2n_table(CmCrcX2n &x) {
  x.v[0] = 1u << 30;  // x^1
  for (int i = 1; i < 32; ++i) x.v[i] = cm_crc_mulmod(x.v[i - 1], x.v[i - 1]);
}
// x^(8 n): what multiplies the CRC-32 of a text when n more bytes follow it (crc32(A || B) = crc32(A) * x^(8 |B|) ^ crc32(B))
CM_HD uint32_t cm_crc_pow(uint32_t n_bytes, const uint32_t *x2n) {
  uint32_t p = 1u << 31;  // x^0
  uint32_t k = 3;         // bytes -> bits
  for (uint32_t n = n_bytes; n; n >>= 1, ++k)
    if (n & 1u) p = cm_crc_mulmod(x2n[k & 31u], p);
  return p;
}
// CRC-32 of win[0 .. isize) by the group: four consecutive slices per lane (four independent table walks in flight) joined in the
// lane, the lanes' CRCs shifted to the end of the text and XORed.  red: G words of shared memory, crc_tab: the 256-entry table,
// x2n: 32 words.
template <class GT>
CM_HD uint32_t cm_bgzf_crc(GT &g, const uint8_t *win, uint32_t isize, const uint32_t *crc_tab, const uint32_t *x2n, uint32_t *red) {
  const uint32_t G = (uint32_t)GT::G;
  const uint32_t S = (isize + 4u * G - 1u) / (4u * G);  // slice length
  uint32_t c[4], b[4], e[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t sl = g.t * 4u + (uint32_t)q;
    b[q] = sl * S < isize ? sl * S : isize;
    e[q] = b[q] + S < isize ? b[q] + S : isize;
    c[q] = 0xffffffffu;
  }
  for (uint32_t i = 0; i < S; ++i) {  // (no branches: the four walks' loads overlap)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool in = b[q] + i < e[q];
      const uint32_t x = win[in ? b[q] + i : 0u];
      const uint32_t nc = crc_tab[(c[q] ^ x) & 0xffu] ^ (c[q] >> 8);
      c[q] = in ? nc : c[q];
    }
  }
  const uint32_t pS = cm_crc_pow(S, x2n);
  uint32_t acc = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t n = e[q] - b[q];
    if (n) acc = cm_crc_mulmod(n == S ? pS : cm_crc_pow(n, x2n), acc) ^ c[q] ^ 0xffffffffu;
  }
  red[g.t] = cm_crc_mulmod(cm_crc_pow(isize - e[3], x2n), acc);
  g.sync();
  uint32_t all = 0;
  for (uint32_t i = 0; i < G; ++i) all ^= red[i];
  g.sync();
  return all;
}

#endif
