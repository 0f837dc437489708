// cm_stages.h -- per-item stage functions of the mapping pipeline.
//
// Every function handles ONE item (pair, read, minimizer, task chunk) and is called from
// a thin __global__ wrapper in cm_kernels.hip (one thread per item).  Semantics follow the
// reference file:line cited at each function (paths relative to /root/reference/src);
// integer wrap-arounds and truncations are deliberate.  The same header compiles with
// g++ for tests/hostemu (test infrastructure; the library has no CPU path).
#ifndef CM_STAGES_H_
#define CM_STAGES_H_

#include "cm_types.h"

// ---------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------
// CharToUint8 (utils.h:87-104)
// branch-free: A C G T are 0x41 0x43 0x47 0x54 once the case bit is cleared; bits 1-2 of those codes are
// 00 01 11 10, Gray-decoded to 0 1 2 3; membership through a 20-bit set indexed by (u - 'A')
CM_HD uint32_t cm_c2u(uint8_t c) {
  const uint32_t u = c & 0xDFu;
  const uint32_t x = (u >> 1) & 3u;
  const uint32_t o = u - 0x41u;
  const bool acgt = o < 20u && ((0x80045u >> (o & 31u)) & 1u);
  return acgt ? x ^ (x >> 1) : 4u;
}
// Uint8ToChar(3 ^ CharToUint8(c)) as used by PrepareNegativeSequenceAt (sequence_batch.h:123-134)
CM_HD uint8_t cm_negchar(uint8_t c) {
  const uint32_t v = 3u ^ cm_c2u(c);
  return v == 0 ? 'A' : v == 1 ? 'C' : v == 2 ? 'G' : v == 3 ? 'T' : 'N';
}

// Hash64 (utils.h:76-85)
CM_HD uint64_t cm_hash64(uint64_t key, uint64_t mask) {
  key = (~key + (key << 21)) & mask;
  key = key ^ key >> 24;
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ key >> 14;
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ key >> 28;
  key = (key + (key << 31)) & mask;
  return key;
}

// Hash64 (utils.h:76-85) for 32 < 2k <= 52 bits on a (low word, high bits) pair: every step of the hash is a multiplication
// by a small odd constant or an xor with a right shift, both of which split cleanly -- the products' upper halves carry
// into the high bits, the shifts funnel high bits into the low word.  The 64-bit formulation costs six 64-bit multiply-adds
// per call (the compiler turns its shift-adds into them); this one two 32 x 32 -> 64 products.  hm = mask of the high bits.
CM_HD uint64_t cm_hash64_split(uint32_t lo, uint32_t hi, uint32_t hm) {
  // key = (~key + (key << 21)) & mask
  {
    const uint64_t t = (uint64_t)(uint32_t)~lo + (uint64_t)(uint32_t)(lo << 21);
    hi = ((~hi) + ((hi << 21) | (lo >> 11)) + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  // key ^= key >> 24
  lo ^= (lo >> 24) | (hi << 8);
  hi ^= hi >> 24;
  // key = key * 265 & mask
  {
    const uint64_t t = (uint64_t)lo * 265u;
    hi = (hi * 265u + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  // key ^= key >> 14
  lo ^= (lo >> 14) | (hi << 18);
  hi ^= hi >> 14;
  // key = key * 21 & mask
  {
    const uint64_t t = (uint64_t)lo * 21u;
    hi = (hi * 21u + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  // key ^= key >> 28
  lo ^= (lo >> 28) | (hi << 4);
  hi ^= hi >> 28;
  // key = (key + (key << 31)) & mask
  {
    const uint64_t t = (uint64_t)lo + (uint64_t)(uint32_t)(lo << 31);
    hi = (hi + ((hi << 31) | (lo >> 1)) + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  return ((uint64_t)hi << 32) | lo;
}
// The same when at most two bits lie above the low word (k = 17, the default): the odd multipliers are 1 modulo 4, so the
// high bits pass through the multiplications unchanged (their own products -- two quarter-rate 32-bit multiplies per call in
// the general form -- disappear), their right shifts vanish, and their left shifts by 21 and 31 fall outside the mask.
CM_HD uint64_t cm_hash64_split2(uint32_t lo, uint32_t hi, uint32_t hm) {
  {
    const uint64_t t = (uint64_t)(uint32_t)~lo + (uint64_t)(uint32_t)(lo << 21);
    hi = ((~hi) + (lo >> 11) + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  lo ^= (lo >> 24) | (hi << 8);
  {
    const uint64_t t = (uint64_t)lo * 265u;
    hi = (hi + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  lo ^= (lo >> 14) | (hi << 18);
  {
    const uint64_t t = (uint64_t)lo * 21u;
    hi = (hi + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  lo ^= (lo >> 28) | (hi << 4);
  {
    const uint64_t t = (uint64_t)lo + (uint64_t)(uint32_t)(lo << 31);
    hi = (hi + (lo >> 1) + (uint32_t)(t >> 32)) & hm;
    lo = (uint32_t)t;
  }
  return ((uint64_t)hi << 32) | lo;
}
// the same with the choice made by the caller once (HV 0: k = 17, 1: 2k in 33..52, 2: any k) instead of per call
template <int HV>
CM_HD uint64_t cm_hash64_v(uint64_t key, uint64_t mask) {
  if (HV == 0) return cm_hash64_split2((uint32_t)key, (uint32_t)(key >> 32), 3u);
  if (HV == 1) return cm_hash64_split((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(mask >> 32));
  return cm_hash64(key, mask);
}
CM_HD uint64_t cm_hash64_k(uint64_t key, int k, uint64_t mask) {
  if (2 * k == 34) return cm_hash64_split2((uint32_t)key, (uint32_t)(key >> 32), 3u);
  if (2 * k > 32 && 2 * k <= 52) return cm_hash64_split((uint32_t)key, (uint32_t)(key >> 32), (uint32_t)(mask >> 32));
  return cm_hash64(key, mask);
}

CM_HD const uint8_t *cm_read_ptr(const CmDev &d, uint32_t r) {
  const uint32_t pair = r >> 1;
  return (r & 1) ? d.rb1 + d.ro1[pair] : d.rb0 + d.ro0[pair];
}
CM_HD uint32_t cm_raw_len(const CmDev &d, uint32_t r) {
  const uint32_t pair = r >> 1;
  return (r & 1) ? d.ro1[pair + 1] - d.ro1[pair] : d.ro0[pair + 1] - d.ro0[pair];
}

// in-place ascending sort of a[0], a[st], ..., a[(n-1)*st] -- insertion sort for short lists,
// heap sort otherwise.  st > 1 is the LDS layout [entry][thread] (conflict-free across lanes).
CM_HD void cm_sort_u64_strided(uint64_t *a, uint32_t n, uint32_t st) {
  if (n < 2) return;
  // Lists of up to 32 hits: insertion sort (the hits of the true locus share one diagonal, so the list is mostly
  // in order and one pass does it); it gives up after 6n moves and the heap sort below takes over -- the array is
  // a permutation of the input at every point.  Longer lists go to the heap sort directly (2 x 100 reads:
  // k_s3b 6.3 -> 5.3 ms with the insertion pass; no gain for the ~37-hit lists of 2 x 150 reads).
  if (n <= 32) {
    uint32_t moves = 0;
    const uint32_t budget = n <= 24 ? ~0u : 6 * n;
    uint32_t i = 1;
    for (; i < n && moves <= budget; ++i) {
      const uint64_t x = a[i * st];
      uint32_t j = i;
      while (j > 0 && a[(j - 1) * st] > x) { a[j * st] = a[(j - 1) * st]; --j; ++moves; }
      a[j * st] = x;
    }
    if (i == n) return;
  }
  for (uint32_t start = n / 2; start-- > 0;) {
    uint32_t root = start;
    const uint64_t x = a[root * st];
    for (;;) {
      uint32_t child = 2 * root + 1;
      if (child >= n) break;
      if (child + 1 < n && a[child * st] < a[(child + 1) * st]) ++child;
      if (!(x < a[child * st])) break;
      a[root * st] = a[child * st];
      root = child;
    }
    a[root * st] = x;
  }
  for (uint32_t end = n - 1; end > 0; --end) {
    const uint64_t x = a[end * st];
    a[end * st] = a[0];
    uint32_t root = 0;
    for (;;) {
      uint32_t child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && a[child * st] < a[(child + 1) * st]) ++child;
      if (!(x < a[child * st])) break;
      a[root * st] = a[child * st];
      root = child;
    }
    a[root * st] = x;
  }
}
CM_HD void cm_sort_u64(uint64_t *a, uint32_t n) { cm_sort_u64_strided(a, n, 1); }

// Candidate::operator< (candidate.h:22-33): count desc, position asc.  "a before b"
CM_HD bool cm_cand_before(uint64_t pa, uint8_t ca, uint64_t pb, uint8_t cb) {
  if (ca != cb) return ca > cb;
  return pa < pb;
}
// in-place sort of (pos[], cnt[]) by cm_cand_before (MappingMetadata::SortCandidates)
CM_HD void cm_sort_cand(uint64_t *p, uint8_t *c, uint32_t n) {
  if (n < 2) return;
  if (n <= 24) {
    for (uint32_t i = 1; i < n; ++i) {
      const uint64_t xp = p[i];
      const uint8_t xc = c[i];
      uint32_t j = i;
      while (j > 0 && cm_cand_before(xp, xc, p[j - 1], c[j - 1])) { p[j] = p[j - 1]; c[j] = c[j - 1]; --j; }
      p[j] = xp;
      c[j] = xc;
    }
    return;
  }
  {  // long lists are sorted beforehand by a wave each (k_sort_lists): nothing left to do then
    bool sorted = true;
    for (uint32_t i = 1; i < n && sorted; ++i) sorted = !cm_cand_before(p[i], c[i], p[i - 1], c[i - 1]);
    if (sorted) return;
  }
  // heap sort with "after" as the heap order (max-heap on the sort order)
  for (uint32_t start = n / 2; start-- > 0;) {
    uint32_t root = start;
    const uint64_t xp = p[root];
    const uint8_t xc = c[root];
    for (;;) {
      uint32_t child = 2 * root + 1;
      if (child >= n) break;
      if (child + 1 < n && cm_cand_before(p[child], c[child], p[child + 1], c[child + 1])) ++child;
      if (!cm_cand_before(xp, xc, p[child], c[child])) break;
      p[root] = p[child]; c[root] = c[child];
      root = child;
    }
    p[root] = xp; c[root] = xc;
  }
  for (uint32_t end = n - 1; end > 0; --end) {
    const uint64_t xp = p[end];
    const uint8_t xc = c[end];
    p[end] = p[0]; c[end] = c[0];
    uint32_t root = 0;
    for (;;) {
      uint32_t child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && cm_cand_before(p[child], c[child], p[child + 1], c[child + 1])) ++child;
      if (!cm_cand_before(xp, xc, p[child], c[child])) break;
      p[root] = p[child]; c[root] = c[child];
      root = child;
    }
    p[root] = xp; c[root] = xc;
  }
}

// in-place sort of draft mappings (pos[], err[]) by position (SortMappingsByPositions,
// mapping_metadata.h:70-78).  Order among equal positions does not affect results.
CM_HD void cm_sort_draft(uint64_t *p, int16_t *e, uint32_t n) {
  if (n < 2) return;
  if (n <= 24) {
    for (uint32_t i = 1; i < n; ++i) {
      const uint64_t xp = p[i];
      const int16_t xe = e[i];
      uint32_t j = i;
      while (j > 0 && p[j - 1] > xp) { p[j] = p[j - 1]; e[j] = e[j - 1]; --j; }
      p[j] = xp;
      e[j] = xe;
    }
    return;
  }
  {  // long lists are sorted beforehand by a wave each (k_sort_lists)
    bool sorted = true;
    for (uint32_t i = 1; i < n && sorted; ++i) sorted = !(p[i] < p[i - 1]);
    if (sorted) return;
  }
  for (uint32_t start = n / 2; start-- > 0;) {
    uint32_t root = start;
    const uint64_t xp = p[root];
    const int16_t xe = e[root];
    for (;;) {
      uint32_t child = 2 * root + 1;
      if (child >= n) break;
      if (child + 1 < n && p[child] < p[child + 1]) ++child;
      if (!(xp < p[child])) break;
      p[root] = p[child]; e[root] = e[child];
      root = child;
    }
    p[root] = xp; e[root] = xe;
  }
  for (uint32_t end = n - 1; end > 0; --end) {
    const uint64_t xp = p[end];
    const int16_t xe = e[end];
    p[end] = p[0]; e[end] = e[0];
    uint32_t root = 0;
    for (;;) {
      uint32_t child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && p[child] < p[child + 1]) ++child;
      if (!(xp < p[child])) break;
      p[root] = p[child]; e[root] = e[child];
      root = child;
    }
    p[root] = xp; e[root] = xe;
  }
}


// ---------------------------------------------------------------------------------------
// K6: barcode correction, one pair per item (Chromap::CorrectBarcodeAt, chromap.cc:572-799,
//     GenerateSeedFromSequence utils.h:111-129).  bc_error_threshold 0 or 1.  The corrected
//     barcode is kept as its 2-bit key (what the record carries); the bases in HBM are
//     not rewritten.  Candidate scores are pow(10,-q/10) (host table) * count/num_sample in
//     double; several candidates are sorted descending by (score, index, base) and summed in
//     that order like the reference (:733-740).
// ---------------------------------------------------------------------------------------
CM_HD uint64_t cm_seed_from_sequence(const uint8_t *seq, uint32_t len) {
  uint64_t seed = 0;
  for (uint32_t i = 0; i < len; ++i) {
    const uint32_t b = cm_c2u(seq[i]);
    seed = b < 4 ? (seed << 2) | b : seed << 2;
  }
  return seed;
}
// whitelist lookup; returns count via *cnt
CM_HD bool cm_wl_get(const CmDev &d, uint64_t key, uint32_t *cnt) {
  const uint64_t x = key * 0x9E3779B97F4A7C15ull;
  uint32_t i = (uint32_t)(x >> 32) & d.wl_mask;
  for (;;) {
    const uint64_t k = d.wl[2 * (uint64_t)i];
    if (k == ~0ull) return false;
    if (k == key) { *cnt = (uint32_t)d.wl[2 * (uint64_t)i + 1]; return true; }
    i = (i + 1) & d.wl_mask;
  }
}
#define CM_BC_MAXC 132  // buffered candidates; more than that takes the selection path below
CM_HD int cm_bc_qual(const uint8_t *q, uint32_t len, uint32_t i) {
  int aq = (int)q[len - 1 - i] - 33;
  aq = aq > 40 ? 40 : aq;
  return aq < 3 ? 3 : aq;
}
// Enumerates the whitelisted barcodes within bc_err substitutions in the reference's order
// (Chromap::CorrectBarcodeAt, chromap.cc:590-718) and calls f(score, key, tag) for each;
// tag = idx1<<24 | base1<<16 | idx2<<8 | base2 orders like BarcodeWithQual (utils.h:23-35).
template <class F>
CM_HD void cm_bc_enumerate(const CmDev &d, uint64_t key, const uint8_t *q, uint32_t len, int nn, int n0, int n1, F &&f) {
  uint32_t i_start = 0, i_end = len, ti_limit = 3;
  if (nn > 0) { i_start = (uint32_t)n0; i_end = i_start + 1; ti_limit = 4; }
  for (uint32_t i = i_start; i < i_end; ++i) {
    const uint64_t cleared = ~(3ull << (2 * i)) & key;
    uint64_t b1 = (key >> (2 * i)) & 3ull;
    for (uint32_t ti = 0; ti < ti_limit; ++ti) {
      b1 = (b1 + 1) & 3ull;
      const uint64_t k1 = cleared | (b1 << (2 * i));
      uint32_t c1;
      if (cm_wl_get(d, k1, &c1))
        f(d.pow10_tab[cm_bc_qual(q, len, i)] * (c1 / d.wl_num_sample), k1, ((len - 1 - i) << 24) | ((uint32_t)"ACGT"[b1] << 16));
      if (d.p.bc_err == 2) {
        uint32_t j_start = i + 1, j_end = len, ti2_limit = 3;
        if (nn == 2) { j_start = (uint32_t)n1; j_end = j_start + 1; ti2_limit = 4; }
        for (uint32_t j = j_start; j < j_end; ++j) {
          const uint64_t cleared2 = ~(3ull << (2 * j)) & k1;
          uint64_t b2 = (k1 >> (2 * j)) & 3ull;
          for (uint32_t ti2 = 0; ti2 < ti2_limit; ++ti2) {
            b2 = (b2 + 1) & 3ull;
            const uint64_t k2 = cleared2 | (b2 << (2 * j));
            uint32_t c2;
            if (cm_wl_get(d, k2, &c2))
              f(d.pow10_tab[cm_bc_qual(q, len, j) + cm_bc_qual(q, len, i)] * (c2 / d.wl_num_sample), k2,
                ((len - 1 - i) << 24) | ((uint32_t)"ACGT"[b1] << 16) | ((len - 1 - j) << 8) | (uint32_t)"ACGT"[b2]);
          }
        }
      }
    }
  }
}

CM_HD void cm_s0b_barcode(const CmDev &d, uint32_t pair, uint32_t *in_wl, uint32_t *corrected) {
  const uint8_t *bc = d.bcb + d.bco[pair];
  const uint8_t *q = d.bcq + d.bco[pair];
  const uint32_t len = d.bco[pair + 1] - d.bco[pair];
  const uint64_t key = cm_seed_from_sequence(bc, len);
  d.bc_key[pair] = key;
  if (!d.wl) { d.bc_ok[pair] = 1; return; }  // no whitelist given: the barcode is taken as read (chromap.h:897-903)
  d.bc_ok[pair] = 0;
  uint32_t cnt = 0;
  const bool found = cm_wl_get(d, key, &cnt);
  int nn = 0, n0 = 0, n1 = 0;
  for (int i = (int)len - 1; i >= 0; --i)
    if (bc[i] == 'N') {  // GetSequenceNsAt: 'N' only, little-endian positions
      if (nn == 0) n0 = (int)len - 1 - i;
      if (nn == 1) n1 = (int)len - 1 - i;
      ++nn;
    }
  if ((uint32_t)nn > (uint32_t)d.p.bc_err) return;
  if (nn == 0 && found) { ++*in_wl; d.bc_ok[pair] = 1; return; }
  if (d.p.bc_err <= 0) return;
  double sc[CM_BC_MAXC];
  uint64_t ck[CM_BC_MAXC];
  uint32_t ci[CM_BC_MAXC];
  uint32_t nc = 0;
  cm_bc_enumerate(d, key, q, len, nn, n0, n1, [&](double score, uint64_t k, uint32_t tag) {
    if (nc < CM_BC_MAXC) { sc[nc] = score; ck[nc] = k; ci[nc] = tag; }
    ++nc;
  });
  if (nc == 0) return;
  uint64_t best_key = ck[0];
  bool apply = true;
  if (nc > 1 && nc <= CM_BC_MAXC) {
    // insertion sort, descending by (score, tag): std::greater<BarcodeWithQual>; the sum runs in
    // that order (double addition is not associative)
    for (uint32_t a = 1; a < nc; ++a) {
      const double xs = sc[a];
      const uint64_t xk = ck[a];
      const uint32_t xi = ci[a];
      uint32_t b = a;
      while (b > 0 && (sc[b - 1] < xs || (sc[b - 1] == xs && ci[b - 1] < xi))) { sc[b] = sc[b - 1]; ck[b] = ck[b - 1]; ci[b] = ci[b - 1]; --b; }
      sc[b] = xs; ck[b] = xk; ci[b] = xi;
    }
    double sum = 0;
    for (uint32_t a = 0; a < nc; ++a) sum += sc[a];
    apply = sc[0] / sum > d.p.bc_prob;
    best_key = ck[0];
  } else if (nc > CM_BC_MAXC) {
    // dense whitelists (short barcodes): selection instead of a sort -- nc passes over the
    // enumeration, each picking the next candidate in descending order
    double prev_s = 0, sum = 0, first_s = 0;
    uint32_t prev_t = 0;
    for (uint32_t r = 0; r < nc; ++r) {
      double bs = -1.0;
      uint32_t bt = 0;
      uint64_t bk = 0;
      cm_bc_enumerate(d, key, q, len, nn, n0, n1, [&](double score, uint64_t k, uint32_t tag) {
        if (r > 0 && !(score < prev_s || (score == prev_s && tag < prev_t))) return;
        if (score > bs || (score == bs && tag > bt)) { bs = score; bt = tag; bk = k; }
      });
      if (r == 0) { first_s = bs; best_key = bk; }
      sum += bs;
      prev_s = bs;
      prev_t = bt;
    }
    apply = first_s / sum > d.p.bc_prob;
  }
  if (apply) {
    d.bc_key[pair] = best_key;
    d.bc_ok[pair] = 1;
    ++*corrected;
  }
}

// ---------------------------------------------------------------------------------------
// S0: length filter + adapter trimming, one pair per item
//   chromap.h:911-924, Chromap::TrimAdapterForPairedEndRead chromap.cc:176-289,
//   SequenceBatch::TrimSequenceAt sequence_batch.h:136-151
// After trimming, the stored reverse complement of a read is exactly the reverse
// complement of its first new_len bases (the front of the revcomp string is erased), so
// only the new lengths need to be kept.
// ---------------------------------------------------------------------------------------
// Fast path of the overlap search for reads made of upper-case A/C/G/T only (anything else takes
// the byte-wise loop below, which is the definition).  With 2-bit codes the reference's search
//   for si in {0,1}: for every position sp where seed si of read1 occurs in revcomp(read2):
//       accept if the whole overlap has <= 1 mismatch          (chromap.cc:200-260)
// becomes: for every alignment s0 = sp - si*seed in [0, l2 - min_overlap], D(s0) = mismatch bits of
// read1[0,n) vs neg2[s0,s0+n), n = min(l1, l2-s0); alignment s0 is accepted through seed 0 if
// popcount(D) <= 1 and D is zero on [0,seed), through seed 1 if zero on [seed,2*seed).  The
// result is the smallest s0 accepted through seed 0, else the smallest accepted through seed 1
// -- the order in which the reference tries them.  Code: (c>>1)&3 (A0 C1 T2 G3), complement = ^2.
CM_HD bool cm_trim_pack(const uint8_t *s, uint32_t len, bool revcomp, uint64_t *w, int nw) {
  for (int i = 0; i < nw; ++i) w[i] = 0;
  bool valid = true;
  for (uint32_t i = 0; i < len; ++i) {
    const uint32_t c = s[revcomp ? len - 1 - i : i];
    const uint32_t o = c - 'A';
    valid = valid && o < 26u && ((0x00080045u >> o) & 1u);  // A, C, G, T
    const uint64_t code = ((c >> 1) & 3u) ^ (revcomp ? 2u : 0u);
    w[i >> 5] |= code << ((i & 31) << 1);
  }
  return valid;
}
template <int W>
CM_HD bool cm_trim_fast(const uint8_t *rd1, uint32_t l1, const uint8_t *lng, uint32_t l2, int min_overlap, int seed,
                        bool *found, uint32_t *s0_out) {
  uint64_t a[W], b[W];
  const bool va = cm_trim_pack(rd1, l1, false, a, W), vb = cm_trim_pack(lng, l2, true, b, W);
  if (!va || !vb) return false;
  const uint64_t m55 = 0x5555555555555555ull;
  const uint64_t seedmask = seed >= 32 ? ~0ull : ((1ull << (2 * seed)) - 1);
  bool fa = false, fb = false;
  uint32_t sa = 0, sb = 0;
  const uint32_t last = l2 - (uint32_t)min_overlap;
  for (uint32_t s0 = 0; s0 <= last && !fa; ++s0) {
    const uint32_t n = l1 < l2 - s0 ? l1 : l2 - s0;
    uint32_t cnt = 0;
    uint64_t d0 = 0, d1 = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const uint64_t x = a[w] ^ b[w];
      uint64_t dd = (x | (x >> 1)) & m55;
      const uint32_t lo = 32u * (uint32_t)w;
      const uint64_t m = n >= lo + 32 ? ~0ull : n <= lo ? 0ull : ((1ull << (2 * (n - lo))) - 1);
      dd &= m;
      cnt += (uint32_t)__builtin_popcountll(dd);
      if (w == 0) d0 = dd;
      if (w == 1) d1 = dd;
    }
    const uint64_t r0 = d0 & seedmask;
    const uint64_t r1 = seed >= 32 ? d1 : (((d0 >> (2 * seed)) | (d1 << (64 - 2 * seed))) & seedmask);
    const bool ok = cnt <= 1;
    if (ok && r0 == 0) { fa = true; sa = s0; }
    if (ok && r1 == 0 && !fb) { fb = true; sb = s0; }
#pragma unroll
    for (int w = 0; w + 1 < W; ++w) b[w] = (b[w] >> 2) | (b[w + 1] << 62);
    b[W - 1] >>= 2;
  }
  *found = fa || fb;
  *s0_out = fa ? sa : sb;
  return true;
}

// s1p / s2p: the pair's reads (global memory, or the block's LDS copy of them)
CM_HD void cm_s0_prep_ptr(const CmDev &d, uint32_t pair, const uint8_t *s1p, const uint8_t *s2p) {
  const uint32_t raw1 = d.ro0[pair + 1] - d.ro0[pair], raw2 = d.ro1[pair + 1] - d.ro1[pair];
  uint32_t len1 = raw1, len2 = raw2;
  bool ok = raw1 >= (uint32_t)d.p.min_read_len && (d.p.single || raw2 >= (uint32_t)d.p.min_read_len);
  // pairs whose barcode is not (correctable to) a whitelisted one are skipped (chromap.h:908-909)
  if (d.bcb && !d.bc_ok[pair] && !d.p.bc_keep) ok = false;
  if (ok && d.p.trim && !d.p.single) {
    const uint8_t *s1 = s1p, *s2 = s2p;
    const bool swap = !(raw1 <= raw2);
    const uint8_t *rd1 = swap ? s2 : s1;        // "read1": the shorter read, forward
    const uint8_t *lng = swap ? s1 : s2;        // "read2": its revcomp is searched
    const uint32_t l1 = swap ? raw2 : raw1, l2 = swap ? raw1 : raw2;
    const int min_overlap = d.p.min_read_len;
    const int seed = min_overlap / 2;
    bool merged = false;
    // fast path (upper-case ACGT reads of <= 256 bases, seed <= 32): same result, 2-bit arithmetic
    bool fast = false, ffound = false;
    uint32_t fs0 = 0;
    if (seed >= 1 && seed <= 32 && 2 * seed <= min_overlap && l1 >= (uint32_t)min_overlap) {
      if (l2 <= 64) fast = cm_trim_fast<2>(rd1, l1, lng, l2, min_overlap, seed, &ffound, &fs0);
      else if (l2 <= 160) fast = cm_trim_fast<5>(rd1, l1, lng, l2, min_overlap, seed, &ffound, &fs0);
      else if (l2 <= 256) fast = cm_trim_fast<8>(rd1, l1, lng, l2, min_overlap, seed, &ffound, &fs0);
    }
    if (fast) {
      merged = true;  // skips the byte-wise search
      if (ffound) {
        int overlap = (int)(l2 - fs0);
        int off2 = 0;
        if (overlap > (int)l1) { off2 = overlap - (int)l1; overlap = (int)l1; }
        const int t1 = swap ? overlap + off2 : overlap;
        const int t2 = swap ? overlap : overlap + off2;
        if (t1 < (int)raw1) len1 = (uint32_t)t1;
        if (t2 < (int)raw2) len2 = (uint32_t)t2;
      }
    }
    // neg2[i] = cm_negchar(lng[l2 - 1 - i])
    for (int si = 0; si < 2 && !merged; ++si) {
      const uint8_t *needle = rd1 + si * seed;
      for (uint32_t sp = 0; sp + (uint32_t)seed <= l2 && !merged; ++sp) {
        // std::string::find: first position >= sp where the seed matches
        bool hit = true;
        for (int j = 0; j < seed; ++j)
          if (cm_negchar(lng[l2 - 1 - (sp + j)]) != needle[j]) { hit = false; break; }
        if (!hit) continue;
        const bool before_ok = sp >= (uint32_t)(si * seed);
        const bool overlap_ok = (int)(l2 - sp + (uint32_t)(seed * si)) >= min_overlap;
        if (!before_ok || !overlap_ok) continue;
        bool can = true;
        int ne = 0;
        for (int i = 0; i < seed * si; ++i) {
          if (cm_negchar(lng[l2 - 1 - (sp - si * seed + i)]) != rd1[i]) ++ne;
          if (ne > 1) { can = false; break; }
        }
        if (can) {
          for (uint32_t i = (uint32_t)seed; i + sp < l2 && (uint32_t)(si * seed) + i < l1; ++i) {
            if (cm_negchar(lng[l2 - 1 - (sp + i)]) != rd1[si * seed + i]) ++ne;
            if (ne > 1) { can = false; break; }
          }
        }
        if (can) {
          int overlap = (int)(l2 - sp + (uint32_t)(si * seed));
          int off2 = 0;
          if (overlap > (int)l1) { off2 = overlap - (int)l1; overlap = (int)l1; }
          const int t1 = swap ? overlap + off2 : overlap;
          const int t2 = swap ? overlap : overlap + off2;
          if (t1 < (int)raw1) len1 = (uint32_t)t1;
          if (t2 < (int)raw2) len2 = (uint32_t)t2;
          merged = true;
        }
      }
    }
  }
  d.rlen[2 * pair] = ok ? len1 : 0;
  d.rlen[2 * pair + 1] = (ok && !d.p.single) ? len2 : 0;
}

CM_HD void cm_s0_prep(const CmDev &d, uint32_t pair) {
  cm_s0_prep_ptr(d, pair, d.rb0 + d.ro0[pair], d.rb1 + d.ro1[pair]);
}

// ---------------------------------------------------------------------------------------
// S1: minimizers of one read (MinimizerGenerator::GenerateMinimizers,
//     minimizer_generator.cc:7-139).  Writes (hash, (pos<<1)|strand) into the read's slot
//     range and the count.  The read's sequence index inside the hit is not kept: no
//     consumer on the mapping path looks at it (index.cc:491-505 uses position+strand).
// ---------------------------------------------------------------------------------------
#define CM_MAX_W 32
// THE minimizer state machine -- every caller that does not use the w = 7 closed form below runs this one
// (reads with a generic window, even k, and the reference chunks of index construction).
// The last w (hash, pos) entries are kept in chronological order (index 0 = oldest).  Equivalent to the
// reference's ring buffer: its scans `j = pib+1..w-1, then 0..pib` walk the ring oldest -> newest,
// "position_in_buffer == min_position" means the running minimum is the entry being evicted (mi < 0 after
// the shift), and a palindromic k-mer neither writes nor advances (:42-45).
//   WT > 0: the window size is a compile-time constant and the window lives in registers;
//   WT == 0: runtime w <= CM_MAX_W (private memory; the rare non-default index).
// Positions [begin, end) of seq are run; `flush` performs the final emission (:136-138).
// emit(n, hash, pos_strand) -> bool: n = emissions counted so far, return value = whether this one counts.
template <int WT, class Emit>
CM_HD uint32_t cm_minimizers_core(const uint8_t *seq, uint32_t begin, uint32_t end, bool flush, int k, int w, Emit &&emit) {
  constexpr int CAP = WT ? WT : CM_MAX_W;
  const int W = WT ? WT : w;
  uint32_t n = 0;
  const uint64_t shift = 2 * (uint64_t)(k - 1);
  const uint64_t mask = (((uint64_t)1) << (2 * k)) - 1;
  uint64_t fw = 0, rv = 0;
  uint64_t wh[CAP];
  uint32_t wp[CAP];
  uint64_t min_h = ~0ull;
  uint32_t min_p = ~0u;
#pragma unroll
  for (int i = 0; i < CAP; ++i) { wh[i] = ~0ull; wp[i] = ~0u; }
  int unamb = 0, mi = 0;
#define CM_EMIT(h, p) do { if (emit(n, (h), (p))) ++n; } while (0)
  for (uint32_t pos = begin; pos < end; ++pos) {
    const uint32_t c = cm_c2u(seq[pos]);
    uint64_t cur_h = ~0ull;
    uint32_t cur_p = ~0u;
    if (c < 4) {
      fw = ((fw << 2) | c) & mask;
      rv = (rv >> 2) | (((uint64_t)(3 ^ c)) << shift);
      if (fw == rv) continue;
      const uint64_t h0 = cm_hash64(fw, mask), h1 = cm_hash64(rv, mask);
      const uint32_t strand = h0 < h1 ? 0u : 1u;
      ++unamb;
      if (unamb >= k) {
        cur_h = cm_hash64(strand ? h1 : h0, mask);
        cur_p = (pos << 1) | strand;
      }
    } else {
      unamb = 0;
    }
#pragma unroll
    for (int j = 0; j < W - 1; ++j) { wh[j] = wh[j + 1]; wp[j] = wp[j + 1]; }
    wh[W - 1] = cur_h;
    wp[W - 1] = cur_p;
    --mi;
    if (unamb == W + k - 1 && min_h != ~0ull && min_h < cur_h) {
#pragma unroll
      for (int j = 0; j < W - 1; ++j)
        if (min_h == wh[j] && wp[j] != min_p) CM_EMIT(wh[j], wp[j]);
    }
    if (cur_h <= min_h) {
      if (unamb >= W + k && min_h != ~0ull) CM_EMIT(min_h, min_p);
      min_h = cur_h;
      min_p = cur_p;
      mi = W - 1;
    } else if (mi < 0) {
      if (unamb >= W + k - 1 && min_h != ~0ull) CM_EMIT(min_h, min_p);
      min_h = ~0ull;
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (min_h >= wh[j]) { min_h = wh[j]; min_p = wp[j]; mi = j; }
      if (unamb >= W + k - 1 && min_h != ~0ull) {
#pragma unroll
        for (int j = 0; j < W; ++j)
          if (min_h == wh[j] && min_p != wp[j]) CM_EMIT(wh[j], wp[j]);
      }
    }
  }
  if (flush && min_h != ~0ull) CM_EMIT(min_h, min_p);
#undef CM_EMIT
  return n;
}

// a whole read, window size W in registers; emit(n, hash, pos_strand)
template <int W, class Emit>
CM_HD uint32_t cm_minimizers_window_e(const uint8_t *seq, uint32_t len, int k, Emit &&emit) {
  return cm_minimizers_core<W>(seq, 0, len, true, k, W, [&](uint32_t n, uint64_t h, uint32_t p) { emit(n, h, p); return true; });
}

// w = 7, odd k (every preset; an odd-length k-mer cannot be its own reverse complement, so the palindrome
// branch never fires): the same emissions as cm_minimizers_window_e<7> without its data-dependent branches --
// on a wave every branch of the state machine is taken by some lane at nearly every position, so the
// wave pays for all of them (~300 instructions per base).  Closed form of what the state machine emits on
// a read without ambiguous bases, k-mers h[0..m-1]:
//   * k-mer j is emitted iff h[j] equals the minimum of some complete window h[i-6..i] containing it
//     (the running minimum is emitted when it is replaced or leaves, equal hashes in the window are
//     emitted by the duplicate scans), in increasing position;
//   * except at the first complete window: if h[6] equals the minimum of h[0..5], those earlier equal
//     k-mers are dropped (the replaced minimum is only emitted from unambiguous_length >= w + k on, and
//     the duplicate scan of that step needs min < h[6]); every later window that holds them also holds
//     k-mer 6, so they are simply struck out;
//   * fewer than 7 k-mers: only the final flush fires, with the last occurrence of the minimum.
// Per base: shift the 7-entry register window, one 7-way minimum, 7 equality tests into a flag mask,
// and the entry that leaves the window is emitted if flagged.  A read with a base outside ACGT is redone by
// the state machine (its behaviour across N runs is not a window rule).
template <class Emit>
CM_HD uint32_t cm_minimizers_w7_oddk_seq(const uint8_t *seq, uint32_t len, int k, Emit &&emit) {
  const uint64_t shift = 2 * (uint64_t)(k - 1);
  const uint64_t mask = (((uint64_t)1) << (2 * k)) - 1;
  uint64_t fw = 0, rv = 0;
  uint64_t H[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) H[j] = ~0ull;
  uint32_t sb = 0, fl = 0, bad = 0, n = 0;
  for (uint32_t pos = 0; pos < len; ++pos) {
    uint32_t c = cm_c2u(seq[pos]);
    bad |= c >> 2;
    c &= 3;
    fw = ((fw << 2) | c) & mask;
    rv = (rv >> 2) | (((uint64_t)(3 ^ c)) << shift);
    if (pos + 1 < (uint32_t)k) continue;
    const uint64_t h0 = cm_hash64_k(fw, k, mask), h1 = cm_hash64_k(rv, k, mask);
    const uint32_t strand = h0 < h1 ? 0u : 1u;
    const uint64_t h = cm_hash64_k(strand ? h1 : h0, k, mask);
    if (fl & 1u) { emit(n, H[0], ((pos - 7) << 1) | (sb & 1u)); ++n; }
#pragma unroll
    for (int j = 0; j < 6; ++j) H[j] = H[j + 1];
    H[6] = h;
    sb = (sb >> 1) | (strand << 6);
    fl >>= 1;
    const uint32_t i = pos + 1 - (uint32_t)k;  // index of this k-mer
    if (i == 6) {
      uint64_t v = H[0];
#pragma unroll
      for (int j = 1; j < 6; ++j) v = H[j] < v ? H[j] : v;
      if (h == v) {
#pragma unroll
        for (int j = 0; j < 6; ++j) H[j] = H[j] == v ? ~0ull : H[j];
      }
    }
    if (i >= 6) {
      uint64_t m = H[0];
#pragma unroll
      for (int j = 1; j < 7; ++j) m = H[j] < m ? H[j] : m;
#pragma unroll
      for (int j = 0; j < 7; ++j) fl |= H[j] == m ? 1u << j : 0u;
    }
  }
  if (bad) return ~0u;
  if (len < (uint32_t)k) return 0;
  const uint32_t m_k = len - (uint32_t)k + 1;
  if (m_k < 7) {  // no complete window: the flush emits the running minimum, ties -> the last one
    uint64_t v = ~0ull;
    int at = -1;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j >= 7 - (int)m_k && H[j] <= v) { v = H[j]; at = j; }
    if (at >= 0) { emit(n, v, ((len - 1 - (uint32_t)(6 - at)) << 1) | ((sb >> at) & 1u)); ++n; }
    return n;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j)
    if ((fl >> j) & 1u) { emit(n, H[j], ((len - 1 - (uint32_t)(6 - j)) << 1) | ((sb >> j) & 1u)); ++n; }
  return n;
}

// The same emissions with the selection done per BLOCK of seven k-mers instead of per k-mer (what the kernels run for reads of
// seven k-mers and more).  The form above recomputes, for every base, the minimum of its window (six 64-bit minima) and tests all
// seven entries against it (seven 64-bit compares): 13 two-word operations next to the three hashes.  Here, with the k-mers cut
// into blocks of seven (h padded with 0 beyond the last k-mer: a window that holds a pad has minimum 0 and so never counts):
//   window minima (van Herk / Gil-Werman): the window that starts at entry j of block c is the suffix of block c from j and the
//     prefix of block c + 1 before j:   M = min(suffix_min_c[j], prefix_min_{c+1}[j - 1]);
//   k-mer i is emitted  <=>  h[i] == max of the minima of the windows that hold it (the rule of k_prep_flat above), and those
//     windows are a suffix of the previous block's seven windows and a prefix of its own block's:
//                           F = max(suffix_max_{c-1}[j + 1], prefix_max_c[j]);
// 36 minima / maxima and 7 compares per seven k-mers -- 6.1 two-word operations per base instead of 13.  The first window's rule
// (the reference drops the earlier k-mers that tie with the seventh, see above) strikes those entries out (all ones) before
// anything else looks at them.  Block c is settled while block c + 1 is hashed; one empty block runs last.
template <int HV, class Emit>
CM_HD uint32_t cm_minimizers_w7_oddk_blocks(const uint8_t *seq, uint32_t len, int k, Emit &&emit) {
  const uint64_t shift = 2 * (uint64_t)(k - 1);
  const uint64_t mask = (((uint64_t)1) << (2 * k)) - 1;
  uint64_t fw = 0, rv = 0;
  uint32_t bad = 0, n = 0;
  for (uint32_t pos = 0; pos + 1 < (uint32_t)k; ++pos) {
    uint32_t c = cm_c2u(seq[pos]);
    bad |= c >> 2;
    c &= 3;
    fw = ((fw << 2) | c) & mask;
    rv = (rv >> 2) | (((uint64_t)(3 ^ c)) << shift);
  }
  const uint32_t m = len - (uint32_t)k + 1, nblk = (m + 6) / 7;
  uint64_t ep[7], Sp[7], SM[7];  // the previous block: hashes, their suffix minima; suffix maxima of the window minima before it
  uint32_t sbp = 0;
#pragma unroll
  for (int j = 0; j < 7; ++j) { ep[j] = 0; Sp[j] = 0; SM[j] = 0; }
  for (uint32_t b = 0; b <= nblk; ++b) {
    uint64_t e[7];
    uint32_t sb = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const uint32_t i = 7 * b + (uint32_t)j;
      e[j] = 0;
      if (i < m) {
        uint32_t c = cm_c2u(seq[i + (uint32_t)k - 1]);
        bad |= c >> 2;
        c &= 3;
        fw = ((fw << 2) | c) & mask;
        rv = (rv >> 2) | (((uint64_t)(3 ^ c)) << shift);
        const uint64_t h0 = cm_hash64_v<HV>(fw, mask), h1 = cm_hash64_v<HV>(rv, mask);
        const uint32_t strand = h0 < h1 ? 0u : 1u;
        e[j] = cm_hash64_v<HV>(strand ? h1 : h0, mask);
        sb |= strand << j;
      }
    }
    if (b == 0) {  // the first window's rule
      uint64_t v = e[0];
#pragma unroll
      for (int j = 1; j < 6; ++j) v = e[j] < v ? e[j] : v;
      if (e[6] == v) {
#pragma unroll
        for (int j = 0; j < 6; ++j) e[j] = e[j] == v ? ~0ull : e[j];
      }
    }
    // minima of the windows that start in the previous block, then their prefix maxima against the block before's suffix maxima
    uint64_t Mw[7];
    {
      uint64_t P = e[0];
      Mw[0] = Sp[0];
#pragma unroll
      for (int j = 1; j < 7; ++j) {
        Mw[j] = Sp[j] < P ? Sp[j] : P;
        P = e[j] < P ? e[j] : P;
      }
    }
    if (b > 0) {
      uint64_t PM = Mw[0];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (j > 0) PM = Mw[j] > PM ? Mw[j] : PM;
        const uint64_t F = j < 6 ? (SM[j < 6 ? j + 1 : 6] > PM ? SM[j < 6 ? j + 1 : 6] : PM) : PM;
        const uint32_t i = 7 * (b - 1) + (uint32_t)j;
        if (F == ep[j] && i < m) { emit(n, ep[j], ((i + (uint32_t)k - 1) << 1) | ((sbp >> j) & 1u)); ++n; }
      }
    }
    SM[6] = Mw[6];
    Sp[6] = e[6];
#pragma unroll
    for (int j = 5; j >= 0; --j) {
      SM[j] = Mw[j] > SM[j + 1] ? Mw[j] : SM[j + 1];
      Sp[j] = e[j] < Sp[j + 1] ? e[j] : Sp[j + 1];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) ep[j] = e[j];
    sbp = sb;
  }
  if (bad) return ~0u;
  return n;
}
template <class Emit>
CM_HD uint32_t cm_minimizers_w7_oddk(const uint8_t *seq, uint32_t len, int k, Emit &&emit) {
  if (len < (uint32_t)k + 6) return cm_minimizers_w7_oddk_seq(seq, len, k, emit);  // fewer than seven k-mers: the flush rule
  if (2 * k == 34) return cm_minimizers_w7_oddk_blocks<0>(seq, len, k, emit);      // (which hash arithmetic: chosen once)
  if (2 * k > 32 && 2 * k <= 52) return cm_minimizers_w7_oddk_blocks<1>(seq, len, k, emit);
  return cm_minimizers_w7_oddk_blocks<2>(seq, len, k, emit);
}

// ---------------------------------------------------------------------------------------
// The same closed form with one LANE PER K-MER POSITION instead of one lane per read (k_prep_flat): the reads of a
// block are packed to 2 bits per base, every lane hashes the k-mer of one position (three Hash64, nothing sequential),
// and the selection is two sliding extrema over the per-read hash arrays:
//   M[j] = min(h[j..j+6])                    for every complete window j of the read
//   k-mer i is emitted  <=>  h[i] == max{ M[j] : j in [i-6, i], j a complete window }
// (every window that contains i has a minimum <= h[i], so the largest of them equals h[i] exactly when one does).
// What the position-parallel form leaves to the sequential code: reads with a base outside ACGT / acgt, reads with
// fewer than 7 k-mers, and reads whose seventh k-mer ties with the minimum of the first six (the first-window rule
// strikes earlier k-mers out there) -- all rare, all redone by cm_minimizers_w7.
// ---------------------------------------------------------------------------------------
// 16 bytes -> 16 2-bit codes (CharToUint8's: A0 C1 G2 T3; byte j at bits 2j) + a mask of the bytes that are not A/C/G/T
// in either case.  Four bytes at a time: bits 1-2 of a letter Gray-decode to the code (cm_c2u); the letter is
// rebuilt from the code and compared with the case-folded byte.
CM_HD uint32_t cm_mmf_pack4(uint32_t w, uint32_t *bad4) {
  const uint32_t x = (w >> 1) & 0x03030303u;
  const uint32_t code = x ^ ((x >> 1) & 0x01010101u);
  const uint32_t b0 = code & 0x01010101u, b1 = (code >> 1) & 0x01010101u;
  const uint32_t canon = 0x41414141u + 2u * b0 + 6u * b1 + 11u * (b0 & b1);  // A C G T = 0x41 + {0, 2, 6, 0x13}
  const uint32_t diff = (w & 0xDFDFDFDFu) ^ canon;
  // non-zero bytes of diff -> one bit each
  const uint32_t nz = ((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff;
  const uint32_t m = (nz >> 7) & 0x01010101u;
  *bad4 = (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
  uint32_t t = (code | (code >> 6)) & 0x000F000Fu;
  t = (t | (t >> 12)) & 0xFFu;
  return t;
}
// the 2k-bit little-endian k-mer (base s+j at bits 2j) that starts at base `at` of a packed range (16 bases per word)
CM_HD uint64_t cm_mmf_kmer(const uint32_t *pk, uint32_t at, int k) {
  const uint32_t w = at >> 4, sh = (at & 15u) << 1;
  const uint64_t lo = (uint64_t)pk[w] | ((uint64_t)pk[w + 1] << 32);
  uint64_t v = lo >> sh;
  if (sh) v |= (uint64_t)pk[w + 2] << (64 - sh);
  return v & ((((uint64_t)1) << (2 * k)) - 1);
}
// reversal of the k 2-bit groups of v
CM_HD uint64_t cm_mmf_revgroups(uint64_t v, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t x = __brevll(v) >> (64 - 2 * k);
#else
  uint64_t x = 0;
  for (int i = 0; i < 64; ++i) x |= ((v >> i) & 1ull) << (63 - i);
  x >>= (64 - 2 * k);
#endif
  return ((x & 0x5555555555555555ull) << 1) | ((x >> 1) & 0x5555555555555555ull);
}
// minimizer hash and strand of the k-mer v (minimizer_generator.cc:47-57): fw has the first base in its top bits,
// rv = the reverse complement = the complemented little-endian k-mer
CM_HD uint64_t cm_mmf_hash(uint64_t v, int k, uint32_t *strand) {
  const uint64_t mask = (((uint64_t)1) << (2 * k)) - 1;
  const uint64_t fw = cm_mmf_revgroups(v, k), rv = ~v & mask;
  const uint64_t h0 = cm_hash64_k(fw, k, mask), h1 = cm_hash64_k(rv, k, mask);
  *strand = h0 < h1 ? 0u : 1u;
  return cm_hash64_k(*strand ? h1 : h0, k, mask);
}
// selection over one read's hashes h[0..m) (m >= 7); Mbuf: m entries of scratch.  Returns false when the read has to be
// redone sequentially (first-window tie); otherwise flag[i] = k-mer i is a minimizer.
CM_HD bool cm_mmf_select(const uint64_t *h, uint32_t m, uint64_t *Mbuf, uint8_t *flag) {
  uint64_t v5 = h[0];
  for (int j = 1; j < 6; ++j) v5 = h[j] < v5 ? h[j] : v5;
  if (h[6] == v5) return false;
  for (uint32_t j = 0; j + 7 <= m; ++j) {
    uint64_t x = h[j];
    for (int q = 1; q < 7; ++q) x = h[j + q] < x ? h[j + q] : x;
    Mbuf[j] = x;
  }
  for (uint32_t i = 0; i < m; ++i) {
    const uint32_t j0 = i >= 6 ? i - 6 : 0, j1 = i + 7 <= m ? i : m - 7;
    uint64_t x = 0;
    for (uint32_t j = j0; j <= j1; ++j) x = Mbuf[j] > x ? Mbuf[j] : x;
    flag[i] = x == h[i] ? 1 : 0;
  }
  return true;
}

// w = 7 front end: closed form for odd k, the state machine otherwise and for reads with ambiguous bases
template <class Emit>
CM_HD uint32_t cm_minimizers_w7(const uint8_t *seq, uint32_t len, int k, Emit &&emit) {
  if (k & 1) {
    const uint32_t n = cm_minimizers_w7_oddk(seq, len, k, emit);
    if (n != ~0u) return n;
  }
  return cm_minimizers_window_e<7>(seq, len, k, emit);
}

template <int W>
CM_HD uint32_t cm_minimizers_window(const uint8_t *seq, uint32_t len, int k, uint64_t *oh, uint32_t *op, uint32_t cap) {
  auto put = [&](uint32_t n, uint64_t h, uint32_t p) { if (oh && n < cap) { oh[n] = h; op[n] = p; } };
  if (W == 7) return cm_minimizers_w7(seq, len, k, put);
  return cm_minimizers_window_e<W>(seq, len, k, put);
}

// generic window size (runtime w): the same state machine with its window in private memory
CM_HD uint32_t cm_minimizers_ring(const uint8_t *seq, uint32_t len, int k, int w, uint64_t *oh, uint32_t *op, uint32_t cap) {
  return cm_minimizers_core<0>(seq, 0, len, true, k, w, [&](uint32_t n, uint64_t h, uint32_t p) {
    if (oh && n < cap) { oh[n] = h; op[n] = p; }
    return true;
  });
}

// Two-pass form used by the kernels: count (writes mm_cnt[r]) and fill (writes the read's
// minimizers straight into the dense arrays at mm_off[r]); seq = the read (global or LDS copy).
CM_HD void cm_s1_count(const CmDev &d, uint32_t r, const uint8_t *seq) {
  const uint32_t len = d.rlen[r];
  d.mm_cnt[r] = d.p.w == 7 ? cm_minimizers_window<7>(seq, len, d.p.k, nullptr, nullptr, 0)
                           : cm_minimizers_ring(seq, len, d.p.k, d.p.w, nullptr, nullptr, 0);
}
CM_HD void cm_s1_fill(const CmDev &d, uint32_t r, const uint8_t *seq) {
  const uint32_t len = d.rlen[r], cnt = d.mm_cnt[r];
  if (cnt == 0) return;
  uint64_t *oh = d.mm_hash + d.mm_off[r];
  uint32_t *op = d.mm_ps + d.mm_off[r];
  const uint32_t n = d.p.w == 7 ? cm_minimizers_window<7>(seq, len, d.p.k, oh, op, cnt)
                                : cm_minimizers_ring(seq, len, d.p.k, d.p.w, oh, op, cnt);
  if (n != cnt) d.stats[CM_ST_ERR] = 3;
}

// ---------------------------------------------------------------------------------------
// Reference minimizers for index construction (Index::Construct, index.cc:19-23), one
// chunk of `chunk` positions per item.  The state machine is cm_minimizers_core;
// it is started `warm` positions early (>= 2w+k, so ring buffer, running minimum and the
// unambiguous-length thresholds have converged to the sequential pass's state before the
// first owned position) and run w+2 positions past the chunk; an emission is kept only
// when the emitted k-mer's end position lies in [start, start+chunk).  Emitted hit =
// ((rid<<32 | pos) << 1) | strand (minimizer_generator.cc:58-60).  Pass oh = nullptr to count.
// ---------------------------------------------------------------------------------------
CM_HD uint32_t cm_ref_chunk_minimizers(const uint8_t *seq, uint32_t len, uint32_t rid, uint32_t s, uint32_t chunk,
                                       uint32_t warm, int k, int w, uint64_t *oh, uint64_t *ot) {
  const uint32_t e = s + chunk < len ? s + chunk : len;  // owned positions [s,e)
  const uint32_t begin = s > warm ? s - warm : 0;
  const uint32_t end = e + 2 * (uint32_t)w + 2 < len ? e + 2 * (uint32_t)w + 2 : len;
  auto own = [&](uint32_t n, uint64_t h, uint32_t p) {
    const uint32_t pp = p >> 1;
    if (pp < s || pp >= e) return false;
    if (oh) { oh[n] = h; ot[n] = (((uint64_t)rid << 32 | pp) << 1) | (p & 1u); }
    return true;
  };
  // final flush (minimizer_generator.cc:136-138): performed by every chunk whose run reaches
  // the end of the sequence; the ownership test keeps exactly one copy
  return w == 7 ? cm_minimizers_core<7>(seq, begin, end, end == len, k, w, own) : cm_minimizers_core<0>(seq, begin, end, end == len, k, w, own);
}


// ---------------------------------------------------------------------------------------
// S2: index probe of ONE minimizer (kh_get, khash.h:232-245, with the hash/equality of
//     index_utils.h:13-17).  Device layout: one 16-B {key,val} bucket at the same bucket
//     index as in the khash arrays, empty buckets hold CM_EMPTY_KEY, so the triangular
//     probe sequence and the hit/miss outcome are those of the reference.
//     Returns the number of buckets visited.
// ---------------------------------------------------------------------------------------
CM_HD uint32_t cm_probe(const uint64_t *bkt, uint32_t bmask, uint64_t hash, uint64_t *val_out,
                        uint8_t *kind_out) {
  uint32_t i = (uint32_t)hash & bmask;
  const uint32_t last = i;
  uint32_t step = 0, visited = 0;
  for (;;) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one 16-byte load per visited bucket
    const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(bkt + 2 * (uint64_t)i);
    const uint64_t key = kv.x, val = kv.y;
#else
    const uint64_t key = bkt[2 * (uint64_t)i], val = bkt[2 * (uint64_t)i + 1];
#endif
    ++visited;
    if (key == CM_EMPTY_KEY) break;
    if (key != CM_DELETED_KEY && (key >> 1) == hash) {
      *val_out = val;
      *kind_out = (key & 1) ? CM_PR_SINGLE : CM_PR_MULTI;
      return visited;
    }
    i = (i + (++step)) & bmask;
    if (i == last) break;
  }
  *val_out = 0;
  *kind_out = CM_PR_MISS;
  return visited;
}

// ---------------------------------------------------------------------------------------
// S3a: per read, how many hits GenerateCandidatePositions will produce and which round
//      is used (CandidateProcessor::GenerateCandidates candidate_processor.cc:12-71,
//      Index::GenerateCandidatePositions index.cc:237-349, UpdateRepetitiveSeedStats
//      index.cc:507-523).  Repetitive-seed statistics do not depend on the round.
// ---------------------------------------------------------------------------------------
CM_HD void cm_s3a_count(const CmDev &d, uint32_t r) {
  const uint32_t pair = r >> 1;
  // BothEndsHaveMinimizers (chromap.h:936); single-end: minimizers_.size() > 0 (chromap.h:416)
  bool live = d.p.single ? ((r & 1) == 0 && d.mm_cnt[r] > 0) : (d.mm_cnt[2 * pair] > 0 && d.mm_cnt[2 * pair + 1] > 0);
  // minimizer arrays that overflowed (the host sees it in the same read-back and reruns with larger ones): nothing to look at
  if (d.mm_cap && (uint64_t)d.mm_off[r] + d.mm_cnt[r] > d.mm_cap) live = false;
  uint32_t tot1 = 0, tot2 = 0, rep_len = 0, rep_cnt = 0, prev = ~0u;
  if (live) {
    const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
    for (uint32_t i = 0; i < n; ++i) {
      const uint8_t kind = d.pr_kind[b + i];
      if (kind == CM_PR_MISS) continue;
      if (kind == CM_PR_SINGLE) { ++tot1; ++tot2; continue; }
      const uint32_t nocc = (uint32_t)d.pr_val[b + i];
      if (!(nocc >= (uint32_t)d.p.f0)) tot1 += nocc;
      if (!(nocc >= (uint32_t)d.p.f1)) tot2 += nocc;
      if (nocc >= (uint32_t)d.p.f0) {
        const uint32_t rp = d.mm_ps[b + i] >> 1;
        if (prev > rp) rep_len += (uint32_t)d.p.k;
        else if (rp < prev + (uint32_t)d.p.k + (uint32_t)d.p.w - 1) rep_len += rp - prev;
        else rep_len += (uint32_t)d.p.k;
        prev = rp;
        ++rep_cnt;
      }
    }
  }
  const bool r2 = live && tot1 == 0;
  d.round2[r] = r2 ? 1 : 0;
  d.hit_tot[r] = live ? (r2 ? tot2 : tot1) : 0;
  d.rep_len[r] = rep_len;
  d.rep_cnt[r] = rep_cnt;
}

// GenerateCandidatePositionFromHits (index.cc:491-505); ps = (read_pos<<1)|strand
CM_HD uint64_t cm_cand_from_hit(uint64_t ref_hit, uint32_t ps, int k, bool *same) {
  const uint32_t ref_pos = (uint32_t)(ref_hit >> 1), read_pos = ps >> 1;
  *same = ((uint32_t)ref_hit & 1u) == (ps & 1u);
  const uint32_t start = *same ? ref_pos - read_pos : ref_pos + read_pos - (uint32_t)k + 1u;
  return ((uint64_t)(uint32_t)(ref_hit >> 33) << 32) | start;
}

// CandidateProcessor::GenerateCandidatesOnOneStrand (candidate_processor.cc:283-342) on a
// sorted hit list h[0..n); candidates are written IN PLACE into h/cnt (the write index
// never passes the read index).  Returns the number of candidates.
CM_HD uint32_t cm_sweep_strided(uint64_t *h, uint8_t *cnt, uint32_t n, int e, int seeds_required, uint32_t num_minimizers,
                                uint32_t st) {
  if (n == 0) return 0;
  uint32_t out = 0;
  int mcount = 1, equal = 1, best_equal = 1;
  uint64_t prev_hit = h[0];
  uint32_t prev_rid = (uint32_t)(prev_hit >> 32), prev_pos = (uint32_t)prev_hit;
  uint64_t best_local = h[0];
  for (uint32_t pi = 1; pi <= n; ++pi) {
    const uint64_t x = pi < n ? h[pi * st] : ~0ull;  // UINT64_MAX sentinel (:286)
    const uint32_t rid = (uint32_t)(x >> 32), pos = (uint32_t)x;
    if (rid != prev_rid || pos > prev_pos + (uint32_t)e ||
        ((uint32_t)mcount >= num_minimizers && pos > (uint32_t)best_local + (uint32_t)e)) {
      if (mcount >= seeds_required) {
        h[out * st] = best_local;
        cnt[out * st] = (uint8_t)best_equal;
        ++out;
      }
      mcount = 1; equal = 1; best_equal = 1;
      best_local = x;
    } else {
      if (x == best_local) { ++equal; ++best_equal; }
      else if (x == prev_hit) {
        ++equal;
        if (equal > best_equal) { best_local = prev_hit; best_equal = equal; }
      } else equal = 1;
      ++mcount;
    }
    prev_hit = x; prev_rid = rid; prev_pos = pos;
  }
  return out;
}

// The sweep cut at its state-free breaks, for lists that a group of lanes clusters together (k_s3b_heavy): a hit
// starts a LOCAL cluster when its rid differs from its predecessor's or its position lies more than e beyond it --
// the first two break conditions of cm_sweep_strided, which look at the two neighbours only.  At such a break the
// sweep's state is reset to (1, 1, 1, x), so the hits [b, end) of one local cluster can be swept on their own; the
// third, state-dependent break (enough minimizers and e beyond the best hit) is applied inside.  Candidates are
// appended to out_h / out_c (nullptr: count only); the concatenation over the local clusters in list order is what
// cm_sweep_strided produces.  Source list with element stride st, outputs dense.
CM_HD bool cm_sweep_local_break(uint64_t prev, uint64_t x, int e) {
  return (uint32_t)(x >> 32) != (uint32_t)(prev >> 32) || (uint32_t)x > (uint32_t)prev + (uint32_t)e;
}
CM_HD uint32_t cm_sweep_cluster(const uint64_t *h, uint32_t st, uint32_t b, uint32_t end, int e, int seeds_required,
                                uint32_t num_minimizers, uint64_t *out_h, uint8_t *out_c, uint64_t out_mask = ~0ull) {
  uint32_t out = 0;
  int mcount = 1, equal = 1, best_equal = 1;
  uint64_t prev_hit = h[(size_t)b * st], best_local = prev_hit;
  for (uint32_t pi = b + 1; pi <= end; ++pi) {
    const bool last = pi == end;
    const uint64_t x = last ? ~0ull : h[(size_t)pi * st];
    if (last || ((uint32_t)mcount >= num_minimizers && (uint32_t)x > (uint32_t)best_local + (uint32_t)e)) {
      if (mcount >= seeds_required) {
        if (out_h) { out_h[out] = best_local & out_mask; out_c[out] = (uint8_t)best_equal; }
        ++out;
      }
      mcount = 1; equal = 1; best_equal = 1;
      best_local = x;
    } else {
      if (x == best_local) { ++equal; ++best_equal; }
      else if (x == prev_hit) {
        ++equal;
        if (equal > best_equal) { best_local = prev_hit; best_equal = equal; }
      } else equal = 1;
      ++mcount;
    }
    prev_hit = x;
  }
  return out;
}

// the same for the local cluster that STARTS at b, its end found on the way (the first hit at or after b + 1 that
// cm_sweep_local_break separates from its predecessor, or n): one walk over the cluster instead of two
CM_HD uint32_t cm_sweep_cluster_from(const uint64_t *h, uint32_t b, uint32_t n, int e, int seeds_required, uint32_t num_minimizers,
                                     uint64_t *out_h, uint8_t *out_c, uint64_t out_mask = ~0ull) {
  uint32_t out = 0;
  int mcount = 1, equal = 1, best_equal = 1;
  uint64_t prev_hit = h[b], best_local = prev_hit;
  uint64_t ahead = b + 1 < n ? h[b + 1] : ~0ull;  // the next hit is requested while the current one is looked at
  for (uint32_t pi = b + 1;; ++pi) {
    uint64_t x = ahead;
    ahead = pi + 1 < n ? h[pi + 1] : ~0ull;
    const bool last = pi >= n || cm_sweep_local_break(prev_hit, x, e);
    if (last) x = ~0ull;
    if (last || ((uint32_t)mcount >= num_minimizers && (uint32_t)x > (uint32_t)best_local + (uint32_t)e)) {
      if (mcount >= seeds_required) {
        if (out_h) { out_h[out] = best_local & out_mask; out_c[out] = (uint8_t)best_equal; }
        ++out;
      }
      if (last) break;
      mcount = 1; equal = 1; best_equal = 1;
      best_local = x;
    } else {
      if (x == best_local) { ++equal; ++best_equal; }
      else if (x == prev_hit) {
        ++equal;
        if (equal > best_equal) { best_local = prev_hit; best_equal = equal; }
      } else equal = 1;
      ++mcount;
    }
    prev_hit = x;
  }
  return out;
}

// Hit keys of the cooperative hit-list stages (cm_coop.h): the 64-bit key of the reference (sequence << 32 | position), or -- a
// reference that fits -- the 32-bit GLOBAL coordinate goff[sequence] + position (CmDev::goff: the sequences laid end to end with gaps
// wider than a read + the error threshold, so hits on different sequences are never within e of each other and the "same sequence"
// test of the sweep is implied by the distance test).  Same order, same clusters, 4 instead of 8 bytes of shared memory per hit and
// buffer.
template <class K> struct CmKeyOps;
template <> struct CmKeyOps<uint64_t> {
  static CM_HD bool brk(uint64_t prev, uint64_t x, int e) { return cm_sweep_local_break(prev, x, e); }
  static CM_HD uint32_t pos(uint64_t x) { return (uint32_t)x; }
  static CM_HD uint64_t top() { return ~0ull; }
  static CM_HD uint64_t cand(uint64_t x, const uint32_t *, uint32_t) { return x & ~(1ull << 63); }
};
template <> struct CmKeyOps<uint32_t> {
  static CM_HD bool brk(uint32_t prev, uint32_t x, int e) { return x > prev + (uint32_t)e; }
  static CM_HD uint32_t pos(uint32_t x) { return x; }
  static CM_HD uint32_t top() { return ~0u; }
  static CM_HD uint64_t cand(uint32_t x, const uint32_t *goff, uint32_t n_seq) {  // back to sequence << 32 | position
    uint32_t lo = 0, hi = n_seq;  // the largest sequence with goff <= x
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (goff[mid] <= x) lo = mid; else hi = mid;
    }
    return ((uint64_t)lo << 32) | (x - goff[lo]);
  }
};
// how a candidate leaves the sweep: as a key (32-bit keys stay keys) or as the reference's sequence << 32 | position
template <class K, class KOUT> struct CmKeyOut { static CM_HD KOUT get(K x, const uint32_t *goff, uint32_t n_seq) { return (KOUT)CmKeyOps<K>::cand(x, goff, n_seq); } };
template <> struct CmKeyOut<uint32_t, uint32_t> { static CM_HD uint32_t get(uint32_t x, const uint32_t *, uint32_t) { return x; } };
// cm_sweep_cluster_from that also says where the local cluster ends (*end_out: the index of the first hit behind it, at most n); the
// parked candidates keep the key type
template <class K>
CM_HD uint32_t cm_sweep_cluster_walk(const K *h, uint32_t b, uint32_t n, int e, int seeds_required, uint32_t num_minimizers,
                                     K *out_h, uint8_t *out_c, uint32_t *end_out) {
  typedef CmKeyOps<K> KO;
  uint32_t out = 0;
  int mcount = 1, equal = 1, best_equal = 1;
  K prev_hit = h[b], best_local = prev_hit;
  K ahead = b + 1 < n ? h[b + 1] : KO::top();
  uint32_t pi = b + 1;
  for (;; ++pi) {
    K x = ahead;
    ahead = pi + 1 < n ? h[pi + 1] : KO::top();
    const bool last = pi >= n || KO::brk(prev_hit, x, e);
    if (last || ((uint32_t)mcount >= num_minimizers && KO::pos(x) > KO::pos(best_local) + (uint32_t)e)) {
      if (mcount >= seeds_required) {
        out_h[out] = best_local; out_c[out] = (uint8_t)best_equal;
        ++out;
      }
      if (last) break;
      mcount = 1; equal = 1; best_equal = 1;
      best_local = x;
    } else {
      if (x == best_local) { ++equal; ++best_equal; }
      else if (x == prev_hit) {
        ++equal;
        if (equal > best_equal) { best_local = prev_hit; best_equal = equal; }
      } else equal = 1;
      ++mcount;
    }
    prev_hit = x;
  }
  *end_out = pi;
  return out;
}

CM_HD uint32_t cm_sweep(uint64_t *h, uint8_t *cnt, uint32_t n, int e, int seeds_required, uint32_t num_minimizers) {
  return cm_sweep_strided(h, cnt, n, e, seeds_required, num_minimizers, 1);
}

// ---------------------------------------------------------------------------------------
// S3b: per read -- expand occurrences into the read's hit segment (+ list from the front,
//      - list from the back), sort both, cluster into candidates in place.
// ---------------------------------------------------------------------------------------
// work buffer h/hc with element stride st (1 = the read's global segment, blockDim = LDS
// [entry][thread] layout).  Candidates end up at h[0..ncp) and h[np..np+ncn) (strided).
#define CM_S3B_GROUP 4
CM_HD void cm_s3b_core(const CmDev &d, uint32_t r, uint64_t *h, uint8_t *hc, uint32_t st, uint32_t *np_out,
                       uint32_t *ncp_out, uint32_t *ncn_out) {
  const uint32_t tot = d.hit_tot[r];
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  const uint32_t maxf = d.round2[r] ? (uint32_t)d.p.f1 : (uint32_t)d.p.f0;
  uint32_t np = 0, nn = 0;
  // minimizers are taken four at a time: their (kind, value, position) triples and the first
  // occurrence of every run are requested together, so four independent gathers are in flight per
  // lane instead of one dependent chain
  for (uint32_t i0 = 0; i0 < n; i0 += CM_S3B_GROUP) {
    uint8_t kind[CM_S3B_GROUP];
    uint64_t val[CM_S3B_GROUP], first[CM_S3B_GROUP];
    uint32_t ps[CM_S3B_GROUP];
#pragma unroll
    for (int q = 0; q < CM_S3B_GROUP; ++q) {
      const bool in = i0 + q < n;
      kind[q] = in ? d.pr_kind[b + i0 + q] : (uint8_t)CM_PR_MISS;
      val[q] = in ? d.pr_val[b + i0 + q] : 0;
      ps[q] = in ? d.mm_ps[b + i0 + q] : 0;
    }
#pragma unroll
    for (int q = 0; q < CM_S3B_GROUP; ++q) {
      const bool run = kind[q] != CM_PR_MISS && kind[q] != CM_PR_SINGLE && (uint32_t)val[q] < maxf && (uint32_t)val[q] > 0;
      first[q] = run ? d.occ[(uint32_t)(val[q] >> 32)] : 0;
    }
#pragma unroll
    for (int q = 0; q < CM_S3B_GROUP; ++q) {
      if (kind[q] == CM_PR_MISS) continue;
      bool same;
      if (kind[q] == CM_PR_SINGLE) {
        const uint64_t cp = cm_cand_from_hit(val[q], ps[q], d.p.k, &same);
        if (same) h[(np++) * st] = cp; else h[(tot - 1 - nn++) * st] = cp;
        continue;
      }
      const uint32_t nocc = (uint32_t)val[q];
      if (nocc >= maxf) continue;
      const uint64_t *o = d.occ + (uint32_t)(val[q] >> 32);
      for (uint32_t oi = 0; oi < nocc; ++oi) {
        const uint64_t cp = cm_cand_from_hit(oi == 0 ? first[q] : o[oi], ps[q], d.p.k, &same);
        if (same) h[(np++) * st] = cp; else h[(tot - 1 - nn++) * st] = cp;
      }
    }
  }
  cm_sort_u64_strided(h, np, st);
  cm_sort_u64_strided(h + (size_t)np * st, nn, st);
  const bool use_high = d.round2[r] && np > 0 && nn > 0;
  int req = (int)n - (int)d.rep_cnt[r];
  req = req > 1 ? req : 1;
  req = req > d.p.min_seeds ? d.p.min_seeds : req;
  if (use_high) req = d.p.min_seeds;
  *np_out = np;
  *ncp_out = cm_sweep_strided(h, hc, np, d.p.e, req, n, st);
  *ncn_out = cm_sweep_strided(h + (size_t)np * st, hc + (size_t)np * st, nn, d.p.e, req, n, st);
}

// lds_h / lds_c: per-block LDS work buffers of lds_cap entries per thread ([entry][thread]
// layout, this thread's column), or nullptr.  Reads with at most lds_cap hits are expanded,
// sorted and clustered in LDS and only their candidates are written to the global segment;
// longer lists work in place in the global segment.
CM_HD void cm_s3b_candidates_lds(const CmDev &d, uint32_t r, uint64_t *lds_h, uint8_t *lds_c, uint32_t lds_cap,
                                 uint32_t lds_stride) {
  const uint32_t tot = d.hit_tot[r];
  d.ncp[r] = 0; d.ncn[r] = 0; d.n_pos_hit[r] = 0;
  if (tot == 0) return;
  uint64_t *h = d.hbuf + d.hit_off[r];
  uint8_t *hc = d.hcnt + d.hit_off[r];
  uint32_t np, ncp, ncn;
  if (lds_h && tot <= lds_cap) {
    cm_s3b_core(d, r, lds_h, lds_c, lds_stride, &np, &ncp, &ncn);
    for (uint32_t i = 0; i < ncp; ++i) { h[i] = lds_h[i * lds_stride]; hc[i] = lds_c[i * lds_stride]; }
    for (uint32_t i = 0; i < ncn; ++i) { h[np + i] = lds_h[(np + i) * lds_stride]; hc[np + i] = lds_c[(np + i) * lds_stride]; }
  } else {
    cm_s3b_core(d, r, h, hc, 1, &np, &ncp, &ncn);
  }
  d.n_pos_hit[r] = np;
  d.ncp[r] = ncp;
  d.ncn[r] = ncn;
}
CM_HD void cm_s3b_candidates(const CmDev &d, uint32_t r) { cm_s3b_candidates_lds(d, r, nullptr, nullptr, 0, 1); }

// ---------------------------------------------------------------------------------------
// Mate rescue (Index::GenerateCandidatePositionsFromRepetitiveReadWithMateInfoOnOneStrand,
// index.cc:351-489).  One function does both the counting pass (out == nullptr) and the
// fill pass.  strand: 0 = kPositive, 1 = kNegative.  Mate candidates are (mp, mc, mn),
// sorted by position.  The merged windows (:383-412) are re-derived per minimizer by a
// streaming merge over the mate candidates instead of being stored.
// Returns max_minimizer_count, or its negation when the search bails out (:371-380).
// ---------------------------------------------------------------------------------------
// The search is split into pieces a group of lanes can share (k_s4a/4b_rescue_list): the best count among the mate
// candidates, the bail-out test, ONE minimizer's hits inside the merged windows (independent of the other minimizers; inside a
// minimizer the windows are chained: a window's binary search starts at the midpoint where the previous one ended, and the scan
// that follows starts at that midpoint without a lower-bound test, :443-470), and the repetitive-seed length over the
// minimizers in order.  cm_rescue is their sequential composition.
CM_HD bool cm_rescue_bails(const CmDev &d, int max_count, int best_num, uint32_t mn) {  // :371-380
  return best_num >= 300 || mn > (uint32_t)d.p.f0 || (max_count <= d.p.min_seeds && best_num >= 200);
}
CM_HD uint32_t cm_rescue_minimizer(const CmDev &d, int strand, const uint64_t *mp, const uint8_t *mc, uint32_t mn, int max_count,
                                   uint8_t kind, uint64_t val, uint32_t ps, uint64_t *out, unsigned long long *reads_out) {
  if (kind == CM_PR_MISS) return 0;
  const uint32_t search_range = 2u * (uint32_t)d.p.max_insert;
  uint32_t cnt = 0;
  unsigned long long reads = 0;
  bool same;
  if (kind == CM_PR_SINGLE) {
    const uint64_t cp = cm_cand_from_hit(val, ps, d.p.k, &same);
    if ((same && strand == 0) || (!same && strand == 1)) { if (out) out[cnt] = cp; ++cnt; }
    return cnt;
  }
  const uint32_t off = (uint32_t)(val >> 32), nocc = (uint32_t)val;
  const uint64_t *o = d.occ + off;
  int32_t prev_l = 0;
  // streaming merge of the windows of the best mate candidates
  uint32_t ci = 0;
  bool have = false;
  uint64_t ws = 0, we = 0;
  for (;;) {
    // find next window [ws,we]
    bool emit = false;
    uint64_t es = 0, ee = 0;
    while (ci < mn) {
      if (mc[ci] != max_count) { ++ci; continue; }
      const uint64_t pos = mp[ci];
      const uint64_t s = pos < search_range ? 0 : pos - search_range;
      const uint64_t en = pos + search_range;
      ++ci;
      if (!have) { ws = s; we = en; have = true; continue; }
      if (we < s) { es = ws; ee = we; emit = true; ws = s; we = en; break; }
      we = en;
    }
    if (!emit) {
      if (!have) break;
      es = ws; ee = we; have = false; emit = true;  // last window
    }
    // binary search for the window start (:443-460)
    int32_t l = prev_l, m = 0, rr = (int32_t)(nocc - 1);
    while (l <= rr) {
      m = (l + rr) / 2;
      const uint64_t cp = o[m] >> 1;
      ++reads;
      if (cp < es) l = m + 1;
      else if (cp > es) rr = m - 1;
      else break;
    }
    prev_l = m;
    for (uint32_t oi = (uint32_t)m; oi < nocc; ++oi) {
      const uint64_t rh = o[oi];
      ++reads;
      if ((rh >> 1) > ee) break;
      const uint64_t cp = cm_cand_from_hit(rh, ps, d.p.k, &same);
      if ((same && strand == 0) || (!same && strand == 1)) { if (out) out[cnt] = cp; ++cnt; }
    }
    if (!have && ci >= mn) break;
  }
  if (reads_out) *reads_out += reads;
  return cnt;
}
// repetitive_seed_length over the minimizers in order (:474-486, index.cc:507-523)
CM_HD void cm_rescue_rep(const CmDev &d, uint8_t kind, uint64_t val, uint32_t ps, uint32_t *rep_len, uint32_t *prev_rep) {
  if (kind == CM_PR_MISS || kind == CM_PR_SINGLE) return;
  const uint32_t nocc = (uint32_t)val;
  if (nocc >= (uint32_t)d.p.f0) {
    const uint32_t rp = ps >> 1;
    if (*prev_rep > rp) *rep_len += (uint32_t)d.p.k;
    else if (rp < *prev_rep + (uint32_t)d.p.k + (uint32_t)d.p.w - 1) *rep_len += rp - *prev_rep;
    else *rep_len += (uint32_t)d.p.k;
    *prev_rep = rp;
  }
}
CM_HD int cm_rescue(const CmDev &d, uint32_t r, int strand, const uint64_t *mp, const uint8_t *mc, uint32_t mn,
                    uint64_t *out, uint32_t *n_out, uint32_t *rep_len_out, unsigned long long *occ_reads) {
  int max_count = 0, best_num = 0;
  for (uint32_t i = 0; i < mn; ++i) {
    const int c = mc[i];
    if (c > max_count) { max_count = c; best_num = 1; }
    else if (c == max_count) ++best_num;
  }
  *n_out = 0;
  if (cm_rescue_bails(d, max_count, best_num, mn)) return -max_count;
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  uint32_t cnt = 0, rep_len = 0, prev_rep = ~0u;
  unsigned long long reads = 0;
  // the lookup results of four minimizers are requested together (one round trip instead of four dependent ones: the
  // rescued reads are few and every one of them is a chain of such trips -- the kernel's duration is the chain's length)
  for (uint32_t mi0 = 0; mi0 < n; mi0 += CM_S3B_GROUP) {
    uint8_t kind_g[CM_S3B_GROUP];
    uint64_t val_g[CM_S3B_GROUP];
    uint32_t ps_g[CM_S3B_GROUP];
#pragma unroll
    for (int q = 0; q < CM_S3B_GROUP; ++q) {
      const bool in = mi0 + q < n;
      kind_g[q] = in ? d.pr_kind[b + mi0 + q] : (uint8_t)CM_PR_MISS;
      val_g[q] = in ? d.pr_val[b + mi0 + q] : 0;
      ps_g[q] = in ? d.mm_ps[b + mi0 + q] : 0;
    }
#pragma unroll
    for (int q = 0; q < CM_S3B_GROUP; ++q) {
      cnt += cm_rescue_minimizer(d, strand, mp, mc, mn, max_count, kind_g[q], val_g[q], ps_g[q], out ? out + cnt : nullptr, &reads);
      cm_rescue_rep(d, kind_g[q], val_g[q], ps_g[q], &rep_len, &prev_rep);
    }
  }
  *n_out = cnt;
  *rep_len_out = rep_len;
  if (occ_reads) *occ_reads += reads;
  return max_count;
}

// candidate lists after GenerateCandidates live in hbuf: + at hit_off[r], - at hit_off[r]+n_pos_hit[r]
CM_HD const uint64_t *cm_c0_pos(const CmDev &d, uint32_t r) { return d.hbuf + d.hit_off[r]; }
CM_HD const uint8_t *cm_c0_pcnt(const CmDev &d, uint32_t r) { return d.hcnt + d.hit_off[r]; }
CM_HD const uint64_t *cm_c0_neg(const CmDev &d, uint32_t r) { return d.hbuf + d.hit_off[r] + d.n_pos_hit[r]; }
CM_HD const uint8_t *cm_c0_ncnt(const CmDev &d, uint32_t r) { return d.hcnt + d.hit_off[r] + d.n_pos_hit[r]; }

// ---------------------------------------------------------------------------------------
// S4a: per read -- SupplementCandidates decision and rescue-hit counting
//      (candidate_processor.cc:75-191)
// ---------------------------------------------------------------------------------------
// S4a in two parts so that a block can run the (rare, long) rescue searches on packed lanes:
// cm_s4a_decide: does this read supplement its candidates from its mate (candidate_processor.cc:
// 75-146)?  cm_s4a_rescue: the counting pass of the two rescue searches for a read that does.
CM_HD bool cm_s4a_decide(const CmDev &d, uint32_t r) {
  const uint32_t pair = r >> 1;
  d.aug[r] = 0; d.res_neg[r] = 0; d.res_pos[r] = 0; d.resc_n[r] = 0; d.resc_p[r] = 0;
  const bool live = d.p.single ? ((r & 1) == 0 && d.mm_cnt[r] > 0) : (d.mm_cnt[2 * pair] > 0 && d.mm_cnt[2 * pair + 1] > 0);
  const uint32_t ncp = d.ncp[r], ncn = d.ncn[r];
  d.m_tot[r] = live ? ncp + ncn : 0;
  if (!live) return false;
  const uint32_t mm_count = d.mm_cnt[r];
  bool augment = !d.p.split && !d.p.single;  // split alignment / single-end never supplement (chromap.h:1021)
  const uint8_t *pc = cm_c0_pcnt(d, r), *nc = cm_c0_ncnt(d, r);
  for (uint32_t i = 0; augment && i < ncp; ++i) if (pc[i] >= mm_count / 2) { augment = false; break; }
  if (augment) for (uint32_t i = 0; i < ncn; ++i) if (nc[i] >= mm_count / 2) { augment = false; break; }
  return augment;
}
CM_HD void cm_s4a_rescue(const CmDev &d, uint32_t r) {
  const uint32_t o = r ^ 1u;
  d.aug[r] = 1;
  uint32_t cntn = 0, cntp = 0, rl = 0;
  int res_neg = 0, res_pos = 0;
  bool set_rl = false;
  uint32_t rl_val = 0;
  unsigned long long occ_reads = 0;
  if (d.ncp[o] > 0) {  // mate + candidates drive a search on our - strand (:147-153)
    res_neg = cm_rescue(d, r, 1, cm_c0_pos(d, o), cm_c0_pcnt(d, o), d.ncp[o], nullptr, &cntn, &rl, &occ_reads);
    if (res_neg >= 0) { set_rl = true; rl_val = rl; }
  }
  if (d.ncn[o] > 0) {
    res_pos = cm_rescue(d, r, 0, cm_c0_neg(d, o), cm_c0_ncnt(d, o), d.ncn[o], nullptr, &cntp, &rl, &occ_reads);
    if (res_pos >= 0) { set_rl = true; rl_val = rl; }
  }
  d.res_neg[r] = res_neg; d.res_pos[r] = res_pos;
  d.resc_n[r] = cntn; d.resc_p[r] = cntp;
  if (set_rl) d.rep_len[r] = rl_val;  // repetitive_seed_length overwritten (:113,131, index.cc:487)
  d.m_tot[r] = d.ncp[r] + d.ncn[r] + cntn + cntp;
}
CM_HD void cm_s4a_rescue_count(const CmDev &d, uint32_t r) {
  if (cm_s4a_decide(d, r)) cm_s4a_rescue(d, r);
}

// CandidateProcessor::MergeCandidates (candidate_processor.cc:345-414).  c1 = original list,
// c2 = augmented list stored at out[n1 ..]; result written to out[0..] (see cm_s4b).
CM_HD uint32_t cm_merge(const uint64_t *p1, const uint8_t *c1, uint32_t n1, uint64_t *out, uint8_t *outc,
                        uint32_t n2, int e) {
  const uint64_t *p2 = out + n1;
  const uint8_t *c2 = outc + n1;
  if (n1 == 0) return n2;  // c1.swap(c2): the augmented list is already in place
  uint32_t i = 0, j = 0, k = 0;
#define CM_BACK_OK(P) (k == 0 || (P) > out[k - 1] + (uint64_t)(int64_t)e)
  while (i < n1 && j < n2) {
    const uint64_t a = p1[i], b = p2[j];
    if (a == b) {
      if (CM_BACK_OK(a)) {
        const uint8_t ca = c1[i], cb = c2[j];
        if (ca > cb) { out[k] = a; outc[k] = ca; } else { out[k] = b; outc[k] = cb; }
        ++k;
      }
      ++i; ++j;
    } else if (a < b) {
      if (CM_BACK_OK(a)) { const uint8_t ca = c1[i]; out[k] = a; outc[k] = ca; ++k; }
      ++i;
    } else {
      if (CM_BACK_OK(b)) { const uint8_t cb = c2[j]; out[k] = b; outc[k] = cb; ++k; }
      ++j;
    }
  }
  while (i < n1) { const uint64_t a = p1[i]; if (CM_BACK_OK(a)) { const uint8_t ca = c1[i]; out[k] = a; outc[k] = ca; ++k; } ++i; }
  while (j < n2) { const uint64_t b = p2[j]; if (CM_BACK_OK(b)) { const uint8_t cb = c2[j]; out[k] = b; outc[k] = cb; ++k; } ++j; }
#undef CM_BACK_OK
  return k;
}

// ---------------------------------------------------------------------------------------
// S4b: per read -- fill rescue hits, cluster them (num_seeds_required = 1), merge with the
//      original candidates (candidate_processor.cc:193-230, 265-281).
//      Region layout in mbuf: + list at m_off[r] (capacity ncp+resc_p), - list right after
//      (capacity ncn+resc_n).  Rescue hits are staged at the END of each region so the
//      merge can write from the front without overtaking unread input.
// ---------------------------------------------------------------------------------------
// mode CM_S4B_ALL: fill, sort, cluster, merge.  CM_S4B_PREFILLED: the rescue hits are already in place (a group of lanes
// wrote them, k_s4b_rescue_list); their numbers are the counts of S4a.  CM_S4B_FILL_ONLY: only write the rescue hits -- a
// group of lanes sorts and merges them afterwards (cm_coop_rescue_merge, cm_coop.h).
#define CM_S4B_ALL 0
#define CM_S4B_PREFILLED 1
#define CM_S4B_FILL_ONLY 2
CM_HD void cm_s4b_rescue_merge(const CmDev &d, uint32_t r, int mode = CM_S4B_ALL) {
  const bool prefilled = mode == CM_S4B_PREFILLED;
  const uint32_t o = r ^ 1u;
  d.mcp[r] = 0; d.mcn[r] = 0;  // (also when only filling: the group that finishes the read may not run -- speculative launch set -- and then nothing stale must be left)
  if (d.m_tot[r] == 0) return;
  const uint32_t ncp = d.ncp[r], ncn = d.ncn[r], rp = d.resc_p[r], rn = d.resc_n[r];
  uint64_t *P = d.mbuf + d.m_off[r];
  uint8_t *PC = d.mcnt + d.m_off[r];
  uint64_t *N = P + ncp + rp;
  uint8_t *NC = PC + ncp + rp;
  const uint64_t *p0 = cm_c0_pos(d, r), *n0 = cm_c0_neg(d, r);
  const uint8_t *pc0 = cm_c0_pcnt(d, r), *nc0 = cm_c0_ncnt(d, r);
  uint32_t naug_p = 0, naug_n = 0;
  if (d.aug[r]) {
    uint32_t cnt = 0, rl = 0;
    if (d.ncp[o] > 0 && d.res_neg[r] >= 0 && rn > 0) {
      if (prefilled) cnt = rn; else cm_rescue(d, r, 1, cm_c0_pos(d, o), cm_c0_pcnt(d, o), d.ncp[o], N + ncn, &cnt, &rl, nullptr);
      if (mode != CM_S4B_FILL_ONLY) {
        cm_sort_u64(N + ncn, cnt);
        naug_n = cm_sweep(N + ncn, NC + ncn, cnt, d.p.e, 1, d.mm_cnt[r]);
      }
    }
    if (d.ncn[o] > 0 && d.res_pos[r] >= 0 && rp > 0) {
      if (prefilled) cnt = rp; else cm_rescue(d, r, 0, cm_c0_neg(d, o), cm_c0_ncnt(d, o), d.ncn[o], P + ncp, &cnt, &rl, nullptr);
      if (mode != CM_S4B_FILL_ONLY) {
        cm_sort_u64(P + ncp, cnt);
        naug_p = cm_sweep(P + ncp, PC + ncp, cnt, d.p.e, 1, d.mm_cnt[r]);
      }
    }
  }
  if (mode == CM_S4B_FILL_ONLY) return;
  if (naug_p > 0) {
    d.mcp[r] = cm_merge(p0, pc0, ncp, P, PC, naug_p, d.p.e);
  } else {
    for (uint32_t i = 0; i < ncp; ++i) { P[i] = p0[i]; PC[i] = pc0[i]; }
    d.mcp[r] = ncp;
  }
  if (naug_n > 0) {
    d.mcn[r] = cm_merge(n0, nc0, ncn, N, NC, naug_n, d.p.e);
  } else {
    for (uint32_t i = 0; i < ncn; ++i) { N[i] = n0[i]; NC[i] = nc0[i]; }
    d.mcn[r] = ncn;
  }
}

// merged candidate list accessors
CM_HD uint64_t *cm_m_pos(const CmDev &d, uint32_t r) { return d.mbuf + d.m_off[r]; }
CM_HD uint8_t *cm_m_pcnt(const CmDev &d, uint32_t r) { return d.mcnt + d.m_off[r]; }
CM_HD uint64_t *cm_m_neg(const CmDev &d, uint32_t r) { return d.mbuf + d.m_off[r] + d.ncp[r] + d.resc_p[r]; }
CM_HD uint8_t *cm_m_ncnt(const CmDev &d, uint32_t r) { return d.mcnt + d.m_off[r] + d.ncp[r] + d.resc_p[r]; }
CM_HD uint64_t *cm_f_pos(const CmDev &d, uint32_t r) { return d.fbuf + d.m_off[r]; }
CM_HD uint8_t *cm_f_pcnt(const CmDev &d, uint32_t r) { return d.fcnt + d.m_off[r]; }
CM_HD uint64_t *cm_f_neg(const CmDev &d, uint32_t r) { return d.fbuf + d.m_off[r] + d.ncp[r] + d.resc_p[r]; }
CM_HD uint8_t *cm_f_ncnt(const CmDev &d, uint32_t r) { return d.fcnt + d.m_off[r] + d.ncp[r] + d.resc_p[r]; }

// ReduceCandidatesForPairedEndReadOnOneDirection (candidate_processor.cc:416-484)
CM_HD void cm_reduce_dir(uint32_t dist, const uint64_t *p1, const uint8_t *c1, uint32_t n1, const uint64_t *p2,
                         const uint8_t *c2, uint32_t n2, uint64_t *f1, uint8_t *fc1, uint32_t *nf1, uint64_t *f2,
                         uint8_t *fc2, uint32_t *nf2) {
  uint32_t i1 = 0, i2 = 0, o1 = 0, o2 = 0;
  int unpaired1 = 0, unpaired2 = 0;
  int max1 = 6, max2 = 6;
  uint32_t prev_end_i2 = 0;
  while (i1 < n1 && i2 < n2) {
    const uint64_t a = p1[i1], b = p2[i2];
    if (a > b + dist) {
      if (i2 >= prev_end_i2 && unpaired2 < 5 && (a >> 32) == (b >> 32) && (int)c2[i2] >= max2) {
        f2[o2] = b; fc2[o2] = c2[i2]; ++o2;
        ++unpaired2;
      }
      ++i2;
    } else if (b > a + dist) {
      if (unpaired1 < 5 && (a >> 32) == (b >> 32) && (int)c1[i1] >= max1) {
        f1[o1] = a; fc1[o1] = c1[i1]; ++o1;
        ++unpaired1;
      }
      ++i1;
    } else {
      f1[o1] = a; fc1[o1] = c1[i1]; ++o1;
      if ((int)c1[i1] > max1) max1 = c1[i1];
      uint32_t cur = i2;
      while (cur < n2 && p2[cur] <= a + dist) {
        if (cur >= prev_end_i2) {
          f2[o2] = p2[cur]; fc2[o2] = c2[cur]; ++o2;
          if ((int)c2[cur] > max2) max2 = c2[cur];
        }
        ++cur;
      }
      prev_end_i2 = cur;
      ++i1;
    }
  }
  *nf1 = o1;
  *nf2 = o2;
}

// ---------------------------------------------------------------------------------------
// S4c: per pair -- SupplementCandidates return value, candidate-count gates and the
//      paired-end filter (chromap.h:1020-1056, candidate_processor.cc:183-191, 233-263)
// ---------------------------------------------------------------------------------------
// Chromap::RerankCandidatesRid (chromap.cc:916-923): the candidates that go to verification get their rid
// replaced by its rank in the custom chromosome order (chromap.h:416-420, 1060-1074)
CM_HD void cm_rerank(const CmDev &d, uint32_t r) {
  if (!d.rid_rank) return;
  uint64_t *fp = cm_f_pos(d, r), *fn = cm_f_neg(d, r);
  for (uint32_t i = 0; i < d.fcp[r]; ++i) fp[i] = (fp[i] & 0xffffffffull) | ((uint64_t)d.rid_rank[(uint32_t)(fp[i] >> 32)] << 32);
  for (uint32_t i = 0; i < d.fcn[r]; ++i) fn[i] = (fn[i] & 0xffffffffull) | ((uint64_t)d.rid_rank[(uint32_t)(fn[i] >> 32)] << 32);
}

// cm_s4c_reduce in three parts so that a group of lanes can run the filter of a pair with long candidate lists
// (cm_coop_s4c, cm_coop.h): cm_s4c_pre does everything up to the filter and says whether it has to run, cm_s4c_filter is
// the two directions, cm_s4c_post the pair's fate and the re-ranking.
CM_HD bool cm_s4c_pre(const CmDev &d, uint32_t pair) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  d.fcp[r1] = d.fcn[r1] = d.fcp[r2] = d.fcn[r2] = 0;
  d.alive[pair] = 0;
  d.force0[pair] = 0;
  if (d.p.single) {  // chromap.h:442-445: candidates go straight to verification
    if (d.mm_cnt[r1] == 0 || d.mcp[r1] + d.mcn[r1] == 0) return false;
    const uint64_t *mp = cm_m_pos(d, r1), *mn = cm_m_neg(d, r1);
    const uint8_t *mpc = cm_m_pcnt(d, r1), *mnc = cm_m_ncnt(d, r1);
    uint64_t *fp = cm_f_pos(d, r1), *fn = cm_f_neg(d, r1);
    uint8_t *fpc = cm_f_pcnt(d, r1), *fnc = cm_f_ncnt(d, r1);
    for (uint32_t i = 0; i < d.mcp[r1]; ++i) { fp[i] = mp[i]; fpc[i] = mpc[i]; }
    for (uint32_t i = 0; i < d.mcn[r1]; ++i) { fn[i] = mn[i]; fnc[i] = mnc[i]; }
    d.fcp[r1] = d.mcp[r1]; d.fcn[r1] = d.mcn[r1];
    d.alive[pair] = 1;
    cm_rerank(d, r1);
    return false;
  }
  if (!(d.mm_cnt[r1] > 0 && d.mm_cnt[r2] > 0)) return false;
  int ret = 0;
  for (uint32_t r = r1; r <= r2; ++r) {
    if (!d.aug[r]) continue;
    const int pr = d.res_neg[r], nr = d.res_pos[r];  // positive_rescue_result / negative_rescue_result
    if (((pr < 0 && nr > 0 && -pr >= nr) || (pr > 0 && nr < 0 && pr <= -nr)) && d.ncp[r] + d.ncn[r] == 0) ret = 1;
  }
  d.force0[pair] = (uint8_t)ret;
  const uint32_t nc1 = d.mcp[r1] + d.mcn[r1], nc2 = d.mcp[r2] + d.mcn[r2];
  if (!(nc1 > 0 && nc2 > 0)) return false;
  if (d.p.split) {  // no paired-end filter (chromap.h:1036-1038): candidates pass through unchanged
    for (uint32_t r = r1; r <= r2; ++r) {
      const uint64_t *mp = cm_m_pos(d, r), *mn = cm_m_neg(d, r);
      const uint8_t *mpc = cm_m_pcnt(d, r), *mnc = cm_m_ncnt(d, r);
      uint64_t *fp = cm_f_pos(d, r), *fn = cm_f_neg(d, r);
      uint8_t *fpc = cm_f_pcnt(d, r), *fnc = cm_f_ncnt(d, r);
      for (uint32_t i = 0; i < d.mcp[r]; ++i) { fp[i] = mp[i]; fpc[i] = mpc[i]; }
      for (uint32_t i = 0; i < d.mcn[r]; ++i) { fn[i] = mn[i]; fnc[i] = mnc[i]; }
      d.fcp[r] = d.mcp[r]; d.fcn[r] = d.mcn[r];
    }
    d.alive[pair] = 1;
    cm_rerank(d, r1); cm_rerank(d, r2);
    return false;
  }
  return true;
}
CM_HD void cm_s4c_filter(const CmDev &d, uint32_t pair) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  uint32_t a, b;
  cm_reduce_dir((uint32_t)d.p.max_insert, cm_m_pos(d, r1), cm_m_pcnt(d, r1), d.mcp[r1], cm_m_neg(d, r2),
                cm_m_ncnt(d, r2), d.mcn[r2], cm_f_pos(d, r1), cm_f_pcnt(d, r1), &a, cm_f_neg(d, r2), cm_f_ncnt(d, r2), &b);
  d.fcp[r1] = a; d.fcn[r2] = b;
  cm_reduce_dir((uint32_t)d.p.max_insert, cm_m_neg(d, r1), cm_m_ncnt(d, r1), d.mcn[r1], cm_m_pos(d, r2),
                cm_m_pcnt(d, r2), d.mcp[r2], cm_f_neg(d, r1), cm_f_ncnt(d, r1), &a, cm_f_pos(d, r2), cm_f_pcnt(d, r2), &b);
  d.fcn[r1] = a; d.fcp[r2] = b;
}
CM_HD void cm_s4c_post(const CmDev &d, uint32_t pair) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  const uint32_t f1 = d.fcp[r1] + d.fcn[r1], f2 = d.fcp[r2] + d.fcn[r2];
  d.alive[pair] = (f1 > 0 && f2 > 0) ? 1 : 0;
  if (d.alive[pair]) { cm_rerank(d, r1); cm_rerank(d, r2); }
}
CM_HD void cm_s4c_reduce(const CmDev &d, uint32_t pair) {
  if (!cm_s4c_pre(d, pair)) return;
  cm_s4c_filter(d, pair);
  cm_s4c_post(d, pair);
}

// ---------------------------------------------------------------------------------------
// K4: verification
// ---------------------------------------------------------------------------------------
// BandedAlignPatternToText (alignment.cc:141-192): Myers/Hyyro bit-vector banded edit
// distance, 32-bit word, band 2e+1.  pattern = reference window, text = read.
// neg: text is the reverse complement of `read` (read[len-1-i] complemented).
CM_HD uint32_t cm_text_code(const uint8_t *read, int L, int i, bool neg) {
  if (!neg) return cm_c2u(read[i]);
  const uint32_t c = cm_c2u(read[L - 1 - i]);
  return c < 4 ? 3u ^ c : 4u;
}
CM_HD uint8_t cm_text_char(const uint8_t *read, int L, int i, bool neg) {
  return neg ? cm_negchar(read[L - 1 - i]) : read[i];
}

CM_HD uint32_t cm_peq_get(const uint32_t *P, uint32_t c) {
  return c == 0 ? P[0] : c == 1 ? P[1] : c == 2 ? P[2] : c == 3 ? P[3] : P[4];
}
CM_HD void cm_peq_or(uint32_t *P, uint32_t c, uint32_t bit) {
  P[0] |= c == 0 ? bit : 0u; P[1] |= c == 1 ? bit : 0u; P[2] |= c == 2 ? bit : 0u;
  P[3] |= c == 3 ? bit : 0u; P[4] |= c == 4 ? bit : 0u;
}

// Sequential byte readers over global memory: one aligned 8-byte load per eight bytes, the next word requested while the
// current one is consumed.  (Their predecessor copied window and read into per-lane arrays first; indexed at run time those
// arrays were scratch memory -- 700 bytes written and read back per alignment; a byte load per step from global memory,
// before that, made every step a dependent round trip.)
// n = bytes that will be read; no word outside [first, last] of them is touched.
// (Round 3 tried 16-byte chunks with two loads in flight: fewer round trips, but every byte then costs a two-word shift --
// k_s5b_verify 7.9 -> 9.4 ms on the repeat workload; the kernel hides its load latency with occupancy already.)
struct CmFwdReader {  // p[0], p[1], ...
  const uint64_t *ap;
  uint64_t cur, nxt;
  uint32_t left, words;  // bytes left in cur; words not yet requested
  CM_HD void init(const uint8_t *p, uint32_t n) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t sh = (uint32_t)(a & 7);
    ap = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
    words = n ? (sh + n + 7) >> 3 : 0;
    cur = 0; nxt = 0; left = 0;
    if (words) { cur = *ap++ >> (sh * 8); left = 8 - sh; --words; }
    if (words) { nxt = *ap++; --words; }
  }
  CM_HD uint8_t next() {
    if (left == 0) {
      cur = nxt; left = 8;
      if (words) { nxt = *ap++; --words; }
    }
    const uint8_t b = (uint8_t)cur;
    cur >>= 8; --left;
    return b;
  }
};
struct CmBwdReader {  // p[0], p[-1], p[-2], ...
  const uint64_t *ap;
  uint64_t cur, nxt;
  uint32_t left, words;
  CM_HD void init(const uint8_t *p, uint32_t n) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint32_t pos = (uint32_t)(a & 7);  // byte of p inside its word
    ap = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
    words = n ? ((7 - pos) + n + 7) >> 3 : 0;
    cur = 0; nxt = 0; left = 0;
    if (words) { cur = *ap-- << ((7 - pos) * 8); left = pos + 1; --words; }
    if (words) { nxt = *ap--; --words; }
  }
  CM_HD uint8_t next() {
    if (left == 0) {
      cur = nxt; left = 8;
      if (words) { nxt = *ap--; --words; }
    }
    const uint8_t b = (uint8_t)(cur >> 56);
    cur <<= 8; --left;
    return b;
  }
};

// BandedAlignPatternToText (alignment.cc:141-192) with the pattern and the text streamed; NEG: the text is the reverse
// complement of the read (bytes read backwards, codes complemented)
template <bool NEG>
CM_HD int cm_banded_align_stream(int e, const uint8_t *pattern, const uint8_t *read, int Lfull, int toff, int L, int *end_pos) {
  CmFwdReader pr;
  pr.init(pattern, (uint32_t)(L + 2 * e));
  CmFwdReader tf;
  CmBwdReader tb;
  if (NEG) tb.init(read + (Lfull - 1 - toff), (uint32_t)L); else tf.init(read + toff, (uint32_t)L);
  uint32_t P[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) cm_peq_or(P, cm_c2u(pr.next()), 1u << i);
  const uint32_t hi = 1u << (2 * e);
  uint32_t VP = 0, VN = 0;
  int err = 0;
  for (int i = 0; i < L; i++) {
    cm_peq_or(P, cm_c2u(pr.next()), hi);
    uint32_t tc;
    if (NEG) { const uint32_t c = cm_c2u(tb.next()); tc = c < 4 ? 3u ^ c : 4u; } else tc = cm_c2u(tf.next());
    uint32_t X = cm_peq_get(P, tc) | VN;
    const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
    const uint32_t HN = VP & D0;
    const uint32_t HP = VN | ~(VP | D0);
    X = D0 >> 1;
    VN = X & HP;
    VP = HN | ~(X | HP);
    err += 1 - (int)(D0 & 1u);
    if (err > 3 * e) return e + 1;
    P[0] >>= 1; P[1] >>= 1; P[2] >>= 1; P[3] >>= 1; P[4] >>= 1;
  }
  const int band_start = L - 1;
  int min_err = err;
  *end_pos = band_start;
  for (int i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *end_pos = band_start + 1 + i;
    }
  }
  return min_err;
}

// text = (neg ? revcomp(read[0..Lfull)) : read) + toff, length L
CM_HD int cm_banded_align(int e, const uint8_t *pattern, const uint8_t *read, int Lfull, bool neg, int toff, int L,
                          int *end_pos) {
  return neg ? cm_banded_align_stream<true>(e, pattern, read, Lfull, toff, L, end_pos)
             : cm_banded_align_stream<false>(e, pattern, read, Lfull, toff, L, end_pos);
}

// ---------------------------------------------------------------------------------------
// BandedAlignPatternToText (alignment.cc:141-192) on BIT PLANES -- the form k_s5b_verify runs.  The byte form above spends
// most of a column on getting at its two symbols (two streamed byte readers, two CharToUint8, five conditional ORs into the
// Peq words, a five-way select out of them: ~130 instructions per column on gfx950, the select compiled to branches) and the
// Myers step itself is twelve.  Here a sequence is three bit planes -- bit i of plane 0 / 1 = the two bits of base i's code,
// plane 2 = the base is none of ACGTacgt (code 4, planes 0 / 1 then 0) -- packed once: the reference when it is loaded, every
// read of the batch in both orientations before the alignments.  The Peq word of text symbol c at column i,
//     bit j  <=>  code(pattern[i + j]) == c,   j = 0 .. 2e
// is then bits [i, i + 2e] of   c < 4 ?  ~(R0 ^ m0) & ~(R1 ^ m1) & ~RN  :  RN    (m0, m1: bit 0 / 1 of c spread over the word),
// a shift of three 64-bit windows and four logic operations; the windows slide by one 32-bit word per 32 columns.  Everything
// after the Peq word is the byte form's code, so the two return the same (distance, end position) for every input.
//   rp:  reference plane records (CmDev::ref_pl);  g: index of the window's first base in the reference bytes
//             (the byte form's `pattern` - d.ref);  tp / tw: the text's planes in the wanted orientation (CmDev::read_pl)
// ---------------------------------------------------------------------------------------
CM_HD uint64_t cm_load8(const uint8_t *p);
#if defined(__HIP_DEVICE_COMPILE__)
#define CM_GLOBAL_U32 const __attribute__((address_space(1))) uint32_t *
#define CM_GLOBAL_PL const __attribute__((address_space(1))) CmPlRec *
#else
#define CM_GLOBAL_U32 const uint32_t *
#define CM_GLOBAL_PL const CmPlRec *
#endif
CM_HD uint32_t cm_brev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(v);
#else
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
  return (v >> 16) | (v << 16);
#endif
}
CM_HD uint32_t cm_funnel32(uint32_t lo, uint32_t hi, uint32_t sh) {  // bits [sh, sh + 32) of hi:lo, sh < 32
  return sh ? (lo >> sh) | (hi << (32u - sh)) : lo;
}
CM_HD int cm_banded_align_planes(int e, const CmPlRec *rp, uint64_t g, const uint32_t *tp, uint32_t tw, int L, int *end_pos) {
  CM_GLOBAL_PL rr = (CM_GLOBAL_PL)rp + (g >> 5);  // one 16-byte load per record: the window's records are neighbours in memory
  CM_GLOBAL_U32 t0 = (CM_GLOBAL_U32)tp;
  CM_GLOBAL_U32 t1 = t0 + tw;
  CM_GLOBAL_U32 tn = t1 + tw;
  const uint32_t sh = (uint32_t)g & 31u;
  const CmPlRec ra = rr[0], rb = rr[1], rc = rr[2];
  uint32_t a0 = ra.p0, a1 = ra.p1, an = ra.pn, b0 = rb.p0, b1 = rb.p1, bn = rb.pn;
  uint32_t c0 = rc.p0, c1 = rc.p1, cn = rc.pn;
  uint32_t x0 = t0[0], x1 = t1[0], xn = tn[0];
  const uint32_t band = (2u << (2 * e)) - 1u;
  uint32_t VP = 0, VN = 0;
  int err = 0;
  const int nchunk = (L + 31) >> 5;
  for (int c = 0; c < nchunk; ++c) {
    // the next chunk's words are requested before this chunk's columns run
    const bool more = c + 1 < nchunk;
    CmPlRec rd = {0u, 0u, 0u, 0u};
    if (more) rd = rr[c + 3];
    const uint32_t d0 = rd.p0, d1 = rd.p1, dn = rd.pn;
    const uint32_t y0 = more ? t0[c + 1] : 0u, y1 = more ? t1[c + 1] : 0u, yn = more ? tn[c + 1] : 0u;
    const uint64_t W0 = (uint64_t)cm_funnel32(a0, b0, sh) | ((uint64_t)cm_funnel32(b0, c0, sh) << 32);
    const uint64_t W1 = (uint64_t)cm_funnel32(a1, b1, sh) | ((uint64_t)cm_funnel32(b1, c1, sh) << 32);
    const uint64_t WN = (uint64_t)cm_funnel32(an, bn, sh) | ((uint64_t)cm_funnel32(bn, cn, sh) << 32);
    const int jn = L - 32 * c < 32 ? L - 32 * c : 32;
    for (int j = 0; j < jn; ++j) {
      const uint32_t B0 = (uint32_t)(W0 >> j), B1 = (uint32_t)(W1 >> j), BN = (uint32_t)(WN >> j);
      const uint32_t m0 = 0u - ((x0 >> j) & 1u), m1 = 0u - ((x1 >> j) & 1u), mn = 0u - ((xn >> j) & 1u);
      const uint32_t eq = ~(B0 ^ m0) & ~(B1 ^ m1) & ~BN;
      uint32_t X = ((((eq ^ BN) & mn) ^ eq) & band) | VN;  // text symbol outside ACGT: the positions where the pattern's is too
      const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
      const uint32_t HN = VP & D0;
      const uint32_t HP = VN | ~(VP | D0);
      X = D0 >> 1;
      VN = X & HP;
      VP = HN | ~(X | HP);
      err += 1 - (int)(D0 & 1u);
      if (err > 3 * e) return e + 1;
    }
    a0 = b0; b0 = c0; c0 = d0; a1 = b1; b1 = c1; c1 = d1; an = bn; bn = cn; cn = dn;
    x0 = y0; x1 = y1; xn = yn;
  }
  const int band_start = L - 1;
  int min_err = err;
  *end_pos = band_start;
  for (int i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *end_pos = band_start + 1 + i;
    }
  }
  return min_err;
}
// BandedAlignPatternToTextWithDropOff / ...WithDropOffFrom3End (alignment.cc:197-376) on bit planes: the split-alignment
// verification (cm_draft_strand_split), same planes, same Peq word, the byte form's (cm_banded_align_dropoff_stream) bookkeeping.
//   REV = false (the + strand's calls): pattern = reference bases g, g + 1, ..; text = read bases tbit, tbit + 1, ..
//   REV = true  (the - strand's calls, "from the 3' end" of the reverse complement): pattern = reference bases g, g - 1, ..
//               (g: the window's LAST base); text = the COMPLEMENT of read bases tbit, tbit + 1, .. -- the reverse complement
//               walked backwards is the read walked forwards, complemented
// tp: the read's FORWARD planes (orientation 0 of CmDev::read_pl) in both cases.
template <bool REV>
CM_HD int cm_banded_align_dropoff_planes(int e, const CmPlRec *rp, uint64_t g, const uint32_t *tp, uint32_t tw, uint32_t tbit, int L,
                                         int *end_pos, int *read_mapping_length) {
  // reference words: REV: the 64-bit field that ENDS at base g - 32 c, bit-reversed; else the field that starts at g + 32 c.
  // REV reaches below the window's first base (the byte form never did): the field of chunk 0 starts 63 bases below g and every
  // further chunk one record lower -- down to record -2 for a window at the very start of the reference buffer, whose bits are
  // not looked at.  The index is signed and CM_PL_LEAD zero records stand in front of record 0.
  const int64_t f0 = REV ? (int64_t)g - 63 : (int64_t)g;  // first base of chunk 0's field
  CM_GLOBAL_PL rr = (CM_GLOBAL_PL)rp + (f0 >> 5);         // (arithmetic shift: floor)
  const uint32_t sh = (uint32_t)(f0 & 31);
  CM_GLOBAL_U32 t0 = (CM_GLOBAL_U32)tp + (tbit >> 5);
  CM_GLOBAL_U32 t1 = t0 + tw;
  CM_GLOBAL_U32 tn = t1 + tw;
  const uint32_t tsh = tbit & 31u;
  const uint32_t tlast = (tbit + (uint32_t)(L > 0 ? L - 1 : 0)) >> 5;  // last text word that holds a base of the text
  const uint32_t tfirst = tbit >> 5;
  const CmPlRec ra = rr[0], rb = rr[1], rc = rr[2];
  uint32_t a0 = ra.p0, a1 = ra.p1, an = ra.pn, b0 = rb.p0, b1 = rb.p1, bn = rb.pn, c0 = rc.p0, c1 = rc.p1, cn = rc.pn;
  uint32_t xa0 = t0[0], xa1 = t1[0], xan = tn[0];
  uint32_t xb0 = tfirst + 1 <= tlast ? t0[1] : 0u, xb1 = tfirst + 1 <= tlast ? t1[1] : 0u, xbn = tfirst + 1 <= tlast ? tn[1] : 0u;
  const uint32_t band = (2u << (2 * e)) - 1u;
  uint32_t VP = 0, VN = 0, prev_VP = 0, prev_VN = 0;
  int err = 0, i = 0, prev_err = 0;
  bool fail_beginning = false, stop = false;
  const int nchunk = (L + 31) >> 5;
  for (int c = 0; c < nchunk && !stop; ++c) {
    // the next chunk's words: one reference word further up (down when REV), one text word further up
    const bool more = c + 1 < nchunk;
    uint32_t d0 = 0, d1 = 0, dn = 0, y0 = 0, y1 = 0, yn = 0;
    if (more) {
      const int ri = REV ? -(c + 1) : c + 3;
      const CmPlRec rd = rr[ri];
      d0 = rd.p0; d1 = rd.p1; dn = rd.pn;
      if (tfirst + (uint32_t)c + 2 <= tlast) { y0 = t0[c + 2]; y1 = t1[c + 2]; yn = tn[c + 2]; }
    }
    uint64_t W0 = (uint64_t)cm_funnel32(a0, b0, sh) | ((uint64_t)cm_funnel32(b0, c0, sh) << 32);
    uint64_t W1 = (uint64_t)cm_funnel32(a1, b1, sh) | ((uint64_t)cm_funnel32(b1, c1, sh) << 32);
    uint64_t WN = (uint64_t)cm_funnel32(an, bn, sh) | ((uint64_t)cm_funnel32(bn, cn, sh) << 32);
    if (REV) {
      W0 = ((uint64_t)cm_brev32((uint32_t)W0) << 32) | cm_brev32((uint32_t)(W0 >> 32));
      W1 = ((uint64_t)cm_brev32((uint32_t)W1) << 32) | cm_brev32((uint32_t)(W1 >> 32));
      WN = ((uint64_t)cm_brev32((uint32_t)WN) << 32) | cm_brev32((uint32_t)(WN >> 32));
    }
    const uint32_t x0 = cm_funnel32(xa0, xb0, tsh), x1 = cm_funnel32(xa1, xb1, tsh), xn = cm_funnel32(xan, xbn, tsh);
    const int jn = L - 32 * c < 32 ? L - 32 * c : 32;
    for (int j = 0; j < jn; ++j, ++i) {
      const uint32_t B0 = (uint32_t)(W0 >> j), B1 = (uint32_t)(W1 >> j), BN = (uint32_t)(WN >> j);
      uint32_t m0 = 0u - ((x0 >> j) & 1u), m1 = 0u - ((x1 >> j) & 1u);
      const uint32_t mn = 0u - ((xn >> j) & 1u);
      if (REV) { m0 = ~m0; m1 = ~m1; }
      const uint32_t eq = ~(B0 ^ m0) & ~(B1 ^ m1) & ~BN;
      uint32_t X = ((((eq ^ BN) & mn) ^ eq) & band) | VN;
      const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
      const uint32_t HN = VP & D0;
      const uint32_t HP = VN | ~(VP | D0);
      X = D0 >> 1;
      prev_VN = VN; prev_VP = VP;
      VN = X & HP;
      VP = HN | ~(X | HP);
      prev_err = err;
      err += 1 - (int)(D0 & 1u);
      if (err > 2 * e) {
        if (i < 4 * e && i < L / 2) fail_beginning = true;
        stop = true;
        break;
      }
    }
    if (REV) { c0 = b0; b0 = a0; a0 = d0; c1 = b1; b1 = a1; a1 = d1; cn = bn; bn = an; an = dn; }
    else { a0 = b0; b0 = c0; c0 = d0; a1 = b1; b1 = c1; c1 = d1; an = bn; bn = cn; cn = dn; }
    xa0 = xb0; xb0 = y0; xa1 = xb1; xb1 = y1; xan = xbn; xbn = yn;
  }
  if (i < L) { err = prev_err; VN = prev_VN; VP = prev_VP; }
  const int band_start = i - 1;
  int min_err = err;
  *read_mapping_length = i;
  *end_pos = band_start;
  for (i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *end_pos = band_start + 1 + i;
    }
  }
  if (fail_beginning || (L > 60 && *end_pos + 1 - e - min_err < 30)) *end_pos = -*end_pos;
  return min_err;
}

// 32 bases -> one word of each plane; n < 32: the bases beyond count as code 4
// pc (optional): the base is a lower-case letter (bit 5 of a byte at or above 'a'; meaningful where pn is 0)
CM_HD void cm_pack_planes32(const uint8_t *bytes, uint32_t n, uint32_t *p0, uint32_t *p1, uint32_t *pn, uint32_t *pc = nullptr) {
  uint32_t q0 = 0, q1 = 0, qn = 0, qc = 0;
  for (uint32_t j0 = 0; j0 < 32; j0 += 8) {
    uint64_t v = j0 < n ? cm_load8(bytes + j0) : 0;
    for (uint32_t j = j0; j < j0 + 8; ++j, v >>= 8) {
      const uint32_t u = j < n ? cm_c2u((uint8_t)v) : 4u;
      q0 |= (u & 1u) << j; q1 |= ((u >> 1) & 1u) << j; qn |= (u >> 2) << j;
      qc |= (j < n && (uint8_t)v >= 'a' && (uint8_t)v <= 'z' ? 1u : 0u) << j;
    }
  }
  *p0 = q0; *p1 = q1; *pn = qn;
  if (pc) *pc = qc;
}
// read r of the batch -> its planes, forward (the read as it is: the + strand's text) and reverse complement (base i =
// complement of read[L - 1 - i], L the trimmed length: the - strand's text, PrepareNegativeSequenceAt); the second from the first:
// word w of the reversed planes is the bit reversal of the forward planes' bits [L - 32 w - 32, L - 32 w).
// words a read's planes take in CmDev::read_pl: 6 W, rounded up to a multiple of four so that every read's planes start on a
// 16-byte boundary and leave in 16-byte stores (30 scalar stores per read of 150 bases made k_pack_reads the largest kernel of the hic
// workload)
CM_HD uint32_t cm_read_pl_stride(uint32_t W) { return (6u * W + 3u) & ~3u; }
CM_HD uint32_t *cm_read_pl_of(const CmDev &d, uint32_t r) { return d.read_pl + (size_t)r * cm_read_pl_stride(d.read_pl_w); }
// Any number of words per plane, the forward words read back from memory:
CM_HD void cm_pack_read_planes_any(const CmDev &d, uint32_t r) {
  const uint32_t W = d.read_pl_w, L = d.rlen[r];
  if (L == 0) return;
  uint32_t *f = cm_read_pl_of(d, r), *v = f + 3 * (size_t)W;
  const uint8_t *read = cm_read_ptr(d, r);
  const uint32_t nw = (L + 31) >> 5;
  for (uint32_t w = 0; w < nw; ++w) cm_pack_planes32(read + 32 * w, L - 32 * w, f + w, f + W + w, f + 2 * W + w);
  for (uint32_t w = 0; w < nw; ++w) {
    const int lo = (int)L - 32 * (int)w - 32;  // first source bit of this word
    uint32_t F[3];
    for (int q = 0; q < 3; ++q) {
      const uint32_t *src = f + (size_t)q * W;
      if (lo >= 0) {
        const uint32_t a = (uint32_t)lo >> 5, sh = (uint32_t)lo & 31u;
        F[q] = cm_funnel32(src[a], a + 1 < nw ? src[a + 1] : 0u, sh);
      } else F[q] = src[0] << (uint32_t)(-lo);
    }
    v[w] = ~cm_brev32(F[0]); v[W + w] = ~cm_brev32(F[1]); v[2 * W + w] = cm_brev32(F[2]);
  }
}
// W words per plane known at compile time: everything in registers -- the reversed planes are the forward words bit-reversed in
// reverse order, moved down by 32 W - L bits -- and the read's 6 W words leave in 16-byte stores (k_pack_reads: the reads of a
// wave are neighbours in the batch, so are their planes)
template <int W>
CM_HD void cm_pack_read_planes_w(const CmDev &d, uint32_t r) {
  const uint32_t L = d.rlen[r];
  if (L == 0) return;
  const uint8_t *read = cm_read_ptr(d, r);
  uint32_t o[6 * W];  // [orientation][plane][word]
  // the read's aligned 8-byte words, all requested before any is looked at (one round trip instead of a chain of them:
  // k_pack_reads was latency-bound with a load per eight bases in turn); nothing beyond the word that holds the last base
  const uintptr_t ra = reinterpret_cast<uintptr_t>(read);
  const uint32_t sh8 = (uint32_t)(ra & 7) * 8;
  const uint64_t *ap = reinterpret_cast<const uint64_t *>(ra & ~(uintptr_t)7);
  uint64_t wv[4 * W + 1];
#pragma unroll
  for (int q = 0; q <= 4 * W; ++q) wv[q] = (uint32_t)q * 8u < (sh8 >> 3) + L ? ap[q] : 0ull;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    uint32_t q0 = 0, q1 = 0, qn = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint64_t v = sh8 ? (wv[4 * w + k] >> sh8) | (wv[4 * w + k + 1] << (64u - sh8)) : wv[4 * w + k];
#pragma unroll
      for (int b = 0; b < 8; ++b, v >>= 8) {
        const uint32_t j = (uint32_t)(8 * k + b);
        const uint32_t u = 32u * (uint32_t)w + j < L ? cm_c2u((uint8_t)v) : 4u;
        q0 |= (u & 1u) << j; q1 |= ((u >> 1) & 1u) << j; qn |= (u >> 2) << j;
      }
    }
    o[w] = q0; o[W + w] = q1; o[2 * W + w] = qn;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    uint32_t *v = o + 3 * W + q * W;
#pragma unroll
    for (int w = 0; w < W; ++w) v[w] = cm_brev32(o[q * W + (W - 1 - w)]);
    uint32_t s = 32u * W - L;
    while (s >= 32) {
#pragma unroll
      for (int w = 0; w + 1 < W; ++w) v[w] = v[w + 1];
      v[W - 1] = 0;
      s -= 32;
    }
    if (s) {
#pragma unroll
      for (int w = 0; w < W; ++w) v[w] = (v[w] >> s) | (w + 1 < W ? v[w + 1] << (32u - s) : 0u);
    }
    if (q < 2) {
#pragma unroll
      for (int w = 0; w < W; ++w) v[w] = ~v[w];
    }
  }
  uint32_t *dst = cm_read_pl_of(d, r);
  struct alignas(16) Q { uint32_t a, b, c, e; };
#pragma unroll
  for (int i = 0; i < 6 * W; i += 4) {  // (the stride is a multiple of four words: the last store may run into the read's own padding)
    Q x = {o[i], i + 1 < 6 * W ? o[i + 1] : 0u, i + 2 < 6 * W ? o[i + 2] : 0u, i + 3 < 6 * W ? o[i + 3] : 0u};
    *reinterpret_cast<Q *>(dst + i) = x;
  }
}
CM_HD void cm_pack_read_planes(const CmDev &d, uint32_t r) {
  switch (d.read_pl_w) {
    case 1: cm_pack_read_planes_w<1>(d, r); break;
    case 2: cm_pack_read_planes_w<2>(d, r); break;
    case 3: cm_pack_read_planes_w<3>(d, r); break;
    case 4: cm_pack_read_planes_w<4>(d, r); break;
    // (2 x 150: the read-back form below took 7.7 ms per 2 M pairs, the largest kernel of the hic workload, profiles/r04a_hic_*)
    case 5: cm_pack_read_planes_w<5>(d, r); break;
    case 6: cm_pack_read_planes_w<6>(d, r); break;
    case 7: cm_pack_read_planes_w<7>(d, r); break;
    case 8: cm_pack_read_planes_w<8>(d, r); break;
    default: cm_pack_read_planes_any(d, r);
  }
}

// ---------------------------------------------------------------------------------------
// The first thing BandedTraceback does (alignment.cc:660-670) is a raw, case-sensitive Hamming
// count of the read against the window at offset e; when it equals the edit distance the start is e
// and nothing else runs -- the common case (substitution errors only).  Here that count works on
// 8 bytes at a time in registers (aligned 8-byte loads + funnel shift, SWAR byte compare) instead
// of byte-wise through the prefetched copies; for the - strand the read bytes are reversed and
// complemented with the same mapping as cm_negchar (PrepareNegativeSequenceAt).
// Reads up to 15 bytes past `pat + L` / the read's chunk: the buffers are padded (cm_api.hip).
// ---------------------------------------------------------------------------------------
CM_HD uint64_t cm_load8(const uint8_t *p) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t sh = (uint32_t)(a & 7) * 8;
  const uint64_t *ap = reinterpret_cast<const uint64_t *>(a & ~(uintptr_t)7);
  const uint64_t lo = ap[0];
  if (sh == 0) return lo;
  return (lo >> sh) | (ap[1] << (64 - sh));
}
CM_HD uint64_t cm_swar_eq(uint64_t v, uint64_t k) {  // 0xFF in every byte of v equal to the byte k, else 0x00
  const uint64_t z = v ^ (k * 0x0101010101010101ull);
  const uint64_t m = ~(((z & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | z) & 0x8080808080808080ull;
  return (m >> 7) * 0xFFull;
}
CM_HD uint64_t cm_swar_negchar(uint64_t v) {
  const uint64_t u = v & 0xDFDFDFDFDFDFDFDFull;
  const uint64_t mA = cm_swar_eq(u, 'A'), mC = cm_swar_eq(u, 'C'), mG = cm_swar_eq(u, 'G'), mT = cm_swar_eq(u, 'T');
  return (mA & 0x5454545454545454ull) | (mC & 0x4747474747474747ull) | (mG & 0x4343434343434343ull) | (mT & 0x4141414141414141ull) |
         (~(mA | mC | mG | mT) & 0x4E4E4E4E4E4E4E4Eull);
}
CM_HD uint64_t cm_bswap64(uint64_t v) {
  v = ((v & 0x00FF00FF00FF00FFull) << 8) | ((v >> 8) & 0x00FF00FF00FF00FFull);
  v = ((v & 0x0000FFFF0000FFFFull) << 16) | ((v >> 16) & 0x0000FFFF0000FFFFull);
  return (v << 32) | (v >> 32);
}
CM_HD int cm_swar_diff_bytes(uint64_t x) {  // number of non-zero bytes
  x |= x >> 1; x |= x >> 2; x |= x >> 4;
  return (int)__builtin_popcountll(x & 0x0101010101010101ull);
}
// #{ i in [0,L) : pat[i] != text(i) }, text = (neg ? revcomp(read[0..Lfull)) : read) + toff
CM_HD int cm_hamming_diag(const uint8_t *pat, const uint8_t *read, int Lfull, bool neg, int toff, int L) {
  int cnt = 0, i = 0;
  if (!neg) {
    const uint8_t *t = read + toff;
    for (; i + 8 <= L; i += 8) cnt += cm_swar_diff_bytes(cm_load8(pat + i) ^ cm_load8(t + i));
    for (; i < L; ++i) cnt += pat[i] != t[i];
  } else {
    const int R = Lfull - 1 - toff;  // text(i) = negchar(read[R - i])
    for (; i + 8 <= L; i += 8) cnt += cm_swar_diff_bytes(cm_load8(pat + i) ^ cm_swar_negchar(cm_bswap64(cm_load8(read + (R - i - 7)))));
    for (; i < L; ++i) cnt += pat[i] != cm_negchar(read[R - i]);
  }
  return cnt;
}

// BandedTraceback's bit-vector pass (alignment.cc:672-718) with both strings streamed backwards; its leading Hamming
// count is cm_hamming_diag (the caller below)
template <bool NEG>
CM_HD int cm_banded_traceback_stream(int e, int min_num_errors, const uint8_t *pattern, const uint8_t *read, int Lfull, int toff, int L) {
  CmBwdReader pb;
  pb.init(pattern + (L - 1 + 2 * e), (uint32_t)(L + 2 * e));
  CmFwdReader tf;
  CmBwdReader tb;
  {
    const int j0 = toff + L - 1;  // text string index of the first column
    if (NEG) tf.init(read + (Lfull - 1 - j0), (uint32_t)L); else tb.init(read + j0, (uint32_t)L);
  }
  uint32_t P[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) cm_peq_or(P, cm_c2u(pb.next()), 1u << i);
  const uint32_t hi = 1u << (2 * e);
  uint32_t VP = 0, VN = 0;
  int err = 0;
  for (int i = 0; i < L; i++) {
    cm_peq_or(P, cm_c2u(pb.next()), hi);
    uint32_t tc = cm_c2u(NEG ? tf.next() : tb.next());
    if (NEG) tc = tc < 4 ? 3u ^ tc : 4u;
    uint32_t X = cm_peq_get(P, tc) | VN;
    const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
    const uint32_t HN = VP & D0;
    const uint32_t HP = VN | ~(VP | D0);
    X = D0 >> 1;
    VN = X & HP;
    VP = HN | ~(X | HP);
    err += 1 - (int)(D0 & 1u);
    P[0] >>= 1; P[1] >>= 1; P[2] >>= 1; P[3] >>= 1; P[4] >>= 1;
  }
  int start = 2 * e;
  for (int i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err == min_num_errors) {
      start = 2 * e - (1 + i);
      if (i + 1 == e) return start;
    }
  }
  return start;
}
CM_HD int cm_banded_traceback(int e, int min_num_errors, const uint8_t *pattern, const uint8_t *read, int Lfull, bool neg,
                              int toff, int L) {
  if (min_num_errors == 0) return e;
  if (cm_hamming_diag(pattern + e, read, Lfull, neg, toff, L) == min_num_errors) return e;
  return neg ? cm_banded_traceback_stream<true>(e, min_num_errors, pattern, read, Lfull, toff, L)
             : cm_banded_traceback_stream<false>(e, min_num_errors, pattern, read, Lfull, toff, L);
}

// BandedAlignPatternToTextWithDropOff (alignment.cc:197-283) when FROM3 == false,
// BandedAlignPatternToTextWithDropOffFrom3End (alignment.cc:285-376) when FROM3 == true.
// text = (NEG ? revcomp(read) : read) + toff, length L.  Pattern and text are streamed (CmFwdReader / CmBwdReader):
// from the 3' end both run backwards over their strings, and the reverse complement flips the read's direction in memory.
template <bool FROM3, bool NEG>
CM_HD int cm_banded_align_dropoff_stream(int e, const uint8_t *pattern, const uint8_t *read, int Lfull, int toff, int L, int *end_pos,
                                         int *read_mapping_length) {
  constexpr bool TBWD = FROM3 != NEG;  // direction of the text in the read's memory
  CmFwdReader pf, tf;
  CmBwdReader pb, tb;
  if (FROM3) pb.init(pattern + (L + 2 * e - 1), (uint32_t)(L + 2 * e)); else pf.init(pattern, (uint32_t)(L + 2 * e));
  {
    // first text string index: FROM3 ? L - 1 : 0, i.e. j = toff + that; memory index NEG ? Lfull - 1 - j : j
    const int j0 = toff + (FROM3 ? L - 1 : 0);
    const uint8_t *t0 = read + (NEG ? Lfull - 1 - j0 : j0);
    if (TBWD) tb.init(t0, (uint32_t)L); else tf.init(t0, (uint32_t)L);
  }
  uint32_t P[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) cm_peq_or(P, cm_c2u(FROM3 ? pb.next() : pf.next()), 1u << i);
  const uint32_t hi = 1u << (2 * e);
  uint32_t VP = 0, VN = 0, prev_VP = 0, prev_VN = 0;
  int err = 0, i = 0, prev_err = 0;
  bool fail_beginning = false;
  for (; i < L; i++) {
    cm_peq_or(P, cm_c2u(FROM3 ? pb.next() : pf.next()), hi);
    uint32_t tc = cm_c2u(TBWD ? tb.next() : tf.next());
    if (NEG) tc = tc < 4 ? 3u ^ tc : 4u;
    uint32_t X = cm_peq_get(P, tc) | VN;
    const uint32_t D0 = ((VP + (X & VP)) ^ VP) | X;
    const uint32_t HN = VP & D0;
    const uint32_t HP = VN | ~(VP | D0);
    X = D0 >> 1;
    prev_VN = VN; prev_VP = VP;
    VN = X & HP;
    VP = HN | ~(X | HP);
    prev_err = err;
    err += 1 - (int)(D0 & 1u);
    if (err > 2 * e) {
      if (i < 4 * e && i < L / 2) fail_beginning = true;
      break;
    }
    P[0] >>= 1; P[1] >>= 1; P[2] >>= 1; P[3] >>= 1; P[4] >>= 1;
  }
  if (i < L) { err = prev_err; VN = prev_VN; VP = prev_VP; }
  const int band_start = i - 1;
  int min_err = err;
  *read_mapping_length = i;
  *end_pos = band_start;
  for (i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *end_pos = band_start + 1 + i;
    }
  }
  if (fail_beginning || (L > 60 && *end_pos + 1 - e - min_err < 30)) *end_pos = -*end_pos;
  return min_err;
}
CM_HD int cm_banded_align_dropoff(int e, const uint8_t *pattern, const uint8_t *read, int Lfull, bool neg, int toff, int L,
                                  bool from3, int *end_pos, int *read_mapping_length) {
  if (from3) return neg ? cm_banded_align_dropoff_stream<true, true>(e, pattern, read, Lfull, toff, L, end_pos, read_mapping_length)
                        : cm_banded_align_dropoff_stream<true, false>(e, pattern, read, Lfull, toff, L, end_pos, read_mapping_length);
  return neg ? cm_banded_align_dropoff_stream<false, true>(e, pattern, read, Lfull, toff, L, end_pos, read_mapping_length)
             : cm_banded_align_dropoff_stream<false, false>(e, pattern, read, Lfull, toff, L, end_pos, read_mapping_length);
}

// AdjustGapBeginning (alignment.cc:24-83) without cigar.  read string = (neg ? revcomp(read)
// : read) + toff; the - branch's loops stop at the string terminators (end of the read
// string, end of the chromosome).
CM_HD int cm_adjust_gap_beginning(int strand, const uint8_t *ref, uint32_t ref_len, const uint8_t *read, int Lfull, bool neg,
                                  int toff, int *gap_beginning, int read_end, int ref_start_position, int ref_end_position) {
  int i, j;
  if (strand == 0) {
    if (*gap_beginning <= 0) return ref_start_position;
    for (i = *gap_beginning - 1, j = ref_start_position - 1; i >= 0 && j >= 0; --i, --j) {
      const uint8_t rc = cm_text_char(read, Lfull, toff + i, neg), fc = ref[j];
      if (rc != fc && (int)rc != (int)fc - 'a' + 'A') break;
    }
    *gap_beginning = i + 1;
    return j + 1;
  }
  if (*gap_beginning <= 0) return ref_end_position;
  for (i = read_end + 1, j = ref_end_position + 1; toff + i < Lfull && (uint32_t)j < ref_len; ++i, ++j) {
    const uint8_t rc = cm_text_char(read, Lfull, toff + i, neg), fc = ref[j];
    if (rc != fc && (int)rc != (int)fc - 'a' + 'A') break;
  }
  *gap_beginning = *gap_beginning + i - (read_end + 1);
  return j - 1;
}

// IsValidCandidate (draft_mapping_generator.cc:59-70)
CM_HD bool cm_valid_candidate(const CmDev &d, uint32_t rid, uint32_t position, uint32_t L) {
  const uint32_t rl = d.ref_len[rid];
  return !(position < (uint32_t)d.p.e || position >= rl || position + L + (uint32_t)d.p.e >= rl);
}

struct CmBest { int min_err, second_err, n_best, n_second; };
CM_HD void cm_update_best(CmBest &m, int ne) {  // draft_mapping_generator.cc:502-528
  if (ne < m.min_err) { m.second_err = m.min_err; m.n_second = m.n_best; m.min_err = ne; m.n_best = 1; }
  else if (ne == m.min_err) m.n_best++;
  else if (ne == m.second_err) m.n_second++;
  else if (ne < m.second_err) { m.n_second = 1; m.second_err = ne; }
}

// verify one candidate; on acceptance append the draft mapping. returns accepted
// banded alignment of one (valid) candidate: returns the edit distance (e+1 = rejected)
CM_HD int cm_verify_compute(const CmDev &d, const uint8_t *read, uint32_t L, int strand, uint64_t cpos, int *end_pos) {
  const int e = d.p.e;
  const uint32_t rid = (uint32_t)(cpos >> 32);
  uint32_t position = (uint32_t)cpos;
  if (strand == 1) position = position - L + 1;
  *end_pos = (int)L;
  return cm_banded_align(e, d.ref + d.ref_off[rid] + position - e, read, (int)L, strand == 1, 0, (int)L, end_pos);
}

// pre_err/pre_end (may be null): results computed beforehand by the per-candidate kernel for
// candidate index ci of this strand's list
CM_HD bool cm_verify_one(const CmDev &d, const uint8_t *read, uint32_t L, int strand, uint64_t cpos, CmBest &bst,
                         uint64_t *dp, int16_t *de, uint32_t *nd, const int16_t *pre_err, const int16_t *pre_end,
                         uint32_t ci) {
  const int e = d.p.e;
  int end_pos = (int)L;
  int ne;
  if (pre_err) { ne = pre_err[ci]; end_pos = pre_end[ci]; }
  else ne = cm_verify_compute(d, read, L, strand, cpos, &end_pos);
  if (ne <= e) {
    cm_update_best(bst, ne);
    dp[*nd] = strand == 0 ? cpos - (uint64_t)e + (uint64_t)(int64_t)end_pos
                          : cpos - L + 1 - (uint64_t)e + (uint64_t)(int64_t)end_pos;
    de[*nd] = (int16_t)ne;
    ++*nd;
    return true;
  }
  return false;
}

// one strand of GenerateDraftMappings: scalar loop (draft_mapping_generator.cc:359-557, non-split)
// or the lane-grouped loop with the candidate_count_threshold break (:159-357)
CM_HD uint32_t cm_draft_strand(const CmDev &d, const uint8_t *read, uint32_t L, int strand, const uint64_t *cp,
                               const uint8_t *cc, uint32_t nc, CmBest &bst, uint64_t *dp, int16_t *de,
                               const int16_t *pre_err = nullptr, const int16_t *pre_end = nullptr) {
  uint32_t nd = 0;
  const int lanes = d.p.lanes;
  if (lanes == 0 || nc < (uint32_t)lanes) {
    for (uint32_t ci = 0; ci < nc; ++ci) {
      const uint32_t rid = (uint32_t)(cp[ci] >> 32);
      uint32_t position = (uint32_t)cp[ci];
      if (strand == 1) position = position - L + 1;
      if (!cm_valid_candidate(d, rid, position, L)) continue;
      cm_verify_one(d, read, L, strand, cp[ci], bst, dp, de, &nd, pre_err, pre_end, ci);
    }
    return nd;
  }
  uint64_t vpos[8];
  uint8_t vcnt[8];
  uint32_t vidx[8];
  uint32_t nvalid = 0, thr = 0, ci = 0;
  while (ci < nc) {
    if (cc[ci] < thr) break;
    const uint32_t rid = (uint32_t)(cp[ci] >> 32);
    uint32_t position = (uint32_t)cp[ci];
    if (strand == 1) position = position - L + 1;
    if (!cm_valid_candidate(d, rid, position, L)) { ++ci; continue; }
    vpos[nvalid] = cp[ci];
    vcnt[nvalid] = cc[ci];
    vidx[nvalid] = ci;
    ++nvalid;
    ++ci;
    if (nvalid < (uint32_t)lanes) continue;
    for (int mi = 0; mi < lanes; ++mi)
      if (!cm_verify_one(d, read, L, strand, vpos[mi], bst, dp, de, &nd, pre_err, pre_end, vidx[mi])) thr = vcnt[mi];
    nvalid = 0;
  }
  for (uint32_t i = 0; i < nvalid; ++i) cm_verify_one(d, read, L, strand, vpos[i], bst, dp, de, &nd, pre_err, pre_end, vidx[i]);
  return nd;
}


// GenerateDraftMappingsOnOneStrand, split-alignment branch (draft_mapping_generator.cc:359-557).
// best_mapping_longest_match is re-initialised per candidate in the reference (:404-405), so
// the second_min adjustment (:511-515) cannot fire and GetLongestMatchLength has no effect.
// tp: the read's forward bit planes (CmDev::read_pl, orientation 0) -- the alignments then run on planes
// (cm_banded_align_dropoff_planes) -- or nullptr: on the bytes
CM_HD uint32_t cm_draft_strand_split(const CmDev &d, const uint8_t *read, uint32_t L, int strand, const uint64_t *cp,
                                     const uint8_t *cc, uint32_t nc, CmBest &bst, uint64_t *dp, int16_t *de, uint32_t *ds,
                                     const uint32_t *tp = nullptr) {
  const int e = d.p.e;
  uint32_t nd = 0, thr = 0;
  for (uint32_t ci = 0; ci < nc; ++ci) {
    if (cc[ci] < thr) break;
    const uint32_t rid = (uint32_t)(cp[ci] >> 32);
    uint32_t position = (uint32_t)cp[ci];
    if (strand == 1) position = position - L + 1;
    if (!cm_valid_candidate(d, rid, position, L)) continue;
    int mep = (int)L, gap_beginning = 0, num_errors = 0, actual = 0, rml = 0;
    const int allow = 20 - e;
    const uint64_t gpat = d.ref_off[rid] + position - (uint32_t)e;  // the window's first base
    const uint8_t *pat = d.ref + gpat;
    const bool pl = tp != nullptr && d.ref_pl != nullptr;
    if (strand == 0) {
      num_errors = pl ? cm_banded_align_dropoff_planes<false>(e, d.ref_pl, gpat, tp, d.read_pl_w, 0u, (int)L, &mep, &rml)
                      : cm_banded_align_dropoff(e, pat, read, (int)L, false, 0, (int)L, false, &mep, &rml);
      if (mep < 0 && allow > 0) {
        const int b_err = num_errors, b_mep = -mep, b_rml = rml;
        num_errors = pl ? cm_banded_align_dropoff_planes<false>(e, d.ref_pl, gpat + (uint32_t)allow, tp, d.read_pl_w, (uint32_t)allow, (int)L - allow, &mep, &rml)
                        : cm_banded_align_dropoff(e, pat + allow, read, (int)L, false, allow, (int)L - allow, false, &mep, &rml);
        if (num_errors > e || mep < 0) { num_errors = b_err; mep = b_mep; rml = b_rml; }
        else { gap_beginning = allow; mep += gap_beginning; rml += gap_beginning; }
      }
    } else {
      num_errors = pl ? cm_banded_align_dropoff_planes<true>(e, d.ref_pl, gpat + L + 2 * (uint32_t)e - 1, tp, d.read_pl_w, 0u, (int)L, &mep, &rml)
                      : cm_banded_align_dropoff(e, pat, read, (int)L, true, 0, (int)L, true, &mep, &rml);
      if (mep < 0 && allow > 0) {
        const int b_err = num_errors, b_mep = -mep, b_rml = rml;
        num_errors = pl ? cm_banded_align_dropoff_planes<true>(e, d.ref_pl, gpat + (L - (uint32_t)allow) + 2 * (uint32_t)e - 1, tp, d.read_pl_w, (uint32_t)allow,
                                                               (int)L - allow, &mep, &rml)
                        : cm_banded_align_dropoff(e, pat, read, (int)L, true, 0, (int)L - allow, true, &mep, &rml);
        if (num_errors > e || mep < 0) { num_errors = b_err; mep = b_mep; rml = b_rml; }
        else { gap_beginning = allow; mep += gap_beginning; rml += gap_beginning; }
      }
    }
    if (mep + 1 - e - num_errors - gap_beginning >= 30) {
      actual = num_errors;
      num_errors = -(mep - e - num_errors - gap_beginning);
    } else {
      num_errors = e + 1;
      actual = e + 1;
    }
    if (num_errors <= e) {
      if (num_errors < bst.min_err) {
        bst.second_err = bst.min_err; bst.n_second = bst.n_best; bst.min_err = num_errors; bst.n_best = 1;
        thr = nc > 50 ? cc[ci] : cc[ci] / 2;
      } else if (num_errors == bst.min_err) bst.n_best++;
      else if (num_errors == bst.second_err) bst.n_second++;
      else if (num_errors < bst.second_err) { bst.n_second = 1; bst.second_err = num_errors; }
      // - strand: --SAM keeps the non-split position rule (draft_mapping_generator.cc:535-547)
      dp[nd] = strand == 0 ? cp[ci] - (uint64_t)e + (uint64_t)(int64_t)mep
                           : (d.p.sam ? cp[ci] - (uint64_t)L + 1 - (uint64_t)e + (uint64_t)(int64_t)mep : cp[ci] - (uint64_t)(int64_t)gap_beginning);
      de[nd] = (int16_t)num_errors;  // -(matched length) in split mode
      ds[nd] = (uint32_t)(((actual & 0xff) << 24) | ((gap_beginning & 0xff) << 16) | (rml & 0xffff));
      ++nd;
    }
  }
  return nd;
}

// ---------------------------------------------------------------------------------------
// S5: per read -- DraftMappingGenerator::GenerateDraftMappings (draft_mapping_generator.cc:9-57)
//      incl. the all-minimizer shortcut (:72-157).  Draft mappings go to dpos/derr at the
//      same offsets as the filtered candidate lists.
// ---------------------------------------------------------------------------------------
CM_HD void cm_s5_verify(const CmDev &d, uint32_t r) {
  const uint32_t pair = r >> 1;
  d.ndp[r] = 0; d.ndn[r] = 0;
  const int e = d.p.e;
  CmBest bst = {e + 1, e + 1, 0, 0};
  if (d.alive[pair]) {
    const uint32_t L = d.rlen[r];
    const uint8_t *read = cm_read_ptr(d, r);
    uint64_t *pp = cm_f_pos(d, r), *np = cm_f_neg(d, r);
    uint8_t *pc = cm_f_pcnt(d, r), *nc = cm_f_ncnt(d, r);
    const uint32_t ncp = d.fcp[r], ncn = d.fcn[r];
    uint64_t *dpp = d.dpos + d.m_off[r], *dpn = d.dpos + d.m_off[r] + d.ncp[r] + d.resc_p[r];
    int16_t *dep = d.derr + d.m_off[r], *den = d.derr + d.m_off[r] + d.ncp[r] + d.resc_p[r];
    bool done = false;
    if (d.p.split) {  // draft_mapping_generator.cc:31-39: no shortcut, scalar drop-off verification
      uint32_t *dsp = d.dsplit + d.m_off[r], *dsn = d.dsplit + d.m_off[r] + d.ncp[r] + d.resc_p[r];
      cm_sort_cand(pp, pc, ncp);
      cm_sort_cand(np, nc, ncn);
      const uint32_t *tp = d.read_pl ? cm_read_pl_of(d, r) : nullptr;  // (the forward planes serve both strands)
      d.ndp[r] = cm_draft_strand_split(d, read, L, 0, pp, pc, ncp, bst, dpp, dep, dsp, tp);
      d.ndn[r] = cm_draft_strand_split(d, read, L, 1, np, nc, ncn, bst, dpn, den, dsn, tp);
      done = true;
    } else if (ncp + ncn == 1) {
      const int strand = ncp == 1 ? 0 : 1;
      const uint64_t cpos = strand == 0 ? pp[0] : np[0];
      const uint8_t cnt = strand == 0 ? pc[0] : nc[0];
      if ((uint32_t)cnt == d.mm_cnt[r]) {
        bst.min_err = 0; bst.n_best = 1; bst.n_second = 0;
        const uint32_t rid = (uint32_t)(cpos >> 32);
        const uint32_t position = strand == 0 ? (uint32_t)cpos : (uint32_t)cpos - L + 1;
        if (cm_valid_candidate(d, rid, position, L)) {
          if (strand == 0) { dpp[0] = cpos + L - 1; dep[0] = 0; d.ndp[r] = 1; }
          else { dpn[0] = cpos; den[0] = 0; d.ndn[r] = 1; }
          done = true;
        }
      }
    }
    if (!done) {
      cm_sort_cand(pp, pc, ncp);
      cm_sort_cand(np, nc, ncn);
      d.ndp[r] = cm_draft_strand(d, read, L, 0, pp, pc, ncp, bst, dpp, dep);
      d.ndn[r] = cm_draft_strand(d, read, L, 1, np, nc, ncn, bst, dpn, den);
    }
  }
  d.min_err[r] = bst.min_err; d.second_err[r] = bst.second_err;
  d.n_best[r] = bst.n_best; d.n_second[r] = bst.n_second;
}


// ---------------------------------------------------------------------------------------
// S5 as three kernels (non-split): (a) per read: shortcut, or sort the candidate lists and
// publish nv = number of candidates; (b) per candidate: the banded alignment, all lanes busy;
// (c) per read: the reference's sequential acceptance loop (lane grouping, count-threshold
// break, best/second-best bookkeeping) over the precomputed (errors, end) results.  Candidates
// beyond the loop's break point are aligned needlessly but never looked at.
// ---------------------------------------------------------------------------------------
// coop_min > 0: a read with more candidates than that gets its lists sorted and its acceptance loop run by a group of lanes
// (cm_coop_s5_sort / cm_coop_s5c, cm_coop.h; the alignments stay with the per-candidate kernel): the function returns true
CM_HD bool cm_s5a_prepare(const CmDev &d, uint32_t r, uint32_t coop_min = 0) {
  const uint32_t pair = r >> 1;
  d.nv[r] = 0;
  if (d.p.split || !d.alive[pair]) { cm_s5_verify(d, r); return false; }  // split alignment keeps the in-place path
  d.ndp[r] = 0; d.ndn[r] = 0;
  const int e = d.p.e;
  CmBest bst = {e + 1, e + 1, 0, 0};
  const uint32_t L = d.rlen[r];
  uint64_t *pp = cm_f_pos(d, r), *np = cm_f_neg(d, r);
  uint8_t *pc = cm_f_pcnt(d, r), *nc = cm_f_ncnt(d, r);
  const uint32_t ncp = d.fcp[r], ncn = d.fcn[r];
  bool done = false;
  if (ncp + ncn == 1) {
    const int strand = ncp == 1 ? 0 : 1;
    const uint64_t cpos = strand == 0 ? pp[0] : np[0];
    const uint8_t cnt = strand == 0 ? pc[0] : nc[0];
    if ((uint32_t)cnt == d.mm_cnt[r]) {
      bst.min_err = 0; bst.n_best = 1; bst.n_second = 0;
      const uint32_t rid = (uint32_t)(cpos >> 32);
      const uint32_t position = strand == 0 ? (uint32_t)cpos : (uint32_t)cpos - L + 1;
      if (cm_valid_candidate(d, rid, position, L)) {
        if (strand == 0) { d.dpos[d.m_off[r]] = cpos + L - 1; d.derr[d.m_off[r]] = 0; d.ndp[r] = 1; }
        else { const uint32_t o = d.m_off[r] + d.ncp[r] + d.resc_p[r]; d.dpos[o] = cpos; d.derr[o] = 0; d.ndn[r] = 1; }
        done = true;
      }
    }
  }
  bool to_group = false;
  if (!done) {
    to_group = coop_min > 0 && ncp + ncn > coop_min;  // a group sorts these lists (cm_coop_sort_cand) before the alignments
    if (!to_group) {
      cm_sort_cand(pp, pc, ncp);
      cm_sort_cand(np, nc, ncn);
    }
    d.nv[r] = ncp + ncn;
  }
  d.min_err[r] = bst.min_err; d.second_err[r] = bst.second_err;
  d.n_best[r] = bst.n_best; d.n_second[r] = bst.n_second;
  return to_group;
}

CM_HD void cm_s5b_verify_at(const CmDev &d, uint32_t r, int strand, uint32_t ci);
// work item j -> (read, strand, candidate): binary search in the exclusive prefix v_off
CM_HD void cm_s5b_verify_item(const CmDev &d, uint32_t j, uint32_t n_reads) {
  uint32_t lo = 0, hi = n_reads;  // largest r with v_off[r] <= j
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (d.v_off[mid] <= j) lo = mid; else hi = mid;
  }
  const uint32_t r = lo, li = j - d.v_off[r];
  const uint32_t ncp = d.fcp[r];
  const int strand = li < ncp ? 0 : 1;
  const uint32_t ci = strand ? li - ncp : li;
  cm_s5b_verify_at(d, r, strand, ci);
}
// candidate ci of read r's strand list: the banded alignment, result to v_err / v_end
CM_HD void cm_s5b_verify_at(const CmDev &d, uint32_t r, int strand, uint32_t ci) {
  const uint32_t o = d.m_off[r] + (strand ? d.ncp[r] + d.resc_p[r] : 0) + ci;
  const uint64_t cpos = d.fbuf[o];
  const uint32_t L = d.rlen[r];
  const uint32_t rid = (uint32_t)(cpos >> 32);
  const uint32_t position = strand == 0 ? (uint32_t)cpos : (uint32_t)cpos - L + 1;
  if (!cm_valid_candidate(d, rid, position, L)) { d.v_err[o] = CM_V_INVALID; d.v_end[o] = 0; return; }
  int end_pos = (int)L;
  int ne;
  if (d.ref_pl && d.read_pl)  // the same alignment on bit planes (cm_banded_align_planes)
    ne = cm_banded_align_planes(d.p.e, d.ref_pl, d.ref_off[rid] + position - (uint32_t)d.p.e,
                                cm_read_pl_of(d, r) + (size_t)strand * 3 * d.read_pl_w, d.read_pl_w, (int)L, &end_pos);
  else ne = cm_verify_compute(d, cm_read_ptr(d, r), L, strand, cpos, &end_pos);
  d.v_err[o] = (int16_t)ne;
  d.v_end[o] = (int16_t)end_pos;
}

CM_HD void cm_s5c_accept(const CmDev &d, uint32_t r);
// coop_min > 0: the reads cm_s5a_prepare left to the groups are skipped (same test)
CM_HD void cm_s5c_finalize(const CmDev &d, uint32_t r, uint32_t coop_min = 0) {
  if (d.nv[r] == 0) return;
  if (coop_min > 0 && d.nv[r] > coop_min) return;
  cm_s5c_accept(d, r);
}
CM_HD void cm_s5c_accept(const CmDev &d, uint32_t r) {
  CmBest bst = {d.min_err[r], d.second_err[r], d.n_best[r], d.n_second[r]};
  const uint32_t L = d.rlen[r];
  const uint8_t *read = cm_read_ptr(d, r);
  const uint32_t op = d.m_off[r], on = d.m_off[r] + d.ncp[r] + d.resc_p[r];
  d.ndp[r] = cm_draft_strand(d, read, L, 0, d.fbuf + op, d.fcnt + op, d.fcp[r], bst, d.dpos + op, d.derr + op, d.v_err + op, d.v_end + op);
  d.ndn[r] = cm_draft_strand(d, read, L, 1, d.fbuf + on, d.fcnt + on, d.fcn[r], bst, d.dpos + on, d.derr + on, d.v_err + on, d.v_end + on);
  d.min_err[r] = bst.min_err; d.second_err[r] = bst.second_err;
  d.n_best[r] = bst.n_best; d.n_second[r] = bst.n_second;
}

// ---------------------------------------------------------------------------------------
// K5: best pair, coordinates, MAPQ
// ---------------------------------------------------------------------------------------
struct CmPe { int min_sum, second_sum, n_best, n_second; uint32_t f_dir, f_i1, f_i2; };

// GenerateBestMappingsForPairedEndReadOnOneDirection, non-split (mapping_generator.h:347-484).
// When want >= 0, stops at the want-th pair (0-based, counted across directions via *seen)
// whose error sum equals final_min and returns it in (f_dir,f_i1,f_i2).
CM_HD bool cm_pair_dir(const CmDev &d, int dir, const uint64_t *ap, const int16_t *ae, uint32_t na,
                       const uint64_t *bp, const int16_t *be, uint32_t nb, uint32_t len1, uint32_t len2, CmPe &pe,
                       int64_t want, int final_min, int64_t *seen) {
  const uint64_t I = (uint64_t)(int64_t)d.p.max_insert;
  const uint64_t mo = (uint32_t)d.p.min_read_len;
  uint32_t i1 = 0, i2 = 0;
  while (i1 < na && i2 < nb) {
    const uint64_t p1 = ap[i1], p2 = bp[i2];
    if ((dir == 1 && p1 > p2 + I - len2) || (dir == 0 && p1 > p2 + len1 - mo)) {
      ++i2;
    } else if ((dir == 0 && p2 > p1 + I - len1) || (dir == 1 && p2 > p1 + len2 - mo)) {
      ++i1;
    } else {
      uint32_t cur = i2;
      while (cur < nb && ((dir == 0 && bp[cur] <= p1 + I - len1) || (dir == 1 && bp[cur] <= p1 + len2 - mo))) {
        const int s = (int)ae[i1] + (int)be[cur];
        if (want >= 0) {
          if (s == final_min) {
            if (*seen == want) { pe.f_dir = (uint32_t)dir; pe.f_i1 = i1; pe.f_i2 = cur; return true; }
            ++*seen;
          }
        } else if (s < pe.min_sum) {
          pe.second_sum = pe.min_sum; pe.n_second = pe.n_best; pe.min_sum = s; pe.n_best = 1;
          pe.f_dir = (uint32_t)dir; pe.f_i1 = i1; pe.f_i2 = cur;
        } else if (s == pe.min_sum) {
          pe.n_best++;
        } else if (s == pe.second_sum) {
          pe.n_second++;
        } else if (s < pe.second_sum) {
          pe.second_sum = s; pe.n_second = 1;
        }
        ++cur;
      }
      ++i1;
    }
  }
  return false;
}

struct CmSpan { uint32_t rid, ref_start, ref_end; };
// GetRefStartEndPositionForReadFromMapping, non-SAM non-split (mapping_generator.h:657-717,
// 762-793, 855-916).  The reference reads up to e bytes past the end of a chromosome when
// ref_pos + e >= length (:703-708); HBM holds zero padding there (kseq's NUL, then zeros).
CM_HD CmSpan cm_ref_start_end(const CmDev &d, uint64_t dpos, int nerr, int strand, const uint8_t *read, int L) {
  const int e = d.p.e;
  const uint32_t rid = (uint32_t)(dpos >> 32), ref_pos = (uint32_t)dpos;
  const uint32_t rl = d.ref_len[rid];
  uint32_t vw = ref_pos + 1 > (uint32_t)(L + e) ? ref_pos + 1 - (uint32_t)L - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)L;
  const int start = cm_banded_traceback(e, nerr, d.ref + d.ref_off[rid] + vw, read, L, strand == 1, 0, L);
  CmSpan s;
  s.rid = rid;
  s.ref_start = vw + (uint32_t)start;
  s.ref_end = ref_pos;
  return s;
}

// ---------------------------------------------------------------------------------------
// --SAM: ksw_semi_global3 (ksw.cc:505-626) as chromap calls it (mapping_generator.h:723-738,
// 807-821): query = reference window of L + 2e bytes, target = the read as mapped (L bases),
// band w = 2e + 1, scores match 1 / mismatch -4 / gap open 6 + extend 1 / ambiguous 0
// (mapping_parameters.h:20-23).  Row i visits columns [i, i + w + 1) (the last row one fewer),
// so the H / E state is a window of W1 = w + 1 registers that shifts by one column per row.
// The move bits of every cell go to sam_z (4 cells per word, interleaved over pairs so a wave's
// stores coalesce); the backtrack walks them from the best of the last w columns.
// W1T: compile-time window size (>= w + 1).
// ---------------------------------------------------------------------------------------
template <int W1T>
// The target is text[toff .. toff + L) of the read as mapped (neg: reverse complement of read[0 .. Lfull)); Lfull < 0: the
// whole read (Lfull = L, toff = 0).
CM_HD int cm_ksw_sg3(const uint8_t *query, const uint8_t *read, int L, bool neg, int e, uint32_t *z, uint64_t zstride,
                     uint32_t *cigar, int *n_cigar, int *start, int *end, int Lfull = -1, int toff = 0) {
  if (Lfull < 0) Lfull = L;
  const int NEG = -0x40000000, o_del = 6, e_del = 1, e_ins = 1, oe_del = 7, oe_ins = 7;
  const int w = 2 * e + 1, w1 = w + 1, qlen = L + 2 * e;
  constexpr int ZW = (W1T + 3) / 4;
  int h[W1T], ev[W1T];
  uint32_t qc[W1T];
#pragma unroll
  for (int jj = 0; jj < W1T; ++jj) { h[jj] = 0; ev[jj] = NEG; qc[jj] = jj < w1 ? cm_c2u(query[jj]) : 4u; }
  for (int i = 0; i < L; ++i) {
    int f = NEG;
    int h1 = i == 0 ? -(o_del + e_del) : NEG;
    const uint32_t tc = cm_text_code(read, Lfull, toff + i, neg);
    const int cnt = i == L - 1 ? w1 - 1 : w1;
    uint32_t packed = 0;
#pragma unroll
    for (int jj = 0; jj < W1T; ++jj) {
      if (jj < cnt) {
        const int sc = (tc > 3 || qc[jj] > 3) ? 0 : (tc == qc[jj] ? 1 : -4);
        int m = h[jj], ee = ev[jj];
        h[jj] = h1;
        m += sc;
        uint32_t dbits = m >= ee ? 0u : 1u;
        int hh = m >= ee ? m : ee;
        dbits = hh >= f ? dbits : 2u;
        hh = hh >= f ? hh : f;
        h1 = hh;
        int t = m - oe_del;
        ee -= e_del;
        dbits |= ee > t ? 4u : 0u;
        ee = ee > t ? ee : t;
        ev[jj] = ee;
        t = m - oe_ins;
        f -= e_ins;
        dbits |= f > t ? 32u : 0u;
        f = f > t ? f : t;
        packed |= dbits << ((jj & 3) * 8);
      }
      if ((jj & 3) == 3 || jj == W1T - 1) {
        if ((jj >> 2) * 4 < cnt) z[((uint64_t)i * ZW + (uint32_t)(jj >> 2)) * zstride] = packed;
        packed = 0;
      }
    }
    if (i != L - 1) {
      const uint32_t nq = i + w1 < qlen ? cm_c2u(query[i + w1]) : 4u;
#pragma unroll
      for (int jj = 0; jj < W1T; ++jj) {
        const bool top = jj == w1 - 1;
        h[jj] = top ? h1 : (jj + 1 < W1T ? h[jj + 1] : NEG);
        ev[jj] = top ? NEG : (jj + 1 < W1T ? ev[jj + 1] : NEG);
        qc[jj] = top ? nq : (jj + 1 < W1T ? qc[jj + 1] : 4u);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < W1T; ++jj) if (jj == w1 - 1) h[jj] = h1;
    }
  }
  // best of the last w columns: H[qlen - j] = h[w1 - 1 - j], j = 0 .. w-1, first maximum wins
  int score = NEG, maxpos = qlen;
#pragma unroll
  for (int jj = W1T - 1; jj >= 1; --jj) {
    if (jj <= w1 - 1) {
      const int j = w1 - 1 - jj;
      if (j == 0 || h[jj] > score) { score = h[jj]; maxpos = qlen - j; }
    }
  }
  *end = maxpos;
  int n = 0, i = L - 1, k = maxpos - 1;
  uint32_t which = 0;
  bool overflow = false;
#define CM_PUSHC(op, len) do { if (n == 0 || (cigar[n - 1] & 0xfu) != (uint32_t)(op)) { if (n < CM_SAM_CIGAR_CAP) cigar[n++] = ((uint32_t)(len) << 4) | (uint32_t)(op); else overflow = true; } \
                               else cigar[n - 1] += (uint32_t)(len) << 4; } while (0)
  while (i >= 0 && k >= 0) {
    const int jj = k - i;
    const uint32_t cell = (z[((uint64_t)i * ZW + (uint32_t)(jj >> 2)) * zstride] >> ((jj & 3) * 8)) & 0xffu;
    which = (cell >> (which << 1)) & 3u;
    if (which == 0) { CM_PUSHC(0, 1); --i; --k; }
    else if (which == 1) { CM_PUSHC(1, 1); --i; }
    else { CM_PUSHC(2, 1); --k; }
  }
  if (i >= 0) CM_PUSHC(1, i + 1);
#undef CM_PUSHC
  *start = k + 1;
  for (int a = 0; a < n >> 1; ++a) { const uint32_t t = cigar[a]; cigar[a] = cigar[n - 1 - a]; cigar[n - 1 - a] = t; }
  *n_cigar = n;
  return overflow ? (int)0x80000000 : score;  // (not -1: a 39-base read with 8 mismatches scores exactly -1)
}

CM_HD uint32_t cm_put_dec(uint8_t *dst, uint32_t cap, uint32_t at, uint32_t v) {
  uint32_t dgt = 1;
  for (uint32_t t = v; t >= 10; t /= 10) ++dgt;
  for (uint32_t i = dgt; i-- > 0;) { if (at + i < cap) dst[at + i] = (uint8_t)('0' + v % 10); v /= 10; }
  return at + dgt;
}

// GenerateNMAndMDTag (alignment.cc:85-139); ref points at the mapping start, the read is taken as mapped
CM_HD uint32_t cm_nm_and_md(const uint8_t *ref, const uint8_t *read, int L, bool neg, const uint32_t *cigar, int n_cigar,
                            uint8_t *md, uint32_t md_cap, uint32_t *md_len, int toff = 0) {
  uint32_t nm = 0, matches = 0, at = 0;
  int rp = toff, fp = 0;  // L = full read length; the alignment starts at text[toff]
  for (int ci = 0; ci < n_cigar; ++ci) {
    const uint32_t op = cigar[ci] & 0xfu, len = cigar[ci] >> 4;
    if (op == 0) {
      for (uint32_t t = 0; t < len; ++t) {
        const uint8_t rc = ref[fp], tc = cm_text_char(read, L, rp, neg);
        if (rc == tc || (int)rc - 'a' + 'A' == (int)tc) ++matches;
        else { ++nm; at = cm_put_dec(md, md_cap, at, matches); matches = 0; if (at < md_cap) md[at] = rc; ++at; }
        ++fp; ++rp;
      }
    } else if (op == 1) { nm += len; rp += (int)len; }
    else if (op == 2) {
      nm += len;
      at = cm_put_dec(md, md_cap, at, matches); matches = 0;
      if (at < md_cap) md[at] = '^';
      ++at;
      for (uint32_t t = 0; t < len; ++t) { if (at < md_cap) md[at] = ref[fp]; ++at; ++fp; }
    }
  }
  at = cm_put_dec(md, md_cap, at, matches);
  *md_len = at;
  return nm;
}

struct CmSamAln { uint32_t n_cigar, md_len, nm; bool overflow; };
// GetRefStartEndPositionForReadFromMapping, SAM branches (mapping_generator.h:657-717, 723-761, 807-854), no split
CM_HD CmSpan cm_ref_start_end_sam(const CmDev &d, uint32_t pair, uint32_t slot, uint64_t dpos, int strand, const uint8_t *read, int L,
                                  CmSamAln *aln) {
  const int e = d.p.e;
  const uint32_t rid = (uint32_t)(dpos >> 32), ref_pos = (uint32_t)dpos;
  const uint32_t rl = d.ref_len[rid];
  uint32_t vw = ref_pos + 1 > (uint32_t)(L + e) ? ref_pos + 1 - (uint32_t)L - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)L;
  const uint8_t *win = d.ref + d.ref_off[rid] + vw;
  uint32_t *cigar = d.sam_cigar + (uint64_t)slot * CM_SAM_CIGAR_CAP;
  int n_cigar = 0, st = 0, en = 0, sc;
  if (2 * e + 2 <= 18) sc = cm_ksw_sg3<18>(win, read, L, strand == 1, e, d.sam_z + pair, d.n_pairs, cigar, &n_cigar, &st, &en);
  else sc = cm_ksw_sg3<32>(win, read, L, strand == 1, e, d.sam_z + pair, d.n_pairs, cigar, &n_cigar, &st, &en);
  uint32_t md_len = 0;
  aln->nm = cm_nm_and_md(win + st, read, L, strand == 1, cigar, n_cigar, d.sam_md + (uint64_t)slot * d.sam_md_cap, d.sam_md_cap, &md_len);
  aln->n_cigar = (uint32_t)n_cigar;
  aln->md_len = md_len;
  aln->overflow = sc == (int)0x80000000 || md_len > d.sam_md_cap;
  CmSpan s;
  s.rid = rid;
  s.ref_start = vw + (uint32_t)st;
  s.ref_end = vw + (uint32_t)en - 1;
  return s;
}

// GetRefStartEndPositionForReadFromMapping, split + SAM branches (mapping_generator.h:657-761 for +, 806-850 for -): ksw on
// the aligned part of the read (the split site pulled in by 3e when the read was cut, :711-717), AdjustGapBeginning extending
// the first / last M of the CIGAR over the matching bases of the gap (alignment.cc:24-83), NM / MD over the extended
// alignment.  The window start is the one of the unshortened part, and the - strand hands AdjustGapBeginning reference
// coordinates without read_start_site -- both as the reference has them.
CM_HD CmSpan cm_ref_start_end_split_sam(const CmDev &d, uint32_t pair, uint32_t slot, uint64_t dpos, uint32_t split_word, int strand,
                                        const uint8_t *read, int full_len, CmSamAln *aln) {
  const int e = d.p.e;
  const uint32_t rid = (uint32_t)(dpos >> 32), ref_pos = (uint32_t)dpos;
  const uint32_t rl = d.ref_len[rid];
  const uint8_t *ref = d.ref + d.ref_off[rid];
  int split_site = (int)(split_word & 0xffff);
  int gap_beginning = (int)((split_word >> 16) & 0xff);
  int read_length = split_site - gap_beginning;
  uint32_t vw = ref_pos + 1 > (uint32_t)(read_length + e) ? ref_pos + 1 - (uint32_t)read_length - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)read_length;
  if (split_site < full_len && split_site > 3 * e) split_site -= 3 * e;
  read_length = split_site - gap_beginning;
  uint32_t *cigar = d.sam_cigar + (uint64_t)slot * CM_SAM_CIGAR_CAP;
  int n_cigar = 0, st = 0, en = 0, sc;
  uint32_t md_len = 0;
  CmSpan s;
  s.rid = rid;
  if (strand == 0) {
    if (2 * e + 2 <= 18) sc = cm_ksw_sg3<18>(ref + vw, read, read_length, false, e, d.sam_z + pair, d.n_pairs, cigar, &n_cigar, &st, &en, full_len, gap_beginning);
    else sc = cm_ksw_sg3<32>(ref + vw, read, read_length, false, e, d.sam_z + pair, d.n_pairs, cigar, &n_cigar, &st, &en, full_len, gap_beginning);
    if (gap_beginning > 0) {
      const int rs = (int)vw + st;
      const int nrs = cm_adjust_gap_beginning(0, ref, rl, read, full_len, false, 0, &gap_beginning, read_length - 1, rs, (int)vw + en - 1);
      if (n_cigar > 0 && (cigar[0] & 0xfu) == 0) cigar[0] += (uint32_t)(rs - nrs) << 4;
      st = nrs - (int)vw;
    }
    aln->nm = cm_nm_and_md(ref + vw + st, read, full_len, false, cigar, n_cigar, d.sam_md + (uint64_t)slot * d.sam_md_cap, d.sam_md_cap, &md_len, gap_beginning);
    s.ref_start = vw + (uint32_t)st;
    s.ref_end = vw + (uint32_t)en - 1;
  } else {
    const int rss = full_len - split_site;  // read_start_site
    if (2 * e + 2 <= 18) sc = cm_ksw_sg3<18>(ref + vw + rss, read, read_length, true, e, d.sam_z + pair, d.n_pairs, cigar, &n_cigar, &st, &en, full_len, rss);
    else sc = cm_ksw_sg3<32>(ref + vw + rss, read, read_length, true, e, d.sam_z + pair, d.n_pairs, cigar, &n_cigar, &st, &en, full_len, rss);
    if (gap_beginning > 0) {
      const int re = (int)vw + en - 1;
      const int nre = cm_adjust_gap_beginning(1, ref, rl, read, full_len, true, rss, &gap_beginning, read_length - 1, (int)vw + st, re);
      if (n_cigar > 0 && (cigar[n_cigar - 1] & 0xfu) == 0) cigar[n_cigar - 1] += (uint32_t)(nre - re) << 4;
      en = nre + 1 - (int)vw - rss;
    }
    aln->nm = cm_nm_and_md(ref + vw + rss + st, read, full_len, true, cigar, n_cigar, d.sam_md + (uint64_t)slot * d.sam_md_cap, d.sam_md_cap, &md_len, rss);
    s.ref_start = vw + (uint32_t)rss + (uint32_t)st;
    s.ref_end = vw + (uint32_t)rss + (uint32_t)en - 1;
  }
  aln->n_cigar = (uint32_t)n_cigar;
  aln->md_len = md_len;
  aln->overflow = sc == (int)0x80000000 || md_len > d.sam_md_cap;
  return s;
}

// one cmgpu_sam_record (40 bytes; include/chromap_amd.h)
CM_HD void cm_put_sam_record(const CmDev &d, uint32_t slot, uint32_t read_id, const CmSpan &me, uint32_t mpos, int32_t mrid, int32_t tlen,
                             uint32_t flag, uint8_t mapq, uint8_t plus, uint8_t is_unique, const CmSamAln &a, uint32_t len_after_trim) {
  uint32_t *o32 = reinterpret_cast<uint32_t *>(d.sam_rec + (uint64_t)slot * 40);
  uint16_t *o16 = reinterpret_cast<uint16_t *>(o32);
  uint8_t *o8 = reinterpret_cast<uint8_t *>(o32);
  o32[0] = read_id; o32[1] = me.rid; o32[2] = me.ref_start; o32[3] = mpos; o32[4] = (uint32_t)mrid; o32[5] = (uint32_t)tlen; o32[6] = a.nm;
  o16[14] = (uint16_t)flag; o16[15] = (uint16_t)a.n_cigar; o16[16] = (uint16_t)a.md_len;
  o8[34] = mapq; o8[35] = plus; o8[36] = is_unique; o8[37] = a.overflow ? 2 : 1;
  o16[19] = (uint16_t)len_after_trim;
}

// (int)(4.343 * log(n + 1) + 0.499) for n >= 1 via host-computed breakpoints:
// nsec_break[v] = smallest n whose value is >= v (monotone step function)
CM_HD int cm_nsec_penalty(const CmMapqTables &t, int n) {
  int v = 0;
  for (int i = 0; i < t.n_break; ++i) {
    if ((uint32_t)n >= t.nsec_break[i]) v = i; else break;
  }
  return v;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define CM_SQRT(x) __dsqrt_rn(x)
#else
#include <math.h>
#define CM_SQRT(x) sqrt(x)
#endif

// GetMAPQForSingleEndRead, non-split (mapping_generator.h:920-1022)
CM_HD uint8_t cm_mapq_single(const CmDev &d, int num_errors, uint16_t alignment_length, int read_length, int max_diff,
                             int second_err, int n_best, int n_second, uint32_t rep_len) {
  alignment_length = alignment_length > read_length ? alignment_length : (uint16_t)read_length;
  const double alignment_identity = 1 - (double)num_errors / alignment_length;
  int mapq = 0;
  int second = second_err;
  if (n_best > 1) {
  } else {
    if (second > num_errors + max_diff) second = num_errors + max_diff;
    double tmp = d.mq.len_coef[alignment_length];
    tmp *= alignment_identity * alignment_identity;
    mapq = (int)(5 * 6.02 * (second - num_errors) * tmp * tmp + 0.499);
  }
  if (n_second > 0) mapq -= cm_nsec_penalty(d.mq, n_second);
  if (mapq > 60) mapq = 60;
  if (mapq < 0) mapq = 0;
  if (rep_len > 0) {
    double frac_rep = rep_len / (double)read_length;
    if (rep_len >= (uint32_t)read_length) frac_rep = 0.999;
    if (alignment_identity <= 0.95) mapq = (int)(mapq * (1 - CM_SQRT(frac_rep)) + 0.499);
    else if (alignment_identity <= 0.97) mapq = (int)(mapq * (1 - frac_rep) + 0.499);
    else if (alignment_identity >= 0.999) mapq = (int)(mapq * (1 - frac_rep * frac_rep * frac_rep * frac_rep) + 0.499);
    else mapq = (int)(mapq * (1 - frac_rep * frac_rep) + 0.499);
  }
  return (uint8_t)mapq;
}

// GetMAPQForPairedEndRead, non-split (mapping_generator.h:1027-1192)
CM_HD uint8_t cm_mapq_paired(const CmDev &d, uint32_t pair, int err1, int err2, uint16_t al1, uint16_t al2, int len1,
                             int len2, int force_mapq, const CmPe &pe) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  uint8_t mapq_pe = 0;
  const int min_unpaired = d.min_err[r1] + d.min_err[r2] + 3;
  if (pe.n_best <= 1) {
    const int adj = pe.second_sum < min_unpaired ? pe.second_sum : min_unpaired;
    mapq_pe = (uint8_t)((int)(5 * 6.02 * (adj - pe.min_sum) / (1) + .499));
    if (pe.n_second > 0) mapq_pe = (uint8_t)(mapq_pe - cm_nsec_penalty(d.mq, pe.n_second));
    if (mapq_pe > 60) mapq_pe = 60;
    const int rep = (int)(d.rep_len[r1] + d.rep_len[r2]);
    if (rep > 0) {
      const double total = len1 + len2;
      double frac_rep = (double)rep / total;
      if (rep >= total) frac_rep = 0.999;
      const double id1 = 1 - (double)err1 / (len1 > al1 ? len1 : al1);
      const double id2 = 1 - (double)err2 / (len2 > al2 ? len2 : al2);
      const double id = id1 < id2 ? id1 : id2;
      if (id <= 0.95) mapq_pe = (uint8_t)(mapq_pe * (1 - CM_SQRT(frac_rep)) + 0.499);
      else if (id <= 0.97) mapq_pe = (uint8_t)(mapq_pe * (1 - frac_rep) + 0.499);
      else if (id >= 0.999) mapq_pe = (uint8_t)(mapq_pe * (1 - frac_rep * frac_rep * frac_rep * frac_rep) + 0.499);
      else mapq_pe = (uint8_t)(mapq_pe * (1 - frac_rep * frac_rep) + 0.499);
    }
  }
  uint8_t mapq1 = cm_mapq_single(d, err1, al1, len1, 2, d.second_err[r1], d.n_best[r1], d.n_second[r1], d.rep_len[r1]);
  uint8_t mapq2 = cm_mapq_single(d, err2, al2, len2, 2, d.second_err[r2], d.n_best[r2], d.n_second[r2], d.rep_len[r2]);
  mapq1 = (uint8_t)(mapq1 > mapq_pe ? (double)mapq1 : mapq_pe < mapq1 + mapq_pe * 0.65 ? (double)mapq_pe : mapq1 + mapq_pe * 0.65);
  mapq2 = (uint8_t)(mapq2 > mapq_pe ? (double)mapq2 : mapq_pe < mapq2 + mapq_pe * 0.65 ? (double)mapq_pe : mapq2 + mapq_pe * 0.65);
  mapq1 = (uint8_t)(mapq1 * 1.2);
  if (mapq1 > 60) mapq1 = 60;
  mapq2 = (uint8_t)(mapq2 * 1.2);
  if (mapq2 > 60) mapq2 = 60;
  uint8_t mapq = mapq1 < mapq2 ? mapq1 : mapq2;
  if (mapq < 60 && force_mapq >= 0 && force_mapq < mapq) mapq = (uint8_t)force_mapq;
  return mapq;
}

CM_HD const uint64_t *cm_d_pos(const CmDev &d, uint32_t r, int strand) {
  return d.dpos + d.m_off[r] + (strand ? d.ncp[r] + d.resc_p[r] : 0);
}
CM_HD const int16_t *cm_d_err(const CmDev &d, uint32_t r, int strand) {
  return d.derr + d.m_off[r] + (strand ? d.ncp[r] + d.resc_p[r] : 0);
}

// EmplaceBackPairedEndMappingRecord<PairsMapping> (mapping_generator.cc:169-210): a read's position is its reference start on the
// + strand and its reference END on the - strand; the end with the smaller (rank of the sequence, position) goes first.
// Record layout = cmgpu_pairs_record (24 bytes).  a / b: read 1 / read 2 of the pair, s1 / s2 their strands (0: +).
CM_HD void cm_put_pairs_record(const CmDev &d, uint64_t slot, uint32_t pair, const CmSpan &a, const CmSpan &b, int s1, int s2, uint8_t mapq, uint8_t is_unique) {
  uint8_t st1 = s1 == 0 ? 1 : 0, st2 = s2 == 0 ? 1 : 0;
  int pos1 = (int)(s1 == 0 ? a.ref_start : a.ref_end), pos2 = (int)(s2 == 0 ? b.ref_start : b.ref_end);
  int rid1 = (int)a.rid, rid2 = (int)b.rid;
  const uint32_t k1 = d.pairs_rank ? d.pairs_rank[rid1] : (uint32_t)rid1, k2 = d.pairs_rank ? d.pairs_rank[rid2] : (uint32_t)rid2;
  if (!(k1 < k2 || (rid1 == rid2 && pos1 < pos2))) {  // mapping_generator.cc:193-203
    int t = rid1; rid1 = rid2; rid2 = t;
    t = pos1; pos1 = pos2; pos2 = t;
    const uint8_t u = st1; st1 = st2; st2 = u;
  }
  uint8_t *o8 = d.rec + slot * 24;
  uint32_t *o32 = reinterpret_cast<uint32_t *>(o8);
  o32[0] = d.first_read_id + pair;
  o32[1] = (uint32_t)rid1;
  o32[2] = (uint32_t)rid2;
  o32[3] = (uint32_t)pos1;
  o32[4] = (uint32_t)pos2;
  o8[20] = st1; o8[21] = st2; o8[22] = mapq; o8[23] = is_unique;
  d.rec_ok[slot] = 1;
}

// Build the output record for the chosen best pair: ProcessBestMappingsForPairedEndReadOn-
// OneDirection (mapping_generator.h:487-653) + EmplaceBackPairedEndMappingRecord
// (mapping_generator.cc:111-125) + PairedEndMappingInMemory getters (mapping_in_memory.h:64-108)
template <bool SAM>
CM_HD void cm_emit_record(const CmDev &d, uint32_t pair, const CmPe &pe, uint32_t nth = 0) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  const uint64_t slot = (uint64_t)pair * (uint32_t)d.p.max_best + nth;  // max_best record slots per pair
  const int dir = (int)pe.f_dir;
  const uint32_t len1 = d.rlen[r1], len2 = d.rlen[r2];
  const int s1 = dir == 0 ? 0 : 1, s2 = dir == 0 ? 1 : 0;
  const uint64_t dp1 = cm_d_pos(d, r1, s1)[pe.f_i1], dp2 = cm_d_pos(d, r2, s2)[pe.f_i2];
  const int e1 = cm_d_err(d, r1, s1)[pe.f_i1], e2 = cm_d_err(d, r2, s2)[pe.f_i2];
  CmSamAln sa1, sa2;
  CmSpan a, b;
  if constexpr (SAM) {
    a = cm_ref_start_end_sam(d, pair, r1, dp1, s1, cm_read_ptr(d, r1), (int)len1, &sa1);
    b = cm_ref_start_end_sam(d, pair, r2, dp2, s2, cm_read_ptr(d, r2), (int)len2, &sa2);
  } else {
    a = cm_ref_start_end(d, dp1, e1, s1, cm_read_ptr(d, r1), (int)len1);
    b = cm_ref_start_end(d, dp2, e2, s2, cm_read_ptr(d, r2), (int)len2);
  }
  const uint16_t al1 = (uint16_t)(a.ref_end - a.ref_start + 1), al2 = (uint16_t)(b.ref_end - b.ref_start + 1);
  const int force_mapq = d.force0[pair] ? 0 : -1;
  const uint8_t mapq = cm_mapq_paired(d, pair, e1, e2, al1, al2, (int)len1, (int)len2, force_mapq, pe);
  const uint8_t is_unique = (pe.n_best == 1 || d.n_best[r1] == 1 || d.n_best[r2] == 1) ? 1 : 0;
  if (!SAM && d.p.pairs_out) {  // --pairs without split alignment: the same pairing as a PairsMapping (mapq: the pair's)
    cm_put_pairs_record(d, slot, pair, a, b, s1, s2, mapq, is_unique);
    return;
  }
  const CmSpan &ps = dir == 0 ? a : b;  // the + strand read
  const CmSpan &ns = dir == 0 ? b : a;
  uint8_t *o = d.rec + slot * 24;
  uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
  uint16_t *o16 = reinterpret_cast<uint16_t *>(o);
  o32[0] = d.first_read_id + pair;
  o32[1] = a.rid;
  o32[2] = ps.ref_start;
  o16[6] = (uint16_t)(int)(ns.ref_end - ps.ref_start + 1);
  o[14] = mapq & 63;
  o[15] = dir == 0 ? 1 : 0;
  o[16] = is_unique;
  o[17] = 1;
  o16[9] = (uint16_t)(ps.ref_end - ps.ref_start + 1);
  o16[10] = (uint16_t)(ns.ref_end - ns.ref_start + 1);
  o16[11] = 0;
  d.rec_ok[slot] = 1;
  if constexpr (SAM) {  // EmplaceBackPairedEndMappingRecord<SAMMapping> (mapping_generator.cc:84-108), flags mapping_generator.h:613-631
    const int tlen = (int)(ns.ref_end - ps.ref_start + 1);
    const bool plus1 = dir == 0;
    cm_put_sam_record(d, r1, d.first_read_id + pair, a, b.ref_start, (int32_t)b.rid, plus1 ? tlen : -tlen, 3u | (plus1 ? 32u : 16u) | 64u,
                      mapq, plus1 ? 1 : 0, is_unique, sa1, len1);
    cm_put_sam_record(d, r2, d.first_read_id + pair, b, a.ref_start, (int32_t)a.rid, plus1 ? -tlen : tlen, 3u | (plus1 ? 16u : 32u) | 128u,
                      mapq, plus1 ? 0 : 1, is_unique, sa2, len2);
  }
}


// ---------------------------------------------------------------------------------------
// split alignment (--preset hic): coordinates, MAPQ, pairs record
// ---------------------------------------------------------------------------------------
// GetRefStartEndPositionForReadFromMapping, split + non-SAM branches
// (mapping_generator.h:657-717, 762-793, 855-916)
CM_HD CmSpan cm_ref_start_end_split(const CmDev &d, uint64_t dpos, uint32_t split_word, int strand, const uint8_t *read,
                                    int full_len) {
  const int e = d.p.e;
  const uint32_t rid = (uint32_t)(dpos >> 32), ref_pos = (uint32_t)dpos;
  const uint32_t rl = d.ref_len[rid];
  const uint8_t *ref = d.ref + d.ref_off[rid];
  const int split_site = (int)(split_word & 0xffff);
  int gap_beginning = (int)((split_word >> 16) & 0xff);
  const int actual = (int)((split_word >> 24) & 0xff);
  const int read_length = split_site - gap_beginning;
  uint32_t vw = ref_pos + 1 > (uint32_t)(read_length + e) ? ref_pos + 1 - (uint32_t)read_length - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)read_length;
  CmSpan s;
  s.rid = rid;
  if (strand == 0) {
    int start = cm_banded_traceback(e, actual, ref + vw, read, full_len, false, gap_beginning, read_length);
    if (gap_beginning > 0) {
      const int nrs = cm_adjust_gap_beginning(0, ref, rl, read, full_len, false, 0, &gap_beginning, read_length - 1,
                                              (int)vw + start, (int)ref_pos);
      start = nrs - (int)vw;
    }
    s.ref_start = vw + (uint32_t)start;
    s.ref_end = ref_pos;
    return s;
  }
  const int read_start_site = full_len - split_site;
  const int start = e;
  int mep = (int)(ref_pos - vw + 1);
  cm_banded_align(e, ref + vw, read, full_len, true, read_start_site, read_length, &mep);
  mep += 1;
  if (gap_beginning > 0) {
    const int nre = cm_adjust_gap_beginning(1, ref, rl, read, full_len, true, read_start_site, &gap_beginning, read_length - 1,
                                            (int)vw + start, (int)vw + mep);
    mep = nre - (int)vw + 1;
  }
  s.ref_start = vw + (uint32_t)start;
  s.ref_end = vw + (uint32_t)mep - 1;
  return s;
}

// GetMAPQForSingleEndRead with split_alignment (mapping_generator.h:920-1022)
CM_HD uint8_t cm_mapq_single_split(const CmDev &d, int num_errors, uint16_t alignment_length, int read_length, int max_diff,
                                   int second_err, int n_best, int n_second, uint32_t rep_len, uint32_t strand_ncand) {
  const int e = d.p.e;
  double alignment_identity = (double)(-num_errors) / alignment_length;
  if (alignment_identity > 1) alignment_identity = 1;
  int mapq = 0;
  int second = second_err;
  if (n_best > 1) {
  } else {
    if (second > num_errors + max_diff) second = num_errors + max_diff;
    double tmp = d.mq.len_coef[alignment_length];
    tmp *= alignment_identity * alignment_identity;
    mapq = (int)(5 * 6.02 * (second - num_errors) * tmp * tmp + 0.499);
  }
  if (n_second > 0) mapq -= cm_nsec_penalty(d.mq, n_second);
  if (mapq > 60) mapq = 60;
  if (mapq < 0) mapq = 0;
  if (rep_len > 0) {
    double frac_rep = rep_len / (double)read_length;
    if (rep_len >= (uint32_t)read_length) frac_rep = 0.999;
    if (alignment_identity <= 0.95) mapq = (int)(mapq * (1 - CM_SQRT(frac_rep)) + 0.499);
    else if (alignment_identity <= 0.97) mapq = (int)(mapq * (1 - frac_rep) + 0.499);
    else if (alignment_identity >= 0.999) mapq = (int)(mapq * (1 - frac_rep * frac_rep * frac_rep * frac_rep) + 0.499);
    else mapq = (int)(mapq * (1 - frac_rep * frac_rep) + 0.499);
  }
  if ((int)alignment_length < read_length - e && second != num_errors) {
    if (rep_len >= alignment_length && rep_len < (uint32_t)read_length && (int)alignment_length < read_length / 3) mapq = 0;
    const int diff = second - num_errors;
    if (second - num_errors <= e * 3 / 4 && strand_ncand >= 5) mapq = (int)((uint32_t)mapq - (strand_ncand / 5 / (uint32_t)diff));
    if (mapq < 0) mapq = 0;
    if (n_second > 0 && second - num_errors <= e * 3 / 4) mapq /= (n_second / diff + 1);
  }
  return (uint8_t)mapq;
}

// orientation o of a split pairing: 0 (+,-), 1 (-,+), 2 (+,+), 3 (-,-)  (mapping_generator.h:176-191)
CM_HD int cm_split_s1(int o) { return o & 1; }
CM_HD int cm_split_s2(int o) { return o == 0 || o == 3 ? 1 : 0; }

// the k-th draft mapping (0-based) of read r on `strand` whose error value equals `want_err`
CM_HD uint32_t cm_kth_best_draft(const CmDev &d, uint32_t r, int strand, int want_err, uint32_t k) {
  const int16_t *de = cm_d_err(d, r, strand);
  const uint32_t n = strand ? d.ndn[r] : d.ndp[r];
  for (uint32_t i = 0; i < n; ++i)
    if ((int)de[i] == want_err) { if (k == 0) return i; --k; }
  return 0;
}
CM_HD uint32_t cm_count_best_draft(const CmDev &d, uint32_t r, int strand, int want_err) {
  const int16_t *de = cm_d_err(d, r, strand);
  const uint32_t n = strand ? d.ndn[r] : d.ndp[r];
  uint32_t c = 0;
  for (uint32_t i = 0; i < n; ++i) c += (int)de[i] == want_err;
  return c;
}

// ProcessBestMappings... + EmplaceBackPairedEndMappingRecord<PairsMapping> for the pairing
// (orientation pe.f_dir, draft indices pe.f_i1 / pe.f_i2) (mapping_generator.h:487-653,
// mapping_generator.cc:169-210 with the default identity rid ranks, chromap.cc:867-877).
// Record layout = cmgpu_pairs_record (24 bytes).
template <bool SAM = false>
CM_HD void cm_emit_pairs_record(const CmDev &d, uint32_t pair, const CmPe &pe, uint32_t nth = 0) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  const int o = (int)pe.f_dir, s1 = cm_split_s1(o), s2 = cm_split_s2(o);
  const uint32_t len1 = d.rlen[r1], len2 = d.rlen[r2];
  const uint64_t dp1 = cm_d_pos(d, r1, s1)[pe.f_i1], dp2 = cm_d_pos(d, r2, s2)[pe.f_i2];
  const int e1 = cm_d_err(d, r1, s1)[pe.f_i1], e2 = cm_d_err(d, r2, s2)[pe.f_i2];
  const uint32_t w1 = (d.dsplit + d.m_off[r1] + (s1 ? d.ncp[r1] + d.resc_p[r1] : 0))[pe.f_i1];
  const uint32_t w2 = (d.dsplit + d.m_off[r2] + (s2 ? d.ncp[r2] + d.resc_p[r2] : 0))[pe.f_i2];
  CmSamAln sa1, sa2;
  CmSpan a, b;
  if constexpr (SAM) {
    a = cm_ref_start_end_split_sam(d, pair, r1, dp1, w1, s1, cm_read_ptr(d, r1), (int)len1, &sa1);
    b = cm_ref_start_end_split_sam(d, pair, r2, dp2, w2, s2, cm_read_ptr(d, r2), (int)len2, &sa2);
  } else {
    a = cm_ref_start_end_split(d, dp1, w1, s1, cm_read_ptr(d, r1), (int)len1);
    b = cm_ref_start_end_split(d, dp2, w2, s2, cm_read_ptr(d, r2), (int)len2);
  }
  const uint16_t al1 = (uint16_t)(a.ref_end - a.ref_start + 1), al2 = (uint16_t)(b.ref_end - b.ref_start + 1);
  uint8_t mapq1 = cm_mapq_single_split(d, e1, al1, (int)len1, 2, d.second_err[r1], d.n_best[r1], d.n_second[r1], d.rep_len[r1],
                                       s1 == 0 ? d.fcp[r1] : d.fcn[r1]);
  uint8_t mapq2 = cm_mapq_single_split(d, e2, al2, (int)len2, 2, d.second_err[r2], d.n_best[r2], d.n_second[r2], d.rep_len[r2],
                                       s2 == 0 ? d.fcp[r2] : d.fcn[r2]);
  mapq1 = (uint8_t)(mapq1 * 1.2);
  if (mapq1 > 60) mapq1 = 60;
  mapq2 = (uint8_t)(mapq2 * 1.2);
  if (mapq2 > 60) mapq2 = 60;
  const uint8_t mapq = mapq1 < mapq2 ? mapq1 : mapq2;  // force_mapq is -1: no supplement in split mode
  const uint8_t is_unique = (pe.n_best == 1 || d.n_best[r1] == 1 || d.n_best[r2] == 1) ? 1 : 0;
  if constexpr (SAM) {  // flags mapping_generator.h:613-631, EmplaceBackPairedEndMappingRecord<SAMMapping> (mapping_generator.cc:84-108),
                        // PairedEndMappingInMemory::GetFragmentLength (mapping_in_memory.h:83-90) whatever the chromosomes
    const int tlen = s1 == 0 ? (int)(b.ref_end - a.ref_start + 1) : (int)(a.ref_end - b.ref_start + 1);
    const uint32_t f1 = 3u | (s1 ? 16u : 0u) | (s2 ? 32u : 0u) | 64u, f2 = 3u | (s2 ? 16u : 0u) | (s1 ? 32u : 0u) | 128u;
    cm_put_sam_record(d, r1, d.first_read_id + pair, a, b.ref_start, (int32_t)b.rid, s1 ? -tlen : tlen, f1, mapq, s1 ? 0 : 1, is_unique, sa1, len1);
    cm_put_sam_record(d, r2, d.first_read_id + pair, b, a.ref_start, (int32_t)a.rid, s2 ? -tlen : tlen, f2, mapq, s2 ? 0 : 1, is_unique, sa2, len2);
    const uint64_t slot = (uint64_t)pair * (uint32_t)d.p.max_best + nth;  // the pair counts as one record; its content is the two SAM slots
    uint32_t *o32 = reinterpret_cast<uint32_t *>(d.rec + slot * 24);
    o32[0] = d.first_read_id + pair; o32[1] = a.rid; o32[2] = b.rid; o32[3] = a.ref_start; o32[4] = b.ref_start; o32[5] = 0;
    d.rec_ok[slot] = 1;
    return;
  }
  cm_put_pairs_record(d, (uint64_t)pair * (uint32_t)d.p.max_best + nth, pair, a, b, s1, s2, mapq, is_unique);
}

// split-mode pairing (mapping_generator.h:389-415): every (best of read1, best of read2)
// combination in the four orientations; returns the pairing with enumeration index `want`
CM_HD void cm_split_pairing(const CmDev &d, uint32_t pair, int64_t want, CmPe &pe) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  const int m1 = d.min_err[r1], m2 = d.min_err[r2];
  uint32_t c1[2], c2[2];
  c1[0] = cm_count_best_draft(d, r1, 0, m1); c1[1] = cm_count_best_draft(d, r1, 1, m1);
  c2[0] = cm_count_best_draft(d, r2, 0, m2); c2[1] = cm_count_best_draft(d, r2, 1, m2);
  pe.min_sum = 2 * d.p.e + 1; pe.second_sum = 2 * d.p.e + 1; pe.n_best = 0; pe.n_second = 0;
  pe.f_dir = 0; pe.f_i1 = 0; pe.f_i2 = 0;
  bool located = false;
  int64_t seen = 0;
  for (int o = 0; o < 4; ++o) {
    const int s1 = cm_split_s1(o), s2 = cm_split_s2(o);
    const uint64_t prod = (uint64_t)c1[s1] * (uint64_t)c2[s2];
    if (prod == 0) continue;
    pe.min_sum = m1 + m2;
    if (!located && want >= seen && want < seen + (int64_t)prod) {
      const uint64_t k = (uint64_t)(want - seen);
      pe.f_dir = (uint32_t)o;
      pe.f_i1 = cm_kth_best_draft(d, r1, s1, m1, (uint32_t)(k / c2[s2]));
      pe.f_i2 = cm_kth_best_draft(d, r2, s2, m2, (uint32_t)(k % c2[s2]));
      located = true;
    }
    seen += (int64_t)prod;
  }
  pe.n_best = seen > 0x7fffffff ? 0x7fffffff : (int)seen;
}


// single-end: ProcessBestMappingsForSingleEndRead (mapping_generator.h:256-344) for the
// `choice`-th best mapping (draft mappings in emission order, + strand first), MAPQ with
// max_num_error_difference = error_threshold, EmplaceBackSingleEndMappingRecord
// <MappingWithoutBarcode> (mapping_generator.cc:7-16)
template <bool SAM>
CM_HD void cm_emit_single_record(const CmDev &d, uint32_t pair, uint32_t choice, uint32_t nth = 0) {
  const uint32_t r = 2 * pair;
  const uint64_t slot = (uint64_t)pair * (uint32_t)d.p.max_best + nth;
  const int me = d.min_err[r];
  uint32_t idx = 0;
  for (int strand = 0; strand < 2; ++strand) {
    const uint64_t *dp = cm_d_pos(d, r, strand);
    const int16_t *de = cm_d_err(d, r, strand);
    const uint32_t n = strand ? d.ndn[r] : d.ndp[r];
    for (uint32_t mi = 0; mi < n; ++mi) {
      if ((int)de[mi] > me) continue;
      if (idx == choice) {
        const uint32_t L = d.rlen[r];
        CmSamAln sa;
        CmSpan sp;
        if constexpr (SAM) sp = cm_ref_start_end_sam(d, pair, pair, dp[mi], strand, cm_read_ptr(d, r), (int)L, &sa);
        else sp = cm_ref_start_end(d, dp[mi], de[mi], strand, cm_read_ptr(d, r), (int)L);
        const uint16_t al = (uint16_t)(sp.ref_end - sp.ref_start + 1);
        const uint8_t mapq = cm_mapq_single(d, de[mi], al, (int)L, d.p.e, d.second_err[r], d.n_best[r], d.n_second[r], d.rep_len[r]);
        uint8_t *o = d.rec + slot * 24;
        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
        uint16_t *o16 = reinterpret_cast<uint16_t *>(o);
        o32[0] = d.first_read_id + pair;
        o32[1] = sp.rid;
        o32[2] = sp.ref_start;
        o16[6] = al;
        o[14] = mapq & 63;
        o[15] = strand == 0 ? 1 : 0;
        o[16] = d.n_best[r] == 1 ? 1 : 0;
        o[17] = 1;
        o16[9] = 0; o16[10] = 0; o16[11] = 0;
        d.rec_ok[slot] = 1;
        if constexpr (SAM)  // EmplaceBackSingleEndMappingRecord<SAMMapping> (mapping_generator.cc:43-57), flag mapping_generator.h:321-326
          cm_put_sam_record(d, pair, d.first_read_id + pair, sp, 0, -1, 0, strand == 0 ? 0u : 16u, mapq, strand == 0 ? 1 : 0,
                            d.n_best[r] == 1 ? 1 : 0, sa, L);
        return;
      }
      ++idx;
    }
  }
}

// ---------------------------------------------------------------------------------------
// S6a: per pair -- sort draft mappings by position, pairing sweeps in both orientations
//      (GenerateBestMappingsForPairedEndRead, mapping_generator.h:160-253), record for pairs
//      with a single best pairing.
// ---------------------------------------------------------------------------------------
// in two parts so that a group of lanes can run the sweeps of a pair with many draft mappings (cm_coop_s6a, cm_coop.h):
// cm_s6a_pre does everything else and says whether the paired-end sweeps have to run
template <bool SAM = false>
CM_HD bool cm_s6a_pre(const CmDev &d, uint32_t pair) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  for (uint32_t t = 0; t < (uint32_t)d.p.max_best; ++t) d.rec_ok[(uint64_t)pair * (uint32_t)d.p.max_best + t] = 0;
  d.pe_nbest[pair] = 0;
  if (!d.alive[pair]) return false;
  const uint32_t nd1 = d.ndp[r1] + d.ndn[r1], nd2 = d.ndp[r2] + d.ndn[r2];
  if (d.p.single) {  // GenerateBestMappingsForSingleEndRead (mapping_generator.h:115-157)
    if (nd1 == 0) return false;
    d.pe_nbest[pair] = d.n_best[r1];
    d.pe_min[pair] = d.min_err[r1]; d.pe_second[pair] = d.second_err[r1]; d.pe_nsecond[pair] = d.n_second[r1];
    if (d.n_best[r1] == 1) cm_emit_single_record<SAM>(d, pair, 0);
    return false;
  }
  if (!(nd1 > 0 && nd2 > 0)) return false;  // chromap.h:1092-1093
  if (d.p.split) {  // drafts stay in emission order (chromap.h:1099-1106)
    CmPe sp;
    cm_split_pairing(d, pair, 0, sp);
    d.pe_min[pair] = sp.min_sum; d.pe_second[pair] = sp.second_sum;
    d.pe_nbest[pair] = sp.n_best; d.pe_nsecond[pair] = sp.n_second;
    d.pe_first[pair] = sp.f_dir; d.pe_i1[pair] = sp.f_i1; d.pe_i2[pair] = sp.f_i2;
    if (sp.n_best == 1) cm_emit_pairs_record<SAM>(d, pair, sp);
    return false;
  }
  return true;
}
template <bool SAM = false>
CM_HD void cm_s6a_sweeps(const CmDev &d, uint32_t pair) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  for (uint32_t r = r1; r <= r2; ++r) {
    cm_sort_draft(const_cast<uint64_t *>(cm_d_pos(d, r, 0)), const_cast<int16_t *>(cm_d_err(d, r, 0)), d.ndp[r]);
    cm_sort_draft(const_cast<uint64_t *>(cm_d_pos(d, r, 1)), const_cast<int16_t *>(cm_d_err(d, r, 1)), d.ndn[r]);
  }
  CmPe pe;
  pe.min_sum = 2 * d.p.e + 1; pe.second_sum = 2 * d.p.e + 1; pe.n_best = 0; pe.n_second = 0;
  pe.f_dir = 0; pe.f_i1 = 0; pe.f_i2 = 0;
  int64_t seen = 0;
  const uint32_t len1 = d.rlen[r1], len2 = d.rlen[r2];
  cm_pair_dir(d, 0, cm_d_pos(d, r1, 0), cm_d_err(d, r1, 0), d.ndp[r1], cm_d_pos(d, r2, 1), cm_d_err(d, r2, 1), d.ndn[r2],
              len1, len2, pe, -1, 0, &seen);
  cm_pair_dir(d, 1, cm_d_pos(d, r1, 1), cm_d_err(d, r1, 1), d.ndn[r1], cm_d_pos(d, r2, 0), cm_d_err(d, r2, 0), d.ndp[r2],
              len1, len2, pe, -1, 0, &seen);
  d.pe_min[pair] = pe.min_sum; d.pe_second[pair] = pe.second_sum;
  d.pe_nbest[pair] = pe.n_best; d.pe_nsecond[pair] = pe.n_second;
  d.pe_first[pair] = pe.f_dir; d.pe_i1[pair] = pe.f_i1; d.pe_i2[pair] = pe.f_i2;
  if (pe.n_best == 1) cm_emit_record<SAM>(d, pair, pe);
}
template <bool SAM = false>
CM_HD void cm_s6a_pair(const CmDev &d, uint32_t pair) {
  if (cm_s6a_pre<SAM>(d, pair)) cm_s6a_sweeps<SAM>(d, pair);
}

// ---------------------------------------------------------------------------------------
// S6b: reservoir sampling of multi-mappers, one taskloop task ("chunk") per item.
//   The reference's generator is firstprivate in each task of `taskloop grainsize(5000)`
//   (chromap.h:863,892): every task starts from std::mt19937(11) and walks its own pairs
//   in order; libgomp gives T = n/grain tasks (1 if T <= 1) of n/T iterations, the first
//   n%T tasks one more.  Draws follow libstdc++'s uniform_int_distribution<int>(0,i)
//   (Lemire's method on 32-bit output).
// ---------------------------------------------------------------------------------------
struct CmMt { uint32_t mt[624]; int idx; };
CM_HD void cm_mt_seed(CmMt &g, uint32_t s) {
  g.mt[0] = s;
  for (int i = 1; i < 624; ++i) g.mt[i] = 1812433253u * (g.mt[i - 1] ^ (g.mt[i - 1] >> 30)) + (uint32_t)i;
  g.idx = 624;
}
CM_HD uint32_t cm_mt_next(CmMt &g) {
  if (g.idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g.mt[i] & 0x80000000u) | (g.mt[(i + 1) % 624] & 0x7fffffffu);
      g.mt[i] = g.mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    }
    g.idx = 0;
  }
  uint32_t y = g.mt[g.idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
CM_HD int cm_mt_uniform(CmMt &g, int hi) {
  const uint32_t range = (uint32_t)hi + 1u;
  uint64_t product = (uint64_t)cm_mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (uint32_t)(0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)cm_mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (int)(product >> 32);
}

// chunk geometry: pairs [lo,hi) of chunk c within the batch
CM_HD uint32_t cm_num_chunks(uint32_t n, uint32_t ref_batch, uint32_t grain) {
  uint32_t tot = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += ref_batch) {
    const uint32_t bn = n - b0 < ref_batch ? n - b0 : ref_batch;
    const uint32_t T = bn / grain;
    tot += T <= 1 ? 1 : T;
  }
  return tot;
}
CM_HD void cm_chunk_range(uint32_t n, uint32_t ref_batch, uint32_t grain, uint32_t c, uint32_t *lo, uint32_t *hi) {
  for (uint32_t b0 = 0; b0 < n; b0 += ref_batch) {
    const uint32_t bn = n - b0 < ref_batch ? n - b0 : ref_batch;
    uint32_t T = bn / grain;
    if (T <= 1) T = 1;
    if (c >= T) { c -= T; continue; }
    if (T == 1) { *lo = b0; *hi = b0 + bn; return; }
    const uint32_t dv = bn / T, md = bn % T;
    const uint32_t s = b0 + c * dv + (c < md ? c : md);
    *lo = s;
    *hi = s + dv + (c < md ? 1 : 0);
    return;
  }
  *lo = *hi = n;
}

// reservoir sampling of K = max_num_best_mappings of nb best mappings (mapping_generator.h:121-139,
// 199-214): slots start as 0..K-1, draws only when nb > K, the chosen indices sorted increasing
CM_HD void cm_reservoir(const CmDev &d, uint32_t pair, int nb, CmMt &g) {
  const int K = d.p.max_best;
  uint32_t *ch = d.pe_choice + (uint64_t)pair * (uint32_t)K;
  for (int i = 0; i < K; ++i) ch[i] = (uint32_t)i;
  if (nb <= K) return;
  for (int i = K; i < nb; ++i) {
    const int j = cm_mt_uniform(g, i);
    if (j < K) ch[j] = (uint32_t)i;
  }
  for (int a = 1; a < K; ++a) {
    const uint32_t v = ch[a];
    int b = a - 1;
    while (b >= 0 && ch[b] > v) { ch[b + 1] = ch[b]; --b; }
    ch[b + 1] = v;
  }
}

CM_HD void cm_s6b_sample(const CmDev &d, uint32_t chunk, CmMt &g) {
  uint32_t lo, hi;
  cm_chunk_range(d.n_pairs, (uint32_t)d.p.ref_batch, (uint32_t)d.p.grain, chunk, &lo, &hi);
  bool seeded = false;
  for (uint32_t pair = lo; pair < hi; ++pair) {
    const int nb = d.pe_nbest[pair];
    if (nb <= 1) continue;
    if (!d.p.single && nb > d.p.drop_rep) continue;  // mapping_generator.h:190-193: returns before drawing (paired-end only: GenerateBestMappingsForSingleEndRead, :115-157, has no such test)
    if (!seeded || d.p.single) { cm_mt_seed(g, 11); seeded = true; }  // single-end: a fresh generator per read (mapping_generator.h:128)
    cm_reservoir(d, pair, nb, g);
  }
}

// ---------------------------------------------------------------------------------------
// S6c: per pair -- record for multi-mappers (re-runs the sweeps to find the chosen pair)
// ---------------------------------------------------------------------------------------
template <bool SAM = false>
CM_HD void cm_s6c_multi(const CmDev &d, uint32_t pair) {
  const int nb = d.pe_nbest[pair];
  if (nb <= 1 || (!d.p.single && nb > d.p.drop_rep)) return;  // --drop-repetitive-reads applies to pairs only
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  CmPe pe;
  pe.min_sum = d.pe_min[pair]; pe.second_sum = d.pe_second[pair]; pe.n_best = nb; pe.n_second = d.pe_nsecond[pair];
  pe.f_dir = d.pe_first[pair]; pe.f_i1 = d.pe_i1[pair]; pe.f_i2 = d.pe_i2[pair];
  const uint32_t K = (uint32_t)d.p.max_best;
  const uint32_t to_report = (uint32_t)nb < K ? (uint32_t)nb : K;
  for (uint32_t t = 0; t < to_report; ++t) {  // the chosen indices are increasing, so are the records of a pair
    const int64_t want = (int64_t)d.pe_choice[(uint64_t)pair * K + t];
    if (d.p.single) { cm_emit_single_record<SAM>(d, pair, (uint32_t)want, t); continue; }
    if (d.p.split) {
      CmPe sp;
      cm_split_pairing(d, pair, want, sp);
      cm_emit_pairs_record<SAM>(d, pair, sp, t);
      continue;
    }
    if (want > 0) {
      int64_t seen = 0;
      const uint32_t len1 = d.rlen[r1], len2 = d.rlen[r2];
      bool found = cm_pair_dir(d, 0, cm_d_pos(d, r1, 0), cm_d_err(d, r1, 0), d.ndp[r1], cm_d_pos(d, r2, 1), cm_d_err(d, r2, 1),
                               d.ndn[r2], len1, len2, pe, want, pe.min_sum, &seen);
      if (!found)
        found = cm_pair_dir(d, 1, cm_d_pos(d, r1, 1), cm_d_err(d, r1, 1), d.ndn[r1], cm_d_pos(d, r2, 0), cm_d_err(d, r2, 0),
                            d.ndp[r2], len1, len2, pe, want, pe.min_sum, &seen);
      if (!found) { d.stats[CM_ST_ERR] = 2; return; }
    } else {
      pe.f_dir = d.pe_first[pair]; pe.f_i1 = d.pe_i1[pair]; pe.f_i2 = d.pe_i2[pair];
    }
    cm_emit_record<SAM>(d, pair, pe, t);
  }
}

#endif  // CM_STAGES_H_
