// cm_coop.h -- stage functions in which a GROUP of lanes works on ONE read (or pair): the long hit, candidate and
// draft-mapping lists of reads from repeats.  cm_stages.h holds the one-lane-per-item definitions these must equal,
// element for element; here the sequential loops of the reference are re-expressed as sorts, scans and searches whose
// results are provably the same (each function says how), so that the hundreds to thousands of list entries of such a
// read are spread over 16 / 64 / 256 lanes.
//
// The functions are templates over a group type GT that supplies
//   static constexpr int G, W      lanes in the group; lanes of a "wave part" (min(G, 64)) -- rank() works inside one
//   uint32_t t                     this lane's index in the group
//   void sync()                    barrier of the group; shared and global writes before it are visible after it
//   uint32_t rank(bool p, uint32_t *total)     lanes of my wave part with p and a lower index; *total = all with p
//   uint32_t scan(uint32_t v, uint32_t *total) exclusive prefix sum over the group (contains barriers)
//   uint64_t max64(uint64_t v)     maximum over the group (contains barriers)
// Every lane of the group calls these the same number of times (uniform control flow around them).
// cm_kernels.hip instantiates them with the device group (wave shuffles, ballots, LDS); tests/hostemu runs the same text
// with one OS thread per lane -- test infrastructure, like the rest of hostemu.
#ifndef CM_COOP_H_
#define CM_COOP_H_

#include "cm_stages.h"

CM_HD uint32_t cm_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// ---------------------------------------------------------------------------------------
// Merge sort of nr contiguous ascending runs: src[0..tot) = runs [rb[i], rb[i+1]) (rb[0] = 0, rb[nr] = tot).  Bottom-up
// pairwise merges with merge-path partitioning: every lane produces ceil(tot / G) consecutive outputs of a level from
// one binary search and a sequential two-way merge.  ceil(log2 nr) levels against the (log2 tot)^2 / 2 compare-exchange
// stages of a bitonic network: a hit list is the union of one sorted occurrence run per minimizer and strand.
// The buffers ping-pong; returns the one that holds the sorted list.  rb, rb2: nr + 1 entries each.
// ---------------------------------------------------------------------------------------
template <class GT>
CM_HD uint64_t *cm_coop_merge_runs(GT &g, uint64_t *src, uint64_t *dst, uint32_t *rb, uint32_t *rb2, uint32_t nr, uint32_t tot) {
  const uint32_t VT = (tot + (uint32_t)GT::G - 1) / (uint32_t)GT::G;
  const uint32_t c0 = cm_min_u32(tot, g.t * VT), c1 = cm_min_u32(tot, c0 + VT);
  while (nr > 1) {
    const uint32_t nr2 = (nr + 1) >> 1;
    uint32_t p = c0, j = 0;
    if (p < c1) {  // pair of p: the largest j with rb[2j] <= p
      uint32_t lo = 0, hi = nr2;
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rb[2 * mid] <= p) lo = mid; else hi = mid;
      }
      j = lo;
    }
    while (p < c1) {
      const uint32_t a0 = rb[2 * j], a1 = rb[cm_min_u32(2 * j + 1, nr)], b1 = rb[cm_min_u32(2 * j + 2, nr)];
      if (p >= b1) { ++j; continue; }  // empty pair
      const uint32_t la = a1 - a0, lb = b1 - a1, diag = p - a0;
      // merge path: how many of the first `diag` outputs come from run a (ties take a first)
      uint32_t lo = diag > lb ? diag - lb : 0, hi = diag < la ? diag : la;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (src[a0 + mid] <= src[a1 + (diag - 1 - mid)]) lo = mid + 1; else hi = mid;
      }
      uint32_t ia = a0 + lo, ib = a1 + (diag - lo);
      const uint32_t pend = c1 < b1 ? c1 : b1;
      uint64_t va = ia < a1 ? src[ia] : 0, vb = ib < b1 ? src[ib] : 0;
      for (; p < pend; ++p) {
        const bool take_a = ib >= b1 || (ia < a1 && va <= vb);
        if (take_a) { dst[p] = va; ++ia; va = ia < a1 ? src[ia] : 0; }
        else { dst[p] = vb; ++ib; vb = ib < b1 ? src[ib] : 0; }
      }
      ++j;
    }
    for (uint32_t q = g.t; q <= nr2; q += (uint32_t)GT::G) rb2[q] = q < nr2 ? rb[2 * q] : tot;
    g.sync();
    { uint64_t *x = src; src = dst; dst = x; }
    { uint32_t *x = rb; rb = rb2; rb2 = x; }
    nr = nr2;
  }
  return src;
}

// boundaries of the maximal ascending runs of a[0..tot): rb[0] = 0 < rb[1] < ... < rb[nr] = tot.  Returns nr, or 0 when
// there are more than cap runs (rb holds cap + 1 entries; nothing useful is written then).  a must be complete (synced).
template <class GT>
CM_HD uint32_t cm_coop_natural_runs(GT &g, const uint64_t *a, uint32_t tot, uint32_t *rb, uint32_t cap) {
  const uint32_t VT = (tot + (uint32_t)GT::G - 1) / (uint32_t)GT::G;
  const uint32_t c0 = cm_min_u32(tot, g.t * VT), c1 = cm_min_u32(tot, c0 + VT);
  uint32_t cnt = 0;
  for (uint32_t i = c0; i < c1; ++i) cnt += (i == 0 || a[i] < a[i - 1]) ? 1u : 0u;
  uint32_t nr;
  uint32_t at = g.scan(cnt, &nr);
  if (nr > cap) return 0;
  for (uint32_t i = c0; i < c1; ++i)
    if (i == 0 || a[i] < a[i - 1]) rb[at++] = i;
  if (g.t == 0) rb[nr] = tot;
  g.sync();
  return nr;
}

// ---------------------------------------------------------------------------------------
// The cluster sweep of a sorted hit list by the group (CandidateProcessor::GenerateCandidatesOnOneStrand,
// candidate_processor.cc:283-342, cut at its state-free breaks: cm_sweep_local_break / cm_sweep_cluster in cm_stages.h).
// S[0..tot) sorted, the + list S[0..np) followed by the - list (keys with bit 63, cleared on output).  Every lane sweeps
// the local clusters whose first hit it owns; oc[i] (tot entries of shared memory) first holds the candidate count of the
// cluster starting at i, then its exclusive prefix.  Candidates of the + list go to out_p / out_pc, of the - list to
// out_n / out_nc, in list order.  *ncp_out, *ncn_out: their numbers (every lane gets them).
// ---------------------------------------------------------------------------------------
template <class GT>
CM_HD void cm_coop_sweep(GT &g, const uint64_t *S, uint32_t tot, uint32_t np, int e, int req, uint32_t num_minimizers, uint16_t *oc,
                         uint64_t *out_p, uint8_t *out_pc, uint64_t *out_n, uint8_t *out_nc, uint32_t *ncp_out, uint32_t *ncn_out) {
  const uint64_t SB = 1ull << 63;
  for (uint32_t i = g.t; i < tot; i += (uint32_t)GT::G) {
    uint32_t c = 0;
    if (i == 0 || cm_sweep_local_break(S[i - 1], S[i], e)) {
      uint32_t end = i + 1;
      while (end < tot && !cm_sweep_local_break(S[end - 1], S[end], e)) ++end;
      c = cm_sweep_cluster(S, 1, i, end, e, req, num_minimizers, nullptr, nullptr);
    }
    oc[i] = (uint16_t)c;
  }
  g.sync();
  // exclusive scan of oc in list order: per-lane chunk sums, group scan, rewrite
  const uint32_t VT = (tot + (uint32_t)GT::G - 1) / (uint32_t)GT::G;
  const uint32_t c0 = cm_min_u32(tot, g.t * VT), c1 = cm_min_u32(tot, c0 + VT);
  uint32_t sum = 0;
  for (uint32_t i = c0; i < c1; ++i) sum += oc[i];
  uint32_t total;
  uint32_t run = g.scan(sum, &total);
  for (uint32_t i = c0; i < c1; ++i) { const uint32_t x = oc[i]; oc[i] = (uint16_t)run; run += x; }
  g.sync();
  const uint32_t ncp = np < tot ? (np > 0 ? (uint32_t)oc[np] : 0u) : total, ncn = total - ncp;
  for (uint32_t i = g.t; i < tot; i += (uint32_t)GT::G) {
    if (i == 0 || cm_sweep_local_break(S[i - 1], S[i], e)) {
      uint32_t end = i + 1;
      while (end < tot && !cm_sweep_local_break(S[end - 1], S[end], e)) ++end;
      const uint32_t off = oc[i];
      if (i < np) cm_sweep_cluster(S, 1, i, end, e, req, num_minimizers, out_p + off, out_pc + off, ~SB);
      else cm_sweep_cluster(S, 1, i, end, e, req, num_minimizers, out_n + (off - ncp), out_nc + (off - ncp), ~SB);
    }
  }
  *ncp_out = ncp;
  *ncn_out = ncn;
}

// shared-memory work area of one group for the hit-list stages (sizes in entries)
struct CmCoopMem {
  uint64_t *A, *B;      // P each
  uint16_t *oc;         // P
  uint8_t *cc;          // P
  uint32_t *rb, *rb2;   // RB + 1 each: run boundaries
  uint32_t *moff;       // MM + 1: start of an included minimizer's segment
  uint32_t *mpc;        // MM: its + hits
  uint32_t *mmi;        // MM: its index in the read's minimizer list
  uint32_t *mps, *mns;  // MM each: start of its + / - sub-list in the compacted list
  uint32_t P, MM, RB;
};
CM_HD size_t cm_coop_mem_bytes(uint32_t P, uint32_t MM, uint32_t RB) {
  return (size_t)P * 19 + ((size_t)2 * (RB + 1) + (size_t)MM * 5 + 1) * 4 + 32;
}
// carve a group's area out of `base` (16-byte aligned, cm_coop_mem_bytes(P, MM, RB) bytes)
CM_HD CmCoopMem cm_coop_mem_at(uint8_t *base, uint32_t P, uint32_t MM, uint32_t RB) {
  CmCoopMem m;
  m.P = P; m.MM = MM; m.RB = RB;
  m.A = reinterpret_cast<uint64_t *>(base);
  m.B = m.A + P;
  m.rb = reinterpret_cast<uint32_t *>(m.B + P);
  m.rb2 = m.rb + RB + 1;
  m.moff = m.rb2 + RB + 1;
  m.mpc = m.moff + MM + 1;
  m.mmi = m.mpc + MM;
  m.mps = m.mmi + MM;
  m.mns = m.mps + MM;
  m.oc = reinterpret_cast<uint16_t *>(m.mns + MM);
  m.cc = reinterpret_cast<uint8_t *>(m.oc + P);
  return m;
}

// ---------------------------------------------------------------------------------------
// S3b for one read with a long hit list (cm_s3b_core's results, element for element):
//   expand   one wave part per minimizer: its occurrence run goes to the run's segment of B, + hits ascending from the
//            front, - hits (bit 63 set) from the back -- the run is sorted by (sequence, position) and a read position
//            is added or subtracted, so either sub-list is ascending unless a diagonal wraps below zero;
//   compact  B -> A: the + sub-lists of all minimizers, then the - sub-lists read backwards (ascending);
//   sort     boundaries of the ascending runs actually present (a wrapped diagonal just starts another run), merge sort;
//   sweep    cm_coop_sweep, candidates straight to the read's global segment (+ at h[0..), - at h[np..)).
// Returns false -- nothing written -- when the read has more included minimizers than m.MM or more runs than m.RB
// (the caller hands it to the bitonic-sort kernel); every lane returns the same value.
// ---------------------------------------------------------------------------------------
template <class GT>
CM_HD bool cm_coop_s3b(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m) {
  const uint32_t tot = d.hit_tot[r];
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  const uint32_t maxf = d.round2[r] ? (uint32_t)d.p.f1 : (uint32_t)d.p.f0;
  const uint64_t SB = 1ull << 63;
  if (tot > m.P) return false;
  // ---- included minimizers and their segments
  uint32_t R = 0, off = 0;
  for (uint32_t base = 0; base < n; base += (uint32_t)GT::G) {
    const uint32_t mi = base + g.t;
    uint32_t len = 0;
    if (mi < n) {
      const uint8_t kind = d.pr_kind[b + mi];
      if (kind == CM_PR_SINGLE) len = 1;
      else if (kind == CM_PR_MULTI) { const uint32_t nocc = (uint32_t)d.pr_val[b + mi]; if (nocc < maxf) len = nocc; }
    }
    uint32_t tv;
    const uint32_t sv = g.scan((len ? 1u << 20 : 0u) | len, &tv);  // tot <= 8192 < 2^20: both sums in one scan
    const uint32_t ri = R + (sv >> 20), o = off + (sv & 0xfffffu);
    if (len && ri < m.MM) { m.mmi[ri] = mi; m.moff[ri] = o; }
    R += tv >> 20;
    off += tv & 0xfffffu;
  }
  if (R > m.MM || off != tot) return false;
  if (g.t == 0) m.moff[R] = tot;
  g.sync();
  // ---- expand
  for (uint32_t ri = g.t / (uint32_t)GT::W; ri < R; ri += (uint32_t)(GT::G / GT::W)) {
    const uint32_t mi = m.mmi[ri], seg = m.moff[ri], len = m.moff[ri + 1] - seg;
    const uint8_t kind = d.pr_kind[b + mi];
    const uint64_t val = d.pr_val[b + mi];
    const uint32_t ps = d.mm_ps[b + mi];
    const uint64_t *o = d.occ + (uint32_t)(val >> 32);
    uint32_t pc = 0, nc = 0;
    for (uint32_t base = 0; base < len; base += (uint32_t)GT::W) {
      const uint32_t oi = base + g.t % (uint32_t)GT::W;
      const bool v = oi < len;
      bool same = false;
      uint64_t cp = 0;
      if (v) cp = cm_cand_from_hit(kind == CM_PR_SINGLE ? val : o[oi], ps, d.p.k, &same);
      uint32_t tp, tn;
      const uint32_t rp = g.rank(v && same, &tp), rn = g.rank(v && !same, &tn);
      if (v) {
        if (same) m.B[seg + pc + rp] = cp;
        else m.B[seg + len - 1 - (nc + rn)] = cp | SB;
      }
      pc += tp;
      nc += tn;
    }
    if (g.t % (uint32_t)GT::W == 0) m.mpc[ri] = pc;
  }
  g.sync();
  // ---- compact: starts of every minimizer's + and - sub-list in A
  uint32_t np = 0;
  {
    uint32_t accp = 0, accn = 0;
    for (uint32_t base = 0; base < R; base += (uint32_t)GT::G) {
      const uint32_t ri = base + g.t;
      const uint32_t pc = ri < R ? m.mpc[ri] : 0u, len = ri < R ? m.moff[ri + 1] - m.moff[ri] : 0u;
      uint32_t tp, tn;
      const uint32_t sp = g.scan(pc, &tp), sn = g.scan(len - pc, &tn);
      if (ri < R) { m.mps[ri] = accp + sp; m.mns[ri] = accn + sn; }
      accp += tp;
      accn += tn;
    }
    np = accp;
  }
  g.sync();
  for (uint32_t x = g.t; x < tot; x += (uint32_t)GT::G) {
    uint32_t lo = 0, hi = R;  // the largest ri with moff[ri] <= x
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (m.moff[mid] <= x) lo = mid; else hi = mid;
    }
    const uint32_t local = x - m.moff[lo], pc = m.mpc[lo], len = m.moff[lo + 1] - m.moff[lo];
    if (local < pc) m.A[m.mps[lo] + local] = m.B[x];
    else m.A[np + m.mns[lo] + (len - 1 - local)] = m.B[x];
  }
  g.sync();
  // ---- sort
  const uint32_t nr = cm_coop_natural_runs(g, m.A, tot, m.rb, m.RB);
  if (nr == 0) return false;
  const uint64_t *S = cm_coop_merge_runs(g, m.A, m.B, m.rb, m.rb2, nr, tot);
  // ---- sweep
  const uint32_t nn = tot - np;
  const bool use_high = d.round2[r] && np > 0 && nn > 0;
  int req = (int)n - (int)d.rep_cnt[r];
  req = req > 1 ? req : 1;
  req = req > d.p.min_seeds ? d.p.min_seeds : req;
  if (use_high) req = d.p.min_seeds;
  uint64_t *h = d.hbuf + d.hit_off[r];
  uint8_t *hc = d.hcnt + d.hit_off[r];
  uint32_t ncp, ncn;
  cm_coop_sweep(g, S, tot, np, d.p.e, req, n, m.oc, h, hc, h + np, hc + np, &ncp, &ncn);
  if (g.t == 0) { d.n_pos_hit[r] = np; d.ncp[r] = ncp; d.ncn[r] = ncn; }
  return true;
}

// ---------------------------------------------------------------------------------------
// S4b for one read whose rescue hits are many (cm_s4b_rescue_merge's results, element for element).  The hits of one
// direction are already in place (out + n1 .., written minimizer by minimizer: ascending runs); the group
//   sorts   them (natural runs + merge sort) and sweeps them with seeds_required = 1 into the augmented list X;
//   merges  X with the read's own candidates c0 (MergeCandidates, candidate_processor.cc:345-414): the two lists are
//           strictly ascending, so the sequential loop emits the merged sequence Z (equal positions once, with the larger
//           count), keeping an entry iff it lies more than e beyond the last KEPT entry.  An entry more than e beyond its
//           predecessor in Z is kept whatever came before ("sure start"); between sure starts the greedy rule runs
//           sequentially -- every lane replays the (short) chain that reaches into its chunk of Z.  Z is laid out in the
//           read's segment of the filtered-candidate arrays (zp / zc: same capacity, written only by S4c later).
// Returns the merged list's length (every lane).  active == false: no augmentation in this direction, c0 is copied.
// ---------------------------------------------------------------------------------------
template <class GT>
CM_HD uint32_t cm_coop_bcast0(GT &g, uint32_t v) { return (uint32_t)g.max64(g.t == 0 ? (uint64_t)v : 0ull); }

template <class GT>
CM_HD void cm_coop_copy_list(GT &g, const uint64_t *sp, const uint8_t *sc, uint64_t *dp, uint8_t *dc, uint32_t n) {
  for (uint32_t i = g.t; i < n; i += (uint32_t)GT::G) { dp[i] = sp[i]; dc[i] = sc[i]; }
}

// the greedy rule over the lane's chunk [z0, z1) of Z: counts the kept entries, or (dp != nullptr) writes them from off on
CM_HD uint32_t cm_coop_accept_walk(const uint64_t *zp, const uint8_t *zc, uint32_t nz, uint32_t z0, uint32_t z1, uint64_t E, uint64_t *dp,
                                   uint8_t *dc, uint32_t off) {
  if (z0 >= z1) return 0;
  uint64_t last = 0;
  if (z0 > 0 && !(zp[z0] > zp[z0 - 1] + E)) {  // the chunk starts inside a chain: replay it from its sure start
    uint32_t s = z0 - 1;
    while (s > 0 && !(zp[s] > zp[s - 1] + E)) --s;
    last = zp[s];
    for (uint32_t i = s + 1; i < z0; ++i)
      if (zp[i] > last + E) last = zp[i];
  }
  uint32_t cnt = 0;
  for (uint32_t i = z0; i < z1; ++i) {
    const uint64_t x = zp[i];
    const bool keep = i == 0 || x > zp[i - 1] + E || x > last + E;
    if (!keep) continue;
    last = x;
    if (dp) {
      uint8_t c = zc[i];
      if (i + 1 < nz && zp[i + 1] == x && zc[i + 1] > c) c = zc[i + 1];
      dp[off + cnt] = x;
      dc[off + cnt] = c;
    }
    ++cnt;
  }
  return cnt;
}

template <class GT>
CM_HD uint32_t cm_coop_rescue_dir(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m, uint64_t *out, uint8_t *outc, uint32_t n1, uint32_t cnt,
                                  bool active, const uint64_t *c0p, const uint8_t *c0c, uint64_t *zp, uint8_t *zc) {
  const int e = d.p.e;
  if (!active || cnt == 0) {
    cm_coop_copy_list(g, c0p, c0c, out, outc, n1);
    return n1;
  }
  uint32_t nr = 0;
  if (cnt <= m.P) {
    for (uint32_t i = g.t; i < cnt; i += (uint32_t)GT::G) m.A[i] = out[n1 + i];
    g.sync();
    nr = cm_coop_natural_runs(g, m.A, cnt, m.rb, m.RB);
  }
  if (nr == 0) {  // more hits or runs than the work area holds: the one-lane definition
    uint32_t k = 0;
    if (g.t == 0) {
      cm_sort_u64(out + n1, cnt);
      const uint32_t naug = cm_sweep(out + n1, outc + n1, cnt, e, 1, d.mm_cnt[r]);
      if (naug > 0) k = cm_merge(c0p, c0c, n1, out, outc, naug, e);
      else { for (uint32_t i = 0; i < n1; ++i) { out[i] = c0p[i]; outc[i] = c0c[i]; } k = n1; }
    }
    return cm_coop_bcast0(g, k);
  }
  uint64_t *S = cm_coop_merge_runs(g, m.A, m.B, m.rb, m.rb2, nr, cnt);
  uint64_t *X = S == m.A ? m.B : m.A;
  uint32_t naug, none;
  cm_coop_sweep(g, S, cnt, cnt, e, 1, d.mm_cnt[r], m.oc, X, m.cc, X, m.cc, &naug, &none);
  g.sync();  // X / cc complete, S free
  if (naug == 0) {
    cm_coop_copy_list(g, c0p, c0c, out, outc, n1);
    return n1;
  }
  if (n1 == 0) {  // c1.swap(c2): the augmented list as it is (no distance rule)
    for (uint32_t i = g.t; i < naug; i += (uint32_t)GT::G) { out[i] = X[i]; outc[i] = m.cc[i]; }
    return naug;
  }
  // ---- Z = merge of c0 (staged in S when it fits) and X, ties: c0 first
  const uint64_t *a = c0p;
  if (n1 <= m.P) {
    for (uint32_t i = g.t; i < n1; i += (uint32_t)GT::G) S[i] = c0p[i];
    g.sync();
    a = S;
  }
  const uint32_t nz = n1 + naug;
  const uint32_t VT = (nz + (uint32_t)GT::G - 1) / (uint32_t)GT::G;
  const uint32_t z0 = cm_min_u32(nz, g.t * VT), z1 = cm_min_u32(nz, z0 + VT);
  if (z0 < z1) {
    uint32_t lo = z0 > naug ? z0 - naug : 0, hi = z0 < n1 ? z0 : n1;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a[mid] <= X[z0 - 1 - mid]) lo = mid + 1; else hi = mid;
    }
    uint32_t ia = lo, ib = z0 - lo;
    for (uint32_t z = z0; z < z1; ++z) {
      const bool take_a = ib >= naug || (ia < n1 && a[ia] <= X[ib]);
      if (take_a) { zp[z] = a[ia]; zc[z] = c0c[ia]; ++ia; }
      else { zp[z] = X[ib]; zc[z] = m.cc[ib]; ++ib; }
    }
  }
  g.sync();
  // ---- keep / drop, compaction
  const uint64_t E = (uint64_t)(int64_t)e;
  const uint32_t mine = cm_coop_accept_walk(zp, zc, nz, z0, z1, E, nullptr, nullptr, 0);
  uint32_t total;
  const uint32_t off = g.scan(mine, &total);
  (void)cm_coop_accept_walk(zp, zc, nz, z0, z1, E, out, outc, off);
  return total;
}

template <class GT>
CM_HD void cm_coop_rescue_merge(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m) {
  const uint32_t o = r ^ 1u;
  const uint32_t ncp = d.ncp[r], ncn = d.ncn[r], rp = d.resc_p[r], rn = d.resc_n[r];
  uint64_t *P = d.mbuf + d.m_off[r];
  uint8_t *PC = d.mcnt + d.m_off[r];
  uint64_t *N = P + ncp + rp;
  uint8_t *NC = PC + ncp + rp;
  uint64_t *ZP = d.fbuf + d.m_off[r];
  uint8_t *ZC = d.fcnt + d.m_off[r];
  const bool aug = d.aug[r] != 0;
  const bool do_n = aug && d.ncp[o] > 0 && d.res_neg[r] >= 0 && rn > 0;
  const bool do_p = aug && d.ncn[o] > 0 && d.res_pos[r] >= 0 && rp > 0;
  const uint32_t mcn = cm_coop_rescue_dir(d, r, g, m, N, NC, ncn, rn, do_n, cm_c0_neg(d, r), cm_c0_ncnt(d, r), ZP + ncp + rp, ZC + ncp + rp);
  g.sync();  // the work area is reused
  const uint32_t mcp = cm_coop_rescue_dir(d, r, g, m, P, PC, ncp, rp, do_p, cm_c0_pos(d, r), cm_c0_pcnt(d, r), ZP, ZC);
  if (g.t == 0) { d.mcp[r] = mcp; d.mcn[r] = mcn; }
}

#endif
