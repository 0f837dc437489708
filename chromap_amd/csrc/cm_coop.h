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
//   uint32_t scanmax(uint32_t v, uint32_t *total)  exclusive prefix maximum (0 in front) and the overall maximum
//   uint64_t max64(uint64_t v), min64  maximum / minimum over the group (contain barriers)
// Every lane of the group calls these the same number of times (uniform control flow around them).
// cm_kernels.hip instantiates them with the device group (wave shuffles, ballots, LDS); tests/hostemu runs the same text
// with one OS thread per lane -- test infrastructure, like the rest of hostemu.
#ifndef CM_COOP_H_
#define CM_COOP_H_

#include "cm_stages.h"

CM_HD uint32_t cm_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
// *p += v, the old value returned: atomic on the device (the emulated lanes of tests/hostemu take turns on one thread)
CM_HD unsigned long long cm_fetch_add64(unsigned long long *p, unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(p, v);
#else
  const unsigned long long old = *p;
  *p = old + v;
  return old;
#endif
}
// measurement aid: lane 0 of a group adds the shader-clock cycles since its last mark to d.prof[k] (nullptr: nothing)
#if defined(__HIP_DEVICE_COMPILE__)
// (one block in 32 is timed: an atomic per mark and group from every block of a launch -- millions on one address, which retire at
// ~90 per microsecond -- made the global-memory phases read 5-10 x too long in round 5's tables)
#define CM_PROF_ON(p) ((p) && (blockIdx.x & 31u) == 0u)
#define CM_PROF_BEGIN(d) long long cm_prof_t_ = CM_PROF_ON((d).prof) ? clock64() : 0
#define CM_PROF_MARK(d, g, k) do { if (CM_PROF_ON((d).prof) && (g).t == 0) { const long long now_ = clock64(); atomicAdd(&(d).prof[k], (unsigned long long)(now_ - cm_prof_t_)); cm_prof_t_ = clock64(); } } while (0)
#define CM_PROF_RESET(d) do { if (CM_PROF_ON((d).prof)) cm_prof_t_ = clock64(); } while (0)
#define CM_PROF_COUNT(d, g, k, v) do { if (CM_PROF_ON((d).prof) && (g).t == 0) atomicAdd(&(d).prof[k], (unsigned long long)(v)); } while (0)
#define CM_PROF_PTR_COUNT(p, g, k, v) do { if ((p) && (g).t == 0) atomicAdd(&(p)[k], (unsigned long long)(v)); } while (0)
#define CM_PROF_PTR_BEGIN(p) long long cm_prof_u_ = CM_PROF_ON(p) ? clock64() : 0
#define CM_PROF_PTR_MARK(p, g, k) do { if (CM_PROF_ON(p) && (g).t == 0) { const long long now_ = clock64(); atomicAdd(&(p)[k], (unsigned long long)(now_ - cm_prof_u_)); cm_prof_u_ = clock64(); } } while (0)
#else
#define CM_PROF_PTR_COUNT(p, g, k, v) do { } while (0)
#define CM_PROF_PTR_BEGIN(p) do { } while (0)
#define CM_PROF_PTR_MARK(p, g, k) do { } while (0)
#define CM_PROF_BEGIN(d) do { } while (0)
#define CM_PROF_MARK(d, g, k) do { } while (0)
#define CM_PROF_RESET(d) do { } while (0)
#define CM_PROF_COUNT(d, g, k, v) do { } while (0)
#endif
// entries per lane when a list of n is cut into one contiguous chunk per lane.  (Making it odd, so that sixteen consecutive lanes
// start in sixteen different 8-byte bank pairs of shared memory, was measured: every cooperative kernel got 15-20 % SLOWER --
// k_s3b_coop<512> 8.6 -> 10.1 ms -- the idle lanes of the longer chunks cost more than the conflicts.)
CM_HD uint32_t cm_coop_chunk(uint32_t n, uint32_t G) { return (n + G - 1) / G; }

// ---------------------------------------------------------------------------------------
// Merge sort of nr contiguous ascending runs: src[0..tot) = runs [rb[i], rb[i+1]) (rb[0] = 0, rb[nr] = tot).  Bottom-up
// pairwise merges with merge-path partitioning: every lane produces ceil(tot / G) consecutive outputs of a level from
// one binary search and a sequential two-way merge.  ceil(log2 nr) levels against the (log2 tot)^2 / 2 compare-exchange
// stages of a bitonic network: a hit list is the union of one sorted occurrence run per minimizer and strand.
// The buffers ping-pong; returns the one that holds the sorted list.  rb, rb2: nr + 1 entries each.
// ---------------------------------------------------------------------------------------
template <class GT, class K>
CM_HD K *cm_coop_merge_runs(GT &g, K *src, K *dst, uint32_t *rb, uint32_t *rb2, uint32_t nr, uint32_t tot, uint32_t levels = 32) {
  const K TOP = CmKeyOps<K>::top();
  const uint32_t VT = cm_coop_chunk(tot, (uint32_t)GT::G);
  const uint32_t c0 = cm_min_u32(tot, g.t * VT), c1 = cm_min_u32(tot, c0 + VT);
  // levels: stop after that many (the caller knows that the runs left then are what it wants: cm_coop_s3b keeps the two strands' lists apart)
  for (; nr > 1 && levels > 0; --levels) {
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
      // an exhausted run reads as the largest value (no key is: a hit's sequence number never has all bits set), so one comparison
      // decides; the next element of the run that gave is requested while the output is written
      K va = ia < a1 ? src[ia] : TOP, vb = ib < b1 ? src[ib] : TOP;
      for (; p < pend; ++p) {
        const bool take_a = va <= vb;
        dst[p] = take_a ? va : vb;
        const uint32_t nx = take_a ? ++ia : ++ib;
        const K nv = nx < (take_a ? a1 : b1) ? src[nx] : TOP;
        va = take_a ? nv : va;
        vb = take_a ? vb : nv;
      }
      ++j;
    }
    for (uint32_t q = g.t; q <= nr2; q += (uint32_t)GT::G) rb2[q] = q < nr2 ? rb[2 * q] : tot;
    g.sync();
    { K *x = src; src = dst; dst = x; }
    { uint32_t *x = rb; rb = rb2; rb2 = x; }
    nr = nr2;
  }
  return src;
}

// boundaries of the maximal ascending runs of a[0..tot): rb[0] = 0 < rb[1] < ... < rb[nr] = tot.  Returns nr, or 0 when
// there are more than cap runs (rb holds cap + 1 entries; nothing useful is written then).  a must be complete (synced).
template <class GT>
CM_HD uint32_t cm_coop_natural_runs(GT &g, const uint64_t *a, uint32_t tot, uint32_t *rb, uint32_t cap) {
  const uint32_t VT = cm_coop_chunk(tot, (uint32_t)GT::G);
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
// candidate_processor.cc:283-342, cut at its state-free breaks: cm_sweep_local_break / cm_sweep_cluster_from in cm_stages.h).
// S[0..tot) sorted, the + list S[0..np) followed by the - list (keys with bit 63, cleared on output).  ONE walk per local
// cluster: the lane that owns a cluster's first hit sweeps it and parks its candidates in the staging arrays at the cluster's
// own slots (xs / xc[i ..]: a cluster has at most as many candidates as hits), oc[i] = their number; after the scan of oc the
// parked candidates are copied to out_p / out_pc (+ list) and out_n / out_nc (- list) in list order.  oc, xs, xc: tot entries
// of the group's work memory, none of them S; the outputs may be S itself (it is dead after the walk).
// *ncp_out, *ncn_out: the candidates' numbers (every lane gets them).
// ---------------------------------------------------------------------------------------
// K: the keys' type (CmKeyOps); KO: the type of the outputs -- K (the candidates stay keys: cm_coop_rescue_dir) or uint64_t with 32-bit keys (the
// candidates leave as sequence << 32 | position: goff / n_seq are CmDev's)
template <class GT, class K, class KOUT>
CM_HD void cm_coop_sweep(GT &g, const K *S, uint32_t tot, uint32_t np, int e, int req, uint32_t num_minimizers, uint16_t *oc, K *xs,
                         uint8_t *xc, KOUT *out_p, uint8_t *out_pc, KOUT *out_n, uint8_t *out_nc, uint32_t *ncp_out, uint32_t *ncn_out,
                         unsigned long long *prof = nullptr, const uint32_t *goff = nullptr, uint32_t n_seq = 0) {
  typedef CmKeyOps<K> KO;
  CM_PROF_PTR_BEGIN(prof);
  // Every lane walks its own chunk of the list, front to back: the hits up to the chunk's first local break continue a cluster that an
  // earlier lane owns and are passed over; every cluster that STARTS in the chunk is swept to its end -- also beyond the chunk's.  So all
  // lanes take about VT steps (plus the tail of their last cluster) instead of one lane in five walking while the others of its wave wait
  // (round 4: a strided pass for the starts, then a walk per start -- 38 % of k_s3b_coop).  The two strands' lists never share a cluster:
  // a walk ends at its list's end (the - keys may or may not carry bit 63).
  const uint32_t VT = cm_coop_chunk(tot, (uint32_t)GT::G);
  const uint32_t c0 = cm_min_u32(tot, g.t * VT), c1 = cm_min_u32(tot, c0 + VT);
  const bool masked = VT <= 64;  // the chunk's cluster starts as a bit each; longer chunks (no work area has them today) mark them in oc
  if (!masked) {
    for (uint32_t i = c0; i < c1; ++i) oc[i] = 0;
    g.sync();  // (a walk writes oc at its start only, but a start may be any lane's: all zeros first)
  }
  unsigned long long starts = 0;
  uint32_t sum = 0, ncp_mine = 0;
  {
    uint32_t i = c0;
    if (i < c1 && i > 0 && i != np) {  // pass over the hits that belong to the cluster of the hit before the chunk
      K prev = S[i - 1];
      while (i < c1 && i != np) {
        const K x = S[i];
        if (KO::brk(prev, x, e)) break;
        prev = x;
        ++i;
      }
    }
    while (i < c1) {
      uint32_t end;
      const uint32_t c = cm_sweep_cluster_walk(S, i, i < np ? np : tot, e, req, num_minimizers, xs + i, xc + i, &end);
      if (c) {
        oc[i] = (uint16_t)c;
        if (masked) starts |= 1ull << (i - c0);
        sum += c;
        if (i < np) ncp_mine += c;
      }
      i = end;
    }
  }
  CM_PROF_PTR_MARK(prof, g, 11);
  // (both sums in one scan: a list has at most 65535 hits -- the offsets are 16-bit -- and no more candidates than hits.  The scan's
  // barriers also end the walks: S is no longer read afterwards, the outputs may overlay it)
  uint32_t both;
  uint32_t run = g.scan(sum << 16 | ncp_mine, &both) >> 16;
  const uint32_t total = both >> 16, ncp = both & 0xffffu;
  const uint32_t ncn = total - ncp;
  CM_PROF_PTR_MARK(prof, g, 13);
  // copy out: every lane the parked candidates of its own clusters (slot i + k -> prefix + k), in list order
  if (masked) {
    while (starts) {
      const uint32_t i = c0 + (uint32_t)__builtin_ctzll(starts);
      starts &= starts - 1;
      const uint32_t c = oc[i];
      for (uint32_t k = 0; k < c; ++k) {
        const KOUT x = CmKeyOut<K, KOUT>::get(xs[i + k], goff, n_seq);
        const uint8_t cc = xc[i + k];
        if (i < np) { out_p[run + k] = x; out_pc[run + k] = cc; }
        else { out_n[run + k - ncp] = x; out_nc[run + k - ncp] = cc; }
      }
      run += c;
    }
  } else {
    for (uint32_t i = c0; i < c1; ++i) {
      const uint32_t c = oc[i];
      for (uint32_t k = 0; k < c; ++k) {
        const KOUT x = CmKeyOut<K, KOUT>::get(xs[i + k], goff, n_seq);
        const uint8_t cc = xc[i + k];
        if (i < np) { out_p[run + k] = x; out_pc[run + k] = cc; }
        else { out_n[run + k - ncp] = x; out_nc[run + k - ncp] = cc; }
      }
      run += c;
    }
  }
  CM_PROF_PTR_MARK(prof, g, 14);
  *ncp_out = ncp;
  *ncn_out = ncn;
}

// shared-memory work area of one group for the hit-list stages (sizes in entries).  two_cc: S4b keeps its candidates' counts
// (cc) next to the parked ones of the sweep (cc2); S3b writes its candidates to global memory and needs one count array.
struct CmCoopMem {
  uint64_t *A, *B;      // P each (64-bit keys)
  uint32_t *A32, *B32;  // P each (32-bit keys, cm_coop_mem_at(..., key32): the two layouts are alternatives)
  uint16_t *oc;         // P: candidates per local cluster
  uint8_t *cc, *cc2;    // P each (cc2: two_cc only)
  uint32_t *rb, *rb2;   // RB + 1 each: run boundaries
  uint32_t *moff;       // MM + 1: start of an included minimizer's occurrences in the hit list
  uint64_t *mval;       // MM: index of its first occurrence in the occurrence table (a singleton: the occurrence itself)
  uint32_t *mps;        // MM: (read position << 1 | strand) of the minimizer, bit 31: singleton
  uint32_t P, MM, RB;
  // lists longer than P: the same algorithms on a slab of global memory that belongs to the group (slower per step, but
  // hundreds of lanes instead of one); gcap entries each, nullptr when the launch brought none
  uint64_t *gA, *gB;
  uint16_t *goc;
  uint8_t *gcc;
  uint32_t gcap;
};
CM_HD size_t cm_coop_slab_bytes(uint32_t gcap) { return (size_t)gcap * 19 + 64; }
CM_HD void cm_coop_slab_at(CmCoopMem &m, uint8_t *slab, uint32_t gcap) {
  m.gcap = slab ? gcap : 0;
  m.gA = reinterpret_cast<uint64_t *>(slab);
  m.gB = slab ? m.gA + gcap : nullptr;
  m.goc = slab ? reinterpret_cast<uint16_t *>(m.gB + gcap) : nullptr;
  m.gcc = slab ? reinterpret_cast<uint8_t *>(m.goc + gcap) : nullptr;
}
CM_HD size_t cm_coop_mem_bytes(uint32_t P, uint32_t MM, uint32_t RB, bool two_cc, bool key32 = false) {
  return (size_t)P * ((two_cc ? 20 : 19) - (key32 ? 8 : 0)) + ((size_t)2 * (RB + 1) + (size_t)MM * 4 + 2) * 4 + 32;
}
// carve a group's area out of `base` (16-byte aligned, cm_coop_mem_bytes(P, MM, RB, two_cc, key32) bytes; P even)
CM_HD CmCoopMem cm_coop_mem_at(uint8_t *base, uint32_t P, uint32_t MM, uint32_t RB, bool two_cc, bool key32 = false) {
  CmCoopMem m;
  m.P = P; m.MM = MM; m.RB = RB;
  m.A = key32 ? nullptr : reinterpret_cast<uint64_t *>(base);
  m.B = key32 ? nullptr : m.A + P;
  m.A32 = key32 ? reinterpret_cast<uint32_t *>(base) : nullptr;
  m.B32 = key32 ? m.A32 + P : nullptr;
  m.mval = reinterpret_cast<uint64_t *>(base + (size_t)P * (key32 ? 8 : 16));
  m.rb = reinterpret_cast<uint32_t *>(m.mval + MM);
  m.rb2 = m.rb + RB + 1;
  m.moff = m.rb2 + RB + 1;
  m.mps = m.moff + MM + 1;
  m.oc = reinterpret_cast<uint16_t *>(m.mps + MM + 1);
  m.cc = reinterpret_cast<uint8_t *>(m.oc + P);
  m.cc2 = two_cc ? m.cc + P : nullptr;
  m.gA = m.gB = nullptr; m.goc = nullptr; m.gcc = nullptr; m.gcap = 0;
  return m;
}

// ---------------------------------------------------------------------------------------
// S3b for one read with a long hit list (cm_s3b_core's results, element for element).  Two front ends: lists in the shared work area
// go through cm_coop_s3b_expand (below: pieces, ballots, run tables without a look at the keys -- and through cm_coop_s3b_k32 on 32-bit
// keys where the reference fits); lists on a slab of global memory (SLAB) keep the form of rounds 3-4:
//   expand   the hit list in minimizer order, occurrence order inside a minimizer: every lane takes hits x, x + G, ... --
//            which minimizer's occurrence it is comes from the table of run starts -- eight independent occurrence loads
//            in flight per lane; the - strand's keys get bit 63;
//   split    stable partition (one scan): the + hits in order, then the - hits in order.  An occurrence run is sorted by
//            (sequence, position) and a read position is added or subtracted, so each minimizer leaves one ascending run per
//            strand unless a diagonal wraps below zero;
//   sort     boundaries of the ascending runs actually present (a wrapped diagonal just starts another run), merge sort;
//   sweep    cm_coop_sweep, candidates straight to the read's global segment (+ at h[0..), - at h[np..)).
// Returns false -- nothing written -- when the read has more included minimizers than m.MM or more runs than m.RB
// (the caller hands it to the bitonic-sort kernel); every lane returns the same value.
// ---------------------------------------------------------------------------------------
// SLAB (compile time): the list buffers are the group's slab of global memory instead of the shared work area.  A template
// parameter, not a run-time choice: a pointer that may be either makes every access a FLAT instruction -- measured, the shared-
// memory form then runs at less than half its speed (its loads are chains of dependent accesses).
// ---- the shared-memory form's front end (round 5).  The hit list is never laid out lane by lane: the occurrence run of every included
// minimizer is cut into pieces of W consecutive occurrences (W: the lanes of a wave part), and a wave part takes a piece at a time --
//   pass 1   one coalesced load of the piece, the candidate position of every occurrence into B at its place in list order, bit 63
//            for the - strand; the piece's number of + hits (a ballot) into a table; a + hit whose diagonal starts before its sequence
//            does (the only way a run is not ascending) makes the group decline the read;
//   scan     of the pieces' + counts: where every piece's + hits and - hits begin in the two strands' lists;
//   pass 2   every piece again, from B: its + hits to the + list, its - hits to the - list, in order (ballot ranks).
// A piece never straddles two minimizers, so the lists' ascending runs are known without looking at the keys: run i of the + list
// starts where minimizer i's first piece does.  Both lists get one (possibly empty) run per minimizer, the + list's table is padded
// with empty runs to a power of two P2 -- then log2(P2) merge levels never pair a + run with a - run, and the level that only
// concatenated the two sorted lists (a quarter of the merge's work at 8 minimizers) is gone, like the passes that partitioned the
// list by strand and looked for run boundaries.
// Work memory: the pieces' tables overlay oc (2 * pieces + 2 entries of 16 bits, unused until the sweep).
// Returns 0 (declined), or the number of merge levels + 1; *np_out: the + list's length, m.rb: the run table (*nr_out runs).
// K = uint32_t: the keys are global coordinates (CmKeyOps; d.goff must be there); a hit's strand then goes to the count array cc
// (free until the sweep) instead of the key's top bit.
template <class K, class GT>
CM_HD uint32_t cm_coop_s3b_expand(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m, K *const A, K *const B, uint32_t tot, uint32_t *np_out, uint32_t *nr_out) {
  const bool K32 = sizeof(K) == 4;
  const uint32_t G = (uint32_t)GT::G, W = (uint32_t)GT::W, NW = G / W;
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  const uint32_t maxf = d.round2[r] ? (uint32_t)d.p.f1 : (uint32_t)d.p.f0;
  const uint64_t SB = 1ull << 63;
  const uint32_t wl = g.t % W, wv = g.t / W;
  uint32_t *const choff = m.rb2;  // first piece of every run (R + 1 entries; rb2 is free until the merge)
  CM_PROF_BEGIN(d);
  // ---- included minimizers: where their occurrences start in the list, and their pieces
  uint32_t R = 0, off = 0, NC = 0;
  for (uint32_t base = 0; base < n; base += G) {
    const uint32_t mi = base + g.t;
    uint32_t len = 0, ps = 0;
    uint64_t val = 0;
    if (mi < n) {
      const uint8_t kind = d.pr_kind[b + mi];
      if (kind != CM_PR_MISS) {
        val = d.pr_val[b + mi];
        ps = d.mm_ps[b + mi];
        if (kind == CM_PR_SINGLE) { len = 1; ps |= 1u << 31; }
        else { const uint32_t nocc = (uint32_t)val; if (nocc < maxf) len = nocc; val >>= 32; }
      }
    }
    uint32_t tv, tc;
    const uint32_t sv = g.scan((len ? 1u << 20 : 0u) | len, &tv);  // tot <= 8192 < 2^20: both sums in one scan
    const uint32_t sc = g.scan((len + W - 1) / W, &tc);
    const uint32_t ri = R + (sv >> 20), o = off + (sv & 0xfffffu);
    if (len && ri < m.MM) { m.moff[ri] = o; m.mval[ri] = val; m.mps[ri] = ps; choff[ri] = NC + sc; }
    R += tv >> 20;
    off += tv & 0xfffffu;
    NC += tc;
  }
  uint32_t P2 = 1, levels = 0;
  while (P2 < R) { P2 <<= 1; ++levels; }
  if (R == 0 || R > m.MM || off != tot || 2 * NC + 2 > m.P || P2 + R > m.RB) return 0;
  uint16_t *const chrun = m.oc, *const cplus = m.oc + NC;  // a piece's run; its + hits, then (scan) the + hits before it
  if (g.t == 0) { m.moff[R] = tot; choff[R] = NC; }
  g.sync();
  for (uint32_t ri = g.t; ri < R; ri += G)
    for (uint32_t c = choff[ri]; c < choff[ri + 1]; ++c) chrun[c] = (uint16_t)ri;
  g.sync();
  CM_PROF_MARK(d, g, 0);
  // ---- pass 1: four pieces per wave part and round, their loads in flight together
  uint32_t wrapped = 0;
  for (uint32_t cb = wv; cb < NC; cb += 4 * NW) {
    uint64_t hit[4];
    uint32_t ps[4], x[4];
    bool in[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = cb + (uint32_t)q * NW;
      in[q] = false; hit[q] = 0; ps[q] = 0; x[q] = 0;
      if (c < NC) {
        const uint32_t ri = chrun[c], o0 = m.moff[ri], j = (c - choff[ri]) * W + wl;
        in[q] = j < m.moff[ri + 1] - o0;
        if (in[q]) {
          ps[q] = m.mps[ri];
          const uint64_t v = m.mval[ri];
          x[q] = o0 + j;
          hit[q] = (ps[q] >> 31) ? v : d.occ[(uint32_t)v + j];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c = cb + (uint32_t)q * NW;
      bool same = false;
      const uint64_t cp = cm_cand_from_hit(hit[q], ps[q] & 0x7fffffffu, d.p.k, &same);
      if (in[q]) {
        if (K32) { B[x[q]] = (K)(d.goff[(uint32_t)(cp >> 32)] + (uint32_t)cp); m.cc[x[q]] = same ? 1 : 0; }
        else B[x[q]] = (K)(same ? cp : (cp | SB));
        if (same && (uint32_t)(hit[q] >> 1) < ((ps[q] & 0x7fffffffu) >> 1)) wrapped = 1;
      }
      uint32_t np_c;
      (void)g.rank(in[q] && same, &np_c);
      if (c < NC && wl == 0) cplus[c] = (uint16_t)np_c;
    }
  }
  g.sync();
  CM_PROF_MARK(d, g, 1);
  // ---- + hits before every piece
  uint32_t np = 0, any_wrapped = 0;
  for (uint32_t base = 0; base < NC || base == 0; base += G) {
    const uint32_t c = base + g.t;
    const uint32_t v = c < NC ? cplus[c] : 0u;
    uint32_t tv;
    const uint32_t sv = g.scan(v | (base == 0 ? wrapped << 20 : 0u), &tv);
    if (c < NC) cplus[c] = (uint16_t)(np + (sv & 0xfffffu));
    np += tv & 0xfffffu;
    any_wrapped |= tv >> 20;
  }
  if (any_wrapped) return 0;
  if (g.t == 0) cplus[NC] = (uint16_t)np;
  g.sync();
  CM_PROF_MARK(d, g, 3);
  // ---- pass 2: the two lists, and their run tables: + runs 0 .. R - 1, empty ones up to P2, then the - runs
  for (uint32_t c = wv; c < NC; c += NW) {
    const uint32_t ri = chrun[c], o0 = m.moff[ri], j = (c - choff[ri]) * W + wl;
    const bool in = j < m.moff[ri + 1] - o0;
    const uint32_t xq = o0 + j;
    const K v = in ? B[xq] : (K)0;
    const bool plus = in && (K32 ? m.cc[xq] != 0 : !((uint64_t)v >> 63));
    uint32_t unused;
    const uint32_t before = (uint32_t)cplus[c] + g.rank(plus, &unused);  // + hits in front of this one, in list order
    if (in) A[plus ? before : np + (xq - before)] = K32 ? v : (K)((uint64_t)v & ~SB);
  }
  for (uint32_t q = g.t; q <= P2 + R; q += G) {
    uint32_t v;
    if (q < R) v = cplus[choff[q]];
    else if (q < P2) v = np;
    else if (q < P2 + R) v = np + (m.moff[q - P2] - (uint32_t)cplus[choff[q - P2]]);
    else v = tot;
    m.rb[q] = v;
  }
  g.sync();
  CM_PROF_MARK(d, g, 2);
  *np_out = np;
  *nr_out = P2 + R;
  return levels + 1;
}

// cm_coop_s3b with 32-bit keys (the shared-memory form only): 11 instead of 19 bytes of shared memory per hit -- the kernel's time is
// the number of reads in flight (measured, round 5: 40 KB more per block, i.e. half the blocks per CU, and k_s3b_coop takes 57 % longer)
template <class GT>
CM_HD bool cm_coop_s3b_k32(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m) {
  const uint32_t tot = d.hit_tot[r];
  const uint32_t n = d.mm_cnt[r];
  // (round 6: what the sweep needs of the read is requested HERE, with the list's length -- behind the merge's barriers each of these was
  //  another trip to global memory on the group's critical path)
  const uint32_t r2 = d.round2[r], repc = d.rep_cnt[r], hoff = d.hit_off[r];
  if (tot > m.P || !d.goff || !m.A32) return false;
  CM_PROF_BEGIN(d);
  uint32_t np, nr;
  const uint32_t lv = cm_coop_s3b_expand<uint32_t>(d, r, g, m, m.A32, m.B32, tot, &np, &nr);
  if (lv == 0) return false;
  // (round 6) the coordinate table next to the keys: every candidate the sweep writes searches it to become sequence << 32 | position
  // again -- five dependent reads, from global memory until now.  m.mps is free once the hits are expanded; the merge's barriers
  // come between this copy and its use.
  const uint32_t *gl = d.goff;
  if (d.n_seq + 1 <= m.MM + 1) {
    for (uint32_t i = g.t; i <= d.n_seq; i += (uint32_t)GT::G) m.mps[i] = d.goff[i];
    gl = m.mps;
  }
  CM_PROF_RESET(d);
  uint32_t *S = cm_coop_merge_runs(g, m.A32, m.B32, m.rb, m.rb2, nr, tot, lv - 1);
  CM_PROF_MARK(d, g, 4);
  CM_PROF_COUNT(d, g, 8, 1); CM_PROF_COUNT(d, g, 9, nr); CM_PROF_COUNT(d, g, 10, tot);
  const uint32_t nn = tot - np;
  const bool use_high = r2 && np > 0 && nn > 0;
  int req = (int)n - (int)repc;
  req = req > 1 ? req : 1;
  req = req > d.p.min_seeds ? d.p.min_seeds : req;
  if (use_high) req = d.p.min_seeds;
  uint64_t *h = d.hbuf + hoff;
  uint8_t *hc = d.hcnt + hoff;
  uint32_t ncp, ncn;
  if (lv == 1) g.sync();  // (no merge level ran: the table's copy is not behind a barrier yet)
  cm_coop_sweep(g, S, tot, np, d.p.e, req, n, m.oc, S == m.A32 ? m.B32 : m.A32, m.cc, h, hc, h + np, hc + np, &ncp, &ncn, d.prof, gl, d.n_seq);
  CM_PROF_MARK(d, g, 5);
  if (g.t == 0) { d.n_pos_hit[r] = np; d.ncp[r] = ncp; d.ncn[r] = ncn; }
  return true;
}

template <bool SLAB, class GT>
CM_HD bool cm_coop_s3b(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m) {
  const uint32_t G = (uint32_t)GT::G;
  const uint32_t tot = d.hit_tot[r];
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  const uint32_t maxf = d.round2[r] ? (uint32_t)d.p.f1 : (uint32_t)d.p.f0;
  const uint64_t SB = 1ull << 63;
  if (SLAB ? (tot > m.gcap || tot > 0xffffu) : tot > m.P) return false;  // (the sweep's offsets are 16-bit)
  uint64_t *const A = SLAB ? m.gA : m.A, *const B = SLAB ? m.gB : m.B;
  CM_PROF_BEGIN(d);
  uint32_t np;
  uint64_t *S;
  if (!SLAB) {
    uint32_t nr;
    const uint32_t lv = cm_coop_s3b_expand<uint64_t>(d, r, g, m, A, B, tot, &np, &nr);
    if (lv == 0) return false;
    CM_PROF_RESET(d);
    S = cm_coop_merge_runs(g, A, B, m.rb, m.rb2, nr, tot, lv - 1);
    CM_PROF_MARK(d, g, 4);
    CM_PROF_COUNT(d, g, 8, 1); CM_PROF_COUNT(d, g, 9, nr); CM_PROF_COUNT(d, g, 10, tot);
  } else {
    // ---- included minimizers and where their occurrences start in the list
    uint32_t R = 0, off = 0;
    for (uint32_t base = 0; base < n; base += G) {
      const uint32_t mi = base + g.t;
      uint32_t len = 0, ps = 0;
      uint64_t val = 0;
      if (mi < n) {
        const uint8_t kind = d.pr_kind[b + mi];
        if (kind != CM_PR_MISS) {
          val = d.pr_val[b + mi];
          ps = d.mm_ps[b + mi];
          if (kind == CM_PR_SINGLE) { len = 1; ps |= 1u << 31; }
          else { const uint32_t nocc = (uint32_t)val; if (nocc < maxf) len = nocc; val >>= 32; }
        }
      }
      uint32_t tv;
      const uint32_t sv = g.scan((len ? 1u << 20 : 0u) | len, &tv);  // tot <= 8192 < 2^20: both sums in one scan
      const uint32_t ri = R + (sv >> 20), o = off + (sv & 0xfffffu);
      if (len && ri < m.MM) { m.moff[ri] = o; m.mval[ri] = val; m.mps[ri] = ps; }
      R += tv >> 20;
      off += tv & 0xfffffu;
    }
    if (R > m.MM || off != tot) return false;
    if (g.t == 0) m.moff[R] = tot;
    g.sync();
    CM_PROF_MARK(d, g, 0);
    // ---- expand
    for (uint32_t x0 = g.t; x0 < tot; x0 += 8 * G) {
      uint64_t hit[8];
      uint32_t ps[8];
  #pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t x = x0 + (uint32_t)q * G;
        hit[q] = 0; ps[q] = 0;
        if (x < tot) {
          uint32_t lo = 0, hi = R;  // the largest ri with moff[ri] <= x
          while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (m.moff[mid] <= x) lo = mid; else hi = mid;
          }
          ps[q] = m.mps[lo];
          const uint64_t v = m.mval[lo];
          hit[q] = (ps[q] >> 31) ? v : d.occ[(uint32_t)v + (x - m.moff[lo])];
        }
      }
  #pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t x = x0 + (uint32_t)q * G;
        if (x < tot) {
          bool same;
          const uint64_t cp = cm_cand_from_hit(hit[q], ps[q] & 0x7fffffffu, d.p.k, &same);
          B[x] = same ? cp : (cp | SB);
        }
      }
    }
    g.sync();
    CM_PROF_MARK(d, g, 1);
    // ---- split: + hits in order, then - hits in order
    {
      const uint32_t VT = cm_coop_chunk(tot, G);
      const uint32_t c0 = cm_min_u32(tot, g.t * VT), c1 = cm_min_u32(tot, c0 + VT);
      uint32_t cnt = 0;
      for (uint32_t x = c0; x < c1; ++x) cnt += (B[x] >> 63) ? 0u : 1u;
      uint32_t pos = g.scan(cnt, &np);
      for (uint32_t x = c0; x < c1; ++x) {
        const uint64_t v = B[x];
        if (v >> 63) A[np + (x - pos)] = v; else A[pos++] = v;
      }
    }
    g.sync();
    CM_PROF_MARK(d, g, 2);
    // ---- sort
    const uint32_t nr = cm_coop_natural_runs(g, A, tot, m.rb, m.RB);
    if (nr == 0) return false;
    CM_PROF_MARK(d, g, 3);
    S = cm_coop_merge_runs(g, A, B, m.rb, m.rb2, nr, tot);
    CM_PROF_MARK(d, g, 4);
    CM_PROF_COUNT(d, g, 8, 1); CM_PROF_COUNT(d, g, 9, nr); CM_PROF_COUNT(d, g, 10, tot);
  }
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
  cm_coop_sweep(g, S, tot, np, d.p.e, req, n, SLAB ? m.goc : m.oc, S == A ? B : A, SLAB ? m.gcc : m.cc, h, hc, h + np, hc + np, &ncp, &ncn, d.prof);
  CM_PROF_MARK(d, g, 5);
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

template <bool SLAB, class GT>
CM_HD uint32_t cm_coop_rescue_dir(const CmDev &d, uint32_t r, GT &g, const CmCoopMem &m, uint64_t *out, uint8_t *outc, uint32_t n1, uint32_t cnt,
                                  bool active, const uint64_t *c0p, const uint8_t *c0c, uint64_t *zp, uint8_t *zc) {
  const int e = d.p.e;
  if (!active || cnt == 0) {
    cm_coop_copy_list(g, c0p, c0c, out, outc, n1);
    return n1;
  }
  CM_PROF_BEGIN(d);
  CM_PROF_COUNT(d, g, 63, 1);
  uint32_t nr = 0;
  // the work buffers: shared memory, or -- a list longer than that -- the hits where they are and the group's slab of global memory
  // (SLAB is a template parameter for the reason given at cm_coop_s3b; the shared-memory form also wants the read's own candidates
  // staged, so a list of those longer than the work area goes to the one-lane path too)
  uint64_t *const A = SLAB ? out + n1 : m.A, *const B = SLAB ? m.gB : m.B;
  uint16_t *const oc = SLAB ? m.goc : m.oc;
  uint8_t *const cc = SLAB ? m.gcc : m.cc;
  const bool fits = SLAB ? (cnt <= m.gcap && cnt <= 0xffffu) : (cnt <= m.P && n1 <= m.P);
  if (!SLAB && fits) {
    for (uint32_t i = g.t; i < cnt; i += (uint32_t)GT::G) m.A[i] = out[n1 + i];
    g.sync();
  }
  CM_PROF_MARK(d, g, 58);
  if (fits) nr = cm_coop_natural_runs(g, A, cnt, m.rb, m.RB);
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
  uint64_t *Ssorted = cm_coop_merge_runs(g, A, B, m.rb, m.rb2, nr, cnt);
  CM_PROF_MARK(d, g, 59);
  uint64_t *X = Ssorted;                      // the augmented list, dense: over the sorted hits once they are swept
  uint64_t *S = Ssorted == A ? B : A;         // the other buffer: parking slots of the sweep, then the staged candidates c0
  uint8_t *const cx = SLAB ? reinterpret_cast<uint8_t *>(m.gA) : m.cc2;  // the parked counts (slab mode: its first buffer is unused here)
  uint32_t naug, none;
  cm_coop_sweep(g, Ssorted, cnt, cnt, e, 1, d.mm_cnt[r], oc, S, cx, X, cc, X, cc, &naug, &none);
  g.sync();  // X / cc complete, the parking slots free
  CM_PROF_MARK(d, g, 60);
  if (naug == 0) {
    cm_coop_copy_list(g, c0p, c0c, out, outc, n1);
    return n1;
  }
  if (n1 == 0) {  // c1.swap(c2): the augmented list as it is (no distance rule)
    for (uint32_t i = g.t; i < naug; i += (uint32_t)GT::G) { const uint64_t x = X[i]; const uint8_t c = cc[i]; out[i] = x; outc[i] = c; }
    return naug;
  }
  // ---- Z = merge of c0 (staged in S when it fits) and X, ties: c0 first
  const uint64_t *a = SLAB ? c0p : S;
  if (!SLAB) {
    for (uint32_t i = g.t; i < n1; i += (uint32_t)GT::G) S[i] = c0p[i];
    g.sync();
  }
  const uint32_t nz = n1 + naug;
  const uint32_t VT = cm_coop_chunk(nz, (uint32_t)GT::G);
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
      else { zp[z] = X[ib]; zc[z] = cc[ib]; ++ib; }
    }
  }
  g.sync();
  CM_PROF_MARK(d, g, 61);
  // ---- keep / drop, compaction
  const uint64_t E = (uint64_t)(int64_t)e;
  const uint32_t mine = cm_coop_accept_walk(zp, zc, nz, z0, z1, E, nullptr, nullptr, 0);
  uint32_t total;
  const uint32_t off = g.scan(mine, &total);
  (void)cm_coop_accept_walk(zp, zc, nz, z0, z1, E, out, outc, off);
  CM_PROF_MARK(d, g, 62);
  return total;
}

template <bool SLAB, class GT>
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
  const uint32_t mcn = cm_coop_rescue_dir<SLAB>(d, r, g, m, N, NC, ncn, rn, do_n, cm_c0_neg(d, r), cm_c0_ncnt(d, r), ZP + ncp + rp, ZC + ncp + rp);
  g.sync();  // the work area is reused
  const uint32_t mcp = cm_coop_rescue_dir<SLAB>(d, r, g, m, P, PC, ncp, rp, do_p, cm_c0_pos(d, r), cm_c0_pcnt(d, r), ZP, ZC);
  if (g.t == 0) { d.mcp[r] = mcp; d.mcn[r] = mcn; }
}

// ---------------------------------------------------------------------------------------
// Mate rescue of one read and direction by a group (cm_rescue's results: the multiset of hits, their number, the
// repetitive-seed length, the return value).  One lane per minimizer walks the merged windows of the mate's best candidates in
// turn -- a binary search in the occurrence run (it starts where the previous window's ended: the chain of index.cc:443-470), then
// the hits inside the window -- and a read whose mate has hundreds of best candidates keeps its lane for milliseconds (a search
// with 300 windows in satellite arrays: ~8 ms, the whole of k_s4a/4b_rescue_list on the mosaic genome).  Here:
//   windows  the best candidates (count == max_count, fewer than 300 or the search bails out) are compacted, the merged windows
//            [es, ee] derived from neighbouring ones (a new window starts where the previous candidate's range ends below the
//            next one's start): once per direction instead of once per minimizer;
//   A        a lane per (minimizer, window) pair: lower bound of es and upper bound of ee in the minimizer's occurrence run (two
//            binary searches, all pairs side by side);
//   B        a lane per minimizer: the reference's search visits midpoints m and compares o[m] with es -- an occurrence run has
//            distinct positions in ascending order, so "o[m] < es" is "m < lower bound" and "o[m] == es" is "m == lower bound and the
//            bound is exact": the chain of searches (each starting at the previous one's last midpoint) is replayed on indices
//            alone, no memory access; the scan that follows starts at that last midpoint (one below the bound or the bound
//            itself, no lower-bound test in the reference: an occurrence just below the window is a hit too) and ends at the upper
//            bound;
//   C        the pairs' ranges are laid end to end (a scan of their lengths), the lanes take the occurrences x, x + G, ... --
//            which pair an occurrence belongs to is a search in the offsets -- and count or write the ones on the wanted strand,
//            in the order cm_rescue writes them (minimizer, window, occurrence).
// Work arrays (shared memory): see CmCoopRescueMem.  Every lane returns the same values.
// ---------------------------------------------------------------------------------------
#define CM_RESCUE_WMAX 304u   // best mate candidates a search can have (cm_rescue_bails: fewer than 300)
#define CM_RESCUE_PAIRS 640u  // (minimizer, window) pairs per round: at least two minimizers' worth
#define CM_RESCUE_SLOTS 32u   // minimizers per round
// the small layout (round 5): the rescue kernels are chains of dependent reads of the occurrence table, their rate is the number of
// searches in flight -- measured on profile 2: a wave's tables of 10.4 KB allow 15 waves per CU; with 10 KB more (7 waves) the two
// kernels take 33 + 28 ms longer, i.e. ~430 ms / (waves per CU) each.  Nearly every search has fewer than 64 best mate candidates
// (tools/coop_profile.py: none above 64 on profile 2), so the common search gets tables of 4.5 KB (32 waves per CU: the wave
// limit) and the few others are handed to a launch with the full tables.
#define CM_RESCUE_WMAX_S 64u
#define CM_RESCUE_PAIRS_S 384u
#define CM_POOL_GRANT 8192u  // (cm_coop_pool_take)
struct CmCoopRescueMem {
  uint64_t *es, *ee;   // wmax each: the merged windows
  uint64_t *bp;        // wmax: the best candidates' positions while the windows are built (overlays pa / pb)
  uint64_t *mval;      // CM_RESCUE_SLOTS: lookup result of the round's minimizers (occurrence offset << 32 | count; a singleton: the occurrence)
  uint32_t *mps;       // CM_RESCUE_SLOTS: position << 1 | strand, bit 31: singleton
  uint32_t *pa, *pb;   // pairs + 1 each: lower bound | occurrences equal to es << 30, upper bound -> first index, length -> first index, offset
  uint8_t *tq;         // pairs: where a window's search ends for each of the three places it can start (phase B)
  uint32_t *arena;     // 2: the group's grant of the rescue-hit pool: next free entry, entries left (cm_coop_rescue_mem_reset before the first search)
  uint32_t wmax, pairs;
  uint32_t grant;      // pool entries asked for at a time (CM_POOL_GRANT; tests: fewer)
};
#define CM_RESCUE_BYTES(WMAX_, PAIRS_) ((WMAX_) * 16 + CM_RESCUE_SLOTS * 12 + ((PAIRS_) + 1) * 8 + (((PAIRS_) + 3u) & ~3u) + 8 + 32)
#define CM_RESCUE_MEM_BYTES CM_RESCUE_BYTES(CM_RESCUE_WMAX, CM_RESCUE_PAIRS)
#define CM_RESCUE_MEM_BYTES_S CM_RESCUE_BYTES(CM_RESCUE_WMAX_S, CM_RESCUE_PAIRS_S)
CM_HD size_t cm_coop_rescue_mem_bytes(uint32_t wmax = CM_RESCUE_WMAX, uint32_t pairs = CM_RESCUE_PAIRS) { return (size_t)CM_RESCUE_BYTES(wmax, pairs); }
CM_HD CmCoopRescueMem cm_coop_rescue_mem_at(uint8_t *base, uint32_t wmax = CM_RESCUE_WMAX, uint32_t pairs = CM_RESCUE_PAIRS) {
  CmCoopRescueMem m;
  m.wmax = wmax; m.pairs = pairs; m.grant = CM_POOL_GRANT;
  m.es = reinterpret_cast<uint64_t *>(base);
  m.ee = m.es + wmax;
  m.mval = m.ee + wmax;
  m.mps = reinterpret_cast<uint32_t *>(m.mval + CM_RESCUE_SLOTS);
  m.pa = m.mps + CM_RESCUE_SLOTS;
  m.pb = m.pa + pairs + 1;
  m.tq = reinterpret_cast<uint8_t *>(m.pb + pairs + 1);
  m.arena = reinterpret_cast<uint32_t *>(m.tq + ((pairs + 3u) & ~3u));
  m.bp = reinterpret_cast<uint64_t *>(m.pa);  // (8-byte aligned: 32 words of mps behind 8-byte arrays; 2 x (pairs + 1) words hold wmax positions)
  return m;
}
// Pool entries are handed out to a group in grants of CM_POOL_GRANT (or what a piece needs, if more): one atomic per grant -- same-address
// atomics retire at ~90 per microsecond, and an atomic per piece (4 M of them per batch of profile 2) cost the counting kernel 60 ms
template <class GT>
CM_HD void cm_coop_rescue_mem_reset(GT &g, const CmCoopRescueMem &m) {
  if (g.t == 0) { m.arena[0] = 0; m.arena[1] = 0; }
  g.sync();
}
// `need` entries of the pool for this group: their first index, or all ones (no room); every lane gets it.  Once the pool has
// refused a request the group asks no more (arena[1] all ones) and only adds up what it would have needed: cm_coop_rescue_mem_flush
// adds that to the cursor, from which the host sizes the next range's pool
template <class GT>
CM_HD uint32_t cm_coop_pool_take(const CmDev &d, GT &g, const CmCoopRescueMem &m, uint32_t need) {
  uint32_t at = 0xffffffffu;
  if (g.t == 0) {
    if (m.arena[1] == 0xffffffffu) m.arena[0] += need;
    else if (m.arena[1] >= need) { at = m.arena[0]; m.arena[0] += need; m.arena[1] -= need; }
    else {
      const unsigned long long grant = need > m.grant ? need : m.grant;
      const unsigned long long a0 = cm_fetch_add64(&d.stats[CM_ST_POOL], grant);
      if (a0 + grant <= d.rs_pool_cap) { at = (uint32_t)a0; m.arena[0] = at + need; m.arena[1] = (uint32_t)grant - need; }
      else if (a0 + need <= d.rs_pool_cap) { at = (uint32_t)a0; m.arena[0] = 0; m.arena[1] = 0; }  // (the pool's last entries)
      else { m.arena[0] = 0; m.arena[1] = 0xffffffffu; }  // (the cursor has the grant: at least what was needed)
    }
  }
  return ~cm_coop_bcast0(g, ~at);
}
template <class GT>
CM_HD void cm_coop_rescue_mem_flush(const CmDev &d, GT &g, const CmCoopRescueMem &m) {
  g.sync();
  if (g.t == 0 && m.arena[1] == 0xffffffffu && m.arena[0]) (void)cm_fetch_add64(&d.stats[CM_ST_POOL], (unsigned long long)m.arena[0]);
}
// how many of a mate's candidates have the best count (what cm_coop_rescue's tables must hold): every lane gets it
template <class GT>
CM_HD uint32_t cm_coop_rescue_best_num(GT &g, const uint8_t *mc, uint32_t mn) {
  uint32_t lmax = 0;
  for (uint32_t i = g.t; i < mn; i += (uint32_t)GT::G) lmax = mc[i] > lmax ? mc[i] : lmax;
  const uint32_t mx = (uint32_t)g.max64((uint64_t)lmax);
  uint32_t lnum = 0;
  for (uint32_t i = g.t; i < mn; i += (uint32_t)GT::G) lnum += mc[i] == mx ? 1u : 0u;
  return g.sum(lnum);
}
// do both searches of read r fit tables with room for wmax best candidates?  (cm_coop_s4a_rescue / cm_coop_s4b_fill with such tables)
template <class GT>
CM_HD bool cm_coop_rescue_fits(const CmDev &d, uint32_t r, GT &g, uint32_t wmax) {
  const uint32_t o = r ^ 1u;
  const uint32_t a = d.ncp[o] > wmax ? cm_coop_rescue_best_num(g, cm_c0_pcnt(d, o), d.ncp[o]) : 0u;
  const uint32_t b = d.ncn[o] > wmax ? cm_coop_rescue_best_num(g, cm_c0_ncnt(d, o), d.ncn[o]) : 0u;
  // (a search that bails out -- 300 best candidates or more, cm_rescue_bails -- needs no tables at all)
  return (a <= wmax || a >= 300u) && (b <= wmax || b >= 300u);
}
// phase C of cm_coop_rescue: the occurrences of the pairs' ranges (first index pa, offsets pb, pb[np] = total) on the wanted strand,
// counted (*cnt += their number) and, out != nullptr, written from out[*cnt] on in pair / occurrence order
template <class GT>
CM_HD void cm_coop_rescue_emit(const CmDev &d, GT &g, const CmCoopRescueMem &m, uint32_t np, uint32_t W, uint32_t total, int strand, uint64_t *out,
                               uint32_t *cnt_io) {
  const uint32_t G = (uint32_t)GT::G;
  uint32_t cnt = *cnt_io;
  for (uint32_t x0 = 0; x0 < total; x0 += 4 * G) {  // four occurrences per lane and round: their loads in flight together
    uint64_t hit[4];
    uint32_t psv[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t x = x0 + (uint32_t)u * G + g.t;
      in[u] = x < total;
      hit[u] = 0; psv[u] = 0;
      if (in[u]) {
        uint32_t lo = 0, hi = np;  // the largest q with pb[q] <= x (the pairs without occurrences share their successor's offset)
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (m.pb[mid] <= x) lo = mid; else hi = mid;
        }
        const uint32_t sl = lo / W;
        const uint32_t ps = m.mps[sl];
        const uint64_t val = m.mval[sl];
        psv[u] = ps & 0x7fffffffu;
        hit[u] = (ps >> 31) ? val : d.occ[(uint32_t)(val >> 32) + m.pa[lo] + (x - m.pb[lo])];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool same = false;
      const uint64_t cp = cm_cand_from_hit(hit[u], psv[u], d.p.k, &same);
      const bool match = in[u] && ((same && strand == 0) || (!same && strand == 1));
      uint32_t tot;
      const uint32_t at = g.scan(match ? 1u : 0u, &tot);
      if (match && out) out[cnt + at] = cp;
      cnt += tot;
    }
  }
  *cnt_io = cnt;
}
template <class GT>
CM_HD int cm_coop_rescue(const CmDev &d, uint32_t r, int strand, const uint64_t *mp, const uint8_t *mc, uint32_t mn, GT &g, const CmCoopRescueMem &m,
                         uint64_t *out, uint32_t *n_out, uint32_t *rep_len_out, uint32_t *pool_off = nullptr) {
  const uint32_t G = (uint32_t)GT::G;
  *n_out = 0;
  *rep_len_out = 0;
  CM_PROF_BEGIN(d);
  // ---- the best count among the mate's candidates and how many have it
  uint32_t lmax = 0;
  for (uint32_t i = g.t; i < mn; i += G) lmax = mc[i] > lmax ? mc[i] : lmax;
  const int max_count = (int)g.max64((uint64_t)lmax);
  uint32_t lnum = 0;
  for (uint32_t i = g.t; i < mn; i += G) lnum += (int)mc[i] == max_count ? 1u : 0u;
  const int best_num = (int)g.sum(lnum);
  if (cm_rescue_bails(d, max_count, best_num, mn)) return -max_count;
  // ---- the best candidates' positions in order, then the merged windows
  const uint64_t sr = 2ull * (uint64_t)(uint32_t)d.p.max_insert;
  uint32_t nb = 0;
  for (uint32_t base = 0; base < mn; base += G) {
    const uint32_t i = base + g.t;
    const bool is = i < mn && (int)mc[i] == max_count;
    uint32_t tot;
    const uint32_t at = g.scan(is ? 1u : 0u, &tot);
    if (is) m.bp[nb + at] = mp[i];
    nb += tot;
  }
  g.sync();
  // candidate j starts a window when the previous candidate's range ends below its own start (index.cc:383-412: the ranges are
  // merged in candidate order); it ends one when the next candidate starts one.  A window's number = the starts before it.
  uint32_t W = 0;
  for (uint32_t base = 0; base < nb; base += G) {
    const uint32_t j = base + g.t;
    bool st = false, last = false;
    uint64_t pos = 0, s = 0;
    if (j < nb) {
      pos = m.bp[j];
      s = pos < sr ? 0 : pos - sr;
      st = j == 0 || m.bp[j - 1] + sr < s;
      if (j + 1 == nb) last = true;
      else { const uint64_t pn = m.bp[j + 1]; last = pos + sr < (pn < sr ? 0 : pn - sr); }
    }
    uint32_t tot;
    const uint32_t at = g.scan(st ? 1u : 0u, &tot);
    const uint32_t w = W + at + (st ? 1u : 0u) - 1u;  // the window candidate j lies in
    if (st) m.es[w] = s;
    if (last) m.ee[w] = pos + sr;
    W += tot;
  }
  g.sync();  // (bp is dead from here on: pa / pb overlay it)
  CM_PROF_MARK(d, g, 48);
  CM_PROF_COUNT(d, g, 55, 1);
  // ---- the minimizers, CM_RESCUE_SLOTS (or as many as m.pairs pairs hold) per round
  const uint32_t b = d.mm_off[r], n = d.mm_cnt[r];
  const bool no_window = W == 0;  // (no mate candidate: cm_rescue then finds the singletons only; callers do not ask)
  if (no_window) { W = 1; if (g.t == 0) { m.es[0] = ~0ull; m.ee[0] = 0; } g.sync(); }
  uint32_t per_round = m.pairs / W;
  if (per_round > CM_RESCUE_SLOTS) per_round = CM_RESCUE_SLOTS;
  if (per_round == 0) per_round = 1;
  uint32_t cnt = 0;  // hits so far (uniform)
  bool pool_ok = !out && pool_off && d.rs_pool;
  uint32_t pool_first = 0xffffffffu, pool_last = 0xffffffffu, pool_last_cnt = 0;
  for (uint32_t m0 = 0; m0 < n; m0 += per_round) {
    const uint32_t ns = n - m0 < per_round ? n - m0 : per_round;
    for (uint32_t s = g.t; s < ns; s += G) {
      const uint8_t kind = d.pr_kind[b + m0 + s];
      uint32_t ps = d.mm_ps[b + m0 + s];
      uint64_t val = d.pr_val[b + m0 + s];
      if (kind == CM_PR_MISS) val = 0;              // no occurrences
      else if (kind == CM_PR_SINGLE) ps |= 1u << 31;
      m.mval[s] = val;
      m.mps[s] = ps;
    }
    g.sync();
    CM_PROF_MARK(d, g, 49);
    const uint32_t np = ns * W;
    // -- A: bounds of every (minimizer, window) pair, four pairs interleaved per lane -- the searches are chains of dependent loads at
    //    global-memory latency and nothing else, so the requests in flight per lane are what the phase's duration divides by.  The lower
    //    bound of es by bisection; the upper bound of ee by galloping from there (a window holds a handful of a minimizer's occurrences:
    //    two or three probes in the cache lines the lower bound just touched instead of another ~13 random ones -- the rescue searches
    //    of a repeat-rich batch are bound by the number of random reads of the occurrence table, 4 G of them per 4 M pairs of profile 2)
    for (uint32_t q0 = g.t; q0 < np; q0 += 4 * G) {
      uint32_t lo[4], hi[4], st[4];
      const uint64_t *o[4];
      uint64_t kes[4], kee[4];
      bool act[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = q0 + (uint32_t)u * G;
        lo[u] = 0; hi[u] = 0; st[u] = 1; o[u] = d.occ; kes[u] = 0; kee[u] = 0; act[u] = false;
        if (q < np) {
          const uint32_t s = q / W, w = q - s * W;
          if (!(m.mps[s] >> 31)) {
            const uint64_t val = m.mval[s];
            hi[u] = (uint32_t)val;
            o[u] = d.occ + (uint32_t)(val >> 32);
            kes[u] = m.es[w]; kee[u] = m.ee[w];
            act[u] = true;
          }
        }
      }
      for (;;) {  // lower bounds: first index with o >> 1 >= es
        // (round 6, measured and not kept: the range cut in four per step -- three probes requested together, ~7 dependent trips instead of
        //  ~14 -- made the phase SLOWER, 70 k -> 88 k cycles per search on profile 2: with four pairs interleaved per lane and 20-30 waves
        //  per CU the occurrence table's random reads are bound by their number, not by the chain's length)
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (lo[u] < hi[u]) {
            const uint32_t mid = (lo[u] + hi[u]) >> 1;
            if ((o[u][mid] >> 1) < kes[u]) lo[u] = mid + 1; else hi[u] = mid;
            any = true;
          }
        if (!any) break;
      }
      uint32_t l2[4], h2[4], nocc[4];
      uint64_t v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = q0 + (uint32_t)u * G;
        nocc[u] = 0;
        if (act[u]) { const uint32_t s = q / W; nocc[u] = (uint32_t)m.mval[s]; }
        l2[u] = lo[u]; h2[u] = 0xffffffffu;  // every index below l2 holds a position <= ee; h2: an index known to hold one above (none yet)
        // occurrences at exactly es (two at most: one per strand; none for an index built by the reference, whose minimizers have one strand
        // per position): the search's "equal" outcome for the midpoints lb .. lb + eq - 1.  Requested here, four pairs' worth together
        // (round 6: a pass of its own after this loop cost a trip to the occurrence table per pair and lane -- a third of "eq + B")
        v0[u] = act[u] && lo[u] < nocc[u] ? o[u][lo[u]] >> 1 : ~0ull;
        v1[u] = act[u] && lo[u] + 1 < nocc[u] ? o[u][lo[u] + 1] >> 1 : ~0ull;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = q0 + (uint32_t)u * G;
        if (q < np) m.pa[q] = act[u] ? lo[u] | (((v0[u] == kes[u] ? 1u : 0u) + (v1[u] == kes[u] ? 1u : 0u)) << 30) : lo[u];
      }
      for (;;) {  // upper bounds: gallop (probe l2, l2 + 1, l2 + 3, ...) until a position above ee or the run's end, then bisect
        bool any = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!act[u]) continue;
          if (h2[u] == 0xffffffffu) {  // galloping
            const uint32_t probe = l2[u] + st[u] - 1;
            if (probe >= nocc[u]) { h2[u] = nocc[u]; }
            else if ((o[u][probe] >> 1) <= kee[u]) { l2[u] = probe + 1; st[u] <<= 1; }
            else h2[u] = probe;
            any = true;
          } else if (l2[u] < h2[u]) {
            const uint32_t mid = (l2[u] + h2[u]) >> 1;
            if ((o[u][mid] >> 1) <= kee[u]) l2[u] = mid + 1; else h2[u] = mid;
            any = true;
          }
        }
        if (!any) break;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = q0 + (uint32_t)u * G;
        if (q < np) m.pb[q] = act[u] ? l2[u] : 0u;
      }
    }
    g.sync();
    CM_PROF_MARK(d, g, 50);
    // -- B: the chain of searches per minimizer, on indices alone: first index and length of every pair's scan.  The reference's search
    //    for window w starts at l = the last midpoint of window w - 1's search, and a search that starts at or below the lower bound lb ends
    //    with its last midpoint at lb - 1 or lb (nothing equal to es), or at lb / lb + 1 (something equal); one that starts above lb stays
    //    where it started.  So a search starts at lb(w - 1) - 1, lb(w - 1) or lb(w - 1) + 1: B1 replays it for all three, a lane per
    //    (minimizer, window) pair, and keeps where each ends relative to lb(w); B2, a lane per minimizer, only follows the chain through
    //    these tables (a window whose numbers do not fit the scheme is replayed in full there: the result is the replay's whatever the
    //    argument above is worth).  Round 4 replayed every window's ~12 steps in B2: W x 12 dependent iterations on 8 lanes of 64 --
    //    as long as phase A's chains of loads for a search with 200 windows.
    for (uint32_t q = g.t; q < np; q += G) {
      const uint32_t s = q / W, w = q - s * W;
      uint8_t t = 0xff;
      const uint32_t nocc = (m.mps[s] >> 31) ? 0u : (uint32_t)m.mval[s];
      if (nocc) {
        const uint32_t lbx = m.pa[q];
        const int32_t lb = (int32_t)(lbx & 0x3fffffffu), le = lb + (int32_t)(lbx >> 30);
        const int32_t lbp = w ? (int32_t)(m.pa[q - 1] & 0x3fffffffu) : 0;
        uint32_t enc = 0;
        bool ok = true;
#pragma unroll
        for (int dl = -1; dl <= 1; ++dl) {
          int32_t l = w ? lbp + dl : 0;
          l = l < 0 ? 0 : (l > (int32_t)nocc - 1 ? (int32_t)nocc - 1 : l);  // (a start outside the run is one no search ends at)
          int32_t mid = 0, rr = (int32_t)(nocc - 1);
          while (l <= rr) {
            mid = (l + rr) / 2;
            if (mid < lb) l = mid + 1;
            else if (mid >= le) rr = mid - 1;
            else break;
          }
          const int32_t dd = mid - lb;
          if (dd < -1 || dd > 1) ok = false;
          enc |= (uint32_t)((dd + 1) & 3) << (2 * (dl + 1));
        }
        if (ok) t = (uint8_t)enc;
      }
      m.tq[q] = t;
    }
    g.sync();
    // Round 6: B2 in three steps.  (a) A lane per minimizer follows the chain through the tables ALONE -- state = where the previous window's
    // search ended relative to its lower bound (0 .. 2, a table entry), one shift per window, the table bytes four at a time and not
    // dependent on the state; an unusable table (0xff) yields state 3.  The state goes to the two free top bits of pb.  (b) A lane per
    // pair checks what (a) took for granted: a usable table, and the start it implies inside the run; a minimizer with a window that
    // fails is flagged (bit 30 of mps).  (c) A lane per pair writes first index / length from its own lower bound and state; the flagged
    // minimizers are replayed window by window as in round 5 (exact whatever the tables are worth).  Round 5 did ALL of it in (a)'s
    // loop: three dependent shared-memory reads, a dozen instructions and two writes per window on ns of 64 lanes -- for a search with
    // 200 windows longer than phase A's loads (tools/coop_profile.py: 110 k of a search's 280 k cycles).
    for (uint32_t s = g.t; s < ns; s += G) {
      if ((m.mps[s] >> 31) || (uint32_t)m.mval[s] == 0 || (uint32_t)m.mval[s] >= (1u << 30)) continue;  // (a run that long: bounds take all 32 bits -- replayed)
      const uint8_t *tqs = m.tq + (size_t)s * W;
      uint32_t *pbs = m.pb + (size_t)s * W;
      uint32_t st = 1;  // (window 0's three entries are equal: any state will do)
      uint32_t w = 0;
      for (; w + 4 <= W; w += 4) {
        const uint32_t t0 = tqs[w], t1 = tqs[w + 1], t2 = tqs[w + 2], t3 = tqs[w + 3];
        const uint32_t u0 = pbs[w], u1 = pbs[w + 1], u2 = pbs[w + 2], u3 = pbs[w + 3];
        st = (t0 >> (2 * st)) & 3u; pbs[w] = u0 | st << 30;
        st = (t1 >> (2 * st)) & 3u; pbs[w + 1] = u1 | st << 30;
        st = (t2 >> (2 * st)) & 3u; pbs[w + 2] = u2 | st << 30;
        st = (t3 >> (2 * st)) & 3u; pbs[w + 3] = u3 | st << 30;
      }
      for (; w < W; ++w) { st = ((uint32_t)tqs[w] >> (2 * st)) & 3u; pbs[w] |= st << 30; }
    }
    g.sync();
    for (uint32_t q = g.t; q < np; q += G) {
      const uint32_t s = q / W, w = q - s * W;
      if (m.mps[s] >> 31) continue;
      const uint32_t nocc = (uint32_t)m.mval[s];
      if (!nocc) continue;
      const uint32_t st = m.pb[q] >> 30;
      bool ok = st != 3u && m.tq[q] != 0xff && nocc < (1u << 30);
      if (w) {  // the start of this window's search: the previous window's last midpoint, inside the run
        const uint32_t pst = m.pb[q - 1] >> 30;
        const int32_t prev_l = (int32_t)(m.pa[q - 1] & 0x3fffffffu) + (int32_t)pst - 1;
        ok = ok && pst != 3u && prev_l >= 0 && prev_l <= (int32_t)nocc - 1;
      }
#ifdef CM_DBG_FORCE_REPLAY
      if (CM_DBG_FORCE_REPLAY && (s & 1u)) ok = false;  // (tests: no table has been seen to fail -- the replay is exercised by decree)
#endif
      if (!ok) m.mps[s] |= 1u << 30;  // (every lane that writes, writes the same bit)
    }
    g.sync();
    for (uint32_t q = g.t; q < np; q += G) {
      const uint32_t s = q / W, w = q - s * W;
      const uint32_t ps = m.mps[s];
      if (ps >> 31) { m.pa[q] = 0; m.pb[q] = w == 0 ? 1u : 0u; continue; }  // a singleton: its one occurrence, whatever the windows (cm_rescue_minimizer)
      const uint32_t nocc = (uint32_t)m.mval[s];
      if (!nocc) { m.pa[q] = 0; m.pb[q] = 0; continue; }
      if (ps & (1u << 30)) continue;  // replayed below
      const uint32_t x = m.pb[q], ub = x & 0x3fffffffu;
      const uint32_t first = (uint32_t)((int32_t)(m.pa[q] & 0x3fffffffu) + (int32_t)(x >> 30) - 1);
      m.pa[q] = first;
      m.pb[q] = ub > first && !no_window ? ub - first : 0u;
    }
    g.sync();  // (the flags are read above and cleared below)
    for (uint32_t s = g.t; s < ns; s += G) {
      const uint32_t ps = m.mps[s];
      if ((ps >> 31) || !(ps & (1u << 30))) continue;
#ifdef CM_DBG_REPLAY
      ++CM_DBG_REPLAY;
#endif
      m.mps[s] = ps & ~(1u << 30);
      const uint64_t val = m.mval[s];
      const uint32_t nocc = (uint32_t)val;
      int32_t prev_l = 0, lbp = 0;
      for (uint32_t w = 0; w < W; ++w) {
        const uint32_t lbx = m.pa[s * W + w], ub = nocc < (1u << 30) ? m.pb[s * W + w] & 0x3fffffffu : m.pb[s * W + w];
        uint32_t first = 0, len = 0;
        if (nocc) {
          const int32_t lb = (int32_t)(lbx & 0x3fffffffu), le = lb + (int32_t)(lbx >> 30);
          const uint32_t t = m.tq[s * W + w];
          const int32_t dl = w ? prev_l - lbp : 0;
          int32_t mid;
          if (t != 0xff && dl >= -1 && dl <= 1 && (w == 0 || (prev_l >= 0 && prev_l <= (int32_t)nocc - 1))) {
            mid = lb + (int32_t)((t >> (2 * (dl + 1))) & 3u) - 1;
          } else {
            int32_t l = prev_l, rr = (int32_t)(nocc - 1);
            mid = 0;
            while (l <= rr) {
              mid = (l + rr) / 2;
              if (mid < lb) l = mid + 1;
              else if (mid >= le) rr = mid - 1;
              else break;
            }
          }
          prev_l = mid;
          lbp = lb;
          first = (uint32_t)mid;
          len = ub > first && !no_window ? ub - first : 0u;
        }
        m.pa[s * W + w] = first;
        m.pb[s * W + w] = len;
      }
    }
    g.sync();
    CM_PROF_MARK(d, g, 51);
    // -- C: exclusive scan of the lengths in pair order (pb), then the occurrences themselves
    uint32_t total;
    {
      const uint32_t VT = cm_coop_chunk(np, G);
      const uint32_t c0 = cm_min_u32(np, g.t * VT), c1 = cm_min_u32(np, c0 + VT);
      uint32_t sum = 0;
      for (uint32_t q = c0; q < c1; ++q) sum += m.pb[q];
      uint32_t run = g.scan(sum, &total);
      for (uint32_t q = c0; q < c1; ++q) { const uint32_t x = m.pb[q]; m.pb[q] = run; run += x; }
      if (g.t == 0) m.pb[np] = total;
      g.sync();
    }
    // counting pass with the pool at hand: the round's hits are written there right away, as a piece of their own -- one header entry
    // (hits << 32 | index of the search's next piece, all ones: none), then the hits -- so that the fill pass copies instead of finding the
    // bounds a second time (phase A: ~17 dependent random reads of the occurrence table per pair, which is what a search costs).  Round 4
    // kept only searches whose tables fit one round; the searches that take several are the long ones (200 windows and more).
    // Round 6: ONE pass over the occurrences instead of a counting one and a writing one -- the piece is taken for `total` hits (the
    // ranges' occurrences on both strands), what the wanted strand leaves unused goes back to the group's grant.
    bool emitted = false;
    if (pool_ok && total > 0) {
      const uint32_t at = cm_coop_pool_take(d, g, m, total + 1u);
      if (at != 0xffffffffu) {
        uint32_t c2 = 0;
        cm_coop_rescue_emit(d, g, m, np, W, total, strand, d.rs_pool + at + 1, &c2);
        emitted = true;
        cnt += c2;
        if (g.t == 0) {
          // (give back: only when the piece is the last thing the grant handed out -- it is, unless it was cut from the pool's last entries)
          const uint32_t used = c2 ? c2 + 1u : 0u;
          if (m.arena[1] != 0xffffffffu && m.arena[0] == at + total + 1u) { m.arena[0] = at + used; m.arena[1] += total + 1u - used; }
          if (c2) {
            d.rs_pool[at] = ((uint64_t)c2 << 32) | 0xffffffffull;
            if (pool_last != 0xffffffffu) d.rs_pool[pool_last] = ((uint64_t)pool_last_cnt << 32) | at;
          }
        }
        if (c2) {
          if (pool_first == 0xffffffffu) pool_first = at;
          pool_last = at; pool_last_cnt = c2;
        }
      } else {
        pool_ok = false;  // no room: the fill pass searches again
      }
    }
    if (!emitted) cm_coop_rescue_emit(d, g, m, np, W, total, strand, out, &cnt);
    CM_PROF_MARK(d, g, 52);
    CM_PROF_COUNT(d, g, 56, total);
    g.sync();  // the tables serve the next round
    CM_PROF_MARK(d, g, 53);
    CM_PROF_COUNT(d, g, 57, 1);
  }
  if (pool_ok && g.t == 0) *pool_off = pool_first;
  // ---- repetitive_seed_length over the minimizers in order
  uint32_t rep_len = 0;
  if (n <= m.pairs) {  // (round 6) every lane fetches a minimizer's lookup result, lane 0 adds up from shared memory: one trip to global memory, not n
    for (uint32_t mi = g.t; mi < n; mi += G) {
      const uint8_t kind = d.pr_kind[b + mi];
      uint32_t rp = 0xffffffffu;  // not a repetitive seed (cm_rescue_rep)
      if (kind != CM_PR_MISS && kind != CM_PR_SINGLE && (uint32_t)d.pr_val[b + mi] >= (uint32_t)d.p.f0) rp = d.mm_ps[b + mi] >> 1;
      m.pa[mi] = rp;
    }
    g.sync();
    if (g.t == 0) {
      uint32_t prev_rep = ~0u;
      const uint32_t kk = (uint32_t)d.p.k, ww = (uint32_t)d.p.w;
      for (uint32_t mi = 0; mi < n; ++mi) {
        const uint32_t rp = m.pa[mi];
        if (rp == 0xffffffffu) continue;
        if (prev_rep > rp) rep_len += kk;
        else if (rp < prev_rep + kk + ww - 1) rep_len += rp - prev_rep;
        else rep_len += kk;
        prev_rep = rp;
      }
    }
  } else if (g.t == 0) {
    uint32_t prev_rep = ~0u;
    for (uint32_t mi = 0; mi < n; ++mi) cm_rescue_rep(d, d.pr_kind[b + mi], d.pr_val[b + mi], d.mm_ps[b + mi], &rep_len, &prev_rep);
  }
  *rep_len_out = cm_coop_bcast0(g, rep_len);
  CM_PROF_MARK(d, g, 54);
  *n_out = cnt;
  return max_count;
}

// S4a's two searches for a read whose mate has many candidates (cm_s4a_rescue's results) ...
template <class GT>
CM_HD void cm_coop_s4a_rescue(const CmDev &d, uint32_t r, GT &g, const CmCoopRescueMem &m) {
  const uint32_t o = r ^ 1u;
  uint32_t cntn = 0, cntp = 0, rl = 0, rl_val = 0;
  int res_neg = 0, res_pos = 0;
  bool set_rl = false;
  uint32_t *const po = d.rs_pool ? d.rs_pool_off + 2 * (size_t)r : nullptr;
  if (po && g.t == 0) { po[0] = 0xffffffffu; po[1] = 0xffffffffu; }
  if (d.ncp[o] > 0) {  // the mate's + candidates drive a search on our - strand (candidate_processor.cc:147-153)
    res_neg = cm_coop_rescue(d, r, 1, cm_c0_pos(d, o), cm_c0_pcnt(d, o), d.ncp[o], g, m, nullptr, &cntn, &rl, po ? po + 1 : nullptr);
    if (res_neg >= 0) { set_rl = true; rl_val = rl; }
  }
  if (d.ncn[o] > 0) {
    res_pos = cm_coop_rescue(d, r, 0, cm_c0_neg(d, o), cm_c0_ncnt(d, o), d.ncn[o], g, m, nullptr, &cntp, &rl, po);
    if (res_pos >= 0) { set_rl = true; rl_val = rl; }
  }
  if (g.t == 0) {
    d.aug[r] = 1;
    d.res_neg[r] = res_neg; d.res_pos[r] = res_pos;
    d.resc_n[r] = cntn; d.resc_p[r] = cntp;
    if (set_rl) d.rep_len[r] = rl_val;  // repetitive_seed_length overwritten (:113,131, index.cc:487)
    d.m_tot[r] = d.ncp[r] + d.ncn[r] + cntn + cntp;
  }
}
// the hits a counting search left in the pool (its pieces in order: cm_coop_rescue), copied to where the fill pass writes them
template <class GT>
CM_HD void cm_coop_pool_copy(const CmDev &d, GT &g, uint32_t at, uint64_t *dst) {
  uint32_t o = 0;
  while (at != 0xffffffffu) {
    const uint64_t hdr = d.rs_pool[at];
    const uint32_t c = (uint32_t)(hdr >> 32);
    for (uint32_t i = g.t; i < c; i += (uint32_t)GT::G) dst[o + i] = d.rs_pool[at + 1 + i];
    o += c;
    at = (uint32_t)hdr;
  }
}
// ... and S4b's fill pass for it: the rescue hits where cm_s4b_rescue_merge(CM_S4B_FILL_ONLY) writes them, in the same order
template <class GT>
CM_HD void cm_coop_s4b_fill(const CmDev &d, uint32_t r, GT &g, const CmCoopRescueMem &m) {
  const uint32_t o = r ^ 1u;
  const uint32_t ncp = d.ncp[r], ncn = d.ncn[r], rp = d.resc_p[r], rn = d.resc_n[r];
  uint64_t *P = d.mbuf + d.m_off[r];
  uint64_t *N = P + ncp + rp;
  uint32_t cnt, rl;
  if (g.t == 0) { d.mcp[r] = 0; d.mcn[r] = 0; }  // (until the group that sorts and merges the hits has run)
  const uint32_t off_p = d.rs_pool ? d.rs_pool_off[2 * (size_t)r] : 0xffffffffu, off_n = d.rs_pool ? d.rs_pool_off[2 * (size_t)r + 1] : 0xffffffffu;
  if (d.ncp[o] > 0 && d.res_neg[r] >= 0 && rn > 0) {
    if (off_n != 0xffffffffu) cm_coop_pool_copy(d, g, off_n, N + ncn);  // found while counting
    else (void)cm_coop_rescue(d, r, 1, cm_c0_pos(d, o), cm_c0_pcnt(d, o), d.ncp[o], g, m, N + ncn, &cnt, &rl);
  }
  if (d.ncn[o] > 0 && d.res_pos[r] >= 0 && rp > 0) {
    if (off_p != 0xffffffffu) cm_coop_pool_copy(d, g, off_p, P + ncp);
    else (void)cm_coop_rescue(d, r, 0, cm_c0_neg(d, o), cm_c0_ncnt(d, o), d.ncn[o], g, m, P + ncp, &cnt, &rl);
  }
}

// ---------------------------------------------------------------------------------------
// Array helpers over the group: exclusive prefix sum / exclusive prefix maximum of a[0..n) in place (shared memory),
// every lane a contiguous chunk, the chunk totals through the group.  Return the total / the overall maximum.
// ---------------------------------------------------------------------------------------
template <class GT>
CM_HD uint32_t cm_coop_array_scan_add(GT &g, uint16_t *a, uint32_t n) {
  const uint32_t VT = cm_coop_chunk(n, (uint32_t)GT::G);
  const uint32_t c0 = cm_min_u32(n, g.t * VT), c1 = cm_min_u32(n, c0 + VT);
  uint32_t sum = 0;
  for (uint32_t i = c0; i < c1; ++i) sum += a[i];
  uint32_t total;
  uint32_t run = g.scan(sum, &total);
  for (uint32_t i = c0; i < c1; ++i) { const uint32_t x = a[i]; a[i] = (uint16_t)run; run += x; }
  g.sync();
  return total;
}
template <class GT>
CM_HD uint32_t cm_coop_array_scan_max(GT &g, uint8_t *a, uint32_t n) {
  const uint32_t VT = cm_coop_chunk(n, (uint32_t)GT::G);
  const uint32_t c0 = cm_min_u32(n, g.t * VT), c1 = cm_min_u32(n, c0 + VT);
  uint32_t mx = 0;
  for (uint32_t i = c0; i < c1; ++i) mx = a[i] > mx ? a[i] : mx;
  uint32_t total;
  uint32_t run = g.scanmax(mx, &total);
  for (uint32_t i = c0; i < c1; ++i) { const uint32_t x = a[i]; a[i] = (uint8_t)run; run = x > run ? x : run; }
  g.sync();
  return total;
}

// ---------------------------------------------------------------------------------------
// ReduceCandidatesForPairedEndReadOnOneDirection (candidate_processor.cc:416-484; cm_reduce_dir in cm_stages.h) by a
// group.  The sequential two-pointer loop over the ascending lists p1 (n1) and p2 (n2) is equivalent to:
//   lo(i)   = first j with p2[j] + dist >= p1[i]   (where the loop's i2 stands when it looks at p1[i]; monotone)
//   the loop ends before the first i with lo(i) == n2 (i_end): later entries of list 1 are never looked at
//   paired(i)  <=> lo(i) < n2 and p2[lo(i)] <= p1[i] + dist; such an entry is kept and raises max1 = max(6, counts of kept paired)
//   an unpaired i < i_end is kept iff same sequence as p2[lo(i)], count >= max1 so far, and it is one of the first five such
//   covered(j) <=> some i has |p1[i] - p2[j]| <= dist  (first i with p1[i] + dist >= p2[j] has p1[i] <= p2[j] + dist): kept,
//              raises max2 = max(6, counts of covered entries before)
//   an uncovered j is looked at by sk(j) = first i with p1[i] > p2[j] + dist, if there is one; kept iff same sequence,
//              count >= max2 so far, one of the first five such   (the loop's "i2 >= prev_end" test holds exactly for the
//              uncovered entries)
// Outputs in list order.  Work arrays (shared): lo1, of1, of2 (u16), k1, k2, x1, x2 (u8), n1 / n2 entries.
// ---------------------------------------------------------------------------------------
struct CmCoopPairMem {
  uint64_t *s1, *s2;          // P each: the two position lists, staged (their binary searches are chains of dependent loads)
  uint16_t *lo1, *of1, *of2;  // P each
  uint8_t *k1, *k2, *x1, *x2; // P each
  uint8_t *sc1, *sc2;         // P each: the two count lists, staged with the positions (round 6: the counts were read where they lie in three of the
                              // function's passes -- three more trips to global memory per direction in a function that is a dozen short passes)
  uint32_t P;
};
// staged == false: no room for the position lists (10 bytes per entry instead of 28) -- the form for the few pairs with lists
// beyond the staged classes, whose searches then run on the lists where they are (cm_coop_reduce_dir<false>)
CM_HD size_t cm_coop_pair_mem_bytes(uint32_t P, bool staged = true) { return (size_t)P * (staged ? 28 : 10) + 32; }
CM_HD CmCoopPairMem cm_coop_pair_mem_at(uint8_t *base, uint32_t P, bool staged = true) {
  CmCoopPairMem m;
  m.P = P;
  m.s1 = reinterpret_cast<uint64_t *>(base);
  m.s2 = m.s1 + (staged ? P : 0);
  m.lo1 = reinterpret_cast<uint16_t *>(m.s2 + (staged ? P : 0));
  m.of1 = m.lo1 + P;
  m.of2 = m.of1 + P;
  m.k1 = reinterpret_cast<uint8_t *>(m.of2 + P);
  m.k2 = m.k1 + P;
  m.x1 = m.k2 + P;
  m.x2 = m.x1 + P;
  m.sc1 = staged ? m.x2 + P : nullptr;
  m.sc2 = staged ? m.sc1 + P : nullptr;
  return m;
}
template <bool STAGED, class GT>
CM_HD void cm_coop_reduce_dir(GT &g, const CmCoopPairMem &m, uint32_t dist, const uint64_t *gp1, const uint8_t *c1, uint32_t n1, const uint64_t *gp2,
                              const uint8_t *c2, uint32_t n2, uint64_t *f1, uint8_t *fc1, uint32_t *nf1, uint64_t *f2, uint8_t *fc2, uint32_t *nf2) {
  const uint32_t G = (uint32_t)GT::G;
  if (STAGED) {
    for (uint32_t i = g.t; i < n1; i += G) { m.s1[i] = gp1[i]; m.sc1[i] = c1[i]; }
    for (uint32_t j = g.t; j < n2; j += G) { m.s2[j] = gp2[j]; m.sc2[j] = c2[j]; }
  }
  g.sync();
  const uint64_t *p1 = STAGED ? m.s1 : gp1, *p2 = STAGED ? m.s2 : gp2;  // (compile-time choice: see cm_coop_s3b)
  if (STAGED) { c1 = m.sc1; c2 = m.sc2; }
  // ---- list 1: lo, paired; x1 = count of a paired entry (for max1), else 0
  uint32_t my_end = n1;
  for (uint32_t i = g.t; i < n1; i += G) {
    const uint64_t a = p1[i];
    uint32_t lo = 0, hi = n2;  // first j with p2[j] + dist >= a
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (a > p2[mid] + dist) lo = mid + 1; else hi = mid;
    }
    m.lo1[i] = (uint16_t)lo;
    const bool paired = lo < n2 && !(p2[lo] > a + dist);
    m.k1[i] = paired ? 1 : 0;
    m.x1[i] = paired ? c1[i] : 0;
    if (lo == n2 && i < my_end) my_end = i;
  }
  const uint32_t i_end = (uint32_t)g.min64((uint64_t)my_end);
  // ---- list 2: covered; x2 = count of a covered entry (for max2), else 0
  for (uint32_t j = g.t; j < n2; j += G) {
    const uint64_t b = p2[j];
    uint32_t lo = 0, hi = n1;  // first i with p1[i] + dist >= b
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (b > p1[mid] + dist) lo = mid + 1; else hi = mid;
    }
    const bool covered = lo < n1 && !(p1[lo] > b + dist);
    m.k2[j] = covered ? 1 : 0;
    m.x2[j] = covered ? c2[j] : 0;
  }
  g.sync();
  (void)cm_coop_array_scan_max(g, m.x1, n1);  // x1[i] = largest count of a paired entry before i
  (void)cm_coop_array_scan_max(g, m.x2, n2);
  // ---- the unpaired / uncovered entries that meet the sequence and count conditions
  for (uint32_t i = g.t; i < n1; i += G) {
    uint32_t cond = 0;
    if (!m.k1[i] && i < i_end) {
      const uint32_t mx = m.x1[i] > 6 ? m.x1[i] : 6;
      cond = ((p1[i] >> 32) == (p2[m.lo1[i]] >> 32) && (uint32_t)c1[i] >= mx) ? 1u : 0u;
    }
    m.of1[i] = (uint16_t)cond;
  }
  for (uint32_t j = g.t; j < n2; j += G) {
    uint32_t cond = 0;
    if (!m.k2[j]) {
      const uint64_t b = p2[j];
      uint32_t lo = 0, hi = n1;  // sk(j): first i with p1[i] > b + dist
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p1[mid] > b + dist) hi = mid; else lo = mid + 1;
      }
      if (lo < n1) {
        const uint32_t mx = m.x2[j] > 6 ? m.x2[j] : 6;
        cond = ((p1[lo] >> 32) == (b >> 32) && (uint32_t)c2[j] >= mx) ? 1u : 0u;
      }
    }
    m.of2[j] = (uint16_t)cond;
  }
  g.sync();
  // keep flags: paired / covered, or among the first five that meet the conditions (of = rank after the scan)
  for (uint32_t i = g.t; i < n1; i += G) m.x1[i] = (uint8_t)m.of1[i];
  for (uint32_t j = g.t; j < n2; j += G) m.x2[j] = (uint8_t)m.of2[j];
  g.sync();
  (void)cm_coop_array_scan_add(g, m.of1, n1);
  (void)cm_coop_array_scan_add(g, m.of2, n2);
  for (uint32_t i = g.t; i < n1; i += G) {
    const bool keep = (m.k1[i] && i < i_end) || (m.x1[i] && m.of1[i] < 5);
    m.k1[i] = keep ? 1 : 0;
  }
  for (uint32_t j = g.t; j < n2; j += G) {
    const bool keep = m.k2[j] || (m.x2[j] && m.of2[j] < 5);
    m.k2[j] = keep ? 1 : 0;
  }
  g.sync();
  for (uint32_t i = g.t; i < n1; i += G) m.of1[i] = m.k1[i];
  for (uint32_t j = g.t; j < n2; j += G) m.of2[j] = m.k2[j];
  g.sync();
  const uint32_t t1 = cm_coop_array_scan_add(g, m.of1, n1);
  const uint32_t t2 = cm_coop_array_scan_add(g, m.of2, n2);
  for (uint32_t i = g.t; i < n1; i += G)
    if (m.k1[i]) { f1[m.of1[i]] = p1[i]; fc1[m.of1[i]] = c1[i]; }
  for (uint32_t j = g.t; j < n2; j += G)
    if (m.k2[j]) { f2[m.of2[j]] = p2[j]; fc2[m.of2[j]] = c2[j]; }
  *nf1 = t1;
  *nf2 = t2;
}

// S4c for one pair with long candidate lists: the two directions of the paired-end filter by the group, the pair's fate,
// the re-ranking (cm_s4c_filter + cm_s4c_post; cm_s4c_pre ran in the per-pair kernel and asked for the filter).
template <bool STAGED, class GT>
CM_HD void cm_coop_s4c(const CmDev &d, uint32_t pair, GT &g, const CmCoopPairMem &m) {
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  const uint32_t G = (uint32_t)GT::G;
  // (round 6: the reads' offsets and list lengths are requested once, here -- taken where they are used, the second direction's came after
  //  the first one's barriers: one more trip to global memory per pair)
  const uint32_t mo1 = d.m_off[r1], mo2 = d.m_off[r2], sh1 = d.ncp[r1] + d.resc_p[r1], sh2 = d.ncp[r2] + d.resc_p[r2];
  const uint32_t mcp1 = d.mcp[r1], mcn1 = d.mcn[r1], mcp2 = d.mcp[r2], mcn2 = d.mcn[r2];
  if (mcp1 > m.P || mcn1 > m.P || mcp2 > m.P || mcn2 > m.P) {  // longer than the work arrays: one lane
    if (g.t == 0) { cm_s4c_filter(d, pair); cm_s4c_post(d, pair); }
    return;
  }
  uint32_t a, b, c, e2;
  cm_coop_reduce_dir<STAGED>(g, m, (uint32_t)d.p.max_insert, d.mbuf + mo1, d.mcnt + mo1, mcp1, d.mbuf + mo2 + sh2, d.mcnt + mo2 + sh2, mcn2,
                     d.fbuf + mo1, d.fcnt + mo1, &a, d.fbuf + mo2 + sh2, d.fcnt + mo2 + sh2, &b);
  g.sync();
  cm_coop_reduce_dir<STAGED>(g, m, (uint32_t)d.p.max_insert, d.mbuf + mo1 + sh1, d.mcnt + mo1 + sh1, mcn1, d.mbuf + mo2, d.mcnt + mo2, mcp2,
                     d.fbuf + mo1 + sh1, d.fcnt + mo1 + sh1, &c, d.fbuf + mo2, d.fcnt + mo2, &e2);
  g.sync();
  const bool alive = a + c > 0 && b + e2 > 0;
  if (g.t == 0) {
    d.fcp[r1] = a; d.fcn[r2] = b; d.fcn[r1] = c; d.fcp[r2] = e2;
    d.alive[pair] = alive ? 1 : 0;
  }
  if (alive && d.rid_rank) {  // cm_rerank of both reads
    uint64_t *l[4] = {d.fbuf + mo1, d.fbuf + mo1 + sh1, d.fbuf + mo2, d.fbuf + mo2 + sh2};
    const uint32_t n[4] = {a, c, e2, b};
    for (int q = 0; q < 4; ++q)
      for (uint32_t i = g.t; i < n[q]; i += G) l[q][i] = (l[q][i] & 0xffffffffull) | ((uint64_t)d.rid_rank[(uint32_t)(l[q][i] >> 32)] << 32);
  }
}

template <class GT>
CM_HD uint32_t cm_coop_array_scan_max16(GT &g, uint16_t *a, uint32_t n) {  // exclusive prefix maximum in place; returns the maximum
  const uint32_t VT = cm_coop_chunk(n, (uint32_t)GT::G);
  const uint32_t c0 = cm_min_u32(n, g.t * VT), c1 = cm_min_u32(n, c0 + VT);
  uint32_t mx = 0;
  for (uint32_t i = c0; i < c1; ++i) mx = a[i] > mx ? a[i] : mx;
  uint32_t total;
  uint32_t run = g.scanmax(mx, &total);
  for (uint32_t i = c0; i < c1; ++i) { const uint32_t x = a[i]; a[i] = (uint16_t)run; run = x > run ? x : run; }
  g.sync();
  return total;
}

// the two smallest distinct values of a multiset with their counts, as cm_update_best / the pairing loop keep them:
// (lo, n_lo, hi, n_hi), an unseen slot is (none, 0).  cm_two_add puts cnt copies of v in.
struct CmTwo { int lo, n_lo, hi, n_hi; };
CM_HD void cm_two_add(CmTwo &s, int v, int cnt) {
  if (cnt <= 0) return;
  if (v < s.lo) { s.hi = s.lo; s.n_hi = s.n_lo; s.lo = v; s.n_lo = cnt; }
  else if (v == s.lo) s.n_lo += cnt;
  else if (v == s.hi) s.n_hi += cnt;
  else if (v < s.hi) { s.hi = v; s.n_hi = cnt; }
}
// the lanes' sets merged; `none` = the value of an unseen slot (larger than every real value).  Every lane gets the result.
template <class GT>
CM_HD CmTwo cm_coop_two_merge(GT &g, const CmTwo &mine, int none) {
  const int BIAS = 1 << 20;  // values may be negative (split mode keeps -length); order is kept under the bias
  CmTwo out;
  out.lo = (int)g.min64((uint64_t)(mine.lo + BIAS)) - BIAS;
  out.n_lo = (int)g.sum(mine.lo == out.lo ? (uint32_t)mine.n_lo : 0u);
  const int c2 = mine.lo > out.lo ? mine.lo : mine.hi;
  out.hi = (int)g.min64((uint64_t)(c2 + BIAS)) - BIAS;
  out.n_hi = (int)g.sum((mine.lo == out.hi ? (uint32_t)mine.n_lo : 0u) + (mine.hi == out.hi ? (uint32_t)mine.n_hi : 0u));
  if (out.lo == none) { out.n_lo = 0; out.hi = none; out.n_hi = 0; }
  if (out.hi == none) out.n_hi = 0;
  return out;
}

// ---------------------------------------------------------------------------------------
// S5c for one read with many candidates (cm_s5c_finalize's results): the acceptance loop of
// DraftMappingGenerator::GenerateDraftMappingsOnOneStrand over the precomputed alignments (cm_draft_strand with pre_err).
// With `lanes` > 0 the reference verifies the valid candidates in groups of `lanes` and, after a group, lowers its count
// threshold to the count of the group's last rejected candidate; it stops at the first candidate whose count is below the
// threshold.  Candidate ci (v valid candidates before it) therefore sees the threshold of the LAST rejected candidate among
// the first lanes * floor(v / lanes) valid ones -- a prefix maximum over the valid candidates' ranks -- and the loop stops
// at B = the first ci whose count is below its threshold; every valid candidate before B is verified (the unfinished group
// at the end too).  Accepted = verified with at most e errors, in list order; best / second best = the two smallest error
// counts with their multiplicities.
// Work arrays (shared, P entries each): of, fr, pm (u16), vf (u8).
// ---------------------------------------------------------------------------------------
struct CmCoopVerMem {
  uint16_t *of, *fr, *pm;  // P + 1 each
  int16_t *pe;             // P: the candidates' error counts, staged (round 6: read where they lie in four of the function's passes, the
  uint8_t *vf;             // P   counts in a fifth -- every pass then began with a trip to global memory)
  uint8_t *sc;             // P: the candidates' counts, staged
  uint32_t P;
};
CM_HD size_t cm_coop_ver_mem_bytes(uint32_t P) { return (size_t)(P + 1) * 6 + (size_t)P * 4 + 32; }
CM_HD CmCoopVerMem cm_coop_ver_mem_at(uint8_t *base, uint32_t P) {
  CmCoopVerMem m;
  m.P = P;
  m.of = reinterpret_cast<uint16_t *>(base);
  m.fr = m.of + P + 1;
  m.pm = m.fr + P + 1;
  m.pe = reinterpret_cast<int16_t *>(m.pm + P + 1);
  m.vf = reinterpret_cast<uint8_t *>(m.pe + P);
  m.sc = m.vf + P;
  return m;
}
template <class GT>
CM_HD uint32_t cm_coop_draft_strand(const CmDev &d, GT &g, const CmCoopVerMem &m, uint32_t L, int strand, const uint64_t *cp, const uint8_t *cc,
                                    uint32_t nc, CmTwo &best, uint64_t *dp, int16_t *de, const int16_t *pre_err, const int16_t *pre_end) {
  const uint32_t G = (uint32_t)GT::G;
  const int e = d.p.e;
  const uint32_t lanes = (uint32_t)d.p.lanes;
  if (nc == 0) return 0;
  for (uint32_t ci = g.t; ci < nc; ci += G) {
    const uint64_t cpos = cp[ci];
    m.pe[ci] = pre_err[ci];  // (one trip to global memory for everything the passes below look at)
    m.sc[ci] = cc[ci];
    const uint32_t rid = (uint32_t)(cpos >> 32);
    uint32_t position = (uint32_t)cpos;
    if (strand == 1) position = position - L + 1;
    const bool valid = cm_valid_candidate(d, rid, position, L);
    m.vf[ci] = valid ? 1 : 0;
    m.of[ci] = valid ? 1 : 0;
  }
  g.sync();
  pre_err = m.pe;
  cc = m.sc;
  const uint32_t nvalid = cm_coop_array_scan_add(g, m.of, nc);  // of[ci] = valid candidates before ci
  uint32_t B = nc;
  if (lanes != 0 && nc >= lanes) {
    for (uint32_t ci = g.t; ci < nc; ci += G)
      if (m.vf[ci]) { const uint32_t k = m.of[ci]; m.fr[k] = (uint16_t)ci; m.pm[k] = pre_err[ci] > e ? (uint16_t)(k + 1) : 0; }
    if (g.t == 0) m.pm[nvalid] = 0;
    g.sync();
    (void)cm_coop_array_scan_max16(g, m.pm, nvalid + 1);  // pm[k] = 1 + rank of the last rejected candidate among ranks < k (0: none)
    uint32_t mine = nc;
    for (uint32_t ci = g.t; ci < nc; ci += G) {
      const uint32_t done = m.of[ci] / lanes * lanes;  // valid candidates in finished groups when the loop looks at ci
      const uint32_t lf = m.pm[done];
      const uint32_t thr = lf ? cc[m.fr[lf - 1]] : 0u;
      if ((uint32_t)cc[ci] < thr && ci < mine) mine = ci;
    }
    B = (uint32_t)g.min64((uint64_t)mine);
  }
  g.sync();  // of is rewritten
  CmTwo mineb = {e + 1, 0, e + 1, 0};
  for (uint32_t ci = g.t; ci < nc; ci += G) {
    const bool acc = m.vf[ci] && ci < B && pre_err[ci] <= e;
    m.vf[ci] = acc ? 1 : 0;
    if (acc) cm_two_add(mineb, (int)pre_err[ci], 1);
  }
  g.sync();
  for (uint32_t ci = g.t; ci < nc; ci += G) m.of[ci] = m.vf[ci];
  g.sync();
  const uint32_t nd = cm_coop_array_scan_add(g, m.of, nc);
  for (uint32_t ci = g.t; ci < nc; ci += G) {
    if (!m.vf[ci]) continue;
    const uint64_t cpos = cp[ci];
    const int end_pos = pre_end[ci];
    dp[m.of[ci]] = strand == 0 ? cpos - (uint64_t)e + (uint64_t)(int64_t)end_pos : cpos - L + 1 - (uint64_t)e + (uint64_t)(int64_t)end_pos;
    de[m.of[ci]] = pre_err[ci];
  }
  const CmTwo all = cm_coop_two_merge(g, mineb, e + 1);
  cm_two_add(best, all.lo, all.n_lo);
  cm_two_add(best, all.hi, all.n_hi);
  g.sync();
  return nd;
}
// ---------------------------------------------------------------------------------------
// MappingMetadata::SortCandidates for one long list by the group: Candidate::operator< is (count descending, position
// ascending) and the list arrives in position order, so the sort is a STABLE partition by count -- a counting sort: every lane
// counts the counts of its contiguous chunk (hist: G x nb_cap 16-bit bins of shared memory), the bins are scanned over the
// lanes from the largest count down, the chunk is scattered to (sp, sc) and copied back.  A list that is not in position
// order (--chr-order re-ranks the sequence ids after the filter) or has a count >= nb_cap is sorted by the group's lane 0.
// ---------------------------------------------------------------------------------------
template <class GT>
CM_HD void cm_coop_sort_cand(GT &g, uint64_t *p, uint8_t *c, uint32_t n, uint64_t *sp, uint8_t *sc, uint16_t *hist, uint32_t nb_cap,
                             uint64_t *lp = nullptr, uint8_t *lc = nullptr, uint32_t lcap = 0, unsigned long long *prof = nullptr) {
  const uint32_t G = (uint32_t)GT::G;
  if (n < 2) return;
  // lp / lc (shared memory, lcap entries): a list that fits is staged there once and scattered from there straight to its final
  // places -- one read and one write of the list in global memory instead of three and two (k_s5_sort_coop was four rounds of
  // global latency per list); longer lists go through sp / sc (global scratch) as before
  const bool staged = lp != nullptr && n <= lcap;
  uint32_t bad = 0;
  uint64_t mx = 0;
  if (staged) {
    for (uint32_t i = g.t; i < n; i += G) { lp[i] = p[i]; const uint8_t ci = c[i]; lc[i] = ci; mx = ci > mx ? ci : mx; }
    g.sync();
    for (uint32_t i = g.t; i < n; i += G) if (i > 0 && lp[i] < lp[i - 1]) bad = 1;
  } else {
    for (uint32_t i = g.t; i < n; i += G) {
      if (i > 0 && p[i] < p[i - 1]) bad = 1;
      mx = c[i] > mx ? c[i] : mx;
    }
  }
  bad = g.sum(bad);
  mx = g.max64(mx);
  const bool wave_bins = G == (uint32_t)GT::W && mx < 4 * G;  // (the bins live in the wave part's lanes: nb_cap limits the histogram form only)
  if (bad || (!wave_bins && mx >= nb_cap) || n > 0xffffu) {
    CM_PROF_PTR_COUNT(prof, g, bad ? 7 : 47, 1);
    if (g.t == 0) cm_sort_cand(p, c, n);
    g.sync();
    return;
  }
  const uint32_t nb = (uint32_t)mx + 1;
  if (G == (uint32_t)GT::W && nb <= 4 * G) {
    // One wave part per list (round 5): lane b keeps bin b -- first the number of candidates with count b, then where the next of them
    // goes.  The list is taken 64 candidates at a time; the distinct counts among them (a handful) are handled one after the other: a
    // ballot of the lanes that hold the count, its population to the bin's lane / the bin's value from that lane, the lane's rank
    // among them.  No histogram per lane: the kernel's shared memory was 50 KB per four waves (12 waves per CU), and its time is
    // the lists' global-memory latency.
    // Round 6: FOUR bins per lane (count v: lane v % G, slot v / G), i.e. every value a count can take with a wave of 64 -- one list in
    // ten of the repeat-rich genome (profile 2) holds a candidate with a count of 64 or more (the modal diagonal of a cluster in a
    // satellite array collects the hits of every copy of the unit), fell through to lane 0's heap sort in global memory and made
    // the sorting waves' mean 156 us per read: 10 % of the lists were 90 % of k_s5_sort_coop.
    const uint8_t *cs = staged ? lc : c;
    const uint32_t NS = (nb + G - 1) / G;  // slots in use (uniform)
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (uint32_t base = 0; base < n; base += G) {
      const uint32_t i = base + g.t;
      const bool in = i < n;
      const uint32_t ci = in ? cs[i] : 0u;
      unsigned long long rem = g.ballot(in);
      while (rem) {
        const uint32_t v = g.bcast(ci, (uint32_t)__builtin_ctzll(rem));
        const unsigned long long mk = g.ballot(in && ci == v);
        const uint32_t pc = (uint32_t)__builtin_popcountll(mk), sl = v / G;
        if (g.t == v % G) { b0 += sl == 0 ? pc : 0u; b1 += sl == 1 ? pc : 0u; b2 += sl == 2 ? pc : 0u; b3 += sl == 3 ? pc : 0u; }
        rem &= ~mk;
      }
    }
    {  // the largest count first: a bin starts behind all candidates with a larger count -- slot by slot from the top
      uint32_t above = 0, tot, below;
      if (NS > 3) { below = g.scan(b3, &tot); b3 = above + (tot - below - b3); above += tot; }
      if (NS > 2) { below = g.scan(b2, &tot); b2 = above + (tot - below - b2); above += tot; }
      if (NS > 1) { below = g.scan(b1, &tot); b1 = above + (tot - below - b1); above += tot; }
      below = g.scan(b0, &tot); b0 = above + (tot - below - b0);
    }
    uint64_t *const dpp = staged ? p : sp;
    uint8_t *const dcc = staged ? c : sc;
    const uint64_t *const spp = staged ? lp : p;
    for (uint32_t base = 0; base < n; base += G) {
      const uint32_t i = base + g.t;
      const bool in = i < n;
      const uint32_t ci = in ? cs[i] : 0u;
      const uint64_t pi = in ? spp[i] : 0ull;
      unsigned long long rem = g.ballot(in);
      uint32_t dst = 0;
      while (rem) {
        const uint32_t v = g.bcast(ci, (uint32_t)__builtin_ctzll(rem));
        const unsigned long long mk = g.ballot(in && ci == v);
        const uint32_t pc = (uint32_t)__builtin_popcountll(mk), sl = v / G;  // (sl: the same for all lanes)
        const uint32_t at = g.bcast(sl == 0 ? b0 : sl == 1 ? b1 : sl == 2 ? b2 : b3, v % G);
        if (in && ci == v) dst = at + (uint32_t)__builtin_popcountll(mk & ((1ull << (g.t % (uint32_t)GT::W)) - 1ull));
        if (g.t == v % G) { b0 += sl == 0 ? pc : 0u; b1 += sl == 1 ? pc : 0u; b2 += sl == 2 ? pc : 0u; b3 += sl == 3 ? pc : 0u; }
        rem &= ~mk;
      }
      if (in) { dpp[dst] = pi; dcc[dst] = (uint8_t)ci; }
    }
    g.sync();
    if (!staged) {
      for (uint32_t i = g.t; i < n; i += G) { p[i] = sp[i]; c[i] = sc[i]; }
      g.sync();
    }
    return;
  }
  // (larger groups) bin b of lane t at hist[b * G + t]: the lanes of a wave touch consecutive 16-bit words
  uint16_t *mine = hist + g.t;
  for (uint32_t b = 0; b < nb; ++b) mine[(size_t)b * G] = 0;
  const uint32_t VT = cm_coop_chunk(n, G);
  const uint32_t c0 = cm_min_u32(n, g.t * VT), c1 = cm_min_u32(n, c0 + VT);
  const uint8_t *cs = staged ? lc : c;
  for (uint32_t i = c0; i < c1; ++i) mine[(size_t)cs[i] * G] += 1;
  uint32_t base = 0;
  for (uint32_t b = nb; b-- > 0;) {  // the largest count first
    uint32_t tot;
    const uint32_t off = g.scan(mine[(size_t)b * G], &tot);
    mine[(size_t)b * G] = (uint16_t)(base + off);
    base += tot;
  }
  if (staged) {
    for (uint32_t i = c0; i < c1; ++i) {
      const uint8_t ci = lc[i];
      const uint32_t dst = mine[(size_t)ci * G]++;
      p[dst] = lp[i];
      c[dst] = ci;
    }
    g.sync();  // (the staging arrays serve the next list)
    return;
  }
  for (uint32_t i = c0; i < c1; ++i) {
    const uint8_t ci = c[i];
    const uint32_t dst = mine[(size_t)ci * G]++;
    sp[dst] = p[i];
    sc[dst] = ci;
  }
  g.sync();
  for (uint32_t i = g.t; i < n; i += G) { p[i] = sp[i]; c[i] = sc[i]; }
  g.sync();
}

// ---------------------------------------------------------------------------------------
// SortMappingsByPositions (mapping_metadata.h:70-78) for the draft mappings a read's acceptance loop just wrote, by the group.
// They stand in candidate order -- count descending, position ascending -- i.e. a few ascending runs (one per count value):
// natural runs + merge sort of the packed keys (position << 6 | errors; the order among equal positions is free).  Lists of up
// to m.P entries in shared memory, longer ones in place with `scratch` (global memory, nd entries) as the second buffer.  Left
// unsorted -- the pairing stage then sorts them as before -- when the keys cannot be packed (more than 2^25 sequences, an error
// threshold above 62) or there are more runs than m.RB.
// ---------------------------------------------------------------------------------------
struct CmCoopSortMem { uint64_t *A, *B; uint32_t *rb, *rb2; uint32_t P, RB; };
CM_HD size_t cm_coop_sort_mem_bytes(uint32_t P, uint32_t RB) { return (size_t)P * 16 + (size_t)(RB + 1) * 8 + 32; }
CM_HD CmCoopSortMem cm_coop_sort_mem_at(uint8_t *base, uint32_t P, uint32_t RB) {
  CmCoopSortMem m;
  m.P = P; m.RB = RB;
  m.A = reinterpret_cast<uint64_t *>(base);
  m.B = m.A + P;
  m.rb = reinterpret_cast<uint32_t *>(m.B + P);
  m.rb2 = m.rb + RB + 1;
  return m;
}
template <bool IN_LDS, class GT>
CM_HD void cm_coop_sort_draft_in(const CmDev &d, GT &g, const CmCoopSortMem &m, uint64_t *dp, int16_t *de, uint32_t nd, uint64_t *scratch) {
  const uint32_t G = (uint32_t)GT::G;
  uint64_t *A = IN_LDS ? m.A : dp, *B = IN_LDS ? m.B : scratch;  // (compile-time choice: see cm_coop_s3b)
  for (uint32_t i = g.t; i < nd; i += G) A[i] = (dp[i] << 6) | (uint64_t)(uint16_t)de[i];
  g.sync();
  const uint32_t nr = cm_coop_natural_runs(g, A, nd, m.rb, m.RB);
  const uint64_t *S = A;
  if (nr > 1) S = cm_coop_merge_runs(g, A, B, m.rb, m.rb2, nr, nd);
  // unpack (nr == 0: too many runs -- the keys go back as they came)
  if (S == dp) {  // in place: every lane its own entries
    for (uint32_t i = g.t; i < nd; i += G) { const uint64_t k = dp[i]; dp[i] = k >> 6; de[i] = (int16_t)(k & 63u); }
  } else {
    for (uint32_t i = g.t; i < nd; i += G) { const uint64_t k = S[i]; dp[i] = k >> 6; de[i] = (int16_t)(k & 63u); }
  }
  g.sync();
}
template <class GT>
CM_HD void cm_coop_sort_draft(const CmDev &d, GT &g, const CmCoopSortMem &m, uint64_t *dp, int16_t *de, uint32_t nd, uint64_t *scratch) {
  if (nd < 2 || d.n_seq > (1u << 25) || d.p.e > 62) return;
  if (nd <= m.P) cm_coop_sort_draft_in<true>(d, g, m, dp, de, nd, scratch); else cm_coop_sort_draft_in<false>(d, g, m, dp, de, nd, scratch);
}

// The candidate lists of a read cm_s5a_prepare left to the groups, sorted (scratch: the read's draft-mapping arrays, written by
// S5c only).  The alignments then run in the per-candidate kernel like everybody's.
template <class GT>
CM_HD void cm_coop_s5_sort(const CmDev &d, uint32_t r, GT &g, uint16_t *hist, uint32_t nb_cap, uint64_t *lp = nullptr, uint8_t *lc = nullptr,
                           uint32_t lcap = 0) {
  const uint32_t op = d.m_off[r], on = op + d.ncp[r] + d.resc_p[r];
  const uint32_t ncp_ = d.fcp[r], ncn_ = d.fcn[r];  // (both lengths before the first list's barriers)
  cm_coop_sort_cand(g, d.fbuf + op, d.fcnt + op, ncp_, d.dpos + op, reinterpret_cast<uint8_t *>(d.derr + op), hist, nb_cap, lp, lc, lcap, d.prof);
  cm_coop_sort_cand(g, d.fbuf + on, d.fcnt + on, ncn_, d.dpos + on, reinterpret_cast<uint8_t *>(d.derr + on), hist, nb_cap, lp, lc, lcap, d.prof);
}
// S5c for such a read (its alignments are in v_err / v_end)
// sm: work area of the draft-mapping sort that follows the acceptance loop (it may overlay m: the loop's arrays are dead by then)
template <class GT>
CM_HD void cm_coop_s5c(const CmDev &d, uint32_t r, GT &g, const CmCoopVerMem &m, const CmCoopSortMem &sm) {
  const uint32_t op = d.m_off[r], on = d.m_off[r] + d.ncp[r] + d.resc_p[r];
  uint32_t ndp, ndn;
  const uint32_t nc_p = d.fcp[r], nc_n = d.fcn[r];
  if (nc_p > m.P || nc_n > m.P) {  // longer than the work arrays: the acceptance loop by one lane -- but the draft mappings it
    // leaves are sorted by the group like everybody's (round 4: they used to stay in candidate order, and the pairing stage's lane 0
    // then heap-sorted lists of thousands of entries in global memory: 38 + 18 ms per batch of the mosaic genome for ~200 reads)
    if (g.t == 0) cm_s5c_accept(d, r);
    g.sync();
    ndp = cm_coop_bcast0(g, g.t == 0 ? d.ndp[r] : 0u);
    ndn = cm_coop_bcast0(g, g.t == 0 ? d.ndn[r] : 0u);
  } else {
    CmTwo best = {d.min_err[r], d.n_best[r], d.second_err[r], d.n_second[r]};
    const uint32_t L = d.rlen[r];
    ndp = cm_coop_draft_strand(d, g, m, L, 0, d.fbuf + op, d.fcnt + op, nc_p, best, d.dpos + op, d.derr + op, d.v_err + op, d.v_end + op);
    ndn = cm_coop_draft_strand(d, g, m, L, 1, d.fbuf + on, d.fcnt + on, nc_n, best, d.dpos + on, d.derr + on, d.v_err + on, d.v_end + on);
    if (g.t == 0) {
      d.ndp[r] = ndp; d.ndn[r] = ndn;
      d.min_err[r] = best.lo; d.second_err[r] = best.hi; d.n_best[r] = best.n_lo; d.n_second[r] = best.n_hi;
    }
  }
  if (!d.p.single) {  // the pairing stage wants them by position (single-end keeps the emission order)
    g.sync();
    cm_coop_sort_draft(d, g, sm, d.dpos + op, d.derr + op, ndp, d.fbuf + op);  // the candidate lists are dead: scratch
    cm_coop_sort_draft(d, g, sm, d.dpos + on, d.derr + on, ndn, d.fbuf + on);
  }
}

// ---------------------------------------------------------------------------------------
// S6a for one pair with many draft mappings (cm_s6a_pair's paired-end, non-split part): the two sweeps of
// GenerateBestMappingsForPairedEndReadOnOneDirection (mapping_generator.h:347-484, cm_pair_dir).  For the mapping i1 of the first
// list the sweep pairs it with the range [lo(i1), hi(i1)) of the second (sorted) list -- lo: the first entry not too far below,
// hi: the first entry too far above; both monotone in i1, so the sequential pointer stands at lo(i1) -- and keeps the two
// smallest error sums with their multiplicities and the first pairing (direction, i1, i2 order) that reaches the minimum.
// A lane takes the mappings i1 = t, t + G, ...; the lanes' results are merged (the first minimal pairing: the smallest packed
// (sum, direction, i1, i2) key).
// ---------------------------------------------------------------------------------------
// STAGED (compile time, for the reason given at cm_coop_s3b): the second list is copied to the group's shared work arrays
// (sp / se, nb entries) first -- its binary searches and partner walks are chains of dependent loads, ~12 + range per first-list
// entry, and at global-memory latency they were ALL of k_s6a_coop (38 ms per 4 M pairs of the mosaic genome at 6 % VALU
// activity, profiles/r04a_harsh_*).  The first list is read once, coalesced, from where it is.
template <bool STAGED, class GT>
CM_HD void cm_coop_pair_dir(const CmDev &d, GT &g, int dir, const uint64_t *ap, const int16_t *ae, uint32_t na, const uint64_t *gbp, const int16_t *gbe,
                            uint32_t nb, uint32_t len1, uint32_t len2, CmTwo &mine, uint64_t &first_key, uint64_t *sp = nullptr, int16_t *se = nullptr) {
  const uint64_t I = (uint64_t)(int64_t)d.p.max_insert;
  const uint64_t mo = (uint32_t)d.p.min_read_len;
  const uint64_t X = dir == 1 ? I - len2 : (uint64_t)len1 - mo;  // an entry p2 is too far below p1 when p1 > p2 + X
  const uint64_t Y = dir == 0 ? I - len1 : (uint64_t)len2 - mo;  // ... too far above when p2 > p1 + Y
  if (na == 0 || nb == 0) return;  // (uniform)
  if (STAGED) {
    g.sync();  // the previous user of the work arrays is done
    for (uint32_t j = g.t; j < nb; j += (uint32_t)GT::G) { sp[j] = gbp[j]; se[j] = gbe[j]; }
    g.sync();
  }
  const uint64_t *bp = STAGED ? sp : gbp;
  const int16_t *be = STAGED ? se : gbe;
  for (uint32_t i1 = g.t; i1 < na; i1 += (uint32_t)GT::G) {
    const uint64_t p1 = ap[i1];
    const int e1 = (int)ae[i1];
    uint32_t lo = 0, hi = nb;  // first i2 with !(p1 > bp[i2] + X)
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (p1 > bp[mid] + X) lo = mid + 1; else hi = mid;
    }
    for (uint32_t cur = lo; cur < nb && bp[cur] <= p1 + Y; ++cur) {
      const int s = e1 + (int)be[cur];
      cm_two_add(mine, s, 1);
      const uint64_t key = ((uint64_t)(uint32_t)(s + 1024) << 49) | ((uint64_t)(uint32_t)dir << 48) | ((uint64_t)i1 << 24) | (uint64_t)cur;
      if (key < first_key) first_key = key;
    }
  }
}
// true when p[0..n) is ascending (every lane gets the answer)
template <class GT>
CM_HD bool cm_coop_is_sorted(GT &g, const uint64_t *p, uint32_t n) {
  uint32_t bad = 0;
  for (uint32_t i = g.t + 1; i < n; i += (uint32_t)GT::G) bad += p[i] < p[i - 1] ? 1u : 0u;
  return g.sum(bad) == 0;
}
// shared work arrays of a group for the pairing stages: the second list of one direction (positions, error counts), P entries
struct CmCoopPeMem { uint64_t *sp; int16_t *se; uint32_t P; };
CM_HD size_t cm_coop_pe_mem_bytes(uint32_t P) { return (size_t)P * 10 + 16; }
CM_HD CmCoopPeMem cm_coop_pe_mem_at(uint8_t *base, uint32_t P) {
  CmCoopPeMem m;
  m.P = P;
  m.sp = reinterpret_cast<uint64_t *>(base);
  m.se = reinterpret_cast<int16_t *>(m.sp + P);
  return m;
}
template <bool SAM, class GT>
CM_HD void cm_coop_s6a(const CmDev &d, uint32_t pair, GT &g, const CmCoopPeMem &m) {
  // cm_s6a_pair's prologue ran in the per-pair kernel (record slots cleared, pe_nbest = 0, both reads have draft mappings)
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  // (round 6: everything the function needs of the two reads is requested at once, and the four "already sorted?" passes read their lists
  //  together -- taken one after the other, behind one another's barriers, they were a dozen dependent trips to global memory per pair)
  const uint32_t mo1 = d.m_off[r1], mo2 = d.m_off[r2], sh1 = d.ncp[r1] + d.resc_p[r1], sh2 = d.ncp[r2] + d.resc_p[r2];
  const uint32_t nd[4] = {d.ndp[r1], d.ndn[r1], d.ndp[r2], d.ndn[r2]};  // (read, strand): (1, +), (1, -), (2, +), (2, -)
  const uint32_t len1 = d.rlen[r1], len2 = d.rlen[r2];
  uint64_t *const lp[4] = {d.dpos + mo1, d.dpos + mo1 + sh1, d.dpos + mo2, d.dpos + mo2 + sh2};
  int16_t *const le[4] = {d.derr + mo1, d.derr + mo1 + sh1, d.derr + mo2, d.derr + mo2 + sh2};
  uint32_t bad[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    for (uint32_t i = g.t + 1; i < nd[q]; i += (uint32_t)GT::G) bad[q] += lp[q][i] < lp[q][i - 1] ? 1u : 0u;
#pragma unroll
  for (int q = 0; q < 4; ++q) bad[q] = g.sum(bad[q]);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (bad[q]) {  // lists beyond the sorting waves' size
      if (g.t == 0) cm_sort_draft(lp[q], le[q], nd[q]);
      g.sync();
    }
  const int none = 2 * d.p.e + 1;
  CmTwo mine = {none, 0, none, 0};
  uint64_t first_key = ~0ull;
  // a second list that fits the work arrays is staged there (every lane takes the same branch: the lengths are the pair's)
  if (nd[3] <= m.P)
    cm_coop_pair_dir<true>(d, g, 0, lp[0], le[0], nd[0], lp[3], le[3], nd[3], len1, len2, mine, first_key, m.sp, m.se);
  else
    cm_coop_pair_dir<false>(d, g, 0, lp[0], le[0], nd[0], lp[3], le[3], nd[3], len1, len2, mine, first_key);
  if (nd[2] <= m.P)
    cm_coop_pair_dir<true>(d, g, 1, lp[1], le[1], nd[1], lp[2], le[2], nd[2], len1, len2, mine, first_key, m.sp, m.se);
  else
    cm_coop_pair_dir<false>(d, g, 1, lp[1], le[1], nd[1], lp[2], le[2], nd[2], len1, len2, mine, first_key);
  const CmTwo all = cm_coop_two_merge(g, mine, none);
  const uint64_t fk = g.min64(first_key);
  CmPe pe;
  pe.min_sum = all.lo; pe.second_sum = all.hi; pe.n_best = all.n_lo; pe.n_second = all.n_hi;
  pe.f_dir = 0; pe.f_i1 = 0; pe.f_i2 = 0;
  if (fk != ~0ull) { pe.f_dir = (uint32_t)(fk >> 48) & 1u; pe.f_i1 = (uint32_t)(fk >> 24) & 0xffffffu; pe.f_i2 = (uint32_t)fk & 0xffffffu; }
  if (g.t == 0) {
    d.pe_min[pair] = pe.min_sum; d.pe_second[pair] = pe.second_sum;
    d.pe_nbest[pair] = pe.n_best; d.pe_nsecond[pair] = pe.n_second;
    d.pe_first[pair] = pe.f_dir; d.pe_i1[pair] = pe.f_i1; d.pe_i2[pair] = pe.f_i2;
    if (pe.n_best == 1) cm_emit_record<SAM>(d, pair, pe);
  }
}

// ---------------------------------------------------------------------------------------
// S6c for a multi-mapped pair with long draft lists: the pairing the sampler chose (the want-th pairing with the minimal sum, in
// the sweep's order: direction 0 before 1, first-list entries ascending, partners ascending) is found by the group instead of one
// lane repeating both sweeps (cm_s6c_multi / cm_pair_dir with want >= 0): every lane counts the minimal-sum partners of its
// first-list entry (partner range by binary search, as cm_coop_pair_dir), a scan places the target in one lane's range, that lane
// walks to it.  *seen: minimal-sum pairings before this direction (in, uniform) / including it when not found (out).
// ---------------------------------------------------------------------------------------
template <bool STAGED, class GT>
CM_HD bool cm_coop_pair_find(const CmDev &d, GT &g, int dir, const uint64_t *ap, const int16_t *ae, uint32_t na, const uint64_t *gbp, const int16_t *gbe,
                             uint32_t nb, uint32_t len1, uint32_t len2, int final_min, uint64_t want, uint64_t *seen, uint32_t *f_i1, uint32_t *f_i2,
                             uint64_t *sp = nullptr, int16_t *se = nullptr) {
  const uint32_t G = (uint32_t)GT::G;
  const uint64_t I = (uint64_t)(int64_t)d.p.max_insert;
  const uint64_t mo = (uint32_t)d.p.min_read_len;
  const uint64_t X = dir == 1 ? I - len2 : (uint64_t)len1 - mo;
  const uint64_t Y = dir == 0 ? I - len1 : (uint64_t)len2 - mo;
  if (na == 0 || nb == 0) return false;  // (uniform; no pairing in this direction)
  if (STAGED) {  // the second list in the group's work arrays (see cm_coop_pair_dir)
    g.sync();
    for (uint32_t j = g.t; j < nb; j += G) { sp[j] = gbp[j]; se[j] = gbe[j]; }
    g.sync();
  }
  const uint64_t *bp = STAGED ? sp : gbp;
  const int16_t *be = STAGED ? se : gbe;
  for (uint32_t base = 0; base < na; base += G) {
    const uint32_t i1 = base + g.t;
    uint32_t cnt = 0, lo = 0;
    uint64_t p1 = 0;
    int e1 = 0;
    if (i1 < na) {
      p1 = ap[i1];
      e1 = (int)ae[i1];
      uint32_t hi = nb;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p1 > bp[mid] + X) lo = mid + 1; else hi = mid;
      }
      for (uint32_t cur = lo; cur < nb && bp[cur] <= p1 + Y; ++cur) cnt += e1 + (int)be[cur] == final_min ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t off = g.scan(cnt, &tot);
    if (*seen + tot > want) {  // the target lies in this round: in the range of the lane whose count interval holds it
      uint64_t found = 0;
      const uint64_t first = *seen + off;
      if (cnt && first <= want && want < first + cnt) {
        uint64_t k = want - first;
        for (uint32_t cur = lo; cur < nb && bp[cur] <= p1 + Y; ++cur)
          if (e1 + (int)be[cur] == final_min) {
            if (k == 0) { found = (((uint64_t)i1 << 32) | cur) + 1; break; }
            --k;
          }
      }
      found = g.max64(found);
      *f_i1 = (uint32_t)((found - 1) >> 32);
      *f_i2 = (uint32_t)(found - 1);
      return true;
    }
    *seen += tot;
  }
  return false;
}
// cm_s6c_multi for one pair (bulk paired-end, not split): the records of the sampled pairings
template <bool SAM, class GT>
CM_HD void cm_coop_s6c(const CmDev &d, uint32_t pair, GT &g, const CmCoopPeMem &m) {
  const int nb = d.pe_nbest[pair];
  if (nb <= 1 || nb > d.p.drop_rep) return;
  const uint32_t r1 = 2 * pair, r2 = r1 + 1;
  CmPe pe;
  pe.min_sum = d.pe_min[pair]; pe.second_sum = d.pe_second[pair]; pe.n_best = nb; pe.n_second = d.pe_nsecond[pair];
  pe.f_dir = d.pe_first[pair]; pe.f_i1 = d.pe_i1[pair]; pe.f_i2 = d.pe_i2[pair];
  const uint32_t K = (uint32_t)d.p.max_best;
  const uint32_t to_report = (uint32_t)nb < K ? (uint32_t)nb : K;
  const uint32_t len1 = d.rlen[r1], len2 = d.rlen[r2];
  // (the lists' places and lengths once, before the searches' barriers: cm_coop_s6a)
  const uint32_t mo1 = d.m_off[r1], mo2 = d.m_off[r2], sh1 = d.ncp[r1] + d.resc_p[r1], sh2 = d.ncp[r2] + d.resc_p[r2];
  const uint32_t nd[4] = {d.ndp[r1], d.ndn[r1], d.ndp[r2], d.ndn[r2]};
  const uint64_t *const lp[4] = {d.dpos + mo1, d.dpos + mo1 + sh1, d.dpos + mo2, d.dpos + mo2 + sh2};
  const int16_t *const le[4] = {d.derr + mo1, d.derr + mo1 + sh1, d.derr + mo2, d.derr + mo2 + sh2};
  for (uint32_t t = 0; t < to_report; ++t) {
    const int64_t want = (int64_t)d.pe_choice[(uint64_t)pair * K + t];
    if (want > 0) {
      uint64_t seen = 0;
      uint32_t i1 = 0, i2 = 0;
      int dir = 0;
      bool found = nd[3] <= m.P
          ? cm_coop_pair_find<true>(d, g, 0, lp[0], le[0], nd[0], lp[3], le[3], nd[3], len1, len2, pe.min_sum, (uint64_t)want, &seen, &i1, &i2, m.sp, m.se)
          : cm_coop_pair_find<false>(d, g, 0, lp[0], le[0], nd[0], lp[3], le[3], nd[3], len1, len2, pe.min_sum, (uint64_t)want, &seen, &i1, &i2);
      if (!found) {
        dir = 1;
        found = nd[2] <= m.P
            ? cm_coop_pair_find<true>(d, g, 1, lp[1], le[1], nd[1], lp[2], le[2], nd[2], len1, len2, pe.min_sum, (uint64_t)want, &seen, &i1, &i2, m.sp, m.se)
            : cm_coop_pair_find<false>(d, g, 1, lp[1], le[1], nd[1], lp[2], le[2], nd[2], len1, len2, pe.min_sum, (uint64_t)want, &seen, &i1, &i2);
      }
      if (!found) { if (g.t == 0) d.stats[CM_ST_ERR] = 2; return; }
      pe.f_dir = (uint32_t)dir; pe.f_i1 = i1; pe.f_i2 = i2;
    } else {
      pe.f_dir = d.pe_first[pair]; pe.f_i1 = d.pe_i1[pair]; pe.f_i2 = d.pe_i2[pair];
    }
    if (g.t == 0) cm_emit_record<SAM>(d, pair, pe, t);
  }
}

#endif
