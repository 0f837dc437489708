// hostemu.cpp -- TEST INFRASTRUCTURE.  Compiles the product's per-item stage functions
// (chromap_amd/csrc/cm_stages.h) with g++ and drives them with plain loops in the same order
// as cmgpu_map_resident (cm_api.hip), so the stage logic can be checked against the oracle
// on a machine without a GPU.  Nothing here is part of, or reachable from, the library.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <math.h>
#include <vector>

#include "../../include/chromap_amd.h"
#include "../../chromap_amd/csrc/cm_mapq_tables.h"
#include "../../chromap_amd/csrc/cm_stages.h"
// coverage counters the stage functions keep for the tests only (cm_coop.h: CM_DBG_REPLAY = minimizers whose rescue chain was replayed
// window by window because a transition table did not apply)
static unsigned long long g_dbg_rescue_replays = 0;
#define CM_DBG_REPLAY g_dbg_rescue_replays
static int g_dbg_force_replay = 0;  // tests: every second minimizer of a round is replayed although its tables apply
#define CM_DBG_FORCE_REPLAY g_dbg_force_replay
#include "../../chromap_amd/csrc/cm_coop.h"
#include "../../chromap_amd/csrc/cm_inflate.h"
#include "emu_group.h"

template <typename T>
static void scan(const T *in, uint32_t *out, uint32_t n) {
  uint32_t s = 0;
  for (uint32_t i = 0; i < n; ++i) { out[i] = s; s += in[i]; }
  out[n] = s;
}

// --chr-order for the next calls (empty: none); set by hostemu_set_chr_order
static std::vector<uint32_t> g_rank;
extern "C" void hostemu_set_chr_order(const uint32_t *rank, uint32_t n) { g_rank.assign(rank, rank + n); }
static std::vector<uint32_t> g_pairs_rank;
extern "C" void hostemu_set_pairs_chr_order(const uint32_t *rank, uint32_t n) { g_pairs_rank.assign(rank, rank + n); }

// cooperative stage functions (cm_coop.h) for the reads / pairs with long lists: group size (0 = off), the list length
// above which an item goes to a group, and the geometry of the group's shared work area
struct EmuCoop { int G; uint32_t thr, P, MM, RB; };
static EmuCoop g_coop = {0, 0, 0, 0, 0};
static bool g_planes = true;  // the bit-plane verification (k_pack_ref / k_pack_reads + cm_banded_align_planes); 0: the byte form
extern "C" void hostemu_set_planes(int on) { g_planes = on != 0; }
static unsigned long long g_coop_items[12];
// the rescue searches' small tables (tests make them smaller so that both layouts are used)
static uint32_t g_rescue_wmax_s = CM_RESCUE_WMAX_S, g_rescue_pairs_s = CM_RESCUE_PAIRS_S;
extern "C" void hostemu_set_rescue_small(uint32_t wmax, uint32_t pairs) { g_rescue_wmax_s = wmax ? wmax : CM_RESCUE_WMAX_S; g_rescue_pairs_s = pairs ? pairs : CM_RESCUE_PAIRS_S; }
  // items that went through each cooperative stage / fell back (tests look at them)
extern "C" void hostemu_set_coop(int G, uint32_t thr, uint32_t P, uint32_t MM, uint32_t RB) {
  g_coop = EmuCoop{G, thr, P, MM, RB};
  memset(g_coop_items, 0, sizeof(g_coop_items));
}
static uint32_t g_coop_slab = 0;  // entries of the global-memory slab lists longer than P use (0: none -- such lists are declined)
extern "C" void hostemu_set_coop_slab(uint32_t cap) { g_coop_slab = cap; }
static bool g_coop_reverse = false;  // lanes take their turns in descending order (emu_group.h)
extern "C" void hostemu_set_coop_order(int reverse) { g_coop_reverse = reverse != 0; }
extern "C" void hostemu_coop_items(unsigned long long *out) { memcpy(out, g_coop_items, sizeof(g_coop_items)); }

static bool g_coop_key32 = true;  // 32-bit hit keys in the cooperative hit-list stage where the reference allows (as on the device)
extern "C" void hostemu_set_coop_key32(int on) { g_coop_key32 = on != 0; }
template <int G>
static void emu_coop_s3b(const CmDev &d, const std::vector<uint32_t> &list, std::vector<uint8_t> &ok) {
  std::vector<uint8_t> mem(cm_coop_mem_bytes(g_coop.P, g_coop.MM, g_coop.RB, false) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  CmCoopMem m = cm_coop_mem_at(base, g_coop.P, g_coop.MM, g_coop.RB, false);
  std::vector<uint8_t> mem32(cm_coop_mem_bytes(g_coop.P, g_coop.MM, g_coop.RB, false, true) + 16);
  const CmCoopMem m32 = cm_coop_mem_at(mem32.data() + ((16 - ((uintptr_t)mem32.data() & 15)) & 15), g_coop.P, g_coop.MM, g_coop.RB, false, true);
  std::vector<uint8_t> slab(g_coop_slab ? cm_coop_slab_bytes(g_coop_slab) + 16 : 0);
  if (g_coop_slab) cm_coop_slab_at(m, slab.data() + ((16 - ((uintptr_t)slab.data() & 15)) & 15), g_coop_slab);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    for (size_t i = 0; i < list.size(); ++i) {
      // a list longer than the work area goes to the slab (the device has a launch of its own for those)
      const bool done = g_coop_slab && d.hit_tot[list[i]] > g_coop.P ? cm_coop_s3b<true>(d, list[i], g, m) : (d.goff ? cm_coop_s3b_k32(d, list[i], g, m32) : cm_coop_s3b<false>(d, list[i], g, m));
      if (g.t == 0) ok[i] = done ? 1 : 0;
      g.sync();
    }
  }, g_coop_reverse);
}

template <int G>
static void emu_coop_rescue(const CmDev &d, const std::vector<uint32_t> &list) {
  std::vector<uint8_t> mem(cm_coop_mem_bytes(g_coop.P, g_coop.MM, g_coop.RB, true) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  CmCoopMem m = cm_coop_mem_at(base, g_coop.P, g_coop.MM, g_coop.RB, true);
  std::vector<uint8_t> slab(g_coop_slab ? cm_coop_slab_bytes(g_coop_slab) + 16 : 0);
  if (g_coop_slab) cm_coop_slab_at(m, slab.data() + ((16 - ((uintptr_t)slab.data() & 15)) & 15), g_coop_slab);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    for (size_t i = 0; i < list.size(); ++i) {
      {
        const uint32_t r_ = list[i], big_ = d.resc_p[r_] > d.resc_n[r_] ? d.resc_p[r_] : d.resc_n[r_];
        if (g_coop_slab && big_ > g_coop.P) cm_coop_rescue_merge<true>(d, r_, g, m); else cm_coop_rescue_merge<false>(d, r_, g, m);
      }
      g.sync();
    }
  }, g_coop_reverse);
}

// the rescue searches of the listed reads by a group each: the counting pass (k_s4a_rescue_list) or the fill pass (k_s4b_rescue_list)
template <int G>
static void emu_coop_rescue_search(const CmDev &d, const std::vector<uint32_t> &list, bool fill) {
  // as on the device (k_s4a/4b_rescue_wave): tables for searches of up to g_rescue_wmax_s best mate candidates; a read with a longer
  // search goes to the full tables
  std::vector<uint64_t> mem(cm_coop_rescue_mem_bytes() / 8 + 4), mem_s(cm_coop_rescue_mem_bytes(g_rescue_wmax_s, g_rescue_pairs_s) / 8 + 4);
  CmCoopRescueMem m = cm_coop_rescue_mem_at((uint8_t *)mem.data());
  CmCoopRescueMem ms = cm_coop_rescue_mem_at((uint8_t *)mem_s.data(), g_rescue_wmax_s, g_rescue_pairs_s);
  m.grant = ms.grant = 100;  // (the emulated pool holds 3000 entries)
  emu_run_group<G>([&](EmuGroup<G> &g) {
    cm_coop_rescue_mem_reset(g, m);
    cm_coop_rescue_mem_reset(g, ms);
    for (size_t i = 0; i < list.size(); ++i) {
      const bool small = cm_coop_rescue_fits(d, list[i], g, ms.wmax);
      if (g.t == 0) ++g_coop_items[small ? 10 : 11];
      if (fill) { if (d.resc_n[list[i]] + d.resc_p[list[i]] > 0) cm_coop_s4b_fill(d, list[i], g, small ? ms : m); }
      else cm_coop_s4a_rescue(d, list[i], g, small ? ms : m);
      g.sync();
    }
    cm_coop_rescue_mem_flush(d, g, m);
    cm_coop_rescue_mem_flush(d, g, ms);
  }, g_coop_reverse);
}

template <int G>
static void emu_coop_s4c(const CmDev &d, const std::vector<uint32_t> &list) {
  const uint32_t P = g_coop.P > 60000 ? 60000 : g_coop.P;
  std::vector<uint8_t> mem(cm_coop_pair_mem_bytes(P) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  const CmCoopPairMem m = cm_coop_pair_mem_at(base, P);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    for (size_t i = 0; i < list.size(); ++i) {
      // every other pair through the form that leaves the position lists in global memory (k_s4c_coop<1024, false>; the work
      // arrays of the staged layout hold the unstaged one's)
      if (list[i] & 1u) cm_coop_s4c<false>(d, list[i], g, m); else cm_coop_s4c<true>(d, list[i], g, m);
      g.sync();
    }
  }, g_coop_reverse);
}

template <int G>
static void emu_coop_s5c(const CmDev &d, const std::vector<uint32_t> &list, int phase) {
  const uint32_t P = g_coop.P > 60000 ? 60000 : g_coop.P;
  std::vector<uint8_t> mem(cm_coop_ver_mem_bytes(P) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  const CmCoopVerMem m = cm_coop_ver_mem_at(base, P);
  std::vector<uint16_t> hist((size_t)G * 64);
  std::vector<uint64_t> stage_p(24);  // the candidate sort's staging area, deliberately small: longer lists take the scratch path
  std::vector<uint8_t> stage_c(24);
  // the sort's work area deliberately small: lists beyond 32 entries sort in global memory
  std::vector<uint8_t> smem(cm_coop_sort_mem_bytes(32, 40) + 16);
  const CmCoopSortMem sm = cm_coop_sort_mem_at(smem.data() + ((16 - ((uintptr_t)smem.data() & 15)) & 15), 32, 40);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    for (size_t i = 0; i < list.size(); ++i) {
      if (phase == 0) cm_coop_s5_sort(d, list[i], g, hist.data(), 64, stage_p.data(), stage_c.data(), 24); else cm_coop_s5c(d, list[i], g, m, sm);
      g.sync();
    }
  }, g_coop_reverse);
}
// the pairing stages' staging arrays, deliberately small: second lists of up to 40 entries are staged, longer ones read in place
#define EMU_PE_P 40u
template <int G>
static void emu_coop_s6a(const CmDev &d, const std::vector<uint32_t> &list) {
  std::vector<uint64_t> pem(cm_coop_pe_mem_bytes(EMU_PE_P) / 8 + 2);
  const CmCoopPeMem m = cm_coop_pe_mem_at((uint8_t *)pem.data(), EMU_PE_P);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    for (size_t i = 0; i < list.size(); ++i) {
      cm_coop_s6a<false>(d, list[i], g, m);
      g.sync();
    }
  }, g_coop_reverse);
}
template <int G>
static void emu_coop_s6c(const CmDev &d, const std::vector<uint32_t> &list) {
  std::vector<uint64_t> pem(cm_coop_pe_mem_bytes(EMU_PE_P) / 8 + 2);
  const CmCoopPeMem m = cm_coop_pe_mem_at((uint8_t *)pem.data(), EMU_PE_P);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    for (size_t i = 0; i < list.size(); ++i) {
      cm_coop_s6c<false>(d, list[i], g, m);
      g.sync();
    }
  }, g_coop_reverse);
}

struct EmuSam {
  cmgpu_sam_record *rec;  // 2n (pairs) or n (single) slots
  uint32_t *cigar;
  char *md;
  uint32_t md_cap;
};

struct EmuBarcodes {
  const cmgpu_barcode_batch *bc;
  const uint64_t *wl_keys;
  uint32_t n_keys;
  uint64_t *bc_key_out;  // [n], packed like the records
  uint64_t *bc_key_all;  // [n] per pair, or NULL
};

static int emu_map_pairs(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                         const cmgpu_batch *in, cmgpu_record *out, uint64_t *n_out, cmgpu_stats *stats,
                         uint32_t *dbg_mm_cnt /* 2n or NULL */, uint32_t *dbg_ncand /* 2n */,
                         uint32_t *dbg_ndraft /* 2n */, int32_t *dbg_nbest /* n */, const EmuBarcodes *eb, bool single = false,
                         const EmuSam *sam = nullptr) {
  const uint32_t n = in->n_pairs, n2 = 2 * n;
  CmDev d;
  memset(&d, 0, sizeof(d));
  // ---- index re-pack (k_repack)
  const uint32_t nb = index->n_buckets;
  std::vector<uint64_t> bkt((size_t)nb * 2);
  for (uint32_t i = 0; i < nb; ++i) {
    const uint32_t f = (index->flags[i >> 4] >> ((i & 0xfU) << 1)) & 3u;
    uint64_t k = index->keys[i], v = index->vals[i];
    if (f & 2u) { k = CM_EMPTY_KEY; v = 0; } else if (f & 1u) { k = CM_DELETED_KEY; v = 0; }
    bkt[2 * (size_t)i] = k; bkt[2 * (size_t)i + 1] = v;
  }
  d.bkt = bkt.data(); d.bmask = nb - 1; d.occ = index->occurrences; d.n_occ = index->n_occurrences;
  // ---- reference layout
  std::vector<uint64_t> roff(ref->n_sequences);
  uint64_t tot = 64;
  for (uint32_t i = 0; i < ref->n_sequences; ++i) { roff[i] = tot; tot += (uint64_t)ref->lengths[i] + 64; tot = (tot + 15) & ~15ull; }
  std::vector<uint8_t> refb(tot, 0);
  for (uint32_t i = 0; i < ref->n_sequences; ++i) memcpy(refb.data() + roff[i], ref->sequences[i], ref->lengths[i]);
  d.ref = refb.data(); d.ref_off = roff.data(); d.ref_len = ref->lengths; d.n_seq = ref->n_sequences;
  std::vector<uint32_t> goff(ref->n_sequences + 1);  // CmDev::goff as cm_fill_dev_range builds it
  {
    uint64_t acc = 0;
    for (uint32_t i = 0; i < ref->n_sequences; ++i) { goff[i] = (uint32_t)acc; acc += (uint64_t)ref->lengths[i] + CM_GOFF_GAP; }
    goff[ref->n_sequences] = (uint32_t)acc;
    d.goff = acc < 0xffff0000ull && g_coop_key32 ? goff.data() : nullptr;
  }
  // the reference as bit planes (k_pack_ref): k_s5b_verify aligns on them
  const uint64_t rplw = (tot + 31) / 32 + 4;
  std::vector<CmPlRec> refpl;
  if (g_planes) {
    refpl.assign((size_t)rplw + CM_PL_LEAD, CmPlRec{0u, 0u, 0u, 0u});
    for (uint64_t w = 0; w * 32 < tot; ++w) {
      CmPlRec &q = refpl[CM_PL_LEAD + w];
      cm_pack_planes32(refb.data() + 32 * w, (uint32_t)(tot - 32 * w < 32 ? tot - 32 * w : 32), &q.p0, &q.p1, &q.pn, &q.pc);
    }
    d.ref_pl = refpl.data() + CM_PL_LEAD; d.ref_pl_words = rplw;
  }
  std::vector<uint64_t> roff_r(ref->n_sequences);
  std::vector<uint32_t> rlen_r(ref->n_sequences);
  if (g_rank.size() == ref->n_sequences) {  // cmgpu_set_chr_order: the stages see the reference reordered by rank
    for (uint32_t i = 0; i < ref->n_sequences; ++i) { roff_r[g_rank[i]] = roff[i]; rlen_r[g_rank[i]] = ref->lengths[i]; }
    d.ref_off = roff_r.data(); d.ref_len = rlen_r.data(); d.rid_rank = g_rank.data();
  }
  if (g_pairs_rank.size() == ref->n_sequences) d.pairs_rank = g_pairs_rank.data();
  CmParams &p = d.p;
  p.e = params->error_threshold; p.min_seeds = params->min_num_seeds; p.f0 = params->max_seed_frequency0;
  p.f1 = params->max_seed_frequency1; p.max_insert = params->max_insert_size; p.min_read_len = params->min_read_length;
  p.max_best = params->max_num_best_mappings; p.drop_rep = params->drop_repetitive_reads; p.trim = params->trim_adapters; p.split = params->split_alignment ? 1 : 0;
  p.bc_err = params->bc_error_threshold; p.bc_keep = params->output_mappings_not_in_whitelist ? 1 : 0; p.bc_prob = params->bc_probability_threshold;
  p.single = single ? 1 : 0;
  p.sam = sam ? 1 : 0;  // also read by the split-alignment draft mappings (position rule of the - strand)
  p.k = index->kmer_size; p.w = index->window_size; p.lanes = p.split ? 0 : (p.e < 8 ? 8 : (p.e < 16 ? 4 : 0));
  p.ref_batch = params->read_batch_size > 0 ? params->read_batch_size : 500000;
  p.grain = params->taskloop_grain_size > 0 ? params->taskloop_grain_size : 5000;
  std::vector<double> coef; std::vector<uint32_t> brk;
  cm_build_len_coef(coef); cm_build_nsec_break(brk);
  d.mq.len_coef = coef.data(); d.mq.nsec_break = brk.data(); d.mq.n_break = (int)brk.size();
  d.n_pairs = n; d.first_read_id = in->first_read_id;
  // padded copies: the byte readers (cm_stages.h) load whole aligned 8-byte words around a byte range
  std::vector<uint64_t> pad0(((size_t)(n ? in->read1_offsets[n] : 0) + 31) / 8 + 2, 0), pad1(((size_t)(n ? in->read2_offsets[n] : 0) + 31) / 8 + 2, 0);
  if (n) { memcpy(pad0.data(), in->read1_bases, in->read1_offsets[n]); memcpy(pad1.data(), in->read2_bases, in->read2_offsets[n]); }
  d.rb0 = (const uint8_t *)pad0.data(); d.rb1 = (const uint8_t *)pad1.data();
  d.ro0 = in->read1_offsets; d.ro1 = in->read2_offsets;
  unsigned long long st[CM_ST_N];
  memset(st, 0, sizeof(st));
  d.stats = st;
#define VEC(name, T, cnt) std::vector<T> v_##name((size_t)(cnt) + 1); d.name = v_##name.data();
  VEC(rlen, uint32_t, n2) VEC(mm_cnt, uint32_t, n2) VEC(mm_off, uint32_t, n2 + 1)
  VEC(hit_tot, uint32_t, n2) VEC(hit_off, uint32_t, n2 + 1) VEC(round2, uint8_t, n2) VEC(rep_cnt, uint32_t, n2)
  VEC(rep_len, uint32_t, n2) VEC(n_pos_hit, uint32_t, n2) VEC(ncp, uint32_t, n2) VEC(ncn, uint32_t, n2)
  VEC(aug, uint8_t, n2) VEC(res_neg, int32_t, n2) VEC(res_pos, int32_t, n2) VEC(resc_n, uint32_t, n2) VEC(resc_p, uint32_t, n2)
  VEC(m_tot, uint32_t, n2) VEC(m_off, uint32_t, n2 + 1) VEC(mcp, uint32_t, n2) VEC(mcn, uint32_t, n2) VEC(force0, uint8_t, n)
  VEC(fcp, uint32_t, n2) VEC(fcn, uint32_t, n2) VEC(alive, uint8_t, n) VEC(ndp, uint32_t, n2) VEC(ndn, uint32_t, n2)
  VEC(min_err, int32_t, n2) VEC(second_err, int32_t, n2) VEC(n_best, int32_t, n2) VEC(n_second, int32_t, n2)
  VEC(pe_min, int32_t, n) VEC(pe_second, int32_t, n) VEC(pe_nbest, int32_t, n) VEC(pe_nsecond, int32_t, n)
  VEC(pe_first, uint32_t, n) VEC(pe_i1, uint32_t, n) VEC(pe_i2, uint32_t, n) VEC(pe_choice, uint32_t, (size_t)n * (size_t)(p.max_best > 0 ? p.max_best : 1))
  VEC(rec, uint8_t, (size_t)n * 24 * (size_t)(p.max_best > 0 ? p.max_best : 1)) VEC(rec_ok, uint8_t, (size_t)n * (size_t)(p.max_best > 0 ? p.max_best : 1))
  // ---- K6 (cmgpu_set_whitelist + cmgpu_compute_barcode_abundance + k_s0b_barcode)
  std::vector<uint64_t> wl_tab;
  std::vector<double> pw(81);
  std::vector<uint64_t> v_bc_key(n + 1);
  std::vector<uint8_t> v_bc_ok(n + 1);
  if (eb) {
    uint32_t wnb = 16;
    while (wnb < 2ull * eb->n_keys + 16) wnb <<= 1;
    wl_tab.assign((size_t)wnb * 2, 0);
    for (uint32_t i = 0; i < wnb; ++i) wl_tab[2 * (size_t)i] = ~0ull;
    for (uint32_t i = 0; i < eb->n_keys; ++i) {
      const uint64_t x = eb->wl_keys[i] * 0x9E3779B97F4A7C15ull;
      uint32_t b = (uint32_t)(x >> 32) & (wnb - 1);
      while (wl_tab[2 * (size_t)b] != ~0ull && wl_tab[2 * (size_t)b] != eb->wl_keys[i]) b = (b + 1) & (wnb - 1);
      wl_tab[2 * (size_t)b] = eb->wl_keys[i];
    }
    for (int q = 0; q <= 80; ++q) pw[q] = pow(10.0, ((-q) / 10.0));
    d.bcb = (const uint8_t *)eb->bc->bases; d.bcq = (const uint8_t *)eb->bc->qualities; d.bco = eb->bc->offsets;
    d.wl = wl_tab.data(); d.wl_mask = wnb - 1; d.pow10_tab = pw.data();
    d.bc_key = v_bc_key.data(); d.bc_ok = v_bc_ok.data();
    // abundance over all barcodes of the input (one reference batch here)
    uint64_t ns = 0;
    for (uint32_t i = 0; i < n; ++i) {
      const uint8_t *sq = d.bcb + d.bco[i];
      const uint32_t l = d.bco[i + 1] - d.bco[i];
      bool has_n = false;
      for (uint32_t j = 0; j < l; ++j) has_n |= sq[j] == 'N';
      if (has_n) continue;
      const uint64_t key = cm_seed_from_sequence(sq, l);
      const uint64_t x = key * 0x9E3779B97F4A7C15ull;
      uint32_t b = (uint32_t)(x >> 32) & d.wl_mask;
      while (wl_tab[2 * (size_t)b] != ~0ull) {
        if (wl_tab[2 * (size_t)b] == key) { wl_tab[2 * (size_t)b + 1] += 1; ++ns; break; }
        b = (b + 1) & d.wl_mask;
      }
    }
    d.wl_num_sample = (double)ns;
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t a = 0, b = 0;
      cm_s0b_barcode(d, i, &a, &b);
      st[CM_ST_BC_INWL] += a;
      st[CM_ST_BC_CORR] += b;
    }
  }
  for (uint32_t i = 0; i < n; ++i) cm_s0_prep(d, i);
  // k_prep_count (above: cm_s0_prep per pair) + cm_s1_count per read, scan, k_mm_fill
  for (uint32_t r = 0; r < n2; ++r) cm_s1_count(d, r, cm_read_ptr(d, r));
  scan(d.mm_cnt, d.mm_off, n2);
  const uint32_t n_mm = d.mm_off[n2];
  VEC(mm_hash, uint64_t, n_mm) VEC(mm_ps, uint32_t, n_mm) VEC(pr_val, uint64_t, n_mm) VEC(pr_kind, uint8_t, n_mm)
  for (uint32_t r = 0; r < n2; ++r) cm_s1_fill(d, r, cm_read_ptr(d, r));
  for (uint32_t i = 0; i < n_mm; ++i) {
    const uint32_t steps = cm_probe(d.bkt, d.bmask, d.mm_hash[i], &d.pr_val[i], &d.pr_kind[i]);
    st[CM_ST_PROBE_STEPS] += steps;
    st[CM_ST_PROBE_HITS] += d.pr_kind[i] != CM_PR_MISS;
  }
  for (uint32_t r = 0; r < n2; ++r) cm_s3a_count(d, r);
  scan(d.hit_tot, d.hit_off, n2);
  VEC(hbuf, uint64_t, d.hit_off[n2]) VEC(hcnt, uint8_t, d.hit_off[n2])
  {  // k_s3b_candidates: lists of <= 16 hits are worked on in a strided (LDS-like) buffer
    std::vector<uint64_t> wh(16 * 7 + 8);
    std::vector<uint8_t> wc(16 * 7 + 8);
    std::vector<uint32_t> heavy;
    for (uint32_t r = 0; r < n2; ++r) {
      if (g_coop.G && d.hit_tot[r] > g_coop.thr) { heavy.push_back(r); continue; }
      cm_s3b_candidates_lds(d, r, wh.data() + (r % 7), wc.data() + (r % 7), 16, 7);
    }
    if (!heavy.empty()) {  // k_s3b_coop: a group of lanes per read; what it declines goes to the one-lane path
      std::vector<uint8_t> ok(heavy.size(), 0);
      if (g_coop.G == 16) emu_coop_s3b<16>(d, heavy, ok); else if (g_coop.G == 64) emu_coop_s3b<64>(d, heavy, ok);
      else if (g_coop.G == 256) emu_coop_s3b<256>(d, heavy, ok); else emu_coop_s3b<1024>(d, heavy, ok);
      for (size_t i = 0; i < heavy.size(); ++i) {
        g_coop_items[ok[i] ? 0 : 1] += 1;
        if (!ok[i]) cm_s3b_candidates(d, heavy[i]);
      }
    }
  }
  // k_s4a_rescue_count / k_s4a_rescue_list: with the cooperative forms on, a read whose mate has 4 candidates or more on a strand
  // has its rescue searches run by a group (cm_coop_rescue; on the device: a wave, CM_RS_WAVE)
  std::vector<uint32_t> rescue_wave;
  std::vector<uint8_t> is_rescue_wave(n2, 0);
  // the pool of rescue hits found while counting (CmDev::rs_pool), small enough to run out on the repeat-rich cases
  std::vector<uint64_t> rs_pool(g_coop.G ? 3000 : 0);
  std::vector<uint32_t> rs_pool_off(g_coop.G ? 2 * (size_t)n2 : 0, 0xffffffffu);
  if (g_coop.G) { d.rs_pool = rs_pool.data(); d.rs_pool_cap = (uint32_t)rs_pool.size(); d.rs_pool_off = rs_pool_off.data(); }
  for (uint32_t r = 0; r < n2; ++r) {
    if (!cm_s4a_decide(d, r)) continue;
    if (g_coop.G && (d.ncp[r ^ 1u] >= 4 || d.ncn[r ^ 1u] >= 4)) { rescue_wave.push_back(r); is_rescue_wave[r] = 1; }
    else cm_s4a_rescue(d, r);
  }
  if (!rescue_wave.empty()) {
    g_coop_items[7] += rescue_wave.size();
    if (g_coop.G == 16) emu_coop_rescue_search<16>(d, rescue_wave, false); else if (g_coop.G == 64) emu_coop_rescue_search<64>(d, rescue_wave, false);
    else if (g_coop.G == 256) emu_coop_rescue_search<256>(d, rescue_wave, false); else emu_coop_rescue_search<1024>(d, rescue_wave, false);
  }
  scan(d.m_tot, d.m_off, n2);
  const uint32_t n_m = d.m_off[n2];
  VEC(mbuf, uint64_t, n_m) VEC(mcnt, uint8_t, n_m) VEC(fbuf, uint64_t, n_m) VEC(fcnt, uint8_t, n_m)
  VEC(dpos, uint64_t, n_m) VEC(derr, int16_t, n_m) VEC(dsplit, uint32_t, n_m)
  {
    std::vector<uint32_t> heavy;
    if (!rescue_wave.empty()) {  // the fill pass of the reads whose searches a group ran
      if (g_coop.G == 16) emu_coop_rescue_search<16>(d, rescue_wave, true); else if (g_coop.G == 64) emu_coop_rescue_search<64>(d, rescue_wave, true);
      else if (g_coop.G == 256) emu_coop_rescue_search<256>(d, rescue_wave, true); else emu_coop_rescue_search<1024>(d, rescue_wave, true);
    }
    for (size_t q = 0; q < rs_pool_off.size(); ++q) g_coop_items[rs_pool_off[q] != 0xffffffffu ? 8 : 9] += is_rescue_wave[q >> 1] ? 1 : 0;
    for (uint32_t r = 0; r < n2; ++r) {
      const uint32_t big = d.resc_p[r] > d.resc_n[r] ? d.resc_p[r] : d.resc_n[r];
      const bool filled = is_rescue_wave[r] && d.resc_p[r] + d.resc_n[r] > 0;
      if (g_coop.G && d.aug[r] && big > g_coop.thr) {  // k_s4b_rescue_list fills, k_s4b_coop sorts and merges
        if (!filled) cm_s4b_rescue_merge(d, r, CM_S4B_FILL_ONLY);
        heavy.push_back(r);
      } else {
        cm_s4b_rescue_merge(d, r, filled ? CM_S4B_PREFILLED : CM_S4B_ALL);
      }
    }
    if (!heavy.empty()) {
      g_coop_items[2] += heavy.size();
      if (g_coop.G == 16) emu_coop_rescue<16>(d, heavy); else if (g_coop.G == 64) emu_coop_rescue<64>(d, heavy);
      else if (g_coop.G == 256) emu_coop_rescue<256>(d, heavy); else emu_coop_rescue<1024>(d, heavy);
    }
  }
  {
    std::vector<uint32_t> heavy;
    for (uint32_t i = 0; i < n; ++i) {
      if (!cm_s4c_pre(d, i)) continue;
      uint32_t big = 0;
      for (uint32_t r = 2 * i; r <= 2 * i + 1; ++r) { big = d.mcp[r] > big ? d.mcp[r] : big; big = d.mcn[r] > big ? d.mcn[r] : big; }
      if (g_coop.G && big > g_coop.thr) { heavy.push_back(i); continue; }
      cm_s4c_filter(d, i);
      cm_s4c_post(d, i);
    }
    if (!heavy.empty()) {  // k_s4c_coop
      g_coop_items[3] += heavy.size();
      if (g_coop.G == 16) emu_coop_s4c<16>(d, heavy); else if (g_coop.G == 64) emu_coop_s4c<64>(d, heavy);
      else if (g_coop.G == 256) emu_coop_s4c<256>(d, heavy); else emu_coop_s4c<1024>(d, heavy);
    }
  }
  std::vector<uint32_t> readpl;  // k_pack_reads
  if (g_planes) {
    uint32_t maxlen = 1;
    for (uint32_t r = 0; r < n2; ++r) maxlen = d.rlen[r] > maxlen ? d.rlen[r] : maxlen;
    d.read_pl_w = (maxlen + 31) / 32;
    readpl.assign((size_t)n2 * cm_read_pl_stride(d.read_pl_w) + 4, 0xA5A5A5A5u);
    d.read_pl = readpl.data();
    for (uint32_t r = 0; r < n2; ++r) cm_pack_read_planes(d, r);
  }
  VEC(nv, uint32_t, n2) VEC(v_off, uint32_t, n2 + 1) VEC(v_err, int16_t, n_m) VEC(v_end, int16_t, n_m)
  std::vector<uint32_t> s5_heavy;  // reads whose verification and acceptance a group of lanes runs (k_s5c_coop)
  for (uint32_t r = 0; r < n2; ++r)
    if (cm_s5a_prepare(d, r, g_coop.G && !d.p.split ? g_coop.thr : 0u)) s5_heavy.push_back(r);
  const uint32_t s5_min = g_coop.G && !d.p.split ? g_coop.thr : 0u;
  auto s5_groups = [&](int phase) {
    if (s5_heavy.empty()) return;
    if (g_coop.G == 16) emu_coop_s5c<16>(d, s5_heavy, phase); else if (g_coop.G == 64) emu_coop_s5c<64>(d, s5_heavy, phase);
    else if (g_coop.G == 256) emu_coop_s5c<256>(d, s5_heavy, phase); else emu_coop_s5c<1024>(d, s5_heavy, phase);
  };
  s5_groups(0);  // k_s5_sort_coop: the heavy reads' candidate lists
  scan(d.nv, d.v_off, n2);
  for (uint32_t j = 0; j < d.v_off[n2]; ++j) cm_s5b_verify_item(d, j, n2);
  for (uint32_t r = 0; r < n2; ++r) cm_s5c_finalize(d, r, s5_min);
  g_coop_items[4] += s5_heavy.size();
  s5_groups(1);  // k_s5c_coop
  // --SAM buffers (cmgpu_map_resident allocates the same per batch)
  std::vector<uint32_t> samz;
  if (sam) {
    uint32_t mx = 1;
    for (uint32_t r = 0; r < n2; ++r) mx = d.rlen[r] > mx ? d.rlen[r] : mx;
    samz.assign((size_t)mx * 8 * (n ? n : 1) + 8, 0);
    d.p.sam = 1; d.sam_rec = (uint8_t *)sam->rec; d.sam_cigar = sam->cigar; d.sam_md = (uint8_t *)sam->md; d.sam_md_cap = sam->md_cap;
    d.sam_z = samz.data();
    memset(sam->rec, 0, (size_t)(single ? n : n2) * sizeof(cmgpu_sam_record));
  }
  {
    std::vector<uint32_t> heavy;
    for (uint32_t i = 0; i < n; ++i) {
      if (sam) { cm_s6a_pair<true>(d, i); continue; }
      if (!cm_s6a_pre<false>(d, i)) continue;
      uint32_t big = 0;
      for (uint32_t r = 2 * i; r <= 2 * i + 1; ++r) { big = d.ndp[r] > big ? d.ndp[r] : big; big = d.ndn[r] > big ? d.ndn[r] : big; }
      if (g_coop.G && big > g_coop.thr) heavy.push_back(i); else cm_s6a_sweeps<false>(d, i);
    }
    if (!heavy.empty()) {  // k_s6a_coop
      g_coop_items[5] += heavy.size();
      if (g_coop.G == 16) emu_coop_s6a<16>(d, heavy); else if (g_coop.G == 64) emu_coop_s6a<64>(d, heavy);
      else if (g_coop.G == 256) emu_coop_s6a<256>(d, heavy); else emu_coop_s6a<1024>(d, heavy);
    }
  }
  const uint32_t nch = cm_num_chunks(n, (uint32_t)p.ref_batch, (uint32_t)p.grain);
  CmMt *g = new CmMt();
  for (uint32_t c = 0; c < nch; ++c) cm_s6b_sample(d, c, *g);
  delete g;
  {
    std::vector<uint32_t> heavy;  // k_s6c_coop: multi-mapped pairs with long draft lists
    for (uint32_t i = 0; i < n; ++i) {
      if (sam) { cm_s6c_multi<true>(d, i); continue; }
      bool to_group = false;
      if (g_coop.G && !d.p.single && !d.p.split && d.pe_nbest[i] > 1) {
        uint32_t big = d.ndp[2 * i] > d.ndn[2 * i] ? d.ndp[2 * i] : d.ndn[2 * i];
        big = d.ndp[2 * i + 1] > big ? d.ndp[2 * i + 1] : big;
        big = d.ndn[2 * i + 1] > big ? d.ndn[2 * i + 1] : big;
        to_group = big > g_coop.thr;
      }
      if (to_group) heavy.push_back(i); else cm_s6c_multi<false>(d, i);
    }
    if (!heavy.empty()) {
      g_coop_items[6] += heavy.size();
      if (g_coop.G == 16) emu_coop_s6c<16>(d, heavy); else if (g_coop.G == 64) emu_coop_s6c<64>(d, heavy);
      else if (g_coop.G == 256) emu_coop_s6c<256>(d, heavy); else emu_coop_s6c<1024>(d, heavy);
    }
  }
  uint64_t k = 0;
  for (uint32_t i = 0; i < n; ++i) {
    // k_stats
    const uint32_t r1 = 2 * i, r2 = r1 + 1;
    if (d.alive[i]) {
      st[CM_ST_CAND] += d.fcp[r1] + d.fcn[r1] + d.fcp[r2] + d.fcn[r2];
      const uint32_t nd1 = d.ndp[r1] + d.ndn[r1], nd2 = d.ndp[r2] + d.ndn[r2];
      const unsigned long long per = single ? 1ull : 2ull;
      if (nd1 > 0 && (single || nd2 > 0)) {
        const int nbst = d.pe_nbest[i];
        if (nbst == 1) st[CM_ST_UNIQ] += per;
        st[CM_ST_MAPPINGS] += per * (unsigned long long)(nbst < p.max_best ? nbst : p.max_best);
        if (nbst > 0) st[CM_ST_MAPPED] += per;
        if (nbst > 1 && (single || nbst <= p.drop_rep)) st[CM_ST_MULTI] += 1;
      }
    }
    st[CM_ST_RESCUED] += d.aug[r1] + d.aug[r2];
    st[CM_ST_OCC] += d.hit_tot[r1] + d.hit_tot[r2];
    for (size_t sl = (size_t)i * (size_t)p.max_best; sl < ((size_t)i + 1) * (size_t)p.max_best; ++sl)  // max_best record slots per pair
      if (d.rec_ok[sl]) { if (eb) eb->bc_key_out[k] = d.bc_key[i]; memcpy(&out[k++], d.rec + sl * 24, 24); }
    if (dbg_nbest) dbg_nbest[i] = d.pe_nbest[i];
  }
  if (eb && eb->bc_key_all) for (uint32_t i = 0; i < n; ++i) eb->bc_key_all[i] = d.bc_key[i];
  for (uint32_t r = 0; r < n2; ++r) {
    if (dbg_mm_cnt) dbg_mm_cnt[r] = d.mm_cnt[r];
    if (dbg_ncand) dbg_ncand[r] = d.alive[r >> 1] ? d.fcp[r] + d.fcn[r] : 0;
    if (dbg_ndraft) dbg_ndraft[r] = d.ndp[r] + d.ndn[r];
  }
  *n_out = k;
  if (stats) {
    stats->num_candidates += st[CM_ST_CAND];
    stats->num_mappings += st[CM_ST_MAPPINGS];
    stats->num_mapped_reads += st[CM_ST_MAPPED];
    stats->num_uniquely_mapped_reads += st[CM_ST_UNIQ];
    stats->num_minimizers += n_mm;
    stats->probe_steps += st[CM_ST_PROBE_STEPS];
    stats->occurrences_read += st[CM_ST_OCC];
    stats->num_pairs_rescued += st[CM_ST_RESCUED];
    stats->num_multi_mappers += st[CM_ST_MULTI];
    stats->num_barcode_in_whitelist += st[CM_ST_BC_INWL];
    stats->num_corrected_barcode += st[CM_ST_BC_CORR];
  }
  return st[CM_ST_ERR] ? -(int)st[CM_ST_ERR] : 0;
}

extern "C" int hostemu_map_pairs(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                 const cmgpu_batch *in, cmgpu_record *out, uint64_t *n_out, cmgpu_stats *stats,
                                 uint32_t *dbg_mm_cnt, uint32_t *dbg_ncand, uint32_t *dbg_ndraft, int32_t *dbg_nbest) {
  return emu_map_pairs(index, ref, params, in, out, n_out, stats, dbg_mm_cnt, dbg_ncand, dbg_ndraft, dbg_nbest, nullptr);
}

extern "C" int hostemu_map_single(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                  const cmgpu_single_batch *in, cmgpu_record *out, uint64_t *n_out, cmgpu_stats *stats) {
  std::vector<uint32_t> zero((size_t)in->n_reads + 1, 0);
  cmgpu_batch b{in->n_reads, in->first_read_id, in->bases, in->offsets, "", zero.data()};
  return emu_map_pairs(index, ref, params, &b, out, n_out, stats, nullptr, nullptr, nullptr, nullptr, nullptr, true);
}

extern "C" int hostemu_map_pairs_bc(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                    const cmgpu_batch *in, const cmgpu_barcode_batch *bc, const uint64_t *wl_keys,
                                    uint32_t n_keys, cmgpu_record_bc *out, uint64_t *n_out, cmgpu_stats *stats) {
  std::vector<cmgpu_record> rec(in->n_pairs + 1);
  std::vector<uint64_t> keys(in->n_pairs + 1);
  EmuBarcodes eb{bc, wl_keys, n_keys, keys.data(), nullptr};
  const int rc = emu_map_pairs(index, ref, params, in, rec.data(), n_out, stats, nullptr, nullptr, nullptr, nullptr, &eb);
  for (uint64_t i = 0; i < *n_out; ++i) { out[i].r = rec[i]; out[i].barcode = keys[i]; }
  return rc;
}

// single-end reads with cell barcodes (cmgpu_map_single_barcoded's stage sequence)
extern "C" int hostemu_map_single_bc(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                     const cmgpu_single_batch *in, const cmgpu_barcode_batch *bc, const uint64_t *wl_keys,
                                     uint32_t n_keys, cmgpu_record_bc *out, uint64_t *n_out, cmgpu_stats *stats) {
  std::vector<uint32_t> zero((size_t)in->n_reads + 1, 0);
  cmgpu_batch b{in->n_reads, in->first_read_id, in->bases, in->offsets, "", zero.data()};
  std::vector<cmgpu_record> rec(in->n_reads + 1);
  std::vector<uint64_t> keys(in->n_reads + 1);
  EmuBarcodes eb{bc, wl_keys, n_keys, keys.data(), nullptr};
  const int rc = emu_map_pairs(index, ref, params, &b, rec.data(), n_out, stats, nullptr, nullptr, nullptr, nullptr, &eb, true);
  for (uint64_t i = 0; i < *n_out; ++i) { out[i].r = rec[i]; out[i].barcode = keys[i]; }
  return rc;
}

// chunked reference-minimizer collection (cm_ref_chunk_minimizers) concatenated in chunk
// order; compared by tests with the oracle's sequential pass
extern "C" long hostemu_ref_minimizers(const cmgpu_ref_view *ref, int k, int w, uint32_t chunk, uint32_t warm,
                                       uint64_t *out_hash, uint64_t *out_hit, long cap) {
  long n = 0;
  for (uint32_t r = 0; r < ref->n_sequences; ++r) {
    std::vector<uint8_t> seq((size_t)ref->lengths[r] + 64, 0);
    memcpy(seq.data(), ref->sequences[r], ref->lengths[r]);
    for (uint64_t s = 0; s < ref->lengths[r]; s += chunk) {
      const uint32_t c = cm_ref_chunk_minimizers(seq.data(), ref->lengths[r], r, (uint32_t)s, chunk, warm, k, w, nullptr, nullptr);
      if (n + c > cap) return -1;
      cm_ref_chunk_minimizers(seq.data(), ref->lengths[r], r, (uint32_t)s, chunk, warm, k, w, out_hash + n, out_hit + n);
      n += c;
    }
  }
  return n;
}

// one read through the w = 7 minimizer front end (variant 1) or the state machine alone (variant 0)
extern "C" int hostemu_minimizers_w7(const uint8_t *seq, uint32_t len, int k, int variant, uint64_t *out_hash, uint32_t *out_ps, uint32_t cap) {
  auto put = [&](uint32_t n, uint64_t h, uint32_t p) { if (n < cap) { out_hash[n] = h; out_ps[n] = p; } };
  return (int)(variant ? cm_minimizers_w7(seq, len, k, put) : cm_minimizers_window_e<7>(seq, len, k, put));
}

// one read through the runtime-window form (cm_minimizers_core<0>: any w <= CM_MAX_W)
extern "C" int hostemu_minimizers_ring(const uint8_t *seq, uint32_t len, int k, int w, uint64_t *out_hash, uint32_t *out_ps, uint32_t cap) {
  return (int)cm_minimizers_ring(seq, len, k, w, out_hash, out_ps, cap);
}

// S0 alone (length filter + adapter trimming): rlen[2*pair], rlen[2*pair+1] as cm_s0_prep leaves them
extern "C" int hostemu_trim(const cmgpu_params *params, const cmgpu_batch *in, uint32_t *rlen) {
  CmDev d;
  memset(&d, 0, sizeof(d));
  d.p.min_read_len = params->min_read_length;
  d.p.trim = params->trim_adapters;
  d.rb0 = (const uint8_t *)in->read1_bases; d.rb1 = (const uint8_t *)in->read2_bases;
  d.ro0 = in->read1_offsets; d.ro1 = in->read2_offsets;
  d.rlen = rlen;
  for (uint32_t i = 0; i < in->n_pairs; ++i) cm_s0_prep(d, i);
  return 0;
}

extern "C" int hostemu_map_pairs_sam(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                     const cmgpu_batch *in, cmgpu_sam_record *rec, uint32_t *cigar, char *md, uint32_t md_cap,
                                     cmgpu_stats *stats) {
  std::vector<cmgpu_record> out(in->n_pairs + 1);
  uint64_t k = 0;
  const EmuSam sam{rec, cigar, md, md_cap};
  return emu_map_pairs(index, ref, params, in, out.data(), &k, stats, nullptr, nullptr, nullptr, nullptr, nullptr, false, &sam);
}

// --SAM for single-cell data: SAM records + the per-pair (corrected) barcode keys the writer sorts on and prints as CB:Z
extern "C" int hostemu_map_pairs_bc_sam(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                        const cmgpu_batch *in, const cmgpu_barcode_batch *bc, const uint64_t *wl_keys, uint32_t n_keys,
                                        cmgpu_sam_record *rec, uint32_t *cigar, char *md, uint32_t md_cap, uint64_t *keys_per_pair,
                                        cmgpu_stats *stats) {
  std::vector<cmgpu_record> out(in->n_pairs + 1);
  std::vector<uint64_t> keys(in->n_pairs + 1);
  uint64_t k = 0;
  const EmuSam sam{rec, cigar, md, md_cap};
  EmuBarcodes eb{bc, wl_keys, n_keys, keys.data(), keys_per_pair};
  return emu_map_pairs(index, ref, params, in, out.data(), &k, stats, nullptr, nullptr, nullptr, nullptr, &eb, false, &sam);
}

extern "C" int hostemu_map_single_sam(const cmgpu_index_view *index, const cmgpu_ref_view *ref, const cmgpu_params *params,
                                      const cmgpu_single_batch *in, cmgpu_sam_record *rec, uint32_t *cigar, char *md, uint32_t md_cap,
                                      cmgpu_stats *stats) {
  std::vector<uint32_t> zero((size_t)in->n_reads + 1, 0);
  cmgpu_batch b{in->n_reads, in->first_read_id, in->bases, in->offsets, "", zero.data()};
  std::vector<cmgpu_record> out(in->n_reads + 1);
  uint64_t k = 0;
  const EmuSam sam{rec, cigar, md, md_cap};
  return emu_map_pairs(index, ref, params, &b, out.data(), &k, stats, nullptr, nullptr, nullptr, nullptr, nullptr, true, &sam);
}

// cm_hamming_diag against the plain loop it replaces (BandedTraceback's first step), for tests
extern "C" int hostemu_hamming(const uint8_t *pat, const uint8_t *read, int Lfull, int neg, int toff, int L, int *naive) {
  int c = 0;
  for (int i = 0; i < L; ++i) c += pat[i] != cm_text_char(read, Lfull, toff + i, neg != 0);
  *naive = c;
  return cm_hamming_diag(pat, read, Lfull, neg != 0, toff, L);
}

// cm_sweep_cluster over the local clusters of a sorted list against cm_sweep_strided (k_s3b_heavy clusters a long hit
// list with one lane per local cluster): returns 0 when the two candidate lists are equal
extern "C" int hostemu_sweep_clusters(const uint64_t *sorted, uint32_t n, int e, int seeds_required, uint32_t num_minimizers) {
  std::vector<uint64_t> a(sorted, sorted + n), oh(n + 1);
  std::vector<uint8_t> ac(n + 1), oc(n + 1);
  const uint32_t want = cm_sweep_strided(a.data(), ac.data(), n, e, seeds_required, num_minimizers, 1);
  uint32_t got = 0;
  for (uint32_t b = 0; b < n;) {
    uint32_t end = b + 1;
    while (end < n && !cm_sweep_local_break(sorted[end - 1], sorted[end], e)) ++end;
    const uint32_t c0 = cm_sweep_cluster(sorted, 1, b, end, e, seeds_required, num_minimizers, nullptr, nullptr);
    const uint32_t c1 = cm_sweep_cluster(sorted, 1, b, end, e, seeds_required, num_minimizers, oh.data() + got, oc.data() + got);
    if (c0 != c1) return 2;
    {  // the form that finds the cluster's end itself (cm_coop_sweep uses it)
      std::vector<uint64_t> fh(c1 + 1);
      std::vector<uint8_t> fc(c1 + 1);
      if (cm_sweep_cluster_from(sorted, b, n, e, seeds_required, num_minimizers, nullptr, nullptr) != c1) return 4;
      if (cm_sweep_cluster_from(sorted, b, n, e, seeds_required, num_minimizers, fh.data(), fc.data()) != c1) return 4;
      for (uint32_t q = 0; q < c1; ++q) if (fh[q] != oh[got + q] || fc[q] != oc[got + q]) return 5;
    }
    got += c1;
    b = end;
  }
  if (got != want) return 1;
  for (uint32_t i = 0; i < want; ++i) if (oh[i] != a[i] || oc[i] != ac[i]) return 3;
  return 0;
}

// one read through the position-parallel formulation of the w = 7 minimizer pass (what k_prep_flat does per lane):
// pack -> k-mer of every position -> hash -> sliding-extrema selection, sequential fallback where the kernel takes it.
// *path: 0 position-parallel, 1 fallback
extern "C" int hostemu_minimizers_flat(const uint8_t *seq, uint32_t len, int k, uint64_t *out_hash, uint32_t *out_ps, uint32_t cap, int *path) {
  auto put = [&](uint32_t n, uint64_t h, uint32_t p) { if (n < cap) { out_hash[n] = h; out_ps[n] = p; } };
  *path = 1;
  if (!(k & 1) || len < (uint32_t)k + 6) return (int)cm_minimizers_w7(seq, len, k, put);
  // pack with an arbitrary start offset inside the packed range, like a read somewhere in a block's staged bytes
  const uint32_t lead = 5;
  std::vector<uint8_t> raw(lead + len + 32, (uint8_t)'A');
  memcpy(raw.data() + lead, seq, len);
  std::vector<uint32_t> pk((raw.size() + 15) / 16 + 3, 0);
  bool bad = false;
  for (size_t w = 0; w * 16 + 16 <= raw.size(); ++w) {
    uint32_t word = 0;
    for (int q = 0; q < 4; ++q) {
      uint32_t in;
      memcpy(&in, raw.data() + w * 16 + q * 4, 4);
      uint32_t b4;
      word |= cm_mmf_pack4(in, &b4) << (8 * q);
      for (int b = 0; b < 4; ++b) {
        const size_t pos = w * 16 + q * 4 + b;
        if (((b4 >> b) & 1u) && pos >= lead && pos < lead + len) bad = true;
      }
    }
    pk[w] = word;
  }
  if (bad) return (int)cm_minimizers_w7(seq, len, k, put);
  const uint32_t m = len - (uint32_t)k + 1;
  std::vector<uint64_t> h(m), M(m);
  std::vector<uint8_t> st(m), fl(m);
  for (uint32_t i = 0; i < m; ++i) {
    uint32_t sd;
    h[i] = cm_mmf_hash(cm_mmf_kmer(pk.data(), lead + i, k), k, &sd);
    st[i] = (uint8_t)sd;
  }
  if (!cm_mmf_select(h.data(), m, M.data(), fl.data())) return (int)cm_minimizers_w7(seq, len, k, put);
  *path = 0;
  uint32_t n = 0;
  for (uint32_t i = 0; i < m; ++i)
    if (fl[i]) { put(n, h[i], ((i + (uint32_t)k - 1) << 1) | st[i]); ++n; }
  return (int)n;
}

// cm_coop_rescue_dir (sort + sweep + MergeCandidates by a group) against the one-lane sequence cm_sort_u64, cm_sweep,
// cm_merge on caller-made lists: c0 (strictly ascending positions + counts) and unsorted rescue hits.  Returns 0 when equal.
template <int G>
static int emu_rescue_dir_check(const uint64_t *c0p, const uint8_t *c0c, uint32_t n1, const uint64_t *hits, uint32_t cnt, int e, uint32_t nm,
                                uint32_t P, uint32_t RB, bool reverse) {
  CmDev d;
  memset(&d, 0, sizeof(d));
  d.p.e = e;
  uint32_t mmc[1] = {nm};
  d.mm_cnt = mmc;
  // sequential
  std::vector<uint64_t> so((size_t)n1 + cnt + 1);
  std::vector<uint8_t> soc((size_t)n1 + cnt + 1);
  for (uint32_t i = 0; i < cnt; ++i) so[n1 + i] = hits[i];
  uint32_t want;
  {
    cm_sort_u64(so.data() + n1, cnt);
    const uint32_t naug = cm_sweep(so.data() + n1, soc.data() + n1, cnt, e, 1, nm);
    if (naug > 0) want = cm_merge(c0p, c0c, n1, so.data(), soc.data(), naug, e);
    else { for (uint32_t i = 0; i < n1; ++i) { so[i] = c0p[i]; soc[i] = c0c[i]; } want = n1; }
  }
  // cooperative
  std::vector<uint64_t> go((size_t)n1 + cnt + 1), zp((size_t)n1 + cnt + 1);
  std::vector<uint8_t> goc((size_t)n1 + cnt + 1), zc((size_t)n1 + cnt + 1);
  for (uint32_t i = 0; i < cnt; ++i) go[n1 + i] = hits[i];
  std::vector<uint8_t> mem(cm_coop_mem_bytes(P, 4, RB, true) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  CmCoopMem m = cm_coop_mem_at(base, P, 4, RB, true);
  std::vector<uint8_t> slab(g_coop_slab ? cm_coop_slab_bytes(g_coop_slab) + 16 : 0);
  if (g_coop_slab) cm_coop_slab_at(m, slab.data() + ((16 - ((uintptr_t)slab.data() & 15)) & 15), g_coop_slab);
  uint32_t got = 0;
  emu_run_group<G>([&](EmuGroup<G> &g) {
    const uint32_t k = g_coop_slab && cnt > P ? cm_coop_rescue_dir<true>(d, 0, g, m, go.data(), goc.data(), n1, cnt, true, c0p, c0c, zp.data(), zc.data())
                                              : cm_coop_rescue_dir<false>(d, 0, g, m, go.data(), goc.data(), n1, cnt, true, c0p, c0c, zp.data(), zc.data());
    if (g.t == 0) got = k;
  }, reverse);
  if (got != want) return 1;
  for (uint32_t i = 0; i < want; ++i) if (go[i] != so[i] || goc[i] != soc[i]) return 2;
  return 0;
}
extern "C" int hostemu_rescue_dir_check(const uint64_t *c0p, const uint8_t *c0c, uint32_t n1, const uint64_t *hits, uint32_t cnt, int e, uint32_t nm,
                                        int G, uint32_t P, uint32_t RB, int reverse) {
  if (G == 16) return emu_rescue_dir_check<16>(c0p, c0c, n1, hits, cnt, e, nm, P, RB, reverse != 0);
  if (G == 64) return emu_rescue_dir_check<64>(c0p, c0c, n1, hits, cnt, e, nm, P, RB, reverse != 0);
  return emu_rescue_dir_check<256>(c0p, c0c, n1, hits, cnt, e, nm, P, RB, reverse != 0);
}


// cm_coop_rescue against cm_rescue on made-up occurrence runs and mate candidates: runs of 1 .. 3000 occurrences (random, or
// evenly spaced as in a satellite array), singletons and misses between them, 1 .. 40 minimizers (several rounds of the pair table),
// 1 .. 420 mate candidates of which 1 .. 350 are the best ones (at 300 the search bails out), windows that touch, nest or start
// exactly on an occurrence (the search's "equal" exit), occurrences on both strands.  Returns the number of cases that differ.
static unsigned long long g_rescue_check_stats[4];  // searches with hits, hits, bail-outs, searches with 64 best candidates or more
extern "C" void hostemu_rescue_search_stats(unsigned long long *out) { memcpy(out, g_rescue_check_stats, sizeof(g_rescue_check_stats)); }
extern "C" unsigned long long hostemu_rescue_replays(void) { return g_dbg_rescue_replays; }
extern "C" void hostemu_force_rescue_replay(int on) { g_dbg_force_replay = on; }
template <int G>
static int emu_rescue_search_check(uint64_t seed, uint32_t rounds, bool reverse) {
  uint64_t st = seed * 0x9E3779B97F4A7C15ull + 11;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  int bad = 0;
  for (uint32_t it = 0; it < rounds; ++it) {
    CmDev d;
    memset(&d, 0, sizeof(d));
    d.p.k = 17; d.p.max_insert = (int)(20 + rnd() % 1500); d.p.f0 = 500; d.p.min_seeds = 2;
    const uint32_t n = 1 + (uint32_t)(rnd() % (it % 5 == 0 ? 40 : 10));
    const uint32_t nrid = 1 + (uint32_t)(rnd() % 3);
    const uint32_t span = (uint32_t)(2000 + rnd() % (it % 3 == 0 ? 60000 : 4000000));
    std::vector<uint64_t> occ;
    std::vector<uint8_t> kind(n);
    std::vector<uint64_t> val(n);
    std::vector<uint32_t> mps(n);
    for (uint32_t mi = 0; mi < n; ++mi) {
      mps[mi] = (uint32_t)(rnd() % 34) << 1 | (uint32_t)(rnd() & 1);
      const uint32_t kk = (uint32_t)(rnd() % 8);
      if (kk == 0) { kind[mi] = CM_PR_MISS; val[mi] = 0; continue; }
      if (kk == 1) { kind[mi] = CM_PR_SINGLE; val[mi] = ((uint64_t)(rnd() % nrid) << 33) | ((uint64_t)(rnd() % span) << 1) | (rnd() & 1); continue; }
      const uint32_t nocc = 2 + (uint32_t)(rnd() % (it % 4 == 0 ? 3000 : 300));
      std::vector<uint64_t> run;
      if (rnd() % 3 == 0) {  // evenly spaced
        const uint32_t step = 1 + (uint32_t)(rnd() % 200), start = (uint32_t)(rnd() % 1000);
        for (uint32_t i = 0; i < nocc; ++i) run.push_back(((uint64_t)(i * nrid / nocc) << 33) | ((uint64_t)(start + i * step) << 1) | (rnd() & 1));
      } else {
        for (uint32_t i = 0; i < nocc; ++i) run.push_back(((uint64_t)(rnd() % nrid) << 33) | ((uint64_t)(rnd() % span) << 1) | (rnd() & 1));
      }
      std::sort(run.begin(), run.end());
      // one strand per position, as the reference's index has it (two occurrences at one position, one per strand, are kept one time in eight)
      std::vector<uint64_t> u;
      for (uint64_t x : run) if (u.empty() || (u.back() >> 1) != (x >> 1) || (it % 8 == 0 && u.back() != x)) u.push_back(x);
      kind[mi] = CM_PR_MULTI;
      val[mi] = ((uint64_t)occ.size() << 32) | (uint32_t)u.size();
      occ.insert(occ.end(), u.begin(), u.end());
    }
    occ.push_back(0);
    const uint32_t mn = 1 + (uint32_t)(rnd() % (it % 6 == 0 ? 420 : 60));
    std::vector<uint64_t> mp(mn);
    std::vector<uint8_t> mc(mn);
    const uint64_t sr = 2ull * (uint64_t)d.p.max_insert;
    for (uint32_t i = 0; i < mn; ++i) {
      uint64_t pos = ((uint64_t)(rnd() % nrid) << 32) | (rnd() % span);
      if (rnd() % 3 == 0 && occ.size() > 1) {  // a window that starts exactly on an occurrence
        const uint64_t o = occ[rnd() % (occ.size() - 1)] >> 1;
        pos = o + sr;
      }
      mp[i] = pos;
    }
    std::sort(mp.begin(), mp.end());
    const uint32_t top = 2 + (uint32_t)(rnd() % 6);
    const uint32_t share = (uint32_t)(rnd() % 4);  // how many get the best count: few .. all
    for (uint32_t i = 0; i < mn; ++i) mc[i] = (uint8_t)((share == 3 || rnd() % (share + 2) == 0) ? top : 1 + rnd() % (top - 1));
    uint32_t mm_off[2] = {0, n}, mm_cnt[1] = {n};
    d.occ = occ.data(); d.n_occ = (uint32_t)occ.size();
    d.pr_kind = kind.data(); d.pr_val = val.data(); d.mm_ps = mps.data(); d.mm_off = mm_off; d.mm_cnt = mm_cnt;
    for (int strand = 0; strand < 2; ++strand) {
      uint32_t c1 = 0, rl1 = 0, c2 = 0, rl2 = 0;
      const int r1 = cm_rescue(d, 0, strand, mp.data(), mc.data(), mn, nullptr, &c1, &rl1, nullptr);
      {
        uint32_t nbest = 0;
        for (uint32_t i = 0; i < mn; ++i) nbest += mc[i] == (uint8_t)(r1 < 0 ? -r1 : r1) ? 1u : 0u;
        g_rescue_check_stats[0] += c1 > 0; g_rescue_check_stats[1] += c1; g_rescue_check_stats[2] += r1 < 0; g_rescue_check_stats[3] += nbest >= 64 && r1 >= 0;
      }
      std::vector<uint64_t> o1(c1 + 1), o2(c1 + 1);
      if (r1 >= 0) (void)cm_rescue(d, 0, strand, mp.data(), mc.data(), mn, o1.data(), &c1, &rl1, nullptr);
      std::vector<uint64_t> mem(cm_coop_rescue_mem_bytes() / 8 + 4);
      const CmCoopRescueMem m = cm_coop_rescue_mem_at((uint8_t *)mem.data());
      int r2 = 0;
      emu_run_group<G>([&](EmuGroup<G> &g) {
        uint32_t c = 0, rl = 0;
        const int rr = cm_coop_rescue(d, 0, strand, mp.data(), mc.data(), mn, g, m, nullptr, &c, &rl);
        if (g.t == (uint32_t)(G - 1)) { r2 = rr; c2 = c; rl2 = rl; }
      }, reverse);
      bool differ = r1 != r2 || (r1 >= 0 && (c1 != c2 || rl1 != rl2));
      if (!differ && r1 >= 0 && c1 > 0) {
        emu_run_group<G>([&](EmuGroup<G> &g) {
          uint32_t c = 0, rl = 0;
          (void)cm_coop_rescue(d, 0, strand, mp.data(), mc.data(), mn, g, m, o2.data(), &c, &rl);
        }, reverse);
        for (uint32_t i = 0; i < c1; ++i) if (o1[i] != o2[i]) { differ = true; break; }
      }
      if (differ) {
        if (bad < 5) fprintf(stderr, "rescue search: case %u strand %d: one lane (%d, %u, %u) group (%d, %u, %u), %u minimizers, %u mate candidates\n", it, strand, r1, c1, rl1, r2, c2, rl2, n, mn);
        ++bad;
      }
    }
  }
  return bad;
}
extern "C" int hostemu_rescue_search_check(uint64_t seed, uint32_t rounds, int G, int reverse) {
  if (G == 16) return emu_rescue_search_check<16>(seed, rounds, reverse != 0);
  if (G == 64) return emu_rescue_search_check<64>(seed, rounds, reverse != 0);
  return emu_rescue_search_check<256>(seed, rounds, reverse != 0);
}

// cm_coop_reduce_dir against cm_reduce_dir on caller-made ascending lists.  Returns 0 when equal.
template <int G>
static int emu_reduce_dir_check(uint32_t dist, const uint64_t *p1, const uint8_t *c1, uint32_t n1, const uint64_t *p2, const uint8_t *c2, uint32_t n2,
                                bool reverse) {
  std::vector<uint64_t> sf1(n1 + 1), sf2(n2 + 1), gf1(n1 + 1), gf2(n2 + 1);
  std::vector<uint8_t> sc1(n1 + 1), sc2(n2 + 1), gc1(n1 + 1), gc2(n2 + 1);
  uint32_t sa, sb, ga = 0, gb = 0;
  cm_reduce_dir(dist, p1, c1, n1, p2, c2, n2, sf1.data(), sc1.data(), &sa, sf2.data(), sc2.data(), &sb);
  const uint32_t P = (n1 > n2 ? n1 : n2) + 1;
  std::vector<uint8_t> mem(cm_coop_pair_mem_bytes(P) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  const CmCoopPairMem m = cm_coop_pair_mem_at(base, P);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    uint32_t a, b;
    if (n1 & 1u) cm_coop_reduce_dir<false>(g, m, dist, p1, c1, n1, p2, c2, n2, gf1.data(), gc1.data(), &a, gf2.data(), gc2.data(), &b);
    else cm_coop_reduce_dir<true>(g, m, dist, p1, c1, n1, p2, c2, n2, gf1.data(), gc1.data(), &a, gf2.data(), gc2.data(), &b);
    if (g.t == 0) { ga = a; gb = b; }
  }, reverse);
  if (ga != sa || gb != sb) return 1;
  for (uint32_t i = 0; i < sa; ++i) if (gf1[i] != sf1[i] || gc1[i] != sc1[i]) return 2;
  for (uint32_t i = 0; i < sb; ++i) if (gf2[i] != sf2[i] || gc2[i] != sc2[i]) return 3;
  return 0;
}
extern "C" int hostemu_reduce_dir_check(uint32_t dist, const uint64_t *p1, const uint8_t *c1, uint32_t n1, const uint64_t *p2, const uint8_t *c2,
                                        uint32_t n2, int G, int reverse) {
  if (G == 16) return emu_reduce_dir_check<16>(dist, p1, c1, n1, p2, c2, n2, reverse != 0);
  if (G == 64) return emu_reduce_dir_check<64>(dist, p1, c1, n1, p2, c2, n2, reverse != 0);
  return emu_reduce_dir_check<256>(dist, p1, c1, n1, p2, c2, n2, reverse != 0);
}

// cm_coop_draft_strand against cm_draft_strand (the acceptance loop over precomputed alignments) on caller-made candidate
// lists (sorted by count descending): returns 0 when the draft mappings and the best / second-best bookkeeping are equal
template <int G>
static int emu_draft_strand_check(const uint64_t *cp, const uint8_t *cc, uint32_t nc, const int16_t *pre_err, const int16_t *pre_end, uint32_t L,
                                  int strand, int e, int lanes, const uint32_t *ref_len, uint32_t n_seq, bool reverse) {
  CmDev d;
  memset(&d, 0, sizeof(d));
  d.p.e = e; d.p.lanes = lanes;
  d.ref_len = ref_len; d.n_seq = n_seq;
  std::vector<uint64_t> sdp(nc + 1), gdp(nc + 1);
  std::vector<int16_t> sde(nc + 1), gde(nc + 1);
  CmBest sb = {e + 1, e + 1, 0, 0};
  const uint32_t want = cm_draft_strand(d, nullptr, L, strand, cp, cc, nc, sb, sdp.data(), sde.data(), pre_err, pre_end);
  const uint32_t P = nc + 1;
  std::vector<uint8_t> mem(cm_coop_ver_mem_bytes(P) + 16);
  uint8_t *base = mem.data() + ((16 - ((uintptr_t)mem.data() & 15)) & 15);
  const CmCoopVerMem m = cm_coop_ver_mem_at(base, P);
  uint32_t got = 0;
  CmTwo gb = {e + 1, 0, e + 1, 0};
  emu_run_group<G>([&](EmuGroup<G> &g) {
    CmTwo b = {e + 1, 0, e + 1, 0};
    const uint32_t k = cm_coop_draft_strand(d, g, m, L, strand, cp, cc, nc, b, gdp.data(), gde.data(), pre_err, pre_end);
    if (g.t == 0) { got = k; gb = b; }
  }, reverse);
  if (got != want) return 1;
  for (uint32_t i = 0; i < want; ++i) if (gdp[i] != sdp[i] || gde[i] != sde[i]) return 2;
  if (gb.lo != sb.min_err || gb.n_lo != sb.n_best || gb.hi != sb.second_err || gb.n_hi != sb.n_second) return 3;
  return 0;
}
extern "C" int hostemu_draft_strand_check(const uint64_t *cp, const uint8_t *cc, uint32_t nc, const int16_t *pre_err, const int16_t *pre_end, uint32_t L,
                                          int strand, int e, int lanes, const uint32_t *ref_len, uint32_t n_seq, int G, int reverse) {
  if (G == 16) return emu_draft_strand_check<16>(cp, cc, nc, pre_err, pre_end, L, strand, e, lanes, ref_len, n_seq, reverse != 0);
  if (G == 64) return emu_draft_strand_check<64>(cp, cc, nc, pre_err, pre_end, L, strand, e, lanes, ref_len, n_seq, reverse != 0);
  return emu_draft_strand_check<256>(cp, cc, nc, pre_err, pre_end, L, strand, e, lanes, ref_len, n_seq, reverse != 0);
}

// cm_coop_pair_dir (both directions, merged) against two cm_pair_dir sweeps: returns 0 when the best / second-best sums, their
// multiplicities and the first best pairing are equal
template <int G>
static int emu_pairing_check(const uint64_t *ap0, const int16_t *ae0, uint32_t na0, const uint64_t *bp0, const int16_t *be0, uint32_t nb0,
                             const uint64_t *ap1, const int16_t *ae1, uint32_t na1, const uint64_t *bp1, const int16_t *be1, uint32_t nb1,
                             uint32_t len1, uint32_t len2, int e, int max_insert, int min_read_len, bool reverse) {
  CmDev d;
  memset(&d, 0, sizeof(d));
  d.p.e = e; d.p.max_insert = max_insert; d.p.min_read_len = min_read_len;
  CmPe pe;
  pe.min_sum = 2 * e + 1; pe.second_sum = 2 * e + 1; pe.n_best = 0; pe.n_second = 0; pe.f_dir = 0; pe.f_i1 = 0; pe.f_i2 = 0;
  int64_t seen = 0;
  cm_pair_dir(d, 0, ap0, ae0, na0, bp0, be0, nb0, len1, len2, pe, -1, 0, &seen);
  cm_pair_dir(d, 1, ap1, ae1, na1, bp1, be1, nb1, len1, len2, pe, -1, 0, &seen);
  CmTwo all = {0, 0, 0, 0};
  uint64_t fk = 0;
  std::vector<uint64_t> stp(100);
  std::vector<int16_t> ste(100);
  emu_run_group<G>([&](EmuGroup<G> &g) {
    const int none = 2 * e + 1;
    CmTwo mine = {none, 0, none, 0};
    uint64_t first_key = ~0ull;
    // second lists of up to 100 entries staged in the group's work arrays (k_s6a_coop), longer ones read where they are
    if (nb0 <= 100) cm_coop_pair_dir<true>(d, g, 0, ap0, ae0, na0, bp0, be0, nb0, len1, len2, mine, first_key, stp.data(), ste.data());
    else cm_coop_pair_dir<false>(d, g, 0, ap0, ae0, na0, bp0, be0, nb0, len1, len2, mine, first_key);
    if (nb1 <= 100) cm_coop_pair_dir<true>(d, g, 1, ap1, ae1, na1, bp1, be1, nb1, len1, len2, mine, first_key, stp.data(), ste.data());
    else cm_coop_pair_dir<false>(d, g, 1, ap1, ae1, na1, bp1, be1, nb1, len1, len2, mine, first_key);
    const CmTwo a = cm_coop_two_merge(g, mine, none);
    const uint64_t k = g.min64(first_key);
    if (g.t == 0) { all = a; fk = k; }
  }, reverse);
  if (all.lo != pe.min_sum || all.n_lo != pe.n_best || all.hi != pe.second_sum || all.n_hi != pe.n_second) return 1;
  if (pe.n_best > 0) {
    if (fk == ~0ull) return 2;
    if (((uint32_t)(fk >> 48) & 1u) != pe.f_dir || ((uint32_t)(fk >> 24) & 0xffffffu) != pe.f_i1 || ((uint32_t)fk & 0xffffffu) != pe.f_i2) return 3;
  } else if (fk != ~0ull) return 4;
  // the want-th minimal-sum pairing in sweep order (what the sampler of a multi-mapped pair asks for): cm_coop_pair_find against the
  // sweep that counts up to it -- every index when there are few, the ends and a spread of the others when there are many
  const int nbest = pe.n_best;
  for (int step = 0; step < nbest; ++step) {
    const int want = nbest <= 48 ? step : (step < 16 ? step : (step < 32 ? nbest - 1 - (step - 16) : (int)(((uint64_t)step * 2654435761u) % (uint64_t)nbest)));
    if (nbest > 48 && step >= 64) break;
    CmPe q = pe;
    int64_t s2 = 0;
    bool found = cm_pair_dir(d, 0, ap0, ae0, na0, bp0, be0, nb0, len1, len2, q, want, pe.min_sum, &s2);
    if (!found) found = cm_pair_dir(d, 1, ap1, ae1, na1, bp1, be1, nb1, len1, len2, q, want, pe.min_sum, &s2);
    if (!found) return 5;
    uint32_t gd = 9, gi1 = 0, gi2 = 0;
    bool gfound = false;
    emu_run_group<G>([&](EmuGroup<G> &g) {
      uint64_t seen2 = 0;
      uint32_t i1 = 0, i2 = 0;
      int dir = 0;
      bool fnd = nb0 <= 100 ? cm_coop_pair_find<true>(d, g, 0, ap0, ae0, na0, bp0, be0, nb0, len1, len2, pe.min_sum, (uint64_t)want, &seen2, &i1, &i2, stp.data(), ste.data())
                            : cm_coop_pair_find<false>(d, g, 0, ap0, ae0, na0, bp0, be0, nb0, len1, len2, pe.min_sum, (uint64_t)want, &seen2, &i1, &i2);
      if (!fnd) {
        dir = 1;
        fnd = nb1 <= 100 ? cm_coop_pair_find<true>(d, g, 1, ap1, ae1, na1, bp1, be1, nb1, len1, len2, pe.min_sum, (uint64_t)want, &seen2, &i1, &i2, stp.data(), ste.data())
                         : cm_coop_pair_find<false>(d, g, 1, ap1, ae1, na1, bp1, be1, nb1, len1, len2, pe.min_sum, (uint64_t)want, &seen2, &i1, &i2);
      }
      if (g.t == (uint32_t)(G - 1)) { gfound = fnd; gd = (uint32_t)dir; gi1 = i1; gi2 = i2; }  // (every lane holds the answer: the last one reports)
    }, reverse);
    if (!gfound || gd != q.f_dir || gi1 != q.f_i1 || gi2 != q.f_i2) return 6;
  }
  return 0;
}
extern "C" int hostemu_pairing_check(const uint64_t *ap0, const int16_t *ae0, uint32_t na0, const uint64_t *bp0, const int16_t *be0, uint32_t nb0,
                                     const uint64_t *ap1, const int16_t *ae1, uint32_t na1, const uint64_t *bp1, const int16_t *be1, uint32_t nb1,
                                     uint32_t len1, uint32_t len2, int e, int max_insert, int min_read_len, int G, int reverse) {
  if (G == 16) return emu_pairing_check<16>(ap0, ae0, na0, bp0, be0, nb0, ap1, ae1, na1, bp1, be1, nb1, len1, len2, e, max_insert, min_read_len, reverse != 0);
  if (G == 64) return emu_pairing_check<64>(ap0, ae0, na0, bp0, be0, nb0, ap1, ae1, na1, bp1, be1, nb1, len1, len2, e, max_insert, min_read_len, reverse != 0);
  return emu_pairing_check<256>(ap0, ae0, na0, bp0, be0, nb0, ap1, ae1, na1, bp1, be1, nb1, len1, len2, e, max_insert, min_read_len, reverse != 0);
}

// cm_coop_sort_cand against cm_sort_cand on caller-made candidate lists.  Returns 0 when equal.
template <int G>
static int emu_sort_cand_check(const uint64_t *p, const uint8_t *c, uint32_t n, uint32_t nb_cap, bool reverse) {
  std::vector<uint64_t> sp(p, p + n), gp(p, p + n), tp(n + 1);
  std::vector<uint8_t> sc(c, c + n), gc(c, c + n), tc(n + 1);
  cm_sort_cand(sp.data(), sc.data(), n);
  std::vector<uint16_t> hist((size_t)G * nb_cap);
  emu_run_group<G>([&](EmuGroup<G> &g) { cm_coop_sort_cand(g, gp.data(), gc.data(), n, tp.data(), tc.data(), hist.data(), nb_cap); }, reverse);
  for (uint32_t i = 0; i < n; ++i) if (gp[i] != sp[i] || gc[i] != sc[i]) return 1;
  // the same with the list staged in (what stands for) shared memory: twice in a row through the same staging arrays, the way
  // k_s5_sort_coop sorts a read's two strands; a staging area of n - 1 entries must leave the list to the scratch path
  for (uint32_t cap : {n, n + 7, n > 1 ? n - 1 : 0u}) {
    std::vector<uint64_t> lp(cap + 1);
    std::vector<uint8_t> lc(cap + 1);
    for (int round = 0; round < 2; ++round) {
      std::vector<uint64_t> hp(p, p + n);
      std::vector<uint8_t> hc(c, c + n);
      emu_run_group<G>([&](EmuGroup<G> &g) { cm_coop_sort_cand(g, hp.data(), hc.data(), n, tp.data(), tc.data(), hist.data(), nb_cap, lp.data(), lc.data(), cap); }, reverse);
      for (uint32_t i = 0; i < n; ++i) if (hp[i] != sp[i] || hc[i] != sc[i]) return 2 + round;
    }
  }
  return 0;
}
extern "C" int hostemu_sort_cand_check(const uint64_t *p, const uint8_t *c, uint32_t n, uint32_t nb_cap, int G, int reverse) {
  if (G == 16) return emu_sort_cand_check<16>(p, c, n, nb_cap, reverse != 0);
  if (G == 64) return emu_sort_cand_check<64>(p, c, n, nb_cap, reverse != 0);
  return emu_sort_cand_check<256>(p, c, n, nb_cap, reverse != 0);
}

// cm_banded_align_planes against cm_banded_align (the byte form) on `rounds` random cases: a reference of ref_len bytes with
// letters of both cases and a share of other bytes, reads cut from it with substitutions / insertions / deletions / Ns, both
// strands, every window offset modulo 32, read lengths 1..max_len, error thresholds 1..max_e.  Returns the number of cases in
// which (distance, end position) differ.
extern "C" int hostemu_align_planes_check(uint64_t seed, uint32_t rounds, uint32_t max_len, int max_e) {
  uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  const uint32_t ref_len = 4096;
  std::vector<uint64_t> refw(ref_len / 8 + 8, 0);
  uint8_t *ref = (uint8_t *)refw.data();
  const char *alpha = "ACGTacgtACGTACGTNnRX";
  for (uint32_t i = 0; i < ref_len; ++i) ref[i] = (uint8_t)alpha[rnd() % 20];
  const uint64_t rw = ref_len / 32 + 4;
  std::vector<CmPlRec> rpv((size_t)rw + CM_PL_LEAD, CmPlRec{0u, 0u, 0u, 0u});
  CmPlRec *rp = rpv.data() + CM_PL_LEAD;
  for (uint32_t w = 0; w * 32 < ref_len; ++w) cm_pack_planes32(ref + 32 * w, ref_len - 32 * w, &rp[w].p0, &rp[w].p1, &rp[w].pn, &rp[w].pc);
  int bad = 0;
  for (uint32_t it = 0; it < rounds; ++it) {
    const int e = 1 + (int)(rnd() % (uint64_t)max_e);
    const uint32_t L = 1 + (uint32_t)(rnd() % max_len);
    const uint32_t g = (uint32_t)(rnd() % (ref_len - L - 2 * (uint32_t)e - 40));
    const int strand = (int)(rnd() & 1);
    // the read: the window's middle with edits (or something unrelated one time in eight)
    std::vector<uint64_t> rdw((L + 31) / 8 + 3, 0);
    uint8_t *fw = (uint8_t *)rdw.data();
    const bool unrelated = rnd() % 8 == 0;
    uint32_t src = g + (uint32_t)e + (uint32_t)(rnd() % 3) - 1;
    for (uint32_t i = 0; i < L; ++i) {
      uint8_t c = unrelated ? (uint8_t)"ACGT"[rnd() % 4] : ref[src < ref_len ? src : ref_len - 1];
      const uint64_t x = rnd() % 64;
      if (x == 0) c = (uint8_t)"ACGT"[rnd() % 4];
      else if (x == 1) { ++src; }
      else if (x == 2 && src > 0) { --src; }
      else if (x == 3) c = 'N';
      else if (x == 4) c = (uint8_t)(c | 0x20);
      fw[i] = c;
      ++src;
    }
    // the read as the batch holds it: for the - strand the stored read is the reverse complement of the text
    std::vector<uint64_t> stw((L + 31) / 8 + 3, 0);
    uint8_t *stored = (uint8_t *)stw.data();
    for (uint32_t i = 0; i < L; ++i) stored[i] = strand ? cm_negchar(fw[L - 1 - i]) : fw[i];
    int end_a = (int)L, end_b = (int)L;
    const int na = cm_banded_align(e, ref + g, stored, (int)L, strand == 1, 0, (int)L, &end_a);
    // planes of the stored read, both orientations, through the product's packer
    CmDev d;
    memset(&d, 0, sizeof(d));
    uint32_t rlen[2] = {L, 0};
    uint32_t ro[3] = {0, L, L};
    d.rlen = rlen; d.rb0 = stored; d.rb1 = stored; d.ro0 = ro; d.ro1 = ro;
    const uint32_t W = (L + 31) / 32;
    std::vector<uint32_t> tp((size_t)cm_read_pl_stride(W) + 8, 0x5A5A5A5Au);
    d.read_pl = tp.data(); d.read_pl_w = W;
    cm_pack_read_planes(d, 0);
    const int nb = cm_banded_align_planes(e, rp, g, tp.data() + (size_t)strand * 3 * W, W, (int)L, &end_b);
    if (na != nb || end_a != end_b) {
      if (bad < 5) fprintf(stderr, "align planes: case %u e %d L %u g %u strand %d: bytes (%d, %d) planes (%d, %d)\n", it, e, L, g, strand, na, end_a, nb, end_b);
      ++bad;
    }
  }
  return bad;
}

// cm_banded_align_dropoff_planes against cm_banded_align_dropoff in the four shapes cm_draft_strand_split calls it in (+ strand:
// whole read / read without its first `allow` bases; - strand from the 3' end: whole read / without the last `allow` bases of the
// reverse complement).  Reads are cut from the reference with edits, a share of them chimeric (the second half from elsewhere:
// the drop-off case).  Returns the number of cases in which (distance, end position, mapped length) differ.
extern "C" int hostemu_dropoff_planes_check(uint64_t seed, uint32_t rounds, uint32_t max_len, int max_e) {
  uint64_t st = seed * 0x9E3779B97F4A7C15ull + 7;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  const uint32_t ref_len = 8192;
  std::vector<uint64_t> refw(ref_len / 8 + 8, 0);
  uint8_t *ref = (uint8_t *)refw.data();
  const char *alpha = "ACGTacgtACGTACGTACGTACGTACGTNnRX";
  for (uint32_t i = 0; i < ref_len; ++i) ref[i] = (uint8_t)alpha[rnd() % 32];
  const uint64_t rw = ref_len / 32 + 4;
  std::vector<CmPlRec> rpv((size_t)rw + CM_PL_LEAD, CmPlRec{0u, 0u, 0u, 0u});
  CmPlRec *rp = rpv.data() + CM_PL_LEAD;
  for (uint32_t w = 0; w * 32 < ref_len; ++w) cm_pack_planes32(ref + 32 * w, ref_len - 32 * w, &rp[w].p0, &rp[w].p1, &rp[w].pn, &rp[w].pc);
  int bad = 0;
  for (uint32_t it = 0; it < rounds; ++it) {
    const int e = 1 + (int)(rnd() % (uint64_t)max_e);
    const uint32_t L = 25 + (uint32_t)(rnd() % (max_len - 24));
    // one case in four at the very start of the reference buffer: the backward (- strand) windows then reach below record 0
    // (CM_PL_LEAD zero records stand there; round-3 advice: those loads used to fall in front of the allocation)
    const uint32_t g = rnd() % 4 == 0 ? (uint32_t)(rnd() % 96) : 128 + (uint32_t)(rnd() % (ref_len - L - 2 * (uint32_t)e - 400));
    const int strand = (int)(rnd() & 1);
    const int allow = (int)(rnd() % 2) ? 20 - e : 0;  // 0: the whole-read call
    if (allow < 0 || (uint32_t)allow + 2 >= L) continue;
    std::vector<uint64_t> fww((L + 31) / 8 + 3, 0);
    uint8_t *fw = (uint8_t *)fww.data();  // the text string (for the - strand: the reverse complement of the stored read)
    const uint32_t brk = rnd() % 3 == 0 ? (uint32_t)(rnd() % L) : L;  // chimeric from here on
    const uint32_t brk_lo = rnd() % 2 ? brk : 0, brk_hi = brk_lo ? L : brk;  // ... or up to here
    uint32_t src = g + (uint32_t)e;
    for (uint32_t i = 0; i < L; ++i) {
      const bool foreign = brk < L && i >= brk_lo && i < brk_hi;
      uint8_t c = foreign ? (uint8_t)"ACGT"[rnd() % 4] : ref[src];
      const uint64_t x = rnd() % 80;
      if (x == 0) c = (uint8_t)"ACGT"[rnd() % 4];
      else if (x == 1) ++src;
      else if (x == 2) --src;
      else if (x == 3) c = 'N';
      fw[i] = c;
      ++src;
    }
    std::vector<uint64_t> stw((L + 31) / 8 + 3, 0);
    uint8_t *stored = (uint8_t *)stw.data();
    for (uint32_t i = 0; i < L; ++i) stored[i] = strand ? cm_negchar(fw[L - 1 - i]) : fw[i];
    CmDev d;
    memset(&d, 0, sizeof(d));
    uint32_t rlen[2] = {L, 0};
    uint32_t ro[3] = {0, L, L};
    d.rlen = rlen; d.rb0 = stored; d.rb1 = stored; d.ro0 = ro; d.ro1 = ro;
    const uint32_t W = (L + 31) / 32;
    std::vector<uint32_t> tp((size_t)cm_read_pl_stride(W) + 8, 0x5A5A5A5Au);
    d.read_pl = tp.data(); d.read_pl_w = W;
    cm_pack_read_planes(d, 0);
    int ea = (int)L, eb = (int)L, la = 0, lb = 0, na, nb;
    const uint8_t *pat = ref + g;
    if (strand == 0) {
      na = cm_banded_align_dropoff(e, pat + allow, stored, (int)L, false, allow, (int)L - allow, false, &ea, &la);
      nb = cm_banded_align_dropoff_planes<false>(e, rp, (uint64_t)g + (uint32_t)allow, tp.data(), W, (uint32_t)allow, (int)L - allow, &eb, &lb);
    } else {
      na = cm_banded_align_dropoff(e, pat, stored, (int)L, true, 0, (int)L - allow, true, &ea, &la);
      nb = cm_banded_align_dropoff_planes<true>(e, rp, (uint64_t)g + (L - (uint32_t)allow) + 2 * (uint32_t)e - 1, tp.data(), W, (uint32_t)allow, (int)L - allow, &eb, &lb);
    }
    if (na != nb || ea != eb || la != lb) {
      if (bad < 5) fprintf(stderr, "dropoff planes: case %u e %d L %u g %u strand %d allow %d: bytes (%d, %d, %d) planes (%d, %d, %d)\n", it, e, L, g, strand, allow, na, ea, la, nb, eb, lb);
      ++bad;
    }
  }
  return bad;
}


// The device's BGZF inflate (cm_inflate.h) on the host: in[0 .. n_in) -> out[0 .. n_out).  Pass 1 by one lane with the device's table
// stride (64 lanes' entries interleaved) to exercise the same indexing: literals to `out`, matches to tokens; pass 2 by an emulated
// wave on a copy of the text (the device's window in LDS): the matches, then the CRC.  Returns the decoder's code.
uint32_t g_inflate_steps = 2048;
extern "C" void hostemu_inflate_steps(uint32_t n) { g_inflate_steps = n; }
extern "C" int hostemu_inflate_block(const uint8_t *in, uint32_t n_in, uint8_t *out, uint32_t n_out, uint32_t want_crc, uint32_t lane) {
  static uint32_t crc_tab[256];
  static CmCrcX2n x2n;
  if (!crc_tab[1]) { for (uint32_t i = 0; i < 256; ++i) crc_tab[i] = cm_crc32_entry(i); cm_crc_x2n_table(x2n); }
  const uint32_t stride = 64;
  std::vector<uint16_t> sym((size_t)CM_INF_SYMS * stride, 0xABCD);
  std::vector<uint8_t> len8((size_t)CM_INF_LENS * stride, 0xEE);
  std::vector<int16_t> delta((size_t)32 * stride, 0x7A7A);
  const uint32_t cap = cm_inf_tok_cap(n_out);
  std::vector<uint32_t> tok((size_t)cap + 2, 0xDEADBEEFu);
  uint32_t n_tok = 0;
  int rc = cm_inflate_tokens(in, n_in, out, n_out, tok.data() + 1, &n_tok, true, sym.data() + (lane & 63u), len8.data() + (lane & 63u), delta.data() + (lane & 63u), stride, g_inflate_steps);
  // the other lanes' entries and the words around the tokens must be untouched
  for (size_t i = 0; i < sym.size(); ++i) if ((i & 63u) != (lane & 63u) && sym[i] != 0xABCD) return 100;
  for (size_t i = 0; i < len8.size(); ++i) if ((i & 63u) != (lane & 63u) && len8[i] != 0xEE) return 101;
  for (size_t i = 0; i < delta.size(); ++i) if ((i & 63u) != (lane & 63u) && delta[i] != 0x7A7A) return 104;
  if (tok[0] != 0xDEADBEEFu || tok[(size_t)cap + 1] != 0xDEADBEEFu || n_tok > cap) return 102;
  if (rc != CM_INF_OK) return rc;
  std::vector<uint8_t> win((size_t)n_out + 16, 0xC3);  // (the 16 bytes behind the text: the resolver's pad)
  memcpy(win.data(), out, n_out);
  std::vector<uint32_t> ends(128), red(64);
  int rc2 = 0;
  uint32_t crc = 0;
  emu_run_group<64>([&](EmuGroup<64> &g) {
    const int r = cm_bgzf_resolve(g, win.data(), tok.data() + 1, n_tok, n_out, ends.data());
    const uint32_t c = cm_bgzf_crc(g, win.data(), n_out, crc_tab, x2n.v, red.data());
    if (g.t == 0) { rc2 = r; crc = c; }
  }, (lane & 1u) != 0);
  if (rc2 != CM_INF_OK) return rc2;
  if (crc != want_crc) return CM_INF_ECRC;
  memcpy(out, win.data(), n_out);
  return CM_INF_OK;
}

// the second pass alone on given tokens (tests: token streams no compressor writes -- chains of overlapping matches, every length
// and distance): win[0 .. isize) holds the literals, the matches' bytes anything.  `reverse`: the emulation's other lane order.
extern "C" int hostemu_bgzf_resolve(uint8_t *win_io, uint32_t isize, const uint32_t *tok, uint32_t n_tok, int reverse) {
  std::vector<uint8_t> win((size_t)isize + 16, 0xC3);
  memcpy(win.data(), win_io, isize);
  std::vector<uint32_t> ends(128);
  int rc = 0;
  emu_run_group<64>([&](EmuGroup<64> &g) {
    const int r = cm_bgzf_resolve(g, win.data(), tok, n_tok, isize, ends.data());
    if (g.t == 0) rc = r;
  }, reverse != 0);
  memcpy(win_io, win.data(), isize);
  return rc;
}
