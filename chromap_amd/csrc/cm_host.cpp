// cm_host.cpp -- host-side pieces either side of the device path: the index-file and FASTA
// loaders (byte-exact readers of the reference's formats) and the post-processing that
// defines the final BED bytes (sort, PCR-duplicate removal, MAPQ filter, Tn5 shift, text).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/chromap_amd.h"

extern "C" void cmgpu_default_params(cmgpu_params *p) {  // mapping_parameters.h:19-61
  memset(p, 0, sizeof(*p));
  p->error_threshold = 8;
  p->min_num_seeds = 2;
  p->max_seed_frequency0 = 500;
  p->max_seed_frequency1 = 1000;
  p->max_insert_size = 1000;
  p->min_read_length = 30;
  p->max_num_best_mappings = 1;
  p->drop_repetitive_reads = 500000;
  p->mapq_threshold = 30;
  p->read_batch_size = 500000;
  p->taskloop_grain_size = 5000;
  p->bc_error_threshold = 1;
  p->bc_probability_threshold = 0.9;
}

extern "C" int cmgpu_apply_preset(cmgpu_params *p, const char *preset) {  // chromap_driver.cc:247-275
  if (!strcmp(preset, "atac")) {
    p->max_insert_size = 2000;
    p->trim_adapters = 1;
    p->remove_pcr_duplicates = 1;
    p->tn5_shift = 1;
    p->low_memory_mode = 1;
  } else if (!strcmp(preset, "chip")) {
    p->max_insert_size = 2000;
    p->remove_pcr_duplicates = 1;
    p->low_memory_mode = 1;
  } else if (!strcmp(preset, "hic")) {
    p->error_threshold = 4;
    p->mapq_threshold = 1;
    p->split_alignment = 1;
    p->low_memory_mode = 1;
  } else {
    return CMGPU_EINVAL;
  }
  return CMGPU_OK;
}

// Index::Load (index.cc:132-169) + kh_load (khash.h:358-373): int k; int w; u32 n_keys;
// {u32 n_buckets,size,n_occupied,upper_bound; u32 flags[max(1,nb/16)]; u64 keys[nb]; u64 vals[nb]};
// u32 n_occ; u64 occ[n_occ].  Little-endian, no magic.
extern "C" int cmgpu_load_index_file(const char *path, cmgpu_index_view *out) {
  memset(out, 0, sizeof(*out));
  FILE *f = fopen(path, "rb");
  if (!f) return CMGPU_EIO;
  int32_t k = 0, w = 0;
  uint32_t n_keys = 0, hdr[4] = {0, 0, 0, 0};
  bool ok = fread(&k, 4, 1, f) == 1 && fread(&w, 4, 1, f) == 1 && fread(&n_keys, 4, 1, f) == 1 && fread(hdr, 4, 4, f) == 4;
  if (!ok) { fclose(f); return CMGPU_EIO; }
  const uint32_t nb = hdr[0];
  if (nb == 0 || (nb & (nb - 1))) { fclose(f); return CMGPU_EIO; }
  const size_t fw = nb < 16 ? 1 : nb >> 4;
  uint32_t *flags = (uint32_t *)malloc(fw * 4);
  uint64_t *keys = (uint64_t *)malloc((size_t)nb * 8);
  uint64_t *vals = (uint64_t *)malloc((size_t)nb * 8);
  uint64_t *occ = nullptr;
  uint32_t n_occ = 0;
  ok = flags && keys && vals && fread(flags, 4, fw, f) == fw && fread(keys, 8, nb, f) == nb && fread(vals, 8, nb, f) == nb &&
       fread(&n_occ, 4, 1, f) == 1;
  if (ok && n_occ) {
    occ = (uint64_t *)malloc((size_t)n_occ * 8);
    ok = occ && fread(occ, 8, n_occ, f) == n_occ;
  }
  fclose(f);
  if (!ok) { free(flags); free(keys); free(vals); free(occ); return CMGPU_EIO; }
  out->kmer_size = k;
  out->window_size = w;
  out->n_buckets = nb;
  out->flags = flags;
  out->keys = keys;
  out->vals = vals;
  out->n_occurrences = n_occ;
  out->occurrences = occ;
  return CMGPU_OK;
}

extern "C" void cmgpu_free_host_index(cmgpu_index_view *v) {
  free((void *)v->flags); free((void *)v->keys); free((void *)v->vals); free((void *)v->occurrences);
  memset(v, 0, sizeof(*v));
}

// SequenceBatch::LoadAllSequences (sequence_batch.cc:84-120) over kseq (kseq.h): name up to
// the first whitespace, sequence lines concatenated, empty records skipped.  Plain text.
extern "C" int cmgpu_load_reference_fasta(const char *path, cmgpu_ref_view *out) {
  memset(out, 0, sizeof(*out));
  FILE *f = fopen(path, "rb");
  if (!f) return CMGPU_EIO;
  std::vector<std::string> names, seqs;
  std::string name, seq;
  bool have = false, in_qual = false;
  size_t qual_len = 0;
  char *line = nullptr;
  size_t cap = 0;
  ssize_t ll;
  auto flush = [&]() {
    if (have && !seq.empty()) { names.push_back(name); seqs.push_back(seq); }
  };
  while ((ll = getline(&line, &cap, f)) >= 0) {
    while (ll > 0 && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
    if (in_qual) { qual_len += (size_t)ll; if (qual_len >= seq.size()) in_qual = false; continue; }
    if (line[0] == '>' || line[0] == '@') {
      flush();
      size_t e = 1;
      while (line[e] && line[e] != ' ' && line[e] != '\t' && line[e] != '\v' && line[e] != '\f') ++e;
      name.assign(line + 1, e - 1);
      seq.clear();
      have = true;
    } else if (line[0] == '+' && have) {
      in_qual = !seq.empty();
      qual_len = 0;
    } else if (have) {
      for (ssize_t i = 0; i < ll; ++i) if (line[i] > 32 && line[i] < 127) seq.push_back(line[i]);
    }
  }
  flush();
  free(line);
  fclose(f);
  const uint32_t n = (uint32_t)names.size();
  char **nm = (char **)calloc(n ? n : 1, sizeof(char *));
  char **sq = (char **)calloc(n ? n : 1, sizeof(char *));
  uint32_t *ln = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
  for (uint32_t i = 0; i < n; ++i) {
    nm[i] = strdup(names[i].c_str());
    sq[i] = (char *)malloc(seqs[i].size() + 1);
    memcpy(sq[i], seqs[i].data(), seqs[i].size());
    sq[i][seqs[i].size()] = 0;
    ln[i] = (uint32_t)seqs[i].size();
  }
  out->n_sequences = n;
  out->names = nm;
  out->sequences = sq;
  out->lengths = ln;
  return CMGPU_OK;
}

extern "C" void cmgpu_free_host_ref(cmgpu_ref_view *v) {
  for (uint32_t i = 0; i < v->n_sequences; ++i) { free((void *)v->names[i]); free((void *)v->sequences[i]); }
  free((void *)v->names); free((void *)v->sequences); free((void *)v->lengths);
  memset(v, 0, sizeof(*v));
}

// ---------------------------------------------------------------------------------------
// BED output for paired-end bulk data
//   sort:   per rid, PairedEndMappingWithoutBarcode::operator< (bed_mapping.h:208-215);
//           the low-memory path's temp-file merge yields the same global order
//           (mapping_processor.h:117-159, mapping_writer.h:166-376)
//   dedup:  runs of operator== (same start and length, bed_mapping.h:216-219) collapse to
//           the FIRST record with the maximal MAPQ, num_dups = min(255, run)
//           (mapping_writer.h:247-289)
//   filter: mapq >= mapq_threshold; then Tn5 shift (bed_mapping.h:224-229); line format
//           mapping_writer.cc:72-83
// ---------------------------------------------------------------------------------------
static inline bool rec_less(const cmgpu_record &a, const cmgpu_record &b) {
  return std::tie(a.rid, a.fragment_start, a.fragment_length, a.mapq, a.direction, a.is_unique, a.read_id,
                  a.positive_alignment_length, a.negative_alignment_length) <
         std::tie(b.rid, b.fragment_start, b.fragment_length, b.mapq, b.direction, b.is_unique, b.read_id,
                  b.positive_alignment_length, b.negative_alignment_length);
}

static inline void put_u32(std::string &s, uint32_t v) {
  char buf[12];
  int n = 0;
  do { buf[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) s.push_back(buf[--n]);
}

extern "C" int64_t cmgpu_write_bed_pe(const char *const *names, uint32_t n_sequences, const cmgpu_params *p,
                                      cmgpu_record *rec, uint64_t n, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return CMGPU_EIO;
  // in-memory flavour (chromap.h:1322-1355): Tn5 shift first, sort, RemovePCRDuplicate keeps the
  // LAST record of a run (mapping_processor.h:160-202); low-memory flavour: see above
  const bool inmem = !p->low_memory_mode;
  const bool shift_late = p->tn5_shift && !inmem;
  if (inmem && p->tn5_shift)
    for (uint64_t i = 0; i < n; ++i) {
      rec[i].fragment_start += 4; rec[i].positive_alignment_length -= 4; rec[i].fragment_length -= 9; rec[i].negative_alignment_length -= 5;
    }
  std::sort(rec, rec + n, rec_less);
  std::string buf;
  buf.reserve(1 << 20);
  int64_t lines = 0;
  auto emit = [&](cmgpu_record r, uint32_t dups) {
    if (r.rid >= n_sequences) return;
    r.num_dups = (uint8_t)(dups > 255 ? 255 : dups);
    if (shift_late) {
      r.fragment_start += 4;
      r.positive_alignment_length -= 4;
      r.fragment_length -= 9;
      r.negative_alignment_length -= 5;
    }
    buf.append(names[r.rid]);
    buf.push_back('\t');
    put_u32(buf, r.fragment_start);
    buf.push_back('\t');
    put_u32(buf, r.fragment_start + r.fragment_length);
    buf.append("\tN\t");
    put_u32(buf, r.mapq);
    buf.append(r.direction ? "\t+\t" : "\t-\t");
    put_u32(buf, r.num_dups);
    buf.push_back('\n');
    ++lines;
    if (buf.size() > (1 << 20) - 256) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
  };
  uint64_t i = 0;
  while (i < n) {
    cmgpu_record last = rec[i];
    uint32_t dups = 1;
    uint64_t j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && rec[j].rid == rec[i].rid && rec[j].fragment_start == rec[i].fragment_start &&
             rec[j].fragment_length == rec[i].fragment_length) {
        ++dups;
        if (inmem || rec[j].mapq > last.mapq) last = rec[j];
        ++j;
      }
    }
    if (last.mapq >= p->mapq_threshold) emit(last, dups);
    i = j;
  }
  if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  return lines;
}

// pairs output (mapping_writer.cc:381-420).  No duplicate removal unless requested (the hic
// preset does not set it): the low-memory merge emits every record whose MAPQ passes.
// MappingWriter<PairsMapping>::OutputHeader (mapping_writer.cc:385-399)
static void pairs_header(std::string &buf, const char *const *names, const uint32_t *lengths, uint32_t n_sequences, const uint32_t *pairs_rank) {
  buf.append("## pairs format v1.0.0\n#shape: upper triangle\n");
  for (uint32_t i = 0; i < n_sequences; ++i) {
    uint32_t rid = i;
    if (pairs_rank) for (uint32_t j = 0; j < n_sequences; ++j) if (pairs_rank[j] == i) rid = j;
    buf.append("#chromsize: ");
    buf.append(names[rid]);
    buf.push_back(' ');
    put_u32(buf, lengths[rid]);
    buf.push_back('\n');
  }
  buf.append("#columns: readID chrom1 pos1 chrom2 pos2 strand1 strand2 pair_type mapq1 mapq2\n");
}
// the header alone, to a new file: the lines of cmgpu_store_format_pairs follow (cmgpu_store_write_text with append = 1)
extern "C" int cmgpu_write_pairs_header(const char *const *names, const uint32_t *lengths, uint32_t n_sequences, const uint32_t *pairs_rank,
                                        const char *out_path) {
  if (!names || !lengths || !out_path) return CMGPU_EINVAL;
  FILE *f = fopen(out_path, "wb");
  if (!f) return CMGPU_EIO;
  std::string buf;
  pairs_header(buf, names, lengths, n_sequences, pairs_rank);
  const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
  return fclose(f) == 0 && ok ? CMGPU_OK : CMGPU_EIO;
}

extern "C" int64_t cmgpu_write_pairs(const char *const *names, const uint32_t *lengths, uint32_t n_sequences,
                                     const cmgpu_params *p, cmgpu_pairs_record *rec, uint64_t n,
                                     const char *const *read_names, uint32_t read_id_base, const char *out_path) {
  return cmgpu_write_pairs_ranked(names, lengths, n_sequences, p, rec, n, read_names, read_id_base, nullptr, out_path);
}

// pairs_rank (--pairs-natural-chr-order) only orders the #chromsize header lines here (mapping_writer.cc:385-399);
// the flip of the two ends was done on the device with the same table (cmgpu_set_pairs_chr_order)
extern "C" int64_t cmgpu_write_pairs_ranked(const char *const *names, const uint32_t *lengths, uint32_t n_sequences,
                                            const cmgpu_params *p, cmgpu_pairs_record *rec, uint64_t n,
                                            const char *const *read_names, uint32_t read_id_base, const uint32_t *pairs_rank,
                                            const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return CMGPU_EIO;
  std::sort(rec, rec + n, [](const cmgpu_pairs_record &a, const cmgpu_pairs_record &b) {
    return std::tie(a.rid1, a.rid2, a.pos1, a.pos2, a.mapq, a.read_id) < std::tie(b.rid1, b.rid2, b.pos1, b.pos2, b.mapq, b.read_id);
  });
  std::string buf;
  buf.reserve(1 << 20);
  pairs_header(buf, names, lengths, n_sequences, pairs_rank);
  int64_t lines = 0;
  auto same = [](const cmgpu_pairs_record &a, const cmgpu_pairs_record &b) {  // PairsMapping::operator== (pairs_mapping.h:45-50)
    return a.rid1 == b.rid1 && a.pos1 == b.pos1 && a.rid2 == b.rid2 && a.pos2 == b.pos2;
  };
  for (uint64_t i = 0; i < n;) {
    // --remove-pcr-duplicates: one record per run of equal positions -- the low-memory merge keeps the FIRST record with the run's largest
    // MAPQ (mapping_writer.h:244-270), the in-memory RemovePCRDuplicate the LAST of the run (mapping_processor.h:178-195)
    uint64_t pick = i, j = i + 1;
    if (p->remove_pcr_duplicates)
      for (; j < n && same(rec[j], rec[i]); ++j)
        if (!p->low_memory_mode || rec[j].mapq > rec[pick].mapq) pick = j;
    const cmgpu_pairs_record &r = rec[pick];
    i = j;
    if (r.mapq < p->mapq_threshold || r.rid1 >= n_sequences || r.rid2 >= n_sequences) continue;
    buf.append(read_names[r.read_id - read_id_base]);
    buf.push_back('\t');
    buf.append(names[r.rid1]);
    buf.push_back('\t');
    put_u32(buf, r.pos1 + 1);
    buf.push_back('\t');
    buf.append(names[r.rid2]);
    buf.push_back('\t');
    put_u32(buf, r.pos2 + 1);
    buf.append(r.strand1 ? "\t+" : "\t-");
    buf.append(r.strand2 ? "\t+" : "\t-");
    buf.append("\tUU\t");
    put_u32(buf, r.mapq);
    buf.push_back('\t');
    put_u32(buf, r.mapq);
    buf.push_back('\n');
    ++lines;
    if (buf.size() > (1 << 20) - 512) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
  }
  if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  return lines;
}

// Chromap::LoadBarcodeWhitelist (chromap.cc:388-490): one barcode per line, key =
// GenerateSeedFromSequence (utils.h:111-129).  Plain text.
extern "C" int cmgpu_load_whitelist_file(const char *path, uint32_t barcode_length, uint64_t **keys_out, uint32_t *n_out) {
  FILE *f = fopen(path, "rb");
  if (!f) return CMGPU_EIO;
  std::vector<uint64_t> keys;
  char line[300];
  while (fgets(line, sizeof(line), f)) {
    size_t l = strlen(line);
    while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
    if (l > 32 || l != barcode_length) { fclose(f); return CMGPU_EINVAL; }  // chromap.cc:403-415
    uint64_t seed = 0;
    for (size_t i = 0; i < l; ++i) {
      const char ch = line[i] & 0xDF;
      const uint64_t b = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
      seed = b < 4 ? (seed << 2) | b : seed << 2;
    }
    keys.push_back(seed);
  }
  fclose(f);
  uint64_t *k = (uint64_t *)malloc((keys.size() ? keys.size() : 1) * 8);
  memcpy(k, keys.data(), keys.size() * 8);
  *keys_out = k;
  *n_out = (uint32_t)keys.size();
  return CMGPU_OK;
}

// BED for PairedEndMappingWithBarcode: sort by operator< (bed_mapping.h:145-153) within rid,
// cell-level duplicate removal on (barcode, start, length) (:154-159) keeping the first record
// with the maximal MAPQ (mapping_writer.h:247-289), MAPQ filter, Tn5 shift, line
// `chr start end barcode num_dups` with Seed2Sequence (mapping_writer.cc:119-131,
// barcode_translator.h:107-116).
extern "C" int64_t cmgpu_write_bed_pe_bc(const char *const *names, uint32_t n_sequences, const cmgpu_params *p,
                                         cmgpu_record_bc *rec, uint64_t n, uint32_t barcode_length, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return CMGPU_EIO;
  const bool inmem = !p->low_memory_mode;
  // bulk-level duplicate removal needs the whitelist abundances, which live on the device: cmgpu_store_format does it
  if (p->dedup_at_bulk_level && !inmem && p->remove_pcr_duplicates) { fclose(f); return CMGPU_EINVAL; }
  if (inmem && p->tn5_shift)
    for (uint64_t t = 0; t < n; ++t) { rec[t].r.fragment_start += 4; rec[t].r.fragment_length -= 9; }
  std::sort(rec, rec + n, [](const cmgpu_record_bc &a, const cmgpu_record_bc &b) {
    return std::tie(a.r.rid, a.r.fragment_start, a.r.fragment_length, a.barcode, a.r.mapq, a.r.direction, a.r.is_unique, a.r.read_id,
                    a.r.positive_alignment_length, a.r.negative_alignment_length) <
           std::tie(b.r.rid, b.r.fragment_start, b.r.fragment_length, b.barcode, b.r.mapq, b.r.direction, b.r.is_unique, b.r.read_id,
                    b.r.positive_alignment_length, b.r.negative_alignment_length);
  });
  std::string buf;
  buf.reserve(1 << 20);
  int64_t lines = 0;
  uint64_t i = 0;
  while (i < n) {
    cmgpu_record_bc last = rec[i];
    uint32_t dups = 1;
    uint64_t j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && rec[j].r.rid == last.r.rid && rec[j].barcode == last.barcode &&
             rec[j].r.fragment_start == last.r.fragment_start && rec[j].r.fragment_length == last.r.fragment_length) {
        ++dups;
        if (inmem || rec[j].r.mapq > last.r.mapq) last = rec[j];
        ++j;
      }
    }
    if (last.r.mapq >= p->mapq_threshold && last.r.rid < n_sequences) {
      cmgpu_record r = last.r;
      if (p->tn5_shift && !inmem) { r.fragment_start += 4; r.fragment_length -= 9; }
      buf.append(names[r.rid]);
      buf.push_back('\t');
      put_u32(buf, r.fragment_start);
      buf.push_back('\t');
      put_u32(buf, r.fragment_start + r.fragment_length);
      buf.push_back('\t');
      for (uint32_t b = 0; b < barcode_length; ++b) buf.push_back("ACGT"[(last.barcode >> ((barcode_length - 1 - b) * 2)) & 3]);
      buf.push_back('\t');
      put_u32(buf, dups > 255 ? 255 : dups);
      buf.push_back('\n');
      ++lines;
      if (buf.size() > (1 << 20) - 256) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    }
    i = j;
  }
  if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  return lines;
}

// single-end bulk BED (MappingWithoutBarcode): see include/chromap_amd.h
extern "C" int64_t cmgpu_write_bed_se(const char *const *names, uint32_t n_sequences, const cmgpu_params *p, cmgpu_record *rec,
                                      uint64_t n, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return CMGPU_EIO;
  const bool inmem = !p->low_memory_mode;
  if (inmem && p->tn5_shift)
    for (uint64_t t = 0; t < n; ++t) { if (rec[t].direction == 1) rec[t].fragment_start += 4; else rec[t].fragment_length -= 5; }
  std::sort(rec, rec + n, [](const cmgpu_record &a, const cmgpu_record &b) {
    return std::tie(a.rid, a.fragment_start, a.fragment_length, a.mapq, a.direction, a.is_unique, a.read_id) <
           std::tie(b.rid, b.fragment_start, b.fragment_length, b.mapq, b.direction, b.is_unique, b.read_id);
  });
  std::string buf;
  buf.reserve(1 << 20);
  int64_t lines = 0;
  uint64_t i = 0;
  while (i < n) {
    cmgpu_record last = rec[i];
    uint32_t dups = 1;
    uint64_t j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && rec[j].rid == rec[i].rid && rec[j].fragment_start == rec[i].fragment_start) {
        ++dups;
        if (inmem || rec[j].mapq > last.mapq) last = rec[j];
        ++j;
      }
    }
    if (last.mapq >= p->mapq_threshold && last.rid < n_sequences) {
      if (p->tn5_shift && !inmem) { if (last.direction == 1) last.fragment_start += 4; else last.fragment_length -= 5; }
      buf.append(names[last.rid]);
      buf.push_back('\t');
      put_u32(buf, last.fragment_start);
      buf.push_back('\t');
      put_u32(buf, last.fragment_start + last.fragment_length);
      buf.append("\tN\t");
      put_u32(buf, last.mapq);
      buf.append(last.direction ? "\t+\t" : "\t-\t");
      put_u32(buf, dups > 255 ? 255 : dups);
      buf.push_back('\n');
      ++lines;
      if (buf.size() > (1 << 20) - 256) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    }
    i = j;
  }
  if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  return lines;
}

// ---------------------------------------------------------------------------------------
// SAM text (mapping_writer.cc:312-356).  Sort: SAMMapping::operator< under the per-chromosome
// vectors (sam_mapping.h:193-199; barcode 0); duplicate runs: operator== (:200-205), survivor as
// in the BED writers (low-memory merge: first maximal MAPQ; in-memory: last); MAPQ filter.
// The sequence is printed as mapped (reverse complement for the - strand, PrepareNegativeSequenceAt),
// the quality reversed with it (sam_mapping.h:172-179), both cut to the trimmed length.
// ---------------------------------------------------------------------------------------
// --barcode-translate (BarcodeTranslator, barcode_translator.h:43-101): table lines are "to<TAB or ,>from"; the key is the 2-bit
// packed `from` (GenerateSeedFromSequence, utils.h:111-129: other letters count as A); the LAST line's `from` length is the
// segment length.  A barcode of several segments is translated segment by segment and joined with '-' -- the segments are cut
// with the reference's own shifts (:80-82), which only isolate a segment when there is one.
struct CmBarcodeTable {
  std::unordered_map<uint64_t, std::string> to;
  uint32_t from_len = 0;
  uint64_t mask = 0;
};
static int parse_barcode_table(const char *text, uint64_t bytes, CmBarcodeTable *t) {
  const char *p = text, *end = text + bytes;
  while (p < end) {
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    if (!nl) nl = end;
    const size_t l = (size_t)(nl - p);
    size_t i = 0;
    while (i < l && p[i] != ',' && p[i] != '\t') ++i;
    if (i < l) {  // (a line without a separator gives the reference a `from` length of -1: such a table is not usable there either)
      t->from_len = (uint32_t)(l - i - 1);
      uint64_t seed = 0;
      for (uint32_t k = 0; k < t->from_len; ++k) {
        const char c = p[i + 1 + k];
        const uint64_t b = (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 0;
        seed = (seed << 2) | b;
      }
      t->to[seed] = std::string(p, i);
    }
    p = nl + 1;
  }
  t->mask = t->from_len >= 32 ? ~0ull : (1ull << (2 * t->from_len)) - 1;
  return t->from_len ? CMGPU_OK : CMGPU_EINVAL;
}

static int64_t write_sam_impl(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *p,
                              const cmgpu_sam_record *rec, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                              const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                              const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                              const char *quals2, const uint32_t *offsets2, const uint64_t *bck, uint32_t bc_len, const char *out_path,
                              const CmBarcodeTable *tr = nullptr) {
  auto bc_of = [&](uint64_t slot) -> uint64_t { return bck ? bck[paired ? slot / 2 : slot] : 0; };
  FILE *f = fopen(out_path, "wb");
  if (!f) return CMGPU_EIO;
  std::string buf;
  buf.reserve(1 << 20);
  for (uint32_t i = 0; i < n_sequences; ++i) {
    buf.append("@SQ\tSN:");
    buf.append(ref_names[i]);
    buf.append("\tLN:");
    put_u32(buf, ref_lengths[i]);
    buf.push_back('\n');
  }
  std::vector<uint64_t> v;
  v.reserve(n_slots);
  for (uint64_t i = 0; i < n_slots; ++i) if (rec[i].valid) v.push_back(i);
  std::sort(v.begin(), v.end(), [&](uint64_t a, uint64_t b) {
    const cmgpu_sam_record &x = rec[a], &y = rec[b];
    const int xf = x.flag & 64, yf = y.flag & 64;
    const uint64_t xb = bc_of(a), yb = bc_of(b);
    return std::tie(x.rid, x.pos, xb, x.mrid, x.mpos, xf, x.mapq, x.read_id) < std::tie(y.rid, y.pos, yb, y.mrid, y.mpos, yf, y.mapq, y.read_id);
  });
  auto same = [&](uint64_t a, uint64_t b) {
    const cmgpu_sam_record &x = rec[a], &y = rec[b];
    return x.pos == y.pos && x.rid == y.rid && bc_of(a) == bc_of(b) && (x.flag & 64) == (y.flag & 64) && x.mrid == y.mrid && x.mpos == y.mpos;
  };
  const bool inmem = !p->low_memory_mode;
  int64_t lines = 0;
  size_t i = 0;
  std::string seq, qual;
  while (i < v.size()) {
    uint64_t last = v[i];
    size_t j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < v.size() && same(v[j], v[i])) {
        if (inmem || rec[v[j]].mapq > rec[last].mapq) last = v[j];
        ++j;
      }
    }
    const cmgpu_sam_record &r = rec[last];
    if ((int)r.mapq >= p->mapq_threshold && r.rid < n_sequences) {
      const uint64_t item = paired ? last / 2 : last;
      const bool mate2 = paired && (last & 1);
      const char *bs = (mate2 ? bases2 : bases1) + (mate2 ? offsets2 : offsets1)[item];
      const char *qs = (mate2 ? quals2 : quals1) + (mate2 ? offsets2 : offsets1)[item];
      uint32_t L = r.length_after_trim;
      const uint32_t *cg = cigar_pool + last * CMGPU_SAM_CIGAR_CAP;
      uint32_t ql = 0;  // SAMMapping::GetSequenceLength (sam_mapping.h:246-256)
      for (uint32_t ci = 0; ci < r.n_cigar; ++ci) { const uint32_t op = cg[ci] & 0xf; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += cg[ci] >> 4; }
      seq.assign(L, 'N');
      qual.assign(L, '!');
      if (r.strand) { seq.assign(bs, L); qual.assign(qs, L); }
      else
        for (uint32_t t = 0; t < L; ++t) {
          const char c = bs[L - 1 - t] & 0xDF;
          seq[t] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
          qual[t] = qs[L - 1 - t];
        }
      if (ql < L) { seq.resize(ql); qual.resize(ql); }
      buf.append(mate2 ? names2[item] : names1[item]);
      buf.push_back('\t');
      put_u32(buf, r.flag);
      buf.push_back('\t');
      buf.append(ref_names[r.rid]);
      buf.push_back('\t');
      put_u32(buf, r.pos + 1);
      buf.push_back('\t');
      put_u32(buf, r.mapq);
      buf.push_back('\t');
      if (r.n_cigar == 0) buf.push_back('*');
      for (uint32_t ci = 0; ci < r.n_cigar; ++ci) { put_u32(buf, cg[ci] >> 4); buf.push_back("MIDNSHP=XB??????"[cg[ci] & 0xf]); }
      buf.push_back('\t');
      if (r.mrid < 0) buf.push_back('*'); else if ((uint32_t)r.mrid == r.rid) buf.push_back('='); else buf.append(ref_names[r.mrid]);
      buf.push_back('\t');
      put_u32(buf, r.mrid < 0 ? 0u : r.mpos + 1);
      buf.push_back('\t');
      if (r.tlen < 0) { buf.push_back('-'); put_u32(buf, (uint32_t)(-(int64_t)r.tlen)); } else put_u32(buf, (uint32_t)r.tlen);
      buf.push_back('\t');
      buf.append(seq);
      buf.push_back('\t');
      buf.append(qual);
      buf.append("\tNM:i:");
      put_u32(buf, r.nm);
      buf.append("\tMD:Z:");
      buf.append(md_pool + last * (uint64_t)md_cap, r.md_len);
      if (bck) {
        buf.append("\tCB:Z:");
        const uint64_t key = bc_of(last);
        if (!tr) {
          for (uint32_t b = 0; b < bc_len; ++b) buf.push_back("ACGT"[(key >> ((bc_len - 1 - b) * 2)) & 3]);  // Seed2Sequence
        } else {
          const uint64_t nseg = bc_len / tr->from_len;
          for (uint64_t sg = 0; sg < nseg; ++sg) {
            const uint64_t sh1 = 2 * sg * tr->from_len, sh2 = 2 * (nseg - 1) * tr->from_len;
            const uint64_t seed = ((sh1 < 64 ? key << sh1 : 0) >> (sh2 < 64 ? sh2 : 63)) & tr->mask;
            const auto it = tr->to.find(seed);
            if (it == tr->to.end()) { fclose(f); return CMGPU_EFORMAT; }  // "Barcode does not exist in the translation table." (exit(-1) there)
            if (sg) buf.push_back('-');
            buf.append(it->second);
          }
        }
      }
      buf.push_back('\n');
      ++lines;
      if (buf.size() > (1 << 20) - 4096) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); }
    }
    i = j;
  }
  if (!buf.empty()) fwrite(buf.data(), 1, buf.size(), f);
  fclose(f);
  return lines;
}

extern "C" int64_t cmgpu_write_sam(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *p,
                                   const cmgpu_sam_record *rec, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                                   const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                                   const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                                   const char *quals2, const uint32_t *offsets2, const char *out_path) {
  return write_sam_impl(ref_names, ref_lengths, n_sequences, p, rec, n_slots, paired, cigar_pool, md_pool, md_cap, names1, names2, bases1, quals1,
                        offsets1, bases2, quals2, offsets2, nullptr, 0, out_path);
}

extern "C" int64_t cmgpu_write_sam_barcoded(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *p,
                                            const cmgpu_sam_record *rec, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                                            const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                                            const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                                            const char *quals2, const uint32_t *offsets2, const uint64_t *barcode_keys, uint32_t barcode_length,
                                            const char *out_path) {
  if (!barcode_keys || barcode_length == 0 || barcode_length > 32) return CMGPU_EINVAL;
  return write_sam_impl(ref_names, ref_lengths, n_sequences, p, rec, n_slots, paired, cigar_pool, md_pool, md_cap, names1, names2, bases1, quals1,
                        offsets1, bases2, quals2, offsets2, barcode_keys, barcode_length, out_path);
}

extern "C" int64_t cmgpu_write_sam_barcoded_translated(const char *const *ref_names, const uint32_t *ref_lengths, uint32_t n_sequences, const cmgpu_params *p,
                                                       const cmgpu_sam_record *rec, uint64_t n_slots, int paired, const uint32_t *cigar_pool,
                                                       const char *md_pool, uint32_t md_cap, const char *const *names1, const char *const *names2,
                                                       const char *bases1, const char *quals1, const uint32_t *offsets1, const char *bases2,
                                                       const char *quals2, const uint32_t *offsets2, const uint64_t *barcode_keys, uint32_t barcode_length,
                                                       const char *translate_table, uint64_t translate_table_bytes, const char *out_path) {
  if (!barcode_keys || barcode_length == 0 || barcode_length > 32 || !translate_table) return CMGPU_EINVAL;
  CmBarcodeTable tr;
  const int rc = parse_barcode_table(translate_table, translate_table_bytes, &tr);
  if (rc) return rc;
  return write_sam_impl(ref_names, ref_lengths, n_sequences, p, rec, n_slots, paired, cigar_pool, md_pool, md_cap, names1, names2, bases1, quals1,
                        offsets1, bases2, quals2, offsets2, barcode_keys, barcode_length, out_path, &tr);
}
