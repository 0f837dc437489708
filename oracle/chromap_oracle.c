/*
 * chromap_oracle.c -- TEST INFRASTRUCTURE ONLY (see chromap_oracle.h).
 *
 * Sequential CPU restatement of the reference's hot path.  Citations are into
 * /root/reference/src.  Written from scratch in C (flat arrays, no STL); the
 * semantics -- including integer wrap-arounds and truncations -- follow the
 * reference so that results are bit-identical.
 */
#define _GNU_SOURCE
#include "chromap_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* small growable arrays                                                      */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t *a; size_t n, cap; } vec64;
static void v64_push(vec64 *v, uint64_t x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 64;
    v->a = (uint64_t *)realloc(v->a, v->cap * sizeof(uint64_t));
  }
  v->a[v->n++] = x;
}
static int cmp_u64(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

/* candidate.h:8-34 */
typedef struct { uint64_t position; uint8_t count; } cand_t;
typedef struct { cand_t *a; size_t n, cap; } vcand;
static void vc_push(vcand *v, cand_t x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 32;
    v->a = (cand_t *)realloc(v->a, v->cap * sizeof(cand_t));
  }
  v->a[v->n++] = x;
}
/* Candidate::operator< : count desc, position asc (candidate.h:22-33) */
static int cmp_cand(const void *a, const void *b) {
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  if (x->count != y->count) return x->count > y->count ? -1 : 1;
  return x->position < y->position ? -1 : x->position > y->position;
}

/* draft_mapping.h:8-24 */
typedef struct { int num_errors; uint64_t position; } draft_t;
typedef struct { draft_t *a; size_t n, cap; } vdraft;
static void vd_push(vdraft *v, int e, uint64_t p) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 32;
    v->a = (draft_t *)realloc(v->a, v->cap * sizeof(draft_t));
  }
  v->a[v->n].num_errors = e;
  v->a[v->n++].position = p;
}
/* stable merge sort by position: std::sort is unstable, but ties between equal positions
 * only reorder mappings with the same end position; see note at sort_drafts(). */
static int cmp_draft_pos(const void *a, const void *b) {
  const draft_t *x = (const draft_t *)a, *y = (const draft_t *)b;
  return x->position < y->position ? -1 : x->position > y->position;
}

typedef struct { uint32_t a, b; } pair32;
typedef struct { pair32 *a; size_t n, cap; } vpair;
static void vp_push(vpair *v, uint32_t x, uint32_t y) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 16;
    v->a = (pair32 *)realloc(v->a, v->cap * sizeof(pair32));
  }
  v->a[v->n].a = x;
  v->a[v->n++].b = y;
}

/* ------------------------------------------------------------------------- */
/* utils.h:76-108                                                             */
/* ------------------------------------------------------------------------- */
uint64_t ora_hash64(uint64_t key, uint64_t mask) {
  key = (~key + (key << 21)) & mask;
  key = key ^ key >> 24;
  key = ((key + (key << 3)) + (key << 8)) & mask;
  key = key ^ key >> 14;
  key = ((key + (key << 2)) + (key << 4)) & mask;
  key = key ^ key >> 28;
  key = (key + (key << 31)) & mask;
  return key;
}

/* CharToUint8 (utils.h:87-104): A/a 0, C/c 1, G/g 2, T/t 3, everything else 4 */
static inline uint8_t c2u(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}
/* Uint8ToChar (utils.h:100-108) */
static inline char u2c(uint8_t i) { return "ACGTNNNN"[i & 7]; }

/* ------------------------------------------------------------------------- */
/* minimizer_generator.cc:7-139                                               */
/* ------------------------------------------------------------------------- */
int ora_minimizers(const char *seq, uint32_t len, uint32_t seq_index, int k, int w,
                   uint64_t *out_hash, uint64_t *out_hit) {
  const uint64_t shift = 2 * (uint64_t)(k - 1);
  const uint64_t mask = (((uint64_t)1) << (2 * k)) - 1;
  uint64_t fw = 0, rv = 0;
  uint64_t bh[256], bp[256]; /* ring buffer of (hash, hit) */
  uint64_t min_h = UINT64_MAX, min_p = UINT64_MAX;
  int n = 0;
  for (int i = 0; i < w; ++i) bh[i] = bp[i] = UINT64_MAX; /* memset 0xff, :21 */
  int unamb = 0, pib = 0, min_pos = 0;
#define EMIT(h, p) do { out_hash[n] = (h); out_hit[n] = (p); ++n; } while (0)
  for (uint32_t pos = 0; pos < len; ++pos) {
    const uint8_t c = c2u(seq[pos]);
    uint64_t cur_h = UINT64_MAX, cur_p = UINT64_MAX;
    if (c < 4) {
      fw = ((fw << 2) | c) & mask;                         /* :35-36 */
      rv = (rv >> 2) | (((uint64_t)(3 ^ c)) << shift);      /* :38-40 */
      if (fw == rv) continue;                              /* :42-45 palindrome: no buffer write */
      const uint64_t h0 = ora_hash64(fw, mask), h1 = ora_hash64(rv, mask);
      const uint64_t strand = h0 < h1 ? 0 : 1;             /* :51-52 */
      ++unamb;
      if (unamb >= k) {
        cur_h = ora_hash64(strand ? h1 : h0, mask);        /* :57 hashed twice */
        cur_p = ((((uint64_t)seq_index) << 32 | (uint32_t)pos) << 1) | strand;
      }
    } else {
      unamb = 0;
    }
    bh[pib] = cur_h;
    bp[pib] = cur_p;
    if (unamb == w + k - 1 && min_h != UINT64_MAX && min_h < cur_h) { /* :69-81 */
      for (int j = pib + 1; j < w; ++j)
        if (min_h == bh[j] && bp[j] != min_p) EMIT(bh[j], bp[j]);
      for (int j = 0; j < pib; ++j)
        if (min_h == bh[j] && bp[j] != min_p) EMIT(bh[j], bp[j]);
    }
    if (cur_h <= min_h) { /* :83-90 */
      if (unamb >= w + k && min_h != UINT64_MAX) EMIT(min_h, min_p);
      min_h = cur_h;
      min_p = cur_p;
      min_pos = pib;
    } else if (pib == min_pos) { /* :91-128 */
      if (unamb >= w + k - 1 && min_h != UINT64_MAX) EMIT(min_h, min_p);
      min_h = UINT64_MAX;
      for (int j = pib + 1; j < w; ++j)
        if (min_h >= bh[j]) { min_h = bh[j]; min_p = bp[j]; min_pos = j; }
      for (int j = 0; j <= pib; ++j)
        if (min_h >= bh[j]) { min_h = bh[j]; min_p = bp[j]; min_pos = j; }
      if (unamb >= w + k - 1 && min_h != UINT64_MAX) {
        for (int j = pib + 1; j < w; ++j)
          if (min_h == bh[j] && min_p != bp[j]) EMIT(bh[j], bp[j]);
        for (int j = 0; j <= pib; ++j)
          if (min_h == bh[j] && min_p != bp[j]) EMIT(bh[j], bp[j]);
      }
    }
    if (++pib == w) pib = 0;
  }
  if (min_h != UINT64_MAX) EMIT(min_h, min_p); /* :136-138 */
#undef EMIT
  return n;
}

/* ------------------------------------------------------------------------- */
/* khash (khash.h:165-350) with the index's hash/eq (index_utils.h:13-17)      */
/* ------------------------------------------------------------------------- */
#define FL_ISEMPTY(f, i) ((f[(i) >> 4] >> (((i) & 0xfU) << 1)) & 2)
#define FL_ISDEL(f, i) ((f[(i) >> 4] >> (((i) & 0xfU) << 1)) & 1)
#define FL_ISEITHER(f, i) ((f[(i) >> 4] >> (((i) & 0xfU) << 1)) & 3)
#define FL_SET_DEL_TRUE(f, i) (f[(i) >> 4] |= 1ul << (((i) & 0xfU) << 1))
#define FL_SET_EMPTY_FALSE(f, i) (f[(i) >> 4] &= ~(2ul << (((i) & 0xfU) << 1)))
#define FL_SET_BOTH_FALSE(f, i) (f[(i) >> 4] &= ~(3ul << (((i) & 0xfU) << 1)))
#define FL_SIZE(m) ((m) < 16 ? 1 : (m) >> 4)
static const double KH_UPPER = 0.77;

uint32_t ora_kh_get(const ora_index *h, uint64_t key, uint64_t *steps) {
  if (!h->n_buckets) return 0;
  uint32_t mask = h->n_buckets - 1, step = 0;
  uint32_t k = (uint32_t)(key >> 1), i = k & mask, last = i;
  uint64_t visited = 1;
  while (!FL_ISEMPTY(h->flags, i) &&
         (FL_ISDEL(h->flags, i) || !((h->keys[i] >> 1) == (key >> 1)))) {
    i = (i + (++step)) & mask;
    ++visited;
    if (i == last) { if (steps) *steps += visited; return h->n_buckets; }
  }
  if (steps) *steps += visited;
  return FL_ISEITHER(h->flags, i) ? h->n_buckets : i;
}

static int kh_resize(ora_index *h, uint32_t new_n) { /* khash.h:246-308 */
  uint32_t *new_flags = 0;
  uint32_t j = 1;
  {
    --new_n; new_n |= new_n >> 1; new_n |= new_n >> 2; new_n |= new_n >> 4;
    new_n |= new_n >> 8; new_n |= new_n >> 16; ++new_n;
    if (new_n < 4) new_n = 4;
    if (h->size >= (uint32_t)(new_n * KH_UPPER + 0.5)) j = 0;
    else {
      new_flags = (uint32_t *)malloc(FL_SIZE(new_n) * sizeof(uint32_t));
      memset(new_flags, 0xaa, FL_SIZE(new_n) * sizeof(uint32_t));
      if (h->n_buckets < new_n) {
        h->keys = (uint64_t *)realloc(h->keys, (size_t)new_n * sizeof(uint64_t));
        h->vals = (uint64_t *)realloc(h->vals, (size_t)new_n * sizeof(uint64_t));
      }
    }
  }
  if (j) {
    for (j = 0; j != h->n_buckets; ++j) {
      if (FL_ISEITHER(h->flags, j) == 0) {
        uint64_t key = h->keys[j], val = h->vals[j];
        uint32_t new_mask = new_n - 1;
        FL_SET_DEL_TRUE(h->flags, j);
        while (1) {
          uint32_t k = (uint32_t)(key >> 1), i = k & new_mask, step = 0;
          while (!FL_ISEMPTY(new_flags, i)) i = (i + (++step)) & new_mask;
          FL_SET_EMPTY_FALSE(new_flags, i);
          if (i < h->n_buckets && FL_ISEITHER(h->flags, i) == 0) {
            uint64_t t = h->keys[i]; h->keys[i] = key; key = t;
            t = h->vals[i]; h->vals[i] = val; val = t;
            FL_SET_DEL_TRUE(h->flags, i);
          } else {
            h->keys[i] = key;
            h->vals[i] = val;
            break;
          }
        }
      }
    }
    if (h->n_buckets > new_n) {
      h->keys = (uint64_t *)realloc(h->keys, (size_t)new_n * sizeof(uint64_t));
      h->vals = (uint64_t *)realloc(h->vals, (size_t)new_n * sizeof(uint64_t));
    }
    free(h->flags);
    h->flags = new_flags;
    h->n_buckets = new_n;
    h->n_occupied = h->size;
    h->upper_bound = (uint32_t)(h->n_buckets * KH_UPPER + 0.5);
  }
  return 0;
}

static uint32_t kh_put(ora_index *h, uint64_t key, int *ret) { /* khash.h:310-350 */
  uint32_t x;
  if (h->n_occupied >= h->upper_bound) {
    if (h->n_buckets > (h->size << 1)) kh_resize(h, h->n_buckets - 1);
    else kh_resize(h, h->n_buckets + 1);
  }
  {
    uint32_t k, i, site, last, mask = h->n_buckets - 1, step = 0;
    x = site = h->n_buckets;
    k = (uint32_t)(key >> 1);
    i = k & mask;
    if (FL_ISEMPTY(h->flags, i)) x = i;
    else {
      last = i;
      while (!FL_ISEMPTY(h->flags, i) &&
             (FL_ISDEL(h->flags, i) || !((h->keys[i] >> 1) == (key >> 1)))) {
        if (FL_ISDEL(h->flags, i)) site = i;
        i = (i + (++step)) & mask;
        if (i == last) { x = site; break; }
      }
      if (x == h->n_buckets) {
        if (FL_ISEMPTY(h->flags, i) && site != h->n_buckets) x = site;
        else x = i;
      }
    }
  }
  if (FL_ISEMPTY(h->flags, x)) {
    h->keys[x] = key;
    FL_SET_BOTH_FALSE(h->flags, x);
    ++h->size; ++h->n_occupied;
    *ret = 1;
  } else if (FL_ISDEL(h->flags, x)) {
    h->keys[x] = key;
    FL_SET_BOTH_FALSE(h->flags, x);
    ++h->size;
    *ret = 2;
  } else *ret = 0;
  return x;
}

typedef struct { uint64_t hash, hit; } mm_t;
static int cmp_mm(const void *a, const void *b) { /* minimizer.h:35-45 */
  const mm_t *x = (const mm_t *)a, *y = (const mm_t *)b;
  if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
  return x->hit < y->hit ? -1 : x->hit > y->hit;
}

int ora_index_build(const ora_ref *ref, int k, int w, ora_index *idx) { /* index.cc:12-89 */
  memset(idx, 0, sizeof(*idx));
  idx->k = k;
  idx->w = w;
  size_t cap = 1024, n = 0;
  mm_t *mm = (mm_t *)malloc(cap * sizeof(mm_t));
  for (uint32_t r = 0; r < ref->n_seq; ++r) {
    uint64_t *hh = (uint64_t *)malloc(((size_t)ref->len[r] + 1) * sizeof(uint64_t));
    uint64_t *pp = (uint64_t *)malloc(((size_t)ref->len[r] + 1) * sizeof(uint64_t));
    int c = ora_minimizers(ref->seq[r], ref->len[r], r, k, w, hh, pp);
    if (n + c > cap) { while (n + c > cap) cap *= 2; mm = (mm_t *)realloc(mm, cap * sizeof(mm_t)); }
    for (int i = 0; i < c; ++i) { mm[n].hash = hh[i]; mm[n].hit = pp[i]; ++n; }
    free(hh); free(pp);
  }
  if (n == 0) { free(mm); return -1; }
  qsort(mm, n, sizeof(mm_t), cmp_mm); /* (hash,hit) pairs are unique -> stable_sort irrelevant */
  vec64 occ = {0};
  uint64_t prev = mm[0].hash << 1, nonsingle = 0;
  uint32_t nprev = 0;
  for (size_t mi = 0; mi <= n; ++mi) {
    const int last = mi == n;
    const uint64_t cur = last ? prev + 1 : mm[mi].hash << 1;
    if (cur != prev) {
      int rc;
      uint32_t it = kh_put(idx, prev, &rc);
      if (nprev == 1) {
        idx->keys[it] |= 1;
        idx->vals[it] = occ.a[occ.n - 1];
        --occ.n;
      } else {
        idx->vals[it] = (nonsingle << 32) | nprev;
        nonsingle += nprev;
      }
      nprev = 1;
    } else {
      ++nprev;
    }
    if (last) break;
    v64_push(&occ, mm[mi].hit);
    prev = cur;
  }
  free(mm);
  idx->n_keys = idx->size;
  idx->n_occ = (uint32_t)occ.n;
  idx->occ = occ.a;
  return 0;
}

int ora_index_save(const char *path, const ora_index *idx) { /* index.cc:91-130 */
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  fwrite(&idx->k, sizeof(int), 1, f);
  fwrite(&idx->w, sizeof(int), 1, f);
  uint32_t sz = idx->size;
  fwrite(&sz, 4, 1, f);
  fwrite(&idx->n_buckets, 4, 1, f);
  fwrite(&idx->size, 4, 1, f);
  fwrite(&idx->n_occupied, 4, 1, f);
  fwrite(&idx->upper_bound, 4, 1, f);
  if (idx->n_buckets) {
    fwrite(idx->flags, 4, FL_SIZE(idx->n_buckets), f);
    fwrite(idx->keys, 8, idx->n_buckets, f);
    fwrite(idx->vals, 8, idx->n_buckets, f);
  }
  fwrite(&idx->n_occ, 4, 1, f);
  if (idx->n_occ) fwrite(idx->occ, 8, idx->n_occ, f);
  fclose(f);
  return 0;
}

int ora_index_load(const char *path, ora_index *idx) { /* index.cc:132-169, khash.h:358-373 */
  memset(idx, 0, sizeof(*idx));
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  int ok = 1;
  ok &= fread(&idx->k, sizeof(int), 1, f) == 1;
  ok &= fread(&idx->w, sizeof(int), 1, f) == 1;
  ok &= fread(&idx->n_keys, 4, 1, f) == 1;
  ok &= fread(&idx->n_buckets, 4, 1, f) == 1;
  ok &= fread(&idx->size, 4, 1, f) == 1;
  ok &= fread(&idx->n_occupied, 4, 1, f) == 1;
  ok &= fread(&idx->upper_bound, 4, 1, f) == 1;
  if (!ok) { fclose(f); return -2; }
  if (idx->n_buckets) {
    size_t fs = FL_SIZE(idx->n_buckets);
    idx->flags = (uint32_t *)malloc(fs * 4);
    idx->keys = (uint64_t *)malloc((size_t)idx->n_buckets * 8);
    idx->vals = (uint64_t *)malloc((size_t)idx->n_buckets * 8);
    ok &= fread(idx->flags, 4, fs, f) == fs;
    ok &= fread(idx->keys, 8, idx->n_buckets, f) == idx->n_buckets;
    ok &= fread(idx->vals, 8, idx->n_buckets, f) == idx->n_buckets;
  }
  ok &= fread(&idx->n_occ, 4, 1, f) == 1;
  if (ok && idx->n_occ) {
    idx->occ = (uint64_t *)malloc((size_t)idx->n_occ * 8);
    ok &= fread(idx->occ, 8, idx->n_occ, f) == idx->n_occ;
  }
  fclose(f);
  return ok ? 0 : -2;
}

/* Builds an ora_index from the device layout exported by cmgpu_export_index (interleaved
 * {key,val} buckets, all-ones key = empty).  Test/bench plumbing, not a reference function. */
int ora_index_from_buckets(const uint64_t *buckets, uint32_t n_buckets, const uint64_t *occ, uint32_t n_occ,
                           int k, int w, ora_index *idx) {
  memset(idx, 0, sizeof(*idx));
  idx->k = k; idx->w = w; idx->n_buckets = n_buckets; idx->n_occ = n_occ;
  const size_t fs = FL_SIZE(n_buckets);
  idx->flags = (uint32_t *)malloc(fs * 4);
  idx->keys = (uint64_t *)malloc((size_t)n_buckets * 8);
  idx->vals = (uint64_t *)malloc((size_t)n_buckets * 8);
  idx->occ = (uint64_t *)malloc((size_t)(n_occ ? n_occ : 1) * 8);
  if (!idx->flags || !idx->keys || !idx->vals || !idx->occ) return -1;
  memset(idx->flags, 0xaa, fs * 4);
  uint32_t sz = 0;
#pragma omp parallel for reduction(+ : sz) schedule(static, 1 << 16)
  for (long i = 0; i < (long)n_buckets; ++i) {
    idx->keys[i] = buckets[2 * (size_t)i];
    idx->vals[i] = buckets[2 * (size_t)i + 1];
    if (buckets[2 * (size_t)i] != UINT64_MAX) ++sz;
  }
  for (uint32_t i = 0; i < n_buckets; ++i)
    if (idx->keys[i] != UINT64_MAX) FL_SET_BOTH_FALSE(idx->flags, i);
  if (n_occ) memcpy(idx->occ, occ, (size_t)n_occ * 8);
  idx->size = idx->n_occupied = idx->n_keys = sz;
  idx->upper_bound = (uint32_t)(n_buckets * KH_UPPER + 0.5);
  return 0;
}

void ora_index_free(ora_index *idx) {
  free(idx->flags); free(idx->keys); free(idx->vals); free(idx->occ);
  memset(idx, 0, sizeof(*idx));
}

/* ------------------------------------------------------------------------- */
/* FASTA/FASTQ reading with kseq.h record semantics                           */
/* ------------------------------------------------------------------------- */
typedef struct { char *a; size_t n, cap; } vchar;
static void vch_append(vchar *v, const char *s, size_t l) {
  if (v->n + l + 1 > v->cap) {
    while (v->n + l + 1 > v->cap) v->cap = v->cap ? v->cap * 2 : 1 << 16;
    v->a = (char *)realloc(v->a, v->cap);
  }
  memcpy(v->a + v->n, s, l);
  v->n += l;
  v->a[v->n] = 0;
}

/* Generic reader: calls cb(name, seq, len) per record. Supports multi-line FASTA and
 * 4-line (or multi-line) FASTQ like kseq_read (kseq.h:175-217). Plain text only. */
typedef void (*rec_cb)(void *ud, const char *name, const char *seq, size_t len);
static long read_fastx(const char *path, rec_cb cb, void *ud) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  char *line = NULL;
  size_t lcap = 0;
  ssize_t ll;
  vchar name = {0}, seq = {0};
  long nrec = 0;
  int have = 0, in_qual = 0;
  size_t qual_len = 0;
  while ((ll = getline(&line, &lcap, f)) >= 0) {
    while (ll > 0 && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
    if (in_qual) {
      qual_len += (size_t)ll;
      if (qual_len >= seq.n) in_qual = 0;
      continue;
    }
    if (line[0] == '>' || line[0] == '@') {
      if (have) { cb(ud, name.a, seq.a ? seq.a : "", seq.n); ++nrec; }
      size_t e = 1;
      while (line[e] && line[e] != ' ' && line[e] != '\t' && line[e] != '\v' && line[e] != '\f') ++e;
      name.n = 0;
      vch_append(&name, line + 1, e - 1);
      seq.n = 0;
      if (seq.a) seq.a[0] = 0;
      have = 1;
    } else if (line[0] == '+' && have) {
      in_qual = 1;
      qual_len = 0;
      if (seq.n == 0) in_qual = 0;
    } else if (have) {
      /* kseq keeps only isgraph() characters of sequence lines */
      size_t o = 0;
      for (ssize_t i = 0; i < ll; ++i)
        if (line[i] > 32 && line[i] < 127) line[o++] = line[i];
      vch_append(&seq, line, o);
    }
  }
  if (have) { cb(ud, name.a, seq.a ? seq.a : "", seq.n); ++nrec; }
  free(line); free(name.a); free(seq.a);
  fclose(f);
  return nrec;
}

static void ref_cb(void *ud, const char *name, const char *seq, size_t len) {
  ora_ref *r = (ora_ref *)ud;
  if (len == 0) return; /* sequence_batch.cc:91 skips empty records */
  r->name = (char **)realloc(r->name, (r->n_seq + 1) * sizeof(char *));
  r->seq = (char **)realloc(r->seq, (r->n_seq + 1) * sizeof(char *));
  r->len = (uint32_t *)realloc(r->len, (r->n_seq + 1) * sizeof(uint32_t));
  r->name[r->n_seq] = strdup(name);
  /* zero padding after the sequence: the reference over-reads up to e bytes past the
   * end in GetRefStartEndPositionForReadFromMapping (mapping_generator.h:703-708); the
   * byte at [len] is kseq's NUL, what follows is undefined there, zeros here. */
  r->seq[r->n_seq] = (char *)calloc(len + 64, 1);
  memcpy(r->seq[r->n_seq], seq, len);
  r->len[r->n_seq] = (uint32_t)len;
  ++r->n_seq;
}

int ora_ref_load(const char *fasta_path, ora_ref *ref) {
  memset(ref, 0, sizeof(*ref));
  return read_fastx(fasta_path, ref_cb, ref) < 0 ? -1 : 0;
}

void ora_ref_free(ora_ref *ref) {
  for (uint32_t i = 0; i < ref->n_seq; ++i) { free(ref->name[i]); free(ref->seq[i]); }
  free(ref->name); free(ref->seq); free(ref->len);
  memset(ref, 0, sizeof(*ref));
}

typedef struct { vchar b; uint32_t *off; size_t n, cap; } fq_acc;
static void fq_cb(void *ud, const char *name, const char *seq, size_t len) {
  (void)name;
  fq_acc *a = (fq_acc *)ud;
  if (len == 0) return; /* sequence_batch.cc:27-30 skips zero-length reads */
  if (a->n + 2 > a->cap) {
    a->cap = a->cap ? a->cap * 2 : 1024;
    a->off = (uint32_t *)realloc(a->off, a->cap * sizeof(uint32_t));
  }
  if (a->n == 0) a->off[0] = 0;
  vch_append(&a->b, seq, len);
  a->off[++a->n] = (uint32_t)a->b.n;
}

long ora_read_fastx(const char *path, char **bases, uint32_t **off) {
  fq_acc a;
  memset(&a, 0, sizeof(a));
  long r = read_fastx(path, fq_cb, &a);
  if (r < 0) return r;
  if (a.n == 0) { a.off = (uint32_t *)calloc(1, sizeof(uint32_t)); a.b.a = (char *)calloc(1, 1); }
  *bases = a.b.a;
  *off = a.off;
  return (long)a.n;
}

/* ------------------------------------------------------------------------- */
/* std::mt19937 + libstdc++-11 uniform_int_distribution<int>(0,i)              */
/* (mapping_generator.h:199-214 draws with these; bits/uniform_int_dist.h _S_nd) */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *g, uint32_t s) {
  g->mt[0] = s;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}
static uint32_t mt_next(mt19937_t *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}
/* uniform int in [0, hi] (Lemire's nearly-divisionless, as in libstdc++ 11) */
static int mt_uniform(mt19937_t *g, int hi) {
  uint32_t range = (uint32_t)hi + 1u;
  if (range == 0) return (int)mt_next(g);
  uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (uint32_t)(-range) % range;
    while (low < threshold) {
      product = (uint64_t)mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (int)(product >> 32);
}

/* ------------------------------------------------------------------------- */
/* per-read scratch (mapping_metadata.h:144-165)                               */
/* ------------------------------------------------------------------------- */
typedef struct {
  uint64_t *mm_hash, *mm_hit;
  int n_mm, mm_cap;
  vec64 pos_hits, neg_hits;
  vcand pos_cand, neg_cand, pos_buf, neg_buf;
  vdraft pos_map, neg_map;
  vec64 pos_split, neg_split; /* split sites (int), parallel to pos_map/neg_map */
  int min_err, second_err, n_best, n_second;
  uint32_t rep_len;
} meta_t;

static void meta_prepare(meta_t *m, uint32_t read_len) {
  if ((int)read_len + 1 > m->mm_cap) {
    m->mm_cap = (int)read_len + 64;
    m->mm_hash = (uint64_t *)realloc(m->mm_hash, m->mm_cap * sizeof(uint64_t));
    m->mm_hit = (uint64_t *)realloc(m->mm_hit, m->mm_cap * sizeof(uint64_t));
  }
  m->n_mm = 0;
  m->pos_hits.n = m->neg_hits.n = 0;
  m->pos_cand.n = m->neg_cand.n = m->pos_buf.n = m->neg_buf.n = 0;
  m->pos_map.n = m->neg_map.n = 0;
  m->pos_split.n = m->neg_split.n = 0;
  m->rep_len = 0;
}
static void meta_free(meta_t *m) {
  free(m->mm_hash); free(m->mm_hit); free(m->pos_hits.a); free(m->neg_hits.a);
  free(m->pos_cand.a); free(m->neg_cand.a); free(m->pos_buf.a); free(m->neg_buf.a);
  free(m->pos_map.a); free(m->neg_map.a); free(m->pos_split.a); free(m->neg_split.a);
}

struct ora_ctx {
  const ora_index *idx;
  const ora_ref *ref;
  ora_params p;
  mt19937_t rng;
  ora_trace *trace;
  /* single-cell run (set by ora_map_pairs_bc for the duration of the call) */
  const ora_whitelist *wl;
  char *bc;
  const char *bcq;
  const uint32_t *bco;
  uint64_t *bc_key;
  uint64_t n_in_wl, n_corr;
  /* --chr-order (ora_set_chr_order): rank of every reference sequence and the reference reordered by it */
  uint32_t *rid_rank;
  ora_ref ref_ranked;
  uint32_t *pairs_rank; /* --pairs-natural-chr-order: rank used by the pairs flip (mapping_generator.cc:193-203) */
  /* --SAM run (set by ora_map_*_sam for the duration of the call) */
  ora_sam_record *sam_rec;
  uint32_t *sam_cigar;
  char *sam_md;
  uint32_t sam_md_cap, sam_first_read_id;
  uint32_t *sam_len;
};

void ora_default_params(ora_params *p) { /* mapping_parameters.h:19-61 */
  memset(p, 0, sizeof(*p));
  p->error_threshold = 8;
  p->min_num_seeds = 2;
  p->max_seed_freq0 = 500;
  p->max_seed_freq1 = 1000;
  p->max_insert_size = 1000;
  p->min_read_length = 30;
  p->max_num_best_mappings = 1;
  p->drop_repetitive_reads = 500000;
  p->mapq_threshold = 30;
  p->bc_error_threshold = 1;
  p->bc_probability_threshold = 0.9;
}

void ora_preset(ora_params *p, const char *preset) { /* chromap_driver.cc:247-275 */
  if (!strcmp(preset, "atac")) {
    p->max_insert_size = 2000;
    p->trim_adapters = 1;
    p->remove_pcr_duplicates = 1;
    p->tn5_shift = 1;
    p->low_mem = 1;
  } else if (!strcmp(preset, "chip")) {
    p->max_insert_size = 2000;
    p->remove_pcr_duplicates = 1;
    p->low_mem = 1;
  } else if (!strcmp(preset, "hic")) {
    p->error_threshold = 4;
    p->mapq_threshold = 1;
    p->split_alignment = 1;
    p->low_mem = 1;
  }
}

ora_ctx *ora_create(const ora_index *idx, const ora_ref *ref, const ora_params *p) {
  ora_ctx *c = (ora_ctx *)calloc(1, sizeof(ora_ctx));
  c->idx = idx;
  c->ref = ref;
  c->p = *p;
  mt_seed(&c->rng, 11);
  return c;
}
/* --chr-order (Chromap::GenerateCustomRidRanks chromap.cc:867-913, SequenceBatch::ReorderSequences, chromap.h:654-659):
 * rank[i] = position of reference sequence i in the custom order.  From here on the context sees the
 * reference reordered by rank (verification, coordinates, records, output names); candidates keep index
 * rids until RerankCandidatesRid (chromap.cc:916-923) rewrites them right before verification. */
int ora_set_chr_order(ora_ctx *c, const uint32_t *rank, uint32_t n) {
  if (n != c->ref->n_seq) return -1;
  c->rid_rank = (uint32_t *)malloc((size_t)n * 4);
  memcpy(c->rid_rank, rank, (size_t)n * 4);
  c->ref_ranked.n_seq = n;
  c->ref_ranked.name = (char **)calloc(n, sizeof(char *));
  c->ref_ranked.seq = (char **)calloc(n, sizeof(char *));
  c->ref_ranked.len = (uint32_t *)calloc(n, 4);
  for (uint32_t i = 0; i < n; ++i) {
    if (rank[i] >= n || c->ref_ranked.seq[rank[i]]) return -1;
    c->ref_ranked.name[rank[i]] = c->ref->name[i];
    c->ref_ranked.seq[rank[i]] = c->ref->seq[i];
    c->ref_ranked.len[rank[i]] = c->ref->len[i];
  }
  c->ref = &c->ref_ranked;
  return 0;
}
const ora_ref *ora_ctx_ref(const ora_ctx *c) { return c->ref; }
int ora_set_pairs_chr_order(ora_ctx *c, const uint32_t *rank, uint32_t n) {
  if (n != c->ref->n_seq) return -1;
  c->pairs_rank = (uint32_t *)malloc((size_t)n * 4);
  memcpy(c->pairs_rank, rank, (size_t)n * 4);
  return 0;
}
static void rerank_candidates(const ora_ctx *c, vcand *v) {
  if (!c->rid_rank) return;
  for (size_t i = 0; i < v->n; ++i) {
    const uint64_t rid = c->rid_rank[(uint32_t)(v->a[i].position >> 32)];
    v->a[i].position = (v->a[i].position & 0xffffffffull) | (rid << 32);
  }
}

void ora_destroy(ora_ctx *c) { free(c); }
void ora_set_trace(ora_ctx *c, ora_trace *t) { c->trace = t; }

/* ------------------------------------------------------------------------- */
/* K2: Index::GenerateCandidatePositions (index.cc:237-349)                    */
/* ------------------------------------------------------------------------- */
typedef struct { uint32_t rep_len, prev_pos; int count; } rep_stats; /* index_utils.h:21-26 */

static void update_rep(const ora_index *idx, uint32_t read_pos, rep_stats *s) { /* index.cc:507-523 */
  if (s->prev_pos > read_pos) {
    s->rep_len += idx->k;
  } else {
    if (read_pos < s->prev_pos + idx->k + idx->w - 1) s->rep_len += read_pos - s->prev_pos;
    else s->rep_len += idx->k;
  }
  s->prev_pos = read_pos;
  ++s->count;
}

/* index.cc:491-505 */
static inline uint64_t cand_from_hits(const ora_index *idx, uint64_t ref_hit, uint64_t read_hit) {
  const uint32_t ref_pos = (uint32_t)(ref_hit >> 1), read_pos = (uint32_t)(read_hit >> 1);
  const uint32_t start = ((ref_hit & 1) == (read_hit & 1)) ? ref_pos - read_pos
                                                          : ref_pos + read_pos - (uint32_t)idx->k + 1;
  return ((uint64_t)(uint32_t)(ref_hit >> 33) << 32) | start;
}

static int gen_candidate_positions(const ora_index *idx, meta_t *m, uint32_t max_freq,
                                   uint32_t rep_freq, ora_stats *st) {
  rep_stats rs = {0, UINT32_MAX, 0};
  for (int mi = 0; mi < m->n_mm; ++mi) {
    uint64_t steps = 0;
    uint32_t it = ora_kh_get(idx, m->mm_hash[mi] << 1, &steps);
    if (st) { st->probe_steps += steps; st->lookups++; }
    if (it == idx->n_buckets) continue;
    const uint64_t key = idx->keys[it], val = idx->vals[it], read_hit = m->mm_hit[mi];
    if (key & 1) { /* singleton: value is the reference hit, :277-287 */
      const uint64_t cp = cand_from_hits(idx, val, read_hit);
      if ((val & 1) == (read_hit & 1)) v64_push(&m->pos_hits, cp);
      else v64_push(&m->neg_hits, cp);
      continue;
    }
    const uint32_t n_occ = (uint32_t)val;
    if (!(n_occ >= max_freq)) { /* !IsFrequentSeed, :291-310 */
      const uint32_t off = (uint32_t)(val >> 32);
      if (st) st->occ_reads += n_occ;
      for (uint32_t oi = 0; oi < n_occ; ++oi) {
        const uint64_t rh = idx->occ[off + oi];
        const uint64_t cp = cand_from_hits(idx, rh, read_hit);
        if ((rh & 1) == (read_hit & 1)) v64_push(&m->pos_hits, cp);
        else v64_push(&m->neg_hits, cp);
      }
    }
    if (n_occ >= rep_freq) update_rep(idx, (uint32_t)(read_hit >> 1), &rs); /* :312-315 */
  }
  /* :318-334 -- both branches (std::sort, or per-list sort + heap merge) yield the
   * ascending sorted multiset. */
  qsort(m->pos_hits.a, m->pos_hits.n, 8, cmp_u64);
  qsort(m->neg_hits.a, m->neg_hits.n, 8, cmp_u64);
  m->rep_len = rs.rep_len;
  return rs.count;
}

/* K3b: CandidateProcessor::GenerateCandidatesOnOneStrand (candidate_processor.cc:283-342) */
static void gen_candidates_one_strand(int e, int seeds_required, uint32_t num_minimizers,
                                      vec64 *hits, vcand *out) {
  v64_push(hits, UINT64_MAX);
  int mcount = 1, equal = 1, best_equal = 1;
  uint64_t prev_hit = hits->a[0];
  uint32_t prev_rid = (uint32_t)(prev_hit >> 32), prev_pos = (uint32_t)prev_hit;
  uint64_t best_local = hits->a[0];
  for (size_t pi = 1; pi < hits->n; ++pi) {
    const uint64_t h = hits->a[pi];
    const uint32_t rid = (uint32_t)(h >> 32), pos = (uint32_t)h;
    if (rid != prev_rid || pos > prev_pos + (uint32_t)e ||
        ((uint32_t)mcount >= num_minimizers && pos > (uint32_t)best_local + (uint32_t)e)) {
      if (mcount >= seeds_required) {
        cand_t c;
        c.position = best_local;
        c.count = (uint8_t)best_equal;
        vc_push(out, c);
      }
      mcount = 1; equal = 1; best_equal = 1;
      best_local = h;
    } else {
      if (h == best_local) { ++equal; ++best_equal; }
      else if (h == prev_hit) {
        ++equal;
        if (equal > best_equal) { best_local = prev_hit; best_equal = equal; }
      } else equal = 1;
      ++mcount;
    }
    prev_hit = h; prev_rid = rid; prev_pos = pos;
  }
}

/* K3a: CandidateProcessor::GenerateCandidates (candidate_processor.cc:12-71) */
static void gen_candidates(const ora_ctx *c, meta_t *m, ora_stats *st) {
  const ora_params *p = &c->p;
  m->rep_len = 0;
  int rep_count = gen_candidate_positions(c->idx, m, p->max_seed_freq0, p->max_seed_freq0, st);
  int use_high = 0;
  if (m->pos_hits.n + m->neg_hits.n == 0) {
    m->pos_hits.n = m->neg_hits.n = 0;
    m->rep_len = 0;
    rep_count = gen_candidate_positions(c->idx, m, p->max_seed_freq1, p->max_seed_freq0, st);
    use_high = 1;
    if (m->pos_hits.n == 0 || m->neg_hits.n == 0) use_high = 0;
  }
  int req = m->n_mm - rep_count;
  req = req > 1 ? req : 1;
  req = req > p->min_num_seeds ? p->min_num_seeds : req;
  if (use_high) req = p->min_num_seeds;
  gen_candidates_one_strand(p->error_threshold, req, m->n_mm, &m->pos_hits, &m->pos_cand);
  gen_candidates_one_strand(p->error_threshold, req, m->n_mm, &m->neg_hits, &m->neg_cand);
}

/* K2': Index::GenerateCandidatePositionsFromRepetitiveReadWithMateInfoOnOneStrand
 * (index.cc:351-489). strand: 0 = kPositive, 1 = kNegative. */
static int rescue_positions(const ora_index *idx, int strand, uint32_t search_range, int min_seeds,
                            int max_freq0, const meta_t *m, const vcand *mate,
                            uint32_t *rep_len, vec64 *out, ora_stats *st) {
  const uint32_t ms = (uint32_t)mate->n;
  int max_count = 0, best_num = 0;
  for (uint32_t i = 0; i < ms; ++i) {
    int cnt = mate->a[i].count;
    if (cnt > max_count) { max_count = cnt; best_num = 1; }
    else if (cnt == max_count) ++best_num;
  }
  if (best_num >= 300 || ms > (uint32_t)max_freq0 || (max_count <= min_seeds && best_num >= 200))
    return -max_count;
  uint64_t *bs = (uint64_t *)malloc((size_t)(best_num + 1) * 2 * sizeof(uint64_t));
  uint32_t raw = 0;
  for (uint32_t ci = 0; ci < ms; ++ci) {
    if (mate->a[ci].count == max_count) {
      const uint64_t pos = mate->a[ci].position;
      bs[2 * raw] = pos < search_range ? 0 : pos - search_range;
      bs[2 * raw + 1] = pos + search_range;
      ++raw;
    }
  }
  if (raw == 0) { free(bs); return max_count; }
  uint32_t nb = 1;
  for (uint32_t bi = 1; bi < raw; ++bi) { /* :399-411 */
    if (bs[2 * (nb - 1) + 1] < bs[2 * bi]) {
      bs[2 * nb] = bs[2 * bi];
      bs[2 * nb + 1] = bs[2 * bi + 1];
      ++nb;
    } else {
      bs[2 * (nb - 1) + 1] = bs[2 * bi + 1];
    }
  }
  rep_stats rs = {0, UINT32_MAX, 0};
  for (int mi = 0; mi < m->n_mm; ++mi) {
    uint64_t steps = 0;
    uint32_t it = ora_kh_get(idx, m->mm_hash[mi] << 1, &steps);
    if (st) { st->probe_steps += steps; st->lookups++; }
    if (it == idx->n_buckets) continue;
    const uint64_t key = idx->keys[it], val = idx->vals[it], read_hit = m->mm_hit[mi];
    const uint32_t read_pos = (uint32_t)(read_hit >> 1);
    if (key & 1) {
      const int same = (val & 1) == (read_hit & 1);
      if ((same && strand == 0) || (!same && strand == 1)) v64_push(out, cand_from_hits(idx, val, read_hit));
      continue;
    }
    const uint32_t off = (uint32_t)(val >> 32), n_occ = (uint32_t)val;
    int32_t prev_l = 0;
    for (uint32_t bi = 0; bi < nb; ++bi) {
      int32_t l = prev_l, mm_ = 0, r = (int32_t)(n_occ - 1);
      const uint64_t boundary = bs[2 * bi];
      while (l <= r) {
        mm_ = (l + r) / 2;
        const uint64_t cp = idx->occ[off + mm_] >> 1;
        if (st) st->occ_reads++;
        if (cp < boundary) l = mm_ + 1;
        else if (cp > boundary) r = mm_ - 1;
        else break;
      }
      prev_l = mm_;
      for (uint32_t oi = (uint32_t)mm_; oi < n_occ; ++oi) {
        const uint64_t rh = idx->occ[off + oi];
        if (st) st->occ_reads++;
        if ((rh >> 1) > bs[2 * bi + 1]) break;
        const int same = (rh & 1) == (read_hit & 1);
        if ((same && strand == 0) || (!same && strand == 1)) v64_push(out, cand_from_hits(idx, rh, read_hit));
      }
    }
    if (n_occ >= (uint32_t)max_freq0) update_rep(idx, read_pos, &rs);
  }
  free(bs);
  qsort(out->a, out->n, 8, cmp_u64);
  *rep_len = rs.rep_len;
  return max_count;
}

/* CandidateProcessor::MergeCandidates (candidate_processor.cc:345-414) */
static void merge_candidates(int e, vcand *c1, vcand *c2, vcand *buffer) {
  if (c1->n == 0) { vcand t = *c1; *c1 = *c2; *c2 = t; return; }
  size_t i = 0, j = 0;
  buffer->n = 0;
#define BACK_OK(P) (buffer->n == 0 || (P) > buffer->a[buffer->n - 1].position + (uint64_t)(int64_t)e)
  while (i < c1->n && j < c2->n) {
    if (c1->a[i].position == c2->a[j].position) {
      if (BACK_OK(c1->a[i].position)) {
        if (c1->a[i].count > c2->a[j].count) vc_push(buffer, c1->a[i]);
        else vc_push(buffer, c2->a[j]);
      }
      ++i; ++j;
    } else if (c1->a[i].position < c2->a[j].position) {
      if (BACK_OK(c1->a[i].position)) vc_push(buffer, c1->a[i]);
      ++i;
    } else {
      if (BACK_OK(c2->a[j].position)) vc_push(buffer, c2->a[j]);
      ++j;
    }
  }
  while (i < c1->n) { if (BACK_OK(c1->a[i].position)) vc_push(buffer, c1->a[i]); ++i; }
  while (j < c2->n) { if (BACK_OK(c2->a[j].position)) vc_push(buffer, c2->a[j]); ++j; }
#undef BACK_OK
  vcand t = *c1; *c1 = *buffer; *buffer = t;
}

/* K3c: CandidateProcessor::SupplementCandidates (candidate_processor.cc:75-231) */
static int supplement_candidates(const ora_ctx *c, meta_t *m1, meta_t *m2, ora_stats *st) {
  const ora_params *p = &c->p;
  const uint32_t search_range = 2u * (uint32_t)p->max_insert_size;
  vcand aug_pos[2] = {{0}, {0}}, aug_neg[2] = {{0}, {0}};
  int ret = 0;
  for (int mate = 0; mate <= 1; ++mate) {
    meta_t *m = mate == 0 ? m1 : m2;
    meta_t *o = mate == 0 ? m2 : m1;
    const uint32_t mm_count = (uint32_t)m->n_mm;
    int augment = 1;
    for (size_t i = 0; i < m->pos_cand.n; ++i)
      if (m->pos_cand.a[i].count >= mm_count / 2) { augment = 0; break; }
    if (augment)
      for (size_t i = 0; i < m->neg_cand.n; ++i)
        if (m->neg_cand.a[i].count >= mm_count / 2) { augment = 0; break; }
    if (augment) {
      m->pos_hits.n = m->neg_hits.n = 0;
      int pos_res = 0, neg_res = 0;
      if (o->pos_cand.n > 0) { /* mate + candidates drive a search on our - strand */
        if (st) st->num_rescue++;
        pos_res = rescue_positions(c->idx, 1, search_range, p->min_num_seeds, p->max_seed_freq0, m,
                                   &o->pos_cand, &m->rep_len, &m->neg_hits, st);
        gen_candidates_one_strand(p->error_threshold, 1, m->n_mm, &m->neg_hits, &aug_neg[mate]);
      }
      if (o->neg_cand.n > 0) {
        if (st) st->num_rescue++;
        neg_res = rescue_positions(c->idx, 0, search_range, p->min_num_seeds, p->max_seed_freq0, m,
                                   &o->neg_cand, &m->rep_len, &m->pos_hits, st);
        gen_candidates_one_strand(p->error_threshold, 1, m->n_mm, &m->pos_hits, &aug_pos[mate]);
      }
      if (((pos_res < 0 && neg_res > 0 && -pos_res >= neg_res) ||
           (pos_res > 0 && neg_res < 0 && pos_res <= -neg_res)) &&
          m->pos_cand.n + m->neg_cand.n == 0)
        ret = 1;
    }
  }
  if (aug_pos[0].n > 0) merge_candidates(p->error_threshold, &m1->pos_cand, &aug_pos[0], &m1->pos_buf);
  if (aug_neg[0].n > 0) merge_candidates(p->error_threshold, &m1->neg_cand, &aug_neg[0], &m1->neg_buf);
  if (aug_pos[1].n > 0) merge_candidates(p->error_threshold, &m2->pos_cand, &aug_pos[1], &m2->pos_buf);
  if (aug_neg[1].n > 0) merge_candidates(p->error_threshold, &m2->neg_cand, &aug_neg[1], &m2->neg_buf);
  free(aug_pos[0].a); free(aug_pos[1].a); free(aug_neg[0].a); free(aug_neg[1].a);
  return ret;
}

/* K3d: ReduceCandidatesForPairedEndReadOnOneDirection (candidate_processor.cc:416-484) */
static void reduce_one_direction(uint32_t dist, const vcand *c1, const vcand *c2, vcand *f1, vcand *f2) {
  uint32_t i1 = 0, i2 = 0;
  int unpaired1 = 0, unpaired2 = 0;
  const int unpaired_thr = 5;
  int max1 = 6, max2 = 6;
  uint32_t prev_end_i2 = 0;
  while (i1 < c1->n && i2 < c2->n) {
    if (c1->a[i1].position > c2->a[i2].position + dist) {
      if (i2 >= prev_end_i2 && unpaired2 < unpaired_thr &&
          (c1->a[i1].position >> 32) == (c2->a[i2].position >> 32) && c2->a[i2].count >= max2) {
        vc_push(f2, c2->a[i2]);
        ++unpaired2;
      }
      ++i2;
    } else if (c2->a[i2].position > c1->a[i1].position + dist) {
      if (unpaired1 < unpaired_thr && (c1->a[i1].position >> 32) == (c2->a[i2].position >> 32) &&
          c1->a[i1].count >= max1) {
        vc_push(f1, c1->a[i1]);
        ++unpaired1;
      }
      ++i1;
    } else {
      vc_push(f1, c1->a[i1]);
      if (c1->a[i1].count > max1) max1 = c1->a[i1].count;
      uint32_t cur = i2;
      while (cur < c2->n && c2->a[cur].position <= c1->a[i1].position + dist) {
        if (cur >= prev_end_i2) {
          vc_push(f2, c2->a[cur]);
          if (c2->a[cur].count > max2) max2 = c2->a[cur].count;
        }
        ++cur;
      }
      prev_end_i2 = cur;
      ++i1;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* K4: verification                                                           */
/* ------------------------------------------------------------------------- */
/* BandedAlignPatternToText (alignment.cc:141-192) */
int ora_banded_align(int e, const char *pattern, const char *text, int read_length,
                     int *mapping_end_position) {
  uint32_t Peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) Peq[c2u(pattern[i])] |= (1u << i);
  const uint32_t hi = 1u << (2 * e), lo = 1;
  uint32_t VP = 0, VN = 0, X, D0, HN, HP;
  int err = 0;
  for (int i = 0; i < read_length; i++) {
    Peq[c2u(pattern[i + 2 * e])] |= hi;
    X = Peq[c2u(text[i])] | VN;
    D0 = ((VP + (X & VP)) ^ VP) | X;
    HN = VP & D0;
    HP = VN | ~(VP | D0);
    X = D0 >> 1;
    VN = X & HP;
    VP = HN | ~(X | HP);
    err += 1 - (int)(D0 & lo);
    if (err > 3 * e) return e + 1;
    for (int ai = 0; ai < 5; ai++) Peq[ai] >>= 1;
  }
  const int band_start = read_length - 1;
  int min_err = err;
  *mapping_end_position = band_start;
  for (int i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *mapping_end_position = band_start + 1 + i;
    }
  }
  return min_err;
}

/* BandedTraceback (alignment.cc:656-718) */
void ora_banded_traceback(int e, int min_num_errors, const char *pattern, const char *text,
                          int read_length, int *mapping_start_position) {
  if (min_num_errors == 0) { *mapping_start_position = e; return; }
  int error_count = 0;
  for (int i = 0; i < read_length; ++i)
    if (pattern[i + e] != text[i]) ++error_count; /* raw, case-sensitive compare (:666) */
  if (error_count == min_num_errors) { *mapping_start_position = e; return; }
  uint32_t Peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) Peq[c2u(pattern[read_length - 1 + 2 * e - i])] |= (1u << i);
  const uint32_t hi = 1u << (2 * e), lo = 1;
  uint32_t VP = 0, VN = 0, X, D0, HN, HP;
  int err = 0;
  for (int i = 0; i < read_length; i++) {
    Peq[c2u(pattern[read_length - 1 - i])] |= hi;
    X = Peq[c2u(text[read_length - 1 - i])] | VN;
    D0 = ((VP + (X & VP)) ^ VP) | X;
    HN = VP & D0;
    HP = VN | ~(VP | D0);
    X = D0 >> 1;
    VN = X & HP;
    VP = HN | ~(X | HP);
    err += 1 - (int)(D0 & lo);
    for (int ai = 0; ai < 5; ai++) Peq[ai] >>= 1;
  }
  *mapping_start_position = 2 * e;
  for (int i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err == min_num_errors) {
      *mapping_start_position = 2 * e - (1 + i);
      if (i + 1 == e) return;
    }
  }
}


/* BandedAlignPatternToTextWithDropOff (alignment.cc:197-283) */
static int banded_align_dropoff(int e, const char *pattern, const char *text, int read_length,
                                int *mapping_end_position, int *read_mapping_length) {
  uint32_t Peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) Peq[c2u(pattern[i])] |= (1u << i);
  const uint32_t hi = 1u << (2 * e), lo = 1;
  uint32_t VP = 0, VN = 0, X, D0, HN, HP, prev_VP = 0, prev_VN = 0;
  int err = 0, i = 0, fail_beginning = 0, prev_err = 0;
  for (; i < read_length; i++) {
    Peq[c2u(pattern[i + 2 * e])] |= hi;
    X = Peq[c2u(text[i])] | VN;
    D0 = ((VP + (X & VP)) ^ VP) | X;
    HN = VP & D0;
    HP = VN | ~(VP | D0);
    X = D0 >> 1;
    prev_VN = VN; prev_VP = VP;
    VN = X & HP;
    VP = HN | ~(X | HP);
    prev_err = err;
    err += 1 - (int)(D0 & lo);
    if (err > 2 * e) {
      if (i < 4 * e && i < read_length / 2) fail_beginning = 1;
      break;
    }
    for (int ai = 0; ai < 5; ai++) Peq[ai] >>= 1;
  }
  if (i < read_length) { err = prev_err; VN = prev_VN; VP = prev_VP; }
  const int band_start = i - 1;
  int min_err = err;
  *read_mapping_length = i;
  *mapping_end_position = band_start;
  for (i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *mapping_end_position = band_start + 1 + i;
    }
  }
  if (fail_beginning || (read_length > 60 && *mapping_end_position + 1 - e - min_err < 30))
    *mapping_end_position = -*mapping_end_position;
  return min_err;
}

/* BandedAlignPatternToTextWithDropOffFrom3End (alignment.cc:285-376) */
static int banded_align_dropoff_3end(int e, const char *pattern, const char *text, int read_length,
                                     int *mapping_end_position, int *read_mapping_length) {
  uint32_t Peq[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * e; i++) Peq[c2u(pattern[read_length + 2 * e - 1 - i])] |= (1u << i);
  const uint32_t hi = 1u << (2 * e), lo = 1;
  uint32_t VP = 0, VN = 0, X, D0, HN, HP, prev_VP = 0, prev_VN = 0;
  int err = 0, i = 0, fail_beginning = 0, prev_err = 0;
  for (; i < read_length; i++) {
    Peq[c2u(pattern[read_length - 1 - i])] |= hi;
    X = Peq[c2u(text[read_length - 1 - i])] | VN;
    D0 = ((VP + (X & VP)) ^ VP) | X;
    HN = VP & D0;
    HP = VN | ~(VP | D0);
    X = D0 >> 1;
    prev_VN = VN; prev_VP = VP;
    VN = X & HP;
    VP = HN | ~(X | HP);
    prev_err = err;
    err += 1 - (int)(D0 & lo);
    if (err > 2 * e) {
      if (i < 4 * e && i < read_length / 2) fail_beginning = 1;
      break;
    }
    for (int ai = 0; ai < 5; ai++) Peq[ai] >>= 1;
  }
  if (i < read_length) { err = prev_err; VN = prev_VN; VP = prev_VP; }
  const int band_start = i - 1;
  int min_err = err;
  *read_mapping_length = i;
  *mapping_end_position = band_start;
  for (i = 0; i < 2 * e; i++) {
    err += (int)((VP >> i) & 1u);
    err -= (int)((VN >> i) & 1u);
    if (err < min_err || (err == min_err && i + 1 == e)) {
      min_err = err;
      *mapping_end_position = band_start + (1 + i);
    }
  }
  if (fail_beginning || (read_length > 60 && *mapping_end_position + 1 - e - min_err < 30))
    *mapping_end_position = -*mapping_end_position;
  return min_err;
}

/* AdjustGapBeginning (alignment.cc:24-83), non-SAM use (no cigar). strand 0 = +.
 * read is NUL-terminated at read_total_len, ref at ref_len (the loops of the - branch
 * stop on the terminators). */
static int adjust_gap_beginning(int strand, const char *ref, uint32_t ref_len, const char *read, int read_total_len,
                                int *gap_beginning, int read_end, int ref_start_position, int ref_end_position) {
  int i, j;
  if (strand == 0) {
    if (*gap_beginning <= 0) return ref_start_position;
    for (i = *gap_beginning - 1, j = ref_start_position - 1; i >= 0 && j >= 0; --i, --j)
      if (read[i] != ref[j] && read[i] != ref[j] - 'a' + 'A') break;
    *gap_beginning = i + 1;
    return j + 1;
  }
  if (*gap_beginning <= 0) return ref_end_position;
  for (i = read_end + 1, j = ref_end_position + 1; i < read_total_len && (uint32_t)j < ref_len; ++i, ++j)
    if (read[i] != ref[j] && read[i] != ref[j] - 'a' + 'A') break;
  *gap_beginning = *gap_beginning + i - (read_end + 1);
  return j - 1;
}

/* DraftMappingGenerator::IsValidCandidate (draft_mapping_generator.cc:59-70) */
static inline int is_valid_candidate(const ora_ref *ref, int e, uint32_t rid, uint32_t position,
                                     uint32_t read_length) {
  const uint32_t rl = ref->len[rid];
  if (position < (uint32_t)e || position >= rl || position + read_length + (uint32_t)e >= rl) return 0;
  return 1;
}

static inline void update_best(meta_t *m, int ne) { /* draft_mapping_generator.cc:502-528 */
  if (ne < m->min_err) {
    m->second_err = m->min_err;
    m->n_second = m->n_best;
    m->min_err = ne;
    m->n_best = 1;
  } else if (ne == m->min_err) {
    m->n_best++;
  } else if (ne == m->second_err) {
    m->n_second++;
  } else if (ne < m->second_err) {
    m->n_second = 1;
    m->second_err = ne;
  }
}

/* One candidate through the scalar routine, pushing the draft mapping on success.
 * Returns 1 when accepted (num_errors <= e). */
static int verify_one(const ora_ctx *c, meta_t *m, int strand, const cand_t *cd, const char *read_seq,
                      uint32_t L, ora_stats *st) {
  const int e = c->p.error_threshold;
  const uint32_t rid = (uint32_t)(cd->position >> 32);
  uint32_t position = (uint32_t)cd->position;
  if (strand == 1) position = position - L + 1;
  int end_pos = (int)L; /* :400 */
  if (st) st->num_verifications++;
  const int ne = ora_banded_align(e, c->ref->seq[rid] + position - e, read_seq, (int)L, &end_pos);
  if (ne <= e) {
    update_best(m, ne);
    if (strand == 0) vd_push(&m->pos_map, ne, cd->position - (uint64_t)e + (uint64_t)(int64_t)end_pos);
    else vd_push(&m->neg_map, ne, cd->position - L + 1 - (uint64_t)e + (uint64_t)(int64_t)end_pos);
    return 1;
  }
  return 0;
}

/* GenerateDraftMappingsOnOneStrand, non-split branch (draft_mapping_generator.cc:359-557):
 * candidate_count_threshold stays 0 there, so every valid candidate is verified. */
static void draft_one_strand_scalar(const ora_ctx *c, meta_t *m, int strand, const char *read_seq,
                                    uint32_t L, ora_stats *st) {
  const vcand *cs = strand == 0 ? &m->pos_cand : &m->neg_cand;
  for (size_t ci = 0; ci < cs->n; ++ci) {
    const uint32_t rid = (uint32_t)(cs->a[ci].position >> 32);
    uint32_t position = (uint32_t)cs->a[ci].position;
    if (strand == 1) position = position - L + 1;
    if (!is_valid_candidate(c->ref, c->p.error_threshold, rid, position, L)) continue;
    verify_one(c, m, strand, &cs->a[ci], read_seq, L, st);
  }
}


/* GenerateDraftMappingsOnOneStrand, split-alignment branch (draft_mapping_generator.cc:359-557).
 * best_mapping_longest_match / longest_match are re-initialised per candidate in the
 * reference (:404-405), so the second_min adjustment at :511-515 never fires. */
static void draft_one_strand_split(const ora_ctx *c, meta_t *m, int strand, const char *read_seq, uint32_t L,
                                   ora_stats *st) {
  const vcand *cs = strand == 0 ? &m->pos_cand : &m->neg_cand;
  const int e = c->p.error_threshold;
  uint32_t thr = 0;
  for (size_t ci = 0; ci < cs->n; ++ci) {
    if (cs->a[ci].count < thr) break;
    const uint32_t rid = (uint32_t)(cs->a[ci].position >> 32);
    uint32_t position = (uint32_t)cs->a[ci].position;
    if (strand == 1) position = position - L + 1;
    if (!is_valid_candidate(c->ref, e, rid, position, L)) continue;
    int mep = (int)L, gap_beginning = 0, num_errors = 0, actual = 0, rml = 0;
    const int allow = 20 - e, len_thr = 30;
    const char *pat = c->ref->seq[rid] + position - e;
    if (st) st->num_verifications++;
    if (strand == 0) {
      num_errors = banded_align_dropoff(e, pat, read_seq, (int)L, &mep, &rml);
      if (mep < 0 && allow > 0) {
        const int b_err = num_errors, b_mep = -mep, b_rml = rml;
        num_errors = banded_align_dropoff(e, pat + allow, read_seq + allow, (int)L - allow, &mep, &rml);
        if (num_errors > e || mep < 0) { num_errors = b_err; mep = b_mep; rml = b_rml; }
        else { gap_beginning = allow; mep += gap_beginning; rml += gap_beginning; }
      }
    } else {
      num_errors = banded_align_dropoff_3end(e, pat, read_seq, (int)L, &mep, &rml);
      if (mep < 0 && allow > 0) {
        const int b_err = num_errors, b_mep = -mep, b_rml = rml;
        num_errors = banded_align_dropoff_3end(e, pat, read_seq, (int)L - allow, &mep, &rml);
        if (num_errors > e || mep < 0) { num_errors = b_err; mep = b_mep; rml = b_rml; }
        else { gap_beginning = allow; mep += gap_beginning; rml += gap_beginning; }
      }
    }
    if (mep + 1 - e - num_errors - gap_beginning >= len_thr) {
      actual = num_errors;
      num_errors = -(mep - e - num_errors - gap_beginning);
    } else {
      num_errors = e + 1;
      actual = e + 1;
    }
    if (num_errors <= e) {
      if (num_errors < m->min_err) {
        m->second_err = m->min_err; m->n_second = m->n_best; m->min_err = num_errors; m->n_best = 1;
        thr = cs->n > 50 ? cs->a[ci].count : cs->a[ci].count / 2;
      } else if (num_errors == m->min_err) m->n_best++;
      else if (num_errors == m->second_err) m->n_second++;
      else if (num_errors < m->second_err) { m->n_second = 1; m->second_err = num_errors; }
      if (strand == 0) {
        vd_push(&m->pos_map, num_errors, cs->a[ci].position - (uint64_t)e + (uint64_t)(int64_t)mep);
        v64_push(&m->pos_split, (uint64_t)(uint32_t)(((actual & 0xff) << 24) | ((gap_beginning & 0xff) << 16) | (rml & 0xffff)));
      } else {
        if (c->p.output_format == 1) /* --SAM keeps the non-split position rule (draft_mapping_generator.cc:535-547) */
          vd_push(&m->neg_map, num_errors, cs->a[ci].position - (uint64_t)L + 1 - (uint64_t)e + (uint64_t)(int64_t)mep);
        else
          vd_push(&m->neg_map, num_errors, cs->a[ci].position - (uint64_t)(int64_t)gap_beginning); /* :534-537 */
        v64_push(&m->neg_split, (uint64_t)(uint32_t)(((actual & 0xff) << 24) | ((gap_beginning & 0xff) << 16) | (rml & 0xffff)));
      }
    }
  }
}

/* GenerateDraftMappingsOnOneStrandUsingSIMD with 4 lanes (draft_mapping_generator.cc:159-357).
 * The 4-lane kernel (alignment.cc:378-501) returns, for every accepted lane, the same
 * (errors, end) as the scalar routine: a lane whose band-start count exceeds 3e keeps
 * running but can never come back under e (band cells differ from the band start by at
 * most 2e), so accept/reject and values agree; only the control flow below differs. */
static void draft_one_strand_lanes(const ora_ctx *c, meta_t *m, int strand, const char *read_seq,
                                   uint32_t L, int lanes, ora_stats *st) {
  const vcand *cs = strand == 0 ? &m->pos_cand : &m->neg_cand;
  const int e = c->p.error_threshold;
  cand_t valid[8];
  uint32_t nvalid = 0, thr = 0;
  size_t ci = 0;
  while (ci < cs->n) {
    if (cs->a[ci].count < thr) break; /* :186-188 */
    const uint32_t rid = (uint32_t)(cs->a[ci].position >> 32);
    uint32_t position = (uint32_t)cs->a[ci].position;
    if (strand == 1) position = position - L + 1;
    if (!is_valid_candidate(c->ref, e, rid, position, L)) { ++ci; continue; }
    valid[nvalid++] = cs->a[ci];
    ++ci;
    if (nvalid < (uint32_t)lanes) continue;
    for (int mi = 0; mi < lanes; ++mi) {
      if (!verify_one(c, m, strand, &valid[mi], read_seq, L, st)) thr = valid[mi].count; /* :299-301 */
    }
    nvalid = 0;
  }
  for (uint32_t i = 0; i < nvalid; ++i) verify_one(c, m, strand, &valid[i], read_seq, L, st); /* :308-356 */
}

/* DraftMappingGenerator::GenerateDraftMappings (draft_mapping_generator.cc:9-57) with the
 * shortcut GenerateNonSplitDraftMappingSupportedByAllMinimizers (:72-157). */
static void gen_draft_mappings(const ora_ctx *c, meta_t *m, const char *read, const char *neg_read,
                               uint32_t L, ora_stats *st) {
  const int e = c->p.error_threshold;
  m->min_err = e + 1; m->n_best = 0; m->second_err = e + 1; m->n_second = 0;
  if (!c->p.split_alignment && m->pos_cand.n + m->neg_cand.n == 1) {
    uint32_t n_all = 0, idx = 0;
    int strand = 0;
    for (size_t i = 0; i < m->pos_cand.n; ++i)
      if (m->pos_cand.a[i].count == (uint32_t)m->n_mm) { idx = (uint32_t)i; ++n_all; }
    for (size_t i = 0; i < m->neg_cand.n; ++i)
      if (m->neg_cand.a[i].count == (uint32_t)m->n_mm) { idx = (uint32_t)i; strand = 1; ++n_all; }
    if (n_all == 1) {
      m->min_err = 0; m->n_best = 1; m->n_second = 0; /* :122-124 (second_err stays e+1) */
      const cand_t *cd = strand == 0 ? &m->pos_cand.a[idx] : &m->neg_cand.a[idx];
      const uint32_t rid = (uint32_t)(cd->position >> 32);
      uint32_t position = strand == 0 ? (uint32_t)cd->position : (uint32_t)cd->position - L + 1;
      if (is_valid_candidate(c->ref, e, rid, position, L)) {
        if (strand == 0) vd_push(&m->pos_map, 0, cd->position + L - 1);
        else vd_push(&m->neg_map, 0, cd->position);
        if (st) st->num_shortcut++;
        return;
      }
      /* falls through with min_err = 0, n_best = 1 already set (reference behaviour) */
    }
  }
  qsort(m->pos_cand.a, m->pos_cand.n, sizeof(cand_t), cmp_cand); /* SortCandidates, mapping_metadata.h:65-68 */
  qsort(m->neg_cand.a, m->neg_cand.n, sizeof(cand_t), cmp_cand);
  if (c->p.split_alignment) { /* :31-39 */
    draft_one_strand_split(c, m, 0, read, L, st);
    draft_one_strand_split(c, m, 1, neg_read, L, st);
    return;
  }
  int lanes = e < 8 ? 8 : (e < 16 ? 4 : 0); /* GetNumVPULanes, mapping_parameters.h:80-88 */
  if (lanes == 0 || m->pos_cand.n < (size_t)lanes) draft_one_strand_scalar(c, m, 0, read, L, st);
  else draft_one_strand_lanes(c, m, 0, read, L, lanes, st);
  if (lanes == 0 || m->neg_cand.n < (size_t)lanes) draft_one_strand_scalar(c, m, 1, neg_read, L, st);
  else draft_one_strand_lanes(c, m, 1, neg_read, L, lanes, st);
}

/* ------------------------------------------------------------------------- */
/* K5: best mappings, coordinates, MAPQ                                        */
/* ------------------------------------------------------------------------- */
typedef struct {
  int min_sum, second_sum, n_best, n_second;
  vpair best[4]; /* [0] = F1R2, [1] = F2R1, split only: [2] = F1F2, [3] = R1R2 */
} pe_meta_t;

/* GenerateBestMappingsForPairedEndReadOnOneDirection, non-split (mapping_generator.h:347-484).
 * dir 0: read1 on + strand, read2 on -; dir 1: read1 -, read2 +. */
static void best_one_direction(const ora_ctx *c, int dir, const meta_t *m1, const meta_t *m2,
                               uint32_t len1, uint32_t len2, pe_meta_t *pe) {
  const vdraft *a = dir == 0 ? &m1->pos_map : &m1->neg_map;
  const vdraft *b = dir == 0 ? &m2->neg_map : &m2->pos_map;
  const uint64_t I = (uint64_t)(int64_t)c->p.max_insert_size;
  const uint64_t min_overlap = (uint32_t)c->p.min_read_length;
  vpair *best = &pe->best[dir];
  uint32_t i1 = 0, i2 = 0;
  while (i1 < a->n && i2 < b->n) {
    const uint64_t p1 = a->a[i1].position, p2 = b->a[i2].position;
    if ((dir == 1 && p1 > p2 + I - len2) || (dir == 0 && p1 > p2 + len1 - min_overlap)) {
      ++i2;
    } else if ((dir == 0 && p2 > p1 + I - len1) || (dir == 1 && p2 > p1 + len2 - min_overlap)) {
      ++i1;
    } else {
      uint32_t cur = i2;
      while (cur < b->n && ((dir == 0 && b->a[cur].position <= p1 + I - len1) ||
                            (dir == 1 && b->a[cur].position <= p1 + len2 - min_overlap))) {
        const int s = a->a[i1].num_errors + b->a[cur].num_errors;
        if (s < pe->min_sum) {
          pe->second_sum = pe->min_sum;
          pe->n_second = pe->n_best;
          pe->min_sum = s;
          pe->n_best = 1;
          best->n = 0;
          vp_push(best, i1, cur);
        } else if (s == pe->min_sum) {
          pe->n_best++;
          vp_push(best, i1, cur);
        } else if (s == pe->second_sum) {
          pe->n_second++;
        } else if (s < pe->second_sum) {
          pe->second_sum = s;
          pe->n_second = 1;
        }
        ++cur;
      }
      ++i1;
    }
  }
}

typedef struct { uint32_t rid, ref_start, ref_end; } span_t;

/* GetRefStartEndPositionForReadFromMapping, non-SAM non-split branches
 * (mapping_generator.h:657-717, 762-793, 855-916). strand 0 = +. */
static span_t ref_start_end(const ora_ctx *c, const draft_t *d, int strand, const char *read_seq, int L) {
  const int e = c->p.error_threshold;
  const uint32_t rid = (uint32_t)(d->position >> 32), ref_pos = (uint32_t)d->position;
  const uint32_t rl = c->ref->len[rid];
  uint32_t vw = ref_pos + 1 > (uint32_t)(L + e) ? ref_pos + 1 - (uint32_t)L - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)L;
  int start = strand == 0 ? 0 : e;
  ora_banded_traceback(e, d->num_errors, c->ref->seq[rid] + vw, read_seq, L, &start);
  span_t s;
  s.rid = rid;
  s.ref_start = vw + (uint32_t)start;
  s.ref_end = ref_pos; /* + strand: :790; - strand: vw + (ref_pos - vw + 1) - 1, :914-915 */
  return s;
}

/* ksw_semi_global3 (ksw.cc:505-626) with chromap's defaults (match 1, mismatch 4, gap open 6 /
 * extend 1 for both kinds, ambiguous base 0; mapping_parameters.h:20-23, mapping_generator.h:660-671):
 * affine-gap DP with the read as rows (target) and the reference window as columns (query).  Row
 * i only visits columns [i, min(i+w+1, qlen)); the first row may start free in columns 0..w; the
 * alignment ends in the best of the last w columns of the last row.  One byte per cell keeps the
 * three move bits (h: bits 0-1, e-extension: bit 2, f-extension: bit 5). */
int ora_ksw_semi_global3(int qlen, const char *query, int tlen, const char *target, int w, uint32_t *cigar, int cigar_cap,
                         int *n_cigar_out, int *start, int *end) {
  const int NEG = -0x40000000, o_del = 6, e_del = 1, o_ins = 6, e_ins = 1, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
  const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
  uint8_t *z = (uint8_t *)malloc((size_t)n_col * (size_t)(tlen > 0 ? tlen : 1));
  int *H = (int *)malloc(((size_t)qlen + 2) * sizeof(int)), *E = (int *)malloc(((size_t)qlen + 2) * sizeof(int));
  H[0] = 0; E[0] = NEG;
  int j;
  for (j = 1; j <= qlen && j <= w; ++j) { H[j] = 0; E[j] = NEG; }
  for (; j <= qlen; ++j) H[j] = E[j] = NEG;
  for (int i = 0; i < tlen; ++i) {
    int f = NEG;
    const uint8_t tc = c2u(target[i]);
    const int beg = i, en = i + w + 1 < qlen ? i + w + 1 : qlen;
    int h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG;
    uint8_t *zi = z + (size_t)i * n_col;
    for (j = beg; j < en; ++j) {
      const uint8_t qc = c2u(query[j]);
      const int sc = (tc > 3 || qc > 3) ? 0 : (tc == qc ? 1 : -4);
      int m = H[j], e = E[j];
      H[j] = h1;
      m += sc;
      uint8_t d = m >= e ? 0 : 1;
      int h = m >= e ? m : e;
      d = h >= f ? d : 2;
      h = h >= f ? h : f;
      h1 = h;
      int t = m - oe_del;
      e -= e_del;
      d |= e > t ? 1 << 2 : 0;
      e = e > t ? e : t;
      E[j] = e;
      t = m - oe_ins;
      f -= e_ins;
      d |= f > t ? 2 << 4 : 0;
      f = f > t ? f : t;
      zi[j - beg] = d;
    }
    H[en] = h1; E[en] = NEG;
  }
  int score = H[qlen], maxpos = qlen;
  for (j = 1; j < w; ++j)
    if (H[qlen - j] > score) { score = H[qlen - j]; maxpos = qlen - j; }
  *end = maxpos;
  int n = 0, i = tlen - 1, k = maxpos - 1, which = 0, overflow = 0;
#define PUSHC(op, len) do { if (n == 0 || (int)(cigar[n - 1] & 0xf) != (op)) { if (n < cigar_cap) cigar[n++] = (uint32_t)(len) << 4 | (uint32_t)(op); else overflow = 1; } \
                            else cigar[n - 1] += (uint32_t)(len) << 4; } while (0)
  while (i >= 0 && k >= 0) {
    which = z[(size_t)i * n_col + (k - i)] >> (which << 1) & 3;
    if (which == 0) { PUSHC(0, 1); --i; --k; }
    else if (which == 1) { PUSHC(1, 1); --i; }
    else { PUSHC(2, 1); --k; }
  }
  if (i >= 0) PUSHC(1, i + 1);
#undef PUSHC
  *start = k + 1;
  for (i = 0; i < n >> 1; ++i) { const uint32_t t = cigar[i]; cigar[i] = cigar[n - 1 - i]; cigar[n - 1 - i] = t; }
  *n_cigar_out = n;
  free(z); free(H); free(E);
  return overflow ? -1 : score;
}

static int put_dec(char *dst, int cap, int at, int v) {
  char tmp[16];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) { if (at < cap) dst[at] = tmp[n - 1]; ++at; --n; }
  return at;
}

/* GenerateNMAndMDTag (alignment.cc:85-139): `ref` points at the mapping start */
static int nm_and_md(const char *ref, const char *read, const uint32_t *cigar, int n_cigar, char *md, int md_cap, int *md_len) {
  int nm = 0, matches = 0, rp = 0, fp = 0, at = 0;
  for (int ci = 0; ci < n_cigar; ++ci) {
    const int op = (int)(cigar[ci] & 0xf), len = (int)(cigar[ci] >> 4);
    if (op == 0) {
      for (int t = 0; t < len; ++t) {
        if (ref[fp] == read[rp] || ref[fp] - 'a' + 'A' == read[rp]) ++matches;
        else { ++nm; at = put_dec(md, md_cap, at, matches); matches = 0; if (at < md_cap) md[at] = ref[fp]; ++at; }
        ++fp; ++rp;
      }
    } else if (op == 1) { nm += len; rp += len; }
    else if (op == 2) {
      nm += len;
      at = put_dec(md, md_cap, at, matches); matches = 0;
      if (at < md_cap) md[at] = '^';
      ++at;
      for (int t = 0; t < len; ++t) { if (at < md_cap) md[at] = ref[fp]; ++at; ++fp; }
    }
  }
  at = put_dec(md, md_cap, at, matches);
  *md_len = at;
  return nm;
}

/* GetRefStartEndPositionForReadFromMapping, SAM branches without split alignment
 * (mapping_generator.h:657-717, 723-761, 807-854): for both strands the read (as mapped) is
 * aligned against the window [vw, vw + L + 2e). */
static span_t ref_start_end_sam(const ora_ctx *c, const draft_t *d, const char *read_seq, int L, uint32_t *cigar, int *n_cigar,
                                char *md, int md_cap, int *md_len, int *nm) {
  const int e = c->p.error_threshold;
  const uint32_t rid = (uint32_t)(d->position >> 32), ref_pos = (uint32_t)d->position;
  const uint32_t rl = c->ref->len[rid];
  uint32_t vw = ref_pos + 1 > (uint32_t)(L + e) ? ref_pos + 1 - (uint32_t)L - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)L;
  int st = 0, en = 0;
  ora_ksw_semi_global3(L + 2 * e, c->ref->seq[rid] + vw, L, read_seq, 2 * e + 1, cigar, ORA_SAM_CIGAR_CAP, n_cigar, &st, &en);
  *nm = nm_and_md(c->ref->seq[rid] + vw + st, read_seq, cigar, *n_cigar, md, md_cap, md_len);
  span_t s;
  s.rid = rid;
  s.ref_start = vw + (uint32_t)st;
  s.ref_end = vw + (uint32_t)en - 1;
  return s;
}

/* GetMAPQForSingleEndRead, non-split (mapping_generator.h:920-1022). */
static uint8_t mapq_single(const ora_ctx *c, int num_errors, uint16_t alignment_length, int read_length,
                           int max_diff, const meta_t *m) {
  int mapq_coef_length = 50;
  int mapq_coef_fraction = (int)log((double)mapq_coef_length);
  alignment_length = alignment_length > read_length ? alignment_length : (uint16_t)read_length;
  double alignment_identity = 1 - (double)num_errors / alignment_length;
  int mapq = 0;
  int second = m->second_err;
  if (m->n_best > 1) {
    /* mapq stays 0 */
  } else {
    if (second > num_errors + max_diff) second = num_errors + max_diff;
    double tmp = alignment_length < mapq_coef_length ? 1.0 : mapq_coef_fraction / log((double)alignment_length);
    tmp *= alignment_identity * alignment_identity;
    mapq = (int)(5 * 6.02 * (second - num_errors) * tmp * tmp + 0.499);
  }
  if (m->n_second > 0) mapq -= (int)(4.343 * log((double)(m->n_second + 1)) + 0.499);
  if (mapq > 60) mapq = 60;
  if (mapq < 0) mapq = 0;
  if (m->rep_len > 0) {
    double frac_rep = (m->rep_len) / (double)read_length;
    if (m->rep_len >= (uint32_t)read_length) frac_rep = 0.999;
    if (alignment_identity <= 0.95) mapq = (int)(mapq * (1 - sqrt(frac_rep)) + 0.499);
    else if (alignment_identity <= 0.97) mapq = (int)(mapq * (1 - frac_rep) + 0.499);
    else if (alignment_identity >= 0.999) mapq = (int)(mapq * (1 - frac_rep * frac_rep * frac_rep * frac_rep) + 0.499);
    else mapq = (int)(mapq * (1 - frac_rep * frac_rep) + 0.499);
  }
  (void)c;
  return (uint8_t)mapq;
}

/* GetMAPQForPairedEndRead, non-split (mapping_generator.h:1027-1192). */
static uint8_t mapq_paired(const ora_ctx *c, int err1, int err2, uint16_t al1, uint16_t al2, int len1,
                           int len2, int force_mapq, const pe_meta_t *pe, const meta_t *m1, const meta_t *m2) {
  uint8_t mapq_pe = 0;
  int min_unpaired = m1->min_err + m2->min_err + 3;
  if (pe->n_best <= 1) {
    int adj = pe->second_sum < min_unpaired ? pe->second_sum : min_unpaired;
    mapq_pe = (uint8_t)((int)(5 * 6.02 * (adj - pe->min_sum) / (1) + .499));
    if (pe->n_second > 0) mapq_pe = (uint8_t)(mapq_pe - (int)(4.343 * log((double)(pe->n_second + 1)) + 0.499));
    if (mapq_pe > 60) mapq_pe = 60;
    int rep = (int)(m1->rep_len + m2->rep_len);
    if (rep > 0) {
      double total = len1 + len2;
      double frac_rep = (double)rep / total;
      if (rep >= total) frac_rep = 0.999;
      double id1 = 1 - (double)err1 / (len1 > al1 ? len1 : al1);
      double id2 = 1 - (double)err2 / (len2 > al2 ? len2 : al2);
      double id = id1 < id2 ? id1 : id2;
      if (id <= 0.95) mapq_pe = (uint8_t)(mapq_pe * (1 - sqrt(frac_rep)) + 0.499);
      else if (id <= 0.97) mapq_pe = (uint8_t)(mapq_pe * (1 - frac_rep) + 0.499);
      else if (id >= 0.999) mapq_pe = (uint8_t)(mapq_pe * (1 - frac_rep * frac_rep * frac_rep * frac_rep) + 0.499);
      else mapq_pe = (uint8_t)(mapq_pe * (1 - frac_rep * frac_rep) + 0.499);
    }
  }
  uint8_t mapq1 = mapq_single(c, err1, al1, len1, 2, m1);
  uint8_t mapq2 = mapq_single(c, err2, al2, len2, 2, m2);
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



/* ------------------------------------------------------------------------- */
/* K6: barcode whitelist, abundance, correction                               */
/* ------------------------------------------------------------------------- */
uint64_t ora_seed_from_sequence(const char *seq, uint32_t seq_len, uint32_t start, uint32_t seed_len) {
  uint64_t seed = 0;
  for (uint32_t i = 0; i < seed_len; ++i) {
    if (start + i < seq_len) {
      const uint8_t b = c2u(seq[i + start]);
      seed = b < 4 ? (seed << 2) | b : seed << 2; /* N -> A */
    } else {
      seed <<= 2; /* pad A */
    }
  }
  return seed;
}

struct ora_whitelist {
  uint64_t *keys;
  uint32_t *cnt;
  uint8_t *used;
  uint32_t mask, size, bc_len;
  uint64_t num_sample;
  uint32_t *order; /* insertion order, for export */
};

static uint32_t wl_slot(const ora_whitelist *w, uint64_t key, int *found) {
  uint64_t x = key * 0x9E3779B97F4A7C15ull;
  uint32_t i = (uint32_t)(x >> 32) & w->mask;
  while (w->used[i]) {
    if (w->keys[i] == key) { *found = 1; return i; }
    i = (i + 1) & w->mask;
  }
  *found = 0;
  return i;
}

ora_whitelist *ora_whitelist_load(const char *path, uint32_t barcode_length) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  size_t cap = 1024, n = 0;
  uint64_t *ks = (uint64_t *)malloc(cap * 8);
  char line[300];
  while (fgets(line, sizeof(line), f)) {
    size_t l = strlen(line);
    while (l > 0 && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
    if (l != barcode_length || l > 32) { free(ks); fclose(f); return NULL; } /* chromap.cc:403-415 */
    if (n == cap) { cap *= 2; ks = (uint64_t *)realloc(ks, cap * 8); }
    ks[n++] = ora_seed_from_sequence(line, (uint32_t)l, 0, (uint32_t)l);
  }
  fclose(f);
  ora_whitelist *w = (ora_whitelist *)calloc(1, sizeof(*w));
  uint32_t nb = 16;
  while (nb < 2 * n + 16) nb <<= 1;
  w->mask = nb - 1;
  w->keys = (uint64_t *)calloc(nb, 8);
  w->cnt = (uint32_t *)calloc(nb, 4);
  w->used = (uint8_t *)calloc(nb, 1);
  w->order = (uint32_t *)calloc(n + 1, 4);
  w->bc_len = barcode_length;
  for (size_t i = 0; i < n; ++i) {
    int found;
    uint32_t sl = wl_slot(w, ks[i], &found);
    if (!found) { w->used[sl] = 1; w->keys[sl] = ks[i]; w->order[w->size++] = sl; }
  }
  free(ks);
  return w;
}
void ora_whitelist_free(ora_whitelist *w) {
  if (!w) return;
  free(w->keys); free(w->cnt); free(w->used); free(w->order); free(w);
}
uint32_t ora_whitelist_size(const ora_whitelist *w) { return w->size; }
void ora_whitelist_export(const ora_whitelist *w, uint64_t *keys, uint32_t *counts) {
  for (uint32_t i = 0; i < w->size; ++i) { keys[i] = w->keys[w->order[i]]; counts[i] = w->cnt[w->order[i]]; }
}

long ora_whitelist_abundance(ora_whitelist *w, const char *bc, const uint32_t *bc_off, uint32_t n) {
  const uint64_t max_sample = 20000000ull; /* chromap.h:211 */
  for (uint32_t b0 = 0; b0 < n; b0 += 500000u) {
    const uint32_t bn = n - b0 < 500000u ? n - b0 : 500000u;
    for (uint32_t i = b0; i < b0 + bn; ++i) {
      const char *s = bc + bc_off[i];
      const uint32_t l = bc_off[i + 1] - bc_off[i];
      int has_n = 0;
      for (uint32_t j = 0; j < l; ++j) if (s[j] == 'N') { has_n = 1; break; } /* GetSequenceNsAt tests 'N' only */
      if (has_n) continue;
      int found;
      const uint32_t sl = wl_slot(w, ora_seed_from_sequence(s, l, 0, l), &found);
      if (found) { w->cnt[sl] += 1; ++w->num_sample; }
    }
    if (w->num_sample * 20 < bn) return -1; /* chromap.cc:523-533 */
    if (w->num_sample >= max_sample) break;
  }
  return (long)w->num_sample;
}

typedef struct { uint32_t idx1; char base1; uint32_t idx2; char base2; double score; } bc_cand; /* utils.h:23-35 */
static int bc_cand_greater(const void *a, const void *b) { /* std::greater: descending */
  const bc_cand *x = (const bc_cand *)a, *y = (const bc_cand *)b;
  if (x->score != y->score) return x->score > y->score ? -1 : 1;
  if (x->idx1 != y->idx1) return x->idx1 > y->idx1 ? -1 : 1;
  if (x->base1 != y->base1) return x->base1 > y->base1 ? -1 : 1;
  if (x->idx2 != y->idx2) return x->idx2 > y->idx2 ? -1 : 1;
  if (x->base2 != y->base2) return x->base2 > y->base2 ? -1 : 1;
  return 0;
}

int ora_correct_barcode(const ora_params *p, const ora_whitelist *w, char *bc, const char *qual, uint32_t len,
                        uint64_t *num_in_whitelist, uint64_t *num_corrected) {
  const uint64_t key = ora_seed_from_sequence(bc, len, 0, len);
  int found;
  wl_slot(w, key, &found);
  int n_pos[64], nn = 0;
  for (int i = (int)len - 1; i >= 0; --i) if (bc[i] == 'N' && nn < 64) n_pos[nn++] = (int)len - 1 - i; /* little endian */
  if ((uint32_t)nn > (uint32_t)p->bc_error_threshold) return 0;
  if (nn == 0 && found) { ++*num_in_whitelist; return 1; }
  if (p->bc_error_threshold <= 0) return 0;
  bc_cand *cs = NULL;
  size_t nc = 0, cap = 0;
  const uint64_t mask = 3;
  uint32_t i_start = 0, i_end = len, ti_limit = 3;
  if (nn > 0) { i_start = (uint32_t)n_pos[0]; i_end = i_start + 1; ti_limit = 4; }
#define PUSH(I1, B1, I2, B2, S) do { if (nc == cap) { cap = cap ? cap * 2 : 64; cs = (bc_cand *)realloc(cs, cap * sizeof(bc_cand)); } \
    cs[nc].idx1 = (I1); cs[nc].base1 = (B1); cs[nc].idx2 = (I2); cs[nc].base2 = (B2); cs[nc].score = (S); ++nc; } while (0)
  for (uint32_t i = i_start; i < i_end; ++i) {
    const uint64_t cleared = ~(mask << (2 * i)) & key;
    uint64_t b1 = (key >> (2 * i)) & mask;
    for (uint32_t ti = 0; ti < ti_limit; ++ti) {
      b1 = (b1 + 1) & mask;
      const uint64_t k1 = cleared | (b1 << (2 * i));
      int f1;
      const uint32_t s1 = wl_slot(w, k1, &f1);
      if (f1) {
        const double abundance = w->cnt[s1] / (double)w->num_sample;
        int aq = qual[len - 1 - i] - 33;
        aq = aq > 40 ? 40 : aq;
        aq = aq < 3 ? 3 : aq;
        const double score = pow(10.0, ((-aq) / 10.0)) * abundance;
        PUSH(len - 1 - i, u2c((uint8_t)b1), 0, 0, score);
      }
      if (p->bc_error_threshold == 2) {
        uint32_t j_start = i + 1, j_end = len, ti2_limit = 3;
        if (nn == 2) { j_start = (uint32_t)n_pos[1]; j_end = j_start + 1; ti2_limit = 4; }
        for (uint32_t j = j_start; j < j_end; ++j) {
          const uint64_t cleared2 = ~(mask << (2 * j)) & k1;
          uint64_t b2 = (k1 >> (2 * j)) & mask;
          for (uint32_t ti2 = 0; ti2 < ti2_limit; ++ti2) {
            b2 = (b2 + 1) & mask;
            const uint64_t k2 = cleared2 | (b2 << (2 * j));
            int f2;
            const uint32_t s2 = wl_slot(w, k2, &f2);
            if (f2) {
              const double abundance = w->cnt[s2] / (double)w->num_sample;
              int aq = qual[len - 1 - j] - 33;
              aq = aq > 40 ? 40 : aq;
              aq = aq < 3 ? 3 : aq;
              int aq1 = qual[len - 1 - i] - 33;
              aq1 = aq1 > 40 ? 40 : aq1;
              aq1 = aq1 < 3 ? 3 : aq1;
              aq += aq1;
              const double score = pow(10.0, ((-aq) / 10.0)) * abundance;
              PUSH(len - 1 - i, u2c((uint8_t)b1), len - 1 - j, u2c((uint8_t)b2), score);
            }
          }
        }
      }
    }
  }
#undef PUSH
  int ret = 0;
  if (nc == 0) {
    ret = 0;
  } else {
    size_t best = 0;
    int apply = 1;
    if (nc > 1) {
      qsort(cs, nc, sizeof(bc_cand), bc_cand_greater);
      double sum = 0;
      for (size_t ci = 0; ci < nc; ++ci) sum += cs[ci].score;
      apply = cs[0].score / sum > p->bc_probability_threshold;
    }
    if (apply) {
      bc[cs[best].idx1] = cs[best].base1;
      if (cs[best].base2 != 0) bc[cs[best].idx2] = cs[best].base2;
      ++*num_corrected;
      ret = 1;
    }
  }
  free(cs);
  return ret;
}

/* ------------------------------------------------------------------------- */
/* K0: adapter trimming (chromap.cc:176-289, sequence_batch.h:136-151)         */
/* ------------------------------------------------------------------------- */
/* memmem-like std::string::find(s, pos, n) on a buffer of length hl */
static long find_from(const char *hay, size_t hl, const char *needle, size_t nl, size_t from) {
  if (nl == 0) return from <= hl ? (long)from : -1;
  if (hl < nl) return -1;
  for (size_t i = from; i + nl <= hl; ++i)
    if (hay[i] == needle[0] && memcmp(hay + i, needle, nl) == 0) return (long)i;
  return -1;
}

/* r1/r2: forward reads, n1/n2: their reverse complements; lengths updated in place.
 * The negative strings are trimmed from the FRONT (caller offsets the pointer). */
static int trim_adapter_pe(const ora_ctx *c, const char *r1, uint32_t *len1, const char *n1,
                           const char *r2, uint32_t *len2, const char *n2, uint32_t *nfront1,
                           uint32_t *nfront2) {
  const uint32_t raw1 = *len1, raw2 = *len2;
  const int swap = !(raw1 <= raw2);
  const char *read1 = swap ? r2 : r1;
  const char *neg2 = swap ? n1 : n2;
  const uint32_t l1 = swap ? raw2 : raw1, l2 = swap ? raw1 : raw2;
  const int min_overlap = c->p.min_read_length;
  const int seed = min_overlap / 2;
  const int err_thr = 1;
  for (int si = 0; si < err_thr + 1; ++si) {
    long sp = find_from(neg2, l2, read1 + si * seed, (size_t)seed, 0);
    while (sp >= 0) {
      const int before_ok = (size_t)sp >= (size_t)(si * seed);
      const int overlap_ok = (int)(l2 - (uint32_t)sp + (uint32_t)(seed * si)) >= min_overlap;
      if (!before_ok || !overlap_ok) {
        sp = find_from(neg2, l2, read1 + si * seed, (size_t)seed, (size_t)sp + 1);
        continue;
      }
      int can = 1, ne = 0;
      for (int i = 0; i < seed * si; ++i) {
        if (neg2[sp - si * seed + i] != read1[i]) ++ne;
        if (ne > err_thr) { can = 0; break; }
      }
      for (uint32_t i = (uint32_t)seed; i + (uint32_t)sp < l2 && (uint32_t)(si * seed) + i < l1; ++i) {
        if (neg2[sp + i] != read1[si * seed + i]) ++ne;
        if (ne > err_thr) { can = 0; break; }
      }
      if (can) {
        int overlap = (int)(l2 - (uint32_t)sp + (uint32_t)(si * seed));
        int off2 = 0;
        if (overlap > (int)l1) { off2 = overlap - (int)l1; overlap = (int)l1; }
        int t1 = swap ? overlap + off2 : overlap; /* new length for batch1 read */
        int t2 = swap ? overlap : overlap + off2;
        if (t1 < (int)raw1) { *nfront1 = raw1 - (uint32_t)t1; *len1 = (uint32_t)t1; }
        if (t2 < (int)raw2) { *nfront2 = raw2 - (uint32_t)t2; *len2 = (uint32_t)t2; }
        return 1;
      }
      sp = find_from(neg2, l2, read1 + si * seed, (size_t)seed, (size_t)sp + 1);
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* the pair loop body (chromap.h:892-1143)                                      */
/* ------------------------------------------------------------------------- */
typedef struct {
  meta_t m1, m2;
  pe_meta_t pe;
  char *neg1, *neg2;
  char *fw1, *fw2;
  size_t cap1, cap2;
  int *best_idx;
} work_t;

static void prep_negative(const char *s, uint32_t len, char *out) { /* sequence_batch.h:123-134 */
  for (uint32_t i = 0; i < len; ++i) out[i] = u2c((uint8_t)3 ^ c2u(s[len - i - 1]));
  out[len] = 0;
}

static void sort_drafts(vdraft *v) {
  /* SortMappingsByPositions (mapping_metadata.h:70-78) uses unstable std::sort on position
   * only.  Draft mappings with equal end positions on one strand are possible (two
   * candidates within e of each other); their relative order only matters when both
   * are co-best, which changes which index pair is recorded but not the record emitted
   * unless their error counts differ -- in which case only one of them is best.  A stable
   * order is used here and in the HIP path. */
  if (v->n > 1) {
    /* insertion sort keeps equal keys in emission order */
    for (size_t i = 1; i < v->n; ++i) {
      draft_t x = v->a[i];
      size_t j = i;
      while (j > 0 && v->a[j - 1].position > x.position) { v->a[j] = v->a[j - 1]; --j; }
      v->a[j] = x;
    }
  }
  (void)cmp_draft_pos;
}

/* ---- split alignment (--preset hic): K5 ------------------------------------------------ */
/* GetRefStartEndPositionForReadFromMapping, split + non-SAM branches
 * (mapping_generator.h:657-717, 762-793, 855-916). read_seq: forward read for +, reverse
 * complement for -, both of full length full_len. */
static span_t ref_start_end_split(const ora_ctx *c, const draft_t *d, int split_site_word, int strand,
                                  const char *read_seq, int full_len) {
  const int e = c->p.error_threshold;
  const uint32_t rid = (uint32_t)(d->position >> 32), ref_pos = (uint32_t)d->position;
  const uint32_t rl = c->ref->len[rid];
  const int split_site = split_site_word & 0xffff;
  int gap_beginning = (split_site_word >> 16) & 0xff;
  const int actual = (split_site_word >> 24) & 0xff;
  int read_length = split_site - gap_beginning;
  uint32_t vw = ref_pos + 1 > (uint32_t)(read_length + e) ? ref_pos + 1 - (uint32_t)read_length - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)read_length;
  span_t s;
  s.rid = rid;
  if (strand == 0) {
    int start = 0;
    ora_banded_traceback(e, actual, c->ref->seq[rid] + vw, read_seq + gap_beginning, read_length, &start);
    if (gap_beginning > 0) {
      const int nrs = adjust_gap_beginning(0, c->ref->seq[rid], rl, read_seq, full_len, &gap_beginning, read_length - 1,
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
  ora_banded_align(e, c->ref->seq[rid] + vw, read_seq + read_start_site, read_length, &mep);
  mep += 1;
  if (gap_beginning > 0) {
    const int nre = adjust_gap_beginning(1, c->ref->seq[rid], rl, read_seq + read_start_site, full_len - read_start_site,
                                         &gap_beginning, read_length - 1, (int)vw + start, (int)vw + mep);
    mep = nre - (int)vw + 1;
  }
  s.ref_start = vw + (uint32_t)start;
  s.ref_end = vw + (uint32_t)mep - 1;
  return s;
}

/* GetRefStartEndPositionForReadFromMapping, split + SAM branches (mapping_generator.h:657-761 for +, 806-850 for -):
 * ksw_semi_global3 on the aligned part of the read (the split site pulled in by 3e when the read was cut, :711-717),
 * AdjustGapBeginning extending the first / last M of the CIGAR over the matching bases of the gap (alignment.cc:24-83),
 * NM / MD over the extended alignment.  The window start is the one of the unshortened part, and the - strand hands
 * AdjustGapBeginning reference coordinates without read_start_site -- both as the reference has them. */
static span_t ref_start_end_split_sam(const ora_ctx *c, const draft_t *d, int split_site_word, int strand, const char *read_seq,
                                      int full_len, uint32_t *cigar, int *n_cigar, char *md, int md_cap, int *md_len, int *nm) {
  const int e = c->p.error_threshold;
  const uint32_t rid = (uint32_t)(d->position >> 32), ref_pos = (uint32_t)d->position;
  const uint32_t rl = c->ref->len[rid];
  const char *ref = c->ref->seq[rid];
  int split_site = split_site_word & 0xffff;
  int gap_beginning = (split_site_word >> 16) & 0xff;
  int read_length = split_site - gap_beginning;
  uint32_t vw = ref_pos + 1 > (uint32_t)(read_length + e) ? ref_pos + 1 - (uint32_t)read_length - (uint32_t)e : 0;
  if (ref_pos + (uint32_t)e >= rl) vw = rl - (uint32_t)e - (uint32_t)read_length;
  if (split_site < full_len && split_site > 3 * e) split_site -= 3 * e;
  read_length = split_site - gap_beginning;
  span_t s;
  s.rid = rid;
  int st = 0, en = 0;
  if (strand == 0) {
    ora_ksw_semi_global3(read_length + 2 * e, ref + vw, read_length, read_seq + gap_beginning, 2 * e + 1, cigar, ORA_SAM_CIGAR_CAP, n_cigar, &st, &en);
    if (gap_beginning > 0) {
      const int rs = (int)vw + st;
      const int nrs = adjust_gap_beginning(0, ref, rl, read_seq, full_len, &gap_beginning, read_length - 1, rs, (int)vw + en - 1);
      if (*n_cigar > 0 && (cigar[0] & 0xf) == 0) cigar[0] += (uint32_t)(rs - nrs) << 4;
      st = nrs - (int)vw;
    }
    *nm = nm_and_md(ref + vw + st, read_seq + gap_beginning, cigar, *n_cigar, md, md_cap, md_len);
    s.ref_start = vw + (uint32_t)st;
    s.ref_end = vw + (uint32_t)en - 1;
    return s;
  }
  const int read_start_site = full_len - split_site;
  ora_ksw_semi_global3(read_length + 2 * e, ref + vw + read_start_site, read_length, read_seq + read_start_site, 2 * e + 1, cigar,
                       ORA_SAM_CIGAR_CAP, n_cigar, &st, &en);
  if (gap_beginning > 0) {
    const int re = (int)vw + en - 1;
    const int nre = adjust_gap_beginning(1, ref, rl, read_seq + read_start_site, full_len - read_start_site, &gap_beginning, read_length - 1,
                                         (int)vw + st, re);
    if (*n_cigar > 0 && (cigar[*n_cigar - 1] & 0xf) == 0) cigar[*n_cigar - 1] += (uint32_t)(nre - re) << 4;
    en = nre + 1 - (int)vw - read_start_site;
  }
  *nm = nm_and_md(ref + vw + read_start_site + st, read_seq + read_start_site, cigar, *n_cigar, md, md_cap, md_len);
  s.ref_start = vw + (uint32_t)read_start_site + (uint32_t)st;
  s.ref_end = vw + (uint32_t)read_start_site + (uint32_t)en - 1;
  return s;
}

/* GetMAPQForSingleEndRead with split_alignment (mapping_generator.h:920-1022). strand_ncand:
 * number of candidates on the mapping's strand. */
static uint8_t mapq_single_split(const ora_ctx *c, int num_errors, uint16_t alignment_length, int read_length,
                                 int max_diff, const meta_t *m, uint32_t strand_ncand) {
  const int e = c->p.error_threshold;
  int mapq_coef_length = 50;
  int mapq_coef_fraction = (int)log((double)mapq_coef_length);
  double alignment_identity = (double)(-num_errors) / alignment_length;
  if (alignment_identity > 1) alignment_identity = 1;
  int mapq = 0;
  int second = m->second_err;
  if (m->n_best > 1) {
  } else {
    if (second > num_errors + max_diff) second = num_errors + max_diff;
    double tmp = alignment_length < mapq_coef_length ? 1.0 : mapq_coef_fraction / log((double)alignment_length);
    tmp *= alignment_identity * alignment_identity;
    mapq = (int)(5 * 6.02 * (second - num_errors) * tmp * tmp + 0.499);
  }
  if (m->n_second > 0) mapq -= (int)(4.343 * log((double)(m->n_second + 1)) + 0.499);
  if (mapq > 60) mapq = 60;
  if (mapq < 0) mapq = 0;
  if (m->rep_len > 0) {
    double frac_rep = (m->rep_len) / (double)read_length;
    if (m->rep_len >= (uint32_t)read_length) frac_rep = 0.999;
    if (alignment_identity <= 0.95) mapq = (int)(mapq * (1 - sqrt(frac_rep)) + 0.499);
    else if (alignment_identity <= 0.97) mapq = (int)(mapq * (1 - frac_rep) + 0.499);
    else if (alignment_identity >= 0.999) mapq = (int)(mapq * (1 - frac_rep * frac_rep * frac_rep * frac_rep) + 0.499);
    else mapq = (int)(mapq * (1 - frac_rep * frac_rep) + 0.499);
  }
  if (alignment_length < read_length - e && second != num_errors) { /* :990-1019 */
    if (m->rep_len >= alignment_length && m->rep_len < (uint32_t)read_length && alignment_length < read_length / 3) mapq = 0;
    const int diff = second - num_errors;
    if (second - num_errors <= e * 3 / 4 && strand_ncand >= 5) mapq = (int)((uint32_t)mapq - (strand_ncand / 5 / (uint32_t)diff));
    if (mapq < 0) mapq = 0;
    if (m->n_second > 0 && second - num_errors <= e * 3 / 4) mapq /= (m->n_second / diff + 1);
  }
  return (uint8_t)mapq;
}

/* GetMAPQForPairedEndRead with split_alignment (mapping_generator.h:1027-1192): mapq_pe is
 * computed but only used by the non-split combination (:1163-1170). */
static uint8_t mapq_paired_split(const ora_ctx *c, int s1, int s2, int err1, int err2, uint16_t al1, uint16_t al2,
                                 int len1, int len2, int force_mapq, const meta_t *m1, const meta_t *m2) {
  uint8_t mapq1 = mapq_single_split(c, err1, al1, len1, 2, m1, (uint32_t)(s1 == 0 ? m1->pos_cand.n : m1->neg_cand.n));
  uint8_t mapq2 = mapq_single_split(c, err2, al2, len2, 2, m2, (uint32_t)(s2 == 0 ? m2->pos_cand.n : m2->neg_cand.n));
  mapq1 = (uint8_t)(mapq1 * 1.2);
  if (mapq1 > 60) mapq1 = 60;
  mapq2 = (uint8_t)(mapq2 * 1.2);
  if (mapq2 > 60) mapq2 = 60;
  uint8_t mapq = mapq1 < mapq2 ? mapq1 : mapq2;
  if (mapq < 60 && force_mapq >= 0 && force_mapq < mapq) mapq = (uint8_t)force_mapq;
  return mapq;
}

/* GenerateBestMappingsForPairedEndRead + ProcessBestMappings... + EmplaceBack<PairsMapping>
 * for split alignment (mapping_generator.h:160-253, 389-415, 487-653; mapping_generator.cc:169-210) */
static long finish_pair_split(const ora_ctx *c, work_t *wk, mt19937_t *rng, uint32_t read_id, const char *r1,
                              const char *neg1, uint32_t len1, const char *r2, const char *neg2, uint32_t len2,
                              ora_record *out, ora_trace *tr) {
  const ora_params *p = &c->p;
  meta_t *m1 = &wk->m1, *m2 = &wk->m2;
  pe_meta_t *pe = &wk->pe;
  pe->min_sum = 2 * p->error_threshold + 1; pe->n_best = 0;
  pe->second_sum = 2 * p->error_threshold + 1; pe->n_second = 0;
  static const int S1[4] = {0, 1, 0, 1}, S2[4] = {1, 0, 0, 1}; /* (+,-) (-,+) (+,+) (-,-) */
  for (int o = 0; o < 4; ++o) {
    pe->best[o].n = 0;
    const vdraft *a = S1[o] == 0 ? &m1->pos_map : &m1->neg_map;
    const vdraft *b = S2[o] == 0 ? &m2->pos_map : &m2->neg_map;
    if (a->n == 0 || b->n == 0) continue;
    for (uint32_t i1 = 0; i1 < a->n; ++i1) {
      if (a->a[i1].num_errors != m1->min_err) continue;
      for (uint32_t i2 = 0; i2 < b->n; ++i2) {
        if (b->a[i2].num_errors != m2->min_err) continue;
        vp_push(&pe->best[o], i1, i2);
        pe->min_sum = m1->min_err + m2->min_err;
        pe->n_best++;
      }
    }
  }
  if (tr) { tr->min_sum = pe->min_sum; tr->nbest = pe->n_best; tr->second_sum = pe->second_sum; tr->nsecond = pe->n_second; tr->force_mapq = -1; }
  long nout = 0;
  if (pe->n_best > p->drop_repetitive_reads) return 0;
  /* reservoir sampling over the best pairings (mapping_generator.h:199-214) */
  const int K = p->max_num_best_mappings > 0 ? p->max_num_best_mappings : 1;
  for (int i = 0; i < K; ++i) wk->best_idx[i] = i;
  if (pe->n_best > K) {
    for (int i = K; i < pe->n_best; ++i) {
      int j = mt_uniform(rng, i);
      if (j < K) wk->best_idx[j] = i;
    }
    for (int i = 1; i < K; ++i) { int x = wk->best_idx[i], j = i; while (j > 0 && wk->best_idx[j - 1] > x) { wk->best_idx[j] = wk->best_idx[j - 1]; --j; } wk->best_idx[j] = x; }
  }
  const int to_report = K < pe->n_best ? K : pe->n_best;
  if (pe->n_best < 1) return 0;
  const uint8_t is_unique = (pe->n_best == 1 || m1->n_best == 1 || m2->n_best == 1) ? 1 : 0;
  int idx = 0;
  for (int o = 0; o < 4 && nout < to_report; ++o) { /* mapping_generator.h:222-253 */
    const vdraft *a = S1[o] == 0 ? &m1->pos_map : &m1->neg_map;
    const vdraft *b = S2[o] == 0 ? &m2->pos_map : &m2->neg_map;
    const vec64 *sa = S1[o] == 0 ? &m1->pos_split : &m1->neg_split;
    const vec64 *sb = S2[o] == 0 ? &m2->pos_split : &m2->neg_split;
    for (size_t mi = 0; mi < pe->best[o].n; ++mi) {
      const uint32_t i1 = pe->best[o].a[mi].a, i2 = pe->best[o].a[mi].b;
      if (a->a[i1].num_errors + b->a[i2].num_errors > pe->min_sum) continue;
      if (idx == wk->best_idx[nout]) {
        const int sam = p->output_format == 1 && c->sam_rec != NULL;
        const size_t slot = 2 * (size_t)(read_id - c->sam_first_read_id);
        int ncig1 = 0, ncig2 = 0, mdl1 = 0, mdl2 = 0, nm1 = 0, nm2 = 0;
        span_t x, y;
        if (sam) {
          x = ref_start_end_split_sam(c, &a->a[i1], (int)sa->a[i1], S1[o], S1[o] == 0 ? r1 : neg1, (int)len1,
                                      c->sam_cigar + slot * ORA_SAM_CIGAR_CAP, &ncig1, c->sam_md + slot * c->sam_md_cap, (int)c->sam_md_cap, &mdl1, &nm1);
          y = ref_start_end_split_sam(c, &b->a[i2], (int)sb->a[i2], S2[o], S2[o] == 0 ? r2 : neg2, (int)len2,
                                      c->sam_cigar + (slot + 1) * ORA_SAM_CIGAR_CAP, &ncig2, c->sam_md + (slot + 1) * c->sam_md_cap, (int)c->sam_md_cap,
                                      &mdl2, &nm2);
        } else {
          x = ref_start_end_split(c, &a->a[i1], (int)sa->a[i1], S1[o], S1[o] == 0 ? r1 : neg1, (int)len1);
          y = ref_start_end_split(c, &b->a[i2], (int)sb->a[i2], S2[o], S2[o] == 0 ? r2 : neg2, (int)len2);
        }
        const uint16_t al1 = (uint16_t)(x.ref_end - x.ref_start + 1), al2 = (uint16_t)(y.ref_end - y.ref_start + 1);
        const uint8_t mapq = mapq_paired_split(c, S1[o], S2[o], a->a[i1].num_errors, b->a[i2].num_errors, al1, al2,
                                               (int)len1, (int)len2, -1, m1, m2);
        if (sam) { /* flags mapping_generator.h:613-631, EmplaceBackPairedEndMappingRecord<SAMMapping> (mapping_generator.cc:84-108);
                    * PairedEndMappingInMemory::GetFragmentLength (mapping_in_memory.h:83-90) whatever the chromosomes */
          const int tlen = S1[o] == 0 ? (int)(y.ref_end - x.ref_start + 1) : (int)(x.ref_end - y.ref_start + 1);
          for (int w = 0; w < 2; ++w) {
            ora_sam_record *q = &c->sam_rec[slot + (size_t)w];
            const span_t *me = w == 0 ? &x : &y, *mate = w == 0 ? &y : &x;
            const int my_neg = w == 0 ? S1[o] : S2[o], mate_neg = w == 0 ? S2[o] : S1[o];
            memset(q, 0, sizeof(*q));
            q->read_id = read_id; q->rid = me->rid; q->pos = me->ref_start; q->mpos = mate->ref_start; q->mrid = (int32_t)mate->rid;
            q->tlen = my_neg ? -tlen : tlen;
            q->flag = (uint16_t)(3 | (my_neg ? 16 : 0) | (mate_neg ? 32 : 0) | (w == 0 ? 64 : 128) | (nout >= 1 ? 256 : 0));
            q->mapq = mapq; q->strand = (uint8_t)(my_neg ? 0 : 1); q->is_unique = is_unique; q->valid = 1;
            q->n_cigar = (uint16_t)(w == 0 ? ncig1 : ncig2); q->md_len = (uint16_t)(w == 0 ? mdl1 : mdl2); q->nm = (uint32_t)(w == 0 ? nm1 : nm2);
            c->sam_len[slot + (size_t)w] = w == 0 ? len1 : len2;
          }
        }
        /* EmplaceBackPairedEndMappingRecord<PairsMapping> (mapping_generator.cc:169-210); default
         * rid ranks are the identity (chromap.cc:867-877) */
        uint8_t st1 = S1[o] == 0 ? 1 : 0, st2 = S2[o] == 0 ? 1 : 0;
        int pos1 = (int)(S1[o] == 0 ? x.ref_start : x.ref_end), pos2 = (int)(S2[o] == 0 ? y.ref_start : y.ref_end);
        int rid1 = (int)x.rid, rid2 = (int)y.rid;
        const uint32_t k1 = c->pairs_rank ? c->pairs_rank[rid1] : (uint32_t)rid1, k2 = c->pairs_rank ? c->pairs_rank[rid2] : (uint32_t)rid2;
        const int smaller = k1 < k2 || (rid1 == rid2 && pos1 < pos2);
        if (!smaller) {
          int t = rid1; rid1 = rid2; rid2 = t;
          t = pos1; pos1 = pos2; pos2 = t;
          uint8_t u = st1; st1 = st2; st2 = u;
        }
        ora_pairs_record *r = (ora_pairs_record *)&out[nout++];
        r->read_id = read_id; r->rid1 = (uint32_t)rid1; r->rid2 = (uint32_t)rid2;
        r->pos1 = (uint32_t)pos1; r->pos2 = (uint32_t)pos2;
        r->strand1 = st1; r->strand2 = st2; r->mapq = mapq; r->is_unique = is_unique;
        if (nout == to_report) break;
      }
      ++idx;
    }
  }
  return nout;
}

static long map_one_pair(const ora_ctx *c, work_t *wk, mt19937_t *rng, uint32_t pair_index,
                         uint32_t read_id, const char *s1, uint32_t len1, const char *s2,
                         uint32_t len2, ora_record *out, ora_stats *st, ora_trace *tr) {
  const ora_params *p = &c->p;
  if (tr) memset(tr, 0, sizeof(*tr));
  if (c->wl) { /* chromap.h:896-909: barcode correction comes first */
    uint64_t a = 0, b = 0;
    char *bs = c->bc + c->bco[pair_index];
    const uint32_t bl = c->bco[pair_index + 1] - c->bco[pair_index];
    const int ok = ora_correct_barcode(p, c->wl, bs, c->bcq + c->bco[pair_index], bl, &a, &b);
    ora_ctx *mc = (ora_ctx *)c;
#pragma omp atomic
    mc->n_in_wl += a;
#pragma omp atomic
    mc->n_corr += b;
    c->bc_key[pair_index] = ora_seed_from_sequence(bs, bl, 0, bl);
    if (!ok && !p->output_mappings_not_in_whitelist) return 0;
  }
  if (len1 < (uint32_t)p->min_read_length || len2 < (uint32_t)p->min_read_length) return 0; /* :911-916 */
  if (len1 + 1 > wk->cap1) { wk->cap1 = len1 + 64; wk->neg1 = (char *)realloc(wk->neg1, wk->cap1); wk->fw1 = (char *)realloc(wk->fw1, wk->cap1); }
  if (len2 + 1 > wk->cap2) { wk->cap2 = len2 + 64; wk->neg2 = (char *)realloc(wk->neg2, wk->cap2); wk->fw2 = (char *)realloc(wk->fw2, wk->cap2); }
  memcpy(wk->fw1, s1, len1); wk->fw1[len1] = 0;
  memcpy(wk->fw2, s2, len2); wk->fw2[len2] = 0;
  prep_negative(s1, len1, wk->neg1);
  prep_negative(s2, len2, wk->neg2);
  const char *neg1 = wk->neg1, *neg2 = wk->neg2;
  if (p->trim_adapters) {
    uint32_t f1 = 0, f2 = 0;
    if (trim_adapter_pe(c, wk->fw1, &len1, wk->neg1, wk->fw2, &len2, wk->neg2, &f1, &f2)) {
      if (st) st->num_trimmed++;
      neg1 += f1; neg2 += f2;            /* erase from the front of the revcomp */
      wk->fw1[len1] = 0; wk->fw2[len2] = 0;
    }
  }
  const char *r1 = wk->fw1, *r2 = wk->fw2;
  meta_t *m1 = &wk->m1, *m2 = &wk->m2;
  meta_prepare(m1, len1);
  meta_prepare(m2, len2);
  m1->n_mm = ora_minimizers(r1, len1, pair_index, c->idx->k, c->idx->w, m1->mm_hash, m1->mm_hit);
  m2->n_mm = ora_minimizers(r2, len2, pair_index, c->idx->k, c->idx->w, m2->mm_hash, m2->mm_hit);
  if (st) st->num_minimizers += (uint64_t)(m1->n_mm + m2->n_mm);
  if (tr) { tr->len1 = len1; tr->len2 = len2; tr->n_mm1 = (uint32_t)m1->n_mm; tr->n_mm2 = (uint32_t)m2->n_mm; tr->force_mapq = -1; }
  if (m1->n_mm == 0 || m2->n_mm == 0) return 0; /* :936 */
  gen_candidates(c, m1, st);
  gen_candidates(c, m2, st);
  int supp = 0;
  if (!p->split_alignment) supp = supplement_candidates(c, m1, m2, st); /* :1020-1034 */
  size_t nc1 = m1->pos_cand.n + m1->neg_cand.n, nc2 = m2->pos_cand.n + m2->neg_cand.n;
  if (nc1 > 0 && nc2 > 0 && !p->split_alignment) { /* :1036-1052 */
    vcand t;
    t = m1->pos_cand; m1->pos_cand = m1->pos_buf; m1->pos_buf = t; m1->pos_cand.n = 0;
    t = m1->neg_cand; m1->neg_cand = m1->neg_buf; m1->neg_buf = t; m1->neg_cand.n = 0;
    t = m2->pos_cand; m2->pos_cand = m2->pos_buf; m2->pos_buf = t; m2->pos_cand.n = 0;
    t = m2->neg_cand; m2->neg_cand = m2->neg_buf; m2->neg_buf = t; m2->neg_cand.n = 0;
    reduce_one_direction((uint32_t)p->max_insert_size, &m1->pos_buf, &m2->neg_buf, &m1->pos_cand, &m2->neg_cand);
    reduce_one_direction((uint32_t)p->max_insert_size, &m1->neg_buf, &m2->pos_buf, &m1->neg_cand, &m2->pos_cand);
    nc1 = m1->pos_cand.n + m1->neg_cand.n;
    nc2 = m2->pos_cand.n + m2->neg_cand.n;
  }
  if (tr) { tr->n_cand1 = (uint32_t)nc1; tr->n_cand2 = (uint32_t)nc2; tr->rep1 = m1->rep_len; tr->rep2 = m2->rep_len; }
  if (!(nc1 > 0 && nc2 > 0)) return 0;
  if (st) st->num_candidates += nc1 + nc2;
  rerank_candidates(c, &m1->pos_cand); rerank_candidates(c, &m1->neg_cand); /* chromap.h:1060-1074 */
  rerank_candidates(c, &m2->pos_cand); rerank_candidates(c, &m2->neg_cand);
  gen_draft_mappings(c, m1, r1, neg1, len1, st);
  gen_draft_mappings(c, m2, r2, neg2, len2, st);
  const size_t nd1 = m1->pos_map.n + m1->neg_map.n, nd2 = m2->pos_map.n + m2->neg_map.n;
  if (tr) {
    tr->n_draft1 = (uint32_t)nd1; tr->n_draft2 = (uint32_t)nd2;
    tr->min_err1 = m1->min_err; tr->min_err2 = m2->min_err; tr->nbest1 = m1->n_best; tr->nbest2 = m2->n_best;
    tr->second1 = m1->second_err; tr->second2 = m2->second_err; tr->nsecond1 = m1->n_second; tr->nsecond2 = m2->n_second;
  }
  if (!(nd1 > 0 && nd2 > 0)) return 0; /* :1092-1093 */
  if (!p->split_alignment) {
    sort_drafts(&m1->pos_map); sort_drafts(&m1->neg_map);
    sort_drafts(&m2->pos_map); sort_drafts(&m2->neg_map);
  }
  if (p->split_alignment) {
    const long k = finish_pair_split(c, wk, rng, read_id, r1, neg1, len1, r2, neg2, len2, out, tr);
    if (st) {
      if (wk->pe.n_best == 1) st->num_uniquely_mapped_reads += 2;
      st->num_mappings += 2 * (uint64_t)(wk->pe.n_best < p->max_num_best_mappings ? wk->pe.n_best : p->max_num_best_mappings);
      if (wk->pe.n_best > 0) st->num_mapped_reads += 2;
    }
    return k;
  }
  const int force_mapq = supp != 0 ? 0 : -1;
  /* GenerateBestMappingsForPairedEndRead (mapping_generator.h:160-253) */
  pe_meta_t *pe = &wk->pe;
  pe->min_sum = 2 * p->error_threshold + 1; pe->n_best = 0;
  pe->second_sum = 2 * p->error_threshold + 1; pe->n_second = 0;
  pe->best[0].n = pe->best[1].n = 0;
  best_one_direction(c, 0, m1, m2, len1, len2, pe);
  best_one_direction(c, 1, m1, m2, len1, len2, pe);
  if (tr) { tr->min_sum = pe->min_sum; tr->nbest = pe->n_best; tr->second_sum = pe->second_sum; tr->nsecond = pe->n_second; tr->force_mapq = force_mapq; }
  long nout = 0;
  if (!(pe->n_best > p->drop_repetitive_reads)) {
    const int K = p->max_num_best_mappings;
    for (int i = 0; i < K; ++i) wk->best_idx[i] = i;
    if (pe->n_best > K) {
      for (int i = K; i < pe->n_best; ++i) {
        int j = mt_uniform(rng, i);
        if (j < K) wk->best_idx[j] = i;
      }
      /* std::sort(best_mapping_indices) */
      for (int i = 1; i < K; ++i) { int x = wk->best_idx[i], j = i; while (j > 0 && wk->best_idx[j - 1] > x) { wk->best_idx[j] = wk->best_idx[j - 1]; --j; } wk->best_idx[j] = x; }
    }
    int best_mapping_index = 0, reported = 0;
    const int to_report = K < pe->n_best ? K : pe->n_best;
    const uint8_t is_unique = (pe->n_best == 1 || m1->n_best == 1 || m2->n_best == 1) ? 1 : 0;
    for (int dir = 0; dir < 2 && reported != to_report; ++dir) {
      /* ProcessBestMappingsForPairedEndReadOnOneDirection (mapping_generator.h:487-653) */
      const vdraft *a = dir == 0 ? &m1->pos_map : &m1->neg_map;
      const vdraft *b = dir == 0 ? &m2->neg_map : &m2->pos_map;
      const vpair *best = &pe->best[dir];
      for (size_t mi = 0; mi < best->n; ++mi) {
        const draft_t *d1 = &a->a[best->a[mi].a], *d2 = &b->a[best->a[mi].b];
        if (d1->num_errors + d2->num_errors > pe->min_sum) continue;
        if (best_mapping_index == wk->best_idx[reported]) {
          span_t s1, s2;
          const int sam = p->output_format == 1 && c->sam_rec != NULL;
          const size_t slot = 2 * (size_t)(read_id - c->sam_first_read_id);
          int ncig1 = 0, ncig2 = 0, mdl1 = 0, mdl2 = 0, nm1 = 0, nm2 = 0;
          if (sam) {
            s1 = ref_start_end_sam(c, d1, dir == 0 ? r1 : neg1, (int)len1, c->sam_cigar + slot * ORA_SAM_CIGAR_CAP, &ncig1,
                                   c->sam_md + slot * c->sam_md_cap, (int)c->sam_md_cap, &mdl1, &nm1);
            s2 = ref_start_end_sam(c, d2, dir == 0 ? neg2 : r2, (int)len2, c->sam_cigar + (slot + 1) * ORA_SAM_CIGAR_CAP, &ncig2,
                                   c->sam_md + (slot + 1) * c->sam_md_cap, (int)c->sam_md_cap, &mdl2, &nm2);
          } else {
            s1 = ref_start_end(c, d1, dir == 0 ? 0 : 1, dir == 0 ? r1 : neg1, (int)len1);
            s2 = ref_start_end(c, d2, dir == 0 ? 1 : 0, dir == 0 ? neg2 : r2, (int)len2);
          }
          const uint16_t al1 = (uint16_t)(s1.ref_end - s1.ref_start + 1); /* GetFragmentLength, mapping_in_memory.h:55-57 */
          const uint16_t al2 = (uint16_t)(s2.ref_end - s2.ref_start + 1);
          const uint8_t mapq = mapq_paired(c, d1->num_errors, d2->num_errors, al1, al2, (int)len1, (int)len2,
                                           force_mapq, pe, m1, m2);
          /* EmplaceBackPairedEndMappingRecord (mapping_generator.cc:111-125) with
           * PairedEndMappingInMemory getters (mapping_in_memory.h:64-108) */
          ora_record *r = &out[nout++];
          r->read_id = read_id;
          r->rid = s1.rid;
          const span_t *ps = dir == 0 ? &s1 : &s2; /* the + strand read */
          const span_t *ns = dir == 0 ? &s2 : &s1;
          r->fragment_start = ps->ref_start;
          r->fragment_length = (uint16_t)(int)(ns->ref_end - ps->ref_start + 1);
          r->mapq = mapq & 63; /* mapq_ : 6 (bed_mapping.h:185) */
          r->direction = dir == 0 ? 1 : 0;
          r->is_unique = is_unique;
          r->num_dups = 1;
          r->pos_aln_len = (uint16_t)(ps->ref_end - ps->ref_start + 1);
          r->neg_aln_len = (uint16_t)(ns->ref_end - ns->ref_start + 1);
          if (sam) { /* EmplaceBackPairedEndMappingRecord<SAMMapping> (mapping_generator.cc:84-108), flags :613-631 */
            const int tlen = (int)(ns->ref_end - ps->ref_start + 1); /* PairedEndMappingInMemory::GetFragmentLength, int */
            for (int w = 0; w < 2; ++w) {
              ora_sam_record *q = &c->sam_rec[slot + (size_t)w];
              const span_t *me = w == 0 ? &s1 : &s2, *mate = w == 0 ? &s2 : &s1;
              const int plus = (w == 0) == (dir == 0); /* read 1 is + in direction 0 */
              memset(q, 0, sizeof(*q));
              q->read_id = read_id; q->rid = me->rid; q->pos = me->ref_start; q->mpos = mate->ref_start; q->mrid = (int32_t)mate->rid;
              q->tlen = plus ? tlen : -tlen;
              uint16_t flag = 3;
              if (!plus) flag |= 16; else flag |= 32; /* in a proper F/R pair the mate is on the other strand */
              flag |= w == 0 ? 64 : 128;
              q->flag = flag;
              q->mapq = mapq; q->strand = (uint8_t)plus; q->is_unique = is_unique; q->valid = 1;
              q->n_cigar = (uint16_t)(w == 0 ? ncig1 : ncig2); q->md_len = (uint16_t)(w == 0 ? mdl1 : mdl2); q->nm = (uint32_t)(w == 0 ? nm1 : nm2);
              c->sam_len[slot + (size_t)w] = w == 0 ? len1 : len2;
            }
          }
          ++reported;
          if (reported == (K < pe->n_best ? K : pe->n_best)) break;
        }
        ++best_mapping_index;
      }
    }
  }
  if (st) {
    if (pe->n_best == 1) st->num_uniquely_mapped_reads += 2;
    st->num_mappings += 2 * (uint64_t)(pe->n_best < p->max_num_best_mappings ? pe->n_best : p->max_num_best_mappings);
    if (pe->n_best > 0) st->num_mapped_reads += 2;
  }
  return nout;
}

static void work_init(work_t *wk, const ora_params *p) {
  memset(wk, 0, sizeof(*wk));
  wk->best_idx = (int *)calloc((size_t)(p->max_num_best_mappings > 0 ? p->max_num_best_mappings : 1), sizeof(int));
}
static void work_free(work_t *wk) {
  meta_free(&wk->m1); meta_free(&wk->m2);
  free(wk->pe.best[0].a); free(wk->pe.best[1].a); free(wk->pe.best[2].a); free(wk->pe.best[3].a);
  free(wk->neg1); free(wk->neg2); free(wk->fw1); free(wk->fw2); free(wk->best_idx);
}

/* Reservoir-sampling RNG scope.  In the reference the generator is declared inside the
 * parallel region (chromap.h:863) and is therefore FIRSTPRIVATE in every task the
 * `taskloop grainsize(5000)` (chromap.h:892) generates: each task starts from a fresh
 * copy of mt19937(11) and consumes it over its own iterations only.  libgomp splits a
 * batch of n pairs (n <= 500000 = read_batch_size_, chromap.h:182) into T = n/5000 tasks
 * (1 if T <= 1) of n/T iterations, the first n%T tasks getting one more.  Verified with
 * GCC 11 libgomp at 1..8 threads; the outcome does not depend on the thread count. */
#define ORA_REF_BATCH 500000u
#define ORA_GRAIN 5000u
typedef struct { uint32_t lo, hi; } chunk_t;
static chunk_t *make_chunks(uint32_t n, size_t *nchunks) {
  size_t cap = (size_t)(n / ORA_GRAIN) + (size_t)(n / ORA_REF_BATCH) + 4, k = 0;
  chunk_t *ch = (chunk_t *)malloc(cap * sizeof(chunk_t));
  for (uint32_t b0 = 0; b0 < n; b0 += ORA_REF_BATCH) {
    const uint32_t bn = n - b0 < ORA_REF_BATCH ? n - b0 : ORA_REF_BATCH;
    uint32_t T = bn / ORA_GRAIN;
    if (T <= 1) { ch[k].lo = b0; ch[k].hi = b0 + bn; ++k; continue; }
    const uint32_t d = bn / T, m = bn % T;
    uint32_t s = b0;
    for (uint32_t t = 0; t < T; ++t) {
      const uint32_t sz = d + (t < m ? 1 : 0);
      ch[k].lo = s; ch[k].hi = s + sz; ++k;
      s += sz;
    }
  }
  *nchunks = k;
  return ch;
}

long ora_map_pairs_mt(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1,
                      const uint32_t *r1_off, const char *r2, const uint32_t *r2_off,
                      ora_record *out, ora_stats *stats) {
  if (threads < 1) threads = 1;
  const int K = c->p.max_num_best_mappings > 0 ? c->p.max_num_best_mappings : 1;
  size_t nch = 0;
  chunk_t *ch = make_chunks(n, &nch);
  long *cnt = (long *)calloc(nch + 1, sizeof(long));
  ora_record **bufs = (ora_record **)calloc(nch + 1, sizeof(ora_record *));
  ora_stats *sts = (ora_stats *)calloc((size_t)threads, sizeof(ora_stats));
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    work_t wk;
    work_init(&wk, &c->p);
#pragma omp for schedule(dynamic, 1)
    for (long ci = 0; ci < (long)nch; ++ci) {
      mt19937_t rng;
      mt_seed(&rng, 11);
      const uint32_t lo = ch[ci].lo, hi = ch[ci].hi;
      ora_record *buf = (ora_record *)malloc(((size_t)(hi - lo) * K + 1) * sizeof(ora_record));
      long k = 0;
      for (uint32_t i = lo; i < hi; ++i)
        k += map_one_pair(c, &wk, &rng, i, first_read_id + i, r1 + r1_off[i], r1_off[i + 1] - r1_off[i],
                          r2 + r2_off[i], r2_off[i + 1] - r2_off[i], buf + k, &sts[t],
                          c->trace ? &c->trace[i] : NULL);
      bufs[ci] = buf;
      cnt[ci] = k;
    }
    work_free(&wk);
  }
  long total = 0;
  for (size_t ci = 0; ci < nch; ++ci) {
    if (bufs[ci]) { memcpy(out + total, bufs[ci], (size_t)cnt[ci] * sizeof(ora_record)); free(bufs[ci]); }
    total += cnt[ci];
  }
  for (int t = 0; t < threads; ++t) {
    if (stats) {
      uint64_t *d = (uint64_t *)stats, *s2 = (uint64_t *)&sts[t];
      for (size_t i = 0; i < sizeof(ora_stats) / 8; ++i) d[i] += s2[i];
    }
  }
  free(cnt); free(bufs); free(sts); free(ch);
  return total;
}

long ora_map_pairs(ora_ctx *c, uint32_t n, uint32_t first_read_id, const char *r1,
                   const uint32_t *r1_off, const char *r2, const uint32_t *r2_off,
                   ora_record *out, ora_stats *stats) {
  return ora_map_pairs_mt(c, 1, n, first_read_id, r1, r1_off, r2, r2_off, out, stats);
}

/* ------------------------------------------------------------------------- */
/* post-processing + BED (mapping_writer.h:166-376, mapping_writer.cc:72-83)    */
/* ------------------------------------------------------------------------- */
/* rid, then PairedEndMappingWithoutBarcode::operator< (bed_mapping.h:208-215) */
static int cmp_rec(const void *a, const void *b) {
  const ora_record *x = (const ora_record *)a, *y = (const ora_record *)b;
#define CMPF(f) if (x->f != y->f) return x->f < y->f ? -1 : 1
  CMPF(rid); CMPF(fragment_start); CMPF(fragment_length); CMPF(mapq); CMPF(direction);
  CMPF(is_unique); CMPF(read_id); CMPF(pos_aln_len); CMPF(neg_aln_len);
#undef CMPF
  return 0;
}

/* --TagAlign for paired-end records (mapping_writer.cc:84-117 bulk, :138-168 single-cell): the + read's and the
 * - read's alignment on two lines, the + one first when read 1 is on the + strand; bulk data prints num_dups
 * at the end of the second line, single-cell data prints neither barcode nor num_dups */
static void tagalign_pe_lines(FILE *f, const ora_ref *ref, const ora_record *r, int with_dups) {
  const uint32_t pe = r->fragment_start + r->pos_aln_len, ne = r->fragment_start + r->fragment_length, ns = ne - r->neg_aln_len;
  const char *nm = ref->name[r->rid];
  char tail[16] = "";
  if (with_dups) snprintf(tail, sizeof(tail), "\t%u", (unsigned)r->num_dups);
  if (r->direction)
    fprintf(f, "%s\t%u\t%u\tN\t%u\t+\n%s\t%u\t%u\tN\t%u\t-%s\n", nm, r->fragment_start, pe, (unsigned)r->mapq, nm, ns, ne,
            (unsigned)r->mapq, tail);
  else
    fprintf(f, "%s\t%u\t%u\tN\t%u\t-\n%s\t%u\t%u\tN\t%u\t+%s\n", nm, ns, ne, (unsigned)r->mapq, nm, r->fragment_start, pe,
            (unsigned)r->mapq, tail);
}

static void bed_line(FILE *f, const ora_ref *ref, const ora_params *p, ora_record r, uint32_t dups) {
  r.num_dups = (uint8_t)(dups > 255 ? 255 : dups);
  if (p->tn5_shift) { /* bed_mapping.h:224-229 */
    r.fragment_start += 4;
    r.pos_aln_len -= 4;
    r.fragment_length -= 9;
    r.neg_aln_len -= 5;
  }
  if (p->output_format == 2) { tagalign_pe_lines(f, ref, &r, 1); return; }
  fprintf(f, "%s\t%u\t%u\tN\t%u\t%s\t%u\n", ref->name[r.rid], r.fragment_start,
          r.fragment_start + r.fragment_length, (unsigned)r.mapq, r.direction ? "+" : "-", (unsigned)r.num_dups);
}

/* Two post-processing flavours (chromap.h:1305-1355):
 *  low_mem: sort, then the temp-file merge (mapping_writer.h:247-289) keeps the FIRST record with
 *           the maximal MAPQ of every operator== run; Tn5 shift at output.
 *  in-memory: ApplyTn5ShiftOnMappings first, then RemovePCRDuplicate (mapping_processor.h:160-202)
 *           sorts and keeps the LAST record of every run. */
long ora_write_bed_pe(const ora_ref *ref, const ora_params *p, ora_record *rec, long n, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  const int inmem = !p->low_mem;
  ora_params q = *p;
  if (inmem && p->tn5_shift) {
    for (long i = 0; i < n; ++i) { rec[i].fragment_start += 4; rec[i].pos_aln_len -= 4; rec[i].fragment_length -= 9; rec[i].neg_aln_len -= 5; }
    q.tn5_shift = 0;
  }
  qsort(rec, (size_t)n, sizeof(ora_record), cmp_rec);
  long lines = 0, i = 0;
  while (i < n) {
    ora_record last = rec[i];
    uint32_t dups = 1;
    long j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && rec[j].rid == last.rid && rec[j].fragment_start == rec[i].fragment_start &&
             rec[j].fragment_length == rec[i].fragment_length) {
        ++dups;
        if (inmem || rec[j].mapq > last.mapq) last = rec[j]; /* :268-270; keys used by == are equal */
        ++j;
      }
    }
    if (last.mapq >= p->mapq_threshold) { bed_line(f, ref, &q, last, dups); ++lines; }
    i = j;
  }
  fclose(f);
  return lines;
}

/* pairs output (--preset hic): sort by PairsMapping::operator< per rid1 (pairs_mapping.h:40-43;
 * rid1 is also the per-chromosome vector index, mapping_generator.cc:205), MAPQ filter, no
 * dedup unless remove_pcr_duplicates (== on rid1,pos1,rid2,pos2), header + lines
 * (mapping_writer.cc:381-420).  read_names[i] = name of read1 of pair i. */
static int cmp_pairs_rec(const void *a, const void *b) {
  const ora_pairs_record *x = (const ora_pairs_record *)a, *y = (const ora_pairs_record *)b;
#define CMPF(f) if (x->f != y->f) return x->f < y->f ? -1 : 1
  CMPF(rid1); CMPF(rid2); CMPF(pos1); CMPF(pos2); CMPF(mapq); CMPF(read_id);
#undef CMPF
  return 0;
}

static const uint32_t *g_pairs_header_rank; /* set by ora_write_pairs_ranked for one call */
long ora_write_pairs_ranked(const ora_ref *ref, const ora_params *p, ora_pairs_record *rec, long n, const char *const *read_names,
                            const uint32_t *pairs_rank, const char *out_path) {
  g_pairs_header_rank = pairs_rank;
  const long k = ora_write_pairs(ref, p, rec, n, read_names, out_path);
  g_pairs_header_rank = NULL;
  return k;
}
long ora_write_pairs(const ora_ref *ref, const ora_params *p, ora_pairs_record *rec, long n,
                     const char *const *read_names, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  qsort(rec, (size_t)n, sizeof(ora_pairs_record), cmp_pairs_rec);
  fprintf(f, "## pairs format v1.0.0\n#shape: upper triangle\n");
  for (uint32_t i = 0; i < ref->n_seq; ++i) { /* header in pairs-rank order (mapping_writer.cc:385-399) */
    uint32_t rid = i;
    if (g_pairs_header_rank) for (uint32_t j = 0; j < ref->n_seq; ++j) if (g_pairs_header_rank[j] == i) rid = j;
    fprintf(f, "#chromsize: %s %u\n", ref->name[rid], ref->len[rid]);
  }
  fprintf(f, "#columns: readID chrom1 pos1 chrom2 pos2 strand1 strand2 pair_type mapq1 mapq2\n");
  long lines = 0;
  for (long i = 0; i < n; ++i) {
    const ora_pairs_record *r = &rec[i];
    if (r->mapq < p->mapq_threshold) continue;
    fprintf(f, "%s\t%s\t%d\t%s\t%d\t%c\t%c\tUU\t%u\t%u\n", read_names[r->read_id], ref->name[r->rid1], (int)r->pos1 + 1,
            ref->name[r->rid2], (int)r->pos2 + 1, r->strand1 ? '+' : '-', r->strand2 ? '+' : '-', (unsigned)r->mapq,
            (unsigned)r->mapq);
    ++lines;
  }
  fclose(f);
  return lines;
}

long ora_map_pairs_bc(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1,
                      const uint32_t *r1_off, const char *r2, const uint32_t *r2_off, char *bc,
                      const char *bc_qual, const uint32_t *bc_off, const ora_whitelist *w,
                      ora_record_bc *out, ora_stats *stats, uint64_t *num_in_whitelist, uint64_t *num_corrected) {
  ora_record *tmp = (ora_record *)malloc(((size_t)n * (size_t)(c->p.max_num_best_mappings > 0 ? c->p.max_num_best_mappings : 1) + 1) * sizeof(ora_record));
  c->wl = w; c->bc = bc; c->bcq = bc_qual; c->bco = bc_off;
  c->bc_key = (uint64_t *)calloc((size_t)n + 1, 8);
  c->n_in_wl = c->n_corr = 0;
  const long k = ora_map_pairs_mt(c, threads, n, first_read_id, r1, r1_off, r2, r2_off, tmp, stats);
  for (long i = 0; i < k; ++i) {
    out[i].r = tmp[i];
    out[i].barcode = c->bc_key[tmp[i].read_id - first_read_id];
  }
  if (num_in_whitelist) *num_in_whitelist += c->n_in_wl;
  if (num_corrected) *num_corrected += c->n_corr;
  free(c->bc_key); free(tmp);
  c->wl = NULL; c->bc = NULL; c->bcq = NULL; c->bco = NULL; c->bc_key = NULL;
  return k;
}

/* PairedEndMappingWithBarcode::operator< (bed_mapping.h:145-153) within rid */
static int cmp_rec_bc(const void *a, const void *b) {
  const ora_record_bc *x = (const ora_record_bc *)a, *y = (const ora_record_bc *)b;
#define CMPF(f) if (x->f != y->f) return x->f < y->f ? -1 : 1
  CMPF(r.rid); CMPF(r.fragment_start); CMPF(r.fragment_length); CMPF(barcode); CMPF(r.mapq); CMPF(r.direction);
  CMPF(r.is_unique); CMPF(r.read_id); CMPF(r.pos_aln_len); CMPF(r.neg_aln_len);
#undef CMPF
  return 0;
}

/* Low-memory merge with cell-level duplicate removal (remove_pcr_duplicates_at_bulk_level ==
 * false, as --preset atac sets it): operator== is (barcode, start, length)
 * (bed_mapping.h:154-159); writer mapping_writer.cc:119-131 with Seed2Sequence
 * (barcode_translator.h:107-116). */
long ora_write_bed_pe_bc(const ora_ref *ref, const ora_params *p, ora_record_bc *rec, long n, uint32_t barcode_length,
                         const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  const int inmem = !p->low_mem;
  if (inmem && p->tn5_shift)
    for (long t = 0; t < n; ++t) { rec[t].r.fragment_start += 4; rec[t].r.pos_aln_len -= 4; rec[t].r.fragment_length -= 9; rec[t].r.neg_aln_len -= 5; }
  qsort(rec, (size_t)n, sizeof(ora_record_bc), cmp_rec_bc);
  long lines = 0, i = 0;
  char bcs[40];
  while (i < n) {
    ora_record_bc last = rec[i];
    uint32_t dups = 1;
    long j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && rec[j].r.rid == last.r.rid && rec[j].barcode == last.barcode &&
             rec[j].r.fragment_start == last.r.fragment_start && rec[j].r.fragment_length == last.r.fragment_length) {
        ++dups;
        if (inmem || rec[j].r.mapq > last.r.mapq) last = rec[j];
        ++j;
      }
    }
    if (last.r.mapq >= p->mapq_threshold) {
      ora_record r = last.r;
      r.num_dups = (uint8_t)(dups > 255 ? 255 : dups);
      if (p->tn5_shift && !inmem) { r.fragment_start += 4; r.pos_aln_len -= 4; r.fragment_length -= 9; r.neg_aln_len -= 5; }
      for (uint32_t b = 0; b < barcode_length; ++b) bcs[b] = u2c((uint8_t)((last.barcode >> ((barcode_length - 1 - b) * 2)) & 3));
      bcs[barcode_length] = 0;
      if (p->output_format == 2) tagalign_pe_lines(f, ref, &r, 0);
      else
      fprintf(f, "%s\t%u\t%u\t%s\t%u\n", ref->name[r.rid], r.fragment_start, r.fragment_start + r.fragment_length, bcs,
              (unsigned)r.num_dups);
      ++lines;
    }
    i = j;
  }
  fclose(f);
  return lines;
}

/* Single-cell BED with duplicate removal at BULK level (the default for barcoded data without
 * --preset atac; low-memory merge only, mapping_writer.h:202-345): a run is every record with the
 * same (rid, start, length) regardless of barcode.  Inside a run the records of one barcode are
 * consecutive; each barcode group is represented by its last record with num_dups_ = 1 for a
 * single record and 2 for any larger group (the merge assigns the incoming record, whose num_dups_
 * is 1, and then adds one: :257-263).  FindBestMappingIndexFromDuplicates (:124-163) takes the
 * group with the larger num_dups_, then the larger whitelist abundance, first one on ties.  The
 * line carries num_dups = min(255, run size).  The MAPQ filter looks at the chosen record -- except
 * for the very last run of the output, where it looks at the first record with the maximal MAPQ of
 * the run (:331-337 test before the replacement). */
long ora_write_bed_pe_bc_bulk(const ora_ref *ref, const ora_params *p, ora_record_bc *rec, long n, uint32_t barcode_length,
                              const ora_whitelist *w, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  qsort(rec, (size_t)n, sizeof(ora_record_bc), cmp_rec_bc);
  long lines = 0, i = 0;
  char bcs[40];
  while (i < n) {
    long j = i + 1;
    while (j < n && rec[j].r.rid == rec[i].r.rid && rec[j].r.fragment_start == rec[i].r.fragment_start &&
           rec[j].r.fragment_length == rec[i].r.fragment_length) ++j;
    long best = -1, maxq = i;
    uint32_t best_nd = 0, best_ab = 0;
    for (long g = i; g < j;) {
      long h = g + 1;
      while (h < j && rec[h].barcode == rec[g].barcode) ++h;
      const uint32_t nd = h - g >= 2 ? 2 : 1;
      int found = 0;
      const uint32_t slot = wl_slot(w, rec[g].barcode, &found);
      const uint32_t ab = found ? w->cnt[slot] : 0;
      if (best < 0 || nd > best_nd || (nd == best_nd && ab > best_ab)) { best = h - 1; best_nd = nd; best_ab = ab; }
      g = h;
    }
    for (long t = i + 1; t < j; ++t) if (rec[t].r.mapq > rec[maxq].r.mapq) maxq = t;
    const uint8_t filter_mapq = j == n ? rec[maxq].r.mapq : rec[best].r.mapq;
    if (filter_mapq >= p->mapq_threshold) {
      ora_record r = rec[best].r;
      const uint32_t dups = (uint32_t)(j - i);
      if (p->tn5_shift) { r.fragment_start += 4; r.fragment_length -= 9; }
      for (uint32_t b = 0; b < barcode_length; ++b) bcs[b] = u2c((uint8_t)((rec[best].barcode >> ((barcode_length - 1 - b) * 2)) & 3));
      bcs[barcode_length] = 0;
      fprintf(f, "%s\t%u\t%u\t%s\t%u\n", ref->name[r.rid], r.fragment_start, r.fragment_start + r.fragment_length, bcs, dups > 255 ? 255u : dups);
      ++lines;
    }
    i = j;
  }
  fclose(f);
  return lines;
}

typedef struct { vchar b, q; uint32_t *off; size_t n, cap; } fqq_acc;
long ora_read_fastq_qual(const char *path, char **bases, char **quals, uint32_t **off) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  fqq_acc a;
  memset(&a, 0, sizeof(a));
  char *line = NULL;
  size_t lcap = 0;
  ssize_t ll;
  int state = 0;
  size_t cur_len = 0;
  a.cap = 1024;
  a.off = (uint32_t *)malloc(a.cap * 4);
  a.off[0] = 0;
  while ((ll = getline(&line, &lcap, f)) >= 0) {
    while (ll > 0 && (line[ll - 1] == '\n' || line[ll - 1] == '\r')) line[--ll] = 0;
    if (state == 0) { if (line[0] == '@') state = 1; }
    else if (state == 1) { vch_append(&a.b, line, (size_t)ll); cur_len = (size_t)ll; state = 2; }
    else if (state == 2) { state = 3; }
    else {
      vch_append(&a.q, line, (size_t)ll);
      if ((size_t)ll != cur_len) { free(line); fclose(f); return -2; }
      if (a.n + 2 > a.cap) { a.cap *= 2; a.off = (uint32_t *)realloc(a.off, a.cap * 4); }
      a.off[++a.n] = (uint32_t)a.b.n;
      state = 0;
    }
  }
  free(line);
  fclose(f);
  if (!a.b.a) { a.b.a = (char *)calloc(1, 1); a.q.a = (char *)calloc(1, 1); }
  *bases = a.b.a; *quals = a.q.a; *off = a.off;
  return (long)a.n;
}

/* ------------------------------------------------------------------------- */
/* single-end: taskloop body chromap.h:385-472, GenerateBestMappingsForSingleEndRead           */
/* (mapping_generator.h:115-157, 256-344), bulk records MappingWithoutBarcode                  */
/* ------------------------------------------------------------------------- */
static long map_one_read(const ora_ctx *c, work_t *wk, uint32_t read_index, uint32_t read_id, const char *s1,
                         uint32_t len1, ora_record *out, ora_stats *st) {
  const ora_params *p = &c->p;
  if (c->wl) { /* chromap.h:389-401: barcode correction comes first */
    uint64_t a = 0, b = 0;
    char *bs = c->bc + c->bco[read_index];
    const uint32_t bl = c->bco[read_index + 1] - c->bco[read_index];
    const int ok = ora_correct_barcode(p, c->wl, bs, c->bcq + c->bco[read_index], bl, &a, &b);
    ora_ctx *mc = (ora_ctx *)c;
#pragma omp atomic
    mc->n_in_wl += a;
#pragma omp atomic
    mc->n_corr += b;
    c->bc_key[read_index] = ora_seed_from_sequence(bs, bl, 0, bl);
    if (!ok && !p->output_mappings_not_in_whitelist) return 0;
  }
  if (len1 < (uint32_t)p->min_read_length) return 0;
  if (len1 + 1 > wk->cap1) { wk->cap1 = len1 + 64; wk->neg1 = (char *)realloc(wk->neg1, wk->cap1); wk->fw1 = (char *)realloc(wk->fw1, wk->cap1); }
  memcpy(wk->fw1, s1, len1); wk->fw1[len1] = 0;
  prep_negative(s1, len1, wk->neg1);
  meta_t *m = &wk->m1;
  meta_prepare(m, len1);
  m->n_mm = ora_minimizers(wk->fw1, len1, read_index, c->idx->k, c->idx->w, m->mm_hash, m->mm_hit);
  if (st) st->num_minimizers += (uint64_t)m->n_mm;
  if (m->n_mm == 0) return 0;
  gen_candidates(c, m, st);
  const size_t nc = m->pos_cand.n + m->neg_cand.n;
  if (nc == 0) return 0;
  if (st) st->num_candidates += nc;
  rerank_candidates(c, &m->pos_cand); rerank_candidates(c, &m->neg_cand); /* chromap.h:416-420 */
  gen_draft_mappings(c, m, wk->fw1, wk->neg1, len1, st);
  if (m->pos_map.n + m->neg_map.n == 0) return 0;
  /* a fresh std::mt19937(11) per read, reservoir sampling when there are more best mappings than
   * max_num_best_mappings (mapping_generator.h:121-139); the chosen indices are reported in increasing order */
  const int K = p->max_num_best_mappings > 0 ? p->max_num_best_mappings : 1;
  int *choices = wk->best_idx;
  for (int i = 0; i < K; ++i) choices[i] = i;
  if (m->n_best > K) {
    mt19937_t g;
    mt_seed(&g, 11);
    for (int i = K; i < m->n_best; ++i) { int j = mt_uniform(&g, i); if (j < K) choices[j] = i; }
    for (int a = 1; a < K; ++a) { /* std::sort */
      const int v = choices[a];
      int b = a - 1;
      while (b >= 0 && choices[b] > v) { choices[b + 1] = choices[b]; --b; }
      choices[b + 1] = v;
    }
  }
  const int to_report = m->n_best < K ? m->n_best : K;
  long nout = 0;
  int idx = 0;
  for (int strand = 0; strand < 2 && nout < to_report; ++strand) {
    const vdraft *v = strand == 0 ? &m->pos_map : &m->neg_map;
    for (size_t mi = 0; mi < v->n && nout < to_report; ++mi) {
      if (v->a[mi].num_errors > m->min_err) continue;
      if (idx == choices[nout]) {
        const int sam = p->output_format == 1 && c->sam_rec != NULL;
        const size_t slot = (size_t)(read_id - c->sam_first_read_id);
        int ncig = 0, mdl = 0, nm = 0;
        const span_t sp = sam ? ref_start_end_sam(c, &v->a[mi], strand == 0 ? wk->fw1 : wk->neg1, (int)len1,
                                                  c->sam_cigar + slot * ORA_SAM_CIGAR_CAP, &ncig, c->sam_md + slot * c->sam_md_cap,
                                                  (int)c->sam_md_cap, &mdl, &nm)
                                : ref_start_end(c, &v->a[mi], strand, strand == 0 ? wk->fw1 : wk->neg1, (int)len1);
        const uint16_t al = (uint16_t)(sp.ref_end - sp.ref_start + 1);
        const uint8_t mapq = mapq_single(c, v->a[mi].num_errors, al, (int)len1, p->error_threshold, m);
        ora_record *r = &out[nout++];
        memset(r, 0, sizeof(*r));
        r->read_id = read_id; r->rid = sp.rid; r->fragment_start = sp.ref_start; r->fragment_length = al;
        r->mapq = mapq & 63; r->direction = strand == 0 ? 1 : 0; r->is_unique = m->n_best == 1; r->num_dups = 1;
        if (sam) { /* EmplaceBackSingleEndMappingRecord<SAMMapping> (mapping_generator.cc:43-57), flag :321-326 */
          ora_sam_record *q = &c->sam_rec[slot];
          memset(q, 0, sizeof(*q));
          q->read_id = read_id; q->rid = sp.rid; q->pos = sp.ref_start; q->mpos = 0; q->mrid = -1; q->tlen = 0;
          q->flag = strand == 0 ? 0 : 16; q->mapq = mapq; q->strand = strand == 0 ? 1 : 0; q->is_unique = m->n_best == 1; q->valid = 1;
          q->n_cigar = (uint16_t)ncig; q->md_len = (uint16_t)mdl; q->nm = (uint32_t)nm;
          c->sam_len[slot] = len1;
        }
      }
      ++idx;
    }
  }
  if (st) {
    st->num_mappings += (uint64_t)(m->n_best < p->max_num_best_mappings ? m->n_best : p->max_num_best_mappings);
    st->num_mapped_reads += 1;
    if (m->n_best == 1) st->num_uniquely_mapped_reads += 1;
  }
  return nout;
}

long ora_map_single(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r, const uint32_t *r_off,
                    ora_record *out, ora_stats *stats) {
  if (threads < 1) threads = 1;
  uint8_t *has = (uint8_t *)calloc((size_t)n + 1, 1);
  const size_t K = c->p.max_num_best_mappings > 0 ? (size_t)c->p.max_num_best_mappings : 1; /* up to K records per read; K < 256 */
  ora_record *tmp = (ora_record *)malloc(((size_t)n * K + 1) * sizeof(ora_record));
  ora_stats *sts = (ora_stats *)calloc((size_t)threads, sizeof(ora_stats));
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    work_t wk;
    work_init(&wk, &c->p);
#pragma omp for schedule(dynamic, 1024)
    for (long i = 0; i < (long)n; ++i)
      has[i] = (uint8_t)map_one_read(c, &wk, (uint32_t)i, first_read_id + (uint32_t)i, r + r_off[i], r_off[i + 1] - r_off[i], &tmp[(size_t)i * K], &sts[t]);
    work_free(&wk);
  }
  long k = 0;
  for (uint32_t i = 0; i < n; ++i)
    for (size_t j = 0; j < has[i]; ++j) out[k++] = tmp[(size_t)i * K + j];
  for (int t = 0; t < threads && stats; ++t) {
    uint64_t *d = (uint64_t *)stats, *s2 = (uint64_t *)&sts[t];
    for (size_t i = 0; i < sizeof(ora_stats) / 8; ++i) d[i] += s2[i];
  }
  free(has); free(tmp); free(sts);
  return k;
}

/* MappingWithoutBarcode: sort (bed_mapping.h:90-95), dedup on fragment_start only (:96-99),
 * Tn5 shift (:104-110), line mapping_writer.cc:44-52 */
static int cmp_rec_se(const void *a, const void *b) {
  const ora_record *x = (const ora_record *)a, *y = (const ora_record *)b;
#define CMPF(f) if (x->f != y->f) return x->f < y->f ? -1 : 1
  CMPF(rid); CMPF(fragment_start); CMPF(fragment_length); CMPF(mapq); CMPF(direction); CMPF(is_unique); CMPF(read_id);
#undef CMPF
  return 0;
}
long ora_write_bed_se(const ora_ref *ref, const ora_params *p, ora_record *rec, long n, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  const int inmem = !p->low_mem;
  if (inmem && p->tn5_shift) /* bed_mapping.h:100-106 */
    for (long t = 0; t < n; ++t) { if (rec[t].direction == 1) rec[t].fragment_start += 4; else rec[t].fragment_length -= 5; }
  qsort(rec, (size_t)n, sizeof(ora_record), cmp_rec_se);
  long lines = 0, i = 0;
  while (i < n) {
    ora_record last = rec[i];
    uint32_t dups = 1;
    long j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && rec[j].rid == last.rid && rec[j].fragment_start == rec[i].fragment_start) {
        ++dups;
        if (inmem || rec[j].mapq > last.mapq) last = rec[j];
        ++j;
      }
    }
    if (last.mapq >= p->mapq_threshold) {
      if (p->tn5_shift && !inmem) { if (last.direction == 1) last.fragment_start += 4; else last.fragment_length -= 5; }
      fprintf(f, "%s\t%u\t%u\tN\t%u\t%s\t%u\n", ref->name[last.rid], last.fragment_start,
              last.fragment_start + last.fragment_length, (unsigned)last.mapq, last.direction ? "+" : "-",
              dups > 255 ? 255u : dups);
      ++lines;
    }
    i = j;
  }
  fclose(f);
  return lines;
}

/* ------------------------------------------------------------------------- */
/* --SAM                                                                        */
/* ------------------------------------------------------------------------- */
long ora_map_pairs_sam(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1, const uint32_t *r1_off,
                       const char *r2, const uint32_t *r2_off, ora_sam_record *out, uint32_t *cigar_pool, char *md_pool,
                       uint32_t md_cap, ora_stats *stats) {
  ora_record *tmp = (ora_record *)malloc(((size_t)n + 1) * sizeof(ora_record));
  uint32_t *lens = (uint32_t *)calloc(2 * (size_t)n + 1, 4);
  memset(out, 0, 2 * (size_t)n * sizeof(ora_sam_record));
  c->sam_rec = out; c->sam_cigar = cigar_pool; c->sam_md = md_pool; c->sam_md_cap = md_cap; c->sam_first_read_id = first_read_id;
  c->sam_len = lens;
  const int fmt = c->p.output_format;
  c->p.output_format = 1;
  ora_map_pairs_mt(c, threads, n, first_read_id, r1, r1_off, r2, r2_off, tmp, stats);
  c->p.output_format = fmt;
  long k = 0;
  for (size_t i = 0; i < 2 * (size_t)n; ++i) { k += out[i].valid; out[i].reserved = (uint16_t)lens[i]; }
  c->sam_rec = NULL; c->sam_cigar = NULL; c->sam_md = NULL; c->sam_len = NULL;
  free(tmp); free(lens);
  return k;
}

long ora_map_single_sam(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r, const uint32_t *r_off,
                        ora_sam_record *out, uint32_t *cigar_pool, char *md_pool, uint32_t md_cap, ora_stats *stats) {
  ora_record *tmp = (ora_record *)malloc(((size_t)n + 1) * sizeof(ora_record));
  uint32_t *lens = (uint32_t *)calloc((size_t)n + 1, 4);
  memset(out, 0, (size_t)n * sizeof(ora_sam_record));
  c->sam_rec = out; c->sam_cigar = cigar_pool; c->sam_md = md_pool; c->sam_md_cap = md_cap; c->sam_first_read_id = first_read_id;
  c->sam_len = lens;
  const int fmt = c->p.output_format;
  c->p.output_format = 1;
  ora_map_single(c, threads, n, first_read_id, r, r_off, tmp, stats);
  c->p.output_format = fmt;
  long k = 0;
  for (size_t i = 0; i < (size_t)n; ++i) { k += out[i].valid; out[i].reserved = (uint16_t)lens[i]; }
  c->sam_rec = NULL; c->sam_cigar = NULL; c->sam_md = NULL; c->sam_len = NULL;
  free(tmp); free(lens);
  return k;
}

typedef struct { const ora_sam_record *r; long slot; uint64_t bc; } sam_ref_t;
/* SAMMapping::operator< under the per-chromosome vectors (sam_mapping.h:201-206); barcode 0 for bulk data */
static int cmp_sam(const void *a, const void *b) {
  const ora_sam_record *x = ((const sam_ref_t *)a)->r, *y = ((const sam_ref_t *)b)->r;
  const uint64_t xb = ((const sam_ref_t *)a)->bc, yb = ((const sam_ref_t *)b)->bc;
#define CMPV(u, v) if ((u) != (v)) return (u) < (v) ? -1 : 1
  CMPV(x->rid, y->rid); CMPV(x->pos, y->pos); CMPV(xb, yb); CMPV(x->mrid, y->mrid); CMPV(x->mpos, y->mpos);
  CMPV(x->flag & 64, y->flag & 64); CMPV(x->mapq, y->mapq); CMPV(x->read_id, y->read_id);
#undef CMPV
  return 0;
}
static int sam_same(const ora_sam_record *x, const ora_sam_record *y) { /* operator== (sam_mapping.h:200-205) */
  return x->pos == y->pos && x->rid == y->rid && (x->flag & 64) == (y->flag & 64) && x->mrid == y->mrid && x->mpos == y->mpos;
}

static long write_sam_impl(const ora_ref *ref, const ora_params *p, const ora_sam_record *rec, long n_slots, int paired,
                           const uint32_t *cigar_pool, const char *md_pool, uint32_t md_cap, const char *const *names1,
                           const char *const *names2, const char *b1, const char *q1, const uint32_t *o1, const char *b2,
                           const char *q2, const uint32_t *o2, const uint32_t *len_after_trim, const uint64_t *bck, uint32_t bc_len,
                           const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  for (uint32_t i = 0; i < ref->n_seq; ++i) fprintf(f, "@SQ\tSN:%s\tLN:%u\n", ref->name[i], ref->len[i]); /* mapping_writer.cc:312-321 */
  sam_ref_t *v = (sam_ref_t *)malloc(((size_t)n_slots + 1) * sizeof(sam_ref_t));
  long n = 0;
  for (long i = 0; i < n_slots; ++i) if (rec[i].valid) { v[n].r = &rec[i]; v[n].slot = i; v[n].bc = bck ? bck[paired ? i / 2 : i] : 0; ++n; }
  qsort(v, (size_t)n, sizeof(sam_ref_t), cmp_sam);
  const int inmem = !p->low_mem;
  long lines = 0, i = 0;
  char *seq = NULL, *qual = NULL;
  size_t cap = 0;
  while (i < n) {
    sam_ref_t last = v[i];
    long j = i + 1;
    if (p->remove_pcr_duplicates) {
      while (j < n && sam_same(v[j].r, v[i].r) && v[j].bc == v[i].bc) {
        if (inmem || v[j].r->mapq > last.r->mapq) last = v[j];
        ++j;
      }
    }
    const ora_sam_record *r = last.r;
    if (r->mapq >= p->mapq_threshold) {
      const long slot = last.slot;
      const long item = paired ? slot / 2 : slot;
      const int mate2 = paired && (slot & 1);
      const char *name = mate2 ? names2[item] : names1[item];
      const char *bs = (mate2 ? b2 : b1) + (mate2 ? o2 : o1)[item];
      const char *qs = (mate2 ? q2 : q1) + (mate2 ? o2 : o1)[item];
      uint32_t L = len_after_trim ? len_after_trim[slot] : r->reserved;
      if (L + 1 > cap) { cap = L + 64; seq = (char *)realloc(seq, cap); qual = (char *)realloc(qual, cap); }
      if (r->strand) { memcpy(seq, bs, L); memcpy(qual, qs, L); }
      else for (uint32_t t = 0; t < L; ++t) { seq[t] = u2c((uint8_t)(3 ^ c2u(bs[L - 1 - t]))); qual[t] = qs[L - 1 - t]; }
      /* sequence length deduced from the CIGAR (sam_mapping.h:180-188) */
      const uint32_t *cg = cigar_pool + (size_t)slot * ORA_SAM_CIGAR_CAP;
      uint32_t ql = 0;
      for (int ci = 0; ci < r->n_cigar; ++ci) { const uint32_t op = cg[ci] & 0xf; if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += cg[ci] >> 4; }
      if (ql != L && ql < L) L = ql;
      seq[L] = 0; qual[L] = 0;
      fprintf(f, "%s\t%u\t%s\t%u\t%u\t", name, (unsigned)r->flag, ref->name[r->rid], r->pos + 1, (unsigned)r->mapq);
      if (r->n_cigar == 0) fputc('*', f);
      for (int ci = 0; ci < r->n_cigar; ++ci) fprintf(f, "%u%c", cg[ci] >> 4, "MIDNSHP=XB"[cg[ci] & 0xf]);
      fprintf(f, "\t%s\t%u\t%d\t%s\t%s\tNM:i:%u\tMD:Z:%.*s", r->mrid < 0 ? "*" : ((uint32_t)r->mrid == r->rid ? "=" : ref->name[r->mrid]),
              r->mrid < 0 ? 0u : r->mpos + 1, r->tlen, seq, qual, r->nm, (int)r->md_len, md_pool + (size_t)slot * md_cap);
      if (bck) { /* mapping_writer.cc:350-354, Seed2Sequence */
        fputs("\tCB:Z:", f);
        for (uint32_t b = 0; b < bc_len; ++b) fputc("ACGT"[(last.bc >> ((bc_len - 1 - b) * 2)) & 3], f);
      }
      fputc('\n', f);
      ++lines;
    }
    i = j;
  }
  free(seq); free(qual); free(v);
  fclose(f);
  return lines;
}

long ora_write_sam(const ora_ref *ref, const ora_params *p, const ora_sam_record *rec, long n_slots, int paired,
                   const uint32_t *cigar_pool, const char *md_pool, uint32_t md_cap, const char *const *names1,
                   const char *const *names2, const char *b1, const char *q1, const uint32_t *o1, const char *b2,
                   const char *q2, const uint32_t *o2, const uint32_t *len_after_trim, const char *out_path) {
  return write_sam_impl(ref, p, rec, n_slots, paired, cigar_pool, md_pool, md_cap, names1, names2, b1, q1, o1, b2, q2, o2, len_after_trim,
                        NULL, 0, out_path);
}

long ora_write_sam_bc(const ora_ref *ref, const ora_params *p, const ora_sam_record *rec, long n_slots, int paired,
                      const uint32_t *cigar_pool, const char *md_pool, uint32_t md_cap, const char *const *names1,
                      const char *const *names2, const char *b1, const char *q1, const uint32_t *o1, const char *b2,
                      const char *q2, const uint32_t *o2, const uint64_t *barcode_keys, uint32_t barcode_length, const char *out_path) {
  return write_sam_impl(ref, p, rec, n_slots, paired, cigar_pool, md_pool, md_cap, names1, names2, b1, q1, o1, b2, q2, o2, NULL,
                        barcode_keys, barcode_length, out_path);
}

/* --SAM with cell barcodes: CorrectBarcodeAt in front of the taskloop body (chromap.h:896-909), the
 * corrected key of every pair goes to keys_per_pair (SAMMapping::cell_barcode_) */
long ora_map_pairs_bc_sam(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r1, const uint32_t *r1_off,
                          const char *r2, const uint32_t *r2_off, char *bc, const char *bc_qual, const uint32_t *bc_off,
                          const ora_whitelist *w, ora_sam_record *out, uint32_t *cigar_pool, char *md_pool, uint32_t md_cap,
                          uint64_t *keys_per_pair, ora_stats *stats) {
  c->wl = w; c->bc = bc; c->bcq = bc_qual; c->bco = bc_off;
  c->bc_key = (uint64_t *)calloc((size_t)n + 1, 8);
  c->n_in_wl = c->n_corr = 0;
  const long k = ora_map_pairs_sam(c, threads, n, first_read_id, r1, r1_off, r2, r2_off, out, cigar_pool, md_pool, md_cap, stats);
  memcpy(keys_per_pair, c->bc_key, (size_t)n * 8);
  free(c->bc_key);
  c->wl = NULL; c->bc = NULL; c->bcq = NULL; c->bco = NULL; c->bc_key = NULL;
  return k;
}

/* ------------------------------------------------------------------------- */
/* single-end reads with cell barcodes: MappingWithBarcode (bed_mapping.h:10-56)               */
/* ------------------------------------------------------------------------- */
long ora_map_single_bc(ora_ctx *c, int threads, uint32_t n, uint32_t first_read_id, const char *r, const uint32_t *r_off,
                       char *bc, const char *bc_qual, const uint32_t *bc_off, const ora_whitelist *w, ora_record_bc *out,
                       ora_stats *stats, uint64_t *num_in_whitelist, uint64_t *num_corrected) {
  ora_record *tmp = (ora_record *)malloc(((size_t)n * (size_t)(c->p.max_num_best_mappings > 0 ? c->p.max_num_best_mappings : 1) + 1) * sizeof(ora_record));
  c->wl = w; c->bc = bc; c->bcq = bc_qual; c->bco = bc_off;
  c->bc_key = (uint64_t *)calloc((size_t)n + 1, 8);
  c->n_in_wl = c->n_corr = 0;
  const long k = ora_map_single(c, threads, n, first_read_id, r, r_off, tmp, stats);
  for (long i = 0; i < k; ++i) {
    out[i].r = tmp[i];
    out[i].barcode = c->bc_key[tmp[i].read_id - first_read_id];
  }
  if (num_in_whitelist) *num_in_whitelist += c->n_in_wl;
  if (num_corrected) *num_corrected += c->n_corr;
  free(c->bc_key); free(tmp);
  c->wl = NULL; c->bc = NULL; c->bcq = NULL; c->bco = NULL; c->bc_key = NULL;
  return k;
}

/* operator< (bed_mapping.h:32-38) under the per-chromosome vectors */
static int cmp_se_bc(const void *a, const void *b) {
  const ora_record_bc *x = (const ora_record_bc *)a, *y = (const ora_record_bc *)b;
#define CMPF(f) if (x->f != y->f) return x->f < y->f ? -1 : 1
  CMPF(r.rid); CMPF(r.fragment_start); CMPF(r.fragment_length); CMPF(barcode); CMPF(r.mapq); CMPF(r.direction);
  CMPF(r.is_unique); CMPF(r.read_id);
#undef CMPF
  return 0;
}
static int se_bc_equal(const ora_record_bc *x, const ora_record_bc *y) { /* operator== (:39-42) */
  return x->barcode == y->barcode && x->r.fragment_start == y->r.fragment_start;
}
static int se_bc_same_position(const ora_record_bc *x, const ora_record_bc *y) { /* :43-46 */
  return x->r.fragment_start == y->r.fragment_start;
}
static void se_bc_tn5(ora_record_bc *x) { /* :48-54 */
  if (x->r.direction == 1) x->r.fragment_start += 4; else x->r.fragment_length = (uint16_t)(x->r.fragment_length - 5);
}
static void se_bc_print(FILE *f, const ora_ref *ref, const ora_record_bc *x, uint32_t barcode_length, int tagalign) {
  if (tagalign) { /* mapping_writer.cc:26-34 */
    fprintf(f, "%s\t%u\t%u\tN\t%u\t%c\n", ref->name[x->r.rid], x->r.fragment_start, x->r.fragment_start + x->r.fragment_length,
            (unsigned)x->r.mapq, x->r.direction ? '+' : '-');
    return;
  }
  char bcs[40]; /* mapping_writer.cc:14-25, Seed2Sequence */
  for (uint32_t b = 0; b < barcode_length; ++b) bcs[b] = u2c((uint8_t)((x->barcode >> ((barcode_length - 1 - b) * 2)) & 3));
  bcs[barcode_length] = 0;
  fprintf(f, "%s\t%u\t%u\t%s\t%u\n", ref->name[x->r.rid], x->r.fragment_start, x->r.fragment_start + x->r.fragment_length, bcs,
          (unsigned)x->r.num_dups);
}
/* FindBestMappingIndexFromDuplicates (mapping_writer.h:125-163) */
static size_t se_bc_best_of(const ora_whitelist *w, const ora_record_bc *d, size_t nd) {
  size_t best = 0;
  int found = 0;
  uint32_t slot = wl_slot(w, d[0].barcode, &found);
  double best_ab = found ? (double)w->cnt[slot] : 0.0;
  for (size_t i = 1; i < nd; ++i) {
    slot = wl_slot(w, d[i].barcode, &found);
    const double ab = found ? (double)w->cnt[slot] : 0.0;
    if (d[i].r.num_dups > d[best].r.num_dups || (d[i].r.num_dups == d[best].r.num_dups && ab > best_ab)) { best = i; best_ab = ab; }
  }
  return best;
}

/* BED / TagAlign for MappingWithBarcode.  low_mem: the merge loop of ProcessAndOutputMappingsInLowMemory
 * (mapping_writer.h:166-376) statement by statement over the globally sorted records (its k-way merge of
 * sorted temp files yields exactly that order); otherwise Tn5 shift, sort, RemovePCRDuplicate
 * (chromap.h:594-608, mapping_processor.h:164-202), OutputMappingsInVector (mapping_writer.h:404-437). */
long ora_write_se_bc(const ora_ref *ref, const ora_params *p, ora_record_bc *rec, long n, uint32_t barcode_length,
                     const ora_whitelist *w, int tagalign, const char *out_path) {
  FILE *f = fopen(out_path, "wb");
  if (!f) return -1;
  long lines = 0;
  for (long t = 0; t < n; ++t) rec[t].r.num_dups = 1; /* constructor argument (mapping_generator.cc:27) */
  if (!p->low_mem) {
    if (p->tn5_shift) for (long t = 0; t < n; ++t) se_bc_tn5(&rec[t]);
    qsort(rec, (size_t)n, sizeof(ora_record_bc), cmp_se_bc);
    long i = 0;
    while (i < n) {
      long j = i + 1;
      if (p->remove_pcr_duplicates)
        while (j < n && rec[j].r.rid == rec[i].r.rid && se_bc_equal(&rec[j], &rec[j - 1])) ++j;
      ora_record_bc keep = rec[j - 1]; /* the last of a run of pairwise-equal neighbours */
      if (p->remove_pcr_duplicates) keep.r.num_dups = (uint8_t)(j - i > 255 ? 255 : j - i);
      if (keep.r.mapq >= p->mapq_threshold) { se_bc_print(f, ref, &keep, barcode_length, tagalign); ++lines; }
      i = j;
    }
    fclose(f);
    return lines;
  }
  qsort(rec, (size_t)n, sizeof(ora_record_bc), cmp_se_bc);
  const int bulk = p->remove_pcr_duplicates && p->dedup_at_bulk_level;
  ora_record_bc *dups = (ora_record_bc *)malloc(((size_t)n + 1) * sizeof(ora_record_bc));
  size_t nd = 0;
  uint32_t last_rid = UINT32_MAX, num_last = 0;
  ora_record_bc last;
  memset(&last, 0, sizeof(last));
  for (long t = 0; t < n; ++t) {
    const ora_record_bc *cur = &rec[t];
    const uint32_t min_rid = cur->r.rid;
    const int first = t == 0;
    const int dup_cell = !first && se_bc_equal(cur, &last);
    const int dup_bulk = !first && bulk && se_bc_same_position(cur, &last);
    const int dup = last_rid == min_rid && (dup_cell || dup_bulk);
    if (p->remove_pcr_duplicates && dup) {
      ++num_last;
      if (bulk) {
        if (nd && se_bc_equal(cur, &dups[nd - 1])) { dups[nd - 1] = *cur; dups[nd - 1].r.num_dups += 1; }
        else { dups[nd] = *cur; dups[nd].r.num_dups = 1; ++nd; }
      }
      if (cur->r.mapq > last.r.mapq) last = *cur;
    } else {
      if (!first) {
        if (bulk) { last = dups[se_bc_best_of(w, dups, nd)]; nd = 0; }
        if (last.r.mapq >= p->mapq_threshold) {
          last.r.num_dups = (uint8_t)(num_last > 255 ? 255 : num_last);
          if (p->tn5_shift) se_bc_tn5(&last);
          se_bc_print(f, ref, &last, barcode_length, tagalign);
          ++lines;
        }
      }
      last = *cur;
      last_rid = min_rid;
      num_last = 1;
      if (bulk) { dups[nd] = *cur; dups[nd].r.num_dups = 1; ++nd; }
    }
  }
  if (n > 0 && last.r.mapq >= p->mapq_threshold) {
    if (bulk) { last = dups[se_bc_best_of(w, dups, nd)]; nd = 0; }
    last.r.num_dups = (uint8_t)(num_last > 255 ? 255 : num_last);
    if (p->tn5_shift) se_bc_tn5(&last);
    se_bc_print(f, ref, &last, barcode_length, tagalign);
    ++lines;
  }
  free(dups);
  fclose(f);
  return lines;
}
