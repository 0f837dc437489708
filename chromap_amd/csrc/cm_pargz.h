// cm_pargz.h -- ordinary (single-stream) gzip inflated by several host threads, for the CLI's ingest (kseq sits behind ONE
// gzread per file, sequence_batch.cc:22-62, which caps a .fastq.gz run at the rate of one inflating core -- 4.5 M pairs/s against 25-36 M
// for BGZF, whose blocks the device inflates).  Host code only; zlib does all the decoding.
//
// A deflate stream has no entry points: a block can start at any BIT, and its back-references reach up to 32 KiB into output nobody has
// produced yet.  The scheme (after pugz / rapidgzip, restated for zlib's own inflate):
//   * the compressed bytes are cut into chunks; every chunk but the first SEARCHES the first position at or behind its start where a
//     dynamic-Huffman block header is well-formed (complete code-length code, complete literal/length and distance codes, an end-of-block
//     symbol: the tests of zlib's inflate_table) and from which zlib decodes on;
//   * it then decodes from there to the first block boundary at or behind the next chunk's start -- TWICE, each time with a different
//     made-up 32 KiB dictionary standing in for the unknown window: dict_A[k] = k & 255, dict_B[k] = 128 | k >> 8.  A byte that came
//     from the stream itself is the same in both outputs; a byte copied (directly or through any chain of copies) from window position
//     k reads (k & 255, 128 | k >> 8) -- A != B marks it, (A, B & 127) name k.  The one collision -- a byte >= 128 that is equal in both:
//     a literal of binary data, or one of the 128 positions with k & 255 == 128 | k >> 8 -- is settled by a third decode with
//     dict_C[k] = 255 ^ (k & 255) (A != C marks exactly the window bytes); text never needs it;
//   * the consumer walks the chunks in order: the first chunk of a group is decoded with the TRUE window (it is known by then) to the first
//     boundary at or behind the second chunk's start; a speculative chunk is accepted iff its start is EXACTLY where the decode before it
//     ended -- then its start was a real block boundary and its decode is the real one, its marked bytes are filled in from the real
//     window; otherwise the stretch is decoded again from the known position.  Nothing that was guessed is ever handed out: a wrong guess
//     costs time, not correctness.  Every member's CRC-32 and length are checked against its trailer as gzread does.
// What it does not take on (open() says no, the caller uses gzread): files under 16 MiB, anything but deflate.  Concatenated members are
// followed; a file of many small members just runs at the serial rate.
#ifndef CM_PARGZ_H_
#define CM_PARGZ_H_

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <thread>
#include <vector>

struct ParGunzip {
  // compressed bytes per chunk (CM_PARGZ_CHUNK_KB: measurements).  A chunk's two decodes write ~11 bytes per compressed byte into buffers
  // that are reused by every other group: the smaller the chunks, the less memory a short job touches for the first time
  size_t kChunk = getenv("CM_PARGZ_CHUNK_KB") && atol(getenv("CM_PARGZ_CHUNK_KB")) >= 64 ? (size_t)atol(getenv("CM_PARGZ_CHUNK_KB")) << 10 : (size_t)2 << 20;
  static constexpr uint32_t kWin = 32768;
  const uint8_t *z = nullptr;  // the file, mapped
  size_t zn = 0;
  int fd = -1;
  int threads = 8;
  bool active = false, done = false;
  std::string error;           // set when the file is corrupt (the caller dies with it, as it does for gzread's errors)
  uint64_t pos = 0;            // bit position the accepted output ends at (a block boundary; a member's first block after its header)
  std::vector<uint8_t> win;    // the last <= 32 KiB of accepted output of the current member (its true window)
  uint32_t crc = 0;            // of the current member's accepted output
  uint64_t member_out = 0;
  // ---- teams of threads that live as long as the file is open.  (Round 6: a team made anew for every group and every finish -- a finish of
  // 81 MB whose threads' work adds up to 130 ms took 7-10 ms or 67-93 ms, depending on how long the new threads took to appear: making
  // a thread maps its stack, which waits for the address space's lock behind the page faults of 60-odd busy threads; more threads made the
  // whole job slower.)  submit(f): every worker runs f once; wait(): until all have.
  struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void()> job;
    uint64_t gen = 0;
    int running = 0;
    bool quit = false;
    void start(int n) {
      stop();
      quit = false;
      for (int i = 0; i < n; ++i)
        th.emplace_back([this]() {
          uint64_t seen = 0;
          for (;;) {
            std::function<void()> f;
            {
              std::unique_lock<std::mutex> lk(m);
              cv.wait(lk, [&]() { return quit || gen != seen; });
              if (quit) return;
              seen = gen;
              f = job;
            }
            f();
            std::lock_guard<std::mutex> lk(m);
            if (--running == 0) cv_done.notify_all();
          }
        });
    }
    void submit(const std::function<void()> &f) {
      std::lock_guard<std::mutex> lk(m);
      job = f;
      running = (int)th.size();
      ++gen;
      cv.notify_all();
    }
    void wait() {
      std::unique_lock<std::mutex> lk(m);
      cv_done.wait(lk, [&]() { return running == 0; });
    }
    void stop() {
      { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
      for (std::thread &x : th) x.join();
      th.clear();
      running = 0;
    }
    ~Pool() { stop(); }
  };
  Pool pool_decode, pool_finish, pool_ahead;  // threads - 1 each (the caller is the team's last member), and the one that decodes ahead

  // accepted output not yet taken by read(): a plain growing buffer (a std::vector would zero-fill hundreds of MB that the finishing
  // threads are about to write)
  struct Raw {
    uint8_t *p = nullptr;
    size_t len = 0, cap = 0;
    size_t size() const { return len; }
    uint8_t *data() { return p; }
    void clear() { len = 0; }
    void grow(size_t more) {
      if (cap - len < more) {
        size_t nc = cap ? cap : (size_t)64 << 20;
        while (nc - len < more) nc *= 2;
        void *q = nullptr;
        if (posix_memalign(&q, (size_t)2 << 20, nc) != 0 || !q) { fprintf(stderr, "out of memory (gunzip buffers)\n"); abort(); }
        if (len) memcpy(q, p, len);
        free(p);
        p = static_cast<uint8_t *>(q);
        cap = nc;
      }
      len += more;
    }
    Raw() = default;
    Raw(const Raw &) = delete;
    Raw &operator=(const Raw &) = delete;
    ~Raw() { free(p); }
  } spill;
  size_t spill_off = 0;
  // statistics (tests, --inflate-only)
  uint64_t n_spec = 0, n_accepted = 0, n_serial = 0;
  double t_decode = 0, t_chain = 0, t_finish = 0;  // seconds in the groups' three phases (CM_PARGZ_DEBUG prints them)
  static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

  // ---- bit reader over the mapped file (deflate packs bits LSB first)
  inline uint32_t bits(uint64_t bit, int n) const {  // n <= 24 bits at `bit`; past the end reads zeros
    const size_t b = (size_t)(bit >> 3);
    uint64_t v = 0;
    for (int i = 0; i < 5; ++i) v |= (uint64_t)(b + (size_t)i < zn ? z[b + i] : 0) << (8 * i);
    return (uint32_t)((v >> (bit & 7)) & ((1u << n) - 1u));
  }
  // is there a well-formed dynamic-block header (BFINAL = 0) at `bit`?  The checks of zlib's inflate (inflate.c: TABLE .. CODELENS)
  bool plausible_header(uint64_t bit) const {
    if (bits(bit, 3) != 4u) return false;  // BFINAL 0, BTYPE 2 (binary 10, LSB first: value 4)
    const uint32_t h = bits(bit + 3, 14);
    const uint32_t hlit = (h & 31u) + 257, hdist = ((h >> 5) & 31u) + 1, hclen = (h >> 10) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    uint64_t p = bit + 17;
    for (uint32_t i = 0; i < hclen; ++i, p += 3) cl[order[i]] = (uint8_t)bits(p, 3);
    // the code-length code must be complete (inflate_table, type CODES: an incomplete set is an error)
    uint32_t cnt[8] = {0};
    for (int s = 0; s < 19; ++s) cnt[cl[s]]++;
    int left = 1;
    for (int l = 1; l <= 7; ++l) { left = (left << 1) - (int)cnt[l]; if (left < 0) return false; }
    if (left != 0) return false;
    // canonical codes of the code-length code, decoded bit by bit (at most 7 bits)
    uint32_t next[9] = {0}, code = 0;
    cnt[0] = 0;
    for (int l = 1; l <= 7; ++l) { code = (code + cnt[l - 1]) << 1; next[l] = code; }
    uint32_t codes[19];
    for (int s = 0; s < 19; ++s) codes[s] = cl[s] ? next[cl[s]]++ : 0;
    auto decode = [&](uint64_t &q) -> int {
      uint32_t c = 0;
      for (int l = 1; l <= 7; ++l) {
        c = (c << 1) | bits(q + (uint64_t)l - 1, 1);
        for (int s = 0; s < 19; ++s) if (cl[s] == l && codes[s] == c) { q += (uint64_t)l; return s; }
      }
      return -1;
    };
    uint8_t len[320];
    uint32_t n = 0;
    const uint32_t total = hlit + hdist;
    while (n < total) {
      if ((p >> 3) + 8 > zn) return false;
      const int s = decode(p);
      if (s < 0) return false;
      if (s < 16) { len[n++] = (uint8_t)s; continue; }
      uint32_t rep, v = 0;
      if (s == 16) { if (n == 0) return false; v = len[n - 1]; rep = 3 + bits(p, 2); p += 2; }
      else if (s == 17) { rep = 3 + bits(p, 3); p += 3; }
      else { rep = 11 + bits(p, 7); p += 7; }
      if (n + rep > total) return false;
      while (rep--) len[n++] = (uint8_t)v;
    }
    if (len[256] == 0) return false;  // no end-of-block code
    auto complete = [](const uint8_t *l, uint32_t m) {  // inflate_table, types LENS / DISTS: over-subscribed is an error, incomplete only with one 1-bit code
      uint32_t c[16] = {0};
      uint32_t mx = 0;
      for (uint32_t i = 0; i < m; ++i) { c[l[i]]++; if (l[i] > mx) mx = l[i]; }
      if (mx == 0) return true;
      int lf = 1;
      for (int b = 1; b <= 15; ++b) { lf = (lf << 1) - (int)c[b]; if (lf < 0) return false; }
      return lf == 0 || mx == 1;
    };
    return complete(len, hlit) && complete(len + hlit, hdist);
  }

  // ---- one zlib stream decoding from a bit position, block by block
  struct Dec {
    z_stream s;
    const uint8_t *base = nullptr;  // the mapped file
    size_t zn = 0;
    bool on = false, ended = false;
    uint64_t at = 0;                // bit position of the last block boundary reached
    // the output: a plain growing buffer (a std::vector would zero-fill what zlib is about to overwrite, on every call)
    struct Buf {
      uint8_t *p = nullptr;
      size_t len = 0, cap = 0;
      size_t first_cap = (size_t)4 << 20;  // (a chunk's decodes: set to ~6 x the chunk's compressed size -- FASTQ inflates ~5-fold; growing by doubling touched 4 + 8 + 16 MB for 11)
      uint8_t *data() { return p; }
      const uint8_t *data() const { return p; }
      size_t size() const { return len; }
      void clear() { len = 0; }
      // (2 MiB-aligned; NOT advised as huge pages: where the system compacts memory on such a fault -- transparent_hugepage/defrag = madvise,
      //  as on the boxes here -- the faults of a team of threads cost more than the small pages they save: a 40 MB file in this round's
      //  container, 4 threads: 19 s with the advice, of which 31 s of system time, against 2.5 s for gzread)
      void room(size_t more) {
        if (cap - len >= more) return;
        size_t nc = cap ? cap * 2 : first_cap;
        while (nc - len < more) nc *= 2;
        void *q = nullptr;
        if (posix_memalign(&q, (size_t)2 << 20, nc) != 0 || !q) { fprintf(stderr, "out of memory (gunzip buffers)\n"); abort(); }
        if (len) memcpy(q, p, len);
        free(p);
        p = static_cast<uint8_t *>(q);
        cap = nc;
      }
      uint8_t &operator[](size_t i) { return p[i]; }
      const uint8_t &operator[](size_t i) const { return p[i]; }
      Buf() = default;
      Buf(const Buf &) = delete;
      Buf &operator=(const Buf &) = delete;
      ~Buf() { free(p); }
    } out;
    bool start(const uint8_t *z, size_t n, uint64_t bit, const uint8_t *dict, uint32_t dict_len) {
      base = z; zn = n;
      memset(&s, 0, sizeof(s));
      if (inflateInit2(&s, -15) != Z_OK) return false;
      on = true; ended = false; at = bit;
      out.clear();
      if (dict_len && inflateSetDictionary(&s, dict, dict_len) != Z_OK) return false;
      size_t byte = (size_t)(bit >> 3);
      const int used = (int)(bit & 7);
      if (used) {  // the rest of the byte the block starts in
        if (byte >= zn || inflatePrime(&s, 8 - used, z[byte] >> used) != Z_OK) return false;
        ++byte;
      }
      s.next_in = const_cast<Bytef *>(z + byte);
      s.avail_in = 0;
      return true;
    }
    // decodes until the first block boundary at or behind `stop` (or the stream's end); false: not a deflate stream here
    bool run(uint64_t stop) {
      for (;;) {
        out.room((size_t)1 << 16);
        const size_t have = out.len, room = out.cap - have;
        s.next_out = out.p + have;
        s.avail_out = (uInt)(room > (1u << 30) ? (1u << 30) : room);
        const size_t in_left = (size_t)(base + zn - s.next_in);
        if (s.avail_in == 0) s.avail_in = (uInt)(in_left > (1u << 30) ? (1u << 30) : in_left);
        const uInt out0 = s.avail_out;
        const int rc = inflate(&s, Z_BLOCK);
        out.len = have + (out0 - s.avail_out);
        if (rc == Z_STREAM_END) { ended = true; at = (uint64_t)(s.next_in - base) * 8; return true; }  // (the trailer starts on a byte)
        if (rc != Z_OK && rc != Z_BUF_ERROR) return false;
        if (rc == Z_BUF_ERROR && s.avail_in == 0 && (size_t)(base + zn - s.next_in) == 0 && s.avail_out != 0) return false;  // truncated
        if ((s.data_type & 128) && !(rc == Z_BUF_ERROR)) {
          // (the end of the stream's LAST block -- bit 64 -- is not a place to stop: what follows is padding and the trailer, not a block.
          //  The next call returns Z_STREAM_END.  Round 6: a member whose last block straddled a chunk's end -- ~2 % of members with 2 MiB
          //  chunks -- stopped here as if more blocks followed, and the decode "from the next block" failed on a valid file; found by
          //  tests/test_cli_cpu.py::test_pipelined_gunzip_over_members_threads_and_chunk_sizes with 64 KiB chunks)
          if (s.data_type & 64) continue;
          at = (uint64_t)(s.next_in - base) * 8 - (uint64_t)(s.data_type & 63);
          if (at >= stop) return true;
        }
      }
    }
    void stop() { if (on) { inflateEnd(&s); on = false; } }
  };

  struct Task {
    size_t r0 = 0, r1 = 0;   // the chunk's compressed byte range
    bool have = false;        // a start was found and the three decodes agree on where they ended
    uint64_t S = 0, E = 0;
    bool ended = false;
    bool three = false;       // the third decode was needed (see decode_spec)
    Dec d[3];
  };

  // ---- gzip framing
  bool parse_header(size_t *off) const {  // RFC 1952; *off: the member's first byte -> its first deflate byte
    size_t p = *off;
    if (p + 18 > zn || z[p] != 0x1f || z[p + 1] != 0x8b || z[p + 2] != 8) return false;
    const uint8_t flg = z[p + 3];
    p += 10;
    if (flg & 4) { if (p + 2 > zn) return false; p += 2 + ((size_t)z[p] | ((size_t)z[p + 1] << 8)); }
    if (flg & 8) { while (p < zn && z[p]) ++p; ++p; }
    if (flg & 16) { while (p < zn && z[p]) ++p; ++p; }
    if (flg & 2) p += 2;
    if (p >= zn) return false;
    *off = p;
    return true;
  }

  bool open(const char *path, int nthreads) {
    struct stat sb;
    // (files under 16 MiB: not worth the teams; CM_PARGZ_MIN_KB: the tests' smaller files)
    const size_t min_bytes = getenv("CM_PARGZ_MIN_KB") ? (size_t)atol(getenv("CM_PARGZ_MIN_KB")) << 10 : (size_t)16 << 20;
    if (stat(path, &sb) != 0 || !S_ISREG(sb.st_mode) || (size_t)sb.st_size < min_bytes) return false;
    fd = ::open(path, O_RDONLY);
    if (fd < 0) return false;
    zn = (size_t)sb.st_size;
    void *m = mmap(nullptr, zn, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { ::close(fd); fd = -1; return false; }
    (void)madvise(m, zn, MADV_SEQUENTIAL);
    z = static_cast<const uint8_t *>(m);
    size_t off = 0;
    if (!parse_header(&off)) { close(); return false; }
    pos = (uint64_t)off * 8;
    threads = nthreads < 2 ? 2 : (nthreads > 64 ? 64 : nthreads);
    win.clear(); crc = (uint32_t)crc32(0L, Z_NULL, 0); member_out = 0;
    active = true; done = false;
    pool_decode.start(threads - 1);
    pool_finish.start(threads - 1);
    pool_ahead.start(1);
    return true;
  }
  ParGunzip() = default;
  ParGunzip(const ParGunzip &) = delete;
  ParGunzip &operator=(const ParGunzip &) = delete;
  ~ParGunzip() { close(); }  // (the teams stop before the buffers they write into go)
  void close() {
    drop_ahead();  // (a team may still be decoding ahead, out of the mapping below)
    pool_ahead.stop(); pool_decode.stop(); pool_finish.stop();
    if (z) munmap(const_cast<uint8_t *>(z), zn);
    z = nullptr;
    if (fd >= 0) ::close(fd);
    fd = -1;
    active = false;
    for (Group &g : grp_) { g.tk.reset(); g.n = 0; free(g.head.out.p); g.head.out.p = nullptr; g.head.out.len = g.head.out.cap = 0; }
    cur_ = 0;
  }

  // ---- accepted output.  The consumer walks a group's chunks in order and only DECIDES (which decode counts, which window it sees);
  // the bytes are moved, filled in and check-summed afterwards by all threads together (finish_pieces): one thread doing that for a
  // gigabyte of text would take as long as inflating it.
  struct Piece {
    Dec::Buf *a = nullptr;        // the bytes (a speculative chunk: the decode with dictionary A)
    const Dec::Buf *b = nullptr;  // ... with dictionary B (nullptr: nothing to fill in)
    const Dec::Buf *c = nullptr;  // ... with dictionary C (nullptr: A != B marks a window byte, see decode_spec)
    std::vector<uint8_t> w;       // the true window in front of the piece (kWin bytes, right-aligned)
    size_t n = 0;
    uint32_t crc = 0;
  };
  std::vector<Piece> pieces;
  std::vector<Dec::Buf *> owned;  // buffers of serial decodes, freed after the pieces are finished

  static inline uint8_t fill(const Piece &p, size_t x) {
    const uint8_t va = (*p.a)[x], vb = (*p.b)[x];
    const bool marked = p.c ? va != (*p.c)[x] : va != vb;
    return marked ? p.w[(size_t)va | ((size_t)(vb & 127u) << 8)] : va;
  }
  // the window behind piece p (for the next one): the last kWin bytes of (p.w + p's filled-in bytes)
  void window_after(const Piece &p) {
    if (p.n >= kWin) {
      win.resize(kWin);
      for (size_t x = 0; x < kWin; ++x) win[x] = p.b ? fill(p, p.n - kWin + x) : (*p.a)[p.n - kWin + x];
    } else {
      std::vector<uint8_t> t(win);
      for (size_t x = 0; x < p.n; ++x) t.push_back(p.b ? fill(p, x) : (*p.a)[x]);
      if (t.size() > kWin) t.erase(t.begin(), t.begin() + (long)(t.size() - kWin));
      win.swap(t);
    }
  }
  void add_piece(Dec::Buf *a, const Dec::Buf *b, const Dec::Buf *c) {
    if (a->len == 0) return;
    Piece p;
    p.a = a; p.b = b; p.c = c; p.n = a->len;
    if (b) { p.w.assign(kWin, 0); if (!win.empty()) memcpy(p.w.data() + (kWin - win.size()), win.data(), win.size()); }
    window_after(p);
    pieces.push_back(std::move(p));
  }
  // all pieces so far: filled in, copied to the caller's buffer (or the spill) and check-summed, by `threads` threads; then the
  // member's running CRC and length
  void finish_pieces(unsigned char *dst, size_t want, size_t *got) {
    if (pieces.empty()) return;
    const double t_f0 = now();
    size_t total = 0;
    std::vector<size_t> at(pieces.size());
    for (size_t i = 0; i < pieces.size(); ++i) { at[i] = total; total += pieces[i].n; }
    // work items of <= 1 MiB so that the threads share a group evenly.  The caller's buffer takes the items that fit it whole, in order;
    // the rest goes to the spill (round 6: a group that did not fit went there whole -- ~340 MB against requests of 256 MB: a zero-filled
    // resize and a second copy, both by one thread, for every group)
    struct Item { size_t piece, x0, x1, g; uint8_t *o; };  // g: the item's offset in the group's output; o: where its first byte goes
    std::vector<Item> items;
    for (size_t i = 0; i < pieces.size(); ++i)
      for (size_t x = 0; x < pieces[i].n; x += (size_t)1 << 20)
        items.push_back({i, x, x + ((size_t)1 << 20) < pieces[i].n ? x + ((size_t)1 << 20) : pieces[i].n, at[i] + x, nullptr});
    const size_t room = spill_off >= spill.size() ? want - *got : 0;  // (bytes still in the spill come first: nothing goes to dst then)
    size_t cut = 0;  // the first `cut` bytes go to dst
    for (const Item &it : items) { if (it.g + (it.x1 - it.x0) <= room) cut = it.g + (it.x1 - it.x0); else break; }
    if (spill_off >= spill.size()) { spill.clear(); spill_off = 0; }
    const size_t spill_old = spill.size();
    spill.grow(total - cut);
    for (Item &it : items) it.o = it.g < cut ? dst + *got + it.g : spill.data() + spill_old + (it.g - cut);
    *got += cut;
    std::vector<uint32_t> icrc(items.size());
    std::atomic<size_t> next{0};
    static const bool dbg_f = getenv("CM_PARGZ_DEBUG") != nullptr;
    std::atomic<uint64_t> ns_fill{0}, ns_crc{0};
    auto work = [&]() {
      for (size_t k; (k = next.fetch_add(1)) < items.size();) {
        const Item &it = items[k];
        const Piece &p = pieces[it.piece];
        uint8_t *o = it.o - it.x0;
        const double tf0 = dbg_f ? now() : 0;
        if (p.b && !p.c) {  // eight bytes at a time where the two decodes agree (a window byte is rare past a chunk's first stretch)
          const uint8_t *pa = p.a->p, *pb = p.b->p;
          size_t x = it.x0;
          for (; x + 8 <= it.x1; x += 8) {
            uint64_t va, vb;
            memcpy(&va, pa + x, 8);
            memcpy(&vb, pb + x, 8);
            if (va == vb) memcpy(o + x, &va, 8);
            else for (size_t k = x; k < x + 8; ++k) o[k] = fill(p, k);
          }
          for (; x < it.x1; ++x) o[x] = fill(p, x);
        } else if (p.b) for (size_t x = it.x0; x < it.x1; ++x) o[x] = fill(p, x);
        else memcpy(o + it.x0, p.a->p + it.x0, it.x1 - it.x0);
        const double tf1 = dbg_f ? now() : 0;
        icrc[k] = (uint32_t)crc32(crc32(0L, Z_NULL, 0), o + it.x0, (uInt)(it.x1 - it.x0));
        if (dbg_f) { ns_fill += (uint64_t)((tf1 - tf0) * 1e9); ns_crc += (uint64_t)((now() - tf1) * 1e9); }
      }
    };
    if (items.size() > 1) { pool_finish.submit(work); work(); pool_finish.wait(); }
    else work();
    for (size_t k = 0; k < items.size(); ++k) crc = (uint32_t)crc32_combine(crc, icrc[k], (z_off_t)(items[k].x1 - items[k].x0));
    member_out += total;
    pieces.clear();
    for (Dec::Buf *b : owned) delete b;
    owned.clear();
    t_finish += now() - t_f0;
    if (dbg_f) fprintf(stderr, "[pargz] finish: %zu MB (%zu to the caller, %zu spilled) in %.1f ms; threads' sums: fill + copy %.1f ms, crc %.1f ms\n", total >> 20, cut >> 20,
                       (total - cut) >> 20, (now() - t_f0) * 1e3, (double)ns_fill.load() / 1e6, (double)ns_crc.load() / 1e6);
  }
  // a member ended at byte `pos / 8`: trailer check (the pieces must be finished), then the next member's header (or the end of the file)
  bool member_end() {
    const size_t t = (size_t)(pos >> 3);
    if (t + 8 > zn) { error = "unexpected end of file"; return false; }
    const uint32_t want_crc = (uint32_t)z[t] | ((uint32_t)z[t + 1] << 8) | ((uint32_t)z[t + 2] << 16) | ((uint32_t)z[t + 3] << 24);
    const uint32_t want_len = (uint32_t)z[t + 4] | ((uint32_t)z[t + 5] << 8) | ((uint32_t)z[t + 6] << 16) | ((uint32_t)z[t + 7] << 24);
    if (want_crc != crc) { error = "incorrect data check"; return false; }
    if (want_len != (uint32_t)member_out) { error = "incorrect length check"; return false; }
    size_t off = t + 8;
    while (off < zn && z[off] == 0) ++off;  // (zero padding behind a member, as gzread skips it)
    if (off >= zn) { done = true; return true; }
    if (!parse_header(&off)) { done = true; return true; }  // (trailing garbage: gzread ignores it)
    pos = (uint64_t)off * 8;
    win.clear(); crc = (uint32_t)crc32(0L, Z_NULL, 0); member_out = 0;
    return true;
  }
  // decodes from `pos` with the true window up to the first block boundary at or behind `stop`: the slow path for a stretch no guess covers
  bool serial_to(uint64_t stop, unsigned char *dst, size_t want, size_t *got) {
    Dec d;
    if (!d.start(z, zn, pos, win.data(), (uint32_t)win.size()) || !d.run(stop)) { d.stop(); if (getenv("CM_PARGZ_DEBUG")) fprintf(stderr, "[pargz] serial decode failed from bit %llu to %llu (window %zu bytes)\n", (unsigned long long)pos, (unsigned long long)stop, win.size()); error = "invalid deflate data"; return false; }
    Dec::Buf *keep = new Dec::Buf();
    keep->p = d.out.p; keep->len = d.out.len; keep->cap = d.out.cap;
    d.out.p = nullptr; d.out.len = d.out.cap = 0;
    owned.push_back(keep);
    add_piece(keep, nullptr, nullptr);
    pos = d.at;
    const bool ended = d.ended;
    d.stop();
    ++n_serial;
    if (ended) { finish_pieces(dst, want, got); return member_end(); }
    return true;
  }

  // a chunk's speculative decodes from bit b: with dictionary A and B; when a byte reads the same value >= 128 in both -- a literal of
  // binary data, or window position k with (k & 255) == (128 | k >> 8) -- a third one with dictionary C tells them apart (text never needs it)
  bool decode_spec(Task &t, uint64_t b, const uint8_t *dA, const uint8_t *dB, const uint8_t *dC) {
    const uint64_t stop = (uint64_t)t.r1 * 8;
    if (!t.d[0].start(z, zn, b, dA, kWin) || !t.d[0].run(stop)) { t.d[0].stop(); return false; }  // zlib disagrees: not a block
    t.S = b; t.E = t.d[0].at; t.ended = t.d[0].ended;
    t.d[0].stop();
    bool ok = t.d[1].start(z, zn, b, dB, kWin) && t.d[1].run(stop) && t.d[1].at == t.E && t.d[1].out.len == t.d[0].out.len;
    t.d[1].stop();
    t.three = false;
    if (ok) {
      const uint8_t *pa = t.d[0].out.p, *pb = t.d[1].out.p;
      const size_t n = t.d[0].out.len;
      size_t x = 0;
      for (; x + 8 <= n && !t.three; x += 8) {  // (text: no byte >= 128 at all)
        uint64_t va;
        memcpy(&va, pa + x, 8);
        if (va & 0x8080808080808080ull) for (size_t k = x; k < x + 8; ++k) if ((pa[k] & 128u) && pa[k] == pb[k]) { t.three = true; break; }
      }
      for (; x < n && !t.three; ++x) if ((pa[x] & 128u) && pa[x] == pb[x]) t.three = true;
      if (t.three) {
        ok = t.d[2].start(z, zn, b, dC, kWin) && t.d[2].run(stop) && t.d[2].at == t.E && t.d[2].out.len == n;
        t.d[2].stop();
      }
    }
    t.have = ok;
    return true;
  }

  // the next group of chunks: decoded side by side, walked in order, finished side by side
  // ---- a group of chunks: decoded side by side, walked in order, finished side by side.  Two sets of tasks take turns: while the threads
  // finish group g (fill in, copy, check-sum: 0.15-0.27 s of a 928 MB file's 0.45-0.57 s, round 6), another team already decodes group g + 1
  // -- where it starts and the window in front of it are known as soon as group g's chain has been walked.
  // (the tasks and their output buffers live as long as the file is open: a group's decodes write ~20 MB per chunk, and buffers made
  //  anew for every group were that many fresh pages to fault in -- and to give back -- per group, by all threads at once)
  struct Group {
    std::unique_ptr<Task[]> tk;
    int n = 0, used = 0;
    Dec head;
    bool head_ok = true;
    uint64_t start = 0;          // the bit position the group was decoded from
    std::vector<uint8_t> w;      // the window its head decode saw
    bool pending = false;        // being decoded ahead by pool_ahead's thread (waited for by the next produce(), or by close())
    double t = 0;
  };
  Group grp_[2];
  int cur_ = 0;
  int fruitless = 0;  // groups in a row none of whose guesses counted (stored blocks: incompressible data): after two, plain serial decoding

  // one chunk per thread.  (Three per thread, taken from a counter, to even out the chunks' costs before the group's barrier:
  // measured worse -- 182 MB on 32 threads: decode 207 -> 240 ms, the walk 19 -> 210 ms, .fastq.gz -> BED 0.90 -> 1.47 s.)
  void decode_group(Group &g, uint64_t from, const std::vector<uint8_t> &window) {
    static const std::vector<uint8_t> dictA = [] { std::vector<uint8_t> v(kWin); for (uint32_t k = 0; k < kWin; ++k) v[k] = (uint8_t)(k & 255); return v; }();
    static const std::vector<uint8_t> dictB = [] { std::vector<uint8_t> v(kWin); for (uint32_t k = 0; k < kWin; ++k) v[k] = (uint8_t)(128u | (k >> 8)); return v; }();
    static const std::vector<uint8_t> dictC = [] { std::vector<uint8_t> v(kWin); for (uint32_t k = 0; k < kWin; ++k) v[k] = (uint8_t)(255 ^ (k & 255)); return v; }();
    const double t0 = now();
    const int nt = threads;
    if (!g.tk || g.n != nt) {
      g.tk.reset(new Task[(size_t)nt]);
      g.n = nt;
      for (int i = 0; i < nt; ++i) for (Dec &d : g.tk[i].d) d.out.first_cap = kChunk * 6;
      g.head.out.first_cap = kChunk * 6;
    }
    g.start = from;
    g.w = window;
    g.head_ok = true;
    const size_t g0 = (size_t)(from >> 3);
    Task *tk = g.tk.get();
    int used = 0;
    for (int i = 0; i < nt; ++i) {
      tk[i].have = false; tk[i].S = tk[i].E = 0; tk[i].ended = false; tk[i].three = false;
      tk[i].r0 = g0 + (size_t)i * kChunk;
      tk[i].r1 = tk[i].r0 + kChunk;
      if (tk[i].r0 >= zn) break;
      ++used;
    }
    g.used = used;
    // task 0: the true decode from `from`; tasks 1..: search + speculative decodes
    std::atomic<int> next_chunk{0};
    auto worker = [&]() {
      for (int i; (i = next_chunk.fetch_add(1)) < used;) {
        if (i == 0) { g.head_ok = g.head.start(z, zn, from, g.w.data(), (uint32_t)g.w.size()) && g.head.run((uint64_t)tk[0].r1 * 8); continue; }
        Task &t = tk[i];
        const uint64_t b0 = (uint64_t)t.r0 * 8, b1 = (uint64_t)(t.r1 < zn ? t.r1 : zn) * 8;
        for (uint64_t b = b0; b < b1; ++b)
          if (plausible_header(b) && decode_spec(t, b, dictA.data(), dictB.data(), dictC.data())) break;
      }
    };
    if (used > 1) { pool_decode.submit(worker); worker(); pool_decode.wait(); }
    else worker();
    g.t = now() - t0;
  }
  void drop_ahead() {
    for (Group &g : grp_) if (g.pending) { pool_ahead.wait(); g.pending = false; g.head.stop(); }
  }

  bool produce(unsigned char *dst, size_t want, size_t *got) {
    if (fruitless >= 2) {
      drop_ahead();
      if (!serial_to(pos + ((uint64_t)64 << 23), dst, want, got)) return false;
      finish_pieces(dst, want, got);
      return true;
    }
    const uint64_t accepted_before = n_accepted;
    Group &g = grp_[cur_];
    if (g.pending) {
      const double tw = now();
      pool_ahead.wait();
      g.pending = false;
      t_decode += now() - tw;  // (what was left of it to wait for)
      if (g.start != pos || g.w != win) { g.head.stop(); decode_group(g, pos, win); t_decode += g.t; }  // (never seen: the chain left pos / win as handed over)
    } else {
      decode_group(g, pos, win);
      t_decode += g.t;
    }
    Task *tk = g.tk.get();
    const int used = g.used;
    Dec &head = g.head;
    const bool head_ok = g.head_ok;
    const double t_c0 = now();
    n_spec += (uint64_t)(used > 1 ? used - 1 : 0);
    if (!head_ok) { head.stop(); if (getenv("CM_PARGZ_DEBUG")) fprintf(stderr, "[pargz] head decode failed from bit %llu (window %zu bytes)\n", (unsigned long long)g.start, g.w.size()); error = "invalid deflate data"; return false; }
    add_piece(&head.out, nullptr, nullptr);
    pos = head.at;
    {
      const bool ended = head.ended;
      head.stop();
      if (ended) { finish_pieces(dst, want, got); if (!member_end()) return false; }
    }
    // the chain: a speculative chunk counts iff it starts exactly where the accepted output ends
    static const bool dbg = getenv("CM_PARGZ_DEBUG") != nullptr;
    for (int i = 1; i < used && !done; ++i) {
      Task &t = tk[i];
      if (dbg) fprintf(stderr, "[pargz] chunk at byte %zu: have %d S %llu E %llu ended %d three %d | pos %llu\n", t.r0, (int)t.have, (unsigned long long)t.S,
                       (unsigned long long)t.E, (int)t.ended, (int)t.three, (unsigned long long)pos);
      if (t.have && t.S > pos && t.S < (uint64_t)t.r1 * 8) {  // a stretch the search skipped (stored / fixed blocks): decode up to the guess
        if (!serial_to(t.S, dst, want, got)) return false;
        if (done) break;
      }
      if (t.have && t.S == pos) {
        add_piece(&t.d[0].out, &t.d[1].out, t.three ? &t.d[2].out : nullptr);
        pos = t.E;
        ++n_accepted;
        if (t.ended) { finish_pieces(dst, want, got); if (!member_end()) return false; }
      } else if (pos < (uint64_t)t.r1 * 8) {
        if (!serial_to((uint64_t)t.r1 * 8, dst, want, got)) return false;  // no usable guess for this stretch
      }
    }
    t_chain += now() - t_c0;
    if (used > 1) fruitless = n_accepted == accepted_before ? fruitless + 1 : 0;
    // the next group's decodes start now, from where this one's chain ended, into the other set of buffers ...
    static const bool no_ahead = getenv("CM_PARGZ_NO_AHEAD") != nullptr;
    if (!done && fruitless < 2 && !no_ahead) {
      Group &nx = grp_[1 - cur_];
      nx.pending = true;
      const uint64_t from = pos;
      pool_ahead.submit([this, &nx, from, w = win]() { decode_group(nx, from, w); });
    }
    // ... while this one's bytes are filled in, copied and check-summed (its tasks' buffers are written again by the group after next:
    // nothing may be left pointing at them)
    finish_pieces(dst, want, got);
    cur_ = 1 - cur_;
    if (dbg) fprintf(stderr, "[pargz] so far: decode %.3f s (waited for), chain %.3f s (incl. mid-group finishes), finish %.3f s\n", t_decode, t_chain, t_finish);
    return true;
  }

  // up to `want` decompressed bytes, in order; *eof once the last member has ended.  false: error (see `error`)
  bool read(unsigned char *dst, size_t want, size_t *got_out, bool *eof) {
    size_t got = 0;
    while (got < want) {
      if (spill_off < spill.size()) {
        const size_t m = spill.size() - spill_off < want - got ? spill.size() - spill_off : want - got;
        memcpy(dst + got, spill.data() + spill_off, m);
        spill_off += m;
        got += m;
        continue;
      }
      if (done) break;
      if (!produce(dst, want, &got)) { *got_out = got; return false; }
    }
    *got_out = got;
    *eof = done && spill_off >= spill.size();
    return true;
  }
};

#endif  // CM_PARGZ_H_
