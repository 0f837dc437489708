// cm_cli.cpp -- `chromap-amd`: command-line host with chromap's flags for the mapping path
// (chromap_driver.cc:216-761), driving the HIP path through the C ABI.  Host work only:
// argument parsing, FASTQ(.gz) ingest into SoA batches (whole multiples of the reference's
// 500000-pair read batch), statistics, and the post-processing that writes BED / pairs.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <unistd.h>
#include <sys/mman.h>
#include <sched.h>
#include <sys/stat.h>
#include <zlib.h>

#include <chrono>
#include <thread>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/chromap_amd.h"
#include "cm_pargz.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The processors this process may really use: what the hardware reports, cut to the affinity mask and to the control group's CPU quota.
// (Round 6: the measurement boxes report 256 hardware threads and run the job in a container with a quota of 16 CPUs -- cpu.max
//  "1600000 100000".  Teams sized from the 256 -- 2 x 32 inflating threads, as many again finishing -- used the quota up early in every 100 ms
//  period and the kernel then stopped ALL of the process's threads until the period ended: 155 of 837 periods throttled, 60-90 ms stalls in the
//  middle of 7 ms jobs, and "more threads" made everything slower.)
static unsigned cpu_budget() {
  static const unsigned budget = []() {
    unsigned n = std::thread::hardware_concurrency();
    if (!n) n = 8;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < n) n = (unsigned)c; }
    double quota = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota us | max> <period us>"
      char q[64];
      double per = 0;
      if (fscanf(f, "%63s %lf", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) quota = atof(q) / per;
      fclose(f);
    } else {  // cgroup v1
      double qu = 0, per = 0;
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &qu) != 1) qu = 0; fclose(g); }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &per) != 1) per = 0; fclose(g); }
      if (qu > 0 && per > 0) quota = qu / per;
    }
    if (quota >= 1 && quota < (double)n) n = (unsigned)quota;
    if (const char *e = getenv("CM_CPU_BUDGET")) { const int v = atoi(e); if (v > 0) n = (unsigned)v; }
    return n < 2 ? 2u : n;
  }();
  return budget;
}

static void die(const std::string &m) {  // ExitWithMessage (utils.h:71-74)
  fprintf(stderr, "%s\n", m.c_str());
  // (_exit: other threads of the process may be inside the HIP runtime -- the warm-up beside the index read, the worker mapping the last batch
  //  while this thread found the next one damaged -- and exit() would run the runtime's teardown under them)
  fflush(stdout);
  fflush(stderr);
  _exit(255);
}

struct FastxReader {
  gzFile f = nullptr;
  std::vector<char> buf;
  bool open(const std::string &path) {
    f = gzopen(path.c_str(), "r");
    if (f) gzbuffer(f, 1 << 20);
    buf.resize(1 << 16);
    return f != nullptr;
  }
  bool line(std::string &out) {
    out.clear();
    for (;;) {
      if (!gzgets(f, buf.data(), (int)buf.size())) return !out.empty();
      size_t l = strlen(buf.data());
      const bool eol = l > 0 && buf[l - 1] == '\n';
      while (l > 0 && (buf[l - 1] == '\n' || buf[l - 1] == '\r')) --l;
      out.append(buf.data(), l);
      if (eol) return true;
    }
  }
  // one FASTQ/FASTA record: name up to the first whitespace, sequence, quality (may be empty)
  std::string pending;
  // like SequenceBatch::LoadOneSequenceAndSaveAt (sequence_batch.cc:22-62): a record with an empty sequence is skipped, per stream
  bool record(std::string &name, std::string &seq, std::string &qual) {
    while (record_any(name, seq, qual))
      if (!seq.empty()) return true;
    return false;
  }
  bool record_any(std::string &name, std::string &seq, std::string &qual) {
    std::string ln;
    for (;;) {
      if (!pending.empty()) { ln.swap(pending); pending.clear(); }
      else if (!line(ln)) return false;
      if (!ln.empty() && (ln[0] == '@' || ln[0] == '>')) break;
    }
    const bool fq = ln[0] == '@';
    size_t e = 1;
    while (e < ln.size() && ln[e] != ' ' && ln[e] != '\t') ++e;
    name.assign(ln, 1, e - 1);
    seq.clear();
    qual.clear();
    std::string s;
    while (line(s)) {
      if (fq && !s.empty() && s[0] == '+') break;
      if (!fq && !s.empty() && s[0] == '>') { pending = s; break; }
      seq += s;
    }
    if (fq) {
      while (qual.size() < seq.size() && line(s)) qual += s;
    }
    return true;
  }
  void close() { if (f) gzclose(f); f = nullptr; }
};

// raw (inflated) file bytes in large chunks for the device-side FASTQ parser.
// bytes grown without zero-filling (the file's bytes overwrite them); reserve keeps the first `keep` bytes
struct RawBuf {
  unsigned char *p = nullptr;
  size_t cap = 0;
  unsigned char *data() { return p; }
  const unsigned char *data() const { return p; }
  const char *text() const { return reinterpret_cast<const char *>(p); }
  void reserve(size_t n, size_t keep) {
    if (n <= cap) return;
    void *qv = nullptr;
    if (posix_memalign(&qv, (size_t)2 << 20, n) != 0 || !qv) die("out of memory (input buffer)");
    (void)madvise(qv, n, MADV_HUGEPAGE);  // (hundreds of MB touched for the first time: 2 MiB pages where the system gives them)
    unsigned char *q = static_cast<unsigned char *>(qv);
    if (keep) memcpy(q, p, keep);
    free(p);
    p = q;
    cap = n;
  }
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  ~RawBuf() { free(p); }
};
// Plain text is read as it is, ordinary gzip goes through zlib's gzread (one inflating thread per file).  BGZF (bgzip; SAM spec 4.1: a
// series of gzip members of at most 64 KiB, each carrying its compressed size in a 'BC' extra field) is not inflated here when one
// GPU maps: the block headers are walked without decoding and whole compressed blocks go to the device (cmgpu_fastq_scan_bgzf).
// With several GPUs taking turns (the text a batch leaves over lives on one of them) it is inflated block-parallel on the host: the
// ISIZE trailers give every block's place in the output, and a team of threads inflates the blocks of a chunk side by side --
// kseq behind one gzread per file (sequence_batch.cc:22-62) caps the reference's ingest at the rate of one
// inflating core.  Except on that host path, the next piece of the file is read (gzip: inflated) by a thread of its own while the
// device works on the one before.
struct ChunkReader {
  gzFile f = nullptr;
  ParGunzip pg;          // ordinary gzip of 16 MiB and more: inflated by several threads (cm_pargz.h); pargz says it is in use
  bool pargz = false;
  FILE *raw = nullptr;   // BGZF and plain text: the file itself
  bool bgzf = false;
  bool plain = false;    // not gzip at all: `raw` is read directly (gzread would copy the bytes twice)
  int team = 4;          // inflating threads for BGZF input
  int files_side_by_side = 1;  // how many files are read at the same time as this one (read 1, read 2, barcodes): the processors are shared
  RawBuf bufs[2];        // the text: bufs[cur][off .. off + len); the other buffer takes the read-ahead (the two swap: their pages stay mapped)
  int cur = 0;
  std::vector<unsigned char> cbuf;
  size_t off = 0, len = 0;
  bool eof = false;
  // read-ahead: while the device parses and maps what fill() returned, a thread reads (plain text), inflates (gzip) or reads the
  // compressed blocks of (BGZF for the device) the next piece behind the bytes in use; the next fill() takes it over
  std::thread ahead;
  bool ahead_on = false, ahead_eof = false;
  size_t ahead_got = 0;
  const char *text() const { return bufs[cur].text() + off; }
  // how much of the file the records handed out so far came from, 0: unknown (a pipe; gzip inflated by several threads).  What a run uses
  // to size its record store once, after the first batch, instead of doubling it as it fills
  uint64_t fsize = 0, fed = 0;  // fed: bytes of the file whose records have been consumed (BGZF for the device: blocks handed over)
  double fraction() const {
    if (!fsize) return 0;
    if (bgzf && dev_inflate) return (double)(fed + zready) / (double)fsize;
    if (plain) return (double)fed / (double)fsize;
    if (f && !pargz) { const z_off_t o = gzoffset(f); return o > 0 ? (double)o / (double)fsize : 0; }
    return 0;
  }
  // `want` bytes of the file itself from raw's position, which moves on: large reads of a regular file by four threads (one thread copies
  // out of the page cache at ~5 GB/s: 185 MB of BGZF blocks per file and 4 M-pair batch took as long as the device took to inflate and
  // parse the batch before -- 0.11 s of a 32 M-pair job's 0.51 s waiting for read-ahead).  Returns the bytes read (< want: the file's end)
  size_t read_raw(unsigned char *dst, size_t want) {
    const off_t pos = fsize ? ftello(raw) : (off_t)-1;
    if (pos < 0 || want < ((size_t)32 << 20) || getenv("CM_READ_THREADS_1")) return fread(dst, 1, want, raw);
    const size_t avail = fsize > (uint64_t)pos ? (size_t)(fsize - (uint64_t)pos) : 0;
    const size_t n = want < avail ? want : avail;
    const int fd = fileno(raw);
    const int k = 4;
    const size_t per = ((n / (size_t)k) + 4095) & ~(size_t)4095;
    std::atomic<bool> ok{true};
    std::thread th[4];
    auto part = [&](int i) {
      size_t o = per * (size_t)i;
      const size_t end = i == k - 1 ? n : (o + per < n ? o + per : n);
      while (o < end) {
        const ssize_t r = pread(fd, dst + o, end - o, pos + (off_t)o);
        if (r <= 0) { ok = false; return; }
        o += (size_t)r;
      }
    };
    for (int i = 1; i < k; ++i) th[i] = std::thread(part, i);
    part(0);
    for (int i = 1; i < k; ++i) th[i].join();
    if (!ok) die("Didn't reach the end of sequence file, which might be corrupted! (read error)");
    if (fseeko(raw, pos + (off_t)n, SEEK_SET) != 0) die("Didn't reach the end of sequence file, which might be corrupted! (seek error)");
    return n;
  }
  // up to `want` bytes of the (inflated) stream: plain text or gzip
  size_t read_some(unsigned char *dst, size_t want, bool *hit_eof) {
    size_t got = 0;
    while (got < want && !*hit_eof) {
      if (plain) {
        const size_t r = read_raw(dst + got, want - got);
        if (r == 0) { if (ferror(raw)) die("Didn't reach the end of sequence file, which might be corrupted! (read error)"); *hit_eof = true; }
        got += r;
      } else if (pargz) {
        size_t g = 0;
        bool e = false;
        if (!pg.read(dst + got, want - got, &g, &e)) die("Didn't reach the end of sequence file, which might be corrupted! (" + pg.error + ")");
        got += g;
        if (e) *hit_eof = true;
      } else {
        const size_t piece = want - got < (1u << 30) ? want - got : (1u << 30);
        const int r = gzread(f, dst + got, (unsigned)piece);
        if (r < 0) { int en = 0; const char *msg = gzerror(f, &en); die(std::string("Didn't reach the end of sequence file, which might be corrupted! (") + (msg ? msg : "read error") + ")"); }
        if (r == 0) *hit_eof = true; else got += (size_t)r;
      }
    }
    return got;
  }
  void join_ahead() {
    if (!ahead_on) return;
    ahead.join();
    ahead_on = false;
    // (plain text / gzip: fill() joins the two buffers; BGZF for the device: fill_bgzf_compressed does)
  }
  bool open(const std::string &path) {
    off = 0;
    len = 0;
    eof = false;
    bgzf = false;
    plain = false;
    // only a regular file is sniffed for BGZF: the 18 bytes read from a FIFO, a process substitution or /dev/stdin would be
    // lost to the gzopen below (the reference opens every input with one gzopen, which works on pipes)
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0) return false;
    fsize = S_ISREG(sb.st_mode) ? (uint64_t)sb.st_size : 0;
    fed = 0;
    if (!S_ISREG(sb.st_mode)) {
      f = gzopen(path.c_str(), "r");
      if (f) gzbuffer(f, 1 << 20);
      return f != nullptr;
    }
    raw = fopen(path.c_str(), "rb");
    if (!raw) return false;
    unsigned char h[18];
    const size_t got = fread(h, 1, 18, raw);
    if (got == 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0) {
      bgzf = true;
      fseek(raw, 0, SEEK_SET);
      return true;
    }
    if (!(got >= 2 && h[0] == 0x1f && h[1] == 0x8b)) {
      plain = true;
      fseek(raw, 0, SEEK_SET);
      return true;
    }
    fclose(raw);
    raw = nullptr;
    // ordinary gzip: several inflating threads for a file of 16 MiB and more (CM_PARGZ=0: always zlib's gzread; CM_PARGZ_THREADS)
    pargz = false;
    const char *off_env = getenv("CM_PARGZ");
    if (!(off_env && off_env[0] == '0')) {
      // a team per file of the budget over the files read side by side.  (Two teams per file work at a time -- one decodes the next group while
      // the other finishes the last -- so this is twice the budget in threads; measured on a 16-CPU quota, two files: teams of 4 / 6 / 8 / 12 /
      // 32: 1.13 / 0.93 / 0.85 / 0.94 / 0.86-1.06 s end to end)
      const unsigned share = cpu_budget() / (unsigned)(files_side_by_side > 0 ? files_side_by_side : 1);
      int nt = (int)(share < 2 ? 2 : (share > 32 ? 32 : share));
      if (getenv("CM_PARGZ_THREADS")) nt = atoi(getenv("CM_PARGZ_THREADS"));
      if (pg.open(path.c_str(), nt)) { pargz = true; return true; }
    }
    f = gzopen(path.c_str(), "r");
    if (f) gzbuffer(f, 1 << 20);
    return f != nullptr;
  }
  // device inflate (one GPU): the blocks stay compressed -- zdata()[0 .. zready) holds whole blocks whose inflated size adds up to at
  // least `target` more text (the device keeps the text itself, cmgpu_fastq_scan_bgzf), zdata()[zready .. zlen) what was read beyond
  // them (the file is read in large pieces); pending: inflated bytes handed over and not yet taken
  bool dev_inflate = false;
  size_t pending = 0, zoff = 0, zlen = 0, zready = 0;  // (zb[zcur][zoff .. zoff + zlen) is in use)
  // two buffers taking turns: the blocks handed over stay where they are while the device inflates them, the read-ahead goes into the OTHER
  // buffer behind a gap, and the next call copies the few bytes left over in front of it.  (One buffer, round 6's first form: room for the
  // read-ahead meant moving the whole piece in use to the buffer's start -- 185 MB per file and batch, 0.11 s of a 32 M-pair job's 0.51 s)
  static constexpr size_t kZGap = (size_t)72 << 20;  // (what a call may leave over: the 64 MiB its last synchronous read took, and a block)
  RawBuf zb[2];
  int zcur = 0;
  const unsigned char *zdata() const { return zb[zcur].data() + zoff; }
  void fill_bgzf_compressed(size_t target) {
    const bool had_ahead = ahead_on;
    if (ahead_on) { ahead.join(); ahead_on = false; }
    fed += zready;
    zoff += zready;  // (handed over by the last call)
    zlen -= zready;
    zready = 0;
    size_t isum = 0;
    auto room = [&](size_t more) {  // the buffer in use takes `more` bytes behind the ones in use
      RawBuf &z = zb[zcur];
      if (zoff + zlen + more <= z.cap) return;
      if (zoff) { memmove(z.data(), z.data() + zoff, zlen); zoff = 0; }
      if (zlen + more > z.cap) z.reserve(zlen + more + (zlen + more) / 2, zlen);
    };
    if (had_ahead) {
      RawBuf &o = zb[1 - zcur];
      if (zlen <= kZGap) {  // what was left over, in front of what was read ahead
        memcpy(o.data() + kZGap - zlen, zb[zcur].data() + zoff, zlen);
        zcur = 1 - zcur;
        zoff = kZGap - zlen;
        zlen += ahead_got;
      } else {  // (more left over than the gap takes: the read-ahead moves behind it)
        room(ahead_got);
        memcpy(zb[zcur].data() + zoff + zlen, o.data() + kZGap, ahead_got);
        zlen += ahead_got;
      }
      if (ahead_eof) eof = true;
    }
    auto need = [&](size_t upto) {  // at least `upto` bytes in use, or the file has no more
      while (zlen < upto && !eof) {
        const size_t want = std::max(upto - zlen, (size_t)64 << 20);
        room(want);
        const size_t got = read_raw(zb[zcur].data() + zoff + zlen, want);
        zlen += got;
        if (got < want) eof = true;
      }
      return zlen >= upto;
    };
    while (pending + isum < target) {
      if (!need(zready + 18)) {
        if (zlen == zready) break;  // the end of the file, at a block's end
        die("Didn't reach the end of sequence file, which might be corrupted! (truncated BGZF block)");
      }
      const unsigned char *h = zdata() + zready;
      if (h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C')
        die("Didn't reach the end of sequence file, which might be corrupted! (not a BGZF block)");
      const size_t bsize = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;
      if (bsize < 26) die("Didn't reach the end of sequence file, which might be corrupted! (BGZF block size)");
      if (!need(zready + bsize)) die("Didn't reach the end of sequence file, which might be corrupted! (truncated BGZF block)");
      const unsigned char *t = zdata() + zready + bsize - 4;
      isum += (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
      zready += bsize;
    }
    pending += isum;
    // (`eof` for the caller: nothing left to hand over after these blocks)
    if (!eof && zlen == zready) {
      const int ch = fgetc(raw);
      if (ch == EOF) eof = true; else ungetc(ch, raw);
    }
    if (!eof) {  // as many bytes again, read while the device works on these -- into the other buffer, with room behind for one more synchronous read
      const size_t want = std::max(zready, (size_t)64 << 20);
      RawBuf &o = zb[1 - zcur];
      o.reserve(kZGap + want + ((size_t)66 << 20), 0);
      unsigned char *dst = o.data() + kZGap;
      ahead_on = true; ahead_got = 0; ahead_eof = false;
      ahead = std::thread([this, dst, want]() { ahead_got = read_raw(dst, want); if (ahead_got < want) ahead_eof = true; });
    }
  }
  bool dev_final() const { return !ahead_on && eof && zlen == zready; }
  struct Block { size_t coff, csize, isize, ooff; };
  void fill_bgzf(size_t target) {
    while (len < target && !eof) {
      std::vector<Block> blocks;
      size_t csum = 0, isum = 0;
      cbuf.clear();
      while (isum < target - len && blocks.size() < 16384) {
        unsigned char h[18];
        const size_t got = fread(h, 1, 18, raw);
        if (got == 0) { eof = true; break; }
        if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[12] != 'B' || h[13] != 'C')
          die("Didn't reach the end of sequence file, which might be corrupted! (not a BGZF block)");
        const size_t bsize = ((size_t)h[16] | ((size_t)h[17] << 8)) + 1;  // whole block
        if (bsize < 26) die("Didn't reach the end of sequence file, which might be corrupted! (BGZF block size)");
        cbuf.resize(csum + bsize);
        memcpy(cbuf.data() + csum, h, 18);
        if (fread(cbuf.data() + csum + 18, 1, bsize - 18, raw) != bsize - 18)
          die("Didn't reach the end of sequence file, which might be corrupted! (truncated BGZF block)");
        const unsigned char *t = cbuf.data() + csum + bsize - 4;
        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
        blocks.push_back({csum, bsize, isize, isum});
        csum += bsize;
        isum += isize;
      }
      if (blocks.empty()) break;
      bufs[cur].reserve(len + isum, len);  // (off is 0 here)
      const int nt = (int)std::min<size_t>((size_t)team, blocks.size());
      std::vector<std::thread> th;
      std::vector<int> bad((size_t)nt, 0);
      for (int ti = 0; ti < nt; ++ti)
        th.emplace_back([&, ti]() {
          z_stream zs;
          memset(&zs, 0, sizeof(zs));
          if (inflateInit2(&zs, -15) != Z_OK) { bad[ti] = 1; return; }
          for (size_t bi = (size_t)ti; bi < blocks.size(); bi += (size_t)nt) {
            const Block &b = blocks[bi];
            if (b.isize == 0) continue;
            inflateReset(&zs);
            zs.next_in = cbuf.data() + b.coff + 18;
            zs.avail_in = (uInt)(b.csize - 26);
            zs.next_out = bufs[cur].data() + len + b.ooff;
            zs.avail_out = (uInt)b.isize;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { bad[ti] = 1; break; }
          }
          inflateEnd(&zs);
        });
      for (std::thread &t : th) t.join();
      for (int x : bad) if (x) die("Didn't reach the end of sequence file, which might be corrupted! (BGZF inflate)");
      len += isum;
    }
  }
  static constexpr size_t kGap = (size_t)64 << 20;  // room in front of the read-ahead for what the last batch left over
  void fill(size_t target) {
    if (bgzf && dev_inflate) { fill_bgzf_compressed(target); return; }
    if (bgzf) {
      if (off) { memmove(bufs[cur].data(), bufs[cur].data() + off, len); off = 0; }
      bufs[cur].reserve(target, len);
      fill_bgzf(target);
      return;
    }
    if (ahead_on) {
      // what was read ahead sits in the other buffer behind a gap: the bytes still in use go in front of it
      join_ahead();
      RawBuf &o = bufs[1 - cur];
      if (len <= kGap) {
        memcpy(o.data() + kGap - len, bufs[cur].data() + off, len);
        off = kGap - len;
      } else {  // (more left over than the gap takes: the read-ahead moves back)
        o.reserve(len + ahead_got + target, kGap + ahead_got);
        memmove(o.data() + len, o.data() + kGap, ahead_got);
        memcpy(o.data(), bufs[cur].data() + off, len);
        off = 0;
      }
      cur = 1 - cur;
      len += ahead_got;
      if (ahead_eof) eof = true;
    }
    bufs[cur].reserve(off + std::max(len, target), off + len);
    if (len < target && !eof) len += read_some(bufs[cur].data() + off + len, target - len, &eof);
    if (!eof) {
      RawBuf &o = bufs[1 - cur];
      o.reserve(kGap + target, 0);
      unsigned char *dst = o.data() + kGap;
      ahead_on = true; ahead_got = 0; ahead_eof = false;
      ahead = std::thread([this, dst, target]() { ahead_got = read_some(dst, target, &ahead_eof); });
    }
  }
  void consume(size_t used) {
    if (bgzf && dev_inflate) { pending -= used < pending ? used : pending; return; }  // (the device keeps the rest)
    off += used;
    len -= used;
    fed += used;
  }
  bool only_whitespace() {
    if (bgzf && dev_inflate) return true;  // (what is left on the device at the end holds no record: cmgpu_fastq_scan_bgzf counted none)
    if (ahead_on) fill(1);  // (takes over what was read ahead; nothing more is read: the file has ended)
    for (size_t i = 0; i < len; ++i) { const char ch = text()[i]; if (ch != '\n' && ch != '\r' && ch != ' ' && ch != '\t') return false; }
    return true;
  }
  void close() { if (ahead_on) { ahead.join(); ahead_on = false; } if (f) gzclose(f); f = nullptr; if (raw) fclose(raw); raw = nullptr; if (pargz) { pg.close(); pargz = false; } }
};

// --read-format (Chromap::ParseReadFormat chromap.cc:825-866, SequenceEffectiveRange): per stream up to four
// [start, end] ranges and a strand
struct ReadFormat {
  std::vector<int32_t> starts, ends;
  char strand = '+';
  bool identity() const { return starts.empty() || (strand == '+' && starts[0] == 0 && ends[0] == -1); }
  uint32_t eff_len(uint32_t len) const {
    if (identity()) return len;
    uint32_t out = 0;
    for (size_t k = 0; k < starts.size(); ++k) {
      int st = starts[k], en = ends[k] == -1 ? (int)len - 1 : ends[k];
      if (en >= (int)len) en = (int)len - 1;
      if (st < 0) st = 0;
      if (en >= st) out += (uint32_t)(en - st + 1);
    }
    return out;
  }
  // SequenceEffectiveRange::Replace for the host parser (pairs / SAM output)
  void apply(std::string &seq, std::string &qual) const {
    if (identity()) return;
    std::string ns, nq;
    for (size_t k = 0; k < starts.size(); ++k) {
      int st = starts[k], en = ends[k] == -1 ? (int)seq.size() - 1 : ends[k];
      if (en >= (int)seq.size()) en = (int)seq.size() - 1;
      if (st < 0) st = 0;
      for (int p = st; p <= en; ++p) { ns.push_back(seq[p]); if ((size_t)p < qual.size()) nq.push_back(qual[p]); }
    }
    if (strand == '-') {
      for (char &c : ns) { const char u = c & 0xDF; c = u == 'A' ? 'T' : u == 'C' ? 'G' : u == 'G' ? 'C' : u == 'T' ? 'A' : 'N'; }
      std::reverse(ns.begin(), ns.end());
      std::reverse(nq.begin(), nq.end());
    }
    seq.swap(ns);
    qual.swap(nq);
  }
};

struct Args {
  std::string index_path, ref_path, out_path, preset, whitelist, chr_order_path, pairs_order_path, translate_path;
  std::vector<std::string> r1, r2, bc;
  cmgpu_params p;
  bool build_index = false, out_bed = true, out_pairs = false, cell_level_dedup = false, host_ingest = false, out_sam = false, out_tagalign = false, skip_bc_check = false;
  size_t chunk_bytes = 256u << 20;
  bool chunk_given = false;  // (--ingest-chunk-mb: also the size of the pieces of block-compressed input handed to the device)
  ReadFormat fmt[3];  // read 1, read 2, barcode
  int k = 17, w = 7, device = 0, gpus = 1;
  bool force_exchange = false;
  uint32_t batch_pairs = 4000000;  // multiple of the reference's 500000-pair read batch
};

static std::vector<std::string> split_commas(const std::string &s) {
  std::vector<std::string> v;
  size_t a = 0;
  while (a <= s.size()) {
    size_t b = s.find(',', a);
    if (b == std::string::npos) b = s.size();
    if (b > a) v.push_back(s.substr(a, b - a));
    a = b + 1;
  }
  return v;
}

static Args parse(int argc, char **argv) {
  Args a;
  cmgpu_default_params(&a.p);
  // presets first, explicit flags override (chromap_driver.cc:247-275)
  for (int i = 1; i + 1 < argc; ++i)
    if (!strcmp(argv[i], "--preset")) {
      a.preset = argv[i + 1];
      if (cmgpu_apply_preset(&a.p, argv[i + 1]) != 0) die(std::string("Unrecognized preset parameters ") + argv[i + 1] + "\n");
      if (a.preset == "hic") { a.out_pairs = true; a.out_bed = false; }
      if (a.preset == "atac") a.cell_level_dedup = true;
    }
  for (int i = 1; i + 1 < argc; ++i)
    if (!strcmp(argv[i], "--min-frag-length")) {  // chromap_driver.cc:277-289; -k / -w override it
      const int mfl = atoi(argv[i + 1]);
      if (mfl <= 60) { a.k = 17; a.w = 7; } else if (mfl <= 80) { a.k = 19; a.w = 10; } else { a.k = 23; a.w = 11; }
    }
  for (int i = 1; i < argc; ++i) {
    const std::string o = argv[i];
    auto need = [&](const char *what) -> const char * { if (i + 1 >= argc) die(std::string("missing value for ") + what); return argv[++i]; };
    if (o == "--preset" || o == "--min-frag-length") { ++i; }
    else if (o == "-i" || o == "--build-index") a.build_index = true;
    else if (o == "-x" || o == "--index") a.index_path = need("-x");
    else if (o == "-r" || o == "--ref") a.ref_path = need("-r");
    else if (o == "-o" || o == "--output") a.out_path = need("-o");
    else if (o == "-1" || o == "--read1") a.r1 = split_commas(need("-1"));
    else if (o == "-2" || o == "--read2") a.r2 = split_commas(need("-2"));
    else if (o == "-b" || o == "--barcode") a.bc = split_commas(need("-b"));
    else if (o == "--barcode-whitelist") a.whitelist = need("--barcode-whitelist");
    else if (o == "-k" || o == "--kmer") a.k = atoi(need("-k"));
    else if (o == "-w" || o == "--window") a.w = atoi(need("-w"));
    else if (o == "-e" || o == "--error-threshold") a.p.error_threshold = atoi(need("-e"));
    else if (o == "-s" || o == "--min-num-seeds") a.p.min_num_seeds = atoi(need("-s"));
    else if (o == "-f" || o == "--max-seed-frequencies") {
      auto v = split_commas(need("-f"));
      if (v.size() != 2) die("Positive integers are required for max seed frequencies!");
      a.p.max_seed_frequency0 = atoi(v[0].c_str());
      a.p.max_seed_frequency1 = atoi(v[1].c_str());
    }
    else if (o == "-l" || o == "--max-insert-size") a.p.max_insert_size = atoi(need("-l"));
    else if (o == "-q" || o == "--MAPQ-threshold") a.p.mapq_threshold = atoi(need("-q"));
    else if (o == "-n" || o == "--max-num-best-mappings") a.p.max_num_best_mappings = atoi(need("-n"));
    else if (o == "--min-read-length") a.p.min_read_length = atoi(need("--min-read-length"));
    else if (o == "--drop-repetitive-reads") a.p.drop_repetitive_reads = atoi(need("--drop-repetitive-reads"));
    else if (o == "--bc-error-threshold") a.p.bc_error_threshold = atoi(need("--bc-error-threshold"));
    else if (o == "--bc-probability-threshold") a.p.bc_probability_threshold = atof(need("--bc-probability-threshold"));
    else if (o == "--output-mappings-not-in-whitelist") a.p.output_mappings_not_in_whitelist = 1;
    else if (o == "--trim-adapters") a.p.trim_adapters = 1;
    else if (o == "--remove-pcr-duplicates") a.p.remove_pcr_duplicates = 1;
    else if (o == "--remove-pcr-duplicates-at-cell-level") a.cell_level_dedup = true;
    else if (o == "--remove-pcr-duplicates-at-bulk-level") a.cell_level_dedup = false;
    else if (o == "--Tn5-shift") a.p.tn5_shift = 1;
    else if (o == "--split-alignment") a.p.split_alignment = 1;
    else if (o == "--low-mem") a.p.low_memory_mode = 1;
    else if (o == "--BED") { a.out_bed = true; a.out_pairs = false; a.out_sam = false; }
    else if (o == "--SAM") { a.out_sam = true; a.out_bed = false; a.out_pairs = false; }
    else if (o == "--TagAlign") { a.out_tagalign = true; a.out_bed = true; a.out_sam = false; a.out_pairs = false; }
    else if (o == "--pairs") { a.out_pairs = true; a.out_bed = false; }
    else if (o == "-t" || o == "--num-threads") need("-t");  // host threads are irrelevant here
    else if (o == "--skip-barcode-check") a.skip_bc_check = true;
    else if (o == "--barcode-translate") a.translate_path = need("--barcode-translate");
    else if (o == "--read-format") {
      const std::string f = need("--read-format");
      size_t i = 0;
      while (i < f.size()) {
        size_t j = f.find(',', i);
        if (j == std::string::npos) j = f.size();
        const std::string tok = f.substr(i, j - i);
        int st = tok.compare(0, 2, "r1") == 0 ? 0 : tok.compare(0, 2, "r2") == 0 ? 1 : tok.compare(0, 2, "bc") == 0 ? 2 : -1;
        if (st < 0 || tok.size() < 4 || tok[2] != ':') die("Unknown read format: " + f + "\n");
        std::vector<std::string> fld;
        size_t p = 3;
        while (p <= tok.size()) { size_t q = tok.find(':', p); if (q == std::string::npos) q = tok.size(); fld.push_back(tok.substr(p, q - p)); p = q + 1; }
        if (fld.size() < 2 || fld.size() > 3) die("Unknown read format: " + f + "\n");
        a.fmt[st].starts.push_back(atoi(fld[0].c_str()));
        a.fmt[st].ends.push_back(atoi(fld[1].c_str()));
        if (fld.size() == 3 && !fld[2].empty()) a.fmt[st].strand = fld[2][0];
        if (a.fmt[st].starts.size() > 4) die("at most four ranges per stream in --read-format");
        i = j + 1;
      }
    }
    else if (o == "--chr-order") a.chr_order_path = need("--chr-order");
    else if (o == "--pairs-natural-chr-order") a.pairs_order_path = need("--pairs-natural-chr-order");
    else if (o == "--device") a.device = atoi(need("--device"));
    else if (o == "--gpus") a.gpus = atoi(need("--gpus"));           // one context + host thread per GPU, records exchanged to chromosome owners
    else if (o == "--force-exchange") a.force_exchange = true;      // the multi-GPU code path with one GPU
    else if (o == "--host-ingest") a.host_ingest = true;   // kseq-style host parser (FASTA / multi-line records)
    else if (o == "--ingest-chunk-mb") { a.chunk_bytes = (size_t)atol(need("--ingest-chunk-mb")) << 20; a.chunk_given = true; }
    else if (o == "--batch-pairs") a.batch_pairs = (uint32_t)atol(need("--batch-pairs"));
    else if (o == "-v" || o == "--version") { printf("chromap-amd 0.1 (hot path of chromap 0.3.3-r521 on gfx950)\n"); exit(0); }
    else if (o == "-h" || o == "--help") {
      printf("Usage: chromap-amd -i -r ref.fa -o index | chromap-amd [--preset atac|chip|hic] -x index -r ref.fa -1 r1.fq[.gz] [-2 r2.fq[.gz]]\n"
             "       [-b barcode.fq --barcode-whitelist wl.txt] -o out [-e -s -f -l -q --min-read-length --trim-adapters\n"
             "       --remove-pcr-duplicates --Tn5-shift --low-mem --BED|--pairs --bc-error-threshold ...]\n");
      exit(0);
    }
    else die("unsupported option " + o + " (PAF and summary outputs are outside this build)");
  }
  if (a.p.max_num_best_mappings > a.p.drop_repetitive_reads) {  // chromap_driver.cc:630-641
    fprintf(stderr, "WARNING: you want to drop mapped reads with more than %d mappings. But you want to output top %d best mappings. "
                    "In this case, only reads with <=%d best mappings will be output.\n",
            a.p.drop_repetitive_reads, a.p.max_num_best_mappings, a.p.drop_repetitive_reads);
    a.p.max_num_best_mappings = a.p.drop_repetitive_reads;
  }
  if (a.p.max_num_best_mappings < 1) die("-n must be at least 1");
  // every pair of a batch has -n record slots in HBM (24 + 5 bytes each), addressed with 32 bits: large -n values map smaller batches (whole
  // reference batches of 500 000 pairs, so that the multi-mappers' sampling is the reference's); beyond 8192 one reference batch has more
  // than 2^32 slots
  if (a.p.max_num_best_mappings > 8192) die("-n above 8192 is outside this build (a 500000-pair batch then needs more than 2^32 record slots)");
  if (a.out_sam) {
    if (a.p.max_num_best_mappings > 1) die("--SAM with -n > 1 is outside this build");
    a.p.output_format = CMGPU_FORMAT_SAM;
  }
  // the reference accepts these combinations; this build has no record type for them -- refuse instead of writing garbage
  if (a.out_pairs && !a.p.split_alignment) a.p.output_format = CMGPU_FORMAT_PAIRS;  // MapPairedEndReads<PairsMapping> on the ordinary pairing (chromap_driver.cc:748-751)
  if (a.gpus < 1 || a.gpus > 64) die("--gpus must be 1..64");
  if (a.p.max_num_best_mappings > 64) {
    const uint64_t budget = 1ull << 30;  // record slots per batch (31 GB of HBM)
    const uint64_t fit = budget / (uint64_t)a.p.max_num_best_mappings;
    if (a.batch_pairs > fit) a.batch_pairs = (uint32_t)fit;
  }
  if (a.batch_pairs < 500000) a.batch_pairs = 500000;
  a.batch_pairs -= a.batch_pairs % 500000;
  return a;
}

int main(int argc, char **argv) {
  if (argc == 3 && !strcmp(argv[1], "--inflate-only")) {  // the ingest reader on its own (tests): inflated bytes of a file to stdout
    ChunkReader rd;
    if (!rd.open(argv[2])) die(std::string("Cannot find sequence file ") + argv[2]);
    rd.team = 4;
    for (;;) {
      rd.fill(3u << 20);
      if (rd.len == 0) break;
      const size_t take = rd.eof ? rd.len : rd.len - rd.len / 3;  // leave a tail, like the FASTQ parser does
      fwrite(rd.text(), 1, take, stdout);
      rd.consume(take);
    }
    if (rd.pargz) fprintf(stderr, "pargz chunks=%llu accepted=%llu serial=%llu\n", (unsigned long long)rd.pg.n_spec, (unsigned long long)rd.pg.n_accepted, (unsigned long long)rd.pg.n_serial);
    else fprintf(stderr, "%s\n", rd.bgzf ? "bgzf" : "gzread");
    rd.close();
    return 0;
  }
  Args a = parse(argc, argv);
  if (a.ref_path.empty() || a.out_path.empty()) die("No reference / output specified!");
  setenv("CM_FQ_EARLY", "1", 0);  // (the files' HIP streams are made with the context: on hardware queues of their own, cm_api.hip)
  // the HIP runtime and the library's device code come up on a thread of their own while this one reads the reference and the index
  // (an error, e.g. no device, is reported by cmgpu_create below)
  struct Warm {
    std::thread th;
    void join() { if (th.joinable()) th.join(); }
    ~Warm() { join(); }
  } warm;
  // ... and the output file is opened and emptied there too (the reference opens its output before it reads a read, chromap.h:277 / :777 -- the
  // MappingWriter is constructed ahead of the loop): a path that cannot be written fails now, not after the mapping, and emptying an existing
  // file of the last run's size -- tens of milliseconds per few hundred MB of page cache -- is not left for the moment the text is ready
  const bool text_out = !a.build_index && !a.out_pairs && !a.out_sam;
  bool out_opened = true;
  {
    const int dev0 = a.device, ndev = a.gpus;
    const std::string outp = a.out_path;
    warm.th = std::thread([dev0, ndev, text_out, outp, &out_opened]() {
      if (text_out) { FILE *of = fopen(outp.c_str(), "wb"); if (of) fclose(of); else out_opened = false; }
      for (int gi = 0; gi < ndev; ++gi) (void)cmgpu_warm_up(dev0 + gi);
    });
  }
  cmgpu_ref_view ref;
  if (cmgpu_load_reference_fasta(a.ref_path.c_str(), &ref) != 0) die("Cannot find sequence file " + a.ref_path);
  fprintf(stderr, "Loaded all sequences successfully, number of sequences: %u.\n", ref.n_sequences);
  if (a.build_index) {
    cmgpu_ctx *bctx = nullptr;
    warm.join();
    if (cmgpu_create_from_reference(&ref, a.k, a.w, &a.p, a.device, &bctx) != CMGPU_OK) die(cmgpu_last_error(nullptr));
    if (cmgpu_save_index_file(bctx, a.out_path.c_str()) != CMGPU_OK) die(cmgpu_last_error(bctx));
    int32_t k, w; uint32_t nb, nocc; uint64_t nmm, nkeys;
    cmgpu_index_info(bctx, &k, &w, &nb, &nocc, &nmm, &nkeys);
    fprintf(stderr, "Collected %llu minimizers.\nLookup table size: %llu, # buckets: %u, occurrence table size: %u.\n",
            (unsigned long long)nmm, (unsigned long long)nkeys, nb, nocc);
    cmgpu_destroy(bctx);
    cmgpu_free_host_ref(&ref);
    return 0;
  }
  if (a.index_path.empty() || a.r1.empty()) die("No index / read files specified!");
  const bool paired = !a.r2.empty();
  if (a.out_pairs && !paired) die("No support for single-end HiC yet!");  // chromap_driver.cc:716-718
  if (paired && a.r1.size() != a.r2.size()) die("Numbers of read1 and read2 files don't match!");
  const bool barcoded = !a.bc.empty();
  if (barcoded && a.bc.size() != a.r1.size()) die("Numbers of read1 and barcode files don't match!");
  if (barcoded && a.whitelist.empty() && a.p.remove_pcr_duplicates && a.p.low_memory_mode && !a.cell_level_dedup)
    die("bulk-level duplicate removal ranks barcodes by whitelist abundance: give --barcode-whitelist or --remove-pcr-duplicates-at-cell-level");
  a.p.dedup_at_bulk_level = barcoded && !a.cell_level_dedup ? 1 : 0;  // remove_pcr_duplicates_at_bulk_level defaults to true (mapping_parameters.h:49)
  cmgpu_index_view idx;
  if (cmgpu_load_index_file(a.index_path.c_str(), &idx) != 0) die("Cannot read index " + a.index_path);
  fprintf(stderr, "Kmer size: %d, window size: %d.\n", idx.kmer_size, idx.window_size);
  // one context (index + reference resident, own streams) per GPU; ctx = the first one, which also runs every single-GPU path
  const bool exchange = a.gpus > 1 || a.force_exchange;
  if (exchange && (a.out_pairs || a.out_sam || a.host_ingest))
    die("--gpus > 1 needs BED / TagAlign output and device-side FASTQ ingest (pairs and SAM records are post-processed on the host)");
  std::vector<cmgpu_ctx *> ctxs((size_t)a.gpus, nullptr);
  warm.join();
  if (!out_opened) die("cannot write " + a.out_path);
  for (int gi = 0; gi < a.gpus; ++gi)
    if (cmgpu_create(&idx, &ref, &a.p, a.device + gi, &ctxs[gi]) != CMGPU_OK) die(cmgpu_last_error(nullptr));
  cmgpu_ctx *ctx = ctxs[0];
  cmgpu_free_host_index(&idx);
  // --chr-order (Chromap::GenerateCustomRidRanks, chromap.cc:867-913): ranks from the listed names, unlisted
  // sequences follow in reference order; names / lengths for the writers are permuted the same way
  std::vector<const char *> out_names(ref.names, ref.names + ref.n_sequences);
  std::vector<uint32_t> out_lengths(ref.lengths, ref.lengths + ref.n_sequences);
  // ranks of `names` under an order file (Chromap::GenerateCustomRidRanks)
  auto ranks_from_file = [&](const std::string &path, const std::vector<const char *> &names) {
    FILE *of = fopen(path.c_str(), "r");
    if (!of) die("Cannot open chromosome order file " + path);
    std::vector<std::string> order;
    char lb[4096];
    while (fgets(lb, sizeof(lb), of)) { size_t l = strlen(lb); while (l && (lb[l - 1] == '\n' || lb[l - 1] == '\r')) lb[--l] = 0; order.push_back(lb); }
    fclose(of);
    std::vector<uint32_t> rank(names.size(), 0xffffffffu);
    // later lines win for a repeated name, like the reference's map assignment
    for (size_t i = 0; i < names.size(); ++i)
      for (size_t j = 0; j < order.size(); ++j) if (order[j] == names[i]) rank[i] = (uint32_t)j;
    std::vector<std::string> uniq(order);
    std::sort(uniq.begin(), uniq.end());
    uint32_t k = (uint32_t)(std::unique(uniq.begin(), uniq.end()) - uniq.begin());  // distinct names = first free rank
    for (size_t i = 0; i < names.size(); ++i) if (rank[i] == 0xffffffffu) rank[i] = k++;
    if (k > names.size()) die("ERROR: unknown chromsome names found in chromosome order file.");
    return rank;
  };
  if (!a.chr_order_path.empty()) {
    const std::vector<uint32_t> rank = ranks_from_file(a.chr_order_path, out_names);
    for (cmgpu_ctx *cx : ctxs) if (cmgpu_set_chr_order(cx, rank.data(), ref.n_sequences) != CMGPU_OK) die(cmgpu_last_error(cx));
    for (uint32_t i = 0; i < ref.n_sequences; ++i) { out_names[rank[i]] = ref.names[i]; out_lengths[rank[i]] = ref.lengths[i]; }
  }
  std::vector<uint32_t> pairs_rank;  // over the (possibly reordered) sequences, like the reference computes it
  if (!a.pairs_order_path.empty() && a.out_pairs) {
    pairs_rank = ranks_from_file(a.pairs_order_path, out_names);
    if (cmgpu_set_pairs_chr_order(ctx, pairs_rank.data(), ref.n_sequences) != CMGPU_OK) die(cmgpu_last_error(ctx));
  }

  cmgpu_stats st;
  memset(&st, 0, sizeof(st));
  std::vector<std::string> read_names;  // pairs output needs read-1 names by read_id
  uint64_t num_reads = 0;
  uint32_t next_read_id = 0, bc_len = 0;

  double t_read = 0, t_parse = 0, t_map = 0, t_post = 0;
  const double t_begin = now_s();
  for (cmgpu_ctx *cx : ctxs) {
    if (a.skip_bc_check) cmgpu_set_barcode_check(cx, 0);
    for (int m = 0; m < 3; ++m)
      if (!a.fmt[m].identity() &&
          cmgpu_fastq_set_format(cx, m, (int)a.fmt[m].starts.size(), a.fmt[m].starts.data(), a.fmt[m].ends.data(), a.fmt[m].strand) != CMGPU_OK)
        die("bad --read-format");
  }
  if (exchange) {  // records travel to the contexts that own their chromosomes (RCCL over xGMI, issued by the library)
    if (cmgpu_exchange_init_all(ctxs.data(), a.gpus) != CMGPU_OK) die(cmgpu_last_error(ctxs[0]));
  }
  const bool device_ingest = !a.out_pairs && !a.out_sam && !a.host_ingest;  // pairs / SAM output need read names (and qualities): host parser
  // --SAM: everything the final sort needs, over all batches
  std::vector<cmgpu_sam_record> sam_rec;
  std::vector<uint32_t> sam_cigar;
  std::vector<std::vector<char>> sam_md_batches;
  std::vector<uint32_t> sam_md_caps;
  std::vector<uint64_t> sam_batch_slots, sam_bc;
  std::vector<std::string> sam_names1, sam_names2;
  std::vector<char> sam_b1, sam_q1, sam_b2, sam_q2;
  std::vector<uint32_t> sam_o1(1, 0), sam_o2(1, 0);
  auto ck = [&](int rc) { if (rc != CMGPU_OK) die(cmgpu_last_error(ctx)); };
  if (barcoded && a.whitelist.empty()) {  // no whitelist: every barcode is kept as read (chromap.h:897-903); only its length is needed
    FastxReader pk;
    if (!pk.open(a.bc[0])) die("Cannot find sequence file " + a.bc[0]);
    std::string nm, sq, ql;
    if (!pk.record(nm, sq, ql)) die("empty barcode file");
    pk.close();
    a.fmt[2].apply(sq, ql);
    bc_len = (uint32_t)sq.size();
  }
  if (device_ingest) {
    // ---- FASTQ text goes to the GPU in chunks; lines, records and the SoA batch are built there
    if (barcoded && !a.whitelist.empty()) {
      // whitelist + abundance pre-pass (chromap.h:750-761), barcode file streamed through the device
      int done = 0;
      uint64_t ns = 0;
      uint32_t nk = 0;
      for (size_t bi = 0; bi < a.bc.size() && !done; ++bi) {  // every barcode file in turn, batches restart per file (chromap.cc:495-543)
      ChunkReader br;
      if (!br.open(a.bc[bi])) die("Cannot find sequence file " + a.bc[bi]);
      size_t target = a.chunk_bytes;
      while (!done) {
        br.fill(target);
        if (br.len == 0) break;
        if (bc_len == 0) {  // length of the first barcode: second line of the file
          const char *p = (const char *)memchr(br.text(), '\n', br.len);
          const char *q = p ? (const char *)memchr(p + 1, '\n', br.len - (size_t)(p + 1 - br.text())) : nullptr;
          if (!p || !q) die("barcode file is not FASTQ");
          bc_len = (uint32_t)(q - p - 1);
          if (bc_len && p[bc_len] == '\r') --bc_len;
          bc_len = a.fmt[2].eff_len(bc_len);
          uint64_t *keys = nullptr;
          if (cmgpu_load_whitelist_file(a.whitelist.c_str(), bc_len, &keys, &nk) != 0) die("ERROR: whitelist and input barcode lengths are not equal!");
          ck(cmgpu_set_whitelist(ctx, keys, nk, bc_len));
          free(keys);
        }
        uint32_t cnt = 0;
        ck(cmgpu_fastq_scan(ctx, 2, br.text(), br.len, br.eof, &cnt));
        uint32_t n = cnt;
        if (!br.eof) n -= n % 500000;
        if (n == 0 && !br.eof) { target *= 2; continue; }
        uint64_t used = 0;
        ck(cmgpu_fastq_take(ctx, 2, n, &used));
        br.consume((size_t)used);
        ck(cmgpu_barcode_abundance_resident(ctx, &ns, &done));
        if (br.eof && n == cnt) break;
      }
      br.close();
      }
      fprintf(stderr, "Loaded %u barcodes.\nCompute barcode abundance using %llu.\n", nk, (unsigned long long)ns);
      for (size_t gi = 1; gi < ctxs.size(); ++gi) ck(cmgpu_copy_whitelist(ctxs[gi], ctx));
    }
    // Batches are dealt to the contexts in turn; a context maps its batch on its own host thread while the next
    // batch is read and parsed for the next context.  With more than one context a round ends with the record
    // exchange (collective: every context takes part, with an empty batch when the input ran out).
    const size_t NG = ctxs.size();
    std::vector<std::thread> workers(NG);
    std::vector<char> busy(NG, 0);
    std::vector<int> wrc(NG, CMGPU_OK);
    std::vector<cmgpu_stats> wst(NG);
    for (cmgpu_stats &x : wst) memset(&x, 0, sizeof(x));
    size_t turn = 0;
    bool store_sized = false;
    const bool overlap1 = NG == 1 && !exchange && !getenv("CM_CLI_NO_OVERLAP");  // (the variable: the serial order, for measurements)
    auto finish_round = [&]() {
      for (size_t gi = 0; gi < NG; ++gi) if (busy[gi]) { workers[gi].join(); busy[gi] = 0; }
      for (size_t gi = 0; gi < NG; ++gi) if (wrc[gi] != CMGPU_OK) die(cmgpu_last_error(ctxs[gi]));
      if (exchange) {
        for (size_t gi = 0; gi < NG; ++gi) workers[gi] = std::thread([&, gi]() { wrc[gi] = cmgpu_exchange_step(ctxs[gi], nullptr, nullptr); });
        for (size_t gi = 0; gi < NG; ++gi) workers[gi].join();
        for (size_t gi = 0; gi < NG; ++gi) if (wrc[gi] != CMGPU_OK) die(cmgpu_last_error(ctxs[gi]));
      }
      turn = 0;
    };
    for (size_t fi = 0; fi < a.r1.size(); ++fi) {
      ChunkReader rd[3];
      const int ns_streams = 1 + (paired ? 1 : 0) + (barcoded ? 1 : 0);
      {
        const int team = (int)std::max(2u, std::min(32u, cpu_budget() / (unsigned)ns_streams));
        for (ChunkReader &x : rd) { x.team = team; x.files_side_by_side = ns_streams; x.dev_inflate = NG == 1; }  // (several GPUs take turns: the text cannot stay on one)
      }
      int sid[3] = {0, paired ? 1 : 2, 2};
      if (!rd[0].open(a.r1[fi])) die("Cannot find sequence file " + a.r1[fi]);
      if (paired && !rd[1].open(a.r2[fi])) die("Cannot find sequence file " + a.r2[fi]);
      if (barcoded && !rd[ns_streams - 1].open(a.bc[fi])) die("Cannot find sequence file " + a.bc[fi]);
      size_t target = a.chunk_bytes;
      for (;;) {
        uint32_t cnt[3] = {0, 0, 0};
        bool all_final = true;
        double t0 = now_s();
        {  // one reader thread per file: gzip inflation of read 1 / read 2 / barcodes runs side by side
          std::thread th[3];
          // (blocks inflated on the device: a scan per batch, not per chunk -- the first pass of the inflate takes the same time for
          //  a few hundred blocks as for tens of thousands)
          // (... sized to the batch: what a take leaves over is copied and scanned again with the next piece, so a piece far larger than
          //  a batch -- 1 GiB against the ~125 MB of a 500 000-pair batch -- would be re-scanned many times)
          const size_t piece = std::min<size_t>((size_t)1 << 30, std::max<size_t>((size_t)64 << 20, (size_t)a.batch_pairs * 256));
          // (a file's FIRST piece is half that: the device starts on it while the rest of a small file -- or the next piece of a large
          //  one -- is still being read; 8 M pairs in two 185 MB files: 30 ms of reading in front of everything else became 17)
          static const size_t first_div = getenv("CM_FIRST_PIECE_DIV") ? (size_t)std::max(1, atoi(getenv("CM_FIRST_PIECE_DIV"))) : 2;
          auto want = [&](int m) {
            if (!(rd[m].bgzf && rd[m].dev_inflate && !a.chunk_given && target < piece)) return target;
            return rd[m].fed == 0 && rd[m].zready == 0 ? std::max(target, piece / first_div) : piece;
          };
          for (int m = 1; m < ns_streams; ++m) th[m] = std::thread([&rd, m, &want]() { rd[m].fill(want(m)); });
          rd[0].fill(want(0));
          for (int m = 1; m < ns_streams; ++m) th[m].join();
        }
        t_read += now_s() - t0;
        t0 = now_s();
        cmgpu_ctx *cx = ctxs[turn];
        auto ckx = [&](int rc) { if (rc != CMGPU_OK) die(cmgpu_last_error(cx)); };
        const bool dbg_times = getenv("CM_CLI_TIMES") != nullptr;
        const double ts0 = now_s();
        // the files' scans (upload, inflate, line index, record checks) run side by side: a host thread and a HIP stream per file
        int src[3] = {CMGPU_OK, CMGPU_OK, CMGPU_OK};
        std::string serr[3];  // a failed scan's own message (two files may fail differently at the same time)
        {
          std::thread th[3];
          auto scan = [&](int m) {
            const bool dev = rd[m].bgzf && rd[m].dev_inflate;
            const bool fin = dev ? rd[m].dev_final() : rd[m].eof;
            src[m] = dev ? cmgpu_fastq_scan_bgzf(cx, sid[m], rd[m].zdata(), rd[m].zready, fin, &cnt[m])
                         : cmgpu_fastq_scan(cx, sid[m], rd[m].text(), rd[m].len, rd[m].eof, &cnt[m]);
            if (src[m] != CMGPU_OK) serr[m] = cmgpu_last_error_thread();
          };
          for (int m = 1; m < ns_streams; ++m) th[m] = std::thread(scan, m);
          scan(0);
          for (int m = 1; m < ns_streams; ++m) th[m].join();
        }
        for (int m = 0; m < ns_streams; ++m) {
          const bool dev = rd[m].bgzf && rd[m].dev_inflate;
          const bool fin = dev ? rd[m].dev_final() : rd[m].eof;
          all_final = all_final && fin;
          const int rc = src[m];
          if (rc == CMGPU_EFORMAT && dev && strstr(serr[m].c_str(), "BGZF"))
            die(std::string("Didn't reach the end of sequence file, which might be corrupted! (") + serr[m] + ")");
          if (rc == CMGPU_EFORMAT) die(serr[m] + " -- rerun with --host-ingest");
          if (rc != CMGPU_OK) die(serr[m]);
        }
        const double ts1 = now_s();
        uint32_t n = cnt[0];
        for (int m = 1; m < ns_streams; ++m) n = cnt[m] < n ? cnt[m] : n;
        if (n > a.batch_pairs) n = a.batch_pairs;
        if (!all_final || n == a.batch_pairs) n -= n % 500000;  // whole reference batches except at the very end
        if (n == 0) {
          if (!all_final) { target *= 2; continue; }
          for (int m = 0; m < ns_streams; ++m)
            if (cnt[m] != 0) die("Numbers of reads and barcodes don't match!");
          // (text inflated on the device: what is left of it must be blank, as only_whitespace() checks for text held here)
          for (int m = 0; m < ns_streams; ++m)
            if (rd[m].bgzf && rd[m].dev_inflate) {
              uint64_t used = 0;
              if (cmgpu_fastq_take(cx, sid[m], 0, &used) != CMGPU_OK)
                die(std::string("Didn't reach the end of sequence file, which might be corrupted! (") + cmgpu_last_error(cx) + ")");
            }
          break;
        }
        for (int m = 0; m < ns_streams; ++m) {
          uint64_t used = 0;
          const int trc = cmgpu_fastq_take(cx, sid[m], n, &used);
          if (trc == CMGPU_EFORMAT && rd[m].bgzf && rd[m].dev_inflate)  // (text behind the file's last whole record)
            die(std::string("Didn't reach the end of sequence file, which might be corrupted! (") + cmgpu_last_error(cx) + ")");
          ckx(trc);
          rd[m].consume((size_t)used);
        }
        // (one context: its last batch was being mapped under this batch's read, scan and take -- the take gathers into staging buffers
        //  that the commit swaps in, cm_ingest.hip; with several contexts the round's join below does the same job)
        const double tj0 = now_s();
        if (overlap1 && busy[0]) {
          workers[0].join();
          busy[0] = 0;
          if (wrc[0] != CMGPU_OK) die(cmgpu_last_error(ctxs[0]));
        }
        const double tj1 = now_s();
        if (!store_sized && overlap1 && a.p.max_num_best_mappings == 1) {
          // the record store sized once from what the first piece says about the files (records scanned / share of the file they came
          // from, over all input files): grown on demand it doubles each time with an allocation, a copy and a synchronous free
          store_sized = true;
          const double fr = rd[0].fraction();
          if (fr > 0 && rd[0].fsize) {
            uint64_t all = 0;
            for (const std::string &pth : a.r1) { struct stat sb; if (stat(pth.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) all += (uint64_t)sb.st_size; }
            const double est = (double)cnt[0] / fr * ((double)all / (double)rd[0].fsize);
            if (est < 2.0e9) (void)cmgpu_store_reserve(cx, (uint64_t)(est * 1.01) + 1000000, barcoded ? 1 : 0);  // (no room: the store grows as before)
          }
        }
        ckx(cmgpu_fastq_commit(cx, n, next_read_id, paired ? 1 : 0, barcoded ? 1 : 0));
        t_parse += now_s() - t0 - (tj1 - tj0);
        t_map += tj1 - tj0;
        if (dbg_times) fprintf(stderr, "[times] scan %.4f take %.4f wait for the batch before %.4f\n", ts1 - ts0, tj0 - ts1, tj1 - tj0);
        t0 = now_s();
        {
          const size_t gi = turn;
          workers[gi] = std::thread([&, gi, cx]() {
            uint64_t k = 0;
            const double tm0 = now_s();
            int rc = cmgpu_map_resident(cx, &k, &wst[gi]);
            const double tm1 = now_s();
            if (rc == CMGPU_OK && !exchange) rc = cmgpu_store_append_resident(cx, nullptr);
            if (getenv("CM_CLI_TIMES")) fprintf(stderr, "[times] map %.4f store_append %.4f\n", tm1 - tm0, now_s() - tm1);
            wrc[gi] = rc;
          });
          busy[gi] = 1;
        }
        if (!overlap1 && ++turn == NG) finish_round();
        t_map += now_s() - t0;
        num_reads += paired ? 2ull * n : n;
        next_read_id += n;
        fprintf(stderr, "Mapped %u read%s.\n", n, paired ? " pairs" : "s");
      }
      for (int m = 0; m < ns_streams; ++m) {
        if (!rd[m].only_whitespace()) die("Didn't reach the end of sequence file, which might be corrupted!");
        rd[m].close();
      }
    }
    {
      const double t0 = now_s();
      if (turn > 0 || exchange || busy[0]) finish_round();  // the last, partial round (empty batches for the contexts beyond it)
      t_map += now_s() - t0;
    }
    for (const cmgpu_stats &x : wst) {
      st.num_candidates += x.num_candidates; st.num_mappings += x.num_mappings; st.num_mapped_reads += x.num_mapped_reads;
      st.num_uniquely_mapped_reads += x.num_uniquely_mapped_reads; st.num_barcode_in_whitelist += x.num_barcode_in_whitelist;
      st.num_corrected_barcode += x.num_corrected_barcode;
    }
  } else {
    // single-cell: whitelist + abundance pre-pass over the whole barcode file (chromap.h:750-761)
    if (barcoded && !a.whitelist.empty()) {
      uint32_t nk = 0;
      uint64_t ns = 0;
      for (size_t bi = 0; bi < a.bc.size() && ns < 20000000ull; ++bi) {  // batches restart per file (chromap.cc:495-543)
        FastxReader br;
        if (!br.open(a.bc[bi])) die("Cannot find sequence file " + a.bc[bi]);
        std::string nm, sq, ql;
        std::vector<char> bb;
        std::vector<uint32_t> bo(1, 0);
        while (br.record(nm, sq, ql)) { a.fmt[2].apply(sq, ql); bb.insert(bb.end(), sq.begin(), sq.end()); bo.push_back((uint32_t)bb.size()); }
        br.close();
        if (bo.size() < 2) { if (bi == 0) die("empty barcode file"); continue; }
        if (bi == 0) {
          bc_len = bo[1] - bo[0];
          uint64_t *keys = nullptr;
          if (cmgpu_load_whitelist_file(a.whitelist.c_str(), bc_len, &keys, &nk) != 0) die("ERROR: whitelist and input barcode lengths are not equal!");
          if (cmgpu_set_whitelist(ctx, keys, nk, bc_len) != 0) die(cmgpu_last_error(ctx));
          free(keys);
        }
        if (cmgpu_compute_barcode_abundance(ctx, bb.data(), bo.data(), (uint32_t)bo.size() - 1, &ns) != 0) die(cmgpu_last_error(ctx));
      }
      fprintf(stderr, "Loaded %u barcodes.\nCompute barcode abundance using %llu.\n", nk, (unsigned long long)ns);
    }

    for (size_t fi = 0; fi < a.r1.size(); ++fi) {
      FastxReader f1, f2, fb;
      if (!f1.open(a.r1[fi])) die("Cannot find sequence file " + a.r1[fi]);
      if (paired && !f2.open(a.r2[fi])) die("Cannot find sequence file " + a.r2[fi]);
      if (barcoded && !fb.open(a.bc[fi])) die("Cannot find sequence file " + a.bc[fi]);
      bool more = true;
      while (more) {
        std::vector<char> b1, b2, bb, bq;
        std::vector<uint32_t> o1(1, 0), o2(1, 0), bo(1, 0);
        std::string n1, s1, q1, n2, s2, q2, nb, sb, qb;
        uint32_t n = 0;
        while (n < a.batch_pairs) {
          const bool g1 = f1.record(n1, s1, q1);
          const bool g2 = paired ? f2.record(n2, s2, q2) : g1;
          const bool gb = barcoded ? fb.record(nb, sb, qb) : g1;
          if (!g1 && !g2 && !gb) { more = false; break; }
          if (!(g1 && g2 && gb)) die("Numbers of reads and barcodes don't match!");
          a.fmt[0].apply(s1, q1);
          if (paired) a.fmt[1].apply(s2, q2);
          if (barcoded) a.fmt[2].apply(sb, qb);
          b1.insert(b1.end(), s1.begin(), s1.end()); o1.push_back((uint32_t)b1.size());
          if (paired) { b2.insert(b2.end(), s2.begin(), s2.end()); o2.push_back((uint32_t)b2.size()); }
          if (barcoded) { bb.insert(bb.end(), sb.begin(), sb.end()); bq.insert(bq.end(), qb.begin(), qb.end()); bq.resize(bb.size(), 'I'); bo.push_back((uint32_t)bb.size()); }
          if (a.out_pairs) read_names.push_back(n1);
          if (a.out_sam) {
            q1.resize(s1.size(), 'I');
            sam_names1.push_back(n1); sam_b1.insert(sam_b1.end(), s1.begin(), s1.end()); sam_q1.insert(sam_q1.end(), q1.begin(), q1.end());
            if (sam_b1.size() > 0xfffffff0ull) die("--SAM holds all reads of a run in host memory with 32-bit offsets: input too large");
            sam_o1.push_back((uint32_t)sam_b1.size());
            if (paired) {
              q2.resize(s2.size(), 'I');
              sam_names2.push_back(n2); sam_b2.insert(sam_b2.end(), s2.begin(), s2.end()); sam_q2.insert(sam_q2.end(), q2.begin(), q2.end());
              if (sam_b2.size() > 0xfffffff0ull) die("--SAM holds all reads of a run in host memory with 32-bit offsets: input too large");
              sam_o2.push_back((uint32_t)sam_b2.size());
            }
          }
          ++n;
        }
        if (n == 0) break;
        num_reads += paired ? 2ull * n : n;
        uint64_t k = 0;
        int rc;
        if (barcoded && !paired) {
          cmgpu_single_batch bt{n, next_read_id, b1.data(), o1.data()};
          cmgpu_barcode_batch bc{bb.data(), bq.data(), bo.data()};
          rc = cmgpu_map_single_barcoded(ctx, &bt, &bc, nullptr, 0, &k, &st);
        } else if (barcoded) {
          cmgpu_batch bt{n, next_read_id, b1.data(), o1.data(), b2.data(), o2.data()};
          cmgpu_barcode_batch bc{bb.data(), bq.data(), bo.data()};
          rc = cmgpu_map_pairs_barcoded(ctx, &bt, &bc, nullptr, 0, &k, &st);
        } else if (paired && !a.out_pairs) {
          cmgpu_batch bt{n, next_read_id, b1.data(), o1.data(), b2.data(), o2.data()};
          rc = cmgpu_map_pairs(ctx, &bt, nullptr, 0, &k, &st);
        } else if (paired) {  // pairs records: they stay in HBM too (sorted and rendered by cmgpu_store_format_pairs)
          cmgpu_batch bt{n, next_read_id, b1.data(), o1.data(), b2.data(), o2.data()};
          rc = cmgpu_map_pairs(ctx, &bt, nullptr, 0, &k, &st);
        } else {
          cmgpu_single_batch bt{n, next_read_id, b1.data(), o1.data()};
          rc = cmgpu_map_single(ctx, &bt, nullptr, 0, &k, &st);
        }
        if (rc != CMGPU_OK) die(cmgpu_last_error(ctx));
        if (a.out_sam) {  // alignment records + CIGAR / MD pools of this batch
          uint64_t slots = 0;
          uint32_t cap = 0;
          cmgpu_sam_layout(ctx, &slots, &cap);
          const size_t base = sam_rec.size();
          sam_rec.resize(base + slots);
          sam_cigar.resize((base + slots) * CMGPU_SAM_CIGAR_CAP);
          sam_md_batches.emplace_back((size_t)slots * cap + 1);
          sam_md_caps.push_back(cap);
          sam_batch_slots.push_back(slots);
          if (cmgpu_download_sam(ctx, sam_rec.data() + base, sam_cigar.data() + base * CMGPU_SAM_CIGAR_CAP, sam_md_batches.back().data()) != CMGPU_OK)
            die(cmgpu_last_error(ctx));
          if (barcoded) {  // CB tag + the barcode's place in the sort key
            const size_t kb = sam_bc.size();
            sam_bc.resize(kb + n);
            if (cmgpu_download_barcode_keys(ctx, sam_bc.data() + kb) != CMGPU_OK) die(cmgpu_last_error(ctx));
          }
        } else if (cmgpu_store_append_resident(ctx, nullptr) != CMGPU_OK) {
          // BED and pairs outputs: the records never leave HBM -- they join the device-side store
          die(cmgpu_last_error(ctx));
        }
        next_read_id += n;
        fprintf(stderr, "Mapped %u read%s.\n", n, paired ? " pairs" : "s");
      }
      f1.close(); f2.close(); fb.close();
    }
  }
  // Chromap::OutputMappingStatistics (chromap.cc:808-823)
  fprintf(stderr, "Number of reads: %llu.\nNumber of mapped reads: %llu.\nNumber of uniquely mapped reads: %llu.\n"
                  "Number of reads have multi-mappings: %llu.\nNumber of candidates: %llu.\nNumber of mappings: %llu.\n"
                  "Number of uni-mappings: %llu.\nNumber of multi-mappings: %llu.\n",
          (unsigned long long)num_reads, (unsigned long long)st.num_mapped_reads, (unsigned long long)st.num_uniquely_mapped_reads,
          (unsigned long long)(st.num_mapped_reads - st.num_uniquely_mapped_reads), (unsigned long long)st.num_candidates,
          (unsigned long long)st.num_mappings, (unsigned long long)st.num_uniquely_mapped_reads,
          (unsigned long long)(st.num_mappings - st.num_uniquely_mapped_reads));
  if (barcoded && !a.whitelist.empty())
    fprintf(stderr, "Number of barcodes in whitelist: %llu.\nNumber of corrected barcodes: %llu.\n",
            (unsigned long long)st.num_barcode_in_whitelist, (unsigned long long)st.num_corrected_barcode);
  int64_t lines;
  uint64_t nl = 0, nbytes = 0;
  if (a.out_sam) {
    uint32_t cap = 1;
    for (uint32_t c : sam_md_caps) cap = c > cap ? c : cap;
    std::vector<char> md(sam_rec.size() * (size_t)cap + 1);
    size_t slot0 = 0;
    for (size_t b = 0; b < sam_md_batches.size(); ++b) {
      for (uint64_t t = 0; t < sam_batch_slots[b]; ++t)
        memcpy(md.data() + (slot0 + t) * cap, sam_md_batches[b].data() + t * sam_md_caps[b], sam_md_caps[b]);
      slot0 += sam_batch_slots[b];
    }
    std::vector<const char *> n1(sam_names1.size()), n2(sam_names2.size() ? sam_names2.size() : 1, "");
    for (size_t i = 0; i < sam_names1.size(); ++i) n1[i] = sam_names1[i].c_str();
    for (size_t i = 0; i < sam_names2.size(); ++i) n2[i] = sam_names2[i].c_str();
    if (barcoded && !a.translate_path.empty()) {  // CB:Z: through the translation table (read here: the table may be gzip-compressed)
      std::string table;
      gzFile tf = gzopen(a.translate_path.c_str(), "r");
      if (!tf) die("Cannot open barcode translation file " + a.translate_path);
      char tb[1 << 16];
      for (int got; (got = gzread(tf, tb, sizeof(tb))) > 0;) table.append(tb, (size_t)got);
      gzclose(tf);
      lines = cmgpu_write_sam_barcoded_translated(out_names.data(), out_lengths.data(), ref.n_sequences, &a.p, sam_rec.data(), sam_rec.size(), paired ? 1 : 0,
                                                  sam_cigar.data(), md.data(), cap, n1.data(), n2.data(), sam_b1.data(), sam_q1.data(), sam_o1.data(),
                                                  paired ? sam_b2.data() : nullptr, paired ? sam_q2.data() : nullptr, paired ? sam_o2.data() : nullptr,
                                                  sam_bc.data(), bc_len, table.data(), table.size(), a.out_path.c_str());
      if (lines == CMGPU_EFORMAT) die("Barcode does not exist in the translation table.");
    } else if (barcoded)
      lines = cmgpu_write_sam_barcoded(out_names.data(), out_lengths.data(), ref.n_sequences, &a.p, sam_rec.data(), sam_rec.size(), paired ? 1 : 0,
                                       sam_cigar.data(), md.data(), cap, n1.data(), n2.data(), sam_b1.data(), sam_q1.data(), sam_o1.data(),
                                       paired ? sam_b2.data() : nullptr, paired ? sam_q2.data() : nullptr, paired ? sam_o2.data() : nullptr,
                                       sam_bc.data(), bc_len, a.out_path.c_str());
    else
    lines = cmgpu_write_sam(out_names.data(), out_lengths.data(), ref.n_sequences, &a.p, sam_rec.data(), sam_rec.size(), paired ? 1 : 0, sam_cigar.data(),
                            md.data(), cap, n1.data(), n2.data(), sam_b1.data(), sam_q1.data(), sam_o1.data(),
                            paired ? sam_b2.data() : nullptr, paired ? sam_q2.data() : nullptr, paired ? sam_o2.data() : nullptr,
                            a.out_path.c_str());
  } else if (a.out_pairs) {
    // sort + MAPQ filter + text on the device; the read names go up once as a blob
    std::string blob;
    std::vector<uint64_t> roff(read_names.size() + 1, 0);
    for (size_t i = 0; i < read_names.size(); ++i) { blob += read_names[i]; roff[i + 1] = blob.size(); }
    uint64_t nl = 0, nbytes = 0;
    if (cmgpu_store_format_pairs(ctx, out_names.data(), ref.n_sequences, &a.p, blob.data(), roff.data(), (uint32_t)read_names.size(), 0, &nl, &nbytes) != CMGPU_OK)
      die(cmgpu_last_error(ctx));
    if (cmgpu_write_pairs_header(out_names.data(), out_lengths.data(), ref.n_sequences, pairs_rank.empty() ? nullptr : pairs_rank.data(), a.out_path.c_str()) != CMGPU_OK)
      die("Cannot write " + a.out_path);
    if (cmgpu_store_write_text(ctx, a.out_path.c_str(), 1) != CMGPU_OK) die(cmgpu_last_error(ctx));
    lines = (long long)nl;
  } else {
    // sort + duplicate removal + MAPQ filter + Tn5 shift + text, all on the device
    const int kind = a.out_tagalign && paired ? (barcoded ? CMGPU_TEXT_TAGALIGN_PE_BC : CMGPU_TEXT_TAGALIGN_PE)
                     : a.out_tagalign && barcoded ? CMGPU_TEXT_TAGALIGN_SE_BC
                     : barcoded ? (paired ? CMGPU_TEXT_BED_PE_BC : CMGPU_TEXT_BED_SE_BC) : paired ? CMGPU_TEXT_BED_PE : CMGPU_TEXT_BED_SE;
    const double t0 = now_s();
    {
      // every context sorts, de-duplicates and renders the chromosomes it owns (all of them with one context)
      std::vector<std::thread> th(ctxs.size());
      std::vector<int> rcs(ctxs.size(), CMGPU_OK);
      std::vector<uint64_t> nls(ctxs.size(), 0), nbs(ctxs.size(), 0);
      for (size_t gi = 0; gi < ctxs.size(); ++gi)
        th[gi] = std::thread([&, gi]() { rcs[gi] = cmgpu_store_format(ctxs[gi], kind, out_names.data(), ref.n_sequences, &a.p, bc_len, &nls[gi], &nbs[gi]); });
      for (size_t gi = 0; gi < ctxs.size(); ++gi) th[gi].join();
      for (size_t gi = 0; gi < ctxs.size(); ++gi) {
        if (rcs[gi] != CMGPU_OK) die(cmgpu_last_error(ctxs[gi]));
        nl += nls[gi];
        nbytes += nbs[gi];
      }
    }
    const double t1 = now_s();
    if (ctxs.size() > 1 && barcoded && !a.translate_path.empty()) die("--barcode-translate with --gpus > 1 is outside this build");
    if (barcoded && !a.translate_path.empty() && (kind == CMGPU_TEXT_BED_PE_BC || kind == CMGPU_TEXT_BED_SE_BC)) {
      // --barcode-translate (BarcodeTranslator, barcode_translator.h:43-101): the device rendered the corrected barcodes;
      // column 4 is rewritten on the way to the file.  Table lines are "to<TAB or ,>from"; a barcode made of several
      // segments of the table's length is translated segment by segment and joined with '-'.
      std::unordered_map<std::string, std::string> table;
      size_t from_len = 0;
      gzFile tf = gzopen(a.translate_path.c_str(), "r");
      if (!tf) die("Cannot open barcode translation file " + a.translate_path);
      char lb[512];
      while (gzgets(tf, lb, sizeof(lb))) {
        size_t l = strlen(lb);
        if (l && lb[l - 1] == '\n') lb[--l] = 0;
        size_t i = 0;
        while (i < l && lb[i] != ',' && lb[i] != '\t') ++i;
        if (i >= l) continue;
        from_len = l - i - 1;
        table[std::string(lb + i + 1, from_len)] = std::string(lb, i);
      }
      gzclose(tf);
      std::vector<char> text(nbytes + 1);
      if (cmgpu_store_text(ctx, text.data(), nbytes) != CMGPU_OK) die(cmgpu_last_error(ctx));
      FILE *of = fopen(a.out_path.c_str(), "wb");
      if (!of) die("cannot write " + a.out_path);
      std::string outb;
      outb.reserve(1 << 20);
      const char *p = text.data(), *end = text.data() + nbytes;
      while (p < end) {
        const char *nlp = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!nlp) nlp = end;
        const char *c1 = (const char *)memchr(p, '\t', (size_t)(nlp - p));
        const char *c2 = c1 ? (const char *)memchr(c1 + 1, '\t', (size_t)(nlp - c1 - 1)) : nullptr;
        const char *c3 = c2 ? (const char *)memchr(c2 + 1, '\t', (size_t)(nlp - c2 - 1)) : nullptr;
        const char *c4 = c3 ? (const char *)memchr(c3 + 1, '\t', (size_t)(nlp - c3 - 1)) : nullptr;
        if (!c4 || from_len == 0) { outb.append(p, (size_t)(nlp - p)); }
        else {
          outb.append(p, (size_t)(c3 + 1 - p));
          const size_t bl = (size_t)(c4 - c3 - 1);
          for (size_t sgm = 0; sgm < bl / from_len; ++sgm) {
            auto it = table.find(std::string(c3 + 1 + sgm * from_len, from_len));
            if (it == table.end()) die("Barcode does not exist in the translation table.");
            if (sgm) outb.push_back('-');
            outb.append(it->second);
          }
          outb.append(c4, (size_t)(nlp - c4));
        }
        outb.push_back('\n');
        if (outb.size() > (1 << 20) - 4096) { fwrite(outb.data(), 1, outb.size(), of); outb.clear(); }
        p = nlp + 1;
      }
      if (!outb.empty()) fwrite(outb.data(), 1, outb.size(), of);
      fclose(of);
    } else {
      // sections in rank order: the owners hold contiguous, increasing chromosome ranges
      for (size_t gi = 0; gi < ctxs.size(); ++gi)
        if (cmgpu_store_write_text(ctxs[gi], a.out_path.c_str(), 1) != CMGPU_OK) die(cmgpu_last_error(ctxs[gi]));  // (emptied at the start)
    }
    t_post = now_s() - t0;
    fprintf(stderr, "Sorted, deduplicated and formatted %llu bytes on the device in %.3fs, wrote them in %.3fs.\n", (unsigned long long)nbytes,
            t1 - t0, now_s() - t1);
    lines = (int64_t)nl;
  }
  if (lines < 0) die("cannot write " + a.out_path);
  fprintf(stderr, "Number of output mappings (passed filters): %lld\n", (long long)lines);
  if (device_ingest)
    fprintf(stderr, "Mapped all reads in %.2fs (file read + inflate %.2fs, H2D + device FASTQ parse %.2fs, mapping %.2fs, post-processing + write %.2fs).\n",
            now_s() - t_begin, t_read, t_parse, t_map, t_post);
  for (cmgpu_ctx *cx : ctxs) cmgpu_destroy(cx);
  cmgpu_free_host_ref(&ref);
  return 0;
}
