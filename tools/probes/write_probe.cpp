// how fast 256 MB reach a new file in the output directory: one write() stream, several pwrite() threads, several threads copying into a
// shared mapping (tools/probes: measurements behind DESIGN 11's writer, not part of the product)
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "/tmp/write_probe.bin";
  const size_t n = (size_t)256 << 20;
  std::vector<char> src(n);
  for (size_t i = 0; i < n; ++i) src[i] = (char)(i * 131);
  for (int nt : {1, 2, 4, 8, 16}) {
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 0 && nt > 1) continue;
      unlink(path);
      const double t0 = now();
      int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { perror("open"); return 1; }
      if (mode == 0) {
        for (size_t o = 0; o < n; o += (8u << 20)) if (write(fd, src.data() + o, 8u << 20) < 0) perror("write");
      } else {
        char *m = nullptr;
        if (mode == 2) {
          if (ftruncate(fd, (off_t)n) != 0) perror("ftruncate");
          m = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
          if (m == MAP_FAILED) { perror("mmap"); close(fd); continue; }
        }
        std::vector<std::thread> th;
        const size_t per = n / (size_t)nt;
        for (int t = 0; t < nt; ++t)
          th.emplace_back([&, t]() {
            const size_t lo = per * (size_t)t, hi = t == nt - 1 ? n : lo + per;
            if (mode == 1) for (size_t o = lo; o < hi; o += (8u << 20)) { if (pwrite(fd, src.data() + o, std::min<size_t>(8u << 20, hi - o), (off_t)o) < 0) perror("pwrite"); }
            else memcpy(m + lo, src.data() + lo, hi - lo);
          });
        for (auto &x : th) x.join();
        if (m) munmap(m, n);
      }
      close(fd);
      printf("%-8s threads %2d: %.1f ms\n", mode == 0 ? "write" : mode == 1 ? "pwrite" : "mmap", nt, (now() - t0) * 1e3);
    }
  }
  unlink(path);
  return 0;
}
