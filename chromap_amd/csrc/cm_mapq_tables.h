// cm_mapq_tables.h -- host-side construction of the MAPQ lookup tables (libm on the host,
// so the device never calls log(); mapping_generator.h:924-925, 963-966, 971-974, 1092-1098)
#ifndef CM_MAPQ_TABLES_H_
#define CM_MAPQ_TABLES_H_
#include <math.h>
#include <stdint.h>

#include <vector>

// len_coef[a] = a < 50 ? 1.0 : (int)log(50) / log(a)
static inline void cm_build_len_coef(std::vector<double> &coef) {
  coef.resize(65536);
  const int mapq_coef_length = 50;
  const int mapq_coef_fraction = (int)log((double)mapq_coef_length);
  for (int a = 0; a < 65536; ++a) coef[a] = a < mapq_coef_length ? 1.0 : mapq_coef_fraction / log((double)a);
}
// brk[v] = smallest n >= 1 with (int)(4.343 * log(n + 1) + 0.499) >= v  (non-decreasing in n)
static inline void cm_build_nsec_break(std::vector<uint32_t> &brk) {
  auto f = [](uint32_t n) { return (int)(4.343 * log((double)n + 1.0) + 0.499); };
  brk.clear();
  for (int v = 0;; ++v) {
    if (f(0x7fffffffu) < v) break;
    uint32_t lo = 1, hi = 0x7fffffffu;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (f(mid) >= v) hi = mid; else lo = mid + 1;
    }
    brk.push_back(lo);
  }
}
#endif
