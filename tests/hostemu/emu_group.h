// emu_group.h -- TEST INFRASTRUCTURE.  The group type of chromap_amd/csrc/cm_coop.h for a machine without a GPU: every
// lane is a fiber (ucontext) of one OS thread; sync() parks the lane until all lanes of the group (rank(): of its wave
// part) have arrived, the cross-lane operations go through a shared slot array between two such barriers.  The lanes
// run one after the other between barriers -- in ascending or descending lane order (`reverse`), so that a stage
// function that reads another lane's shared data without a sync() in between sees stale data under one of the orders.
#ifndef EMU_GROUP_H_
#define EMU_GROUP_H_
#include <stdint.h>
#include <stdlib.h>
#include <ucontext.h>

#include <functional>
#include <vector>

struct EmuTeam {
  int G, W;
  bool reverse = false;
  std::vector<ucontext_t> ctx;
  ucontext_t main_ctx;
  std::vector<char *> stacks;
  std::vector<uint64_t> slot;
  std::vector<int> done;
  int cur = 0;
  // barrier state: arrivals and generation for the whole group and per wave part
  int all_arrived = 0;
  unsigned all_gen = 0;
  std::vector<int> wave_arrived;
  std::vector<unsigned> wave_gen;
  std::function<void(int)> body;
  EmuTeam(int g, int w) : G(g), W(w), ctx((size_t)g), stacks((size_t)g, nullptr), slot((size_t)g), done((size_t)g, 0),
                          wave_arrived((size_t)(g / w), 0), wave_gen((size_t)(g / w), 0) {}
  ~EmuTeam() { for (char *s : stacks) free(s); }
  // hand the processor to the next unfinished lane (round robin in the configured direction)
  void yield() {
    const int from = cur;
    int nxt = from;
    for (int step = 0; step < G; ++step) {
      nxt = reverse ? (nxt + G - 1) % G : (nxt + 1) % G;
      if (!done[(size_t)nxt]) break;
    }
    if (nxt == from) return;
    cur = nxt;
    swapcontext(&ctx[(size_t)from], &ctx[(size_t)nxt]);
  }
  void barrier_all() {
    const unsigned gen = all_gen;
    if (++all_arrived == G) { all_arrived = 0; ++all_gen; return; }
    while (all_gen == gen) yield();
  }
  void barrier_wave(int wv) {
    const unsigned gen = wave_gen[(size_t)wv];
    if (++wave_arrived[(size_t)wv] == W) { wave_arrived[(size_t)wv] = 0; ++wave_gen[(size_t)wv]; return; }
    while (wave_gen[(size_t)wv] == gen) yield();
  }
  static void trampoline(unsigned lo, unsigned hi) {
    EmuTeam *self = reinterpret_cast<EmuTeam *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    const int t = self->cur;
    self->body(t);
    self->done[(size_t)t] = 1;
    // leave to another unfinished lane, or back to the caller when this was the last one
    for (int i = 0; i < self->G; ++i)
      if (!self->done[(size_t)i]) { self->yield(); }
    setcontext(&self->main_ctx);
  }
  void run(std::function<void(int)> f) {
    body = std::move(f);
    const size_t STK = 256 * 1024;
    for (int t = 0; t < G; ++t) {
      if (!stacks[(size_t)t]) stacks[(size_t)t] = (char *)malloc(STK);
      getcontext(&ctx[(size_t)t]);
      ctx[(size_t)t].uc_stack.ss_sp = stacks[(size_t)t];
      ctx[(size_t)t].uc_stack.ss_size = STK;
      ctx[(size_t)t].uc_link = nullptr;
      const uintptr_t p = (uintptr_t)this;
      makecontext(&ctx[(size_t)t], (void (*)())trampoline, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
      done[(size_t)t] = 0;
    }
    cur = reverse ? G - 1 : 0;
    volatile bool started = false;
    getcontext(&main_ctx);
    if (!started) {
      started = true;
      setcontext(&ctx[(size_t)cur]);
    }
  }
};

template <int G_>
struct EmuGroup {
  static constexpr int G = G_;
  static constexpr int W = G_ < 64 ? G_ : 64;
  uint32_t t;
  EmuTeam *team;
  void sync() { team->barrier_all(); }
  void wsync() { team->barrier_wave((int)(t / W)); }
  uint64_t ballot(bool p) {  // (G <= 64)
    team->slot[t] = p ? 1 : 0;
    wsync();
    uint64_t m = 0;
    const uint32_t base = t / W * W;
    for (uint32_t i = 0; i < (uint32_t)W; ++i) m |= (uint64_t)(team->slot[base + i] ? 1 : 0) << i;
    wsync();
    return m;
  }
  uint32_t bcast(uint32_t v, uint32_t lane) {  // lane: the same for the whole wave part
    team->slot[t] = v;
    wsync();
    const uint32_t r = (uint32_t)team->slot[t / W * W + lane];
    wsync();
    return r;
  }
  uint32_t rank(bool p, uint32_t *total) {
    team->slot[t] = p ? 1 : 0;
    wsync();
    const uint32_t base = t / W * W;
    uint32_t r = 0, tot = 0;
    for (uint32_t i = 0; i < (uint32_t)W; ++i) {
      const uint32_t x = (uint32_t)team->slot[base + i];
      if (base + i < t) r += x;
      tot += x;
    }
    wsync();
    *total = tot;
    return r;
  }
  uint32_t scan(uint32_t v, uint32_t *total) {
    team->slot[t] = v;
    sync();
    uint32_t r = 0, tot = 0;
    for (uint32_t i = 0; i < (uint32_t)G; ++i) {
      const uint32_t x = (uint32_t)team->slot[i];
      if (i < t) r += x;
      tot += x;
    }
    sync();
    *total = tot;
    return r;
  }
  uint32_t scanmax(uint32_t v, uint32_t *total) {
    team->slot[t] = v;
    sync();
    uint32_t r = 0, tot = 0;
    for (uint32_t i = 0; i < (uint32_t)G; ++i) {
      const uint32_t x = (uint32_t)team->slot[i];
      if (i < t && x > r) r = x;
      if (x > tot) tot = x;
    }
    sync();
    *total = tot;
    return r;
  }
  uint64_t max64(uint64_t v) {
    team->slot[t] = v;
    sync();
    uint64_t mx = 0;
    for (uint32_t i = 0; i < (uint32_t)G; ++i) mx = team->slot[i] > mx ? team->slot[i] : mx;
    sync();
    return mx;
  }
  uint64_t min64(uint64_t v) { return ~max64(~v); }
  uint32_t sum(uint32_t v) { uint32_t tot; (void)scan(v, &tot); return tot; }
};

// runs body(group) on G lanes; body loops over its work items itself (every lane the same items in the same order)
template <int G, class F>
static void emu_run_group(F &&body, bool reverse = false) {
  EmuTeam team(G, G < 64 ? G : 64);
  team.reverse = reverse;
  team.run([&](int t) {
    EmuGroup<G> g;
    g.t = (uint32_t)t;
    g.team = &team;
    body(g);
  });
}
#endif
