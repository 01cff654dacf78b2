// Host-side batch sampler of libggad_hip.so.
//
// The sampler is a bit-exact re-implementation of what CPython's `random` module does for
// random.seed(int) / random.shuffle(list): the reference draws its batches with exactly these
// calls inside its timed loop (src/model_handler.py:29-30 seed, :314 shuffle of the ~1.05 M train
// list per epoch, :341 shuffle of the 55,275-element pseudo-anomaly pool PER BATCH = 28 ms/batch
// in CPython).  Algorithms restated from their published descriptions:
//   * MT19937 (Matsumoto & Nishimura 1998): init_genrand(19650218) + init_by_array(key), the key
//     being the 32-bit little-endian limbs of |seed| (CPython Modules/_randommodule.c);
//   * getrandbits(k), k <= 32: top k bits of one 32-bit output;
//   * _randbelow(n): k = n.bit_length(); draw getrandbits(k) until < n;
//   * shuffle: for i = len-1 .. 1: j = _randbelow(i+1); swap(x[i], x[j])  (CPython Lib/random.py).
// Pinned by tests/golden/sampler_shuffle.npz (captured from CPython 3.10 itself).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include "../../include/ggad_hip.h"

#pragma GCC visibility push(hidden)
extern "C" {          // sampler_x86.cpp
int ggad_x86_has_avx2(void);
int ggad_x86_accept8(const uint32_t *y, int sh, uint32_t bound, int32_t *out);
int ggad_x86_accept_run(const uint32_t *y, int avail, int64_t n, int64_t *c_io, int32_t *T);
void ggad_x86_temper(const uint32_t *in, uint32_t *out, int n);
void ggad_x86_mt_twist(uint32_t *mt624);
}
#pragma GCC visibility pop

struct ggad_mt19937 {
  uint32_t mt[624];
  int index;
};

namespace {
constexpr int MT_N = 624, MT_M = 397;

void mt_init_genrand(ggad_mt19937 *g, uint32_t s) {
  g->mt[0] = s;
  for (int i = 1; i < MT_N; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->index = MT_N;
}

void mt_init_by_array(ggad_mt19937 *g, const uint32_t *key, int len) {
  mt_init_genrand(g, 19650218u);
  int i = 1, j = 0;
  for (int k = (MT_N > len ? MT_N : len); k; --k) {
    g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
    if (++i >= MT_N) { g->mt[0] = g->mt[MT_N - 1]; i = 1; }
    if (++j >= len) j = 0;
  }
  for (int k = MT_N - 1; k; --k) {
    g->mt[i] = (g->mt[i] ^ ((g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
    if (++i >= MT_N) { g->mt[0] = g->mt[MT_N - 1]; i = 1; }
  }
  g->mt[0] = 0x80000000u;
}

void mt_twist(uint32_t *mt) {
  static const bool avx2 = ggad_x86_has_avx2() != 0;
  if (avx2) { ggad_x86_mt_twist(mt); return; }
  int kk = 0;
  for (; kk < MT_N - MT_M; ++kk) {
    uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
    mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  for (; kk < MT_N - 1; ++kk) {
    uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
    mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
  uint32_t y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
  mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

inline uint32_t mt_next(ggad_mt19937 *g) {
  if (g->index >= MT_N) {
    mt_twist(g->mt);
    g->index = 0;
  }
  uint32_t y = g->mt[g->index++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

}  // namespace

extern "C" {

ggad_mt19937 *ggad_mt_new(void) {
  ggad_mt19937 *g = new ggad_mt19937;
  uint32_t key = 0;
  mt_init_by_array(g, &key, 1);
  return g;
}
void ggad_mt_free(ggad_mt19937 *g) { delete g; }

int ggad_mt_seed_u64(ggad_mt19937 *g, uint64_t seed) {
  if (!g) return GGAD_E_INVALID;
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32)};
  mt_init_by_array(g, key, key[1] ? 2 : 1);
  return GGAD_OK;
}

int ggad_mt_set_state(ggad_mt19937 *g, const uint32_t *mt624_host, int32_t index) {
  if (!g || !mt624_host || index < 0 || index > MT_N) return GGAD_E_INVALID;
  std::memcpy(g->mt, mt624_host, sizeof(g->mt));
  g->index = index;
  return GGAD_OK;
}

int ggad_mt_get_state(const ggad_mt19937 *g, uint32_t *mt624_host, int32_t *index_host) {
  if (!g || !mt624_host || !index_host) return GGAD_E_INVALID;
  std::memcpy(mt624_host, g->mt, sizeof(g->mt));
  *index_host = g->index;
  return GGAD_OK;
}

uint32_t ggad_mt_getrandbits32(ggad_mt19937 *g) { return mt_next(g); }

static void temper_block(const uint32_t *blk, uint32_t *out, int n) {
  static const bool avx2 = ggad_x86_has_avx2() != 0;
  if (avx2) { ggad_x86_temper(blk, out, n); return; }
  for (int k = 0; k < n; ++k) {
    uint32_t y = blk[k];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[k] = y;
  }
}

// The acceptance walk of one shuffle over `avail` tempered generator outputs: position i = n - 1 - c takes its partner
// T[c] = first draw (top bit_length(i + 1) bits of an output) that is <= i.  Returns the outputs consumed; stops at `avail`
// or when c reaches `need`.  Windows of 64 / 8 outputs without the loop-carried chain i -> bound -> clz -> shift -> compare
// (sampler_x86.cpp): inside a window of w outputs the bound can drop by at most w, so a draw r <= bound - w is accepted
// and a draw r >= bound is rejected WHATEVER the draws before it did (the shift is constant while bound and bound - w have
// the same bit length).  Only a draw in the window between the two (probability ~ w / 2^k) makes the walk take exact steps.
static int walk_stream(const uint32_t *tmp, int avail, int64_t n, int64_t need, int64_t *c_io, int32_t *T) {
  static const bool avx2 = ggad_x86_has_avx2() != 0;
  int used = 0;
  int64_t c = *c_io;
  while (used < avail && c < need) {
    if (avx2) used += ggad_x86_accept_run(tmp + used, avail - used, n, &c, T);     // as many whole windows as the rule allows
    if (used >= avail || c >= need) break;
    const int64_t i = n - 1 - c;
    const uint32_t bound = (uint32_t)i + 1u;
    const int sh = __builtin_clz(bound);
    if (!avx2 && used + 8 <= avail && i >= 64 && __builtin_clz(bound - 8u) == sh) {
      const uint32_t lim = bound - 8u;
      uint32_t r[8];
      unsigned bad = 0;
      for (int k = 0; k < 8; ++k) {
        const uint32_t rk = tmp[used + k] >> sh;
        r[k] = rk;
        bad |= (unsigned)(rk > lim) & (unsigned)(rk < bound);
      }
      if (!bad) {
        int64_t pos = c;
        for (int k = 0; k < 8; ++k) {
          T[pos] = (int32_t)r[k];            // a rejected draw is overwritten by the next write to the same slot
          pos += (r[k] <= lim);
        }
        c = pos;
        used += 8;
        continue;
      }
    }
    const uint32_t r = tmp[used++] >> sh;      // exact step: window tails, power-of-two crossings, ambiguous windows
    T[c] = (int32_t)r;                         // overwritten by the redraw if rejected
    c += (r < bound) ? 1 : 0;
  }
  *c_io = c;
  return used;
}

// ---- the two halves of random.shuffle: (1) consume the generator -> swap targets T[c] (partner of position n - 1 - c);
// data-independent, so it can run ahead of (2) applying the swaps to a list.  T must hold n + 16 ints.
static void shuffle_targets(ggad_mt19937 *g, int64_t n, int32_t *T) {
  const int64_t need = n - 1;                    // accepted draws of one shuffle
  int64_t c = 0;
  uint32_t tmp[MT_N + 8];
  while (c < need) {
    if (g->index >= MT_N) { mt_twist(g->mt); g->index = 0; }
    const int avail = MT_N - g->index;
    temper_block(g->mt + g->index, tmp, avail);
    g->index += walk_stream(tmp, avail, n, need, &c, T);
  }
}

static void apply_swaps(int64_t *data, int64_t n, const int32_t *T) {
  for (int64_t k = n - 1; k >= 1; --k) {
    const int64_t cc = n - 1 - k;
    if (k >= 16) __builtin_prefetch(&data[T[cc + 16]], 1, 1);
    const int64_t j = T[cc];
    const int64_t t = data[k];
    data[k] = data[j];
    data[j] = t;
  }
}

int ggad_mt_shuffle_i64(ggad_mt19937 *g, int64_t *data, int64_t n) {
  if (!g || (!data && n > 0) || n < 0 || n > 0x7fffffffLL) return GGAD_E_INVALID;
  // This IS the per-batch cost of the reference's schedule (55,275 dependent draws).  CPython's _randbelow redraws
  // until the value is below the bound (rejected ~28 % of the time, unpredictably).  Passes per MT block: (1) temper the
  // block (vectorised), (2) walk the tempered outputs, 8 at a time where the accept rule cannot depend on the walk,
  // (3) apply the recorded swaps in order, prefetching the random targets ahead.  Same outputs consumed in the same
  // order -> same permutation, same generator state.
  static thread_local std::vector<int32_t> tgt;
  if ((int64_t)tgt.size() < n + 16) tgt.resize((size_t)n + 16);
  if (n >= 2) {
    shuffle_targets(g, n, tgt.data());
    apply_swaps(data, n, tgt.data());
  }
  return GGAD_OK;
}

/* the two halves of ggad_mt_shuffle_i64, exported for callers that pipeline them (and for scripts/sampler_bench.py) */
int ggad_mt_shuffle_targets(ggad_mt19937 *g, int64_t n, int32_t *targets_out) {
  if (!g || !targets_out || n < 0 || n > 0x7fffffffLL) return GGAD_E_INVALID;
  if (n >= 2) shuffle_targets(g, n, targets_out);
  return GGAD_OK;
}
int ggad_apply_swaps_i64(int64_t *data, int64_t n, const int32_t *targets) {
  if (!data || !targets || n < 0) return GGAD_E_INVALID;
  if (n >= 2) apply_swaps(data, n, targets);
  return GGAD_OK;
}

}  // extern "C"

// ---- the reference's batch stream as a pipeline of threads -----------------------------------------------------------
// random.shuffle is serial in TWO places: the generator (one MT19937 stream; how many outputs a draw consumes depends on
// the rejections before it) and the list (every swap sees the swaps before it).  Everything else can move off that path:
//   stage A (1 thread)   MT state transitions + tempering: the output stream does not depend on how it is consumed;
//   stage B (1 thread)   the acceptance walk over that stream -> swap targets of every shuffle, in stream order;
//   stage C (W threads)  for a pool shuffle: the permutation P it amounts to (the swaps applied to 0..n-1) -- independent of
//                        the list's contents, so consecutive shuffles are built CONCURRENTLY; for the epoch shuffle of the
//                        ~1.05 M train list: a copy of the list with the swaps applied (once per epoch, into the other half of a
//                        double buffer, while the caller still copies batches out of the old order);
//   stage M (2 threads)  the pool's state after every 8th shuffle: X_{s+1}[p] = X_s[P_f[ ... P_l[p]]], each thread its share of p;
//   stage D (caller)     the n_pseudo positions a batch reads, followed through the permutations since the last composed state,
//                        and the copy of the batch.
// Same outputs consumed in the same order, same permutations, same final lists and generator state as the per-shuffle calls.
namespace {
inline void spin_wait_step(int &spins) {
  if (++spins < 512) _mm_pause();
  else std::this_thread::yield();
}
template <class F>
inline void wait_until(F cond) {
  int spins = 0;
  while (!cond()) spin_wait_step(spins);
}

constexpr int RING_BLOCKS = 256;            // generator blocks between stage A and stage B (256 x 624 words = 640 KB)
constexpr int POOL_RING = 128;              // pool shuffles in flight between B and D
constexpr int SEG = 8;                      // pool shuffles per composed segment (stage M); POOL_RING holds 16 of them
static_assert(POOL_RING % SEG == 0 && POOL_RING >= 4 * SEG, "a slot is reused only after its whole segment");

struct SchedBuffers {                       // ~75 MB at DGraph-Fin size; reused across calls: no page faults in steady state
  std::vector<uint32_t> temp, raw;
  std::vector<int32_t> pool_t, pool_p, train_t;
  std::vector<int32_t> pool_a, pool_b;
  std::vector<int64_t> train_alt;
};
// One set per CONCURRENT call, kept process-wide: they used to be thread_local, and the trainer's producer thread -- a new thread
// per start_stream() -- paid the first touch of all of it again (~30 ms: a quarter of a 3,000-step run).
struct SchedBufferLease {
  static std::mutex &mu() { static std::mutex m; return m; }
  static std::vector<std::unique_ptr<SchedBuffers>> &idle() { static std::vector<std::unique_ptr<SchedBuffers>> v; return v; }
  std::unique_ptr<SchedBuffers> b;
  SchedBufferLease() {
    std::lock_guard<std::mutex> lk(mu());
    if (!idle().empty()) { b = std::move(idle().back()); idle().pop_back(); }
    else b.reset(new SchedBuffers());
  }
  ~SchedBufferLease() {
    std::lock_guard<std::mutex> lk(mu());
    idle().push_back(std::move(b));
  }
};

// The stages hand each other ~0.9 MB per batch (outputs -> targets -> permutation -> list).  On a multi-die host (EPYC: 8 cores per
// L3) those hand-offs cross dies unless the threads share a last-level cache: cross-die they were measured 2x SLOWER than the
// same work on one core.  So the threads of a call are confined to the CPUs that share the L3 of the CPU the caller runs on
// (intersected with the mask the process is allowed); the caller's own mask is restored when the call returns.
static bool llc_cpus_of_current(cpu_set_t *out) {
  static const bool off = [] { const char *e = getenv("GGAD_SCHED_PIN"); return e && e[0] == '0'; }();
  if (off) return false;
  const int cpu = sched_getcpu();
  if (cpu < 0) return false;
  char path[128];
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  FILE *f = fopen(path, "r");
  if (!f) return false;
  char buf[512];
  const bool got = fgets(buf, sizeof(buf), f) != nullptr;
  fclose(f);
  if (!got) return false;
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
  CPU_ZERO(out);
  int n = 0;
  for (const char *p = buf; *p;) {                      // "0-7,128-135"
    char *e;
    const long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    p = e;
    if (*p == '-') { b = strtol(p + 1, &e, 10); p = e; }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (CPU_ISSET((int)c, &allowed)) { CPU_SET((int)c, out); ++n; }
    while (*p == ',' || *p == ' ' || *p == '\n') ++p;
  }
  return n >= 6;                                        // fewer CPUs than threads: leave the placement to the scheduler
}

static void perm_from_targets(int32_t *P, int64_t n, const int32_t *T) {
  for (int64_t k = 0; k < n; ++k) P[k] = (int32_t)k;
  for (int64_t k = n - 1; k >= 1; --k) {
    const int64_t cc = n - 1 - k;
    if (k >= 16) __builtin_prefetch(&P[T[cc + 16]], 1, 1);
    const int32_t j = T[cc];
    const int32_t t = P[k];
    P[k] = P[j];
    P[j] = t;
  }
}
}  // namespace

extern "C" {

/* The reference's batch stream (src/model_handler.py:310-345) for `count` consecutive batches.  *in_epoch_io is the index of
 * the next batch inside the epoch (>= batches_per_epoch forces the epoch shuffle first).  out_nodes: count x (batch_size +
 * n_pseudo) int64, out_len[b] = nodes of batch b.  train / pool end up shuffled in place exactly as the per-shuffle calls
 * would leave them. */
int ggad_sched_batches(ggad_mt19937 *g, int64_t *train, int64_t n_train, int64_t *pool, int64_t n_pool, int32_t batch_size,
                       int32_t n_pseudo, int32_t batches_per_epoch, int32_t *in_epoch_io, int32_t count, int64_t *out_nodes,
                       int32_t *out_len) {
  if (!g || !train || !pool || !in_epoch_io || !out_nodes || !out_len) return GGAD_E_INVALID;
  if (n_train < 1 || n_pool < n_pseudo || batch_size < 1 || n_pseudo < 0 || batches_per_epoch < 1 || count < 0) return GGAD_E_INVALID;
  if (n_train > 0x7fffffffLL || n_pool > 0x7fffffffLL) return GGAD_E_INVALID;
  if (count == 0) return GGAD_OK;
  for (int64_t p = 0; p < n_pool; ++p)
    if (pool[p] < 0 || pool[p] > 0x7fffffffLL) return GGAD_E_INVALID;
  // work items in stream order: kind 0 = epoch shuffle of train, kind 1 = batch shuffle of pool
  // free_after: item whose consumption frees this item's target slot; free_prev (epoch shuffles): the previous epoch shuffle
  struct Item { int kind; int slot; int64_t free_after; int64_t free_prev; int64_t ord; };     // ord: ordinal among the pool shuffles
  std::vector<Item> items;
  std::vector<int64_t> pool_idx;                                // item index of every pool shuffle
  {
    std::vector<int64_t> train_idx;
    int ie = *in_epoch_io;
    for (int b = 0; b < count; ++b) {
      if (ie >= batches_per_epoch) {
        const size_t r = train_idx.size();
        items.push_back({0, (int)(r % 2), r >= 2 ? train_idx[r - 2] : -1, r >= 1 ? train_idx[r - 1] : -1, -1});
        train_idx.push_back((int64_t)items.size() - 1);
        ie = 0;
      }
      const size_t q = pool_idx.size();
      items.push_back({1, (int)(q % POOL_RING), q >= (size_t)POOL_RING ? pool_idx[q - POOL_RING] : -1, -1, (int64_t)q});
      pool_idx.push_back((int64_t)items.size() - 1);
      ++ie;
    }
  }
  const int64_t n_items = (int64_t)items.size();
  SchedBufferLease lease;
  SchedBuffers &B = *lease.b;
  const size_t ring_words = (size_t)RING_BLOCKS * MT_N;
  if (B.temp.size() < ring_words + 64) { B.temp.assign(ring_words + 64, 0u); B.raw.assign(ring_words, 0u); }
  const size_t pstride = (size_t)n_pool + 16;
  if (B.pool_t.size() < pstride * POOL_RING) { B.pool_t.assign(pstride * POOL_RING, 0); B.pool_p.assign(pstride * POOL_RING, 0); }
  if (B.pool_a.size() < (size_t)n_pool) { B.pool_a.assign((size_t)n_pool, 0); B.pool_b.assign((size_t)n_pool, 0); }
  const size_t tstride = (size_t)n_train + 16;
  bool any_train = false;
  for (const Item &it : items) any_train |= it.kind == 0;
  if (any_train && B.train_t.size() < 2 * tstride) B.train_t.assign(2 * tstride, 0);
  if (any_train && B.train_alt.size() < (size_t)n_train) B.train_alt.assign((size_t)n_train, 0);
  uint32_t *temp = B.temp.data(), *raw = B.raw.data();
  int32_t *pool_t = B.pool_t.data(), *pool_p = B.pool_p.data(), *train_t = any_train ? B.train_t.data() : nullptr;
  int32_t *pool_a = B.pool_a.data(), *pool_b = B.pool_b.data();
  // the train list is double-buffered: epoch shuffle r reads tb[r % 2] and leaves the new order in tb[(r + 1) % 2], so the
  // caller's thread keeps copying batches of the running epoch out of the old order while a worker builds the next one
  int64_t *tb[2] = {train, any_train ? B.train_alt.data() : nullptr};

  // one cache line each: every one of them is polled by a thread other than its writer
  alignas(64) std::atomic<int64_t> produced{0};
  alignas(64) std::atomic<int64_t> released{0};
  alignas(64) std::atomic<int64_t> done_items{0};
  alignas(64) std::atomic<int64_t> next_claim{0};
  alignas(64) std::atomic<bool> stop{false};
  alignas(64) std::atomic<int64_t> mat_done{0};              // segments whose END state has been composed (stage M)
  std::unique_ptr<std::atomic<int>[]> status(new std::atomic<int>[(size_t)n_items]);
  for (int64_t i = 0; i < n_items; ++i) status[i].store(0, std::memory_order_relaxed);

  const unsigned hw = std::thread::hardware_concurrency();
  cpu_set_t llc, caller_mask;
  const bool pin = llc_cpus_of_current(&llc) && pthread_getaffinity_np(pthread_self(), sizeof(caller_mask), &caller_mask) == 0 &&
                   pthread_setaffinity_np(pthread_self(), sizeof(llc), &llc) == 0;      // threads created below inherit the mask

  // ---- stage A: block 0 is the generator's block as it stands; every later block is one state transition further
  std::thread stage_a([&] {
    uint32_t mt[MT_N];
    std::memcpy(mt, g->mt, sizeof(mt));
    std::memcpy(raw, mt, sizeof(mt));
    temper_block(mt, temp, MT_N);
    produced.store(1, std::memory_order_release);
    for (;;) {
      int spins = 0;
      while (!stop.load(std::memory_order_relaxed) &&
             produced.load(std::memory_order_relaxed) - released.load(std::memory_order_acquire) >= RING_BLOCKS)
        spin_wait_step(spins);
      if (stop.load(std::memory_order_relaxed)) return;
      mt_twist(mt);
      const int64_t b = produced.load(std::memory_order_relaxed);
      const size_t off = (size_t)(b % RING_BLOCKS) * MT_N;
      std::memcpy(raw + off, mt, sizeof(mt));
      temper_block(mt, temp + off, MT_N);
      produced.store(b + 1, std::memory_order_release);
    }
  });

  // ---- stage B: the acceptance walk, item after item, over the ring of tempered outputs
  int64_t w_end = g->index;          // absolute position (in outputs, from the start of block 0) of the next unconsumed output
  std::thread stage_b([&] {
    int64_t w = g->index;
    for (int64_t i = 0; i < n_items; ++i) {
      const Item &it = items[(size_t)i];
      if (it.free_after >= 0) {
        if (it.kind == 1) {
          // the permutation in this slot serves EVERY later batch of its segment (their chains run through it) and the composition
          // of the segment's end state: free once the caller's thread is past the segment and stage M has composed it
          const int64_t s_old = (it.ord - POOL_RING) / SEG;
          const int64_t seg_last_item = pool_idx[(size_t)(s_old * SEG + SEG - 1)];
          wait_until([&] { return done_items.load(std::memory_order_acquire) > seg_last_item; });
          wait_until([&] { return mat_done.load(std::memory_order_acquire) > s_old; });
        } else {
          wait_until([&] { return done_items.load(std::memory_order_acquire) > it.free_after; });
        }
      }
      const int64_t n = it.kind == 0 ? n_train : n_pool;
      int32_t *T = it.kind == 0 ? train_t + (size_t)it.slot * tstride : pool_t + (size_t)it.slot * pstride;
      const int64_t need = n - 1;
      int64_t c = 0;
      while (c < need) {
        const int64_t have = produced.load(std::memory_order_acquire) * MT_N;
        if (have <= w) { int s = 0; while (produced.load(std::memory_order_acquire) * MT_N <= w) spin_wait_step(s); continue; }
        const size_t pos = (size_t)(w % (int64_t)ring_words);
        int64_t avail = have - w;
        if (avail > (int64_t)(ring_words - pos)) avail = (int64_t)(ring_words - pos);
        if (avail > (1 << 20)) avail = 1 << 20;
        w += walk_stream(temp + pos, (int)avail, n, need, &c, T);
        const int64_t keep = w / MT_N - 1;          // the block of the next output and the one before it stay readable
        if (keep > released.load(std::memory_order_relaxed)) released.store(keep, std::memory_order_release);
      }
      status[i].store(1, std::memory_order_release);
    }
    w_end = w;
  });

  // ---- stage C: permutations of the pool shuffles (concurrently), the epoch shuffle into the other train buffer
  static const int env_c = [] { const char *e = getenv("GGAD_SCHED_C"); return e ? atoi(e) : 0; }();
  int n_workers = env_c > 0 ? env_c : (hw >= 8 ? 3 : (hw >= 6 ? 2 : 1));
  if ((int64_t)n_workers > n_items) n_workers = (int)n_items;
  std::vector<std::thread> workers;
  for (int wk = 0; wk < n_workers; ++wk) {
    workers.emplace_back([&] {
      for (;;) {
        const int64_t i = next_claim.fetch_add(1, std::memory_order_relaxed);
        if (i >= n_items) return;
        const Item &it = items[(size_t)i];
        wait_until([&] { return status[i].load(std::memory_order_acquire) >= 1; });
        if (it.kind == 1) {
          perm_from_targets(pool_p + (size_t)it.slot * pstride, n_pool, pool_t + (size_t)it.slot * pstride);
        } else {
          // the previous epoch shuffle has been consumed: tb[slot] is final and nobody reads tb[1 - slot] any more
          if (it.free_prev >= 0) wait_until([&] { return done_items.load(std::memory_order_acquire) > it.free_prev; });
          int64_t *dst = tb[1 - it.slot];
          std::memcpy(dst, tb[it.slot], (size_t)n_train * sizeof(int64_t));
          if (n_train >= 2) apply_swaps(dst, n_train, train_t + (size_t)it.slot * tstride);
        }
        status[i].store(2, std::memory_order_release);
      }
    });
  }

  // ---- stage M (NM threads): the pool's state after every SEG-th shuffle.  x_q = x_{q-1}[P_q[.]], so the state after the shuffles
  // f .. l of a segment is X_s[P_f[P_{f+1}[... P_l[p]]]]: every element is a chain of SEG look-ups, independent of the others --
  // each thread composes its share of the positions (the caller's thread used to gather all 55 K elements after EVERY shuffle: 32 us
  // per batch, the bottleneck of the pipeline; it now follows only the chains of the n_pseudo positions a batch reads).
  const int64_t n_pool_items = (int64_t)pool_idx.size();
  const int64_t n_full_seg = n_pool_items / SEG;             // segments that end inside this call
  int32_t *Xb[2] = {pool_a, pool_b};                          // X_s (state before segment s) lives in Xb[s % 2]
  for (int64_t p = 0; p < n_pool; ++p) Xb[0][p] = (int32_t)pool[p];
  std::unique_ptr<std::atomic<int>[]> seg_arrived(new std::atomic<int>[(size_t)n_full_seg + 1]);
  for (int64_t sgi = 0; sgi <= n_full_seg; ++sgi) seg_arrived[sgi].store(0, std::memory_order_relaxed);
  auto perm_of = [&](int64_t q) -> const int32_t * { return pool_p + (size_t)items[(size_t)pool_idx[(size_t)q]].slot * pstride; };
  // positions [p0, p1) of X_s[P_first[ ... P_last[p]]].  Level by level over blocks of positions (each level a plain gather: independent
  // loads; following one position through all levels at a time is a chain of dependent cache misses: 43 against 32 us per batch)
  auto compose = [&](const int32_t *X, int32_t *out, int64_t first, int64_t last, int64_t p0, int64_t p1) {
    const int n_chain = (int)(last - first + 1);
    const int32_t *chain[SEG];
    for (int k = 0; k < n_chain; ++k) chain[k] = perm_of(last - k);          // applied last-to-first
    constexpr int BLK = 2048;
    int32_t idx[BLK];
    for (int64_t b0 = p0; b0 < p1; b0 += BLK) {
      const int nb = (int)std::min<int64_t>(BLK, p1 - b0);
      for (int j = 0; j < nb; ++j) idx[j] = chain[0][b0 + j];
      for (int k = 1; k < n_chain; ++k) {
        const int32_t *c = chain[k];
        for (int j = 0; j < nb; ++j) idx[j] = c[idx[j]];
      }
      for (int j = 0; j < nb; ++j) out[b0 + j] = X[idx[j]];
    }
  };
  static const int env_m = [] { const char *e = getenv("GGAD_SCHED_M"); return e ? atoi(e) : 0; }();
  const int NM = env_m > 0 ? env_m : (hw >= 8 ? 2 : 1);
  std::vector<std::thread> composers;
  if (n_pool >= 2) {
    for (int m = 0; m < NM; ++m) {
      composers.emplace_back([&, m] {
        for (int64_t sg = 0; sg < n_full_seg; ++sg) {
          const int64_t first = sg * SEG, last = first + SEG - 1;
          for (int64_t q = first; q <= last; ++q)
            wait_until([&] { return status[pool_idx[(size_t)q]].load(std::memory_order_acquire) == 2; });
          wait_until([&] { return mat_done.load(std::memory_order_acquire) >= sg; });                    // X_sg is there
          if (sg >= 1)                                           // the buffer to write still serves the batches of segment sg - 1
            wait_until([&] { return done_items.load(std::memory_order_acquire) > pool_idx[(size_t)(first - 1)]; });
          const int64_t p0 = n_pool * m / NM, p1 = n_pool * (m + 1) / NM;
          compose(Xb[sg & 1], Xb[(sg + 1) & 1], first, last, p0, p1);
          if (seg_arrived[sg].fetch_add(1, std::memory_order_acq_rel) + 1 == NM) mat_done.store(sg + 1, std::memory_order_release);
        }
      });
    }
  }

  // ---- stage D (this thread): follow the chains of the positions a batch reads, copy the batches out
  int ie = *in_epoch_io;
  const int stride = batch_size + n_pseudo;
  int b = 0;
  const int64_t *cur_train = train;
  for (int64_t i = 0; i < n_items; ++i) {
    const Item &it = items[(size_t)i];
    wait_until([&] { return status[i].load(std::memory_order_acquire) == 2; });
    if (it.kind == 0) {
      ie = 0;
      cur_train = tb[1 - it.slot];
    } else {
      const int64_t i0 = (int64_t)ie * batch_size;
      int64_t i1 = i0 + batch_size;
      if (i1 > n_train) i1 = n_train;
      const int64_t nt = i1 > i0 ? i1 - i0 : 0;
      int64_t *dst = out_nodes + (int64_t)b * stride;
      if (nt > 0) std::memcpy(dst, cur_train + i0, (size_t)nt * sizeof(int64_t));
      const int64_t sg = it.ord / SEG;
      if (n_pool >= 2) {
        wait_until([&] { return mat_done.load(std::memory_order_acquire) >= sg; });
        int32_t head[256];
        for (int64_t c0 = 0; c0 < n_pseudo; c0 += 256) {
          const int64_t c1 = std::min<int64_t>(n_pseudo, c0 + 256);
          compose(Xb[sg & 1], head - c0, sg * SEG, it.ord, c0, c1);
          for (int64_t p = c0; p < c1; ++p) dst[nt + p] = (int64_t)head[p - c0];
        }
      } else {
        for (int p = 0; p < n_pseudo; ++p) dst[nt + p] = (int64_t)Xb[0][p];
      }
      out_len[b] = (int32_t)(nt + n_pseudo);
      ++b;
      ++ie;
    }
    done_items.store(i + 1, std::memory_order_release);
  }
  for (auto &t : composers) t.join();
  // the pool's final state: the end of the last full segment, or the shuffles of a last partial segment applied to it
  int32_t *cur = Xb[n_full_seg & 1];
  if (n_pool >= 2 && n_pool_items > n_full_seg * SEG) {
    int32_t *fin = Xb[(n_full_seg + 1) & 1];
    compose(cur, fin, n_full_seg * SEG, n_pool_items - 1, 0, n_pool);
    cur = fin;
  }
  for (auto &t : workers) t.join();
  stage_b.join();
  stop.store(true, std::memory_order_relaxed);
  stage_a.join();
  if (pin) (void)pthread_setaffinity_np(pthread_self(), sizeof(caller_mask), &caller_mask);
  for (int64_t p = 0; p < n_pool; ++p) pool[p] = (int64_t)cur[p];
  if (cur_train != train) std::memcpy(train, cur_train, (size_t)n_train * sizeof(int64_t));
  // generator state: the block that holds the next unconsumed output (CPython leaves index = 624 on a block boundary)
  {
    int64_t blk = w_end / MT_N;
    int idx = (int)(w_end % MT_N);
    if (idx == 0 && w_end > 0) { blk -= 1; idx = MT_N; }
    std::memcpy(g->mt, raw + (size_t)(blk % RING_BLOCKS) * MT_N, sizeof(g->mt));
    g->index = idx;
  }
  *in_epoch_io = ie;
  return GGAD_OK;
}

}  // extern "C"
