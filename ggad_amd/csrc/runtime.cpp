// Host-side runtime pieces of libggad_hip.so: error reporting and the batch sampler.
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
#include <cstring>
#include <string>
#include <vector>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "common.h"

static thread_local std::string g_last_error;

extern "C" {          // sampler_x86.cpp
int ggad_x86_has_avx2(void);
int ggad_x86_accept8(const uint32_t *y, int sh, uint32_t bound, int32_t *out);
int ggad_x86_accept_run(const uint32_t *y, int avail, int64_t n, int64_t *c_io, int32_t *T);
void ggad_x86_temper(const uint32_t *in, uint32_t *out, int n);
}

void ggad_set_error(hipError_t e, const char *where) {
  g_last_error = std::string(where) + ": " + hipGetErrorString(e);
}

// ---- CU-partitioned streams.  The plan of chunk c+1 (two launches that fill the chip for milliseconds) and the dense
// steps of chunk c (hundreds of dependent launches of a few microseconds) only overlap if they do not compete for
// the same compute units: a tiny launch queued behind 600k gather waves waits for a free CU.  Two streams with
// disjoint CU masks give the dense chain its own CUs.
extern "C" int ggad_stream_create_cu_mask(const uint32_t *mask, int32_t n_words, ggad_stream_t *out) {
  if (!mask || n_words <= 0 || !out) return GGAD_E_INVALID;
  hipStream_t st = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask);
  if (e != hipSuccess) { ggad_set_error(e, "stream_create_cu_mask"); return GGAD_E_LAUNCH; }
  *out = reinterpret_cast<ggad_stream_t>(st);
  return GGAD_OK;
}

extern "C" int ggad_stream_destroy(ggad_stream_t stream) {
  if (!stream) return GGAD_OK;
  hipError_t e = hipStreamDestroy(reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) { ggad_set_error(e, "stream_destroy"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}

extern "C" int ggad_device_cu_count(int32_t device, int32_t *out) {
  if (!out) return GGAD_E_INVALID;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { ggad_set_error(e, "device_cu_count"); return GGAD_E_LAUNCH; }
  *out = prop.multiProcessorCount;
  return GGAD_OK;
}

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

inline uint32_t mt_next(ggad_mt19937 *g) {
  if (g->index >= MT_N) {
    uint32_t *mt = g->mt;
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
    g->index = 0;
  }
  uint32_t y = g->mt[g->index++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline int bit_length_u64(uint64_t n) { return n ? 64 - __builtin_clzll(n) : 0; }
}  // namespace

extern "C" {

int ggad_abi_version(void) { return 2; }
const char *ggad_last_error(void) { return g_last_error.c_str(); }

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

// ---- the two halves of random.shuffle: (1) consume the generator -> swap targets T[c] (partner of position n - 1 - c);
// data-independent, so it can run ahead of (2) applying the swaps to a list.  T must hold n + 16 ints.
static void shuffle_targets(ggad_mt19937 *g, int64_t n, int32_t *T) {
  static const bool avx2 = ggad_x86_has_avx2() != 0;
  const int64_t need = n - 1;                    // accepted draws of one shuffle
  int64_t c = 0;
  uint32_t tmp[MT_N + 8];
  while (c < need) {
    if (g->index >= MT_N) { (void)mt_next(g); g->index = 0; }      // regenerate the block (mt_next twists, we rewind)
    const int avail = MT_N - g->index;
    const uint32_t *blk = g->mt + g->index;
    if (avx2) {
      ggad_x86_temper(blk, tmp, avail);
    } else {
      for (int k = 0; k < avail; ++k) {
        uint32_t y = blk[k];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        tmp[k] = y;
      }
    }
    int used = 0;
    // Blocks of 8 outputs without the loop-carried chain i -> bound -> clz -> shift -> compare: inside a block the bound can
    // drop by at most 8, so a draw r <= bound - 8 is accepted and a draw r >= bound is rejected WHATEVER the draws before
    // it did (the shift is constant while bound and bound - 8 have the same bit length).  Only a draw in the 7-wide
    // window between the two (probability ~ 8 / 2^k) makes the walk take one exact scalar step instead.
    while (used < avail && c < need) {
      if (avx2) used += ggad_x86_accept_run(tmp + used, avail - used, n, &c, T);     // as many whole blocks as the rule allows
      if (used >= avail || c >= need) break;
      const int64_t i = n - 1 - c;
      const uint32_t bound = (uint32_t)i + 1u;
      const int sh = __builtin_clz(bound);
      if (!avx2 && used + 8 <= avail && i >= 64 && __builtin_clz(bound - 8u) == sh) {
        {
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
      }
      const uint32_t r = tmp[used++] >> sh;      // exact step: block tails, power-of-two crossings, ambiguous blocks
      T[c] = (int32_t)r;                         // overwritten by the redraw if rejected
      c += (r < bound) ? 1 : 0;
    }
    g->index += used;
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

/* The reference's batch stream (src/model_handler.py:310-345) for `count` consecutive batches, two threads deep: a helper
 * thread walks the generator (targets of the per-epoch shuffle of `train` and of the per-batch shuffle of `pool`), this
 * thread applies the swaps and copies every batch out: train[i0:i1] ++ pool[:n_pseudo].  *in_epoch_io is the index of the
 * next batch inside the epoch (>= batches_per_epoch forces the epoch shuffle first).  out_nodes: count x (batch_size +
 * n_pseudo) int64, out_len[b] = nodes of batch b.  Same permutations and generator state as the one-call-per-shuffle path. */
int ggad_sched_batches(ggad_mt19937 *g, int64_t *train, int64_t n_train, int64_t *pool, int64_t n_pool, int32_t batch_size,
                       int32_t n_pseudo, int32_t batches_per_epoch, int32_t *in_epoch_io, int32_t count, int64_t *out_nodes,
                       int32_t *out_len) {
  if (!g || !train || !pool || !in_epoch_io || !out_nodes || !out_len) return GGAD_E_INVALID;
  if (n_train < 1 || n_pool < n_pseudo || batch_size < 1 || n_pseudo < 0 || batches_per_epoch < 1 || count < 0) return GGAD_E_INVALID;
  if (n_train > 0x7fffffffLL || n_pool > 0x7fffffffLL) return GGAD_E_INVALID;
  if (count == 0) return GGAD_OK;
  // work items in stream order: kind 0 = epoch shuffle of train, kind 1 = batch shuffle of pool
  struct Item { int kind; };
  std::vector<Item> items;
  int ie = *in_epoch_io;
  for (int b = 0; b < count; ++b) {
    if (ie >= batches_per_epoch) { items.push_back({0}); ie = 0; }
    items.push_back({1});
    ++ie;
  }
  constexpr int RING = 4;
  std::vector<int32_t> ring[RING];
  const int64_t tmax = (n_train > n_pool ? n_train : n_pool) + 16;
  for (auto &r : ring) r.resize((size_t)tmax);
  std::mutex mu;
  std::condition_variable cv;
  size_t produced = 0, consumed = 0;
  std::thread producer([&] {
    for (size_t i = 0; i < items.size(); ++i) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return produced - consumed < RING; });
      }
      const int64_t n = items[i].kind == 0 ? n_train : n_pool;
      if (n >= 2) shuffle_targets(g, n, ring[i % RING].data());
      {
        std::lock_guard<std::mutex> lk(mu);
        ++produced;
      }
      cv.notify_all();
    }
  });
  ie = *in_epoch_io;
  const int stride = batch_size + n_pseudo;
  int b = 0;
  for (size_t i = 0; i < items.size(); ++i) {
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return produced > i; });
    }
    if (items[i].kind == 0) {
      if (n_train >= 2) apply_swaps(train, n_train, ring[i % RING].data());
      ie = 0;
    } else {
      if (n_pool >= 2) apply_swaps(pool, n_pool, ring[i % RING].data());
      const int64_t i0 = (int64_t)ie * batch_size;
      int64_t i1 = i0 + batch_size;
      if (i1 > n_train) i1 = n_train;
      const int64_t nt = i1 > i0 ? i1 - i0 : 0;
      int64_t *dst = out_nodes + (int64_t)b * stride;
      if (nt > 0) std::memcpy(dst, train + i0, (size_t)nt * sizeof(int64_t));
      std::memcpy(dst + nt, pool, (size_t)n_pseudo * sizeof(int64_t));
      out_len[b] = (int32_t)(nt + n_pseudo);
      ++b;
      ++ie;
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      ++consumed;
    }
    cv.notify_all();
  }
  producer.join();
  *in_epoch_io = ie;
  return GGAD_OK;
}

}  // extern "C"
