// AVX2 block step of the CPython-compatible shuffle (runtime.cpp: ggad_mt_shuffle_i64).  Plain C++ (no HIP): the
// functions carry target attributes and are only reached after a run-time CPU check, so the library still loads on a
// host without AVX2.
//
// accept8: eight tempered MT outputs y[0..8) against the current bound (bound = i + 1 of the Fisher-Yates walk).  A draw
// r = y >> sh is accepted for sure if r <= bound - 8 and rejected for sure if r >= bound, whatever the seven other draws
// do; a draw in between makes the block ambiguous (return -1, caller takes exact scalar steps).  Accepted draws are
// left-packed to out[0..count) in order.
#include <cstdint>
#include <immintrin.h>

namespace {
struct PackLut {
  alignas(32) uint32_t idx[256][8];
  PackLut() {
    for (int m = 0; m < 256; ++m) {
      int c = 0;
      for (int k = 0; k < 8; ++k)
        if (m & (1 << k)) idx[m][c++] = (uint32_t)k;
      for (; c < 8; ++c) idx[m][c] = 0;
    }
  }
};
const PackLut g_lut;

// One window of NV x 8 outputs at *used_io / *c_io.  Returns false (nothing consumed) if the window rule does not apply: too few
// outputs left, bound - w crossing a power of two, or an ambiguous draw.
template <int NV>
__attribute__((target("avx2,popcnt"))) static inline bool accept_window(const uint32_t *y, int avail, int64_t n, int *used_io, int64_t *c_io,
                                                                       int32_t *T) {
  constexpr int W = 8 * NV;
  const int used = *used_io;
  int64_t c = *c_io;
  if (used + W > avail) return false;
  const uint32_t bound = (uint32_t)(n - 1 - c) + 1u;
  const int sh = __builtin_clz(bound);
  if (__builtin_clz(bound - (uint32_t)W) != sh) return false;
  const __m128i shv = _mm_cvtsi32_si128(sh);
  const __m256i lim = _mm256_set1_epi32((int)(bound - (uint32_t)W)), bnd = _mm256_set1_epi32((int)bound);
  __m256i r[NV], amb = _mm256_setzero_si256();
  int acc[NV];
#pragma GCC unroll 8
  for (int v = 0; v < NV; ++v) {
    r[v] = _mm256_srl_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(y + used + 8 * v)), shv);
    const __m256i gt_lim = _mm256_cmpgt_epi32(r[v], lim);
    amb = _mm256_or_si256(amb, _mm256_and_si256(gt_lim, _mm256_cmpgt_epi32(bnd, r[v])));
    acc[v] = (~_mm256_movemask_ps(_mm256_castsi256_ps(gt_lim))) & 0xFF;
  }
  if (!_mm256_testz_si256(amb, amb)) return false;
#pragma GCC unroll 8
  for (int v = 0; v < NV; ++v) {
    const __m256i perm = _mm256_load_si256(reinterpret_cast<const __m256i *>(g_lut.idx[acc[v]]));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(T + c), _mm256_permutevar8x32_epi32(r[v], perm));
    c += _mm_popcnt_u32((unsigned)acc[v]);
  }
  *used_io = used + W;
  *c_io = c;
  return true;
}

}  // namespace

#pragma GCC visibility push(hidden)        // internal to the library: not part of the C ABI (tests/test_abi.py)
extern "C" {

int ggad_x86_has_avx2(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }

// all values are < 2^31 (sh >= 1 because bound < 2^31), so signed compares are exact
__attribute__((target("avx2,popcnt"))) int ggad_x86_accept8(const uint32_t *y, int sh, uint32_t bound, int32_t *out) {
  const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(y));
  const __m256i r = _mm256_srl_epi32(v, _mm_cvtsi32_si128(sh));
  const __m256i lim = _mm256_set1_epi32((int)(bound - 8u));
  const __m256i bnd = _mm256_set1_epi32((int)bound);
  const __m256i gt_lim = _mm256_cmpgt_epi32(r, lim);            // r >  bound - 8
  const __m256i lt_bnd = _mm256_cmpgt_epi32(bnd, r);            // r <  bound
  if (!_mm256_testz_si256(gt_lim, lt_bnd)) return -1;           // some draw inside the window: ambiguous
  const int acc = (~_mm256_movemask_ps(_mm256_castsi256_ps(gt_lim))) & 0xFF;      // r <= bound - 8
  const __m256i perm = _mm256_load_si256(reinterpret_cast<const __m256i *>(g_lut.idx[acc]));
  _mm256_storeu_si256(reinterpret_cast<__m256i *>(out), _mm256_permutevar8x32_epi32(r, perm));
  return _mm_popcnt_u32((unsigned)acc);
}

// A run of blocks in one call: consumes outputs y[0..avail) eight at a time while the block rule applies (position
// i = n - 1 - c >= 64, no power-of-two crossing inside the block, no ambiguous draw); returns the outputs consumed,
// *c_io advanced by the accepted draws, their values left-packed at T[c...].
// A run of windows in one call: consumes outputs y[0..avail) while a window rule applies; returns the outputs consumed, *c_io
// advanced by the accepted draws, their values left-packed at T[c...].  Inside a window of w outputs the bound can drop by
// at most w, so r <= bound - w is accepted and r >= bound rejected whatever the other draws do -- all compares of a window
// use the SAME two thresholds (no loop-carried count -> bound -> compare chain, which made a fixed 8-wide walk latency-bound
// at ~25 cycles per block); the only serial part left is the store cursor.  A draw inside the (w - 1)-wide gap (probability
// w / 2^k each) fails the window: the width is chosen from the position (64 / 32 / 16 / 8 outputs), and a failed window is
// retried one size down.
__attribute__((target("avx2,popcnt"))) int ggad_x86_accept_run(const uint32_t *y, int avail, int64_t n, int64_t *c_io, int32_t *T) {
  int used = 0;
  int64_t c = *c_io;
  for (;;) {
    const int64_t i = n - 1 - c;
    if (i < 64) break;
    bool ok = false;
    if (i >= 8192) ok = accept_window<8>(y, avail, n, &used, &c, T);
    if (!ok && i >= 2048) ok = accept_window<4>(y, avail, n, &used, &c, T);
    if (!ok && i >= 512) ok = accept_window<2>(y, avail, n, &used, &c, T);
    if (!ok) ok = accept_window<1>(y, avail, n, &used, &c, T);
    if (!ok) break;                      // caller takes one exact step (or fetches more outputs)
  }
  *c_io = c;
  return used;
}

// MT19937 state transition ("twist") of the whole 624-word block, eight words at a time.  Word kk reads the OLD words kk, kk + 1
// and -- for kk < 227 -- the old word kk + 397, for kk >= 227 the NEW word kk - 227: an 8-wide step never reads a word it
// writes itself (distances 1 ahead -- still old when the step starts -- and 227 behind), so the vector walk equals the scalar one.
#define GGAD_TWIST8(kk, src)                                                                                          \
  do {                                                                                                                \
    const __m256i a_ = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(mt + (kk)));                              \
    const __m256i b_ = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(mt + (kk) + 1));                          \
    const __m256i m_ = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(mt + (src)));                             \
    const __m256i y_ = _mm256_or_si256(_mm256_and_si256(a_, upper), _mm256_and_si256(b_, lower));                     \
    const __m256i odd_ = _mm256_sub_epi32(_mm256_setzero_si256(), _mm256_and_si256(y_, one)); /* ones where y is odd */ \
    const __m256i r_ = _mm256_xor_si256(_mm256_xor_si256(m_, _mm256_srli_epi32(y_, 1)), _mm256_and_si256(odd_, mag)); \
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(mt + (kk)), r_);                                                  \
  } while (0)
#define GGAD_TWIST1(kk, nxt, src)                                                   \
  do {                                                                              \
    const uint32_t y_ = (mt[kk] & 0x80000000u) | (mt[nxt] & 0x7fffffffu);           \
    mt[kk] = mt[src] ^ (y_ >> 1) ^ ((y_ & 1u) ? 0x9908b0dfu : 0u);                  \
  } while (0)
__attribute__((target("avx2"))) void ggad_x86_mt_twist(uint32_t *mt) {
  const __m256i upper = _mm256_set1_epi32((int)0x80000000u), lower = _mm256_set1_epi32(0x7fffffff);
  const __m256i mag = _mm256_set1_epi32((int)0x9908b0dfu), one = _mm256_set1_epi32(1);
  int kk = 0;
  for (; kk + 8 <= 227; kk += 8) GGAD_TWIST8(kk, kk + 397);          // 0 .. 223
  for (; kk < 227; ++kk) GGAD_TWIST1(kk, kk + 1, kk + 397);           // 224 .. 226
  for (; kk + 8 <= 623; kk += 8) GGAD_TWIST8(kk, kk - 227);           // 227 .. 618 (word kk + 8 <= 623 read as "next" is still old)
  for (; kk < 623; ++kk) GGAD_TWIST1(kk, kk + 1, kk - 227);
  GGAD_TWIST1(623, 0, 396);
}
#undef GGAD_TWIST8
#undef GGAD_TWIST1

// MT19937 tempering of n outputs (vectorised)
__attribute__((target("avx2"))) void ggad_x86_temper(const uint32_t *in, uint32_t *out, int n) {
  int k = 0;
  const __m256i m1 = _mm256_set1_epi32((int)0x9d2c5680u), m2 = _mm256_set1_epi32((int)0xefc60000u);
  for (; k + 8 <= n; k += 8) {
    __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(in + k));
    y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
    y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), m1));
    y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), m2));
    y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + k), y);
  }
  for (; k < n; ++k) {
    uint32_t y = in[k];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    out[k] = y;
  }
}

}  // extern "C"
#pragma GCC visibility pop
