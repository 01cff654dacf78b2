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
}  // namespace

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
__attribute__((target("avx2,popcnt"))) int ggad_x86_accept_run(const uint32_t *y, int avail, int64_t n, int64_t *c_io, int32_t *T) {
  int used = 0;
  int64_t c = *c_io;
  while (used + 8 <= avail) {
    const int64_t i = n - 1 - c;
    if (i < 64) break;
    const uint32_t bound = (uint32_t)i + 1u;
    const int sh = __builtin_clz(bound);
    if (__builtin_clz(bound - 8u) != sh) break;
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(y + used));
    const __m256i r = _mm256_srl_epi32(v, _mm_cvtsi32_si128(sh));
    const __m256i gt_lim = _mm256_cmpgt_epi32(r, _mm256_set1_epi32((int)(bound - 8u)));
    const __m256i lt_bnd = _mm256_cmpgt_epi32(_mm256_set1_epi32((int)bound), r);
    if (!_mm256_testz_si256(gt_lim, lt_bnd)) break;
    const int acc = (~_mm256_movemask_ps(_mm256_castsi256_ps(gt_lim))) & 0xFF;
    const __m256i perm = _mm256_load_si256(reinterpret_cast<const __m256i *>(g_lut.idx[acc]));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(T + c), _mm256_permutevar8x32_epi32(r, perm));
    c += _mm_popcnt_u32((unsigned)acc);
    used += 8;
  }
  *c_io = c;
  return used;
}

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
