// Device helpers shared by the launch-chain step kernels (step.hip) and the XCD-resident chunk kernel (step_xcd.hip).
#pragma once
#include "common.h"

namespace {

struct ParamLayout {
  int D, F;
  __host__ __device__ int o_w() const { return 0; }
  __host__ __device__ int o_W() const { return D; }
  __host__ __device__ int o_fc() const { return D + D * F; }
  __host__ __device__ int n_train() const { return D + D * F + D * D; }
  __host__ __device__ int o_Wt() const { return n_train(); }
  __host__ __device__ int o_fcT() const { return n_train() + F * D; }
  __host__ __device__ int n_total() const { return n_train() + F * D + D * D; }
};

// ---- wave reductions: 4 DPP row rotations (VALU) + 2 cross-row permutes instead of 6 LDS-crossbar
// permutes; every lane receives the total.  Fixed order -> deterministic.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_fast(float v) {
  v += dpp_f<0x128>(v);   // row_ror:8
  v += dpp_f<0x124>(v);   // row_ror:4
  v += dpp_f<0x122>(v);   // row_ror:2
  v += dpp_f<0x121>(v);   // row_ror:1   -> every lane holds the sum of its 16-lane row
  v += __shfl_xor(v, 16, GGAD_WAVE);
  v += __shfl_xor(v, 32, GGAD_WAVE);
  return v;
}

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

struct PosVals { float s, aff, na, nbn, nac, nbc; };

__device__ __forceinline__ PosVals eval_position(float wd, float c, float nb) {
  PosVals v;
  v.s = wave_sum_fast(wd * c);                                          // scores = weight.mm(embeds)  graphsage.py:174
  v.na = sqrtf(wave_sum_fast(c * c));
  v.nbn = sqrtf(wave_sum_fast(nb * nb));
  v.nac = fmaxf(v.na, 1e-8f);                                           // cosine_similarity eps      graphsage.py:234
  v.nbc = fmaxf(v.nbn, 1e-8f);
  v.aff = wave_sum_fast((c / v.nac) * (nb / v.nbc));
  return v;
}

__host__ __device__ inline int loss_nwg(int B) { return (B + 3) / 4; }

__device__ __forceinline__ void adam_update(float *__restrict__ params, float *__restrict__ m, float *__restrict__ v,
                                            ParamLayout L, int i, float g, float wd, float step_size, float bc2s) {
  float p = params[i];
  g = fmaf(wd, p, g);                                       // grad.add(param, alpha=weight_decay)
  float mi = m[i], vi = v[i];
  mi = fmaf(g - mi, 0.1f, mi);                              // exp_avg.lerp_(grad, 1 - beta1)
  vi = fmaf(0.001f * g, g, vi * 0.999f);                    // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(vi) / bc2s + 1e-8f;
  p = p - step_size * (mi / denom);                         // addcdiv_(exp_avg, denom, -step_size)
  params[i] = p; m[i] = mi; v[i] = vi;
  const int D = L.D, F = L.F;
  if (i >= L.o_W() && i < L.o_fc()) {
    const int u = i - L.o_W(); const int d = u / F, f = u - d * F;
    params[L.o_Wt() + f * D + d] = p;
  } else if (i >= L.o_fc()) {
    const int u = i - L.o_fc(); const int d = u / D, d2 = u - d * D;
    params[L.o_fcT() + d2 * D + d] = p;
  }
}

// ---- one-shot data-parallel exchange (host side: exchange.cpp).  Every parameter travels as ONE aligned
// 8-byte granule {gradient bits, step number}: the thread that owns parameter i stores its granule into slot (parity, rank, i) of
// EVERY rank's buffer (fine-grained memory; over xGMI for the peers; an aligned 8-byte store is indivisible) and then reads the
// granules (parity, q, i) of its own buffer until each carries this step's number.  No flag, no fence, no dependency between
// threads: the data is its own "ready" signal (the layout RCCL's low-latency protocol uses).  The W values are added in rank
// order -- the same order on every rank, so all replicas stay bit-identical.  Two parities: a rank cannot finish step s + 1
// before every peer has published s + 1, i.e. finished reading s, so it never overwrites a slot that is still being read.  A wait
// is bounded by the wall clock: a lost peer sets *err instead of hanging the GPU.
__device__ __forceinline__ float xchg_sum(const ggad_xchg_view &X, uint32_t xstep, int i, float g) {
  const int W = X.world, par = (int)(xstep & 1u);
  const int64_t n = X.n;
  const uint64_t mine = ((uint64_t)xstep << 32) | (uint64_t)__float_as_uint(g);
  for (int r = 0; r < W; ++r) {
    uint64_t *dst = reinterpret_cast<uint64_t *>(X.peer[r]) + ((int64_t)par * W + X.rank) * n + i;
    __hip_atomic_store(dst, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const uint64_t *src = reinterpret_cast<const uint64_t *>(X.peer[X.rank]) + (int64_t)par * W * n + i;
  float s = 0.0f;
  const unsigned long long t0 = wall_clock64();                            // 100 MHz
  for (int q = 0; q < W; ++q) {
    uint64_t gr = __hip_atomic_load(src + (int64_t)q * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int spins = 0;
    while ((uint32_t)(gr >> 32) != xstep) {
      __builtin_amdgcn_s_sleep(1);
      // wall-clock bound (X.timeout_ticks, GGAD_XCHG_TIMEOUT_S): a peer busy with rank-0-only host work is waited for, a lost
      // one sets the error word that every rank agrees on at its next check (DGraphTrainer.check_exchange: all-reduce MAX)
      if ((++spins & 255) == 0 && wall_clock64() - t0 > X.timeout_ticks) { *X.err = 1; break; }
      gr = __hip_atomic_load(src + (int64_t)q * n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const float gq = __uint_as_float((uint32_t)gr);
    s = q == 0 ? gq : s + gq;
  }
  return s;
}

}  // namespace
