// One launch per CHUNK: the optimiser steps of all batches of a chunk inside one persistent kernel (single GPU, F == 17).
//
// The launch chain of step.hip costs 5-6 launches per step, each 4-9 us although its work is a microsecond: ~2.5 us of launch
// and ~1.9 us of boundary per kernel, and every first load of a kernel misses.  scripts/grid_barrier_bench.hip measured what a
// persistent kernel pays instead: a grid barrier over 64 workgroups (one relaxed agent-scope atomic add per workgroup + a
// poll) with a hand-off of data between workgroups costs 1.3-1.6 us per phase, and data written with relaxed agent-scope
// atomic stores (write-through) is never read stale by relaxed agent-scope atomic loads on other XCDs (no release fence:
// buffer_wbl2 would write back the dirty lines of the plan kernels that share the L2s).
//
// Five phases per batch, G workgroups x 8 waves, lane = embedding channel d:
//   P1  entry chunks (<= 16 entries of ONE row): partial sums of relu(W x2[own(e)]);  rows: h1 = relu(W x1)
//   P3  positions of combined_all (with the row sums: nbar = (1/r) sum of the row's chunk partials in chunk order, and
//       gen = relu(fc nbar) of label-1 source rows): score, BCE, cosine affinity, norms, recon norm (graphsage.py:174,234,246)
//   P4  rows: loss scalars (every wave, same order), gradients w.r.t. h1 / gen / nbar folded into the backward coefficients
//   P5  entry chunks + rows: dW partial per workgroup; the relu mask of relu(W x2) is recomputed (bit-identical), not stored
//   P6  gradient reduction (fixed order) + Adam on the packed parameter block, transposed copies refreshed
// Hub rows never serialise anything: the unit of work of P1 / P5 is a 16-entry chunk, dealt over all waves.
// Everything a later phase reads from another wave goes through cld() / cst(); plan outputs (x1, x2, entry tables, labels)
// were written before the launch and are read normally.  Summation orders are fixed -> deterministic; they differ from the
// launch chain's (chunks of 16 instead of waves striding by 8), so results agree with it to fp32 round-off, not bit for bit.
#include <cmath>

#include "common.h"

namespace {

struct ParamLayoutP {                                  // same packed block as step.hip: w | W | fc | W^T | fc^T
  int D, F;
  __host__ __device__ int o_W() const { return D; }
  __host__ __device__ int o_fc() const { return D + D * F; }
  __host__ __device__ int n_train() const { return D + D * F + D * D; }
  __host__ __device__ int o_Wt() const { return n_train(); }
  __host__ __device__ int o_fcT() const { return n_train() + F * D; }
};

constexpr int PS_WAVES = 8;                            // waves per workgroup: 217 VGPRs, no spills (16 waves = 128 VGPRs spilled: 55 us per step)
constexpr int PS_CH = 16;                              // entries per chunk
constexpr int PS_FT = 17;                              // feature width (DGraph-Fin)
constexpr int PS_MAXROWS = 256;                        // rows per batch (150 + 50 on DGraph-Fin)

template <int CTRL>
__device__ __forceinline__ float dpp_p(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wsum(float v) {       // same reduction tree as step.hip's wave_sum_fast
  v += dpp_p<0x128>(v); v += dpp_p<0x124>(v); v += dpp_p<0x122>(v); v += dpp_p<0x121>(v);
  v += __shfl_xor(v, 16, GGAD_WAVE);
  v += __shfl_xor(v, 32, GGAD_WAVE);
  return v;
}
__device__ __forceinline__ float cld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cst(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rl(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float log_sigmoid_p(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

struct PersistentArgs {
  ggad_mb_step s;
  const int32_t *batch_ptr, *batch_ent_ptr;            // device, n_batches + 1 offsets each
  int n_batches, log_base;
  float *loss_log;
  float *chunk_part, *pos_scal, *pos_o, *gw_row, *dw_part;
  unsigned *bar;
};

// h = W x for the feature row held by lanes 0..16 of xv; same fma order as WCol::dot of step.hip
__device__ __forceinline__ float wdot(const float (&Wr)[PS_FT], float xv) {
  float h = 0.0f;
#pragma unroll
  for (int f = 0; f < PS_FT; ++f) h = fmaf(Wr[f], rl(xv, f), h);
  return h;
}

__global__ void __launch_bounds__(PS_WAVES * GGAD_WAVE) k_train_chunk_persistent(PersistentArgs A) {
  __shared__ int cpre[PS_MAXROWS + 1];
  __shared__ int cnt01[2];                                   // label-0 / label-1 positions of the batch
  __shared__ float accw[PS_WAVES / 2][PS_FT * GGAD_WAVE];   // dW combine: waves 8..15 hand theirs to waves 0..7 first (35 KB)
  __shared__ float red[PS_WAVES][GGAD_WAVE];
  __shared__ float fct[GGAD_MAX_D * (GGAD_MAX_D + 1)];       // fc^T of this step, rows padded to 65 floats (both access patterns
  __shared__ float sc[2];                                    // of P2 / P4 are then free of bank conflicts)
  const ggad_mb_step &S = A.s;
  const ParamLayoutP L{S.D, S.F};
  const int D = S.D;
  const int lane = lane_id(), wid = threadIdx.x / GGAD_WAVE;
  const bool on = lane < D;
  const int d = on ? lane : D - 1;
  const int fl = lane < PS_FT ? lane : PS_FT - 1;      // feature index of this lane for row loads
  const int G = gridDim.x, NWV = G * PS_WAVES, gw = blockIdx.x * PS_WAVES + wid;
  const int step0 = *S.step_counter;
  unsigned bar_k = 0;
  float *params = S.params;
  // phase clocks of workgroup 0 (100 MHz wall clock), accumulated behind the barrier word: read by scripts / tests only
  unsigned long long *prof = reinterpret_cast<unsigned long long *>(A.bar) + 2;
  unsigned long long t_prev = wall_clock64();
#define PS_TICK(slot)                                                          \
  if (blockIdx.x == 0 && threadIdx.x == 0) {                                   \
    const unsigned long long t_now = wall_clock64();                           \
    prof[slot] += t_now - t_prev;                                              \
    t_prev = t_now;                                                            \
  }

  for (int b = 0; b < A.n_batches; ++b) {
    const int row0 = A.batch_ptr[b], B = A.batch_ptr[b + 1] - row0;
    float *log8 = A.loss_log + (int64_t)8 * (A.log_base + b);
    // ---- per step, per workgroup: chunk table and label counts (LDS), Adam scalars, fc^T (LDS), W^T column (registers)
    if (wid == 0) {
      int cnt[4], tot = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane * 4 + k;
        const int r = i < B ? S.ent_ptr[row0 + i + 1] - S.ent_ptr[row0 + i] : 0;
        cnt[k] = (r + PS_CH - 1) / PS_CH;
        tot += cnt[k];
      }
      int incl = tot;                                   // inclusive wave scan of the per-lane totals
#pragma unroll
      for (int off = 1; off < GGAD_WAVE; off <<= 1) {
        const int up = __shfl_up(incl, off, GGAD_WAVE);
        if (lane >= off) incl += up;
      }
      int run = incl - tot;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = lane * 4 + k;
        if (i <= B) cpre[i] = run;
        run += cnt[k];
      }
      if (lane == GGAD_WAVE - 1 && B == PS_MAXROWS) cpre[B] = run;
    } else if (wid == 1) {
      int ones = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = lane * 4 + k;
        ones += (q < B) ? (S.pos_meta[row0 + q] & 1) : 0;
      }
      ones = wave_sum_i(ones);
      if (lane == 0) { cnt01[0] = B - ones; cnt01[1] = ones; }
    }
    for (int idx = threadIdx.x; idx < D * D; idx += PS_WAVES * GGAD_WAVE) {
      const int r2 = idx / D, c2 = idx - r2 * D;
      fct[r2 * (GGAD_MAX_D + 1) + c2] = cld(params + L.o_fcT() + idx);
    }
    float Wr[PS_FT];
#pragma unroll
    for (int f = 0; f < PS_FT; ++f) Wr[f] = cld(params + L.o_Wt() + f * D + d);
    // static (plan) data of this wave's row / position: requested now, used in P2 .. P6
    const int iw = gw < B ? gw : B - 1;                 // row AND position handled by this wave (waves >= B idle there)
    const bool has_row = gw < B;
    const int roww = row0 + iw;
    const int y_w = S.labels[roww];
    const int r_w = S.ent_ptr[roww + 1] - S.ent_ptr[roww];
    const int q1_w = S.row_pos[roww];
    const int meta_w = S.pos_meta[row0 + iw];
    const int y1_w = S.pos_meta[row0 + q1_w] & 1;
    const int rsrc_w = S.ent_ptr[(meta_w >> 2) + 1] - S.ent_ptr[meta_w >> 2];       // entries of the source row of position iw
    const int ir = NWV - 1 - gw;                        // row whose h1 / dW row item this wave computes (from the far end)
    const bool has_ir = ir < B;
    const float x1v = S.x1[(int64_t)(row0 + (has_ir ? ir : 0)) * PS_FT + fl];
    __syncthreads();
    const int nchunks = cpre[B];
    const int n0 = cnt01[0], n1 = cnt01[1];

    auto chunk_row = [&](int c) {                       // row i with cpre[i] <= c < cpre[i + 1]
      int lo = 0, hi = B;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cpre[mid] <= c) lo = mid; else hi = mid; }
      return lo;
    };
    // entries of chunk c: owner entries in `ov` (lane k = entry k), their x2 rows in xv[k] (lanes 0..16 = features)
    auto load_chunk = [&](int c, int &i, int &cnt, float (&xv)[PS_CH]) {
      i = chunk_row(c);
      const int eb = S.ent_ptr[row0 + i], ee = S.ent_ptr[row0 + i + 1];
      const int e_lo = eb + (c - cpre[i]) * PS_CH;
      cnt = min(PS_CH, ee - e_lo);
      const int ov = S.ent_own[e_lo + (lane < cnt ? lane : 0)];
#pragma unroll
      for (int k = 0; k < PS_CH; ++k) {
        const int o = __builtin_amdgcn_readlane(ov, k < cnt ? k : 0);
        xv[k] = S.x2[(int64_t)o * PS_FT + fl];
      }
    };

    // ================================================================ P1: chunk partials of relu(W x2), rows' h1
    for (int c = gw; c < nchunks; c += NWV) {
      int i, cnt; float xv[PS_CH];
      load_chunk(c, i, cnt, xv);
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < PS_CH; ++k)
        if (k < cnt) acc += fmaxf(wdot(Wr, xv[k]), 0.0f);
      cst(A.chunk_part + (int64_t)c * GGAD_WAVE + lane, acc);
    }
    if (has_ir) {                                                                               // h1 = relu(W x1[row])   :412
      const float h = fmaxf(wdot(Wr, x1v), 0.0f);
      if (on) cst(S.h1 + (int64_t)(row0 + ir) * D + lane, h);
    }
    for (int i = ir + NWV; i < B; i += NWV) {
      const float xv = S.x1[(int64_t)(row0 + i) * PS_FT + fl];
      const float h = fmaxf(wdot(Wr, xv), 0.0f);
      if (on) cst(S.h1 + (int64_t)(row0 + i) * D + lane, h);
    }
    grid_barrier(A.bar, ++bar_k * G);
    PS_TICK(0)

    // ================================================================ P2+P3: row sums, outlier generation and the positions of
    // combined_all in ONE phase: the wave of position q sums the chunk partials of row q (its nbar, stored for P4 / P6) and,
    // when the column comes from a label-1 row, those of that source row too, whose gen = relu(fc nbar) it computes and stores
    // (every label-1 row is the source of exactly one position).  One grid barrier less than with a separate row phase
    // (43.3 -> 41.8 us per step).
    for (int q = gw; q < B; q += NWV) {
      const int meta = (q == gw) ? meta_w : S.pos_meta[row0 + q];
      const int src = meta >> 2, y = meta & 1;
      const bool from_gen = (meta & 2) != 0;
      const int srel = src - row0;
      const int rq = (q == gw) ? r_w : S.ent_ptr[row0 + q + 1] - S.ent_ptr[row0 + q];
      const int rs = (q == gw) ? rsrc_w : S.ent_ptr[src + 1] - S.ent_ptr[src];
      const float wd_r = cld(params + d);
      const float hs_r = cld(S.h1 + (int64_t)src * D + d);
      // chunk partials of row q and (from_gen) of the source row, 8 + 8 loads in flight, chunk order
      const int qa = cpre[q], qb = cpre[q + 1];
      const int sa = from_gen ? cpre[srel] : 0, sb = from_gen ? cpre[srel + 1] : 0;
      float totq = 0.0f, tots = 0.0f;
      for (int o = 0; o < max(qb - qa, sb - sa); o += 8) {
        float vq[8], vs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          vq[k] = cld(A.chunk_part + (int64_t)min(qa + o + k, qb - 1) * GGAD_WAVE + lane);
          vs[k] = cld(A.chunk_part + (int64_t)(from_gen ? min(sa + o + k, sb - 1) : qa) * GGAD_WAVE + lane);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          totq += (qa + o + k < qb) ? vq[k] : 0.0f;
          tots += (sa + o + k < sb) ? vs[k] : 0.0f;
        }
      }
      const float nb_r = (1.0f / (float)rq) * totq;                                             // mask_row = mask / rowsum  :317
      if (on) cst(S.nbar + (int64_t)(row0 + q) * D + lane, nb_r);                               // to_feats_neigh[q, :]
      float c_r = hs_r;                                                                         // combined_all[:, q] = h1[src] ...
      if (from_gen) {                                                                           // ... or gen[src] = relu(fc nbar[src])  :428-430
        const float nbs = (1.0f / (float)rs) * tots;
        const float nbm = on ? nbs : 0.0f;
        float a = 0.0f;
        for (int d2 = 0; d2 < D; ++d2) a = fmaf(fct[d2 * (GGAD_MAX_D + 1) + d], rl(nbm, d2), a);
        c_r = fmaxf(a, 0.0f);
        if (on) cst(S.gen + (int64_t)src * D + lane, c_r);
      }
      const float wd = on ? wd_r : 0.0f, c = on ? c_r : 0.0f, nb = on ? nb_r : 0.0f;
      const float hs = (on && from_gen) ? hs_r : 0.0f;
      const float s = wsum(wd * c);                                                             // scores = weight.mm(embeds)  :174
      const float na = sqrtf(wsum(c * c)), nbn = sqrtf(wsum(nb * nb));
      const float nac = fmaxf(na, 1e-8f), nbc = fmaxf(nbn, 1e-8f);                              // cosine_similarity eps       :234
      const float aff = wsum((c / nac) * (nb / nbc));
      float recn = 0.0f;
      if (from_gen) { const float dl = hs - c; recn = sqrtf(wsum(dl * dl)); }                   // recon2                      :197-198
      if (lane == 0) {
        float *ps = A.pos_scal + (int64_t)q * 8, *po = A.pos_o + (int64_t)q * 8;
        cst(ps + 0, s); cst(ps + 1, aff); cst(ps + 2, na); cst(ps + 3, nbn); cst(ps + 4, recn);
        cst(po + 0, (1.0f - (float)y) * s - log_sigmoid_p(s));                                  // BCEWithLogits, pos_weight 1 :246
        cst(po + 1, y == 0 ? aff : 0.0f); cst(po + 2, y == 1 ? aff : 0.0f); cst(po + 3, recn);
      }
    }
    grid_barrier(A.bar, ++bar_k * G);
    PS_TICK(2)

    // ================================================================ P4: loss scalars, row gradients -> backward coefficients
    for (int i = gw; i < B; i += NWV) {
      const int row = row0 + i;
      const int y = (i == gw) ? y_w : S.labels[row];
      const int r = (i == gw) ? r_w : S.ent_ptr[row + 1] - S.ent_ptr[row];
      const int q1 = (i == gw) ? q1_w : S.row_pos[row];
      const int m2 = (i == gw) ? meta_w : S.pos_meta[row0 + i];
      const int y1 = (i == gw) ? y1_w : (S.pos_meta[row0 + q1] & 1);
      const int64_t off = (int64_t)row * D + d;
      // every load of the phase first (one memory round trip), then the arithmetic
      float pv[4][PS_MAXROWS / GGAD_WAVE];               // BCE, affinity (label 0 / 1), reconstruction; the two counts are static
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < PS_MAXROWS / GGAD_WAVE; ++j) pv[k][j] = cld(A.pos_o + (int64_t)min(lane + j * GGAD_WAVE, B - 1) * 8 + k);
      const float wd_r = cld(params + d);
      const float H1 = cld(S.h1 + off), NB = cld(S.nbar + off), G_r = cld(S.gen + off);
      const float *p1 = A.pos_scal + (int64_t)q1 * 8;
      const float s1 = cld(p1), aff1 = cld(p1 + 1), na1 = cld(p1 + 2), nbn1 = cld(p1 + 3), recn = cld(p1 + 4);
      const float nbq = cld(S.nbar + (int64_t)(row0 + q1) * D + d);
      const float *p2 = A.pos_scal + (int64_t)i * 8;
      const float aff2 = cld(p2 + 1), na2 = cld(p2 + 2), nbn2 = cld(p2 + 3);
      const float *c2src = (m2 & 2) ? S.gen : S.h1;
      const float c2 = cld(c2src + (int64_t)(m2 >> 2) * D + d);
      float t[6];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v = 0.0f;
#pragma unroll
        for (int j = 0; j < PS_MAXROWS / GGAD_WAVE; ++j) v += (lane + j * GGAD_WAVE < B) ? pv[k][j] : 0.0f;
        t[k] = wsum(v);
      }
      t[4] = (float)n0; t[5] = (float)n1;
      const float fB = (float)B;
      const float cls = t[0] / fB;
      const float an = t[1] / t[4], ab = t[2] / t[5];
      const float mg = 1.0f - (an - ab);                                                        // confidence_margin = 1   :236-240
      const float active = (mg >= 0.0f) ? 1.0f : 0.0f;
      const float rec_coef = 0.1f / t[5];
      if (i == 0 && lane == 0) {
        const float margin = fmaxf(mg, 0.0f), rec = t[3] / t[5];
        log8[0] = cls + margin + 0.1f * rec;                                                    // :258
        log8[1] = cls; log8[2] = margin; log8[3] = rec;
        log8[4] = rec_coef; log8[5] = active; log8[6] = t[4]; log8[7] = t[5];
      }
      const float wd = on ? wd_r : 0.0f;
      const float Gv = (y == 1) ? G_r : 0.0f;
      const float C = (y == 1) ? Gv : H1;                                                       // this row's column of combined_all
      const float nac1 = fmaxf(na1, 1e-8f), nbc1 = fmaxf(nbn1, 1e-8f);
      const float ds = (1.0f / (1.0f + expf(-s1)) - (float)y1) / fB;
      const float gq1 = active * (y1 == 0 ? -1.0f / t[4] : 1.0f / t[5]);
      const float ca = na1 > 0.0f ? C / na1 : 0.0f;
      const float dC = ds * wd + gq1 * ((nbq / nbc1) / nac1 - (aff1 / nac1) * ca);
      float gH = dC, gG = 0.0f;
      if (y == 1) {                                                                             // 0.1 * mean_i |h1_i - gen_i|
        const float tt = rec_coef * ((H1 - Gv) / recn);
        gH = tt; gG = dC - tt;
      }
      const float gwd = ds * C;
      const float nac2 = fmaxf(na2, 1e-8f), nbc2 = fmaxf(nbn2, 1e-8f);
      const float gq2 = active * (y == 0 ? -1.0f / t[4] : 1.0f / t[5]);
      const float cb = nbn2 > 0.0f ? NB / nbn2 : 0.0f;
      float dNb = gq2 * ((c2 / nac2) / nbc2 - (aff2 / nbc2) * cb);
      if (y == 1) {
        const float dZ = (Gv > 0.0f) ? gG : 0.0f;                                               // relu(fc(.))
        if (on) cst(S.dz + off, dZ);
        const float dZm = on ? dZ : 0.0f;
        float a = 0.0f;
        for (int dd = 0; dd < D; ++dd) a = fmaf(fct[d * (GGAD_MAX_D + 1) + dd], rl(dZm, dd), a);   // fc^T dZ: fc[dd][d] = fct[d][dd]
        dNb += a;
      }
      if (on) {
        cst(S.coef_a + off, (H1 > 0.0f) ? gH : 0.0f);
        cst(S.coef_g + off, dNb * (1.0f / (float)r));
      }
      cst(A.gw_row + (int64_t)i * GGAD_WAVE + lane, on ? gwd : 0.0f);
    }
    grid_barrier(A.bar, ++bar_k * G);
    PS_TICK(3)

    // ================================================================ P5: dW partial of this workgroup
    if (threadIdx.x == PS_WAVES * GGAD_WAVE - 1) {       // Adam scalars of this step (double pow: off the critical path here)
      const double t = (double)(step0 + b + 1);
      const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
      sc[0] = (float)((double)S.lr / bc1);             // step_size
      sc[1] = (float)sqrt(bc2);                        // bias_correction2_sqrt
    }
    {
      float acc[PS_FT];
#pragma unroll
      for (int f = 0; f < PS_FT; ++f) acc[f] = 0.0f;
      for (int c = gw; c < nchunks; c += NWV) {            // (x2 rows re-read: L2-hot since P1)
        int i, cnt; float xv[PS_CH];
        load_chunk(c, i, cnt, xv);
        const float cg = on ? cld(S.coef_g + (int64_t)(row0 + i) * D + d) : 0.0f;
#pragma unroll
        for (int k = 0; k < PS_CH; ++k) {
          if (k < cnt) {
            const float coef = (wdot(Wr, xv[k]) > 0.0f) ? cg : 0.0f;                            // [h2 > 0], h2 recomputed as in P1
#pragma unroll
            for (int f = 0; f < PS_FT; ++f) acc[f] = fmaf(coef, rl(xv[k], f), acc[f]);
          }
        }
      }
      if (has_ir) {
        const float ca = on ? cld(S.coef_a + (int64_t)(row0 + ir) * D + d) : 0.0f;
#pragma unroll
        for (int f = 0; f < PS_FT; ++f) acc[f] = fmaf(ca, rl(x1v, f), acc[f]);
      }
      for (int i = ir + NWV; i < B; i += NWV) {
        const float xv = S.x1[(int64_t)(row0 + i) * PS_FT + fl];
        const float ca = on ? cld(S.coef_a + (int64_t)(row0 + i) * D + d) : 0.0f;
#pragma unroll
        for (int f = 0; f < PS_FT; ++f) acc[f] = fmaf(ca, rl(xv, f), acc[f]);
      }
      if (wid >= PS_WAVES / 2) {
#pragma unroll
        for (int f = 0; f < PS_FT; ++f) accw[wid - PS_WAVES / 2][f * GGAD_WAVE + lane] = acc[f];
      }
      __syncthreads();
      if (wid < PS_WAVES / 2) {
#pragma unroll
        for (int f = 0; f < PS_FT; ++f) acc[f] += accw[wid][f * GGAD_WAVE + lane];             // wave w + wave w + 8
      }
      __syncthreads();
      if (wid < PS_WAVES / 2) {
#pragma unroll
        for (int f = 0; f < PS_FT; ++f) accw[wid][f * GGAD_WAVE + lane] = acc[f];
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < PS_FT * GGAD_WAVE; idx += PS_WAVES * GGAD_WAVE) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < PS_WAVES / 2; ++w) s += accw[w][idx];                               // fixed order
        cst(A.dw_part + (int64_t)blockIdx.x * (PS_FT * GGAD_WAVE) + idx, s);
      }
    }
    grid_barrier(A.bar, ++bar_k * G);
    PS_TICK(4)

    // ================================================================ P6: gradient reduction + Adam
    {
      const int n_train = L.n_train();
      const int n_groups = (n_train + GGAD_WAVE - 1) / GGAD_WAVE;
      for (int gidx = blockIdx.x; gidx < n_groups; gidx += G) {
        const int t = gidx * GGAD_WAVE + lane;
        float g = 0.0f;
        int pidx = -1;
        if (t < D) pidx = t;
        else if (t < D + D * PS_FT) { const int u = t - D; const int f = u / D, dd = u - f * D; pidx = L.o_W() + dd * PS_FT + f; }
        else if (t < n_train) pidx = L.o_fc() + (t - D - D * PS_FT);
        float p_pre = 0.0f, m_pre = 0.0f, v_pre = 0.0f;
        if (wid == 0) {                                   // parameter and moments: requested with the first terms
          const int pc = pidx < 0 ? 0 : pidx;
          p_pre = cld(params + pc); m_pre = S.exp_avg[pc]; v_pre = S.exp_avg_sq[pc];
        }
        // every sub-reducer (wave) takes the terms wid, wid + 16, ...; their loads are issued 4 at a time from clamped indices
        if (t < D) {
          for (int k0 = wid; k0 < B; k0 += 4 * PS_WAVES) {                                      // d w = sum_q ds_q c_q
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = cld(A.gw_row + (int64_t)min(k0 + j * PS_WAVES, B - 1) * GGAD_WAVE + t);
#pragma unroll
            for (int j = 0; j < 4; ++j) g += (k0 + j * PS_WAVES < B) ? v[j] : 0.0f;
          }
        } else if (t < D + D * PS_FT) {
          const int u = t - D;
          const int f = u / D, dd = u - f * D;
          const float *src = A.dw_part + f * GGAD_WAVE + dd;
          for (int k0 = wid; k0 < G; k0 += 4 * PS_WAVES) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = cld(src + (int64_t)min(k0 + j * PS_WAVES, G - 1) * (PS_FT * GGAD_WAVE));
#pragma unroll
            for (int j = 0; j < 4; ++j) g += (k0 + j * PS_WAVES < G) ? v[j] : 0.0f;
          }
        } else if (t < n_train) {
          const int u = t - D - D * PS_FT;
          const int dd = u / D, d2 = u - dd * D;                // d fc[dd][d2] = sum over label-1 rows of dZ[dd] * nbar[d2]
          for (int j0 = wid; j0 < n1; j0 += 4 * PS_WAVES) {
            int ra[4]; float za[4], na[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = S.pos_meta[row0 + n0 + min(j0 + j * PS_WAVES, n1 - 1)] >> 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { za[j] = cld(S.dz + (int64_t)ra[j] * D + dd); na[j] = cld(S.nbar + (int64_t)ra[j] * D + d2); }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j0 + j * PS_WAVES < n1) g = fmaf(za[j], na[j], g);
          }
        }
        red[wid][lane] = g;
        __syncthreads();
        if (wid == 0 && pidx >= 0) {
          float gs = 0.0f;
#pragma unroll
          for (int k = 0; k < PS_WAVES; ++k) gs += red[k][lane];                                // fixed order
          S.grads[pidx] = gs;
          float p = p_pre, mi = m_pre, vi = v_pre;
          gs = fmaf(S.weight_decay, p, gs);                     // grad.add(param, alpha=weight_decay)
          mi = fmaf(gs - mi, 0.1f, mi);                         // exp_avg.lerp_(grad, 1 - beta1)
          vi = fmaf(0.001f * gs, gs, vi * 0.999f);              // mul_(beta2).addcmul_(g, g, 1 - beta2)
          const float denom = sqrtf(vi) / sc[1] + 1e-8f;
          p = p - sc[0] * (mi / denom);                         // addcdiv_(exp_avg, denom, -step_size)
          cst(params + pidx, p);
          S.exp_avg[pidx] = mi; S.exp_avg_sq[pidx] = vi;
          if (pidx >= L.o_W() && pidx < L.o_fc()) {
            const int u = pidx - L.o_W(); const int dd = u / PS_FT, f = u - dd * PS_FT;
            cst(params + L.o_Wt() + f * D + dd, p);
          } else if (pidx >= L.o_fc()) {
            const int u = pidx - L.o_fc(); const int dd = u / D, d2 = u - dd * D;
            cst(params + L.o_fcT() + d2 * D + dd, p);
          }
        }
        __syncthreads();
      }
    }
    grid_barrier(A.bar, ++bar_k * G);
    PS_TICK(5)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *S.step_counter = step0 + A.n_batches;
#undef PS_TICK
}

}  // namespace

extern "C" {

int32_t ggad_mb_persistent_chunk_len(void) { return PS_CH; }
int32_t ggad_mb_persistent_max_rows(void) { return PS_MAXROWS; }

int64_t ggad_mb_persistent_ws_elems(int32_t max_chunks, int32_t n_workgroups) {
  // chunk partials | pos_scal | pos_o | gw_row | dw_part | barrier word (+ padding)
  return (int64_t)max_chunks * GGAD_WAVE + 2 * PS_MAXROWS * 8 + (int64_t)PS_MAXROWS * GGAD_WAVE +
         (int64_t)n_workgroups * PS_FT * GGAD_WAVE + 64;
}

int ggad_mb_train_chunk_persistent(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr_dev,
                                   const int32_t *batch_ent_ptr_dev, int32_t max_rows, int32_t max_chunks, int32_t n_workgroups,
                                   float *loss_log, int32_t log_base, float *workspace, ggad_stream_t stream) {
  GGAD_REQUIRE(tmpl && batch_ptr_dev && batch_ent_ptr_dev && loss_log && workspace && n_batches >= 0 && log_base >= 0);
  GGAD_REQUIRE(tmpl->params && tmpl->exp_avg && tmpl->exp_avg_sq && tmpl->grads && tmpl->step_counter && tmpl->x1 && tmpl->x2 &&
               tmpl->ent_ptr && tmpl->ent_own && tmpl->labels && tmpl->pos_meta && tmpl->row_pos && tmpl->h1 && tmpl->nbar &&
               tmpl->gen && tmpl->dz && tmpl->coef_a && tmpl->coef_g);
  GGAD_REQUIRE(tmpl->F == PS_FT && tmpl->D >= 1 && tmpl->D <= GGAD_MAX_D && max_rows >= 1 && max_rows <= PS_MAXROWS);
  GGAD_REQUIRE(max_chunks >= 1 && n_workgroups >= 1 && n_workgroups <= 1024);
  if (n_batches == 0) return GGAD_OK;
  PersistentArgs A;
  A.s = *tmpl;
  A.batch_ptr = batch_ptr_dev; A.batch_ent_ptr = batch_ent_ptr_dev;
  A.n_batches = n_batches; A.log_base = log_base; A.loss_log = loss_log;
  float *w = workspace;
  A.chunk_part = w; w += (int64_t)max_chunks * GGAD_WAVE;
  A.pos_scal = w; w += PS_MAXROWS * 8;
  A.pos_o = w; w += PS_MAXROWS * 8;
  A.gw_row = w; w += (int64_t)PS_MAXROWS * GGAD_WAVE;
  A.dw_part = w; w += (int64_t)n_workgroups * PS_FT * GGAD_WAVE;
  A.bar = reinterpret_cast<unsigned *>(w);
  hipStream_t st = as_stream(stream);
  hipError_t e = hipMemsetAsync(A.bar, 0, 64 * sizeof(float), st);
  if (e != hipSuccess) { ggad_set_error(e, "train_chunk_persistent memset"); return GGAD_E_LAUNCH; }
  k_train_chunk_persistent<<<dim3(n_workgroups), dim3(PS_WAVES * GGAD_WAVE), 0, st>>>(A);
  GGAD_CHECK_LAUNCH("train_chunk_persistent");
  return GGAD_OK;
}

}  // extern "C"
