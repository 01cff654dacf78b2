// Dense per-batch step of the DGraph mini-batch path (gfx950): GCNEncoder.forward + GCN.loss
// (src/graphsage.py:395-454, 171-258), its backward and Adam (src/model_handler.py:363-364).
//
// Shapes are tiny (B = 200 rows, a few thousand neighbourhood entries, F = 17, D = 64, 5,248
// trainable scalars) and every step depends on the previous one through the weights, so these
// kernels are latency-bound, not bandwidth-bound (SURVEY.md §7 "hard parts").  What matters is
// (1) balance: batch rows have wildly different neighbourhood sizes (power-law), so the heavy
// parts are flat over ENTRIES, not rows; (2) short dependent-load chains: indices are fetched
// once per wave with one vector load and broadcast with v_readlane, rows are then fetched
// through the scalar cache (wave-uniform) or as 256-byte coalesced vector loads; (3) few launches:
//
//   project   flat over entries   h2[u] = relu(W x2[u]) at owner entries          graphsage.py:419
//   fwd_rows  16 waves per row    nbar = mean_{e in row} h2[own(e)], h1 = relu(W x1), gen = relu(fc nbar)
//   loss_pos  1 wave / position   scores, BCE, cosine affinity, norms, recon norms + per-workgroup partial sums
//   loss_rows 1 wave / row        loss scalars, gradients w.r.t. (h1, gen, nbar, w) folded into the per-row
//                                 backward coefficients
//   bwd_flat  flat over entries   dW partials = sum coef (x) x, 4 waves per workgroup, combined in LDS
//   grad_reduce [+ adam]          partials -> packed gradient block -> (all-reduce) -> Adam
//
// lane = embedding channel d (D <= 64).  The projection uses the f32 VALU: at K = 17 an f32 MFMA tile
// (32x32x2) has the same FLOP rate and would waste 32/17 of it on padding; MFMA is used for the wide
// full-graph projections only (gemm.hip).
#include <cstdlib>

#include "common.h"
#include "step_common.h"

namespace {

constexpr int BWD_PARTS = 256;   // workgroups (= partial dW blocks) of bwd_flat (dense chain alone: 64 -> 37.5, 128 -> 33.6, 192 -> 32.0, 256 -> 31.5, 384 -> 31.1 us per step)

// W^T column of this lane, either in registers (FT > 0: compile-time F) or read from LDS.
template <int FT>
struct WCol {
  float reg[FT > 0 ? FT : 1];
  const float *lds;
  int D, F, d;
  __device__ __forceinline__ void load(const float *__restrict__ Wt, float *wt_lds, int D_, int F_, int d_, int tid, int nthreads) {
    D = D_; F = F_; d = d_; lds = wt_lds;
    if constexpr (FT > 0) {
#pragma unroll
      for (int f = 0; f < FT; ++f) reg[f] = Wt[f * D + d];
    } else {
      for (int i = tid; i < F * D; i += nthreads) wt_lds[i] = Wt[i];
      __syncthreads();
    }
  }
  // x: wave-uniform row (scalar-cache loads)
  __device__ __forceinline__ float dot(const float *__restrict__ x) const {
    float acc = 0.0f;
    if constexpr (FT > 0) {
#pragma unroll
      for (int f = 0; f < FT; ++f) acc = fmaf(reg[f], x[f], acc);
    } else {
      for (int f = 0; f < F; ++f) acc = fmaf(lds[f * D + d], x[f], acc);
    }
    return acc;
  }
};

// ------------------------------------------------------------------ project: h2 at owner entries
constexpr int PROJ_EPW = 8;   // entries per wave
template <int FT>
__global__ void __launch_bounds__(256) k_project(const float *__restrict__ params, ParamLayout L,
                                                 const float *__restrict__ x2, const int32_t *__restrict__ ent_own,
                                                 int ent0, int n_ents, float *__restrict__ h2) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = L.D, F = (FT > 0) ? FT : L.F;
  const int lane = lane_id(), wid = threadIdx.x / 64, d = lane < D ? lane : D - 1;
  WCol<FT> W;
  W.load(params + L.o_Wt(), lds, D, F, d, threadIdx.x, blockDim.x);
  const int base = (blockIdx.x * 4 + wid) * PROJ_EPW;
  if (base >= n_ents) return;
  const int ov = (lane < PROJ_EPW && base + lane < n_ents) ? ent_own[ent0 + base + lane] : -1;
#pragma unroll
  for (int i = 0; i < PROJ_EPW; ++i) {
    const int o = __builtin_amdgcn_readlane(ov, i);
    if (o != ent0 + base + i) continue;                 // not an owner (or past the end): wave-uniform branch
    const float h = fmaxf(W.dot(x2 + (int64_t)o * F), 0.0f);                       // relu(W x2[u])   graphsage.py:419
    if (lane < D) h2[(int64_t)(base + i) * D + lane] = h;
  }
}

// ------------------------------------------------------------------ forward rows (FWD_NW waves per row)
constexpr int FWD_NW = 16;      // waves per row workgroup of k_fwd_rows: a hub row (thousands of entries) is its critical path
constexpr int FWDV_NW = 8;      // ... of the fused k_fwd_rows_v (only taken by batches WITHOUT hub rows): twice the workgroups per CU
template <int FT>
__global__ void __launch_bounds__(FWD_NW * 64) k_fwd_rows(const float *__restrict__ params, ParamLayout L,
                                                  const float *__restrict__ x1, const float *__restrict__ h2,
                                                  const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ ent_own,
                                                  const int32_t *__restrict__ labels, int row0, int ent0,
                                                  float *__restrict__ h1, float *__restrict__ nbar, float *__restrict__ gen) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = L.D, F = (FT > 0) ? FT : L.F;
  const int lane = lane_id(), wid = threadIdx.x / 64, d = lane < D ? lane : D - 1;
  const int row = row0 + blockIdx.x;
  float *part = lds;                 // [FWD_NW][64]
  float *ns = lds + FWD_NW * 64;     // [64]
  float *wt_lds = ns + 64;           // F*D (FT == 0)
  const int e0 = ent_ptr[row], e1 = ent_ptr[row + 1];
  const int r = e1 - e0;
  // partial sum of h2[own(e)] over e = e0 + wid, e0 + wid + NW, ...  (256-byte coalesced row loads)
  float acc = 0.0f;
  for (int blk = wid; blk < r; blk += FWD_NW * 64) {
    const int my = blk + FWD_NW * lane;
    const int ov = (my < r) ? ent_own[e0 + my] : 0;
    const int cnt = min(64, (r - blk + FWD_NW - 1) / FWD_NW);
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {                   // hub rows: 8 row loads in flight per wave (same summation order)
      float a[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) a[q] = h2[(int64_t)(__builtin_amdgcn_readlane(ov, i + q) - ent0) * D + d];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc += a[q];
    }
    for (; i + 4 <= cnt; i += 4) {
      const int o0 = __builtin_amdgcn_readlane(ov, i), o1 = __builtin_amdgcn_readlane(ov, i + 1);
      const int o2 = __builtin_amdgcn_readlane(ov, i + 2), o3 = __builtin_amdgcn_readlane(ov, i + 3);
      const float a0 = h2[(int64_t)(o0 - ent0) * D + d], a1 = h2[(int64_t)(o1 - ent0) * D + d];
      const float a2 = h2[(int64_t)(o2 - ent0) * D + d], a3 = h2[(int64_t)(o3 - ent0) * D + d];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; i < cnt; ++i) {
      const int o = __builtin_amdgcn_readlane(ov, i);
      acc += h2[(int64_t)(o - ent0) * D + d];
    }
  }
  part[wid * 64 + lane] = acc;
  __syncthreads();
  const float inv_r = 1.0f / (float)r;                                      // mask_row = mask / rowsum  graphsage.py:317
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < FWD_NW; ++k) tot += part[k * 64 + lane];              // fixed order
  const float nb = inv_r * tot;
  const int y = labels[row];
  if (wid == 0) {
    if (lane < D) nbar[(int64_t)row * D + lane] = nb;                       // mask_row.mm(...)          graphsage.py:421
    ns[lane] = (lane < D) ? nb : 0.0f;
  }
  {                                                                         // h1 = relu(W x1[row])      graphsage.py:412
    WCol<FT> W;
    if (FT == 0 || wid == 1)                                                // FT == 0: all waves fill LDS (barrier inside)
      W.load(params + L.o_Wt(), wt_lds, D, F, d, threadIdx.x, blockDim.x);
    if (wid == 1) {
      const float h = fmaxf(W.dot(x1 + (int64_t)row * F), 0.0f);
      if (lane < D) h1[(int64_t)row * D + lane] = h;
    }
  }
  if (y != 1) return;                                                       // block-uniform exit
  __syncthreads();
  // outlier generation gen = relu(fc nbar): the waves split the d2 range              graphsage.py:428-430
  const float *fcT = params + L.o_fcT();
  const int q = (D + FWD_NW - 1) / FWD_NW;
  float a = 0.0f;
  for (int d2 = wid * q; d2 < min(D, (wid + 1) * q); ++d2) a = fmaf(fcT[d2 * D + d], ns[d2], a);
  __syncthreads();
  part[wid * 64 + lane] = a;
  __syncthreads();
  if (wid == 0 && lane < D) {
    float g = 0.0f;
#pragma unroll
    for (int k = 0; k < FWD_NW; ++k) g += part[k * 64 + lane];
    gen[(int64_t)row * D + lane] = fmaxf(g, 0.0f);
  }
}

// ------------------------------------------------------------------ loss: two launches, both spread over the chip
// (a single-workgroup version was VALU-issue-bound on ONE CU: ~400 wave-instructions per position x 200
//  positions on 4 SIMDs = 45 us; spread over 50 workgroups each launch is a few us)
//   k_loss_pos : one wave per POSITION q of combined_all: score, BCE term, cosine affinity, norms, recon norm
//                -> pos_scal[q][8] and per-workgroup partial sums
//   k_loss_rows: one wave per ROW: reduces the partials (every wave, same fixed order), then the gradients
//                w.r.t. this row's h1 / gen / nbar, folded straight into the backward coefficients.
// pos_meta[q] = (src_row << 2) | (src_is_label1 << 1) | label_of_position_q     (host-built, graphsage.py:450 order)
__global__ void __launch_bounds__(256) k_loss_pos(const float *__restrict__ params, int D, const float *__restrict__ h1,
                                                  const float *__restrict__ nbar, const float *__restrict__ gen,
                                                  const int32_t *__restrict__ pos_meta, int row0, int B,
                                                  float *__restrict__ pos_scal, float *__restrict__ part) {
  __shared__ float red[4][8];
  const int lane = lane_id(), wid = threadIdx.x / 64;
  const bool on = lane < D;
  const int q = blockIdx.x * 4 + wid;
  float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (q < B) {
    const int meta = pos_meta[row0 + q];
    const int src = meta >> 2, y = meta & 1;
    const bool from_gen = (meta & 2) != 0;
    const float wd = on ? params[lane] : 0.0f;
    const float c = on ? (from_gen ? gen[(int64_t)src * D + lane] : h1[(int64_t)src * D + lane]) : 0.0f;   // combined_all[:, q]
    const float nb = on ? nbar[(int64_t)(row0 + q) * D + lane] : 0.0f;                                    // to_feats_neigh[q, :]
    const float hs = (on && from_gen) ? h1[(int64_t)src * D + lane] : 0.0f;
    const PosVals v = eval_position(wd, c, nb);
    float recn = 0.0f;
    if (from_gen) { const float dl = hs - c; recn = sqrtf(wave_sum_fast(dl * dl)); }   // recon2   graphsage.py:197-198
    o[0] = (1.0f - (float)y) * v.s - log_sigmoid(v.s);                   // BCEWithLogits, pos_weight 1 graphsage.py:246
    o[1] = y == 0 ? v.aff : 0.0f; o[2] = y == 1 ? v.aff : 0.0f; o[3] = recn;
    o[4] = y == 0 ? 1.0f : 0.0f;  o[5] = y == 1 ? 1.0f : 0.0f;
    if (lane == 0) {
      float *ps = pos_scal + (int64_t)q * 8;
      ps[0] = v.s; ps[1] = v.aff; ps[2] = v.na; ps[3] = v.nbn; ps[4] = recn;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wid][k] = o[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    part[(int64_t)blockIdx.x * 8 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

__global__ void __launch_bounds__(256) k_loss_rows(const float *__restrict__ params, ParamLayout L,
                                                   const float *__restrict__ h1, const float *__restrict__ nbar,
                                                   const float *__restrict__ gen, const int32_t *__restrict__ labels,
                                                   const int32_t *__restrict__ pos_meta, const int32_t *__restrict__ row_pos,
                                                   const int32_t *__restrict__ ent_ptr, int row0, int B,
                                                   const float *__restrict__ pos_scal, const float *__restrict__ part,
                                                   float *__restrict__ gw_part, float *__restrict__ losses8,
                                                   float *__restrict__ d_h1, float *__restrict__ d_gen,
                                                   float *__restrict__ d_nbar, float *__restrict__ dz,
                                                   float *__restrict__ coef_a, float *__restrict__ coef_g,
                                                   int32_t *__restrict__ step_counter) {
  __shared__ float gw[4][64];
  __shared__ float zs[4][64];
  __shared__ float fc_lds[GGAD_MAX_D * GGAD_MAX_D];
  const int D = L.D;
  const int lane = lane_id(), wid = threadIdx.x / 64;
  const bool on = lane < D;
  const int d = on ? lane : D - 1;
  for (int i = threadIdx.x; i < D * D; i += 256) fc_lds[i] = params[L.o_fc() + i];
  // every wave reduces the per-workgroup partials the same way -> identical scalars everywhere
  const int nwg = loss_nwg(B);
  float t[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float v = 0.0f;
    for (int g = lane; g < nwg; g += 64) v += part[(int64_t)g * 8 + k];
    t[k] = wave_sum_fast(v);
  }
  const float fB = (float)B;
  const float cls = t[0] / fB;
  const float an = t[1] / t[4], ab = t[2] / t[5];
  const float mg = 1.0f - (an - ab);                                     // confidence_margin = 1      graphsage.py:236-240
  const float active = (mg >= 0.0f) ? 1.0f : 0.0f;                       // clamp_min backward: pass where x >= min
  const float rec_coef = 0.1f / t[5];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float margin = fmaxf(mg, 0.0f), rec = t[3] / t[5];
    losses8[0] = cls + margin + 0.1f * rec;                              // graphsage.py:258
    losses8[1] = cls; losses8[2] = margin; losses8[3] = rec;
    losses8[4] = rec_coef; losses8[5] = active; losses8[6] = t[4]; losses8[7] = t[5];
    if (step_counter) *step_counter += 1;
  }
  __syncthreads();   // fc_lds ready
  const int i = blockIdx.x * 4 + wid;
  float gwd = 0.0f;
  if (i < B) {
    const int row = row0 + i;
    const int y = labels[row];
    const int r = ent_ptr[row + 1] - ent_ptr[row];
    const float wd = on ? params[lane] : 0.0f;
    const int64_t off = (int64_t)row * D + d;
    const float H1 = h1[off];
    const float NB = nbar[off];
    const float G = (y == 1) ? gen[off] : 0.0f;
    const float C = (y == 1) ? G : H1;                                   // this row's column of combined_all
    // ---- C-side: this row is the source of column q1
    const int q1 = row_pos[row];
    const float *p1 = pos_scal + (int64_t)q1 * 8;
    const float s1 = p1[0], aff1 = p1[1], na1 = p1[2], nbn1 = p1[3], recn = p1[4];
    const int y1 = pos_meta[row0 + q1] & 1;
    const float nbq = nbar[(int64_t)(row0 + q1) * D + d];
    const float nac1 = fmaxf(na1, 1e-8f), nbc1 = fmaxf(nbn1, 1e-8f);
    const float ds = (1.0f / (1.0f + expf(-s1)) - (float)y1) / fB;
    const float gq1 = active * (y1 == 0 ? -1.0f / t[4] : 1.0f / t[5]);
    // aff = sum (c/nac)(nb/nbc); torch clamps a detached copy of the norms, so autograd sees
    // d aff / d c = (nb/nbc)/nac - (aff/nac) * c/|c|   (and symmetrically for nb)
    const float ca = na1 > 0.0f ? C / na1 : 0.0f;
    const float dC = ds * wd + gq1 * ((nbq / nbc1) / nac1 - (aff1 / nac1) * ca);
    float gH = dC, gG = 0.0f;
    if (y == 1) {                                                        // recon term 0.1 * mean_i |h1_i - gen_i|  graphsage.py:258
      const float tt = rec_coef * ((H1 - G) / recn);
      gH = tt; gG = dC - tt;
    }
    gwd = ds * C;
    // ---- nb-side: position i pairs nbar[row] with column i of combined_all
    const float *p2 = pos_scal + (int64_t)i * 8;
    const float aff2 = p2[1], na2 = p2[2], nbn2 = p2[3];
    const int m2 = pos_meta[row0 + i];
    const int src2 = m2 >> 2;
    const float c2 = (m2 & 2) ? gen[(int64_t)src2 * D + d] : h1[(int64_t)src2 * D + d];
    const float nac2 = fmaxf(na2, 1e-8f), nbc2 = fmaxf(nbn2, 1e-8f);
    const float gq2 = active * (y == 0 ? -1.0f / t[4] : 1.0f / t[5]);
    const float cb = nbn2 > 0.0f ? NB / nbn2 : 0.0f;
    float dNb = gq2 * ((c2 / nac2) / nbc2 - (aff2 / nbc2) * cb);
    if (d_h1 != nullptr && on) { d_h1[off] = gH; d_gen[off] = gG; d_nbar[off] = dNb; }
    // ---- backward coefficients of this row
    if (y == 1) {
      const float dZ = (G > 0.0f) ? gG : 0.0f;                           // relu(fc(.))
      if (on) dz[off] = dZ;
      zs[wid][lane] = on ? dZ : 0.0f;
      float a = 0.0f;
      for (int dd = 0; dd < D; ++dd) a = fmaf(fc_lds[dd * D + d], zs[wid][dd], a);   // fc^T dZ
      dNb += a;
    }
    if (on) {
      coef_a[off] = (H1 > 0.0f) ? gH : 0.0f;
      coef_g[off] = dNb * (1.0f / (float)r);
    }
  }
  gw[wid][lane] = on ? gwd : 0.0f;
  __syncthreads();
  if (threadIdx.x < 64)
    gw_part[(int64_t)blockIdx.x * 64 + threadIdx.x] = (gw[0][threadIdx.x] + gw[1][threadIdx.x]) + (gw[2][threadIdx.x] + gw[3][threadIdx.x]);
}

// Per-row backward coefficients for ARBITRARY upstream gradients (layered autograd API):
//   coef_a = d_h1 * [h1 > 0]                     multiplies x1[row]      in dW
//   dz     = d_gen * [gen > 0]  (label-1 rows)   outer(dz, nbar) = d fc
//   coef_g = (d_nbar + fc^T dz) / r              multiplies x2[own(e)] * [h2 > 0] for the row's entries
__device__ __forceinline__ void row_coefs(const float *__restrict__ fc, int D, int lane, int row, int y, int r,
                                          const float *__restrict__ h1, const float *__restrict__ gen,
                                          const float *__restrict__ d_h1, const float *__restrict__ d_gen,
                                          const float *__restrict__ d_nbar, float *__restrict__ zs_wave,
                                          float *__restrict__ dz, float *__restrict__ coef_a, float *__restrict__ coef_g) {
  const bool on = lane < D;
  const int d = on ? lane : D - 1;
  const int64_t off = (int64_t)row * D + d;
  const float H1 = h1[off];
  const float dH1 = d_h1[off];
  float dNb = d_nbar[off];
  if (y == 1) {
    const float G = gen[off];
    const float dG = d_gen[off];
    const float dZ = (G > 0.0f) ? dG : 0.0f;                            // relu(fc(.))
    if (on) dz[off] = dZ;
    zs_wave[lane] = on ? dZ : 0.0f;                                     // per-wave LDS slice, same wave reads it back
    float a = 0.0f;
    for (int dd = 0; dd < D; ++dd) a = fmaf(fc[dd * D + d], zs_wave[dd], a);   // fc^T dZ
    dNb += a;
  }
  if (on) {
    coef_a[off] = (H1 > 0.0f) ? dH1 : 0.0f;
    coef_g[off] = dNb * (1.0f / (float)r);
  }
}

// stand-alone VJP front end (layered autograd API): upstream gradients given by the caller
__global__ void __launch_bounds__(256) k_row_coefs(const float *__restrict__ params, ParamLayout L,
                                                   const int32_t *__restrict__ labels, const int32_t *__restrict__ ent_ptr,
                                                   int row0, int B, const float *__restrict__ h1, const float *__restrict__ gen,
                                                   const float *__restrict__ d_h1, const float *__restrict__ d_gen,
                                                   const float *__restrict__ d_nbar, float *__restrict__ dz,
                                                   float *__restrict__ coef_a, float *__restrict__ coef_g) {
  __shared__ float zs[4][64];
  const int wid = threadIdx.x / 64;
  const int i = blockIdx.x * 4 + wid;
  if (i >= B) return;
  const int row = row0 + i;
  row_coefs(params + L.o_fc(), L.D, lane_id(), row, labels[row], ent_ptr[row + 1] - ent_ptr[row], h1, gen, d_h1, d_gen,
            d_nbar, zs[wid], dz, coef_a, coef_g);
}

// ------------------------------------------------------------------ backward, flat over entries + rows
// work item idx < n_ents : entry e = ent0 + idx :  coef = coef_g[row(e)] * [h2[own(e)] > 0],  x = x2[own(e)]
//           idx >= n_ents: row  = row0 + idx - n_ents:  coef = coef_a[row],                  x = x1[row]
// dW[d][f] = sum_items coef_d * x_f.  Every workgroup (4 waves) writes one partial [F][D].
template <int FT>
__global__ void __launch_bounds__(256) k_bwd_flat(ParamLayout L, const float *__restrict__ x1, const float *__restrict__ x2,
                                                  const float *__restrict__ h2, const int32_t *__restrict__ ent_own,
                                                  const int32_t *__restrict__ ent_row, int row0, int n_rows, int ent0,
                                                  int n_ents, const float *__restrict__ coef_a,
                                                  const float *__restrict__ coef_g, float *__restrict__ dw_part,
                                                  int h2_by_entry, const float *__restrict__ Wt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [4][F*D]
  const int D = L.D, F = (FT > 0) ? FT : L.F;
  const int lane = lane_id(), wid = threadIdx.x / 64;
  const bool on = lane < D;
  const int d = on ? lane : D - 1;
  // h2_by_entry == 2 (FT > 0 only): h2 is not stored anywhere; its sign is recomputed from the feature row of the item
  float Wc[FT > 0 ? FT : 1];
  if constexpr (FT > 0) {
    if (h2_by_entry == 2) {
#pragma unroll
      for (int f = 0; f < FT; ++f) Wc[f] = Wt[f * D + d];
    }
  }
  float acc[FT > 0 ? FT : 1];
  float *acc_lds = lds + wid * F * D;
  if constexpr (FT > 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f) acc[f] = 0.0f;
  } else {
    for (int i = lane; i < F * D; i += 64) acc_lds[i] = 0.0f;
  }
  const int n_items = n_ents + n_rows;
  const int wave_g = blockIdx.x * 4 + wid, n_waves = gridDim.x * 4;
  for (int base = wave_g; base < n_items; base += n_waves * 64) {
    // lane l holds the indices of item base + l * n_waves
    const int idx = base + lane * n_waves;
    int xo = -1, co = 0, ho = 0;                     // xo: row in x2 (>= 0) or x1 (encoded as -2 - row); co: coef row; ho: h2 row
    if (idx < n_ents) {
      const int e = ent0 + idx;
      const int o = ent_own[e];
      xo = o; ho = h2_by_entry ? idx : o - ent0; co = ent_row[e];      // h2 stored per owner (k_project) or per entry (k_fwd_rows_v)
    } else if (idx < n_items) {
      co = row0 + idx - n_ents; xo = -2 - co;
    }
    const int cnt = min(64, (n_items - base + n_waves - 1) / n_waves);
    // item -> (coefficient of this lane's channel, wave-uniform feature row); two items in flight per iteration
    auto fetch = [&](int i, float &coef, const float *&xr) {
      const int sx = __builtin_amdgcn_readlane(xo, i);
      const int sc = __builtin_amdgcn_readlane(co, i);
      const int sh = __builtin_amdgcn_readlane(ho, i);
      if (sx >= 0) {
        const float g = coef_g[(int64_t)sc * D + d];
        xr = x2 + (int64_t)sx * F;
        float hv;
        if (FT > 0 && h2_by_entry == 2) {
          hv = 0.0f;
#pragma unroll
          for (int f = 0; f < (FT > 0 ? FT : 1); ++f) hv = fmaf(Wc[f], xr[f], hv);      // same fma order as the forward
        } else {
          hv = h2[(int64_t)sh * D + d];
        }
        coef = (on && hv > 0.0f) ? g : 0.0f;
      } else {
        coef = on ? coef_a[(int64_t)sc * D + d] : 0.0f;
        xr = x1 + (int64_t)(-2 - sx) * F;
      }
    };
    int i = 0;
    if constexpr (FT > 0) {
      for (; i + 2 <= cnt; i += 2) {
        float c0, c1;
        const float *xa, *xb;
        fetch(i, c0, xa);
        fetch(i + 1, c1, xb);
        float va[FT], vb[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) { va[f] = xa[f]; vb[f] = xb[f]; }
#pragma unroll
        for (int f = 0; f < FT; ++f) { acc[f] = fmaf(c0, va[f], acc[f]); acc[f] = fmaf(c1, vb[f], acc[f]); }
      }
    }
    for (; i < cnt; ++i) {
      float coef;
      const float *xr;
      fetch(i, coef, xr);
      if constexpr (FT > 0) {
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = fmaf(coef, xr[f], acc[f]);
      } else if (on) {
        for (int f = 0; f < F; ++f) acc_lds[f * D + d] = fmaf(coef, xr[f], acc_lds[f * D + d]);
      }
    }
  }
  if constexpr (FT > 0) {
    if (on) {
#pragma unroll
      for (int f = 0; f < FT; ++f) acc_lds[f * D + d] = acc[f];
    }
  }
  __syncthreads();
  float *out = dw_part + (int64_t)blockIdx.x * F * D;
  for (int i = threadIdx.x; i < F * D; i += 256)
    out[i] = (lds[i] + lds[F * D + i]) + (lds[2 * F * D + i] + lds[3 * F * D + i]);
}

// ------------------------------------------------------------------ gradient reduce (+ fused Adam), sync, score
__device__ __forceinline__ void adam_scalars(float *sc, const int32_t *step_counter, float lr) {
  if (threadIdx.x == 0) {
    const double t = (double)(*step_counter);
    const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
    sc[0] = (float)((double)lr / bc1);      // step_size
    sc[1] = (float)sqrt(bc2);               // bias_correction2_sqrt
  }
  __syncthreads();
}

// block = 64 parameters x GR_SUB sub-reducers (threadIdx.x = parameter, threadIdx.y = slice of the terms): every thread
// has at most 8 independent loads in flight and one round of them (128 partials / 16 slices), so the launch is one
// memory round trip + the LDS combine instead of four dependent rounds.
constexpr int GR_SUB = 16;
// MODE 0: gradients only; 1: + Adam (single GPU); 2: + the one-shot data-parallel exchange (xchg_sum) and Adam.
template <int MODE>
__global__ void __launch_bounds__(64 * GR_SUB) k_grad_reduce(ParamLayout L, const int32_t *__restrict__ pos_meta, int row0,
                                                     const float *__restrict__ losses8, const float *__restrict__ nbar,
                                                     const float *__restrict__ dw_part, int n_parts,
                                                     const float *__restrict__ dz, const float *__restrict__ gw_part,
                                                     int n_gw, float *__restrict__ grads, float *__restrict__ params,
                                                     float *__restrict__ m, float *__restrict__ v, float lr, float wd,
                                                     const int32_t *__restrict__ step_counter, ggad_xchg_view X, uint32_t xstep,
                                                     float grad_scale) {
  constexpr bool FUSE_ADAM = MODE != 0;
  __shared__ float sc[2];
  __shared__ float red[GR_SUB][64];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  if (FUSE_ADAM) {
    if (tid == 0) {
      const double t = (double)(*step_counter);
      const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
      sc[0] = (float)((double)lr / bc1);      // step_size
      sc[1] = (float)sqrt(bc2);               // bias_correction2_sqrt
    }
  }
  const int D = L.D, F = L.F;
  const int t = blockIdx.x * 64 + threadIdx.x;
  const int sub = threadIdx.y;
  float g = 0.0f;
  int pidx = -1;
  if (t < D) {
    for (int k = sub; k < n_gw; k += GR_SUB) g += gw_part[(int64_t)k * 64 + t];      // d w = sum_q ds_q * combined_all[:, q]
    pidx = t;
  } else if (t < D + D * F) {
    const int u = t - D;
    const int f = u / D, d = u - f * D;           // consecutive threads -> consecutive d (coalesced reads)
    const float *p = dw_part + f * D + d;
    const int stride = F * D;
    const int per = (n_parts + GR_SUB - 1) / GR_SUB;
    const int b0 = sub * per, b1 = min(n_parts, b0 + per);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += p[(int64_t)(b + k) * stride];
    }
    for (; b < b1; ++b) s[0] += p[(int64_t)b * stride];
    g = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    pidx = L.o_W() + d * F + f;
  } else if (t < L.n_train()) {
    const int u = t - D - D * F;
    const int dd = u / D, d2 = u - dd * D;        // d fc[dd][d2] = sum_{label-1 rows i} dZ_i[dd] * nbar_i[d2]
    const int n0 = (int)losses8[6], n1 = (int)losses8[7];
    float s0 = 0.0f;
    for (int j = sub; j < n1; j += GR_SUB) {      // label-1 rows = sources of the last n1 columns, in order
      const int ra = pos_meta[row0 + n0 + j] >> 2;
      s0 = fmaf(dz[(int64_t)ra * D + dd], nbar[(int64_t)ra * D + d2], s0);
    }
    g = s0; pidx = L.o_fc() + u;
  }
  red[sub][threadIdx.x] = g;
  __syncthreads();
  if (sub != 0 || (MODE != 2 && pidx < 0)) return;
  g = 0.0f;
#pragma unroll
  for (int k = 0; k < GR_SUB; ++k) g += red[k][threadIdx.x];          // fixed order
  if (pidx >= 0) grads[pidx] = g;
  if (MODE == 1) adam_update(params, m, v, L, pidx, g, wd, sc[0], sc[1]);
  if (MODE == 2) {
    if (pidx < 0) return;
    const float s = xchg_sum(X, xstep, pidx, g);
    adam_update(params, m, v, L, pidx, s * grad_scale, wd, sc[0], sc[1]);
  }
}

__global__ void __launch_bounds__(256) k_adam(float *__restrict__ params, float *__restrict__ m, float *__restrict__ v,
                                              const float *__restrict__ grads, ParamLayout L, float lr, float wd,
                                              float grad_scale, const int32_t *__restrict__ step_counter) {
  __shared__ float sc[2];
  adam_scalars(sc, step_counter, lr);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L.n_train()) return;
  adam_update(params, m, v, L, i, grads[i] * grad_scale, wd, sc[0], sc[1]);
}

// The exchange + Adam on an existing packed gradient block (self-test of a connection, tests): see xchg_sum.
__global__ void __launch_bounds__(256) k_xchg_adam(float *__restrict__ params, float *__restrict__ m, float *__restrict__ v,
                                                   const float *__restrict__ grads, ParamLayout L, float lr, float wd,
                                                   float grad_scale, const int32_t *__restrict__ step_counter, ggad_xchg_view X,
                                                   uint32_t step) {
  __shared__ float sc[2];
  adam_scalars(sc, step_counter, lr);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L.n_train()) return;
  const float s = xchg_sum(X, step, i, grads[i]);
  adam_update(params, m, v, L, i, s * grad_scale, wd, sc[0], sc[1]);
}

__global__ void __launch_bounds__(256) k_params_sync(float *__restrict__ params, ParamLayout L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int D = L.D, F = L.F;
  if (i >= L.o_W() && i < L.o_fc()) {
    const int u = i - L.o_W(); const int d = u / F, f = u - d * F;
    params[L.o_Wt() + f * D + d] = params[i];
  } else if (i >= L.o_fc() && i < L.n_train()) {
    const int u = i - L.o_fc(); const int d = u / D, d2 = u - d * D;
    params[L.o_fcT() + d2 * D + d] = params[i];
  }
}

__global__ void __launch_bounds__(256) k_score(const float *__restrict__ params, ParamLayout L, const float *__restrict__ x1,
                                               int n_rows, float *__restrict__ prob) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = L.D, F = L.F;
  const float *Wt = params + L.o_Wt();
  for (int i = threadIdx.x; i < F * D; i += blockDim.x) lds[i] = Wt[i];
  __syncthreads();
  const int lane = lane_id(), d = lane < D ? lane : D - 1;
  const float wd = lane < D ? params[lane] : 0.0f;
  const int wpb = blockDim.x / 64;
  for (int row = blockIdx.x * wpb + threadIdx.x / 64; row < n_rows; row += gridDim.x * wpb) {
    const float *x = x1 + (int64_t)row * F;
    float acc = 0.0f;
    for (int f = 0; f < F; ++f) acc = fmaf(lds[f * D + d], x[f], acc);
    const float s = wave_sum_fast(wd * fmaxf(acc, 0.0f));
    if (lane == 0) prob[row] = 1.0f / (1.0f + expf(-s));     // torch.sigmoid            graphsage.py:180
  }
}

// h[row][d] = relu(W x1[row]) for every row (inference embeddings, GCNEncoder.forward with train_flag False)
__global__ void __launch_bounds__(256) k_encode(const float *__restrict__ params, ParamLayout L, const float *__restrict__ x1,
                                                int n_rows, float *__restrict__ h) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = L.D, F = L.F;
  const float *Wt = params + L.o_Wt();
  for (int i = threadIdx.x; i < F * D; i += blockDim.x) lds[i] = Wt[i];
  __syncthreads();
  const int lane = lane_id(), d = lane < D ? lane : D - 1;
  const int wpb = blockDim.x / 64;
  for (int row = blockIdx.x * wpb + threadIdx.x / 64; row < n_rows; row += gridDim.x * wpb) {
    const float *x = x1 + (int64_t)row * F;
    float acc = 0.0f;
    for (int f = 0; f < F; ++f) acc = fmaf(lds[f * D + d], x[f], acc);
    if (lane < D) h[(int64_t)row * D + lane] = fmaxf(acc, 0.0f);                  // graphsage.py:412
  }
}

// ------------------------------------------------------------------ forward rows with the projection fused in
// k_project + k_fwd_rows in one launch (one launch boundary and one 8 us latency-bound kernel less per step).  The
// projection h2 = relu(W x2[own(e)]) is computed by the workgroup of the entry's row: the x2 rows are fetched with
// VECTOR loads, 3 rows per instruction (lane = (row slot g, feature f)), four instructions in flight, and the 17
// features of a row reach every channel lane through v_readlane (SGPR broadcast) (a scalar-load version has
// one dependent scalar-cache round trip per 4 entries and measured 2x slower on hub rows).
// h2 is written per ENTRY (not per owner) for the relu mask of k_bwd_flat.  Same fma / summation order as
// k_project + k_fwd_rows: bit-identical h1 / nbar / gen.
template <int FT>
__global__ void __launch_bounds__(FWDV_NW * 64) k_fwd_rows_v(const float *__restrict__ params, ParamLayout L,
                                                    const float *__restrict__ x1, const float *__restrict__ x2,
                                                    const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ ent_own,
                                                    const int32_t *__restrict__ labels, int row0, int ent0,
                                                    float *__restrict__ h2, float *__restrict__ h1,
                                                    float *__restrict__ nbar, float *__restrict__ gen) {
  static_assert(FT > 0 && FT <= 32, "register-resident W^T column, >= 2 rows per load");
  constexpr int RPI = 64 / FT;              // x2 rows per load instruction
  constexpr int U = 8;                      // load instructions in flight (one memory round trip per 24 entries)
  __shared__ float part[FWDV_NW][64];
  __shared__ float ns[64];
  const int D = L.D;
  const int lane = lane_id(), wid = threadIdx.x / 64, d = lane < D ? lane : D - 1;
  const int g = lane / FT, f = lane - g * FT;
  const bool ld_on = g < RPI;
  const int row = row0 + blockIdx.x;
  WCol<FT> W;
  W.load(params + L.o_Wt(), nullptr, D, FT, d, threadIdx.x, blockDim.x);
  const int e0 = ent_ptr[row], e1 = ent_ptr[row + 1];
  const int r = e1 - e0;
  float acc = 0.0f;
  for (int blk = wid; blk < r; blk += FWDV_NW * 64) {
    const int my = blk + FWDV_NW * lane;                 // lane l holds the owner of entry blk + 16 l of the row
    const int ov = (my < r) ? ent_own[e0 + my] : 0;
    const int cnt = min(64, (r - blk + FWDV_NW - 1) / FWDV_NW);
    for (int i0 = 0; i0 < cnt; i0 += RPI * U) {
      float xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int src = i0 + u * RPI + g;
        const int o = __shfl(ov, src & 63, GGAD_WAVE);
        xv[u] = (ld_on && src < cnt) ? x2[(int64_t)o * FT + f] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int gg = 0; gg < RPI; ++gg) {
          const int idx = i0 + u * RPI + gg;            // wave-uniform
          if (idx < cnt) {
            float h = 0.0f;
#pragma unroll
            for (int ff = 0; ff < FT; ++ff)
              h = fmaf(W.reg[ff], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[u]), gg * FT + ff)), h);
            h = fmaxf(h, 0.0f);                                                        // relu(W x2[u])   graphsage.py:419
            acc += h;
            if (lane < D) h2[(int64_t)(e0 - ent0 + blk + FWDV_NW * idx) * D + lane] = h;
          }
        }
      }
    }
  }
  part[wid][lane] = acc;
  __syncthreads();
  const float inv_r = 1.0f / (float)r;                                      // mask_row = mask / rowsum  graphsage.py:317
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < FWDV_NW; ++k) tot += part[k][lane];                    // fixed order
  const float nb = inv_r * tot;
  const int y = labels[row];
  if (wid == 0) {
    if (lane < D) nbar[(int64_t)row * D + lane] = nb;                       // mask_row.mm(...)          graphsage.py:421
    ns[lane] = (lane < D) ? nb : 0.0f;
  }
  if (wid == 1) {                                                           // h1 = relu(W x1[row])      graphsage.py:412
    const float h = fmaxf(W.dot(x1 + (int64_t)row * FT), 0.0f);
    if (lane < D) h1[(int64_t)row * D + lane] = h;
  }
  if (y != 1) return;                                                       // block-uniform exit
  __syncthreads();
  const float *fcT = params + L.o_fcT();                                    // gen = relu(fc nbar)   graphsage.py:428-430
  const int q = (D + FWDV_NW - 1) / FWDV_NW;
  float a = 0.0f;
  for (int d2 = wid * q; d2 < min(D, (wid + 1) * q); ++d2) a = fmaf(fcT[d2 * D + d], ns[d2], a);
  __syncthreads();
  part[wid][lane] = a;
  __syncthreads();
  if (wid == 0 && lane < D) {
    float gs = 0.0f;
#pragma unroll
    for (int k = 0; k < FWDV_NW; ++k) gs += part[k][lane];
    gen[(int64_t)row * D + lane] = fmaxf(gs, 0.0f);
  }
}

// ------------------------------------------------------------------ hub batches: chunk-parallel forward (2 launches)
// A batch with a hub row used to take k_project (owner-parallel) + k_fwd_rows (one 16-wave workgroup per row: the hub row's
// workgroup sums thousands of h2 rows) + k_loss_pos: 17.5 + 4.8 us.  Here the unit of work is a CHUNK of <= 16 consecutive
// entries of one row (tables built by the plan, ggad_mb_row_chunks):
//   k_fwd_chunks    one wave per chunk: partial sum of relu(W x2[own(e)]) -> chunk_part[c];  extra waves: h1 = relu(W x1)
//   k_loss_pos_ck   k_loss_pos, whose wave first sums the chunk partials of its row (nbar, stored) and, for a column that
//                   comes from a label-1 row, of that source row too (gen = relu(fc nbar), stored); fc^T staged in LDS
// h2 is not stored: k_bwd_flat recomputes the relu mask (same fma order, bit-identical) from the feature row it reads anyway.
// Row sums run in chunk order: deterministic; other order than k_fwd_rows (waves striding by 16), equal to fp32 round-off.
__global__ void __launch_bounds__(256) k_fwd_chunks(const float *__restrict__ params, ParamLayout L, const float *__restrict__ x1,
                                                    const float *__restrict__ x2, const int32_t *__restrict__ ent_own,
                                                    const int32_t *__restrict__ row_ck_ptr, const int32_t *__restrict__ ck_rc,
                                                    const int32_t *__restrict__ ck_e0, int row0, int n_rows,
                                                    float *__restrict__ h1, float *__restrict__ chunk_part) {
  constexpr int FT = 17, CH = 16;
  const int D = L.D;
  const int lane = lane_id(), wid = threadIdx.x / 64, d = lane < D ? lane : D - 1;
  const int fl = lane < FT ? lane : FT - 1;
  const int w = blockIdx.x * 4 + wid;
  const int ck0 = row_ck_ptr[row0], nck = row_ck_ptr[row0 + n_rows] - ck0;
  if (w >= nck + n_rows) return;
  float Wr[FT];
  const float *Wt = params + L.o_Wt();
#pragma unroll
  for (int f = 0; f < FT; ++f) Wr[f] = Wt[f * D + d];
  if (w < nck) {
    const int c = ck0 + w;
    const int cnt = ck_rc[c] & 63, e0 = ck_e0[c];
    const int ov = ent_own[e0 + (lane < cnt ? lane : 0)];
    float xv[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) xv[k] = x2[(int64_t)__builtin_amdgcn_readlane(ov, k < cnt ? k : 0) * FT + fl];
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (k < cnt) {
        float h = 0.0f;
#pragma unroll
        for (int f = 0; f < FT; ++f) h = fmaf(Wr[f], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[k]), f)), h);
        acc += fmaxf(h, 0.0f);                                                        // relu(W x2[u])   graphsage.py:419
      }
    }
    chunk_part[(int64_t)c * 64 + lane] = acc;
  } else {                                                                            // h1 = relu(W x1[row])      graphsage.py:412
    const int row = row0 + (w - nck);
    const float xv = x1[(int64_t)row * FT + fl];
    float h = 0.0f;
#pragma unroll
    for (int f = 0; f < FT; ++f) h = fmaf(Wr[f], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), f)), h);
    if (lane < D) h1[(int64_t)row * D + lane] = fmaxf(h, 0.0f);
  }
}

__global__ void __launch_bounds__(256) k_loss_pos_ck(const float *__restrict__ params, ParamLayout L, const float *__restrict__ h1,
                                                     float *__restrict__ nbar, float *__restrict__ gen,
                                                     const int32_t *__restrict__ pos_meta, const int32_t *__restrict__ ent_ptr,
                                                     const int32_t *__restrict__ row_ck_ptr, const float *__restrict__ chunk_part,
                                                     int row0, int B, float *__restrict__ pos_scal, float *__restrict__ part) {
  __shared__ float red[4][8];
  __shared__ float fct[GGAD_MAX_D * (GGAD_MAX_D + 1)];           // fc^T, rows padded: conflict-free column reads
  const int D = L.D;
  const int lane = lane_id(), wid = threadIdx.x / 64;
  const bool on = lane < D;
  const int dl = on ? lane : D - 1;
  const int q = blockIdx.x * 4 + wid;
  for (int idx = threadIdx.x; idx < D * D; idx += 256) fct[(idx / D) * (GGAD_MAX_D + 1) + idx % D] = params[L.o_fcT() + idx];
  float o[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int meta = 0, src = row0, qa = 0, qb = 0, sa = 0, sb = 0, rq = 1, rs = 1;
  float wd_r = 0.f, hs_r = 0.f;
  const bool act = q < B;
  if (act) {
    meta = pos_meta[row0 + q];
    src = meta >> 2;
    const int row = row0 + q;
    qa = row_ck_ptr[row]; qb = row_ck_ptr[row + 1];
    rq = ent_ptr[row + 1] - ent_ptr[row];
    if (meta & 2) { sa = row_ck_ptr[src]; sb = row_ck_ptr[src + 1]; rs = ent_ptr[src + 1] - ent_ptr[src]; }
    wd_r = params[dl];
    hs_r = h1[(int64_t)src * D + dl];
  }
  float totq = 0.0f, tots = 0.0f;
  constexpr int PF = 16;                                         // a hub row of 2,000 entries has 125 partials: 8 rounds
  for (int c0 = 0; c0 < max(qb - qa, sb - sa); c0 += PF) {       // chunk order, 16 + 16 loads in flight from clamped indices
    float vq[PF], vs[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      vq[k] = chunk_part[(int64_t)min(qa + c0 + k, max(qb - 1, qa)) * 64 + lane];
      vs[k] = chunk_part[(int64_t)(sb > sa ? min(sa + c0 + k, sb - 1) : qa) * 64 + lane];
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      totq += (qa + c0 + k < qb) ? vq[k] : 0.0f;
      tots += (sa + c0 + k < sb) ? vs[k] : 0.0f;
    }
  }
  __syncthreads();                                               // fct staged
  if (act) {
    const int y = meta & 1;
    const bool from_gen = (meta & 2) != 0;
    const float nb_r = (1.0f / (float)rq) * totq;                                            // mask_row = mask / rowsum  graphsage.py:317
    if (on) nbar[(int64_t)(row0 + q) * D + lane] = nb_r;                                     // to_feats_neigh[q, :]
    float c_r = hs_r;                                                                        // combined_all[:, q] = h1[src] ...
    if (from_gen) {                                                                          // ... or gen[src] = relu(fc nbar[src])
      const float nbm = on ? (1.0f / (float)rs) * tots : 0.0f;
      float a = 0.0f;
      for (int d2 = 0; d2 < D; ++d2)
        a = fmaf(fct[d2 * (GGAD_MAX_D + 1) + dl], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nbm), d2)), a);
      c_r = fmaxf(a, 0.0f);
      if (on) gen[(int64_t)src * D + lane] = c_r;
    }
    const float wd = on ? wd_r : 0.0f, c = on ? c_r : 0.0f, nb = on ? nb_r : 0.0f;
    const float hs = (on && from_gen) ? hs_r : 0.0f;
    const PosVals v = eval_position(wd, c, nb);
    float recn = 0.0f;
    if (from_gen) { const float dl2 = hs - c; recn = sqrtf(wave_sum_fast(dl2 * dl2)); }      // recon2   graphsage.py:197-198
    o[0] = (1.0f - (float)y) * v.s - log_sigmoid(v.s);                   // BCEWithLogits, pos_weight 1 graphsage.py:246
    o[1] = y == 0 ? v.aff : 0.0f; o[2] = y == 1 ? v.aff : 0.0f; o[3] = recn;
    o[4] = y == 0 ? 1.0f : 0.0f;  o[5] = y == 1 ? 1.0f : 0.0f;
    if (lane == 0) {
      float *ps = pos_scal + (int64_t)q * 8;
      ps[0] = v.s; ps[1] = v.aff; ps[2] = v.na; ps[3] = v.nbn; ps[4] = recn;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red[wid][k] = o[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    part[(int64_t)blockIdx.x * 8 + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

bool dims_ok(int D, int F) { return D >= 1 && D <= GGAD_MAX_D && F >= 1 && (size_t)(4 * F * D + 512) * 4 <= 150 * 1024; }

}  // namespace

extern "C" {

int ggad_max_embed_dim(void) { return GGAD_MAX_D; }
int ggad_max_feat_dim(void) { return GGAD_MAX_F; }
int ggad_mb_bwd_parts(void) { return BWD_PARTS; }

int64_t ggad_mb_param_count(int32_t D, int32_t F) { return (int64_t)D + (int64_t)D * F + (int64_t)D * D; }
int64_t ggad_mb_param_block_elems(int32_t D, int32_t F) { return ggad_mb_param_count(D, F) + (int64_t)F * D + (int64_t)D * D; }

int ggad_mb_params_sync(float *params, int32_t D, int32_t F, ggad_stream_t stream) {
  GGAD_REQUIRE(params && dims_ok(D, F));
  ParamLayout L{D, F};
  k_params_sync<<<dim3((L.n_train() + 255) / 256), dim3(256), 0, as_stream(stream)>>>(params, L);
  GGAD_CHECK_LAUNCH("mb_params_sync");
  return GGAD_OK;
}

int ggad_mb_project(const float *params, int32_t D, int32_t F, const float *x2, const int32_t *ent_own, int32_t ent0,
                    int32_t n_ents, float *h2, ggad_stream_t stream) {
  GGAD_REQUIRE(params && x2 && ent_own && h2 && dims_ok(D, F) && ent0 >= 0 && n_ents >= 0);
  if (n_ents == 0) return GGAD_OK;
  ParamLayout L{D, F};
  const int blocks = (n_ents + 4 * PROJ_EPW - 1) / (4 * PROJ_EPW);
  if (F == 17)
    k_project<17><<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(params, L, x2, ent_own, ent0, n_ents, h2);
  else
    k_project<0><<<dim3(blocks), dim3(256), (size_t)F * D * 4, as_stream(stream)>>>(params, L, x2, ent_own, ent0, n_ents, h2);
  GGAD_CHECK_LAUNCH("mb_project");
  return GGAD_OK;
}

int ggad_mb_fwd_rows(const float *params, int32_t D, int32_t F, const float *x1, const float *h2, const int32_t *ent_ptr,
                     const int32_t *ent_own, const int32_t *labels, int32_t row0, int32_t n_rows, int32_t ent0, float *h1,
                     float *nbar, float *gen, ggad_stream_t stream) {
  GGAD_REQUIRE(params && x1 && h2 && ent_ptr && ent_own && labels && h1 && nbar && gen && dims_ok(D, F));
  GGAD_REQUIRE(n_rows >= 0 && row0 >= 0 && ent0 >= 0);
  if (n_rows == 0) return GGAD_OK;
  ParamLayout L{D, F};
  if (F == 17)
    k_fwd_rows<17><<<dim3(n_rows), dim3(FWD_NW * 64), (FWD_NW + 1) * 64 * 4, as_stream(stream)>>>(
        params, L, x1, h2, ent_ptr, ent_own, labels, row0, ent0, h1, nbar, gen);
  else
    k_fwd_rows<0><<<dim3(n_rows), dim3(FWD_NW * 64), (size_t)((FWD_NW + 1) * 64 + F * D) * 4, as_stream(stream)>>>(
        params, L, x1, h2, ent_ptr, ent_own, labels, row0, ent0, h1, nbar, gen);
  GGAD_CHECK_LAUNCH("mb_fwd_rows");
  return GGAD_OK;
}

int64_t ggad_mb_loss_workspace_elems(int32_t n_rows) {
  // pos_scal[n_rows][8] | part[nwg][8] | gw: [nwg][64] (sized for n_rows workgroups)
  return (int64_t)n_rows * 8 + (int64_t)loss_nwg(n_rows) * 8 + (int64_t)n_rows * 64;
}

int ggad_mb_loss(const float *params, int32_t D, int32_t F, const float *h1, const float *nbar, const float *gen,
                 const int32_t *labels, const int32_t *pos_meta, const int32_t *row_pos, const int32_t *ent_ptr,
                 int32_t row0, int32_t n_rows, float *loss_ws, float *losses8, float *d_h1, float *d_gen, float *d_nbar,
                 float *dz, float *coef_a, float *coef_g, int32_t *step_counter, ggad_stream_t stream) {
  GGAD_REQUIRE(params && h1 && nbar && gen && labels && pos_meta && row_pos && ent_ptr && loss_ws && losses8);
  GGAD_REQUIRE(dz && coef_a && coef_g);
  GGAD_REQUIRE((d_h1 == nullptr) == (d_gen == nullptr) && (d_h1 == nullptr) == (d_nbar == nullptr));
  GGAD_REQUIRE(dims_ok(D, F) && n_rows >= 1 && row0 >= 0);
  ParamLayout L{D, F};
  const int nwg = loss_nwg(n_rows);
  float *pos_scal = loss_ws, *part = loss_ws + (int64_t)n_rows * 8, *gw_part = part + (int64_t)nwg * 8;
  hipStream_t st = as_stream(stream);
  k_loss_pos<<<dim3(nwg), dim3(256), 0, st>>>(params, D, h1, nbar, gen, pos_meta, row0, n_rows, pos_scal, part);
  k_loss_rows<<<dim3(nwg), dim3(256), 0, st>>>(params, L, h1, nbar, gen, labels, pos_meta, row_pos, ent_ptr, row0, n_rows,
                                              pos_scal, part, gw_part, losses8, d_h1, d_gen, d_nbar, dz, coef_a, coef_g,
                                              step_counter);
  GGAD_CHECK_LAUNCH("mb_loss");
  return GGAD_OK;
}

int ggad_mb_row_coefs(const float *params, int32_t D, int32_t F, const int32_t *labels, const int32_t *ent_ptr,
                      int32_t row0, int32_t n_rows, const float *h1, const float *gen, const float *d_h1,
                      const float *d_gen, const float *d_nbar, float *dz, float *coef_a, float *coef_g,
                      ggad_stream_t stream) {
  GGAD_REQUIRE(params && labels && ent_ptr && h1 && gen && d_h1 && d_gen && d_nbar && dz && coef_a && coef_g);
  GGAD_REQUIRE(dims_ok(D, F) && n_rows >= 1 && row0 >= 0);
  ParamLayout L{D, F};
  k_row_coefs<<<dim3((n_rows + 3) / 4), dim3(256), 0, as_stream(stream)>>>(params, L, labels, ent_ptr, row0, n_rows, h1, gen,
                                                                          d_h1, d_gen, d_nbar, dz, coef_a, coef_g);
  GGAD_CHECK_LAUNCH("mb_row_coefs");
  return GGAD_OK;
}

static int bwd_flat_launch(int32_t D, int32_t F, const float *x1, const float *x2, const float *h2, const int32_t *ent_own,
                           const int32_t *ent_row, int32_t row0, int32_t n_rows, int32_t ent0, int32_t n_ents,
                           const float *coef_a, const float *coef_g, float *dw_part, int h2_by_entry, ggad_stream_t stream,
                           const float *Wt = nullptr);

int ggad_mb_bwd_flat(int32_t D, int32_t F, const float *x1, const float *x2, const float *h2, const int32_t *ent_own,
                     const int32_t *ent_row, int32_t row0, int32_t n_rows, int32_t ent0, int32_t n_ents,
                     const float *coef_a, const float *coef_g, float *dw_part, ggad_stream_t stream) {
  return bwd_flat_launch(D, F, x1, x2, h2, ent_own, ent_row, row0, n_rows, ent0, n_ents, coef_a, coef_g, dw_part, 0, stream);
}

static int bwd_flat_launch(int32_t D, int32_t F, const float *x1, const float *x2, const float *h2, const int32_t *ent_own,
                           const int32_t *ent_row, int32_t row0, int32_t n_rows, int32_t ent0, int32_t n_ents,
                           const float *coef_a, const float *coef_g, float *dw_part, int h2_by_entry, ggad_stream_t stream,
                           const float *Wt) {
  GGAD_REQUIRE(x1 && x2 && ent_own && ent_row && coef_a && coef_g && dw_part && dims_ok(D, F));
  GGAD_REQUIRE(h2_by_entry == 2 ? (Wt != nullptr && F == 17) : (h2 != nullptr));
  GGAD_REQUIRE(n_rows >= 1 && row0 >= 0 && ent0 >= 0 && n_ents >= 0);
  ParamLayout L{D, F};
  const size_t lds = (size_t)4 * F * D * 4;
  if (F == 17)
    k_bwd_flat<17><<<dim3(BWD_PARTS), dim3(256), lds, as_stream(stream)>>>(L, x1, x2, h2, ent_own, ent_row, row0, n_rows, ent0,
                                                                           n_ents, coef_a, coef_g, dw_part, h2_by_entry, Wt);
  else
    k_bwd_flat<0><<<dim3(BWD_PARTS), dim3(256), lds, as_stream(stream)>>>(L, x1, x2, h2, ent_own, ent_row, row0, n_rows, ent0,
                                                                          n_ents, coef_a, coef_g, dw_part, h2_by_entry, Wt);
  GGAD_CHECK_LAUNCH("mb_bwd_flat");
  return GGAD_OK;
}

int ggad_mb_grad_reduce(int32_t D, int32_t F, const int32_t *pos_meta, int32_t row0, int32_t n_rows,
                        const float *losses8, const float *nbar, const float *dw_part, const float *dz,
                        const float *loss_ws, float *grads, ggad_stream_t stream) {
  GGAD_REQUIRE(pos_meta && losses8 && nbar && dw_part && dz && loss_ws && grads && dims_ok(D, F) && n_rows >= 1);
  ParamLayout L{D, F};
  const int nwg = loss_nwg(n_rows);
  const float *gw_part = loss_ws + (int64_t)n_rows * 8 + (int64_t)nwg * 8;
  k_grad_reduce<0><<<dim3((L.n_train() + 63) / 64), dim3(64, GR_SUB), 0, as_stream(stream)>>>(
      L, pos_meta, row0, losses8, nbar, dw_part, BWD_PARTS, dz, gw_part, nwg, grads, nullptr, nullptr, nullptr, 0.f, 0.f,
      nullptr, ggad_xchg_view{}, 0u, 1.0f);
  GGAD_CHECK_LAUNCH("mb_grad_reduce");
  return GGAD_OK;
}

int ggad_mb_adam(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int32_t D, int32_t F,
                 float lr, float weight_decay, float grad_scale, const int32_t *step_counter,
                 ggad_stream_t stream) {
  GGAD_REQUIRE(params && exp_avg && exp_avg_sq && grads && step_counter && dims_ok(D, F));
  ParamLayout L{D, F};
  k_adam<<<dim3((L.n_train() + 255) / 256), dim3(256), 0, as_stream(stream)>>>(params, exp_avg, exp_avg_sq, grads, L, lr,
                                                                              weight_decay, grad_scale, step_counter);
  GGAD_CHECK_LAUNCH("mb_adam");
  return GGAD_OK;
}

int ggad_mb_score(const float *params, int32_t D, int32_t F, const float *x1, int32_t n_rows, float *prob,
                  ggad_stream_t stream) {
  GGAD_REQUIRE(params && x1 && prob && dims_ok(D, F) && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  ParamLayout L{D, F};
  const int blocks = (n_rows + 3) / 4 < 4096 ? (n_rows + 3) / 4 : 4096;
  k_score<<<dim3(blocks), dim3(256), (size_t)F * D * 4, as_stream(stream)>>>(params, L, x1, n_rows, prob);
  GGAD_CHECK_LAUNCH("mb_score");
  return GGAD_OK;
}

int ggad_mb_encode(const float *params, int32_t D, int32_t F, const float *x1, int32_t n_rows, float *h,
                   ggad_stream_t stream) {
  GGAD_REQUIRE(params && x1 && h && dims_ok(D, F) && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  ParamLayout L{D, F};
  const int blocks = (n_rows + 3) / 4 < 4096 ? (n_rows + 3) / 4 : 4096;
  k_encode<<<dim3(blocks), dim3(256), (size_t)F * D * 4, as_stream(stream)>>>(params, L, x1, n_rows, h);
  GGAD_CHECK_LAUNCH("mb_encode");
  return GGAD_OK;
}

/* One whole training step for one batch.  chain 0 (default): fwd_rows_v (projection fused, F == 17; else project ->
 * fwd_rows) -> loss_pos -> loss_rows -> bwd_flat -> grad_reduce; chain 2: always project -> fwd_rows -> ...
 * (the generic layered chain).
 * Adam is fused into the last launch when fuse_adam != 0; otherwise the caller all-reduces s->grads and calls
 * ggad_mb_adam. */
int64_t ggad_mb_dw_part_elems(int32_t n_rows, int32_t D, int32_t F) {
  return (int64_t)(n_rows > BWD_PARTS ? n_rows : BWD_PARTS) * F * D;
}

static int train_step_impl(const ggad_mb_step *s, int32_t fuse_adam, const ggad_xchg_view *xv, uint32_t xstep, float grad_scale,
                           ggad_stream_t stream);

int ggad_mb_train_step(const ggad_mb_step *s, int32_t fuse_adam, ggad_stream_t stream) {
  return train_step_impl(s, fuse_adam ? 1 : 0, nullptr, 0u, 1.0f, stream);
}

// fuse_adam 0: gradients only; 1: Adam in the last launch; 2: one-shot exchange + Adam in the last launch (xv, xstep)
static int train_step_impl(const ggad_mb_step *s, int32_t fuse_adam, const ggad_xchg_view *xv, uint32_t xstep, float grad_scale,
                           ggad_stream_t stream) {
  GGAD_REQUIRE(s && s->params && s->exp_avg && s->exp_avg_sq && s->grads && s->step_counter);
  GGAD_REQUIRE(s->chain == 0 || s->chain == 2);
  const int D = s->D, F = s->F;
  int rc;
  // 5 launches: the projection is done by the forward-rows kernel -- unless the batch holds a hub row (one workgroup would
  // project thousands of entries while the flat k_project spreads them over the chip)
  static const int fuse_max_row = [] { const char *e = getenv("GGAD_FUSE_MAX_ROW"); return e ? atoi(e) : 256; }();
  const bool fused_fwd = s->chain == 0 && F == 17 && s->max_row_entries > 0 && s->max_row_entries <= fuse_max_row;
  // hub batches: chunk-parallel forward + position kernel with the row sums folded in (2 launches instead of 3)
  static const bool chunk_path_on = [] { const char *e = getenv("GGAD_CHUNK_FWD"); return !e || atoi(e) != 0; }();
  const bool chunk_fwd = chunk_path_on && s->chain == 0 && F == 17 && !fused_fwd && s->row_ck_ptr && s->ck_rc && s->ck_e0 &&
                         s->chunk_part;
  if (chunk_fwd) {
    GGAD_REQUIRE(s->x1 && s->x2 && s->ent_ptr && s->ent_own && s->ent_row && s->labels && s->pos_meta && s->row_pos && s->h1 &&
                 s->nbar && s->gen && s->dz && s->coef_a && s->coef_g && s->dw_part && s->loss_ws && s->losses8 && dims_ok(D, F));
    GGAD_REQUIRE(s->n_rows >= 1 && s->row0 >= 0 && s->ent0 >= 0 && s->n_ents >= 0);
    ParamLayout L{D, F};
    hipStream_t st = as_stream(stream);
    const int cl = 16;
    const int max_waves = (s->n_ents + cl - 1) / cl + 2 * s->n_rows;        // >= chunks of the batch + its rows
    k_fwd_chunks<<<dim3((max_waves + 3) / 4), dim3(256), 0, st>>>(s->params, L, s->x1, s->x2, s->ent_own, s->row_ck_ptr, s->ck_rc,
                                                                 s->ck_e0, s->row0, s->n_rows, s->h1, s->chunk_part);
    const int nwg = loss_nwg(s->n_rows);
    float *pos_scal = s->loss_ws, *part = s->loss_ws + (int64_t)s->n_rows * 8, *gw_part = part + (int64_t)nwg * 8;
    k_loss_pos_ck<<<dim3(nwg), dim3(256), 0, st>>>(s->params, L, s->h1, s->nbar, s->gen, s->pos_meta, s->ent_ptr, s->row_ck_ptr,
                                                  s->chunk_part, s->row0, s->n_rows, pos_scal, part);
    k_loss_rows<<<dim3(nwg), dim3(256), 0, st>>>(s->params, L, s->h1, s->nbar, s->gen, s->labels, s->pos_meta, s->row_pos,
                                                s->ent_ptr, s->row0, s->n_rows, pos_scal, part, gw_part, s->losses8, nullptr, nullptr,
                                                nullptr, s->dz, s->coef_a, s->coef_g, s->step_counter);
    GGAD_CHECK_LAUNCH("mb_train_step (chunk forward)");
    if ((rc = bwd_flat_launch(D, F, s->x1, s->x2, nullptr, s->ent_own, s->ent_row, s->row0, s->n_rows, s->ent0, s->n_ents,
                              s->coef_a, s->coef_g, s->dw_part, 2, stream, s->params + L.o_Wt()))) return rc;
  } else if (fused_fwd) {
    GGAD_REQUIRE(s->x1 && s->x2 && s->h2 && s->ent_ptr && s->ent_own && s->labels && s->h1 && s->nbar && s->gen && dims_ok(D, F));
    GGAD_REQUIRE(s->n_rows >= 1 && s->row0 >= 0 && s->ent0 >= 0);
    ParamLayout L{D, F};
    k_fwd_rows_v<17><<<dim3(s->n_rows), dim3(FWDV_NW * 64), 0, as_stream(stream)>>>(s->params, L, s->x1, s->x2, s->ent_ptr, s->ent_own,
                                                                                  s->labels, s->row0, s->ent0, s->h2, s->h1, s->nbar,
                                                                                  s->gen);
  } else {
    if ((rc = ggad_mb_project(s->params, D, F, s->x2, s->ent_own, s->ent0, s->n_ents, s->h2, stream))) return rc;
    if ((rc = ggad_mb_fwd_rows(s->params, D, F, s->x1, s->h2, s->ent_ptr, s->ent_own, s->labels, s->row0, s->n_rows, s->ent0,
                               s->h1, s->nbar, s->gen, stream))) return rc;
  }
  if (!chunk_fwd) {
    if ((rc = ggad_mb_loss(s->params, D, F, s->h1, s->nbar, s->gen, s->labels, s->pos_meta, s->row_pos, s->ent_ptr, s->row0,
                           s->n_rows, s->loss_ws, s->losses8, nullptr, nullptr, nullptr, s->dz, s->coef_a, s->coef_g,
                           s->step_counter, stream))) return rc;
    if ((rc = bwd_flat_launch(D, F, s->x1, s->x2, s->h2, s->ent_own, s->ent_row, s->row0, s->n_rows, s->ent0, s->n_ents,
                              s->coef_a, s->coef_g, s->dw_part, fused_fwd ? 1 : 0, stream))) return rc;
  }
  if (!fuse_adam)
    return ggad_mb_grad_reduce(D, F, s->pos_meta, s->row0, s->n_rows, s->losses8, s->nbar, s->dw_part, s->dz, s->loss_ws,
                               s->grads, stream);
  ParamLayout L{D, F};
  const int nwg = loss_nwg(s->n_rows);
  const float *gw_part = s->loss_ws + (int64_t)s->n_rows * 8 + (int64_t)nwg * 8;
  if (fuse_adam == 2) {
    GGAD_REQUIRE(xv != nullptr);
    k_grad_reduce<2><<<dim3((L.n_train() + 63) / 64), dim3(64, GR_SUB), 0, as_stream(stream)>>>(
        L, s->pos_meta, s->row0, s->losses8, s->nbar, s->dw_part, BWD_PARTS, s->dz, gw_part, nwg, s->grads, s->params,
        s->exp_avg, s->exp_avg_sq, s->lr, s->weight_decay, s->step_counter, *xv, xstep, grad_scale);
  } else {
    k_grad_reduce<1><<<dim3((L.n_train() + 63) / 64), dim3(64, GR_SUB), 0, as_stream(stream)>>>(
        L, s->pos_meta, s->row0, s->losses8, s->nbar, s->dw_part, BWD_PARTS, s->dz, gw_part, nwg, s->grads, s->params,
        s->exp_avg, s->exp_avg_sq, s->lr, s->weight_decay, s->step_counter, ggad_xchg_view{}, 0u, 1.0f);
  }
  GGAD_CHECK_LAUNCH("mb_train_step");
  return GGAD_OK;
}

/* The dense steps of a whole chunk in ONE host call: step b = batch b of the chunk, rows [batch_ptr[b], batch_ptr[b+1]),
 * entries [batch_ent_ptr[b], batch_ent_ptr[b+1]) (HOST arrays of n_batches + 1 offsets), loss record of step b at
 * loss_log + 8 * (log_base + b) (device), largest row of batch b = batch_max_row[b] (host, optional).  Every other member
 * comes from *tmpl.  The host cost per step drops to the six
 * launches themselves, so the launch queue stays ahead of the 5-8 us kernels. */
int ggad_mb_train_chunk(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr, const int64_t *batch_ent_ptr,
                        const int32_t *batch_max_row, float *loss_log, int32_t log_base, int32_t fuse_adam,
                        ggad_stream_t stream) {
  GGAD_REQUIRE(tmpl && batch_ptr && batch_ent_ptr && loss_log && n_batches >= 0 && log_base >= 0);
  for (int b = 0; b < n_batches; ++b) {
    ggad_mb_step s = *tmpl;
    s.row0 = batch_ptr[b];
    s.n_rows = batch_ptr[b + 1] - batch_ptr[b];
    s.ent0 = (int32_t)batch_ent_ptr[b];
    s.n_ents = (int32_t)(batch_ent_ptr[b + 1] - batch_ent_ptr[b]);
    s.losses8 = loss_log + (int64_t)8 * (log_base + b);
    s.max_row_entries = batch_max_row ? batch_max_row[b] : 0;
    const int rc = ggad_mb_train_step(&s, fuse_adam, stream);
    if (rc) return rc;
  }
  return GGAD_OK;
}

/* Data-parallel variant of ggad_mb_train_chunk: per batch  backward (grads) -> exchange(user) -> Adam(grad_scale).
 * `exchange` is the caller's all-reduce of tmpl->grads on `stream` (e.g. torch.distributed / RCCL); it returns 0 on success.
 * Keeps the per-step host work to the launches plus that one callback. */
int ggad_mb_train_chunk_dp(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr, const int64_t *batch_ent_ptr,
                           const int32_t *batch_max_row, float *loss_log, int32_t log_base, float grad_scale,
                           int (*exchange)(void *), void *user, ggad_stream_t stream) {
  GGAD_REQUIRE(tmpl && batch_ptr && batch_ent_ptr && loss_log && exchange && n_batches >= 0 && log_base >= 0);
  for (int b = 0; b < n_batches; ++b) {
    ggad_mb_step s = *tmpl;
    s.row0 = batch_ptr[b];
    s.n_rows = batch_ptr[b + 1] - batch_ptr[b];
    s.ent0 = (int32_t)batch_ent_ptr[b];
    s.n_ents = (int32_t)(batch_ent_ptr[b + 1] - batch_ent_ptr[b]);
    s.losses8 = loss_log + (int64_t)8 * (log_base + b);
    s.max_row_entries = batch_max_row ? batch_max_row[b] : 0;
    int rc = ggad_mb_train_step(&s, 0, stream);
    if (rc) return rc;
    if (exchange(user) != 0) return GGAD_E_LAUNCH;
    rc = ggad_mb_adam(s.params, s.exp_avg, s.exp_avg_sq, s.grads, s.D, s.F, s.lr, s.weight_decay, grad_scale, s.step_counter,
                      stream);
    if (rc) return rc;
  }
  return GGAD_OK;
}

/* Data-parallel form without any host call between the launches: per batch  backward (packed gradients) -> k_xchg_adam
 * (publish to all peers' buffers, wait for theirs, sum in rank order, Adam with grad_scale).  Every rank calls it with the same
 * number of batches.  xchg: handle of ggad_xchg_create / ggad_xchg_connect.  The exchange runs INSIDE the gradient-reduce launch
 * (k_grad_reduce<2>: its workgroup of 64 parameters is the publishing unit), so the data-parallel step has the launch count of the
 * single-GPU step. */
int ggad_mb_train_chunk_xchg(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr, const int64_t *batch_ent_ptr,
                             const int32_t *batch_max_row, float *loss_log, int32_t log_base, float grad_scale, ggad_xchg *xchg,
                             ggad_stream_t stream) {
  GGAD_REQUIRE(tmpl && batch_ptr && batch_ent_ptr && loss_log && xchg && n_batches >= 0 && log_base >= 0);
  ParamLayout L{tmpl->D, tmpl->F};
  GGAD_REQUIRE(xchg->view.n >= L.n_train());
  for (int q = 0; q < xchg->view.world; ++q) GGAD_REQUIRE(xchg->view.peer[q] != nullptr);
  for (int b = 0; b < n_batches; ++b) {
    ggad_mb_step s = *tmpl;
    s.row0 = batch_ptr[b];
    s.n_rows = batch_ptr[b + 1] - batch_ptr[b];
    s.ent0 = (int32_t)batch_ent_ptr[b];
    s.n_ents = (int32_t)(batch_ent_ptr[b + 1] - batch_ent_ptr[b]);
    s.losses8 = loss_log + (int64_t)8 * (log_base + b);
    s.max_row_entries = batch_max_row ? batch_max_row[b] : 0;
    xchg->step += 1;
    const int rc = train_step_impl(&s, 2, &xchg->view, xchg->step, grad_scale, stream);     // exchange inside the reduce launch
    if (rc) return rc;
  }
  return GGAD_OK;
}

/* One exchange + Adam on the caller's packed gradient block (tests, self-test of a new connection). */
int ggad_xchg_adam(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int32_t D, int32_t F, float lr,
                   float weight_decay, float grad_scale, const int32_t *step_counter, ggad_xchg *xchg, ggad_stream_t stream) {
  GGAD_REQUIRE(params && exp_avg && exp_avg_sq && grads && step_counter && xchg && dims_ok(D, F));
  ParamLayout L{D, F};
  GGAD_REQUIRE(xchg->view.n >= L.n_train());
  for (int q = 0; q < xchg->view.world; ++q) GGAD_REQUIRE(xchg->view.peer[q] != nullptr);
  xchg->step += 1;
  k_xchg_adam<<<dim3((L.n_train() + 255) / 256), dim3(256), 0, as_stream(stream)>>>(params, exp_avg, exp_avg_sq, grads, L, lr,
                                                                                    weight_decay, grad_scale, step_counter,
                                                                                    xchg->view, xchg->step);
  GGAD_CHECK_LAUNCH("xchg_adam");
  return GGAD_OK;
}

}  // extern "C"
