// Dense per-batch step of the DGraph mini-batch path (gfx950): GCNEncoder.forward + GCN.loss
// (src/graphsage.py:395-454, 171-258), its backward and Adam (src/model_handler.py:363-364).
//
// Shapes are tiny (B = 200 rows, ~4k neighbourhood entries, F = 17, D = 64, 5,248 trainable
// scalars) and every step depends on the previous one through the weights, so these kernels are
// latency-bound, not bandwidth-bound (SURVEY.md §7 "hard parts").  Design: one wave per batch row,
// lane = embedding channel d (D <= 64), W^T resident in LDS (or in registers when F is a compile
// time constant), neighbour rows fetched through the scalar cache (wave-uniform addresses), relu
// masks recomputed in the backward instead of stored.  Four launches per step:
//   fwd_rows -> loss (one workgroup) -> bwd_rows -> grad_reduce [-> all-reduce] -> adam
// The projection uses the f32 VALU: at K = 17 an f32 MFMA tile (32x32x2) has the same FLOP rate
// and would waste 32/17 of it on padding; MFMA is used for the wide full-graph projections only.
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

// h_d = sum_f Wt[f][d] * x[f] for a wave-uniform row x (scalar-cache loads), Wt in LDS.
__device__ __forceinline__ float project_lds(const float *__restrict__ wt_lds, int D, int F, int d,
                                             const float *__restrict__ x) {
  float acc = 0.0f;
  for (int f = 0; f < F; ++f) acc = fmaf(wt_lds[f * D + d], x[f], acc);
  return acc;
}

template <int FT>
__device__ __forceinline__ float project_reg(const float (&wreg)[FT > 0 ? FT : 1], const float *__restrict__ x) {
  float acc = 0.0f;
#pragma unroll
  for (int f = 0; f < FT; ++f) acc = fmaf(wreg[f], x[f], acc);
  return acc;
}

// ------------------------------------------------------------------ forward rows
template <int FT>
__global__ void __launch_bounds__(64) k_fwd_rows(const float *__restrict__ params, ParamLayout L,
                                                 const float *__restrict__ x1, const float *__restrict__ x2,
                                                 const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ ent_own,
                                                 const int32_t *__restrict__ labels, int row0, int train,
                                                 float *__restrict__ h1, float *__restrict__ nbar, float *__restrict__ gen) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = L.D, F = (FT > 0) ? FT : L.F;
  const int lane = threadIdx.x, d = lane < D ? lane : D - 1;
  const int row = row0 + blockIdx.x;
  float *wt_lds = lds;            // F*D   (unused when FT > 0)
  float *ns = lds + F * D;        // D
  const float *Wt = params + L.o_Wt();
  float wreg[FT > 0 ? FT : 1];
  if constexpr (FT > 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f) wreg[f] = Wt[f * D + d];
  } else {
    for (int i = lane; i < F * D; i += 64) wt_lds[i] = Wt[i];
    __syncthreads();
  }
  auto project = [&](const float *__restrict__ x) -> float {
    if constexpr (FT > 0) return project_reg<FT>(wreg, x);
    else return project_lds(wt_lds, D, F, d, x);
  };
  const float *xr = x1 + (int64_t)row * F;
  const float h = fmaxf(project(xr), 0.0f);                          // graphsage.py:412
  if (lane < D) h1[(int64_t)row * D + lane] = h;
  if (!train) return;
  const int e0 = ent_ptr[row], e1 = ent_ptr[row + 1];
  const int r = e1 - e0;
  const float inv_r = 1.0f / (float)r;                                // mask_row = mask / rowsum   graphsage.py:317
  float nb = 0.0f;
  for (int blk = 0; blk < r; blk += 64) {
    const int ov = (blk + lane < r) ? ent_own[e0 + blk + lane] : 0;
    const int cnt = min(64, r - blk);
    for (int i = 0; i < cnt; ++i) {
      const int o = __builtin_amdgcn_readlane(ov, i);
      const float he = fmaxf(project(x2 + (int64_t)o * F), 0.0f);     // relu(W x2[u])              graphsage.py:419
      nb = fmaf(inv_r, he, nb);                                       // mask_row.mm(...)           graphsage.py:421
    }
  }
  if (lane < D) nbar[(int64_t)row * D + lane] = nb;
  if (labels[row] == 1) {                                             // outlier generation         graphsage.py:428-430
    if (lane < D) ns[lane] = nb;
    __syncthreads();
    const float *fcT = params + L.o_fcT();
    float acc = 0.0f;
    for (int d2 = 0; d2 < D; ++d2) acc = fmaf(fcT[d2 * D + d], ns[d2], acc);
    if (lane < D) gen[(int64_t)row * D + lane] = fmaxf(acc, 0.0f);
  }
}

// ------------------------------------------------------------------ loss (one workgroup per batch)
constexpr int LOSS_T = 1024;
constexpr int LOSS_W = LOSS_T / 64;

__device__ __forceinline__ float log_sigmoid(float x) {
  return fminf(x, 0.0f) - log1pf(expf(-fabsf(x)));
}

struct PosVals { float s, aff, na, nbn, nac, nbc, c, nb; int y; int src; };

__device__ __forceinline__ PosVals eval_position(const float *__restrict__ w, int D, const float *__restrict__ h1,
                                                 const float *__restrict__ nbar, const float *__restrict__ gen,
                                                 const int32_t *__restrict__ labels, const int32_t *__restrict__ src_of_pos,
                                                 int row0, int q, int lane) {
  PosVals v;
  const int prow = row0 + q;
  v.src = src_of_pos[prow];
  v.y = labels[prow];
  const bool on = lane < D;
  const bool from_gen = labels[v.src] == 1;
  v.c = on ? (from_gen ? gen[(int64_t)v.src * D + lane] : h1[(int64_t)v.src * D + lane]) : 0.0f;   // combined_all[:, q]
  v.nb = on ? nbar[(int64_t)prow * D + lane] : 0.0f;                                              // to_feats_neigh[q, :]
  const float wd = on ? w[lane] : 0.0f;
  v.s = wave_sum(wd * v.c);                                           // scores = weight.mm(embeds)  graphsage.py:174
  v.na = sqrtf(wave_sum(v.c * v.c));
  v.nbn = sqrtf(wave_sum(v.nb * v.nb));
  v.nac = fmaxf(v.na, 1e-8f);                                         // cosine_similarity eps      graphsage.py:234
  v.nbc = fmaxf(v.nbn, 1e-8f);
  v.aff = wave_sum((v.c / v.nac) * (v.nb / v.nbc));
  return v;
}

__global__ void __launch_bounds__(LOSS_T) k_loss(const float *__restrict__ params, int D, const float *__restrict__ h1,
                                                 const float *__restrict__ nbar, const float *__restrict__ gen,
                                                 const int32_t *__restrict__ labels, const int32_t *__restrict__ src_of_pos,
                                                 int row0, int B, float *__restrict__ losses8, float *__restrict__ d_h1,
                                                 float *__restrict__ d_gen, float *__restrict__ d_nbar,
                                                 float *__restrict__ grad_w, int32_t *__restrict__ step_counter) {
  __shared__ float red[LOSS_W][6];
  __shared__ float bc[8];
  __shared__ float gw[LOSS_W][64];
  const int lane = lane_id(), wid = threadIdx.x / 64;
  const float *w = params;   // o_w = 0
  float s_bce = 0.f, s_a0 = 0.f, s_a1 = 0.f, s_rec = 0.f; int n0 = 0, n1 = 0;
  for (int q = wid; q < B; q += LOSS_W) {
    const PosVals v = eval_position(w, D, h1, nbar, gen, labels, src_of_pos, row0, q, lane);
    s_bce += (1.0f - (float)v.y) * v.s - log_sigmoid(v.s);            // BCEWithLogits, pos_weight 1 graphsage.py:246
    if (v.y == 0) { s_a0 += v.aff; n0++; } else { s_a1 += v.aff; n1++; }
    if (v.y == 1) {                                                   // recon2 over label-1 rows    graphsage.py:197-198
      const int prow = row0 + q;
      const float dl = (lane < D) ? h1[(int64_t)prow * D + lane] - gen[(int64_t)prow * D + lane] : 0.0f;
      s_rec += sqrtf(wave_sum(dl * dl));
    }
  }
  if (lane == 0) {
    red[wid][0] = s_bce; red[wid][1] = s_a0; red[wid][2] = s_a1; red[wid][3] = s_rec;
    red[wid][4] = (float)n0; red[wid][5] = (float)n1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < LOSS_W; ++k)
      for (int m = 0; m < 6; ++m) t[m] += red[k][m];
    const float cls = t[0] / (float)B;
    const float an = t[1] / t[4], ab = t[2] / t[5];
    const float m = 1.0f - (an - ab);                                 // confidence_margin = 1      graphsage.py:236-240
    const float margin = fmaxf(m, 0.0f);
    const float rec = t[3] / t[5];
    losses8[0] = cls + margin + 0.1f * rec;                           // graphsage.py:258
    losses8[1] = cls; losses8[2] = margin; losses8[3] = rec;
    const float active = (m >= 0.0f) ? 1.0f : 0.0f;                   // clamp_min backward: pass where x >= min
    losses8[4] = 0.1f / t[5];                                         // d total / d rec_i
    losses8[5] = active; losses8[6] = t[4]; losses8[7] = t[5];
    bc[0] = active; bc[1] = t[4]; bc[2] = t[5]; bc[3] = 0.1f / t[5];
    if (step_counter) *step_counter += 1;
  }
  __syncthreads();
  const float active = bc[0], fn0 = bc[1], fn1 = bc[2], rec_coef = bc[3];
  float gacc = 0.0f;
  for (int q = wid; q < B; q += LOSS_W) {
    const PosVals v = eval_position(w, D, h1, nbar, gen, labels, src_of_pos, row0, q, lane);
    const float ds = (1.0f / (1.0f + expf(-v.s)) - (float)v.y) / (float)B;
    const float gq = active * (v.y == 0 ? -1.0f / fn0 : 1.0f / fn1);
    // aff = sum (c/nac)(nb/nbc); the eps clamp is applied outside autograd (torch clamps a detached copy),
    // so d aff / d c = (nb/nbc)/nac - (aff/nac) * c/|c|
    const float ca = v.na > 0.0f ? v.c / v.na : 0.0f;
    const float cb = v.nbn > 0.0f ? v.nb / v.nbn : 0.0f;
    const float dC = ds * (lane < D ? w[lane] : 0.0f) + gq * ((v.nb / v.nbc) / v.nac - (v.aff / v.nac) * ca);
    const float dN = gq * ((v.c / v.nac) / v.nbc - (v.aff / v.nbc) * cb);
    // column q of combined_all is h1[src] (label-0 source) or the generated outlier gen[src] (label-1 source);
    // a label-1 source row also carries the recon term 0.1 * mean_i |h1_i - gen_i|      graphsage.py:197-198,258
    float gH = dC, gG = 0.0f;
    if (labels[v.src] == 1) {
      const float hs = (lane < D) ? h1[(int64_t)v.src * D + lane] : 0.0f;
      const float dl = hs - v.c;
      const float nrm = sqrtf(wave_sum(dl * dl));
      const float t = rec_coef * (dl / nrm);
      gH = t;
      gG = dC - t;
    }
    if (lane < D) {
      d_h1[(int64_t)v.src * D + lane] = gH;
      d_gen[(int64_t)v.src * D + lane] = gG;
      d_nbar[(int64_t)(row0 + q) * D + lane] = dN;
    }
    gacc = fmaf(ds, v.c, gacc);
  }
  gw[wid][lane] = gacc;
  __syncthreads();
  if (threadIdx.x < D) {
    float t = 0.0f;
    for (int k = 0; k < LOSS_W; ++k) t += gw[k][threadIdx.x];
    grad_w[threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------ backward rows
template <int FT>
__global__ void __launch_bounds__(64) k_bwd_rows(const float *__restrict__ params, ParamLayout L,
                                                 const float *__restrict__ x1, const float *__restrict__ x2,
                                                 const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ ent_own,
                                                 const int32_t *__restrict__ labels, int row0,
                                                 const float *__restrict__ h1, const float *__restrict__ nbar,
                                                 const float *__restrict__ gen, const float *__restrict__ d_h1,
                                                 const float *__restrict__ d_gen, const float *__restrict__ d_nbar,
                                                 float *__restrict__ dw_part, float *__restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = L.D, F = (FT > 0) ? FT : L.F;
  const int lane = threadIdx.x, d = lane < D ? lane : D - 1;
  const bool on = lane < D;
  const int row = row0 + blockIdx.x;
  float *wt_lds = lds;              // F*D  (FT == 0)
  float *acc_lds = lds + F * D;     // F*D  (FT == 0)
  float *zs = lds + 2 * F * D;      // D
  const float *Wt = params + L.o_Wt();
  float wreg[FT > 0 ? FT : 1];
  float acc[FT > 0 ? FT : 1];
  if constexpr (FT > 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f) { wreg[f] = Wt[f * D + d]; acc[f] = 0.0f; }
  } else {
    for (int i = lane; i < F * D; i += 64) { wt_lds[i] = Wt[i]; acc_lds[i] = 0.0f; }
    __syncthreads();
  }
  auto project = [&](const float *__restrict__ x) -> float {
    if constexpr (FT > 0) return project_reg<FT>(wreg, x);
    else return project_lds(wt_lds, D, F, d, x);
  };
  auto accumulate = [&](float coef, const float *__restrict__ x) {
    if constexpr (FT > 0) {
#pragma unroll
      for (int f = 0; f < FT; ++f) acc[f] = fmaf(coef, x[f], acc[f]);
    } else if (on) {   // lanes >= D alias channel D-1: they must not touch its accumulator
      for (int f = 0; f < F; ++f) acc_lds[f * D + d] = fmaf(coef, x[f], acc_lds[f * D + d]);
    }
  };
  const int y = labels[row];
  const float H1 = on ? h1[(int64_t)row * D + lane] : 0.0f;
  const float dH1 = on ? d_h1[(int64_t)row * D + lane] : 0.0f;
  float dNb = on ? d_nbar[(int64_t)row * D + lane] : 0.0f;
  if (y == 1) {
    const float G = on ? gen[(int64_t)row * D + lane] : 0.0f;
    const float dG = on ? d_gen[(int64_t)row * D + lane] : 0.0f;
    const float dZ = (G > 0.0f) ? dG : 0.0f;                          // relu(fc(.))
    if (on) { dz[(int64_t)row * D + lane] = dZ; zs[lane] = dZ; }
    __syncthreads();
    const float *fc = params + L.o_fc();
    float a = 0.0f;
    for (int dd = 0; dd < D; ++dd) a = fmaf(fc[dd * D + d], zs[dd], a);   // fc^T dZ
    dNb += a;
  }
  const float dA = (H1 > 0.0f) ? dH1 : 0.0f;
  accumulate(on ? dA : 0.0f, x1 + (int64_t)row * F);
  const int e0 = ent_ptr[row], e1 = ent_ptr[row + 1];
  const int r = e1 - e0;
  const float g = on ? dNb * (1.0f / (float)r) : 0.0f;
  for (int blk = 0; blk < r; blk += 64) {
    const int ov = (blk + lane < r) ? ent_own[e0 + blk + lane] : 0;
    const int cnt = min(64, r - blk);
    for (int i = 0; i < cnt; ++i) {
      const int o = __builtin_amdgcn_readlane(ov, i);
      const float *xr = x2 + (int64_t)o * F;
      const float he = project(xr);
      accumulate(he > 0.0f ? g : 0.0f, xr);
    }
  }
  float *out = dw_part + (int64_t)blockIdx.x * F * D;
  if constexpr (FT > 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f)
      if (on) out[f * D + lane] = acc[f];
  } else {
    __syncthreads();
    for (int i = lane; i < F * D; i += 64) out[i] = acc_lds[i];
  }
}

// ------------------------------------------------------------------ gradient reduce, Adam, sync, score
__global__ void __launch_bounds__(256) k_grad_reduce(ParamLayout L, const int32_t *__restrict__ labels, int row0, int B,
                                                     const float *__restrict__ nbar, const float *__restrict__ dw_part,
                                                     const float *__restrict__ dz, const float *__restrict__ grad_w,
                                                     float *__restrict__ grads) {
  const int D = L.D, F = L.F;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < D) { grads[t] = grad_w[t]; return; }
  int u = t - D;
  if (u < D * F) {
    const int f = u / D, d = u - f * D;           // consecutive threads -> consecutive d (coalesced reads)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 4 <= B; b += 4) {
      s0 += dw_part[((int64_t)(b + 0) * F + f) * D + d];
      s1 += dw_part[((int64_t)(b + 1) * F + f) * D + d];
      s2 += dw_part[((int64_t)(b + 2) * F + f) * D + d];
      s3 += dw_part[((int64_t)(b + 3) * F + f) * D + d];
    }
    for (; b < B; ++b) s0 += dw_part[((int64_t)b * F + f) * D + d];
    grads[L.o_W() + d * F + f] = (s0 + s1) + (s2 + s3);
    return;
  }
  u -= D * F;
  if (u < D * D) {
    const int dd = u / D, d2 = u - dd * D;        // d fc[dd][d2] = sum_i dZ_i[dd] * nbar_i[d2]
    float s = 0.0f;
    for (int b = 0; b < B; ++b) {
      const int row = row0 + b;
      if (labels[row] == 1) s = fmaf(dz[(int64_t)row * D + dd], nbar[(int64_t)row * D + d2], s);
    }
    grads[L.o_fc() + u] = s;
  }
}

__global__ void __launch_bounds__(256) k_adam(float *__restrict__ params, float *__restrict__ m, float *__restrict__ v,
                                              const float *__restrict__ grads, ParamLayout L, float lr, float wd,
                                              float grad_scale, const int32_t *__restrict__ step_counter) {
  __shared__ float sc[2];
  if (threadIdx.x == 0) {
    const double t = (double)(*step_counter);
    const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
    sc[0] = (float)((double)lr / bc1);      // step_size
    sc[1] = (float)sqrt(bc2);               // bias_correction2_sqrt
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L.n_train()) return;
  const float step_size = sc[0], bc2s = sc[1];
  float p = params[i];
  float g = grads[i] * grad_scale;
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
    const float h = fmaxf(project_lds(lds, D, F, d, x1 + (int64_t)row * F), 0.0f);
    const float s = wave_sum(wd * h);
    if (lane == 0) prob[row] = 1.0f / (1.0f + expf(-s));     // torch.sigmoid            graphsage.py:180
  }
}

template <typename KernelFn, typename... Args>
inline void launch_rows(KernelFn k, int n_rows, size_t lds_bytes, hipStream_t st, Args... args) {
  k<<<dim3(n_rows), dim3(64), lds_bytes, st>>>(args...);
}

}  // namespace

extern "C" {

int ggad_max_embed_dim(void) { return GGAD_MAX_D; }
int ggad_max_feat_dim(void) { return GGAD_MAX_F; }

int64_t ggad_mb_param_count(int32_t D, int32_t F) { return (int64_t)D + (int64_t)D * F + (int64_t)D * D; }
int64_t ggad_mb_param_block_elems(int32_t D, int32_t F) { return ggad_mb_param_count(D, F) + (int64_t)F * D + (int64_t)D * D; }

static bool dims_ok(int D, int F) { return D >= 1 && D <= GGAD_MAX_D && F >= 1 && (size_t)(2 * F * D + D) * 4 <= 60 * 1024; }

int ggad_mb_params_sync(float *params, int32_t D, int32_t F, ggad_stream_t stream) {
  GGAD_REQUIRE(params && dims_ok(D, F));
  ParamLayout L{D, F};
  k_params_sync<<<dim3((L.n_train() + 255) / 256), dim3(256), 0, as_stream(stream)>>>(params, L);
  GGAD_CHECK_LAUNCH("mb_params_sync");
  return GGAD_OK;
}

int ggad_mb_fwd_rows(const float *params, int32_t D, int32_t F, const float *x1, const float *x2,
                     const int32_t *ent_ptr, const int32_t *ent_own, const int32_t *labels, int32_t row0,
                     int32_t n_rows, int32_t train, float *h1, float *nbar, float *gen, ggad_stream_t stream) {
  GGAD_REQUIRE(params && x1 && h1 && dims_ok(D, F) && n_rows >= 0 && row0 >= 0);
  GGAD_REQUIRE(!train || (x2 && ent_ptr && ent_own && labels && nbar && gen));
  if (n_rows == 0) return GGAD_OK;
  ParamLayout L{D, F};
  hipStream_t st = as_stream(stream);
  if (F == 17)
    k_fwd_rows<17><<<dim3(n_rows), dim3(64), (size_t)(17 * D + D) * 4, st>>>(params, L, x1, x2, ent_ptr, ent_own, labels, row0,
                                                                             train, h1, nbar, gen);
  else
    k_fwd_rows<0><<<dim3(n_rows), dim3(64), (size_t)(F * D + D) * 4, st>>>(params, L, x1, x2, ent_ptr, ent_own, labels, row0,
                                                                           train, h1, nbar, gen);
  GGAD_CHECK_LAUNCH("mb_fwd_rows");
  return GGAD_OK;
}

int ggad_mb_loss(const float *params, int32_t D, const float *h1, const float *nbar, const float *gen,
                 const int32_t *labels, const int32_t *src_of_pos, int32_t row0, int32_t n_rows, float *losses8,
                 float *d_h1, float *d_gen, float *d_nbar, float *grad_w, int32_t *step_counter, ggad_stream_t stream) {
  GGAD_REQUIRE(params && h1 && nbar && gen && labels && src_of_pos && losses8 && d_h1 && d_gen && d_nbar && grad_w);
  GGAD_REQUIRE(D >= 1 && D <= GGAD_MAX_D && n_rows >= 1 && row0 >= 0);
  k_loss<<<dim3(1), dim3(LOSS_T), 0, as_stream(stream)>>>(params, D, h1, nbar, gen, labels, src_of_pos, row0, n_rows, losses8,
                                                         d_h1, d_gen, d_nbar, grad_w, step_counter);
  GGAD_CHECK_LAUNCH("mb_loss");
  return GGAD_OK;
}

int ggad_mb_bwd_rows(const float *params, int32_t D, int32_t F, const float *x1, const float *x2,
                     const int32_t *ent_ptr, const int32_t *ent_own, const int32_t *labels, int32_t row0,
                     int32_t n_rows, const float *h1, const float *nbar, const float *gen, const float *d_h1,
                     const float *d_gen, const float *d_nbar, float *dw_part, float *dz, ggad_stream_t stream) {
  GGAD_REQUIRE(params && x1 && x2 && ent_ptr && ent_own && labels && h1 && nbar && gen && d_h1 && d_gen && d_nbar &&
               dw_part && dz);
  GGAD_REQUIRE(dims_ok(D, F) && n_rows >= 1 && row0 >= 0);
  ParamLayout L{D, F};
  hipStream_t st = as_stream(stream);
  if (F == 17)
    k_bwd_rows<17><<<dim3(n_rows), dim3(64), (size_t)(2 * 17 * D + D) * 4, st>>>(
        params, L, x1, x2, ent_ptr, ent_own, labels, row0, h1, nbar, gen, d_h1, d_gen, d_nbar, dw_part, dz);
  else
    k_bwd_rows<0><<<dim3(n_rows), dim3(64), (size_t)(2 * F * D + D) * 4, st>>>(
        params, L, x1, x2, ent_ptr, ent_own, labels, row0, h1, nbar, gen, d_h1, d_gen, d_nbar, dw_part, dz);
  GGAD_CHECK_LAUNCH("mb_bwd_rows");
  return GGAD_OK;
}

int ggad_mb_grad_reduce(int32_t D, int32_t F, const int32_t *labels, int32_t row0, int32_t n_rows,
                        const float *nbar, const float *dw_part, const float *dz, const float *grad_w,
                        float *grads, ggad_stream_t stream) {
  GGAD_REQUIRE(labels && nbar && dw_part && dz && grad_w && grads && dims_ok(D, F) && n_rows >= 1);
  ParamLayout L{D, F};
  k_grad_reduce<<<dim3((L.n_train() + 255) / 256), dim3(256), 0, as_stream(stream)>>>(L, labels, row0, n_rows, nbar, dw_part,
                                                                                     dz, grad_w, grads);
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
  const int blocks = (int)fminf((float)((n_rows + 3) / 4), 4096.0f);
  k_score<<<dim3(blocks), dim3(256), (size_t)F * D * 4, as_stream(stream)>>>(params, L, x1, n_rows, prob);
  GGAD_CHECK_LAUNCH("mb_score");
  return GGAD_OK;
}

}  // extern "C"
