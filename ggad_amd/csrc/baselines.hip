// Loss kernels of the comparison models that run on GGAD's aggregation kernels: the reconstruction terms of the mini-batch
// DOMINANT / AnomalyDAE variants (reference src/graphsage_dominant.py:154-171, src/graphsage_anomalydae.py:154-175, scored by
// src/utils.py:140-172; B x F is 150 x 17 on DGraph-Fin: one workgroup, fixed summation order) and the one-class loss of the
// full-graph OCGNN (ocgnn.py:83-118).
#include "common.h"

#define RC_WAVES 16

// loss = mean_c sqrt( sum_b w(a_bc) (a_bc - t_bc)^2 ),  w = w_pos where a > 0 else w_neg   -- the sum runs over the BATCH
// axis (torch.sum(diff, 0)), as the reference writes it.  da = d loss / d a (the weight does not depend on a for autograd).
__global__ __launch_bounds__(RC_WAVES * GGAD_WAVE) void k_recon_cols(const float *__restrict__ a, const float *__restrict__ t,
                                                                    int n_rows, int n_cols, float w_pos, float w_neg,
                                                                    float *__restrict__ loss, float *__restrict__ col_sum,
                                                                    float *__restrict__ da) {
  extern __shared__ float s_col[];                     // n_cols square-rooted column sums
  const int wave = threadIdx.x / GGAD_WAVE, lane = lane_id();
  for (int c = wave; c < n_cols; c += RC_WAVES) {
    float s = 0.f;
    for (int b = lane; b < n_rows; b += GGAD_WAVE) {
      float av = a[(size_t)b * n_cols + c], d = av - t[(size_t)b * n_cols + c];
      s += (d * d) * (av > 0.f ? w_pos : w_neg);
    }
    s = wave_sum(s);
    if (lane == 0) {
      s_col[c] = sqrtf(s);
      if (col_sum) col_sum[c] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int c = 0; c < n_cols; ++c) acc += s_col[c];
    loss[0] = acc / (float)n_cols;
  }
  if (!da) return;
  const float inv_cols = 1.f / (float)n_cols;
  for (int i = threadIdx.x; i < n_rows * n_cols; i += RC_WAVES * GGAD_WAVE) {
    int c = i % n_cols;
    float av = a[i], d = av - t[i];
    // d sqrt(s)/ds = 1/(2 sqrt(s)); ds/da = 2 w d.  A zero column gives 0/0 = NaN exactly like torch's sqrt backward.
    da[i] = ((av > 0.f ? w_pos : w_neg) * d / s_col[c]) * inv_cols;
  }
}

// out[b] = sqrt( sum_c (a_bc - t_bc)^2 ): the anomaly score of test_recon (src/utils.py:158-159).  One wave per row.
__global__ __launch_bounds__(256) void k_recon_rows(const float *__restrict__ a, const float *__restrict__ t, long n_rows,
                                                    int n_cols, float *__restrict__ out) {
  long row = (long)blockIdx.x * 4 + threadIdx.x / GGAD_WAVE;
  if (row >= n_rows) return;
  const int lane = lane_id();
  float s = 0.f;
  for (int c = lane; c < n_cols; c += GGAD_WAVE) {
    float d = a[row * n_cols + c] - t[row * n_cols + c];
    s += d * d;
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = sqrtf(s);
}

// One-class hypersphere loss of the full-graph OCGNN comparison model (reference ocgnn.py:83-118): for the rows idx[],
//   score_i = ||emb_i - c||^2 - r^2,   loss = r^2 + (1/beta) mean_i relu(score_i)
// (the reference re-creates c = 0, r = 0 on every call, so its warm-up update never takes effect; c and r stay arguments
// here).  Pass 1: one wave per listed row writes score_i and, when asked, d loss / d emb_i = (2 / (beta n)) [score_i > 0]
// (emb_i - c) into the row's slot of a zero-initialised gradient.  Pass 2: one workgroup sums relu(score) in index order.
__global__ __launch_bounds__(256) void k_ocgnn_rows(const float *__restrict__ emb, const int64_t *__restrict__ idx, long n_idx,
                                                    int h, const float *__restrict__ center, float r, float beta,
                                                    float *__restrict__ score, float *__restrict__ demb) {
  long i = (long)blockIdx.x * 4 + threadIdx.x / GGAD_WAVE;
  if (i >= n_idx) return;
  const int lane = lane_id();
  const long row = idx ? idx[i] : i;
  const float *e = emb + row * h;
  float s = 0.f;
  for (int c = lane; c < h; c += GGAD_WAVE) {
    float d = e[c] - (center ? center[c] : 0.f);
    s += d * d;
  }
  s = wave_sum(s) - r * r;
  if (lane == 0) score[i] = s;
  if (demb) {
    const float g = s > 0.f ? 2.f / (beta * (float)n_idx) : 0.f;
    for (int c = lane; c < h; c += GGAD_WAVE) demb[row * h + c] = g * (e[c] - (center ? center[c] : 0.f));
  }
}

__global__ __launch_bounds__(1024) void k_ocgnn_reduce(const float *__restrict__ score, long n, float r, float beta,
                                                       float *__restrict__ loss) {
  __shared__ float part[16];
  // fixed assignment of contiguous index ranges to threads, then a fixed-order combine: deterministic for given n
  const long per = (n + 1023) / 1024;
  const long lo = (long)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
  float s = 0.f;
  for (long i = lo; i < hi; ++i) s += fmaxf(score[i], 0.f);
  s = wave_sum(s);
  if (lane_id() == 0) part[threadIdx.x / GGAD_WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int w = 0; w < 16; ++w) acc += part[w];
    loss[0] = r * r + (1.f / beta) * (acc / (float)n);
  }
}

extern "C" {

int ggad_ocgnn_loss_f32(const float *emb, const int64_t *idx, int64_t n_idx, int32_t h, const float *center, float r, float beta,
                        float *loss, float *score, float *demb, ggad_stream_t stream) {
  GGAD_REQUIRE(emb && loss && score && n_idx >= 1 && h >= 1 && beta > 0.f);
  k_ocgnn_rows<<<dim3((unsigned)((n_idx + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(emb, idx, (long)n_idx, h, center, r, beta,
                                                                                      score, demb);
  GGAD_CHECK_LAUNCH("ocgnn_rows");
  k_ocgnn_reduce<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(score, (long)n_idx, r, beta, loss);
  GGAD_CHECK_LAUNCH("ocgnn_reduce");
  return GGAD_OK;
}

int ggad_recon_cols_f32(const float *a, const float *t, int32_t n_rows, int32_t n_cols, float w_pos, float w_neg, float *loss,
                        float *col_sum, float *da, ggad_stream_t stream) {
  GGAD_REQUIRE(a && t && loss && n_rows >= 1 && n_cols >= 1 && n_cols <= 8192);
  GGAD_REQUIRE((int64_t)n_rows * n_cols < (1ll << 31));
  k_recon_cols<<<dim3(1), dim3(RC_WAVES * GGAD_WAVE), n_cols * sizeof(float), as_stream(stream)>>>(a, t, n_rows, n_cols, w_pos,
                                                                                                  w_neg, loss, col_sum, da);
  GGAD_CHECK_LAUNCH("recon_cols");
  return GGAD_OK;
}

int ggad_recon_rows_f32(const float *a, const float *t, int64_t n_rows, int32_t n_cols, float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(a && t && out && n_rows >= 0 && n_cols >= 1);
  if (n_rows == 0) return GGAD_OK;
  k_recon_rows<<<dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(a, t, (long)n_rows, n_cols, out);
  GGAD_CHECK_LAUNCH("recon_rows");
  return GGAD_OK;
}

}  // extern "C"
