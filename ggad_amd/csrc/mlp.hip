// Fused scorer MLP of the full-graph GGAD path (gfx950, exact-f32 matrix cores):
//
//   f_1 = relu(fc1(emb_combine)); f_2 = relu(fc2(f_1)); f_3 = fc3(f_2)            model.py:176-180   (H -> H/2 -> H/4 -> 1, no bias)
//
// The reference runs three nn.Linear (+ two ReLU) forward and, through autograd, three data-gradient products and two ReLU
// backward passes; round 3 ran them as three GEMM launches forward and a dozen launches of 5-16 us backward, on 1,200-6,500 rows
// (the normal + outlier rows the loss reads): launch-bound.  Here
//   k_mlp_fwd     one launch: a workgroup takes 16 rows, keeps them (and f_1, f_2) in LDS and streams W1 / W2 from L2 in the
//                 fragment layout of v_mfma_f32_16x16x4_f32; writes f_1, f_2 (the backward needs them) and f_3
//   k_mlp_dgrad   one launch: dz_2 = (g_3 w3) [f_2 > 0], dz_1 = (dz_2 W2) [f_1 > 0], d_x = dz_1 W1 (+ an incoming gradient of x);
//                 writes dz_2, dz_1 (operands of the two weight gradients, which stay split-K GEMMs over all rows) and d_x
// A 16 x 16 x 4 MFMA sums over the 4 k its lane groups hold; WHICH k that is only has to agree between the A and the B fragment,
// so lane group q of step s of the j-th 16-k block takes k = 16 j + 4 q + s: every lane fetches its 4 values with ONE 16-byte
// load (A from LDS, B from the weight row in L2).  Results differ from a k-ordered chain by fp32 round-off only.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

typedef float mlp_f4 __attribute__((ext_vector_type(4)));
constexpr int MLP_R = 16;                 // rows per workgroup (one MFMA row tile)
constexpr int MLP_HMAX = 512, MLP_H1MAX = 256, MLP_H2MAX = 128;
constexpr int MLP_T1 = MLP_H1MAX / 16 / 4, MLP_T2 = MLP_H2MAX / 16 / 4;     // column tiles per wave at most (stage 1: 4, stage 2: 2)

__host__ __device__ inline int mlp_pad(int n) { return ((n + 15) / 16) * 16 + 4; }   // LDS row stride in floats: 16-byte aligned rows, banks staggered

// B fragment values of one 16-k block for column n of a weight stored [n][k] (k fastest): k = kb .. kb + 3.  Every load is
// UNCONDITIONAL (row and k clamped into the matrix, the value discarded by a select afterwards): a load inside a divergent branch
// costs hipcc a branch and an `s_waitcnt vmcnt(0)` each, which serialised the whole k loop (44 us for 19 blocks).  VEC: floats per
// load the row alignment allows (4: K % 4 == 0; 2: K even).  TAIL: the last, partial 16-k block (per-float loads).
template <int VEC, bool TAIL>
__device__ __forceinline__ mlp_f4 mlp_ldw_kfast(const float *__restrict__ W, int n_c, bool n_ok, int K, int kb) {
  const float *row = W + (int64_t)n_c * K;
  mlp_f4 v;
  if constexpr (!TAIL) {
    if constexpr (VEC == 4) v = *reinterpret_cast<const mlp_f4 *>(row + kb);
    else { const float2 a = *reinterpret_cast<const float2 *>(row + kb), b = *reinterpret_cast<const float2 *>(row + kb + 2); v = mlp_f4{a.x, a.y, b.x, b.y}; }
    return n_ok ? v : mlp_f4{0.f, 0.f, 0.f, 0.f};
  } else {
    const int k1 = K - 1;
    v.x = row[min(kb, k1)]; v.y = row[min(kb + 1, k1)]; v.z = row[min(kb + 2, k1)]; v.w = row[min(kb + 3, k1)];
    v.x = (n_ok && kb < K) ? v.x : 0.f; v.y = (n_ok && kb + 1 < K) ? v.y : 0.f;
    v.z = (n_ok && kb + 2 < K) ? v.z : 0.f; v.w = (n_ok && kb + 3 < K) ? v.w : 0.f;
    return v;
  }
}

// round 6: inside a training epoch the weights were rewritten by Adam a moment ago -- no XCD's L2 holds them -- and a layer's k loop
// requests them three 16-k blocks at a time: six to seven DEPENDENT first-touch round trips per layer (the stand-alone timings of
// round 4, back to back on warm L2s, never saw them: 17 us there, 22.9 us inside the Reddit-size epoch).  Every thread therefore
// touches the 128-byte lines of both weight matrices once at the top of the kernel: one round trip, behind which the k loops hit.
// GGAD_MLP_TOUCH=0 at build time (-DGGAD_MLP_TOUCH=0) takes it out (A/B).
#ifndef GGAD_MLP_TOUCH
#define GGAD_MLP_TOUCH 1
#endif
// The touches are LDS-DMA loads (global_load_lds_dword: no destination register to clobber, counted by vmcnt like any load) into a
// 1-KB scratch piece of LDS nobody reads; the first workgroup barrier of the kernel waits for them together with the row loads.
__device__ __forceinline__ void mlp_touch(const float *__restrict__ W, int n_floats, int tid, float *junk) {
#if GGAD_MLP_TOUCH
  const int wave = tid >> 6;
  for (int i = tid * 32; i < n_floats; i += 256 * 32)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(W + i),
                                     (__attribute__((address_space(3))) void *)(junk + wave * 64), 4, 0, 0);
#endif
}

// out (16 x n_out, LDS tile `dst` with stride ld_dst, and global `gdst` rows row0.., ld = n_out) = act(src (16 x K in LDS) x W^T),
// W = [n_out][K].  Every wave takes the column tiles t = wave, wave + 4, ...
template <int MAXT, bool RELU, int VEC>
__device__ __forceinline__ void mlp_layer_kfast_nt(const float *__restrict__ src, int ld_src, int K, const float *__restrict__ W, int n_out,
                                                  float *__restrict__ dst, int ld_dst, float *__restrict__ gdst, int row0, int n_rows,
                                                  int wave, int lane) {
  // MAXT here = the EXACT number of column tiles of this wave (dispatched below): no guard inside the k loop
  const int m = lane & 15, q = lane >> 4;
  mlp_f4 acc[MAXT];
  int n_c[MAXT];
  bool n_ok[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    acc[t] = mlp_f4{0.f, 0.f, 0.f, 0.f};
    const int n = (wave + 4 * t) * 16 + m;
    n_ok[t] = n < n_out;
    n_c[t] = n_ok[t] ? n : 0;
  }
  const int nfull = K / 16, nblk = (K + 15) / 16;            // full 16-k blocks; one partial block behind them when K % 16
  auto mma = [&](const mlp_f4 &a, const mlp_f4 (&b)[MAXT]) {
    // (tiles interleaved inside a k step: a 16x16x4 MFMA issues every 32 cycles but its accumulator is ready after 40)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[t][c], acc[t], 0, 0, 0);
    }
  };
  // three 16-k blocks of W in flight (an L2 round trip is ~3 blocks of MFMAs long); blocks past the last full one are clamped
  // to it (a redundant, unconditional load)
  auto ldb = [&](mlp_f4 (&b)[MAXT], int j) {
    const int kb = 16 * min(j, nfull - 1) + 4 * q;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) b[t] = mlp_ldw_kfast<VEC, false>(W, n_c[t], n_ok[t], K, kb);
  };
  auto lda = [&](int j) { return *reinterpret_cast<const mlp_f4 *>(src + m * ld_src + 16 * j + 4 * q); };      // (LDS tile zero-padded to 16 k)
  mlp_f4 b0[MAXT], b1[MAXT], b2[MAXT], b_cur[MAXT];
  if (nfull > 0) { ldb(b0, 0); ldb(b1, 1); ldb(b2, 2); }
  for (int j = 0; j < nfull; j += 3) {
    mma(lda(j), b0);
    ldb(b0, j + 3);
    if (j + 1 < nfull) { mma(lda(j + 1), b1); ldb(b1, j + 4); }
    if (j + 2 < nfull) { mma(lda(j + 2), b2); ldb(b2, j + 5); }
  }
  if (nblk > nfull) {
    const int kb = 16 * nfull + 4 * q;
    const mlp_f4 a = *reinterpret_cast<const mlp_f4 *>(src + m * ld_src + kb);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) b_cur[t] = mlp_ldw_kfast<VEC, true>(W, n_c[t], n_ok[t], K, kb);
    mma(a, b_cur);
  }
  // C / D layout: column = lane & 15, rows 4 (lane >> 4) + i
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int n = (wave + 4 * t) * 16 + m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      float v = acc[t][i];
      if (RELU) v = fmaxf(v, 0.f);
      if (n < n_out) {
        dst[r * ld_dst + n] = v;
        if (gdst && r < n_rows) gdst[(int64_t)(row0 + r) * n_out + n] = v;
      } else if (n < ld_dst) {
        dst[r * ld_dst + n] = 0.f;               // zero padding up to the next multiple of 16: the next layer's k blocks read it
      }
    }
  }
}

// wave w takes the column tiles w, w + 4, ...: dispatch on how many that is (<= MAXT)
template <int MAXT, bool RELU, int VEC>
__device__ __forceinline__ void mlp_layer_kfast(const float *__restrict__ src, int ld_src, int K, const float *__restrict__ W, int n_out,
                                               float *__restrict__ dst, int ld_dst, float *__restrict__ gdst, int row0, int n_rows,
                                               int wave, int lane) {
  const int n_tiles = (n_out + 15) / 16;
  const int nt = n_tiles > wave ? (n_tiles - wave + 3) / 4 : 0;
  if (nt == 1) mlp_layer_kfast_nt<1, RELU, VEC>(src, ld_src, K, W, n_out, dst, ld_dst, gdst, row0, n_rows, wave, lane);
  else if (nt == 2) mlp_layer_kfast_nt<2, RELU, VEC>(src, ld_src, K, W, n_out, dst, ld_dst, gdst, row0, n_rows, wave, lane);
  else if (MAXT >= 3 && nt == 3) mlp_layer_kfast_nt<(MAXT >= 3 ? 3 : 1), RELU, VEC>(src, ld_src, K, W, n_out, dst, ld_dst, gdst, row0, n_rows, wave, lane);
  else if (MAXT >= 4 && nt == 4) mlp_layer_kfast_nt<(MAXT >= 4 ? 4 : 1), RELU, VEC>(src, ld_src, K, W, n_out, dst, ld_dst, gdst, row0, n_rows, wave, lane);
}

__global__ void __launch_bounds__(256) k_mlp_fwd(const float *__restrict__ X, int64_t ldx, int R, int H, int H1, int H2,
                                                 const float *__restrict__ W1, const float *__restrict__ W2, const float *__restrict__ w3,
                                                 float *__restrict__ f1, float *__restrict__ f2, float *__restrict__ f3) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ldx_s = mlp_pad(H), ld1 = mlp_pad(H1), ld2 = mlp_pad(H2);
  float *xs = smem, *h1 = xs + MLP_R * ldx_s, *h2 = h1 + MLP_R * ld1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * MLP_R;
  const int n_rows = min(MLP_R, R - row0);
  __shared__ float touch_junk[256];
  mlp_touch(W1, H1 * H, tid, touch_junk);
  mlp_touch(W2, H2 * H1, tid, touch_junk);
  // rows of x -> LDS (float4 along the row; zero behind the row end and for rows past R)
  for (int i = tid; i < MLP_R * (ldx_s / 4); i += 256) {
    const int r = i / (ldx_s / 4), c4 = i - r * (ldx_s / 4);
    const bool ok = r < n_rows && 4 * c4 + 4 <= H;               // (clamped address + select: no load inside a divergent branch)
    const mlp_f4 ld = *reinterpret_cast<const mlp_f4 *>(X + (int64_t)(row0 + (ok ? r : 0)) * ldx + (ok ? 4 * c4 : 0));
    *reinterpret_cast<mlp_f4 *>(xs + r * ldx_s + 4 * c4) = ok ? ld : mlp_f4{0.f, 0.f, 0.f, 0.f};
  }
  // (columns [H1, ld1) / [H2, ld2) of the hidden tiles beyond the last written column tile: zero -- the layer epilogue zeroes the
  //  rest of a partly used tile, whole unused 4-float tails are cleared here)
  for (int i = tid; i < MLP_R * 4; i += 256) {
    h1[(i >> 2) * ld1 + ld1 - 4 + (i & 3)] = 0.f;
    h2[(i >> 2) * ld2 + ld2 - 4 + (i & 3)] = 0.f;
  }
  __syncthreads();
  mlp_layer_kfast<MLP_T1, true, 4>(xs, ldx_s, H, W1, H1, h1, ld1, f1, row0, n_rows, wave, lane);
  __syncthreads();
  if ((H1 & 3) == 0) mlp_layer_kfast<MLP_T2, true, 4>(h1, ld1, H1, W2, H2, h2, ld2, f2, row0, n_rows, wave, lane);
  else mlp_layer_kfast<MLP_T2, true, 2>(h1, ld1, H1, W2, H2, h2, ld2, f2, row0, n_rows, wave, lane);
  __syncthreads();
  if (wave == 0) {                                   // f_3[r] = <f_2[r], w3>: 4 lanes per row, fixed order
    const int r = lane >> 2, part = lane & 3;
    float s = 0.f;
    for (int n = part; n < H2; n += 4) s = fmaf(h2[r * ld2 + n], w3[n], s);
    s += __shfl_xor(s, 1, GGAD_WAVE);
    s += __shfl_xor(s, 2, GGAD_WAVE);
    if (part == 0 && r < n_rows) f3[row0 + r] = s;
  }
}

// B fragment values of one 16-k block for column n of a weight stored [k][n] (n fastest): rows kb .. kb + 3 (unconditional
// loads from clamped rows, selects afterwards)
__device__ __forceinline__ mlp_f4 mlp_ldw_nfast(const float *__restrict__ W, int n_c, bool n_ok, int K, int N, int kb) {
  const int k1 = K - 1;
  const float *p = W + n_c;
  mlp_f4 v;
  v.x = p[(int64_t)min(kb, k1) * N]; v.y = p[(int64_t)min(kb + 1, k1) * N]; v.z = p[(int64_t)min(kb + 2, k1) * N]; v.w = p[(int64_t)min(kb + 3, k1) * N];
  v.x = (n_ok && kb < K) ? v.x : 0.f; v.y = (n_ok && kb + 1 < K) ? v.y : 0.f;
  v.z = (n_ok && kb + 2 < K) ? v.z : 0.f; v.w = (n_ok && kb + 3 < K) ? v.w : 0.f;
  return v;
}

// d (16 x n_out) = src (16 x K in LDS) x W, W = [K][n_out]; then d *= [gate > 0] (gate: 16 x n_out rows of a forward activation,
// global, or null); result to the LDS tile `dst` (or null) and to global `gdst` (+ `gadd` if given).  NT = the exact number of column
// tiles (t0, t0 + 4, ...) this call walks: no guard inside the k loop.
template <int NT>
__device__ __forceinline__ void mlp_layer_nfast_nt(const float *__restrict__ src, int ld_src, int K, const float *__restrict__ W, int n_out,
                                                  const float *__restrict__ gate, float *__restrict__ dst, int ld_dst,
                                                  float *__restrict__ gdst, int64_t ld_g, const float *__restrict__ gadd, int64_t ld_add,
                                                  int row0, int n_rows, int t0, int lane) {
  const int m = lane & 15, q = lane >> 4;
  const int nblk = (K + 15) / 16;
  mlp_f4 acc[NT];
  int n_c[NT];
  bool n_ok[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = mlp_f4{0.f, 0.f, 0.f, 0.f};
    const int n = (t0 + 4 * t) * 16 + m;
    n_ok[t] = n < n_out;
    n_c[t] = n_ok[t] ? n : 0;
  }
  auto ldb = [&](mlp_f4 (&b)[NT], int j) {                        // (blocks past the end: every k >= K, i.e. zeros)
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = mlp_ldw_nfast(W, n_c[t], n_ok[t], K, n_out, 16 * j + 4 * q);
  };
  auto mma = [&](int j, const mlp_f4 (&b)[NT]) {
    const mlp_f4 a = *reinterpret_cast<const mlp_f4 *>(src + m * ld_src + 16 * j + 4 * q);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], b[t][c], acc[t], 0, 0, 0);
    }
  };
  mlp_f4 b0[NT], b1[NT];
  ldb(b0, 0);
  ldb(b1, 1);
  for (int j = 0; j < nblk; j += 2) {                             // two blocks of W in flight
    mma(j, b0);
    ldb(b0, j + 2);
    if (j + 1 < nblk) { mma(j + 1, b1); ldb(b1, j + 3); }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = (t0 + 4 * t) * 16 + m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;
      float v = acc[t][i];
      const bool live = r < n_rows && n < n_out;
      const int64_t rr = row0 + (live ? r : 0), nn = live ? n : 0;      // (clamped: the loads below are unconditional)
      if (gate) v = (live && gate[rr * n_out + nn] > 0.f) ? v : 0.f;
      const float add = gadd ? gadd[rr * ld_add + nn] : 0.f;
      if (n < n_out) {
        if (dst) dst[r * ld_dst + n] = v;
        if (live) gdst[rr * ld_g + nn] = v + add;
      } else if (dst && n < ld_dst) {
        dst[r * ld_dst + n] = 0.f;
      }
    }
  }
}

// wave w takes the column tiles w, w + 4, ...: in sweeps of at most MAXT tiles, each with its exact count
template <int MAXT>
__device__ __forceinline__ void mlp_layer_nfast(const float *__restrict__ src, int ld_src, int K, const float *__restrict__ W, int n_out,
                                               const float *__restrict__ gate, float *__restrict__ dst, int ld_dst,
                                               float *__restrict__ gdst, int64_t ld_g, const float *__restrict__ gadd, int64_t ld_add,
                                               int row0, int n_rows, int wave, int lane) {
  static_assert(MAXT >= 1 && MAXT <= 5, "sweeps of 1 .. 5 column tiles");
  const int n_tiles = (n_out + 15) / 16;
  const int total = n_tiles > wave ? (n_tiles - wave + 3) / 4 : 0;
  for (int done = 0; done < total;) {
    const int take = min(MAXT, total - done), t0 = wave + 4 * done;
#define MLP_NF(N_) mlp_layer_nfast_nt<((N_) <= MAXT ? (N_) : 1)>(src, ld_src, K, W, n_out, gate, dst, ld_dst, gdst, ld_g, gadd, ld_add, row0, n_rows, t0, lane)
    if (take == 1) MLP_NF(1);
    else if (take == 2) MLP_NF(2);
    else if (take == 3) MLP_NF(3);
    else if (take == 4) MLP_NF(4);
    else MLP_NF(5);
#undef MLP_NF
    done += take;
  }
}

__global__ void __launch_bounds__(256) k_mlp_dgrad(const float *__restrict__ g3, int R, int H, int H1, int H2,
                                                   const float *__restrict__ f1, const float *__restrict__ f2,
                                                   const float *__restrict__ W1, const float *__restrict__ W2, const float *__restrict__ w3,
                                                   float *__restrict__ dz2, float *__restrict__ dz1, float *__restrict__ dx, int64_t ld_dx,
                                                   const float *__restrict__ dx_add, int64_t ld_add) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int ld1 = mlp_pad(H1), ld2 = mlp_pad(H2);
  float *z2 = smem, *z1 = z2 + MLP_R * ld2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * MLP_R;
  const int n_rows = min(MLP_R, R - row0);
  __shared__ float touch_junk[256];
  mlp_touch(W2, H2 * H1, tid, touch_junk);
  mlp_touch(W1, H1 * H, tid, touch_junk);
  // dz_2 = g_3 w3^T masked by f_2 > 0 (model.py:178-180 backwards); zero-padded to the next 16 k
  for (int i = tid; i < MLP_R * ld2; i += 256) {
    const int r = i / ld2, n = i - r * ld2;
    const bool ok = r < n_rows && n < H2;
    const int64_t rr = row0 + (ok ? r : 0);
    const int nn = ok ? n : 0;
    const float v = (ok && f2[rr * H2 + nn] > 0.f) ? g3[rr] * w3[nn] : 0.f;
    if (ok) dz2[rr * H2 + nn] = v;
    z2[i] = v;
  }
  for (int i = tid; i < MLP_R * 4; i += 256) z1[(i >> 2) * ld1 + ld1 - 4 + (i & 3)] = 0.f;
  __syncthreads();
  mlp_layer_nfast<MLP_T1>(z2, ld2, H2, W2, H1, f1, z1, ld1, dz1, H1, nullptr, 0, row0, n_rows, wave, lane);     // dz_1 = (dz_2 W2) [f_1 > 0]
  __syncthreads();
  mlp_layer_nfast<5>(z1, ld1, H1, W1, H, nullptr, nullptr, 0, dx, ld_dx, dx_add, ld_add, row0, n_rows, wave, lane);   // d_x = dz_1 W1
}


// ---- the three weight gradients of the scorer in ONE launch (+ one reduction): dW1 = dz_1^T x, dW2 = dz_2^T f_1, dW3 = g_3^T f_2
// (model.py:176-180 backwards).  They were three split-K GEMM launches + three reductions of 9-13 + 4 us each for 0.2 GFLOP -- 47 us
// of a 0.54-ms Reddit epoch, launch- and latency-bound (profiles/r04_reddit_last_epoch.txt).  All three are P^T Q over the SAME R rows:
// a wave takes a tile of the output and a range of rows, and per step of 4 rows loads ONE vector of P's columns and ONE of Q's per
// lane -- lane (a, g) of v_mfma_f32_16x16x4_f32 holds A[a][g] = P[row g][.] and B[g][a] = Q[row g][.]; with VP / VQ consecutive
// columns per lane, component t of the vector belongs to the INTERLEAVED tile t (columns m0 + VP a + t), so one 16-byte load of x
// feeds four column tiles and a 16 VP x 16 VQ wave tile costs two loads per VP VQ MFMAs.  Partials per row range go to the workspace,
// k_mlp_wgrad_reduce adds them in range order (fixed: bit-reproducible).
struct WgProd {                       // one product D (m x n) = P^T (m x R) Q (R x n)
  const float *P, *Q;
  int64_t ldp, ldq;
  int m, n, vp, vq;                   // columns per lane of P / Q (4, 2 or 1: what their alignment allows)
  int splits, tiles_m, tiles_n;       // row ranges; wave tiles of 16 vp x 16 vq
  int task0;                          // first wave task of this product
  int64_t part0;                      // its partials in the workspace: [splits][m][n]
};
struct WgArgs {
  WgProd p[3];
  int R, n_tasks;
  float *ws;
};

template <int V> struct wg_vec;
template <> struct wg_vec<4> { typedef mlp_f4 T; };
template <> struct wg_vec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct wg_vec<1> { typedef float T; };
template <int V> __device__ __forceinline__ float wg_get(const typename wg_vec<V>::T &v, int t) { if constexpr (V == 1) return v; else return v[t]; }

template <int VP, int VQ>
__device__ __forceinline__ void wg_task(const WgProd &W, int R, float *__restrict__ ws, int tile, int split, int lane) {
  typedef typename wg_vec<VP>::T TP;
  typedef typename wg_vec<VQ>::T TQ;
  const int a = lane & 15, g = lane >> 4;
  const int tm = tile / W.tiles_n, tn = tile - tm * W.tiles_n;
  const int m0 = tm * 16 * VP, n0 = tn * 16 * VQ;
  // rows [r0, r1) of this range: multiples of 4 (the last range takes the tail; rows beyond R contribute zero through P)
  const int steps_all = (R + 3) >> 2;
  const int s0 = (int)((int64_t)steps_all * split / W.splits), s1 = (int)((int64_t)steps_all * (split + 1) / W.splits);
  const int pc = min(m0 + VP * a, W.m - VP), qc = min(n0 + VQ * a, W.n - VQ);      // clamped: a lane beyond the edge re-reads valid columns,
  const float *pp = W.P + pc, *qq = W.Q + qc;                                      // its outputs are not stored
  mlp_f4 acc[VP][VQ];
#pragma unroll
  for (int t = 0; t < VP; ++t)
#pragma unroll
    for (int u = 0; u < VQ; ++u) acc[t][u] = mlp_f4{0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 6;                                      // steps per batch: the loads of batch b + 1 are in flight under the MFMAs of batch b
  TP pv[2][UN];
  TQ qv[2][UN];
  auto load_batch = [&](int sb, TP (&pd)[UN], TQ (&qd)[UN]) {
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const int row = min(4 * min(sb + i, s1 - 1) + g, R - 1);
      pd[i] = *reinterpret_cast<const TP *>(pp + (int64_t)row * W.ldp);
      qd[i] = *reinterpret_cast<const TQ *>(qq + (int64_t)row * W.ldq);
    }
  };
  auto mul_batch = [&](int sb, const TP (&pd)[UN], const TQ (&qd)[UN]) {
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const bool on = sb + i < s1 && 4 * (sb + i) + g < R;
#pragma unroll
      for (int t = 0; t < VP; ++t) {
        const float av = on ? wg_get<VP>(pd[i], t) : 0.0f;
#pragma unroll
        for (int u = 0; u < VQ; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wg_get<VQ>(qd[i], u), acc[t][u], 0, 0, 0);
      }
    }
  };
  if (s0 < s1) load_batch(s0, pv[0], qv[0]);
  for (int sb = s0; sb < s1; sb += 2 * UN) {
    if (sb + UN < s1) load_batch(sb + UN, pv[1], qv[1]);
    mul_batch(sb, pv[0], qv[0]);
    if (sb + UN < s1) {
      if (sb + 2 * UN < s1) load_batch(sb + 2 * UN, pv[0], qv[0]);
      mul_batch(sb + UN, pv[1], qv[1]);
    }
  }
  // D[4 g + v][a] of tile (t, u) = output (m0 + VP (4 g + v) + t, n0 + VQ a + u): VQ consecutive columns per lane and row
  float *part = ws + W.part0 + (int64_t)split * W.m * W.n;
  if (n0 + VQ * a + VQ <= W.n) {
#pragma unroll
    for (int t = 0; t < VP; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int mi = m0 + VP * (4 * g + v) + t;
        if (mi < W.m) {
          TQ o;
          if constexpr (VQ == 1) o = acc[t][0][v];
          else {
#pragma unroll
            for (int u = 0; u < VQ; ++u) o[u] = acc[t][u][v];
          }
          *reinterpret_cast<TQ *>(part + (int64_t)mi * W.n + n0 + VQ * a) = o;
        }
      }
  }
}

__global__ void __launch_bounds__(256) k_mlp_wgrad(WgArgs A) {
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= A.n_tasks) return;
  const int q = task >= A.p[2].task0 ? 2 : (task >= A.p[1].task0 ? 1 : 0);        // wave-uniform
  const WgProd &W = A.p[q];
  const int local = task - W.task0;
  const int tiles = W.tiles_m * W.tiles_n;
  const int split = local / tiles, tile = local - split * tiles;
  if (W.vp == 2 && W.vq == 4) wg_task<2, 4>(W, A.R, A.ws, tile, split, lane);
  else if (W.vp == 1 && W.vq == 2) wg_task<1, 2>(W, A.R, A.ws, tile, split, lane);
  else if (W.vp == 1 && W.vq == 4) wg_task<1, 4>(W, A.R, A.ws, tile, split, lane);
  else if (W.vp == 2 && W.vq == 2) wg_task<2, 2>(W, A.R, A.ws, tile, split, lane);
  else wg_task<1, 1>(W, A.R, A.ws, tile, split, lane);
}

struct WgRed { float *out[3]; int64_t part0[3]; int len[3], splits[3]; };
__global__ void __launch_bounds__(256) k_mlp_wgrad_reduce(WgRed Q, const float *__restrict__ ws) {
  int i = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (i < Q.len[q]) {
      const float *p = ws + Q.part0[q] + i;
      float s = 0.0f;
      const int64_t st = Q.len[q];
      int k = 0;
      for (; k + 8 <= Q.splits[q]; k += 8) {                                   // range order: fixed; eight loads in flight
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(k + j) * st];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
      }
      for (; k < Q.splits[q]; ++k) s += p[k * st];
      Q.out[q][i] = s;
      return;
    }
    i -= Q.len[q];
  }
}

static int wg_pick_vec(const float *p, int64_t ld, int n, int want) {
  for (int v = want; v > 1; v >>= 1)
    if (ld % v == 0 && n % v == 0 && n >= v && ((uintptr_t)p & (size_t)(4 * v - 1)) == 0) return v;
  return 1;
}
// splits of a product: enough wave tasks to cover the chip a few times, >= 32 rows per range
static int wg_splits(int R, int tiles, int weight) {
  (void)tiles; (void)weight;
  return std::min(std::max(1, R / 64), 64);                  // ranges of 64-100 rows: a wave's task is two or three batches of loads
}
static void wg_setup(WgArgs &A, int R, int H, int H1, int H2, const float *x, int64_t ldx, const float *dz1, const float *f1, const float *dz2,
                     const float *f2, const float *g3) {
  const float *P[3] = {dz1, dz2, g3}, *Q[3] = {x, f1, f2};
  const int64_t ldp[3] = {H1, H2, 1}, ldq[3] = {ldx, H1, H2};
  const int m[3] = {H1, H2, 1}, n[3] = {H, H1, H2};
  const int weight[3] = {512, 192, 32};                      // wave tasks aimed at per product (the first one holds 9 / 10 of the arithmetic)
  int task = 0;
  int64_t part = 0;
  for (int q = 0; q < 3; ++q) {
    WgProd &W = A.p[q];
    W.P = P[q]; W.Q = Q[q]; W.ldp = ldp[q]; W.ldq = ldq[q]; W.m = m[q]; W.n = n[q];
    W.vp = std::min(2, wg_pick_vec(P[q], ldp[q], m[q], 2));
    W.vq = wg_pick_vec(Q[q], ldq[q], n[q], 4);
    if (!((W.vp == 2 && W.vq == 4) || (W.vp == 1 && W.vq == 2) || (W.vp == 1 && W.vq == 4) || (W.vp == 2 && W.vq == 2))) { W.vp = 1; W.vq = 1; }
    W.tiles_m = (m[q] + 16 * W.vp - 1) / (16 * W.vp);
    W.tiles_n = (n[q] + 16 * W.vq - 1) / (16 * W.vq);
    W.splits = wg_splits(R, W.tiles_m * W.tiles_n, weight[q]);
    W.task0 = task;
    W.part0 = part;
    task += W.tiles_m * W.tiles_n * W.splits;
    part += (int64_t)W.splits * m[q] * n[q];
  }
  A.R = R;
  A.n_tasks = task;
}

}  // namespace

// ---- the same kernel for ONE weight gradient of any layer: D (m x n, contiguous) = P^T Q over R rows (P: R x m, Q: R x n, row-major with
// strides ldp / ldq).  C++ linkage, called by ggad_gemm_f32 for its "TN" products with a long K (the GCN layers' dW = dZ^T X: model.py:27
// backwards; they ran as 64 x 64 tiles split over K + k_splitk_reduce).  Returns 1 when launched, 0 when the shape is not taken.
int64_t ggad_int_wgrad_tn_ws(int R, int m, int n) {
  // skinny outputs only (the first layer's dW at 10-25 features: 43.6 -> 32.5 us at T-Finance size); for 300 x 300 the wave tiles of 32 x 64
  // re-read the operands too often (51.5 against 48.1 us for the split-K tiles, scripts/gemm_wgrad_time.py)
  if (m < 1 || n < 1 || m > 512 || n > 512 || R < 1024 || std::min(m, n) > 32) return 0;
  return (int64_t)std::min(std::max(1, R / 64), 64) * m * n + 16;
}
int ggad_int_wgrad_tn(const float *P, int64_t ldp, const float *Q, int64_t ldq, int R, int m, int n, float *D, float *ws, hipStream_t st) {
  static const bool off = [] { const char *e = getenv("GGAD_GEMM_WGRAD_TN"); return e && e[0] == '0'; }();
  if (off || ggad_int_wgrad_tn_ws(R, m, n) == 0 || ((uintptr_t)ws & 15) != 0) return 0;
  WgArgs A;
  WgProd &W = A.p[0];
  W.P = P; W.Q = Q; W.ldp = ldp; W.ldq = ldq; W.m = m; W.n = n;
  W.vp = std::min(2, wg_pick_vec(P, ldp, m, 2));
  W.vq = wg_pick_vec(Q, ldq, n, 4);
  if (!((W.vp == 2 && W.vq == 4) || (W.vp == 1 && W.vq == 2) || (W.vp == 1 && W.vq == 4) || (W.vp == 2 && W.vq == 2))) { W.vp = 1; W.vq = 1; }
  W.tiles_m = (m + 16 * W.vp - 1) / (16 * W.vp);
  W.tiles_n = (n + 16 * W.vq - 1) / (16 * W.vq);
  W.splits = wg_splits(R, 0, 0);
  W.task0 = 0; W.part0 = 0;
  A.n_tasks = W.tiles_m * W.tiles_n * W.splits;
  A.p[1] = W; A.p[2] = W;
  A.p[1].task0 = A.p[2].task0 = A.n_tasks;                   // (no second / third product)
  A.R = R; A.ws = ws;
  k_mlp_wgrad<<<dim3((unsigned)((A.n_tasks + 3) / 4)), dim3(256), 0, st>>>(A);
  WgRed Q3;
  Q3.out[0] = D; Q3.out[1] = Q3.out[2] = nullptr;
  Q3.part0[0] = 0; Q3.part0[1] = Q3.part0[2] = 0;
  Q3.len[0] = m * n; Q3.len[1] = Q3.len[2] = 0;
  Q3.splits[0] = W.splits; Q3.splits[1] = Q3.splits[2] = 0;
  k_mlp_wgrad_reduce<<<dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, st>>>(Q3, ws);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}

extern "C" {

int32_t ggad_mlp_score_supported(int32_t H, int32_t H1, int32_t H2) {
  return (H >= 4 && H <= MLP_HMAX && (H & 3) == 0 && H1 >= 2 && H1 <= MLP_H1MAX && (H1 & 1) == 0 && H2 >= 1 && H2 <= MLP_H2MAX) ? 1 : 0;
}

int ggad_mlp_score_fwd_f32(const float *X, int64_t ldx, int32_t R, int32_t H, int32_t H1, int32_t H2, const float *W1, const float *W2,
                           const float *w3, float *f1, float *f2, float *f3, ggad_stream_t stream) {
  GGAD_REQUIRE(X && W1 && W2 && w3 && f1 && f2 && f3 && R >= 0 && ldx >= H && (ldx & 3) == 0);
  GGAD_REQUIRE(ggad_mlp_score_supported(H, H1, H2));
  GGAD_REQUIRE((((uintptr_t)X | (uintptr_t)W1) & 15) == 0 && ((uintptr_t)W2 & 7) == 0);
  if (R == 0) return GGAD_OK;
  const size_t lds = (size_t)MLP_R * (mlp_pad(H) + mlp_pad(H1) + mlp_pad(H2)) * sizeof(float);
  k_mlp_fwd<<<dim3((unsigned)((R + MLP_R - 1) / MLP_R)), dim3(256), lds, as_stream(stream)>>>(X, ldx, R, H, H1, H2, W1, W2, w3, f1, f2, f3);
  GGAD_CHECK_LAUNCH("mlp_score_fwd_f32");
  return GGAD_OK;
}

int ggad_mlp_score_dgrad_f32(const float *g3, int32_t R, int32_t H, int32_t H1, int32_t H2, const float *f1, const float *f2,
                             const float *W1, const float *W2, const float *w3, float *dz2, float *dz1, float *dx, int64_t ld_dx,
                             const float *dx_add, int64_t ld_add, ggad_stream_t stream) {
  GGAD_REQUIRE(g3 && f1 && f2 && W1 && W2 && w3 && dz2 && dz1 && dx && R >= 0 && ld_dx >= H && (!dx_add || ld_add >= H));
  GGAD_REQUIRE(ggad_mlp_score_supported(H, H1, H2));
  if (R == 0) return GGAD_OK;
  const size_t lds = (size_t)MLP_R * (mlp_pad(H1) + mlp_pad(H2)) * sizeof(float);
  k_mlp_dgrad<<<dim3((unsigned)((R + MLP_R - 1) / MLP_R)), dim3(256), lds, as_stream(stream)>>>(g3, R, H, H1, H2, f1, f2, W1, W2, w3, dz2, dz1,
                                                                                             dx, ld_dx, dx_add, ld_add);
  GGAD_CHECK_LAUNCH("mlp_score_dgrad_f32");
  return GGAD_OK;
}

int64_t ggad_mlp_score_wgrad_workspace_elems(int32_t R, int32_t H, int32_t H1, int32_t H2) {
  if (R <= 0 || !ggad_mlp_score_supported(H, H1, H2)) return 0;
  // an upper bound that does not depend on the operands' alignment (which decides the wave tiles and with them the ranges per product)
  const int64_t max_splits = std::min(std::max(1, R / 64), 64);
  return max_splits * ((int64_t)H1 * H + (int64_t)H2 * H1 + H2) + 16;
}

int ggad_mlp_score_wgrad_f32(const float *x, int64_t ldx, const float *dz1, const float *f1, const float *dz2, const float *f2, const float *g3,
                             int32_t R, int32_t H, int32_t H1, int32_t H2, float *dW1, float *dW2, float *dW3, float *workspace,
                             ggad_stream_t stream) {
  GGAD_REQUIRE(x && dz1 && f1 && dz2 && f2 && g3 && dW1 && dW2 && dW3 && workspace && R >= 1 && ldx >= H);
  GGAD_REQUIRE(ggad_mlp_score_supported(H, H1, H2) && ((uintptr_t)workspace & 15) == 0);
  WgArgs A;
  wg_setup(A, R, H, H1, H2, x, ldx, dz1, f1, dz2, f2, g3);
  A.ws = workspace;
  hipStream_t st = as_stream(stream);
  k_mlp_wgrad<<<dim3((unsigned)((A.n_tasks + 3) / 4)), dim3(256), 0, st>>>(A);
  WgRed Q;
  Q.out[0] = dW1; Q.out[1] = dW2; Q.out[2] = dW3;
  int total = 0;
  for (int q = 0; q < 3; ++q) { Q.part0[q] = A.p[q].part0; Q.len[q] = A.p[q].m * A.p[q].n; Q.splits[q] = A.p[q].splits; total += Q.len[q]; }
  k_mlp_wgrad_reduce<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(Q, workspace);
  GGAD_CHECK_LAUNCH("mlp_score_wgrad_f32");
  return GGAD_OK;
}

}  // extern "C"
