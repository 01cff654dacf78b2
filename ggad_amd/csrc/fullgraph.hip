// Sparse / element-wise kernels of the full-graph GGAD path (gfx950).
//
// The reference multiplies DENSE N x N matrices: torch.bmm(adj, X W) in every GCN layer (model.py:31),
// adj[0, abn, :] @ emb for outlier generation (model.py:151-155), and emb_norm @ emb_norm.T * raw_adj for the
// local affinity (run.py:182-188).  All three are sparse products over the edges of the graph:
//   k_spmm          CSR SpMM with optional row subset, bias, PReLU epilogue     (GCN layer, outlier rows,
//                   affinity  S = R^T e_hat,  and every backward  A^T dZ)
//   k_prelu_bwd     dZ = g * prelu'(z), with column sums (bias grad) and slope-grad partials fused
//   k_rownorm(_bwd) row L2 normalisation of the embedding (run.py:177-180) and its VJP
//   k_rowdot        aff_j = r_inv_j * <e_hat_j, S_j>                           (run.py:188)
//   k_full_loss     BCE + margin(0.7) + the axis-quirk reconstruction term and their gradients (run.py:165-210)
// One wave per output row; a row of W floats is read as float4 (W % 4 == 0, W <= 1024: W = 300 -> 75 float4,
// 2 load instructions per neighbour); neighbour ids / values are fetched 64 at a time and broadcast by
// v_readlane.  Summation order is fixed (CSR order) -> deterministic.
#include <mutex>

#include "common.h"

namespace {

constexpr int SPMM_MAXCH = 4;   // float4 chunks of 64 lanes: W <= 1024
constexpr int SPMM_SEG = 64;    // neighbours per segment (= one wave-wide index load)

typedef float spmm_f4 __attribute__((ext_vector_type(4)));
typedef int spmm_i2 __attribute__((ext_vector_type(2)));
typedef int spmm_i4 __attribute__((ext_vector_type(4)));

// NT: streaming stores (the rows leave the XCD's L2 early: the L2 of a gathering kernel is for the operand it gathers from)
template <bool NT = false>
__device__ __forceinline__ void spmm_epilogue_store(float4 z, int vi, const float *__restrict__ bias, const float *prelu_a,
                                                    float a, float *__restrict__ out_row, float *__restrict__ pre_row) {
  if (bias) {
    const float4 b = reinterpret_cast<const float4 *>(bias)[vi];
    z.x += b.x; z.y += b.y; z.z += b.z; z.w += b.w;                          // out += bias              model.py:32-33
  }
  if (pre_row) {
    if (NT) __builtin_nontemporal_store((spmm_f4){z.x, z.y, z.z, z.w}, reinterpret_cast<spmm_f4 *>(pre_row) + vi);
    else reinterpret_cast<float4 *>(pre_row)[vi] = z;
  }
  if (prelu_a) {                                                               // PReLU                    model.py:35
    z.x = z.x > 0.f ? z.x : a * z.x; z.y = z.y > 0.f ? z.y : a * z.y;
    z.z = z.z > 0.f ? z.z : a * z.z; z.w = z.w > 0.f ? z.w : a * z.w;
  }
  if (NT) __builtin_nontemporal_store((spmm_f4){z.x, z.y, z.z, z.w}, reinterpret_cast<spmm_f4 *>(out_row) + vi);
  else reinterpret_cast<float4 *>(out_row)[vi] = z;
}

// One wave per SEGMENT (<= 64 consecutive neighbours of one row): rows are split so that a hub row does not
// serialise the launch (power-law graphs: max degree ~ N/8).  seg_out[s] >= 0: the row has this single segment,
// finish it here (bias / PReLU epilogue, write out[seg_out]); seg_out[s] < 0: write the partial sum to
// part[-seg_out[s] - 1]; k_spmm_combine adds the partials of such rows in slot order (a row's slots are consecutive;
// the launch order of the segments is free, e.g. grouped by column range).
__global__ void __launch_bounds__(256) k_spmm_seg(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                  const int32_t *__restrict__ seg_beg, const int32_t *__restrict__ seg_end,
                                                  const int32_t *__restrict__ seg_out, int n_seg,
                                                  const float *__restrict__ X, int64_t ldx, int W,
                                                  const float *__restrict__ bias, const float *__restrict__ prelu_a,
                                                  float *__restrict__ out, int64_t ldo, float *__restrict__ out_pre,
                                                  float *__restrict__ part) {
  const int sidx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sidx >= n_seg) return;
  const int lane = lane_id();
  const int s = seg_beg[sidx], t = seg_end[sidx];
  const int nvec = W >> 2;
  float4 acc[SPMM_MAXCH];
#pragma unroll
  for (int c = 0; c < SPMM_MAXCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int e = s + lane;
  const int cv = (e < t) ? col[e] : 0;
  const float vv = (e < t) ? (val ? val[e] : 1.0f) : 0.0f;
  const int cnt = t - s;
  int i = 0;
  for (; i + 4 <= cnt; i += 4) {                       // 4 neighbours x up to 4 float4 loads in flight per lane
    int c[4]; float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c[u] = __builtin_amdgcn_readlane(cv, i + u);
      v[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), i + u));
    }
#pragma unroll
    for (int ch = 0; ch < SPMM_MAXCH; ++ch) {
      const int vi = ch * 64 + lane;
      if (vi < nvec) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = reinterpret_cast<const float4 *>(X + (int64_t)c[u] * ldx)[vi];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[ch].x = fmaf(v[u], x[u].x, acc[ch].x); acc[ch].y = fmaf(v[u], x[u].y, acc[ch].y);
          acc[ch].z = fmaf(v[u], x[u].z, acc[ch].z); acc[ch].w = fmaf(v[u], x[u].w, acc[ch].w);
        }
      }
    }
  }
  for (; i < cnt; ++i) {
    const int c = __builtin_amdgcn_readlane(cv, i);
    const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), i));
    const float4 *xr = reinterpret_cast<const float4 *>(X + (int64_t)c * ldx);
#pragma unroll
    for (int ch = 0; ch < SPMM_MAXCH; ++ch) {
      const int vi = ch * 64 + lane;
      if (vi < nvec) {
        const float4 x = xr[vi];
        acc[ch].x = fmaf(v, x.x, acc[ch].x); acc[ch].y = fmaf(v, x.y, acc[ch].y);
        acc[ch].z = fmaf(v, x.z, acc[ch].z); acc[ch].w = fmaf(v, x.w, acc[ch].w);
      }
    }
  }
  const int orow = seg_out[sidx];
  const float a = prelu_a ? *prelu_a : 1.0f;
#pragma unroll
  for (int ch = 0; ch < SPMM_MAXCH; ++ch) {
    const int vi = ch * 64 + lane;
    if (vi < nvec) {
      if (orow >= 0)
        spmm_epilogue_store(acc[ch], vi, bias, prelu_a, a, out + (int64_t)orow * ldo, out_pre ? out_pre + (int64_t)orow * ldo : nullptr);
      else
        reinterpret_cast<float4 *>(part + (int64_t)(-orow - 1) * W)[vi] = acc[ch];
    }
  }
}

// ---- column-sliced kernel for SPARSE neighbourhoods (Reddit, Photo: ~17 entries per row, W = 300, operand 9-13 MB) -----------
// The wave-per-segment kernel above gathers 1,200-byte rows from an operand no XCD's 4 MB L2 holds (every fetch comes from the
// Infinity Cache: 4 nnz H = 215 MB per Reddit product for 28 MB of compulsory traffic), its second load instruction of a row carries
// 11 of 64 lanes, and rows of more than 64 entries need a second launch.  Here a workgroup owns ONE slice of RS_W = 40 columns --
// slice = blockIdx % n_slices, 8 slices at W = 300, i.e. one per XCD: each L2 then serves N x 160 B = 1.8 MB, resident after the
// first touch -- and a wave works on SIX rows at once: lane (g, q) = (lane / 10, lane % 10) accumulates float4 q of the slice for row
// g of the unit, walking that row's entries sequentially (rows of a unit have similar lengths: the host sorts them by degree), so a
// load instruction carries 60 of 64 lanes, there is no cross-lane reduction and every row is finished by one lane group (epilogue in
// place, no partial sums, no combine launch).  Rows of more than RS_LONG entries (hubs) take a whole wave: six entries per load,
// the six partial sums added by shuffles.  Summation order: CSR order inside a row (short rows), six interleaved running sums
// combined in group order (long rows): deterministic.
constexpr int RS_W = 40, RS_Q40 = RS_W / 4, RS_G40 = 6;        // the 40-column slices; the line variant: 8 float4, 8 rows per wave
constexpr int RS_U = 8;                  // loads in flight per lane: the walk of a row is a chain of memory round trips
constexpr int RS_SHORT = 32;             // rows of <= 32 entries: one lane group each (<= 4 round trips)
constexpr int RS_LONG = 192;             // rows of <= 192 entries: one wave each, 48 entries per round trip; longer (hubs): a workgroup

// entries e0, e0 + step, ... < t of one row, RS_U per round trip, accumulated into acc (this lane's float4 of the slice).  The ten
// lanes of a group fetch the next ten (column, value) pairs with ONE load each and hand them round by shuffles: per 8 entries the
// vector-memory path sees 8 + 2 instructions instead of 24 (it accepts one wave-wide load per ~17 clocks per CU whatever it carries)
template <int Q>
__device__ __forceinline__ void rs_walk(const int32_t *__restrict__ col, const float *__restrict__ val, const float *__restrict__ X,
                                        int64_t ldx, int vic, int e0, int step, int t, bool lane_on, int g, int q, float4 &acc) {
  static_assert(RS_U <= Q, "a group's lanes hold one round of (column, value) pairs");
  const int gbase = g * Q;
  int eq = min(e0 + q * step, max(t - 1, 0));
  int ce = col[eq];                                                 // (column, value) pairs of the first round
  float ve = (lane_on && e0 + q * step < t) ? (val ? val[eq] : 1.0f) : 0.0f;
  for (int e = e0; e < t; e += RS_U * step) {
    int c[RS_U]; float v[RS_U]; float4 x[RS_U];
#pragma unroll
    for (int k = 0; k < RS_U; ++k) {
      c[k] = __shfl(ce, gbase + k, GGAD_WAVE);
      v[k] = __shfl(ve, gbase + k, GGAD_WAVE);
    }
#pragma unroll
    for (int k = 0; k < RS_U; ++k) x[k] = reinterpret_cast<const float4 *>(X + (int64_t)c[k] * ldx)[vic];
    const int en = e + RS_U * step;                                 // the next round's pairs travel with this round's rows
    eq = min(en + q * step, t - 1);
    ce = col[eq];
    ve = (lane_on && en + q * step < t) ? (val ? val[eq] : 1.0f) : 0.0f;
#pragma unroll
    for (int k = 0; k < RS_U; ++k) {
      acc.x = fmaf(v[k], x[k].x, acc.x); acc.y = fmaf(v[k], x[k].y, acc.y);
      acc.z = fmaf(v[k], x[k].z, acc.z); acc.w = fmaf(v[k], x[k].w, acc.w);
    }
  }
}

// blocks [0, n_slices * ublocks): 4 units each (a unit = 6 short rows, or one medium row); then n_slices blocks per hub row
template <int RS_Q, int RS_G>
__global__ void __launch_bounds__(256) k_spmm_rowslice(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                       const float *__restrict__ val, const int32_t *__restrict__ unit_rows,
                                                       const int32_t *__restrict__ unit_out, int n_units,
                                                       const int32_t *__restrict__ long_rows, const int32_t *__restrict__ long_out,
                                                       int n_long, const int32_t *__restrict__ hub_rows,
                                                       const int32_t *__restrict__ hub_out, int n_hub, int n_slices,
                                                       const float *__restrict__ X, int64_t ldx, int W,
                                                       const float *__restrict__ bias, const float *__restrict__ prelu_a,
                                                       float *__restrict__ out, int64_t ldo, float *__restrict__ out_pre) {
  __shared__ float4 hub_part[4][RS_Q];
  const int wid = threadIdx.x >> 6;
  const int ublocks = (n_units + n_long + 3) / 4;
  const int slice = blockIdx.x % n_slices, ub = blockIdx.x / n_slices;
  const int lane = lane_id();
  const int g = lane / RS_Q, q = lane - g * RS_Q;
  const int vi = slice * RS_Q + q;                                 // float4 index inside a row of X / out
  const bool lane_on = g < RS_G;
  const int vic = vi < (W >> 2) ? vi : 0;
  const int gc = lane_on ? g : 0;
  const float a = prelu_a ? *prelu_a : 1.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ub < ublocks) {
    const int u = ub * 4 + wid;
    if (u >= n_units + n_long) return;
    if (u < n_units) {                                             // six short rows, one per lane group
      const int row = lane_on ? unit_rows[u * RS_G + g] : -1;
      int e = 0, t = 0;
      if (row >= 0) { e = rowptr[row]; t = rowptr[row + 1]; }
      rs_walk<RS_Q>(col, val, X, ldx, vic, e, 1, t, lane_on, gc, q, acc);
      if (row >= 0 && vi < (W >> 2)) {
        const int orow = unit_out[u * RS_G + g];
        spmm_epilogue_store(acc, vi, bias, prelu_a, a, out + (int64_t)orow * ldo, out_pre ? out_pre + (int64_t)orow * ldo : nullptr);
      }
      return;
    }
    const int lr = u - n_units;                                    // a medium row: the six lane groups take every sixth entry
    const int row = long_rows[lr];
    rs_walk<RS_Q>(col, val, X, ldx, vic, rowptr[row] + gc, RS_G, rowptr[row + 1], lane_on, gc, q, acc);
    if (!lane_on) acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 tot = acc;                                              // group 0 + 1 + ... + 5, in that order
#pragma unroll
    for (int k = 1; k < RS_G; ++k) {
      const int src = q + k * RS_Q;
      tot.x += __shfl(acc.x, src, GGAD_WAVE); tot.y += __shfl(acc.y, src, GGAD_WAVE);
      tot.z += __shfl(acc.z, src, GGAD_WAVE); tot.w += __shfl(acc.w, src, GGAD_WAVE);
    }
    if (g == 0 && vi < (W >> 2)) {
      const int orow = long_out[lr];
      spmm_epilogue_store(tot, vi, bias, prelu_a, a, out + (int64_t)orow * ldo, out_pre ? out_pre + (int64_t)orow * ldo : nullptr);
    }
    return;
  }
  // a hub row: the 24 lane groups of the workgroup take every 24th entry; waves combined through LDS in wave order
  const int hr = ub - ublocks;
  if (hr >= n_hub) return;
  const int row = hub_rows[hr];
  rs_walk<RS_Q>(col, val, X, ldx, vic, rowptr[row] + wid * RS_G + gc, 4 * RS_G, rowptr[row + 1], lane_on, gc, q, acc);
  if (!lane_on) acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 tot = acc;
#pragma unroll
  for (int k = 1; k < RS_G; ++k) {
    const int src = q + k * RS_Q;
    tot.x += __shfl(acc.x, src, GGAD_WAVE); tot.y += __shfl(acc.y, src, GGAD_WAVE);
    tot.z += __shfl(acc.z, src, GGAD_WAVE); tot.w += __shfl(acc.w, src, GGAD_WAVE);
  }
  if (g == 0) hub_part[wid][q] = tot;
  __syncthreads();
  if (wid == 0 && g == 0 && vi < (W >> 2)) {
    float4 r = hub_part[0][q];
#pragma unroll
    for (int k = 1; k < 4; ++k) { const float4 p2 = hub_part[k][q]; r.x += p2.x; r.y += p2.y; r.z += p2.z; r.w += p2.w; }
    const int orow = hub_out[hr];
    spmm_epilogue_store(r, vi, bias, prelu_a, a, out + (int64_t)orow * ldo, out_pre ? out_pre + (int64_t)orow * ldo : nullptr);
  }
}

// ---- the same product, LINE-granular and persistent, for operands whose rows are 128-byte aligned ---------------------------------
// (row stride a multiple of 32 floats: the producers of the path pad their 300-float rows to 320, fullgraph.py::padded_rows).
// Two things the counters of k_spmm_rowslice showed (profiles/r03_spmm_reddit_pmc.csv): a 160-byte slice of a 1,200-byte row
// straddles 2-3 lines, so every XCD fetched 2.3 x the bytes it used; and 21,000 waves of 2,800 cycles each kept less than ONE wave
// per SIMD resident -- the launch was bound by the rate workgroups are dispatched at, not by memory.  Here a slice is one 128-byte
// line (8 float4: a wave holds 8 rows, all 64 lanes of a load carry data, nothing is fetched that is not used) and the grid is a
// fixed number of workgroups per XCD whose waves loop over the work items of that XCD: line x of every unit (W = 300 is 10 lines for
// 8 XCDs: the lines 8 and 9 are split by rows over the XCDs x % 2 == 0 / == 1, a quarter of the rows each, so an L2 keeps two lines
// of every row -- 2.8 MB at Reddit size -- and every XCD has 1.25 lines of work).  The host hands (first entry, end, output row) per
// row slot in one table, so a wave's chain of dependent round trips is table -> columns -> rows of X instead of unit -> rowptr ->
// columns -> rows.  Items: the medium rows (one wave each, longest first), then the units of 8 short rows (longest first); hub rows
// (a workgroup each) before them.  workgroup b runs on XCD b % 8 (dispatcher rotation).  Summation order as in k_spmm_rowslice with
// 8 groups: CSR order inside a short row; 8 (medium) or 8 x waves (hub) interleaved running sums combined in a fixed order.
constexpr int RL_G = 8;                  // rows per wave = float4 per line

__device__ __forceinline__ float4 rl_group_sum(float4 acc, int q) {          // group 0 + 1 + ... + 7, in that order
  float4 tot = acc;
#pragma unroll
  for (int k = 1; k < RL_G; ++k) {
    const int src = q + k * RL_G;
    tot.x += __shfl(acc.x, src, GGAD_WAVE); tot.y += __shfl(acc.y, src, GGAD_WAVE);
    tot.z += __shfl(acc.z, src, GGAD_WAVE); tot.w += __shfl(acc.w, src, GGAD_WAVE);
  }
  return tot;
}

typedef float rl_f2 __attribute__((ext_vector_type(2)));

// the (column, value) pair of entry e0 + q * step of this lane's row (the last entry again, with value 0, past the end t): lane q
// of a group fetches pair q of a round of 8
__device__ __forceinline__ int2 rl_pair(const int2 *__restrict__ ent, int e0, int step, int t, int q) {
  const int at = e0 + q * step;
  int2 p = ent[min(at, max(t - 1, 0))];
  if (at >= t) p.y = 0;
  return p;
}

constexpr int RL_R = 4;                  // rounds of 8 entries a wave item has at most: short rows <= 32 entries, medium rows <= 192 over 8 groups

// A wave item: entries e0, e0 + step, ... < t of this lane group's row (<= RL_R rounds of 8), accumulated into acc (this lane's
// float4 of the line).  Its pairs are ALL in registers (lane q holds pairs q, q + 8, ..., fetched while the previous item was
// walked) and go to the wave's 2 KB of LDS in one go -- (byte offset of the row of X, value) per ds_write_b64 -- so the rounds depend
// on nothing but their own loads: four ds_read_b128 give a lane the 8 pairs of its row (the 8 lanes of a group read the same
// addresses: broadcasts, no bank conflict), then 8 loads of X rows with 32-bit offsets from a uniform base, 8 adds and 16
// v_pk_fma_f32.  (The first version handed the pairs round by ds_bpermute and walked 64-bit addresses: 360 vector instructions per
// item.  Per-workgroup clocks -- scripts/rowline_clocks.py -- then showed what actually bounds the launch: the hub rows and the
// number of load instructions of the vector-memory path, 16 clocks each per CU.)
// Only the loads some lane group needs are issued: n_max = the longest of the wave's 8 rows (entries of this item), wave-uniform --
// the vector-memory path takes ~16 clocks per CU for every load instruction whatever its active lanes, and a third of the 436 K
// instructions of a Reddit-size product were the padding of short rows to a round of 8 (profiles/r04_spmm_reddit_pmc.csv).
__device__ __forceinline__ void rl_walk_item(const char *__restrict__ Xb, uint32_t ldx4, uint32_t voff, int e0, int step, int t,
                                             int lane, int g, const int2 (&pe)[RL_R], int2 *__restrict__ lds_w, float4 &acc) {
  rl_f2 a01 = {acc.x, acc.y}, a23 = {acc.z, acc.w};
  int n_max = 0;                                                   // entries of the longest row of the wave (uniform)
  {
    const int mine = t > e0 ? (t - e0 + step - 1) / step : 0;
#pragma unroll
    for (int k = 0; k < RL_G; ++k) n_max = max(n_max, __builtin_amdgcn_readlane(mine, k * RL_G));
  }
#pragma unroll
  for (int r = 0; r < RL_R; ++r) lds_w[r * 64 + lane] = make_int2((int)__umul24((uint32_t)pe[r].x, ldx4), pe[r].y);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < RL_R; ++r) {
    const int cnt = n_max - r * RL_G;                              // loads of this round that somebody needs (uniform)
    if (cnt > 0) {
      const int4 *rd = reinterpret_cast<const int4 *>(lds_w + r * 64 + g * RL_G);
      int4 p[RL_G / 2];
#pragma unroll
      for (int k = 0; k < RL_G / 2; ++k) p[k] = rd[k];
      float4 x[RL_G];
#pragma unroll
      for (int k = 0; k < RL_G / 2; ++k) {
        x[2 * k] = x[2 * k + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (2 * k < cnt) x[2 * k] = *reinterpret_cast<const float4 *>(Xb + ((uint32_t)p[k].x + voff));
        if (2 * k + 1 < cnt) x[2 * k + 1] = *reinterpret_cast<const float4 *>(Xb + ((uint32_t)p[k].z + voff));
      }
#pragma unroll
      for (int k = 0; k < RL_G / 2; ++k) {
        const float v0 = __int_as_float(p[k].y), v1 = __int_as_float(p[k].w);
        const rl_f2 s0 = {v0, v0}, s1 = {v1, v1};
        a01 = __builtin_elementwise_fma(s0, (rl_f2){x[2 * k].x, x[2 * k].y}, a01);
        a23 = __builtin_elementwise_fma(s0, (rl_f2){x[2 * k].z, x[2 * k].w}, a23);
        a01 = __builtin_elementwise_fma(s1, (rl_f2){x[2 * k + 1].x, x[2 * k + 1].y}, a01);
        a23 = __builtin_elementwise_fma(s1, (rl_f2){x[2 * k + 1].z, x[2 * k + 1].w}, a23);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();                                 // (the next item's ds_writes stay behind these reads)
  acc = make_float4(a01.x, a01.y, a23.x, a23.y);
}

#ifdef GGAD_RL_PROF
__device__ unsigned long long g_rl_prof[4 * 8192];          // per workgroup: start, hub rows done, end (wall_clock64, 100 MHz), items walked
#endif

#ifdef GGAD_RL_WAVES                     // (occupancy experiments: -DGGAD_RL_WAVES=6 asks the compiler for <= 80 VGPRs)
#define RL_OCC __attribute__((amdgpu_waves_per_eu(GGAD_RL_WAVES, GGAD_RL_WAVES)))
#else
#define RL_OCC
#endif
template <int NW>
__global__ void __launch_bounds__(NW * 64) RL_OCC k_spmm_rowline(const int2 *__restrict__ ent, const int4 *__restrict__ unit_tab, int n_units,
                                                          const int4 *__restrict__ long_tab, int n_long,
                                                          const int4 *__restrict__ hub_tab, int n_hub, int n_lines,
                                                          const float *__restrict__ X, int64_t ldx, int W,
                                                          const float *__restrict__ bias, const float *__restrict__ prelu_a,
                                                          float *__restrict__ out, int64_t ldo, float *__restrict__ out_pre) {
  __shared__ float4 hub_part[NW][RL_G];
  __shared__ int2 pair_lds[NW][RL_R * 64];
#ifdef GGAD_RL_PROF
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_rl_prof[4 * blockIdx.x] = wall_clock64();
#endif
  const int wid = threadIdx.x >> 6, lane = lane_id(), g = lane >> 3, q = lane & 7;
  const int x = blockIdx.x & 7, b = blockIdx.x >> 3, B = gridDim.x >> 3;
  const int extra = n_lines - 8;                                   // lines beyond one per XCD (0 <= extra <= 8)
  const int k = extra > 0 ? x % extra : 0, r = extra > 0 ? x / extra : 0;      // this XCD is the r-th of the XCDs k, k + extra, ... serving line 8 + k
  const int servers = extra > 0 ? (8 - k + extra - 1) / extra : 1;
  const int nv = W >> 2;
  const float a = prelu_a ? *prelu_a : 1.0f;
  const char *Xb = reinterpret_cast<const char *>(X);
  const uint32_t ldx4 = (uint32_t)ldx * 4u;
  int2 *lds_w = pair_lds[wid];
  // wave items, software-pipelined: while item i is walked, the pair of the first round of item i + stride and the table entry of
  // item i + 2 stride are in flight -- the chain of dependent round trips per item is the rounds of X rows alone
  const int NU = n_long + n_units;
  const int share = extra > 0 ? (NU + servers - 1) / servers : 0;
  const int n_items = NU + share, stride = B * NW;
  auto unit_of = [&](int i, int &slice) -> int {                   // -1: nothing (past the end, or past this XCD's share of an extra line)
    slice = x;
    if (i >= n_items) return -1;
    if (i < NU) return i;
    slice = 8 + k;
    const int u = r * share + (i - NU);
    return u < NU ? u : -1;
  };
  auto fetch_tab = [&](int i) -> int4 {                            // (first entry of this lane group, end, output row, step)
    int slice;
    const int u = unit_of(i, slice);
    if (u < 0) return make_int4(0, 0, -1, 1);
    if (u < n_long) { int4 m = long_tab[u]; m.x += g; m.w = RL_G; return m; }
    int4 m = unit_tab[(int64_t)(u - n_long) * RL_G + g];
    m.w = 1;
    return m;
  };
  const int i0 = b * NW + wid;
  {                                                                // hub rows (BEFORE the item pipeline is primed, round 6): the 8 x NW lane groups of the
    const int hshare = extra > 0 ? (n_hub + servers - 1) / servers : 0;      // workgroup take every (8 NW)-th entry, 8 NW x 8 RL_R entries per pass
    for (int i = b; i < n_hub + hshare; i += B) {
      int slice = x, h = i;
      if (i >= n_hub) { slice = 8 + k; h = r * hshare + (i - n_hub); if (h >= n_hub) break; }
      const int4 m = hub_tab[h];
      const int vi = slice * RL_G + q, vic = vi < nv ? vi : 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int step = NW * RL_G, pass = step * RL_G * RL_R;
      // round 6: (1) the loop runs on the WAVE's first entry (uniform: a lane group past the end walks padded pairs -- value 0 -- instead
      // of leaving on its own, so the readlane of rl_walk_item's n_max never meets an inactive lane: ADVICE r5);  (2) the pairs of pass
      // p + 1 are requested before pass p is walked -- a pass was pairs -> LDS -> rows of X -> adds, two dependent round trips, and the
      // longest hub row (1,373 entries at Reddit size = six passes of one workgroup) is the launch's critical path.  The 8 registers of
      // that prefetch come from priming the item pipeline AFTER the hub rows (on top of it they made 106 VGPRs: four waves per SIMD
      // instead of five, the launch 5 % slower).
      const int eb0 = m.x + wid * RL_G;
      int2 ph[RL_R];
#pragma unroll
      for (int rr = 0; rr < RL_R; ++rr) ph[rr] = rl_pair(ent, eb0 + g + rr * RL_G * step, step, m.y, q);
      for (int eb = eb0; eb < m.y; eb += pass) {
        const int e0 = eb + g;
        int2 pn[RL_R];
#pragma unroll
        for (int rr = 0; rr < RL_R; ++rr) pn[rr] = rl_pair(ent, e0 + pass + rr * RL_G * step, step, m.y, q);
        rl_walk_item(Xb, ldx4, (uint32_t)vic * 16u, e0, step, m.y, lane, g, ph, lds_w, acc);
#pragma unroll
        for (int rr = 0; rr < RL_R; ++rr) ph[rr] = pn[rr];
      }
      const float4 tot = rl_group_sum(acc, q);
      if (g == 0) hub_part[wid][q] = tot;
      __syncthreads();
      if (wid == 0 && g == 0 && vi < nv) {
        float4 t = hub_part[0][q];
#pragma unroll
        for (int w = 1; w < NW; ++w) { const float4 p2 = hub_part[w][q]; t.x += p2.x; t.y += p2.y; t.z += p2.z; t.w += p2.w; }
        spmm_epilogue_store(t, vi, bias, prelu_a, a, out + (int64_t)m.z * ldo, out_pre ? out_pre + (int64_t)m.z * ldo : nullptr);
      }
      __syncthreads();
    }
  }
#ifdef GGAD_RL_PROF
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_rl_prof[4 * blockIdx.x + 1] = wall_clock64();
  int n_walked = 0;
#endif
  int4 m0 = fetch_tab(i0), m1 = fetch_tab(i0 + stride);
  int2 pe[RL_R];
#pragma unroll
  for (int rr = 0; rr < RL_R; ++rr) pe[rr] = rl_pair(ent, m0.x + rr * RL_G * m0.w, m0.w, m0.y, q);
  for (int i = i0; i < n_items; i += stride) {
#ifdef GGAD_RL_PROF
    ++n_walked;
#endif
    int slice;
    const int u = unit_of(i, slice);
    const int vi = slice * RL_G + q, vic = vi < nv ? vi : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // (inside: this item's pairs go to LDS, then the loads below -- next item's pairs, the table entry after it -- are in flight
    // behind the rounds' loads of X rows)
    const int4 m2 = fetch_tab(i + 2 * stride);
    int2 pn[RL_R];
#pragma unroll
    for (int rr = 0; rr < RL_R; ++rr) pn[rr] = rl_pair(ent, m1.x + rr * RL_G * m1.w, m1.w, m1.y, q);
    rl_walk_item(Xb, ldx4, (uint32_t)vic * 16u, m0.x, m0.w, m0.y, lane, g, pe, lds_w, acc);
    if (u >= 0 && u < n_long) {                                    // a medium row: the 8 lane groups took every 8th entry
      const float4 tot = rl_group_sum(acc, q);
      if (g == 0 && vi < nv)
        spmm_epilogue_store(tot, vi, bias, prelu_a, a, out + (int64_t)m0.z * ldo, out_pre ? out_pre + (int64_t)m0.z * ldo : nullptr);
    } else if (m0.z >= 0 && vi < nv) {                             // 8 short rows, one per lane group (an empty slot: m.z < 0, no entries)
      spmm_epilogue_store(acc, vi, bias, prelu_a, a, out + (int64_t)m0.z * ldo, out_pre ? out_pre + (int64_t)m0.z * ldo : nullptr);
    }
    m0 = m1; m1 = m2;
#pragma unroll
    for (int rr = 0; rr < RL_R; ++rr) pe[rr] = pn[rr];
  }
#ifdef GGAD_RL_PROF
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x < 8192) { g_rl_prof[4 * blockIdx.x + 2] = wall_clock64(); g_rl_prof[4 * blockIdx.x + 3] = n_walked; }
#endif
}

// ---- XCD-sliced variant for dense neighbourhoods (T-Finance: 470 neighbours per row, X = 47 MB) -----------------------
// Measured on MI355X (scripts/spmm_locality_probe.py): the vector-memory path accepts ONE wave-wide load instruction per
// ~17 clocks per CU whatever its number of active lanes (W = 260 costs what W = 512 costs), and a 47 MB operand that every
// XCD gathers from misses its 4 MB L2 and comes from the Infinity Cache at ~8 TB/s (2.6 ms per product against 1.3 ms with
// an L2-resident operand).  So: (1) X is re-laid slice-major, XS[slice][row][8 float4] -- one 128-byte line per (row, slice)
// -- and a workgroup gathers ONE slice, chosen from blockIdx % 8 (the dispatcher places block b on XCD b % 8), so each L2
// serves N x 128 B instead of N x 1200 B; (2) the 64 lanes of a load carry 8 neighbours x 8 float4, so a 64-neighbour
// segment of one slice takes 8 full-width loads (80 per segment for W = 300, against 128 mostly narrow ones).
// Slices are dealt to the XCDs in rounds of 8; the slices of a last incomplete round (2 for W = 300) are each shared by
// several XCDs, which split the segments between them.
constexpr int SPMM_XCDS = 8;
constexpr int SPMM_SL = 8;                          // float4 lanes per (row, slice): 32 floats = one cache line

__host__ __device__ inline int spmm_n_slices(int W) { return ((W >> 2) + SPMM_SL - 1) / SPMM_SL; }

__global__ void __launch_bounds__(256) k_slice_major(const float *__restrict__ X, int64_t ldx, int n_rows, int W, int n_slices,
                                                     const float *__restrict__ row_scale, float4 *__restrict__ XS) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = n_slices * SPMM_SL;
  if (t >= (int64_t)n_rows * per_row) return;
  const int r = (int)(t / per_row), q = (int)(t - (int64_t)r * per_row);       // q = slice * 8 + j = float4 index in the row
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < (W >> 2)) v = reinterpret_cast<const float4 *>(X + (int64_t)r * ldx)[q];
  if (row_scale) { const float c = row_scale[r]; v.x *= c; v.y *= c; v.z *= c; v.w *= c; }
  XS[((int64_t)(q >> 3) * n_rows + r) * SPMM_SL + (q & 7)] = v;
}

constexpr int SPMM_LONG_SEG = 512;                  // segment length of the sliced plan: 8 index loads per wave, ONE reduction

// number of blocks of the sliced launch: full rounds use U = ceil(n_seg / 4) units per XCD, the last incomplete round
// ceil(U / m) units, m = the smallest number of XCDs sharing one of its slices
__host__ inline int64_t spmm_sliced_blocks(int n_seg, int n_slices) {
  const int64_t U = (n_seg + 3) / 4;
  const int full = n_slices / SPMM_XCDS, R = n_slices % SPMM_XCDS;
  const int m = R ? SPMM_XCDS / R : 1;              // slice r of the round is served by the XCDs x with x % R == r: >= 8 / R of them
  return SPMM_XCDS * (full * U + (R ? (U + m - 1) / m : 0));
}

__global__ void __launch_bounds__(256) k_spmm_sliced(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                     const int32_t *__restrict__ seg_beg, const int32_t *__restrict__ seg_end,
                                                     const int32_t *__restrict__ seg_out, int n_seg,
                                                     const float4 *__restrict__ XS, int n_src, int n_slices, int W,
                                                     const float *__restrict__ bias, const float *__restrict__ prelu_a,
                                                     float *__restrict__ out, int64_t ldo, float *__restrict__ out_pre,
                                                     float *__restrict__ part) {
  const int xcd = blockIdx.x & (SPMM_XCDS - 1);
  const int64_t unit = blockIdx.x >> 3;
  const int64_t U = (n_seg + 3) / 4;
  const int full = n_slices / SPMM_XCDS, R = n_slices - full * SPMM_XCDS;
  int slice; int64_t group;
  if (unit < (int64_t)full * U) {                   // complete rounds: XCD x owns slice round * 8 + x
    const int round = (int)(unit / U);
    slice = round * SPMM_XCDS + xcd;
    group = unit - (int64_t)round * U;
  } else {                                          // incomplete round: slice r shared by the XCDs x % R == r
    const int r = xcd % R, k = xcd / R;
    const int nx = (SPMM_XCDS - r + R - 1) / R;     // XCDs serving slice r
    slice = full * SPMM_XCDS + r;
    group = (unit - (int64_t)full * U) * nx + k;
  }
  const int64_t sidx64 = group * 4 + (int64_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (sidx64 >= n_seg) return;
  const int sidx = (int)sidx64;
  const int lane = lane_id();
  const int g = lane >> 3, j = lane & 7;            // 8 neighbours per load instruction, 8 float4 each
  const float4 *__restrict__ xs = XS + (int64_t)slice * n_src * SPMM_SL + j;
  const int s = seg_beg[sidx], t = seg_end[sidx];
  const int nch = (t - s + GGAD_WAVE - 1) / GGAD_WAVE;           // chunks of 64 neighbours (any segment length)
  // indices / values of a chunk: one coalesced load each from a clamped (always valid) position; the RAW values stay in
  // registers and are masked only when the chunk's row loads are issued, so that nothing waits on them inside the pipeline
  const float *__restrict__ vsrc = val ? val : reinterpret_cast<const float *>(col);
#define SPMM_IDX(C_, V_, CH_)                                                         \
  {                                                                                   \
    const int e_ = s + (CH_) * GGAD_WAVE + lane;                                      \
    const int ec_ = e_ < t ? e_ : t - 1;                                              \
    C_ = col[ec_];                                                                    \
    V_ = vsrc[ec_];                                                                   \
  }
  // the 8 row loads of a chunk, unconditional (a guarded load costs a branch + vmcnt(0) each); lanes past the segment's end
  // gather row 0 with value 0.0f
#define SPMM_ISSUE(X_, V_, C_, VV_, CH_)                                              \
  {                                                                                   \
    const bool in_ = s + (CH_) * GGAD_WAVE + lane < t;                                \
    const int cm_ = in_ ? C_ : 0;                                                     \
    const float vm_ = in_ ? (val ? VV_ : 1.0f) : 0.0f;                                \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                   \
      const int k = u * 8 + g;                                                        \
      const int c = __shfl(cm_, k, GGAD_WAVE);                                        \
      V_[u] = __shfl(vm_, k, GGAD_WAVE);                                              \
      X_[u] = xs[(int64_t)c * SPMM_SL];                                               \
    }                                                                                 \
  }
#define SPMM_CONSUME(X_, V_, CH_)                                                     \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                     \
    const bool ok = (CH_) * GGAD_WAVE + u * 8 + g < t - s;   /* lanes past the end loaded row 0: drop it (it may hold NaN) */ \
    acc.x = fmaf(V_[u], ok ? X_[u].x : 0.f, acc.x); acc.y = fmaf(V_[u], ok ? X_[u].y : 0.f, acc.y); \
    acc.z = fmaf(V_[u], ok ? X_[u].z : 0.f, acc.z); acc.w = fmaf(V_[u], ok ? X_[u].w : 0.f, acc.w); \
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 xa[8], xb[8]; float va[8], vb[8];
  int c0 = 0, c1 = 0; float w0 = 0.f, w1 = 0.f;
  if (nch > 0) SPMM_IDX(c0, w0, 0)                  // (an empty row keeps row 0 / 0.0f: nothing is accumulated)
  if (nch > 1) SPMM_IDX(c1, w1, 1)
  SPMM_ISSUE(xa, va, c0, w0, 0)
  for (int ch = 0; ch < nch; ch += 2) {             // software pipeline: chunk ch + 1 is in flight while chunk ch is accumulated
    if (ch + 2 < nch) SPMM_IDX(c0, w0, ch + 2)
    if (ch + 1 < nch) SPMM_ISSUE(xb, vb, c1, w1, ch + 1)
    SPMM_CONSUME(xa, va, ch)
    if (ch + 3 < nch) SPMM_IDX(c1, w1, ch + 3)
    if (ch + 2 < nch) SPMM_ISSUE(xa, va, c0, w0, ch + 2)
    if (ch + 1 < nch) SPMM_CONSUME(xb, vb, ch + 1)
  }
#undef SPMM_IDX
#undef SPMM_ISSUE
#undef SPMM_CONSUME
  // sum of the 8 lane groups (fixed butterfly over lane bits 3..5)
#pragma unroll
  for (int off = 8; off < GGAD_WAVE; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, GGAD_WAVE); acc.y += __shfl_xor(acc.y, off, GGAD_WAVE);
    acc.z += __shfl_xor(acc.z, off, GGAD_WAVE); acc.w += __shfl_xor(acc.w, off, GGAD_WAVE);
  }
  const int vi = slice * SPMM_SL + j;
  if (g != 0 || vi >= (W >> 2)) return;
  const int orow = seg_out[sidx];
  const float a = prelu_a ? *prelu_a : 1.0f;
  if (orow >= 0)
    spmm_epilogue_store(acc, vi, bias, prelu_a, a, out + (int64_t)orow * ldo, out_pre ? out_pre + (int64_t)orow * ldo : nullptr);
  else
    reinterpret_cast<float4 *>(part + (int64_t)(-orow - 1) * W)[vi] = acc;
}

// ---- LDS-panel variant for dense neighbourhoods whose values factor as  val[i][j] = rs[i] * cs[j]  (+ a diagonal) -------
// (normalize_adj: D^-1/2 A D^-1/2 (+ I), utils.py:47-54 -- every full N x N x H product of an epoch.)  The sliced kernel above
// fetches one 128-byte line per (entry, slice) through the vector L1 and runs at that path's rate.  Here a workgroup of 16
// waves owns (slice, row block) and walks the operand in PANELS of 1,270 consecutive source rows x 32 floats staged in LDS
// (159 KB + two zero rows: all of a CU's LDS): every entry then costs one ds_read_b128 of 8 lanes instead of a global line.  The operand is
// pre-scaled by cs[] while it is re-laid slice-major, the sum is scaled by rs[] in the epilogue, so the entry stream
// carries no values: it is a host-built list of 16-bit panel row indices, packed per (wave, panel, round of 8 rows) in steps of 8
// entries -- lane group g = lane / 8 accumulates row g of the round, a shorter row is padded with the index of a zero row -- and
// read 8 steps (an oct) at a time (one 16-byte load per lane, the 8 lanes of a group share it).  A wave keeps its <= 8 rounds
// (64 rows x 32 floats) in registers across all panels; rows are dealt to rounds in order of degree (the 8 rows of a round
// have similar lengths: 72-80 % of the step slots carry an entry on the power-law graphs), a hub row gets a WIDE round of its
// own (its entries over all 8 lane groups, the 8 accumulators added in the epilogue), and rounds are dealt to workgroups and
// waves by their work (csrc/spmm_panel_build.cpp, Csr.panel_plan).  Summation order: fixed by the plan -> deterministic.
typedef float pan_f4 __attribute__((ext_vector_type(4)));
constexpr int PAN_R = 1270;                         // source rows per LDS panel (with two zero rows: 162,816 of the 163,840 bytes)
constexpr int PAN_WAVES = 16;                       // waves per workgroup
constexpr int PAN_KR = 8;                           // rounds (of 8 rows) per wave at most
constexpr int PAN_LDS = (PAN_R + 2) * SPMM_SL * 16; // bytes: panel + two zero rows (one per bank half)

__global__ void __launch_bounds__(1024) k_spmm_panel(const int32_t *__restrict__ wg_tab, const uint32_t *__restrict__ dir,
                                                     const uint4 *__restrict__ stream, const int32_t *__restrict__ row_tab,
                                                     int n_chunks, const float4 *__restrict__ XS, int n_src, int W,
                                                     const float *__restrict__ row_scale, const float *__restrict__ diag,
                                                     const float *__restrict__ X, int64_t ldx, const float *__restrict__ bias,
                                                     const float *__restrict__ prelu_a, float *__restrict__ out, int64_t ldo,
                                                     float *__restrict__ out_pre) {
  extern __shared__ __attribute__((aligned(16))) float4 panel[];
  const int slice = wg_tab[2 * blockIdx.x], block = wg_tab[2 * blockIdx.x + 1];
  if (slice < 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 3, j = lane & 7;
  if (tid < 2 * SPMM_SL) panel[PAN_R * SPMM_SL + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[PAN_KR];
#pragma unroll
  for (int k = 0; k < PAN_KR; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 *__restrict__ xs = XS + (int64_t)slice * n_src * SPMM_SL;
  const uint32_t *__restrict__ d = dir + (int64_t)(block * PAN_WAVES + wave) * n_chunks * 8;
  const uint32_t lane_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)panel + j * 16, row_bytes = SPMM_SL * 16;
  // entry stream of this wave: 16-bit panel row indices, one 16-byte load per lane and OCT (8 steps), contiguous over rounds
  // and panels; hipcc unrolls the oct loop by two and keeps two loads in flight (vmcnt(1)).  Measured and dropped: 32-bit
  // offsets in quads of 4 steps (fill 0.70 instead of 0.64, but the loop runs at the latency of its stream loads: +7 %), four
  // quads in flight through a rotating register set (the moves wait for the youngest load), LDS reads one oct ahead of the
  // additions (same reason: 735 -> 1,185 us).
  const uint4 *__restrict__ sp = stream + (int64_t)d[0] * 8 + g;
  uint4 o = sp[0], o2 = sp[8];                       // the two octs to come
  for (int c = 0; c < n_chunks; ++c) {
    const int left = n_src - c * PAN_R;
    const int nf4 = (left < PAN_R ? left : PAN_R) * SPMM_SL;
    const float4 *__restrict__ src = xs + (int64_t)c * PAN_R * SPMM_SL;
    if (nf4 == PAN_R * SPMM_SL) {                    // the loads of the next panel fly while the slower waves finish this one
      constexpr int LAST = PAN_R * SPMM_SL - 9 * 1024;                          // float4 of the tenth pass
      static_assert(LAST > 0 && LAST <= 1024, "staging is written for 9 full passes of 1,024 float4 and a partial one");
      const float4 t0 = src[tid], t1 = src[1024 + tid], t2 = src[2048 + tid], t3 = src[3072 + tid], t4 = src[4096 + tid];
      const float4 t5 = src[5120 + tid], t6 = src[6144 + tid], t7 = src[7168 + tid], t8 = src[8192 + tid];
      const float4 t9 = src[tid < LAST ? 9216 + tid : 0];
      __syncthreads();                               // the previous panel has been consumed
      panel[tid] = t0; panel[1024 + tid] = t1; panel[2048 + tid] = t2; panel[3072 + tid] = t3; panel[4096 + tid] = t4;
      panel[5120 + tid] = t5; panel[6144 + tid] = t6; panel[7168 + tid] = t7; panel[8192 + tid] = t8;
      if (tid < LAST) panel[9216 + tid] = t9;
    } else {
      __syncthreads();
      for (int i = tid; i < nf4; i += 1024) panel[i] = src[i];
    }
    __syncthreads();
    // address of a row = index * 128 + lane part, from either half of a stream word in ONE instruction (v_mad_u32_u16; hipcc
    // takes and / shift / add -- 2.7 instead of 1 VALU per step, and the loop is bound by issue slots as much as by LDS)
#define PAN_STEP(WORD_, HI_)                                                                  \
  {                                                                                           \
    uint32_t ad_;                                                                             \
    if (HI_) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(ad_) : "v"(WORD_), "v"(row_bytes), "v"(lane_off)); \
    else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(ad_) : "v"(WORD_), "v"(row_bytes), "v"(lane_off)); \
    const pan_f4 x = *(const __attribute__((address_space(3))) pan_f4 *)(uintptr_t)(ad_);              \
    acc[k].x += x.x; acc[k].y += x.y; acc[k].z += x.z; acc[k].w += x.w;                       \
  }
#define PAN_OCT(O_)                                                                           \
  PAN_STEP(O_.x, 0) PAN_STEP(O_.x, 1) PAN_STEP(O_.y, 0) PAN_STEP(O_.y, 1)                      \
  PAN_STEP(O_.z, 0) PAN_STEP(O_.z, 1) PAN_STEP(O_.w, 0) PAN_STEP(O_.w, 1)
#pragma unroll
    for (int k = 0; k < PAN_KR; ++k) {
      // the directory counts QUADS (4 steps): whole octs, then -- when the longest row of the tile ends in the first half of its last
      // oct -- only that half (a tenth of the steps were padding of the second half)
      const int nh = (int)((d[c * 8 + 1 + (k >> 1)] >> ((k & 1) * 16)) & 0xffffu);
      const int nq = nh >> 1;
      int q = 0;
      for (; q + 2 <= nq; q += 2) {                  // two octs per trip: their successors are loaded a whole trip ahead
        const uint4 n1 = sp[16], n2 = sp[24];
        sp += 16;
        PAN_OCT(o) PAN_OCT(o2)
        o = n1; o2 = n2;
      }
      if (q < nq) {
        const uint4 n1 = sp[16];
        sp += 8;
        PAN_OCT(o)
        o = o2; o2 = n1;
      }
      if (nh & 1) {
        const uint4 n1 = sp[16];
        sp += 8;
        PAN_STEP(o.x, 0) PAN_STEP(o.x, 1) PAN_STEP(o.y, 0) PAN_STEP(o.y, 1)
        o = o2; o2 = n1;
      }
    }
#undef PAN_OCT
#undef PAN_STEP
  }
  const int vi = slice * SPMM_SL + j;
  if (vi >= (W >> 2)) return;
  const float a = prelu_a ? *prelu_a : 1.0f;
  const int32_t *__restrict__ rt = row_tab + (int64_t)(block * PAN_WAVES + wave) * PAN_KR * 8 + g;
#pragma unroll
  for (int k = 0; k < PAN_KR; ++k) {
    const int rowv = rt[k * 8];
    if (rowv < 0) continue;                          // (the 8 slots of a round are all set or -- past the last row -- a tail of -1)
    float4 z = acc[k];
    const int row = rowv & ~GGAD_SPMM_PANEL_WIDE;
    if (rowv & GGAD_SPMM_PANEL_WIDE) {               // wide round: ONE row over the 8 lane groups (same flag in all 8 slots)
#pragma unroll
      for (int off = 8; off < GGAD_WAVE; off <<= 1) {                           // fixed butterfly over lane bits 3..5
        z.x += __shfl_xor(z.x, off, GGAD_WAVE); z.y += __shfl_xor(z.y, off, GGAD_WAVE);
        z.z += __shfl_xor(z.z, off, GGAD_WAVE); z.w += __shfl_xor(z.w, off, GGAD_WAVE);
      }
      if (g != 0) continue;
    }
    if (row_scale) { const float r = row_scale[row]; z.x *= r; z.y *= r; z.z *= r; z.w *= r; }
    if (diag) {
      const float dv = diag[row];
      const float4 x = reinterpret_cast<const float4 *>(X + (int64_t)row * ldx)[vi];
      z.x = fmaf(dv, x.x, z.x); z.y = fmaf(dv, x.y, z.y); z.z = fmaf(dv, x.z, z.z); z.w = fmaf(dv, x.w, z.w);
    }
    spmm_epilogue_store(z, vi, bias, prelu_a, a, out + (int64_t)row * ldo, out_pre ? out_pre + (int64_t)row * ldo : nullptr);
  }
}

// ---- LDS-RING variant of the panel product (round 4): staging overlapped with the walk, flexible schedule ----------------------
// k_spmm_panel above fills the whole LDS with one 1,270-row panel, so nothing can be staged while it is walked (two barriers per
// panel, 35 % of the kernel), and pads every (round, panel) tile to its longest row (27 % of the step slots empty).  Here the
// operand slice passes through a RING of RING_S slots of RING_RS rows: during phase j the walkers read the slots j .. j+V-1 while a
// LOADER wave (wave 15: LDS-DMA, global_load_lds_dwordx4, no registers, its own vmcnt) fetches slot j+S-1 into the buffer phase
// j-1 released; ONE barrier per phase hands the buffer over.  The schedule is built on the host (csrc/spmm_ring_build.cpp): a
// lane group that has no open entry in the slot that is about to be overwritten walks its next entries anywhere in the window,
// so the 8 rows of a round drift apart by up to a window instead of being padded per panel (0.73 -> 0.85 of the steps carry an
// entry).  The price of small slots is tiny tiles (4 steps per (round, phase) on average), so the walk is FLAT: a per-wave list of
// quads (4 steps x 8 lane groups), each tagged with its accumulator; the accumulator is selected with the VGPR index mode
// (s_set_gpr_idx_on: M0-relative destination / second source of the v_pk_add_f32 -- VOP3P takes the index too, scripts/gpr_idx_probe.hip) instead of one unrolled loop nest per
// accumulator -- no per-tile loop overhead, reads of quad q + 1 in flight while quad q is added.  The walk is one asm statement
// with pinned registers (the 64 accumulators v[64:127] are its outputs); everything around it is plain HIP.
#ifndef GGAD_RING_S
#define GGAD_RING_S 3                // (round 5: 3 x 416 instead of 5 x 248 -- the whole-matrix products are within +-2 % of each other, the products over
#define GGAD_RING_RS 416             //  a row subset pay per PHASE: 95 instead of 159 of them at T-Finance size; epochs 1.950 -> 1.929 / 0.698 -> 0.681 ms)
#endif
constexpr int RING_S = GGAD_RING_S;                 // ring slots
constexpr int RING_RS = GGAD_RING_RS;               // source rows per slot (a multiple of 8: whole 1 KB LDS-DMA pieces)
constexpr int RING_V = RING_S - 1;                  // slots readable during a phase
constexpr int RING_WALK = 15;                       // walker waves (wave 15 loads)
#ifndef GGAD_RING_WALK_SUBSET
#define GGAD_RING_WALK_SUBSET 13
#endif
constexpr int RING_WALK_SUBSET = GGAD_RING_WALK_SUBSET;    // ... of the loader-bound products (round 6): waves 13..15 load
constexpr int RING_KR = 16;                         // rounds (of 8 lane rows) per walker: 64 accumulator registers
constexpr int RING_LDS = (RING_S * RING_RS + 2) * SPMM_SL * 16;   // ring + two zero rows
static_assert(RING_LDS <= 160 * 1024 && RING_RS % 8 == 0, "ring must fit the 160 KB of a CU; slots are whole 1 KB pieces");
typedef float ring_v32f __attribute__((ext_vector_type(32)));

// (timing experiments only, scripts/ring_variants.sh: -DRING_NO_BARRIER / -DRING_NO_ADD / -DRING_NO_LOAD / -DRING_NO_IDX give wrong results)
#ifdef RING_NO_IDX
#define RING_IDX_ON ""
#define RING_IDX_OFF ""
#else
#define RING_IDX_ON "s_set_gpr_idx_on s42, 0xa\n"
#define RING_IDX_OFF "s_set_gpr_idx_off\n"
#endif
#ifdef RING_NO_ADD
#define RING_ADD2(AB_, CD_) ""
#else
#define RING_ADD2(AB_, CD_) "v_pk_add_f32 v[64:65], " AB_ ", v[64:65]\n v_pk_add_f32 v[66:67], " CD_ ", v[66:67]\n"
#endif
#ifdef RING_NO_STREAM_ADV
#define RING_ADV "s_add_u32 s38, s38, 4\n s_addc_u32 s39, s39, 0\n"          /* every walker re-reads its first super-block of indices */
#else
#define RING_ADV "s_add_u32 s36, s36, 0x100\n s_addc_u32 s37, s37, 0\n s_add_u32 s38, s38, 4\n s_addc_u32 s39, s39, 0\n"
#endif
#ifdef RING_NO_BARRIER
#define RING_BARRIER ""
#define RING_LOADER_BARRIER()
#else
#define RING_BARRIER "s_barrier\n"
#define RING_LOADER_BARRIER() __builtin_amdgcn_s_barrier()
#endif
// The walk is a software pipeline over PAIRS of steps, 4 pair slots of 8 registers (v[16:47]): at pair time t the two reads of pair t
// have returned (at most the 6 of the pairs t+1 .. t+3 are outstanding), they are added into the accumulator s42 selects, and the reads
// of pair t + 4 -- two quads ahead in the list -- go into the slot that just became free: 6 to 8 reads of every walker in flight at
// all times.  A quad whose control byte has bit 6 set ends a phase on the READ side: after its last read is issued the walker waits
// for all its reads and meets the barrier, then goes on reading (the additions of the two quads before run on behind it).
#define RING_SLOT0(F_) F_("v[16:17]", "v[18:19]", "v[20:21]", "v[22:23]", "v[16:19]", "v[20:23]")
#define RING_SLOT1(F_) F_("v[24:25]", "v[26:27]", "v[28:29]", "v[30:31]", "v[24:27]", "v[28:31]")
#define RING_SLOT2(F_) F_("v[32:33]", "v[34:35]", "v[36:37]", "v[38:39]", "v[32:35]", "v[36:39]")
#define RING_SLOT3(F_) F_("v[40:41]", "v[42:43]", "v[44:45]", "v[46:47]", "v[40:43]", "v[44:47]")
#define RING_ADDS_(A_, B_, C_, D_, R0_, R1_) RING_IDX_ON RING_ADD2(A_, B_) RING_ADD2(C_, D_) RING_IDX_OFF
#define RING_READS_(A_, B_, C_, D_, R0_, R1_) "ds_read_b128 " R0_ ", v8\n ds_read_b128 " R1_ ", v9\n"
// A: the additions of the pair in SLOT_ (LGKM_ = reads that may still be outstanding), into the accumulator of control byte BFE_ of CTL_
#define RING_A(SLOT_, LGKM_, CTL_, BFE_) "s_bfe_u32 s42, " CTL_ ", " BFE_ "\n s_waitcnt lgkmcnt(" LGKM_ ")\n" SLOT_(RING_ADDS_)
// R: addresses (16-bit LDS row index * 128 + lane part, from either half of the stream word W_) and reads of a pair into SLOT_
#define RING_R(SLOT_, W_) "v_mad_u32_u16 v8, " W_ ", s43, v14\n v_mad_u32_u16 v9, " W_ ", s43, v14 op_sel:[1,0,0,0]\n" SLOT_(RING_READS_)
// F: end of a phase after the reads of a quad whose control byte (bit BIT_ of CTL_) says so
#define RING_F(CTL_, BIT_) "s_bitcmp1_b32 " CTL_ ", " BIT_ "\n s_cbranch_scc0 9f\n s_waitcnt lgkmcnt(0)\n" RING_BARRIER "9:\n"
// one super-block (4 quads = 8 pair times): words C4_ .. C7_ (quads 2, 3: still to be read) and control s41 of the current super-block,
// N0_ .. N3_ (quads 0, 1) and control CTLV_ -> s44 of the next one.  The stream of the super-block after next is requested as early as
// its registers are free: first half + control at the top (LOAD0_), second half after the current quads 2, 3 have been read (LOAD1_):
// every stream load has 1.5 super-blocks (6 quads) of walk to arrive.  Loads return in order: vmcnt(3) leaves the 3 younger ones out.
#define RING_BODY(C4_, C5_, C6_, C7_, N0_, N1_, N2_, N3_, CTLV_, LOAD0_, LOAD1_)                                     \
  "s_waitcnt vmcnt(3)\n" RING_ADV LOAD0_                                                                             \
  RING_A(RING_SLOT0, "6", "s41", "0x60000") RING_R(RING_SLOT0, C4_)                                                  \
  RING_A(RING_SLOT1, "6", "s41", "0x60000") RING_R(RING_SLOT1, C5_) RING_F("s41", "22")                              \
  RING_A(RING_SLOT2, "6", "s41", "0x60008") RING_R(RING_SLOT2, C6_)                                                  \
  RING_A(RING_SLOT3, "6", "s41", "0x60008") RING_R(RING_SLOT3, C7_) RING_F("s41", "30")                              \
  "s_waitcnt vmcnt(3)\n v_readfirstlane_b32 s44, " CTLV_ "\n" LOAD1_                                                 \
  RING_A(RING_SLOT0, "6", "s41", "0x60010") RING_R(RING_SLOT0, N0_)                                                  \
  RING_A(RING_SLOT1, "6", "s41", "0x60010") RING_R(RING_SLOT1, N1_) RING_F("s44", "6")                               \
  RING_A(RING_SLOT2, "6", "s41", "0x60018") RING_R(RING_SLOT2, N2_)                                                  \
  RING_A(RING_SLOT3, "6", "s41", "0x60018") RING_R(RING_SLOT3, N3_) RING_F("s44", "14")                              \
  "s_mov_b32 s41, s44\n"
// the last super-block of a walker: its quads 2, 3 are read, nothing after them
#define RING_TAIL(C4_, C5_, C6_, C7_)                                                                                \
  "s_waitcnt vmcnt(3)\n"                                                                                             \
  RING_A(RING_SLOT0, "6", "s41", "0x60000") RING_R(RING_SLOT0, C4_)                                                  \
  RING_A(RING_SLOT1, "6", "s41", "0x60000") RING_R(RING_SLOT1, C5_) RING_F("s41", "22")                              \
  RING_A(RING_SLOT2, "6", "s41", "0x60008") RING_R(RING_SLOT2, C6_)                                                  \
  RING_A(RING_SLOT3, "6", "s41", "0x60008") RING_R(RING_SLOT3, C7_) RING_F("s41", "30")                              \
  RING_A(RING_SLOT0, "6", "s41", "0x60010") RING_A(RING_SLOT1, "4", "s41", "0x60010")                                \
  RING_A(RING_SLOT2, "2", "s41", "0x60018") RING_A(RING_SLOT3, "0", "s41", "0x60018")
#define RING_LOAD0_A "global_load_dwordx4 v[48:51], v15, s[36:37]\n global_load_dword v12, v7, s[38:39]\n"
#define RING_LOAD1_A "global_load_dwordx4 v[52:55], v15, s[36:37] offset:128\n"
#define RING_LOAD0_B "global_load_dwordx4 v[56:59], v15, s[36:37]\n global_load_dword v13, v7, s[38:39]\n"
#define RING_LOAD1_B "global_load_dwordx4 v[60:63], v15, s[36:37] offset:128\n"
#define RING_ZERO8(B_) "v_mov_b32 v" #B_ ", 0\n"
#define RING_CLOB_V "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",   \
  "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",      \
  "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",      \
  "v62", "v63"

// round 6: NWALK walker waves + (16 - NWALK) LOADER waves.  A loader wave is ISSUE-bound -- ~100 clocks per 1-KB LDS-DMA instruction
// inside a busy phase, 52 of them per 416-row slot = ~2 us per slot -- which the whole-matrix products hide (a phase of theirs walks
// 3-4 us) but the products over a row SUBSET (the loss rows, their transpose) do not: their walkers have 10-20 steps per phase and the
// timing variant without any load ran 187 -> 103 us (T-Finance loss rows) and 65.5 -> 27.8 us (Amazon): profiles/r06_ring_noload.log.
// Those products run with three loader waves (pieces dealt round-robin) and 13 walkers.
template <int NWALK>
__global__ void __launch_bounds__(1024) k_spmm_ring(const int32_t *__restrict__ wg_tab, const int32_t *__restrict__ wave_sb,
                                                    const uint16_t *__restrict__ idx, const uint32_t *__restrict__ ctl,
                                                    const int32_t *__restrict__ row_tab, int n_phases,
                                                    const float4 *__restrict__ XS, int n_src, int W,
                                                    const float *__restrict__ row_scale, const float *__restrict__ diag,
                                                    const float *__restrict__ X, int64_t ldx, const float *__restrict__ bias,
                                                    const float *__restrict__ prelu_a, float *__restrict__ out, int64_t ldo,
                                                    float *__restrict__ out_pre) {
  extern __shared__ __attribute__((aligned(16))) float4 panel[];
  const int slice = wg_tab[2 * blockIdx.x], block = wg_tab[2 * blockIdx.x + 1];
  if (slice < 0) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float4 *__restrict__ xs = XS + (int64_t)slice * n_src * SPMM_SL;
  if (wave >= NWALK) {
    // ---- loaders: slot s = source rows [s * RS, (s + 1) * RS) -> buffer s % S, 1 KB (8 rows) per instruction, lanes past the operand off;
    // loader li of NL takes the pieces li, li + NL, ...
    constexpr int NL = 16 - NWALK;
    const int li = wave - NWALK;
    auto issue = [&](int slot) {
      const int r0 = slot * RING_RS;
      const int nf4 = (n_src - r0 < RING_RS ? n_src - r0 : RING_RS) * SPMM_SL;
      const float4 *__restrict__ src = xs + (int64_t)r0 * SPMM_SL;
      float4 *dst = panel + (slot % RING_S) * RING_RS * SPMM_SL;
#pragma unroll
      for (int p = li; p < RING_RS * SPMM_SL / 64; p += NL) {
        const int i = p * 64 + lane;
        if (i < nf4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i),
                                           (__attribute__((address_space(3))) void *)(dst + p * 64), 16, 0, 0);
      }
    };
#ifdef RING_NO_LOAD
    __builtin_amdgcn_s_barrier();
    for (int j = 0; j < n_phases; ++j) RING_LOADER_BARRIER();
    return;
#endif
    for (int s = 0; s < RING_S - 1 && s < n_phases; ++s) issue(s);
    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0) (lgkmcnt / expcnt untouched)
    __builtin_amdgcn_s_barrier();                    // B0: the first window is resident
    for (int j = 0; j < n_phases; ++j) {
      if (j + RING_S - 1 < n_phases) issue(j + RING_S - 1);        // into the buffer phase j - 1 released
      __builtin_amdgcn_s_waitcnt(0x0f70);
      RING_LOADER_BARRIER();                         // end of phase j: slot j is free, slot j + V has landed
    }
    return;
  }
  const int g = lane >> 3, j = lane & 7;
  if (tid < 2 * SPMM_SL) panel[RING_S * RING_RS * SPMM_SL + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int gw = block * NWALK + wave;
  const int sb0 = __builtin_amdgcn_readfirstlane(wave_sb[2 * gw]), nsb = __builtin_amdgcn_readfirstlane(wave_sb[2 * gw + 1]);
  const uint16_t *ip = idx + (int64_t)sb0 * 128;
  const uint32_t *cp = ctl + sb0;
  const uint32_t lane_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)panel + j * 16, grp_off = g * 16;
  ring_v32f a0, a1;
  asm volatile(
      "s_mov_b64 s[36:37], %[ip]\n s_mov_b64 s[38:39], %[cp]\n s_mov_b32 s40, %[nsb]\n s_movk_i32 s43, 0x80\n"
      "v_mov_b32 v14, %[lane]\n v_mov_b32 v15, %[grp]\n v_mov_b32 v7, 0\n"
      RING_LOAD0_A RING_LOAD1_A                      // super-block 0 -> v[48:55] / v12, super-block 1 -> v[56:63] / v13
      RING_ADV
      RING_LOAD0_B RING_LOAD1_B
      RING_ZERO8(64) RING_ZERO8(65) RING_ZERO8(66) RING_ZERO8(67) RING_ZERO8(68) RING_ZERO8(69) RING_ZERO8(70) RING_ZERO8(71)
      RING_ZERO8(72) RING_ZERO8(73) RING_ZERO8(74) RING_ZERO8(75) RING_ZERO8(76) RING_ZERO8(77) RING_ZERO8(78) RING_ZERO8(79)
      RING_ZERO8(80) RING_ZERO8(81) RING_ZERO8(82) RING_ZERO8(83) RING_ZERO8(84) RING_ZERO8(85) RING_ZERO8(86) RING_ZERO8(87)
      RING_ZERO8(88) RING_ZERO8(89) RING_ZERO8(90) RING_ZERO8(91) RING_ZERO8(92) RING_ZERO8(93) RING_ZERO8(94) RING_ZERO8(95)
      RING_ZERO8(96) RING_ZERO8(97) RING_ZERO8(98) RING_ZERO8(99) RING_ZERO8(100) RING_ZERO8(101) RING_ZERO8(102) RING_ZERO8(103)
      RING_ZERO8(104) RING_ZERO8(105) RING_ZERO8(106) RING_ZERO8(107) RING_ZERO8(108) RING_ZERO8(109) RING_ZERO8(110) RING_ZERO8(111)
      RING_ZERO8(112) RING_ZERO8(113) RING_ZERO8(114) RING_ZERO8(115) RING_ZERO8(116) RING_ZERO8(117) RING_ZERO8(118) RING_ZERO8(119)
      RING_ZERO8(120) RING_ZERO8(121) RING_ZERO8(122) RING_ZERO8(123) RING_ZERO8(124) RING_ZERO8(125) RING_ZERO8(126) RING_ZERO8(127)
      "s_waitcnt vmcnt(4)\n v_readfirstlane_b32 s41, v12\n"
      "s_waitcnt lgkmcnt(0)\n s_barrier\n"           // B0 (the zero rows were written before this statement): the first window is resident
      RING_R(RING_SLOT0, "v48") RING_R(RING_SLOT1, "v49") RING_F("s41", "6")           // quads 0, 1 of the list: the pipeline fills
      RING_R(RING_SLOT2, "v50") RING_R(RING_SLOT3, "v51") RING_F("s41", "14")
      "10:\n s_cmp_eq_u32 s40, 1\n s_cbranch_scc1 12f\n"
      RING_BODY("v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v13", RING_LOAD0_A, RING_LOAD1_A)
      "s_sub_u32 s40, s40, 1\n s_cmp_eq_u32 s40, 1\n s_cbranch_scc1 13f\n"
      RING_BODY("v60", "v61", "v62", "v63", "v48", "v49", "v50", "v51", "v12", RING_LOAD0_B, RING_LOAD1_B)
      "s_sub_u32 s40, s40, 1\n s_branch 10b\n"
      "12:\n" RING_TAIL("v52", "v53", "v54", "v55") "s_branch 14f\n"
      "13:\n" RING_TAIL("v60", "v61", "v62", "v63")
      "14:\n s_waitcnt vmcnt(0)\n"                   // no stream load may land after the statement
      : "={v[64:95]}"(a0), "={v[96:127]}"(a1)
      : [ip] "s"(ip), [cp] "s"(cp), [nsb] "s"(nsb), [lane] "v"(lane_off), [grp] "v"(grp_off)
      : RING_CLOB_V, "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "m0", "scc", "memory");
  const int vi = slice * SPMM_SL + j;
  if (vi >= (W >> 2)) return;
  const float a = prelu_a ? *prelu_a : 1.0f;
  const int32_t *__restrict__ rt = row_tab + (int64_t)gw * RING_KR * 8 + g;
#pragma unroll
  for (int k = 0; k < RING_KR; ++k) {
    const int rowv = rt[k * 8];
    if (rowv < 0) continue;
    float4 z = k < 8 ? make_float4(a0[4 * k], a0[4 * k + 1], a0[4 * k + 2], a0[4 * k + 3])
                     : make_float4(a1[4 * (k & 7)], a1[4 * (k & 7) + 1], a1[4 * (k & 7) + 2], a1[4 * (k & 7) + 3]);
    const int row = rowv & ~GGAD_SPMM_PANEL_WIDE;
    if (rowv & GGAD_SPMM_PANEL_WIDE) {               // wide round: ONE row over the 8 lane groups (same flag in all 8 slots)
#pragma unroll
      for (int off = 8; off < GGAD_WAVE; off <<= 1) {                           // fixed butterfly over lane bits 3..5
        z.x += __shfl_xor(z.x, off, GGAD_WAVE); z.y += __shfl_xor(z.y, off, GGAD_WAVE);
        z.z += __shfl_xor(z.z, off, GGAD_WAVE); z.w += __shfl_xor(z.w, off, GGAD_WAVE);
      }
      if (g != 0) continue;
    }
    if (row_scale) { const float r = row_scale[row]; z.x *= r; z.y *= r; z.z *= r; z.w *= r; }
    if (diag) {
      const float dv = diag[row];
      const float4 x = reinterpret_cast<const float4 *>(X + (int64_t)row * ldx)[vi];
      z.x = fmaf(dv, x.x, z.x); z.y = fmaf(dv, x.y, z.y); z.z = fmaf(dv, x.z, z.z); z.w = fmaf(dv, x.w, z.w);
    }
    spmm_epilogue_store(z, vi, bias, prelu_a, a, out + (int64_t)row * ldo, out_pre ? out_pre + (int64_t)row * ldo : nullptr);
  }
}

// rows split into several segments: out[row] = epilogue(sum of part[first .. first + count))   (fixed order)
__global__ void __launch_bounds__(256) k_spmm_combine(const int32_t *__restrict__ multi_row, const int32_t *__restrict__ multi_first,
                                                      const int32_t *__restrict__ multi_count, int n_multi,
                                                      const float *__restrict__ part, int W, const float *__restrict__ bias,
                                                      const float *__restrict__ prelu_a, float *__restrict__ out, int64_t ldo,
                                                      float *__restrict__ out_pre) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= n_multi) return;
  const int lane = lane_id();
  const int orow = multi_row[m], first = multi_first[m], count = multi_count[m];
  const int nvec = W >> 2;
  const float a = prelu_a ? *prelu_a : 1.0f;
#pragma unroll
  for (int ch = 0; ch < SPMM_MAXCH; ++ch) {
    const int vi = ch * 64 + lane;
    if (vi < nvec) {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = 0;
      for (; k + 4 <= count; k += 4) {               // four partials in flight, added in slot order
        const float4 *src = reinterpret_cast<const float4 *>(part + (int64_t)(first + k) * W) + vi;
        const float4 p0 = src[0], p1 = src[(W >> 2)], p2 = src[2 * (W >> 2)], p3 = src[3 * (W >> 2)];
        z.x += p0.x; z.y += p0.y; z.z += p0.z; z.w += p0.w;
        z.x += p1.x; z.y += p1.y; z.z += p1.z; z.w += p1.w;
        z.x += p2.x; z.y += p2.y; z.z += p2.z; z.w += p2.w;
        z.x += p3.x; z.y += p3.y; z.z += p3.z; z.w += p3.w;
      }
      for (; k < count; ++k) {
        const float4 p = reinterpret_cast<const float4 *>(part + (int64_t)(first + k) * W)[vi];
        z.x += p.x; z.y += p.y; z.z += p.z; z.w += p.w;
      }
      spmm_epilogue_store(z, vi, bias, prelu_a, a, out + (int64_t)orow * ldo, out_pre ? out_pre + (int64_t)orow * ldo : nullptr);
    }
  }
}

// dZ = g * (z > 0 ? 1 : a); column partials of dZ (bias grad) and of g * z * [z <= 0] (slope grad).
// grid = (ceil(W/64), S); block = 4 waves; wave w of split s takes rows s*4 + w, + 4*S, ...
__global__ void __launch_bounds__(256) k_prelu_bwd(const float *__restrict__ g, const float *__restrict__ z,
                                                   const float *__restrict__ prelu_a, int M, int W, float *__restrict__ dz,
                                                   float *__restrict__ part_db, float *__restrict__ part_da) {
  __shared__ float sb[4][64], sa[4][64];
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int S = gridDim.y;
  const float a = *prelu_a;
  float adb = 0.f, ada = 0.f;
  if (c < W) {
    for (int r = blockIdx.y * 4 + wid; r < M; r += 4 * S) {
      const int64_t o = (int64_t)r * W + c;
      const float gv = g[o], zv = z[o];
      const float d = zv > 0.f ? gv : a * gv;
      dz[o] = d;
      adb += d;
      ada += zv > 0.f ? 0.f : gv * zv;
    }
  }
  sb[wid][lane] = adb; sa[wid][lane] = ada;
  __syncthreads();
  if (wid == 0 && c < W) {
    part_db[(int64_t)blockIdx.y * W + c] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
    part_da[(int64_t)blockIdx.y * W + c] = (sa[0][lane] + sa[1][lane]) + (sa[2][lane] + sa[3][lane]);
  }
}

// db[c] = sum_s part_db[s][c];  da = sum_c sum_s part_da[s][c]   (single block, fixed order)
// The same pass for W % 4 == 0 (every layer width here): float4 accesses, 256 / (W / 4) rows per workgroup pass, four row
// steps (8 loads) in flight per thread, one workgroup per split.  The scalar version above moved 1.7 TB/s (84 us on
// 39,357 x 300: one 256-byte row piece per wave and trip, 5 waves per CU).
__global__ void __launch_bounds__(256) k_prelu_bwd_v4(const float4 *__restrict__ g, const float4 *__restrict__ z,
                                                      const float *__restrict__ prelu_a, int M, int nvec, float4 *__restrict__ dz,
                                                      int64_t ldz, float4 *__restrict__ part_db, float4 *__restrict__ part_da) {
  // ldz: row stride of dz in float4 (nvec when dense; more when the next product wants 128-byte aligned rows)
  __shared__ float4 sb[256], sa[256];
  const int t = threadIdx.x;
  const int RP = 256 / nvec;                        // rows per pass of the workgroup (nvec <= 256)
  const int ro = t / nvec, cv = t - ro * nvec;
  const float a = *prelu_a;
  float4 adb = make_float4(0.f, 0.f, 0.f, 0.f), ada = make_float4(0.f, 0.f, 0.f, 0.f);
#define PRELU_ONE(G_, Z_, O_)                                                                                     \
  {                                                                                                               \
    float4 d;                                                                                                     \
    d.x = Z_.x > 0.f ? G_.x : a * G_.x; d.y = Z_.y > 0.f ? G_.y : a * G_.y;                                       \
    d.z = Z_.z > 0.f ? G_.z : a * G_.z; d.w = Z_.w > 0.f ? G_.w : a * G_.w;                                       \
    dz[(O_) / nvec * ldz + (O_) % nvec] = d;                                                                     \
    adb.x += d.x; adb.y += d.y; adb.z += d.z; adb.w += d.w;                                                       \
    ada.x += Z_.x > 0.f ? 0.f : G_.x * Z_.x; ada.y += Z_.y > 0.f ? 0.f : G_.y * Z_.y;                             \
    ada.z += Z_.z > 0.f ? 0.f : G_.z * Z_.z; ada.w += Z_.w > 0.f ? 0.f : G_.w * Z_.w;                             \
  }
  if (ro < RP) {
    const int64_t stride = (int64_t)gridDim.x * RP;
    int64_t r = (int64_t)blockIdx.x * RP + ro;
    for (; r + 3 * stride < M; r += 4 * stride) {
      const int64_t o0 = r * nvec + cv, o1 = o0 + stride * nvec, o2 = o1 + stride * nvec, o3 = o2 + stride * nvec;
      const float4 g0 = g[o0], g1 = g[o1], g2 = g[o2], g3 = g[o3];
      const float4 z0 = z[o0], z1 = z[o1], z2 = z[o2], z3 = z[o3];
      PRELU_ONE(g0, z0, o0) PRELU_ONE(g1, z1, o1) PRELU_ONE(g2, z2, o2) PRELU_ONE(g3, z3, o3)
    }
    for (; r < M; r += stride) {
      const int64_t o0 = r * nvec + cv;
      const float4 g0 = g[o0], z0 = z[o0];
      PRELU_ONE(g0, z0, o0)
    }
  }
#undef PRELU_ONE
  sb[t] = adb; sa[t] = ada;
  __syncthreads();
  if (t < nvec) {
    float4 b = sb[t], q = sa[t];
    for (int k = 1; k < RP; ++k) {
      const float4 b2 = sb[k * nvec + t], q2 = sa[k * nvec + t];
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w;
    }
    part_db[(int64_t)blockIdx.x * nvec + t] = b;
    part_da[(int64_t)blockIdx.x * nvec + t] = q;
  }
}

// any width: every thread walks columns c, c + 1024, ... and the S partials of each in order; the slope gradient = the sum of the
// column sums, added thread by thread through LDS in a fixed tree.  (The single-round kernel below needs W <= 1,024.)
__global__ void __launch_bounds__(1024) k_prelu_bwd_final_wide(const float *__restrict__ part_db, const float *__restrict__ part_da,
                                                               int S, int W, float *__restrict__ db, float *__restrict__ da) {
  __shared__ float qa[1024];
  float tot_a = 0.f;
  for (int c = threadIdx.x; c < W; c += 1024) {
    float b = 0.f, a = 0.f;
    for (int s0 = 0; s0 < S; ++s0) { b += part_db[(int64_t)s0 * W + c]; a += part_da[(int64_t)s0 * W + c]; }
    if (db) db[c] = b;
    tot_a += a;
  }
  qa[threadIdx.x] = tot_a;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) qa[threadIdx.x] += qa[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && da) *da = qa[0];
}

// Hand-offs between workgroups of ONE launch without an agent-scope release FENCE: that fence writes back every dirty line of the
// XCD's L2 -- in a kernel that is itself streaming out N x H floats (dz) each of its 320 fences flushed that stream: 50 us instead of
// 20.  The partial sums are instead STORED with agent-scope relaxed atomics (sc1: written through to memory), the wave drains its
// store counter (s_waitcnt vmcnt(0)) before the workgroup barrier, one thread draws the ticket (agent-scope RMW), and the reader uses
// agent-scope relaxed atomic LOADS (sc1: not served from a stale L2 line of an earlier launch).  Same hardware reasoning as the
// tagged-slot barrier of step_xcd.hip and the granules of exchange.cpp; tests/test_fullgraph_gpu.py re-runs the kernel on the same
// workspace with different data and checks every sum.
__device__ __forceinline__ void st_agent16(float4 *p, float4 v) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  __hip_atomic_store(q, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_agent16(const float4 *p) {
  const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}
__device__ __forceinline__ float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the fixed-order reduction of the S partial rows by ONE workgroup of 1,024 threads (W <= 1024): shared by the trailing launch
// k_prelu_bwd_final and by the last workgroup of k_prelu_bwd_one
template <bool AGENT = false>
__device__ __forceinline__ void prelu_final_body(const float *__restrict__ part_db, const float *__restrict__ part_da, int S, int W,
                                                 float *__restrict__ db, float *__restrict__ da, float *qb, float *qa) {
  auto ldp = [](const float *p) -> float { if constexpr (AGENT) return ld_agent(p); else return *p; };
  // one workgroup of 1,024 threads = G groups of (W rounded up to 64) threads, W <= 1024: group g sums its share of the S
  // partials of every column, eight partials of each array in flight (one dependent load per partial made a single-block
  // version 21 us at S = 43); fixed order: four interleaved running sums per group, combined pairwise, groups added in order
  const int Wp = (W + 63) & ~63, G = 1024 / Wp;
  const int g = threadIdx.x / Wp, c = threadIdx.x - g * Wp;
  float sum_b = 0.f, sum_a = 0.f;
  if (g < G && c < W) {
    const int s_lo = (int)((int64_t)S * g / G), s_hi = (int)((int64_t)S * (g + 1) / G);
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, a4[4] = {0.f, 0.f, 0.f, 0.f};
    int s0 = s_lo;
    for (; s0 + 16 <= s_hi; s0 += 16) {                    // (16 partials of each array in flight: S = 172 at Reddit size is 57 per group --
      float vb[16], va[16];                                //  four round trips instead of seven; the same additions in the same order)
#pragma unroll
      for (int k = 0; k < 16; ++k) { vb[k] = ldp(part_db + (int64_t)(s0 + k) * W + c); va[k] = ldp(part_da + (int64_t)(s0 + k) * W + c); }
#pragma unroll
      for (int k = 0; k < 16; ++k) { b4[k & 3] += vb[k]; a4[k & 3] += va[k]; }
    }
    for (; s0 + 8 <= s_hi; s0 += 8) {
      float vb[8], va[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { vb[k] = ldp(part_db + (int64_t)(s0 + k) * W + c); va[k] = ldp(part_da + (int64_t)(s0 + k) * W + c); }
#pragma unroll
      for (int k = 0; k < 8; ++k) { b4[k & 3] += vb[k]; a4[k & 3] += va[k]; }
    }
    for (; s0 < s_hi; ++s0) { b4[(s0 - s_lo) & 3] += ldp(part_db + (int64_t)s0 * W + c); a4[(s0 - s_lo) & 3] += ldp(part_da + (int64_t)s0 * W + c); }
    sum_b = (b4[0] + b4[1]) + (b4[2] + b4[3]);
    sum_a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  qb[threadIdx.x] = sum_b; qa[threadIdx.x] = sum_a;
  __syncthreads();
  float col_a = 0.f;
  if (g == 0 && c < W) {
    float b = qb[c];
    col_a = qa[c];
    for (int k = 1; k < G; ++k) { b += qb[k * Wp + c]; col_a += qa[k * Wp + c]; }
    if (db) db[c] = b;
  }
  __syncthreads();
  qa[threadIdx.x] = col_a;
  __syncthreads();
  for (int off = 512; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) qa[threadIdx.x] += qa[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0 && da) *da = qa[0];
}

__global__ void __launch_bounds__(1024) k_prelu_bwd_final(const float *__restrict__ part_db, const float *__restrict__ part_da,
                                                          int S, int W, float *__restrict__ db, float *__restrict__ da) {
  __shared__ float qb[1024], qa[1024];
  prelu_final_body(part_db, part_da, S, W, db, da, qb, qa);
}

// round 6: k_prelu_bwd_v4 and k_prelu_bwd_final in ONE launch (the trailing single-workgroup launch was 8.6 us of
// a 20-us pair, twice per epoch).  Workgroups of 1,024 threads (13 rows of W = 300 per pass instead of 3: four times the loads in
// flight per compute unit); every workgroup writes its partial column sums as before and draws a ticket (one agent-scope release per
// workgroup); the LAST one runs the same fixed-order reduction the trailing launch ran -- the result does not depend on which
// workgroup that is.  The ticket word returns to zero.
constexpr int PB_GRP = 4;          // workgroups per first-level group
constexpr int PB_MAXS = 320;       // workgroups of a launch at most (80 groups: the last group's reduction reads 27 rows per thread group)
__global__ void __launch_bounds__(1024) k_prelu_bwd_one(const float4 *__restrict__ g, const float4 *__restrict__ z,
                                                        const float *__restrict__ prelu_a, int M, int nvec, float4 *__restrict__ dz,
                                                        int64_t ldz, float4 *__restrict__ part_db, float4 *__restrict__ part_da,
                                                        float4 *__restrict__ gpart_db, float4 *__restrict__ gpart_da,
                                                        float *__restrict__ db, float *__restrict__ da, int32_t *__restrict__ ticket) {
  __shared__ float4 sb[1024], sa[1024];
  __shared__ int last;
  const int t = threadIdx.x;
  const int RP = 1024 / nvec;                       // rows per pass of the workgroup (nvec <= 256)
  const int ro = t / nvec, cv = t - ro * nvec;
  const float a = *prelu_a;
  float4 adb = make_float4(0.f, 0.f, 0.f, 0.f), ada = make_float4(0.f, 0.f, 0.f, 0.f);
#define PRELU_ONE(G_, Z_, O_)                                                                                     \
  {                                                                                                               \
    float4 d;                                                                                                     \
    d.x = Z_.x > 0.f ? G_.x : a * G_.x; d.y = Z_.y > 0.f ? G_.y : a * G_.y;                                       \
    d.z = Z_.z > 0.f ? G_.z : a * G_.z; d.w = Z_.w > 0.f ? G_.w : a * G_.w;                                       \
    dz[(O_) / nvec * ldz + (O_) % nvec] = d;                                                                     \
    adb.x += d.x; adb.y += d.y; adb.z += d.z; adb.w += d.w;                                                       \
    ada.x += Z_.x > 0.f ? 0.f : G_.x * Z_.x; ada.y += Z_.y > 0.f ? 0.f : G_.y * Z_.y;                             \
    ada.z += Z_.z > 0.f ? 0.f : G_.z * Z_.z; ada.w += Z_.w > 0.f ? 0.f : G_.w * Z_.w;                             \
  }
  if (ro < RP) {
    const int64_t stride = (int64_t)gridDim.x * RP;
    int64_t r = (int64_t)blockIdx.x * RP + ro;
    for (; r + 3 * stride < M; r += 4 * stride) {
      const int64_t o0 = r * nvec + cv, o1 = o0 + stride * nvec, o2 = o1 + stride * nvec, o3 = o2 + stride * nvec;
      const float4 g0 = g[o0], g1 = g[o1], g2 = g[o2], g3 = g[o3];
      const float4 z0 = z[o0], z1 = z[o1], z2 = z[o2], z3 = z[o3];
      PRELU_ONE(g0, z0, o0) PRELU_ONE(g1, z1, o1) PRELU_ONE(g2, z2, o2) PRELU_ONE(g3, z3, o3)
    }
    for (; r < M; r += stride) {
      const int64_t o0 = r * nvec + cv;
      const float4 g0 = g[o0], z0 = z[o0];
      PRELU_ONE(g0, z0, o0)
    }
  }
#undef PRELU_ONE
  sb[t] = adb; sa[t] = ada;
  __syncthreads();
  if (t < nvec) {
    float4 b = sb[t], q = sa[t];
    for (int k = 1; k < RP; ++k) {
      const float4 b2 = sb[k * nvec + t], q2 = sa[k * nvec + t];
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      q.x += q2.x; q.y += q2.y; q.z += q2.z; q.w += q2.w;
    }
    st_agent16(part_db + (int64_t)blockIdx.x * nvec + t, b);
    st_agent16(part_da + (int64_t)blockIdx.x * nvec + t, q);
  }
  // two levels of tickets (the elementwise pass wants >= one workgroup per compute unit, the final reduction few partial rows):
  //   groups of PB_GRP consecutive workgroups -- the last of a group to finish adds the group's partial rows (block order) into ONE row
  //   of gpart;  the last GROUP to finish runs the fixed-order reduction over the group rows.  Every sum has a fixed order; which
  //   workgroup performs it does not matter.  ticket[0]: groups done, ticket[1 + g]: workgroups of group g done; all left at zero.
  const int gid = blockIdx.x / PB_GRP, n_groups = (gridDim.x + PB_GRP - 1) / PB_GRP;
  const int gsize = min(PB_GRP, (int)gridDim.x - gid * PB_GRP);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores are acknowledged ...
  __syncthreads();                                   // ... every wave's are: the ticket may be drawn
  if (t == 0) last = (__hip_atomic_fetch_add(ticket + 1 + gid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  if (t == 0) ticket[1 + gid] = 0;
  if (t < 2 * nvec) {                                // float4 column t of [db | da] of the group row
    const float4 *src = (t < nvec ? part_db : part_da) + (int64_t)gid * PB_GRP * nvec + (t < nvec ? t : t - nvec);
    float4 v[PB_GRP];
#pragma unroll
    for (int k = 0; k < PB_GRP; ++k) v[k] = ld_agent16(src + (int64_t)min(k, gsize - 1) * nvec);
    float4 acc = v[0];
#pragma unroll
    for (int k = 1; k < PB_GRP; ++k)
      if (k < gsize) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    st_agent16((t < nvec ? gpart_db : gpart_da) + (int64_t)gid * nvec + (t < nvec ? t : t - nvec), acc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  if (t == 0) *ticket = 0;
  prelu_final_body<true>(reinterpret_cast<const float *>(gpart_db), reinterpret_cast<const float *>(gpart_da), n_groups, nvec * 4, db, da,
                         reinterpret_cast<float *>(sb), reinterpret_cast<float *>(sa));
}

__global__ void __launch_bounds__(256) k_relu_bwd(const float *__restrict__ g, const float *__restrict__ y, int64_t n,
                                                  float *__restrict__ dz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dz[i] = y[i] > 0.f ? g[i] : 0.f;
}

// out = z > 0 ? z : a z  (PReLU forward on its own: the layer whose aggregate A_hat X is cached has no SpMM to fuse it into)
__global__ void __launch_bounds__(256) k_prelu_fwd(const float *__restrict__ z, const float *__restrict__ prelu_a, int64_t n,
                                                   float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = z[i]; out[i] = v > 0.f ? v : *prelu_a * v; }
}
__global__ void __launch_bounds__(256) k_prelu_fwd_v4(const float4 *__restrict__ z, const float *__restrict__ prelu_a, int64_t n4,
                                                      float4 *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float a = *prelu_a;
  float4 v = z[i];
  v.x = v.x > 0.f ? v.x : a * v.x; v.y = v.y > 0.f ? v.y : a * v.y; v.z = v.z > 0.f ? v.z : a * v.z; v.w = v.w > 0.f ? v.w : a * v.w;
  out[i] = v;
}

// inv[r] = 1/|x_r| (inf -> 0), xn = x * inv                                     run.py:177-180
__global__ void __launch_bounds__(256) k_rownorm(const float *__restrict__ X, int M, int W, float *__restrict__ inv,
                                                 float *__restrict__ Xn) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const int lane = lane_id();
  const float *x = X + (int64_t)r * W;
  float ss = 0.f;
  for (int c = lane; c < W; c += 64) ss = fmaf(x[c], x[c], ss);
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  float iv = 1.0f / nrm;
  if (isinf(iv)) iv = 0.0f;
  if (lane == 0) inv[r] = iv;
  for (int c = lane; c < W; c += 64) Xn[(int64_t)r * W + c] = x[c] * iv;
}

// dX_r = inv_r * (dXn_r - Xn_r <Xn_r, dXn_r>) ; rows with inv = 0 (zero vectors) get the masked value 0 * dXn
__global__ void __launch_bounds__(256) k_rownorm_bwd(const float *__restrict__ Xn, const float *__restrict__ inv,
                                                     const float *__restrict__ dXn, int M, int W, float *__restrict__ dX) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const int lane = lane_id();
  const float *xn = Xn + (int64_t)r * W, *d = dXn + (int64_t)r * W;
  float dot = 0.f;
  for (int c = lane; c < W; c += 64) dot = fmaf(xn[c], d[c], dot);
  dot = wave_sum(dot);
  const float iv = inv[r];
  for (int c = lane; c < W; c += 64) dX[(int64_t)r * W + c] = iv * (d[c] - xn[c] * dot);
}

// out[p] = scale[p] * <A[sel[p]], B[p]>
__global__ void __launch_bounds__(256) k_rowdot(const float *__restrict__ A, const int32_t *__restrict__ sel,
                                                const float *__restrict__ B, int n, int W, const float *__restrict__ scale,
                                                float *__restrict__ out) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int lane = lane_id();
  const float *a = A + (int64_t)(sel ? sel[p] : p) * W, *b = B + (int64_t)p * W;
  float dot = 0.f;
  for (int c = lane; c < W; c += 64) dot = fmaf(a[c], b[c], dot);
  dot = wave_sum(dot);
  if (lane == 0) out[p] = (scale ? scale[p] : 1.0f) * dot;
}

// out[e] = || X[row(e)] - X[col[e]] ||_2 for every stored entry e of a CSR matrix (TAM's calc_distance, utils_tam.py:190-199:
// the attribute distance of every edge, computed once per dataset).  One wave per entry; lanes stride over the W attributes.
__global__ void __launch_bounds__(256) k_edge_dist(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const float *__restrict__ X, int n_rows, int W, float *__restrict__ out) {
  const int row = blockIdx.x;
  if (row >= n_rows) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const int s = rowptr[row], e = rowptr[row + 1];
  const float *xi = X + (int64_t)row * W;
  for (int p = s + wid; p < e; p += 4) {
    const float *xj = X + (int64_t)col[p] * W;
    float acc = 0.f;
    for (int c = lane; c < W; c += 64) { const float d = xi[c] - xj[c]; acc = fmaf(d, d, acc); }
    acc = wave_sum(acc);
    if (lane == 0) out[p] = sqrtf(acc);
  }
}

// out[p][:] = coef[p] * X[sel[p]][:]   (gather + scale);  with add_to: out[sel[p]][:] += coef[p] * X[p][:]  (sel unique)
__global__ void __launch_bounds__(256) k_rows_scale(const float *__restrict__ X, const int32_t *__restrict__ sel,
                                                    const float *__restrict__ coef, int n, int W, int scatter_add,
                                                    float *__restrict__ out) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int lane = lane_id();
  const float cf = coef ? coef[p] : 1.0f;
  if (!scatter_add) {
    const float *x = X + (int64_t)sel[p] * W;
    for (int c = lane; c < W; c += 64) out[(int64_t)p * W + c] = cf * x[c];
  } else {
    const float *x = X + (int64_t)p * W;
    float *o = out + (int64_t)sel[p] * W;
    for (int c = lane; c < W; c += 64) o[c] = fmaf(cf, x[c], o[c]);
  }
}

// ---- the head of the training forward from `emb` on (model.py:140-182) as ONE autograd node (fullgraph.py GgadHeadFn): the
// index_select / add / cat / index_copy glue of the reference and the gradient accumulation autograd does for the five consumers
// of `emb` were ~35 full-tensor torch kernels per epoch.  Wave per row, W columns.
// out[p] = X[idx[p]] (+ add[p])                                                        emb[abn] + noise     model.py:141-145
__global__ void __launch_bounds__(256) k_head_gather(const float *__restrict__ X, const int32_t *__restrict__ idx,
                                                     const float *__restrict__ add, int n, int W, float *__restrict__ out) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int lane = lane_id();
  const float *x = X + (int64_t)idx[p] * W;
  for (int c = lane; c < W; c += 64) out[(int64_t)p * W + c] = x[c] + (add ? add[(int64_t)p * W + c] : 0.0f);
}
// out = [X[nrm]; con]                                                                   cat((emb[normal], emb_con))  :159
__global__ void __launch_bounds__(256) k_head_combine(const float *__restrict__ X, const int32_t *__restrict__ nrm, int n_nrm,
                                                      const float *__restrict__ con, int n_con, int W, float *__restrict__ out) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_nrm + n_con) return;
  const int lane = lane_id();
  const float *x = p < n_nrm ? X + (int64_t)nrm[p] * W : con + (int64_t)(p - n_nrm) * W;
  for (int c = lane; c < W; c += 64) out[(int64_t)p * W + c] = x[c];
}
// out[i] = abn_pos[i] >= 0 ? con[abn_pos[i]] : X[i]                                     emb[:, abn, :] = emb_con      :182
__global__ void __launch_bounds__(256) k_head_emb_out(const float *__restrict__ X, const int32_t *__restrict__ abn_pos,
                                                      const float *__restrict__ con, int n, int W, float *__restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = lane_id();
  const int q = abn_pos[i];
  const float *x = q >= 0 ? con + (int64_t)q * W : X + (int64_t)i * W;
  for (int c = lane; c < W; c += 64) out[(int64_t)i * W + c] = x[c];
}
// round 6: k_head_gather and the emb[normal] half of k_head_combine in ONE launch (both read rows of emb before anything else of the head
// runs; the emb_con half of emb_combine is written in place by the product that forms it):
//   p < n_abn:  out_abn[p] = X[abn[p]] + add[p]          otherwise:  out_comb[p - n_abn] = X[nrm[p - n_abn]]
__global__ void __launch_bounds__(256) k_head_rows(const float *__restrict__ X, const int32_t *__restrict__ abn,
                                                   const float *__restrict__ add, int n_abn, const int32_t *__restrict__ nrm, int n_nrm,
                                                   int W, float *__restrict__ out_abn, float *__restrict__ out_comb) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_abn + n_nrm) return;
  const int lane = lane_id();
  if (p < n_abn) {
    const float *x = X + (int64_t)abn[p] * W;
    for (int c = lane; c < W; c += 64) out_abn[(int64_t)p * W + c] = x[c] + (add ? add[(int64_t)p * W + c] : 0.0f);
  } else {
    const int q = p - n_abn;
    const float *x = X + (int64_t)nrm[q] * W;
    for (int c = lane; c < W; c += 64) out_comb[(int64_t)q * W + c] = x[c];
  }
}
// X[abn[p]] = con[p]  (abn duplicate-free)                                               emb[:, abn, :] = emb_con IN PLACE   model.py:182
__global__ void __launch_bounds__(256) k_head_emb_put(const float *__restrict__ con, const int32_t *__restrict__ abn, int n_abn, int W,
                                                      float *__restrict__ X) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_abn) return;
  const int lane = lane_id();
  float *x = X + (int64_t)abn[p] * W;
  for (int c = lane; c < W; c += 64) x[c] = con[(int64_t)p * W + c];
}
// dz[p] = [y[p] > 0] * (g_con[p] + g_out[abn[p]] + g_tail[p]): the three gradients that reach emb_con = relu(fc4(.)) (the loss, the
// rows written back into emb, the tail of emb_combine) and the relu in one pass; absent terms are null
__global__ void __launch_bounds__(256) k_head_con_grad(const float *__restrict__ g_con, const float *__restrict__ g_out,
                                                       const int32_t *__restrict__ abn, const float *__restrict__ g_tail,
                                                       const float *__restrict__ y, int n, int W, float *__restrict__ dz) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const int lane = lane_id();
  const int64_t o = (int64_t)p * W, r = g_out ? (int64_t)abn[p] * W : 0;
  for (int c = lane; c < W; c += 64) {
    float t = g_con ? g_con[o + c] : 0.0f;
    if (g_out) t += g_out[r + c];
    if (g_tail) t += g_tail[o + c];
    dz[o + c] = y[o + c] > 0.0f ? t : 0.0f;
  }
}
// d emb[i] = [i not in abn] g_out[i] + [i in normal] g_comb[nrm_pos[i]] + [i in abn] g_abn[abn_pos[i]] + sp[i]
// (index_copy, index_select x 2, the rows product A_hat[abn, :] emb: every consumer of emb in one pass; absent terms are null)
__global__ void __launch_bounds__(256) k_head_emb_grad(const float *__restrict__ g_out, const int32_t *__restrict__ abn_pos,
                                                       const int32_t *__restrict__ nrm_pos, const float *__restrict__ g_comb,
                                                       const float *__restrict__ g_abn, const float *__restrict__ sp, int n, int W,
                                                       float *__restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = lane_id();
  const int qa = abn_pos[i], qn = nrm_pos[i];
  const int64_t o = (int64_t)i * W;
  for (int c = lane; c < W; c += 64) {
    float t = (g_out && qa < 0) ? g_out[o + c] : 0.0f;
    if (g_comb && qn >= 0) t += g_comb[(int64_t)qn * W + c];
    if (g_abn && qa >= 0) t += g_abn[(int64_t)qa * W + c];
    if (sp) t += sp[o + c];
    out[o + c] = t;
  }
}

// backward of the loss block: every product with the incoming d total in one launch (autograd: five tiny torch kernels)
//   c = g_aff r_inv_J g     d logits = d_logits g     d emb_con = dD g     d emb_abnormal = -dD g
__global__ void __launch_bounds__(256) k_loss_bwd_scale(const float *__restrict__ g_total, const float *__restrict__ g_aff,
                                                        const float *__restrict__ r_inv_j, const float *__restrict__ d_logits,
                                                        const float *__restrict__ dD, int L, int64_t n_rec, float *__restrict__ c,
                                                        float *__restrict__ dl, float *__restrict__ d_con, float *__restrict__ d_abn) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float g = g_total[0];
  if (i < L) { c[i] = (g_aff[i] * r_inv_j[i]) * g; dl[i] = d_logits[i] * g; }
  if (i < n_rec) { const float v = dD[i] * g; d_con[i] = v; d_abn[i] = -v; }
}

__device__ __forceinline__ float block_sum_1024(float v, float *red) {
  v = wave_sum(v);
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < 16; ++k) t += red[k];
  return t;
}

// Loss block of run.py:165-210 on the small tensors (one workgroup):
//   logits[L], L = Nn + A, labels 0 (first Nn) / 1 (last A)           -> bce, d_logits
//   aff[L]: affinity at normal_idx (first Nn) and abnormal_idx (last A) -> margin(0.7), g_aff[L] = d total / d aff
//   D = emb_con - emb_abnormal (A x H): rec = mean_h sqrt(sum_a D^2)   (axis quirk, run.py:207-208) -> dD
__global__ void __launch_bounds__(1024) k_full_loss(const float *__restrict__ logits, const float *__restrict__ aff, int Nn,
                                                    int A, float margin_c, float *__restrict__ losses4,
                                                    float *__restrict__ d_logits, float *__restrict__ g_aff) {
  __shared__ float red[16];
  const int L = Nn + A;
  float s_bce = 0.f, s_n = 0.f, s_a = 0.f;
  for (int i = threadIdx.x; i < L; i += 1024) {
    const float x = logits[i];
    const float y = i < Nn ? 0.f : 1.f;
    s_bce += (1.0f - y) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));      // BCEWithLogits, pos_weight 1
    d_logits[i] = (1.0f / (1.0f + expf(-x)) - y) / (float)L;
    if (i < Nn) s_n += aff[i]; else s_a += aff[i];
  }
  const float bce = block_sum_1024(s_bce, red) / (float)L;
  const float an = block_sum_1024(s_n, red) / (float)Nn;
  const float ab = block_sum_1024(s_a, red) / (float)A;
  const float m = margin_c - (an - ab);
  const float active = m >= 0.f ? 1.f : 0.f;
  for (int i = threadIdx.x; i < L; i += 1024) g_aff[i] = active * (i < Nn ? -1.0f / (float)Nn : 1.0f / (float)A);
  if (threadIdx.x == 0) {
    const float margin = fmaxf(m, 0.f);
    losses4[0] = margin + bce; losses4[1] = margin; losses4[2] = bce; losses4[3] = 0.f;     // k_rec_apply adds the reconstruction term
  }
}

// Reconstruction term rec = mean_h sqrt(sum_a D[a][h]^2), D = emb_con - emb_abn (A x H; the axis quirk of run.py:207-208),
// and dD = d total / d D = D / (H * colnorm).  A single workgroup walking the A rows for every column took 0.44 ms at
// T-Finance size (A = 844), 3 % of the epoch; here the rows are cut into blocks of 32:
//   k_rec_part : partial column sums of squares per row block                       -> ws[block][H]
//   k_rec_apply: column norms from the partials (block order, identical in every workgroup), dD for the block's rows;
//                workgroup 0 also reduces rec and adds it to the loss record
constexpr int REC_ROWS = 8;     // (32 rows walked one dependent-looking load at a time made the two launches 17 + 13 us on 238 x 300)
__global__ void __launch_bounds__(256) k_rec_part(const float *__restrict__ emb_con, const float *__restrict__ emb_abn, int A, int H,
                                                  float *__restrict__ ws) {
  const int a0 = blockIdx.x * REC_ROWS, a1 = min(A, a0 + REC_ROWS);
  for (int h = threadIdx.x; h < H; h += 256) {
    float ss = 0.f;
#pragma unroll 8
    for (int a = a0; a < a1; ++a) { const float d = emb_con[(int64_t)a * H + h] - emb_abn[(int64_t)a * H + h]; ss = fmaf(d, d, ss); }
    ws[(int64_t)blockIdx.x * H + h] = ss;
  }
}

__global__ void __launch_bounds__(256) k_rec_apply(const float *__restrict__ emb_con, const float *__restrict__ emb_abn, int A, int H,
                                                   const float *__restrict__ ws, int n_blocks, float *__restrict__ dD,
                                                   float *__restrict__ losses4) {
  __shared__ float red[4];
  const int a0 = blockIdx.x * REC_ROWS, a1 = min(A, a0 + REC_ROWS);
  float s_rec = 0.f;
  for (int h = threadIdx.x; h < H; h += 256) {
    float ss = 0.f;
#pragma unroll 8
    for (int b = 0; b < n_blocks; ++b) ss += ws[(int64_t)b * H + h];          // fixed order
    const float nrm = sqrtf(ss);
    s_rec += nrm;
    const float k = 1.0f / ((float)H * nrm);
#pragma unroll 8
    for (int a = a0; a < a1; ++a) {
      const float d = emb_con[(int64_t)a * H + h] - emb_abn[(int64_t)a * H + h];
      dD[(int64_t)a * H + h] = d * k;
    }
  }
  if (blockIdx.x != 0) return;
  const float w = wave_sum(s_rec);
  if (lane_id() == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float rec = ((red[0] + red[1]) + (red[2] + red[3])) / (float)H;
    losses4[3] = rec;
    losses4[0] += rec;                                                         // total = margin + bce + rec  (run.py:210)
  }
}

// ---- round 6: the loss block in THREE launches forward (k_rownorm, the R^T product, k_loss_fwd_fused) and three backward
// (k_loss_bwd_fused, the R[:, J] product, k_rownorm_bwd_add) instead of six + six + a torch fill (a 3-10 us launch
// per row-local step was 67 us of a 483-us Reddit epoch).
//
// k_loss_fwd_fused: workgroups [0, nb_dot) form aff[p] = r_inv_J[p] <e_hat[J[p]], S[p]> for LD_ROWS rows each (what k_rowdot did),
// workgroups [nb_dot, nb_dot + nb_rec) the partial column sums of squares of D = emb_con - emb_abn per block of REC_ROWS rows (k_rec_part).
// Every workgroup then draws a ticket (one agent-scope atomic after an agent-scope release of its stores); the LAST one -- whichever it
// is: it only reads what the others published and sums it in a fixed order, so the result does not depend on who it was -- does what
// k_full_loss and the column-norm half of k_rec_apply did: BCE and d_logits, the margin and g_aff, the column norms -> rec and
// kcol[h] = 1 / (H |D[:, h]|), the four loss values.  dD itself (A x H) is NOT formed here by one workgroup: the backward launch forms
// d emb_con = g D kcol over all its workgroups.  The ticket counter is put back to zero by the last workgroup (graph replays).
constexpr int LD_ROWS = 16;      // rows of the affinity per workgroup: one per wave (L = 1.3-6.5 K rows -> 80-400 tickets of ~7 ns each)
constexpr int LF_T = 1024;       // threads per workgroup: the LAST workgroup's serial part is a chain of memory round trips per thread --
                                 // the first version (256 threads, one load per loop trip) took 28 us at Reddit size, as long as the four
                                 // launches it replaced; here every thread has all its loads in flight at once
__global__ void __launch_bounds__(LF_T) k_loss_fwd_fused(const float *__restrict__ en, const int32_t *__restrict__ J,
                                                         const float *__restrict__ S, const float *__restrict__ r_inv_j, int L, int W,
                                                         const float *__restrict__ logits, int Nn, int A,
                                                         const float *__restrict__ emb_con, const float *__restrict__ emb_abn,
                                                         float margin_c, int nb_dot, int nb_part, float *__restrict__ aff,
                                                         float *__restrict__ ws, float *__restrict__ kcol, float *__restrict__ losses4,
                                                         float *__restrict__ d_logits, float *__restrict__ g_aff,
                                                         int32_t *__restrict__ ticket) {
  __shared__ float red[16];
  __shared__ float qb[LF_T];
  __shared__ int last;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  if ((int)blockIdx.x < nb_dot) {
    const int p = blockIdx.x * LD_ROWS + wid;
    if (p < L) {
      const float *a = en + (int64_t)J[p] * W, *b = S + (int64_t)p * W;
      float dot = 0.f;
      for (int c = lane; c < W; c += 64) dot = fmaf(a[c], b[c], dot);
      dot = wave_sum(dot);
      if (lane == 0) aff[p] = r_inv_j[p] * dot;
    }
  } else {
    // four partial rows (REC_ROWS rows of D each) per workgroup: thread group tid >> 8 takes one, its 256 threads the columns
    const int part = ((int)blockIdx.x - nb_dot) * 4 + (threadIdx.x >> 8);
    if (part < nb_part) {
      const int a0 = part * REC_ROWS, a1 = min(A, a0 + REC_ROWS);
      for (int h = threadIdx.x & 255; h < W; h += 256) {
        float ss = 0.f;
#pragma unroll 8
        for (int a = a0; a < a1; ++a) { const float d = emb_con[(int64_t)a * W + h] - emb_abn[(int64_t)a * W + h]; ss = fmaf(d, d, ss); }
        ws[(int64_t)part * W + h] = ss;
      }
    }
  }
  // this workgroup's aff / ws stores reach memory before its ticket: the barrier orders every wave's stores before thread 0 (they have
  // reached the L2: workgroup-scope release), thread 0's agent-scope release writes the L2's dirty lines back -- ONE write-back per
  // workgroup (a release fence in every wave -- 2,300 L2 write-backs per launch -- made this kernel 28-32 us at Reddit size)
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // ... and everyone else's are read from memory, not from a stale line
  if (threadIdx.x == 0) *ticket = 0;
  auto block_sum = [&](float v) {                             // 16 waves, added in wave order
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    return t;
  };
  // BCE, d_logits, the two affinity means: elements tid, tid + 1024, ...; eight of them per trip, their 16 loads issued before the first use
  float s_bce = 0.f, s_n = 0.f, s_a = 0.f;
  for (int i0 = threadIdx.x; i0 < L; i0 += 8 * LF_T) {
    float xv[8], av[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * LF_T;
      xv[k] = i < L ? logits[i] : 0.f;
      av[k] = i < L ? aff[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = i0 + k * LF_T;
      if (i < L) {
        const float x = xv[k];
        const float y = i < Nn ? 0.f : 1.f;
        s_bce += (1.0f - y) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));      // BCEWithLogits, pos_weight 1
        d_logits[i] = (1.0f / (1.0f + expf(-x)) - y) / (float)L;
        if (i < Nn) s_n += av[k]; else s_a += av[k];
      }
    }
  }
  // column norms of D from the partial rows: G thread groups of Wp threads, group gi sums its share of the partial rows of column c
  // (eight loads in flight), the groups are added in order
  const int Wp = (W + 63) & ~63, G = LF_T / Wp;
  const int gi = threadIdx.x / Wp, c = threadIdx.x - gi * Wp;
  float part_sum = 0.f;
  if (gi < G && c < W) {
    const int b_lo = (int)((int64_t)nb_part * gi / G), b_hi = (int)((int64_t)nb_part * (gi + 1) / G);
    int b = b_lo;
    for (; b + 8 <= b_hi; b += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ws[(int64_t)(b + k) * W + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) part_sum += v[k];
    }
    for (; b < b_hi; ++b) part_sum += ws[(int64_t)b * W + c];
  }
  qb[threadIdx.x] = part_sum;
  const float bce = block_sum(s_bce) / (float)L;              // (its barriers also publish qb)
  const float an = block_sum(s_n) / (float)Nn;
  const float ab = block_sum(s_a) / (float)A;
  const float m = margin_c - (an - ab);
  const float active = m >= 0.f ? 1.f : 0.f;
  for (int i = threadIdx.x; i < L; i += LF_T) g_aff[i] = active * (i < Nn ? -1.0f / (float)Nn : 1.0f / (float)A);
  float s_rec = 0.f;
  if (gi == 0 && c < W) {
    float ss = qb[c];
    for (int k = 1; k < G; ++k) ss += qb[k * Wp + c];
    const float nrm = sqrtf(ss);
    s_rec = nrm;
    kcol[c] = 1.0f / ((float)W * nrm);
  }
  const float rec = block_sum(s_rec) / (float)W;
  if (threadIdx.x == 0) {
    const float margin = fmaxf(m, 0.f);
    losses4[0] = (margin + bce) + rec; losses4[1] = margin; losses4[2] = bce; losses4[3] = rec;     // total = margin + bce + rec (run.py:210)
  }
}

// backward of the loss block, the row-local half, in one launch: with g = d total,
//   workgroups [0, nb_l):  c[p] = g_aff[p] r_inv_J[p] g,  d logits[p] = d_logits[p] g,  xc[p][:] = c[p] e_hat[J[p]][:]    (4 rows per workgroup)
//   the others:            d emb_con[a][h] = ((con - abn)[a][h] kcol[h]) g,  d emb_abnormal = -d emb_con                (4 rows per workgroup)
__global__ void __launch_bounds__(256) k_loss_bwd_fused(const float *__restrict__ g_total, const float *__restrict__ g_aff,
                                                        const float *__restrict__ r_inv_j, const float *__restrict__ d_logits,
                                                        const float *__restrict__ en, const int32_t *__restrict__ J, int L, int W,
                                                        const float *__restrict__ emb_con, const float *__restrict__ emb_abn,
                                                        const float *__restrict__ kcol, int A, int nb_l, float *__restrict__ c,
                                                        float *__restrict__ dl, float *__restrict__ xc, float *__restrict__ d_con,
                                                        float *__restrict__ d_abn) {
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const float g = g_total[0];
  if ((int)blockIdx.x < nb_l) {
    const int p = blockIdx.x * 4 + wid;
    if (p >= L) return;
    const float cf = (g_aff[p] * r_inv_j[p]) * g;
    if (lane == 0) { c[p] = cf; dl[p] = d_logits[p] * g; }
    const float *x = en + (int64_t)J[p] * W;
    for (int col = lane; col < W; col += 64) xc[(int64_t)p * W + col] = cf * x[col];
  } else {
    const int a = (blockIdx.x - nb_l) * 4 + wid;
    if (a >= A) return;
    const int64_t o = (int64_t)a * W;
    for (int h = lane; h < W; h += 64) {
      const float v = ((emb_con[o + h] - emb_abn[o + h]) * kcol[h]) * g;
      d_con[o + h] = v;
      d_abn[o + h] = -v;
    }
  }
}

// k_rownorm_bwd with the scatter-add of c_j S_j onto the rows J folded in: pos_n[r] / pos_a[r] = position of row r in the normal /
// abnormal segment of J or -1 (each segment is duplicate-free; a node in both gets both terms, normal first -- the order of the two
// k_rows_scale launches this replaces):  dXn_r = den_r (+ c[q] S[q]) ...;  dX_r = inv_r (dXn_r - Xn_r <Xn_r, dXn_r>)
__global__ void __launch_bounds__(256) k_rownorm_bwd_add(const float *__restrict__ Xn, const float *__restrict__ inv,
                                                         const float *__restrict__ den, const int32_t *__restrict__ pos_n,
                                                         const int32_t *__restrict__ pos_a, const float *__restrict__ c,
                                                         const float *__restrict__ S, int M, int W, float *__restrict__ dX) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= M) return;
  const int lane = lane_id();
  const float *xn = Xn + (int64_t)r * W, *d = den + (int64_t)r * W;
  const int qn = pos_n[r], qa = pos_a[r];
  const float cn = qn >= 0 ? c[qn] : 0.f, ca = qa >= 0 ? c[qa] : 0.f;
  const float *sn = S + (int64_t)(qn >= 0 ? qn : 0) * W, *sa = S + (int64_t)(qa >= 0 ? qa : 0) * W;
  const float iv = inv[r];
  if (W <= 512) {                                              // the row in registers: every operand is read once
    float v[8], x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int col = lane + 64 * k;
      v[k] = col < W ? d[col] : 0.f;
      x[k] = col < W ? xn[col] : 0.f;
    }
    if (qn >= 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int col = lane + 64 * k; if (col < W) v[k] = fmaf(cn, sn[col], v[k]); }
    }
    if (qa >= 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int col = lane + 64 * k; if (col < W) v[k] = fmaf(ca, sa[col], v[k]); }
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot = fmaf(x[k], v[k], dot);   // (columns past W hold zeros; same order as the loop below)
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int col = lane + 64 * k; if (col < W) dX[(int64_t)r * W + col] = iv * (v[k] - x[k] * dot); }
    return;
  }
  float dot = 0.f;
  for (int col = lane; col < W; col += 64) {
    float v = d[col];
    if (qn >= 0) v = fmaf(cn, sn[col], v);
    if (qa >= 0) v = fmaf(ca, sa[col], v);
    dot = fmaf(xn[col], v, dot);
  }
  dot = wave_sum(dot);
  for (int col = lane; col < W; col += 64) {
    float v = d[col];
    if (qn >= 0) v = fmaf(cn, sn[col], v);
    if (qa >= 0) v = fmaf(ca, sa[col], v);
    dX[(int64_t)r * W + col] = iv * (v - xn[col] * dot);
  }
}

// torch.optim.Adam.step on a flat fp32 block (betas .9/.999, eps 1e-8, L2 weight decay), step index on device
__global__ void __launch_bounds__(256) k_adam_flat(float *__restrict__ p, float *__restrict__ m, float *__restrict__ v,
                                                   const float *__restrict__ g, int64_t n, float lr, float wd,
                                                   int32_t *__restrict__ step_counter, int bump) {
  __shared__ float sc[2];
  if (threadIdx.x == 0) {
    const double t = (double)(*step_counter + (bump ? 1 : 0));
    sc[0] = (float)((double)lr / (1.0 - pow(0.9, t)));
    sc[1] = (float)sqrt(1.0 - pow(0.999, t));
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float pi = p[i];
    float gi = fmaf(wd, pi, g[i]);
    float mi = m[i], vi = v[i];
    mi = fmaf(gi - mi, 0.1f, mi);
    vi = fmaf(0.001f * gi, gi, vi * 0.999f);
    const float denom = sqrtf(vi) / sc[1] + 1e-8f;
    pi = pi - sc[0] * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

__global__ void k_bump(int32_t *c) { *c += 1; }

// The same update for up to ADAM_MAX_T tensors in ONE launch (the pointers travel as kernel arguments): a model step was one
// k_adam_flat + one k_bump per parameter -- 20 launches of a few microseconds in an epoch of ~125.
constexpr int ADAM_MAX_T = 16;
struct AdamMulti {
  float *p[ADAM_MAX_T], *m[ADAM_MAX_T], *v[ADAM_MAX_T];
  const float *g[ADAM_MAX_T];
  int32_t *ctr[ADAM_MAX_T];
  int64_t n[ADAM_MAX_T];
  int blk0[ADAM_MAX_T + 1];          // first workgroup of every tensor
  int n_t;
};
// round 6: ADAM_EPB elements per workgroup (four per thread, 256 apart: coalesced) and -- with `tickets` -- no trailing k_bump_multi
// launch: every workgroup of tensor t draws a ticket on tickets[t] when it is done; the last one (all others have read the counter
// by then: they read it first) advances the tensor's step counter and puts the ticket back to zero.  <= 88 tickets per address at
// H = 300 (a ticket per 256 elements on ONE address was tried in round 5: 1,100 x ~7 ns, slower than the 4-us launch it replaced).
constexpr int ADAM_EPB = 1024;
__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti A, float lr, float wd, int32_t *__restrict__ tickets) {
  __shared__ float sc[2];
  int t = 0;
  while (t + 1 < A.n_t && (int)blockIdx.x >= A.blk0[t + 1]) ++t;          // workgroup-uniform
  float *__restrict__ p = A.p[t], *__restrict__ m = A.m[t], *__restrict__ v = A.v[t];
  const float *__restrict__ gr = A.g[t];
  const int64_t i0 = (int64_t)((int)blockIdx.x - A.blk0[t]) * ADAM_EPB + threadIdx.x, nt = A.n[t];
  constexpr int NK = ADAM_EPB / 256;
  float pv[NK], gv[NK], mv[NK], vv[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {                               // all 16 loads of the thread first (one dependent round trip, not four),
    const int64_t i = i0 + k * 256;                            // in flight while thread 0 forms the two bias corrections in double
    const bool ok = i < nt;
    pv[k] = ok ? p[i] : 0.f; gv[k] = ok ? gr[i] : 0.f; mv[k] = ok ? m[i] : 0.f; vv[k] = ok ? v[i] : 0.f;
  }
  if (threadIdx.x == 0) sc[0] = (float)((double)lr / (1.0 - pow(0.9, (double)(*A.ctr[t] + 1))));       // (two waves: the two double pow()
  if (threadIdx.x == 64) sc[1] = (float)sqrt(1.0 - pow(0.999, (double)(*A.ctr[t] + 1)));               //  side by side, not one after the other)
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int64_t i = i0 + k * 256;
    if (i < nt) {
      float pi = pv[k];
      float gi = fmaf(wd, pi, gv[k]);
      float mi = mv[k], vi = vv[k];
      mi = fmaf(gi - mi, 0.1f, mi);
      vi = fmaf(0.001f * gi, gi, vi * 0.999f);
      const float denom = sqrtf(vi) / sc[1] + 1e-8f;
      pi = pi - sc[0] * (mi / denom);
      p[i] = pi; m[i] = mi; v[i] = vi;
    }
  }
  if (tickets) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nb = A.blk0[t + 1] - A.blk0[t];
      if (__hip_atomic_fetch_add(tickets + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
        tickets[t] = 0;
        *A.ctr[t] += 1;
      }
    }
  }
}
__global__ void k_bump_multi(AdamMulti A) { if ((int)threadIdx.x < A.n_t) *A.ctr[threadIdx.x] += 1; }

}  // namespace

extern "C" {

int ggad_spmm_seg_len(void) { return SPMM_SEG; }
int ggad_spmm_sliced_seg_len(void) { return SPMM_LONG_SEG; }

int ggad_spmm_csr_f32(const int32_t *col, const float *val, const int32_t *seg_beg, const int32_t *seg_end,
                      const int32_t *seg_out, int32_t n_seg, const int32_t *multi_row, const int32_t *multi_first,
                      const int32_t *multi_count, int32_t n_multi, const float *X, int64_t ldx, int32_t W, const float *bias,
                      const float *prelu_a, float *out, int64_t ldo, float *out_pre, float *part, ggad_stream_t stream) {
  GGAD_REQUIRE(col && seg_beg && seg_end && seg_out && X && out && W >= 4 && (W & 3) == 0 && W <= 256 * SPMM_MAXCH);
  GGAD_REQUIRE((ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= W && ldo >= W && n_seg >= 0 && n_multi >= 0);
  GGAD_REQUIRE(n_multi == 0 || (multi_row && multi_first && multi_count && part));
  if (n_seg == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  k_spmm_seg<<<dim3((n_seg + 3) / 4), dim3(256), 0, st>>>(col, val, seg_beg, seg_end, seg_out, n_seg, X, ldx, W, bias, prelu_a, out,
                                                         ldo, out_pre, part);
  if (n_multi > 0)
    k_spmm_combine<<<dim3((n_multi + 3) / 4), dim3(256), 0, st>>>(multi_row, multi_first, multi_count, n_multi, part, W, bias,
                                                                 prelu_a, out, ldo, out_pre);
  GGAD_CHECK_LAUNCH("spmm_csr_f32");
  return GGAD_OK;
}

int32_t ggad_spmm_rowslice_group(void) { return RS_G40; }
int32_t ggad_spmm_rowslice_short(void) { return RS_SHORT; }
int32_t ggad_spmm_rowslice_long(void) { return RS_LONG; }

int ggad_spmm_rowslice_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *unit_rows,
                           const int32_t *unit_out, int32_t n_units, const int32_t *long_rows, const int32_t *long_out, int32_t n_long,
                           const int32_t *hub_rows, const int32_t *hub_out, int32_t n_hub, const float *X, int64_t ldx, int32_t W,
                           const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && X && out && W >= 4 && (W & 3) == 0 && n_units >= 0 && n_long >= 0 && n_hub >= 0);
  GGAD_REQUIRE((n_units == 0 || (unit_rows && unit_out)) && (n_long == 0 || (long_rows && long_out)) && (n_hub == 0 || (hub_rows && hub_out)));
  GGAD_REQUIRE((ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= W && ldo >= W);
  if (n_units + n_long + n_hub == 0) return GGAD_OK;
  const int n_slices = (W + RS_W - 1) / RS_W;
  const int64_t blocks = (int64_t)n_slices * ((n_units + n_long + 3) / 4 + n_hub);
  GGAD_REQUIRE(blocks < (1ll << 31));
  k_spmm_rowslice<RS_Q40, RS_G40><<<dim3((unsigned)blocks), dim3(256), 0, as_stream(stream)>>>(
      rowptr, col, val, unit_rows, unit_out, n_units, long_rows, long_out, n_long, hub_rows, hub_out, n_hub, n_slices, X, ldx, W, bias,
      prelu_a, out, ldo, out_pre);
  GGAD_CHECK_LAUNCH("spmm_rowslice_f32");
  return GGAD_OK;
}

/* The line-granular persistent variant (k_spmm_rowline): X rows 128-byte aligned (pointer and ldx * 4), 256 <= W <= 512, the whole
 * operand addressable with 32 bits.  ent: (column, value bits) per entry, CSR order.  Tables of (first entry, end, output row, 0):
 * unit_tab 8 slots per unit of short rows (empty slot: 0, 0, -1), long_tab / hub_tab one per row. */
int32_t ggad_spmm_rowline_supported(const float *X, int64_t ldx, int32_t W, int64_t n_src_rows) {
  return (X && (((uintptr_t)X) & 127) == 0 && (ldx & 31) == 0 && ldx >= W && ldx <= 4096 && W >= 256 && W <= 512 && (W & 3) == 0 &&
          n_src_rows >= 1 && n_src_rows < (1 << 24) && n_src_rows * ldx * 4 < (1ll << 32)) ? 1 : 0;
}

int ggad_spmm_rowline_f32(const int32_t *ent, const int32_t *unit_tab, int32_t n_units, const int32_t *long_tab, int32_t n_long,
                          const int32_t *hub_tab, int32_t n_hub, const float *X, int64_t ldx, int32_t W, int64_t n_src_rows,
                          const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre, ggad_stream_t stream) {
  GGAD_REQUIRE(ent && X && out && n_units >= 0 && n_long >= 0 && n_hub >= 0 && ggad_spmm_rowline_supported(X, ldx, W, n_src_rows));
  GGAD_REQUIRE((n_units == 0 || unit_tab) && (n_long == 0 || long_tab) && (n_hub == 0 || hub_tab));
  GGAD_REQUIRE((ldo & 3) == 0 && ldo >= W && ((((uintptr_t)unit_tab) | ((uintptr_t)long_tab) | ((uintptr_t)hub_tab)) & 15) == 0 && (((uintptr_t)ent) & 7) == 0);
  if (n_units + n_long + n_hub == 0) return GGAD_OK;
  static const int env_bpx = [] { const char *e = getenv("GGAD_ROWLINE_BPX"); return e ? atoi(e) : 0; }();
  const int n_lines = (W + 31) / 32, nw = 4;
  const int64_t items = (int64_t)(n_units + n_long) * 5 / 4 + 1;           // per XCD at 10 lines
  int bpx = env_bpx > 0 ? env_bpx : 256;                                   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  bpx = (int)std::max<int64_t>(1, std::min<int64_t>(bpx, std::max<int64_t>((items + nw - 1) / nw, n_hub)));
  k_spmm_rowline<4><<<dim3(8u * bpx), dim3(256), 0, as_stream(stream)>>>(
      reinterpret_cast<const int2 *>(ent), reinterpret_cast<const int4 *>(unit_tab), n_units, reinterpret_cast<const int4 *>(long_tab), n_long,
      reinterpret_cast<const int4 *>(hub_tab), n_hub, n_lines, X, ldx, W, bias, prelu_a, out, ldo, out_pre);
  GGAD_CHECK_LAUNCH("spmm_rowline_f32");
  return GGAD_OK;
}

#ifdef GGAD_RL_PROF
int ggad_debug_rowline_prof(unsigned long long *dst, int32_t n_words) {      // scripts/rowline_clocks.py (GGAD_EXTRA_HIPFLAGS=-DGGAD_RL_PROF builds only)
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rl_prof), sizeof(unsigned long long) * n_words) == hipSuccess ? 0 : -1;
}
#endif

int64_t ggad_spmm_sliced_workspace_elems(int64_t n_src_rows, int32_t W) {
  return n_src_rows * spmm_n_slices(W) * SPMM_SL * 4;
}

int ggad_spmm_sliced_f32(const int32_t *col, const float *val, const int32_t *seg_beg, const int32_t *seg_end,
                         const int32_t *seg_out, int32_t n_seg, const int32_t *multi_row, const int32_t *multi_first,
                         const int32_t *multi_count, int32_t n_multi, const float *X, int64_t ldx, int32_t W, int64_t n_src_rows,
                         float *xs_workspace, const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre,
                         float *part, ggad_stream_t stream) {
  GGAD_REQUIRE(col && seg_beg && seg_end && seg_out && X && out && xs_workspace && W >= 4 && (W & 3) == 0);
  GGAD_REQUIRE((ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= W && ldo >= W && n_seg >= 0 && n_multi >= 0 && n_src_rows >= 1);
  GGAD_REQUIRE(n_multi == 0 || (multi_row && multi_first && multi_count && part));
  const int S = spmm_n_slices(W);
  GGAD_REQUIRE(n_src_rows * S * SPMM_SL < (1ll << 31));
  if (n_seg == 0) return GGAD_OK;
  const int64_t nb = spmm_sliced_blocks(n_seg, S);
  GGAD_REQUIRE(nb < (1ll << 31));
  hipStream_t st = as_stream(stream);
  const int64_t nt = n_src_rows * S * SPMM_SL;
  k_slice_major<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st>>>(X, ldx, (int)n_src_rows, W, S, nullptr,
                                                                        reinterpret_cast<float4 *>(xs_workspace));
  k_spmm_sliced<<<dim3((unsigned)nb), dim3(256), 0, st>>>(col, val, seg_beg, seg_end, seg_out, n_seg,
                                                         reinterpret_cast<const float4 *>(xs_workspace), (int)n_src_rows, S, W, bias,
                                                         prelu_a, out, ldo, out_pre, part);
  if (n_multi > 0)
    k_spmm_combine<<<dim3((n_multi + 3) / 4), dim3(256), 0, st>>>(multi_row, multi_first, multi_count, n_multi, part, W, bias,
                                                                 prelu_a, out, ldo, out_pre);
  GGAD_CHECK_LAUNCH("spmm_sliced_f32");
  return GGAD_OK;
}

// k_spmm_panel needs PAN_LDS (159 KB) of dynamic LDS per workgroup: the attribute is set -- and its result checked -- once per DEVICE
// (0 unknown, 1 ready, -1 not available), under a mutex
static bool panel_lds_ready() {
  static std::mutex mu;
  static int state[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (state[dev] == 0) {
    int max_lds = 0;
    bool ok = hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess;
    (void)hipGetLastError();
    // (the attribute reports the default limit on some stacks; the authoritative answer is whether the opt-in succeeds)
    ok = hipFuncSetAttribute((const void *)k_spmm_panel, hipFuncAttributeMaxDynamicSharedMemorySize, PAN_LDS) == hipSuccess;
    (void)hipGetLastError();
    state[dev] = ok ? 1 : -1;
  }
  return state[dev] == 1;
}
int32_t ggad_spmm_panel_available(void) { return panel_lds_ready() ? 1 : 0; }

int32_t ggad_spmm_panel_rows(void) { return PAN_R; }
int32_t ggad_spmm_panel_waves(void) { return PAN_WAVES; }
int32_t ggad_spmm_panel_rounds(void) { return PAN_KR; }

int ggad_spmm_panel_f32(const int32_t *wg_tab, int32_t n_wg, const uint32_t *dir, const uint32_t *stream, const int32_t *row_tab,
                        int32_t n_chunks, const float *col_scale, const float *row_scale, const float *diag, const float *X,
                        int64_t ldx, int32_t W, int64_t n_src_rows, float *xs_workspace, const float *bias, const float *prelu_a,
                        float *out, int64_t ldo, float *out_pre, ggad_stream_t stream_) {
  GGAD_REQUIRE(wg_tab && dir && stream && row_tab && X && out && xs_workspace && W >= 4 && (W & 3) == 0 && n_wg >= 0);
  GGAD_REQUIRE((ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= W && ldo >= W && n_src_rows >= 1);
  const int S = spmm_n_slices(W);
  GGAD_REQUIRE(n_src_rows * S * SPMM_SL < (1ll << 31));
  GGAD_REQUIRE(n_chunks == (int32_t)((n_src_rows + PAN_R - 1) / PAN_R));
  if (n_wg == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream_);
  if (!panel_lds_ready()) return GGAD_E_INVALID;          // this device cannot give a workgroup PAN_LDS bytes: callers take the sliced kernel
  const int64_t nt = n_src_rows * S * SPMM_SL;
  k_slice_major<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st>>>(X, ldx, (int)n_src_rows, W, S, col_scale,
                                                                        reinterpret_cast<float4 *>(xs_workspace));
  k_spmm_panel<<<dim3((unsigned)n_wg), dim3(1024), PAN_LDS, st>>>(wg_tab, dir, reinterpret_cast<const uint4 *>(stream), row_tab, n_chunks,
                                                                 reinterpret_cast<const float4 *>(xs_workspace), (int)n_src_rows, W,
                                                                 row_scale, diag, X, ldx, bias, prelu_a, out, ldo, out_pre);
  GGAD_CHECK_LAUNCH("spmm_panel_f32");
  return GGAD_OK;
}

// k_spmm_ring: the same opt-in for its RING_LDS bytes (per device, under a mutex)
static bool ring_lds_ready() {
  static std::mutex mu;
  static int state[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (state[dev] == 0) {
    const bool ok = hipFuncSetAttribute((const void *)k_spmm_ring<RING_WALK>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS) == hipSuccess &&
                    hipFuncSetAttribute((const void *)k_spmm_ring<RING_WALK_SUBSET>, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS) == hipSuccess;
    (void)hipGetLastError();
    state[dev] = ok ? 1 : -1;
  }
  return state[dev] == 1;
}
int32_t ggad_spmm_ring_available(void) { return ring_lds_ready() ? 1 : 0; }
int32_t ggad_spmm_ring_slot_rows(void) { return RING_RS; }
int32_t ggad_spmm_ring_slots(void) { return RING_S; }
int32_t ggad_spmm_ring_window(void) { return RING_V; }
int32_t ggad_spmm_ring_walkers(void) { return RING_WALK; }
int32_t ggad_spmm_ring_walkers_subset(void) { return RING_WALK_SUBSET; }
int32_t ggad_spmm_ring_rounds(void) { return RING_KR; }

int ggad_spmm_ring_f32(const int32_t *wg_tab, int32_t n_wg, const int32_t *wave_sb, const uint16_t *idx, const uint32_t *ctl,
                       const int32_t *row_tab, int32_t n_phases, int32_t n_walkers, const float *col_scale, const float *row_scale,
                       const float *diag, const float *X, int64_t ldx, int32_t W, int64_t n_src_rows, float *xs_workspace,
                       const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre, ggad_stream_t stream_) {
  GGAD_REQUIRE(n_walkers == RING_WALK || n_walkers == RING_WALK_SUBSET);      /* the plan was dealt to that many walkers per workgroup */
  GGAD_REQUIRE(wg_tab && wave_sb && idx && ctl && row_tab && X && out && xs_workspace && W >= 4 && (W & 3) == 0 && n_wg >= 0);
  GGAD_REQUIRE((ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= W && ldo >= W && n_src_rows >= 1);
  const int S = spmm_n_slices(W);
  GGAD_REQUIRE(n_src_rows * S * SPMM_SL < (1ll << 31));
  GGAD_REQUIRE(n_phases == (int32_t)((n_src_rows + RING_RS - 1) / RING_RS));
  if (n_wg == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream_);
  if (!ring_lds_ready()) return GGAD_E_INVALID;
  const int64_t nt = n_src_rows * S * SPMM_SL;
  k_slice_major<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st>>>(X, ldx, (int)n_src_rows, W, S, col_scale,
                                                                        reinterpret_cast<float4 *>(xs_workspace));
  if (n_walkers == RING_WALK)
    k_spmm_ring<RING_WALK><<<dim3((unsigned)n_wg), dim3(1024), RING_LDS, st>>>(wg_tab, wave_sb, idx, ctl, row_tab, n_phases,
                                                                            reinterpret_cast<const float4 *>(xs_workspace), (int)n_src_rows, W,
                                                                            row_scale, diag, X, ldx, bias, prelu_a, out, ldo, out_pre);
  else
    k_spmm_ring<RING_WALK_SUBSET><<<dim3((unsigned)n_wg), dim3(1024), RING_LDS, st>>>(wg_tab, wave_sb, idx, ctl, row_tab, n_phases,
                                                                                   reinterpret_cast<const float4 *>(xs_workspace), (int)n_src_rows,
                                                                                   W, row_scale, diag, X, ldx, bias, prelu_a, out, ldo, out_pre);
  GGAD_CHECK_LAUNCH("spmm_ring_f32");
  return GGAD_OK;
}

int32_t ggad_prelu_bwd_splits(int32_t M) { int s = (M + 63) / 64; return s < 1 ? 1 : (s > 256 ? 256 : s); }

int ggad_prelu_bwd_ld_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, int64_t ld_dz, float *db,
                          float *da, float *workspace, ggad_stream_t stream) {
  GGAD_REQUIRE(g && z && prelu_a && dz && workspace && M >= 1 && W >= 1 && ld_dz >= W);
  const int S = ggad_prelu_bwd_splits(M);
  float *pdb = workspace, *pda = workspace + (int64_t)S * W;
  hipStream_t st = as_stream(stream);
  if ((W & 3) == 0 && (ld_dz & 3) == 0 && W <= 1024 && (((uintptr_t)g | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)workspace) & 15) == 0 && (((int64_t)S * W) & 3) == 0)
    k_prelu_bwd_v4<<<dim3(S), dim3(256), 0, st>>>(reinterpret_cast<const float4 *>(g), reinterpret_cast<const float4 *>(z), prelu_a, M, W >> 2,
                                                  reinterpret_cast<float4 *>(dz), ld_dz >> 2, reinterpret_cast<float4 *>(pdb), reinterpret_cast<float4 *>(pda));
  else {
    GGAD_REQUIRE(ld_dz == W);                                    // (the scalar kernel writes dense rows)
    k_prelu_bwd<<<dim3((W + 63) / 64, S), dim3(256), 0, st>>>(g, z, prelu_a, M, W, dz, pdb, pda);
  }
  if (W <= 1024)
    k_prelu_bwd_final<<<dim3(1), dim3(1024), 0, st>>>(pdb, pda, S, W, db, da);
  else
    k_prelu_bwd_final_wide<<<dim3(1), dim3(1024), 0, st>>>(pdb, pda, S, W, db, da);      // layers wider than 1,024 (tam.py --embedding_dim > 512)
  GGAD_CHECK_LAUNCH("prelu_bwd_f32");
  return GGAD_OK;
}

// workgroups of the one-launch kernel: one per pass of 1024 / (W / 4) rows up to PB_MAXS (>= one per compute unit on every layer here)
static int prelu_one_blocks(int32_t M, int32_t W) {
  const int rp = std::max(1, 1024 / std::max(1, W >> 2));
  return std::max(1, std::min(PB_MAXS, (M + rp - 1) / rp));
}
/* floats of `workspace` and int32 of `ticket` ggad_prelu_bwd_one_f32 needs (>= what ggad_prelu_bwd_ld_f32 needs: its fallback) */
int64_t ggad_prelu_bwd_one_workspace_elems(int32_t M, int32_t W) {
  const int64_t S = prelu_one_blocks(M, W), NG = (S + PB_GRP - 1) / PB_GRP;
  return std::max<int64_t>(2 * (S + NG) * (int64_t)W, 2 * (int64_t)ggad_prelu_bwd_splits(M) * W);
}
int32_t ggad_prelu_bwd_one_tickets(void) { return 1 + (PB_MAXS + PB_GRP - 1) / PB_GRP; }

/* ggad_prelu_bwd_ld_f32 in ONE launch where the vector kernel takes the shape (W % 4 == 0, W <= 1024, 16-byte aligned operands: every
 * layer width of the path); elsewhere the two launches above.  `ticket`: ggad_prelu_bwd_one_tickets() int32 in device memory, zero before
 * the first call and left zero (not shared between launches that may run concurrently); `workspace`: ggad_prelu_bwd_one_workspace_elems. */
int ggad_prelu_bwd_one_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, int64_t ld_dz, float *db,
                           float *da, float *workspace, int32_t *ticket, ggad_stream_t stream) {
  GGAD_REQUIRE(g && z && prelu_a && dz && workspace && ticket && M >= 1 && W >= 1 && ld_dz >= W);
  const int S0 = ggad_prelu_bwd_splits(M);
  if (!((W & 3) == 0 && (ld_dz & 3) == 0 && W <= 1024 && (((uintptr_t)g | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)workspace) & 15) == 0 &&
        (((int64_t)S0 * W) & 3) == 0))
    return ggad_prelu_bwd_ld_f32(g, z, prelu_a, M, W, dz, ld_dz, db, da, workspace, stream);
  const int S = prelu_one_blocks(M, W), NG = (S + PB_GRP - 1) / PB_GRP;
  float *pdb = workspace, *pda = pdb + (int64_t)S * W, *gdb = pda + (int64_t)S * W, *gda = gdb + (int64_t)NG * W;
  k_prelu_bwd_one<<<dim3(S), dim3(1024), 0, as_stream(stream)>>>(reinterpret_cast<const float4 *>(g), reinterpret_cast<const float4 *>(z), prelu_a,
                                                                M, W >> 2, reinterpret_cast<float4 *>(dz), ld_dz >> 2,
                                                                reinterpret_cast<float4 *>(pdb), reinterpret_cast<float4 *>(pda),
                                                                reinterpret_cast<float4 *>(gdb), reinterpret_cast<float4 *>(gda), db, da, ticket);
  GGAD_CHECK_LAUNCH("prelu_bwd_one_f32");
  return GGAD_OK;
}

int ggad_prelu_bwd_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, float *db,
                       float *da, float *workspace, ggad_stream_t stream) {
  return ggad_prelu_bwd_ld_f32(g, z, prelu_a, M, W, dz, W, db, da, workspace, stream);
}

int ggad_relu_bwd_f32(const float *g, const float *y, int64_t n, float *dz, ggad_stream_t stream) {
  GGAD_REQUIRE(g && y && dz && n >= 0);
  if (n == 0) return GGAD_OK;
  k_relu_bwd<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(g, y, n, dz);
  GGAD_CHECK_LAUNCH("relu_bwd_f32");
  return GGAD_OK;
}

int ggad_head_gather_f32(const float *X, const int32_t *idx, const float *add, int32_t n, int32_t W, float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(X && idx && out && n >= 0 && W >= 1);
  if (n == 0) return GGAD_OK;
  k_head_gather<<<dim3((n + 3) / 4), dim3(256), 0, as_stream(stream)>>>(X, idx, add, n, W, out);
  GGAD_CHECK_LAUNCH("head_gather_f32");
  return GGAD_OK;
}
int ggad_head_rows_f32(const float *X, const int32_t *abn, const float *add, int32_t n_abn, const int32_t *nrm, int32_t n_nrm, int32_t W,
                       float *out_abn, float *out_comb, ggad_stream_t stream) {
  GGAD_REQUIRE(X && n_abn >= 0 && n_nrm >= 0 && W >= 1 && (n_abn == 0 || (abn && out_abn)) && (n_nrm == 0 || (nrm && out_comb)));
  if (n_abn + n_nrm == 0) return GGAD_OK;
  k_head_rows<<<dim3((n_abn + n_nrm + 3) / 4), dim3(256), 0, as_stream(stream)>>>(X, abn, add, n_abn, nrm, n_nrm, W, out_abn, out_comb);
  GGAD_CHECK_LAUNCH("head_rows_f32");
  return GGAD_OK;
}
int ggad_head_emb_put_f32(const float *con, const int32_t *abn, int32_t n_abn, int32_t W, float *X, ggad_stream_t stream) {
  GGAD_REQUIRE(con && abn && X && n_abn >= 0 && W >= 1);
  if (n_abn == 0) return GGAD_OK;
  k_head_emb_put<<<dim3((n_abn + 3) / 4), dim3(256), 0, as_stream(stream)>>>(con, abn, n_abn, W, X);
  GGAD_CHECK_LAUNCH("head_emb_put_f32");
  return GGAD_OK;
}
int ggad_head_combine_f32(const float *X, const int32_t *nrm, int32_t n_nrm, const float *con, int32_t n_con, int32_t W, float *out,
                          ggad_stream_t stream) {
  GGAD_REQUIRE(X && out && n_nrm >= 0 && n_con >= 0 && W >= 1 && (n_nrm == 0 || nrm) && (n_con == 0 || con));
  if (n_nrm + n_con == 0) return GGAD_OK;
  k_head_combine<<<dim3((n_nrm + n_con + 3) / 4), dim3(256), 0, as_stream(stream)>>>(X, nrm, n_nrm, con, n_con, W, out);
  GGAD_CHECK_LAUNCH("head_combine_f32");
  return GGAD_OK;
}
int ggad_head_emb_out_f32(const float *X, const int32_t *abn_pos, const float *con, int32_t n, int32_t W, float *out,
                          ggad_stream_t stream) {
  GGAD_REQUIRE(X && abn_pos && con && out && n >= 0 && W >= 1);
  if (n == 0) return GGAD_OK;
  k_head_emb_out<<<dim3((n + 3) / 4), dim3(256), 0, as_stream(stream)>>>(X, abn_pos, con, n, W, out);
  GGAD_CHECK_LAUNCH("head_emb_out_f32");
  return GGAD_OK;
}
int ggad_head_con_grad_f32(const float *g_con, const float *g_out, const int32_t *abn, const float *g_tail, const float *y, int32_t n,
                           int32_t W, float *dz, ggad_stream_t stream) {
  GGAD_REQUIRE(y && dz && n >= 0 && W >= 1 && (!g_out || abn));
  if (n == 0) return GGAD_OK;
  k_head_con_grad<<<dim3((n + 3) / 4), dim3(256), 0, as_stream(stream)>>>(g_con, g_out, abn, g_tail, y, n, W, dz);
  GGAD_CHECK_LAUNCH("head_con_grad_f32");
  return GGAD_OK;
}
int ggad_head_emb_grad_f32(const float *g_out, const int32_t *abn_pos, const int32_t *nrm_pos, const float *g_comb, const float *g_abn,
                           const float *sp, int32_t n, int32_t W, float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(abn_pos && nrm_pos && out && n >= 0 && W >= 1);
  if (n == 0) return GGAD_OK;
  k_head_emb_grad<<<dim3((n + 3) / 4), dim3(256), 0, as_stream(stream)>>>(g_out, abn_pos, nrm_pos, g_comb, g_abn, sp, n, W, out);
  GGAD_CHECK_LAUNCH("head_emb_grad_f32");
  return GGAD_OK;
}

int ggad_prelu_fwd_f32(const float *z, const float *prelu_a, int64_t n, float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(z && prelu_a && out && n >= 0);
  if (n == 0) return GGAD_OK;
  if ((n & 3) == 0 && ((((uintptr_t)z) | ((uintptr_t)out)) & 15) == 0)
    k_prelu_fwd_v4<<<dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(reinterpret_cast<const float4 *>(z), prelu_a, n / 4,
                                                                                              reinterpret_cast<float4 *>(out));
  else
    k_prelu_fwd<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(z, prelu_a, n, out);
  GGAD_CHECK_LAUNCH("prelu_fwd_f32");
  return GGAD_OK;
}

int ggad_rownorm_f32(const float *X, int32_t M, int32_t W, float *inv, float *Xn, ggad_stream_t stream) {
  GGAD_REQUIRE(X && inv && Xn && M >= 0 && W >= 1);
  if (M == 0) return GGAD_OK;
  k_rownorm<<<dim3((M + 3) / 4), dim3(256), 0, as_stream(stream)>>>(X, M, W, inv, Xn);
  GGAD_CHECK_LAUNCH("rownorm_f32");
  return GGAD_OK;
}

int ggad_rownorm_bwd_f32(const float *Xn, const float *inv, const float *dXn, int32_t M, int32_t W, float *dX,
                         ggad_stream_t stream) {
  GGAD_REQUIRE(Xn && inv && dXn && dX && M >= 0 && W >= 1);
  if (M == 0) return GGAD_OK;
  k_rownorm_bwd<<<dim3((M + 3) / 4), dim3(256), 0, as_stream(stream)>>>(Xn, inv, dXn, M, W, dX);
  GGAD_CHECK_LAUNCH("rownorm_bwd_f32");
  return GGAD_OK;
}

int ggad_edge_dist_f32(const int32_t *rowptr, const int32_t *col, const float *X, int32_t n_rows, int32_t W, float *out,
                       ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && X && out && n_rows >= 0 && W >= 1);
  if (n_rows == 0) return GGAD_OK;
  k_edge_dist<<<dim3(n_rows), dim3(256), 0, as_stream(stream)>>>(rowptr, col, X, n_rows, W, out);
  GGAD_CHECK_LAUNCH("edge_dist_f32");
  return GGAD_OK;
}

int ggad_rowdot_f32(const float *A, const int32_t *sel, const float *B, int32_t n, int32_t W, const float *scale, float *out,
                    ggad_stream_t stream) {
  GGAD_REQUIRE(A && B && out && n >= 0 && W >= 1);
  if (n == 0) return GGAD_OK;
  k_rowdot<<<dim3((n + 3) / 4), dim3(256), 0, as_stream(stream)>>>(A, sel, B, n, W, scale, out);
  GGAD_CHECK_LAUNCH("rowdot_f32");
  return GGAD_OK;
}

int ggad_rows_scale_f32(const float *X, const int32_t *sel, const float *coef, int32_t n, int32_t W, int32_t scatter_add,
                        float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(X && sel && out && n >= 0 && W >= 1);
  if (n == 0) return GGAD_OK;
  k_rows_scale<<<dim3((n + 3) / 4), dim3(256), 0, as_stream(stream)>>>(X, sel, coef, n, W, scatter_add, out);
  GGAD_CHECK_LAUNCH("rows_scale_f32");
  return GGAD_OK;
}

int64_t ggad_full_loss_workspace_elems(int32_t n_out, int32_t H) { return (int64_t)((n_out + REC_ROWS - 1) / REC_ROWS) * H; }

int ggad_full_loss_f32(const float *logits, const float *aff, int32_t n_normal, int32_t n_out, const float *emb_con,
                       const float *emb_abn, int32_t H, float margin, float *losses4, float *d_logits, float *g_aff,
                       float *dD, float *workspace, ggad_stream_t stream) {
  GGAD_REQUIRE(logits && aff && emb_con && emb_abn && losses4 && d_logits && g_aff && dD && workspace);
  GGAD_REQUIRE(n_normal >= 1 && n_out >= 1 && H >= 1);
  hipStream_t st = as_stream(stream);
  const int nb = (n_out + REC_ROWS - 1) / REC_ROWS;
  k_full_loss<<<dim3(1), dim3(1024), 0, st>>>(logits, aff, n_normal, n_out, margin, losses4, d_logits, g_aff);
  k_rec_part<<<dim3(nb), dim3(256), 0, st>>>(emb_con, emb_abn, n_out, H, workspace);
  k_rec_apply<<<dim3(nb), dim3(256), 0, st>>>(emb_con, emb_abn, n_out, H, workspace, nb, dD, losses4);
  GGAD_CHECK_LAUNCH("full_loss_f32");
  return GGAD_OK;
}

int ggad_full_loss_bwd_scale_f32(const float *g_total, const float *g_aff, const float *r_inv_j, const float *d_logits, const float *dD,
                                 int32_t L, int64_t n_rec, float *c, float *dl, float *d_con, float *d_abn, ggad_stream_t stream) {
  GGAD_REQUIRE(g_total && g_aff && r_inv_j && d_logits && dD && c && dl && d_con && d_abn && L >= 0 && n_rec >= 0);
  const int64_t n = std::max<int64_t>(L, n_rec);
  if (n == 0) return GGAD_OK;
  k_loss_bwd_scale<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(g_total, g_aff, r_inv_j, d_logits, dD, L, n_rec,
                                                                                         c, dl, d_con, d_abn);
  GGAD_CHECK_LAUNCH("full_loss_bwd_scale_f32");
  return GGAD_OK;
}

/* round 6: the loss block of run.py:165-210 behind ONE forward and ONE backward launch for its row-local parts (the two products with R
 * stay launches of their own).  ggad_full_loss_fused_workspace_elems floats of `workspace` + one zero-initialised int32 `ticket` (the
 * kernel leaves it zero).  Outputs of the forward: aff[L], kcol[H], losses4 = {total, margin, bce, rec}, d_logits[L], g_aff[L]. */
int64_t ggad_full_loss_fused_workspace_elems(int32_t n_out, int32_t H) { return (int64_t)((n_out + REC_ROWS - 1) / REC_ROWS) * H; }

int ggad_full_loss_fused_f32(const float *e_hat, const int32_t *J, const float *S, const float *r_inv_j, int32_t n_normal, int32_t n_out,
                             int32_t H, const float *logits, const float *emb_con, const float *emb_abn, float margin, float *aff,
                             float *kcol, float *losses4, float *d_logits, float *g_aff, float *workspace, int32_t *ticket,
                             ggad_stream_t stream) {
  GGAD_REQUIRE(e_hat && J && S && r_inv_j && logits && emb_con && emb_abn && aff && kcol && losses4 && d_logits && g_aff && workspace && ticket);
  GGAD_REQUIRE(n_normal >= 1 && n_out >= 1 && H >= 1);
  GGAD_REQUIRE(H <= LF_T);                                    /* one thread per column in the last workgroup */
  const int L = n_normal + n_out;
  const int nb_dot = (L + LD_ROWS - 1) / LD_ROWS, nb_part = (n_out + REC_ROWS - 1) / REC_ROWS;
  k_loss_fwd_fused<<<dim3(nb_dot + (nb_part + 3) / 4), dim3(LF_T), 0, as_stream(stream)>>>(e_hat, J, S, r_inv_j, L, H, logits, n_normal, n_out,
                                                                                        emb_con, emb_abn, margin, nb_dot, nb_part, aff,
                                                                                        workspace, kcol, losses4, d_logits, g_aff, ticket);
  GGAD_CHECK_LAUNCH("full_loss_fused_f32");
  return GGAD_OK;
}

int ggad_full_loss_bwd_fused_f32(const float *g_total, const float *g_aff, const float *r_inv_j, const float *d_logits, const float *e_hat,
                                 const int32_t *J, int32_t L, int32_t H, const float *emb_con, const float *emb_abn, const float *kcol,
                                 int32_t n_out, float *c, float *dl, float *xc, float *d_con, float *d_abn, ggad_stream_t stream) {
  GGAD_REQUIRE(g_total && g_aff && r_inv_j && d_logits && e_hat && J && emb_con && emb_abn && kcol && c && dl && xc && d_con && d_abn);
  GGAD_REQUIRE(L >= 1 && H >= 1 && n_out >= 1);
  const int nb_l = (L + 3) / 4, nb_a = (n_out + 3) / 4;
  k_loss_bwd_fused<<<dim3(nb_l + nb_a), dim3(256), 0, as_stream(stream)>>>(g_total, g_aff, r_inv_j, d_logits, e_hat, J, L, H, emb_con, emb_abn,
                                                                          kcol, n_out, nb_l, c, dl, xc, d_con, d_abn);
  GGAD_CHECK_LAUNCH("full_loss_bwd_fused_f32");
  return GGAD_OK;
}

int ggad_rownorm_bwd_add_f32(const float *Xn, const float *inv, const float *dXn, const int32_t *pos_n, const int32_t *pos_a,
                             const float *c, const float *S, int32_t M, int32_t W, float *dX, ggad_stream_t stream) {
  GGAD_REQUIRE(Xn && inv && dXn && pos_n && pos_a && c && S && dX && M >= 0 && W >= 1);
  if (M == 0) return GGAD_OK;
  k_rownorm_bwd_add<<<dim3((M + 3) / 4), dim3(256), 0, as_stream(stream)>>>(Xn, inv, dXn, pos_n, pos_a, c, S, M, W, dX);
  GGAD_CHECK_LAUNCH("rownorm_bwd_add_f32");
  return GGAD_OK;
}

int ggad_adam_f32(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int64_t n, float lr,
                  float weight_decay, int32_t *step_counter, int32_t bump_after, ggad_stream_t stream) {
  GGAD_REQUIRE(params && exp_avg && exp_avg_sq && grads && step_counter && n >= 0);
  if (n == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  // step index used = *step_counter + 1; the counter itself is advanced by a trailing launch when asked to
  k_adam_flat<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(params, exp_avg, exp_avg_sq, grads, n, lr, weight_decay,
                                                                     step_counter, 1);
  if (bump_after) k_bump<<<dim3(1), dim3(1), 0, st>>>(step_counter);
  GGAD_CHECK_LAUNCH("adam_f32");
  return GGAD_OK;
}


int32_t ggad_adam_multi_max(void) { return ADAM_MAX_T; }

/* ggad_adam_f32 (bump_after = 1) for n_tensors <= ggad_adam_multi_max() tensors in one launch (two without `tickets`); the arrays are
 * HOST arrays of device pointers / element counts; every tensor has its own step counter (torch keeps `step` per parameter). */
int ggad_adam_multi_f32(int32_t n_tensors, float *const *params, float *const *exp_avg, float *const *exp_avg_sq,
                        const float *const *grads, const int64_t *n_elems, int32_t *const *step_counters, float lr,
                        float weight_decay, int32_t *tickets, ggad_stream_t stream) {
  GGAD_REQUIRE(n_tensors >= 0 && n_tensors <= ADAM_MAX_T && params && exp_avg && exp_avg_sq && grads && n_elems && step_counters);
  if (n_tensors == 0) return GGAD_OK;
  AdamMulti A;
  int blocks = 0, k = 0;
  for (int t = 0; t < n_tensors; ++t) {
    GGAD_REQUIRE(params[t] && exp_avg[t] && exp_avg_sq[t] && grads[t] && step_counters[t] && n_elems[t] >= 0);
    if (n_elems[t] == 0) continue;
    A.p[k] = params[t]; A.m[k] = exp_avg[t]; A.v[k] = exp_avg_sq[t]; A.g[k] = grads[t]; A.ctr[k] = step_counters[t];
    A.n[k] = n_elems[t];
    A.blk0[k] = blocks;
    blocks += (int)((n_elems[t] + ADAM_EPB - 1) / ADAM_EPB);
    ++k;
  }
  if (k == 0) return GGAD_OK;
  A.blk0[k] = blocks;
  A.n_t = k;
  hipStream_t st = as_stream(stream);
  // `tickets` (optional; ggad_adam_multi_max() int32, zero before the first call, left zero): the counters advance inside the launch
  k_adam_multi<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(A, lr, weight_decay, tickets);
  if (!tickets) k_bump_multi<<<dim3(1), dim3(64), 0, st>>>(A);
  GGAD_CHECK_LAUNCH("adam_multi_f32");
  return GGAD_OK;
}

}  // extern "C"
