// Batch sub-graph plan + 1-hop gather-aggregate kernels of the DGraph mini-batch path (gfx950), and the device-atomic
// fallback of the 2-hop stage.
//
// Replaces GCNAggregator.forward of the reference (src/graphsage.py:295-360): python set unions, a dense B x U and a
// dense U x U2 0/1 mask, their row/column sums, and two dense mask.mm(feature) products -- by CSR walks, per-batch
// integer histograms in HBM-resident counter slots and wave-level gathers of feature rows.
//
// Unit of work = a PIECE of <= 16 consecutive entries of one batch row (tables built on the host by ggad_mb_plan_build:
// closed degrees are known there, so the entry offsets need no device scan).  16 lanes own one piece, a wave four of them:
//   k_expand          1 lane / entry : the entry's column (closed neighbourhood N(i)+{i}, ascending), histogram c_j,
//                                      owner election
//   k_gather1c        1 lane / entry : c_j, owner, weight; owner metadata + pair-count storage + per-node owner lists for the
//                                      2-hop stage; then the wave gathers the feature rows of its four pieces
//                                      (3 rows of 17 floats per load instruction, all loads of a piece in flight)
//   k_combine1_reset  1 wave / row   : x1 of rows made of several pieces (fixed order); 1 lane / entry: c_j slots back to zero
// A 2,000-neighbour hub row is 125 independent pieces instead of one wave's 32 dependent 64-entry blocks (that loop was
// a fixed 200 us of every plan, whatever its size).
// Fallback 2-hop (chunks the LDS-counting stage of hop2_ldsw.hip cannot take):
//   k_count2 / k_gather2 / k_plan_reset   1 wave / entry, device atomics on per-batch counter slots
#include "common.h"

namespace {

// ------------------------------------------------------------------ exclusive scan
constexpr int SCAN_T = 256;      // threads per block
constexpr int SCAN_ITEMS = 8;    // items per thread
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
  // v: per-thread value; returns exclusive prefix within the block (256 threads = 4 waves)
  __shared__ int wsum[SCAN_T / GGAD_WAVE];
  const int lane = lane_id(), wid = threadIdx.x / GGAD_WAVE;
  int inc = v;
#pragma unroll
  for (int off = 1; off < GGAD_WAVE; off <<= 1) {
    int t = __shfl_up(inc, off, GGAD_WAVE);
    if (lane >= off) inc += t;
  }
  if (lane == GGAD_WAVE - 1) wsum[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_T / GGAD_WAVE; ++w) {
    int s = wsum[w];
    if (w < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_T) scan_tile_sums(const int32_t *__restrict__ in, int64_t n, int32_t *__restrict__ tile_sum) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) s += (base + i < n) ? in[base + i] : 0;
  int tot;
  block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

// single block: exclusive scan of up to SCAN_TILE tile sums in place; writes grand total to *total_out
__global__ void __launch_bounds__(SCAN_T) scan_tile_offsets(int32_t *__restrict__ tile_sum, int n_tiles, int32_t *__restrict__ total_out) {
  int v[SCAN_ITEMS];
  const int base = threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n_tiles) ? tile_sum[base + i] : 0; s += v[i]; }
  int tot;
  int ex = block_exclusive_scan(s, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n_tiles) tile_sum[base + i] = ex; ex += v[i]; }
  if (threadIdx.x == 0) *total_out = tot;
}

__global__ void __launch_bounds__(SCAN_T) scan_apply(const int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n,
                                                     const int32_t *__restrict__ tile_off) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = (base + i < n) ? in[base + i] : 0; s += v[i]; }
  int tot;
  int ex = block_exclusive_scan(s, &tot) + tile_off[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < n) out[base + i] = ex; ex += v[i]; }
}

// Weighted gather of feature rows for up to 64 neighbours held one per lane (ids in `j`, weights in `w`,
// w = 0 for padding lanes).  Lane layout: g = lane / F selects one of `rpi` neighbours per instruction,
// f = lane % F the feature.  Accumulates into acc (per lane partial for (g, f)).
__device__ __forceinline__ void gather_block(const float *__restrict__ feat, int F, int fbase, int rpi, int g, int f,
                                             bool lane_active, int j, float w, int count, float &acc) {
  // F here is the ROW STRIDE in floats (== feature width for an unpadded table)
  const int iters = (count + rpi - 1) / rpi;
  int t = 0;
  for (; t + 4 <= iters; t += 4) {
    float x[4], ww[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int src = (t + u) * rpi + g;              // < 64 + rpi; shfl wraps modulo 64, weight 0 guards
      // shuffles are executed by ALL lanes (a bpermute returns 0 from an inactive source lane)
      const int jj = __shfl(j, src & 63, GGAD_WAVE);
      const float ws = __shfl(w, src & 63, GGAD_WAVE);
      ww[u] = (src < count) ? ws : 0.0f;
      x[u] = (lane_active && src < count) ? feat[(int64_t)jj * F + fbase + f] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = fmaf(ww[u], x[u], acc);
  }
  for (; t < iters; ++t) {
    const int src = t * rpi + g;
    const int jj = __shfl(j, src & 63, GGAD_WAVE);
    const float ws = __shfl(w, src & 63, GGAD_WAVE);
    const float ww = (src < count) ? ws : 0.0f;
    const float x = (lane_active && src < count) ? feat[(int64_t)jj * F + fbase + f] : 0.0f;
    acc = fmaf(ww, x, acc);
  }
}

__device__ __forceinline__ float reduce_groups(float acc, int F, int rpi) {
  // sum the rpi group partials into lanes [0, F)
  float tot = acc;
  for (int gg = 1; gg < rpi; ++gg) tot += __shfl(acc, (lane_id() + gg * F) & 63, GGAD_WAVE);
  return tot;
}

// ------------------------------------------------------------------ plan kernels
// counters[] of a plan (GGAD_PLAN_COUNTERS ints, one counter per 64-byte line, zeroed by k_expand): groups, work items, partial slots of the 2-hop gather,
// 3 its work cursor, 4 pair-count storage cursor (pc[] allocation of k_gather1c)
__global__ void __launch_bounds__(256) k_expand(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                const int32_t *__restrict__ nodes, const int32_t *__restrict__ row_slot,
                                                const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ ck_rc,
                                                const int32_t *__restrict__ ck_e0, int n_chunks, int64_t n_nodes,
                                                int32_t *__restrict__ ent_col, int32_t *__restrict__ ent_slot,
                                                int32_t *__restrict__ ent_row, int32_t *__restrict__ cnt1,
                                                int32_t *__restrict__ own1, int32_t *__restrict__ counters, int skip) {
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  if (vbx == 0 && threadIdx.x < GGAD_PLAN_COUNTERS && counters != nullptr) counters[threadIdx.x] = 0;
  const int ck = (int)((vbx * 256u + threadIdx.x) >> 4);
  const int i = threadIdx.x & 15;
  if (ck >= n_chunks) return;
  const int rc = ck_rc[ck];
  if (i >= (rc & 63)) return;
  const int row = rc >> 6;
  const int e = ck_e0[ck] + i;
  const int e_row = ent_ptr[row];
  const int r = ent_ptr[row + 1] - e_row;                 // |N(v) + {v}| (exact, from the host)
  const int idx = e - e_row;
  const int v = nodes[row];
  const int s = rowptr[v];
  const int deg = rowptr[v + 1] - s;
  // element idx of the sorted closed neighbourhood, decided from the two CSR neighbours around it (no search):
  // col[0..p), v, col[p..deg) with p = #{neighbours < v}; r == deg means v is its own neighbour already
  int j;
  if (r == deg) {
    j = col[s + idx];
  } else {
    const int a = idx > 0 ? col[s + idx - 1] : -1;
    const int b = idx < deg ? col[s + idx] : 0x7fffffff;
    j = (b < v) ? b : (a < v ? v : a);
  }
  const int slot = row_slot[row];
  ent_col[e] = j;
  ent_slot[e] = slot;
  ent_row[e] = row;
  const int64_t soff = (int64_t)slot * n_nodes;
  const int old = atomicAdd(&cnt1[soff + j], 1);            // c_j: column sums of the dense B x U mask   graphsage.py:315
  if (old == 0) own1[soff + j] = e;                         // first arrival owns (batch, j)
}

// 1-hop aggregate.  Scalar phase (lane = entry): c_j, owner, weight 1 / (sqrt r_i sqrt c_j) (graphsage.py:314-318); for
// train plans with the LDS-counting 2-hop stage also the owner's CSR row start / degree, the storage of its pair counts in
// pc[] (one atomic per WAVE on the cursor: the layout of pc[] is free, only sequential per owner) and the per-node list of
// owner entries (atomicExch on node_head: order of the list is irrelevant).  Vector phase: the wave gathers the feature
// rows of each of its four pieces, rpi = 64 / F rows per load instruction, and leaves the piece's partial sum in
// part1[piece] -- or directly in x1 when the row is that single piece.
__global__ void __launch_bounds__(256) k_gather1c(const float *__restrict__ feat, int F, int stride,
                                                  const int32_t *__restrict__ row_slot, const int32_t *__restrict__ ent_ptr,
                                                  const int32_t *__restrict__ row_ck_ptr, const int32_t *__restrict__ ck_rc,
                                                  const int32_t *__restrict__ ck_e0, int n_chunks, int64_t n_nodes,
                                                  const int32_t *__restrict__ cnt1, const int32_t *__restrict__ own1,
                                                  const int32_t *__restrict__ ent_col, int32_t *__restrict__ ent_own,
                                                  int32_t *__restrict__ ent_c1, float *__restrict__ part1, int pstride,
                                                  float *__restrict__ x1, const int32_t *__restrict__ rowptr,
                                                  int32_t *__restrict__ own_deg, int32_t *__restrict__ own_rp,
                                                  int32_t *__restrict__ pw_base, int32_t *__restrict__ node_head,
                                                  int32_t *__restrict__ own_next, int32_t *__restrict__ counters, int ldsw, int skip) {
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  const int wave = (int)((vbx * 256u + threadIdx.x) >> 6);
  const int lane = lane_id();
  const int ck = wave * 4 + (lane >> 4), i = lane & 15;
  int len = 0, row = 0, e = 0;
  if (ck < n_chunks) {
    const int rc = ck_rc[ck];
    len = rc & 63;
    row = rc >> 6;
    e = ck_e0[ck] + i;
  }
  const bool valid = i < len;
  int j = 0, deg = 0;
  float w = 0.0f;
  if (valid) {
    j = ent_col[e];
    const int64_t soff = (int64_t)row_slot[row] * n_nodes;
    const int c = cnt1[soff + j];
    const int own = own1[soff + j];
    ent_own[e] = own;
    ent_c1[e] = c;
    const int r = ent_ptr[row + 1] - ent_ptr[row];
    const float inv_sr = 1.0f / sqrtf((float)r);           // mask.div(row_normalized)   graphsage.py:318
    w = inv_sr / sqrtf((float)c);                          // .div(col_normalized)
    if (ldsw) {
      int rp = 0;
      if (own == e) {
        rp = rowptr[j];
        deg = rowptr[j + 1] - rp;
        own_next[e] = atomicExch(&node_head[j], e + 1);
      }
      own_rp[e] = rp;
      own_deg[e] = deg;
    }
  }
  if (ldsw) {   // wave-uniform: storage for the owners' per-pair counts, deg(u) uint16 each
    int inc = deg;
#pragma unroll
    for (int off = 1; off < GGAD_WAVE; off <<= 1) {
      const int t = __shfl_up(inc, off, GGAD_WAVE);
      if (lane >= off) inc += t;
    }
    const int total = __shfl(inc, GGAD_WAVE - 1, GGAD_WAVE);
    // one atomic per WORKGROUP on the cursor (the four waves' totals meet in LDS): same-address atomics are serialised chip-wide
    __shared__ int wg_tot[5];
    const int wid = threadIdx.x >> 6;
    if (lane == 0) wg_tot[wid] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
      for (int w = 0; w < 4; ++w) { const int t = wg_tot[w]; wg_tot[w] = sum; sum += t; }
      wg_tot[4] = sum > 0 ? atomicAdd(&counters[GGAD_CTR_PC], sum) : 0;
    }
    __syncthreads();
    const int base = wg_tot[4] + wg_tot[wid];
    if (valid) pw_base[e] = base + inc - deg;
  }
  const int rpi = F <= 64 ? 64 / F : 1;
  const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
  for (int c4 = 0; c4 < 4; ++c4) {
    const int len_c = __shfl(len, c4 * 16, GGAD_WAVE);      // lane c4*16 holds the piece's length (0: no such piece)
    if (len_c == 0) continue;                               // wave-uniform
    const int row_c = __shfl(row, c4 * 16, GGAD_WAVE);
    const bool single = (row_ck_ptr[row_c + 1] - row_ck_ptr[row_c]) == 1;
    const int iters = (len_c + rpi - 1) / rpi;              // <= 6 for F = 17
    for (int fc = 0; fc < fchunks; ++fc) {
      const int fbase = fc * 64;
      const int fw = F <= 64 ? F : min(64, F - fbase);
      const int g = lane / fw, f = lane - g * fw;
      const bool act = g < rpi;
      const int gl = act ? g : rpi - 1;                       // padding lanes re-read the last group's row (no extra cache line)
      const int fo = act ? f : 0;
      float acc = 0.0f;
      int t = 0;
      // unconditional loads: ids of padding / out-of-piece lanes are 0 or a neighbour of the wave (readable rows), their
      // weight is 0 -- a guarded load costs a branch and a full wait each
      for (; t + 6 <= iters; t += 6) {                      // every row of a 16-entry piece in flight at once (F = 17: 6 loads)
        float x[6], ww[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int si = (t + q) * rpi + g;
          const int jj = __shfl(j, (c4 * 16 + (t + q) * rpi + gl) & 63, GGAD_WAVE);
          const float ws = __shfl(w, (c4 * 16 + si) & 63, GGAD_WAVE);
          ww[q] = (act && si < len_c) ? ws : 0.0f;
          x[q] = feat[(int64_t)jj * stride + fbase + fo];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) acc = fmaf(ww[q], ww[q] != 0.0f ? x[q] : 0.0f, acc);
      }
      for (; t < iters; ++t) {
        const int si = t * rpi + g;
        const int jj = __shfl(j, (c4 * 16 + t * rpi + gl) & 63, GGAD_WAVE);
        const float ws = __shfl(w, (c4 * 16 + si) & 63, GGAD_WAVE);
        const float wv = (act && si < len_c) ? ws : 0.0f;
        const float x = feat[(int64_t)jj * stride + fbase + fo];
        acc = fmaf(wv, wv != 0.0f ? x : 0.0f, acc);
      }
      const float tot = (F <= 64) ? reduce_groups(acc, fw, rpi) : acc;
      if (lane < fw) {
        if (single) x1[(int64_t)row_c * F + fbase + lane] = tot;
        else part1[(int64_t)(wave * 4 + c4) * pstride + fbase + lane] = tot;
      }
    }
  }
}

// Two roles in one launch (both only need k_gather1c to have finished): blocks [0, row_blocks) sum the piece partials of
// multi-piece rows into x1 -- lane group g takes pieces g, g + rpi, ..., the groups are added in order: a fixed order per
// row -- ; the remaining blocks put the c_j counter of every entry's (batch, column) back to zero, so the slots are clean
// for the next plan (the device-atomic 2-hop fallback still needs own1 / the slots and resets later).
__global__ void __launch_bounds__(256) k_combine1_reset(int row_blocks, int n_rows, const int32_t *__restrict__ row_ck_ptr,
                                                        const float *__restrict__ part1, int pstride, int F,
                                                        float *__restrict__ x1, int do_reset, int n_ents,
                                                        const int32_t *__restrict__ ent_col,
                                                        const int32_t *__restrict__ ent_slot, int64_t n_nodes,
                                                        int32_t *__restrict__ cnt1, int skip) {
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  if ((int)vbx >= row_blocks) {
    if (!do_reset) return;
    const int e = (vbx - row_blocks) * 256 + threadIdx.x;
    if (e < n_ents) cnt1[(int64_t)ent_slot[e] * n_nodes + ent_col[e]] = 0;
    return;
  }
  const int row = vbx * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int c0 = row_ck_ptr[row];
  const int nck = row_ck_ptr[row + 1] - c0;
  if (nck <= 1) return;
  const int lane = lane_id();
  const int rpi = F <= 64 ? 64 / F : 1;
  const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
  for (int fc = 0; fc < fchunks; ++fc) {
    const int fbase = fc * 64;
    const int fw = F <= 64 ? F : min(64, F - fbase);
    const int g = lane / fw, f = lane - g * fw;
    float acc = 0.0f;
    if (g < rpi) {
      const float *__restrict__ src = part1 + (int64_t)c0 * pstride + fbase + f;
      int c = g;
      for (; c + 7 * rpi < nck; c += 8 * rpi) {      // eight pieces in flight (a 2,000-entry hub row is 125 pieces: 42 loads one
        float v[8];                                  // after the other per lane group made this launch a fixed ~15 us); same order
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(c + u * rpi) * pstride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; c < nck; c += rpi) acc += src[(int64_t)c * pstride];
    }
    const float tot = (F <= 64) ? reduce_groups(acc, fw, rpi) : acc;
    if (lane < fw) x1[(int64_t)row * F + fbase + lane] = tot;
  }
}

// out[row] = mean of feat rows over an explicit ragged neighbour list (MeanAggregator, graphsage.py:66-99; IntraAgg 1-hop,
// layers.py:213-225), or, with seg_w, the weighted sum over it (IntraAgg 2-hop 1/(sqrt r sqrt c) mask, layers.py:227-242)
__global__ void __launch_bounds__(256) k_seg_mean(const float *__restrict__ feat, int F, const int32_t *__restrict__ seg_ptr,
                                                  const int32_t *__restrict__ seg_col, const float *__restrict__ seg_w,
                                                  int n_rows, float *__restrict__ out) {
  const int row = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (row >= n_rows) return;
  const int lane = lane_id();
  const int e0 = seg_ptr[row], e1 = seg_ptr[row + 1];
  const int r = e1 - e0;
  const float inv = 1.0f / (float)r;                       // mask.div(num_neigh)            graphsage.py:92-93
  const int rpi = F <= 64 ? 64 / F : 1;
  const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
  for (int fc = 0; fc < fchunks; ++fc) {
    const int fbase = fc * 64;
    const int fw = F <= 64 ? F : min(64, F - fbase);
    const int g = lane / fw, f = lane - g * fw;
    const bool lane_active = g < rpi;
    float acc = 0.0f;
    for (int blk = 0; blk < r; blk += GGAD_WAVE) {
      const int idx = blk + lane;
      int j = 0; float w = 0.0f;
      if (idx < r) { j = seg_col[e0 + idx]; w = seg_w ? seg_w[e0 + idx] : inv; }
      gather_block(feat, F, fbase, rpi, g, f, lane_active, j, w, min(GGAD_WAVE, r - blk), acc);
    }
    float tot = (F <= 64) ? reduce_groups(acc, fw, rpi) : acc;
    if (r == 0) tot = inv * 0.0f;                            // empty list: the dense 0/0 mask row -> NaN (quirk 3)
    if (lane < fw) out[(int64_t)row * F + fbase + lane] = tot;
  }
}

__global__ void __launch_bounds__(256) k_count2(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                const int32_t *__restrict__ ent_col, const int32_t *__restrict__ ent_slot,
                                                const int32_t *__restrict__ ent_total, int64_t n_nodes,
                                                const int32_t *__restrict__ own1, int32_t *__restrict__ cnt2) {
  const int e = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (e >= *ent_total) return;
  const int u = ent_col[e];
  const int64_t soff = (int64_t)ent_slot[e] * n_nodes;
  if (own1[soff + u] != e) return;                         // each distinct u of the batch once (set semantics)
  const int s = rowptr[u], t = rowptr[u + 1];
  for (int i = s + lane_id(); i < t; i += GGAD_WAVE) atomicAdd(&cnt2[soff + col[i]], 1);
}

__global__ void __launch_bounds__(256) k_gather2(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const float *__restrict__ feat, int F, int stride, const int32_t *__restrict__ ent_col,
                                                 const int32_t *__restrict__ ent_slot, const int32_t *__restrict__ ent_own,
                                                 const int32_t *__restrict__ ent_total, int64_t n_nodes,
                                                 const int32_t *__restrict__ cnt2, float *__restrict__ x2) {
  const int e = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (e >= *ent_total) return;
  if (ent_own[e] != e) return;
  const int lane = lane_id();
  const int u = ent_col[e];
  const int64_t soff = (int64_t)ent_slot[e] * n_nodes;
  const int s = rowptr[u], t = rowptr[u + 1];
  const int deg = t - s;
  const float inv_sr = 1.0f / sqrtf((float)deg);         // deg = 0 -> inf * 0 = NaN, as the dense 0/0 row (quirk 3)
  const int rpi = F <= 64 ? 64 / F : 1;
  const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
  for (int fc = 0; fc < fchunks; ++fc) {
    const int fbase = fc * 64;
    const int fw = F <= 64 ? F : min(64, F - fbase);
    const int g = lane / fw, f = lane - g * fw;
    const bool lane_active = g < rpi;
    float acc = 0.0f;
    for (int blk = 0; blk < deg; blk += GGAD_WAVE) {
      const int idx = blk + lane;
      int k = 0; float w = 0.0f;
      if (idx < deg) {
        k = col[s + idx];
        w = inv_sr / sqrtf((float)cnt2[soff + k]);
      }
      gather_block(feat, stride, fbase, rpi, g, f, lane_active, k, w, min(GGAD_WAVE, deg - blk), acc);
    }
    float tot = (F <= 64) ? reduce_groups(acc, fw, rpi) : acc;
    if (deg == 0) tot = inv_sr * 0.0f;
    if (lane < fw) x2[(int64_t)e * F + fbase + lane] = tot;
  }
}

__global__ void __launch_bounds__(256) k_plan_reset(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const int32_t *__restrict__ ent_col, const int32_t *__restrict__ ent_slot,
                                                    const int32_t *__restrict__ ent_own, const int32_t *__restrict__ ent_total,
                                                    int64_t n_nodes, int32_t *__restrict__ cnt1, int32_t *__restrict__ cnt2,
                                                    int with_hop2) {
  const int e = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (e >= *ent_total) return;
  const int u = ent_col[e];
  const int64_t soff = (int64_t)ent_slot[e] * n_nodes;
  if (lane_id() == 0) cnt1[soff + u] = 0;
  if (!with_hop2 || ent_own[e] != e) return;
  const int s = rowptr[u], t = rowptr[u + 1];
  for (int i = s + lane_id(); i < t; i += GGAD_WAVE) cnt2[soff + col[i]] = 0;
}

}  // namespace

// ------------------------------------------------------------------ launchers used by ggad_mb_plan_build (plan_build.cpp)
int ggad_int_hop1(const ggad_mb_plan *P, const ggad_plan_view &V, int ldsw, int reset_now, hipStream_t st) {
  if (V.n_chunks == 0) return GGAD_OK;
  const unsigned eb = (unsigned)((V.n_chunks + 15) / 16);       // 16 pieces (4 waves x 4) per workgroup
  const int skip = P->xcd_skip >= 0 && P->xcd_skip < 8 ? P->xcd_skip : -1;
  k_expand<<<dim3(ggad_skip_grid(eb, skip)), dim3(256), 0, st>>>(P->rowptr, P->col, V.nodes, V.row_slot, V.ent_ptr, V.ck_rc, V.ck_e0, V.n_chunks,
                                            P->n_nodes, P->ent_col, P->ent_slot, P->ent_row, P->cnt1, P->own1, P->counters, skip);
  k_gather1c<<<dim3(ggad_skip_grid(eb, skip)), dim3(256), 0, st>>>(P->feat, P->feat_dim, P->feat_stride, V.row_slot, V.ent_ptr, V.row_ck_ptr, V.ck_rc,
                                              V.ck_e0, V.n_chunks, P->n_nodes, P->cnt1, P->own1, P->ent_col, P->ent_own, P->ent_c1,
                                              P->ck_part, P->ck_part_stride, P->x1, P->rowptr, P->own_deg, P->own_rp, P->pw_base,
                                              P->node_head, P->own_next, P->counters, ldsw, skip);
  const int row_blocks = (V.n_rows + 3) / 4;
  const int ent_blocks = reset_now ? (V.n_ents + 255) / 256 : 0;
  k_combine1_reset<<<dim3(ggad_skip_grid((unsigned)(row_blocks + ent_blocks), skip)), dim3(256), 0, st>>>(
      row_blocks, V.n_rows, V.row_ck_ptr, P->ck_part, P->ck_part_stride, P->feat_dim, P->x1, reset_now, V.n_ents, P->ent_col,
      P->ent_slot, P->n_nodes, P->cnt1, skip);
  GGAD_CHECK_LAUNCH("mb_plan_build (1-hop)");
  return GGAD_OK;
}

// device-atomic 2-hop stage + reset of both counter families (chunks the LDS-counting stage cannot take)
int ggad_int_global_hop2(const ggad_mb_plan *P, const ggad_plan_view &V, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  if (V.n_ents == 0) return GGAD_OK;
  const int32_t *tot = V.ent_ptr + V.n_rows;
  ggad_stream_t s = reinterpret_cast<ggad_stream_t>(st);
  int rc = ggad_mb_count2(P->rowptr, P->col, P->ent_col, P->ent_slot, tot, V.n_ents, P->n_nodes, P->own1, P->cnt2, s);
  if (rc) return rc;
  if (ev0) (void)hipEventRecord(ev0, st);
  rc = ggad_mb_gather2(P->rowptr, P->col, P->feat, P->feat_dim, P->feat_stride, P->ent_col, P->ent_slot, P->ent_own, tot, V.n_ents,
                       P->n_nodes, P->cnt2, P->x2, s);
  if (rc) return rc;
  if (ev1) (void)hipEventRecord(ev1, st);
  // zeroing whole slots streams 4 n bytes per batch; walking touches one 64-byte sector per counted 2-hop neighbour
  const double walk_bytes = (double)V.n_ents * (double)P->mean_nbr_deg * 64.0;
  const bool memset_cnt2 = walk_bytes > (double)V.n_batches * (double)P->n_nodes * 8.0;
  if (memset_cnt2) {
    const hipError_t me = hipMemsetAsync(P->cnt2, 0, (size_t)V.n_batches * (size_t)P->n_nodes * sizeof(int32_t), st);
    if (me != hipSuccess) { ggad_set_error(me, "mb_plan_build (memset cnt2)"); return GGAD_E_LAUNCH; }
  }
  return ggad_mb_plan_reset(P->rowptr, P->col, P->ent_col, P->ent_slot, P->ent_own, tot, V.n_ents, P->n_nodes, P->cnt1, P->cnt2,
                            memset_cnt2 ? 0 : 1, s);
}

// ------------------------------------------------------------------ C ABI
extern "C" {

int64_t ggad_scan_workspace_elems(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

int ggad_exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *workspace, ggad_stream_t stream) {
  GGAD_REQUIRE(in && out && workspace && n >= 0);
  const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  GGAD_REQUIRE(tiles <= SCAN_TILE);
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    (void)hipMemsetAsync(out, 0, sizeof(int32_t), st);
    GGAD_CHECK_LAUNCH("scan memset");
    return GGAD_OK;
  }
  scan_tile_sums<<<dim3((unsigned)tiles), dim3(SCAN_T), 0, st>>>(in, n, workspace);
  scan_tile_offsets<<<dim3(1), dim3(SCAN_T), 0, st>>>(workspace, (int)tiles, out + n);
  scan_apply<<<dim3((unsigned)tiles), dim3(SCAN_T), 0, st>>>(in, out, n, workspace);
  GGAD_CHECK_LAUNCH("exclusive_scan");
  return GGAD_OK;
}

int32_t ggad_mb_chunk_len(void) { return 16; }

int ggad_seg_mean(const float *feat, int32_t feat_dim, const int32_t *seg_ptr, const int32_t *seg_col, int32_t n_rows,
                  float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(feat && seg_ptr && seg_col && out && feat_dim >= 1 && feat_dim <= GGAD_MAX_F && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  k_seg_mean<<<dim3((n_rows + 3) / 4), dim3(256), 0, as_stream(stream)>>>(feat, feat_dim, seg_ptr, seg_col, nullptr, n_rows, out);
  GGAD_CHECK_LAUNCH("seg_mean");
  return GGAD_OK;
}

int ggad_seg_wsum(const float *feat, int32_t feat_dim, const int32_t *seg_ptr, const int32_t *seg_col, const float *seg_w,
                  int32_t n_rows, float *out, ggad_stream_t stream) {
  GGAD_REQUIRE(feat && seg_ptr && seg_col && seg_w && out && feat_dim >= 1 && feat_dim <= GGAD_MAX_F && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  k_seg_mean<<<dim3((n_rows + 3) / 4), dim3(256), 0, as_stream(stream)>>>(feat, feat_dim, seg_ptr, seg_col, seg_w, n_rows, out);
  GGAD_CHECK_LAUNCH("seg_wsum");
  return GGAD_OK;
}

int ggad_mb_count2(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                   const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes, const int32_t *own1,
                   int32_t *cnt2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && ent_col && ent_slot && ent_total && own1 && cnt2 && n_entries_cap >= 0);
  if (n_entries_cap == 0) return GGAD_OK;
  k_count2<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(rowptr, col, ent_col, ent_slot, ent_total,
                                                                                          n_nodes, own1, cnt2);
  GGAD_CHECK_LAUNCH("mb_count2");
  return GGAD_OK;
}

int ggad_mb_gather2(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                    const int32_t *ent_col, const int32_t *ent_slot, const int32_t *ent_own, const int32_t *ent_total,
                    int64_t n_entries_cap, int64_t n_nodes, const int32_t *cnt2, float *x2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && feat && ent_col && ent_slot && ent_own && ent_total && cnt2 && x2);
  GGAD_REQUIRE(feat_dim >= 1 && feat_dim <= GGAD_MAX_F && feat_stride >= feat_dim && n_entries_cap >= 0);
  if (n_entries_cap == 0) return GGAD_OK;
  k_gather2<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
      rowptr, col, feat, feat_dim, feat_stride, ent_col, ent_slot, ent_own, ent_total, n_nodes, cnt2, x2);
  GGAD_CHECK_LAUNCH("mb_gather2");
  return GGAD_OK;
}

int ggad_mb_plan_reset(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                       const int32_t *ent_own, const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes,
                       int32_t *cnt1, int32_t *cnt2, int32_t with_hop2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && ent_col && ent_slot && ent_own && ent_total && cnt1 && (!with_hop2 || cnt2));
  if (n_entries_cap == 0) return GGAD_OK;
  k_plan_reset<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
      rowptr, col, ent_col, ent_slot, ent_own, ent_total, n_nodes, cnt1, cnt2, with_hop2);
  GGAD_CHECK_LAUNCH("mb_plan_reset");
  return GGAD_OK;
}

}  // extern "C"
