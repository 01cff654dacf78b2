// Batch sub-graph plan + gather-aggregate kernels of the DGraph mini-batch path (gfx950).
//
// Replaces GCNAggregator.forward of the reference (src/graphsage.py:295-360): python set
// unions, a dense B x U and a dense U x U2 0/1 mask, their row/column sums, and two dense
// mask.mm(feature) products -- by CSR walks, per-batch integer histograms in HBM-resident
// counter slots and wave-level gathers of feature rows.
//
// Work decomposition (wave = 64 lanes):
//   row_degree     1 thread / batch row
//   expand1        1 wave   / batch row      (entries of N(i)+{i}, histogram c_j, owner election)
//   gather1        1 wave   / batch row      (1-hop aggregate, 4F+8 B per entry)
//   count2         1 wave   / entry          (histogram c'_k over N(u), owners only)
//   gather2        1 wave   / entry          (2-hop aggregate, owners only)  <- HBM-bound, dominant
//   plan_reset     1 wave   / entry
// Feature rows are F consecutive floats; a wave reads floor(64/F) rows per load instruction
// (3 rows of 17 floats: 51 active lanes, each instruction touches 3 x 68 contiguous bytes).
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

// ------------------------------------------------------------------ plan kernels
__global__ void __launch_bounds__(256) k_row_degree(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                    const int32_t *__restrict__ nodes, const int32_t *__restrict__ batch_ptr,
                                                    int n_batches, int n_rows, int32_t *__restrict__ row_r,
                                                    int32_t *__restrict__ row_slot) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int v = nodes[row];
  const int s = rowptr[v], e = rowptr[v + 1];
  const int p = lower_bound_i32(col, s, e, v);
  const int self_in = (p < e && col[p] == v) ? 1 : 0;
  row_r[row] = (e - s) + (1 - self_in);
  // slot = last g with batch_ptr[g] <= row
  int lo = 0, hi = n_batches;  // invariant: batch_ptr[lo] <= row < batch_ptr[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (batch_ptr[mid] <= row) lo = mid; else hi = mid;
  }
  row_slot[row] = lo;
}

__global__ void __launch_bounds__(256) k_expand1(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const int32_t *__restrict__ nodes, const int32_t *__restrict__ row_slot,
                                                 const int32_t *__restrict__ ent_ptr, int n_rows, int64_t n_nodes,
                                                 int32_t *__restrict__ ent_col, int32_t *__restrict__ ent_slot,
                                                 int32_t *__restrict__ ent_row, int32_t *__restrict__ cnt1,
                                                 int32_t *__restrict__ own1) {
  const int row = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (row >= n_rows) return;
  const int lane = lane_id();
  const int v = nodes[row];
  const int s = rowptr[v], e = rowptr[v + 1];
  const int p = lower_bound_i32(col, s, e, v);
  const bool self_in = (p < e && col[p] == v);
  const int pself = p - s;
  const int r = (e - s) + (self_in ? 0 : 1);
  const int base = ent_ptr[row];
  const int slot = row_slot[row];
  const int64_t soff = (int64_t)slot * n_nodes;
  for (int idx = lane; idx < r; idx += GGAD_WAVE) {
    int j;
    if (self_in || idx < pself) j = col[s + idx];
    else if (idx == pself) j = v;
    else j = col[s + idx - 1];
    ent_col[base + idx] = j;
    ent_slot[base + idx] = slot;
    ent_row[base + idx] = row;
    const int old = atomicAdd(&cnt1[soff + j], 1);
    if (old == 0) own1[soff + j] = base + idx;   // first arrival owns (batch, j)
  }
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

__global__ void __launch_bounds__(256) k_gather1(const float *__restrict__ feat, int F, int stride, const int32_t *__restrict__ row_slot,
                                                 const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ ent_col,
                                                 int n_rows, int64_t n_nodes, const int32_t *__restrict__ cnt1,
                                                 const int32_t *__restrict__ own1, int32_t *__restrict__ ent_own,
                                                 int32_t *__restrict__ ent_c1, float *__restrict__ x1) {
  const int row = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (row >= n_rows) return;
  const int lane = lane_id();
  const int e0 = ent_ptr[row], e1 = ent_ptr[row + 1];
  const int r = e1 - e0;
  const int64_t soff = (int64_t)row_slot[row] * n_nodes;
  const float inv_sr = 1.0f / sqrtf((float)r);           // mask.div(row_normalized)   graphsage.py:318
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
      if (idx < r) {
        j = ent_col[e0 + idx];
        const int c = cnt1[soff + j];
        if (fc == 0) { ent_own[e0 + idx] = own1[soff + j]; ent_c1[e0 + idx] = c; }
        w = inv_sr / sqrtf((float)c);                      // .div(col_normalized)
      }
      gather_block(feat, stride, fbase, rpi, g, f, lane_active, j, w, min(GGAD_WAVE, r - blk), acc);
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
                                                const int32_t *__restrict__ own1, int32_t *__restrict__ cnt2,
                                                float *__restrict__ featp, int stride, int F) {
  const int e = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (e >= *ent_total) return;
  const int u = ent_col[e];
  const int slot = ent_slot[e];
  const int64_t soff = (int64_t)slot * n_nodes;
  if (own1[soff + u] != e) return;                         // each distinct u of the batch once (set semantics)
  const int s = rowptr[u], t = rowptr[u + 1];
  if (cnt2 != nullptr) {
    for (int i = s + lane_id(); i < t; i += GGAD_WAVE) atomicAdd(&cnt2[soff + col[i]], 1);
  } else {   // packed layout: the counter of (slot, k) lives in k's feature row, word F + slot
    int32_t *base = reinterpret_cast<int32_t *>(featp) + F + slot;
    for (int i = s + lane_id(); i < t; i += GGAD_WAVE) atomicAdd(base + (int64_t)col[i] * stride, 1);
  }
}

// Packed layout (row = F features + per-slot int32 counters, 128-byte aligned): ONE random line per gathered
// neighbour delivers both x_k and c'_k.  Lanes per row = F + 1 (the extra lane fetches the counter word).
__global__ void __launch_bounds__(256) k_gather2_packed(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const float *__restrict__ featp, int F, int stride,
                                                        const int32_t *__restrict__ ent_col, const int32_t *__restrict__ ent_slot,
                                                        const int32_t *__restrict__ ent_own, const int32_t *__restrict__ ent_total,
                                                        float *__restrict__ x2) {
  const int e = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (e >= *ent_total) return;
  if (ent_own[e] != e) return;
  const int lane = lane_id();
  const int u = ent_col[e];
  const int slot = ent_slot[e];
  const int s = rowptr[u], t = rowptr[u + 1];
  const int deg = t - s;
  const float inv_sr = 1.0f / sqrtf((float)deg);
  const int lpr = F + 1, rpi = 64 / lpr;
  const int g = lane / lpr, f = lane - g * lpr;
  const bool lane_active = g < rpi;
  const int foff = (f < F) ? f : F + slot;                 // word inside the row this lane fetches
  float acc = 0.0f;
  for (int blk = 0; blk < deg; blk += GGAD_WAVE) {
    const int idx = blk + lane;
    const int k = (idx < deg) ? col[s + idx] : 0;
    const int count = min(GGAD_WAVE, deg - blk);
    const int iters = (count + rpi - 1) / rpi;
    int tt = 0;
    for (; tt + 4 <= iters; tt += 4) {
      float x[4]; bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int src = (tt + q) * rpi + g;
        const int kk = __shfl(k, src & 63, GGAD_WAVE);
        ok[q] = lane_active && src < count;
        x[q] = ok[q] ? featp[(int64_t)kk * stride + foff] : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float cw = __shfl(x[q], (g * lpr + F) & 63, GGAD_WAVE);        // counter word of this lane's group
        const float w = inv_sr / sqrtf((float)__float_as_int(cw));
        acc = ok[q] ? fmaf(w, x[q], acc) : acc;
      }
    }
    for (; tt < iters; ++tt) {
      const int src = tt * rpi + g;
      const int kk = __shfl(k, src & 63, GGAD_WAVE);
      const bool ok = lane_active && src < count;
      const float x = ok ? featp[(int64_t)kk * stride + foff] : 0.0f;
      const float cw = __shfl(x, (g * lpr + F) & 63, GGAD_WAVE);
      const float w = inv_sr / sqrtf((float)__float_as_int(cw));
      acc = ok ? fmaf(w, x, acc) : acc;
    }
  }
  // sum the rpi group partials of feature f into lanes [0, F)
  float tot = acc;
  for (int gg = 1; gg < rpi; ++gg) tot += __shfl(acc, (lane + gg * lpr) & 63, GGAD_WAVE);
  if (deg == 0) tot = inv_sr * 0.0f;
  if (lane < F) x2[(int64_t)e * F + lane] = tot;
}

__global__ void __launch_bounds__(256) k_reset_packed(float *__restrict__ featp, int64_t n_nodes, int stride, int F, int n_slots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes * n_slots) return;
  const int64_t node = i / n_slots;
  const int sl = (int)(i - node * n_slots);
  reinterpret_cast<int32_t *>(featp)[node * stride + F + sl] = 0;
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
                                                    int with_hop2, float *__restrict__ featp, int stride, int F) {
  const int e = blockIdx.x * (blockDim.x / GGAD_WAVE) + threadIdx.x / GGAD_WAVE;
  if (e >= *ent_total) return;
  const int u = ent_col[e];
  const int slot = ent_slot[e];
  const int64_t soff = (int64_t)slot * n_nodes;
  if (lane_id() == 0) cnt1[soff + u] = 0;
  if (!with_hop2 || ent_own[e] != e) return;
  const int s = rowptr[u], t = rowptr[u + 1];
  if (cnt2 != nullptr) {
    for (int i = s + lane_id(); i < t; i += GGAD_WAVE) cnt2[soff + col[i]] = 0;
  } else {
    int32_t *base = reinterpret_cast<int32_t *>(featp) + F + slot;
    for (int i = s + lane_id(); i < t; i += GGAD_WAVE) base[(int64_t)col[i] * stride] = 0;
  }
}

}  // namespace

// ------------------------------------------------------------------ C ABI
extern "C" {

int64_t ggad_scan_workspace_elems(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

int ggad_exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *workspace, ggad_stream_t stream) {
  GGAD_REQUIRE(in && out && workspace && n >= 0);
  const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  GGAD_REQUIRE(tiles <= SCAN_TILE);
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    hipMemsetAsync(out, 0, sizeof(int32_t), st);
    GGAD_CHECK_LAUNCH("scan memset");
    return GGAD_OK;
  }
  scan_tile_sums<<<dim3((unsigned)tiles), dim3(SCAN_T), 0, st>>>(in, n, workspace);
  scan_tile_offsets<<<dim3(1), dim3(SCAN_T), 0, st>>>(workspace, (int)tiles, out + n);
  scan_apply<<<dim3((unsigned)tiles), dim3(SCAN_T), 0, st>>>(in, out, n, workspace);
  GGAD_CHECK_LAUNCH("exclusive_scan");
  return GGAD_OK;
}

int ggad_mb_row_degree(const int32_t *rowptr, const int32_t *col, const int32_t *nodes, const int32_t *batch_ptr,
                       int32_t n_batches, int32_t n_rows, int32_t *row_r, int32_t *row_slot, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && nodes && batch_ptr && row_r && row_slot && n_batches > 0 && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  k_row_degree<<<dim3((n_rows + 255) / 256), dim3(256), 0, as_stream(stream)>>>(rowptr, col, nodes, batch_ptr, n_batches,
                                                                               n_rows, row_r, row_slot);
  GGAD_CHECK_LAUNCH("mb_row_degree");
  return GGAD_OK;
}

int ggad_mb_expand1(const int32_t *rowptr, const int32_t *col, const int32_t *nodes, const int32_t *row_slot,
                    const int32_t *ent_ptr, int32_t n_rows, int64_t n_nodes, int32_t *ent_col, int32_t *ent_slot,
                    int32_t *ent_row, int32_t *cnt1, int32_t *own1, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && nodes && row_slot && ent_ptr && ent_col && ent_slot && ent_row && cnt1 && own1 && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  k_expand1<<<dim3((n_rows + 3) / 4), dim3(256), 0, as_stream(stream)>>>(rowptr, col, nodes, row_slot, ent_ptr, n_rows,
                                                                         n_nodes, ent_col, ent_slot, ent_row, cnt1, own1);
  GGAD_CHECK_LAUNCH("mb_expand1");
  return GGAD_OK;
}

int ggad_mb_gather1(const float *feat, int32_t feat_dim, int32_t feat_stride, const int32_t *row_slot, const int32_t *ent_ptr,
                    const int32_t *ent_col, int32_t n_rows, int64_t n_nodes, const int32_t *cnt1,
                    const int32_t *own1, int32_t *ent_own, int32_t *ent_c1, float *x1, ggad_stream_t stream) {
  GGAD_REQUIRE(feat && row_slot && ent_ptr && ent_col && cnt1 && own1 && ent_own && ent_c1 && x1);
  GGAD_REQUIRE(feat_dim >= 1 && feat_dim <= GGAD_MAX_F && feat_stride >= feat_dim && n_rows >= 0);
  if (n_rows == 0) return GGAD_OK;
  k_gather1<<<dim3((n_rows + 3) / 4), dim3(256), 0, as_stream(stream)>>>(feat, feat_dim, feat_stride, row_slot, ent_ptr, ent_col, n_rows,
                                                                         n_nodes, cnt1, own1, ent_own, ent_c1, x1);
  GGAD_CHECK_LAUNCH("mb_gather1");
  return GGAD_OK;
}

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

// ---- row chunks: every batch row cut into pieces of <= chunk_len consecutive entries (the unit of work of the chunk-parallel
// forward kernel of step.hip: a hub row of thousands of entries is hundreds of independent pieces, not one workgroup's loop)
__global__ void __launch_bounds__(256) k_row_chunk_counts(const int32_t *__restrict__ ent_ptr, int n_rows, int chunk_len,
                                                          int32_t *__restrict__ nck) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_rows) nck[i] = (ent_ptr[i + 1] - ent_ptr[i] + chunk_len - 1) / chunk_len;
}
__global__ void __launch_bounds__(256) k_fill_chunks(const int32_t *__restrict__ ent_ptr, const int32_t *__restrict__ row_ck_ptr,
                                                     int n_rows, int chunk_len, int32_t *__restrict__ ck_rc,
                                                     int32_t *__restrict__ ck_e0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows) return;
  const int e0 = ent_ptr[i], e1 = ent_ptr[i + 1];
  int c = row_ck_ptr[i];
  for (int e = e0; e < e1; e += chunk_len, ++c) {
    ck_rc[c] = (i << 6) | min(chunk_len, e1 - e);          // row (chunk-relative to the plan) | entries of the piece
    ck_e0[c] = e;
  }
}

int32_t ggad_mb_chunk_len(void) { return 16; }

int ggad_mb_row_chunks(const int32_t *ent_ptr, int32_t n_rows, int32_t *nck_tmp, int32_t *row_ck_ptr, int32_t *ck_rc,
                       int32_t *ck_e0, int32_t *scan_ws, ggad_stream_t stream) {
  GGAD_REQUIRE(ent_ptr && nck_tmp && row_ck_ptr && ck_rc && ck_e0 && scan_ws && n_rows >= 0 && n_rows < (1 << 25));
  if (n_rows == 0) return GGAD_OK;
  const int cl = ggad_mb_chunk_len();
  k_row_chunk_counts<<<dim3((n_rows + 255) / 256), dim3(256), 0, as_stream(stream)>>>(ent_ptr, n_rows, cl, nck_tmp);
  GGAD_CHECK_LAUNCH("mb_row_chunks counts");
  const int rc = ggad_exclusive_scan_i32(nck_tmp, row_ck_ptr, n_rows, scan_ws, stream);
  if (rc) return rc;
  k_fill_chunks<<<dim3((n_rows + 255) / 256), dim3(256), 0, as_stream(stream)>>>(ent_ptr, row_ck_ptr, n_rows, cl, ck_rc, ck_e0);
  GGAD_CHECK_LAUNCH("mb_row_chunks fill");
  return GGAD_OK;
}

int ggad_mb_packed_stride(int32_t feat_dim) { return ((feat_dim + 1 + 31) / 32) * 32; }

int ggad_mb_count2(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                   const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes, const int32_t *own1,
                   int32_t *cnt2, float *feat_packed, int32_t feat_dim, int32_t feat_stride, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && ent_col && ent_slot && ent_total && own1 && n_entries_cap >= 0);
  GGAD_REQUIRE(cnt2 || (feat_packed && feat_stride > feat_dim && feat_dim >= 1));
  if (n_entries_cap == 0) return GGAD_OK;
  k_count2<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
      rowptr, col, ent_col, ent_slot, ent_total, n_nodes, own1, cnt2, feat_packed, feat_stride, feat_dim);
  GGAD_CHECK_LAUNCH("mb_count2");
  return GGAD_OK;
}

int ggad_mb_gather2(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                    const int32_t *ent_col, const int32_t *ent_slot, const int32_t *ent_own, const int32_t *ent_total,
                    int64_t n_entries_cap, int64_t n_nodes, const int32_t *cnt2, float *x2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && feat && ent_col && ent_slot && ent_own && ent_total && x2);
  GGAD_REQUIRE(feat_dim >= 1 && feat_dim <= GGAD_MAX_F && feat_stride >= feat_dim && n_entries_cap >= 0);
  GGAD_REQUIRE(cnt2 || (feat_stride > feat_dim && feat_dim + 1 <= 64));
  if (n_entries_cap == 0) return GGAD_OK;
  if (cnt2 == nullptr)
    k_gather2_packed<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        rowptr, col, feat, feat_dim, feat_stride, ent_col, ent_slot, ent_own, ent_total, x2);
  else
    k_gather2<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
        rowptr, col, feat, feat_dim, feat_stride, ent_col, ent_slot, ent_own, ent_total, n_nodes, cnt2, x2);
  GGAD_CHECK_LAUNCH("mb_gather2");
  return GGAD_OK;
}

int ggad_mb_reset_packed(float *feat_packed, int64_t n_nodes, int32_t feat_dim, int32_t feat_stride, int32_t n_slots,
                         ggad_stream_t stream) {
  GGAD_REQUIRE(feat_packed && n_nodes >= 0 && feat_dim >= 1 && n_slots >= 0 && feat_dim + n_slots <= feat_stride);
  const int64_t tot = n_nodes * n_slots;
  if (tot == 0) return GGAD_OK;
  k_reset_packed<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(feat_packed, n_nodes, feat_stride,
                                                                                           feat_dim, n_slots);
  GGAD_CHECK_LAUNCH("mb_reset_packed");
  return GGAD_OK;
}

int ggad_mb_plan_reset(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                       const int32_t *ent_own, const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes,
                       int32_t *cnt1, int32_t *cnt2, int32_t with_hop2, float *feat_packed, int32_t feat_dim,
                       int32_t feat_stride, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && ent_col && ent_slot && ent_own && ent_total && cnt1);
  GGAD_REQUIRE(!with_hop2 || cnt2 || (feat_packed && feat_stride > feat_dim));
  if (n_entries_cap == 0) return GGAD_OK;
  k_plan_reset<<<dim3((unsigned)((n_entries_cap + 3) / 4)), dim3(256), 0, as_stream(stream)>>>(
      rowptr, col, ent_col, ent_slot, ent_own, ent_total, n_nodes, cnt1, cnt2, with_hop2, feat_packed, feat_stride, feat_dim);
  GGAD_CHECK_LAUNCH("mb_plan_reset");
  return GGAD_OK;
}

}  // extern "C"
