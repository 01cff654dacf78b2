// LDS-tiled 2-hop aggregation of the DGraph mini-batch path (gfx950).
//
// Computes, for every owner entry u of a batch (the reference's `unique_nodes_list`, graphsage.py:306),
//     x2[u] = sum_{k in N(u)} feat[k] / (sqrt(|N(u)|) sqrt(c'_k)),     c'_k = #{owners u' of the batch : k in N(u')}
// i.e. mask_neigh.mm(embed_matrix_expand) with the batch-dependent column normalisation of graphsage.py:335-355.
//
// The straightforward formulation (plan.hip: k_count2 + k_gather2) makes three random DRAM accesses per
// (u, k) pair -- an atomic on a 14.8 MB per-batch counter array, a read of that counter, a read of the feature
// row -- and is bound by random 64-byte sector traffic (~3.7 TB/s of sectors = 1.8 TB/s algorithmic).
// Here the node id space is cut into TILES of 65,536 consecutive ids.  One workgroup owns one batch and
// walks the tiles in order; inside a tile
//   * c'_k lives in LDS (65,536 16-bit counters = 128 KB of the CU's 160 KB), incremented with ds_add;
//   * the neighbours of owner u that fall into the tile are a CONTIGUOUS piece of u's sorted CSR row,
//     found through a static per-node table of tile offsets (built once per graph);
//   * the feature rows touched are confined to one 4.4 MB slab of the table, which every batch-workgroup
//     reads at about the same time -> served by L2 / Infinity Cache instead of DRAM.
// No global atomics, no per-batch counter arrays in HBM, fixed summation order (tiles ascending, CSR order
// inside a tile) -> deterministic.  Requires < 65,536 owners per batch (16-bit counters); the host falls back
// to the global-counter path otherwise.
#include "common.h"

namespace {

constexpr int TILE_SHIFT = 16;
constexpr int TILE = 1 << TILE_SHIFT;
constexpr int H2_T = 1024;             // threads per workgroup (16 waves)
constexpr int H2_W = H2_T / 64;
constexpr int SHORT_MAX = 4;           // segments up to this length are processed one owner per lane

__global__ void __launch_bounds__(256) k_tile_offsets(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      int64_t n_nodes, int n_tiles, int shift, int32_t *__restrict__ off) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n_nodes) return;
  const int s = rowptr[u], e = rowptr[u + 1];
  int32_t *o = off + u * (n_tiles + 1);
  int t = 0;
  for (int i = s; i < e; ++i) {
    const int tile = col[i] >> shift;
    while (t <= tile) { o[t] = i - s; ++t; }
  }
  while (t <= n_tiles) { o[t] = e - s; ++t; }
}

__global__ void __launch_bounds__(256) k_owner_flags(const int32_t *__restrict__ ent_own, const int32_t *__restrict__ ent_total,
                                                     int64_t n_cap, int32_t *__restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cap) return;
  flags[e] = (e < *ent_total && ent_own[e] == (int32_t)e) ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_owner_compact(const int32_t *__restrict__ flags, const int32_t *__restrict__ pos,
                                                       int64_t n_cap, int32_t *__restrict__ own_list) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_cap) return;
  if (flags[e]) own_list[pos[e]] = (int32_t)e;
}

__device__ __forceinline__ void lds_count(uint32_t *cnt, int k) {
  const int loc = k & (TILE - 1);
  atomicAdd(&cnt[loc >> 1], 1u << ((loc & 1) << 4));
}
__device__ __forceinline__ float lds_weight(const uint32_t *cnt, int k, float inv_sr) {
  const int loc = k & (TILE - 1);
  const uint32_t c = (cnt[loc >> 1] >> ((loc & 1) << 4)) & 0xFFFFu;
  return inv_sr / sqrtf((float)c);                       // mask.div(row_normalized).div(col_normalized)  graphsage.py:348
}

template <int FT>
__global__ void __launch_bounds__(H2_T) k_hop2_tiled(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const float *__restrict__ feat, int F_rt, int stride,
                                                     const int32_t *__restrict__ tile_off, int n_tiles,
                                                     const int32_t *__restrict__ own_list, const int32_t *__restrict__ own_pos,
                                                     const int32_t *__restrict__ batch_ent_ptr,
                                                     const int32_t *__restrict__ ent_col, float *__restrict__ x2) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];     // TILE / 2 words
  const int F = (FT > 0) ? FT : F_rt;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int o0 = own_pos[batch_ent_ptr[b]], o1 = own_pos[batch_ent_ptr[b + 1]];
  const int n_own = o1 - o0;
  const int NT1 = n_tiles + 1;
  const int rpi = F <= 64 ? 64 / F : 1;
  const int g = lane / (F <= 64 ? F : 64), f = lane - g * (F <= 64 ? F : 64);
  const bool lane_active = g < rpi;
  for (int t = 0; t < n_tiles; ++t) {
    for (int i = threadIdx.x; i < TILE / 2; i += H2_T) cnt[i] = 0u;
    __syncthreads();
    // ---------------- count c'_k for the keys of this tile
    for (int base = wid * 64; base < n_own; base += H2_W * 64) {
      const int i = base + lane;
      const bool valid = i < n_own;
      int rp = 0, lo = 0, n = 0;
      if (valid) {
        const int u = ent_col[own_list[o0 + i]];
        rp = rowptr[u];
        lo = tile_off[(int64_t)u * NT1 + t];
        n = tile_off[(int64_t)u * NT1 + t + 1] - lo;
      }
      if (n > 0 && n <= SHORT_MAX) {
#pragma unroll
        for (int j = 0; j < SHORT_MAX; ++j)
          if (j < n) lds_count(cnt, col[rp + lo + j]);
      }
      unsigned long long mask = __ballot(n > SHORT_MAX);
      while (mask) {
        const int l = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int beg = __builtin_amdgcn_readlane(rp, l) + __builtin_amdgcn_readlane(lo, l);
        const int nl = __builtin_amdgcn_readlane(n, l);
        for (int idx = lane; idx < nl; idx += 64) lds_count(cnt, col[beg + idx]);
      }
    }
    __syncthreads();
    // ---------------- gather: x2[u] += sum over u's neighbours in this tile
    for (int base = wid * 64; base < n_own; base += H2_W * 64) {
      const int i = base + lane;
      const bool valid = i < n_own;
      int rp = 0, lo = 0, n = 0, e = 0;
      float inv_sr = 0.0f;
      if (valid) {
        e = own_list[o0 + i];
        const int u = ent_col[e];
        rp = rowptr[u];
        const int deg = rowptr[u + 1] - rp;
        inv_sr = 1.0f / sqrtf((float)deg);
        lo = tile_off[(int64_t)u * NT1 + t];
        n = tile_off[(int64_t)u * NT1 + t + 1] - lo;
      }
      if (n > 0 && n <= SHORT_MAX) {                      // one owner per lane: short segment, own accumulators
        float *dst = x2 + (int64_t)e * F;
        if constexpr (FT > 0) {
          float a[FT];
#pragma unroll
          for (int q = 0; q < FT; ++q) a[q] = 0.0f;
#pragma unroll
          for (int j = 0; j < SHORT_MAX; ++j) {
            if (j < n) {
              const int k = col[rp + lo + j];
              const float w = lds_weight(cnt, k, inv_sr);
              const float *x = feat + (int64_t)k * stride;
#pragma unroll
              for (int q = 0; q < FT; ++q) a[q] = fmaf(w, x[q], a[q]);
            }
          }
#pragma unroll
          for (int q = 0; q < FT; ++q) dst[q] += a[q];
        } else {
          for (int j = 0; j < n; ++j) {
            const int k = col[rp + lo + j];
            const float w = lds_weight(cnt, k, inv_sr);
            const float *x = feat + (int64_t)k * stride;
            for (int q = 0; q < F; ++q) dst[q] = fmaf(w, x[q], dst[q]);
          }
        }
      }
      unsigned long long mask = __ballot(n > SHORT_MAX);
      while (mask) {                                       // long segment: the whole wave works on one owner
        const int l = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int beg = __builtin_amdgcn_readlane(rp, l) + __builtin_amdgcn_readlane(lo, l);
        const int nl = __builtin_amdgcn_readlane(n, l);
        const int el = __builtin_amdgcn_readlane(e, l);
        const float isr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_sr), l));
        const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
        for (int fc = 0; fc < fchunks; ++fc) {
          const int fbase = fc * 64;
          const int fw = F <= 64 ? F : min(64, F - fbase);
          const int gg = F <= 64 ? g : lane / fw, ff = F <= 64 ? f : lane - (lane / fw) * fw;
          const bool act = F <= 64 ? lane_active : (lane / fw) < 1;
          float acc = 0.0f;
          for (int blk = 0; blk < nl; blk += 64) {
            const int idx = blk + lane;
            int k = 0; float w = 0.0f;
            if (idx < nl) { k = col[beg + idx]; w = lds_weight(cnt, k, isr); }
            const int count = min(64, nl - blk);
            const int iters = (count + rpi - 1) / rpi;
            for (int tt = 0; tt < iters; ++tt) {
              const int src = tt * rpi + gg;
              const int kk = __shfl(k, src & 63, GGAD_WAVE);
              const float ws = __shfl(w, src & 63, GGAD_WAVE);
              const float x = (act && src < count) ? feat[(int64_t)kk * stride + fbase + ff] : 0.0f;
              acc = fmaf((src < count) ? ws : 0.0f, x, acc);
            }
          }
          float tot = acc;
          if (F <= 64)
            for (int q = 1; q < rpi; ++q) tot += __shfl(acc, (lane + q * fw) & 63, GGAD_WAVE);
          if (lane < fw) x2[(int64_t)el * F + fbase + lane] += tot;
        }
      }
    }
    __syncthreads();
  }
  // isolated owners: the dense 0/0 row of the reference -> NaN (quirk 3)
  for (int i = threadIdx.x; i < n_own; i += H2_T) {
    const int e = own_list[o0 + i];
    const int u = ent_col[e];
    if (rowptr[u + 1] == rowptr[u]) {
      const float nanv = (1.0f / sqrtf(0.0f)) * 0.0f;
      for (int q = 0; q < F; ++q) x2[(int64_t)e * F + q] = nanv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// K-TILE-MAJOR variant with the counters in HBM slots (as plan.hip) but the WORK ordered by tile: launch t touches
// only counters [t*65536, (t+1)*65536) of every slot (256 KB per batch -> the whole working set of a launch, a few
// tens of MB, stays in L2 / Infinity Cache) and one 4.4 MB slab of the feature table.  One wave per owner entry;
// x2[e] accumulates over the launches in stream order (tiles ascending) -> deterministic.
__global__ void __launch_bounds__(256) k_count2_tile(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ tile_off, int n_tiles, int tile,
                                                     const int32_t *__restrict__ own_list, const int32_t *__restrict__ n_own,
                                                     const int32_t *__restrict__ ent_col, const int32_t *__restrict__ ent_slot,
                                                     int64_t n_nodes, int32_t *__restrict__ cnt2) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= *n_own) return;
  const int e = own_list[p];
  const int u = ent_col[e];
  const int64_t ob = (int64_t)u * (n_tiles + 1) + tile;
  const int lo = tile_off[ob], hi = tile_off[ob + 1];
  if (hi <= lo) return;
  const int64_t soff = (int64_t)ent_slot[e] * n_nodes;
  const int beg = rowptr[u];
  for (int i = lo + lane_id(); i < hi; i += GGAD_WAVE) atomicAdd(&cnt2[soff + col[beg + i]], 1);
}

__global__ void __launch_bounds__(256) k_gather2_tile(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      const float *__restrict__ feat, int F, int stride,
                                                      const int32_t *__restrict__ tile_off, int n_tiles, int tile,
                                                      const int32_t *__restrict__ own_list, const int32_t *__restrict__ n_own,
                                                      const int32_t *__restrict__ ent_col, const int32_t *__restrict__ ent_slot,
                                                      int64_t n_nodes, const int32_t *__restrict__ cnt2, float *__restrict__ x2) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= *n_own) return;
  const int lane = lane_id();
  const int e = own_list[p];
  const int u = ent_col[e];
  const int64_t ob = (int64_t)u * (n_tiles + 1) + tile;
  const int lo = tile_off[ob], hi = tile_off[ob + 1];
  const int beg = rowptr[u];
  const int deg = rowptr[u + 1] - beg;
  if (deg == 0 && tile == 0) {                          // isolated owner: NaN row (quirk 3)
    const float nanv = (1.0f / sqrtf(0.0f)) * 0.0f;
    for (int q = lane; q < F; q += 64) x2[(int64_t)e * F + q] = nanv;
    return;
  }
  if (hi <= lo) return;
  const int64_t soff = (int64_t)ent_slot[e] * n_nodes;
  const float inv_sr = 1.0f / sqrtf((float)deg);
  const int nl = hi - lo;
  const int rpi = F <= 64 ? 64 / F : 1;
  const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
  for (int fc = 0; fc < fchunks; ++fc) {
    const int fbase = fc * 64;
    const int fw = F <= 64 ? F : min(64, F - fbase);
    const int g = lane / fw, f = lane - g * fw;
    const bool act = g < rpi;
    float acc = 0.0f;
    for (int blk = 0; blk < nl; blk += 64) {
      const int idx = blk + lane;
      int k = 0; float w = 0.0f;
      if (idx < nl) { k = col[beg + lo + idx]; w = inv_sr / sqrtf((float)cnt2[soff + k]); }
      const int count = min(64, nl - blk);
      const int iters = (count + rpi - 1) / rpi;
      int tt = 0;
      for (; tt + 4 <= iters; tt += 4) {
        float x[4], ww[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int src = (tt + q) * rpi + g;
          const int kk = __shfl(k, src & 63, GGAD_WAVE);
          const float ws = __shfl(w, src & 63, GGAD_WAVE);
          ww[q] = (src < count) ? ws : 0.0f;
          x[q] = (act && src < count) ? feat[(int64_t)kk * stride + fbase + f] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fmaf(ww[q], x[q], acc);
      }
      for (; tt < iters; ++tt) {
        const int src = tt * rpi + g;
        const int kk = __shfl(k, src & 63, GGAD_WAVE);
        const float ws = __shfl(w, src & 63, GGAD_WAVE);
        const float x = (act && src < count) ? feat[(int64_t)kk * stride + fbase + f] : 0.0f;
        acc = fmaf((src < count) ? ws : 0.0f, x, acc);
      }
    }
    float tot = acc;
    if (F <= 64)
      for (int q = 1; q < rpi; ++q) tot += __shfl(acc, (lane + q * fw) & 63, GGAD_WAVE);
    if (lane < fw) x2[(int64_t)e * F + fbase + lane] += tot;
  }
}

// ---------------------------------------------------------------------------------------------------------
// "ldsw": LDS counting, weights through HBM.  Measured facts behind the split (scripts/atomic_bench.hip):
// device-scope atomics run at ~27 G/s whatever the footprint or clustering, so one global atomic per (u, k) pair
// cannot beat 10 ms per 277 M pairs; random 4-byte reads run at 50-60 G/s, sequential ones are free.
//   k_tile_counts : one workgroup per (tile of 32,768 ids, batch).  All pairs of the batch whose k lies in the tile
//                   are enumerated PAIR-parallel (block scan of the segment lengths + binary search in LDS, so hub
//                   owners and 1-neighbour owners cost the same per pair), counted with ds_add into 16-bit LDS
//                   counters, and the final count of every pair is written to pc[] at the pair's position inside
//                   its owner's CSR row.  No global atomics, no counter arrays in HBM.
//   k_gather2_w   : one wave per owner, exactly k_gather2's access pattern but the weight comes from the
//                   SEQUENTIAL pc[] stream instead of a random counter read: one random row per neighbour.
constexpr int TW_SHIFT = 15;
constexpr int TW_TILE = 1 << TW_SHIFT;
constexpr int TW_MAXOWN = 6144;          // owners per batch the LDS arrays hold (falls back to "global" beyond)
constexpr int TW_T = 1024;

__global__ void __launch_bounds__(256) k_owner_meta(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ own_list,
                                                    const int32_t *__restrict__ n_own, const int32_t *__restrict__ ent_col,
                                                    int64_t n_cap, int32_t *__restrict__ own_deg, int32_t *__restrict__ own_rp) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_cap) return;
  if (p >= *n_own) { own_deg[p] = 0; own_rp[p] = 0; return; }
  const int u = ent_col[own_list[p]];
  const int rp = rowptr[u];
  own_rp[p] = rp;
  own_deg[p] = rowptr[u + 1] - rp;
}

// seg_t[t][p] = tile_off[node(p)][t]: the per-node table rows of the chunk's owners, transposed to TILE-major so that the
// workgroup of (tile t, batch b) reads its owners' segment bounds as two contiguous runs.  (Reading tile_off directly
// costs one random 64-byte sector per (owner, tile): 68 M of them per chunk, 2/3 of k_tile_counts' time.)
// One workgroup = 64 owners: rows read coalesced (one wave per owner row), transposed through LDS, written as 256-byte runs.
constexpr int TT_SLAB = 128;
__global__ void __launch_bounds__(256) k_seg_transpose(const int32_t *__restrict__ tile_off, int n_tiles,
                                                       const int32_t *__restrict__ own_list, const int32_t *__restrict__ n_own,
                                                       const int32_t *__restrict__ ent_col, int64_t n_cap,
                                                       int32_t *__restrict__ seg_t) {
  __shared__ int tl[TT_SLAB][65];
  const int p0 = blockIdx.x * 64;
  const int no = *n_own;
  if (p0 >= no) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const int NT1 = n_tiles + 1;
  for (int t0 = 0; t0 < NT1; t0 += TT_SLAB) {
    for (int j = wid; j < 64; j += 4) {
      const int p = p0 + j;
      if (p < no) {
        const int u = ent_col[own_list[p]];
        const int32_t *row = tile_off + (int64_t)u * NT1 + t0;
        if (t0 + lane < NT1) tl[lane][j] = row[lane];
        if (t0 + 64 + lane < NT1) tl[64 + lane][j] = row[64 + lane];
      }
    }
    __syncthreads();
    const int nt = min(TT_SLAB, NT1 - t0);
    for (int idx = threadIdx.x; idx < nt * 64; idx += 256) {
      const int tt = idx >> 6, j = idx & 63;
      if (p0 + j < no) seg_t[(int64_t)(t0 + tt) * n_cap + p0 + j] = tl[tt][j];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int block_excl_scan_1024(int v, int *warp_buf, int *total) {
  // exclusive scan over 1024 threads (16 waves); warp_buf: 16 ints of LDS
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) warp_buf[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { const int sv = warp_buf[w]; if (w < wid) base += sv; tot += sv; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ void __launch_bounds__(TW_T) k_tile_counts(const int32_t *__restrict__ col, const int32_t *__restrict__ seg_t,
                                                      int64_t n_cap, const int32_t *__restrict__ own_rp,
                                                      const int32_t *__restrict__ own_pos,
                                                      const int32_t *__restrict__ batch_ent_ptr,
                                                      const int32_t *__restrict__ pw_base, uint16_t *__restrict__ pc) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  uint32_t *cnt = lds_u;                                   // TW_TILE / 2 words
  int *offs = reinterpret_cast<int *>(lds_u + TW_TILE / 2);  // TW_MAXOWN + 1
  int *segbeg = offs + TW_MAXOWN + 1;                      // TW_MAXOWN   (index into col[])
  int *dst = segbeg + TW_MAXOWN;                           // TW_MAXOWN   (index into pc[])
  __shared__ int wbuf[16];
  const int t = blockIdx.x, b = blockIdx.y;
  const int o0 = own_pos[batch_ent_ptr[b]], o1 = own_pos[batch_ent_ptr[b + 1]];
  const int n_own_all = o1 - o0;
  const int32_t *seg_lo = seg_t + (int64_t)t * n_cap, *seg_hi = seg_lo + n_cap;
  for (int i = threadIdx.x; i < TW_TILE / 2; i += TW_T) cnt[i] = 0u;
  constexpr int IPT = TW_MAXOWN / TW_T;
  // batches with more owners than the LDS tables hold are walked in slabs of TW_MAXOWN owners (counts accumulate
  // over the slabs in pass 0; pass 1 re-derives each slab's tables).  The usual batch is one slab: tables built once.
  const int n_slabs = (n_own_all + TW_MAXOWN - 1) / TW_MAXOWN;
  const bool one_slab = n_slabs == 1;
  constexpr int CACHE_IT = 16;
  int kc[CACHE_IT], dc[CACHE_IT];
  for (int pass = 0; pass < 2; ++pass) {
    for (int slab = 0; slab < n_slabs; ++slab) {
      const int ob0 = o0 + slab * TW_MAXOWN;
      const int n_own = min(TW_MAXOWN, o1 - ob0);
      if (pass == 0 || n_slabs > 1) {
        // segment of every owner inside this tile; exclusive scan of the lengths (IPT owners per thread).
        // The four table reads are coalesced (thread-contiguous owners), staged through LDS for the per-thread scan.
        for (int i = threadIdx.x; i < n_own; i += TW_T) {
          const int lo = seg_lo[ob0 + i], hi = seg_hi[ob0 + i];
          offs[i] = hi - lo;
          segbeg[i] = own_rp[ob0 + i] + lo;
          dst[i] = pw_base[ob0 + i] + lo;
        }
        __syncthreads();
        int len[IPT];
        int mysum = 0;
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
          const int i = threadIdx.x * IPT + q;
          len[q] = (i < n_own) ? offs[i] : 0;
          mysum += len[q];
        }
        __syncthreads();
        int P0;
        int ex = block_excl_scan_1024(mysum, wbuf, &P0);
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
          const int i = threadIdx.x * IPT + q;
          if (i < n_own) offs[i] = ex;
          ex += len[q];
        }
        if (threadIdx.x == 0) offs[n_own] = P0;               // number of pairs of the slab
        __syncthreads();
      }
      const int P = offs[n_own];
      if (P > 0) {                                             // uniform
        // pair p -> owner = last index i with offs[i] <= p.  Branch-free descent over the LDS table, four pairs per
        // thread in flight (the LDS round trips of one search are dependent; four searches interleave), so a hub
        // owner and a 1-neighbour owner cost the same per pair.
        const int s0 = 1 << (31 - __clz(n_own));
        auto locate = [&](int pp, int &own, int &j) {
          int lo = 0;
          for (int st = s0; st > 0; st >>= 1) { const int m = lo + st; lo = (offs[min(m, n_own)] <= pp) ? m : lo; }
          own = lo; j = pp - offs[lo];
        };
        auto locate4 = [&](const int (&pp)[4], int (&own)[4], int (&j)[4]) {
          int lo[4] = {0, 0, 0, 0};
          for (int st = s0; st > 0; st >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int m = lo[q] + st; lo[q] = (offs[min(m, n_own)] <= pp[q]) ? m : lo[q]; }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) { own[q] = lo[q]; j[q] = pp[q] - offs[lo[q]]; }
        };
        int p_start = 0;
        if (one_slab) {
          // the first CACHE_IT * 1024 pairs keep (k, pc index) in registers between the passes: pass 1 is then a
          // counter read + a 2-byte store
          if (pass == 0) {
#pragma unroll
            for (int g4 = 0; g4 < CACHE_IT; g4 += 4) {
              int pp[4], own[4], j[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) pp[q] = min((g4 + q) * TW_T + (int)threadIdx.x, P - 1);
              if (g4 * TW_T < P) {                             // uniform
                locate4(pp, own, j);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const bool on = (g4 + q) * TW_T + (int)threadIdx.x < P;
                  const int k = col[segbeg[own[q]] + j[q]];
                  const int loc = k & (TW_TILE - 1);
                  kc[g4 + q] = on ? loc : -1;
                  dc[g4 + q] = dst[own[q]] + j[q];
                  if (on) atomicAdd(&cnt[loc >> 1], 1u << ((loc & 1) << 4));
                }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) kc[g4 + q] = -1;
              }
            }
          } else {
#pragma unroll
            for (int it = 0; it < CACHE_IT; ++it)
              if (kc[it] >= 0) pc[dc[it]] = (uint16_t)((cnt[kc[it] >> 1] >> ((kc[it] & 1) << 4)) & 0xFFFFu);
          }
          p_start = CACHE_IT * TW_T;
        }
        for (int p0 = p_start; p0 < P; p0 += 4 * TW_T) {
          int pp[4], own[4], j[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pp[q] = min(p0 + q * TW_T + (int)threadIdx.x, P - 1);
          locate4(pp, own, j);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (p0 + q * TW_T + (int)threadIdx.x < P) {
              const int k = col[segbeg[own[q]] + j[q]];
              const int loc = k & (TW_TILE - 1);
              if (pass == 0) atomicAdd(&cnt[loc >> 1], 1u << ((loc & 1) << 4));
              else pc[(int64_t)dst[own[q]] + j[q]] = (uint16_t)((cnt[loc >> 1] >> ((loc & 1) << 4)) & 0xFFFFu);
            }
          }
        }
        (void)locate;
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256) k_gather2_w(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const float *__restrict__ feat, int F, int stride,
                                                   const int32_t *__restrict__ own_list, const int32_t *__restrict__ n_own,
                                                   const int32_t *__restrict__ ent_col, const int32_t *__restrict__ pw_base,
                                                   const uint16_t *__restrict__ pc, float *__restrict__ x2) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= *n_own) return;
  const int lane = lane_id();
  const int e = own_list[p];
  const int u = ent_col[e];
  const int s = rowptr[u];
  const int deg = rowptr[u + 1] - s;
  const int64_t pb = pw_base[p];
  const float inv_sr = 1.0f / sqrtf((float)deg);          // deg = 0 -> inf * 0 = NaN, as the dense 0/0 row (quirk 3)
  const int rpi = F <= 64 ? 64 / F : 1;
  const int fchunks = F <= 64 ? 1 : (F + 63) / 64;
  for (int fc = 0; fc < fchunks; ++fc) {
    const int fbase = fc * 64;
    const int fw = F <= 64 ? F : min(64, F - fbase);
    const int g = lane / fw, f = lane - g * fw;
    const bool act = g < rpi;
    float acc = 0.0f;
    for (int blk = 0; blk < deg; blk += 64) {
      const int idx = blk + lane;
      int k = 0; float w = 0.0f;
      if (idx < deg) { k = col[s + idx]; w = inv_sr / sqrtf((float)pc[pb + idx]); }   // .div(row).div(col)  graphsage.py:348
      const int count = min(64, deg - blk);
      const int iters = (count + rpi - 1) / rpi;
      int tt = 0;
      for (; tt + 4 <= iters; tt += 4) {
        float x[4], ww[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int src = (tt + q) * rpi + g;
          const int kk = __shfl(k, src & 63, GGAD_WAVE);
          const float ws = __shfl(w, src & 63, GGAD_WAVE);
          ww[q] = (src < count) ? ws : 0.0f;
          x[q] = (act && src < count) ? feat[(int64_t)kk * stride + fbase + f] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fmaf(ww[q], x[q], acc);
      }
      for (; tt < iters; ++tt) {
        const int src = tt * rpi + g;
        const int kk = __shfl(k, src & 63, GGAD_WAVE);
        const float ws = __shfl(w, src & 63, GGAD_WAVE);
        const float x = (act && src < count) ? feat[(int64_t)kk * stride + fbase + f] : 0.0f;
        acc = fmaf((src < count) ? ws : 0.0f, x, acc);
      }
    }
    float tot = acc;
    if (F <= 64)
      for (int q = 1; q < rpi; ++q) tot += __shfl(acc, (lane + q * fw) & 63, GGAD_WAVE);
    if (deg == 0) tot = inv_sr * 0.0f;
    if (lane < fw) x2[(int64_t)e * F + fbase + lane] = tot;
  }
}

// ---- node-major gather: all occurrences of a node in the chunk share ONE pass over its neighbour rows.
// A node u is an owner in ~ B * deg(u) / N of the batches: the 2,000-neighbour hubs that produce most of the 2-hop
// pairs recur in ~16 of 150 batches, each time with other weights 1/sqrt(r c') but the SAME neighbour rows.  The owner
// entries of the chunk are linked per node (atomicExch on node_head[u]: order of the list is irrelevant, every
// occurrence has its own accumulator and its own fixed summation order), the list is cut into groups of up to 8
// occurrences and one wave per group walks the node's neighbours: one random 128-byte row fetch per neighbour serves
// up to 8 (batch, owner) pairs, their counts streaming in from pc[]; the groups of one node run side by side and find
// each other's rows in L2.  Results are bit-identical to k_gather2_w.
__global__ void __launch_bounds__(256) k_link_owners(const int32_t *__restrict__ own_list, const int32_t *__restrict__ n_own,
                                                     const int32_t *__restrict__ ent_col, int32_t *__restrict__ node_head,
                                                     int32_t *__restrict__ own_next) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= *n_own) return;
  const int u = ent_col[own_list[p]];
  own_next[p] = atomicExch(&node_head[u], p + 1);
}

template <int G>
__device__ __forceinline__ void gather_occurrences(const int32_t *__restrict__ col, const float *__restrict__ feat, int F,
                                                   int stride, int s, int deg, float inv_sr, const int (&e)[8],
                                                   const int64_t (&pb)[8], const uint16_t *__restrict__ pc,
                                                   float *__restrict__ x2, int lane) {
  const int rpi = 64 / F;
  const int g = lane / F, f = lane - g * F;
  const bool act = g < rpi;
  float acc[G];
#pragma unroll
  for (int j = 0; j < G; ++j) acc[j] = 0.0f;
  for (int blk = 0; blk < deg; blk += 64) {
    const int idx = blk + lane;
    int k = 0;
    float w[G];
#pragma unroll
    for (int j = 0; j < G; ++j) w[j] = 0.0f;
    if (idx < deg) {
      k = col[s + idx];
#pragma unroll
      for (int j = 0; j < G; ++j)
        if (e[j] >= 0) w[j] = inv_sr / sqrtf((float)pc[pb[j] + idx]);            // .div(row).div(col)  graphsage.py:348
    }
    const int count = min(64, deg - blk);
    const int iters = (count + rpi - 1) / rpi;
    int tt = 0;
    for (; tt + 4 <= iters; tt += 4) {
      float x[4];
      int src[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        src[q] = (tt + q) * rpi + g;
        const int kk = __shfl(k, src[q] & 63, GGAD_WAVE);
        x[q] = (act && src[q] < count) ? feat[(int64_t)kk * stride + f] : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const float ws = __shfl(w[j], src[q] & 63, GGAD_WAVE);
          acc[j] = fmaf((src[q] < count) ? ws : 0.0f, x[q], acc[j]);
        }
      }
    }
    for (; tt < iters; ++tt) {
      const int sr = tt * rpi + g;
      const int kk = __shfl(k, sr & 63, GGAD_WAVE);
      const float x = (act && sr < count) ? feat[(int64_t)kk * stride + f] : 0.0f;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const float ws = __shfl(w[j], sr & 63, GGAD_WAVE);
        acc[j] = fmaf((sr < count) ? ws : 0.0f, x, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < G; ++j) {
    float tot = acc[j];
    for (int q = 1; q < rpi; ++q) tot += __shfl(acc[j], (lane + q * F) & 63, GGAD_WAVE);
    if (deg == 0) tot = inv_sr * 0.0f;                       // inf * 0 = NaN, as the dense 0/0 row (quirk 3)
    if (lane < F && e[j] >= 0) x2[(int64_t)e[j] * F + lane] = tot;
  }
}

// One thread per owner entry; the entry that was linked last acts for its node: it cuts the node's occurrence list into
// groups of <= 8 and appends them to the compact group table (all groups of a node adjacent, so that their waves are
// dispatched together and share the node's neighbour rows in L2), then clears node_head[u] for the next chunk.
__global__ void __launch_bounds__(256) k_build_groups(const int32_t *__restrict__ own_list, const int32_t *__restrict__ n_own,
                                                      const int32_t *__restrict__ ent_col, int32_t *__restrict__ node_head,
                                                      const int32_t *__restrict__ own_next, int32_t *__restrict__ grp_p,
                                                      int32_t *__restrict__ n_groups) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= *n_own) return;
  const int u = ent_col[own_list[p]];
  if (node_head[u] != p + 1) return;
  int m = 0;
  for (int cur = p + 1; cur != 0; cur = own_next[cur - 1]) ++m;
  const int ng = (m + 7) >> 3;
  const int base = atomicAdd(n_groups, ng);
  int cur = p + 1;
  for (int g = 0; g < ng; ++g) {
    int32_t *dst = grp_p + (int64_t)(base + g) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int v = -1;
      if (cur != 0) { v = cur - 1; cur = own_next[v]; }
      dst[j] = v;
    }
  }
  node_head[u] = 0;
}

__global__ void __launch_bounds__(256) k_gather2_groups(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const float *__restrict__ feat, int F, int stride,
                                                        const int32_t *__restrict__ own_list, const int32_t *__restrict__ ent_col,
                                                        const int32_t *__restrict__ pw_base, const uint16_t *__restrict__ pc,
                                                        const int32_t *__restrict__ grp_p, const int32_t *__restrict__ n_groups,
                                                        float *__restrict__ x2) {
  const int gi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gi >= *n_groups) return;
  const int lane = lane_id();
  const int pv = (lane < 8) ? grp_p[(int64_t)gi * 8 + lane] : -1;
  int e[8];
  int64_t pb[8];
  int n = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int pj = __builtin_amdgcn_readlane(pv, j);
    e[j] = -1; pb[j] = 0;
    if (pj >= 0) { e[j] = own_list[pj]; pb[j] = pw_base[pj]; ++n; }
  }
  const int u = ent_col[e[0]];
  const int s = rowptr[u];
  const int deg = rowptr[u + 1] - s;
  const float inv_sr = 1.0f / sqrtf((float)deg);
  if (n == 1) gather_occurrences<1>(col, feat, F, stride, s, deg, inv_sr, e, pb, pc, x2, lane);
  else if (n == 2) gather_occurrences<2>(col, feat, F, stride, s, deg, inv_sr, e, pb, pc, x2, lane);
  else if (n <= 4) gather_occurrences<4>(col, feat, F, stride, s, deg, inv_sr, e, pb, pc, x2, lane);
  else gather_occurrences<8>(col, feat, F, stride, s, deg, inv_sr, e, pb, pc, x2, lane);
}

}  // namespace

extern "C" {

int ggad_mb_tile_size(void) { return TILE; }
int ggad_mb_ldsw_tile_shift(void) { return TW_SHIFT; }
int ggad_mb_ldsw_max_owners(void) { return TW_MAXOWN; }
int64_t ggad_mb_tile_offsets_elems(int64_t n_nodes, int32_t tile_shift) {
  return n_nodes * (((n_nodes + (1LL << tile_shift) - 1) >> tile_shift) + 1);
}

int ggad_mb_tile_offsets(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t tile_shift, int32_t *tile_off,
                         ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && tile_off && n_nodes >= 0 && tile_shift >= 8 && tile_shift <= 24);
  if (n_nodes == 0) return GGAD_OK;
  const int n_tiles = (int)((n_nodes + (1LL << tile_shift) - 1) >> tile_shift);
  k_tile_offsets<<<dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(rowptr, col, n_nodes, n_tiles,
                                                                                               tile_shift, tile_off);
  GGAD_CHECK_LAUNCH("mb_tile_offsets");
  return GGAD_OK;
}

/* flags[e] = entry e is an owner; caller scans flags -> own_pos (exclusive, n_cap + 1 values) */
int ggad_mb_owner_flags(const int32_t *ent_own, const int32_t *ent_total, int64_t n_entries_cap, int32_t *flags,
                        ggad_stream_t stream) {
  GGAD_REQUIRE(ent_own && ent_total && flags && n_entries_cap >= 0);
  if (n_entries_cap == 0) return GGAD_OK;
  k_owner_flags<<<dim3((unsigned)((n_entries_cap + 255) / 256)), dim3(256), 0, as_stream(stream)>>>(ent_own, ent_total,
                                                                                                   n_entries_cap, flags);
  GGAD_CHECK_LAUNCH("mb_owner_flags");
  return GGAD_OK;
}

int ggad_mb_hop2_tiled(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                       int64_t n_nodes, const int32_t *tile_off, const int32_t *flags, const int32_t *own_pos,
                       int32_t *own_list, const int32_t *batch_ent_ptr, int32_t n_batches, const int32_t *ent_col,
                       int64_t n_entries_cap, float *x2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && feat && tile_off && flags && own_pos && own_list && batch_ent_ptr && ent_col && x2);
  GGAD_REQUIRE(feat_dim >= 1 && feat_dim <= GGAD_MAX_F && feat_stride >= feat_dim && n_batches >= 0 && n_entries_cap >= 0);
  if (n_batches == 0 || n_entries_cap == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  const int n_tiles = (int)((n_nodes + TILE - 1) >> TILE_SHIFT);
  k_owner_compact<<<dim3((unsigned)((n_entries_cap + 255) / 256)), dim3(256), 0, st>>>(flags, own_pos, n_entries_cap, own_list);
  const size_t lds = (size_t)TILE / 2 * sizeof(uint32_t);
  if (feat_dim == 17) {
    static bool attr17 = false;
    if (!attr17) { hipFuncSetAttribute((const void *)k_hop2_tiled<17>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr17 = true; }
    k_hop2_tiled<17><<<dim3(n_batches), dim3(H2_T), lds, st>>>(rowptr, col, feat, feat_dim, feat_stride, tile_off, n_tiles, own_list,
                                                              own_pos, batch_ent_ptr, ent_col, x2);
  } else {
    static bool attr0 = false;
    if (!attr0) { hipFuncSetAttribute((const void *)k_hop2_tiled<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr0 = true; }
    k_hop2_tiled<0><<<dim3(n_batches), dim3(H2_T), lds, st>>>(rowptr, col, feat, feat_dim, feat_stride, tile_off, n_tiles, own_list,
                                                             own_pos, batch_ent_ptr, ent_col, x2);
  }
  GGAD_CHECK_LAUNCH("mb_hop2_tiled");
  return GGAD_OK;
}

/* "ldsw" 2-hop, stage 1: compact the owners, pw_base = exclusive scan of their degrees, then LDS counting per
 * (32,768-id tile, batch) writing the per-pair counts pc[] (uint16, one per 2-hop pair, laid out like the owners' CSR rows).
 * tile_off must be built with shift ggad_mb_ldsw_tile_shift().  own_deg, own_rp: n_entries_cap ints; pw_base:
 * n_entries_cap + 1; scan_ws: ggad_scan_workspace_elems(n_entries_cap); seg_t: ggad_mb_ldsw_seg_elems(n_nodes,
 * n_entries_cap) ints (the owners' tile_off rows, transposed tile-major); pc: >= pw_base[n_entries_cap] elements. */
int64_t ggad_mb_ldsw_seg_elems(int64_t n_nodes, int64_t n_entries_cap) {
  return (((n_nodes + TW_TILE - 1) >> TW_SHIFT) + 1) * n_entries_cap;
}

int ggad_mb_hop2_ldsw_count(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, const int32_t *tile_off,
                            const int32_t *flags, const int32_t *own_pos, int32_t *own_list, const int32_t *batch_ent_ptr,
                            int32_t n_batches, const int32_t *ent_col, int64_t n_entries_cap, int32_t *own_deg,
                            int32_t *own_rp, int32_t *pw_base, int32_t *scan_ws, int32_t *seg_t, uint16_t *pc,
                            ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && tile_off && flags && own_pos && own_list && batch_ent_ptr && ent_col && own_deg && own_rp &&
               pw_base && scan_ws && seg_t && pc);
  GGAD_REQUIRE(n_batches >= 0 && n_entries_cap >= 0 && n_nodes >= 0);
  if (n_batches == 0 || n_entries_cap == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  const int n_tiles = (int)((n_nodes + TW_TILE - 1) >> TW_SHIFT);
  const unsigned eb = (unsigned)((n_entries_cap + 255) / 256);
  k_owner_compact<<<dim3(eb), dim3(256), 0, st>>>(flags, own_pos, n_entries_cap, own_list);
  const int32_t *n_own = own_pos + n_entries_cap;
  k_owner_meta<<<dim3(eb), dim3(256), 0, st>>>(rowptr, own_list, n_own, ent_col, n_entries_cap, own_deg, own_rp);
  int rc = ggad_exclusive_scan_i32(own_deg, pw_base, n_entries_cap, scan_ws, stream);
  if (rc) return rc;
  k_seg_transpose<<<dim3((unsigned)((n_entries_cap + 63) / 64)), dim3(256), 0, st>>>(tile_off, n_tiles, own_list, n_own, ent_col,
                                                                                   n_entries_cap, seg_t);
  const size_t lds = (size_t)(TW_TILE / 2) * 4 + (size_t)(3 * TW_MAXOWN + 1) * 4;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void *)k_tile_counts, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  k_tile_counts<<<dim3(n_tiles, n_batches), dim3(TW_T), lds, st>>>(col, seg_t, n_entries_cap, own_rp, own_pos, batch_ent_ptr,
                                                                   pw_base, pc);
  GGAD_CHECK_LAUNCH("mb_hop2_ldsw_count");
  return GGAD_OK;
}

/* "ldsw" 2-hop, stage 2: x2[owner] = (1/sqrt(deg)) * sum_k x_k / sqrt(pc[..]), streaming col[] and pc[], one random
 * feature row per neighbour.  With node_head (int32[n_nodes], zero on entry, zero again on return), own_next
 * (int32[n_entries_cap]), grp (int32[8 * n_entries_cap + 1]: group table, last word = group counter) and feat_dim <= 64:
 * node-major -- the occurrences of a node in the chunk share the fetch of its neighbour rows, 8 per wave.
 * node_head == NULL: one wave per owner entry.  Same results either way. */
int ggad_mb_hop2_ldsw_gather(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                             const int32_t *own_pos, const int32_t *own_list, const int32_t *ent_col, int64_t n_entries_cap,
                             const int32_t *pw_base, const uint16_t *pc, int32_t *node_head, int32_t *own_next, int32_t *grp,
                             float *x2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && feat && own_pos && own_list && ent_col && pw_base && pc && x2);
  GGAD_REQUIRE(feat_dim >= 1 && feat_dim <= GGAD_MAX_F && feat_stride >= feat_dim && n_entries_cap >= 0);
  GGAD_REQUIRE((node_head == nullptr) == (own_next == nullptr) && (node_head == nullptr) == (grp == nullptr));
  if (n_entries_cap == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  const int32_t *n_own = own_pos + n_entries_cap;
  const unsigned wb = (unsigned)((n_entries_cap + 3) / 4);
  const unsigned tb = (unsigned)((n_entries_cap + 255) / 256);
  if (node_head != nullptr && feat_dim <= 64) {
    int32_t *n_groups = grp + 8 * n_entries_cap;
    const hipError_t me = hipMemsetAsync(n_groups, 0, sizeof(int32_t), st);
    if (me != hipSuccess) { ggad_set_error(me, "mb_hop2_ldsw_gather(memset)"); return GGAD_E_LAUNCH; }
    k_link_owners<<<dim3(tb), dim3(256), 0, st>>>(own_list, n_own, ent_col, node_head, own_next);
    k_build_groups<<<dim3(tb), dim3(256), 0, st>>>(own_list, n_own, ent_col, node_head, own_next, grp, n_groups);
    k_gather2_groups<<<dim3(wb), dim3(256), 0, st>>>(rowptr, col, feat, feat_dim, feat_stride, own_list, ent_col, pw_base, pc, grp,
                                                     n_groups, x2);
  } else {
    k_gather2_w<<<dim3(wb), dim3(256), 0, st>>>(rowptr, col, feat, feat_dim, feat_stride, own_list, n_own, ent_col, pw_base, pc, x2);
  }
  GGAD_CHECK_LAUNCH("mb_hop2_ldsw_gather");
  return GGAD_OK;
}

/* K-tile-major 2-hop with HBM counter slots: for every tile one count launch over all owners, then (after all
 * counts) one gather launch per tile; own_list / own_pos as in ggad_mb_hop2_tiled (compaction done here).
 * cnt2[n_slots][n_nodes] must be zero on entry; x2 must be zero on entry. */
int ggad_mb_hop2_ktile(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                       int64_t n_nodes, const int32_t *tile_off, const int32_t *flags, const int32_t *own_pos,
                       int32_t *own_list, const int32_t *ent_col, const int32_t *ent_slot, int64_t n_entries_cap,
                       int32_t *cnt2, float *x2, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && feat && tile_off && flags && own_pos && own_list && ent_col && ent_slot && cnt2 && x2);
  GGAD_REQUIRE(feat_dim >= 1 && feat_dim <= GGAD_MAX_F && feat_stride >= feat_dim && n_entries_cap >= 0);
  if (n_entries_cap == 0) return GGAD_OK;
  hipStream_t st = as_stream(stream);
  const int n_tiles = (int)((n_nodes + TILE - 1) >> TILE_SHIFT);
  k_owner_compact<<<dim3((unsigned)((n_entries_cap + 255) / 256)), dim3(256), 0, st>>>(flags, own_pos, n_entries_cap, own_list);
  const int32_t *n_own = own_pos + n_entries_cap;      // total of the exclusive scan
  const unsigned blocks = (unsigned)((n_entries_cap + 3) / 4);
  for (int t = 0; t < n_tiles; ++t)
    k_count2_tile<<<dim3(blocks), dim3(256), 0, st>>>(rowptr, col, tile_off, n_tiles, t, own_list, n_own, ent_col, ent_slot,
                                                     n_nodes, cnt2);
  for (int t = 0; t < n_tiles; ++t)
    k_gather2_tile<<<dim3(blocks), dim3(256), 0, st>>>(rowptr, col, feat, feat_dim, feat_stride, tile_off, n_tiles, t, own_list,
                                                      n_own, ent_col, ent_slot, n_nodes, cnt2, x2);
  GGAD_CHECK_LAUNCH("mb_hop2_ktile");
  return GGAD_OK;
}

}  // extern "C"
