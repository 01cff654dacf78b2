// LDS-counting 2-hop aggregation of the DGraph mini-batch path (gfx950): the second half of ggad_mb_plan_build.
//
// Computes, for every owner entry e of a batch (column u = ent_col[e]; the owners are the reference's deduplicated
// `unique_nodes_list`, graphsage.py:306),
//     x2[e] = sum_{k in N(u)} feat[k] / (sqrt(|N(u)|) sqrt(c'_k)),     c'_k = #{owners u' of the batch : k in N(u')}
// i.e. mask_neigh.mm(embed_matrix_expand) with the batch-dependent column normalisation of graphsage.py:335-355.
//
// What the hardware allows (measured, scripts/atomic_bench.hip, scripts/gather_bench.hip): device-scope atomics run at
// ~27 G/s whatever their footprint or clustering, random 4-byte reads at 50-60 G/s, sequential streams are nearly free.
// So (a) the counts c'_k are taken in LDS, (b) they reach the gather as a SEQUENTIAL stream (pc[], one uint16 per
// (owner, neighbour) pair, laid out like the owner's CSR row), which leaves the feature row as the only random access,
// (c) a node that is an owner in several batches of the chunk has its neighbour rows fetched once for up to 8 occurrences.
//
// Every entry of the chunk is an "owner slot" (97 % of the entries ARE owners; the others carry an empty neighbour
// list), so there is no compaction pass and no prefix sum: the per-owner tables are indexed by entry, pc[] storage is
// handed out by k_gather1c with one atomic per wave.
//   k_seg_transpose   tile_off rows of the owners, transposed tile-major (contiguous runs for the counting workgroups)
//   k_tile_counts     one workgroup per (tile of 32,768 ids, batch): pair-parallel LDS counting, writes pc[]
//   k_build_groups    per-node owner lists -> groups of <= 8 occurrences x SLICES of <= 256 neighbours = work items
//   k_gather2_items   one wave per work item, dealt dynamically: every row of a 64-neighbour block in flight at once
//   k_gather2_combine partial sums of multi-slice owners, slices in order
// A hub of 2,000 neighbours is 8 independent work items per group instead of one wave's 32 dependent blocks (that single
// wave WAS the gather's fixed 0.35 ms on small chunks).  Summation order per owner: slices ascending, inside a slice the
// CSR order -- independent of grouping, chunk composition and launch geometry: bit-reproducible.
#include <algorithm>
#include <atomic>
#include <climits>
#include <mutex>

#include "common.h"

namespace {

// Phase clocks of k_gather2_items (a build with -DGGAD_G2_PROF only; scripts/g2_phase_clocks.py): every mark first drains the
// wave's memory counters, so the time a wave spends waiting for each kind of load is attributed to the phase that issued it.
// Totals per launch in counters[GGAD_CTR_PROF ...] as 64-bit words: 0 cursor, 1 item metadata, 2 first ids / counts of a slice,
// 3 rows (issue + wait), 4 weights + fma, 5 stores, 6 whole kernel per wave.
#ifdef GGAD_G2_PROF
struct G2Prof { unsigned long long a[8], t; };
#define G2_ARG , G2Prof &prof
#define G2_PASS , prof
#define G2_MARK(i) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = wall_clock64(); \
                        prof.a[i] += t_ - prof.t; prof.t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
// k_tile_counts, same build: thread 0's clocks per phase (a barrier before every mark), 64-bit words 8..14 of the same block:
// zero + tables, scan, pass 0 register-cached pairs, pass 0 beyond them, pass 1 cached, pass 1 beyond, workgroups.
#define TC_MARK(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); tcp[i] += t_ - tct; tct = t_; } } while (0)
#else
#define G2_ARG
#define G2_PASS
#define G2_MARK(i)
#define TC_MARK(i)
#endif

constexpr int SLICE = 256;              // neighbours per work item
constexpr int GRP_W = 10;               // ints per group record: 8 owner entries (-1 = empty), partial-slot base, slices
constexpr int ITEM_GRAB = 4;            // work items a WAVE takes from the cursor at a time

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

// ---- the tile-major copy of col (round 5; once per graph).  k_tile_counts reads, for every (owner, tile), the 3-4 ids of the owner's
// sorted row that fall into the tile: 16 bytes of a 64-byte sector whose other bytes belong to other tiles, i.e. to other workgroups
// at other times (19.7 bytes of fabric reads per 4-byte id).  col_t holds the same ids TILE-major -- all segments of tile t, in node
// order, contiguous (n_edges / n_tiles ids: 2.6 MB on the bench graph) -- and tile_start[u][t] the position of segment (u, t) in it,
// so the workgroups of one tile (one per batch of the chunk), placed on ONE XCD, read a region that its L2 holds.
//   k_tm_blocksum: ids per (block of 256 nodes, tile);  k_tm_scan: exclusive prefix over the blocks per tile + the tile bases;
//   k_tm_fill: tile_start and the copy.  Lane = tile everywhere (rows of tile_off are read as 256-byte runs).
__global__ void __launch_bounds__(256) k_tm_blocksum(const int32_t *__restrict__ off, int64_t n_nodes, int n_tiles, int32_t *__restrict__ bsum) {
  __shared__ int wsum[4][64];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int NT1 = n_tiles + 1;
  const int64_t u0 = (int64_t)blockIdx.x * 256 + 64 * w;
  for (int t0 = 0; t0 < n_tiles; t0 += 64) {
    const int t = t0 + lane;
    int acc = 0;
    if (t < n_tiles)
      for (int i = 0; i < 64 && u0 + i < n_nodes; ++i) { const int32_t *o = off + (u0 + i) * NT1 + t; acc += o[1] - o[0]; }
    wsum[w][lane] = acc;
    __syncthreads();
    if (w == 0 && t < n_tiles) bsum[(int64_t)blockIdx.x * n_tiles + t] = wsum[0][lane] + wsum[1][lane] + wsum[2][lane] + wsum[3][lane];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(1024) k_tm_scan(int32_t *__restrict__ bsum, int64_t n_blocks, int n_tiles, int32_t *__restrict__ tile_base) {
  // thread = tile: the blocks of 256 nodes in order (reads of a block's row are contiguous over the threads); then the tile bases
  extern __shared__ int tot[];
  for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
    int run = 0;
    int64_t b = 0;
    for (; b + 8 <= n_blocks; b += 8) {
      int v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = bsum[(b + q) * n_tiles + t];
#pragma unroll
      for (int q = 0; q < 8; ++q) { bsum[(b + q) * n_tiles + t] = run; run += v[q]; }
    }
    for (; b < n_blocks; ++b) { const int v = bsum[b * n_tiles + t]; bsum[b * n_tiles + t] = run; run += v; }
    tot[t] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < n_tiles; ++t) { tile_base[t] = run; run += tot[t]; }
    tile_base[n_tiles] = run;
  }
}

__global__ void __launch_bounds__(256) k_tm_fill(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                 const int32_t *__restrict__ off, int64_t n_nodes, int n_tiles,
                                                 const int32_t *__restrict__ bsum, const int32_t *__restrict__ tile_base,
                                                 int32_t *__restrict__ tile_start, int32_t *__restrict__ col_t) {
  __shared__ int wsum[4][64];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int NT1 = n_tiles + 1;
  const int64_t u0 = (int64_t)blockIdx.x * 256 + 64 * w;
  for (int t0 = 0; t0 <= n_tiles; t0 += 64) {
    const int t = t0 + lane;
    int acc = 0;
    if (t < n_tiles)
      for (int i = 0; i < 64 && u0 + i < n_nodes; ++i) { const int32_t *o = off + (u0 + i) * NT1 + t; acc += o[1] - o[0]; }
    wsum[w][lane] = acc;
    __syncthreads();
    if (t < n_tiles) {
      int pos = tile_base[t] + bsum[(int64_t)blockIdx.x * n_tiles + t];
      for (int ww = 0; ww < w; ++ww) pos += wsum[ww][lane];
      for (int i = 0; i < 64 && u0 + i < n_nodes; ++i) {
        const int64_t u = u0 + i;
        const int32_t *o = off + u * NT1 + t;
        const int o0 = o[0], o1 = o[1];
        tile_start[u * NT1 + t] = pos;
        const int32_t *src = col + rowptr[u] + o0;
        for (int j = 0; j < o1 - o0; ++j) col_t[pos + j] = src[j];
        pos += o1 - o0;
      }
    } else if (t == n_tiles) {
      for (int i = 0; i < 64 && u0 + i < n_nodes; ++i) tile_start[(u0 + i) * NT1 + t] = 0;      // (the unused last column)
    }
    __syncthreads();
  }
}

constexpr int TW_SHIFT = 15;
constexpr int TW_TILE = 1 << TW_SHIFT;
constexpr int TW_MAXOWN = 6144;          // owner slots (entries) per batch the LDS arrays hold; larger batches: slabs
constexpr int TW_T = 1024;
constexpr int TW_BRK = 1536;              // bracket entries: (tile, batch) pairs up to 64 * (TW_BRK - 1), beyond that the full descent

// seg_t[t][e] = tile_off[ent_col[e]][t] for owner entries (0 for the others: empty segments), tile-major so that the
// workgroup of (tile t, batch b) reads its owners' segment bounds as two contiguous runs.  (Reading tile_off directly
// costs one random 64-byte sector per (owner, tile).)  One workgroup = 64 entries: table rows read coalesced (one wave per
// row), transposed through LDS, written as 256-byte runs.
constexpr int TT_SLAB = 128;
// With the tile-major copy (tile_start != null) a second plane follows seg_t's n_tiles + 1 rows: seg_s[t][e] = tile_start[ent_col[e]][t],
// where the segment of (owner, tile) begins in col_t.
__global__ void __launch_bounds__(256) k_seg_transpose(const int32_t *__restrict__ tile_off, const int32_t *__restrict__ tile_start,
                                                       int n_tiles, const int32_t *__restrict__ ent_own,
                                                       const int32_t *__restrict__ ent_col, int n_ents, int64_t seg_stride,
                                                       int32_t *__restrict__ seg_t, int skip) {
  __shared__ int tl[TT_SLAB][65];
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  const int p0 = vbx * 64;
  if (p0 >= n_ents) return;
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  const int NT1 = n_tiles + 1;
  // lane j of every wave knows entry p0 + j (owner?, node); a wave then has the 32 row loads of its 16 entries in flight at once
  // (every load unconditional: clamped column, node 0 for non-owners, the value selected afterwards) -- one entry at a time
  // the kernel ran at the latency of two dependent loads per entry (286 us per 150-batch plan for 0.56 GB)
  int u_l = 0, own_l = 0;
  if (p0 + lane < n_ents) { own_l = ent_own[p0 + lane] == p0 + lane ? 1 : 0; u_l = own_l ? ent_col[p0 + lane] : 0; }
  for (int which = 0; which < (tile_start ? 2 : 1); ++which) {
  const int32_t *table = which ? tile_start : tile_off;
  int32_t *plane = seg_t + (which ? (int64_t)NT1 * seg_stride : 0);
  for (int t0 = 0; t0 < NT1; t0 += TT_SLAB) {
    int v0[16], v1[16];
    const int c0 = min(t0 + lane, NT1 - 1), c1 = min(t0 + 64 + lane, NT1 - 1);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = wid + 4 * jj;
      const int u = __shfl(u_l, j, GGAD_WAVE);
      const int32_t *row = table + (int64_t)u * NT1;
      v0[jj] = row[c0];
      v1[jj] = row[c1];
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = wid + 4 * jj;
      const int own = __shfl(own_l, j, GGAD_WAVE);
      if (t0 + lane < NT1) tl[lane][j] = own ? v0[jj] : 0;
      if (t0 + 64 + lane < NT1) tl[64 + lane][j] = own ? v1[jj] : 0;
    }
    __syncthreads();
    const int nt = min(TT_SLAB, NT1 - t0);
    for (int idx = threadIdx.x; idx < nt * 64; idx += 256) {
      const int tt = idx >> 6, j = idx & 63;
      if (p0 + j < n_ents) plane[(int64_t)(t0 + tt) * seg_stride + p0 + j] = tl[tt][j];
    }
    __syncthreads();
  }
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
                                                      const int32_t *__restrict__ batch_ent_ptr,
                                                      const int32_t *__restrict__ pw_base, uint16_t *__restrict__ pc, int n_tiles,
                                                      int n_batches, int32_t *__restrict__ counters, int skip, int tile_major) {
  // tile_major: col = the tile-major copy col_t, the plane behind seg_t's n_tiles + 1 rows = where each segment begins in it, and
  // the workgroups of a tile -- one per batch -- are consecutive members of ONE residue class of blockIdx % 8, i.e. run on ONE XCD,
  // whose L2 then holds the tile's region of col_t (2.6 MB on the bench graph) while its batches pass
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_u[];
  uint32_t *cnt = lds_u;                                   // TW_TILE / 2 words
  int *offs = reinterpret_cast<int *>(lds_u + TW_TILE / 2);  // TW_MAXOWN + 1
  int *segbeg = offs + TW_MAXOWN + 1;                      // TW_MAXOWN   (index into col[])
  int *dst = segbeg + TW_MAXOWN;                           // TW_MAXOWN   (index into pc[])
  int *brk = dst + TW_MAXOWN;                              // TW_BRK: owner of every 64th pair (the search window of a wave's trip)
  __shared__ int wbuf[16];
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  int t, b;
  const bool class_map = (tile_major & 2) != 0;
  tile_major &= 1;
  if (class_map) {
    const int nx = skip < 0 ? 8 : 7, c = (int)(blockIdx.x & 7u);
    const int ci = c - ((skip >= 0 && c > skip) ? 1 : 0);          // this class among the classes that work
    const unsigned j = blockIdx.x >> 3;
    const int nt_c = ci < n_tiles ? (n_tiles - ci + nx - 1) / nx : 0;
    if (j >= (unsigned)nt_c * (unsigned)n_batches) return;
    t = ci + nx * (int)(j / (unsigned)n_batches);
    b = (int)(j % (unsigned)n_batches);
  } else {
    if (vbx >= (unsigned)n_tiles * (unsigned)n_batches) return;
    t = (int)(vbx % (unsigned)n_tiles); b = (int)(vbx / (unsigned)n_tiles);
  }
  const int o0 = batch_ent_ptr[b], o1 = batch_ent_ptr[b + 1];      // "owner slots" = the entries of batch b
  const int n_own_all = o1 - o0;
  const int32_t *seg_lo = seg_t + (int64_t)t * n_cap, *seg_hi = seg_lo + n_cap;
  const int32_t *seg_st = tile_major ? seg_t + (int64_t)(n_tiles + 1 + t) * n_cap : own_rp;      // (plain: the row's start, + lo below)
#ifdef GGAD_G2_PROF
  unsigned long long tcp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tct = wall_clock64();      // phase clocks of thread 0 (TC_MARK)
#endif
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
        {                                                       // all 4 IPT loads of a thread in flight (clamped, unconditional):
          int lo_[IPT], hi_[IPT], rp_[IPT], pb_[IPT];           // one iteration at a time this phase was IPT dependent round trips
#pragma unroll
          for (int q = 0; q < IPT; ++q) {
            const int i = min((int)threadIdx.x + q * TW_T, n_own - 1);
            lo_[q] = seg_lo[ob0 + i]; hi_[q] = seg_hi[ob0 + i]; rp_[q] = seg_st[ob0 + i]; pb_[q] = pw_base[ob0 + i];
          }
#pragma unroll
          for (int q = 0; q < IPT; ++q) {
            const int i = (int)threadIdx.x + q * TW_T;
            if (i < n_own) { offs[i] = hi_[q] - lo_[q]; segbeg[i] = rp_[q] + (tile_major ? 0 : lo_[q]); dst[i] = pb_[q] + lo_[q]; }
          }
        }
        __syncthreads();
        TC_MARK(0);
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
        TC_MARK(1);
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
        // Bracket table: the owner of every 64th pair, one full descent per BLOCK of 64 pairs (not per pair).  The 64 consecutive
        // pairs of a wave's trip lie between two neighbouring brackets -- ~16 owners apart on the bench graph --, so a pair's own
        // descent starts there and takes log2 of that distance (4-5) LDS reads at random addresses instead of 12-13: the located
        // pairs were bound by LDS bank conflicts (9.8 of 27 us per workgroup, scripts/g2_phase_clocks.py).
        const int nblk = ((P - 1) >> 6) + 1;
        const bool use_brk = nblk + 1 <= TW_BRK;                 // uniform
        if (use_brk && (pass == 0 || n_slabs > 1)) {
          for (int b = threadIdx.x; b <= nblk; b += TW_T) {
            int own = n_own, j = 0;
            if (b < nblk) locate(min(64 * b, P - 1), own, j);
            brk[b] = own;
          }
          __syncthreads();
        }
        auto locate4 = [&](const int (&pp)[4], int (&own)[4], int (&j)[4]) {
          int lo[4] = {0, 0, 0, 0};
          int smax = s0;
          if (use_brk) {
            smax = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int b = pp[q] >> 6;
              lo[q] = brk[b];
              const int w = brk[b + 1] - lo[q];
              smax = max(smax, w > 0 ? 1 << (31 - __clz(w)) : 0);
            }
          }
          for (int st = smax; st > 0; st >>= 1) {
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
            // phase A: locate all CACHE_IT pairs of this thread (LDS only); phase B: ALL their column loads in flight at once
            // (pairs clamped to P - 1, so every load is unconditional); phase C: the LDS counts
#pragma unroll
            for (int g4 = 0; g4 < CACHE_IT; g4 += 4) {
              if (g4 * TW_T < P) {                                // (uniform: a group of four trips wholly beyond P is not located)
                int pp[4], own[4], j[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) pp[q] = min((g4 + q) * TW_T + (int)threadIdx.x, P - 1);
                locate4(pp, own, j);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  kc[g4 + q] = col[segbeg[own[q]] + j[q]];      // requested now: in flight while the next four pairs are located
                  dc[g4 + q] = dst[own[q]] + j[q];
                }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) { kc[g4 + q] = 0; dc[g4 + q] = 0; }
              }
            }
#pragma unroll
            for (int it = 0; it < CACHE_IT; ++it) {
              const bool on = it * TW_T + (int)threadIdx.x < P;
              const int loc = kc[it] & (TW_TILE - 1);
              if (on) atomicAdd(&cnt[loc >> 1], 1u << ((loc & 1) << 4));
              kc[it] = on ? loc : -1;
            }
          } else {
#pragma unroll
            for (int it = 0; it < CACHE_IT; ++it)
              if (kc[it] >= 0) pc[dc[it]] = (uint16_t)((cnt[kc[it] >> 1] >> ((kc[it] & 1) << 4)) & 0xFFFFu);
          }
          p_start = CACHE_IT * TW_T;
          TC_MARK(2 + 2 * pass);
        }
        for (int p0 = p_start; p0 < P; p0 += 4 * TW_T) {
          int pp[4], own[4], j[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pp[q] = min(p0 + q * TW_T + (int)threadIdx.x, P - 1);
          locate4(pp, own, j);
          int k4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) k4[q] = col[segbeg[own[q]] + j[q]];      // (pairs clamped to P - 1: unconditional, all four in flight;
#pragma unroll                                                               //  eight per trip measured slower)
          for (int q = 0; q < 4; ++q) {
            if (p0 + q * TW_T + (int)threadIdx.x < P) {
              const int loc = k4[q] & (TW_TILE - 1);
              if (pass == 0) atomicAdd(&cnt[loc >> 1], 1u << ((loc & 1) << 4));
              else pc[(int64_t)dst[own[q]] + j[q]] = (uint16_t)((cnt[loc >> 1] >> ((loc & 1) << 4)) & 0xFFFFu);
            }
          }
        }
        TC_MARK(3 + 2 * pass);
      }
      __syncthreads();
    }
  }
#ifdef GGAD_G2_PROF
  if (threadIdx.x == 0) {
    unsigned long long *out = reinterpret_cast<unsigned long long *>(counters + GGAD_CTR_PROF) + 8;
    for (int k = 0; k < 6; ++k) atomicAdd(out + k, tcp[k]);
    atomicAdd(out + 6, 1ull);
  }
#endif
}

// ---- the gather.  One SLICE of one owner group: neighbours [n0, n1) of node u (CSR row start s), weights
// 1 / (sqrt deg(u) sqrt c') from the streamed counts of up to G occurrences, rpi = 64 / F rows per load instruction.
// All row loads of a 64-neighbour block are issued before the first is used (FT = 17: 22 loads in flight per wave); the
// fma order per accumulator is t ascending, blocks ascending, whatever the batching of the loads.
template <int G, int FT>
__device__ __forceinline__ void gather_slice(const int32_t *__restrict__ col, const float *__restrict__ feat, int F_rt, int stride,
                                             int s, int n0, int n1, float inv_sr, int n_occ, const int (&pb)[8],
                                             const uint16_t *__restrict__ pc, int lane, int k_first, float (&tot)[G] G2_ARG) {
  const int F = FT ? FT : F_rt;
  const int rpi = 64 / F;
  const int g = lane / F, f = lane - g * F;
  const bool act = g < rpi;
  const int gl = act ? g : rpi - 1;      // padding lanes (64 - rpi F of them) re-read the last group's row: the vector L1 works per
                                         // 128-byte line (~3.8 clocks each), a 4th line per load instruction costs a third more
  // the table is < 4 GB (checked on the host): a row load is "uniform base + 32-bit byte offset", one address VGPR per load
  const char *fb = reinterpret_cast<const char *>(feat);
  const uint32_t rs = (uint32_t)stride * 4u, fo = (uint32_t)f * 4u;
  auto row_elem = [&](int kk) -> float { return *reinterpret_cast<const float *>(fb + ((uint32_t)kk * rs + fo)); };
  float acc[G];
#pragma unroll
  for (int j = 0; j < G; ++j) acc[j] = 0.0f;
  // ids and streamed counts of block b + 1 are requested while the rows of block b are in flight (one memory round trip per
  // 64 neighbours instead of two); the weights of a block are formed from its counts before the request for the next block
  // re-uses their registers.  EVERY load is unconditional (indices clamped into the slice, node 0 for padding lanes, weight
  // 0 for what must not count): a guarded load compiles to a branch + s_waitcnt each and serialises the wave.
  // The ids of the first block come from the caller (requested while the previous work item was being summed); the counts are
  // requested here and only waited for AFTER the block's row loads have been issued: a slice starts with ONE memory round trip
  // (rows; counts beside them), not three (ids, counts, rows).
  const int last = max(n1 - 1, 0);
  int k_nx = k_first;
  uint32_t c_nx[G];
  {
    const int ix = min(n0 + lane, last);
#pragma unroll
    for (int j = 0; j < G; ++j) c_nx[j] = pc[pb[j] + ix];
  }
  G2_MARK(2);
  for (int blk = n0; blk < n1; blk += 64) {
    const bool mine = blk + lane < n1;
    const int k = mine ? k_nx : 0;
    const int count = min(64, n1 - blk);
    float w[G];
    if (FT != 17) {
#pragma unroll
      for (int j = 0; j < G; ++j) w[j] = (j < n_occ && mine) ? inv_sr / sqrtf((float)c_nx[j]) : 0.0f;   // .div(row).div(col)  graphsage.py:348
    }
    if (FT == 17) {
      constexpr int IT = 22;                                   // ceil(64 / 3)
      float x[IT];
#pragma unroll
      for (int t = 0; t < IT; ++t) {
        const int kk = __shfl(k, (t * 3 + gl) & 63, GGAD_WAVE);   // a valid id or 0: always a readable row
        x[t] = row_elem(kk);
      }
#pragma unroll
      for (int j = 0; j < G; ++j) w[j] = (j < n_occ && mine) ? inv_sr / sqrtf((float)c_nx[j]) : 0.0f;   // .div(row).div(col)  graphsage.py:348
      {
        const int ix = min(blk + 64 + lane, last);
        k_nx = col[s + ix];
#pragma unroll
        for (int j = 0; j < G; ++j) c_nx[j] = pc[pb[j] + ix];
      }
      G2_MARK(3);
#pragma unroll
      for (int t = 0; t < IT; ++t) {
        const int src = t * 3 + g;
        const float xv = (act && src < count) ? x[t] : 0.0f;
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const float ws = __shfl(w[j], src & 63, GGAD_WAVE);
          acc[j] = fmaf((src < count) ? ws : 0.0f, xv, acc[j]);
        }
      }
      G2_MARK(4);
    } else {
      const int iters = (count + rpi - 1) / rpi;
      int tt = 0;
      for (; tt + 8 <= iters; tt += 8) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int kk = __shfl(k, ((tt + q) * rpi + gl) & 63, GGAD_WAVE);
          x[q] = row_elem(kk);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int src = (tt + q) * rpi + g;
          const float xv = (act && src < count) ? x[q] : 0.0f;
#pragma unroll
          for (int j = 0; j < G; ++j) {
            const float ws = __shfl(w[j], src & 63, GGAD_WAVE);
            acc[j] = fmaf((src < count) ? ws : 0.0f, xv, acc[j]);
          }
        }
      }
      for (; tt < iters; ++tt) {
        const int src = tt * rpi + g;
        const int kk = __shfl(k, (tt * rpi + gl) & 63, GGAD_WAVE);
        const float xr = row_elem(kk);
        const float xv = (act && src < count) ? xr : 0.0f;
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const float ws = __shfl(w[j], src & 63, GGAD_WAVE);
          acc[j] = fmaf((src < count) ? ws : 0.0f, xv, acc[j]);
        }
      }
      const int ix = min(blk + 64 + lane, last);
      k_nx = col[s + ix];
#pragma unroll
      for (int j = 0; j < G; ++j) c_nx[j] = pc[pb[j] + ix];
    }
  }
#pragma unroll
  for (int j = 0; j < G; ++j) {
    float t = acc[j];
    for (int q = 1; q < rpi; ++q) t += __shfl(acc[j], (lane + q * F) & 63, GGAD_WAVE);
    tot[j] = t;
  }
}

// ---- the same slice on the matrix cores (F = 17, rows padded to 32 floats = one 128-byte line: the trainer's table).
//   x2[j][f] += sum_k w[j][k] x[k][f]  is a (8 occurrences x 64 neighbours) . (64 x 32) product per block:
//   v_mfma_f32_16x16x4_f32, lane (a = l % 16, g = l / 16) holds A[a][g], B[g][a], D[4 g + v][a].  K-step i takes the neighbours
//   {i, 16 + i, 32 + i, 48 + i} of the block (k = g), A[a][g] = w[occurrence a][16 g + i], B[g][a] = the features (2 a, 2 a + 1) of
//   that neighbour: ONE 8-byte load per lane and k-step = 4 rows x 128 bytes per instruction, 16 load instructions per block
//   instead of 22, and two products per k-step (even / odd features; columns >= 17 are the zero padding of the rows).
//   Ids and weights change layout (lane = neighbour -> lane = (occurrence | feature pair, neighbour group)) through a
//   wave-private piece of LDS: 1 + G stores and 8 16-byte reads per block.  The VALU version spends 22 G ds_bpermute + 22 G fma
//   per block on the weights: 41 % of the wave time of a 150-batch launch (scripts/g2_phase_clocks.py).
//   Order of the sum per (occurrence, feature): k-step i ascending, inside a step g ascending; blocks ascending: fixed.
constexpr int MF_WS = 68;                                   // floats per weight row in LDS (16-byte reads of 8 rows: conflict-free)
constexpr int MF_LDS = 8 * MF_WS + 64;                      // floats per wave
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef float mf2 __attribute__((ext_vector_type(2)));
typedef int mi4 __attribute__((ext_vector_type(4)));

template <int G>
__device__ __forceinline__ void gather_slice_mfma(const int32_t *__restrict__ col, const float *__restrict__ feat, int s, int n0, int n1,
                                                  float inv_sr, int n_occ, const int (&pb)[8], const uint16_t *__restrict__ pc,
                                                  int lane, float *lw, int k_first, float (&tot)[G] G2_ARG) {
  const int a = lane & 15, g = lane >> 4;
  const char *fb = reinterpret_cast<const char *>(feat) + a * 8;
  int *lids = reinterpret_cast<int *>(lw + 8 * MF_WS);
  const int aa = a < G ? a : 0;
  mf4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const int last = max(n1 - 1, 0);
  int k_nx = k_first;                                        // (clamped by the caller: always the id of a neighbour of this slice)
  uint32_t c_nx[G];
  {
    const int ix = min(n0 + lane, last);
#pragma unroll
    for (int j = 0; j < G; ++j) c_nx[j] = pc[pb[j] + ix];
  }
  G2_MARK(2);
  for (int blk = n0; blk < n1; blk += 64) {
    const bool mine = blk + lane < n1;
    // ids first: their LDS round trip and the 16 row requests do not wait for the counts
    lids[lane] = k_nx;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int id[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const mi4 iv = *reinterpret_cast<const mi4 *>(lids + 16 * g + 4 * c);
#pragma unroll
      for (int q = 0; q < 4; ++q) id[4 * c + q] = iv[q];
    }
    mf2 x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = *reinterpret_cast<const mf2 *>(fb + ((uint32_t)id[i] << 7));
#pragma unroll
    for (int j = 0; j < G; ++j)
      lw[j * MF_WS + lane] = (j < n_occ && mine) ? inv_sr / sqrtf((float)c_nx[j]) : 0.0f;     // .div(row).div(col)  graphsage.py:348
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float wa[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const mf4 wv = *reinterpret_cast<const mf4 *>(lw + aa * MF_WS + 16 * g + 4 * c);
#pragma unroll
      for (int q = 0; q < 4; ++q) wa[4 * c + q] = a < G ? wv[q] : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    {
      const int ix = min(blk + 64 + lane, last);
      k_nx = col[s + ix];
#pragma unroll
      for (int j = 0; j < G; ++j) c_nx[j] = pc[pb[j] + ix];
    }
    G2_MARK(3);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], x[i].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], x[i].y, acc1, 0, 0, 0);
    }
    G2_MARK(4);
  }
  // D[j][f]: tile f & 1, column f >> 1, i.e. lane (f >> 1) + 16 (j >> 2), register j & 3  ->  lane f of tot[j]
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int src = (lane >> 1) + 16 * (j >> 2);
    const float v0 = __shfl(acc0[j & 3], src & 63, GGAD_WAVE), v1 = __shfl(acc1[j & 3], src & 63, GGAD_WAVE);
    tot[j] = (lane & 1) ? v1 : v0;
  }
}

// One thread per entry.  The owner entry that was linked LAST into its node's list (k_gather1c) acts for the node: it cuts
// the list into groups of <= 8 occurrences and reserves the partial-sum slots of owners that are summed in parts; the cursors are
// bumped once per WORKGROUP (prefix sums over the lanes, wave totals through LDS).  Clears node_head[u].
//   deg <= range_deg (always, unless ggad_mb_set_gather_options turns the ranges on: measured slower): every group
//                          becomes ceil(deg / SLICE) work items of the common list (any XCD takes them);
//   deg >  range_deg:      the group goes on the `big` list with the node's GGAD_RANGES + 1 range boundaries (positions inside its
//                          CSR row where the neighbour ids cross an eighth of the tile range: tile_off[u][tile(r)]).  The waves of
//                          XCD x walk that list for range x, so an L2 only ever sees an eighth of the id space from these owners
//                          (84 % of the pairs of the bench graph) and keeps an eighth of the hot rows: scripts/gather_range_bench.hip
//                          measures 72.6 against 46.2 G rows/s for this access pattern.  The GGAD_RANGES partial sums of an
//                          occurrence are added in range order (k_gather2_combine): fixed by the ids, not by the launch.
// Groups of one node are adjacent in both lists (their waves run side by side and find each other's rows in L2).
__device__ __forceinline__ int range_tile(int r, int n_tiles) { return (int)(((int64_t)r * n_tiles) / GGAD_RANGES); }

__global__ void __launch_bounds__(256) k_build_groups(const int32_t *__restrict__ ent_own, const int32_t *__restrict__ ent_col,
                                                      const int32_t *__restrict__ own_deg, int n_ents,
                                                      int32_t *__restrict__ node_head, const int32_t *__restrict__ own_next,
                                                      const int32_t *__restrict__ tile_off, int n_tiles,
                                                      int32_t *__restrict__ grp, int32_t *__restrict__ items,
                                                      int32_t *__restrict__ big, int32_t *__restrict__ gbnd,
                                                      int32_t *__restrict__ counters, int range_deg, int skip) {
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  const int e = vbx * 256 + threadIdx.x;
  const int lane = lane_id();
  int u = 0, m = 0, ng = 0, ns = 0, deg = 0;
  if (e < n_ents && ent_own[e] == e) {
    u = ent_col[e];
    if (node_head[u] == e + 1) {
      for (int cur = e + 1; cur != 0; cur = own_next[cur - 1]) ++m;
      ng = (m + 7) >> 3;
      deg = own_deg[e];
      ns = deg > range_deg ? GGAD_RANGES : (deg > SLICE ? (deg + SLICE - 1) / SLICE : 1);
    }
  }
  const bool is_big = deg > range_deg;
  const int c1 = is_big ? 0 : ng * ns, c2 = ns > 1 ? m * ns : 0, c3 = is_big ? ng : 0;
  int v0 = ng, v1 = c1, v2 = c2, v3 = c3;                       // inclusive prefix sums over the wave
#pragma unroll
  for (int off = 1; off < GGAD_WAVE; off <<= 1) {
    const int t0 = __shfl_up(v0, off, GGAD_WAVE), t1 = __shfl_up(v1, off, GGAD_WAVE), t2 = __shfl_up(v2, off, GGAD_WAVE),
              t3 = __shfl_up(v3, off, GGAD_WAVE);
    if (lane >= off) { v0 += t0; v1 += t1; v2 += t2; v3 += t3; }
  }
  // one reservation per WORKGROUP (the four waves' totals meet in LDS): same-address atomics are serialised chip-wide
  __shared__ int wg_tot[4][5];
  const int wid = threadIdx.x >> 6;
  if (lane == GGAD_WAVE - 1) { wg_tot[0][wid] = v0; wg_tot[1][wid] = v1; wg_tot[2][wid] = v2; wg_tot[3][wid] = v3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    int sum = 0;
    for (int w = 0; w < 4; ++w) { const int t = wg_tot[k][w]; wg_tot[k][w] = sum; sum += t; }
    int32_t *ctr = counters + (k == 0 ? GGAD_CTR_GROUPS : (k == 1 ? GGAD_CTR_ITEMS : (k == 2 ? GGAD_CTR_PART : GGAD_CTR_BIG)));
    wg_tot[k][4] = sum > 0 ? atomicAdd(ctr, sum) : 0;
  }
  __syncthreads();
  const int b0 = wg_tot[0][4] + wg_tot[0][wid], b1 = wg_tot[1][4] + wg_tot[1][wid], b2 = wg_tot[2][4] + wg_tot[2][wid],
            b3 = wg_tot[3][4] + wg_tot[3][wid];
  if (ng == 0) return;
  const int gbase = b0 + v0 - ng, ibase = b1 + v1 - c1, pbase = b2 + v2 - c2, bbase = b3 + v3 - c3;
  int bnd[GGAD_RANGES + 1];
  if (is_big) {
    const int32_t *row = tile_off + (int64_t)u * (n_tiles + 1);
#pragma unroll
    for (int r = 0; r <= GGAD_RANGES; ++r) bnd[r] = row[range_tile(r, n_tiles)];      // bnd[0] = 0, bnd[GGAD_RANGES] = deg
  }
  int cur = e + 1;
  for (int g = 0; g < ng; ++g) {
    int32_t *rec = grp + (int64_t)(gbase + g) * GRP_W;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int v = -1;
      if (cur != 0) { v = cur - 1; cur = own_next[v]; }
      rec[j] = v;
    }
    rec[8] = pbase + g * 8 * ns;
    rec[9] = ns;
    if (is_big) {
      big[bbase + g] = gbase + g;
      int32_t *bo = gbnd + (int64_t)(bbase + g) * (GGAD_RANGES + 1);
#pragma unroll
      for (int r = 0; r <= GGAD_RANGES; ++r) bo[r] = bnd[r];
    } else {
      for (int sl = 0; sl < ns; ++sl) {        // slice-major: consecutive items = the same neighbour rows for the node's groups
        int32_t *it = items + (int64_t)(ibase + sl * ng + g) * 2;
        it[0] = gbase + g;
        it[1] = sl;
      }
    }
  }
  node_head[u] = 0;
}

template <int FT, bool MF>
__global__ void __launch_bounds__(256) k_gather2_items(const int32_t *__restrict__ col, const float *__restrict__ feat, int F_rt,
                                                       int stride, const int32_t *__restrict__ own_rp,
                                                       const int32_t *__restrict__ own_deg, const int32_t *__restrict__ pw_base,
                                                       const uint16_t *__restrict__ pc, const int32_t *__restrict__ grp,
                                                       const int32_t *__restrict__ items, const int32_t *__restrict__ big,
                                                       const int32_t *__restrict__ gbnd, int32_t *__restrict__ counters,
                                                       float *__restrict__ x2, float *__restrict__ part2, int affine, int take_cap, int skip) {
  static_assert(ITEM_GRAB == 4 && GRP_W <= 10, "lane layout of the metadata loads: 16 lanes per item of a grab (10, 11: range bounds)");
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  const int F = FT ? FT : F_rt;
  const int lane = lane_id();
  __shared__ __attribute__((aligned(16))) float mf_lds[MF ? 4 * MF_LDS : 4];
  float *lw = mf_lds + (MF ? (threadIdx.x >> 6) * MF_LDS : 0);
  const int n_small = counters[GGAD_CTR_ITEMS], n_big = counters[GGAD_CTR_BIG];
  const int q_l = lane >> 4, r_l = lane & 15;                 // metadata phase: lane = (item of the grab, word)
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= GGAD_RANGES - 1;
  if (!affine) xcc = (vbx / GGAD_RANGES) & (GGAD_RANGES - 1);      // (experiment: the same work items without the XCD binding)
  // Work lists in the order a wave visits them: range `xcc` of the big list (the rows of that eighth of the id space stay in
  // THIS XCD's L2), the common list of the small owners, then the other ranges -- whatever their XCDs have not finished (an XCD
  // left to the resident step kernel has no waves here: its range is taken by everybody at the end).
  // Guided self-scheduling on the cursor of a list: a wave reserves rem / (2 waves) items at a time (64 at most, ITEM_GRAB at
  // least; rem is what its previous reservation saw left), i.e. big bites while there is plenty and single grabs at the end.  A
  // same-address atomic completes every ~7 ns chip-wide: one atomic per 4 items (268 K per 150-batch launch) was 1.9 of the
  // kernel's 2.3 ms; a list that is already exhausted is recognised by a plain load.
  const int n_waves_all = (int)vgx * 4;
#ifdef GGAD_G2_PROF
  G2Prof prof = {};
  const unsigned long long t_begin = wall_clock64();
  prof.t = t_begin;
#endif
  for (int ph = 0; ph <= GGAD_RANGES; ++ph) {
    const int cls = ph == 0 ? (int)xcc : (ph == 1 ? GGAD_RANGES : (int)((xcc + ph - 1) & (GGAD_RANGES - 1)));
    const bool bigc = cls < GGAD_RANGES;
    const int n_items = bigc ? n_big : n_small;
    if (n_items == 0) continue;
    int32_t *cursor = counters + (bigc ? GGAD_CTR_RANGE0 + cls * GGAD_CTR_STRIDE : GGAD_CTR_CURSOR);
    const int n_waves = ph == 0 ? max(n_waves_all / GGAD_RANGES, 1) : n_waves_all;
    int seen = 0;
    if (ph != 0) {
      seen = __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= n_items) continue;
    }
  for (;;) {
    const int rem = n_items - seen;
    int take = (rem / (2 * n_waves)) & ~(ITEM_GRAB - 1);
    take = take < ITEM_GRAB ? ITEM_GRAB : (take > take_cap ? take_cap : take);
    int first = 0;
    if (lane == 0) first = atomicAdd(cursor, take);
    first = __builtin_amdgcn_readfirstlane(first);
    G2_MARK(0);
    if (first >= n_items) break;
    seen = first + take;
    const int last = min(n_items, first + take);
#pragma unroll 1
    for (int base = first; base < last; base += ITEM_GRAB) {
    // three dependent loads for the FOUR items of the grab (item -> group record -> owner metadata) instead of three per item
    const bool it_ok = base + q_l < last;
    int iv;                                                   // r 0: group, r 1: slice (common list) / range (big list)
    if (bigc) iv = r_l == 0 ? (it_ok ? big[base + q_l] : 0) : cls;
    else iv = (it_ok && r_l < 2) ? items[(int64_t)(base + q_l) * 2 + r_l] : 0;
    const int bv = (bigc && it_ok && (r_l == 10 || r_l == 11)) ? gbnd[(int64_t)(base + q_l) * (GGAD_RANGES + 1) + cls + (r_l - 10)] : 0;
    const int gi_l = __shfl(iv, q_l * 16, GGAD_WAVE);
    const int rv = (it_ok && r_l < GRP_W) ? grp[(int64_t)gi_l * GRP_W + r_l] : -1;
    const int pbv = (r_l < 8 && rv >= 0) ? pw_base[rv] : 0;
    const int rpv = (it_ok && r_l == 0) ? own_rp[rv] : 0;
    const int dgv = (it_ok && r_l == 0) ? own_deg[rv] : 0;
    G2_MARK(1);
    // neighbour range of item q of the grab (CSR row start, [n0, n1), degree)
    auto item_span = [&](int q, int &s_, int &n0_, int &n1_, int &deg_) {
      s_ = __builtin_amdgcn_readlane(rpv, q * 16);
      deg_ = __builtin_amdgcn_readlane(dgv, q * 16);
      const int sl_ = __builtin_amdgcn_readlane(iv, q * 16 + 1);
      n0_ = bigc ? __builtin_amdgcn_readlane(bv, q * 16 + 10) : sl_ * SLICE;
      n1_ = bigc ? __builtin_amdgcn_readlane(bv, q * 16 + 11) : min(deg_, n0_ + SLICE);
    };
    auto first_ids = [&](int s_, int n0_, int n1_) { return col[s_ + min(n0_ + lane, max(n1_ - 1, 0))]; };
    int k_ahead = 0;                                           // ids of the first block of the NEXT item: in flight while this one is summed
    { int s_, a_, b_, d_; item_span(0, s_, a_, b_, d_); k_ahead = first_ids(s_, a_, b_); }
#pragma unroll 1
    for (int q = 0; q < ITEM_GRAB; ++q) {
      if (base + q >= last) break;
      int e[8], pb[8];
      int n = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = __builtin_amdgcn_readlane(rv, q * 16 + j);
        pb[j] = __builtin_amdgcn_readlane(pbv, q * 16 + j);
        n += e[j] >= 0 ? 1 : 0;
      }
      const int pbase = __builtin_amdgcn_readlane(rv, q * 16 + 8), ns = __builtin_amdgcn_readlane(rv, q * 16 + 9);
      const int sl = __builtin_amdgcn_readlane(iv, q * 16 + 1);
      int s, n0, n1, deg;
      item_span(q, s, n0, n1, deg);
      const float inv_sr = 1.0f / sqrtf((float)deg);        // deg = 0 -> inf * 0 = NaN, as the dense 0/0 row (quirk 3)
      const int k_first = k_ahead;
      if (q + 1 < ITEM_GRAB && base + q + 1 < last) {
        int s_, a_, b_, d_;
        item_span(q + 1, s_, a_, b_, d_);
        k_ahead = first_ids(s_, a_, b_);
      }
      float tot[8];
      if (MF) {
        if (n == 1) { float t1[1]; gather_slice_mfma<1>(col, feat, s, n0, n1, inv_sr, n, pb, pc, lane, lw, k_first, t1 G2_PASS); tot[0] = t1[0]; }
        else if (n == 2) { float t2[2]; gather_slice_mfma<2>(col, feat, s, n0, n1, inv_sr, n, pb, pc, lane, lw, k_first, t2 G2_PASS); tot[0] = t2[0]; tot[1] = t2[1]; }
        else if (n <= 4) { float t4[4]; gather_slice_mfma<4>(col, feat, s, n0, n1, inv_sr, n, pb, pc, lane, lw, k_first, t4 G2_PASS);
#pragma unroll
          for (int j = 0; j < 4; ++j) tot[j] = t4[j]; }
        else gather_slice_mfma<8>(col, feat, s, n0, n1, inv_sr, n, pb, pc, lane, lw, k_first, tot G2_PASS);
      } else
      if (n == 1) { float t1[1]; gather_slice<1, FT>(col, feat, F, stride, s, n0, n1, inv_sr, n, pb, pc, lane, k_first, t1 G2_PASS); tot[0] = t1[0]; }
      else if (n == 2) { float t2[2]; gather_slice<2, FT>(col, feat, F, stride, s, n0, n1, inv_sr, n, pb, pc, lane, k_first, t2 G2_PASS); tot[0] = t2[0]; tot[1] = t2[1]; }
      else if (n <= 4) { float t4[4]; gather_slice<4, FT>(col, feat, F, stride, s, n0, n1, inv_sr, n, pb, pc, lane, k_first, t4 G2_PASS);
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[j] = t4[j]; }
      else gather_slice<8, FT>(col, feat, F, stride, s, n0, n1, inv_sr, n, pb, pc, lane, k_first, tot G2_PASS);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < n && lane < F) {
          if (ns == 1) x2[(int64_t)e[j] * F + lane] = (deg == 0) ? inv_sr * 0.0f : tot[j];
          else part2[((int64_t)pbase + (int64_t)j * ns + sl) * F + lane] = tot[j];
        }
      }
      G2_MARK(5);
    }
    }
  }
  }
#ifdef GGAD_G2_PROF
  if (lane == 0) {
    prof.a[6] = wall_clock64() - t_begin;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(counters + GGAD_CTR_PROF);
    for (int k = 0; k < 7; ++k) atomicAdd(out + k, prof.a[k]);
  }
#endif
}

// x2 of owners summed in parts: slices / ranges in order (0 + P_0 + P_1 + ...).
__global__ void __launch_bounds__(256) k_gather2_combine(const int32_t *__restrict__ grp, const int32_t *__restrict__ counters,
                                                         const float *__restrict__ part2, int F, float *__restrict__ x2, int skip) {
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  const int lane = lane_id();
  const int n_groups = counters[GGAD_CTR_GROUPS];
  const int rpi = 64 / F;
  const int g = lane / F, f = lane - g * F;
  for (int gi = vbx * 4 + (threadIdx.x >> 6); gi < n_groups; gi += vgx * 4) {
    const int rv = (lane < GRP_W) ? grp[(int64_t)gi * GRP_W + lane] : -1;
    const int ns = __builtin_amdgcn_readlane(rv, 9);
    if (ns <= 1) continue;
    const int pbase = __builtin_amdgcn_readlane(rv, 8);
    for (int j0 = 0; j0 < 8; j0 += rpi) {
      const int j = j0 + g;
      const int ej = __shfl(rv, j & 7, GGAD_WAVE);
      if (g < rpi && j < 8 && ej >= 0) {
        float acc = 0.0f;
#pragma unroll 4
        for (int sl = 0; sl < ns; ++sl) acc += part2[((int64_t)pbase + (int64_t)j * ns + sl) * F + f];
        x2[(int64_t)ej * F + f] = acc;
      }
    }
  }
}

// One wave per owner entry (feat_dim > 64, or node_major off): same slices, same order, no sharing of row fetches.
__global__ void __launch_bounds__(256) k_gather2_w(const int32_t *__restrict__ col, const float *__restrict__ feat, int F, int stride,
                                                   const int32_t *__restrict__ ent_own, int n_ents,
                                                   const int32_t *__restrict__ own_rp, const int32_t *__restrict__ own_deg,
                                                   const int32_t *__restrict__ pw_base, const uint16_t *__restrict__ pc,
                                                   float *__restrict__ x2, int skip) {
  unsigned vbx, vgx;
  if (!ggad_vblock(skip, vbx, vgx)) return;
  const int e0 = vbx * 4 + (threadIdx.x >> 6);
  if (e0 >= n_ents || ent_own[e0] != e0) return;
  const int lane = lane_id();
  const int s = own_rp[e0];
  const int deg = own_deg[e0];
  const int pb0 = pw_base[e0];
  const float inv_sr = 1.0f / sqrtf((float)deg);
  if (F <= 64) {
    const int pb[8] = {pb0, 0, 0, 0, 0, 0, 0, 0};
    float total = 0.0f;
#ifdef GGAD_G2_PROF
    G2Prof prof = {};
#endif
    for (int n0 = 0; n0 < deg; n0 += SLICE) {
      float t1[1];
      const int kf = col[s + min(n0 + lane, max(min(deg, n0 + SLICE) - 1, 0))];
      if (F == 17) gather_slice<1, 17>(col, feat, F, stride, s, n0, min(deg, n0 + SLICE), inv_sr, 1, pb, pc, lane, kf, t1 G2_PASS);
      else gather_slice<1, 0>(col, feat, F, stride, s, n0, min(deg, n0 + SLICE), inv_sr, 1, pb, pc, lane, kf, t1 G2_PASS);
      total += t1[0];
    }
    if (deg == 0) total = inv_sr * 0.0f;
    if (lane < F) x2[(int64_t)e0 * F + lane] = total;
    return;
  }
  for (int fbase = 0; fbase < F; fbase += 64) {               // wide rows: lane = feature, one row per load instruction
    const int fw = min(64, F - fbase);
    float total = 0.0f;
    for (int n0 = 0; n0 < deg; n0 += SLICE) {
      const int n1 = min(deg, n0 + SLICE);
      float acc = 0.0f;
      for (int blk = n0; blk < n1; blk += 64) {
        const int idx = blk + lane;
        int k = 0;
        float w = 0.0f;
        if (idx < n1) { k = col[s + idx]; w = inv_sr / sqrtf((float)pc[pb0 + idx]); }
        const int count = min(64, n1 - blk);
        for (int t = 0; t < count; ++t) {
          const int kk = __shfl(k, t, GGAD_WAVE);
          const float ws = __shfl(w, t, GGAD_WAVE);
          const float x = lane < fw ? feat[(int64_t)kk * stride + fbase + lane] : 0.0f;
          acc = fmaf(ws, x, acc);
        }
      }
      total += acc;
    }
    if (deg == 0) total = inv_sr * 0.0f;
    if (lane < fw) x2[(int64_t)e0 * F + fbase + lane] = total;
  }
}

// process-wide options of the 2-hop gather (ggad_mb_set_gather_options): the environment gives the defaults
std::atomic<int> g_mfma_batches{[] { const char *e = getenv("GGAD_GATHER_MFMA_BATCHES"); return e ? atoi(e) : 0; }()};
std::atomic<int> g_range_deg{[] { const char *e = getenv("GGAD_RANGE_DEG"); return e ? atoi(e) : 0; }()};

}  // namespace

int ggad_int_range_deg() { return g_range_deg.load() > 0 ? std::max(g_range_deg.load(), GGAD_RANGE_DEG) : INT32_MAX; }

int ggad_int_ldsw_hop2(const ggad_mb_plan *P, const ggad_plan_view &V, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
  if (V.n_ents == 0 || V.n_batches == 0) return GGAD_OK;
  const int n_tiles = (int)((P->n_nodes + TW_TILE - 1) >> TW_SHIFT);
  const int skip = P->xcd_skip >= 0 && P->xcd_skip < 8 ? P->xcd_skip : -1;      // leave one XCD to the resident chunk kernel
  const bool tile_major = P->tile_start != nullptr && P->col_t != nullptr;      // (the tile-major copy of col: ggad_mb_tile_major)
  k_seg_transpose<<<dim3(ggad_skip_grid((unsigned)((V.n_ents + 63) / 64), skip)), dim3(256), 0, st>>>(
      P->tile_off, tile_major ? P->tile_start : nullptr, n_tiles, P->ent_own, P->ent_col, V.n_ents, V.seg_stride, P->seg_t, skip);
  const size_t lds = (size_t)(TW_TILE / 2) * 4 + (size_t)(3 * TW_MAXOWN + 1) * 4 + (size_t)TW_BRK * 4;
  {  // the opt-in for 139 KB of dynamic LDS is a per-DEVICE attribute of the kernel: set (and checked) once per device of the process
    static std::mutex mu;
    static int state[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GGAD_E_INVALID;
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev] == 0) {
      const bool ok = hipFuncSetAttribute((const void *)k_tile_counts, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      (void)hipGetLastError();
      state[dev] = ok ? 1 : -1;
    }
    if (state[dev] != 1) { ggad_set_error(hipErrorInvalidValue, "mb_plan_build: this device cannot give k_tile_counts its LDS"); return GGAD_E_LAUNCH; }
  }
  if (P->ev_tile0) (void)hipEventRecord(static_cast<hipEvent_t>(P->ev_tile0), st);
  // (tile-major: eight residue classes of blockIdx, the tiles dealt to the classes that work, a tile's batches consecutive in its class)
  static const int tc_map = [] { const char *e = getenv("GGAD_TC_MAP"); return e ? atoi(e) : -1; }();      // (experiment: the mapping alone)
  const bool class_map = tc_map >= 0 ? tc_map != 0 : tile_major;
  const unsigned tc_grid = class_map ? 8u * (unsigned)((n_tiles + (skip < 0 ? 8 : 7) - 1) / (skip < 0 ? 8 : 7)) * (unsigned)V.n_batches
                                      : ggad_skip_grid((unsigned)n_tiles * (unsigned)V.n_batches, skip);
  k_tile_counts<<<dim3(tc_grid), dim3(TW_T), lds, st>>>(tile_major ? P->col_t : P->col, P->seg_t, V.seg_stride, P->own_rp, V.batch_ent_ptr,
                                                        P->pw_base, P->pc, n_tiles, V.n_batches, P->counters, skip, (tile_major ? 1 : 0) | (class_map ? 2 : 0));
  if (P->ev_tile1) (void)hipEventRecord(static_cast<hipEvent_t>(P->ev_tile1), st);
  if (ev0) (void)hipEventRecord(ev0, st);
  const int F = P->feat_dim;
  if (P->node_major && F <= 64) {
    // ggad_mb_plan::items holds 12 * item_cap ints: [0, 2 cap) the common work items, [2 cap, 3 cap) the big list (group ids),
    // [3 cap, 12 cap) its range boundaries (GGAD_RANGES + 1 per big group; groups <= entries <= item_cap)
    const int range_deg = ggad_int_range_deg();
    static const int affine = [] { const char *e = getenv("GGAD_RANGE_AFFINE"); return e ? atoi(e) : 1; }();
    int32_t *big = P->items + 2 * (int64_t)P->item_cap, *gbnd = P->items + 3 * (int64_t)P->item_cap;
    k_build_groups<<<dim3(ggad_skip_grid((unsigned)((V.n_ents + 255) / 256), skip)), dim3(256), 0, st>>>(
        P->ent_own, P->ent_col, P->own_deg, V.n_ents, P->node_head, P->own_next, P->tile_off, n_tiles, P->grp, P->items, big, gbnd,
        P->counters, range_deg, skip);
    // waves take ITEM_GRAB work items at a time from a cursor: enough workgroups to fill the chip, no more than there can be items
    const int64_t max_items = (int64_t)V.n_ents + (V.pair_bound > 0 ? V.pair_bound : P->pair_cap) / SLICE;      // (of this build, not of the buffers)
    // as many workgroups as are RESIDENT (4 per CU at 105 VGPRs), reservations of 12 items at most: with 2,048 workgroups the second
    // thousand only started when the first left -- at the very end -- and the reservation size followed the nominal wave count
    // (4 items at 20 batches, 32-64 at 150).  Measured (20 / 150 batches): 8 per CU, cap 64: 553 / 1,440 us; 4 per CU, cap 12: 410 / 1,390 us
    // (3 per CU 418 / 1,660; cap 32: 408 / 1,490-1,540).
    // Launches of <= 32 batches (round 4, scripts/k20_knob_sweep.sh, same box, three runs each at 20 batches): 3 per CU with cap 8
    // 428-434 us against 439-442 (4, 12); 5 / 6 per CU 482 / 520; 2 per CU 486-502; cap 4 490: the small launch wants fewer waves on
    // the memory system and smaller reservations, the 150-batch launch does not (3 per CU 1,660 us above).
    static const int env_wg = [] { const char *e = getenv("GGAD_G2_WG_PER_CU"); return e ? atoi(e) : 0; }();
    static const int env_cap = [] { const char *e = getenv("GGAD_G2_TAKE_CAP"); return e ? atoi(e) : 0; }();
    const bool small_launch = V.n_batches <= 32;
    const int wg_per_cu = env_wg > 0 ? env_wg : (small_launch ? 3 : 4);
    const int take_cap = env_cap > 0 ? env_cap : (small_launch ? 8 : 12);
    const unsigned wgs = (unsigned)std::min<int64_t>((max_items + 4 * ITEM_GRAB - 1) / (4 * ITEM_GRAB), 256 * wg_per_cu);
    // the trainer's table (rows padded to one 128-byte line): the matrix-core slice, at every launch size by default.  Measured on
    // the bench graph (plan alone, MFMA / VALU slice): 20 batches 562-571 / 559-562 us, 32: 617-645 / 639-672, 48: 750 / 814-840,
    // 64: 850 / 990, 150: 1,430 / 1,950 us.  (Before the ids of a slice were requested one work item ahead and its rows ahead of
    // its weights, the VALU slice won below 96 batches.)
    if (F == 17 && P->feat_stride == 32 && V.n_batches >= g_mfma_batches.load())
      k_gather2_items<17, true><<<dim3(ggad_skip_grid(wgs, skip)), dim3(256), 0, st>>>(P->col, P->feat, F, P->feat_stride, P->own_rp, P->own_deg,
                                                                                       P->pw_base, P->pc, P->grp, P->items, big, gbnd, P->counters,
                                                                                       P->x2, P->part2, affine, take_cap, skip);
    else if (F == 17)
      k_gather2_items<17, false><<<dim3(ggad_skip_grid(wgs, skip)), dim3(256), 0, st>>>(P->col, P->feat, F, P->feat_stride, P->own_rp, P->own_deg,
                                                                                 P->pw_base, P->pc, P->grp, P->items, big, gbnd, P->counters,
                                                                                 P->x2, P->part2, affine, take_cap, skip);
    else
      k_gather2_items<0, false><<<dim3(ggad_skip_grid(wgs, skip)), dim3(256), 0, st>>>(P->col, P->feat, F, P->feat_stride, P->own_rp, P->own_deg,
                                                                                P->pw_base, P->pc, P->grp, P->items, big, gbnd, P->counters,
                                                                                P->x2, P->part2, affine, take_cap, skip);
    const unsigned cg = (unsigned)std::min<int64_t>(((int64_t)V.n_ents + 3) / 4, 16384);        // ~2 groups per wave: every group costs a dependent load of its record
    k_gather2_combine<<<dim3(ggad_skip_grid(cg, skip)), dim3(256), 0, st>>>(P->grp, P->counters, P->part2, F, P->x2, skip);
  } else {
    k_gather2_w<<<dim3(ggad_skip_grid((unsigned)((V.n_ents + 3) / 4), skip)), dim3(256), 0, st>>>(
        P->col, P->feat, F, P->feat_stride, P->ent_own, V.n_ents, P->own_rp, P->own_deg, P->pw_base, P->pc, P->x2, skip);
  }
  if (ev1) (void)hipEventRecord(ev1, st);
  GGAD_CHECK_LAUNCH("mb_plan_build (ldsw 2-hop)");
  return GGAD_OK;
}

extern "C" {

int ggad_mb_ldsw_tile_shift(void) { return TW_SHIFT; }
int ggad_mb_ldsw_max_owners(void) { return TW_MAXOWN; }
int32_t ggad_mb_slice_len(void) { return SLICE; }
int32_t ggad_mb_group_words(void) { return GRP_W; }
int ggad_mb_set_gather_options(int32_t mfma_min_batches, int32_t range_deg) {
  if (mfma_min_batches >= 0) g_mfma_batches.store(mfma_min_batches);
  if (range_deg >= 0) g_range_deg.store(range_deg);
  return GGAD_OK;
}
int32_t ggad_mb_item_words(void) { return 3 + GGAD_RANGES + 1; }      // ints of ggad_mb_plan::items per unit of item_cap
int32_t ggad_mb_plan_counter_elems(void) { return GGAD_PLAN_COUNTERS; }
int64_t ggad_mb_tile_offsets_elems(int64_t n_nodes, int32_t tile_shift) {
  return n_nodes * (((n_nodes + (1LL << tile_shift) - 1) >> tile_shift) + 1);
}
int64_t ggad_mb_ldsw_seg_elems(int64_t n_nodes, int64_t n_entries_cap) {
  return 2 * (((n_nodes + TW_TILE - 1) >> TW_SHIFT) + 1) * ((n_entries_cap + 63) / 64 * 64);      // (second plane: tile-major segment starts)
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


int64_t ggad_mb_tile_major_workspace_elems(int64_t n_nodes, int32_t tile_shift) {
  if (n_nodes < 0 || tile_shift < 8 || tile_shift > 24) return 0;
  const int64_t n_tiles = (n_nodes + (1LL << tile_shift) - 1) >> tile_shift;
  return ((n_nodes + 255) / 256) * n_tiles + n_tiles + 1;
}

/* The tile-major copy of col and the table of its segment starts (see the kernels' header); tile_off from ggad_mb_tile_offsets,
 * tile_start: int32 x ggad_mb_tile_offsets_elems, col_t: int32 x n_edges, workspace: int32 x ggad_mb_tile_major_workspace_elems. */
int ggad_mb_tile_major(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t tile_shift, const int32_t *tile_off,
                       int32_t *tile_start, int32_t *col_t, int32_t *workspace, ggad_stream_t stream) {
  GGAD_REQUIRE(rowptr && col && tile_off && tile_start && col_t && workspace && n_nodes >= 0 && tile_shift >= 8 && tile_shift <= 24);
  if (n_nodes == 0) return GGAD_OK;
  const int n_tiles = (int)((n_nodes + (1LL << tile_shift) - 1) >> tile_shift);
  const int64_t n_blocks = (n_nodes + 255) / 256;
  GGAD_REQUIRE(n_tiles <= 16384 && n_blocks < (1LL << 31));
  hipStream_t st = as_stream(stream);
  int32_t *bsum = workspace, *tile_base = workspace + n_blocks * n_tiles;
  k_tm_blocksum<<<dim3((unsigned)n_blocks), dim3(256), 0, st>>>(tile_off, n_nodes, n_tiles, bsum);
  k_tm_scan<<<dim3(1), dim3(1024), (size_t)n_tiles * sizeof(int), st>>>(bsum, n_blocks, n_tiles, tile_base);
  k_tm_fill<<<dim3((unsigned)n_blocks), dim3(256), 0, st>>>(rowptr, col, tile_off, n_nodes, n_tiles, bsum, tile_base, tile_start, col_t);
  GGAD_CHECK_LAUNCH("mb_tile_major");
  return GGAD_OK;
}

}  // extern "C"
