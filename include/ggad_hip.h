/*
 * ggad_hip.h -- C ABI of the MI355X-native GGAD hot path (libggad_hip.so).
 *
 * The reference (mala-lab/GGAD) has no FFI / plugin layer: its boundary is the Python
 * class surface.  This header is the C-ABI that sits UNDER that surface:
 * every entry point replaces a group of ATen call sites of the reference (cited per
 * function as file:line relative to the reference tree) and is what a binding in the
 * reference would call (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless the name
 *     ends in _host; no ownership transfer: the caller (PyTorch allocator in the Python
 *     host layer) allocates every input, output and workspace;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*) and is
 *     asynchronous; functions are re-entrant per stream and keep no global mutable
 *     state: the only cross-call state is in buffers the caller passes (counter slots,
 *     optimiser state);
 *   - return value: 0 = success, negative = error (GGAD_E_*), never throws;
 *   - indices are int32, values fp32; row-major;
 *   - graph = CSR (rowptr[n+1], col[nnz]) with sorted, de-duplicated columns per row.
 */
#ifndef GGAD_HIP_H
#define GGAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGAD_OK 0
#define GGAD_E_INVALID (-1)   /* bad argument (null pointer, unsupported size)          */
#define GGAD_E_LAUNCH (-2)    /* HIP launch / runtime error, see ggad_last_error()       */
#define GGAD_E_CAPACITY (-3)  /* caller-provided workspace too small                     */

typedef void *ggad_stream_t;  /* hipStream_t */

/* ABI version of this header; bumped on any signature change. */
int ggad_abi_version(void);
/* Text of the last HIP error seen by this thread ("" if none). Host pointer, static storage. */
const char *ggad_last_error(void);
/* Upper limits compiled into the kernels (embedding width, feature width). */
int ggad_max_embed_dim(void);
int ggad_max_feat_dim(void);

/* ------------------------------------------------------------------------------------
 * Generic device primitives
 * ---------------------------------------------------------------------------------- */

/* out[0..n] = exclusive prefix sum of in[0..n-1] (out[n] = total).  n <= 4,194,304.
 * workspace: int32[ggad_scan_workspace_elems(n)]. */
int64_t ggad_scan_workspace_elems(int64_t n);
int ggad_exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, int32_t *workspace, ggad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Mini-batch path (DGraph-Fin): batch sub-graph plan + gather-aggregate
 * Replaces GCNAggregator.forward, src/graphsage.py:295-360 (python set unions, dense
 * B x U and U x U2 masks, sum/sqrt/div normalisation, Embedding gather, mask.mm).
 *
 * A "chunk" is G batches processed together; rows = all batch nodes of the chunk
 * (batch g owns rows [batch_ptr[g], batch_ptr[g+1])).  Batch g uses counter slot g:
 * slot arrays are int32[G * n_nodes], must be all-zero on entry and are all-zero again
 * after ggad_mb_plan_reset.
 *
 * Entry e (0 <= e < E) is one element j of the closed neighbourhood N(i)+{i} of row i
 * (graphsage.py:305), rows in order, columns ascending.  The "owner" entry of (batch, j)
 * is one entry of that batch with column j; 2-hop rows are stored at owner entries, so
 * the deduplicated set U of graphsage.py:306 is { e : ent_own[e] == e }.
 * ---------------------------------------------------------------------------------- */

/* row_r[i] = |N(i) + {i}|, row_slot[i] = batch (slot) of row i.          graphsage.py:305 */
int ggad_mb_row_degree(const int32_t *rowptr, const int32_t *col, const int32_t *nodes, const int32_t *batch_ptr,
                       int32_t n_batches, int32_t n_rows, int32_t *row_r, int32_t *row_slot, ggad_stream_t stream);

/* Materialise entries (ent_ptr = exclusive scan of row_r), count c_j = number of rows of the
 * batch whose closed neighbourhood holds j (column sums of the dense mask, graphsage.py:315)
 * into cnt1[slot][j], elect owners into own1[slot][j]; ent_slot[e] = slot, ent_row[e] = row.   graphsage.py:305-311 */
int ggad_mb_expand1(const int32_t *rowptr, const int32_t *col, const int32_t *nodes, const int32_t *row_slot,
                    const int32_t *ent_ptr, int32_t n_rows, int64_t n_nodes, int32_t *ent_col, int32_t *ent_slot,
                    int32_t *ent_row, int32_t *cnt1, int32_t *own1, ggad_stream_t stream);

/* ent_own[e], ent_c1[e] from the slot arrays, and the 1-hop aggregate
 * x1[i] = sum_j feat[j] / (sqrt(r_i) sqrt(c_j)).                        graphsage.py:314-326
 * feat rows are feat_stride floats apart (feat_stride == feat_dim for a plain table). */
int ggad_mb_gather1(const float *feat, int32_t feat_dim, int32_t feat_stride, const int32_t *row_slot, const int32_t *ent_ptr,
                    const int32_t *ent_col, int32_t n_rows, int64_t n_nodes, const int32_t *cnt1,
                    const int32_t *own1, int32_t *ent_own, int32_t *ent_c1, float *x1, ggad_stream_t stream);

/* out[i] = mean of feat rows over the explicit ragged list seg_col[seg_ptr[i] .. seg_ptr[i+1])
 * (MeanAggregator.forward with host-side sampling, graphsage.py:66-99). */
int ggad_seg_mean(const float *feat, int32_t feat_dim, const int32_t *seg_ptr, const int32_t *seg_col, int32_t n_rows,
                  float *out, ggad_stream_t stream);
/* Same ragged gather with explicit weights: out[i] = sum_e seg_w[e] * feat[seg_col[e]] -- the 2-hop mask of IntraAgg,
 * 1 / (sqrt(row sum) sqrt(column sum)) per element (src/layers.py:227-242). */
int ggad_seg_wsum(const float *feat, int32_t feat_dim, const int32_t *seg_ptr, const int32_t *seg_col, const float *seg_w,
                  int32_t n_rows, float *out, ggad_stream_t stream);

/* Reconstruction term of the mini-batch DOMINANT / AnomalyDAE comparison models, which run on the same 1-hop batch
 * aggregate as GGAD (src/graphsage_dominant.py:154-157,167-171; src/graphsage_anomalydae.py:154-162,172-176):
 *   loss = mean_c sqrt( sum_b w(a[b][c]) * (a[b][c] - t[b][c])^2 ),  w = w_pos where a > 0, else w_neg
 * -- the inner sum runs over the batch axis, as written there (torch.sum(diff, 0)).  a, t: (n_rows, n_cols) row-major.
 * loss: 1 float.  col_sum (n_cols floats, the sums under the root) and da (n_rows * n_cols, d loss / d a) may be NULL. */
int ggad_recon_cols_f32(const float *a, const float *t, int32_t n_rows, int32_t n_cols, float w_pos, float w_neg, float *loss,
                        float *col_sum, float *da, ggad_stream_t stream);
/* out[b] = sqrt( sum_c (a[b][c] - t[b][c])^2 ): the per-node anomaly score of test_recon (src/utils.py:158-159). */
int ggad_recon_rows_f32(const float *a, const float *t, int64_t n_rows, int32_t n_cols, float *out, ggad_stream_t stream);

/* One-class hypersphere loss of the full-graph OCGNN comparison model (ocgnn.py:83-118, :180-184) on the rows idx[0..n_idx)
 * of emb (row-major, h columns; idx NULL = all of the first n_idx rows):
 *   score[i] = ||emb[idx[i]] - center||^2 - r^2,   loss = r^2 + (1 / beta) * mean_i max(score[i], 0)
 * center NULL = the origin.  demb (may be NULL): d loss / d emb, written ONLY on the listed rows -- the caller zero-fills the
 * rest; an index listed twice gets one row's gradient, not the sum (the reference's index lists are duplicate-free). */
int ggad_ocgnn_loss_f32(const float *emb, const int64_t *idx, int64_t n_idx, int32_t h, const float *center, float r, float beta,
                        float *loss, float *score, float *demb, ggad_stream_t stream);

/* The per-entry kernels below launch one wave per entry for n_entries_cap entries (a host-side
 * upper bound, e.g. sum(deg+1)) and read the true count from *ent_total (= ent_ptr[n_rows]).
 *
 * cnt2[slot][k] += 1 for every k in N(u), u an owner entry: column sums of the U x U2 mask
 * (graphsage.py:335-348; rows are adj_list.get(u) WITHOUT self union).
 *
 * PACKED layout (cnt2 == NULL): the feature table has rows of ggad_mb_packed_stride(F) floats (128-byte
 * aligned): F features followed by one int32 counter per slot; the counter of (slot, k) is word F + slot of
 * row k.  One random 128-byte line per gathered neighbour then delivers both x_k and c'_k (measured +48 %
 * rows/s over a separate counter array, scripts/gather_bench.hip).  At most stride - F slots per chunk. */
/* Rows cut into pieces of <= ggad_mb_chunk_len() consecutive entries: row_ck_ptr[i] = first chunk of row i (n_rows + 1
 * values, exclusive scan), ck_rc[c] = (row << 6) | entries of chunk c, ck_e0[c] = its first entry.  nck_tmp: n_rows ints,
 * scan_ws: ggad_scan_workspace_elems(n_rows).  Capacity of ck_rc / ck_e0: total entries / chunk_len + n_rows. */
int32_t ggad_mb_chunk_len(void);
int ggad_mb_row_chunks(const int32_t *ent_ptr, int32_t n_rows, int32_t *nck_tmp, int32_t *row_ck_ptr, int32_t *ck_rc,
                       int32_t *ck_e0, int32_t *scan_ws, ggad_stream_t stream);
int ggad_mb_packed_stride(int32_t feat_dim);
int ggad_mb_count2(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                   const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes, const int32_t *own1,
                   int32_t *cnt2, float *feat_packed, int32_t feat_dim, int32_t feat_stride, ggad_stream_t stream);

/* 2-hop aggregate at owner entries:
 * x2[e] = sum_{k in N(u)} feat[k] / (sqrt(|N(u)|) sqrt(c'_k)).          graphsage.py:346-355
 * Dominant kernel of the path: HBM gather of feat rows, 4*F+8 algorithmic bytes per neighbour. */
int ggad_mb_gather2(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                    const int32_t *ent_col, const int32_t *ent_slot, const int32_t *ent_own, const int32_t *ent_total,
                    int64_t n_entries_cap, int64_t n_nodes, const int32_t *cnt2, float *x2, ggad_stream_t stream);

/* LDS-TILED 2-hop aggregation (hop2_tiled.hip): same result as ggad_mb_count2 + ggad_mb_gather2 without any
 * per-batch counter array in HBM.  Node ids are cut into tiles of ggad_mb_tile_size() = 65,536 ids; one
 * workgroup per batch walks the tiles, keeps c'_k of the current tile in LDS (16-bit counters, so every batch
 * must have < 65,536 owners) and reads each owner's neighbours of the tile as one contiguous piece of its CSR
 * row through tile_off[n_nodes][n_tiles + 1] (ggad_mb_tile_offsets with tile_shift 16, built once per graph;
 * int32 x ggad_mb_tile_offsets_elems(n_nodes, 16)).  x2 must be zero on entry.
 *   flags[e]     = 1 if entry e is an owner (ggad_mb_owner_flags)
 *   own_pos[]    = exclusive scan of flags (n_entries_cap + 1 values, ggad_exclusive_scan_i32)
 *   own_list[]   = workspace, n_entries_cap ints;  batch_ent_ptr[g] = first entry of batch g (n_batches + 1). */
int ggad_mb_tile_size(void);
int64_t ggad_mb_tile_offsets_elems(int64_t n_nodes, int32_t tile_shift);     /* tile = 1 << tile_shift node ids */
int ggad_mb_tile_offsets(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t tile_shift, int32_t *tile_off,
                         ggad_stream_t stream);
int ggad_mb_owner_flags(const int32_t *ent_own, const int32_t *ent_total, int64_t n_entries_cap, int32_t *flags,
                        ggad_stream_t stream);
int ggad_mb_hop2_tiled(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                       int64_t n_nodes, const int32_t *tile_off, const int32_t *flags, const int32_t *own_pos,
                       int32_t *own_list, const int32_t *batch_ent_ptr, int32_t n_batches, const int32_t *ent_col,
                       int64_t n_entries_cap, float *x2, ggad_stream_t stream);

/* "LDSW" 2-hop (hop2_tiled.hip): device-scope atomics are capped at ~27 G/s on MI355X whatever their locality
 * (scripts/atomic_bench.hip), so counting goes to LDS: one workgroup per (32,768-id tile, batch) enumerates the
 * batch's pairs of that tile pair-parallel, counts them in 16-bit LDS counters and writes every pair's final count to
 * pc[] at the pair's position in its owner's CSR row (pw_base = exclusive scan of the owners' degrees); the gather then
 * streams pc[] and makes ONE random access per neighbour (the feature row).  tile_off must be built with
 * tile_shift = ggad_mb_ldsw_tile_shift(); every batch needs <= ggad_mb_ldsw_max_owners() owners.
 * Workspaces: own_deg[n_entries_cap], own_rp[n_entries_cap], pw_base[n_entries_cap + 1],
 * scan_ws[ggad_scan_workspace_elems(n_entries_cap)], seg_t[ggad_mb_ldsw_seg_elems(n_nodes, n_entries_cap)] (the owners'
 * tile_off rows transposed tile-major, so that a (tile, batch) workgroup reads its segment bounds as contiguous runs),
 * pc[sum of the owners' degrees]. */
int ggad_mb_ldsw_tile_shift(void);
int ggad_mb_ldsw_max_owners(void);
int64_t ggad_mb_ldsw_seg_elems(int64_t n_nodes, int64_t n_entries_cap);     /* ints of seg_t */
int ggad_mb_hop2_ldsw_count(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, const int32_t *tile_off,
                            const int32_t *flags, const int32_t *own_pos, int32_t *own_list, const int32_t *batch_ent_ptr,
                            int32_t n_batches, const int32_t *ent_col, int64_t n_entries_cap, int32_t *own_deg,
                            int32_t *own_rp, int32_t *pw_base, int32_t *scan_ws, int32_t *seg_t, uint16_t *pc,
                            ggad_stream_t stream);
/* node_head (int32[n_nodes], zero on entry and again on return) + own_next (int32[n_entries_cap]) + grp
 * (int32[8 * n_entries_cap + 1]): NODE-MAJOR gather -- a node that is an owner in several batches of the chunk (hubs:
 * ~B deg / N of them) has its neighbour rows fetched once for up to 8 occurrences, each with its own streamed counts
 * and accumulator; bit-identical to the per-owner kernel that runs when the three are NULL (or feat_dim > 64). */
int ggad_mb_hop2_ldsw_gather(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                             const int32_t *own_pos, const int32_t *own_list, const int32_t *ent_col, int64_t n_entries_cap,
                             const int32_t *pw_base, const uint16_t *pc, int32_t *node_head, int32_t *own_next, int32_t *grp,
                             float *x2, ggad_stream_t stream);

/* K-TILE-MAJOR 2-hop: same tables as ggad_mb_hop2_tiled but with the per-batch counters in HBM slots
 * (cnt2[n_slots][n_nodes], zero on entry) and the WORK ordered by tile: launch t touches only the counters and
 * feature rows of ids [t*65536, (t+1)*65536), so the working set of a launch stays in L2 / Infinity Cache. */
int ggad_mb_hop2_ktile(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                       int64_t n_nodes, const int32_t *tile_off, const int32_t *flags, const int32_t *own_pos,
                       int32_t *own_list, const int32_t *ent_col, const int32_t *ent_slot, int64_t n_entries_cap,
                       int32_t *cnt2, float *x2, ggad_stream_t stream);

/* Packed layout: zero the first n_slots counters of every feature row (streaming pass). */
int ggad_mb_reset_packed(float *feat_packed, int64_t n_nodes, int32_t feat_dim, int32_t feat_stride, int32_t n_slots,
                         ggad_stream_t stream);
/* Restore the counter slots to zero by re-walking the chunk (with_hop2 = 0 for inference plans). */
int ggad_mb_plan_reset(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                       const int32_t *ent_own, const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes,
                       int32_t *cnt1, int32_t *cnt2, int32_t with_hop2, float *feat_packed, int32_t feat_dim,
                       int32_t feat_stride, ggad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Mini-batch path: dense step = GCNEncoder.forward + GCN.loss + backward + Adam
 * Replaces src/graphsage.py:395-454 (projection W, neighbour mean, outlier generation fc,
 * permuted concat), :171-258 (score, BCE, cosine-affinity margin, recon, total),
 * autograd backward and torch.optim.Adam.step (src/model_handler.py:363-364).
 *
 * Parameter block `params` (fp32): w[D] | W[D*F] | fc[D*D]  (= state_dict keys `weight`,
 * `enc.weight`, `enc.fc.weight`), followed by kernel-private transposed copies
 * Wt[F*D] | fcT[D*D]; total ggad_mb_param_block_elems(D,F).  Gradients are packed the same
 * way (first D + D*F + D*D elements): this is the buffer the data-parallel layer
 * all-reduces.
 * ---------------------------------------------------------------------------------- */
int64_t ggad_mb_param_count(int32_t D, int32_t F);       /* D + D*F + D*D               */
int64_t ggad_mb_param_block_elems(int32_t D, int32_t F); /* + transposed copies         */
/* Refresh the transposed copies after the host wrote w/W/fc (load_state_dict). */
int ggad_mb_params_sync(float *params, int32_t D, int32_t F, ggad_stream_t stream);

/* One batch = rows [row0, row0+n_rows) and entries [ent0, ent0+n_ents) of the chunk.
 *
 * ggad_mb_project : h2[e-ent0] = relu(W x2[e]) at owner entries (flat over entries).       graphsage.py:419
 * ggad_mb_fwd_rows: nbar = mean over the closed neighbourhood of h2[owner] (:421), h1 = relu(W x1) (:412)
 *                   and, on label-1 rows, the generated outlier gen = relu(fc nbar) (:428-430). */
int ggad_mb_project(const float *params, int32_t D, int32_t F, const float *x2, const int32_t *ent_own, int32_t ent0,
                    int32_t n_ents, float *h2, ggad_stream_t stream);
int ggad_mb_fwd_rows(const float *params, int32_t D, int32_t F, const float *x1, const float *h2, const int32_t *ent_ptr,
                     const int32_t *ent_own, const int32_t *labels, int32_t row0, int32_t n_rows, int32_t ent0, float *h1,
                     float *nbar, float *gen, ggad_stream_t stream);

/* Batch loss (graphsage.py:174,192-258), two launches (one wave per position, then one wave per row).
 * pos_meta[q] = (src << 2) | (src_is_label1 << 1) | label[q], src = row whose embedding sits at column q
 * of `combined_all` (label-0 rows first, generated outliers last, :450); row_pos[row] = column of that row;
 * labels are paired in ORIGINAL order (quirk 1).
 * Outputs: losses8 = {total, cls, margin, rec, 0.1/n1, margin_active, n0, n1}; the per-row backward
 * coefficients dz / coef_a / coef_g (see ggad_mb_row_coefs) of the TOTAL loss; the partial sums of d w in
 * loss_ws (float[ggad_mb_loss_workspace_elems(n_rows)], consumed by ggad_mb_grad_reduce); optionally the raw
 * gradients d_h1 / d_gen / d_nbar (all three or none).  If step_counter is not NULL it is incremented. */
int64_t ggad_mb_loss_workspace_elems(int32_t n_rows);
int ggad_mb_loss(const float *params, int32_t D, int32_t F, const float *h1, const float *nbar, const float *gen,
                 const int32_t *labels, const int32_t *pos_meta, const int32_t *row_pos, const int32_t *ent_ptr,
                 int32_t row0, int32_t n_rows, float *loss_ws, float *losses8, float *d_h1, float *d_gen, float *d_nbar,
                 float *dz, float *coef_a, float *coef_g, int32_t *step_counter, ggad_stream_t stream);

/* Vector-Jacobian product of project + fwd_rows for arbitrary upstream gradients (layered autograd API):
 *   row_coefs: coef_a = d_h1 [h1>0];  dz = d_gen [gen>0];  coef_g = (d_nbar + fc^T dz) / r
 *   bwd_flat : dW partials, flat over the batch's entries and rows, ggad_mb_bwd_parts() blocks of [F][D]. */
int ggad_mb_row_coefs(const float *params, int32_t D, int32_t F, const int32_t *labels, const int32_t *ent_ptr,
                      int32_t row0, int32_t n_rows, const float *h1, const float *gen, const float *d_h1,
                      const float *d_gen, const float *d_nbar, float *dz, float *coef_a, float *coef_g,
                      ggad_stream_t stream);
int ggad_mb_bwd_parts(void);
int ggad_mb_bwd_flat(int32_t D, int32_t F, const float *x1, const float *x2, const float *h2, const int32_t *ent_own,
                     const int32_t *ent_row, int32_t row0, int32_t n_rows, int32_t ent0, int32_t n_ents,
                     const float *coef_a, const float *coef_g, float *dw_part, ggad_stream_t stream);

/* Reduce the partials into the packed gradient buffer grads[D + D*F + D*D] (w | W | fc). */
int ggad_mb_grad_reduce(int32_t D, int32_t F, const int32_t *pos_meta, int32_t row0, int32_t n_rows,
                        const float *losses8, const float *nbar, const float *dw_part, const float *dz,
                        const float *loss_ws, float *grads, ggad_stream_t stream);

/* torch.optim.Adam.step (betas .9/.999, eps 1e-8, L2 weight decay added to the gradient) on the
 * packed block; grad_scale multiplies the gradient first (1/world_size after an all-reduce sum).
 * Step index is read from *step_counter.  Refreshes the transposed copies. */
int ggad_mb_adam(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int32_t D, int32_t F,
                 float lr, float weight_decay, float grad_scale, const int32_t *step_counter,
                 ggad_stream_t stream);

/* Whole training step of one batch in one host call.  chain 0 (default): fwd_rows_v (h2 = relu(W x2) computed by the
 * row's workgroup, F == 17) or project -> fwd_rows, then loss_pos -> loss_rows -> bwd_flat -> grad_reduce (5 or 6
 * launches); chain 2: always 6.  chain 1 (F == 17): THREE launches, one workgroup per batch row -- k_fwd_rows_x (h2
 * recomputed per entry), k_loss_bwd_rows (all positions of the batch evaluated from LDS tiles in every workgroup, then
 * the row's backward coefficients and its dW partial), k_grad_reduce; same results, measured slower (step.hip).
 * Adam is fused into the last launch when fuse_adam != 0 (single GPU); with fuse_adam == 0 the caller all-reduces
 * `grads` and then calls ggad_mb_adam.  All members are device pointers. */
typedef struct ggad_mb_step {
  float *params, *exp_avg, *exp_avg_sq, *grads;
  int32_t *step_counter;
  const float *x1, *x2;
  const int32_t *ent_ptr, *ent_own, *ent_row, *labels, *pos_meta, *row_pos;
  float *h1, *nbar, *gen, *dz, *coef_a, *coef_g;
  float *h2, *dw_part, *loss_ws, *losses8;
  int32_t D, F, row0, n_rows, ent0, n_ents;
  float lr, weight_decay;
  int32_t chain;          /* 0 (default): 5 launches when F == 17 and the batch has no hub row (projection fused into the
                             forward-rows kernel, h2 per ENTRY), else 6; 2: always 6; 1: row-wise 3-launch chain, F == 17 */
  int32_t max_row_entries; /* largest closed neighbourhood among the batch rows (host knowledge; 0 = unknown -> 6 launches) */
  /* optional (all four or none): row-chunk tables of the PLAN (ggad_mb_row_chunks over all rows of the chunk) and the
   * partial-sum buffer, float[(chunks of the plan) * 64].  With them a chain-0 batch that holds a hub row takes the
   * chunk-parallel forward (k_fwd_chunks + k_loss_pos_ck: 5 launches, h2 not stored, relu mask recomputed in bwd_flat) instead
   * of project -> fwd_rows -> loss_pos (6 launches). */
  const int32_t *row_ck_ptr, *ck_rc, *ck_e0;
  float *chunk_part;
} ggad_mb_step;
/* dw_part must hold ggad_mb_dw_part_elems(n_rows, D, F) floats (one [F][D] partial per row or per bwd_flat part). */
int64_t ggad_mb_dw_part_elems(int32_t n_rows, int32_t D, int32_t F);
int ggad_mb_train_step(const ggad_mb_step *step, int32_t fuse_adam, ggad_stream_t stream);
/* All steps of a chunk in one host call (the reference's per-batch loop, src/model_handler.py:330-364): batch b covers rows
 * [batch_ptr[b], batch_ptr[b+1]) and entries [batch_ent_ptr[b], batch_ent_ptr[b+1]) -- HOST arrays of n_batches + 1
 * offsets --, its largest row has batch_max_row[b] entries (host, may be NULL), its loss record goes to
 * loss_log + 8 * (log_base + b) (device); everything else is taken from *tmpl. */
int ggad_mb_train_chunk(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr, const int64_t *batch_ent_ptr,
                        const int32_t *batch_max_row, float *loss_log, int32_t log_base, int32_t fuse_adam,
                        ggad_stream_t stream);
/* Data-parallel form: per batch  backward -> exchange(user) -> Adam with grad_scale (1 / world size).  `exchange` is the
 * caller's all-reduce(SUM) of tmpl->grads, enqueued on `stream` (torch.distributed / RCCL); non-zero return aborts. */
int ggad_mb_train_chunk_dp(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr, const int64_t *batch_ent_ptr,
                           const int32_t *batch_max_row, float *loss_log, int32_t log_base, float grad_scale,
                           int (*exchange)(void *), void *user, ggad_stream_t stream);

/* The same per-batch loop as ggad_mb_train_chunk inside ONE persistent launch (single GPU, F == 17, at most
 * ggad_mb_persistent_max_rows() rows per batch): n_workgroups x 512 threads loop over the batches, five phases per batch
 * separated by grid barriers (a relaxed agent-scope atomic add + poll per workgroup); everything a later phase reads from
 * another wave is exchanged with write-through stores / L1-bypassing loads (step_persistent.hip).  ALL n_workgroups workgroups
 * must be resident at once: pass at most the number of compute units of `stream` (one workgroup per CU).  Rows are cut into
 * chunks of ggad_mb_persistent_chunk_len() entries; max_chunks >= the largest per-batch sum of ceil(row entries / that).
 * batch_ptr_dev / batch_ent_ptr_dev: DEVICE arrays of n_batches + 1 int32 offsets (rows, entries).  tmpl->h2, dw_part, loss_ws
 * and max_row_entries are not used (the relu mask of the 2-hop projection is recomputed in the backward phase).
 * workspace: ggad_mb_persistent_ws_elems(max_chunks, n_workgroups) floats.  Deterministic; agrees with the launch chain to
 * fp32 round-off (other, fixed summation order of the row sums and partial reductions).  Measured against the launch
 * chain on MI355X at best equal (41.9 vs 42 us per step alone, 51 vs 51-55 us in bench.py, slower on sparse graphs): an opt-in variant (chain 3 of the Python engine), not the default. */
int32_t ggad_mb_persistent_chunk_len(void);
int32_t ggad_mb_persistent_max_rows(void);
int64_t ggad_mb_persistent_ws_elems(int32_t max_chunks, int32_t n_workgroups);
int ggad_mb_train_chunk_persistent(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr_dev,
                                   const int32_t *batch_ent_ptr_dev, int32_t max_rows, int32_t max_chunks, int32_t n_workgroups,
                                   float *loss_log, int32_t log_base, float *workspace, ggad_stream_t stream);

/* Inference embeddings: h[i] = relu(W x1[i])  (GCNEncoder.forward, train_flag False).   graphsage.py:412 */
int ggad_mb_encode(const float *params, int32_t D, int32_t F, const float *x1, int32_t n_rows, float *h,
                   ggad_stream_t stream);

/* Inference: prob[i] = sigmoid(w . relu(W x1[i]))                   graphsage.py:178-181 */
int ggad_mb_score(const float *params, int32_t D, int32_t F, const float *x1, int32_t n_rows, float *prob,
                  ggad_stream_t stream);

/* Streams restricted to a subset of the compute units (bit i of mask = CU i; n_words 32-bit words).  Used to run the
 * chunk plan (chip-filling gathers) and the dense step chain (tiny dependent launches) side by side on disjoint CUs. */
int ggad_stream_create_cu_mask(const uint32_t *mask, int32_t n_words, ggad_stream_t *out);
int ggad_stream_destroy(ggad_stream_t stream);
int ggad_device_cu_count(int32_t device, int32_t *out);

/* ------------------------------------------------------------------------------------
 * Full-graph path (run.py + model.py): sparse products over the edges instead of dense N x N matrices
 * ---------------------------------------------------------------------------------- */

/* C[M x N] = epilogue(A * B), fp32 on the matrix cores (v_mfma_f32_32x32x2_f32, exact f32).
 * A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]: NN / NT / TN by strides.  Replaces nn.Linear /
 * torch.mm and their autograd (model.py:27,156,176-180).  bias[N] and relu are optional epilogues.
 * workspace: float[ggad_gemm_workspace_elems(M,N,K)] (0 = no split-K needed; may be NULL). */
int64_t ggad_gemm_workspace_elems(int32_t M, int32_t N, int32_t K);
int ggad_gemm_f32(const float *A, const float *B, float *C, int32_t M, int32_t N, int32_t K, int64_t sam, int64_t sak,
                  int64_t sbk, int64_t sbn, int64_t ldc, const float *bias, int32_t relu, float *workspace,
                  ggad_stream_t stream);

/* out[r] = act( sum_e val[e] * X[col[e]] + bias ) over CSR rows cut into SEGMENTS of <= ggad_spmm_seg_len()
 * consecutive entries [seg_beg, seg_end): one wave per segment, so a hub row never serialises the launch.
 * seg_out[s] >= 0: the row consists of this one segment and is finished in place (output row seg_out[s]);
 * seg_out[s] < 0: the partial sum goes to part[-seg_out[s] - 1][W] and the row is listed in multi_row / multi_first /
 * multi_count (output row, first slot, number of slots), summed in slot order by a second launch; part holds
 * n_seg x W floats, slots are unique and < n_seg.  The order of the segments in the tables is the launch order and is free.
 * The segment tables are built once per matrix (and per row subset) on the host.
 * act = PReLU with slope *prelu_a if given; out_pre (optional) receives the pre-activation.  W % 4 == 0.
 * Replaces torch.bmm(adj, .) + bias + PReLU (model.py:31-35), adj[0, abn, :] @ emb (model.py:151-155),
 * the column sums of sim * raw_adj (run.py:182-188, as R^T e_hat) and every transposed product in backward. */
int ggad_spmm_seg_len(void);
int ggad_spmm_csr_f32(const int32_t *col, const float *val, const int32_t *seg_beg, const int32_t *seg_end,
                      const int32_t *seg_out, int32_t n_seg, const int32_t *multi_row, const int32_t *multi_first,
                      const int32_t *multi_count, int32_t n_multi, const float *X, int64_t ldx, int32_t W, const float *bias,
                      const float *prelu_a, float *out, int64_t ldo, float *out_pre, float *part, ggad_stream_t stream);

/* Same product for dense neighbourhoods (hundreds of neighbours per row, X larger than an XCD's 4 MB L2): X is first
 * re-laid slice-major into xs_workspace (ggad_spmm_sliced_workspace_elems(n_src_rows, W) floats; column slices of 32 floats =
 * one cache line per row), then every workgroup gathers ONE slice, chosen by the XCD it runs on, 8 neighbours per load.  n_src_rows = rows of X.  Same
 * segment tables, epilogue and outputs as ggad_spmm_csr_f32; the summation order inside a segment differs (lane groups
 * take every 8th neighbour), so results agree to fp32 round-off, and are deterministic. */
int64_t ggad_spmm_sliced_workspace_elems(int64_t n_src_rows, int32_t W);
int ggad_spmm_sliced_seg_len(void); /* recommended segment length of ITS segment tables (any length is accepted; empty segments are not) */
int ggad_spmm_sliced_f32(const int32_t *col, const float *val, const int32_t *seg_beg, const int32_t *seg_end,
                         const int32_t *seg_out, int32_t n_seg, const int32_t *multi_row, const int32_t *multi_first,
                         const int32_t *multi_count, int32_t n_multi, const float *X, int64_t ldx, int32_t W, int64_t n_src_rows,
                         float *xs_workspace, const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre,
                         float *part, ggad_stream_t stream);

/* PReLU backward: dz = g * (z > 0 ? 1 : a); db[W] = column sums of dz; *da = sum g * z * [z <= 0].
 * workspace: float[2 * ggad_prelu_bwd_splits(M) * W].  db / da may be NULL. */
int32_t ggad_prelu_bwd_splits(int32_t M);
int ggad_prelu_bwd_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, float *db,
                       float *da, float *workspace, ggad_stream_t stream);
/* out = PReLU(z) with slope *prelu_a (model.py:35), for a GCN layer whose aggregate is taken from a cache (no SpMM epilogue) */
int ggad_prelu_fwd_f32(const float *z, const float *prelu_a, int64_t n, float *out, ggad_stream_t stream);
/* dz = g * [y > 0] */
int ggad_relu_bwd_f32(const float *g, const float *y, int64_t n, float *dz, ggad_stream_t stream);

/* Row L2 normalisation e_hat = e / |e| with 1/0 -> 0 (run.py:177-180) and its vector-Jacobian product. */
int ggad_rownorm_f32(const float *X, int32_t M, int32_t W, float *inv, float *Xn, ggad_stream_t stream);
int ggad_rownorm_bwd_f32(const float *Xn, const float *inv, const float *dXn, int32_t M, int32_t W, float *dX,
                         ggad_stream_t stream);
/* out[p] = scale[p] * <A[sel[p]], B[p]>   (affinity_j = r_inv_j <e_hat_j, (R^T e_hat)_j>, run.py:188) */
int ggad_rowdot_f32(const float *A, const int32_t *sel, const float *B, int32_t n, int32_t W, const float *scale, float *out,
                    ggad_stream_t stream);
/* scatter_add = 0: out[p] = coef[p] * X[sel[p]];  1: out[sel[p]] += coef[p] * X[p]  (sel must be duplicate-free) */
int ggad_rows_scale_f32(const float *X, const int32_t *sel, const float *coef, int32_t n, int32_t W, int32_t scatter_add,
                        float *out, ggad_stream_t stream);

/* The scalar part of the loss block of run.py:165-210 and its gradients: logits[L], aff[L] (L = n_normal + n_out,
 * normal_idx entries first), emb_con / emb_abn (n_out x H).  losses4 = {total, margin, bce, rec};
 * d_logits[L]; g_aff[L] = d total / d aff; dD = d total / d (emb_con - emb_abn). */
int64_t ggad_full_loss_workspace_elems(int32_t n_out, int32_t H);   /* floats of `workspace` (partial column sums of the recon term) */
int ggad_full_loss_f32(const float *logits, const float *aff, int32_t n_normal, int32_t n_out, const float *emb_con,
                       const float *emb_abn, int32_t H, float margin, float *losses4, float *d_logits, float *g_aff,
                       float *dD, float *workspace, ggad_stream_t stream);

/* torch.optim.Adam.step on a flat fp32 block; uses step index *step_counter + 1 and (bump_after != 0) advances it. */
int ggad_adam_f32(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int64_t n, float lr,
                  float weight_decay, int32_t *step_counter, int32_t bump_after, ggad_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Host-side sampler: bit-exact CPython random.shuffle (MT19937 + getrandbits rejection),
 * replaces the python shuffles inside the reference's timed loop
 * (src/model_handler.py:314,341; 28 ms / batch there).  HOST pointers.
 * ---------------------------------------------------------------------------------- */
typedef struct ggad_mt19937 ggad_mt19937;
ggad_mt19937 *ggad_mt_new(void);
void ggad_mt_free(ggad_mt19937 *);
/* random.seed(int) for 0 <= seed < 2^64 (init_by_array over the 32-bit limbs). */
int ggad_mt_seed_u64(ggad_mt19937 *, uint64_t seed);
/* state exchange with random.getstate()[1]: 624 words + index. */
int ggad_mt_set_state(ggad_mt19937 *, const uint32_t *mt624_host, int32_t index);
int ggad_mt_get_state(const ggad_mt19937 *, uint32_t *mt624_host, int32_t *index_host);
/* random.shuffle(list) in place on an int64 array. */
int ggad_mt_shuffle_i64(ggad_mt19937 *, int64_t *data_host, int64_t n);
uint32_t ggad_mt_getrandbits32(ggad_mt19937 *);
/* `count` consecutive batches of the reference's stream (src/model_handler.py:310-345: random.shuffle(train) at every
 * epoch start, random.shuffle(pool) before every batch, batch = train[i0:i1] ++ pool[:n_pseudo]) in one call, the
 * generator walk running one shuffle ahead of the swaps in a helper thread.  HOST pointers; train / pool are shuffled in
 * place exactly as the per-shuffle calls would; *in_epoch_io = index of the next batch in its epoch (>= batches_per_epoch:
 * shuffle train first).  out_nodes: count x (batch_size + n_pseudo), out_len[count]. */
int ggad_sched_batches(ggad_mt19937 *, int64_t *train, int64_t n_train, int64_t *pool, int64_t n_pool, int32_t batch_size,
                       int32_t n_pseudo, int32_t batches_per_epoch, int32_t *in_epoch_io, int32_t count, int64_t *out_nodes,
                       int32_t *out_len);

#ifdef __cplusplus
}
#endif
#endif /* GGAD_HIP_H */
