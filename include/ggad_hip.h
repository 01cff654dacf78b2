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
#define GGAD_E_UNSUPPORTED (-4) /* this entry point does not take the shape; nothing was launched (the documented fallback applies) */

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
 * slot arrays are int32[G * n_nodes], all-zero on entry and all-zero again when
 * ggad_mb_plan_build returns.
 *
 * Entry e (0 <= e < E) is one element j of the closed neighbourhood N(i)+{i} of row i
 * (graphsage.py:305), rows in order, columns ascending.  The "owner" entry of (batch, j)
 * is one entry of that batch with column j; 2-hop rows are stored at owner entries, so
 * the deduplicated set U of graphsage.py:306 is { e : ent_own[e] == e }.
 * ---------------------------------------------------------------------------------- */

/* ---- one chunk plan in ONE host call ------------------------------------------------------------------------------
 * ggad_mb_plan_build replaces, for all batches of a chunk, what GCNAggregator.forward does per batch
 * (src/graphsage.py:295-360): the closed-neighbourhood entry lists (:305-311), the column sums c_j / c'_k of the dense
 * B x U and U x U2 masks (:315,:342), the set of distinct columns U (owner entries), the 1-hop aggregate
 *   x1[i] = sum_{j in N(i)+{i}} feat[j] / (sqrt(r_i) sqrt(c_j))                                              (:314-326)
 * and (train plans) the 2-hop aggregate at owner entries
 *   x2[e] = sum_{k in N(u)} feat[k] / (sqrt(|N(u)|) sqrt(c'_k)),  u = ent_col[e]                              (:335-355).
 * Host part (no device round trip): closed degrees -> entry offsets, the pieces of <= ggad_mb_chunk_len() consecutive
 * entries every row is cut into (the unit of work of the entry-parallel kernels: a 2,000-neighbour hub row is 125
 * independent pieces, not one wave's loop), the permutation "label-0 rows first" of graphsage.py:450 (pos_meta / row_pos,
 * see ggad_mb_loss), all packed into ONE pinned staging block and uploaded with one copy.  Device part: k_expand
 * (entries, c_j histogram, owner election) -> k_gather1c (x1 partials per piece, owner metadata, pair-count storage,
 * per-node owner lists) -> k_combine1_reset (x1 of multi-piece rows; 1-hop counter slots back to zero) and, for train plans,
 * the LDS-counting 2-hop stage: k_seg_transpose -> k_tile_counts -> k_build_groups -> k_gather2_items -> k_gather2_combine
 * (hop2_ldsw.hip), or the device-atomic fallback k_count2 -> k_gather2 -> reset when a chunk exceeds the LDS path's limits
 * (>= 65,536 entries in a batch, >= 2^31 2-hop pairs).  Every floating-point sum has a fixed order; which duplicate entry of
 * (batch, column) becomes the owner is a race, owners are storage locations only (read through ent_own).
 *
 * All buffers are the caller's.  Slot arrays cnt1 / own1 (/ cnt2): int32[max_batches * n_nodes], cnt1 / cnt2 zero on entry
 * and zero again on return.  Counters: int32[8].  The staging block holds, at the element offsets returned in the info
 * struct: batch_ptr[nb+1], batch_ent_ptr[nb+1], nodes[R], labels[R], pos_meta[R], row_pos[R], row_slot[R], ent_ptr[R+1],
 * row_ck_ptr[R+1], ck_rc[C], ck_e0[C]   (ck_rc[c] = (row << 6) | entries of piece c, ck_e0[c] = its first entry). */
typedef struct ggad_mb_plan {
  /* graph on the device: CSR, feature table (rows feat_stride floats apart), ldsw tile table (ggad_mb_tile_offsets) */
  const int32_t *rowptr, *col;
  const float *feat;
  const int32_t *tile_off;
  /* graph on the host: |N(i) + {i}| per node; sum_{k in N(i)+{i}} deg(k) per node (upper bound of the 2-hop pairs) */
  const int32_t *closed_deg_host;
  const int64_t *pair_bound_host;
  /* staging: pinned host block, its device twin, an event of ggad_event_create guarding the host block */
  int32_t *stage_host, *stage;
  void *stage_event;
  /* per-batch counter slots */
  int32_t *cnt1, *own1, *cnt2;
  /* per entry (ent_cap) */
  int32_t *ent_col, *ent_slot, *ent_row, *ent_own, *ent_c1;
  float *x1, *x2;
  float *ck_part;            /* float[ck_cap * ck_part_stride]: per-piece partial sums (shared with the step kernels) */
  /* LDS-counting 2-hop stage (train plans) */
  int32_t *own_deg, *own_rp, *pw_base, *seg_t, *node_head, *own_next, *grp, *items, *counters;
  uint16_t *pc;
  float *part2;
  void *ev_gather0, *ev_gather1; /* optional events recorded around the 2-hop gather launches (roofline timing) */
  int64_t n_nodes, ent_cap, ck_cap, pair_cap, item_cap, part2_cap, stage_cap, seg_cap;
  int32_t feat_dim, feat_stride, max_batches, rows_cap, ck_part_stride;
  int32_t train;             /* 0: inference plan (1-hop only) */
  int32_t hop2;              /* 1: LDS counting ("ldsw"), 2: device atomics ("global") */
  int32_t node_major;        /* ldsw gather: occurrences of a node in the chunk share the fetch of its neighbour rows */
  float mean_nbr_deg;        /* sum deg^2 / sum deg (memset vs walk when the global counters are cleared) */
  int32_t xcd_skip;          /* -1: plain launches.  0..7: the plan kernels leave one XCD to the XCD-resident chunk kernel: the
                                workgroups with blockIdx % 8 == xcd_skip return at once (ggad_xcd_first_of_stream tells which
                                residue the stream's dispatcher puts on which XCD); results do not depend on it */
  void *ev_tile0, *ev_tile1;  /* optional events recorded around k_tile_counts (the pair counting of the LDS 2-hop stage) */
  const int64_t *node_pack_host; /* optional (ABI 6): (closed_deg_host[i] << 40) | pair_bound_host[i] per node -- the sizing pass of
                                    ggad_mb_plan_build then takes ONE cache miss per batch node instead of two (it is on the critical
                                    path of a one-chunk run).  Null: the two tables above are read. */
  const int32_t *tile_start, *col_t; /* optional (ABI 8): the tile-major copy of col and the table of its segment starts
                                        (ggad_mb_tile_major).  With both set the pair counting of the LDS 2-hop stage reads col_t -- a
                                        tile's segments contiguous, its workgroups on one XCD -- instead of 16-byte pieces of rows. */
} ggad_mb_plan;

typedef struct ggad_mb_plan_info {
  int64_t pair_bound;
  int64_t off_batch_ptr, off_batch_ent_ptr, off_nodes, off_labels, off_pos_meta, off_row_pos, off_row_slot, off_ent_ptr,
      off_row_ck_ptr, off_ck_rc, off_ck_e0;
  /* capacities this chunk needs (also filled when the call returns GGAD_E_CAPACITY: grow and call again) */
  int64_t need_rows, need_ents, need_chunks, need_pairs, need_items, need_part2, need_stage, need_seg;
  int32_t n_batches, n_rows, n_ents, n_chunks;
  int32_t mode;              /* 0 inference, 1 ldsw, 2 global */
  int32_t need_cnt2;
} ggad_mb_plan_info;

int32_t ggad_mb_chunk_len(void);          /* 16 */
int32_t ggad_mb_slice_len(void);          /* neighbours per work item of the 2-hop gather */
int32_t ggad_mb_group_words(void);        /* ints per record of grp[] */
/* Options of the node-major 2-hop gather, process-wide (a negative value leaves an option as it is):
 *   mfma_min_batches  builds of at least this many batches take the matrix-core slice when feat_dim = 17 and feat_stride = 32
 *                     (default 0 = always, GGAD_GATHER_MFMA_BATCHES; INT32_MAX = never).  The two slices add the same
 *                     products in different fixed orders: x2 agrees to ~1e-7 relative, not bit for bit.
 *   range_deg         owners with more neighbours are gathered by eighths of the id space, one per XCD (default 0 = off,
 *                     GGAD_RANGE_DEG; values below 256 mean 256).  Measured slower than slices of 256. */
int ggad_mb_set_gather_options(int32_t mfma_min_batches, int32_t range_deg);
int32_t ggad_mb_item_words(void);         /* ints of items[] per unit of item_cap: work items, the list of range-partitioned groups, their range bounds */
int32_t ggad_mb_plan_counter_elems(void); /* ints of ggad_mb_plan::counters (one counter per 64-byte line) */
/* nodes_host: the batches back to back; batch_ptr_host[nb+1]; labels_host (0/1, NULL for inference plans).  Host outputs
 * (each may be NULL): ent_ptr_host_out[R+1], batch_ent_ptr_host_out[nb+1], batch_max_row_host_out[nb]. */
int ggad_mb_plan_build(const ggad_mb_plan *plan, const int64_t *nodes_host, const int32_t *batch_ptr_host, int32_t n_batches,
                       const int64_t *labels_host, ggad_mb_plan_info *info, int64_t *ent_ptr_host_out,
                       int64_t *batch_ent_ptr_host_out, int32_t *batch_max_row_host_out, ggad_stream_t stream);

/* HIP events through the C-ABI (timing of single launches on the stream they run on; guards of pinned staging memory). */
int ggad_event_create(int32_t timing, void **out);
int ggad_event_destroy(void *event);
int ggad_event_record(void *event, ggad_stream_t stream);
int ggad_event_synchronize(void *event);
int ggad_event_elapsed_ms(void *start, void *stop, float *ms_host);

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

/* Device-atomic 2-hop stage (fallback of ggad_mb_plan_build, exported for completeness).  One wave per entry for
 * n_entries_cap entries (a host-side upper bound), true count read from *ent_total (= ent_ptr[n_rows]).
 * cnt2[slot][k] += 1 for every k in N(u), u an owner entry: column sums of the U x U2 mask
 * (graphsage.py:335-348; rows are adj_list.get(u) WITHOUT self union). */
int ggad_mb_count2(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                   const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes, const int32_t *own1,
                   int32_t *cnt2, ggad_stream_t stream);
/* x2[e] = sum_{k in N(u)} feat[k] / (sqrt(|N(u)|) sqrt(c'_k)) at owner entries.          graphsage.py:346-355 */
int ggad_mb_gather2(const int32_t *rowptr, const int32_t *col, const float *feat, int32_t feat_dim, int32_t feat_stride,
                    const int32_t *ent_col, const int32_t *ent_slot, const int32_t *ent_own, const int32_t *ent_total,
                    int64_t n_entries_cap, int64_t n_nodes, const int32_t *cnt2, float *x2, ggad_stream_t stream);
/* Restore the counter slots to zero by re-walking the chunk (with_hop2 = 0: only cnt1). */
int ggad_mb_plan_reset(const int32_t *rowptr, const int32_t *col, const int32_t *ent_col, const int32_t *ent_slot,
                       const int32_t *ent_own, const int32_t *ent_total, int64_t n_entries_cap, int64_t n_nodes,
                       int32_t *cnt1, int32_t *cnt2, int32_t with_hop2, ggad_stream_t stream);

/* Static per-node table of the LDS-counting 2-hop stage: tile_off[u][t] = offset inside u's sorted CSR row of its first
 * neighbour with id >= t << ggad_mb_ldsw_tile_shift() (t = 0 .. n_tiles): the neighbours of u inside a tile of 32,768 ids are
 * one contiguous piece of its row.  int32 x ggad_mb_tile_offsets_elems(n_nodes, shift), built once per graph. */
int ggad_mb_ldsw_tile_shift(void);
int ggad_mb_ldsw_max_owners(void);          /* entries of a batch one pass of the LDS tables holds (more: walked in slabs) */
int64_t ggad_mb_tile_offsets_elems(int64_t n_nodes, int32_t tile_shift);
int ggad_mb_tile_offsets(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t tile_shift, int32_t *tile_off,
                         ggad_stream_t stream);
int64_t ggad_mb_ldsw_seg_elems(int64_t n_nodes, int64_t n_entries_cap);     /* ints of seg_t */
/* Tile-major copy of col for the pair counting (reference op: the duplicate counts of src/graphsage.py:335-348): col_t = the ids of
 * every (node, tile) segment, all segments of a tile contiguous in node order; tile_start[u][t] (same shape as tile_off) = where
 * segment (u, t) begins in col_t.  Built once per graph from tile_off; workspace: int32 x ggad_mb_tile_major_workspace_elems. */
int64_t ggad_mb_tile_major_workspace_elems(int64_t n_nodes, int32_t tile_shift);
int ggad_mb_tile_major(const int32_t *rowptr, const int32_t *col, int64_t n_nodes, int32_t tile_shift, const int32_t *tile_off,
                       int32_t *tile_start, int32_t *col_t, int32_t *workspace, ggad_stream_t stream);

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
 * launches); chain 2: always 6.
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
                             forward-rows kernel, h2 per ENTRY), else 6; 2: always 6 */
  int32_t max_row_entries; /* largest closed neighbourhood among the batch rows (host knowledge; 0 = unknown -> 6 launches) */
  /* optional (all four or none): row-piece tables of the PLAN (staging block of ggad_mb_plan_build) and the
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

/* One-shot gradient exchange of the data-parallel step (the reference is single-device): buffers in
 * fine-grained device memory, one per rank, exported / imported as HIP IPC handles; the kernel of step s writes this rank's
 * 5,248 gradients into every rank's buffer (over xGMI), raises flags, waits for the peers', sums the W blocks in rank order and
 * applies Adam -- one launch, no collective library call, no host callback (exchange.cpp, k_xchg_adam in step.hip).
 * HOST handle; create -> exchange the ggad_xchg_handle_bytes()-byte handles (e.g. all_gather) -> connect.  A wait that times
 * out (a lost peer) sets the error word read by ggad_xchg_error instead of hanging. */
typedef struct ggad_xchg ggad_xchg;
int ggad_xchg_create(int32_t rank, int32_t world, int64_t n_floats, ggad_xchg **out);
int32_t ggad_xchg_handle_bytes(void);
int ggad_xchg_handle(ggad_xchg *xchg, void *handle_host_out);
int ggad_xchg_connect(ggad_xchg *xchg, const void *handles_host);     /* world x handle bytes, rank order */
int ggad_xchg_error(ggad_xchg *xchg, int32_t *err_host);
int ggad_xchg_destroy(ggad_xchg *xchg);
int ggad_xchg_adam(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int32_t D, int32_t F, float lr,
                   float weight_decay, float grad_scale, const int32_t *step_counter, ggad_xchg *xchg, ggad_stream_t stream);
/* ggad_mb_train_chunk with the exchange: per batch  backward -> (publish, wait, sum in rank order, Adam x grad_scale). */
int ggad_mb_train_chunk_xchg(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr, const int64_t *batch_ent_ptr,
                             const int32_t *batch_max_row, float *loss_log, int32_t log_base, float grad_scale, ggad_xchg *xchg,
                             ggad_stream_t stream);

/* XCD-resident chunk kernel (csrc/step_xcd.hip): the dense steps of a WHOLE chunk -- the reference's per-batch loop
 * src/model_handler.py:330-364 (GCNEncoder.forward + GCN.loss src/graphsage.py:395-454,171-258, backward, Adam) -- as ONE
 * launch whose workgroups all stay on one XCD (one shared L2): hand-offs between the phases of a step are plain stores + L2-served
 * loads, barriers are tagged-slot all-gathers inside that L2 (0.43 us), nothing is written back or invalidated between steps.
 * The launch has 8 n_wg workgroups (n_wg = 24, 28 or 32, 0 = 32: the compute units of one XCD the stream may use, the same number on each of its four shader engines); the dispatcher
 * deals exactly n_wg to every XCD, the ones on XCD GGAD_XCD_ID (default 0) stay, the rest leave at once.  Results are
 * deterministic and independent of n_wg's placement; if fewer than n_wg workgroups ever reached that XCD the launch times out
 * (error 2) instead of hanging.  Requirements: F == 17, D <= 64, the row-piece tables of the plan
 * (row_ck_ptr / ck_rc / ck_e0 / chunk_part of ggad_mb_step).  batch_ptr_dev: DEVICE int32[n_batches + 1] row offsets (the
 * staging block of ggad_mb_plan_build holds them); max_rows: rows of the largest batch; n_rows / n_pieces / n_entries: totals of
 * the chunk (rows_cap / pieces_cap: what the workspace was sized for); workspace:
 * float[ggad_mb_xcd_workspace_elems(max_rows, D, F, rows_cap, pieces_cap)], 16-byte aligned.  A whole-chip launch first
 * flattens the plan's tables into one 32-byte record per piece / per position (in the workspace) and copies the x2 row of every
 * owner entry to the other entries of its (batch, column) -- x2 is written, at non-owner entries only; xchg NULL = single GPU, else the one-shot exchange runs
 * inside the reduction phase of every step (grad_scale = 1 / world size).  Workgroups have 512 threads and need a compute unit
 * each: the stream must be able to keep n_wg of them resident on that XCD (ggad_mb_xcd_grid() = the largest grid, 8 x 32).
 * ggad_mb_xcd_status: control words of the last launch on `workspace` (synchronises `stream`): out[0] error (0 ok, 1 barrier
 * time-out, 2 registration time-out; GGAD_XCD_TIMEOUT_S seconds, default 10), out[1] workgroups that stayed, out[2] their XCD,
 * out[3..10] (only with GGAD_XCD_DEBUG=4, else 0) wall clocks of rank 0 in 10 ns ticks: phase A, barrier, R, barrier, C, barrier, E, barrier; out[11..18] sub-phase
 * clocks (diagnostics: they are INCLUDED in the phase that follows them). */
int32_t ggad_mb_xcd_grid(void);
/* XCD on which block 0 of a launch on `stream` runs (block b then runs on XCD (first + b) % 8; a constant of the stream's
 * hardware queue, measured by a one-wave probe launch; synchronises the stream). */
int ggad_xcd_first_of_stream(int32_t *first_host, ggad_stream_t stream);
int64_t ggad_mb_xcd_workspace_elems(int32_t max_rows, int32_t D, int32_t F, int64_t rows_cap, int64_t pieces_cap);
int ggad_mb_train_chunk_xcd(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr_dev, int32_t max_rows,
                            int32_t n_rows, int32_t n_pieces, int32_t n_ents, int64_t rows_cap, int64_t pieces_cap, float *loss_log,
                            int32_t log_base, float *workspace, float grad_scale, ggad_xchg *xchg, int32_t n_wg, const int32_t *records,
                            ggad_stream_t stream);
/* `records` NULL: the launch builds the chunk's records itself, on `stream`.  Otherwise they were built by ggad_mb_xcd_prepare on a
 * stream of the caller's choice (the plan's: a whole-chip pass that does not belong on the chunk kernel's 28 compute units) into
 * int32[ggad_mb_xcd_record_elems(rows_cap, pieces_cap)] owned by the chunk; the caller orders the two launches (stream / event). */
int64_t ggad_mb_xcd_record_elems(int64_t rows_cap, int64_t pieces_cap);
int ggad_mb_xcd_prepare(const ggad_mb_step *tmpl, int32_t n_batches, const int32_t *batch_ptr_dev, int32_t n_rows, int32_t n_pieces,
                        int32_t n_ents, int64_t rows_cap, int64_t pieces_cap, int32_t *records, ggad_stream_t stream);
int ggad_mb_xcd_status(const float *workspace, int64_t *out19, ggad_stream_t stream);
/* out19[0] of ggad_mb_xcd_status is STICKY: the control block is cleared at every launch, the first error code of any launch since the
 * last ggad_mb_xcd_clear_error(workspace, 0, stream) is not (the caller allocates the workspace zeroed).  code != 0 sets the word as a
 * launch that timed out would: the host's recovery path (ggad_amd/trainer.py: restore the entry snapshot, replay on the launch chain)
 * is tested with it.  Synchronises `stream`. */
int ggad_mb_xcd_clear_error(float *workspace, int32_t code, ggad_stream_t stream);

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

/* Same product for SPARSE neighbourhoods with a wide operand (Reddit / Photo: ~17 entries per row, W = 300; model.py:31 on those
 * configs): a workgroup owns one slice of 40 columns (8 slices at W = 300: one per XCD, whose L2 then holds its slice of X); every
 * row is finished in place by whoever owns it (no partial sums, no second launch): rows of <= ggad_spmm_rowslice_short() entries by one
 * lane group -- ggad_spmm_rowslice_group() = 6 rows per wave, unit_rows / unit_out: int32[n_units x 6] CSR rows and output rows, -1 =
 * empty slot, the caller groups rows of similar length --, rows of <= ggad_spmm_rowslice_long() entries by one wave (long_rows /
 * long_out), longer ones (hubs) by one workgroup (hub_rows / hub_out).  rowptr: the CSR row pointers (DEVICE).  Same epilogue as
 * ggad_spmm_csr_f32; results agree with it to fp32 round-off. */
int32_t ggad_spmm_rowslice_group(void);
int32_t ggad_spmm_rowslice_short(void);
int32_t ggad_spmm_rowslice_long(void);
int ggad_spmm_rowslice_f32(const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *unit_rows,
                           const int32_t *unit_out, int32_t n_units, const int32_t *long_rows, const int32_t *long_out, int32_t n_long,
                           const int32_t *hub_rows, const int32_t *hub_out, int32_t n_hub, const float *X, int64_t ldx, int32_t W,
                           const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre, ggad_stream_t stream);
/* The same product, LINE-granular and persistent, for operands whose rows are 128-byte aligned (X and ldx * 4 multiples of 128: the
 * producers of the path pad their 300-float rows to 320): a slice = one 128-byte line of a row, a wave = 8 rows, line x of all rows on
 * XCD x, the lines beyond the eighth split by rows over the XCDs, a fixed grid whose waves loop over the XCD's items (k_spmm_rowline
 * in fullgraph.hip; model.py:31 on the sparse configs).  ent: int32 x 2 = (column, bits of the fp32 value) per stored entry in CSR
 * order; n_src_rows * ldx * 4 < 2^32.  Tables of int32 x 4 = (first entry, end, output row, 0):
 * unit_tab 8 consecutive slots per unit of short rows (empty slot: 0, 0, -1; rows of <= ggad_spmm_rowslice_short() entries, longest
 * first), long_tab one per medium row (<= ggad_spmm_rowslice_long() entries), hub_tab one per longer row.  256 <= W <= 512. */
int32_t ggad_spmm_rowline_supported(const float *X, int64_t ldx, int32_t W, int64_t n_src_rows);
int ggad_spmm_rowline_f32(const int32_t *ent, const int32_t *unit_tab, int32_t n_units, const int32_t *long_tab, int32_t n_long,
                          const int32_t *hub_tab, int32_t n_hub, const float *X, int64_t ldx, int32_t W, int64_t n_src_rows,
                          const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre, ggad_stream_t stream);
/* ggad_prelu_bwd_f32 with a row stride for dz (ld_dz >= W, multiple of 4): the gradient a following line-granular product reads. */
int ggad_prelu_bwd_ld_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, int64_t ld_dz, float *db,
                          float *da, float *workspace, ggad_stream_t stream);
/* The same in ONE launch (ABI 10; autograd of `self.act(out)`, /root/reference/model.py:35): two levels of tickets -- the last workgroup
 * of every group of four adds the group's partial column sums, the last group reduces the group rows -- every sum in a fixed order.
 * `workspace`: ggad_prelu_bwd_one_workspace_elems(M, W) floats; `ticket`: ggad_prelu_bwd_one_tickets() int32 in device memory, zero
 * before the first call, left zero, not shared between launches that may run concurrently.  Shapes the vector kernel does not take run
 * the two launches. */
int64_t ggad_prelu_bwd_one_workspace_elems(int32_t M, int32_t W);
int32_t ggad_prelu_bwd_one_tickets(void);
int ggad_prelu_bwd_one_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, int64_t ld_dz, float *db,
                           float *da, float *workspace, int32_t *ticket, ggad_stream_t stream);

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

/* Same product (whole matrix, or a row subset of a matrix without separate diagonal) for dense neighbourhoods whose values factor as value[i][j] = row_scale[i] * col_scale[j]
 * off the diagonal (normalize_adj, utils.py:47-54; scales may be NULL = 1) plus an optional diagonal diag[i] (NULL = none,
 * its entries are then part of the stream): out[i] = act(row_scale[i] * sum_j col_scale[j] X[j] + diag[i] X[i] + bias).
 * A workgroup of ggad_spmm_panel_waves() waves owns (32-float column slice, row block) and walks the operand in panels of
 * ggad_spmm_panel_rows() source rows staged in LDS; the entries are a host-built stream of 16-bit panel row indices (layout at
 * k_spmm_panel in fullgraph.hip; built by ggad_amd/fullgraph.py::Csr.panel_plan): wg_tab[n_wg][2] = (slice, block) or (-1, -1),
 * dir[(block * waves + wave) * n_chunks + chunk][8] = {first oct of the wave's tiles of that panel, 8 x 16-bit counts of the QUADS
 * (4 steps) walked per round: whole octs, then half of the last one}, stream = [oct][8 lane groups][8 steps] uint16, two
 * per uint32, + 8 spare octs, row_tab[(block * waves + wave) * rounds + round][8] = output row (| GGAD_SPMM_PANEL_WIDE) or -1.  xs_workspace as for ggad_spmm_sliced_f32.
 * Deterministic; agrees with the other two kernels to fp32 round-off (the scales are applied outside the sum). */
/* Host half of its plan (csrc/spmm_panel_build.cpp, host pointers, threads; n_threads <= 0: one per core, 32 at most):
 * round_rows[n_rounds][8] = the 8 rows of a round (-1: none); ggad_spmm_panel_count -> steps_rc[round][panel] = entries of the
 * round's longest row in that panel of panel_rows columns; ggad_spmm_panel_fill writes the entry stream: tile (round, panel)
 * starts at oct tile_oct[round][panel], [oct][lane group][step] 16-bit panel row indices (ceil(steps_rc / 8) octs), empty
 * slots = a zero row (panel_rows / panel_rows + 1), spare_octs zero octs after the last tile.  skip_diag: entries col == row
 * are left out (diag[] of ggad_spmm_panel_f32 carries them).  round_wide[round] != 0 (NULL: none): the round is ONE row (slot 0 of
 * round_rows) dealt over all 8 lane groups -- even panel rows to groups 0 1 4 5, odd to 3 2 7 6 -- and its row_tab entries carry
 * GGAD_SPMM_PANEL_WIDE: the kernel adds the 8 accumulators before its epilogue. */
#define GGAD_SPMM_PANEL_WIDE 0x40000000
int ggad_spmm_panel_count(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                          const int32_t *round_wide, int32_t skip_diag, int32_t panel_rows, int32_t n_panels, int32_t *steps_rc,
                          int32_t n_threads);
int ggad_spmm_panel_fill(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                         const int32_t *round_wide, int32_t skip_diag, int32_t panel_rows, int32_t n_panels, const int32_t *steps_rc,
                         const int64_t *tile_oct, uint16_t *stream, int64_t total_octs, int32_t spare_octs, int32_t n_threads);
/* 1: every off-diagonal value[i][j] equals (float)(r[i] * r[j]) to a relative rtol; 0: not; < 0: invalid arguments. */
int ggad_spmm_panel_values_factor(const int64_t *rowptr, const int32_t *col, const float *val, const double *r, int32_t n_rows,
                                  double rtol, int32_t n_threads);
/* 1 when the current device can give a workgroup the panel kernel's 159 KB of LDS (queried and opted into once per device), else 0:
 * ggad_spmm_panel_f32 then returns an error and callers keep the sliced kernel. */
int32_t ggad_spmm_panel_available(void);
int32_t ggad_spmm_panel_rows(void);
int32_t ggad_spmm_panel_waves(void);
int32_t ggad_spmm_panel_rounds(void);
int ggad_spmm_panel_f32(const int32_t *wg_tab, int32_t n_wg, const uint32_t *dir, const uint32_t *stream, const int32_t *row_tab,
                        int32_t n_chunks, const float *col_scale, const float *row_scale, const float *diag, const float *X,
                        int64_t ldx, int32_t W, int64_t n_src_rows, float *xs_workspace, const float *bias, const float *prelu_a,
                        float *out, int64_t ldo, float *out_pre, ggad_stream_t stream_);

/* The same product (replaces torch.bmm(adj, .), model.py:31, under the conditions of ggad_spmm_panel_f32) with the operand slice
 * passing through a RING of ggad_spmm_ring_slots() LDS slots of ggad_spmm_ring_slot_rows() source rows: during phase j the
 * ggad_spmm_ring_walkers() walker waves of a workgroup read the slots j .. j + ggad_spmm_ring_window() - 1 while one more wave
 * stages slot j + slots - 1 by LDS-DMA; one barrier per phase.  The entries are a host-built per-walker list of QUADS (4 steps x 8
 * lane groups x 16-bit LDS row index; layout at k_spmm_ring in fullgraph.hip, built by ggad_amd/fullgraph.py::Csr.ring_plan):
 * wg_tab[n_wg][2] = (slice, block) or (-1, -1); wave_sb[block * walkers + wave][2] = {first super-block (4 quads), count};
 * idx = [super-block][2 halves][8 lane groups][2 quads][4 steps] uint16 (+ one spare super-block); ctl = one byte per quad:
 * bits 0..5 = 4 * accumulator slot (< ggad_spmm_ring_rounds()), bit 6 = last quad of its phase (every walker flags every phase);
 * row_tab[(block * walkers + wave) * rounds + slot][8] = output row (| GGAD_SPMM_PANEL_WIDE) or -1; n_phases = ceil(n_src_rows /
 * slot_rows).  Deterministic; agrees with the other SpMM kernels to fp32 round-off.
 * Host half (csrc/spmm_ring_build.cpp, host pointers, threads): ggad_spmm_ring_count -> quads_rp[round][phase] = quads the round
 * walks in that phase under the flexible schedule (a lane group without open entry in the slot that is overwritten next walks
 * its next entries anywhere in the window); ggad_spmm_ring_fill writes idx given the absolute first quad of every (round, phase)
 * tile (quad_off), n_sb = super-blocks of idx including the spare one.  round_rows / round_wide / skip_diag as for the panel plan. */
int ggad_spmm_ring_count(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                         const int32_t *round_wide, int32_t skip_diag, int32_t slot_rows, int32_t n_ring_slots, int32_t window,
                         int32_t n_phases, uint16_t *quads_rp, int32_t n_threads);
int ggad_spmm_ring_fill(const int64_t *rowptr, const int32_t *col, int32_t n_rounds, const int32_t *round_rows,
                        const int32_t *round_wide, int32_t skip_diag, int32_t slot_rows, int32_t n_ring_slots, int32_t window,
                        int32_t n_phases, const uint16_t *quads_rp, const int64_t *quad_off, uint16_t *idx, int64_t n_sb,
                        int32_t n_threads);
int32_t ggad_spmm_ring_available(void);
int32_t ggad_spmm_ring_slot_rows(void);
int32_t ggad_spmm_ring_slots(void);
int32_t ggad_spmm_ring_window(void);
int32_t ggad_spmm_ring_walkers(void);
int32_t ggad_spmm_ring_rounds(void);
/* n_walkers (ABI 10): the walker waves per workgroup the plan was dealt to -- ggad_spmm_ring_walkers() (one loader wave: the whole-matrix
 * products) or ggad_spmm_ring_walkers_subset() (three loader waves: products whose walkers have a handful of steps per phase and would wait
 * for a single, issue-bound loader -- the loss-row products of run.py:182-188 and their transposes). */
int32_t ggad_spmm_ring_walkers_subset(void);
int ggad_spmm_ring_f32(const int32_t *wg_tab, int32_t n_wg, const int32_t *wave_sb, const uint16_t *idx, const uint32_t *ctl,
                       const int32_t *row_tab, int32_t n_phases, int32_t n_walkers, const float *col_scale, const float *row_scale,
                       const float *diag, const float *X, int64_t ldx, int32_t W, int64_t n_src_rows, float *xs_workspace,
                       const float *bias, const float *prelu_a, float *out, int64_t ldo, float *out_pre, ggad_stream_t stream_);

/* Fused scorer MLP (model.py:176-180: f_1 = relu(fc1 x), f_2 = relu(fc2 f_1), f_3 = fc3 f_2; Linear weights [out][in], no bias) on
 * the exact-f32 matrix cores, one launch each way instead of three GEMMs forward and five launches for the data gradients backward
 * (SURVEY.md 2.1 F8 / 8b `mlp_score_f32`).  X: R x H (row stride ldx, multiple of 4, 16-byte aligned); W1: H1 x H, W2: H2 x H1,
 * w3: H2.  fwd writes f1 (R x H1), f2 (R x H2) -- the backward's operands -- and f3 (R).  dgrad: dz2 = (g3 w3^T) [f2 > 0] (R x H2),
 * dz1 = (dz2 W2) [f1 > 0] (R x H1), dx = dz1 W1 (+ dx_add, e.g. another gradient of x; NULL = none), row strides ld_dx / ld_add; the
 * weight gradients dz1^T x, dz2^T f1, g3^T f2 stay ggad_gemm_f32 calls (split over the R rows).  ggad_mlp_score_supported: H <= 512
 * and a multiple of 4, H1 <= 256 and even, H2 <= 128 (else callers take the three GEMMs).  Sums run over k in blocks of 16 with
 * a fixed lane-to-k map: deterministic, equal to a k-ordered product to fp32 round-off. */
int32_t ggad_mlp_score_supported(int32_t H, int32_t H1, int32_t H2);
int ggad_mlp_score_fwd_f32(const float *X, int64_t ldx, int32_t R, int32_t H, int32_t H1, int32_t H2, const float *W1, const float *W2,
                           const float *w3, float *f1, float *f2, float *f3, ggad_stream_t stream);
/* The three WEIGHT gradients of the scorer in one launch + one reduction (ABI 7; they were three split-K ggad_gemm_f32 calls):
 *   dW1 (H1 x H) = dz1^T x,  dW2 (H2 x H1) = dz2^T f1,  dW3 (1 x H2) = g3^T f2      over the same R rows.
 * x: R x H with row stride ldx; dz1, f1: R x H1; dz2, f2: R x H2; g3: R -- all contiguous but x.  Outputs contiguous.  workspace:
 * float[ggad_mlp_score_wgrad_workspace_elems(R, H, H1, H2)], 16-byte aligned (partials per row range, added in range order). */
int64_t ggad_mlp_score_wgrad_workspace_elems(int32_t R, int32_t H, int32_t H1, int32_t H2);
int ggad_mlp_score_wgrad_f32(const float *x, int64_t ldx, const float *dz1, const float *f1, const float *dz2, const float *f2, const float *g3,
                             int32_t R, int32_t H, int32_t H1, int32_t H2, float *dW1, float *dW2, float *dW3, float *workspace,
                             ggad_stream_t stream);
int ggad_mlp_score_dgrad_f32(const float *g3, int32_t R, int32_t H, int32_t H1, int32_t H2, const float *f1, const float *f2,
                             const float *W1, const float *W2, const float *w3, float *dz2, float *dz1, float *dx, int64_t ld_dx,
                             const float *dx_add, int64_t ld_add, ggad_stream_t stream);

/* PReLU backward: dz = g * (z > 0 ? 1 : a); db[W] = column sums of dz; *da = sum g * z * [z <= 0].
 * workspace: float[2 * ggad_prelu_bwd_splits(M) * W].  db / da may be NULL. */
int32_t ggad_prelu_bwd_splits(int32_t M);
int ggad_prelu_bwd_f32(const float *g, const float *z, const float *prelu_a, int32_t M, int32_t W, float *dz, float *db,
                       float *da, float *workspace, ggad_stream_t stream);
/* out = PReLU(z) with slope *prelu_a (model.py:35), for a GCN layer whose aggregate is taken from a cache (no SpMM epilogue) */
int ggad_prelu_fwd_f32(const float *z, const float *prelu_a, int64_t n, float *out, ggad_stream_t stream);
/* z = A W^T + bias and out = PReLU(z) in ONE launch (reference model.py:27-35 with A_hat X cached: nn.Linear, bias, PReLU) when the
 * slab GEMM takes the shape (M >= 4096; K % 4 == 0 in 17..32, 49..64 or 241..320; N % 4 == 0; 16-byte aligned rows): GGAD_OK, or
 * GGAD_E_UNSUPPORTED with nothing launched (run ggad_gemm_f32 + ggad_prelu_fwd_f32 instead).  A: M x K (rows lda floats apart),
 * W: N x K (rows ldw apart), z / out: M x N (rows ldz / ldo apart). */
int ggad_linear_prelu_f32(const float *A, int64_t lda, const float *W, int64_t ldw, const float *bias, const float *prelu_a, int32_t M,
                          int32_t N, int32_t K, float *z, int64_t ldz, float *out, int64_t ldo, ggad_stream_t stream);
/* dz = g * [y > 0] */
int ggad_relu_bwd_f32(const float *g, const float *y, int64_t n, float *dz, ggad_stream_t stream);

/* The head of the training forward from `emb` (N x W) on -- model.py:140-182 -- and its backward, the glue between the products:
 *   head_gather    out[p] = X[idx[p]] (+ add[p])                          emb[abn] + noise                       model.py:141-145
 *   head_combine   out = [X[nrm]; con]                                    torch.cat((emb[normal], emb_con))      :159
 *   head_emb_out   out[i] = abn_pos[i] >= 0 ? con[abn_pos[i]] : X[i]      emb[:, abn, :] = emb_con               :182
 *   head_con_grad  dz = [y > 0] (g_con + g_out[abn] + g_tail)             every gradient that reaches emb_con = relu(fc4(.)), :156
 *   head_emb_grad  d emb[i] = [i not in abn] g_out[i] + [i in normal] g_comb[nrm_pos[i]] + [i in abn] g_abn[abn_pos[i]] + sp[i]
 * abn_pos / nrm_pos: position of node i in the (duplicate-free) index list, -1 if absent.  Gradient terms may be null. */
int ggad_head_gather_f32(const float *X, const int32_t *idx, const float *add, int32_t n, int32_t W, float *out, ggad_stream_t stream);
int ggad_head_combine_f32(const float *X, const int32_t *nrm, int32_t n_nrm, const float *con, int32_t n_con, int32_t W, float *out,
                          ggad_stream_t stream);
/* ABI 10: ggad_head_gather_f32 and the emb[normal] rows of ggad_head_combine_f32 in one launch (model.py:141-145,159: both read rows of
 * emb before anything else of the head runs); and emb[:, abn, :] = emb_con IN PLACE as the reference writes it (model.py:182; abn
 * duplicate-free) instead of a second N x H tensor. */
int ggad_head_rows_f32(const float *X, const int32_t *abn, const float *add, int32_t n_abn, const int32_t *nrm, int32_t n_nrm, int32_t W,
                       float *out_abn, float *out_comb, ggad_stream_t stream);
int ggad_head_emb_put_f32(const float *con, const int32_t *abn, int32_t n_abn, int32_t W, float *X, ggad_stream_t stream);
int ggad_head_emb_out_f32(const float *X, const int32_t *abn_pos, const float *con, int32_t n, int32_t W, float *out,
                          ggad_stream_t stream);
int ggad_head_con_grad_f32(const float *g_con, const float *g_out, const int32_t *abn, const float *g_tail, const float *y, int32_t n,
                           int32_t W, float *dz, ggad_stream_t stream);
int ggad_head_emb_grad_f32(const float *g_out, const int32_t *abn_pos, const int32_t *nrm_pos, const float *g_comb, const float *g_abn,
                           const float *sp, int32_t n, int32_t W, float *out, ggad_stream_t stream);

/* Row L2 normalisation e_hat = e / |e| with 1/0 -> 0 (run.py:177-180) and its vector-Jacobian product. */
int ggad_rownorm_f32(const float *X, int32_t M, int32_t W, float *inv, float *Xn, ggad_stream_t stream);
int ggad_rownorm_bwd_f32(const float *Xn, const float *inv, const float *dXn, int32_t M, int32_t W, float *dX,
                         ggad_stream_t stream);
/* out[e] = || X[row(e)] - X[col[e]] ||_2 per stored entry of a CSR matrix (TAM calc_distance, utils_tam.py:190-199) */
int ggad_edge_dist_f32(const int32_t *rowptr, const int32_t *col, const float *X, int32_t n_rows, int32_t W, float *out,
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
/* Backward of the loss block, every product with the incoming d total (a device scalar) in one launch:
 *   c = g_aff r_inv_J g (what ggad_rows_scale_f32 needs next), d logits = d_logits g, d emb_con = dD g, d emb_abnormal = -dD g. */
int ggad_full_loss_bwd_scale_f32(const float *g_total, const float *g_aff, const float *r_inv_j, const float *d_logits, const float *dD,
                                 int32_t L, int64_t n_rec, float *c, float *dl, float *d_con, float *d_abn, ggad_stream_t stream);

/* Round 6 (ABI 10): the same loss block (run.py:165-210, reference file /root/reference/run.py) with its row-local parts fused --
 * forward: affinity row dots r_inv_J <e_hat[J], S>, the partial column sums of the recon term and, in the last workgroup to finish, BCE,
 * margin, recon, the four loss values and the coefficients (replaces ggad_rowdot_f32 + ggad_full_loss_f32: four launches);
 * backward: c, d logits, xc = c e_hat[J], d emb_con = g (con - abn) kcol, d emb_abnormal = -that (replaces ggad_full_loss_bwd_scale_f32 +
 * ggad_rows_scale_f32);  ggad_rownorm_bwd_add_f32 = ggad_rownorm_bwd_f32 with dXn[r] += c[q] S[q] for r = J[q] folded in (pos_n / pos_a:
 * position of row r in the normal / abnormal segment of J, or -1).  `ticket`: one int32, zero before the first call (left zero). */
int64_t ggad_full_loss_fused_workspace_elems(int32_t n_out, int32_t H);
int ggad_full_loss_fused_f32(const float *e_hat, const int32_t *J, const float *S, const float *r_inv_j, int32_t n_normal, int32_t n_out,
                             int32_t H, const float *logits, const float *emb_con, const float *emb_abn, float margin, float *aff,
                             float *kcol, float *losses4, float *d_logits, float *g_aff, float *workspace, int32_t *ticket,
                             ggad_stream_t stream);
int ggad_full_loss_bwd_fused_f32(const float *g_total, const float *g_aff, const float *r_inv_j, const float *d_logits, const float *e_hat,
                                 const int32_t *J, int32_t L, int32_t H, const float *emb_con, const float *emb_abn, const float *kcol,
                                 int32_t n_out, float *c, float *dl, float *xc, float *d_con, float *d_abn, ggad_stream_t stream);
int ggad_rownorm_bwd_add_f32(const float *Xn, const float *inv, const float *dXn, const int32_t *pos_n, const int32_t *pos_a,
                             const float *c, const float *S, int32_t M, int32_t W, float *dX, ggad_stream_t stream);

/* torch.optim.Adam.step on a flat fp32 block; uses step index *step_counter + 1 and (bump_after != 0) advances it. */
int ggad_adam_f32(float *params, float *exp_avg, float *exp_avg_sq, const float *grads, int64_t n, float lr,
                  float weight_decay, int32_t *step_counter, int32_t bump_after, ggad_stream_t stream);
/* The same update (bump_after = 1) for up to ggad_adam_multi_max() tensors: HOST arrays of device pointers / element counts; one step
 * counter per tensor (torch keeps `step` per parameter; parameters without a gradient are left out).  `tickets` (ABI 10; optional):
 * ggad_adam_multi_max() int32 in device memory, zero before the first call and left zero -- the step counters then advance inside the
 * one launch (the last workgroup of a tensor does it); NULL: a second, trailing launch advances them. */
int32_t ggad_adam_multi_max(void);
int ggad_adam_multi_f32(int32_t n_tensors, float *const *params, float *const *exp_avg, float *const *exp_avg_sq,
                        const float *const *grads, const int64_t *n_elems, int32_t *const *step_counters, float lr,
                        float weight_decay, int32_t *tickets, ggad_stream_t stream);

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
/* The two halves of random.shuffle for callers that pipeline them: (1) consume the generator -> targets_out[c] = swap partner
 * of position n - 1 - c (int32[n + 16], data-independent); (2) apply recorded swaps to a list. */
int ggad_mt_shuffle_targets(ggad_mt19937 *, int64_t n, int32_t *targets_host_out);
int ggad_apply_swaps_i64(int64_t *data_host, int64_t n, const int32_t *targets_host);
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
