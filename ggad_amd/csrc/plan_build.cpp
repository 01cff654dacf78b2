// ggad_mb_plan_build: the host half of the chunk plan (include/ggad_hip.h) and the HIP-event helpers of the C-ABI.
//
// The reference builds, per batch, python sets and dense masks (src/graphsage.py:295-360).  Here the host only does what
// needs no device data: closed degrees (a static per-node table) -> entry offsets; the pieces of <= 16 consecutive entries
// every row is cut into; the "label-0 rows first, generated outliers last" permutation of graphsage.py:450.  Everything
// goes to the device in ONE pinned block with ONE copy, followed by 8 kernel launches (3 for an inference plan) -- the
// previous version of this path issued 28 launches and ~0.25 ms of Python per chunk, which bounded short runs.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace {
inline int64_t align4(int64_t x) { return (x + 3) & ~(int64_t)3; }
}

extern "C" {

int ggad_event_create(int32_t timing, void **out) {
  if (!out) return GGAD_E_INVALID;
  hipEvent_t ev = nullptr;
  const hipError_t e = hipEventCreateWithFlags(&ev, timing ? hipEventDefault : hipEventDisableTiming);
  if (e != hipSuccess) { ggad_set_error(e, "event_create"); return GGAD_E_LAUNCH; }
  *out = ev;
  return GGAD_OK;
}
int ggad_event_destroy(void *event) {
  if (!event) return GGAD_OK;
  const hipError_t e = hipEventDestroy(static_cast<hipEvent_t>(event));
  if (e != hipSuccess) { ggad_set_error(e, "event_destroy"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}
int ggad_event_record(void *event, ggad_stream_t stream) {
  if (!event) return GGAD_E_INVALID;
  const hipError_t e = hipEventRecord(static_cast<hipEvent_t>(event), as_stream(stream));
  if (e != hipSuccess) { ggad_set_error(e, "event_record"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}
int ggad_event_synchronize(void *event) {
  if (!event) return GGAD_E_INVALID;
  const hipError_t e = hipEventSynchronize(static_cast<hipEvent_t>(event));
  if (e != hipSuccess) { ggad_set_error(e, "event_synchronize"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}
int ggad_event_elapsed_ms(void *start, void *stop, float *ms_host) {
  if (!start || !stop || !ms_host) return GGAD_E_INVALID;
  const hipError_t e = hipEventElapsedTime(ms_host, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
  if (e != hipSuccess) { ggad_set_error(e, "event_elapsed_ms"); return GGAD_E_LAUNCH; }
  return GGAD_OK;
}

int ggad_mb_plan_build(const ggad_mb_plan *P, const int64_t *nodes_host, const int32_t *batch_ptr_host, int32_t n_batches,
                       const int64_t *labels_host, ggad_mb_plan_info *info, int64_t *ent_ptr_host_out,
                       int64_t *batch_ent_ptr_host_out, int32_t *batch_max_row_host_out, ggad_stream_t stream) {
  GGAD_REQUIRE(P && nodes_host && batch_ptr_host && info);
  GGAD_REQUIRE(P->rowptr && P->col && P->feat && P->closed_deg_host && P->stage_host && P->stage && P->stage_event);
  GGAD_REQUIRE(P->cnt1 && P->own1 && P->ent_col && P->ent_slot && P->ent_row && P->ent_own && P->ent_c1 && P->x1 && P->ck_part);
  GGAD_REQUIRE(P->feat_dim >= 1 && P->feat_dim <= GGAD_MAX_F && P->feat_stride >= P->feat_dim && P->n_nodes >= 1);
  GGAD_REQUIRE(P->ck_part_stride >= (P->feat_dim < 64 ? 64 : P->feat_dim));
  GGAD_REQUIRE(n_batches >= 1 && n_batches <= P->max_batches && batch_ptr_host[0] == 0);
  const bool train = P->train != 0;
  if (train) GGAD_REQUIRE(labels_host && P->x2 && (P->hop2 == 1 || P->hop2 == 2));
  std::memset(info, 0, sizeof(*info));
  static const bool timing = getenv("GGAD_BUILD_TIMING") != nullptr;      // host phases of this call on stderr (us)
  const auto tt0 = std::chrono::steady_clock::now();
  const int nb = n_batches;
  const int64_t rows = batch_ptr_host[nb];
  GGAD_REQUIRE(rows >= 1 && rows < (1 << 25));
  for (int b = 0; b < nb; ++b) GGAD_REQUIRE(batch_ptr_host[b + 1] > batch_ptr_host[b]);     // no empty batch
  const int CL = 16;
  const int SL = ggad_mb_slice_len();

  // ---- sizes (host knowledge only)
  int64_t n_ents = 0, n_chunks = 0, bound = 0;
  int64_t max_batch_ents = 0;
  const bool want_ldsw = train && P->hop2 == 1;
  if (want_ldsw) GGAD_REQUIRE(P->pair_bound_host && P->tile_off && P->own_deg && P->own_rp && P->pw_base && P->node_head &&
                              P->own_next && P->grp && P->counters);
  // the two static per-node tables are read at random (one cache miss per batch node each): request them ahead of the walk and
  // keep the closed degree of every row for the second pass (this loop is on the critical path of a run of ONE chunk)
  static thread_local std::vector<int32_t> row_deg;
  if ((int64_t)row_deg.size() < rows) row_deg.resize((size_t)rows);
  constexpr int PF = 24;
  const int64_t *pack = P->node_pack_host;                 // (degree << 40) | pair bound: one miss per node instead of two
  constexpr int64_t PACK_MASK = (1LL << 40) - 1;
  for (int64_t i = 0; i < rows && i < PF; ++i) {
    const int64_t v = nodes_host[i];
    GGAD_REQUIRE(v >= 0 && v < P->n_nodes);
    if (pack) { __builtin_prefetch(&pack[v]); continue; }
    __builtin_prefetch(&P->closed_deg_host[v]);
    if (want_ldsw) __builtin_prefetch(&P->pair_bound_host[v]);
  }
  for (int b = 0; b < nb; ++b) {
    int64_t be = 0;
    for (int64_t i = batch_ptr_host[b]; i < batch_ptr_host[b + 1]; ++i) {
      if (i + PF < rows) {
        const int64_t vn = nodes_host[i + PF];
        GGAD_REQUIRE(vn >= 0 && vn < P->n_nodes);
        if (pack) __builtin_prefetch(&pack[vn]);
        else {
          __builtin_prefetch(&P->closed_deg_host[vn]);
          if (want_ldsw) __builtin_prefetch(&P->pair_bound_host[vn]);
        }
      }
      const int64_t v = nodes_host[i];
      const int64_t pk = pack ? pack[v] : 0;
      const int64_t r = pack ? (pk >> 40) : P->closed_deg_host[v];
      row_deg[(size_t)i] = (int32_t)r;
      be += r;
      n_chunks += (r + CL - 1) / CL;
      if (want_ldsw) bound += pack ? (pk & PACK_MASK) : P->pair_bound_host[v];
    }
    n_ents += be;
    max_batch_ents = std::max(max_batch_ents, be);
  }
  GGAD_REQUIRE(n_ents < (1LL << 31) - 1);
  int mode = train ? P->hop2 : 0;
  // LDS path: 16-bit LDS counters, int32 offsets into pc[], 32-bit byte offsets into the feature table
  if (mode == 1 && (max_batch_ents >= 65536 || bound >= (1LL << 31) - 1 || P->n_nodes * (int64_t)P->feat_stride * 4 >= (1LL << 32)))
    mode = 2;
  const int64_t off_bp = 0, off_bep = align4(nb + 1), off_nodes = off_bep + align4(nb + 1), off_labels = off_nodes + align4(rows),
                off_meta = off_labels + align4(rows), off_rpos = off_meta + align4(rows), off_slot = off_rpos + align4(rows),
                off_eptr = off_slot + align4(rows), off_ckp = off_eptr + align4(rows + 1), off_ckrc = off_ckp + align4(rows + 1),
                off_cke0 = off_ckrc + align4(n_chunks), stage_need = off_cke0 + align4(n_chunks);
  const int64_t seg_stride = (n_ents + 63) / 64 * 64;
  const int64_t n_tiles1 = ((P->n_nodes + (1LL << ggad_mb_ldsw_tile_shift()) - 1) >> ggad_mb_ldsw_tile_shift()) + 1;
  info->pair_bound = bound;
  info->n_batches = nb;
  info->n_rows = (int32_t)rows;
  info->n_ents = (int32_t)n_ents;
  info->n_chunks = (int32_t)n_chunks;
  info->mode = mode;
  info->off_batch_ptr = off_bp; info->off_batch_ent_ptr = off_bep; info->off_nodes = off_nodes; info->off_labels = off_labels;
  info->off_pos_meta = off_meta; info->off_row_pos = off_rpos; info->off_row_slot = off_slot; info->off_ent_ptr = off_eptr;
  info->off_row_ck_ptr = off_ckp; info->off_ck_rc = off_ckrc; info->off_ck_e0 = off_cke0;
  info->need_rows = rows; info->need_ents = n_ents; info->need_chunks = n_chunks; info->need_stage = stage_need;
  if (mode == 1) {
    info->need_pairs = bound;
    info->need_items = n_ents + bound / SL;
    // slots of feat_dim floats: ceil(deg / SL) <= 2 deg / SL per occurrence of an owner with > SL neighbours, GGAD_RANGES per
    // occurrence of one with > GGAD_RANGE_DEG
    info->need_part2 = (2 + (ggad_int_range_deg() == INT32_MAX ? 0 : (int64_t)GGAD_RANGES * SL / GGAD_RANGE_DEG)) * (bound / SL) + 8;
    // (a second plane -- where the segments begin in the tile-major copy of col -- only when that opt-in copy is handed in: it was
    //  allocated unconditionally, hundreds of MB at DGraph size for a plane nobody wrote: ADVICE round 5)
    info->need_seg = ((P->tile_start && P->col_t) ? 2 : 1) * n_tiles1 * seg_stride;
  }
  info->need_cnt2 = (mode == 2 && P->cnt2 == nullptr) ? 1 : 0;
  if (rows > P->rows_cap || n_ents > P->ent_cap || n_chunks > P->ck_cap || stage_need > P->stage_cap || info->need_cnt2 ||
      (mode == 1 && (info->need_pairs > P->pair_cap || info->need_items > P->item_cap || info->need_part2 > P->part2_cap ||
                     info->need_seg > P->seg_cap)))
    return GGAD_E_CAPACITY;
  if (mode == 1) GGAD_REQUIRE(P->seg_t && P->pc && P->items && P->part2);        // sized by the capacities checked above

  const auto tt1 = std::chrono::steady_clock::now();
  // ---- staging block
  hipStream_t st = as_stream(stream);
  hipError_t he = hipEventSynchronize(static_cast<hipEvent_t>(P->stage_event));      // the previous upload has left the block
  if (he != hipSuccess) { ggad_set_error(he, "mb_plan_build (stage event)"); return GGAD_E_LAUNCH; }
  const auto tt2 = std::chrono::steady_clock::now();
  int32_t *S = P->stage_host;
  int32_t *bp = S + off_bp, *bep = S + off_bep, *nd = S + off_nodes, *lb = S + off_labels, *meta = S + off_meta,
          *rpos = S + off_rpos, *slot = S + off_slot, *eptr = S + off_eptr, *ckp = S + off_ckp, *ckrc = S + off_ckrc,
          *cke0 = S + off_cke0;
  int64_t e = 0, c = 0;
  for (int b = 0; b < nb; ++b) {
    const int r0 = batch_ptr_host[b], r1 = batch_ptr_host[b + 1];
    bp[b] = r0;
    bep[b] = (int32_t)e;
    if (batch_ent_ptr_host_out) batch_ent_ptr_host_out[b] = e;
    int max_r = 0;
    for (int i = r0; i < r1; ++i) {
      const int32_t v = (int32_t)nodes_host[i];
      const int r = row_deg[(size_t)i];
      nd[i] = v;
      slot[i] = b;
      eptr[i] = (int32_t)e;
      ckp[i] = (int32_t)c;
      if (ent_ptr_host_out) ent_ptr_host_out[i] = e;
      for (int k = 0; k < r; k += CL) {
        ckrc[c] = (i << 6) | std::min(CL, r - k);
        cke0[c] = (int32_t)(e + k);
        ++c;
      }
      e += r;
      max_r = std::max(max_r, r);
    }
    if (batch_max_row_host_out) batch_max_row_host_out[b] = max_r;
    if (train) {
      // column q of `combined_all` holds the label-0 rows in order, then the label-1 rows (graphsage.py:450);
      // pos_meta[q] = (src_row << 2) | (label[src] << 1) | label[q];  row_pos[row] = column of that row (batch-relative)
      int q = 0;
      for (int pass = 0; pass < 2; ++pass)
        for (int i = r0; i < r1; ++i) {
          const int64_t l = labels_host[i];
          if (l != 0 && l != 1) return GGAD_E_INVALID;
          if ((int)l != pass) continue;
          meta[r0 + q] = (i << 2) | ((int)l << 1);
          rpos[i] = q;
          ++q;
        }
      for (int i = r0; i < r1; ++i) {
        lb[i] = (int32_t)labels_host[i];
        meta[i] |= (int32_t)labels_host[i];
      }
    }
  }
  bp[nb] = (int32_t)rows;
  bep[nb] = (int32_t)e;
  eptr[rows] = (int32_t)e;
  ckp[rows] = (int32_t)c;
  if (batch_ent_ptr_host_out) batch_ent_ptr_host_out[nb] = e;
  if (ent_ptr_host_out) ent_ptr_host_out[rows] = e;
  const auto tt3 = std::chrono::steady_clock::now();
  he = hipMemcpyAsync(P->stage, S, (size_t)stage_need * sizeof(int32_t), hipMemcpyHostToDevice, st);
  if (he != hipSuccess) { ggad_set_error(he, "mb_plan_build (upload)"); return GGAD_E_LAUNCH; }
  he = hipEventRecord(static_cast<hipEvent_t>(P->stage_event), st);
  if (he != hipSuccess) { ggad_set_error(he, "mb_plan_build (stage event record)"); return GGAD_E_LAUNCH; }

  const auto tt4 = std::chrono::steady_clock::now();
  // ---- device part
  ggad_plan_view V;
  const int32_t *D = P->stage;
  V.batch_ptr = D + off_bp; V.batch_ent_ptr = D + off_bep; V.nodes = D + off_nodes; V.row_slot = D + off_slot;
  V.ent_ptr = D + off_eptr; V.row_ck_ptr = D + off_ckp; V.ck_rc = D + off_ckrc; V.ck_e0 = D + off_cke0;
  V.n_batches = nb; V.n_rows = (int32_t)rows; V.n_ents = (int32_t)n_ents; V.n_chunks = (int32_t)n_chunks;
  V.seg_stride = (int32_t)seg_stride;
  V.pair_bound = bound;
  int rc = ggad_int_hop1(P, V, mode == 1 ? 1 : 0, mode != 2 ? 1 : 0, st);
  if (rc) return rc;
  hipEvent_t ev0 = static_cast<hipEvent_t>(P->ev_gather0), ev1 = static_cast<hipEvent_t>(P->ev_gather1);
  if (mode == 1) rc = ggad_int_ldsw_hop2(P, V, st, ev0, ev1);
  else if (mode == 2) rc = ggad_int_global_hop2(P, V, st, ev0, ev1);
  if (timing) {
    const auto tt5 = std::chrono::steady_clock::now();
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "[ggad_mb_plan_build] %d batches: sizes %.1f  stage-event wait %.1f  fill %.1f  copy + event %.1f  launches %.1f us (%lld KB staged)\n", nb,
            us(tt0, tt1), us(tt1, tt2), us(tt2, tt3), us(tt3, tt4), us(tt4, tt5), (long long)(stage_need * 4 / 1024));
  }
  return rc;
}

}  // extern "C"
