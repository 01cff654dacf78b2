"""Host driver of the DGraph mini-batch hot path on one MI355X.

Two objects:

* ``BatchChunk`` -- the "plan" for G batches processed together: closed-neighbourhood entries,
  per-batch column counts, owner election and the 1-hop / 2-hop gather-aggregates.  This is the
  device replacement of ``GCNAggregator.forward`` (`src/graphsage.py:295-360`).  The aggregation
  does not depend on the weights (the feature table is frozen, `src/model_handler.py:264`), so a
  whole epoch's worth of batches is aggregated by a handful of large launches instead of 150 tiny
  ones -- that is what makes the gather HBM-bound rather than launch-bound.
* ``MiniBatchEngine`` -- parameters + Adam state in one packed fp32 block and the per-step kernel
  chain (``GCNEncoder.forward`` + ``GCN.loss`` + backward + Adam, `src/graphsage.py:171-258,395-454`).

PyTorch is used for device memory and streams only; all arithmetic happens in libggad_hip.so.
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr
from .graph import DeviceGraph


def _i32(n, device):
    return torch.empty(int(max(n, 1)), dtype=torch.int32, device=device)


def _f32(n, device):
    return torch.empty(int(max(n, 1)), dtype=torch.float32, device=device)


def _check_i32(x: int) -> None:
    if x >= 2 ** 31:
        raise ValueError("chunk too large for int32 entry indices; use fewer batches per chunk")


def pack_features(feat: torch.Tensor) -> torch.Tensor:
    """(N, F) feature table -> packed (N, stride) table: F features then zeroed per-slot int32 counters, rows
    128-byte aligned (`ggad_mb_packed_stride`).  One random line per gathered neighbour then holds x_k AND c'_k."""
    n, f = feat.shape
    stride = int(_lib.load().ggad_mb_packed_stride(f))
    out = torch.zeros(n, stride, dtype=torch.float32, device=feat.device)
    out[:, :f] = feat
    return out


class BatchChunk:
    """Device plan of up to ``max_batches`` batches (see module docstring).

    Buffers have fixed capacity and fixed addresses (so that a captured hipGraph can be replayed
    after a rebuild); ``build`` grows them only when a chunk does not fit (``generation`` then
    changes, which invalidates captured graphs).
    """

    def __init__(self, graph: DeviceGraph, feat: torch.Tensor, embed_dim: int, max_batches: int,
                 rows_cap: int, ent_cap: int, train: bool = True, reset_mode: str = "auto", feat_dim: Optional[int] = None,
                 hop2: str = "global", node_major: bool = True):
        self.lib = _lib.load()
        self.node_major = bool(node_major)     # "ldsw": one pass over a node's neighbour rows for all its occurrences in the chunk
        self.g = graph
        self.feat = feat
        self.stride = int(feat.shape[1])
        self.F = int(feat_dim) if feat_dim is not None else self.stride
        # stride > F with hop2 != "ldsw": counters of the 2-hop histogram live inside the feature rows ("packed");
        # with "ldsw" a wider stride is plain padding (rows aligned to 128 B for the one-random-access gather)
        self.packed = self.stride > self.F and hop2 != "ldsw"
        # 2-hop mode: "ldsw" = LDS counting per (tile, batch) + streamed per-pair counts; "global" = atomics on per-batch
        # slots in HBM; "tiled" / "ktile" = earlier LDS / tile-major variants kept for comparison
        self.hop2 = "global" if self.packed else hop2
        if self.hop2 not in ("tiled", "ktile", "global", "ldsw"):
            raise ValueError("hop2 must be 'ldsw', 'tiled', 'ktile' or 'global'")
        if self.packed and train and max_batches > self.stride - self.F:
            raise ValueError(f"packed feature rows hold {self.stride - self.F} counter slots, chunk wants {max_batches}")
        self.D = int(embed_dim)
        self.dev = feat.device
        self.train = train
        self.max_batches = int(max_batches)
        self.reset_mode = reset_mode
        _check_i32(self.max_batches * graph.n)
        # per-batch counter slots: int32[max_batches][n]; zero on entry, zeroed again by reset()
        self.cnt1 = torch.zeros(self.max_batches * graph.n, dtype=torch.int32, device=self.dev)
        self.own1 = torch.zeros(self.max_batches * graph.n, dtype=torch.int32, device=self.dev)
        self.cnt2 = None
        self._cnt2_elems = self.max_batches * graph.n
        if train and not self.packed and self.hop2 in ("global", "ktile"):
            self.cnt2 = torch.zeros(self._cnt2_elems, dtype=torch.int32, device=self.dev)
        self.rows_cap = 0
        self.ent_cap = 0
        self.generation = 0
        self._stage_evt = None
        self._alloc_rows(rows_cap)
        self._alloc_ents(ent_cap)
        self.n_batches = 0
        self.n_rows = 0
        self.n_ents = 0
        self.batch_ptr_host = np.zeros(1, dtype=np.int32)
        self.ent_ptr_host = np.zeros(1, dtype=np.int64)
        self.batch_max_row = np.zeros(1, dtype=np.int32)
        self.dirty = False
        self.build_count = 0
        self.gather2_events = None      # optional (start, end) torch events recorded around the gather2 launch

    # ---- allocation
    def _alloc_rows(self, cap: int) -> None:
        cap = int(cap)
        d = self.dev
        self.rows_cap = cap
        self.generation += 1
        # one int32 staging block uploaded per build: batch_ptr | nodes | labels | pos_meta | row_pos
        n_stage = 2 * (self.max_batches + 1) + 4 * cap
        self.stage_host = torch.empty(n_stage, dtype=torch.int32)
        if d.type == "cuda":
            self.stage_host = self.stage_host.pin_memory()
        self.stage = _i32(n_stage, d)
        o = self.max_batches + 1
        self.batch_ptr = self.stage[:o]
        self.batch_ent_ptr = self.stage[n_stage - o:]            # first entry of every batch (tail of the block)
        self.nodes = self.stage[o:o + cap]
        self.labels = self.stage[o + cap:o + 2 * cap]
        self.pos_meta = self.stage[o + 2 * cap:o + 3 * cap]
        self.row_pos = self.stage[o + 3 * cap:o + 4 * cap]
        self.row_r = _i32(cap, d)
        self.row_slot = _i32(cap, d)
        self.ent_ptr = _i32(cap + 1, d)
        self.scan_ws = _i32(self.lib.ggad_scan_workspace_elems(cap), d)
        self.x1 = _f32(cap * self.F, d)
        if self.train:
            n = cap * self.D
            self.h1, self.nbar, self.gen = _f32(n, d), _f32(n, d), _f32(n, d)
            self.dz, self.coef_a, self.coef_g = _f32(n, d), _f32(n, d), _f32(n, d)
            self.d_h1 = self.d_gen = self.d_nbar = None        # raw loss gradients: allocated on demand (debug / tests)
        if getattr(self, "ent_cap", 0):
            self._alloc_chunks()

    def _alloc_chunks(self) -> None:
        """Row-chunk tables of the plan (pieces of <= 16 entries of one row) and their partial-sum buffer: the unit of work
        of the chunk-parallel forward kernels for batches with hub rows (`ggad_mb_row_chunks`, step.hip)."""
        if not self.train or self.rows_cap == 0 or self.ent_cap == 0:
            return
        cl = int(self.lib.ggad_mb_chunk_len())
        n = self.ent_cap // cl + self.rows_cap + 8
        d = self.dev
        self.ck_cap = n
        self.row_ck_ptr, self.nck_tmp = _i32(self.rows_cap + 1, d), _i32(self.rows_cap, d)
        self.ck_rc, self.ck_e0 = _i32(n, d), _i32(n, d)
        self.chunk_part = _f32(n * 64, d)

    def _alloc_ents(self, cap: int) -> None:
        cap = int(cap)
        d = self.dev
        self.ent_cap = cap
        self.generation += 1
        self.ent_col, self.ent_slot, self.ent_row = _i32(cap, d), _i32(cap, d), _i32(cap, d)
        self.ent_own, self.ent_c1 = _i32(cap, d), _i32(cap, d)
        self.x2 = _f32(cap * self.F, d) if self.train else None
        if self.train and self.hop2 in ("tiled", "ktile", "ldsw"):
            self.own_flags, self.own_pos, self.own_list = _i32(cap, d), _i32(cap + 1, d), _i32(cap, d)
            self.own_scan_ws = _i32(self.lib.ggad_scan_workspace_elems(cap), d)
        if self.train and self.hop2 == "ldsw":
            self.own_deg, self.own_rp, self.pw_base = _i32(cap, d), _i32(cap, d), _i32(cap + 1, d)
            self.seg_t = _i32(int(self.lib.ggad_mb_ldsw_seg_elems(self.g.n, cap)), d)
            self.own_next = _i32(cap, d)
            self.grp = _i32(8 * cap + 1, d)        # group table of the node-major gather (+ its counter word)
            if not hasattr(self, "node_head"):
                self.node_head = torch.zeros(self.g.n, dtype=torch.int32, device=d)   # zero between gathers (the kernel cleans up)
            if not hasattr(self, "pc"):
                self.pc = torch.empty(0, dtype=torch.int16, device=d)      # uint16 per-pair counts (raw storage)
        self._alloc_chunks()

    # ---- build
    def build(self, batches: Sequence[np.ndarray], labels: Optional[Sequence[np.ndarray]] = None) -> None:
        """Upload the batches and run the plan + gather kernels on the current stream."""
        if self.dirty:
            self.reset()
        nb = len(batches)
        if nb > self.max_batches or nb == 0:
            raise ValueError(f"chunk holds 1..{self.max_batches} batches, got {nb}")
        sizes = np.fromiter((len(b) for b in batches), dtype=np.int64, count=nb)
        if (sizes == 0).any():
            raise ValueError("empty batch")
        rows = int(sizes.sum())
        nodes = np.concatenate([np.asarray(b, dtype=np.int64) for b in batches])
        if nodes.min() < 0 or nodes.max() >= self.g.n:
            raise ValueError("batch node id out of range")
        r_host = self.g.closed_degrees(nodes)                  # exact |N(i) + {i}| per row
        ent_ptr_host = np.zeros(rows + 1, dtype=np.int64)
        np.cumsum(r_host, out=ent_ptr_host[1:])
        n_ents = int(ent_ptr_host[-1])
        _check_i32(n_ents)
        if rows > self.rows_cap:
            self._alloc_rows(int(rows * 1.25) + 64)
        if n_ents > self.ent_cap:
            self._alloc_ents(int(n_ents * 1.25) + 1024)
        bp = np.zeros(nb + 1, dtype=np.int32)
        np.cumsum(sizes, out=bp[1:])
        if self._stage_evt is not None:
            self._stage_evt.synchronize()       # previous upload has left the pinned staging block
        st = self.stage_host.numpy()
        o = self.max_batches + 1
        st[:nb + 1] = bp
        st[nb + 1:o] = bp[-1]
        cap = self.rows_cap
        st[o:o + rows] = nodes
        if self.train:
            if labels is None:
                raise ValueError("training chunk needs labels")
            lab = np.concatenate([np.asarray(l, dtype=np.int64) for l in labels])
            if len(lab) != rows:
                raise ValueError("labels / batches length mismatch")
            if ((lab != 0) & (lab != 1)).any():
                raise ValueError("labels must be 0/1")
            st[o + cap:o + cap + rows] = lab
            # column q of `combined_all` holds the label-0 rows in order, then the label-1 rows (graphsage.py:450);
            # pos_meta[q] = (src_row << 2) | (label[src] << 1) | label[q]
            # row_pos[row] = column of that row (inverse permutation, batch-relative)
            meta = np.empty(rows, dtype=np.int64)
            rpos = np.empty(rows, dtype=np.int64)
            for b in range(nb):
                r0, r1 = bp[b], bp[b + 1]
                lb = lab[r0:r1]
                order = np.argsort(lb != 0, kind="stable")
                meta[r0:r1] = ((order + r0) << 2) | (lb[order] << 1) | lb
                rpos[r0 + order] = np.arange(r1 - r0)
            st[o + 2 * cap:o + 2 * cap + rows] = meta
            st[o + 3 * cap:o + 3 * cap + rows] = rpos
        bep = ent_ptr_host[bp]                                   # entry offsets of the batch starts
        st[len(st) - o:len(st) - o + nb + 1] = bep
        st[len(st) - o + nb + 1:] = bep[-1]
        self.stage.copy_(self.stage_host, non_blocking=True)
        if self.dev.type == "cuda":
            self._stage_evt = torch.cuda.Event()
            self._stage_evt.record()
        self.n_batches, self.n_rows, self.n_ents = nb, rows, n_ents
        self.build_count += 1
        self.batch_ptr_host = bp
        self.ent_ptr_host = ent_ptr_host
        self.batch_max_row = np.maximum.reduceat(r_host, bp[:-1]).astype(np.int32)     # largest closed neighbourhood per batch
        g = self.g
        call("ggad_mb_row_degree", ptr(g.rowptr), ptr(g.col), ptr(self.nodes), ptr(self.batch_ptr), nb, rows,
             ptr(self.row_r), ptr(self.row_slot))
        call("ggad_exclusive_scan_i32", ptr(self.row_r), ptr(self.ent_ptr), rows, ptr(self.scan_ws))
        call("ggad_mb_expand1", ptr(g.rowptr), ptr(g.col), ptr(self.nodes), ptr(self.row_slot), ptr(self.ent_ptr), rows,
             g.n, ptr(self.ent_col), ptr(self.ent_slot), ptr(self.ent_row), ptr(self.cnt1), ptr(self.own1))
        self.dirty = True
        call("ggad_mb_gather1", ptr(self.feat), self.F, self.stride, ptr(self.row_slot), ptr(self.ent_ptr), ptr(self.ent_col), rows,
             g.n, ptr(self.cnt1), ptr(self.own1), ptr(self.ent_own), ptr(self.ent_c1), ptr(self.x1))
        if self.train:
            call("ggad_mb_row_chunks", ptr(self.ent_ptr), rows, ptr(self.nck_tmp), ptr(self.row_ck_ptr), ptr(self.ck_rc),
                 ptr(self.ck_e0), ptr(self.scan_ws))
        if self.train and self._use_tiled():
            tot = self.ent_total_ptr()
            e = n_ents
            self.x2[:e * self.F].zero_()
            call("ggad_mb_owner_flags", ptr(self.ent_own), tot, e, ptr(self.own_flags))
            call("ggad_exclusive_scan_i32", ptr(self.own_flags), ptr(self.own_pos), e, ptr(self.own_scan_ws))
            if self.gather2_events is not None:
                self.gather2_events[0].record()
            call("ggad_mb_hop2_tiled", ptr(g.rowptr), ptr(g.col), ptr(self.feat), self.F, self.stride, g.n,
                 ptr(g.tile_offsets()), ptr(self.own_flags), ptr(self.own_pos), ptr(self.own_list), ptr(self.batch_ent_ptr),
                 nb, ptr(self.ent_col), e, ptr(self.x2))
            if self.gather2_events is not None:
                self.gather2_events[1].record()
        elif self.train and self._use_ldsw(nodes):
            tot = self.ent_total_ptr()
            e = n_ents
            call("ggad_mb_owner_flags", ptr(self.ent_own), tot, e, ptr(self.own_flags))
            call("ggad_exclusive_scan_i32", ptr(self.own_flags), ptr(self.own_pos), e, ptr(self.own_scan_ws))
            call("ggad_mb_hop2_ldsw_count", ptr(g.rowptr), ptr(g.col), g.n,
                 ptr(g.tile_offsets(self.lib.ggad_mb_ldsw_tile_shift())), ptr(self.own_flags), ptr(self.own_pos),
                 ptr(self.own_list), ptr(self.batch_ent_ptr), nb, ptr(self.ent_col), e, ptr(self.own_deg), ptr(self.own_rp),
                 ptr(self.pw_base), ptr(self.own_scan_ws), ptr(self.seg_t), ptr(self.pc))
            if self.gather2_events is not None:
                self.gather2_events[0].record()
            nm = self.node_major and self.F <= 64
            call("ggad_mb_hop2_ldsw_gather", ptr(g.rowptr), ptr(g.col), ptr(self.feat), self.F, self.stride, ptr(self.own_pos),
                 ptr(self.own_list), ptr(self.ent_col), e, ptr(self.pw_base), ptr(self.pc),
                 ptr(self.node_head) if nm else 0, ptr(self.own_next) if nm else 0, self.grp_ptr(e) if nm else 0, ptr(self.x2))
            if self.gather2_events is not None:
                self.gather2_events[1].record()
        elif self.train and self.hop2 == "ktile" and self.n_ents <= (1 << 22):
            self.last_hop2 = "ktile"
            tot = self.ent_total_ptr()
            e = n_ents
            self.x2[:e * self.F].zero_()
            call("ggad_mb_owner_flags", ptr(self.ent_own), tot, e, ptr(self.own_flags))
            call("ggad_exclusive_scan_i32", ptr(self.own_flags), ptr(self.own_pos), e, ptr(self.own_scan_ws))
            if self.gather2_events is not None:
                self.gather2_events[0].record()
            call("ggad_mb_hop2_ktile", ptr(g.rowptr), ptr(g.col), ptr(self.feat), self.F, self.stride, g.n,
                 ptr(g.tile_offsets()), ptr(self.own_flags), ptr(self.own_pos), ptr(self.own_list), ptr(self.ent_col),
                 ptr(self.ent_slot), e, ptr(self.cnt2), ptr(self.x2))
            if self.gather2_events is not None:
                self.gather2_events[1].record()
        elif self.train:
            if self.cnt2 is None and not self.packed:        # tiled mode fell back (a batch with >= 65,536 entries)
                self.cnt2 = torch.zeros(self._cnt2_elems, dtype=torch.int32, device=self.dev)
            tot = self.ent_total_ptr()
            call("ggad_mb_count2", ptr(g.rowptr), ptr(g.col), ptr(self.ent_col), ptr(self.ent_slot), tot, n_ents, g.n,
                 ptr(self.own1), ptr(self.cnt2) if self.cnt2 is not None else 0, ptr(self.feat), self.F, self.stride)
            if self.gather2_events is not None:
                self.gather2_events[0].record()
            call("ggad_mb_gather2", ptr(g.rowptr), ptr(g.col), ptr(self.feat), self.F, self.stride, ptr(self.ent_col),
                 ptr(self.ent_slot), ptr(self.ent_own), tot, n_ents, g.n, ptr(self.cnt2) if self.cnt2 is not None else 0,
                 ptr(self.x2))
            if self.gather2_events is not None:
                self.gather2_events[1].record()

    def _use_tiled(self) -> bool:
        """LDS-tiled 2-hop needs < 65,536 owners per batch (16-bit counters) and <= 4 Mi entries per chunk (scan)."""
        self.last_hop2 = "global"
        if self.hop2 != "tiled":
            return False
        per_batch = np.diff(self.ent_ptr_host[self.batch_ptr_host])
        if per_batch.max() >= 65536 or self.n_ents > (1 << 22):
            return False
        self.last_hop2 = "tiled"
        return True

    def _use_ldsw(self, nodes: np.ndarray) -> bool:
        """"ldsw" needs < 65,536 owners per batch (16-bit LDS counters), <= 4 Mi entries per chunk (scan) and < 2^31
        2-hop pairs per chunk (int32 offsets into pc[]).  The pair count is bounded on the host from the static per-node
        table sum_{k in N(i) + i} deg(k) -- the exact number is only known on the device and reading it back would
        serialise the host with the GPU."""
        if self.hop2 != "ldsw":
            return False
        # the static tile-offset table is N x (N / 32,768 + 1) ints: fine at DGraph size (1.7 GB), quadratic beyond ~30 M nodes
        shift = int(self.lib.ggad_mb_ldsw_tile_shift())
        if 4 * int(self.lib.ggad_mb_tile_offsets_elems(self.g.n, shift)) > (32 << 30):
            self.last_hop2 = "global"
            return False
        per_batch = np.diff(self.ent_ptr_host[self.batch_ptr_host])
        bound = int(self.g.pair_bound_host[nodes].sum())
        if per_batch.max() >= 65536 or self.n_ents > (1 << 22) or bound >= (1 << 31) - 1:
            self.last_hop2 = "global"
            return False
        if bound > self.pc.numel():
            self.pc = torch.empty(int(bound * 1.25) + 1024, dtype=torch.int16, device=self.dev)
            self.generation += 1
        self.last_hop2 = "ldsw"
        return True

    def grp_ptr(self, e: int) -> int:
        """The group table is laid out for n_entries_cap = e of THIS build (counter word right behind 8 * e ints)."""
        return self.grp.data_ptr()

    def ent_total_ptr(self) -> int:
        return self.ent_ptr.data_ptr() + 4 * self.n_rows

    def _memset_is_cheaper(self) -> bool:
        """Zeroing whole slots streams 4*n bytes per batch; walking touches one 64-byte sector per counted
        2-hop neighbour.  Estimated from the graph's mean neighbour degree (sum deg^2 / sum deg)."""
        if self.reset_mode in ("memset", "walk"):
            return self.reset_mode == "memset"
        g = self.g
        if not hasattr(g, "_mean_nbr_deg"):
            d = g.deg_host.astype(np.float64)
            g._mean_nbr_deg = float((d * d).sum() / max(1.0, d.sum()))
        walk_bytes = self.n_ents * g._mean_nbr_deg * 64.0
        return walk_bytes > self.n_batches * g.n * 4.0 * 2.0

    def reset(self) -> None:
        """Zero the counter slots again (memset of the used slots, or a walk over the entries of the last build)."""
        if not self.dirty:
            return
        g = self.g
        if self.train and getattr(self, "last_hop2", "global") in ("tiled", "ldsw"):
            # only the small 1-hop slots were touched
            call("ggad_mb_plan_reset", ptr(g.rowptr), ptr(g.col), ptr(self.ent_col), ptr(self.ent_slot), ptr(self.ent_own),
                 self.ent_total_ptr(), self.n_ents, g.n, ptr(self.cnt1), 0, 0, 0, self.F, self.stride)
            self.dirty = False
            return
        stream_reset = self.train and self._memset_is_cheaper()
        if stream_reset and not self.packed:
            used = self.n_batches * g.n
            self.cnt1[:used].zero_()
            self.cnt2[:used].zero_()
        else:
            # walk the entries: always for the (small) 1-hop counters, and for the 2-hop ones on sparse graphs
            hop2 = 1 if (self.train and not stream_reset) else 0
            call("ggad_mb_plan_reset", ptr(g.rowptr), ptr(g.col), ptr(self.ent_col), ptr(self.ent_slot), ptr(self.ent_own),
                 self.ent_total_ptr(), self.n_ents, g.n, ptr(self.cnt1), ptr(self.cnt2) if self.cnt2 is not None else 0, hop2,
                 ptr(self.feat) if self.packed else 0, self.F, self.stride)
            if stream_reset:   # packed: one streaming pass over the counter words of the used slots
                call("ggad_mb_reset_packed", ptr(self.feat), g.n, self.F, self.stride, self.n_batches)
        self.dirty = False

    def batch_rows(self, b: int):
        return int(self.batch_ptr_host[b]), int(self.batch_ptr_host[b + 1])

    def batch_ents(self, b: int):
        r0, r1 = self.batch_rows(b)
        return int(self.ent_ptr_host[r0]), int(self.ent_ptr_host[r1])

    def max_batch_ents(self) -> int:
        e = self.ent_ptr_host[self.batch_ptr_host]
        return int(np.diff(e).max()) if len(e) > 1 else 0


def reduce_gradients(grads: torch.Tensor, world_size: int, allreduce: Optional[Callable]) -> float:
    """The data-parallel exchange step: ONE all-reduce(SUM) of the packed gradient block; returns the scale
    (1/W) the optimiser applies.  Kept separate so that the CPU (gloo) tests drive the very same contract."""
    if world_size > 1:
        if allreduce is None:
            raise ValueError("world_size > 1 needs an all-reduce callable")
        allreduce(grads)
    return 1.0 / world_size


class MiniBatchEngine:
    """Parameters, optimiser state and the per-batch kernel chain."""

    def __init__(self, feat_dim: int, embed_dim: int, device, lr: float = 1e-3, weight_decay: float = 0.007,
                 chain: int = 0):
        self.lib = _lib.load()
        # 0: fused-forward step (5 launches, F == 17), 1: row-wise 3-launch step, 2: 6 launches,
        # 3: ONE persistent launch per chunk (single GPU, F == 17, <= 256 rows per batch; else the chain-0 launches)
        self.chain = int(chain)
        self.persistent_wgs = 64         # workgroups of the persistent kernel: at most the CUs of the stream it runs on
        self.ps_ws = None
        self.F, self.D = int(feat_dim), int(embed_dim)
        if self.D > self.lib.ggad_max_embed_dim():
            raise ValueError(f"emb_size {self.D} > {self.lib.ggad_max_embed_dim()} is not supported by the HIP step kernels")
        self.dev = torch.device(device)
        self.lr, self.wd = float(lr), float(weight_decay)
        self.n_train = int(self.lib.ggad_mb_param_count(self.D, self.F))
        n_block = int(self.lib.ggad_mb_param_block_elems(self.D, self.F))
        self.params = torch.zeros(n_block, dtype=torch.float32, device=self.dev)
        self.exp_avg = torch.zeros(self.n_train, dtype=torch.float32, device=self.dev)
        self.exp_avg_sq = torch.zeros(self.n_train, dtype=torch.float32, device=self.dev)
        self.grads = torch.zeros(self.n_train, dtype=torch.float32, device=self.dev)
        self.loss_ws = _f32(self.lib.ggad_mb_loss_workspace_elems(256), self.dev)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.n_parts = int(self.lib.ggad_mb_bwd_parts())
        self.dw_part = torch.zeros(int(self.lib.ggad_mb_dw_part_elems(256, self.D, self.F)), dtype=torch.float32, device=self.dev)
        self.h2 = _f32(1024 * self.D, self.dev)
        self.loss_log = torch.zeros(8 * 256, dtype=torch.float32, device=self.dev)
        D, F = self.D, self.F
        # views with the reference's state_dict names/shapes
        self.weight = self.params[0:D].view(1, D)
        self.enc_weight = self.params[D:D + D * F].view(D, F)
        self.enc_fc_weight = self.params[D + D * F:D + D * F + D * D].view(D, D)

    # ---- parameters
    def load_params(self, weight, enc_weight, enc_fc_weight) -> None:
        with torch.no_grad():
            self.weight.copy_(torch.as_tensor(weight, dtype=torch.float32).reshape(1, self.D))
            self.enc_weight.copy_(torch.as_tensor(enc_weight, dtype=torch.float32).reshape(self.D, self.F))
            self.enc_fc_weight.copy_(torch.as_tensor(enc_fc_weight, dtype=torch.float32).reshape(self.D, self.D))
        self.sync_params()

    def sync_params(self) -> None:
        call("ggad_mb_params_sync", ptr(self.params), self.D, self.F)

    def reset_optimizer(self) -> None:
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_counter.zero_()

    def ensure_capacity(self, ch: BatchChunk, log_slots: int) -> None:
        """Grow scratch that depends on the chunk (never called inside a captured region)."""
        need = max(1, ch.max_batch_ents()) * self.D
        if self.h2.numel() < need:
            self.h2 = _f32(int(need * 1.25), self.dev)
        max_rows = int(np.diff(ch.batch_ptr_host).max()) if ch.n_batches else 1
        need = int(self.lib.ggad_mb_loss_workspace_elems(max_rows))
        if self.loss_ws.numel() < need:
            self.loss_ws = _f32(need, self.dev)
        need = int(self.lib.ggad_mb_dw_part_elems(max_rows, self.D, self.F))
        if self.dw_part.numel() < need:
            self.dw_part = torch.zeros(need, dtype=torch.float32, device=self.dev)
        if self.loss_log.numel() < 8 * log_slots:
            new = torch.zeros(8 * max(log_slots, 2 * (self.loss_log.numel() // 8)), dtype=torch.float32, device=self.dev)
            new[:self.loss_log.numel()].copy_(self.loss_log)
            self.loss_log = new

    # ---- kernel chain
    def step_desc(self, ch: BatchChunk, b: int, log_slot: int) -> "_lib.MbStep":
        r0, r1 = ch.batch_rows(b)
        e0, e1 = ch.batch_ents(b)
        s = _lib.MbStep()
        s.params, s.exp_avg, s.exp_avg_sq, s.grads = ptr(self.params), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.grads)
        s.step_counter = ptr(self.step_counter)
        s.x1, s.x2 = ptr(ch.x1), ptr(ch.x2)
        s.ent_ptr, s.ent_own, s.ent_row = ptr(ch.ent_ptr), ptr(ch.ent_own), ptr(ch.ent_row)
        s.labels, s.pos_meta, s.row_pos = ptr(ch.labels), ptr(ch.pos_meta), ptr(ch.row_pos)
        s.h1, s.nbar, s.gen = ptr(ch.h1), ptr(ch.nbar), ptr(ch.gen)
        s.dz, s.coef_a, s.coef_g = ptr(ch.dz), ptr(ch.coef_a), ptr(ch.coef_g)
        s.h2, s.dw_part, s.loss_ws = ptr(self.h2), ptr(self.dw_part), ptr(self.loss_ws)
        s.losses8 = self.loss_log.data_ptr() + 32 * log_slot
        s.D, s.F, s.row0, s.n_rows, s.ent0, s.n_ents = self.D, self.F, r0, r1 - r0, e0, e1 - e0
        s.lr, s.weight_decay = self.lr, self.wd
        s.chain = self.chain if self.chain != 3 else 0
        s.max_row_entries = int(ch.batch_max_row[b])
        if ch.train and getattr(ch, "row_ck_ptr", None) is not None:
            s.row_ck_ptr, s.ck_rc, s.ck_e0, s.chunk_part = ptr(ch.row_ck_ptr), ptr(ch.ck_rc), ptr(ch.ck_e0), ptr(ch.chunk_part)
        return s

    def loss_and_grads(self, ch: BatchChunk, b: int, log_slot: int = 0) -> None:
        """forward + loss + backward for batch b; gradients land in self.grads (packed w | W | fc)."""
        self.ensure_capacity(ch, log_slot + 1)
        s = self.step_desc(ch, b, log_slot)
        _lib.check(self.lib.ggad_mb_train_step(ctypes.byref(s), 0, _lib.current_stream()), "ggad_mb_train_step")

    def adam_step(self, grad_scale: float = 1.0) -> None:
        call("ggad_mb_adam", ptr(self.params), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.grads), self.D, self.F,
             self.lr, self.wd, float(grad_scale), ptr(self.step_counter))

    def train_chunk(self, ch: BatchChunk, allreduce=None, world_size: int = 1, log_base: int = 0) -> None:
        """One optimiser step per batch of the chunk (src/model_handler.py:330-364)."""
        self.ensure_capacity(ch, log_base + ch.n_batches)
        stream = _lib.current_stream()
        fuse = 1 if (allreduce is None and world_size == 1) else 0
        if fuse and self.chain == 3 and self._train_chunk_persistent(ch, log_base, stream):
            return
        if fuse:
            # single GPU: the whole chunk in one host call (the C loop issues the launches; Python per step costs more
            # than the 5-8 us kernels it feeds)
            s = self.step_desc(ch, 0, log_base)
            bp = np.ascontiguousarray(ch.batch_ptr_host[:ch.n_batches + 1], dtype=np.int32)
            ep = np.ascontiguousarray(ch.ent_ptr_host[bp], dtype=np.int64)
            mr = np.ascontiguousarray(ch.batch_max_row, dtype=np.int32)
            _lib.check(self.lib.ggad_mb_train_chunk(ctypes.byref(s), ch.n_batches, bp.ctypes.data, ep.ctypes.data, mr.ctypes.data,
                                                    self.loss_log.data_ptr(), log_base, 1, stream), "ggad_mb_train_chunk")
            return
        # data parallel: backward -> all-reduce -> Adam per batch; the loop runs in C, Python only serves the exchange
        scale = 1.0 / world_size
        if world_size > 1 and allreduce is None:
            raise ValueError("world_size > 1 needs an all-reduce callable")
        err = []

        def exchange(_user):
            try:
                if allreduce is not None:
                    allreduce(self.grads)
                return 0
            except BaseException as exc:      # never let an exception cross the C frame
                err.append(exc)
                return 1
        cb = _lib.EXCHANGE_CB(exchange)
        s = self.step_desc(ch, 0, log_base)
        bp = np.ascontiguousarray(ch.batch_ptr_host[:ch.n_batches + 1], dtype=np.int32)
        ep = np.ascontiguousarray(ch.ent_ptr_host[bp], dtype=np.int64)
        mr = np.ascontiguousarray(ch.batch_max_row, dtype=np.int32)
        rc = self.lib.ggad_mb_train_chunk_dp(ctypes.byref(s), ch.n_batches, bp.ctypes.data, ep.ctypes.data, mr.ctypes.data,
                                             self.loss_log.data_ptr(), log_base, scale, cb, None, stream)
        if err:
            raise err[0]
        _lib.check(rc, "ggad_mb_train_chunk_dp")

    def _train_chunk_persistent(self, ch: BatchChunk, log_base: int, stream) -> bool:
        """All steps of the chunk in one persistent launch (`ggad_mb_train_chunk_persistent`); False = not eligible."""
        bp = ch.batch_ptr_host[:ch.n_batches + 1]
        max_rows = int(np.diff(bp).max())
        if self.F != 17 or max_rows > int(self.lib.ggad_mb_persistent_max_rows()) or not ch.train:
            return False
        cl = int(self.lib.ggad_mb_persistent_chunk_len())
        r = np.diff(ch.ent_ptr_host[:ch.n_rows + 1])
        per_row = (r + cl - 1) // cl
        max_chunks = int(np.add.reduceat(per_row, bp[:-1].astype(np.int64)).max())
        need = int(self.lib.ggad_mb_persistent_ws_elems(max_chunks, self.persistent_wgs))
        if self.ps_ws is None or self.ps_ws.numel() < need:
            self.ps_ws = _f32(int(need * 1.25), self.dev)
        s = self.step_desc(ch, 0, log_base)
        _lib.check(self.lib.ggad_mb_train_chunk_persistent(ctypes.byref(s), ch.n_batches, ptr(ch.batch_ptr), ptr(ch.batch_ent_ptr),
                                                           max_rows, max_chunks, self.persistent_wgs, self.loss_log.data_ptr(),
                                                           log_base, ptr(self.ps_ws), stream), "ggad_mb_train_chunk_persistent")
        return True

    def forward_batch(self, ch: BatchChunk, b: int) -> None:
        """project + fwd_rows only (layered API / tests): fills ch.h1, ch.nbar, ch.gen for batch b."""
        self.ensure_capacity(ch, 1)
        r0, r1 = ch.batch_rows(b)
        e0, e1 = ch.batch_ents(b)
        call("ggad_mb_project", ptr(self.params), self.D, self.F, ptr(ch.x2), ptr(ch.ent_own), e0, e1 - e0, ptr(self.h2))
        call("ggad_mb_fwd_rows", ptr(self.params), self.D, self.F, ptr(ch.x1), ptr(self.h2), ptr(ch.ent_ptr), ptr(ch.ent_own),
             ptr(ch.labels), r0, r1 - r0, e0, ptr(ch.h1), ptr(ch.nbar), ptr(ch.gen))

    def score_chunk(self, ch: BatchChunk, out: torch.Tensor) -> None:
        """to_prob for every row of an inference chunk (src/graphsage.py:178-181)."""
        call("ggad_mb_score", ptr(self.params), self.D, self.F, ptr(ch.x1), ch.n_rows, ptr(out))

    def losses(self, n: int) -> np.ndarray:
        """(n, 4) array of {total, cls, margin, rec} for the first n logged steps (synchronises)."""
        return self.loss_log[:8 * n].view(n, 8)[:, :4].cpu().numpy()
