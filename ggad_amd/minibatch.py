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
import os
import sys
import time
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


_BUILD_TIMING = bool(os.environ.get("GGAD_BUILD_TIMING"))


def _addr(a: np.ndarray) -> int:
    return a.__array_interface__["data"][0]


class RowList(list):
    """The rows of a C-contiguous 2-D int64 matrix as a list that remembers the matrix: what `BatchSchedule.next_batches` hands out.
    `BatchChunk.build` then takes the matrix as it is -- no per-row work (checking the 40 row pointers of a 20-batch build in
    Python cost 70-90 us on the critical path of a one-chunk run).  A slice is again a RowList; anything that changes the list in
    place forgets the matrix, and the list is then treated like any other list of arrays."""

    __slots__ = ("matrix",)

    def __init__(self, matrix: np.ndarray):
        super().__init__(matrix)
        self.matrix = matrix if (matrix.ndim == 2 and matrix.dtype == np.int64 and matrix.flags.c_contiguous) else None

    def __getitem__(self, i):
        if isinstance(i, slice) and self.matrix is not None:
            n = list.__len__(self)
            start, stop, step = i.indices(n)
            if step == 1 and n == self.matrix.shape[0]:
                if start == 0 and stop == n:
                    return self                                   # (the whole list: no new row views)
                return RowList(self.matrix[start:max(start, stop)])
        return super().__getitem__(i)

    def _forget(name):                                    # noqa: N805
        def method(self, *a, **kw):
            self.matrix = None
            return getattr(list, name)(self, *a, **kw)
        method.__name__ = name
        return method

    for _n in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove", "reverse", "sort", "clear"):
        locals()[_n] = _forget(_n)
    del _n, _forget


def _rows_as_flat(rows):
    """The rows of a list as ONE flat int64 view when they are consecutive rows of a C-contiguous 2-D int64 matrix (else None)."""
    if type(rows) is RowList:
        m = rows.matrix
        return m.reshape(-1) if (m is not None and m.shape[0] == len(rows) and m.shape[0] > 0) else _rows_as_flat(list(rows))
    try:
        a, z = rows[0], rows[-1]
        base = a.base
        if base is None or z.base is not base or base.ndim != 2 or base.dtype != np.int64 or not base.flags.c_contiguous:
            return None
        w = base.shape[1]
        if a.ndim != 1 or len(a) != w or len(z) != w or a.dtype != np.int64:
            return None
        p0, pa, pz = base.ctypes.data, a.ctypes.data, z.ctypes.data
        r0, nb = (pa - p0) // (8 * w), len(rows)
        if (pa - p0) % (8 * w) or pz - pa != 8 * w * (nb - 1) or r0 + nb > base.shape[0]:
            return None
        if nb > 2:       # EVERY row where the matrix has it (a list that merely starts and ends like a slice -- rows permuted by a
            #              hook, say -- must not be built in matrix order): one vectorised comparison of nb pointers
            ptrs = np.fromiter((r.__array_interface__["data"][0] if (r.base is base and r.ndim == 1 and len(r) == w) else -1
                                for r in rows), dtype=np.int64, count=nb)
            if not np.array_equal(ptrs, pa + 8 * w * np.arange(nb, dtype=np.int64)):
                return None
        return base[r0:r0 + nb].reshape(-1)
    except (AttributeError, IndexError, TypeError):
        return None


def _check_i32(x: int) -> None:
    if x >= 2 ** 31:
        raise ValueError("chunk too large for int32 entry indices; use fewer batches per chunk")


class BatchChunk:
    """Device plan of up to ``max_batches`` batches (see module docstring).

    All buffers belong to this object (PyTorch allocations handed to ``ggad_mb_plan_build`` as plain pointers); they
    keep their addresses until a chunk does not fit, then ``build`` grows them and ``generation`` changes.  The 1-hop
    counter slots are clean again when ``build`` returns (the plan resets them itself).
    """

    HOP2_MODES = {"ldsw": 1, "global": 2}

    def __init__(self, graph: DeviceGraph, feat: torch.Tensor, embed_dim: int, max_batches: int,
                 rows_cap: int, ent_cap: int, train: bool = True, feat_dim: Optional[int] = None,
                 hop2: str = "ldsw", node_major: bool = True):
        self.lib = _lib.load()
        self.g = graph
        self.feat = feat
        self.stride = int(feat.shape[1])
        self.F = int(feat_dim) if feat_dim is not None else self.stride     # stride > F: rows padded (128-byte aligned rows)
        if hop2 not in self.HOP2_MODES:
            raise ValueError("hop2 must be 'ldsw' (LDS counting, default) or 'global' (device atomics)")
        self.hop2 = hop2
        self.node_major = bool(node_major)
             # ldsw: all occurrences of a node in the chunk share one pass over its rows
        self.xcd_skip = -1              # 0..7: the plan kernels leave the workgroups of one XCD out (the resident chunk kernel's)
        self.D = int(embed_dim)
        self.dev = feat.device
        self.train = bool(train)
        self.max_batches = int(max_batches)
        _check_i32(self.max_batches * graph.n)
        d = self.dev
        # per-batch counter slots: int32[max_batches][n]; zero on entry, zero again when build() returns
        self.cnt1 = torch.zeros(self.max_batches * graph.n, dtype=torch.int32, device=d)
        self.own1 = torch.zeros(self.max_batches * graph.n, dtype=torch.int32, device=d)
        self.cnt2 = None
        if self.train and self.hop2 == "global":
            self.cnt2 = torch.zeros(self.max_batches * graph.n, dtype=torch.int32, device=d)
        self.counters = torch.zeros(int(self.lib.ggad_mb_plan_counter_elems()), dtype=torch.int32, device=d)
        self.cl = int(self.lib.ggad_mb_chunk_len())
        self.part_stride = max(64, self.F)
        self.generation = 0
        self.rows_cap = self.ent_cap = self.ck_cap = self.stage_cap = 0
        self.pair_cap = self.item_cap = self.part2_cap = self.seg_cap = 0
        self._stage_event = ctypes.c_void_p()
        _lib.check(self.lib.ggad_event_create(0, ctypes.byref(self._stage_event)), "ggad_event_create")
        self.info = _lib.MbPlanInfo()
        self.plan = _lib.MbPlan()
        self.gather2_events = None      # optional (start, stop) handles of ggad_event_create recorded around the 2-hop gather
        self.tile_events = None         # ... and around k_tile_counts (the pair counting before it)
        self.n_batches = self.n_rows = self.n_ents = self.n_chunks = 0
        self.last_hop2 = "none"
        self.build_count = 0
        self.x2 = self.pc = self.part2 = self.seg_t = self.items = self.grp = None
        if self.train and self.hop2 == "ldsw":
            self.node_head = torch.zeros(graph.n, dtype=torch.int32, device=d)        # zero between builds (the plan cleans up)
        self._alloc(rows_cap, ent_cap, ent_cap // self.cl + rows_cap + 8, 0, 0, 0, 0, 0)
        # host tables of the last build, written by the native call (views trimmed to the build: the properties below)
        self._bp_host = np.zeros(self.max_batches + 1, dtype=np.int32)
        self._bep_host = np.zeros(self.max_batches + 1, dtype=np.int64)
        self._bmr_host = np.zeros(self.max_batches, dtype=np.int32)

    @property
    def batch_ptr_host(self) -> np.ndarray:
        return self._bp_host[:self.n_batches + 1]

    @property
    def batch_ent_host(self) -> np.ndarray:
        return self._bep_host[:self.n_batches + 1]

    @property
    def batch_max_row(self) -> np.ndarray:
        return self._bmr_host[:self.n_batches]

    @property
    def ent_ptr_host(self) -> np.ndarray:
        return self._ep_host[:self.n_rows + 1]

    def __del__(self):
        try:
            if getattr(self, "_stage_event", None) is not None and self._stage_event.value:
                self.lib.ggad_event_destroy(self._stage_event)
        except Exception:
            pass

    # ---- allocation
    def _alloc(self, rows, ents, chunks, stage, pairs, items, part2, seg) -> None:
        """Grow whatever is smaller than asked (25 % head room on growth); refresh the descriptor handed to the C side."""
        d = self.dev
        if self.generation > 0 and d.type == "cuda":
            # a grow replaces the pinned staging block and device buffers that an earlier build's asynchronous upload, or the other
            # stream's chunk kernels, may still be using; torch's allocators do not see the native copies: drain the device first
            torch.cuda.synchronize(d)
        grow = lambda need, have: int(need * 1.25) + 64 if need > have else have      # noqa: E731
        rows_cap, ent_cap, ck_cap = grow(rows, self.rows_cap), grow(ents, self.ent_cap), grow(chunks, self.ck_cap)
        stage_need = max(stage, 2 * (self.max_batches + 8) + 7 * (rows_cap + 8) + 2 * (ck_cap + 8))
        if rows_cap != self.rows_cap:
            self.rows_cap = rows_cap
            self.x1 = _f32(rows_cap * self.F, d)
            self._ep_host = np.zeros(rows_cap + 1, dtype=np.int64)
            if self.train:
                n = rows_cap * self.D
                self.h1, self.nbar, self.gen = _f32(n, d), _f32(n, d), _f32(n, d)
                self.dz, self.coef_a, self.coef_g = _f32(n, d), _f32(n, d), _f32(n, d)
        if ent_cap != self.ent_cap:
            self.ent_cap = ent_cap
            self.ent_col, self.ent_slot, self.ent_row = _i32(ent_cap, d), _i32(ent_cap, d), _i32(ent_cap, d)
            self.ent_own, self.ent_c1 = _i32(ent_cap, d), _i32(ent_cap, d)
            if self.train:
                self.x2 = _f32(ent_cap * self.F, d)
                if self.hop2 == "ldsw":
                    self.own_deg, self.own_rp, self.pw_base = _i32(ent_cap, d), _i32(ent_cap, d), _i32(ent_cap, d)
                    self.own_next = _i32(ent_cap, d)
                    self.grp = _i32(int(self.lib.ggad_mb_group_words()) * ent_cap, d)
        if ck_cap != self.ck_cap:
            self.ck_cap = ck_cap
            self.chunk_part = _f32(ck_cap * self.part_stride, d)      # piece partials: x1 in the plan, relu(W x2) sums in the steps
        if stage_need > self.stage_cap:
            self.stage_cap = int(stage_need)
            self.stage_host = torch.empty(self.stage_cap, dtype=torch.int32)
            if d.type == "cuda":
                self.stage_host = self.stage_host.pin_memory()
            self.stage = _i32(self.stage_cap, d)
        if self.train and self.hop2 == "ldsw":
            if pairs > self.pair_cap:
                self.pair_cap = int(pairs * 1.25) + 1024
                self.pc = torch.empty(self.pair_cap, dtype=torch.int16, device=d)      # uint16 per-pair counts (raw storage)
            if items > self.item_cap:
                self.item_cap = int(items * 1.25) + 1024
                self.items = _i32(int(self.lib.ggad_mb_item_words()) * self.item_cap, d)
            if part2 > self.part2_cap:
                self.part2_cap = int(part2 * 1.25) + 64
                self.part2 = _f32(self.part2_cap * self.F, d)
            if seg > self.seg_cap:
                self.seg_cap = int(seg * 1.25) + 1024
                self.seg_t = _i32(self.seg_cap, d)
        self.generation += 1
        self._fill_descriptor()

    def _fill_descriptor(self) -> None:
        P, g = self.plan, self.g
        ldsw = self.train and self.hop2 == "ldsw"
        P.rowptr, P.col, P.feat = ptr(g.rowptr), ptr(g.col), ptr(self.feat)
        P.tile_off = ptr(g.tile_offsets(self.lib.ggad_mb_ldsw_tile_shift())) if ldsw else None
        if ldsw and os.environ.get("GGAD_TILE_MAJOR", "0") == "1":      # opt-in: built in round 5, measured SLOWER than the rows of col
            # (k_tile_counts 193 -> 216 us at 20 batches with the same order of workgroups, 287 with a tile's workgroups on one XCD;
            #  k_seg_transpose 21 -> 36 us: profiles/r05_tile_major.txt)
            ts, ct = g.tile_major(self.lib.ggad_mb_ldsw_tile_shift())
            P.tile_start, P.col_t = ptr(ts), ptr(ct)
        else:
            P.tile_start, P.col_t = None, None
        P.closed_deg_host = g.closed_deg_i32.ctypes.data
        P.pair_bound_host = g.pair_bound_host.ctypes.data if ldsw else None
        P.node_pack_host = g.node_pack_host.ctypes.data if ldsw else None
        P.stage_host, P.stage, P.stage_event = self.stage_host.data_ptr(), ptr(self.stage), self._stage_event
        P.cnt1, P.own1, P.cnt2 = ptr(self.cnt1), ptr(self.own1), ptr(self.cnt2)
        P.ent_col, P.ent_slot, P.ent_row = ptr(self.ent_col), ptr(self.ent_slot), ptr(self.ent_row)
        P.ent_own, P.ent_c1, P.x1, P.x2, P.ck_part = ptr(self.ent_own), ptr(self.ent_c1), ptr(self.x1), ptr(self.x2), ptr(self.chunk_part)
        if ldsw:
            P.own_deg, P.own_rp, P.pw_base, P.seg_t = ptr(self.own_deg), ptr(self.own_rp), ptr(self.pw_base), ptr(self.seg_t)
            P.node_head, P.own_next, P.grp, P.items = ptr(self.node_head), ptr(self.own_next), ptr(self.grp), ptr(self.items)
            P.pc, P.part2 = ptr(self.pc), ptr(self.part2)
        P.counters = ptr(self.counters)
        P.n_nodes, P.ent_cap, P.ck_cap, P.stage_cap = g.n, self.ent_cap, self.ck_cap, self.stage_cap
        P.pair_cap, P.item_cap, P.part2_cap, P.seg_cap = self.pair_cap, self.item_cap, self.part2_cap, self.seg_cap
        P.feat_dim, P.feat_stride, P.max_batches, P.rows_cap = self.F, self.stride, self.max_batches, self.rows_cap
        P.ck_part_stride = self.part_stride
        P.train, P.hop2, P.node_major = int(self.train), self.HOP2_MODES[self.hop2], int(self.node_major)
        P.mean_nbr_deg = float(g.mean_nbr_deg)
        P.xcd_skip = int(getattr(self, "xcd_skip", -1))

    # ---- build
    def build(self, batches: Sequence[np.ndarray], labels: Optional[Sequence[np.ndarray]] = None) -> None:
        """Upload the batches and run the plan + gather kernels on the current stream (one native call)."""
        nb = len(batches)
        if nb > self.max_batches or nb == 0:
            raise ValueError(f"chunk holds 1..{self.max_batches} batches, got {nb}")
        self._t_build0 = time.perf_counter()
        bp = self._bp_host
        # what `BatchSchedule.next_batches` hands out: consecutive rows of ONE contiguous int64 matrix (ids) and of another (labels)
        # -- no copy, no per-batch Python work (24 us per 20-batch build otherwise, on the critical path of a one-chunk run)
        nodes = _rows_as_flat(batches)
        if nodes is not None and (not self.train or labels is not None):
            lab = _rows_as_flat(labels) if self.train else None
            if not self.train or (lab is not None and len(lab) == len(nodes)):
                w = len(batches[0])
                bp[:nb + 1] = np.arange(0, (nb + 1) * w, w, dtype=np.int32) if w > 0 else 0
                if w == 0:
                    raise ValueError("empty batch")
                self.build_arrays(nodes, bp, nb, lab)
                return
        sizes = np.fromiter((len(b) for b in batches), dtype=np.int64, count=nb)
        if (sizes == 0).any():
            raise ValueError("empty batch")
        nodes = np.concatenate(batches).astype(np.int64, copy=False)          # (ids and labels are range-checked by the native call)
        lab = None
        if self.train:
            if labels is None:
                raise ValueError("training chunk needs labels")
            lab = np.concatenate(labels).astype(np.int64, copy=False)
            if len(lab) != len(nodes):
                raise ValueError("labels / batches length mismatch")
        bp[0] = 0
        np.cumsum(sizes, out=bp[1:nb + 1])
        self.build_arrays(nodes, bp, nb, lab)

    def build_arrays(self, nodes: np.ndarray, bp: np.ndarray, nb: int, lab: Optional[np.ndarray]) -> None:
        """`build` on already concatenated int64 node ids / labels and int32 batch offsets (no per-batch Python work)."""
        if bp is not self._bp_host:
            self._bp_host[:nb + 1] = bp[:nb + 1]
            bp = self._bp_host
        rows = int(bp[nb])
        if rows > self.rows_cap:
            self._alloc(rows, 0, 0, 0, 0, 0, 0, 0)
        P, I = self.plan, self.info
        P.xcd_skip = int(self.xcd_skip)
        if self.gather2_events is not None:
            P.ev_gather0, P.ev_gather1 = self.gather2_events
        else:
            P.ev_gather0 = P.ev_gather1 = None
        P.ev_tile0, P.ev_tile1 = self.tile_events if self.tile_events is not None else (None, None)
        stream = _lib.current_stream()
        for attempt in range(3):
            if _BUILD_TIMING:
                print("[BatchChunk.build] python before the native call %.1f us" % ((time.perf_counter() - self._t_build0) * 1e6), file=sys.stderr)
            # (`.ctypes.data` builds a ctypes object per call: ~1 us each on the host of a GPU box, six of them on the critical path of a
            # one-chunk run; the array interface is a third of that)
            rc = self.lib.ggad_mb_plan_build(ctypes.byref(P), _addr(nodes), _addr(bp), nb, _addr(lab) if lab is not None else None,
                                             ctypes.byref(I), _addr(self._ep_host), _addr(self._bep_host), _addr(self._bmr_host), stream)
            if rc != -3:
                break
            _check_i32(int(I.need_ents))
            if I.need_cnt2 and self.cnt2 is None:       # the LDS path cannot take this chunk: device-atomic counters
                self.cnt2 = torch.zeros(self.max_batches * self.g.n, dtype=torch.int32, device=self.dev)
            # the 2-hop buffers (pair counts, work items, partial sums, segment table) are sized for a build of `max_batches`
            # batches the first time they have to grow: a grow is a device synchronisation, a reallocation and a second host pass,
            # and a run that ramps 5 -> 20 -> 150 batches per build paid it inside every one of its first windows
            f = 1.1 * min(64.0, self.max_batches / max(1, nb)) if nb < self.max_batches else 1.0
            lim = 1_600_000_000                                   # (int32 offsets into pc[] / items[] / part2[])
            if self.dev.type == "cuda":
                # ... but never beyond a quarter of the memory that is free now: a small first build that happens to hold hub rows
                # extrapolates to tens of gigabytes (pair counts 2 B, items 8-48 B, partial sums 4 F B per unit); what this build
                # needs is always granted
                free = torch.cuda.mem_get_info(self.dev)[0]
                per_f = 2.0 * int(I.need_pairs) + 48.0 * int(I.need_items) + 4.0 * self.stride * int(I.need_part2) + 1.0
                f = max(1.0, min(f, 0.25 * free / per_f))
            self._alloc(I.need_rows, I.need_ents, I.need_chunks, I.need_stage, max(int(I.need_pairs), min(lim, int(I.need_pairs * f))),
                        max(int(I.need_items), min(lim // 16, int(I.need_items * f))),
                        max(int(I.need_part2), min(lim // 32, int(I.need_part2 * f))), max(int(I.need_seg), int(I.need_seg * min(f, self.ent_cap / max(1, int(I.need_ents))))))
        if rc == -1:
            raise ValueError("ggad_mb_plan_build: a batch node id is out of range, a label is not 0/1, or the plan descriptor is incomplete")
        _lib.check(rc, "ggad_mb_plan_build")
        self.n_batches, self.n_rows, self.n_ents, self.n_chunks = nb, int(I.n_rows), int(I.n_ents), int(I.n_chunks)
        self.last_hop2 = ("none", "ldsw", "global")[int(I.mode)]
        self.build_count += 1
        st = self.stage
        R, C = self.n_rows, self.n_chunks
        self.batch_ptr = st[I.off_batch_ptr:I.off_batch_ptr + nb + 1]
        self.batch_ent_ptr = st[I.off_batch_ent_ptr:I.off_batch_ent_ptr + nb + 1]
        self.nodes = st[I.off_nodes:I.off_nodes + R]
        self.labels = st[I.off_labels:I.off_labels + R]
        self.pos_meta = st[I.off_pos_meta:I.off_pos_meta + R]
        self.row_pos = st[I.off_row_pos:I.off_row_pos + R]
        self.row_slot = st[I.off_row_slot:I.off_row_slot + R]
        self.ent_ptr = st[I.off_ent_ptr:I.off_ent_ptr + R + 1]
        self.row_ck_ptr = st[I.off_row_ck_ptr:I.off_row_ck_ptr + R + 1]
        self.ck_rc = st[I.off_ck_rc:I.off_ck_rc + max(C, 1)]
        self.ck_e0 = st[I.off_ck_e0:I.off_ck_e0 + max(C, 1)]

    def reset(self) -> None:
        """Kept for callers of the previous interface: the plan leaves its counter slots clean by itself."""

    def owner_entries(self) -> torch.Tensor:
        """Entries that own their (batch, column): the reference's deduplicated `unique_nodes_list` (graphsage.py:306)."""
        e = torch.arange(self.n_ents, device=self.dev, dtype=torch.int32)
        return e[self.ent_own[:self.n_ents] == e].long()

    def ent_total_ptr(self) -> int:
        return self.ent_ptr.data_ptr() + 4 * self.n_rows

    def batch_rows(self, b: int):
        return int(self.batch_ptr_host[b]), int(self.batch_ptr_host[b + 1])

    def batch_ents(self, b: int):
        return int(self.batch_ent_host[b]), int(self.batch_ent_host[b + 1])

    def max_batch_ents(self) -> int:
        return int(np.diff(self.batch_ent_host).max()) if self.n_batches else 0


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
                 chain: int = 0, resident: Optional[bool] = None):
        """`resident`: run the dense steps of a chunk as ONE launch resident on one XCD (`ggad_mb_train_chunk_xcd`,
        csrc/step_xcd.hip) instead of 5 launches per step.  None = whenever the kernel supports the shapes (F == 17,
        D <= 64, chain 0) and GGAD_XCD is not 0."""
        self.lib = _lib.load()
        # 0: fused-forward step (5 launches, F == 17); 2: always the generic 6 launches (project -> fwd_rows -> ...)
        self.chain = int(chain)
        if self.chain not in (0, 2):
            raise ValueError("chain must be 0 or 2")
        self.F, self.D = int(feat_dim), int(embed_dim)
        import os
        can = int(feat_dim) == 17 and int(embed_dim) <= 64 and self.chain == 0
        if resident is None:
            resident = can and os.environ.get("GGAD_XCD", "1") != "0"
        elif resident and not can:
            raise ValueError("the XCD-resident chunk kernel needs F == 17, D <= 64 and chain 0")
        self.resident = bool(resident)
        self.xcd_ws = None
        self.xcd_wgs = 0                 # workgroups of the resident launch that stay (0 = 32: a whole XCD)
        self._xcd_rows = 0
        self._xcd_caps = (0, 0)
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
        if self.resident and (self.xcd_ws is None or self._xcd_rows < max_rows or self._xcd_caps[0] < ch.n_rows
                              or self._xcd_caps[1] < ch.n_chunks):
            self._xcd_rows = max(256, int(max_rows * 1.25), self._xcd_rows)
            # (the chunk's own capacities: a 5-batch warm-up followed by a 20-batch window otherwise re-allocated inside the window)
            self._xcd_caps = (max(int(ch.n_rows * 1.25) + 64, int(ch.rows_cap), self._xcd_caps[0]),
                              max(int(ch.n_chunks * 1.25) + 64, int(ch.ck_cap), self._xcd_caps[1]))
            old_ws = self.xcd_ws
            self.xcd_ws = torch.zeros(int(self.lib.ggad_mb_xcd_workspace_elems(self._xcd_rows, self.D, self.F, *self._xcd_caps)),
                                      dtype=torch.float32, device=self.dev)
            if old_ws is not None:
                # an error recorded by a launch on the old workspace (sticky word 255, `err` of its last launch: the first 256 floats
                # are the control block) must survive the growth -- the host reads the status once per run, not once per chunk
                self.xcd_ws[:256].copy_(old_ws[:256])
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
        s.chain = self.chain
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

    def train_chunk(self, ch: BatchChunk, allreduce=None, world_size: int = 1, log_base: int = 0, exchange=None) -> None:
        """One optimiser step per batch of the chunk (src/model_handler.py:330-364).  Data parallel: `exchange` (a connected
        `OneShotExchange`: gradients written straight into the peers' buffers, summed in rank order inside the Adam launch,
        no host call in the loop) or `allreduce` (a callable, e.g. RCCL through torch.distributed, served by a callback)."""
        self.ensure_capacity(ch, log_base + ch.n_batches)
        stream = _lib.current_stream()
        if self.resident and allreduce is None and (world_size == 1 or exchange is not None) \
                and getattr(ch, "row_ck_ptr", None) is not None:
            s = self.step_desc(ch, 0, log_base)
            # records prepared on the plan's stream (xcd_prepare) for THIS build and with the capacities the launch will use?
            recs = None
            if getattr(ch, "xcd_prepared", None) == (ch.build_count, self._xcd_caps):
                recs = ptr(ch.xcd_recs)
            _lib.check(self.lib.ggad_mb_train_chunk_xcd(ctypes.byref(s), ch.n_batches, ptr(ch.batch_ptr), self._xcd_rows,
                                                         ch.n_rows, ch.n_chunks, ch.n_ents, self._xcd_caps[0], self._xcd_caps[1],
                                                         self.loss_log.data_ptr(), log_base, ptr(self.xcd_ws), 1.0 / world_size,
                                                         exchange.handle if exchange is not None else None, int(self.xcd_wgs), recs,
                                                         stream),
                       "ggad_mb_train_chunk_xcd")
            return
        if exchange is not None:
            s = self.step_desc(ch, 0, log_base)
            bp, ep, mr = ch._bp_host, ch._bep_host, ch._bmr_host
            _lib.check(self.lib.ggad_mb_train_chunk_xchg(ctypes.byref(s), ch.n_batches, bp.ctypes.data, ep.ctypes.data,
                                                         mr.ctypes.data, self.loss_log.data_ptr(), log_base, 1.0 / world_size,
                                                         exchange.handle, stream), "ggad_mb_train_chunk_xchg")
            return
        fuse = 1 if (allreduce is None and world_size == 1) else 0
        if fuse:
            # single GPU: the whole chunk in one host call (the C loop issues the launches; Python per step costs more
            # than the 5-8 us kernels it feeds)
            s = self.step_desc(ch, 0, log_base)
            bp, ep, mr = ch._bp_host, ch._bep_host, ch._bmr_host        # host tables written by the plan
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
        bp, ep, mr = ch._bp_host, ch._bep_host, ch._bmr_host
        rc = self.lib.ggad_mb_train_chunk_dp(ctypes.byref(s), ch.n_batches, bp.ctypes.data, ep.ctypes.data, mr.ctypes.data,
                                             self.loss_log.data_ptr(), log_base, scale, cb, None, stream)
        if err:
            raise err[0]
        _lib.check(rc, "ggad_mb_train_chunk_dp")

    def xcd_prepare(self, ch: BatchChunk) -> None:
        """Records of the chunk for the XCD-resident launch (`ggad_mb_xcd_prepare`: a 32-byte record per piece / position, x2
        completed per entry) on the CURRENT stream -- the plan's, right behind `ch.build`: a whole-chip pass that took 0.45 ms per
        chunk on the chunk kernel's 28 compute units.  Without this call `train_chunk` builds them inside its own launch."""
        if not (self.resident and ch.train and getattr(ch, "row_ck_ptr", None) is not None and ch.n_batches > 0):
            return
        if os.environ.get("GGAD_XCD_PREP_ON_PLAN", "1") == "0":       # (A/B switch: records inside the chunk kernel's own launch)
            return
        caps = self._xcd_caps
        if self.xcd_ws is None or caps[0] < ch.n_rows or caps[1] < ch.n_chunks:
            return        # the launch's workspace has to grow first: `train_chunk` does that on ITS stream (an allocation made here,
                          # under the plan's stream, could be recycled while the other stream's kernel still uses it)
        need = int(self.lib.ggad_mb_xcd_record_elems(*caps))
        recs = getattr(ch, "xcd_recs", None)
        if recs is None or recs.numel() < need:
            if recs is not None:
                torch.cuda.synchronize(self.dev)             # a launch on the other stream may still read the old block
            ch.xcd_recs = torch.empty(need, dtype=torch.int32, device=self.dev)
        s = self.step_desc(ch, 0, 0)
        _lib.check(self.lib.ggad_mb_xcd_prepare(ctypes.byref(s), ch.n_batches, ptr(ch.batch_ptr), ch.n_rows, ch.n_chunks, ch.n_ents,
                                                caps[0], caps[1], ptr(ch.xcd_recs), _lib.current_stream()), "ggad_mb_xcd_prepare")
        ch.xcd_prepared = (ch.build_count, caps)

    def xcd_status(self) -> dict:
        """Control words of the last XCD-resident launch (synchronises the current stream): error code, surviving workgroups,
        their XCD, and rank 0's wall time per phase in microseconds (A, barrier, R, barrier, C, barrier, E, barrier)."""
        if self.xcd_ws is None:
            raise RuntimeError("no XCD-resident launch has run")
        out = (ctypes.c_int64 * 19)()
        _lib.check(self.lib.ggad_mb_xcd_status(ptr(self.xcd_ws), out, _lib.current_stream()), "ggad_mb_xcd_status")
        names = ("A", "bar1", "R", "bar2", "C", "bar3", "E", "bar4")
        return dict(error=int(out[0]), workgroups=int(out[1]), xcc=int(out[2]),
                    phase_us={n: out[3 + k] / 100.0 for k, n in enumerate(names)},
                    sub_us=[out[11 + k] / 100.0 for k in range(8)])

    def set_resident_error(self, code: int = 0) -> None:
        """Clear the sticky error word of the resident kernel's workspace (code 0), or set it as a timed-out launch would (tests)."""
        if self.xcd_ws is not None:
            _lib.check(self.lib.ggad_mb_xcd_clear_error(ptr(self.xcd_ws), int(code), _lib.current_stream()), "ggad_mb_xcd_clear_error")

    def check_resident(self) -> None:
        """Raise if the last XCD-resident launch timed out at a barrier (it leaves instead of hanging the GPU)."""
        if self.xcd_ws is not None:
            st = self.xcd_status()
            if st["error"]:
                raise RuntimeError(f"ggad_amd: XCD-resident chunk kernel timed out (code {st['error']}, {st['workgroups']} workgroups)")

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
