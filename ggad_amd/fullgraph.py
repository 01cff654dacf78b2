"""Full-graph GGAD on one MI355X: sparse structures + autograd wrappers over the HIP kernels.

The reference keeps `adj` and `raw_adj` as dense (1,N,N) tensors and multiplies them densely
(`run.py:98-110`, `model.py:31,151-155`, `run.py:182-188`).  Here they are CSR in HBM
(`FullGraphAdj`), every product is an edge-parallel HIP kernel and autograd only chains those kernels:

    GcnLayerFn   X W^T (MFMA GEMM) -> A_hat . (CSR SpMM, fused bias + PReLU) ; backward = PReLU' , A_hat^T SpMM, GEMMs
    LinearFn     nn.Linear without bias (+ fused ReLU)                       ; backward = two GEMMs
    SpmmRowsFn   A_hat[abn, :] @ emb  (outlier generation, model.py:151-155) ; backward = static transposed sub-CSR
    GgadLossFn   BCE + local-affinity margin + reconstruction (run.py:165-210), affinity as
                 aff_j = r_inv_j <e_hat_j, (R^T e_hat)_j>  evaluated only where the loss reads it
"""
from __future__ import annotations

import os

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, ptr_rows


def _dev_i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def _dev_f32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


class Csr:
    """Device CSR (int32 indices, fp32 values) with sorted columns."""

    def __init__(self, mat, dev):
        m = mat.tocsr().copy()
        m.sum_duplicates()
        m.sort_indices()
        if m.nnz >= 2 ** 31:
            raise ValueError("matrix too large for int32 CSR")
        self.shape = m.shape
        self.nnz = int(m.nnz)
        self.rowptr = _dev_i32(m.indptr, dev)
        self.col = _dev_i32(m.indices, dev)
        self.val = _dev_f32(m.data.astype(np.float32), dev)     # run.py:103-109 casts the fp64 values to fp32
        self.host = m
        self.dev = dev
        self._plans = {}

    def entries(self) -> torch.Tensor:
        """(column, bits of the value) per stored entry, interleaved int32 (what k_spmm_rowline fetches with one 8-byte load)."""
        e = self._plans.get("entries")
        if e is None:
            e = self._plans["entries"] = torch.stack((self.col, self.val.view(torch.int32)), 1).contiguous()
        return e

    def plan(self, rows_sel: Optional[np.ndarray] = None, key=None, seg: Optional[int] = None, col_ranges: int = 1):
        """Segment tables for ggad_spmm_csr_f32 / ggad_spmm_sliced_f32 (whole matrix, or the row subset `rows_sel`); cached.
        `seg`: segment length (default: the 64 of the wave-per-segment kernel).  `col_ranges` > 1: segments never cross
        the boundaries of that many equal column ranges and are launched range by range (columns are sorted inside a row,
        so a row is cut where its columns cross a boundary) -- the gathered operand rows of one phase then fit an L2."""
        key = "all" if rows_sel is None else key
        if key is not None and (seg is not None or col_ranges > 1):
            key = (key, "seg", seg, col_ranges)
        p = self._plans.get(key) if key is not None else None
        if p is not None:
            return p
        seg = int(_lib.load().ggad_spmm_seg_len()) if seg is None else int(seg)
        rp = self.host.indptr.astype(np.int64)
        rows = np.arange(self.shape[0], dtype=np.int64) if rows_sel is None else np.asarray(rows_sel, dtype=np.int64)
        beg, end = rp[rows], rp[rows + 1]
        if col_ranges > 1:
            # runs of (row, column range): cut positions inside every selected row
            width = -(-self.shape[1] // int(col_ranges))
            bounds = np.arange(1, int(col_ranges), dtype=np.int64) * width
            cnt = end - beg
            tot = int(cnt.sum())
            off = np.zeros(len(rows) + 1, dtype=np.int64)
            np.cumsum(cnt, out=off[1:])
            pos = np.repeat(beg - off[:-1], cnt) + np.arange(tot, dtype=np.int64)        # CSR position of every selected entry
            bucket = np.searchsorted(bounds, self.host.indices[pos], side="right")
            owner_e = np.repeat(np.arange(len(rows), dtype=np.int64), cnt)
            keyv = owner_e * int(col_ranges) + bucket
            start = np.flatnonzero(np.concatenate(([True], keyv[1:] != keyv[:-1]))) if tot else np.zeros(0, dtype=np.int64)
            run_beg = pos[start] if tot else np.zeros(0, dtype=np.int64)
            run_len = np.diff(np.concatenate((start, [tot]))) if tot else np.zeros(0, dtype=np.int64)
            run_row, run_bucket = owner_e[start], bucket[start]
            empty = np.flatnonzero(cnt == 0)                                              # rows without entries keep one empty segment
            run_beg = np.concatenate((run_beg, beg[empty]))
            run_len = np.concatenate((run_len, np.zeros(len(empty), dtype=np.int64)))
            run_row = np.concatenate((run_row, empty))
            run_bucket = np.concatenate((run_bucket, np.zeros(len(empty), dtype=np.int64)))
            o = np.lexsort((run_bucket, run_row))                                         # row-major: slots of a row are consecutive
            run_beg, run_len, run_row, run_bucket = run_beg[o], run_len[o], run_row[o], run_bucket[o]
        else:
            run_beg, run_len, run_row = beg, end - beg, np.arange(len(rows), dtype=np.int64)
            run_bucket = np.zeros(len(rows), dtype=np.int64)
        npiece = np.maximum(1, (run_len + seg - 1) // seg)
        firstp = np.zeros(len(run_beg) + 1, dtype=np.int64)
        np.cumsum(npiece, out=firstp[1:])
        total = int(firstp[-1])
        run_of = np.repeat(np.arange(len(run_beg), dtype=np.int64), npiece)              # run of every segment (slot order)
        k = np.arange(total, dtype=np.int64) - firstp[run_of]
        sbeg = run_beg[run_of] + k * seg
        send = np.minimum(run_beg[run_of] + run_len[run_of], sbeg + seg)
        owner = run_row[run_of]                                                           # output row of every segment
        nseg = np.bincount(owner, minlength=len(rows)).astype(np.int64)                  # segments per output row
        first = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(nseg, out=first[1:])                                                    # slot order == (row, range, piece) order
        single = nseg[owner] == 1
        seg_out = np.where(single, owner, -(np.arange(total, dtype=np.int64) + 1))       # < 0: partial sum slot = -seg_out - 1
        if col_ranges > 1:
            launch = np.argsort(run_bucket[run_of], kind="stable")                        # launch order: column range by range
            sbeg, send, seg_out = sbeg[launch], send[launch], seg_out[launch]
        multi = np.nonzero(nseg > 1)[0]
        dev = self.dev
        p = dict(seg_beg=_dev_i32(sbeg, dev), seg_end=_dev_i32(send, dev), seg_out=_dev_i32(seg_out, dev), n_seg=total,
                 multi_row=_dev_i32(multi, dev), multi_first=_dev_i32(first[multi], dev), multi_count=_dev_i32(nseg[multi], dev),
                 n_multi=int(len(multi)), n_out=int(len(rows)), part=None, nnz=int((end - beg).sum()),
                 rows=None if rows_sel is None else rows, long=None)
        if key is not None:
            self._plans[key] = p
        return p

    def rowslice_plan(self, p, lines: bool = False):
        """Units of the column-sliced kernel for sparse neighbourhoods (k_spmm_rowslice) for the rows of segment plan `p` (whole
        matrix or a row subset): rows sorted by length into groups of 6 (8 for the line-granular variant, `lines`), rows of more
        than `ggad_spmm_rowslice_long()` entries apart.  Cached on the plan."""
        slot = "rowline" if lines else "rowslice"
        rs = p.get(slot)
        if rs is not None:
            return rs
        lib = _lib.load()
        G, SHORT, LONG = int(lib.ggad_spmm_rowslice_group()), int(lib.ggad_spmm_rowslice_short()), int(lib.ggad_spmm_rowslice_long())
        if lines:
            G = 8
        rp = self.host.indptr.astype(np.int64)
        rows = np.arange(self.shape[0], dtype=np.int64) if p.get("rows") is None else np.asarray(p["rows"], dtype=np.int64)
        outr = np.arange(len(rows), dtype=np.int64)
        deg = rp[rows + 1] - rp[rows]
        is_short, is_hub = deg <= SHORT, deg > LONG
        is_med = ~is_short & ~is_hub
        order = np.argsort(-deg[is_short], kind="stable")
        sr, so = rows[is_short][order], outr[is_short][order]
        n_units = (len(sr) + G - 1) // G
        ur = np.full(n_units * G, -1, dtype=np.int64)
        uo = np.zeros(n_units * G, dtype=np.int64)
        ur[:len(sr)], uo[:len(so)] = sr, so
        mo = np.argsort(-deg[is_med], kind="stable")                    # longest first: they start first
        ho = np.argsort(-deg[is_hub], kind="stable")
        dev = self.dev
        if lines:                                                       # (first entry, end, output row, 0) per slot: k_spmm_rowline
            def tab(r, o):
                t = np.zeros((len(r), 4), dtype=np.int32)
                ok = r >= 0
                t[ok, 0], t[ok, 1] = rp[r[ok]], rp[r[ok] + 1]
                t[:, 2] = np.where(ok, o, -1)
                return torch.from_numpy(t.reshape(-1)).to(dev)
            rs = dict(unit_tab=tab(ur, uo), n_units=int(n_units), long_tab=tab(rows[is_med][mo], outr[is_med][mo]), n_long=int(is_med.sum()),
                      hub_tab=tab(rows[is_hub][ho], outr[is_hub][ho]), n_hub=int(is_hub.sum()))
        else:
            rs = dict(unit_rows=_dev_i32(ur, dev), unit_out=_dev_i32(uo, dev), n_units=int(n_units),
                      long_rows=_dev_i32(rows[is_med][mo], dev), long_out=_dev_i32(outr[is_med][mo], dev), n_long=int(is_med.sum()),
                      hub_rows=_dev_i32(rows[is_hub], dev), hub_out=_dev_i32(outr[is_hub], dev), n_hub=int(is_hub.sum()))
        p[slot] = rs
        return rs

    def value_factors(self):
        """(rs, cs, diag) such that value[i][j] = rs[i] * cs[j] off the diagonal (to fp32 round-off) and value[i][i] = diag[i] --
        the shape `normalize_adj` (`utils.py:47-54`: D^-1/2 A D^-1/2 of a 0/1 matrix, with or without self loops, `+ I`
        afterwards or not) gives every adjacency of this code base.  (None, None, None): all stored values are 1.
        False: the values do not factor (the LDS-panel product is then not used).  Host arrays, cached."""
        f = self._plans.get("factors")
        if f is not None:
            return f
        m = self.host
        val = np.ascontiguousarray(m.data, dtype=np.float32)
        f = False
        if np.all(val == np.float32(1.0)):
            f = (None, None, None)
        elif m.shape[0] == m.shape[1]:
            n = m.shape[0]
            lib = _lib.load()
            rowptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
            colv = np.ascontiguousarray(m.indices, dtype=np.int32)
            cnt = np.diff(rowptr)
            diag = m.diagonal().astype(np.float32)
            n_off = cnt - (diag != 0)
            for degree in (n_off, cnt):                                   # normalize_adj(A) [+ I]  /  normalize_adj(A + I)
                with np.errstate(divide="ignore"):
                    r = np.power(degree.astype(np.float64), -0.5)
                r[np.isinf(r)] = 0.0
                ok = int(lib.ggad_spmm_panel_values_factor(rowptr.ctypes.data, colv.ctypes.data, val.ctypes.data, r.ctypes.data, n,
                                                           4e-7, 0))
                if ok < 0:
                    raise _lib.GgadKernelError("ggad_spmm_panel_values_factor: invalid arguments")
                if ok == 1:
                    r32 = r.astype(np.float32)
                    f = (r32, r32, diag if np.any(diag != 0) else None)
                    break
        self._plans["factors"] = f
        return f

    def panel_plan(self, n_slices: int, rows_sel: Optional[np.ndarray] = None, cache: Optional[dict] = None):
        """Entry stream, directory, row table and workgroup table of `ggad_spmm_panel_f32` (layout described at k_spmm_panel in
        fullgraph.hip) for the whole matrix or the row subset `rows_sel` (output row i = matrix row rows_sel[i]), or None when
        the values do not factor / the step slots would be less than 40 % full / a row subset has a separate diagonal.
        Built in the library (0.1 s at 21 M entries), cached per slice count (in `cache`, default: on the matrix)."""
        key = ("panel", int(n_slices))
        store = self._plans if cache is None else cache
        if key in store:
            return store[key]
        lib = _lib.load()
        R, NW, KR = int(lib.ggad_spmm_panel_rows()), int(lib.ggad_spmm_panel_waves()), int(lib.ggad_spmm_panel_rounds())
        fac = self.value_factors()
        plan = None
        if fac is not False and (rows_sel is None or fac[2] is None):
            plan = self._build_panel(int(n_slices), R, NW, KR, fac, rows_sel)
        store[key] = plan
        return plan

    def _build_panel(self, n_slices, R, NW, KR, fac, rows_sel=None):
        import heapq
        lib = _lib.load()
        m = self.host
        n_rows, n_src = m.shape
        rs, cs, diag = fac
        skip_diag = 1 if diag is not None else 0                          # the diagonal is applied in the epilogue
        rowptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        colv = np.ascontiguousarray(m.indices, dtype=np.int32)
        cnt = np.diff(rowptr)
        if skip_diag:
            cnt = cnt - (m.diagonal() != 0)
        rows = None if rows_sel is None else np.asarray(rows_sel, dtype=np.int64)
        if rows is not None:                                              # output row i = matrix row rows[i]
            cnt = cnt[rows]
            n_rows = len(rows)
        nnz = int(cnt.sum())
        if nnz == 0:
            return None
        order = np.argsort(-cnt, kind="stable")                           # rows of similar length share a round
        NC = (n_src + R - 1) // R
        # Hub rows get a round of their own, WIDE: the row's entries are dealt over all 8 lane groups and the kernel adds the 8
        # accumulators in its epilogue.  A round of 8 hub rows of 7,000 entries is 7,000 steps for ONE wave -- 1.4 times the share
        # of a wave on the T-Finance-size graph, so the slowest wave of a panel had 1.5 x the mean work however the rounds were
        # dealt; as 8 wide rounds of 900 steps the same work spreads over 8 waves.  Wide: rows longer than half a wave's share.
        nb0 = max(1, 256 // n_slices)
        while -(-((n_rows + 7) // 8) // (nb0 * NW)) > KR:
            nb0 += max(1, 256 // n_slices)
        share = nnz / 8.0 / (nb0 * NW)                                    # steps of a wave if all slots were full
        n_wide = int(np.searchsorted(-cnt[order], -max(256.0, share / 2.0), side="left"))
        # (bounded: a wide row takes a round slot of its own, and the workgroup count -- every workgroup stages the whole operand --
        # must not grow for it: with 76 instead of 51 row blocks the T-Finance product took 654 instead of 573 us)
        room = nb0 * NW * KR - (n_rows + 7) // 8
        n_wide = max(0, min(n_wide, int(room * 8 // 7 * 0.9)))
        n_norm_rounds = (n_rows - n_wide + 7) // 8
        n_rounds = n_wide + n_norm_rounds
        nb = kr = None
        for mult in range(1, 65):                                         # workgroups ~ a multiple of the 256 CUs
            nb = max(1, (256 * mult) // n_slices)
            kr = -(-n_rounds // (nb * NW))
            if kr <= KR:
                break
        if kr is None or kr > KR:
            return None
        src_rows = (order if rows is None else rows[order]).astype(np.int32)     # matrix row of every output row, longest first
        round_rows = np.full((n_rounds, 8), -1, dtype=np.int32)           # matrix rows of a round (lane group = position)
        round_out = np.full((n_rounds, 8), -1, dtype=np.int32)            # output rows of a round (row_tab)
        WIDE = 0x40000000
        round_rows[:n_wide, :] = src_rows[:n_wide, None]
        round_out[:n_wide, :] = (order[:n_wide, None] | WIDE).astype(np.int32)
        rest = n_rows - n_wide
        round_rows[n_wide:].reshape(-1)[:rest] = src_rows[n_wide:]
        round_out[n_wide:].reshape(-1)[:rest] = order[n_wide:].astype(np.int32)
        round_wide = np.zeros(n_rounds, dtype=np.int32)
        round_wide[:n_wide] = 1
        round_rows = np.ascontiguousarray(round_rows.reshape(-1))
        steps_rc = np.empty(n_rounds * NC, dtype=np.int32)
        hp = lambda a: a.ctypes.data
        _lib.check(lib.ggad_spmm_panel_count(hp(rowptr), hp(colv), n_rounds, hp(round_rows), hp(round_wide), skip_diag, R, NC,
                                             hp(steps_rc), 0), "ggad_spmm_panel_count")
        # work of a round = its octs over all panels (the longest of its 8 rows counts).  A workgroup waits at two barriers per
        # panel for its slowest wave, and the rounds of the hub rows are many times longer than the others: rounds are dealt to the
        # workgroups in a snake over the work order and inside a workgroup to the least loaded wave that has a free slot
        # (longest first) -- dealt in rank order, the slowest wave of a panel had 2.2 x the mean work on the T-Finance-size graph
        # (1.5 x with 8-row rounds only: the longest round alone was 1.4 mean wave loads -- hence the wide rounds above).
        octs_rc = (steps_rc.astype(np.int64) + 7) // 8
        quads_rc = (steps_rc.astype(np.int64) + 3) // 4                   # what the kernel walks: whole octs, then half of the last one
        work = quads_rc.reshape(n_rounds, NC).sum(1)
        by_work = np.argsort(-work, kind="stable")
        pos_w = np.arange(n_rounds, dtype=np.int64)
        lap, idx = pos_w // nb, pos_w % nb
        blk_of_round = np.empty(n_rounds, dtype=np.int64)
        blk_of_round[by_work] = np.where(lap % 2 == 0, idx, nb - 1 - idx)
        wave_of_round = np.empty(n_rounds, dtype=np.int64)
        k_of_round = np.empty(n_rounds, dtype=np.int64)
        members = [[] for _ in range(nb)]
        for r_ in by_work.tolist():                                       # every workgroup's rounds, longest first
            members[blk_of_round[r_]].append(r_)
        work_l = work.tolist()
        for b in range(nb):
            heap = [(0, w) for w in range(NW)]
            used = [0] * NW
            for r_ in members[b]:
                load, w = heapq.heappop(heap)
                wave_of_round[r_], k_of_round[r_] = w, used[w]
                used[w] += 1
                if used[w] < KR:
                    heapq.heappush(heap, (load + work_l[r_], w))
        gwave_of_round = blk_of_round * NW + wave_of_round                # (block, wave)
        n_tiles = nb * NW * NC * KR
        tq = np.zeros(n_tiles, dtype=np.int64)                            # octs (8 steps = one 16-byte load per lane) per tile
        tile_of_rc = ((gwave_of_round[:, None] * NC + np.arange(NC, dtype=np.int64)[None, :]) * KR + k_of_round[:, None]).reshape(-1)
        tq[tile_of_rc] = octs_rc
        th = np.zeros(n_tiles, dtype=np.int64)                            # quads (4 steps) walked per tile: 2 * octs or 2 * octs - 1
        th[tile_of_rc] = quads_rc
        total_q = int(tq.sum())
        fill = nnz / float(max(1, int(th.sum())) * 32)
        if fill < 0.4 or total_q + 8 >= 2 ** 28 or int(th.max()) > 0xffff:
            return None
        offq = np.zeros(n_tiles + 1, dtype=np.int64)
        np.cumsum(tq, out=offq[1:])
        tile_oct = np.ascontiguousarray(offq[tile_of_rc])
        stream = np.empty((total_q + 8) * 64, dtype=np.uint16)            # [oct][lane group][step]: panel row index; 8 spare octs (read-ahead)
        _lib.check(lib.ggad_spmm_panel_fill(hp(rowptr), hp(colv), n_rounds, hp(round_rows), hp(round_wide), skip_diag, R, NC,
                                            hp(steps_rc), hp(tile_oct), hp(stream), total_q, 8, 0), "ggad_spmm_panel_fill")
        stream = stream.view(np.uint32)
        tq2 = th.reshape(nb * NW * NC, KR).astype(np.uint32)             # the directory counts quads
        dirv = np.zeros((nb * NW * NC, 8), dtype=np.uint32)
        dirv[:, 0] = offq[:-1].reshape(nb * NW * NC, KR)[:, 0].astype(np.uint32)
        for k in range(KR):
            dirv[:, 1 + (k >> 1)] |= tq2[:, k] << np.uint32(16 * (k & 1))
        row_tab = np.full((nb * NW * KR, 8), -1, dtype=np.int32)
        row_tab[gwave_of_round * KR + k_of_round] = round_out             # (a wide round: its row in all 8 slots, flagged)
        # workgroup table: the workgroups of a slice share an XCD (workgroup b runs on XCD b % 8) and its L2; the slices of an
        # incomplete round of 8 are dealt over all XCDs
        lists = [[] for _ in range(8)]
        full = (n_slices // 8) * 8
        for sl in range(full):
            lists[sl % 8].extend((sl, b) for b in range(nb))
        rest = [(sl, b) for b in range(nb) for sl in range(full, n_slices)]
        for i, it in enumerate(rest):
            lists[i % 8].append(it)
        L = max(len(x) for x in lists)
        wg = np.full((L * 8, 2), -1, dtype=np.int32)
        for x in range(8):
            if lists[x]:
                wg[x + 8 * np.arange(len(lists[x]))] = np.asarray(lists[x], dtype=np.int32)
        dev = self.dev
        as_i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
        if rs is not None and rows is not None:
            rs = rs[rows]
        return dict(wg=_dev_i32(wg, dev), n_wg=int(L * 8), dir=as_i32(dirv), stream=as_i32(stream), row_tab=_dev_i32(row_tab, dev),
                    n_chunks=int(NC), rs=None if rs is None else _dev_f32(rs, dev), cs=None if cs is None else _dev_f32(cs, dev),
                    diag=None if diag is None else _dev_f32(diag, dev), fill=fill, blocks=int(nb), rounds=int(kr))


    def ring_plan(self, n_slices: int, rows_sel: Optional[np.ndarray] = None, cache: Optional[dict] = None):
        """Quad stream, control bytes, per-wave extents, row table and workgroup table of `ggad_spmm_ring_f32` (layout described at
        k_spmm_ring in fullgraph.hip and in csrc/spmm_ring_build.cpp) for the whole matrix or the row subset `rows_sel`, or None
        under the conditions of `panel_plan`.  Cached per slice count."""
        key = ("ring", int(n_slices), os.environ.get("GGAD_RING_XCD", "slice"))
        store = self._plans if cache is None else cache
        if key in store:
            return store[key]
        fac = self.value_factors()
        plan = None
        if fac is not False and (rows_sel is None or fac[2] is None):
            lib = _lib.load()
            plan = self._build_ring(int(n_slices), fac, rows_sel, int(lib.ggad_spmm_ring_walkers()))
            # round 6: ONE loader wave is issue-bound (~100 clocks per 1-KB LDS-DMA instruction: ~2 us per 416-row slot).  The products over
            # a row subset and their transposes (10-20 steps per walker and phase) waited for it outright -- 276 -> 153 us (T-Finance loss
            # rows), 92 -> 52 us (Amazon) with three loader waves and 13 walkers -- and so did the whole-matrix product at Amazon size
            # (56 steps per phase: 112.6 -> 93.8 us); at T-Finance size (91 steps) 13 walkers are 2 % faster than 15.  Same-box sweep,
            # profiles/r06_ring_loaders_ab.log: epochs 2.013 -> 1.868 ms (T-Finance), 0.730 -> 0.639 (Amazon).  So every plan takes the
            # three-loader variant; GGAD_RING_SUBSET_STEPS = the steps per walker and phase below which a plan does (0: never = round 5).
            lim = float(os.environ.get("GGAD_RING_SUBSET_STEPS", "1e9"))
            if plan is not None and plan["steps_per_phase"] < lim and int(lib.ggad_spmm_ring_walkers_subset()) != plan["walkers"]:
                alt = self._build_ring(int(n_slices), fac, rows_sel, int(lib.ggad_spmm_ring_walkers_subset()))
                if alt is not None:
                    plan = alt
        store[key] = plan
        return plan

    def _build_ring(self, n_slices, fac, rows_sel=None, NW=None):
        import heapq
        lib = _lib.load()
        RS, S, V = int(lib.ggad_spmm_ring_slot_rows()), int(lib.ggad_spmm_ring_slots()), int(lib.ggad_spmm_ring_window())
        KR = int(lib.ggad_spmm_ring_rounds())
        NW = int(lib.ggad_spmm_ring_walkers()) if NW is None else int(NW)
        m = self.host
        n_rows, n_src = m.shape
        rs, cs, diag = fac
        skip_diag = 1 if diag is not None else 0                          # the diagonal is applied in the epilogue
        rowptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        colv = np.ascontiguousarray(m.indices, dtype=np.int32)
        cnt = np.diff(rowptr)
        if skip_diag:
            cnt = cnt - (m.diagonal() != 0)
        rows = None if rows_sel is None else np.asarray(rows_sel, dtype=np.int64)
        if rows is not None:                                              # output row i = matrix row rows[i]
            cnt = cnt[rows]
            n_rows = len(rows)
        nnz = int(cnt.sum())
        if nnz == 0:
            return None
        order = np.argsort(-cnt, kind="stable")                           # rows of similar length share a round
        NP = (n_src + RS - 1) // RS                                       # phases = slots of the operand
        # hub rows get a WIDE round of their own (the row over all 8 lane groups), as in the panel plan: rows longer than half a
        # wave's share, bounded by the accumulator slots that are free
        nb0 = max(1, 256 // n_slices)
        while -(-((n_rows + 7) // 8) // (nb0 * NW)) > KR:
            nb0 += max(1, 256 // n_slices)
        share = nnz / 8.0 / (nb0 * NW)
        n_wide = int(np.searchsorted(-cnt[order], -max(256.0, share / 2.0), side="left"))
        room = nb0 * NW * KR - (n_rows + 7) // 8
        n_wide = max(0, min(n_wide, int(room * 8 // 7 * 0.9)))
        n_norm_rounds = (n_rows - n_wide + 7) // 8
        n_rounds = n_wide + n_norm_rounds
        nb = kr = None
        for mult in range(1, 65):                                         # workgroups ~ a multiple of the 256 CUs
            nb = max(1, (256 * mult) // n_slices)
            kr = -(-n_rounds // (nb * NW))
            if kr <= KR:
                break
        if kr is None or kr > KR:
            return None
        src_rows = (order if rows is None else rows[order]).astype(np.int32)
        round_rows = np.full((n_rounds, 8), -1, dtype=np.int32)
        round_out = np.full((n_rounds, 8), -1, dtype=np.int32)
        WIDE = 0x40000000
        round_rows[:n_wide, :] = src_rows[:n_wide, None]
        round_out[:n_wide, :] = (order[:n_wide, None] | WIDE).astype(np.int32)
        rest = n_rows - n_wide
        round_rows[n_wide:].reshape(-1)[:rest] = src_rows[n_wide:]
        round_out[n_wide:].reshape(-1)[:rest] = order[n_wide:].astype(np.int32)
        round_wide = np.zeros(n_rounds, dtype=np.int32)
        round_wide[:n_wide] = 1
        round_rows = np.ascontiguousarray(round_rows.reshape(-1))
        quads = np.empty(n_rounds * NP, dtype=np.uint16)
        hp = lambda a: a.ctypes.data
        _lib.check(lib.ggad_spmm_ring_count(hp(rowptr), hp(colv), n_rounds, hp(round_rows), hp(round_wide), skip_diag, RS, S, V, NP,
                                            hp(quads), 0), "ggad_spmm_ring_count")
        quads = quads.reshape(n_rounds, NP)
        work = quads.astype(np.int64).sum(1)
        fill = nnz / float(max(1, int(work.sum())) * 32)
        if fill < 0.4:
            return None
        # rounds to workgroups in a snake over their work, inside a workgroup longest first to the least loaded walker with a free slot
        by_work = np.argsort(-work, kind="stable")
        pos_w = np.arange(n_rounds, dtype=np.int64)
        lap, ix = pos_w // nb, pos_w % nb
        blk_of_round = np.empty(n_rounds, dtype=np.int64)
        blk_of_round[by_work] = np.where(lap % 2 == 0, ix, nb - 1 - ix)
        wave_of_round = np.empty(n_rounds, dtype=np.int64)
        k_of_round = np.empty(n_rounds, dtype=np.int64)
        members = [[] for _ in range(nb)]
        for r_ in by_work.tolist():
            members[blk_of_round[r_]].append(r_)
        work_l = work.tolist()
        for b in range(nb):
            heap = [(0, w) for w in range(NW)]
            used = [0] * NW
            for r_ in members[b]:
                load, w = heapq.heappop(heap)
                wave_of_round[r_], k_of_round[r_] = w, used[w]
                used[w] += 1
                if used[w] < KR:
                    heapq.heappush(heap, (load + work_l[r_], w))
        gw_of_round = blk_of_round * NW + wave_of_round
        n_gw = nb * NW
        # stream of a walker: phase-major, inside a phase its rounds in accumulator order; a phase without work gets one dummy quad
        # (zero rows) that carries the end-of-phase flag -- every walker meets every barrier
        if n_gw * NP * (KR + 1) * 32 > (4 << 30):                         # the dense (walker, phase, slot) host tables below: ~32 B per cell
            return None                                                  # (a very large operand: the panel / sliced kernels take it)
        tab = np.zeros((n_gw, NP, KR + 1), dtype=np.int64)
        tab[gw_of_round, :, k_of_round] = quads
        tab[:, :, KR] = tab[:, :, :KR].sum(2) == 0
        flat = tab.reshape(n_gw, NP * (KR + 1))
        ends = np.cumsum(flat, axis=1)                                   # quads of the walker up to and including (phase, slot)
        tq = ends[:, -1]
        nsb = (tq + 3) // 4
        sb_off = np.zeros(n_gw + 1, dtype=np.int64)
        np.cumsum(nsb, out=sb_off[1:])
        total_sb = int(sb_off[-1])
        if total_sb * 128 >= 2 ** 31:
            return None
        starts = (ends - flat).reshape(n_gw, NP, KR + 1) + (sb_off[:-1] * 4)[:, None, None]
        quad_off = np.ascontiguousarray(starts[gw_of_round, :, k_of_round])         # (n_rounds, NP): absolute quad of every tile
        # control byte per quad: accumulator offset (4 * slot) in bits 0..5, bit 6 = last quad of its phase
        ctl = np.zeros((total_sb + 2) * 4, dtype=np.uint8)
        kbyte = np.tile(np.concatenate((np.arange(KR, dtype=np.int64) * 4, [0])), NP)
        for g_ in range(n_gw):
            base = int(sb_off[g_]) * 4
            ctl[base:base + int(tq[g_])] = np.repeat(kbyte, flat[g_]).astype(np.uint8)
            phase_end = ends[g_].reshape(NP, KR + 1)[:, -1] - 1 + base
            ctl[phase_end] |= 0x40
        idx = np.empty((total_sb + 2) * 128, dtype=np.uint16)            # two spare super-blocks: the walk requests its stream two ahead
        _lib.check(lib.ggad_spmm_ring_fill(hp(rowptr), hp(colv), n_rounds, hp(round_rows), hp(round_wide), skip_diag, RS, S, V, NP,
                                           hp(np.ascontiguousarray(quads.reshape(-1))), hp(quad_off.reshape(-1)), hp(idx), total_sb + 2, 0),
                   "ggad_spmm_ring_fill")
        wave_sb = np.stack((sb_off[:-1], nsb), axis=1).astype(np.int32)
        row_tab = np.full((n_gw * KR, 8), -1, dtype=np.int32)
        row_tab[gw_of_round * KR + k_of_round] = round_out
        # workgroup table (workgroup i runs on XCD i % 8).  "slice": the workgroups of a slice share an XCD and its L2 (slot loads hit);
        # "block": the workgroups of a row block do (their common quad stream hits)
        lists = [[] for _ in range(8)]
        if os.environ.get("GGAD_RING_XCD", "slice") == "block":
            allw = [(sl, b) for b in range(nb) for sl in range(n_slices)]
            q, rem = divmod(len(allw), 8)
            at = 0
            for x in range(8):
                take = q + (1 if x < rem else 0)
                lists[x] = allw[at:at + take]
                at += take
        else:
            full = (n_slices // 8) * 8
            for sl in range(full):
                lists[sl % 8].extend((sl, b) for b in range(nb))
            for i, it in enumerate([(sl, b) for b in range(nb) for sl in range(full, n_slices)]):
                lists[i % 8].append(it)
        L = max(len(x) for x in lists)
        wg = np.full((L * 8, 2), -1, dtype=np.int32)
        for x in range(8):
            if lists[x]:
                wg[x + 8 * np.arange(len(lists[x]))] = np.asarray(lists[x], dtype=np.int32)
        dev = self.dev
        if rs is not None and rows is not None:
            rs = rs[rows]
        t16 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(dev)
        per_phase = flat.reshape(n_gw, NP, KR + 1).sum(2)                 # quads per (walker, phase): the barrier waits for the longest
        return dict(wg=_dev_i32(wg, dev), n_wg=int(L * 8), wave_sb=_dev_i32(wave_sb, dev), idx=t16(idx),
                    ctl=torch.from_numpy(ctl.view(np.int32)).to(dev), row_tab=_dev_i32(row_tab, dev), n_phases=int(NP),
                    rs=None if rs is None else _dev_f32(rs, dev), cs=None if cs is None else _dev_f32(cs, dev),
                    diag=None if diag is None else _dev_f32(diag, dev), fill=fill, blocks=int(nb), rounds=int(kr),
                    quads=int(tq.sum()), phase_skew=float(per_phase.reshape(nb, NW, NP).max(1).sum() / max(1.0, per_phase.sum() / NW)),
                    walkers=int(NW), steps_per_phase=float(4.0 * tq.sum() / max(1, n_gw * NP)))


class FullGraphAdj:
    """Everything the full-graph step needs from the two adjacency matrices, built once on the host.

    ``adj_norm``  = normalize_adj(A) + I  (`utils.py:47-54`, `run.py:101`)   -> A_hat and A_hat^T
    ``raw``       = A + I                 (`run.py:100`)                      -> R^T (column view) and 1/colsum
    """

    def __init__(self, adj_norm, raw, device):
        import scipy.sparse as sp
        self.dev = torch.device(device)
        an = sp.csr_matrix(adj_norm)
        self.n = an.shape[0]
        self.A = Csr(an, self.dev)
        at = an.T.tocsr()
        self.symmetric = (abs(an - at).nnz == 0)
        self.At = self.A if self.symmetric else Csr(at, self.dev)
        rw = sp.csr_matrix(raw)
        self.raw_host = rw
        self.Rt = Csr(rw.T.tocsr(), self.dev)                   # row j of R^T = column j of raw_adj
        colsum = np.asarray(rw.astype(np.float32).sum(0)).reshape(-1).astype(np.float32)   # torch.sum(raw_adj, 0), fp32
        with np.errstate(divide="ignore"):
            r_inv = np.float32(1.0) / colsum
        r_inv[np.isinf(r_inv)] = 0.0                            # run.py:185-186
        self.r_inv_host = r_inv
        self._abn: Dict[tuple, tuple] = {}
        self._loss: Dict[tuple, tuple] = {}
        self._head: Dict[tuple, object] = {}
        self._ax: Dict[str, object] = {}                         # cached A_hat X of the constant input layer (`cached_aggregate`)
        self._devc: Dict[str, torch.Tensor] = {}                 # small per-node device vectors (`r_inv_dev`, `arange_dev`)

    def r_inv_dev(self) -> torch.Tensor:
        """1 / colsum(raw) (inf -> 0) on the device, one value per node."""
        t = self._devc.get("r_inv")
        if t is None:
            t = self._devc["r_inv"] = _dev_f32(self.r_inv_host, self.dev)
        return t

    def arange_dev(self) -> torch.Tensor:
        t = self._devc.get("arange")
        if t is None:
            t = self._devc["arange"] = torch.arange(self.n, dtype=torch.int32, device=self.dev)
        return t

    @classmethod
    def from_dense(cls, adj, raw_adj, device):
        import scipy.sparse as sp
        a = adj.detach().cpu().numpy() if isinstance(adj, torch.Tensor) else np.asarray(adj)
        r = raw_adj.detach().cpu().numpy() if isinstance(raw_adj, torch.Tensor) else np.asarray(raw_adj)
        return cls(sp.csr_matrix(a.reshape(a.shape[-2], a.shape[-1])), sp.csr_matrix(r.reshape(r.shape[-2], r.shape[-1])),
                   device)

    def abn_structs(self, abn_idx) -> Tuple[torch.Tensor, Csr]:
        """SpMM plan for the rows A_hat[abn, :] and the transposed sub-matrix (N x A) for its backward."""
        key = tuple(int(i) for i in abn_idx)
        s = self._abn.get(key)
        if s is None:
            idx = np.asarray(key, dtype=np.int64)
            sub = self.A.host[idx, :]
            s = (self.A.plan(idx, key=("rows", key)), Csr(sub.T.tocsr(), self.dev))
            self._abn[key] = s
        return s

    def head_structs(self, normal_idx, abn_idx):
        """Device index lists of the training head (`model.py:140-182`) and their inverse maps (position of node i in the list, -1 if
        absent) for `GgadHeadFn`; None when a list holds a node twice (the fused head assumes duplicate-free lists, as `run.py` draws them)."""
        nrm = np.ascontiguousarray(np.asarray(normal_idx, dtype=np.int64))         # C-speed key: no per-element Python work per forward
        abn = np.ascontiguousarray(np.asarray(abn_idx, dtype=np.int64))
        key = (nrm.size, abn.size, hash(nrm.tobytes()), hash(abn.tobytes()))
        s = self._head.get(key)
        if s is not None and s is not False and not (np.array_equal(s["nrm_host"], nrm) and np.array_equal(s["abn_host"], abn)):
            s = None                                                                 # (hash collision)
        if s is None:
            if len(np.unique(nrm)) != len(nrm) or len(np.unique(abn)) != len(abn) or len(abn) == 0:
                s = False
            else:
                nrm_pos = np.full(self.n, -1, dtype=np.int32)
                abn_pos = np.full(self.n, -1, dtype=np.int32)
                nrm_pos[nrm] = np.arange(len(nrm), dtype=np.int32)
                abn_pos[abn] = np.arange(len(abn), dtype=np.int32)
                rows_plan, sub_t = self.abn_structs(abn_idx)
                s = dict(nrm=_dev_i32(nrm, self.dev), abn=_dev_i32(abn, self.dev), nrm_pos=_dev_i32(nrm_pos, self.dev),
                         abn_pos=_dev_i32(abn_pos, self.dev), n_nrm=len(nrm), n_abn=len(abn), rows_plan=rows_plan, sub_t=sub_t,
                         nrm_host=nrm, abn_host=abn)
            if len(self._head) >= 8:
                self._head.clear()
            self._head[key] = s
        return s or None

    def loss_structs(self, normal_idx, abn_idx):
        """J = normal_idx ++ abn_idx (the rows whose affinity the loss reads) and R[:, J] as an N x |J| CSR."""
        key = (tuple(int(i) for i in normal_idx), tuple(int(i) for i in abn_idx))
        s = self._loss.get(key)
        if s is None:
            J = np.asarray(key[0] + key[1], dtype=np.int64)
            sub = self.Rt.host[J, :]                            # |J| x N : rows of R^T
            # position of every row in the normal / abnormal segment of J (-1: not there): the backward adds c_j S_j to the rows J inside
            # the row-normalisation's backward launch (round 6) instead of a scatter launch per segment
            n_rows = self.Rt.host.shape[1]
            pos_n = np.full(n_rows, -1, dtype=np.int32)
            pos_a = np.full(n_rows, -1, dtype=np.int32)
            nn_ = len(key[0])
            pos_n[J[:nn_]] = np.arange(nn_, dtype=np.int32)
            pos_a[J[nn_:]] = np.arange(nn_, len(J), dtype=np.int32)
            seg_unique = len(np.unique(J[:nn_])) == nn_ and len(np.unique(J[nn_:])) == len(J) - nn_
            s = dict(J=_dev_i32(J, self.dev), n_normal=len(key[0]), n_out=len(key[1]), distinct=bool(len(np.unique(J)) == len(J)),
                     r_inv_J=_dev_f32(self.r_inv_host[J], self.dev), RJ=Csr(sub.T.tocsr(), self.dev),
                     Rt_plan=self.Rt.plan(J, key=("rows", key)), pos_n=_dev_i32(pos_n, self.dev), pos_a=_dev_i32(pos_a, self.dev),
                     seg_unique=bool(seg_unique))
            self._loss[key] = s
        return s


# ------------------------------------------------------------------------------------------------ kernels
def gemm(A: torch.Tensor, B: torch.Tensor, trans_a: bool, trans_b: bool, bias=None, relu: bool = False, out=None) -> torch.Tensor:
    """C = op(A) op(B) on the matrix cores; A, B row-major 2-D fp32.  `out`: an (M, N) destination with unit column stride (its
    row stride may exceed N: `padded_rows`)."""
    A, B = A.contiguous(), B.contiguous()
    M, K = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    K2, N = (B.shape[1], B.shape[0]) if trans_b else (B.shape[0], B.shape[1])
    if K != K2:
        raise ValueError("gemm shape mismatch")
    sam, sak = (1, A.shape[1]) if trans_a else (A.shape[1], 1)
    sbk, sbn = (1, B.shape[1]) if trans_b else (B.shape[1], 1)
    C = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)
    if tuple(C.shape) != (M, N) or C.stride(1) != 1 or C.stride(0) < N or C.dtype != torch.float32:
        raise ValueError("gemm: out must be (M, N) fp32 with unit column stride")
    lib = _lib.load()
    ws_n = int(lib.ggad_gemm_workspace_elems(M, N, K))
    ws = torch.empty(ws_n, dtype=torch.float32, device=A.device) if ws_n else None
    call("ggad_gemm_f32", ptr(A), ptr(B), ptr_rows(C), M, N, K, sam, sak, sbk, sbn, C.stride(0), ptr(bias) if bias is not None else 0,
         1 if relu else 0, ptr(ws) if ws is not None else 0)
    return C


_XS_WORKSPACE = {}


def _use_sliced(csr: Csr, p, X: torch.Tensor) -> bool:
    """Dense neighbourhoods gathered from an operand that does not fit an XCD's L2: the XCD-sliced kernel (fullgraph.hip).
    GGAD_SPMM_SLICED=0 / 1 forces the choice (A/B measurements)."""
    force = os.environ.get("GGAD_SPMM_SLICED")
    if force is not None:
        return force == "1"
    w = X.shape[1]
    return w >= 64 and X.shape[0] * w * 4 >= (6 << 20) and p["n_seg"] > 0 and p.get("nnz", csr.nnz) >= 24 * p["n_seg"]


def _use_rowslice(csr: Csr, p, X: torch.Tensor) -> bool:
    """Sparse neighbourhoods gathered from a wide operand that no XCD's L2 holds (Reddit, Photo): the column-sliced six-rows-per-wave
    kernel (k_spmm_rowslice).  GGAD_SPMM_ROWSLICE=0 / 1 forces the choice (A/B measurements)."""
    force = os.environ.get("GGAD_SPMM_ROWSLICE")
    if force is not None:
        return force == "1"
    w = X.shape[1]
    # A smaller operand too when the rows are short (< 32 entries per output row: the transposed loss-row products of Reddit / Photo, a
    # 1.4-2.2 MB operand, took the segment kernel + its combine: Photo 0.555 -> 0.542 ms, Reddit 0.496 -> 0.488); with longer rows the
    # segment kernel stays (Amazon / T-Finance: +0.5 / +1.2 % with this kernel, scripts/rowslice_threshold_ab.sh)
    if w < 160:
        return False
    nnz = p.get("nnz", csr.nnz)
    # (short rows AND short columns: the transposed head product of T-Finance has 7 entries per output row but 540 per source row)
    return X.shape[0] * w * 4 >= (4 << 20) or (nnz < 32 * max(int(p["n_out"]), 1) and nnz < 64 * max(int(X.shape[0]), 1))


def _use_rowline(X: torch.Tensor) -> bool:
    """A row-strided operand the line-granular kernel (k_spmm_rowslice<8, 8, true>) takes as it is: 128-byte aligned rows."""
    return bool(_lib.load().ggad_spmm_rowline_supported(ptr_rows(X), X.stride(0), X.shape[1], X.shape[0]))


def padded_rows(n_rows: int, W: int, dev, consumer=None):
    """An (n_rows, W) fp32 matrix for a producer kernel to fill.  When the product that will read it -- `consumer` = (csr, plan) --
    is one of the sparse-neighbourhood products (what `_use_rowslice` selects: Reddit, Photo), its rows start on 128-byte lines
    (a view of a matrix with the row stride rounded up to 32 floats; the padding columns are never read) and `spmm` then runs the
    line-granular kernel; a plain contiguous matrix otherwise.  GGAD_SPMM_ROWLINE=0: always contiguous."""
    ld = (W + 31) // 32 * 32
    if consumer is not None and ld != W and os.environ.get("GGAD_SPMM_ROWLINE", "1") != "0":
        csr, p = consumer
        p = p if p is not None else csr.plan()
        X = torch.empty(n_rows, ld, dtype=torch.float32, device=dev)[:, :W]
        if csr.nnz > 0 and _use_rowline(X) and _use_panel(csr, p, X) is None and not _use_sliced(csr, p, X) and _use_rowslice(csr, p, X):
            return X
    return torch.empty(n_rows, W, dtype=torch.float32, device=dev)


def _use_panel(csr: Csr, p, X: torch.Tensor):
    """Products with dense neighbourhoods and factoring values (whole matrix, or a row subset of a matrix without separate
    diagonal): the LDS-panel kernel (k_spmm_panel).  Returns its plan or None.  GGAD_SPMM_PANEL=0 turns it off, =1 forces it wherever a plan can be built."""
    force = os.environ.get("GGAD_SPMM_PANEL")
    if force == "0" or X.shape[0] != csr.shape[1]:
        return None
    with torch.cuda.device(X.device):
        if not _lib.load().ggad_spmm_panel_available():          # a device without 159 KB of LDS per workgroup: the sliced kernel
            return None
    w = X.shape[1]
    nnz = p.get("nnz", csr.nnz)
    # (the floors were 2^20 entries and 64 per output row: the products over Amazon's loss rows -- 1,751 rows x 368 entries, and their transpose
    #  with 54 entries per row -- then took the sliced / segment kernels at ~80 ps per entry: epoch 0.725 against 0.694-0.698 ms with the ring,
    #  scripts/panel_threshold_ab.sh; sparse neighbourhoods (Reddit, Photo: 16 per row) fail the per-row test and keep their kernels)
    min_nnz = int(os.environ.get("GGAD_SPMM_PANEL_MIN_NNZ", str(1 << 18)))
    min_row = int(os.environ.get("GGAD_SPMM_PANEL_MIN_ROW", "32"))
    if force != "1" and not (w >= 64 and nnz >= min_row * p["n_out"] and nnz >= min_nnz):
        return None
    ring = os.environ.get("GGAD_SPMM_RING", "1") != "0" and bool(_lib.load().ggad_spmm_ring_available())
    # the LDS ring (k_spmm_ring) unless it is turned off or its plan does not qualify (fill, rounds, stream size): then the panels
    for build in ((Csr.ring_plan, Csr.panel_plan) if ring else (Csr.panel_plan,)):
        if p.get("rows") is not None:                                     # row subset: the plan lives on the segment plan
            plan = build(csr, (w // 4 + 7) // 8, p["rows"], p.setdefault("panel_cache", {}))
        else:
            plan = build(csr, (w // 4 + 7) // 8)
        if plan is not None:
            return plan
    return None


def _xs_workspace(X: torch.Tensor, n_ws: int) -> torch.Tensor:
    key = (str(X.device), torch.cuda.current_stream(X.device).cuda_stream)
    xs = _XS_WORKSPACE.get(key)
    if xs is None or xs.numel() < n_ws:          # one staging buffer per stream: consumed by the launch that follows its fill
        xs = _XS_WORKSPACE[key] = torch.empty(n_ws, dtype=torch.float32, device=X.device)
    return xs


def _part_buffer(p, W: int, dev):
    """Partial sums of the rows that are split into several segments (cached on the plan)."""
    if p["n_multi"] == 0:
        return None
    part = p.get("part")
    if part is None or part.numel() < p["n_seg"] * W:
        part = p["part"] = torch.empty(p["n_seg"] * W, dtype=torch.float32, device=dev)
    return part


def spmm(csr: Csr, X: torch.Tensor, plan=None, bias=None, prelu_a=None, want_pre=False, out=None):
    """out = act(csr[rows] @ X + bias); `plan` = csr.plan(...) selects the rows (default: all).  `out`: destination (and shape of
    the pre-activation copy) with unit column stride; its row stride may exceed W."""
    strided = X.dim() == 2 and X.stride(1) == 1 and X.stride(0) > X.shape[1] and csr.nnz > 0 and _use_rowline(X)     # padded rows: see padded_rows()
    if not strided:
        X = X.contiguous()
    W = X.shape[1]
    p = plan if plan is not None else csr.plan()
    if out is None:
        out = torch.empty(p["n_out"], W, dtype=torch.float32, device=X.device)
    elif tuple(out.shape) != (p["n_out"], W) or out.dtype != torch.float32:
        raise ValueError("spmm: out must be (rows, W) fp32")
    pre = torch.empty_strided(out.shape, out.stride(), dtype=torch.float32, device=X.device) if want_pre else None
    opt = (ptr(bias) if bias is not None else 0, ptr(prelu_a) if prelu_a is not None else 0, ptr_rows(out), out.stride(0),
           ptr_rows(pre) if pre is not None else 0)
    pp = None if strided else _use_panel(csr, p, X)
    if strided:
        rs = csr.rowslice_plan(p, lines=True)
        call("ggad_spmm_rowline_f32", ptr(csr.entries()), ptr(rs["unit_tab"]), rs["n_units"], ptr(rs["long_tab"]), rs["n_long"],
             ptr(rs["hub_tab"]), rs["n_hub"], ptr_rows(X), X.stride(0), W, X.shape[0], *opt)
    elif pp is not None:
        xs = _xs_workspace(X, int(_lib.load().ggad_spmm_sliced_workspace_elems(X.shape[0], W)))
        opt_p = lambda t: ptr(t) if t is not None else 0
        if "wave_sb" in pp:
            call("ggad_spmm_ring_f32", ptr(pp["wg"]), pp["n_wg"], ptr(pp["wave_sb"]), ptr(pp["idx"]), ptr(pp["ctl"]), ptr(pp["row_tab"]),
                 pp["n_phases"], pp["walkers"], opt_p(pp["cs"]), opt_p(pp["rs"]), opt_p(pp["diag"]), ptr(X), W, W, X.shape[0], ptr(xs), *opt)
        else:
            call("ggad_spmm_panel_f32", ptr(pp["wg"]), pp["n_wg"], ptr(pp["dir"]), ptr(pp["stream"]), ptr(pp["row_tab"]), pp["n_chunks"],
                 opt_p(pp["cs"]), opt_p(pp["rs"]), opt_p(pp["diag"]), ptr(X), W, W, X.shape[0], ptr(xs), *opt)
    elif _use_sliced(csr, p, X):
        if p["long"] is None:      # the sliced kernel walks long segments (8 index loads, one reduction and store per wave)
            # (GGAD_SPMM_COL_RANGES > 1 cuts them at column-range boundaries and launches range by range so that one phase's
            # operand rows fit an L2 -- measured slower, 1.35 -> 1.67 ms with 2 ranges on T-Finance: the sliced kernel is bound
            # by the L1 line rate, not by L2 capacity, and shorter segments cost more prologues; kept for experiments)
            ranges = max(1, int(os.environ.get("GGAD_SPMM_COL_RANGES", 1)))
            p["long"] = csr.plan(p["rows"], key=None, seg=int(_lib.load().ggad_spmm_sliced_seg_len()), col_ranges=ranges)
        p = p["long"]
        part = _part_buffer(p, W, X.device)
        xs = _xs_workspace(X, int(_lib.load().ggad_spmm_sliced_workspace_elems(X.shape[0], W)))
        call("ggad_spmm_sliced_f32", ptr(csr.col), ptr(csr.val), ptr(p["seg_beg"]), ptr(p["seg_end"]), ptr(p["seg_out"]),
             p["n_seg"], ptr(p["multi_row"]), ptr(p["multi_first"]), ptr(p["multi_count"]), p["n_multi"], ptr(X), W, W, X.shape[0],
             ptr(xs), *opt, ptr(part) if part is not None else 0)
    elif _use_rowslice(csr, p, X):
        rs = csr.rowslice_plan(p)
        call("ggad_spmm_rowslice_f32", ptr(csr.rowptr), ptr(csr.col), ptr(csr.val), ptr(rs["unit_rows"]), ptr(rs["unit_out"]), rs["n_units"],
             ptr(rs["long_rows"]), ptr(rs["long_out"]), rs["n_long"], ptr(rs["hub_rows"]), ptr(rs["hub_out"]), rs["n_hub"], ptr(X), W, W, *opt)
    else:
        part = _part_buffer(p, W, X.device)
        call("ggad_spmm_csr_f32", ptr(csr.col), ptr(csr.val), ptr(p["seg_beg"]), ptr(p["seg_end"]), ptr(p["seg_out"]), p["n_seg"],
             ptr(p["multi_row"]), ptr(p["multi_first"]), ptr(p["multi_count"]), p["n_multi"], ptr(X), W, W, *opt,
             ptr(part) if part is not None else 0)
    return (out, pre) if want_pre else out


# ------------------------------------------------------------------------------------------------ autograd
def _reorder_layer(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """A_hat (X W^T) == (A_hat X) W^T: when the layer's input is a constant (the feature matrix: no gradient) narrower than its
    output, A_hat X is computed once and cached, and the layer needs no SpMM at all -- neither forward nor backward (the weight
    gradient is dZ^T (A_hat X)).  T-Finance: F = 10 against H = 300, two of the five N x N x 300 products of an epoch go.
    Same operations, the two products associated the other way round: results agree to fp32 round-off.
    GGAD_GCN_REORDER=0 keeps the reference's order."""
    return (not x.requires_grad) and x.shape[1] < weight.shape[0] and os.environ.get("GGAD_GCN_REORDER", "1") != "0"


def cached_aggregate(adj: "FullGraphAdj", x: torch.Tensor) -> torch.Tensor:
    """A_hat X for a constant X (N x F), computed once per (X storage, version) with the SpMM kernel on zero-padded columns."""
    # the cache entry keeps a reference to the tensor: its storage stays alive, so the caching allocator cannot hand the same
    # address (with version 0) to another tensor of the same shape while the entry exists
    key = (x.data_ptr(), x._version, tuple(x.shape))
    hit = adj._ax.get("key") == key
    if not hit:
        f = x.shape[1]
        fp = (f + 3) // 4 * 4
        xp = x if fp == f else torch.cat((x, torch.zeros(x.shape[0], fp - f, device=x.device)), 1)
        with torch.no_grad():
            ax = spmm(adj.A, xp.contiguous())
        # a second copy with zero columns up to a multiple of 4 (at least 20: two 16-k steps of the slab kernel) when F is not one: the
        # layer's product then takes the slab kernel instead of the generic tiles (Amazon F = 25: 13.1 us, T-Finance F = 10: 23.4 us)
        axp = None
        if f % 4 != 0 and f <= 64 and os.environ.get("GGAD_PAD_FEATURES", "1") != "0":
            wp = max(fp, 20)
            axp = torch.zeros(x.shape[0], wp, dtype=torch.float32, device=x.device)
            axp[:, :f] = ax[:, :f]
        adj._ax = {"key": key, "x": x, "ax": ax[:, :f].contiguous(), "axp": axp}
    return adj._ax["ax"]


def padded_constant(adj: "FullGraphAdj", x: torch.Tensor):
    """A constant layer input (the feature matrix) whose width is not a multiple of 4 -- Photo's 745 -- with zero columns up to the next
    one, made once per (storage, version): the products of that layer then move 16-byte chunks (LDS-DMA tiles, split-K) instead of
    single floats (x W^T 62 -> 56 us, dW = dZ^T x 74 -> 54 us at Photo size, scripts/gemm_k745_time.py).  None when x needs nothing."""
    f = x.shape[1]
    if f % 4 == 0 or f < 32 or x.requires_grad or x.dim() != 2 or os.environ.get("GGAD_PAD_FEATURES", "1") == "0":
        return None
    key = (x.data_ptr(), x._version, tuple(x.shape))
    ent = adj.__dict__.get("_xpad")
    if ent is None or ent["key"] != key:
        xp = torch.zeros(x.shape[0], (f + 3) // 4 * 4, dtype=torch.float32, device=x.device)
        xp[:, :f] = x
        ent = adj.__dict__["_xpad"] = {"key": key, "x": x, "xp": xp}      # (keeps x alive: see cached_aggregate)
    return ent["xp"]


_PRELU_ONE = os.environ.get("GGAD_PRELU_ONE", "1") != "0"            # 0: k_prelu_bwd_v4 + the trailing k_prelu_bwd_final launch (A/B)
_TICKET_WORDS = {}


def _ticket_word(device) -> torch.Tensor:
    """One zeroed int32 per (device, stream): the ticket of the single-launch reductions (launches of one stream are ordered, so they
    can share it; the kernels leave it zero).  Allocated on first use -- in an eager epoch, before any capture."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
    t = _TICKET_WORDS.get(key)
    if t is None:
        t = _TICKET_WORDS[key] = torch.zeros(max(1, int(_lib.load().ggad_prelu_bwd_one_tickets())), dtype=torch.int32, device=device)
    return t


class GcnLayerFn(torch.autograd.Function):
    """out = PReLU(A_hat (X W^T) + b)   (reference GCN.forward, `model.py:26-35`)."""

    @staticmethod
    def forward(ctx, x, weight, bias, prelu_a, adj: FullGraphAdj):
        ctx.reordered = _reorder_layer(x, weight)
        if ctx.reordered:
            ax = cached_aggregate(adj, x)
            axp = adj._ax.get("axp")
            a_op = ax if axp is None else axp                                # (zero columns behind A_hat X and W: the same sums)
            if axp is None:
                w_op = weight.contiguous()
            else:                                                            # the weight behind zero columns: ONE copy into a kept buffer
                wkey = (weight.data_ptr(), tuple(weight.shape), axp.shape[1])
                ent = adj.__dict__.get("_wpad")
                if ent is None or ent[0] != wkey:
                    ent = adj.__dict__["_wpad"] = (wkey, torch.zeros(weight.shape[0], axp.shape[1], dtype=torch.float32, device=weight.device))
                w_op = ent[1]
                w_op[:, :weight.shape[1]].copy_(weight.detach())
            z = torch.empty(a_op.shape[0], w_op.shape[0], dtype=torch.float32, device=a_op.device)
            out = torch.empty_like(z)
            # product, bias and activation in one launch when the slab kernel takes the shape (ggad_linear_prelu_f32), else two
            rc = _lib.GGAD_E_UNSUPPORTED
            if os.environ.get("GGAD_LINEAR_PRELU_FUSED", "1") != "0":
                rc = int(_lib.load().ggad_linear_prelu_f32(ptr(a_op), a_op.stride(0), ptr(w_op), w_op.stride(0), ptr(bias) if bias is not None else 0,
                                                           ptr(prelu_a), a_op.shape[0], w_op.shape[0], a_op.shape[1], ptr(z), z.stride(0),
                                                           ptr(out), out.stride(0), _lib.current_stream()))
            if rc == _lib.GGAD_E_UNSUPPORTED:                                # nothing was launched: the documented fallback
                z = gemm(a_op, w_op, False, True, bias=bias, out=z)          # (A_hat X) W^T + b
                call("ggad_prelu_fwd_f32", ptr(z), ptr(prelu_a), z.numel(), ptr(out))
            else:
                _lib.check(rc, "ggad_linear_prelu_f32")
            ctx.save_for_backward(ax, weight, z, prelu_a)
        else:
            xq = padded_constant(adj, x)
            if xq is not None:                                               # (zero columns behind x and W: the same sums)
                wq = torch.nn.functional.pad(weight, (0, xq.shape[1] - x.shape[1]))
                t = gemm(xq, wq, False, True, out=padded_rows(x.shape[0], weight.shape[0], x.device, (adj.A, None)))
            else:
                t = gemm(x, weight, False, True, out=padded_rows(x.shape[0], weight.shape[0], x.device, (adj.A, None)))      # seq_fts = fc(seq)   model.py:27
            out, z = spmm(adj.A, t, bias=bias, prelu_a=prelu_a, want_pre=True)   # bmm(adj, .) + bias, act  model.py:31-35
            ctx.save_for_backward(x, weight, z, prelu_a)
        ctx.adj = adj
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, z, prelu_a = ctx.saved_tensors
        adj = ctx.adj
        g = g.contiguous()
        M, W = z.shape
        lib = _lib.load()
        S = int(lib.ggad_prelu_bwd_splits(M))
        ws = torch.empty(int(lib.ggad_prelu_bwd_one_workspace_elems(M, W)) if _PRELU_ONE else 2 * S * W, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z) if ctx.reordered else padded_rows(M, W, z.device, (adj.At, None))      # read by A_hat^T dZ below
        db = torch.empty(W, dtype=torch.float32, device=z.device)
        da = torch.empty(1, dtype=torch.float32, device=z.device)
        if _PRELU_ONE:                                                       # round 6: the column reduction inside the same launch
            call("ggad_prelu_bwd_one_f32", ptr(g), ptr(z), ptr(prelu_a), M, W, ptr_rows(dz), dz.stride(0), ptr(db), ptr(da), ptr(ws),
                 ptr(_ticket_word(z.device)))
        else:
            call("ggad_prelu_bwd_ld_f32", ptr(g), ptr(z), ptr(prelu_a), M, W, ptr_rows(dz), dz.stride(0), ptr(db), ptr(da), ptr(ws))
        if ctx.reordered:                                                    # x holds A_hat X here
            dw = gemm(dz, x, True, False)                                    # (H x N)(N x F), no transposed product needed
            return None, dw, (db if ctx.has_bias else None), da.view_as(prelu_a), None
        dt = spmm(adj.At, dz)                                                # A_hat^T dZ
        xq = padded_constant(adj, x)
        dw = gemm(dt, x, True, False) if xq is None else gemm(dt, xq, True, False)[:, :x.shape[1]].contiguous()      # (H x N)(N x F)
        dx = gemm(dt, weight, False, False) if ctx.needs_input_grad[0] else None
        return dx, dw, (db if ctx.has_bias else None), da.view_as(prelu_a), None


class LinearFn(torch.autograd.Function):
    """y = [relu](x W^T): nn.Linear(bias=False) of the scorer MLP / fc4 (`model.py:156,176-180`)."""

    @staticmethod
    def forward(ctx, x, weight, relu: bool):
        y = gemm(x, weight, False, True, relu=relu)
        ctx.save_for_backward(x, weight, y)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        g = g.contiguous()
        if ctx.relu:
            dz = torch.empty_like(g)
            call("ggad_relu_bwd_f32", ptr(g), ptr(y), g.numel(), ptr(dz))
        else:
            dz = g
        dx = gemm(dz, weight, False, False) if ctx.needs_input_grad[0] else None
        dw = gemm(dz, x, True, False)
        return dx, dw, None


def mlp_score_supported(w1: torch.Tensor, w2: torch.Tensor, w3: torch.Tensor) -> bool:
    """The fused scorer-MLP kernels (csrc/mlp.hip) take these weights?  GGAD_MLP_FUSED=0 turns them off (A/B: the three GEMMs)."""
    if os.environ.get("GGAD_MLP_FUSED", "1") == "0" or w3.shape[0] != 1 or w1.shape[0] != w2.shape[1] or w2.shape[0] != w3.shape[1]:
        return False
    return bool(_lib.load().ggad_mlp_score_supported(int(w1.shape[1]), int(w1.shape[0]), int(w2.shape[0])))


def mlp_score_fwd(x: torch.Tensor, w1, w2, w3):
    """(f1, f2, f3) = relu(x W1^T), relu(f1 W2^T), f2 w3^T in ONE launch (model.py:176-180)."""
    x, w1, w2, w3 = x.contiguous(), w1.contiguous(), w2.contiguous(), w3.contiguous()
    r, h = x.shape
    h1, h2 = w1.shape[0], w2.shape[0]
    if w1.shape[1] != h or w2.shape[1] != h1 or tuple(w3.shape) != (1, h2):
        raise ValueError(f"mlp_score shape mismatch: x {tuple(x.shape)}, w1 {tuple(w1.shape)}, w2 {tuple(w2.shape)}, w3 {tuple(w3.shape)}")
    f1 = torch.empty(r, h1, dtype=torch.float32, device=x.device)
    f2 = torch.empty(r, h2, dtype=torch.float32, device=x.device)
    f3 = torch.empty(r, 1, dtype=torch.float32, device=x.device)
    call("ggad_mlp_score_fwd_f32", ptr(x), h, r, h, h1, h2, ptr(w1), ptr(w2), ptr(w3), ptr(f1), ptr(f2), ptr(f3))
    return f1, f2, f3


def mlp_score_dgrad(g3: torch.Tensor, f1, f2, w1, w2, w3, dx_add: Optional[torch.Tensor] = None):
    """(dz2, dz1, dx): the data gradients of the scorer MLP in ONE launch; dx = dz1 W1 (+ dx_add)."""
    g3 = g3.contiguous().reshape(-1)
    r, h1 = f1.shape
    h2, h = f2.shape[1], w1.shape[1]
    if (f2.shape[0] != r or g3.numel() != r or tuple(w1.shape) != (h1, h) or tuple(w2.shape) != (h2, h1) or tuple(w3.shape) != (1, h2)
            or (dx_add is not None and tuple(dx_add.shape) != (r, h))):
        raise ValueError(f"mlp_score_dgrad shape mismatch: g3 {g3.numel()}, f1 {tuple(f1.shape)}, f2 {tuple(f2.shape)}, "
                         f"w1 {tuple(w1.shape)}, w2 {tuple(w2.shape)}, w3 {tuple(w3.shape)}")
    f1, f2 = f1.contiguous(), f2.contiguous()
    dz2, dz1 = torch.empty_like(f2), torch.empty_like(f1)
    dx = torch.empty(r, h, dtype=torch.float32, device=f1.device)
    if dx_add is not None:
        dx_add = dx_add.contiguous()
    call("ggad_mlp_score_dgrad_f32", ptr(g3), r, h, h1, h2, ptr(f1), ptr(f2), ptr(w1.contiguous()), ptr(w2.contiguous()), ptr(w3.contiguous()),
         ptr(dz2), ptr(dz1), ptr(dx), h, ptr(dx_add) if dx_add is not None else 0, h)
    return dz2, dz1, dx


def mlp_score_wgrad(x: torch.Tensor, dz1, f1, dz2, f2, g3):
    """(dW1, dW2, dW3) = (dz1^T x, dz2^T f1, g3^T f2): the three weight gradients of the scorer in one launch + one reduction
    (`ggad_mlp_score_wgrad_f32`; they were three split-K GEMMs + three reductions, 47 us of a Reddit epoch)."""
    r, h = x.shape
    h1, h2 = f1.shape[1], f2.shape[1]
    g3 = g3.contiguous().reshape(-1)
    dz1, f1, dz2, f2 = dz1.contiguous(), f1.contiguous(), dz2.contiguous(), f2.contiguous()
    if tuple(dz1.shape) != (r, h1) or tuple(dz2.shape) != (r, h2) or f1.shape[0] != r or f2.shape[0] != r or g3.numel() != r:
        raise ValueError("mlp_score_wgrad shape mismatch")
    dev = x.device
    dw1 = torch.empty(h1, h, dtype=torch.float32, device=dev)
    dw2 = torch.empty(h2, h1, dtype=torch.float32, device=dev)
    dw3 = torch.empty(1, h2, dtype=torch.float32, device=dev)
    ws = torch.empty(int(_lib.load().ggad_mlp_score_wgrad_workspace_elems(r, h, h1, h2)), dtype=torch.float32, device=dev)
    call("ggad_mlp_score_wgrad_f32", ptr_rows(x), x.stride(0), ptr(dz1), ptr(f1), ptr(dz2), ptr(f2), ptr(g3), r, h, h1, h2,
         ptr(dw1), ptr(dw2), ptr(dw3), ptr(ws))
    return dw1, dw2, dw3


class MlpScoreFn(torch.autograd.Function):
    """f_3 = fc3(relu(fc2(relu(fc1(x))))) (`model.py:176-180`) on the fused kernels: one launch forward, one for the data gradients,
    three split-K GEMMs for the weight gradients."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3):
        f1, f2, f3 = mlp_score_fwd(x, w1, w2, w3)
        ctx.save_for_backward(x, f1, f2, w1, w2, w3)
        return f3

    @staticmethod
    def backward(ctx, g):
        x, f1, f2, w1, w2, w3 = ctx.saved_tensors
        g = g.contiguous()
        dz2, dz1, dx = mlp_score_dgrad(g, f1, f2, w1, w2, w3)
        if os.environ.get("GGAD_MLP_WGRAD_FUSED", "1") != "0" and x.stride(1) == 1:
            dw1, dw2, dw3 = mlp_score_wgrad(x, dz1, f1, dz2, f2, g)
        else:
            dw1, dw2, dw3 = gemm(dz1, x, True, False), gemm(dz2, f1, True, False), gemm(g.reshape(-1, 1), f2, True, False)
        return (dx if ctx.needs_input_grad[0] else None, dw1, dw2, dw3)


class SpmmRowsFn(torch.autograd.Function):
    """A_hat[rows, :] @ emb  (`model.py:151-155`)."""

    @staticmethod
    def forward(ctx, emb, adj: FullGraphAdj, rows_plan, sub_t: Csr):
        ctx.sub_t = sub_t
        return spmm(adj.A, emb, plan=rows_plan)

    @staticmethod
    def backward(ctx, g):
        return spmm(ctx.sub_t, g.contiguous()), None, None, None


_HEAD_LEAN = os.environ.get("GGAD_HEAD_LEAN", "1") != "0"            # 0: the round-5 glue (separate gathers, emb_out as a second N x H tensor)


class GgadHeadFn(torch.autograd.Function):
    """The training forward from `emb` on (`model.py:140-182`) as ONE autograd node:

        emb_abnormal = emb[abn] + noise                                   :141-145
        emb_con      = relu(fc4(A_hat[abn, :] emb))                       :151-156
        emb_combine  = cat(emb[normal], emb_con)                          :159
        f_3          = fc3(relu(fc2(relu(fc1(emb_combine)))))             :176-180
        emb_out      = emb with rows abn replaced by emb_con              :182

    Same products as the op-by-op path (`Model._head_unfused`: `LinearFn`, `SpmmRowsFn`, torch indexing); what is fused is the glue
    -- gather / add / cat / index_copy forward, and in the backward the gradient accumulation over the five consumers of `emb` and
    the three of `emb_con`, which autograd does with one full-tensor kernel per term (~35 launches of 5 us per Reddit epoch).
    The sums of gradient terms are formed in one fixed order (`k_head_con_grad`, `k_head_emb_grad`); autograd's order is another,
    so the two paths agree to fp32 round-off, not bit for bit (tests/test_fullgraph_gpu.py)."""

    @staticmethod
    def forward(ctx, emb, noise, w4, w1, w2, w3, adj: FullGraphAdj, hs):
        emb_in = emb
        emb = emb.contiguous()
        n, h = emb.shape
        dev = emb.device
        na, nn_ = hs["n_abn"], hs["n_nrm"]
        noise = noise.reshape(na, h).contiguous() if noise is not None else None
        emb_abn = torch.empty(na, h, dtype=torch.float32, device=dev)
        comb = torch.empty(nn_ + na, h, dtype=torch.float32, device=dev)
        lean = _HEAD_LEAN and emb_in.is_contiguous()
        if lean:
            # round 6: both row gathers from emb in one launch; emb_con is written by its product straight into the tail of emb_combine
            call("ggad_head_rows_f32", ptr(emb), ptr(hs["abn"]), ptr(noise), na, ptr(hs["nrm"]), nn_, h, ptr(emb_abn), ptr(comb))
            con_pre = spmm(adj.A, emb, plan=hs["rows_plan"])
            emb_con = gemm(con_pre, w4, False, True, relu=True, out=comb[nn_:])
        else:
            call("ggad_head_gather_f32", ptr(emb), ptr(hs["abn"]), ptr(noise), na, h, ptr(emb_abn))
            con_pre = spmm(adj.A, emb, plan=hs["rows_plan"])
            emb_con = gemm(con_pre, w4, False, True, relu=True)
            call("ggad_head_combine_f32", ptr(emb), ptr(hs["nrm"]), nn_, ptr(emb_con), na, h, ptr(comb))
        if mlp_score_supported(w1, w2, w3):
            f1, f2, f3 = mlp_score_fwd(comb, w1, w2, w3)                   # one launch (csrc/mlp.hip)
        else:
            f1 = gemm(comb, w1, False, True, relu=True)
            f2 = gemm(f1, w2, False, True, relu=True)
            f3 = gemm(f2, w3, False, True)
        if lean:
            # emb[:, abn, :] = emb_con in place, as the reference writes it (model.py:182): A rows instead of a second N x H tensor.  Every
            # read of the old rows (emb[abn] + noise, A_hat[abn, :] emb, emb[normal]) is behind us; GcnLayerFn keeps z, not its output
            call("ggad_head_emb_put_f32", ptr(emb_con), ptr(hs["abn"]), na, h, ptr(emb))
            ctx.mark_dirty(emb_in)
            emb_out = emb_in
        else:
            emb_out = torch.empty_like(emb)
            call("ggad_head_emb_out_f32", ptr(emb), ptr(hs["abn_pos"]), ptr(emb_con), n, h, ptr(emb_out))
        ctx.save_for_backward(con_pre, emb_con, comb, f1, f2, w4, w1, w2, w3)
        ctx.hs, ctx.shape = hs, (n, h)
        ctx.set_materialize_grads(False)
        return emb_out, comb, f3, emb_con, emb_abn

    @staticmethod
    def backward(ctx, g_out, g_comb, g_f3, g_con, g_abn):
        con_pre, emb_con, comb, f1, f2, w4, w1, w2, w3 = ctx.saved_tensors
        hs = ctx.hs
        n, h = ctx.shape
        na, nn_ = hs["n_abn"], hs["n_nrm"]
        dev = emb_con.device
        c = lambda t: t.contiguous() if t is not None else None          # noqa: E731
        g_out, g_comb, g_f3, g_con, g_abn = c(g_out), c(g_comb), c(g_f3), c(g_con), c(g_abn)
        dw1 = dw2 = dw3 = None
        d_comb = g_comb
        if g_f3 is not None and mlp_score_supported(w1, w2, w3):           # scorer MLP: data gradients in one launch (+ g_comb)
            dz2, dz1, d_comb = mlp_score_dgrad(g_f3, f1, f2, w1, w2, w3, g_comb)
            if os.environ.get("GGAD_MLP_WGRAD_FUSED", "1") != "0" and comb.stride(1) == 1:
                dw1, dw2, dw3 = mlp_score_wgrad(comb, dz1, f1, dz2, f2, g_f3)       # the three weight gradients: one launch + one reduction
            else:
                dw3 = gemm(g_f3.reshape(-1, 1), f2, True, False)
                dw2 = gemm(dz2, f1, True, False)
                dw1 = gemm(dz1, comb, True, False)
        elif g_f3 is not None:                                             # ... or as LinearFn.backward three times
            dw3 = gemm(g_f3, f2, True, False)
            df2 = gemm(g_f3, w3, False, False)
            dz2 = torch.empty_like(df2)
            call("ggad_relu_bwd_f32", ptr(df2), ptr(f2), df2.numel(), ptr(dz2))
            dw2 = gemm(dz2, f1, True, False)
            df1 = gemm(dz2, w2, False, False)
            dz1 = torch.empty_like(df1)
            call("ggad_relu_bwd_f32", ptr(df1), ptr(f1), df1.numel(), ptr(dz1))
            dw1 = gemm(dz1, comb, True, False)
            d_comb = gemm(dz1, w1, False, False)
            if g_comb is not None:
                d_comb = d_comb + g_comb
        dz4 = torch.empty(na, h, dtype=torch.float32, device=dev)
        tail = d_comb[nn_:] if d_comb is not None else None              # rows of emb_combine that are emb_con (contiguous slice)
        call("ggad_head_con_grad_f32", ptr(g_con), ptr(g_out), ptr(hs["abn"]), ptr(tail), ptr(emb_con), na, h, ptr(dz4))
        dw4 = gemm(dz4, con_pre, True, False)
        g_emb = None
        if ctx.needs_input_grad[0]:
            d_pre = gemm(dz4, w4, False, False)
            sp = spmm(hs["sub_t"], d_pre)
            g_emb = torch.empty(n, h, dtype=torch.float32, device=dev)
            call("ggad_head_emb_grad_f32", ptr(g_out), ptr(hs["abn_pos"]), ptr(hs["nrm_pos"]), ptr(d_comb), ptr(g_abn), ptr(sp), n, h,
                 ptr(g_emb))
        return g_emb, None, dw4, dw1, dw2, dw3, None, None


# GGAD_LOSS_FUSED=0: the round-5 launch sequence of the loss block (A/B measurements; the tests run both)
_LOSS_FUSED = os.environ.get("GGAD_LOSS_FUSED", "1") != "0"


class GgadLossFn(torch.autograd.Function):
    """(total, margin, bce, rec) of `run.py:165-210`; only `total` is differentiable."""

    @staticmethod
    def forward(ctx, emb, logits, emb_con, emb_abn, adj: FullGraphAdj, ls, margin: float):
        ctx.set_materialize_grads(False)       # only `total` is differentiated: no zero tensors for the three other outputs
        emb = emb.contiguous()
        n, h = emb.shape
        dev = emb.device
        inv = torch.empty(n, dtype=torch.float32, device=dev)
        en = torch.empty_like(emb)
        call("ggad_rownorm_f32", ptr(emb), n, h, ptr(inv), ptr(en))                       # run.py:177-180
        J, L = ls["J"], int(ls["J"].numel())
        s_j = spmm(adj.Rt, en, plan=ls["Rt_plan"])                                        # (R^T e_hat)[J]
        aff = torch.empty(L, dtype=torch.float32, device=dev)
        losses = torch.empty(4, dtype=torch.float32, device=dev)
        d_logits = torch.empty(L, dtype=torch.float32, device=dev)
        g_aff = torch.empty(L, dtype=torch.float32, device=dev)
        emb_con, emb_abn, logits = emb_con.contiguous(), emb_abn.contiguous(), logits.contiguous()
        fused = _LOSS_FUSED and ls.get("seg_unique", False) and h <= 1024
        ctx.fused = fused
        if fused:
            # round 6: row dots, recon partials and -- in the last workgroup to finish -- BCE / margin / recon / coefficients: ONE launch
            ws = ls.get("loss_ws_fused")
            if ws is None:
                ws = ls["loss_ws_fused"] = (torch.empty(int(_lib.load().ggad_full_loss_fused_workspace_elems(ls["n_out"], h)),
                                                        dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
            kcol = torch.empty(h, dtype=torch.float32, device=dev)
            call("ggad_full_loss_fused_f32", ptr(en), ptr(J), ptr(s_j), ptr(ls["r_inv_J"]), ls["n_normal"], ls["n_out"], h, ptr(logits),
                 ptr(emb_con), ptr(emb_abn), float(margin), ptr(aff), ptr(kcol), ptr(losses), ptr(d_logits), ptr(g_aff), ptr(ws[0]), ptr(ws[1]))
            ctx.save_for_backward(en, inv, s_j, g_aff, d_logits, kcol, emb_con, emb_abn)
        else:
            call("ggad_rowdot_f32", ptr(en), ptr(J), ptr(s_j), L, h, ptr(ls["r_inv_J"]), ptr(aff))   # run.py:182-188
            dD = torch.empty_like(emb_con)
            ws = ls.get("loss_ws")
            if ws is None:
                ws = ls["loss_ws"] = torch.empty(int(_lib.load().ggad_full_loss_workspace_elems(ls["n_out"], h)), dtype=torch.float32,
                                                 device=dev)
            call("ggad_full_loss_f32", ptr(logits), ptr(aff), ls["n_normal"], ls["n_out"], ptr(emb_con), ptr(emb_abn), h,
                 float(margin), ptr(losses), ptr(d_logits), ptr(g_aff), ptr(dD), ptr(ws))
            ctx.save_for_backward(en, inv, s_j, g_aff, d_logits, dD)
        ctx.adj, ctx.ls = adj, ls
        ctx.affinity = aff
        return losses[0], losses[1], losses[2], losses[3]

    @staticmethod
    def backward(ctx, g_total, g_margin, g_bce, g_rec):
        adj, ls = ctx.adj, ctx.ls
        J, L, nn_ = ls["J"], int(ls["J"].numel()), ls["n_normal"]
        if g_total is None:
            return None, None, None, None, None, None, None
        if g_total.dtype != torch.float32:
            g_total = g_total.to(torch.float32)
        g_total = g_total.reshape(1)
        if ctx.fused:
            en, inv, s_j, g_aff, d_logits, kcol, emb_con, emb_abn = ctx.saved_tensors
            n, h = en.shape
            c, dl = torch.empty_like(g_aff), torch.empty_like(d_logits)                   # c = d total / d (e_hat_j . S_j)
            d_con, d_abn = torch.empty_like(emb_con), torch.empty_like(emb_con)
            xc = torch.empty(L, h, dtype=torch.float32, device=en.device)                 # c_j e_hat_j
            call("ggad_full_loss_bwd_fused_f32", ptr(g_total), ptr(g_aff), ptr(ls["r_inv_J"]), ptr(d_logits), ptr(en), ptr(J), L, h,
                 ptr(emb_con), ptr(emb_abn), ptr(kcol), ls["n_out"], ptr(c), ptr(dl), ptr(xc), ptr(d_con), ptr(d_abn))
            den = spmm(ls["RJ"], xc)                                                      # sum_j R_ij c_j e_hat_j
            d_emb = torch.empty_like(en)
            call("ggad_rownorm_bwd_add_f32", ptr(en), ptr(inv), ptr(den), ptr(ls["pos_n"]), ptr(ls["pos_a"]), ptr(c), ptr(s_j), n, h,
                 ptr(d_emb))                                                              # + c_j S_j on the rows J, then the normalisation's VJP
            return d_emb, dl, d_con, d_abn, None, None, None
        en, inv, s_j, g_aff, d_logits, dD = ctx.saved_tensors
        n, h = en.shape
        c, dl = torch.empty_like(g_aff), torch.empty_like(d_logits)                       # c = d total / d (e_hat_j . S_j)
        d_con, d_abn = torch.empty_like(dD), torch.empty_like(dD)
        g_total = g_total.contiguous()
        call("ggad_full_loss_bwd_scale_f32", ptr(g_total), ptr(g_aff), ptr(ls["r_inv_J"]), ptr(d_logits), ptr(dD), L, dD.numel(),
             ptr(c), ptr(dl), ptr(d_con), ptr(d_abn))
        xc = torch.empty(L, h, dtype=torch.float32, device=en.device)
        call("ggad_rows_scale_f32", ptr(en), ptr(J), ptr(c), L, h, 0, ptr(xc))            # c_j e_hat_j
        den = spmm(ls["RJ"], xc)                                                          # sum_j R_ij c_j e_hat_j
        # + c_j S_j on rows J; normal and abnormal segments are each duplicate-free
        if ls.get("distinct"):                                                            # no node in both lists (run.py's draws): one launch
            call("ggad_rows_scale_f32", ptr(s_j), ptr(J), ptr(c), L, h, 1, ptr(den))
        else:
            call("ggad_rows_scale_f32", ptr(s_j), ptr(J), ptr(c), nn_, h, 1, ptr(den))
            call("ggad_rows_scale_f32", s_j.data_ptr() + 4 * nn_ * h, J.data_ptr() + 4 * nn_, c.data_ptr() + 4 * nn_, L - nn_, h, 1,
                 ptr(den))
        d_emb = torch.empty_like(en)
        call("ggad_rownorm_bwd_f32", ptr(en), ptr(inv), ptr(den), n, h, ptr(d_emb))
        return d_emb, dl, d_con, d_abn, None, None, None


def ggad_loss(emb, logits, emb_con, emb_abnormal, adj: FullGraphAdj, ls, margin: float):
    """`GgadLossFn` on what `Model.forward` returns ((1, N, H), (1, L, 1), (A, H), (1, A, H)): the 2-D views are reshapes, whose
    backward is a view again -- `emb[0]` / `logits[0, :, 0]` cost a zero-fill and a copy of the whole tensor each in autograd."""
    h = emb.shape[-1]
    return GgadLossFn.apply(emb.reshape(-1, h), logits.reshape(-1), emb_con, emb_abnormal.reshape(-1, h), adj, ls, margin)


_ADAM_TICKETS = os.environ.get("GGAD_ADAM_TICKETS", "1") != "0"      # 0: the trailing k_bump_multi launch of round 5 (A/B)


class FlatAdam:
    """torch.optim.Adam semantics (lr, weight_decay, betas .9/.999, eps 1e-8) executed by the HIP Adam kernel on
    the parameters that received a gradient (params with ``grad is None`` are skipped, as torch does)."""

    def __init__(self, params: Sequence[torch.nn.Parameter], lr: float, weight_decay: float = 0.0):
        self.params = [p for p in params]
        self.lr, self.wd = float(lr), float(weight_decay)
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self):
        """One multi-tensor launch per <= 16 parameters with a gradient (the step counters advance inside it)."""
        import ctypes
        todo = []
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = (torch.zeros_like(p.data), torch.zeros_like(p.data),
                                      torch.zeros(1, dtype=torch.int32, device=p.device))
            todo.append((p, st, p.grad.contiguous()))
        cap = int(_lib.load().ggad_adam_multi_max())
        if todo and getattr(self, "_tickets", None) is None:
            # one zeroed ticket word per tensor slot and launch group: the last workgroup of a tensor advances its step counter (round 6:
            # no trailing launch); allocated once, outside any capture (the first steps of every caller are eager)
            self._tickets = torch.zeros((len(self.params) + cap - 1) // cap * cap, dtype=torch.int32, device=todo[0][0].device)
        for i0 in range(0, len(todo), cap):
            part = todo[i0:i0 + cap]
            n = len(part)
            arr = ctypes.c_void_p * n
            call("ggad_adam_multi_f32", n, arr(*[ptr(p.data) for p, _, _ in part]), arr(*[ptr(st[0]) for _, st, _ in part]),
                 arr(*[ptr(st[1]) for _, st, _ in part]), arr(*[ptr(g) for _, _, g in part]),
                 (ctypes.c_int64 * n)(*[p.numel() for p, _, _ in part]), arr(*[ptr(st[2]) for _, st, _ in part]), self.lr, self.wd,
                 self._tickets.data_ptr() + 4 * i0 if _ADAM_TICKETS else 0)
