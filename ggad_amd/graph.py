"""Device-resident CSR graph container used by both GGAD paths.

The reference keeps the DGraph adjacency as a pickled ``defaultdict(set)`` (`src/utils.py:96-112`)
and the full-graph adjacency as a dense (1,N,N) tensor (`run.py:98-110`).  Here the graph lives in
HBM once, as int32 CSR with sorted, de-duplicated columns; the reference containers are accepted
and converted once (`from_adj_lists`, `from_scipy`).
"""
from __future__ import annotations

from typing import Mapping, Optional

import numpy as np
import torch


def _check_i32(x: int, what: str) -> None:
    if x >= 2 ** 31:
        raise ValueError(f"{what} = {x} does not fit the int32 indices of the HIP kernels")


class DeviceGraph:
    """CSR adjacency on one GPU (replicated per rank in the data-parallel setting)."""

    def __init__(self, rowptr: np.ndarray, col: np.ndarray, device, val: Optional[np.ndarray] = None):
        rowptr = np.ascontiguousarray(rowptr)
        col = np.ascontiguousarray(col)
        self.n = int(len(rowptr) - 1)
        self.nnz = int(rowptr[-1])
        _check_i32(self.n + 1, "number of nodes")
        _check_i32(self.nnz, "number of directed entries")
        if len(col) != self.nnz:
            raise ValueError("rowptr[-1] != len(col)")
        self.rowptr_host = rowptr.astype(np.int32, copy=False)
        self.col_host = col.astype(np.int32, copy=False)
        self.deg_host = np.diff(self.rowptr_host.astype(np.int64))
        self._closed_deg_host = None
        self.device = torch.device(device)
        self.rowptr = torch.from_numpy(self.rowptr_host).to(self.device)
        self.col = torch.from_numpy(self.col_host).to(self.device)
        self.val = None if val is None else torch.from_numpy(np.ascontiguousarray(val, dtype=np.float32)).to(self.device)

    # ---- constructors from the reference's containers
    @classmethod
    def from_adj_lists(cls, adj_lists: Mapping[int, set], n: Optional[int] = None, device="cuda") -> "DeviceGraph":
        """dict node -> set(neighbours), the reference's `adj_lists` (`src/model_handler.py:234-239`)."""
        if n is None:
            n = (max(adj_lists.keys()) + 1) if len(adj_lists) else 0
        deg = np.zeros(n, dtype=np.int64)
        for k, s in adj_lists.items():
            deg[int(k)] = len(s)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(deg, out=rowptr[1:])
        col = np.empty(int(rowptr[-1]), dtype=np.int32)
        for k, s in adj_lists.items():
            k = int(k)
            if deg[k]:
                col[rowptr[k]:rowptr[k + 1]] = np.sort(np.fromiter(s, dtype=np.int64, count=int(deg[k])))
        _check_i32(int(rowptr[-1]), "number of directed entries")
        return cls(rowptr.astype(np.int32), col, device)

    @classmethod
    def from_scipy(cls, mat, device="cuda", keep_values: bool = False) -> "DeviceGraph":
        m = mat.tocsr().copy()
        m.sum_duplicates()
        m.sort_indices()
        return cls(m.indptr.astype(np.int32), m.indices.astype(np.int32), device,
                   val=m.data if keep_values else None)

    # ---- cached binary CSR: converting the reference's 3.7 M-key dict of python sets takes minutes, the
    # cache is two int32 arrays
    def save_csr(self, path: str) -> None:
        """Write rowptr / col as an uncompressed .npz (atomic rename), to be reloaded with `load_csr`."""
        import os
        tmp = path + ".tmp.npz"
        np.savez(tmp, rowptr=self.rowptr_host, col=self.col_host, n=np.int64(self.n), nnz=np.int64(self.nnz))
        os.replace(tmp, path)

    @classmethod
    def load_csr(cls, path: str, device="cuda") -> "DeviceGraph":
        with np.load(path) as f:
            rowptr, col = f["rowptr"], f["col"]
            if int(f["n"]) != len(rowptr) - 1 or int(f["nnz"]) != len(col) or int(rowptr[-1]) != len(col):
                raise ValueError(f"{path}: inconsistent CSR cache")
        return cls(rowptr, col, device)

    @classmethod
    def from_adj_lists_cached(cls, adj_lists, n: Optional[int], device, cache_path: Optional[str]) -> "DeviceGraph":
        """`from_adj_lists` through a binary cache: reused when it describes a graph of the same size, else rebuilt."""
        import os
        if cache_path and os.path.exists(cache_path):
            try:
                g = cls.load_csr(cache_path, device)
                if n is None or g.n == n:
                    return g
            except (ValueError, OSError, KeyError):
                pass
        g = cls.from_adj_lists(adj_lists, n, device)
        if cache_path:
            try:
                g.save_csr(cache_path)
            except OSError:
                pass                      # read-only data directory: run without the cache
        return g

    def adj_lists(self):
        """Back-conversion (tests / interchange only)."""
        from .synth import csr_to_adj_lists
        return csr_to_adj_lists(self.rowptr_host, self.col_host)

    @property
    def closed_deg_host(self) -> np.ndarray:
        """|N(i) + {i}| per node = deg + 1 - [i in N(i)]  (exact; computed once, vectorised)."""
        if self._closed_deg_host is None:
            has_self = np.zeros(self.n, dtype=bool)
            step = 1 << 20                                    # nodes per slab (bounds the temporary)
            rp = self.rowptr_host
            for a in range(0, self.n, step):
                b = min(self.n, a + step)
                rows = np.repeat(np.arange(a, b, dtype=np.int32), self.deg_host[a:b])
                hit = rows == self.col_host[rp[a]:rp[b]]
                has_self[rows[hit]] = True
            self._closed_deg_host = (self.deg_host + 1 - has_self).astype(np.int64)
        return self._closed_deg_host

    @property
    def pair_bound_host(self) -> np.ndarray:
        """int64[n]: sum of deg(k) over k in N(i) + {i} -- upper bound of the 2-hop pairs row i contributes to a batch."""
        t = self.__dict__.get("_pair_bound")
        if t is None:
            rp = self.rowptr.long()
            deg = rp[1:] - rp[:-1]
            rows = torch.repeat_interleave(torch.arange(self.n, device=self.device), deg)
            acc = deg.clone()
            acc.index_add_(0, rows, deg[self.col.long()])
            t = acc.cpu().numpy()
            self.__dict__["_pair_bound"] = t
        return t

    def tile_offsets(self, shift: int = 16) -> torch.Tensor:
        """Static per-node table of CSR-row offsets at the (1 << shift)-id tile boundaries (tiled 2-hop kernels)."""
        cache = self.__dict__.setdefault("_tile_off", {})
        t = cache.get(shift)
        if t is None:
            from . import _lib
            lib = _lib.load()
            t = torch.empty(int(lib.ggad_mb_tile_offsets_elems(self.n, shift)), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.call("ggad_mb_tile_offsets", self.rowptr.data_ptr(), self.col.data_ptr(), self.n, shift, t.data_ptr())
            cache[shift] = t
        return t

    def closed_degrees(self, nodes: np.ndarray) -> np.ndarray:
        return self.closed_deg_host[np.asarray(nodes, dtype=np.int64)]
