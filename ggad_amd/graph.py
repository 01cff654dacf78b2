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
    def save_csr(self, path: str, fingerprint: str = "") -> None:
        """Write rowptr / col as an uncompressed .npz (per-process temporary name + atomic rename, so that ranks converting
        the same pickle at the same time never replace each other's half-written file), to be reloaded with `load_csr`.
        `fingerprint` identifies the source the CSR was converted from (`source_fingerprint`)."""
        import os
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        np.savez(tmp, rowptr=self.rowptr_host, col=self.col_host, n=np.int64(self.n), nnz=np.int64(self.nnz),
                 fingerprint=np.array(fingerprint))
        os.replace(tmp, path)

    @staticmethod
    def source_fingerprint(source_path: Optional[str], adj_lists=None) -> str:
        """Identity of the adjacency the cache was built from: size + mtime of the source file when there is one, else the
        number of keys and directed entries of the dict (a regenerated graph of the same node count changes either)."""
        import os
        if source_path and os.path.exists(source_path):
            s = os.stat(source_path)
            return f"file:{s.st_size}:{int(s.st_mtime)}"
        if adj_lists is not None:
            return f"dict:{len(adj_lists)}:{sum(len(v) for v in adj_lists.values())}"
        return ""

    @classmethod
    def load_csr(cls, path: str, device="cuda", fingerprint: Optional[str] = None) -> "DeviceGraph":
        with np.load(path) as f:
            rowptr, col = f["rowptr"], f["col"]
            if int(f["n"]) != len(rowptr) - 1 or int(f["nnz"]) != len(col) or int(rowptr[-1]) != len(col):
                raise ValueError(f"{path}: inconsistent CSR cache")
            if fingerprint is not None and str(f["fingerprint"]) != fingerprint:
                raise ValueError(f"{path}: CSR cache was built from another source")
        return cls(rowptr, col, device)

    @classmethod
    def from_adj_lists_cached(cls, adj_lists, n: Optional[int], device, cache_path: Optional[str],
                              source_path: Optional[str] = None) -> "DeviceGraph":
        """`from_adj_lists` through a binary cache: reused when it was built from the same source (fingerprint: size + mtime of
        `source_path`, else key / entry counts of the dict) and describes a graph of the same size; else rebuilt.  A truncated
        or foreign cache file is ignored, never fatal."""
        import os
        import zipfile
        fp = cls.source_fingerprint(source_path, adj_lists)
        if cache_path and os.path.exists(cache_path):
            try:
                g = cls.load_csr(cache_path, device, fingerprint=fp)
                if n is None or g.n == n:
                    return g
            except (ValueError, OSError, KeyError, EOFError, zipfile.BadZipFile):
                pass
        g = cls.from_adj_lists(adj_lists, n, device)
        if cache_path:
            try:
                g.save_csr(cache_path, fp)
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
    def closed_deg_i32(self) -> np.ndarray:
        """`closed_deg_host` as the contiguous int32 table the native plan builder reads."""
        t = self.__dict__.get("_closed_deg_i32")
        if t is None:
            t = np.ascontiguousarray(self.closed_deg_host, dtype=np.int32)
            self.__dict__["_closed_deg_i32"] = t
        return t

    @property
    def mean_nbr_deg(self) -> float:
        """sum deg^2 / sum deg: the expected degree of a random neighbour (size of a 2-hop walk per entry)."""
        t = self.__dict__.get("_mean_nbr_deg")
        if t is None:
            d = self.deg_host.astype(np.float64)
            t = float((d * d).sum() / max(1.0, d.sum()))
            self.__dict__["_mean_nbr_deg"] = t
        return t

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
            t = np.ascontiguousarray(acc.cpu().numpy(), dtype=np.int64)
            self.__dict__["_pair_bound"] = t
        return t

    @property
    def node_pack_host(self) -> np.ndarray:
        """int64[n]: (closed degree << 40) | pair bound -- `ggad_mb_plan.node_pack_host`, one cache miss per batch node in the
        sizing pass of the plan builder instead of two."""
        t = self.__dict__.get("_node_pack")
        if t is None:
            pb = self.pair_bound_host
            if int(pb.max(initial=0)) >= (1 << 40):
                raise ValueError("pair bound of a node does not fit 40 bits")
            t = np.ascontiguousarray((self.closed_deg_i32.astype(np.int64) << 40) | pb)
            self.__dict__["_node_pack"] = t
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

    def tile_major(self, shift: int = 16):
        """The tile-major copy of `col` and the table of its segment starts (`ggad_mb_tile_major`, once per graph): `(tile_start,
        col_t)`.  The pair counting of the LDS 2-hop stage reads a tile's segments from one contiguous region instead of 16-byte
        pieces of the rows."""
        cache = self.__dict__.setdefault("_tile_major", {})
        t = cache.get(shift)
        if t is None:
            from . import _lib
            lib = _lib.load()
            off = self.tile_offsets(shift)
            start = torch.empty_like(off)
            col_t = torch.empty_like(self.col)
            ws = torch.empty(int(lib.ggad_mb_tile_major_workspace_elems(self.n, shift)), dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.call("ggad_mb_tile_major", self.rowptr.data_ptr(), self.col.data_ptr(), self.n, shift, off.data_ptr(),
                          start.data_ptr(), col_t.data_ptr(), ws.data_ptr())
                torch.cuda.synchronize(self.device)          # (the workspace goes back to the allocator)
            t = cache[shift] = (start, col_t)
        return t

    def closed_degrees(self, nodes: np.ndarray) -> np.ndarray:
        return self.closed_deg_host[np.asarray(nodes, dtype=np.int64)]
