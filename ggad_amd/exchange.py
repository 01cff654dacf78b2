"""One-shot gradient exchange of the data-parallel step (the reference is single-device, `src/main.py:9`).

`OneShotExchange` wraps the `ggad_xchg_*` C-ABI: a buffer in fine-grained device memory per rank, HIP IPC handles
exchanged once through `torch.distributed`, then every optimiser step is ONE kernel per rank that writes its 5,248
gradients into all peers' buffers over xGMI, waits for theirs, sums in rank order and applies Adam (`k_xchg_adam`).
`connect()` ends with a self-test -- one exchange of a known pattern with a bounded wait -- and reports whether every
rank saw every peer; callers fall back to the RCCL all-reduce when it does not (e.g. no peer access between two devices).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib


class OneShotExchange:
    def __init__(self, rank: int, world: int, n_floats: int, device):
        self.lib = _lib.load()
        self.rank, self.world, self.n = int(rank), int(world), int(n_floats)
        self.dev = torch.device(device)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.ggad_xchg_create(self.rank, self.world, self.n, ctypes.byref(self._h)), "ggad_xchg_create")
        self.ok = self.world == 1
        # why `connect` declined (None = it did not): "export" (this rank's IPC handle), "peer_export" (another rank's), "map" (a peer's buffer
        # could not be opened on some rank: no peer access / IPC refused), "selftest" (the known-pattern exchange timed out or disagreed)
        self.fail_stage = None

    @property
    def handle(self):
        return self._h

    def connect(self, dist, group=None) -> bool:
        """Exchange the IPC handles over `dist` (any backend), map the peers' buffers, run the self-test.  Returns True when the
        one-shot path is usable on every rank."""
        if self.world == 1:
            self.ok = True
            return True
        nb = int(self.lib.ggad_xchg_handle_bytes())
        mine = (ctypes.c_ubyte * nb)()
        usable = 1
        try:
            with torch.cuda.device(self.dev):
                _lib.check(self.lib.ggad_xchg_handle(self._h, mine), "ggad_xchg_handle")
        except _lib.GgadKernelError:
            usable = 0
        backend = dist.get_backend(group)
        dev = self.dev if backend == "nccl" else torch.device("cpu")
        t = torch.tensor(list(bytes(mine)) + [usable], dtype=torch.uint8, device=dev)
        allh = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(allh, t, group=group)
        allh = [a.cpu().numpy() for a in allh]
        if not all(int(a[-1]) for a in allh):
            self.fail_stage = "export" if not usable else "peer_export"
            return self._agree(dist, group, False)
        blob = np.concatenate([a[:nb] for a in allh]).astype(np.uint8)
        try:
            with torch.cuda.device(self.dev):
                _lib.check(self.lib.ggad_xchg_connect(self._h, blob.ctypes.data), "ggad_xchg_connect")
            good = True
        except _lib.GgadKernelError:
            good = False
        if not self._agree(dist, group, good):
            self.fail_stage = "map"
            return False
        if not self._agree(dist, group, self._selftest()):
            self.fail_stage = "selftest"
            return False
        return True

    @staticmethod
    def decline(dist, device, group=None) -> bool:
        """For a rank that could not create its buffer: take part in the collectives of `connect` so that the other ranks do not
        hang, voting "not usable".  Returns False."""
        world = dist.get_world_size(group)
        nb = int(_lib.load().ggad_xchg_handle_bytes())
        backend = dist.get_backend(group)
        dev = torch.device(device) if backend == "nccl" else torch.device("cpu")
        t = torch.zeros(nb + 1, dtype=torch.uint8, device=dev)
        dist.all_gather([torch.zeros_like(t) for _ in range(world)], t, group=group)
        f = torch.zeros(1, dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        return False

    def _agree(self, dist, group, good: bool) -> bool:
        backend = dist.get_backend(group)
        dev = self.dev if backend == "nccl" else torch.device("cpu")
        f = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        self.ok = bool(int(f.item()))
        return self.ok

    def _selftest(self) -> bool:
        """One exchange + Adam on a dummy block: every rank publishes rank + 1 everywhere; with lr = 1, zero state and no weight
        decay the first Adam step moves every parameter by -sign(sum) = -1; a wait that times out sets the error word."""
        D, F = 8, 4                                              # 8 + 32 + 64 = 104 parameters: one workgroup
        n = int(self.lib.ggad_mb_param_count(D, F))
        nblock = int(self.lib.ggad_mb_param_block_elems(D, F))
        if n > self.n:
            return False
        dev = self.dev
        with torch.cuda.device(dev):
            params = torch.zeros(nblock, dtype=torch.float32, device=dev)
            m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            g = torch.full((n,), float(self.rank + 1), device=dev)
            sc = torch.zeros(1, dtype=torch.int32, device=dev)
            sc += 1
            _lib.check(self.lib.ggad_xchg_adam(params.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr(), D, F, 1.0, 0.0,
                                               1.0 / self.world, sc.data_ptr(), self._h, _lib.current_stream()), "ggad_xchg_adam")
            torch.cuda.synchronize()
            err = ctypes.c_int32(0)
            _lib.check(self.lib.ggad_xchg_error(self._h, ctypes.byref(err)), "ggad_xchg_error")
            want = (self.world + 1) / 2.0                        # mean of 1 .. W, also what exp_avg holds / 0.1
            got = (m[:n] / 0.1).cpu().numpy()
        return err.value == 0 and bool(np.allclose(got, want, rtol=1e-5))

    def error(self) -> int:
        err = ctypes.c_int32(0)
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.ggad_xchg_error(self._h, ctypes.byref(err)), "ggad_xchg_error")
        return int(err.value)

    def close(self) -> None:
        if self._h:
            self.lib.ggad_xchg_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
