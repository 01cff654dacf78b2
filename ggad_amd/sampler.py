"""Batch sampler: bit-exact CPython ``random`` semantics, native speed.

The reference draws its DGraph batches with ``random.shuffle`` inside the timed loop
(`src/model_handler.py:314` once per epoch over ~1.05 M ids, `:341` once per BATCH over the
55,275-id pseudo-anomaly pool = 28 ms/batch in CPython, a ~7 K nodes/s ceiling for any backend).
``PyCompatRandom`` wraps the MT19937 in libggad_hip.so (`ggad_mt_*`), reproducing the same
permutations for the same seed so that batches are identical to the reference's.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib


class PyCompatRandom:
    def __init__(self, seed: int = 0):
        self._lib = _lib.load()
        self._h = self._lib.ggad_mt_new()
        if not self._h:
            raise MemoryError("ggad_mt_new")
        self.seed(seed)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.ggad_mt_free(h)

    def seed(self, seed: int) -> None:
        seed = abs(int(seed))
        if seed >= 2 ** 64:
            raise ValueError("seeds >= 2^64 are not supported by the native sampler")
        _lib.check(self._lib.ggad_mt_seed_u64(self._h, seed), "ggad_mt_seed_u64")

    @classmethod
    def from_python_state(cls, state) -> "PyCompatRandom":
        """Continue a CPython stream: ``state = random.getstate()``."""
        version, internal, _gauss = state
        if version != 3:
            raise ValueError("unsupported random state version")
        self = cls(0)
        arr = (ctypes.c_uint32 * 624)(*internal[:624])
        _lib.check(self._lib.ggad_mt_set_state(self._h, arr, int(internal[624])), "ggad_mt_set_state")
        return self

    def to_python_state(self):
        arr = (ctypes.c_uint32 * 624)()
        idx = ctypes.c_int32(0)
        _lib.check(self._lib.ggad_mt_get_state(self._h, arr, ctypes.byref(idx)), "ggad_mt_get_state")
        return (3, tuple(int(x) for x in arr) + (int(idx.value),), None)

    def shuffle(self, a: np.ndarray) -> None:
        """random.shuffle(list) on a contiguous int64 array, in place."""
        if a.dtype != np.int64 or not a.flags.c_contiguous:
            raise ValueError("shuffle wants a contiguous int64 array")
        p = a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        _lib.check(self._lib.ggad_mt_shuffle_i64(self._h, p, a.shape[0]), "ggad_mt_shuffle_i64")

    def getrandbits32(self) -> int:
        return int(self._lib.ggad_mt_getrandbits32(self._h))
