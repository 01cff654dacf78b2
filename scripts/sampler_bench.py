"""Host sampler stage times (no GPU):  python scripts/sampler_bench.py"""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import _lib  # noqa: E402

lib = _lib.load()
g = ctypes.c_void_p(lib.ggad_mt_new())
lib.ggad_mt_seed_u64(g, 72)
n = 55275
T = np.zeros(n + 16, np.int32)
data = np.arange(n, dtype=np.int64)
reps = 200


def t(fn):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


print("targets (generator walk)  us", t(lambda: lib.ggad_mt_shuffle_targets(g, n, T.ctypes.data)))
print("apply swaps               us", t(lambda: lib.ggad_apply_swaps_i64(data.ctypes.data, n, T.ctypes.data)))
print("shuffle (both, serial)    us", t(lambda: lib.ggad_mt_shuffle_i64(g, data.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n)))
nt = 1_050_000
train = np.arange(nt, dtype=np.int64)
pool = np.arange(n, dtype=np.int64)
ie = ctypes.c_int32(150)
cnt = 600
out = np.zeros((cnt, 200), np.int64)
ol = np.zeros(cnt, np.int32)
def sched():
    return lib.ggad_sched_batches(g, train.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), nt, pool.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n, 150, 50, 150,
                       ctypes.byref(ie), cnt, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ol.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))


sched()                                  # first call: buffer allocation + first touch
for _ in range(3):
    t0 = time.perf_counter()
    sched()
    print("sched_batches             us per batch", (time.perf_counter() - t0) / cnt * 1e6, "(600 batches, 4 epoch shuffles of 1.05 M included)")
Tt = np.ones(nt + 16, np.int32)
lib.ggad_mt_shuffle_targets(g, nt, Tt.ctypes.data)
t0 = time.perf_counter(); lib.ggad_mt_shuffle_targets(g, nt, Tt.ctypes.data); t1 = time.perf_counter()
lib.ggad_apply_swaps_i64(train.ctypes.data, nt, Tt.ctypes.data); t2 = time.perf_counter()
print("epoch shuffle n=1.05M: targets ms", (t1 - t0) * 1e3, "apply ms", (t2 - t1) * 1e3)
