"""Per-workgroup clocks of k_spmm_rowline (start, hub rows done, end) on the N x N x 300 product of Reddit / Photo.
Needs a library built with GGAD_EXTRA_HIPFLAGS=-DGGAD_RL_PROF.  Usage (GPU box): python scripts/rowline_clocks.py [reddit]"""
import ctypes
import random
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import _lib, fullgraph as FG  # noqa: E402
from ggad_amd.fullgraph_bench import make_dataset  # noqa: E402
from ggad_amd.utils import normalize_adj  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
lib = ctypes.CDLL(_lib.load()._name)
for name in (sys.argv[1:] or ["reddit"]):
    random.seed(0); np.random.seed(0)
    ds = make_dataset(name, 0)
    n = ds["n"]
    full = FG.FullGraphAdj(normalize_adj(ds["adj"]) + sp.eye(n), ds["adj"] + sp.eye(n), dev)
    deg = np.diff(full.A.host.indptr)
    print(name, "degrees: max", deg.max(), "mean", deg.mean(), "rows > 192:", (deg > 192).sum(), "> 1024:", (deg > 1024).sum())
    xp = torch.empty(n, 320, device=dev)[:, :300]
    xp.copy_(torch.randn(n, 300, device=dev))
    for _ in range(5):
        FG.spmm(full.A, xp)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (4 * 8192))()
    assert lib.ggad_debug_rowline_prof(buf, 4 * 8192) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t[:, :3] - t0) / 100.0
    print(f"{len(t)} workgroups; start: median {np.median(us[:, 0]):.1f} max {us[:, 0].max():.1f} us; hub phase: median {np.median(us[:, 1] - us[:, 0]):.1f} max {(us[:, 1] - us[:, 0]).max():.1f} us; "
          f"item phase: median {np.median(us[:, 2] - us[:, 1]):.1f} max {(us[:, 2] - us[:, 1]).max():.1f} us; end: median {np.median(us[:, 2]):.1f} max {us[:, 2].max():.1f} us")
    for x in range(8):
        m = us[x::8]
        print(f"  XCD {x}: start max {m[:, 0].max():5.1f}  hub max {(m[:, 1] - m[:, 0]).max():5.1f}  items median {np.median(m[:, 2] - m[:, 1]):5.1f} max {(m[:, 2] - m[:, 1]).max():5.1f}  end max {m[:, 2].max():5.1f}  items/wave {t[x::8, 3].mean():.2f}")
    order = np.argsort(-us[:, 2])[:8]
    print("  last to finish (block, start, hub done, end, items):", [(int(b), *np.round(us[b], 1), int(t[b, 3])) for b in order])
