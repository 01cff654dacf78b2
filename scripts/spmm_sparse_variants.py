"""N x N x 300 product on the SPARSE-neighbourhood configs (Reddit, Photo): wave-per-segment kernel against the XCD-sliced kernel and the
column-sliced six-rows-per-wave kernel (k_spmm_rowslice).  Usage (GPU box): python scripts/spmm_sparse_variants.py [reddit photo]"""
import os
import random
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import fullgraph as FG  # noqa: E402
from ggad_amd.fullgraph_bench import make_dataset, _time_call  # noqa: E402
from ggad_amd.utils import normalize_adj  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for name in (sys.argv[1:] or ["reddit", "photo"]):
    random.seed(0); np.random.seed(0)
    ds = make_dataset(name, 0)
    n = ds["n"]
    full = FG.FullGraphAdj(normalize_adj(ds["adj"]) + sp.eye(n), ds["adj"] + sp.eye(n), dev)
    x = torch.randn(n, 300, device=dev)
    nnz = int(full.A.nnz)
    floor_us = (8.0 * nnz + 4.0 * (n + 1) + 8.0 * n * 300) / 8.0e12 * 1e6
    os.environ["GGAD_SPMM_PANEL"] = "0"
    res = {}
    for tag, env, rs in (("seg", "0", "0"), ("sliced", "1", "0"), ("rowslice", "0", "1")):
        os.environ["GGAD_SPMM_SLICED"] = env
        os.environ["GGAD_SPMM_ROWSLICE"] = rs
        full.A._plans = {} if hasattr(full.A, "_plans") else None
        out = FG.spmm(full.A, x)
        res[tag] = out.clone()
        t = _time_call(lambda: FG.spmm(full.A, x), 50)
        print(f"{name:8s} {tag:7s}: {t * 1e6:7.1f} us  ({floor_us / (t * 1e6):.3f} of the 8 TB/s floor {floor_us:.2f} us)", flush=True)
    for tag in ("sliced", "rowslice"):
        print(f"{name:8s} max rel diff {tag} vs seg: {((res[tag] - res['seg']).abs().max() / res['seg'].abs().max()).item():.2e}", flush=True)
