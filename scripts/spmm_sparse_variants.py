"""N x N x 300 product on the SPARSE-neighbourhood configs (Reddit, Photo): wave-per-segment kernel against the XCD-sliced kernel and the
column-sliced six-rows-per-wave kernel (k_spmm_rowslice) and its line-granular form on 128-byte aligned rows (rowline).  Usage (GPU box): python scripts/spmm_sparse_variants.py [reddit photo]"""
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
    xp = torch.empty(n, 320, device=dev)[:, :300]
    xp.copy_(x)
    assert FG._use_rowline(xp)
    rsp = full.A.rowslice_plan(full.A.plan(), lines=True)
    print(name, "rowline plan: units", rsp["n_units"], "medium", rsp["n_long"], "hub", rsp["n_hub"], flush=True)
    res["rowline"] = FG.spmm(full.A, xp).clone()
    t = _time_call(lambda: FG.spmm(full.A, xp), 50)
    print(f"{name:8s} rowline: {t * 1e6:7.1f} us  ({floor_us / (t * 1e6):.3f} of the 8 TB/s floor {floor_us:.2f} us)", flush=True)
    op = torch.empty(n, 320, device=dev)[:, :300]
    res["rowline_po"] = FG.spmm(full.A, xp, out=op).clone()
    t = _time_call(lambda: FG.spmm(full.A, xp, out=op), 50)
    print(f"{name:8s} rowline, padded out: {t * 1e6:7.1f} us", flush=True)
    for tag in ("sliced", "rowslice", "rowline", "rowline_po"):
        print(f"{name:8s} max rel diff {tag} vs seg: {((res[tag] - res['seg']).abs().max() / res['seg'].abs().max()).item():.2e}", flush=True)
