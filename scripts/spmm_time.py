"""N x N x 300 product of the full-graph path alone, every kernel variant:  python scripts/spmm_time.py [t_finance|Amazon ...]"""
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
for name in (sys.argv[1:] or ["t_finance", "Amazon"]):
    random.seed(0); np.random.seed(0)
    ds = make_dataset(name, 0)
    n = ds["n"]
    full = FG.FullGraphAdj(normalize_adj(ds["adj"]) + sp.eye(n), ds["adj"] + sp.eye(n), dev)
    x = torch.randn(n, 300, device=dev)
    b = torch.randn(300, device=dev)
    a = torch.tensor([0.25], device=dev)
    os.environ["GGAD_SPMM_PANEL"] = "0"
    ref = FG.spmm(full.A, x, bias=b, prelu_a=a)
    t = _time_call(lambda: FG.spmm(full.A, x), 30)
    print(name, "sliced  N x N x 300: %.1f us" % (t * 1e6), flush=True)
    os.environ["GGAD_SPMM_PANEL"] = "1"
    import time
    t0 = time.time(); full.A.value_factors(); t1 = time.time(); pp = full.A.panel_plan(10); t2 = time.time()
    print(name, "host: value factors %.2f s, panel plan %.2f s" % (t1 - t0, t2 - t1), flush=True)
    if pp is None:
        print(name, "no panel plan"); continue
    got = FG.spmm(full.A, x, bias=b, prelu_a=a)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    t = _time_call(lambda: FG.spmm(full.A, x), 30)
    print(name, "panel   N x N x 300: %.1f us   (fill %.2f, %d blocks x %d rounds, %d workgroups; max rel diff to sliced %.2e)"
          % (t * 1e6, pp["fill"], pp["blocks"], pp["rounds"], pp["n_wg"], err), flush=True)
