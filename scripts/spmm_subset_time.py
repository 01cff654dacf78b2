"""Row-subset product of the affinity term, (R^T e_hat)[J] (run.py:182-188), with and without the LDS-panel kernel:
python scripts/spmm_subset_time.py [t_finance ...]"""
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
for name in (sys.argv[1:] or ["t_finance"]):
    random.seed(0); np.random.seed(0)
    ds = make_dataset(name, 0)
    n = ds["n"]
    full = FG.FullGraphAdj(normalize_adj(ds["adj"]) + sp.eye(n), ds["adj"] + sp.eye(n), dev)
    rng = np.random.default_rng(0)
    J = rng.permutation(n)[: int(0.165 * n)]
    plan = full.Rt.plan(J, key=("rows", "probe"))
    x = torch.randn(n, 300, device=dev)
    for mode in ("0", None):
        if mode is None:
            os.environ.pop("GGAD_SPMM_PANEL", None)
        else:
            os.environ["GGAD_SPMM_PANEL"] = mode
        pp = FG._use_panel(full.Rt, plan, x)
        out = FG.spmm(full.Rt, x, plan=plan)
        t = _time_call(lambda: FG.spmm(full.Rt, x, plan=plan), 30)
        print(name, "rows %d entries %d: %s %.1f us  checksum %.6e" % (len(J), plan["nnz"], "panel " if pp is not None else "sliced", t * 1e6,
                                                                      float(out.double().sum())), flush=True)
