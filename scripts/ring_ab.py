"""LDS-ring product (k_spmm_ring) against the LDS-panel product (k_spmm_panel) on the published dense-neighbourhood sizes:
results (max relative difference, determinism) and time per N x N x 300 product, for both workgroup -> XCD maps of the ring.

    python scripts/ring_ab.py [t_finance|Amazon ...]           (GPU)
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, ".")
from ggad_amd import fullgraph as FG                            # noqa: E402
from ggad_amd import fullgraph_bench as FB                      # noqa: E402
from ggad_amd.utils import normalize_adj                        # noqa: E402

dev = torch.device("cuda:0")
for name in (sys.argv[1:] or ["Amazon", "t_finance"]):
    ds = FB.make_dataset(name)
    n = ds["n"]
    csr = FG.Csr(normalize_adj(ds["adj"]) + sp.eye(n), dev)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, 300)).astype(np.float32)).to(dev)
    res = {}
    for label, env in (("panel", {"GGAD_SPMM_RING": "0"}), ("ring/slice", {"GGAD_SPMM_RING": "1", "GGAD_RING_XCD": "slice"}),
                       ("ring/block", {"GGAD_SPMM_RING": "1", "GGAD_RING_XCD": "block"})):
        os.environ.update(env)
        pp = FG._use_panel(csr, csr.plan(), x)
        out = FG.spmm(csr, x)
        torch.cuda.synchronize()
        again = FG.spmm(csr, x)
        t = FB._time_call(lambda: FG.spmm(csr, x), reps=20)
        res[label] = out
        ref = res["panel"]
        print(f"{name} {label}: {t * 1e6:.1f} us per product (with the slice-major re-layout)  fill {pp['fill']:.3f}  "
              f"max rel diff vs panel {((out - ref).abs().max() / ref.abs().max()).item():.2e}  deterministic {torch.equal(out, again)}",
              flush=True)
