"""Time of the N x N x 300 product through k_spmm_ring alone (HIP events around the kernel launches; the slice-major re-layout is
timed apart):  python scripts/ring_time.py [dataset] [label]"""
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, ".")
from ggad_amd import _lib, fullgraph as FG, fullgraph_bench as FB          # noqa: E402
from ggad_amd.utils import normalize_adj                                   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "t_finance"
label = sys.argv[2] if len(sys.argv) > 2 else "baseline"
dev = torch.device("cuda:0")
import os
cache = f"/tmp/ring_time_{name}.npz"                          # (the timing variants run one process each: build the graph once)
if os.path.exists(cache):
    z = np.load(cache)
    mat = sp.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
else:
    ds = FB.make_dataset(name)
    mat = (normalize_adj(ds["adj"]) + sp.eye(ds["n"])).tocsr()
    np.savez(cache, data=mat.data, indices=mat.indices, indptr=mat.indptr, shape=np.array(mat.shape))
n = mat.shape[0]
csr = FG.Csr(mat, dev)
x = torch.from_numpy(np.random.default_rng(1).standard_normal((n, 300)).astype(np.float32)).to(dev)
t0 = time.time()
pp = FG._use_panel(csr, csr.plan(), x)
t_plan = time.time() - t0
t = FB._time_call(lambda: FG.spmm(csr, x), reps=30)
print(f"{name} {label}: {t * 1e6:.1f} us per product incl. re-layout; plan {t_plan:.2f} s fill {pp['fill']:.3f} skew {pp.get('phase_skew', 0):.3f}", flush=True)
