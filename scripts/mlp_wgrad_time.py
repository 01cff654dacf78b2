"""The scorer's three weight gradients: one fused launch + reduction against the three split-K GEMMs (hipGraph-timed), at the row counts of
the four full-graph configs.  python scripts/mlp_wgrad_time.py"""
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import fullgraph as FG  # noqa: E402
from ggad_amd.fullgraph_bench import _time_call  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
for name, r in (("reddit", 1834), ("photo", 1203), ("amazon", 1888), ("t_finance", 6466)):
    h, h1, h2 = 300, 150, 75
    x = torch.randn(r, h, device=dev)
    dz1, f1 = torch.randn(r, h1, device=dev), torch.randn(r, h1, device=dev)
    dz2, f2 = torch.randn(r, h2, device=dev), torch.randn(r, h2, device=dev)
    g3 = torch.randn(r, 1, device=dev)
    t_f = _time_call(lambda: FG.mlp_score_wgrad(x, dz1, f1, dz2, f2, g3), 30)
    t_g = _time_call(lambda: (FG.gemm(dz1, x, True, False), FG.gemm(dz2, f1, True, False), FG.gemm(g3, f2, True, False)), 30)
    print(f"{name:10s} R = {r:5d}: fused {t_f * 1e6:6.1f} us   three GEMMs {t_g * 1e6:6.1f} us", flush=True)
