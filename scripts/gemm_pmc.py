"""GEMM shapes of the full-graph path for a rocprofv3 pass (kernel trace or --pmc):  python scripts/gemm_pmc.py
Launches, in this order, 5 x (10984 x 300 x 300), 5 x (39357 x 300 x 300), 3 x (4096^3) of `ggad_gemm_f32` (NT: x @ w.T)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggad_amd.fullgraph import gemm  # noqa: E402

for m, n, k, reps in ((10984, 300, 300, 5), (39357, 300, 300, 5), (4096, 4096, 4096, 3)):
    a = torch.randn(m, k, device="cuda")
    b = torch.randn(n, k, device="cuda")
    for _ in range(reps):
        gemm(a, b, False, True)
    torch.cuda.synchronize()
