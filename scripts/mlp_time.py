"""Fused scorer-MLP kernels against the GEMM launches they replace (rows = normal + outlier rows of each config):  python scripts/mlp_time.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ggad_amd import fullgraph as FG                       # noqa: E402
from ggad_amd.fullgraph_bench import _time_call            # noqa: E402

dev = torch.device("cuda:0")
h, h1, h2 = 300, 150, 75
for name, r in (("reddit", 1830), ("photo", 1179), ("Amazon", 1751), ("t_finance", 6476)):
    x = torch.randn(r, h, device=dev)
    w1, w2, w3 = torch.randn(h1, h, device=dev) / 17, torch.randn(h2, h1, device=dev) / 12, torch.randn(1, h2, device=dev) / 9
    g3, gx = torch.randn(r, 1, device=dev), torch.randn(r, h, device=dev)
    f1, f2, f3 = FG.mlp_score_fwd(x, w1, w2, w3)
    t_f = _time_call(lambda: FG.mlp_score_fwd(x, w1, w2, w3), 50)
    t_b = _time_call(lambda: FG.mlp_score_dgrad(g3, f1, f2, w1, w2, w3, gx), 50)

    def gemm_fwd():
        a = FG.gemm(x, w1, False, True, relu=True)
        b = FG.gemm(a, w2, False, True, relu=True)
        return FG.gemm(b, w3, False, True)

    def gemm_bwd():
        df2 = FG.gemm(g3, w3, False, False)
        dz2 = torch.empty_like(df2)
        FG.call("ggad_relu_bwd_f32", FG.ptr(df2), FG.ptr(f2), df2.numel(), FG.ptr(dz2))
        df1 = FG.gemm(dz2, w2, False, False)
        dz1 = torch.empty_like(df1)
        FG.call("ggad_relu_bwd_f32", FG.ptr(df1), FG.ptr(f1), df1.numel(), FG.ptr(dz1))
        return FG.gemm(dz1, w1, False, False) + gx
    t_gf = _time_call(gemm_fwd, 50)
    t_gb = _time_call(gemm_bwd, 50)
    print(f"{name} rows {r}: forward fused {t_f * 1e6:.1f} us vs 3 GEMMs {t_gf * 1e6:.1f} us; data gradients fused {t_b * 1e6:.1f} us vs "
          f"3 GEMMs + 2 relu + add {t_gb * 1e6:.1f} us   (eager launches, back to back)", flush=True)
