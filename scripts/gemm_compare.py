#!/usr/bin/env python3
"""k_gemm_f32 (exact-f32 MFMA) vs the library GEMM torch.mm dispatches to (hipBLASLt / rocBLAS) on the full-graph shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggad_amd.fullgraph import gemm

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

torch.backends.cuda.matmul.allow_tf32 = False
for (M, K, N, tag) in [(10984, 64, 300, "reddit L1"), (10984, 300, 300, "reddit L2"), (7535, 745, 300, "photo L1"), (39357, 300, 300, "tfin L2"),
                       (300, 10984, 300, "wgrad reddit (TN)"), (4096, 4096, 4096, "square")]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    mine = t(lambda: gemm(a, b, False, True))
    lib = t(lambda: torch.mm(a, b.t()))
    fl = 2.0 * M * K * N
    err = (gemm(a, b, False, True) - torch.mm(a, b.t())).abs().max().item()
    print(f"{tag:20s} M={M:6d} K={K:5d} N={N:4d}  mine {mine*1e6:8.1f} us {fl/mine/1e12:6.1f} TF   lib {lib*1e6:8.1f} us {fl/lib/1e12:6.1f} TF   max|diff| {err:.2e}")
