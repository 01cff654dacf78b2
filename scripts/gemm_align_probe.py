"""Does the leading dimension of the operands matter for the K = 300 projections?  ggad_gemm_f32 on x (M x 300) w^T (300 x 300)
with rows of 300 floats (1,200 B: a 128-byte row chunk straddles two cache lines) against rows padded to 320 floats."""
import sys
import time

import torch

sys.path.insert(0, '.')
from ggad_amd._lib import call, ptr  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for M in (10984, 39357):
    N = K = 300
    for ld in (300, 320, 384):
        A = torch.randn(M, ld, device="cuda")
        B = torch.randn(N, ld, device="cuda")
        C = torch.empty(M, N, device="cuda")
        dt = t(lambda: call("ggad_gemm_f32", ptr(A), ptr(B), ptr(C), M, N, K, ld, 1, 1, ld, N, 0, 0, 0))
        print(f"M={M} K=300 N=300 leading dimension {ld}: {dt * 1e6:7.1f} us  {2.0 * M * N * K / dt / 1e12:6.1f} TF")
