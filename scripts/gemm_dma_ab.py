"""k_gemm_dma (LDS-DMA staging, round 4) against k_gemm_f32 (register staging) on the products of a full-graph epoch: one process per
kernel (GGAD_GEMM_DMA is read once).   python scripts/gemm_dma_ab.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ".")
    from ggad_amd.fullgraph import gemm
    from ggad_amd.fullgraph_bench import _time_call
    dev = "cuda"
    for (m, k, n, ta, tb, tag) in [(10984, 300, 300, False, True, "reddit x W^T"), (10984, 300, 300, False, False, "reddit dz W"),
                                   (300, 10984, 300, True, False, "reddit dz^T x (split-K)"), (7535, 300, 300, False, True, "photo x W^T"),
                                   (11944, 300, 300, False, True, "amazon x W^T"), (39357, 300, 300, False, True, "t_finance x W^T"),
                                   (39357, 300, 300, False, False, "t_finance dz W"), (300, 39357, 300, True, False, "t_finance dz^T x"),
                                   (10984, 64, 300, False, True, "reddit layer 1"), (4096, 4096, 4096, False, True, "4096^3")]:
        a = torch.randn((k, m) if ta else (m, k), device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        t = _time_call(lambda: gemm(a, b, ta, tb), 30)
        print(f"{sys.argv[1]:>4s} {tag:26s} M={m:6d} N={n:4d} K={k:6d}: {t * 1e6:8.1f} us  {2.0 * m * n * k / t / 1e12:6.1f} TF  ({2.0 * m * n * k / t / 155e12:.2f} of 155)", flush=True)
else:
    for v in ("0", "1"):
        subprocess.run([sys.executable, __file__, "dma" + v], env=dict(os.environ, GGAD_GEMM_DMA=v), check=False)
