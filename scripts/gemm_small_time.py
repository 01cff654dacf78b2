"""The small products of the outlier head (fc4 on the 239-843 abnormal rows and its gradients) with k_gemm_small against the tiled kernels:
python scripts/gemm_small_time.py  (one process per setting)"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    from ggad_amd.fullgraph import gemm
    from ggad_amd.fullgraph_bench import _time_call
    torch.manual_seed(0)
    for (m, k, n, ta, tb, tag) in [(239, 300, 300, False, True, "fc4 fwd (reddit)"), (239, 300, 300, False, False, "d_pre"), (300, 239, 300, True, False, "dw4"),
                                   (843, 300, 300, False, True, "fc4 fwd (t_finance)"), (300, 843, 300, True, False, "dw4 (t_finance)"), (127, 300, 300, False, True, "photo")]:
        a = torch.randn((k, m) if ta else (m, k), device="cuda")
        b = torch.randn((n, k) if tb else (k, n), device="cuda")
        t = _time_call(lambda: gemm(a, b, ta, tb), 30)
        print(f"{sys.argv[1]:>6s} {tag:22s} M={m:4d} N={n:4d} K={k:4d}: {t * 1e6:6.1f} us", flush=True)
else:
    for tag, env in (("small", {}), ("tiled", {"GGAD_GEMM_SMALL": "0"})):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, **env), check=False)
