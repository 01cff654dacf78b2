"""Reddit's first layer (64 features): (A_hat X) W^T + b, M = 10,984, K = 64, N = 300 -- the slab kernel (KS = 4) against the tiled kernels."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    from ggad_amd.fullgraph import gemm
    from ggad_amd.fullgraph_bench import _time_call
    torch.manual_seed(0)
    for (m, k, n) in [(10984, 64, 300), (39357, 20, 300), (11944, 28, 300), (10984, 52, 300)]:
        x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.1; b = torch.randn(n, device="cuda")
        got = gemm(x, w, False, True, bias=b)
        ref = x.double() @ w.double().T + b.double()
        err = ((got.double() - ref).abs().max() / (ref.abs().max() + 1.0)).item()
        t = _time_call(lambda: gemm(x, w, False, True, bias=b), 30)
        print(f"{sys.argv[1]:>6s} M={m:6d} K={k:3d} N={n}: {t * 1e6:6.1f} us  err {err:.1e}", flush=True)
else:
    for tag, env in (("slab", {}), ("tiles", {"GGAD_GEMM_SLAB": "0"})):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, **env), check=False)
