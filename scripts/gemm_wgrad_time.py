"""The GCN layers' weight gradients dW = dZ^T X (K = number of nodes) with the row-range kernel against the split-K tiles: one process per setting."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    from ggad_amd.fullgraph import gemm
    from ggad_amd.fullgraph_bench import _time_call
    torch.manual_seed(0)
    for (m, k, n, tag) in [(300, 10984, 300, "reddit layer 2"), (300, 10984, 64, "reddit layer 1"), (300, 7535, 300, "photo layer 2"), (300, 7535, 745, "photo layer 1"),
                           (300, 11944, 300, "amazon layer 2"), (300, 39357, 300, "t_finance layer 2"), (300, 39357, 10, "t_finance layer 1")]:
        a = torch.randn(k, m, device="cuda")
        b = torch.randn(k, n, device="cuda")
        got = gemm(a, b, True, False)
        ref = a.double().T @ b.double()
        err = ((got.double() - ref).abs().max() / (ref.abs().max() + 1.0)).item()
        t = _time_call(lambda: gemm(a, b, True, False), 30)
        print(f"{sys.argv[1]:>6s} {tag:20s} M={m:4d} N={n:4d} K={k:6d}: {t * 1e6:6.1f} us  {2.0 * m * n * k / t / 1e12:5.1f} TF  err {err:.1e}", flush=True)
else:
    for tag, env in (("ranges", {}), ("tiles", {"GGAD_GEMM_WGRAD_TN": "0"})):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, **env), check=False)
