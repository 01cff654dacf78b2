"""k_gemm_slab (round 5: 80-column slabs resident in LDS, row blocks dealt to SIMDs) against the round-4 kernels on the tall products of a
full-graph epoch, with a check against torch in float64: one process per setting (the switches are read once).
    python scripts/gemm_slab_ab.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, __file__.rsplit("/", 2)[0])
    from ggad_amd.fullgraph import gemm
    from ggad_amd.fullgraph_bench import _time_call
    dev = "cuda"
    torch.manual_seed(0)
    for (m, k, n, ta, tb, tag) in [(10984, 300, 300, False, True, "reddit x W^T"), (10984, 300, 300, False, False, "reddit dz W"),
                                   (7535, 300, 300, False, True, "photo x W^T"), (7535, 300, 300, False, False, "photo dz W"),
                                   (11944, 300, 300, False, True, "amazon x W^T"), (11944, 300, 300, False, False, "amazon dz W"),
                                   (39357, 300, 300, False, True, "t_finance x W^T"), (39357, 300, 300, False, False, "t_finance dz W"),
                                   (10984, 256, 512, False, True, "256 -> 512"), (5000, 320, 260, False, False, "320 -> 260"),
                                   (4111, 252, 196, False, True, "ragged 252 -> 196")]:
        a = torch.randn((k, m) if ta else (m, k), device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        bias = torch.randn(n, device=dev)
        got = gemm(a, b, ta, tb, bias=bias, relu=True)
        ref = torch.relu((a.double().T if ta else a.double()) @ (b.double().T if tb else b.double()) + bias.double())
        err = ((got.double() - ref).abs().max() / (ref.abs().max() + 1.0)).item()
        t = _time_call(lambda: gemm(a, b, ta, tb), 30)
        print(f"{sys.argv[1]:>8s} {tag:20s} M={m:6d} N={n:4d} K={k:4d}: {t * 1e6:7.1f} us  {2.0 * m * n * k / t / 1e12:6.1f} TF  ({2.0 * m * n * k / t / 155e12:.2f} of 155)  max err {err:.1e}", flush=True)
else:
    for tag, env in (("slab", {}), ("r04", {"GGAD_GEMM_SLAB": "0"})):
        subprocess.run([sys.executable, __file__, tag], env=dict(os.environ, **env), check=False)
