"""Photo's first layer (F = 745 features, not a multiple of 4): x W^T and dW = dZ^T x as they run today (K = 745: per-float loads of the generic
tiles) against operands padded to K = 748 (16-byte chunks: the LDS-DMA tiles / split-K)."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from ggad_amd.fullgraph import gemm  # noqa: E402
from ggad_amd.fullgraph_bench import _time_call  # noqa: E402

torch.manual_seed(0)
n, f, h = 7535, 745, 300
x = torch.randn(n, f, device="cuda")
w = torch.randn(h, f, device="cuda") * 0.05
dz = torch.randn(n, h, device="cuda")
fp = (f + 3) // 4 * 4
xp = torch.zeros(n, fp, device="cuda"); xp[:, :f] = x
wp = torch.zeros(h, fp, device="cuda"); wp[:, :f] = w
for tag, fn in (("x W^T           K=745", lambda: gemm(x, w, False, True)), ("x W^T           K=748", lambda: gemm(xp, wp, False, True)),
                ("dW = dZ^T x     N=745", lambda: gemm(dz, x, True, False)), ("dW = dZ^T x     N=748", lambda: gemm(dz, xp, True, False)),
                ("pad W (copy)         ", lambda: wp[:, :f].copy_(w)), ("unpad dW (contiguous)", lambda: wp[:, :f].contiguous())):
    t = _time_call(fn, 30)
    print(f"{tag}: {t * 1e6:6.1f} us  {2.0 * n * f * h / t / 1e12:5.1f} TF", flush=True)
a, b = gemm(x, w, False, True), gemm(xp, wp, False, True)
print("max rel diff fwd", ((a - b).abs().max() / a.abs().max()).item())
a, b = gemm(dz, x, True, False), gemm(dz, xp, True, False)[:, :f]
print("max rel diff dW ", ((a - b).abs().max() / a.abs().max()).item())
