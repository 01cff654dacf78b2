"""Per-kernel timing of the full-graph path at the BASELINE dataset sizes (synthetic graphs): SpMM, GEMMs."""
import sys, json
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from ggad_amd import synth, fullgraph as FG, utils as U
from run import SIZES
dev = torch.device('cuda:0')
H = 300

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

rows = []
for ds in ['reddit', 'photo', 'Amazon', 't_finance']:
    n, ne, f, _ = SIZES[ds]
    rp, ci = synth.make_graph(n, ne, 0, kind='powerlaw', max_degree=max(64, n // 8))
    a = synth.csr_to_scipy(rp, ci, n)
    fa = FG.FullGraphAdj(U.normalize_adj(a) + sp.eye(n), a + sp.eye(n), dev)
    nnz = fa.A.nnz
    X = torch.randn(n, H, device=dev)
    bias = torch.randn(H, device=dev); pa = torch.full((1,), 0.25, device=dev)
    t = timeit(lambda: FG.spmm(fa.A, X, bias=bias, prelu_a=pa, want_pre=True))
    alg = 8 * nnz + 4 * (n + 1) + 2 * 4 * n * H
    flops = 2 * nnz * H
    rows.append(dict(ds=ds, op='spmm A_hat(N x N) @ (N x 300) +bias+prelu', n=n, nnz=nnz, us=t * 1e6, GBs=alg / t / 1e9, TFs=flops / t / 1e12,
                     hbm_frac=alg / t / 8e12, fma_frac=flops / t / 157.3e12))
    Xf = torch.randn(n, f, device=dev); W1 = torch.randn(H, f, device=dev); W2 = torch.randn(H, H, device=dev)
    for name, fn, fl in [(f'gemm X W1^T ({n}x{f}x{H})', lambda: FG.gemm(Xf, W1, False, True), 2 * n * f * H),
                         (f'gemm h W2^T ({n}x{H}x{H})', lambda: FG.gemm(X, W2, False, True), 2 * n * H * H),
                         (f'gemm dT W2 NN ({n}x{H}x{H})', lambda: FG.gemm(X, W2, False, False), 2 * n * H * H),
                         (f'gemm dT^T h TN wgrad ({H}x{n}x{H})', lambda: FG.gemm(X, X, True, False), 2 * n * H * H)]:
        t = timeit(fn)
        rows.append(dict(ds=ds, op=name, us=t * 1e6, TFs=fl / t / 1e12, mfma_frac=fl / t / 157.3e12))
for r in rows:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
