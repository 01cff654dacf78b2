#!/usr/bin/env python3
"""Where does the dense-neighbourhood SpMM spend its time?  Same kernel, same segment table and instruction stream, but the
column indices folded into the first K rows of X (col % K): K rows x 1200 B always fit the L2s for small K, so the run time
against K separates "bound by where the rows come from" from "bound by the load path / issue rate"."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ggad_amd import synth  # noqa: E402
from ggad_amd import fullgraph as FG  # noqa: E402
from ggad_amd.utils import normalize_adj  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def main():
    dev = torch.device("cuda", 0)
    n, ne, w = 39357, 21222543, int(os.environ.get("W", 300))
    rowptr, col = synth.make_graph(n, ne, 0, kind="powerlaw", max_degree=n // 8)
    a = synth.csr_to_scipy(rowptr, col, n)
    csr = FG.Csr(normalize_adj(a) + sp.eye(n), dev)
    x = torch.randn(n, w, device=dev)
    col0 = csr.col.clone()
    print(f"nnz {csr.nnz}, W {w}")
    for k in (() if os.environ.get("AB_ONLY") else (256, 1024, 2048, 4096, 8192, 16384, n)):
        csr.col.copy_(col0 % k)
        t = timeit(lambda: FG.spmm(csr, x))
        print(f"rows folded into {k:6d} ({k * w * 4 / 1e6:6.1f} MB): {t * 1e3:7.3f} ms  {csr.nnz * w * 4 / t / 1e12:6.2f} TB/s through L1"
              f"  {2 * csr.nnz * w / t / 1e12:5.2f} TFLOP/s")
    csr.col.copy_(col0)
    # A/B: wave-per-segment row-major gather vs the XCD-sliced kernel, same inputs
    res = {}
    for mode in ("0", "1"):
        os.environ["GGAD_SPMM_SLICED"] = mode
        res[mode] = FG.spmm(csr, x)
        t = timeit(lambda: FG.spmm(csr, x))
        print(f"GGAD_SPMM_SLICED={mode}: {t * 1e3:7.3f} ms  {2 * csr.nnz * w / t / 1e12:5.2f} TFLOP/s")
    d = (res["0"] - res["1"]).abs().max().item()
    print("max |row-major - sliced| =", d, " scale", res["0"].abs().max().item())


if __name__ == "__main__":
    main()
