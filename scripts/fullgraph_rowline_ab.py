"""Epoch time of the sparse-neighbourhood configs with / without the line-granular product (padded rows: GGAD_SPMM_ROWLINE).
Usage (GPU box): python scripts/fullgraph_rowline_ab.py"""
import os
import sys

sys.path.insert(0, '.')
import torch  # noqa: E402

from ggad_amd.fullgraph_bench import bench_fullgraph  # noqa: E402

for rep in range(2):
    for env in ("0", "1"):
        os.environ["GGAD_SPMM_ROWLINE"] = env
        r = bench_fullgraph(torch.device("cuda:0"), 30, ["reddit", "photo"])
        print("ROWLINE", env, {k: round(v["epoch_ms"], 4) for k, v in r.items()}, flush=True)
