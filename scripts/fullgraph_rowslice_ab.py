import os, sys, json
sys.path.insert(0, '.')
import torch
from ggad_amd.fullgraph_bench import bench_fullgraph
for env in ("0", "1"):
    os.environ["GGAD_SPMM_ROWSLICE"] = env
    r = bench_fullgraph(torch.device("cuda:0"), 30, ["reddit", "photo"])
    print("ROWSLICE", env, {k: (round(v["epoch_ms"], 4), round(v["spmm_NxNxH"]["us"], 1), v["spmm_NxNxH"]["kernel"]) for k, v in r.items()}, flush=True)
