"""The full-graph leg of bench.py alone: python scripts/fullgraph_leg.py [names...] [--epochs N]"""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd.fullgraph_bench import bench_fullgraph  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith("--")] or None
res = bench_fullgraph(torch.device("cuda:0"), 30, names)
for k, v in res.items():
    print(k, json.dumps(v))
