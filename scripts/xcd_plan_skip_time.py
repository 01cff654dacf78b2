"""Plan of a 150-batch chunk ALONE on the chip, wall time per build (events), four ways: plain launches on the null stream; the
index-skipping launches (one XCD left out) on the null stream; the same on the trainer's plan stream (CU mask: 4 CUs of XCD 0 +
31 of each other XCD); plain launches on that masked stream.  Usage (GPU box): python scripts/xcd_plan_skip_time.py"""
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ggad_amd import synth  # noqa: E402
from ggad_amd.dgraph import normalize_features, split_dgraphfin  # noqa: E402
from ggad_amd.graph import DeviceGraph  # noqa: E402
from ggad_amd.sampler import PyCompatRandom  # noqa: E402
from ggad_amd.trainer import BatchSchedule, DGraphTrainer  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)).to(dev)
lab = synth.make_labels(n, 15509.0 / 3700550.0, 72).astype(np.int32)
sp = split_dgraphfin(lab, 72, with_test=False)
sched = BatchSchedule(sp['idx_train'], sp['idx_anomaly'], sp['labels'], 150, PyCompatRandom.from_python_state(random.getstate()))
tr = DGraphTrainer(g, feat, 64, sched)
ch = tr.chunks[0]
bn, bl = sched.next_batches(150)
null = torch.cuda.default_stream(dev)
for name, st, skip in (("null stream, plain", null, -1), ("null stream, skipping", null, tr._xcd_skip),
                       ("plan stream (masked), skipping", tr.side, tr._xcd_skip), ("plan stream (masked), plain", tr.side, -1)):
    ts = []
    for rep in range(4):
        ch.xcd_skip = skip
        with torch.cuda.stream(st):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ch.build(bn, bl)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    print(f"{name:34s}: build of 150 batches best {1e3 * min(ts):.2f} ms, median {1e3 * sorted(ts)[len(ts) // 2]:.2f} ms", flush=True)
