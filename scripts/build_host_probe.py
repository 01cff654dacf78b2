"""Host time of one chunk plan (BatchChunk.build): Python part vs the native call, for fresh 20- and 150-batch chunks.
Usage (GPU box): python scripts/build_host_probe.py"""
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
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)).to(dev)
lab = synth.make_labels(n, 15509.0 / 3700550.0, 72).astype(np.int32)
sp = split_dgraphfin(lab, 72, with_test=False)
sched = BatchSchedule(sp['idx_train'], sp['idx_anomaly'], sp['labels'], 150, PyCompatRandom.from_python_state(random.getstate()))
tr = DGraphTrainer(g, feat, 64, sched, overlap=False)
ch = tr.chunk
native = ch.lib.ggad_mb_plan_build
spent = []


def timed(*a):
    t0 = time.perf_counter()
    r = native(*a)
    spent.append(time.perf_counter() - t0)
    return r


class Lib:
    def __getattr__(self, k):
        return timed if k == "ggad_mb_plan_build" else getattr(ch.__dict__["_real_lib"], k)


ch.__dict__["_real_lib"] = ch.lib
ch.lib = Lib()
bn, bl = sched.next_batches(150)
ch.build(bn, bl)
torch.cuda.synchronize()
for k in (20, 150):
    tot = []
    spent.clear()
    for rep in range(6):
        bn, bl = sched.next_batches(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ch.build(bn, bl)
        tot.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print(f"{k:4d} batches: build() returns after {1e6 * np.mean(tot[1:]):.0f} us, of which the native call {1e6 * np.mean(spent[1:]):.0f} us")
