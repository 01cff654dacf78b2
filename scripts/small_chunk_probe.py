"""Fixed cost of small chunks (the driver's `bench.py --steps 20 --warmup 5` regime): host and device time of
`BatchChunk.build` and `MiniBatchEngine.train_chunk` as a function of the number of batches per chunk, alone on the chip.
Usage (GPU box): python scripts/small_chunk_probe.py"""
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
torch.manual_seed(0)
tr.engine.load_params(torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)),
                      torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
bn, bl = sched.next_batches(150)
tr.chunk.build(bn, bl)
tr.engine.train_chunk(tr.chunk)
torch.cuda.synchronize()
for k in (1, 2, 3, 5, 10, 20, 40, 150):
    res = []
    for rep in range(4):
        bn, bl = sched.next_batches(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.chunk.build(bn, bl)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tr.engine.train_chunk(tr.chunk)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        res.append((t1 - t0, t2 - t0, t3 - t2, t4 - t2))
    r = np.array(res[1:]).mean(0) * 1e3
    print(f'batches {k:4d}: build host {r[0]:.3f} ms, build done {r[1]:.3f} ms ({1e3 * r[1] / k:.1f} us/batch) | '
          f'train host {r[2]:.3f} ms, done {r[3]:.3f} ms ({1e3 * r[3] / k:.1f} us/batch)', flush=True)
