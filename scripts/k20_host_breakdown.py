"""Where the host time of the driver's K = 20 window goes: python preparation of the build, the native plan call, the chunk-kernel call,
and the wait for the GPU -- run_steps(20) on prepared batches, like bench.py's timed region.  Usage (GPU box): python scripts/k20_host_breakdown.py"""
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
torch.manual_seed(0)
tr.engine.load_params(torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)),
                      torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
tr.run_steps(20, prepared=sched.next_batches(20)); torch.cuda.synchronize()
ch, eng = tr.chunk, tr.engine
for rep in range(5):
    bn, bl = sched.next_batches(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.run_steps(20, prepared=(bn, bl))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # the same, piece by piece
    bn, bl = sched.next_batches(20)
    torch.cuda.synchronize()
    a0 = time.perf_counter()
    ch.xcd_skip = -1
    ch.build(bn, bl)
    a1 = time.perf_counter()
    eng.train_chunk(ch)
    a2 = time.perf_counter()
    torch.cuda.synchronize()
    a3 = time.perf_counter()
    print(f"run_steps(20): host {1e6 * (t1 - t0):7.1f} us, until the GPU is done {1e6 * (t2 - t0):7.1f} us | build() host {1e6 * (a1 - a0):6.1f}, "
          f"train_chunk() host {1e6 * (a2 - a1):6.1f}, wait {1e6 * (a3 - a2):7.1f}, total {1e6 * (a3 - a0):7.1f} us", flush=True)

# GPU-side view of the same window: an event recorded at t0 (the stream is idle: it completes at once) and one behind the chunk
# kernel -- their distance minus the kernels' own time is what the GPU waited for the host before its first kernel
for rep in range(4):
    bn, bl = sched.next_batches(20)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ea.record()
    tr.run_steps(20, prepared=(bn, bl))
    eb.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"window: host returns after {1e6 * (t1 - t0):6.1f} us, GPU from the t0 marker to the end of the chunk kernel {1e3 * ea.elapsed_time(eb):7.1f} us, "
          f"wall until synchronize returns {1e6 * (t2 - t0):7.1f} us", flush=True)
