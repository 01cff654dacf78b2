"""Dense steps of a chunk: the XCD-resident launch (csrc/step_xcd.hip) against the 5-launch chain, alone on the chip, DGraph-size
graph.  Prints us per step for both, the phase clocks of the resident kernel, and the largest |difference| of losses / weights.
Usage (GPU box): python scripts/xcd_step_time.py [batches ...]"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ggad_amd import synth  # noqa: E402
from ggad_amd.dgraph import normalize_features, split_dgraphfin  # noqa: E402
from ggad_amd.graph import DeviceGraph  # noqa: E402
from ggad_amd.minibatch import MiniBatchEngine  # noqa: E402
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
tr = DGraphTrainer(g, feat, 64, sched, overlap=False)
torch.manual_seed(0)
w0 = (torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)),
      torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
sizes = [int(a) for a in sys.argv[1:]] or [20, 150]
# GGAD_XCD_DEBUG=4 turns the kernel's phase clocks on (they cost ~1 us per step)
for k in sizes:
    bn, bl = sched.next_batches(k)
    tr.chunk.build(bn, bl)
    torch.cuda.synchronize()
    ents = tr.chunk.n_ents / k
    out = {}
    for name, resident in (("chain", False), ("xcd", True)):
        eng = MiniBatchEngine(17, 64, dev, resident=resident)
        ts = []
        for rep in range(5):
            eng.load_params(*w0)
            eng.reset_optimizer()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.train_chunk(tr.chunk)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        out[name] = (eng.params.cpu().numpy().copy(), eng.losses(k).copy())
        line = f"{k:4d} batches ({ents:.0f} entries, {tr.chunk.n_chunks / k:.0f} pieces per batch)  {name:5s}: best {1e6 * min(ts) / k:6.2f} us/step, median {1e6 * sorted(ts)[2] / k:6.2f}"
        if resident:
            st = eng.xcd_status()
            ph = "  ".join(f"{a} {b / k:.2f}" for a, b in st["phase_us"].items())
            line += f"  | wgs {st['workgroups']} xcc {st['xcc']} err {st['error']} | us/step by phase: {ph} | sub: " + " ".join(f"{x / k:.2f}" for x in st["sub_us"][:8])
        print(line, flush=True)
    dl = np.abs(out["xcd"][1][:, :4] - out["chain"][1][:, :4]).max()
    dw = np.abs(out["xcd"][0] - out["chain"][0]).max()
    print(f"     max |d loss| {dl:.2e}  max |d params| {dw:.2e}  first-step losses equal: {np.array_equal(out['xcd'][1][0], out['chain'][1][0])}",
          flush=True)
