"""Where the waves of k_gather2_items spend their time (a library built with GGAD_EXTRA_HIPFLAGS=-DGGAD_G2_PROF): per-phase wall
clocks summed over the waves of a launch, every mark behind s_waitcnt 0.  Usage: python scripts/g2_phase_clocks.py 20,150 [reps]"""
import random
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import synth  # noqa: E402
from ggad_amd.dgraph import normalize_features, split_dgraphfin  # noqa: E402
from ggad_amd.graph import DeviceGraph  # noqa: E402
from ggad_amd.minibatch import BatchChunk  # noqa: E402
from ggad_amd.sampler import PyCompatRandom  # noqa: E402
from ggad_amd.trainer import BatchSchedule  # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,150").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device('cuda:0')
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)).to(dev)
table = torch.zeros(n, 32, dtype=torch.float32, device=dev)
table[:, :17] = feat
lab = synth.make_labels(n, 15509.0 / 3700550.0, 72).astype(np.int32)
sp = split_dgraphfin(lab, 72, with_test=False)
sched = BatchSchedule(sp['idx_train'], sp['idx_anomaly'], sp['labels'], 150, PyCompatRandom.from_python_state(random.getstate()))
ch = BatchChunk(g, table, 64, 150, 150 * 200, 1 << 20, train=True, feat_dim=17, hop2="ldsw")
names = ["cursor", "item metadata", "first ids/counts", "rows issue+wait", "weights+fma", "stores", "wave total"]
for k in sizes:
    for rep in range(reps):
        bn, bl = sched.next_batches(k)
        ch.build(bn, bl)
        torch.cuda.synchronize()
        c32 = ch.counters.cpu().numpy()
        c = c32[::16]
        prof = c32[14 * 16:].view(np.int64)[:7].astype(np.float64) / 100.0        # 100 MHz -> us
        tot = prof[6]
        print(f"chunk {k}: items {int(c[1])} big groups {int(c[5])} groups {int(c[0])} pairs {int(c[4])} | wave-us total {tot:.0f}: " +
              "  ".join(f"{nm} {100 * v / tot:.1f}%" for nm, v in zip(names[:6], prof[:6])) +
              f"  (other {100 * (tot - prof[:6].sum()) / tot:.1f}%)", flush=True)
        tc = c32[14 * 16:].view(np.int64)[8:15].astype(np.float64)
        if tc[6] > 0:
            nm2 = ["zero + tables", "scan", "pass 0 cached", "pass 0 beyond 16 K", "pass 1 cached", "pass 1 beyond 16 K"]
            print(f"   k_tile_counts: {int(tc[6])} workgroups, {tc[:6].sum() / 100.0 / tc[6]:.1f} us each: " +
                  "  ".join(f"{n_} {v / 100.0 / tc[6]:.2f} us" for n_, v in zip(nm2, tc[:6])), flush=True)
