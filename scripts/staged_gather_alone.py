"""(main: only the one-list plan; the staged lists live on branch r05-staged2.)  The 2-hop gather of a K-batch chunk alone (no chunk kernel beside it): one list (usual plan) against the staged plan's lists worked
through by ONE launch (round 5, k_gather2_items SG) -- groups, work items per list, and the gather's time between its events.
Usage: python scripts/staged_gather_alone.py [K=20] [reps=3]"""
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

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
import ctypes  # noqa: E402
from ggad_amd import _lib  # noqa: E402
lib = _lib.load()
ev = [ctypes.c_void_p(), ctypes.c_void_p()]
for h in ev:
    _lib.check(lib.ggad_event_create(1, ctypes.byref(h)), "ggad_event_create")
ch.gather2_events = (ev[0], ev[1])
for rep in range(reps):
    bn, bl = sched.next_batches(K)
    for n_st in (0,):      # (< 0: two stages, the first of that many batches)
        ch.stages = ([(K * q) // n_st for q in range(1, n_st)] if n_st > 0 else [-n_st]) if n_st else None
        ch.build(bn, bl)
        ch.stages = None
        if getattr(ch, "staged", False):
            ch.gather_staged(-1)
        torch.cuda.synchronize()
        c = ch.counters.cpu().numpy()[::16]
        lists = [int(c[1]), int(c[6]), int(c[7]), int(c[8])] if n_st else [int(c[1])]
        ms = ctypes.c_float()
        _lib.check(lib.ggad_event_elapsed_ms(ev[0], ev[1], ctypes.byref(ms)), "ggad_event_elapsed_ms")
        print(f"K {K} stages {n_st}: groups {int(c[0])} partial slots {int(c[2])} items per list {lists} (sum {sum(lists)})  "
              f"gather {ms.value * 1e3:.0f} us", flush=True)
        c32 = ch.counters.cpu().numpy()
        prof = c32[14 * 16:].view(np.int64)[:7].astype(np.float64) / 100.0        # (-DGGAD_G2_PROF builds, <= 2 stages: 100 MHz -> us)
        if prof[6] > 0 and n_st <= 2:
            names = ["cursor", "item metadata", "first ids/counts", "rows issue+wait", "weights+fma", "stores"]
            print("    wave-us total %.0f: " % prof[6] + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, prof[:6])) +
                  f"  other {prof[6] - prof[:6].sum():.0f}", flush=True)
            import os
            if os.environ.get("GGAD_G2_WAVES"):      # (a -DGGAD_G2_WAVES build: words 0..3 are maxima, not phase sums)
                w = c32[14 * 16:].view(np.uint64)
                t0 = int(~w[1])
                print(f"    waves: mean {prof[6] / (256 * 3 * 4):.0f} us, longest {int(w[0]) / 100.0:.0f} us, latest start {(int(w[2]) - t0) / 100.0:.0f} us, "
                      f"latest end {(int(w[3]) - t0) / 100.0:.0f} us after the first start", flush=True)
                if n_st:
                    print("    lists retired at " + ", ".join(f"{(int(w[8 + p]) - t0) / 100.0:.0f}" for p in range(2)) + " us", flush=True)
