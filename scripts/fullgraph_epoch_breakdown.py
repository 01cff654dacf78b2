#!/usr/bin/env python3
"""Where does a full-graph epoch spend its time on this box?  CPU noise draw, host->device copy, GPU work."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ds = sys.argv[1] if len(sys.argv) > 1 else "t_finance"
A, H = {"t_finance": 844, "reddit": 238, "photo": 153, "Amazon": 83}[ds], 300
torch.manual_seed(0)
t = []
for _ in range(20):
    t0 = time.perf_counter(); n = torch.randn(1, A, H) * 0.0 + 0.0; t1 = time.perf_counter()
    d = n.to("cuda:0"); torch.cuda.synchronize(); t2 = time.perf_counter()
    t.append((t1 - t0, t2 - t1))
t = np.array(t[5:]) * 1e3
print(f"{ds}: randn({A}x{H}) {np.median(t[:,0]):.3f} ms (max {t[:,0].max():.2f}), H2D+sync {np.median(t[:,1]):.3f} ms; torch threads {torch.get_num_threads()}, cpus {os.cpu_count()}, load {os.getloadavg()}")
