#!/usr/bin/env python3
"""How often does a node recur as an owner inside one chunk, weighted by its degree (= 2-hop row fetches)?
Usage (GPU box): python scripts/dedup_stats.py [entries]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggad_amd import synth
from ggad_amd.dgraph import split_dgraphfin
from ggad_amd.sampler import PyCompatRandom
from ggad_amd.trainer import BatchSchedule

n = 3700550
entries = int(sys.argv[1]) if len(sys.argv) > 1 else 73105508
rp, ci = synth.make_graph_torch(n, entries, 72, "cuda:0", kind="powerlaw", max_degree=2000)
deg = np.diff(rp).astype(np.int64)
labels0 = synth.make_labels(n, 15509.0 / 3700550.0, 72).astype(np.int32)
split = split_dgraphfin(labels0, 72, with_test=False)
import random as pyrandom
rng = PyCompatRandom.from_python_state(pyrandom.getstate())
sched = BatchSchedule(split["idx_train"], split["idx_anomaly"], split["labels"], 150, rng)
bn, bl = sched.next_batches(150, 0, 1)
occ = np.zeros(n, dtype=np.int32)
pairs = 0
for nodes in bn:
    parts = [ci[rp[v]:rp[v + 1]] for v in nodes]
    u = np.unique(np.concatenate(parts + [np.asarray(nodes, dtype=np.int32)]))
    occ[u] += 1
    pairs += int(deg[u].sum())
nz = occ > 0
print("pairs (per-occurrence row fetches)", pairs)
print("distinct owners", int(nz.sum()), "sum deg over distinct", int(deg[nz].sum()))
print("fetches with groups of 8", int((deg[nz] * ((occ[nz] + 7) // 8)).sum()), "groups of 16", int((deg[nz] * ((occ[nz] + 15) // 16)).sum()))
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 150)):
    sel = (occ >= lo) & (occ <= hi)
    print(f"occ {lo}-{hi}: nodes {int(sel.sum())}, pair share {float((deg[sel] * occ[sel]).sum()) / pairs:.3f}, mean deg {float(deg[sel].mean()) if sel.any() else 0:.0f}")
