#!/usr/bin/env python3
"""Time the validation sweep (test_sage: 1,736,598 test nodes in reference batches of 150) at DGraph size."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggad_amd import synth
from ggad_amd.dgraph import normalize_features
from ggad_amd.graph import DeviceGraph
from ggad_amd.graphsage import GCN, FeatureTable, GCNAggregator, GCNEncoder
from ggad_amd.sage_utils import score_nodes, test_sage
n = 3700550
entries = int(sys.argv[1]) if len(sys.argv) > 1 else 73105508
rp, ci = synth.make_graph_torch(n, entries, 72, "cuda:0", kind="powerlaw", max_degree=2000)
graph = DeviceGraph(rp, ci, "cuda:0")
feat = normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)
features = FeatureTable(torch.from_numpy(feat))
enc = GCNEncoder(features, 17, 64, graph, GCNAggregator(features, cuda=True), gcn=True, cuda=True)
model = GCN(2, enc)
cases = np.random.default_rng(0).permutation(n)[:1736598]
y = (np.random.default_rng(1).random(len(cases)) < 0.0042).astype(np.int64); y[:2] = (0, 1)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    p = score_nodes(model, cases, 150, device=True)
    torch.cuda.synchronize(); t1 = time.time()
    res = test_sage(cases, y, model, 150, 0.4, verbose=False)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"rep {rep}: score sweep {t1 - t0:.2f} s ({len(cases) / (t1 - t0) / 1e6:.2f} M nodes/s), test_sage incl. device metrics {t2 - t1:.2f} s")
