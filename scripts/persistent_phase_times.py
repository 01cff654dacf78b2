#!/usr/bin/env python3
"""Per-phase time of the persistent chunk kernel (workgroup 0's clock between grid barriers), DGraph-size synthetic graph."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ggad_amd import synth  # noqa: E402
from ggad_amd.dgraph import normalize_features  # noqa: E402
from ggad_amd.graph import DeviceGraph  # noqa: E402
from ggad_amd.minibatch import BatchChunk, MiniBatchEngine  # noqa: E402

dev = "cuda:0"
n, entries = 3700550, int(os.environ.get("ENTRIES", 73105508))
rowptr, col = synth.make_graph_torch(n, entries, 72, dev, kind="powerlaw", max_degree=2000)
graph = DeviceGraph(rowptr, col, dev)
feat = normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)
table = torch.zeros(n, 32, device=dev)
table[:, :17] = torch.from_numpy(feat).to(dev)
rng = np.random.default_rng(0)
batches = [rng.choice(n, size=200, replace=False) for _ in range(150)]
labels = [np.r_[np.zeros(150, dtype=np.int64), np.ones(50, dtype=np.int64)] for _ in range(150)]
ch = BatchChunk(graph, table, 64, 150, 150 * 200, 1 << 20, train=True, feat_dim=17, hop2="ldsw")
ch.build(batches, labels)
eng = MiniBatchEngine(17, 64, dev, chain=3)
eng.persistent_wgs = int(os.environ.get("WGS", 64))
torch.manual_seed(0)
eng.load_params(torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)),
                torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
for it in range(3):
    eng.train_chunk(ch)
    torch.cuda.synchronize()
    tail = eng.ps_ws[-64:].view(torch.int64)          # not the exact offset: find the barrier word
# the barrier block sits at a fixed offset: recompute it
from ggad_amd import _lib  # noqa: E402
lib = _lib.load()
r = np.diff(ch.ent_ptr_host[:ch.n_rows + 1])
per_row = (r + 15) // 16
bp = ch.batch_ptr_host[:151]
max_chunks = int(np.add.reduceat(per_row, bp[:-1].astype(np.int64)).max())
need = int(lib.ggad_mb_persistent_ws_elems(max_chunks, eng.persistent_wgs))
blk = eng.ps_ws[need - 64:need].view(torch.int64).cpu().numpy()
ticks = blk[2:8].astype(np.float64)
print("chunks per batch (max)", max_chunks, "workgroups", eng.persistent_wgs)
print("per step, us:", " ".join(f"P{k + 1} {t / 150 / 100:.2f}" for k, t in enumerate(ticks)), f"| total {ticks.sum() / 150 / 100:.2f}")
