#!/usr/bin/env python3
"""Which stream is the critical path of the overlapped trainer?  From a rocprofv3 rocpd trace of bench.py: for the steady-state
chunks, the busy time and the span of the plan kernels (everything that is not a dense step kernel) and of the dense step
kernels, per chunk (a chunk = one k_gather2_groups launch).   Usage: python scripts/rocpd_streams.py results.db"""
import sqlite3
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = cur.execute(sel).fetchall()
dense = {"k_project", "k_fwd_rows", "k_fwd_rows_v", "k_fwd_chunks", "k_loss_pos", "k_loss_pos_ck", "k_loss_rows", "k_bwd_flat",
         "k_grad_reduce", "k_train_chunk_persistent"}
ev = [(short(n).split("<")[0], s, e, q) for n, s, e, q in rows]
g = [i for i, x in enumerate(ev) if x[0] == "k_gather2_items"]
print("streams/queues seen:", sorted({x[3] for x in ev}))
for a, b in zip(g[5:-1], g[6:]):
    win = ev[a:b]
    t0, t1 = ev[a][1], ev[b][1]
    plan = [x for x in win if x[0] not in dense]
    den = [x for x in win if x[0] in dense]
    pb = sum(x[2] - x[1] for x in plan) / 1e6
    dbusy = sum(x[2] - x[1] for x in den) / 1e6
    top = {}
    for x in plan:
        top[x[0]] = top.get(x[0], 0.0) + (x[2] - x[1]) / 1e6
    tops = sorted(top.items(), key=lambda kv: -kv[1])[:7]
    print(f"chunk period {(t1 - t0) / 1e6:6.2f} ms | plan busy {pb:5.2f} ms, span {(max(x[2] for x in plan) - t0) / 1e6:5.2f} | "
          f"dense busy {dbusy:5.2f} ms over {len(den)} kernels | " + ", ".join(f"{k} {v:.2f}" for k, v in tops))
