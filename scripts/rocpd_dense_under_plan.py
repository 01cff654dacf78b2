#!/usr/bin/env python3
"""How much does each plan kernel slow the dense step chain that runs beside it?  From a rocprofv3 rocpd trace of bench.py: every
dense step kernel instance is attributed to the plan kernel that was running at its start (or "nothing"), and the mean duration
per dense kernel is printed per concurrent plan kernel.   Usage: python scripts/rocpd_dense_under_plan.py results.db"""
import bisect
import sqlite3
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
dense = {"k_fwd_rows_v", "k_fwd_chunks", "k_loss_pos", "k_loss_pos_ck", "k_loss_rows", "k_bwd_flat", "k_grad_reduce"}
plan_names = {"k_expand", "k_gather1c", "k_combine1_reset", "k_seg_transpose", "k_tile_counts", "k_build_groups", "k_gather2_items",
              "k_gather2_combine"}
ev = [(short(n).split("<")[0], s, e) for n, s, e in rows]
plan = [x for x in ev if x[0] in plan_names]
starts = [x[1] for x in plan]
acc = {}
for name, s, e in ev:
    if name not in dense:
        continue
    i = bisect.bisect_right(starts, s) - 1
    beside = plan[i][0] if i >= 0 and plan[i][2] > s else "nothing"
    acc.setdefault(beside, {}).setdefault(name, []).append((e - s) / 1e3)
kn = sorted(dense)
print("beside".ljust(20) + "".join(k[2:].rjust(13) for k in kn) + "   sum of 5-kernel step (us)")
for beside in sorted(acc, key=lambda b: -sum(len(v) for v in acc[b].values())):
    d = acc[beside]
    means = {k: (np.mean(d[k]) if k in d else float("nan")) for k in kn}
    step = sum(means[k] for k in ("k_fwd_chunks", "k_loss_pos_ck", "k_loss_rows", "k_bwd_flat", "k_grad_reduce"))
    n = sum(len(v) for v in d.values())
    print(beside.ljust(20) + "".join(f"{means[k]:13.2f}" for k in kn) + f"   {step:8.1f}   ({n} launches)")
