#!/usr/bin/env python3
"""Idle time between consecutive kernels of the dense step chain in a rocprofv3 rocpd trace.
Usage: python scripts/rocpd_gaps.py results.db"""
import sqlite3, sys
import numpy as np
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
chain = ["k_project", "k_fwd_rows", "k_loss_pos", "k_loss_rows", "k_bwd_flat", "k_grad_reduce"]
ev = [(short(n).split("<")[0], s, e) for n, s, e in rows]
ev = [x for x in ev if x[0] in chain]
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(ev[:-1], ev[1:]):
    gaps.setdefault(n0 + "->" + n1, []).append((s1 - e0) / 1e3)
for k, v in gaps.items():
    v = np.array(v)
    if len(v) > 20:
        print(f"{k:28s} n={len(v):5d} median {np.median(v):6.2f} us  mean {v.mean():6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():8.1f}")
