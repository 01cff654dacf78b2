#!/usr/bin/env python3
"""Every kernel of the LAST chunk of a traced `bench.py --steps 20` run (from its k_expand to the end of the chunk kernel), with start and
end relative to the chunk's first kernel -- kernels of several streams overlap (staged plans).  Usage: rocpd_timeline_all.py results.db"""
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from rocpd_stats import short  # noqa: E402

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = [(short(n).split("<")[0], s, e) for n, s, e in db.execute(f"select {name_col}, start, end from kernels order by start")]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
idx = [i for i, r in enumerate(rows) if r[0] == "k_expand"]
a = idx[which]
seg = rows[a:(idx[which + 1] if which + 1 < 0 and which + 1 != 0 else len(rows))]
t_end = max(r[2] for r in seg if r[0] in ("k_grad_reduce", "k_train_chunk_xcd"))
seg = [r for r in seg if r[1] <= t_end]
t0 = seg[0][1]
for n, s, e in seg:
    print(f"{n:24s} start {(s - t0) / 1e3:9.1f}  end {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f} us")
print(f"chunk span {(max(e for _, _, e in seg) - t0) / 1e3:.1f} us, kernels {len(seg)}")
