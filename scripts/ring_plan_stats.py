"""Plan statistics of the LDS-ring product (Csr.ring_plan) against the LDS-panel plan on the published full-graph sizes (CPU only):
fill of the step slots, quads, per-phase skew between the walkers of a workgroup (the barrier waits for the longest), build time.

    python scripts/ring_plan_stats.py [t_finance|Amazon ...]
"""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
from ggad_amd import fullgraph_bench as FB                     # noqa: E402
from ggad_amd.fullgraph import Csr                             # noqa: E402
from ggad_amd.utils import normalize_adj                       # noqa: E402

for name in (sys.argv[1:] or ["t_finance", "Amazon"]):
    ds = FB.make_dataset(name)
    n = ds["n"]
    csr = Csr(normalize_adj(ds["adj"]) + sp.eye(n), "cpu")
    csr.value_factors()
    t = time.time()
    ring = csr.ring_plan(10)
    t_ring = time.time() - t
    t = time.time()
    pan = csr.panel_plan(10)
    t_pan = time.time() - t
    ws = ring["wave_sb"].numpy().reshape(-1, 2)
    # LDS cycles of a ds_read_b128: 4 service groups of 16 lanes; the rows of the lane groups (0, 3) / (1, 2) [and (4, 7) / (5, 6)] share two
    # of them and collide when their LDS rows have the same parity: + 2 cycles per half wave with a collision
    par = ring["idx"].numpy().view(np.uint16).reshape(-1, 2, 8, 8)[:-1] & 1          # [super-block][half][lane group][quad, step]
    lo = (par[:, :, 0] == par[:, :, 3]) | (par[:, :, 1] == par[:, :, 2])
    hi = (par[:, :, 4] == par[:, :, 7]) | (par[:, :, 5] == par[:, :, 6])
    extra = 2.0 * (lo.mean() + hi.mean())
    print(f"{name}: bank conflicts: {extra:.3f} extra LDS cycles per step (4 without) -> {ring['quads'] * 4 * (4 + extra) * 10 / 256 / 1e3:.0f} K LDS cycles per CU per product")
    nnz = csr.nnz - n
    print(f"{name}: nnz {nnz}  ring: fill {ring['fill']:.3f} quads {ring['quads']} (x32 slots: real fill {nnz / (ring['quads'] * 32):.3f}) blocks {ring['blocks']} "
          f"rounds {ring['rounds']} phases {ring['n_phases']} phase_skew {ring['phase_skew']:.3f} super-blocks/walker {ws[:, 1].mean():.1f} "
          f"(max {ws[:, 1].max()}, min {ws[:, 1].min()}) build {t_ring:.2f}s | panel: fill {pan['fill']:.3f} blocks {pan['blocks']} rounds {pan['rounds']} build {t_pan:.2f}s")
