"""Host-side structures of the LDS-ring product (`Csr.ring_plan` + csrc/spmm_ring_build.cpp, consumed by k_spmm_ring in
fullgraph.hip): the per-walker quad list is replayed in numpy exactly as the kernel walks it -- phase by phase, with the ring
content a loader wave would have staged -- and compared with the sparse product (reference op: torch.bmm(adj, .), model.py:31)."""
import numpy as np
import scipy.sparse as sp

from ggad_amd import _lib
from ggad_amd.fullgraph import Csr
from test_fullgraph_panel_cpu import WIDE, _normalized


def _replay(plan, x, n_rows, n_src):
    lib = _lib.load()
    RS, S, V = int(lib.ggad_spmm_ring_slot_rows()), int(lib.ggad_spmm_ring_slots()), int(lib.ggad_spmm_ring_window())
    NW, KR = int(plan["walkers"]), int(lib.ggad_spmm_ring_rounds())          # 15, or 13 with three loader waves (round 6)
    assert NW in (int(lib.ggad_spmm_ring_walkers()), int(lib.ggad_spmm_ring_walkers_subset()))
    wave_sb = plan["wave_sb"].numpy().reshape(-1, 2)
    idx = plan["idx"].numpy().view(np.uint16).reshape(-1, 2, 8, 2, 4)         # [super-block][half][lane group][quad in half][step]
    ctl = plan["ctl"].numpy().view(np.uint8)
    row_tab = plan["row_tab"].numpy().reshape(-1, KR, 8)
    NP = plan["n_phases"]
    assert NP == (n_src + RS - 1) // RS and len(wave_sb) == plan["blocks"] * NW
    xs = x if plan["cs"] is None else x * plan["cs"].numpy()[:, None]
    out = np.zeros((n_rows, x.shape[1]), dtype=np.float64)
    seen = pad = 0
    zero0 = S * RS
    for gw, (sb0, nsb) in enumerate(wave_sb.tolist()):
        phase = 0
        for q in range(4 * nsb):
            sb, qi = sb0 + q // 4, q % 4
            c = int(ctl[4 * sb + qi])
            k = (c & 0x3f) // 4
            assert (c & 3) == 0 and k < KR
            steps = idx[sb, qi // 2, :, qi % 2, :]                            # (8 lane groups, 4 steps)
            for g in range(8):
                r = int(row_tab[gw, k, g])
                r = r if r < 0 else r & ~WIDE
                for lr in steps[g].tolist():
                    assert lr <= zero0 + 1
                    if lr >= zero0:
                        pad += 1
                        continue
                    assert phase < NP, "an entry after the last phase"
                    buf, off = divmod(lr, RS)
                    # the slot in buffer `buf` during `phase`: the one of the window phase .. phase + V - 1 congruent to buf
                    slot = phase + (buf - phase) % S
                    assert slot < phase + V, "entry outside the resident window"
                    src = slot * RS + off
                    assert src < n_src and r >= 0
                    out[r] += xs[src]
                    seen += 1
            if c & 0x40:
                phase += 1
        assert phase == NP, "every walker flags every phase exactly once"
    if plan["rs"] is not None:
        out *= plan["rs"].numpy()[:, None]
    if plan["diag"] is not None:
        out += plan["diag"].numpy()[:, None] * x
    return out, seen, pad


import pytest


@pytest.mark.parametrize("subset_steps", ["0", "1e9"])
def test_ring_plan_replays_to_the_sparse_product(subset_steps, monkeypatch):
    """Both dealings of the plan: 15 walkers + one loader wave (GGAD_RING_SUBSET_STEPS=0: never the other) and 13 walkers + three
    loader waves (every plan whose walkers have fewer steps per phase than the limit: the default takes all)."""
    monkeypatch.setenv("GGAD_RING_SUBSET_STEPS", subset_steps)
    lib = _lib.load()
    RS, S = int(lib.ggad_spmm_ring_slot_rows()), int(lib.ggad_spmm_ring_slots())
    n = 9 * RS + 101                                                       # 10 phases (two trips round the ring), the last slot partial
    for inside in (False, True):
        m = _normalized(n, 0.04, 3, inside)
        csr = Csr(m, "cpu")
        plan = csr.ring_plan(3)
        assert plan is not None and plan["n_phases"] == 10
        assert plan["walkers"] == (int(lib.ggad_spmm_ring_walkers()) if subset_steps == "0" else int(lib.ggad_spmm_ring_walkers_subset()))
        x = np.random.default_rng(0).standard_normal((n, 5))
        out, seen, pad = _replay(plan, x, n, n)
        assert seen == m.nnz - n                                           # every off-diagonal entry exactly once
        np.testing.assert_allclose(out, m @ x, rtol=2e-6, atol=1e-6)
        assert seen + pad == 32 * 4 * int(plan["wave_sb"].numpy().reshape(-1, 2)[:, 1].sum())      # 4 quads of 32 slots per super-block
        assert plan["quads"] * 32 <= seen + pad < (plan["quads"] + 4 * len(plan["wave_sb"])) * 32 and plan["fill"] > 0.5
        rt = plan["row_tab"].numpy().reshape(-1, 8)
        wide = rt[(rt[:, 0] >= 0) & ((rt[:, 0] & WIDE) != 0)]
        assert len(wide) >= 8 and (wide == wide[:, :1]).all()
        normal = rt[(rt[:, 0] & WIDE) == 0].reshape(-1)
        assert sorted(normal[normal >= 0].tolist() + (wide[:, 0] & ~WIDE).tolist()) == list(range(n))
        st = plan["idx"].numpy().view(np.uint16).reshape(-1, 8, 8)         # rows of a bank-sharing pair alternate parities
        same = (st & 1) == (st[:, [3, 2, 1, 0, 7, 6, 5, 4], :] & 1)
        assert same.mean() < 0.25
        wg = plan["wg"].numpy().reshape(-1, 2)
        real = wg[wg[:, 0] >= 0]
        assert len(real) == 3 * plan["blocks"] and len({tuple(t) for t in real.tolist()}) == len(real)


def test_ring_plan_pattern_matrix_row_subset_and_short_operand(monkeypatch):
    n = 1500
    m = _normalized(n, 0.08, 5, False)
    pat = (m != 0).astype(np.float64).tocsr()
    csr = Csr(pat, "cpu")
    x = np.random.default_rng(1).standard_normal((n, 3))
    for xcd in ("slice", "block"):
        monkeypatch.setenv("GGAD_RING_XCD", xcd)
        plan = csr.ring_plan(2)
        assert plan is not None and plan["rs"] is None and plan["diag"] is None
        out, seen, _ = _replay(plan, x, n, n)
        assert seen == pat.nnz
        np.testing.assert_allclose(out, pat @ x, rtol=1e-9, atol=1e-9)
        wg = plan["wg"].numpy().reshape(-1, 2)
        assert (wg[:, 0] >= 0).sum() == 2 * plan["blocks"]
    rows = np.array([7, 1499, 3, 3, 640, 0] + list(range(100, 160)))       # row subset: output row i = matrix row rows[i]
    sub = csr.ring_plan(2, rows, {})
    assert sub is not None
    out, seen, _ = _replay(sub, x, len(rows), n)
    assert seen == pat[rows].nnz
    np.testing.assert_allclose(out, pat[rows] @ x, rtol=1e-9, atol=1e-9)
    assert Csr(m, "cpu").ring_plan(2, rows, {}) is None                     # separate diagonal: no subset plan
    small = sp.random(100, 100, density=0.5, random_state=np.random.default_rng(4), format="csr")      # operand shorter than one slot
    small.data[:] = 1.0
    c2 = Csr(small, "cpu")
    p2 = c2.ring_plan(1)
    x2 = np.random.default_rng(2).standard_normal((100, 2))
    out, seen, _ = _replay(p2, x2, 100, 100)
    assert p2["n_phases"] == 1 and seen == small.nnz
    np.testing.assert_allclose(out, small @ x2, rtol=1e-9, atol=1e-9)
