"""Host-side structures of the LDS-panel product (`Csr.panel_plan`, consumed by k_spmm_panel in fullgraph.hip): the entry
stream / directory / row table are replayed in numpy exactly as the kernel walks them and compared with the sparse product."""
import numpy as np
import pytest
import scipy.sparse as sp

from ggad_amd import _lib
from ggad_amd.fullgraph import Csr


def _normalized(n, density, seed, self_loops_inside):
    rng = np.random.default_rng(seed)
    a = sp.random(n, n, density=density, random_state=rng, format="csr")
    a.data[:] = 1.0
    a = a.tolil()
    for hub in range(0, n, max(1, n // 12)):                              # a dozen hub rows: they become WIDE rounds of the plan
        a[hub, rng.permutation(n)[: n // 2]] = 1.0
    a = ((a.tocsr() + a.tocsr().T) > 0).astype(np.float64).tolil()
    a.setdiag(0)
    a = a.tocsr()
    a.eliminate_zeros()
    if self_loops_inside:
        a = a + sp.eye(n)
    d = np.asarray(a.sum(1)).reshape(-1)
    with np.errstate(divide="ignore"):
        r = np.power(d, -0.5)
    r[np.isinf(r)] = 0.0
    m = sp.diags(r) @ a @ sp.diags(r)
    return (m if self_loops_inside else m + sp.eye(n)).tocsr()


WIDE = 0x40000000


def _replay(plan, x, n_rows, R, NW, KR):
    """out = rs * (sum over the stream of cs-scaled rows) + diag * x, in the kernel's order."""
    dirv = plan["dir"].numpy().view(np.uint32).reshape(-1, 8)
    stream = plan["stream"].numpy().view(np.uint16)
    row_tab = plan["row_tab"].numpy().reshape(-1, 8)
    nc = plan["n_chunks"]
    xs = x if plan["cs"] is None else x * plan["cs"].numpy()[:, None]
    out = np.zeros((n_rows, x.shape[1]), dtype=np.float64)
    seen = 0
    for wv in range(dirv.shape[0] // nc):
        for c in range(nc):
            d = dirv[wv * nc + c]
            off = int(d[0])
            for k in range(KR):
                nh = int((d[1 + (k >> 1)] >> (16 * (k & 1))) & 0xffff)       # quads: whole octs, then half of the last one
                nq = (nh + 1) // 2
                for q in range(nq):
                    octv = stream[(off + q) * 64:(off + q + 1) * 64].reshape(8, 8)
                    steps = 4 if (q == nq - 1 and nh % 2 == 1) else 8
                    assert (octv[:, steps:] >= R).all()                      # nothing but padding in a half that is not walked
                    for g in range(8):
                        r = row_tab[wv * KR + k, g]
                        r = r if r < 0 else r & ~WIDE                        # a wide round: the same row in all 8 lane groups
                        for o in octv[g, :steps]:
                            assert o <= R + 1
                            if o < R:
                                assert r >= 0
                                out[r] += xs[c * R + o]
                                seen += 1
                off += nq
    if plan["rs"] is not None:
        out *= plan["rs"].numpy()[:, None]
    if plan["diag"] is not None:
        out += plan["diag"].numpy()[:, None] * x
    return out, seen


@pytest.mark.parametrize("inside", [False, True])
def test_panel_plan_replays_to_the_sparse_product(inside):
    lib = _lib.load()
    R, NW, KR = int(lib.ggad_spmm_panel_rows()), int(lib.ggad_spmm_panel_waves()), int(lib.ggad_spmm_panel_rounds())
    n = 2 * R + 452                                                       # 3 panels, the last one partial
    m = _normalized(n, 0.05, 3, inside)
    csr = Csr(m, "cpu")
    rs, cs, diag = csr.value_factors()
    assert rs is not None and diag is not None
    plan = csr.panel_plan(3)
    assert plan is not None and plan["n_chunks"] == 3
    x = np.random.default_rng(0).standard_normal((n, 5))
    out, seen = _replay(plan, x, n, R, NW, KR)
    assert seen == m.nnz - n                                              # every off-diagonal entry exactly once
    np.testing.assert_allclose(out, m @ x, rtol=2e-6, atol=1e-6)
    rt = plan["row_tab"].numpy().reshape(-1, 8)
    wide = rt[(rt[:, 0] >= 0) & ((rt[:, 0] & WIDE) != 0)]
    assert len(wide) >= 8 and (wide == wide[:, :1]).all()                # the hub rows: one row in all 8 slots of its round
    normal = rt[(rt[:, 0] & WIDE) == 0].reshape(-1)
    assert sorted(normal[normal >= 0].tolist() + (wide[:, 0] & ~WIDE).tolist()) == list(range(n))   # every output row once
    st = plan["stream"].numpy().view(np.uint16).reshape(-1, 8, 8)      # rows of a bank-sharing pair alternate parities
    same = (st & 1) == (st[:, [3, 2, 1, 0, 7, 6, 5, 4], :] & 1)
    assert same.mean() < 0.25
    wg = plan["wg"].numpy().reshape(-1, 2)
    real = wg[wg[:, 0] >= 0]
    assert len(real) == 3 * plan["blocks"] and len({tuple(t) for t in real.tolist()}) == len(real)


def test_pattern_matrix_keeps_its_diagonal_in_the_stream_and_unfactorable_values_are_refused():
    n = 1500
    m = _normalized(n, 0.08, 5, False)
    pat = (m != 0).astype(np.float64).tocsr()
    csr = Csr(pat, "cpu")
    assert csr.value_factors() == (None, None, None)
    plan = csr.panel_plan(2)
    assert plan is not None and plan["rs"] is None and plan["diag"] is None
    lib = _lib.load()
    x = np.random.default_rng(1).standard_normal((n, 3))
    out, seen = _replay(plan, x, n, int(lib.ggad_spmm_panel_rows()), int(lib.ggad_spmm_panel_waves()), int(lib.ggad_spmm_panel_rounds()))
    assert seen == pat.nnz
    np.testing.assert_allclose(out, pat @ x, rtol=1e-9, atol=1e-9)
    rows = np.array([7, 1499, 3, 3, 640, 0] + list(range(100, 160)))         # row subset: output row i = matrix row rows[i]
    sub = csr.panel_plan(2, rows, {})
    assert sub is not None
    out, seen = _replay(sub, x, len(rows), int(lib.ggad_spmm_panel_rows()), int(lib.ggad_spmm_panel_waves()), int(lib.ggad_spmm_panel_rounds()))
    assert seen == pat[rows].nnz
    np.testing.assert_allclose(out, pat[rows] @ x, rtol=1e-9, atol=1e-9)
    assert Csr(m, "cpu").panel_plan(2, rows, {}) is None                     # separate diagonal: no subset plan
    w = m.copy()
    w.data = w.data * np.random.default_rng(2).uniform(0.5, 1.5, size=w.nnz)
    bad = Csr(w, "cpu")
    assert bad.value_factors() is False and bad.panel_plan(2) is None
