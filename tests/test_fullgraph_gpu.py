"""GPU parity tests of the full-graph GGAD path (CSR SpMM / MFMA GEMM / affinity kernels through the C-ABI and the
drop-in `Model` / `GCN` classes) against golden vectors captured from the imported reference (dense N x N path)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
from ggad_amd import synth

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ggad_amd import fullgraph as FG
    from ggad_amd import utils as U
    from ggad_amd.model import GCN, Model

DEV = "cuda:0"


def _adj(g):
    import scipy.sparse as sp
    n = int(g["n"])
    a = synth.csr_to_scipy(g["rowptr"], g["col"], n)
    adj_norm = U.normalize_adj(a) + sp.eye(n)            # run.py:98,101
    raw = a + sp.eye(n)                                  # run.py:100
    return FG.FullGraphAdj(adj_norm, raw, DEV)


@pytest.mark.parametrize("shape", [(70, 50, 33), (300, 300, 7535), (1, 75, 1000), (130, 1, 40), (64, 64, 16), (257, 129, 3000),
                                   (132, 68, 36), (131, 67, 745), (7535, 300, 745), (66, 302, 130), (500, 300, 10), (257, 64, 3),
                                   (300, 10, 5000), (10984, 300, 300), (300, 300, 10984), (1830, 300, 300), (300, 64, 10984),
                                   (260, 132, 64), (64, 64, 32), (68, 300, 36), (4096, 512, 1024),
                                   (5000, 260, 320), (2048, 68, 128), (3000, 1024, 256), (2500, 300, 132), (4100, 96, 300), (39357, 300, 300),
                                   (4111, 196, 252), (4500, 198, 300), (10984, 512, 256), (4200, 128, 244), (6000, 320, 320), (4097, 176, 316)])
@pytest.mark.parametrize("dma", ["1", "0"])
def test_gemm_f32_all_layouts(shape, dma, monkeypatch):
    """C = op(A) op(B) (+ bias, ReLU) on the exact-f32 matrix cores for every layout and for shapes with ragged edges, split-K
    (weight gradients: K = number of nodes), the tall products with a small second operand (M >= 4096, 244 <= K <= 320, N a sum of
    5- and 4-tile slabs: k_gemm_slab, round 5 -- 80-column slabs of op(B) resident in LDS, both memory layouts of B, 16-byte and scalar
    stores; M >= 2048, 128 <= K <= 320 otherwise: k_gemm_bres) and the aligned K = 300 products of the path, which take the LDS-DMA kernel (k_gemm_dma;
    dma = "0" is decided when the library first reads GGAD_GEMM_DMA, so that run only re-checks the register-staged kernel when it is
    the first in the process), against numpy in float64."""
    monkeypatch.setenv("GGAD_GEMM_DMA", dma)
    m, n, k = shape
    rng = np.random.default_rng(m + n + k)
    for ta in (False, True):
        for tb in (False, True):
            a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
            b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)   # asymmetric, non-square: catches transposes
            bias = rng.standard_normal(n).astype(np.float32)
            ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
            got = FG.gemm(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), ta, tb).cpu().numpy()
            scale = np.abs(ref).max() + 1.0
            assert np.abs(got - ref).max() / scale < 2e-6, (shape, ta, tb)
            got2 = FG.gemm(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV), ta, tb, bias=torch.from_numpy(bias).to(DEV),
                           relu=True).cpu().numpy()
            assert np.abs(got2 - np.maximum(ref + bias, 0)).max() / scale < 2e-6


@pytest.mark.parametrize("ldc", [300, 304, 320, 301])
def test_gemm_slab_strided_output_and_switch(ldc, monkeypatch):
    """k_gemm_slab writes into a row-strided destination (`padded_rows`: what GcnLayerFn hands it) with 16-byte stores when the stride
    allows and scalar ones otherwise, never outside the N columns of a row; and the product equals the round-4 kernels' to round-off."""
    rng = np.random.default_rng(ldc)
    m, n, k = 9000, 300, 300
    a = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)).to(DEV)
    for tb in (True, False):
        b = torch.from_numpy(rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)).to(DEV)
        bias = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
        buf = torch.full((m, ldc), -7.0, dtype=torch.float32, device=DEV)
        out = buf[:, :n]
        got = FG.gemm(a, b, False, tb, bias=bias, relu=True, out=out)
        assert got.data_ptr() == buf.data_ptr()
        ref = torch.relu(a.double() @ (b.double().T if tb else b.double()) + bias.double())
        assert ((got.double() - ref).abs().max() / (ref.abs().max() + 1.0)).item() < 2e-6
        if ldc > n:
            assert bool((buf[:, n:] == -7.0).all())                     # the padding of every row is untouched


@pytest.mark.parametrize("name", ["fullgraph_reddit_like.npz", "fullgraph_amazon_like.npz"])
def test_preprocessing_matches_reference(name):
    g = load_golden(name)
    np.testing.assert_allclose(U.preprocess_features(g["feat_raw"]).astype(np.float32), g["features"], atol=1e-7, rtol=0)
    fa = _adj(g)
    import scipy.sparse as sp
    n = int(g["n"])
    ref = sp.coo_matrix((g["adjn_val"], (g["adjn_row"], g["adjn_col"])), shape=(n, n)).tocsr()
    ref.sum_duplicates(); ref.sort_indices()
    assert np.array_equal(fa.A.rowptr.cpu().numpy(), ref.indptr) and np.array_equal(fa.A.col.cpu().numpy(), ref.indices)
    np.testing.assert_array_equal(fa.A.val.cpu().numpy(), ref.data.astype(np.float32))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["fullgraph_reddit_like.npz", "fullgraph_amazon_like.npz"])
def test_model_forward_loss_backward_trajectory(name, fused, monkeypatch):
    """The reference's own tensors, losses, gradients and weights (tests/golden/make_golden.py) -- with the head of the forward as
    one autograd node (`GgadHeadFn`, the default) and the loss block's row-local parts in one launch each way (round 6), and op by op
    with the round-5 launch sequence of the loss block."""
    monkeypatch.setattr(FG, "_LOSS_FUSED", bool(fused))
    g = load_golden(name)
    fa = _adj(g)
    f, h = int(g["f"]), int(g["n_h"])
    torch.manual_seed(123)
    model = Model(f, h, "prelu", 1, "avg")
    sd = {k[len("init."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")}
    assert sorted(sd.keys()) == sorted(model.state_dict().keys())        # same parameter names as the reference
    model.load_state_dict(sd)
    model.to(DEV)
    model.fused_head = fused
    opt = FG.FlatAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    feats = torch.from_numpy(g["features"])[None].to(DEV)
    abn, nrm = g["abn_idx"].tolist(), g["normal_idx"].tolist()
    args = types.SimpleNamespace(mean=float(g["mean"]), var=float(g["var"]))
    ls = fa.loss_structs(nrm, abn)
    assert (fa.head_structs(nrm, abn) is not None) and fa.head_structs(nrm + nrm[:1], abn) is None      # duplicate-free lists only
    for step in range(len(g["losses"])):
        model.train()
        opt.zero_grad()
        torch.manual_seed(1000 + step)
        emb, emb_combine, logits, emb_con, emb_abnormal = model(feats, fa, abn, nrm, True, args)
        assert emb.shape == (1, int(g["n"]), h) and logits.shape == (1, len(nrm) + len(abn), 1)
        total, l_margin, l_bce, l_rec = FG.GgadLossFn.apply(emb[0], logits[0, :, 0], emb_con, emb_abnormal[0], fa, ls, 0.7)
        total.backward()
        np.testing.assert_allclose([total.item(), l_margin.item(), l_bce.item(), l_rec.item()], g["losses"][step], atol=1e-5)
        if step == 0:
            np.testing.assert_allclose(emb[0].detach().cpu().numpy(), g["emb"], atol=3e-6)
            np.testing.assert_allclose(emb_combine[0].detach().cpu().numpy(), g["emb_combine"], atol=3e-6)
            np.testing.assert_allclose(logits[0, :, 0].detach().cpu().numpy(), g["logits"], atol=3e-6)
            np.testing.assert_allclose(emb_con.detach().cpu().numpy(), g["emb_con"], atol=3e-6)
            np.testing.assert_allclose(emb_abnormal[0].detach().cpu().numpy(), g["emb_abnormal"], atol=3e-6)
            for k, p in model.named_parameters():
                if ("grad." + k) in g:
                    np.testing.assert_allclose(p.grad.cpu().numpy(), g["grad." + k], atol=4e-6, rtol=2e-4, err_msg=k)
                else:
                    assert p.grad is None, k          # gcn3 / fc5 / fc6 / disc never receive a gradient
        opt.step()
        if step == 0:
            for k, v in model.state_dict().items():
                np.testing.assert_allclose(v.cpu().numpy(), g["step1." + k], atol=3e-6, err_msg=k)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["final." + k], atol=3e-5, err_msg=k)
    model.eval()
    torch.manual_seed(5000)
    with torch.no_grad():
        _, comb, le, con, _ = model(feats, fa, abn, nrm, False, args)
    assert comb is None and con is None
    le = le[0, :, 0].cpu().numpy()
    np.testing.assert_allclose(le, g["eval_logits"], atol=5e-5)
    from sklearn.metrics import average_precision_score, roc_auc_score
    yt = g["ano"][g["idx_test"]]
    assert abs(roc_auc_score(yt, le[g["idx_test"]]) - float(g["eval_auc"])) < 1e-4
    assert abs(average_precision_score(yt, le[g["idx_test"]]) - float(g["eval_ap"])) < 1e-4


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("n,nn_,na,h", [(3000, 700, 90, 300), (500, 61, 7, 64), (12000, 1592, 238, 300)])
def test_fused_loss_block_equals_the_launch_sequence(n, nn_, na, h, overlap, monkeypatch):
    """Round 6: `ggad_full_loss_fused_f32` / `ggad_full_loss_bwd_fused_f32` / `ggad_rownorm_bwd_add_f32` (run.py:165-210 in three + three
    launches) against the round-5 sequence of six + six on the same inputs: the four loss values (the last workgroup sums 256-wide, the
    old kernel 1024-wide: 1e-6), the affinity and every gradient (same operations per element: 1e-6 of the scale); with `overlap` a node
    sits in BOTH index lists (both c_j S_j terms reach its row, normal first); repeated calls are bit-identical (the ticket returns to 0)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(n + na)
    a = sp.random(n, n, density=8.0 / n, random_state=3, format="csr", dtype=np.float64)
    a.data[:] = 1.0
    a = ((a + a.T) > 0).astype(np.float64)
    fa = FG.FullGraphAdj(U.normalize_adj(a) + sp.eye(n), a + sp.eye(n), DEV)
    perm = rng.permutation(n)
    nrm, abn = perm[:nn_].tolist(), perm[nn_:nn_ + na].tolist()
    if overlap:
        abn[0] = nrm[3]
    ls = fa.loss_structs(nrm, abn)
    assert ls["seg_unique"] and ls["distinct"] == (not overlap)
    emb0 = torch.randn(n, h, device=DEV)
    emb0[5] = 0.0                                              # a zero row: 1 / |e| = inf -> 0 (run.py:179)
    logits0 = torch.randn(nn_ + na, device=DEV)
    con0, abn0 = torch.randn(na, h, device=DEV), torch.randn(na, h, device=DEV)

    def run(flag):
        monkeypatch.setattr(FG, "_LOSS_FUSED", flag)
        ins = [t.clone().requires_grad_() for t in (emb0, logits0, con0, abn0)]
        out = FG.GgadLossFn.apply(ins[0], ins[1], ins[2], ins[3], fa, ls, 0.7)
        out[0].backward(gradient=torch.full((), 1.7, device=DEV))
        return [o.item() for o in out], [t.grad.clone() for t in ins]
    l_new, g_new = run(True)
    l_old, g_old = run(False)
    np.testing.assert_allclose(l_new, l_old, rtol=2e-6, atol=2e-6)
    for a_, b_ in zip(g_new, g_old):
        scale = float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= 2e-6 * max(scale, 1e-3)
    l_again, g_again = run(True)
    assert l_again == l_new and all(torch.equal(x, y) for x, y in zip(g_again, g_new))


@pytest.mark.parametrize("tickets", [True, False])
def test_flat_adam_against_torch_adam_with_and_without_ticket_bump(tickets, monkeypatch):
    """`ggad_adam_multi_f32` (torch.optim.Adam.step of run.py:118,213): 1,024 elements per workgroup; with `tickets` (round 6) the last
    workgroup of every tensor advances its step counter inside the launch, without them a trailing launch does.  Six steps on tensors of
    awkward sizes (1 element, 1,023, 1,025, 300 x 300, a tensor that never gets a gradient) against torch's own Adam; counters = steps."""
    monkeypatch.setattr(FG, "_ADAM_TICKETS", tickets)
    torch.manual_seed(3)
    shapes = [(1,), (1023,), (1025,), (300, 300), (7, 64), (300,), (2048,)]
    ps = [torch.nn.Parameter(torch.randn(*s_, device=DEV)) for s_ in shapes] + [torch.nn.Parameter(torch.randn(5, device=DEV))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, topt = FG.FlatAdam(ps, lr=1e-3, weight_decay=0.007), torch.optim.Adam(ref, lr=1e-3, weight_decay=0.007)
    for step in range(6):
        opt.zero_grad(); topt.zero_grad()
        for p, r in zip(ps[:-1], ref[:-1]):
            gval = torch.randn_like(p)
            p.grad, r.grad = gval.clone(), gval.clone()
        opt.step(); topt.step()
    for p, r in zip(ps, ref):
        np.testing.assert_allclose(p.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    assert all(int(opt.state[p][2].item()) == 6 for p in ps[:-1]) and ps[-1] not in opt.state
    if tickets:
        assert int(opt._tickets.abs().sum().item()) == 0          # every ticket word is back at zero


def test_gcn_layer_accepts_dense_adjacency_like_the_reference(g_full_reddit):
    g = g_full_reddit
    import scipy.sparse as sp
    n, f, h = int(g["n"]), int(g["f"]), int(g["n_h"])
    a = synth.csr_to_scipy(g["rowptr"], g["col"], n)
    dense = torch.FloatTensor(np.asarray((U.normalize_adj(a) + sp.eye(n)).todense()))[None]     # run.py:101-108
    layer = GCN(f, h, "prelu").to(DEV)
    with torch.no_grad():
        layer.fc.weight.copy_(torch.from_numpy(g["init.gcn1.fc.weight"]))
        layer.bias.copy_(torch.from_numpy(g["init.gcn1.bias"]))
    x = torch.from_numpy(g["features"])[None]
    out = layer(x.to(DEV), dense)
    ref = torch.nn.functional.prelu(torch.bmm(dense, x @ layer.fc.weight.detach().cpu().t()) + layer.bias.detach().cpu(),
                                    layer.act.weight.detach().cpu())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), atol=3e-6)


def test_run_py_captured_epoch_equals_eager():
    """run.py replays a captured hipGraph of the training epoch after two eager epochs; losses, AUROC and AP printed
    along the way must be exactly those of the all-eager run (same kernels, same order, noise from the same CPU draws)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--no_graph"]):
        r = subprocess.run([sys.executable, os.path.join(root, "run.py"), "--dataset", "reddit", "--synthetic", "--num_epoch", "13"] + extra,
                           capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        keep = [l for l in r.stdout.splitlines() if l.startswith("Epoch:") or l.startswith("Testing")]
        assert len(keep) >= 7 * 4 + 4
        outs.append(keep)
    assert outs[0] == outs[1]


def test_ocgnn_model_loss_backward_trajectory():
    """Full-graph OCGNN comparison model (`model_ocgnn.py` + the loss / step of `ocgnn.py`) against the imported reference."""
    from ggad_amd.model_ocgnn import Model as OcModel, ocgnn_loss
    g = load_golden("fullgraph_ocgnn.npz")
    fa = _adj(g)
    f, h = int(g["f"]), int(g["n_h"])
    torch.manual_seed(int(g["seed"]))
    model = OcModel(f, h, "prelu", 1, "avg")
    sd = {k[len("init."):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("init.")}
    assert sorted(sd.keys()) == sorted(model.state_dict().keys())        # same parameter names as the reference
    for k, v in model.state_dict().items():                              # same seed -> same initial weights
        np.testing.assert_array_equal(v.numpy(), sd[k].numpy(), err_msg=k)
    model.to(DEV)
    opt = FG.FlatAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
    feats = torch.from_numpy(g["features"])[None].to(DEV)
    nrm = torch.from_numpy(g["normal_idx"]).to(DEV)
    for step in range(len(g["losses"])):
        model.train()
        opt.zero_grad()
        emb = model(feats, fa)
        assert emb.shape == (1, int(g["n"]), h)
        loss, score = ocgnn_loss(emb[0], nrm)
        loss.backward()
        assert abs(loss.item() - g["losses"][step]) < 1e-5
        if step == 0:
            np.testing.assert_allclose(emb[0].detach().cpu().numpy(), g["emb"], atol=3e-6)
            np.testing.assert_allclose(score.cpu().numpy(), g["score"], atol=1e-5)
            for k, p in model.named_parameters():
                if ("grad." + k) in g:
                    np.testing.assert_allclose(p.grad.cpu().numpy(), g["grad." + k], atol=4e-6, rtol=2e-4, err_msg=k)
                else:
                    assert p.grad is None, k                              # the discriminator is never used
        opt.step()
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["final." + k], atol=3e-5, err_msg=k)
    model.eval()
    with torch.no_grad():
        _, sc = ocgnn_loss(model(feats, fa)[0])
    sc = sc.cpu().numpy()
    np.testing.assert_allclose(sc, g["eval_score"], atol=5e-5)
    from sklearn.metrics import average_precision_score, roc_auc_score
    yt = g["ano"][g["idx_test"]]
    assert abs(roc_auc_score(yt, sc[g["idx_test"]]) - float(g["eval_auc"])) < 1e-4
    assert abs(average_precision_score(yt, sc[g["idx_test"]]) - float(g["eval_ap"])) < 1e-4


def test_ocgnn_loss_kernel_centre_radius_and_duplicates_free_rows():
    from ggad_amd.model_ocgnn import ocgnn_loss
    rng = np.random.default_rng(5)
    emb = torch.from_numpy(rng.standard_normal((500, 77)).astype(np.float32))
    idx = torch.from_numpy(rng.permutation(500)[:123].astype(np.int64))
    c = torch.from_numpy(rng.standard_normal(77).astype(np.float32) * 0.3)
    for r, beta, cc in ((0.0, 0.5, None), (8.5, 0.25, c)):
        e = emb.clone().requires_grad_(True)
        d = torch.sum(torch.pow(e[idx] - (cc if cc is not None else 0.0), 2), 1) - r ** 2
        ref = r ** 2 + torch.mean(torch.relu(d)) / beta
        ref.backward()
        ed = emb.to(DEV).requires_grad_(True)
        loss, score = ocgnn_loss(ed, idx.to(DEV), cc.to(DEV) if cc is not None else None, r, beta)
        loss.backward()
        assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
        np.testing.assert_allclose(score.cpu().numpy(), d.detach().numpy(), rtol=2e-6, atol=2e-5)
        np.testing.assert_allclose(ed.grad.cpu().numpy(), e.grad.numpy(), rtol=2e-5, atol=1e-7)


def test_ocgnn_script_captured_epoch_equals_eager():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--no_graph"]):
        r = subprocess.run([sys.executable, os.path.join(root, "ocgnn.py"), "--dataset", "reddit", "--synthetic", "--num_epoch", "12"] + extra,
                           capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        keep = [l for l in r.stdout.splitlines() if l.startswith("Epoch:") or l.startswith("Testing")]
        assert len(keep) == 3 * 3
        outs.append(keep)
    assert outs[0] == outs[1]


def test_spmm_sliced_column_ranges_and_empty_rows(monkeypatch):
    """Segments cut at column-range boundaries and launched range by range (3 ranges forced on a small matrix), rows
    without entries, hub rows longer than one 512-entry segment, a row subset with repeated structure."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    n, m, w = 1500, 2100, 300
    a = sp.random(n, m, density=0.04, random_state=3, format="lil", dtype=np.float32)
    a[7, :] = 0
    a[8, :] = 0
    a[100, :] = rng.standard_normal(m).astype(np.float32)          # 2100 entries: 5 long segments, crosses every range
    a[n - 1, :] = 0
    csr = FG.Csr(a.tocsr(), DEV)
    x = torch.from_numpy(rng.standard_normal((m, w)).astype(np.float32)).to(DEV)
    ref = csr.host.astype(np.float64) @ x.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max() + 1.0
    rows = np.array([100, 7, 3, 1499, 100 - 1, 8, 640])
    monkeypatch.setenv("GGAD_SPMM_SLICED", "1")
    for ranges in ("1", "3", "7"):
        monkeypatch.setenv("GGAD_SPMM_COL_RANGES", ranges)
        for p in (csr.plan(), csr.plan(rows, key=("cr", ranges))):
            p["long"] = None
        got = FG.spmm(csr, x).cpu().numpy()
        assert np.abs(got - ref).max() / scale < 2e-6, ranges
        assert (got[[7, 8, n - 1]] == 0).all()
        sub = FG.spmm(csr, x, plan=csr.plan(rows, key=("cr", ranges))).cpu().numpy()
        assert np.abs(sub - ref[rows]).max() / scale < 2e-6, ranges
    lp = csr.plan()["long"]
    assert lp["n_multi"] > 0 and int((lp["seg_out"] < 0).sum().item()) > 5


@pytest.mark.parametrize("w", [300, 64, 256, 28, 4, 512])
def test_spmm_xcd_sliced_equals_row_major(w, monkeypatch):
    """The XCD-sliced SpMM (slice-major operand, 64 / L neighbours per load) against the wave-per-segment kernel: whole
    matrix with bias + PReLU + pre-activation, and a row subset, on a graph with hub rows (multi-segment) and leaves."""
    import scipy.sparse as sp
    n = 3000
    rowptr, col = synth.make_graph(n, 240000, 9, kind="powerlaw", max_degree=n // 3)
    a = synth.csr_to_scipy(rowptr, col, n)
    csr = FG.Csr(U.normalize_adj(a) + sp.eye(n), DEV)
    rng = np.random.default_rng(w)
    x = torch.from_numpy(rng.standard_normal((n, w)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.standard_normal(w).astype(np.float32)).to(DEV)
    slope = torch.tensor([0.25], device=DEV)
    rows = rng.permutation(n)[:257]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GGAD_SPMM_SLICED", mode)
        o, pre = FG.spmm(csr, x, bias=bias, prelu_a=slope, want_pre=True)
        sub = FG.spmm(csr, x, plan=csr.plan(rows, key=("t", w)))
        outs[mode] = (o.cpu().numpy(), pre.cpu().numpy(), sub.cpu().numpy())
    ref = (csr.host.astype(np.float64) @ x.cpu().numpy().astype(np.float64))
    scale = np.abs(ref).max() + 1.0
    for k in range(3):
        assert np.abs(outs["0"][k] - outs["1"][k]).max() / scale < 2e-6
    np.testing.assert_allclose(outs["1"][1], ref + bias.cpu().numpy(), atol=2e-6 * scale)
    np.testing.assert_allclose(outs["1"][2], ref[rows], atol=2e-6 * scale)
    monkeypatch.delenv("GGAD_SPMM_SLICED")
    assert FG._use_sliced(csr, csr.plan(), x) == (w >= 64 and n * w * 4 >= (6 << 20))


@pytest.mark.parametrize("shape", [(39357, 300), (1000, 300), (5, 12), (777, 28), (513, 30), (300, 1024)])
def test_prelu_backward_kernel_against_autograd(shape):
    """ggad_prelu_bwd_f32 (float4 path for widths that are multiples of 4, scalar path otherwise): dZ, the bias gradient (column
    sums of dZ) and the slope gradient against torch autograd of PReLU(z) with the same upstream gradient; deterministic."""
    from ggad_amd import _lib
    from ggad_amd._lib import call, ptr
    M, W = shape
    gen = torch.Generator(device="cpu").manual_seed(M + W)
    z = torch.randn(M, W, generator=gen).to(DEV)
    g = torch.randn(M, W, generator=gen).to(DEV)
    a = torch.tensor([0.25], device=DEV)
    S = int(_lib.load().ggad_prelu_bwd_splits(M))
    outs = []
    for _ in range(2):
        ws = torch.empty(2 * S * W, dtype=torch.float32, device=DEV)
        dz, db, da = torch.empty_like(z), torch.empty(W, device=DEV), torch.empty(1, device=DEV)
        call("ggad_prelu_bwd_f32", ptr(g), ptr(z), ptr(a), M, W, ptr(dz), ptr(db), ptr(da), ptr(ws))
        outs.append((dz, db, da))
    zr = z.double().requires_grad_(True)
    ar = a.double().requires_grad_(True)
    (torch.nn.functional.prelu(zr, ar) * g.double()).sum().backward()
    dz, db, da = outs[0]
    assert torch.equal(dz.double(), zr.grad.float().double())
    assert (db.double() - zr.grad.sum(0)).abs().max().item() <= 2e-5 * (1.0 + zr.grad.abs().sum(0).max().item())
    assert abs(da.item() - ar.grad.item()) <= 2e-5 * (1.0 + (g.double() * z.double()).abs().sum().item())
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    # round 6: the same in ONE launch (`ggad_prelu_bwd_one_f32`: the last workgroup reduces the partial column sums; shapes the vector
    # kernel does not take fall through to the two launches): dZ bit-equal, the two reductions to the same bound, repeatable bit for
    # bit (whichever workgroup draws the last ticket), the ticket word back at zero
    lib = _lib.load()
    tick = torch.zeros(int(lib.ggad_prelu_bwd_one_tickets()), dtype=torch.int32, device=DEV)
    one = []
    for _ in range(3):
        ws = torch.empty(int(lib.ggad_prelu_bwd_one_workspace_elems(M, W)), dtype=torch.float32, device=DEV)
        dz1, db1, da1 = torch.empty_like(z), torch.empty(W, device=DEV), torch.empty(1, device=DEV)
        call("ggad_prelu_bwd_one_f32", ptr(g), ptr(z), ptr(a), M, W, ptr(dz1), W, ptr(db1), ptr(da1), ptr(ws), ptr(tick))
        one.append((dz1, db1, da1))
    assert int(tick.abs().sum().item()) == 0
    assert torch.equal(one[0][0], dz)
    assert (one[0][1].double() - zr.grad.sum(0)).abs().max().item() <= 2e-5 * (1.0 + zr.grad.abs().sum(0).max().item())
    assert abs(one[0][2].item() - ar.grad.item()) <= 2e-5 * (1.0 + (g.double() * z.double()).abs().sum().item())
    for k in (1, 2):
        for x, y in zip(one[0], one[k]):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))


def test_prelu_backward_one_launch_same_workspace_changing_data():
    """`ggad_prelu_bwd_one_f32` hands its partial sums from workgroup to workgroup INSIDE the launch (write-through stores, tickets,
    L2-bypassing loads -- no fence): a stale line of the previous call's partials in some XCD's L2 would show as the previous call's
    sums.  Twelve calls on ONE workspace and ticket block with fresh data every time (sizes of the four configs' layers), each checked
    against an fp64 reduction; interleaved with a kernel that dirties the L2."""
    from ggad_amd import _lib
    from ggad_amd._lib import call, ptr
    lib = _lib.load()
    a = torch.tensor([0.25], device=DEV)
    for M, W in [(10984, 300), (39357, 300), (7535, 300), (513, 64)]:
        ws = torch.empty(int(lib.ggad_prelu_bwd_one_workspace_elems(M, W)), dtype=torch.float32, device=DEV)
        tick = torch.zeros(int(lib.ggad_prelu_bwd_one_tickets()), dtype=torch.int32, device=DEV)
        junk = torch.empty(8 << 20, device=DEV)
        for it in range(12):
            z = torch.randn(M, W, device=DEV) * (1.0 + it)
            g = torch.randn(M, W, device=DEV) + 0.1 * it
            dz, db, da = torch.empty_like(z), torch.empty(W, device=DEV), torch.empty(1, device=DEV)
            call("ggad_prelu_bwd_one_f32", ptr(g), ptr(z), ptr(a), M, W, ptr(dz), W, ptr(db), ptr(da), ptr(ws), ptr(tick))
            junk.add_(1.0)                                       # 32 MB of dirty lines between the calls
            ref_dz = torch.where(z > 0, g, 0.25 * g).double()
            ref_da = torch.where(z > 0, torch.zeros_like(g), g * z).double().sum()
            assert torch.equal(dz.double(), ref_dz.float().double())
            assert (db.double() - ref_dz.sum(0)).abs().max().item() <= 2e-5 * (1.0 + ref_dz.abs().sum(0).max().item()), (M, W, it)
            assert abs(da.item() - ref_da.item()) <= 2e-5 * (1.0 + (g.double() * z.double()).abs().sum().item()), (M, W, it)
        assert int(tick.abs().sum().item()) == 0


@pytest.mark.parametrize("r,h,ldx", [(1830, 300, 300), (6476, 300, 320), (1203, 300, 300), (37, 64, 64), (16, 12, 12), (5, 512, 512), (100, 20, 24),
                                     (3001, 128, 128)])
def test_fused_scorer_mlp_weight_gradients(r, h, ldx, monkeypatch):
    """The three weight gradients of the scorer (`model.py:176-180` backwards: dW1 = dz1^T x, dW2 = dz2^T f1, dW3 = g3^T f2) in ONE
    launch + one reduction (`ggad_mlp_score_wgrad_f32`, round 5: interleaved 16 VP x 16 VQ wave tiles, row ranges summed in order)
    against float64 and against the three split-K GEMMs they replace; row-strided x; deterministic; through `MlpScoreFn`."""
    h1, h2 = h // 2, h // 4
    rng = np.random.default_rng(7 * r + h)
    buf = torch.from_numpy(rng.standard_normal((r, ldx)).astype(np.float32)).to(DEV)
    x = buf[:, :h]
    dz1, f1 = (torch.from_numpy(rng.standard_normal((r, h1)).astype(np.float32)).to(DEV) for _ in range(2))
    dz2, f2 = (torch.from_numpy(rng.standard_normal((r, h2)).astype(np.float32)).to(DEV) for _ in range(2))
    g3 = torch.from_numpy(rng.standard_normal((r, 1)).astype(np.float32)).to(DEV)
    d1, d2, d3 = FG.mlp_score_wgrad(x, dz1, f1, dz2, f2, g3)
    assert tuple(d1.shape) == (h1, h) and tuple(d2.shape) == (h2, h1) and tuple(d3.shape) == (1, h2)
    for got, p, q in ((d1, dz1, x), (d2, dz2, f1), (d3, g3, f2)):
        ref = p.double().T @ q.double()
        assert ((got.double() - ref).abs().max() / (ref.abs().max() + 1.0)).item() < 2e-6
        via_gemm = FG.gemm(p.contiguous(), q.contiguous(), True, False)
        assert ((got - via_gemm).abs().max() / (ref.abs().max() + 1.0)).item() < 2e-6
    again = FG.mlp_score_wgrad(x, dz1, f1, dz2, f2, g3)
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((d1, d2, d3), again))
    # the autograd node: fused weight gradients against the GEMM ones
    w1 = torch.from_numpy((rng.standard_normal((h1, h)) / np.sqrt(h)).astype(np.float32)).to(DEV)
    w2 = torch.from_numpy((rng.standard_normal((h2, h1)) / np.sqrt(h1)).astype(np.float32)).to(DEV)
    w3 = torch.from_numpy((rng.standard_normal((1, h2)) / np.sqrt(h2)).astype(np.float32)).to(DEV)
    grads = []
    for fused in ("1", "0"):
        monkeypatch.setenv("GGAD_MLP_WGRAD_FUSED", fused)
        a, b, c = (t.clone().requires_grad_() for t in (w1, w2, w3))
        FG.MlpScoreFn.apply(x.contiguous(), a, b, c).backward(g3)
        grads.append((a.grad, b.grad, c.grad))
    for u, v in zip(*grads):
        assert ((u - v).abs().max() / (v.abs().max() + 1.0)).item() < 2e-6


@pytest.mark.parametrize("r,h", [(1830, 300), (6476, 300), (37, 64), (16, 12), (5, 512), (100, 20)])
def test_fused_scorer_mlp_forward_and_data_gradients(r, h, monkeypatch):
    """The scorer MLP of `model.py:176-180` (fc1 + relu, fc2 + relu, fc3, Linear weights without bias) as ONE launch forward and
    ONE launch for the data gradients (csrc/mlp.hip, v_mfma_f32_16x16x4_f32) against torch in float64: f_1, f_2, f_3, the masked
    gradients dz_2, dz_1, d_x (+ an incoming gradient of x), the three weight gradients through `MlpScoreFn`; deterministic; equal to
    the three-GEMM path to fp32 round-off."""
    h1, h2 = h // 2, h // 4
    rng = np.random.default_rng(r + h)
    x = torch.from_numpy(rng.standard_normal((r, h)).astype(np.float32)).to(DEV)
    w1 = torch.from_numpy((rng.standard_normal((h1, h)) / np.sqrt(h)).astype(np.float32)).to(DEV)
    w2 = torch.from_numpy((rng.standard_normal((h2, h1)) / np.sqrt(h1)).astype(np.float32)).to(DEV)
    w3 = torch.from_numpy((rng.standard_normal((1, h2)) / np.sqrt(h2)).astype(np.float32)).to(DEV)
    g3 = torch.from_numpy(rng.standard_normal((r, 1)).astype(np.float32)).to(DEV)
    gx = torch.from_numpy(rng.standard_normal((r, h)).astype(np.float32)).to(DEV)
    assert FG.mlp_score_supported(w1, w2, w3)
    f1, f2, f3 = FG.mlp_score_fwd(x, w1, w2, w3)
    xd, w1d, w2d, w3d = (t.double().cpu().requires_grad_() for t in (x, w1, w2, w3))
    r1 = torch.relu(xd @ w1d.T)
    r2 = torch.relu(r1 @ w2d.T)
    r3 = r2 @ w3d.T
    tol = lambda ref: 3e-6 * (1.0 + float(ref.abs().max()))            # noqa: E731
    assert (f1.double().cpu() - r1).abs().max().item() < tol(r1)
    assert (f2.double().cpu() - r2).abs().max().item() < tol(r2)
    assert (f3.double().cpu() - r3).abs().max().item() < tol(r3)
    again = FG.mlp_score_fwd(x, w1, w2, w3)
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip((f1, f2, f3), again))
    # data gradients: masks taken from the kernel's own f_1 / f_2 (an activation within round-off of 0 may flip in float64)
    dz2, dz1, dx = FG.mlp_score_dgrad(g3, f1, f2, w1, w2, w3, gx)
    m2, m1 = (f2 > 0).double().cpu(), (f1 > 0).double().cpu()
    e2 = (g3.double().cpu() @ w3d.detach()) * m2
    e1 = (e2 @ w2d.detach()) * m1
    ex = e1 @ w1d.detach() + gx.double().cpu()
    assert (dz2.double().cpu() - e2).abs().max().item() < tol(e2)
    assert (dz1.double().cpu() - e1).abs().max().item() < tol(e1)
    assert (dx.double().cpu() - ex).abs().max().item() < tol(ex)
    _, _, dx0 = FG.mlp_score_dgrad(g3, f1, f2, w1, w2, w3)
    assert (dx0.double().cpu() + gx.double().cpu() - ex).abs().max().item() < tol(ex)
    # autograd node against the three LinearFn nodes (GEMM path)
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("GGAD_MLP_FUSED", fused)
        xs, a, b, c = (t.clone().requires_grad_() for t in (x, w1, w2, w3))
        if fused == "1":
            y = FG.MlpScoreFn.apply(xs, a, b, c)
        else:
            assert not FG.mlp_score_supported(a, b, c)
            y = FG.LinearFn.apply(FG.LinearFn.apply(FG.LinearFn.apply(xs, a, True), b, True), c, False)
        (y * g3).sum().backward()
        outs.append([y.detach()] + [t.grad for t in (xs, a, b, c)])
    for u, v in zip(*outs):
        assert (u - v).abs().max().item() < 2e-5 * (1.0 + v.abs().max().item())


@pytest.mark.parametrize("ring", ["1", "0"])
@pytest.mark.parametrize("w", [300, 64, 4])
def test_spmm_lds_panel_equals_row_major(w, ring, monkeypatch):
    """The LDS-ring product (k_spmm_ring, ring = "1": operand slice through a ring of LDS slots staged by a loader wave, flexible
    schedule) and the LDS-panel product (k_spmm_panel: operand staged in 1,270-row panels, values factored into row / column scales, the
    diagonal applied in the epilogue) against the wave-per-segment kernel and scipy: normalised adjacency with `+ I`
    (3 panels, the last one partial) with bias + PReLU + pre-activation, and a 0/1 pattern matrix (diagonal inside the
    stream); deterministic; row subsets (panels for the pattern matrix, segments where the diagonal is separate); a matrix whose values do
    not factor keeps the other kernels."""
    import scipy.sparse as sp
    n = 3000
    rowptr, col = synth.make_graph(n, 240000, 9, kind="powerlaw", max_degree=n // 3)
    a = synth.csr_to_scipy(rowptr, col, n)
    rng = np.random.default_rng(w)
    x = torch.from_numpy(rng.standard_normal((n, w)).astype(np.float32)).to(DEV)
    bias = torch.from_numpy(rng.standard_normal(w).astype(np.float32)).to(DEV)
    slope = torch.tensor([0.25], device=DEV)
    monkeypatch.setenv("GGAD_SPMM_RING", ring)
    for mat in (U.normalize_adj(a) + sp.eye(n), U.normalize_adj(a + sp.eye(n)), (a + sp.eye(n)).tocsr()):
        csr = FG.Csr(mat, DEV)
        monkeypatch.setenv("GGAD_SPMM_PANEL", "0")
        monkeypatch.setenv("GGAD_SPMM_SLICED", "0")
        o0, pre0 = FG.spmm(csr, x, bias=bias, prelu_a=slope, want_pre=True)
        monkeypatch.setenv("GGAD_SPMM_PANEL", "1")
        assert ("wave_sb" in FG._use_panel(csr, csr.plan(), x)) == (ring == "1")
        o1, pre1 = FG.spmm(csr, x, bias=bias, prelu_a=slope, want_pre=True)
        o2 = FG.spmm(csr, x, bias=bias, prelu_a=slope)
        ref = csr.host.astype(np.float64) @ x.cpu().numpy().astype(np.float64) + bias.cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max() + 1.0
        assert np.abs(pre1.cpu().numpy() - ref).max() / scale < 2e-6
        assert (pre1 - pre0).abs().max().item() / scale < 2e-6 and (o1 - o0).abs().max().item() / scale < 2e-6
        assert torch.equal(o1.view(torch.int32), o2.view(torch.int32))
        rows = np.concatenate((rng.permutation(n)[:100], [5, 5]))               # a row subset (one row twice): panels when the
        sp_plan = csr.plan(rows, key=("p", w))                                 # matrix has no separate diagonal, else segments
        assert (FG._use_panel(csr, sp_plan, x) is None) == (csr.value_factors()[2] is not None)
        sub = FG.spmm(csr, x, plan=sp_plan, bias=bias).cpu().numpy()
        assert np.abs(sub - ref[rows]).max() / scale < 2e-6
    weighted = (U.normalize_adj(a) + sp.eye(n)).tocsr()
    weighted.data = weighted.data * rng.uniform(0.5, 1.5, size=weighted.nnz)
    csr = FG.Csr(weighted, DEV)
    assert FG._use_panel(csr, csr.plan(), x) is None


@pytest.mark.parametrize("ring", ["1", "0"])
def test_spmm_lds_panel_single_partial_panel_and_rectangular(ring, monkeypatch):
    """Operands shorter than one panel / ring slot (one partial panel, the zero rows behind it) and a rectangular pattern matrix (the N x |J|
    column subset of the affinity backward, `run.py:182-188`): forced LDS-panel product against scipy."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    monkeypatch.setenv("GGAD_SPMM_PANEL", "1")
    monkeypatch.setenv("GGAD_SPMM_RING", ring)
    for n_rows, n_src, dens in ((900, 900, 0.2), (2000, 700, 0.15), (300, 2900, 0.1), (150, 200, 0.5)):
        a = sp.random(n_rows, n_src, density=dens, random_state=7, format="csr", dtype=np.float32)
        a.data[:] = 1.0
        csr = FG.Csr(a, DEV)
        x = torch.from_numpy(rng.standard_normal((n_src, 300)).astype(np.float32)).to(DEV)
        assert FG._use_panel(csr, csr.plan(), x) is not None
        got = FG.spmm(csr, x).cpu().numpy()
        ref = a.astype(np.float64) @ x.cpu().numpy().astype(np.float64)
        assert np.abs(got - ref).max() / (np.abs(ref).max() + 1.0) < 2e-6, (n_rows, n_src)


def test_spmm_at_t_finance_size_against_scipy_and_exact_scaling():
    """BASELINE's largest full-graph config (39,357 nodes, 21.2 M directed entries, H = 300): the automatically chosen
    (LDS-panel) product against scipy in float64, exact under a factor 2, deterministic, with bias + PReLU epilogue; the
    XCD-sliced kernel on the same matrix."""
    import scipy.sparse as sp
    n, ne, w = 39357, 21222543, 300
    rowptr, col = synth.make_graph(n, ne, 0, kind="powerlaw", max_degree=n // 8, exact=True)
    assert abs(int(rowptr[-1]) - ne) <= 1
    a = synth.csr_to_scipy(rowptr, col, n)
    csr = FG.Csr(U.normalize_adj(a) + sp.eye(n), DEV)
    rng = np.random.default_rng(1)
    xh = rng.standard_normal((n, w)).astype(np.float32)
    x = torch.from_numpy(xh).to(DEV)
    assert FG._use_sliced(csr, csr.plan(), x) and FG._use_panel(csr, csr.plan(), x) is not None
    bias = torch.from_numpy(rng.standard_normal(w).astype(np.float32)).to(DEV)
    slope = torch.tensor([0.1], device=DEV)
    out, pre = FG.spmm(csr, x, bias=bias, prelu_a=slope, want_pre=True)
    ref = csr.host.astype(np.float64) @ xh.astype(np.float64) + bias.cpu().numpy().astype(np.float64)
    scale = np.abs(ref).max()
    assert np.abs(pre.cpu().numpy() - ref).max() / scale < 2e-6
    np.testing.assert_allclose(out.cpu().numpy(), np.where(ref > 0, ref, 0.1 * ref), atol=3e-6 * scale)
    plain = FG.spmm(csr, x)
    again = FG.spmm(csr, x)
    assert torch.equal(plain.view(torch.int32), again.view(torch.int32))                      # fixed summation order
    assert torch.equal((plain * 2.0).view(torch.int32), FG.spmm(csr, x * 2.0).view(torch.int32))
    os.environ["GGAD_SPMM_PANEL"] = "0"
    try:
        sliced = FG.spmm(csr, x)
    finally:
        del os.environ["GGAD_SPMM_PANEL"]
    assert (sliced - plain).abs().max().item() / scale < 2e-6


def test_gcn_layer_cached_aggregate_equals_reference_order(monkeypatch):
    """Input layer as (A_hat X) W^T with A_hat X cached (no SpMM per epoch, weight gradient dZ^T (A_hat X)) against the
    reference's order A_hat (X W^T): output, weight / bias / slope gradients; cache invalidated by an in-place change of X."""
    import scipy.sparse as sp
    n, f, h = 5000, 10, 300
    rowptr, col = synth.make_graph(n, 200000, 4, kind="powerlaw", max_degree=n // 8)
    a = synth.csr_to_scipy(rowptr, col, n)
    fa = FG.FullGraphAdj(U.normalize_adj(a) + sp.eye(n), a + sp.eye(n), DEV)
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.random((n, f)).astype(np.float32)).to(DEV)
    gout = torch.from_numpy(rng.standard_normal((n, h)).astype(np.float32)).to(DEV)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GGAD_GCN_REORDER", mode)
        torch.manual_seed(0)
        layer = GCN(f, h, "prelu").to(DEV)
        with torch.no_grad():
            layer.bias.copy_(torch.from_numpy(rng.standard_normal(h).astype(np.float32) * 0.1 if mode == "0" else res["bias"]))
        res.setdefault("bias", layer.bias.detach().cpu().numpy().copy())
        out = layer(x[None], fa)
        (out[0] * gout).sum().backward()
        res[mode] = (out[0].detach().cpu().numpy(), layer.fc.weight.grad.cpu().numpy(), layer.bias.grad.cpu().numpy(),
                     layer.act.weight.grad.cpu().numpy())
    for k in range(4):
        scale = np.abs(res["0"][k]).max() + 1e-6
        assert np.abs(res["0"][k] - res["1"][k]).max() / scale < 5e-6, k
    # second call hits the cache; an in-place update of X must not
    monkeypatch.setenv("GGAD_GCN_REORDER", "1")
    with torch.no_grad():
        o1 = layer(x[None], fa)[0].clone()
        x.mul_(2.0)
        o2 = layer(x[None], fa)[0]
        monkeypatch.setenv("GGAD_GCN_REORDER", "0")
        o3 = layer(x[None], fa)[0]
    assert not torch.allclose(o1, o2) and torch.allclose(o2, o3, rtol=1e-5, atol=1e-5)



def _padded(t, ld):
    """`t` copied into a view of an (rows, ld) matrix whose padding holds NaN (nothing may read it)."""
    buf = torch.full((t.shape[0], ld), float("nan"), device=t.device)
    v = buf[:, :t.shape[1]]
    v.copy_(t)
    return v


@pytest.mark.parametrize("w", [300, 256, 260, 512, 384])
def test_spmm_rowline_on_padded_rows_against_scipy(w):
    """The line-granular persistent product (k_spmm_rowline: operand rows on 128-byte lines, one line per XCD, the lines beyond the
    eighth split by rows) against scipy in float64: short rows, empty rows, medium rows (one wave), hub rows of one and of several
    passes (> 1,024 entries), bias + PReLU + pre-activation copy, a row subset with repeats, a padded destination.  The padding
    columns hold NaN: a kernel that reads them fails."""
    import scipy.sparse as sp
    rng = np.random.default_rng(w)
    n, m = 2600, 3100
    a = sp.random(n, m, density=0.006, random_state=5, format="lil", dtype=np.float32)
    a[7, :] = 0
    a[n - 1, :] = 0
    for r, k in ((100, 2500), (101, 1025), (640, 193), (641, 192), (900, 33), (901, 64), (1200, 1024)):
        a[r, :] = 0
        cols = rng.permutation(m)[:k]
        a[r, cols] = rng.standard_normal(k).astype(np.float32)
    csr = FG.Csr(a.tocsr(), DEV)
    xh = rng.standard_normal((m, w)).astype(np.float32)
    ld = (w + 31) // 32 * 32 + (32 if w % 32 == 0 else 0)           # (an aligned width gets a whole line of padding: still strided)
    x = _padded(torch.from_numpy(xh).to(DEV), ld)
    assert x.stride(0) == ld and FG._use_rowline(x)
    ref = csr.host.astype(np.float64) @ xh.astype(np.float64)
    scale = np.abs(ref).max() + 1.0
    got = FG.spmm(csr, x)
    assert got.is_contiguous()
    assert np.abs(got.cpu().numpy() - ref).max() / scale < 2e-6
    assert (got[[7, n - 1]] == 0).all()
    rs = csr.plan()["rowline"]
    assert rs["n_hub"] == 4 and rs["n_long"] >= 4 and rs["n_units"] * 8 >= n - rs["n_hub"] - rs["n_long"]
    bias = torch.from_numpy(rng.standard_normal(w).astype(np.float32)).to(DEV)
    slope = torch.tensor([0.25], device=DEV)
    dst = torch.full((n, ld), float("nan"), device=DEV)[:, :w]
    o, pre = FG.spmm(csr, x, bias=bias, prelu_a=slope, want_pre=True, out=dst)
    assert o.data_ptr() == dst.data_ptr() and pre.stride(0) == ld
    z = ref + bias.cpu().numpy()
    assert np.abs(pre.cpu().numpy() - z).max() / scale < 2e-6
    assert np.abs(o.cpu().numpy() - np.where(z > 0, z, 0.25 * z)).max() / scale < 2e-6
    assert torch.isnan(dst.as_strided((n, ld - w), (ld, 1), dst.storage_offset() + w)).all()          # the padding is not written
    rows = np.array([100, 7, 3, n - 1, 99, 100, 640, 641, 901] + list(range(1190, 1230)))
    sub = FG.spmm(csr, x, plan=csr.plan(rows, key=("rl", w)))
    assert np.abs(sub.cpu().numpy() - ref[rows]).max() / scale < 2e-6
    again = FG.spmm(csr, x)
    assert torch.equal(again.view(torch.int32), got.view(torch.int32))                                  # fixed summation order


def test_gcn_layer_with_padded_rows_equals_dense_rows(monkeypatch):
    """GcnLayerFn with the projection and the PReLU gradient written into 128-byte aligned rows (padded_rows -> k_spmm_rowline)
    against the same layer on contiguous rows (GGAD_SPMM_ROWLINE=0): output and every gradient; plus ggad_prelu_bwd_ld_f32 and a
    gemm with a padded destination on their own."""
    import scipy.sparse as sp
    from ggad_amd import _lib
    from ggad_amd._lib import call, ptr, ptr_rows
    n, f, h = 4000, 300, 300
    rowptr, col = synth.make_graph(n, 70000, 4, kind="powerlaw", max_degree=n // 8)
    a = synth.csr_to_scipy(rowptr, col, n)
    fa = FG.FullGraphAdj(U.normalize_adj(a) + sp.eye(n), a + sp.eye(n), DEV)
    rng = np.random.default_rng(2)
    xh = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).to(DEV)
    gout = torch.from_numpy(rng.standard_normal((n, h)).astype(np.float32)).to(DEV)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GGAD_SPMM_ROWLINE", mode)
        t = FG.padded_rows(n, h, DEV, (fa.A, None))
        assert (t.stride(0) == 320) == (mode == "1")
        torch.manual_seed(0)
        layer = GCN(f, h, "prelu").to(DEV)
        x = xh.clone().requires_grad_(True)
        out = layer(x[None], fa)
        (out[0] * gout).sum().backward()
        res[mode] = (out[0].detach().cpu().numpy(), x.grad.cpu().numpy(), layer.fc.weight.grad.cpu().numpy(), layer.bias.grad.cpu().numpy(),
                     layer.act.weight.grad.cpu().numpy())
    for k in range(5):
        scale = np.abs(res["0"][k]).max() + 1e-6
        assert np.abs(res["0"][k] - res["1"][k]).max() / scale < 5e-6, k
    # the strided PReLU gradient equals the dense one bit for bit; so does a gemm into a padded destination
    z, g = torch.randn(n, h, device=DEV), torch.randn(n, h, device=DEV)
    slope = torch.tensor([0.25], device=DEV)
    S = int(_lib.load().ggad_prelu_bwd_splits(n))
    ws = torch.empty(2 * S * h, device=DEV)
    dz0, db0, da0 = torch.empty_like(z), torch.empty(h, device=DEV), torch.empty(1, device=DEV)
    call("ggad_prelu_bwd_f32", ptr(g), ptr(z), ptr(slope), n, h, ptr(dz0), ptr(db0), ptr(da0), ptr(ws))
    dz1 = torch.full((n, 320), float("nan"), device=DEV)[:, :h]
    db1, da1 = torch.empty(h, device=DEV), torch.empty(1, device=DEV)
    call("ggad_prelu_bwd_ld_f32", ptr(g), ptr(z), ptr(slope), n, h, ptr_rows(dz1), 320, ptr(db1), ptr(da1), ptr(ws))
    assert torch.equal(dz0, dz1) and torch.equal(db0, db1) and torch.equal(da0, da1)
    w = torch.randn(h, f, device=DEV)
    c0 = FG.gemm(xh, w, False, True)
    c1 = FG.gemm(xh, w, False, True, out=torch.full((n, 320), float("nan"), device=DEV)[:, :h])
    assert c1.stride(0) == 320 and torch.equal(c0, c1)


def test_end_of_training_parity_full_graph_photo_schedule(capsys, monkeypatch):
    """BASELINE north_star, "AUROC/AUPRC within 1e-4": the WHOLE schedule the reference's script runs for `--dataset photo` (100 Adam
    epochs, noise N(0.02, 0.01), an evaluation every 10th epoch; run.py:137-240) through `run.fit` -- two eager epochs, then the captured
    epoch replayed 98 times -- against the imported reference's dense run on the same seeds (tests/golden/make_golden.py --part
    long_full).  N = 4,200 and H = 300: every projection goes through k_gemm_slab.  The deltas are printed (README quotes them);
    asserted: what 100 sequential fp32 Adam steps leave standing -- the loss curve to 2e-4, every AUROC / AP of the run to 1e-4."""
    import parity_long
    monkeypatch.setenv("GGAD_CAPTURE_BELOW_S", "10")      # (run.fit captures when the second eager epoch took less than this: a busy host must not decide the path)
    r = parity_long.full_graph_long()
    with capsys.disabled():
        print("\n[end-of-training parity, full graph]", r)
    assert r["epochs"] == 100 and r["captured"]
    assert r["loss_delta_max"] < 2e-4
    assert r["eval_auc_delta_max"] <= 1e-4 and r["eval_ap_delta_max"] <= 1e-4
    assert r["final_auc_delta"] <= 1e-4 and r["final_ap_delta"] <= 1e-4
    assert r["weight_norm_rel_delta_max"] < 1e-4


@pytest.mark.gpu
def test_end_of_training_parity_full_graph_planted_anomalies(capsys, monkeypatch):
    """End-of-training parity where AUROC MEANS something (round 6; tests/golden/make_golden.py --part planted_full,
    `synth.plant_anomalies`: attenuated features + neighbourhoods rewired towards each other, raw features as run.py keeps them for
    photo): `run.py --dataset photo --num_epoch 50` of the imported reference ends at AUROC 0.923 / AP 0.654 -- a ranking that separates
    the classes, not the 0.46 of labels drawn independently of the inputs -- and is well-conditioned there (ITS final AUROC / AP move
    by 3e-6 / 2e-5 under a 1e-7 relative change of its initial weights, stored in the fixture).  The HIP path must reproduce every
    AUROC / AP of the run to 1e-4 (north_star)."""
    import parity_long
    monkeypatch.setenv("GGAD_CAPTURE_BELOW_S", "10")
    r = parity_long.full_graph_long(fixture="fullgraph_long_planted.npz")
    with capsys.disabled():
        print("\n[end-of-training parity, full graph, planted anomalies, 50 epochs]", r)
    assert r["epochs"] == 50 and r["captured"]
    assert r["final_auc"][1] >= 0.8 and r["final_ap"][1] >= 0.5          # the REFERENCE separates the classes on this fixture
    assert r["loss_delta_max"] < 2e-4
    assert r["eval_auc_delta_max"] <= 1e-4 and r["eval_ap_delta_max"] <= 1e-4
    assert r["final_auc_delta"] <= 1e-4 and r["final_ap_delta"] <= 1e-4
    assert r["weight_norm_rel_delta_max"] < 1e-4


@pytest.mark.gpu
def test_end_of_training_planted_anomalies_100_epochs_within_the_reference_own_sensitivity(capsys):
    """The same planted problem over the script's default 100 epochs: at epoch 59 the margin hinge of run.py:195 re-activates for one
    step and the trajectory becomes ILL-conditioned -- the reference's own final AUROC / AP move by 6e-4 / 3e-3 when its initial weights
    change by 1e-7 relative (measured by the generator, stored as self_sens_*).  No implementation that is not bit-identical can hold
    1e-4 there; asserted: every evaluation up to epoch 50 to 1e-4 (the well-conditioned stretch), the final AUROC / AP within 3 x the
    reference's own sensitivity, and that the reference still separates the classes (0.938 / 0.769)."""
    import parity_long
    from conftest import load_golden
    g = load_golden("fullgraph_long_planted_100.npz")
    r = parity_long.full_graph_long(fixture="fullgraph_long_planted_100.npz")
    with capsys.disabled():
        print("\n[end-of-training parity, full graph, planted anomalies, 100 epochs]", r,
              {"reference_self_sensitivity": {"auc": float(g["self_sens_auc"]), "ap": float(g["self_sens_ap"]), "perturbation": float(g["self_sens_perturb"])}})
    assert r["epochs"] == 100 and r["final_auc"][1] >= 0.9 and r["final_ap"][1] >= 0.7
    assert float(g["self_sens_auc"]) > 1e-4 and float(g["self_sens_ap"]) > 1e-4          # (the reason this fixture is not held to 1e-4)
    assert max(r["eval_auc_delta_by_eval"][:6]) <= 1e-4 and max(r["eval_ap_delta_by_eval"][:6]) <= 1e-4      # evaluations of epochs 0 .. 50
    assert r["first_epoch_loss_delta_above"][1e-4] == -1 or r["first_epoch_loss_delta_above"][1e-4] >= 59
    assert r["final_auc_delta"] <= 3 * float(g["self_sens_auc"]) and r["final_ap_delta"] <= 3 * float(g["self_sens_ap"])


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,k", [(5000, 300, 64), (4100, 300, 28), (39357, 300, 20), (6000, 300, 300), (1000, 300, 64), (5000, 300, 25)])
def test_linear_prelu_in_one_launch_equals_gemm_then_prelu(m, n, k):
    """`ggad_linear_prelu_f32` (round 5, ABI 9; reference model.py:27-35 on a cached aggregate): where the slab GEMM takes the shape, z and
    out = PReLU(z) from one launch are BIT-identical to `ggad_gemm_f32` + `ggad_prelu_fwd_f32` (the same kernel computes z) and z matches an
    fp64 product; elsewhere (M < 4,096, K not a multiple of 4) the call returns GGAD_E_UNSUPPORTED and touches nothing."""
    from ggad_amd import _lib
    from ggad_amd._lib import call, ptr
    lib = _lib.load()
    torch.manual_seed(m + k)
    a = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * 0.2
    b = torch.randn(n, device=DEV)
    pa = torch.tensor([0.25], device=DEV)
    z = torch.full((m, n), 7.0, device=DEV)
    out = torch.full((m, n), 9.0, device=DEV)
    rc = int(lib.ggad_linear_prelu_f32(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(b), ptr(pa), m, n, k, ptr(z), z.stride(0), ptr(out),
                                       out.stride(0), _lib.current_stream()))
    torch.cuda.synchronize()
    if m < 4096 or k % 4:
        assert rc == -4 and bool((z == 7.0).all()) and bool((out == 9.0).all())
        return
    assert rc == 0
    z_ref = FG.gemm(a, w, False, True, bias=b)
    o_ref = torch.empty_like(z_ref)
    call("ggad_prelu_fwd_f32", ptr(z_ref), ptr(pa), z_ref.numel(), ptr(o_ref))
    assert torch.equal(z, z_ref) and torch.equal(out, o_ref)
    ref = a.double() @ w.double().T + b.double()
    assert float((z.double() - ref).abs().max()) <= 3e-6 * (1.0 + float(ref.abs().max()))
