"""GPU parity tests of the mini-batch comparison models that run on GGAD's 1-hop batch aggregate (drop-ins for
`src/graphsage_dominant.py`, `src/graphsage_anomalydae.py` and their handlers) against golden vectors captured from the
imported reference (`tests/golden/minibatch_baselines.npz`) and against the CPU oracle."""
import importlib
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from ggad_amd import synth
from oracle import ggad_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ggad_amd._lib import call, ptr
    from ggad_amd.fullgraph import FlatAdam
    from ggad_amd import sage_utils as SU
    from ggad_amd.sage_utils import recon_scores

DEV = "cuda:0"
CASES = [("dominant", "ggad_amd.graphsage_dominant", None), ("anomalydae", "ggad_amd.graphsage_anomalydae", 0.5)]


def _build(g, tag, modname):
    m = importlib.import_module(modname)
    adj = synth.csr_to_adj_lists(g["rowptr"], g["col"])          # the reference's container: dict of sets
    feats = torch.nn.Embedding(int(g["n"]), int(g["f"]))
    feats.weight = torch.nn.Parameter(torch.from_numpy(g["feat"]), requires_grad=False)
    agg = m.GCNAggregator(feats, cuda=True)
    enc = m.GCNEncoder(feats, int(g["f"]), int(g["d"]), adj, agg, gcn=True, cuda=True)
    model = m.GCN(2, enc)
    keys = sorted(k for k in model.state_dict().keys() if "features" not in k)
    assert keys == sorted(k[len(tag) + 6:] for k in g if k.startswith(tag + ".init."))     # the reference's parameter names
    with torch.no_grad():
        for k in keys:
            model.state_dict()[k].copy_(torch.from_numpy(g[f"{tag}.init.{k}"]))
    return adj, agg, enc, model


@pytest.mark.parametrize("tag,modname,pw", CASES)
def test_modules_and_training_trajectory(tag, modname, pw):
    g = load_golden("minibatch_baselines.npz")
    adj, agg, enc, model = _build(g, tag, modname)
    feat = torch.from_numpy(g["feat"])
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3, weight_decay=0.007)
    for step, nodes in enumerate(g["batches"]):
        nodes = nodes.tolist()
        opt.zero_grad()
        if step == 0:
            to_feats = agg.forward(nodes, [adj[int(v)] for v in nodes])                  # explicit neighbour sets
            assert to_feats.shape == g[f"{tag}.agg_to_feats"].shape
            np.testing.assert_allclose(to_feats.cpu().numpy(), g[f"{tag}.agg_to_feats"], atol=2e-6, rtol=0)
            x1, bp = agg.aggregate([nodes], adj)                                         # device CSR plan: the same numbers
            np.testing.assert_allclose(x1.cpu().numpy(), g[f"{tag}.agg_to_feats"], atol=2e-6, rtol=0)
            out = enc.forward(nodes)
            assert out.shape == g[f"{tag}.enc_out"].shape
            np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"{tag}.enc_out"], atol=3e-6, rtol=0)
            np.testing.assert_allclose(model.to_prob(nodes, None).detach().cpu().numpy(), g[f"{tag}.enc_out"], atol=3e-6, rtol=0)
        loss = model.loss(nodes, feat[nodes])                     # a CPU tensor, as the handler of the reference builds it
        loss.backward()
        assert abs(loss.item() - g[f"{tag}.losses"][step]) < 1e-5
        if step == 0:
            np.testing.assert_allclose(enc.weight.grad.cpu().numpy(), g[f"{tag}.grad.enc.weight"], atol=3e-6, rtol=1e-4)
            np.testing.assert_allclose(enc.fc.weight.grad.cpu().numpy(), g[f"{tag}.grad.enc.fc.weight"], atol=3e-6, rtol=1e-4)
            assert model.weight.grad is None                       # the scorer vector is never used by these models
        opt.step()
        if step == 0:
            np.testing.assert_allclose(enc.weight.detach().cpu().numpy(), g[f"{tag}.step1.enc.weight"], atol=3e-6, rtol=0)
            np.testing.assert_allclose(enc.fc.weight.detach().cpu().numpy(), g[f"{tag}.step1.enc.fc.weight"], atol=3e-6, rtol=0)
    np.testing.assert_allclose(enc.weight.detach().cpu().numpy(), g[f"{tag}.final.enc.weight"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(enc.fc.weight.detach().cpu().numpy(), g[f"{tag}.final.enc.fc.weight"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(model.weight.detach().cpu().numpy(), g[f"{tag}.init.weight"], atol=0, rtol=0)
    # test_recon: slices of 30 with a ragged tail, all slices in one plan
    sc = recon_scores(model, g["test_nodes"], int(g["test_bs"]), feat).cpu().numpy()
    np.testing.assert_allclose(sc, g[f"{tag}.test_scores"], atol=2e-5, rtol=0)
    from sklearn.metrics import average_precision_score, roc_auc_score
    y = (np.arange(len(sc)) % 5 == 0).astype(np.int64)
    auc, ap = SU.test_recon(g["test_nodes"], y, model, int(g["test_bs"]), feat, verbose=False)
    assert abs(auc - roc_auc_score(y, g[f"{tag}.test_scores"])) < 1e-4
    assert abs(ap - average_precision_score(y, g[f"{tag}.test_scores"])) < 1e-4


@pytest.mark.parametrize("shape", [(1, 1), (150, 17), (7, 300), (1000, 64)])
def test_recon_kernels_against_torch(shape):
    b, f = shape
    rng = np.random.default_rng(b * 31 + f)
    a = torch.from_numpy((rng.standard_normal((b, f)) * (rng.random((b, f)) > 0.3)).astype(np.float32))   # zeros like a ReLU output
    t = torch.from_numpy(rng.random((b, f)).astype(np.float32))
    for pw in (None, 0.5, 0.8):
        ar = a.clone().requires_grad_(True)
        ref = O.baseline_recon(ar, t, pw)
        ref.backward()
        wp, wn = (1.0, 1.0) if pw is None else (pw, 1.0 - pw)
        ad, td = a.to(DEV), t.to(DEV)
        loss = torch.empty(1, device=DEV)
        cs = torch.empty(f, device=DEV)
        da = torch.empty(b, f, device=DEV)
        call("ggad_recon_cols_f32", ptr(ad), ptr(td), b, f, wp, wn, ptr(loss), ptr(cs), ptr(da))
        assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
        np.testing.assert_allclose(da.cpu().numpy(), ar.grad.numpy(), atol=1e-6, rtol=2e-5)
    rows = torch.empty(b, device=DEV)
    call("ggad_recon_rows_f32", ptr(ad), ptr(td), b, f, ptr(rows))
    np.testing.assert_allclose(rows.cpu().numpy(), torch.sqrt(torch.sum((a - t) ** 2, 1)).numpy(), rtol=2e-6, atol=1e-7)


def _handler_cfg(data, **kw):
    cfg = dict(data_name="dgraphfin", data_dir="./data/", train_ratio=0.4, test_ratio=0.67, save_dir="./pytorch_models/",
               model="GCN", multi_relation="GNN", emb_size=64, thres=0.4, rho=0.5, seed=72, optimizer="adam", lr=0.001,
               weight_decay=0.007, batch_size=150, num_epochs=2, valid_epochs=5, alpha=2, no_cuda=False, cuda_id="0", data=data)
    cfg.update(kw)
    return cfg


@pytest.mark.parametrize("which,pw,frac", [("dominate", None, 0.10), ("anomalydae", 0.5, 0.05)])
def test_handler_epochs_equal_the_oracle_loop(which, pw, frac):
    """Two epochs of the handler (split, in-place shuffles continuing python's `random` stream, 150-slice schedule, Adam)
    against the same loop written with the CPU oracle (`src/model_handler_dominate.py:29-56,133-163`)."""
    mh = importlib.import_module(f"ggad_amd.model_handler_{which}")
    n, f = 30000, 17
    rowptr, col = synth.make_graph(n, 150000, 3, kind="powerlaw", max_degree=300)
    feat_raw = synth.make_features(n, f, 3)
    y = synth.make_labels(n, 0.02, 3).astype(np.int32)
    nb = 12
    torch.manual_seed(72)
    np.random.seed(72)
    h = mh.ModelHandler(_handler_cfg(((rowptr, col), feat_raw, y), num_batches=nb))
    idx_train0 = list(h.dataset["idx_train"])
    state_after_split = random.getstate()
    h.train()
    state_after_train = random.getstate()
    # oracle loop: same initial weights (same torch RNG draws), python's own shuffle
    torch.manual_seed(72)
    torch.nn.Embedding(n, f)
    w = torch.nn.init.xavier_uniform_(torch.empty(64, f)).requires_grad_(True)
    fc = torch.nn.Linear(64, f, bias=False).weight.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([w, fc], lr=0.001, weight_decay=0.007)
    feat = np.asarray(h.dataset["feat_data"], dtype=np.float32)
    random.setstate(state_after_split)
    idx = idx_train0
    for epoch in range(2):
        random.shuffle(idx)
        for b in range(nb):
            nodes = idx[b * 150:(b + 1) * 150]
            opt.zero_grad()
            loss, _ = O.baseline_loss(w, fc, rowptr, col, feat, nodes, feat[nodes], pw)
            loss.backward()
            opt.step()
            assert abs(loss.item() - h.epoch_losses[epoch][b]) < 2e-5, (epoch, b)
    np.testing.assert_allclose(h.model.enc.weight.detach().cpu().numpy(), w.detach().numpy(), atol=3e-5, rtol=0)
    np.testing.assert_allclose(h.model.enc.fc.weight.detach().cpu().numpy(), fc.detach().numpy(), atol=3e-5, rtol=0)
    assert random.getstate() == state_after_train          # the handler handed python's `random` stream back where the loop left it
    assert len(h.valid_history) == 1                       # validated at epoch 0 only (valid_epochs = 5)
    # the validation numbers of epoch 0 came from weights we no longer have; re-score with the final weights instead
    sc = recon_scores(h.model, h.dataset["idx_valid"][:1000], 150, torch.from_numpy(feat)).cpu().numpy()
    ref = O.baseline_scores(w.detach(), fc.detach(), rowptr, col, feat, h.dataset["idx_valid"][:1000], 150, feat)
    np.testing.assert_allclose(sc, ref, atol=3e-5, rtol=0)
    # the pseudo-anomaly fraction of this handler's split
    n_lab = len(h.dataset["idx_labeled"])
    assert int(h.dataset["labels"].sum()) == int(y.sum()) + int(n_lab * frac)


def test_aegis_minibatch_model_against_the_oracle_restatement(tmp_path, capsys):
    """Mini-batch AEGIS-style model (`src/graphsage_aegis.py`, `src/model_handler_aegis.py`).  PARITY UNPINNED: its
    discriminator is `torch_geometric.nn.MLP`, absent from the image, so the HIP path (two 1-hop aggregates per batch from one plan,
    MFMA projections and MLP linears) is compared with the oracle's restatement of the same published layer stack: logits, both losses,
    every gradient, a 4-step Adam trajectory; `to_prob`; the handler end to end (schedule, validation prints)."""
    import random
    from ggad_amd import graphsage_aegis as M
    from ggad_amd.fullgraph import FlatAdam
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.graphsage import FeatureTable
    n, f, d = 6000, 17, 64
    rowptr, col = synth.make_graph(n, 60000, 5, kind="powerlaw", max_degree=300)
    feat = O.normalize_rows(synth.make_features(n, f, 5)).astype(np.float32)
    torch.manual_seed(11)
    features = FeatureTable(torch.from_numpy(feat))
    agg = M.GCNAggregator(features, feat, cuda=True)
    enc = M.GCNEncoder(features, f, d, DeviceGraph(rowptr, col, DEV), agg, gcn=True, cuda=True)
    model = M.GCN(2, enc).to(DEV)
    features.to(DEV)
    names = [k for k, p in model.named_parameters() if p.requires_grad]
    assert "enc.discriminator2.lins.0.weight" in names and "enc.discriminator2.norms.0.module.weight" in names and "enc.weight" in names
    P = {k: p.detach().cpu().clone().requires_grad_() for k, p in model.named_parameters() if p.requires_grad}
    noise = agg.noise.numpy()
    rng = np.random.default_rng(3)
    batches = [rng.choice(n, size=150, replace=False) for _ in range(4)]
    # forward / losses / gradients of the first batch
    la, lg, lab = model(batches[0])
    ra, rg, rlab = O.aegis_forward(P, rowptr, col, feat, noise, batches[0])
    np.testing.assert_allclose(la[:, 0].detach().cpu().numpy(), ra.detach().numpy(), atol=3e-6, rtol=0)
    np.testing.assert_allclose(lg[:, 0].detach().cpu().numpy(), rg.detach().numpy(), atol=3e-6, rtol=0)
    assert np.array_equal(lab.cpu().numpy(), rlab.numpy())
    np.testing.assert_allclose(model.to_prob(batches[0])[:, 0].detach().cpu().numpy(), ra.detach().numpy()[:150], atol=3e-6, rtol=0)
    # 4 optimiser steps: both losses back-propagated, one Adam step (src/model_handler_aegis.py:152-158)
    opt = FlatAdam([p for p in model.parameters() if p.requires_grad], lr=0.005, weight_decay=0.007)
    used = [k for k in names if k == "enc.weight" or k.startswith("enc.discriminator2")]
    ref_opt = O.make_adam([P[k] for k in used], 0.005, 0.007)
    for b in range(4):
        opt.zero_grad()
        l1, l2 = model.loss(batches[b])
        (l1 + l2).backward()
        ref_opt.zero_grad()
        r1, r2 = O.aegis_loss(P, rowptr, col, feat, noise, batches[b])
        (r1 + r2).backward()
        np.testing.assert_allclose([l1.item(), l2.item()], [r1.item(), r2.item()], atol=5e-6, rtol=0)
        if b == 0:
            got = dict(model.named_parameters())
            for k in used:
                np.testing.assert_allclose(got[k].grad.cpu().numpy(), P[k].grad.numpy(), atol=5e-6, rtol=2e-4, err_msg=k)
            for k in names:
                if k not in used:
                    assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k      # generator / discriminator / fc / weight: unused
        opt.step()
        ref_opt.step()
    got = dict(model.named_parameters())
    for k in used:
        np.testing.assert_allclose(got[k].detach().cpu().numpy(), P[k].detach().numpy(), atol=2e-5, rtol=0, err_msg=k)
    # the handler: schedule (idx_train + idx_test shuffled per epoch, 100 -> here 5 batches), both losses, validation
    from ggad_amd.model_handler_aegis import ModelHandler
    lab = synth.make_labels(n, 0.05, 5)
    cfg = dict(data_name="synthetic", data_dir="", data=((rowptr, col), synth.make_features(n, f, 5), lab), seed=72, model="GCN",
               multi_relation="GNN", emb_size=64, thres=0.4, lr=0.005, weight_decay=0.007, batch_size=90, num_epochs=3, valid_epochs=2,
               num_batches=5, save_dir=str(tmp_path) + "/", test_ratio=0.67, device=0)
    random.seed(72)
    np.random.seed(72)
    torch.manual_seed(72)
    h = ModelHandler(cfg)
    assert h.train() is None
    out = capsys.readouterr().out
    assert "loss_g:" in out and "loss_gen:" in out and "Testing AUC" in out and "Testing AP:" in out
    assert len(h.epoch_losses) == 3 and h.epoch_losses[0].shape == (5, 2) and np.isfinite(np.stack(h.epoch_losses)).all()
    assert len(h.valid_history) == 2 and all(0.0 <= v[1] <= 1.0 for v in h.valid_history)
    assert np.stack(h.epoch_losses)[-1, :, 0].mean() < np.stack(h.epoch_losses)[0, :, 0].mean()      # the discriminator learns
