"""Pin the CPU oracle (oracle/ggad_oracle.py) against outputs captured from the imported reference.

The reference has no tests or vectors of its own; tests/golden/*.npz hold what its
code returned on seeded synthetic inputs in the build container (tests/golden/make_golden.py).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from ggad_amd import synth
from oracle import ggad_oracle as O

TOL = 2e-6


def _mini_params(g, prefix):
    return O.MiniParams(torch.tensor(g[prefix + ".weight"], requires_grad=True),
                        torch.tensor(g[prefix + ".enc.weight"], requires_grad=True),
                        torch.tensor(g[prefix + ".enc.fc.weight"], requires_grad=True))


@pytest.mark.parametrize("name", ["minibatch_small.npz", "minibatch_dense.npz"])
def test_inputs_regenerate(name):
    g = load_golden(name)
    assert synth.crc_of(g["rowptr"], g["col"], g["feat_raw"]) == int(g["inputs_crc"])
    np.testing.assert_allclose(O.normalize_rows(g["feat_raw"]).astype(np.float32), g["feat"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("name", ["minibatch_small.npz", "minibatch_dense.npz"])
def test_aggregator_closed_form(name):
    g = load_golden(name)
    nodes = g["batches"][0]
    agg = O.aggregate_batch(g["rowptr"], g["col"], g["feat"], nodes, True)
    np.testing.assert_allclose(agg.to_feats, g["agg_to_feats"], atol=TOL, rtol=0)
    # the reference orders U by python-set iteration; map through the recorded list
    ref_u = g["agg_unique"]
    assert sorted(ref_u.tolist()) == agg.unique.tolist()
    perm = np.searchsorted(agg.unique, ref_u)
    np.testing.assert_allclose(agg.to_feats_neigh[perm], g["agg_to_feats_neigh"], atol=TOL, rtol=0)
    np.testing.assert_allclose(agg.mask_row_dense()[:, perm], g["agg_mask_row"], atol=1e-7, rtol=0)
    # dense-faithful port agrees as well
    adj = synth.csr_to_adj_lists(g["rowptr"], g["col"])
    tf, tfn, mrow, ulist = O.aggregate_batch_dense(adj, torch.from_numpy(g["feat"]), nodes.tolist(), True)
    assert ulist == ref_u.tolist()
    np.testing.assert_allclose(tf.numpy(), g["agg_to_feats"], atol=1e-7, rtol=0)
    np.testing.assert_allclose(tfn.numpy(), g["agg_to_feats_neigh"], atol=1e-7, rtol=0)
    np.testing.assert_allclose(mrow.numpy(), g["agg_mask_row"], atol=0, rtol=0)


@pytest.mark.parametrize("name", ["minibatch_small.npz", "minibatch_dense.npz"])
def test_encoder_loss_grads_and_adam_trajectory(name):
    g = load_golden(name)
    p = _mini_params(g, "init")
    opt = O.make_adam(p.tensors(), 1e-3, 0.007)
    for step, (nodes, lab) in enumerate(zip(g["batches"], g["labels"])):
        agg = O.aggregate_batch(g["rowptr"], g["col"], g["feat"], nodes, True)
        if step == 0:
            with torch.no_grad():
                ca, nbar, af, afn = O.encoder_forward(p, agg, lab, True)
            np.testing.assert_allclose(ca.numpy(), g["enc_combined_all"], atol=TOL, rtol=0)
            np.testing.assert_allclose(nbar.numpy(), g["enc_to_feats_neigh"], atol=TOL, rtol=0)
            np.testing.assert_allclose(af.numpy(), g["enc_anomaly_feat"], atol=TOL, rtol=0)
            np.testing.assert_allclose(afn.numpy(), g["enc_anomaly_feat_new"], atol=TOL, rtol=0)
        opt.zero_grad()
        total, cls, margin, rec = O.batch_loss(p, agg, lab)
        total.backward()
        got = np.array([total.item(), cls.item(), margin.item(), rec.item()])
        np.testing.assert_allclose(got, g["losses"][step], atol=5e-6, rtol=0)
        if step == 0:
            np.testing.assert_allclose(p.weight.grad.numpy(), g["grad.weight"], atol=TOL, rtol=1e-5)
            np.testing.assert_allclose(p.enc_weight.grad.numpy(), g["grad.enc.weight"], atol=TOL, rtol=1e-5)
            np.testing.assert_allclose(p.enc_fc_weight.grad.numpy(), g["grad.enc.fc.weight"], atol=TOL, rtol=1e-5)
        opt.step()
        if step == 0:
            np.testing.assert_allclose(p.enc_weight.detach().numpy(), g["step1.enc.weight"], atol=TOL, rtol=0)
    np.testing.assert_allclose(p.weight.detach().numpy(), g["final.weight"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(p.enc_weight.detach().numpy(), g["final.enc.weight"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(p.enc_fc_weight.detach().numpy(), g["final.enc.fc.weight"], atol=2e-5, rtol=0)


@pytest.mark.parametrize("name", ["minibatch_small.npz", "minibatch_dense.npz"])
def test_to_prob_reference_batches(name):
    g = load_golden(name)
    p = _mini_params(g, "final")
    bs = int(g["test_bs"])
    nodes = g["test_nodes"]
    got = []
    for s in range(0, len(nodes), bs):
        got.extend(O.to_prob(p, g["rowptr"], g["col"], g["feat"], nodes[s:s + bs]).tolist())
    np.testing.assert_allclose(np.array(got, dtype=np.float32), g["test_probs"], atol=TOL, rtol=0)


def test_mean_aggregator_and_encoder(g_mini_small):
    g = g_mini_small
    nodes = g["batches"][0]
    mean = O.mean_aggregate(g["rowptr"], g["col"], g["feat"], nodes, gcn=False)
    np.testing.assert_allclose(mean, g["sage_mean"], atol=TOL, rtol=0)
    w = torch.from_numpy(g["sage_weight"])
    comb = torch.cat((torch.from_numpy(g["feat"][nodes]), torch.from_numpy(mean)), 1)
    np.testing.assert_allclose(torch.relu(w.mm(comb.t())).numpy(), g["sage_enc"], atol=TOL, rtol=0)
    mean_g = O.mean_aggregate(g["rowptr"], g["col"], g["feat"], nodes, gcn=True)
    w2 = torch.from_numpy(g["sage_gcn_weight"])
    np.testing.assert_allclose(torch.relu(w2.mm(torch.from_numpy(mean_g).t())).numpy(), g["sage_gcn_enc"],
                               atol=TOL, rtol=0)


@pytest.mark.parametrize("name", ["minibatch_small.npz", "minibatch_dense.npz"])
def test_intra_agg(name):
    """IntraAgg (src/layers.py:179-244) against what the reference module returned (columns / rows matched by node id:
    the reference orders them by python-set iteration)."""
    g = load_golden(name)
    tf, tfn, mask, unique = O.intra_aggregate(g["rowptr"], g["col"], g["feat"], g["batches"][0], g["intra_weight"])
    order = np.argsort(g["intra_unique"])
    assert np.array_equal(g["intra_unique"][order], unique)
    np.testing.assert_allclose(tf, g["intra_to_feats"], atol=TOL, rtol=0)
    np.testing.assert_allclose(tfn, g["intra_to_feats_neigh"][order], atol=TOL, rtol=0)
    np.testing.assert_allclose(mask, g["intra_mask"][:, order], atol=TOL, rtol=0)


# ------------------------------------------------------------------ full graph
def _full_setup(g):
    adjn_rp, adjn_ci, adjn_va, raw_rp, raw_ci, raw_va = O.normalize_adj(g["rowptr"], g["col"])
    return (adjn_rp, adjn_ci, adjn_va), (raw_rp, raw_ci, raw_va)


@pytest.mark.parametrize("name", ["fullgraph_reddit_like.npz", "fullgraph_amazon_like.npz"])
def test_full_preprocessing(name):
    g = load_golden(name)
    assert synth.crc_of(g["rowptr"], g["col"], g["feat_raw"], g["ano"]) == int(g["inputs_crc"])
    np.testing.assert_allclose(O.preprocess_features(g["feat_raw"]).astype(np.float32), g["features"],
                               atol=1e-7, rtol=0)
    (rp, ci, va), _ = _full_setup(g)
    n = int(g["n"])
    import scipy.sparse as sp
    ref = sp.coo_matrix((g["adjn_val"], (g["adjn_row"], g["adjn_col"])), shape=(n, n)).tocsr()
    ref.sum_duplicates(); ref.sort_indices()
    assert np.array_equal(ref.indptr, rp) and np.array_equal(ref.indices, ci)
    np.testing.assert_allclose(va, ref.data, atol=1e-15, rtol=0)


@pytest.mark.parametrize("name", ["fullgraph_reddit_like.npz", "fullgraph_amazon_like.npz"])
def test_full_forward_loss_grads_trajectory(name):
    g = load_golden(name)
    adjn, raw = _full_setup(g)
    P = {k: torch.tensor(g["init." + k], requires_grad=True) for k in O.FULL_PARAM_ORDER}
    opt = O.make_adam(list(P.values()), 1e-3, 0.0)
    feat = torch.from_numpy(g["features"])
    abn, nrm = g["abn_idx"], g["normal_idx"]
    mean, var, h = float(g["mean"]), float(g["var"]), int(g["n_h"])
    for step in range(len(g["losses"])):
        torch.manual_seed(1000 + step)
        noise = torch.randn(1, len(abn), h)[0] * var + mean
        opt.zero_grad()
        emb, comb, logits, con, eab = O.full_forward(P, feat, adjn, abn, nrm, noise, True)
        total, lm, lb, lr, aff = O.full_loss(emb, logits, con, eab, raw, abn, nrm)
        total.backward()
        np.testing.assert_allclose([total.item(), lm.item(), lb.item(), lr.item()], g["losses"][step], atol=1e-5)
        if step == 0:
            np.testing.assert_allclose(emb.detach().numpy(), g["emb"], atol=TOL)
            np.testing.assert_allclose(comb.detach().numpy(), g["emb_combine"], atol=TOL)
            np.testing.assert_allclose(logits.detach().numpy(), g["logits"], atol=TOL)
            np.testing.assert_allclose(con.detach().numpy(), g["emb_con"], atol=TOL)
            np.testing.assert_allclose(eab.detach().numpy(), g["emb_abnormal"], atol=TOL)
            np.testing.assert_allclose(aff.detach().numpy(), g["affinity"], atol=TOL)
            for k in O.FULL_PARAM_ORDER:
                np.testing.assert_allclose(P[k].grad.numpy(), g["grad." + k], atol=3e-6, rtol=1e-4, err_msg=k)
        opt.step()
    for k in O.FULL_PARAM_ORDER:
        np.testing.assert_allclose(P[k].detach().numpy(), g["final." + k], atol=3e-5, err_msg=k)
    torch.manual_seed(5000)
    noise = torch.randn(1, len(abn), h)[0] * var + mean
    with torch.no_grad():
        _, _, le, _, _ = O.full_forward(P, feat, adjn, abn, nrm, noise, False)
    np.testing.assert_allclose(le.numpy(), g["eval_logits"], atol=3e-5)
    from sklearn.metrics import roc_auc_score, average_precision_score
    yt = g["ano"][g["idx_test"]]
    assert abs(roc_auc_score(yt, le.numpy()[g["idx_test"]]) - float(g["eval_auc"])) < 1e-4
    assert abs(average_precision_score(yt, le.numpy()[g["idx_test"]]) - float(g["eval_ap"])) < 1e-4


@pytest.mark.parametrize("name", ["fullgraph_reddit_like.npz", "fullgraph_amazon_like.npz"])
def test_full_loss_by_column_equals_per_edge_form_and_reference(name):
    """The column-sum association of the affinity (what the full-size GPU tests use as their oracle: the per-edge form needs an
    (edges x H) intermediate) against the reference's vectors: loss terms, affinity, every gradient of the first step."""
    g = load_golden(name)
    adjn, raw = _full_setup(g)
    P = {k: torch.tensor(g["init." + k], requires_grad=True) for k in O.FULL_PARAM_ORDER}
    feat = torch.from_numpy(g["features"])
    abn, nrm = g["abn_idx"], g["normal_idx"]
    torch.manual_seed(1000)
    noise = torch.randn(1, len(abn), int(g["n_h"]))[0] * float(g["var"]) + float(g["mean"])
    emb, comb, logits, con, eab = O.full_forward(P, feat, adjn, abn, nrm, noise, True)
    total, lm, lb, lr, aff = O.full_loss(emb, logits, con, eab, raw, abn, nrm, by_column=True)
    total.backward()
    np.testing.assert_allclose([total.item(), lm.item(), lb.item(), lr.item()], g["losses"][0], atol=1e-5)
    np.testing.assert_allclose(aff.detach().numpy(), g["affinity"], atol=TOL)
    for k in O.FULL_PARAM_ORDER:
        np.testing.assert_allclose(P[k].grad.numpy(), g["grad." + k], atol=3e-6, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("tag,pw", [("dominant", None), ("anomalydae", 0.5)])
def test_baseline_models_on_the_1hop_aggregate(tag, pw):
    """DOMINANT / AnomalyDAE mini-batch variants (src/graphsage_dominant.py, src/graphsage_anomalydae.py): aggregate, decoder
    output, loss trajectory under Adam(1e-3, wd 0.007), first-step gradients, scores of test_recon."""
    g = load_golden("minibatch_baselines.npz")
    assert synth.crc_of(g["rowptr"], g["col"], g["feat_raw"]) == int(g["inputs_crc"])
    feat = g["feat"]
    w = torch.tensor(g[f"{tag}.init.enc.weight"], requires_grad=True)
    fc = torch.tensor(g[f"{tag}.init.enc.fc.weight"], requires_grad=True)
    opt = torch.optim.Adam([w, fc], lr=1e-3, weight_decay=0.007)
    for step, nodes in enumerate(g["batches"]):
        opt.zero_grad()
        if step == 0:
            agg = O.aggregate_batch(g["rowptr"], g["col"], feat, nodes, False)
            np.testing.assert_allclose(agg.to_feats, g[f"{tag}.agg_to_feats"], atol=TOL, rtol=0)
        loss, rec = O.baseline_loss(w, fc, g["rowptr"], g["col"], feat, nodes, feat[nodes], pw)
        loss.backward()
        assert abs(loss.item() - g[f"{tag}.losses"][step]) < 2e-6
        if step == 0:
            np.testing.assert_allclose(rec.detach().numpy(), g[f"{tag}.enc_out"], atol=TOL, rtol=0)
            np.testing.assert_allclose(w.grad.numpy(), g[f"{tag}.grad.enc.weight"], atol=TOL, rtol=0)
            np.testing.assert_allclose(fc.grad.numpy(), g[f"{tag}.grad.enc.fc.weight"], atol=TOL, rtol=0)
        opt.step()
        if step == 0:
            np.testing.assert_allclose(w.detach().numpy(), g[f"{tag}.step1.enc.weight"], atol=TOL, rtol=0)
    np.testing.assert_allclose(w.detach().numpy(), g[f"{tag}.final.enc.weight"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(fc.detach().numpy(), g[f"{tag}.final.enc.fc.weight"], atol=1e-5, rtol=0)
    sc = O.baseline_scores(w.detach(), fc.detach(), g["rowptr"], g["col"], feat, g["test_nodes"], int(g["test_bs"]), feat)
    np.testing.assert_allclose(sc, g[f"{tag}.test_scores"], atol=1e-5, rtol=0)


def test_ocgnn_forward_loss_trajectory():
    """Full-graph OCGNN comparison model (`model_ocgnn.py`, loss / step of `ocgnn.py:83-118,170-186`)."""
    g = load_golden("fullgraph_ocgnn.npz")
    assert synth.crc_of(g["rowptr"], g["col"], g["feat_raw"], g["ano"]) == int(g["inputs_crc"])
    adjn, _ = _full_setup(g)
    keys = ["gcn1.bias", "gcn1.fc.weight", "gcn1.act.weight", "gcn2.bias", "gcn2.fc.weight", "gcn2.act.weight"]
    P = {k: torch.tensor(g["init." + k], requires_grad=True) for k in keys}
    opt = O.make_adam(list(P.values()), 1e-3, 0.0)
    feat = torch.from_numpy(g["features"])
    nrm = torch.from_numpy(g["normal_idx"]).long()
    for step in range(len(g["losses"])):
        opt.zero_grad()
        emb = O.ocgnn_forward(P, feat, adjn)
        loss, score = O.ocgnn_loss(emb[nrm])
        loss.backward()
        assert abs(loss.item() - g["losses"][step]) < 1e-5
        if step == 0:
            np.testing.assert_allclose(emb.detach().numpy(), g["emb"], atol=TOL)
            np.testing.assert_allclose(score.detach().numpy(), g["score"], atol=1e-5)
            for k in keys:
                np.testing.assert_allclose(P[k].grad.numpy(), g["grad." + k], atol=3e-6, rtol=1e-4, err_msg=k)
        opt.step()
    for k in keys:
        np.testing.assert_allclose(P[k].detach().numpy(), g["final." + k], atol=3e-5, err_msg=k)
    with torch.no_grad():
        _, sc = O.ocgnn_loss(O.ocgnn_forward(P, feat, adjn))
    np.testing.assert_allclose(sc.numpy(), g["eval_score"], atol=5e-5)


def test_aegis_mlp_restatement_equals_torch_layers():
    """The oracle's restatement of torch_geometric.nn.MLP's 2-layer stack (Linear -> BatchNorm1d in training mode -> act -> Linear;
    parity with the absent library itself is UNPINNED) against torch's own layers with the same parameters."""
    import torch.nn as nn
    torch.manual_seed(5)
    lin0, bn, lin1 = nn.Linear(64, 64), nn.BatchNorm1d(64), nn.Linear(64, 1)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.2, 0.2)
    P = {"d.lins.0.weight": lin0.weight, "d.lins.0.bias": lin0.bias, "d.norms.0.module.weight": bn.weight,
         "d.norms.0.module.bias": bn.bias, "d.lins.1.weight": lin1.weight, "d.lins.1.bias": lin1.bias}
    x = torch.randn(300, 64)
    bn.train()
    ref = lin1(torch.sigmoid(bn(lin0(x))))
    got = O.aegis_mlp(P, "d", x, torch.sigmoid)
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), atol=2e-6, rtol=0)


def test_planted_anomaly_generator_keeps_the_graph_simple_and_pins_the_fixtures():
    """`synth.plant_anomalies` (round 6: end-of-training parity on labels that mean something): the planted graph stays symmetric and
    simple with every node keeping a neighbour, only the labelled rows of the feature table change, the same seed gives the same
    arrays -- and the inputs the three planted fixtures were generated on are exactly what the generator produces today (the GPU tests
    assert the same CRC before they run; a drift of the generator shows here, without a GPU)."""
    import numpy as np
    from conftest import load_golden
    from ggad_amd import synth
    n, f = 3000, 24
    rp, col = synth.make_graph(n, 40000, 5, kind="powerlaw", max_degree=n // 8)
    feat = synth.make_features(n, f, 5)
    y = synth.make_labels(n, 0.05, 5)
    for kw in (dict(scale=0.25, rewire=0.5), dict(scale=0.5, dims=0.5, rewire=0.0, max_degree=2), dict(scale=1.0, rewire=1.0, shift=0.3)):
        rp2, col2, f2 = synth.plant_anomalies(rp, col, feat, y, 5, **kw)
        a = synth.csr_to_scipy(rp2, col2, n)
        assert (a != a.T).nnz == 0 and a.diagonal().sum() == 0 and int(a.data.max()) == 1          # symmetric, no self loops, no duplicates
        assert int(np.diff(rp2).min()) >= 1
        assert np.array_equal(f2[y == 0], feat[y == 0]) and not np.array_equal(f2[y == 1], feat[y == 1])
        if kw.get("max_degree"):
            deg = np.diff(rp2)
            assert int(deg[y == 1].max()) <= kw["max_degree"] + 1                                   # (+1: the edge a node without any is given)
        again = synth.plant_anomalies(rp, col, feat, y, 5, **kw)
        assert all(np.array_equal(u, v) for u, v in zip((rp2, col2, f2), again))
    # the fixtures' inputs
    for name, rate, maxdeg_of in (("fullgraph_long_planted.npz", 0.06, None), ("fullgraph_long_planted_100.npz", 0.06, None),
                                  ("handler_dgraph_like_planted.npz", 0.02, 200)):
        g = load_golden(name)
        seed = int(g["seed"]) if "seed" in g else int(g["graph_seed"])
        nn = int(g["n"])
        rp0, c0 = synth.make_graph(nn, int(g["n_entries"]), seed, kind="powerlaw", max_degree=(maxdeg_of or nn // 8))
        ft = synth.make_features(nn, int(g["f"]), seed)
        lab = synth.make_labels(nn, rate, seed)
        kw = {k[len("planted."):]: float(g[k]) for k in g if k.startswith("planted.")}
        if "max_degree" in kw:
            kw["max_degree"] = int(kw["max_degree"])
        assert kw, name
        rp1, c1, ft1 = synth.plant_anomalies(rp0, c0, ft, lab, seed, **kw)
        assert synth.crc_of(rp1, c1, ft1, lab) == int(g["inputs_crc"]), name
