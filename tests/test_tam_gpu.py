"""TAM comparison model on the MI355X (`ggad_amd/model_tam.py`, `ggad_amd/tam_utils.py`) against the vectors captured from the
imported reference (`tests/golden/fullgraph_tam.npz`): distances, forward, affinity, loss, gradients, the k-step trajectory with
the reference's once-per-round `zero_grad`, final weights, scores -- eager and captured."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(HERE, "golden", "fullgraph_tam.npz"))


def _raw(g):
    n = int(g["n"])
    a = sp.csr_matrix((np.ones(len(g["col"]), np.float32), g["col"], g["rowptr"]), shape=(n, n))
    r = (a + sp.eye(n)).tocsr()
    r.sort_indices()
    return r


def _cut_adj(g, cut, raw, dev):
    from ggad_amd.fullgraph import FullGraphAdj
    from ggad_amd import tam_utils as T
    n = int(g["n"])
    nz = g[f"cut{cut}.adj_nz"]
    pat = sp.csr_matrix((np.ones(len(nz), np.float32), (nz[:, 0], nz[:, 1])), shape=(n, n))
    pat.sort_indices()
    return FullGraphAdj(T.normalize_adj_tensor(pat), raw, dev)


def _model(g, cut, dev):
    from ggad_amd.model_tam import Model
    m = Model(int(g["f"]), int(g["n_h"]), "prelu", 2, "avg").to(dev)
    pre = f"init{cut}."
    m.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)})
    return m


def test_edge_distances(g):
    from ggad_amd import tam_utils as T
    dev = torch.device("cuda:0")
    d = T.calc_distance(_raw(g), torch.from_numpy(g["features"]).to(dev))
    np.testing.assert_allclose(d, g["dis_array_nz"], atol=2e-7, rtol=1e-6)


@pytest.mark.parametrize("cut", [0, 1])
def test_forward_affinity_loss_gradients(g, cut):
    from ggad_amd import tam_utils as T
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    adj = _cut_adj(g, cut, _raw(g), dev)
    model = _model(g, cut, dev)
    feats = torch.from_numpy(g["features"])[None].to(dev)
    emb, f1, f2 = model.forward(feats, adj)
    np.testing.assert_allclose(emb[0].detach().cpu().numpy(), g[f"cut{cut}.emb"], atol=3e-6)
    np.testing.assert_allclose(f1[0].detach().cpu().numpy(), g[f"cut{cut}.feat1"], atol=3e-6)
    np.testing.assert_allclose(f2[0].detach().cpu().numpy(), g[f"cut{cut}.feat2"], atol=3e-6)
    loss, m = T.max_message(emb[0], adj, g["normal_idx"])
    np.testing.assert_allclose(m.detach().cpu().numpy(), g[f"cut{cut}.message_norm"], atol=5e-6)
    np.testing.assert_allclose(T.inference(emb[0].detach(), adj).cpu().numpy(), g[f"cut{cut}.message"], atol=3e-6)
    assert abs(loss.item() - g[f"cut{cut}.losses"][0]) < 2e-4
    loss.backward()
    for k, p in model.named_parameters():
        gk = f"cut{cut}.grad." + k
        if gk in g.files:
            ref = g[gk]
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, atol=3e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
        else:
            assert p.grad is None, k                     # fc1 / fc2 / Model.act are outside the loss, as in the reference


@pytest.mark.parametrize("use_graph", [False, True])
def test_trajectory_with_accumulating_gradients_and_scores(g, use_graph):
    """Both truncation rounds of the captured run: k steps of Adam (lr as captured) WITHOUT clearing the gradients between the
    epochs (`tam.py:182`), losses, final weights, last messages and the AUROC / AP of the averaged score."""
    from ggad_amd import tam_utils as T
    from ggad_amd.fullgraph import FlatAdam
    from ggad_amd.metrics import average_precision, roc_auc
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    raw = _raw(g)
    feats = torch.from_numpy(g["features"])[None].to(dev)
    k_steps = len(g["cut0.losses"])
    msgs = []
    for cut in range(2):
        adj = _cut_adj(g, cut, raw, dev)
        model = _model(g, cut, dev)
        opt = FlatAdam(model.parameters(), lr=float(g["lr"]), weight_decay=0.0)
        opt.zero_grad()
        losses, msg = T.train_cut(model, opt, feats, adj, g["normal_idx"], k_steps, use_graph=use_graph)
        np.testing.assert_allclose(losses.cpu().numpy(), g[f"cut{cut}.losses"], atol=5e-4)
        np.testing.assert_allclose(msg.cpu().numpy(), g[f"cut{cut}.message_last"], atol=1e-5)
        sd = model.state_dict()
        for k in sd:
            np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"cut{cut}.final." + k], atol=3e-5, err_msg=k)
        msgs.append(msg.detach())
    mean_msg = torch.stack(msgs).mean(0)
    score = 1 - (mean_msg - mean_msg.min()) / (mean_msg.max() - mean_msg.min())
    np.testing.assert_allclose(score.cpu().numpy(), g["score"], atol=2e-5)
    y = torch.from_numpy(g["ano"].astype(np.int64)).to(dev)
    assert abs(roc_auc(score, y) - float(g["auc"])) < 1e-6
    assert abs(average_precision(score, y) - float(g["ap"])) < 1e-6
