"""Full-graph GGAD at BASELINE's full sizes (Amazon 11,944 nodes / 4,398,392 entries, Photo 7,535 / 119,043 / 745 features,
T-Finance 39,357 / 21,222,543; Reddit for completeness): one whole training step of `run.py:142-214` -- `Model.forward`, the loss
block, backward, Adam -- through the HIP kernels against the CPU oracle's sparse restatement (`oracle/ggad_oracle.py`:
`full_forward`, `full_loss(by_column=True)`, torch autograd, torch Adam) on the same synthetic inputs, plus bit-determinism of
everything.  The reference's dense N x N formulation cannot run at T-Finance size (6.2 GB per N x N matrix); the oracle's sparse
forms are pinned against the reference's vectors on small graphs (tests/test_oracle_golden.py).

Tolerances: sums over hundreds of neighbours in a different order than torch's CSR kernels -> 2e-5 relative to the tensor's
largest magnitude on embeddings and gradients, 2e-5 absolute on the four loss terms, 3e-6 on the weights after one Adam step
(lr 1e-3: an Adam step is +-lr whatever the gradient's magnitude, so a sign flip of a near-zero gradient is the only way to differ
more -- excluded by masking gradient entries below 1e-6 of the tensor's scale)."""
import random
import types

import numpy as np
import pytest
import torch

from ggad_amd import synth
from ggad_amd.fullgraph_bench import SIZES, build_model, make_dataset

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ggad_amd import fullgraph as FG
from oracle import ggad_oracle as O

DEV = "cuda:0"


@pytest.mark.parametrize("name", ["reddit", "photo", "Amazon", "t_finance"])
def test_exact_published_sizes(name):
    n, ne, f, rate = SIZES[name]
    rowptr, col = synth.make_graph(n, ne, 0, kind="powerlaw", max_degree=max(64, n // 8), exact=True)
    assert len(rowptr) == n + 1 and abs(int(rowptr[-1]) - ne) <= 1                # published directed-entry count (up to parity)
    a = synth.csr_to_scipy(rowptr, col, n)
    assert abs(a - a.T).nnz == 0 and a.diagonal().sum() == 0 and np.diff(rowptr).min() >= 1


def _gpu_step(ds, full, model, opt, feats, seed_noise):
    args = types.SimpleNamespace(mean=ds["mean"], var=ds["var"])
    abn, nrm = ds["abn_idx"], ds["normal_idx"]
    ls = full.loss_structs(nrm, abn)
    model.train()
    opt.zero_grad()
    torch.manual_seed(seed_noise)
    emb, emb_combine, logits, emb_con, emb_abnormal = model(feats, full, abn, nrm, True, args)
    total, l_margin, l_bce, l_rec = FG.GgadLossFn.apply(emb[0], logits[0, :, 0], emb_con, emb_abnormal[0], full, ls, 0.7)
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    return dict(losses=np.array([total.item(), l_margin.item(), l_bce.item(), l_rec.item()]), emb=emb[0].detach().clone(),
                logits=logits[0, :, 0].detach().clone(), emb_con=emb_con.detach().clone(), grads=grads,
                weights={k: v.detach().clone() for k, v in model.state_dict().items()})


@pytest.mark.parametrize("name", ["reddit", "photo", "Amazon", "t_finance"])
def test_training_step_at_full_size_against_the_oracle(name):
    random.seed(0)
    np.random.seed(0)
    ds = make_dataset(name, 0)
    n, h = ds["n"], 300
    assert ds["adj"].shape == (n, n) and abs(int(ds["adj"].nnz) - SIZES[name][1]) <= 1
    full, model, opt, feats = build_model(ds, DEV, h, seed=0)
    init = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    got = _gpu_step(ds, full, model, opt, feats, seed_noise=1000)

    # ---- the oracle on the host: same initial weights, same noise draw, CSR forms of both adjacency matrices
    an = full.A.host
    rw = full.raw_host.tocsr()
    rw.sort_indices()
    adjn = (an.indptr, an.indices, an.data.astype(np.float32))
    raw = (rw.indptr, rw.indices, rw.data.astype(np.float32))
    P = {k: init[k].clone().requires_grad_() for k in O.FULL_PARAM_ORDER}
    adam = O.make_adam(list(P.values()), 1e-3, 0.0)
    abn, nrm = ds["abn_idx"], ds["normal_idx"]
    torch.manual_seed(1000)
    noise = torch.randn(1, len(abn), h)[0] * ds["var"] + ds["mean"]
    emb, comb, logits, con, eab = O.full_forward(P, torch.from_numpy(ds["features"]), adjn, abn, nrm, noise, True)
    total, lm, lb, lr, aff = O.full_loss(emb, logits, con, eab, raw, abn, nrm, by_column=True)
    total.backward()
    ref_losses = np.array([total.item(), lm.item(), lb.item(), lr.item()])
    np.testing.assert_allclose(got["losses"], ref_losses, atol=2e-5, rtol=2e-5)

    def close(a, b, what, tol=2e-5):
        a, b = a.cpu().numpy(), b.detach().numpy()
        scale = np.abs(b).max() + 1e-12
        assert np.abs(a - b).max() / scale < tol, (name, what, float(np.abs(a - b).max() / scale))
    close(got["emb"], emb, "emb")
    close(got["logits"], logits, "logits")
    close(got["emb_con"], con, "emb_con")
    for k in O.FULL_PARAM_ORDER:
        close(got["grads"][k], P[k].grad, "grad " + k, tol=1e-4)
    assert sorted(got["grads"].keys()) == sorted(O.FULL_PARAM_ORDER)              # gcn3 / fc5 / fc6 / disc: no gradient
    adam.step()
    for k in O.FULL_PARAM_ORDER:
        g = P[k].grad.numpy()
        sure = np.abs(g) > 1e-6 * (np.abs(g).max() + 1e-30)                       # away from the sign flip of a ~0 gradient
        d = np.abs(got["weights"][k].cpu().numpy() - P[k].detach().numpy())
        assert d[sure].max() < 3e-6, (name, k, float(d[sure].max()))
        assert d.max() < 2.1e-3                                                   # at most one opposite Adam step
    for k, v in got["weights"].items():
        if k not in O.FULL_PARAM_ORDER:
            assert torch.equal(v.cpu(), init[k]), k                               # parameters without gradient are not touched

    # ---- bit-determinism: the same step from the same state gives the same bits everywhere
    full2, model2, opt2, feats2 = build_model(ds, DEV, h, seed=0)
    again = _gpu_step(ds, full2, model2, opt2, feats2, seed_noise=1000)
    assert np.array_equal(got["losses"], again["losses"])
    assert torch.equal(got["emb"].view(torch.int32), again["emb"].view(torch.int32))
    for k in got["grads"]:
        assert torch.equal(got["grads"][k].view(torch.int32), again["grads"][k].view(torch.int32)), k
    for k in got["weights"]:
        assert torch.equal(got["weights"][k], again["weights"][k]), k
