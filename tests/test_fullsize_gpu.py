"""The mini-batch hot path at BASELINE.json's FULL size (DGraph-Fin: 3,700,550 nodes, 73.1 M directed entries, batches of
150 + 50, chunks of 150 batches -- exactly what `bench.py` times), checked through what stays cheap at that size:

  * the CPU oracle on sampled batches of the full-size chunk (one batch touches ~4 K entries and ~90 K 2-hop pairs, so the
    restatement still finishes in a second) -- plan, both aggregates, loss and gradients;
  * size-independent properties of the whole 150-batch chunk: exact linearity under a power-of-two scaling of the feature
    table, independence of the batches (a chunk of one batch gives the same rows), determinism of a rebuild and of 150
    optimiser steps, the validation sweep against per-batch scoring.
"""
import random

import numpy as np
import pytest
import torch

from ggad_amd import synth
from oracle import ggad_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ggad_amd.dgraph import normalize_features, split_dgraphfin
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.minibatch import BatchChunk, MiniBatchEngine
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule

DEV = "cuda:0"
N, ENTRIES, F, D = 3700550, 73105508, 17, 64


@pytest.fixture(scope="module")
def full():
    rowptr, col = synth.make_graph_torch(N, ENTRIES, 72, DEV, kind="powerlaw", max_degree=2000)
    graph = DeviceGraph(rowptr, col, DEV)
    feat_np = normalize_features(synth.make_features(N, F, 72)).astype(np.float32)
    labels0 = synth.make_labels(N, 15509.0 / 3700550.0, 72).astype(np.int32)
    split = split_dgraphfin(labels0, 72, with_test=False)                  # src/model_handler.py:150-178 at full size
    sched = BatchSchedule(split["idx_train"], split["idx_anomaly"], split["labels"], 150,
                          PyCompatRandom.from_python_state(random.getstate()))
    batches, labels = sched.next_batches(150, 0, 1)                        # one epoch of the reference's schedule
    table = torch.zeros(N, 32, dtype=torch.float32, device=DEV)            # 128-byte rows, as the trainer lays them out
    table[:, :F] = torch.from_numpy(feat_np).to(DEV)
    ch = BatchChunk(graph, table, D, max_batches=150, rows_cap=150 * 200, ent_cap=1 << 20, train=True, feat_dim=F, hop2="ldsw")
    ch.build(batches, labels)
    torch.cuda.synchronize()
    assert ch.last_hop2 == "ldsw"                                           # the default path, not a fallback
    return dict(graph=graph, rowptr=np.asarray(rowptr), col=np.asarray(col), feat=feat_np, table=table, batches=batches,
                labels=labels, ch=ch, split=split)


def _owners_of_batch(ch, b):
    e0, e1 = ch.batch_ents(b)
    own = torch.unique(ch.ent_own[e0:e1].long())
    return e0, e1, own


def test_sampled_batches_of_the_full_size_chunk_against_the_oracle(full):
    ch, rp, ci, feat = full["ch"], full["rowptr"], full["col"], full["feat"]
    assert len(full["batches"]) == 150 and all(len(b) == 200 for b in full["batches"])
    ent_ptr = ch.ent_ptr[:ch.n_rows + 1].cpu().numpy()
    assert np.array_equal(ent_ptr, ch.ent_ptr_host)
    torch.manual_seed(3)
    w = torch.nn.init.xavier_uniform_(torch.empty(1, D))
    W = torch.nn.init.xavier_uniform_(torch.empty(D, F))
    fc = torch.nn.init.xavier_uniform_(torch.empty(D, D))
    eng = MiniBatchEngine(F, D, DEV, lr=1e-3, weight_decay=0.007)
    eng.load_params(w, W, fc)
    x1 = ch.x1[:ch.n_rows * F].view(-1, F)
    x2 = ch.x2[:ch.n_ents * F].view(-1, F)
    for slot, b in enumerate((0, 61, 149)):
        nodes, lab = full["batches"][b], full["labels"][b]
        agg = O.aggregate_batch(rp, ci, feat, nodes, True)
        r0, r1 = ch.batch_rows(b)
        e0, e1, own = _owners_of_batch(ch, b)
        assert e1 - e0 == int(agg.ent_ptr[-1]) and np.array_equal(ent_ptr[r0:r1 + 1] - ent_ptr[r0], agg.ent_ptr)
        assert np.array_equal(ch.ent_col[e0:e1].cpu().numpy(), agg.unique[agg.ent_pos])
        np.testing.assert_allclose(x1[r0:r1].cpu().numpy(), agg.to_feats, atol=3e-6, rtol=0)
        assert len(own) == len(agg.unique)                                  # one owner per distinct column of the batch
        pos = np.searchsorted(agg.unique, ch.ent_col[own].cpu().numpy())
        np.testing.assert_allclose(x2[own].cpu().numpy(), agg.to_feats_neigh[pos], atol=2e-5, rtol=0, equal_nan=True)
        # loss and gradients of that batch (forward, 4 loss terms, backward through both aggregates' projections)
        p = O.MiniParams(w.clone().requires_grad_(), W.clone().requires_grad_(), fc.clone().requires_grad_())
        tot, cls, mar, rec = O.batch_loss(p, agg, lab)
        tot.backward()
        eng.loss_and_grads(ch, b, slot)
        ref = np.concatenate([t.grad.numpy().reshape(-1) for t in p.tensors()])
        np.testing.assert_allclose(eng.grads.cpu().numpy(), ref, atol=5e-6, rtol=5e-5)
        np.testing.assert_allclose(eng.losses(slot + 1)[slot], [tot.item(), cls.item(), mar.item(), rec.item()], atol=2e-5)


def test_power_of_two_scaling_is_exact_over_the_whole_chunk(full):
    """Both aggregates are linear in the feature table and a factor 2 commutes with every fp32 rounding: the plan of the
    same 150 batches on 2 X must give exactly 2 x1 and 2 x2, element for element (any dropped / doubled pair would show)."""
    ch = full["ch"]
    ch2 = BatchChunk(full["graph"], full["table"] * 2.0, D, max_batches=150, rows_cap=150 * 200, ent_cap=ch.ent_cap, train=True,
                     feat_dim=F, hop2="ldsw")
    ch2.build(full["batches"], full["labels"])
    torch.cuda.synchronize()
    assert ch2.n_ents == ch.n_ents and ch2.last_hop2 == "ldsw"
    a1, b1 = ch.x1[:ch.n_rows * F], ch2.x1[:ch.n_rows * F]
    assert torch.equal((a1 * 2.0).view(torch.int32), b1.view(torch.int32))
    n_own = int(ch.owner_entries().numel())
    assert n_own == int(ch2.owner_entries().numel()) and n_own > 500000         # ~4 K distinct columns per batch
    # owner election is a race between duplicate entries of a batch: compare per (batch, column)
    def keyed(c):
        own = c.owner_entries()
        bnd = torch.as_tensor(c.ent_ptr_host[c.batch_ptr_host][1:], device=DEV)
        k = torch.bucketize(own, bnd, right=True) * N + c.ent_col[own].long()
        o = torch.argsort(k)
        return k[o], c.x2.view(-1, F)[own][o]
    ka, xa = keyed(ch)
    kb, xb = keyed(ch2)
    assert torch.equal(ka, kb)
    same = (xa * 2.0).view(torch.int32) == xb.view(torch.int32)
    assert bool((same | (torch.isnan(xa) & torch.isnan(xb))).all())


def test_batches_are_independent_and_rebuilds_deterministic(full):
    ch = full["ch"]
    x1_all = ch.x1[:ch.n_rows * F].view(-1, F).clone()
    one = BatchChunk(full["graph"], full["table"], D, max_batches=1, rows_cap=256, ent_cap=8192, train=True, feat_dim=F, hop2="ldsw")
    for b in (7, 149):
        one.build([full["batches"][b]], [full["labels"][b]])
        torch.cuda.synchronize()
        r0, r1 = ch.batch_rows(b)
        assert torch.equal(one.x1[:200 * F].view(-1, F).view(torch.int32), x1_all[r0:r1].view(torch.int32))
        e0, e1, own = _owners_of_batch(ch, b)
        _, _, own1 = _owners_of_batch(one, 0)
        ca, cb = ch.ent_col[own], one.ent_col[own1]
        oa, ob = torch.argsort(ca), torch.argsort(cb)
        assert torch.equal(ca[oa], cb[ob])
        xa, xb = ch.x2.view(-1, F)[own][oa], one.x2.view(-1, F)[own1][ob]
        # same pairs, same weights; the node-major gather may add an owner's neighbours in another order than a 1-batch plan
        assert torch.allclose(xa, xb, rtol=2e-6, atol=1e-7, equal_nan=True)
    # rebuild of the same chunk: x1 bit-identical
    ch.build(full["batches"], full["labels"])
    torch.cuda.synchronize()
    assert torch.equal(ch.x1[:ch.n_rows * F].view(-1, F).view(torch.int32), x1_all.view(torch.int32))


def test_one_epoch_of_steps_is_deterministic_and_finite(full):
    ch = full["ch"]
    torch.manual_seed(5)
    w = torch.nn.init.xavier_uniform_(torch.empty(1, D))
    W = torch.nn.init.xavier_uniform_(torch.empty(D, F))
    fc = torch.nn.init.xavier_uniform_(torch.empty(D, D))
    runs = []
    for _ in range(2):
        eng = MiniBatchEngine(F, D, DEV, lr=1e-3, weight_decay=0.007)
        eng.load_params(w, W, fc)
        eng.train_chunk(ch)                                                  # the C-side loop over the 150 batches
        torch.cuda.synchronize()
        runs.append((eng.losses(150).copy(), eng.params.cpu().numpy().copy()))
    assert np.isfinite(runs[0][0]).all() and np.isfinite(runs[0][1]).all()
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert runs[0][0][:, 0].mean() < 3.0 and not np.array_equal(runs[0][1], np.concatenate([t.numpy().reshape(-1) for t in (w, W, fc)]))


def test_validation_sweep_slices_equal_single_batch_scoring(full):
    """`test_sage` scores consecutive slices of 150 nodes, each normalised on its own (src/utils.py:216-230): a sweep of
    30,000 full-size test nodes planned 145 slices at a time must give, for any slice, the oracle's `to_prob` of that slice."""
    from ggad_amd.graphsage import GCN, FeatureTable, GCNAggregator, GCNEncoder
    from ggad_amd.sage_utils import score_nodes
    features = FeatureTable(torch.from_numpy(full["feat"]))
    enc = GCNEncoder(features, F, D, full["graph"], GCNAggregator(features, cuda=True), gcn=True, cuda=True)
    model = GCN(2, enc)
    rng = np.random.default_rng(9)
    cases = rng.permutation(N)[:30000 + 77]                                  # ragged tail
    probs = score_nodes(model, cases, 150)
    assert probs.shape == (len(cases),) and np.isfinite(probs).all()
    p = O.MiniParams(model.weight.detach().cpu(), enc.weight.detach().cpu(), enc.fc.weight.detach().cpu())
    for s in (0, 150 * 101, 150 * 200):
        part = cases[s:s + 150]
        ref = O.to_prob(p, full["rowptr"], full["col"], full["feat"], part)
        np.testing.assert_allclose(probs[s:s + len(part)], ref.reshape(-1), atol=2e-6, rtol=0)
