#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *imported* reference.

Runs ONLY in the build container, where the reference tree is mounted
read-only at /root/reference.  Nothing of the reference travels: this script
imports its Python modules (with stub modules for the two absent, unused
third-party imports `dgl` and `torch_geometric`), feeds them
seeded synthetic inputs from ``ggad_amd.synth`` and stores inputs + every
returned tensor as small ``.npz`` files.  The fixtures are data only.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Two interpreter passes are needed because the reference has same-named
modules (`model.py`, `utils.py`) at its root and under `src/`; the script
re-executes itself with ``--part full`` / ``--part mini``.
"""
from __future__ import annotations

import argparse
import os
import random
import subprocess
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ggad_amd import synth  # noqa: E402


def _stub_third_party():
    dgl = types.ModuleType("dgl")
    sys.modules["dgl"] = dgl
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    tgnn.GCNConv = object
    tg.nn = tgnn
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tgnn


def _np(t):
    if t is None:
        return np.zeros((0,), dtype=np.float32)
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


# --------------------------------------------------------------------------
# Part 1: full-graph GGAD (reference root: model.py, utils.py; loop of run.py)
# --------------------------------------------------------------------------
def full_graph_case(tag, n, n_entries, f, n_h, seed, mean, var, kind, k_steps,
                    outlier_rate=0.15, self_loop_frac=0.0):
    import scipy.sparse as sp
    from model import Model                      # /root/reference/model.py
    import utils as rutils                       # /root/reference/utils.py

    rowptr, col = synth.make_graph(n, n_entries, seed, kind=kind, max_degree=n // 4,
                                   self_loop_frac=self_loop_frac)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.06, seed)
    adj_sp = synth.csr_to_scipy(rowptr, col)

    # label bookkeeping exactly as utils.load_mat does it (utils.py:95-140), driven by python `random`
    random.seed(seed)
    all_idx = list(range(n))
    random.shuffle(all_idx)
    num_train, num_val = int(n * 0.3), int(n * 0.1)
    idx_train = all_idx[:num_train]
    idx_test = all_idx[num_train + num_val:]
    all_normal = [i for i in idx_train if ano[i] == 0]
    normal_idx = all_normal[: int(len(all_normal) * 0.5)]
    random.shuffle(normal_idx)
    abn_idx = normal_idx[: int(len(normal_idx) * outlier_rate)]

    feats_dense, _ = rutils.preprocess_features(sp.lil_matrix(feat))   # utils.py:37-44
    adj_norm = rutils.normalize_adj(adj_sp)                             # utils.py:47-54
    raw_dense = (adj_sp + sp.eye(n)).todense()                          # run.py:100
    adj_dense = (adj_norm + sp.eye(n)).todense()                        # run.py:101
    features = torch.FloatTensor(np.asarray(feats_dense)[np.newaxis])
    adj = torch.FloatTensor(np.asarray(adj_dense)[np.newaxis])
    raw_adj = torch.FloatTensor(np.asarray(raw_dense)[np.newaxis])

    torch.manual_seed(seed)
    model = Model(f, n_h, "prelu", 1, "avg")                             # run.py:117
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
    init_state = {k: _np(v).copy() for k, v in model.state_dict().items()}
    args = types.SimpleNamespace(mean=mean, var=var)
    bce = torch.nn.BCEWithLogitsLoss(reduction="none", pos_weight=torch.tensor([1]))

    out = dict(n=n, f=f, n_h=n_h, seed=seed, mean=mean, var=var,
               rowptr=rowptr, col=col, feat_raw=feat, ano=ano,
               inputs_crc=synth.crc_of(rowptr, col, feat, ano),
               features=_np(features[0]), idx_test=np.array(idx_test),
               normal_idx=np.array(normal_idx), abn_idx=np.array(abn_idx))
    # normalised adjacency in COO (values are fp64 from scipy, cast to fp32 by run.py:103-109)
    an = (adj_norm + sp.eye(n)).tocoo()
    out.update(adjn_row=an.row.astype(np.int32), adjn_col=an.col.astype(np.int32),
               adjn_val=an.data.astype(np.float64))
    for k, v in init_state.items():
        out["init." + k] = v

    def loss_block(emb, logits, emb_con, emb_abnormal):
        # dense restatement of the reference's inline loss (run.py:165-210)
        lbl = torch.cat((torch.zeros(len(normal_idx)), torch.ones(len(emb_con)))).unsqueeze(1).unsqueeze(0)
        l_bce = torch.mean(bce(logits, lbl))
        e = torch.squeeze(emb)
        inv = torch.pow(torch.norm(e, dim=-1, keepdim=True), -1)
        inv[torch.isinf(inv)] = 0.0
        en = e * inv
        sim = torch.mm(en, en.T) * torch.squeeze(raw_adj)
        r_inv = torch.pow(torch.sum(torch.squeeze(raw_adj), 0), -1)
        r_inv[torch.isinf(r_inv)] = 0.0
        aff = torch.sum(sim, 0) * r_inv
        l_margin = (0.7 - (torch.mean(aff[normal_idx]) - torch.mean(aff[abn_idx]))).clamp_min(min=0)
        l_rec = torch.mean(torch.sqrt(torch.sum(torch.pow(emb_con - emb_abnormal, 2), 1)))
        return l_margin + l_bce + l_rec, l_margin, l_bce, l_rec, aff

    losses = []
    for step in range(k_steps):
        model.train()
        opt.zero_grad()
        torch.manual_seed(1000 + step)     # pins the CPU-generator noise of model.py:143
        emb, emb_combine, logits, emb_con, emb_abnormal = model(
            features, adj, abn_idx, normal_idx, True, args)
        total, l_margin, l_bce, l_rec, aff = loss_block(emb, logits, emb_con, emb_abnormal)
        total.backward()
        if step == 0:
            out.update(emb=_np(emb[0]), emb_combine=_np(emb_combine[0]), logits=_np(logits[0, :, 0]),
                       emb_con=_np(emb_con), emb_abnormal=_np(emb_abnormal[0]), affinity=_np(aff))
            for k, p in model.named_parameters():
                if p.grad is not None:
                    out["grad." + k] = _np(p.grad).copy()
        losses.append([total.item(), l_margin.item(), l_bce.item(), l_rec.item()])
        opt.step()
        if step == 0:
            for k, v in model.state_dict().items():
                out["step1." + k] = _np(v).copy()
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in model.state_dict().items():
        out["final." + k] = _np(v).copy()

    # eval forward (run.py:231-239): train_flag False, still draws noise (quirk 5)
    model.eval()
    torch.manual_seed(5000)
    with torch.no_grad():
        _, _, logits_eval, _, _ = model(features, adj, abn_idx, normal_idx, False, args)
    from sklearn.metrics import roc_auc_score, average_precision_score
    le = _np(logits_eval[0, :, 0])
    out["eval_logits"] = le
    yt = ano[np.array(idx_test)]
    out["eval_auc"] = roc_auc_score(yt, le[np.array(idx_test)])
    out["eval_ap"] = average_precision_score(yt, le[np.array(idx_test)], average="macro", pos_label=1)
    path = os.path.join(HERE, f"fullgraph_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "losses[0]", losses[0], "auc", out["eval_auc"])


def ocgnn_case():
    """Full-graph OCGNN comparison model: the imported `model_ocgnn.Model` driven by the training block of `ocgnn.py`
    (`:83-118` loss, `:170-186` step; the script itself needs dgl and a dataset file, so its loop is restated here)."""
    import scipy.sparse as sp
    from model_ocgnn import Model                # /root/reference/model_ocgnn.py
    import utils as rutils                       # /root/reference/utils.py
    n, n_entries, f, n_h, seed, k_steps = 300, 2400, 20, 48, 4, 5
    rowptr, col = synth.make_graph(n, n_entries, seed, kind="powerlaw", max_degree=n // 4, self_loop_frac=0.05)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.06, seed)
    adj_sp = synth.csr_to_scipy(rowptr, col)
    random.seed(seed)
    all_idx = list(range(n))
    random.shuffle(all_idx)
    idx_train, idx_test = all_idx[:int(n * 0.3)], all_idx[int(n * 0.3) + int(n * 0.1):]
    all_normal = [i for i in idx_train if ano[i] == 0]
    normal_idx = all_normal[: int(len(all_normal) * 0.5)]
    feats_dense, _ = rutils.preprocess_features(sp.lil_matrix(feat))
    adj_norm = rutils.normalize_adj(adj_sp)
    features = torch.FloatTensor(np.asarray(feats_dense)[np.newaxis])
    adj = torch.FloatTensor(np.asarray((adj_norm + sp.eye(n)).todense())[np.newaxis])
    torch.manual_seed(seed)
    model = Model(f, n_h, "prelu", 1, "avg")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
    out = dict(n=n, f=f, n_h=n_h, seed=seed, rowptr=rowptr, col=col, feat_raw=feat, ano=ano,
               inputs_crc=synth.crc_of(rowptr, col, feat, ano), features=_np(features[0]),
               idx_test=np.array(idx_test), normal_idx=np.array(normal_idx))
    for k, v in model.state_dict().items():
        out["init." + k] = _np(v).copy()

    def loss_func(emb):                          # ocgnn.py:83-118 with its constants
        r, beta = 0, 0.5
        c = torch.zeros(n_h)
        dist = torch.sum(torch.pow(emb - c, 2), 1)
        score = dist - r ** 2
        return r ** 2 + 1 / beta * torch.mean(torch.relu(score)), score

    losses = []
    for step in range(k_steps):
        model.train()
        opt.zero_grad()
        emb = model(features, adj)
        loss, score = loss_func(torch.squeeze(emb)[normal_idx])
        loss.backward()
        if step == 0:
            out.update(emb=_np(emb[0]), score=_np(score))
            for k, p in model.named_parameters():
                if p.grad is not None:
                    out["grad." + k] = _np(p.grad).copy()
        losses.append(loss.item())
        opt.step()
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in model.state_dict().items():
        out["final." + k] = _np(v).copy()
    model.eval()
    with torch.no_grad():
        _, score = loss_func(torch.squeeze(model(features, adj)))
    from sklearn.metrics import roc_auc_score, average_precision_score
    sc = _np(score)
    yt = ano[np.array(idx_test)]
    out.update(eval_score=sc, eval_auc=roc_auc_score(yt, sc[np.array(idx_test)]),
               eval_ap=average_precision_score(yt, sc[np.array(idx_test)], average="macro", pos_label=1))
    path = os.path.join(HERE, "fullgraph_ocgnn.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "losses", losses, "auc", out["eval_auc"])


def tam_case():
    """Full-graph TAM comparison model: the imported `model_tam.Model` / `utils_tam` functions and the two functions `tam.py`
    defines (`max_message`, `inference`; pulled out of the script by name with `ast`, because importing `tam.py` runs it),
    driven by the loop of `tam.py:153-214` on a small graph: distances, two successive truncations, normalisation, forward,
    loss, gradients, a k-step Adam trajectory, scores."""
    import ast
    import contextlib
    import io
    import tempfile
    import scipy.io as sio
    import scipy.sparse as sp
    import utils_tam as U                        # /root/reference/utils_tam.py
    from model_tam import Model                  # /root/reference/model_tam.py
    ns = {"torch": torch}
    tree = ast.parse(open(os.path.join(REF, "tam.py")).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("max_message", "inference"):
            exec(compile(ast.Module(body=[node], type_ignores=[]), "tam.py", "exec"), ns)
    max_message, inference = ns["max_message"], ns["inference"]
    n, n_entries, f, n_h, seed, k_steps, lr = 300, 2400, 20, 32, 9, 5, 1e-3
    rowptr, col = synth.make_graph(n, n_entries, seed, kind="powerlaw", max_degree=n // 4)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.08, seed)
    adj_sp = synth.csr_to_scipy(rowptr, col)
    out = dict(n=n, f=f, n_h=n_h, seed=seed, lr=lr, rowptr=rowptr, col=col, feat_raw=feat, ano=ano,
               inputs_crc=synth.crc_of(rowptr, col, feat, ano))
    tmp = tempfile.mkdtemp(prefix="ggad_tam_")
    os.makedirs(os.path.join(tmp, "data"))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        sio.savemat(os.path.join("data", "tiny.mat"), {"Network": sp.csr_matrix(adj_sp), "Attributes": sp.csr_matrix(feat),
                                                        "Label": ano.reshape(-1, 1)})
        random.seed(seed)
        adj, features, ano_label, _, _, normal_label_idx, idx_test = U.load_mat("tiny")
        out["split.tail"] = np.array([random.getrandbits(32) for _ in range(3)], dtype=np.int64)
    finally:
        os.chdir(cwd)
    out.update(normal_idx=np.array(normal_label_idx, dtype=np.int64), idx_test=np.array(idx_test, dtype=np.int64))
    features, _ = U.preprocess_features(features)                       # tam.py:55-57 (the 'Amazon' branch)
    raw_features = features
    raw_adj = (adj + sp.eye(n)).todense()
    raw_features_t = torch.FloatTensor(raw_features[np.newaxis])
    features_t = torch.FloatTensor(features[np.newaxis])
    raw_adj_t = torch.FloatTensor(raw_adj[np.newaxis])
    out["features"] = _np(features_t[0])
    with contextlib.redirect_stdout(io.StringIO()):
        dis_array = U.calc_distance(raw_adj_t[0], raw_features_t[0])   # tam.py:168
    out["dis_array_nz"] = _np(dis_array)[np.asarray(raw_adj) > 0]      # row-major over the stored entries of A + I
    np.random.seed(seed)
    all_cut = raw_adj_t.clone()
    torch.manual_seed(seed)
    models = [Model(f, n_h, "prelu", 2, "avg") for _ in range(2)]
    for k, v in models[0].state_dict().items():
        out["init0." + k] = _np(v).copy()
    for k, v in models[1].state_dict().items():
        out["init1." + k] = _np(v).copy()
    msgs = []
    for cut in range(2):
        cut_adj = U.graph_nsgt(dis_array, all_cut[0]).unsqueeze(0)      # tam.py:180 (in place on all_cut[0] too)
        out[f"cut{cut}.adj_nz"] = np.argwhere(_np(cut_adj[0]) > 0).astype(np.int32)
        adj_norm = U.normalize_adj_tensor(cut_adj)
        out[f"cut{cut}.adj_norm_vals"] = _np(adj_norm[0])[_np(cut_adj[0]) > 0]
        model = models[cut]
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
        opt.zero_grad()                                                 # ONCE per cut, as tam.py:182 (gradients accumulate)
        model.train()
        losses = []
        for epoch in range(k_steps):
            node_emb, feat1, feat2 = model.forward(features_t, adj_norm)
            loss, message_sum1 = max_message(node_emb[0], raw_adj_t[0], normal_label_idx)
            message_sum = inference(node_emb[0], raw_adj_t[0])
            loss.backward()
            if epoch == 0:
                out.update({f"cut{cut}.emb": _np(node_emb[0]), f"cut{cut}.feat1": _np(feat1[0]), f"cut{cut}.feat2": _np(feat2[0]),
                            f"cut{cut}.message_norm": _np(message_sum1), f"cut{cut}.message": _np(message_sum)})
                for kk, pp in model.named_parameters():
                    if pp.grad is not None:
                        out[f"cut{cut}.grad." + kk] = _np(pp.grad).copy()
            opt.step()
            losses.append(float(loss.detach()))
        out[f"cut{cut}.losses"] = np.array(losses, dtype=np.float64)
        out[f"cut{cut}.message_last"] = _np(message_sum)
        for k, v in model.state_dict().items():
            out[f"cut{cut}.final." + k] = _np(v).copy()
        msgs.append(message_sum.detach().unsqueeze(0))
        all_cut[0] = cut_adj[0]
    from sklearn.metrics import roc_auc_score, average_precision_score
    mean_msg = torch.mean(torch.cat(msgs), 0).numpy()
    score = 1 - U.normalize_score(mean_msg)
    out.update(score=score, auc=roc_auc_score(ano_label, score),
               ap=average_precision_score(ano_label, score, average="macro", pos_label=1),
               nprandom_tail=np.random.random_sample(3))
    path = os.path.join(HERE, "fullgraph_tam.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: out[k] for k in ("auc", "ap")}, "losses", out["cut0.losses"], out["cut1.losses"])


def ingest_case():
    """`load_mat` of the reference (utils.py:66-141) on a small .mat written from seeded synthetic data: the index lists the
    split produces (python `random` driven), for the two outlier-seed fractions ('Amazon' 5 %, every other name 15 %)."""
    import tempfile
    import scipy.io as sio
    import scipy.sparse as sp
    import utils as rutils                       # /root/reference/utils.py
    n, ne, f, seed = 400, 3000, 12, 31
    rowptr, col = synth.make_graph(n, ne, seed, kind="powerlaw", max_degree=60)
    adj = synth.csr_to_scipy(rowptr, col)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.08, seed)
    out = dict(n=n, n_entries=ne, f=f, seed=seed, inputs_crc=synth.crc_of(rowptr, col, feat, ano))
    tmp = tempfile.mkdtemp(prefix="ggad_ingest_")
    os.makedirs(os.path.join(tmp, "dataset"))
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        for name, keys in (("Amazon", ("Network", "Attributes", "Label")), ("tiny", ("A", "X", "gnd"))):
            sio.savemat(os.path.join("dataset", name + ".mat"),
                        {keys[0]: sp.csr_matrix(adj), keys[1]: sp.csr_matrix(feat), keys[2]: ano.reshape(-1, 1)})
            random.seed(seed)
            r = rutils.load_mat(name)
            a, ft = r[0], r[1]
            assert (abs(a - adj).nnz == 0) and np.allclose(np.asarray(ft.todense()), feat)
            out[name + ".all_idx"] = np.array(r[3], dtype=np.int64)
            out[name + ".idx_train"] = np.array(r[4], dtype=np.int64)
            out[name + ".idx_val"] = np.array(r[5], dtype=np.int64)
            out[name + ".idx_test"] = np.array(r[6], dtype=np.int64)
            out[name + ".normal_idx"] = np.array(r[10], dtype=np.int64)
            out[name + ".abnormal_idx"] = np.array(r[11], dtype=np.int64)
            out[name + ".tail"] = np.array([random.getrandbits(32) for _ in range(3)], dtype=np.int64)   # RNG consumption
    finally:
        os.chdir(cwd)
    path = os.path.join(HERE, "ingest_mat.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: len(v) for k, v in out.items() if hasattr(v, "__len__") and "idx" in k})



def full_graph_long_case(tag="long_photo_schedule", n=4200, n_entries=60000, f=64, n_h=300, seed=0, mean=0.02, var=0.01,
                         num_epoch=100, outlier_rate=0.15, planted=None, normalise=True, self_sensitivity=0.0):
    """End-of-training parity (BASELINE north_star: "AUROC/AUPRC within 1e-4"): the WHOLE training schedule of the reference's
    script for `--dataset photo` (run.py:46-48: 100 epochs; README: --mean 0.02 --var 0.01; Adam lr 1e-3) on a graph the dense
    reference still runs here, restated around the imported `Model` exactly as run.py:137-240 drives it: ONE seeding at the
    start, the noise of every training forward AND of every evaluation forward (every 10th epoch, quirk 5) drawn from the same
    global CPU generator in program order.  Stored: the four loss terms of every epoch, AUROC / AP of every evaluation, and one
    extra evaluation after the last epoch with the scores of all nodes.  Inputs are regenerated from the seed (CRC stored);
    the initial weights are those of `torch.manual_seed(seed); Model(...)` (CRC stored).  N = 4200 rows and H = 300 send the
    projections of every epoch through the round-5 slab GEMM."""
    import scipy.sparse as sp
    from sklearn.metrics import average_precision_score, roc_auc_score
    from model import Model
    import utils as rutils

    rowptr, col = synth.make_graph(n, n_entries, seed, kind="powerlaw", max_degree=n // 8)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.06, seed)
    if planted:
        # round 6: labels that MEAN something -- the labelled nodes get attenuated features and a neighbourhood rewired
        # towards each other (synth.plant_anomalies), the features stay raw as run.py keeps them for `--dataset photo` (run.py:87): the
        # reference then ends its 100 epochs at AUROC ~0.94 instead of ~0.49
        rowptr, col, feat = synth.plant_anomalies(rowptr, col, feat, ano, seed, **planted)
    adj_sp = synth.csr_to_scipy(rowptr, col)
    random.seed(seed)
    all_idx = list(range(n))
    random.shuffle(all_idx)
    num_train, num_val = int(n * 0.3), int(n * 0.1)
    idx_train = all_idx[:num_train]
    idx_test = all_idx[num_train + num_val:]
    all_normal = [i for i in idx_train if ano[i] == 0]
    normal_idx = all_normal[: int(len(all_normal) * 0.5)]
    random.shuffle(normal_idx)
    abn_idx = normal_idx[: int(len(normal_idx) * outlier_rate)]

    if normalise:
        feats_dense, _ = rutils.preprocess_features(sp.lil_matrix(feat))
    else:
        feats_dense = feat                                              # run.py:87-88: photo / T-Finance features are used as loaded
    adj_norm = rutils.normalize_adj(adj_sp)
    raw_adj = torch.FloatTensor(np.asarray((adj_sp + sp.eye(n)).todense()))
    adj = torch.FloatTensor(np.asarray((adj_norm + sp.eye(n)).todense())[np.newaxis])
    features = torch.FloatTensor(np.asarray(feats_dense)[np.newaxis])
    args = types.SimpleNamespace(mean=mean, var=var)
    bce = torch.nn.BCEWithLogitsLoss(reduction="none", pos_weight=torch.tensor([1]))

    r_inv = torch.pow(torch.sum(raw_adj, 0), -1)
    r_inv[torch.isinf(r_inv)] = 0.0
    yt = ano[np.array(idx_test)]

    def train(perturb=0.0, verbose=True):
        torch.manual_seed(seed)                                         # run.py:62 (the one seeding of the run)
        model = Model(f, n_h, "prelu", 1, "avg")
        init_crc = synth.crc_of(*[_np(v) for _, v in sorted(model.state_dict().items())])
        if perturb:
            # conditioning probe (round 6): the SAME run from initial weights moved by a relative N(0, perturb^2) -- how far the reference's
            # own end-of-training numbers move under a change of the size of one fp32 rounding.  Drawn from a private generator: the
            # global stream -- the noise draws of the run -- is where the unperturbed run has it
            gp = torch.Generator().manual_seed(999)
            with torch.no_grad():
                for p_ in model.parameters():
                    p_.mul_(1.0 + perturb * torch.randn(p_.shape, generator=gp))
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=0.0)
        losses, evals = [], []

        def evaluate():
            model.eval()
            with torch.no_grad():
                _, _, lg, _, _ = model(features, adj, abn_idx, normal_idx, False, args)
            le = _np(lg[0, :, 0])
            return le, roc_auc_score(yt, le[np.array(idx_test)]), average_precision_score(yt, le[np.array(idx_test)], average="macro", pos_label=1)

        import time as _t
        t0 = _t.time()
        for epoch in range(num_epoch):
            model.train()
            opt.zero_grad()
            emb, emb_combine, logits, emb_con, emb_abnormal = model(features, adj, abn_idx, normal_idx, True, args)
            lbl = torch.cat((torch.zeros(len(normal_idx)), torch.ones(len(emb_con)))).unsqueeze(1).unsqueeze(0)
            l_bce = torch.mean(bce(logits, lbl))
            e = torch.squeeze(emb)
            inv = torch.pow(torch.norm(e, dim=-1, keepdim=True), -1)
            inv[torch.isinf(inv)] = 0.0
            en = e * inv
            aff = torch.sum(torch.mm(en, en.T) * raw_adj, 0) * r_inv
            l_margin = (0.7 - (torch.mean(aff[normal_idx]) - torch.mean(aff[abn_idx]))).clamp_min(min=0)
            l_rec = torch.mean(torch.sqrt(torch.sum(torch.pow(emb_con - emb_abnormal, 2), 1)))
            total = l_margin + l_bce + l_rec
            total.backward()
            opt.step()
            losses.append([total.item(), l_margin.item(), l_bce.item(), l_rec.item()])
            if epoch % 10 == 0:
                _, auc, ap = evaluate()
                evals.append([epoch, auc, ap])
                if verbose:
                    print(f"epoch {epoch}: loss {losses[-1][0]:.6f} auc {auc:.6f} ap {ap:.6f}  ({_t.time() - t0:.0f} s)", flush=True)
        le, auc, ap = evaluate()
        return model, init_crc, np.array(losses, dtype=np.float64), np.array(evals, dtype=np.float64), le, auc, ap

    model, init_crc, losses, evals, le, auc, ap = train()
    out = dict(n=n, n_entries=n_entries, f=f, n_h=n_h, seed=seed, mean=mean, var=var, num_epoch=num_epoch,
               inputs_crc=synth.crc_of(rowptr, col, feat, ano), init_crc=init_crc,
               idx_test=np.array(idx_test), normal_idx=np.array(normal_idx), abn_idx=np.array(abn_idx),
               losses=losses, evals=evals, final_logits=le, final_auc=auc, final_ap=ap, normalise=int(bool(normalise)))
    if self_sensitivity:
        _, _, l2, e2, le2, auc2, ap2 = train(perturb=self_sensitivity, verbose=False)
        dl = np.abs(l2 - losses).max(axis=1)
        out.update(self_sens_perturb=np.float64(self_sensitivity), self_sens_auc=np.float64(abs(auc2 - auc)), self_sens_ap=np.float64(abs(ap2 - ap)),
                   self_sens_loss_by_epoch=dl, self_sens_eval_auc=np.abs(e2[:, 1] - evals[:, 1]), self_sens_eval_ap=np.abs(e2[:, 2] - evals[:, 2]),
                   self_sens_score=np.float64(np.abs(le2 - le).max()))
        print(f"self-sensitivity (initial weights x (1 + {self_sensitivity} N(0,1))): final auc moves {abs(auc2 - auc):.2e}, ap {abs(ap2 - ap):.2e}, "
              f"loss curve {dl.max():.2e}")
    for k, v in (planted or {}).items():
        out["planted." + k] = np.float64(v)
    for k, v in model.state_dict().items():
        out["final_norm." + k] = np.float64(np.linalg.norm(_np(v).astype(np.float64)))
    path = os.path.join(HERE, f"fullgraph_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "final auc", auc, "ap", ap)


# planted-anomaly parameters of the two round-6 fixtures (synth.plant_anomalies; chosen so that the REFERENCE separates the classes)
PLANTED_FULL = dict(scale=0.25, rewire=0.5)
PLANTED_MINI = dict(scale=0.5, dims=0.5, rewire=0.0, max_degree=2)


def part_full():
    _stub_third_party()
    sys.path.insert(0, REF)
    # matplotlib/networkx imports inside utils.py are present in the container
    full_graph_case("reddit_like", n=320, n_entries=2600, f=24, n_h=64, seed=0, mean=0.02, var=0.01,
                    kind="powerlaw", k_steps=6)
    full_graph_case("amazon_like", n=200, n_entries=9000, f=10, n_h=32, seed=3, mean=0.0, var=0.0,
                    kind="er", k_steps=4, outlier_rate=0.05 * 3, self_loop_frac=0.1)


# --------------------------------------------------------------------------
# Part 2: mini-batch GGAD (reference src/: graphsage.py, utils.py, layers.py)
# --------------------------------------------------------------------------
def _mini_setup(n, n_entries, f, seed, kind, self_loop_frac):
    import utils as sutils                       # /root/reference/src/utils.py
    rowptr, col = synth.make_graph(n, n_entries, seed, kind=kind, max_degree=max(8, n // 20),
                                   self_loop_frac=self_loop_frac)
    feat_raw = synth.make_features(n, f, seed)
    feat = np.asarray(sutils.normalize(feat_raw))             # src/utils.py:74-84
    adj_lists = synth.csr_to_adj_lists(rowptr, col)
    return rowptr, col, feat_raw, feat, adj_lists


def mini_module_case(tag, n, n_entries, f, d, seed, n_norm, n_ano, kind, k_steps, self_loop_frac=0.0):
    import torch.nn as nn
    import graphsage as gs                       # /root/reference/src/graphsage.py
    rowptr, col, feat_raw, feat, adj_lists = _mini_setup(n, n_entries, f, seed, kind, self_loop_frac)
    rng = np.random.default_rng(seed + 1)

    torch.manual_seed(seed)
    features = nn.Embedding(n, f)
    features.weight = nn.Parameter(torch.FloatTensor(feat), requires_grad=False)
    agg = gs.GCNAggregator(features, cuda=False)
    enc = gs.GCNEncoder(features, f, d, adj_lists, agg, gcn=True, cuda=False)
    model = gs.GCN(2, enc)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, weight_decay=0.007)

    out = dict(n=n, f=f, d=d, seed=seed, rowptr=rowptr, col=col, feat_raw=feat_raw, feat=feat.astype(np.float32),
               inputs_crc=synth.crc_of(rowptr, col, feat_raw))
    out["init.weight"] = _np(model.weight).copy()
    out["init.enc.weight"] = _np(enc.weight).copy()
    out["init.enc.fc.weight"] = _np(enc.fc.weight).copy()

    batches, labels_all, losses = [], [], []
    for step in range(k_steps):
        b = n_norm + n_ano
        nodes = rng.choice(n, size=b, replace=False).tolist()
        lab = np.zeros(b, dtype=np.int64)
        lab[n_norm:] = 1
        # "contamination" label-1 nodes in the middle of the batch (quirk 1, model_handler.py:168)
        lab[rng.choice(n_norm, size=2, replace=False)] = 1
        batches.append(nodes)
        labels_all.append(lab)
        opt.zero_grad()
        if step == 0:
            # module-level goldens: aggregator and encoder outputs (graphsage.py:295-360, 395-454)
            to_feats, to_feats_neigh, mask_row = agg.forward(nodes, [adj_lists[int(x)] for x in nodes],
                                                             adj_lists, True)
            # unique_nodes_list order is python-set order; recover it from mask_row's columns
            samp = [adj_lists[int(x)].union({int(x)}) for x in nodes]
            ulist = list(set.union(*samp))
            out.update(agg_to_feats=_np(to_feats), agg_to_feats_neigh=_np(to_feats_neigh),
                       agg_mask_row=_np(mask_row), agg_unique=np.array(ulist, dtype=np.int64))
            ca, tfn, af, afn = enc.forward(nodes, torch.LongTensor(lab), True)
            out.update(enc_combined_all=_np(ca), enc_to_feats_neigh=_np(tfn), enc_anomaly_feat=_np(af),
                       enc_anomaly_feat_new=_np(afn))
        total, l_cls, l_margin, l_rec = model.loss(nodes, torch.LongTensor(lab))
        total.backward()
        if step == 0:
            out["grad.weight"] = _np(model.weight.grad).copy()
            out["grad.enc.weight"] = _np(enc.weight.grad).copy()
            out["grad.enc.fc.weight"] = _np(enc.fc.weight.grad).copy()
        losses.append([total.item(), l_cls.item(), l_margin.item(), l_rec.item()])
        opt.step()
        if step == 0:
            out["step1.weight"] = _np(model.weight).copy()
            out["step1.enc.weight"] = _np(enc.weight).copy()
            out["step1.enc.fc.weight"] = _np(enc.fc.weight).copy()
    out["batches"] = np.array(batches, dtype=np.int64)
    out["labels"] = np.array(labels_all, dtype=np.int64)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["final.weight"] = _np(model.weight).copy()
    out["final.enc.weight"] = _np(enc.weight).copy()
    out["final.enc.fc.weight"] = _np(enc.fc.weight).copy()

    # inference (graphsage.py:178-181; src/utils.py:216-230): batches of `bs`, last one ragged
    test_nodes = rng.choice(n, size=min(n, 95), replace=False).tolist()
    bs = 30
    probs = []
    with torch.no_grad():
        for it in range(int(len(test_nodes) / bs) + 1):
            chunk = test_nodes[it * bs:(it + 1) * bs]
            if len(chunk) == 0:
                continue
            probs.extend(_np(model.to_prob(chunk, None)).reshape(-1).tolist())
    out["test_nodes"] = np.array(test_nodes, dtype=np.int64)
    out["test_bs"] = bs
    out["test_probs"] = np.array(probs, dtype=np.float32)

    # secondary modules: MeanAggregator / Encoder (graphsage.py:66-154), no sampling so it is deterministic
    torch.manual_seed(seed + 5)
    magg = gs.MeanAggregator(features, cuda=False, gcn=False)
    menc = gs.Encoder(features, f, d, adj_lists, magg, num_sample=None, gcn=False, cuda=False)
    nodes0 = batches[0]
    out["sage_weight"] = _np(menc.weight).copy()
    out["sage_mean"] = _np(magg.forward(nodes0, [adj_lists[int(x)] for x in nodes0], None))
    out["sage_enc"] = _np(menc.forward(nodes0))
    menc2 = gs.Encoder(features, f, d, adj_lists, gs.MeanAggregator(features, cuda=False, gcn=True),
                       num_sample=None, gcn=True, cuda=False)
    out["sage_gcn_weight"] = _np(menc2.weight).copy()
    out["sage_gcn_enc"] = _np(menc2.forward(nodes0))

    # IntraAgg (layers.py:179-244): same primitive ops, module-level only
    import layers as rl
    torch.manual_seed(seed + 9)
    intra = rl.IntraAgg(features, f, d, [], 0.5, cuda=False)
    tf, tfn, msk = intra.forward(nodes0, None, None, None, None, None, None, True, adj_lists)
    samp = [adj_lists[int(x)] for x in nodes0]
    out.update(intra_weight=_np(intra.weight).copy(), intra_to_feats=_np(tf), intra_to_feats_neigh=_np(tfn),
               intra_mask=_np(msk), intra_unique=np.array(list(set.union(*samp)), dtype=np.int64))

    path = os.path.join(HERE, f"minibatch_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "losses[0]", losses[0])


def sampler_case():
    """CPython `random.shuffle` sequences the native sampler must reproduce bit-for-bit
    (model_handler.py:29-30,314,341: seed 72, shuffle of the train list and of the pseudo-anomaly pool)."""
    out = {}
    for seed, sizes in ((72, (1, 2, 3, 7, 64, 1000, 55275)), (0, (10, 4097)), (2 ** 40 + 5, (33,))):
        random.seed(seed)
        for s in sizes:
            lst = list(range(s))
            random.shuffle(lst)
            random.shuffle(lst)            # state carries over between calls
            out[f"seed{seed}_n{s}"] = np.array(lst, dtype=np.int64)
        out[f"seed{seed}_tail"] = np.array([random.getrandbits(32) for _ in range(4)], dtype=np.int64)
    path = os.path.join(HERE, "sampler_shuffle.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def handler_case():
    """End-to-end `ModelHandler` run of the reference on a synthetic 'dgraphfin' (model_handler.py:23-414).

    The handler wants ../data/dgraphfin.npz and ./data/dgraphfin_adj_list relative to cwd, so a scratch
    tree is built under /tmp.  Only seeds + outputs are stored; the graph is regenerated from the seed.
    """
    import pickle
    import tempfile
    n, n_entries, f, seed = 90000, 300000, 17, 11
    rowptr, col = synth.make_graph(n, n_entries, seed, kind="powerlaw", max_degree=200)
    feat_raw = synth.make_features(n, f, seed)
    y = synth.make_labels(n, 0.02, seed)
    adj_lists = synth.csr_to_adj_lists(rowptr, col)
    tmp = tempfile.mkdtemp(prefix="ggad_golden_")
    os.makedirs(os.path.join(tmp, "data"))
    os.makedirs(os.path.join(tmp, "work", "data"))
    np.savez(os.path.join(tmp, "data", "dgraphfin.npz"), x=feat_raw, y=y)
    with open(os.path.join(tmp, "work", "data", "dgraphfin_adj_list"), "wb") as fh:
        pickle.dump(adj_lists, fh)
    cwd = os.getcwd()
    os.chdir(os.path.join(tmp, "work"))
    try:
        import model_handler as mh                 # /root/reference/src/model_handler.py
        cfg = dict(data_name="dgraphfin", data_dir="./data/", train_ratio=0.4, test_ratio=0.67,
                   save_dir="./pytorch_models/", model="GCN", multi_relation="GNN", emb_size=64, thres=0.4,
                   rho=0.5, seed=72, optimizer="adam", lr=0.001, weight_decay=0.007, batch_size=150,
                   num_epochs=1, valid_epochs=5, alpha=2, no_cuda=True, cuda_id="0")
        torch.manual_seed(72)                      # main.py:19-22 set_random_seed
        np.random.seed(72)
        handler = mh.ModelHandler(cfg)
        ds = handler.dataset
        out = dict(n=n, n_entries=n_entries, f=f, graph_seed=seed, inputs_crc=synth.crc_of(rowptr, col, feat_raw, y),
                   idx_train_head=np.array(ds["idx_train"][:2000], dtype=np.int64),
                   idx_train_len=len(ds["idx_train"]),
                   idx_train_crc=synth.crc_of(np.array(ds["idx_train"], dtype=np.int64)),
                   idx_anomaly=np.array(ds["idx_anomaly"], dtype=np.int64),
                   idx_test_head=np.array(ds["idx_test"][:2000], dtype=np.int64),
                   idx_test_len=len(ds["idx_test"]),
                   idx_test_crc=synth.crc_of(np.array(ds["idx_test"], dtype=np.int64)),
                   y_test_sum=int(np.sum(ds["y_test"])), labels_sum=int(np.sum(ds["labels"])),
                   feat_crc=synth.crc_of(np.asarray(ds["feat_data"], dtype=np.float32)))
        # capture the per-batch losses the training loop computes
        import graphsage as gs
        rec = []
        orig_loss = gs.GCN.loss

        def spy(self, nodes, labels):
            r = orig_loss(self, nodes, labels)
            rec.append([float(x) for x in r] + [float(len(nodes))])
            return r
        gs.GCN.loss = spy
        saved = {}
        orig_save = torch.save

        def save_spy(obj, path, *a, **k):
            saved.update({kk: _np(vv).copy() for kk, vv in obj.items() if "features" not in kk})
            return orig_save(obj, path, *a, **k)
        torch.save = save_spy
        res = handler.train()
        torch.save = orig_save
        gs.GCN.loss = orig_loss
        out["batch_losses"] = np.array(rec, dtype=np.float64)
        out["metrics"] = np.array(res, dtype=np.float64)   # f1_mac, f1_1, f1_0, auc, gmean
        for k, v in saved.items():
            out["ckpt." + k] = v
    finally:
        os.chdir(cwd)
    path = os.path.join(HERE, "handler_dgraph_like.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "metrics", out["metrics"])



def handler_long_case(num_epochs=5, valid_epochs=2, tag="5ep", n_entries=300000, planted=None):
    """`handler_case` over FIVE epochs of 150 batches with a validation sweep at epochs 0, 2 and 4 (model_handler.py:379-392), the
    best checkpoint restored and the test sweep at the end (:405-414) -- the end-of-training parity case of the mini-batch path.
    Stored: the loss terms of all 750 batches, the five metrics of every validation sweep, the weights each sweep saw (the last one:
    the weights at the END of training), and the final test metrics."""
    import pickle
    import tempfile
    n, f, seed = 90000, 17, 11
    rowptr, col = synth.make_graph(n, n_entries, seed, kind="powerlaw", max_degree=200)
    feat_raw = synth.make_features(n, f, seed)
    y = synth.make_labels(n, 0.02, seed)
    if planted:
        # round 6: planted anomalies (synth.plant_anomalies) -- sparse neighbourhoods + a feature-profile change: the reference's sweeps then
        # end well above chance instead of at 0.46-0.49, so the AUROC / AP parity below compares rankings that separate the classes
        rowptr, col, feat_raw = synth.plant_anomalies(rowptr, col, feat_raw, y, seed, **planted)
    adj_lists = synth.csr_to_adj_lists(rowptr, col)
    tmp = tempfile.mkdtemp(prefix="ggad_golden_")
    os.makedirs(os.path.join(tmp, "data"))
    os.makedirs(os.path.join(tmp, "work", "data"))
    np.savez(os.path.join(tmp, "data", "dgraphfin.npz"), x=feat_raw, y=y)
    with open(os.path.join(tmp, "work", "data", "dgraphfin_adj_list"), "wb") as fh:
        pickle.dump(adj_lists, fh)
    cwd = os.getcwd()
    os.chdir(os.path.join(tmp, "work"))
    try:
        import model_handler as mh
        cfg = dict(data_name="dgraphfin", data_dir="./data/", train_ratio=0.4, test_ratio=0.67,
                   save_dir="./pytorch_models/", model="GCN", multi_relation="GNN", emb_size=64, thres=0.4,
                   rho=0.5, seed=72, optimizer="adam", lr=0.001, weight_decay=0.007, batch_size=150,
                   num_epochs=num_epochs, valid_epochs=valid_epochs, alpha=2, no_cuda=True, cuda_id="0")
        torch.manual_seed(72)
        np.random.seed(72)
        handler = mh.ModelHandler(cfg)
        out = dict(n=n, n_entries=n_entries, f=f, graph_seed=seed, inputs_crc=synth.crc_of(rowptr, col, feat_raw, y),
                   num_epochs=num_epochs, valid_epochs=valid_epochs)
        import graphsage as gs
        rec = []
        orig_loss = gs.GCN.loss

        def spy(self, nodes, labels):
            r = orig_loss(self, nodes, labels)
            rec.append([float(x) for x in r])
            return r
        gs.GCN.loss = spy
        sweeps, seen, aps = [], [], []
        orig_test = mh.test_sage
        import utils as sutils                                                # src/utils.py: test_sage computes AP but only prints it (:232)
        orig_ap = sutils.average_precision_score

        def ap_spy(*a, **k):
            v = orig_ap(*a, **k)
            aps.append(float(v))
            return v
        sutils.average_precision_score = ap_spy

        def test_spy(cases, labels, model, batch_size, thres=0.5):
            r = orig_test(cases, labels, model, batch_size, thres)
            sweeps.append([float(x) for x in r])
            seen.append({k: _np(v).copy() for k, v in model.state_dict().items() if "features" not in k})
            return r
        mh.test_sage = test_spy
        res = handler.train()
        mh.test_sage = orig_test
        sutils.average_precision_score = orig_ap
        gs.GCN.loss = orig_loss
        out["sweep_ap"] = np.array(aps, dtype=np.float64)                     # the AP every sweep printed (src/utils.py:232), same order as `sweeps`
        for k, v in (planted or {}).items():
            out["planted." + k] = np.float64(v)
        out["batch_losses"] = np.array(rec, dtype=np.float64)                 # (750, 4): total, cls, margin, rec
        out["sweeps"] = np.array(sweeps, dtype=np.float64)                    # validations at epochs 0, 2, 4, then the test sweep
        out["metrics"] = np.array(res, dtype=np.float64)
        for k, v in seen[-2].items():                                         # the last validation: weights at the end of training
            out["end." + k] = v
        for k, v in seen[-1].items():                                         # the test sweep: the restored best checkpoint
            out["best." + k] = v
    finally:
        os.chdir(cwd)
    path = os.path.join(HERE, f"handler_dgraph_like_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "sweeps", out["sweeps"], "metrics", out["metrics"])


def baseline_case():
    """Mini-batch comparison models that share GGAD's 1-hop aggregate (src/graphsage_dominant.py, src/graphsage_anomalydae.py):
    module outputs, k training steps of the loop in src/model_handler_dominate.py:133-163, scores of src/utils.py:140-172."""
    import importlib
    import torch.nn as nn
    n, n_entries, f, d, seed, bsz, k_steps = 700, 3600, 17, 64, 21, 48, 5
    rowptr, col, feat_raw, feat, adj_lists = _mini_setup(n, n_entries, f, seed, "powerlaw", 0.05)
    feat = np.asarray(feat, dtype=np.float32)
    out = dict(n=n, f=f, d=d, seed=seed, rowptr=rowptr, col=col, feat_raw=feat_raw, feat=feat,
               inputs_crc=synth.crc_of(rowptr, col, feat_raw))
    rng = np.random.default_rng(seed + 1)
    batches = [rng.choice(n, size=bsz, replace=False).tolist() for _ in range(k_steps)]
    test_nodes = rng.choice(n, size=95, replace=False).tolist()
    out["batches"] = np.array(batches, dtype=np.int64)
    out["test_nodes"] = np.array(test_nodes, dtype=np.int64)
    out["test_bs"] = 30
    for tag, modname in (("dominant", "graphsage_dominant"), ("anomalydae", "graphsage_anomalydae")):
        gs = importlib.import_module(modname)            # /root/reference/src/graphsage_{dominant,anomalydae}.py
        torch.manual_seed(seed)
        features = nn.Embedding(n, f)
        features.weight = nn.Parameter(torch.FloatTensor(feat), requires_grad=False)
        agg = gs.GCNAggregator(features, cuda=False)
        enc = gs.GCNEncoder(features, f, d, adj_lists, agg, gcn=True, cuda=False)
        model = gs.GCN(2, enc)
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=1e-3, weight_decay=0.007)
        for k, v in model.state_dict().items():
            if "features" not in k:
                out[f"{tag}.init.{k}"] = _np(v).copy()
        losses = []
        for step, nodes in enumerate(batches):
            opt.zero_grad()
            if step == 0:
                out[f"{tag}.agg_to_feats"] = _np(agg.forward(nodes, [adj_lists[int(x)] for x in nodes]))
                out[f"{tag}.enc_out"] = _np(enc.forward(nodes))
            loss = model.loss(nodes, torch.tensor(feat)[nodes])              # model_handler_dominate.py:157
            loss.backward()
            if step == 0:
                out[f"{tag}.grad.enc.weight"] = _np(enc.weight.grad).copy()
                out[f"{tag}.grad.enc.fc.weight"] = _np(enc.fc.weight.grad).copy()
                assert model.weight.grad is None
            losses.append(loss.item())
            opt.step()
            if step == 0:
                out[f"{tag}.step1.enc.weight"] = _np(enc.weight).copy()
                out[f"{tag}.step1.enc.fc.weight"] = _np(enc.fc.weight).copy()
        out[f"{tag}.losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}.final.enc.weight"] = _np(enc.weight).copy()
        out[f"{tag}.final.enc.fc.weight"] = _np(enc.fc.weight).copy()
        # test_recon's score loop (src/utils.py:150-159); the last slice is ragged
        scores = []
        attr = torch.tensor(feat)
        with torch.no_grad():
            for it in range(int(len(test_nodes) / 30) + 1):
                chunk = test_nodes[it * 30:(it + 1) * 30]
                emb = model.to_prob(chunk, None)
                scores.extend(torch.sqrt(torch.sum(torch.pow(emb - attr[chunk], 2), 1)).numpy().tolist())
        out[f"{tag}.test_scores"] = np.array(scores, dtype=np.float32)
        print(tag, "losses", losses)
    path = os.path.join(HERE, "minibatch_baselines.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


def sage_train_case():
    """The vanilla GraphSAGE model of the reference (src/graphsage.py:19-154: MeanAggregator with python `random.sample`
    neighbour sampling, Encoder relu(W [self || mean]), GraphSage + CrossEntropyLoss) trained the way `ModelHandler` builds it
    for `model: 'SAGE'` (src/model_handler.py:278-293: gcn=False, `enc_sage.num_samples = 5` -- a typo that creates a new
    attribute, the encoder keeps num_sample = 10) with the batch loop of :310-365 (epoch shuffle of the train list, per-batch
    shuffle of the pseudo-anomaly pool, batch + first 50 pool nodes).  The reference's own loop cannot run this model: it
    unpacks four values from `GraphSage.loss` (which returns one) and calls `to_prob(nodes, None)` (which takes one argument);
    the harness makes exactly those two repairs.  Captured: the batches, per-step losses, weights, `to_prob` of test nodes."""
    import torch.nn as nn
    import graphsage as gs                       # /root/reference/src/graphsage.py
    n, f, d, seed = 900, 17, 64, 11
    rowptr, col, feat_raw, feat, adj_lists = _mini_setup(n, 9000, f, seed, "powerlaw", 0.0)
    rng = np.random.default_rng(seed)
    labels = (rng.random(n) < 0.1).astype(np.int64)
    idx_train = list(range(100, 700))
    idx_anomaly = [int(i) for i in np.nonzero(labels)[0][:60]]
    random.seed(72)
    np.random.seed(72)
    torch.manual_seed(72)
    features = nn.Embedding(n, f)
    features.weight = nn.Parameter(torch.FloatTensor(feat), requires_grad=False)
    agg = gs.MeanAggregator(features, cuda=False)
    enc = gs.Encoder(features, f, d, adj_lists, agg, gcn=False, cuda=False)
    enc.num_samples = 5                          # model_handler.py:291 (the typo is part of the reference)
    model = gs.GraphSage(2, enc)
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=0.001, weight_decay=0.007)
    out = dict(rowptr=rowptr, col=col, feat=feat.astype(np.float32), labels=labels, idx_train=np.array(idx_train),
               idx_anomaly=np.array(idx_anomaly), f=f, d=d)
    out["init.enc.weight"] = _np(enc.weight).copy()
    out["init.weight"] = _np(model.weight).copy()
    bs, nb, n_pseudo = 40, 4, 10
    losses, batches = [], []
    for epoch in range(2):
        random.shuffle(idx_train)                                    # :314
        for b in range(nb):
            batch_nodes = idx_train[b * bs:(b + 1) * bs]
            random.shuffle(idx_anomaly)                              # :341
            batch_nodes = batch_nodes + idx_anomaly[:n_pseudo]       # :342,347
            batch_label = labels[np.array(batch_nodes)]
            opt.zero_grad()
            loss = model.loss(batch_nodes, torch.LongTensor(batch_label))    # repair 1: one return value
            loss.backward()
            opt.step()
            losses.append(float(loss.item()))
            batches.append(np.array(batch_nodes, dtype=np.int64))
    out["losses"] = np.array(losses, dtype=np.float64)
    out["batches"] = np.stack(batches)
    out["final.enc.weight"] = _np(enc.weight).copy()
    out["final.weight"] = _np(model.weight).copy()
    test_nodes = list(range(700, 790))
    with torch.no_grad():
        probs = np.concatenate([_np(model.to_prob(test_nodes[s:s + 30])) for s in range(0, 90, 30)])   # repair 2: one argument
    out["test_nodes"] = np.array(test_nodes)
    out["test_probs"] = probs.astype(np.float32)
    st = random.getstate()
    out["py_random_after"] = np.array(st[1], dtype=np.uint64)
    path = os.path.join(HERE, "minibatch_sage.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "losses", losses)


def pcgnn_case():
    """PC-GNN skeleton of the reference (src/layers.py:11-153 `InterAgg`, src/model.py:8-48 `PCALayer`) on three synthetic relation
    graphs: the reference's `ModelHandler` cannot reach it on dgraphfin (one relation there, `test_pcgnn` undefined), so the vectors
    are module level -- embeddings, affinity, the two loss values, the gradient of every parameter that receives one, `to_prob`."""
    import torch.nn as nn
    import layers as rl                          # /root/reference/src/layers.py
    import model as rm                           # /root/reference/src/model.py
    import utils as sutils
    n, f, d, seed = 500, 17, 32, 21
    rels = [synth.make_graph(n, 3000 + 1500 * k, seed + k, kind="powerlaw", max_degree=40) for k in range(3)]
    feat = np.asarray(sutils.normalize(synth.make_features(n, f, seed))).astype(np.float32)
    adjs = [synth.csr_to_adj_lists(rp, ci) for rp, ci in rels]
    rng = np.random.default_rng(seed)
    nodes = [int(x) for x in rng.choice(n, size=48, replace=False)]
    labels = np.zeros(48, dtype=np.int64)
    labels[36:] = 1
    labels[[3, 17]] = 1
    torch.manual_seed(seed)
    features = nn.Embedding(n, f)
    features.weight = nn.Parameter(torch.FloatTensor(feat), requires_grad=False)
    intras = [rl.IntraAgg(features, f, d, [], 0.5, cuda=False) for _ in range(3)]
    inter = rl.InterAgg(features, f, d, [], adjs, intras, inter="GNN", cuda=False)
    model = rm.PCALayer(2, inter, 2)
    out = dict(f=f, d=d, feat=feat, nodes=np.array(nodes), labels=labels)
    for k, (rp, ci) in enumerate(rels):
        out[f"rowptr{k}"], out[f"col{k}"] = rp, ci
    names = {"inter1.weight": inter.weight, "inter1.intra_agg1.weight": intras[0].weight, "inter1.intra_agg2.weight": intras[1].weight,
             "inter1.intra_agg3.weight": intras[2].weight, "weight": model.weight}
    for k, p in names.items():
        out["init." + k] = _np(p).copy()
    lab_t = torch.LongTensor(labels)
    emb, aff = inter.forward(nodes, lab_t, True)
    out["combined"], out["affinity"] = _np(emb), _np(aff)
    loss, lcon = model.loss(nodes, lab_t, True)
    loss.backward()
    out["loss"] = np.array([loss.item(), lcon.item()], dtype=np.float64)
    for k, p in names.items():
        out["grad." + k] = _np(p.grad).copy()
    with torch.no_grad():
        gs_, ls_ = model.to_prob(nodes, lab_t, False)
    out["prob_gnn"], out["prob_label"] = _np(gs_), _np(ls_)
    path = os.path.join(HERE, "minibatch_pcgnn.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", out["loss"])


def part_mini(with_handler: bool):
    _stub_third_party()
    sys.path.insert(0, os.path.join(REF, "src"))
    mini_module_case("small", n=600, n_entries=3000, f=17, d=64, seed=5, n_norm=40, n_ano=10,
                     kind="powerlaw", k_steps=6, self_loop_frac=0.05)
    mini_module_case("dense", n=150, n_entries=3000, f=9, d=32, seed=8, n_norm=24, n_ano=6,
                     kind="er", k_steps=3, self_loop_frac=1.0)
    sampler_case()
    sage_train_case()
    pcgnn_case()
    if with_handler:
        handler_case()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", choices=["all", "full", "mini", "baselines", "ocgnn", "ingest", "sage", "pcgnn", "tam", "long_full", "long_mini", "planted_full", "planted_mini"], default="all")
    ap.add_argument("--no-handler", action="store_true", help="skip the slow end-to-end ModelHandler case")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; golden fixtures can only be regenerated in the build container")
    torch.set_num_threads(8)
    if a.part == "all":
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for p in ("full", "mini", "baselines", "ocgnn", "ingest", "sage", "pcgnn", "tam", "long_full", "long_mini", "planted_full", "planted_mini"):
            cmd = [sys.executable, os.path.abspath(__file__), "--part", p] + (["--no-handler"] if a.no_handler else [])
            subprocess.check_call(cmd, env=env)
    elif a.part == "full":
        part_full()
    elif a.part == "long_full":                    # end-of-training parity, full-graph script (~2 min of dense CPU epochs)
        _stub_third_party()
        sys.path.insert(0, REF)
        full_graph_long_case()
    elif a.part == "long_mini":                    # end-of-training parity, ModelHandler over 5 epochs
        _stub_third_party()
        sys.path.insert(0, os.path.join(REF, "src"))
        handler_long_case()
    elif a.part == "planted_full":                 # the same two schedules on PLANTED anomalies (round 6): AUROC that means something
        _stub_third_party()
        sys.path.insert(0, REF)
        # `--dataset photo --num_epoch 50`: the schedule on which the reference itself is well-conditioned (its final AUROC / AP move by
        # 3e-6 / 2e-5 under a 1e-7 relative change of its initial weights) -> the strict 1e-4 fixture;  the default 100 epochs run through
        # a re-activation of the margin hinge at epoch 59 after which the reference's OWN numbers move by 7e-4 / 9e-3 under that change:
        # kept as a second fixture with that self-sensitivity stored beside it
        full_graph_long_case(tag="long_planted", planted=PLANTED_FULL, normalise=False, num_epoch=50, self_sensitivity=1e-7)
        full_graph_long_case(tag="long_planted_100", planted=PLANTED_FULL, normalise=False, num_epoch=100, self_sensitivity=1e-7)
    elif a.part == "planted_mini":
        _stub_third_party()
        sys.path.insert(0, os.path.join(REF, "src"))
        handler_long_case(tag="planted", n_entries=600000, planted=PLANTED_MINI)
    elif a.part == "ingest":
        _stub_third_party()
        sys.path.insert(0, REF)
        ingest_case()
    elif a.part == "ocgnn":
        _stub_third_party()
        sys.path.insert(0, REF)
        ocgnn_case()
    elif a.part == "tam":
        _stub_third_party()
        sys.modules["torch_geometric.nn"].GINConv = object
        sys.path.insert(0, REF)
        tam_case()
    elif a.part == "pcgnn":
        _stub_third_party()
        sys.path.insert(0, os.path.join(REF, "src"))
        pcgnn_case()
    elif a.part == "sage":
        _stub_third_party()
        sys.path.insert(0, os.path.join(REF, "src"))
        sage_train_case()
    elif a.part == "baselines":
        _stub_third_party()
        sys.path.insert(0, os.path.join(REF, "src"))
        baseline_case()
    else:
        part_mini(not a.no_handler)
