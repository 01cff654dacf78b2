"""End-of-training parity (BASELINE north_star: "node anomaly scores, AUROC/AUPRC match the CPU reference within 1e-4 on identical
seeds"): the HIP path runs the WHOLE schedules the reference ran for tests/golden/fullgraph_long_photo_schedule.npz (100 epochs of the
full-graph script, run.py:137-240) and tests/golden/handler_dgraph_like_5ep.npz (ModelHandler over 5 epochs of 150 batches with three
validation sweeps, src/model_handler.py:310-414) and the deltas are REPORTED, whatever they are.  Shared by the `-m gpu` tests, which
assert what holds, and by scripts/end_of_training_report.py, which prints the table README.md quotes."""
import importlib.util
import os
import random
import types

import numpy as np
import torch

from conftest import ROOT, load_golden
from ggad_amd import synth


def _planted(g):
    """The synth.plant_anomalies keywords a fixture was generated with (empty: labels independent of everything, the round-5 fixtures)."""
    out = {k[len("planted."):]: float(g[k]) for k in g.keys() if k.startswith("planted.")}
    if "max_degree" in out:
        out["max_degree"] = int(out["max_degree"])
    return out


def full_graph_long(dev="cuda:0", no_graph=False, fixture="fullgraph_long_photo_schedule.npz"):
    import scipy.sparse as sp
    from ggad_amd import utils as U
    from ggad_amd.fullgraph import FullGraphAdj
    from ggad_amd.model import Model
    spec = importlib.util.spec_from_file_location("ggad_run_script", os.path.join(ROOT, "run.py"))
    run = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run)
    g = load_golden(fixture)
    n, f, h, seed = int(g["n"]), int(g["f"]), int(g["n_h"]), int(g["seed"])
    rowptr, col = synth.make_graph(n, int(g["n_entries"]), seed, kind="powerlaw", max_degree=n // 8)
    feat = synth.make_features(n, f, seed)
    ano = synth.make_labels(n, 0.06, seed)
    planted = _planted(g)
    if planted:
        rowptr, col, feat = synth.plant_anomalies(rowptr, col, feat, ano, seed, **planted)
    assert synth.crc_of(rowptr, col, feat, ano) == int(g["inputs_crc"])
    adj = synth.csr_to_scipy(rowptr, col, n)
    # (run.py:87-88: features are row-normalised for Amazon / reddit / elliptic only -- the planted fixture keeps them raw like photo)
    features = U.preprocess_features(sp.lil_matrix(feat)) if ("normalise" not in g or int(g["normalise"])) else feat
    dev = torch.device(dev)
    torch.cuda.set_device(dev)
    full = FullGraphAdj(U.normalize_adj(adj) + sp.eye(n), adj + sp.eye(n), dev)
    feats = torch.FloatTensor(np.asarray(features, dtype=np.float32)[np.newaxis]).to(dev)
    torch.manual_seed(seed)                                              # the one seeding of the run (run.py:62)
    model = Model(f, h, "prelu", 1, "avg")
    assert synth.crc_of(*[v.detach().numpy() for _, v in sorted(model.state_dict().items())]) == int(g["init_crc"])      # same initial weights
    model.to(dev)
    args = types.SimpleNamespace(mean=float(g["mean"]), var=float(g["var"]), lr=1e-3, weight_decay=0.0, num_epoch=int(g["num_epoch"]),
                                 embedding_dim=h, quiet=True, no_graph=no_graph, dataset="photo")
    hist = {}
    run.fit(args, dev, full, feats, model, g["normal_idx"].tolist(), g["abn_idx"].tolist(), g["idx_test"].tolist(), ano, history=hist)
    losses, ref = np.array(hist["losses"], dtype=np.float64), g["losses"]
    evals, ref_e = np.array(hist["eval"], dtype=np.float64), g["evals"]
    dl = np.abs(losses - ref).max(axis=1)
    first = {tol: (int(np.argmax(dl > tol)) if (dl > tol).any() else -1) for tol in (1e-5, 1e-4, 1e-3)}
    norms = {k[len("final_norm."):]: float(v) for k, v in g.items() if k.startswith("final_norm.")}
    got_norms = {k: float(np.linalg.norm(v.detach().double().cpu().numpy())) for k, v in model.state_dict().items()}
    return dict(epochs=len(ref), captured=bool(hist["captured"]),
                loss_delta_max=float(dl.max()), loss_delta_at=[float(dl[i]) for i in (0, 9, 49, len(ref) - 1)],
                first_epoch_loss_delta_above=first,
                eval_auc_delta_max=float(np.abs(evals[:, 1] - ref_e[:, 1]).max()), eval_ap_delta_max=float(np.abs(evals[:, 2] - ref_e[:, 2]).max()),
                eval_auc_delta_by_eval=np.abs(evals[:, 1] - ref_e[:, 1]).tolist(), eval_ap_delta_by_eval=np.abs(evals[:, 2] - ref_e[:, 2]).tolist(),
                final_auc=(float(hist["final_auc"]), float(g["final_auc"])), final_ap=(float(hist["final_ap"]), float(g["final_ap"])),
                final_auc_delta=abs(float(hist["final_auc"]) - float(g["final_auc"])), final_ap_delta=abs(float(hist["final_ap"]) - float(g["final_ap"])),
                final_score_delta_max=float(np.abs(hist["final_logits"] - g["final_logits"]).max()),
                final_score_scale=float(np.abs(g["final_logits"]).max()),
                weight_norm_rel_delta_max=max(abs(got_norms[k] - norms[k]) / (norms[k] + 1e-12) for k in norms),
                last_losses=(losses[-1].tolist(), ref[-1].tolist()))


def handler_long(tmp_dir, dev_id=0, fixture="handler_dgraph_like_5ep.npz"):
    from ggad_amd.model_handler import ModelHandler
    g = load_golden(fixture)
    n, seed = int(g["n"]), int(g["graph_seed"])
    rowptr, col = synth.make_graph(n, int(g["n_entries"]), seed, kind="powerlaw", max_degree=200)
    feat_raw = synth.make_features(n, int(g["f"]), seed)
    y = synth.make_labels(n, 0.02, seed)
    planted = _planted(g)
    if planted:
        rowptr, col, feat_raw = synth.plant_anomalies(rowptr, col, feat_raw, y, seed, **planted)
    assert synth.crc_of(rowptr, col, feat_raw, y) == int(g["inputs_crc"])
    cwd = os.getcwd()
    os.chdir(tmp_dir)
    try:
        cfg = dict(data_name="dgraphfin", data_dir="./data/", train_ratio=0.4, test_ratio=0.67, save_dir="./pytorch_models/",
                   model="GCN", multi_relation="GNN", emb_size=64, thres=0.4, rho=0.5, seed=72, optimizer="adam", lr=0.001,
                   weight_decay=0.007, batch_size=150, num_epochs=int(g["num_epochs"]), valid_epochs=int(g["valid_epochs"]), alpha=2,
                   no_cuda=False, cuda_id=str(dev_id), data=((rowptr, col), feat_raw, (y == 1).astype(np.int32)))
        random.seed(72)                                                  # main.py:19-22 set_random_seed (the handler re-seeds python's stream itself)
        torch.manual_seed(72)
        np.random.seed(72)
        h = ModelHandler(cfg)
        res = h.train()
    finally:
        os.chdir(cwd)
    losses = np.concatenate(h.epoch_losses, axis=0)
    ref = g["batch_losses"]
    dl = np.abs(losses[:, :4] - ref[:, :4]).max(axis=1)
    sweeps = np.array([m for _, m in h.valid_history] + [res], dtype=np.float64)
    names = ("f1_macro", "f1_1", "f1_0", "auc", "gmean")
    end = {k: float(np.abs(v.cpu().numpy() - g["end." + k]).max()) for k, v in h.end_state.items() if ("end." + k) in g}
    best = {k: float(np.abs(v.detach().cpu().numpy() - g["best." + k]).max()) for k, v in h.model.state_dict().items() if ("best." + k) in g}
    return dict(batches=len(ref), resident=bool(h.trainer.engine.resident), fallbacks=int(h.trainer.resident_fallbacks),
                loss_delta_max=float(dl.max()), loss_delta_by_epoch=[float(dl[150 * e:150 * (e + 1)].max()) for e in range(len(ref) // 150)],
                sweep_delta_max={nm: float(np.abs(sweeps[:, i] - g["sweeps"][:, i]).max()) for i, nm in enumerate(names)},
                test_metrics=(np.array(res, dtype=np.float64).tolist(), g["metrics"].tolist()),
                test_auc_delta=abs(float(res[3]) - float(g["metrics"][3])),
                sweep_auc=(sweeps[:, 3].tolist(), g["sweeps"][:, 3].tolist()),
                sweep_ap=((list(h.sweep_ap), g["sweep_ap"].tolist()) if "sweep_ap" in g else None),
                sweep_ap_delta_max=(float(np.abs(np.array(h.sweep_ap, dtype=np.float64) - g["sweep_ap"]).max()) if "sweep_ap" in g else None),
                end_weight_delta_max=max(end.values()), best_weight_delta_max=max(best.values()),
                valid_epochs=[e for e, _ in h.valid_history])
