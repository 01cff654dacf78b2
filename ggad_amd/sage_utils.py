"""Evaluation helpers of the DGraph path (reference `src/utils.py:207-260,324-326`).

Scores come from the HIP inference path (`GCN.to_prob` semantics: per-batch normalisation with the
reference's batch boundaries) and stay on the device; the metrics follow sklearn's definitions (the reference's
metric library) and are computed on the device too (`ggad_amd/metrics.py`; `device_metrics=False` runs sklearn)."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
from sklearn.metrics import average_precision_score, confusion_matrix, f1_score, roc_auc_score

from .minibatch import BatchChunk


def prob2pred(y_prob: np.ndarray, thres: float = 0.5) -> np.ndarray:
    """`src/utils.py:250-260`."""
    return (np.asarray(y_prob) >= thres).astype(np.int32)


def conf_gmean(conf: np.ndarray) -> float:
    """`src/utils.py:324-326`."""
    tn, fp, fn, tp = conf.ravel()
    return float((tp * tn / ((tp + fn) * (tn + fp))) ** 0.5)


def pos_neg_split(nodes, labels):
    """`src/utils.py:115-130`: the nodes with label 1 and the others, both in their original order."""
    nodes = list(nodes)
    lab = np.asarray(labels).reshape(-1)
    pos_nodes = [n for n, l in zip(nodes, lab) if l == 1]
    if len(set(nodes)) != len(nodes):                 # duplicated ids: keep the reference's `list.remove` semantics
        return pos_nodes, _remove_first(nodes, pos_nodes)
    drop = set(pos_nodes)
    return pos_nodes, [n for n in nodes if n not in drop]


def _remove_first(nodes, remove):
    out = list(nodes)                                 # first occurrence only
    for r in remove:
        out.remove(r)
    return out


def pick_step(idx_train, y_train, adj_list, size):
    """`src/utils.py:133-137`: degree / label-frequency weighted sampling with python's `random.choices` (same RNG stream)."""
    import random
    y = np.asarray(y_train)
    degree = np.array([len(adj_list[node]) for node in idx_train])
    lf = (y.sum() - len(y)) * y + len(y)
    return random.choices(idx_train, weights=degree / lf, k=size)


def score_nodes(model, test_cases: Sequence[int], batch_size: int, batches_per_launch: int = 2048, device: bool = False,
                dist=None):
    """Probabilities for `test_cases`, batched EXACTLY like `test_sage` (`src/utils.py:216-230`):
    consecutive slices of `batch_size`; the column counts of the aggregation are per slice (quirk 2).
    Thousands of reference batches are planned and scored per launch group.

    With an initialised `torch.distributed` module as `dist`, the sweep is sharded: rank r scores a
    contiguous range of the reference's batches (boundaries unchanged, so the per-batch normalisation is the same)
    and one all-reduce of the zero-padded score vector hands every rank all scores."""
    enc = model.enc
    eng = enc.engine
    cases = np.asarray(test_cases, dtype=np.int64)
    n = len(cases)
    out = torch.empty(n, dtype=torch.float32, device=eng.dev)
    if n == 0:
        return out if device else out.cpu().numpy()
    from .graphsage import _as_graph
    graph = _as_graph(enc.adj_lists, enc.features.weight.shape[0], eng.dev)
    # slots are N ints per batch: bound the number of batches in flight by memory (<= ~2 GiB of counters)
    per_launch = int(max(1, min(batches_per_launch, (1 << 29) // max(1, graph.n))))
    key = ("eval", id(graph), per_launch)
    cache = getattr(model, "_eval_chunks", None)
    if cache is None:
        cache = model._eval_chunks = {}
    ch = cache.get(key)
    if ch is None:
        ch = cache[key] = BatchChunk(graph, enc.features.weight.data, enc.embed_dim, per_launch,
                                     per_launch * batch_size, per_launch * batch_size * 8, train=False)
    eng.sync_params()
    step = per_launch * batch_size
    lo, hi = 0, n
    if dist is not None and dist.get_world_size() > 1:
        world, rank = dist.get_world_size(), dist.get_rank()
        nb = (n + batch_size - 1) // batch_size
        lo = min(n, ((nb * rank) // world) * batch_size)
        hi = min(n, ((nb * (rank + 1)) // world) * batch_size)
        out.zero_()
    for s in range(lo, hi, step):
        part = cases[s:min(hi, s + step)]
        batches = [part[i:i + batch_size] for i in range(0, len(part), batch_size)]
        ch.build(batches)
        eng.score_chunk(ch, out[s:s + len(part)])
    ch.reset()
    if dist is not None and dist.get_world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)              # disjoint slices, zeros elsewhere
    return out if device else out.cpu().numpy()


def test_sage(test_cases, labels, model, batch_size, thres=0.5, device_metrics=True, dist=None, verbose=True):
    """Reference `test_sage` (`src/utils.py:207-247`): same prints, same return tuple.  `dist`: shard the sweep over the
    ranks (every rank returns the same metrics; pass verbose=False on the ranks that must not print)."""
    if device_metrics:
        from .metrics import binary_report
        probs = score_nodes(model, test_cases, batch_size, device=True, dist=dist)
        r = binary_report(probs, torch.as_tensor(np.asarray(labels), device=probs.device), thres)
        f1_binary_1, f1_binary_0, f1_macro, auc_gnn, ap, gmean = r["f1_1"], r["f1_0"], r["f1_macro"], r["auc"], r["ap"], r["gmean"]
        tn, fp, fn, tp = r["tn"], r["fp"], r["fn"], r["tp"]
    else:
        probs = score_nodes(model, test_cases, batch_size, dist=dist)
        preds = prob2pred(probs, thres)
        labels = np.asarray(labels)
        auc_gnn = roc_auc_score(labels, probs)
        ap = average_precision_score(labels, probs, average="macro", pos_label=1, sample_weight=None)
        f1_binary_1 = f1_score(labels, preds, pos_label=1, average="binary")
        f1_binary_0 = f1_score(labels, preds, pos_label=0, average="binary")
        f1_macro = f1_score(labels, preds, average="macro")
        conf = confusion_matrix(labels, preds)
        tn, fp, fn, tp = conf.ravel()
        gmean = conf_gmean(conf)
    if verbose:
        print(f"   GNN F1-binary-1: {f1_binary_1:.4f}\tF1-binary-0: {f1_binary_0:.4f}" +
              f"\tF1-macro: {f1_macro:.4f}\tG-Mean: {gmean:.4f}\tAUC: {auc_gnn:.4f}")
        print("Testing AP:", ap)
        print(f"   GNN TP: {tp}\tTN: {tn}\tFN: {fn}\tFP: {fp}")
    test_sage.last_ap = float(ap)          # the reference only PRINTS the AP (src/utils.py:232); kept here for the parity tests
    return f1_macro, f1_binary_1, f1_binary_0, auc_gnn, gmean


def recon_scores(model, test_cases: Sequence[int], batch_size: int, test_attr, batches_per_launch: int = 1024):
    """Per-node reconstruction error of the DOMINANT / AnomalyDAE baselines, batched EXACTLY like `test_recon`
    (`src/utils.py:150-159`): consecutive slices of `batch_size` (the aggregation normalises per slice), many slices per
    plan; the decoder is row-wise, so all rows of a plan go through it at once.  Returns a device tensor."""
    enc = model.enc
    dev = enc.weight.device
    cases = np.asarray(test_cases, dtype=np.int64)
    out = torch.empty(len(cases), dtype=torch.float32, device=dev)
    attr = test_attr if isinstance(test_attr, torch.Tensor) else torch.as_tensor(np.asarray(test_attr))
    attr = attr.to(device=dev, dtype=torch.float32)
    from ._lib import call, ptr
    from .graphsage import _as_graph
    graph = _as_graph(enc.adj_lists, enc.features.weight.shape[0], dev)
    per_launch = int(max(1, min(batches_per_launch, (1 << 29) // max(1, graph.n))))
    step = per_launch * batch_size
    with torch.no_grad():
        for s in range(0, len(cases), step):
            part = cases[s:s + step]
            batches = [part[i:i + batch_size] for i in range(0, len(part), batch_size)]
            x1, _ = enc.aggregator.aggregate(batches, graph, per_launch)
            emb = enc.decode(x1).contiguous()
            tgt = attr[torch.from_numpy(part).to(dev)].contiguous()
            call("ggad_recon_rows_f32", ptr(emb), ptr(tgt), len(part), emb.shape[1], ptr(out[s:s + len(part)]))
    return out


def test_recon(test_cases, labels, model, batch_size, test_attr, thres=0.5, verbose=True):
    """Reference `test_recon` (`src/utils.py:140-172`): AUROC / AP of the reconstruction error; same prints.  The
    reference returns nothing; the two numbers are returned here as well."""
    from .metrics import average_precision, roc_auc
    scores = recon_scores(model, test_cases, batch_size, test_attr)
    y = torch.as_tensor(np.asarray(labels), device=scores.device)
    auc_gnn, ap = roc_auc(scores, y), average_precision(scores, y)
    if verbose:
        print("Testing AUC", auc_gnn)
        print("Testing AP:", ap)
    return auc_gnn, ap


def aegis_scores(model, test_cases: Sequence[int], batch_size: int, batches_per_launch: int = 512) -> torch.Tensor:
    """`to_prob` of the AEGIS-style model for every reference batch of `test_cases` (`src/utils.py:184-193`): the aggregation of
    many batches per plan, the discriminator batch by batch (its batch norm uses the statistics of each batch: the reference never
    leaves training mode)."""
    cases = np.asarray(list(test_cases), dtype=np.int64)
    n_it = int(len(cases) / batch_size) + 1
    slices = [cases[i * batch_size:min((i + 1) * batch_size, len(cases))] for i in range(n_it)]
    slices = [s for s in slices if len(s)]
    enc = model.enc
    out = []
    with torch.no_grad():
        for g0 in range(0, len(slices), batches_per_launch):
            grp = slices[g0:g0 + batches_per_launch]
            x_feat, x_noise, bp = enc.aggregator.aggregate(grp, enc.adj_lists, len(grp))
            for b in range(len(grp)):
                logits, _, _ = enc.discriminate(x_feat[bp[b]:bp[b + 1]], x_noise[bp[b]:bp[b + 1]])
                out.append(logits[:int(len(logits) / 2), 0].clone())
    return torch.cat(out)


def test_aegis(test_cases, labels, model, batch_size, thres=0.5, verbose=True):
    """Reference `test_aegis` (`src/utils.py:175-204`): AP / AUROC of the discriminator's score of the real nodes; same prints.  The
    reference returns nothing; the two numbers are returned here as well."""
    from .metrics import average_precision, roc_auc
    scores = aegis_scores(model, test_cases, batch_size)
    y = torch.as_tensor(np.asarray(labels), device=scores.device)
    auc_gnn, ap = roc_auc(scores, y), average_precision(scores, y)
    if verbose:
        print("Testing AP:", ap)
        print("Testing AUC", auc_gnn)
    return auc_gnn, ap
