"""Host / device helpers of the full-graph TAM comparison model (`tam.py`, `utils_tam.py` of the reference) on CSR matrices.

The reference holds the adjacency, the distance table and every truncated graph as dense N x N tensors and walks them row by
row in Python; here they are CSR (scipy on the host for the once-per-cut truncation, whose only sequential part is the
order of numpy's random draws; HIP kernels for everything per epoch):

    load_mat / split_nodes            `utils_tam.py:140-179`   (python `random` driven split, incl. its index quirk)
    calc_distance                     `utils_tam.py:190-199`   -> `ggad_edge_dist_f32`, one value per stored entry of A + I
    graph_nsgt                        `utils_tam.py:222-240`   -> vectorised over rows, same draws from numpy's stream
    normalize_adj_tensor              `utils_tam.py:45-53`
    max_message / inference           `tam.py:113-146`         -> `AffinityFn` (HIP row-normalise, CSR SpMM, row dots)
"""
from __future__ import annotations

import random as _pyrandom
from typing import List, Tuple

import numpy as np
import torch

from ._lib import call, ptr
from .fullgraph import FullGraphAdj, spmm


def split_nodes(ano_labels: np.ndarray, rng=_pyrandom) -> Tuple[List[int], np.ndarray]:
    """(normal_label_idx, idx_test) of `load_mat` (`utils_tam.py:163-179`): 30 / 10 / 60 % split of a shuffled index list, 80 %
    of the normal training nodes, plus int(0.15 * #anomalies) nodes taken from the SHUFFLED list at the positions of the
    anomalies (`:172` -- the reference's "contamination" are therefore random nodes; kept)."""
    ano_labels = np.asarray(ano_labels)
    n = len(ano_labels)
    all_idx = list(range(n))
    rng.shuffle(all_idx)
    n_train, n_val = int(n * 0.3), int(n * 0.1)
    idx_train, idx_test = all_idx[:n_train], all_idx[n_train + n_val:]
    all_normal = [i for i in idx_train if ano_labels[i] == 0]
    normal = all_normal[: int(len(all_normal) * 0.8)]
    real_abnormal = np.array(all_idx)[np.argwhere(ano_labels == 1).squeeze()].tolist()
    add_rate = 0.15 * len(real_abnormal)
    rng.shuffle(real_abnormal)
    add = real_abnormal[:int(add_rate)]
    return normal + add, np.setdiff1d(idx_test, add, False)


def load_mat(dataset: str, root: str = "./data"):
    """`load_mat` of `utils_tam.py:140-179`: (adj, feat, ano_labels, str_ano_labels, attr_ano_labels, normal_label_idx, idx_test)."""
    import scipy.io as sio
    import scipy.sparse as sp
    data = sio.loadmat("{}/{}.mat".format(root, dataset))
    label = data["Label"] if ("Label" in data) else data["gnd"]
    attr = data["Attributes"] if ("Attributes" in data) else data["X"]
    network = data["Network"] if ("Network" in data) else data["A"]
    adj, feat = sp.csr_matrix(network), sp.lil_matrix(attr)
    ano = np.squeeze(np.array(label))
    str_l = np.squeeze(np.array(data["str_anomaly_label"])) if "str_anomaly_label" in data else None
    attr_l = np.squeeze(np.array(data["attr_anomaly_label"])) if "str_anomaly_label" in data else None
    normal, idx_test = split_nodes(ano)
    return adj, feat, ano, str_l, attr_l, normal, idx_test


def calc_distance(raw, feats: torch.Tensor) -> np.ndarray:
    """Attribute distance of every stored entry of `raw` (scipy CSR of A + I, sorted indices), aligned with `raw.indices`.
    The reference's N x N `dis_array` holds exactly these values at the non-zero positions of raw_adj and 0 elsewhere."""
    dev = feats.device
    x = feats.reshape(-1, feats.shape[-1]).contiguous().float()
    rp = torch.from_numpy(raw.indptr.astype(np.int32)).to(dev)
    ci = torch.from_numpy(raw.indices.astype(np.int32)).to(dev)
    out = torch.empty(raw.nnz, dtype=torch.float32, device=dev)
    call("ggad_edge_dist_f32", ptr(rp), ptr(ci), ptr(x), raw.shape[0], x.shape[1], ptr(out))
    return out.cpu().numpy()


def graph_nsgt(raw, dis_vals: np.ndarray, adj, nprandom=np.random):
    """One truncation step (`graph_nsgt`, `utils_tam.py:222-240`).  `raw`: scipy CSR of the ORIGINAL A + I (sorted indices) with
    `dis_vals` = the distance of each of its entries (`calc_distance`); `adj`: scipy CSR pattern of the current graph (a subset
    of `raw`'s pattern).  Per row with at least one neighbour whose largest distance exceeds the mean non-zero distance over
    the current entries, ONE number is taken from numpy's global stream (in row order, like the reference's loop) and the
    row's entries farther than mean + u (max - mean) are cut; an entry survives if either direction survives (adj + adj.T).
    Returns the new CSR pattern (data = 1)."""
    import scipy.sparse as sp
    adj = sp.csr_matrix(adj)
    adj.sort_indices()
    n = adj.shape[0]
    deg = np.diff(adj.indptr)
    key_raw = np.repeat(np.arange(n, dtype=np.int64), np.diff(raw.indptr)) * n + raw.indices
    key_adj = np.repeat(np.arange(n, dtype=np.int64), deg) * n + adj.indices
    pos = np.searchsorted(key_raw, key_adj)
    if len(pos) and (pos.max() >= len(key_raw) or not np.array_equal(key_raw[pos], key_adj)):
        raise ValueError("graph_nsgt: the current graph has an entry the original one lacks")
    dis = np.asarray(dis_vals, dtype=np.float32)[pos]
    nz = dis[dis != 0]
    mean_dis = torch.from_numpy(nz).mean().numpy() if len(nz) else np.float32("nan")      # fp32 mean, torch's summation order
    rows = np.nonzero(deg > 0)[0]
    mx = np.maximum.reduceat(dis, adj.indptr[rows]) if len(rows) else np.zeros(0, np.float32)
    qual = mx > mean_dis
    u = nprandom.random_sample(int(qual.sum()))                  # one draw per qualifying row, in row order
    thr = np.full(n, np.inf, dtype=np.float32)
    mean32 = np.float32(mean_dis)
    thr[rows[qual]] = (mx[qual] - mean32).astype(np.float32) * u.astype(np.float32) + mean32
    keep = ~(dis > np.repeat(thr, deg))
    cut = sp.csr_matrix((keep.astype(np.float32), adj.indices.copy(), adj.indptr.copy()), shape=adj.shape)
    cut.eliminate_zeros()                                        # (in place: hence the copies above)
    sym = (cut + cut.T).tocsr()
    sym.data[:] = 1.0
    sym.sort_indices()
    return sym


def normalize_adj_tensor(adj):
    """`normalize_adj_tensor` (`utils_tam.py:45-53`) on a scipy CSR 0/1 matrix: entries r_i r_j with r = colsum^-1/2 (inf -> 0),
    evaluated in fp32 like the reference's dense tensors."""
    import scipy.sparse as sp
    adj = sp.csr_matrix(adj, dtype=np.float32)
    colsum = torch.from_numpy(np.asarray(adj.sum(0)).reshape(-1).astype(np.float32))
    r = torch.pow(colsum, -0.5)
    r[torch.isinf(r)] = 0.0
    r = r.numpy()
    rows = np.repeat(np.arange(adj.shape[0]), np.diff(adj.indptr))
    vals = (r[rows] * (adj.data * r[adj.indices]).astype(np.float32)).astype(np.float32)
    return sp.csr_matrix((vals, adj.indices.copy(), adj.indptr.copy()), shape=adj.shape)


class AffinityFn(torch.autograd.Function):
    """message_i = r_inv_i <e_hat_i, (R e_hat)_i> for every node: the row sums of (e_hat e_hat^T) * raw_adj divided by the column
    sums of raw_adj (`tam.py:113-127`), without the N x N product.  `adj.Rt` is used as R: TAM's raw adjacency is symmetric
    (`FullGraphAdj` built from A + I).  Rows of zero norm give 0 (the reference's `max_message` zeroes their NaN products)."""

    @staticmethod
    def forward(ctx, emb, adj: FullGraphAdj):
        emb = emb.contiguous()
        n, h = emb.shape
        dev = emb.device
        inv = torch.empty(n, dtype=torch.float32, device=dev)
        en = torch.empty_like(emb)
        call("ggad_rownorm_f32", ptr(emb), n, h, ptr(inv), ptr(en))
        s = spmm(adj.Rt, en)
        r_inv = adj.r_inv_dev()
        aff = torch.empty(n, dtype=torch.float32, device=dev)
        call("ggad_rowdot_f32", ptr(en), None, ptr(s), n, h, ptr(r_inv), ptr(aff))
        ctx.save_for_backward(en, inv, s, r_inv)
        ctx.adj = adj
        return aff

    @staticmethod
    def backward(ctx, g):
        en, inv, s, r_inv = ctx.saved_tensors
        adj = ctx.adj
        n, h = en.shape
        c = (g * r_inv).contiguous()                                     # d loss / d <e_hat_i, S_i>
        ar = adj.arange_dev()
        xc = torch.empty_like(en)
        call("ggad_rows_scale_f32", ptr(en), ptr(ar), ptr(c), n, h, 0, ptr(xc))          # c_i e_hat_i
        den = spmm(adj.Rt, xc)                                           # R^T (c . e_hat)  (R symmetric)
        call("ggad_rows_scale_f32", ptr(s), ptr(ar), ptr(c), n, h, 1, ptr(den))           # + c_i S_i
        d_emb = torch.empty_like(en)
        call("ggad_rownorm_bwd_f32", ptr(en), ptr(inv), ptr(den), n, h, ptr(d_emb))
        return d_emb, None


def inference(emb: torch.Tensor, adj: FullGraphAdj) -> torch.Tensor:
    """`inference` (`tam.py:136-146`): the affinity of every node."""
    return AffinityFn.apply(emb.reshape(-1, emb.shape[-1]), adj)


def max_message(emb: torch.Tensor, adj: FullGraphAdj, normal_label_idx) -> Tuple[torch.Tensor, torch.Tensor]:
    """`max_message` (`tam.py:113-133`): (- sum of the min-max normalised affinity over the labelled normal nodes, that vector)."""
    m = AffinityFn.apply(emb.reshape(-1, emb.shape[-1]), adj)
    m = (m - torch.min(m)) / (torch.max(m) - torch.min(m))
    idx = normal_label_idx if isinstance(normal_label_idx, torch.Tensor) else torch.as_tensor(
        np.asarray(normal_label_idx, dtype=np.int64), device=m.device)
    return -torch.sum(m.index_select(0, idx.reshape(-1).long())), m


def normalize_score(ano_score: np.ndarray) -> np.ndarray:
    return (ano_score - np.min(ano_score)) / (np.max(ano_score) - np.min(ano_score))             # utils_tam.py:56-59


def train_cut(model, optimiser, features: torch.Tensor, adj: FullGraphAdj, normal_label_idx, num_epoch: int, use_graph: bool = True,
              log_every: int = 0):
    """The epoch loop of one truncation round (`tam.py:186-201`): forward, `max_message` loss, `inference`, backward, Adam step.
    The reference calls `zero_grad()` ONCE per round (`:182`), so the gradients of a round accumulate from epoch to epoch;
    the caller does the same (this function never clears them).  After two eager epochs the epoch (forward, both affinity
    passes, backward into the accumulating gradients, fused Adam) is captured into one hipGraph and replayed.
    Returns (losses [num_epoch] fp32 tensor on the device, message of the last epoch)."""
    dev = features.device
    idx = torch.as_tensor(np.asarray(normal_label_idx, dtype=np.int64), device=dev)
    losses = torch.zeros(max(1, int(num_epoch)), dtype=torch.float32, device=dev)
    state = {}

    def epoch():
        node_emb, feat1, feat2 = model.forward(features, adj)
        loss, _ = max_message(node_emb[0], adj, idx)
        with torch.no_grad():
            state["message"] = inference(node_emb[0].detach(), adj)
        loss.backward()
        optimiser.step()
        return loss.detach()

    model.train()
    n_eager = min(int(num_epoch), 2 if use_graph else int(num_epoch))
    for e in range(n_eager):
        losses[e] = epoch()
        if log_every and e % log_every == 0:
            print("mean_loss is {}".format(losses[e].item()))
    left = int(num_epoch) - n_eager
    if left > 0:
        import gc
        gc.collect()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        static_loss = torch.zeros((), dtype=torch.float32, device=dev)
        with torch.cuda.graph(graph):
            static_loss.copy_(epoch())
            static_msg = state["message"]
        # the capture does not execute: every remaining epoch is one replay
        for e in range(n_eager, int(num_epoch)):
            graph.replay()
            losses[e] = static_loss
            if log_every and e % log_every == 0:
                print("mean_loss is {}".format(losses[e].item()))
        state["message"] = static_msg
    return losses[:int(num_epoch)], state["message"]
