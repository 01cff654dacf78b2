"""CPU oracle for the GGAD training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This module restates, on the CPU (numpy / scipy / torch-CPU fp32), the arithmetic of
the reference's hot path.  It exists only so that the HIP path can be
checked: the only importers allowed are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``ggad_amd/`` imports it, and the
product path raises if its HIP library is missing rather than falling back to this.

Pinning.  The reference ships no tests, golden vectors or fixtures, so
there is nothing of its own to pin against ("parity unpinned by the reference").  This
oracle is instead pinned against the reference ITSELF, imported and run in the build
container: ``tests/golden/make_golden.py`` captured its outputs on seeded synthetic
inputs into ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks every
function below against those captures (<= 2e-6 absolute, fp32 round-off of a different
summation order).  Third-party arithmetic (torch Linear/mm/PReLU/BCEWithLogits/
cosine_similarity/Adam, scipy sparse products) is used through the container's own
torch 2.10 / scipy 1.15, the same versions the goldens were produced with; the
reference pins torch==1.11.0 (`requirements.txt:7`).

Two formulations are provided where they differ in cost:
  * sparse (CSR / per-edge) -- what the HIP kernels implement;
  * dense-faithful          -- the same dense ops the reference executes (dense batch
    masks, dense N x N products); used as the honest "reference CPU path" timing in
    bench.py (``cpu_baseline.kind = "port"``).

All `file:line` citations are relative to /root/reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------
# Pre-processing (one-off, fp64 in scipy then cast to fp32 by the callers)
# ----------------------------------------------------------------------------------


def preprocess_features(feat: np.ndarray) -> np.ndarray:
    """Row-normalise: x / rowsum, inf -> 0  (`utils.py:37-44`, fp64)."""
    feat = np.asarray(feat, dtype=np.float64)
    rowsum = feat.sum(1)
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    return feat * r_inv[:, None]


def normalize_rows(feat: np.ndarray) -> np.ndarray:
    """DGraph variant: x / (rowsum + 0.01)  (`src/utils.py:74-84`).

    The reference multiplies a scipy ``diags`` by an fp32 ndarray, which promotes to fp64;
    ``torch.FloatTensor`` casts back (`src/model_handler.py:264`)."""
    feat = np.asarray(feat)
    rowsum = feat.sum(1).astype(np.float64) + 0.01 if feat.dtype != np.float32 else \
        np.array(feat.sum(1)) + 0.01
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    return (r_inv.astype(np.float64)[:, None] * feat.astype(np.float64))


def normalize_adj(rowptr: np.ndarray, col: np.ndarray, val: Optional[np.ndarray] = None):
    """D^-1/2 A D^-1/2 with inf -> 0, then + I  (`utils.py:47-54`, `run.py:98-101`).

    Returns (rowptr, col, val_fp64) of  normalize_adj(A) + I  as a CSR with sorted columns,
    plus the CSR of  A + I  ("raw_adj", `run.py:100`).  The reference computes
    ``adj.dot(D).transpose().dot(D)``, i.e. entry (j,i) of the result is
    a_ij * d_j^-1/2 * d_i^-1/2 with d = ROW sums of A.
    """
    import scipy.sparse as sp
    n = len(rowptr) - 1
    data = np.ones(len(col), dtype=np.float64) if val is None else np.asarray(val, dtype=np.float64)
    a = sp.csr_matrix((data, col.astype(np.int64), rowptr.astype(np.int64)), shape=(n, n))
    rowsum = np.asarray(a.sum(1)).reshape(-1)
    with np.errstate(divide="ignore"):
        d = np.power(rowsum, -0.5)
    d[np.isinf(d)] = 0.0
    dm = sp.diags(d)
    norm = a.dot(dm).transpose().dot(dm)
    adjn = (norm + sp.eye(n)).tocsr()
    adjn.sum_duplicates()
    adjn.sort_indices()
    raw = (a + sp.eye(n)).tocsr()
    raw.sum_duplicates()
    raw.sort_indices()
    return (adjn.indptr.astype(np.int32), adjn.indices.astype(np.int32), adjn.data.astype(np.float64),
            raw.indptr.astype(np.int32), raw.indices.astype(np.int32), raw.data.astype(np.float64))


# ----------------------------------------------------------------------------------
# Mini-batch path  (src/graphsage.py)
# ----------------------------------------------------------------------------------


@dataclass
class BatchAgg:
    """Result of the batch aggregation (GCNAggregator.forward, `src/graphsage.py:295-360`)."""
    to_feats: np.ndarray            # (B, F)   1-hop, weights 1/(sqrt(r_i) sqrt(c_j))
    unique: np.ndarray              # (U,)     sorted ids of  U = union_i N(i) + {i}
    to_feats_neigh: Optional[np.ndarray]   # (U, F) 2-hop rows (train only)
    ent_ptr: np.ndarray             # (B+1,)   closed-neighbourhood CSR of the batch rows
    ent_pos: np.ndarray             # (S1,)    column = position in `unique`
    r: np.ndarray                   # (B,)     |N(i) + {i}|

    def mask_row_dense(self) -> np.ndarray:
        """mask / rowsum as the dense (B, U) matrix the reference returns (`graphsage.py:317`)."""
        b = len(self.r)
        m = np.zeros((b, len(self.unique)), dtype=np.float32)
        for i in range(b):
            m[i, self.ent_pos[self.ent_ptr[i]:self.ent_ptr[i + 1]]] = np.float32(1.0) / np.float32(self.r[i])
        return m


def _closed_rows(rowptr, col, nodes) -> List[np.ndarray]:
    rows = []
    for v in nodes:
        v = int(v)
        nb = col[rowptr[v]:rowptr[v + 1]].astype(np.int64)
        rows.append(np.union1d(nb, np.array([v], dtype=np.int64)))     # `graphsage.py:305`
    return rows


def aggregate_batch(rowptr, col, feat, nodes: Sequence[int], train_flag: bool) -> BatchAgg:
    """Closed form of the dense-mask aggregation, fp32 accumulate.

    1-hop (`graphsage.py:305-326`): rows = batch nodes, columns = U; mask[i,j]=1 iff j in N(i)+{i};
    weight = 1/(sqrt(rowsum_i) sqrt(colsum_j)),  colsum_j = #rows of THIS batch containing j.
    2-hop (`:335-355`, train only): rows = U, columns = U2 = union N(u) (no self union);
    same normalisation with its own row / column sums.  The CPU branch adds no residual.
    """
    feat = np.asarray(feat, dtype=np.float32)
    rows = _closed_rows(rowptr, col, nodes)
    unique = np.unique(np.concatenate(rows)) if rows else np.zeros(0, dtype=np.int64)
    b = len(rows)
    ent_ptr = np.zeros(b + 1, dtype=np.int64)
    for i, rw in enumerate(rows):
        ent_ptr[i + 1] = ent_ptr[i] + len(rw)
    ent_col = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
    ent_pos = np.searchsorted(unique, ent_col)
    c = np.bincount(ent_pos, minlength=len(unique)).astype(np.float32)
    r = np.diff(ent_ptr).astype(np.float32)
    to_feats = np.zeros((b, feat.shape[1]), dtype=np.float32)
    for i in range(b):
        p = ent_pos[ent_ptr[i]:ent_ptr[i + 1]]
        w = (np.float32(1.0) / np.sqrt(r[i])) / np.sqrt(c[p])      # mask.div(row).div(col)
        to_feats[i] = (w[:, None] * feat[unique[p]]).sum(0, dtype=np.float32)
    to_feats_neigh = None
    if train_flag:
        nrows = [col[rowptr[int(u)]:rowptr[int(u) + 1]].astype(np.int64) for u in unique]   # `:339`
        allk = np.concatenate(nrows) if nrows else np.zeros(0, dtype=np.int64)
        u2, inv = np.unique(allk, return_inverse=True)
        c2 = np.bincount(inv, minlength=len(u2)).astype(np.float32)
        to_feats_neigh = np.zeros((len(unique), feat.shape[1]), dtype=np.float32)
        off = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            for ui, nb in enumerate(nrows):
                k = len(nb)
                pos = inv[off:off + k]
                off += k
                rr = np.float32(k)
                w = (np.float32(1.0) / np.sqrt(rr)) / np.sqrt(c2[pos])
                if k == 0:
                    to_feats_neigh[ui] = np.nan          # 0/0 row of the dense mask (quirk 3)
                else:
                    to_feats_neigh[ui] = (w[:, None] * feat[nb]).sum(0, dtype=np.float32)
    return BatchAgg(to_feats, unique, to_feats_neigh, ent_ptr, ent_pos, np.diff(ent_ptr))


def aggregate_batch_dense(adj_lists, feat_t: torch.Tensor, nodes, train_flag: bool):
    """Dense-faithful port: the very ops of `graphsage.py:295-360` (python sets, dense masks, mm).

    This is what the reference executes per batch on its CPU path and is the routine timed as
    ``cpu_baseline`` (kind "port").  Returns (to_feats, to_feats_neigh, mask_row, unique_list)."""
    samp = [adj_lists[int(v)].union({int(v)}) for v in nodes]
    ulist = list(set.union(*samp))
    index = {n: i for i, n in enumerate(ulist)}
    mask = torch.zeros(len(samp), len(ulist))
    cols = [index[n] for s in samp for n in s]
    rws = [i for i in range(len(samp)) for _ in range(len(samp[i]))]
    mask[rws, cols] = 1.0
    rn = mask.sum(1, keepdim=True).sqrt()
    cn = mask.sum(0, keepdim=True).sqrt()
    mask_row = mask.div(mask.sum(1, keepdim=True))
    mask = mask.div(rn).div(cn)
    to_feats = mask.mm(feat_t[torch.LongTensor(ulist)])
    tfn = None
    if train_flag:
        samp2 = [adj_lists[n] for n in ulist]
        ulist2 = list(set.union(*samp2))
        index2 = {n: i for i, n in enumerate(ulist2)}
        m2 = torch.zeros(len(samp2), len(ulist2))
        cols = [index2[n] for s in samp2 for n in s]
        rws = [i for i in range(len(samp2)) for _ in range(len(samp2[i]))]
        m2[rws, cols] = 1.0
        m2 = m2.div(m2.sum(1, keepdim=True).sqrt()).div(m2.sum(0, keepdim=True).sqrt())
        tfn = m2.mm(feat_t[torch.LongTensor(ulist2)])
    return to_feats, tfn, mask_row, ulist


class LazyAdjLists:
    """dict-of-sets view of a CSR, sets built on first touch (the reference unpickles ALL of them up
    front, `src/utils.py:26-28`; for the bounded cpu_baseline sample only the touched ones are needed;
    call ``warm`` before timing so that set construction is not billed to the reference's loop)."""

    def __init__(self, rowptr, col):
        self.rowptr, self.col, self._c = rowptr, col, {}

    def __getitem__(self, v):
        v = int(v)
        s = self._c.get(v)
        if s is None:
            s = set(self.col[self.rowptr[v]:self.rowptr[v + 1]].tolist())
            self._c[v] = s
        return s

    def get(self, v):
        return self[v]

    def warm(self, nodes):
        for v in nodes:
            for u in self[v]:
                self[u]


def dense_port_step(adj_lists, feat_t: torch.Tensor, p: "MiniParams", opt, nodes, labels) -> float:
    """One training step exactly as the reference executes it on its CPU path
    (`src/model_handler.py:356-364` -> `graphsage.py:244-258,395-454,295-360`): dense masks, dense mm."""
    nodes = [int(v) for v in nodes]
    to_feats, tfn, mask_row, _ = aggregate_batch_dense(adj_lists, feat_t, nodes, True)
    lab = torch.as_tensor(labels)
    combined = F.relu(p.enc_weight.mm(to_feats.t()))
    expand = F.relu(p.enc_weight.mm(tfn.t()))
    nbar = mask_row.mm(expand.t())
    a_feat = combined[:, lab == 1]
    new = F.relu(nbar.T[:, lab == 1].t().mm(p.enc_fc_weight.t()))
    combined_all = torch.cat((combined[:, lab == 0], new.t()), 1)
    scores = p.weight.mm(combined_all).t()
    cls = torch.mean(F.binary_cross_entropy_with_logits(scores.squeeze(), lab.float(), reduction="none",
                                                        pos_weight=torch.tensor([1])))
    aff = torch.cosine_similarity(combined_all, nbar.t(), dim=0)
    margin = (1 - (torch.mean(aff[torch.argwhere(lab == 0)], 0) - torch.mean(aff[torch.argwhere(lab == 1)], 0))).clamp_min(min=0)
    rec = torch.mean(torch.sqrt(torch.sum(torch.pow(a_feat - new.t(), 2), 0)))
    total = 1 * cls + 1 * margin + 0.1 * rec
    opt.zero_grad()
    total.backward()
    opt.step()
    return total.item()


@dataclass
class MiniParams:
    """Trainable tensors of the DGraph model; names = the reference's state_dict keys."""
    weight: torch.Tensor          # (1, D)    GCN.weight              `graphsage.py:168`
    enc_weight: torch.Tensor      # (D, F)    GCNEncoder.weight       `:388-390`
    enc_fc_weight: torch.Tensor   # (D, D)    GCNEncoder.fc.weight    `:391`

    def tensors(self):
        return [self.weight, self.enc_weight, self.enc_fc_weight]


def encoder_forward(p: MiniParams, agg: BatchAgg, labels: np.ndarray, train_flag: bool):
    """GCNEncoder.forward (`graphsage.py:395-454`) on an aggregated batch; torch fp32, autograd-able.

    Returns (combined_all (D,B), to_feats_neigh (B,D), anomaly_feat (D,A), anomaly_feat_new (D,A))."""
    x1 = torch.from_numpy(agg.to_feats)
    combined = F.relu(p.enc_weight.mm(x1.t()))                            # `:412`
    if not train_flag:
        return combined, None, None, None
    x2 = torch.from_numpy(agg.to_feats_neigh)
    expand = F.relu(p.enc_weight.mm(x2.t()))                              # `:419`  (D, U)
    b = len(agg.r)
    rows = torch.from_numpy(np.repeat(np.arange(b), np.diff(agg.ent_ptr)))
    pos = torch.from_numpy(agg.ent_pos.astype(np.int64))
    inv_r = torch.from_numpy((np.float32(1.0) / agg.r.astype(np.float32)))
    # mask_row.mm(expand.t()): mean over N(i)+{i} of the 1-hop embeddings      `:421`
    gathered = expand.t()[pos] * inv_r[rows][:, None]
    nbar = torch.zeros(b, expand.shape[0]).index_add(0, rows, gathered)
    lab = torch.as_tensor(labels)
    anomaly_feat = combined[:, lab == 1]                                  # `:427`
    anomaly_feat2 = nbar.t()[:, lab == 1]                                 # `:428`
    new = F.relu(anomaly_feat2.t().mm(p.enc_fc_weight.t()))               # `:430`  fc has no bias
    combined_all = torch.cat((combined[:, lab == 0], new.t()), 1)         # `:450` normals first
    return combined_all, nbar, anomaly_feat, new.t()


def batch_loss(p: MiniParams, agg: BatchAgg, labels: np.ndarray):
    """GCN.loss (`graphsage.py:244-258`): (total, cls, margin, rec), all torch scalars."""
    combined_all, nbar, a_feat, a_new = encoder_forward(p, agg, labels, True)
    scores = p.weight.mm(combined_all).t()                                # `:174-176`
    lab_f = torch.as_tensor(labels, dtype=torch.float32)
    lab = torch.as_tensor(labels)
    cls = torch.mean(F.binary_cross_entropy_with_logits(scores.squeeze(), lab_f, reduction="none",
                                                        pos_weight=torch.tensor([1])))       # `:246`
    aff = torch.cosine_similarity(combined_all, nbar.t(), dim=0)          # `:234` (eps 1e-8)
    a_norm = torch.mean(aff[torch.argwhere(lab == 0)], 0)
    a_abn = torch.mean(aff[torch.argwhere(lab == 1)], 0)
    margin = (1 - (a_norm - a_abn)).clamp_min(min=0)                      # `:235-240`
    rec = torch.mean(torch.sqrt(torch.sum(torch.pow(a_feat - a_new, 2), 0)))      # `:197-198`
    total = 1 * cls + 1 * margin + 0.1 * rec                              # `:258`
    return total, cls, margin, rec


def to_prob(p: MiniParams, rowptr, col, feat, nodes) -> np.ndarray:
    """GCN.to_prob (`graphsage.py:178-181`): sigmoid(w . relu(W . agg1hop)), one reference batch."""
    agg = aggregate_batch(rowptr, col, feat, nodes, False)
    with torch.no_grad():
        combined, _, _, _ = encoder_forward(p, agg, None, False)
        return torch.sigmoid(p.weight.mm(combined).t()).numpy().reshape(-1)


def mean_aggregate(rowptr, col, feat, nodes, gcn: bool) -> np.ndarray:
    """MeanAggregator.forward with num_sample=None (`graphsage.py:66-99`)."""
    feat = np.asarray(feat, dtype=np.float32)
    out = np.zeros((len(nodes), feat.shape[1]), dtype=np.float32)
    for i, v in enumerate(nodes):
        nb = col[rowptr[int(v)]:rowptr[int(v) + 1]].astype(np.int64)
        if gcn:
            nb = np.union1d(nb, np.array([int(v)]))
        out[i] = feat[nb].sum(0, dtype=np.float32) / np.float32(len(nb))
    return out


def intra_aggregate(rowptr, col, feat, nodes, weight):
    """IntraAgg.forward (`src/layers.py:179-244`): mean over the OPEN neighbourhood + relu(. W); 2-hop rows = the union U of
    those neighbourhoods (returned sorted -- the reference's order is python-set order), mask 1/(sqrt r sqrt c) over
    their neighbours + relu(. W).  Returns (to_feats (B, D), to_feats_neigh (U, D), mask (B, U), unique (U,))."""
    feat = np.asarray(feat, dtype=np.float32)
    W = np.asarray(weight, dtype=np.float32)
    nbrs = [col[rowptr[int(v)]:rowptr[int(v) + 1]].astype(np.int64) for v in nodes]
    unique = np.unique(np.concatenate(nbrs))
    pos = {int(n): i for i, n in enumerate(unique)}
    mask = np.zeros((len(nodes), len(unique)), dtype=np.float32)
    for i, nb in enumerate(nbrs):
        mask[i, [pos[int(k)] for k in nb]] = 1.0
    mask = mask / mask.sum(1, keepdims=True)                                   # layers.py:216-217
    to_feats = np.maximum((mask @ feat[unique]) @ W, 0.0)                      # :224-226
    nb2 = [col[rowptr[int(u)]:rowptr[int(u) + 1]].astype(np.int64) for u in unique]
    u2 = np.unique(np.concatenate(nb2))
    p2 = {int(n): i for i, n in enumerate(u2)}
    m2 = np.zeros((len(unique), len(u2)), dtype=np.float32)
    for i, nb in enumerate(nb2):
        m2[i, [p2[int(k)] for k in nb]] = 1.0
    m2 = (m2 / np.sqrt(m2.sum(1, keepdims=True))) / np.sqrt(m2.sum(0, keepdims=True))      # :236-238
    to_feats_neigh = np.maximum((m2 @ feat[u2]) @ W, 0.0)                      # :243-244
    return to_feats.astype(np.float32), to_feats_neigh.astype(np.float32), mask, unique


# ----------------------------------------------------------------------------------
# Mini-batch comparison models on the same 1-hop aggregate (DOMINANT / AnomalyDAE variants)
# ----------------------------------------------------------------------------------
def baseline_recon(rec: torch.Tensor, target: torch.Tensor, pos_weight: Optional[float] = None) -> torch.Tensor:
    """`GCN.reconstruction` of `src/graphsage_dominant.py:154-157` (pos_weight None) and of
    `src/graphsage_anomalydae.py:154-162` (pos_weight 0.5): mean over columns of sqrt(sum over the BATCH axis)."""
    diff = torch.pow(rec - target, 2)
    if pos_weight is not None:
        diff = torch.where(rec > 0, diff * pos_weight, diff * (1 - pos_weight))
    return torch.mean(torch.sqrt(torch.sum(diff, 0)))


def baseline_decode(weight: torch.Tensor, fc_weight: torch.Tensor, to_feats: torch.Tensor) -> torch.Tensor:
    """`GCNEncoder.forward` after the aggregation (`src/graphsage_dominant.py:274-276`): relu(fc(relu(W agg^T)^T)), (B, F)."""
    combined = F.relu(weight.mm(to_feats.t()))
    return F.relu(F.linear(combined.t(), fc_weight))


def baseline_loss(weight, fc_weight, rowptr, col, feat, nodes, target, pos_weight: Optional[float] = None):
    """`GCN.loss(nodes, features)` (`:167-171`) from the CSR: 1-hop batch aggregate (same closed form as GGAD's, `:194-226`),
    decode, reconstruction against `target` = the batch rows of the normalised feature table."""
    agg = aggregate_batch(rowptr, col, feat, nodes, False)
    rec = baseline_decode(weight, fc_weight, torch.from_numpy(agg.to_feats))
    return baseline_recon(rec, torch.as_tensor(np.asarray(target), dtype=torch.float32), pos_weight), rec


def baseline_scores(weight, fc_weight, rowptr, col, feat, cases, batch_size: int, attr) -> np.ndarray:
    """Scores of `test_recon` (`src/utils.py:150-159`): per slice of `batch_size`, sqrt(sum_c (rec - attr)^2)."""
    out = []
    attr = np.asarray(attr, dtype=np.float32)
    with torch.no_grad():
        for s in range(0, len(cases), batch_size):
            part = np.asarray(cases[s:s + batch_size])
            agg = aggregate_batch(rowptr, col, feat, part, False)
            rec = baseline_decode(weight, fc_weight, torch.from_numpy(agg.to_feats))
            out.append(torch.sqrt(torch.sum(torch.pow(rec - torch.from_numpy(attr[part]), 2), 1)).numpy())
    return np.concatenate(out) if out else np.zeros(0, dtype=np.float32)


def aegis_mlp(P: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, act, bn_state: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """The 2-layer stack of `torch_geometric.nn.MLP` (2.1.0: Linear -> BatchNorm1d (training mode: batch statistics, eps 1e-5) -> act ->
    Linear) as `src/graphsage_aegis.py:283-290` instantiates `discriminator2`; parameters named like PyG's state_dict.  PARITY
    UNPINNED: torch_geometric is absent from the image; this restates its published layer stack, it was never run against it."""
    h = x.mm(P[pre + ".lins.0.weight"].t()) + P[pre + ".lins.0.bias"]
    mean, var = h.mean(0), h.var(0, unbiased=False)
    h = (h - mean) / torch.sqrt(var + 1e-5) * P[pre + ".norms.0.module.weight"] + P[pre + ".norms.0.module.bias"]
    h = act(h)
    return h.mm(P[pre + ".lins.1.weight"].t()) + P[pre + ".lins.1.bias"]


def aegis_forward(P: Dict[str, torch.Tensor], rowptr, col, feat, noise, nodes):
    """`GCNEncoder.forward` of `src/graphsage_aegis.py:292-320` from the CSR: the batch's 1-hop aggregate of the feature table and of
    the noise table (`:194-226`), relu(W agg^T) of both, the discriminator on real ++ noise and on noise alone, sigmoid.
    Returns (logits_all (2B,), logits_gen (B,), label (2B,))."""
    a = torch.from_numpy(aggregate_batch(rowptr, col, feat, nodes, False).to_feats)
    z = torch.from_numpy(aggregate_batch(rowptr, col, noise, nodes, False).to_feats)
    combined = F.relu(a.mm(P["enc.weight"].t()))
    combined_noise = F.relu(z.mm(P["enc.weight"].t()))
    emb_all = torch.cat([combined, combined_noise], 0)
    label = torch.cat([torch.zeros(len(combined)), torch.ones(len(combined_noise))])
    logits_all = torch.sigmoid(aegis_mlp(P, "enc.discriminator2", emb_all, torch.sigmoid))[:, 0]
    logits_gen = torch.sigmoid(aegis_mlp(P, "enc.discriminator2", combined_noise, torch.sigmoid))[:, 0]
    return logits_all, logits_gen, label


def aegis_loss(P, rowptr, col, feat, noise, nodes):
    """`GCN.loss` (`src/graphsage_aegis.py:167-173`): (BCE(discriminator, real = 0 / noise = 1), BCE(discriminator(noise), 0))."""
    la, lg, label = aegis_forward(P, rowptr, col, feat, noise, nodes)
    return F.binary_cross_entropy(la, label), F.binary_cross_entropy(lg, torch.zeros_like(lg))


def make_adam(params: Sequence[torch.Tensor], lr: float, weight_decay: float):
    """The optimiser both entry points use (`run.py:118`, `src/model_handler.py:299-300`)."""
    return torch.optim.Adam(list(params), lr=lr, weight_decay=weight_decay)


# ----------------------------------------------------------------------------------
# Full-graph path  (model.py + loss block of run.py), sparse formulation
# ----------------------------------------------------------------------------------

FULL_PARAM_ORDER = ["gcn1.bias", "gcn1.fc.weight", "gcn1.act.weight", "gcn2.bias", "gcn2.fc.weight",
                    "gcn2.act.weight", "fc1.weight", "fc2.weight", "fc3.weight", "fc4.weight"]


def _spmm(rowptr, col, val, x: torch.Tensor) -> torch.Tensor:
    a = torch.sparse_csr_tensor(torch.from_numpy(rowptr.astype(np.int64)), torch.from_numpy(col.astype(np.int64)),
                                torch.from_numpy(np.asarray(val, dtype=np.float32)),
                                size=(len(rowptr) - 1, x.shape[0]))
    return torch.sparse.mm(a, x)


def full_forward(P: Dict[str, torch.Tensor], feat: torch.Tensor, adjn, abn_idx, normal_idx,
                 noise: torch.Tensor, train_flag: bool):
    """Model.forward (`model.py:133-191`) with the adjacency in CSR.

    ``adjn`` = (rowptr, col, val) of normalize_adj(A)+I; ``noise`` = the N(mean,var) draw of
    `model.py:143` (A x H), supplied by the caller so that RNG parity is the caller's business.
    Returns (emb_after_overwrite, emb_combine, logits, emb_con, emb_abnormal)."""
    rp, ci, va = adjn

    def gcn(x, pre):
        t = x.mm(P[pre + ".fc.weight"].t())                               # `model.py:27`
        out = _spmm(rp, ci, va, t) + P[pre + ".bias"]                     # `:31-33`
        return F.prelu(out, P[pre + ".act.weight"])                       # `:35`

    emb = gcn(gcn(feat, "gcn1"), "gcn2")
    abn = torch.as_tensor(np.asarray(abn_idx), dtype=torch.long)
    nrm = torch.as_tensor(np.asarray(normal_idx), dtype=torch.long)
    emb_abnormal = emb[abn] + noise                                       # `:141-144`
    if not train_flag:
        f3 = F.relu(F.relu(emb.mm(P["fc1.weight"].t())).mm(P["fc2.weight"].t())).mm(P["fc3.weight"].t())
        return emb, None, f3[:, 0], None, emb_abnormal
    # rows abn of the normalised adjacency times emb                      `:151-155`
    sub_rp = np.zeros(len(abn_idx) + 1, dtype=np.int64)
    cols, vals = [], []
    for k, a in enumerate(abn_idx):
        s, e = rp[a], rp[a + 1]
        cols.append(ci[s:e])
        vals.append(va[s:e])
        sub_rp[k + 1] = sub_rp[k] + (e - s)
    sub = torch.sparse_csr_tensor(torch.from_numpy(sub_rp), torch.from_numpy(np.concatenate(cols).astype(np.int64)),
                                  torch.from_numpy(np.concatenate(vals).astype(np.float32)),
                                  size=(len(abn_idx), emb.shape[0]))
    emb_con = F.relu(torch.sparse.mm(sub, emb).mm(P["fc4.weight"].t()))   # `:155-156`
    emb_combine = torch.cat((emb[nrm], emb_con), 0)                       # `:159`
    f3 = F.relu(F.relu(emb_combine.mm(P["fc1.weight"].t())).mm(P["fc2.weight"].t())).mm(P["fc3.weight"].t())
    emb2 = emb.index_copy(0, abn, emb_con)                                # `:182` in-place overwrite
    return emb2, emb_combine, f3[:, 0], emb_con, emb_abnormal


def ocgnn_forward(P: Dict[str, torch.Tensor], feat: torch.Tensor, adjn) -> torch.Tensor:
    """`model_ocgnn.Model.forward` (`model_ocgnn.py:128-131`): two GCN layers (`:26-35`) on the CSR adjacency."""
    rp, ci, va = adjn

    def gcn(x, pre):
        return F.prelu(_spmm(rp, ci, va, x.mm(P[pre + ".fc.weight"].t())) + P[pre + ".bias"], P[pre + ".act.weight"])
    return gcn(gcn(feat, "gcn1"), "gcn2")


def ocgnn_loss(emb: torch.Tensor, r: float = 0.0, beta: float = 0.5):
    """`loss_func` of `ocgnn.py:83-118`: the centre and radius are rebuilt (zeros, 0) on every call there, so the warm-up
    branch has no effect.  Returns (loss, score)."""
    dist = torch.sum(torch.pow(emb, 2), 1)
    score = dist - r ** 2
    return r ** 2 + 1 / beta * torch.mean(torch.relu(score)), score


def full_loss(emb, logits, emb_con, emb_abnormal, raw, abn_idx, normal_idx, margin_c: float = 0.7, by_column: bool = False):
    """Loss block of `run.py:165-210`, affinity as a per-edge SDDMM over raw_adj + I.

    affinity_j = sum_i cos(emb_i, emb_j) R_ij / sum_i R_ij   (column sums, `run.py:182-188`).
    loss_rec reduces over the OUTLIER axis (quirk 4, `run.py:207-208`).
    ``by_column``: the same sums associated per column, affinity_j = <e_hat_j, (R^T e_hat)_j> / colsum_j -- one sparse product
    instead of an (edges x H) intermediate (21 M x 300 floats at T-Finance size do not fit a host); pinned against the
    per-edge form and the reference's vectors in tests/test_oracle_golden.py."""
    rp, ci, va = raw
    if by_column:
        return _full_loss_by_column(emb, logits, emb_con, emb_abnormal, raw, abn_idx, normal_idx, margin_c)
    n_norm, n_out = len(normal_idx), emb_con.shape[0]
    lbl = torch.cat((torch.zeros(n_norm), torch.ones(n_out)))
    l_bce = torch.mean(F.binary_cross_entropy_with_logits(logits, lbl, reduction="none",
                                                          pos_weight=torch.tensor([1])))
    inv = torch.pow(torch.norm(emb, dim=-1, keepdim=True), -1)
    inv = torch.where(torch.isinf(inv), torch.zeros_like(inv), inv)
    en = emb * inv
    rows = torch.from_numpy(np.repeat(np.arange(len(rp) - 1), np.diff(rp)).astype(np.int64))
    cols = torch.from_numpy(ci.astype(np.int64))
    vals = torch.from_numpy(np.asarray(va, dtype=np.float32))
    per_edge = (en[rows] * en[cols]).sum(1) * vals
    colsum = torch.zeros(emb.shape[0]).index_add(0, cols, per_edge)
    rsum = torch.zeros(emb.shape[0]).index_add(0, cols, vals)
    r_inv = torch.pow(rsum, -1)
    r_inv = torch.where(torch.isinf(r_inv), torch.zeros_like(r_inv), r_inv)
    aff = colsum * r_inv
    abn = torch.as_tensor(np.asarray(abn_idx), dtype=torch.long)
    nrm = torch.as_tensor(np.asarray(normal_idx), dtype=torch.long)
    l_margin = (margin_c - (torch.mean(aff[nrm]) - torch.mean(aff[abn]))).clamp_min(min=0)
    diff = torch.pow(emb_con - emb_abnormal.unsqueeze(0), 2)              # (1, A, H)
    l_rec = torch.mean(torch.sqrt(torch.sum(diff, 1)))                    # sums over A  -> (1, H)
    return l_margin + l_bce + l_rec, l_margin, l_bce, l_rec, aff


def _full_loss_by_column(emb, logits, emb_con, emb_abnormal, raw, abn_idx, normal_idx, margin_c):
    import scipy.sparse as sp
    rp, ci, va = raw
    n = emb.shape[0]
    n_norm, n_out = len(normal_idx), emb_con.shape[0]
    lbl = torch.cat((torch.zeros(n_norm), torch.ones(n_out)))
    l_bce = torch.mean(F.binary_cross_entropy_with_logits(logits, lbl, reduction="none", pos_weight=torch.tensor([1])))
    inv = torch.pow(torch.norm(emb, dim=-1, keepdim=True), -1)
    inv = torch.where(torch.isinf(inv), torch.zeros_like(inv), inv)
    en = emb * inv
    rt = sp.csr_matrix((np.asarray(va, dtype=np.float32), ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).T.tocsr()
    rt.sort_indices()
    rten = _spmm(rt.indptr, rt.indices, rt.data, en)                      # (R^T e_hat)_j = sum_i R_ij e_hat_i
    rsum = torch.from_numpy(np.asarray(rt.sum(1), dtype=np.float32).reshape(-1))
    r_inv = torch.pow(rsum, -1)
    r_inv = torch.where(torch.isinf(r_inv), torch.zeros_like(r_inv), r_inv)
    aff = (en * rten).sum(1) * r_inv
    abn = torch.as_tensor(np.asarray(abn_idx), dtype=torch.long)
    nrm = torch.as_tensor(np.asarray(normal_idx), dtype=torch.long)
    l_margin = (margin_c - (torch.mean(aff[nrm]) - torch.mean(aff[abn]))).clamp_min(min=0)
    diff = torch.pow(emb_con - emb_abnormal.unsqueeze(0), 2)
    l_rec = torch.mean(torch.sqrt(torch.sum(diff, 1)))
    return l_margin + l_bce + l_rec, l_margin, l_bce, l_rec, aff


def full_step_dense(P: Dict[str, torch.Tensor], feat: torch.Tensor, adj: torch.Tensor, raw: torch.Tensor, abn_idx, normal_idx,
                    noise: torch.Tensor, margin_c: float = 0.7):
    """One training forward + loss of `run.py:146-210` with the DENSE N x N operands the reference holds (`adj` = normalize_adj(A) + I,
    `raw` = A + I as (N, N) float tensors): `torch.bmm(adj, seq_fts)` per GCN layer (`model.py:26-35`), `adj[0, abn, :] @ emb`
    (`model.py:151-156`), the N x N similarity `emb_n @ emb_n.T * raw_adj` and its column sums (`run.py:177-188`).  Same arithmetic as
    full_forward + full_loss, other association; used as the dense-faithful CPU baseline of bench.py.  Returns the total loss."""
    def gcn(x, pre):
        t = x.mm(P[pre + ".fc.weight"].t())                               # `model.py:27`
        out = torch.bmm(adj[None], t[None])[0] + P[pre + ".bias"]         # `:31-33`
        return F.prelu(out, P[pre + ".act.weight"])                       # `:35`
    emb = gcn(gcn(feat, "gcn1"), "gcn2")
    abn = torch.as_tensor(np.asarray(abn_idx), dtype=torch.long)
    nrm = torch.as_tensor(np.asarray(normal_idx), dtype=torch.long)
    emb_abnormal = emb[abn] + noise                                       # `model.py:141-144`
    emb_con = F.relu(adj[abn, :].mm(emb).mm(P["fc4.weight"].t()))         # `:151-156`
    emb_combine = torch.cat((emb[nrm], emb_con), 0)                       # `:159`
    logits = F.relu(F.relu(emb_combine.mm(P["fc1.weight"].t())).mm(P["fc2.weight"].t())).mm(P["fc3.weight"].t())[:, 0]
    emb = emb.index_copy(0, abn, emb_con)                                 # `:182`
    lbl = torch.cat((torch.zeros(len(normal_idx)), torch.ones(emb_con.shape[0])))
    l_bce = torch.mean(F.binary_cross_entropy_with_logits(logits, lbl, reduction="none", pos_weight=torch.tensor([1])))
    inv = torch.pow(torch.norm(emb, dim=-1, keepdim=True), -1)            # `run.py:177-181`
    inv = torch.where(torch.isinf(inv), torch.zeros_like(inv), inv)
    en = emb * inv
    sim = en.mm(en.t()) * raw                                             # `:182-184`
    r_inv = torch.pow(raw.sum(0), -1)
    r_inv = torch.where(torch.isinf(r_inv), torch.zeros_like(r_inv), r_inv)
    aff = sim.sum(0) * r_inv                                              # `:185-188`
    l_margin = (margin_c - (torch.mean(aff[nrm]) - torch.mean(aff[abn]))).clamp_min(min=0)
    diff = torch.pow(emb_con - emb_abnormal.unsqueeze(0), 2)
    l_rec = torch.mean(torch.sqrt(torch.sum(diff, 1)))                    # quirk 4: reduces over the outlier axis
    return l_margin + l_bce + l_rec


# ----------------------------------------------------------------------------------
# TAM comparison model (`tam.py`, `model_tam.py`, `utils_tam.py`): truncated affinity maximisation
# ----------------------------------------------------------------------------------
def tam_split(ano_labels: np.ndarray, rng) -> Tuple[List[int], np.ndarray]:
    """The index lists of `load_mat` (`utils_tam.py:163-179`): 30 % / 10 % / 60 % split of a python-`random` shuffle, 80 % of
    the normal training nodes, plus 15 % "contamination" -- the reference indexes the SHUFFLED list with the positions of
    the anomalies (`:172`), so the contaminating nodes are random nodes, not anomalies (kept).  `rng` = python's `random`.
    Returns (normal_label_idx, idx_test)."""
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


def tam_calc_distance(rowptr, col, feat: np.ndarray) -> np.ndarray:
    """`calc_distance` (`utils_tam.py:190-199`) on the stored entries of A + I: Euclidean distance of the two attribute rows,
    aligned with `col` (the reference fills a dense N x N array entry by entry)."""
    x = torch.from_numpy(np.asarray(feat, dtype=np.float32))
    rows = np.repeat(np.arange(len(rowptr) - 1), np.diff(rowptr))
    d = x[torch.from_numpy(rows.astype(np.int64))] - x[torch.from_numpy(np.asarray(col, dtype=np.int64))]
    return torch.sqrt(torch.sum(d * d, 1)).numpy()


def tam_graph_nsgt(dis: torch.Tensor, adj: torch.Tensor, nprandom) -> torch.Tensor:
    """`graph_nsgt` (`utils_tam.py:222-240`), dense like the reference (small graphs only): per row, if the largest distance to
    a neighbour exceeds the mean non-zero distance over the current edges, draw ONE number from numpy's global stream and
    cut the row's edges longer than min + u (max - min); an edge survives if either direction survives (adj + adj.T)."""
    adj = adj.clone()
    du = dis * adj
    mean_dis = du[du != 0].mean()
    for i in range(dis.shape[0]):
        node_index = torch.argwhere(adj[i, :] > 0)
        if node_index.shape[0] != 0:
            max_dis = dis[i, node_index].max()
            if max_dis > mean_dis:
                random_value = (max_dis - mean_dis) * nprandom.random_sample() + mean_dis
                cutting = torch.argwhere(dis[i, node_index[:, 0]] > random_value)
                if cutting.shape[0] != 0:
                    adj[i, node_index[cutting[:, 0]]] = 0
    adj = adj + adj.T
    adj[adj > 1] = 1
    return adj


def tam_normalize_adj(adj: torch.Tensor) -> torch.Tensor:
    """`normalize_adj_tensor` (`utils_tam.py:45-53`): D^-1/2 A D^-1/2 with D = column sums, inf -> 0."""
    r_inv = torch.pow(torch.sum(adj, 0), -0.5).flatten()
    r_inv[torch.isinf(r_inv)] = 0.0
    return torch.mm(torch.diag_embed(r_inv), torch.mm(adj, torch.diag_embed(r_inv)))


def tam_forward(P: Dict[str, torch.Tensor], feat: torch.Tensor, adj_norm: torch.Tensor):
    """`model_tam.Model.forward` (`model_tam.py:150-157`): two GCN layers (`:36-45`), then the two unused projections."""
    def gcn(x, pre):
        return F.prelu(adj_norm.mm(x.mm(P[pre + ".fc.weight"].t())) + P[pre + ".bias"], P[pre + ".act.weight"])
    emb = gcn(gcn(feat, "gcn1"), "gcn2")
    return emb, emb.mm(P["fc1.weight"].t()), emb.mm(P["fc2.weight"].t())


def tam_message(emb: torch.Tensor, raw_adj: torch.Tensor, zero_nan: bool) -> torch.Tensor:
    """The local affinity of `tam.py:113-127,136-146`: row sums of (e_hat e_hat^T) * raw_adj, divided by the column sums of
    raw_adj (inf -> 0); `max_message` also zeroes inf / NaN products, `inference` does not."""
    f = emb / torch.norm(emb, dim=-1, keepdim=True)
    sim = torch.mm(f, f.T) * raw_adj
    if zero_nan:
        sim = torch.where(torch.isinf(sim) | torch.isnan(sim), torch.zeros_like(sim), sim)
    r_inv = torch.pow(torch.sum(raw_adj, 0), -1).flatten()
    r_inv = torch.where(torch.isinf(r_inv), torch.zeros_like(r_inv), r_inv)
    return torch.sum(sim, 1) * r_inv


def tam_max_message(emb: torch.Tensor, raw_adj: torch.Tensor, normal_idx) -> Tuple[torch.Tensor, torch.Tensor]:
    """`max_message` (`tam.py:113-133`): min-max normalised affinity, loss = - sum over the labelled normal nodes."""
    m = tam_message(emb, raw_adj, True)
    m = (m - torch.min(m)) / (torch.max(m) - torch.min(m))
    return -torch.sum(m[torch.as_tensor(np.asarray(normal_idx), dtype=torch.long)]), m
