"""`IntraAgg` of the reference's `src/layers.py:163-244` (PC-GNN style intra-relation aggregator with the GGAD 2-hop
extension) behind the same constructor / forward signature / parameter name.

Module-level parity only, as in the reference itself: its training entry point is unreachable there.
The neighbour lists come from python sets exactly as in the reference (the column order of the returned mask and the
row order of `to_feats_neigh` are the iteration order of `set.union`), the aggregations run in the HIP ragged-gather
kernel (`ggad_seg_mean` / `ggad_seg_wsum`), the projections on the MFMA GEMM with autograd (`LinearFn`)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from ._lib import call, ptr
from .fullgraph import LinearFn
from .graphsage import _features, _node_array


def _ragged(lists, index=None):
    sizes = np.fromiter((len(s) for s in lists), dtype=np.int64, count=len(lists))
    seg_ptr = np.zeros(len(lists) + 1, dtype=np.int32)
    np.cumsum(sizes, out=seg_ptr[1:])
    cols = np.fromiter((v for s in lists for v in s), dtype=np.int64, count=int(seg_ptr[-1]))
    return sizes, seg_ptr, cols


class IntraAgg(nn.Module):
    def __init__(self, features, feat_dim, embed_dim, train_pos, rho, cuda=False):
        super().__init__()
        self.features = _features(features)
        self.cuda = cuda
        self.feat_dim = feat_dim
        self.embed_dim = embed_dim
        self.train_pos = train_pos
        self.rho = rho
        w = torch.empty(self.feat_dim, self.embed_dim)
        init.xavier_uniform_(w)                                                        # layers.py:176-177
        self.weight = nn.Parameter(w.to(self.features.weight.device))

    def forward(self, nodes, batch_labels, to_neighs_list, batch_scores, neigh_scores, pos_scores, sample_list,
                train_flag, adj_list):
        """Returns (to_feats (B, D), to_feats_neigh (U, D), mask (B, U)) like `layers.py:179-244`; only `nodes` and
        `adj_list` (dict node -> set) are used, as in the reference (its neighbour filtering is commented out)."""
        nodes = _node_array(nodes)
        dev = self.features.weight.device
        feat = self.features.weight.data
        f = feat.shape[1]
        samp = [adj_list[int(n)] for n in nodes]
        unique_nodes_list = list(set.union(*samp))                                     # layers.py:205 (python set order)
        unique = {n: i for i, n in enumerate(unique_nodes_list)}
        sizes, seg_ptr, cols = _ragged(samp)
        sp = torch.from_numpy(seg_ptr).to(dev)
        sc = torch.from_numpy(cols.astype(np.int32)).to(dev)
        agg = torch.empty(len(samp), f, device=dev)
        call("ggad_seg_mean", ptr(feat), f, ptr(sp), ptr(sc), len(samp), ptr(agg))     # mask.div(num_neigh).mm(embed)   :216-224
        wt = self.weight.t().contiguous()
        to_feats = LinearFn.apply(agg, wt, True)                                       # relu(agg.mm(weight))           :226
        mask = torch.zeros(len(samp), len(unique_nodes_list), device=dev)
        rows = np.repeat(np.arange(len(samp)), sizes)
        ucol = np.fromiter((unique[int(v)] for v in cols), dtype=np.int64, count=len(cols))
        mask[torch.from_numpy(rows).to(dev), torch.from_numpy(ucol).to(dev)] = 1.0
        mask = mask / mask.sum(1, keepdim=True)
        # 2-hop: rows = unique nodes, columns = their neighbours, 1 / (sqrt(row sum) sqrt(column sum))        :228-242
        samp2 = [adj_list[int(u)] for u in unique_nodes_list]
        sizes2, seg_ptr2, cols2 = _ragged(samp2)
        _, inv = np.unique(cols2, return_inverse=True)
        col_cnt = np.bincount(inv).astype(np.float32)[inv]
        row_cnt = np.repeat(sizes2.astype(np.float32), sizes2)
        w2 = ((np.float32(1.0) / np.sqrt(row_cnt)) / np.sqrt(col_cnt)).astype(np.float32)
        sp2 = torch.from_numpy(seg_ptr2).to(dev)
        sc2 = torch.from_numpy(cols2.astype(np.int32)).to(dev)
        sw2 = torch.from_numpy(w2).to(dev)
        agg2 = torch.empty(len(samp2), f, device=dev)
        call("ggad_seg_wsum", ptr(feat), f, ptr(sp2), ptr(sc2), ptr(sw2), len(samp2), ptr(agg2))
        to_feats_neigh = LinearFn.apply(agg2, wt, True)
        self.last_unique = np.asarray(unique_nodes_list, dtype=np.int64)
        return to_feats, to_feats_neigh, mask


class InterAgg(nn.Module):
    """Inter-relation aggregator of the reference's PC-GNN skeleton (`src/layers.py:11-153`): three `IntraAgg`s (one per relation
    graph), their batch embeddings concatenated and projected, the same for the neighbourhood means of their 2-hop embeddings,
    and the cosine affinity between the two (`:125-153`).  Same constructor, parameter names and return values
    (`combined.t()` (D, B), `affinity` (B,)).  The reference also evaluates a label-aware score head (`label_clf`, `:80-101`) whose
    results only feed the neighbour filtering it has commented out (`:193-198`): the head is kept as a parameter (state_dict and RNG
    parity), its evaluation is skipped.  Aggregations: HIP ragged gathers inside `IntraAgg`; every product (the per-relation
    projections, mask @ to_feats_neigh, the two (3D -> D) projections) runs on the MFMA GEMM with autograd (`LinearFn`)."""

    def __init__(self, features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter="GNN", cuda=True):
        super().__init__()
        self.features = _features(features)
        self.dropout = 0.6
        self.adj_lists = adj_lists
        self.intra_agg1, self.intra_agg2, self.intra_agg3 = intraggs[0], intraggs[1], intraggs[2]
        self.embed_dim = embed_dim
        self.feat_dim = feature_dim
        self.inter = inter
        self.cuda = cuda
        for a in intraggs:
            a.cuda = cuda
        self.train_pos = train_pos
        self.thresholds = [0.5, 0.5, 0.5]
        dev = self.features.weight.device
        w = torch.empty(self.embed_dim * len(intraggs), self.embed_dim)
        self.weight_gen = nn.Parameter(torch.zeros(self.embed_dim, self.embed_dim, device=dev))     # unused in the reference too (:49)
        init.xavier_uniform_(w)                                                                    # :51
        self.weight = nn.Parameter(w.to(dev))
        self.label_clf = nn.Linear(self.feat_dim, 2).to(dev)                                       # :54
        self.weights_log, self.thresholds_log, self.relation_score_log = [], [self.thresholds], []

    def forward(self, nodes, labels, train_flag=True):
        nodes = [int(v) for v in _node_array(nodes)]
        r_feats, nb_feats = [], []
        for agg, adj in zip((self.intra_agg1, self.intra_agg2, self.intra_agg3), self.adj_lists):
            to_feats, to_feats_neigh, mask = agg.forward(nodes, labels, None, None, None, None, None, train_flag, adj)   # :104-112
            r_feats.append(to_feats)
            nb_feats.append(LinearFn.apply(mask.contiguous(), to_feats_neigh.t().contiguous(), False))   # mask.mm(to_feats_neigh) :131-133
        wt = self.weight.t().contiguous()
        combined = LinearFn.apply(torch.cat(r_feats, dim=1), wt, True)                                  # :127-129
        neigh = LinearFn.apply(torch.cat(nb_feats, dim=1), wt, True)                                    # :135-136
        cn = combined / torch.norm(combined, dim=-1, keepdim=True)                                      # :139-145
        cn = torch.where(torch.isnan(cn), torch.full_like(cn, 0), cn)
        nn_ = neigh / torch.norm(neigh, dim=-1, keepdim=True)
        nn_ = torch.where(torch.isnan(nn_), torch.full_like(nn_, 0), nn_)
        affinity = (nn_ * cn).sum(1)                                                                    # diag(mm(.,.T))  :146
        return combined.t(), affinity


class PCALayer(nn.Module):
    """`PCALayer` of the reference's `src/model.py:8-48`: class scores from the inter-relation embedding, cross entropy +
    5 x the affinity margin (margin 1); `loss` returns (total, margin term), `to_prob` the two sigmoid score sets."""

    def __init__(self, num_classes, inter1, lambda_1):
        super().__init__()
        self.inter1 = inter1
        self.xent = nn.CrossEntropyLoss()
        w = torch.empty(num_classes, inter1.embed_dim)
        init.xavier_uniform_(w)
        self.weight = nn.Parameter(w.to(inter1.weight.device))
        self.lambda_1 = lambda_1
        self.epsilon = 0.1

    def forward(self, nodes, labels, train_flag=True):
        embeds1, affinity = self.inter1(nodes, labels, train_flag)
        scores_t = LinearFn.apply(embeds1.t().contiguous(), self.weight, False)        # (weight.mm(embeds1)).t()   :25-26
        return scores_t, affinity

    def to_prob(self, nodes, labels, train_flag=True):
        gnn_logits, label_logits = self.forward(nodes, labels, train_flag)
        return torch.sigmoid(gnn_logits), torch.sigmoid(label_logits)

    def affinity(self, affinity, labels):
        a0 = torch.mean(affinity[torch.argwhere(labels == 0)], 0)
        a1 = torch.mean(affinity[torch.argwhere(labels == 1)], 0)
        return (1 - (a0 - a1)).clamp_min(min=0)                                        # :36-41

    def loss(self, nodes, labels, train_flag=True):
        labels = torch.as_tensor(labels, device=self.weight.device).long()
        label_scores, affinity = self.forward(nodes, labels, train_flag)
        loss_cls = self.xent(label_scores, labels.squeeze())
        loss_constraint = self.affinity(affinity, labels)
        return loss_cls + 5 * loss_constraint, loss_constraint
