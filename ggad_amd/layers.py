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
