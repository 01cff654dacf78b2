"""Mini-batch DOMINANT-style comparison model of the reference (`src/graphsage_dominant.py`), on the GGAD kernels.

The reference's DGraph baselines reuse the 1-hop half of GGAD's batch aggregator (symmetric normalisation over the batch
sub-graph, `:194-226`), encode with relu(W agg) and decode back to feature space with relu(fc .) (`:264-276`), and train on a
reconstruction error whose inner sum runs over the batch axis (`:154-157`).  Same class names, signatures, parameter names:

    GCNAggregator(features, cuda=False, gcn=False).forward(nodes, to_neighs)        -> to_feats (B, F)
    GCNEncoder(features, feature_dim, embed_dim, adj_lists, aggregator, ...).forward(nodes) -> (B, F)
    GCN(num_classes, enc): .forward(nodes) / .to_prob(nodes, label) / .reconstruction(a, b) / .normalize(emb) / .loss(nodes, features)

The aggregation is the plan + `ggad_mb_gather1` kernels of the GGAD path (`BatchChunk`, x1), the projections run on the
exact-f32 MFMA GEMM with autograd (`LinearFn`), the loss and its gradient in `ggad_recon_cols_f32`.  No CPU path.
`MeanAggregator` / `Encoder` of that file are the ones of `src/graphsage.py` and are re-exported from `ggad_amd.graphsage`.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from ._lib import call, ptr
from .fullgraph import LinearFn
from .graph import DeviceGraph
from .graphsage import Encoder, FeatureTable, MeanAggregator, _as_graph, _features, _node_array  # noqa: F401  (re-exports)
from .minibatch import BatchChunk


class GCNAggregator(nn.Module):
    """to_feats = D_r^-1/2 M D_c^-1/2 X over the batch sub-graph: M[i][u] = 1 for u in N(i) + {i}, row sums r_i, column
    sums c_u counted over the rows of THIS batch (`src/graphsage_dominant.py:194-226`)."""

    def __init__(self, features, cuda=False, gcn=False):
        super().__init__()
        self.features = _features(features)
        self.cuda = cuda          # signature compatibility; the kernels always run on the GPU
        self.gcn = gcn
        self._chunks = {}

    def chunk(self, graph: DeviceGraph, max_batches: int = 1) -> BatchChunk:
        key = (id(graph), int(max_batches))
        ch = self._chunks.get(key)
        if ch is None:
            ch = BatchChunk(graph, self.features.weight.data, 64, max_batches, 256 * max_batches, 8192 * max_batches, train=False)
            self._chunks[key] = ch
        return ch

    def aggregate(self, batches, adj_lists, max_batches: int = 1):
        """Fast path: neighbourhoods from the device CSR; several batches per plan.  Returns (x1 (sum B, F) view that is
        valid until the next call, batch_ptr host array)."""
        graph = _as_graph(adj_lists, self.features.weight.shape[0], self.features.weight.device)
        ch = self.chunk(graph, max(max_batches, len(batches)))
        ch.build([_node_array(b) for b in batches])
        return ch.x1[:ch.n_rows * ch.F].view(ch.n_rows, ch.F), ch.batch_ptr_host

    def forward(self, nodes, to_neighs):
        """Explicit neighbour sets, as the reference passes them (`:266`): the ragged weighted gather kernel."""
        nodes = _node_array(nodes)
        samp = [set(tn).union({int(nodes[i])}) for i, tn in enumerate(to_neighs)]
        sizes = np.fromiter((len(s) for s in samp), dtype=np.int64, count=len(samp))
        seg_ptr = np.zeros(len(samp) + 1, dtype=np.int32)
        np.cumsum(sizes, out=seg_ptr[1:])
        cols = np.fromiter((v for s in samp for v in sorted(s)), dtype=np.int64, count=int(seg_ptr[-1]))
        _, inv = np.unique(cols, return_inverse=True)
        col_cnt = np.bincount(inv).astype(np.float32)[inv]
        row_cnt = np.repeat(sizes.astype(np.float32), sizes)
        w = ((np.float32(1.0) / np.sqrt(row_cnt)) / np.sqrt(col_cnt)).astype(np.float32)        # mask.div(row).div(col)  :212-216
        dev = self.features.weight.device
        f = self.features.weight.shape[1]
        sp, sc, sw = (torch.from_numpy(a).to(dev) for a in (seg_ptr, cols.astype(np.int32), w))
        out = torch.empty(len(samp), f, device=dev)
        call("ggad_seg_wsum", ptr(self.features.weight.data), f, ptr(sp), ptr(sc), ptr(sw), len(samp), ptr(out))
        return out


class GCNEncoder(nn.Module):
    def __init__(self, features, feature_dim, embed_dim, adj_lists, aggregator, num_sample=10, base_model=None, gcn=False,
                 cuda=False, feature_transform=False):
        super().__init__()
        self.features = _features(features)
        self.feat_dim = feature_dim
        self.adj_lists = adj_lists
        self.aggregator = aggregator
        self.aggregator.features = self.features
        self.num_sample = num_sample
        if base_model is not None:
            self.base_model = base_model
        self.gcn = gcn
        self.embed_dim = embed_dim
        self.cuda = cuda
        self.aggregator.cuda = cuda
        dev = self.features.weight.device
        w = torch.empty(embed_dim, self.feat_dim)
        init.xavier_uniform_(w)                                                  # :259-261, same RNG draws in the same order
        self.weight = nn.Parameter(w.to(dev))
        self.fc = nn.Linear(embed_dim, feature_dim, bias=False).to(dev)          # :265

    def decode(self, neigh_feats):
        """relu(fc(relu(W agg^T)^T)) (`:274-276`): (B, F) -> (B, F)."""
        hidden = LinearFn.apply(neigh_feats, self.weight, True)
        return LinearFn.apply(hidden, self.fc.weight, True)

    def forward(self, nodes):
        x1, _ = self.aggregator.aggregate([nodes], self.adj_lists)
        return self.decode(x1)


class _ReconCols(torch.autograd.Function):
    """mean_c sqrt(sum_b w (a - t)^2) and its gradient w.r.t. a, one launch (`ggad_recon_cols_f32`)."""

    @staticmethod
    def forward(ctx, a, t, w_pos: float, w_neg: float):
        a = a.contiguous()
        t = t.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        call("ggad_recon_cols_f32", ptr(a), ptr(t), a.shape[0], a.shape[1], float(w_pos), float(w_neg), ptr(loss), 0,
             ptr(da) if da is not None else 0)
        ctx.da = da
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.da * g if ctx.da is not None else None), None, None, None


class GCN(nn.Module):
    # weights of the squared error where the reconstruction is > 0 / <= 0; (1, 1) = plain error (`:154-157`)
    recon_weights = (1.0, 1.0)

    def __init__(self, num_classes, enc):
        super().__init__()
        self.enc = enc
        self.xent = nn.BCEWithLogitsLoss(reduction="none", pos_weight=torch.tensor([1]))
        w = torch.empty(1, enc.embed_dim)
        init.xavier_uniform_(w)                                                  # :136-137 (never receives a gradient)
        self.weight = nn.Parameter(w.to(enc.weight.device))

    def forward(self, nodes):
        return self.enc(nodes)

    def to_prob(self, nodes, label):
        return self.forward(nodes)

    def _target(self, t, like):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        return t.to(device=like.device, dtype=torch.float32)          # the kernels compute in f32 (baselines)

    def reconstruction(self, anomaly_feat, anomaly_feat_new):
        if anomaly_feat.dim() != 2:
            raise ValueError("reconstruction expects (rows, columns) matrices")
        wp, wn = self.recon_weights
        return _ReconCols.apply(anomaly_feat, self._target(anomaly_feat_new, anomaly_feat), wp, wn)

    def normalize(self, emb):
        """emb / ||emb|| row-wise with 1/0 -> 0 (`:159-164`); not on the training path."""
        inv = torch.pow(torch.norm(emb, dim=-1, keepdim=True), -1)
        inv = torch.where(torch.isinf(inv), torch.zeros_like(inv), inv)
        return emb * inv

    def loss(self, nodes, features):
        return self.reconstruction(self.forward(nodes), features)

    def loss_rows(self, x1_rows, target_rows):
        """Same loss from an already aggregated batch (rows of a multi-batch plan): what the chunked trainer calls."""
        return self.reconstruction(self.enc.decode(x1_rows), target_rows)
