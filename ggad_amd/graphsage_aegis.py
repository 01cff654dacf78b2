"""Mini-batch AEGIS-style comparison model of the reference (`src/graphsage_aegis.py`), on the GGAD kernels.

The reference's DGraph AEGIS baseline aggregates the batch sub-graph ONCE per table with the 1-hop half of GGAD's aggregator
(symmetric normalisation, `:194-226`) -- the frozen feature table and a fixed N x F table of standard-normal "noise features" drawn
at construction (`:185-186`) --, projects both with the same weight, relu(W agg^T) (`:307-308`), and trains a discriminator to tell the
two apart (`:310-318`): loss_dis = BCE(sigmoid(D(real ++ noise)), 0...0 1...1), loss_g = BCE(sigmoid(D(noise)), 0) (`:169-173`).
Same class names, signatures and parameter names:

    GCNAggregator(features, features_data, cuda=False, gcn=False).forward(nodes, to_neighs) -> (to_feats, to_noise_feats)
    GCNEncoder(features, feature_dim, embed_dim, adj_lists, aggregator, ...).forward(nodes)  -> (logits_all (2B, 1), logits_gen (B, 1), label (2B))
    GCN(num_classes, enc): .forward(nodes) / .to_prob(nodes) / .loss(nodes) -> (loss_dis, loss_g)

**Parity is unpinned for this model.**  Its three MLPs are `torch_geometric.nn.MLP` (requirements.txt pins torch_geometric 2.1.0), a
library that is absent from this image, so no vectors of the reference could be captured.  `MLP` below restates the published layer
stack of that class for the arguments the reference passes (`in/hidden/out_channels`, `num_layers`, `dropout=0`, a callable `act`; the
defaults `batch_norm=True`, `act_first=False`, `plain_last=True`, `bias=True`):  [Linear -> BatchNorm1d -> act -> dropout] x (L - 1) ->
Linear, with PyG's `Linear` initialisation (kaiming-uniform with a = sqrt(5) over fan_in, bias U(+-1/sqrt(fan_in)): the draws of
`torch.nn.Linear`) and its parameter names (`lins.k.weight`, `norms.k.module.weight`, ...).  The tests compare the HIP path with the
oracle's restatement of the same stack (tests/test_baselines_gpu.py), not with the reference.

The aggregation is the plan + `ggad_mb_gather1` kernels of the GGAD path (`BatchChunk`, x1), once per table; every projection and
MLP layer runs on the exact-f32 MFMA GEMM with autograd (`LinearFn`); batch norm, the sigmoids and the two BCE means are torch
elementwise kernels on a (2B, 64) matrix.  No CPU path.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from ._lib import call, ptr
from .fullgraph import LinearFn
from .graph import DeviceGraph
from .graphsage import Encoder, FeatureTable, MeanAggregator, _as_graph, _features, _node_array  # noqa: F401  (re-exports)
from .minibatch import BatchChunk


class _BatchNorm(nn.Module):
    """`torch_geometric.nn.norm.BatchNorm`: a wrapper whose parameters live under `.module` (the state_dict names of PyG 2.1)."""

    def __init__(self, channels: int):
        super().__init__()
        self.module = nn.BatchNorm1d(channels)

    def forward(self, x):
        return self.module(x)


class MLP(nn.Module):
    """The layer stack of `torch_geometric.nn.MLP` (2.1.0) for the reference's call sites (`src/graphsage_aegis.py:268-290`)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout=0.0, act=F.relu):
        super().__init__()
        chans = [in_channels] + [hidden_channels] * (num_layers - 1) + [out_channels]
        self.lins = nn.ModuleList([nn.Linear(a, b) for a, b in zip(chans[:-1], chans[1:])])
        self.norms = nn.ModuleList([_BatchNorm(c) for c in chans[1:-1]])
        self.dropout = float(dropout)
        self.act = act

    def forward(self, x):
        for lin, norm in zip(self.lins[:-1], self.norms):
            x = LinearFn.apply(x, lin.weight, False) + lin.bias
            x = norm(x)
            x = self.act(x)
            x = F.dropout(x, p=self.dropout, training=self.training)
        return LinearFn.apply(x, self.lins[-1].weight, False) + self.lins[-1].bias


class GCNAggregator(nn.Module):
    """(to_feats, to_noise_feats) = D_r^-1/2 M D_c^-1/2 (X, Z) over the batch sub-graph (`src/graphsage_aegis.py:176-226`); Z is the
    fixed N x F standard-normal table of `:185-186`, drawn from the CPU generator at construction like there."""

    def __init__(self, features, features_data, cuda=False, gcn=False):
        super().__init__()
        self.features = _features(features)
        self.noise_dim = int(np.asarray(features_data).shape[1])
        self.noise = torch.randn(int(np.asarray(features_data).shape[0]), self.noise_dim)     # :186 (CPU draw, then resident in HBM)
        self.cuda = cuda
        self.gcn = gcn
        self._chunks = {}
        self._noise_dev = None

    def noise_table(self):
        dev = self.features.weight.device
        if self._noise_dev is None or self._noise_dev.device != dev:
            self._noise_dev = self.noise.to(dev).contiguous()
        return self._noise_dev

    def _chunk(self, graph: DeviceGraph, table: torch.Tensor, tag: str, max_batches: int) -> BatchChunk:
        key = (id(graph), tag, int(max_batches))
        ch = self._chunks.get(key)
        if ch is None:
            ch = BatchChunk(graph, table, 64, max_batches, 256 * max_batches, 8192 * max_batches, train=False)
            self._chunks[key] = ch
        return ch

    def aggregate(self, batches, adj_lists, max_batches: int = 1):
        """Neighbourhoods from the device CSR; several batches per plan.  Returns (x_feat, x_noise, batch_ptr): views valid until
        the next call."""
        graph = _as_graph(adj_lists, self.features.weight.shape[0], self.features.weight.device)
        nb = max(max_batches, len(batches))
        nodes = [_node_array(b) for b in batches]
        outs = []
        for tag, table in (("x", self.features.weight.data), ("z", self.noise_table())):
            ch = self._chunk(graph, table, tag, nb)
            ch.build(nodes)
            outs.append(ch.x1[:ch.n_rows * ch.F].view(ch.n_rows, ch.F))
        return outs[0], outs[1], ch.batch_ptr_host

    def forward(self, nodes, to_neighs):
        """Explicit neighbour sets, as the reference passes them (`:296`): the ragged weighted gather kernel, once per table."""
        nodes = _node_array(nodes)
        samp = [set(tn).union({int(nodes[i])}) for i, tn in enumerate(to_neighs)]
        sizes = np.fromiter((len(s) for s in samp), dtype=np.int64, count=len(samp))
        seg_ptr = np.zeros(len(samp) + 1, dtype=np.int32)
        np.cumsum(sizes, out=seg_ptr[1:])
        cols = np.fromiter((v for s in samp for v in sorted(s)), dtype=np.int64, count=int(seg_ptr[-1]))
        _, inv = np.unique(cols, return_inverse=True)
        col_cnt = np.bincount(inv).astype(np.float32)[inv]
        row_cnt = np.repeat(sizes.astype(np.float32), sizes)
        w = ((np.float32(1.0) / np.sqrt(row_cnt)) / np.sqrt(col_cnt)).astype(np.float32)        # mask.div(row).div(col)  :212-214
        dev = self.features.weight.device
        f = self.features.weight.shape[1]
        sp, sc, sw = (torch.from_numpy(a).to(dev) for a in (seg_ptr, cols.astype(np.int32), w))
        outs = []
        for table in (self.features.weight.data, self.noise_table()):
            out = torch.empty(len(samp), f, device=dev)
            call("ggad_seg_wsum", ptr(table), f, ptr(sp), ptr(sc), ptr(sw), len(samp), ptr(out))
            outs.append(out)
        return outs[0], outs[1]


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
        init.xavier_uniform_(w)                                                  # :258-260, same RNG draws in the same order
        self.weight = nn.Parameter(w.to(dev))
        self.fc = nn.Linear(embed_dim, feature_dim, bias=False).to(dev)          # :264 (never used in forward)
        noise_dim, hid_dim, num_layers, in_dim = 16, 64, 4, 64                   # :265-270
        generator_layers, encoder_layers = math.floor(num_layers / 2), math.ceil(num_layers / 2)
        self.generator = MLP(noise_dim, hid_dim, in_dim, generator_layers, 0.0, F.relu).to(dev)           # unused in forward
        self.discriminator = MLP(in_dim, hid_dim, hid_dim, encoder_layers, 0.0, F.relu).to(dev)           # unused in forward
        self.discriminator2 = MLP(in_dim, hid_dim, 1, encoder_layers, 0.0, torch.sigmoid).to(dev)
        if embed_dim != in_dim:
            raise ValueError("the reference's discriminator reads 64 channels: emb_size must be 64 (`src/graphsage_aegis.py:270`)")

    def discriminate(self, x_feat, x_noise):
        """From the two aggregates of a batch: (logits_all, logits_gen, label) of `:307-318`."""
        combined = LinearFn.apply(x_feat, self.weight, True)                     # relu(W agg^T)^T: (B, embed)
        combined_noise = LinearFn.apply(x_noise, self.weight, True)
        emb_all = torch.cat([combined, combined_noise], 0)
        label = torch.cat([torch.zeros(combined.shape[0], device=combined.device), torch.ones(combined_noise.shape[0], device=combined.device)])
        logits_all = torch.sigmoid(self.discriminator2(emb_all))
        logits_gen = torch.sigmoid(self.discriminator2(combined_noise))
        return logits_all, logits_gen, label

    def forward(self, nodes):
        x_feat, x_noise, _ = self.aggregator.aggregate([nodes], self.adj_lists)
        return self.discriminate(x_feat, x_noise)


class GCN(nn.Module):
    def __init__(self, num_classes, enc):
        super().__init__()
        self.enc = enc
        self.xent = nn.BCEWithLogitsLoss(reduction="none", pos_weight=torch.tensor([1]))
        w = torch.empty(1, enc.embed_dim)
        init.xavier_uniform_(w)                                                  # :134-135 (never receives a gradient)
        self.weight = nn.Parameter(w.to(enc.weight.device))

    def forward(self, nodes):
        return self.enc(nodes)

    def to_prob(self, nodes):
        logits, logits_gen, label = self.forward(nodes)
        return logits[:int(len(logits) / 2)]                                      # :143-145: the real nodes' half

    @staticmethod
    def losses(logits, logits_noise, label):
        loss_dis = F.binary_cross_entropy(logits[:, 0], label)                    # :170
        loss_g = F.binary_cross_entropy(logits_noise[:, 0], torch.zeros_like(logits_noise[:, 0]))       # :171
        return loss_dis, loss_g

    def loss(self, nodes):
        return self.losses(*self.forward(nodes))

    def loss_rows(self, x_feat, x_noise):
        """The same pair from already aggregated rows (a slice of a multi-batch plan): what the chunked trainer calls."""
        return self.losses(*self.enc.discriminate(x_feat, x_noise))
