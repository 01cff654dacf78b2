"""Drop-in replacements for the reference's `src/graphsage.py` classes, backed by libggad_hip.so.

Same class names, constructor / ``forward`` signatures, return arities and state_dict keys as the
reference, so `model_handler.py`-style code keeps working:

    GCNAggregator(features, cuda=False, gcn=False).forward(nodes, to_neighs, adj_list, train_flag)
        -> (to_feats (B,F), to_feats_neigh (U,F) | None, mask_row (B,U))        reference :280,295,360
    GCNEncoder(features, feature_dim, embed_dim, adj_lists, aggregator, ...).forward(nodes, label, train_flag)
        -> (combined_all (D,B), to_feats_neigh (B,D), anomaly_feat (D,A), anomaly_feat_new (D,A))   :368,395,454
    GCN(num_classes, enc): .forward / .to_prob / .loss -> (total, cls, margin, rec)                 :163-258
    MeanAggregator / Encoder / GraphSage                                                            :19-154

Differences that are deliberate and documented:
  * all arithmetic runs on the GPU in HIP kernels; there is no CPU path (the library must load);
  * the graph is converted ONCE to a device CSR (``adj_lists`` may also be a ``DeviceGraph``);
  * the column order of U (``unique_nodes_list``) is the kernels' owner order, not CPython's
    set-iteration order; every returned tensor is consistent with that order;
  * ``GCN.loss`` runs the fused kernel chain (forward + loss + backward in one native call) and hands
    the gradients to autograd; ``loss.backward(); optimizer.step()`` works with any torch optimizer.
    The trainer in `ggad_amd.model_handler` skips autograd altogether and uses the in-kernel Adam.
"""
from __future__ import annotations

import random

import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

from . import _lib
from ._lib import call, ptr
from .graph import DeviceGraph
from .fullgraph import LinearFn
from .minibatch import BatchChunk, MiniBatchEngine

_GRAPH_CACHE = {}


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.GgadLibraryError("GGAD HIP modules need a GPU: there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _as_graph(adj_lists, n_nodes: int, device) -> DeviceGraph:
    if isinstance(adj_lists, DeviceGraph):
        return adj_lists
    key = (id(adj_lists), str(device))
    g = _GRAPH_CACHE.get(key)
    if g is None or g[0] is not adj_lists:
        g = (adj_lists, DeviceGraph.from_adj_lists(adj_lists, n_nodes, device))
        _GRAPH_CACHE[key] = g
    return g[1]


class FeatureTable(nn.Module):
    """Frozen node-feature table on the device; stands in for the reference's frozen ``nn.Embedding``
    (`src/model_handler.py:263-264`) and keeps its state_dict key (``...features.weight``)."""

    def __init__(self, weight):
        super().__init__()
        w = weight.weight if isinstance(weight, (nn.Embedding, FeatureTable)) else weight
        w = torch.as_tensor(np.asarray(w) if not isinstance(w, torch.Tensor) else w.detach(), dtype=torch.float32)
        self.weight = nn.Parameter(w.to(_device()).contiguous(), requires_grad=False)

    def forward(self, index):
        return self.weight[torch.as_tensor(index, device=self.weight.device, dtype=torch.long)]


def _features(features) -> FeatureTable:
    return features if isinstance(features, FeatureTable) else FeatureTable(features)


def _node_array(nodes) -> np.ndarray:
    if isinstance(nodes, torch.Tensor):
        nodes = nodes.detach().cpu().numpy()
    return np.asarray(nodes, dtype=np.int64).reshape(-1)


def _label_array(label, n) -> np.ndarray:
    if label is None:
        return np.zeros(n, dtype=np.int64)
    if isinstance(label, torch.Tensor):
        label = label.detach().cpu().numpy()
    return np.asarray(label, dtype=np.int64).reshape(-1)


# ------------------------------------------------------------------------------------------------
class GCNAggregator(nn.Module):
    """1-hop + 2-hop batch-normalised aggregation (reference `src/graphsage.py:275-360`)."""

    def __init__(self, features, cuda=False, gcn=False):
        super().__init__()
        self.features = _features(features)
        self.cuda = cuda          # kept for signature compatibility; the HIP path is always on the GPU
        self.gcn = gcn
        self._chunks = {}

    def _chunk(self, graph: DeviceGraph, embed_dim: int, train: bool) -> BatchChunk:
        key = (id(graph), embed_dim, train)
        ch = self._chunks.get(key)
        if ch is None:
            ch = BatchChunk(graph, self.features.weight.data, embed_dim, 1, 256, 8192, train=train)
            self._chunks[key] = ch
        return ch

    def plan(self, nodes, adj_list, train_flag, labels=None, embed_dim: int = 64) -> BatchChunk:
        graph = _as_graph(adj_list, self.features.weight.shape[0], self.features.weight.device)
        ch = self._chunk(graph, embed_dim, bool(train_flag))
        nodes = _node_array(nodes)
        ch.build([nodes], [_label_array(labels, len(nodes))] if train_flag else None)
        return ch

    def forward(self, nodes, to_neighs, adj_list, train_flag):
        """``to_neighs`` (the per-node neighbour sets) is accepted for compatibility; neighbourhoods are
        read from the device CSR of ``adj_list``, which is where the reference got them from (`:404`)."""
        ch = self.plan(nodes, adj_list, train_flag)
        return self.export(ch, train_flag)

    def export(self, ch: BatchChunk, train_flag):
        b, f = ch.n_rows, ch.F
        e = ch.n_ents
        to_feats = ch.x1[:b * f].view(b, f).clone()
        ent_own = ch.ent_own[:e].long()
        ent_row = ch.ent_row[:e].long()
        is_owner = ent_own == torch.arange(e, device=ent_own.device)
        owners = torch.nonzero(is_owner).reshape(-1)
        upos = torch.cumsum(is_owner.long(), 0) - 1                  # position of every owner entry in U
        r = (ch.ent_ptr[1:b + 1] - ch.ent_ptr[:b]).float()
        mask_row = torch.zeros(b, owners.numel(), device=to_feats.device)
        mask_row[ent_row, upos[ent_own]] = (1.0 / r)[ent_row]
        to_feats_neigh = ch.x2[:e * f].view(e, f)[owners].clone() if train_flag else None
        self.last_unique = ch.ent_col[:e].long()[owners]            # node ids of the U columns (owner order)
        return to_feats, to_feats_neigh, mask_row


class _EncoderRows(torch.autograd.Function):
    """h1 = relu(W x1), nbar = mean_{N(i)+i} relu(W x2), gen = relu(fc nbar) on label-1 rows -- HIP forward and VJP."""

    @staticmethod
    def forward(ctx, weight, fc_weight, enc, ch):
        eng = enc.engine
        eng.sync_params()
        eng.forward_batch(ch, 0)
        b, d = ch.n_rows, eng.D
        ctx.enc, ctx.ch = enc, ch
        ctx.gen_stamp = ch.build_count
        h1 = ch.h1[:b * d].view(b, d).clone()
        nbar = ch.nbar[:b * d].view(b, d).clone()
        lab = ch.labels[:b] == 1
        gen = torch.where(lab[:, None], ch.gen[:b * d].view(b, d), torch.zeros((), device=h1.device))
        return h1, nbar, gen

    @staticmethod
    def backward(ctx, d_h1, d_nbar, d_gen):
        enc, ch = ctx.enc, ctx.ch
        if ch.build_count != ctx.gen_stamp:
            raise RuntimeError("the batch plan was rebuilt between forward and backward")
        eng = enc.engine
        b, d, f = ch.n_rows, eng.D, eng.F
        e0, e1 = ch.batch_ents(0)
        d_h1 = d_h1.contiguous().float()
        d_nbar = d_nbar.contiguous().float()
        d_gen = d_gen.contiguous().float()
        call("ggad_mb_row_coefs", ptr(eng.params), d, f, ptr(ch.labels), ptr(ch.ent_ptr), 0, b, ptr(ch.h1), ptr(ch.gen),
             ptr(d_h1), ptr(d_gen), ptr(d_nbar), ptr(ch.dz), ptr(ch.coef_a), ptr(ch.coef_g))
        call("ggad_mb_bwd_flat", d, f, ptr(ch.x1), ptr(ch.x2), ptr(eng.h2), ptr(ch.ent_own), ptr(ch.ent_row), 0, b, e0,
             e1 - e0, ptr(ch.coef_a), ptr(ch.coef_g), ptr(eng.dw_part))
        lab = ch.labels[:b]
        n1 = int((lab == 1).sum())
        losses8 = torch.zeros(8, device=d_h1.device)
        losses8[6], losses8[7] = float(b - n1), float(n1)
        ws = torch.zeros(int(eng.lib.ggad_mb_loss_workspace_elems(b)), device=d_h1.device)
        grads = torch.empty(eng.n_train, device=d_h1.device)
        call("ggad_mb_grad_reduce", d, f, ptr(ch.pos_meta), 0, b, ptr(losses8), ptr(ch.nbar), ptr(eng.dw_part), ptr(ch.dz),
             ptr(ws), ptr(grads))
        return grads[d:d + d * f].view(d, f), grads[d + d * f:].view(d, d), None, None


class GCNEncoder(nn.Module):
    """GCN encoder with outlier generation (reference `src/graphsage.py:363-454`)."""

    def __init__(self, features, feature_dim, embed_dim, adj_lists, aggregator, num_sample=10, base_model=None,
                 gcn=False, cuda=False, feature_transform=False):
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
        self.engine = MiniBatchEngine(feature_dim, embed_dim, dev)
        # same CPU RNG consumption as the reference constructor (:388-391): xavier for weight, Linear default for fc
        w = torch.empty(embed_dim, feature_dim)
        init.xavier_uniform_(w)
        fc = nn.Linear(embed_dim, embed_dim, bias=False)
        self.weight = nn.Parameter(self.engine.enc_weight)
        self.fc = nn.Linear(embed_dim, embed_dim, bias=False, device="meta")
        self.fc.weight = nn.Parameter(self.engine.enc_fc_weight)
        with torch.no_grad():
            self.weight.copy_(w)
            self.fc.weight.copy_(fc.weight)
        self.engine.sync_params()

    def plan(self, nodes, label, train_flag) -> BatchChunk:
        return self.aggregator.plan(nodes, self.adj_lists, train_flag, label, self.embed_dim)

    def forward(self, nodes, label, train_flag):
        ch = self.plan(nodes, label, train_flag)
        if not train_flag:
            b, d = ch.n_rows, self.embed_dim
            self.engine.sync_params()
            out = torch.empty(b * d, device=self.weight.device)
            call("ggad_mb_encode", ptr(self.engine.params), d, self.feat_dim, ptr(ch.x1), b, ptr(out))
            return out.view(b, d).t(), None, None, None
        lab = torch.as_tensor(_label_array(label, ch.n_rows), device=self.weight.device)
        h1, nbar, gen = _EncoderRows.apply(self.weight, self.fc.weight, self, ch)
        combined = h1.t()                                              # (D, B)      reference :412
        anomaly_feat = combined[:, lab == 1]                           # :427
        anomaly_feat_new = gen[lab == 1]                               # (A, D)      :430
        combined_all = torch.cat((combined[:, lab == 0], anomaly_feat_new.t()), 1)   # :450
        return combined_all, nbar, anomaly_feat, anomaly_feat_new.t()


class _FusedBatchLoss(torch.autograd.Function):
    """GCN.loss through the fused native chain; gradients of `total` are produced in the forward."""

    @staticmethod
    def forward(ctx, weight, enc_weight, fc_weight, model, ch):
        eng = model.enc.engine
        eng.sync_params()
        eng.loss_and_grads(ch, 0, 0)
        d, f = eng.D, eng.F
        g = eng.grads.clone()
        ctx.save_for_backward(g)
        ctx.dims = (d, f)
        l = eng.loss_log[:4].clone()
        return l[0], l[1], l[2], l[3]

    @staticmethod
    def backward(ctx, g_total, g_cls, g_margin, g_rec):
        (g,) = ctx.saved_tensors
        d, f = ctx.dims
        for extra in (g_cls, g_margin, g_rec):
            if extra is not None and bool((extra != 0).any()):
                raise RuntimeError("only the total loss of GCN.loss is differentiable in the fused HIP path")
        g = g * g_total
        return g[:d].view(1, d), g[d:d + d * f].view(d, f), g[d + d * f:].view(d, d), None, None


class GCN(nn.Module):
    """GGAD mini-batch model (reference `src/graphsage.py:157-272`)."""

    def __init__(self, num_classes, enc):
        super().__init__()
        self.enc = enc
        self.xent = nn.BCEWithLogitsLoss(reduction="none", pos_weight=torch.tensor([1]))
        w = torch.empty(1, enc.embed_dim)
        init.xavier_uniform_(w)                                        # same RNG draw as the reference (:168-169)
        self.weight = nn.Parameter(enc.engine.weight)
        with torch.no_grad():
            self.weight.copy_(w)

    def forward(self, nodes, label, train_flag):
        embeds, to_feats_neigh, anomaly_feat, anomaly_feat_new = self.enc(nodes, label, train_flag)
        scores = self.weight.mm(embeds)
        return scores.t(), to_feats_neigh, embeds, anomaly_feat, anomaly_feat_new

    def to_prob(self, nodes, label=None):
        """sigmoid scores of one reference batch, (B,1)  (`:178-181`)."""
        ch = self.enc.plan(nodes, None, False)
        eng = self.enc.engine
        eng.sync_params()
        out = torch.empty(ch.n_rows, dtype=torch.float32, device=self.weight.device)
        eng.score_chunk(ch, out)
        return out.view(-1, 1)

    def loss(self, nodes, labels):
        """(total, cls, margin, rec); ``total.backward()`` fills .grad of weight / enc.weight / enc.fc.weight."""
        lab = _label_array(labels, len(_node_array(nodes)))
        ch = self.enc.plan(nodes, lab, True)
        return _FusedBatchLoss.apply(self.weight, self.enc.weight, self.enc.fc.weight, self, ch)


# ------------------------------------------------------------------------------------------------
# Vanilla GraphSAGE pieces (reference :19-154).  Module-level parity only: the reference's own
# ModelHandler cannot train them.
class MeanAggregator(nn.Module):
    def __init__(self, features, cuda=False, gcn=False):
        super().__init__()
        self.features = _features(features)
        self.cuda = cuda
        self.gcn = gcn

    def forward(self, nodes, to_neighs, num_sample=10):
        """Mean of (sampled) neighbour features; sampling uses python's ``random.sample`` exactly like the
        reference (`:75-78`), the aggregation is the HIP segment-mean kernel."""
        nodes = _node_array(nodes)
        if num_sample is not None:
            # tuple(set): what CPython <= 3.10 does internally for a set population (and valid on >= 3.11)
            samp = [set(random.sample(tuple(tn), num_sample)) if len(tn) >= num_sample else tn for tn in to_neighs]
        else:
            samp = to_neighs
        if self.gcn:
            samp = [s.union({int(nodes[i])}) for i, s in enumerate(samp)]
        sizes = np.fromiter((len(s) for s in samp), dtype=np.int64, count=len(samp))
        seg_ptr = np.zeros(len(samp) + 1, dtype=np.int32)
        np.cumsum(sizes, out=seg_ptr[1:])
        seg_col = np.fromiter((v for s in samp for v in sorted(s)), dtype=np.int32, count=int(seg_ptr[-1]))
        dev = self.features.weight.device
        f = self.features.weight.shape[1]
        sp, sc = torch.from_numpy(seg_ptr).to(dev), torch.from_numpy(seg_col).to(dev)
        out = torch.empty(len(samp), f, device=dev)
        call("ggad_seg_mean", ptr(self.features.weight.data), f, ptr(sp), ptr(sc), len(samp), ptr(out))
        return out


class Encoder(nn.Module):
    def __init__(self, features, feature_dim, embed_dim, adj_lists, aggregator, num_sample=10, base_model=None,
                 gcn=False, cuda=False, feature_transform=False):
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
        w = torch.empty(embed_dim, self.feat_dim if self.gcn else 2 * self.feat_dim)
        init.xavier_uniform_(w)
        self.weight = nn.Parameter(w.to(self.features.weight.device))

    def forward(self, nodes):
        nodes_np = _node_array(nodes)
        adj = self.adj_lists
        # the reference hands the adjacency's own set objects to the aggregator (`:137`): `random.sample` walks a set in ITS
        # iteration order, which a copy does not preserve -- so no copy; a CSR graph yields sets filled in ascending order, the
        # construction of `synth.csr_to_adj_lists`
        neigh_sets = [adj[int(v)] for v in nodes_np] if not isinstance(adj, DeviceGraph) else \
            [set(int(c) for c in adj.col_host[adj.rowptr_host[v]:adj.rowptr_host[v + 1]]) for v in nodes_np]
        neigh_feats = self.aggregator.forward(nodes_np, neigh_sets, self.num_sample)
        if not self.gcn:
            combined = torch.cat((self.features(nodes_np), neigh_feats), dim=1)
        else:
            combined = neigh_feats
        # relu(W . combined^T) (`:152`) on the exact-f32 MFMA GEMM with the ReLU epilogue; autograd = two more GEMMs (LinearFn)
        return LinearFn.apply(combined.contiguous(), self.weight, True).t()


class GraphSage(nn.Module):
    def __init__(self, num_classes, enc):
        super().__init__()
        self.enc = enc
        self.xent = nn.CrossEntropyLoss()
        w = torch.empty(num_classes, enc.embed_dim)
        init.xavier_uniform_(w)
        self.weight = nn.Parameter(w.to(enc.weight.device))

    def forward(self, nodes):
        embeds = self.enc(nodes)                                            # (D, B)
        return LinearFn.apply(embeds.t().contiguous(), self.weight, False)  # scores.t() = (weight . embeds)^T   (`:32-35`)

    def to_prob(self, nodes):
        return torch.sigmoid(self.forward(nodes))

    def loss(self, nodes, labels):
        return self.xent(self.forward(nodes), torch.as_tensor(labels, device=self.weight.device).long().squeeze())
