"""`ModelHandler` of the mini-batch AEGIS-style comparison model -- drop-in for `src/model_handler_aegis.py`.

    ModelHandler(config).train() -> None          (prints per-epoch loss_g / loss_gen / time and the validation AP / AUC)

Same config keys as the GGAD handler (`src/dgraph.yml`), the reference's split (15 % of the real anomalies contaminate the training list,
5 % of the labelled normals are relabelled, `:29-56`), its batch schedule (per epoch the concatenation idx_train + idx_test, shuffled
with python's `random`, the first 100 slices of `batch_size`, `:131-147`), both losses back-propagated before one Adam step
(`:156-158`), validation with `test_aegis` every `valid_epochs` (`:167-169`).  Where the work runs differs: one plan per epoch for its
100 batch sub-graphs and their two 1-hop aggregates (feature table and noise table: the GGAD plan / gather kernels), then per batch
the projections and the discriminator's linears on the MFMA GEMM and the flat Adam kernel.

Parity of this model is unpinned (see `ggad_amd/graphsage_aegis.py`: `torch_geometric.nn.MLP` is restated, not imported).
Extra, optional config keys: ``device``, ``num_batches`` (default = the reference's hard override 100), ``data`` = (adj_lists |
DeviceGraph | (rowptr, col), feat_data, labels).  Results: ``self.epoch_losses`` [(loss_g, loss_gen) per batch], ``self.epoch_times``,
``self.valid_history`` [(epoch, auc, ap)].
"""
from __future__ import annotations

import random
import time

import numpy as np
import torch
import torch.nn as nn

from . import graphsage_aegis as _model
from .fullgraph import FlatAdam
from .graph import DeviceGraph
from .graphsage import FeatureTable
from .model_handler_dominate import ModelHandler as _Base
from .sage_utils import test_aegis
from .sampler import PyCompatRandom


class ModelHandler(_Base):
    model_module = _model
    pseudo_frac = 0.05                  # src/model_handler_aegis.py:43
    default_num_batches = 100           # :136

    def build_model(self, dev):
        args = self.args
        feat_data, adj_lists = self.dataset["feat_data"], self.dataset["adj_lists"]
        n, f = feat_data.shape
        nn.Embedding(n, f)                  # the reference's frozen table draws N x F normals before the model is built (:105)
        if isinstance(adj_lists, DeviceGraph):
            graph = adj_lists
        elif isinstance(adj_lists, tuple):
            graph = DeviceGraph(adj_lists[0], adj_lists[1], dev)
        else:
            graph = DeviceGraph.from_adj_lists(adj_lists, n, dev)
        m = self.model_module
        features = FeatureTable(torch.FloatTensor(np.asarray(feat_data, dtype=np.float32)))
        agg_gcn = m.GCNAggregator(features, feat_data, cuda=True)                                    # :110 (draws the noise table)
        enc_gcn = m.GCNEncoder(features, f, args.emb_size, graph, agg_gcn, gcn=True, cuda=True)     # :111-112
        return graph, features, m.GCN(2, enc_gcn)

    def train(self):
        args = self.args
        if not torch.cuda.is_available():
            raise RuntimeError("ModelHandler.train needs an MI355X: there is no CPU fallback")
        dev = torch.device("cuda", int(getattr(args, "device", torch.cuda.current_device())))
        torch.cuda.set_device(dev)
        graph, features, gnn_model = self.build_model(dev)
        gnn_model.to(dev)
        self.model = gnn_model
        enc = gnn_model.enc
        optimizer = FlatAdam([p for p in gnn_model.parameters() if p.requires_grad], lr=args.lr, weight_decay=args.weight_decay)
        base = np.concatenate([np.asarray(self.dataset["idx_train"], dtype=np.int64), np.asarray(self.dataset["idx_test"], dtype=np.int64)])
        idx_valid, y_valid = self.dataset["idx_valid"], self.dataset["y_valid"]
        num_batches = int(getattr(args, "num_batches", self.default_num_batches))
        bs = int(args.batch_size)
        if (num_batches - 1) * bs >= len(base):
            raise ValueError(f"{num_batches} batches of {bs} do not fit idx_train + idx_test ({len(base)} nodes)")
        rng = PyCompatRandom.from_python_state(random.getstate())
        self.epoch_losses, self.epoch_times, self.valid_history = [], [], []
        losses = torch.empty(num_batches, 2, dtype=torch.float32, device=dev)
        for epoch in range(args.num_epochs):
            sampled = base.copy()                                                # :132 a fresh concatenation every epoch
            rng.shuffle(sampled)                                                 # :133
            t0 = time.time()
            batches = [sampled[b * bs:min((b + 1) * bs, len(sampled))] for b in range(num_batches)]
            x_feat, x_noise, bp = enc.aggregator.aggregate(batches, graph, num_batches)      # all batch sub-graphs, both tables
            for b in range(num_batches):
                optimizer.zero_grad()
                loss_g, loss_gen = gnn_model.loss_rows(x_feat[bp[b]:bp[b + 1]], x_noise[bp[b]:bp[b + 1]])
                (loss_g + loss_gen).backward()                                   # :156-157: two backward passes into the same .grad
                optimizer.step()
                losses[b, 0], losses[b, 1] = loss_g.detach(), loss_gen.detach()
            torch.cuda.synchronize()
            epoch_time = time.time() - t0
            l = losses.cpu().numpy().astype(np.float64)
            self.epoch_losses.append(l)
            self.epoch_times.append(epoch_time)
            print(f"Epoch: {epoch}, loss_g: {l[:, 0].sum() / num_batches}, loss_gen: {l[:, 1].sum() / num_batches}, time: {epoch_time}s")
            if epoch % args.valid_epochs == 0:
                print("Valid at epoch {}".format(epoch))
                auc, ap = test_aegis(idx_valid, y_valid, gnn_model, bs, args.thres)
                self.valid_history.append((epoch, auc, ap))
        random.setstate(rng.to_python_state())
        return None
