"""`ModelHandler` of the mini-batch DOMINANT-style comparison model -- drop-in for `src/model_handler_dominate.py`.

    ModelHandler(config).train() -> None          (prints per-epoch loss / time and the validation AUROC / AP)

Same config keys as the GGAD handler (`src/dgraph.yml`), the reference's split (15 % of the real anomalies contaminate the
training list, 10 % of the labelled normals are relabelled, `:29-56`), its batch schedule (one in-place `random.shuffle` of
the whole training list per epoch, the first 150 slices of `batch_size`, `:134-151`), Adam with weight decay.  Where the work
runs differs: one plan of the epoch's 150 batch sub-graphs + 1-hop aggregates (the GGAD plan / gather kernels), then per batch
two MFMA projections, the reconstruction kernel and the flat Adam kernel; validation scores a thousand slices per plan.

Extra, optional config keys: ``device``, ``num_batches`` (default = the reference's hard override), ``data`` =
(adj_lists | DeviceGraph | (rowptr, col), feat_data, labels).  Results: ``self.epoch_losses``, ``self.epoch_times``,
``self.valid_history`` [(epoch, auc, ap)].
"""
from __future__ import annotations

import argparse
import random
import time

import numpy as np
import torch
import torch.nn as nn

from . import graphsage_dominant as _model
from .dgraph import load_dgraphfin, normalize_features, split_dgraphfin
from .fullgraph import FlatAdam
from .graph import DeviceGraph
from .graphsage import FeatureTable
from .sage_utils import test_recon
from .sampler import PyCompatRandom


class ModelHandler(object):
    model_module = _model
    pseudo_frac = 0.10                  # src/model_handler_dominate.py:43
    default_num_batches = 150           # :140

    def __init__(self, config):
        args = argparse.Namespace(**config)
        data = getattr(args, "data", None)
        if data is not None:
            homo, feat_data, labels = data
            labels = np.array(labels)
        elif args.data_name == "dgraphfin":
            homo, feat_data, labels = load_dgraphfin("../data/dgraphfin.npz", args.data_dir + "dgraphfin_adj_list")
        else:
            raise ValueError("only data_name 'dgraphfin' (or an explicit `data` entry) is supported")
        sp = split_dgraphfin(labels, args.seed, getattr(args, "test_ratio", 0.67), real_frac=0.15, pseudo_frac=self.pseudo_frac)
        labels = sp["labels"]
        print(f"Run on {args.data_name}, postive/total num: {np.sum(labels)}/{len(labels)}, train num {len(sp['y_train'])}," +
              f"valid num {len(sp['y_valid'])}, valid positive num {np.sum(sp['y_valid'])} , test num {len(sp['y_test'])}, "
              f"test positive num {np.sum(sp['y_test'])}")
        print(f"Classification threshold: {args.thres}")
        print(f"Feature dimension: {feat_data.shape[1]}")
        feat_data = normalize_features(feat_data)
        print(f"Model: {args.model}, multi-relation aggregator: {args.multi_relation}, emb_size: {args.emb_size}.")
        self.args = args
        self.dataset = {"feat_data": feat_data, "labels": labels, "adj_lists": homo, "homo": homo,
                        "idx_train": sp["idx_train"], "idx_valid": sp["idx_valid"], "idx_test": sp["idx_test"],
                        "y_train": sp["y_train"], "y_valid": sp["y_valid"], "y_test": sp["y_test"],
                        "idx_labeled": sp["idx_labeled"]}

    def build_model(self, dev):
        args = self.args
        feat_data, adj_lists = self.dataset["feat_data"], self.dataset["adj_lists"]
        n, f = feat_data.shape
        nn.Embedding(n, f)                  # the reference's frozen table draws N x F normals before the model is built (:109)
        if isinstance(adj_lists, DeviceGraph):
            graph = adj_lists
        elif isinstance(adj_lists, tuple):
            graph = DeviceGraph(adj_lists[0], adj_lists[1], dev)
        else:
            graph = DeviceGraph.from_adj_lists(adj_lists, n, dev)
        m = self.model_module
        features = FeatureTable(torch.FloatTensor(np.asarray(feat_data, dtype=np.float32)))
        agg_gcn = m.GCNAggregator(features, cuda=True)
        enc_gcn = m.GCNEncoder(features, f, args.emb_size, graph, agg_gcn, gcn=True, cuda=True)
        return graph, features, m.GCN(2, enc_gcn)

    def train(self):
        args = self.args
        if not torch.cuda.is_available():
            raise RuntimeError("ModelHandler.train needs an MI355X: there is no CPU fallback")
        dev = torch.device("cuda", int(getattr(args, "device", torch.cuda.current_device())))
        torch.cuda.set_device(dev)
        graph, features, gnn_model = self.build_model(dev)
        self.model = gnn_model
        enc = gnn_model.enc
        optimizer = FlatAdam([p for p in gnn_model.parameters() if p.requires_grad], lr=args.lr, weight_decay=args.weight_decay)
        idx_train = np.asarray(self.dataset["idx_train"], dtype=np.int64).copy()
        idx_valid, y_valid = self.dataset["idx_valid"], self.dataset["y_valid"]
        num_batches = int(getattr(args, "num_batches", self.default_num_batches))
        bs = int(args.batch_size)
        if (num_batches - 1) * bs >= len(idx_train):
            raise ValueError(f"{num_batches} batches of {bs} do not fit the training list ({len(idx_train)} nodes)")
        rng = PyCompatRandom.from_python_state(random.getstate())
        attr = features.weight.data
        self.epoch_losses, self.epoch_times, self.valid_history = [], [], []
        rows = min(num_batches * bs, len(idx_train))
        target = torch.empty(rows, attr.shape[1], dtype=torch.float32, device=dev)      # static: the captured epoch reads it
        losses = torch.empty(num_batches, dtype=torch.float32, device=dev)
        # epoch 0 runs eagerly (it also creates the Adam state); after it the 150 optimiser steps of an epoch are ONE
        # hipGraph, replayed on the plan buffers of the new epoch (fixed addresses, fixed batch boundaries)
        capture = bool(getattr(args, "capture", True))
        epoch_graph, graph_x1 = None, None

        def run_batches(x1, bp):
            for b in range(num_batches):
                optimizer.zero_grad()
                loss = gnn_model.loss_rows(x1[bp[b]:bp[b + 1]], target[bp[b]:bp[b + 1]])
                loss.backward()
                optimizer.step()
                losses[b] = loss.detach()

        for epoch in range(args.num_epochs):
            rng.shuffle(idx_train)                                               # :136, in place: epochs compound
            t0 = time.time()
            batches = [idx_train[b * bs:min((b + 1) * bs, len(idx_train))] for b in range(num_batches)]
            x1, bp = enc.aggregator.aggregate(batches, graph, num_batches)       # all 150 batch sub-graphs in one plan
            nodes_dev = torch.from_numpy(np.concatenate(batches)).to(dev)
            torch.index_select(attr, 0, nodes_dev, out=target)                   # torch.tensor(feat_data)[batch_nodes]  :155
            if capture and epoch >= 1 and (epoch_graph is None or graph_x1 != x1.data_ptr()):
                torch.cuda.synchronize()
                optimizer.zero_grad()
                epoch_graph, graph_x1 = torch.cuda.CUDAGraph(), x1.data_ptr()
                with torch.cuda.graph(epoch_graph):
                    run_batches(x1, bp)
            if epoch_graph is not None and graph_x1 == x1.data_ptr():
                epoch_graph.replay()
            else:
                run_batches(x1, bp)
            torch.cuda.synchronize()
            epoch_time = time.time() - t0
            l = losses.cpu().numpy().astype(np.float64)
            self.epoch_losses.append(l)
            self.epoch_times.append(epoch_time)
            # the reference prints (last batch loss * 2) / num_batches (`loss += loss.item()` on the tensor, :162-164)
            print(f"Epoch: {epoch}, loss: {2.0 * l[-1] / num_batches},  time: {epoch_time}s")
            if epoch % args.valid_epochs == 0:
                print("Valid at epoch {}".format(epoch))
                auc, ap = test_recon(idx_valid, y_valid, gnn_model, bs, attr, args.thres)
                self.valid_history.append((epoch, auc, ap))
        random.setstate(rng.to_python_state())
        return None
