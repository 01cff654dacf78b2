"""`ModelHandler` -- drop-in for the reference's DGraph driver (`src/model_handler.py`).

    ModelHandler(config).train() -> (f1_macro, f1_binary_1, f1_binary_0, auc, gmean)

Same config keys (`src/dgraph.yml`), same split, same batch schedule for equal seeds, same prints,
same checkpoint format (state_dict keys of `GCN`).  What differs is where the work runs: the batch
sub-graphs, the aggregation, the model step and Adam are HIP kernels on the MI355X; the python
`random` stream is continued by the native sampler; validation scores thousands of reference batches
per launch.  ``model: 'GCN'`` is the GGAD model.  ``model: 'SAGE'`` (the vanilla GraphSAGE baseline of
`src/graphsage.py:19-154`) cannot run through the reference's own loop -- `GraphSage.loss` returns one value where the loop
unpacks four (`:358-362`), and `test_sage` calls `to_prob(nodes, None)` where `GraphSage.to_prob` takes one argument
 -- here it is REPAIRED with exactly those two changes (`_train_sage`).

Extra, optional config keys:  ``device`` (cuda index), ``num_batches`` (default 150, the reference's
hard override `:317`), ``data`` = (adj_lists | DeviceGraph | (rowptr, col), feat_data, labels) to bypass
the file loader, ``log_every``.  Under `torch.distributed` (backend nccl = RCCL) batches are dealt
round-robin over the ranks and gradients are all-reduced once per step.
"""
from __future__ import annotations

import argparse
import datetime
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from .dgraph import load_dgraphfin, normalize_features, split_dgraphfin
from .graph import DeviceGraph
from .graphsage import GCN, Encoder, FeatureTable, GCNAggregator, GCNEncoder, GraphSage, MeanAggregator
from .sage_utils import test_sage
from .sampler import PyCompatRandom
from .trainer import BatchSchedule, DGraphTrainer


class ModelHandler(object):

    def __init__(self, config):
        args = argparse.Namespace(**config)
        data = getattr(args, "data", None)
        if data is not None:
            homo, feat_data, labels = data
            labels = np.array(labels)
        elif args.data_name == "dgraphfin":
            homo, feat_data, labels = load_dgraphfin("../data/dgraphfin.npz", args.data_dir + "dgraphfin_adj_list")
        else:
            raise ValueError("only data_name 'dgraphfin' (or an explicit `data` entry) is supported by the GGAD path")
        sp = split_dgraphfin(labels, args.seed, getattr(args, "test_ratio", 0.67))     # model_handler.py:29-30,150-178
        labels = sp["labels"]
        print(f"Run on {args.data_name}, postive/total num: {np.sum(labels)}/{len(labels)}, train num {len(sp['y_train'])}," +
              f"valid num {len(sp['y_valid'])}, valid positive num {np.sum(sp['y_valid'])} , test num {len(sp['y_test'])}, "
              f"test positive num {np.sum(sp['y_test'])}")
        print(f"Classification threshold: {args.thres}")
        print(f"Feature dimension: {feat_data.shape[1]}")
        feat_data = normalize_features(feat_data)                                      # model_handler.py:225
        print(f"Model: {args.model}, multi-relation aggregator: {args.multi_relation}, emb_size: {args.emb_size}.")
        self.args = args
        self.dataset = {"feat_data": feat_data, "labels": labels, "adj_lists": homo, "homo": homo,
                        "idx_train": sp["idx_train"], "idx_valid": sp["idx_valid"], "idx_test": sp["idx_test"],
                        "y_train": sp["y_train"], "y_valid": sp["y_valid"], "y_test": sp["y_test"],
                        "idx_labeled": sp["idx_labeled"], "idx_anomaly": sp["idx_anomaly"]}

    def train(self):
        args = self.args
        if args.model == "SAGE":
            return self._train_sage()
        if args.model == "PCGNN":
            return self._train_pcgnn()
        if args.model != "GCN":
            raise NotImplementedError("models 'GCN' (GGAD), 'SAGE' and 'PCGNN' are trainable")
        if not torch.cuda.is_available():
            raise RuntimeError("ModelHandler.train needs an MI355X: the GGAD hot path has no CPU fallback")
        dist = torch.distributed if (torch.distributed.is_available() and torch.distributed.is_initialized()) else None
        world = dist.get_world_size() if dist else 1
        rank = dist.get_rank() if dist else 0
        dev = torch.device("cuda", int(getattr(args, "device", torch.cuda.current_device())))
        torch.cuda.set_device(dev)
        feat_data, adj_lists = self.dataset["feat_data"], self.dataset["adj_lists"]
        idx_train = self.dataset["idx_train"]
        idx_valid, y_valid, idx_test, y_test = (self.dataset["idx_test"], self.dataset["y_test"],
                                                self.dataset["idx_test"], self.dataset["y_test"])   # :260-261
        n, f = feat_data.shape
        # same RNG consumption as the reference: nn.Embedding's default init draws N x F normals (:263)
        nn.Embedding(n, f)
        if isinstance(adj_lists, DeviceGraph):
            graph = adj_lists
        elif isinstance(adj_lists, tuple):
            graph = DeviceGraph(adj_lists[0], adj_lists[1], dev)
        else:
            # the reference's pickled dict of sets: converted once, then served from a binary CSR cache beside it
            # (keyed by size + mtime of the pickle; written under a per-process temporary name: ranks may convert concurrently)
            cache = src = None
            if getattr(args, "data_name", "") == "dgraphfin" and getattr(args, "data", None) is None:
                src = args.data_dir + "dgraphfin_adj_list"
                cache = os.path.join(args.data_dir, "dgraphfin_adj_list.csr.npz")
            graph = DeviceGraph.from_adj_lists_cached(adj_lists, n, dev, cache, source_path=src)
        features = FeatureTable(torch.FloatTensor(np.asarray(feat_data, dtype=np.float32)))
        agg_gcn = GCNAggregator(features, cuda=True)
        enc_gcn = GCNEncoder(features, f, args.emb_size, graph, agg_gcn, gcn=True, cuda=True)
        gnn_model = GCN(2, enc_gcn)
        engine = enc_gcn.engine
        engine.lr, engine.wd = float(args.lr), float(args.weight_decay)
        engine.sync_params()

        timestamp = datetime.datetime.fromtimestamp(int(time.time())).strftime("%Y-%m-%d %H-%M-%S")
        dir_saver = args.save_dir + timestamp
        path_saver = os.path.join(dir_saver, "{}_{}.pkl".format(args.data_name, args.model))
        f1_mac_best, auc_best, ep_best = 0, 0, -1

        num_batches = int(getattr(args, "num_batches", 150))                            # :317
        rng = PyCompatRandom.from_python_state(random.getstate())
        # data parallel: `dp_sampler: independent` (the default when world > 1) gives every rank a batch stream of its own
        # (seed * 1000003 + rank + 1).  The bit-exact sampler is ONE serial Mersenne-Twister stream (~31 us per batch on the host):
        # dealt to W ranks (`dp_sampler: shared`, rank r takes batches r, r + W, ...) every rank must generate all W batches of a
        # step, which caps an end-to-end run at ~6.4 M nodes/s for ANY W -- below one GPU's steady state.  The trajectory is not the
        # reference's at W > 1 in either mode (global batch W x 150), so nothing is lost; `shared` stays as the opt-in for checks
        # against the W-batch gradient-averaging oracle (tests/test_distributed_cpu.py).  W = 1: the reference's own stream.
        own_stream = world > 1 and str(getattr(args, "dp_sampler", "independent")) != "shared"
        if own_stream:
            rng = PyCompatRandom(int(getattr(args, "seed", 0)) * 1000003 + rank + 1)
        sched = BatchSchedule(idx_train, self.dataset["idx_anomaly"], self.dataset["labels"], args.batch_size, rng,
                              n_pseudo=50, batches_per_epoch=num_batches)
        allreduce = None
        if world > 1:
            def allreduce(t):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        exchange = None
        want_oneshot = os.environ.get("GGAD_EXCHANGE", "oneshot") == "oneshot"
        if world > 1:                                   # rank 0's choice holds for everybody (the hand-shake below is collective)
            flag = torch.tensor([int(want_oneshot)], dtype=torch.int32,
                                device=features.weight.device if dist.get_backend() == "nccl" else "cpu")
            dist.broadcast(flag, src=0)
            want_oneshot = bool(flag.item())
        if world > 1 and want_oneshot:
            # one-shot peer-write exchange inside the gradient / Adam launch; every rank agrees on whether
            # it is usable, else all of them keep the all-reduce
            from .exchange import OneShotExchange
            n_par = int(engine.n_train)
            try:
                exchange = OneShotExchange(rank, world, n_par, features.weight.device)
            except Exception:
                exchange = None
            good = exchange.connect(dist) if exchange is not None else OneShotExchange.decline(dist, features.weight.device)
            if not good:
                exchange = None
        steps_per_epoch = max(1, num_batches // world)
        trainer = DGraphTrainer(graph, features.weight.data, args.emb_size, sched, chunk_batches=steps_per_epoch, rank=rank,
                                world_size=world, allreduce=allreduce, engine=engine, exchange=exchange, own_stream=own_stream)
        self.trainer, self.model = trainer, gnn_model
        self.sweep_ap = []                                      # AP of every sweep, validation sweeps then the test sweep (the reference prints it)
        self.epoch_losses, self.valid_history = [], []          # {total, cls, margin, rec} of every batch; (epoch, five metrics) of every sweep
        trainer.start_stream(steps_per_epoch * args.num_epochs)        # sampler thread alive across the validation pauses
        total_time = 0.0
        epoch = 0
        while epoch < args.num_epochs:
            # run up to (and including) the next validation epoch in one call: the sampler thread then prefetches
            # the following epochs' batch schedules while the GPU trains (reference validates when epoch % valid_epochs == 0)
            last = epoch
            while last % args.valid_epochs != 0 and last + 1 < args.num_epochs:
                last += 1
            n_ep = last - epoch + 1
            t0 = time.time()
            trainer.run_steps(steps_per_epoch * n_ep)
            torch.cuda.synchronize()
            trainer.check_exchange(dist if world > 1 else None)
            block_time = time.time() - t0
            lall = engine.losses(steps_per_epoch * n_ep).astype(np.float64)
            for j in range(n_ep):
                l = lall[j * steps_per_epoch:(j + 1) * steps_per_epoch]
                self.last_epoch_losses = l
                self.epoch_losses.append(l)
                if rank == 0:
                    print(f"Epoch: {epoch + j}, loss: {l[:, 0].mean()}, marigin_loss: {l[:, 2].mean()}, time: {block_time / n_ep}s")
                    print("loss_cls", l[:, 1].mean())
                    print("total_time is", total_time)
                    print("loss_constraint", l[:, 2].mean())
            epoch = last
            if epoch % args.valid_epochs == 0:
                # the validation sweep is sharded over the ranks (contiguous ranges of the reference's batches, one
                # all-reduce of the scores); every rank sees the same metrics, rank 0 prints and saves
                if rank == 0:
                    print("Valid at epoch {}".format(epoch))
                f1_mac_val, f1_1_val, f1_0_val, auc_val, gmean_val = test_sage(idx_valid, y_valid, gnn_model, args.batch_size,
                                                                               args.thres, dist=dist, verbose=rank == 0)
                self.valid_history.append((epoch, (f1_mac_val, f1_1_val, f1_0_val, auc_val, gmean_val)))
                self.sweep_ap.append(getattr(test_sage, "last_ap", None))
                if auc_val > auc_best:
                    f1_mac_best, auc_best, ep_best = f1_mac_val, auc_val, epoch
                    if rank == 0:
                        if not os.path.exists(dir_saver):
                            os.makedirs(dir_saver)
                        print("  Saving model ...")
                        torch.save(gnn_model.state_dict(), path_saver)
            if dist:
                dist.barrier()
            total_time += time.time() - t0
            epoch += 1
        random.setstate(rng.to_python_state())       # hand the stream back to python `random`
        self.end_state = {k: v.detach().clone() for k, v in gnn_model.state_dict().items() if "features" not in k}   # before the restore
        if ep_best >= 0:
            if rank == 0:
                print("Restore model from epoch {}".format(ep_best))
                print("Model path: {}".format(path_saver))
                gnn_model.load_state_dict(torch.load(path_saver))
            if dist and world > 1:
                dist.broadcast(engine.params, src=0)          # every rank tests the restored weights
            engine.sync_params()
        res = test_sage(idx_test, y_test, gnn_model, args.batch_size, args.thres, dist=dist, verbose=rank == 0)
        self.sweep_ap.append(getattr(test_sage, "last_ap", None))
        return res


    # ---------------------------------------------------------------------------------------------------------------- SAGE
    def _train_sage(self):
        """`model: 'SAGE'`: MeanAggregator -> Encoder(gcn=False) -> GraphSage(2, enc) as `src/model_handler.py:278-293` builds
        them, trained by the loop of `:310-370` with the two repairs named in the module docstring.  Host side as in the reference
        (python `random` shuffles and neighbour sampling, python sets); the arithmetic -- segment mean of sampled neighbour
        rows, the two projections and their gradients, Adam -- runs in libggad_hip.so."""
        from .fullgraph import FlatAdam
        args = self.args
        if not torch.cuda.is_available():
            raise RuntimeError("ModelHandler.train needs an MI355X: there is no CPU fallback")
        dev = torch.device("cuda", int(getattr(args, "device", torch.cuda.current_device())))
        torch.cuda.set_device(dev)
        feat_data, adj_lists = self.dataset["feat_data"], self.dataset["adj_lists"]
        idx_train = list(self.dataset["idx_train"])
        idx_valid, y_valid, idx_test, y_test = (self.dataset["idx_test"], self.dataset["y_test"],
                                                self.dataset["idx_test"], self.dataset["y_test"])   # :260-261
        n, f = feat_data.shape
        nn.Embedding(n, f)                                            # RNG consumption of :263
        if isinstance(adj_lists, tuple):
            adj_lists = DeviceGraph(adj_lists[0], adj_lists[1], dev)
        features = FeatureTable(torch.FloatTensor(np.asarray(feat_data, dtype=np.float32)))
        agg_sage = MeanAggregator(features, cuda=True)
        enc_sage = Encoder(features, f, args.emb_size, adj_lists, agg_sage, gcn=False, cuda=True)
        enc_sage.num_samples = 5                                       # :291 (a new attribute: the encoder keeps num_sample = 10)
        gnn_model = GraphSage(2, enc_sage).to(dev)
        features.to(dev)
        optimizer = FlatAdam([p for p in gnn_model.parameters() if p.requires_grad], lr=args.lr, weight_decay=args.weight_decay)
        self.model = gnn_model
        num_batches = int(getattr(args, "num_batches", 150))           # :317
        n_pseudo = int(getattr(args, "n_pseudo", 50))
        idx_anomaly = list(self.dataset["idx_anomaly"])
        labels = self.dataset["labels"]
        timestamp = datetime.datetime.fromtimestamp(int(time.time())).strftime("%Y-%m-%d %H-%M-%S")
        dir_saver = args.save_dir + timestamp
        path_saver = os.path.join(dir_saver, "{}_{}.pkl".format(args.data_name, args.model))
        f1_mac_best, auc_best, ep_best = 0, 0, -1
        total_time = 0.0
        self.sage_losses = []
        for epoch in range(args.num_epochs):
            t_epoch = time.time()
            random.shuffle(idx_train)                                  # :314
            loss_sum, epoch_time = 0.0, 0.0
            for batch in range(num_batches):
                t0 = time.time()
                i0, i1 = batch * args.batch_size, min((batch + 1) * args.batch_size, len(idx_train))
                batch_nodes = idx_train[i0:i1]
                random.shuffle(idx_anomaly)                            # :341
                batch_nodes = batch_nodes + idx_anomaly[:n_pseudo]     # :342,347
                batch_label = labels[np.array(batch_nodes)]
                optimizer.zero_grad()
                loss = gnn_model.loss(batch_nodes, torch.as_tensor(batch_label, device=dev).long())    # repair 1
                loss.backward()
                optimizer.step()
                epoch_time += time.time() - t0
                self.sage_losses.append(float(loss.item()))
                loss_sum += self.sage_losses[-1]
            print(f"Epoch: {epoch}, loss: {loss_sum / num_batches}, time: {epoch_time}s")
            total_time += time.time() - t_epoch
            if epoch % args.valid_epochs == 0:
                print("Valid at epoch {}".format(epoch))
                f1_mac_val, f1_1_val, f1_0_val, auc_val, gmean_val = self._test_graphsage(idx_valid, y_valid, gnn_model, args.batch_size,
                                                                                          args.thres)
                if auc_val > auc_best:
                    f1_mac_best, auc_best, ep_best = f1_mac_val, auc_val, epoch
                    if not os.path.exists(dir_saver):
                        os.makedirs(dir_saver)
                    print("  Saving model ...")
                    torch.save(gnn_model.state_dict(), path_saver)
        if ep_best >= 0:
            print("Restore model from epoch {}".format(ep_best))
            print("Model path: {}".format(path_saver))
            gnn_model.load_state_dict(torch.load(path_saver))
        return self._test_graphsage(idx_test, y_test, gnn_model, args.batch_size, args.thres)

    def _train_pcgnn(self):
        """`model: 'PCGNN'`: IntraAgg x 3 -> InterAgg -> PCALayer(2, inter1, alpha) as `src/model_handler.py:269-277,287-288` builds
        them, trained by the loop of `:310-370` and validated / tested through the `test_pcgnn` call sites of `:392,411`.  The reference
        cannot run this branch: its handler hands the homogeneous adjacency to `InterAgg`, which indexes three relations
        (`src/layers.py:43-45`), and `test_pcgnn` is not defined anywhere.  Two repairs, nothing else:
        (1) `adj_lists` must be a list of THREE relation adjacencies -- config key `relations` (dicts of sets / (rowptr, col) pairs),
        or `data = ([r1, r2, r3], feat, labels)`; (2) `test_pcgnn` = `test_sage`'s protocol on `PCALayer.to_prob(nodes, labels,
        train_flag=False)[0][:, 1]` (the class-1 GNN score).  No vectors of the reference exist for this loop (it never ran): the
        modules are pinned (tests/golden/minibatch_pcgnn.npz), the loop is checked for self-consistency (tests/test_dropin_gpu.py)."""
        from .fullgraph import FlatAdam
        from .layers import InterAgg, IntraAgg, PCALayer
        from . import synth
        args = self.args
        if not torch.cuda.is_available():
            raise RuntimeError("ModelHandler.train needs an MI355X: there is no CPU fallback")
        dev = torch.device("cuda", int(getattr(args, "device", torch.cuda.current_device())))
        torch.cuda.set_device(dev)
        feat_data = self.dataset["feat_data"]
        relations = getattr(args, "relations", None)
        if relations is None and isinstance(self.dataset["adj_lists"], (list, tuple)) and len(self.dataset["adj_lists"]) == 3 \
                and not isinstance(self.dataset["adj_lists"][0], np.ndarray):
            relations = self.dataset["adj_lists"]
        if relations is None or len(relations) != 3:
            raise ValueError("model 'PCGNN' needs three relation graphs: config key `relations` = [r1, r2, r3] "
                             "(dict of neighbour sets, or (rowptr, col)); the reference's branch never ran (its handler passes one)")
        adjs = [synth.csr_to_adj_lists(r[0], r[1]) if isinstance(r, tuple) else r for r in relations]
        idx_train = list(self.dataset["idx_train"])
        idx_valid, y_valid, idx_test, y_test = (self.dataset["idx_test"], self.dataset["y_test"],
                                                self.dataset["idx_test"], self.dataset["y_test"])   # :260-261
        n, f = feat_data.shape
        nn.Embedding(n, f)                                            # RNG consumption of :263
        features = FeatureTable(torch.FloatTensor(np.asarray(feat_data, dtype=np.float32)))
        train_pos = [i for i in idx_train if self.dataset["labels"][i] == 1]          # pos_neg_split(idx_train, y_train)[0]
        rho, alpha = float(getattr(args, "rho", 0.5)), float(getattr(args, "alpha", 2))
        intras = [IntraAgg(features, f, args.emb_size, train_pos, rho, cuda=True) for _ in range(3)]         # :270-275
        inter1 = InterAgg(features, f, args.emb_size, train_pos, adjs, intras, inter=args.multi_relation, cuda=True)    # :276-277
        gnn_model = PCALayer(2, inter1, alpha).to(dev)                 # :288
        features.to(dev)
        optimizer = FlatAdam([p for p in gnn_model.parameters() if p.requires_grad], lr=args.lr, weight_decay=args.weight_decay)
        self.model = gnn_model
        num_batches = int(getattr(args, "num_batches", 150))           # :317
        n_pseudo = int(getattr(args, "n_pseudo", 50))
        idx_anomaly = list(self.dataset["idx_anomaly"])
        labels = self.dataset["labels"]
        timestamp = datetime.datetime.fromtimestamp(int(time.time())).strftime("%Y-%m-%d %H-%M-%S")
        dir_saver = args.save_dir + timestamp
        path_saver = os.path.join(dir_saver, "{}_{}.pkl".format(args.data_name, args.model))
        f1_mac_best, auc_best, ep_best = 0, 0, -1
        self.pcgnn_losses = []
        for epoch in range(args.num_epochs):
            random.shuffle(idx_train)                                  # :314
            loss_sum, con_sum, epoch_time = 0.0, 0.0, 0.0
            for batch in range(num_batches):
                t0 = time.time()
                i0, i1 = batch * args.batch_size, min((batch + 1) * args.batch_size, len(idx_train))
                batch_nodes = idx_train[i0:i1]
                random.shuffle(idx_anomaly)                            # :341
                batch_nodes = batch_nodes + idx_anomaly[:n_pseudo]     # :342,347
                batch_label = torch.as_tensor(labels[np.array(batch_nodes)], device=dev).long()
                optimizer.zero_grad()
                loss, loss_constraint = gnn_model.loss(batch_nodes, batch_label)           # :352-353
                loss.backward()
                optimizer.step()
                epoch_time += time.time() - t0
                self.pcgnn_losses.append((float(loss.item()), float(loss_constraint.item())))
                loss_sum += self.pcgnn_losses[-1][0]
                con_sum += self.pcgnn_losses[-1][1]
            print(f"Epoch: {epoch}, loss: {loss_sum / num_batches}, loss_constraint: {con_sum / num_batches}, time: {epoch_time}s")
            if epoch % args.valid_epochs == 0:
                print("Valid at epoch {}".format(epoch))
                f1_mac_val, f1_1_val, f1_0_val, auc_val, gmean_val = self._test_pcgnn(idx_valid, y_valid, gnn_model, args.batch_size,
                                                                                      args.thres)
                if auc_val > auc_best:
                    f1_mac_best, auc_best, ep_best = f1_mac_val, auc_val, epoch
                    if not os.path.exists(dir_saver):
                        os.makedirs(dir_saver)
                    print("  Saving model ...")
                    torch.save(gnn_model.state_dict(), path_saver)
        if ep_best >= 0:
            print("Restore model from epoch {}".format(ep_best))
            print("Model path: {}".format(path_saver))
            gnn_model.load_state_dict(torch.load(path_saver))
        return self._test_pcgnn(idx_test, y_test, gnn_model, args.batch_size, args.thres)

    @staticmethod
    def _test_pcgnn(test_cases, labels, model, batch_size, thres=0.5):
        """What the undefined `test_pcgnn` of `src/model_handler.py:392,411` has to be for that call to work: `test_sage`'s protocol
        (`src/utils.py:207-247`) on the class-1 GNN score of `PCALayer.to_prob(nodes, labels, train_flag=False)`."""
        from .metrics import binary_report
        test_cases = list(test_cases)
        labels = np.asarray(labels)
        probs = []
        with torch.no_grad():
            for it in range(int(len(test_cases) / batch_size) + 1):
                lo, hi = it * batch_size, min((it + 1) * batch_size, len(test_cases))
                if hi <= lo:
                    continue
                chunk = test_cases[lo:hi]
                dev = next(model.parameters()).device
                gnn_prob, _ = model.to_prob(chunk, torch.as_tensor(labels[lo:hi], device=dev).long(), train_flag=False)
                probs.append(gnn_prob[:, 1])
        probs = torch.cat(probs)
        r = binary_report(probs, torch.as_tensor(labels, device=probs.device), thres)
        print(f"   GNN F1-binary-1: {r['f1_1']:.4f}\tF1-binary-0: {r['f1_0']:.4f}" +
              f"\tF1-macro: {r['f1_macro']:.4f}\tG-Mean: {r['gmean']:.4f}\tAUC: {r['auc']:.4f}")
        print("Testing AP:", r["ap"])
        return r["f1_macro"], r["f1_1"], r["f1_0"], r["auc"], r["gmean"]

    @staticmethod
    def _test_graphsage(test_cases, labels, model, batch_size, thres=0.5):
        """`test_sage` (`src/utils.py:207-247`) for the two-class GraphSage head: `to_prob(nodes)` (repair 2: one argument), the
        score of a node = its class-1 probability."""
        from .metrics import binary_report
        test_cases = list(test_cases)
        probs = []
        with torch.no_grad():
            for it in range(int(len(test_cases) / batch_size) + 1):
                chunk = test_cases[it * batch_size:min((it + 1) * batch_size, len(test_cases))]
                if not chunk:
                    continue
                probs.append(model.to_prob(chunk)[:, 1])
        probs = torch.cat(probs)
        r = binary_report(probs, torch.as_tensor(np.asarray(labels), device=probs.device), thres)
        print(f"   GNN F1-binary-1: {r['f1_1']:.4f}\tF1-binary-0: {r['f1_0']:.4f}" +
              f"\tF1-macro: {r['f1_macro']:.4f}\tG-Mean: {r['gmean']:.4f}\tAUC: {r['auc']:.4f}")
        print("Testing AP:", r["ap"])
        return r["f1_macro"], r["f1_1"], r["f1_0"], r["auc"], r["gmean"]
