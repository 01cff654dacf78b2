#!/usr/bin/env python3
"""Full-graph TAM comparison run (truncated affinity maximisation) on one MI355X:  python tam.py --dataset photo [--synthetic]

Same command line and flow as the reference's `tam.py`: `--cutting` truncation rounds x `--N_tree` models of 500 epochs each
(lr 1e-5, fixed there at `:35-36`), the attribute distance of every edge computed once (cached under `distance_save/` like the
reference), per round `graph_nsgt` -> `normalize_adj_tensor` -> epochs of forward / `max_message` loss / `inference` / backward /
Adam with `zero_grad()` once per round, then the AUROC / AP prints of `:204-232`.  The reference pins the seeds only through
PYTHONHASHSEED (its `random` / `numpy` / `torch` seeding is commented out, `:42-47`); here `--seed` seeds them so that a run can
be repeated.  Adjacency, distances and truncated graphs are CSR; the GCN layers, both affinity passes, the backward and Adam
run in the kernels of libggad_hip.so; after two eager epochs every epoch of a round is one replayed hipGraph.
`--synthetic` / `--device` / `--quiet` / `--no_graph` / `--num_epoch` / `--lr` are additions.
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ggad_amd import synth  # noqa: E402
from ggad_amd import tam_utils as T  # noqa: E402
from ggad_amd.fullgraph import FlatAdam, FullGraphAdj  # noqa: E402
from ggad_amd.fullgraph_bench import SIZES  # noqa: E402
from ggad_amd.metrics import average_precision, roc_auc  # noqa: E402
from ggad_amd.model_tam import Model  # noqa: E402
from ggad_amd.utils import preprocess_features  # noqa: E402


def parse():
    p = argparse.ArgumentParser(description="Truncated Affinity Maximization for Graph Anomaly Detection")
    p.add_argument("--dataset", type=str, default="photo")
    p.add_argument("--lr", type=float)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--embedding_dim", type=int, default=128)
    p.add_argument("--num_epoch", type=int)
    p.add_argument("--drop_prob", type=float, default=0.0)
    p.add_argument("--subgraph_size", type=int, default=15)
    p.add_argument("--readout", type=str, default="avg")
    p.add_argument("--margin", type=int, default=2)
    p.add_argument("--negsamp_ratio", type=int, default=2)
    p.add_argument("--cutting", type=int, default=8)
    p.add_argument("--N_tree", type=int, default=1)
    p.add_argument("--lamda", type=int, default=0)
    p.add_argument("--dataset_model", type=str, default="photo")
    p.add_argument("--synthetic", action="store_true", help="generate a graph of the dataset's size instead of loading ./data/*.mat")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--quiet", action="store_true")
    p.add_argument("--no_graph", action="store_true", help="do not replay a captured hipGraph of the training epoch")
    a = p.parse_args()
    if a.lr is None:
        a.lr = 1e-5                                                    # tam.py:35
    if a.num_epoch is None:
        a.num_epoch = 500                                              # tam.py:36
    return a


def load(args):
    if args.synthetic or not os.path.exists("./data/{}.mat".format(args.dataset)):
        if not args.synthetic:
            print("./data/{}.mat not found: using a synthetic graph of the same size".format(args.dataset))
        n, ne, f, rate = SIZES[args.dataset]
        rowptr, col = synth.make_graph(n, ne, args.seed, kind="powerlaw", max_degree=max(64, n // 8), exact=True)
        adj = synth.csr_to_scipy(rowptr, col, n)
        feat = sp.lil_matrix(synth.make_features(n, f, args.seed))
        ano = synth.make_labels(n, rate, args.seed)
        normal, idx_test = T.split_nodes(ano)
        return adj, feat, ano, normal, idx_test
    adj, feat, ano, _, _, normal, idx_test = T.load_mat(args.dataset)
    return adj, feat, ano, normal, idx_test


def main():
    args = parse()
    print("Dataset: ", args.dataset)
    os.environ["PYTHONHASHSEED"] = str(args.seed)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if not torch.cuda.is_available():
        sys.exit("tam.py needs an MI355X: there is no CPU fallback")
    dev = torch.device("cuda", args.device)
    torch.cuda.set_device(dev)
    adj, features, ano_label, normal_label_idx, idx_test = load(args)
    if args.dataset in ["Amazon", "YelpChi", "Amazon-all", "YelpChi-all", "elliptic_no_isolate"]:       # tam.py:55-57
        features = np.asarray(preprocess_features(features))
    else:
        features = np.asarray(features.todense())
    nb_nodes, ft_size = features.shape
    print(adj.sum())
    raw = (adj + sp.eye(nb_nodes)).tocsr()                             # raw_adj = adj + I                  tam.py:68-70
    raw.sort_indices()
    feats = torch.FloatTensor(np.asarray(features, dtype=np.float32)[np.newaxis]).to(dev)
    models, optimisers = [], []
    for _ in range(args.cutting * args.N_tree):                        # tam.py:78-87
        model = Model(ft_size, args.embedding_dim, "prelu", args.negsamp_ratio, args.readout).to(dev)
        models.append(model)
        optimisers.append(FlatAdam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay))
    y_all = torch.as_tensor(np.asarray(ano_label).astype(np.int64), device=dev)
    idx_test_dev = torch.as_tensor(np.asarray(idx_test, dtype=np.int64), device=dev)

    start = time.time()
    print("<<<<<<Start to calculate distance<<<<<")
    # The reference caches a DENSE N x N distance array as distance_save/dis_array_{dataset_model}.npy (tam.py:164-170); here the cache
    # holds one value per stored entry of A + I and is keyed by the DATASET and its entry count (a file of its own name: a second
    # dataset, or a cache the reference wrote, is never mistaken for it).  A dense reference cache of the right shape is converted.
    dis_path = "distance_save/dis_edges_{}_{}.npy".format(args.dataset, raw.nnz)
    ref_path = "distance_save/dis_array_{}.npy".format(args.dataset_model)
    dis_vals = None
    if not args.synthetic and os.path.exists(dis_path):
        cand = np.load(dis_path)
        if cand.shape == (raw.nnz,):
            dis_vals = cand
    if dis_vals is None and not args.synthetic and os.path.exists(ref_path):
        cand = np.load(ref_path, mmap_mode="r")
        if cand.shape == tuple(raw.shape):                             # the reference's dense cache: gather at the stored entries
            coo = raw.tocoo()
            dis_vals = np.asarray(cand[coo.row, coo.col], dtype=np.float32)
    if dis_vals is None:
        dis_vals = T.calc_distance(raw, feats[0])                      # one value per entry of A + I      tam.py:168
        if not args.synthetic:
            os.makedirs("distance_save", exist_ok=True)
            np.save(dis_path, dis_vals)
    all_cut = [raw.copy() for _ in range(args.N_tree)]                 # tam.py:159-161
    index = 0
    message_mean_list = []
    epoch_times = []
    for n_cut in range(args.cutting):
        print("n_cut.{}".format(n_cut))
        message_list = []
        for n_t in range(args.N_tree):
            cut = T.graph_nsgt(raw, dis_vals, all_cut[n_t])            # tam.py:180
            optimisers[index].zero_grad()                              # once per round                    tam.py:181
            print("<<<< cutting num .{}<<<<<<".format(n_cut))
            full = FullGraphAdj(T.normalize_adj_tensor(cut), raw, dev)
            torch.cuda.synchronize()
            t0 = time.time()
            losses, message_sum = T.train_cut(models[index], optimisers[index], feats, full, normal_label_idx, args.num_epoch,
                                              use_graph=not args.no_graph, log_every=0 if args.quiet else 50)
            torch.cuda.synchronize()
            epoch_times.append((time.time() - t0) / max(1, args.num_epoch))
            message_list.append(message_sum.detach().unsqueeze(0))
            all_cut[n_t] = cut
            index += 1
        for mes in message_list:                                       # tam.py:203-207
            m = mes[0]
            score = 1 - (m - m.min()) / (m.max() - m.min())
            print("{} AUC:{:.4f}".format(args.dataset, roc_auc(score, y_all)))
        message_mean = torch.mean(torch.cat(message_list), 0)
        message_mean_list.append(message_mean.unsqueeze(0))
        score = 1 - (message_mean - message_mean.min()) / (message_mean.max() - message_mean.min())
        auc = roc_auc(score[idx_test_dev], y_all[idx_test_dev])        # tam.py:221-225
        print("AP:", average_precision(score[idx_test_dev], y_all[idx_test_dev]))
        print("{} AUC:{:.4f}".format(args.dataset, auc))
        mm = torch.mean(torch.cat(message_mean_list), 0)               # tam.py:226-232
        score = 1 - (mm - mm.min()) / (mm.max() - mm.min())
        print("AP:", average_precision(score, y_all))
        print("{} AUC:{:.4f}".format(args.dataset, roc_auc(score, y_all)))
    end = time.time()
    print(end - start)
    print("epochs/s (training windows of all rounds): {:.1f}; nodes/s: {:.1f}".format(
        1.0 / float(np.mean(epoch_times)), nb_nodes / float(np.mean(epoch_times))))


if __name__ == "__main__":
    main()
