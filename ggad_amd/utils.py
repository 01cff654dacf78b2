"""Data handling of the full-graph program (reference `utils.py`), host side, one-off.

`load_mat` keeps the reference's split logic and its use of python's `random` (`utils.py:66-141`) so that the
same seed yields the same index lists; `normalize_adj` / `preprocess_features` are computed in fp64 with scipy
exactly like the reference (`utils.py:37-54`) because the CSR values must come out identical before the fp32 cast.
"""
from __future__ import annotations

import random
from collections import Counter

import numpy as np
import scipy.io as sio
import scipy.sparse as sp


def preprocess_features(features):
    """Row-normalise (x / rowsum, inf -> 0); returns a dense matrix like the reference's first return value."""
    features = sp.csr_matrix(features, dtype=np.float64) if not sp.issparse(features) else features.astype(np.float64)
    rowsum = np.asarray(features.sum(1)).reshape(-1)
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    return np.asarray(sp.diags(r_inv).dot(features).todense())


def normalize_adj(adj):
    """D^-1/2 A D^-1/2 (inf -> 0) as COO, the order of operations of the reference (`utils.py:47-54`)."""
    adj = sp.coo_matrix(adj, dtype=np.float64)
    rowsum = np.asarray(adj.sum(1)).reshape(-1)
    with np.errstate(divide="ignore"):
        d = np.power(rowsum, -0.5)
    d[np.isinf(d)] = 0.0
    dm = sp.diags(d)
    return adj.dot(dm).transpose().dot(dm).tocoo()


def split_nodes(ano_labels, dataset: str, train_rate=0.3, val_rate=0.1, verbose=True):
    """Split + labelled-normal sampling + outlier seeds of `load_mat` (`utils.py:89-140`), python `random` driven."""
    num_node = len(ano_labels)
    num_train, num_val = int(num_node * train_rate), int(num_node * val_rate)
    all_idx = list(range(num_node))
    random.shuffle(all_idx)
    idx_train = all_idx[:num_train]
    idx_val = all_idx[num_train:num_train + num_val]
    idx_test = all_idx[num_train + num_val:]
    if verbose:
        print("Training", Counter(np.squeeze(ano_labels[idx_train])))
        print("Test", Counter(np.squeeze(ano_labels[idx_test])))
    all_normal = [i for i in idx_train if ano_labels[i] == 0]
    rate = 0.5
    normal_label_idx = all_normal[: int(len(all_normal) * rate)]
    if verbose:
        print("Training rate", rate)
    random.shuffle(normal_label_idx)
    frac = 0.05 if dataset in ["Amazon"] else 0.15
    abnormal_label_idx = normal_label_idx[: int(len(normal_label_idx) * frac)]
    return all_idx, idx_train, idx_val, idx_test, normal_label_idx, abnormal_label_idx


def load_mat(dataset, train_rate=0.3, val_rate=0.1, path=None):
    """Same 12-tuple as the reference's `load_mat` (`utils.py:66-141`)."""
    data = sio.loadmat(path or "./dataset/{}.mat".format(dataset))
    label = data["Label"] if ("Label" in data) else data["gnd"]
    attr = data["Attributes"] if ("Attributes" in data) else data["X"]
    network = data["Network"] if ("Network" in data) else data["A"]
    adj = sp.csr_matrix(network)
    feat = sp.lil_matrix(attr)
    ano_labels = np.squeeze(np.array(label))
    str_ano = np.squeeze(np.array(data["str_anomaly_label"])) if "str_anomaly_label" in data else None
    attr_ano = np.squeeze(np.array(data["attr_anomaly_label"])) if "attr_anomaly_label" in data else None
    all_idx, idx_train, idx_val, idx_test, normal_idx, abnormal_idx = split_nodes(ano_labels, dataset, train_rate, val_rate)
    return (adj, feat, ano_labels, all_idx, idx_train, idx_val, idx_test, ano_labels, str_ano, attr_ano, normal_idx,
            abnormal_idx)
