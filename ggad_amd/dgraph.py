"""DGraph-Fin data handling: loaders and the train/test split of the reference, host side.

`split_dgraphfin` re-states `src/model_handler.py:150-178`; it deliberately uses the same CPython
containers (``random.shuffle``, ``set`` / ``list(set)`` ordering, sklearn's ``train_test_split``) so
that, for equal seeds, the train list, the pseudo-anomaly pool and the test list come out in the
reference's order.  One-off work, outside every timed region.
"""
from __future__ import annotations

import pickle
import random
from typing import Dict

import numpy as np


def load_dgraphfin(npz_path: str, adj_path: str):
    """`src/utils.py:15-31`: x, y from the npz (labels = (y == 1)), pickled dict-of-sets adjacency."""
    f = np.load(npz_path)
    labels = (np.asarray(f["y"]).astype(np.float32) == 1).astype(np.int32)
    feat = np.asarray(f["x"]).astype(np.float32)
    with open(adj_path, "rb") as fh:
        homo = pickle.load(fh)
    return homo, feat, labels


def sparse_to_adjlist(sp_matrix, filename=None):
    """`src/utils.py:96-112` (the definition in effect: no self loops added): scipy sparse matrix -> `defaultdict(set)` with
    both directions of every stored entry, pickled to `filename` -- the `dgraphfin_adj_list` file `load_dgraphfin` reads.
    Built from the symmetrised CSR instead of a python loop over the entries (73 M of them on DGraph-Fin)."""
    from collections import defaultdict

    import scipy.sparse as sp
    m = sp.csr_matrix(sp_matrix)
    pat = sp.csr_matrix((np.ones(m.nnz, dtype=np.int8), m.indices, m.indptr), shape=m.shape)
    sym = (pat + pat.T).tocsr()
    sym.sort_indices()
    adj_lists = defaultdict(set)
    for node in np.flatnonzero(np.diff(sym.indptr)):
        adj_lists[int(node)] = set(sym.indices[sym.indptr[node]:sym.indptr[node + 1]].tolist())
    if filename is not None:
        with open(filename, "wb") as fh:
            pickle.dump(adj_lists, fh)
    return adj_lists


def normalize_features(mx: np.ndarray) -> np.ndarray:
    """`src/utils.py:74-84`: x / (rowsum + 0.01), inf -> 0; scipy promotes to fp64, caller casts to fp32."""
    mx = np.asarray(mx)
    rowsum = np.array(mx.sum(1)) + 0.01
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.0
    return r_inv.astype(np.float64)[:, None] * mx.astype(np.float64)


def split_dgraphfin(labels: np.ndarray, seed: int, test_ratio: float = 0.67, with_test: bool = True,
                    real_frac: float = 0.2, pseudo_frac: float = 0.05) -> Dict[str, object]:
    """The 'dgraphfin' branch of ModelHandler.__init__ (`src/model_handler.py:29-30,150-178`).  The handlers of the
    comparison models differ only in the two fractions: 0.15 / 0.10 (`src/model_handler_dominate.py:34,43`) and
    0.15 / 0.05 (`src/model_handler_anomalydae.py:34,43`).

    Mutates a COPY of ``labels`` (pseudo anomalies get label 1, :165) and returns the lists the
    training loop uses.  Consumes the global python/numpy RNGs exactly like the reference."""
    from sklearn.model_selection import train_test_split
    labels = np.array(labels).copy()
    np.random.seed(seed)
    random.seed(seed)
    index = list(range(len(labels)))
    idx_normal = [i for i in index if labels[i] == 0]
    idx_real_abnormal = [i for i in index if labels[i] == 1]
    idx_real_abnormal = idx_real_abnormal[0: int(len(idx_real_abnormal) * real_frac)]
    random.shuffle(idx_normal)
    idx_labeled = idx_normal[0: int(len(idx_normal) * 0.3)]
    idx_anomaly = idx_labeled[0: int(len(idx_labeled) * pseudo_frac)]
    labels[idx_anomaly] = 1
    idx_train = list(set(idx_labeled).difference(set(idx_anomaly)))
    idx_train = idx_train + idx_real_abnormal                       # "contamination"
    y_train = labels[idx_train]
    idx_rest = list(set(index).difference(set(idx_labeled)))
    idx_rest = list(set(idx_rest).difference(set(idx_real_abnormal)))
    y_rest = labels[idx_rest]
    if not with_test:      # bench / tests that only need the training lists
        return dict(labels=labels, idx_train=idx_train, y_train=y_train, idx_labeled=idx_labeled,
                    idx_anomaly=idx_anomaly, idx_rest=idx_rest)
    idx_valid, idx_test, y_valid, y_test = train_test_split(idx_rest, y_rest, stratify=y_rest, test_size=test_ratio,
                                                            random_state=2, shuffle=True)
    return dict(labels=labels, idx_train=idx_train, y_train=y_train, idx_labeled=idx_labeled, idx_anomaly=idx_anomaly,
                idx_valid=idx_valid, y_valid=y_valid, idx_test=idx_test, y_test=y_test)
