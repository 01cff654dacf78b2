"""Device metrics (`ggad_amd/metrics.py`) against scikit-learn -- the reference's metric library -- on CPU tensors
(the functions are plain tensor programs; tests/test_dropin_gpu.py exercises them on the GPU through test_sage)."""
import numpy as np
import pytest
import torch
from sklearn.metrics import average_precision_score, confusion_matrix, f1_score, roc_auc_score

from ggad_amd.metrics import average_precision, binary_report, roc_auc


@pytest.mark.parametrize("n,rate,decimals", [(50, 0.3, None), (4000, 0.07, None), (4000, 0.07, 2), (30000, 0.004, 1), (1000, 0.5, 0)])
def test_matches_sklearn(n, rate, decimals):
    rng = np.random.default_rng(n + (decimals or 9))
    y = (rng.random(n) < rate).astype(np.int64)
    y[:2] = (0, 1)
    s = (rng.random(n) * 0.6 + 0.35 * y * rng.random(n)).astype(np.float32)
    if decimals is not None:
        s = np.round(s, decimals)                      # heavy ties: sklearn collapses equal scores into one threshold
    r = binary_report(torch.from_numpy(s), torch.from_numpy(y), 0.4)
    p = (s >= 0.4).astype(int)
    assert abs(r["auc"] - roc_auc_score(y, s)) < 1e-12
    assert abs(r["ap"] - average_precision_score(y, s, average="macro", pos_label=1)) < 1e-12
    assert abs(r["f1_1"] - f1_score(y, p, pos_label=1, average="binary", zero_division=0)) < 1e-12
    assert abs(r["f1_0"] - f1_score(y, p, pos_label=0, average="binary", zero_division=0)) < 1e-12
    assert abs(r["f1_macro"] - f1_score(y, p, average="macro", zero_division=0)) < 1e-12
    assert (r["tn"], r["fp"], r["fn"], r["tp"]) == tuple(confusion_matrix(y, p, labels=[0, 1]).ravel())


def test_constant_scores_and_single_class():
    y = torch.tensor([0, 1, 0, 1, 1])
    s = torch.full((5,), 0.25)
    assert roc_auc(s, y) == 0.5 and abs(average_precision(s, y) - 0.6) < 1e-15      # one threshold: chance level / prevalence
    with pytest.raises(ValueError):
        roc_auc(s, torch.zeros(5, dtype=torch.int64))
