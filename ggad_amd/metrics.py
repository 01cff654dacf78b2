"""Binary classification metrics on the device: AUROC, average precision, F1 (class 1 / class 0 /
macro), confusion matrix and G-mean of `test_sage` (`src/utils.py:232-247`) and of `run.py:236-239`, computed from
scores that never leave HBM: one device sort + prefix sums in fp64 instead of a 1.7 M-element copy to the host and
sklearn.  Definitions follow scikit-learn's (the reference's metric library), including its handling of tied scores:

  roc_auc_score            trapezoids over the DISTINCT thresholds of the ROC curve
  average_precision_score  sum_n (R_n - R_{n-1}) P_n over the distinct thresholds, descending
  f1_score / confusion     from the thresholded predictions (score >= thres)

sklearn itself stays the checker in the tests (tests/test_metrics.py)."""
from __future__ import annotations

from typing import Dict

import torch


def _curve_counts(scores: torch.Tensor, labels: torch.Tensor):
    """tps / fps at the last element of every run of equal scores, scores descending (sklearn `_binary_clf_curve`)."""
    s, order = torch.sort(scores.double(), descending=True, stable=True)
    y = labels[order].double()
    n = s.numel()
    last = torch.ones(n, dtype=torch.bool, device=s.device)
    if n > 1:
        last[:-1] = s[:-1] != s[1:]
    tps = torch.cumsum(y, 0)[last]
    fps = torch.cumsum(1.0 - y, 0)[last]
    return tps, fps


def roc_auc(scores: torch.Tensor, labels: torch.Tensor) -> float:
    tps, fps = _curve_counts(scores, labels)
    if tps.numel() == 0 or tps[-1] == 0 or fps[-1] == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    zero = torch.zeros(1, dtype=torch.float64, device=tps.device)
    tpr = torch.cat([zero, tps / tps[-1]])
    fpr = torch.cat([zero, fps / fps[-1]])
    return float(torch.trapezoid(tpr, fpr))


def average_precision(scores: torch.Tensor, labels: torch.Tensor) -> float:
    tps, fps = _curve_counts(scores, labels)
    if tps.numel() == 0 or tps[-1] == 0:
        return 0.0
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=tps.device), recall[:-1]])
    return float(((recall - prev) * precision).sum())


def binary_report(scores: torch.Tensor, labels: torch.Tensor, thres: float) -> Dict[str, float]:
    """Everything `test_sage` prints and returns, one host read-back of a handful of scalars."""
    labels = labels.to(scores.device)
    pred = scores >= thres                                     # prob2pred, src/utils.py:250-260
    pos = labels == 1
    tp = int((pred & pos).sum())
    fp = int((pred & ~pos).sum())
    fn = int((~pred & pos).sum())
    tn = int((~pred & ~pos).sum())

    def f1(t, f_p, f_n):
        d = 2 * t + f_p + f_n
        return 2.0 * t / d if d else 0.0

    f1_1, f1_0 = f1(tp, fp, fn), f1(tn, fn, fp)
    gmean = float((tp * tn / ((tp + fn) * (tn + fp))) ** 0.5) if (tp + fn) and (tn + fp) else float("nan")
    return {"auc": roc_auc(scores, labels), "ap": average_precision(scores, labels), "f1_1": f1_1, "f1_0": f1_0,
            "f1_macro": 0.5 * (f1_1 + f1_0), "gmean": gmean, "tp": tp, "fp": fp, "fn": fn, "tn": tn}
