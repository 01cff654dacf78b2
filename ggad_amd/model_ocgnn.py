"""Full-graph OCGNN comparison model -- drop-in for the reference's `model_ocgnn.py` on the GGAD full-graph kernels.

    Model(n_in, n_h, activation, negsamp_round, readout).forward(seq1, adj, sparse=False) -> h_2 (1, N, n_h)      :106-131

Two of GGAD's GCN layers (`ggad_amd.model.GCN`: exact-f32 MFMA projection + CSR SpMM with bias / PReLU epilogue) and the
same constructor order and parameter names as the reference (gcn1, gcn2, then the unused readout / discriminator, which
consume the RNG like there).  `ocgnn_loss` is the loss block of the training script (`ocgnn.py:83-118`).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ._lib import call, ptr
from .model import AvgReadout, Discriminator, GCN, MaxReadout, MinReadout, WSReadout, as_full_adj  # noqa: F401


class Model(nn.Module):
    def __init__(self, n_in, n_h, activation, negsamp_round, readout):
        super().__init__()
        self.read_mode = readout
        self.gcn1 = GCN(n_in, n_h, activation)
        self.gcn2 = GCN(n_h, n_h, activation)
        self.act = nn.ReLU()
        if readout == "max":
            self.read = MaxReadout()
        elif readout == "min":
            self.read = MinReadout()
        elif readout == "avg":
            self.read = AvgReadout()
        elif readout == "weighted_sum":
            self.read = WSReadout()
        self.disc = Discriminator(n_h, negsamp_round)

    def forward(self, seq1, adj, sparse=False):
        h_1 = self.gcn1(seq1, adj, sparse)
        return self.gcn2(h_1, adj, sparse)


class _OcgnnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, idx, center, r: float, beta: float):
        emb = emb.contiguous()
        n_idx = int(idx.numel()) if idx is not None else emb.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=emb.device)
        score = torch.empty(n_idx, dtype=torch.float32, device=emb.device)
        demb = torch.zeros_like(emb) if ctx.needs_input_grad[0] else None
        call("ggad_ocgnn_loss_f32", ptr(emb), ptr(idx) if idx is not None else 0, n_idx, emb.shape[1],
             ptr(center) if center is not None else 0, float(r), float(beta), ptr(loss), ptr(score),
             ptr(demb) if demb is not None else 0)
        ctx.demb = demb
        ctx.mark_non_differentiable(score)
        return loss[0], score

    @staticmethod
    def backward(ctx, g, _gs):
        return (ctx.demb * g if ctx.demb is not None else None), None, None, None, None


def ocgnn_loss(emb, idx=None, center=None, r: float = 0.0, beta: float = 0.5):
    """(loss, score) of `ocgnn.py:83-118` on the rows `idx` (int64 device tensor, duplicate-free; None = all rows) of the
    (N, H) embedding: score_i = ||emb_i - c||^2 - r^2, loss = r^2 + mean(relu(score)) / beta.  The reference rebuilds
    c = 0 and r = 0 inside every call (its warm-up update is dead code), which are the defaults here."""
    if emb.dim() != 2:
        raise ValueError("ocgnn_loss expects an (N, H) embedding")
    if idx is not None:
        idx = torch.as_tensor(idx, dtype=torch.int64, device=emb.device).contiguous()
    if center is not None:
        center = torch.as_tensor(center, dtype=torch.float32, device=emb.device).contiguous()
    return _OcgnnLoss.apply(emb, idx, center, r, beta)
