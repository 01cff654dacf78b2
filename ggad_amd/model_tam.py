"""Drop-in replacement for the reference's `model_tam.py` (TAM comparison model), backed by libggad_hip.so.

    GCN(in_ft, out_ft, act, bias=True).forward(seq, adj, sparse=False)                       reference :15-45
    Model(n_in, n_h, activation, negsamp_round, readout).forward(seq, adj, sparse=False)
        -> (feat, feat1, feat2)                                                              reference :132-157

Same constructor order (the same seed gives the same initial weights), parameter names and output shapes.  The two GCN
layers are the full-graph layer of the GGAD path (`GcnLayerFn`: exact-f32 MFMA projection + CSR SpMM with bias / PReLU
epilogue, input layer on the cached aggregate), `fc1` / `fc2` run on the MFMA GEMM (`LinearFn`).  `adj` is a
`ggad_amd.fullgraph.FullGraphAdj` (normalised truncated adjacency + raw adjacency) or the reference's dense (1, N, N) tensor.
The GraphSAGE / GIN helper classes of the reference file are not used by `tam.py` and are not provided.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .fullgraph import LinearFn
from .model import GCN, AvgReadout, MaxReadout, MinReadout, WSReadout  # the TAM file repeats these classes verbatim (`:15-84`)

__all__ = ["GCN", "Model", "AvgReadout", "MaxReadout", "MinReadout", "WSReadout"]


class Model(nn.Module):
    def __init__(self, n_in, n_h, activation, negsamp_round, readout):
        super().__init__()
        self.read_mode = readout
        self.gcn1 = GCN(n_in, 2 * n_h, activation)
        self.gcn2 = GCN(2 * n_h, n_h, activation)
        self.act = nn.PReLU()
        self.fc1 = nn.Linear(n_h, 2 * n_h, bias=False)
        self.fc2 = nn.Linear(n_h, 2 * n_h, bias=False)
        self.ReLU = nn.ReLU()
        if readout == "max":
            self.read = MaxReadout()
        elif readout == "min":
            self.read = MinReadout()
        elif readout == "avg":
            self.read = AvgReadout()
        elif readout == "weighted_sum":
            self.read = WSReadout()

    def forward(self, seq, adj, sparse=False):
        feat = self.gcn1(seq, adj)
        feat = self.gcn2(feat, adj)
        x = feat.reshape(-1, feat.shape[-1])
        feat1 = LinearFn.apply(x, self.fc1.weight, False)
        feat2 = LinearFn.apply(x, self.fc2.weight, False)
        if feat.dim() == 3:
            feat1, feat2 = feat1.unsqueeze(0), feat2.unsqueeze(0)
        return feat, feat1, feat2
