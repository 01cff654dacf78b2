"""Mini-batch AnomalyDAE-style comparison model (`src/graphsage_anomalydae.py`): the DOMINANT-style model of
`ggad_amd.graphsage_dominant` with a sign-weighted squared error -- `torch.where(rec > 0, diff * 0.5, diff * (1 - 0.5))`
(`:157-160`), which is a uniform 0.5 with the reference's `pos_weight_a` but is kept as two weights here."""
from __future__ import annotations

from . import graphsage_dominant as _base
from .graphsage_dominant import Encoder, GCNAggregator, GCNEncoder, MeanAggregator  # noqa: F401  (same classes, `:13-120,174-281`)


class GCN(_base.GCN):
    pos_weight_a = 0.5
    recon_weights = (pos_weight_a, 1.0 - pos_weight_a)
