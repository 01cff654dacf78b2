"""`ModelHandler` of the mini-batch AnomalyDAE-style comparison model -- drop-in for `src/model_handler_anomalydae.py`:
the DOMINANT-style handler with 5 % relabelled normals (`:43`), 50 batches per epoch (`:139`) and the sign-weighted
reconstruction error of `ggad_amd.graphsage_anomalydae`."""
from __future__ import annotations

from . import graphsage_anomalydae as _model
from . import model_handler_dominate as _base


class ModelHandler(_base.ModelHandler):
    model_module = _model
    pseudo_frac = 0.05
    default_num_batches = 50
