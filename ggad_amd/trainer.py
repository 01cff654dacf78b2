"""DGraph mini-batch training driver: batch schedule (reference-exact), chunked planning,
per-step kernel chain, optional data parallelism.

Mirrors the loop of `src/model_handler.py:310-370`:
  per epoch   random.shuffle(idx_train)                          (:314)
  per batch   batch = train[i*bs:(i+1)*bs]                       (:333-335)
              random.shuffle(idx_anomaly); += idx_anomaly[:50]   (:340-347)
              loss -> backward -> Adam                           (:356-364)
with ``num_batches = 150`` (:317).  The python ``random`` stream is continued bit-exactly by the
native sampler, so batch b of epoch e holds the same node ids as in the reference for equal seeds.

Data parallel (absent in the reference): the global batch stream is dealt round
robin, optimiser step s uses batches s*W .. s*W+W-1, rank r takes batch s*W+r; gradients of the
packed 5,248-float block are summed by ONE all-reduce per step and divided by W inside the Adam
kernel.  W = 1 is exactly the reference schedule.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graph import DeviceGraph
from .minibatch import BatchChunk, MiniBatchEngine, pack_features, reduce_gradients  # noqa: F401 (re-exported)
from .sampler import PyCompatRandom


class BatchSchedule:
    """Generates the reference's batch stream (host side)."""

    def __init__(self, idx_train: Sequence[int], idx_anomaly: Sequence[int], labels: np.ndarray, batch_size: int,
                 rng: PyCompatRandom, n_pseudo: int = 50, batches_per_epoch: int = 150):
        self.train = np.ascontiguousarray(np.asarray(idx_train, dtype=np.int64))
        self.pool = np.ascontiguousarray(np.asarray(idx_anomaly, dtype=np.int64))
        self.labels = np.asarray(labels)
        self.bs = int(batch_size)
        self.n_pseudo = int(n_pseudo)
        self.bpe = int(batches_per_epoch)
        self.rng = rng
        self._in_epoch = self.bpe      # forces a shuffle before the first batch
        self.global_batch = 0

    def next_batch(self) -> Tuple[np.ndarray, np.ndarray]:
        if self._in_epoch >= self.bpe:
            self.rng.shuffle(self.train)                                   # model_handler.py:314
            self._in_epoch = 0
        i0 = self._in_epoch * self.bs
        i1 = min((self._in_epoch + 1) * self.bs, len(self.train))
        self.rng.shuffle(self.pool)                                        # model_handler.py:341
        nodes = np.concatenate([self.train[i0:i1], self.pool[:self.n_pseudo]])
        self._in_epoch += 1
        self.global_batch += 1
        return nodes, self.labels[nodes].astype(np.int64)

    def next_batches(self, count: int, rank: int = 0, world: int = 1) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """`count` optimiser steps worth of batches for this rank (every rank advances the same stream)."""
        nodes, labs = [], []
        for s in range(count):
            for r in range(world):
                n, l = self.next_batch()
                if r == rank:
                    nodes.append(n)
                    labs.append(l)
        return nodes, labs


class DGraphTrainer:
    def __init__(self, graph: DeviceGraph, feat: torch.Tensor, embed_dim: int, schedule: BatchSchedule,
                 lr: float = 1e-3, weight_decay: float = 0.007, chunk_batches: int = 150, rank: int = 0,
                 world_size: int = 1, allreduce: Optional[Callable[[torch.Tensor], None]] = None,
                 engine: Optional[MiniBatchEngine] = None, packed: bool = True):
        """`feat` is the plain (N, F) table; with `packed` (default) the plans run on a private padded copy whose
        rows also hold the per-batch 2-hop counters (15 slots for F = 17), so a chunk is at most that many batches."""
        self.graph, self.feat = graph, feat
        self.schedule = schedule
        self.rank, self.world = int(rank), int(world_size)
        self.allreduce = allreduce if self.world > 1 else None
        self.engine = engine or MiniBatchEngine(feat.shape[1], embed_dim, feat.device, lr, weight_decay)
        self.chunk_batches = int(chunk_batches)
        f = int(feat.shape[1])
        self.packed = bool(packed) and f + 1 <= 64
        table = feat
        if self.packed:
            table = pack_features(feat)
            self.chunk_batches = max(1, min(self.chunk_batches, table.shape[1] - f))
        rows = self.chunk_batches * (schedule.bs + schedule.n_pseudo)
        mean_deg = max(1.0, graph.nnz / max(1, graph.n))
        self.chunk = BatchChunk(graph, table, embed_dim, self.chunk_batches, rows, int(rows * (mean_deg + 1) * 1.5) + 1024,
                                train=True, feat_dim=f)
        self.steps_done = 0

    def run_steps(self, n_steps: int, prepared: Optional[Tuple[List[np.ndarray], List[np.ndarray]]] = None,
                  gather_hook=None) -> int:
        """Run n optimiser steps; returns nodes processed by THIS rank."""
        done = 0
        nodes_seen = 0
        pos = 0
        while done < n_steps:
            k = min(self.chunk_batches, n_steps - done)
            if prepared is not None:
                bn, bl = prepared[0][pos:pos + k], prepared[1][pos:pos + k]
                pos += k
            else:
                bn, bl = self.schedule.next_batches(k, self.rank, self.world)
            self.chunk.build(bn, bl) if gather_hook is None else gather_hook(self.chunk, bn, bl)
            self.engine.train_chunk(self.chunk, self.allreduce, self.world, log_base=done)   # loss log slot = step index
            nodes_seen += sum(len(b) for b in bn)
            done += k
        self.steps_done += n_steps
        return nodes_seen

