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

import os
import sys
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .graph import DeviceGraph
from .minibatch import BatchChunk, MiniBatchEngine, RowList, reduce_gradients  # noqa: F401 (re-exported)
from .sampler import PyCompatRandom


def _other_llc_cpus():
    """CPUs (allowed to this process) of an L3 domain OTHER than the one the calling thread runs on -- the home of the sampler
    pipeline -- or None (one L3 only, no sysfs, GGAD_SAMPLER_LLC=same).  Deterministic: the next domain in CPU order."""
    import ctypes
    if os.environ.get("GGAD_SAMPLER_LLC", "other") == "same":
        return None
    try:
        here = ctypes.CDLL(None).sched_getcpu()
        allowed = os.sched_getaffinity(0)
        groups = {}
        for cpu in sorted(allowed):
            with open(f"/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list") as fh:
                groups.setdefault(fh.read().strip(), set()).add(cpu)
        doms = sorted(groups.values(), key=min)
        if len(doms) < 2:
            return None
        mine = next((i for i, d in enumerate(doms) if here in d), 0)
        for step in range(1, len(doms)):
            d = doms[(mine + step) % len(doms)]
            if len(d) >= 6:                  # (the sampler wants six CPUs for its stages)
                return d
        return None
    except (OSError, AttributeError, ValueError):
        return None


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
        """`count` optimiser steps worth of batches for this rank (every rank advances the same stream).  One native call:
        a pipeline of threads -- generator, acceptance walk, permutation builders, composition -- confined to one L3 (`ggad_sched_batches`, csrc/sampler.cpp)."""
        import ctypes
        total = int(count) * int(world)
        if total == 0:
            return [], []
        lib = self.rng._lib
        stride = self.bs + self.n_pseudo
        out = np.empty((total, stride), dtype=np.int64)
        lens = np.empty(total, dtype=np.int32)
        ie = ctypes.c_int32(self._in_epoch)
        from . import _lib
        _lib.check(lib.ggad_sched_batches(self.rng._h, self.train.ctypes.data, len(self.train), self.pool.ctypes.data, len(self.pool),
                                          self.bs, self.n_pseudo, self.bpe, ctypes.byref(ie), total, out.ctypes.data,
                                          lens.ctypes.data), "ggad_sched_batches")
        self._in_epoch = int(ie.value)
        self.global_batch += total
        mine = out[rank::world][:count]                      # this rank's batches (rows of the fresh array: views, no copies)
        mlen = lens[rank::world][:count]
        if int(mlen.min()) == stride:
            lab2d = self.labels[mine].astype(np.int64)      # one vectorised lookup for the whole call
            if world > 1:
                mine = np.ascontiguousarray(mine)           # (this rank's rows as their own matrix)
            return RowList(mine), RowList(lab2d)
        nodes = [mine[s, :mlen[s]] for s in range(count)]
        labs = [self.labels[n].astype(np.int64) for n in nodes]
        return nodes, labs


class DGraphTrainer:
    def __init__(self, graph: DeviceGraph, feat: torch.Tensor, embed_dim: int, schedule: BatchSchedule,
                 lr: float = 1e-3, weight_decay: float = 0.007, chunk_batches: int = 150, rank: int = 0,
                 world_size: int = 1, allreduce: Optional[Callable[[torch.Tensor], None]] = None,
                 engine: Optional[MiniBatchEngine] = None, hop2: str = "ldsw",
                 overlap: bool = True, prefetch: bool = True, chain: int = 0, dense_cus: Optional[int] = None,
                 ramp: Optional[Sequence[int]] = None, exchange=None, own_stream: bool = False,
                 resident: Optional[bool] = None):
        """`feat` is the plain (N, F) table.  hop2 = "ldsw" (default): 2-hop counts in LDS per (tile, batch), per-pair
        counts streamed to the gather, feature rows padded to one 128-byte line; "global": per-batch counter slots in HBM +
        device atomics (the fallback the LDS path takes by itself when a chunk exceeds its limits).
        `overlap` (default): two chunk buffers; the plan + gather of chunk c+1 runs on one stream while the dense steps
        of chunk c run on another, the two streams confined to DISJOINT compute units (`dense_cus` CUs for the dense
        chain, the rest for the plan; `ggad_stream_create_cu_mask`).  `dense_cus` None = 64 (round-2 sweep at DGraph size:
        24 -> 56.0, 32 -> 51.5, 48 -> 49.7, 64 -> 46.3, 80 -> 50.3, 96 -> 51.2, 128 -> 59.7 us per step).  With plain streams
        (`dense_cus=0`) there is no gain: the tiny dependent launches of the dense chain queue behind the chip-filling
        gather launches.
        `ramp`: sizes (batches) of the FIRST chunks of a `run_steps` call.  The plan of the first chunk cannot overlap
        anything, so a run starts with small chunks (the dense chain starts after a fraction of a millisecond instead of
        after a 150-batch plan) and grows to `chunk_batches`; None = `default_ramp`.
        `own_stream`: data parallel with one batch stream PER RANK instead of one stream dealt to the ranks.  The bit-exact sampler
        is serial (one Mersenne-Twister stream: ~37 us per batch on the host); dealing one stream to W ranks makes every rank
        generate W batches per step, which bounds an end-to-end run at about two GPUs' worth of steps.  With W > 1 the
        trajectory is not the reference's anyway (global batch W x 150), so ranks may draw from streams of their own.
        `exchange`: a connected `OneShotExchange` (ggad_amd/exchange.py) -- the data-parallel gradient exchange then happens
        inside the Adam launch over peer-mapped buffers, with no host call per step; else `allreduce` (RCCL) is used.
        `prefetch`: the host sampler (bit-exact CPython shuffle, ~0.1 ms per batch for the 55k pool) runs in a
        background thread one chunk ahead; the C call releases the GIL, so sampling overlaps the GPU work."""
        self.graph, self.feat = graph, feat
        self.schedule = schedule
        self.rank, self.world = int(rank), int(world_size)
        # which batches of `schedule` are this rank's: every W-th one of the stream all ranks generate (default: the reference's
        # stream dealt to the ranks), or all of them (`own_stream`: the schedule is this rank's alone, e.g. seeded per rank)
        self.sched_rank, self.sched_world = (0, 1) if own_stream else (self.rank, self.world)
        self.allreduce = allreduce if self.world > 1 else None
        self.exchange = exchange if (self.world > 1 and exchange is not None and exchange.ok) else None
        # `resident`: the dense steps of a chunk as one launch resident on one XCD (None = whenever supported).  Ranks that SHARE a
        # device (tests) must pass False: two resident kernels cannot both hold the same XCD
        self.engine = engine or MiniBatchEngine(feat.shape[1], embed_dim, feat.device, lr, weight_decay, chain=chain,
                                                resident=(None if resident is None or resident else False))
        self.chunk_batches = int(chunk_batches)
        self.ramp = None if ramp is None else [int(k) for k in ramp]
        f = int(feat.shape[1])
        table = feat
        if hop2 == "ldsw" and f <= 32 and (f * 4) % 128 != 0:
            # one random access per neighbour: keep every feature row inside one 128-byte line
            table = torch.zeros(feat.shape[0], 32, dtype=torch.float32, device=feat.device)
            table[:, :f] = feat
        rows = self.chunk_batches * (schedule.bs + schedule.n_pseudo)
        mean_deg = max(1.0, graph.nnz / max(1, graph.n))
        ent_cap = int(rows * (mean_deg + 1) * 1.5) + 1024
        self.chunk = BatchChunk(graph, table, embed_dim, self.chunk_batches, rows, ent_cap, train=True, feat_dim=f, hop2=hop2)
        self.overlap = bool(overlap) and feat.device.type == "cuda"
        self.dense_cus = 0
        self.resident_split = False
        if dense_cus is not None and int(dense_cus) not in (0, 24, 28):
            self.engine.resident = False                 # an explicit split of the chip: the launch chain of round 2
        self.chunks = [self.chunk]
        if self.overlap:
            self.chunks.append(BatchChunk(graph, table, embed_dim, self.chunk_batches, rows, ent_cap, train=True, feat_dim=f,
                                          hop2=hop2))
            # XCD-resident dense steps (MiniBatchEngine.resident): the chunk kernel owns `xcd_wgs` compute units of XCD 0, the plan
            # kernels of the next chunk run on the other seven XCDs (they skip the workgroups the dispatcher would put on XCD 0).
            # dense_cus 24 / 28 (default 28): CU-masked streams; 0: plain streams (the skipping launches alone keep the plan off
            # XCD 0); any other explicit value: the cross-XCD stream pair of round 2 with the 5-launch chain
            import os
            self.resident_split = bool(self.engine.resident) and (self.world == 1 or self.exchange is not None) \
                and os.environ.get("GGAD_XCD_SPLIT", "1") != "0"
            if dense_cus is not None and int(dense_cus) not in (0, 24, 28):
                self.resident_split = False
            if dense_cus is None:
                dense_cus = 28 if self.resident_split else 64
            self.dense_cus = int(dense_cus)
            self.side, self.hi = self._make_streams(feat.device, int(dense_cus))
            if self.resident_split and not getattr(self, "_xcd_ready", False):
                self.resident_split = False
            if not self.resident_split:
                self.engine.resident = False             # (a masked stream without a whole XCD cannot hold the chunk kernel)
        if self.engine.resident:
            # the same workgroup count with and without overlap: results are bit-identical across the two paths
            self.engine.xcd_wgs = self.dense_cus if (self.overlap and self.dense_cus in (24, 28)) else 28
        self.steps_done = 0
        self.prefetch = bool(prefetch)
        self._stream = None
        self._pending = ([], [])        # batches taken from the sampler stream but not consumed yet
        # recovery from a time-out of the XCD-resident chunk kernel (single process): optimiser state at the last check point and
        # the batches of every run since -- `check_exchange` replays them on the launch chain when the kernel reported an error
        self._snap = None
        self._window = None             # list of (n_steps, nodes, labels) since the snapshot, or None: nothing recorded
        self.resident_fallbacks = 0

    def plan_stream_first_xcd(self) -> int:
        """XCD on which block 0 of a launch on the plan stream runs right now (diagnostics: the index-skipping plan launches were
        calibrated with this value when the streams were made)."""
        import ctypes
        from . import _lib
        first = ctypes.c_int32(-1)
        _lib.check(_lib.load().ggad_xcd_first_of_stream(ctypes.byref(first), self._raw_streams[0]), "ggad_xcd_first_of_stream")
        return int(first.value)

    def check_exchange(self, dist=None) -> None:
        """Raise if a one-shot exchange step timed out waiting for a peer (the kernel's bounded wait sets an error word instead of
        hanging), or if the XCD-resident chunk kernel did.  Reads device memory: call after a synchronisation point (end of a run
        / before a validation sweep).  With `dist` (torch.distributed, initialised) the error words are all-reduced (MAX) first, so
        EVERY rank raises together instead of one rank raising while its peers walk into the next collective."""
        err = 0
        if self.exchange is not None:
            err = int(self.exchange.error() != 0)
        if self.engine.xcd_ws is not None and self.engine.xcd_status()["error"]:
            if self.world == 1 and self._window:
                self._replay_on_chain()                      # recoverable here: state of the last check point + the batches since
            else:
                err |= 2
        elif self._window is not None:
            self._window = None                              # a clean check point: the next run takes a fresh snapshot
        if dist is not None and self.world > 1 and dist.is_available() and dist.is_initialized():
            t = torch.tensor([err], dtype=torch.int32, device=self.feat.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            err = int(t.item())
        if err & 1:
            raise RuntimeError("ggad_amd: one-shot gradient exchange timed out waiting for a peer rank (on this or another rank; "
                               "weights are no longer in step across the ranks)")
        if err & 2:
            raise RuntimeError("ggad_amd: the XCD-resident chunk kernel timed out (placement or barrier; see MiniBatchEngine.xcd_status)")

    def default_ramp(self, n_steps: int) -> List[int]:
        """Chunk sizes of a run of n optimiser steps.  With overlap the first plan is exposed and the last chunk's dense steps
        are: start small, grow geometrically to `chunk_batches`, and keep the tail chunk short when the run itself is short."""
        import os
        env = os.environ.get("GGAD_RAMP")
        if self.ramp is not None or env:
            head = list(self.ramp) if self.ramp is not None else [int(x) for x in env.split(",") if x]
        elif not self.overlap:
            head = []
        elif n_steps <= 32:
            # measured (MI355X, DGraph size, 20 steps): one chunk on the whole chip 2.04 M nodes/s; 2 or 3 overlapped chunks
            # 1.6-1.8 M -- below ~16 batches a plan costs more per batch than the dense steps it could hide (little reuse of
            # neighbour rows across so few batches), and the CU split slows both
            head = [n_steps]
        else:
            head = [16, 32, 64]
        sizes, left = [], int(n_steps)
        for k in head:
            if left <= 0 or k >= self.chunk_batches:
                break
            sizes.append(min(k, left))
            left -= sizes[-1]
        while left > 0:
            sizes.append(min(self.chunk_batches, left))
            left -= sizes[-1]
        return sizes

    def _make_streams(self, device, dense_cus: int):
        """(plan stream, dense stream).  dense_cus > 0: CU-masked HIP streams (dense chain on CUs [0, dense_cus), plan on
        the rest); dense_cus == 0: plain torch streams, dense chain on the high-priority queue."""
        import ctypes
        if dense_cus <= 0:
            side, hi = torch.cuda.Stream(device=device, priority=0), torch.cuda.Stream(device=device, priority=-1)
            if getattr(self, "resident_split", False):
                from . import _lib
                first = ctypes.c_int32(-1)
                with torch.cuda.device(device):
                    _lib.check(_lib.load().ggad_xcd_first_of_stream(ctypes.byref(first), side.cuda_stream), "ggad_xcd_first_of_stream")
                self._raw_streams = [side.cuda_stream, hi.cuda_stream]
                self._xcd_skip = (0 - int(first.value)) % 8
                self._xcd_ready = True
            return side, hi
        import warnings
        from . import _lib
        lib = _lib.load()
        try:
            n = ctypes.c_int32(0)
            idx = device.index if device.index is not None else torch.cuda.current_device()
            _lib.check(lib.ggad_device_cu_count(idx, ctypes.byref(n)), "ggad_device_cu_count")
            n_cu = int(n.value)
            if not 0 < dense_cus < n_cu:
                raise ValueError(f"dense_cus must be in (0, {n_cu})")
            words = (n_cu + 31) // 32
            raw, out = [], []
            with torch.cuda.device(device):
                # mask bit i = CU i, numbered round-robin over the 8 XCDs (measured: bits 0..31 = 4 CUs on every XCD -> 60 us/step;
                # every 8th bit = one whole XCD for the dense chain -> 79 us/step), so a contiguous range spreads the dense
                # chain over all XCDs
                dense = set(range(dense_cus))
                plan = set(range(n_cu)) - dense
                if getattr(self, "resident_split", False):
                    # mask bit i = CU i // 8 of XCD i % 8.  Chunk kernel: CUs 0 .. dense_cus-1 of XCD 0 (+ the last CU of every
                    # other XCD, where its workgroups only pass through); plan: the rest
                    # (multiples of 4 only: the XCD deals workgroups to its four shader engines in rotation, and a workgroup waits
                    #  for a compute unit of ITS engine -- with 26 or 30 enabled CUs one engine gets more workgroups than it has CUs
                    #  and the launch never becomes resident: measured, the registration wait times out)
                    if dense_cus not in (24, 28):
                        raise ValueError("with the XCD-resident chunk kernel dense_cus is 24 or 28 compute units of one XCD")
                    per = n_cu // 8
                    dense = {8 * cu for cu in range(dense_cus)} | {8 * (per - 1) + x for x in range(1, 8)}
                    plan = set(range(n_cu)) - dense
                for members in (plan, dense):
                    mask = (ctypes.c_uint32 * words)()
                    for cu in members:
                        mask[cu // 32] |= 1 << (cu % 32)
                    h = ctypes.c_void_p()
                    _lib.check(lib.ggad_stream_create_cu_mask(ctypes.cast(mask, ctypes.c_void_p), words, ctypes.byref(h)),
                               "ggad_stream_create_cu_mask")
                    raw.append(h.value)
                    out.append(torch.cuda.ExternalStream(h.value, device=device))
            self._raw_streams = raw
            self._own_streams = list(raw)              # created here (hipExtStreamCreateWithCUMask): destroyed by `close`
            import atexit
            import weakref
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref().close())     # before the HIP runtime goes away (a profiler's exit
            #                                                                  hook crashed on streams that were still alive)
            if getattr(self, "resident_split", False):
                first = ctypes.c_int32(-1)
                _lib.check(lib.ggad_xcd_first_of_stream(ctypes.byref(first), raw[0]), "ggad_xcd_first_of_stream")
                skip = (0 - int(first.value)) % 8               # blocks b with (first + b) % 8 == 0 would run on XCD 0
                self._xcd_skip = skip
                self._xcd_ready = True
            return out[0], out[1]
        except ValueError:
            raise
        except Exception as exc:      # no CU masking on this stack: plain streams still give a correct (slower) overlap
            warnings.warn(f"CU-masked streams unavailable ({exc}); overlapping with plain streams")
            return torch.cuda.Stream(device=device, priority=0), torch.cuda.Stream(device=device, priority=-1)

    def close(self) -> None:
        """Destroy the CU-masked HIP streams this trainer created (idempotent; the trainer must not run afterwards)."""
        own, self._own_streams = getattr(self, "_own_streams", None), None
        if own:
            try:
                torch.cuda.synchronize(self.feat.device)
                from . import _lib
                lib = _lib.load()
                for h in own:
                    lib.ggad_stream_destroy(h)
            except Exception:
                pass

    def start_stream(self, total_steps: int) -> None:
        """Persistent sampler thread for the next `total_steps` optimiser steps: it keeps up to 3 chunks of batches ready
        ACROSS `run_steps` calls (the reference-exact sampler is the end-to-end bottleneck; a thread per call would sit
        idle during validation and start every block with an empty queue).  It consumes exactly `total_steps` steps of the
        stream, never more, so the generator state handed back after training is the reference's."""
        import queue
        import threading
        if getattr(self, "_stream", None) is not None:
            raise RuntimeError("a batch stream is already running")
        q = queue.Queue(maxsize=3)
        # the first deliveries are small (the chunk ramp of `default_ramp`): the GPU starts after 16 batches of sampling instead of 150.
        # Later calls of the native sampler cover FOUR chunks: a call that starts on an epoch boundary begins with the epoch shuffle
        # of the ~1 M-element train list (targets, copy, swaps: ~3.4 ms during which its pipeline has nothing else to do), which
        # costs 23 us per batch on a call of 150 batches and 6 us on a call of 600
        # (the calls grow 1, 2, 4 chunks: a long call delivers late, and the GPU must not run dry while it is under way)
        ramp = self.default_ramp(int(total_steps))
        sizes, acc, want = [], 0, 1
        for k in ramp:
            if k < self.chunk_batches:
                sizes.append(k)
            else:
                acc += k
                if acc >= want * self.chunk_batches:
                    sizes.append(acc)
                    acc = 0
                    want = min(4, want * 2)
        if acc:
            sizes.append(acc)

        home = _other_llc_cpus()           # (decided on the launching thread: "other" = not the L3 this thread runs on now)

        def produce():
            try:
                if home:
                    # the native sampler confines its pipeline (5-6 spinning threads) to the L3 of the thread that calls it: keep that
                    # away from the L3 of the thread that launches the GPU work -- on the same L3 the two fight for its cores and
                    # the end-to-end rate drops to 4-5 M nodes/s although the sampler alone sustains 6.3 M and the GPU 7.1 M
                    try:
                        os.sched_setaffinity(threading.get_native_id(), home)
                    except OSError:
                        pass
                for k in sizes:
                    q.put((k, self.schedule.next_batches(k, self.sched_rank, self.sched_world)))
            except BaseException as exc:      # surface sampler errors in the consumer
                q.put((0, exc))
        th = threading.Thread(target=produce, daemon=True)
        self._stream = dict(q=q, thread=th, left=int(total_steps))
        th.start()

    def _stream_take(self, k: int):
        """k batches of the running sampler stream (the producer delivers whole chunks; what a shorter request leaves over
        is kept for the next one)."""
        st = self._stream
        bn, bl = self._pending
        while len(bn) < k:
            kk, item = st["q"].get()
            if isinstance(item, BaseException):
                self._stream = None
                raise item
            bn = bn + item[0]
            bl = bl + item[1]
        self._pending = (bn[k:], bl[k:])
        st["left"] -= k
        if st["left"] <= 0:
            st["thread"].join()
            self._stream = None
        return bn[:k], bl[:k]

    def run_steps(self, n_steps: int, prepared: Optional[Tuple[List[np.ndarray], List[np.ndarray]]] = None,
                  gather_hook=None) -> int:
        """Run n optimiser steps; returns nodes processed by THIS rank."""
        t_entry = time.perf_counter()
        sizes = self.default_ramp(n_steps)
        pos = [0]
        record = self._replay_begin()                 # (None unless the resident kernel runs and a replay could be needed)

        producer = None
        stream = getattr(self, "_stream", None) if prepared is None else None
        if stream is not None and n_steps > stream["left"]:
            raise RuntimeError("run_steps beyond the end of the running batch stream")
        if prepared is None and stream is None and self.prefetch and len(sizes) > 1:
            import queue
            import threading
            q = queue.Queue(maxsize=2)

            home = _other_llc_cpus()

            def produce():
                try:
                    if home:                      # the sampler's pipeline away from the launching thread's L3 (see start_stream)
                        try:
                            os.sched_setaffinity(threading.get_native_id(), home)
                        except OSError:
                            pass
                    for k in sizes:
                        q.put(self.schedule.next_batches(k, self.sched_rank, self.sched_world))
                except BaseException as exc:      # surface sampler errors in the consumer
                    q.put(exc)
            producer = threading.Thread(target=produce, daemon=True)
            producer.start()

        def take_raw(k):
            if prepared is not None:
                bn, bl = prepared[0][pos[0]:pos[0] + k], prepared[1][pos[0]:pos[0] + k]
                pos[0] += k
                return bn, bl
            if stream is not None:
                return self._stream_take(k)
            if producer is not None:
                item = q.get()
                if isinstance(item, BaseException):
                    raise item
                return item
            return self.schedule.next_batches(k, self.sched_rank, self.sched_world)

        def take(k):
            bn, bl = take_raw(k)
            if record is not None:
                record[1].extend(bn)
                record[2].extend(bl)
            return bn, bl

        def build(ch, bn, bl):
            ch.build(bn, bl) if gather_hook is None else gather_hook(ch, bn, bl)
            self._t_built = time.perf_counter()
            self.engine.xcd_prepare(ch)        # records of the resident chunk kernel: on the plan's stream, not on its 28 CUs

        nodes_seen, done = 0, 0
        split = getattr(self, "resident_split", False)
        if not self.overlap or len(sizes) == 1:
            if split:                                  # nothing runs beside the chunk kernel: the plan takes the whole chip
                self.chunk.xcd_skip = -1
            timing = os.environ.get("GGAD_RUN_TIMING")               # host phases of a chunk on stderr (us)
            for k in sizes:
                if timing:
                    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev_a.record()
                t0 = time.perf_counter()
                bn, bl = take(k)
                t1 = time.perf_counter()
                build(self.chunk, bn, bl)
                t2 = time.perf_counter()
                self.engine.train_chunk(self.chunk, self.allreduce, self.world, log_base=done, exchange=self.exchange)   # loss log slot = step index
                if timing:
                    ev_b.record()
                    t3 = time.perf_counter()
                    torch.cuda.synchronize()
                    t4 = time.perf_counter()
                    print("[run_steps] device: first event -> last event %.1f us" % (ev_a.elapsed_time(ev_b) * 1e3), file=sys.stderr)
                    print("[run_steps] %d batches: entry -> loop %.1f  take %.1f  build + xcd_prepare %.1f (build %.1f)  train_chunk call %.1f  device drain %.1f us"
                          % (k, (t0 - t_entry) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (self._t_built - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6), file=sys.stderr, flush=True)
                nodes_seen += sum(len(b) for b in bn)
                done += k
        else:
            if split:
                for ch in self.chunks:
                    ch.xcd_skip = self._xcd_skip
            outer = torch.cuda.current_stream()
            main = self.hi
            main.wait_stream(outer)
            side = self.side
            built = [torch.cuda.Event(), torch.cuda.Event()]
            freed = [torch.cuda.Event(), torch.cuda.Event()]
            side.wait_stream(main)
            bn, bl = take(sizes[0])
            with torch.cuda.stream(side):
                build(self.chunks[0], bn, bl)
                built[0].record(side)
            for c, k in enumerate(sizes):
                cur, nxt = c % 2, (c + 1) % 2
                nodes_seen += sum(len(b) for b in bn)
                if c + 1 < len(sizes):
                    bn, bl = take(sizes[c + 1])
                    with torch.cuda.stream(side):
                        if c >= 1:
                            side.wait_event(freed[nxt])        # the dense steps that read this buffer have finished
                        build(self.chunks[nxt], bn, bl)
                        built[nxt].record(side)
                main.wait_event(built[cur])
                self.chunk = self.chunks[cur]
                with torch.cuda.stream(main):
                    self.engine.train_chunk(self.chunks[cur], self.allreduce, self.world, log_base=done, exchange=self.exchange)
                freed[cur].record(main)
                done += k
            outer.wait_stream(main)
            outer.wait_stream(side)
        if producer is not None:
            producer.join()
        self.steps_done += n_steps
        if record is not None and self._window is not None:
            self._window.append((int(n_steps), record[1], record[2]))
        return nodes_seen

    # ---- recovery from a time-out of the resident chunk kernel
    _REPLAY_MAX_STEPS = 2048             # (a caller that never checks does not accumulate batches for ever)

    def _replay_begin(self):
        """Called at the top of `run_steps`.  While the resident kernel is in use (one process: with peers a replay would have to be
        collective) the first run after a check point copies the optimiser state aside -- one fused multi-tensor copy on the current
        stream -- and every run records its batches (references to the host arrays the sampler delivered)."""
        e = self.engine
        if not (e.resident and self.world == 1 and e.dev.type == "cuda"):
            self._window = None
            return None
        live = [e.params, e.exp_avg, e.exp_avg_sq, e.step_counter]
        if self._window is not None and sum(w[0] for w in self._window) > self._REPLAY_MAX_STEPS:
            # the window is about to be rolled: a time-out inside it must not be swallowed by the fresh snapshot
            if e.xcd_ws is not None and e.xcd_status()["error"]:
                self._replay_on_chain()
                return self._replay_begin()
        if self._window is None or sum(w[0] for w in self._window) > self._REPLAY_MAX_STEPS:
            if self._snap is None:
                self._snap = [torch.empty_like(t) for t in live]
            torch._foreach_copy_(self._snap, live)
            self._window = []
            self._window_steps0 = self.steps_done
        return (None, [], [])

    def _replay_on_chain(self) -> None:
        """The resident kernel reported a time-out since the last check point: some optimiser steps are missing or half applied.
        Restore the state of the check point, turn the resident kernel off for the rest of this trainer's life and run the
        recorded batches again through the launch chain (the same kernels `resident=False` uses: bit-identical to a run that
        never used the resident kernel)."""
        import warnings
        e = self.engine
        window, self._window = self._window, None
        warnings.warn("ggad_amd: the XCD-resident chunk kernel timed out (code %d); replaying %d optimiser steps on the launch "
                      "chain and continuing without it" % (e.xcd_status()["error"], sum(w[0] for w in window)))
        torch.cuda.synchronize(e.dev)
        e.set_resident_error(0)
        e.resident = False
        self.resident_split = False
        for ch in self.chunks:
            ch.xcd_skip = -1
        torch._foreach_copy_([e.params, e.exp_avg, e.exp_avg_sq, e.step_counter], self._snap)
        self.steps_done = self._window_steps0
        self.resident_fallbacks += 1
        for n, bn, bl in window:
            self.run_steps(n, prepared=(bn, bl))
        torch.cuda.synchronize(e.dev)
