#!/usr/bin/env python3
"""bench.py -- training nodes/s of the DGraph-Fin mini-batch GGAD hot path on N MI355X.

Workload (BASELINE.json metric, configs[4]): synthetic graph of DGraph-Fin size
(3,700,550 nodes, 73,105,508 directed entries, 17 features), emb 64, batches of 150 + 50 nodes
(`src/dgraph.yml`, `src/model_handler.py:317,342`), fp32.  One "step" = one optimiser step = one
batch of 200 nodes per GPU: batch sub-graph plan + 1-hop/2-hop gather-aggregate + encoder + loss +
backward + Adam.  Plans are built per chunk of batches INSIDE the timed region (a run starts with
small chunks so that the dense chain of chunk c overlaps the plan of chunk c+1 from the first
fraction of a millisecond on; `DGraphTrainer.default_ramp`).  Inputs resident in HBM before the
timed region: graph CSR, feature table, labels; the batch schedule (node ids, host memory) is
produced by the reference-exact host sampler beforehand -- `value` is therefore the GPU path's
throughput; `e2e_with_sampler` (extra key) times the reference's own window
(`src/model_handler.py:332-365`) with the sampler thread inside it.

    python bench.py --gpus 1 --steps 20 --warmup 5          # what the round driver runs
    python bench.py                                          # default: 9000 steps (steady state)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `roofline` refers to the 2-hop gather-aggregate
(`k_build_groups` + `k_gather2_items` + `k_gather2_combine`, the dominant launches of the path):
`achieved` = ALGORITHMIC bytes (74 B per gathered neighbour of every (batch, owner) occurrence:
68-byte feature row + 4-byte id + 2-byte streamed pair count) / launch time measured
with HIP events on the launch stream.  Because occurrences of a node in several batches of a chunk
share one fetch of its neighbour rows, algorithmic bytes exceed the HBM traffic; `hbm_bytes_per_neighbour`
/ `hbm_frac` (from the rocprofv3 PMC pass of the same command, profiles/) say what reaches memory.
Extra keys (single GPU): `steady_state` (long run after the timed region), `e2e_with_sampler`,
`fullgraph` (epoch times of the four full-graph configs), `cpu_baseline` (dense-faithful CPU port
of the reference's step on a bounded sample, rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9000)
    ap.add_argument("--warmup", type=int, default=150)
    ap.add_argument("--nodes", type=int, default=3_700_550)
    ap.add_argument("--entries", type=int, default=73_105_508)
    ap.add_argument("--feat", type=int, default=17)
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--max-degree", type=int, default=2000)
    ap.add_argument("--graph-kind", default="powerlaw", choices=["powerlaw", "er"])
    ap.add_argument("--chunk", type=int, default=150, help="batches planned per launch group (reference epoch = 150)")
    ap.add_argument("--cpu-batches", type=int, default=3, help="batches timed on the dense-faithful CPU port (0 = skip the CPU legs)")
    ap.add_argument("--cpu-epochs", type=int, default=4, help="epochs of 150 batches of the sparse CPU variant: 1 warm-up + the rest timed "
                    "(BASELINE.md section 3 protocol; 0 = the same few batches as the dense port)")
    ap.add_argument("--cpu-epoch-budget", type=float, default=120.0, help="seconds after which the sparse CPU variant stops adding epochs")
    ap.add_argument("--e2e-reps", type=int, default=5, help="repetitions of the end-to-end leg (median reported)")
    ap.add_argument("--dp-sampler", default="independent", choices=["independent", "shared"],
                    help="multi-GPU batch streams: one per rank (default, scales end to end) or the reference's one stream dealt to the ranks")
    ap.add_argument("--no-overlap", action="store_true", help="plan and dense steps on one stream")
    ap.add_argument("--dense-cus", type=int, default=-1, help="CUs reserved for the dense step chain when overlapping (-1 = by graph density: 32 or 64) "
                    "(CU-masked streams; 0 = plain streams with priorities)")
    ap.add_argument("--hop2", default="ldsw", choices=["ldsw", "global"],
                    help="ldsw: LDS counting per (tile, batch) + streamed per-pair counts (default); "
                         "global: atomics on per-batch counter slots in HBM (the fallback path)")
    ap.add_argument("--chain", type=int, default=0, choices=[0, 2], help="per-step kernel chain: 0 five launches (projection fused), 2 the generic six (include/ggad_hip.h: ggad_mb_step.chain)")
    ap.add_argument("--ramp", default="", help="comma-separated sizes of the first chunks of a run (default: DGraphTrainer.default_ramp)")
    ap.add_argument("--steady-steps", type=int, default=3000, help="extra leg after the timed region: steps of one long run (0 = skip; "
                    "skipped when --steps is already >= this)")
    ap.add_argument("--e2e-steps", type=int, default=3000, help="extra leg: steps timed with the reference-exact sampler inside the window (0 = skip)")
    ap.add_argument("--fullgraph-epochs", type=int, default=30, help="extra leg: epochs timed per full-graph config (0 = skip)")
    ap.add_argument("--sparse-entries", type=int, default=8_600_000, help="extra leg: directed entries of the public-degree DGraph regime (0 = skip)")
    ap.add_argument("--cpu-fullgraph-budget", type=float, default=12.0, help="seconds of CPU time per variant and full-graph config of the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (no steady-state / e2e / full-graph / CPU legs)")
    ap.add_argument("--exchange", default="oneshot", choices=["oneshot", "rccl"],
                    help="multi-GPU gradient exchange: oneshot = peer-mapped buffers written inside the Adam launch (falls back to "
                         "rccl when its self-test fails), rccl = torch.distributed all-reduce between backward and Adam")
    ap.add_argument("--dp-path", action="store_true", help="1 GPU only: run the data-parallel step chain (backward -> exchange + Adam "
                    "launch with a world of one) to measure what the multi-GPU step costs without the xGMI hop")
    ap.add_argument("--seed", type=int, default=72)
    return ap.parse_args()


def cpu_fullgraph_baseline(name, budget_s):
    """The reference's timing window `run.py:146 -> 214` (forward, loss block, backward, Adam) on the host cores of this box, for one
    full-graph config at its published size: the oracle's CSR step on 24 threads (`sparse_variant`) and, where
    the dense operands fit and finish (Reddit / Photo / Amazon), the dense-faithful restatement of the reference's N x N products
    (`dense_variant`).  >= 3 warm-up + >= 10 timed epochs unless `budget_s` seconds per variant run out first (the counts are reported)."""
    import random as _random
    import scipy.sparse as sp
    from ggad_amd.fullgraph_bench import make_dataset
    from ggad_amd.utils import normalize_adj
    from oracle import ggad_oracle as O
    threads = min(24, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    state = _random.getstate()
    _random.seed(0)
    np.random.seed(0)
    ds = make_dataset(name, 0)
    _random.setstate(state)
    n, f, h = ds["n"], ds["f"], 300
    an = (normalize_adj(ds["adj"]) + sp.eye(n)).tocsr()
    an.sort_indices()
    rw = (ds["adj"] + sp.eye(n)).tocsr()
    rw.sort_indices()
    adjn = (an.indptr, an.indices, an.data.astype(np.float32))
    raw = (rw.indptr, rw.indices, rw.data.astype(np.float32))
    gen = torch.Generator().manual_seed(0)
    shapes = {"gcn1.bias": (h,), "gcn1.fc.weight": (h, f), "gcn1.act.weight": (1,), "gcn2.bias": (h,), "gcn2.fc.weight": (h, h),
              "gcn2.act.weight": (1,), "fc1.weight": (h // 2, h), "fc2.weight": (h // 4, h // 2), "fc3.weight": (1, h // 4), "fc4.weight": (h, h)}
    feat_t = torch.from_numpy(ds["features"])
    abn, nrm = ds["abn_idx"], ds["normal_idx"]

    def fresh():
        P = {}
        for k in O.FULL_PARAM_ORDER:
            shp = shapes[k]
            t = torch.full(shp, 0.25) if k.endswith("act.weight") else (torch.zeros(shp) if k.endswith("bias") else
                                                                       torch.randn(shp, generator=gen) * (1.0 / np.sqrt(shp[-1])))
            P[k] = t.requires_grad_()
        return P, O.make_adam(list(P.values()), 1e-3, 0.0)

    def run(step, P, adam):
        ts, warm, t_all = [], 0, time.perf_counter()
        while True:
            t0 = time.perf_counter()
            adam.zero_grad()
            noise = torch.randn(len(abn), h) * ds["var"] + ds["mean"]
            loss = step(P, noise)
            loss.backward()
            adam.step()
            dt = time.perf_counter() - t0
            if warm < 3 and (warm == 0 or time.perf_counter() - t_all < 0.3 * budget_s):
                warm += 1
            else:
                ts.append(dt)
            if len(ts) >= 10 or (len(ts) >= 3 and time.perf_counter() - t_all > budget_s) or (len(ts) >= 1 and time.perf_counter() - t_all > 3 * budget_s):
                break
        return {"epoch_ms": 1e3 * float(np.median(ts)), "nodes_per_s": n / float(np.median(ts)), "warmup_epochs": warm, "epochs_timed": len(ts)}

    out = {"cores": threads, "kind": "port", "window": "run.py:146-214 (forward, loss block, backward, Adam), noise drawn per epoch"}
    P, adam = fresh()
    out["sparse_variant"] = run(lambda P, noise: _sparse_step(O, P, feat_t, adjn, raw, abn, nrm, noise), P, adam)
    if (name != "t_finance") and n <= 16000:
        A = torch.from_numpy(np.asarray(an.todense(), dtype=np.float32))
        Rw = torch.from_numpy(np.asarray(rw.todense(), dtype=np.float32))
        P, adam = fresh()
        out["dense_variant"] = run(lambda P, noise: O.full_step_dense(P, feat_t, A, Rw, abn, nrm, noise), P, adam)
    else:
        out["dense_variant"] = None      # 39,357^2 floats x 2 operands and ~5 TFLOP per epoch: not run
    return out


def _sparse_step(O, P, feat_t, adjn, raw, abn, nrm, noise):
    emb, comb, logits, con, eab = O.full_forward(P, feat_t, adjn, abn, nrm, noise, True)
    return O.full_loss(emb, logits, con, eab, raw, abn, nrm, by_column=True)[0]


def sparse_regime_leg(a, dev, feat, sp, w, W, fc):
    """Steady-state throughput on a DGraph-size graph with the PUBLIC degree (about 8.6 M directed entries, average 2.3): its own
    graph, plans and trainer; same features, split and batch schedule shape as the headline run."""
    import random as _random
    from ggad_amd import synth
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule, DGraphTrainer
    rp, ci = synth.make_graph_torch(a.nodes, a.sparse_entries, a.seed + 1, dev, max_degree=a.max_degree)
    g2 = DeviceGraph(rp, ci, dev)
    sched = BatchSchedule(sp["idx_train"], sp["idx_anomaly"], sp["labels"], 150, PyCompatRandom.from_python_state(_random.getstate()))
    tr = DGraphTrainer(g2, feat, a.emb, sched, lr=1e-3, weight_decay=0.007, chunk_batches=a.chunk, overlap=not a.no_overlap, chain=a.chain,
                       dense_cus=(None if a.dense_cus_arg < 0 else a.dense_cus_arg))
    tr.engine.load_params(w, W, fc)
    batch = sched.next_batches(a.steady_steps)
    tr.run_steps(min(300, a.steady_steps), prepared=(batch[0][:300], batch[1][:300]))      # warm-up: allocations, first plans
    torch.cuda.synchronize()
    ts = time.perf_counter()
    nn = tr.run_steps(a.steady_steps, prepared=batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - ts
    tr.check_exchange()
    return {"directed_entries": int(g2.nnz), "steps": a.steady_steps, "value": nn / dt, "unit": "nodes/s", "ms_per_step": 1e3 * dt / a.steady_steps,
            "dense_steps": "XCD-resident chunk kernel" if tr.engine.resident else "launch chain",
            "note": "steady-state run of this many steps (schedule prepared beforehand, plans inside) on the public-degree graph"}


def e2e_leg(a, trainer, sched, barrier, dist=None, world=1, rank=0):
    """The reference's own timing window (`src/model_handler.py:332 -> 365`: the per-batch random.shuffle of the pseudo-anomaly pool
    and the step) -- the bit-exact native sampler produces the batches in its own threads INSIDE the window, one persistent stream
    across all calls (`DGraphTrainer.start_stream`: look-ahead of up to three deliveries kept across `run_steps` calls).  300 untimed
    steps first (as the steady-state leg), then `--e2e-reps` windows; the median is the value, the spread is reported.  Also times
    the sampler ALONE (`sampler_us_per_batch`): with ~31 us per batch it caps this leg at ~6.4 M nodes/s whatever the GPU does."""
    n_e2e = (a.e2e_steps // trainer.chunk_batches) * trainer.chunk_batches or a.e2e_steps
    reps = max(1, int(a.e2e_reps))
    per = sched.bs + sched.n_pseudo
    sched.next_batches(150, trainer.sched_rank, trainer.sched_world)      # sampler alone: warm buffers and threads
    ts = time.perf_counter()
    sched.next_batches(600, trainer.sched_rank, trainer.sched_world)      # (shared mode: W batches generated per batch of this rank)
    sampler_us = 1e6 * (time.perf_counter() - ts) / 600.0
    warm = min(300, n_e2e)
    barrier()
    trainer.start_stream(warm + reps * n_e2e)
    trainer.run_steps(warm)
    barrier()
    vals = []
    for _ in range(reps):
        ts = time.perf_counter()
        n_nodes = trainer.run_steps(n_e2e)
        barrier()
        dt = time.perf_counter() - ts
        if world > 1:
            tt = torch.tensor([dt, float(n_nodes)], dtype=torch.float64, device=trainer.feat.device)
            dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
            dt, n_nodes = float(tt[0].item()), float(tt[1].item())
        vals.append(n_nodes / dt)
    trainer.check_exchange(dist if world > 1 else None)
    med = float(np.median(vals))
    return {"steps": n_e2e, "reps": reps, "value": med, "unit": "nodes/s", "ms_per_step": 1e3 * per * world / med,
            "min": float(min(vals)), "max": float(max(vals)), "spread": float((max(vals) - min(vals)) / med), "values": [float(v) for v in vals],
            "sampler_us_per_batch": sampler_us, "sampler_cap_nodes_per_s": per / (sampler_us * 1e-6) * world,
            "warmup_steps": warm,
            "note": "median of `reps` windows of `steps` optimiser steps, batch schedule generated INSIDE the window by the reference-exact sampler "
                    "threads (CPython random.shuffle of the 55k pool per batch, of the 1.05M train list per epoch), one persistent stream; "
                    "sampler_us_per_batch = the sampler alone per batch THIS RANK consumes (host-bound ceiling of the leg)"}


def respawn_command(n_gpus, argv, port=None):
    """The launch line a bare `python bench.py --gpus N ...` turns itself into (the driver's own form):
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>`."""
    if port is None:
        port = os.environ.get("MASTER_PORT")
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: become the launcher (one rank per GPU); rank 0's JSON line is this process's output
        cmd = respawn_command(a.gpus, sys.argv[1:])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(cmd[0], cmd)
    if world != a.gpus:
        sys.exit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {a.gpus} ... bench.py --gpus {a.gpus}), or run it bare")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one rank per GPU; GGAD_BENCH_BACKEND=gloo lets several ranks share one GPU (tests of the multi-rank path on a 1-GPU box)
    backend = os.environ.get("GGAD_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ggad_amd import synth
    from ggad_amd.dgraph import normalize_features, split_dgraphfin
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule, DGraphTrainer

    t0 = time.time()
    # ---------------- synthetic DGraph-size inputs (identical on every rank: same seeds)
    rowptr, col = synth.make_graph_torch(a.nodes, a.entries, a.seed, dev, kind=a.graph_kind, max_degree=a.max_degree)
    graph = DeviceGraph(rowptr, col, dev)
    feat_raw = synth.make_features(a.nodes, a.feat, a.seed)
    feat_np = normalize_features(feat_raw).astype(np.float32)            # src/model_handler.py:225
    feat = torch.from_numpy(feat_np).to(dev)
    labels0 = synth.make_labels(a.nodes, 15509.0 / 3700550.0, a.seed).astype(np.int32)
    split = split_dgraphfin(labels0, a.seed, with_test=False)
    import random as pyrandom
    rng = PyCompatRandom.from_python_state(pyrandom.getstate())          # continue the python stream (model_handler.py:30)
    sched_shared = BatchSchedule(split["idx_train"], split["idx_anomaly"], split["labels"], 150, rng)
    # multi-GPU: one batch stream per rank by default (the bit-exact sampler is ONE serial stream: dealt to W ranks every rank must
    # generate all W batches of a step, which caps an end-to-end run at ~6.4 M nodes/s for any W; the trajectory is not the
    # reference's at W > 1 in either mode).  W = 1: the reference's own stream.
    own_stream = world > 1 and a.dp_sampler == "independent"
    sched_own = BatchSchedule(split["idx_train"], split["idx_anomaly"], split["labels"], 150, PyCompatRandom(a.seed * 1000003 + rank + 1)) \
        if world > 1 else None
    sched = sched_own if own_stream else sched_shared
    allreduce = None
    exchange = None
    exchange_note = {"asked": a.exchange, "rank0_reason": None}
    if world > 1:
        def allreduce(t):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if a.exchange == "oneshot":
            # every rank must take the same path: a rank that cannot even allocate its buffer still joins the hand-shake
            from ggad_amd.exchange import OneShotExchange
            try:
                exchange = OneShotExchange(rank, world, a.emb + a.emb * a.feat + a.emb * a.emb, dev)
            except Exception as exc:
                exchange = None
                exchange_note["rank0_reason"] = f"fine-grained buffer not created on rank {rank}: {exc!r}"
            good = exchange.connect(dist) if exchange is not None else OneShotExchange.decline(dist, dev)
            if not good:                                 # e.g. no peer access between two devices: the RCCL all-reduce instead
                if exchange is not None:
                    exchange_note["rank0_reason"] = {"export": "this rank could not export its IPC handle", "peer_export": "a peer could not create / export its buffer",
                                                     "map": "a peer's buffer could not be mapped on some rank (no peer access / IPC refused)",
                                                     "selftest": "the known-pattern self-test exchange timed out or disagreed on some rank"}.get(
                                                         exchange.fail_stage, "declined by the agreement round")
                exchange = None
        else:
            exchange_note["rank0_reason"] = "--exchange rccl was asked for"
    trainer = DGraphTrainer(graph, feat, a.emb, sched, lr=1e-3, weight_decay=0.007, chunk_batches=a.chunk, rank=rank,
                            world_size=world, allreduce=allreduce, hop2=a.hop2, overlap=not a.no_overlap, chain=a.chain,
                            dense_cus=(None if a.dense_cus < 0 else a.dense_cus),
                            ramp=([int(x) for x in a.ramp.split(",") if x] if a.ramp else None), exchange=exchange, own_stream=own_stream,
                            # ranks that share a device (the 1-GPU tests of the launch line) cannot both keep a chunk kernel
                            # resident on the same XCD: they take the launch chain
                            resident=(False if world > torch.cuda.device_count() else None))
    a.dense_cus_arg = a.dense_cus
    a.dense_cus = getattr(trainer, "dense_cus", 0 if a.dense_cus < 0 else a.dense_cus)
    if a.dp_path and world == 1:
        # the data-parallel step chain on one GPU: backward -> k_xchg_adam with a world of one (publish to itself, flag, wait,
        # sum, Adam) -- what the multi-GPU step costs without the xGMI hop; --exchange rccl: backward -> no-op callback -> Adam
        if a.exchange == "oneshot":
            from ggad_amd.exchange import OneShotExchange
            trainer.exchange = OneShotExchange(0, 1, a.emb + a.emb * a.feat + a.emb * a.emb, dev)
        else:
            trainer.allreduce = lambda t: None
    torch.manual_seed(a.seed)
    w = torch.nn.init.xavier_uniform_(torch.empty(1, a.emb))
    W = torch.nn.init.xavier_uniform_(torch.empty(a.emb, a.feat))
    fc = torch.nn.Linear(a.emb, a.emb, bias=False).weight.detach()
    trainer.engine.load_params(w, W, fc)
    # batch schedule for warmup + timed steps, generated before the timed region (inputs of the hot path)
    warm = sched.next_batches(a.warmup, trainer.sched_rank, trainer.sched_world)
    timed = sched.next_batches(a.steps, trainer.sched_rank, trainer.sched_world)
    setup_s = time.time() - t0

    # ---------------- instrumentation of the dominant launches (2-hop gather) with HIP events on the launch stream
    import ctypes
    from ggad_amd import _lib
    lib = _lib.load()
    ev_pairs = []
    # A HIP event between two kernels of a stream is a barrier packet: ~6 us of idle device per event (profiles/r04_k20_timeline.txt).
    # A run of ONE chunk (the driver's --steps 20) therefore carries only the two events around the gather inside the timed region;
    # the pair-counting kernel and the resident chunk kernel are event-timed in an instrumented REPEAT of the same window right
    # after it (same batches, same sizes; not part of `value`).  Long runs (many chunks) are instrumented in place: 12 launches.
    instrument = {"all": len(trainer.default_ramp(a.steps)) > 1}

    def new_event():
        h = ctypes.c_void_p()
        _lib.check(lib.ggad_event_create(1, ctypes.byref(h)), "ggad_event_create")
        return h
    ev_pool = [new_event() for _ in range(48)]          # created before the timed region: (gather, pair counting) x 12 launches
    chunk_ev_pool = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
    chunk_ev = []

    def timed_build(chunk, bn, bl):
        # ggad_mb_plan_build records these two events around its gather launches (k_build_groups, k_gather2_items,
        # k_gather2_combine) on the stream they run on
        if len(ev_pairs) >= 12:           # the first 12 launches of the timed region are instrumented (host-side neighbour
            chunk.build(bn, bl)           # counting for the roofline costs ~1 s per 150-batch launch afterwards)
            return
        e0, e1, t0_, t1_ = ev_pool[4 * len(ev_pairs):4 * len(ev_pairs) + 4]
        chunk.gather2_events = (e0, e1)
        chunk.tile_events = (t0_, t1_) if instrument["all"] else None
        chunk.build(bn, bl)
        with_tile = chunk.tile_events is not None and chunk.last_hop2 == "ldsw"      # (the device-atomic fallback records none; an event
        chunk.gather2_events = None                                                   #  that was never recorded must not be queried: the
        chunk.tile_events = None                                                      #  error stays in the runtime's last-error slot)
        ev_pairs.append((e0, e1, bn, t0_ if with_tile else None, t1_))

    orig_train_chunk = trainer.engine.train_chunk

    def timed_train_chunk(ch, *args, **kw):
        # events on the stream the dense steps are launched on (torch's current stream inside `with torch.cuda.stream(main)`)
        if 2 * len(chunk_ev) + 2 > len(chunk_ev_pool) or not instrument["all"]:
            return orig_train_chunk(ch, *args, **kw)
        c0, c1 = chunk_ev_pool[2 * len(chunk_ev)], chunk_ev_pool[2 * len(chunk_ev) + 1]
        c0.record()
        orig_train_chunk(ch, *args, **kw)
        c1.record()
        chunk_ev.append((c0, c1, ch.n_batches))

    def hop2_neighbours(bn):
        """S2 = sum over batches of sum_{u in U_b} deg(u): the neighbours one gather2 launch reads (host numpy)."""
        rp, ci, deg = graph.rowptr_host, graph.col_host, graph.deg_host
        tot = 0
        for nodes in bn:
            parts = [ci[rp[v]:rp[v + 1]] for v in nodes]
            u = np.unique(np.concatenate(parts + [np.asarray(nodes, dtype=np.int32)]))
            tot += int(deg[u].sum())
        return tot

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- warmup
    if a.warmup > 0:
        trainer.run_steps(a.warmup, prepared=warm)
    barrier()
    # ---------------- timed region
    trainer.engine.train_chunk = timed_train_chunk
    t1 = time.perf_counter()
    nodes_local = trainer.run_steps(a.steps, prepared=timed, gather_hook=timed_build)
    if world > 1:
        torch.cuda.synchronize()
        elapsed_local = time.perf_counter() - t1      # this rank's own finish, before it waits for the others (reported per rank)
    barrier()
    elapsed = time.perf_counter() - t1
    if world == 1:
        elapsed_local = elapsed
    n_timed_pairs = len(ev_pairs)
    by_kernel_from = "the timed region"
    losses = trainer.engine.losses(a.steps)                # (before the repeat below overwrites the log slots)
    if not instrument["all"] and world == 1:
        instrument["all"] = True
        trainer.run_steps(a.steps, prepared=timed, gather_hook=timed_build)
        barrier()
        by_kernel_from = "an instrumented repeat of the timed window (same batches) right after it"
    trainer.engine.train_chunk = orig_train_chunk
    trainer.check_exchange(dist if world > 1 else None)

    def window(tr, prepared, n_steps):
        """One more K-step window (same bracket as the timed region): (max-over-ranks seconds, nodes of all ranks, this rank's seconds)."""
        barrier()
        ts = time.perf_counter()
        n_loc = tr.run_steps(n_steps, prepared=prepared)
        torch.cuda.synchronize()
        loc = time.perf_counter() - ts
        barrier()
        dt = time.perf_counter() - ts
        if world > 1:
            t3 = torch.tensor([dt, float(n_loc)], dtype=torch.float64, device=dev)
            dist.all_reduce(t3[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(t3[1:], op=dist.ReduceOp.SUM)
            return float(t3[0].item()), float(t3[1].item()), loc
        return dt, float(n_loc), loc

    # ---------------- the same window again, R times on FRESH batches (one 1.3 ms sample per headline; box-to-box and
    # run-to-run spread is the size of most changes).  `value` stays the first window above; these are extra keys.
    value_repeats = None
    if not a.no_extras and not a.dp_path and a.steps <= 3000:
        reps_n = 7 if a.steps <= 1000 else 3
        vals = []
        for _ in range(reps_n):
            fresh = sched.next_batches(a.steps, trainer.sched_rank, trainer.sched_world)
            dt_r, n_r, _ = window(trainer, fresh, a.steps)
            vals.append(n_r / dt_r)
        trainer.check_exchange(dist if world > 1 else None)
        value_repeats = {"reps": reps_n, "median": float(np.median(vals)), "min": float(min(vals)), "max": float(max(vals)),
                         "spread": float((max(vals) - min(vals)) / np.median(vals)), "values": [float(v) for v in vals],
                         "note": "the timed window repeated on fresh batches of the same schedule (same bracket: barrier + synchronize, max over "
                                 "ranks); `value` is the FIRST window, this is how far a single window moves"}

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        nn = torch.tensor([nodes_local], dtype=torch.float64, device=dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        nodes_total = float(nn.item())
    else:
        nodes_total = float(nodes_local)

    # ---------------- N > 1: what actually ran (the first multi-GPU run must explain itself)
    multi = None
    if world > 1:
        tl = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
        alll = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(alll, tl)
        per_rank = [1e3 * float(x.item()) / a.steps for x in alll]
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # a collective over the group that is actually initialised: must equal its size
        oneshot = trainer.exchange is not None
        multi = {"backend": dist.get_backend(), "rccl_ranks": int(dist.get_world_size()) if dist.get_backend() == "nccl" else 0,
                 "world_size": int(dist.get_world_size()), "allreduce_of_ones": float(ones.item()),
                 "devices_visible_per_rank": int(torch.cuda.device_count()), "ranks_share_a_device": bool(world > torch.cuda.device_count()),
                 "gradient_exchange": "oneshot peer writes + in-kernel sum" if oneshot else "rccl all-reduce",
                 "gradient_exchange_asked": exchange_note["asked"],
                 "gradient_exchange_fallback_reason": None if oneshot else exchange_note["rank0_reason"],
                 "dense_steps": "XCD-resident chunk kernel" if trainer.engine.resident else "launch chain (5 launches per step)",
                 "batch_streams": "independent (one per rank)" if own_stream else "shared (the reference's stream dealt to the ranks)",
                 "per_rank_ms_per_step": {"min": float(min(per_rank)), "median": float(np.median(per_rank)), "max": float(max(per_rank)),
                                          "values": [float(v) for v in per_rank],
                                          "note": "each rank's own finish of the timed window (before the closing barrier); ms_per_step of the line is the max-over-ranks bracket"}}

    # ---------------- roofline of the dominant kernel
    gather_ms, gather_nbrs, gather_batches, tile_ms = [], [], [], []
    tile_nbrs, nbr_cache = [], {}
    for i, (e0, e1, bn, t0_, t1_) in enumerate(ev_pairs):
        ms = ctypes.c_float()
        _lib.check(lib.ggad_event_elapsed_ms(e0, e1, ctypes.byref(ms)), "ggad_event_elapsed_ms")
        key = (len(bn), bn[0].__array_interface__["data"][0])              # (the repeat builds the same batches again)
        if key not in nbr_cache:
            nbr_cache[key] = hop2_neighbours(bn)
        if i < n_timed_pairs:                                                # (the roofline's launches: the timed region's own)
            gather_ms.append(float(ms.value))
            gather_nbrs.append(nbr_cache[key])
            gather_batches.append(len(bn))
        if t0_ is not None and lib.ggad_event_elapsed_ms(t0_, t1_, ctypes.byref(ms)) == 0:      # (none inside a one-chunk timed region)
            tile_ms.append(float(ms.value))
            tile_nbrs.append(nbr_cache[key])
    dense_ms = [c0.elapsed_time(c1) for c0, c1, _ in chunk_ev]
    dense_steps = [nb for _, _, nb in chunk_ev]
    del chunk_ev[:], chunk_ev_pool[:]          # (HIP events must not outlive the runtime: destroyed here, not at interpreter exit)
    for h in ev_pool:
        lib.ggad_event_destroy(h)
    mode = trainer.chunk.last_hop2
    sizes = trainer.default_ramp(a.steps)
    overlapped = bool(trainer.overlap and len(sizes) > 1)
    # algorithmic bytes per gathered neighbour (per occurrence), SURVEY.md section 8d: the feature row (4 F) + the column id (4) =
    # 72 B at F = 17.  (The 2-byte per-pair count "ldsw" streams besides is this implementation's traffic, not the algorithm's.)
    per_nbr = 4 * a.feat + 4
    alg_bytes = [per_nbr * nb for nb in gather_nbrs]
    ach = (sum(alg_bytes) / 1e9) / (sum(gather_ms) / 1e3) if gather_ms and sum(gather_ms) > 0 else None
    kname = ("k_build_groups + k_gather2_items + k_gather2_combine (node-major 2-hop gather-aggregate: work items of <= 8 occurrences "
             "x 256 neighbours, streamed pair counts)") if mode == "ldsw" else "k_gather2 (2-hop gather-aggregate, device-atomic counters)"
    # HBM-side bytes per gathered neighbour from the rocprofv3 PMC passes of this command (profiles/, FETCH_SIZE doubled as
    # MI355X_MICROARCH prescribes for gfx950), keyed by batches per launch; nearest measured chunk size, else null
    traffic = hbm_per_nbr = hbm_src = tile_hbm_per_nbr = None
    pmc_name = next((f for f in ("r06_pmc_gather2_items.json", "r05_pmc_gather2_items.json", "r04_pmc_gather2_items.json", "r03_pmc_gather2_items.json")
                     if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
    if mode == "ldsw" and pmc_name and gather_nbrs:
        with open(os.path.join(ROOT, "profiles", pmc_name)) as fh:
            pmc_doc = json.load(fh)
            table = pmc_doc.get("by_batches_per_launch", {})
            pmc_commit = pmc_doc.get("commit", "unknown")
        if table:
            mean_b = float(np.mean(gather_batches))
            key = min(table, key=lambda k: abs(float(k) - mean_b))
            if abs(float(key) - mean_b) <= 0.5 * mean_b:
                hbm_per_nbr = float(table[key]["hbm_bytes_per_neighbour"])
                traffic = hbm_per_nbr * float(np.mean(gather_nbrs))
                tile_hbm_per_nbr = table[key].get("k_tile_counts_hbm_bytes_per_neighbour")
                hbm_src = f"profiles/{pmc_name}[{key} batches per launch] (PMC passes of the build whose gather sources last changed in {pmc_commit})"
    avg_ms = float(np.mean(gather_ms)) if gather_ms else None
    roofline = {"kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic,
                "achieved_is": "algorithmic bytes / launch time: every (batch, owner) occurrence is charged its neighbours' rows, "
                               "although occurrences of a node in several batches of a launch share one fetch -- not HBM traffic",
                "hbm_bytes_per_neighbour": hbm_per_nbr, "hbm_traffic_source": hbm_src,
                "hbm_frac": (traffic / 1e9 / (avg_ms / 1e3) / HBM_PEAK_GBS) if (traffic and avg_ms) else None,
                "launches": len(gather_ms), "avg_launch_ms": avg_ms,
                "batches_per_launch": gather_batches,
                "alg_bytes_per_launch": float(np.mean(alg_bytes)) if alg_bytes else None, "alg_bytes_per_neighbour": per_nbr,
                "gather_share_of_timed_region": (sum(gather_ms) / 1e3 * (len(sizes) / max(1, len(gather_ms)))) / elapsed if gather_ms else None,
                # with overlap the launches run on the plan stream's CU partition while the dense chain of the previous
                # chunk runs on the other CUs (the first chunk of a run has nothing to overlap with)
                "concurrent_with_dense_chain": overlapped,
                "cus": (256 - a.dense_cus) if (overlapped and a.dense_cus > 0) else 256}

    # ---------------- the rest of the timed region, kernel by kernel (the gather is a third of it)
    n_chunks_run = len(sizes)
    scale_launches = n_chunks_run / max(1, len(gather_ms))
    by_kernel = [{"kernel": "k_build_groups + k_gather2_items + k_gather2_combine", "replaces": "src/graphsage.py:335-348 (2-hop mask.mm)",
                  "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": roofline["frac"], "traffic": traffic,
                  "avg_launch_ms": avg_ms, "share_of_timed_region": roofline["gather_share_of_timed_region"]}]
    if tile_ms:
        # pair counting: reads the 4-byte column id of every (batch, owner) neighbour and writes its 2-byte pair count
        t_alg = [6.0 * nb for nb in tile_nbrs]
        t_ach = (sum(t_alg) / 1e9) / (sum(tile_ms) / 1e3)
        t_traffic = (float(tile_hbm_per_nbr) * float(np.mean(gather_nbrs))) if tile_hbm_per_nbr else None
        by_kernel.append({"kernel": "k_tile_counts (LDS pair counting per (tile of 32,768 ids, batch))", "replaces": "src/graphsage.py:335-348 (the U x U2 mask)",
                          "bound": "hbm", "achieved": t_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": t_ach / HBM_PEAK_GBS,
                          "traffic": t_traffic, "alg_bytes_per_neighbour": 6, "hbm_bytes_per_neighbour": tile_hbm_per_nbr,
                          "avg_launch_ms": float(np.mean(tile_ms)),
                          "note": "fetches several times its input (one 139-KB workgroup per CU: no round trip overlaps another, DESIGN 9.2)",
                          "share_of_timed_region": (sum(tile_ms) / 1e3 * scale_launches) / elapsed})
    if dense_ms:
        us_step = 1e3 * sum(dense_ms) / max(1, sum(dense_steps))
        resident = bool(trainer.engine.resident)
        # issue time of the chunk kernel's instructions from the hardware counters of THIS command (scripts/pmc_chunk_xcd.sh ->
        # profiles/r05_pmc_chunk_xcd.json, stamped with the commit of its sources, tests/test_profiles.py): SQ_ACTIVE_INST_ANY x 4
        # clocks over the 112 SIMDs of the XCD's 28 compute units at 2.4 GHz
        cx = None
        cx_path = os.path.join(ROOT, "profiles", "r05_pmc_chunk_xcd.json")
        if resident and os.path.exists(cx_path):
            with open(cx_path) as fh:
                cx = json.load(fh)
        issue_us = float(cx["issue_us_per_step"]) if cx and cx.get("issue_us_per_step") else None
        by_kernel.append({"kernel": ("k_train_chunk_xcd (all optimiser steps of a chunk in one launch resident on one XCD, with its memset and "
                                     "Adam-scalar launch)" if resident else "launch chain of the dense steps (5 launches per step)"),
                          "replaces": "src/graphsage.py:395-454,171-258 + src/model_handler.py:356-364 (encoder, loss, backward, Adam)",
                          "bound": "issue", "us_per_step": us_step, "avg_launch_ms": float(np.mean(dense_ms)),
                          # < 30 MFLOP and < 2 MB per step: ~1 % of the FP32 and HBM rooflines of the whole chip; what the launch is made of
                          # is instruction issue on one XCD and waiting (69 % of its wave cycles are parked at a barrier or a load)
                          "issue_floor_us_per_step": issue_us, "frac": (issue_us / us_step) if issue_us else None,
                          "issue_floor_source": (f"profiles/r05_pmc_chunk_xcd.json (SQ_ACTIVE_INST_ANY x 4 clocks / 112 SIMDs / 2.4 GHz; PMC passes of the "
                                                 f"build whose chunk-kernel sources last changed in {cx.get('commit', 'unknown')})") if cx else None,
                          "wave_instructions_per_wave_per_step": cx.get("wave_instructions_per_wave_per_step") if cx else None,
                          "waves_parked_frac": (cx["counters"]["SQ_WAIT_ANY"] / cx["counters"]["SQ_WAVE_CYCLES"]) if cx else None,
                          "flop_per_step": 3.0e7, "bytes_per_step": 2.0e6,
                          "frac_of_fp32_peak": 3.0e7 / (us_step * 1e-6) / 157.3e12, "frac_of_hbm_peak": 2.0e6 / (us_step * 1e-6) / 8.0e12,
                          "share_of_timed_region": (sum(dense_ms) / 1e3) / elapsed})
    if roofline["frac"] and roofline["frac"] > 1.0:
        roofline["frac_note"] = ("above 1 because `achieved` charges every (batch, owner) occurrence its neighbours' rows (the reference gathers "
                                 "batch by batch) while a launch of many batches fetches the rows of a node that occurs in several of them "
                                 "once: `hbm_frac` is the fraction of the HBM peak the launch actually moves")
    roofline["by_kernel"] = by_kernel
    roofline["by_kernel_measured_in"] = by_kernel_from
    roofline["timed_region_accounted"] = float(sum(k["share_of_timed_region"] or 0.0 for k in by_kernel))

    # ---------------- extra legs (single GPU): steady state, end to end with the sampler, the full-graph programs
    extras = {}
    if rank == 0 and world == 1 and not a.no_extras and not a.dp_path:
        if a.steady_steps > a.steps:
            # (300 untimed steps first: the buffers of 150-batch chunks grow on first use -- a device synchronisation each --, the
            #  timed region above only planned 20 batches)
            wb = sched.next_batches(min(300, a.steady_steps), rank, world)
            trainer.run_steps(len(wb[0]), prepared=wb)
            batch = sched.next_batches(a.steady_steps, rank, world)
            barrier()
            ts = time.perf_counter()
            n_st = trainer.run_steps(a.steady_steps, prepared=batch)
            barrier()
            dt = time.perf_counter() - ts
            extras["steady_state"] = {"steps": a.steady_steps, "value": n_st / dt, "unit": "nodes/s", "ms_per_step": 1e3 * dt / a.steady_steps,
                                      "note": "one run of this many steps after the timed region and 300 untimed steps (schedule prepared beforehand, plans inside)"}
        if a.e2e_steps > 0:
            extras["e2e_with_sampler"] = e2e_leg(a, trainer, sched, barrier)
        if a.sparse_entries > 0 and a.steady_steps > a.steps:
            # BASELINE.md section 3 quotes DGraph-Fin in two degree regimes: 73.1 M entries (the headline run above) and the
            # public graph's ~8.6 M directed entries (average degree 2.3): same nodes, same schedule, its own graph and plans
            try:
                extras["dgraph_sparse_regime"] = sparse_regime_leg(a, dev, feat, split, w, W, fc)
            except Exception as exc:
                extras["dgraph_sparse_regime"] = {"error": repr(exc)}
        if a.fullgraph_epochs > 0:
            try:
                from ggad_amd.fullgraph_bench import bench_fullgraph
                extras["fullgraph"] = bench_fullgraph(dev, a.fullgraph_epochs)
            except Exception as exc:          # the extra leg must never take the headline line down
                extras["fullgraph"] = {"error": repr(exc)}
            if a.cpu_batches > 0 and isinstance(extras["fullgraph"], dict) and "error" not in extras["fullgraph"]:
                for name in list(extras["fullgraph"].keys()):
                    try:
                        extras["fullgraph"][name]["cpu_baseline"] = cpu_fullgraph_baseline(name, a.cpu_fullgraph_budget)
                        ep = extras["fullgraph"][name]["cpu_baseline"]["sparse_variant"]["epoch_ms"]
                        extras["fullgraph"][name]["gpu_over_cpu_sparse"] = ep / extras["fullgraph"][name]["epoch_ms"]
                    except Exception as exc:
                        extras["fullgraph"][name]["cpu_baseline"] = {"error": repr(exc)}

    if world > 1 and not a.no_extras and a.e2e_steps > 0:
        # multi-GPU: `value` above runs on a pre-generated schedule; what a ModelHandler.train() user sees has the sampler inside the
        # window -- per sampler mode, on every rank (all ranks take part; rank 0 reports)
        legs = {}
        for mode_name in ("independent", "shared"):
            if mode_name == "independent":
                trainer.schedule, trainer.sched_rank, trainer.sched_world = sched_own, 0, 1
            else:
                trainer.schedule, trainer.sched_rank, trainer.sched_world = sched_shared, rank, world
            try:
                legs[mode_name] = e2e_leg(a, trainer, trainer.schedule, barrier, dist, world, rank)
            except Exception as exc:
                legs[mode_name] = {"error": repr(exc)}
        legs["default_mode"] = a.dp_sampler
        extras["e2e_with_sampler"] = legs

    # ---------------- CPU baseline: dense-faithful port of the reference's step, bounded sample
    cpu = None
    if rank == 0 and world == 1 and a.cpu_batches > 0 and not a.no_extras:
        from oracle import ggad_oracle as O
        ncores = os.cpu_count() or 1
        threads = min(24, ncores)                                       # README.md:21 "24-core CPU"
        torch.set_num_threads(threads)
        adj = O.LazyAdjLists(graph.rowptr_host, graph.col_host)
        feat_t = torch.from_numpy(feat_np)
        p = O.MiniParams(w.clone().requires_grad_(), W.clone().requires_grad_(), fc.clone().requires_grad_())
        opt = O.make_adam(p.tensors(), 1e-3, 0.007)
        nb = min(a.cpu_batches, len(timed[0]))
        for b in range(nb):
            adj.warm(timed[0][b])
        O.dense_port_step(adj, feat_t, p, opt, timed[0][0], timed[1][0])     # warm-up step (allocator, threads)
        tc = time.perf_counter()
        n_cpu = 0
        for b in range(nb):
            O.dense_port_step(adj, feat_t, p, opt, timed[0][b], timed[1][b])
            n_cpu += len(timed[0][b])
        cpu_s = time.perf_counter() - tc
        cpu = {"value": n_cpu / cpu_s, "unit": "nodes/s", "cores": threads, "kind": "port",
               "sample": f"{nb} of the timed batches ({n_cpu} nodes), dense-mask step incl. backward+Adam, {cpu_s:.1f} s"}
        # the "fair CPU" number of SURVEY 8d: same step with the sparse closed-form aggregation instead of the dense masks, by
        # BASELINE.md section 3's protocol: 1 warm-up epoch + 3 timed epochs of 150 batches, median epoch (the dense-faithful port
        # above cannot follow it: 7 s per batch = 17 minutes per epoch at this degree -- it stays at `--cpu-batches` batches)
        p2 = O.MiniParams(w.clone().requires_grad_(), W.clone().requires_grad_(), fc.clone().requires_grad_())
        opt2 = O.make_adam(p2.tensors(), 1e-3, 0.007)

        def sparse_batches(bn, bl):
            n_ = 0
            for nodes_b, lab_b in zip(bn, bl):
                agg = O.aggregate_batch(graph.rowptr_host, graph.col_host, feat_np, nodes_b, True)
                opt2.zero_grad()
                O.batch_loss(p2, agg, lab_b)[0].backward()
                opt2.step()
                n_ += len(nodes_b)
            return n_
        if a.cpu_epochs >= 2:
            t_all, ep_s, ep_nodes = time.perf_counter(), [], []
            for ep in range(a.cpu_epochs):
                bn, bl = sched.next_batches(150, trainer.sched_rank, trainer.sched_world)
                ts = time.perf_counter()
                n_sp = sparse_batches(bn, bl)
                if ep > 0:
                    ep_s.append(time.perf_counter() - ts)
                    ep_nodes.append(n_sp)
                if time.perf_counter() - t_all > a.cpu_epoch_budget and ep_s:
                    break
            med = float(np.median(ep_s))
            cpu["sparse_variant"] = {"value": float(np.median(ep_nodes)) / med, "unit": "nodes/s", "cores": threads, "epoch_s": med,
                                     "warmup_epochs": 1, "epochs_timed": len(ep_s),
                                     "sample": f"1 warm-up + {len(ep_s)} timed epochs of 150 batches (30,000 nodes each), median epoch {med:.1f} s; sparse "
                                               "aggregation (numpy) + torch autograd + Adam; BASELINE.md section 3 protocol"}
        else:
            ts = time.perf_counter()
            n_sp = sparse_batches(timed[0][:nb], timed[1][:nb])
            sp_s = time.perf_counter() - ts
            cpu["sparse_variant"] = {"value": n_sp / sp_s, "unit": "nodes/s",
                                     "sample": f"same {nb} batches, sparse aggregation (numpy) + torch autograd + Adam, {sp_s:.1f} s"}
        cpu["dense_port_note"] = ("the dense-faithful port (the reference's own ops: dense U x U2 masks) takes ~7 s per batch at this degree, "
                                  "i.e. ~17 min per epoch: it is timed on a few batches, the sparse variant on whole epochs")

    if rank == 0:
        value = nodes_total / elapsed
        out = {
            "metric": "training nodes/sec (DGraph-Fin mini-batch GGAD)", "value": value, "unit": "nodes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DGraph-Fin-size synthetic graph, mini-batch GGAD (GCN encoder)", "nodes": a.nodes,
                       "directed_entries": int(graph.nnz), "feat": a.feat, "emb": a.emb, "batch": "150+50",
                       "graph": f"{a.graph_kind}(alpha=2.1,max_degree={a.max_degree})", "chunk_batches": trainer.chunk_batches,
                       "chunks_of_timed_region": sizes[:16], "hop2": a.hop2, "overlap": overlapped,
                       "dense_cus": (a.dense_cus if overlapped else 0), "chain": a.chain,
                       "dense_steps": ("XCD-resident chunk kernel: one launch per chunk on %d CUs of one XCD%s" %
                                       (trainer.engine.xcd_wgs or 32, ", plan kernels of the next chunk on the other 7 XCDs" if overlapped else "")
                                       if trainer.engine.resident else "launch chain: 5 launches per step"),
                       "parallelism": f"dp{world}", "optimizer": "adam(lr=1e-3,wd=0.007)",
                       "batch_streams": ("the reference's stream" if world == 1 else
                                         ("one per rank (seed * 1000003 + rank + 1)" if own_stream else "the reference's one stream dealt to the ranks")),
                       "gradient_exchange": (None if world == 1 else ("oneshot peer writes + in-kernel sum" if trainer.exchange is not None
                                                                      else "rccl all-reduce"))},
            "value_is": "GPU path (plans + dense steps inside the window; batch schedule prepared by the host sampler beforehand)",
            "roofline": roofline, "cpu_baseline": cpu,
            # two ratios, named for what they divide by (measurement hygiene): the reference's own dense-mask step (a few
            # batches) and the sparse CPU variant under BASELINE.md's protocol (whole epochs)
            "gpu_over_cpu": (value / cpu["value"]) if cpu else None,
            "gpu_over_cpu_is": "value / cpu_baseline.value (dense-faithful port of the reference's step)" if cpu else None,
            "gpu_over_cpu_sparse": (value / cpu["sparse_variant"]["value"]) if (cpu and cpu.get("sparse_variant")) else None,
            "first_loss": float(losses[0][0]), "last_loss": float(losses[-1][0]), "setup_s": setup_s,
        }
        out.update(extras)
    else:
        out = None
    # ---------------- N > 1: the same window under the CONTRACTED design beside the default (RCCL all-reduce, one shared
    # schedule) -- three short legs on fresh batches.  Every rank takes part; a watchdog prints the line without them if a leg hangs.
    if world > 1 and not a.no_extras:
        import threading
        done_flag = threading.Event()

        def give_up():
            if done_flag.wait(float(os.environ.get("GGAD_BENCH_ALT_TIMEOUT_S", "240"))):
                return
            if out is not None:
                out["multi_gpu"] = dict(multi or {}, alt_legs={"error": "an alternative leg did not finish in time; the line is printed without it"})
                out["value_repeats"] = value_repeats
                sys.stdout.flush()
                print(json.dumps(out), flush=True)
            os._exit(0)
        threading.Thread(target=give_up, daemon=True).start()
        alt = {}
        try:
            # (a) the default trainer on the OTHER sampler mode (pre-generated batches: the GPU path sees different node lists, nothing else)
            o_sched, o_r, o_w = (sched_shared, rank, world) if own_stream else (sched_own, 0, 1)
            o_sched.next_batches(a.warmup, o_r, o_w)
            fresh = o_sched.next_batches(a.steps, o_r, o_w)
            dt_a, n_a, _ = window(trainer, fresh, a.steps)
            alt["dp_sampler_" + ("shared" if own_stream else "independent")] = {"value": n_a / dt_a, "ms_per_step": 1e3 * dt_a / a.steps}
            trainer.check_exchange(dist)
            # (b) the other gradient exchange: a second trainer on the same graph / features / weights
            other_is_rccl = trainer.exchange is not None
            if other_is_rccl or exchange_note["asked"] == "rccl":
                ex2 = None
                if not other_is_rccl:                                   # the run was asked for rccl: try the one-shot path as the alternative
                    from ggad_amd.exchange import OneShotExchange
                    try:
                        ex2 = OneShotExchange(rank, world, a.emb + a.emb * a.feat + a.emb * a.emb, dev)
                    except Exception:
                        ex2 = None
                    good2 = ex2.connect(dist) if ex2 is not None else OneShotExchange.decline(dist, dev)
                    if not good2:
                        ex2 = None
                if other_is_rccl or ex2 is not None:
                    tr2 = DGraphTrainer(graph, feat, a.emb, sched, lr=1e-3, weight_decay=0.007, chunk_batches=a.chunk, rank=rank, world_size=world,
                                        allreduce=allreduce, hop2=a.hop2, overlap=not a.no_overlap, chain=a.chain,
                                        dense_cus=(None if a.dense_cus_arg < 0 else a.dense_cus_arg), exchange=ex2, own_stream=own_stream,
                                        resident=(False if world > torch.cuda.device_count() else None))
                    tr2.engine.load_params(w, W, fc)
                    wb = sched.next_batches(max(a.warmup, 1), trainer.sched_rank, trainer.sched_world)
                    tr2.run_steps(len(wb[0]), prepared=wb)
                    fresh = sched.next_batches(a.steps, trainer.sched_rank, trainer.sched_world)
                    dt_b, n_b, _ = window(tr2, fresh, a.steps)
                    tr2.check_exchange(dist)
                    alt["exchange_" + ("rccl" if other_is_rccl else "oneshot")] = {
                        "value": n_b / dt_b, "ms_per_step": 1e3 * dt_b / a.steps,
                        "dense_steps": "XCD-resident chunk kernel" if tr2.engine.resident else "launch chain (5 launches per step)"}
                    tr2.close()
                else:
                    alt["exchange_oneshot"] = {"error": "one-shot exchange not available between these ranks"}
        except Exception as exc:
            alt["error"] = repr(exc)
        done_flag.set()
        if multi is not None:
            multi["alt_legs"] = alt
            multi["alt_legs_note"] = ("same K-step window (fresh batches, same bracket) with ONE thing changed against the default of this line: the sampler "
                                      "mode, the gradient exchange.  SURVEY 8e's contracted design = exchange rccl + dp_sampler shared")
    if out is not None:
        out["multi_gpu"] = multi
        out["value_repeats"] = value_repeats

    # The JSON line must be the LAST thing on stdout: RCCL prints its version banner through C stdio (buffered, flushed at
    # exit, i.e. after a Python print).  Every rank flushes its C streams, all ranks meet, THEN rank 0 prints; the process
    # group is torn down afterwards.
    def flush_c():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
    flush_c()
    pg = torch.distributed.is_available() and torch.distributed.is_initialized()
    if pg:
        torch.distributed.barrier()
        torch.cuda.synchronize()
        flush_c()
    if out is not None:
        print(json.dumps(out), flush=True)
    trainer.close()
    if pg:
        torch.distributed.destroy_process_group()
        os._exit(0) if rank != 0 else None       # non-zero ranks leave without running exit-time stdio flushes


if __name__ == "__main__":
    main()
