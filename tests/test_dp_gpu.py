"""Data-parallel path on the GPU with 2 ranks sharing ONE device (gloo carries the CUDA tensors): the very code the
8-GPU run uses -- rank-dealt batches, chunk plans on CU-masked streams, one all-reduce of the packed gradient block
per step between the backward and the Adam kernel -- checked against a one-process run that averages the two
gradients itself, and the multi-rank `bench.py` launch line end to end."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    from ggad_amd import synth
    from oracle import ggad_oracle as O
    n = 40000
    rowptr, col = synth.make_graph(n, 400000, 6, kind="powerlaw", max_degree=500)
    feat = O.normalize_rows(synth.make_features(n, 17, 6)).astype(np.float32)
    labels = np.zeros(n, dtype=np.int64)
    pool = np.arange(500, 1300)
    labels[pool] = 1
    train = np.arange(2000, 30000)
    torch.manual_seed(8)
    w = torch.nn.init.xavier_uniform_(torch.empty(1, 64))
    W = torch.nn.init.xavier_uniform_(torch.empty(64, 17))
    fc = torch.nn.init.xavier_uniform_(torch.empty(64, 64))
    return rowptr, col, feat, labels, train, pool, (w, W, fc)


def _worker(rank, world, port, steps, out_dir, oneshot=False, own_stream=False, resident=False):
    import torch.distributed as dist
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule, DGraphTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if resident:
        os.environ["GGAD_XCD_ID"] = str(rank)      # the ranks share ONE device: each rank's chunk kernel on an XCD of its own
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rowptr, col, feat, labels, train, pool, (w, W, fc) = _inputs()
    graph = DeviceGraph(rowptr, col, "cuda:0")
    ft = torch.from_numpy(feat).to("cuda:0")
    sched = BatchSchedule(train.copy(), pool.copy(), labels, 90, PyCompatRandom(72 + (rank if own_stream else 0)), n_pseudo=30,
                          batches_per_epoch=7)
    exchange = None
    if oneshot:
        from ggad_amd.exchange import OneShotExchange
        exchange = OneShotExchange(rank, world, 64 + 64 * 17 + 64 * 64, "cuda:0")
        assert exchange.connect(dist), "one-shot exchange: IPC hand-shake or self-test failed"
    if resident:
        # (no overlap: the trainer's CU masks and index-skipping plans are laid out for a chunk kernel on XCD 0)
        tr = DGraphTrainer(graph, ft, 64, sched, chunk_batches=3, rank=rank, world_size=world, allreduce=None, exchange=exchange,
                           own_stream=own_stream, overlap=False, resident=True)
        assert tr.engine.resident and tr.exchange is not None
    else:
        tr = DGraphTrainer(graph, ft, 64, sched, chunk_batches=3, rank=rank, world_size=world,
                           allreduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM), exchange=exchange, own_stream=own_stream,
                           resident=False)      # the ranks share ONE device: two XCD-resident chunk kernels cannot both hold XCD 0
        assert tr.overlap and (tr.exchange is not None) == oneshot
    tr.engine.load_params(w, W, fc)
    tr.run_steps(steps)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), tr.engine.params.cpu().numpy())
    np.save(os.path.join(out_dir, f"seen_{rank}.npy"), np.array([sched.global_batch]))
    if exchange is not None:
        assert exchange.error() == 0
    if resident:
        st = tr.engine.xcd_status()
        assert st["error"] == 0 and st["xcc"] == rank, st
    dist.barrier()
    if exchange is not None:
        exchange.close()
    dist.destroy_process_group()


def test_one_shot_exchange_two_processes_one_gpu_bit_equal_to_allreduce(tmp_path):
    """The one-shot exchange (each rank's gradient-Adam launch writes its block into the peer's IPC-mapped fine-grained
    buffer, polls the peer's flags, sums in rank order) between two PROCESSES sharing one device: after 8 steps both ranks
    hold bit-identical weights, and they are bit-identical to the all-reduce(SUM) + Adam(1 / W) path (g0 + g1 has one order)."""
    import torch.multiprocessing as mp
    steps, world = 8, 2
    a, b = tmp_path / "oneshot", tmp_path / "allreduce"
    a.mkdir(); b.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), steps, str(a), True), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), steps, str(b), False), nprocs=world, join=True)
    p0, p1 = np.load(a / "params_0.npy"), np.load(a / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(p0, np.load(b / "params_0.npy"))


@pytest.mark.parametrize("world", [2, 4])
def test_one_shot_exchange_inside_resident_chunk_kernels_of_several_ranks(tmp_path, world):
    """The exchange inside phase E of the XCD-resident chunk kernel (`k_train_chunk_xcd<2>`) between PROCESSES: what `bench.py --gpus N`
    runs on a node, minus the xGMI hop.  The ranks share one device here, so each rank's chunk kernel takes an XCD of its own
    (GGAD_XCD_ID = rank).  After 6 steps all ranks hold bit-identical weights, equal to the launch chain + all-reduce path to the
    tolerance the resident kernel is held to against the chain (2e-6)."""
    import torch.multiprocessing as mp
    steps = 6
    a, b = tmp_path / "resident", tmp_path / "allreduce"
    a.mkdir(); b.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), steps, str(a), True, False, True), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), steps, str(b), False), nprocs=world, join=True)
    ps = [np.load(a / f"params_{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(ps[0], ps[r])
    assert np.isfinite(ps[0]).all()
    np.testing.assert_allclose(ps[0], np.load(b / "params_0.npy"), atol=2e-6, rtol=2e-5)


@pytest.mark.parametrize("world", [4, 8])
def test_one_shot_exchange_four_and_eight_processes_one_gpu(tmp_path, world):
    """W = 4 and W = 8 PROCESSES on one device through the one-shot exchange (what a node of 8 GPUs runs, minus the xGMI hop): every
    rank's buffer holds 2 x W slots, each step is W peer writes + W polls per parameter, bounded by the wall clock.  After 6 steps all
    ranks hold bit-identical weights, equal to the all-reduce(SUM) + Adam(1 / W) path."""
    import torch.multiprocessing as mp
    steps = 6
    a, b = tmp_path / "oneshot", tmp_path / "allreduce"
    a.mkdir(); b.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), steps, str(a), True), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), steps, str(b), False), nprocs=world, join=True)
    ps = [np.load(a / f"params_{r}.npy") for r in range(world)]
    for r in range(1, world):
        np.testing.assert_array_equal(ps[0], ps[r])
    ref = np.load(b / "params_0.npy")
    assert np.isfinite(ps[0]).all()
    # the rank-order sum g0 + g1 + ... + g(W-1) of the exchange against the all-reduce's tree: equal to fp32 round-off of a sum of W terms
    np.testing.assert_allclose(ps[0], ref, atol=2e-7, rtol=1e-5)


def test_exchange_inside_the_resident_chunk_kernel_world_one():
    """The one-shot exchange code path INSIDE the XCD-resident chunk kernel (phase E: publish the granule, poll, rank-order sum,
    Adam x 1/W) at world size 1 -- granule stores and polls on the rank's own fine-grained buffer -- gives bit for bit the weights
    of the same kernel without exchange, and reports no time-out.  (Two ranks cannot share a device with this kernel; over xGMI the
    peers' buffers are the same kind of mapping.)"""
    from ggad_amd.exchange import OneShotExchange
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.minibatch import BatchChunk, MiniBatchEngine
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule
    rowptr, col, feat, labels, train, pool, (w, W, fc) = _inputs()
    graph = DeviceGraph(rowptr, col, "cuda:0")
    ft = torch.from_numpy(feat).to("cuda:0")
    sched = BatchSchedule(train.copy(), pool.copy(), labels, 90, PyCompatRandom(72), n_pseudo=30, batches_per_epoch=7)
    bn, bl = sched.next_batches(6, 0, 1)
    res = []
    for use_x in (False, True):
        eng = MiniBatchEngine(17, 64, "cuda:0", resident=True)
        eng.load_params(w, W, fc)
        ch = BatchChunk(graph, ft, 64, max_batches=6, rows_cap=64, ent_cap=64, train=True, hop2="ldsw")
        ch.build(bn, bl)
        x = OneShotExchange(0, 1, 64 + 64 * 17 + 64 * 64, "cuda:0") if use_x else None
        eng.train_chunk(ch, exchange=x)
        torch.cuda.synchronize()
        assert eng.xcd_status()["error"] == 0
        if x is not None:
            assert x.error() == 0
        res.append((eng.params.cpu().numpy().copy(), eng.losses(6).copy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_rank_owned_streams_keep_the_ranks_in_step(tmp_path):
    """`own_stream`: every rank draws its batches from a schedule of its own (here seeded per rank) and generates only the
    batches it trains on; the averaged update is still identical on every rank."""
    import torch.multiprocessing as mp
    steps, world = 6, 2
    mp.spawn(_worker, args=(world, _free_port(), steps, str(tmp_path), False, True), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)
    assert np.isfinite(p0).all()
    assert int(np.load(tmp_path / "seen_0.npy")[0]) == steps and int(np.load(tmp_path / "seen_1.npy")[0]) == steps


def test_two_ranks_one_gpu_equal_gradient_averaging(tmp_path):
    import torch.multiprocessing as mp
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.minibatch import BatchChunk, MiniBatchEngine
    from ggad_amd.sampler import PyCompatRandom
    from ggad_amd.trainer import BatchSchedule
    steps, world = 8, 2
    mp.spawn(_worker, args=(world, _free_port(), steps, str(tmp_path)), nprocs=world, join=True)
    p0 = np.load(tmp_path / "params_0.npy")
    p1 = np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)                       # identical update on every rank
    # one process, same global batch stream: step s averages the gradients of batches 2s and 2s + 1
    rowptr, col, feat, labels, train, pool, (w, W, fc) = _inputs()
    graph = DeviceGraph(rowptr, col, "cuda:0")
    ft = torch.from_numpy(feat).to("cuda:0")
    sched = BatchSchedule(train.copy(), pool.copy(), labels, 90, PyCompatRandom(72), n_pseudo=30, batches_per_epoch=7)
    bn, bl = sched.next_batches(steps * world, 0, 1)
    eng = MiniBatchEngine(17, 64, "cuda:0")
    eng.load_params(w, W, fc)
    ch = BatchChunk(graph, ft, 64, max_batches=world, rows_cap=64, ent_cap=64, train=True, hop2="ldsw")
    for s in range(steps):
        ch.build(bn[world * s:world * s + world], bl[world * s:world * s + world])
        eng.loss_and_grads(ch, 0, 0)
        g0 = eng.grads.clone()
        eng.loss_and_grads(ch, 1, 1)
        eng.step_counter.sub_(1)                                 # two backward passes, ONE optimiser step
        eng.grads.add_(g0)                                       # what the all-reduce(SUM) leaves on every rank
        eng.adam_step(1.0 / world)
    np.testing.assert_allclose(p0, eng.params.cpu().numpy(), atol=1e-6, rtol=1e-5)


def _score_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from ggad_amd.graph import DeviceGraph
    from ggad_amd.graphsage import GCN, FeatureTable, GCNAggregator, GCNEncoder
    from ggad_amd.sage_utils import score_nodes, test_sage
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rowptr, col, feat, labels, train, pool, (w, W, fc) = _inputs()
    graph = DeviceGraph(rowptr, col, "cuda:0")
    features = FeatureTable(torch.from_numpy(feat))
    enc = GCNEncoder(features, 17, 64, graph, GCNAggregator(features, cuda=True), gcn=True, cuda=True)
    model = GCN(2, enc)
    enc.engine.load_params(w, W, fc)
    cases = np.arange(1000, 1000 + 7 * 150 + 37)                  # 8 reference batches, the last one ragged
    full = score_nodes(model, cases, 150)
    shard = score_nodes(model, cases, 150, dist=dist)
    np.save(os.path.join(out_dir, f"scores_{rank}.npy"), np.stack([full, shard]))
    y = (np.arange(len(cases)) % 7 == 0).astype(np.int64)
    res = test_sage(cases, y, model, 150, 0.4, dist=dist, verbose=False)
    np.save(os.path.join(out_dir, f"metrics_{rank}.npy"), np.asarray(res))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_validation_sweep_two_ranks_one_gpu(tmp_path):
    """The validation sweep sharded over 2 ranks (contiguous ranges of the reference's batches + one all-reduce) gives
    every rank exactly the scores and metrics of the one-process sweep."""
    import torch.multiprocessing as mp
    mp.spawn(_score_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "scores_0.npy"), np.load(tmp_path / "scores_1.npy")
    np.testing.assert_array_equal(s0[0], s0[1])
    np.testing.assert_array_equal(s1[0], s1[1])
    np.testing.assert_array_equal(s0[0], s1[0])
    np.testing.assert_array_equal(np.load(tmp_path / "metrics_0.npy"), np.load(tmp_path / "metrics_1.npy"))


def test_bench_launch_line_two_ranks_one_gpu():
    env = dict(os.environ, GGAD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "30",
           "--nodes", "200000", "--entries", "4000000", "--chunk", "30",
           "--e2e-steps", "60", "--e2e-reps", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 prints ONE json line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 60 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["value"] > 0 and np.isfinite(d["last_loss"]) and d["config"]["parallelism"] == "dp2"
    # one batch stream per rank by default; the end-to-end legs (sampler inside the window) are reported for BOTH sampler modes, so a
    # scaling run cannot hide the serial stream's ~6.4 M nodes/s ceiling of the shared mode
    assert "one per rank" in d["config"]["batch_streams"]
    e2e = d["e2e_with_sampler"]
    assert e2e["default_mode"] == "independent"
    for mode in ("independent", "shared"):
        assert e2e[mode]["value"] > 0 and e2e[mode]["sampler_us_per_batch"] > 0 and e2e[mode]["reps"] >= 1, e2e[mode]
    # the N > 1 line explains itself: which group was initialised, which exchange ran and why, every rank's own time,
    # and the same window under the contracted design (RCCL all-reduce, one shared schedule) beside the default
    mg = d["multi_gpu"]
    assert mg["world_size"] == 2 and mg["backend"] == "gloo" and mg["rccl_ranks"] == 0           # (nccl on a node: rccl_ranks == N)
    assert mg["allreduce_of_ones"] == 2.0 and mg["ranks_share_a_device"] is True
    assert mg["gradient_exchange"] == d["config"]["gradient_exchange"]
    assert (mg["gradient_exchange_fallback_reason"] is None) == mg["gradient_exchange"].startswith("oneshot")
    pr = mg["per_rank_ms_per_step"]
    assert len(pr["values"]) == 2 and 0 < pr["min"] <= pr["median"] <= pr["max"] <= d["ms_per_step"] * 1.05
    alt = mg["alt_legs"]
    assert "error" not in alt, alt
    assert alt["dp_sampler_shared"]["value"] > 0
    other = "exchange_rccl" if mg["gradient_exchange"].startswith("oneshot") else "exchange_oneshot"
    assert other in alt and (alt[other].get("value", 0) > 0 or "error" in alt[other]), alt
    vr = d["value_repeats"]
    assert vr["reps"] == 7 and len(vr["values"]) == 7 and vr["min"] <= vr["median"] <= vr["max"]


def test_bench_bare_line_spawns_two_ranks_one_gpu():
    """`python bench.py --gpus 2 ...` as typed (no torchrun, WORLD_SIZE unset): the process re-executes itself through
    torch.distributed.run, rank 0 prints the one JSON line with n_gpus 2, rc 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(GGAD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "20",
           "--nodes", "200000", "--entries", "4000000", "--chunk", "20", "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["value"] > 0 and d["config"]["parallelism"] == "dp2"
