"""Data-parallel layer on CPU: 2 gloo ranks.

The HIP kernels cannot run here, so the per-batch gradient engine is the CPU oracle; what is under test
is the host logic the GPU path uses unchanged: the rank-independent batch stream and its round-robin
dealing (`BatchSchedule.next_batches`), the single all-reduce(SUM) + 1/W contract (`reduce_gradients`)
and that every rank ends with the same weights as a one-process run that averages the W gradients."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from ggad_amd import synth
from ggad_amd.sampler import PyCompatRandom
from ggad_amd.trainer import BatchSchedule, reduce_gradients
from oracle import ggad_oracle as O


def _inputs():
    n, f, d = 3000, 17, 16
    rowptr, col = synth.make_graph(n, 20000, 4, kind="powerlaw", max_degree=60)
    feat = O.normalize_rows(synth.make_features(n, f, 4)).astype(np.float32)
    labels = np.zeros(n, dtype=np.int64)
    pool = np.arange(100, 400)
    labels[pool] = 1
    train = np.arange(400, 2400)
    torch.manual_seed(3)
    w = torch.nn.init.xavier_uniform_(torch.empty(1, d))
    W = torch.nn.init.xavier_uniform_(torch.empty(d, f))
    fc = torch.nn.init.xavier_uniform_(torch.empty(d, d))
    return rowptr, col, feat, labels, train, pool, (w, W, fc)


def _schedule(labels, train, pool):
    return BatchSchedule(train.copy(), pool.copy(), labels, batch_size=30, rng=PyCompatRandom(72), n_pseudo=10,
                         batches_per_epoch=5)


def _flat_grads(p, rowptr, col, feat, nodes, lab):
    for t in p.tensors():
        t.grad = None
    agg = O.aggregate_batch(rowptr, col, feat, nodes, True)
    O.batch_loss(p, agg, lab)[0].backward()
    return torch.cat([t.grad.reshape(-1) for t in p.tensors()])


def _apply(p, opt, flat, scale):
    off = 0
    for t in p.tensors():
        k = t.numel()
        t.grad = (flat[off:off + k] * scale).view_as(t).clone()
        off += k
    opt.step()


def _worker(rank, world, port, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    rowptr, col, feat, labels, train, pool, init = _inputs()
    sched = _schedule(labels, train, pool)
    p = O.MiniParams(*[t.clone().requires_grad_() for t in init])
    opt = O.make_adam(p.tensors(), 1e-3, 0.007)
    seen = []
    for s in range(steps):
        bn, bl = sched.next_batches(1, rank, world)
        seen.append(bn[0].copy())
        g = _flat_grads(p, rowptr, col, feat, bn[0], bl[0])
        scale = reduce_gradients(g, world, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        _apply(p, opt, g, scale)
    out[rank] = (torch.cat([t.detach().reshape(-1) for t in p.tensors()]).numpy(), np.stack(seen))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_data_parallel_matches_gradient_averaging():
    world, steps = 2, 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), steps, out), nprocs=world, join=True)
    w0, seen0 = out[0]
    w1, seen1 = out[1]
    np.testing.assert_array_equal(w0, w1)                       # identical update on every rank
    # single-process restatement: same stream, W batches per step, averaged gradients
    rowptr, col, feat, labels, train, pool, init = _inputs()
    sched = _schedule(labels, train, pool)
    p = O.MiniParams(*[t.clone().requires_grad_() for t in init])
    opt = O.make_adam(p.tensors(), 1e-3, 0.007)
    for s in range(steps):
        gs = []
        for r in range(world):
            nodes, lab = sched.next_batch()
            assert np.array_equal(nodes, (seen0, seen1)[r][s])   # rank r took batch s*W + r of the common stream
            gs.append(_flat_grads(p, rowptr, col, feat, nodes, lab))
        _apply(p, opt, gs[0] + gs[1], 1.0 / world)
    ref = torch.cat([t.detach().reshape(-1) for t in p.tensors()]).numpy()
    np.testing.assert_allclose(w0, ref, atol=1e-6, rtol=0)


def _worker_independent(rank, world, port, steps, seed, out):
    """`dp_sampler: independent` (the default at W > 1, ggad_amd/model_handler.py): rank r draws from a stream of its own, seeded
    seed * 1000003 + r + 1, and takes EVERY batch of it (`DGraphTrainer(own_stream=True)` -> next_batches(k, 0, 1))."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    rowptr, col, feat, labels, train, pool, init = _inputs()
    sched = BatchSchedule(train.copy(), pool.copy(), labels, batch_size=30, rng=PyCompatRandom(seed * 1000003 + rank + 1),
                          n_pseudo=10, batches_per_epoch=5)
    p = O.MiniParams(*[t.clone().requires_grad_() for t in init])
    opt = O.make_adam(p.tensors(), 1e-3, 0.007)
    seen = []
    for s in range(steps):
        bn, bl = sched.next_batches(1, 0, 1)
        seen.append(bn[0].copy())
        g = _flat_grads(p, rowptr, col, feat, bn[0], bl[0])
        scale = reduce_gradients(g, world, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
        _apply(p, opt, g, scale)
    out[rank] = (torch.cat([t.detach().reshape(-1) for t in p.tensors()]).numpy(), np.stack(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_independent_streams_equal_single_rank_streams_with_those_seeds():
    """The default multi-GPU sampler mode: the batches rank r sees at W = 2 are exactly the batches of a ONE-rank run seeded like rank r
    (so a rank never generates a peer's batches: the host cost per step is one batch, not W), and the weights equal the
    gradient-averaging restatement over the two streams."""
    world, steps, seed = 2, 7, 72                              # 7 steps: crosses an epoch boundary (5 batches per epoch)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_independent, args=(world, _free_port(), steps, seed, out), nprocs=world, join=True)
    (w0, seen0), (w1, seen1) = out[0], out[1]
    np.testing.assert_array_equal(w0, w1)
    rowptr, col, feat, labels, train, pool, init = _inputs()
    solo = [BatchSchedule(train.copy(), pool.copy(), labels, batch_size=30, rng=PyCompatRandom(seed * 1000003 + r + 1), n_pseudo=10,
                          batches_per_epoch=5) for r in range(world)]
    p = O.MiniParams(*[t.clone().requires_grad_() for t in init])
    opt = O.make_adam(p.tensors(), 1e-3, 0.007)
    for s in range(steps):
        gs = []
        for r in range(world):
            nodes, lab = solo[r].next_batch()                  # the single-rank stream with that seed
            assert np.array_equal(nodes, (seen0, seen1)[r][s])
            gs.append(_flat_grads(p, rowptr, col, feat, nodes, lab))
        _apply(p, opt, gs[0] + gs[1], 1.0 / world)
    ref = torch.cat([t.detach().reshape(-1) for t in p.tensors()]).numpy()
    np.testing.assert_allclose(w0, ref, atol=1e-6, rtol=0)
    assert not np.array_equal(seen0, seen1)                    # two different streams


def test_model_handler_defaults_to_independent_streams_at_world_gt_1():
    import types
    src = open(os.path.join(os.path.dirname(__file__), "..", "ggad_amd", "model_handler.py")).read()
    assert 'getattr(args, "dp_sampler", "independent")) != "shared"' in src
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "src", "dgraph.yml")))
    assert cfg["dp_sampler"] == "independent"


def test_world_one_is_the_reference_schedule():
    """W = 1: batch b of epoch e = train[b*bs:(b+1)*bs] + first n_pseudo of the freshly shuffled pool,
    driven by the python `random` stream (model_handler.py:314,333-347)."""
    import random
    _, _, _, labels, train, pool, _ = _inputs()
    sched = _schedule(labels, train, pool)
    random.seed(72)
    tr, pl = train.tolist(), pool.tolist()
    for epoch in range(2):
        random.shuffle(tr)
        for b in range(5):
            batch = tr[b * 30:(b + 1) * 30]
            random.shuffle(pl)
            batch = batch + pl[:10]
            nodes, lab = sched.next_batch()
            assert nodes.tolist() == batch
            assert lab.tolist() == labels[np.array(batch)].tolist()
    assert reduce_gradients(torch.zeros(3), 1, None) == 1.0
    with pytest.raises(ValueError):
        reduce_gradients(torch.zeros(3), 2, None)


def test_bench_bare_gpus_n_respawn_command_is_well_formed():
    """`python bench.py --gpus N` without WORLD_SIZE re-executes itself through torch.distributed.run: the command is the
    driver's own launch line (one rank per GPU, rendezvous on 127.0.0.1) with the user's argv passed through untouched."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ggad_bench_cli", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.respawn_command(8, argv, port=29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv
    # torch.distributed.run accepts that line (argument parsing only, nothing is launched)
    from torch.distributed.run import get_args_parser
    ns = get_args_parser().parse_args(cmd[3:])
    assert ns.nproc_per_node == "8" and ns.master_addr == "127.0.0.1" and ns.training_script.endswith("bench.py")
    assert ns.training_script_args == argv
    # a free port is picked when none is given
    p = int(bench.respawn_command(2, [])[bench.respawn_command(2, []).index("--master-port") + 1])
    assert 1024 < p < 65536
    # without a GPU the bare line still gets as far as the launcher and every rank reports the missing GPU (no "must be launched with")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras"], env=env,
                             capture_output=True, text=True, timeout=300)
        assert "needs a GPU" in out.stderr and "must be launched" not in out.stderr
