"""Cost of the dense step chain (5 dependent launches per optimiser step) by KIND of HIP stream, alone on the chip, and of
the plan by kind of stream: null stream / torch pool stream (non-blocking) / CU-masked stream with every CU / with 64 CUs.
Usage (GPU box): python scripts/stream_kind_probe.py"""
import ctypes
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from ggad_amd import _lib, synth  # noqa: E402
from ggad_amd.dgraph import normalize_features, split_dgraphfin  # noqa: E402
from ggad_amd.graph import DeviceGraph  # noqa: E402
from ggad_amd.sampler import PyCompatRandom  # noqa: E402
from ggad_amd.trainer import BatchSchedule, DGraphTrainer  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)).to(dev)
lab = synth.make_labels(n, 15509.0 / 3700550.0, 72).astype(np.int32)
sp = split_dgraphfin(lab, 72, with_test=False)
sched = BatchSchedule(sp['idx_train'], sp['idx_anomaly'], sp['labels'], 150, PyCompatRandom.from_python_state(random.getstate()))
tr = DGraphTrainer(g, feat, 64, sched, overlap=False)
torch.manual_seed(0)
tr.engine.load_params(torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)),
                      torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
lib = _lib.load()


def masked(n_cu):
    words = 8
    mask = (ctypes.c_uint32 * words)()
    for cu in range(n_cu):
        mask[cu // 32] |= 1 << (cu % 32)
    h = ctypes.c_void_p()
    _lib.check(lib.ggad_stream_create_cu_mask(ctypes.cast(mask, ctypes.c_void_p), words, ctypes.byref(h)), "mask")
    return torch.cuda.ExternalStream(h.value, device=dev)


streams = {"null": None, "torch pool (non-blocking)": torch.cuda.Stream(device=dev), "masked, 256 CUs": masked(256),
           "masked, 64 CUs": masked(64)}
for k in (20, 150):
    bn, bl = sched.next_batches(k)
    tr.chunk.build(bn, bl)
    torch.cuda.synchronize()
    for name, st in streams.items():
        res_t, res_b = [], []
        for rep in range(4):
            ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
            with ctx:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tr.engine.train_chunk(tr.chunk)
                torch.cuda.synchronize()
                res_t.append(time.perf_counter() - t0)
                t0 = time.perf_counter()
                tr.chunk.build(bn, bl)
                torch.cuda.synchronize()
                res_b.append(time.perf_counter() - t0)
        print(f"{k:4d} batches | {name:28s} | dense chain {1e6 * np.mean(res_t[1:]) / k:6.1f} us/step | plan {1e3 * np.mean(res_b[1:]):.3f} ms",
              flush=True)

# fresh batches every time, as the bench does: plan + dense chain back to back without a sync in between
for k in (20,):
    tot, sep_b, sep_t = [], [], []
    for rep in range(6):
        bn, bl = sched.next_batches(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.chunk.build(bn, bl)
        tr.engine.train_chunk(tr.chunk)
        torch.cuda.synchronize()
        tot.append(time.perf_counter() - t0)
        bn, bl = sched.next_batches(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.chunk.build(bn, bl)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.engine.train_chunk(tr.chunk)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        sep_b.append(t1 - t0); sep_t.append(t2 - t1)
    print(f"{k} fresh batches: plan + dense back to back {1e3 * np.mean(tot[1:]):.3f} ms | with a sync in between: plan {1e3 * np.mean(sep_b[1:]):.3f} ms + "
          f"dense {1e3 * np.mean(sep_t[1:]):.3f} ms ({1e6 * np.mean(sep_t[1:]) / k:.1f} us/step)")
