"""Every GEMM of one full-graph training epoch with its shape and its time alone (HIP events around each call):
python scripts/gemm_shapes_epoch.py [reddit ...]"""
import random
import sys
import types

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import fullgraph as FG  # noqa: E402
from ggad_amd.fullgraph_bench import build_model, make_dataset  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for name in (sys.argv[1:] or ["reddit"]):
    random.seed(0); np.random.seed(0)
    ds = make_dataset(name, 0)
    full, model, opt, feats = build_model(ds, dev, 300, 0)
    args = types.SimpleNamespace(mean=ds["mean"], var=ds["var"])
    abn, nrm = ds["abn_idx"], ds["normal_idx"]
    ls = full.loss_structs(nrm, abn)
    log = []
    real = FG.gemm

    def timed(A, B, trans_a, trans_b, bias=None, relu=False, out=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        C = real(A, B, trans_a, trans_b, bias, relu, out)
        e1.record()
        log.append((tuple(A.shape), tuple(B.shape), trans_a, trans_b, e0, e1))
        return C

    def epoch():
        opt.zero_grad()
        emb, emb_combine, logits, emb_con, emb_abnormal = model(feats, full, abn, nrm, True, args)
        out = FG.GgadLossFn.apply(emb[0], logits[0, :, 0], emb_con, emb_abnormal[0], full, ls, 0.7)
        out[0].backward()
        opt.step()

    model.train()
    for _ in range(3):
        epoch()
    FG.gemm = timed
    tot = {}
    for _ in range(5):
        log.clear()
        epoch()
        torch.cuda.synchronize()
        for i, (a, b, ta, tb, e0, e1) in enumerate(log):
            tot.setdefault(i, []).append(e0.elapsed_time(e1) * 1e3)
    FG.gemm = real
    s = 0.0
    for i, (a, b, ta, tb, _, _) in enumerate(log):
        M, K = (a[1], a[0]) if ta else a
        N = b[0] if tb else b[1]
        us = float(np.median(tot[i]))
        s += us
        print("%s gemm %2d: M %6d N %4d K %6d  %s%s  %.1f us  %.1f TF" % (name, i, M, N, K, "T" if ta else "N", "T" if tb else "N", us, 2.0 * M * N * K / us / 1e6))
    print(name, "sum %.1f us over %d calls" % (s, len(log)))
