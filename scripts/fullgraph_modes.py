"""Eager epochs against replayed hipGraph epochs of the full-graph path, same process:  python scripts/fullgraph_modes.py [name]"""
import gc
import random
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ggad_amd import fullgraph as FG  # noqa: E402
from ggad_amd.fullgraph_bench import build_model, make_dataset  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "t_finance"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
torch.set_num_threads(8)
random.seed(0); np.random.seed(0)
ds = make_dataset(name, 0)
full, model, opt, feats = build_model(ds, dev, 300, 0)
args = types.SimpleNamespace(mean=ds["mean"], var=ds["var"])
abn, nrm = ds["abn_idx"], ds["normal_idx"]
ls = full.loss_structs(nrm, abn)


def train_epoch():
    opt.zero_grad()
    emb, emb_combine, logits, emb_con, emb_abnormal = model(feats, full, abn, nrm, True, args)
    out = FG.GgadLossFn.apply(emb[0], logits[0, :, 0], emb_con, emb_abnormal[0], full, ls, 0.7)
    out[0].backward()
    opt.step()
    return out


def med(fn, n=20):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return np.median(ts) * 1e3, np.min(ts) * 1e3


def gpu_ms(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


model.train()
for _ in range(3):
    train_epoch()
print(name, "eager   median/min ms", med(train_epoch), "back-to-back", gpu_ms(train_epoch))
noise_buf = torch.zeros(1, len(abn), 300, device=dev)
model.noise_override = noise_buf
opt.zero_grad(); gc.collect(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static = train_epoch()
model.noise_override = None
print(name, "replay  median/min ms", med(g.replay), "back-to-back", gpu_ms(g.replay))
print(name, "eager again", med(train_epoch), "back-to-back", gpu_ms(train_epoch))
