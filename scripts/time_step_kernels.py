"""Micro-timing of the per-step kernels on a bench-like batch (GPU)."""
import sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from ggad_amd import synth, _lib
from ggad_amd._lib import call, ptr
from ggad_amd.dgraph import normalize_features
from ggad_amd.graph import DeviceGraph
from ggad_amd.minibatch import BatchChunk, MiniBatchEngine

dev = torch.device('cuda:0')
n, ne = 1_000_000, 20_000_000
rp, ci = synth.make_graph_torch(n, ne, 1, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 1)).astype(np.float32)).to(dev)
rng = np.random.default_rng(0)
nb = 8
batches = [rng.choice(n, 200, replace=False) for _ in range(nb)]
labels = []
for b in range(nb):
    l = np.zeros(200, dtype=np.int64); l[150:] = 1; l[rng.choice(150, 1)] = 1; labels.append(l)
ch = BatchChunk(g, feat, 64, nb, 2000, 100000, True)
eng = MiniBatchEngine(17, 64, dev)
torch.manual_seed(0)
eng.load_params(torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)),
                torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
ch.build(batches, labels)
eng.ensure_capacity(ch, 16)
torch.cuda.synchronize()
print('ents per batch', [ch.batch_ents(b)[1] - ch.batch_ents(b)[0] for b in range(nb)], 'max row', int(np.diff(ch.ent_ptr_host).max()))

def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

b = 0
r0, r1 = ch.batch_rows(b); e0_, e1_ = ch.batch_ents(b)
D, F = 64, 17
s = eng.step_desc(ch, b, 0)
lib = _lib.load(); st = _lib.current_stream()
def k_project(): call("ggad_mb_project", ptr(eng.params), D, F, ptr(ch.x2), ptr(ch.ent_own), e0_, e1_ - e0_, ptr(eng.h2))
def k_fwd(): call("ggad_mb_fwd_rows", ptr(eng.params), D, F, ptr(ch.x1), ptr(eng.h2), ptr(ch.ent_ptr), ptr(ch.ent_own), ptr(ch.labels), r0, r1 - r0, e0_, ptr(ch.h1), ptr(ch.nbar), ptr(ch.gen))
def k_loss():
    call("ggad_mb_loss", ptr(eng.params), D, F, ptr(ch.h1), ptr(ch.nbar), ptr(ch.gen), ptr(ch.labels), ptr(ch.pos_meta), ptr(ch.row_pos), ptr(ch.ent_ptr), r0, r1 - r0,
         ptr(eng.loss_ws), eng.loss_log.data_ptr(), 0, 0, 0, ptr(ch.dz), ptr(ch.coef_a), ptr(ch.coef_g), 0)
def k_bwd(): call("ggad_mb_bwd_flat", D, F, ptr(ch.x1), ptr(ch.x2), ptr(eng.h2), ptr(ch.ent_own), ptr(ch.ent_row), r0, r1 - r0, e0_, e1_ - e0_, ptr(ch.coef_a), ptr(ch.coef_g), ptr(eng.dw_part))
def k_red(): call("ggad_mb_grad_reduce", D, F, ptr(ch.pos_meta), r0, r1 - r0, eng.loss_log.data_ptr(), ptr(ch.nbar), ptr(eng.dw_part), ptr(ch.dz), ptr(eng.loss_ws), ptr(eng.grads))
def k_step(): lib.ggad_mb_train_step(ctypes.byref(s), 1, st)
def k_empty(): call("ggad_mb_params_sync", ptr(eng.params), D, F)
k_project(); k_fwd(); k_loss(); k_bwd(); k_red()
for name, fn in [('empty(params_sync)', k_empty), ('project', k_project), ('fwd_rows', k_fwd), ('loss (2 launches)', k_loss),
                 ('bwd_flat', k_bwd), ('grad_reduce', k_red), ('train_step(6 kernels)', k_step)]:
    print(f'{name:28s} {timeit(fn):8.2f} us')
# host-side cost of the python call itself
t = time.perf_counter()
for _ in range(2000): k_step()
torch.cuda.synchronize()
print('python+launch wall per step', (time.perf_counter() - t) / 2000 * 1e6, 'us')
