import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from ggad_amd import synth
from ggad_amd.dgraph import normalize_features, split_dgraphfin
from ggad_amd.graph import DeviceGraph
from ggad_amd.sampler import PyCompatRandom
from ggad_amd.trainer import BatchSchedule, DGraphTrainer
import random
dev = torch.device('cuda:0')
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)).to(dev)
lab = synth.make_labels(n, 15509.0 / 3700550.0, 72).astype(np.int32)
sp = split_dgraphfin(lab, 72, with_test=False)
sched = BatchSchedule(sp['idx_train'], sp['idx_anomaly'], sp['labels'], 150, PyCompatRandom.from_python_state(random.getstate()))
tr = DGraphTrainer(g, feat, 64, sched, overlap=False)
torch.manual_seed(0)
tr.engine.load_params(torch.nn.init.xavier_uniform_(torch.empty(1, 64)), torch.nn.init.xavier_uniform_(torch.empty(64, 17)), torch.nn.init.xavier_uniform_(torch.empty(64, 64)))
t = time.perf_counter(); bn, bl = sched.next_batches(150); print('sampler 150 batches host ms', (time.perf_counter() - t) * 1e3)
tr.chunk.build(bn, bl); tr.engine.train_chunk(tr.chunk); torch.cuda.synchronize()
for rep in range(3):
    bn, bl = sched.next_batches(150)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); tr.chunk.build(bn, bl); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tr.engine.train_chunk(tr.chunk); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'build host {1e3*(t1-t0):.2f} ms, build gpu-complete {1e3*(t2-t0):.2f} ms | train_chunk host {1e3*(t3-t2):.2f} ms, complete {1e3*(t4-t2):.2f} ms')
