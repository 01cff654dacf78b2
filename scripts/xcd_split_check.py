import sys, random, numpy as np, torch, time
sys.path.insert(0,'.')
from ggad_amd import synth
from ggad_amd.dgraph import normalize_features, split_dgraphfin
from ggad_amd.graph import DeviceGraph
from ggad_amd.sampler import PyCompatRandom
from ggad_amd.trainer import BatchSchedule, DGraphTrainer
dev=torch.device('cuda:0'); torch.cuda.set_device(dev)
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
g = DeviceGraph(rp, ci, dev)
feat = torch.from_numpy(normalize_features(synth.make_features(n, 17, 72)).astype(np.float32)).to(dev)
lab = synth.make_labels(n, 15509.0/3700550.0, 72).astype(np.int32)
sp = split_dgraphfin(lab, 72, with_test=False)
sched = BatchSchedule(sp['idx_train'], sp['idx_anomaly'], sp['labels'], 150, PyCompatRandom.from_python_state(random.getstate()))
tr = DGraphTrainer(g, feat, 64, sched)
print('split', tr.resident_split, 'skip', getattr(tr,'_xcd_skip',None), 'first now', tr.plan_stream_first_xcd())
batch = sched.next_batches(1500)
tr.run_steps(1500, prepared=batch); torch.cuda.synchronize()
print('first after run', tr.plan_stream_first_xcd(), [tr.plan_stream_first_xcd() for _ in range(3)])
for rep in range(2):
    batch = sched.next_batches(3000)
    torch.cuda.synchronize(); t0=time.perf_counter(); nn = tr.run_steps(3000, prepared=batch); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print('steady', nn/dt/1e6, 'M nodes/s', 1e6*dt/3000, 'us/step', tr.engine.xcd_status())
