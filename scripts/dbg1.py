import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from ggad_amd.graph import DeviceGraph
from ggad_amd.minibatch import BatchChunk
from oracle import ggad_oracle as O
g = load_golden('minibatch_small.npz')
graph = DeviceGraph(g['rowptr'], g['col'], 'cuda:0')
feat = torch.from_numpy(np.ascontiguousarray(g['feat'])).cuda()
ch = BatchChunk(graph, feat, 64, 8, 64, 64, True)
batches = [b for b in g['batches']]; labels=[l for l in g['labels']]
ch.build(batches, labels); torch.cuda.synchronize()
F=17
x1 = ch.x1[:ch.n_rows*F].view(-1,F).cpu().numpy()
ent_ptr = ch.ent_ptr[:ch.n_rows+1].cpu().numpy()
for b in range(2):
    r0,r1 = ch.batch_rows(b)
    agg = O.aggregate_batch(g['rowptr'], g['col'], g['feat'], batches[b], True)
    err = np.abs(x1[r0:r1]-agg.to_feats).max(1)
    bad = np.nonzero(err>1e-5)[0]
    print('batch',b,'bad rows',bad, 'r', agg.r[bad], 'err', err[bad])
    print(' all r', agg.r)
    for i in bad[:3]:
        print('  row',i,'got',x1[r0+i][:6],'want',agg.to_feats[i][:6], 'ratio', x1[r0+i][:6]/agg.to_feats[i][:6])
