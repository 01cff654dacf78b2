import sys, numpy as np, torch
sys.path.insert(0, '.')
from ggad_amd import synth
dev = torch.device('cuda:0')
n, ne = 3_700_550, 73_105_508
rp, ci = synth.make_graph_torch(n, ne, 72, dev, max_degree=2000)
deg = np.diff(rp)
print('deg mean', deg.mean(), 'max', deg.max(), 'mean nbr deg', (deg.astype(float)**2).sum()/deg.sum())
rng = np.random.default_rng(0)
for t in range(3):
    nodes = rng.choice(n, 200, replace=False)
    U = np.unique(np.concatenate([ci[rp[v]:rp[v+1]] for v in nodes] + [nodes.astype(np.int32)]))
    pairs = np.concatenate([ci[rp[u]:rp[u+1]] for u in U])
    k, c = np.unique(pairs, return_counts=True)
    cc = c[np.searchsorted(k, pairs)]
    print('U', len(U), 'S2', len(pairs), 'distinct k', len(k), 'frac pairs with c>1', (cc > 1).mean(), 'max c', c.max(),
          'frac pairs c>255', (cc > 255).mean(), 'hub share of S2 (deg u >500)', deg[U][deg[U] > 500].sum() / len(pairs))
