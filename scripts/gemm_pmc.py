import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggad_amd.fullgraph import gemm
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
for _ in range(3): gemm(a, b, False, True)
torch.cuda.synchronize()
