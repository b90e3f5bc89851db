import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build()
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(99)
n = 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)
for _ in range(5): s.prox_bang(y, psi, q, 1.0)
torch.cuda.synchronize()
