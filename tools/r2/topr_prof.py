import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
for n in [int(a) for a in sys.argv[1:]]:
    g = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.IndBallL0(max(1, n // 100)), x, 1.0, s.NormLinf(1.0)), sj)
    for _ in range(10): s.prox_bang(y, psi, q, 1.0)
    torch.cuda.synchronize()
