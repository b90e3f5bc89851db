"""ShiftedNormL1B2 at n = 1e8: trust region active (scaled branch) and inactive"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build()
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(99)
n = 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
for delta in (1.0, 1e9):
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, delta, s.NormL2(1.0)), sj)
    s.prox_bang(y, psi, q, 1.0); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
    s.synchronize()
    print("Delta = %g: %.3f ms per call" % (delta, (time.perf_counter() - t0) / 5 * 1e3))
