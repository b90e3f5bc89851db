"""SPX_LIB_NAME=libspx_b2dbg.so (built with -DSPX_B2_DEBUG): the iteration log of k_b2_coop at small n."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); ctx = s.context("cuda:0")
for n in (10_000, 1_000_000):
    g = torch.Generator(device="cuda:0").manual_seed(99)
    xk = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)
    print("---- n", n, flush=True)
    s.prox_bang(y, psi, q, 1.0); torch.cuda.synchronize()
