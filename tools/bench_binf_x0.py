"""GroupNormL2Binf at x0 = 0 inside a wide trust region (root = lmax = ||S|| in every group): must run on the fast path"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1)
ng, gs = 1_000_000, 128; n = ng * gs
xk = torch.zeros(n, dtype=torch.float64, device=dev); sj = torch.zeros(n, dtype=torch.float64, device=dev)
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
for delta in (100.0, 1.0):
    psi = s.shifted(s.shifted(s.GroupNormL2.uniform(lam, gs), xk, delta, s.NormLinf(1.0)), sj)
    ts = []
    for rnd in range(4):
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 5)
    ts.sort(); print("x0 = 0, Delta = %g: %.3f ms" % (delta, ts[1]))
