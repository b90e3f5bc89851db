"""Round 2: GroupNormL2Binf at 1e6 x 128 on lattice data with Delta ON the lattice (|x_i| == Delta in every group: the literal path)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
ng = 1_000_000; gs = 128; n = ng * gs
g = torch.Generator(device="cuda:0").manual_seed(1)
x0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); s0 = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
q0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q0)
lam = torch.rand(ng, dtype=torch.float64, device="cuda:0", generator=g) + 0.5
h = s.GroupNormL2.uniform(lam, gs)
for kind, lev in (("continuous", 0), ("lattice 1/4", 4), ("lattice 1", 1)):
    x, sj, q = (x0, s0, q0) if lev == 0 else tuple(torch.round(v * lev) / lev for v in (x0, s0, q0))
    for delta in (1.0, 0.9):
        psi = s.shifted(s.shifted(h, x, delta, s.NormLinf(1.0)), sj)
        for _ in range(2): s.prox_bang(y, psi, q, 1.0)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(3): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms)); print("%-12s Delta %-4g %9.3f ms per call" % (kind, delta, ms.value / 3), flush=True)
