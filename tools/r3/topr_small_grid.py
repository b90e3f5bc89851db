"""One-launch register-resident top-r (8192 < n <= 2 Mi): time per call against the number of workgroups (elements per lane)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
chi = s.NormLinf(1.0)
for nn in (10_000, 30_000, 100_000, 300_000, 1_000_000, 2_000_000):
    x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    for rr in (max(1, nn // 100), nn // 2):
        psi = s.shifted(s.shifted(s.IndBallL0(rr), x, 1.0, chi), sj)
        for _ in range(20): s.prox_bang(y, psi, q, 1.0)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(5):
            ms = ctypes.c_float()
            L.spx_timer_start(ctx)
            for _ in range(200): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms))
            best = min(best, ms.value / 200 * 1e3)
        print("n=%-8d r=%-8d %7.2f us per call" % (nn, rr, best), flush=True)
