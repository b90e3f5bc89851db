"""Per-call latency at solver-iteration sizes: HIP-event time and host wall time of 200 back-to-back calls (both per call)."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
chi = s.NormLinf(1.0)
for nn in (10_000, 100_000, 1_000_000, 4_000_000):
    x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    ops = (("L1Box", s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, chi), sj)),
           ("top-r n/100", s.shifted(s.shifted(s.IndBallL0(max(1, nn // 100)), x, 1.0, chi), sj)),
           ("B2", s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, s.NormL2(1.0)), sj)))
    for name, psi in ops:
        for _ in range(20): s.prox_bang(y, psi, q, 1.0)
        torch.cuda.synchronize()
        best_ev, best_wall = 1e9, 1e9
        for rnd in range(5):
            ms = ctypes.c_float()
            t0 = time.perf_counter()
            L.spx_timer_start(ctx)
            for _ in range(200): s.prox_bang(y, psi, q, 1.0)
            t1 = time.perf_counter()
            L.spx_timer_stop(ctx, ctypes.byref(ms))
            best_ev = min(best_ev, ms.value / 200 * 1e3); best_wall = min(best_wall, (t1 - t0) / 200 * 1e6)
        print("n=%-8d %-12s %7.2f us per call by HIP events | host issue %6.2f us per call" % (nn, name, best_ev, best_wall), flush=True)
