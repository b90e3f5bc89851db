"""GroupNormL2 / Binf throughput for several group sizes (same total of ~1.28e8 elements)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
total = 128_000_000
for gs in [int(v) for v in (sys.argv[1:] or ["32", "64", "100", "128", "250", "256", "512", "1000", "4096"])]:
    ng = total // gs; m = ng * gs
    xk = torch.randn(m, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(m, dtype=torch.float64, device=dev, generator=g) - 0.5
    q = torch.randn(m, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
    lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
    h = s.GroupNormL2.uniform(lam, gs)
    for name, psi in (("l2", s.shifted(s.shifted(h, xk), sj)), ("binf", s.shifted(s.shifted(h, xk, 1.0, chi), sj))):
        ts = []
        for rnd in range(4):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(5): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 5)
        ts.sort(); med = ts[len(ts) // 2]
        print("gsize %5d %-4s %.4f ms  %.0f GB/s" % (gs, name, med, (32 * m + 8 * ng) / med / 1e6))
    del xk, sj, q, y, lam
