"""Ragged groups (CSR offsets, sizes 64..192) on a sparse iterate under a strong lambda, Binf: ms per call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev); g = torch.Generator(device=dev).manual_seed(5)
rng = np.random.default_rng(5)
ng = 1_000_000
sizes = rng.integers(64, 193, size=ng); offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64); n = int(offs[-1])
lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
xk0 = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
keep = torch.from_numpy(np.repeat(rng.random(ng) < 0.1, sizes).astype(np.float64)).to(dev)
for kind, xk in (("dense iterate", xk0), ("90 % zero groups", xk0 * keep), ("small iterate x*0.05", xk0 * 0.05)):
    for ls in (1.0, 30.0):
        h = s.GroupNormL2.ragged(lam * ls, offs)
        psi = s.shifted(s.shifted(h, xk, 1.0, s.NormLinf(1.0)), sj)
        s.prox_bang(y, psi, q, 1.0); ts = []
        for _ in range(3):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(5): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 5)
        t = sorted(ts)[1]
        print("ragged 64..192, %-22s lambda x %-4g %.3f ms -> %.0f GB/s" % (kind, ls, t, 32 * n / t / 1e6), flush=True)
