"""Ragged consecutive groups (CSR offsets): sizes uniform in [lo, hi], ~1.28e8 elements."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
rng = np.random.default_rng(0)
for lo, hi in ((16, 64), (64, 128), (100, 500), (200, 2000)):
    sizes = rng.integers(lo, hi + 1, size=int(128_000_000 / ((lo + hi) / 2)))
    off = np.concatenate([[0], np.cumsum(sizes)]); n = int(off[-1]); ng = sizes.size
    xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
    lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
    offd = torch.from_numpy(off).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for name, binf in (("l2", False), ("binf", True)):
        for hint in (int(sizes.max()), 0):
            ts = []
            for rnd in range(3):
                ms = ctypes.c_float(); L.spx_timer_start(ctx)
                for _ in range(3):
                    if binf: L.spx_prox_group_l2_binf(ctx, p(y), p(q), p(xk), p(sj), n, p(offd), hint, ng, p(lam), 1.0, 1.0)
                    else: L.spx_prox_group_l2(ctx, p(y), p(q), p(xk), p(sj), n, p(offd), hint, ng, p(lam), 1.0)
                L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 3)
            ts.sort()
            print("sizes %4d..%4d %-4s hint %4d: %.3f ms  %.0f GB/s" % (lo, hi, name, hint, ts[1], (32 * n + 8 * ng) / ts[1] / 1e6))
    del xk, sj, q, y
