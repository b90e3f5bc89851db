"""Round 2: ShiftedNormL1B2 at n = 1e8 over input seeds -- does the pass count (visible in the time) depend on the draw?"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = int(os.environ.get("SPX_N", "100000000"))
for seed in [int(v) for v in os.environ.get("SPX_SEEDS", "1,2,3,99,20250613").split(",")]:
    for lam, delta in ((1.0, 1.0), (0.3, 1.0), (1.0, 100.0)):
        g = torch.Generator(device="cuda:0").manual_seed(seed)
        xk = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
        q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
        psi = s.shifted(s.shifted(s.NormL1(lam), xk, delta, s.NormL2(1.0)), sj)
        for _ in range(3): s.prox_bang(y, psi, q, 1.0)
        ts = []
        for rnd in range(3):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(5): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 5 * 1e3)
        print("seed %9d lambda %-4g Delta %-5g: %8.1f us  (min %.1f max %.1f)" % (seed, lam, delta, sorted(ts)[1], min(ts), max(ts)), flush=True)
        del xk, sj, q, y, psi
