"""Round 2: ShiftedIndBallL0BInf on the sample-predicted pipeline (n > 2^21): time per call for the library named by SPX_LIB_NAME."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
for n in (3_000_000, 10_000_000, 100_000_000):
    row = []
    g = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    for r in (1000, n // 100, n // 2):
        psi = s.shifted(s.shifted(s.IndBallL0(r), x, 1.0, s.NormLinf(1.0)), sj)
        for _ in range(3): s.prox_bang(y, psi, q, 1.0)
        ts = []
        for rnd in range(5):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(20): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 20 * 1e3)
        row.append(sorted(ts)[2])
    print("n %9d: r=1000 %7.1f us   r=n/100 %7.1f us   r=n/2 %7.1f us" % (n, row[0], row[1], row[2]), flush=True)
    del x, sj, q, y
