"""RootNormLhalf(Box) over lambda*sigma and the bounds at n = 1e8: the share of elements that evaluate the stationary point
(candidate 4) depends on them; ms per call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev); g = torch.Generator(device=dev).manual_seed(1)
n = 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
def timed(f):
    f(); ts = []
    for _ in range(5):
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(10): f()
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 10)
    return sorted(ts)[2]
for lam in (1e-4, 0.01, 0.3, 1.0, 3.0, 100.0):
    psi = s.shifted(s.shifted(s.RootNormLhalf(lam), xk), sj)
    t0 = timed(lambda: s.prox_bang(y, psi, q, 1.0))
    row = "lambda*sigma = %-7g RootNormLhalf %.4f ms |" % (lam, t0)
    for delta in (0.1, 1.0, 1e3):
        psi = s.shifted(s.shifted(s.RootNormLhalf(lam), xk, delta, s.NormLinf(1.0)), sj)
        row += "  Box Delta=%-5g %.4f ms" % (delta, timed(lambda: s.prox_bang(y, psi, q, 1.0)))
    print(row, flush=True)
