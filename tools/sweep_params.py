"""Cliff hunt: ms per call of GroupNormL2Binf (1e6 x 128) and NormL1B2 (n = 1e8) over trust-region radius, lambda*sigma and
data kind.  The kernels' work per group / number of reduction passes depends on the data; nothing here should be far
from the figures bench.py reports for its one configuration."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1)
gs = int(os.environ.get("SPX_GS", "128")); ng = 128_000_000 // gs; n = ng * gs   # SPX_GS: other uniform group sizes
def timed(f, reps=5):
    f(); ts = []
    for _ in range(3):
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(reps): f()
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / reps)
    return sorted(ts)[1]
x0 = torch.randn(n, dtype=torch.float64, device=dev, generator=g); s0 = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q0 = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q0)
lam1 = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
which = sys.argv[1] if len(sys.argv) > 1 else "all"
only = sys.argv[2] if len(sys.argv) > 2 else None
for kind in ("normal", "x=0,s=0", "lattice8", "q*100", "q*0.01", "x*0.05", "90% zero groups"):
    if only and only not in kind: continue
    if kind == "normal": xk, sj, q = x0, s0, q0
    elif kind == "x=0,s=0": xk, sj, q = torch.zeros_like(x0), torch.zeros_like(x0), q0
    elif kind == "lattice8": xk, sj, q = (torch.round(v * 8) / 8 for v in (x0, s0, q0))
    elif kind == "q*100": xk, sj, q = x0, s0, q0 * 100
    elif kind == "q*0.01": xk, sj, q = x0, s0, q0 * 0.01
    elif kind == "x*0.05": xk, sj, q = x0 * 0.05, s0, q0
    else:
        keep = (torch.rand(ng, device=dev, generator=g) < 0.1).to(torch.float64).repeat_interleave(gs)
        xk, sj, q = x0 * keep, s0, q0
    if which in ("all", "binf"):
        for delta in (0.01, 0.25, 1.0, 4.0, 100.0):
            for lscale in (0.01, 1.0, 30.0):
                psi = s.shifted(s.shifted(s.GroupNormL2.uniform(lam1 * lscale, gs), xk, delta, s.NormLinf(1.0)), sj)
                print("GroupL2Binf %-9s Delta=%-6g lambda~%-5g %.3f ms" % (kind, delta, lscale, timed(lambda: s.prox_bang(y, psi, q, 1.0))), flush=True)
    if which in ("all", "b2"):
        for delta in (1e-3, 1.0, 1e3, 1.3e4, 1e6):
            for lam in (0.01, 1.0, 30.0):
                psi = s.shifted(s.shifted(s.NormL1(lam), xk, delta, s.NormL2(1.0)), sj)
                print("NormL1B2    %-9s Delta=%-6g lambda=%-5g %.3f ms" % (kind, delta, lam, timed(lambda: s.prox_bang(y, psi, q, 1.0))), flush=True)
