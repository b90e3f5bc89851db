"""iprox! (ShiftedNormL1Box / L0Box) over the sign pattern of d at n = 1e8 (the three branches d > eps, d < -eps, |d| <= eps
diverge inside a wavefront when the signs are mixed); ms per call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev); g = torch.Generator(device=dev).manual_seed(1)
n = 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
gq = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(gq)
mag = torch.rand(n, dtype=torch.float64, device=dev, generator=g) + 0.5
sgn = torch.randint(0, 3, (n,), device=dev, generator=g).to(torch.float64) - 1.0     # -1, 0, +1
def timed(f):
    f(); ts = []
    for _ in range(5):
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(10): f()
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 10)
    return sorted(ts)[2]
for name, d in (("d > 0", mag), ("d < 0", -mag), ("d = 0", torch.zeros_like(mag)), ("mixed -/0/+ per element", mag * sgn)):
    row = "%-26s" % name
    for H, tag in ((s.NormL1, "L1Box"), (s.NormL0, "L0Box")):
        psi = s.shifted(s.shifted(H(1.0), xk, 1.0, s.NormLinf(1.0)), sj)
        row += "  %s %.4f ms" % (tag, timed(lambda: s.iprox_bang(y, psi, gq, d, check=False)))
    print(row, flush=True)
