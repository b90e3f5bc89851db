"""Times one operator at full size: tools/bench_op.py binf|group|indball|lhalfbox|l1box [iters]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
which = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
n = 100_000_000
def vecs(m):
    return (torch.randn(m, dtype=torch.float64, device=dev, generator=g), torch.rand(m, dtype=torch.float64, device=dev, generator=g) - 0.5,
            torch.randn(m, dtype=torch.float64, device=dev, generator=g))
if which in ("binf", "group"):
    ng = 1_000_000; m = ng * 128; xk, sj, q = vecs(m)
    lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
    h = s.GroupNormL2.uniform(lam, 128)
    psi = s.shifted(s.shifted(h, xk, 1.0, chi), sj) if which == "binf" else s.shifted(s.shifted(h, xk), sj)
    bytes_ = (32 + 8 / 128) * m
else:
    xk, sj, q = vecs(n); bytes_ = 32 * n
    psi = {"indball": lambda: s.shifted(s.shifted(s.IndBallL0(n // 100), xk, 1.0, chi), sj),
           "lhalfbox": lambda: s.shifted(s.shifted(s.RootNormLhalf(1.0), xk, 1.0, chi), sj),
           "l1box": lambda: s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)}[which]()
y = torch.empty_like(q)
if len(sys.argv) > 3 and sys.argv[3] == "obj":  # psi(y): synchronous, host wall time per call
    import time
    y.copy_(q * 0.01); psi(y)
    ts = []
    for rnd in range(5):
        t0 = time.perf_counter()
        for _ in range(iters): psi(y)
        ts.append((time.perf_counter() - t0) / iters * 1e3)
    ts.sort(); print("%s objective: median %.4f ms (wall, incl. read-back) -> %.0f GB/s on 24 B/element" % (which, ts[2], 24 * y.numel() / ts[2] / 1e6))
    sys.exit(0)
ts = []
for rnd in range(5):
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(iters): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / iters)
ts.sort()
print("%s: median %.4f ms min %.4f ms -> %.0f GB/s (%.1f%% of 8 TB/s)" % (which, ts[2], ts[0], bytes_ / ts[2] / 1e6, bytes_ / ts[2] / 8e7))
