"""ShiftedNormL1Box on 8-byte-aligned views (all four vectors start 8 bytes off a 16-byte boundary), n = 1e8"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1)
n = 100_000_000
mk = lambda: torch.randn(n + 1, dtype=torch.float64, device=dev, generator=g)[1:]
xk, sj, q, y = mk(), mk(), mk(), mk()
sj.mul_(0.3)
assert all(t.data_ptr() % 16 == 8 for t in (xk, sj, q, y))
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj)
ts = []
for rnd in range(5):
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(10): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 10)
ts.sort(); print("misaligned views: median %.4f ms -> %.0f GB/s" % (ts[2], 32 * n / ts[2] / 1e6))
# top-r on the same views: the sample-predicted path runs on the aligned rest (element 0 rides with wave 0)
r = n // 100
psi = s.shifted(s.shifted(s.IndBallL0(r), xk, 1.0, s.NormLinf(1.0)), sj)
L.spx_ctx_set_tuning(s.context("cuda:0"), 2, 0); ref = s.prox_bang(torch.empty_like(y), psi, q, 1.0).clone(); L.spx_ctx_set_tuning(s.context("cuda:0"), 2, 1)
ts = []
for rnd in range(5):
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(10): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 10)
assert torch.equal(y, ref)
ts.sort(); print("misaligned views, ShiftedIndBallL0BInf r = n/100: median %.4f ms -> %.0f GB/s" % (ts[2], 32 * n / ts[2] / 1e6))
# the other operator families on the same views
def timed(f, reps=10):
    f(); ts = []
    for rnd in range(5):
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(reps): f()
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / reps)
    return sorted(ts)[2]
ng, gs = 781_250, 128
lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
cases = {
    "ShiftedGroupNormL2 781250x128": (s.shifted(s.shifted(s.GroupNormL2.uniform(lam, gs), xk), sj), 32),
    "ShiftedGroupNormL2Binf 781250x128": (s.shifted(s.shifted(s.GroupNormL2.uniform(lam, gs), xk, 1.0, s.NormLinf(1.0)), sj), 32),
    "ShiftedNormL1B2 (active)": (s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj), 32),
    "ShiftedRootNormLhalfBox": (s.shifted(s.shifted(s.RootNormLhalf(1.0), xk, 1.0, s.NormLinf(1.0)), sj), 32),
}
for name, (psi, bpe) in cases.items():
    t = timed(lambda: s.prox_bang(y, psi, q, 1.0))
    print("misaligned views, %-36s %.4f ms -> %.0f GB/s" % (name, t, bpe * n / t / 1e6), flush=True)
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj)
t = timed(lambda: psi(y)); print("misaligned views, %-36s %.4f ms -> %.0f GB/s" % ("psi(y) ShiftedNormL1Box", t, 24 * n / t / 1e6))
t = timed(lambda: s.prox_value_bang(y, psi, q, 1.0)); print("misaligned views, %-36s %.4f ms -> %.0f GB/s" % ("prox_value ShiftedNormL1Box", t, 32 * n / t / 1e6))
d = torch.rand(n + 1, dtype=torch.float64, device=dev, generator=g)[1:] + 0.5
t = timed(lambda: s.iprox_bang(y, psi, q, d)); print("misaligned views, %-36s %.4f ms -> %.0f GB/s" % ("iprox ShiftedNormL1Box", t, 40 * n / t / 1e6))
