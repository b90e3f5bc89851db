import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); n = 100_000_000
g = torch.Generator(device=dev).manual_seed(1)
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); d = torch.rand(n, dtype=torch.float64, device=dev, generator=g) + 0.5
y = torch.empty_like(q); ctx = s.context(dev); chi = s.NormLinf(1.0)
ops = {"iprox_l1box": s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj), "iprox_l0box": s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, chi), sj),
       "iprox_l1": s.shifted(s.shifted(s.NormL1(1.0), xk), sj), "iprox_l0": s.shifted(s.shifted(s.NormL0(1.0), xk), sj)}
def t(psi, iters=20):
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(iters): s.iprox_bang(y, psi, q, d, check=False)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); return ms.value / iters
res = {}
for rnd in range(5):
    for name, psi in ops.items():
        for lds in (0, 1):
            L.spx_ctx_set_tuning(s.context("cuda:0"), 3, lds)
            if rnd == 0: t(psi, 3)
            res.setdefault((name, lds), []).append(t(psi))
L.spx_ctx_set_tuning(s.context("cuda:0"), 3, 1)
for k in sorted(res):
    v = sorted(res[k]); med = v[len(v)//2]
    print("%-14s %-9s median %.4f ms -> %.0f GB/s" % (k[0], "lds" if k[1] else "registers", med, 40*n/med/1e6))
