"""Separable-skeleton A/B: register-staged (tuning key 3 = 0) vs LDS-staged (1), all separable operators,
interleaved rounds in one process (guide rule 24)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
g = torch.Generator(device=dev).manual_seed(1)
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
lv = -1.0 - 0.1 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
uv = 1.0 + 0.1 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
y = torch.empty_like(q)
ctx = s.context(dev)
chi = s.NormLinf(1.0)
ops = {"l1box": (32, s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)),
       "l1": (32, s.shifted(s.shifted(s.NormL1(1.0), xk), sj)),
       "l0box": (32, s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, chi), sj)),
       "lhalf": (32, s.shifted(s.shifted(s.RootNormLhalf(1.0), xk), sj)),
       "lhalfbox": (32, s.shifted(s.shifted(s.RootNormLhalf(1.0), xk, 1.0, chi), sj)),
       "l1box_vecbounds": (48, s.shifted(s.shifted(s.NormL1(1.0), xk, lv, uv), sj)),
       "l1box_mask": (33, s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi, range(0, n, 2)), sj))}
def t(psi, iters=20):
    ms = ctypes.c_float()
    L.spx_timer_start(ctx)
    for _ in range(iters): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms))
    return ms.value / iters
res = {}
for rnd in range(5):
    for name, (bpe, psi) in ops.items():
        for lds in (0, 1):
            L.spx_ctx_set_tuning(s.context("cuda:0"), 3, lds)
            if rnd == 0: t(psi, 3)
            res.setdefault((name, lds), []).append(t(psi))
L.spx_ctx_set_tuning(s.context("cuda:0"), 3, 1)
for k in sorted(res):
    v = sorted(res[k]); med = v[len(v)//2]
    print("%-16s %-9s median %.4f ms  min %.4f ms  -> %.0f GB/s" % (k[0], "lds" if k[1] else "registers", med, v[0], ops[k[0]][0]*n/med/1e6))
