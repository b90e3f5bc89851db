"""Kernel tuning sweep for the separable skeleton (interleaved rounds in one process, guide rule 24)."""
import ctypes, sys, os, json
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
y = torch.empty_like(q)
ctx = s.context(dev)
ops = {"l1box": s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj),
       "l1": s.shifted(s.shifted(s.NormL1(1.0), xk), sj),
       "l0box": s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, s.NormLinf(1.0)), sj)}
def t(psi, iters=20):
    ms = ctypes.c_float()
    L.spx_timer_start(ctx)
    for _ in range(iters): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms))
    return ms.value / iters
configs = [(b, nt) for b in (16, 64, 128, 256, 0) for nt in (0, 1)]
res = {}
for rnd in range(5):
    for name, psi in ops.items():
        for b, nt in configs:
            L.spx_set_tuning(0, b); L.spx_set_tuning(1, nt)
            if rnd == 0: t(psi, 3)
            res.setdefault((name, b, nt), []).append(t(psi))
for k in sorted(res):
    v = sorted(res[k]); med = v[len(v)//2]
    print("%-6s blocks/CU=%2d nt=%d  median %.4f ms  min %.4f ms  -> %.0f GB/s (median)" % (k[0], k[1], k[2], med, v[0], 32*n/med/1e6))
