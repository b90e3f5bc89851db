"""Why does the headline kernel time vary 0.50-0.53 ms between tensors / moments on one box?"""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); dev = torch.device("cuda:0"); ctx = s.context(dev)
n = 100_000_000
def t(psi, y, q, iters=20, rounds=5):
    out = []
    for r in range(rounds):
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(iters): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms)); out.append(ms.value / iters)
    return min(out), sorted(out)[len(out)//2]
sets = []
for k in range(4):
    gen = torch.Generator(device=dev).manual_seed(20250613 + k)
    xk = torch.randn(n, dtype=torch.float64, device=dev, generator=gen); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=gen) - 0.5
    q = torch.randn(n, dtype=torch.float64, device=dev, generator=gen); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj)
    sets.append((psi, y, q, [hex(v.data_ptr()) for v in (xk, sj, q, y)]))
    print("set", k, "allocated: first timing", "min %.4f med %.4f ms" % t(psi, y, q), sets[-1][3], flush=True)
for rep in range(3):
    for k, (psi, y, q, _) in enumerate(sets):
        print("rep", rep, "set", k, "min %.4f med %.4f ms" % t(psi, y, q), flush=True)
