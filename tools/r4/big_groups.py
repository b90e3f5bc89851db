"""ShiftedGroupNormL2 / ShiftedGroupNormL2Binf prox! and psi(y) on FEW, HUGE groups -- first of all the reference's default
GroupNormL2 (one group over the whole vector: shifted(NormL2(lambda), xk), src/shiftedGroupNormL2.jl:34-35).
usage: big_groups.py [n ...]   (inputs from spx_synth_fill; HIP events through spx_timer_*)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
chi = s.NormLinf(1.0)
ns = [int(float(v)) for v in (sys.argv[1:] or ["1e6", "1e7", "1e8"])]

def fill(n, stream, kind):   # spx_synth_fill: the generator bench.py uses (kind 0: U(-1/2, 1/2), 1: ~N(0, 1))
    t = torch.empty(n, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), n, 20250613, stream, kind, 1.0))
    return t

def timed(fn, iters):
    ms = ctypes.c_float(); ts = []
    for _ in range(3):
        L.spx_timer_start(ctx)
        for _ in range(iters): fn()
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / iters)
    ts.sort(); return ts[1]

for n in ns:
    xk, sj, q = fill(n, 0, 1), fill(n, 1, 0), fill(n, 2, 1)
    y = torch.empty_like(q)
    for ng in (1, 7):
        if ng == 1:
            h = s.GroupNormL2([0.5 * n ** 0.5])   # lambda sigma ~ ||S|| / 3: the group is shrunk, not zeroed
        else:
            cuts = sorted(set([0, n] + [int(n * f) for f in (0.09, 0.22, 0.31, 0.55, 0.6, 0.93)]))
            lam = [0.5 * (b - a) ** 0.5 for a, b in zip(cuts, cuts[1:])]
            h = s.GroupNormL2(lam, [range(a, b) for a, b in zip(cuts, cuts[1:])])
        for name, psi in (("l2", s.shifted(s.shifted(h, xk), sj)), ("binf", s.shifted(s.shifted(h, xk, 1.0, chi), sj))):
            iters = 3 if n >= 10**8 else 10
            t0 = time.time(); s.prox_bang(y, psi, q, 1.0); torch.cuda.synchronize(); first = time.time() - t0
            iters = 1 if first > 0.5 else iters
            ms = timed(lambda: s.prox_bang(y, psi, q, 1.0), iters)
            print("n %9d groups %d %-4s prox! %10.4f ms  %7.1f GB/s (32 B/elt)" % (n, ng, name, ms, 32 * n / ms / 1e6), flush=True)
            mo = timed(lambda: psi(y), iters)
            print("n %9d groups %d %-4s psi(y) %9.4f ms  %7.1f GB/s (24 B/elt)" % (n, ng, name, mo, 24 * n / mo / 1e6), flush=True)
    del xk, sj, q, y
