"""Uniform large groups: the one-workgroup-per-group kernels (k_group_mem<256> / k_group_lds) against the team form with teams of
one workgroup that take several groups in turn (tuning key 16 = groups per workgroup up to which the team form is used).
Total 1e8 elements."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev); chi = s.NormLinf(1.0)
def synth(n, stream, kind):
    t = torch.empty(n, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), n, 20250613 + 1000, stream, kind, 1.0)); return t
def timed(fn, iters=5):
    ms = ctypes.c_float(); ts = []
    for _ in range(3):
        L.spx_timer_start(ctx)
        for _ in range(iters): fn()
        L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / iters)
    ts.sort(); return ts[1]
total = 100_000_000
xk, sj, q = synth(total, 0, 1), synth(total, 1, 0), synth(total, 2, 1); y = torch.empty_like(q); y2 = torch.empty_like(q)
for gs in [int(v) for v in (sys.argv[1:] or ["5000", "10000", "20000", "50000", "100000", "200000", "400000"])]:
    ng = total // gs; m = ng * gs
    lam = synth(ng, 3, 0) * 0.0 + 0.4 * gs ** 0.5
    h = s.GroupNormL2.uniform(lam, gs)
    for name, psi in (("l2", s.shifted(s.shifted(h, xk[:m]), sj[:m])), ("binf", s.shifted(s.shifted(h, xk[:m], 1.0, chi), sj[:m]))):
        out = []
        for factor, yy in ((1, y), (1000000, y2)):
            L.spx_ctx_set_tuning(ctx, 16, factor)
            s.prox_bang(yy[:m], psi, q[:m], 1.0)
            out.append(timed(lambda: s.prox_bang(yy[:m], psi, q[:m], 1.0)))
        L.spx_ctx_set_tuning(ctx, 16, 0)
        d = float((y[:m] - y2[:m]).abs().max())
        print("gsize %7d x %6d groups %-4s one workgroup per group %8.3f ms   team form %8.3f ms   max |dy| %.1e" % (gs, ng, name, out[0], out[1], d), flush=True)
