"""Round 4: per-call latency of EVERY call of a solver iteration at solver sizes (n = 1e4 ... 4e6): prox! of each operator
family, psi(y) into a device value, prox + value, iprox!, the group operators (plain / Binf; groups of 8, 100, one group).
HIP-event time and host issue time of 200 back-to-back calls, best of 5, both per call.  SPX_NS overrides the sizes."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
os.environ.setdefault("SPX_NO_BUILD", "1")
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
dev = torch.device("cuda:0")
SEED = 20250613 + 1000
chi = s.NormLinf(1.0)
sizes = [int(float(a)) for a in os.environ.get("SPX_NS", "1e4,1e5,1e6,4e6").split(",")]


def synth(m, stream, kind):
    t = torch.empty(m, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), m, SEED, stream, kind, ctypes.c_double(1.0)))
    return t


def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best_ev, best_wall = 1e9, 1e9
    for rnd in range(5):
        ms = ctypes.c_float()
        t0 = time.perf_counter()
        L.spx_timer_start(ctx)
        for _ in range(reps): fn()
        t1 = time.perf_counter()
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        best_ev = min(best_ev, ms.value / reps * 1e3); best_wall = min(best_wall, (t1 - t0) / reps * 1e6)
    return best_ev, best_wall


print("us per call by HIP events (host issue us)   " + "".join("%18d" % n for n in sizes), flush=True)
rows = {}
for nn in sizes:
    x, sj, q = synth(nn, 0, 1), synth(nn, 1, 0), synth(nn, 2, 1)
    y = torch.empty_like(q); d = synth(nn, 3, 0) + 1.0
    val = torch.zeros(1, dtype=torch.float64, device=dev)
    def grp(gs):
        ng = nn // gs
        lam = synth(ng, 4, 0) + 1.0
        return s.GroupNormL2.uniform(lam, gs), ng * gs
    cases = []
    box = s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, chi), sj)
    cases.append(("prox! L1Box", lambda psi=box: s.prox_bang(y, psi, q, 1.0)))
    l0 = s.shifted(s.shifted(s.NormL0(1.0), x, 1.0, chi), sj)
    cases.append(("prox! L0Box", lambda psi=l0: s.prox_bang(y, psi, q, 1.0)))
    lh = s.shifted(s.shifted(s.RootNormLhalf(1.0), x, 1.0, chi), sj)
    cases.append(("prox! LhalfBox", lambda psi=lh: s.prox_bang(y, psi, q, 1.0)))
    def obj(psi):
        def f():
            with s.device_values(val): psi(y)
        return f
    cases.append(("psi(y) L1Box -> device", obj(box)))
    def pv(psi):
        def f():
            with s.device_values(val): s.prox_value_bang(y, psi, q, 1.0)
        return f
    cases.append(("prox+value L1Box -> device", pv(box)))
    cases.append(("iprox! L1Box", lambda psi=box: s.iprox_bang(y, psi, q, d)))
    tr = s.shifted(s.shifted(s.IndBallL0(max(1, nn // 100)), x, 1.0, chi), sj)
    cases.append(("prox! top-r n/100 Binf", lambda psi=tr: s.prox_bang(y, psi, q, 1.0)))
    b2 = s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, s.NormL2(1.0)), sj)
    cases.append(("prox! L1B2", lambda psi=b2: s.prox_bang(y, psi, q, 1.0)))
    for gs in (8, 100):
        H, m = grp(gs)
        pl = s.shifted(s.shifted(H, x[:m]), sj[:m]); bi = s.shifted(s.shifted(H, x[:m], 1.0, chi), sj[:m])
        cases.append(("prox! GroupL2 of %d" % gs, lambda psi=pl, m=m: s.prox_bang(y[:m], psi, q[:m], 1.0)))
        cases.append(("prox! GroupL2Binf of %d" % gs, lambda psi=bi, m=m: s.prox_bang(y[:m], psi, q[:m], 1.0)))
        def gobj(psi, m):
            def f():
                with s.device_values(val): psi(y[:m])
            return f
        cases.append(("psi(y) GroupL2Binf of %d -> device" % gs, gobj(bi, m)))
    one = s.shifted(s.shifted(s.NormL2(1.0), x), sj); oneb = s.shifted(s.shifted(s.NormL2(1.0), x, 1.0, chi), sj)
    cases.append(("prox! NormL2 (one group)", lambda psi=one: s.prox_bang(y, psi, q, 1.0)))
    cases.append(("prox! NormL2 + Binf (one group)", lambda psi=oneb: s.prox_bang(y, psi, q, 1.0)))
    cases.append(("psi(y) NormL2 + Binf -> device", obj(oneb)))
    for name, fn in cases:
        ev, wall = timed(fn)
        rows.setdefault(name, []).append("%8.1f (%5.1f)" % (ev, wall))
    del x, sj, q, y, d
for name, cols in rows.items():
    print("%-44s" % name + "".join("%18s" % c for c in cols), flush=True)
