"""Round 4 (third session): top-r at 4 Mi < n <= 6 Mi: the one-launch form with 16 elements per lane in LDS and 8 in registers
(k_sel_lds<.., REGX = 8>, tuning key 11 = 1, default) against the sample-predicted pipeline (key 11 = 2), bit for bit, and
against the exact select that parks v in y (key 2 = 0, key 11 = 2).  Aliased forms (y === q, y === xk) included."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
os.environ.setdefault("SPX_NO_BUILD", "1")
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(5)
chi = s.NormLinf(1.0)
def timed(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(5):
        ms = ctypes.c_float()
        L.spx_timer_start(ctx)
        for _ in range(reps): fn()
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        best = min(best, ms.value / reps * 1e3)
    return best
sizes = [int(float(v)) for v in os.environ.get("SPX_NS", "4194306,4500001,5000000,6000000,6291456,6291458,8000000").split(",")]
bad = 0
for nn in sizes:
    x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g)
    for kind in ("continuous", "lattice 1/4", "constant"):
        if kind == "lattice 1/4":
            q = torch.round(q * 4) / 4; x = torch.round(x * 4) / 4; sj = torch.round(sj * 4) / 4
        elif kind == "constant":
            q = torch.full_like(q, 1.5); x = torch.zeros_like(x); sj = torch.zeros_like(sj)
        for rr, binf in ((nn // 100, False), (nn // 2, False), (nn // 100, True), (nn - 7, True), (1, False)):
            psi = s.shifted(s.shifted(s.IndBallL0(rr), x, 1.0, chi), sj) if binf else s.shifted(s.shifted(s.IndBallL0(rr), x), sj)
            ys, ts = [], []
            for key in (1, 2):
                L.spx_ctx_set_tuning(ctx, 11, key)
                y = torch.full_like(q, float("nan"))
                ts.append(timed(lambda: s.prox_bang(y, psi, q, 1.0)) if kind != "constant" or rr == nn // 2 else 0.0)
                if ts[-1] == 0.0: s.prox_bang(y, psi, q, 1.0)
                ys.append(y.clone())
            L.spx_ctx_set_tuning(ctx, 2, 0)   # the exact select with v parked in y
            yex = torch.empty_like(q); s.prox_bang(yex, psi, q, 1.0)
            L.spx_ctx_set_tuning(ctx, 2, 1); L.spx_ctx_set_tuning(ctx, 11, 1)
            qa = q.clone(); s.prox_bang(qa, psi, qa, 1.0)                         # y === q
            same = (torch.equal(ys[0].view(torch.int64), ys[1].view(torch.int64)) and torch.equal(ys[0].view(torch.int64), yex.view(torch.int64))
                    and torch.equal(qa.view(torch.int64), yex.view(torch.int64)))
            bad += 0 if same else 1
            print("n=%-8d %-12s r=%-8d %-5s one launch %6.1f us | pipeline %6.1f us | %s" %
                  (nn, kind, rr, "Binf" if binf else "", ts[0], ts[1], "bit-identical (pipeline, exact select, y === q)" if same else "MISMATCH"), flush=True)
print("mismatching cases:", bad, "sync rc", L.spx_sync(ctx))
sys.exit(1 if bad else 0)
