"""Top-r at 2 Mi < n <= 4 Mi: the LDS-resident one-launch select (tuning key 11 = 1, default) against the sample-predicted
pipeline (key 11 = 0).  Bit-equality of the two results is checked on every case."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(5)
chi = s.NormLinf(1.0)
def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(5):
        ms = ctypes.c_float()
        L.spx_timer_start(ctx)
        for _ in range(reps): fn()
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        best = min(best, ms.value / reps * 1e3)
    return best
for nn in (2_200_000, 3_000_000, 4_000_000, 4_194_304):
    x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g)
    for kind in ("continuous", "lattice 1/4"):
        if kind != "continuous":
            q = torch.round(q * 4) / 4; x = torch.round(x * 4) / 4; sj = torch.round(sj * 4) / 4
        for rr, binf in ((nn // 100, False), (nn // 2, False), (nn // 100, True)):
            psi = s.shifted(s.shifted(s.IndBallL0(rr), x, 1.0, chi), sj) if binf else s.shifted(s.shifted(s.IndBallL0(rr), x), sj)
            ys, ts = [], []
            for key in (1, 0):
                L.spx_ctx_set_tuning(ctx, 11, key)
                y = torch.empty_like(q)
                ts.append(timed(lambda: s.prox_bang(y, psi, q, 1.0)))
                ys.append(y.clone())
            L.spx_ctx_set_tuning(ctx, 11, 1)
            same = torch.equal(ys[0].view(torch.int64), ys[1].view(torch.int64))
            print("n=%-8d %-12s r=%-8d %-5s one launch (LDS) %6.1f us | pipeline %6.1f us | %s" %
                  (nn, kind, rr, "Binf" if binf else "", ts[0], ts[1], "bit-identical" if same else "MISMATCH"), flush=True)
