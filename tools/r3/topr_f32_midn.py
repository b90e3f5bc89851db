"""Float32 top-r at 1 Mi < n <= 8 Mi: v parked in LDS (tuning key 11 = 1, default) against round 3's earlier dispatch
(registers to 2 Mi, v parked in y beyond; key 11 = 0), bit-equality checked."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
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
for nn in (1_500_000, 2_000_000, 3_000_000, 4_000_000, 8_000_000, 8_388_608):
    x = torch.randn(nn, dtype=torch.float32, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float32, device="cuda:0", generator=g) - 0.5
    q = torch.randn(nn, dtype=torch.float32, device="cuda:0", generator=g)
    for rr in (nn // 100, nn // 2):
        psi = s.shifted(s.shifted(s.IndBallL0(rr), x, 1.0, chi), sj)
        ys, ts = [], []
        for key in (1, 0):
            L.spx_ctx_set_tuning(ctx, 11, key)
            y = torch.empty_like(q)
            ts.append(timed(lambda: s.prox_bang(y, psi, q, 1.0)))
            ys.append(y.clone())
        L.spx_ctx_set_tuning(ctx, 11, 1)
        same = torch.equal(ys[0].view(torch.int32), ys[1].view(torch.int32))
        print("n=%-8d r=%-8d v in LDS %6.1f us | key 11 = 0 %6.1f us | %s" % (nn, rr, ts[0], ts[1], "bit-identical" if same else "MISMATCH"), flush=True)
