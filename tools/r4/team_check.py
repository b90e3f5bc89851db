"""Team-of-workgroups group form (spx_group_team.hip) against the oracle, over sizes that hit each of its forms
(on chip / streamed, one team / several, ragged plan) -- a development check; the driver-run tests are in tests/."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
from oracle import oracle
dev = torch.device("cuda:0"); ctx = s.context(dev)
chi = s.NormLinf(1.0)
rng = np.random.default_rng(7)
bad = 0

def bar(y, ref, xs, S, offs):
    """|dy_i| <= 1e-12 max(|y_i|, |xk_i + sj_i|, ||S_group||)"""
    worst = 0.0
    for a, b in zip(offs[:-1], offs[1:]):
        scale = np.maximum(np.maximum(np.abs(ref[a:b]), np.abs(xs[a:b])), np.linalg.norm(S[a:b]))
        worst = max(worst, float(np.max(np.abs(y[a:b] - ref[a:b]) / np.maximum(scale, 1e-300))) if b > a else 0.0)
    return worst

def case(n, cuts, lamscale, delta, tag, fast=1, x0=False, offset=0):
    global bad
    offs = [0] + [int(n * f) for f in cuts] + [n]
    xk = rng.standard_normal(n + offset)[offset:] * (0.0 if x0 else 1.0); sj = rng.random(n + offset)[offset:] - 0.5; q = rng.standard_normal(n + offset)[offset:]
    lam = np.array([lamscale * (b - a) ** 0.5 for a, b in zip(offs[:-1], offs[1:])])
    big = torch.empty(n + offset, dtype=torch.float64, device=dev)
    def d(a):
        t = torch.empty(n + offset, dtype=torch.float64, device=dev); t[offset:] = torch.from_numpy(np.ascontiguousarray(a)).to(dev); return t[offset:]
    xd, sd, qd = d(xk), d(sj), d(q)
    yd = torch.empty(n + offset, dtype=torch.float64, device=dev)[offset:]
    groups = [range(a, b) for a, b in zip(offs[:-1], offs[1:])]
    h = s.GroupNormL2(lam, groups)
    L.spx_ctx_set_tuning(ctx, 14, fast)
    for name, psi, ref in (("l2", s.shifted(s.shifted(h, xd), sd), lambda: oracle.prox_group_l2(q, xk, sj, lam, 0.7, offsets=np.array(offs, dtype=np.int64))),
                           ("binf", s.shifted(s.shifted(h, xd, delta, chi), sd), lambda: oracle.prox_group_l2_binf(q, xk, sj, lam, 0.7, delta, offsets=np.array(offs, dtype=np.int64)))):
        s.prox_bang(yd, psi, qd, 0.7); torch.cuda.synchronize()
        y = yd.cpu().numpy()
        r = ref()
        S = (q + xk) + sj
        w = bar(y, r, xk + sj, S, offs)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(5): s.prox_bang(yd, psi, qd, 0.7)
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        ok = w <= 1e-12
        # psi(y) at the prox and at a point outside the trust region
        for yy in (yd, yd * 1.7):
            v = psi(yy)
            vr = oracle.obj_group_l2(yy.cpu().numpy(), xk, sj, lam, offsets=np.array(offs, dtype=np.int64), delta=(delta if name == "binf" else None))
            okv = (v == vr) or abs(v - vr) <= 1e-12 * abs(vr)
            if not okv: print("   psi(y) %r vs oracle %r" % (v, vr))
            ok = ok and okv
        bad += 0 if ok else 1
        print("%-28s n %9d groups %d %-4s fast %d  worst %.2e  %s  %8.1f us  nnz(y+xs) %d" % (tag, n, len(groups), name, fast, w, "ok" if ok else "FAIL", ms.value / 5 * 1e3, int(np.count_nonzero(y + (xk + sj)))), flush=True)
    L.spx_ctx_set_tuning(ctx, 14, 1)

sizes = [int(float(v)) for v in (sys.argv[1:] or ["5000", "20000", "300000", "2000001", "3000000", "20000000"])]
for n in sizes:
    case(n, [], 0.5, 1.0, "one group")
    case(n, [], 0.5, 1.0, "one group, generic", fast=0)
    case(n, [], 0.5, 0.05, "one group, small Delta")
    case(n, [], 5.0, 1.0, "one group, strong lambda")
    case(n, [], 0.5, 1.0, "one group, x0 = 0", x0=True)
    case(n, [], 0.5, 1.0, "one group, odd start", offset=1)
    if n >= 200000:
        case(n, (0.09, 0.22, 0.31, 0.55, 0.6, 0.93), 0.5, 1.0, "7 ragged")
        case(n, (0.09, 0.22, 0.31, 0.55, 0.6, 0.93), 0.5, 1.0, "7 ragged, generic", fast=0)
        case(n, (0.001, 0.0011, 0.31, 0.310001, 0.6, 0.93), 0.5, 1.0, "ragged, small + large")
print("FAILURES: %d" % bad)
sys.exit(1 if bad else 0)
