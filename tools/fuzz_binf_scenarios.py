"""One-off: structured scenarios for GroupNormL2Binf vs the oracle (cancellation-aware tolerance), all kernel families."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(21)
nbad = 0; worst = 0.0
def scen(k, ng, gs):
    n = ng * gs
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    if k == 0: x[:] = 0.0
    elif k == 1: x *= 1e-3
    elif k == 2: q = -sj.copy()                       # S = X
    elif k == 3: q = -(x + sj)                        # S = 0
    elif k == 4: q *= 1e-6; sj *= 1e-6; x *= 1e-6     # tiny against lambda
    elif k == 5: q *= 1e6                             # huge against lambda, Delta
    elif k == 6: x = np.sign(x) * 1.0                 # |X| = Delta exactly (Delta = 1 below)
    elif k == 7: q[:] = 0.25; x[:] = 0.5; sj[:] = 0.0 # constant groups
    elif k == 8: x[::2] = 0.0; q[1::2] = 0.0
    elif k == 9:
        x = rng.standard_cauchy(n); q = rng.standard_cauchy(n)
    return x, sj, q
for gs in (2, 8, 16, 40, 128, 200, 512, 700, 3000):
    ng = 120 if gs <= 512 else 20
    for k in range(10):
        for sigma, delta in ((1.0, 1.0), (0.01, 1.0), (30.0, 0.1), (1.0, 50.0)):
            x, sj, q = scen(k, ng, gs)
            lam = 10.0 ** rng.uniform(-3, 2, size=ng)
            xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
            h = s.GroupNormL2.uniform(lam.tolist(), gs)
            with np.errstate(all="ignore"):
                ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, gsize=gs)
            y = s.prox(s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd), qd, sigma).cpu().numpy()
            S = ((q + x) + sj).reshape(ng, gs); nS = np.linalg.norm(S, axis=1)
            sc = np.maximum(np.abs(ref).reshape(ng, gs), nS[:, None])
            canc = np.maximum(1.0, sigma * lam / np.maximum(nS, 1e-300))[:, None]
            err = np.abs(y - ref).reshape(ng, gs) / np.maximum(sc, 1e-300) / canc
            e = float(np.nanmax(err)) if np.isfinite(err).any() else 0.0
            worst = max(worst, e)
            if e > 1e-10 or not np.array_equal(np.isnan(y), np.isnan(ref)):
                nbad += 1
                g = int(np.nanargmax(np.nanmax(err, axis=1)))
                print("gs %d scen %d sigma %g delta %g: err %.2e group %d lam %.3g nS %.3g nX %.3g" % (gs, k, sigma, delta, e, g, lam[g], nS[g], np.linalg.norm(x.reshape(ng, gs)[g])))
print("worst %.2e failing %d" % (worst, nbad))
sys.exit(1 if nbad else 0)
