"""One-off: many small random groups (rare-event hunt) for GroupNormL2Binf vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(31)
tot_bad = 0
for gs in (1, 2, 3, 4, 8, 16, 64):
    ng = 200_000 if gs <= 16 else 40_000
    n = ng * gs
    for sigma, delta, lscale, xscale in ((1.0, 1.0, 1.0, 1.0), (0.3, 0.2, 0.1, 1.0), (2.0, 3.0, 3.0, 0.3), (1.0, 0.5, 1.0, 3.0), (1.0, 1.0, 30.0, 1.0)):
        x = rng.normal(size=n) * xscale; sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
        lam = rng.uniform(0.05, 2.0, size=ng) * lscale
        xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
        h = s.GroupNormL2.uniform(lam, gs) if False else s.GroupNormL2.uniform(lam.tolist(), gs)
        with np.errstate(all="ignore"):
            ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, gsize=gs)
        y = s.prox(s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd), qd, sigma).cpu().numpy()
        S = ((q + x) + sj).reshape(ng, gs); nS = np.linalg.norm(S, axis=1)
        sc = np.maximum(np.abs(ref).reshape(ng, gs), nS[:, None])
        canc = np.maximum(1.0, sigma * lam / np.maximum(nS, 1e-300))[:, None]
        err = (np.abs(y - ref).reshape(ng, gs) / np.maximum(sc, 1e-300) / canc).max(axis=1)
        bad = int((err > 1e-10).sum()); tot_bad += bad
        print("gs %3d sigma %g delta %g lam x%g xk x%g: worst %.2e bad groups %d / %d" % (gs, sigma, delta, lscale, xscale, float(err.max()), bad, ng))
        if bad:
            g = int(np.argmax(err)); a, b = g * gs, (g + 1) * gs
            print("   e.g. group", g, "lam", lam[g], "S", S[g], "X", x[a:b], "gpu", y[a:b], "ref", ref[a:b])
print("total bad", tot_bad)
sys.exit(1 if tot_bad else 0)
