"""One-off rare-event hunt for GroupNormL2Binf on mid-size groups over a parameter grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(55)
tot = 0
for gs in (24, 32, 64, 100, 128, 256, 300):
    ng = 20_000; n = ng * gs
    for sigma, delta, lscale, xscale, qscale in ((1, 1, 1, 1, 1), (1, 100, 1, 1, 1), (1, 0.01, 1, 1, 1), (0.1, 1, 10, 1, 1), (10, 1, 0.1, 1, 1),
                                                 (1, 1, 1, 0.01, 1), (1, 1, 1, 1, 0.01), (1, 3, 20, 1, 3), (1, 1, 0.001, 1, 1)):
        x = rng.normal(size=n) * xscale; sj = rng.uniform(-0.5, 0.5, size=n) * min(1.0, xscale + qscale); q = rng.normal(size=n) * qscale
        lam = rng.uniform(0.05, 2.0, size=ng) * lscale
        xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
        h = s.GroupNormL2.uniform(torch.from_numpy(lam).cuda(), gs)
        with np.errstate(all="ignore"):
            ref = orc.prox_group_l2_binf(q, x, sj, lam, float(sigma), float(delta), gsize=gs)
        y = s.prox(s.shifted(s.shifted(h, xd, float(delta), s.NormLinf(1.0)), sd), qd, float(sigma)).cpu().numpy()
        S = ((q + x) + sj).reshape(ng, gs); nS = np.linalg.norm(S, axis=1)
        sc = np.maximum(np.abs(ref).reshape(ng, gs), nS[:, None])
        canc = np.maximum(1.0, sigma * lam / np.maximum(nS, 1e-300))[:, None]
        err = (np.abs(y - ref).reshape(ng, gs) / np.maximum(sc, 1e-300) / canc).max(axis=1)
        bad = int((err > 1e-9).sum()); tot += bad
        print("gs %3d sigma %g delta %g lam x%g xk x%g q x%g: worst %.2e bad %d" % (gs, sigma, delta, lscale, xscale, qscale, float(err.max()), bad), flush=True)
print("total bad", tot)
sys.exit(1 if tot else 0)
