"""One-off: structured scenarios for ShiftedNormL1B2 vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(9)
nbad = 0; worst = 0.0
for n in (1, 2, 5, 64, 1000, 100_003):
    for k in range(9):
        x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
        if k == 0: x[:] = 0.0
        elif k == 1: q[:] = 0.0; sj[:] = 0.0
        elif k == 2: x *= 1e-8
        elif k == 3: x *= 1e8
        elif k == 4: x = np.round(x * 4) / 4; q = np.round(q * 4) / 4; sj = np.round(sj * 4) / 4
        elif k == 5: x[:] = 1.0; q[:] = -0.5; sj[:] = 0.25
        elif k == 6: x[::2] = 0.0
        elif k == 7: q = -sj + rng.choice([-1.0, 1.0], size=n) * 0.5
        elif k == 8: x = rng.standard_cauchy(n)
        for lam, sigma, delta, chil in ((1.0, 1.0, 1.0, 1.0), (0.01, 1.0, 0.1, 1.0), (5.0, 2.0, 1e-3, 0.5), (0.5, 0.3, 1e3, 2.0), (1.0, 1.0, 1e-12, 1.0)):
            xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
            with np.errstate(all="ignore"):
                ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, chil)
            y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(chil)), sd), qd, sigma).cpu().numpy()
            scale = max(np.linalg.norm(ref), np.linalg.norm(x), np.linalg.norm(sj + q), 1e-300)
            e = float(np.max(np.abs(y - ref))) / scale if np.all(np.isfinite(ref)) else (0.0 if np.array_equal(np.isnan(y), np.isnan(ref)) else 1.0)
            worst = max(worst, e)
            if e > 1e-12:
                nbad += 1
                print("n %d scen %d lam %g sigma %g delta %g chi %g: err %.2e  |ref| %.3g |y| %.3g" % (n, k, lam, sigma, delta, chil, e, np.linalg.norm(ref), np.linalg.norm(y)))
print("worst %.2e failing %d" % (worst, nbad))
sys.exit(1 if nbad else 0)
