"""One-off stress of GroupNormL2Binf on lattice data: S, X on multiples of 1/4 so that |tau S - X| = Delta boundaries,
equal elements, zero groups and exact roots occur constantly; every group kernel family."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(11)
worst = 0.0; nbad = 0
for gs in (3, 16, 32, 100, 128, 256, 300, 1024):
    for rep in range(6):
        ng = 200 if gs <= 300 else 30
        n = ng * gs
        x = rng.integers(-8, 9, size=n) / 4.0
        sj = rng.integers(-2, 3, size=n) / 4.0
        q = rng.integers(-12, 13, size=n) / 4.0
        if rep % 3 == 1: x[: n // 2] = 0.0
        if rep % 3 == 2: q[:] = np.repeat(rng.integers(-4, 5, size=ng) / 4.0, gs)   # constant groups
        lam = rng.choice([0.0, 0.25, 0.5, 1.0, 2.0, 8.0], size=ng)
        sigma = float(rng.choice([0.25, 0.5, 1.0, 2.0])); delta = float(rng.choice([0.25, 0.5, 1.0, 3.0]))
        xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
        h = s.GroupNormL2.uniform(lam.tolist(), gs)
        with np.errstate(all="ignore"):
            ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, gsize=gs)
        y = s.prox(s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd), qd, sigma).cpu().numpy()
        S = ((q + x) + sj).reshape(ng, gs)
        sc = np.maximum(np.abs(ref).reshape(ng, gs), np.linalg.norm(S, axis=1, keepdims=True))
        canc = np.maximum(1.0, sigma * lam / np.maximum(np.linalg.norm(S, axis=1), 1e-300))[:, None]
        err = np.abs(y - ref).reshape(ng, gs) / np.maximum(sc, 1e-300) / canc
        e = float(np.nanmax(err)); worst = max(worst, e)
        nanbad = not np.array_equal(np.isnan(y), np.isnan(ref))
        if e > 1e-11 or nanbad:
            nbad += 1
            g = int(np.nanargmax(err.max(axis=1)))
            print("gs %d rep %d: err %.2e (group %d lam %g sigma %g delta %g) nan_mismatch=%s" % (gs, rep, e, g, lam[g], sigma, delta, nanbad))
print("worst scaled error %.2e, failing configs %d" % (worst, nbad))
sys.exit(1 if nbad else 0)
