"""One-off rare-event hunt over many random small instances: B2, top-r, gather groups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(77)
dev = lambda *a: [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in a]
bits = lambda a, b: np.array_equal(a.view(np.int64), b.view(np.int64))
bad = 0; worst = 0.0
for t in range(3000):
    n = int(rng.integers(1, 60))
    x = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 2); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n) * 10.0 ** rng.uniform(-1, 1)
    lam, sigma, delta, chil = 10.0 ** rng.uniform(-2, 1), 10.0 ** rng.uniform(-1, 1), 10.0 ** rng.uniform(-2, 2), 10.0 ** rng.uniform(-0.5, 0.5)
    xd, sd, qd = dev(x, sj, q)
    ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, chil)
    y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(chil)), sd), qd, sigma).cpu().numpy()
    e = float(np.max(np.abs(y - ref))) / max(np.linalg.norm(ref), np.linalg.norm(x), np.linalg.norm(sj + q), 1e-300)
    worst = max(worst, e)
    if e > 1e-12: bad += 1; print("B2 t %d n %d err %.2e lam %g sigma %g delta %g chi %g" % (t, n, e, lam, sigma, delta, chil))
print("B2: worst %.2e bad %d" % (worst, bad))
bad2 = 0
for t in range(1500):
    n = int(rng.integers(1, 4000))
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    if t % 2: q = np.round(q * 4) / 4; x = np.round(x * 4) / 4; sj = np.round(sj * 4) / 4
    r = int(rng.integers(1, n + 1)); delta = float(rng.choice([0.25, 1.0, 10.0]))
    xd, sd, qd = dev(x, sj, q)
    y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, delta, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
    if not bits(y, orc.prox_indball_l0_binf(q, x, sj, r, delta)): bad2 += 1; print("top-r mismatch t %d n %d r %d" % (t, n, r))
print("top-r: bad %d" % bad2)
bad3 = 0; worst3 = 0.0
for t in range(300):
    n = int(rng.integers(2, 3000)); ngr = int(rng.integers(1, 40))
    groups = [rng.choice(n, size=int(rng.integers(1, min(n, 200) + 1)), replace=False).tolist() for _ in range(ngr)]
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n); y0 = rng.normal(size=n)
    lam = rng.uniform(0.05, 2.0, size=ngr); sigma = 10.0 ** rng.uniform(-1, 1); delta = 10.0 ** rng.uniform(-1, 1)
    xd, sd, qd = dev(x, sj, q)
    h = s.GroupNormL2(lam.tolist(), groups)
    for binf in (False, True):
        psi = s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd) if binf else s.shifted(s.shifted(h, xd), sd)
        ref = orc.prox_group_l2_idx(q, x, sj, lam, sigma, groups, delta=delta if binf else None, y0=y0)
        y = s.prox_bang(dev(y0)[0], psi, qd, sigma).cpu().numpy()
        S = (q + x) + sj
        sc = np.abs(ref).copy(); canc = np.ones(n)
        for g, idx in enumerate(groups):
            nS = np.linalg.norm(S[idx]); sc[idx] = np.maximum(sc[idx], nS); canc[idx] = np.maximum(canc[idx], sigma * lam[g] / max(nS, 1e-300))
        e = float(np.max(np.abs(y - ref) / np.maximum(sc, 1e-300) / canc)); worst3 = max(worst3, e)
        if e > 1e-9: bad3 += 1; print("gather t %d binf %s err %.2e" % (t, binf, e))
print("gather: worst %.2e bad %d" % (worst3, bad3))
sys.exit(1 if (bad or bad2 or bad3) else 0)
