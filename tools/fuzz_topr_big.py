"""One-off stress of the top-r paths at n just above the fast-path threshold with awkward distributions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
n = (1 << 22) + 4321
rng = np.random.default_rng(7)
def dists():
    yield "cauchy", rng.standard_cauchy(n)
    yield "loguniform_300", np.sign(rng.normal(size=n)) * 10.0 ** rng.uniform(-300, 300, size=n)
    yield "constant_outliers", np.where(rng.random(n) < 1e-5, 1e9, 1.0) * np.sign(rng.normal(size=n))
    yield "descending", -np.sort(-np.abs(rng.normal(size=n)))
    yield "zeros", np.zeros(n)
    yield "denormals", rng.integers(0, 1 << 20, size=n).astype(np.float64) * 5e-324
    yield "two_clusters", np.where(rng.random(n) < 0.5, 1.0, 1.0 + 1e-15) * np.sign(rng.normal(size=n))
    yield "blocks", np.repeat(rng.normal(size=n // 1024 + 1), 1024)[:n]
bad = 0
for name, q in dists():
    x = np.zeros(n); sj = np.zeros(n)
    if name in ("cauchy", "blocks"):
        x = rng.normal(size=n) * 0.1; sj = rng.uniform(-0.05, 0.05, size=n)
    xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
    for r in (1, 100, n // 1000, n // 10, n // 2, n - 1):
        with np.errstate(all="ignore"):
            ref = orc.prox_indball_l0_binf(q, x, sj, r, 0.7)
        y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.7, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
        ok = np.array_equal(y.view(np.int64), ref.view(np.int64))
        bad += not ok
        print("%-18s r=%-8d %s" % (name, r, "ok" if ok else "MISMATCH"))
print("mismatches", bad)
sys.exit(1 if bad else 0)
