"""One-off rare-event hunt for the sample-predicted top-r path: random seeds, r, tie levels at n just above 2^22."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
bad = 0
for seed in range(24):
    rng = np.random.default_rng(5000 + seed)
    n = (1 << 22) + int(rng.integers(0, 100000))
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n) * 10.0 ** rng.uniform(-1, 1)
    lev = int(rng.choice([0, 0, 4, 64, 4096]))
    if lev: x, sj, q = (np.round(v * lev) / lev for v in (x, sj, q))
    xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
    for r in sorted({1, int(rng.integers(1, 100)), int(rng.integers(1, n // 100)), int(rng.integers(1, n)), n - int(rng.integers(1, 50))}):
        ref = orc.prox_indball_l0_binf(q, x, sj, r, 0.9)
        y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.9, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
        ok = np.array_equal(y.view(np.int64), ref.view(np.int64))
        q2 = qd.clone(); s.prox_bang(q2, s.shifted(s.shifted(s.IndBallL0(r), xd, 0.9, s.NormLinf(1.0)), sd), q2, 1.0)   # aliased form
        ok2 = np.array_equal(q2.cpu().numpy().view(np.int64), ref.view(np.int64))
        if not (ok and ok2): bad += 1; print("MISMATCH seed %d n %d r %d lev %d disjoint_ok %s aliased_ok %s" % (seed, n, r, lev, ok, ok2))
    print("seed", seed, "done", flush=True)
print("mismatches", bad)
sys.exit(1 if bad else 0)
