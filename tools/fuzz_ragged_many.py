"""One-off rare-event hunt: ragged groups (CSR offsets + size bound) with many small / mid groups, plain and Binf."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(66)
tot = 0
for maxsize, ng in ((4, 150_000), (9, 100_000), (33, 50_000), (130, 20_000), (600, 4_000), (2500, 600)):
    for sigma, delta, lscale in ((1.0, 1.0, 1.0), (0.3, 5.0, 0.2), (3.0, 0.3, 3.0)):
        sizes = rng.integers(1, maxsize + 1, size=ng); off = np.concatenate([[0], np.cumsum(sizes)]); n = int(off[-1])
        x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
        lam = rng.uniform(0.05, 2.0, size=ng) * lscale
        xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
        groups = [range(int(a), int(b)) for a, b in zip(off[:-1], off[1:])]
        h = s.GroupNormL2.ragged(torch.from_numpy(lam).cuda(), off)
        S = (q + x) + sj
        nS = np.sqrt(np.add.reduceat(S * S, off[:-1]))
        for binf in (False, True):
            with np.errstate(all="ignore"):
                ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, offsets=off) if binf else orc.prox_group_l2(q, x, sj, lam, sigma, offsets=off)
            psi = s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd) if binf else s.shifted(s.shifted(h, xd), sd)
            y = s.prox(psi, qd, sigma).cpu().numpy()
            scg = np.repeat(nS, sizes); cg = np.repeat(np.maximum(1.0, sigma * lam / np.maximum(nS, 1e-300)), sizes)
            err = np.abs(y - ref) / np.maximum(np.maximum(np.abs(ref), scg), 1e-300) / cg
            bad = int((err > 1e-9).sum()); tot += bad
            print("max size %4d %-5s sigma %g delta %g: worst %.2e bad %d" % (maxsize, "binf" if binf else "plain", sigma, delta, float(err.max()), bad), flush=True)
print("total bad", tot)
sys.exit(1 if tot else 0)
