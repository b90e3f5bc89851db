"""One-off: top-r on 8-byte-misaligned views (scalar full-vector path) at small and large n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(4)
bad = 0
for n in (1, 2, 3, 1000, 70001, (1 << 22) + 17):
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = np.round(rng.normal(size=n) * 16) / 16
    mk = lambda a: torch.cat([torch.zeros(1, dtype=torch.float64), torch.from_numpy(a)]).cuda()[1:]
    for mis in ((True, True, True), (True, False, False), (False, False, True)):
        xd = mk(x) if mis[0] else torch.from_numpy(x).cuda()
        sd = mk(sj) if mis[1] else torch.from_numpy(sj).cuda()
        qd = mk(q) if mis[2] else torch.from_numpy(q).cuda()
        for r in sorted({1, max(1, n // 3), n}):
            ref = orc.prox_indball_l0_binf(q, x, sj, r, 0.8)
            psi = s.shifted(s.shifted(s.IndBallL0(r), xd, 0.8, s.NormLinf(1.0)), sd)
            y = s.prox(psi, qd, 1.0).cpu().numpy()
            yv = mk(np.zeros(n)); s.prox_bang(yv, psi, qd, 1.0)
            ok = np.array_equal(y.view(np.int64), ref.view(np.int64)) and np.array_equal(yv.cpu().numpy().view(np.int64), ref.view(np.int64))
            if not ok: bad += 1; print("MISMATCH n %d mis %s r %d" % (n, mis, r))
print("mismatches", bad)
sys.exit(1 if bad else 0)
