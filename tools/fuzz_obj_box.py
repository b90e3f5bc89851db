"""One-off: psi(y) of the Box forms with vector bounds, masks, boundary-exact feasibility, vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(23)
SQ = 1.4901161193847656e-08
bad = 0
dev = lambda *a: [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in a]
for t in range(400):
    n = int(rng.integers(1, 5000))
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n)
    lo = rng.normal(size=n) - 1.0; up = lo + np.abs(rng.normal(size=n)) + 0.1
    y = lo + (up - lo) * rng.random(n) - sj                                   # feasible
    mode = t % 5
    i = int(rng.integers(0, n))
    if mode == 1: y[i] = up[i] - sj[i] + SQ                                   # on the slack
    elif mode == 2: y[i] = (up[i] + SQ) * (1 + 1e-15) - sj[i] if up[i] + SQ > 0 else (up[i] + SQ) * (1 - 1e-15) - sj[i]
    elif mode == 3: y[i] = lo[i] - sj[i] - 2 * SQ                             # infeasible
    elif mode == 4: y[i] = np.nan
    selected = sorted(rng.choice(n, size=int(rng.integers(1, n + 1)), replace=False).tolist())
    mask = orc.mask_from_selected([k + 1 for k in selected], n)
    xd, sd, yd, ld, ud = dev(x, sj, y, lo, up)
    for kind, H in (("l1", s.NormL1), ("l0", s.NormL0), ("lhalf", s.RootNormLhalf)):
        for use_mask in (False, True):
            psi = s.shifted(s.shifted(H(0.9), xd, ld, ud, selected), sd) if use_mask else s.shifted(s.shifted(H(0.9), xd, ld, ud), sd)
            with np.errstate(all="ignore"):
                a = psi(yd); b = orc.obj_box(kind, y, x, sj, 0.9, lo, up, mask=mask if use_mask else None)
            ok = (a == b) or (np.isnan(a) and np.isnan(b)) or (np.isfinite(a) and np.isfinite(b) and abs(a - b) <= 1e-12 * max(abs(a), abs(b)))
            if not ok: bad += 1; print("t %d mode %d %s mask %s: gpu %r oracle %r" % (t, mode, kind, use_mask, a, b))
print("mismatches", bad)
sys.exit(1 if bad else 0)
