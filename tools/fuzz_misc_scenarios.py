"""One-off: scenarios for psi(y) (all forms), unboxed iprox!, plain GroupNormL2 and prox_value vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(17)
nbad = 0
dev = lambda *a: [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in a]
close = lambda a, b: (a == b) or (np.isfinite(a) and np.isfinite(b) and abs(a - b) <= 1e-12 * max(abs(a), abs(b)))
SQ = 1.4901161193847656e-08
for n in (1, 7, 1000, 65537):
    for k in range(8):
        x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); y = rng.normal(size=n) * 0.3
        lo, up = -1.0, 1.0
        if k == 1: y = (up - sj) + SQ                      # exactly on the feasibility slack
        elif k == 2: y = (up - sj) + SQ * (1 + 4e-16)     # one ulp outside
        elif k == 3: y = (lo - sj) - SQ
        elif k == 4: y = -(x + sj)                         # v = 0 everywhere
        elif k == 5: y[:] = 0.0; x *= 1e-300
        elif k == 6: y[n // 2] = np.nan
        elif k == 7: x *= 1e150
        xd, sd, yd = dev(x, sj, y)
        lam = 0.7
        for kind, H in (("l1", s.NormL1), ("l0", s.NormL0), ("lhalf", s.RootNormLhalf)):
            with np.errstate(all="ignore"):
                a = s.shifted(s.shifted(H(lam), xd), sd)(yd); b = orc.obj_plain(kind, y, x, sj, lam)
                if not (close(a, b) or (np.isnan(a) and np.isnan(b))): nbad += 1; print("obj plain", kind, n, k, a, b)
                a = s.shifted(s.shifted(H(lam), xd, lo, up), sd)(yd); b = orc.obj_box(kind, y, x, sj, lam, lo, up)
                if not (close(a, b) or (np.isnan(a) and np.isnan(b))): nbad += 1; print("obj box", kind, n, k, a, b)
        with np.errstate(all="ignore"):
            for r in (0, 1, n // 2, n):
                a = s.shifted(s.shifted(s.IndBallL0(max(r, 1)), xd, 0.9, s.NormLinf(1.0)), sd)(yd)
                b = orc.obj_indball_l0(y, x, sj, max(r, 1), delta=0.9)
                if not (a == b or (np.isnan(a) and np.isnan(b))): nbad += 1; print("obj indball", n, k, r, a, b)
            a = s.shifted(s.shifted(s.NormL1(lam), xd, 1.0, s.NormL2(1.0)), sd)(yd); b = orc.obj_l1_b2(y, x, sj, lam, 1.0)
            if not (close(a, b) or (np.isnan(a) and np.isnan(b))): nbad += 1; print("obj b2", n, k, a, b)
        # unboxed iprox: tiny / huge d
        g = rng.normal(size=n)
        for dscale in (1e-300, 1e-8, 1.0, 1e8, 1e300):
            d = np.abs(rng.normal(size=n)) * dscale + dscale * 1e-3
            gd, dd = dev(g, d)
            for kind, H in (("l1", s.NormL1), ("l0", s.NormL0)):
                with np.errstate(all="ignore"):
                    ref = getattr(orc, "iprox_" + kind)(g, d, x, sj, lam)
                yy = s.iprox(s.shifted(s.shifted(H(lam), xd), sd), gd, dd).cpu().numpy()
                nan = np.isnan(yy) & np.isnan(ref)
                if not np.all(nan | (yy.view(np.int64) == ref.view(np.int64))): nbad += 1; print("iprox", kind, n, k, dscale)
print("failing", nbad)
sys.exit(1 if nbad else 0)
