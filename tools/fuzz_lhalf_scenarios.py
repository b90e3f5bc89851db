"""One-off: structured scenarios for RootNormLhalf(Box) vs the oracle (1e-12 on the operand scale; ties judged by objective)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(13)
nbad = 0
n = 20000
for k in range(11):
    for lam, sigma in ((1.0, 1.0), (1e-3, 1.0), (10.0, 0.1), (0.5, 100.0), (1e3, 1e-3)):
        for lo, up in ((-1.0, 1.0), (-0.1, 2.0), (0.0, 0.0), (-np.inf, np.inf), (0.5, 1.5)):
            x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
            if k == 0: q = -(x + sj)                      # xsq = 0
            elif k == 1: x[:] = -lo if np.isfinite(lo) else 0.0   # x = -l: bound candidate sits at zero of sqrt
            elif k == 2: x[:] = -up if np.isfinite(up) else 0.0
            elif k == 3: x[:] = 0.0; sj[:] = 0.0
            elif k == 4: q *= 1e-9
            elif k == 5: q *= 1e6; x *= 1e6
            elif k == 6:                                   # right at the threshold p = 1.5 (sigma lambda)^(2/3): a = 1
                p = 1.5 * (sigma * lam) ** (2.0 / 3.0)
                q = np.sign(q) * p * (1 + rng.integers(-3, 4, size=n) * 2.2e-16) - (x + sj)
            elif k == 7: sj = -x.copy()                    # xs = 0
            elif k == 8: x = np.abs(x) * 1e-300
            elif k == 9: q = np.round(q, 1); x = np.round(x, 1); sj = np.round(sj, 1)
            elif k == 10: x = rng.standard_cauchy(n)
            xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
            with np.errstate(all="ignore"):
                ref_u = orc.prox_lhalf(q, x, sj, lam, sigma)
                ref_b = orc.prox_lhalf_box(q, x, sj, lam, sigma, lo, up)
            yu = s.prox(s.shifted(s.shifted(s.RootNormLhalf(lam), xd), sd), qd, sigma).cpu().numpy()
            yb = s.prox(s.shifted(s.shifted(s.RootNormLhalf(lam), xd, lo, up), sd), qd, sigma).cpu().numpy()
            sc = np.maximum(np.maximum(np.abs(x + sj), np.abs(q)), 1e-300)
            for tag, y, ref in (("lhalf", yu, ref_u), ("box", yb, ref_b)):
                scl = np.maximum(sc, np.abs(ref))
                m = np.abs(y - ref) > 1e-12 * scl
                if tag == "box" and m.any():
                    with np.errstate(all="ignore"):
                        f = lambda t: (t - q) ** 2 / 2 / sigma + lam * np.sqrt(np.abs(t + (x + sj)))
                        tie = np.abs(f(y) - f(ref)) <= 1e-12 * np.maximum(np.abs(f(ref)), 1e-300)
                    m = m & ~tie
                if m.any():
                    nbad += 1; i = int(np.nonzero(m)[0][0])
                    print("%s scen %d lam %g sigma %g [%g,%g]: %d bad; q=%.17g x=%.17g s=%.17g gpu=%.17g ref=%.17g" % (tag, k, lam, sigma, lo, up, int(m.sum()), q[i], x[i], sj[i], y[i], ref[i]))
print("failing", nbad)
sys.exit(1 if nbad else 0)
