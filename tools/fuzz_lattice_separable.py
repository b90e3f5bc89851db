"""One-off: lattice data (multiples of 1/4) for the separable operators, B2 and top-r: exact ties everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(3)
bits = lambda a, b: np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))
bad = 0
for rep in range(40):
    n = int(rng.integers(1, 20000))
    x = rng.integers(-8, 9, size=n) / 4.0; sj = rng.integers(-4, 5, size=n) / 4.0; q = rng.integers(-12, 13, size=n) / 4.0
    lam = float(rng.choice([0.0, 0.25, 0.5, 1.0, 2.0])); sigma = float(rng.choice([0.25, 0.5, 1.0, 2.0, 4.0]))
    lo = float(rng.choice([-2.0, -1.0, -0.5, 0.0])); up = float(rng.choice([0.0, 0.5, 1.0, 2.0]))
    xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
    with np.errstate(all="ignore"):
        for H, nm in ((s.NormL1, "l1"), (s.NormL0, "l0")):
            y = s.prox(s.shifted(s.shifted(H(lam), xd), sd), qd, sigma).cpu().numpy()
            if not bits(y, getattr(orc, "prox_" + nm)(q, x, sj, lam, sigma)): bad += 1; print("MISMATCH", nm, rep)
            y = s.prox(s.shifted(s.shifted(H(lam), xd, lo, up), sd), qd, sigma).cpu().numpy()
            if not bits(y, getattr(orc, "prox_%s_box" % nm)(q, x, sj, lam, sigma, lo, up)): bad += 1; print("MISMATCH", nm, "box", rep)
            d = rng.choice([-2.0, -1.0, 0.0, 0.5, 1.0, 4.0], size=n); dd = torch.from_numpy(d).cuda()
            y = s.iprox(s.shifted(s.shifted(H(lam), xd, lo, up), sd), qd, dd).cpu().numpy()
            if not bits(y, getattr(orc, "iprox_%s_box" % nm)(q, d, x, sj, lam, lo, up)): bad += 1; print("MISMATCH iprox", nm, rep)
        for box in (False, True):
            if box:
                y = s.prox(s.shifted(s.shifted(s.RootNormLhalf(lam), xd, lo, up), sd), qd, sigma).cpu().numpy()
                ref = orc.prox_lhalf_box(q, x, sj, lam, sigma, lo, up)
            else:
                y = s.prox(s.shifted(s.shifted(s.RootNormLhalf(lam), xd), sd), qd, sigma).cpu().numpy()
                ref = orc.prox_lhalf(q, x, sj, lam, sigma)
            sc = np.maximum(np.maximum(np.abs(ref), np.abs(x + sj)), np.abs(q))
            m = np.abs(y - ref) > 1e-12 * np.maximum(sc, 1e-300)
            if m.any():
                f = lambda t: (t - q) ** 2 / 2 / sigma + lam * np.sqrt(np.abs(t + (x + sj)))
                tie = np.abs(f(y) - f(ref)) <= 1e-13 * np.maximum(np.abs(f(ref)), 1e-300)
                if np.any(m & ~tie):
                    bad += 1; i = int(np.nonzero(m & ~tie)[0][0])
                    print("MISMATCH lhalf box=%s rep %d: q=%g x=%g s=%g lam=%g sigma=%g lo=%g up=%g gpu=%.17g ref=%.17g" % (box, rep, q[i], x[i], sj[i], lam, sigma, lo, up, y[i], ref[i]))
                else:
                    print("  (lhalf box=%s rep %d: %d exact ties resolved to another candidate)" % (box, rep, int(m.sum())))
        r = int(rng.integers(1, n + 1))
        y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.75, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
        if not bits(y, orc.prox_indball_l0_binf(q, x, sj, r, 0.75)): bad += 1; print("MISMATCH indball", rep)
        delta = float(rng.choice([0.5, 2.0, 50.0]))
        y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, sigma).cpu().numpy()
        ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, 1.0)
        if np.max(np.abs(y - ref)) > 1e-12 * max(np.linalg.norm(ref), np.linalg.norm(x), 1.0): bad += 1; print("MISMATCH b2", rep, float(np.max(np.abs(y - ref))))
print("mismatches", bad)
sys.exit(1 if bad else 0)
