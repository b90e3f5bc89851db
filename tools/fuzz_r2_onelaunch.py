"""Randomised hunt over the one-launch kernels of round 2 (k_sel_coop with the folded first digit, k_b2_coop with flag-in-data
exchange): random sizes across the size classes, data scales that reach the catch-all bins of the fold, tie levels, r and
trust-region regimes; every case against the oracle (top-r: bits; B2: 1e-12).  tools/fuzz_r2_onelaunch.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
bad = 0; t0 = time.time()
for c in range(cases):
    n = int(2 ** rng.uniform(3, 21.3))
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    kind = rng.choice(["plain", "scaled", "wide", "ties", "sparse", "mixed_tiny"])
    if kind == "scaled":
        f = 2.0 ** rng.integers(-70, 70); x, sj, q = x * f, sj * f, q * f
    elif kind == "wide":
        e = rng.integers(-60, 60, size=n); x, sj, q = x * 2.0 ** e, sj * 2.0 ** e, q * 2.0 ** e
    elif kind == "ties":
        lev = int(rng.choice([1, 4, 64])); x, sj, q = (np.round(v * lev) / lev for v in (x, sj, q))
    elif kind == "sparse":
        m = rng.random(n) < 0.8; x, sj, q = np.where(m, 0.0, x), np.where(m, 0.0, sj), np.where(m, 0.0, q)
    elif kind == "mixed_tiny":
        m = rng.random(n) < 0.5; f = 2.0 ** -50; x, sj, q = np.where(m, x * f, x), np.where(m, sj * f, sj), np.where(m, q * f, q)
    xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
    # top-r
    for r in sorted({1, int(rng.integers(1, max(2, n // 50))), int(rng.integers(1, n + 1)), n}):
        delta = float(rng.choice([0.5, 1.0, 1e30])) * max(np.median(np.abs(x)), 1e-300)
        ref = orc.prox_indball_l0_binf(q, x, sj, r, delta)
        y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, delta, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
        if not np.array_equal(y.view(np.int64), ref.view(np.int64)):
            bad += 1; print("TOPR MISMATCH case %d n %d kind %s r %d: %d elements" % (c, n, kind, r, int(np.sum(y.view(np.int64) != ref.view(np.int64)))), flush=True)
    # B2
    nrm = float(np.linalg.norm(x))
    for lam, delta in ((1.0, 1.0), (float(rng.uniform(0.1, 3.0)), float(rng.uniform(0.01, 2.0)) * max(nrm, 1e-300)), (1.0, 1e300)):
        if not np.isfinite(nrm * nrm): continue
        sigma = float(rng.choice([0.5, 1.0, 2.0]))
        ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, 1.0)
        y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, sigma).cpu().numpy()
        scale = max(np.linalg.norm(ref), nrm, 1e-300)
        err = np.max(np.abs(y - ref)) / scale if n else 0.0
        if not (err <= 1e-12):
            bad += 1; print("B2 MISMATCH case %d n %d kind %s lam %g delta %g sigma %g: %.3e" % (c, n, kind, lam, delta, sigma, err), flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez("gpurun_out/fuzz_r2_b2_case%d.npz" % c, x=x, sj=sj, q=q, lam=lam, sigma=sigma, delta=delta, y=y, ref=ref)
    if c % 10 == 9: print("case %d done, %d bad, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
