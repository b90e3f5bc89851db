"""One-off: the sample-predicted top-r pipeline on degenerate data at the three front-kernel sample sizes (1 / 2 / 4 samples
per lane: n just above 2^21, 2^23, 2^25); bits against the oracle.  tools/fuzz_r2_topr_pipeline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
bad = 0; t0 = time.time()
for n in ((1 << 21) + 12345, (1 << 23) + 777, (1 << 25) + 31):
    rng = np.random.default_rng(n)
    for kind in ("constant", "two_values", "lattice4", "wide", "tiny+normal", "inf_nan", "sorted", "few_big"):
        x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
        if kind == "constant": x[:] = 0.5; sj[:] = 0.25; q[:] = -2.0
        elif kind == "two_values": x[:] = 0.0; sj[:] = 0.0; q = np.where(rng.random(n) < 0.3, 1.5, -0.75)
        elif kind == "lattice4": x, sj, q = (np.round(v * 4) / 4 for v in (x, sj, q))
        elif kind == "wide": e = rng.integers(-40, 40, size=n); x, sj, q = x * 2.0 ** e, sj * 2.0 ** e, q * 2.0 ** e
        elif kind == "tiny+normal": m = rng.random(n) < 0.5; f = 2.0 ** -45; x, sj, q = np.where(m, x * f, x), np.where(m, sj * f, sj), np.where(m, q * f, q)
        elif kind == "inf_nan": q[rng.integers(0, n, size=50)] = np.inf; q[rng.integers(0, n, size=30)] = np.nan
        elif kind == "sorted": q = np.sort(q); x[:] = 0.0; sj[:] = 0.0
        elif kind == "few_big": x[:] = 0.0; sj[:] = 0.0; q = q * 1e-6; q[rng.integers(0, n, size=1000)] = rng.normal(size=1000) * 1e3
        xd, sd, qd = (torch.from_numpy(a).cuda() for a in (x, sj, q))
        for r in (1, 1000, n // 100, n // 2, n - 5):
            with np.errstate(all="ignore"):
                ref = orc.prox_indball_l0_binf(q, x, sj, r, 0.9)
            y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.9, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
            same = np.array_equal(y.view(np.int64), ref.view(np.int64)) or bool(np.all((y.view(np.int64) == ref.view(np.int64)) | (np.isnan(y) & np.isnan(ref))))
            if not same:
                bad += 1; print("MISMATCH n %d kind %s r %d: %d elements" % (n, kind, r, int(np.sum(y.view(np.int64) != ref.view(np.int64)))), flush=True)
        print("n %d kind %s done, %d bad, %.0f s" % (n, kind, bad, time.time() - t0), flush=True)
        del xd, sd, qd
print("bad", bad)
sys.exit(1 if bad else 0)
