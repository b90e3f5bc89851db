"""One-off: GroupNormL2Binf / GroupNormL2 on integer and quarter-integer lattices (exact coincidences: ||S|| == sigma lambda,
|X_i| == Delta, zero groups, constant groups), many seeds and group sizes; adjudicated by tests/arbiter.py as in the suite.
tools/fuzz_r2_binf_lattice.py [first_seed] [count]"""
import os, sys, time, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
import test_gpu_stress as T
s = ge.build(); orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(70000 + seed)
    gs = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 17, 32, 64, 100, 128, 200, 256, 300, 512, 600, 1024]))
    ng = int(rng.integers(1, 300 if gs <= 128 else 40))
    n = ng * gs
    lev = float(rng.choice([1.0, 2.0, 4.0]))
    x = rng.integers(-8, 9, size=n) / lev; sj = rng.integers(-2, 3, size=n) / lev; q = rng.integers(-12, 13, size=n) / lev
    mode = int(rng.integers(0, 4))
    if mode == 1: x[: n // 2] = 0.0
    if mode == 2: q[:] = np.repeat(rng.integers(-4, 5, size=ng) / lev, gs)
    if mode == 3: x[:] = 0.0; sj[:] = 0.0
    lam = rng.choice([0.0, 0.25, 0.5, 1.0, 2.0, 3.0, 8.0], size=ng)
    sigma = float(rng.choice([0.25, 0.5, 1.0, 2.0])); delta = float(rng.choice([0.25, 0.5, 1.0, 2.0, 3.0]))
    offs = np.arange(0, n + 1, gs)
    try:
        h = s.GroupNormL2.uniform(lam.tolist(), gs)
        T._binf_run_and_check(s, orc, h, x, sj, q, lam, sigma, delta, offs, "seed %d gs %d" % (seed, gs))
        # plain GroupNormL2 on the same data: 1e-12 against the oracle
        xd, sd, qd = T._dev(x, sj, q)
        y = s.prox(s.shifted(s.shifted(h, xd), sd), qd, sigma).cpu().numpy()
        ref = orc.prox_group_l2(q, x, sj, lam, sigma, offsets=offs)
        sc = max(np.max(np.abs(ref)), np.max(np.abs(x + sj)), 1e-300)
        assert np.max(np.abs(y - ref)) <= 1e-12 * sc, ("plain", seed, gs, float(np.max(np.abs(y - ref))))
    except AssertionError:
        bad += 1; print("FAIL seed", seed, "gs", gs, "ng", ng, "lev", lev, "mode", mode, "sigma", sigma, "delta", delta); traceback.print_exc(limit=3)
    if seed % 20 == 19: print("seed %d done, %d bad, %.0f s" % (seed, bad, time.time() - t0), flush=True)
print("seeds", count, "bad", bad)
sys.exit(1 if bad else 0)
