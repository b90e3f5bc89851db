"""One-off: GroupNormL2Binf on continuous data over EXTREME parameter ranges (lambda sigma and Delta over 16 decades, data
scales over 12, sparse iterates, mixed group sizes incl. ragged), adjudicated by tests/arbiter.py as in the suite.
tools/fuzz_r2_binf_random.py [first_seed] [count]"""
import os, sys, time, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
import test_gpu_stress as T
s = ge.build(); orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(90000 + seed)
    ragged = rng.random() < 0.3
    if ragged:
        ng = int(rng.integers(1, 400)); sizes = rng.integers(1, int(rng.choice([8, 64, 300, 2000])), size=ng)
    else:
        gs = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 128, 129, 256, 500, 512, 700, 3000])); ng = int(rng.integers(1, 300 if gs <= 512 else 12))
        sizes = np.full(ng, gs)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64); n = int(offs[-1])
    scale = 10.0 ** rng.uniform(-6, 6)
    x = rng.normal(size=n) * scale * 10.0 ** rng.uniform(-3, 1); sj = rng.uniform(-0.5, 0.5, size=n) * scale * float(rng.choice([0.0, 1.0]))
    q = rng.normal(size=n) * scale * 10.0 ** rng.uniform(-2, 2)
    if rng.random() < 0.4:   # sparse iterate: most groups of x are zero
        keep = np.repeat(rng.random(ng) < 0.15, sizes); x = np.where(keep, x, 0.0)
    lam = scale * 10.0 ** rng.uniform(-8, 8, size=ng) * float(rng.choice([1.0, 1.0, 0.0]) if rng.random() < 0.1 else 1.0)
    sigma = 10.0 ** rng.uniform(-3, 3); delta = scale * 10.0 ** rng.uniform(-8, 8)
    try:
        if ragged:
            h = s.GroupNormL2(lam.tolist(), [range(int(a), int(b)) for a, b in zip(offs[:-1], offs[1:])])
        else:
            h = s.GroupNormL2.uniform(lam.tolist(), int(sizes[0]))
        T._binf_run_and_check(s, orc, h, x, sj, q, lam, sigma, delta, offs, "seed %d" % seed)
    except AssertionError:
        bad += 1; print("FAIL seed", seed, "ragged", ragged, "ng", ng, "n", n, "scale %.3g sigma %.3g delta %.3g" % (scale, sigma, delta)); traceback.print_exc(limit=3)
    except Exception as e:
        bad += 1; print("ERROR seed", seed, type(e).__name__, e)
    if seed % 20 == 19: print("seed %d done, %d bad, %.0f s" % (seed, bad, time.time() - t0), flush=True)
print("seeds", count, "bad", bad)
sys.exit(1 if bad else 0)
