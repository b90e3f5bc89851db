"""Runs fixed-seed stress tests of tests/test_gpu_stress.py over MANY more seeds than the suite does (one-off hunts):
tools/fuzz_r2_more_seeds.py [first_seed] [count]"""
import os, sys, time, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import __graft_entry__ as ge
from oracle import oracle as orc
import test_gpu_stress as T
s = ge.build(); orc.build()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0; t0 = time.time()
for seed in range(first, first + count):
    try:
        T.test_lattice_separable_b2_topr.__wrapped__(s, orc, seed) if hasattr(T.test_lattice_separable_b2_topr, "__wrapped__") else T.test_lattice_separable_b2_topr(s, orc, seed)
    except AssertionError:
        bad += 1; print("FAIL lattice seed", seed); traceback.print_exc(limit=2)
    if seed % 10 == 9: print("seed %d done, %d bad, %.0f s" % (seed, bad, time.time() - t0), flush=True)
print("seeds", count, "bad", bad)
sys.exit(1 if bad else 0)
