"""Round 3, late: ShiftedNormL1B2 at sizes whose passes take their tiles on demand (n in [1.3e7, 3e7]) -- random SEQUENCES of calls
on one context (active / inactive / borderline Delta, so that the speculative pass is right, wrong and too close to call),
y disjoint and y === q, each result against the CPU oracle (1e-12 of the norms; inactive results bit for bit).
usage: fuzz_b2_large.py [instances] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
for it in range(N):
    n = int(rng.integers(13_000_000, 30_000_000))
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    kind = int(rng.integers(0, 4))
    if kind == 1: x, sj, q = (np.round(v * 8) / 8 for v in (x, sj, q))
    elif kind == 2: x[rng.random(n) < 0.9] = 0.0
    elif kind == 3: x = np.sort(x); q = np.sort(q)
    xd, sd, qd = (torch.from_numpy(v).cuda() for v in (x, sj, q))
    lam = float(10.0 ** rng.uniform(-1.5, 1.0))
    y_in = orc.prox_l1_b2(q, x, sj, lam, 1.0, 1e300, 1.0)
    chi = float(np.linalg.norm(sj + y_in))
    choices = [1e300, 1e300, chi * 0.5, chi * 0.01, chi * (1 + 1e-10), chi * (1 - 1e-10), chi * (1 + 1e-14), chi]
    for step in range(5):
        delta = float(rng.choice(choices))
        psi = s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd)
        ref = y_in if delta == 1e300 else orc.prox_l1_b2(q, x, sj, lam, 1.0, delta, 1.0)
        alias = rng.random() < 0.25
        if alias:
            yd = qd.clone(); s.prox_bang(yd, psi, yd, 1.0)
        else:
            yd = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda:0"); s.prox_bang(yd, psi, qd, 1.0)
        y = yd.cpu().numpy()
        scale = max(np.linalg.norm(ref), np.linalg.norm(x), np.linalg.norm(sj + q), 1e-300)
        err = float(np.max(np.abs(y - ref))) / scale
        ok = err <= 1e-12 and not np.isnan(y).any()
        if delta == 1e300: ok = ok and np.array_equal(y.view(np.int64), ref.view(np.int64))
        rc = L.spx_sync(ctx)
        if not ok or rc:
            bad += 1
            print("MISMATCH it=%d step=%d n=%d kind=%d lam=%g delta=%.17g alias=%s err=%.3e rc=%d" % (it, step, n, kind, lam, delta, alias, err, rc), flush=True)
    print("... instance %d (n = %d, kind %d): %d mismatches so far" % (it, n, kind, bad), flush=True)
print("done: %d instances x 5 calls, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
