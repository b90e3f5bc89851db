"""Round 3 hunt: ShiftedNormL1B2, two-pass streaming form, the form with xk in LDS
(2^21 < n <= 2^22, every other instance) and the register-resident form, against the CPU oracle
(1e-12 of the norms) on random instances: sizes either side of the form switch, random lambda / sigma / Delta over decades,
data kinds (normal, x = 0, lattices, sparse x, scaled q or x, sorted, heavy tails), views of mixed alignment, y === q.
usage: fuzz_b2.py [instances] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
for it in range(N):
    n = int(rng.choice([int(rng.integers(2, 70_000)), int(rng.integers(1_900_000, 2_300_000)), int(rng.integers((1 << 21) + 1, 3_200_000))]))
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    kind = int(rng.integers(0, 8))
    if kind == 1: x[:] = 0.0
    elif kind == 2: x, sj, q = (np.round(v * 8) / 8 for v in (x, sj, q))
    elif kind == 3: x[rng.random(n) < 0.9] = 0.0
    elif kind == 4: q *= 10.0 ** rng.integers(-3, 3)
    elif kind == 5: x *= 10.0 ** rng.integers(-3, 3)
    elif kind == 6: x = np.sort(x); q = np.sort(q)
    elif kind == 7: x = rng.standard_cauchy(size=n); q = rng.standard_cauchy(size=n)
    lam = float(10.0 ** rng.uniform(-2.5, 1.5)); sigma = float(10.0 ** rng.uniform(-1, 1))
    delta = float(10.0 ** rng.uniform(-3, 3)) * (np.linalg.norm(x) + 1.0) / 10.0 ** rng.integers(0, 4)
    off = [int(v) for v in rng.integers(0, 2, size=4)] if it % 4 == 0 else [0, 0, 0, 0]       # views of mixed alignment
    mk = lambda a, o: torch.cat([torch.zeros(o, dtype=torch.float64), torch.from_numpy(np.ascontiguousarray(a))]).cuda()[o:]
    xd, sd, qd, yd = mk(x, off[0]), mk(sj, off[1]), mk(q, off[2]), mk(np.full(n, np.nan), off[3])
    with np.errstate(all="ignore"):
        ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, 1.0)
    psi = s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd)
    L.spx_ctx_set_tuning(ctx, 12, it % 2)      # 2^21 < n <= 2^22: xk parked in LDS (1, the default) / the streaming form (0)
    s.prox_bang(yd, psi, qd, sigma)
    y = yd.cpu().numpy()
    scale = max(np.linalg.norm(ref), np.linalg.norm(x), np.linalg.norm(sj + q), 1e-300)
    err = float(np.max(np.abs(y - ref))) / scale
    ok = err <= 1e-12
    if it % 3 == 0:
        qa = qd.clone(); s.prox_bang(qa, psi, qa, sigma)
        ok = ok and float(np.max(np.abs(qa.cpu().numpy() - ref))) / scale <= 1e-12
    L.spx_ctx_set_tuning(ctx, 12, 1)
    rc = L.spx_sync(ctx)
    if not ok or rc:
        bad += 1
        print("MISMATCH it=%d n=%d kind=%d lam=%g sigma=%g delta=%g off=%s err=%.3e rc=%d" % (it, n, kind, lam, sigma, delta, off, err, rc), flush=True)
    if it % 10 == 9:
        print("... %d instances, %d mismatches" % (it + 1, bad), flush=True)
print("done: %d instances, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
