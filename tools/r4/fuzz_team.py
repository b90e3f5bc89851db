"""Randomised hunt on the team form (csrc/spx_group_team.hip) against the CPU oracle: random sizes on both sides of the on-chip
limit, one group / uniform groups / ragged layouts, data kinds (Gaussian, x = 0, lattices with |x| = Delta, sparse, scaled),
lambda from 'zeroes the group' to 'barely shrinks', Delta from 0.01 to 100, sigma over three decades, fast path on and off, views
from an odd element.  Bar: |dy_i| <= 1e-12 max(|y_i|, |xk_i + sj_i|, ||S_group||) (plain) / the same with the binary128 arbiter
above it (Binf; arbitrated groups are listed).  usage: fuzz_team.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import __graft_entry__ as ge
import arbiter
s = ge.build(); L = s._lib.load()
from oracle import oracle
ctx = s.context("cuda:0"); chi = s.NormLinf(1.0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time(); ncase = 0; nfail = 0; narb = 0; nval_arb = 0; worst_or = 0.0; worst_gpu = 0.0
def dev(a, off):
    t = torch.zeros(a.shape[0] + 2, dtype=torch.float64, device="cuda:0"); t[off:off + a.shape[0]] = torch.from_numpy(a).cuda(); return t[off:off + a.shape[0]]
while time.time() - t0 < budget:
    kind = rng.choice(["one", "one", "uniform", "ragged", "ragged"])
    if kind == "one":
        n = int(10 ** rng.uniform(3.7, 6.7)); offsets = [0, n]
    elif kind == "uniform":
        gs = int(10 ** rng.uniform(3.7, 5.3)); ng = int(rng.integers(2, 40)); n = gs * ng; offsets = list(range(0, n + 1, gs))
    else:
        n = int(10 ** rng.uniform(4.5, 6.5)); k = int(rng.integers(1, 12))
        cuts = sorted(set(int(n * f) for f in rng.uniform(0, 1, size=k)) | {0, n})
        if rng.random() < 0.5: cuts = sorted(set(cuts) | {c + int(rng.integers(1, 40)) for c in cuts[:-1]} - {n + 1})
        offsets = [c for c in cuts if 0 <= c <= n]
    if n > 6_000_000: continue
    data = rng.choice(["gauss", "gauss", "x0", "lattice", "sparse", "small_x", "big_q"])
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    delta = float(10 ** rng.uniform(-2, 2)); sigma = float(10 ** rng.uniform(-1.5, 1.5))
    if data == "x0": x[:] = 0.0
    elif data == "lattice":
        qq = int(rng.choice([1, 2, 4])); x, sj, q = (np.round(v * qq) / qq for v in (x, sj, q)); delta = float(rng.choice([0.5, 1.0, 2.0]))
    elif data == "sparse": x *= (rng.random(n) < 0.05)
    elif data == "small_x": x *= 0.1 * delta
    elif data == "big_q": q *= 100.0
    lam = []
    for a, b in zip(offsets[:-1], offsets[1:]):
        nS = np.linalg.norm((q[a:b] + x[a:b]) + sj[a:b]) if b > a else 1.0
        lam.append(float(rng.choice([0.01, 0.3, 0.9, 0.999, 1.001, 1.5, 30.0])) * max(nS, 1e-3) / sigma)
    off8 = int(rng.integers(0, 2)); fast = int(rng.integers(0, 2)) if kind != "uniform" else 1
    xd, sd, qd = dev(x, off8), dev(sj, off8), dev(q, off8)
    groups = [range(a, b) for a, b in zip(offsets[:-1], offsets[1:])]
    h = s.GroupNormL2(lam, groups)
    offs = np.asarray(offsets, dtype=np.int64)
    L.spx_ctx_set_tuning(ctx, 14, fast)
    for binf in (False, True):
        psi = s.shifted(s.shifted(h, xd, delta, chi), sd) if binf else s.shifted(s.shifted(h, xd), sd)
        y = s.prox(psi, qd, sigma).cpu().numpy()
        with np.errstate(all="ignore"):
            ref = oracle.prox_group_l2_binf(q, x, sj, lam, sigma, delta, offsets=offs) if binf else oracle.prox_group_l2(q, x, sj, lam, sigma, offsets=offs)
        # psi(y) at the prox (the chunked form on large groups), against the oracle
        yd = torch.from_numpy(y).cuda() if off8 == 0 else dev(y, off8)
        if np.all(np.isfinite(y)):
            val = psi(yd)
            with np.errstate(all="ignore"):
                vr = oracle.obj_group_l2(y, x, sj, lam, offsets=offs, delta=(delta if binf else None))
            if not (val == vr or abs(val - vr) <= 1e-12 * abs(vr)):
                # above the plain bar: which side is off?  The oracle adds the squares of a group one after the other (as a generic
                # `norm` loop would): on data with few distinct values its rounding errors line up (n eps, not sqrt(n) eps).  Exact
                # value: math.fsum of the squares (error-free), the square root in extended precision.
                import math
                if binf and np.any(np.abs(sj + y) > 1.1 * delta):
                    exact = float("inf")
                else:
                    w = ((sj + y) + x) if binf else ((x + sj) + y)
                    exact = float(sum(np.longdouble(l) * np.sqrt(np.longdouble(math.fsum((w[a:b] * w[a:b]).tolist()))) for l, a, b in zip(lam, offsets[:-1], offsets[1:])))
                eg, eo = abs(val - exact), abs(vr - exact)
                nval_arb += 1
                if not (eg <= 1e-12 * abs(exact) + eo):
                    nfail += 1
                    print("FAIL psi(y) kind %s data %s n %d groups %d binf %d: gpu %r oracle %r exact %r" % (kind, data, n, len(groups), binf, val, vr, exact), flush=True)
                else:
                    worst_or = max(worst_or, eo / abs(exact)); worst_gpu = max(worst_gpu, eg / abs(exact))
        ncase += 1
        if ncase % 100 == 0: print("... %d cases, %.0f s, %d failures, %d groups arbitrated" % (ncase, time.time() - t0, nfail, narb), flush=True)
        try:
            assert np.array_equal(np.isnan(y), np.isnan(ref))
            v = arbiter.check_group(oracle, y, ref, q, x, sj, np.asarray(lam), sigma, offs, delta=delta if binf else None, what="fuzz", max_arbitrated=4 if n > 1_500_000 else 64)
            narb += v.n_checked
        except AssertionError as e:
            nfail += 1
            print("FAIL kind %s data %s n %d groups %d binf %d fast %d off %d delta %.4g sigma %.4g lam %s: %s" % (kind, data, n, len(groups), binf, fast, off8, delta, sigma, lam[:4], str(e)[:200]), flush=True)
    L.spx_ctx_set_tuning(ctx, 14, 1)
print("fuzz_team: %d cases in %.0f s, %d failures, %d groups arbitrated in binary128 (all sided with the GPU or within the bar)" % (ncase, time.time() - t0, nfail, narb))
print("   psi(y): %d values above the plain 1e-12 bar against the Float64 oracle, adjudicated with an exact sum: worst relative error of the oracle's sequential sum %.2e, of the GPU %.2e" % (nval_arb, worst_or, worst_gpu))
sys.exit(1 if nfail else 0)
