"""One-off: GroupNormL2Binf, reversed brackets with entries OUTSIDE the trust region (the piece iteration of the deferred-list
kernel and the literal evaluation behind it), small groups where jumps of R are likeliest, vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(2025)
nbad = 0; worst = 0.0; total = 0; nonzero = 0
for gs in (1, 2, 3, 4, 8, 16, 40, 128):
    ng = 20000 if gs <= 16 else 4000
    n = ng * gs
    for rep in range(12):
        xs = float(rng.choice([0.05, 0.2, 0.5])) / np.sqrt(gs)
        x = rng.normal(size=n) * xs
        sj = rng.uniform(-0.5, 0.5, size=n) * float(rng.choice([0.0, 1.0]))
        q = rng.normal(size=n) * float(rng.choice([1.0, 0.1, 3.0]))
        sigma = float(10.0 ** rng.uniform(-1, 1))
        delta = float(10.0 ** rng.uniform(-3, -0.5)) * xs * np.sqrt(gs) * 4
        S = ((q + x) + sj).reshape(ng, gs); nS = np.linalg.norm(S, axis=1); nX = np.linalg.norm(x.reshape(ng, gs), axis=1)
        lam = np.maximum(nS, 1e-3) * 10.0 ** rng.uniform(0, 1.5, size=ng) / sigma / np.maximum(1e-3, 1.0 - np.minimum(nX, 0.95))
        xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
        h = s.GroupNormL2.uniform(lam.tolist(), gs)
        with np.errstate(all="ignore"):
            ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, gsize=gs)
        y = s.prox(s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd), qd, sigma).cpu().numpy()
        rev = nS + sigma * lam * nX < sigma * lam
        out = np.abs(x.reshape(ng, gs)).max(axis=1) > delta
        total += int((rev & out).sum())
        nonzero += int(((np.abs(ref + (x + sj)).reshape(ng, gs).max(axis=1) > 0) & rev & out).sum())
        fin = np.isfinite(ref)
        sc = np.maximum(np.abs(np.where(fin, ref, 0.0)).reshape(ng, gs), np.maximum(nS, 1e-300)[:, None])
        gerr = (np.abs(np.where(fin, y - ref, 0.0)).reshape(ng, gs) / sc).max(axis=1)
        q99, emax = float(np.quantile(gerr, 0.99)), float(gerr.max())
        worst = max(worst, emax)
        if q99 > 1e-12 or emax > 1e-6 or not np.array_equal(fin, np.isfinite(y)):
            nbad += 1
            g = int(np.argmax(gerr))
            print("gs %d rep %d sigma %.3g delta %.3g xs %.3g: q99 %.2e max %.2e group %d lam %.4g nS %.4g nX %.4g" % (gs, rep, sigma, delta, xs, q99, emax, g, lam[g], nS[g], nX[g]))
    print("gs", gs, "done", flush=True)
# the same regime through the other kernel families: LDS-resident (700, 2500), general (5000), ragged CSR with and without
# the size hint (register tiles with a memory-resident deferred list / LDS / general), gather-index groups
def check(tag, h, x, sj, q, lam, sigma, delta, offs):
    global nbad, worst, total
    n = x.size
    xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
    with np.errstate(all="ignore"):
        ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, offsets=offs)
    y = s.prox(s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd), qd, sigma).cpu().numpy()
    S = (q + x) + sj
    fin = np.isfinite(ref)
    err = []
    for g in range(len(offs) - 1):
        lo, hi = offs[g], offs[g + 1]
        sc = np.maximum(np.abs(np.where(fin[lo:hi], ref[lo:hi], 0.0)), max(np.linalg.norm(S[lo:hi]), 1e-300))
        err.append(float(np.max(np.abs(np.where(fin[lo:hi], y[lo:hi] - ref[lo:hi], 0.0)) / sc)) if hi > lo else 0.0)
    err = np.array(err); q99, emax = float(np.quantile(err, 0.99)), float(err.max())
    worst = max(worst, emax); total += len(offs) - 1
    if q99 > 1e-12 or emax > 1e-6 or not np.array_equal(fin, np.isfinite(y)):
        nbad += 1; print("%s: q99 %.2e max %.2e" % (tag, q99, emax))
for rep in range(6):
    for tag in ("lds700", "lds2500", "general5000", "ragged_hint", "ragged_nohint", "gather"):
        if tag in ("lds700", "lds2500", "general5000"):
            gs = int(tag.lstrip("ldsgenra")); ng = 60; sizes = np.full(ng, gs)
        else:
            ng = 3000; sizes = rng.integers(1, 41, size=ng)
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64); n = int(offs[-1])
        scale = 1.0 / np.sqrt(np.repeat(sizes, sizes))
        xs = float(rng.choice([0.05, 0.2, 0.5]))
        x = rng.normal(size=n) * xs * scale; sj = rng.uniform(-0.5, 0.5, size=n) * scale * 4; q = rng.normal(size=n) * scale * 4
        sigma = float(10.0 ** rng.uniform(-1, 1)); delta = float(10.0 ** rng.uniform(-3, -0.5)) * xs / np.sqrt(np.median(sizes)) * 4
        S = (q + x) + sj
        nS = np.array([np.linalg.norm(S[offs[g]:offs[g + 1]]) for g in range(ng)]); nX = np.array([np.linalg.norm(x[offs[g]:offs[g + 1]]) for g in range(ng)])
        lam = np.maximum(nS, 1e-3) * 10.0 ** rng.uniform(0, 1.5, size=ng) / sigma / np.maximum(1e-3, 1.0 - np.minimum(nX, 0.95))
        if tag == "gather":
            h = s.GroupNormL2(lam.tolist(), [list(range(int(offs[g + 1]) - 1, int(offs[g]) - 1, -1)) for g in range(ng)])  # reversed order: gather kernel
        elif tag == "ragged_nohint":
            h = s.GroupNormL2(lam.tolist(), [range(int(offs[g]), int(offs[g + 1])) for g in range(ng)])
        else:
            h = s.GroupNormL2.ragged(lam.tolist(), offs)
        check("%s rep %d" % (tag, rep), h, x, sj, q, lam, sigma, delta, offs)
    print("other kernel families rep", rep, "done", flush=True)
print("reversed brackets with entries outside the trust region: %d (reference result nonzero in %d)  worst %.2e  failing configs %d" % (total, nonzero, worst, nbad))
sys.exit(1 if nbad else 0)
