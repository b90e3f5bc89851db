"""Round-2 smoke of the in-launch synchronised top-r kernels: bit-exact vs the oracle over the size classes, then timings."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
bits = lambda a, b: np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))
bad = 0
for n in (65537, 100_003, 1_000_000, (1 << 21), (1 << 21) + 1, 3_000_001, (1 << 22) + 4321):
    rng = np.random.default_rng(n)
    for quant in (None, 16):
        x, sj, q = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)
        if quant: x, sj, q = (np.round(v * quant) / quant for v in (x, sj, q))
        xd, sd, qd = (torch.from_numpy(v).cuda() for v in (x, sj, q))
        for r in (1, 7, n // 100, n // 2, n - 3, n, n + 5):
            ref = orc.prox_indball_l0_binf(q, x, sj, r, 0.8)
            psi = s.shifted(s.shifted(s.IndBallL0(r), xd, 0.8, s.NormLinf(1.0)), sd)
            y = s.prox(psi, qd, 1.0).cpu().numpy()
            ok = bits(y, ref)
            q2 = qd.clone(); s.prox_bang(q2, psi, q2, 1.0)      # aliased
            ok2 = bits(q2.cpu().numpy(), ref)
            if not (ok and ok2): bad += 1; print("MISMATCH n %d quant %s r %d disjoint %s aliased %s" % (n, quant, r, ok, ok2), flush=True)
    print("n", n, "done", flush=True)
print("mismatches", bad)
def timed(n, r, iters=50):
    g = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.IndBallL0(r), x, 1.0, s.NormLinf(1.0)), sj)
    out = {}
    for mode in (1, 0):
        L.spx_ctx_set_tuning(ctx, 7, mode)
        for _ in range(3): s.prox_bang(y, psi, q, 1.0)
        ts = []
        for rnd in range(5):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(iters): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / iters * 1e3)
        ts.sort(); out[mode] = ts[2]
    L.spx_ctx_set_tuning(ctx, 7, 1)
    print("n %10d r %9d: in-launch %.1f us   multi-launch %.1f us   (32 B/el at 8 TB/s: %.1f us)" % (n, r, out[1], out[0], 32 * n / 8e6), flush=True)
for n in (100_000, 1_000_000, 2_000_000, 2_500_000, 4_000_000, 10_000_000, 100_000_000):
    timed(n, max(1, n // 100), 50 if n < 5e7 else 20)
timed(100_000_000, 50_000_000, 20)
sys.exit(1 if bad else 0)
