"""Round 2: ShiftedIndBallL0BInf at mid n on tie-heavy data (one-launch select): us per call."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
for n in (5000, 100_000, 1_000_000, 2_000_000, 4_000_000):
    g = torch.Generator(device="cuda:0").manual_seed(1)
    q0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
    z = torch.zeros(n, dtype=torch.float64, device="cuda:0"); y = torch.empty_like(q0)
    row = []
    for kind in ("continuous", "lattice 1/4", "constant"):
        q = q0 if kind == "continuous" else (torch.round(q0 * 4) / 4 if kind == "lattice 1/4" else torch.full_like(q0, 2.0))
        for r in (max(1, n // 100), n // 2):
            psi = s.shifted(s.shifted(s.IndBallL0(r), z, 1.0, s.NormLinf(1.0)), z)
            for _ in range(3): s.prox_bang(y, psi, q, 1.0)
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(20): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); row.append(ms.value / 20 * 1e3)
    print("n %8d: continuous %7.1f / %7.1f us   lattice %7.1f / %7.1f us   constant %7.1f / %7.1f us   (r = n/100 / n/2)" % ((n,) + tuple(row)), flush=True)
