"""Round 2: ShiftedIndBallL0BInf at n = 1e8 on tie-heavy data (the sample-predicted band overflows -> exact fallback): ms per call."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = 100_000_000
g = torch.Generator(device="cuda:0").manual_seed(int(os.environ.get("SPX_SEED", "1")))
q0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
z = torch.zeros(n, dtype=torch.float64, device="cuda:0"); y = torch.empty_like(q0)
for kind in os.environ.get("SPX_KINDS", "continuous,lattice 1/4,lattice 1,two values,constant").split(","):
    if kind == "continuous": q = q0
    elif kind == "lattice 1/4": q = torch.round(q0 * 4) / 4
    elif kind == "lattice 1": q = torch.round(q0)
    elif kind == "two values": q = torch.where(q0 > 0.5, torch.full_like(q0, 1.5), torch.full_like(q0, -0.75))
    elif kind == "sorted": q = torch.sort(q0)[0]
    elif kind == "sorted |v|": q = q0[torch.argsort(q0.abs())]
    elif kind == "blocks": q = q0 * (1.0 + 4.0 * ((torch.arange(n, device=q0.device) // 1_000_000) % 2 == 0))
    else: q = torch.full_like(q0, 2.0)
    for r in [int(v) for v in os.environ.get("SPX_RS", "%d,%d" % (n // 100, n // 2)).split(",")]:
        psi = s.shifted(s.shifted(s.IndBallL0(r), z, 1.0, s.NormLinf(1.0)), z)
        for _ in range(2): s.prox_bang(y, psi, q, 1.0)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        print("%-12s r=%-9d %8.3f ms per call" % (kind, r, ms.value / 5), flush=True)
