"""Round 3: ShiftedNormL1B2 (two streaming passes) -- ms per call over (lambda, Delta) and data kinds at n = SPX_N (1e8), checked
against an independent evaluation of the fixed point: froot(eta) recomputed by torch from the returned y
(||sj + y|| must equal Delta in the scaled branch to 1e-12) and, at SPX_CHECK_N elements, against the oracle."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = int(float(os.environ.get("SPX_N", "1e8")))
g = torch.Generator(device="cuda:0").manual_seed(int(os.environ.get("SPX_SEED", "1")))
x0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
s0 = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
q0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
y = torch.empty_like(q0)
for kind in os.environ.get("SPX_KINDS", "normal,x=0,q*0.01,x*0.05,sparse_x,sorted,lattice8").split(","):
    x, sj, q = x0, s0, q0
    if kind == "x=0": x = torch.zeros_like(x0); sj = torch.zeros_like(s0)
    elif kind == "q*0.01": q = q0 * 0.01
    elif kind == "x*0.05": x = x0 * 0.05
    elif kind == "sparse_x": x = torch.where(torch.rand(n, device="cuda:0", generator=g) < 0.9, torch.zeros_like(x0), x0)
    elif kind == "sorted": x = torch.sort(x0)[0]; q = torch.sort(q0)[0]
    elif kind == "lattice8": x, sj, q = (torch.round(v * 8) / 8 for v in (x0, s0, q0))
    for lam, delta in ((1.0, 1.0), (0.01, 1e-3), (30.0, 1.0), (1.0, 1000.0), (1.0, 1e9)):
        psi = s.shifted(s.shifted(s.NormL1(lam), x, delta, s.NormL2(1.0)), sj)
        s.prox_bang(y, psi, q, 1.0); s.prox_bang(y, psi, q, 1.0)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        nrm = float(torch.linalg.vector_norm(sj + y))
        rc = L.spx_sync(ctx)
        print("%-9s lambda %-5g Delta %-6g %8.3f ms per call | ||sj + y|| / Delta = %.15f | sync rc %d" % (kind, lam, delta, ms.value / 5, nrm / delta, rc), flush=True)
