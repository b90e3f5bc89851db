"""Round 2: ShiftedNormL1B2 -- in-launch iteration (k_b2_coop, key 7 = 1) vs the host-driven loop (key 7 = 0): per-call time and agreement."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
def run(n, delta, lam, iters):
    g = torch.Generator(device="cuda:0").manual_seed(99)
    xk = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.NormL1(lam), xk, delta, s.NormL2(1.0)), sj)
    out = {}; ys = {}
    for mode in (1, 0):
        L.spx_ctx_set_tuning(ctx, 7, mode)
        for _ in range(2): s.prox_bang(y, psi, q, 1.0)
        ys[mode] = y.clone()
        ts = []
        for rnd in range(3):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(iters): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / iters * 1e3)
        out[mode] = sorted(ts)[1]
    L.spx_ctx_set_tuning(ctx, 7, 1)
    diff = float((ys[1] - ys[0]).abs().max() / max(float(ys[0].abs().max()), 1e-300))
    extra = ""
    if n <= 2_000_000:
        ref = orc.prox_l1_b2(q.cpu().numpy(), xk.cpu().numpy(), sj.cpu().numpy(), lam, 1.0, delta, 1.0)
        extra = "  vs oracle %.1e" % (np.max(np.abs(ys[1].cpu().numpy() - ref)) / max(np.linalg.norm(ref), 1e-300))
    print("n %10d Delta %-8g lambda %-5g: in-launch %8.1f us   host loop %8.1f us   max rel diff %.1e%s" % (n, delta, lam, out[1], out[0], diff, extra), flush=True)
for n in (10_000, 1_000_000, 2_000_000):
    for delta in (1.0, 1e9): run(n, delta, 1.0, 30)
for delta, lam in ((1.0, 1.0), (1e9, 1.0), (1e3, 1.0), (1e-3, 30.0)): run(100_000_000, delta, lam, 5)
