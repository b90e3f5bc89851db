"""Round 2: ShiftedNormL1B2, register-resident one-launch form: time per call and agreement with the oracle over n (active and inactive trust region)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
for n in (1000, 10_000, 16_000, 30_000, 100_000, 300_000, 1_000_000, 2_000_000):
    for delta in (1.0, 1e9):
        g = torch.Generator(device="cuda:0").manual_seed(99)
        xk = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
        q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
        psi = s.shifted(s.shifted(s.NormL1(1.0), xk, delta, s.NormL2(1.0)), sj)
        ref = orc.prox_l1_b2(q.cpu().numpy(), xk.cpu().numpy(), sj.cpu().numpy(), 1.0, 1.0, delta, 1.0)
        row = []
        for epl in (8,):
            pass
            for _ in range(3): s.prox_bang(y, psi, q, 1.0)
            err = np.max(np.abs(y.cpu().numpy() - ref)) / max(np.linalg.norm(ref), 1e-300)
            ts = []
            for rnd in range(5):
                ms = ctypes.c_float(); L.spx_timer_start(ctx)
                for _ in range(50): s.prox_bang(y, psi, q, 1.0)
                L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 50 * 1e3)
            row.append((sorted(ts)[2], err))
        pass
        print("n %8d Delta %-6g: %6.1f us (err %.1e)" % (n, delta, row[0][0], row[0][1]), flush=True)
