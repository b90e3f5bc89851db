"""Round 2: ShiftedIndBallL0BInf at n <= 65536 -- the one-workgroup kernel (key 6 = 1) vs the register-resident one-launch select."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
for n in (100, 1000, 4096, 8192, 10_000, 16_384, 30_000, 65_536):
    row = []
    for mode in (1, 0):
        L.spx_ctx_set_tuning(ctx, 6, mode)
        g = torch.Generator(device="cuda:0").manual_seed(1)
        x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
        q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
        psi = s.shifted(s.shifted(s.IndBallL0(max(1, n // 100)), x, 1.0, s.NormLinf(1.0)), sj)
        for _ in range(3): s.prox_bang(y, psi, q, 1.0)
        ts = []
        for rnd in range(5):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(50): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 50 * 1e3)
        row.append(sorted(ts)[2])
    L.spx_ctx_set_tuning(ctx, 6, 1)
    print("n %6d: one workgroup %6.1f us   one-launch grid %6.1f us" % (n, row[0], row[1]), flush=True)
