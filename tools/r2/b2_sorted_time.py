"""Round 2: ShiftedNormL1B2 at n = 1e8 on sorted / clustered input (the sample that predicts the second trial is drawn in chunks)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = 100_000_000
g = torch.Generator(device="cuda:0").manual_seed(3)
x0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); q0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5; y = torch.empty_like(q0)
for kind in ("generic", "x sorted", "x and q sorted", "|x| sorted", "x in blocks"):
    if kind == "generic": x, q = x0, q0
    elif kind == "x sorted": x, q = torch.sort(x0)[0], q0
    elif kind == "x and q sorted": x, q = torch.sort(x0)[0], torch.sort(q0)[0]
    elif kind == "|x| sorted": x, q = x0[torch.argsort(x0.abs())], q0
    else: x, q = x0 * (1.0 + 9.0 * ((torch.arange(n, device=x0.device) // 3_000_000) % 2 == 0)), q0
    for lam, delta in ((1.0, 1.0), (0.3, 100.0)):
        psi = s.shifted(s.shifted(s.NormL1(lam), x, delta, s.NormL2(1.0)), sj)
        for _ in range(2): s.prox_bang(y, psi, q, 1.0)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms)); print("%-16s lambda %-4g Delta %-5g %8.3f ms per call" % (kind, lam, delta, ms.value / 5), flush=True)
