"""Round 2: iprox! with d <= 0 EVERYWHERE (the reference's @assert fails): time per call at n = 1e8 -- an error path must not stall."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = 100_000_000
x = torch.randn(n, dtype=torch.float64, device="cuda"); z = torch.zeros_like(x); g = torch.randn_like(x); y = torch.empty_like(x)
for name, d in (("d > 0", torch.ones_like(x)), ("d < 0 everywhere", -torch.ones_like(x))):
    psi = s.shifted(s.shifted(s.NormL1(1.0), x), z)
    for _ in range(2): s.iprox_bang(y, psi, g, d, check=False)
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(5): s.iprox_bang(y, psi, g, d, check=False)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); print("%-18s %8.3f ms per call" % (name, ms.value / 5), flush=True)
