"""ShiftedNormL1Box with vector bounds (48 B/element) at n = 1e8"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1)
n = 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
lv = -1.0 - 0.1 * torch.rand(n, dtype=torch.float64, device=dev, generator=g); uv = 1.0 + 0.1 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, lv, uv), sj)
ts = []
for rnd in range(5):
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(20): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 20)
ts.sort(); print("vector bounds: median %.4f ms -> %.0f GB/s" % (ts[2], 48 * n / ts[2] / 1e6))
