"""Gather-index groups (idx::Vector{Vector{Int}}): throughput of the generality path, 1e5 groups of 128 indices drawn
from a random permutation (worst case: no two neighbours in a group) and from contiguous runs given as index lists."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
ng, gs = 100_000, 128
n = ng * gs
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
lam = (torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5)
rng = np.random.default_rng(0)
perm = rng.permutation(n).reshape(ng, gs)
runs = np.arange(n).reshape(ng, gs)[rng.permutation(ng)]          # contiguous runs, groups in random order
for tag, idx in (("random permutation", perm), ("contiguous runs, shuffled order", runs)):
    h = s.GroupNormL2(lam, [row for row in idx])
    for name, psi in (("l2", s.shifted(s.shifted(h, xk), sj)), ("binf", s.shifted(s.shifted(h, xk, 1.0, chi), sj))):
        assert psi._layout.index is not None
        ts = []
        for rnd in range(3):
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(3): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 3)
        ts.sort()
        print("%-34s %-4s %.3f ms  %.0f GB/s on 32 B/element" % (tag, name, ts[1], 32 * n / ts[1] / 1e6))
