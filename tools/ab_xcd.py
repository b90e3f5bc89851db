"""A/B of the blockIdx -> tile mapping of the LDS-staged separable kernel (spx_ctx_set_tuning key 5), interleaved in-process."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
n = 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
ref = s.prox_bang(torch.empty_like(q), psi, q, 1.0).clone()
res = {0: [], 1: []}
for rnd in range(8):
    for mode in (0, 1):
        L.spx_ctx_set_tuning(s.context("cuda:0"), 5, mode)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(20): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms)); res[mode].append(ms.value / 20)
        assert torch.equal(y, ref)
L.spx_ctx_set_tuning(s.context("cuda:0"), 5, 0)
for mode in (0, 1):
    t = sorted(res[mode][1:]); med = t[len(t) // 2]
    print("%-28s median %.4f ms min %.4f  -> %.0f GB/s" % ("tile = workgroup id" if mode == 0 else "XCD-contiguous tile ranges", med, t[0], 32 * n / med / 1e6))
