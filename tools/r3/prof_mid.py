"""The one-launch on-chip forms at n = 4e6 (top-r with v in LDS, ShiftedNormL1B2 with xk in LDS), the register form at
n = 1e6, and the group operators on groups of 8, a few calls each: for rocprofv3 (kernel stats, PMC traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
chi = s.NormLinf(1.0)
n = 4_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
for r in (n // 100, n // 2):
    psi = s.shifted(s.shifted(s.IndBallL0(r), xk, 1.0, chi), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)
for _ in range(5): s.prox_bang(y, psi, q, 1.0)
m = 1_000_000
psi = s.shifted(s.shifted(s.IndBallL0(m // 100), xk[:m], 1.0, chi), sj[:m])
for _ in range(5): s.prox_bang(y[:m], psi, q[:m], 1.0)
ng = n // 8
lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
H = s.GroupNormL2.uniform(lam, 8)
for psi in (s.shifted(s.shifted(H, xk), sj), s.shifted(s.shifted(H, xk, 1.0, chi), sj)):
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
val = torch.zeros(1, dtype=torch.float64, device=dev)
with s.device_values(val):
    for _ in range(5): psi(y)
torch.cuda.synchronize()
