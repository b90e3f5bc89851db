"""one GroupNormL2Binf configuration, a few calls (for rocprofv3 --kernel-trace): python3 tools/exp/binf_one.py xscale delta lscale"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build()
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
gs = int(os.environ.get("SPX_GS", "128")); ng = 128_000_000 // gs; n = ng * gs
xs, delta, ls = float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g) * xs; sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
lam = (torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5) * ls
psi = s.shifted(s.shifted(s.GroupNormL2.uniform(lam, gs), xk, delta, s.NormLinf(1.0)), sj)
for _ in range(3): s.prox_bang(y, psi, q, 1.0)
s.synchronize()
