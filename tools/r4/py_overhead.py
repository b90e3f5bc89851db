"""Where the ~5 us per call of the Python mirror go (cProfile of 20 000 prox_bang / psi calls at n = 1e4; the C ABI alone: 3.9 us)."""
import cProfile, ctypes, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
os.environ.setdefault("SPX_NO_BUILD", "1")
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = 10000
g = torch.Generator(device="cuda:0").manual_seed(1)
x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
psi = s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, s.NormLinf(1.0)), sj)
for _ in range(1000): s.prox_bang(y, psi, q, 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000): s.prox_bang(y, psi, q, 1.0)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("prox_bang: %.2f us per call (host issue)" % ((t1 - t0) / 20000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(20000): s.prox_bang(y, psi, q, 1.0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
