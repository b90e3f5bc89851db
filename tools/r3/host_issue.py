"""Where does the host time of one top-r call go at n = 4e6?  Host issue time per call (40 calls per round, synchronised
between rounds) through the Python operator layer and through ctypes directly, with a one-launch operator beside it."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
nn = int(os.environ.get("SPX_N", "4000000"))
x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
chi = s.NormLinf(1.0)
psi = s.shifted(s.shifted(s.IndBallL0(nn // 100), x, 1.0, chi), sj)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def direct():
    return L.spx_prox_indball_l0_binf(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(nn), ctypes.c_int64(nn // 100), ctypes.c_double(1.0))
def direct_l1():
    return L.spx_prox_l1(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(nn), ctypes.c_double(1.0), ctypes.c_double(1.0))
def layer(): s.prox_bang(y, psi, q, 1.0)
for name, fn in (("top-r through the operator layer", layer), ("top-r ctypes direct", direct), ("L1 ctypes direct", direct_l1)):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9; tot = 1e9
    for rnd in range(8):
        t0 = time.perf_counter()
        for _ in range(40): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / 40 * 1e6); tot = min(tot, (t2 - t0) / 40 * 1e6)
    print("%-36s host issue %6.2f us per call | issue + drain %6.2f us per call" % (name, best, tot), flush=True)
