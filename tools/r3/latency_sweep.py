"""Time per call (HIP events over 100 back-to-back calls, best of 5) of every prox operator family across sizes, with the
bytes each call must move (algorithmic) and the rate that corresponds to.  Looks for cliffs between the forms of an operator."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
chi = s.NormLinf(1.0)
def timed(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(5):
        ms = ctypes.c_float()
        L.spx_timer_start(ctx)
        for _ in range(reps): fn()
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        best = min(best, ms.value / reps * 1e3)
    return best
sizes = [int(a) for a in os.environ.get("SPX_NS", "10000,100000,1000000,2000000,3000000,4000000,6000000,8000000,16000000").split(",")]
print("%-26s" % "us per call (TB/s)" + "".join("%16d" % n for n in sizes))
rows = []
def run(name, bytes_per_el, make):
    out = []
    for nn in sizes:
        x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
        q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
        fn = make(nn, x, sj, q, y)
        t = timed(fn)
        out.append("%9.1f (%4.2f)" % (t, bytes_per_el * nn / (t * 1e-6) / 1e12))
        del x, sj, q, y, fn
    print("%-26s" % name + "".join("%16s" % o for o in out), flush=True)
run("L1 + Binf (box)", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, chi), sj)))
run("L0 + Binf (box)", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.NormL0(1.0), x, 1.0, chi), sj)))
run("Lhalf", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.RootNormLhalf(1.0), x), sj)))
run("top-r r=n/100 + Binf", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.IndBallL0(max(1, nn // 100)), x, 1.0, chi), sj)))
run("top-r r=n/2", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.IndBallL0(max(1, nn // 2)), x), sj)))
run("L1 + B2 (active)", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, s.NormL2(1.0)), sj)))
run("L1 + B2 (inactive)", 32, lambda nn, x, sj, q, y: (lambda psi: (lambda: s.prox_bang(y, psi, q, 1.0)))(s.shifted(s.shifted(s.NormL1(1.0), x, 1e9, s.NormL2(1.0)), sj)))
def grp(binf, gs):
    def make(nn, x, sj, q, y):
        ng = nn // gs
        lam = torch.rand(ng, dtype=torch.float64, device="cuda:0", generator=g) + 0.5
        m = ng * gs
        H = s.GroupNormL2.uniform(lam, gs)
        psi = s.shifted(s.shifted(H, x[:m], 1.0, chi), sj[:m]) if binf else s.shifted(s.shifted(H, x[:m]), sj[:m])
        return lambda: s.prox_bang(y[:m], psi, q[:m], 1.0)
    return make
run("GroupL2 (groups of 100)", 32, grp(False, 100))
run("GroupL2 + Binf (of 100)", 32, grp(True, 100))
run("GroupL2 + Binf (of 8)", 32, grp(True, 8))
def ipr(nn, x, sj, q, y):
    d = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) + 0.5
    psi = s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, chi), sj)
    return lambda: s.iprox_bang(y, psi, q, d, check=False)
run("iprox L1 + Binf", 40, ipr)
