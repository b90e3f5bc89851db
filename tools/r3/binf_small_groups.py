"""ShiftedGroupNormL2 / ShiftedGroupNormL2Binf on small uniform groups: time per call at n = 1.6e7 (SPX_N) and the rate the
32 B per element correspond to.  SPX_BINF=0: the plain operator.  (The tile experiments behind the selection in
csrc/spx_group.hip run_group were done with a build that read the tile from the environment; their table is in
profiles/r03_small_groups.txt.)"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
chi = s.NormLinf(1.0)
n = int(os.environ.get("SPX_N", "16000000"))
x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
for binf in ((True, False) if "SPX_BINF" not in os.environ else (os.environ["SPX_BINF"] == "1",)):
    print("ShiftedGroupNormL2Binf" if binf else "ShiftedGroupNormL2")
    for gs in (2, 3, 4, 5, 8, 10, 12, 16, 32, 64, 100, 128):
        ng = n // gs; m = ng * gs
        lam = torch.rand(ng, dtype=torch.float64, device="cuda:0", generator=g) + 0.5
        H = s.GroupNormL2.uniform(lam, gs)
        psi = s.shifted(s.shifted(H, x[:m], 1.0, chi), sj[:m]) if binf else s.shifted(s.shifted(H, x[:m]), sj[:m])
        y = torch.empty(m, dtype=torch.float64, device="cuda:0")
        for _ in range(5): s.prox_bang(y, psi, q[:m], 1.0)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            ms = ctypes.c_float()
            L.spx_timer_start(ctx)
            for _ in range(20): s.prox_bang(y, psi, q[:m], 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms))
            best = min(best, ms.value / 20 * 1e3)
        print("groups of %-4d %8.1f us per call  %5.2f TB/s" % (gs, best, 32.0 * m / best / 1e6), flush=True)
