"""psi(y) of the group operators on small uniform groups (value kept on the device: spx_ctx_set_value_target), time per call."""
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
y = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g) * 0.1
val = torch.zeros(1, dtype=torch.float64, device="cuda:0")
for binf in (False, True):
    print("psi(y), ShiftedGroupNormL2Binf" if binf else "psi(y), ShiftedGroupNormL2")
    for gs in (2, 4, 8, 16, 32, 64, 128, 1000):
        ng = n // gs; m = ng * gs
        lam = torch.rand(ng, dtype=torch.float64, device="cuda:0", generator=g) + 0.5
        H = s.GroupNormL2.uniform(lam, gs)
        psi = s.shifted(s.shifted(H, x[:m], 1.0, chi), sj[:m]) if binf else s.shifted(s.shifted(H, x[:m]), sj[:m])
        with s.device_values(val):
            for _ in range(5): psi(y[:m])
            torch.cuda.synchronize()
            best = 1e9
            for rnd in range(3):
                ms = ctypes.c_float()
                L.spx_timer_start(ctx)
                for _ in range(20): psi(y[:m])
                L.spx_timer_stop(ctx, ctypes.byref(ms))
                best = min(best, ms.value / 20 * 1e3)
        print("groups of %-5d %8.1f us per call  %5.2f TB/s (24 B per element)" % (gs, best, 24.0 * m / best / 1e6), flush=True)
