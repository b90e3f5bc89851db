"""ShiftedGroupNormL2Binf on small / odd uniform groups on bench.py's inputs (spx_synth_fill, seed + 1000): time per call and the
rate of the 32 B per element, for A/B builds (SPX_LIB_NAME).  SPX_N (default 1e8), SPX_GS (comma list, default 2,4,8,16,100,128),
SPX_BINF=0: the plain ShiftedGroupNormL2.  SPX_DUMP=path: the first 2^18 elements of y per group size as .npy (to compare two builds bit for bit / to 1e-12)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import torch
import __graft_entry__ as ge
os.environ.setdefault("SPX_NO_BUILD", "1")
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
dev = torch.device("cuda:0")
SEED = 20250613 + 1000
n = int(float(os.environ.get("SPX_N", "1e8")))
sizes = [int(v) for v in os.environ.get("SPX_GS", "2,4,8,16,100,128").split(",")]


def synth(m, stream, kind):
    t = torch.empty(m, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), m, SEED, stream, kind, ctypes.c_double(1.0)))
    return t


x, sj, q = synth(n, 6, 1), synth(n, 7, 0), synth(n, 8, 1)
chi = s.NormLinf(1.0)
y = torch.empty(n, dtype=torch.float64, device=dev)
print("lib", os.environ.get("SPX_LIB_NAME", "libspx.so"), "n", n, flush=True)
for gs in sizes:
    ng = n // gs; m = ng * gs
    lam = synth(ng, 4, 0) + 1.0
    H = s.GroupNormL2.uniform(lam, gs)
    psi = s.shifted(s.shifted(H, x[:m], 1.0, chi), sj[:m]) if os.environ.get("SPX_BINF", "1") == "1" else s.shifted(s.shifted(H, x[:m]), sj[:m])
    for _ in range(3): s.prox_bang(y[:m], psi, q[:m], 1.0)
    torch.cuda.synchronize()
    best = 1e9; allr = []
    for rnd in range(5):
        ms = ctypes.c_float()
        L.spx_timer_start(ctx)
        for _ in range(10): s.prox_bang(y[:m], psi, q[:m], 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        allr.append(ms.value / 10 * 1e3)
    allr.sort()
    print("groups of %-4d median %8.1f us  best %8.1f us  %5.2f TB/s (median)" % (gs, allr[2], allr[0], 32.0 * m / allr[2] / 1e6), flush=True)
    if "SPX_DUMP" in os.environ:
        k = min(m, (1 << 18) // gs * gs)
        np.save("%s_gs%d.npy" % (os.environ["SPX_DUMP"], gs), y[:k].cpu().numpy())
