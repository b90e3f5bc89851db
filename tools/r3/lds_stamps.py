"""Phase time stamps of workgroup 0 in k_sel_lds (build with -DSPX_SEL_PROFILE as libspx_prof.so)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
raw = ctypes.CDLL(s._lib.LIB_PATH)
for n in [int(a) for a in sys.argv[1:]]:
    g = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.IndBallL0(max(1, n // int(os.environ.get("SPX_RDIV", "100")))), x, 1.0, s.NormLinf(1.0)), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    raw.spx_debug_sel_stamps(buf)
    st = list(buf); t0 = st[31]
    names = {31: "start", 32: "loaded", 63: "stored"}
    for p in range(4):
        names[23 + p] = "p%d digits" % p; names[33 + 3 * p] = "p%d flush issued" % p; names[27 + p] = "p%d flush drained" % p
        names[34 + 3 * p] = "p%d met" % p; names[35 + 3 * p] = "p%d scanned" % p
    ev = sorted((st[k], k) for k in names if st[k] >= t0 and st[k] - t0 < 100000)
    print("n = %d:" % n, " | ".join("%s +%.1f" % (names[k], (v - t0) / 100.0) for v, k in ev))
