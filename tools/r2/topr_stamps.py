import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
raw = ctypes.CDLL(s._lib.LIB_PATH)
if "SPX_KEY10" in os.environ: L.spx_ctx_set_tuning(ctx, 10, int(os.environ["SPX_KEY10"]))  # (round 4: the front kernel's sample, tools/r4/topr_front_ab.py)
for n in [int(a) for a in sys.argv[1:]]:
    g = torch.Generator(device="cuda:0").manual_seed(1)
    x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    if os.environ.get("SPX_LATTICE") == "1":   # (round 4: tie mode of the front kernel) a 1/4 lattice, xk = sj = 0
        q = torch.round(q * 4) / 4; x.zero_(); sj.zero_()
    psi = s.shifted(s.shifted(s.IndBallL0(max(1, n // int(os.environ.get("SPX_RDIV", "100")))), x, 1.0, s.NormLinf(1.0)), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    raw.spx_debug_sel_stamps(buf)
    st = list(buf)
    def seg(name, ks):
        ks = [k for k in ks if st[k]]
        if len(ks) < 2: return
        print("  %-6s" % name, "  ".join("%d:+%.1fus" % (k, (st[k] - st[ks[0]]) / 100.0) for k in ks))
    print("n =", n)
    seg("front", list(range(0, 16))); seg("tail", list(range(16, 23))); seg("coop", list(range(31, 64)))
