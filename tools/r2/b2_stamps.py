"""SPX_LIB_NAME=libspx_prof.so (built with -DSPX_B2_PROFILE): time line of workgroup 0 inside k_b2_coop."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); ctx = s.context("cuda:0")
raw = ctypes.CDLL(s._lib.LIB_PATH)
for n in [int(a) for a in sys.argv[1:]]:
    for delta in (1.0, 1e9):
        g = torch.Generator(device="cuda:0").manual_seed(99)
        xk = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
        q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
        psi = s.shifted(s.shifted(s.NormL1(1.0), xk, delta, s.NormL2(1.0)), sj)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)(); cnt = ctypes.c_int()
        raw.spx_debug_b2_stamps(buf, ctypes.byref(cnt))
        st = list(buf)[:cnt.value]
        print("n %8d Delta %-6g:" % (n, delta), "  ".join("+%.1f" % ((v - st[0]) / 100.0) for v in st), "us", flush=True)
