"""ShiftedNormL1B2 at solver-iteration sizes: time per call (HIP events) and the result against the CPU restatement."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
check = os.environ.get("SPX_CHECK", "0") == "1"
if check:
    sys.path.insert(0, os.path.join(R, "oracle"))
    import oracle as orc
for nn in [int(a) for a in os.environ.get("SPX_NS", "100000,500000,1000000,2000000,2500000,3000000,4000000,4194304").split(",")]:
    x = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(nn, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(nn, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
    for delta in (1.0, 1e9):
        psi = s.shifted(s.shifted(s.NormL1(1.0), x, delta, s.NormL2(1.0)), sj)
        for _ in range(20): s.prox_bang(y, psi, q, 1.0)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(5):
            ms = ctypes.c_float()
            L.spx_timer_start(ctx)
            for _ in range(100): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms))
            best = min(best, ms.value / 100 * 1e3)
        note = ""
        if check:
            ref = orc.prox_l1_b2(q.cpu().numpy(), x.cpu().numpy(), sj.cpu().numpy(), 1.0, 1.0, delta, 1.0)
            err = np.max(np.abs(y.cpu().numpy() - ref)) / max(np.linalg.norm(ref), np.linalg.norm(x.cpu().numpy()))
            note = " | max err / scale %.2e" % err
        print("n=%-8d Delta=%-6g %7.2f us per call%s" % (nn, delta, best, note), flush=True)
