"""Per-call cost at small n (BASELINE config 0: ShiftedNormL1 prox!, n = 1e4): mirrored Python API vs the raw C entry."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
for n in (10_000, 1_000_000):
    g = torch.Generator(device=dev).manual_seed(1)
    xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
    for name, psi in (("ShiftedNormL1", s.shifted(s.shifted(s.NormL1(1.0), xk), sj)),
                      ("ShiftedNormL1Box", s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj)),
                      ("ShiftedIndBallL0BInf", s.shifted(s.shifted(s.IndBallL0(max(1, n // 100)), xk, 1.0, s.NormLinf(1.0)), sj)),
                      ("ShiftedGroupNormL2Binf", s.shifted(s.shifted(s.GroupNormL2.uniform([1.0] * (n // 100), 100), xk, 1.0, s.NormLinf(1.0)), sj)),
                      ("ShiftedNormL1B2", s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj))):
        reps = 2000
        for _ in range(20): s.prox_bang(y, psi, q, 1.0)
        s.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): s.prox_bang(y, psi, q, 1.0)
        t_issue = time.perf_counter() - t0
        s.synchronize(); t_all = time.perf_counter() - t0
        print("n=%-8d %-24s python API: %.1f us/call to issue, %.1f us/call incl. completion" % (n, name, t_issue / reps * 1e6, t_all / reps * 1e6), flush=True)
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormLinf(1.0)), sj)
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): psi(y)
    print("n=%-8d %-24s python API: %.1f us/call (synchronous: returns the value)" % (n, "psi(y) ShiftedNormL1Box", (time.perf_counter() - t0) / reps * 1e6), flush=True)
    yp, qp, xp, sp = (ctypes.c_void_p(t.data_ptr()) for t in (y, q, xk, sj))
    f = L.spx_prox_l1
    s.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f(ctx, yp, qp, xp, sp, n, 1.0, 1.0)
    t_issue = time.perf_counter() - t0
    s.synchronize(); t_all = time.perf_counter() - t0
    print("n=%-8d %-24s raw ctypes:  %.1f us/call to issue, %.1f us/call incl. completion" % (n, "spx_prox_l1", t_issue / reps * 1e6, t_all / reps * 1e6), flush=True)
