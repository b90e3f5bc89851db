"""Where does the sample-predicted top-r path start to pay?  ms per call over n, sample-predicted path (library built
with -DSPX_SEL_FAST_MIN_LOG2=17: `SPX_LIB_NAME=libspx_t.so csrc/build.sh -DSPX_SEL_FAST_MIN_LOG2=17`, then
`SPX_LIB_NAME=libspx_t.so SPX_NO_BUILD=1 python tools/exp/topr_threshold.py`) vs the full-vector path (tuning key 2 = 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
for lg in (17, 18, 19, 20, 21, 22, 23):
    n = (1 << lg) + 12
    xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q)
    for r in (max(1, n // 1000), n // 100, n // 4):
        psi = s.shifted(s.shifted(s.IndBallL0(r), xk, 1.0, chi), sj)
        res = {}
        for mode in (0, 1):
            L.spx_ctx_set_tuning(s.context("cuda:0"), 2, mode)
            s.prox_bang(y, psi, q, 1.0); ref = y.clone() if mode == 0 else ref
            ts = []
            for rnd in range(5):
                ms = ctypes.c_float(); L.spx_timer_start(ctx)
                for _ in range(50): s.prox_bang(y, psi, q, 1.0)
                L.spx_timer_stop(ctx, ctypes.byref(ms)); ts.append(ms.value / 50)
            res[mode] = sorted(ts)[2]
            assert torch.equal(y, ref)
        print("n = 2^%d r = %-8d full-vector %.1f us   sample-predicted %.1f us" % (lg, r, res[0] * 1e3, res[1] * 1e3), flush=True)
L.spx_ctx_set_tuning(s.context("cuda:0"), 2, 1)
