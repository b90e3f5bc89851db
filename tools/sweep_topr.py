"""Top-r (ShiftedIndBallL0BInf.prox!) over r and data scale at n = 1e8: ms per call of the sample-predicted path, checked
bit for bit against the full-vector radix select (spx_ctx_set_tuning key 2 = 0) on the same inputs.  A time near the
full-vector figure means the prediction fell back."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(1); chi = s.NormLinf(1.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
q0 = torch.randn(n, dtype=torch.float64, device=dev, generator=g); y = torch.empty_like(q0)
def timed(psi, q, reps):
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    for _ in range(reps): s.prox_bang(y, psi, q, 1.0)
    L.spx_timer_stop(ctx, ctypes.byref(ms)); return ms.value / reps
bad = 0
for kind in ("normal", "x1.37", "x0.69", "cauchy", "lattice64"):
    if kind == "normal": q = q0
    elif kind == "x1.37": q = q0 * 1.37
    elif kind == "x0.69": q = q0 * 0.69
    elif kind == "cauchy": q = torch.tan((torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5) * 3.141592653589793)
    else: q = torch.round(q0 * 64) / 64
    for r in (1, 37, 1000, 100_000, n // 100, n // 10, n // 2, (9 * n) // 10, n - 100_000, n - 1000):
        psi = s.shifted(s.shifted(s.IndBallL0(r), xk, 1.0, chi), sj)
        L.spx_ctx_set_tuning(s.context("cuda:0"), 2, 0); s.prox_bang(y, psi, q, 1.0); ref = y.clone(); t_full = timed(psi, q, 2)
        L.spx_ctx_set_tuning(s.context("cuda:0"), 2, 1); s.prox_bang(y, psi, q, 1.0); t = timed(psi, q, 10)
        ok = torch.equal(y, ref); bad += 0 if ok else 1
        print("%-10s r=%-10d fast %.4f ms  full-vector %.4f ms  %s%s" % (kind, r, t, t_full, "ok" if ok else "MISMATCH", "  (fell back)" if t > 0.8 * t_full + 0.3 else ""), flush=True)
    del q
print("mismatches", bad)
sys.exit(1 if bad else 0)
