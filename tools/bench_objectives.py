"""psi(y) (objective value incl. the read-back) for every operator family at full size: wall ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build()
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(3); chi = s.NormLinf(1.0)
ng, gs = 1_000_000, 128; n = ng * gs
xk = torch.randn(n, dtype=torch.float64, device=dev, generator=g); sj = torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 0.5
y = torch.randn(n, dtype=torch.float64, device=dev, generator=g) * 0.3
lam = torch.rand(ng, dtype=torch.float64, device=dev, generator=g) + 0.5
cases = {
    "ShiftedNormL1": s.shifted(s.shifted(s.NormL1(1.0), xk), sj),
    "ShiftedNormL0": s.shifted(s.shifted(s.NormL0(1.0), xk), sj),
    "ShiftedRootNormLhalf": s.shifted(s.shifted(s.RootNormLhalf(1.0), xk), sj),
    "ShiftedNormL1Box": s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj),
    "ShiftedNormL0Box": s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, chi), sj),
    "ShiftedRootNormLhalfBox": s.shifted(s.shifted(s.RootNormLhalf(1.0), xk, 1.0, chi), sj),
    "ShiftedIndBallL0": s.shifted(s.shifted(s.IndBallL0(n // 100), xk), sj),
    "ShiftedIndBallL0BInf": s.shifted(s.shifted(s.IndBallL0(n // 100), xk, 1.0, chi), sj),
    "ShiftedGroupNormL2": s.shifted(s.shifted(s.GroupNormL2.uniform(lam, gs), xk), sj),
    "ShiftedGroupNormL2Binf": s.shifted(s.shifted(s.GroupNormL2.uniform(lam, gs), xk, 1.0, chi), sj),
    "ShiftedNormL1B2": s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj),
}
for name, psi in cases.items():
    v = psi(y); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): v = psi(y)
    t = (time.perf_counter() - t0) / 10 * 1e3
    print("psi(y) %-26s %.4f ms  -> %.0f GB/s on 24 B/element   value %.6g" % (name, t, 24 * n / t / 1e6, v), flush=True)
