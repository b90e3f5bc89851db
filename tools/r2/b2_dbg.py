import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build()
rng = np.random.default_rng(sum(map(ord, "normal")))
n = 2_300_001
x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
xd, sd, qd = (torch.from_numpy(v).cuda() for v in (x, sj, q))
lam, delta = 0.01, 1e-3
ref = orc.prox_l1_b2(q, x, sj, lam, 1.0, delta, 1.0)
y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, 1.0).cpu().numpy()
torch.cuda.synchronize()
print("err", np.max(np.abs(y - ref)), "norm ref", np.linalg.norm(ref + sj), "norm y", np.linalg.norm(y + sj))
bad = np.flatnonzero(np.abs(y - ref) > 1e-9)
print("bad count", bad.size, "first", bad[:5], "last", bad[-5:], "y", y[bad[:3]], "ref", ref[bad[:3]])
L = s._lib.load(); L.spx_ctx_set_tuning(s.context("cuda:0"), 7, 0)
y0 = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, 1.0).cpu().numpy()
print("host loop err", np.max(np.abs(y0 - ref)))
