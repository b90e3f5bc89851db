import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
import arbiter
s = ge.build()
out = {}
def run(tag, x, sj, q, lam, sigma, delta, gs):
    n = x.size; ng = n // gs
    xd, sd, qd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (x, sj, q))
    h = s.GroupNormL2.uniform(torch.from_numpy(np.asarray(lam)).cuda(), gs)
    with np.errstate(all="ignore"):
        ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, gsize=gs)
    y = s.prox(s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd), qd, sigma).cpu().numpy()
    offs = np.arange(0, n + 1, gs)
    sc = arbiter.group_scale(ref, q, x, sj, offs)
    bad = np.abs(y - ref) > 1e-12 * sc
    groups = np.unique(np.flatnonzero(bad) // gs)
    print(tag, "groups above bar:", groups.size, flush=True)
    if groups.size == 0: return
    groups = groups[:100]
    idx = (groups[:, None] * gs + np.arange(gs)[None, :]).ravel()
    yq, br, root = orc.q_prox_group_l2(q, x, sj, lam, sigma, groups, gsize=gs, binf_delta=delta, details=True)
    out[tag] = dict(x=x[idx], sj=sj[idx], q=q[idx], lam=np.asarray(lam)[groups], sigma=sigma, delta=delta, gs=gs, y=y[idx], ref=ref[idx], yq=yq[idx], br=br, root=root)
# test_group_binf_many_small_groups
for gs in (1, 4, 16):
    rng = np.random.default_rng(900 + gs)
    ng = 100_000; n = ng * gs
    for ci, (sigma, delta, lscale, xscale) in enumerate(((1.0, 1.0, 1.0, 1.0), (0.3, 0.2, 0.1, 1.0), (2.0, 3.0, 3.0, 0.3), (1.0, 0.5, 1.0, 3.0), (1.0, 1.0, 30.0, 1.0))):
        x = rng.normal(size=n) * xscale; sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
        lam = rng.uniform(0.05, 2.0, size=ng) * lscale
        run("small_gs%d_c%d" % (gs, ci), x, sj, q, lam, sigma, delta, gs)
# test_group_binf_zero_groups_strong_lambda
for gs in (7, 128):
    rng = np.random.default_rng(900 + gs)
    ng = 600; n = ng * gs
    x = rng.normal(size=n).reshape(ng, gs); zero_g = rng.random(ng) < 0.6; x[zero_g] = 0.0; x = x.reshape(n)
    sj, q = rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)
    nS = np.linalg.norm(((q + x) + sj).reshape(ng, gs), axis=1)
    lam = nS * rng.choice([0.5, 1.0 - 1e-12, 1.0, 1.0 + 1e-12, 1.5, 4.0, 40.0], size=ng)
    for ci, (sigma, delta) in enumerate(((1.0, 0.01), (1.0, 0.6), (0.37, 4.0), (2.0, 100.0))):
        run("zero_gs%d_c%d" % (gs, ci), x, sj, q, lam / sigma, sigma, delta, gs)
flat = {}
for tag, d in out.items():
    for k, v in d.items(): flat[tag + "__" + k] = np.asarray(v)
np.savez_compressed("gpurun_out/binf_fail2.npz", **flat)
print("saved", len(out))
