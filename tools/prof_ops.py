"""Runs each operator a few times at full size (for rocprofv3 --kernel-trace --stats)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
n = int(os.environ.get("SPX_N", "100000000"))
which = os.environ.get("SPX_OPS", "l1box,l1,l0,l0box,vecbounds,lhalf,lhalfbox,indball,indball_ties,small,iprox,objective,proxvalue,b2,f32,team,group,binf").split(",")
g = torch.Generator(device=dev).manual_seed(1)   # (only where the values do not steer the kernel: bounds, d)
chi = s.NormLinf(1.0)
def synth(m, stream, kind, scale=1.0):
    """spx_synth_fill(seed + 1000, stream, kind): the bits bench.py's other_operators time (round 4: the group workloads here
    came from torch's Philox before, and the data-dependent Binf kernels could not be tied to the bench line)"""
    t = torch.empty(m, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), m, 20250613 + 1000, stream, kind, scale))
    return t
def vecs(m, first=0):
    return synth(m, first, 1), synth(m, first + 1, 0), synth(m, first + 2, 1)
xk, sj, q = vecs(n); y = torch.empty_like(q)
if "l1box" in which:
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "l1" in which:
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "l0" in which:
    psi = s.shifted(s.shifted(s.NormL0(1.0), xk), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "vecbounds" in which:
    lv = -1.0 - 0.1 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    uv = 1.0 + 0.1 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, lv, uv), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
    del lv, uv, psi
if "small" in which:   # solver-iteration sizes: the one-launch forms (v / xk in LDS at 4e6, registers below)
    for nn in (6_000_000, 4_000_000, 1_000_000, 100_000, 10_000):
        if nn <= n:
            for psi in (s.shifted(s.shifted(s.IndBallL0(max(1, nn // 100)), xk[:nn], 1.0, chi), sj[:nn]),
                        s.shifted(s.shifted(s.IndBallL0(max(1, nn // 2)), xk[:nn], 1.0, chi), sj[:nn]),
                        s.shifted(s.shifted(s.NormL1(1.0), xk[:nn], 1.0, s.NormL2(1.0)), sj[:nn])):
                for _ in range(5): s.prox_bang(y[:nn], psi, q[:nn], 1.0)
if "l0box" in which:
    psi = s.shifted(s.shifted(s.NormL0(1.0), xk, 1.0, chi), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "lhalf" in which:
    psi = s.shifted(s.shifted(s.RootNormLhalf(1.0), xk), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "iprox" in which:
    d = torch.rand(n, dtype=torch.float64, device=dev, generator=g) + 0.5
    for H in (s.NormL1, s.NormL0):
        psi = s.shifted(s.shifted(H(1.0), xk, 1.0, chi), sj)
        for _ in range(5): s.iprox_bang(y, psi, q, d, check=False)
    del d
if "objective" in which:
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
    for _ in range(5): psi(y)
if "f32" in which:
    x32, s32, q32 = (t.to(torch.float32) for t in (xk, sj, q)); y32 = torch.empty_like(q32)
    psi = s.shifted(s.shifted(s.NormL1(1.0), x32, 1.0, chi), s32)
    for _ in range(5): s.prox_bang(y32, psi, q32, 1.0)
    del x32, s32, q32, y32
if "b2" in which:
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "indball" in which:
    psi = s.shifted(s.shifted(s.IndBallL0(n // 100), xk, 1.0, chi), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "indball_ties" in which:   # (round 3) tie mode of the top-r pipeline: q on a 1/4 lattice, xk = sj = 0 in buffers of their own
    z1, z2 = torch.zeros_like(xk), torch.zeros_like(sj)
    q4 = torch.round(q * 4.0) / 4.0
    for r in (n // 100, n // 2):
        psi = s.shifted(s.shifted(s.IndBallL0(r), z1, 1.0, chi), z2)
        for _ in range(5): s.prox_bang(y, psi, q4, 1.0)
    del z1, z2, q4
if "proxvalue" in which:
    psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, chi), sj)
    vout = torch.zeros(1, dtype=torch.float64, device=dev)
    with s.device_values(vout):
        for _ in range(5): s.prox_value_bang(y, psi, q, 1.0)
if "lhalfbox" in which:
    psi = s.shifted(s.shifted(s.RootNormLhalf(1.0), xk, 1.0, chi), sj)
    for _ in range(5): s.prox_bang(y, psi, q, 1.0)
if "team" in which:   # (round 4) one group over the vector / seven ragged groups: the team form + the chunked psi(y)
    cuts = sorted(set([0, n] + [int(n * f) for f in (0.09, 0.22, 0.31, 0.55, 0.6, 0.93)]))
    vout = torch.zeros(1, dtype=torch.float64, device=dev)
    for hg in (s.GroupNormL2([0.5 * n ** 0.5]),
               s.GroupNormL2([0.5 * (b - a) ** 0.5 for a, b in zip(cuts, cuts[1:])], [range(a, b) for a, b in zip(cuts, cuts[1:])])):
        for psi in (s.shifted(s.shifted(hg, xk), sj), s.shifted(s.shifted(hg, xk, 1.0, chi), sj)):
            for _ in range(5): s.prox_bang(y, psi, q, 1.0)
            with s.device_values(vout):
                for _ in range(5): psi(y)
if "group" in which or "binf" in which:
    ng = n // 100; m = ng * 128
    del xk, sj, q, y
    xk, sj, q = vecs(m, 6); y = torch.empty_like(q)
    lam = synth(ng, 3, 0) + 1.0
    h = s.GroupNormL2.uniform(lam, 128)
    if "group" in which:
        psi = s.shifted(s.shifted(h, xk), sj)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
    if "binf" in which:
        psi = s.shifted(s.shifted(h, xk, 1.0, chi), sj)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
    if "group" in which or "binf" in which:   # small groups (round 3: tiles of one or two lanes per group)
        ng8 = min(m, n) // 8; m8 = ng8 * 8
        lam8 = synth(ng8, 4, 0) + 1.0
        h8 = s.GroupNormL2.uniform(lam8, 8)
        for psi in (s.shifted(s.shifted(h8, xk[:m8]), sj[:m8]), s.shifted(s.shifted(h8, xk[:m8], 1.0, chi), sj[:m8])):
            for _ in range(5): s.prox_bang(y[:m8], psi, q[:m8], 1.0)
torch.cuda.synchronize()
