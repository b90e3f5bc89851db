"""GPU parity of the TEAM form (csrc/spx_group_team.hip): ShiftedGroupNormL2(Binf) prox! on few, large contiguous groups --
first of all ONE group over the whole vector, which is what the reference's `shifted(NormL2(lambda), xk)` builds
(/root/reference/src/shiftedGroupNormL2.jl:34-35, src/shiftedGroupNormL2Binf.jl:48-49, src/groupNormL2.jl:30-31) -- and of
psi(y) on such groups (the chunked form in csrc/spx_objective.hip).  Everything through the C ABI against the CPU oracle;
bar: |dy_i| <= 1e-12 max(|y_i|, |xk_i + sj_i|, ||S_group||) (tests/test_gpu_parity.py), the binary128 arbiter above it.
The full-size cases (one group over n = 1e8 / 1e7) are in tests/test_gpu_fullsize.py, the reference's NormL2 test flow at
n = 1e6 in tests/test_gpu_reference_suite.py."""
import ctypes

import numpy as np
import pytest

import arbiter  # tests/arbiter.py

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.build()


def _data(n, seed, quant=None, xscale=1.0):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=n) * xscale
    sj = rng.uniform(-0.5, 0.5, size=n)
    q = rng.normal(size=n)
    if quant:
        x, sj, q = (np.round(v * quant) / quant for v in (x, sj, q))
    return x, sj, q


def _dev(*arrs, offset=0):
    import torch
    out = []
    for a in arrs:
        t = torch.zeros(a.shape[0] + 2, dtype=torch.float64, device="cuda:0")
        t[offset:offset + a.shape[0]] = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        out.append(t[offset:offset + a.shape[0]])
    return out


def _tuning(s, key, value):
    L = s._lib.load()
    s._lib.check(L.spx_ctx_set_tuning(s.context("cuda:0"), key, value))


def _run(s, orc, x, sj, q, offsets, lam, sigma, delta, binf, offset=0, devs=None, ref=None):
    """prox! through the mirrored API on contiguous groups `offsets`; returns (y, ref, psi)."""
    xd, sd, qd = devs if devs is not None else _dev(x, sj, q, offset=offset)
    groups = [range(int(a), int(b)) for a, b in zip(offsets[:-1], offsets[1:])]
    h = s.GroupNormL2(list(map(float, lam)), groups)
    off = np.asarray(offsets, dtype=np.int64)
    if binf:
        psi = s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd)
        if ref is None:
            ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, offsets=off)
    else:
        psi = s.shifted(s.shifted(h, xd), sd)
        if ref is None:
            ref = orc.prox_group_l2(q, x, sj, lam, sigma, offsets=off)
    y = s.prox(psi, qd, sigma).cpu().numpy()
    return y, ref, psi


def _check(orc, y, ref, q, x, sj, lam, sigma, offsets, delta, what):
    assert np.array_equal(np.isnan(y), np.isnan(ref)), what
    arbiter.check_group(orc, y, ref, q, x, sj, np.asarray(lam, dtype=np.float64), sigma, offsets, delta=delta, what=what, max_arbitrated=8)


# one group over the vector: sizes on both sides of every switch -- the one-workgroup LDS kernel (<= 2048 plain / 4096 Binf), one
# workgroup of the team form on chip (9216 elements), several (2048 elements per workgroup at least), the whole chip on chip
# (256 x 9216 = 2 359 296), streamed beyond; odd sizes leave an element off the 16-byte pair grid
@pytest.mark.parametrize("n", [4097, 9216, 9217, 20_001, 300_000, 2_359_296, 2_359_298, 6_000_001])
@pytest.mark.parametrize("binf", [False, True])
def test_one_group_over_the_vector(s, orc, n, binf):
    x, sj, q = _data(n, 4000 + n % 1000)
    lam = [0.5 * n ** 0.5]  # sigma lambda ~ ||S|| / 3: the group is shrunk, not zeroed
    sigma, delta = 0.7, 1.0
    y, ref, _ = _run(s, orc, x, sj, q, [0, n], lam, sigma, delta, binf)
    _check(orc, y, ref, q, x, sj, lam, sigma, [0, n], delta if binf else None, "one group n=%d" % n)
    assert np.count_nonzero(y + (x + sj)) > 0  # (not the trivial all-zero prox)


# the regimes of the Binf root find on a large group, on chip (n = 100 000) and streamed (n = 2 500 000), through the fast path
# and through the generic body (tuning key 14 = 0): every decision of binf_root must come out the same
SCENARIOS = {
    "plain":          dict(),
    "small_delta":    dict(delta=0.05),
    "large_delta":    dict(delta=50.0),                 # nothing active: the all-inactive piece
    "strong_lambda":  dict(lamscale=5.0),               # sigma lambda > ||S||: zeros (reversed bracket, entries outside the trust region)
    "x0":             dict(xscale=0.0),                 # x = 0: lmax = ||S||, the root at the bracket's end
    "inside_tr":      dict(xscale=0.3, delta=2.0),      # every |x_i| <= Delta: no sign certificate for froot(lmin) -- the full pass
    "lattice":        dict(quant=4, delta=0.5),         # |x_i| = Delta exactly for many i: the reference's literal evaluation
    "tiny_sigma":     dict(sigma=1e-3),
    "barely_nonzero": dict(lamscale=1.4142),            # sigma lambda just below ||S||: the root next to the pole of step(n)
}


@pytest.mark.parametrize("scenario", sorted(SCENARIOS))
@pytest.mark.parametrize("n", [100_000, 2_500_000])
def test_binf_regimes_on_a_large_group(s, orc, scenario, n):
    p = SCENARIOS[scenario]
    x, sj, q = _data(n, 77, quant=p.get("quant"), xscale=p.get("xscale", 1.0))
    sigma, delta = p.get("sigma", 0.7), p.get("delta", 1.0)
    nS = np.linalg.norm((q + x) + sj)
    lam = [p.get("lamscale", 0.35) * nS / sigma]
    devs = _dev(x, sj, q)
    ys = {}
    ref = None
    for fast in (1, 0):
        _tuning(s, 14, fast)
        try:
            y, ref, _ = _run(s, orc, x, sj, q, [0, n], lam, sigma, delta, True, devs=devs, ref=ref)
        finally:
            _tuning(s, 14, 1)
        _check(orc, y, ref, q, x, sj, lam, sigma, [0, n], delta, "%s n=%d fast=%d" % (scenario, n, fast))
        ys[fast] = y
    # the two paths agree on WHICH result this is (zeros or not); their values may differ in the last bits
    assert np.array_equal(ys[0] == -(x + sj), ys[1] == -(x + sj))


def _ragged_offsets(n, cuts):
    return [0] + [int(n * f) for f in cuts] + [n]


@pytest.mark.parametrize("binf", [False, True])
@pytest.mark.parametrize("layout", ["seven", "small_and_large", "many_large"])
def test_ragged_layouts_plan_on_the_device(s, orc, binf, layout):
    # CSR offsets: the large groups (more than the LDS-resident kernel holds: > 2048 plain / > 4096 Binf) go to teams sized by their share of the elements (k_team_plan), the
    # others to the one-workgroup-per-group kernel, which skips the large ones
    n = 3_000_000
    if layout == "seven":
        offsets = _ragged_offsets(n, (0.09, 0.22, 0.31, 0.55, 0.6, 0.93))
    elif layout == "small_and_large":
        offsets = _ragged_offsets(n, (0.001, 0.0011, 0.0011001, 0.31, 0.310001, 0.6, 0.93))
    else:  # 40 large groups of uneven size + a few small ones between them
        rng = np.random.default_rng(5)
        cuts = np.sort(rng.uniform(0.0, 1.0, size=39))
        offsets = sorted(set(_ragged_offsets(n, cuts) + [int(n * c) + 7 for c in cuts[:5]]))
    x, sj, q = _data(n, 31)
    lam = [0.4 * max(b - a, 1) ** 0.5 for a, b in zip(offsets[:-1], offsets[1:])]
    y, ref, _ = _run(s, orc, x, sj, q, offsets, lam, 1.0, 1.0, binf)
    _check(orc, y, ref, q, x, sj, lam, 1.0, offsets, 1.0 if binf else None, layout)


@pytest.mark.parametrize("binf", [False, True])
def test_few_workgroups_resident(s, orc, binf):
    # tuning key 8 pretends only 4 workgroups are resident: teams of one workgroup taking several groups in turn (uniform
    # groups and a device-side plan with more large groups than workgroups), and one team of 4 on a streamed group
    _tuning(s, 8, 4)
    try:
        n = 6 * 50_000
        x, sj, q = _data(n, 41)
        lam = [30.0] * 6
        offsets = list(range(0, n + 1, 50_000))
        y, ref, _ = _run(s, orc, x, sj, q, offsets, lam, 1.0, 1.0, binf)          # uniform: 6 groups on 4 workgroups
        _check(orc, y, ref, q, x, sj, lam, 1.0, offsets, 1.0 if binf else None, "uniform loop")
        offsets = [0, 50_000, 110_000, 150_000, 150_010, 220_000, 260_000, n]     # ragged: 6 large + 1 small
        lam = [30.0] * 7
        y, ref, _ = _run(s, orc, x, sj, q, offsets, lam, 1.0, 1.0, binf)
        _check(orc, y, ref, q, x, sj, lam, 1.0, offsets, 1.0 if binf else None, "plan loop")
        y, ref, _ = _run(s, orc, x, sj, q, [0, n], [150.0], 1.0, 1.0, binf)       # one team of 4, streamed
        _check(orc, y, ref, q, x, sj, [150.0], 1.0, [0, n], 1.0 if binf else None, "team of 4")
    finally:
        _tuning(s, 8, 0)


@pytest.mark.parametrize("binf", [False, True])
@pytest.mark.parametrize("n", [50_001, 3_000_001])
def test_views_and_aliasing(s, orc, binf, n):
    import torch
    x, sj, q = _data(n, 9)
    lam, sigma, delta = [0.4 * n ** 0.5], 1.0, 1.0
    off = np.array([0, n], dtype=np.int64)
    ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, offsets=off) if binf else orc.prox_group_l2(q, x, sj, lam, sigma, offsets=off)
    h = s.GroupNormL2(lam)

    def psi_of(xd, sd):
        return s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd) if binf else s.shifted(s.shifted(h, xd), sd)

    # every vector 8 bytes off a 16-byte boundary: the pair grid starts at element 1
    xd, sd, qd = _dev(x, sj, q, offset=1)
    yd = torch.zeros(n + 2, dtype=torch.float64, device="cuda:0")[1:n + 1]
    assert xd.data_ptr() % 16 == 8
    _check(orc, s.prox_bang(yd, psi_of(xd, sd), qd, sigma).cpu().numpy(), ref, q, x, sj, lam, sigma, [0, n], delta if binf else None, "all odd")
    # mixed alignment: 8-byte loads
    xd2, = _dev(x, offset=0)
    _check(orc, s.prox_bang(yd, psi_of(xd2, sd), qd, sigma).cpu().numpy(), ref, q, x, sj, lam, sigma, [0, n], delta if binf else None, "mixed")
    # y === q  (/root/reference/test/test_allocs.jl:108)
    qa = qd.clone()
    s.prox_bang(qa, psi_of(xd, sd), qa, sigma)
    _check(orc, qa.cpu().numpy(), ref, q, x, sj, lam, sigma, [0, n], delta if binf else None, "aliased")
    # run to run: the same bits (fixed summation order in every workgroup and across the team)
    y1 = s.prox_bang(yd, psi_of(xd, sd), qd, sigma).clone()
    y2 = s.prox_bang(yd, psi_of(xd, sd), qd, sigma)
    assert torch.equal(y1.view(torch.int64), y2.view(torch.int64))


def test_objective_on_structured_data_against_an_exact_sum(s, orc):
    """psi(y) of ONE group over 2e6 lattice-valued elements: the Float64 oracle adds the squares one after the other, and on data
    with few distinct values its rounding errors line up (found by tools/r4/fuzz_team.py: up to 1.6e-11 relative, the GPU's
    chunked fixed-order sum 2e-16).  Adjudicated like the prox values (tests/arbiter.py): against the exact sum of the squares
    (math.fsum, square root in extended precision) the GPU must be within 1e-12 + the oracle's own error."""
    import math
    import torch
    n = 2_000_000
    x, sj, q = _data(n, 21, quant=4)
    lam = 0.7
    off = np.array([0, n], dtype=np.int64)
    xd, sd, qd = _dev(x, sj, q)
    psi = s.shifted(s.shifted(s.NormL2(lam), xd), sd)
    y = s.prox(psi, qd, 0.9)
    yh = y.cpu().numpy()
    v = psi(y)
    vr = orc.obj_group_l2(yh, x, sj, [lam], offsets=off)
    w = (x + sj) + yh
    exact = float(np.longdouble(lam) * np.sqrt(np.longdouble(math.fsum((w * w).tolist()))))
    assert abs(v - exact) <= 1e-12 * exact + abs(vr - exact), (v, vr, exact)
    assert abs(v - exact) <= 1e-14 * exact, (v, exact)   # (the GPU's own error: a few ulps)


@pytest.mark.parametrize("layout", ["one", "uniform", "ragged"])
@pytest.mark.parametrize("binf", [False, True])
def test_objective_on_large_groups(s, orc, layout, binf):
    # psi(y) = sum_g lambda_g ||(xk + sj + y)[g]|| (+ IndBallLinf(1.1 Delta)(sj + y)): the chunked form
    n = 1_500_000
    x, sj, q = _data(n, 13)
    offsets = {"one": [0, n], "uniform": list(range(0, n + 1, 100_000)),
               "ragged": [1000] + [int(n * f) for f in (0.09, 0.22, 0.2200001, 0.55, 0.93)] + [n - 77]}[layout]
    lam = np.random.default_rng(1).uniform(0.5, 1.5, size=len(offsets) - 1)
    delta = 0.9
    xd, sd, qd = _dev(x, sj, q, offset=(1 if layout == "ragged" else 0))
    groups = [range(a, b) for a, b in zip(offsets[:-1], offsets[1:])]
    h = s.GroupNormL2(lam.tolist(), groups)
    psi = s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd) if binf else s.shifted(s.shifted(h, xd), sd)
    off = np.asarray(offsets, dtype=np.int64)
    import torch
    for scale in (0.0, 0.3, 3.0):   # 3.0: outside 1.1 Delta in the Binf form
        yv = np.clip(q, -1.0, 1.0) * scale * 0.3
        if binf and scale == 0.3 and layout == "ragged":
            yv[5] = 10.0 - sj[5]  # an index in NO group, outside the trust region: the indicator covers every index
        yd = torch.from_numpy(yv).to("cuda:0")
        v = psi(yd)
        vr = orc.obj_group_l2(yv, x, sj, lam, offsets=off, delta=(delta if binf else None))
        assert (v == vr) or abs(v - vr) <= 1e-12 * abs(vr), (layout, binf, scale, v, vr)
