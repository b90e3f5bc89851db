"""Host-pointer forms (spx_host_*, include/spx.h): ψ on plain numpy float64 arrays, as the reference's own tests hold
plain Vector{Float64}.  The vectors are staged through the GPU and run the same kernels, so every result must be
BIT-IDENTICAL to the device-pointer form on the same inputs; the golden fixtures and the oracle are checked as well.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.build()


def _data(n, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)


def _dev(*arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in arrs]


def _bits_equal(a, b):
    return np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))


def _both(s, make, x, sj, q, sigma=0.9):
    """prox through the host form and through the device form of the same ψ"""
    yh = s.prox(s.shifted(make(x), sj), q, sigma)
    assert isinstance(yh, np.ndarray)
    xd, sd, qd = _dev(x, sj, q)
    yd = s.prox(s.shifted(make(xd), sd), qd, sigma).cpu().numpy()
    return yh, yd


def test_config1_host_vectors(s, orc):
    # BASELINE config 1: ShiftedNormL1 prox! on n = 10^4 host vectors, nu = 1.0 (the reference's CPU-runnable case)
    x, sj, q = _data(10_000, 20250613)
    psi = s.shifted(s.shifted(s.NormL1(1.0), x), sj)
    y = s.prox(psi, q, 1.0)
    assert y is psi.sol and _bits_equal(y, orc.prox_l1(q, x, sj, 1.0, 1.0))


@pytest.mark.parametrize("n", [0, 1, 7, 1000, 100_003])
def test_host_equals_device_all_operators(s, orc, n):
    x, sj, q = _data(n, 4000 + n)
    lam = 0.7
    makers = {
        "l1": lambda v: s.shifted(s.NormL1(lam), v),
        "l0": lambda v: s.shifted(s.NormL0(lam), v),
        "lhalf": lambda v: s.shifted(s.RootNormLhalf(lam), v),
        "l1_box": lambda v: s.shifted(s.NormL1(lam), v, 0.8, s.NormLinf(1.0)),
        "l0_box": lambda v: s.shifted(s.NormL0(lam), v, 0.8, s.NormLinf(1.0)),
        "lhalf_box": lambda v: s.shifted(s.RootNormLhalf(lam), v, 0.8, s.NormLinf(1.0)),
        "l1_b2": lambda v: s.shifted(s.NormL1(lam), v, 0.8, s.NormL2(1.0)),
    }
    if n > 0:
        r = max(1, n // 10)
        makers["indball_l0"] = lambda v: s.shifted(s.IndBallL0(r), v)
        makers["indball_l0_binf"] = lambda v: s.shifted(s.IndBallL0(r), v, 0.6, s.NormLinf(1.0))
    for name, make in makers.items():
        yh, yd = _both(s, make, x, sj, q)
        assert _bits_equal(yh, yd), name
    if n > 0:
        assert _bits_equal(_both(s, makers["l1_box"], x, sj, q)[0], orc.prox_l1_box(q, x, sj, lam, 0.9, -0.8, 0.8))
        assert _bits_equal(_both(s, makers["indball_l0"], x, sj, q)[0], orc.prox_indball_l0(q, x, sj, max(1, n // 10)))


def test_host_vector_bounds_mask_and_updates(s, orc):
    n = 5000
    x, sj, q = _data(n, 77)
    rng = np.random.default_rng(5)
    lo, up = -1.0 - 0.1 * rng.random(n), 1.0 + 0.1 * rng.random(n)
    selected = sorted(rng.choice(n, size=n // 3, replace=False).tolist())
    mask = orc.mask_from_selected([i + 1 for i in selected], n)
    for op, H in (("l1_box", s.NormL1), ("l0_box", s.NormL0), ("lhalf_box", s.RootNormLhalf)):
        psi = s.shifted(s.shifted(H(0.5), x, lo, up, selected), sj)
        y = s.prox(psi, q, 0.8).copy()
        ref = getattr(orc, "prox_" + op)(q, x, sj, 0.5, 0.8, lo, up, mask=mask)
        if op == "lhalf_box":
            np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12 * 4)
        else:
            assert _bits_equal(y, ref), op
    # state updates act on the caller's arrays (src/ShiftedProximalOperators.jl:72-111)
    psi = s.shifted(s.shifted(s.NormL1(0.5), x, lo, up), sj)
    new_s = rng.uniform(-0.3, 0.3, size=n)
    s.shift_bang(psi, new_s)
    assert np.array_equal(sj, new_s)
    s.set_bounds_bang(psi, -0.5, 0.75)
    assert _bits_equal(s.prox(psi, q, 1.0), orc.prox_l1_box(q, x, sj, 0.5, 1.0, -0.5, 0.75))
    # y may alias q
    q2 = q.copy()
    s.prox_bang(q2, psi, q2, 1.0)
    assert _bits_equal(q2, orc.prox_l1_box(q, x, sj, 0.5, 1.0, -0.5, 0.75))
    # mixing host and device vectors is an error, and so is a wrong length
    with pytest.raises(TypeError):
        s.prox_bang(_dev(q)[0], psi, q, 1.0)
    with pytest.raises(IndexError):
        s.prox(psi, q[:-1].copy(), 1.0)
    with pytest.raises(ValueError):
        s.shifted(s.NormL1(1.0), x, up, lo)


def test_host_groups_iprox_objective(s, orc):
    rng = np.random.default_rng(9)
    sizes = [1, 5, 64, 129, 1000, 2, 33]
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    x, sj, q = _data(n, 31)
    lam = rng.uniform(0.2, 2.0, size=len(sizes))
    idx = [range(int(a), int(b)) for a, b in zip(off[:-1], off[1:])]
    h = s.GroupNormL2(lam.tolist(), idx)
    for make in (lambda v: s.shifted(h, v), lambda v: s.shifted(h, v, 0.9, s.NormLinf(1.0))):
        yh, yd = _both(s, make, x, sj, q)
        assert _bits_equal(yh, yd)
        xd, sd = _dev(x, sj)
        assert s.shifted(make(x), sj)(yh) == s.shifted(make(xd), sd)(_dev(yh)[0])
    hu = s.GroupNormL2.uniform(rng.uniform(0.5, 1.5, size=40).tolist(), 100)
    xu, su, qu = _data(4000, 32)
    yh, yd = _both(s, lambda v: s.shifted(hu, v, 1.0, s.NormLinf(1.0)), xu, su, qu)
    assert _bits_equal(yh, yd)
    # iprox: exact testsbox cases through host vectors (test/testsbox.jl:101-304)
    g, d = rng.normal(size=n), rng.choice([1.0, -1.0, 0.0], size=n) * rng.uniform(0.2, 3.0, size=n)
    for H, name in ((s.NormL1, "iprox_l1_box"), (s.NormL0, "iprox_l0_box")):
        psi = s.shifted(s.shifted(H(0.6), x, -0.7, 0.9), sj)
        y = s.iprox(psi, g, d)
        assert _bits_equal(y, getattr(orc, name)(g, d, x, sj, 0.6, -0.7, 0.9))
    dpos = np.abs(d) + 0.1
    psi = s.shifted(s.shifted(s.NormL1(0.6), x), sj)
    assert _bits_equal(s.iprox(psi, g, dpos), orc.iprox_l1(g, dpos, x, sj, 0.6))
    with pytest.raises(AssertionError):
        s.iprox(psi, g, d)
    # objective values: host form == device form
    xd, sd, yd_ = _dev(x, sj, q * 0.1)
    for make in (lambda v: s.shifted(s.NormL1(0.6), v), lambda v: s.shifted(s.NormL0(0.6), v, 5.0, s.NormLinf(1.0)),
                 lambda v: s.shifted(s.RootNormLhalf(0.6), v), lambda v: s.shifted(s.IndBallL0(n), v)):
        assert s.shifted(make(x), sj)(q * 0.1) == s.shifted(make(xd), sd)(yd_)


def test_host_form_aliased_call_matches_device_form(s, orc):
    """prox!(q, psi, q, sigma) on host vectors (y is q: test/test_allocs.jl:108-113).  The device form and the reference
    run ShiftedNormL1's two-pass body there, whose broadcast overwrites q first (src/shiftedNormL1.jl:47-51): the host
    form must give the same numbers -- it used to stage y and q separately and return the disjoint result."""
    import torch
    n = 4099
    rng = np.random.default_rng(17)
    x, sj, q = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)
    bits = lambda a, b: np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))
    # ShiftedNormL1, aliased: -(xk) - sj clamped against the OVERWRITTEN q, i.e. (-x) - sj itself
    qh = q.copy()
    s.prox_bang(qh, s.shifted(s.shifted(s.NormL1(1.0), x), sj), qh, 1.0)
    assert bits(qh, (-x) - sj)
    qd = torch.from_numpy(q.copy()).cuda()
    s.prox_bang(qd, s.shifted(s.shifted(s.NormL1(1.0), torch.from_numpy(x).cuda()), torch.from_numpy(sj).cuda()), qd, 1.0)
    assert bits(qh, qd.cpu().numpy())
    # Box form and top-r: aliasing does not change the result, host == oracle
    qh = q.copy()
    s.prox_bang(qh, s.shifted(s.shifted(s.NormL1(1.0), x, 1.0, s.NormLinf(1.0)), sj), qh, 1.0)
    assert bits(qh, orc.prox_l1_box(q, x, sj, 1.0, 1.0, -1.0, 1.0))
    qh = q.copy()
    s.prox_bang(qh, s.shifted(s.shifted(s.IndBallL0(77), x), sj), qh, 1.0)
    assert bits(qh, orc.prox_indball_l0(q, x, sj, 77))
