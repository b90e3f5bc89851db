"""The reference's own test flow for the operators of the prox! path (test/runtests.jl, v0.2.2), re-expressed against
the host mirror on plain HOST vectors (numpy float64 = the reference's Vector{Float64}): construction, field
aliasing, psi(y) identities, golden prox values, shift!, set_radius!, second shifts.  Everything numeric runs on the
GPU through the spx_host_* forms of the C ABI.

Differences from the reference that these tests state instead of hide:
  * `ψ(y) == h(x + y)` is exact (==) in the reference because both sides sum sequentially on the CPU; here psi(y) is a
    GPU reduction, so sums compare to VALUE_RTOL; counts (NormL0) and +-Inf are exact.
  * "test different types": the strided view `view(y, 1:2:10)` is taken (view + packed copy); Float32 host arrays raise
    TypeError (the Float32 forms exist for device vectors: tests/test_gpu_f32.py runs that block in Float32).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
VALUE_RTOL = 1e-13


@pytest.fixture(scope="module")
def s():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.build()


def _h(orc, kind, x, lam):  # h(x) of NormL1 / NormL0 / RootNormLhalf  [ext: ProximalOperators.jl, src/rootNormLhalf.jl:27-29]
    return orc.obj_plain(kind, np.zeros_like(x), x, np.zeros_like(x), lam)


def _close(a, b):
    return a == b or abs(a - b) <= VALUE_RTOL * max(abs(a), abs(b))


UNBOXED = [("NormL0", "ShiftedNormL0", "l0"), ("NormL1", "ShiftedNormL1", "l1"),
           ("RootNormLhalf", "ShiftedRootNormLhalf", "lhalf")]


@pytest.mark.parametrize("op,shifted_op,kind", UNBOXED)
def test_operators_without_trust_region(s, orc, op, shifted_op, kind):  # runtests.jl:157-214
    rng = np.random.default_rng(0)
    h = getattr(s, op)(1.2)
    x = np.ones(3)
    psi = s.shifted(h, x)
    assert type(psi).__name__ == shifted_op
    assert np.all(psi.sj == 0) and psi.xk is x          # xk is borrowed, not copied
    assert psi.λ == 1.2
    assert _close(psi(np.zeros(3)), _h(orc, kind, x, 1.2))
    y = rng.random(3)
    assert _close(psi(y), _h(orc, kind, x + y, 1.2))
    # shift update
    x0 = x.copy()
    s.shift_bang(psi, y)
    assert np.all(psi.sj == 0) and np.all(psi.xk == y)
    # shift a shifted operator (the reference builds it from a ψ whose xk now equals y)
    sv = np.ones(3) / 2
    phi = s.shifted(psi, sv)
    assert phi.sj is sv and phi.xk is psi.xk and phi.shifted_twice
    assert _close(phi(np.zeros(3)), _h(orc, kind, psi.xk + sv, 1.2))
    t = rng.random(3)
    assert _close(phi(t), _h(orc, kind, psi.xk + sv + t, 1.2))
    # "different types" (runtests.jl:196-209): x = view(y, 1:2:10), psi = shifted(h, x), psi(zeros(5)) == h(x).  The strided view
    # is kept as view + packed copy; Float32 HOST arrays have no host-pointer form (device Float32: tests/test_gpu_f32.py)
    with pytest.raises(TypeError):
        s.shifted(getattr(s, op)(1.2), np.ones(3, dtype=np.float32))
    y10 = rng.random(10)
    hv = getattr(s, op)(1.2)
    pv = s.shifted(hv, y10[::2])
    assert _close(pv(np.zeros(5)), _h(orc, kind, np.ascontiguousarray(y10[::2]), 1.2))
    assert pv(np.zeros(5)) == hv(y10[::2])
    del x0


def test_norml2_as_one_group(s, orc):  # runtests.jl:216-277
    rng = np.random.default_rng(1)
    lam = rng.random()
    x = np.ones(6)
    nu = rng.random()
    q = rng.normal(size=6)
    psi = s.shifted(s.NormL2(lam), x)
    assert type(psi).__name__ == "ShiftedGroupNormL2"
    assert np.all(psi.sj == 0) and psi.xk is x and list(psi.λ) == [lam]
    assert _close(psi(np.zeros(6)), lam * np.linalg.norm(x))
    y = rng.random(6)
    assert _close(psi(y), lam * np.linalg.norm(x + y))
    ypsi = np.empty(6)
    s.prox_bang(ypsi, psi, q, nu)
    v = q + x                                            # NormL2 prox [ext]: max(1 - nu lam / ||v||, 0) v
    yp = max(1 - nu * lam / np.linalg.norm(v), 0.0) * v
    assert np.sqrt(np.sum((ypsi - (yp - x)) ** 2)) <= 1e-11
    s.shift_bang(psi, y)
    assert np.all(psi.sj == 0) and np.all(psi.xk == y)
    sv = np.ones(6) / 2
    phi = s.shifted(psi, sv)
    assert phi.sj is sv and phi.xk is psi.xk
    assert _close(phi(np.zeros(6)), lam * np.linalg.norm(psi.xk + sv))


@pytest.mark.parametrize("binf", [False, True])
def test_norml2_as_one_group_1e6_on_the_device(s, orc, binf):
    """The reference's NormL2 flow (/root/reference/test/runtests.jl:213-284 plain, :555-606 with the l-infinity trust region) at
    n = 1e6 on device vectors: `shifted(NormL2(lambda), x)` is ONE group over the vector -- the team form of csrc/spx_group_team.hip
    (on chip at this size) and the chunked psi(y)."""
    import torch
    n = 1_000_000
    rng = np.random.default_rng(11)
    lam, nu, Delta = float(rng.random()) * 40.0, float(rng.random()), 0.01
    xh = np.ones(n)
    qh = rng.normal(size=n)
    x = torch.from_numpy(xh).cuda()
    q = torch.from_numpy(qh).cuda()
    chi = s.NormLinf(1.0)
    psi = s.shifted(s.NormL2(lam), x, Delta, chi) if binf else s.shifted(s.NormL2(lam), x)
    assert type(psi).__name__ == ("ShiftedGroupNormL2Binf" if binf else "ShiftedGroupNormL2")
    assert bool((psi.sj == 0).all()) and psi.xk is x and list(psi.λ) == [lam]
    h = lambda z: lam * np.linalg.norm(z)
    assert _close(psi(torch.zeros_like(x)), h(xh))                      # psi(0) == h(x)
    yh = rng.random(n)
    if binf:
        yh *= Delta / np.max(np.abs(yh)) / 2                             # inside the trust region (:579-581)
    y = torch.from_numpy(yh).cuda()
    assert _close(psi(y), h(xh + yh))
    if binf:
        assert psi(3 * y) == np.inf                                      # outside the trust region
    ypsi = s.prox_bang(torch.empty_like(x), psi, q, nu).cpu().numpy()
    zero = np.zeros(n)
    off = np.array([0, n], dtype=np.int64)
    if binf:
        ref = orc.prox_group_l2_binf(qh, xh, zero, [lam], nu, Delta, offsets=off)
        assert np.max(np.abs(ypsi)) <= Delta * (1 + 1e-12)               # chi(s) <= Delta (:603)
    else:
        v = qh + xh                                                      # NormL2 prox [ext]: max(1 - nu lam / ||v||, 0) v  (:247-251)
        yp = max(1 - nu * lam / np.linalg.norm(v), 0.0) * v
        assert np.sqrt(np.sum((ypsi - (yp - xh)) ** 2)) <= 1e-11 * np.sqrt(n)
        ref = orc.prox_group_l2(qh, xh, zero, [lam], nu, offsets=off)
    scale = np.maximum(np.maximum(np.abs(ref), np.abs(xh)), np.linalg.norm(qh + xh))
    assert float(np.max(np.abs(ypsi - ref) / scale)) <= 1e-12
    s.shift_bang(psi, y)                                                 # shift update (:253-256): xk is the caller's array
    assert bool((psi.sj == 0).all()) and bool((psi.xk == y).all()) and psi.xk is x
    sv = torch.full_like(x, 0.5) * (Delta if binf else 1.0)
    phi = s.shifted(psi, sv)                                             # shift a shifted operator (:258-264)
    assert phi.sj is sv and phi.xk is psi.xk
    assert _close(phi(torch.zeros_like(x)), h(yh + sv.cpu().numpy()))


def test_groupnorml2_index_vectors(s, orc):  # runtests.jl:286-330: v = [collect(1:3), collect(4:6)]
    rng = np.random.default_rng(2)
    v = [[0, 1, 2], [3, 4, 5]]
    lam = rng.random(2)
    h = s.GroupNormL2(lam.tolist(), v)
    x = np.ones(6)
    nu = rng.random()
    q = rng.normal(size=6)
    psi = s.shifted(h, x)
    assert type(psi).__name__ == "ShiftedGroupNormL2" and np.all(psi.sj == 0) and psi.xk is x
    hval = lambda z: sum(l * np.linalg.norm(z[g]) for l, g in zip(lam, v))
    assert _close(psi(np.zeros(6)), hval(x))
    y = rng.random(6)
    assert _close(psi(y), hval(x + y))
    ypsi = np.empty(6)
    s.prox_bang(ypsi, psi, q, nu)
    yp = np.empty(6)
    for l, g in zip(lam, v):                             # per-group NormL2 prox of q + x
        z = (q + x)[g]
        yp[g] = max(1 - nu * l / np.linalg.norm(z), 0.0) * z
    assert np.sqrt(np.sum((ypsi - (yp - x)) ** 2)) <= 1e-11


def test_indballl0(s, orc):  # runtests.jl:362-414
    rng = np.random.default_rng(3)
    h = s.IndBallL0(1)
    x = np.ones(3)
    psi = s.shifted(h, x)
    assert type(psi).__name__ == "ShiftedIndBallL0" and psi.xk is x and psi.r == 1
    assert psi(np.zeros(3)) == np.inf                    # h(x): three nonzeros > r
    assert s.shifted(s.IndBallL0(3), x)(np.zeros(3)) == 0.0
    y = rng.random(3)
    assert psi(y) == orc.obj_indball_l0(y, x, np.zeros(3), 1)
    s.shift_bang(psi, y)
    assert np.all(psi.xk == y)
    sv = np.ones(3) / 2
    phi = s.shifted(psi, sv)
    assert phi.sj is sv and phi.xk is psi.xk
    assert phi(-(psi.xk + sv)) == 0.0                    # xk + sj + t = 0: inside every l0 ball


TR = [("NormL0", "NormLinf", "ShiftedNormL0Box", "l0"), ("NormL1", "NormLinf", "ShiftedNormL1Box", "l1"),
      ("NormL1", "NormL2", "ShiftedNormL1B2", "l1"), ("RootNormLhalf", "NormLinf", "ShiftedRootNormLhalfBox", "lhalf")]


@pytest.mark.parametrize("op,tr,shifted_op,kind", TR)
def test_operators_with_trust_region(s, orc, kats, op, tr, shifted_op, kind):  # runtests.jl:417-556
    rng = np.random.default_rng(4)
    chi_norm = (lambda z: np.max(np.abs(z))) if tr == "NormLinf" else (lambda z: np.linalg.norm(z))
    chi = getattr(s, tr)(1.0)
    n = 5
    h = getattr(s, op)(1.0)
    x = np.ones(n)
    delta = 0.01
    psi = s.shifted(h, x, delta, chi)
    assert type(psi).__name__ == shifted_op
    if shifted_op == "ShiftedNormL1B2":
        assert psi.Δ == delta
    assert np.all(psi.sj == 0) and psi.xk is x and psi.λ == 1.0
    assert _close(psi(np.zeros(n)), _h(orc, kind, x, 1.0))
    y = rng.random(n)
    y *= delta / chi_norm(y) / 2
    assert _close(psi(y), _h(orc, kind, x + y, 1.0))     # y inside the trust region
    assert psi(3 * y) == np.inf                          # y outside the trust region
    # prox golden (runtests.jl:449-494)
    k = kats["box_golden"]
    q = np.array(k["q"])
    sol = s.prox(psi, q, k["sigma"])
    assert sol is psi.sol
    np.testing.assert_allclose(sol, k["expected"][shifted_op], rtol=k["rtol"], atol=0)
    assert chi_norm(sol) <= delta * (1 + 1e-12)
    # shift update
    s.shift_bang(psi, y)
    assert np.all(psi.sj == 0) and np.all(psi.xk == y)
    # radius / bounds update
    d2 = 1.1
    s.set_radius_bang(psi, d2)
    if shifted_op == "ShiftedNormL1B2":
        assert psi.Δ == d2
    else:
        assert psi.l == -d2 and psi.u == d2
    # shift a shifted operator
    sv = np.ones(n)
    sv /= 2 * chi_norm(sv)
    phi = s.shifted(psi, sv)
    assert phi.sj is sv and phi.xk is psi.xk
    assert _close(phi(np.zeros(n)), _h(orc, kind, psi.xk + sv, 1.0))
    t = rng.random(n)
    t *= d2 / chi_norm(t) / 2
    assert _close(phi(t), _h(orc, kind, psi.xk + sv + t, 1.0))   # inside
    assert phi(3 * t) == np.inf                                   # outside


def test_groupnorml2_binf_flow(s, orc, kats):  # runtests.jl:558-650
    rng = np.random.default_rng(5)
    k = kats["group_l2_binf_single"]
    x = np.array(k["x"])
    n = x.size
    lam = k["lambda"][0]
    delta = k["delta"]
    psi = s.shifted(s.NormL2(lam), x, delta, s.NormLinf(1.0))
    assert type(psi).__name__ == "ShiftedGroupNormL2Binf" and psi.Δ == delta
    assert np.all(psi.sj == 0) and psi.xk is x and list(psi.λ) == [lam]
    assert _close(psi(np.zeros(n)), lam * np.linalg.norm(x))
    y = rng.random(n)
    y *= delta / np.max(np.abs(y)) / 2
    assert _close(psi(y), lam * np.linalg.norm(x + y))
    assert psi(3 * y) == np.inf
    sol = s.prox(psi, np.array(k["q"]), k["sigma"])
    np.testing.assert_allclose(sol, k["expected"], rtol=k["rtol"], atol=0)
    assert np.max(np.abs(sol)) <= delta * (1 + 1e-8)
    s.set_radius_bang(psi, 1.1)
    assert psi.Δ == 1.1


def test_unshifted_rootnormlhalf(s, orc, kats):  # runtests.jl:113-126, src/rootNormLhalf.jl:27-51
    k = kats["rootnormlhalf_unshifted"]
    q = np.array(k["q"])
    h = s.RootNormLhalf(k["lambda"])
    y = np.empty_like(q)
    ret = s.prox_bang(y, h, q, k["nu"])                    # prox!(y, h, q, ν) returns h(y)
    assert float(np.sum((y - np.array(k["expected"])) ** 2)) <= k["sumsq_tol"]
    yo, vo = orc.rootnormlhalf_prox(q, k["lambda"], k["nu"])
    assert np.max(np.abs(y - yo)) <= 1e-12 and abs(ret - vo) <= 1e-12 * abs(vo)
    assert _close(h(q), k["lambda"] * float(np.sum(np.sqrt(np.abs(q)))))      # (f::RootNormLhalf)(x)
    # device vectors take the same path
    import torch
    qd = torch.from_numpy(q).cuda()
    yd = torch.empty_like(qd)
    retd = s.prox_bang(yd, h, qd, k["nu"])
    assert np.array_equal(yd.cpu().numpy(), y) and retd == ret
    with pytest.raises(ValueError):
        s.RootNormLhalf(-1.0)


def test_unshifted_groupnorml2(s, orc):  # runtests.jl:128-155, src/groupNormL2.jl:33-58
    rng = np.random.default_rng(6)
    x = rng.random(6)
    v = [range(0, 3), [3, 4, 5]]                           # `v = [1:3, collect(4:6)]`
    lam = rng.random(2)
    nu = float(rng.random())
    h = s.GroupNormL2(lam.tolist(), v)
    y = np.empty_like(x)
    ysum = s.prox_bang(y, h, x, nu)
    ytrue = np.empty_like(x)
    ysumt = 0.0
    for l, g in zip(lam, v):                               # per group NormL2 prox [ext]: returns λ‖prox‖
        xg = x[list(g)]
        ng = np.linalg.norm(xg)
        ytrue[list(g)] = max(1 - nu * l / ng, 0.0) * xg
        ysumt += l * ng
    assert float(np.sum((y - ytrue) ** 2)) <= 1e-11
    assert abs(ysum - ysumt) <= 1e-12 * ysumt               # the reference returns Σ λ ‖x_g‖ of the INPUT (:52)
    assert abs(h(x) - ysumt) <= 1e-11                       # norm value (runtests.jl:148-154)
    # the other value types evaluate too
    assert _close(s.NormL1(0.5)(x), 0.5 * float(np.sum(np.abs(x))))
    assert s.NormL0(2.0)(np.array([0.0, 1.0, -0.0, 3.0])) == 4.0
    assert s.IndBallL0(1)(np.array([0.0, 1.0, 0.0])) == 0.0 and s.IndBallL0(1)(np.array([2.0, 1.0, 0.0])) == np.inf
    assert _close(s.NormL2(0.3)(x), 0.3 * float(np.linalg.norm(x)))
