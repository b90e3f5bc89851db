"""Pins the CPU oracle against every known-answer vector the reference's tests hold for the
prox! hot path (tests/golden/reference_kats.json, transcribed from /root/reference/test), plus an
independent brute-force objective minimiser.  CPU only."""
import numpy as np
import pytest

BOX_FN = {"ShiftedNormL0Box": "prox_l0_box", "ShiftedNormL1Box": "prox_l1_box",
          "ShiftedRootNormLhalfBox": "prox_lhalf_box"}


@pytest.mark.parametrize("op", list(BOX_FN))
def test_box_golden_runtests(orc, kats, op):
    # test/runtests.jl:449-494
    k = kats["box_golden"]
    y = getattr(orc, BOX_FN[op])(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], -k["delta"], k["delta"])
    np.testing.assert_allclose(y, k["expected"][op], rtol=k["rtol"], atol=0)


@pytest.mark.parametrize("op", list(BOX_FN))
def test_testsbox_enumerated_cases(orc, kats, op):
    # test/testsbox.jl:13-97: psi = shifted(h, x, l, u); omega = shifted(psi, s); prox(omega, q, sigma)
    t = kats["testsbox"]
    c = t[op]
    for qi, xi, lam, sol in zip(c["q"], c["x"], c["lambda"], c["sol"]):
        y = getattr(orc, BOX_FN[op])([qi], [xi], [t["s"]], lam, t["sigma"], t["l"], t["u"])
        assert abs(y[0] - sol) <= t["atol"], (op, qi, xi, lam, y[0], sol)
        # vector-bounds form gives the same answer
        y2 = getattr(orc, BOX_FN[op])([qi], [xi], [t["s"]], lam, t["sigma"], [t["l"]], [t["u"]])
        assert y2[0] == y[0]


def test_testsbox_lhalf_exact_values(orc, kats):
    # exact values behind the rounded 1.6054 / 2.702 of testsbox.jl:77
    t = kats["testsbox"]
    c = t["ShiftedRootNormLhalfBox"]
    y4 = orc.prox_lhalf_box([c["q"][3]], [c["x"][3]], [t["s"]], c["lambda"][3], 1.0, 0.0, 3.0)[0]
    y8 = orc.prox_lhalf_box([c["q"][7]], [c["x"][7]], [t["s"]], c["lambda"][7], 1.0, 0.0, 3.0)[0]
    assert abs(y4 - 1.6053779404795958) < 1e-13
    assert abs(y8 - 2.7015158583813426) < 1e-13


def test_rootnormlhalf_unshifted_golden(orc, kats):
    # test/runtests.jl:113-126
    k = kats["rootnormlhalf_unshifted"]
    y, val = orc.rootnormlhalf_prox(k["q"], k["lambda"], k["nu"])
    assert np.sum((y - np.array(k["expected"])) ** 2) <= k["sumsq_tol"]
    assert abs(val - k["lambda"] * np.sum(np.sqrt(np.abs(y)))) < 1e-14


@pytest.mark.parametrize("name", ["group_l2_binf_single", "group_l2_binf_two"])
def test_group_l2_binf_golden(orc, kats, name):
    # test/runtests.jl:587-606, 658-705
    k = kats[name]
    y = orc.prox_group_l2_binf(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], k["delta"], offsets=k["offsets"])
    np.testing.assert_allclose(y, k["expected"], rtol=k["rtol"], atol=0)
    assert np.max(np.abs(y)) <= k["delta"] * (1 + 1e-8)


@pytest.mark.parametrize("name", ["group_l2_binf_single", "group_l2_binf_two"])
def test_group_index_set_form_matches_ranges_and_golden(orc, kats, name):
    # the reference test builds its groups as index VECTORS (`v = [collect(1:3), collect(4:6)]`, runtests.jl:658):
    # the index-set restatement must give the golden too, and equal the range restatement bit for bit
    k = kats[name]
    off = k["offsets"]
    groups = [list(range(a, b)) for a, b in zip(off[:-1], off[1:])]
    y = orc.prox_group_l2_idx(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], groups, delta=k["delta"])
    np.testing.assert_allclose(y, k["expected"], rtol=k["rtol"], atol=0)
    assert np.array_equal(y, orc.prox_group_l2_binf(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], k["delta"],
                                                    offsets=off))
    y2 = orc.prox_group_l2_idx(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], groups)
    assert np.array_equal(y2, orc.prox_group_l2(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], offsets=off))


def test_group_index_sets_sequential_semantics(orc):
    # overlapping groups: the later group's value stays; an index in no group keeps y on entry (minus the shift for
    # ShiftedGroupNormL2, src/shiftedGroupNormL2.jl:77; untouched for Binf, src/shiftedGroupNormL2Binf.jl:116)
    rng = np.random.default_rng(3)
    n = 9
    q, x, s = rng.normal(size=n), rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n)
    y0 = rng.normal(size=n)
    groups = [[0, 1, 2, 3], [3, 4, 5]]
    lam = [0.3, 0.4]
    y = orc.prox_group_l2_idx(q, x, s, lam, 0.5, groups, y0=y0)
    S = (q + x) + s
    second = _norml2_prox(S[[3, 4, 5]], 0.4, 0.5)
    first = _norml2_prox(S[[0, 1, 2, 3]], 0.3, 0.5)
    np.testing.assert_allclose(y[[3, 4, 5]], second - (x + s)[[3, 4, 5]], rtol=1e-14)
    np.testing.assert_allclose(y[[0, 1, 2]], first[:3] - (x + s)[[0, 1, 2]], rtol=1e-14)
    assert np.array_equal(y[6:], y0[6:] - (x[6:] + s[6:]))
    yb = orc.prox_group_l2_idx(q, x, s, lam, 0.5, groups, delta=0.7, y0=y0)
    assert np.array_equal(yb[6:], y0[6:])


def _norml2_prox(x, lam, gamma):
    # ProximalOperators.NormL2 prox [ext]: max(1 - gamma*lam/||x||, 0) * x
    nx = np.linalg.norm(x)
    return np.zeros_like(x) if nx == 0 else max(1 - gamma * lam / nx, 0.0) * x


def test_group_l2_equals_per_group_norml2(orc):
    # test/runtests.jl:244-251, 318-329
    rng = np.random.default_rng(7)
    for trial in range(20):
        x, q = rng.random(6), rng.random(6)
        lam, nu = rng.random(2), rng.random()
        y = orc.prox_group_l2(q, x, np.zeros(6), lam, nu, offsets=[0, 3, 6])
        yp = np.concatenate([_norml2_prox((q + x)[:3], lam[0], nu), _norml2_prox((q + x)[3:], lam[1], nu)])
        assert np.linalg.norm(y - (yp - x)) <= 1e-11
        # NormL2 -> single group [:]
        y1 = orc.prox_group_l2(q, x, np.zeros(6), [lam[0]], nu, offsets=[0, 6])
        assert np.linalg.norm(y1 - (_norml2_prox(q + x, lam[0], nu) - x)) <= 1e-11


def test_l1box_equals_softthreshold_then_clamp(orc):
    # test/runtests.jl:814-843
    rng = np.random.default_rng(11)
    for trial in range(50):
        n = 4
        delta = 2 * rng.random()
        q = 2 * (rng.random(n) - 0.5)
        nu = rng.random()
        xk = rng.random(n) - 0.5
        st = lambda v: np.sign(v) * np.maximum(np.abs(v) - nu, 0.0)
        p1 = np.minimum(np.maximum(st(xk + q), xk - delta), xk + delta) - xk
        p2 = orc.prox_l1_box(q, xk, np.zeros(n), 1.0, nu, -delta, delta)
        np.testing.assert_allclose(p1, p2, rtol=1.5e-8, atol=1e-15)
        sj = rng.random(n) - 0.5
        p1 = np.minimum(np.maximum(st(xk + sj + q), xk - delta), xk + delta) - (xk + sj)
        p2 = orc.prox_l1_box(q, xk, sj, 1.0, nu, -delta, delta)
        np.testing.assert_allclose(p1, p2, rtol=1.5e-8, atol=1e-15)


@pytest.mark.parametrize("op", list(BOX_FN))
def test_partial_prox(orc, op):
    # test/partial_prox.jl:14-39: selected = 1:2:n -> selected entries equal the full prox,
    # the others equal prox_zero = clamp(q, l - s, u - s)
    rng = np.random.default_rng(3)
    n = 5
    x, s, q = rng.random(n), rng.random(n), rng.random(n) - 0.5
    if op == "ShiftedRootNormLhalfBox":
        l, u = -0.5, 0.5
    else:
        l, u = np.zeros(n), np.ones(n)
    fn = getattr(orc, BOX_FN[op])
    y = fn(q, x, s, 3.14, 1.0, l, u)
    mask = orc.mask_from_selected(range(1, n + 1, 2), n)
    z = fn(q, x, s, 3.14, 1.0, l, u, mask=mask)
    p = np.minimum(np.maximum(q, l - s), u - s)
    for i in range(n):
        assert z[i] == (y[i] if mask[i] else p[i])


def test_derived_kats(orc, kats):
    d = kats["derived"]
    A = d["setA_unboxed"]
    np.testing.assert_array_equal(orc.prox_l1(A["q"], A["x"], A["s"], A["lambda"], A["sigma"]), A["ShiftedNormL1"])
    np.testing.assert_array_equal(orc.prox_l0(A["q"], A["x"], A["s"], A["lambda"], A["sigma"]), A["ShiftedNormL0"])
    np.testing.assert_allclose(orc.prox_lhalf(A["q"], A["x"], A["s"], A["lambda"], A["sigma"]),
                               A["ShiftedRootNormLhalf"], rtol=1e-12)
    # consistency link to the reference goldens: clamping the unboxed answers to +-0.01 reproduces
    # the boxed golden vectors (runtests.jl:460-490)
    g = kats["box_golden"]
    for un, bx in (("ShiftedNormL1", "ShiftedNormL1Box"), ("ShiftedNormL0", "ShiftedNormL0Box"),
                   ("ShiftedRootNormLhalf", "ShiftedRootNormLhalfBox")):
        np.testing.assert_allclose(np.clip(A[un], -0.01, 0.01), g["expected"][bx], rtol=g["rtol"])
    B = d["setB"]
    y = orc.prox_l1(B["q"], B["x"], B["s"], B["lambda"], B["sigma"])
    np.testing.assert_array_equal(y, B["ShiftedNormL1"])
    assert np.signbit(y[6])  # -0.0 as in Julia
    y = orc.prox_l0(B["q"], B["x"], B["s"], B["lambda"], B["sigma"])
    np.testing.assert_array_equal(y, B["ShiftedNormL0"])
    np.testing.assert_allclose(orc.prox_lhalf(B["q"], B["x"], B["s"], B["lambda"], B["sigma"]),
                               B["ShiftedRootNormLhalf"], rtol=1e-12, atol=0)
    np.testing.assert_array_equal(orc.prox_indball_l0(B["q"], B["x"], B["s"], 3), B["ShiftedIndBallL0_r3"])
    np.testing.assert_array_equal(orc.prox_indball_l0_binf(B["q"], B["x"], B["s"], 3, 0.6),
                                  B["ShiftedIndBallL0BInf_r3_delta0.6"])
    T = d["tiebreak"]
    for r in (2, 3, 4):
        np.testing.assert_array_equal(orc.prox_indball_l0(T["q"], T["x"], T["s"], r), T["r%d" % r])


# ---- independent second oracle: brute-force minimisation of the prox objective over the box ----
def _brute(obj, lo, hi, extra):
    grid = np.linspace(lo, hi, 200001)
    cand = np.concatenate([grid, [c for c in extra if lo <= c <= hi]])
    return np.min(obj(cand))


@pytest.mark.parametrize("op", list(BOX_FN))
def test_box_prox_is_objective_minimiser(orc, op):
    rng = np.random.default_rng(5)
    h = {"ShiftedNormL0Box": lambda v: (v != 0).astype(float),
         "ShiftedNormL1Box": np.abs,
         "ShiftedRootNormLhalfBox": lambda v: np.sqrt(np.abs(v))}[op]
    for trial in range(300):
        x, q = rng.normal(), rng.normal()
        s = rng.uniform(-0.5, 0.5)
        lam, sigma = rng.uniform(0.1, 2.0), rng.uniform(0.1, 2.0)
        l, u = -rng.uniform(0.2, 1.5), rng.uniform(0.2, 1.5)
        if trial % 3 == 0:  # make 0 in v-space often infeasible / box tight
            l, u = sorted((rng.normal(), rng.normal()))
            s = rng.uniform(l, u)
        t = getattr(orc, BOX_FN[op])([q], [x], [s], lam, sigma, l, u)[0]
        obj = lambda tt: (tt - q) ** 2 / (2 * sigma) + lam * h(x + s + tt)
        assert l - s - 1e-12 <= t <= u - s + 1e-12
        best = _brute(obj, l - s, u - s, [-(x + s), q])
        assert obj(np.array([t]))[0] <= best + 1e-9 * max(1.0, abs(best)), (op, trial, t)


def test_unboxed_prox_is_objective_minimiser(orc):
    rng = np.random.default_rng(6)
    for name, h in (("prox_l1", np.abs), ("prox_l0", lambda v: (v != 0).astype(float)),
                    ("prox_lhalf", lambda v: np.sqrt(np.abs(v)))):
        for trial in range(200):
            x, q, s = rng.normal(), rng.normal(), rng.uniform(-0.5, 0.5)
            lam, sigma = rng.uniform(0.1, 2.0), rng.uniform(0.1, 2.0)
            t = getattr(orc, name)([q], [x], [s], lam, sigma)[0]
            obj = lambda tt: (tt - q) ** 2 / (2 * sigma) + lam * h(x + s + tt)
            best = _brute(obj, q - 6, q + 6, [-(x + s), q])
            assert obj(np.array([t]))[0] <= best + 1e-9 * max(1.0, abs(best)), (name, trial)


def test_indball_l0_matches_numpy_stable_sort(orc):
    rng = np.random.default_rng(9)
    n = 2000
    x, s = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n)
    q = np.round(rng.normal(size=n) * 8) / 8  # many equal magnitudes after adding quantised x below
    x = np.round(x * 8) / 8
    s = np.round(s * 8) / 8
    for r in (0, 1, 17, 500, n - 1, n, n + 5):
        v = (x + s) + q
        order = np.argsort(-np.abs(v), kind="stable")
        keep = np.zeros(n, bool)
        keep[order[:r]] = True
        expect = np.where(keep, v, 0.0) - (x + s)
        np.testing.assert_array_equal(orc.prox_indball_l0(q, x, s, r), expect)
        np.testing.assert_array_equal(orc.prox_indball_l0_binf(q, x, s, r, 0.7), np.clip(expect, -0.7, 0.7))


def test_l1_aliased_call_matches_reference_two_pass_semantics(orc):
    # reference shiftedNormL1.jl:47 overwrites y before the loop reads q: with y === q the result is -(xk)-sj
    import ctypes
    x = np.array([0.5, -1.0, 2.0]); s = np.array([0.25, 0.5, -0.5]); q = np.array([3.0, -4.0, 0.1])
    L = orc.lib()
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    L.orc_prox_l1(dp(q), dp(q), dp(x), dp(s), 3, 1.0, 1.0)
    np.testing.assert_array_equal(q, (-x) - s)


# ---- iprox! ----------------------------------------------------------------
@pytest.mark.parametrize("op", ["ShiftedNormL0Box", "ShiftedNormL1Box"])
def test_iprox_testsbox_cases_exact(orc, kats, op):
    # test/testsbox.jl:101-304: 14 enumerated cases per operator, exact ==
    t = kats["iprox_testsbox"]
    c = t[op]
    fn = orc.iprox_l0_box if op == "ShiftedNormL0Box" else orc.iprox_l1_box
    for k, (d, g, x, lam, sol) in enumerate(zip(c["d"], c["g"], c["x"], c["lambda"], c["sol"])):
        y = fn([g], [d], [x], [t["s"]], lam, [t["l"]], [t["u"]])
        assert y[0] == sol, (op, k + 1, y[0], sol)
        assert fn([g], [d], [x], [t["s"]], lam, t["l"], t["u"])[0] == sol  # scalar-bounds form


@pytest.mark.parametrize("op", ["l0", "l1"])
def test_iprox_partial_and_unboxed(orc, op):
    # test/partial_prox.jl:41-72
    rng = np.random.default_rng(4)
    n = 5
    x, s, q = rng.random(n), rng.random(n), rng.random(n) - 0.5
    l, u = np.zeros(n), np.ones(n)
    box = getattr(orc, "iprox_%s_box" % op)
    mask = orc.mask_from_selected(range(1, n + 1, 2), n)
    for d in (np.ones(n), -np.ones(n), np.zeros(n)):
        y = box(q, d, x, s, 3.14, l, u)
        z = box(q, d, x, s, 3.14, l, u, mask=mask)
        p = np.array([orc.iprox_zero(d[i], q[i], (l - s)[i], (u - s)[i]) for i in range(n)])
        for i in range(n):
            assert z[i] == (y[i] if mask[i] else p[i])
    unb = getattr(orc, "iprox_" + op)
    with pytest.raises(AssertionError):  # @test_throws AssertionError iprox(psi, q, zeros(n))
        unb(q, np.zeros(n), x, np.zeros(n), 3.14)
    y = unb(q, 2 * np.ones(n), x, s, 3.14)
    assert np.all(np.isfinite(y))


def test_iprox_box_is_objective_minimiser(orc):
    # independent check: brute-force minimisation of 1/2 d y^2 + g y + lambda h(x + s + y) over the box
    rng = np.random.default_rng(8)
    for name, h in (("iprox_l1_box", np.abs), ("iprox_l0_box", lambda v: (v != 0).astype(float))):
        for trial in range(300):
            x, g = rng.normal(), rng.normal()
            d = rng.choice([rng.uniform(0.2, 2.0), -rng.uniform(0.2, 2.0), 0.0])
            lam = rng.uniform(0.1, 2.0)
            l, u = -rng.uniform(0.2, 1.5), rng.uniform(0.2, 1.5)
            s = rng.uniform(l, u) * 0.5
            t = getattr(orc, name)([g], [d], [x], [s], lam, l, u)[0]
            obj = lambda yy: 0.5 * d * yy ** 2 + g * yy + lam * h(x + s + yy)
            lo, hi = l - s, u - s
            assert lo - 1e-12 <= t <= hi + 1e-12
            grid = np.concatenate([np.linspace(lo, hi, 200001), [c for c in (-(x + s),) if lo <= c <= hi]])
            best = np.min(obj(grid))
            assert obj(np.array([t]))[0] <= best + 1e-9 * max(1.0, abs(best)), (name, trial, t, d)


# ---- psi(y): the identities the reference's tests check ---------------------------------
def test_objective_identities(orc):
    # runtests.jl:172-178, 438-447: psi(0) == h(x); psi(y) == h(x + y) inside the trust region; Inf outside
    rng = np.random.default_rng(12)
    n = 7
    x = rng.random(n)
    z = np.zeros(n)
    assert orc.obj_plain("l1", z, x, z, 1.0) == np.sum(np.abs(x))
    assert orc.obj_plain("l0", z, x, z, 2.0) == 2.0 * np.count_nonzero(x)
    y = rng.random(n)
    y *= 0.01 / np.max(np.abs(y)) / 2
    for kind, h in (("l1", lambda v: np.sum(np.abs(v))), ("l0", lambda v: float(np.count_nonzero(v))),
                    ("lhalf", lambda v: np.sum(np.sqrt(np.abs(v))))):
        assert abs(orc.obj_box(kind, y, x, z, 1.0, -0.01, 0.01) - h(x + y)) <= 1e-14 * max(1.0, h(x + y))
        assert orc.obj_box(kind, 3 * y, x, z, 1.0, -0.01, 0.01) == np.inf
    assert orc.obj_indball_l0(y, x, z, n, delta=0.01) == 0.0
    assert orc.obj_indball_l0(y, x, z, n - 1, delta=0.01) == np.inf
    assert orc.obj_indball_l0(3 * y, x, z, n, delta=0.01) == np.inf   # 1.5 * Delta > 1.1 * Delta
    lam = np.array([0.4, 0.5])
    g = orc.obj_group_l2(y[:6], x[:6], z[:6], lam, offsets=[0, 3, 6])
    want = 0.4 * np.linalg.norm((x + y)[:3]) + 0.5 * np.linalg.norm((x + y)[3:6])
    assert abs(g - want) <= 1e-14
    assert orc.obj_group_l2(3 * y[:6], x[:6], z[:6], lam, offsets=[0, 3, 6], delta=0.01) == np.inf


# ---- ShiftedNormL1B2 --------------------------------------------------------------------
def test_l1b2_golden_and_minimiser(orc, kats):
    k = kats["box_golden"]  # same inputs, test/runtests.jl:467-474
    y = orc.prox_l1_b2(k["q"], k["x"], k["s"], k["lambda"], k["sigma"], k["delta"], 1.0)
    np.testing.assert_allclose(y, k["expected"]["ShiftedNormL1B2"], rtol=k["rtol"], atol=0)
    assert np.linalg.norm(y) <= k["delta"] * (1 + 1e-12)  # chi(s) <= Delta, runtests.jl:494
    # small independent check: the result minimises 1/(2 sigma)||t - q||^2 + lambda||x + s + t||_1 over ||s + t||_2 <= Delta
    rng = np.random.default_rng(21)
    for trial in range(20):
        n = 3
        x, s, q = rng.normal(size=n), rng.uniform(-0.2, 0.2, size=n), rng.normal(size=n)
        lam, sigma, delta = rng.uniform(0.2, 1.5), rng.uniform(0.3, 1.5), rng.uniform(0.5, 1.5)
        t = orc.prox_l1_b2(q, x, s, lam, sigma, delta, 1.0)
        obj = lambda tt: np.sum((tt - q) ** 2, axis=-1) / (2 * sigma) + lam * np.sum(np.abs(x + s + tt), axis=-1)
        assert np.linalg.norm(s + t) <= delta * (1 + 1e-9)
        cand = rng.normal(size=(200000, n))
        cand = cand / np.linalg.norm(cand, axis=1, keepdims=True) * delta * rng.random((200000, 1)) ** (1 / n) - s
        assert obj(t) <= np.min(obj(cand)) + 1e-6


def test_l1b2_objective(orc):
    # (psi::ShiftedNormL1B2)(y) = h(xk + sj + y) + IndBallL2(Delta)(sj + y)   src/shiftedNormL1B2.jl:32; the reference
    # test checks inside -> h(x + y), outside -> Inf (test/runtests.jl:443-447)
    rng = np.random.default_rng(8)
    n = 5
    x, s0 = np.ones(n), np.zeros(n)
    y = rng.random(n)
    y *= 0.01 / np.linalg.norm(y) / 2
    assert orc.obj_l1_b2(np.zeros(n), x, s0, 1.0, 0.01) == 5.0
    assert abs(orc.obj_l1_b2(y, x, s0, 1.0, 0.01) - np.sum(np.abs(x + y))) <= 1e-15
    assert orc.obj_l1_b2(3 * y, x, s0, 1.0, 0.01) == np.inf
    # on the sphere (||sj + y|| = Delta up to rounding): inside, by IndBallL2's isapprox slack
    yb = y / np.linalg.norm(y) * 0.01
    assert np.isfinite(orc.obj_l1_b2(yb, x, s0, 1.0, 0.01))
    assert orc.obj_l1_b2(yb * (1 + 1e-6), x, s0, 1.0, 0.01) == np.inf


def test_f32_oracle_agrees_with_f64_on_dyadic_data(orc):
    """The Float32 build of the L1 / L0 restatements (oracle/spx_oracle_f32.c) against the Float64 one on data where both
    are exact: multiples of 1/16 below 2^7 with dyadic lambda, sigma -- every sum, product and square is then representable
    in both formats and the two builds must agree value for value (the Float32 build has no reference vectors to pin it)."""
    rng = np.random.default_rng(11)
    n = 20_000
    x = rng.integers(-64, 65, size=n) / 16.0
    sj = rng.integers(-16, 17, size=n) / 16.0
    q = rng.integers(-96, 97, size=n) / 16.0
    l = -(rng.integers(8, 33, size=n) / 16.0)
    u = rng.integers(8, 33, size=n) / 16.0
    mask = (rng.random(n) < 0.7).astype(np.uint8)
    for lam, sigma in ((0.5, 1.0), (1.0, 0.25), (2.0, 2.0)):
        for op in ("l1", "l0"):
            if op == "l0" and not float(np.sqrt(2 * lam * sigma)).is_integer():
                continue                                   # sqrt(2 lambda sigma) must be exact in both formats
            a = orc.prox_f32(op, q, x, sj, lam, sigma)
            b = getattr(orc, "prox_" + op)(q, x, sj, lam, sigma)
            assert np.array_equal(a.astype(np.float64), b), (op, lam, sigma)
        for op in ("l1_box", "l0_box"):
            for lo, uo, m in ((-1.0, 1.0, None), (l, u, None), (l, u, mask)):
                a = orc.prox_f32(op, q, x, sj, lam, sigma, lo, uo, mask=m)
                b = getattr(orc, "prox_" + op)(q, x, sj, lam, sigma, lo, uo, mask=m)
                assert np.array_equal(a.astype(np.float64), b), (op, lam, sigma)
    # aliased ShiftedNormL1
    a = orc.prox_f32("l1", q, x, sj, 0.5, 1.0, aliased=True)
    assert np.array_equal(a.astype(np.float64), (-x) - sj)


def test_f32_iprox_oracle_agrees_with_f64_on_dyadic_data(orc):
    """The Float32 iprox! restatement (generated from the Float64 block, oracle/Makefile) against the Float64 one where both are
    exact: dyadic data, d in {1/2, 1, 2, 4} (every quotient and product representable in both formats), and the d = 0 /
    d < 0 branches of the Box forms.  lambda = 2 d for the L0 forms (sqrt(2 lambda d) exact)."""
    rng = np.random.default_rng(12)
    n = 20_000
    x = rng.integers(-64, 65, size=n) / 16.0
    sj = rng.integers(-16, 17, size=n) / 16.0
    g = rng.integers(-96, 97, size=n) / 16.0
    l = -(rng.integers(8, 33, size=n) / 16.0)
    u = rng.integers(8, 33, size=n) / 16.0
    mask = (rng.random(n) < 0.7).astype(np.uint8)
    d = rng.choice([0.5, 1.0, 2.0, 4.0], size=n)
    a, bad = orc.iprox_f32("l1", g, d, x, sj, 0.5)
    b = orc.iprox_l1(g, d, x, sj, 0.5)
    assert bad == -1 and np.array_equal(a.astype(np.float64), b)
    d1 = np.full(n, 2.0)
    a, _ = orc.iprox_f32("l0", g, d1, x, sj, 4.0)       # sqrt(2 * 4 * 2) = 4
    b = orc.iprox_l0(g, d1, x, sj, 4.0)
    assert np.array_equal(a.astype(np.float64), b)
    dsg = rng.choice([0.5, 1.0, 2.0, -0.5, -1.0, -2.0, 0.0], size=n)
    for op in ("l1_box", "l0_box"):
        for lo, uo, m in ((-1.0, 1.0, None), (l, u, None), (l, u, mask)):
            a = orc.iprox_f32(op, g, dsg, x, sj, 0.5, lo, uo, mask=m)
            b = getattr(orc, "iprox_" + op)(g, dsg, x, sj, 0.5, lo, uo, mask=m)
            assert np.array_equal(a.astype(np.float64), b), op
    # the first d <= 0 is reported at the same index
    d2 = d.copy(); d2[123] = 0.0
    assert orc.iprox_f32("l1", g, d2, x, sj, 0.5)[1] == 123
    with pytest.raises(orc.AssertionErrorAt) as e:
        orc.iprox_l1(g, d2, x, sj, 0.5)
    assert e.value.index == 123


def test_synthetic_generator_twins_agree(orc):
    """SURVEY 8d's shared generator: the C host twin (oracle.synth_fill) and the numpy one (oracle/synth.py) give the same
    bits; the draws have the advertised moments; streams and seeds are independent of each other."""
    from oracle import synth
    oracle = orc
    n = 200_003
    for kind in (0, 1):
        a = oracle.synth_fill(n, 20250613, 2, kind, 0.75, threads=3)
        b = synth.fill(n, 20250613, 2, kind, 0.75)
        assert np.array_equal(a.view(np.int64), b.view(np.int64))
        # a window of the same stream equals the slice (counter-based: no state)
        w = synth.fill(1000, 20250613, 2, kind, 0.75, start=77_000)
        assert np.array_equal(w.view(np.int64), a[77_000:78_000].view(np.int64))
    u = oracle.synth_fill(n, 1, 0, 0)
    g = oracle.synth_fill(n, 1, 0, 1)
    assert u.min() >= -0.5 and u.max() < 0.5 and abs(u.mean()) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    assert abs(g.mean()) < 1e-2 and abs(g.var() - 1.0) < 2e-2 and np.abs(g).max() <= 6.0
    assert abs(np.corrcoef(g, oracle.synth_fill(n, 1, 1, 1))[0, 1]) < 1e-2
    assert abs(np.corrcoef(g, oracle.synth_fill(n, 2, 0, 1))[0, 1]) < 1e-2


def test_l1b2_exact_zero_at_a_bracket_end(orc):
    """Integer lattice data where froot(eta) = eta - chi(ProjB(-xk eta / Delta)) vanishes EXACTLY at a point the bracket search
    lands on (eta = 4 = 2^2 Delta): find_zero returns that point.  The restatement used to step over it and returned 2 eta
    (found by tools/fuzz_r2_onelaunch.py; the GPU had the root).  Checked against the definition: froot(eta) == 0 at the eta
    the result implies, in Float64 and in the binary128 arbiter."""
    x = np.array([-2.0, -1.0, -1.0, -1.0, 0.0, 2.0, -2.0, 0.0, -0.0, -0.0, -0.0, -0.0, 1.0, 1.0, -0.0, -0.0, 1.0, 0.0, 2.0])
    q = np.array([0.0, -0.0, -2.0, -0.0, -3.0, 0.0, -1.0, 1.0, -0.0, 1.0, -1.0, 1.0, 0.0, 0.0, 0.0, -1.0, -1.0, -1.0, -0.0])
    sj = np.zeros(19)
    eta = 4.0
    proj = np.minimum(np.maximum(-x * eta, q - 1.0), q + 1.0)
    assert eta - np.linalg.norm(proj) == 0.0                      # the root, exactly
    want = proj / eta - sj
    got = orc.prox_l1_b2(q, x, sj, 1.0, 1.0, 1.0, 1.0)
    assert np.array_equal(got, want)
    assert np.max(np.abs(orc.q_prox_l1_b2(q, x, sj, 1.0, 1.0, 1.0, 1.0) - want)) <= 1e-15


def test_topr_many_ranks_from_one_sort(orc):
    """oracle.TopR (the reference's sortperm computed once, reused for every r) returns what the literal restatements return,
    bit for bit: ties (index order), NaN / Inf (isless order), r = 0 ... n + 3, with and without the clamp."""
    rng = np.random.default_rng(4)
    n = 5000
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = np.round(rng.normal(size=n) * 4) / 4
    q[[7, 99, 1234]] = np.nan
    q[[8, 100]] = np.inf
    x[17] = -np.inf
    with np.errstate(all="ignore"):
        top = orc.TopR(q, x, sj)
        for r in (0, 1, 2, 5, 6, 7, 50, n // 2, n - 1, n, n + 3):
            a, b = top.prox(r), orc.prox_indball_l0(q, x, sj, r)
            assert np.array_equal(a.view(np.int64), b.view(np.int64)) or np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
            a, b = top.prox(r, 0.6), orc.prox_indball_l0_binf(q, x, sj, r, 0.6)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.int64), b[~np.isnan(b)].view(np.int64))
        o = orc.topr_order_f32(q, x, sj)
        for r in (1, 7, n // 3):
            a, b = orc.prox_indball_l0_f32(q, x, sj, r, 0.6, _order=o), orc.prox_indball_l0_f32(q, x, sj, r, 0.6)
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.int32), b[~np.isnan(b)].view(np.int32))
