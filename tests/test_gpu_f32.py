"""Float32 forms of the NormL1 / NormL0 families (round 2 widening; the reference is generic in R <: Real,
src/shiftedNormL1Box.jl:89-94, and builds Float32 operators on views, test/runtests.jl:196-209).
Bar: BIT-EXACT against the Float32 build of the oracle (oracle/spx_oracle_f32.c) -- every operation of these bodies is a
Float32 operation in the reference."""
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


def _data(n, seed, quant=None):
    rng = np.random.default_rng(seed)
    x, sj, q = (rng.normal(size=n).astype(np.float32), rng.uniform(-0.5, 0.5, size=n).astype(np.float32),
                rng.normal(size=n).astype(np.float32))
    if quant:
        x, sj, q = (np.round(v * quant) / np.float32(quant) for v in (x, sj, q))
    return x.astype(np.float32), sj.astype(np.float32), q.astype(np.float32)


def _bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.int32), np.asarray(b, dtype=np.float32).view(np.int32))


def _dev(*arrs, off=0):
    """device float32 vectors; off > 0: views that start `off` elements into a larger allocation (4-byte misalignment)"""
    import torch
    out = []
    for a in arrs:
        t = torch.cat([torch.zeros(off, dtype=torch.float32), torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))]).cuda()
        out.append(t[off:])
    return out


OPS = {"l1": "NormL1", "l0": "NormL0", "l1_box": "NormL1", "l0_box": "NormL0"}


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 1023, 4096, 4097, 100_003, 2_000_003])
@pytest.mark.parametrize("op", list(OPS))
def test_f32_bit_exact(s, orc, op, n):
    x, sj, q = _data(n, 40 + n, quant=16 if n % 2 else None)   # lattice data on the odd sizes: ties with thresholds / bounds
    lam, sigma = np.float32(0.7), np.float32(1.3)
    h = getattr(s, OPS[op])(float(lam))
    xd, sd, qd = _dev(x, sj, q)
    assert xd.dtype.is_floating_point and xd.element_size() == 4
    if op.endswith("box"):
        psi = s.shifted(s.shifted(h, xd, 0.9, s.NormLinf(1.0)), sd)
        ref = orc.prox_f32(op, q, x, sj, lam, sigma, np.float32(-0.9), np.float32(0.9))
    else:
        psi = s.shifted(s.shifted(h, xd), sd)
        ref = orc.prox_f32(op, q, x, sj, lam, sigma)
    assert psi.f32 and psi.sol.element_size() == 4
    y = s.prox(psi, qd, float(sigma)).cpu().numpy()
    assert _bits(y, ref), (op, n)
    # y === q (ShiftedNormL1: the two-pass body reads the overwritten q)
    q2 = qd.clone()
    s.prox_bang(q2, psi, q2, float(sigma))
    ref_a = orc.prox_f32(op, q, x, sj, lam, sigma, np.float32(-0.9), np.float32(0.9), aliased=True) if op.endswith("box") else \
        orc.prox_f32(op, q, x, sj, lam, sigma, aliased=True)
    assert _bits(q2.cpu().numpy(), ref_a), (op, n, "aliased")


@pytest.mark.parametrize("off", [1, 2, 3])
@pytest.mark.parametrize("op", list(OPS))
def test_f32_views_from_any_element(s, orc, op, off):
    """Vectors that start 4, 8 or 12 bytes off a 16-byte boundary (a unit-stride view from element `off`): peeled to the
    boundary when all vectors share the offset, 4-byte accesses when they do not."""
    import torch
    n = 50_001
    x, sj, q = _data(n, 900 + off)
    lam, sigma = np.float32(1.1), np.float32(0.6)
    h = getattr(s, OPS[op])(float(lam))
    box = op.endswith("box")
    ref = orc.prox_f32(op, q, x, sj, lam, sigma, np.float32(-1.0), np.float32(0.8)) if box else orc.prox_f32(op, q, x, sj, lam, sigma)
    for offs in ((off, off, off, off), (off, 0, off, 0), (0, off, 0, off)):      # all alike / mixed
        xd = _dev(x, off=offs[0])[0]; sd = _dev(sj, off=offs[1])[0]; qd = _dev(q, off=offs[2])[0]
        yd = _dev(np.zeros(n, np.float32), off=offs[3])[0]
        assert xd.data_ptr() % 16 == (4 * offs[0]) % 16
        psi = s.shifted(s.shifted(h, xd, -1.0, 0.8), sd) if box else s.shifted(s.shifted(h, xd), sd)
        s.prox_bang(yd, psi, qd, float(sigma))
        assert _bits(yd.cpu().numpy(), ref), (op, offs)


@pytest.mark.parametrize("op", ["l1_box", "l0_box"])
def test_f32_vector_bounds_and_mask(s, orc, op):
    rng = np.random.default_rng(5)
    for n, off in ((1000, 0), (65_537, 0), (65_537, 3)):
        x, sj, q = _data(n, 77 + n)
        l = (-1.0 - 0.1 * rng.random(n)).astype(np.float32)
        u = (1.0 + 0.1 * rng.random(n)).astype(np.float32)
        lam, sigma = np.float32(0.9), np.float32(0.8)
        h = getattr(s, OPS[op])(float(lam))
        xd, sd, qd, ld, ud = _dev(x, sj, q, l, u, off=off)
        for lo, uo, ldv, udv in ((l, u, ld, ud), (l, np.float32(1.05), ld, 1.05), (np.float32(-1.05), u, -1.05, ud)):
            psi = s.shifted(s.shifted(h, xd, ldv, udv), sd)
            assert _bits(s.prox(psi, qd, float(sigma)).cpu().numpy(), orc.prox_f32(op, q, x, sj, lam, sigma, lo, uo)), (op, n, off)
            selected = sorted(rng.choice(n, size=n // 3, replace=False).tolist())
            mask = orc.mask_from_selected([i + 1 for i in selected], n)
            psi = s.shifted(s.shifted(h, xd, ldv, udv, selected), sd)
            assert _bits(s.prox(psi, qd, float(sigma)).cpu().numpy(), orc.prox_f32(op, q, x, sj, lam, sigma, lo, uo, mask=mask)), (op, n, off, "mask")
    with pytest.raises(ValueError, match="lower bound is greater"):
        s.shifted(getattr(s, OPS[op])(1.0), xd, ud, ld)


def test_f32_special_values(s, orc):
    """{+-0, +-1, 0.5, 2, +-Inf, NaN, denormals, huge}^3 for (q, xk, sj): NaN where the reference has NaN, the same bits elsewhere."""
    vals = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e38, -3e38, np.sqrt(2.0)], dtype=np.float32)
    g = np.array(np.meshgrid(vals, vals, vals, indexing="ij")).reshape(3, -1)
    q, x, sj = g[0].copy(), g[1].copy(), g[2].copy()
    xd, sd, qd = _dev(x, sj, q)
    for op in OPS:
        for lam, sigma, lo, up in ((1.0, 1.0, -1.0, 1.0), (0.5, 2.0, -np.inf, 0.5), (2.0, 0.25, 0.0, 0.0)):
            h = getattr(s, OPS[op])(lam)
            with np.errstate(all="ignore"):
                if op.endswith("box"):
                    psi = s.shifted(s.shifted(h, xd, lo, up), sd)
                    ref = orc.prox_f32(op, q, x, sj, lam, sigma, np.float32(lo), np.float32(up))
                else:
                    psi = s.shifted(s.shifted(h, xd), sd)
                    ref = orc.prox_f32(op, q, x, sj, lam, sigma)
            y = s.prox(psi, qd, sigma).cpu().numpy()
            nan = np.isnan(y) & np.isnan(ref)
            assert bool(np.all(nan | (y.view(np.int32) == ref.view(np.int32)))), (op, lam, sigma)


def test_f32_type_rules(s):
    import torch
    x32 = torch.ones(8, dtype=torch.float32, device="cuda:0")
    x64 = torch.ones(8, dtype=torch.float64, device="cuda:0")
    psi = s.shifted(s.NormL1(1.0), x32)
    assert psi.sj.dtype == torch.float32 and psi.sol.dtype == torch.float32
    with pytest.raises(TypeError):
        s.prox(psi, x64, 1.0)                                  # element types must agree (no MethodError-free mixing)
    with pytest.raises(TypeError):
        s.shifted(psi, x64)
    with pytest.raises(TypeError, match="no Float32 form"):
        s.prox(s.shifted(s.RootNormLhalf(1.0), x32), x32, 1.0)   # computed in Float64 by the reference itself
    assert psi(x32) == 16.0                                    # psi(y) has a Float32 form: |1 + 0 + 1| x 8


# ----------------------------------------------------------------------------------------------------------------------
# psi(y) on Float32 vectors, every operator (spx_obj_*_f32).  The reference's tests build each shifted operator on Float32
# data and evaluate it (test/runtests.jl:196-209, 268-282, 346-360, 397-412, 524-550, 630-646).  Expected values: numpy with
# every element operation in float32, as the reference does them, sums in float64, rounded to float32 at the end (the
# reference sums pairwise in Float32: agreement to a few float32 ulps); counts and +Inf decisions exact.
# ----------------------------------------------------------------------------------------------------------------------
F32 = np.float32
EPS32 = F32(np.sqrt(np.finfo(np.float32).eps))


def _close32(got, exp):
    if np.isinf(exp) or np.isinf(got):
        return got == exp
    return abs(got - exp) <= 4 * np.finfo(np.float32).eps * max(abs(exp), 1e-30)


@pytest.mark.parametrize("n", [1, 5, 1000, 300_001])
def test_f32_objective_every_operator(s, n):
    import torch
    rng = np.random.default_rng(3200 + n)
    x = rng.normal(size=n).astype(F32); sj = rng.uniform(-0.5, 0.5, size=n).astype(F32)
    lam = F32(1.2)
    xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(sj).cuda()
    for scale in (0.0, 0.3, 1.0):
        y = (rng.normal(size=n) * scale).astype(F32)
        if n >= 5:
            y[1] = -(x[1] + sj[1])           # an exact zero of xk + sj + y in Float32 (NormL0 / IndBallL0 count it out)
        yd = torch.from_numpy(y).cuda()
        xsy = (x + sj) + y                    # float32 + float32, as `@. xsy = xk + sj + y`
        assert xsy.dtype == np.float32
        t = sj + y
        # unboxed forms (src/ShiftedProximalOperators.jl:51-54)
        exp = {"l1": np.sum(np.abs(xsy), dtype=np.float64), "l0": float(np.count_nonzero(xsy)),
               "lhalf": np.sum(np.sqrt(np.abs(xsy)), dtype=np.float64)}
        for H, key in ((s.NormL1, "l1"), (s.NormL0, "l0"), (s.RootNormLhalf, "lhalf")):
            psi = s.shifted(s.shifted(H(float(lam)), xd), sd)
            assert _close32(psi(yd), float(F32(float(lam) * exp[key]))), (key, n, scale)
            # Box forms: feasibility of sj + y against [l - sqrt(eps32), u + sqrt(eps32)], all in Float32 (shiftedNormL1Box.jl:70-82)
            for lo, up in ((-10.0, 10.0), (float(t.min()), float(t.max())), (float(t.min()) + 1e-3, 10.0),
                           (float(F32(t.min()) + F32(0.5) * EPS32), 10.0)):
                lo32, up32 = F32(lo), F32(up)
                feasible = bool(np.all((F32(lo32 - EPS32) <= t) & (t <= F32(up32 + EPS32))))
                pb = s.shifted(s.shifted(H(float(lam)), xd, float(lo32), float(up32)), sd)
                want = float(F32(float(lam) * exp[key])) if feasible else np.inf
                assert _close32(pb(yd), want), (key, "box", n, scale, lo, up, feasible)
        # IndBallL0 / IndBallL0BInf (count <= r; |sj + y| <= 1.1 Delta with the product in Float64: shiftedIndBallL0BInf.jl:44-49)
        nnz = int(np.count_nonzero(xsy))
        for r in (max(nnz - 1, 0), nnz, n):
            if r < 1:
                continue
            pi = s.shifted(s.shifted(s.IndBallL0(r), xd), sd)
            assert pi(yd) == (0.0 if nnz <= r else np.inf), (n, scale, r)
            for delta in (F32(0.25), F32(np.abs(t).max() / 1.1 * 1.0001), F32(10.0)):
                inside = bool(np.all(np.abs(t.astype(np.float64)) <= 1.1 * float(delta)))
                pbi = s.shifted(s.shifted(s.IndBallL0(r), xd, float(delta), s.NormLinf(1.0)), sd)
                assert pbi(yd) == (0.0 if (nnz <= r and inside) else np.inf), (n, scale, r, float(delta))
        # GroupNormL2 / GroupNormL2Binf on uniform groups
        for gs in (1, 5, 100):
            if n % gs:
                continue
            ng = n // gs
            lam_g = rng.uniform(0.5, 1.5, size=ng).astype(F32)
            norms = np.sqrt(np.sum(xsy.astype(np.float64).reshape(ng, gs) ** 2, axis=1))
            want = float(F32(np.sum(lam_g.astype(np.float64) * norms)))
            h = s.GroupNormL2.uniform(torch.from_numpy(lam_g).cuda(), gs)
            pg = s.shifted(s.shifted(h, xd), sd)
            assert _close32(pg(yd), want), ("group", n, gs)
            for delta in (F32(0.25), F32(10.0)):
                inside = bool(np.all(np.abs(t.astype(np.float64)) <= 1.1 * float(delta)))
                pgb = s.shifted(s.shifted(h, xd, float(delta), s.NormLinf(1.0)), sd)
                assert _close32(pgb(yd), want if inside else np.inf), ("group binf", n, gs, float(delta))


def test_f32_reference_type_block(s):
    """The "test different types" block the reference runs for every shifted operator: h = Op(Float32(1.2)), x Float32,
    x = view(y, 1:2:10) -- a STRIDED view --, psi = shifted(h, x); psi.lambda == h.lambda; psi(zeros(Float32, n)) == h(x)."""
    import torch
    base = torch.rand(10, dtype=torch.float32, device="cuda")
    x = base[0::2]                                  # `x = view(y, 1:2:10)`: a strided view (kept as view + packed copy)
    z = torch.zeros(5, dtype=torch.float32, device="cuda")
    chi = s.NormLinf(1.0)
    for h in (s.NormL0(1.2), s.NormL1(1.2), s.RootNormLhalf(1.2), s.IndBallL0(3), s.GroupNormL2.uniform([1.2], 5)):
        psi = s.shifted(h, x)
        assert psi.xk.dtype == torch.float32 and psi.sj.dtype == torch.float32 and psi.sol.dtype == torch.float32
        assert psi(z) == h(x), type(h).__name__
        if not isinstance(h, s.IndBallL0):
            assert psi.λ == h.lam or list(np.atleast_1d(psi.λ)) == list(np.atleast_1d(h.lam))
    for h in (s.NormL0(1.2), s.NormL1(1.2), s.RootNormLhalf(1.2)):
        for psi in (s.shifted(h, x, -0.5, 0.5), s.shifted(h, x, 0.5, chi)):   # runtests.jl:524-550
            assert psi(z) == h(x), type(h).__name__
    for h in (s.IndBallL0(3), s.GroupNormL2.uniform([1.2], 5)):
        psi = s.shifted(h, x, 0.5, chi)                                        # runtests.jl:630-646, 749-
        assert psi(z) == h(x), type(h).__name__


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_strided_xk_views(s, orc, dtype):
    """xk as a strided view of the caller's array (test/runtests.jl:196-209 builds every operator on `view(y, 1:2:10)`): the
    mirror keeps the view and a packed copy, refreshed before every call (spx_copy_strided); a change of the caller's array
    is seen by the next call and shift! writes through to it, as with the reference's SubArray."""
    import torch
    td = getattr(torch, dtype)
    nd = np.float64 if dtype == "float64" else np.float32
    n, st = 50_001, 3
    rng = np.random.default_rng(12)
    base = rng.normal(size=n * st).astype(nd)
    bd = torch.from_numpy(base).cuda()
    xv = bd[::st]
    assert xv.stride(0) == st and xv.numel() == n
    sj = rng.uniform(-0.5, 0.5, size=n).astype(nd); q = rng.normal(size=n).astype(nd)
    sd, qd = torch.from_numpy(sj).cuda(), torch.from_numpy(q).cuda()
    psi = s.shifted(s.shifted(s.NormL1(1.0), xv, 1.0, s.NormLinf(1.0)), sd)

    def expect(xh):
        if dtype == "float64":
            return orc.prox_l1_box(q, np.ascontiguousarray(xh), sj, 1.0, 1.0, -1.0, 1.0)
        return orc.prox_f32("l1_box", q, np.ascontiguousarray(xh), sj, 1.0, 1.0, nd(-1.0), nd(1.0))

    same = lambda a, b: np.array_equal(np.asarray(a).view(np.int64 if dtype == "float64" else np.int32),
                                       np.asarray(b).view(np.int64 if dtype == "float64" else np.int32))
    assert same(s.prox(psi, qd, 1.0).cpu().numpy(), expect(base[::st]))
    bd[::st] += 0.25                                   # the caller changes its array: the next call sees it
    assert same(s.prox(psi, qd, 1.0).cpu().numpy(), expect(bd[::st].cpu().numpy()))
    others = bd.clone()
    parent = s.shifted(s.NormL1(1.0), xv, 1.0, s.NormLinf(1.0))
    new = torch.from_numpy(rng.normal(size=n).astype(nd)).cuda()
    s.shift_bang(parent, new)                          # `ψ.xk .= shift`: into the caller's strided storage
    assert torch.equal(bd[::st], new)
    mask = torch.ones(n * st, dtype=torch.bool, device="cuda"); mask[::st] = False
    assert torch.equal(bd[mask], others[mask])         # nothing between the strided elements was touched
    assert abs(parent(torch.zeros(n, dtype=td, device="cuda")) - float(new.abs().sum())) <= 1e-5 * float(new.abs().sum())


# ------------------------------------------------------------------------------------------------------------------
# iprox! in Float32 (round 3): the reference's iprox! methods are generic in R too; thresholds eps(Float32)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 3, 4, 5, 1023, 4097, 300_001])
def test_f32_iprox_bit_exact(s, orc, n):
    import torch
    rng = np.random.default_rng(70 + n)
    x, sj, g = _data(n, 70 + n, quant=8 if n % 2 else None)
    dpos = rng.uniform(0.3, 2.0, size=n).astype(np.float32)
    # Box forms take any sign of d, exact zeros and values at +-eps(Float32)
    dany = np.resize(np.array([1.0, -1.0, 0.0, -0.0, 2.0, 1e-8, -1e-8, 1.1920929e-07, -1.1920929e-07, 0.5, -3.0], dtype=np.float32), n)
    dany = (dany * rng.uniform(0.5, 1.5, size=n).astype(np.float32)).astype(np.float32)
    lam = np.float32(0.7)
    xd, sd, gd = _dev(x, sj, g)
    for op, H in (("l1", s.NormL1), ("l0", s.NormL0)):
        psi = s.shifted(s.shifted(H(float(lam)), xd), sd)
        dd = _dev(dpos)[0]
        y = s.iprox(psi, gd, dd).cpu().numpy()
        ref, bad = orc.iprox_f32(op, g, dpos, x, sj, lam)
        assert bad == -1 and _bits(y, ref), (op, n)
    lo_v = (-0.9 - 0.2 * rng.random(n)).astype(np.float32)
    up_v = (0.9 + 0.2 * rng.random(n)).astype(np.float32)
    sel = np.flatnonzero(rng.random(n) < 0.8)
    mask = np.zeros(n, dtype=np.uint8); mask[sel] = 1
    for op, H in (("l1_box", s.NormL1), ("l0_box", s.NormL0)):
        for dv in (dpos, dany):
            dd = _dev(dv)[0]
            psi = s.shifted(s.shifted(H(float(lam)), xd, -0.9, 0.9), sd)
            assert _bits(s.iprox(psi, gd, dd).cpu().numpy(), orc.iprox_f32(op, g, dv, x, sj, lam, np.float32(-0.9), np.float32(0.9))), (op, n)
            lod, upd = _dev(lo_v, up_v)
            psi = s.shifted(s.shifted(H(float(lam)), xd, lod, upd, torch.from_numpy(sel).cuda()), sd)
            assert _bits(s.iprox(psi, gd, dd).cpu().numpy(), orc.iprox_f32(op, g, dv, x, sj, lam, lo_v, up_v, mask=mask)), (op, n, "vector bounds + mask")


def test_f32_iprox_asserts_d_positive_and_views(s, orc):
    """unboxed forms: `@assert d[i] > 0` (check=True raises AssertionError); views from any element"""
    n = 10_001
    x, sj, g = _data(n, 5)
    d = np.random.default_rng(6).uniform(0.3, 2.0, size=n).astype(np.float32)
    for off in (1, 2, 3):
        xd, sd, gd, dd = _dev(x, sj, g, d, off=off)
        psi = s.shifted(s.shifted(s.NormL1(0.4), xd), sd)
        y = s.iprox(psi, gd, dd).cpu().numpy()
        assert _bits(y, orc.iprox_f32("l1", g, d, x, sj, np.float32(0.4))[0]), off
    d2 = d.copy(); d2[777] = 0.0
    xd, sd, gd, dd = _dev(x, sj, g, d2)
    with pytest.raises(AssertionError):
        s.iprox(s.shifted(s.shifted(s.NormL0(0.4), xd), sd), gd, dd)


# ------------------------------------------------------------------------------------------------------------------
# ShiftedIndBallL0(BInf) in Float32 (round 3): exact select on Float32 keys, every size class of the one-launch kernels
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 5, 1000, 8192, 8193, 50_001, 1_000_003, (1 << 21) + 1, (1 << 21) + 2, (1 << 21) + 3, 2_500_001,
                               pytest.param(5_000_002, marks=pytest.mark.soak), pytest.param((1 << 23) - 1, marks=pytest.mark.soak),
                               pytest.param(1 << 23, marks=pytest.mark.soak), pytest.param((1 << 23) + 3, marks=pytest.mark.soak),
                               -2_500_001])
def test_f32_topr_bit_exact(s, orc, n):
    """one workgroup (n <= 8192), register-resident (<= 2^21), v parked in LDS (<= 2^23: 32 elements per lane; sizes that leave
    1, 2, 3 elements behind the last 16-byte vector), v parked in y (beyond, and for a negative n: the same size with the LDS
    form switched off, tuning key 11 = 0): bit for bit against the numpy
    restatement in Float32 (stable descending sort by |v|): continuous data, a 1/8 lattice (ties: lowest index first), NaN and
    Inf entries (NaN is the largest magnitude, all NaNs tie), y === q, views from an odd element."""
    lds = n > 0
    n = abs(n)
    L = s._lib.load()
    s._lib.check(L.spx_ctx_set_tuning(s.context("cuda:0"), 11, 1 if lds else 0))
    try:
        _f32_topr_cases(s, orc, n)
    finally:
        s._lib.check(L.spx_ctx_set_tuning(s.context("cuda:0"), 11, 1))


def _f32_topr_cases(s, orc, n):
    for kind in ("continuous", "lattice", "special"):
        x, sj, q = _data(n, 300 + n, quant=8 if kind == "lattice" else None)
        if kind == "special" and n >= 50:
            rng = np.random.default_rng(n)
            q[rng.choice(n, size=3, replace=False)] = np.nan
            q[rng.choice(n, size=2, replace=False)] = np.inf
            q[rng.choice(n, size=2, replace=False)] = -np.inf
        xd, sd, qd = _dev(x, sj, q)
        with np.errstate(all="ignore"):
            order = orc.topr_order_f32(q, x, sj)   # (the restatement's stable sort once per input, every r from it)
        for r in sorted({1, max(1, n // 100), max(1, n // 2), max(1, n - 1), n, n + 5}):
            with np.errstate(all="ignore"):
                ref = orc.prox_indball_l0_f32(q, x, sj, r, 0.75, _order=order)
                ref0 = orc.prox_indball_l0_f32(q, x, sj, r, _order=order)
            y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.75, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
            same = (y.view(np.int32) == ref.view(np.int32)) | (np.isnan(y) & np.isnan(ref))
            assert same.all(), (n, kind, r, int((~same).sum()))
            y0 = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd), sd), qd, 1.0).cpu().numpy()
            same = (y0.view(np.int32) == ref0.view(np.int32)) | (np.isnan(y0) & np.isnan(ref0))
            assert same.all(), (n, kind, r, "plain", int((~same).sum()))
        if n >= 5:
            r = max(1, n // 3)
            with np.errstate(all="ignore"):
                ref0 = orc.prox_indball_l0_f32(q, x, sj, r)
            qa = qd.clone()
            s.prox_bang(qa, s.shifted(s.shifted(s.IndBallL0(r), xd), sd), qa, 1.0)     # y === q
            ya = qa.cpu().numpy()
            assert ((ya.view(np.int32) == ref0.view(np.int32)) | (np.isnan(ya) & np.isnan(ref0))).all(), (n, kind, "aliased")
            xo, so, qo = _dev(x, sj, q, off=1)
            yo = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xo), so), qo, 1.0).cpu().numpy()
            assert ((yo.view(np.int32) == ref0.view(np.int32)) | (np.isnan(yo) & np.isnan(ref0))).all(), (n, kind, "view")


@pytest.mark.parametrize("kind", ["tiny", "huge", "tiny+normal", "normal+huge", "zeros", "constant", "binade_edges", "wide_exponents"])
@pytest.mark.parametrize("n", [70_001, 2_500_001])
def test_f32_topr_folded_first_digit(s, orc, kind, n):
    """The one-launch selects fold their first digit for Float32 keys too (csrc/spx_select.hip fold_digit_f32: 62 binades around
    1.0 x 64 mantissa steps, catch-all bins below 2^-32 and above 2^30): data that puts the threshold into a catch-all bin
    (restart with the plain top digit), onto a bin edge, into ties at key 0, or all of the vector into one bin -- through the
    register form (n = 70 001) and the LDS form (n = 2 500 001)."""
    rng = np.random.default_rng(sum(map(ord, kind)) + n)
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    if kind == "tiny":
        x, sj, q = x * 2.0 ** -60, sj * 2.0 ** -60, q * 2.0 ** -60
    elif kind == "huge":
        x, sj, q = x * 2.0 ** 70, sj * 2.0 ** 70, q * 2.0 ** 70
    elif kind == "tiny+normal":
        m = rng.random(n) < 0.5
        x, sj, q = np.where(m, x * 2.0 ** -45, x), np.where(m, sj * 2.0 ** -45, sj), np.where(m, q * 2.0 ** -45, q)
    elif kind == "normal+huge":
        m = rng.random(n) < 0.3
        x, sj, q = np.where(m, x * 2.0 ** 40, x), np.where(m, sj * 2.0 ** 40, sj), np.where(m, q * 2.0 ** 40, q)
    elif kind == "zeros":
        m = rng.random(n) < 0.7
        x, sj, q = np.where(m, 0.0, x), np.where(m, 0.0, sj), np.where(m, 0.0, q)
    elif kind == "constant":
        x, sj, q = np.full(n, 0.5), np.full(n, 0.25), np.full(n, -2.0)
    elif kind == "binade_edges":  # |v| on the edges of the folded bins: 2^k (1 + j/64), exactly
        k = rng.integers(-3, 4, size=n); j = rng.integers(0, 64, size=n)
        x, sj = np.zeros(n), np.zeros(n); q = 2.0 ** k * (1.0 + j / 64.0) * rng.choice([-1.0, 1.0], size=n)
    else:
        e = rng.integers(-40, 40, size=n)
        x, sj, q = x * 2.0 ** e, sj * 2.0 ** e, q * 2.0 ** e
    x, sj, q = (a.astype(np.float32) for a in (x, sj, q))
    xd, sd, qd = _dev(x, sj, q)
    for r in sorted({1, 2, 26, n // 100, n // 3, n // 2, int(0.7 * n), n - 1}):
        with np.errstate(all="ignore"):
            ref = orc.prox_indball_l0_f32(q, x, sj, r, 0.8)
        y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.8, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
        same = (y.view(np.int32) == ref.view(np.int32)) | (np.isnan(y) & np.isnan(ref))
        assert same.all(), (kind, n, r, int((~same).sum()))


# ------------------------------------------------------------------------------------------------------------------
# ShiftedGroupNormL2 in Float32 (round 3): elementwise operations in Float32, the norm to a Float32 ulp
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gs", [1, 2, 3, 4, 7, 9, 16, 33, 64, 128, 1000])
def test_f32_group_l2(s, gs):
    """uniform groups and ragged CSR groups (incl. indices in no group: y - (xk + sj), src/shiftedGroupNormL2.jl:77), y === q.
    Reference: the method restated in numpy with Float32 operations and the norm formed in Float64 and rounded once (the
    reference's own `norm` is BLAS / a generic loop: neither pins the last ulp) -- bar 1e-6 of the operands' scale."""
    import torch
    rng = np.random.default_rng(500 + gs)
    ng = 3000
    n = ng * gs
    x, sj, q = _data(n, 500 + gs)
    lam = rng.uniform(0.2, 3.0, size=ng).astype(np.float32)
    sigma = np.float32(0.8)

    def ref(offs, y_in):
        S = ((q + x) + sj).astype(np.float32)
        y = y_in.copy()
        for g in range(len(offs) - 1):
            lo, hi = offs[g], offs[g + 1]
            sn = np.float32(np.sqrt(np.sum(S[lo:hi].astype(np.float64) ** 2)))
            if sn == 0:
                y[lo:hi] = 0
            else:
                a = np.maximum(np.float32(1) - sigma * lam[g] / sn, np.float32(0))
                y[lo:hi] = a * S[lo:hi]
        return (y - (x + sj)).astype(np.float32), S

    xd, sd, qd = _dev(x, sj, q)
    h = s.GroupNormL2.uniform(torch.from_numpy(lam).cuda(), gs)
    psi = s.shifted(s.shifted(h, xd), sd)
    assert psi.f32
    y = s.prox(psi, qd, float(sigma)).cpu().numpy()
    offs = np.arange(0, n + 1, gs)
    want, S = ref(offs, np.zeros(n, dtype=np.float32))
    nrm = np.repeat(np.sqrt(np.add.reduceat(S.astype(np.float64) ** 2, offs[:-1])), gs)
    scale = np.maximum(np.maximum(np.abs(want), np.abs(x + sj)), nrm)
    assert np.all(np.abs(y - want) <= 1e-6 * np.maximum(scale, 1e-30)), float(np.max(np.abs(y - want) / np.maximum(scale, 1e-30)))
    qa = qd.clone()
    s.prox_bang(qa, psi, qa, float(sigma))                 # y === q
    assert np.all(np.abs(qa.cpu().numpy() - want) <= 1e-6 * np.maximum(scale, 1e-30))
