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
    with pytest.raises(TypeError, match="no Float32 form"):
        psi(x32)                                               # psi(y): Float64 only
