"""Stress runs that used to live in tools/fuzz_*.py, reduced to fixed seeds and sizes that fit the driver's GPU suite.

Each test names the kernel code it guards (csrc file + function).  They were confirmed to fail on planted faults
(tools/planted_faults.sh builds libspx with one guard flipped at a time and runs this file against it).
Parity bars: bit-exact for the L1 / L0 / top-r families; 1e-12 for RootNormLhalf(Box) / GroupNormL2(Binf) / NormL1B2 with
binary128 adjudication of everything above it (tests/arbiter.py) -- no fixed looser tolerance anywhere.
"""
import numpy as np
import pytest

import arbiter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def s():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.build()


def _dev(*arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in arrs]


def _bits(a, b):
    return np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))


def _binf_run_and_check(s, orc, h, x, sj, q, lam, sigma, delta, offs, what, no_size_hint=False):
    xd, sd, qd = _dev(x, sj, q)
    with np.errstate(all="ignore"):
        ref = orc.prox_group_l2_binf(q, x, sj, lam, sigma, delta, offsets=offs)
    psi = s.shifted(s.shifted(h, xd, delta, s.NormLinf(1.0)), sd)
    if no_size_hint:     # CSR offsets without the size bound: libspx then takes its general (memory-resident) kernel
        assert psi._layout.offsets is not None
        psi._layout.group_size = 0
    y = s.prox(psi, qd, sigma).cpu().numpy()
    assert np.array_equal(np.isnan(y), np.isnan(ref)) and np.array_equal(np.isfinite(y), np.isfinite(ref)), what
    fin = np.isfinite(ref)
    v = arbiter.check_group(orc, np.where(fin, y, 0.0), np.where(fin, ref, 0.0), q, x, sj, lam, sigma, offs, delta=delta, what=what)
    return y, ref, v


def _norms(S, x, offs):
    ng = len(offs) - 1
    nS = np.sqrt(np.add.reduceat(S * S, offs[:-1]))
    nX = np.sqrt(np.add.reduceat(x * x, offs[:-1]))
    mX = np.maximum.reduceat(np.abs(x), offs[:-1])
    return nS[:ng], nX[:ng], mX[:ng]


# ----------------------------------------------------------------------------------------------------------------------
# GroupNormL2Binf, reversed bracket with entries OUTSIDE the trust region (|xk_i| > Delta):
#   csrc/spx_group.hip  binf_literal_root / binf_literal_reg "piece iteration" n <- sigma*lambda - R(n) and the literal
#   bisection behind it (commits d6ad02e, 2795649, 16eae9b of round 1), in every kernel family that reaches them:
#   register tiles + deferred list (uniform <= 512), LDS-resident (700, 2500), general (5000), ragged CSR with and without
#   the size hint, gather-index groups.
# ----------------------------------------------------------------------------------------------------------------------
def _outside_tr_data(rng, sizes, xs):
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    scale = 1.0 / np.sqrt(np.repeat(sizes, sizes))
    x = rng.normal(size=n) * xs * scale
    sj = rng.uniform(-0.5, 0.5, size=n) * scale * float(rng.choice([0.0, 4.0]))
    q = rng.normal(size=n) * scale * float(rng.choice([4.0, 0.4, 12.0]))
    sigma = float(10.0 ** rng.uniform(-1, 1))
    delta = float(10.0 ** rng.uniform(-3, -0.5)) * xs / np.sqrt(np.median(sizes)) * 4
    S = (q + x) + sj
    nS, nX, mX = _norms(S, x, offs)
    lam = np.maximum(nS, 1e-3) * 10.0 ** rng.uniform(0, 1.5, size=sizes.size) / sigma / np.maximum(1e-3, 1.0 - np.minimum(nX, 0.95))
    rev = nS + sigma * lam * nX < sigma * lam
    out = mX > delta
    return offs, x, sj, q, lam, sigma, delta, rev & out


@pytest.mark.parametrize("layout", ["uniform1", "uniform2", "uniform3", "uniform4", "uniform8", "uniform16", "uniform40", "uniform128",
                                    "uniform300", "uniform512", "lds700", "lds2500", "general5000", "ragged_hint", "ragged_nohint",
                                    "gather"])
def test_binf_reversed_bracket_entries_outside_trust_region(s, orc, layout):
    rng = np.random.default_rng(2025 + sum(map(ord, layout)))
    seen = nonzero = arbitrated = total = 0
    reps = 4
    for rep in range(reps):
        if layout.startswith(("uniform", "lds", "general")):
            gs = int(layout.lstrip("uniformldsgenral"))
            ng = 6000 if gs <= 16 else (1200 if gs <= 128 else (300 if gs <= 512 else 40))
            sizes = np.full(ng, gs)
        else:
            ng = 3000
            sizes = rng.integers(1, 41, size=ng)
        xs = float(rng.choice([0.05, 0.2, 0.5]))
        offs, x, sj, q, lam, sigma, delta, regime = _outside_tr_data(rng, sizes, xs)
        if layout == "gather":          # the same ranges listed back to front: explicit index vectors -> gather kernel
            h = s.GroupNormL2(lam.tolist(), [list(range(int(offs[g + 1]) - 1, int(offs[g]) - 1, -1)) for g in range(ng)])
        elif layout.startswith("ragged"):
            h = s.GroupNormL2.ragged(lam.tolist(), offs)
        else:
            h = s.GroupNormL2.uniform(lam.tolist(), int(sizes[0]))
        y, ref, v = _binf_run_and_check(s, orc, h, x, sj, q, lam, sigma, delta, offs, "%s rep %d" % (layout, rep),
                                        no_size_hint=(layout == "ragged_nohint"))
        arbitrated += v.n_checked
        seen += int(regime.sum())
        total += ng
        zref = arbiter.zero_pattern(ref, x, sj, offs)
        nonzero += int((regime & ~zref).sum())
        # the zero / non-zero decision of every group in the regime agrees with the oracle, except where the arbiter had to
        # be called in (and sided with the GPU)
        zg = arbiter.zero_pattern(y, x, sj, offs)
        assert int(np.sum((zg != zref) & regime)) <= v.n_checked, (layout, rep, int(np.sum((zg != zref) & regime)), v)
    assert seen > total // 4, (layout, seen, total)       # the regime really occurs ...
    if not layout.startswith(("lds", "general")):
        assert nonzero > 0, (layout, seen, nonzero)       # ... including groups that take the non-zero literal branch
    assert arbitrated <= max(10, seen // 20), (layout, arbitrated, seen)


# ----------------------------------------------------------------------------------------------------------------------
# GroupNormL2Binf, reversed brackets in general (groups that are, or are about to be, zero under a strong sigma*lambda):
#   csrc/spx_group.hip  binf_root in-kernel decisions "xk == 0 on the group" and "every |xk_i| < Delta", and the deferred
#   literal list for what is left (entries on / outside the trust region, |xk_i| == Delta exactly).
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gs", [1, 2, 3, 8, 16, 31, 64, 128, 200, 256, 512, 700, 2500])
def test_binf_reversed_bracket_regimes(s, orc, gs):
    rng = np.random.default_rng(77 + gs)
    ng = 1500 if gs <= 64 else (400 if gs <= 512 else 24)
    n = ng * gs
    offs = np.arange(0, n + 1, gs)
    nrev = arbitrated = 0
    for rep in range(6):
        xs = float([0.0, 1e-8, 0.02, 0.3, 1.0, 0.3][rep]) / np.sqrt(gs)
        x = rng.normal(size=n) * xs
        zero_g = rng.random(ng) < rng.choice([0.0, 0.5, 0.95])
        x = np.where(np.repeat(zero_g, gs), 0.0, x)
        sj = rng.uniform(-0.5, 0.5, size=n) * float(rng.choice([0.0, 1.0, 1e-3]))
        q = rng.normal(size=n) * float(rng.choice([1.0, 1e-3, 30.0]))
        sigma = float(10.0 ** rng.uniform(-2, 1.5))
        delta = float(10.0 ** rng.uniform(-3, 2))
        if rep % 3 == 2 and xs > 0 and np.any(np.abs(x) > 0):   # some entries exactly on the trust-region boundary
            delta = float(np.abs(x[np.abs(x) > 0][0]))
        S = (q + x) + sj
        nS, nX, _ = _norms(S, x, offs)
        lam = np.maximum(nS, 1e-3) * 10.0 ** rng.uniform(-1, 2, size=ng) / sigma
        h = s.GroupNormL2.uniform(lam.tolist(), gs)
        y, ref, v = _binf_run_and_check(s, orc, h, x, sj, q, lam, sigma, delta, offs, "reversed gs %d rep %d" % (gs, rep))
        nrev += int((nS + sigma * lam * nX < sigma * lam).sum())
        arbitrated += v.n_checked
    assert nrev > ng, (gs, nrev)
    assert arbitrated <= max(6, 6 * ng // 20), (gs, arbitrated)   # (the reference's own Float64 error next to the pole, see arbiter.py)


# ----------------------------------------------------------------------------------------------------------------------
# GroupNormL2Binf on lattice data incl. constant groups: activity boundaries |tau S - X| = Delta, exact roots, exact
# zero froot(lmax): csrc/spx_group.hip binf_root "root at the bracket's end" / "knife edge at lmin" paths.
# (test_gpu_parity.py::test_group_binf_lattice_and_zero_x covers x = 0; this one adds the constant-group variant and the
#  8-lane / 32-lane tiles.)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gs", [3, 16, 32, 100, 128, 256, 300, 1024])
def test_binf_lattice_constant_groups(s, orc, gs):
    rng = np.random.default_rng(11 + gs)
    for rep in range(6):
        ng = 200 if gs <= 300 else 30
        n = ng * gs
        x = rng.integers(-8, 9, size=n) / 4.0
        sj = rng.integers(-2, 3, size=n) / 4.0
        q = rng.integers(-12, 13, size=n) / 4.0
        if rep % 3 == 1:
            x[: n // 2] = 0.0
        if rep % 3 == 2:
            q[:] = np.repeat(rng.integers(-4, 5, size=ng) / 4.0, gs)   # constant groups
        lam = rng.choice([0.0, 0.25, 0.5, 1.0, 2.0, 8.0], size=ng)
        sigma = float(rng.choice([0.25, 0.5, 1.0, 2.0]))
        delta = float(rng.choice([0.25, 0.5, 1.0, 3.0]))
        h = s.GroupNormL2.uniform(lam.tolist(), gs)
        _binf_run_and_check(s, orc, h, x, sj, q, lam, sigma, delta, np.arange(0, n + 1, gs), "lattice gs %d rep %d" % (gs, rep))


# ----------------------------------------------------------------------------------------------------------------------
# Top-r on 8-byte-misaligned views, every mix of alignments (csrc/spx_select.hip: the one-launch forms with 8-byte accesses --
# v in registers, v in LDS; the peeled sample-predicted path when all four vectors are off by 8 bytes is covered at n > 2^22
# by test_gpu_parity.py::test_indball_l0_misaligned_views_fast_path).
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 1000, 70001, (1 << 21) + 17])
def test_topr_misaligned_views(s, orc, n):
    import torch
    rng = np.random.default_rng(4 + n)
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = np.round(rng.normal(size=n) * 16) / 16
    mk = lambda a: torch.cat([torch.zeros(1, dtype=torch.float64), torch.from_numpy(a)]).cuda()[1:]
    top = orc.TopR(q, x, sj)   # (the reference's sortperm once, every r from it)
    for mis in ((True, True, True), (True, False, False), (False, False, True)):
        xd = mk(x) if mis[0] else torch.from_numpy(x).cuda()
        sd = mk(sj) if mis[1] else torch.from_numpy(sj).cuda()
        qd = mk(q) if mis[2] else torch.from_numpy(q).cuda()
        for r in sorted({1, max(1, n // 3), n}):
            ref = top.prox(r, 0.8)
            psi = s.shifted(s.shifted(s.IndBallL0(r), xd, 0.8, s.NormLinf(1.0)), sd)
            y = s.prox(psi, qd, 1.0).cpu().numpy()
            yv = mk(np.zeros(n))
            s.prox_bang(yv, psi, qd, 1.0)
            assert _bits(y, ref) and _bits(yv.cpu().numpy(), ref), (n, mis, r)


# ----------------------------------------------------------------------------------------------------------------------
# Top-r, one-launch select (k_sel_coop): its first digit is FOLDED (csrc/spx_select.hip fold_digit: 62 binades around 1.0 x
# 64 mantissa steps, catch-all bins below 2^-31 and above 2^30).  Data that puts the threshold into a catch-all bin (restart
# with the plain top digit), onto a bin edge, into ties at key 0 / Inf / NaN, or all of the vector into one bin.
# Sizes: v in registers (grids of 35, 74 and 128 workgroups), v in LDS (k_sel_lds: the first digit is histogrammed while the
# vectors are loaded), the form that parks v in y (mixed alignment, n > 2^21, k_sel_lds switched off).
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["tiny", "huge", "tiny+normal", "normal+huge", "zeros", "inf_nan", "constant", "binade_edges",
                                  "wide_exponents"])
@pytest.mark.parametrize("n", [70_001, 300_000, 1_000_003, (1 << 21) - 5, (1 << 21) + 4099, -((1 << 21) + 4099)])
def test_topr_folded_first_digit(s, orc, kind, n):
    lds = n > 0                      # (a negative size: the same size with k_sel_lds off)
    n = abs(n)
    s._lib.check(s._lib.load().spx_ctx_set_tuning(s.context("cuda:0"), 11, 1 if lds else 0))
    try:
        _folded_first_digit_cases(s, orc, kind, n)
    finally:
        s._lib.check(s._lib.load().spx_ctx_set_tuning(s.context("cuda:0"), 11, 1))


def _folded_first_digit_cases(s, orc, kind, n):
    import torch
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
    elif kind == "inf_nan":
        q = q.copy()
        q[rng.integers(0, n, size=40)] = np.inf
        q[rng.integers(0, n, size=40)] = -np.inf
        q[rng.integers(0, n, size=25)] = np.nan
    elif kind == "constant":
        x, sj, q = np.full(n, 0.5), np.full(n, 0.25), np.full(n, -2.0)
    elif kind == "binade_edges":  # |v| on the edges of the folded bins: 2^k (1 + j/64), exactly
        k = rng.integers(-3, 4, size=n); j = rng.integers(0, 64, size=n)
        v = 2.0 ** k * (1.0 + j / 64.0) * rng.choice([-1.0, 1.0], size=n)
        x, sj = np.zeros(n), np.zeros(n); q = v
    elif kind == "wide_exponents":
        e = rng.integers(-80, 80, size=n)
        x, sj, q = x * 2.0 ** e, sj * 2.0 ** e, q * 2.0 ** e
    mixed = n > (1 << 21)  # one vector 8 bytes off: the one-launch form that parks v in y
    mk = lambda a: torch.cat([torch.zeros(1, dtype=torch.float64), torch.from_numpy(a)]).cuda()[1:]
    xd, sd, qd = (mk(x) if mixed else torch.from_numpy(x).cuda()), torch.from_numpy(sj).cuda(), torch.from_numpy(q).cuda()
    with np.errstate(all="ignore"):
        top = orc.TopR(q, x, sj)
    for r in sorted({1, 2, 26, 70, n // 100, n // 3, n // 2, int(0.7 * n), n - 1}):
        with np.errstate(all="ignore"):
            ref = top.prox(r, 0.8)
        psi = s.shifted(s.shifted(s.IndBallL0(r), xd, 0.8, s.NormLinf(1.0)), sd)
        y = s.prox(psi, qd, 1.0).cpu().numpy()
        assert _bits(y, ref), (kind, n, r, int(np.sum(y.view(np.int64) != ref.view(np.int64))))


# ----------------------------------------------------------------------------------------------------------------------
# Lattice data (multiples of 1/4) through the separable operators, iprox!, NormL1B2 and top-r: exact ties with every
# threshold and bound (csrc/spx_separable.hip functors; spx_b2.hip piece roots hit exactly; spx_select.hip index digits).
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", range(4))
def test_lattice_separable_b2_topr(s, orc, seed):
    import torch
    rng = np.random.default_rng(3 + seed)
    for rep in range(8):
        n = int(rng.integers(1, 20000))
        x = rng.integers(-8, 9, size=n) / 4.0; sj = rng.integers(-4, 5, size=n) / 4.0; q = rng.integers(-12, 13, size=n) / 4.0
        lam = float(rng.choice([0.0, 0.25, 0.5, 1.0, 2.0])); sigma = float(rng.choice([0.25, 0.5, 1.0, 2.0, 4.0]))
        lo = float(rng.choice([-2.0, -1.0, -0.5, 0.0])); up = float(rng.choice([0.0, 0.5, 1.0, 2.0]))
        xd, sd, qd = _dev(x, sj, q)
        with np.errstate(all="ignore"):
            for H, nm in ((s.NormL1, "l1"), (s.NormL0, "l0")):
                y = s.prox(s.shifted(s.shifted(H(lam), xd), sd), qd, sigma).cpu().numpy()
                assert _bits(y, getattr(orc, "prox_" + nm)(q, x, sj, lam, sigma)), (nm, seed, rep)
                y = s.prox(s.shifted(s.shifted(H(lam), xd, lo, up), sd), qd, sigma).cpu().numpy()
                assert _bits(y, getattr(orc, "prox_%s_box" % nm)(q, x, sj, lam, sigma, lo, up)), (nm, "box", seed, rep)
                d = rng.choice([-2.0, -1.0, 0.0, 0.5, 1.0, 4.0], size=n)
                y = s.iprox(s.shifted(s.shifted(H(lam), xd, lo, up), sd), qd, _dev(d)[0]).cpu().numpy()
                assert _bits(y, getattr(orc, "iprox_%s_box" % nm)(q, d, x, sj, lam, lo, up)), (nm, "iprox", seed, rep)
            # RootNormLhalf: unboxed has no candidate choice -> arbiter only
            y = s.prox(s.shifted(s.shifted(s.RootNormLhalf(lam), xd), sd), qd, sigma).cpu().numpy()
            arbiter.check_lhalf(orc, y, orc.prox_lhalf(q, x, sj, lam, sigma), q, x, sj, lam, sigma, what="lattice lhalf")
            # boxed: on a lattice two candidates often have EXACTLY the same objective value; findmin's first-minimum rule then
            # depends on last-bit rounding of the candidate values (must match "except on exact ties").  An element
            # whose candidate differs but whose objective (in extended precision) equals the oracle's is such a tie.
            y = s.prox(s.shifted(s.shifted(s.RootNormLhalf(lam), xd, lo, up), sd), qd, sigma).cpu().numpy()
            ref = orc.prox_lhalf_box(q, x, sj, lam, sigma, lo, up)
            sc = arbiter.lhalf_scale(ref, x, sj, q)
            m = np.abs(y - ref) > 1e-12 * np.maximum(sc, 1e-300)
            if m.any():
                ld_ = np.longdouble
                f = lambda t: (ld_(t) - ld_(q)) ** 2 / 2 / ld_(sigma) + ld_(lam) * np.sqrt(np.abs(ld_(t) + ld_(x + sj)))
                tie = np.abs(f(y) - f(ref)) <= 4e-16 * np.maximum(np.abs(f(ref)), 1e-300)
                arbiter.check_lhalf(orc, np.where(m & tie, ref, y), ref, q, x, sj, lam, sigma, box=(lo, up), what="lattice lhalf_box")
            r = int(rng.integers(1, n + 1))
            y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.75, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
            assert _bits(y, orc.prox_indball_l0_binf(q, x, sj, r, 0.75)), ("indball", seed, rep)
            delta = float(rng.choice([0.5, 2.0, 50.0]))
            y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, sigma).cpu().numpy()
            ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, 1.0)
            assert np.max(np.abs(y - ref)) <= 1e-12 * max(np.linalg.norm(ref), np.linalg.norm(x), 1.0), ("b2", seed, rep)


# ----------------------------------------------------------------------------------------------------------------------
# ShiftedNormL1B2, streaming form of the one-launch kernel (n > 2^21: csrc/spx_b2.hip k_b2_coop<false>): the sample's root as
# a second trial of the first pass, the bracket closed at the unevaluated a-priori bound (x = 0: the bound IS the root --
# a version without that bisected for 30 passes), the quadratic stopping rule.  Against the Float64 oracle, 1e-12 of the norms.
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("form", ["streaming", "lds"])
@pytest.mark.parametrize("kind", ["normal", "x=0", "lattice8", "q*0.01", "x*0.05", "sparse_x"])
def test_b2_streaming_form_scenarios(s, orc, kind, form):
    """form = "streaming": tuning key 12 = 0 -- this size (kept small for the CPU oracle) then takes the two-pass streaming form
    as every n > 2^22 does; "lds": the default at this size, xk parked in LDS (k_b2_coop<.., LDSX>), same scenarios."""
    L = s._lib.load()
    s._lib.check(L.spx_ctx_set_tuning(s.context("cuda:0"), 12, 0 if form == "streaming" else 1))
    try:
        _b2_scenarios(s, orc, kind)
    finally:
        s._lib.check(L.spx_ctx_set_tuning(s.context("cuda:0"), 12, 1))


def _b2_scenarios(s, orc, kind):
    rng = np.random.default_rng(sum(map(ord, kind)))
    n = 2_300_001
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    if kind == "x=0": x[:] = 0.0; sj[:] = 0.0
    elif kind == "lattice8": x, sj, q = (np.round(v * 8) / 8 for v in (x, sj, q))
    elif kind == "q*0.01": q *= 0.01
    elif kind == "x*0.05": x *= 0.05
    elif kind == "sparse_x": x[rng.random(n) < 0.9] = 0.0
    xd, sd, qd = _dev(x, sj, q)
    for lam, delta in ((0.01, 1e-3), (1.0, 1.0), (30.0, 1.0), (1.0, 1000.0), (0.01, 1e6)):
        with np.errstate(all="ignore"):
            ref = orc.prox_l1_b2(q, x, sj, lam, 1.0, delta, 1.0)
        y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, 1.0).cpu().numpy()
        scale = max(np.linalg.norm(ref), np.linalg.norm(x), np.linalg.norm(sj + q), 1e-300)
        assert float(np.max(np.abs(y - ref))) <= 1e-12 * scale, (kind, lam, delta, float(np.max(np.abs(y - ref))) / scale)
        # y === q (no speculative stores, the final pass reads q before it writes y)
        q2 = qd.clone()
        s.prox_bang(q2, s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), q2, 1.0)
        assert float(np.max(np.abs(q2.cpu().numpy() - ref))) <= 1e-12 * scale, (kind, lam, delta, "aliased")


# ----------------------------------------------------------------------------------------------------------------------
# ShiftedNormL1B2, one-launch forms (csrc/spx_b2.hip k_b2_coop): the register-resident form holds 8192 elements per workgroup
# (512 lanes x 16) and exchanges partial sums through words that carry their own ready flag (b2_put); sizes either side of
# one / two / many / all 256 workgroups, of the switch to the form with xk in LDS (above 2^21) and to the streaming form (above
# 2^22), each with the trust region active,
# inactive twice in a row (the second call stores y in its first pass), active again, and y aliased to q.  Two sets of
# words alternate between launches and a launch zeroes the other set: the sequence of sizes exercises that too.
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 511, 8191, 8192, 8193, 16385, 65_537, 1_000_003, (1 << 21) - 1, 1 << 21, (1 << 21) + 2,
                               (1 << 22) - 1, 1 << 22, (1 << 22) + 2])
def test_b2_one_launch_forms_at_their_boundaries(s, orc, n):
    import torch
    rng = np.random.default_rng(77 + n)
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    xd, sd, qd = _dev(x, sj, q)
    nrm = float(np.linalg.norm(x))
    for lam, sigma, delta in ((1.0, 1.0, 1.0), (1.0, 1.0, 1e9), (1.0, 1.0, 1e9), (0.3, 0.7, 0.5 * nrm + 1e-3), (2.0, 1.0, 1e-3)):
        psi = s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd)
        ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, 1.0)
        scale = max(np.linalg.norm(ref), nrm, 1e-300)
        y = s.prox(psi, qd, sigma).cpu().numpy()
        assert np.max(np.abs(y - ref)) <= 1e-12 * scale, (n, lam, sigma, delta)
        qa = qd.clone()
        s.prox_bang(qa, psi, qa, sigma)
        assert np.max(np.abs(qa.cpu().numpy() - ref)) <= 1e-12 * scale, (n, lam, sigma, delta, "aliased")
    # a NaN anywhere poisons the norm, as in the reference (every entry of the scaled branch becomes NaN)
    if n >= 3:
        qn = q.copy(); qn[n // 2] = np.nan
        psi = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, s.NormL2(1.0)), sd)
        y = s.prox(psi, torch.from_numpy(qn).cuda(), 1.0).cpu().numpy()
        ref = orc.prox_l1_b2(qn, x, sj, 1.0, 1.0, 1.0, 1.0)
        assert np.array_equal(np.isnan(y), np.isnan(ref)), n


def test_b2_streaming_form_tiles_on_demand(s, orc):
    """n = 1.3e7: enough tiles per workgroup for the passes that take their tiles from an atomic counter (csrc/spx_b2.hip: the
    storing pass; the speculative pass of a call that follows an inactive one).  Sequence on one context: active, inactive,
    inactive (speculation right: one pass), active (speculation wrong: the ordinary passes follow, static mapping), a barely
    inactive / barely active pair around Delta = chi(y), y === q.  Against the Float64 oracle, 1e-12 of the norms; the inactive
    results bit for bit (y = ProjB(-xk) - sj is elementwise)."""
    import torch
    n = 13_000_001   # (>= 8 tiles of 6144 elements per workgroup of a 256-workgroup grid: 12.6e6)
    rng = np.random.default_rng(2026)
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
    xd, sd, qd = _dev(x, sj, q)
    nrm = float(np.linalg.norm(x))
    # chi(y) of the unscaled result: Delta just above / below it
    y1 = orc.prox_l1_b2(q, x, sj, 1.0, 1.0, 1e12, 1.0)
    chi1 = float(np.linalg.norm(sj + y1))
    seq = [(1.0, 1.0), (1.0, 1e12), (1.0, 1e12), (1.0, 0.3 * nrm), (1.0, chi1 * (1 + 1e-9)), (1.0, chi1 * (1 - 1e-9)), (1.0, 1e12), (0.2, 5.0)]
    refs = {(1.0, 1e12): y1}   # (one oracle run per distinct (lambda, Delta))
    for lam, delta in seq:
        psi = s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd)
        if (lam, delta) not in refs:
            refs[(lam, delta)] = orc.prox_l1_b2(q, x, sj, lam, 1.0, delta, 1.0)
        ref = refs[(lam, delta)]
        y = s.prox(psi, qd, 1.0).cpu().numpy()
        scale = max(np.linalg.norm(ref), nrm, np.linalg.norm(sj + q))
        assert float(np.max(np.abs(y - ref))) <= 1e-12 * scale, (lam, delta, float(np.max(np.abs(y - ref))) / scale)
        if delta == 1e12:
            assert _bits(y, ref), "inactive trust region: elementwise result"
    qa = qd.clone()
    psi = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, s.NormL2(1.0)), sd)
    s.prox_bang(qa, psi, qa, 1.0)
    ref = refs[(1.0, 1.0)]
    assert float(np.max(np.abs(qa.cpu().numpy() - ref))) <= 1e-12 * max(np.linalg.norm(ref), nrm), "aliased"
    s._lib.check(s._lib.load().spx_sync(s.context("cuda:0")))


def test_b2_alternating_sizes_share_the_exchange_words(s, orc):
    """Calls of different sizes on one context, interleaved: the partial-sum words of a 256-workgroup launch must be clean
    again when a 2-workgroup launch (and then another 256-workgroup one) comes to use the same set."""
    rng = np.random.default_rng(5)
    data = {}
    for n in (12_000, (1 << 21) - 7, 3_000_001, 70_000):
        x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
        data[n] = (x, sj, q) + tuple(_dev(x, sj, q)) + (orc.prox_l1_b2(q, x, sj, 1.0, 1.0, 1.0, 1.0),)
    order = [12_000, (1 << 21) - 7, 12_000, 3_000_001, (1 << 21) - 7, 70_000, 12_000, 70_000, 3_000_001, 12_000]
    for n in order:
        x, sj, q, xd, sd, qd, ref = data[n]
        psi = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, s.NormL2(1.0)), sd)
        y = s.prox(psi, qd, 1.0).cpu().numpy()
        assert np.max(np.abs(y - ref)) <= 1e-12 * max(np.linalg.norm(ref), np.linalg.norm(x)), n


def test_b2_integer_lattices_exact_roots(s, orc):
    """Integer data, lambda = sigma = Delta = 1 and friends: froot vanishes exactly at bracket ends and breakpoints (the case
    that exposed a defect of the ORACLE's bracket search, tools/fuzz_r2_onelaunch.py: GPU right, restatement wrong).  400 small
    problems through every form the size selects."""
    import torch
    rng = np.random.default_rng(82)
    for rep in range(400):
        n = int(rng.integers(1, 80))
        lev = int(rng.choice([1, 1, 2, 4]))
        x = np.round(rng.normal(size=n) * lev) / lev; sj = np.round(rng.uniform(-0.5, 0.5, size=n) * lev) / lev
        q = np.round(rng.normal(size=n) * lev) / lev
        lam, sigma, delta = float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.25, 0.5, 1.0, 2.0, 4.0]))
        xd, sd, qd = _dev(x, sj, q)
        ref = orc.prox_l1_b2(q, x, sj, lam, sigma, delta, 1.0)
        y = s.prox(s.shifted(s.shifted(s.NormL1(lam), xd, delta, s.NormL2(1.0)), sd), qd, sigma).cpu().numpy()
        assert np.max(np.abs(y - ref)) <= 1e-12 * max(np.linalg.norm(ref), np.linalg.norm(x), 1e-300), (rep, n, lev, lam, sigma, delta)


# ----------------------------------------------------------------------------------------------------------------------
# Top-r, tie mode of the sample-predicted pipeline (round 3; csrc/spx_select.hip: FastState::tie, ClassCount, k_s2_tail).  The
# threshold key is shared by per cents of the vector (or by all of it): its members are counted per wavefront, stored on the
# sample's guess of the cut, the index cut comes from a prefix sum and the mis-guessed index range is rewritten; moderately
# tied keys go through the radix select over the candidate records.  Against the CPU oracle (stable sort), bit for bit;
# odd n and views from an odd element put members of the classes into the stragglers' slots; y === q takes the form
# without speculative stores.
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("form", ["pipeline", "one launch"])
@pytest.mark.parametrize("kind", ["constant", "two values", "lattice 1/4", "lattice 2^-8", "90% zeros", "sorted lattice",
                                  "lattice + noise on half"])
def test_topr_tie_mode_against_exact_select(s, orc, kind, form):
    """form = "pipeline": tuning key 11 = 0, so that this size (kept small for the CPU oracle) takes the sample-predicted path
    as every n > 2^22 does; "one launch": the default at this size, v parked in LDS (k_sel_lds) -- the same data through its
    key and index digits."""
    import torch
    s._lib.check(s._lib.load().spx_ctx_set_tuning(s.context("cuda:0"), 11, 0 if form == "pipeline" else 1))
    try:
        _tie_mode_cases(s, orc, kind)
    finally:
        s._lib.check(s._lib.load().spx_ctx_set_tuning(s.context("cuda:0"), 11, 1))


def _tie_mode_cases(s, orc, kind):
    import torch
    rng = np.random.default_rng(sum(map(ord, kind)))
    n = 2_400_001 + int(rng.integers(0, 7)) * 2          # odd: the last element is a straggler of wave 0
    g = rng.normal(size=n)
    if kind == "constant": q = np.full(n, -2.0)
    elif kind == "two values": q = np.where(g > 0.5, 1.5, -0.75)
    elif kind == "lattice 1/4": q = np.round(g * 4) / 4
    elif kind == "lattice 2^-8": q = np.round(g * 256) / 256
    elif kind == "90% zeros": q = np.where(rng.random(n) < 0.9, 0.0, g)
    elif kind == "sorted lattice": q = np.sort(np.round(g * 4) / 4)
    else: q = np.where(np.arange(n) % 2 == 0, np.round(g * 2) / 2, g)
    x = np.zeros(n); sj = np.zeros(n)
    top = orc.TopR(q, x, sj)
    for head in (0, 1):                                   # head = 1: views from an odd element (8 bytes off a 16-byte boundary)
        mk = (lambda a: torch.cat([torch.zeros(1, dtype=torch.float64), torch.from_numpy(a)]).cuda()[1:]) if head else \
            (lambda a: torch.from_numpy(a).cuda())
        xd, sd, qd = mk(x), mk(sj), mk(q)
        for r in (n // 100, n // 2, n - n // 20, 3):
            ref = top.prox(r, 1.25)
            y = mk(np.full(n, np.nan))
            psi = s.shifted(s.shifted(s.IndBallL0(r), xd, 1.25, s.NormLinf(1.0)), sd)
            s.prox_bang(y, psi, qd, 1.0)
            got = y.cpu().numpy()
            assert np.array_equal(got.view(np.int64), ref.view(np.int64)), (kind, head, r, int(np.sum(got.view(np.int64) != ref.view(np.int64))))
            if r == n // 2:
                qa = qd.clone() if not head else mk(q.copy())
                s.prox_bang(qa, psi, qa, 1.0)              # y === q: nothing is stored before the cut is known
                assert np.array_equal(qa.cpu().numpy().view(np.int64), ref.view(np.int64)), (kind, head, r, "aliased")
    s._lib.check(s._lib.load().spx_sync(s.context("cuda:0")))
