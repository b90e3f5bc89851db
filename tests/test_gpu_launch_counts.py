"""Round 4: the one-launch psi(y) (the reducing kernel's last workgroup finishes: spx_fin_ticket / ObjFin, csrc/spx_objective.hip)
and the Binf group operators without the zero-fill launch of their deferred list (count words that alternate:
SpxSyncHeader::grp_deferred, csrc/spx_group.hip).  Tuning key 17 = 0 restores the launches of rounds 1-3: both forms must give
the same bits, leave their device state clean for the next call -- whatever operator runs in between -- and the oracle checks
of tests/test_gpu_parity.py already run on the new forms (the default)."""
import ctypes

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


def _dev(*arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0") for a in arrs]


def _key17(s, v):
    L = s._lib.load()
    s._lib.check(L.spx_ctx_set_tuning(s.context("cuda:0"), 17, v))


def _psis(s, xd, sd, n, gs):
    chi = s.NormLinf(1.0)
    lam_g = np.random.default_rng(3).uniform(0.5, 1.5, size=n // gs)
    return [s.shifted(s.shifted(s.NormL1(0.7), xd), sd), s.shifted(s.shifted(s.NormL0(0.7), xd), sd),
            s.shifted(s.shifted(s.RootNormLhalf(0.7), xd), sd), s.shifted(s.shifted(s.NormL1(0.7), xd, 0.9, chi), sd),
            s.shifted(s.shifted(s.NormL0(0.7), xd, 0.9, chi), sd), s.shifted(s.shifted(s.RootNormLhalf(0.7), xd, 0.9, chi), sd),
            s.shifted(s.shifted(s.IndBallL0(max(1, n // 3)), xd), sd), s.shifted(s.shifted(s.IndBallL0(max(1, n // 3)), xd, 0.9, chi), sd),
            s.shifted(s.shifted(s.GroupNormL2.uniform(lam_g.tolist(), gs), xd), sd),
            s.shifted(s.shifted(s.GroupNormL2.uniform(lam_g.tolist(), gs), xd, 0.9, chi), sd)]


@pytest.mark.parametrize("n", [1, 2, 50, 4096, 50_000, 1_000_000, 6_000_000])
def test_objective_one_launch_same_bits(s, n):
    """psi(y) of every operator family, feasible and infeasible y, one launch against three: the same double (the partials
    are added in the same order), also when an infeasible call is followed by a feasible one (the flag word is reset by the
    last workgroup) and when another operator has used the library's scratch in between."""
    import torch
    rng = np.random.default_rng(100 + n)
    gs = 1 if n < 50 else 50
    n = n // gs * gs
    x, sj = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n)
    xd, sd = _dev(x, sj)
    yd = _dev(rng.uniform(-0.3, 0.3, size=n))[0]
    ybad = yd * 10.0
    qd = _dev(rng.normal(size=n))[0]
    psis = _psis(s, xd, sd, n, gs)
    topr = s.shifted(s.shifted(s.IndBallL0(max(1, n // 7)), xd, 0.9, s.NormLinf(1.0)), sd)
    scratch_user = torch.empty_like(qd)
    try:
        for psi in psis:
            _key17(s, 0)
            want = [psi(yd), psi(ybad), psi(yd)]
            _key17(s, 1)
            got = [psi(yd), psi(ybad)]
            s.prox_bang(scratch_user, topr, qd, 1.0)   # writes all over spx_ctx::ws
            got.append(psi(yd))
            for a, b in zip(got, want):
                assert a == b or (np.isinf(a) and np.isinf(b)) or (a != a and b != b), (type(psi).__name__, n, got, want)
            for _ in range(3):                           # back to back, nothing in between
                assert psi(yd) == want[0] or (want[0] != want[0])
    finally:
        _key17(s, 1)


@pytest.mark.parametrize("n,ngroups", [(8192, 1), (10_000, 1), (100_001, 1), (1_000_000, 1), (3_000_000, 3), (16_000_000, 1),
                                        (16_777_216, 2), (17_000_000, 1), (4_194_304, 256), (4_210_688, 257)])
def test_objective_of_large_groups_one_launch(s, n, ngroups):
    """psi(y) of ONE group over the vector (shifted(NormL2(lambda), xk[, Delta, chi]): the reference's default GroupNormL2) and of a
    few large uniform groups: chunk sums, group sums and the final sum in ONE launch (k_obj_chunks' last workgroup) up to 256
    groups and 4096 chunks, three launches beyond (the last parameters); the same bits as with tuning key 17 = 0, feasible and
    infeasible points, and within 1e-12 of the value formed in numpy."""
    import torch
    rng = np.random.default_rng(700 + n)
    n = n // ngroups * ngroups
    x, sj = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n)
    yh = rng.uniform(-0.3, 0.3, size=n)
    xd, sd, yd = _dev(x, sj, yh)
    ybad = yd * 10.0
    lam = rng.uniform(0.5, 1.5, size=ngroups)
    H = s.GroupNormL2(lam.tolist()) if ngroups == 1 else s.GroupNormL2.uniform(lam.tolist(), n // ngroups)
    psis = [s.shifted(s.shifted(H, xd), sd), s.shifted(s.shifted(H, xd, 0.9, s.NormLinf(1.0)), sd)]
    want_np = float(np.sum(lam * np.sqrt(np.sum((((x + sj) + yh) ** 2).reshape(ngroups, -1), axis=1))))
    qd = _dev(rng.normal(size=n))[0]
    tmp = torch.empty_like(qd)
    other = s.shifted(s.shifted(s.IndBallL0(max(1, n // 7)), xd, 0.9, s.NormLinf(1.0)), sd)
    try:
        for psi in psis:
            _key17(s, 0)
            want = [psi(yd), psi(ybad), psi(yd)]
            _key17(s, 1)
            got = [psi(yd), psi(ybad)]
            s.prox_bang(tmp, other, qd, 1.0)               # writes all over the library's scratch
            got.append(psi(yd))
            for a, b in zip(got, want):
                assert a == b or (np.isinf(a) and np.isinf(b)), (type(psi).__name__, n, ngroups, got, want)
            assert abs(got[0] - want_np) <= 1e-12 * want_np, (got[0], want_np)
            for _ in range(3):
                assert psi(yd) == want[0]
    finally:
        _key17(s, 1)


def test_objective_one_launch_into_a_device_double_back_to_back(s):
    """200 psi(y) calls queued without a synchronisation, alternating feasible / infeasible points, values into a device
    array slot by slot: the tickets and the flag of one launch must be back at zero before the next one starts."""
    import torch
    n = 300_000
    rng = np.random.default_rng(5)
    xd, sd = _dev(rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n))
    yd = _dev(rng.uniform(-0.3, 0.3, size=n))[0]
    ybad = yd * 10.0
    psi = s.shifted(s.shifted(s.NormL1(0.7), xd, 0.9, s.NormLinf(1.0)), sd)
    want_ok, want_bad = psi(yd), psi(ybad)
    assert np.isfinite(want_ok) and np.isinf(want_bad)
    out = torch.full((200,), -1.0, dtype=torch.float64, device="cuda:0")
    L = s._lib.load(); ctx = s.context("cuda:0")
    for k in range(200):
        s._lib.check(L.spx_ctx_set_value_target(ctx, ctypes.c_void_p(out[k:].data_ptr())))
        psi(ybad if k % 3 == 1 else yd)
    s._lib.check(L.spx_ctx_set_value_target(ctx, None))
    got = out.cpu().numpy()
    for k in range(200):
        assert (np.isinf(got[k]) if k % 3 == 1 else got[k] == want_ok), (k, got[k])


@pytest.mark.parametrize("n", [2, 64, 3000, 50_000, 1_000_000, 1_000_001, 6_291_456, 7_000_000])
def test_prox_value_one_launch(s, orc, n):
    """prox! fused with the value of h at the result: when one launch covers the vector (aligned, even n) and its grid is at
    most 2048 workgroups, the last workgroup adds the partial sums (value_publish, csrc/spx_separable.hip) and the
    k_value_reduce launch is not queued.  y bit for bit and the value: equal to the two-launch form (key 17 = 0) -- both add
    a short list in the same order --, host and device-target forms, and within 1e-12 of the oracle's psi at the result."""
    import torch
    rng = np.random.default_rng(300 + n)
    x, sj, q = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)
    xd, sd, qd = _dev(x, sj, q)
    chi = s.NormLinf(1.0)
    psis = [("l1", s.shifted(s.shifted(s.NormL1(0.7), xd), sd)), ("l0", s.shifted(s.shifted(s.NormL0(0.7), xd), sd)),
            ("lhalf", s.shifted(s.shifted(s.RootNormLhalf(0.7), xd), sd)), ("l1", s.shifted(s.shifted(s.NormL1(0.7), xd, 0.9, chi), sd)),
            ("l0", s.shifted(s.shifted(s.NormL0(0.7), xd, 0.9, chi), sd))]
    out = torch.full((1,), -1.0, dtype=torch.float64, device="cuda:0")
    try:
        for name, psi in psis:
            _key17(s, 0)
            y0, v0 = s.prox_value(psi, qd, 1.1)
            y0 = y0.clone()
            _key17(s, 1)
            for rep in range(3):
                y1, v1 = s.prox_value(psi, qd, 1.1)
                assert torch.equal(y1.view(torch.int64), y0.view(torch.int64)), (name, n)
                assert v1 == v0, (name, n, v1, v0)
            with s.device_values(out):
                y2, v2 = s.prox_value(psi, qd, 1.1)
            assert v2 != v2 and float(out.item()) == v0 and torch.equal(y2.view(torch.int64), y0.view(torch.int64))
            yh = y0.cpu().numpy()
            term = {"l1": np.abs, "l0": lambda v: (v != 0).astype(float), "lhalf": lambda v: np.sqrt(np.abs(v))}[name]
            want = 0.7 * float(np.sum(term((x + sj) + yh)))
            assert abs(v0 - want) <= 1e-12 * max(1.0, abs(want)), (name, n, v0, want)
    finally:
        _key17(s, 1)


@pytest.mark.parametrize("gs", [2, 8, 16, 50, 128, 300])
def test_binf_deferred_list_without_the_zero_launch(s, gs):
    """Binf groups with a NON-EMPTY deferred list (degenerate brackets, |X_i| == Delta: the literal evaluation) several calls
    in a row, with other operators in between: the count word of a call must start at zero although nothing zeroes it in
    front of the call (the previous call's second launch did).  Against the three-launch form, bit for bit."""
    import torch
    rng = np.random.default_rng(2000 + gs)
    ng = 3000
    n = ng * gs
    scale = 0.6 / np.sqrt(gs)
    x = rng.normal(size=n) * scale
    sj = rng.uniform(-0.5, 0.5, size=n) * scale
    q = rng.normal(size=n) * scale
    delta = 0.3 * scale
    x[rng.integers(0, n, size=n // 20)] = delta           # entries ON the trust-region boundary: handed to the literal evaluation
    lam = rng.choice([2.0, 10.0, 0.7], size=ng)
    xd, sd, qd = _dev(x, sj, q)
    H = s.GroupNormL2.uniform(lam.tolist(), gs)
    psi = s.shifted(s.shifted(H, xd, delta, s.NormLinf(1.0)), sd)
    other = s.shifted(s.shifted(s.NormL1(0.7), xd, 0.9, s.NormLinf(1.0)), sd)
    topr = s.shifted(s.shifted(s.IndBallL0(n // 7), xd, 0.9, s.NormLinf(1.0)), sd)
    y0 = torch.empty_like(qd); y = torch.empty_like(qd); tmp = torch.empty_like(qd)
    try:
        _key17(s, 0)
        s.prox_bang(y0, psi, qd, 2.0)
        _key17(s, 1)
        for k in range(6):
            y.fill_(float("nan"))
            s.prox_bang(y, psi, qd, 2.0)
            assert torch.equal(y.view(torch.int64), y0.view(torch.int64)), (gs, k)
            if k % 2 == 0:
                s.prox_bang(tmp, topr, qd, 1.0)
                other(tmp)
        # a layout with an EMPTY list after one with a full one, and back
        easy = s.shifted(s.shifted(H, xd * 0.0, 1e6, s.NormLinf(1.0)), sd)
        e0 = torch.empty_like(qd); e1 = torch.empty_like(qd)
        s.prox_bang(e1, easy, qd, 2.0)
        s.prox_bang(y, psi, qd, 2.0)
        _key17(s, 0)
        s.prox_bang(e0, easy, qd, 2.0)
        assert torch.equal(e0.view(torch.int64), e1.view(torch.int64))
        assert torch.equal(y.view(torch.int64), y0.view(torch.int64))
        assert s._lib.load().spx_sync(s.context("cuda:0")) == 0
    finally:
        _key17(s, 1)


def test_fewer_launches_inside_and_after_a_graph(s):
    """A captured psi(y) + Binf prox! replays correctly (the tickets reset themselves; under a capture the deferred list keeps
    its zero-fill node) and eager calls after the capture still work."""
    import torch
    n = 40_000
    gs = 8
    rng = np.random.default_rng(77)
    xd, sd, qd = _dev(rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n))
    lam = rng.uniform(0.5, 1.5, size=n // gs)
    psi = s.shifted(s.shifted(s.GroupNormL2.uniform(lam.tolist(), gs), xd, 0.9, s.NormLinf(1.0)), sd)
    y = torch.empty_like(qd)
    out = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    s.prox_bang(y, psi, qd, 1.0)
    want_y = y.clone()
    want_v = psi(y)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        s.prox_bang(y, psi, qd, 1.0)                      # (scratch and state of this stream's context exist before the capture)
        with s.device_values(out):
            psi(y)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        s.prox_bang(y, psi, qd, 1.0)
        with s.device_values(out):
            psi(y)
    for _ in range(4):
        torch.cuda.synchronize()
        y.fill_(float("nan")); out.fill_(-1.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int64), want_y.view(torch.int64))
        assert float(out.item()) == want_v
    with torch.cuda.stream(side):                          # eager, on the captured context, after the capture
        y.fill_(float("nan"))
        s.prox_bang(y, psi, qd, 1.0)
        side.synchronize()
        assert torch.equal(y.view(torch.int64), want_y.view(torch.int64))
        assert psi(y) == want_v
    torch.cuda.synchronize()
