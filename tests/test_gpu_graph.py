"""A solver iteration captured into a graph (hipGraph through torch.cuda.CUDAGraph) and replayed on new data.

The kernels that synchronise inside one launch (top-r, ShiftedNormL1B2) keep device state between launches that the host
tracks by alternating sets; a captured launch would replay one set for ever.  csrc: spx_ctx::graph_safe -- once a call finds
its stream capturing, every such launch is preceded by a memset node for exactly the state it uses.  Calls that would
synchronise or allocate refuse to run while capturing."""
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


def _bits(a, b):
    return np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))


@pytest.mark.parametrize("n", [6_000, 50_000, 1_000_000, 2_600_000, 4_300_000, 6_400_000, pytest.param(15_000_000, marks=pytest.mark.soak)])
def test_iteration_in_a_graph_replays_on_new_data(s, orc, n):
    """n = 6e3: one-workgroup top-r, one-workgroup B2; 5e4 / 1e6: the register-resident one-launch forms; 2.6e6: the forms that
    park a vector in LDS (256 workgroups); 4.3e6: top-r with 16 elements per lane in LDS and 8 in registers and the streaming B2
    form; 6.4e6: the sample-predicted top-r pipeline; 1.5e7: the B2
    passes that take their tiles from an atomic counter (zeroed by a node of the graph) -- the iteration then also calls B2 with
    an inactive trust region twice, so that the speculative pass is wrong once and right once per replay."""
    import torch
    rng = np.random.default_rng(n)
    ng = n // 128
    m = ng * 128
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n)
    lam_g = rng.uniform(0.5, 1.5, size=ng)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(sj).cuda()
        qd = torch.zeros(n, dtype=torch.float64, device="cuda")
        ys = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(6)]
        big = n >= 10_000_000
        val = torch.zeros(1, dtype=torch.float64, device="cuda")
        chi = s.NormLinf(1.0)
        r = max(1, n // 50)
        psi_box = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, chi), sd)
        psi_top = s.shifted(s.shifted(s.IndBallL0(r), xd, 0.8, chi), sd)
        psi_b2 = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, s.NormL2(1.0)), sd)
        psi_b2_in = s.shifted(s.shifted(s.NormL1(1.0), xd, 1e12, s.NormL2(1.0)), sd)
        psi_grp = s.shifted(s.shifted(s.GroupNormL2.uniform(torch.from_numpy(lam_g).cuda(), 128), xd[:m], 1.0, chi), sd[:m])

        def iteration():
            s.prox_bang(ys[0], psi_box, qd, 1.0)
            with s.device_values(val):
                psi_box(ys[0])
            s.prox_bang(ys[1], psi_top, qd, 1.0)
            s.prox_bang(ys[2], psi_b2, qd, 1.0)
            s.prox_bang(ys[3][:m], psi_grp, qd[:m], 1.0)
            if big:
                s.prox_bang(ys[4], psi_b2_in, qd, 1.0)   # follows an active call: no speculation
                s.prox_bang(ys[5], psi_b2_in, qd, 1.0)   # follows an inactive call: the speculative pass is right
                # (and the next replay's first B2 call follows an inactive one with an active trust region: speculation wrong)

        qd.copy_(torch.from_numpy(rng.normal(size=n)))
        iteration(); iteration()           # warm-up on the capture stream: the workspaces reach their sizes
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        iteration()

    def check(q, what):
        torch.cuda.synchronize()
        assert _bits(ys[0].cpu().numpy(), orc.prox_l1_box(q, x, sj, 1.0, 1.0, -1.0, 1.0)), what
        exp = orc.obj_box("l1", ys[0].cpu().numpy(), x, sj, 1.0, -1.0, 1.0)
        got = float(val.item())
        assert got == exp or abs(got - exp) <= 1e-12 * abs(exp), (what, got, exp)
        assert _bits(ys[1].cpu().numpy(), orc.prox_indball_l0_binf(q, x, sj, r, 0.8)), what
        ref = orc.prox_l1_b2(q, x, sj, 1.0, 1.0, 1.0, 1.0)
        assert np.max(np.abs(ys[2].cpu().numpy() - ref)) <= 1e-12 * max(np.linalg.norm(ref), np.linalg.norm(x)), what
        ref = orc.prox_group_l2_binf(q[:m], x[:m], sj[:m], lam_g, 1.0, 1.0, gsize=128)
        err = np.max(np.abs(ys[3][:m].cpu().numpy() - ref))
        assert err <= 1e-9 * max(1.0, np.max(np.abs(ref))), (what, err)   # (the arbiter-based bar lives in test_gpu_parity.py)
        if big:
            ref = orc.prox_l1_b2(q, x, sj, 1.0, 1.0, 1e12, 1.0)
            assert _bits(ys[4].cpu().numpy(), ref) and _bits(ys[5].cpu().numpy(), ref), what

    for rep in range(4 if n < 1_000_000 else 2):   # (each check is five oracle calls on n elements)
        q = rng.normal(size=n) * (1.0 + rep)
        qd.copy_(torch.from_numpy(q))
        for t in ys:
            t.fill_(-777.0)
        torch.cuda.synchronize()
        g.replay()
        check(q, "replay %d" % rep)
    # eager calls on the same context after the replays (graph-safe mode is sticky), then one more replay
    q = rng.normal(size=n)
    qd.copy_(torch.from_numpy(q))
    with torch.cuda.stream(side):
        iteration(); iteration()
    check(q, "eager after replays")
    q = rng.normal(size=n) * 0.5
    qd.copy_(torch.from_numpy(q))
    torch.cuda.synchronize()
    g.replay()
    check(q, "replay after eager")


@pytest.mark.parametrize("n", [120_000, 2_600_000])
def test_one_group_over_the_vector_in_a_graph(s, orc, n):
    """Round 4: shifted(NormL2(lambda), x[, Delta, chi]) -- one group over the vector, the team form (csrc/spx_group_team.hip) and
    the chunked psi(y) -- captured and replayed on new data: n = 1.2e5 on chip, 2.6e6 streamed (three launches per Binf call:
    their exchange words and tile counters are zeroed by nodes of the graph)."""
    import torch
    rng = np.random.default_rng(n + 1)
    x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n)
    lam = 0.4 * n ** 0.5
    off = np.array([0, n], dtype=np.int64)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(sj).cuda()
        qd = torch.zeros(n, dtype=torch.float64, device="cuda")
        ys = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(2)]
        vals = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(2)]
        psi_g = s.shifted(s.shifted(s.NormL2(lam), xd), sd)
        psi_b = s.shifted(s.shifted(s.NormL2(lam), xd, 0.8, s.NormLinf(1.0)), sd)

        def iteration():
            s.prox_bang(ys[0], psi_g, qd, 0.9)
            with s.device_values(vals[0]):
                psi_g(ys[0])
            s.prox_bang(ys[1], psi_b, qd, 0.9)
            with s.device_values(vals[1]):
                psi_b(ys[1])

        qd.copy_(torch.from_numpy(rng.normal(size=n)))
        iteration(); iteration()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        iteration()
    for rep in range(3):
        q = rng.normal(size=n) * (1.0 + rep)
        qd.copy_(torch.from_numpy(q))
        for t in ys:
            t.fill_(-777.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        nS = np.linalg.norm((q + x) + sj)
        for k, delta in ((0, None), (1, 0.8)):
            ref = orc.prox_group_l2(q, x, sj, [lam], 0.9, offsets=off) if delta is None else orc.prox_group_l2_binf(q, x, sj, [lam], 0.9, delta, offsets=off)
            y = ys[k].cpu().numpy()
            scale = np.maximum(np.maximum(np.abs(ref), np.abs(x + sj)), nS)
            assert float(np.max(np.abs(y - ref) / scale)) <= 1e-12, (rep, k)
            vr = orc.obj_group_l2(y, x, sj, [lam], offsets=off, delta=delta)
            got = float(vals[k].item())
            assert got == vr or abs(got - vr) <= 1e-12 * abs(vr), (rep, k, got, vr)


def test_calls_that_synchronise_refuse_to_be_captured(s):
    import torch
    n = 10_000
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xd = torch.randn(n, dtype=torch.float64, device="cuda"); sd = torch.zeros_like(xd); qd = torch.randn_like(xd)
        y = torch.empty_like(qd)
        psi = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, s.NormLinf(1.0)), sd)
        s.prox_bang(y, psi, qd, 1.0); psi(y)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        s.prox_bang(y, psi, qd, 1.0)
        with pytest.raises(s._lib.SpxError, match="captured"):
            psi(y)                      # returns a host double: would have to synchronise
    g.replay()
    torch.cuda.synchronize()
