"""One libspx context per (device, HIP stream): calls issued under two torch streams run on those streams, keep their
own scratch, and give the results of the serial order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_streams_interleaved():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    s = ge.build()
    dev = torch.device("cuda:0")
    n = 6_400_000                                # (top-r: the sample-predicted pipeline, which uses the context scratch)
    g = torch.Generator(device=dev).manual_seed(5)
    mk = lambda: torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    xa, sa, qa, xb, sb, qb = mk(), mk() * 0.3, mk(), mk(), mk() * 0.3, mk()
    chi = s.NormLinf(1.0)
    psi_a = s.shifted(s.shifted(s.IndBallL0(n // 50), xa, 0.8, chi), sa)          # top-r: uses the context scratch
    lam = torch.rand(n // 100, dtype=torch.float64, device=dev, generator=g) + 0.5
    psi_b = s.shifted(s.shifted(s.GroupNormL2.uniform(lam, 100), xb, 0.9, chi), sb)   # Binf groups: deferred list in scratch
    ref_a = s.prox(psi_a, qa, 1.0).clone()
    ref_b = s.prox(psi_b, qb, 1.0).clone()
    va, vb = psi_a(ref_a), psi_b(ref_b)
    torch.cuda.synchronize()
    st1, st2 = torch.cuda.Stream(), torch.cuda.Stream()
    ya, yb = torch.empty_like(qa), torch.empty_like(qb)
    ctxs = set()
    for rep in range(6):
        with torch.cuda.stream(st1):
            s.prox_bang(ya, psi_a, qa, 1.0)
            ctxs.add(s.context(dev).value)
        with torch.cuda.stream(st2):
            s.prox_bang(yb, psi_b, qb, 1.0)
            ctxs.add(s.context(dev).value)
    with torch.cuda.stream(st1):
        wa = psi_a(ya)
    with torch.cuda.stream(st2):
        wb = psi_b(yb)
    torch.cuda.synchronize()
    assert len(ctxs) == 2                                          # one context per stream
    assert torch.equal(ya, ref_a) and torch.equal(yb, ref_b)
    assert wa == va and wb == vb


def test_two_streams_both_in_launch_synchronised_kernels():
    """Two contexts (two torch streams) both run top-r, whose kernels synchronise inside one launch (grid barriers between
    resident workgroups).  Two such launches side by side could starve each other of CUs; libspx chains them through a
    per-device event whenever more than one context exists (CoopLaunchGuard, csrc/spx_select.hip).  Results must be those of
    the serial order; a deadlock would show as the pytest timeout."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    s = ge.build()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    cases = []
    for n in (300_000, (1 << 21) + 77, (1 << 22) + 77, 6 * (1 << 20) + 77):   # one launch: v in registers / in LDS / in both; the sample-predicted pipeline
        mk = lambda: torch.randn(n, dtype=torch.float64, device=dev, generator=g)
        x, sj, q = mk(), mk() * 0.3, mk()
        psi = s.shifted(s.shifted(s.IndBallL0(n // 37), x, 0.8, s.NormLinf(1.0)), sj)
        cases.append((psi, q, s.prox(psi, q, 1.0).clone()))
    torch.cuda.synchronize()
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.empty_like(c[1]) for c in cases] for _ in st]
    for rep in range(8):
        for k, stream in enumerate(st):
            with torch.cuda.stream(stream):
                for j, (psi, q, _) in enumerate(cases if (rep + k) % 2 == 0 else cases[::-1]):
                    jj = j if (rep + k) % 2 == 0 else len(cases) - 1 - j
                    s.prox_bang(outs[k][jj], psi, q, 1.0)
    torch.cuda.synchronize()
    for k in range(2):
        for j, (_, _, ref) in enumerate(cases):
            assert torch.equal(outs[k][j], ref), (k, j)
