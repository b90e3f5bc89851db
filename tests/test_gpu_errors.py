"""Error behaviour of the C ABI (SURVEY 8b "Errors"): every misuse returns a status and a message -- nothing crashes,
nothing is written.  Called through ctypes with raw pointers, as a foreign host would."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    s = ge.build()
    L = s._lib.load()
    ctx = s.context("cuda:0")
    return s, L, ctx, torch


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def test_invalid_arguments_return_status_and_message(env):
    s, L, ctx, torch = env
    n = 1000
    v = [torch.randn(n, dtype=torch.float64, device="cuda:0") for _ in range(4)]
    y, q, x, sj = v
    y0 = y.clone()
    INVALID = 1
    null = ctypes.c_void_p(0)
    cases = [
        lambda: L.spx_prox_l1(None, _p(y), _p(q), _p(x), _p(sj), n, 1.0, 1.0),                  # ctx NULL
        lambda: L.spx_prox_l1(ctx, null, _p(q), _p(x), _p(sj), n, 1.0, 1.0),                     # y NULL
        lambda: L.spx_prox_l1(ctx, _p(y), _p(q), _p(x), _p(sj), -1, 1.0, 1.0),                   # n < 0
        lambda: L.spx_prox_l1_box(ctx, _p(y), null, _p(x), _p(sj), n, 1.0, 1.0, null, null, -1.0, 1.0, null),
        lambda: L.spx_prox_group_l2(ctx, _p(y), _p(q), _p(x), _p(sj), n, null, 7, 11, _p(x), 1.0),      # 7 * 11 != n
        lambda: L.spx_prox_group_l2(ctx, _p(y), _p(q), _p(x), _p(sj), n, null, 0, 10, _p(x), 1.0),      # group_size 0
        lambda: L.spx_prox_group_l2_binf(ctx, _p(y), _p(q), _p(x), _p(sj), n, null, 10, 100, null, 1.0, 1.0),  # lambda NULL
        lambda: L.spx_prox_group_l2_gather(ctx, _p(y), _p(q), _p(x), _p(sj), n, null, null, 3, 0, _p(x), 1.0),  # ptr NULL
        lambda: L.spx_ctx_set_tuning(s.context("cuda:0"), 99, 1),
    ]
    for k, call in enumerate(cases):
        rc = call()
        assert rc == INVALID, (k, rc)
        assert len(L.spx_last_error()) > 0
    torch.cuda.synchronize()
    assert torch.equal(y, y0)                                    # nothing was written
    # n == 0 is valid and a no-op (NULL vectors allowed)
    assert L.spx_prox_l1(ctx, null, null, null, null, 0, 1.0, 1.0) == 0
    assert L.spx_prox_indball_l0(ctx, null, null, null, null, 0, 3) == 0
    out = ctypes.c_double(-1.0)
    assert L.spx_obj_l1(ctx, null, null, null, 0, 1.0, ctypes.byref(out)) == 0 and out.value == 0.0


def test_gather_index_out_of_range_is_refused_before_any_store(env):
    s, L, ctx, torch = env
    n = 64
    y, q, x, sj = (torch.randn(n, dtype=torch.float64, device="cuda:0") for _ in range(4))
    y0 = y.clone()
    ptr = torch.tensor([0, 3, 5], dtype=torch.int64, device="cuda:0")
    lam = torch.ones(2, dtype=torch.float64, device="cuda:0")
    for bad in ([0, 1, 64, 2, 3], [0, -1, 5, 2, 3]):            # the reference: BoundsError
        idx = torch.tensor(bad, dtype=torch.int64, device="cuda:0")
        rc = L.spx_prox_group_l2_gather(ctx, _p(y), _p(q), _p(x), _p(sj), n, _p(ptr), _p(idx), 2, 5, _p(lam), 1.0)
        assert rc == 1 and b"BoundsError" in L.spx_last_error()
        torch.cuda.synchronize()
        assert torch.equal(y, y0)
    badptr = torch.tensor([0, 4, 3], dtype=torch.int64, device="cuda:0")       # decreasing
    idx = torch.tensor([0, 1, 2, 3, 4], dtype=torch.int64, device="cuda:0")
    assert L.spx_prox_group_l2_gather(ctx, _p(y), _p(q), _p(x), _p(sj), n, _p(badptr), _p(idx), 2, 5, _p(lam), 1.0) == 1
    assert torch.equal(y, y0)


def test_iprox_assertion_status(env):
    s, L, ctx, torch = env
    n = 100
    y, g, x, sj = (torch.randn(n, dtype=torch.float64, device="cuda:0") for _ in range(4))
    d = torch.ones(n, dtype=torch.float64, device="cuda:0")
    assert L.spx_iprox_l1(ctx, _p(y), _p(g), _p(d), _p(x), _p(sj), n, 1.0, 1) == 0
    d[17] = 0.0                                                   # the reference: `@assert d[i] > 0`
    assert L.spx_iprox_l1(ctx, _p(y), _p(g), _p(d), _p(x), _p(sj), n, 1.0, 1) == 6
    assert L.spx_iprox_l0(ctx, _p(y), _p(g), _p(d), _p(x), _p(sj), n, 1.0, 1) == 6
    assert L.spx_iprox_l1(ctx, _p(y), _p(g), _p(d), _p(x), _p(sj), n, 1.0, 0) == 0    # unchecked: asynchronous
    psi = s.shifted(s.NormL1(1.0), x)
    with pytest.raises(AssertionError):
        s.iprox(psi, g, d)


def test_mirror_errors(env):
    s, L, ctx, torch = env
    x = torch.randn(10, dtype=torch.float64, device="cuda:0")
    psi = s.shifted(s.NormL1(1.0), x)
    with pytest.raises(IndexError):                               # BoundsError: length mismatch
        s.prox(psi, torch.randn(9, dtype=torch.float64, device="cuda:0"), 1.0)
    with pytest.raises(TypeError):
        s.prox(psi, torch.randn(10, dtype=torch.float32, device="cuda:0"), 1.0)
    with pytest.raises(TypeError):
        s.prox(psi, torch.randn(20, dtype=torch.float64, device="cuda:0")[::2], 1.0)
    with pytest.raises(TypeError):
        s.iprox(s.shifted(s.RootNormLhalf(1.0), x), x, x)         # MethodError: no iprox! for RootNormLhalf
    with pytest.raises(AttributeError):
        s.set_radius_bang(psi, 1.0)                               # no field Δ
    with pytest.raises(TypeError):
        s.shifted(s.NormL0(1.0), x, 1.0, s.NormL2(1.0))           # no ShiftedNormL0B2 in the reference either
