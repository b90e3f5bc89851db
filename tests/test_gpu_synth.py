"""spx_synth_fill (include/spx.h): the device side of SURVEY 8d's shared generator gives the bits of its host twins."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def spx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.build()


@pytest.mark.parametrize("kind", [0, 1])
@pytest.mark.parametrize("n", [0, 1, 255, 65_537, 3_000_001])
def test_device_generator_matches_host_twin(spx, kind, n):
    import torch
    from oracle import oracle, synth
    s = spx
    L = s._lib.load()
    ctx = s.context(torch.device("cuda", 0))
    out = torch.full((n + 8,), 123.0, dtype=torch.float64, device="cuda")
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(out.data_ptr()), n, 20250613, 5, kind, 1.25))
    torch.cuda.synchronize()
    h = out.cpu().numpy()
    assert np.all(h[n:] == 123.0)
    ref = oracle.synth_fill(n, 20250613, 5, kind, 1.25, threads=2)
    assert np.array_equal(h[:n].view(np.int64), ref.view(np.int64))
    if n <= 65_537:
        assert np.array_equal(ref.view(np.int64), synth.fill(n, 20250613, 5, kind, 1.25).view(np.int64))


def test_generator_rejects_bad_arguments(spx):
    import torch
    s = spx
    L = s._lib.load()
    ctx = s.context(torch.device("cuda", 0))
    out = torch.empty(4, dtype=torch.float64, device="cuda")
    assert L.spx_synth_fill(ctx, ctypes.c_void_p(out.data_ptr()), 4, 1, 0, 2, 1.0) != 0
    assert L.spx_synth_fill(ctx, None, 4, 1, 0, 0, 1.0) != 0
    assert L.spx_synth_fill(ctx, ctypes.c_void_p(out.data_ptr()), -1, 1, 0, 0, 1.0) != 0
