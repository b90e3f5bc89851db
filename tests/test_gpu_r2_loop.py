"""The library inside the loop it is a drop-in for: R2 (examples/r2_lasso.py) on the GPU against the same loop on the CPU
with the oracle's prox -- same accept / reject decisions, same step sizes, iterates equal to rounding."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_r2_lasso_trajectory_matches_cpu_loop():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    ge.build()
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from r2_lasso import r2_lasso
    rng = np.random.default_rng(0)
    m, n = 300, 2000
    A = rng.normal(size=(m, n)) / np.sqrt(m)
    xtrue = np.zeros(n)
    xtrue[rng.choice(n, size=25, replace=False)] = rng.normal(size=25) * 3
    b = A @ xtrue + 0.01 * rng.normal(size=m)
    lam = 0.05
    x_cpu, h_cpu = r2_lasso(A, b, lam, np.zeros(n), "oracle", max_iter=120, nu0=0.2)
    Ad, bd = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()
    x_gpu, h_gpu = r2_lasso(Ad, bd, lam, torch.zeros(n, dtype=torch.float64, device="cuda"), "gpu", max_iter=120, nu0=0.2)
    assert len(h_cpu) == len(h_gpu) and len(h_gpu) > 10
    for (i0, o0, nu0, a0), (i1, o1, nu1, a1) in zip(h_cpu, h_gpu):
        assert a0 == a1 and nu0 == nu1 and abs(o0 - o1) <= 1e-10 * abs(o0)
    xg = x_gpu.cpu().numpy()
    assert np.max(np.abs(xg - x_cpu)) <= 1e-9 * max(1.0, np.max(np.abs(x_cpu)))
    assert np.array_equal(xg != 0, x_cpu != 0)                    # same support
    assert h_gpu[-1][1] < 0.5 * h_gpu[0][1]                       # and it did minimise
