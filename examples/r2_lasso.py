"""R2 (quadratic regularisation, RegularizedOptimization.jl [ext]) for  min_x 1/2 ||A x - b||^2 + lambda ||x||_1  with the
prox! hot path on the GPU -- the caller this library is a drop-in for.  The loop below is the reference solver's inner
loop in miniature, using only the mirrored API:

    psi = shifted(h, xk)                       # borrows xk: updating xk in place re-centres psi (shift!)
    s, hkn = prox_value(psi, grad, nu, q_scale=-nu)   # mnu_grad = -nu grad; prox!(s, psi, mnu_grad, nu); psi(s): one pass
    shift_bang(psi, xk)                        # after an accepted step

Everything stays in device memory; torch supplies the smooth part (A x, A' r).  `backend="oracle"` runs the same loop on
the CPU with the reference restatement (test infrastructure) -- tests/test_gpu_r2_loop.py compares the two trajectories.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repository root: spx_amd, oracle


def r2_lasso(A, b, lam, x0, backend, max_iter=200, tol=1e-6, eta1=1e-4, eta2=0.9, gamma=3.0, nu0=None):
    if backend == "gpu":
        import torch
        import spx_amd as spx
        dot = lambda u, v: float(torch.dot(u, v))
        norm2 = lambda v: float(torch.dot(v, v))
        xk = x0.clone()
        psi = spx.shifted(spx.NormL1(lam), xk)
        prox_val = lambda grad, nu: spx.prox_value(psi, grad, nu, q_scale=-nu)   # q = -nu grad formed on the fly
        hval = lambda: psi(torch.zeros_like(xk))
    else:
        from oracle import oracle
        dot = lambda u, v: float(np.dot(u, v))
        norm2 = lambda v: float(np.dot(v, v))
        xk = x0.copy()
        zero = np.zeros_like(xk)

        def prox_val(grad, nu):
            s = oracle.prox_l1(-nu * grad, xk, zero, lam, nu)
            return s, oracle.obj_plain("l1", s, xk, zero, lam)
        hval = lambda: oracle.obj_plain("l1", zero, xk, zero, lam)

    res = A @ xk - b
    fk, hk = 0.5 * norm2(res), hval()
    grad = A.T @ res
    nu = nu0 if nu0 is not None else 1.0
    hist = []
    for it in range(max_iter):
        s, hkn = prox_val(grad, nu)
        xi = hk - (dot(grad, s) + hkn)                      # model decrease
        if xi < 0 or np.sqrt(max(xi, 0.0) / nu) < tol:
            hist.append((it, fk + hk, nu, None))
            break
        xkn = xk + s
        resn = A @ xkn - b
        fkn = 0.5 * norm2(resn)
        rho = (fk + hk - fkn - hkn) / xi
        accepted = rho >= eta1
        hist.append((it, fk + hk, nu, accepted))
        if accepted:
            if backend == "gpu":
                xk.copy_(xkn)                                # in place: psi.xk IS xk  (shift!(psi, xk))
            else:
                xk[:] = xkn
            res, fk, hk = resn, fkn, hkn
            grad = A.T @ res
        if rho >= eta2:
            nu *= gamma
        elif rho < eta1:
            nu /= gamma
    return xk, hist


if __name__ == "__main__":
    import torch
    torch.manual_seed(0)
    m, n = 2000, 20000
    A = torch.randn(m, n, dtype=torch.float64, device="cuda") / np.sqrt(m)
    xtrue = torch.zeros(n, dtype=torch.float64, device="cuda")
    xtrue[torch.randperm(n)[:100]] = torch.randn(100, dtype=torch.float64, device="cuda") * 3
    b = A @ xtrue + 0.01 * torch.randn(m, dtype=torch.float64, device="cuda")
    x, hist = r2_lasso(A, b, 0.1, torch.zeros(n, dtype=torch.float64, device="cuda"), "gpu", nu0=0.1)
    print("iterations", len(hist), "objective", hist[-1][1], "nnz", int((x != 0).sum()))
