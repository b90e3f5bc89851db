"""Trust-region proximal gradient for a group lasso at the BASELINE group shape (1e6 groups x 128):

    min_x  1/2 ||x - b||^2 + sum_g lambda_g ||x_g||_2 ,      step:  s = prox!(psi, -nu grad, nu),  psi = shifted(h, xk, Delta, chi)

i.e. ShiftedGroupNormL2Binf (src/shiftedGroupNormL2Binf.jl) called the way a TR solver of RegularizedOptimization.jl [ext]
calls it.  b is zero on 90 % of the groups, so after the first steps most groups of xk are exactly zero and sit under a
sigma*lambda above ||S|| -- the reversed-bracket regime of the reference's root find.  The point of the
example: the per-iteration prox! time stays at the bandwidth figure while the iterate becomes sparse.

    gpurun -- 'python examples/tr_group_lasso.py'
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
g = torch.Generator(device=dev).manual_seed(11)
ng, gs = (int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000), 128
n = ng * gs
active = (torch.rand(ng, dtype=torch.float64, device=dev, generator=g) < 0.1).to(torch.float64).repeat_interleave(gs)
b = (3.0 * torch.randn(n, dtype=torch.float64, device=dev, generator=g) + 0.3) * active
b += 0.05 * torch.randn(n, dtype=torch.float64, device=dev, generator=g)            # noise everywhere
lam = torch.full((ng,), 4.0, dtype=torch.float64, device=dev)                     # kills the noise-only groups
h = s.GroupNormL2.uniform(lam, gs)
xk = 0.1 * torch.randn(n, dtype=torch.float64, device=dev, generator=g)            # dense start
sj = torch.zeros_like(xk)
delta, nu = 1.0, 0.5
psi = s.shifted(s.shifted(h, xk, delta, s.NormLinf(1.0)), sj)
step = torch.empty_like(xk); q = torch.empty_like(xk)
obj = lambda: 0.5 * float(torch.dot(xk - b, xk - b)) + float((lam * xk.view(ng, gs).norm(dim=1)).sum())
torch.mul(xk - b, -nu, out=q)
s.prox_bang(step, psi, q, nu)                        # untimed first call: the context allocates its scratch here
print("it   objective      zero groups   prox! ms")
for it in range(25):
    torch.mul(xk - b, -nu, out=q)                                                   # q = -nu grad f(xk)
    ms = ctypes.c_float(); L.spx_timer_start(ctx)
    s.prox_bang(step, psi, q, nu)
    L.spx_timer_stop(ctx, ctypes.byref(ms))
    xk.add_(step)                                                                   # psi borrows xk: re-centred in place
    zero_groups = int((xk.view(ng, gs).abs().amax(dim=1) == 0).sum())
    print("%2d   %.6e   %8d      %.3f" % (it, obj(), zero_groups, ms.value), flush=True)
    if float(step.abs().max()) < 1e-9:
        break
    delta = min(4.0 * delta, 64.0)                                                  # every step is a descent step here
    s.set_radius_bang(psi, delta)
