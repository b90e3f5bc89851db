"""Group prox on index sets (spx_prox_group_l2[_binf]_gather) and on ragged contiguous groups WITHOUT a size bound
(group_size = 0), small groups: time per call through the C ABI."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(3)
n = int(os.environ.get("SPX_N", "4000000"))
x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); y = torch.empty_like(q)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timed(fn):
    for _ in range(3): assert fn() == 0, L.spx_last_error()
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(3):
        ms = ctypes.c_float()
        L.spx_timer_start(ctx)
        for _ in range(10): fn()
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        best = min(best, ms.value / 10 * 1e3)
    return best
for gs in (2, 4, 8, 16, 48, 128):
    ng = n // gs; m = ng * gs
    lam = torch.rand(ng, dtype=torch.float64, device="cuda:0", generator=g) + 0.5
    ptr = torch.arange(0, m + 1, gs, dtype=torch.int64, device="cuda:0")
    index = torch.randperm(m, device="cuda:0", generator=g).to(torch.int64)   # every index once, scattered
    t1 = timed(lambda: L.spx_prox_group_l2_gather(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(m), P(ptr), P(index), ctypes.c_int64(ng), ctypes.c_int64(m), P(lam), ctypes.c_double(1.0)))
    t2 = timed(lambda: L.spx_prox_group_l2_binf_gather(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(m), P(ptr), P(index), ctypes.c_int64(ng), ctypes.c_int64(m), P(lam), ctypes.c_double(1.0), ctypes.c_double(1.0)))
    t3 = timed(lambda: L.spx_prox_group_l2(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(m), P(ptr), ctypes.c_int64(0), ctypes.c_int64(ng), P(lam), ctypes.c_double(1.0)))
    t4 = timed(lambda: L.spx_prox_group_l2_binf(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(m), P(ptr), ctypes.c_int64(0), ctypes.c_int64(ng), P(lam), ctypes.c_double(1.0), ctypes.c_double(1.0)))
    print("groups of %-4d index sets: plain %8.1f us  Binf %8.1f us | contiguous, no size bound: plain %8.1f us  Binf %8.1f us" % (gs, t1, t2, t3, t4), flush=True)
