"""Round 3: ShiftedIndBallL0BInf on tie-heavy data -- bit-for-bit against the exact select (tuning key 2 = 0, the
full-vector radix select: an independent route inside the library) and ms per call (HIP events, 5 calls).
env: SPX_N (default 1e8), SPX_KINDS, SPX_RS, SPX_ALIAS=1 (y === q form as well)"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = int(float(os.environ.get("SPX_N", "1e8")))
g = torch.Generator(device="cuda:0").manual_seed(int(os.environ.get("SPX_SEED", "1")))
q0 = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
z = torch.zeros(n, dtype=torch.float64, device="cuda:0"); z2 = torch.zeros_like(z); y = torch.empty_like(q0); yref = torch.empty_like(q0)
kinds = os.environ.get("SPX_KINDS", "continuous,lattice 1/4,lattice 1,lattice 2^-8,two values,constant,90% zeros,sorted lattice").split(",")
for kind in kinds:
    if kind == "continuous": q = q0
    elif kind == "lattice 1/4": q = torch.round(q0 * 4) / 4
    elif kind == "lattice 1": q = torch.round(q0)
    elif kind == "lattice 2^-8": q = torch.round(q0 * 256) / 256
    elif kind == "two values": q = torch.where(q0 > 0.5, torch.full_like(q0, 1.5), torch.full_like(q0, -0.75))
    elif kind == "90% zeros": q = torch.where(torch.rand(n, device="cuda:0", generator=g) < 0.9, torch.zeros_like(q0), q0)
    elif kind == "sorted lattice": q = torch.sort(torch.round(q0 * 4) / 4)[0]
    elif kind == "sorted": q = torch.sort(q0)[0]
    else: q = torch.full_like(q0, 2.0)
    rs = [int(float(v)) for v in os.environ.get("SPX_RS", "%d,%d,%d" % (n // 100, n // 2, n - n // 20)).split(",")]
    for r in rs:
        psi = s.shifted(s.shifted(s.IndBallL0(r), z, 1.0, s.NormLinf(1.0)), z2)   # (xk and sj in buffers of their own: honest traffic)
        L.spx_ctx_set_tuning(ctx, 2, 0)
        s.prox_bang(yref, psi, q, 1.0)
        L.spx_ctx_set_tuning(ctx, 2, 1)
        y.fill_(float("nan"))
        s.prox_bang(y, psi, q, 1.0)
        same = bool(torch.equal(y.view(torch.int64), yref.view(torch.int64)))
        nbad = 0 if same else int((y.view(torch.int64) != yref.view(torch.int64)).sum())
        s.prox_bang(y, psi, q, 1.0)
        ms = ctypes.c_float(); L.spx_timer_start(ctx)
        for _ in range(5): s.prox_bang(y, psi, q, 1.0)
        L.spx_timer_stop(ctx, ctypes.byref(ms))
        extra = ""
        if os.environ.get("SPX_ALIAS") == "1":
            qa = q.clone()
            s.prox_bang(qa, psi, qa, 1.0)
            extra = " | y===q %s" % ("ok" if torch.equal(qa.view(torch.int64), yref.view(torch.int64)) else "DIFFERS")
        rc = L.spx_sync(ctx)
        print("%-14s n=%-10d r=%-10d %8.3f ms per call | %s%s | sync rc %d" % (
            kind, n, r, ms.value / 5, "bit-identical to the exact select" if same else "DIFFERS in %d elements" % nbad, extra, rc), flush=True)
