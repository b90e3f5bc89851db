"""Round 3, late additions -- cross-checks on random instances (usage: fuzz_late.py [instances] [seed]):
 (a) group prox on small groups: the register tiles (uniform size), the team kernel for ragged groups without a size bound
     (group_size = 0) and the gather kernel (index sets) are three independent kernels: on the same contiguous uniform groups
     their results must agree to 1e-12 of the scale (plain: 1e-14), psi(y) must match a torch Float64 evaluation to 1e-12;
 (b) Float32 top-r: one-launch forms (registers / LDS, folded first digit) against the numpy restatement (stable sort), bits;
 (c) Float64 top-r at 2^20 < n <= 2^22 (LDS form) against the form that parks v in y (keys 2 = 0, 11 = 0), bits."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
from oracle import oracle as orc
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
g = torch.Generator(device="cuda:0").manual_seed(seed)
P = lambda t: ctypes.c_void_p(t.data_ptr())
bad = 0
def fail(msg):
    global bad
    bad += 1
    print("MISMATCH", msg, flush=True)
for it in range(N):
    # ---------------- (a) groups
    gs = int(rng.integers(1, 21)); ng = int(rng.integers(1, 300_000)); n = gs * ng
    kind = int(rng.integers(0, 5))
    x = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g); sj = torch.rand(n, dtype=torch.float64, device="cuda:0", generator=g) - 0.5
    q = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g)
    if kind == 1: x = torch.round(x * 4) / 4; q = torch.round(q * 4) / 4; sj = torch.round(sj * 4) / 4
    elif kind == 2: x = x * (torch.rand(n, device="cuda:0", generator=g) < 0.2)
    elif kind == 3: q = q * 1e-3
    elif kind == 4: x = x * 0.0; sj = sj * 0.0
    lam = (torch.rand(ng, dtype=torch.float64, device="cuda:0", generator=g) * 2.0 + 0.05) * float(rng.choice([0.1, 1.0, 10.0]))
    sigma = float(rng.choice([0.3, 1.0, 2.0])); delta = float(rng.choice([0.2, 1.0, 5.0]))
    ptr = torch.arange(0, n + 1, gs, dtype=torch.int64, device="cuda:0")
    index = torch.arange(0, n, dtype=torch.int64, device="cuda:0")
    scale = float(max(torch.linalg.norm(x + sj + q).item() / max(ng, 1) ** 0.5, 1e-300)) + float(x.abs().max()) + 1.0
    for binf in (False, True):
        ys = []
        for form in range(3):
            y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda:0")
            if form == 0:
                rc = (L.spx_prox_group_l2_binf(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(n), None, ctypes.c_int64(gs), ctypes.c_int64(ng), P(lam), ctypes.c_double(sigma), ctypes.c_double(delta)) if binf else
                      L.spx_prox_group_l2(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(n), None, ctypes.c_int64(gs), ctypes.c_int64(ng), P(lam), ctypes.c_double(sigma)))
            elif form == 1:
                rc = (L.spx_prox_group_l2_binf(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(n), P(ptr), ctypes.c_int64(0), ctypes.c_int64(ng), P(lam), ctypes.c_double(sigma), ctypes.c_double(delta)) if binf else
                      L.spx_prox_group_l2(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(n), P(ptr), ctypes.c_int64(0), ctypes.c_int64(ng), P(lam), ctypes.c_double(sigma)))
            else:
                rc = (L.spx_prox_group_l2_binf_gather(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(n), P(ptr), P(index), ctypes.c_int64(ng), ctypes.c_int64(n), P(lam), ctypes.c_double(sigma), ctypes.c_double(delta)) if binf else
                      L.spx_prox_group_l2_gather(ctx, P(y), P(q), P(x), P(sj), ctypes.c_int64(n), P(ptr), P(index), ctypes.c_int64(ng), ctypes.c_int64(n), P(lam), ctypes.c_double(sigma)))
            if rc: fail("rc %d %s" % (rc, L.spx_last_error()))
            ys.append(y)
        tol = (1e-9 if binf else 1e-13) * scale   # (Binf next to the pole: the forms may differ by the reference's own 1e-9 there, tests/arbiter.py)
        for k in (1, 2):
            d = float((ys[k] - ys[0]).abs().max())
            if not (d <= tol) or bool(torch.isnan(ys[k]).any()):
                fail("groups it=%d gs=%d ng=%d kind=%d binf=%s form=%d maxdiff=%.3e scale=%.3e" % (it, gs, ng, kind, binf, k, d, scale))
    # psi(y)
    yv = torch.randn(n, dtype=torch.float64, device="cuda:0", generator=g) * 0.1
    val = ctypes.c_double()
    rc = L.spx_obj_group_l2(ctx, P(yv), P(x), P(sj), ctypes.c_int64(n), None, ctypes.c_int64(gs), ctypes.c_int64(ng), P(lam), ctypes.byref(val))
    want = float((lam * torch.linalg.norm((x + sj + yv).view(ng, gs), dim=1)).sum())
    if rc or abs(val.value - want) > 1e-11 * abs(want) + 1e-300: fail("obj it=%d gs=%d ng=%d got %.17g want %.17g rc %d" % (it, gs, ng, val.value, want, rc))
    del x, sj, q, lam, ptr, index, ys, yv
    # ---------------- (b) Float32 top-r against numpy
    if it % 2 == 0:
        n = int(rng.choice([int(rng.integers(8193, 300_000)), int(rng.integers(1_000_000, 2_400_000)), int(rng.integers(2_097_153, 3_200_000))]))
        xf = rng.normal(size=n).astype(np.float32); sf = rng.uniform(-0.5, 0.5, size=n).astype(np.float32); qf = rng.normal(size=n).astype(np.float32)
        k2 = int(rng.integers(0, 4))
        if k2 == 1: qf = (np.round(qf * 8) / 8).astype(np.float32); xf[:] = 0; sf[:] = 0
        elif k2 == 2: e = rng.integers(-30, 30, size=n); qf = (qf * 2.0 ** e).astype(np.float32)
        elif k2 == 3: qf = (qf * np.float32(2.0 ** -50)); xf = xf * np.float32(2.0 ** -50); sf = sf * np.float32(2.0 ** -50)
        r = int(rng.choice([1, n // 100, n // 2, n - 3, int(rng.integers(1, n))]))
        off = int(rng.integers(0, 4)) if it % 4 == 0 else 0
        mk = lambda a: torch.cat([torch.zeros(off, dtype=torch.float32), torch.from_numpy(a)]).cuda()[off:]
        xd, sd, qd = mk(xf), mk(sf), mk(qf)
        with np.errstate(all="ignore"):
            ref = orc.prox_indball_l0_f32(qf, xf, sf, r, 0.75)
        y = s.prox(s.shifted(s.shifted(s.IndBallL0(r), xd, 0.75, s.NormLinf(1.0)), sd), qd, 1.0).cpu().numpy()
        same = (y.view(np.int32) == ref.view(np.int32)) | (np.isnan(y) & np.isnan(ref))
        if not same.all(): fail("f32 top-r it=%d n=%d kind=%d r=%d off=%d ndiff=%d" % (it, n, k2, r, off, int((~same).sum())))
    # ---------------- (c) Float64 LDS form against the parked form
    else:
        n = int(rng.integers((1 << 20) + 1, (1 << 22) + 1))
        base = torch.randn(n + 1, dtype=torch.float64, device="cuda:0", generator=g)
        k3 = int(rng.integers(0, 4))
        sc = float(2.0 ** rng.integers(-4, 5))
        qq = base if k3 == 0 else torch.round(base * sc) / sc if k3 == 1 else base * torch.exp2(torch.randint(-40, 40, (n + 1,), device="cuda:0", generator=g).double()) if k3 == 2 else torch.full_like(base, 1.5)
        xx = torch.randn(n + 1, dtype=torch.float64, device="cuda:0", generator=g) * float(rng.choice([0.0, 1.0]))
        zz = torch.zeros(n + 1, dtype=torch.float64, device="cuda:0")
        head = int(rng.integers(0, 2))
        qv, xv, sv = qq[head:head + n], xx[head:head + n], zz[head:head + n]
        r = int(rng.choice([1, 3, n // 100, n // 2, n - 5, int(rng.integers(1, n))]))
        psi = s.shifted(s.shifted(s.IndBallL0(r), xv, 0.9, s.NormLinf(1.0)), sv)
        yref = torch.empty(n + 1, dtype=torch.float64, device="cuda:0")[head:head + n]
        y = torch.full((n + 1,), float("nan"), dtype=torch.float64, device="cuda:0")[head:head + n]
        L.spx_ctx_set_tuning(ctx, 2, 0); L.spx_ctx_set_tuning(ctx, 11, 0); s.prox_bang(yref, psi, qv, 1.0)
        L.spx_ctx_set_tuning(ctx, 2, 1); L.spx_ctx_set_tuning(ctx, 11, 1)
        s.prox_bang(y, psi, qv, 1.0)
        a, b = y.view(torch.int64), yref.view(torch.int64)
        if not bool(((a == b) | (torch.isnan(y) & torch.isnan(yref))).all()): fail("f64 LDS top-r it=%d n=%d kind=%d head=%d r=%d ndiff=%d" % (it, n, k3, head, r, int((a != b).sum())))
    rc = L.spx_sync(ctx)
    if rc: fail("sync rc %d it=%d" % (rc, it))
    if it % 20 == 19: print("... %d instances, %d mismatches" % (it + 1, bad), flush=True)
print("done: %d instances, %d mismatches" % (N, bad))
sys.exit(1 if bad else 0)
