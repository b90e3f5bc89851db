"""Round 4 (third session): randomised hunt on the one-launch forms -- psi(y), prox! + value, Binf groups without the zero-fill
launch -- against the forms of rounds 1-3 (tuning key 17 = 0), bit for bit.  Random sizes (1 ... 3e6, biased to the boundaries of
the grids: 2048 workgroups of the reductions, one tile per workgroup of the separable skeleton), random operator, feasible and
infeasible points, calls queued back to back without a synchronisation, other operators in between.  usage: [seeds]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
import __graft_entry__ as ge
os.environ.setdefault("SPX_NO_BUILD", "1")
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
chi = s.NormLinf(1.0)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
special = [1, 2, 3, 255, 256, 257, 2047 * 8, 2048 * 8, 2048 * 8 + 1, 2048 * 2048, 2048 * 2048 + 2, 3072 * 2048, 3072 * 2048 + 2, 3072 * 2049]
for seed in range(nseeds):
    rng = np.random.default_rng(9000 + seed)
    n = int(special[seed % len(special)] + rng.integers(0, 3)) if seed % 3 == 0 else int(10 ** rng.uniform(0, 6.4))
    gs = int(rng.choice([1, 2, 5, 8, 50, 128]))
    n = max(gs, n // gs * gs)
    x = torch.from_numpy(rng.normal(size=n)).cuda(); sj = torch.from_numpy(rng.uniform(-0.5, 0.5, size=n)).cuda()
    q = torch.from_numpy(rng.normal(size=n)).cuda()
    y_ok = torch.from_numpy(rng.uniform(-0.3, 0.3, size=n)).cuda(); y_bad = y_ok * 10.0
    lam = rng.uniform(0.5, 1.5, size=n // gs)
    H = s.GroupNormL2.uniform(lam.tolist(), gs)
    ops = [s.shifted(s.shifted(s.NormL1(0.7), x), sj), s.shifted(s.shifted(s.NormL0(0.7), x, 0.9, chi), sj),
           s.shifted(s.shifted(s.RootNormLhalf(0.7), x, 0.9, chi), sj), s.shifted(s.shifted(s.IndBallL0(max(1, n // 3)), x, 0.9, chi), sj),
           s.shifted(s.shifted(H, x), sj), s.shifted(s.shifted(H, x, 0.9, chi), sj)]
    psi = ops[int(rng.integers(0, len(ops)))]
    seq = [y_ok if rng.random() < 0.6 else y_bad for _ in range(6)]
    L.spx_ctx_set_tuning(ctx, 17, 0)
    want = [psi(yy) for yy in seq]
    L.spx_ctx_set_tuning(ctx, 17, 1)
    out = torch.full((len(seq),), -1.0, dtype=torch.float64, device="cuda:0")
    tmp = torch.empty_like(q)
    for k, yy in enumerate(seq):   # queued without a synchronisation, another operator in between now and then
        L.spx_ctx_set_value_target(ctx, ctypes.c_void_p(out[k:].data_ptr()))
        psi(yy)
        if rng.random() < 0.3:
            L.spx_ctx_set_value_target(ctx, None)
            s.prox_bang(tmp, ops[int(rng.integers(0, len(ops)))], q, 1.0)
    L.spx_ctx_set_value_target(ctx, None)
    got = out.cpu().numpy()
    same = all((a == b) or (np.isinf(a) and np.isinf(b)) or (a != a and b != b) for a, b in zip(got, want))
    # prox! + value on a separable operator
    pv = ops[int(rng.integers(0, 3))]
    L.spx_ctx_set_tuning(ctx, 17, 0); y0, v0 = s.prox_value(pv, q, 1.1); y0 = y0.clone()
    L.spx_ctx_set_tuning(ctx, 17, 1); y1, v1 = s.prox_value(pv, q, 1.1)
    same_pv = torch.equal(y0.view(torch.int64), y1.view(torch.int64)) and (v0 == v1)
    # Binf prox! twice in a row against the three-launch form
    bi = ops[5]
    L.spx_ctx_set_tuning(ctx, 17, 0); s.prox_bang(tmp, bi, q, 1.3); b0 = tmp.clone()
    L.spx_ctx_set_tuning(ctx, 17, 1)
    same_b = True
    for _ in range(2):
        tmp.fill_(float("nan")); s.prox_bang(tmp, bi, q, 1.3)
        same_b = same_b and torch.equal(tmp.view(torch.int64), b0.view(torch.int64))
    if not (same and same_pv and same_b):
        bad += 1
        print("seed %d n %d gs %d %s: psi %s prox+value %s binf %s | got %s want %s" % (seed, n, gs, type(psi).__name__, same, same_pv, same_b, got, want), flush=True)
    if seed % 50 == 49: print("seed %d done, %d bad" % (seed, bad), flush=True)
rc = L.spx_sync(ctx)
print("seeds %d bad %d sync rc %d" % (nseeds, bad, rc))
sys.exit(1 if bad or rc else 0)
