"""Round 4: the top-r front kernel's sample on 64 or 256 workgroups (tuning key 10: 4 = 64 x 4, 160 = 64 x 16 [round 3's r = n/2
form], 16 = 256 x 4 [the same 1 Mi samples on every CU], 32 = 256 x 8).  bench.py's inputs (spx_synth_fill, seed + 1000),
ms per call by HIP events (median of 5 x 10 calls, interleaved over the configurations), y compared bit for bit with the
exact select (tuning key 2 = 0).  The values 32 and 160 and the 256-workgroup meaning of 16 only exist in the experiment's build
(profiles/r04_topr_front_wide_tried.txt: measured, not faster, not kept); on the library as committed the tool compares 1 / 2 / 4 / 16.  env: SPX_N (1e8), SPX_RDIVS ("100,2"), SPX_CFGS ("0,4,160,16,32"), SPX_TIES=1 (lattice 1/4)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import __graft_entry__ as ge
os.environ.setdefault("SPX_NO_BUILD", "1")
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
dev = torch.device("cuda:0")
SEED = 20250613 + 1000
n = int(float(os.environ.get("SPX_N", "1e8")))
rdivs = [float(v) for v in os.environ.get("SPX_RDIVS", "100,2").split(",")]
cfgs = [int(v) for v in os.environ.get("SPX_CFGS", "0,4,160,16,32").split(",")]


def synth(m, stream, kind):
    t = torch.empty(m, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), m, SEED, stream, kind, ctypes.c_double(1.0)))
    return t


x, sj, q = synth(n, 0, 1), synth(n, 1, 0), synth(n, 2, 1)
if os.environ.get("SPX_TIES") == "1":
    q = torch.round(q * 4) / 4; x.zero_(); sj.zero_()
y = torch.empty_like(q); yref = torch.empty_like(q)
print("lib", os.environ.get("SPX_LIB_NAME", "libspx.so"), "n", n, "ties" if os.environ.get("SPX_TIES") == "1" else "generic", flush=True)
for rd in rdivs:
    r = max(1, int(n / rd))
    psi = s.shifted(s.shifted(s.IndBallL0(r), x, 1.0, s.NormLinf(1.0)), sj)
    L.spx_ctx_set_tuning(ctx, 2, 0)
    s.prox_bang(yref, psi, q, 1.0)
    L.spx_ctx_set_tuning(ctx, 2, 1)
    times = {c: [] for c in cfgs}
    same = {}
    for c in cfgs:
        L.spx_ctx_set_tuning(ctx, 10, c)
        y.fill_(float("nan"))
        s.prox_bang(y, psi, q, 1.0)
        same[c] = bool(torch.equal(y.view(torch.int64), yref.view(torch.int64)))
        for _ in range(3): s.prox_bang(y, psi, q, 1.0)
        if os.environ.get("SPX_VERBOSE") == "1": print("  checked key10 =", c, same[c], flush=True)
    for rnd in range(5):
        for c in cfgs:
            L.spx_ctx_set_tuning(ctx, 10, c)
            s.prox_bang(y, psi, q, 1.0)
            ms = ctypes.c_float(); L.spx_timer_start(ctx)
            for _ in range(10): s.prox_bang(y, psi, q, 1.0)
            L.spx_timer_stop(ctx, ctypes.byref(ms))
            times[c].append(ms.value / 10 * 1e3)
            if os.environ.get("SPX_VERBOSE") == "1": print("  timed key10 =", c, times[c][-1], flush=True)
    L.spx_ctx_set_tuning(ctx, 10, 0)
    rc = L.spx_sync(ctx)
    for c in cfgs:
        t = sorted(times[c])
        print("r = n/%-5g key10 = %-4d median %7.1f us  best %7.1f us | %s" % (
            rd, c, t[2], t[0], "bit-identical to the exact select" if same[c] else "DIFFERS"), flush=True)
    print("sync rc", rc, flush=True)
