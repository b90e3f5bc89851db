"""In-kernel time line of the team form's Binf streaming path (build: SPX_LIB_NAME=libspx_tmprof.so csrc/build.sh -DSPX_TEAM_PROFILE)."""
import ctypes, os, sys
os.environ.setdefault("SPX_LIB_NAME", "libspx_tmprof.so"); os.environ["SPX_NO_BUILD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
lib = ctypes.CDLL(os.path.join(os.path.dirname(s._lib.__file__), "lib", os.environ["SPX_LIB_NAME"]))
dev = torch.device("cuda:0"); ctx = s.context(dev); chi = s.NormLinf(1.0)
def fill(n, stream, kind):
    t = torch.empty(n, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), n, 20250613, stream, kind, 1.0)); return t
names = ["sample solved", "pass 1 streamed + exchanged", "root found", "y stored"]
for n in [int(float(v)) for v in (sys.argv[1:] or ["3e6", "1e7", "1e8"])]:
    xk, sj, q = fill(n, 0, 1), fill(n, 1, 0), fill(n, 2, 1); y = torch.empty_like(q)
    psi = s.shifted(s.shifted(s.GroupNormL2([0.5 * n ** 0.5]), xk, 1.0, chi), sj)
    for rep in range(3):
        s.prox_bang(y, psi, q, 1.0); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)(); cnt = ctypes.c_int()
    lib.spx_debug_team_stamps(out, ctypes.byref(cnt))
    st = [out[k] for k in range(cnt.value)]
    print("n %d: status %d, %d reductions; " % (n, out[30], out[31]) + "; ".join("%s +%.1f us" % (names[k] if k < len(names) else "?", (st[k + 1] - st[k]) / 100.0) for k in range(len(st) - 1)), flush=True)
