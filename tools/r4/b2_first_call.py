"""ShiftedNormL1B2 at n = 1e8, five calls on a FRESH context, for rocprofv3 --pmc WRITE_SIZE (per-dispatch rows): the first
call on a context speculates "trust region inactive" (SpxSyncHeader::b2_last_scaled starts at 0) and stores y in its first
pass; the trust region is active, so the storing pass stores it again (same element -> lane mapping: ordered).  That is the
0.967 GB (one run) against 0.801 GB (another, where smaller B2 calls had run first) of profiles/r03_traffic_all_ops.txt."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load()
dev = torch.device("cuda:0"); ctx = s.context(dev)
n = int(float(os.environ.get("SPX_N", "1e8")))
def synth(stream, kind):
    t = torch.empty(n, dtype=torch.float64, device=dev)
    s._lib.check(L.spx_synth_fill(ctx, ctypes.c_void_p(t.data_ptr()), n, 20250613 + 1000, stream, kind, 1.0)); return t
xk, sj, q = synth(0, 1), synth(1, 0), synth(2, 1); y = torch.empty_like(q)
psi = s.shifted(s.shifted(s.NormL1(1.0), xk, 1.0, s.NormL2(1.0)), sj)
for _ in range(5): s.prox_bang(y, psi, q, 1.0)
torch.cuda.synchronize()
