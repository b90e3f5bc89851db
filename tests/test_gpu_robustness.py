"""Round 3: what happens when the assumptions of the in-launch synchronised kernels do not hold (ADVICE).

* a grid larger than residency (planted through a test build, -DSPX_TEST_HOOKS, tuning key 100): the waiting workgroups give
  up after a bounded number of polls, the kernel stores NaN, and the NEXT libspx call -- any entry point, no spx_sync --
  returns SPX_ERR_INTERNAL; spx_sync acknowledges and the context works again.  Never a plausible wrong result.
* the residency cap (key 8) sends every such operator through its smaller-grid form: same bits.
* a captured graph keeps working after a later, larger eager call has outgrown the workspace it was captured with."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def s():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import __graft_entry__ as ge
    return ge.build()


def _bits(a, b):
    return np.array_equal(np.asarray(a).view(np.int64), np.asarray(b).view(np.int64))


_CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = 3_000_000
rng = np.random.default_rng(5)
x = torch.from_numpy(rng.normal(size=n)).cuda(); sj = torch.zeros_like(x); q = torch.from_numpy(rng.normal(size=n)).cuda()
y = torch.empty_like(q)
psi = s.shifted(s.shifted(s.IndBallL0(n // 10), x, 1.0, s.NormLinf(1.0)), sj)
L.spx_ctx_set_tuning(ctx, 2, 0)              # the one-launch exact select (v parked in y), not the sampled pipeline
L.spx_ctx_set_tuning(ctx, 11, 0)             # ... nor the form that parks v in LDS (its grid is what the occupancy query allows)
s.prox_bang(y, psi, q, 1.0); torch.cuda.synchronize()
good = y.clone()
assert L.spx_ctx_set_tuning(ctx, 100, 4096) == 0   # 4096 workgroups of 1024 lanes: at most 512 are resident at once
y.fill_(7.0)
rc_launch = L.spx_prox_indball_l0_binf(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, n // 10, ctypes.c_double(1.0))
torch.cuda.synchronize()                      # the caller synchronises ITS way: no spx_sync
L.spx_ctx_set_tuning(ctx, 100, 0)
nan = int(torch.isnan(y).sum())
rc_next = L.spx_prox_l1(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, ctypes.c_double(1.0), ctypes.c_double(1.0))
msg = L.spx_last_error().decode()
rc_again = L.spx_prox_l1(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, ctypes.c_double(1.0), ctypes.c_double(1.0))
rc_sync = L.spx_sync(ctx)
rc_sync2 = L.spx_sync(ctx)
s.prox_bang(y, psi, q, 1.0); torch.cuda.synchronize()
same = bool(torch.equal(y.view(torch.int64), good.view(torch.int64)))
print("RESULT", rc_launch, nan, rc_next, rc_again, rc_sync, rc_sync2, int(same), "|", msg)
"""


def test_a_grid_beyond_residency_is_an_error_never_garbage(s):
    """Runs in a child process with the hooks build (libspx_hooks.so: key 100 + a poll limit of 2^12 instead of 2^22, so that the
    planted wait costs milliseconds)."""
    lib = os.path.join(ROOT, "shiftedproximaloperators.jl_amd", "lib", "libspx_hooks.so")
    if not os.path.exists(lib):
        pytest.skip("libspx_hooks.so not built")
    env = dict(os.environ, SPX_LIB_NAME="libspx_hooks.so", SPX_NO_BUILD="1")
    out = subprocess.run([sys.executable, "-c", _CHILD % ROOT], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, (out.stdout[-2000:], out.stderr[-2000:])
    f = line[0].split("|")[0].split()[1:]
    rc_launch, nan, rc_next, rc_again, rc_sync, rc_sync2, same = (int(v) for v in f)
    n = 3_000_000
    assert rc_launch == 0                       # the launch itself is asynchronous and cannot know
    assert nan == n, "the abandoned launch must poison its whole result (%d of %d NaN)" % (nan, n)
    assert rc_next == 7 and rc_again == 7, "every entry point refuses while the status word is raised"
    assert "timed out" in line[0]
    assert rc_sync == 7 and rc_sync2 == 0, "spx_sync reports once and resets"
    assert same == 1, "the context works again after the acknowledgement"


_CHILD_TAIL = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
n = 6_000_000                                  # above what the LDS form holds on chip; key 11 = 2: no register slots beyond
assert L.spx_ctx_set_tuning(ctx, 11, 2) == 0   # it (round 4: 6 Mi otherwise) -- the sampled pipeline, whose tail kernel is the subject
rng = np.random.default_rng(5)
res = []
for case in ("ties", "moderate_ties", "ties_aliased"):
    if case == "moderate_ties":                # 5000 candidates share the cut's key (more than the short list of k_s2_finish
        qh = rng.normal(size=n)                # holds, fewer than the 0.1 %% of the vector that make a class): kTodoCandSelect
        at = rng.choice(n, size=5000, replace=False)
        qh[at] = np.where(rng.random(5000) < 0.5, 1.0, -1.0)
        r = int((np.abs(qh) > 1.0).sum()) + 2500
        q = torch.from_numpy(qh).cuda()
    else:                                      # a lattice: the cut lies inside a class (kTodoTieScan + kTodoFinal)
        q = torch.from_numpy(np.round(rng.normal(size=n) * 4) / 4).cuda()
        r = n // 3
    x = torch.zeros_like(q); sj = torch.zeros_like(q)
    y = q if case == "ties_aliased" else torch.empty_like(q)   # y === q: the two-pass form (k_sel_final_q stores y)
    q0 = q.clone()
    rc0 = L.spx_prox_indball_l0_binf(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, r, ctypes.c_double(1.0))
    torch.cuda.synchronize()
    good = y.clone(); q.copy_(q0)
    assert rc0 == 0 and not bool(torch.isnan(good).any())
    assert L.spx_ctx_set_tuning(ctx, 101, 1) == 0   # the last workgroup of k_s2_tail never arrives
    if y is not q: y.fill_(7.0)
    rc_launch = L.spx_prox_indball_l0_binf(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, r, ctypes.c_double(1.0))
    torch.cuda.synchronize()
    L.spx_ctx_set_tuning(ctx, 101, 0)
    nan = int(torch.isnan(y).sum())
    rc_next = L.spx_prox_l1(ctx, good.data_ptr(), q0.data_ptr(), x.data_ptr(), sj.data_ptr(), n, ctypes.c_double(1.0), ctypes.c_double(1.0))
    rc_sync = L.spx_sync(ctx)
    q.copy_(q0)
    rc_after = L.spx_prox_indball_l0_binf(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, r, ctypes.c_double(1.0))
    torch.cuda.synchronize()
    res.append((case, rc_launch, nan, rc_next, rc_sync, rc_after, int(torch.isnan(y).sum())))
print("RESULT", res)
"""


def test_the_top_r_tail_kernel_poisons_its_result_too(s):
    """a wait that expires inside k_s2_tail (the candidate select over the regions, the tie scan) used to leave the
    main pass's speculative values / a made-up index cut in y.  Hooks build, key 101: one workgroup of the tail never arrives."""
    lib = os.path.join(ROOT, "shiftedproximaloperators.jl_amd", "lib", "libspx_hooks.so")
    if not os.path.exists(lib):
        pytest.skip("libspx_hooks.so not built")
    env = dict(os.environ, SPX_LIB_NAME="libspx_hooks.so", SPX_NO_BUILD="1")
    out = subprocess.run([sys.executable, "-c", _CHILD_TAIL % ROOT], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, (out.stdout[-2000:], out.stderr[-2000:])
    res = eval(line[0][len("RESULT"):])
    assert len(res) == 3
    for case, rc_launch, nan, rc_next, rc_sync, rc_after, nan_after in res:
        assert rc_launch == 0, case
        assert nan == 6_000_000, "%s: the abandoned tail must poison the whole result (%d NaN)" % (case, nan)
        assert rc_next == 7 and rc_sync == 7, case
        assert rc_after == 0 and nan_after == 0, case


_CHILD_TEAM = r"""
import ctypes, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
s = ge.build(); L = s._lib.load(); ctx = s.context("cuda:0")
rng = np.random.default_rng(5)
res = []
for n, binf, fast in ((120_000, 0, 1), (120_000, 1, 1), (3_000_000, 0, 1), (3_000_000, 1, 1), (3_000_000, 1, 0)):
    x = torch.from_numpy(rng.normal(size=n)).cuda(); sj = torch.from_numpy(rng.uniform(-0.5, 0.5, size=n)).cuda()
    q = torch.from_numpy(rng.normal(size=n)).cuda(); y = torch.empty_like(q)
    lam = torch.tensor([0.4 * n ** 0.5], dtype=torch.float64, device="cuda")
    L.spx_ctx_set_tuning(ctx, 14, fast)
    def call():
        if binf:
            return L.spx_prox_group_l2_binf(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, None, n, 1, lam.data_ptr(), ctypes.c_double(0.9), ctypes.c_double(1.0))
        return L.spx_prox_group_l2(ctx, y.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, None, n, 1, lam.data_ptr(), ctypes.c_double(0.9))
    assert call() == 0
    torch.cuda.synchronize()
    good = y.clone()
    assert not bool(torch.isnan(good).any())
    assert L.spx_ctx_set_tuning(ctx, 102, 1) == 0      # the last workgroup of the team arrives late
    y.fill_(7.0)
    rc_launch = call()
    torch.cuda.synchronize()
    L.spx_ctx_set_tuning(ctx, 102, 0)
    nan = int(torch.isnan(y).sum())
    rc_next = L.spx_prox_l1(ctx, good.data_ptr(), q.data_ptr(), x.data_ptr(), sj.data_ptr(), n, ctypes.c_double(1.0), ctypes.c_double(1.0))
    rc_sync = L.spx_sync(ctx)
    rc_after = call()
    torch.cuda.synchronize()
    same = bool(torch.equal(y.view(torch.int64), (good if True else y).view(torch.int64))) if False else int(torch.isnan(y).sum()) == 0
    res.append((n, binf, fast, rc_launch, nan, rc_next, rc_sync, rc_after, int(same)))
L.spx_ctx_set_tuning(ctx, 14, 1)
print("RESULT", res)
"""


def test_a_team_of_workgroups_poisons_its_result_too(s):
    """Round 4: the team form (csrc/spx_group_team.hip) synchronises inside its launches like top-r and B2.  Hooks build, tuning key
    102: the last workgroup of the team takes part in no reduction and only runs once the others have given up -- on chip and
    streamed, plain and Binf (fast path and generic body): NaN over the whole group, SPX_ERR_INTERNAL on the next call, a clean
    result after spx_sync."""
    lib = os.path.join(ROOT, "shiftedproximaloperators.jl_amd", "lib", "libspx_hooks.so")
    if not os.path.exists(lib):
        pytest.skip("libspx_hooks.so not built")
    env = dict(os.environ, SPX_LIB_NAME="libspx_hooks.so", SPX_NO_BUILD="1")
    out = subprocess.run([sys.executable, "-c", _CHILD_TEAM % ROOT], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, (out.stdout[-2000:], out.stderr[-2000:])
    res = eval(line[0][len("RESULT"):])
    assert len(res) == 5
    for n, binf, fast, rc_launch, nan, rc_next, rc_sync, rc_after, clean in res:
        what = (n, binf, fast)
        assert rc_launch == 0, what
        assert nan == n, "%r: the abandoned launch must poison the whole group (%d of %d NaN)" % (what, nan, n)
        assert rc_next == 7 and rc_sync == 7, what
        assert rc_after == 0 and clean == 1, what


def test_residency_cap_takes_the_smaller_grid_forms(s, orc):
    """key 8: B2 and top-r with the resident grid capped at 1, 5 and 100 workgroups (register-resident forms hand over to the
    streaming / parked forms, the sampled pipeline to the exact select) -- bits / 1e-12 as without the cap."""
    import torch
    L = s._lib.load()
    ctx = s.context("cuda:0")
    rng = np.random.default_rng(11)
    try:
        for n in (20_000, 300_000, 2_500_000, 4_200_001, 6_400_001):
            x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n); q = rng.normal(size=n)
            xd, sd, qd = (torch.from_numpy(t).cuda() for t in (x, sj, q))
            r = n // 20
            ref_top = orc.prox_indball_l0_binf(q, x, sj, r, 0.7)
            ref_b2 = orc.prox_l1_b2(q, x, sj, 1.0, 1.0, 1.0, 1.0)
            for cap in (0, 1, 5, 100):
                s._lib.check(L.spx_ctx_set_tuning(ctx, 8, cap))
                psi = s.shifted(s.shifted(s.IndBallL0(r), xd, 0.7, s.NormLinf(1.0)), sd)
                assert _bits(s.prox(psi, qd, 1.0).cpu().numpy(), ref_top), (n, cap)
                psi = s.shifted(s.shifted(s.NormL1(1.0), xd, 1.0, s.NormL2(1.0)), sd)
                got = s.prox(psi, qd, 1.0).cpu().numpy()
                assert np.max(np.abs(got - ref_b2)) <= 1e-12 * max(np.linalg.norm(ref_b2), np.linalg.norm(x)), (n, cap)
    finally:
        s._lib.check(L.spx_ctx_set_tuning(ctx, 8, 0))
    s._lib.check(L.spx_sync(ctx))


def test_replay_after_a_larger_eager_call_outgrew_the_workspace(s, orc):
    """the graph's kernel nodes hold the workspace address of capture time; a later eager call that needs a larger
    workspace used to free that block.  Now it is retired until spx_ctx_destroy."""
    import torch
    n = 40_000
    rng = np.random.default_rng(3)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ng = n // 8
        x = rng.normal(size=n); sj = rng.uniform(-0.5, 0.5, size=n)
        xd, sd = torch.from_numpy(x).cuda(), torch.from_numpy(sj).cuda()
        qd = torch.zeros(n, dtype=torch.float64, device="cuda")
        y = torch.zeros_like(qd); val = torch.zeros(1, dtype=torch.float64, device="cuda")
        lam = rng.uniform(0.5, 1.5, size=ng)
        psi = s.shifted(s.shifted(s.GroupNormL2.uniform(torch.from_numpy(lam).cuda(), 8), xd, 1.0, s.NormLinf(1.0)), sd)

        def it():
            s.prox_bang(y, psi, qd, 1.0)       # deferred-group list in the workspace
            with s.device_values(val):
                psi(y)                         # objective partials in the workspace
        qd.copy_(torch.from_numpy(rng.normal(size=n)))
        it(); it()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        it()
    # a much larger eager call on the SAME context (same stream): the workspace grows
    with torch.cuda.stream(side):
        big = 7_000_000   # (beyond what the one-launch top-r forms hold on chip: the pipeline's candidate regions)
        xb = torch.randn(big, dtype=torch.float64, device="cuda"); zb = torch.zeros_like(xb); qb = torch.randn_like(xb)
        yb = torch.empty_like(qb)
        s.prox_bang(yb, s.shifted(s.shifted(s.IndBallL0(big // 7), xb, 1.0, s.NormLinf(1.0)), zb), qb, 1.0)   # candidate regions: ~150 MB
        junk = torch.full((50_000_000,), 3.0e300, dtype=torch.float64, device="cuda")   # whatever reuses a freed block holds junk
    side.synchronize()
    for rep in range(3):
        q = rng.normal(size=n) * (1.0 + rep)
        qd.copy_(torch.from_numpy(q))
        y.fill_(-777.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ref = orc.prox_group_l2_binf(q, x, sj, lam, 1.0, 1.0, gsize=8)
        assert np.max(np.abs(y.cpu().numpy() - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), rep
    del junk
    s._lib.check(s._lib.load().spx_sync(s.context("cuda:0")))
