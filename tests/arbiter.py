"""Adjudicated parity for the floating-point operators (RootNormLhalf(Box), GroupNormL2(Binf), NormL1B2).

The bar of BASELINE.json is 1e-12 relative against the reference's Float64 evaluation.  Where the reference's own formula is
ill-conditioned (y = val - (xk + sj) cancels; alpha = 1 - sigma*lambda/||w|| cancels; a root next to the pole of step(n);
acos'(a) -> infinity at a -> 1) two correct Float64 evaluations of it differ by more than that, and a fixed looser
tolerance would hide real defects.  These helpers never loosen the bar.  They first apply it as is; every element / group
that fails it is evaluated in binary128 (oracle/spx_oracle_q.c, y_q) and must satisfy

        |y_gpu - y_q|  <=  TOL * scale  +  max_v |y_ref_v - y_q|

where y_ref_v runs over the reference's OWN Float64 evaluations: the literal restatement (oracle/spx_oracle.c) and the same
restatement with the arithmetic the reference does not pin moved by a few ulps either way (`LinearAlgebra.norm`:
BLAS dnrm2 vs a generic loop, NORM_ULPS(m); the last ulp of libm's `^`: POW_ULPS) -- SURVEY 8c(2),(5).  In words: the HIP
result is within the bar of the reference's Float64 value, or it is no further from the exact value of the reference's
formula than the reference's own Float64 evaluation can be.  A unit where the GPU is the worse side fails the test.
`scale` is the scale the plain 1e-12 test uses; nothing is divided by a conditioning factor.

Why an ensemble and not the single literal restatement: measured on the lambda = 1e6 group of
test_group_binf_goldens_and_edge_branches, the literal oracle and the GPU land on the same double root n and are BOTH
2.790e-06 away from the binary128 value (the granularity of n next to the pole), while they differ from each other by
6e-11 (the last ulp of ||w|| in alpha = 1 - sigma*lambda/||w||): which of the two is "closer" is decided by that ulp.
"""
import math

import numpy as np

TOL = 1e-12
POW_ULPS = 1      # Julia's pure-Julia `^` and glibc's pow are each within 1 ulp


def NORM_ULPS(m):
    """ulps by which two legitimate Float64 2-norms of m terms may differ (sequential sum: ~sqrt(m/3)/2 ulp rms)."""
    return 2 + int(math.ceil(math.sqrt(max(int(m), 1)) / 4.0))


class Verdict:
    """What the arbiter saw: n_checked = units that failed the plain bar and went to binary128; gpu_closer = how many of
    them had the GPU strictly closer to the exact value than the literal Float64 oracle; worst_excess = max over units of
    (|gpu - q| - max_v |ref_v - q|) / scale (<= TOL when the test passes)."""

    def __init__(self):
        self.n_units = 0
        self.n_checked = 0
        self.gpu_closer = 0
        self.worst_plain = 0.0
        self.worst_excess = 0.0
        self.worst_ref_error = 0.0   # the reference's own distance from the exact value, on the units checked (/ scale)

    def __repr__(self):
        return ("arbiter: %d of %d units above 1e-12 vs oracle64 (worst %.2e); there the reference's own Float64 error reaches %.2e; "
                "GPU closer to binary128 than the literal oracle in %d; worst excess %.2e" % (
                    self.n_checked, self.n_units, self.worst_plain, self.worst_ref_error, self.gpu_closer, self.worst_excess))


def check_elements(y, ref, scale, exact_fn, variants_fn=None, tol=TOL, max_arbitrated=200_000, what=""):
    """Separable operators.  exact_fn(idx) -> binary128 values (as Float64) at the element indices idx;
    variants_fn(idx) -> list of further Float64 evaluations of the reference at idx (the ensemble)."""
    y, ref = np.asarray(y), np.asarray(ref)
    v = Verdict()
    v.n_units = y.size
    nan_ok = np.isnan(y) & np.isnan(ref)
    diff = np.where(nan_ok, 0.0, np.abs(y - ref))
    scale = np.maximum(scale, 1e-300)
    with np.errstate(invalid="ignore"):
        bad = ~(diff <= tol * scale)
    v.worst_plain = float(np.nanmax(diff / scale)) if y.size else 0.0
    idx = np.flatnonzero(bad)
    v.n_checked = idx.size
    if idx.size == 0:
        return v
    assert idx.size <= max_arbitrated, "%s: %d elements above the bar -- not a conditioning effect" % (what, idx.size)
    yq = exact_fn(idx)
    eg, eo = np.abs(y[idx] - yq), np.abs(ref[idx] - yq)
    env = eo.copy()
    for alt in (variants_fn(idx) if variants_fn else ()):
        env = np.maximum(env, np.abs(alt - yq))
    excess = (eg - env) / scale[idx]
    v.worst_excess = float(np.max(excess))
    v.worst_ref_error = float(np.max(env / scale[idx]))
    v.gpu_closer = int(np.sum(eg < eo))
    worse = excess > tol
    assert not worse.any(), "%s: GPU further from the binary128 value than the reference's Float64 evaluations at %d elements, first %d: gpu %r orc %r q %r" % (
        what, int(worse.sum()), int(idx[worse][0]), float(y[idx[worse][0]]), float(ref[idx[worse][0]]), float(yq[worse][0]))
    return v


def check_groups(y, ref, scale, offsets, exact_fn, variants_fn=None, tol=TOL, max_arbitrated=5000, what=""):
    """Group operators.  offsets: CSR offsets (len ngroups + 1) over y.  exact_fn(groups) -> full-length array holding the
    binary128 values (as Float64) on the listed groups; variants_fn(groups) -> list of such arrays from the ensemble.  A
    group goes to the arbiter when any of its elements fails the plain bar; inside it every element must satisfy the
    arbiter inequality."""
    y, ref = np.asarray(y), np.asarray(ref)
    offsets = np.asarray(offsets, dtype=np.int64)
    ng = offsets.size - 1
    v = Verdict()
    v.n_units = ng
    scale = np.maximum(scale, 1e-300)
    both_nan = np.isnan(y) & np.isnan(ref)
    diff = np.where(both_nan, 0.0, np.abs(y - ref))
    with np.errstate(invalid="ignore"):
        bad = ~(diff <= tol * scale)
    v.worst_plain = float(np.nanmax(diff / scale)) if y.size else 0.0
    if not bad.any():
        return v
    gid = np.searchsorted(offsets, np.flatnonzero(bad), side="right") - 1
    groups = np.unique(gid)
    v.n_checked = groups.size
    assert groups.size <= max_arbitrated, "%s: %d groups above the bar -- not a conditioning effect" % (what, groups.size)
    yq = exact_fn(groups)
    alts = variants_fn(groups) if variants_fn else []
    for g in groups:
        sl = slice(int(offsets[g]), int(offsets[g + 1]))
        eg, eo = np.abs(y[sl] - yq[sl]), np.abs(ref[sl] - yq[sl])
        env = eo.copy()
        for alt in alts:
            env = np.maximum(env, np.abs(alt[sl] - yq[sl]))
        excess = (eg - env) / scale[sl]
        v.worst_excess = max(v.worst_excess, float(np.max(excess)))
        v.worst_ref_error = max(v.worst_ref_error, float(np.max(env / scale[sl])))
        v.gpu_closer += int(np.max(eg) < np.max(eo))
        assert np.all(excess <= tol), (
            "%s: group %d: GPU further from the binary128 value than the reference's Float64 evaluations: excess %.3e "
            "(gpu err %.3e, literal oracle64 err %.3e, ensemble %.3e, scale %.3e)" % (
                what, int(g), float(np.max(excess)), float(np.max(eg)), float(np.max(eo)), float(np.max(env)), float(np.max(scale[sl]))))
    return v


# ---------------------------------------------------------------- operator-specific front ends
def lhalf_scale(ref, x, sj, q):
    """y = val - (xk + sj): the operands' scale."""
    return np.maximum(np.maximum(np.abs(ref), np.abs(x + sj)), np.abs(q))


def check_lhalf(orc, y, ref, q, x, sj, lam, sigma, box=None, mask=None, what="lhalf"):
    """box = None (ShiftedRootNormLhalf) or (l, u) with scalars or arrays."""
    scale = lhalf_scale(ref, x, sj, q)

    def sub(idx):
        if box is None:
            return None, None, None
        l, u = box
        li = l if np.ndim(l) == 0 else np.asarray(l)[idx]
        ui = u if np.ndim(u) == 0 else np.asarray(u)[idx]
        return li, ui, (None if mask is None else np.asarray(mask)[idx])

    def exact(idx):
        li, ui, mi = sub(idx)
        if box is None:
            return orc.q_prox_lhalf(q[idx], x[idx], sj[idx], lam, sigma)
        return orc.q_prox_lhalf_box(q[idx], x[idx], sj[idx], lam, sigma, li, ui, mask=mi)

    def variants(idx):
        li, ui, mi = sub(idx)
        out = []
        for k in (POW_ULPS, -POW_ULPS):
            with orc.perturbed(0, k), np.errstate(all="ignore"):
                out.append(orc.prox_lhalf(q[idx], x[idx], sj[idx], lam, sigma) if box is None else
                           orc.prox_lhalf_box(q[idx], x[idx], sj[idx], lam, sigma, li, ui, mask=mi))
        return out

    return check_elements(y, ref, scale, exact, variants, what=what)


def group_scale(ref, q, x, sj, offsets):
    """max(|y_i|, |xk_i + sj_i|, ||S||_2 of the group): norm-relative inside a group and, as for RootNormLhalf,
    the scale of the operands of the last step y[idx] .-= xk[idx] + sj[idx] (:116) -- with S = 0 (q = -(xk + sj)) the group
    norm vanishes and that subtraction is all that is left."""
    S = (q + x) + sj
    offsets = np.asarray(offsets, dtype=np.int64)
    sizes = np.diff(offsets)
    lo, hi = int(offsets[0]), int(offsets[-1])
    ss = np.zeros(sizes.size)
    nz = sizes > 0
    if nz.any():
        ss[nz] = np.add.reduceat((S[lo:hi] ** 2), (offsets[:-1] - lo)[nz])
    scale = np.maximum(np.abs(ref), np.abs(x + sj))
    scale[lo:hi] = np.maximum(scale[lo:hi], np.repeat(np.sqrt(ss), sizes))
    return scale


def check_group(orc, y, ref, q, x, sj, lam, sigma, offsets, delta=None, what="group", max_arbitrated=5000):
    """ShiftedGroupNormL2 (delta None) / ShiftedGroupNormL2Binf on contiguous groups given by CSR offsets."""
    offsets = np.asarray(offsets, dtype=np.int64)
    lam = np.asarray(lam, dtype=np.float64)
    scale = group_scale(ref, q, x, sj, offsets)

    def exact(groups):
        return orc.q_prox_group_l2(q, x, sj, lam, sigma, groups, offsets=offsets, binf_delta=delta)

    def variants(groups):
        # the listed groups packed into a small problem of their own, run through the Float64 oracle with its norms moved
        sizes = (offsets[groups + 1] - offsets[groups]).astype(np.int64)
        sub_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        take = np.concatenate([np.arange(offsets[g], offsets[g + 1]) for g in groups]) if groups.size else np.zeros(0, np.int64)
        k = NORM_ULPS(sizes.max() if sizes.size else 1)
        out = []
        for ku in (k, -k):
            with orc.perturbed(ku, 0), np.errstate(all="ignore"):
                sub = (orc.prox_group_l2(q[take], x[take], sj[take], lam[groups], sigma, offsets=sub_off) if delta is None else
                       orc.prox_group_l2_binf(q[take], x[take], sj[take], lam[groups], sigma, delta, offsets=sub_off))
            full = np.full(y.shape, np.nan)
            full[take] = sub
            out.append(full)
        return out

    return check_groups(y, ref, scale, offsets, exact, variants, what=what, max_arbitrated=max_arbitrated)


def zero_pattern(y, x, sj, offsets):
    """Per group: True when the prox put the whole group to zero, i.e. y == -(xk + sj) exactly on it."""
    offsets = np.asarray(offsets, dtype=np.int64)
    z = (y == -(x + sj)) | ((y == 0) & ((x + sj) == 0))
    sizes = np.diff(offsets)
    out = np.ones(sizes.size, dtype=bool)
    nz = sizes > 0
    if nz.any():
        out[nz] = np.logical_and.reduceat(z[int(offsets[0]):int(offsets[-1])], (offsets[:-1] - offsets[0])[nz])
    return out
