"""Adjudicated parity for the floating-point operators (RootNormLhalf(Box), GroupNormL2(Binf), NormL1B2).

The bar of BASELINE.json is 1e-12 relative against the reference's Float64 evaluation.  Where the reference's own formula is
ill-conditioned (y = val - (xk + sj) cancels; alpha = 1 - sigma*lambda/||w|| cancels; a root next to the pole of step(n);
acos'(a) -> infinity at a -> 1) two correct Float64 evaluations of it differ by more than that, and a fixed looser
tolerance would hide real defects.  These helpers never loosen the bar.  They first apply it as is; for every element /
group that fails it they evaluate the same formula in binary128 (oracle/spx_oracle_q.c) and require

        |y_gpu - y_q|  <=  TOL * scale  +  |y_oracle64 - y_q|

i.e. the HIP result is within the bar of the reference's Float64 value, or it is at least as close to the exact value of
the reference's formula as the reference's own Float64 evaluation is.  A group/element where the GPU is the worse side
fails the test.  `scale` is the same scale the plain 1e-12 test uses; nothing is divided by a conditioning factor.
"""
import numpy as np

TOL = 1e-12


class Verdict:
    """What the arbiter saw: n_checked = units that failed the plain bar and went to binary128; gpu_closer = how many of
    them had the GPU strictly closer to the exact value than the Float64 oracle; worst_excess = max over units of
    (|gpu - q| - |orc - q|) / scale (<= TOL when the test passes)."""

    def __init__(self):
        self.n_units = 0
        self.n_checked = 0
        self.gpu_closer = 0
        self.worst_plain = 0.0
        self.worst_excess = 0.0

    def __repr__(self):
        return "arbiter: %d of %d units above 1e-12 vs oracle64 (worst %.2e); GPU closer to binary128 in %d; worst excess %.2e" % (
            self.n_checked, self.n_units, self.worst_plain, self.gpu_closer, self.worst_excess)


def check_elements(y, ref, scale, exact_fn, tol=TOL, max_arbitrated=200_000, what=""):
    """Separable operators.  exact_fn(idx) -> binary128 values (as Float64) at the element indices idx."""
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
    excess = (eg - eo) / scale[idx]
    v.worst_excess = float(np.max(excess))
    v.gpu_closer = int(np.sum(eg < eo))
    worse = excess > tol
    assert not worse.any(), "%s: GPU further from the binary128 value than the Float64 oracle at %d elements, first %d: gpu %r orc %r q %r" % (
        what, int(worse.sum()), int(idx[worse][0]), float(y[idx[worse][0]]), float(ref[idx[worse][0]]), float(yq[worse][0]))
    return v


def check_groups(y, ref, scale, offsets, exact_fn, tol=TOL, max_arbitrated=5000, what=""):
    """Group operators.  offsets: CSR offsets (len ngroups + 1) over y.  exact_fn(groups) -> full-length array holding the
    binary128 values (as Float64) on the listed groups.  A group goes to the arbiter when any of its elements fails the
    plain bar; inside it every element must satisfy the arbiter inequality."""
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
    for g in groups:
        sl = slice(int(offsets[g]), int(offsets[g + 1]))
        eg, eo = np.abs(y[sl] - yq[sl]), np.abs(ref[sl] - yq[sl])
        excess = (eg - eo) / scale[sl]
        v.worst_excess = max(v.worst_excess, float(np.max(excess)))
        v.gpu_closer += int(np.max(eg) < np.max(eo))
        assert np.all(excess <= tol), "%s: group %d: GPU further from the binary128 value than the Float64 oracle: excess %.3e (gpu err %.3e, oracle64 err %.3e, scale %.3e)" % (
            what, int(g), float(np.max(excess)), float(np.max(eg)), float(np.max(eo)), float(np.max(scale[sl])))
    return v


# ---------------------------------------------------------------- operator-specific front ends
def lhalf_scale(ref, x, sj, q):
    """y = val - (xk + sj): the operands' scale."""
    return np.maximum(np.maximum(np.abs(ref), np.abs(x + sj)), np.abs(q))


def check_lhalf(orc, y, ref, q, x, sj, lam, sigma, box=None, mask=None, what="lhalf"):
    """box = None (ShiftedRootNormLhalf) or (l, u) with scalars or arrays."""
    scale = lhalf_scale(ref, x, sj, q)

    def exact(idx):
        if box is None:
            return orc.q_prox_lhalf(q[idx], x[idx], sj[idx], lam, sigma)
        l, u = box
        li = l if np.ndim(l) == 0 else np.asarray(l)[idx]
        ui = u if np.ndim(u) == 0 else np.asarray(u)[idx]
        return orc.q_prox_lhalf_box(q[idx], x[idx], sj[idx], lam, sigma, li, ui, mask=None if mask is None else np.asarray(mask)[idx])

    return check_elements(y, ref, scale, exact, what=what)


def group_scale(ref, q, x, sj, offsets):
    """|y_i| or the group's ||S||_2, whichever is larger (norm-relative inside a group)."""
    S = (q + x) + sj
    offsets = np.asarray(offsets, dtype=np.int64)
    sizes = np.diff(offsets)
    lo, hi = int(offsets[0]), int(offsets[-1])
    ss = np.zeros(sizes.size)
    nz = sizes > 0
    if nz.any():
        ss[nz] = np.add.reduceat((S[lo:hi] ** 2), (offsets[:-1] - lo)[nz])
    scale = np.abs(ref).copy()
    scale[lo:hi] = np.maximum(scale[lo:hi], np.repeat(np.sqrt(ss), sizes))
    return scale


def check_group(orc, y, ref, q, x, sj, lam, sigma, offsets, delta=None, what="group", max_arbitrated=5000):
    """ShiftedGroupNormL2 (delta None) / ShiftedGroupNormL2Binf on contiguous groups given by CSR offsets."""
    offsets = np.asarray(offsets, dtype=np.int64)
    scale = group_scale(ref, q, x, sj, offsets)

    def exact(groups):
        return orc.q_prox_group_l2(q, x, sj, lam, sigma, groups, offsets=offsets, binf_delta=delta)

    return check_groups(y, ref, scale, offsets, exact, what=what, max_arbitrated=max_arbitrated)


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
