"""ctypes/numpy front end of the CPU oracle (oracle/spx_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Every function takes/returns numpy float64 arrays and mirrors one reference prox! body
(file:line citations are in spx_oracle.c).  `mask` is a uint8 array (1 = index selected) or None.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SPX_ORACLE_SO") or os.path.join(_HERE, "libspx_oracle.so")  # override: sanitizer build (make asan)

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_c_uint8_p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile oracle/spx_oracle.c with gcc (no FMA contraction, no fast-math)."""
    src = os.path.join(_HERE, "spx_oracle.c")
    if os.environ.get("SPX_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] if force else ["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        d, i64, dp, ip, up = ctypes.c_double, ctypes.c_int64, _c_double_p, _c_int64_p, _c_uint8_p
        base = [dp, dp, dp, dp, i64]
        L.orc_prox_l1.argtypes = base + [d, d]
        L.orc_prox_l0.argtypes = base + [d, d]
        L.orc_prox_lhalf.argtypes = base + [d, d]
        box = base + [d, d, dp, dp, d, d, up]
        L.orc_prox_l1_box.argtypes = box
        L.orc_prox_l0_box.argtypes = box
        L.orc_prox_lhalf_box.argtypes = box
        L.orc_prox_indball_l0.argtypes = base + [i64]
        L.orc_prox_indball_l0_binf.argtypes = base + [i64, d]
        L.orc_sortperm_indball.argtypes = [ip, dp, dp, dp, i64]
        L.orc_sortperm_indball.restype = None
        L.orc_prox_indball_l0_perm.argtypes = base + [ip, i64, d, ctypes.c_int]
        L.orc_prox_indball_l0_perm.restype = None
        L.orc_prox_group_l2.argtypes = base + [ip, i64, i64, dp, d]
        L.orc_prox_group_l2_binf.argtypes = base + [ip, i64, i64, dp, d, d]
        ib = [dp, dp, dp, dp, dp, i64, d]
        L.orc_iprox_l1.argtypes = ib
        L.orc_iprox_l0.argtypes = ib
        L.orc_iprox_l1.restype = i64
        L.orc_iprox_l0.restype = i64
        L.orc_iprox_l1_box.argtypes = ib + [dp, dp, d, d, up]
        L.orc_iprox_l0_box.argtypes = ib + [dp, dp, d, d, up]
        L.orc_iprox_l1_box.restype = None
        L.orc_iprox_l0_box.restype = None
        L.orc_iprox_zero.argtypes = [d, d, d, d]
        L.orc_iprox_zero.restype = d
        L.orc_obj_plain.argtypes = [ctypes.c_int, dp, dp, dp, i64, d]
        L.orc_obj_plain.restype = d
        L.orc_obj_box.argtypes = [ctypes.c_int, dp, dp, dp, i64, d, dp, dp, d, d, up]
        L.orc_obj_box.restype = d
        L.orc_obj_indball_l0.argtypes = [dp, dp, dp, i64, i64, d]
        L.orc_obj_indball_l0.restype = d
        L.orc_obj_group_l2.argtypes = [dp, dp, dp, i64, ip, i64, i64, dp, d]
        L.orc_obj_group_l2.restype = d
        L.orc_prox_group_l2_idx.argtypes = base + [ip, ip, i64, dp, d]
        L.orc_prox_group_l2_idx.restype = None
        L.orc_prox_group_l2_binf_idx.argtypes = base + [ip, ip, i64, dp, d, d]
        L.orc_prox_group_l2_binf_idx.restype = None
        L.orc_obj_group_l2_idx.argtypes = [dp, dp, dp, i64, ip, ip, i64, dp, d]
        L.orc_obj_group_l2_idx.restype = d
        L.orc_obj_l1_b2.argtypes = [dp, dp, dp, i64, d, d]
        L.orc_obj_l1_b2.restype = d
        L.orc_prox_l1_b2.argtypes = base + [d, d, d, d]
        L.orc_prox_l1_b2.restype = None
        L.orc_rootnormlhalf_prox.argtypes = [dp, dp, i64, d, d]
        L.orc_rootnormlhalf_prox.restype = d
        L.orc_prox_l1_box_mt.argtypes = box + [ctypes.c_int]
        L.orc_prox_l1_box_mt.restype = None
        L.orc_synth_fill.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_int]
        L.orc_synth_fill.restype = None
        L.orc_set_perturbation.argtypes = [ctypes.c_int, ctypes.c_int]
        L.orc_set_perturbation.restype = None
        for name in ("orc_prox_l1", "orc_prox_l0", "orc_prox_lhalf", "orc_prox_l1_box", "orc_prox_l0_box",
                     "orc_prox_lhalf_box", "orc_prox_indball_l0", "orc_prox_indball_l0_binf",
                     "orc_prox_group_l2", "orc_prox_group_l2_binf"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


class perturbed:
    """with oracle.perturbed(norm_ulps, pow_ulps): ...  -- sensitivity probe (spx_oracle.c, "sensitivity probes"): every
    norm / every `^` of the RootNormLhalf closed forms is moved by that many ulps inside the block.  tests/arbiter.py only."""

    def __init__(self, norm_ulps=0, pow_ulps=0):
        self.k = (int(norm_ulps), int(pow_ulps))

    def __enter__(self):
        lib().orc_set_perturbation(*self.k)

    def __exit__(self, *exc):
        lib().orc_set_perturbation(0, 0)
        return False


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def _dp(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


def _prep(q, xk, sj):
    q, xk, sj = _f64(q), _f64(xk), _f64(sj)
    n = q.shape[0]
    assert xk.shape[0] == n and sj.shape[0] == n
    return q, xk, sj, n, np.empty(n, dtype=np.float64)


def _bounds(l, u, n):
    lv = uv = None
    ls = us = 0.0
    if np.ndim(l) == 0:
        ls = float(l)
    else:
        lv = _f64(l)
        assert lv.shape[0] == n
    if np.ndim(u) == 0:
        us = float(u)
    else:
        uv = _f64(u)
        assert uv.shape[0] == n
    return lv, uv, ls, us


def _mask(mask, n):
    if mask is None:
        return None, None
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    assert m.shape[0] == n
    return m, m.ctypes.data_as(_c_uint8_p)


def mask_from_selected(selected, n):
    """`selected` = iterable of 1-based indices (any order, duplicates allowed); membership only."""
    m = np.zeros(n, dtype=np.uint8)
    idx = np.asarray(list(selected), dtype=np.int64) - 1
    idx = idx[(idx >= 0) & (idx < n)]
    m[idx] = 1
    return m


def prox_l1(q, xk, sj, lam, sigma):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lib().orc_prox_l1(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma)
    return y


def prox_l0(q, xk, sj, lam, sigma):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lib().orc_prox_l0(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma)
    return y


def prox_lhalf(q, xk, sj, lam, sigma):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lib().orc_prox_lhalf(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma)
    return y


def _box(fn, q, xk, sj, lam, sigma, l, u, mask):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lv, uv, ls, us = _bounds(l, u, n)
    m, mp = _mask(mask, n)
    fn(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma, _dp(lv), _dp(uv), ls, us, mp)
    return y


def prox_l1_box(q, xk, sj, lam, sigma, l, u, mask=None):
    return _box(lib().orc_prox_l1_box, q, xk, sj, lam, sigma, l, u, mask)


def prox_l1_box_mt(q, xk, sj, lam, sigma, l, u, threads, mask=None, out=None):
    """orc_prox_l1_box over `threads` host threads (an upper bound for the CPU, not the single-threaded reference)"""
    q, xk, sj, n, y = _prep(q, xk, sj)
    if out is not None:
        y = out
    lv, uv, ls, us = _bounds(l, u, n)
    m, mp = _mask(mask, n)
    lib().orc_prox_l1_box_mt(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma, _dp(lv), _dp(uv), ls, us, mp, int(threads))
    return y


def synth_fill(n, seed, stream, kind, scale=1.0, threads=1):
    """host twin of the library's spx_synth_fill (C; oracle/synth.py is the same thing in numpy)"""
    out = np.empty(int(n), dtype=np.float64)
    lib().orc_synth_fill(_dp(out), int(n), int(seed), int(stream), int(kind), float(scale), int(threads))
    return out


def prox_l0_box(q, xk, sj, lam, sigma, l, u, mask=None):
    return _box(lib().orc_prox_l0_box, q, xk, sj, lam, sigma, l, u, mask)


def prox_lhalf_box(q, xk, sj, lam, sigma, l, u, mask=None):
    return _box(lib().orc_prox_lhalf_box, q, xk, sj, lam, sigma, l, u, mask)


def prox_indball_l0(q, xk, sj, r):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lib().orc_prox_indball_l0(_dp(y), _dp(q), _dp(xk), _dp(sj), n, int(r))
    return y


def prox_indball_l0_binf(q, xk, sj, r, delta):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lib().orc_prox_indball_l0_binf(_dp(y), _dp(q), _dp(xk), _dp(sj), n, int(r), delta)
    return y


def _groups(n, offsets, gsize):
    if offsets is not None:
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        return off, off.ctypes.data_as(_c_int64_p), 0, off.shape[0] - 1
    assert gsize > 0 and n % gsize == 0
    return None, None, int(gsize), n // gsize


def prox_group_l2(q, xk, sj, lam, sigma, offsets=None, gsize=0):
    """Groups = contiguous ranges: 0-based CSR `offsets` (len ngroups+1) or uniform `gsize`."""
    q, xk, sj, n, y = _prep(q, xk, sj)
    y[:] = 0.0
    off, offp, gs, ng = _groups(n, offsets, gsize)
    lam = _f64(lam)
    assert lam.shape[0] == ng
    lib().orc_prox_group_l2(_dp(y), _dp(q), _dp(xk), _dp(sj), n, offp, gs, ng, _dp(lam), sigma)
    return y


def prox_group_l2_binf(q, xk, sj, lam, sigma, delta, offsets=None, gsize=0):
    q, xk, sj, n, y = _prep(q, xk, sj)
    y[:] = 0.0
    off, offp, gs, ng = _groups(n, offsets, gsize)
    lam = _f64(lam)
    assert lam.shape[0] == ng
    lib().orc_prox_group_l2_binf(_dp(y), _dp(q), _dp(xk), _dp(sj), n, offp, gs, ng, _dp(lam), sigma, delta)
    return y


def _index_sets(groups):
    """list of 0-based index lists -> (ptr, index) int64 arrays"""
    ptr = np.zeros(len(groups) + 1, dtype=np.int64)
    for k, g in enumerate(groups):
        ptr[k + 1] = ptr[k] + len(g)
    index = np.ascontiguousarray(np.concatenate([np.asarray(list(g), dtype=np.int64) for g in groups])
                                 if len(groups) else np.zeros(0, dtype=np.int64))
    return ptr, index


def prox_group_l2_idx(q, xk, sj, lam, sigma, groups, delta=None, y0=None):
    """Groups = arbitrary 0-based index sets (the reference's idx::Vector{Vector{Int}}); delta = None: ShiftedGroupNormL2,
    else the Binf form.  y0 = y on entry (indices in no group keep it; default zeros)."""
    q, xk, sj, n, y = _prep(q, xk, sj)
    y[:] = 0.0 if y0 is None else _f64(y0)
    ptr, index = _index_sets(groups)
    lam = _f64(lam)
    assert lam.shape[0] == len(groups)
    ip = lambda a: a.ctypes.data_as(_c_int64_p)
    if delta is None:
        lib().orc_prox_group_l2_idx(_dp(y), _dp(q), _dp(xk), _dp(sj), n, ip(ptr), ip(index), len(groups), _dp(lam), sigma)
    else:
        lib().orc_prox_group_l2_binf_idx(_dp(y), _dp(q), _dp(xk), _dp(sj), n, ip(ptr), ip(index), len(groups), _dp(lam),
                                         sigma, delta)
    return y


def obj_group_l2_idx(y, xk, sj, lam, groups, delta=None):
    y, xk, sj = _f64(y), _f64(xk), _f64(sj)
    ptr, index = _index_sets(groups)
    lam = _f64(lam)
    ip = lambda a: a.ctypes.data_as(_c_int64_p)
    return lib().orc_obj_group_l2_idx(_dp(y), _dp(xk), _dp(sj), y.shape[0], ip(ptr), ip(index), len(groups), _dp(lam),
                                      -1.0 if delta is None else delta)


def rootnormlhalf_prox(x, lam, gamma):
    x = _f64(x).ravel()
    y = np.empty_like(x)
    val = lib().orc_rootnormlhalf_prox(_dp(y), _dp(x), x.shape[0], lam, gamma)
    return y, val


# ---- iprox! (indefinite prox) ---------------------------------------------------------------
class AssertionErrorAt(AssertionError):
    """The reference's `@assert d[i] > 0` (0-based index in .index; y holds the entries written before it)."""

    def __init__(self, index, y):
        super().__init__("AssertionError: d[i] > 0 at i = %d" % index)
        self.index, self.y = index, y


def _iprox_unboxed(fn, g, d, xk, sj, lam):
    g, xk, sj, n, y = _prep(g, xk, sj)
    d = _f64(d)
    assert d.shape[0] == n
    bad = fn(_dp(y), _dp(g), _dp(d), _dp(xk), _dp(sj), n, lam)
    if bad >= 0:
        raise AssertionErrorAt(int(bad), y)
    return y


def iprox_l1(g, d, xk, sj, lam):
    return _iprox_unboxed(lib().orc_iprox_l1, g, d, xk, sj, lam)


def iprox_l0(g, d, xk, sj, lam):
    return _iprox_unboxed(lib().orc_iprox_l0, g, d, xk, sj, lam)


def _iprox_box(fn, g, d, xk, sj, lam, l, u, mask):
    g, xk, sj, n, y = _prep(g, xk, sj)
    d = _f64(d)
    assert d.shape[0] == n
    lv, uv, ls, us = _bounds(l, u, n)
    m, mp = _mask(mask, n)
    fn(_dp(y), _dp(g), _dp(d), _dp(xk), _dp(sj), n, lam, _dp(lv), _dp(uv), ls, us, mp)
    return y


def iprox_l1_box(g, d, xk, sj, lam, l, u, mask=None):
    return _iprox_box(lib().orc_iprox_l1_box, g, d, xk, sj, lam, l, u, mask)


def iprox_l0_box(g, d, xk, sj, lam, l, u, mask=None):
    return _iprox_box(lib().orc_iprox_l0_box, g, d, xk, sj, lam, l, u, mask)


def iprox_zero(d, g, l, u):
    return lib().orc_iprox_zero(float(d), float(g), float(l), float(u))


# ---- psi(y): objective value -----------------------------------------------------------------
_KIND = {"l1": 0, "l0": 1, "lhalf": 2}


def obj_plain(kind, y, xk, sj, lam):
    y, xk, sj, n, _ = _prep(y, xk, sj)
    return lib().orc_obj_plain(_KIND[kind], _dp(y), _dp(xk), _dp(sj), n, lam)


def obj_box(kind, y, xk, sj, lam, l, u, mask=None):
    y, xk, sj, n, _ = _prep(y, xk, sj)
    lv, uv, ls, us = _bounds(l, u, n)
    m, mp = _mask(mask, n)
    return lib().orc_obj_box(_KIND[kind], _dp(y), _dp(xk), _dp(sj), n, lam, _dp(lv), _dp(uv), ls, us, mp)


def obj_indball_l0(y, xk, sj, r, delta=None):
    y, xk, sj, n, _ = _prep(y, xk, sj)
    return lib().orc_obj_indball_l0(_dp(y), _dp(xk), _dp(sj), n, int(r), -1.0 if delta is None else float(delta))


def obj_group_l2(y, xk, sj, lam, offsets=None, gsize=0, delta=None):
    y, xk, sj, n, _ = _prep(y, xk, sj)
    off, offp, gs, ng = _groups(n, offsets, gsize)
    lam = _f64(lam)
    return lib().orc_obj_group_l2(_dp(y), _dp(xk), _dp(sj), n, offp, gs, ng, _dp(lam), -1.0 if delta is None else float(delta))


def obj_l1_b2(y, xk, sj, lam, delta):
    y, xk, sj = _f64(y), _f64(xk), _f64(sj)
    return lib().orc_obj_l1_b2(_dp(y), _dp(xk), _dp(sj), y.shape[0], lam, delta)


def prox_l1_b2(q, xk, sj, lam, sigma, delta, chi_lambda=1.0):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lib().orc_prox_l1_b2(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma, delta, chi_lambda)
    return y


# ---- binary128 arbiter (oracle/spx_oracle_q.c) ------------------------------------------------
# Same literal restatement evaluated in __float128: used by tests/arbiter.py to decide which side is further from
# the exact value of the reference's formula where the HIP result and the Float64 oracle above differ by more than 1e-12.
_SOQ = os.path.join(_HERE, "libspx_oracle_q.so")
_libq = None


def libq():
    global _libq
    if _libq is None:
        src = os.path.join(_HERE, "spx_oracle_q.c")
        if not os.path.exists(_SOQ) or os.path.getmtime(_SOQ) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libspx_oracle_q.so"])
        L = ctypes.CDLL(_SOQ)
        d, i64, dp, ip, up = ctypes.c_double, ctypes.c_int64, _c_double_p, _c_int64_p, _c_uint8_p
        base = [dp, dp, dp, dp, i64]
        L.orcq_prox_lhalf.argtypes = base + [d, d]
        L.orcq_prox_lhalf_box.argtypes = base + [d, d, dp, dp, d, d, up, ctypes.POINTER(ctypes.c_int8)]
        L.orcq_prox_group_l2.argtypes = base + [ip, i64, i64, dp, d, ip, i64]
        L.orcq_prox_group_l2_binf.argtypes = base + [ip, i64, i64, dp, d, d, ip, i64, ctypes.POINTER(ctypes.c_int32), dp]
        L.orcq_prox_l1_b2.argtypes = base + [d, d, d, d]
        for name in ("orcq_prox_lhalf", "orcq_prox_lhalf_box", "orcq_prox_group_l2", "orcq_prox_group_l2_binf",
                     "orcq_prox_l1_b2"):
            getattr(L, name).restype = None
        _libq = L
    return _libq


def q_prox_lhalf(q, xk, sj, lam, sigma):
    q, xk, sj, n, y = _prep(q, xk, sj)
    libq().orcq_prox_lhalf(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma)
    return y


def q_prox_lhalf_box(q, xk, sj, lam, sigma, l, u, mask=None, return_candidate=False):
    q, xk, sj, n, y = _prep(q, xk, sj)
    lv, uv, ls, us = _bounds(l, u, n)
    m, mp = _mask(mask, n)
    cand = np.empty(n, dtype=np.int8)
    libq().orcq_prox_lhalf_box(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma, _dp(lv), _dp(uv), ls, us, mp,
                               cand.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)))
    return (y, cand) if return_candidate else y


def q_prox_group_l2(q, xk, sj, lam, sigma, which, offsets=None, gsize=0, binf_delta=None, y0=None, details=False):
    """binary128 evaluation of ShiftedGroupNormL2 (binf_delta None) / ShiftedGroupNormL2Binf on the groups listed in
    `which` only (it is ~1000x slower than the Float64 oracle); other entries of the result are y0 (default NaN).
    details=True (Binf): also returns (branch, root) per listed group -- branch 0 = zeros by :102, 1 = root, 2 = zeros by :107."""
    q, xk, sj, n, y = _prep(q, xk, sj)
    y[:] = np.nan if y0 is None else _f64(y0)
    off, offp, gs, ng = _groups(n, offsets, gsize)
    lam = _f64(lam)
    assert lam.shape[0] == ng
    which = np.ascontiguousarray(which, dtype=np.int64)
    wp = which.ctypes.data_as(_c_int64_p)
    if binf_delta is None:
        libq().orcq_prox_group_l2(_dp(y), _dp(q), _dp(xk), _dp(sj), n, offp, gs, ng, _dp(lam), sigma, wp, which.shape[0])
        return y
    branch = np.zeros(which.shape[0], dtype=np.int32)
    root = np.zeros(which.shape[0], dtype=np.float64)
    libq().orcq_prox_group_l2_binf(_dp(y), _dp(q), _dp(xk), _dp(sj), n, offp, gs, ng, _dp(lam), sigma, float(binf_delta), wp,
                                   which.shape[0], branch.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _dp(root))
    return (y, branch, root) if details else y


def q_prox_l1_b2(q, xk, sj, lam, sigma, delta, chi_lambda=1.0):
    q, xk, sj, n, y = _prep(q, xk, sj)
    libq().orcq_prox_l1_b2(_dp(y), _dp(q), _dp(xk), _dp(sj), n, lam, sigma, delta, chi_lambda)
    return y


# ---- Float32 build (oracle/spx_oracle_f32.c): the NormL1 / NormL0 families with R = Float32 ---------------------------
_SO32 = os.path.join(_HERE, "libspx_oracle_f32.so")
_lib32 = None
_c_float_p = ctypes.POINTER(ctypes.c_float)


def lib32():
    global _lib32
    if _lib32 is None:
        src = os.path.join(_HERE, "spx_oracle_f32.c")
        src64 = os.path.join(_HERE, "spx_oracle.c")   # (the iprox! block is generated from it)
        if not os.path.exists(_SO32) or os.path.getmtime(_SO32) < max(os.path.getmtime(src), os.path.getmtime(src64)):
            subprocess.check_call(["make", "-C", _HERE, "-s", "libspx_oracle_f32.so"])
        L = ctypes.CDLL(_SO32)
        f, i64, fp, up = ctypes.c_float, ctypes.c_int64, _c_float_p, _c_uint8_p
        L.orc32_prox_l1.argtypes = [fp, fp, fp, fp, i64, f, f]
        L.orc32_prox_l0.argtypes = [fp, fp, fp, fp, i64, f, f]
        L.orc32_prox_l1_box.argtypes = [fp, fp, fp, fp, i64, f, f, fp, fp, f, f, up]
        L.orc32_prox_l0_box.argtypes = [fp, fp, fp, fp, i64, f, f, fp, fp, f, f, up]
        for name in ("orc32_prox_l1", "orc32_prox_l0", "orc32_prox_l1_box", "orc32_prox_l0_box"):
            getattr(L, name).restype = None
        L.orc32_iprox_l1.argtypes = [fp, fp, fp, fp, fp, i64, f]
        L.orc32_iprox_l0.argtypes = [fp, fp, fp, fp, fp, i64, f]
        L.orc32_iprox_l1.restype = L.orc32_iprox_l0.restype = i64
        L.orc32_iprox_l1_box.argtypes = [fp, fp, fp, fp, fp, i64, f, fp, fp, f, f, up]
        L.orc32_iprox_l0_box.argtypes = [fp, fp, fp, fp, fp, i64, f, fp, fp, f, f, up]
        L.orc32_iprox_l1_box.restype = L.orc32_iprox_l0_box.restype = None
        _lib32 = L
    return _lib32


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_c_float_p) if a is not None else None


def prox_f32(op, q, xk, sj, lam, sigma, l=None, u=None, mask=None, aliased=False):
    """op in {"l1", "l0", "l1_box", "l0_box"}; Float32 arrays in, Float32 array out.  aliased=True: y is q itself
    (prox!(q, psi, q, sigma)): ShiftedNormL1's two-pass body then reads the overwritten q."""
    q, xk, sj = _f32(q), _f32(xk), _f32(sj)
    n = q.shape[0]
    y = q.copy() if aliased else np.empty(n, dtype=np.float32)
    qq = y if aliased else q
    lam, sigma = np.float32(lam), np.float32(sigma)
    if op in ("l1", "l0"):
        getattr(lib32(), "orc32_prox_" + op)(_fp(y), _fp(qq), _fp(xk), _fp(sj), n, lam, sigma)
        return y
    lv = uv = None
    ls = us = np.float32(0)
    if np.ndim(l) == 0: ls = np.float32(l)
    else: lv = _f32(l)
    if np.ndim(u) == 0: us = np.float32(u)
    else: uv = _f32(u)
    m, mp = _mask(mask, n)
    getattr(lib32(), "orc32_prox_" + op)(_fp(y), _fp(qq), _fp(xk), _fp(sj), n, lam, sigma, _fp(lv), _fp(uv), ls, us, mp)
    return y


def iprox_f32(op, g, d, xk, sj, lam, l=None, u=None, mask=None):
    """iprox! with R = Float32 (oracle/_gen/iprox_f32.inc); op in {"l1", "l0", "l1_box", "l0_box"}.  The unboxed forms return
    (y, first index with d <= 0 or -1) like their Float64 twins."""
    g, d, xk, sj = _f32(g), _f32(d), _f32(xk), _f32(sj)
    n = g.shape[0]
    y = np.empty(n, dtype=np.float32)
    lam = np.float32(lam)
    if op in ("l1", "l0"):
        bad = getattr(lib32(), "orc32_iprox_" + op)(_fp(y), _fp(g), _fp(d), _fp(xk), _fp(sj), n, lam)
        return y, int(bad)
    lv = uv = None
    ls = us = np.float32(0)
    if np.ndim(l) == 0: ls = np.float32(l)
    else: lv = _f32(l)
    if np.ndim(u) == 0: us = np.float32(u)
    else: uv = _f32(u)
    m, mp = _mask(mask, n)
    getattr(lib32(), "orc32_iprox_" + op)(_fp(y), _fp(g), _fp(d), _fp(xk), _fp(sj), n, lam, _fp(lv), _fp(uv), ls, us, mp)
    return y


def topr_order_f32(q, xk, sj):
    """the stable descending sort of prox_indball_l0_f32, for reuse over many r (its `_order` argument)"""
    q, xk, sj = _f32(q), _f32(xk), _f32(sj)
    v = ((xk + sj).astype(np.float32) + q).astype(np.float32)
    key = (v.view(np.uint32) & np.uint32(0x7fffffff)).astype(np.int64)
    key = np.where(key > 0x7f800000, 0x7fc00000, key)
    return np.argsort(-key, kind="stable")


class TopR:
    """ShiftedIndBallL0(BInf).prox! for MANY r on one (q, xk, sj): the reference's sortperm (:68 / :87 -- stable, |v| descending)
    is computed once, each prox(r[, delta]) then zeroes all but the first r of it and finishes as the reference does.  Bit for
    bit what prox_indball_l0 / prox_indball_l0_binf return (tests/test_oracle_golden.py checks that)."""

    def __init__(self, q, xk, sj):
        self.q, self.xk, self.sj, self.n, _ = _prep(q, xk, sj)
        self.p = np.empty(max(self.n, 1), dtype=np.int64)
        lib().orc_sortperm_indball(self.p.ctypes.data_as(_c_int64_p), _dp(self.q), _dp(self.xk), _dp(self.sj), self.n)

    def prox(self, r, delta=None):
        y = np.empty(self.n, dtype=np.float64)
        lib().orc_prox_indball_l0_perm(_dp(y), _dp(self.q), _dp(self.xk), _dp(self.sj), self.n, self.p.ctypes.data_as(_c_int64_p),
                                       int(r), 0.0 if delta is None else float(delta), 0 if delta is None else 1)
        return y


def prox_indball_l0_f32(q, xk, sj, r, delta=None, _order=None):
    """ShiftedIndBallL0(BInf).prox! with R = Float32 (src/shiftedIndBallL0.jl:54-72, shiftedIndBallL0BInf.jl:73-95), restated in
    numpy: v = (xk + sj) + q in Float32, stable descending sort by |v| (isless order: NaN largest, all NaNs tie; ties by
    ascending index), zero all but the first r, subtract xk + sj, clamp to +-delta.  Small cases only (a full argsort)."""
    q, xk, sj = _f32(q), _f32(xk), _f32(sj)
    n = q.shape[0]
    xs = (xk + sj).astype(np.float32)
    v = (xs + q).astype(np.float32)
    if _order is None:
        key = (v.view(np.uint32) & np.uint32(0x7fffffff)).astype(np.int64)
        key = np.where(key > 0x7f800000, 0x7fc00000, key)
        order = np.argsort(-key, kind="stable")
    else:
        order = _order  # (the caller's cached sort of the same q, xk, sj: topr_order_f32)
    kept = v.copy()
    kept[order[max(int(r), 0):]] = np.float32(0)
    y = (kept - xs).astype(np.float32)
    if delta is not None:
        d = np.float32(delta)
        # Julia min / max: NaN propagates (a NaN entry stays NaN); -0.0 < +0.0 is irrelevant for a clamp at +-delta != 0
        y = np.where(np.isnan(y), y, np.minimum(np.maximum(y, -d), d)).astype(np.float32)
    return y
