"""Host-side mirror of the reference's operator API for the prox! hot path.

Same names, argument meaning and error behaviour as ShiftedProximalOperators.jl v0.2.2 (Julia's `f!` is
spelled `f_bang` here):

    ψ = shifted(h, xk)                 # h(xk + s)                       src/shifted*.jl constructors
    ψ = shifted(h, xk, Δ, χ[, selected])   # ... + indicator of the l∞ ball of radius Δ (χ = NormLinf(1.0))
    ψ = shifted(h, xk, l, u[, selected])   # ... + indicator of the box [l, u] (scalars or vectors)
    ω = shifted(ψ, sj)                 # second shift, shares ψ.xk / l / u / selected, borrows sj
    prox_bang(y, ψ, q, σ) -> y         # prox!   src/ShiftedProximalOperators.jl:135-152
    iprox_bang(y, ψ, g, d) -> y        # iprox!  :154-180 (ShiftedNormL1/L0 and their Box forms)
    prox(ψ, q, σ) -> ψ.sol             # prox    :189-190
    shift_bang(ψ, v), set_radius_bang(ψ, Δ), set_bounds_bang(ψ, l, u)        # :72-111

Vectors are torch float64 CUDA tensors (device memory plumbing only); like the reference, `xk`, `sj`,
`l`, `u` are borrowed BY REFERENCE (shiftedNormL1Box.jl:22-47) and shift_bang writes into the caller's
tensor.  All arithmetic happens in libspx (HIP); nothing here computes a prox on the host, and there is
no CPU path: CPU torch tensors raise TypeError.  Index sets (`selected`, group ranges) are 0-based here.

Host vectors: a ψ built on numpy float64 arrays (the reference's plain `Vector{Float64}`, as in all of
test/runtests.jl) binds the `spx_host_*` forms of the same entry points: every call stages its vectors through
device memory, runs the same HIP kernels and copies y back (PCIe per call -- for small problems and for running
the reference's tests as written; solvers keep ψ on the device).  All vectors of one ψ / one call must be of the
same kind.
"""
import ctypes
import numbers

import numpy as np
import torch

from . import _lib
from .functions import (GroupNormL2, IndBallL0, NormL0, NormL1, NormL2, NormLinf, ProximableFunction, RootNormLhalf)

# ---------------------------------------------------------------------------------------------
# contexts: one libspx context per (device, stream) so calls are ordered with the caller's torch work
# ---------------------------------------------------------------------------------------------
_ctxs = {}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (the handle without building a Stream object: ~1.5 us per call)


def _ctx(device):
    if device is None:  # host vectors: the current device
        idx = torch.cuda.current_device()
    else:
        idx = device.index if device.index is not None else torch.cuda.current_device()
    if _raw_stream is not None:
        key = (idx, _raw_stream(idx))
    else:
        key = (idx, torch.cuda.current_stream(torch.device("cuda", idx)).cuda_stream)
    c = _ctxs.get(key)
    if c is None:
        L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(L.spx_ctx_create_on_stream(key[0], ctypes.c_void_p(key[1]), ctypes.byref(h)))
        c = _ctxs[key] = h
    return c


_Tensor = torch.Tensor


def _is_host(t):
    return isinstance(t, np.ndarray)


def _n(t):
    return t.numel() if type(t) is _Tensor else (t.size if _is_host(t) else t.numel())


def _dev(t):
    """torch device of a device vector, None for a host (numpy) vector"""
    return t.device if type(t) is _Tensor else (None if _is_host(t) else t.device)


def _vec(t, name, n=None, like=None):
    """Validates a vector argument: a float64 contiguous torch CUDA tensor, or a float64 contiguous numpy array
    (host-pointer forms).  `like`: a vector it must share its kind (and device) with."""
    # (round 4: a call at solver sizes costs ~9 us in this mirror against 3.9 through the C ABI alone -- tools/r4/py_overhead.py;
    #  the usual case, a plain device tensor checked against ψ.xk, takes the few tests below and nothing else)
    if type(t) is _Tensor and type(like) is _Tensor and t.is_cuda and t.dtype is like.dtype and t.dim() == 1:
        m = t.numel()
        if (n is None or m == n) and (m <= 1 or t.stride(0) == 1) and t.device == like.device:
            return t
    if _is_host(t):
        if t.dtype != np.float64:
            raise TypeError("%s must be float64 (got %s)" % (name, t.dtype))
        if t.ndim != 1 or (t.size > 1 and t.strides[0] != 8):
            raise TypeError("%s must be a contiguous vector" % name)
    else:
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch tensor on the GPU or a numpy array (got %s)" % (name, type(t)))
        if not t.is_cuda:
            raise TypeError("%s must live in device memory (cuda); libspx computes nothing on the CPU (numpy arrays "
                            "are staged through the GPU)" % name)
        # float64, or float32 for the operators that have a Float32 form (the reference is generic in R <: Real); all the
        # vectors of one ψ share the element type
        want = like.dtype if isinstance(like, torch.Tensor) else None
        if want is None and t.dtype not in (torch.float64, torch.float32):
            raise TypeError("%s must be float64 or float32 (got %s)" % (name, t.dtype))
        if want is not None and t.dtype != want:
            raise TypeError("%s must be %s like ψ.xk (got %s)" % (name, want, t.dtype))
        if t.dim() != 1 or (t.numel() > 1 and t.stride(0) != 1):
            raise TypeError("%s must be a contiguous vector" % name)
    if like is not None and _dev(t) != _dev(like):
        raise TypeError("%s must be of the same kind (host array / device tensor on the same device) as ψ.xk" % name)
    if n is not None and _n(t) != n:
        raise IndexError("BoundsError: %s has length %d, expected %d" % (name, _n(t), n))
    return t


_NULL = ctypes.c_void_p(0)


def _ptr(t):
    if t is None:
        return _NULL
    if type(t) is _Tensor:
        return ctypes.c_void_p(t.data_ptr()) if t.numel() else _NULL
    if _n(t) == 0:
        return _NULL
    return ctypes.c_void_p(t.ctypes.data if _is_host(t) else t.data_ptr())


def _iptr(t):
    """address of an index array even when it is empty-but-present (ptr of zero groups has one entry)"""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.ctypes.data if _is_host(t) else t.data_ptr())


def _empty_like(t):
    return np.empty_like(t) if _is_host(t) else torch.empty_like(t)


def _zeros_like(t):
    return np.zeros_like(t) if _is_host(t) else torch.zeros_like(t)


def _copy_into(dst, src):
    if _is_host(dst):
        np.copyto(dst, src)
    else:
        dst.copy_(src)


def _is_real(v):
    return type(v) is float or type(v) is int or isinstance(v, numbers.Real)


# ---------------------------------------------------------------------------------------------
# Strided views as xk (the reference's `x = view(y, 1:2:10)`, test/runtests.jl:196-209).  libspx takes unit-stride vectors:
# a strided xk is kept as (the caller's view, a packed copy); the copy is refreshed from the view before every call that
# reads xk (spx_copy_strided: one extra pass over xk) and shift! writes through to the view -- the caller's array stays the
# storage, as in the reference.  sj, sol, y, q are dense in the reference too (`similar(xk)` is a Vector).
# ---------------------------------------------------------------------------------------------
def _stride_of(t):
    """element stride of a 1-D vector (1 = contiguous), or None if it is not a positive-stride 1-D vector"""
    if _is_host(t):
        if t.ndim != 1 or t.itemsize == 0 or t.strides[0] % t.itemsize:
            return None
        st = t.strides[0] // t.itemsize
    else:
        if not isinstance(t, torch.Tensor) or t.dim() != 1:
            return None
        st = t.stride(0)
    return st if (st >= 1 or _n(t) <= 1) else None


class _StridedRef:
    def __init__(self, view):
        self.view = view
        self.stride = _stride_of(view)
        self.packed = (np.empty(view.shape, dtype=view.dtype) if _is_host(view)
                       else torch.empty(view.numel(), dtype=view.dtype, device=view.device))
        self.refresh()

    def _copy(self, dst, ds, src, ss):
        if _is_host(self.view):
            np.copyto(dst, src)
            return
        n = self.view.numel()
        _lib.check(_lib.load().spx_copy_strided(_ctx(self.view.device), ctypes.c_void_p(dst.data_ptr()), ds,
                                                ctypes.c_void_p(src.data_ptr()), ss, n, self.view.element_size()))

    def refresh(self):      # packed <- view
        self._copy(self.packed, 1, self.view, self.stride)

    def write_through(self):  # view <- packed
        self._copy(self.view, self.stride, self.packed, 1)


def _unstrided(x, name):
    """(vector to hand to libspx, _StridedRef or None): a strided 1-D view is packed, everything else goes through _vec"""
    st = _stride_of(x)
    if st is not None and st > 1 and _n(x) > 1:
        if _is_host(x):
            if x.dtype != np.float64:
                raise TypeError("%s must be float64 (got %s)" % (name, x.dtype))
        elif not x.is_cuda or x.dtype not in (torch.float64, torch.float32):
            return x, None   # (_vec raises the right error)
        ref = _StridedRef(x)
        return ref.packed, ref
    return x, None


# ---------------------------------------------------------------------------------------------
# base type                                      src/ShiftedProximalOperators.jl:18,113-121
# ---------------------------------------------------------------------------------------------
class ShiftedProximableFunction:
    def __init__(self, h, xk, sj, shifted_twice):
        self.h = h
        self.xk = _vec(xk, "xk")
        self.sj = _vec(sj, "sj", _n(xk), like=xk)
        self.sol = _empty_like(xk)  # `sol = similar(xk)`
        self.shifted_twice = bool(shifted_twice)
        self.host = _is_host(xk)
        self.f32 = (not self.host) and xk.dtype == torch.float32
        self._xk_ref = None  # set by shifted() when the caller's xk is a strided view (see _StridedRef)

    def _refresh(self):
        if self._xk_ref is not None:
            self._xk_ref.refresh()

    def _sym(self, L, name):
        """the entry point `name` of libspx, its host-pointer form for a ψ on host arrays, or its Float32 form"""
        if self.f32:
            if name + "_f32" not in _lib.SIGNATURES:
                raise TypeError("MethodError: %s has no Float32 form in libspx (Float32 covers psi(y) of every operator and prox! "
                                "of the NormL1 / NormL0 families; convert to float64 for the rest)" % name)
            return getattr(L, name + "_f32")
        return getattr(L, "spx_host_" + name[4:] if self.host else name)

    # ψ.λ / ψ.r sugar (getproperty, :113-121)
    @property
    def λ(self):
        return self.h.lam

    lam = λ

    @property
    def r(self):
        return self.h.r

    def _prox(self, L, ctx, y, q, sigma):
        raise NotImplementedError

    def __call__(self, y):
        """ψ(y) = h(xk + sj + y) [+ indicator]  (src/ShiftedProximalOperators.jl:51-54 and the Box / BInf methods);
        evaluated on the device, returned as a Python float (synchronises)."""
        _vec(y, "y", _n(self.xk), like=self.xk)
        self._refresh()
        out = ctypes.c_double(0.0)
        self._obj(_lib.load(), _ctx(_dev(y)), y, ctypes.byref(out))
        # Float32 ψ: the reference returns a Float32 (every term is one; libspx adds them up in Float64): rounded here
        return float(np.float32(out.value)) if self.f32 else out.value

    def _obj(self, L, ctx, y, out):
        raise TypeError("MethodError: objects of type %s are not callable" % type(self).__name__)

    def _iprox(self, L, ctx, y, g, d, check):
        raise TypeError("MethodError: no method matching iprox!(::%s, ...)" % type(self).__name__)


class _Unboxed(ShiftedProximableFunction):
    _fn = None
    _ifn = None

    def _prox(self, L, ctx, y, q, sigma):
        fn = self._sym(L, self._fn)
        _lib.check(fn(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.lam, sigma))

    def _iprox(self, L, ctx, y, g, d, check):
        if self._ifn is None:
            return super()._iprox(L, ctx, y, g, d, check)
        st = self._sym(L, self._ifn)(ctx, _ptr(y), _ptr(g), _ptr(d), _ptr(self.xk), _ptr(self.sj), _n(y),
                                   self.h.lam, 1 if check else 0)
        if st == 6:  # SPX_ERR_ASSERT: the reference's `@assert d[i] > 0`
            raise AssertionError("d[i] > 0")
        _lib.check(st)


def _unboxed_obj(self, L, ctx, y, out):
    _lib.check(self._sym(L, self._ofn)(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.lam, out))


_Unboxed._obj = _unboxed_obj


class ShiftedNormL1(_Unboxed):  # src/shiftedNormL1.jl
    _fn = "spx_prox_l1"
    _ifn = "spx_iprox_l1"
    _ofn = "spx_obj_l1"


class ShiftedNormL0(_Unboxed):  # src/shiftedNormL0.jl
    _fn = "spx_prox_l0"
    _ifn = "spx_iprox_l0"
    _ofn = "spx_obj_l0"


class ShiftedRootNormLhalf(_Unboxed):  # src/shiftedRootNormLhalf.jl
    _fn = "spx_prox_lhalf"
    _ofn = "spx_obj_lhalf"


_UNSET = object()


class _Boxed(ShiftedProximableFunction):
    _fn = None
    _check_bounds = True  # L0Box / L1Box constructors error on any(l .> u); RootNormLhalfBox does not

    def __init__(self, h, xk, sj, l, u, shifted_twice, selected, _mask=_UNSET):
        super().__init__(h, xk, sj, shifted_twice)
        n = _n(xk)
        self.l = l if _is_real(l) else _vec(l, "l", n, like=xk)
        self.u = u if _is_real(u) else _vec(u, "u", n, like=xk)
        self.selected = selected
        self._mask = _build_mask(selected, xk) if _mask is _UNSET else _mask  # shared by a second shift
        if self._check_bounds:
            lv = None if _is_real(self.l) else self.l
            uv = None if _is_real(self.u) else self.u
            if self.host:  # constructor validation on the caller's host arrays: `any(l .> u)`
                bad = bool(np.any(np.asarray(self.l) > np.asarray(self.u)))
            elif self.f32:  # (spx_check_bounds reads Float64 vectors)
                lt = self.l if lv is not None else torch.tensor(float(self.l), dtype=torch.float32, device=xk.device)
                ut = self.u if uv is not None else torch.tensor(float(self.u), dtype=torch.float32, device=xk.device)
                bad = bool((lt > ut).any())
            else:
                flag = ctypes.c_int(0)
                L = _lib.load()
                _lib.check(L.spx_check_bounds(_ctx(xk.device), _ptr(lv), _ptr(uv),
                                              float(self.l) if lv is None else 0.0,
                                              float(self.u) if uv is None else 0.0, n, ctypes.byref(flag)))
                bad = bool(flag.value)
            if bad:
                raise ValueError("Error: at least one lower bound is greater than the upper bound.")

    def _prox(self, L, ctx, y, q, sigma):
        n = _n(y)
        lv = None if _is_real(self.l) else _vec(self.l, "l", n, like=y)
        uv = None if _is_real(self.u) else _vec(self.u, "u", n, like=y)
        fn = self._sym(L, self._fn)
        _lib.check(fn(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), n, self.h.lam, sigma,
                      _ptr(lv), _ptr(uv), float(self.l) if lv is None else 0.0,
                      float(self.u) if uv is None else 0.0,
                      _ptr(self._mask[0]) if self._mask is not None else ctypes.c_void_p(0)))


class ShiftedNormL1Box(_Boxed):  # src/shiftedNormL1Box.jl
    _fn = "spx_prox_l1_box"
    _ifn = "spx_iprox_l1_box"
    _ofn = "spx_obj_l1_box"


class ShiftedNormL0Box(_Boxed):  # src/shiftedNormL0Box.jl
    _fn = "spx_prox_l0_box"
    _ifn = "spx_iprox_l0_box"
    _ofn = "spx_obj_l0_box"


def _boxed_iprox(self, L, ctx, y, g, d, check):
    if self._ifn is None:
        return ShiftedProximableFunction._iprox(self, L, ctx, y, g, d, check)
    n = _n(y)
    lv = None if _is_real(self.l) else _vec(self.l, "l", n, like=y)
    uv = None if _is_real(self.u) else _vec(self.u, "u", n, like=y)
    _lib.check(self._sym(L, self._ifn)(ctx, _ptr(y), _ptr(g), _ptr(d), _ptr(self.xk), _ptr(self.sj), n, self.h.lam,
                                     _ptr(lv), _ptr(uv), float(self.l) if lv is None else 0.0,
                                     float(self.u) if uv is None else 0.0,
                                     _ptr(self._mask[0]) if self._mask is not None else ctypes.c_void_p(0)))


def _boxed_obj(self, L, ctx, y, out):
    n = _n(y)
    lv = None if _is_real(self.l) else _vec(self.l, "l", n, like=y)
    uv = None if _is_real(self.u) else _vec(self.u, "u", n, like=y)
    _lib.check(self._sym(L, self._ofn)(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), n, self.h.lam, _ptr(lv), _ptr(uv),
                                     float(self.l) if lv is None else 0.0, float(self.u) if uv is None else 0.0,
                                     _ptr(self._mask[0]) if self._mask is not None else ctypes.c_void_p(0), out))


_Boxed._ifn = None
_Boxed._iprox = _boxed_iprox
_Boxed._obj = _boxed_obj


class ShiftedRootNormLhalfBox(_Boxed):  # src/shiftedRootNormLhalfBox.jl (no l > u check, :22-44)
    _fn = "spx_prox_lhalf_box"
    _ofn = "spx_obj_lhalf_box"
    _check_bounds = False


def _build_mask(selected, xk):
    """`selected` (0-based indices: range / list / tensor; any order, duplicates allowed) -> device byte
    mask, built once by libspx.  None or a full range(0, n) -> no mask (every index selected)."""
    n = _n(xk)
    if selected is None:
        return None
    if isinstance(selected, range) and selected.step == 1 and selected.start <= 0 and selected.stop >= n:
        return None
    if _is_host(xk):  # host ψ: the byte mask is host data too (index bookkeeping, no prox arithmetic)
        idx = np.asarray(selected.cpu() if isinstance(selected, torch.Tensor) else list(selected), dtype=np.int64)
        mask = np.zeros(n, dtype=np.uint8)
        mask[idx[(idx >= 0) & (idx < n)]] = 1
        return (mask, None)
    if isinstance(selected, torch.Tensor):
        idx = selected.to(device=xk.device, dtype=torch.int64).contiguous()
    else:
        idx = torch.as_tensor(list(selected), dtype=torch.int64).to(xk.device)
    mask = torch.empty(n, dtype=torch.uint8, device=xk.device)
    L = _lib.load()
    _lib.check(L.spx_build_mask(_ctx(xk.device), _ptr(mask), n, _ptr(idx), idx.numel()))
    return (mask, idx)  # keep idx alive until the scatter has run on the stream


class ShiftedNormL1B2(ShiftedProximableFunction):  # src/shiftedNormL1B2.jl
    def __init__(self, h, xk, sj, Δ, χ, shifted_twice):
        super().__init__(h, xk, sj, shifted_twice)
        self.Δ = float(Δ)
        self.χ = χ

    def _prox(self, L, ctx, y, q, sigma):
        _lib.check(self._sym(L, "spx_prox_l1_b2")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.lam, sigma,
                                    self.Δ, self.χ.lam))

    def _obj(self, L, ctx, y, out):  # :32  h(xk + sj + y) + IndBallL2(Δ)(sj + y)
        _lib.check(self._sym(L, "spx_obj_l1_b2")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.lam, self.Δ, out))


class _TopR(ShiftedProximableFunction):
    pass


class ShiftedIndBallL0(_TopR):  # src/shiftedIndBallL0.jl
    def _prox(self, L, ctx, y, q, sigma):
        _lib.check(self._sym(L, "spx_prox_indball_l0")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.r))

    def _obj(self, L, ctx, y, out):
        _lib.check(self._sym(L, "spx_obj_indball_l0")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.r, out))


class ShiftedIndBallL0BInf(_TopR):  # src/shiftedIndBallL0BInf.jl
    def __init__(self, h, xk, sj, Δ, χ, shifted_twice):
        super().__init__(h, xk, sj, shifted_twice)
        self.Δ = float(Δ)
        self.χ = χ

    def _prox(self, L, ctx, y, q, sigma):
        _lib.check(self._sym(L, "spx_prox_indball_l0_binf")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y),
                                              self.h.r, self.Δ))

    def _obj(self, L, ctx, y, out):
        _lib.check(self._sym(L, "spx_obj_indball_l0_binf")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y), self.h.r, self.Δ, out))


class _GroupLayout:
    """Device-side description of GroupNormL2.idx / .lambda for a vector of length n."""

    def __init__(self, h, n, device, dtype=None):
        from .functions import UniformGroups
        self._dtype = dtype if dtype is not None else torch.float64
        if isinstance(h.idx, UniformGroups):
            if h.idx.size * h.idx.count != n:
                raise IndexError("BoundsError: %d groups of %d do not tile a vector of length %d" % (h.idx.count, h.idx.size, n))
            self.ngroups, self.offsets, self.group_size, self.index = h.idx.count, None, h.idx.size, None
            self.lam = self._lam(h.lam, device, self._dtype)
            return
        self.index = None  # gather mode: (ptr, index) instead of offsets
        from .functions import RaggedGroups
        if isinstance(h.idx, RaggedGroups):
            off = h.idx.offsets
            if off[-1] > n:
                raise IndexError("BoundsError: group offsets reach %d, vector length %d" % (int(off[-1]), n))
            self.ngroups = len(h.idx)
            sizes = np.diff(off)
            if self.ngroups and n > 0 and off[0] == 0 and off[-1] == n and np.all(sizes == sizes[0]):
                self.offsets, self.group_size = None, int(sizes[0])
            else:
                self.offsets = off if device is None else torch.from_numpy(off).to(device)
                self.group_size = int(sizes.max()) if self.ngroups else 0   # size bound (hint for the tile kernels)
            self.lam = self._lam(h.lam, device, self._dtype)
            return
        bounds, sets, contiguous = [], [], True
        for g in h.idx:
            if isinstance(g, slice):
                a, b, st = g.indices(n)
                g = range(a, b, st)
            if isinstance(g, range):
                a, b, st = g.start, g.stop, g.step
                if st == 1 and a <= b:
                    if a < 0 or b > n:
                        raise IndexError("BoundsError: group %s outside 0:%d" % (g, n))
                    bounds.append((a, b))
                    sets.append(g)
                    continue
                idx = list(g)
            else:
                idx = [int(v) for v in (g.tolist() if hasattr(g, "tolist") else g)]
            # an explicit index vector (the reference's `collect(4:6)`, runtests.jl:290): a contiguous run is a range
            if idx and all(j - i == 1 for i, j in zip(idx, idx[1:])):
                if idx[0] < 0 or idx[-1] >= n:
                    raise IndexError("BoundsError: group index outside 0:%d" % n)
                bounds.append((idx[0], idx[-1] + 1))
            else:
                contiguous = False
            sets.append(idx)
        if contiguous and any(a1 != b0 for (a0, b0), (a1, b1) in zip(bounds, bounds[1:])):
            contiguous = False  # gaps / overlaps / out of order: general index sets
        self.ngroups = len(sets)
        if not contiguous:
            # gather form: literal reference semantics for arbitrary (overlapping, partial) index sets
            ptr = np.zeros(self.ngroups + 1, dtype=np.int64)
            for k, g in enumerate(sets):
                ptr[k + 1] = ptr[k] + len(g)
            index = (np.concatenate([np.asarray(g, dtype=np.int64) for g in sets]) if self.ngroups
                     else np.zeros(0, dtype=np.int64))
            if index.size and (index.min() < 0 or index.max() >= n):
                raise IndexError("BoundsError: group index outside 0:%d" % n)
            self.nnz = int(index.size)
            self.offsets = ptr if device is None else torch.from_numpy(ptr).to(device)
            self.index = index if device is None else torch.from_numpy(index).to(device)
            self.group_size = 0
            self.lam = self._lam(h.lam, device, self._dtype)
            return
        sizes = {b - a for a, b in bounds}
        if self.ngroups and len(sizes) == 1 and bounds[0][0] == 0 and bounds[-1][1] == n and n > 0:
            self.offsets = None
            self.group_size = sizes.pop()
        else:
            off = [bounds[0][0]] + [b for _, b in bounds] if bounds else [0]
            self.offsets = (np.asarray(off, dtype=np.int64) if device is None
                            else torch.tensor(off, dtype=torch.int64, device=device))
            # with offsets, group_size is an upper bound on the group sizes: it lets libspx pick its register-tile kernels
            self.group_size = max(sizes) if sizes else 0
        self.lam = self._lam(h.lam, device, self._dtype)

    @staticmethod
    def _lam(lam, device, dtype=torch.float64):
        if device is None:  # host ψ
            return np.ascontiguousarray(lam.cpu().numpy() if isinstance(lam, torch.Tensor) else lam, dtype=np.float64)
        if isinstance(lam, torch.Tensor):
            return lam.to(device=device, dtype=dtype).contiguous()
        return torch.tensor(lam, dtype=dtype, device=device)


class ShiftedGroupNormL2(ShiftedProximableFunction):  # src/shiftedGroupNormL2.jl
    def __init__(self, h, xk, sj, shifted_twice, _layout=None):
        super().__init__(h, xk, sj, shifted_twice)
        self._layout = _layout or _GroupLayout(h, _n(xk), _dev(xk), None if _is_host(xk) else xk.dtype)

    def _prox(self, L, ctx, y, q, sigma):
        g = self._layout
        if g.index is not None:
            _lib.check(self._sym(L, "spx_prox_group_l2_gather")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y),
                                                                _iptr(g.offsets), _iptr(g.index), g.ngroups, g.nnz,
                                                                _ptr(g.lam), sigma))
            return
        _lib.check(self._sym(L, "spx_prox_group_l2")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y),
                                       _ptr(g.offsets), g.group_size, g.ngroups, _ptr(g.lam), sigma))

    def _obj(self, L, ctx, y, out):
        g = self._layout
        if g.index is not None:
            _lib.check(self._sym(L, "spx_obj_group_l2_gather")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y),
                                                               _iptr(g.offsets), _iptr(g.index), g.ngroups, g.nnz,
                                                               _ptr(g.lam), out))
            return
        _lib.check(self._sym(L, "spx_obj_group_l2")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y), _ptr(g.offsets),
                                      g.group_size, g.ngroups, _ptr(g.lam), out))


class ShiftedGroupNormL2Binf(ShiftedProximableFunction):  # src/shiftedGroupNormL2Binf.jl
    def __init__(self, h, xk, sj, Δ, χ, shifted_twice, _layout=None):
        super().__init__(h, xk, sj, shifted_twice)
        self.Δ = float(Δ)
        self.χ = χ
        self._layout = _layout or _GroupLayout(h, _n(xk), _dev(xk), None if _is_host(xk) else xk.dtype)

    def _prox(self, L, ctx, y, q, sigma):
        g = self._layout
        if g.index is not None:
            _lib.check(self._sym(L, "spx_prox_group_l2_binf_gather")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj),
                                                                     _n(y), _iptr(g.offsets), _iptr(g.index), g.ngroups,
                                                                     g.nnz, _ptr(g.lam), sigma, self.Δ))
            return
        _lib.check(self._sym(L, "spx_prox_group_l2_binf")(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), _n(y),
                                            _ptr(g.offsets), g.group_size, g.ngroups, _ptr(g.lam), sigma, self.Δ))

    def _obj(self, L, ctx, y, out):
        g = self._layout
        if g.index is not None:
            _lib.check(self._sym(L, "spx_obj_group_l2_binf_gather")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y),
                                                                    _iptr(g.offsets), _iptr(g.index), g.ngroups, g.nnz,
                                                                    _ptr(g.lam), self.Δ, out))
            return
        _lib.check(self._sym(L, "spx_obj_group_l2_binf")(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), _n(y), _ptr(g.offsets),
                                           g.group_size, g.ngroups, _ptr(g.lam), self.Δ, out))


# ---------------------------------------------------------------------------------------------
# shifted(...)                                            constructors cited in SURVEY.md §3.1
# ---------------------------------------------------------------------------------------------
_BOX = {NormL1: ShiftedNormL1Box, NormL0: ShiftedNormL0Box, RootNormLhalf: ShiftedRootNormLhalfBox}
_PLAIN = {NormL1: ShiftedNormL1, NormL0: ShiftedNormL0, RootNormLhalf: ShiftedRootNormLhalf}


def shifted(h, x, *args):
    # second shift: shifted(ψ, sj)
    if isinstance(h, ShiftedProximableFunction):
        if args:
            raise TypeError("MethodError: shifted(ψ, sj) takes no trust-region arguments")
        ψ, sj = h, _vec(x, "sj", _n(h.xk), like=h.xk)
        if isinstance(ψ, _Boxed):
            ω = type(ψ)(ψ.h, ψ.xk, sj, ψ.l, ψ.u, True, ψ.selected, _mask=ψ._mask)
        elif isinstance(ψ, ShiftedIndBallL0BInf):
            ω = ShiftedIndBallL0BInf(ψ.h, ψ.xk, sj, ψ.Δ, ψ.χ, True)
        elif isinstance(ψ, ShiftedNormL1B2):
            ω = ShiftedNormL1B2(ψ.h, ψ.xk, sj, ψ.Δ, ψ.χ, True)
        elif isinstance(ψ, ShiftedGroupNormL2Binf):
            ω = ShiftedGroupNormL2Binf(ψ.h, ψ.xk, sj, ψ.Δ, ψ.χ, True, _layout=ψ._layout)
        elif isinstance(ψ, ShiftedGroupNormL2):
            ω = ShiftedGroupNormL2(ψ.h, ψ.xk, sj, True, _layout=ψ._layout)
        else:
            ω = type(ψ)(ψ.h, ψ.xk, sj, True)
        ω._xk_ref = ψ._xk_ref   # (the packed copy of a strided xk is shared like xk itself)
        return ω

    x, ref = _unstrided(x, "xk")
    if ref is not None:
        ψ = shifted(h, x, *args)
        ψ._xk_ref = ref
        return ψ
    xk = _vec(x, "xk")
    zero = lambda: _zeros_like(xk)  # `zero(xk)`
    if isinstance(h, NormL2):  # shiftedGroupNormL2.jl:34-35, shiftedGroupNormL2Binf.jl:48-49
        h = GroupNormL2([h.lam])
    if len(args) == 0:
        if type(h) in _PLAIN:
            return _PLAIN[type(h)](h, xk, zero(), False)
        if isinstance(h, IndBallL0):
            return ShiftedIndBallL0(h, xk, zero(), False)
        if isinstance(h, GroupNormL2):
            return ShiftedGroupNormL2(h, xk, zero(), False)
        raise TypeError("MethodError: no accelerated shifted() for %s" % type(h).__name__)
    if len(args) == 2 and isinstance(args[1], NormL2) and isinstance(h, NormL1):  # shiftedNormL1B2.jl:35-36
        return ShiftedNormL1B2(h, xk, zero(), float(args[0]), args[1], False)
    if len(args) in (2, 3) and isinstance(args[1], NormLinf):  # shifted(h, xk, Δ, χ[, selected])
        Δ, χ = float(args[0]), args[1]
        selected = args[2] if len(args) == 3 else None
        if type(h) in _BOX:
            return _BOX[type(h)](h, xk, zero(), -Δ, Δ, False, selected)
        if selected is not None:
            raise TypeError("MethodError: `selected` is only supported by the Box operators")
        if isinstance(h, IndBallL0):
            return ShiftedIndBallL0BInf(h, xk, zero(), Δ, χ, False)
        if isinstance(h, GroupNormL2):
            return ShiftedGroupNormL2Binf(h, xk, zero(), Δ, χ, False)
        raise TypeError("MethodError: no accelerated shifted(h, x, Δ, χ) for %s" % type(h).__name__)
    if len(args) in (2, 3) and type(h) in _BOX:  # shifted(h, xk, l, u[, selected])
        l, u = args[0], args[1]
        selected = args[2] if len(args) == 3 else None
        return _BOX[type(h)](h, xk, zero(), l, u, False, selected)
    raise TypeError("MethodError: no method matching shifted(%s, x, %s)" %
                    (type(h).__name__, ", ".join(type(a).__name__ for a in args)))


# ---------------------------------------------------------------------------------------------
# prox!, prox, shift!, set_radius!, set_bounds!
# ---------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# the unshifted value types: h(x) and prox!(y, h, x, γ)
#   RootNormLhalf  src/rootNormLhalf.jl:27-51,  GroupNormL2  src/groupNormL2.jl:33-58 (defined by the reference itself);
#   NormL0 / NormL1 / NormL2 / IndBallL0 evaluate like their ProximalOperators.jl namesakes [ext].
# Run as the shifted operators with xk = sj = 0 (the same kernels: q + (0 + 0) = q and y - (0 + 0) = y are exact).
# ---------------------------------------------------------------------------------------------
_zero_cache = {}


def _zeros_for(x):
    """a zero vector of x's kind / length / device, cached (read-only use)"""
    key = ("host", _n(x)) if _is_host(x) else (str(x.device), _n(x), x.dtype)
    z = _zero_cache.get(key)
    if z is None:
        if len(_zero_cache) > 8:
            _zero_cache.clear()
        z = _zero_cache[key] = _zeros_like(x)
    return z


def _as_shifted(h, x):
    z = _zeros_for(_vec(x, "x"))
    if isinstance(h, NormL2):
        h = GroupNormL2([h.lam])
    if type(h) in _PLAIN:
        ψ = _PLAIN[type(h)](h, z, z, False)
    elif isinstance(h, IndBallL0):
        ψ = ShiftedIndBallL0(h, z, z, False)
    elif isinstance(h, GroupNormL2):
        ψ = ShiftedGroupNormL2(h, z, z, False)
    else:
        raise TypeError("MethodError: %s has no accelerated evaluation" % type(h).__name__)
    return ψ


def value(h, x):
    """h(x) for an unshifted value type, evaluated on the GPU (device tensor or host array x); a Python float."""
    x, _ = _unstrided(x, "x")   # (a strided view is evaluated through a packed copy: h only reads x)
    return _as_shifted(h, x)(x)


def _unshifted_prox_bang(y, h, x, γ):
    """prox!(y, h, x, γ) of the value types; returns what the reference's method returns: RootNormLhalf -> h(y)
    (src/rootNormLhalf.jl:50), GroupNormL2 -> Σ λ_g ‖x_g‖ over the groups of the INPUT with a nonzero norm
    (src/groupNormL2.jl:48-57), others -> h(y)."""
    ψ = _as_shifted(h, x)
    n = _n(x)
    _vec(y, "y", n, like=x)
    if isinstance(ψ, _Unboxed) and not ψ.host:
        return prox_value_bang(y, ψ, x, γ)[1]           # one pass: y and h(y)
    if isinstance(h, (GroupNormL2, NormL2)):
        ysum = ψ(x)                                       # Σ λ_g ‖x_g‖ of the input (zero groups add nothing)
        ψ._prox(_lib.load(), _ctx(_dev(y)), y, x, float(γ))
        return ysum
    ψ._prox(_lib.load(), _ctx(_dev(y)), y, x, float(γ))
    return ψ(y)


def prox_bang(y, ψ, q, σ=1.0):
    """prox!(y, ψ, q, σ): y <- argmin_t ½σ⁻¹‖t − q‖² + ψ(t); returns y.  Asynchronous on the current
    torch stream.  y may be q itself or ψ.sol.
    With an unshifted value type h in place of ψ this is the reference's prox!(y, h, x, γ) (src/rootNormLhalf.jl:31-51,
    src/groupNormL2.jl:41-58) and returns the value that method returns."""
    if isinstance(ψ, ProximableFunction):
        return _unshifted_prox_bang(y, ψ, q, σ)
    if not isinstance(ψ, ShiftedProximableFunction):
        raise TypeError("ψ must be a ShiftedProximableFunction")
    n = _n(ψ.xk)
    _vec(q, "q", n, like=ψ.xk)
    _vec(y, "y", n, like=ψ.xk)
    ψ._refresh()
    ψ._prox(_lib.load(), _ctx(_dev(y)), y, q, float(σ))
    return y


def prox(ψ, q, σ):
    """prox(ψ, q, σ) = prox!(ψ.sol, ψ, q, σ)   (src/ShiftedProximalOperators.jl:189-190)"""
    return prox_bang(ψ.sol, ψ, q, σ)


def prox_value_bang(y, ψ, q, σ, q_scale=1.0):
    """prox!(y, ψ, q_scale .* q, σ) and h(xk + sj + y) of the result in ONE pass over the vectors (the pair a solver iteration makes:
    R2's `prox!(s, ψ, …)` followed by `ψ(s)`): returns (y, value).  Device vectors, the separable operators
    (ShiftedNormL1 / NormL0 / RootNormLhalf and their Box forms); synchronises to return the value.  The Box forms
    return the h part of ψ(y): the prox lies inside the box by construction.  q_scale: the prox is taken at q_scale * q,
    formed on the fly (R2: `prox_value(ψ, ∇f, ν, q_scale=-ν)` instead of materialising -ν∇f)."""
    if not isinstance(ψ, (_Unboxed, _Boxed)) or ψ.host:
        raise TypeError("prox_value is available for the separable operators on device vectors")
    n = _n(ψ.xk)
    _vec(q, "q", n, like=ψ.xk)
    _vec(y, "y", n, like=ψ.xk)
    ψ._refresh()
    L, ctx = _lib.load(), _ctx(_dev(y))
    out = ctypes.c_double(0.0)
    fn = getattr(L, ψ._fn.replace("spx_prox_", "spx_proxval_"))
    if isinstance(ψ, _Boxed):
        lv = None if _is_real(ψ.l) else _vec(ψ.l, "l", n, like=y)
        uv = None if _is_real(ψ.u) else _vec(ψ.u, "u", n, like=y)
        _lib.check(fn(ctx, _ptr(y), _ptr(q), _ptr(ψ.xk), _ptr(ψ.sj), n, ψ.h.lam, float(σ), _ptr(lv), _ptr(uv),
                      float(ψ.l) if lv is None else 0.0, float(ψ.u) if uv is None else 0.0,
                      _ptr(ψ._mask[0]) if ψ._mask is not None else ctypes.c_void_p(0), float(q_scale), ctypes.byref(out)))
    else:
        _lib.check(fn(ctx, _ptr(y), _ptr(q), _ptr(ψ.xk), _ptr(ψ.sj), n, ψ.h.lam, float(σ), float(q_scale),
                      ctypes.byref(out)))
    return y, out.value


def prox_value(ψ, q, σ, q_scale=1.0):
    """(prox(ψ, q_scale .* q, σ), h(xk + sj + prox)) in one pass; see prox_value_bang"""
    return prox_value_bang(ψ.sol, ψ, q, σ, q_scale)


def iprox_bang(y, ψ, g, d, check=True):
    """iprox!(y, ψ, g, d): y <- argmin_t ½ tᵀDt + gᵀt + ψ(t), D = diag(d); returns y.  Defined for ShiftedNormL1/L0 and
    their Box forms (as in the reference).  The unboxed forms assert d .> 0 like the reference (`check=True`
    synchronises to raise AssertionError; pass check=False to stay asynchronous)."""
    if not isinstance(ψ, ShiftedProximableFunction):
        raise TypeError("ψ must be a ShiftedProximableFunction")
    n = _n(ψ.xk)
    _vec(g, "g", n, like=ψ.xk)
    _vec(d, "d", n, like=ψ.xk)
    _vec(y, "y", n, like=ψ.xk)
    ψ._refresh()
    ψ._iprox(_lib.load(), _ctx(_dev(y)), y, g, d, check)
    return y


def iprox(ψ, g, d):
    """iprox(ψ, g, d) = iprox!(ψ.sol, ψ, g, d)   (src/ShiftedProximalOperators.jl:180)"""
    return iprox_bang(ψ.sol, ψ, g, d)


def shift_bang(ψ, shift):
    """shift!(ψ, v): in-place copy into ψ.sj (twice shifted) or ψ.xk -- i.e. into the caller's tensor
    (src/ShiftedProximalOperators.jl:72-79)."""
    _vec(shift, "shift", _n(ψ.xk), like=ψ.xk)
    _copy_into(ψ.sj if ψ.shifted_twice else ψ.xk, shift)
    if not ψ.shifted_twice and ψ._xk_ref is not None:
        ψ._xk_ref.write_through()   # ψ.xk IS the caller's strided view in the reference: `ψ.xk .= shift` lands there
    return ψ


def set_bounds_bang(ψ, l, u):
    """set_bounds!(ψ, l, u)   (src/ShiftedProximalOperators.jl:107-111): a scalar replaces; a vector
    replaces a stored scalar, otherwise it is copied into the stored vector."""
    if not isinstance(ψ, _Boxed):
        raise AttributeError("type %s has no field l" % type(ψ).__name__)
    n = _n(ψ.xk)
    for name, new in (("l", l), ("u", u)):
        cur = getattr(ψ, name)
        if _is_real(new):
            setattr(ψ, name, new)
        elif _is_real(cur):
            setattr(ψ, name, _vec(new, name, n, like=ψ.xk))
        else:
            _copy_into(cur, _vec(new, name, n, like=ψ.xk))
    return ψ


def set_radius_bang(ψ, Δ):
    """set_radius!(ψ, Δ)   (src/ShiftedProximalOperators.jl:93-99)"""
    if isinstance(ψ, _Boxed):
        return set_bounds_bang(ψ, -Δ, Δ)
    if not hasattr(ψ, "Δ"):
        raise AttributeError("type %s has no field Δ" % type(ψ).__name__)
    ψ.Δ = float(Δ)
    return ψ


def context(device="cuda"):
    """The libspx context (opaque handle) bound to `device`'s current torch stream -- for callers that
    drive the C ABI directly (bench.py's HIP-event stopwatch)."""
    d = torch.device(device)
    if d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return _ctx(d)


class device_values:
    """`with device_values(out):` -- inside the block psi(y), prox_value(...) and prox_value_bang(...) on `out`'s device store
    their value in out[0] (a 1-element float64 device tensor) and return NaN: nothing is read back and the stream is not
    synchronised (spx_ctx_set_value_target).  The accept / reject logic of a solver can then run on the device, or fetch
    out.item() when it needs the number."""

    def __init__(self, out):
        if not (isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float64 and out.numel() >= 1):
            raise TypeError("device_values needs a float64 device tensor with at least one element")
        self.out = out
        self.ctx = None

    def __enter__(self):
        self.ctx = _ctx(self.out.device)
        _lib.check(_lib.load().spx_ctx_set_value_target(self.ctx, ctypes.c_void_p(self.out.data_ptr())))
        return self.out

    def __exit__(self, *exc):
        _lib.check(_lib.load().spx_ctx_set_value_target(self.ctx, None))
        return False


def synchronize(device=None):
    """Wait for the libspx contexts' work (same as torch.cuda.synchronize for borrowed streams)."""
    L = _lib.load()
    for c in list(_ctxs.values()):
        _lib.check(L.spx_sync(c))


# h(x) on the value types themselves: `h(x)` as in the reference (`(f::RootNormLhalf)(x)`, `(f::GroupNormL2)(x)`, ...)
for _T in (NormL0, NormL1, NormL2, RootNormLhalf, IndBallL0, GroupNormL2):
    _T.__call__ = lambda self, x: value(self, x)
