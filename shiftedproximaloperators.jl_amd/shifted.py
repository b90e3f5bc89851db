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
no CPU path: host tensors raise TypeError.  Index sets (`selected`, group ranges) are 0-based here.
"""
import ctypes
import numbers

import torch

from . import _lib
from .functions import (GroupNormL2, IndBallL0, NormL0, NormL1, NormL2, NormLinf, RootNormLhalf)

# ---------------------------------------------------------------------------------------------
# contexts: one libspx context per (device, stream) so calls are ordered with the caller's torch work
# ---------------------------------------------------------------------------------------------
_ctxs = {}


def _ctx(device):
    stream = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream.cuda_stream)
    c = _ctxs.get(key)
    if c is None:
        L = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(L.spx_ctx_create_on_stream(key[0], ctypes.c_void_p(key[1]), ctypes.byref(h)))
        c = _ctxs[key] = h
    return c


def _vec(t, name, n=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch tensor on the GPU (got %s); libspx has no host path" % (name, type(t)))
    if not t.is_cuda:
        raise TypeError("%s must live in device memory (cuda); libspx has no host path" % name)
    if t.dtype != torch.float64:
        raise TypeError("%s must be float64 (got %s)" % (name, t.dtype))
    if t.dim() != 1 or (t.numel() > 1 and t.stride(0) != 1):
        raise TypeError("%s must be a contiguous vector" % name)
    if n is not None and t.numel() != n:
        raise IndexError("BoundsError: %s has length %d, expected %d" % (name, t.numel(), n))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _is_real(v):
    return isinstance(v, numbers.Real)


# ---------------------------------------------------------------------------------------------
# base type                                      src/ShiftedProximalOperators.jl:18,113-121
# ---------------------------------------------------------------------------------------------
class ShiftedProximableFunction:
    def __init__(self, h, xk, sj, shifted_twice):
        self.h = h
        self.xk = _vec(xk, "xk")
        self.sj = _vec(sj, "sj", xk.numel())
        self.sol = torch.empty_like(xk)  # `sol = similar(xk)`
        self.shifted_twice = bool(shifted_twice)

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
        _vec(y, "y", self.xk.numel())
        out = ctypes.c_double(0.0)
        self._obj(_lib.load(), _ctx(y.device), y, ctypes.byref(out))
        return out.value

    def _obj(self, L, ctx, y, out):
        raise TypeError("MethodError: objects of type %s are not callable" % type(self).__name__)

    def _iprox(self, L, ctx, y, g, d, check):
        raise TypeError("MethodError: no method matching iprox!(::%s, ...)" % type(self).__name__)


class _Unboxed(ShiftedProximableFunction):
    _fn = None
    _ifn = None

    def _prox(self, L, ctx, y, q, sigma):
        fn = getattr(L, self._fn)
        _lib.check(fn(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), y.numel(), self.h.lam, sigma))

    def _iprox(self, L, ctx, y, g, d, check):
        if self._ifn is None:
            return super()._iprox(L, ctx, y, g, d, check)
        st = getattr(L, self._ifn)(ctx, _ptr(y), _ptr(g), _ptr(d), _ptr(self.xk), _ptr(self.sj), y.numel(),
                                   self.h.lam, 1 if check else 0)
        if st == 6:  # SPX_ERR_ASSERT: the reference's `@assert d[i] > 0`
            raise AssertionError("d[i] > 0")
        _lib.check(st)


def _unboxed_obj(self, L, ctx, y, out):
    _lib.check(getattr(L, self._ofn)(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), y.numel(), self.h.lam, out))


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
        n = xk.numel()
        self.l = l if _is_real(l) else _vec(l, "l", n)
        self.u = u if _is_real(u) else _vec(u, "u", n)
        self.selected = selected
        self._mask = _build_mask(selected, xk) if _mask is _UNSET else _mask  # shared by a second shift
        if self._check_bounds:
            lv = None if _is_real(self.l) else self.l
            uv = None if _is_real(self.u) else self.u
            flag = ctypes.c_int(0)
            L = _lib.load()
            _lib.check(L.spx_check_bounds(_ctx(xk.device), _ptr(lv), _ptr(uv),
                                          float(self.l) if lv is None else 0.0,
                                          float(self.u) if uv is None else 0.0, n, ctypes.byref(flag)))
            if flag.value:
                raise ValueError("Error: at least one lower bound is greater than the upper bound.")

    def _prox(self, L, ctx, y, q, sigma):
        n = y.numel()
        lv = None if _is_real(self.l) else _vec(self.l, "l", n)
        uv = None if _is_real(self.u) else _vec(self.u, "u", n)
        fn = getattr(L, self._fn)
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
    n = y.numel()
    lv = None if _is_real(self.l) else _vec(self.l, "l", n)
    uv = None if _is_real(self.u) else _vec(self.u, "u", n)
    _lib.check(getattr(L, self._ifn)(ctx, _ptr(y), _ptr(g), _ptr(d), _ptr(self.xk), _ptr(self.sj), n, self.h.lam,
                                     _ptr(lv), _ptr(uv), float(self.l) if lv is None else 0.0,
                                     float(self.u) if uv is None else 0.0,
                                     _ptr(self._mask[0]) if self._mask is not None else ctypes.c_void_p(0)))


def _boxed_obj(self, L, ctx, y, out):
    n = y.numel()
    lv = None if _is_real(self.l) else _vec(self.l, "l", n)
    uv = None if _is_real(self.u) else _vec(self.u, "u", n)
    _lib.check(getattr(L, self._ofn)(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), n, self.h.lam, _ptr(lv), _ptr(uv),
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
    n = xk.numel()
    if selected is None:
        return None
    if isinstance(selected, range) and selected.step == 1 and selected.start <= 0 and selected.stop >= n:
        return None
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
        _lib.check(L.spx_prox_l1_b2(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), y.numel(), self.h.lam, sigma,
                                    self.Δ, self.χ.lam))


class _TopR(ShiftedProximableFunction):
    pass


class ShiftedIndBallL0(_TopR):  # src/shiftedIndBallL0.jl
    def _prox(self, L, ctx, y, q, sigma):
        _lib.check(L.spx_prox_indball_l0(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), y.numel(), self.h.r))

    def _obj(self, L, ctx, y, out):
        _lib.check(L.spx_obj_indball_l0(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), y.numel(), self.h.r, out))


class ShiftedIndBallL0BInf(_TopR):  # src/shiftedIndBallL0BInf.jl
    def __init__(self, h, xk, sj, Δ, χ, shifted_twice):
        super().__init__(h, xk, sj, shifted_twice)
        self.Δ = float(Δ)
        self.χ = χ

    def _prox(self, L, ctx, y, q, sigma):
        _lib.check(L.spx_prox_indball_l0_binf(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), y.numel(),
                                              self.h.r, self.Δ))

    def _obj(self, L, ctx, y, out):
        _lib.check(L.spx_obj_indball_l0_binf(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), y.numel(), self.h.r, self.Δ, out))


class _GroupLayout:
    """Device-side description of GroupNormL2.idx / .lambda for a vector of length n."""

    def __init__(self, h, n, device):
        from .functions import UniformGroups
        if isinstance(h.idx, UniformGroups):
            if h.idx.size * h.idx.count != n:
                raise IndexError("BoundsError: %d groups of %d do not tile a vector of length %d" % (h.idx.count, h.idx.size, n))
            self.ngroups, self.offsets, self.group_size = h.idx.count, None, h.idx.size
            lam = h.lam
            self.lam = (lam.to(device=device, dtype=torch.float64).contiguous() if isinstance(lam, torch.Tensor)
                        else torch.tensor(lam, dtype=torch.float64, device=device))
            return
        bounds = []
        for g in h.idx:
            if isinstance(g, slice):
                a, b, st = g.indices(n)
            elif isinstance(g, range):
                a, b, st = g.start, g.stop, g.step
            else:
                # an explicit index vector (the reference's `collect(4:6)`): accepted when it is a contiguous run
                idx = [int(v) for v in (g.tolist() if hasattr(g, "tolist") else g)]
                if not idx or any(j - i != 1 for i, j in zip(idx, idx[1:])):
                    raise NotImplementedError("gather-index groups (non-contiguous index vectors) are not on the "
                                              "accelerated path; pass contiguous ranges")
                a, b, st = idx[0], idx[-1] + 1, 1
            if st != 1 or a < 0 or b > n or a > b:
                raise NotImplementedError("groups must be contiguous 0-based ranges inside 0:n")
            bounds.append((a, b))
        for (a0, b0), (a1, b1) in zip(bounds, bounds[1:]):
            if a1 != b0:
                raise NotImplementedError("groups must be consecutive ranges (CSR offsets); got a gap/overlap")
        self.ngroups = len(bounds)
        sizes = {b - a for a, b in bounds}
        if self.ngroups and len(sizes) == 1 and bounds[0][0] == 0 and bounds[-1][1] == n and n > 0:
            self.offsets = None
            self.group_size = sizes.pop()
        else:
            off = [bounds[0][0]] + [b for _, b in bounds] if bounds else [0]
            self.offsets = torch.tensor(off, dtype=torch.int64, device=device)
            self.group_size = 0
        lam = h.lam
        if isinstance(lam, torch.Tensor):
            self.lam = lam.to(device=device, dtype=torch.float64).contiguous()
        else:
            self.lam = torch.tensor(lam, dtype=torch.float64, device=device)


class ShiftedGroupNormL2(ShiftedProximableFunction):  # src/shiftedGroupNormL2.jl
    def __init__(self, h, xk, sj, shifted_twice, _layout=None):
        super().__init__(h, xk, sj, shifted_twice)
        self._layout = _layout or _GroupLayout(h, xk.numel(), xk.device)

    def _prox(self, L, ctx, y, q, sigma):
        g = self._layout
        _lib.check(L.spx_prox_group_l2(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), y.numel(),
                                       _ptr(g.offsets), g.group_size, g.ngroups, _ptr(g.lam), sigma))

    def _obj(self, L, ctx, y, out):
        g = self._layout
        _lib.check(L.spx_obj_group_l2(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), y.numel(), _ptr(g.offsets),
                                      g.group_size, g.ngroups, _ptr(g.lam), out))


class ShiftedGroupNormL2Binf(ShiftedProximableFunction):  # src/shiftedGroupNormL2Binf.jl
    def __init__(self, h, xk, sj, Δ, χ, shifted_twice, _layout=None):
        super().__init__(h, xk, sj, shifted_twice)
        self.Δ = float(Δ)
        self.χ = χ
        self._layout = _layout or _GroupLayout(h, xk.numel(), xk.device)

    def _prox(self, L, ctx, y, q, sigma):
        g = self._layout
        _lib.check(L.spx_prox_group_l2_binf(ctx, _ptr(y), _ptr(q), _ptr(self.xk), _ptr(self.sj), y.numel(),
                                            _ptr(g.offsets), g.group_size, g.ngroups, _ptr(g.lam), sigma, self.Δ))

    def _obj(self, L, ctx, y, out):
        g = self._layout
        _lib.check(L.spx_obj_group_l2_binf(ctx, _ptr(y), _ptr(self.xk), _ptr(self.sj), y.numel(), _ptr(g.offsets),
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
        ψ, sj = h, _vec(x, "sj", h.xk.numel())
        if isinstance(ψ, _Boxed):
            return type(ψ)(ψ.h, ψ.xk, sj, ψ.l, ψ.u, True, ψ.selected, _mask=ψ._mask)
        if isinstance(ψ, ShiftedIndBallL0BInf):
            return ShiftedIndBallL0BInf(ψ.h, ψ.xk, sj, ψ.Δ, ψ.χ, True)
        if isinstance(ψ, ShiftedNormL1B2):
            return ShiftedNormL1B2(ψ.h, ψ.xk, sj, ψ.Δ, ψ.χ, True)
        if isinstance(ψ, ShiftedGroupNormL2Binf):
            return ShiftedGroupNormL2Binf(ψ.h, ψ.xk, sj, ψ.Δ, ψ.χ, True, _layout=ψ._layout)
        if isinstance(ψ, ShiftedGroupNormL2):
            return ShiftedGroupNormL2(ψ.h, ψ.xk, sj, True, _layout=ψ._layout)
        return type(ψ)(ψ.h, ψ.xk, sj, True)

    xk = _vec(x, "xk")
    zero = lambda: torch.zeros_like(xk)  # `zero(xk)`
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
def prox_bang(y, ψ, q, σ):
    """prox!(y, ψ, q, σ): y <- argmin_t ½σ⁻¹‖t − q‖² + ψ(t); returns y.  Asynchronous on the current
    torch stream.  y may be q itself or ψ.sol."""
    if not isinstance(ψ, ShiftedProximableFunction):
        raise TypeError("ψ must be a ShiftedProximableFunction")
    n = ψ.xk.numel()
    _vec(q, "q", n)
    _vec(y, "y", n)
    if q.device != ψ.xk.device or y.device != ψ.xk.device:
        raise TypeError("y, q and ψ.xk must live on the same device")
    ψ._prox(_lib.load(), _ctx(y.device), y, q, float(σ))
    return y


def prox(ψ, q, σ):
    """prox(ψ, q, σ) = prox!(ψ.sol, ψ, q, σ)   (src/ShiftedProximalOperators.jl:189-190)"""
    return prox_bang(ψ.sol, ψ, q, σ)


def iprox_bang(y, ψ, g, d, check=True):
    """iprox!(y, ψ, g, d): y <- argmin_t ½ tᵀDt + gᵀt + ψ(t), D = diag(d); returns y.  Defined for ShiftedNormL1/L0 and
    their Box forms (as in the reference).  The unboxed forms assert d .> 0 like the reference (`check=True`
    synchronises to raise AssertionError; pass check=False to stay asynchronous)."""
    if not isinstance(ψ, ShiftedProximableFunction):
        raise TypeError("ψ must be a ShiftedProximableFunction")
    n = ψ.xk.numel()
    _vec(g, "g", n)
    _vec(d, "d", n)
    _vec(y, "y", n)
    ψ._iprox(_lib.load(), _ctx(y.device), y, g, d, check)
    return y


def iprox(ψ, g, d):
    """iprox(ψ, g, d) = iprox!(ψ.sol, ψ, g, d)   (src/ShiftedProximalOperators.jl:180)"""
    return iprox_bang(ψ.sol, ψ, g, d)


def shift_bang(ψ, shift):
    """shift!(ψ, v): in-place copy into ψ.sj (twice shifted) or ψ.xk -- i.e. into the caller's tensor
    (src/ShiftedProximalOperators.jl:72-79)."""
    _vec(shift, "shift", ψ.xk.numel())
    (ψ.sj if ψ.shifted_twice else ψ.xk).copy_(shift)
    return ψ


def set_bounds_bang(ψ, l, u):
    """set_bounds!(ψ, l, u)   (src/ShiftedProximalOperators.jl:107-111): a scalar replaces; a vector
    replaces a stored scalar, otherwise it is copied into the stored vector."""
    if not isinstance(ψ, _Boxed):
        raise AttributeError("type %s has no field l" % type(ψ).__name__)
    n = ψ.xk.numel()
    for name, new in (("l", l), ("u", u)):
        cur = getattr(ψ, name)
        if _is_real(new):
            setattr(ψ, name, new)
        elif _is_real(cur):
            setattr(ψ, name, _vec(new, name, n))
        else:
            cur.copy_(_vec(new, name, n))
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


def synchronize(device=None):
    """Wait for the libspx contexts' work (same as torch.cuda.synchronize for borrowed streams)."""
    L = _lib.load()
    for c in list(_ctxs.values()):
        _lib.check(L.spx_sync(c))
