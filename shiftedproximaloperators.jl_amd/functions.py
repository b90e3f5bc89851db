"""Value types of the proximable functions h that the shifted operators wrap.

NormL0 / NormL1 / NormL2 / IndBallL0 / NormLinf come from ProximalOperators.jl in the reference
(Project.toml:6-12; only their fields `lambda` / `r` are used on the prox! path); RootNormLhalf and
GroupNormL2 are defined by the reference itself (src/rootNormLhalf.jl:14-25, src/groupNormL2.jl:15-31).
Constructor checks and messages follow those files.
"""


class ProximableFunction:
    pass


class NormL0(ProximableFunction):
    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        self.lam = float(lam)

    lambda_ = property(lambda self: self.lam)


class NormL1(ProximableFunction):
    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        self.lam = float(lam)

    lambda_ = property(lambda self: self.lam)


class NormL2(ProximableFunction):
    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")
        self.lam = float(lam)

    lambda_ = property(lambda self: self.lam)


class NormLinf(ProximableFunction):
    """Stands for `Conjugate{IndBallL1}` = NormLinf(1.0), the trust-region norm argument χ of
    shifted(h, x, Δ, χ) (src/shiftedNormL1Box.jl:57-63)."""

    def __init__(self, lam=1.0):
        self.lam = float(lam)

    def __call__(self, y):
        return self.lam * float(abs(y).max()) if len(y) else 0.0


class RootNormLhalf(ProximableFunction):
    """h(x) = λ Σ sqrt|x_i|   (src/rootNormLhalf.jl:14-29)"""

    def __init__(self, lam=1.0):
        if lam < 0:
            raise ValueError("parameter λ must be nonnegative")  # rootNormLhalf.jl:17-18
        self.lam = float(lam)

    lambda_ = property(lambda self: self.lam)


class IndBallL0(ProximableFunction):
    """Indicator of {x : ||x||_0 <= r}."""

    def __init__(self, r=1):
        if int(r) != r or r <= 0:
            raise ValueError("parameter r must be a positive integer")
        self.r = int(r)


class GroupNormL2(ProximableFunction):
    """h(x) = Σ_g λ_g ||x[idx_g]||_2   (src/groupNormL2.jl:15-39).

    `lam`: sequence (or device tensor) of group weights.  `idx`: list of groups; each group is a Python
    `range`/`slice` with step 1 over 0-based indices, or `slice(None)` (= Julia `:`) for "everything".
    Default `[slice(None)]` as in the reference (`idx = [:]`, groupNormL2.jl:30-31).
    """

    def __init__(self, lam=(1.0,), idx=None):
        import torch
        idx = [slice(None)] if idx is None else list(idx)
        if isinstance(lam, torch.Tensor):
            if bool((lam < 0).any()):
                raise ValueError("weights λ must be nonnegative")  # groupNormL2.jl:20-21
            nlam = lam.numel()
        else:
            lam = [float(v) for v in lam]
            if any(v < 0 for v in lam):
                raise ValueError("weights λ must be nonnegative")
            nlam = len(lam)
        if nlam != len(idx):
            raise ValueError("number of weights and groups must be the same")  # groupNormL2.jl:22-23
        self.lam = lam
        self.idx = idx

    lambda_ = property(lambda self: self.lam)

    @classmethod
    def uniform(cls, lam, group_size):
        """Groups 0:gs, gs:2gs, ... (one weight each) without materialising a million Python ranges; equivalent to
        GroupNormL2(lam, [range(i, i + gs) for i in range(0, n, gs)])."""
        import torch
        nlam = lam.numel() if isinstance(lam, torch.Tensor) else len(lam)
        obj = cls(lam, [None] * nlam)
        obj.idx = UniformGroups(int(group_size), nlam)
        return obj

    @classmethod
    def ragged(cls, lam, offsets):
        """Consecutive groups of different sizes from their CSR offsets (length ngroups + 1, 0-based), without a Python
        range per group; equivalent to GroupNormL2(lam, [range(o[g], o[g+1]) for g in ...])."""
        groups = RaggedGroups(offsets)
        obj = cls(lam, [None] * len(groups))
        obj.idx = groups
        return obj


class RaggedGroups:
    """idx of GroupNormL2.ragged: consecutive groups [offsets[g], offsets[g+1]) given by their CSR offsets."""

    def __init__(self, offsets):
        import numpy as np
        off = np.ascontiguousarray(offsets.cpu().numpy() if hasattr(offsets, "cpu") else offsets, dtype=np.int64)
        if off.ndim != 1 or off.size < 1 or off[0] < 0 or np.any(np.diff(off) < 0):
            raise ValueError("offsets must be a non-decreasing 1-D sequence starting at >= 0")
        self.offsets = off

    def __len__(self):
        return self.offsets.size - 1

    def __iter__(self):
        return (range(int(a), int(b)) for a, b in zip(self.offsets[:-1], self.offsets[1:]))


class UniformGroups:
    """idx of GroupNormL2.uniform: `count` consecutive groups of `size` indices."""

    def __init__(self, size, count):
        if size <= 0:
            raise ValueError("group size must be positive")
        self.size, self.count = size, count

    def __len__(self):
        return self.count

    def __iter__(self):
        return (range(g * self.size, (g + 1) * self.size) for g in range(self.count))
