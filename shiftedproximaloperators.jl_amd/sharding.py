"""Multi-GPU layout of the path: the separable and group operators are embarrassingly parallel, so a
vector is partitioned into contiguous index ranges (group-aligned for the group operators), one per
rank, with NO collective on the data path ("no RCCL sharding is built"; the top-r
operators would need one histogram all-reduce and are run per replica only)."""


def shard_range(n, rank, world, align=1):
    """Contiguous [lo, hi) of 0:n owned by `rank` out of `world`; boundaries are multiples of `align`
    (use the group size for uniform groups).  The ranges of all ranks tile 0:n exactly."""
    if n < 0 or world <= 0 or not (0 <= rank < world) or align <= 0:
        raise ValueError("bad shard arguments")
    units = (n + align - 1) // align
    base, rem = divmod(units, world)
    lo_u = rank * base + min(rank, rem)
    hi_u = lo_u + base + (1 if rank < rem else 0)
    return min(lo_u * align, n), min(hi_u * align, n)
