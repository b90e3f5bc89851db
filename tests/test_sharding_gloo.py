"""CPU, world_size = 2 over gloo: the multi-GPU layout of the path.  Separable and group operators shard by
contiguous (group-aligned) index ranges with NO data-path collective; the only collectives bench.py uses
are a barrier and a MAX of the elapsed time.  The per-shard arithmetic is stood in for by the CPU oracle
(test infrastructure) -- what is under test is the partitioning and the aggregation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, gsize, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import spx_amd
    from oracle import oracle
    rng = np.random.default_rng(123)  # every rank derives the same global vectors, then keeps its shard
    x, sj, q = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)
    lam = rng.uniform(0.5, 1.5, size=n // gsize)
    lo, hi = spx_amd.shard_range(n, rank, world, align=gsize)
    y_sep = oracle.prox_l1_box(q[lo:hi], x[lo:hi], sj[lo:hi], 1.0, 1.0, -1.0, 1.0)
    y_grp = oracle.prox_group_l2_binf(q[lo:hi], x[lo:hi], sj[lo:hi], lam[lo // gsize:hi // gsize], 1.0, 1.0, gsize=gsize)
    # bench.py's aggregation: barrier, then MAX over ranks of the local elapsed time
    dist.barrier()
    t = torch.tensor([0.25 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # test-only gather of the shards to rank 0 (the product path never gathers)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, y_sep, y_grp))
    if rank == 0:
        parts.sort(key=lambda p: p[0])
        assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        np.save(out_path + ".sep.npy", np.concatenate([p[2] for p in parts]))
        np.save(out_path + ".grp.npy", np.concatenate([p[3] for p in parts]))
        np.save(out_path + ".t.npy", t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_groups", [37, 64])
def test_two_rank_sharding_matches_unsharded(tmp_path, orc, n_groups):
    gsize, world = 16, 2
    n = n_groups * gsize
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, _free_port(), n, gsize, out), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    x, sj, q = rng.normal(size=n), rng.uniform(-0.5, 0.5, size=n), rng.normal(size=n)
    lam = rng.uniform(0.5, 1.5, size=n // gsize)
    np.testing.assert_array_equal(np.load(out + ".sep.npy"), orc.prox_l1_box(q, x, sj, 1.0, 1.0, -1.0, 1.0))
    np.testing.assert_array_equal(np.load(out + ".grp.npy"),
                                  orc.prox_group_l2_binf(q, x, sj, lam, 1.0, 1.0, gsize=gsize))
    assert float(np.load(out + ".t.npy")[0]) == 1.25  # MAX over ranks
