"""The N > 1 path on CPU: limb sharding, the base-conversion all-gather and the bench's
max-over-ranks reduction, world_size 2 and 3 over gloo (no GPU needed)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_limbs, N, out_dir):
    import torch
    import torch.distributed as dist

    from fhe_reliability_gpu_amd.dist import gather_limbs, limb_shard, max_over_ranks

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(n_limbs * N, dtype=torch.int64).reshape(n_limbs, N) * 7 + 3
        lo, hi = limb_shard(n_limbs, world, rank)
        got = gather_limbs(full[lo:hi].clone(), n_limbs)
        assert got.shape == full.shape and torch.equal(got, full)
        t = max_over_ranks(0.25 + rank)
        assert t == 0.25 + world - 1
        dist.barrier()
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_limbs", [(2, 16), (2, 5), (3, 7)])
def test_gather_limbs_gloo(tmp_path, world, n_limbs):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_limbs, 64, str(tmp_path)), nprocs=world, join=True)
    bounds = [np.load(tmp_path / f"ok{r}.npy") for r in range(world)]
    # slabs tile [0, n_limbs) without gaps, sizes differ by at most one
    assert bounds[0][0] == 0 and bounds[-1][1] == n_limbs
    for a, b in zip(bounds, bounds[1:]):
        assert a[1] == b[0]
    sizes = [int(b[1] - b[0]) for b in bounds]
    assert max(sizes) - min(sizes) <= 1


def test_limb_shard_properties():
    from fhe_reliability_gpu_amd.dist import limb_shard, shard_table
    for L in (1, 7, 16, 32, 44):
        for G in (1, 2, 4, 8):
            tab = shard_table(L, G)
            assert tab[0][0] == 0 and tab[-1][1] == L
            assert all(a[1] == b[0] for a, b in zip(tab, tab[1:]))
            assert sorted((hi - lo for lo, hi in tab), reverse=True) == [hi - lo for lo, hi in tab]
    assert limb_shard(44, 8, 0) == (0, 6) and limb_shard(44, 8, 7) == (39, 44)
    with pytest.raises(ValueError):
        limb_shard(4, 2, 2)
