"""Multi-GPU layer: RNS limbs shard across ranks, one exchange at the base-conversion join.

The reference has no multi-device code; NTT, INTT and coefficient-wise
products are independent per limb, so ranks need no communication for them.  Only base
conversion needs every input limb of a coefficient (motivation/baseConv.py:75-78): there the
per-rank slabs are all-gathered (RCCL over xGMI when the backend is "nccl"; the same code
runs on gloo/CPU tensors, which is how the tests cover it without GPUs).

One process per GPU, launched by torch.distributed.run; no collective inside the NTT path.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def limb_shard(n_limbs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slab [lo, hi) of limbs owned by `rank`: sizes differ by at most one,
    earlier ranks take the larger slabs."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad world/rank")
    base, extra = divmod(n_limbs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_table(n_limbs: int, world: int) -> List[Tuple[int, int]]:
    return [limb_shard(n_limbs, world, r) for r in range(world)]


def gather_limbs(local, n_limbs: int, group=None):
    """All-gather the per-rank limb slabs ``local`` (shape [hi-lo, N], any integer dtype,
    CPU or GPU tensor) into the full [n_limbs, N] matrix on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    shards = shard_table(n_limbs, world)
    n = local.shape[-1]
    rows = max(hi - lo for lo, hi in shards)
    # equal-sized buffers for all_gather; short slabs are padded with zero rows
    pad = torch.zeros((rows, n), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(shards)], dim=0)


def sharded_base_conversion(eng, local_in, mod_in: Sequence[int], mod_out: Sequence[int], group=None, exact: bool = True):
    """Base conversion with limbs sharded over ranks (BASELINE config 4 shape).

    ``local_in``: this rank's slab of the input residues, a CUDA int64 tensor [hi-lo, N]
    (slab bounds from limb_shard(len(mod_in), world, rank)).  Returns this rank's slab of
    the OUTPUT limbs (bounds from limb_shard(len(mod_out), ...)) as a CUDA int64 tensor.
    The only collective is the all-gather of the input slabs.
    """
    import ctypes as C

    import torch
    import torch.distributed as dist

    from ._lib import check, lib
    from .engine import BaseConv

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    full = gather_limbs(local_in, len(mod_in), group).contiguous()
    lo, hi = limb_shard(len(mod_out), world, rank)
    n = full.shape[1]
    out = torch.empty((hi - lo, n), dtype=torch.int64, device=full.device)
    if hi > lo:
        plan = BaseConv(eng, mod_in, list(mod_out[lo:hi]))
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = lib.fhe_baseconv_exact if exact else lib.fhe_baseconv_fast
        check(f(eng._h, C.c_void_p(out.data_ptr()), C.c_void_p(full.data_ptr()), plan._h, n, stream))
        torch.cuda.current_stream().synchronize()
    return out


def max_over_ranks(value: float, device=None, group=None) -> float:
    """The bench contract's timing reduction: MAX over ranks of a host-measured duration."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
