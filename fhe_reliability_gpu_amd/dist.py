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


# ---------------------------------------------------------------------------
# Key switching with the RNS limbs sharded over ranks (BASELINE configs 4-5).
#
# Rank r owns the contiguous slab limb_shard(L + K, world, r) of the L ciphertext primes followed by the K
# special primes: its limbs of the input, of every key digit and of the result.  NTT / INTT and the products
# with the key are per limb, so they stay on the owner.  The two base conversions need every limb of their
# input, hence the only two collectives:
#   1. all-gather of the input in coefficient form  (L x N)       -> digit extension to the owner's limbs
#   2. all-gather of the special limbs after the inner product (2 x K x N) -> mod-down to the owner's limbs
# The per-limb work goes through a small interface (`GpuLimbOps` below: the C ABI on this rank's GPU) so the
# same orchestration runs over gloo on CPU tensors in the tests, where the interface is backed by the oracle.
# ---------------------------------------------------------------------------
def gather_rows(local, bounds: Sequence[Tuple[int, int]], group=None):
    """All-gather row blocks of unequal height: rank r contributes rows [bounds[r][0], bounds[r][1]) of the
    result (its ``local`` has that many rows); every rank gets the concatenation."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rows = max(1, max(hi - lo for lo, hi in bounds))
    pad = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(bounds)], dim=0)


class GpuLimbOps:
    """Per-limb primitives of the sharded key switch on this rank's GPU (CUDA int64 tensors [rows, N], rows =
    consecutive table limbs starting at ``start``), through the C ABI on torch's current stream."""

    def __init__(self, eng, tables):
        self.eng, self.t = eng, tables
        self._plans = {}

    def _call(self, fn, *args):
        import ctypes as C

        import torch

        from ._lib import check
        check(fn(self.eng._h, *args, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    @staticmethod
    def _p(x):
        import ctypes as C
        return C.c_void_p(x.data_ptr())

    def intt(self, x, start):
        from ._lib import lib
        if x.shape[0]:
            self._call(lib.fhe_ntt_inverse_batch, self._p(x), self.t._h, 1, x.shape[0], start)
        return x

    def ntt(self, x, start):
        from ._lib import lib
        if x.shape[0]:
            self._call(lib.fhe_ntt_forward_batch, self._p(x), self.t._h, 1, x.shape[0], start)
        return x

    def baseconv(self, x, mod_in, mod_out):
        """Exact conversion of the full input rows ``x`` ([len(mod_in), N]) to the moduli ``mod_out``."""
        import torch

        from ._lib import lib
        from .engine import BaseConv
        out = torch.empty((len(mod_out), x.shape[1]), dtype=torch.int64, device=x.device)
        if mod_out:
            key = (tuple(mod_in), tuple(mod_out))
            if key not in self._plans:
                self._plans[key] = BaseConv(self.eng, list(mod_in), list(mod_out))
            self._call(lib.fhe_baseconv_exact, self._p(out), self._p(x.contiguous()), self._plans[key]._h, x.shape[1])
        return out

    def mul_acc(self, acc, a, b, start):
        from ._lib import lib
        if acc.shape[0]:
            self._call(lib.fhe_modmul_acc, self._p(acc), self._p(a.contiguous()), self._p(b.contiguous()), self.t._h, 1, acc.shape[0], start)
        return acc

    def sub_scale(self, a, b, scal, start):
        """(a - b) * scal[l] mod q_l, rows = limbs start .. start + rows."""
        import ctypes as C

        import torch

        from ._lib import lib
        out = torch.empty_like(a)
        if a.shape[0]:
            self._call(lib.fhe_modsub, self._p(out), self._p(a.contiguous()), self._p(b.contiguous()), self.t._h, 1, a.shape[0], start)
            mul = (C.c_uint64 * a.shape[0])(*[int(s) for s in scal])
            self._call(lib.fhe_scalar_affine, self._p(out), self._p(out), mul, None, self.t._h, 1, a.shape[0], start)
        return out

    def zeros(self, rows, n, like):
        import torch
        return torch.zeros((rows, n), dtype=torch.int64, device=like.device)


def sharded_keyswitch(ops, qs: Sequence[int], L: int, K: int, dnum: int, c_local, evk_local, group=None):
    """Hybrid RNS key switching (same arithmetic as fhe_keyswitch_apply, CKKS-style mod-down) with limbs sharded.

    qs: the L ciphertext primes followed by the K special primes.  With (mlo, mhi) = limb_shard(L + K, world, rank):
      c_local   : rows = this rank's ciphertext limbs j in [mlo, min(mhi, L)), NTT form, [rows, N]
      evk_local : [dnum, 2, mhi - mlo, N], this rank's limbs of every key digit, NTT form
    Returns (out0_local, out1_local): this rank's ciphertext limbs of the result, NTT form.
    """
    import torch
    import torch.distributed as dist

    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    M = L + K
    alpha = -(-L // dnum)
    slabs = shard_table(M, world)
    mlo, mhi = slabs[rank]
    clo, chi = min(mlo, L), min(mhi, L)                      # owned ciphertext limbs
    slo, shi = max(mlo, L), max(mhi, L)                      # owned special limbs
    n = c_local.shape[-1] if c_local.shape[0] else evk_local.shape[-1]
    c_bounds = [(min(lo, L), min(hi, L)) for lo, hi in slabs]
    s_bounds = [(max(lo, L) - L, max(hi, L) - L) for lo, hi in slabs]

    # 1. input to coefficient form on its owner, then every rank gets all L limbs
    coef_local = ops.intt(c_local.clone(), clo)
    coef = gather_rows(coef_local, c_bounds, group)

    # 2. per digit: extension to this rank's limbs, transform, inner product with this rank's part of the key
    acc = [ops.zeros(mhi - mlo, n, evk_local) for _ in range(2)]
    for d in range(dnum):
        lo, hi = d * alpha, min(L, (d + 1) * alpha)
        ext = ops.zeros(mhi - mlo, n, evk_local)
        # owned limbs outside the digit come from the base extension (two runs: below and above the digit) ...
        for a, b in ((mlo, min(mhi, lo)), (max(mlo, hi), mhi)):
            if b > a:
                conv = ops.baseconv(coef[lo:hi], list(qs[lo:hi]), list(qs[a:b]))
                ext[a - mlo:b - mlo] = ops.ntt(conv, a)
        # ... the digit's own limbs that this rank owns are the input itself
        a, b = max(mlo, lo), min(mhi, hi)
        if b > a:
            ext[a - mlo:b - mlo] = c_local[a - clo:b - clo]
        for h in range(2):
            ops.mul_acc(acc[h], ext, evk_local[d, h], mlo)

    # 3. mod-down: special limbs to coefficient form on their owners, all-gather, conversion to the owned ciphertext limbs
    outs = []
    pinv = []
    for j in range(clo, chi):
        pm = 1
        for pk in qs[L:]:
            pm = pm * (pk % qs[j]) % qs[j]
        pinv.append(pow(pm, -1, qs[j]))
    for h in range(2):
        tP_local = ops.intt(acc[h][slo - mlo:shi - mlo].clone(), slo)
        tP = gather_rows(tP_local, s_bounds, group)
        conv = ops.ntt(ops.baseconv(tP, list(qs[L:]), list(qs[clo:chi])), clo)
        outs.append(ops.sub_scale(acc[h][clo - mlo:chi - mlo], conv, pinv, clo))
    return outs[0], outs[1]
