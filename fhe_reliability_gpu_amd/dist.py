"""Multi-GPU layer: RNS limbs shard across ranks, one exchange at the base-conversion join.

The reference has no multi-device code; NTT, INTT and coefficient-wise
products are independent per limb, so ranks need no communication for them.  Only base
conversion needs every input limb of a coefficient (motivation/baseConv.py:75-78): there the
per-rank slabs are all-gathered in place into preallocated buffers (RCCL over xGMI when the backend is
"nccl"; the same sequencing runs on gloo/CPU tensors, which is how the tests cover it without GPUs).

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
    """All-gather the per-rank limb slabs ``local`` (shape [hi-lo, N], any integer dtype, CPU or GPU tensor; slab bounds
    from limb_shard) into the full [n_limbs, N] matrix on every rank: ONE in-place all-gather of a preallocated
    [world * rows, N] buffer (rows = the largest slab); when the slabs are equal the buffer IS the result, otherwise the
    padding rows are squeezed out with one copy."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shards = shard_table(n_limbs, world)
    n = local.shape[-1]
    rows = max(hi - lo for lo, hi in shards)
    buf = torch.zeros((world * rows, n), dtype=local.dtype, device=local.device)
    buf[rank * rows:rank * rows + local.shape[0]] = local
    all_gather_slots(buf, rows, group)
    if n_limbs == world * rows:
        return buf
    return torch.cat([buf[r * rows:r * rows + hi - lo] for r, (lo, hi) in enumerate(shards)], dim=0)


def sharded_base_conversion(eng, local_in, mod_in: Sequence[int], mod_out: Sequence[int], group=None, exact: bool = True):
    """Base conversion with limbs sharded over ranks (the join of BASELINE config 4).

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
        torch.cuda.current_stream().synchronize()       # the gathered input is complete (a NULL handle below = the engine's own stream)
        check(f(eng._h, C.c_void_p(out.data_ptr()), C.c_void_p(full.data_ptr()), plan._h, n, stream))
        eng.sync(stream if stream.value else None)
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
# Rank r owns a slab of the L ciphertext primes and a slab of the K special primes (ks_layout: the C ABI's
# fhe_keyswitch_shard_layout): its limbs of the input, of every key digit and of the result.  NTT / INTT, the digit
# extensions to the owned limbs and the products with the key run on the owner, batched exactly as on one device
# (include/fhe_mi355x.h, fhe_keyswitch_shard_*).  The two base conversions need every limb of their input, hence the
# only two collectives, both in-place all-gathers of preallocated buffers (RCCL over xGMI under backend "nccl"):
#   1. the input in coefficient form            gather1 [world][cmax][N]      (after `begin`)
#   2. the special limbs after the inner product gather2 [world][2][smax][N]  (after `inner`)
# `sharded_keyswitch` only sequences phases and collectives over a plan object; `ShardedKeySwitch` is the plan on this
# rank's GPU.  tests/ run the same sequencing over gloo with a plan backed by the CPU oracle.
# ---------------------------------------------------------------------------
def ks_layout(L: int, K: int, world: int, rank: int) -> dict:
    """Owned limbs of `rank`: ciphertext limbs [clo, clo+cn), special limbs [slo, slo+sn) (table indices), and the
    per-rank row counts of the two gather buffers (cmax, smax)."""
    import ctypes as C

    from ._lib import check, lib
    out = (C.c_int * 6)()
    check(lib.fhe_keyswitch_shard_layout(L, K, world, rank, out))
    return dict(zip(("clo", "cn", "slo", "sn", "cmax", "smax"), (int(x) for x in out)))


def _group_info(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def all_gather_slots(buf, rows: int, group=None):
    """In-place all-gather of ``buf`` ([world * rows, N]): every rank has filled rows [rank*rows, (rank+1)*rows).
    One collective on a preallocated buffer, no staging copies (ncclAllGather in place under "nccl").  CUDA tensors
    under a CPU-only backend (gloo: the 2-ranks-on-one-GPU rehearsal in tests/) are staged through host memory."""
    import torch
    import torch.distributed as dist

    world, rank = _group_info(group)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return
    mine = buf[rank * rows:(rank + 1) * rows]
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(buf.shape, dtype=buf.dtype)
        host[rank * rows:(rank + 1) * rows] = mine.cpu()
        dist.all_gather_into_tensor(host, host[rank * rows:(rank + 1) * rows], group=group)
        buf.copy_(host)
        return
    dist.all_gather_into_tensor(buf, mine, group=group)


class ShardedKeySwitch:
    """This rank's part of a limb-sharded hybrid key switch on its GPU: the C-ABI plan (fhe_keyswitch_create_sharded)
    plus the two gather buffers it was bound to.  All work is enqueued on torch's current stream."""

    def __init__(self, eng, tables, L: int, K: int, dnum: int, group=None, force_phases: bool = True):
        import ctypes as C

        import torch

        from ._lib import check, lib, vp
        self.eng, self.t, self.L, self.K, self.dnum, self.group = eng, tables, L, K, dnum, group
        self.world, self.rank = _group_info(group)
        self.lay = ks_layout(L, K, self.world, self.rank)
        n = tables.N
        dev = torch.device("cuda", eng.device)
        self.g1 = torch.zeros((self.world * self.lay["cmax"], n), dtype=torch.int64, device=dev)
        self.g2 = torch.zeros((self.world * 2 * self.lay["smax"], n), dtype=torch.int64, device=dev)
        self.bc = torch.zeros((3, n), dtype=torch.int64, device=dev)       # last limbs in coefficient form (rescale broadcast)
        h = vp()
        check(lib.fhe_keyswitch_create_sharded(eng._h, tables._h, L, K, dnum, self.world, self.rank, C.c_void_p(self.g1.data_ptr()),
                                               C.c_void_p(self.g2.data_ptr()), C.c_void_p(self.bc.data_ptr()), C.byref(h)))
        self._h = h
        self.rows1, self.rows2 = self.lay["cmax"], 2 * self.lay["smax"]
        own, rows = C.c_int(), C.c_int()
        check(lib.fhe_rescale_shard_info(h, C.byref(own), C.byref(rows)))
        self.owns_last, self.rs_rows = bool(own.value), rows.value
        self.last_owner = next(r for r in range(self.world) if (lambda l: l["clo"] <= L - 1 < l["clo"] + l["cn"])(ks_layout(L, K, self.world, r)))
        self._side = None

    def stream_scope(self):
        """The library takes a NULL stream handle to mean the engine's own (non-blocking) stream, which torch's legacy
        default stream does not order against.  When the caller is on that default stream, the phases and the collectives
        run on a side stream of the plan, fenced against the default stream at both ends."""
        import contextlib

        import torch
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != 0:
            return contextlib.nullcontext()
        if self._side is None:
            self._side = torch.cuda.Stream()
        side = self._side

        @contextlib.contextmanager
        def scope():
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                yield
            cur.wait_stream(side)
        return scope()

    @staticmethod
    def _p(x):
        import ctypes as C
        return C.c_void_p(x.data_ptr() if x is not None and x.numel() else 0)

    def _stream(self):
        import ctypes as C

        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def begin(self, c_local):
        from ._lib import check, lib
        check(lib.fhe_keyswitch_shard_begin(self.eng._h, self._h, self._p(c_local), self._stream()))

    def inner(self, c_local, evk_local):
        from ._lib import check, lib
        check(lib.fhe_keyswitch_shard_inner(self.eng._h, self._h, self._p(c_local), self._p(evk_local), self._stream()))

    def finish(self, add0=None, add1=None):
        import torch

        from ._lib import check, lib
        cn = self.lay["cn"]
        # both parts in ONE [2, cn, N] tensor (the views returned are its halves): a rescale that follows takes it as it stands,
        # where stacking two separate tensors cost a copy of the whole ciphertext (64 MiB at config 4: 25 us of a 620 us call)
        out = torch.empty((2, cn, self.t.N), dtype=torch.int64, device=self.g1.device)
        check(lib.fhe_keyswitch_shard_finish(self.eng._h, self._h, self._p(out[0]), self._p(out[1]), self._p(add0), self._p(add1), self._stream()))
        return out[0], out[1]

    def set_plain_modulus(self, t: int):
        """BGV form of the mod-down and of the rescale on this rank's rows (0 = CKKS-style flooring); every rank sets the same value."""
        from ._lib import check, lib
        check(lib.fhe_keyswitch_set_plain_modulus(self._h, t))

    # the three phases of a rotation: the automorphism rides on loads of the key switch's own launches (fhe_rotate_shard_*)
    def rotate_begin(self, c1_local, galois_elt: int):
        from ._lib import check, lib
        check(lib.fhe_rotate_shard_begin(self.eng._h, self._h, self._p(c1_local), galois_elt, self._stream()))

    def rotate_inner(self, gk_local):
        from ._lib import check, lib
        check(lib.fhe_rotate_shard_inner(self.eng._h, self._h, self._p(gk_local), self._stream()))

    def rotate_finish(self, c0_local, galois_elt: int):
        import torch

        from ._lib import check, lib
        cn = self.lay["cn"]
        out0 = torch.empty((cn, self.t.N), dtype=torch.int64, device=self.g1.device)
        out1 = torch.empty_like(out0)
        check(lib.fhe_rotate_shard_finish(self.eng._h, self._h, self._p(out0), self._p(out1), self._p(c0_local), galois_elt, self._stream()))
        return out0, out1

    # hoisted rotations: one decomposition (and ONE input all-gather) for all Galois elements, one all-gather per element
    def prepare_galois_key(self, gk_local, galois_elt: int):
        """This rank's rows of a Galois key in the un-rotated frame (fhe_galois_key_prepare); once per key."""
        import torch

        from ._lib import check, lib
        out = torch.empty_like(gk_local)
        check(lib.fhe_galois_key_prepare(self.eng._h, self._h, self._p(out), self._p(gk_local), galois_elt, self._stream()))
        return out

    def hoisted_begin(self, c1_local):
        from ._lib import check, lib
        check(lib.fhe_rotate_hoisted_shard_begin(self.eng._h, self._h, self._p(c1_local), self._stream()))

    def hoisted_extend(self):
        from ._lib import check, lib
        check(lib.fhe_rotate_hoisted_shard_extend(self.eng._h, self._h, self._stream()))

    def hoisted_inner(self, c1_local, pk_local, galois_elt: int):
        from ._lib import check, lib
        check(lib.fhe_rotate_hoisted_shard_inner(self.eng._h, self._h, self._p(c1_local), self._p(pk_local), galois_elt, self._stream()))

    def hoisted_finish(self, c0_local, galois_elt: int):
        import torch

        from ._lib import check, lib
        cn = self.lay["cn"]
        out = torch.empty((2, cn, self.t.N), dtype=torch.int64, device=self.g1.device)
        check(lib.fhe_rotate_hoisted_shard_finish(self.eng._h, self._h, self._p(out[0]), self._p(out[1]), self._p(c0_local), galois_elt, self._stream()))
        return out[0], out[1]

    def tensor(self, a0, a1, b0, b1):
        """(d0, d1, d2) of the owned limbs (phantom::multiply, dotprod_test.cu:113): no exchange."""
        import torch

        from ._lib import check, lib
        d = [torch.empty_like(a0) for _ in range(3)]
        if a0.shape[0]:
            check(lib.fhe_tensor_product(self.eng._h, self._p(d[0]), self._p(d[1]), self._p(d[2]), self._p(a0), self._p(a1), self._p(b0), self._p(b1),
                                         self.t._h, a0.shape[0], self.lay["clo"], self._stream()))
        return tuple(d)

    def rescale_begin(self, parts_local):
        from ._lib import check, lib
        check(lib.fhe_rescale_shard_begin(self.eng._h, self._h, self._p(parts_local), parts_local.shape[0], self._stream()))

    def rescale_finish(self, parts_local):
        import torch

        from ._lib import check, lib
        out = torch.empty((parts_local.shape[0], self.rs_rows, self.t.N), dtype=torch.int64, device=self.g1.device)
        check(lib.fhe_rescale_shard_finish(self.eng._h, self._h, self._p(out), self._p(parts_local), parts_local.shape[0], self._stream()))
        return out

    # hmult with the mod-down and the rescale behind one forward transform (fhe_hmult_shard_finish_*): after the second all-gather
    @property
    def fused_rescale(self) -> bool:
        from ._lib import lib
        return bool(lib.fhe_hmult_shard_fusable(self.eng._h, self._h))

    def hm_finish_begin(self, add0, add1):
        from ._lib import check, lib
        check(lib.fhe_hmult_shard_finish_begin(self.eng._h, self._h, self._p(add0), self._p(add1), self._stream()))

    def hm_finish_end(self, add0, add1):
        import torch

        from ._lib import check, lib
        out = torch.empty((2, self.rs_rows, self.t.N), dtype=torch.int64, device=self.g1.device)
        check(lib.fhe_hmult_shard_finish_end(self.eng._h, self._h, self._p(out[0]), self._p(out[1]), self._p(add0), self._p(add1), self._stream()))
        return out

    def close(self):
        from ._lib import lib
        if getattr(self, "_h", None) and self.eng._h:
            lib.fhe_keyswitch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sharded_keyswitch(plan, c_local, evk_local, add0=None, add1=None, timings=None):
    """One key switch over the ranks of ``plan.group``: begin, all-gather 1, inner, all-gather 2, finish.

    c_local   : this rank's ciphertext limbs, NTT form, [cn, N]
    evk_local : this rank's rows of every key digit, [dnum, 2, cn + sn, N] (ciphertext rows first), NTT form
    add0/add1 : optional [cn, N] terms added to the two output parts (rotation: sigma(c0); relinearisation: d0, d1)
    Returns this rank's ciphertext limbs of the two output parts.  ``timings`` (a dict) receives per-phase CUDA events
    when given: compute phases and the two joins are then reported separately (bench.py's strong-scaling leg)."""
    import contextlib
    with (plan.stream_scope() if hasattr(plan, "stream_scope") else contextlib.nullcontext()):
        return _sharded_keyswitch(plan, c_local, evk_local, add0, add1, timings)


def _sharded_keyswitch(plan, c_local, evk_local, add0, add1, timings, galois=0):
    """galois != 0: a rotation -- c_local is c1, add0 is c0, both un-permuted; the plan applies sigma on its loads"""
    ev = None
    if timings is not None:
        import torch
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        ev[0].record()
    if galois:
        plan.rotate_begin(c_local, galois)
    else:
        plan.begin(c_local)
    if ev:
        ev[1].record()
    all_gather_slots(plan.g1, plan.rows1, plan.group)
    if ev:
        ev[2].record()
    if galois:
        plan.rotate_inner(evk_local)
    else:
        plan.inner(c_local, evk_local)
    if ev:
        ev[3].record()
    all_gather_slots(plan.g2, plan.rows2, plan.group)
    if ev:
        ev[4].record()
    out = plan.rotate_finish(add0, galois) if galois else plan.finish(add0, add1)
    if ev:
        ev[5].record()
        timings.setdefault("events", []).append(ev)
    return out


def sharded_rotate(plan, c0_local, c1_local, galois_elt: int, gk_local, timings=None):
    """ROTATE with limbs sharded: the automorphism permutes slots inside each limb (no exchange) and rides on the loads of the key
    switch's own launches: the key switch of sigma(c1) with sigma(c0) added to the first part, same two joins."""
    with plan.stream_scope():
        return _sharded_keyswitch(plan, c1_local, gk_local, c0_local, None, timings, galois=galois_elt)


def sharded_rotate_hoisted(plan, c0_local, c1_local, galois_elts, prepared_keys_local):
    """Rotations of ONE sharded ciphertext by several Galois elements (the baby steps of profile_framewk/src/matmul_ckks.cpp:45-113):
    the input's all-gather and the digit extension happen once, each element costs the inner product, ONE all-gather (the special limbs of
    its sums) and the mod-down.  prepared_keys_local: this rank's key rows in the un-rotated frame (plan.prepare_galois_key).
    Returns [(out0_local, out1_local), ...]."""
    import contextlib
    with (plan.stream_scope() if hasattr(plan, "stream_scope") else contextlib.nullcontext()):
        plan.hoisted_begin(c1_local)
        all_gather_slots(plan.g1, plan.rows1, plan.group)
        plan.hoisted_extend()
        outs = []
        for g, pk in zip(galois_elts, prepared_keys_local):
            plan.hoisted_inner(c1_local, pk, int(g))
            all_gather_slots(plan.g2, plan.rows2, plan.group)
            outs.append(plan.hoisted_finish(c0_local, int(g)))
        return outs


def broadcast_rows(buf, src: int, group=None):
    """In-place broadcast of ``buf`` from rank ``src`` OF ``group`` (CUDA tensors under gloo are staged through host memory).
    torch.distributed.broadcast takes the source as a GLOBAL rank: a group rank is translated first."""
    import torch.distributed as dist

    world, _ = _group_info(group)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return
    if group is not None:
        src = dist.get_global_rank(group, src)
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        dist.broadcast(host, src=src, group=group)
        buf.copy_(host)
        return
    dist.broadcast(buf, src=src, group=group)


def sharded_rescale(plan, parts_local, timings=None):
    """mod_switch_to_next / rescale with the limbs sharded: parts_local = [n_parts, cn, N] (this rank's rows of every part).  The
    owner of the last ciphertext limb turns it to coefficient form, ONE broadcast of n_parts x N words, then every rank forms
    (c - delta) / q_last on its rows below the dropped limb.  Returns [n_parts, rows, N].  ``timings``: CUDA events
    (start, before the broadcast, after it, end) are appended under "rescale_events"."""
    import contextlib
    with (plan.stream_scope() if hasattr(plan, "stream_scope") else contextlib.nullcontext()):
        ev = None
        if timings is not None:
            import torch
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        plan.rescale_begin(parts_local)
        if ev:
            ev[1].record()
        broadcast_rows(plan.bc, plan.last_owner, plan.group)
        if ev:
            ev[2].record()
        out = plan.rescale_finish(parts_local)
        if ev:
            ev[3].record()
            timings.setdefault("rescale_events", []).append(ev)
        return out


def sharded_hmult(plan, a0, a1, b0, b1, rlk_local, rescale: bool = True, timings=None):
    """multiply -> relinearize -> mod_switch_to_next (reliability_test/dotprod_test.cu:113-115; BASELINE config 4) on this rank's
    rows: tensor product (no exchange), the sharded key switch of d2 with d0 / d1 as addends (two all-gathers), the sharded
    rescale (one broadcast).  Inputs [cn, N] each; returns the two parts' owned rows ([rows, N] each, the last limb dropped)."""
    import contextlib

    import torch
    with (plan.stream_scope() if hasattr(plan, "stream_scope") else contextlib.nullcontext()):
        d0, d1, d2 = plan.tensor(a0, a1, b0, b1)
        if rescale and getattr(plan, "fused_rescale", False):
            # the mod-down and the rescale share one forward transform: the broadcast (y = the last limbs after the mod-down, in
            # coefficient form) sits between the conversion and that transform; same three collectives
            ev = None
            if timings is not None:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
                ev[0].record()
            plan.begin(d2)
            if ev:
                ev[1].record()
            all_gather_slots(plan.g1, plan.rows1, plan.group)
            if ev:
                ev[2].record()
            plan.inner(d2, rlk_local)
            if ev:
                ev[3].record()
            all_gather_slots(plan.g2, plan.rows2, plan.group)
            if ev:
                ev[4].record()
            plan.hm_finish_begin(d0, d1)
            if ev:
                ev[5].record()
            broadcast_rows(plan.bc[:2], plan.last_owner, plan.group)
            if ev:
                ev[6].record()
            r = plan.hm_finish_end(d0, d1)
            if ev:
                ev[7].record()
                timings.setdefault("events", []).append(ev[:6])
                timings.setdefault("rescale_events", []).append([ev[4], ev[5], ev[6], ev[7]])
            return r[0], r[1]
        c0, c1 = sharded_keyswitch(plan, d2, rlk_local, add0=d0, add1=d1, timings=timings)
        if not rescale:
            return c0, c1
        both = c0._base if getattr(c0, "_base", None) is not None and c0._base.dim() == 3 and c0._base.shape[0] == 2 else torch.stack([c0, c1])
        r = sharded_rescale(plan, both, timings=timings)
        return r[0], r[1]


def own_ct_rows(lay) -> List[int]:
    """Table indices of the ciphertext limbs a rank with layout ``lay`` owns (rows of its input / output slabs)."""
    return list(range(lay["clo"], lay["clo"] + lay["cn"]))


def own_rows(lay) -> List[int]:
    """Table indices of every limb the rank owns, in the row order of its key slab: ciphertext limbs, then special limbs."""
    return own_ct_rows(lay) + list(range(lay["slo"], lay["slo"] + lay["sn"]))
