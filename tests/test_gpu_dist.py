"""The limb-sharded key switch on the GPU: the C-ABI phase path against fhe_keyswitch_apply and the
oracle, an executed RCCL join (backend "nccl", one rank -- a one-GPU box cannot host more RCCL ranks), and the real
multi-rank plan with two and three ranks sharing the one GPU, exchanged over gloo."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(logn, L, K, dnum, bits=50, seed=0):
    import fhe_reliability_gpu_amd as F
    N = 1 << logn
    qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
    rng = np.random.default_rng(seed + 1000 * L + 10 * K + dnum)
    c = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    add = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    evk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
    return qs, c, evk, add


def _to_cuda(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()


def _from_cuda(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("logn,L,K,dnum,bits", [(13, 5, 2, 3, 50), (12, 6, 3, 2, 61), (10, 22, 11, 2, 50)])
def test_phase_path_on_one_rank_equals_apply(logn, L, K, dnum, bits):
    """fhe_keyswitch_create_sharded with world = 1 (gather buffers bound, slots = whole buffers) run phase by phase equals
    fhe_keyswitch_apply word for word, with and without an addend."""
    import torch

    import fhe_reliability_gpu_amd as F
    from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, sharded_keyswitch
    eng = F.default_engine()
    qs, c, evk, add = _case(logn, L, K, dnum, bits)
    t = eng.tables(logn, qs)
    w0, w1 = F.KeySwitch(eng, t, L, K, dnum).apply(eng.upload(c), eng.upload(evk))
    w0, w1 = w0.download(), w1.download()
    plan = ShardedKeySwitch(eng, t, L, K, dnum)
    with torch.cuda.stream(torch.cuda.Stream()):
        o0, o1 = sharded_keyswitch(plan, _to_cuda(c), _to_cuda(evk))
        a0, a1 = sharded_keyswitch(plan, _to_cuda(c), _to_cuda(evk), add0=_to_cuda(add), add1=_to_cuda(add))
        torch.cuda.current_stream().synchronize()
    assert (_from_cuda(o0) == w0).all() and (_from_cuda(o1) == w1).all()
    qcol = np.array(qs[:L], dtype=np.uint64)[:, None]
    assert (_from_cuda(a0) == (w0 + add) % qcol).all() and (_from_cuda(a1) == (w1 + add) % qcol).all()


def _nccl_worker(rank, world, port, out_dir):
    # fresh process: nothing has touched the GPU before the process group exists
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        import fhe_reliability_gpu_amd as F
        from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, limb_shard, sharded_base_conversion, sharded_keyswitch, sharded_rotate
        from oracle import cport as O
        from oracle.keyswitch_ref import rotate_ref
        eng = F.Engine(0)
        logn, L, K, dnum = 13, 6, 2, 3
        qs, c, evk, add = _case(logn, L, K, dnum)
        t = eng.tables(logn, qs)
        w0, w1 = F.KeySwitch(eng, t, L, K, dnum).apply(eng.upload(c), eng.upload(evk))
        plan = ShardedKeySwitch(eng, t, L, K, dnum)
        tm = {}
        o0, o1 = sharded_keyswitch(plan, _to_cuda(c), _to_cuda(evk), timings=tm)
        r0, r1 = sharded_rotate(plan, _to_cuda(add), _to_cuda(c), 5, _to_cuda(evk))
        torch.cuda.synchronize()
        ok_ks = bool((_from_cuda(o0) == w0.download()).all() and (_from_cuda(o1) == w1.download()).all())
        v0, v1 = rotate_ref(add, c, 5, evk, qs, L, K, dnum, logn)
        ok_rot = bool((_from_cuda(r0) == v0).all() and (_from_cuda(r1) == v1).all())
        ev = tm["events"][0]
        gather_ms = ev[1].elapsed_time(ev[2]) + ev[3].elapsed_time(ev[4])
        # the base-conversion join of config 4 through the same backend
        lo, hi = limb_shard(L, world, rank)
        out = sharded_base_conversion(eng, _to_cuda(c[lo:hi]), qs[:L], qs[L:])
        olo, ohi = limb_shard(K, world, rank)
        ok_bc = bool((_from_cuda(out) == O.baseconv_exact(c, qs[:L], qs[L:])[olo:ohi]).all())
        # config 4's composite through the same backend: sharded hmult (two all-gathers + one broadcast) = fhe_hmult
        from fhe_reliability_gpu_amd.dist import sharded_hmult
        ks1 = F.KeySwitch(eng, t, L, K, dnum)
        h0, h1 = ks1.hmult(eng.upload(c), eng.upload(add), eng.upload(add), eng.upload(c), eng.upload(evk))
        s0, s1 = sharded_hmult(plan, _to_cuda(c), _to_cuda(add), _to_cuda(add), _to_cuda(c), _to_cuda(evk))
        torch.cuda.synchronize()
        ok_hm = bool((_from_cuda(s0) == h0.download()).all() and (_from_cuda(s1) == h1.download()).all())
        # the same composites issued from a USER stream (no side stream, no fences of stream_scope: phase kernels, the in-place
        # ncclAllGather and the broadcast are ordered by the one stream) and on a plan bound to a NON-default group (the broadcast's
        # source is a group rank there: dist.broadcast_rows translates it)
        sub = dist.new_group([0])
        plan_g = ShardedKeySwitch(eng, t, L, K, dnum, group=sub)
        user = torch.cuda.Stream()
        with torch.cuda.stream(user):
            u0, u1 = sharded_rotate(plan, _to_cuda(add), _to_cuda(c), 5, _to_cuda(evk))
            m0, m1 = sharded_hmult(plan_g, _to_cuda(c), _to_cuda(add), _to_cuda(add), _to_cuda(c), _to_cuda(evk))
            user.synchronize()
        ok_user = bool((_from_cuda(u0) == v0).all() and (_from_cuda(u1) == v1).all() and (_from_cuda(m0) == h0.download()).all()
                       and (_from_cuda(m1) == h1.download()).all())
        np.save(os.path.join(out_dir, f"nccl{rank}.npy"), np.array([ok_ks, ok_rot, ok_bc, ok_hm, gather_ms >= 0.0, dist.get_backend() == "nccl", ok_user]))
    finally:
        dist.destroy_process_group()


def test_rccl_join_executes_on_one_rank(tmp_path):
    """A real RCCL process group (backend "nccl"): sharded_keyswitch, sharded_rotate and sharded_base_conversion issue their
    all-gathers through ncclAllGather and reproduce the single-device results.  One rank: RCCL refuses two ranks on one GPU,
    and this box has one; the driver's 8-GPU run is where world > 1 meets RCCL."""
    import torch.multiprocessing as mp
    mp.spawn(_nccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    flags = np.load(tmp_path / "nccl0.npy")
    assert flags.all(), f"(keyswitch, rotate, baseconv, hmult, timings, backend, user stream + subgroup) = {flags.tolist()}"


def _gloo_gpu_hmult_worker(rank, world, port, logn, L, K, dnum, bits, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fhe_reliability_gpu_amd as F
        from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, ks_layout, own_ct_rows, own_rows, sharded_hmult, sharded_rescale
        eng = F.Engine(0)
        qs, a0, rlk, a1 = _case(logn, L, K, dnum, bits)
        _, b0, _, b1 = _case(logn, L, K, dnum, bits, seed=5)
        t = eng.tables(logn, qs)
        lay = ks_layout(L, K, world, rank)
        rows = own_ct_rows(lay)
        plan = ShardedKeySwitch(eng, t, L, K, dnum)
        o0, o1 = sharded_hmult(plan, _to_cuda(a0[rows]), _to_cuda(a1[rows]), _to_cuda(b0[rows]), _to_cuda(b1[rows]), _to_cuda(rlk[:, :, own_rows(lay)]))
        # where the shape allows, that ran fhe_hmult_shard_finish_begin / _end (one transform for the mod-down and the rescale); the
        # separate key-switch finish + rescale must give the same words
        fused = plan.fused_rescale
        assert fused == (logn >= 13 and K >= 2)
        eng.set_option("hmult_fused_rescale", 0)
        assert not plan.fused_rescale
        p0, p1 = sharded_hmult(plan, _to_cuda(a0[rows]), _to_cuda(a1[rows]), _to_cuda(b0[rows]), _to_cuda(b1[rows]), _to_cuda(rlk[:, :, own_rows(lay)]))
        eng.set_option("hmult_fused_rescale", 1)
        assert torch.equal(o0, p0) and torch.equal(o1, p1)
        # the BGV form of the rescale on the same plan: the owner of the last limb broadcasts t [c t^-1]_q_last's source limb
        plan.set_plain_modulus(786433)
        rb = sharded_rescale(plan, torch.stack([_to_cuda(a0[rows]), _to_cuda(a1[rows])]))
        plan.set_plain_modulus(0)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"h{rank}.npy"), np.stack([_from_cuda(o0), _from_cuda(o1), _from_cuda(rb[0]), _from_cuda(rb[1])]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,logn,L,K,dnum,bits", [(1, 13, 5, 2, 3, 50), (2, 12, 6, 2, 3, 50), (3, 13, 7, 3, 2, 61), (2, 14, 4, 1, 4, 50), (2, 13, 4, 2, 2, 50)])
def test_sharded_hmult_real_plan(tmp_path, world, logn, L, K, dnum, bits):
    """BASELINE config 4's composite with the limbs sharded, on the real C-ABI plan (ranks share cuda:0, joins over gloo): tensor
    product on the owned rows, sharded relinearisation (d0 / d1 as addends of its last launch), sharded rescale (the last limb's owner
    broadcasts it) -- concatenated results equal the oracle's hmult_ref, i.e. what fhe_hmult gives on one device."""
    import torch.multiprocessing as mp
    from oracle.keyswitch_ref import hmult_ref, rescale_ref
    mp.spawn(_gloo_gpu_hmult_worker, args=(world, _free_port(), logn, L, K, dnum, bits, str(tmp_path)), nprocs=world, join=True)
    qs, a0, rlk, a1 = _case(logn, L, K, dnum, bits)
    _, b0, _, b1 = _case(logn, L, K, dnum, bits, seed=5)
    w0, w1 = hmult_ref(a0, a1, b0, b1, rlk, qs, L, K, dnum, logn, rescale=True)
    got = [np.load(tmp_path / f"h{r}.npy") for r in range(world)]
    g0 = np.concatenate([g[0] for g in got], axis=0)
    g1 = np.concatenate([g[1] for g in got], axis=0)
    assert g0.shape == w0.shape and (g0 == w0).all() and (g1 == w1).all()
    rb = rescale_ref(np.stack([a0, a1]), qs, L, logn, plain_modulus=786433)          # sharded BGV rescale of (a0, a1)
    assert (np.concatenate([g[2] for g in got], axis=0) == rb[0]).all() and (np.concatenate([g[3] for g in got], axis=0) == rb[1]).all()


def _gloo_gpu_worker(rank, world, port, logn, L, K, dnum, bits, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fhe_reliability_gpu_amd as F
        from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, ks_layout, own_ct_rows, own_rows, sharded_keyswitch, sharded_rotate
        eng = F.Engine(0)
        qs, c, evk, add = _case(logn, L, K, dnum, bits)
        t = eng.tables(logn, qs)
        lay = ks_layout(L, K, world, rank)
        plan = ShardedKeySwitch(eng, t, L, K, dnum)
        assert plan.lay == lay
        c_l, add_l, evk_l = _to_cuda(c[own_ct_rows(lay)]), _to_cuda(add[own_ct_rows(lay)]), _to_cuda(evk[:, :, own_rows(lay)])
        o0, o1 = sharded_keyswitch(plan, c_l, evk_l, add0=add_l)
        # a rotation (the automorphism on the loads of every rank's own launches) and the BGV form of the mod-down, same plan
        r0, r1 = sharded_rotate(plan, add_l, c_l, 5, evk_l)
        plan.set_plain_modulus(65537)
        b0, b1 = sharded_keyswitch(plan, c_l, evk_l, add0=add_l)
        plan.set_plain_modulus(0)
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"g{rank}.npy"), np.stack([_from_cuda(x) for x in (o0, o1, r0, r1, b0, b1)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,logn,L,K,dnum,bits", [(2, 12, 5, 3, 2, 50), (3, 11, 7, 2, 3, 50), (2, 13, 4, 1, 4, 61), (3, 10, 2, 4, 1, 50)])
def test_real_plan_with_ranks_sharing_the_gpu(tmp_path, world, logn, L, K, dnum, bits):
    """The C-ABI sharded plan with world = 2 / 3 (every rank computes on cuda:0, the joins go over gloo through host memory):
    gather-buffer row maps, per-digit gaps, a rank without special limbs, a rank without ciphertext limbs -- the concatenated
    results equal the oracle composite.  This is the multi-rank device code the 8-GPU RCCL run executes."""
    import torch.multiprocessing as mp
    from fhe_reliability_gpu_amd.dist import ks_layout
    from oracle.keyswitch_ref import keyswitch_ref, rotate_ref
    mp.spawn(_gloo_gpu_worker, args=(world, _free_port(), logn, L, K, dnum, bits, str(tmp_path)), nprocs=world, join=True)
    qs, c, evk, add = _case(logn, L, K, dnum, bits)
    w0, w1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn, add0=add)
    got = [np.load(tmp_path / f"g{r}.npy") for r in range(world)]
    g0 = np.concatenate([g[0] for g in got], axis=0)
    g1 = np.concatenate([g[1] for g in got], axis=0)
    assert g0.shape == w0.shape and (g0 == w0).all() and (g1 == w1).all()
    # the sharded rotation (fhe_rotate_shard_*) and the sharded BGV mod-down
    v0, v1 = rotate_ref(add, c, 5, evk, qs, L, K, dnum, logn)
    assert (np.concatenate([g[2] for g in got], axis=0) == v0).all() and (np.concatenate([g[3] for g in got], axis=0) == v1).all()
    p0, p1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn, add0=add, plain_modulus=65537)
    assert (np.concatenate([g[4] for g in got], axis=0) == p0).all() and (np.concatenate([g[5] for g in got], axis=0) == p1).all()
    for r in range(world):
        assert got[r].shape[1] == ks_layout(L, K, world, r)["cn"]


def _gloo_gpu_hoisted_worker(rank, world, port, logn, L, K, dnum, bits, elts, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fhe_reliability_gpu_amd as F
        from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, ks_layout, own_ct_rows, own_rows, sharded_rotate, sharded_rotate_hoisted
        eng = F.Engine(0)
        qs, c1, gk, c0 = _case(logn, L, K, dnum, bits)
        _, _, gk2, _ = _case(logn, L, K, dnum, bits, seed=9)
        t = eng.tables(logn, qs)
        lay = ks_layout(L, K, world, rank)
        plan = ShardedKeySwitch(eng, t, L, K, dnum)
        c0_l, c1_l = _to_cuda(c0[own_ct_rows(lay)]), _to_cuda(c1[own_ct_rows(lay)])
        keys = [_to_cuda(k[:, :, own_rows(lay)]) for k in (gk, gk2, gk)]
        prepared = [plan.prepare_galois_key(k, e) for k, e in zip(keys, elts)]
        outs = sharded_rotate_hoisted(plan, c0_l, c1_l, elts, prepared)
        # the plan still serves a plain sharded rotation afterwards (buffers shared between the two forms)
        p0, p1 = sharded_rotate(plan, c0_l, c1_l, elts[0], keys[0])
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"hr{rank}.npy"), np.stack([_from_cuda(x) for o in outs for x in o] + [_from_cuda(p0), _from_cuda(p1)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,logn,L,K,dnum,bits", [(1, 13, 5, 2, 3, 50), (2, 12, 6, 2, 3, 50), (3, 13, 7, 3, 2, 61), (2, 14, 4, 1, 4, 50), (3, 10, 2, 4, 1, 50)])
def test_sharded_hoisted_rotations_real_plan(tmp_path, world, logn, L, K, dnum, bits):
    """Hoisted rotations with the limbs sharded (fhe_rotate_hoisted_shard_*; ranks share cuda:0, the joins go over gloo): ONE input
    all-gather for three Galois elements, one special-limb all-gather per element; concatenated rows equal the oracle composite
    rotate_hoisted_ref, i.e. what fhe_rotate_hoisted gives on one device."""
    import torch.multiprocessing as mp
    from oracle.keyswitch_ref import rotate_hoisted_ref, rotate_ref
    N = 1 << logn
    elts = [3, 2 * N - 1, 5]
    mp.spawn(_gloo_gpu_hoisted_worker, args=(world, _free_port(), logn, L, K, dnum, bits, elts, str(tmp_path)), nprocs=world, join=True)
    qs, c1, gk, c0 = _case(logn, L, K, dnum, bits)
    _, _, gk2, _ = _case(logn, L, K, dnum, bits, seed=9)
    got = [np.load(tmp_path / f"hr{r}.npy") for r in range(world)]
    cat = lambda i: np.concatenate([g[i] for g in got], axis=0)
    for r, (e, key) in enumerate(zip(elts, (gk, gk2, gk))):
        w0, w1 = rotate_hoisted_ref(c0, c1, e, key, qs, L, K, dnum, logn)
        assert (cat(2 * r) == w0).all() and (cat(2 * r + 1) == w1).all(), f"galois element {e}"
    v0, v1 = rotate_ref(c0, c1, elts[0], gk, qs, L, K, dnum, logn)
    assert (cat(6) == v0).all() and (cat(7) == v1).all()


@pytest.mark.parametrize("world,logn,L,K,dnum,bits", [(8, 13, 10, 4, 5, 50), (8, 13, 9, 8, 3, 61), (6, 14, 7, 2, 7, 50)])
def test_many_ranks_in_one_process_real_plan(world, logn, L, K, dnum, bits):
    """The driver's largest launch has 8 ranks; a GPU box admits 6 processes.  Here EVERY rank's C-ABI plan (fhe_keyswitch_create_sharded with
    world = 8 and rank = 0..7: row maps, gaps, ranks that own no special limb, ranks that own one ciphertext limb) runs in ONE process, the
    three collectives replaced by the slot copies they amount to: a sharded rotation, three hoisted rotations and a sharded multiply + relinearize + rescale (the flow with
    the broadcast between the conversion and the last transform where the shape allows, and the separate-rescale flow) -- concatenated rows
    against fhe_rotate / fhe_hmult on one device, word for word."""
    import ctypes as C

    import torch

    import fhe_reliability_gpu_amd as F
    from fhe_reliability_gpu_amd._lib import check, lib, vp
    from fhe_reliability_gpu_amd.dist import ks_layout, own_ct_rows, own_rows
    eng = F.Engine(0)
    qs, c1, key, c0 = _case(logn, L, K, dnum, bits)
    _, b0, _, b1 = _case(logn, L, K, dnum, bits, seed=5)
    N = 1 << logn
    t = eng.tables(logn, qs)
    P = lambda x: C.c_void_p(x.data_ptr() if x is not None and x.numel() else 0)
    lays = [ks_layout(L, K, world, r) for r in range(world)]
    cmax, smax = lays[0]["cmax"], lays[0]["smax"]
    zeros = lambda rows: torch.zeros((rows, N), dtype=torch.int64, device="cuda")
    g1 = [zeros(world * cmax) for _ in range(world)]
    g2 = [zeros(world * 2 * smax) for _ in range(world)]
    bc = [zeros(3) for _ in range(world)]
    plans = []
    for r in range(world):
        h = vp()
        check(lib.fhe_keyswitch_create_sharded(eng._h, t._h, L, K, dnum, world, r, P(g1[r]), P(g2[r]), P(bc[r]), C.byref(h)))
        plans.append(h)
    owner = next(r for r, l in enumerate(lays) if l["clo"] <= L - 1 < l["clo"] + l["cn"])
    # (the library's launches run on the engine's own non-blocking stream, torch's fills on torch's: outputs are torch.empty, and the
    # buffers that were zero-filled are settled before the first launch)
    empty = lambda *shape: torch.empty(shape, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    loc = lambda a, r: _to_cuda(a[own_ct_rows(lays[r])])
    keyl = [_to_cuda(key[:, :, own_rows(lays[r])]) for r in range(world)]

    def gather(bufs, rows):
        torch.cuda.synchronize()
        for r in range(world):
            for s in range(world):
                if r != s:
                    bufs[r][s * rows:(s + 1) * rows] = bufs[s][s * rows:(s + 1) * rows]
        torch.cuda.synchronize()

    def bcast(n_rows):
        torch.cuda.synchronize()
        for r in range(world):
            if r != owner:
                bc[r][:n_rows] = bc[owner][:n_rows]
        torch.cuda.synchronize()

    try:
        # ---- rotation
        g = 5
        c0l, c1l = [loc(c0, r) for r in range(world)], [loc(c1, r) for r in range(world)]
        for r in range(world):
            check(lib.fhe_rotate_shard_begin(eng._h, plans[r], P(c1l[r]), g, None))
        gather(g1, cmax)
        for r in range(world):
            check(lib.fhe_rotate_shard_inner(eng._h, plans[r], P(keyl[r]), None))
        gather(g2, 2 * smax)
        rot = []
        for r in range(world):
            o0, o1 = empty(lays[r]["cn"], N), empty(lays[r]["cn"], N)
            check(lib.fhe_rotate_shard_finish(eng._h, plans[r], P(o0), P(o1), P(c0l[r]), g, None))
            rot.append((o0, o1))
        eng.sync()
        ks = F.KeySwitch(eng, t, L, K, dnum)
        w0, w1 = ks.rotate(eng.upload(c0), eng.upload(c1), g, eng.upload(key))
        assert (np.concatenate([_from_cuda(o[0]) for o in rot]) == w0.download()).all()
        assert (np.concatenate([_from_cuda(o[1]) for o in rot]) == w1.download()).all()
        # ---- hoisted rotations: ONE gather of the input for all elements, one gather of the special limbs per element
        elts = [3, 2 * N - 1, 9]
        pk_full = [ks.prepare_galois_key(eng.upload(key), e) for e in elts]
        want = ks.rotate_hoisted(eng.upload(c0), eng.upload(c1), elts, pk_full)
        pkl = []
        for r in range(world):
            per = []
            for e in elts:
                o = torch.empty_like(keyl[r])
                check(lib.fhe_galois_key_prepare(eng._h, plans[r], P(o), P(keyl[r]), e, None))
                per.append(o)
            pkl.append(per)
        for r in range(world):
            check(lib.fhe_rotate_hoisted_shard_begin(eng._h, plans[r], P(c1l[r]), None))
        gather(g1, cmax)
        for r in range(world):
            check(lib.fhe_rotate_hoisted_shard_extend(eng._h, plans[r], None))
        for i, e in enumerate(elts):
            for r in range(world):
                check(lib.fhe_rotate_hoisted_shard_inner(eng._h, plans[r], P(c1l[r]), P(pkl[r][i]), e, None))
            gather(g2, 2 * smax)
            hz = []
            for r in range(world):
                o = empty(2, lays[r]["cn"], N)
                check(lib.fhe_rotate_hoisted_shard_finish(eng._h, plans[r], P(o[0]), P(o[1]), P(c0l[r]), e, None))
                hz.append(o)
            eng.sync()
            assert (np.concatenate([_from_cuda(o[0]) for o in hz]) == want[i][0].download()).all(), e
            assert (np.concatenate([_from_cuda(o[1]) for o in hz]) == want[i][1].download()).all(), e
        # ---- multiply + relinearize + rescale, both flows
        h0, h1 = ks.hmult(eng.upload(c0), eng.upload(c1), eng.upload(b0), eng.upload(b1), eng.upload(key), rescale=True)
        h0, h1 = h0.download(), h1.download()
        rs_rows = [max(0, min(l["clo"] + l["cn"], L - 1) - l["clo"]) for l in lays]
        fusable = bool(lib.fhe_hmult_shard_fusable(eng._h, plans[0]))
        assert fusable == (logn >= 13 and K >= 2)
        for fused in ([True, False] if fusable else [False]):
            eng.set_option("hmult_fused_rescale", 1 if fused else 0)
            d = []
            for r in range(world):
                a0, a1, x0, x1 = loc(c0, r), loc(c1, r), loc(b0, r), loc(b1, r)
                dd = [torch.empty_like(a0) for _ in range(3)]
                if lays[r]["cn"]:
                    check(lib.fhe_tensor_product(eng._h, P(dd[0]), P(dd[1]), P(dd[2]), P(a0), P(a1), P(x0), P(x1), t._h, lays[r]["cn"], lays[r]["clo"], None))
                d.append(dd)
            for r in range(world):
                check(lib.fhe_keyswitch_shard_begin(eng._h, plans[r], P(d[r][2]), None))
            gather(g1, cmax)
            for r in range(world):
                check(lib.fhe_keyswitch_shard_inner(eng._h, plans[r], P(d[r][2]), P(keyl[r]), None))
            gather(g2, 2 * smax)
            outs = []
            if fused:
                for r in range(world):
                    check(lib.fhe_hmult_shard_finish_begin(eng._h, plans[r], P(d[r][0]), P(d[r][1]), None))
                bcast(2)
                for r in range(world):
                    o = empty(2, rs_rows[r], N)
                    check(lib.fhe_hmult_shard_finish_end(eng._h, plans[r], P(o[0]), P(o[1]), P(d[r][0]), P(d[r][1]), None))
                    outs.append(o)
            else:
                mid = []
                for r in range(world):
                    m = empty(2, lays[r]["cn"], N)
                    check(lib.fhe_keyswitch_shard_finish(eng._h, plans[r], P(m[0]), P(m[1]), P(d[r][0]), P(d[r][1]), None))
                    mid.append(m)
                for r in range(world):
                    check(lib.fhe_rescale_shard_begin(eng._h, plans[r], P(mid[r]), 2, None))
                bcast(2)
                for r in range(world):
                    o = empty(2, rs_rows[r], N)
                    check(lib.fhe_rescale_shard_finish(eng._h, plans[r], P(o), P(mid[r]), 2, None))
                    outs.append(o)
            eng.sync()
            assert (np.concatenate([_from_cuda(o[0]) for o in outs]) == h0).all(), fused
            assert (np.concatenate([_from_cuda(o[1]) for o in outs]) == h1).all(), fused
    finally:
        eng.set_option("hmult_fused_rescale", 1)
        for h in plans:
            lib.fhe_keyswitch_destroy(h)
