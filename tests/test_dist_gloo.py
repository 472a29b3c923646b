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


# ---------------------------------------------------------------- sharded key switch (limbs over ranks, two all-gathers)
def _ks_case(logn, L, K, dnum):
    from oracle import cport as O
    N = 1 << logn
    qs = [int(q) for q in O.gen_primes(N, 30, L + K)]
    rng = np.random.default_rng(L * 100 + K * 10 + dnum)
    c = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    add = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    evk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
    return qs, c, evk, add


def _ks_worker(rank, world, port, logn, L, K, dnum, out_dir):
    import torch
    import torch.distributed as dist

    from fhe_reliability_gpu_amd.dist import ks_layout, own_ct_rows, own_rows, sharded_keyswitch
    from helpers.oracle_shard_plan import OracleShardPlan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qs, c, evk, add = _ks_case(logn, L, K, dnum)
        lay = ks_layout(L, K, world, rank)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy())
        plan = OracleShardPlan(qs, logn, L, K, dnum)
        o0, o1 = sharded_keyswitch(plan, to_t(c[own_ct_rows(lay)]), to_t(evk[:, :, own_rows(lay)]), add0=to_t(add[own_ct_rows(lay)]))
        np.save(os.path.join(out_dir, f"ks{rank}.npy"), np.stack([o0.numpy().view(np.uint64), o1.numpy().view(np.uint64)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,L,K,dnum", [(2, 4, 2, 2), (3, 5, 2, 3), (2, 3, 1, 3), (3, 2, 4, 1), (8, 10, 4, 5)])   # (8 ranks: the driver's largest launch; four of them own no special limb)
def test_sharded_keyswitch_gloo(tmp_path, world, L, K, dnum):
    """Limbs sharded over ranks, in-place all-gathers at the two base-conversion joins: the concatenated per-rank results equal
    the single-device key switch (oracle/keyswitch_ref.py, the composite the GPU tests pin fhe_keyswitch_apply to).  Covers
    uneven slabs, a rank without special limbs and (L = 2, world = 3) a rank without ciphertext limbs."""
    import torch.multiprocessing as mp
    from fhe_reliability_gpu_amd.dist import ks_layout
    from oracle.keyswitch_ref import keyswitch_ref

    logn = 6
    port = _free_port()
    mp.spawn(_ks_worker, args=(world, port, logn, L, K, dnum, str(tmp_path)), nprocs=world, join=True)
    qs, c, evk, add = _ks_case(logn, L, K, dnum)
    want0, want1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn, add0=add)
    got = [np.load(tmp_path / f"ks{r}.npy") for r in range(world)]
    got0 = np.concatenate([g[0] for g in got], axis=0)
    got1 = np.concatenate([g[1] for g in got], axis=0)
    assert got0.shape == want0.shape and (got0 == want0).all() and (got1 == want1).all()
    for r in range(world):
        assert got[r].shape[1] == ks_layout(L, K, world, r)["cn"]


def _hoist_worker(rank, world, port, logn, L, K, dnum, elts, out_dir):
    import torch
    import torch.distributed as dist

    from fhe_reliability_gpu_amd.dist import ks_layout, own_ct_rows, own_rows, sharded_rotate_hoisted
    from helpers.oracle_shard_plan import OracleShardPlan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qs, c1, gk, c0 = _ks_case(logn, L, K, dnum)
        lay = ks_layout(L, K, world, rank)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy())
        plan = OracleShardPlan(qs, logn, L, K, dnum)
        gk_l = to_t(gk[:, :, own_rows(lay)])
        prepared = [plan.prepare_galois_key(gk_l, e) for e in elts]
        outs = sharded_rotate_hoisted(plan, to_t(c0[own_ct_rows(lay)]), to_t(c1[own_ct_rows(lay)]), elts, prepared)
        np.save(os.path.join(out_dir, f"hz{rank}.npy"), np.stack([x.numpy().view(np.uint64) for o in outs for x in o]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,L,K,dnum", [(2, 4, 2, 2), (3, 5, 2, 3), (3, 2, 4, 1)])
def test_sharded_rotate_hoisted_gloo(tmp_path, world, L, K, dnum):
    """dist.sharded_rotate_hoisted's sequencing over gloo with an oracle-backed plan: ONE all-gather of the input for all Galois
    elements, one all-gather of the special limbs per element; the concatenated rows equal the single-device hoisted rotation
    (oracle/keyswitch_ref.py rotate_hoisted_ref)."""
    import torch.multiprocessing as mp
    from oracle.keyswitch_ref import rotate_hoisted_ref

    logn = 6
    elts = [3, 2 * (1 << logn) - 1]
    mp.spawn(_hoist_worker, args=(world, _free_port(), logn, L, K, dnum, elts, str(tmp_path)), nprocs=world, join=True)
    qs, c1, gk, c0 = _ks_case(logn, L, K, dnum)
    got = [np.load(tmp_path / f"hz{r}.npy") for r in range(world)]
    for r, e in enumerate(elts):
        w0, w1 = rotate_hoisted_ref(c0, c1, e, gk, qs, L, K, dnum, logn)
        assert (np.concatenate([g[2 * r] for g in got], axis=0) == w0).all() and (np.concatenate([g[2 * r + 1] for g in got], axis=0) == w1).all(), e


def _hm_worker(rank, world, port, logn, L, K, dnum, out_dir, fused=True):
    import torch
    import torch.distributed as dist

    from fhe_reliability_gpu_amd.dist import ks_layout, own_ct_rows, own_rows, sharded_hmult
    from helpers.oracle_shard_plan import OracleShardPlan

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qs, a0, rlk, a1 = _ks_case(logn, L, K, dnum)
        _, b0, _, b1 = _ks_case(logn, L, K, dnum + 7)
        b0, b1 = b0 % np.array(qs[:L], dtype=np.uint64)[:, None], b1 % np.array(qs[:L], dtype=np.uint64)[:, None]
        lay = ks_layout(L, K, world, rank)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy())
        rows = own_ct_rows(lay)
        plan = OracleShardPlan(qs, logn, L, K, dnum)
        plan.fused_rescale = fused       # True: the broadcast sits between the conversion and the last transform (hm_finish_begin / _end)
        o0, o1 = sharded_hmult(plan, to_t(a0[rows]), to_t(a1[rows]), to_t(b0[rows]), to_t(b1[rows]), to_t(rlk[:, :, own_rows(lay)]))
        np.save(os.path.join(out_dir, f"hm{rank}.npy"), np.stack([o0.numpy().view(np.uint64), o1.numpy().view(np.uint64)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,L,K,dnum,fused", [(2, 4, 2, 2, True), (3, 5, 2, 3, True), (3, 3, 1, 3, True), (2, 4, 2, 2, False), (3, 5, 2, 3, False), (8, 9, 8, 3, True)])
def test_sharded_hmult_gloo(tmp_path, world, L, K, dnum, fused):
    """BASELINE config 4's composite with the limbs sharded: tensor product on the owned rows, the sharded key switch with d0 / d1 as
    addends, the rescale with ONE broadcast of the last limbs -- concatenated per-rank results equal oracle hmult_ref.  Both flows of
    dist.sharded_hmult: the broadcast after the key switch's finish (separate rescale) and between its conversion and last transform."""
    import torch.multiprocessing as mp
    from oracle.keyswitch_ref import hmult_ref

    logn = 6
    mp.spawn(_hm_worker, args=(world, _free_port(), logn, L, K, dnum, str(tmp_path), fused), nprocs=world, join=True)
    qs, a0, rlk, a1 = _ks_case(logn, L, K, dnum)
    _, b0, _, b1 = _ks_case(logn, L, K, dnum + 7)
    qcol = np.array(qs[:L], dtype=np.uint64)[:, None]
    w0, w1 = hmult_ref(a0, a1, b0 % qcol, b1 % qcol, rlk, qs, L, K, dnum, logn, rescale=True)
    got = [np.load(tmp_path / f"hm{r}.npy") for r in range(world)]
    g0 = np.concatenate([g[0] for g in got], axis=0)
    g1 = np.concatenate([g[1] for g in got], axis=0)
    assert g0.shape == w0.shape == (L - 1, 1 << logn) and (g0 == w0).all() and (g1 == w1).all()


def test_ks_layout_tiles_both_limb_sets():
    """Every ciphertext limb and every special limb has exactly one owner; slab sizes differ by at most one; the total
    per rank (the work per rank) differs by at most one more; cmax / smax are the largest slabs."""
    from fhe_reliability_gpu_amd.dist import ks_layout, own_ct_rows, own_rows
    for L, K in ((44, 11), (44, 4), (44, 1), (32, 8), (4, 2), (2, 4), (16, 4)):
        for world in (1, 2, 3, 4, 8):
            lays = [ks_layout(L, K, world, r) for r in range(world)]
            ct = sorted(sum((own_ct_rows(l) for l in lays), []))
            allr = sorted(sum((own_rows(l) for l in lays), []))
            assert ct == list(range(L)) and allr == list(range(L + K))
            assert max(l["cn"] for l in lays) - min(l["cn"] for l in lays) <= 1
            assert max(l["sn"] for l in lays) - min(l["sn"] for l in lays) <= 1
            assert all(l["cmax"] == max(x["cn"] for x in lays) and l["smax"] == max(x["sn"] for x in lays) for l in lays)
            tot = [l["cn"] + l["sn"] for l in lays]
            assert max(tot) - min(tot) <= 1, (L, K, world, tot)
