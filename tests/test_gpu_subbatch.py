"""Sub-batched two-launch transforms (capi.cpp: sub_batch_polys / for_sub_batches): a call whose batch cannot stay in the Infinity
Cache runs as sub-batches, alternating between the caller's stream and the context's side stream, handing over through per-stream
scratch.  A piece is a window of limbs x a range of polynomials (as many polynomials of as few limbs as fit).  Same words as the
oracle whatever the cut, the ragged tails in both directions, the arithmetic path, the stream split or the hand-off."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    import fhe_reliability_gpu_amd as f
    return f


@pytest.fixture(scope="module")
def eng(F):
    return F.default_engine()


@pytest.fixture(scope="module")
def O():
    from oracle import cport
    return cport


@pytest.fixture
def small_chunks(eng):
    """1 MiB sub-batches with no lower bound on the batch size, so that small inputs are cut"""
    eng.set_option("ntt_chunk_mib", 1)
    eng.set_option("ntt_chunk_floor_mib", 0)
    yield
    eng.sync()
    eng.check()
    for k, v in (("ntt_chunk_mib", 96), ("ntt_chunk_floor_mib", 192), ("ntt_split", -1), ("ntt_pingpong", -1), ("ntt_stream", -1)):
        eng.set_option(k, v)


def _oracle_forward(O, data, qs, logn, start=0):
    rps = np.stack([O.root_powers(q, logn) for q in qs])
    return np.stack([O.nwt_forward_batch(p, qs, rps) for p in data])


@pytest.mark.parametrize("split,pingpong", [(1, -1), (0, -1), (0, 0), (1, 1)])
@pytest.mark.parametrize("logn,bits,n_poly", [(13, [50, 50, 61], 37), (13, [50] * 5, 6), (13, [61] * 7, 3), (14, [61], 19), (16, [50], 7), (16, [50, 61], 5),
                                              (16, [50, 50, 50], 1), (17, [50], 3)])
def test_cut_batches_match_the_oracle(F, eng, O, small_chunks, logn, bits, n_poly, split, pingpong):
    N = 1 << logn
    eng.set_option("ntt_split", split)
    eng.set_option("ntt_pingpong", pingpong)
    qs = F.create_moduli(N, bits)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 7 + n_poly)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    data[n_poly - 1, 0, :] = qs[0] - 1              # extreme values in the ragged last sub-batch
    d = eng.upload(data)
    t.forward(d, n_poly=n_poly)
    got = d.download().reshape(data.shape)
    assert (got == _oracle_forward(O, data, qs, logn)).all()
    t.inverse(d, n_poly=n_poly)
    assert (d.download().reshape(data.shape) == data).all()


def test_limb_window_and_repeated_calls_reuse_the_side_stream(F, eng, O, small_chunks):
    logn, N, n_poly = 13, 1 << 13, 40
    qs = F.create_moduli(N, [50] * 4)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(3)
    start, limbs = 1, 2
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[start:start + limbs]]) for _ in range(n_poly)])
    want = _oracle_forward(O, data, qs[start:start + limbs], logn)
    for nt in (-1, 0, 1):                            # the scratch and the fork / join events are reused call after call;
        eng.set_option("ntt_stream", nt)             # non-temporal accesses on the pieces' external side: by default / never / always
        d = eng.upload(data)
        t.forward(d, n_poly=n_poly, limbs=limbs, start=start)
        assert (d.download().reshape(data.shape) == want).all()
        t.inverse(d, n_poly=n_poly, limbs=limbs, start=start)
        assert (d.download().reshape(data.shape) == data).all()


@pytest.mark.parametrize("pingpong", [-1, 1])
@pytest.mark.parametrize("logn,bits", [(13, [50, 61]), (16, [50])])
def test_checked_transforms_cut_the_same_way(F, eng, O, small_chunks, logn, bits, pingpong):
    """The ABFT transforms (whole-transform and per-phase checks) run as the same sub-batches: clean flags, same output;
    a corrupted word (the one-shot hooks transform the batch as one launch pair) is still flagged in the right unit."""
    from fhe_reliability_gpu_amd._lib import check, lib
    N, n_poly = 1 << logn, 21 if logn == 13 else 5
    eng.set_option("ntt_pingpong", pingpong)         # 1: the checked launches hand over through the scratch too
    qs = F.create_moduli(N, bits)
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(logn)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    want = _oracle_forward(O, data, qs, logn)
    d = eng.upload(data)
    flags = ab.forward_checked(d, n_poly=n_poly)
    assert not flags.any() and (d.download().reshape(data.shape) == want).all()
    d = eng.upload(data)
    flags3 = ab.forward_checked_phases(d, n_poly=n_poly)
    assert not flags3.any() and (d.download().reshape(data.shape) == want).all()
    unit = n_poly * len(qs) - 2
    d = eng.upload(data)
    check(lib.fhe_ctx_inject_fault(eng._h, unit * N + 77, 9))
    flags = ab.forward_checked(d, n_poly=n_poly)
    assert flags.tolist() == [1 if u == unit else 0 for u in range(n_poly * len(qs))]


def test_capture_keeps_the_sub_batches_on_the_capturing_stream(F, eng, small_chunks):
    """Inside a stream capture the side stream is not used (no fork out of the capture): the graph replays to the same words."""
    import torch
    from fhe_reliability_gpu_amd._lib import check, lib
    logn, N, n_poly = 13, 1 << 13, 48
    qs = F.create_moduli(N, [50])
    t = eng.tables(logn, qs)
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    src = torch.randint(0, qs[0], (n_poly, N), generator=g, device="cuda", dtype=torch.int64)
    buf = src.clone()

    def call(s):
        check(lib.fhe_ntt_forward_batch(eng._h, C.c_void_p(buf.data_ptr()), t._h, n_poly, 1, 0, C.c_void_p(s.cuda_stream)))

    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    call(s)
    torch.cuda.synchronize()
    want = buf.clone()
    buf.copy_(src)
    graph, cap = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=cap):
        call(cap)
    buf.copy_(src)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(buf, want)


def test_default_policy_on_a_batch_that_streams_from_hbm(F, eng, O):
    """Library defaults on 208 MiB (416 polynomials of N = 2^16: sub-batches of 96 MiB, the last one ragged): equal to the
    same call made as one launch pair, and to the oracle on a polynomial of each sub-batch."""
    import torch
    from fhe_reliability_gpu_amd._lib import check, lib
    logn, N, n_poly = 16, 1 << 16, 416
    qs = F.create_moduli(N, [50])
    t = eng.tables(logn, qs)
    g = torch.Generator(device="cuda")
    g.manual_seed(8)
    src = torch.randint(0, qs[0], (n_poly, N), generator=g, device="cuda", dtype=torch.int64)
    a, b = src.clone(), src.clone()
    torch.cuda.synchronize()
    check(lib.fhe_ntt_forward_batch(eng._h, C.c_void_p(a.data_ptr()), t._h, n_poly, 1, 0, None))
    eng.sync()
    eng.set_option("ntt_chunk_mib", 0)
    try:
        check(lib.fhe_ntt_forward_batch(eng._h, C.c_void_p(b.data_ptr()), t._h, n_poly, 1, 0, None))
        eng.sync()
    finally:
        eng.set_option("ntt_chunk_mib", 96)
    assert torch.equal(a, b)
    rp = O.root_powers(qs[0], logn)
    for p in (0, 191, 192, 383, 384, 415):
        assert (a[p].cpu().numpy().view(np.uint64) == O.nwt_forward(src[p].cpu().numpy().view(np.uint64), qs[0], rp)).all()
    check(lib.fhe_ntt_inverse_batch(eng._h, C.c_void_p(a.data_ptr()), t._h, n_poly, 1, 0, None))
    eng.sync()
    assert torch.equal(a, src)
    eng.check()


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("logn,bits,n_poly", [(13, [50, 61, 50, 50, 50], 7), (14, [50, 50], 9), (16, [50, 61], 3)])
def test_negacyclic_product_cut_into_pieces(F, eng, O, small_chunks, logn, bits, n_poly, split):
    """fhe_polymul of operands that (with the result) exceed the sub-batch size runs piece by piece -- limb windows x polynomial
    ranges on alternating streams: same words as the oracle's product, with the result in its own buffer, aliasing an operand, and
    for a square."""
    N = 1 << logn
    eng.set_option("ntt_split", split)
    qs = F.create_moduli(N, bits)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn + n_poly)
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    b = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    want, sq = np.empty_like(a), np.empty_like(a)
    for p in range(n_poly):
        for l, q in enumerate(qs):
            want[p, l] = O.polymul_ntt(a[p, l], b[p, l], t.psi[l], q)
            sq[p, l] = O.polymul_ntt(a[p, l], a[p, l], t.psi[l], q)
    da, db, dc = eng.upload(a), eng.upload(b), eng.upload(np.zeros_like(a))
    t.polymul(dc, da, db, n_poly=n_poly)
    assert (dc.download().reshape(a.shape) == want).all()
    da, db = eng.upload(a), eng.upload(b)
    t.polymul(da, da, db, n_poly=n_poly)                    # the result aliases the first operand
    assert (da.download().reshape(a.shape) == want).all()
    da = eng.upload(a)
    t.polymul(dc, da, da, n_poly=n_poly)                    # squaring: both factors one buffer
    assert (dc.download().reshape(a.shape) == sq).all()


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("logn,n_vec", [(13, 40), (14, 21), (16, 7)])
def test_natural_order_transforms_cut_the_same_way(F, eng, O, small_chunks, logn, n_vec, split):
    """The batched four-step / cyclic transform (two launches through a hand-off buffer) of a batch past the sub-batch size: pieces on
    alternating streams, each stream's own scratch as the hand-off; forward against the oracle's cyclic transform, then the inverse back."""
    from fhe_reliability_gpu_amd._lib import check, lib
    eng.set_option("ntt_split", split)
    mod, g = 998244353, 3
    N = 1 << logn
    rng = np.random.default_rng(logn + n_vec)
    a = rng.integers(0, mod, (n_vec, N), dtype=np.uint64)
    got = np.array(F.four_step_ntt(a, N, mod, g, n1=1 << (logn // 2)), dtype=np.uint64)
    for v in range(n_vec):
        assert (got[v] == O.ntt_cyclic(a[v], mod, g)).all()
    d, s = eng.upload(got), eng.alloc(got.size)
    check(lib.fhe_ntt_cyclic(eng._h, d.ptr, s.ptr, logn, n_vec, mod, g, 0, 1, None))
    assert (d.download().reshape(n_vec, N) == a).all()


@pytest.mark.parametrize("logn", [5, 8, 10, 11, 12])
def test_single_launch_sizes_with_non_temporal_accesses(F, eng, O, logn):
    """Batches of the single-launch sizes that stream from HBM run the same kernels with non-temporal loads and stores
    ("ntt_stream" 1 forces them on a small batch): same words as the oracle, both arithmetic paths, forward and inverse."""
    N, n_poly = 1 << logn, 5
    qs = F.create_moduli(N, [50, 61])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    data[0, 0, :] = qs[0] - 1
    eng.set_option("ntt_stream", 1)
    try:
        d = eng.upload(data)
        t.forward(d, n_poly=n_poly)
        assert (d.download().reshape(data.shape) == _oracle_forward(O, data, qs, logn)).all()
        t.inverse(d, n_poly=n_poly)
        assert (d.download().reshape(data.shape) == data).all()
    finally:
        eng.set_option("ntt_stream", -1)
    eng.check()


@pytest.mark.parametrize("nt", [-1, 1])
@pytest.mark.parametrize("logn,bits", [(13, [50, 61]), (16, [50]), (17, [61])])
def test_the_two_launches_one_at_a_time(F, eng, O, logn, bits, nt):
    """"ntt_only_pass" 0 then 1 (bench.py times the two kernels of the headline this way, with the non-temporal variants): first
    launch, second launch = the whole transform, forward and inverse."""
    N, n_poly = 1 << logn, 2
    qs = F.create_moduli(N, bits)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    eng.set_option("ntt_stream", nt)
    try:
        d = eng.upload(data)
        for which in (0, 1):
            eng.set_option("ntt_only_pass", which)
            t.forward(d, n_poly=n_poly)
        assert (d.download().reshape(data.shape) == _oracle_forward(O, data, qs, logn)).all()
        for which in (0, 1):
            eng.set_option("ntt_only_pass", which)
            t.inverse(d, n_poly=n_poly)
        assert (d.download().reshape(data.shape) == data).all()
    finally:
        eng.set_option("ntt_only_pass", -1)
        eng.set_option("ntt_stream", -1)
    eng.check()


@pytest.mark.parametrize("split", [1, 0])
def test_two_caller_streams_free_running(F, eng, split):
    """Two caller streams transform their halves of a 1.2 GiB mixed-path batch forward and back, call after call with no
    synchronisation in between (each with its own side stream and scratch when the split is on): the round trips return the input
    word for word -- no piece reads a scratch another piece still owns."""
    import torch
    from fhe_reliability_gpu_amd._lib import check, lib
    N, polys = 1 << 16, 800               # per call: a 400 MiB run of two FP64 limbs and a 200 MiB run of one 61-bit limb, both cut
    qs = F.create_moduli(N, [50, 50, 61])
    t = eng.tables(16, qs)
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    data = torch.empty((polys, 3, N), dtype=torch.int64, device="cuda")
    for l, q in enumerate(qs):
        data[:, l, :] = torch.randint(0, q, (polys, N), generator=g, device="cuda", dtype=torch.int64)
    ref = data.clone()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    half = polys // 2
    torch.cuda.synchronize()
    eng.set_option("ntt_split", split)
    try:
        for _ in range(4):
            for i, s in enumerate(streams):
                ptr, sp = C.c_void_p(data.data_ptr() + i * half * 3 * N * 8), C.c_void_p(s.cuda_stream)
                check(lib.fhe_ntt_forward_batch(eng._h, ptr, t._h, half, 3, 0, sp))
                check(lib.fhe_ntt_inverse_batch(eng._h, ptr, t._h, half, 3, 0, sp))
        torch.cuda.synchronize()
    finally:
        eng.set_option("ntt_split", -1)
    assert torch.equal(data, ref)
    eng.check()
