"""On-device ABFT detector: the reference's weighted-checksum ECC
(rfhe_framewk/src/negaclic_ntt.py:130-149) around the GPU forward NTT, exercised with a bit flip
injected between the two launches of the transform."""
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


def _ref_weights(n, logn):
    p = 1 << (logn // 2)
    return [(i % p + 1) + (i // p + 1) for i in range(n)]


@pytest.mark.parametrize("logn,bits", [(10, 50), (14, 50), (16, 50), (14, 61)])
def test_checksums_agree_with_reference_ecc(F, eng, logn, bits):
    from oracle import pyport as P
    from oracle import cport as O
    N = 1 << logn
    qs = F.create_moduli(N, [bits, bits])
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(logn)
    a = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])
    d = eng.upload(a)
    cin = ab.checksum(d, 0)
    w = _ref_weights(N, logn)
    for l, q in enumerate(qs):
        assert int(cin[l]) == sum(wi * int(x) for wi, x in zip(w, a[l])) % q        # checksum (negaclic_ntt.py:143)
    t.forward(d)
    cout = ab.checksum(d, 1)
    assert (cin == cout).all()                                                        # == checksum_hat (:144-145)
    if logn == 10:
        # w_hat as the reference computes it (:133-138), read at bit-reversed indices, gives the same output-side sum
        q, psi = qs[0], t.psi[0]
        w_pre = [wi * pow(pow(psi, -1, q), i, q) % q for i, wi in enumerate(w)]
        w_hat = P.intt_nthroot(w_pre, pow(psi, 2, q), q)
        a_hat = d.download()[0]
        idx = [P.bit_reverse(j, logn) for j in range(N)]
        assert sum(w_hat[idx[j]] * int(a_hat[j]) for j in range(N)) % q == int(cin[0])


@pytest.mark.parametrize("bits", [50, 61])
def test_detector_flags_only_the_limb_hit_in_flight(F, eng, bits):
    from fhe_reliability_gpu_amd._lib import check, lib
    logn, N, limbs, polys = 14, 1 << 14, 3, 4
    qs = F.create_moduli(N, [bits] * limbs)
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(99)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(polys)])
    # clean run: nothing flagged
    d = eng.upload(data)
    assert not ab.forward_checked(d, n_poly=polys).any()
    clean = d.download()
    # a flip between the two launches of the transform: exactly that limb-polynomial is flagged
    for unit, word, bit in ((5, 1234, 7), (0, 0, 30), (11, N - 1, 3)):
        d = eng.upload(data)
        check(lib.fhe_ctx_inject_fault(eng._h, unit * N + word, bit))
        flags = ab.forward_checked(d, n_poly=polys)
        assert flags.tolist() == [1 if u == unit else 0 for u in range(polys * limbs)], (unit, flags)
        out = d.download().reshape(polys * limbs, N)
        bad = (out != clean.reshape(polys * limbs, N)).any(axis=1)
        assert bad.tolist() == [u == unit for u in range(polys * limbs)]
    # a fault already present in the INPUT is not a transform fault: consistent checksums, no flag
    faulty = data.copy()
    faulty[1, 2, 77] ^= np.uint64(1 << 5)
    d = eng.upload(faulty)
    assert not ab.forward_checked(d, n_poly=polys).any()


@pytest.mark.parametrize("logn", [4, 5, 8, 11, 12, 13, 15, 16, 17])
def test_checked_transform_every_plan_shape(F, eng, logn):
    """The checksums ride on the transform's passes (2^5 and up; tiny sizes use separate reductions): same output
    as the unchecked transform, no flag on a clean run, mixed arithmetic paths and a limb window in one call."""
    from oracle import cport as O
    N = 1 << logn
    qs = F.create_moduli(N, [50, 61, 61, 50])
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(logn)
    start, limbs, polys = 1, 3, 2
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[start:start + limbs]]) for _ in range(polys)])
    data[1, 0, 3] = np.uint64(2**64 - 5)                      # an out-of-range word: reduced modulo its prime on both sides
    d = eng.upload(data)
    flags = ab.forward_checked(d, n_poly=polys, limbs=limbs, start=start)
    assert not flags.any()
    got = d.download()
    for p in range(polys):
        for l in range(limbs):
            q = qs[start + l]
            assert (got[p, l] == O.nwt_forward(data[p, l] % np.uint64(q), q, O.root_powers(q, logn))).all()


def test_checked_transform_fault_at_full_size(F, eng):
    # N = 2^16: a flip between the column pass and the row pass of one limb-polynomial out of 6
    from fhe_reliability_gpu_amd._lib import check, lib
    logn, N, limbs, polys = 16, 1 << 16, 2, 3
    qs = F.create_moduli(N, [50, 61])
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(5)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(polys)])
    for unit in (0, 3, 5):
        d = eng.upload(data)
        check(lib.fhe_ctx_inject_fault(eng._h, unit * N + 4242, 11))
        flags = ab.forward_checked(d, n_poly=polys)
        assert flags.tolist() == [1 if u == unit else 0 for u in range(polys * limbs)]


# ------------------------------------------------------------------ per-phase detector
def _tiles(logn):
    """workgroups per limb-polynomial of the column pass / row pass (16-column tiles; 4096-point row tiles)"""
    pc = {13: 5, 14: 6, 15: 7, 16: 8, 17: 8}[logn]
    return (1 << (logn - pc)) // 16, (1 << logn) // 4096


@pytest.mark.parametrize("logn,bits", [(14, 50), (16, 50), (14, 61), (17, 50)])
def test_per_phase_detector_localises_the_fault(F, eng, logn, bits):
    """Where the reference checks its four-step flow (batch_check of the column transforms, check_inter, batch_check of the row
    transforms: rfhe_framewk/src/ntt_test/relia_ntt_sim.cpp:235-292,331-355) the engine has three flags per limb-polynomial.
    A clean run raises none; a flip inside the column pass, between the launches, or inside the row pass raises exactly the
    flag of that phase on exactly the limb-polynomial it hit; a fault already in the input raises nothing."""
    from fhe_reliability_gpu_amd._lib import check, lib
    from oracle import cport as O
    N, limbs, polys = 1 << logn, 2, 3
    qs = F.create_moduli(N, [bits] * limbs)
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(logn + bits)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(polys)])
    d = eng.upload(data)
    flags = ab.forward_checked_phases(d, n_poly=polys)
    assert flags.shape == (polys * limbs, 3) and not flags.any()
    clean = d.download()
    for p in range(polys):
        for l, q in enumerate(qs):
            assert (clean[p, l] == O.nwt_forward(data[p, l], q, O.root_powers(q, logn))).all()
    tc, tr = _tiles(logn)

    def unit_of(block, tiles):          # launch order is limb-major; flags are in [poly][limb] order
        u = block // tiles
        return (u % polys) * limbs + u // polys

    # (phase flag, how the fault is injected, the unit it lands in)
    cases = [(1, ("between", 4 * N + 77, 9), 4), (1, ("between", 0, 40), 0)]
    for blk in (0, 2 * tc + 1, polys * limbs * tc - 1):
        cases.append((0, ("pass", 0, blk, 5, 30), unit_of(blk, tc)))
    for blk in (1, 3 * tr, polys * limbs * tr - 1):
        cases.append((2, ("pass", 1, blk, 37, 33), unit_of(blk, tr)))
    for phase, how, unit in cases:
        d = eng.upload(data)
        if how[0] == "between":
            check(lib.fhe_ctx_inject_fault(eng._h, how[1], how[2]))
        else:
            check(lib.fhe_ctx_inject_fault_in_pass(eng._h, how[1], how[2], how[3], how[4]))
        flags = ab.forward_checked_phases(d, n_poly=polys)
        want = np.zeros((polys * limbs, 3), dtype=np.uint32)
        want[unit, phase] = 1
        assert (flags == want).all(), (how, unit, flags.tolist())
        out = d.download().reshape(polys * limbs, N)
        bad = (out != clean.reshape(polys * limbs, N)).any(axis=1)
        assert bad.tolist() == [u == unit for u in range(polys * limbs)], how
    # the hooks are one-shot: the next run is clean again
    d = eng.upload(data)
    assert not ab.forward_checked_phases(d, n_poly=polys).any()
    # an input fault is not a transform fault
    faulty = data.copy()
    faulty[2, 1, 123] ^= np.uint64(1 << 7)
    assert not ab.forward_checked_phases(eng.upload(faulty), n_poly=polys).any()


def test_per_phase_detector_mixed_paths_and_window(F, eng):
    from oracle import cport as O
    logn, N = 13, 1 << 13
    qs = F.create_moduli(N, [50, 61, 61, 50, 50])
    t = eng.tables(logn, qs)
    ab = F.Abft(eng, t)
    rng = np.random.default_rng(3)
    start, limbs, polys = 1, 4, 2
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[start:start + limbs]]) for _ in range(polys)])
    data[0, 1, 9] = np.uint64(2**64 - 3)                      # out-of-range word: reduced on load, consistent on every side
    d = eng.upload(data)
    assert not ab.forward_checked_phases(d, n_poly=polys, limbs=limbs, start=start).any()
    got = d.download()
    for p in range(polys):
        for l in range(limbs):
            q = qs[start + l]
            assert (got[p, l] == O.nwt_forward(data[p, l] % np.uint64(q), q, O.root_powers(q, logn))).all()
    # single-launch sizes have one phase: the call says so instead of inventing flags
    t12 = eng.tables(12, F.create_moduli(4096, [50]))
    with pytest.raises(Exception, match="per-phase"):
        F.Abft(eng, t12).forward_checked_phases(eng.upload(np.zeros(4096, dtype=np.uint64)))
