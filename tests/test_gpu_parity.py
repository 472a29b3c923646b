"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden
vectors generated from the reference's Python.  Bit-exact: every comparison is ==.
Run on the MI355X box with ``pytest -m gpu``."""
import hashlib
import random

import numpy as np
import pytest

from conftest import load_golden, sha_u64

pytestmark = pytest.mark.gpu

Q61 = 2305843009211596801
PHANTOM_PRIMES = [1125899903107073, 1125899903500289, 1125899903795201,
                  1125899903827969, 1125899903991809, 1125899904679937]


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


def _rand_limbs(rng, qs, N, n_poly=1):
    return np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])


# ------------------------------------------------------------------ a2 / a3
@pytest.fixture(params=[1, 0], ids=["fused", "twopass"])
def mode(request, eng):
    eng.set_option("ntt_mode", request.param)
    yield request.param
    eng.check()
    eng.set_option("ntt_mode", 0)


@pytest.mark.parametrize("logn", list(range(1, 19)))
@pytest.mark.parametrize("bits", [50, 61, 30])
def test_forward_inverse_match_oracle(F, eng, O, logn, bits, mode):
    if bits == 30 and logn > 14:
        pytest.skip("small-modulus FP64 case covered at the smaller sizes")
    if mode == 1 and not 13 <= logn <= 17:
        pytest.skip("the fused launch only exists for 2^13..2^17")
    N = 1 << logn
    limbs = 3 if logn <= 14 else (2 if logn <= 16 else 1)
    n_poly = 2 if logn <= 13 else 1
    qs = F.create_moduli(N, [bits] * limbs)
    assert qs == O.gen_primes(N, bits, limbs)            # a10: same primes as the oracle's rule
    t = eng.tables(logn, qs)
    assert t.paths == [0 if bits <= 50 else 1] * limbs
    assert t.psi == [O.min_primitive_root(q, 2 * N) for q in qs]
    rps = [O.root_powers(q, logn) for q in qs]
    rng = np.random.default_rng(1000 * logn + bits)
    data = _rand_limbs(rng, qs, N, n_poly)
    data[0, 0, :] = qs[0] - 1                            # extreme values for the lazy ranges: the all-sums register doubles at every stage
    if limbs >= 2:
        data[0, 1, 0::2], data[0, 1, 1::2] = qs[1] - 1, 0                              # ... and the difference branches carry q - 1
    if n_poly >= 2:
        data[1, 0, :] = rng.integers(0, 2, N, dtype=np.uint64) * np.uint64(qs[0] - 1)   # random corners of [0, q)^N
    d = eng.upload(data)
    t.forward(d, n_poly=n_poly)
    fwd = d.download()
    for p in range(n_poly):
        for l in range(limbs):
            assert (fwd[p, l] == O.nwt_forward(data[p, l], qs[l], rps[l])).all(), (p, l)
    t.inverse(d, n_poly=n_poly)
    assert (d.download() == data).all()
    # inverse on its own input (not a round trip)
    d2 = eng.upload(data)
    t.inverse(d2, n_poly=n_poly)
    inv = d2.download()
    for l in range(limbs):
        assert (inv[0, l] == O.nwt_inverse(data[0, l], qs[l], rps[l])).all()


@pytest.mark.parametrize("logn", [13, 14])
def test_resident_and_two_launch_forms_agree(F, eng, O, logn):
    """"ntt_resident" 1 runs 2^13 / 2^14 as one LDS-resident pass (the limb fits a CU's 160 KiB LDS); the default is the
    two-launch form the larger sizes use.  Both against the oracle, mixed arithmetic paths, out-of-range words."""
    N = 1 << logn
    qs = F.create_moduli(N, [50, 61, 50])
    t = eng.tables(logn, qs)
    rps = [O.root_powers(q, logn) for q in qs]
    rng = np.random.default_rng(logn)
    data = _rand_limbs(rng, qs, N, 3)
    data[2, 1, 5] = np.uint64(2**64 - 1)
    red = data.copy()
    red[2, 1, 5] %= np.uint64(qs[1])
    try:
        for resident in (1, 0):
            eng.set_option("ntt_resident", resident)
            d = eng.upload(data)
            t.forward(d, n_poly=3)
            fwd = d.download()
            for p in range(3):
                for l, q in enumerate(qs):
                    assert (fwd[p, l] == O.nwt_forward(red[p, l], q, rps[l])).all(), (resident, p, l)
            t.inverse(d, n_poly=3)
            assert (d.download() == red).all()
    finally:
        eng.set_option("ntt_resident", 0)


def test_kat2_phantom_prime_ring(F, eng):
    # the dotprod_test ring, first logged Phantom prime
    q, logn = PHANTOM_PRIMES[0], 14
    t = eng.tables(logn, [q])
    assert t.psi == [32853495844]
    a = np.array([(i * i + 1) % q for i in range(1 << logn)], dtype=np.uint64)
    d = eng.upload(a)
    t.forward(d)
    out = d.download()
    assert [int(x) for x in out[:4]] == [1037891225979181, 928784233542420, 626683383818997, 933202470215772]
    assert sha_u64(out) == "79d3fd044499dd0c43406c0e0d282a73b2aba27be4fdb7efba6342e00d9a3a4d"


def test_mixed_paths_and_limb_window(F, eng, O):
    # 50-bit and 61-bit limbs in one table set, transformed through a start_idx window
    logn, N = 12, 4096
    qs = F.create_moduli(N, [50, 61, 50, 50, 61, 61])
    t = eng.tables(logn, qs)
    assert t.paths == [0, 1, 0, 0, 1, 1]
    rng = np.random.default_rng(5)
    start, limbs, n_poly = 1, 4, 3
    data = _rand_limbs(rng, qs[start:start + limbs], N, n_poly)
    d = eng.upload(data)
    t.forward(d, limbs=limbs, start=start, n_poly=n_poly)
    got = d.download()
    for p in range(n_poly):
        for l in range(limbs):
            q = qs[start + l]
            assert (got[p, l] == O.nwt_forward(data[p, l], q, O.root_powers(q, logn))).all()
    t.inverse(d, limbs=limbs, start=start, n_poly=n_poly)
    assert (d.download() == data).all()


@pytest.mark.parametrize("bits", [50, 61])
def test_out_of_range_words(F, eng, O, bits):
    # the fault-injection harness flips arbitrary bits (reliability_test/ntt_test.cu:104-135)
    logn, N = 13, 8192
    q = F.create_moduli(N, [bits])[0]
    t = eng.tables(logn, [q])
    rng = np.random.default_rng(9)
    a = rng.integers(0, q, N, dtype=np.uint64)
    a[3] ^= np.uint64(1 << 63)
    a[77] ^= np.uint64(1 << 52)
    a[500] = np.uint64(q)
    a[501] = np.uint64(2**64 - 1)
    d = eng.upload(a)
    t.forward(d)
    assert (d.download() == O.nwt_forward(a, q, O.root_powers(q, logn))).all()


def test_tables_from_caller_roots(F, eng, O):
    # DNTTTable::set with host tables (ntt_test.cu:61-69), forcing each arithmetic path
    logn, N = 10, 1024
    q = PHANTOM_PRIMES[1]
    psi = O.min_primitive_root(q, 2 * N)
    psi = pow(psi, 3, q)                      # a non-minimal root: still a valid table
    rp = O.root_powers(q, logn, psi)
    rng = np.random.default_rng(2)
    a = rng.integers(0, q, N, dtype=np.uint64)
    want = O.nwt_forward(a, q, rp)
    for path in (0, 1):
        t = eng.tables_from_roots(logn, [q], rp, force_path=path)
        assert t.paths == [path] and t.psi == [psi]
        d = eng.upload(a)
        t.forward(d)
        assert (d.download() == want).all()
        t.inverse(d)
        assert (d.download() == a).all()


# ------------------------------------------------------------- golden vectors
def test_golden_negacyclic_natural_order(F):
    g = load_golden("negacyclic.json")
    for case in g["cases"]:
        n, q, psi = case["n"], case["q"], case["psi"]
        random.seed(case["seed"])
        a = [random.randrange(q) for _ in range(n)]
        fwd = F.negacyclic_ntt(a, psi, q)
        assert sha_u64(fwd) == case["sha256_fwd"]
        assert fwd[:8] == case["head"] and fwd[-8:] == case["tail"]
        assert F.negacyclic_intt(fwd, psi, q) == a
    for pm in g["polymul"]:
        assert F.poly_mul_negacyclic_ntt(pm["a"], pm["b"], pm["psi"], pm["q"]) == pm["c"]


def test_golden_cyclic(F):
    g = load_golden("cyclic_ntt.json")
    d = g["demo"]                              # composite modulus 15728641 (motivation/ntt.py:35-43)
    assert F.ntt(d["a"], d["mod"], d["root"]) == d["A"]
    for case in g["cases"]:
        if case["lg"] < 1:
            continue
        random.seed(case["seed"])
        a = [random.randrange(case["mod"]) for _ in range(1 << case["lg"])]
        out = F.ntt(a, case["mod"], case["root"])
        assert sha_u64(out) == case["sha256"]
        assert out[:8] == case["head"] and out[-8:] == case["tail"]
    t = g["bsgs_twin"]
    assert F.ntt(t["a"], t["mod"], t["root"]) == t["fwd"]
    assert F.intt(t["fwd"], t["mod"], t["root"]) == t["a"]
    r = g["rfhe_twin"]
    assert F.ntt(r["a"], r["mod"], r["root"]) == r["fwd"]
    assert F.intt(r["a"], r["mod"], r["root"]) == r["inv"]
    r = g["nthroot"]
    assert F.ntt_nthroot(r["a"], r["root"], r["mod"]) == r["fwd"]
    assert F.intt_nthroot(r["a"], r["root"], r["mod"]) == r["inv"]


def test_golden_four_step(F, O):
    g = load_golden("four_step.json")
    for case in g["cases"]:
        assert F.four_step_ntt(case["a"], case["N"], g["mod"], g["g"]) == case["y"]
    # n1 != n2 (config 4's 2^17 = 512 x 256 shape, scaled down) equals the direct transform
    random.seed(3)
    a = [random.randrange(g["mod"]) for _ in range(2048)]
    want = [int(x) for x in O.ntt_cyclic(a, g["mod"], g["g"])]
    assert F.four_step_ntt(a, 2048, g["mod"], g["g"], n1=64) == want
    assert F.four_step_ntt(a, 2048, g["mod"], g["g"], n1=32) == want


def test_four_step_2_16_equals_cyclic(F, O):
    mod, g, N = 998244353, 3, 1 << 16
    rng = np.random.default_rng(16)
    a = rng.integers(0, mod, N, dtype=np.uint64)
    want = O.ntt_cyclic(a, mod, g)
    assert (np.array(F.four_step_ntt(a, N, mod, g), dtype=np.uint64) == want).all()


def test_golden_base_conversion(F):
    g = load_golden("baseconv.json")
    for key in ("exact", "exact50"):
        e = g[key]
        assert F.base_conv_fixed(e["res"], e["mod_in"], e["mod_out"]) == e["out"]
    for key in ("fast", "fast31"):
        f = g[key]
        assert F.bConv(f["res"], f["mod_in"], f["mod_out"]) == f["out"]


def test_golden_bsgs(F):
    g = load_golden("bsgs.json")
    s = g["small"]
    assert F.diag_block_hadamard_matvec(s["M"], s["v"]).tolist() == s["y"]
    np.random.seed(g["numpy_seed"])           # motivation/bsgs.py:89-101
    M = [np.random.randint(0, g["mod"], size=g["block_size"]) for _ in range(g["k"])]
    v = np.random.randint(0, g["mod"], size=g["block_size"] * g["k"])
    y = F.diag_block_hadamard_matvec(np.array(M), v)
    assert y[:32].tolist() == g["y_head"]
    assert sha_u64(y.astype(np.uint64)) == g["sha256_y"]


# ------------------------------------------------------------------- a4 / a5
def test_modmul_and_accumulate(F, eng, O):
    logn, N = 12, 4096
    qs = F.create_moduli(N, [50, 61, 40])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(4)
    a, b, c = (_rand_limbs(rng, qs, N, 2) for _ in range(3))
    a[0, 0, :7] = np.uint64(2**64 - 1)         # unreduced operands are reduced first
    da, db, dc = eng.upload(a), eng.upload(b), eng.upload(c)
    out = eng.alloc(a.size)
    out.shape = a.shape
    t.modmul(out, da, db, n_poly=2)
    got = out.download()
    t.modmul(dc, da, db, n_poly=2, acc=True)
    acc = dc.download()
    for p in range(2):
        for l, q in enumerate(qs):
            assert (got[p, l] == O.modmul(a[p, l], b[p, l], q)).all()
            assert (acc[p, l] == O.modmul_acc(c[p, l], a[p, l], b[p, l], q)).all()


@pytest.mark.parametrize("bits", [50, 61])
def test_polymul_matches_schoolbook(F, eng, O, bits):
    logn, N = 8, 256
    qs = F.create_moduli(N, [bits, bits])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(8)
    a, b = _rand_limbs(rng, qs, N), _rand_limbs(rng, qs, N)
    da, db = eng.upload(a), eng.upload(b)
    t.polymul(da, da, db)
    got = da.download()
    for l, q in enumerate(qs):
        assert (got[0, l] == O.polymul_naive(a[0, l], b[0, l], q)).all()


@pytest.mark.parametrize("logn", list(range(1, 18)))
def test_polymul_every_size_both_paths_and_aliasing(F, eng, O, logn):
    """fhe_polymul takes the fused route (column passes + one middle launch) for 2^5..2^20 and the plain
    NTT, NTT, modmul, INTT sequence below that; both against the oracle's NTT-based product (itself
    pinned to the schoolbook product and the reference's golden vectors)."""
    N = 1 << logn
    qs = F.create_moduli(N, [50, 61, 50])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(100 + logn)
    n_poly = 2 if logn <= 14 else 1
    a, b = _rand_limbs(rng, qs, N, n_poly), _rand_limbs(rng, qs, N, n_poly)
    want = np.empty_like(a)
    sq = np.empty_like(a)
    for p in range(n_poly):
        for l, q in enumerate(qs):
            want[p, l] = O.polymul_ntt(a[p, l], b[p, l], t.psi[l], q)
            sq[p, l] = O.polymul_ntt(a[p, l], a[p, l], t.psi[l], q)
    da, db, dc = eng.upload(a), eng.upload(b), eng.upload(np.zeros_like(a))
    t.polymul(dc, da, db, n_poly=n_poly)                    # separate output
    assert (dc.download() == want).all()
    da, db = eng.upload(a), eng.upload(b)
    t.polymul(da, da, db, n_poly=n_poly)                    # c aliases a
    assert (da.download() == want).all()
    da, db = eng.upload(a), eng.upload(b)
    t.polymul(db, da, db, n_poly=n_poly)                    # c aliases b
    assert (db.download() == want).all()
    da = eng.upload(a)
    t.polymul(dc, da, da, n_poly=n_poly)                    # squaring, both factors one buffer
    assert (dc.download() == sq).all()


def test_polymul_out_of_range_words_and_limb_window(F, eng, O):
    # arbitrary 64-bit input words are taken modulo their limb's modulus, as everywhere in the engine
    logn, N = 13, 1 << 13
    qs = F.create_moduli(N, [50, 50, 61, 61])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(77)
    start, limbs = 1, 3
    a = rng.integers(0, 2**64, size=(1, limbs, N), dtype=np.uint64)
    b = rng.integers(0, 2**64, size=(1, limbs, N), dtype=np.uint64)
    da, db = eng.upload(a), eng.upload(b)
    t.polymul(da, da, db, limbs=limbs, start=start)
    got = da.download()
    for l in range(limbs):
        q = qs[start + l]
        assert (got[0, l] == O.polymul_ntt(a[0, l] % np.uint64(q), b[0, l] % np.uint64(q), t.psi[start + l], q)).all()


# ----------------------------------------------------------------------- a8
def test_base_conversion_against_oracle_large(F, eng, O):
    N = 1 << 12
    mi = F.create_moduli(N, [50] * 6)
    mo = F.create_moduli(N, [61] * 3) + [1073741827]
    rng = np.random.default_rng(12)
    res = np.stack([rng.integers(0, p, N, dtype=np.uint64) for p in mi])
    assert (np.array(F.base_conv_fixed(res, mi, mo), dtype=np.uint64) == O.baseconv_exact(res, mi, mo)).all()
    assert (np.array(F.bConv(res, mi, mo), dtype=np.uint64) == O.bconv_fast(res, mi, mo)).all()


def test_crt_garner_matches_oracle(F, O):
    rng = np.random.default_rng(13)
    for mod in ([1048583, 1048589, 1048601, 1048609], [1073741827, 1073741831, 1073741833, 1073741839],
                PHANTOM_PRIMES[:5]):   # the last one exercises the 128-bit wrap-around of the reference kernel
        N = 1000
        res = np.stack([rng.integers(0, p, N, dtype=np.uint64) for p in mod])
        lo, hi = F.crt_garner(res, mod)
        wlo, whi = O.crt_garner(res, mod)
        assert (lo == wlo).all() and (hi == whi).all()


# ------------------------------------------------------------ fault injection
def test_flip_bit_and_error_propagation(F, eng):
    from fhe_reliability_gpu_amd._lib import check, lib
    # one flipped input symbol corrupts every output of its limb and no other limb
    # (reliability_test/data/flipimpact_ntt.csv: symbol error rate 1.0; exp_log.txt:3)
    logn, N, limbs = 12, 4096, 4
    qs = F.create_moduli(N, [50] * limbs)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(21)
    data = _rand_limbs(rng, qs, N)
    d = eng.upload(data)
    t.forward(d)
    clean = d.download()
    d2 = eng.upload(data)
    check(lib.fhe_flip_bit(eng._h, d2.ptr, 2 * N + 17, 13, None))
    assert (d2.download().reshape(-1)[2 * N + 17] == (data.reshape(-1)[2 * N + 17] ^ np.uint64(1 << 13)))
    t.forward(d2)
    faulty = d2.download()
    diff = clean != faulty
    assert diff[0, 2].all() and not diff[0, [0, 1, 3]].any()
    ber = sum(bin(int(x)).count("1") for x in (clean[0, 2] ^ faulty[0, 2])) / (64 * N)
    assert 0.36 < ber < 0.42                     # ~25/64 for 50-bit primes (flipimpact_ntt.csv)


# ---------------------------------------------------- full-size properties
@pytest.mark.parametrize("bits", [50, 61])
def test_full_size_round_trip_and_linearity(F, eng, bits):
    # BASELINE config 3 shape: N = 2^16, L = 16 limbs; size-independent properties
    logn, N, L = 16, 1 << 16, 16
    qs = F.create_moduli(N, [bits] * L)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(33)
    a, b = _rand_limbs(rng, qs, N), _rand_limbs(rng, qs, N)
    qcol = np.array(qs, dtype=np.uint64)[None, :, None]
    s = ((a.astype(object) + b.astype(object)) % qcol.astype(object)).astype(np.uint64)
    da, db, ds = eng.upload(a), eng.upload(b), eng.upload(s)
    for x in (da, db, ds):
        t.forward(x)
    fa, fb, fs = da.download(), db.download(), ds.download()
    lin = ((fa.astype(object) + fb.astype(object)) % qcol.astype(object)).astype(np.uint64)
    assert (fs == lin).all()
    assert (fa < qcol).all()
    t.inverse(da)
    assert (da.download() == a).all()
    # NTT-domain product == negacyclic product: check x * 1 and x * X (a shift with sign flip)
    one = np.zeros_like(a)
    one[:, :, 0] = 1
    shift = np.zeros_like(a)
    shift[:, :, 1] = 1
    dx, d1, dsft = eng.upload(a), eng.upload(one), eng.upload(shift)
    out = eng.alloc(a.size)
    out.shape = a.shape
    t.forward(dx)
    t.forward(d1)
    t.forward(dsft)
    t.modmul(out, dx, d1)
    t.inverse(out)
    assert (out.download() == a).all()
    t.modmul(out, dx, dsft)
    t.inverse(out)
    want = np.roll(a, 1, axis=2)
    want[:, :, 0] = (qcol[:, :, 0] - want[:, :, 0]) % qcol[:, :, 0]
    assert (out.download() == want).all()


# ------------------------------------------------------ fused hand-off under load
@pytest.mark.parametrize("bits,inverse", [(50, False), (61, False), (50, True)])
def test_fused_equals_twopass_on_large_batches(F, eng, bits, inverse):
    """The fused kernel hands a limb from its first to its second pass through one XCD's L2.
    Compare every word with the two-launch path on batches far larger than the number of
    resident workgroups, for several pipeline distances and grid sizes (uneven load, warm
    caches: repeated in-place launches)."""
    logn, N = 16, 1 << 16
    limbs, polys = 4, 96                     # 384 limb-polynomials, 192 MiB
    qs = F.create_moduli(N, [bits] * limbs)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(77)
    data = _rand_limbs(rng, qs, N, polys)
    run = (lambda d: t.inverse(d, n_poly=polys)) if inverse else (lambda d: t.forward(d, n_poly=polys))
    eng.set_option("ntt_mode", 0)
    d = eng.upload(data)
    run(d)
    run(d)                                   # two rounds: second launch starts from warm caches
    ref = d.download()
    try:
        for dist, wgs, variant, skip in ((1, 1024, 7, 0), (2, 300, 2, 0), (4, 768, 3, 0), (7, 512, 4, 0), (3, 64, 5, 0),
                                         (3, 768, 6, 0), (3, 768, 7, 0b00100101), (2, 512, 2, 0b11111110),
                                         (2, 384, 10, 0), (3, 512, 12, 0), (4, 256, 14, 0b00010000)):   # 10/12/14: fat tiles
            # skip != 0: teams that "received no workgroup" -- their limbs must come out of the fix-up launch
            eng.set_option("ntt_mode", 1)
            eng.set_option("fused_dist", dist)
            eng.set_option("fused_wgs", wgs)
            eng.set_option("fused_variant", variant)
            eng.set_option("fused_skip_teams", skip)
            d2 = eng.upload(data)
            run(d2)
            run(d2)
            eng.check()
            got = d2.download()
            assert (got == ref).all(), f"fused(dist={dist}, wgs={wgs}, variant={variant}, skip={skip}) differs in {(got != ref).sum()} words"
            d2.free()
    finally:
        eng.set_option("ntt_mode", 0)
        eng.set_option("fused_dist", 4)
        eng.set_option("fused_wgs", 768)
        eng.set_option("fused_variant", 7)
        eng.set_option("fused_skip_teams", 0)


# ------------------------------------------------------- CLI text protocol (b1)
def _run_cli(name, *args):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fhe_reliability_gpu_amd", "bin", name)
    return subprocess.run([exe, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)


def test_ntt_test_cli_protocol():
    import re
    # 4-argument form driven by test_scripts/gen_errorimpact.py:19-31 (log_dim 12, batch 1)
    r = _run_cli("ntt_test", 12, 1, 3, 1)
    assert r.returncode == 0
    assert "[2D] Flipping 3 bits across 1 symbols...\n" in r.stdout
    m = re.search(r"\[FAULT DETECTED\] Bit error: (\d+)/(\d+) = ([0-9.]+), Symbol error: (\d+)/(\d+) = ([0-9.]+)", r.stderr)
    assert m, r.stderr
    assert int(m.group(2)) == 4096 * 64 and int(m.group(4)) == int(m.group(5)) == 4096     # symbol error rate 1.0
    assert 0.33 < float(m.group(3)) < 0.45                                                  # ~25/64 (flipimpact_ntt.csv)
    assert re.search(r"ERROR! Total bitwise Hamming distance = \d+ \(bit error rate = [0-9.e-]+\)\n"
                     r"       Affected symbols = 4096/4096 \(symbol error rate = 1\)\n", r.stdout)
    # two flipped symbols at batch 32: at most two limbs change (exp_log.txt:3 shows 8192/131072)
    r = _run_cli("ntt_test", 12, 32, 1, 2)
    m = re.search(r"Symbol error: (\d+)/(\d+)", r.stderr)
    assert m and int(m.group(2)) == 131072 and int(m.group(1)) in (4096, 8192)
    # legacy 3-argument form of run_bench_test.sh:9
    r = _run_cli("ntt_test", 12, 32, 2)
    assert r.returncode == 0 and "[2D] Flipping 2 unique random bits in the data buffer...\n" in r.stdout
    assert "[FAULT DETECTED]" in r.stderr
    # usage / argument errors exit with 1 (ntt_test.cu:203-216)
    assert _run_cli("ntt_test", 12, 1).returncode == 1
    assert _run_cli("ntt_test", 12, 1, 0, 1).returncode == 1
    assert _run_cli("ntt_test", 12, 1, 65, 1).returncode == 1


def test_ntt_real_test_cli_all_correct():
    # run_real_test.sh:24 runs "./ntt_test 17 1 1" on the no-flip build and expects ALL CORRECT
    r = _run_cli("ntt_real_test", 17, 1, 1)
    assert r.returncode == 0 and r.stdout == "ALL CORRECT\n" and r.stderr == ""
    r = _run_cli("ntt_real_test", 16, 4, 1)
    assert r.stdout == "ALL CORRECT\n"


# ----------------------------------------------- dotprod_test
def test_dotprod_test_cli_matches_reference_log_format():
    """BGV encrypted dot product over the engine (include/phantom_bgv_shim.hpp).  The line layout is the one
    logged at reliability_test/data/bits1-16_num1.txt:4-37; without a fault the decrypted dot product equals
    the CPU one (dotprod_test.cu:180-181), with a flipped ciphertext bit every slot is corrupted (:134-137)."""
    import re
    r = _run_cli("dotprod_test", 0, 0)
    assert r.returncode == 0, r.stderr
    head = ("/\n| Encryption parameters :\n|   scheme: BGV\n|   poly_modulus_degree: 16384\n"
            "|   coeff_modulus size: 300 (50 + 50 + 50 + 50 + 50 + 50) bits\n\n"
            "1125899903107073 ,  1125899903500289 ,  1125899903795201 ,  1125899903827969 ,  1125899903991809 ,  1125899904679937 ,  \n\n\\\n\n"
            "Example: BGV HomMul test\nPlaintext matrix row size: 8192\nInput vector 1: \n    [ ")
    assert r.stdout.startswith(head)
    assert "Compute x * y homomorphically...\nRaw product vector: \n    [ " in r.stdout
    assert "Elementwise symbol errors: 0 / 16384\nElementwise Hamming distance (bit errors): 0\n" in r.stdout
    m = re.search(r"Decrypted dot product = (\d+)\nExpected \(CPU\)         = (\d+)\n"
                  r"Dot product bit errors \(Hamming distance\): 0\nAbsolute difference   = 0\nPercentage error      = 0%\n"
                  r"✔ Dot product matches CPU result.\n$", r.stdout)
    assert m and m.group(1) == m.group(2)
    # one flipped bit in the ciphertext (run_dotprod_simu.sh:17)
    r = _run_cli("dotprod_test", 1, 1)
    assert r.returncode == 0
    assert re.search(r"^Injected bitflip @ idx=\d+, bit=\d+$", r.stderr, re.M)
    m = re.search(r"Elementwise symbol errors: (\d+) / 16384\nElementwise Hamming distance \(bit errors\): (\d+)\n", r.stdout)
    assert m
    # ciphertexts are NTT-domain, so any flipped bit of any word spreads over every slot: the reference logs 16384 (791 x)
    # or 16383 (9 x) corrupted slots in its 800 runs, never a match (reliability_test/data/bits1-16_num1.txt)
    assert int(m.group(1)) >= 16383
    assert 120000 < int(m.group(2)) < 200000          # the reference logs ~158800 (bits1-16_num1.txt:30)
    assert "✖ MISMATCH detected!" in r.stdout
    assert _run_cli("dotprod_test", 1).returncode == 1       # usage error (dotprod_test.cu:190-194)


def test_dotprod_real_test_cli():
    """dotprod_real_test.cu: no arguments, no software flip -> the encrypted dot product matches the CPU one."""
    r = _run_cli("dotprod_real_test")
    assert r.returncode == 0, r.stderr
    assert "Injected bitflip" not in r.stderr
    assert "Elementwise symbol errors: 0 / 16384\n" in r.stdout
    assert "Percentage error      = 0 %\n" in r.stdout
    assert r.stdout.endswith("✔ Dot product matches CPU result.\n")


def test_naive_gemm_test_cli():
    """naive_gemm_test.cu:94-100: 100 encrypted dot products with the keys reused; silent timing body."""
    r = _run_cli("naive_gemm_test")
    assert r.returncode == 0, r.stderr
    assert "MISMATCH" not in r.stdout


# ------------------------------------------------------------------ edges: empty, limits, error status
def test_empty_batches_are_no_ops(F, eng):
    from fhe_reliability_gpu_amd._lib import lib
    logn, N = 12, 4096
    qs = F.create_moduli(N, [50, 50])
    t = eng.tables(logn, qs)
    data = np.arange(2 * N, dtype=np.uint64).reshape(1, 2, N) % np.uint64(qs[0])
    d = eng.upload(data)
    for fn in (lib.fhe_ntt_forward_batch, lib.fhe_ntt_inverse_batch):
        assert fn(eng._h, d.ptr, t._h, 0, 2, 0, None) == 0            # no polynomials
        assert fn(eng._h, d.ptr, t._h, 1, 0, 0, None) == 0            # no limbs
    assert lib.fhe_modmul(eng._h, d.ptr, d.ptr, d.ptr, t._h, 0, 2, 0, None) == 0
    assert lib.fhe_polymul(eng._h, d.ptr, d.ptr, d.ptr, t._h, 0, 2, 0, None) == 0
    eng.check()
    assert (d.download() == data).all()


def test_error_status_instead_of_launch(F, eng):
    """Every entry point reports bad shapes through its status (and fhe_last_error), never through a fault."""
    from fhe_reliability_gpu_amd._lib import lib
    logn, N = 10, 1024
    qs = F.create_moduli(N, [50, 50])
    t = eng.tables(logn, qs)
    d = eng.upload(np.zeros((1, 2, N), dtype=np.uint64))
    assert lib.fhe_ntt_forward_batch(eng._h, d.ptr, t._h, 1, 3, 0, None) == 1        # more limbs than the table set has
    assert lib.fhe_ntt_forward_batch(eng._h, d.ptr, t._h, 1, 2, 1, None) == 1        # window runs past the last limb
    assert lib.fhe_ntt_forward_batch(eng._h, None, t._h, 1, 2, 0, None) == 1         # null data
    assert b"" != lib.fhe_last_error()
    with pytest.raises(F.FheError):
        eng.tables(21, [qs[0]])                                                      # beyond 2^20
    with pytest.raises(F.FheError):
        eng.tables(logn, [(1 << 62) + 1])                                            # modulus too wide for the integer path
    with pytest.raises(F.FheError):
        F.BaseConv(eng, [15, 21], [qs[0]])                                           # input moduli not coprime
    with pytest.raises(F.FheError):
        F.KeySwitch(eng, t, 2, 1, 1)                                                 # L + K exceeds the table set
    eng.check()


@pytest.mark.parametrize("logn", [19, 20])
def test_largest_sizes(F, eng, O, logn):
    # the upper end of the plan table (2^20: 8 MiB per limb); one limb per arithmetic path, forward + inverse
    N = 1 << logn
    qs = F.create_moduli(N, [50, 61])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn)
    data = _rand_limbs(rng, qs, N, 1)
    d = eng.upload(data)
    t.forward(d)
    fwd = d.download()
    for l, q in enumerate(qs):
        assert (fwd[0, l] == O.nwt_forward(data[0, l], q, O.root_powers(q, logn))).all()
    t.inverse(d)
    assert (d.download() == data).all()


def test_two_host_threads_two_streams(F, eng, O):
    """One context, two host threads, each with its own stream and buffers (the library serialises only its small
    per-context maps): forward + product + inverse loops must not disturb each other."""
    import ctypes as C
    import threading
    from fhe_reliability_gpu_amd._lib import check, lib
    logn, N = 13, 1 << 13
    qs = F.create_moduli(N, [50, 61])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(3)
    errors = []

    def worker(seed):
        try:
            import torch
            s = torch.cuda.Stream()
            sp = C.c_void_p(s.cuda_stream)
            r = np.random.default_rng(seed)
            a, b = _rand_limbs(r, qs, N, 2), _rand_limbs(r, qs, N, 2)
            want = np.stack([np.stack([O.polymul_ntt(a[p, l], b[p, l], t.psi[l], q) for l, q in enumerate(qs)]) for p in range(2)])
            for _ in range(10):
                da, db, dc = eng.upload(a), eng.upload(b), eng.upload(np.zeros_like(a))
                check(lib.fhe_ntt_forward_batch(eng._h, da.ptr, t._h, 2, 2, 0, sp))
                check(lib.fhe_ntt_forward_batch(eng._h, db.ptr, t._h, 2, 2, 0, sp))
                check(lib.fhe_modmul(eng._h, dc.ptr, da.ptr, db.ptr, t._h, 2, 2, 0, sp))
                check(lib.fhe_ntt_inverse_batch(eng._h, dc.ptr, t._h, 2, 2, 0, sp))
                check(lib.fhe_sync(eng._h, sp))
                if not (dc.download() == want).all():
                    errors.append(seed)
        except Exception as e:          # noqa: BLE001 -- reported to the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(11 + i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    del rng


def test_packed_handoff_equals_plain(F, eng, O):
    """"ntt_packed": the forward 2^16 transform of FP64 limbs hands its intermediate over as 50-bit residues in 16x16 blocks
    (19 % fewer bytes on two of the four sweeps).  Same words as the plain form; limbs of the integer path in the same call
    keep the plain form; the inverse is untouched."""
    logn, N = 16, 1 << 16
    qs = F.create_moduli(N, [50, 61, 50])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(4)
    data = _rand_limbs(rng, qs, N, 5)
    data[0, 0, :] = qs[0] - 1
    data[3, 2, 17] = np.uint64(2**64 - 3)                    # out-of-range word
    out = {}
    try:
        for packed in (1, 0):
            eng.set_option("ntt_packed", packed)
            d = eng.upload(data)
            t.forward(d, n_poly=5)
            out[packed] = d.download()
            t.inverse(d, n_poly=5)
            back = d.download()
            red = data.copy()
            red[3, 2, 17] %= np.uint64(qs[2])
            assert (back == red).all()
    finally:
        eng.set_option("ntt_packed", 0)
    assert (out[0] == out[1]).all()
    for p, l in ((0, 0), (4, 2), (2, 1)):
        assert (out[1][p, l] == O.nwt_forward(data[p, l] % np.uint64(qs[l]), qs[l], O.root_powers(qs[l], logn))).all()


# ------------------------------------------------------------------ natural-order transforms as first-class launches
@pytest.mark.parametrize("n1,n2", [(512, 256), (256, 512), (256, 256), (64, 128), (4, 8), (2, 2)])
def test_four_step_unequal_factors_match_reference_flow(F, O, n1, n2):
    """four_step_ntt with n1 != n2 (BASELINE config 4's 2^17 = 512 x 256 among them) against the oracle's restatement of the
    reference's four-step flow (reliability_test/four_step_ntt_prot.py:71-109, generalised to n1 != n2) -- which itself equals
    ntt_direct."""
    mod, g = 998244353, 3
    N = n1 * n2
    rng = np.random.default_rng(n1 + 3 * n2)
    a = rng.integers(0, mod, N, dtype=np.uint64)
    want = O.four_step_ntt(a, n1, n2, mod, g)
    assert (np.array(F.four_step_ntt(a, N, mod, g, n1=n1), dtype=np.uint64) == want).all()


@pytest.mark.parametrize("logn,n_vec", [(5, 3), (9, 4), (12, 5), (13, 5), (16, 3), (18, 2)])
def test_four_step_batch_and_cyclic_round_trip(F, eng, O, logn, n_vec):
    """A batch of vectors in one call (two launches in total), every size class of the natural-order transform; then
    motivation/bsgs.py:31-36's intt brings the batch back (the scale n^-1 rides on the last stage)."""
    import ctypes as C
    from fhe_reliability_gpu_amd._lib import check, lib
    mod, g = 998244353, 3
    N = 1 << logn
    rng = np.random.default_rng(logn)
    a = rng.integers(0, mod, (n_vec, N), dtype=np.uint64)
    a[0, :4] = [0, 1, mod - 1, 2**64 - 1]                       # an out-of-range word is reduced first
    got = np.array(F.four_step_ntt(a, N, mod, g, n1=1 << (logn // 2)), dtype=np.uint64)
    for v in range(n_vec):
        assert (got[v] == O.ntt_cyclic(a[v] % np.uint64(mod), mod, g)).all()
    d, s = eng.upload(got), eng.alloc(got.size)
    check(lib.fhe_ntt_cyclic(eng._h, d.ptr, s.ptr, logn, n_vec, mod, g, 0, 1, None))
    assert (d.download().reshape(n_vec, N) == a % np.uint64(mod)).all()


def test_cyclic_61_bit_and_composite_modulus_large(F, O):
    """the integer path and a composite modulus (motivation/ntt.py:36's 15728641 = 173 x 90917) through the natural-order launches"""
    rng = np.random.default_rng(61)
    a = rng.integers(0, Q61, 1 << 14, dtype=np.uint64)
    assert F.ntt([int(x) for x in a], Q61, 37) == [int(x) for x in O.ntt_cyclic(a, Q61, 37)]
    mod = 15728641
    b = rng.integers(0, mod, 1 << 13, dtype=np.uint64)
    assert F.ntt([int(x) for x in b], mod, 3) == [int(x) for x in O.ntt_cyclic(b, mod, 3)]
