"""BASELINE configs 4 and 5 on one GPU, word for word against the oracle composites (oracle/keyswitch_ref.py):

  config 4  N = 2^17, L = 32 (+K), CKKS-shaped hmult with baseConv: multiply -> relinearize -> mod_switch
            (reliability_test/dotprod_test.cu:113-115; op sequence profile_framewk/build/data/ckks/16384_4:388-451)
  config 5  N = 2^16, L = 44 (+K), rotation / key switching, dnum in {4, 11, 44}
            (dnum sweep of profile_framewk/draw_dnum_rot_mul.py:63-65; ROTATE of 16384_4:452-539)

plus the large-base exact conversions those shapes land on (digits of alpha = 9..64 limbs: the scratch-indexed
instantiations and the FP64 accumulator folds of aux_kernels.hip) through the single-job and the job-list launches.
SEAL's / Phantom's own values are unavailable ("parity unpinned" there); the algebraic properties pin the meaning.
Bit-exact: every comparison is ==."""
import random

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


def _limbs(rng, qs, N):
    return np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])


def _key(rng, qs, dnum, N):
    out = np.empty((dnum, 2, len(qs), N), dtype=np.uint64)
    for d in range(dnum):
        for h in range(2):
            for j, q in enumerate(qs):
                out[d, h, j] = rng.integers(0, q, N, dtype=np.uint64)
    return out


# ------------------------------------------------------------------ large bases
@pytest.mark.parametrize("m", [8, 9, 16, 17, 33, 64])
@pytest.mark.parametrize("path", ["f64", "u64"])
def test_baseconv_exact_large_bases(F, eng, m, path):
    """m input limbs -> k outputs, single-job launch.  f64: every modulus below 2^50 (FP64 accumulators with the relax()
    folds past 8 terms); u64: 61-bit moduli mixed in (Shoup path)."""
    from oracle import cport as O
    N, k = 4096, 5
    if path == "f64":
        qs = F.create_moduli(N, [50] * (m + k))
    else:
        qs = F.create_moduli(N, [50 if i % 3 else 61 for i in range(m + k)])
    mod_in, mod_out = qs[:m], qs[m:]
    rng = np.random.default_rng(m * 7 + len(path))
    r = _limbs(rng, mod_in, N)
    r[:, :4] = np.array([[0, 1, q - 1, q // 2] for q in mod_in], dtype=np.uint64)      # edge residues
    bc = F.BaseConv(eng, mod_in, mod_out)
    out = eng.alloc(k * N)
    bc.exact(out, eng.upload(r), N)
    assert (out.download().reshape(k, N) == O.baseconv_exact(r, mod_in, mod_out)).all()


@pytest.mark.parametrize("logn,L,K,dnum,bits", [(12, 22, 11, 2, 50), (10, 34, 17, 2, 50), (11, 18, 9, 2, 61), (10, 12, 12, 1, 50)])
def test_keyswitch_large_digits_match_oracle(F, eng, logn, L, K, dnum, bits):
    """alpha = ceil(L/dnum) in {9, 11, 12, 17}: the job-list launch of the digit extensions with scratch-indexed digits."""
    from oracle.keyswitch_ref import keyswitch_ref
    N = 1 << logn
    qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 1000 + L)
    c = _limbs(rng, qs[:L], N)
    evk = _key(rng, qs, dnum, N)
    o0, o1 = F.KeySwitch(eng, t, L, K, dnum).apply(eng.upload(c), eng.upload(evk))
    w0, w1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn)
    assert (o0.download() == w0).all() and (o1.download() == w1).all()


# ------------------------------------------------------------------ homomorphic multiply, small shapes
@pytest.mark.parametrize("logn,L,K,dnum,bits,t_plain", [(10, 4, 2, 2, 50, 0), (12, 6, 2, 3, 50, 0), (13, 3, 1, 3, 61, 0), (11, 5, 2, 5, 50, 786433)])
def test_hmult_pieces_and_composite_match_oracle(F, eng, logn, L, K, dnum, bits, t_plain):
    """fhe_tensor_product, fhe_relinearize, fhe_rescale one by one and fhe_hmult as a whole (CKKS flooring and the BGV
    form with a plain modulus) against the oracle composites."""
    from oracle import keyswitch_ref as R
    N = 1 << logn
    qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn + 31 * L)
    a0, a1, b0, b1 = (_limbs(rng, qs[:L], N) for _ in range(4))
    rlk = _key(rng, qs, dnum, N)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    if t_plain:
        ks.set_plain_modulus(t_plain)
    up = eng.upload
    d = ks.tensor(up(a0), up(a1), up(b0), up(b1))
    w = R.tensor_ref(a0, a1, b0, b1, qs)
    for got, want in zip(d, w):
        assert (got.download() == want).all()
    if not t_plain:          # the plain-modulus mod-down inside the key switch is covered by the BGV harness tests
        c0, c1 = ks.relinearize(d[0], d[1], d[2], up(rlk))
        w0, w1 = R.keyswitch_ref(w[2], rlk, qs, L, K, dnum, logn, add0=w[0], add1=w[1])
        assert (c0.download() == w0).all() and (c1.download() == w1).all()
        both = np.stack([w0, w1])
        assert (ks.rescale(up(both), 2).download() == R.rescale_ref(both, qs, L, logn)).all()
        three = np.stack([w[0], w[1], w[2]])
        assert (ks.rescale(up(three), 3).download() == R.rescale_ref(three, qs, L, logn)).all()
        for resc in (True, False):
            o0, o1 = ks.hmult(up(a0), up(a1), up(b0), up(b1), up(rlk), rescale=resc)
            h0, h1 = R.hmult_ref(a0, a1, b0, b1, rlk, qs, L, K, dnum, logn, rescale=resc)
            assert (o0.download() == h0).all() and (o1.download() == h1).all()
    else:
        both = np.stack([a0, a1])
        assert (ks.rescale(up(both), 2).download() == R.rescale_ref(both, qs, L, logn, plain_modulus=t_plain)).all()


@pytest.mark.parametrize("logn,L,K,dnum,ct_bits,sp_bits", [
    (13, 4, 2, 2, [50] * 4, [50] * 2),            # FP64 limbs throughout
    (14, 5, 2, 5, [50] * 5, [50] * 2),            # one-limb digits (SEAL's form), K = 2
    (13, 3, 2, 3, [61] * 3, [61] * 2),            # integer path throughout
    (13, 5, 3, 2, [50, 61, 50, 50, 61], [50, 61, 50]),   # mixed: the last limb on the integer path under FP64 limbs
    (13, 4, 2, 2, [61, 50, 61, 50], [50, 50]),    # mixed: an FP64 last limb under integer-path limbs
    (15, 2, 2, 1, [50] * 2, [50] * 2),            # L = 2: one limb survives
])
def test_hmult_fused_rescale_matches_oracle_and_two_step_form(F, eng, logn, L, K, dnum, ct_bits, sp_bits):
    """fhe_hmult at the two-launch sizes with K >= 2 runs the mod-down and the rescale behind ONE forward transform
    (capi_keyswitch.cpp ks_finish_rescale: NTT(conv_j + P y) instead of NTT(conv_j) and NTT(y)).  Its words must be the oracle's
    multiply -> relinearize -> mod_switch (reliability_test/dotprod_test.cu:113-115) and the engine's own two-step form's."""
    from oracle import keyswitch_ref as R
    N = 1 << logn
    qs = F.create_moduli(N, list(ct_bits) + list(sp_bits))
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 100 + L * 10 + K)
    a0, a1, b0, b1 = (_limbs(rng, qs[:L], N) for _ in range(4))
    a0[:, :3] = np.array([[0, 1, q - 1] for q in qs[:L]], dtype=np.uint64)
    rlk = _key(rng, qs, dnum, N)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    up = eng.upload
    h0, h1 = R.hmult_ref(a0, a1, b0, b1, rlk, qs, L, K, dnum, logn, rescale=True)
    got = {}
    try:
        for fused in (1, 0):
            eng.set_option("hmult_fused_rescale", fused)
            o0, o1 = ks.hmult(up(a0), up(a1), up(b0), up(b1), up(rlk), rescale=True)
            got[fused] = (o0.download(), o1.download())
    finally:
        eng.set_option("hmult_fused_rescale", 1)
    for fused in (1, 0):
        assert (got[fused][0] == h0).all() and (got[fused][1] == h1).all(), f"fused={fused}"
    # the un-rescaled form is untouched by the switch
    o0, o1 = ks.hmult(up(a0), up(a1), up(b0), up(b1), up(rlk), rescale=False)
    w0, w1 = R.hmult_ref(a0, a1, b0, b1, rlk, qs, L, K, dnum, logn, rescale=False)
    assert (o0.download() == w0).all() and (o1.download() == w1).all()


def _switch_key(O, qs, L, K, dnum, logn, s, s_target, rnd):
    """evk_d = (-a_d s + e_d + P Qhat_d [Qhat_d^-1]_{Q_d} s_target, a_d), NTT domain, (dnum, 2, L+K, N)."""
    N, M = 1 << logn, L + K
    Q, P = qs[:L], qs[L:]
    Qprod, Pprod = int(np.prod([int(x) for x in Q], dtype=object)), int(np.prod([int(x) for x in P], dtype=object))
    alpha = -(-L // dnum)
    rps = [O.root_powers(q, logn) for q in qs]
    res = lambda v, q: np.array([x % q for x in v], dtype=np.uint64)
    evk = np.zeros((dnum, 2, M, N), dtype=np.uint64)
    for d in range(dnum):
        lo, hi = d * alpha, min(L, (d + 1) * alpha)
        Qd = int(np.prod([int(x) for x in Q[lo:hi]], dtype=object))
        Qhat = Qprod // Qd
        Fd = Pprod * Qhat * pow(Qhat, -1, Qd)
        e = [rnd.randint(-4, 4) for _ in range(N)]
        for j, q in enumerate(qs):
            a = np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64)
            psi = O.min_primitive_root(q, 2 * N)
            a_s = O.polymul_ntt(a, res(s, q), psi, q)
            b = (res(e, q).astype(object) - a_s.astype(object) + (Fd % q) * res(s_target, q).astype(object)) % q
            evk[d, 0, j] = O.nwt_forward(b.astype(np.uint64), q, rps[j])
            evk[d, 1, j] = O.nwt_forward(a, q, rps[j])
    return evk


def _negacyclic_int(a, b):
    """product of two integer coefficient lists mod x^N + 1, exact (Python integers)."""
    N = len(a)
    r = [0] * N
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                k = i + j
                if k >= N:
                    r[k - N] -= x * y
                else:
                    r[k] += x * y
    return r


def test_hmult_is_a_homomorphic_product():
    """q_last * (out0 + out1 s) = (a0 + a1 s)(b0 + b1 s) + small  (mod Q / q_last): the property that defines
    multiply + relinearize + rescale, checked with big integers at N = 2^8 (ternary s, relin key for s^2)."""
    import fhe_reliability_gpu_amd as F
    from oracle import cport as O
    eng = F.default_engine()
    logn, L, K, dnum = 8, 4, 2, 2
    N = 1 << logn
    qs = F.create_moduli(N, [50] * L + [61] * K)
    Q = qs[:L]
    rnd = random.Random(5)
    s = [rnd.choice((-1, 0, 1)) for _ in range(N)]
    s2 = _negacyclic_int(s, s)
    rlk = _switch_key(O, qs, L, K, dnum, logn, s, s2, rnd)
    rps = [O.root_powers(q, logn) for q in qs]
    parts = [[np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64) for q in Q] for _ in range(4)]     # a0 a1 b0 b1, coefficient form
    ntt = lambda p: np.stack([O.nwt_forward(p[j], Q[j], rps[j]) for j in range(L)])
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    o0, o1 = ks.hmult(*(eng.upload(ntt(p)) for p in parts), eng.upload(rlk), rescale=True)
    o0, o1 = o0.download(), o1.download()
    Qr = Q[:L - 1]
    Qrp = int(np.prod([int(x) for x in Qr], dtype=object))
    ql = Q[L - 1]
    res = lambda v, q: np.array([x % q for x in v], dtype=np.uint64)
    err = []
    for j, q in enumerate(Qr):
        psi = O.min_primitive_root(q, 2 * N)
        mul = lambda x, y: O.polymul_ntt(np.asarray(x, dtype=np.uint64), np.asarray(y, dtype=np.uint64), psi, q).astype(object)
        A = (parts[0][j].astype(object) + mul(parts[1][j], res(s, q))) % q
        B = (parts[2][j].astype(object) + mul(parts[3][j], res(s, q))) % q
        lhs = (O.nwt_inverse(o0[j], q, rps[j]).astype(object) + mul(O.nwt_inverse(o1[j], q, rps[j]), res(s, q))) * (ql % q)
        err.append((lhs - mul((A % q).astype(np.uint64), (B % q).astype(np.uint64))) % q)
    worst = 0
    for i in range(N):
        x = 0
        for j, q in enumerate(Qr):
            Mj = Qrp // q
            x += int(err[j][i]) * Mj * pow(Mj, -1, q)
        x %= Qrp
        worst = max(worst, min(x, Qrp - x))
    assert worst < ql * (N + 2) + 64 * N, f"rescaled product is off by {worst} (q_last = {ql})"


# ------------------------------------------------------------------ config 5: N = 2^16, L = 44, rotation / key switch
@pytest.mark.parametrize("dnum", [4, 11, 44])
def test_config5_rotation_full_shape(F, eng, dnum):
    """BASELINE configs[4] on one GPU: every limb of both output parts of fhe_rotate at N = 2^16, L = 44, K = ceil(L/dnum)
    50-bit primes equals the oracle composite (INTT, per-digit exact extension, NTT, inner product with the key,
    mod-down, plus sigma(c0))."""
    from oracle.keyswitch_ref import rotate_ref
    logn, L = 16, 44
    K = -(-L // dnum)
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(44 + dnum)
    c0, c1 = _limbs(rng, qs[:L], N), _limbs(rng, qs[:L], N)
    gk = _key(rng, qs, dnum, N)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    galois = 5
    o0, o1 = ks.rotate(eng.upload(c0), eng.upload(c1), galois, eng.upload(gk))
    w0, w1 = rotate_ref(c0, c1, galois, gk, qs, L, K, dnum, logn)
    g0, g1 = o0.download(), o1.download()
    bad = [(h, j) for h, (g, w) in enumerate(((g0, w0), (g1, w1))) for j in range(L) if not (g[j] == w[j]).all()]
    assert not bad, f"limbs that differ from the oracle (part, limb): {bad[:8]}"


def test_config5_keyswitch_switches_keys_full_shape(F, eng):
    """The defining property at the full config-5 shape (N = 2^16, L = 44, K = 11, dnum = 4): out0 + out1 s = c s' + small,
    checked with big integers on a sample of coefficients (all 44 primes each)."""
    from oracle import cport as O
    logn, L, dnum = 16, 44, 4
    K = 11
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    Q = qs[:L]
    Qprod = int(np.prod([int(x) for x in Q], dtype=object))
    rnd = random.Random(16)
    nprng = np.random.default_rng(16)
    s = nprng.integers(-1, 2, N).tolist()
    s2 = nprng.integers(-1, 2, N).tolist()
    rps = np.stack([O.root_powers(q, logn) for q in qs])
    res = lambda v, q: (np.asarray(v, dtype=np.int64) % np.int64(q)).astype(np.uint64)
    # the key, built limb-wise with the oracle's transforms (vectorised: 220 limb products)
    M, alpha = L + K, 11
    Pprod = int(np.prod([int(x) for x in qs[L:]], dtype=object))
    s_ntt = O.nwt_forward_batch(np.stack([res(s, q) for q in qs]), qs, rps)
    s2_ntt = O.nwt_forward_batch(np.stack([res(s2, q) for q in qs]), qs, rps)
    evk = np.zeros((dnum, 2, M, N), dtype=np.uint64)
    for d in range(dnum):
        lo, hi = d * alpha, min(L, (d + 1) * alpha)
        Qd = int(np.prod([int(x) for x in Q[lo:hi]], dtype=object))
        Qhat = Qprod // Qd
        Fd = Pprod * Qhat * pow(Qhat, -1, Qd)
        e = nprng.integers(-4, 5, N)
        a = np.stack([nprng.integers(0, q, N, dtype=np.uint64) for q in qs])
        e_ntt = O.nwt_forward_batch(np.stack([res(e, q) for q in qs]), qs, rps)
        fd = np.stack([np.full(N, Fd % q, dtype=np.uint64) for q in qs])
        # b = e - a s + Fd s'   (a is drawn directly in the NTT domain: uniform either way)
        a_s = O.modmul_batch(a, s_ntt, qs)
        b = O.modmul_batch(fd, s2_ntt, qs, acc=e_ntt)
        for j, q in enumerate(qs):
            b[j] = (b[j] + (np.uint64(q) - a_s[j])) % np.uint64(q)
        evk[d, 0], evk[d, 1] = b, a
    c = np.stack([nprng.integers(0, q, N, dtype=np.uint64) for q in Q])          # coefficient domain
    c_ntt = O.nwt_forward_batch(c, Q, rps[:L])
    t = eng.tables(logn, qs)
    o0, o1 = F.KeySwitch(eng, t, L, K, dnum).apply(eng.upload(c_ntt), eng.upload(evk))
    o0, o1 = o0.download(), o1.download()
    # r = out0 + out1 s - c s'  per prime (NTT domain, then back), CRT on a sample of coefficients
    r = O.modmul_batch(o1, s_ntt[:L], Q, acc=o0)
    cs = O.modmul_batch(c_ntt, s2_ntt[:L], Q)
    for j, q in enumerate(Q):
        r[j] = (r[j] + (np.uint64(q) - cs[j])) % np.uint64(q)
    r = O.nwt_inverse_batch(r, Q, rps[:L])
    Mj = [Qprod // q for q in Q]
    Mi = [pow(Mj[j], -1, q) for j, q in enumerate(Q)]
    worst = 0
    for i in [0, 1, N - 1] + [rnd.randrange(N) for _ in range(253)]:
        x = sum(int(r[j][i]) * Mj[j] * Mi[j] for j in range(L)) % Qprod
        worst = max(worst, min(x, Qprod - x))
    assert worst < 8 * N * 64, f"key-switch noise {worst} is not small (Q has {Qprod.bit_length()} bits)"


# ------------------------------------------------------------------ config 4: N = 2^17, L = 32, hmult with baseConv
@pytest.mark.parametrize("dnum", [4])
def test_config4_hmult_full_shape(F, eng, dnum):
    """BASELINE configs[3] on one GPU: multiply -> relinearize -> mod_switch at N = 2^17, L = 32, K = 8 (dnum = 4),
    every limb of both outputs against the oracle composite."""
    from oracle.keyswitch_ref import hmult_ref
    logn, L = 17, 32
    K = -(-L // dnum)
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(17)
    a0, a1, b0, b1 = (_limbs(rng, qs[:L], N) for _ in range(4))
    rlk = _key(rng, qs, dnum, N)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    up = eng.upload
    o0, o1 = ks.hmult(up(a0), up(a1), up(b0), up(b1), up(rlk), rescale=True)
    w0, w1 = hmult_ref(a0, a1, b0, b1, rlk, qs, L, K, dnum, logn, rescale=True)
    g0, g1 = o0.download(), o1.download()
    assert g0.shape == (L - 1, N) and g1.shape == (L - 1, N)
    bad = [(h, j) for h, (g, w) in enumerate(((g0, w0), (g1, w1))) for j in range(L - 1) if not (g[j] == w[j]).all()]
    assert not bad, f"limbs that differ from the oracle (part, limb): {bad[:8]}"


# ------------------------------------------------------------------ round 3: hoisted baby steps at the config-5 shape
def test_config5_hoisted_rotations_full_shape(F, eng):
    """BASELINE configs[4] names a BSGS rotation workload: three baby rotations of ONE ciphertext at N = 2^16, L = 44, K = 11, dnum = 4
    with the decomposition shared (fhe_rotate_hoisted), every limb of every output against the oracle composite
    (oracle/keyswitch_ref.py rotate_hoisted_ref; profile_framewk/src/matmul_ckks.cpp:45-113 for the workload)."""
    from oracle.keyswitch_ref import rotate_hoisted_ref
    logn, L, K, dnum = 16, 44, 11, 4
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(2025)
    c0, c1 = _limbs(rng, qs[:L], N), _limbs(rng, qs[:L], N)
    elts = [3, 9, 2 * N - 1]
    keys = [_key(rng, qs, dnum, N) for _ in elts[:2]]
    keys.append(keys[0])                                   # (a third element on the first key's words: the arithmetic does not care)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    prepared = [ks.prepare_galois_key(eng.upload(k), e) for k, e in zip(keys, elts)]
    outs = ks.rotate_hoisted(eng.upload(c0), eng.upload(c1), elts, prepared)
    for (o0, o1), e, k in zip(outs, elts, keys):
        w0, w1 = rotate_hoisted_ref(c0, c1, e, k, qs, L, K, dnum, logn)
        assert (o0.download() == w0).all() and (o1.download() == w1).all(), f"galois element {e}"
    eng.check()


def test_rescale_error_is_bounded_at_the_decryption_level(F, eng):
    """fhe_rescale floors: c' = (c - [c]_q_last) / q_last exactly, so for a ciphertext (c0, c1) with c0 + c1 s = m (mod Q):
    c0' + c1' s = (m - ([c0] + [c1] s)) / q_last, i.e. the scaled plaintext up to (1 + |s|_1) in absolute value.  SEAL / Phantom round
    instead of flooring: this test pins the engine's choice, a later switch would change the bound's centre."""
    from oracle import cport as O
    logn, N, L, K = 10, 1024, 4, 1
    qs = F.create_moduli(N, [50] * (L + K))
    Q = qs[:L]
    ql = Q[-1]
    rnd = random.Random(3)
    s = [rnd.choice((-1, 0, 1)) for _ in range(N)]
    rps = [O.root_powers(q, logn) for q in Q]
    res = lambda v, q: np.array([x % q for x in v], dtype=np.uint64)
    psi = [O.min_primitive_root(q, 2 * N) for q in Q]
    big_m = [rnd.randrange(-(1 << 120), 1 << 120) for _ in range(N)]
    c1 = np.stack([np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64) for q in Q])
    c0 = np.stack([((res(big_m, q).astype(object) - O.polymul_ntt(c1[j], res(s, q), psi[j], q).astype(object)) % q).astype(np.uint64) for j, q in enumerate(Q)])
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, L)
    parts = np.stack([np.stack([O.nwt_forward(c[j], Q[j], rps[j]) for j in range(L)]) for c in (c0, c1)])
    out = ks.rescale(eng.upload(parts), 2).download().reshape(2, L - 1, N)
    # decrypt over Q' = Q / q_last, CRT to a centred integer, compare with m / q_last
    Qp = Q[:-1]
    Qprod = 1
    for q in Qp:
        Qprod *= int(q)
    dec = []
    for j, q in enumerate(Qp):
        x0 = O.nwt_inverse(out[0, j], q, rps[j]).astype(object)
        x1 = O.polymul_ntt(O.nwt_inverse(out[1, j], q, rps[j]), res(s, q), psi[j], q).astype(object)
        dec.append((x0 + x1) % q)
    bound = 1 + sum(abs(v) for v in s)
    for i in range(0, N, 7):
        x = 0
        for j, q in enumerate(Qp):
            Mj = Qprod // int(q)
            x += int(dec[j][i]) * Mj * pow(Mj, -1, int(q))
        x %= Qprod
        x = x - Qprod if x > Qprod // 2 else x
        err = x * int(ql) - big_m[i]                         # = -(delta0 + delta1 s)_i up to the wrap of c0 + c1 s modulo Q
        # c0 + c1 s = m + k Q for a small integer polynomial k (|k| <= 1 + |s|_1): remove the multiple of Q that rides along
        Qfull = Qprod * int(ql)
        err -= round(err / Qfull) * Qfull
        assert abs(err) <= bound * int(ql), (i, err)
    eng.check()


def test_fourstep_range_is_pinned(F, eng):
    """n1, n2 <= 2^20, n1 * n2 <= 2^26 and mod < 2^61 (include/fhe_mi355x.h): the sizes beyond return a status, not a wrong answer
    (ADVICE round 2).  Up to 2^20 one natural-order plan; above, the reference's own composition (next test)."""
    import ctypes as C
    from fhe_reliability_gpu_amd._lib import lib
    h = C.c_void_p()
    assert lib.fhe_fourstep_create(eng._h, 1 << 14, 1 << 13, 998244353, 3, C.byref(h)) != 0     # N = 2^27 (and it does not divide mod - 1)
    assert b"2^26" in lib.fhe_last_error()
    assert lib.fhe_fourstep_create(eng._h, 1 << 21, 2, 998244353, 3, C.byref(h)) != 0                             # a factor past 2^20
    assert lib.fhe_fourstep_create(eng._h, 1 << 10, 1 << 10, 998244353, 3, C.byref(h)) == 0                        # N = 2^20: the largest single plan
    lib.fhe_fourstep_destroy(h)
    assert lib.fhe_fourstep_create(eng._h, 1 << 12, 1 << 11, 998244353, 3, C.byref(h)) == 0                        # N = 2^23 = the largest N dividing mod - 1
    lib.fhe_fourstep_destroy(h)


@pytest.mark.parametrize("n1,n2,n_vec,mod_bits", [(1 << 11, 1 << 10, 2, 0), (1 << 10, 1 << 12, 1, 0), (1 << 11, 1 << 12, 1, 0), (1 << 20, 2, 1, 0), (8, 1 << 18, 2, 50),
                                                   (1 << 11, 1 << 10, 1, 61)])
def test_fourstep_past_the_largest_plan_matches_oracle(F, eng, n1, n2, n_vec, mod_bits):
    """N = 2^21 .. 2^23: four_step_ntt as the reference composes it (reliability_test/four_step_ntt_prot.py:71-109: transpose, n1 transforms
    of length n2, twiddle w^(k2 t1), n2 transforms of length n1, transpose), each factor through the natural-order transform of its length --
    against the oracle (the DFT that flow equals).  mod 998244353 (the reference's default, N up to 2^23), a 50-bit and a 61-bit prime."""
    import ctypes as C
    from fhe_reliability_gpu_amd._lib import check, lib
    from oracle import cport as O
    N = n1 * n2
    if mod_bits:
        mod = F.create_moduli(N // 2, [mod_bits])[0]          # = 1 mod N
        g = next(x for x in range(2, 200) if pow(x, (mod - 1) // 2, mod) == mod - 1)     # a non-residue: g^((mod-1)/N) has order N
    else:
        mod, g = 998244353, 3
    rng = np.random.default_rng(n1 + n2)
    a = rng.integers(0, mod, (n_vec, N), dtype=np.uint64)
    a[0, :3] = [0, 1, mod - 1]
    h = C.c_void_p()
    check(lib.fhe_fourstep_create(eng._h, n1, n2, mod, g, C.byref(h)))
    try:
        src, dst = eng.upload(a), eng.alloc(a.size)
        check(lib.fhe_fourstep_ntt_batch(eng._h, dst.ptr, src.ptr, h, n_vec, None))
        got = dst.download().reshape(n_vec, N)
    finally:
        lib.fhe_fourstep_destroy(h)
    # (the oracle's four_step_ntt runs its sub-transforms as direct sums, O(N (n1 + n2)): minutes at these sizes; the flow equals the DFT with
    # w = g^((mod-1)/N) (four_step_ntt_prot.py:244-245), which the oracle's O(N log N) cyclic transform computes -- the same check
    # test_four_step_batch_and_cyclic_round_trip makes; the four-step restatement itself is pinned against it at small sizes in tests/)
    for v in range(n_vec):
        assert (got[v] == O.ntt_cyclic(a[v], mod, g)).all(), v


def test_round3_entry_points_report_misuse(F, eng):
    """Statuses, not crashes: the sharded one-transform hmult phases on a plan without gather buffers, on a shape that rescales in a
    separate step, with null outputs; the option switch; an oversized batch of the large-N four-step."""
    import ctypes as C
    from fhe_reliability_gpu_amd._lib import lib
    N = 1 << 13
    qs = F.create_moduli(N, [50] * 6)
    t = eng.tables(13, qs)
    ks = F.KeySwitch(eng, t, 4, 2, 2)
    assert lib.fhe_hmult_shard_fusable(eng._h, ks._h) == 0                                   # one-device plan: fhe_hmult does it itself
    assert lib.fhe_hmult_shard_finish_begin(eng._h, ks._h, None, None, None) != 0
    assert lib.fhe_hmult_shard_finish_end(eng._h, ks._h, None, None, None, None, None) != 0
    assert lib.fhe_hmult_shard_fusable(None, None) == 0
    with pytest.raises(Exception):
        eng.set_option("no_such_option", 1)
    eng.set_option("hmult_fused_rescale", 1)
    h = C.c_void_p()
    assert lib.fhe_fourstep_create(eng._h, 1 << 11, 1 << 10, 998244353, 3, C.byref(h)) == 0
    try:
        buf = eng.alloc(1 << 21)
        assert lib.fhe_fourstep_ntt_batch(eng._h, buf.ptr, buf.ptr, h, 1 << 14, None) != 0       # 2^14 vectors of 2^21 words: refused before any launch
        assert lib.fhe_fourstep_ntt_batch(eng._h, buf.ptr, buf.ptr, h, 0, None) == 0             # empty batch
    finally:
        lib.fhe_fourstep_destroy(h)
