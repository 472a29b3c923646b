"""Hoisted rotations (profile_framewk/src/matmul_ckks.cpp:45-113, reliability_test/dotprod_test.cu:143-148):
several Galois elements applied to ONE ciphertext with the decomposition of c1 shared.  Word for word against the oracle composite
(oracle/keyswitch_ref.py rotate_hoisted_ref: sigma applied to the extended digits), at the decryption level against fhe_rotate (the
two key switches differ word by word -- the exact extension lifts to [0, P_d), which sigma's sign flips do not preserve -- and
decrypt to the same plaintext), the prepared key against sigma^-1 of the key, and the error statuses of the new entry points."""
import ctypes as C
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


def _rand_case(F, logn, L, K, dnum, bits, seed):
    N = 1 << logn
    qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
    rng = np.random.default_rng(seed)
    c0 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    c1 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    keys = [np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)]) for _ in range(3)]
    return qs, c0, c1, keys


@pytest.mark.parametrize("logn,L,K,dnum,bits", [(5, 3, 1, 3, 50), (10, 4, 2, 2, 50), (12, 6, 2, 3, 61), (13, 4, 1, 4, 50), (13, 5, 2, 2, 50),
                                                (14, 5, 2, 3, 61), (16, 3, 1, 3, 50)])
def test_hoisted_rotations_match_oracle_composite(F, eng, logn, L, K, dnum, bits):
    from oracle.keyswitch_ref import rotate_hoisted_ref, rotate_ref
    N = 1 << logn
    qs, c0, c1, keys = _rand_case(F, logn, L, K, dnum, bits, logn * 17 + L)
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    elts = [3, 2 * N - 1, (1 << (logn - 1)) + 1]
    d0, d1 = eng.upload(c0), eng.upload(c1)
    prepared = []
    for k, key in zip(elts, keys):
        pk = ks.prepare_galois_key(eng.upload(key), k)
        prepared.append(pk)
    outs = ks.rotate_hoisted(d0, d1, elts, prepared)
    for (o0, o1), k, key in zip(outs, elts, keys):
        w0, w1 = rotate_hoisted_ref(c0, c1, k, key, qs, L, K, dnum, logn)
        assert (o0.download() == w0).all() and (o1.download() == w1).all(), f"galois element {k}"
    assert (d0.download() == c0).all() and (d1.download() == c1).all()          # inputs untouched
    # a second batch on the same plan (buffers reused), one element, and the plain rotation still gives ITS oracle
    (r0, r1), = ks.rotate_hoisted(d0, d1, elts[1:2], prepared[1:2])
    w0, w1 = rotate_hoisted_ref(c0, c1, elts[1], keys[1], qs, L, K, dnum, logn)
    assert (r0.download() == w0).all() and (r1.download() == w1).all()
    p0, p1 = ks.rotate(d0, d1, elts[0], eng.upload(keys[0]))
    v0, v1 = rotate_ref(c0, c1, elts[0], keys[0], qs, L, K, dnum, logn)
    assert (p0.download() == v0).all() and (p1.download() == v1).all()
    eng.check()


def test_prepared_key_is_the_inverse_automorphism_of_the_key(F, eng):
    from oracle import cport as O
    from oracle.keyswitch_ref import galois_coeff
    logn, L, K, dnum, k = 10, 3, 2, 2, 5
    N = 1 << logn
    qs, _, _, keys = _rand_case(F, logn, L, K, dnum, 50, 3)
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    got = ks.prepare_galois_key(eng.upload(keys[0]), k).download().reshape(keys[0].shape)
    kinv = pow(k, -1, 2 * N)
    for d in range(dnum):
        for h in range(2):
            for j, q in enumerate(qs):
                rp = O.root_powers(q, logn)
                co = galois_coeff(O.nwt_inverse(keys[0][d, h, j], q, rp), kinv, q)
                assert (got[d, h, j] == O.nwt_forward(co, q, rp)).all()


def _poly_mul(a, b, q, logn):
    from oracle import cport as O
    psi = O.min_primitive_root(q, 2 << logn)
    return O.polymul_ntt(np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64), psi, q)


def test_hoisted_and_plain_rotation_decrypt_to_the_same_plaintext(F, eng):
    """decrypt(hoisted rotation) = decrypt(fhe_rotate) = sigma(m) up to key-switch noise, although the ciphertext words differ."""
    from oracle import cport as O
    from oracle.keyswitch_ref import galois_coeff
    logn, N, L, K, dnum = 10, 1024, 4, 2, 2
    M = L + K
    qs = F.create_moduli(N, [50] * L + [61] * K)
    Q, P = qs[:L], qs[L:]
    Qprod = int(np.prod([int(x) for x in Q], dtype=object))
    Pprod = int(np.prod([int(x) for x in P], dtype=object))
    alpha = -(-L // dnum)
    rnd = random.Random(11)
    s = [rnd.choice((-1, 0, 1)) for _ in range(N)]
    rps = [O.root_powers(q, logn) for q in qs]
    res = lambda v, q: np.array([x % q for x in v], dtype=np.uint64)

    def galois_key(k):
        sig_s = [0] * N
        for i, v in enumerate(s):
            j = (i * k) % (2 * N)
            if j >= N:
                sig_s[j - N] = -v
            else:
                sig_s[j] = v
        gk = np.zeros((dnum, 2, M, N), dtype=np.uint64)
        for d in range(dnum):
            lo, hi = d * alpha, min(L, (d + 1) * alpha)
            Qd = int(np.prod([int(x) for x in Q[lo:hi]], dtype=object))
            Qhat = Qprod // Qd
            Fd = Pprod * Qhat * pow(Qhat, -1, Qd)
            e = [rnd.randint(-4, 4) for _ in range(N)]
            for j, q in enumerate(qs):
                a = np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64)
                a_s = _poly_mul(a, res(s, q), q, logn)
                b = (res(e, q).astype(object) - a_s.astype(object) + (Fd % q) * res(sig_s, q).astype(object)) % q
                gk[d, 0, j] = O.nwt_forward(b.astype(np.uint64), q, rps[j])
                gk[d, 1, j] = O.nwt_forward(a, q, rps[j])
        return gk

    m = [rnd.randrange(1 << 30) for _ in range(N)]
    c1 = np.stack([np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64) for q in Q])
    c0 = np.stack([((res(m, q).astype(object) - _poly_mul(c1[j], res(s, q), q, logn).astype(object)) % q).astype(np.uint64) for j, q in enumerate(Q)])
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    d0 = eng.upload(np.stack([O.nwt_forward(c0[j], Q[j], rps[j]) for j in range(L)]))
    d1 = eng.upload(np.stack([O.nwt_forward(c1[j], Q[j], rps[j]) for j in range(L)]))
    elts = [5, 25, 2 * N - 1]
    gks = [galois_key(k) for k in elts]
    outs = ks.rotate_hoisted(d0, d1, elts, [ks.prepare_galois_key(eng.upload(g), k) for g, k in zip(gks, elts)])
    differ = 0
    for (h0, h1), k, gk in zip(outs, elts, gks):
        p0, p1 = ks.rotate(d0, d1, k, eng.upload(gk))
        h0, h1, p0, p1 = h0.download(), h1.download(), p0.download(), p1.download()
        differ += int((h0 != p0).any() or (h1 != p1).any())
        want = [int(x) for x in galois_coeff(np.array(m, dtype=np.uint64), k, 1 << 62)]
        want = [w if w < (1 << 61) else w - (1 << 62) for w in want]
        for o0, o1 in ((h0, h1), (p0, p1)):
            q = Q[0]
            x0 = O.nwt_inverse(o0[0], q, rps[0]).astype(object)
            x1 = _poly_mul(O.nwt_inverse(o1[0], q, rps[0]), res(s, q), q, logn).astype(object)
            dec = (x0 + x1) % q
            worst = 0
            for i in range(N):
                diff = (int(dec[i]) - want[i]) % q
                worst = max(worst, abs(diff - q if diff > q // 2 else diff))
            assert worst < 16 * N, (k, worst)
    assert differ == len(elts)      # the two forms are different ciphertexts of the same plaintext (see the module docstring)


def test_hoisted_error_statuses(F, eng):
    from fhe_reliability_gpu_amd._lib import lib, vp
    logn, L, K, dnum = 10, 3, 1, 3
    qs, c0, c1, keys = _rand_case(F, logn, L, K, dnum, 50, 9)
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    d0, d1, key = eng.upload(c0), eng.upload(c1), eng.upload(keys[0])
    with pytest.raises(F.FheError):
        ks.prepare_galois_key(key, 4)                      # even element
    pk = ks.prepare_galois_key(key, 3)
    with pytest.raises(F.FheError):
        ks.rotate_hoisted(d0, d1, [6], [pk])               # even element
    # out of place: an output may not be one of the parts
    o = eng.alloc(L << logn)
    a0, a1, kk, ge = (vp * 1)(d0.ptr), (vp * 1)(o.ptr), (vp * 1)(pk.ptr), (C.c_uint32 * 1)(3)
    assert lib.fhe_rotate_hoisted(eng._h, ks._h, a0, a1, d0.ptr, d1.ptr, ge, kk, 1, None) != 0
    assert b"out of place" in lib.fhe_last_error()
    # plain rotate: the same contract
    assert lib.fhe_rotate(eng._h, ks._h, d1.ptr, o.ptr, d0.ptr, d1.ptr, 3, key.ptr, None) != 0
    # rescale: input and output strides differ, in place is refused
    parts = eng.upload(np.stack([c0, c1]))
    assert lib.fhe_rescale(eng._h, ks._h, parts.ptr, parts.ptr, 2, None) != 0
    assert b"out of place" in lib.fhe_last_error()
    # an empty batch is a no-op
    assert lib.fhe_rotate_hoisted(eng._h, ks._h, None, None, d0.ptr, d1.ptr, None, None, 0, None) == 0
    eng.check()


@pytest.mark.parametrize("logn,L,K,dnum,n1,n2", [(10, 3, 1, 3, 3, 2), (13, 4, 2, 2, 2, 3), (12, 3, 1, 1, 4, 1), (13, 3, 1, 3, 1, 2)])
def test_bsgs_matvec_matches_oracle_composite(F, eng, logn, L, K, dnum, n1, n2):
    """fhe_bsgs_matvec (hoisted baby rotations, one launch per inner sum, giant rotations accumulated) word for word against the oracle's
    rotate / multiply_plain / add sequence (profile_framewk/src/matmul_ckks.cpp:45-113)."""
    from oracle.keyswitch_ref import bsgs_matvec_ref
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn + n1 * 10 + n2)
    mk = lambda rows: np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:rows]])
    key = lambda: np.stack([np.stack([mk(L + K) for _ in range(2)]) for _ in range(dnum)])
    c0, c1 = mk(L), mk(L)
    diags = np.stack([np.stack([mk(L) for _ in range(n1)]) for _ in range(n2)])
    baby_elts = [pow(3, b, 2 * N) for b in range(1, n1)]
    giant_elts = [pow(3, g * n1, 2 * N) for g in range(1, n2)]
    baby_keys, giant_keys = [key() for _ in baby_elts], [key() for _ in giant_elts]
    ks = F.KeySwitch(eng, t, L, K, dnum)
    prepared = [ks.prepare_galois_key(eng.upload(k), e) for k, e in zip(baby_keys, baby_elts)]
    o0, o1 = ks.bsgs_matvec(eng.upload(c0), eng.upload(c1), eng.upload(diags), n1, n2, baby_elts, prepared, giant_elts, [eng.upload(k) for k in giant_keys])
    w0, w1 = bsgs_matvec_ref(c0, c1, diags, baby_elts, baby_keys, giant_elts, giant_keys, qs, L, K, dnum, logn)
    assert (o0.download() == w0).all() and (o1.download() == w1).all()
    eng.check()


def test_bsgs_matvec_on_a_trivial_ciphertext_is_the_plaintext_formula(F, eng):
    """x = (m, 0): a key switch of the zero polynomial is exactly zero, so every rotation is exact and the product must equal
    sum_g sigma_G( sum_b diag[g][b] * sigma_B(m) ) computed with plain polynomial automorphisms -- the structure of
    motivation/bsgs.py:39-52 / matmul_ckks.cpp:45-113 checked without going through the key-switch oracle."""
    from oracle import cport as O
    from oracle.keyswitch_ref import galois_coeff
    logn, L, K, dnum, n1, n2 = 11, 3, 1, 3, 3, 3
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    Q = qs[:L]
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(5)
    mk = lambda rows: np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:rows]])
    key = lambda: np.stack([np.stack([mk(L + K) for _ in range(2)]) for _ in range(dnum)])
    rps = [O.root_powers(q, logn) for q in Q]
    m = mk(L)                                                   # NTT form
    diags = np.stack([np.stack([mk(L) for _ in range(n1)]) for _ in range(n2)])
    baby_elts = [pow(5, b, 2 * N) for b in range(1, n1)]
    giant_elts = [pow(5, g * n1, 2 * N) for g in range(1, n2)]
    ks = F.KeySwitch(eng, t, L, K, dnum)
    prepared = [ks.prepare_galois_key(eng.upload(key()), e) for e in baby_elts]
    zero = eng.upload(np.zeros((L, N), dtype=np.uint64))
    o0, o1 = ks.bsgs_matvec(eng.upload(m), zero, eng.upload(diags), n1, n2, baby_elts, prepared, giant_elts, [eng.upload(key()) for _ in giant_elts])

    def sigma(x, k):                                            # NTT form -> NTT form
        return np.stack([O.nwt_forward(galois_coeff(O.nwt_inverse(x[l], Q[l], rps[l]), k, Q[l]), Q[l], rps[l]) for l in range(L)])
    want = np.zeros((L, N), dtype=np.uint64)
    for g in range(n2):
        inner = None
        for b in range(n1):
            inner = O.modmul_batch(diags[g, b], m if b == 0 else sigma(m, baby_elts[b - 1]), Q, acc=inner)
        if g:
            inner = sigma(inner, giant_elts[g - 1])
        want = np.stack([(want[l] + inner[l]) % np.uint64(Q[l]) for l in range(L)])
    assert (o0.download() == want).all()
    assert not o1.download().any()
    eng.check()


@pytest.mark.parametrize("logn,L,K,dnum,n_rot,bits", [(13, 4, 2, 2, 5, 50), (12, 3, 2, 3, 9, 61), (14, 5, 1, 5, 2, 50)])
def test_hoisted_rotations_one_stream_two_streams_and_capture_agree(F, eng, logn, L, K, dnum, n_rot, bits):
    """fhe_rotate_hoisted alternates its rotations between the caller's stream and the context's side stream (second buffer set);
    with the side stream switched off ("ntt_split" 0) and inside a stream capture (where it must not fork) the words are the same.
    On one stream the inner products of up to four rotations are formed in one pass over the shared digits (launch_ks_mac_multi):
    batches of 5 and 9 cross the group boundary, the 61-bit shape takes the integer path."""
    import torch
    from fhe_reliability_gpu_amd._lib import check, lib
    n = 1 << logn
    qk = F.create_moduli(n, [bits] * (L + K))
    tk = eng.tables(logn, qk)
    ks = F.KeySwitch(eng, tk, L, K, dnum)
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    mk = lambda *shape: torch.randint(0, qk[0], shape, generator=g, device="cuda", dtype=torch.int64)
    c0, c1 = mk(L, n), mk(L, n)
    keys = [mk(dnum, 2, L + K, n) for _ in range(n_rot)]
    elts = [pow(3, b + 1, 2 * n) for b in range(n_rot)]
    o0 = [torch.zeros((L, n), dtype=torch.int64, device="cuda") for _ in range(n_rot)]
    o1 = [torch.zeros((L, n), dtype=torch.int64, device="cuda") for _ in range(n_rot)]
    vp_t = C.c_void_p * n_rot
    a0, a1 = vp_t(*[x.data_ptr() for x in o0]), vp_t(*[x.data_ptr() for x in o1])
    kk, ge = vp_t(*[k.data_ptr() for k in keys]), (C.c_uint32 * n_rot)(*elts)
    P = lambda x: C.c_void_p(x.data_ptr())

    def call(s):
        check(lib.fhe_rotate_hoisted(eng._h, ks._h, a0, a1, P(c0), P(c1), ge, kk, n_rot, C.c_void_p(s.cuda_stream)))

    s = torch.cuda.Stream()
    torch.cuda.synchronize()          # operands and zero-filled outputs settled before the first call on another stream
    call(s)
    torch.cuda.synchronize()
    want = [(x.clone(), y.clone()) for x, y in zip(o0, o1)]
    eng.set_option("ntt_split", 0)
    try:
        for x in o0 + o1:
            x.zero_()
        torch.cuda.synchronize()      # (the fills run on torch's default stream, the call on another one)
        call(s)
        torch.cuda.synchronize()
        assert all(bool((x == w[0]).all()) and bool((y == w[1]).all()) for x, y, w in zip(o0, o1, want))
    finally:
        eng.set_option("ntt_split", -1)
    graph, cap = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=cap):
        call(cap)
    for x in o0 + o1:
        x.zero_()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert all(bool((x == w[0]).all()) and bool((y == w[1]).all()) for x, y, w in zip(o0, o1, want))
    eng.check()
