"""Rotation / key-switch composite on the GPU: word-for-word against the
oracle-side restatement of the same sequence, and the algebraic property that defines key
switching (big-integer check).  SEAL's own values are unavailable: "parity unpinned" there."""
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


def _poly_mul(a, b, q, logn):
    from oracle import cport as O
    psi = O.min_primitive_root(q, 2 << logn)
    return O.polymul_ntt(np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64), psi, q)


@pytest.mark.parametrize("logn", [4, 10, 13])
@pytest.mark.parametrize("k", [3, 5, 2 * 16 - 1])
def test_automorphism_coefficient_and_ntt_domain(F, eng, logn, k):
    from oracle import cport as O
    from oracle.keyswitch_ref import galois_coeff
    N = 1 << logn
    qs = F.create_moduli(N, [50, 61])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 100 + k)
    a = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])
    want = np.stack([galois_coeff(a[l], k, q) for l, q in enumerate(qs)])
    d = eng.upload(a)
    assert (F.automorphism(eng, t, d, k).download() == want).all()
    # NTT domain: sigma_k commutes with the transform
    t.forward(d)
    rot = F.automorphism(eng, t, d, k, ntt_domain=True)
    t.inverse(rot)
    assert (rot.download() == want).all()
    # composition: sigma_k o sigma_k^-1 = id
    kinv = pow(k, -1, 2 * N)
    back = F.automorphism(eng, t, F.automorphism(eng, t, eng.upload(a), k), kinv)
    assert (back.download() == a).all()


@pytest.mark.parametrize("logn,L,K,dnum,bits", [(10, 4, 1, 4, 50), (10, 4, 2, 2, 50), (12, 6, 2, 3, 50), (13, 3, 2, 1, 61)])
def test_keyswitch_matches_oracle_composite(F, eng, logn, L, K, dnum, bits):
    from oracle.keyswitch_ref import keyswitch_ref
    N, M = 1 << logn, L + K
    qs = F.create_moduli(N, [bits] * L + [61] * K) if bits != 61 else F.create_moduli(N, [61] * M)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn + L * 7 + dnum)
    c = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    evk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
    ks = F.KeySwitch(eng, t, L, K, dnum)
    o0, o1 = ks.apply(eng.upload(c), eng.upload(evk))
    w0, w1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn)
    assert (o0.download() == w0).all() and (o1.download() == w1).all()


@pytest.mark.parametrize("L,K,dnum", [(4, 2, 2), (3, 1, 3)])
def test_keyswitch_switches_keys(F, eng, L, K, dnum):
    """out0 + out1*s = c*s' + small noise (mod Q) when evk_d = (-a_d s + e_d + P Qhat_d [Qhat_d^-1]_{Q_d} s', a_d)."""
    from oracle import cport as O
    logn, N, M = 10, 1024, L + K
    qs = F.create_moduli(N, [50] * L + [61] * K)
    Q, P = qs[:L], qs[L:]
    Qprod, Pprod = int(np.prod([int(x) for x in Q], dtype=object)), int(np.prod([int(x) for x in P], dtype=object))
    alpha = -(-L // dnum)
    rnd = random.Random(42)
    s = [rnd.choice((-1, 0, 1)) for _ in range(N)]
    s2 = [rnd.choice((-1, 0, 1)) for _ in range(N)]
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
            a_s = _poly_mul(a, res(s, q), q, logn)
            b = (res(e, q).astype(object) - a_s.astype(object) + (Fd % q) * res(s2, q).astype(object)) % q
            evk[d, 0, j] = O.nwt_forward(b.astype(np.uint64), q, rps[j])
            evk[d, 1, j] = O.nwt_forward(a, q, rps[j])
    t = eng.tables(logn, qs)
    c = np.stack([np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64) for q in Q])        # coefficient domain
    c_ntt = np.stack([O.nwt_forward(c[j], Q[j], rps[j]) for j in range(L)])
    o0, o1 = F.KeySwitch(eng, t, L, K, dnum).apply(eng.upload(c_ntt), eng.upload(evk))
    o0, o1 = o0.download(), o1.download()
    # r_j = out0 + out1*s - c*s'  per prime, then CRT to a centred integer
    r = []
    for j, q in enumerate(Q):
        x0 = O.nwt_inverse(o0[j], q, rps[j]).astype(object)
        x1 = _poly_mul(O.nwt_inverse(o1[j], q, rps[j]), res(s, q), q, logn).astype(object)
        cs = _poly_mul(c[j], res(s2, q), q, logn).astype(object)
        r.append((x0 + x1 - cs) % q)
    worst = 0
    for i in range(N):
        x = 0
        for j, q in enumerate(Q):
            Mj = Qprod // q
            x += int(r[j][i]) * Mj * pow(Mj, -1, q)
        x %= Qprod
        if x > Qprod // 2:
            x -= Qprod
        worst = max(worst, abs(x))
    assert worst < 8 * N, f"key-switch noise {worst} is not small (Q has {Qprod.bit_length()} bits)"


def test_rotate_and_trace_format(F, eng):
    """ROTATE composite: decrypt(rotate(ct)) = sigma(decrypt(ct)) up to key-switch noise, and the trace it
    emits parses with the patterns of the reference's profile_framewk/build/sum_trace.py:16-19."""
    import re
    from oracle import cport as O
    from oracle.keyswitch_ref import galois_coeff
    logn, N, L, K, dnum, k = 10, 1024, 4, 2, 2, 5
    M = L + K
    qs = F.create_moduli(N, [50] * L + [61] * K)
    Q, P = qs[:L], qs[L:]
    Qprod = int(np.prod([int(x) for x in Q], dtype=object))
    Pprod = int(np.prod([int(x) for x in P], dtype=object))
    alpha = -(-L // dnum)
    rnd = random.Random(7)
    s = [rnd.choice((-1, 0, 1)) for _ in range(N)]
    rps = [O.root_powers(q, logn) for q in qs]
    res = lambda v, q: np.array([x % q for x in v], dtype=np.uint64)
    # sigma(s) as signed coefficients
    sig_s = [0] * N
    for i, v in enumerate(s):
        j = (i * k) % (2 * N)
        if j >= N:
            sig_s[j - N] = -v
        else:
            sig_s[j] = v
    gk = np.zeros((dnum, 2, M, N), dtype=np.uint64)          # Galois key: switches sigma(s) -> s
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
    # a "ciphertext": c1 random, c0 = m - c1*s  (so c0 + c1*s = m exactly)
    m = [rnd.randrange(1 << 30) for _ in range(N)]
    c1 = np.stack([np.array([rnd.randrange(q) for _ in range(N)], dtype=np.uint64) for q in Q])
    c0 = np.stack([((res(m, q).astype(object) - _poly_mul(c1[j], res(s, q), q, logn).astype(object)) % q).astype(np.uint64)
                   for j, q in enumerate(Q)])
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    d0 = eng.upload(np.stack([O.nwt_forward(c0[j], Q[j], rps[j]) for j in range(L)]))
    d1 = eng.upload(np.stack([O.nwt_forward(c1[j], Q[j], rps[j]) for j in range(L)]))
    eng.trace(True)
    o0, o1 = ks.rotate(d0, d1, k, eng.upload(gk))
    eng.trace(False)
    text = eng.trace_text()
    o0, o1 = o0.download(), o1.download()
    want = [int(x) for x in galois_coeff(np.array(m, dtype=np.uint64), k, 1 << 62)]          # sigma(m), entries m_i or 2^62 - m_i
    want = [w if w < (1 << 61) else w - (1 << 62) for w in want]
    worst = 0
    for j, q in enumerate(Q[:1]):                                    # noise is tiny: one prime suffices to see it
        x0 = O.nwt_inverse(o0[j], q, rps[j]).astype(object)
        x1 = _poly_mul(O.nwt_inverse(o1[j], q, rps[j]), res(s, q), q, logn).astype(object)
        dec = (x0 + x1) % q
        for i in range(N):
            diff = (int(dec[i]) - want[i]) % q
            diff = diff - q if diff > q // 2 else diff
            worst = max(worst, abs(diff))
    assert worst < 8 * N, worst
    # trace format (sum_trace.py: start_re, end_re, ntt_re, cost_re)
    lines = text.strip().split("\n")
    assert re.match(r"^frontend: ROTATE$", lines[0])
    assert re.match(r"^frontend: ROTATE\[(\d+)\s+microseconds\]$", lines[-1])
    assert sum(1 for l in lines if re.match(r"^evaluator: KEYSWITCH\[\d+ microseconds\]$", l)) == 1
    leaf = [l for l in lines[1:-1] if not l.startswith("evaluator:")]
    tags = [re.match(r"^\[([^\]]+)\] total cost\s+(\d+)\s+µs", l).group(1) for l in leaf]
    # one line per phase: base extension of all digits (+ their batched NTT, nested), inner product with the key, mod-down
    assert tags.count("MODREDUCTION") == 1 and tags.count("MULTEVK") == 1 and tags.count("MODSWITCH") == 1
    assert set(tags) == {"NTT", "MODREDUCTION", "MULTEVK", "MODSWITCH"}      # the tag set of profile_framewk/build/sample.txt
    # INTT of the input, one forward batch over every extended limb of every digit, INTT + NTT (both halves) in the mod-down
    assert tags.count("NTT") == 4


def test_rotate_is_stream_capture_safe(F, eng):
    """fhe_rotate / fhe_keyswitch_apply issue only asynchronous work on the caller's stream (no allocation, no
    synchronisation when tracing is off), so a caller can capture them into a HIP graph.  (Measured on MI355X:
    replaying the graph is not faster than the eager launches -- 70 vs 63 us at N = 2^14, L = 4 -- the cost is
    per kernel on the device side, which is why the engine batches launches instead.)"""
    import ctypes as C
    import torch
    from fhe_reliability_gpu_amd._lib import check, lib
    logn, L, K, dnum = 13, 3, 1, 3
    n = 1 << logn
    qk = F.create_moduli(n, [50] * (L + K))
    tk = eng.tables(logn, qk)
    ks = F.KeySwitch(eng, tk, L, K, dnum)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    c0 = torch.randint(0, qk[0], (L, n), generator=g, device="cuda", dtype=torch.int64)
    c1 = torch.randint(0, qk[0], (L, n), generator=g, device="cuda", dtype=torch.int64)
    evk = torch.randint(0, qk[0], (dnum, 2, L + K, n), generator=g, device="cuda", dtype=torch.int64)
    o0, o1 = torch.empty_like(c0), torch.empty_like(c0)
    P = lambda x: C.c_void_p(x.data_ptr())

    def call(s):
        check(lib.fhe_rotate(eng._h, ks._h, P(o0), P(o1), P(c0), P(c1), 3, P(evk), C.c_void_p(s.cuda_stream)))

    s = torch.cuda.Stream()
    call(s)
    torch.cuda.synchronize()
    want0, want1 = o0.clone(), o1.clone()
    graph, cap = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.graph(graph, stream=cap):
        call(cap)
    o0.zero_()
    o1.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert bool((o0 == want0).all()) and bool((o1 == want1).all())


@pytest.mark.parametrize("logn,L,K,dnum,bits", [(13, 4, 1, 4, 50), (14, 3, 1, 3, 61), (10, 4, 2, 2, 50), (13, 5, 2, 5, 50), (12, 4, 1, 4, 50)])
def test_keyswitch_bgv_form_and_one_limb_conversions(F, eng, logn, L, K, dnum, bits):
    """The BGV mod-down (plain modulus set: delta = t [acc t^-1]_P) against the oracle composite, including the shapes whose
    conversions convert ONE limb (dnum = L and / or K = 1 at the two-launch sizes): there x mod q_j rides on the column pass's
    load and the factor t on the fused tail, instead of a conversion launch and a scalar pass."""
    from oracle.keyswitch_ref import keyswitch_ref
    N, t_plain = 1 << logn, 65537
    qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 31 + K)
    c = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    add = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    evk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
    ks = F.KeySwitch(eng, t, L, K, dnum)
    for tp in (0, t_plain):
        ks.set_plain_modulus(tp)
        o0, o1 = ks.relinearize(eng.upload(add), eng.upload(c), eng.upload(c), eng.upload(evk))      # key switch of c, + add / + c
        w0, w1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn, add0=add, add1=c, plain_modulus=tp)
        assert (o0.download() == w0).all() and (o1.download() == w1).all(), tp



@pytest.mark.parametrize("logn,L,K,dnum,bits", [(4, 2, 1, 2, 50), (5, 3, 1, 3, 50), (8, 3, 2, 2, 61), (12, 4, 1, 2, 50), (13, 4, 1, 4, 50),
                                                (14, 5, 2, 3, 61), (16, 3, 1, 3, 50)])
def test_rotate_with_the_automorphism_on_the_loads(F, eng, logn, L, K, dnum, bits):
    """fhe_rotate applies sigma_k on the loads of the key switch's own launches (the opening INTT reads c1 through the NTT-domain
    Galois map, the mod-down's tail reads c0 through it): every word equals the oracle composite -- automorphism as a pass of
    its own, then the key switch -- for several Galois elements, on both arithmetic paths, at single-launch and two-launch sizes,
    and in the settings that take the separate-launch route (N < 2^5, experimental transform variants)."""
    from oracle.keyswitch_ref import rotate_ref
    N = 1 << logn
    qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 13 + L)
    c0 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    c1 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    gk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
    ks = F.KeySwitch(eng, t, L, K, dnum)
    d0, d1, dk = eng.upload(c0), eng.upload(c1), eng.upload(gk)
    for k in (3, 2 * N - 1, 5 if logn < 6 else (1 << (logn - 1)) + 1):
        w0, w1 = rotate_ref(c0, c1, k, gk, qs, L, K, dnum, logn)
        o0, o1 = ks.rotate(d0, d1, k, dk)
        assert (o0.download() == w0).all() and (o1.download() == w1).all(), f"galois element {k}"
        assert (d0.download() == c0).all() and (d1.download() == c1).all()          # inputs untouched
        if 13 <= logn <= 14:
            eng.set_option("ntt_resident", 1)
            try:
                r0, r1 = ks.rotate(d0, d1, k, dk)
            finally:
                eng.set_option("ntt_resident", 0)
            assert (r0.download() == w0).all() and (r1.download() == w1).all(), f"galois element {k}, separate-launch route"
    eng.check()


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("logn,L,K,dnum,bits", [(5, 3, 1, 3, 50), (10, 4, 2, 2, 61), (13, 4, 1, 2, 61), (14, 5, 2, 3, 50)])
def test_both_forms_of_the_inner_product(F, eng, logn, L, K, dnum, bits, fused):
    """"ks_fused" 0 / 1 forces the key switch's inner product to run as a launch of its own after the extended limbs' row pass, or
    fused with that row pass (the library picks by shape, so small shapes never see the fused form and the BASELINE shapes never see
    the other one): both equal the oracle composite, on both arithmetic paths and with special primes of the other path."""
    from oracle.keyswitch_ref import keyswitch_ref
    N = 1 << logn
    qs = F.create_moduli(N, [bits] * L + [50 if bits == 61 else 61] * K)       # special primes on the OTHER arithmetic path
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(logn * 3 + fused)
    c = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    add = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    evk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
    ks = F.KeySwitch(eng, t, L, K, dnum)
    eng.set_option("ks_fused", fused)
    try:
        o0, o1 = ks.apply(eng.upload(c), eng.upload(evk))
        w0, w1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn)
        assert (o0.download() == w0).all() and (o1.download() == w1).all()
        d0, d1, d2 = eng.upload(add), eng.upload(c), eng.upload(c)
        r0, r1 = ks.relinearize(d0, d1, d2, eng.upload(evk))                  # addends on both parts
        v0, v1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn, add0=add, add1=c)
        assert (r0.download() == v0).all() and (r1.download() == v1).all()
    finally:
        eng.set_option("ks_fused", -1)
    eng.check()


@pytest.mark.parametrize("bits", [50, 61])
def test_wide_tile_geometry_of_the_2_16_transform(F, eng, bits):
    """"tile_geo" 0 (the widest column tile, kept for 2^16 tuning runs) gives the same words as the default geometry."""
    from oracle import cport as O
    logn, N, n_poly = 16, 1 << 16, 2
    qs = F.create_moduli(N, [bits])
    t = eng.tables(logn, qs)
    rng = np.random.default_rng(bits)
    data = rng.integers(0, qs[0], (n_poly, 1, N), dtype=np.uint64)
    eng.set_option("tile_geo", 0)
    try:
        d = eng.upload(data)
        t.forward(d, n_poly=n_poly)
        got = d.download().reshape(data.shape)
        rp = O.root_powers(qs[0], logn)
        for p in range(n_poly):
            assert (got[p, 0] == O.nwt_forward(data[p, 0], qs[0], rp)).all()
        t.inverse(d, n_poly=n_poly)
        assert (d.download().reshape(data.shape) == data).all()
    finally:
        eng.set_option("tile_geo", 1)
    eng.check()
