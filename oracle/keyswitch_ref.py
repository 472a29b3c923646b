"""CPU restatement of the key-switch / rotation / homomorphic-multiply composites (TEST INFRASTRUCTURE ONLY).

The reference has no source for these (SEAL/Phantom are absent); what it holds is the operation sequence of
KEYSWITCH + MODSWITCH, ROTATE, MULTIPLY_CKKS and RELIN in its SEAL traces
(profile_framewk/build/data/ckks/16384_4:388-539, summarised by profile_framewk/build/sum_trace.py:10-94) and the
call sequence multiply -> relinearize -> mod_switch -> rotate of reliability_test/dotprod_test.cu:113-115,143-148.
This file states those sequences with the oracle's pinned primitives (oracle/cport.py) so the GPU composites can be
checked word for word at any size; their VALUES have no reference counterpart ("parity unpinned" against SEAL), so
tests/ also check the defining algebraic properties (out0 + out1*s ~ c*s', decrypt(hmult) ~ m1*m2) with big integers.
"""
import numpy as np

from . import cport as O


def galois_coeff(a, k, q):
    """x -> x^k on a coefficient vector mod (x^N + 1, q)."""
    a = np.asarray(a, dtype=np.uint64)
    N = a.size
    j = (np.arange(N, dtype=np.uint64) * np.uint64(k)) % np.uint64(2 * N)
    v = a % np.uint64(q)
    neg = j >= N
    out = np.zeros(N, dtype=np.uint64)
    out[(j % np.uint64(N)).astype(np.int64)] = np.where(neg, (np.uint64(q) - v) % np.uint64(q), v)
    return out


def _tables(qs, logn):
    return np.stack([O.root_powers(q, logn) for q in qs])


def keyswitch_ref(c, evk, qs, L, K, dnum, logn, add0=None, add1=None, rps=None, plain_modulus=0, sigma_ext=0):
    """c: (L, N) NTT domain; evk: (dnum, 2, L+K, N) NTT domain; returns (out0, out1), each (L, N).
    add0 / add1: optional (L, N) terms added to the outputs (rotation: sigma(c0); relinearisation: d0, d1).
    sigma_ext (a Galois element; hoisted rotations): the automorphism is applied to every digit's extension AFTER it was formed
    from the un-rotated input, ext_d <- sigma(ext_d), instead of to the input (the decomposition is then shared by all elements).
    plain_modulus t (BGV form of the mod-down, the scheme of reliability_test/dotprod_test.cu:199-204): the removed part is
    t * [acc t^-1]_P instead of [acc]_P, so that it vanishes modulo t."""
    M, N = L + K, 1 << logn
    alpha = -(-L // dnum)
    qs = [int(q) for q in qs]
    rps = _tables(qs, logn) if rps is None else rps
    c = np.asarray(c, dtype=np.uint64)
    coef = O.nwt_inverse_batch(c, qs[:L], rps[:L])                            # INTT of the input limbs
    acc = np.zeros((2, M, N), dtype=np.uint64)
    for d in range(dnum):
        lo, hi = d * alpha, min(L, (d + 1) * alpha)
        other = [j for j in range(M) if j < lo or j >= hi]
        conv = O.baseconv_exact(coef[lo:hi], qs[lo:hi], [qs[j] for j in other])               # MODREDUCTION
        ext = np.zeros((M, N), dtype=np.uint64)
        ext[other] = O.nwt_forward_batch(conv, [qs[j] for j in other], rps[other])
        ext[lo:hi] = c[lo:hi]
        if sigma_ext:
            co = O.nwt_inverse_batch(ext, qs, rps)
            co = np.stack([galois_coeff(co[j], sigma_ext, qs[j]) for j in range(M)])
            ext = O.nwt_forward_batch(co, qs, rps)
        for h in range(2):                                                                     # MULTEVALK
            acc[h] = O.modmul_batch(ext, evk[d, h], qs, acc=acc[h])
    outs = []
    P, Q = qs[L:], qs[:L]
    for h in range(2):                                                                         # MODSWITCH
        tP = O.nwt_inverse_batch(acc[h, L:], P, rps[L:])
        if plain_modulus:
            tP = np.stack([O.modmul(tP[k], np.full(N, pow(plain_modulus % P[k], -1, P[k]), dtype=np.uint64), P[k]) for k in range(K)])
        conv = O.baseconv_exact(tP, P, Q)
        if plain_modulus:
            conv = np.stack([O.modmul(conv[j], np.full(N, plain_modulus % Q[j], dtype=np.uint64), Q[j]) for j in range(L)])
        cn = O.nwt_forward_batch(conv, Q, rps[:L])
        out = np.zeros((L, N), dtype=np.uint64)
        add = (add0, add1)[h]
        for j in range(L):
            pm = 1
            for pk in P:
                pm = pm * (pk % Q[j]) % Q[j]
            pinv = np.full(N, pow(pm, -1, Q[j]), dtype=np.uint64)
            qj = np.uint64(Q[j])
            diff = (acc[h, j] + (qj - cn[j])) % qj
            out[j] = O.modmul(diff, pinv, Q[j])
            if add is not None:
                out[j] = (out[j] + np.asarray(add[j], dtype=np.uint64) % qj) % qj
        outs.append(out)
    return outs[0], outs[1]


def rotate_ref(c0, c1, k, gk, qs, L, K, dnum, logn):
    """ROTATE (16384_4:452-539 / rotate_inplace, dotprod_test.cu:146): sigma_k on both parts, key switch of sigma(c1)
    with the Galois key, plus sigma(c0).  sigma is applied in the coefficient domain here (the engine permutes the
    NTT-domain slots instead)."""
    qs = [int(q) for q in qs]
    rps = _tables(qs, logn)

    def sigma(x):
        co = O.nwt_inverse_batch(np.asarray(x, dtype=np.uint64), qs[:L], rps[:L])
        co = np.stack([galois_coeff(co[l], k, qs[l]) for l in range(L)])
        return O.nwt_forward_batch(co, qs[:L], rps[:L])

    return keyswitch_ref(sigma(c1), gk, qs, L, K, dnum, logn, add0=sigma(c0), rps=rps)


def rotate_hoisted_ref(c0, c1, k, gk, qs, L, K, dnum, logn):
    """One rotation out of a hoisted batch (fhe_rotate_hoisted): the digits of the UN-rotated c1 are extended once, sigma_k is applied to
    the extended digits, then inner product with the (standard) Galois key, mod-down, plus sigma(c0).  Differs from rotate_ref word by
    word -- the exact extension lifts a digit to [0, P_d), which sigma's sign flips do not preserve -- and decrypts to the same value."""
    qs = [int(q) for q in qs]
    rps = _tables(qs, logn)
    co = O.nwt_inverse_batch(np.asarray(c0, dtype=np.uint64), qs[:L], rps[:L])
    co = np.stack([galois_coeff(co[l], k, qs[l]) for l in range(L)])
    sig0 = O.nwt_forward_batch(co, qs[:L], rps[:L])
    return keyswitch_ref(c1, gk, qs, L, K, dnum, logn, add0=sig0, rps=rps, sigma_ext=k)


def bsgs_matvec_ref(c0, c1, diags, baby_elts, baby_keys, giant_elts, giant_keys, qs, L, K, dnum, logn):
    """fhe_bsgs_matvec: y = sum_g sigma_G_g( sum_b diag[g][b] * sigma_B_b(x) ) (profile_framewk/src/matmul_ckks.cpp:45-113, the
    rotate / multiply_plain / add sequence in the n1 + n2 - 2 arrangement).  diags: (n2, n1, L, N); baby rotations hoisted
    (rotate_hoisted_ref), giant ones plain (rotate_ref); keys in their standard form."""
    n2, n1 = diags.shape[0], diags.shape[1]
    Q = [int(q) for q in qs[:L]]
    R = [(np.asarray(c0, dtype=np.uint64), np.asarray(c1, dtype=np.uint64))]
    for b in range(1, n1):
        R.append(rotate_hoisted_ref(c0, c1, baby_elts[b - 1], baby_keys[b - 1], qs, L, K, dnum, logn))
    y = None
    for g in range(n2):
        inner = []
        for h in range(2):
            s = None
            for b in range(n1):
                s = O.modmul_batch(diags[g, b], R[b][h], Q, acc=s)
            inner.append(s)
        if g:
            inner = rotate_ref(inner[0], inner[1], giant_elts[g - 1], giant_keys[g - 1], qs, L, K, dnum, logn)
        if y is None:
            y = [inner[0].copy(), inner[1].copy()]
        else:
            for h in range(2):
                y[h] = np.stack([(y[h][l] + inner[h][l]) % np.uint64(Q[l]) for l in range(L)])
    return y[0], y[1]


def tensor_ref(a0, a1, b0, b1, qs):
    """MULTIPLY_CKKS (16384_4:388-389 / multiply, dotprod_test.cu:113), NTT domain, per limb."""
    L = a0.shape[0]
    qs = [int(q) for q in qs[:L]]
    d0 = O.modmul_batch(a0, b0, qs)
    d1 = O.modmul_batch(a1, b0, qs, acc=O.modmul_batch(a0, b1, qs))
    d2 = O.modmul_batch(a1, b1, qs)
    return d0, d1, d2


def rescale_ref(parts, qs, L, logn, plain_modulus=0):
    """mod_switch_to_next (dotprod_test.cu:115): parts (n, L, N) NTT domain -> (n, L-1, N);
    c' = (c - delta) / q_last with delta = [c]_{q_last} (plain_modulus = 0) or t [c t^-1]_{q_last} (BGV)."""
    qs = [int(q) for q in qs[:L]]
    ql, N = qs[L - 1], 1 << logn
    rps = _tables(qs, logn)
    out = []
    for c in parts:
        c = np.asarray(c, dtype=np.uint64)
        y = O.nwt_inverse(c[L - 1], ql, rps[L - 1])
        if plain_modulus:
            y = O.modmul(y, np.full(N, pow(plain_modulus % ql, -1, ql), dtype=np.uint64), ql)
        delta = np.stack([y % np.uint64(q) for q in qs[:L - 1]])
        if plain_modulus:
            delta = np.stack([O.modmul(delta[j], np.full(N, plain_modulus % qs[j], dtype=np.uint64), qs[j]) for j in range(L - 1)])
        dn = O.nwt_forward_batch(delta, qs[:L - 1], rps[:L - 1])
        r = np.zeros((L - 1, N), dtype=np.uint64)
        for j in range(L - 1):
            qj = np.uint64(qs[j])
            r[j] = O.modmul((c[j] % qj + (qj - dn[j])) % qj, np.full(N, pow(ql % qs[j], -1, qs[j]), dtype=np.uint64), qs[j])
        out.append(r)
    return np.stack(out)


def hmult_ref(a0, a1, b0, b1, rlk, qs, L, K, dnum, logn, rescale=True, plain_modulus=0):
    """multiply -> relinearize -> mod_switch_to_next (dotprod_test.cu:113-115)."""
    d0, d1, d2 = tensor_ref(a0, a1, b0, b1, qs)
    c0, c1 = keyswitch_ref(d2, rlk, qs, L, K, dnum, logn, add0=d0, add1=d1)
    if not rescale:
        return c0, c1
    r = rescale_ref([c0, c1], qs, L, logn, plain_modulus)
    return r[0], r[1]
