"""CPU restatement of the key-switch composite (TEST INFRASTRUCTURE ONLY).

The reference has no key-switch source (SEAL/Phantom are absent); what it holds is the
operation sequence of one KEYSWITCH + MODSWITCH in its SEAL traces
(profile_framewk/build/data/ckks/16384_4:466-539, summarised by
profile_framewk/build/sum_trace.py:10-94).  This file states that sequence with the
oracle's primitives so the GPU composite can be checked word for word; its VALUES have
no reference counterpart ("parity unpinned" against SEAL), so tests/ also check the
defining algebraic property (out0 + out1*s ~ c*s') with big integers.
"""
import numpy as np

from . import cport as O


def galois_coeff(a, k, q):
    """x -> x^k on a coefficient vector mod (x^N + 1, q)."""
    N = len(a)
    out = np.zeros(N, dtype=np.uint64)
    for i in range(N):
        j = (i * k) % (2 * N)
        v = int(a[i]) % q
        if j >= N:
            out[j - N] = (q - v) % q
        else:
            out[j] = v
    return out


def keyswitch_ref(c, evk, qs, L, K, dnum, logn):
    """c: (L, N) NTT domain; evk: (dnum, 2, L+K, N) NTT domain; returns (out0, out1), each (L, N)."""
    M, N = L + K, 1 << logn
    alpha = -(-L // dnum)
    rps = [O.root_powers(q, logn) for q in qs]
    coef = [O.nwt_inverse(c[l], qs[l], rps[l]) for l in range(L)]             # INTT of the input limbs
    acc = np.zeros((2, M, N), dtype=np.uint64)
    for d in range(dnum):
        lo, hi = d * alpha, min(L, (d + 1) * alpha)
        other = [j for j in range(M) if j < lo or j >= hi]
        conv = O.baseconv_exact(np.stack(coef[lo:hi]), qs[lo:hi], [qs[j] for j in other])   # MODREDUCTION
        ext = [None] * M
        for pos, j in enumerate(other):
            ext[j] = O.nwt_forward(conv[pos], qs[j], rps[j])
        for j in range(lo, hi):
            ext[j] = np.asarray(c[j], dtype=np.uint64)
        for h in range(2):
            for j in range(M):                                                               # MULTEVALK
                acc[h, j] = O.modmul_acc(acc[h, j], ext[j], evk[d, h, j], qs[j])
    outs = []
    P, Q = qs[L:], qs[:L]
    for h in range(2):                                                                       # MODSWITCH
        tP = np.stack([O.nwt_inverse(acc[h, L + k], P[k], rps[L + k]) for k in range(K)])
        conv = O.baseconv_exact(tP, P, Q)
        out = np.zeros((L, N), dtype=np.uint64)
        for j in range(L):
            cn = O.nwt_forward(conv[j], Q[j], rps[j])
            pm = 1
            for pk in P:
                pm = pm * (pk % Q[j]) % Q[j]
            pinv = pow(pm, -1, Q[j])
            diff = (acc[h, j].astype(object) - cn.astype(object)) % Q[j]
            out[j] = ((diff * pinv) % Q[j]).astype(np.uint64)
        outs.append(out)
    return outs[0], outs[1]
