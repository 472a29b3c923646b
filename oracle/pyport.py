"""Pure-Python restatement of the reference's Python algorithms (TEST INFRASTRUCTURE ONLY).

Each function cites the reference file:line it follows.  Python big ints, so every
result is exact; use for small cases, for validating ``fhe_oracle.c`` and as the
``cpu_python`` baseline timed by bench.py.
"""
from __future__ import annotations

import math
from typing import List, Sequence


def bit_reverse(x: int, bits: int) -> int:
    """rfhe_framewk/src/negaclic_ntt.py:16-21."""
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


# ---------------------------------------------------------------- a1 cyclic NTT
def ntt_cyclic(a: Sequence[int], mod: int, root: int) -> List[int]:
    """motivation/ntt.py:8-32 -- ``root`` is a generator of Z_mod*; natural order in/out."""
    a = list(a)
    n = len(a)
    j = 0
    for i in range(1, n):  # :10-18 bit-reversal permutation
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j ^= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:  # :20-31 butterfly stages
        wlen = pow(root, (mod - 1) // length, mod)
        half = length // 2
        for i in range(0, n, length):
            w = 1
            for k in range(half):
                u = a[i + k]
                v = a[i + k + half] * w % mod
                a[i + k] = (u + v) % mod
                a[i + k + half] = (u - v + mod) % mod
                w = w * wlen % mod
        length *= 2
    return a


def intt_cyclic(a: Sequence[int], mod: int, root: int) -> List[int]:
    """motivation/bsgs.py:31-36."""
    n = len(a)
    inv_root = pow(root, mod - 2, mod)
    r = ntt_cyclic(a, mod, inv_root)
    inv_n = pow(n, mod - 2, mod)
    return [x * inv_n % mod for x in r]


def ntt_nthroot(a: Sequence[int], root: int, mod: int) -> List[int]:
    """rfhe_framewk/src/negaclic_ntt.py:38-57 -- ``root`` is a primitive n-th root."""
    n = len(a)
    bits = n.bit_length() - 1
    a = [a[bit_reverse(i, bits)] for i in range(n)]
    length = 2
    while length <= n:
        wlen = pow(root, n // length, mod)
        half = length // 2
        for start in range(0, n, length):
            w = 1
            for k in range(half):
                u = a[start + k]
                v = a[start + k + half] * w % mod
                a[start + k] = (u + v) % mod
                a[start + k + half] = (u - v) % mod
                w = w * wlen % mod
        length <<= 1
    return a


def intt_nthroot(a: Sequence[int], root: int, mod: int) -> List[int]:
    """rfhe_framewk/src/negaclic_ntt.py:77-83."""
    n = len(a)
    inv_n = pow(n, mod - 2, mod)
    r = ntt_nthroot(a, pow(root, mod - 2, mod), mod)
    return [x * inv_n % mod for x in r]


# ------------------------------------------------------- a2/a3/a5 negacyclic
def negacyclic_ntt(a: Sequence[int], psi: int, mod: int) -> List[int]:
    """rfhe_framewk/src/negaclic_ntt.py:86-92 (natural-order output)."""
    n = len(a)
    a_pw = [a[i] * pow(psi, i, mod) % mod for i in range(n)]
    return ntt_nthroot(a_pw, pow(psi, 2, mod), mod)


def negacyclic_intt(A: Sequence[int], psi: int, mod: int) -> List[int]:
    """rfhe_framewk/src/negaclic_ntt.py:102-109."""
    n = len(A)
    inv_A = intt_nthroot(A, pow(psi, 2, mod), mod)
    psi_inv = pow(psi, mod - 2, mod)
    return [inv_A[i] * pow(psi_inv, i, mod) % mod for i in range(n)]


def nwt_forward(a: Sequence[int], q: int, rp: Sequence[int]) -> List[int]:
    """SURVEY appendix A4 -- the ordering nwt_2d_radix8_forward_inplace
    (reliability_test/ntt_test.cu:95) produces: natural in, bit-reversed out."""
    a = [x % q for x in a]
    n = len(a)
    t = n
    m = 1
    while m < n:
        t >>= 1
        for i in range(m):
            S = rp[m + i]
            j1 = 2 * i * t
            for j in range(j1, j1 + t):
                U = a[j]
                V = a[j + t] * S % q
                a[j] = (U + V) % q
                a[j + t] = (U - V) % q
        m <<= 1
    return a


def root_powers(q: int, logn: int, psi: int) -> List[int]:
    """rp[bitrev(i)] = psi^i (NTT::get_from_root_powers, ntt_test.cu:60-64)."""
    n = 1 << logn
    rp = [0] * n
    p = 1
    for i in range(n):
        rp[bit_reverse(i, logn)] = p
        p = p * psi % q
    return rp


def poly_mul_naive_negacyclic(a, b, mod):
    """rfhe_framewk/src/negaclic_ntt.py:112-120."""
    n = len(a)
    res = [0] * n
    for i in range(n):
        for j in range(n):
            k = (i + j) % n
            sign = mod - 1 if (i + j) >= n else 1
            res[k] = (res[k] + a[i] * b[j] * sign) % mod
    return res


def poly_mul_negacyclic_ntt(a, b, psi, mod):
    """rfhe_framewk/src/negaclic_ntt.py:123-127."""
    A = negacyclic_ntt(a, psi, mod)
    B = negacyclic_ntt(b, psi, mod)
    C = [A[i] * B[i] % mod for i in range(len(a))]
    return negacyclic_intt(C, psi, mod)


# --------------------------------------------------------------- a6 four-step
def four_step_ntt(a: Sequence[int], n1: int, n2: int, mod: int, g: int) -> List[int]:
    """reliability_test/four_step_ntt_prot.py:71-109, n1 != n2 allowed."""
    N = n1 * n2
    w = pow(g, (mod - 1) // N, mod)
    w_n1 = pow(w, n1, mod)
    w_n2 = pow(w, n2, mod)
    B = [[sum(a[t1 + n1 * t2] * pow(w_n1, k2 * t2, mod) for t2 in range(n2)) % mod
          for k2 in range(n2)] for t1 in range(n1)]
    C = [[B[t1][k2] * pow(w, k2 * t1, mod) % mod for k2 in range(n2)] for t1 in range(n1)]
    y = [0] * N
    for k2 in range(n2):
        for k1 in range(n1):
            y[k1 * n2 + k2] = sum(C[t1][k2] * pow(w_n2, k1 * t1, mod) for t1 in range(n1)) % mod
    return y


# ------------------------------------------------------------------ a7 Barrett
def barrett_ctx(q: int):
    """rfhe_framewk/src/barrett_final.cpp:68-79; rfhe_framewk/src/barrett_inner.py:38-41."""
    K = (q - 1).bit_length()
    return K, (1 << (2 * K)) // q


def barrett_reduce(t: int, q: int, K: int, mu: int) -> int:
    """rfhe_framewk/src/barrett_final.cpp:120-141."""
    s = (t * mu) >> (2 * K)
    c = t - s * q
    while c >= q:
        c -= q
    return c


# ------------------------------------------------------------- a8 base conversion
def base_conv_fixed(residue_arrays, moduli_in, moduli_out):
    """motivation/baseConv.py:67-83 (exact); returns [k][i]."""
    m = len(moduli_in)
    N = len(residue_arrays[0])
    P = math.prod(moduli_in)
    hat_p = [P // p for p in moduli_in]
    inv_hat_p = [pow(hat_p[j] % moduli_in[j], -1, moduli_in[j]) for j in range(m)]
    by_elem = []
    for i in range(N):
        x = 0
        for j in range(m):
            x += residue_arrays[j][i] * hat_p[j] * inv_hat_p[j]
        x %= P
        by_elem.append([x % q for q in moduli_out])
    return [[by_elem[i][k] for i in range(N)] for k in range(len(moduli_out))]


def bconv_fast(residue_arrays, moduli, moduli_out):
    """rfhe_framewk/src/baseConv.py:10-40 (unreduced sums); returns [i][k]."""
    m = len(moduli)
    N = len(residue_arrays[0])
    P = math.prod(moduli)
    hat_p = [P // p for p in moduli]
    inv_hat_p = [pow(hat_p[j], -1, moduli[j]) for j in range(m)]
    out = []
    for i in range(N):
        out.append([sum((int(residue_arrays[j][i]) * hat_p[j] * inv_hat_p[j]) % q
                        for j in range(m)) for q in moduli_out])
    return out


def crt_garner(residues, moduli):
    """rfhe_framewk/src/baseConv.cu:85-120, 157-169 -- returns list of (lo, hi),
    every product/sum wrapped to 128 bits as the kernel's unsigned __int128."""
    M128 = (1 << 128) - 1
    m = len(moduli)
    N = len(residues[0])
    pref = [1] * m
    for j in range(1, m):
        pref[j] = (pref[j - 1] * moduli[j - 1]) & M128
    inv = [0] * m
    for j in range(1, m):
        inv[j] = pow(pref[j] % moduli[j], -1, moduli[j])
    out = []
    for i in range(N):
        c = [0] * m
        c[0] = residues[0][i]
        for j in range(1, m):
            t = residues[j][i] % moduli[j]
            for k in range(j):
                r = ((c[k] * pref[k]) & M128) % moduli[j]
                t = (t + moduli[j] - r) % moduli[j]
            c[j] = t * inv[j] % moduli[j]
        x = 0
        for k in range(m):
            x = (x + c[k] * pref[k]) & M128
        out.append((x & ((1 << 64) - 1), x >> 64))
    return out


# -------------------------------------------------------------------- a9 BSGS
def diag_block_hadamard_matvec(M_blocks, v):
    """motivation/bsgs.py:39-52; int64 wrap-around like NumPy's dtype=int."""
    bs = len(M_blocks[0])
    k = len(M_blocks)
    mask = (1 << 64) - 1
    y = [0] * (bs * k)
    for i in range(k):
        for e in range(bs):
            acc = 0
            for j in range(k):
                acc += int(M_blocks[(j - i) % k][e]) * int(v[j * bs + e])
            acc &= mask
            y[i * bs + e] = acc - (1 << 64) if acc >> 63 else acc
    return y


# ---------------------------------------------------------------- a10 tables
def is_prime(n: int) -> bool:
    """motivation/baseConv.py:10-36."""
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23):
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 325, 9375, 28178, 450775, 9780504, 1795265022):
        if a % n == 0:
            continue
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = pow(x, 2, n)
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_primes(N: int, bits: int, count: int) -> List[int]:
    """CoeffModulus::Create rule (ntt_test.cu:44)."""
    factor = 2 * N
    v = ((1 << bits) - 1) // factor * factor + 1
    found = []
    while len(found) < count and v > (1 << (bits - 1)):
        if is_prime(v):
            found.append(v)
        v -= factor
    return found[::-1]


def min_primitive_root(q: int, order: int) -> int:
    """SURVEY appendix A3."""
    assert (q - 1) % order == 0
    g = None
    c = 2
    while g is None:
        r = pow(c, (q - 1) // order, q)
        if pow(r, order // 2, q) == q - 1:
            g = r
        c += 1
    g2 = g * g % q
    cur = best = g
    for _ in range(order // 2 - 1):
        cur = cur * g2 % q
        if cur < best:
            best = cur
    return best
