/*
 * fhe_oracle.c -- CPU restatement of the reference's NTT / RNS arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fhe_reliability_gpu_amd/ may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to the reference
 * repository root) whose algorithm it restates.  Integer arithmetic only
 * (unsigned __int128 for products), so results are bit-exact by construction.
 *
 * Pinning (see tests/test_oracle_golden.py):
 *   - cyclic NTT, negacyclic NTT/INTT, polymul, four-step, base conversion and
 *     the BSGS Hadamard are checked against golden vectors produced by importing
 *     the reference's own Python (tests/golden/make_golden.py);
 *   - prime selection is checked against reliability_test/data/bits1-16_num1.txt:10;
 *   - the SEAL/Phantom-ordered forward transform (orc_nwt_forward) has no reference
 *     output to compare with (libPhantom.so is absent): it is pinned through its
 *     identity with negacyclic_ntt() read at bit-reversed indices and the KATs of
 *     SURVEY.md appendix A4; Phantom's own output VALUES remain "parity unpinned".
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ */
/* scalar helpers                                                      */
/* ------------------------------------------------------------------ */

u64 orc_mulmod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }

u64 orc_addmod(u64 a, u64 b, u64 m) { return (u64)(((u128)a + b) % m); }

u64 orc_submod(u64 a, u64 b, u64 m) { return (u64)(((u128)a + m - (b % m)) % m); }

u64 orc_powmod(u64 b, u64 e, u64 m)
{
    u64 r = 1 % m;
    b %= m;
    while (e) {
        if (e & 1) r = orc_mulmod(r, b, m);
        b = orc_mulmod(b, b, m);
        e >>= 1;
    }
    return r;
}

/* modular inverse by Fermat (all moduli on this path are prime) or by
 * extended Euclid when they are not (motivation/baseConv.py:56-57 uses pow(a,-1,m)) */
u64 orc_invmod(u64 a, u64 m)
{
    __int128 t = 0, nt = 1, r = m, nr = a % m;
    while (nr != 0) {
        __int128 q = r / nr, tmp;
        tmp = t - q * nt; t = nt; nt = tmp;
        tmp = r - q * nr; r = nr; nr = tmp;
    }
    if (r != 1) return 0; /* not invertible */
    if (t < 0) t += m;
    return (u64)t;
}

/* deterministic Miller-Rabin for 64-bit n; witness set as in
 * motivation/baseConv.py:10-36 (is_prime) */
int orc_is_prime(u64 n)
{
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23};
    static const u64 wit[] = {2, 325, 9375, 28178, 450775, 9780504, 1795265022};
    if (n < 2) return 0;
    for (unsigned i = 0; i < sizeof small / sizeof *small; i++)
        if (n % small[i] == 0) return n == small[i];
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; s++; }
    for (unsigned i = 0; i < sizeof wit / sizeof *wit; i++) {
        u64 a = wit[i] % n;
        if (a == 0) continue;
        u64 x = orc_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

static unsigned bitrev(unsigned x, int bits)
{
    unsigned r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

unsigned orc_bitrev(unsigned x, int bits) { return bitrev(x, bits); }

/* ------------------------------------------------------------------ */
/* a10: moduli + twiddle tables                                        */
/* ------------------------------------------------------------------ */

/* CoeffModulus::Create(N, {bits...}) as called at reliability_test/ntt_test.cu:44.
 * Rule (verified against reliability_test/data/bits1-16_num1.txt:10):
 * for one bit size, walk v = floor((2^b - 1)/2N)*2N + 1, v-2N, ... while v > 2^(b-1),
 * keep the first `count` primes, hand them out smallest first.
 * Returns number written (== count on success). */
int orc_gen_primes(u64 N, int bits, int count, u64 *out)
{
    u64 factor = 2 * N;
    u64 hi = ((u64)1 << bits) - 1;
    u64 lo = (u64)1 << (bits - 1);
    u64 v = hi / factor * factor + 1;
    int got = 0;
    u64 *tmp = (u64 *)malloc(sizeof(u64) * (size_t)count);
    while (got < count && v > lo) {
        if (orc_is_prime(v)) tmp[got++] = v;
        v -= factor;
    }
    for (int i = 0; i < got; i++) out[i] = tmp[got - 1 - i];
    free(tmp);
    return got;
}

/* smallest primitive `order`-th root of unity mod prime q (order a power of two
 * dividing q-1).  SURVEY appendix A3: minimise over the odd powers of any
 * primitive root of that order. */
u64 orc_min_primitive_root(u64 q, u64 order)
{
    if ((q - 1) % order) return 0;
    u64 g = 0;
    for (u64 c = 2; c < q; c++) {
        u64 r = orc_powmod(c, (q - 1) / order, q);
        if (orc_powmod(r, order / 2, q) == q - 1) { g = r; break; }
    }
    if (!g) return 0;
    u64 g2 = orc_mulmod(g, g, q), cur = g, best = g;
    for (u64 k = 1; k < order / 2; k++) {
        cur = orc_mulmod(cur, g2, q);
        if (cur < best) best = cur;
    }
    return best;
}

/* rp[bitrev(i, logN)] = psi^i ; shoup[k] = floor(rp[k] * 2^64 / q)
 * (NTT::get_from_root_powers[_shoup], reliability_test/ntt_test.cu:60-64) */
void orc_root_powers(u64 q, int logN, u64 psi, u64 *rp, u64 *shoup)
{
    u64 N = (u64)1 << logN, p = 1;
    for (u64 i = 0; i < N; i++) {
        unsigned k = bitrev((unsigned)i, logN);
        rp[k] = p;
        if (shoup) shoup[k] = (u64)(((u128)p << 64) / q);
        p = orc_mulmod(p, psi, q);
    }
}

/* ------------------------------------------------------------------ */
/* a1: cyclic NTT, motivation semantics                                */
/* ------------------------------------------------------------------ */

/* motivation/ntt.py:8-32: in-place bit-reversal swap loop (:10-18) followed by
 * DIT stages with wlen = root^((mod-1)/len) (:22) and a running twiddle (:30).
 * Natural order in and out; valid for any modulus (the demo uses a composite). */
void orc_ntt_cyclic(u64 *a, u64 n, u64 mod, u64 root)
{
    u64 j = 0;
    for (u64 i = 1; i < n; i++) {
        u64 bit = n >> 1;
        while (j & bit) { j ^= bit; bit >>= 1; }
        j ^= bit;
        if (i < j) { u64 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (u64 len = 2; len <= n; len <<= 1) {
        u64 wlen = orc_powmod(root, (mod - 1) / len, mod);
        for (u64 i = 0; i < n; i += len) {
            u64 w = 1 % mod;
            for (u64 k = 0; k < len / 2; k++) {
                u64 u = a[i + k] % mod;
                u64 v = orc_mulmod(a[i + k + len / 2], w, mod);
                a[i + k] = orc_addmod(u, v, mod);
                a[i + k + len / 2] = orc_submod(u, v, mod);
                w = orc_mulmod(w, wlen, mod);
            }
        }
    }
}

/* rfhe_framewk/src/negaclic_ntt.py:38-57 (twin rfhe_framewk/src/ntt.py:38-55):
 * same flow, but `root` is already a primitive n-th root: wlen = root^(n/len) (:46);
 * out-of-place bit-reverse copy (:42). */
void orc_ntt_cyclic_nthroot(u64 *a, u64 n, u64 root, u64 mod)
{
    int bits = 0;
    while (((u64)1 << bits) < n) bits++;
    u64 *b = (u64 *)malloc(sizeof(u64) * n);
    for (u64 i = 0; i < n; i++) b[i] = a[bitrev((unsigned)i, bits)] % mod;
    for (u64 len = 2; len <= n; len <<= 1) {
        u64 wlen = orc_powmod(root, n / len, mod);
        for (u64 s = 0; s < n; s += len) {
            u64 w = 1 % mod, half = len / 2;
            for (u64 k = 0; k < half; k++) {
                u64 u = b[s + k];
                u64 v = orc_mulmod(b[s + k + half], w, mod);
                b[s + k] = orc_addmod(u, v, mod);
                b[s + k + half] = orc_submod(u, v, mod);
                w = orc_mulmod(w, wlen, mod);
            }
        }
    }
    memcpy(a, b, sizeof(u64) * n);
    free(b);
}

/* rfhe_framewk/src/negaclic_ntt.py:77-83 / motivation/bsgs.py:31-36:
 * forward transform with root^-1, then scale by n^-1. */
void orc_intt_cyclic_nthroot(u64 *a, u64 n, u64 root, u64 mod)
{
    u64 inv_n = orc_powmod(n % mod, mod - 2, mod);
    u64 inv_root = orc_powmod(root, mod - 2, mod);
    orc_ntt_cyclic_nthroot(a, n, inv_root, mod);
    for (u64 i = 0; i < n; i++) a[i] = orc_mulmod(a[i], inv_n, mod);
}

/* motivation/bsgs.py:31-36: intt(a, mod, root) with the generator convention of
 * motivation/ntt.py (root is a generator, inverse generator = root^(mod-2)). */
void orc_intt_cyclic(u64 *a, u64 n, u64 mod, u64 root)
{
    u64 inv_root = orc_powmod(root, mod - 2, mod);
    u64 inv_n = orc_powmod(n % mod, mod - 2, mod);
    orc_ntt_cyclic(a, n, mod, inv_root);
    for (u64 i = 0; i < n; i++) a[i] = orc_mulmod(a[i], inv_n, mod);
}

/* ------------------------------------------------------------------ */
/* a2/a3/a5: negacyclic transforms                                     */
/* ------------------------------------------------------------------ */

/* rfhe_framewk/src/negaclic_ntt.py:86-92: pre-weight by psi^i, then cyclic NTT
 * with root psi^2.  Natural-order output. */
void orc_negacyclic_ntt_natural(u64 *a, u64 n, u64 psi, u64 mod)
{
    u64 p = 1 % mod;
    for (u64 i = 0; i < n; i++) {
        a[i] = orc_mulmod(a[i] % mod, p, mod);
        p = orc_mulmod(p, psi, mod);
    }
    orc_ntt_cyclic_nthroot(a, n, orc_mulmod(psi, psi, mod), mod);
}

/* rfhe_framewk/src/negaclic_ntt.py:102-109: inverse cyclic with psi^2, then
 * post-weight by psi^-i. */
void orc_negacyclic_intt_natural(u64 *a, u64 n, u64 psi, u64 mod)
{
    orc_intt_cyclic_nthroot(a, n, orc_mulmod(psi, psi, mod), mod);
    u64 psi_inv = orc_powmod(psi, mod - 2, mod), p = 1 % mod;
    for (u64 i = 0; i < n; i++) {
        a[i] = orc_mulmod(a[i], p, mod);
        p = orc_mulmod(p, psi_inv, mod);
    }
}

/* The result nwt_2d_radix8_forward_inplace (reliability_test/ntt_test.cu:95,144)
 * must produce for one limb -- SURVEY appendix A4 (SEAL/Phantom ordering):
 * Cooley-Tukey, natural in, bit-reversed out, twiddles rp[m+i], result in [0,q).
 * Inputs are first reduced mod q (the engine defines the transform of any u64
 * word as the transform of its residue; the fault-injection harness at
 * ntt_test.cu:104-135 feeds words with arbitrary flipped bits). */
void orc_nwt_forward(u64 *a, int logN, u64 q, const u64 *rp)
{
    u64 N = (u64)1 << logN, t = N;
    for (u64 i = 0; i < N; i++) a[i] %= q;
    for (u64 m = 1; m < N; m <<= 1) {
        t >>= 1;
        for (u64 i = 0; i < m; i++) {
            u64 S = rp[m + i], j1 = 2 * i * t;
            for (u64 j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = orc_mulmod(a[j + t], S, q);
                a[j] = orc_addmod(U, V, q);
                a[j + t] = orc_submod(U, V, q);
            }
        }
    }
}

/* Inverse of orc_nwt_forward: Gentleman-Sande, bit-reversed in, natural out,
 * times N^-1 (semantics of
 * rfhe_framewk/src/negaclic_ntt.py:102-109 composed with the bit reversal). */
void orc_nwt_inverse(u64 *a, int logN, u64 q, const u64 *rp)
{
    u64 N = (u64)1 << logN, t = 1;
    for (u64 i = 0; i < N; i++) a[i] %= q;
    for (u64 m = N >> 1; m >= 1; m >>= 1) {
        for (u64 i = 0; i < m; i++) {
            u64 Sinv = orc_invmod(rp[m + i], q), j1 = 2 * i * t;
            for (u64 j = j1; j < j1 + t; j++) {
                u64 U = a[j], V = a[j + t];
                a[j] = orc_addmod(U, V, q);
                a[j + t] = orc_mulmod(orc_submod(U, V, q), Sinv, q);
            }
        }
        t <<= 1;
    }
    u64 ninv = orc_invmod(N % q, q);
    for (u64 i = 0; i < N; i++) a[i] = orc_mulmod(a[i], ninv, q);
}

/* ------------------------------------------------------------------ */
/* a4/a5: coefficient-wise products and negacyclic polymul             */
/* ------------------------------------------------------------------ */

/* rfhe_framewk/src/negaclic_ntt.py:126: C_hat[i] = A_hat[i]*B_hat[i] % mod */
void orc_modmul(u64 *c, const u64 *a, const u64 *b, u64 n, u64 mod)
{
    for (u64 i = 0; i < n; i++) c[i] = orc_mulmod(a[i] % mod, b[i] % mod, mod);
}

/* accumulate form used by keyswitch / BSGS inner products: c = (c + a*b) mod q */
void orc_modmul_acc(u64 *c, const u64 *a, const u64 *b, u64 n, u64 mod)
{
    for (u64 i = 0; i < n; i++)
        c[i] = orc_addmod(c[i] % mod, orc_mulmod(a[i] % mod, b[i] % mod, mod), mod);
}

/* rfhe_framewk/src/negaclic_ntt.py:112-120: schoolbook product mod x^n + 1 */
void orc_polymul_naive_negacyclic(u64 *res, const u64 *a, const u64 *b, u64 n, u64 mod)
{
    memset(res, 0, sizeof(u64) * n);
    for (u64 i = 0; i < n; i++)
        for (u64 j = 0; j < n; j++) {
            u64 k = (i + j) % n;
            u64 p = orc_mulmod(a[i] % mod, b[j] % mod, mod);
            if (i + j >= n) p = orc_mulmod(p, mod - 1, mod);
            res[k] = orc_addmod(res[k], p, mod);
        }
}

/* rfhe_framewk/src/negaclic_ntt.py:123-127 */
void orc_polymul_negacyclic_ntt(u64 *res, const u64 *a, const u64 *b, u64 n, u64 psi, u64 mod)
{
    u64 *A = (u64 *)malloc(sizeof(u64) * n), *B = (u64 *)malloc(sizeof(u64) * n);
    memcpy(A, a, sizeof(u64) * n);
    memcpy(B, b, sizeof(u64) * n);
    orc_negacyclic_ntt_natural(A, n, psi, mod);
    orc_negacyclic_ntt_natural(B, n, psi, mod);
    orc_modmul(res, A, B, n, mod);
    orc_negacyclic_intt_natural(res, n, psi, mod);
    free(A);
    free(B);
}

/* ------------------------------------------------------------------ */
/* a6: four-step NTT                                                   */
/* ------------------------------------------------------------------ */

/* reliability_test/four_step_ntt_prot.py:49-58 (ntt_direct): y_k = sum a_t w^(kt) */
void orc_ntt_direct(u64 *y, const u64 *a, u64 N, u64 mod, u64 g)
{
    u64 w = orc_powmod(g, (mod - 1) / N, mod);
    for (u64 k = 0; k < N; k++) {
        u64 acc = 0, wk = orc_powmod(w, k, mod), p = 1 % mod;
        for (u64 t = 0; t < N; t++) {
            acc = orc_addmod(acc, orc_mulmod(a[t] % mod, p, mod), mod);
            p = orc_mulmod(p, wk, mod);
        }
        y[k] = acc;
    }
}

/* reliability_test/four_step_ntt_prot.py:71-109, generalised from n1 == n2 to any
 * n1*n2 == N (the reference asserts a perfect square at :73-74):
 *   A[t2][t1] = a[t1 + n1*t2]                            (:81)
 *   B[t1][k2] = sum_t2 A[t2][t1] (w^n1)^(k2 t2)          (:84-90)
 *   C[t1][k2] = B[t1][k2] w^(k2 t1)                      (:93)
 *   Y[k1][k2] = sum_t1 C[t1][k2] (w^n2)^(k1 t1)          (:96-102)
 *   y[k1*n2 + k2] = Y[k1][k2]                            (:105-108)
 * Dense sub-DFTs exactly as the reference writes them. */
void orc_four_step_ntt(u64 *y, const u64 *a, u64 n1, u64 n2, u64 mod, u64 g)
{
    u64 N = n1 * n2;
    u64 w = orc_powmod(g, (mod - 1) / N, mod);
    u64 w_n1 = orc_powmod(w, n1, mod), w_n2 = orc_powmod(w, n2, mod);
    u64 *B = (u64 *)malloc(sizeof(u64) * N); /* [t1][k2] */
    u64 *C = (u64 *)malloc(sizeof(u64) * N);
    for (u64 t1 = 0; t1 < n1; t1++)
        for (u64 k2 = 0; k2 < n2; k2++) {
            u64 s = 0, step = orc_powmod(w_n1, k2, mod), p = 1 % mod;
            for (u64 t2 = 0; t2 < n2; t2++) {
                s = orc_addmod(s, orc_mulmod(a[t1 + n1 * t2] % mod, p, mod), mod);
                p = orc_mulmod(p, step, mod);
            }
            B[t1 * n2 + k2] = s;
        }
    for (u64 t1 = 0; t1 < n1; t1++)
        for (u64 k2 = 0; k2 < n2; k2++)
            C[t1 * n2 + k2] = orc_mulmod(B[t1 * n2 + k2], orc_powmod(w, k2 * t1, mod), mod);
    for (u64 k2 = 0; k2 < n2; k2++)
        for (u64 k1 = 0; k1 < n1; k1++) {
            u64 s = 0, step = orc_powmod(w_n2, k1, mod), p = 1 % mod;
            for (u64 t1 = 0; t1 < n1; t1++) {
                s = orc_addmod(s, orc_mulmod(C[t1 * n2 + k2], p, mod), mod);
                p = orc_mulmod(p, step, mod);
            }
            y[k1 * n2 + k2] = s;
        }
    free(B);
    free(C);
}

/* ------------------------------------------------------------------ */
/* a7: Barrett reduction                                               */
/* ------------------------------------------------------------------ */

/* rfhe_framewk/src/barrett_final.cpp:68-79 (make_barrett_ctx): K = bitlen(q-1),
 * mu = floor(2^(2K)/q).  Valid for q < 2^63 here (2K <= 126). */
void orc_barrett_ctx(u64 q, int *K, u64 *mu_lo, u64 *mu_hi)
{
    int k = 64 - __builtin_clzll(q - 1);
    u128 mu = ((u128)1 << (2 * k)) / q;
    *K = k;
    *mu_lo = (u64)mu;
    *mu_hi = (u64)(mu >> 64);
}

/* rfhe_framewk/src/barrett_final.cpp:120-141 (barrett_reduce_signatures) lifted
 * from a 64-bit t to the full 2K-bit product the NTT needs: s = floor(t*mu / 2^(2K)),
 * c = t - s*q, at most two conditional subtractions (one in the in-tree 37-bit
 * model; the 256-bit t*mu is formed exactly here so one always suffices when
 * t < q^2, a second is kept for t up to 2^(2K)). */
u64 orc_barrett_reduce(u64 t_lo, u64 t_hi, u64 q, int K, u64 mu_lo, u64 mu_hi)
{
    /* 128x128 -> 256 bit product, keep bits [2K, 2K+128) */
    u64 t[2] = {t_lo, t_hi}, mu[2] = {mu_lo, mu_hi}, p[4] = {0, 0, 0, 0};
    for (int i = 0; i < 2; i++) {
        u64 carry = 0;
        for (int j = 0; j < 2; j++) {
            u128 cur = (u128)t[i] * mu[j] + p[i + j] + carry;
            p[i + j] = (u64)cur;
            carry = (u64)(cur >> 64);
        }
        p[i + 2] += carry;
    }
    int sh = 2 * K; /* 2..126 */
    u64 s[2];
    for (int w = 0; w < 2; w++) {
        int bit = sh + 64 * w, idx = bit / 64, off = bit % 64;
        u64 v = idx < 4 ? p[idx] >> off : 0;
        if (off && idx + 1 < 4) v |= p[idx + 1] << (64 - off);
        s[w] = v;
    }
    u128 S = ((u128)s[1] << 64) | s[0];
    u128 T = ((u128)t_hi << 64) | t_lo;
    u128 c = T - S * q;
    if (c >= q) c -= q;
    if (c >= q) c -= q;
    return (u64)c;
}

/* ------------------------------------------------------------------ */
/* a8: base conversion / CRT                                           */
/* ------------------------------------------------------------------ */

/* Garner mixed-radix digits shared by the exact conversion and crt_kernel.
 * rfhe_framewk/src/baseConv.cu:98-113: c_0 = r_0; c_j = ((r_j - sum_{k<j} c_k pref_k) inv_j) mod p_j,
 * pref_0 = 1, pref_j = prod_{i<j} p_i (taken mod 2^128 exactly as the kernel's
 * unsigned __int128 table, :157-161), inv_j = (pref_j mod p_j)^-1 (:162-169). */
static void garner_digits(u64 *c, const u64 *res, u64 stride, u64 i, const u64 *p, int m)
{
    u128 pref[64];
    pref[0] = 1;
    for (int j = 1; j < m; j++) pref[j] = pref[j - 1] * p[j - 1];
    c[0] = res[i];
    for (int j = 1; j < m; j++) {
        u64 t = res[(u64)j * stride + i] % p[j];
        for (int k = 0; k < j; k++) {
            /* (c_k * pref_k) % p_j with the product wrapped to 128 bits, as the
             * kernel computes it (baseConv.cu:104-105) */
            u128 prod = (u128)c[k] * pref[k];
            u64 r = (u64)(prod % p[j]);
            t = (u64)(((u128)t + p[j] - r) % p[j]);
        }
        u64 inv = orc_invmod((u64)(pref[j] % p[j]), p[j]);
        c[j] = orc_mulmod(t, inv, p[j]);
    }
}

/* rfhe_framewk/src/baseConv.cu:85-120 (crt_kernel): x = sum c_k pref_k as a
 * 128-bit integer (lo, hi), wrapping mod 2^128 like the kernel (:115-119).
 * residues: m x N row-major u64. */
void orc_crt_garner(u64 *x_lo, u64 *x_hi, const u64 *residues, const u64 *moduli, int m, u64 N)
{
    u128 pref[64];
    u64 c[64];
    pref[0] = 1;
    for (int j = 1; j < m; j++) pref[j] = pref[j - 1] * moduli[j - 1];
    for (u64 i = 0; i < N; i++) {
        garner_digits(c, residues, N, i, moduli, m);
        u128 x = 0;
        for (int k = 0; k < m; k++) x += (u128)c[k] * pref[k];
        x_lo[i] = (u64)x;
        x_hi[i] = (u64)(x >> 64);
    }
}

/* motivation/baseConv.py:67-83 (base_conv_fixed): x = sum r_j Phat_j inv_j mod P,
 * out[k][i] = x mod q_k.  The Python uses bignums; x in [0,P) is unique, so it is
 * evaluated here from its mixed-radix digits with every prefix product reduced
 * mod q_k (exact for any number of limbs; requires residues < p_j, pairwise
 * coprime p_j).  out: k x N row-major. */
void orc_baseconv_exact(u64 *out, const u64 *residues, const u64 *mod_in, int m,
                        const u64 *mod_out, int k, u64 N)
{
    /* constants of the digit recurrence, once per call: pin[l][j] = p_0..p_{l-1} mod p_j (l <= j),
     * inv[j] = (p_0..p_{j-1})^-1 mod p_j, pout[l][o] = p_0..p_{l-1} mod q_o */
    u64 *pin = (u64 *)malloc(sizeof(u64) * (size_t)m * (size_t)m);
    u64 *pout = (u64 *)malloc(sizeof(u64) * (size_t)m * (size_t)k);
    u64 inv[64];
    for (int j = 0; j < m; j++) {
        u64 pj = mod_in[j], pref = 1 % pj;
        for (int l = 0; l < j; l++) {
            pin[(size_t)l * m + j] = pref;
            pref = orc_mulmod(pref, mod_in[l] % pj, pj);
        }
        inv[j] = j ? orc_invmod(pref, pj) : 1 % pj;
    }
    for (int o = 0; o < k; o++) {
        u64 q = mod_out[o], pref = 1 % q;
        for (int l = 0; l < m; l++) {
            pout[(size_t)l * k + o] = pref;
            pref = orc_mulmod(pref, mod_in[l] % q, q);
        }
    }
#pragma omp parallel for schedule(static)
    for (long long ii = 0; ii < (long long)N; ii++) {
        const u64 i = (u64)ii;
        u64 c[64];
        /* digits with exact (not wrapped) prefix residues */
        c[0] = residues[i] % mod_in[0];
        for (int j = 1; j < m; j++) {
            u64 pj = mod_in[j], t = residues[(u64)j * N + i] % pj;
            u64 acc = 0;
            for (int l = 0; l < j; l++) acc = orc_addmod(acc, orc_mulmod(c[l] % pj, pin[(size_t)l * m + j], pj), pj);
            c[j] = orc_mulmod(orc_submod(t, acc, pj), inv[j], pj);
        }
        for (int o = 0; o < k; o++) {
            u64 q = mod_out[o], acc = 0;
            for (int l = 0; l < m; l++) acc = orc_addmod(acc, orc_mulmod(c[l] % q, pout[(size_t)l * k + o], q), q);
            out[(u64)o * N + i] = acc;
        }
    }
    free(pin);
    free(pout);
}

/* rfhe_framewk/src/baseConv.py:10-40 (bConv): out[i][k] = sum_j ((r_j * Phat_j * inv_j) mod q_k),
 * the sum NOT reduced (:31-37); the product r_j*Phat_j*inv_j is the raw integer
 * (:27), so its residue is (r_j mod q)(Phat_j mod q)(inv_j mod q) mod q.
 * inv_j = (Phat_j)^-1 mod p_j (:18).  out: N x k row-major (element-major, as the
 * reference returns it). */
void orc_bconv_fast(u64 *out, const u64 *residues, const u64 *mod_in, int m,
                    const u64 *mod_out, int k, u64 N)
{
    u64 *coef = (u64 *)malloc(sizeof(u64) * (size_t)m * (size_t)k); /* (Phat_j inv_j) mod q_k */
    for (int j = 0; j < m; j++) {
        u64 pj = mod_in[j], hat_mod_pj = 1 % pj;
        for (int l = 0; l < m; l++)
            if (l != j) hat_mod_pj = orc_mulmod(hat_mod_pj, mod_in[l] % pj, pj);
        u64 inv = orc_invmod(hat_mod_pj, pj);
        for (int o = 0; o < k; o++) {
            u64 q = mod_out[o], hat_mod_q = 1 % q;
            for (int l = 0; l < m; l++)
                if (l != j) hat_mod_q = orc_mulmod(hat_mod_q, mod_in[l] % q, q);
            coef[j * k + o] = orc_mulmod(hat_mod_q, inv % q, q);
        }
    }
    for (u64 i = 0; i < N; i++)
        for (int o = 0; o < k; o++) {
            u64 q = mod_out[o], total = 0;
            for (int j = 0; j < m; j++)
                total += orc_mulmod(residues[(u64)j * N + i] % q, coef[j * k + o], q);
            out[i * (u64)k + o] = total;
        }
    free(coef);
}

/* ------------------------------------------------------------------ */
/* a9: BSGS block-diagonal Hadamard mat-vec                            */
/* ------------------------------------------------------------------ */

/* motivation/bsgs.py:39-52: y_i = sum_j M[(j - i) mod k] (.) v_j over k blocks of
 * `bs`; NumPy int64 arithmetic, no modular reduction (:50) -- wraps mod 2^64. */
void orc_bsgs_hadamard(int64_t *y, const int64_t *M_blocks, const int64_t *v, int k, int bs)
{
    for (int i = 0; i < k; i++)
        for (int e = 0; e < bs; e++) {
            u64 acc = 0;
            for (int j = 0; j < k; j++) {
                int b = ((j - i) % k + k) % k;
                acc += (u64)M_blocks[b * bs + e] * (u64)v[j * bs + e];
            }
            y[i * bs + e] = (int64_t)acc;
        }
}

/* modular variant used by the keyswitch-shaped accumulate (same index pattern,
 * every product and the sum reduced mod q) */
void orc_bsgs_hadamard_mod(u64 *y, const u64 *M_blocks, const u64 *v, int k, int bs, u64 q)
{
    for (int i = 0; i < k; i++)
        for (int e = 0; e < bs; e++) {
            u64 acc = 0;
            for (int j = 0; j < k; j++) {
                int b = ((j - i) % k + k) % k;
                acc = orc_addmod(acc, orc_mulmod(M_blocks[b * bs + e] % q, v[j * bs + e] % q, q), q);
            }
            y[i * bs + e] = acc;
        }
}

/* ------------------------------------------------------------------ */
/* batch helper for the CPU baseline                                   */
/* ------------------------------------------------------------------ */

/* `limbs` forward transforms, limb-major L x N, limb l with rp + l*N
 * (the loop nwt_2d_radix8_forward_inplace replaces, ntt_test.cu:95).  Limbs are independent:
 * with an OpenMP build they are spread over the host cores (OMP_NUM_THREADS / orc_set_threads). */
void orc_nwt_forward_batch(u64 *a, int logN, int limbs, const u64 *q, const u64 *rp)
{
    u64 N = (u64)1 << logN;
#pragma omp parallel for schedule(dynamic, 1)
    for (int l = 0; l < limbs; l++) orc_nwt_forward(a + (u64)l * N, logN, q[l], rp + (u64)l * N);
}

void orc_nwt_inverse_batch(u64 *a, int logN, int limbs, const u64 *q, const u64 *rp)
{
    u64 N = (u64)1 << logN;
#pragma omp parallel for schedule(dynamic, 1)
    for (int l = 0; l < limbs; l++) orc_nwt_inverse(a + (u64)l * N, logN, q[l], rp + (u64)l * N);
}

/* c[l] = (c[l] +) a[l] * b[l] mod q[l] over `limbs` limbs of n words (rfhe_framewk/src/negaclic_ntt.py:126 per limb) */
void orc_modmul_batch(u64 *c, const u64 *a, const u64 *b, u64 n, int limbs, const u64 *q, int accumulate)
{
#pragma omp parallel for schedule(static)
    for (int l = 0; l < limbs; l++) {
        if (accumulate) orc_modmul_acc(c + (u64)l * n, a + (u64)l * n, b + (u64)l * n, n, q[l]);
        else orc_modmul(c + (u64)l * n, a + (u64)l * n, b + (u64)l * n, n, q[l]);
    }
}

/* worker threads of the batch helpers above (1 = the scalar port); returns the count in effect */
#ifdef _OPENMP
#include <omp.h>
int orc_set_threads(int n)
{
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}
#else
int orc_set_threads(int n)
{
    (void)n;
    return 1;
}
#endif
