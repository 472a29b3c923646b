#include "host_math.hpp"

#include <algorithm>
#include <map>

namespace fhe {
namespace host {

u64 pow_mod(u64 b, u64 e, u64 m)
{
    if (m == 1) return 0;
    u64 r = 1;
    b %= m;
    for (; e; e >>= 1) {
        if (e & 1) r = mul_mod(r, b, m);
        b = mul_mod(b, b, m);
    }
    return r;
}

u64 inv_mod(u64 a, u64 m)
{
    // extended Euclid on signed 128-bit cofactors
    __int128 r0 = m, r1 = a % m, t0 = 0, t1 = 1;
    while (r1) {
        __int128 q = r0 / r1;
        __int128 r2 = r0 - q * r1, t2 = t0 - q * t1;
        r0 = r1; r1 = r2;
        t0 = t1; t1 = t2;
    }
    if (r0 != 1) return 0;
    return (u64)(t0 < 0 ? t0 + m : t0);
}

bool is_prime(u64 n)
{
    if (n < 2) return false;
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : bases) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int s = __builtin_ctzll(d);
    d >>= s;
    // the first twelve primes are a deterministic witness set below 3.3e24
    for (u64 a : bases) {
        u64 x = pow_mod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s && witness; r++) {
            x = mul_mod(x, x, n);
            if (x == n - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

unsigned bit_reverse(unsigned x, int bits)
{
    unsigned r = 0;
    for (int i = 0; i < bits; i++, x >>= 1) r = (r << 1) | (x & 1u);
    return r;
}

bool primes_for(u64 N, int bits, int count, std::vector<u64> &out)
{
    out.clear();
    if (bits < 2 || bits > 61 || N == 0) return false;
    const u64 step = 2 * N;
    const u64 floor_ = (u64)1 << (bits - 1);
    u64 cand = ((((u64)1 << bits) - 1) / step) * step + 1;
    while ((int)out.size() < count && cand > floor_) {
        if (is_prime(cand)) out.push_back(cand);
        if (cand < step) break;
        cand -= step;
    }
    if ((int)out.size() != count) return false;
    std::reverse(out.begin(), out.end());
    return true;
}

bool create_moduli(u64 N, const int *bits, int count, u64 *out)
{
    std::map<int, int> need;
    for (int i = 0; i < count; i++) need[bits[i]]++;
    std::map<int, std::vector<u64>> pool;
    for (auto &kv : need)
        if (!primes_for(N, kv.first, kv.second, pool[kv.first])) return false;
    std::map<int, int> used;
    for (int i = 0; i < count; i++) out[i] = pool[bits[i]][used[bits[i]]++];
    return true;
}

void const_ratio(u64 q, u64 out[3])
{
    // 2^128 / q by long division of (2^128 - 1) and a fix-up
    u128 all = ~(u128)0;
    u128 quo = all / q;
    u64 rem = (u64)(all % q) + 1;
    if (rem == q) { quo += 1; rem = 0; }
    out[0] = (u64)quo;
    out[1] = (u64)(quo >> 64);
    out[2] = rem;
}

bool min_primitive_root(u64 q, u64 order, u64 &root)
{
    if (order < 2 || (order & (order - 1)) || (q - 1) % order) return false;
    // any element whose (order/2)-th power is -1 after raising to (q-1)/order.  For a prime q every second
    // candidate works, so the search is bounded: a modulus that yields nothing in 4096 tries is not a prime
    // with a subgroup of that order (the unbounded loop would run for 2^60 steps on a composite modulus).
    u64 g = 0;
    const u64 last = q - 1 < 4098 ? q - 1 : 4098;
    for (u64 c = 2; c <= last && !g; c++) {
        u64 r = pow_mod(c, (q - 1) / order, q);
        if (pow_mod(r, order / 2, q) == q - 1) g = r;
    }
    if (!g) return false;
    // all primitive roots of that order are the odd powers of g
    const u64 g2 = mul_mod(g, g, q);
    u64 cur = g, best = g;
    for (u64 k = 1; k < order / 2; k++) {
        cur = mul_mod(cur, g2, q);
        best = std::min(best, cur);
    }
    root = best;
    return true;
}

void root_powers(u64 q, int logn, u64 psi, u64 *rp)
{
    const u64 N = (u64)1 << logn;
    u64 p = 1 % q;
    for (u64 i = 0; i < N; i++) {
        rp[bit_reverse((unsigned)i, logn)] = p;
        p = mul_mod(p, psi, q);
    }
}

void cyclic_table(u64 mod, int logn, u64 root, bool nth_root_convention, u64 *tw)
{
    const u64 n = (u64)1 << logn;
    tw[0] = 1 % mod;
    for (int s = 0; s < logn; s++) {
        const u64 m = (u64)1 << s, len = 2 * m;
        const u64 wlen = nth_root_convention ? pow_mod(root, n / len, mod) : pow_mod(root, (mod - 1) / len, mod);
        // running power in natural order, scattered to bit-reversed block index
        u64 p = 1 % mod;
        for (u64 k = 0; k < m; k++) {
            tw[m + bit_reverse((unsigned)k, s)] = p;
            p = mul_mod(p, wlen, mod);
        }
    }
}

} // namespace host
} // namespace fhe
