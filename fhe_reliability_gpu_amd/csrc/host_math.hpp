// host_math.hpp -- host-side number theory for building moduli and twiddle tables
// (the role of phantom::arith::{Modulus, CoeffModulus::Create, NTT} at
// reliability_test/ntt_test.cu:44,57-69).  Product code: does not use oracle/.
#pragma once
#include <cstdint>
#include <vector>

namespace fhe {
namespace host {

typedef uint64_t u64;
typedef unsigned __int128 u128;

inline u64 mul_mod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }
u64 pow_mod(u64 b, u64 e, u64 m);
// inverse of a modulo m (any m, gcd(a,m) must be 1); returns 0 when not invertible
u64 inv_mod(u64 a, u64 m);
bool is_prime(u64 n);
unsigned bit_reverse(unsigned x, int bits);

// SEAL/Phantom CoeffModulus::Create for one bit size: the `count` largest primes
// below 2^bits congruent to 1 mod 2N, returned smallest first.
bool primes_for(u64 N, int bits, int count, std::vector<u64> &out);
// Full rule for a list of bit sizes (sizes may repeat; each occurrence of a size
// receives the next prime of that size, smallest first, as SEAL hands them out).
bool create_moduli(u64 N, const int *bits, int count, u64 *out);

// floor(2^128 / q) as {lo, hi} and 2^128 mod q  (Modulus::const_ratio)
void const_ratio(u64 q, u64 out[3]);

// numerically smallest primitive `order`-th root of unity mod prime q (order = 2^k)
bool min_primitive_root(u64 q, u64 order, u64 &root);

// Negacyclic tables: rp[bitrev(i)] = psi^i.
void root_powers(u64 q, int logn, u64 psi, u64 *rp);
// Cyclic tables in the engine's network order, generator convention of
// motivation/ntt.py:22 (wlen = root^((mod-1)/len)) or the n-th-root convention of
// rfhe_framewk/src/negaclic_ntt.py:46 (wlen = root^(n/len)):
//   tw[m + i] = wlen(2m)^bitrev_{log2 m}(i),  m = 1,2,..,n/2
void cyclic_table(u64 mod, int logn, u64 root, bool nth_root_convention, u64 *tw);

} // namespace host
} // namespace fhe
