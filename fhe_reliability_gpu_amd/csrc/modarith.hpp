// modarith.hpp -- modular arithmetic policies shared by every kernel.
//
// Two arithmetic paths, chosen per RNS limb when the tables are built:
//
//  * ArithF64  (q < 2^50, the size the reference uses: reliability_test/ntt_test.cu:44
//    creates 50-bit primes).  Residues live in FP64 registers as exact integers in
//    a signed lazy range; a modular multiply is 6 FP64 ops (2 mul, 1 rndne, 2 fma,
//    1 add) on the half-rate FP64 pipe of CDNA4 instead of ~10 quarter-rate 32-bit
//    integer multiplies.  Every intermediate is an exactly representable integer
//    (|x| < 2^53), so the result is bit-exact, not approximate.
//  * ArithU64  (q < 2^61): Harvey lazy butterflies with Shoup twiddle quotients,
//    values in [0,4q).
//
// The file compiles under hipcc (device) and under g++ (tests/emu: the same
// template code run thread-by-thread on the CPU to check indexing and bounds).
// Contraction must be off (-ffp-contract=off): the algorithm needs the individually
// rounded product h = a*w next to fma(a,w,-h).
#pragma once
#include <cstddef>
#include <cstdint>

// Bit-exactness of ArithF64 depends on a*w and fma(a,w,-h) being rounded separately: never let the
// compiler contract them, whatever flags the build passes (the Makefile also sets -ffp-contract=off).
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FHE_HD __host__ __device__ __forceinline__
#define FHE_D __device__ __forceinline__
#else
#define FHE_HD inline
#define FHE_D inline
#endif
// Twiddle tables are reached through pointers stored in device memory; telling the
// compiler they are global (address space 1) turns flat_load into global_load /
// s_load.  Host passes and the CPU emulation see a plain pointer.
#if defined(__HIP_DEVICE_COMPILE__)
#define FHE_GLOBAL __attribute__((address_space(1)))
#define FHE_CONSTANT __attribute__((address_space(4)))   // read-only for the whole launch
#else
#define FHE_GLOBAL
#define FHE_CONSTANT
#endif
#if defined(__clang__)
#define FHE_ASSUME(x) __builtin_assume(x)
#else
#define FHE_ASSUME(x) ((void)0)
#endif

namespace fhe {

typedef uint64_t u64;
typedef uint32_t u32;

// One twiddle = 16 bytes so a single dwordx4 load fetches the factor and its
// precomputed quotient.  ArithU64: {w, floor(w*2^64/q)}; ArithF64: {bits(w), bits(w/q)}.
struct alignas(16) Tw {
    u64 a, b;
};

// Per-limb constants, resident in device memory next to the tables.
struct alignas(16) LimbParams {
    u64 q;           // modulus
    u64 two_q;       // 2q (ArithU64)
    double n;        // (double)q    (ArithF64)
    double ninv;     // 1.0 / n, correctly rounded
    Tw inv_n;        // N^-1 mod q encoded as a twiddle of this limb's path
    const Tw *fwd;   // forward table, N entries, entry k = psi^bitrev(k)  (entry 0 unused)
    const Tw *inv;   // inverse table, entry k = fwd[k]^-1
    u64 barrett_lo;  // floor(2^128/q) low / high words (Modulus::const_ratio, ntt_test.cu:49-53)
    u64 barrett_hi;
    int path;        // 0 = ArithF64, 1 = ArithU64
    int pad_;
    u64 pad2_;
};

enum { PATH_F64 = 0, PATH_U64 = 1 };

typedef const Tw FHE_GLOBAL *TwPtr;
FHE_HD TwPtr as_global(const Tw *p) { return (TwPtr)p; }

// Streaming accesses (touched once per launch): non-temporal hint so they do not displace
// the fused kernel's hand-off lines and twiddles from the XCD's L2.
FHE_D u64 load_stream_u64(const u64 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
FHE_D void store_stream_u64(u64 *p, u64 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// 64-bit load that must observe data another workgroup of the same XCD has stored
// and drained to L2 (fused NTT hand-off): agent-scope relaxed atomic load =
// global_load_dwordx2 sc1, served by L2, never by this CU's L1.
FHE_D u64 load_coherent_u64(const u64 *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    return *p;
#endif
}

FHE_HD u64 mulhi64(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

FHE_HD double u64_bits_to_double(u64 b) { return __builtin_bit_cast(double, b); }
FHE_HD u64 double_to_u64_bits(double d) { return __builtin_bit_cast(u64, d); }

// Full reduction of an arbitrary 64-bit word (only reached for out-of-range inputs,
// e.g. the bit-flipped symbols of reliability_test/ntt_test.cu:104-135).
#if defined(__HIPCC__)
__host__ __device__ __attribute__((noinline)) inline u64 reduce_any_u64(u64 x, u64 q) { return x % q; }
#else
inline u64 reduce_any_u64(u64 x, u64 q) { return x % q; }
#endif

// (hi:lo) mod q with ratio = floor(2^128/q) = (r1:r0).  Exact quotient estimate:
// floor(x*ratio/2^128) is floor(x/q) or one less, so one subtraction finishes; a
// second is kept for robustness.  This is the 128->64 Barrett step Phantom's
// DModulus carries its const_ratio for (reliability_test/ntt_test.cu:49-53).
FHE_HD u64 barrett128(u64 lo, u64 hi, u64 q, u64 r0, u64 r1)
{
    const u64 c = mulhi64(lo, r0);
    const u64 t1l = lo * r1, t1h = mulhi64(lo, r1);
    const u64 t2l = hi * r0, t2h = mulhi64(hi, r0);
    u64 s = t1l + t2l;
    u64 carry = s < t1l;
    const u64 s2 = s + c;
    carry += s2 < s;
    const u64 qhat = hi * r1 + t1h + t2h + carry;
    u64 r = lo - qhat * q;
    r = r >= q ? r - q : r;
    r = r >= q ? r - q : r;
    return r;
}

// ---------------------------------------------------------------------------
// ArithF64
// ---------------------------------------------------------------------------
struct ArithF64 {
    typedef double elem;
    static constexpr int PATH = PATH_F64;
    // Growth bound bookkeeping (units of q), see DESIGN.md "lazy FP64 ranges":
    // forward butterfly  B' = B + 0.5 + B*q*2^-52 <= 1.25 B + 0.5  -> from canonical input 5
    // stages, after a reduce() 6 stages stay below 8q < 2^53.
    // inverse butterfly  B' = 2B                                    -> 2 resp. 3 stages.
    static constexpr int FWD_FIRST = 5, FWD_NEXT = 6, INV_FIRST = 2, INV_NEXT = 3;

    struct Ctx {
        double n, ninv;
        u64 q;
    };
    static FHE_HD Ctx make_ctx(const LimbParams &p) { return Ctx{p.n, p.ninv, p.q}; }

    static FHE_HD bool in_range(u64 raw, const Ctx &c) { return raw < c.q; }
    // raw < q < 2^50: OR the exponent of 2^52 into the high word, subtract 2^52.
    static FHE_HD elem from_canonical(u64 raw) { return u64_bits_to_double(raw | 0x4330000000000000ull) - 4503599627370496.0; }
    // x integer in [0, 2^52)
    static FHE_HD u64 to_u64(elem x) { return double_to_u64_bits(x + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull; }
    static FHE_HD elem load_lazy(u64 raw) { return u64_bits_to_double(raw); }   // between passes: raw double bits
    static FHE_HD u64 store_lazy(elem x) { return double_to_u64_bits(x); }

    // x -> x - q*rint(x/q): |result| <= q/2 (+1 ulp of the quotient, harmless)
    static FHE_HD void reduce(elem &x, const Ctx &c)
    {
        double k = __builtin_rint(x * c.ninv);
        x = __builtin_fma(-k, c.n, x);
    }
    static FHE_HD u64 canonical(elem x, const Ctx &c)
    {
        reduce(x, c);
        if (x < 0.0) x += c.n;
        return to_u64(x);
    }
    // a*w mod q in (-q(0.5+|a|2^-52), +...): exact integer arithmetic in FP64
    static FHE_HD elem mulmod(elem a, const Tw &t, const Ctx &c)
    {
        double w = u64_bits_to_double(t.a), wp = u64_bits_to_double(t.b);
        double h = a * w;
        double k = __builtin_rint(a * wp);
        double l = __builtin_fma(a, w, -h);
        double r = __builtin_fma(-k, c.n, h);
        return r + l;
    }
    // the same product with the factor given as a number: wp ~ w/q to within 2 ulp (e.g. w * ninv), |result| < 0.9 q
    static FHE_HD elem mulmod_w(elem a, double w, double wp, const Ctx &c)
    {
        double h = a * w;
        double k = __builtin_rint(a * wp);
        double l = __builtin_fma(a, w, -h);
        double r = __builtin_fma(-k, c.n, h);
        return r + l;
    }
    // Cooley-Tukey: (X, Y) -> (X + wY, X - wY)
    static FHE_HD void bfly_fwd(elem &X, elem &Y, const Tw &t, const Ctx &c)
    {
        double v = mulmod(Y, t, c);
        double x = X;
        X = x + v;
        Y = x - v;
    }
    // Gentleman-Sande: (X, Y) -> (X + Y, (X - Y) w)
    static FHE_HD void bfly_inv(elem &X, elem &Y, const Tw &t, const Ctx &c)
    {
        double s = X + Y, d = X - Y;
        X = s;
        Y = mulmod(d, t, c);
    }
    // last inverse stage with N^-1 folded in: (X, Y) -> ((X + Y) n, (X - Y) wn), n = N^-1, wn = w N^-1
    static FHE_HD void bfly_inv_scaled(elem &X, elem &Y, const Tw &n, const Tw &wn, const Ctx &c)
    {
        double s = X + Y, d = X - Y;
        X = mulmod(s, n, c);
        Y = mulmod(d, wn, c);
    }
    static FHE_HD elem add(elem a, elem b) { return a + b; }
    // product of two canonical residues, canonical result.  The quotient estimate rint(x * (y/q)) is off by
    // less than 0.5 + 0.375 (three roundings on a value below 2^50), so h - k*q + l is an exact integer
    // in (-0.875 q, 0.875 q).
    static FHE_HD u64 mulvar(u64 x, u64 y, const LimbParams &p)
    {
        const Ctx c = make_ctx(p);
        const double a = from_canonical(x), b = from_canonical(y);
        const double h = a * b;
        const double k = __builtin_rint(a * (b * c.ninv));
        const double l = __builtin_fma(a, b, -h);
        return canonical(__builtin_fma(-k, c.n, h) + l, c);
    }
    // product of two values in this arithmetic's lazy range, result again lazy (|r| < 0.6 q): the fused negacyclic
    // product multiplies the two forward transforms where they sit in LDS, without a trip through canonical words.
    // Both factors are folded to |.| <= q/2 first, so h - k*q + l stays an exact integer well below 2^53.
    static FHE_HD elem mulvar_lazy(elem a, elem b, const Ctx &c)
    {
        reduce(a, c);
        reduce(b, c);
        const double h = a * b;
        const double k = __builtin_rint(a * (b * c.ninv));
        const double l = __builtin_fma(a, b, -h);
        return __builtin_fma(-k, c.n, h) + l;
    }
    // running sum of mulmod() / mulmod_w() results (each below 0.9 q in magnitude): fold back every fourth term
    static FHE_HD void lazy_acc(elem &s, elem v, int terms, const Ctx &c)
    {
        s += v;
        if ((terms & 3) == 0) reduce(s, c);
    }
    // host: encode a residue w (< q) as a twiddle
    static inline Tw encode(u64 w, u64 q)
    {
        double dw = (double)w, dq = (double)q;
        return Tw{double_to_u64_bits(dw), double_to_u64_bits(dw / dq)};
    }
};

// ---------------------------------------------------------------------------
// ArithU64
// ---------------------------------------------------------------------------
struct ArithU64 {
    typedef u64 elem;
    static constexpr int PATH = PATH_U64;
    static constexpr int FWD_FIRST = 1 << 20, FWD_NEXT = 1 << 20, INV_FIRST = 1 << 20, INV_NEXT = 1 << 20;

    struct Ctx {
        u64 q, two_q;
    };
    static FHE_HD Ctx make_ctx(const LimbParams &p) { return Ctx{p.q, p.two_q}; }

    static FHE_HD bool in_range(u64 raw, const Ctx &c) { return raw < c.q; }
    static FHE_HD elem from_canonical(u64 raw) { return raw; }
    static FHE_HD elem load_lazy(u64 raw) { return raw; }   // [0,4q) forward, [0,2q) inverse
    static FHE_HD u64 store_lazy(elem x) { return x; }
    static FHE_HD void reduce(elem &, const Ctx &) {}
    static FHE_HD u64 canonical(elem x, const Ctx &c)
    {
        x = x >= c.two_q ? x - c.two_q : x;
        return x >= c.q ? x - c.q : x;
    }
    // Shoup lazy product: result in [0, 2q) for ANY 64-bit a
    static FHE_HD elem mulmod(elem a, const Tw &t, const Ctx &c)
    {
        u64 qh = mulhi64(a, t.b);
        return a * t.a - qh * c.q;
    }
    // Harvey CT: inputs/outputs in [0, 4q)
    static FHE_HD void bfly_fwd(elem &X, elem &Y, const Tw &t, const Ctx &c)
    {
        u64 x = X >= c.two_q ? X - c.two_q : X;
        u64 v = mulmod(Y, t, c);
        X = x + v;
        Y = x - v + c.two_q;
    }
    // Harvey GS: inputs/outputs in [0, 2q)
    static FHE_HD void bfly_inv(elem &X, elem &Y, const Tw &t, const Ctx &c)
    {
        u64 s = X + Y;
        u64 d = X - Y + c.two_q;
        X = s >= c.two_q ? s - c.two_q : s;
        Y = mulmod(d, t, c);
    }
    // last inverse stage with N^-1 folded in (inputs in [0, 2q), outputs in [0, 2q))
    static FHE_HD void bfly_inv_scaled(elem &X, elem &Y, const Tw &n, const Tw &wn, const Ctx &c)
    {
        u64 s = X + Y;
        u64 d = X - Y + c.two_q;
        X = mulmod(s, n, c);
        Y = mulmod(d, wn, c);
    }
    static FHE_HD u64 mulvar(u64 x, u64 y, const LimbParams &p) { return barrett128(x * y, mulhi64(x, y), p.q, p.barrett_lo, p.barrett_hi); }
    // lazy x lazy -> canonical (which is also a valid lazy input of the inverse): factors below 4q < 2^63, so the
    // 128-bit product is below 2^126 and one Barrett step finishes
    static FHE_HD elem mulvar_lazy(elem a, elem b, const LimbParams &p) { return barrett128(a * b, mulhi64(a, b), p.q, p.barrett_lo, p.barrett_hi); }
    // running sum of mulmod() results (each in [0, 2q)), kept in [0, 2q)
    static FHE_HD void lazy_acc(elem &s, elem v, int, const Ctx &c)
    {
        s += v;
        s = s >= c.two_q ? s - c.two_q : s;
    }
    static inline Tw encode(u64 w, u64 q) { return Tw{w, (u64)(((unsigned __int128)w << 64) / q)}; }
};

// The inverse transform of ArithU64 keeps values in [0,2q): entering it needs one
// conditional subtraction from the canonical/[0,4q) range -- inputs are canonical
// (< q) or reduced through reduce_any_u64, so nothing to do.

} // namespace fhe
