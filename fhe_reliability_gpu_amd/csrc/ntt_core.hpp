// ntt_core.hpp -- the butterfly passes of the batched NTT, written once as
// templates over the arithmetic policy (modarith.hpp) and the compile-time plan.
//
// Transform definition (what nwt_2d_radix8_forward_inplace at
// reliability_test/ntt_test.cu:95 must produce): Cooley-Tukey,
// stage s = 0..logN-1 has m = 2^s blocks, pairs differ in index bit (logN-1-s),
// twiddle table entry (2^s + block).  Natural order in, bit-reversed out.  The
// inverse walks the same network backwards with Gentleman-Sande butterflies and the
// inverse table.  A cyclic transform (motivation/ntt.py:8-32) is the same network
// with a different table plus a final bit-reversal gather (see capi.cpp).
//
// Decomposition: logN = PC + PR.  The "column pass" performs the PC stages whose
// pair distance is >= 2^PR on a tile of TC adjacent columns (lanes run along the
// contiguous index, so every global and LDS access is coalesced); the "row pass"
// performs the PR stages inside contiguous rows of 2^PR points.  Inside a pass a
// thread keeps 2^K points in registers for K stages (K <= 4), and the points are
// exchanged through LDS between such steps.
//
// Everything here is callable per (thread id, step) so that tests/emu can run the
// exact same code on the CPU with a loop standing in for the workgroup.
#pragma once
#include "modarith.hpp"

namespace fhe {

FHE_HD constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Lazy-range schedule for ArithF64: bit u of the mask = "reduce all registers
// before stage u of this pass".  `stage0` is the number of stages already done
// since the data was canonical, FIRST/NEXT the stage budgets of the arithmetic.
FHE_HD constexpr u32 reduce_mask(int stage0, int nstages, int first, int next)
{
    u32 mask = 0;
    for (int u = 0; u < nstages; u++) {
        int s = stage0 + u;
        if (s >= first && (s - first) % next == 0) mask |= 1u << u;
    }
    return mask;
}

// K forward stages on R = 2^K registers.  Register r holds the point whose K-bit
// field (most significant bit = first stage) equals r.  `s` = global index of the
// first stage, `prefix` = value of the s index bits above the field.
template <class A, int K, u32 RED, int U0>
FHE_D void radix_fwd(typename A::elem (&x)[1 << K], const Tw *__restrict__ tw, u32 s, u32 prefix, const typename A::Ctx &c)
{
    constexpr int R = 1 << K;
#pragma unroll
    for (int u = 0; u < K; u++) {
        if ((RED >> (U0 + u)) & 1) {
#pragma unroll
            for (int r = 0; r < R; r++) A::reduce(x[r], c);
        }
        const int half = R >> (u + 1);
#pragma unroll
        for (int b = 0; b < (1 << u); b++) {
            const Tw w = tw[(1u << (s + u)) + (prefix << u) + b];
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_fwd(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// K inverse stages, undoing radix_fwd: forward stage s+K-1 first.
template <class A, int K, u32 RED, int U0>
FHE_D void radix_inv(typename A::elem (&x)[1 << K], const Tw *__restrict__ tw, u32 s, u32 prefix, const typename A::Ctx &c)
{
    constexpr int R = 1 << K;
#pragma unroll
    for (int v = 0; v < K; v++) {
        const int u = K - 1 - v; // forward stage index inside the field
        if ((RED >> (U0 + v)) & 1) {
#pragma unroll
            for (int r = 0; r < R; r++) A::reduce(x[r], c);
        }
        const int half = R >> (u + 1);
#pragma unroll
        for (int b = 0; b < (1 << u); b++) {
            const Tw w = tw[(1u << (s + u)) + (prefix << u) + b];
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_inv(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// Plan of one pass: P stages split into up to three register steps.
template <int K0_, int K1_, int K2_>
struct Steps {
    static constexpr int K0 = K0_, K1 = K1_, K2 = K2_;
    static constexpr int P = K0_ + K1_ + K2_;
    static constexpr int NSTEP = (K0_ > 0) + (K1_ > 0) + (K2_ > 0);
    static constexpr int KMAX = cmax(K0_, cmax(K1_, K2_));
    FHE_HD static constexpr int k(int e) { return e == 0 ? K0_ : e == 1 ? K1_ : K2_; }
    FHE_HD static constexpr int done(int e) { return e == 0 ? 0 : e == 1 ? K0_ : K0_ + K1_; } // stages before step e
};

// Row-pass LDS image: 16 bytes of padding after every 16 points keeps the
// 128-byte-per-lane ds_read_b128 / ds_write_b128 of the contiguous step conflict free.
FHE_HD constexpr u32 row_pad(u32 g) { return g + ((g >> 4) << 1); }

enum IoMode { IO_CANONICAL = 0, IO_LAZY = 1 };

// ---------------------------------------------------------------------------
// Column pass: stages [S0, S0+P) of a transform of 2^LOGN points whose pair
// distances are multiples of STRIDE = 2^(LOGN-S0-P); the tile is TC adjacent
// "columns" (consecutive values of the low index bits).  With S0 = 0 this is the
// first pass of the forward transform / last pass of the inverse.
// ---------------------------------------------------------------------------
template <class A, class ST, int LOGN, int S0, int TC, int NTHREADS, bool INVERSE, int IN_MODE, int OUT_MODE, u32 RED>
struct ColPass {
    typedef A Arith;
    typedef typename A::elem elem;
    static constexpr int P = ST::P;
    static constexpr int NPTS = 1 << P;
    static constexpr int LOGSTRIDE = LOGN - S0 - P;
    static constexpr u32 STRIDE = 1u << LOGSTRIDE;      // distance between consecutive points of a column
    static constexpr int LDS_ELEMS = NPTS * TC;
    static constexpr int TCOLS = TC;
    static constexpr int NSTEP = ST::NSTEP;
    static constexpr int TILES = (1 << (LOGN - P)) / TC; // tiles per limb (per outer block when S0 > 0)

    // `base` points at the first element of this tile's column 0, point 0.
    // step index e counts in execution order (for the inverse the field order is reversed).
    template <int E>
    static FHE_D void step(int tid, u64 *__restrict__ base, elem *__restrict__ lds, const Tw *__restrict__ tw,
                           u32 hi_prefix, const typename A::Ctx &c, const Tw &inv_n)
    {
        constexpr int F = INVERSE ? ST::NSTEP - 1 - E : E;   // which field (forward numbering) this step handles
        constexpr int K = ST::k(F);
        constexpr int R = 1 << K;
        constexpr int DONE = ST::done(F);                    // forward stages above this field inside the pass
        constexpr int LOGS = P - DONE - K;                   // log2 of the point stride of this field
        constexpr u32 S = 1u << LOGS;
        constexpr int NSETS = NPTS / R;
        constexpr bool FIRST = E == 0, LAST = E == ST::NSTEP - 1;
        constexpr int U0 = INVERSE ? (P - DONE - K) : DONE;  // stage offset inside the pass, execution order
#pragma unroll 1
        for (int u = tid; u < NSETS * TC; u += NTHREADS) {
            const u32 col = (u32)u % TC, su = (u32)u / TC;
            const u32 a = su >> LOGS, cc = su & (S - 1);
            const u32 g0 = ((a << K) << LOGS) + cc;          // point index of register 0
            elem x[R];
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u64 raw = base[(size_t)(g0 + ((u32)r << LOGS)) * STRIDE + col];
                    if (IN_MODE == IO_LAZY) x[r] = A::load_lazy(raw);
                    else {
                        if (!A::in_range(raw, c)) raw = reduce_any_u64(raw, c.q);
                        x[r] = A::from_canonical(raw);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = lds[(g0 + ((u32)r << LOGS)) * TC + col];
            }
            const u32 prefix = (hi_prefix << DONE) | a;
            if (INVERSE) radix_inv<A, K, RED, U0>(x, tw, S0 + DONE, prefix, c);
            else radix_fwd<A, K, RED, U0>(x, tw, S0 + DONE, prefix, c);
            if (LAST) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u64 out;
                    if (OUT_MODE == IO_LAZY) out = A::store_lazy(x[r]);
                    else if (INVERSE) out = A::canonical(A::mulmod(x[r], inv_n, c), c);
                    else out = A::canonical(x[r], c);
                    base[(size_t)(g0 + ((u32)r << LOGS)) * STRIDE + col] = out;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) lds[(g0 + ((u32)r << LOGS)) * TC + col] = x[r];
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Row pass: the last P forward stages (pair distance < 2^P) on a tile of TR
// contiguous rows of 2^P points.
// ---------------------------------------------------------------------------
template <class A, class ST, int LOGN, int TR, int NTHREADS, bool INVERSE, int IN_MODE, int OUT_MODE, u32 RED>
struct RowPass {
    typedef A Arith;
    typedef typename A::elem elem;
    static constexpr int P = ST::P;
    static constexpr int NPTS = 1 << P;
    static constexpr int S0 = LOGN - P;
    static constexpr int ROW_LDS = NPTS + 2 * ((NPTS + 15) / 16); // padded row, in elements
    static constexpr int LDS_ELEMS = ROW_LDS * TR;
    static constexpr int TROWS = TR;
    static constexpr int NSTEP = ST::NSTEP;
    static constexpr int TILES = (1 << S0) / TR;

    // `base` = first element of the tile's first row; `row0` = index of that row in the limb.
    template <int E>
    static FHE_D void step(int tid, u64 *__restrict__ base, elem *__restrict__ lds, const Tw *__restrict__ tw,
                           u32 row0, const typename A::Ctx &c, const Tw &inv_n)
    {
        constexpr int F = INVERSE ? ST::NSTEP - 1 - E : E;
        constexpr int K = ST::k(F);
        constexpr int R = 1 << K;
        constexpr int DONE = ST::done(F);
        constexpr int LOGS = P - DONE - K;
        constexpr u32 S = 1u << LOGS;
        constexpr int NSETS = NPTS / R;
        constexpr bool FIRST = E == 0, LAST = E == ST::NSTEP - 1;
        constexpr int U0 = INVERSE ? (P - DONE - K) : DONE;
#pragma unroll 1
        for (int u = tid; u < NSETS * TR; u += NTHREADS) {
            const u32 row = (u32)u / NSETS, su = (u32)u % NSETS;
            const u32 a = su >> LOGS, cc = su & (S - 1);
            const u32 g0 = ((a << K) << LOGS) + cc;
            u64 *__restrict__ grow = base + (size_t)row * NPTS;
            elem *__restrict__ lrow = lds + row * ROW_LDS;
            elem x[R];
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u64 raw = grow[g0 + ((u32)r << LOGS)];
                    if (IN_MODE == IO_LAZY) x[r] = A::load_lazy(raw);
                    else {
                        if (!A::in_range(raw, c)) raw = reduce_any_u64(raw, c.q);
                        x[r] = A::from_canonical(raw);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = lrow[row_pad(g0 + ((u32)r << LOGS))];
            }
            const u32 prefix = ((row0 + row) << DONE) | a;
            if (INVERSE) radix_inv<A, K, RED, U0>(x, tw, S0 + DONE, prefix, c);
            else radix_fwd<A, K, RED, U0>(x, tw, S0 + DONE, prefix, c);
            if (LAST) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u64 out;
                    if (OUT_MODE == IO_LAZY) out = A::store_lazy(x[r]);
                    else if (INVERSE) out = A::canonical(A::mulmod(x[r], inv_n, c), c);
                    else out = A::canonical(x[r], c);
                    grow[g0 + ((u32)r << LOGS)] = out;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) lrow[row_pad(g0 + ((u32)r << LOGS))] = x[r];
            }
        }
    }
};

} // namespace fhe
