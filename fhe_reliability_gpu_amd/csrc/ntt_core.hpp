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
// exchanged through LDS between such steps.  A register step whose points are
// adjacent in memory (stride 1) never touches global memory directly: the tile is
// staged through LDS so that HBM only sees whole-line, lane-contiguous accesses.
//
// Everything here is callable per (thread id, phase) so that tests/emu can run the
// exact same code on the CPU with a loop standing in for the workgroup.
#pragma once
#include "modarith.hpp"

namespace fhe {

// the low `bits` bits of v reversed (bits <= 32)
FHE_HD u32 brev_bits(u32 v, int bits)
{
    if (bits <= 0) return 0;
#if defined(__clang__)
    return __builtin_bitreverse32(v) >> (32 - bits);
#else
    u32 r = 0;
    for (int b = 0; b < bits; b++) r |= ((v >> b) & 1u) << (bits - 1 - b);
    return r;
#endif
}

FHE_HD constexpr int cmax(int a, int b) { return a > b ? a : b; }
FHE_HD constexpr int cmin(int a, int b) { return a < b ? a : b; }

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

// Table layout.  Entry for (stage sigma, block i) is at 2^sigma + i for sigma < SBLK ("natural").
// The stages from SBLK on -- the ones the stride-1 register step of the row pass handles -- are
// stored "blocked": with u = sigma - SBLK, p = i >> u, b = i & (2^u - 1) the entry sits at
//   2^SBLK * (2^u + b) + p
// so that lanes holding consecutive p (consecutive 16-point groups of a row) read consecutive
// 16-byte entries instead of one cache line each (measured: up to 16 % of a 2^16 transform).
FHE_HD constexpr u32 tw_index(int sblk, int sigma, u32 i)
{
    return sigma < sblk ? (1u << sigma) + i
                        : (((1u << (sigma - sblk)) + (i & ((1u << (sigma - sblk)) - 1u))) << sblk) + (i >> (sigma - sblk));
}

// Where a register step takes its twiddles from.  The default source is the limb's table in global memory (TwPtr); a row pass that
// runs several transforms of ONE limb on the same rows (the key switch's fused inner product: every digit's extension of a limb) can
// instead keep the rows' factors in LDS, 8 bytes each, and form the quotient w/q as w * (1/q) on the fly (RowTwLds): the factors then
// cross the fabric once per workgroup instead of once per transform -- they weigh twice what the data does (16 bytes per point).
template <int SBLK> FHE_D Tw tw_get(TwPtr tw, int sigma, u32 i) { return tw[tw_index(SBLK, sigma, i)]; }
struct RowTwLds {
    const double *w;   // [rows of the tile][2^P - 1]: row-major, inside a row stage after stage (stage S0 + t at offset 2^t - 1, block j)
    double ninv;       // 1 / q
    u32 row0;          // first row of the tile
    int s0, per_row;   // S0 (first stage of the row pass), 2^P - 1
};
// (the quotient estimate w * ninv is within 1.5 ulp of w/q instead of 0.5: a product's magnitude bound grows from (0.5 + B/4) q to
// (0.5 + 3B/8) q for an input of magnitude B q -- passes that use this source reduce every fourth stage)
template <int SBLK> FHE_D Tw tw_get(const RowTwLds &t, int sigma, u32 i)
{
    const int tt = sigma - t.s0;
    const u32 row = (i >> tt) - t.row0, j = i & ((1u << tt) - 1u);
    const double w = t.w[row * (u32)t.per_row + ((1u << tt) - 1u) + j];
    return Tw{double_to_u64_bits(w), double_to_u64_bits(w * t.ninv)};
}

// K forward stages on R = 2^K registers.  Register r holds the point whose K-bit
// field (most significant bit = first stage) equals r.  `s` = global index of the
// first stage, `prefix` = value of the s index bits above the field.
template <class A, int K, u32 RED, int U0, int SBLK, class TWS = TwPtr>
FHE_D void radix_fwd(typename A::elem (&x)[1 << K], const TWS &tw, int s, u32 prefix, const typename A::Ctx &c)
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
            const Tw w = tw_get<SBLK>(tw, s + u, (prefix << u) + b);
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_fwd(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// K inverse stages, undoing radix_fwd: forward stage s+K-1 first.
// FOLD (only where the step ends with forward stage 0, i.e. s == 0, and the pass writes final words): N^-1 is folded
// into that last stage -- X' = (X + Y) N^-1, Y' = (X - Y) (w N^-1) with the product w N^-1 kept in the inverse
// table's otherwise unused entry 0 -- instead of multiplying every output afterwards: one product per butterfly less.
template <class A, int K, u32 RED, int U0, int SBLK, bool FOLD = false>
FHE_D void radix_inv(typename A::elem (&x)[1 << K], TwPtr tw, int s, u32 prefix, const typename A::Ctx &c, const Tw *inv_n = nullptr)
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
        if (FOLD && u == 0) {
            const Tw wn = tw[0];
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_inv_scaled(x[j], x[j + half], *inv_n, wn, c);
            continue;
        }
#pragma unroll
        for (int b = 0; b < (1 << u); b++) {
            const Tw w = tw[tw_index(SBLK, s + u, (prefix << u) + b)];
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_inv(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
}

// ---------------------------------------------------------------------------
// Inverse passes of ArithF64 with the lazy range tracked PER REGISTER (RED = INV_LAZY | entry bound in eighths of q).
// A Gentleman-Sande stage maps bounds (bX, bY) (units of q) to (bX + bY, 0.5 + (bX + bY)/4): only a register that keeps taking the SUM
// branch grows, every product output starts again near q/2.  Which branch a register takes at each stage of a register step is
// known at compile time, so instead of folding all 2^K registers back every second or third stage (reduce_mask above) the plan below
// folds exactly the registers whose pair would exceed 8q < 2^53 (every sum, difference and product input stays an exact integer
// in a double: tests/ + the derivation in DESIGN.md "lazy FP64 ranges"), and at the end of a step those above 2q -- a thread of the NEXT
// step holds values from one register index of this step, unknown at compile time there, so steps hand over under a common bound.
// 2^16: 2.7 folds per point instead of 5.
// ---------------------------------------------------------------------------
constexpr u32 INV_LAZY = 0x80000000u;
constexpr int INV_LAZY_LIMIT8 = 64;        // 8 q
constexpr int INV_LAZY_EXIT8 = 16;         // 2 q: bound under which a step hands its registers over
template <int K> struct InvLazyPlan {
    u32 before[K];      // bit r: fold register r before executed stage v
    u32 at_exit;        // bit r: fold register r after the last stage
    int out8;           // bound of every register afterwards, eighths of q, rounded up
};
// bounds are kept in 1/1024 q, rounded up at every operation (a fold leaves |x| <= q/2 + an ulp of the quotient: 513/1024;
// a product 0.5 + (bX + bY)/4 + the same slack)
template <int K> FHE_HD constexpr InvLazyPlan<K> inv_lazy_plan(int in8, int exit8, bool fold_last)
{
    constexpr int R = 1 << K;
    InvLazyPlan<K> p{};
    int b[R] = {};
    for (int r = 0; r < R; r++) b[r] = in8 * 128;
    const int limit = INV_LAZY_LIMIT8 * 128, folded = 513;
    for (int v = 0; v < K; v++) {
        const int u = K - 1 - v, half = R >> (u + 1);
        u32 m = 0;
        for (int blk = 0; blk < (1 << u); blk++)
            for (int j = 0; j < half; j++) {
                const int i0 = blk * 2 * half + j, i1 = i0 + half;
                if (b[i0] + b[i1] > limit) {
                    const int big = b[i0] >= b[i1] ? i0 : i1;
                    b[big] = folded;
                    m |= 1u << big;
                    if (b[i0] + b[i1] > limit) {
                        const int other = big == i0 ? i1 : i0;
                        b[other] = folded;
                        m |= 1u << other;
                    }
                }
                const int sum = b[i0] + b[i1], prod = 513 + (sum + 3) / 4;
                b[i0] = (fold_last && u == 0) ? prod : sum;
                b[i1] = prod;
            }
        p.before[v] = m;
    }
    int mx = 0;
    for (int r = 0; r < R; r++) {
        if (b[r] > exit8 * 128) {
            b[r] = folded;
            p.at_exit |= 1u << r;
        }
        mx = b[r] > mx ? b[r] : mx;
    }
    p.out8 = (mx + 127) / 128;
    return p;
}
// entry bound of executed step `se` of an inverse pass whose first step starts from in8 (ST = the pass's Steps; the inverse runs them last to first)
template <class ST> FHE_HD constexpr int inv_lazy_step_in8(int in8, int se)
{
    int b = in8;
    for (int i = 0; i < se; i++) {
        const int k = ST::k(ST::NSTEP - 1 - i);
        b = k == 1 ? inv_lazy_plan<1>(b, INV_LAZY_EXIT8, false).out8 : k == 2 ? inv_lazy_plan<2>(b, INV_LAZY_EXIT8, false).out8
          : k == 3 ? inv_lazy_plan<3>(b, INV_LAZY_EXIT8, false).out8 : k == 4 ? inv_lazy_plan<4>(b, INV_LAZY_EXIT8, false).out8
                                                                             : inv_lazy_plan<5>(b, INV_LAZY_EXIT8, false).out8;
    }
    return b;
}
// bound a pass that hands lazy words to the next launch leaves behind
template <class ST> FHE_HD constexpr int inv_lazy_pass_out8(int in8) { return inv_lazy_step_in8<ST>(in8, ST::NSTEP); }

template <class A, int K, int SBLK, bool FOLD, int IN8, int EXIT8>
FHE_D void radix_inv_lazy(typename A::elem (&x)[1 << K], TwPtr tw, int s, u32 prefix, const typename A::Ctx &c, const Tw *inv_n)
{
    constexpr int R = 1 << K;
    constexpr InvLazyPlan<K> plan = inv_lazy_plan<K>(IN8, EXIT8, FOLD);
#pragma unroll
    for (int v = 0; v < K; v++) {
        const int u = K - 1 - v;
#pragma unroll
        for (int r = 0; r < R; r++)
            if ((plan.before[v] >> r) & 1) A::reduce(x[r], c);
        const int half = R >> (u + 1);
        if (FOLD && u == 0) {
            const Tw wn = tw[0];
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_inv_scaled(x[j], x[j + half], *inv_n, wn, c);
            continue;
        }
#pragma unroll
        for (int b = 0; b < (1 << u); b++) {
            const Tw w = tw[tw_index(SBLK, s + u, (prefix << u) + b)];
#pragma unroll
            for (int j = 0; j < half; j++) A::bfly_inv(x[b * 2 * half + j], x[b * 2 * half + j + half], w, c);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        if ((plan.at_exit >> r) & 1) A::reduce(x[r], c);
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

enum IoMode { IO_CANONICAL = 0, IO_LAZY = 1 };

// raw 64-bit words -> arithmetic elements, with ONE branch for the out-of-range case
template <class A, int R, int IN_MODE>
FHE_D void convert_in(typename A::elem (&x)[R], u64 (&raw)[R], const typename A::Ctx &c)
{
    if (IN_MODE == IO_LAZY) {
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = A::load_lazy(raw[r]);
    } else {
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; r++) bad |= !A::in_range(raw[r], c);
        if (__builtin_expect(bad, 0)) {
#pragma unroll
            for (int r = 0; r < R; r++) raw[r] = A::in_range(raw[r], c) ? raw[r] : reduce_any_u64(raw[r], c.q);
        }
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = A::from_canonical(raw[r]);
    }
}

template <class A, int OUT_MODE, bool INVERSE, bool SCALED = false>
FHE_D u64 convert_out(typename A::elem x, const typename A::Ctx &c, const Tw &inv_n)
{
    if (OUT_MODE == IO_LAZY) return A::store_lazy(x);
    if (INVERSE && !SCALED) return A::canonical(A::mulmod(x, inv_n, c), c);   // SCALED: N^-1 already folded into the last stage
    return A::canonical(x, c);
}

// ---------------------------------------------------------------------------
// Packed hand-off between the two passes (forward, FP64 path, 2^16): the intermediate is private to the
// transform, so it does not have to be 8-byte words in place.  Canonical residues below 2^50 are packed 16 to
// a 104-byte chunk (13 words: 800 bits + 32 bits of padding); 16 chunks -- 16 consecutive points of 16 adjacent
// columns -- make a 1664-byte block (13 cache lines instead of 16).  A column tile produces the 16 blocks of one
// block column (26,624 contiguous bytes), a row tile consumes the 16 blocks of one block row, both staged through
// LDS so that the fabric only sees 16 bytes per lane, lanes contiguous.  Two of the four sweeps over the batch
// shrink by 19 % (tools/pack_ubench.hip: 79.8 -> 67.0 us per 256 limbs for the bare access patterns).
// ---------------------------------------------------------------------------
constexpr int PK_BITS = 50;
constexpr int PK_WORDS = 13;                       // u64 words per chunk of 16 values
constexpr int PK_BLOCK_WORDS = 16 * PK_WORDS;      // 16 chunks
constexpr u64 PK_MASK = ((u64)1 << PK_BITS) - 1;

// 16 values below 2^50 -> 13 words (compile-time bit offsets)
FHE_HD void pk_pack(const u64 (&v)[16], u64 (&w)[PK_WORDS])
{
#pragma unroll
    for (int j = 0; j < PK_WORDS; j++) w[j] = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int j = (PK_BITS * k) >> 6, t = (PK_BITS * k) & 63;
        w[j] |= v[k] << t;
        if (t > 64 - PK_BITS) w[j + 1] |= v[k] >> (64 - t);
    }
}
// value number `idx` (run-time) of the chunk at `w`
template <class P> FHE_HD u64 pk_extract(P w, u32 idx)
{
    const u32 bit = PK_BITS * idx, j = bit >> 6, t = bit & 63;
    const u64 lo = w[j], hi = w[j + 1 < PK_WORDS ? j + 1 : j];     // idx 15 ends in word 12: j + 1 never leaves the chunk
    const u64 v = t > 64 - PK_BITS ? (lo >> t) | (hi << (64 - t)) : lo >> t;
    return v & PK_MASK;
}

// Optional observer of a pass ("tap"): sees every element the pass reads from global memory (after the
// conversion to the arithmetic's element type) and every final word it writes, with the element's index
// relative to the tile base.  The ABFT detector hangs its weighted checksums here (ntt_kernels.hip), so
// that checking a transform costs arithmetic only, not two more sweeps over the data.
// (a tap that declares PRELOAD = true instead offers load(idx) / apply(x, word, ctx): a second input of the pass, see AddSrcTap)
template <class T, class = void> struct tap_preloads { static constexpr bool value = false; };
template <class T> struct tap_preloads<T, decltype((void)T::PRELOAD)> { static constexpr bool value = T::PRELOAD; };
struct NoTap {
    static constexpr bool ACTIVE = false;
    static constexpr bool MID = false;    // MID: the tap also sees the lazy words a column pass hands to the next launch
    static constexpr bool STORES = false; // STORES: the tap writes the forward row pass's results itself (an epilogue fused into the pass)
};

// ---------------------------------------------------------------------------
// Column pass: stages [S0, S0+P) of a transform of 2^LOGN points whose pair
// distances are multiples of STRIDE = 2^(LOGN-S0-P); the tile is TC adjacent
// "columns" (consecutive values of the low index bits).  With S0 = 0 this is the
// first pass of the forward transform / last pass of the inverse.
// LDS image: [point][column], 8-byte elements; TC extra elements after every 16
// points keep the stride-1 step's ds_read_b64 conflict free when TC < 32.
// ---------------------------------------------------------------------------
template <class A, class ST, int LOGN, int S0, int TC, int NTHREADS, bool INVERSE, int IN_MODE, int OUT_MODE, u32 RED,
          int SBLK, int COHERENT_IN = 0, bool STREAM = false>
struct ColPass {
    typedef A Arith;
    typedef typename A::elem elem;
    static constexpr int THREADS = NTHREADS;
    static constexpr int P = ST::P;
    static constexpr int NPTS = 1 << P;
    static constexpr int LOGSTRIDE = LOGN - S0 - P;
    static constexpr u32 STRIDE = 1u << LOGSTRIDE;      // distance between consecutive points of a column
    static constexpr int PADC = (TC < 32 && (NPTS * TC + ((NPTS + 15) / 16) * TC) * 8 <= 65536) ? TC : 0;
    static constexpr int LDS_ELEMS = NPTS * TC + ((NPTS + 15) / 16) * PADC;
    static constexpr int TCOLS = TC;
    static constexpr int NSTEP = ST::NSTEP;
    static constexpr int NPHASE = ST::NSTEP;             // barrier-separated phases
    static constexpr int TILES = (1 << (LOGN - P)) / TC; // tiles per limb (per outer block when S0 > 0)

    static FHE_HD u32 lidx(u32 g, u32 col) { return g * TC + col + (g >> 4) * PADC; }

    // `base` points at the first element of this tile's column 0, point 0.
    // Phase index E counts in execution order (for the inverse the field order is reversed).
    // `src` (optional): the loading step reads the tile from there (same offsets) instead of `base` -- out-of-place first launch.
    template <int E, class TAP = NoTap>
    static FHE_D void phase(int tid, u64 *__restrict__ base, elem *__restrict__ lds, TwPtr tw, u32 hi_prefix,
                            const typename A::Ctx &c, const Tw &inv_n, TAP *tap = nullptr, const u64 *src = nullptr)
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
        FHE_ASSUME(tid >= 0 && tid < NTHREADS);
#pragma unroll 1
        for (int u = tid; u < NSETS * TC; u += NTHREADS) {
            const u32 col = (u32)u % TC, su = (u32)u / TC;
            // DONE == 0: the field is the top of the index, a is identically 0 (keeps the twiddle index uniform)
            const u32 a = DONE == 0 ? 0u : su >> LOGS, cc = DONE == 0 ? su : su & (S - 1);
            const u32 g0 = ((a << K) << LOGS) + cc;          // point index of register 0
            elem x[R];
            if (FIRST) {
                u64 raw[R];
                const u64 *ldb = src ? src : base;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const u64 *from = ldb + (size_t)(g0 + ((u32)r << LOGS)) * STRIDE + col;
                    raw[r] = COHERENT_IN == 1 ? load_coherent_u64(from) : (COHERENT_IN == 2 || (STREAM && IN_MODE == IO_CANONICAL)) ? load_stream_u64(from) : *from;
                }
                if constexpr (tap_preloads<TAP>::value) {
                    // a tap with a second input: its words are requested together with the tile's, before anything waits
                    u64 extra[R];
#pragma unroll
                    for (int r = 0; r < R; r++) extra[r] = tap->load((g0 + ((u32)r << LOGS)) * STRIDE + col);
                    convert_in<A, R, IN_MODE>(x, raw, c);
                    tap->prepare(extra, c);
#pragma unroll
                    for (int r = 0; r < R; r++) tap->apply(x[r], extra[r], c);
                } else {
                    convert_in<A, R, IN_MODE>(x, raw, c);
                    if constexpr (TAP::ACTIVE) {
#pragma unroll
                        for (int r = 0; r < R; r++) tap->in((g0 + ((u32)r << LOGS)) * STRIDE + col, x[r], c);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = lds[lidx(g0 + ((u32)r << LOGS), col)];
            }
            const u32 prefix = (hi_prefix << DONE) | a;
            constexpr bool FOLD = INVERSE && LAST && OUT_MODE == IO_CANONICAL && S0 + DONE == 0;
            if constexpr (INVERSE && (RED & INV_LAZY) != 0 && A::PATH == PATH_F64) {
                constexpr int IN8 = inv_lazy_step_in8<ST>((int)(RED & 0xFFu), E);
                constexpr int EX8 = (LAST && OUT_MODE == IO_CANONICAL) ? INV_LAZY_LIMIT8 : INV_LAZY_EXIT8;
                radix_inv_lazy<A, K, SBLK, FOLD, IN8, EX8>(x, tw, S0 + DONE, prefix, c, &inv_n);
            } else if (INVERSE) radix_inv<A, K, RED, U0, SBLK, FOLD>(x, tw, S0 + DONE, prefix, c, &inv_n);
            else radix_fwd<A, K, RED, U0, SBLK>(x, tw, S0 + DONE, prefix, c);
            if (LAST) {
                if constexpr (TAP::ACTIVE) {
                    if constexpr (TAP::MID) {
#pragma unroll
                        for (int r = 0; r < R; r++) tap->mid((g0 + ((u32)r << LOGS)) * STRIDE + col, x[r], c);
                    }
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    u64 *dst = base + (size_t)(g0 + ((u32)r << LOGS)) * STRIDE + col;
                    const u64 out = convert_out<A, OUT_MODE, INVERSE, FOLD>(x[r], c, inv_n);
                    if (STREAM && OUT_MODE == IO_CANONICAL) store_stream_u64(dst, out);
                    else *dst = out;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) lds[lidx(g0 + ((u32)r << LOGS), col)] = x[r];
            }
        }
    }
    // ---- packed hand-off, producer side (forward column pass whose last step owns 16 consecutive points) ----
    static constexpr bool PACKABLE = !INVERSE && A::PATH == PATH_F64 && TC == 16 && NTHREADS == 256 && NPTS == 256 && ST::NSTEP == 2 &&
                                     ST::k(1) == 4 && PK_BLOCK_WORDS * 16 <= LDS_ELEMS;
    // last register step, results canonicalised and packed into the thread's chunk (kept in registers over the barrier
    // that protects the LDS image the other threads are still reading)
    static FHE_D void phase_last_packed(int tid, elem *__restrict__ lds, TwPtr tw, u32 hi_prefix, const typename A::Ctx &c, u64 (&chunk)[PK_WORDS])
    {
        constexpr int F = ST::NSTEP - 1, K = ST::k(F), R = 1 << K, DONE = ST::done(F);
        static_assert(R == 16 && P - DONE - K == 0, "one thread = 16 consecutive points of one column");
        const u32 col = (u32)tid % TC, a = (u32)tid / TC, g0 = a << K;
        elem x[R];
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = lds[lidx(g0 + (u32)r, col)];
        radix_fwd<A, K, RED, DONE, SBLK>(x, tw, S0 + DONE, (hi_prefix << DONE) | a, c);
        u64 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = A::canonical(x[r], c);
        pk_pack(v, chunk);
    }
    // chunk -> staging image [block row a][column][13 words] (aliases the pass's LDS image: call after a barrier)
    static FHE_D void pack_stage(int tid, u64 *__restrict__ stage, const u64 (&chunk)[PK_WORDS])
    {
        u64 *dst = stage + (size_t)tid * PK_WORDS;        // tid = a * 16 + col
#pragma unroll
        for (int j = 0; j < PK_WORDS; j++) dst[j] = chunk[j];
    }
    // staging -> scratch: the tile's 16 blocks are contiguous (16 bytes per lane, lanes contiguous)
    static FHE_D void pack_copy_out(int tid, const u64 *__restrict__ stage, u64 *__restrict__ tile_scratch)
    {
        for (int i = tid; i < 16 * PK_BLOCK_WORDS / 2; i += NTHREADS) {
            tile_scratch[2 * i] = stage[2 * i];
            tile_scratch[2 * i + 1] = stage[2 * i + 1];
        }
    }
};

// ---------------------------------------------------------------------------
// Row pass: the last P forward stages (pair distance < 2^P) on a tile of TR
// contiguous rows of 2^P points.
// LDS image: [row][point], 16 bytes of padding after every 16 points (row_pad): the
// stride-1 step moves 128 bytes per lane with ds_read_b128 / ds_write_b128 and stays
// conflict free.  The stride-1 step is the LAST one of a forward pass and the FIRST
// one of an inverse pass; on that side the tile goes HBM <-> LDS in a separate,
// fully coalesced copy phase (16 bytes per lane, lanes contiguous).
// NTT-domain Galois map x -> x^k on a limb in the transform's bit-reversed slot order: slot j evaluates at psi^(2 bitrev(j) + 1)
// and (sigma_k f)(x) = f(x^k), so sigma_k(f)[j] = f[j'] with 2 bitrev(j') + 1 = (2 bitrev(j) + 1) k mod 2N (the index map behind
// phantom::rotate_inplace, reliability_test/dotprod_test.cu:146).  An aligned block of 64 slots maps onto an aligned block of 64
// slots, so a wavefront that reads 64 adjacent slots through the map touches one 512-byte segment.
FHE_HD u32 galois_slot(u32 j, int logn, u32 k)
{
    const u32 e = 2 * brev_bits(j, logn) + 1;
    const u32 e2 = (u32)(((u64)e * k) & ((2ull << logn) - 1ull));
    return brev_bits((e2 - 1) >> 1, logn);
}

// ---------------------------------------------------------------------------
FHE_HD constexpr u32 row_pad(u32 g) { return g + ((g >> 4) << 1); }

// STAGE_BOTH: also stage the side that is not stride-1 (forward: the input) through a coalesced copy phase (tried for
// the LDS-resident single pass: no gain, see ntt_plan.hpp; kept as a switch).
// ROWMODE 1 (first launch of the natural-order transforms, INVERSE network only): the tile's TR rows are the network rows
// r_i = bitrev(row0 + i) and its input comes from a NATURAL-order source -- network position (r, k) holds
// src[bitrev(k) * 2^S0 + bitrev(r)], so the tile reads TR adjacent columns of the source seen as 2^P rows of 2^S0 words
// (whole 8*TR-byte segments) and the bit reversal happens on the way into LDS.  That is the four-step's "column
// transforms" batch reading its columns (reliability_test/four_step_ntt_prot.py:81-90) with no separate transpose.
template <class A, class ST, int LOGN, int TR, int NTHREADS, bool INVERSE, int IN_MODE, int OUT_MODE, u32 RED,
          int SBLK, int COHERENT_IN = 0, bool STREAM = false, bool STAGE_BOTH = false, int ROWMODE = 0>
struct RowPass {
    typedef A Arith;
    typedef typename A::elem elem;
    static constexpr int THREADS = NTHREADS;
    static constexpr int P = ST::P;
    static constexpr int NPTS = 1 << P;
    static constexpr int S0 = LOGN - P;
    static constexpr int ROW_LDS = NPTS + 2 * ((NPTS + 15) / 16); // padded row, in elements
    static constexpr int LDS_ELEMS = ROW_LDS * TR;
    static constexpr int TROWS = TR;
    static constexpr int NSTEP = ST::NSTEP;
    static constexpr bool STAGED = NPTS >= 32;                   // copy phase on the stride-1 side
    static constexpr bool STAGE_IN = STAGED && (INVERSE || STAGE_BOTH), STAGE_OUT = STAGED && (!INVERSE || STAGE_BOTH);
    static constexpr int NPHASE = ST::NSTEP + (STAGE_IN ? 1 : 0) + (STAGE_OUT ? 1 : 0);
    static constexpr int TILES = (1 << S0) / TR;
    static_assert(ROWMODE == 0 || (INVERSE && NPTS >= 32), "the gathering first launch belongs to the inverse-structured network");

    // network row of the tile's local row
    static FHE_HD u32 net_row(u32 row0, u32 row)
    {
        if (ROWMODE == 0) return row0 + row;
        return brev_bits(row0 + row, S0);
    }
    // ROWMODE 1: natural-order source (limb base) -> LDS image, bit reversal folded in
    static FHE_D void gather_in(int tid, const u64 *__restrict__ src, u32 row0, elem *__restrict__ lds, bool nt_src)
    {
        // (unrolled: the loads of several rounds are in flight together; one round at a time this phase is bound by their latency --
        // 92 -> 70 us per 128 MiB, the plain inverse row pass takes 69: profiles/r02_fourstep_kernels.txt)
#pragma unroll 8
        for (int i = tid; i < TR * NPTS; i += NTHREADS) {
            const u32 ki = (u32)i % TR, rho = (u32)i / TR;
            const u32 k = brev_bits(rho, P);
            const u64 *from = src + ((size_t)rho << S0) + row0 + ki;
            const u64 v = nt_src ? load_stream_u64(from) : *from;      // (pieces of a batch that streams from HBM: touched once)
            lds[ki * ROW_LDS + row_pad(k)] = __builtin_bit_cast(elem, v);
        }
    }

    // coalesced copy HBM -> LDS (inverse, raw words): 2 points (16 bytes) per lane
    static FHE_D void copy_in(int tid, const u64 *__restrict__ base, elem *__restrict__ lds)
    {
        // (unroll 8 -- all of a thread's loads in flight at once -- measured slower: 0.398 against 0.391 ms per 512 MiB inverse, and slower for
        // the key switch's small INTT launches as well: 7.2 / 11.5 / 24.7 us against 6.4 / 10.9 / 23.4 at 16 / 256 / 704 workgroups)
#pragma unroll 2
        for (int i = tid; i < TR * NPTS / 2; i += NTHREADS) {
            const u32 row = (u32)i / (NPTS / 2), g = ((u32)i % (NPTS / 2)) * 2;
            const u64 *src = base + (size_t)row * NPTS + g;
            u64 v0, v1;
            if (COHERENT_IN == 1) {
                v0 = load_coherent_u64(src);
                v1 = load_coherent_u64(src + 1);
            } else if (COHERENT_IN == 2 || (STREAM && IN_MODE == IO_CANONICAL)) {
                v0 = load_stream_u64(src);
                v1 = load_stream_u64(src + 1);
            } else {
                v0 = src[0];
                v1 = src[1];
            }
            elem *dst = lds + row * ROW_LDS + row_pad(g);
            dst[0] = __builtin_bit_cast(elem, v0);
            dst[1] = __builtin_bit_cast(elem, v1);
        }
    }
    // the same tile read through the Galois map: LDS image of sigma_k(src) (src_limb = the source limb's first word); one word per
    // lane, so a wavefront reads one 512-byte segment; `copy` (optional) receives the mapped tile as is
    static FHE_D void copy_in_galois(int tid, const u64 *__restrict__ src_limb, u32 row0, elem *__restrict__ lds, u32 k, u64 *__restrict__ copy)
    {
#pragma unroll 8
        for (int i = tid; i < TR * NPTS; i += NTHREADS) {
            const u32 row = (u32)i / NPTS, g = (u32)i % NPTS;
            const u64 v = src_limb[galois_slot((row0 + row) * NPTS + g, LOGN, k)];
            lds[row * ROW_LDS + row_pad(g)] = __builtin_bit_cast(elem, v);
            if (copy) copy[i] = v;
        }
    }
    // coalesced copy LDS -> HBM (forward; the image already holds final 64-bit words)
    template <class TAP = NoTap>
    static FHE_D void copy_out(int tid, u64 *__restrict__ base, const elem *__restrict__ lds, const typename A::Ctx &c, TAP *tap = nullptr)
    {
#pragma unroll 2
        for (int i = tid; i < TR * NPTS / 2; i += NTHREADS) {
            const u32 row = (u32)i / (NPTS / 2), g = ((u32)i % (NPTS / 2)) * 2;
            const elem *src = lds + row * ROW_LDS + row_pad(g);
            u64 *dst = base + (size_t)row * NPTS + g;
            if constexpr (TAP::ACTIVE) {
                if constexpr (TAP::STORES) {
                    tap->store(row * NPTS + g, __builtin_bit_cast(u64, src[0]), __builtin_bit_cast(u64, src[1]));
                    continue;
                } else {
                    tap->out(row * NPTS + g, __builtin_bit_cast(u64, src[0]), c);
                    tap->out(row * NPTS + g + 1, __builtin_bit_cast(u64, src[1]), c);
                }
            }
            if (STREAM && OUT_MODE == IO_CANONICAL) {
                store_stream_u64(dst, __builtin_bit_cast(u64, src[0]));
                store_stream_u64(dst + 1, __builtin_bit_cast(u64, src[1]));
            } else {
                dst[0] = __builtin_bit_cast(u64, src[0]);
                dst[1] = __builtin_bit_cast(u64, src[1]);
            }
        }
    }

    // `base` = first element of the tile's first row; `row0` = index of that row in the limb.  (ROWMODE 1: `base` = the
    // LIMB's first element -- the tile's rows are not adjacent -- and `src` = the natural-order source limb.)
    // `src` (optional, ROWMODE 0): the loading step reads the tile from there instead of `base` -- out-of-place first launch.
    template <int E, class TAP = NoTap, class TWS = TwPtr>
    static FHE_D void phase(int tid, u64 *__restrict__ base, elem *__restrict__ lds, const TWS &tw, u32 row0,
                            const typename A::Ctx &c, const Tw &inv_n, TAP *tap = nullptr, const u64 *src = nullptr, u32 galois = 0,
                            u64 *galois_copy = nullptr, bool nt_src = false)
    {
        if constexpr (STAGE_IN && E == 0) {
            if constexpr (ROWMODE == 1) gather_in(tid, src, row0, lds, nt_src);
            else if (galois && src) copy_in_galois(tid, src - (size_t)row0 * NPTS, row0, lds, galois, galois_copy);
            else copy_in(tid, src ? src : base, lds);
        } else if constexpr (STAGE_OUT && E == NPHASE - 1) {
            copy_out<TAP>(tid, base, lds, c, tap);
        } else {
            constexpr int SE = STAGE_IN ? E - 1 : E;   // register-step index in execution order
            constexpr int F = INVERSE ? ST::NSTEP - 1 - SE : SE;
            constexpr int K = ST::k(F);
            constexpr int R = 1 << K;
            constexpr int DONE = ST::done(F);
            constexpr int LOGS = P - DONE - K;
            constexpr u32 S = 1u << LOGS;
            constexpr int NSETS = NPTS / R;
            constexpr bool FIRST = SE == 0, LAST = SE == ST::NSTEP - 1;
            constexpr bool FROM_GLOBAL = FIRST && !STAGE_IN;
            constexpr bool TO_GLOBAL = LAST && !STAGE_OUT;
            constexpr int U0 = INVERSE ? (P - DONE - K) : DONE;
            FHE_ASSUME(tid >= 0 && tid < NTHREADS);
#pragma unroll 1
            for (int u = tid; u < NSETS * TR; u += NTHREADS) {
                const u32 row = (u32)u / NSETS, su = (u32)u % NSETS;
                const u32 a = su >> LOGS, cc = su & (S - 1);
                const u32 g0 = ((a << K) << LOGS) + cc;
                u64 *__restrict__ grow = base + (size_t)(ROWMODE == 1 ? net_row(row0, row) : row) * NPTS;
                elem *__restrict__ lrow = lds + row * ROW_LDS;
                elem x[R];
                if (FIRST) {
                    u64 raw[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (FROM_GLOBAL) {
                            const u64 *from = (src ? src + (size_t)row * NPTS : grow) + g0 + ((u32)r << LOGS);
                            raw[r] = COHERENT_IN == 1 ? load_coherent_u64(from) : (COHERENT_IN == 2 || (STREAM && IN_MODE == IO_CANONICAL)) ? load_stream_u64(from) : *from;
                        } else {
                            raw[r] = __builtin_bit_cast(u64, lrow[row_pad(g0 + ((u32)r << LOGS))]);
                        }
                    }
                    convert_in<A, R, IN_MODE>(x, raw, c);
                    if constexpr (TAP::ACTIVE && FROM_GLOBAL) {
#pragma unroll
                        for (int r = 0; r < R; r++) tap->in(row * NPTS + g0 + ((u32)r << LOGS), x[r], c);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) x[r] = lrow[row_pad(g0 + ((u32)r << LOGS))];
                }
                const u32 prefix = (net_row(row0, row) << DONE) | a;
                constexpr bool FOLD = INVERSE && LAST && OUT_MODE == IO_CANONICAL && S0 + DONE == 0;
                if constexpr (INVERSE && (RED & INV_LAZY) != 0 && A::PATH == PATH_F64) {
                    constexpr int IN8 = inv_lazy_step_in8<ST>((int)(RED & 0xFFu), SE);
                    constexpr int EX8 = (LAST && OUT_MODE == IO_CANONICAL) ? INV_LAZY_LIMIT8 : INV_LAZY_EXIT8;
                    radix_inv_lazy<A, K, SBLK, FOLD, IN8, EX8>(x, tw, S0 + DONE, prefix, c, &inv_n);
                } else if constexpr (INVERSE) radix_inv<A, K, RED, U0, SBLK, FOLD>(x, tw, S0 + DONE, prefix, c, &inv_n);
                else radix_fwd<A, K, RED, U0, SBLK>(x, tw, S0 + DONE, prefix, c);
                if (LAST) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const u64 out = convert_out<A, OUT_MODE, INVERSE, FOLD>(x[r], c, inv_n);
                        if (TO_GLOBAL) {
                            if (STREAM && OUT_MODE == IO_CANONICAL) store_stream_u64(grow + g0 + ((u32)r << LOGS), out);
                            else grow[g0 + ((u32)r << LOGS)] = out;
                        }
                        else lrow[row_pad(g0 + ((u32)r << LOGS))] = __builtin_bit_cast(elem, out);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; r++) lrow[row_pad(g0 + ((u32)r << LOGS))] = x[r];
                }
            }
        }
    }
    // ---- packed hand-off, consumer side (forward row pass whose first step gathers 16 points 16 apart) ----
    static constexpr bool PACKABLE = !INVERSE && A::PATH == PATH_F64 && TR == 16 && NTHREADS == 256 && NPTS == 256 && ST::NSTEP == 2 &&
                                     ST::k(0) == 4 && PK_BLOCK_WORDS * 16 <= LDS_ELEMS;
    // scratch -> staging: block (P, C) of the unit for C = 0..15 (each 1664 contiguous bytes, 26,624 bytes apart)
    static FHE_D void unpack_copy_in(int tid, u64 *__restrict__ stage, const u64 *__restrict__ unit_scratch, u32 P_)
    {
        for (int i = tid; i < 16 * PK_BLOCK_WORDS / 2; i += NTHREADS) {
            const u32 C = (u32)i / (PK_BLOCK_WORDS / 2), o = (u32)i % (PK_BLOCK_WORDS / 2);
            const u64 *src = unit_scratch + ((size_t)C * 16 + P_) * PK_BLOCK_WORDS + 2 * o;
            stage[2 * i] = src[0];
            stage[2 * i + 1] = src[1];
        }
    }
    // first register step fed from the staging image; the results stay in registers over the barrier that protects the
    // staging image (it aliases the pass's own LDS image)
    static FHE_D void phase_first_packed(int tid, const u64 *__restrict__ stage, TwPtr tw, u32 row0, const typename A::Ctx &c, elem (&x)[16])
    {
        constexpr int K = ST::k(0), LOGS = P - K;
        static_assert(K == 4 && LOGS == 4, "one thread = 16 points 16 apart");
        const u32 row = (u32)tid / 16, cc = (u32)tid % 16;
#pragma unroll
        for (int r = 0; r < 16; r++) x[r] = A::from_canonical(pk_extract(stage + ((size_t)r * 16 + cc) * PK_WORDS, row));
        radix_fwd<A, K, RED, 0, SBLK>(x, tw, S0, row0 + row, c);
    }
    static FHE_D void phase_first_store(int tid, elem *__restrict__ lds, const elem (&x)[16])
    {
        const u32 row = (u32)tid / 16, cc = (u32)tid % 16;
        elem *__restrict__ lrow = lds + row * ROW_LDS;
#pragma unroll
        for (int r = 0; r < 16; r++) lrow[row_pad(cc + ((u32)r << 4))] = x[r];
    }
};

} // namespace fhe
