// ntt_kernels.hip -- gfx950 kernels for the batched forward / inverse NTT.
// One workgroup (256 threads = 4 waves of 64) owns one tile of one limb:
//   column pass: TC adjacent columns x 2^PC points, lanes along the columns;
//   row pass   : TR contiguous rows of 2^PR points, lanes along the row.
// The butterfly code lives in ntt_core.hpp; this file only binds it to the grid.
#include "ntt_launch.hpp"
#include "ntt_plan.hpp"

#include <cstdlib>
#include <type_traits>

namespace fhe {

template <class PASS, bool IS_COL> struct PASS_TCOLS { static constexpr int value = 0; };
template <class PASS> struct PASS_TCOLS<PASS, true> { static constexpr int value = PASS::TCOLS; };

// where the loading step of a launch reads its tile: nullptr = in place; mirrored layout of a.src; one source limb per polynomial
// (src_bcast); or the unit list's own source limb
template <class PASS, int LOGN, bool IS_COL>
FHE_D const u64 *pass_source(const PassArgs &a, const u64 *base, u32 row0)
{
    if (!a.src) return nullptr;
    const u32 unit = blockIdx.x / PASS::TILES, tile = blockIdx.x % PASS::TILES;
    const size_t toff = IS_COL ? (size_t)tile * PASS_TCOLS<PASS, IS_COL>::value : (size_t)row0 * PASS::NPTS;
    if (a.map) {
        const u32 s = a.map[unit].src;
        if (s != 0xFFFFFFFFu) return a.src + ((size_t)s << LOGN) + toff;
    } else if (a.src_bcast) {
        const u32 polys = a.units / a.limbs;
        return a.src + (size_t)(unit % polys) * a.src_bcast + toff;
    }
    if (a.src_stride) {       // the source keeps its polynomials at another distance (compact hand-off scratch of a limb window)
        const u32 polys = a.units / a.limbs;
        return a.src + (((size_t)(unit % polys) * a.src_stride + unit / polys) << LOGN) + toff;
    }
    return a.src + (base - a.data);
}

template <class PASS, int LOGN, bool INV, bool IS_COL>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(PassArgs a)
{
    typedef typename PASS::Arith A;
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS > 0 ? PASS::LDS_ELEMS : 1];
    u32 limb, row0 = 0;
    u64 *base;
    if constexpr (IS_COL) base = col_tile<PASS, LOGN>(blockIdx.x, a, limb);
    else base = row_tile<PASS, LOGN>(blockIdx.x, a, limb, row0);
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(INV ? p.inv : p.fwd);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    const u64 *from = pass_source<PASS, LOGN, IS_COL>(a, base, row0);      // out-of-place: this launch loads from the source buffer
    if constexpr (!IS_COL && INV) {   // (a rotation's automorphism on the opening INTT's load, PassArgs::galois)
        PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n, (NoTap *)nullptr, from, a.galois, a.galois_copy ? a.galois_copy + (base - a.data) : nullptr);
    } else {
        PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n, (NoTap *)nullptr, from);
    }
    if constexpr (PASS::NPHASE > 1) {
        __syncthreads();
        PASS::template phase<1>(tid, base, lds, tw, row0, ctx, inv_n);
    }
    if constexpr (PASS::NPHASE > 2) {
        __syncthreads();
        PASS::template phase<2>(tid, base, lds, tw, row0, ctx, inv_n);
    }
    if constexpr (PASS::NPHASE > 3) {
        __syncthreads();
        PASS::template phase<3>(tid, base, lds, tw, row0, ctx, inv_n);
    }
}

// First launch of the natural-order (cyclic / four-step) transforms: the inverse-structured row pass on network rows
// bitrev(row0 + i), fed from the natural-order source a.src (ntt_core.hpp RowPass ROWMODE 1), results to a.data.
template <class PASS, int LOGN>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_gs_first(PassArgs a)
{
    typedef typename PASS::Arith A;
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS];
    const u32 unit = blockIdx.x / PASS::TILES, tile = blockIdx.x % PASS::TILES;
    const u32 polys = a.units / a.limbs, l = unit / polys, poly = unit % polys;
    const size_t off = ((size_t)poly * a.poly_stride + l) << LOGN;
    const LimbParams &p = a.lp[a.limb0 + l];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.inv);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    const u32 row0 = tile * PASS::TROWS;
    u64 *base = a.data + off;
    const u64 *from = a.src + off;
    PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n, (NoTap *)nullptr, from, 0u, nullptr, a.stream_hint != 0);
    __syncthreads();
    PASS::template phase<1>(tid, base, lds, tw, row0, ctx, inv_n);
    if constexpr (PASS::NPHASE > 2) {
        __syncthreads();
        PASS::template phase<2>(tid, base, lds, tw, row0, ctx, inv_n);
    }
    if constexpr (PASS::NPHASE > 3) {
        __syncthreads();
        PASS::template phase<3>(tid, base, lds, tw, row0, ctx, inv_n);
    }
}

template <class PASS, int LOGN, bool INV, bool IS_COL>
static hipError_t launch_pass(hipStream_t st, const PassArgs &a)
{
    const u32 blocks = a.units * PASS::TILES;
    static const u32 extra_lds = getenv("FHE_DBG_EXTRA_LDS") ? (u32)atoi(getenv("FHE_DBG_EXTRA_LDS")) : 0;     // occupancy experiment: dynamic LDS nobody uses
    hipLaunchKernelGGL((k_ntt_pass<PASS, LOGN, INV, IS_COL>), dim3(blocks), dim3(NTT_THREADS), extra_lds, st, a);
    return hipGetLastError();
}

// Forward transform with the packed hand-off (ntt_core.hpp "Packed hand-off"): column pass -> 50-bit blocks in a.scratch,
// row pass <- those blocks.  The unit number (blockIdx / 16) indexes the scratch and is the same in both launches.
template <class PASS, int LOGN>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_col_packed(PassArgs a)
{
    typedef typename PASS::Arith A;
    static_assert(PASS::PACKABLE, "geometry without a packed form");
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS];
    u32 limb;
    u64 *base = col_tile<PASS, LOGN>(blockIdx.x, a, limb);
    const u32 unit = blockIdx.x / PASS::TILES, tile = blockIdx.x % PASS::TILES;
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.fwd);
    const int tid = threadIdx.x;
    PASS::template phase<0>(tid, base, lds, tw, 0u, ctx, p.inv_n);
    __syncthreads();
    u64 chunk[PK_WORDS];
    PASS::phase_last_packed(tid, lds, tw, 0u, ctx, chunk);
    __syncthreads();                                        // every thread has read its points: the image becomes the staging area
    u64 *stage = reinterpret_cast<u64 *>(lds);
    PASS::pack_stage(tid, stage, chunk);
    __syncthreads();
    PASS::pack_copy_out(tid, stage, a.scratch + ((size_t)unit * 256 + (size_t)tile * 16) * PK_BLOCK_WORDS);
}

template <class PASS, int LOGN>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_row_packed(PassArgs a)
{
    typedef typename PASS::Arith A;
    static_assert(PASS::PACKABLE && PASS::NPHASE == 3, "geometry without a packed form");
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS];
    u32 limb, row0 = 0;
    u64 *base = row_tile<PASS, LOGN>(blockIdx.x, a, limb, row0);
    const u32 unit = blockIdx.x / PASS::TILES, tile = blockIdx.x % PASS::TILES;
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.fwd);
    const int tid = threadIdx.x;
    u64 *stage = reinterpret_cast<u64 *>(lds);
    PASS::unpack_copy_in(tid, stage, a.scratch + (size_t)unit * 256 * PK_BLOCK_WORDS, tile);
    __syncthreads();
    typename PASS::elem x[16];
    PASS::phase_first_packed(tid, stage, tw, row0, ctx, x);
    __syncthreads();                                        // every thread has unpacked: the staging area becomes the image
    PASS::phase_first_store(tid, lds, x);
    __syncthreads();
    PASS::template phase<1>(tid, base, lds, tw, row0, ctx, p.inv_n);
    __syncthreads();
    PASS::template phase<2>(tid, base, lds, tw, row0, ctx, p.inv_n);
}

// sizes / directions / paths that have the packed form
template <class A, int LOGN, bool INV, int GEO> constexpr bool packed_ok()
{
    typedef Passes<A, LOGN, INV, GEO> PS;
    if constexpr (PS::G::TWO_PASS && !INV && A::PATH == PATH_F64) return PS::Col::PACKABLE && PS::Row::PACKABLE;
    return false;
}

// LDS-resident single pass (ntt_plan.hpp ResidentPlan): dynamic LDS above the 64 KiB static limit
template <class PASS, int LOGN, bool INV>
__global__ __launch_bounds__(PASS::THREADS) void k_ntt_resident(PassArgs a)
{
    typedef typename PASS::Arith A;
    extern __shared__ __attribute__((aligned(16))) unsigned char resident_lds[];
    typename PASS::elem *lds = reinterpret_cast<typename PASS::elem *>(resident_lds);
    u32 limb, row0 = 0;
    u64 *base = row_tile<PASS, LOGN>(blockIdx.x, a, limb, row0);
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(INV ? p.inv : p.fwd);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    static_assert(PASS::NPHASE == 4 || PASS::NPHASE == 5, "three register steps and one or two copy phases");
    PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n);
    __syncthreads();
    PASS::template phase<1>(tid, base, lds, tw, row0, ctx, inv_n);
    __syncthreads();
    PASS::template phase<2>(tid, base, lds, tw, row0, ctx, inv_n);
    __syncthreads();
    PASS::template phase<3>(tid, base, lds, tw, row0, ctx, inv_n);
    if constexpr (PASS::NPHASE > 4) {
        __syncthreads();
        PASS::template phase<4>(tid, base, lds, tw, row0, ctx, inv_n);
    }
}

template <class A, int LOGN, bool INV>
static hipError_t launch_resident(hipStream_t st, const PassArgs &a)
{
    typedef typename ResidentPass<A, LOGN, INV>::Pass PASS;
    constexpr size_t bytes = (size_t)PASS::LDS_ELEMS * sizeof(typename PASS::elem);
    static_assert(bytes <= 160 * 1024, "limb does not fit a CU's LDS");
    static bool raised = false;   // per instantiation; the attribute is per function, setting it twice is harmless
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ntt_resident<PASS, LOGN, INV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL((k_ntt_resident<PASS, LOGN, INV>), dim3(a.units * PASS::TILES), dim3(PASS::THREADS), bytes, st, a);
    return hipGetLastError();
}

// which: -1 = whole transform, 0 / 1 = only the first / second launch of a two-pass size (used by
// the fault-injection hook that corrupts the intermediate between the passes)
template <class A, int LOGN, bool INV, int GEO>
static hipError_t launch_transform(hipStream_t st, const PassArgs &a, int which, bool resident)
{
    typedef Passes<A, LOGN, INV, GEO> PS;
    if constexpr (ResidentPlan<LOGN>::OK) {
        if (resident && which == -1) return launch_resident<A, LOGN, INV>(st, a);
    }
    if constexpr (packed_ok<A, LOGN, INV, GEO>()) {
        if (a.scratch && which == -1) {
            hipLaunchKernelGGL((k_ntt_col_packed<typename PS::Col, LOGN>), dim3(a.units * PS::Col::TILES), dim3(NTT_THREADS), 0, st, a);
            hipLaunchKernelGGL((k_ntt_row_packed<typename PS::Row, LOGN>), dim3(a.units * PS::Row::TILES), dim3(NTT_THREADS), 0, st, a);
            return hipGetLastError();
        }
    }
    PassArgs first = a, second = a;            // only the first launch of a transform reads from a.src
    second.src = nullptr;
    second.galois = 0;
    second.galois_copy = nullptr;
    if (a.tmp && which == -1) {                // ping-pong: data (or src) -> tmp -> data, both launches out of place
        if (!first.src) first.src = a.data;
        first.data = a.tmp;
        second.src = a.tmp;
        if (a.tmp_stride) {
            first.src_stride = a.poly_stride;
            first.poly_stride = a.tmp_stride;
            second.src_stride = a.tmp_stride;
        }
    }
    if constexpr (!PS::G::TWO_PASS) {
        if (which == 1) return hipSuccess;
        if (a.stream_hint && !a.src) return launch_pass<typename PS::SingleNt, LOGN, INV, false>(st, a);      // a batch that streams from HBM
        return launch_pass<typename PS::Single, LOGN, INV, false>(st, a);
    } else if (GEO == 1 && a.stream_hint) {
        // pieces of a batch that streams from HBM: non-temporal accesses on the external side (the first launch's loads, the second
        // one's stores), so that the Infinity Cache keeps the hand-off and not words that are touched once (+3-4 % on the 512 MiB
        // batch, -17 % on one that lives in the cache: profiles/r02_nt_sweep.txt -- hence only where the caller says so)
        if constexpr (GEO == 1) {
            hipError_t e = which == 1 ? hipSuccess
                                      : INV ? launch_pass<typename PS::RowNt, LOGN, INV, false>(st, first) : launch_pass<typename PS::ColNt, LOGN, INV, true>(st, first);
            if (e != hipSuccess || which == 0) return e;
            return INV ? launch_pass<typename PS::ColNt, LOGN, INV, true>(st, second) : launch_pass<typename PS::RowNt, LOGN, INV, false>(st, second);
        }
        return hipErrorInvalidValue;
    } else if constexpr (!INV) {
        hipError_t e = which == 1 ? hipSuccess : launch_pass<typename PS::Col, LOGN, INV, true>(st, first);
        if (e != hipSuccess || which == 0) return e;
        return launch_pass<typename PS::Row, LOGN, INV, false>(st, second);
    } else {
        hipError_t e = hipSuccess;
        if (which != 1) {
            bool done = false;
            if constexpr (LOGN == 17 && GEO == 1) {
                // a key switch's INTTs at 2^17 are launches of 500-1000 workgroups whose time is one workgroup's own chain of phases: with
                // 8-row tiles a thread walks every phase twice (512 register sets for 256 threads); 4-row tiles make it once and put twice
                // as many waves on a CU (opening INTT of 32 limbs: 32.5 -> 26.3 us; batches that fill the chip keep the 8-row tile)
                typedef RowPass<A, typename PS::PL::Row, LOGN, 4, NTT_THREADS, true, IO_CANONICAL, IO_LAZY, PS::RED_FIRST, PS::SB> RowSmall;
                if (a.units * PS::Row::TILES >= 768u && a.units * PS::Row::TILES <= 2048u) {     // (576 workgroups: 14.7 us either way, 15.7 with the small tile)
                    e = launch_pass<RowSmall, LOGN, INV, false>(st, first);
                    done = true;
                }
            }
            if (!done) e = launch_pass<typename PS::Row, LOGN, INV, false>(st, first);
        }
        if (e != hipSuccess || which == 0) return e;
        return launch_pass<typename PS::Col, LOGN, INV, true>(st, second);
    }
}

// Natural-order transform (motivation/ntt.py:8-32; the four-step flow of reliability_test/four_step_ntt_prot.py:71-109):
// the inverse-structured network with a cyclic table in the inverse slot maps bit-reversed input to natural output, and the
// bit reversal is folded into the first launch's loads.  a.src = natural-order input, a.data = natural-order output,
// tmp = hand-off buffer of the same layout (two-launch sizes; a.src may equal a.data).
template <class A, int LOGN>
static hipError_t launch_gs_t(hipStream_t st, const PassArgs &a, u64 *tmp)
{
    constexpr int GEO = LOGN >= 13 ? 1 : 0;
    typedef Passes<A, LOGN, true, GEO> PS;
    constexpr bool TWO = PS::G::TWO_PASS;
    typedef RowPass<A, typename PS::PL::Row, LOGN, TWO ? PS::G::TR : 1, NTT_THREADS, true, IO_CANONICAL, TWO ? IO_LAZY : IO_CANONICAL, PS::RED_FIRST,
                    PS::SB, 0, false, false, 1> First;
    if constexpr (!TWO) {
        hipLaunchKernelGGL((k_ntt_gs_first<First, LOGN>), dim3(a.units * First::TILES), dim3(NTT_THREADS), 0, st, a);
        return hipGetLastError();
    } else {
        if (!tmp) return hipErrorInvalidValue;
        PassArgs first = a, second = a;
        first.data = tmp;
        second.src = tmp;
        hipLaunchKernelGGL((k_ntt_gs_first<First, LOGN>), dim3(a.units * First::TILES), dim3(NTT_THREADS), 0, st, first);
        if (a.stream_hint && GEO == 1) {
            if constexpr (GEO == 1) return launch_pass<typename PS::ColNt, LOGN, true, true>(st, second);
        }
        return launch_pass<typename PS::Col, LOGN, true, true>(st, second);
    }
}

bool ntt_gs_supported(int logn) { return logn >= 5 && logn <= NTT_MAX_LOGN; }

hipError_t launch_ntt_gs(hipStream_t st, const PassArgs &a, u64 *tmp, int logn, int path)
{
    if (a.units == 0) return hipSuccess;
    if (!a.src || a.map || !ntt_gs_supported(logn)) return hipErrorInvalidValue;
    switch (logn) {
#define FHE_CASE(L) \
    case L: return path == PATH_F64 ? launch_gs_t<ArithF64, L>(st, a, tmp) : launch_gs_t<ArithU64, L>(st, a, tmp);
        FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13)
        FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default: return hipErrorInvalidValue;
    }
}

template <class A>
static hipError_t launch_size(hipStream_t st, const PassArgs &a, int logn, bool inverse, int geo, int which, bool resident)
{
    if (logn == 16 && geo == 0)   // tuning: the wide-tile geometry is kept for 2^16 only
        return inverse ? launch_transform<A, 16, true, 0>(st, a, which, false) : launch_transform<A, 16, false, 0>(st, a, which, false);
    switch (logn) {
#define FHE_CASE(L)                                                        \
    case L:                                                                \
        return inverse ? launch_transform<A, L, true, (L >= 13 ? 1 : 0)>(st, a, which, resident) : launch_transform<A, L, false, (L >= 13 ? 1 : 0)>(st, a, which, resident);
        FHE_CASE(1) FHE_CASE(2) FHE_CASE(3) FHE_CASE(4) FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8)
        FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13) FHE_CASE(14) FHE_CASE(15) FHE_CASE(16)
        FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default:
        return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// Checked forward transform (ABFT, rfhe_framewk/src/negaclic_ntt.py:130-149): the weighted checksums ride
// on the passes.  The pass that reads the input accumulates sum_i w_i x_i from the registers it has just
// loaded, the pass that writes the result accumulates sum_j w^_j X_j from the words it is about to store;
// each workgroup stores its canonical partial sum in the unit's row of a [units][tiles] array, and the
// comparison kernel adds the rows up modulo q (no atomics: deterministic, and no 64-bit wrap-around with 61-bit primes).
// Weights arrive encoded like twiddles (Tw per element, arithmetic path of the limb).
// ---------------------------------------------------------------------------
struct AbftArgs {
    const Tw *win;      // [table limbs][N]  input-side weights, twiddle encoding (read by ArithU64 limbs)
    const Tw *wout;     // [table limbs][N]  output-side weights (N^-1 folded in), twiddle encoding (ArithU64 limbs)
    const u64 *wout8;   // [table limbs][N]  output-side weights as residues (ArithF64 limbs)
    int logp;           // log2 of the weight pattern's period (logn / 2)
    u64 *sum_in;        // [units][tiles of the pass that reads the input], units in [poly][limb] order
    u64 *sum_out;       // [units][tiles of the pass that writes the result]
};

template <class A, bool IN, bool OUT>
struct ChecksumTap {
    static constexpr bool ACTIVE = true;
    static constexpr bool MID = false;
    static constexpr bool STORES = false;
    typedef typename A::elem elem;
    TwPtr win, wout;                   // ArithU64: Shoup-encoded weights of this limb, offset to the tile's first element
    const u64 FHE_GLOBAL *wout8;       // ArithF64: output-side weights as plain residues (the quotient factor is one multiply)
    u32 pos0;                          // index of the tile's first element inside its limb
    int logp;                          // input-side weight of element i = (i mod 2^logp + 1) + (i div 2^logp + 1)
    elem acc_in, acc_out;
    int n_in, n_out;
    FHE_D void in(u32 idx, elem x, const typename A::Ctx &c)
    {
        if constexpr (IN) {
            if constexpr (A::PATH == PATH_F64) {
                // generate_weights (negaclic_ntt.py:7-13) computed in place of a table read: small integers
                const u32 i = pos0 + idx;
                // (the small integer becomes a double through the exponent trick of from_canonical: one subtraction, no v_cvt_f64_u32)
                const double w = ArithF64::from_canonical((u64)((i & ((1u << logp) - 1u)) + (i >> logp) + 2u));
                A::lazy_acc(acc_in, A::mulmod_w(x, w, w * c.ninv, c), ++n_in, c);
            } else {
                A::lazy_acc(acc_in, A::mulmod(x, win[idx], c), ++n_in, c);
            }
        }
    }
    FHE_D void out(u32 idx, u64 v, const typename A::Ctx &c)
    {
        if constexpr (OUT) {
            if constexpr (A::PATH == PATH_F64) {
                const double w = A::from_canonical(wout8[idx]);
                A::lazy_acc(acc_out, A::mulmod_w(A::from_canonical(v), w, w * c.ninv, c), ++n_out, c);
            } else {
                A::lazy_acc(acc_out, A::mulmod(A::from_canonical(v), wout[idx], c), ++n_out, c);
            }
        }
    }
};

// Per-phase detector (the reference checks its four-step flow phase by phase: batch_check of the column transforms,
// check_inter around the twiddle step, batch_check of the row transforms -- rfhe_framewk/src/ntt_test/relia_ntt_sim.cpp:235-292,
// 331-355; reliability_test/four_step_ntt_prot.py:185-194).  The engine's two launches ARE that flow -- column transforms,
// then row transforms with the twiddle folded into their butterflies -- so the checks sit at the same three places:
//   column pass :  sum_i w_i x_i  (words it loads)        ==  sum_i u_i y_i  (words it stores),   u = P1^-T w
//   hand-off    :  sum_i u_i y_i  (as stored)             ==  sum_i u_i y_i  (as loaded by the row pass)
//   row pass    :  sum_i u_i y_i  (words it loads)        ==  sum_j w^_j X_j (words it stores),   w^ = T^-T w
// PASS 0 = column pass (in: w, mid: u), PASS 1 = row pass (in: u, out: w^).
template <class A, int PASS>
struct PhaseTap {
    static constexpr bool ACTIVE = true;
    static constexpr bool MID = PASS == 0;
    static constexpr bool STORES = false;
    typedef typename A::elem elem;
    TwPtr win, umid, wout;             // ArithU64: Shoup-encoded weights of this limb, offset to the tile's first element
    const u64 FHE_GLOBAL *umid8, *wout8;   // ArithF64: the same weights as plain residues
    u32 pos0;
    int logp;
    elem acc_a, acc_b;
    int n_a, n_b;
    FHE_D void weigh(elem &acc, int &n, elem x, u32 idx, TwPtr tw, const u64 FHE_GLOBAL *tw8, const typename A::Ctx &c)
    {
        if constexpr (A::PATH == PATH_F64) {
            const double w = A::from_canonical(tw8[idx]);
            A::lazy_acc(acc, A::mulmod_w(x, w, w * c.ninv, c), ++n, c);
        } else {
            A::lazy_acc(acc, A::mulmod(x, tw[idx], c), ++n, c);
        }
    }
    FHE_D void in(u32 idx, elem x, const typename A::Ctx &c)
    {
        if constexpr (PASS == 0) {
            if constexpr (A::PATH == PATH_F64) {
                const u32 i = pos0 + idx;
                const double w = ArithF64::from_canonical((u64)((i & ((1u << logp) - 1u)) + (i >> logp) + 2u));     // generate_weights, negaclic_ntt.py:7-13
                A::lazy_acc(acc_a, A::mulmod_w(x, w, w * c.ninv, c), ++n_a, c);
            } else {
                A::lazy_acc(acc_a, A::mulmod(x, win[idx], c), ++n_a, c);
            }
        } else {
            weigh(acc_a, n_a, x, idx, umid, umid8, c);
        }
    }
    FHE_D void mid(u32 idx, elem x, const typename A::Ctx &c) { weigh(acc_b, n_b, x, idx, umid, umid8, c); }
    FHE_D void out(u32 idx, u64 v, const typename A::Ctx &c) { weigh(acc_b, n_b, A::from_canonical(v), idx, wout, wout8, c); }
};

// modular sum of one canonical value per thread over the workgroup, stored to *dst by one lane
__device__ __forceinline__ void block_sum_mod(u64 v, u64 q, u64 *dst, u64 *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v += __shfl_down(v, off, 64);
        v = v >= q ? v - q : v;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 s = 0;
        for (int w = 0; w < NTT_THREADS / 64; w++) {
            s += red[w];
            s = s >= q ? s - q : s;
        }
        *dst = s;
    }
}

template <class PASS, int LOGN, bool IS_COL, bool IN, bool OUT>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass_abft(PassArgs a, AbftArgs ab)
{
    typedef typename PASS::Arith A;
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS > 0 ? PASS::LDS_ELEMS : 1];
    __shared__ u64 red[2][NTT_THREADS / 64];
    u32 limb, row0 = 0;
    u64 *base;
    if constexpr (IS_COL) base = col_tile<PASS, LOGN>(blockIdx.x, a, limb);
    else base = row_tile<PASS, LOGN>(blockIdx.x, a, limb, row0);
    // position of the tile inside its limb and index of the unit in [poly][limb] order
    const u32 unit = blockIdx.x / PASS::TILES, polys = a.units / a.limbs;
    const u32 l = unit / polys, poly = unit % polys, slot = poly * a.poly_stride + l;
    const u32 pos0 = (u32)((base - a.data) & (((size_t)1 << LOGN) - 1));
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.fwd);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    ChecksumTap<A, IN, OUT> tap{as_global(ab.win) + ((size_t)limb << LOGN) + pos0, as_global(ab.wout) + ((size_t)limb << LOGN) + pos0,
                                (const u64 FHE_GLOBAL *)ab.wout8 + ((size_t)limb << LOGN) + pos0, pos0, ab.logp,
                                typename A::elem(0), typename A::elem(0), 0, 0};
    const u64 *from = pass_source<PASS, LOGN, IS_COL>(a, base, row0);      // ping-pong hand-off: this launch loads from a.src
    PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n, &tap, from);
    if constexpr (PASS::NPHASE > 1) {
        __syncthreads();
        PASS::template phase<1>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    if constexpr (PASS::NPHASE > 2) {
        __syncthreads();
        PASS::template phase<2>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    if constexpr (PASS::NPHASE > 3) {
        __syncthreads();
        PASS::template phase<3>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    const u32 tile = blockIdx.x % PASS::TILES;
    if constexpr (IN) block_sum_mod(A::canonical(tap.acc_in, ctx), p.q, ab.sum_in + (size_t)slot * PASS::TILES + tile, red[0]);
    if constexpr (OUT) block_sum_mod(A::canonical(tap.acc_out, ctx), p.q, ab.sum_out + (size_t)slot * PASS::TILES + tile, red[1]);
}

template <class A, int LOGN>
static hipError_t launch_checked(hipStream_t st, const PassArgs &a, const AbftArgs &ab, int which)
{
    constexpr int GEO = LOGN >= 13 ? 1 : 0;
    typedef Passes<A, LOGN, false, GEO> PS;
    if constexpr (!PS::G::TWO_PASS) {
        if (which == 1) return hipSuccess;
        hipLaunchKernelGGL((k_ntt_pass_abft<typename PS::Single, LOGN, false, true, true>), dim3(a.units * PS::Single::TILES), dim3(NTT_THREADS), 0, st, a, ab);
    } else {
        PassArgs first = a, second = a;
        if (a.tmp && which == -1) {                // ping-pong: data -> tmp -> data, both launches out of place (launch_transform)
            first.src = a.data;
            first.data = a.tmp;
            second.src = a.tmp;
        }
        if (a.stream_hint && GEO == 1) {      // pieces of a batch that streams from HBM: non-temporal accesses on the external side (launch_transform)
            if (which != 1) hipLaunchKernelGGL((k_ntt_pass_abft<typename PS::ColNt, LOGN, true, true, false>), dim3(a.units * PS::Col::TILES), dim3(NTT_THREADS), 0, st, first, ab);
            if (which != 0) hipLaunchKernelGGL((k_ntt_pass_abft<typename PS::RowNt, LOGN, false, false, true>), dim3(a.units * PS::Row::TILES), dim3(NTT_THREADS), 0, st, second, ab);
            return hipGetLastError();
        }
        if (which != 1) hipLaunchKernelGGL((k_ntt_pass_abft<typename PS::Col, LOGN, true, true, false>), dim3(a.units * PS::Col::TILES), dim3(NTT_THREADS), 0, st, first, ab);
        if (which != 0) hipLaunchKernelGGL((k_ntt_pass_abft<typename PS::Row, LOGN, false, false, true>), dim3(a.units * PS::Row::TILES), dim3(NTT_THREADS), 0, st, second, ab);
    }
    return hipGetLastError();
}

// ---- per-phase variant (two-launch sizes) ----
template <class PASS, int LOGN, bool IS_COL, int PASSID>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass_phase(PassArgs a, PhaseArgs ph)
{
    typedef typename PASS::Arith A;
    static_assert(PASS::NPHASE >= 2, "per-phase checks are for the two-launch sizes");
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS];
    __shared__ u64 red[2][NTT_THREADS / 64];
    u32 limb, row0 = 0;
    u64 *base;
    if constexpr (IS_COL) base = col_tile<PASS, LOGN>(blockIdx.x, a, limb);
    else base = row_tile<PASS, LOGN>(blockIdx.x, a, limb, row0);
    const u32 unit = blockIdx.x / PASS::TILES, polys = a.units / a.limbs;
    const u32 l = unit / polys, poly = unit % polys, slot = poly * a.poly_stride + l;
    const u32 pos0 = (u32)((base - a.data) & (((size_t)1 << LOGN) - 1));
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.fwd);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    const size_t woff = ((size_t)limb << LOGN) + pos0;
    PhaseTap<A, PASSID> tap{as_global(ph.win) + woff, as_global(ph.umid) + woff, as_global(ph.wout) + woff, (const u64 FHE_GLOBAL *)ph.umid8 + woff,
                            (const u64 FHE_GLOBAL *)ph.wout8 + woff, pos0, ph.logp, typename A::elem(0), typename A::elem(0), 0, 0};
    const u64 *from = pass_source<PASS, LOGN, IS_COL>(a, base, row0);
    PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n, &tap, from);
    __syncthreads();
    // test hook: a soft error INSIDE this pass -- one bit of one word of the LDS image between two register steps
    if (ph.fault_pass == PASSID && ph.fault_block == blockIdx.x) {
        if (tid == 0) reinterpret_cast<u64 *>(lds)[ph.fault_word % (u32)PASS::LDS_ELEMS] ^= (u64)1 << ph.fault_bit;
        __syncthreads();
    }
    PASS::template phase<1>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    if constexpr (PASS::NPHASE > 2) {
        __syncthreads();
        PASS::template phase<2>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    if constexpr (PASS::NPHASE > 3) {
        __syncthreads();
        PASS::template phase<3>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    const u32 tile = blockIdx.x % PASS::TILES;
    block_sum_mod(A::canonical(tap.acc_a, ctx), p.q, ph.sum_a + (size_t)slot * PASS::TILES + tile, red[0]);
    block_sum_mod(A::canonical(tap.acc_b, ctx), p.q, ph.sum_b + (size_t)slot * PASS::TILES + tile, red[1]);
}

template <class A, int LOGN>
static hipError_t launch_phases_t(hipStream_t st, const PassArgs &a, const PhaseArgs &p1, const PhaseArgs &p2, int which)
{
    typedef Passes<A, LOGN, false, 1> PS;
    if constexpr (PS::G::TWO_PASS) {
        PassArgs first = a, second = a;
        if (a.tmp && which == -1) {                // ping-pong hand-off (launch_transform): the hand-off check then covers the scratch
            first.src = a.data;
            first.data = a.tmp;
            second.src = a.tmp;
        }
        if (a.stream_hint) {
            if (which != 1) hipLaunchKernelGGL((k_ntt_pass_phase<typename PS::ColNt, LOGN, true, 0>), dim3(a.units * PS::Col::TILES), dim3(NTT_THREADS), 0, st, first, p1);
            if (which != 0) hipLaunchKernelGGL((k_ntt_pass_phase<typename PS::RowNt, LOGN, false, 1>), dim3(a.units * PS::Row::TILES), dim3(NTT_THREADS), 0, st, second, p2);
            return hipGetLastError();
        }
        if (which != 1) hipLaunchKernelGGL((k_ntt_pass_phase<typename PS::Col, LOGN, true, 0>), dim3(a.units * PS::Col::TILES), dim3(NTT_THREADS), 0, st, first, p1);
        if (which != 0) hipLaunchKernelGGL((k_ntt_pass_phase<typename PS::Row, LOGN, false, 1>), dim3(a.units * PS::Row::TILES), dim3(NTT_THREADS), 0, st, second, p2);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

bool ntt_phases_supported(int logn) { return logn >= 13 && logn <= NTT_MAX_LOGN; }

int ntt_column_stages(int logn)
{
    switch (logn) {
#define FHE_CASE(L) case L: return Plan<L>::Col::P;
        FHE_CASE(1) FHE_CASE(2) FHE_CASE(3) FHE_CASE(4) FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10)
        FHE_CASE(11) FHE_CASE(12) FHE_CASE(13) FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default: return 0;
    }
}

hipError_t launch_ntt_phases(hipStream_t st, const PassArgs &a, const PhaseArgs &p1, const PhaseArgs &p2, int logn, int path, int which)
{
    if (a.units == 0) return hipSuccess;
    if (a.map || !ntt_phases_supported(logn)) return hipErrorInvalidValue;
    switch (logn) {
#define FHE_CASE(L) \
    case L: return path == PATH_F64 ? launch_phases_t<ArithF64, L>(st, a, p1, p2, which) : launch_phases_t<ArithU64, L>(st, a, p1, p2, which);
        FHE_CASE(13) FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default: return hipErrorInvalidValue;
    }
}

bool ntt_checked_supported(int logn) { return logn >= 5 && logn <= NTT_MAX_LOGN; }

template <int LOGN> static void checked_tiles_t(u32 *tin, u32 *tout)
{
    typedef Passes<ArithF64, LOGN, false, (LOGN >= 13 ? 1 : 0)> PS;
    if constexpr (!PS::G::TWO_PASS) *tin = *tout = PS::Single::TILES;
    else {
        *tin = PS::Col::TILES;
        *tout = PS::Row::TILES;
    }
}
// partial sums per unit on the input / output side (row lengths of sum_in / sum_out)
void ntt_checked_tiles(int logn, u32 *tin, u32 *tout)
{
    *tin = *tout = 1;
    switch (logn) {
#define FHE_CASE(L) case L: checked_tiles_t<L>(tin, tout); break;
        FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13)
        FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default: break;
    }
}

// forward transform with the checksums fused in; which as in launch_ntt (the fault hook splits the launches)
hipError_t launch_ntt_checked(hipStream_t st, const PassArgs &a, const Tw *win, const Tw *wout, const u64 *wout8, u64 *sum_in, u64 *sum_out,
                              int logn, int path, int which)
{
    if (a.units == 0) return hipSuccess;
    if (a.map) return hipErrorInvalidValue;
    const AbftArgs ab{win, wout, wout8, logn / 2, sum_in, sum_out};
    switch (logn) {
#define FHE_CASE(L) \
    case L: return path == PATH_F64 ? launch_checked<ArithF64, L>(st, a, ab, which) : launch_checked<ArithU64, L>(st, a, ab, which);
        FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13)
        FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// Negacyclic product, middle launch: for one row tile, finish the forward transform of a and of b
// (row pass), multiply, and run the first inverse pass (the row pass again), reading each input tile
// once and writing the product tile once.  The forward row pass leaves final words in the LDS image
// where its copy-out phase would read them, and the inverse row pass expects raw words where its
// copy-in phase would put them: same image, so a's tile goes to registers (one pair per lane and
// 2*NTT_THREADS words), b's tile is multiplied in place in LDS, and the inverse steps carry on.
// For single-pass sizes this is the whole product in one launch.
// ---------------------------------------------------------------------------
struct PolymulArgs {
    PassArgs a;     // a.data = first factor (tile mapping as for a row pass)
    const u64 *b;   // second factor, same layout
    u64 *c;         // product (may alias either factor)
};

template <class FR, int E = 0, class TWS = TwPtr>
FHE_D void fwd_steps(int tid, u64 *base, typename FR::elem *lds, const TWS &tw, u32 row0, const typename FR::Arith::Ctx &ctx, const Tw &inv_n)
{
    if constexpr (E < FR::NSTEP) {
        if (E > 0) __syncthreads();
        FR::template phase<E, NoTap, TWS>(tid, base, lds, tw, row0, ctx, inv_n);
        fwd_steps<FR, E + 1, TWS>(tid, base, lds, tw, row0, ctx, inv_n);
    }
}
template <class IR, int E = 1>
FHE_D void inv_steps(int tid, u64 *base, typename IR::elem *lds, TwPtr tw, u32 row0, const typename IR::Arith::Ctx &ctx, const Tw &inv_n)
{
    if constexpr (E < IR::NPHASE) {
        __syncthreads();
        IR::template phase<E>(tid, base, lds, tw, row0, ctx, inv_n);
        inv_steps<IR, E + 1>(tid, base, lds, tw, row0, ctx, inv_n);
    }
}

// the row passes of the middle launch: the forward one leaves its results in the LDS image in the arithmetic's LAZY form
// and the inverse one takes them from there in that form, so the product never goes through canonical words
template <class A, int LOGN, int GEO>
struct MidPasses {
    typedef Passes<A, LOGN, false, GEO> F;
    typedef Passes<A, LOGN, true, GEO> I;
    typedef typename F::PL PL;
    static constexpr bool TWO = F::G::TWO_PASS;
    static constexpr int TR = TWO ? F::G::TR : 1;
    static constexpr int SB = BlkStage<LOGN>::value;
    typedef RowPass<A, typename PL::Row, LOGN, TR, NTT_THREADS, false, TWO ? IO_LAZY : IO_CANONICAL, IO_LAZY, TWO ? F::RED_SECOND : F::RED_FIRST, SB> Fwd;
    typedef RowPass<A, typename PL::Row, LOGN, TR, NTT_THREADS, true, IO_LAZY, TWO ? IO_LAZY : IO_CANONICAL, I::RED_FIRST, SB> Inv;
};

template <class A, int LOGN, int GEO>
__global__ __launch_bounds__(NTT_THREADS) void k_polymul_mid(PolymulArgs pa)
{
    typedef typename MidPasses<A, LOGN, GEO>::Fwd FR;
    typedef typename MidPasses<A, LOGN, GEO>::Inv IR;
    static_assert(FR::STAGED && IR::STAGED && FR::LDS_ELEMS == IR::LDS_ELEMS, "fused product needs the staged row pass");
    typedef typename FR::elem elem;
    constexpr int PAIRS = FR::TROWS * FR::NPTS / 2;
    constexpr int PER = (PAIRS + NTT_THREADS - 1) / NTT_THREADS;
    __shared__ __attribute__((aligned(16))) elem lds[FR::LDS_ELEMS];
    u32 limb, row0;
    u64 *ta = row_tile<FR, LOGN>(blockIdx.x, pa.a, limb, row0);
    const size_t off = (size_t)(ta - pa.a.data);
    const LimbParams &p = pa.a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;

    fwd_steps<FR>(tid, ta, lds, as_global(p.fwd), row0, ctx, inv_n);
    __syncthreads();
    elem ra[PER][2];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * NTT_THREADS;
        if (PAIRS % NTT_THREADS == 0 || i < PAIRS) {
            const u32 row = (u32)i / (FR::NPTS / 2), g = ((u32)i % (FR::NPTS / 2)) * 2;
            const elem *src = lds + row * FR::ROW_LDS + row_pad(g);
            ra[k][0] = src[0];
            ra[k][1] = src[1];
        }
    }
    __syncthreads();
    fwd_steps<FR>(tid, const_cast<u64 *>(pa.b) + off, lds, as_global(p.fwd), row0, ctx, inv_n);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * NTT_THREADS;
        if (PAIRS % NTT_THREADS == 0 || i < PAIRS) {
            const u32 row = (u32)i / (FR::NPTS / 2), g = ((u32)i % (FR::NPTS / 2)) * 2;
            elem *dst = lds + row * FR::ROW_LDS + row_pad(g);
            if constexpr (A::PATH == PATH_F64) {
                dst[0] = A::mulvar_lazy(ra[k][0], dst[0], ctx);
                dst[1] = A::mulvar_lazy(ra[k][1], dst[1], ctx);
            } else {
                dst[0] = A::mulvar_lazy(ra[k][0], dst[0], p);
                dst[1] = A::mulvar_lazy(ra[k][1], dst[1], p);
            }
        }
    }
    inv_steps<IR>(tid, pa.c + off, lds, as_global(p.inv), row0, ctx, inv_n);
}

template <class A, int LOGN>
static hipError_t launch_mid(hipStream_t st, const PolymulArgs &pa)
{
    constexpr int GEO = LOGN >= 13 ? 1 : 0;
    typedef typename MidPasses<A, LOGN, GEO>::Fwd FR;
    hipLaunchKernelGGL((k_polymul_mid<A, LOGN, GEO>), dim3(pa.a.units * FR::TILES), dim3(NTT_THREADS), 0, st, pa);
    return hipGetLastError();
}

bool polymul_fused_supported(int logn) { return logn >= 5 && logn <= NTT_MAX_LOGN; }

// c = a * b in Z_q[x]/(x^N + 1) for a.units limb-polynomials: forward column passes (two-pass sizes), the
// middle launch above, inverse column pass.  a and b are scratch afterwards; c may alias either.
hipError_t launch_polymul(hipStream_t st, const PassArgs &a, u64 *b, u64 *c, int logn, int path)
{
    if (a.units == 0) return hipSuccess;
    if (!polymul_fused_supported(logn)) return hipErrorInvalidValue;
    hipError_t e;
    PassArgs pb = a, pc = a;
    pb.data = b;
    pc.data = c;
    if (logn >= 13) {
        if ((e = launch_ntt(st, a, logn, false, path, 1, 0)) != hipSuccess) return e;
        if (b != a.data && (e = launch_ntt(st, pb, logn, false, path, 1, 0)) != hipSuccess) return e;
    }
    const PolymulArgs pa{a, b, c};
    switch (logn) {
#define FHE_CASE(L)                                                                                     \
    case L:                                                                                             \
        e = path == PATH_F64 ? launch_mid<ArithF64, L>(st, pa) : launch_mid<ArithU64, L>(st, pa);       \
        break;
        FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13)
        FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default:
        return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    if (logn >= 13) return launch_ntt(st, pc, logn, true, path, 1, 1);
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// Forward transform whose last pass ends in the mod-down / rescale tail instead of a plain store:
//   out_h[l] = (a_h[l] - NTT(x_h[l])) * scal[l] (+ add_h[l])  mod q_l
// (k_sub_scale's arithmetic, aux_kernels.hip, applied to the words the row pass is about to write: the transformed limbs
// are never stored and re-read, and the separate launch disappears).  Units: [part h][limb l] of `a.data`.
// ---------------------------------------------------------------------------
template <class A>
struct SubScaleTap {
    static constexpr bool ACTIVE = true;
    static constexpr bool MID = false;
    static constexpr bool STORES = true;
    const u64 *acc, *add;       // offset to the tile's first element; add may be null
    u64 *out;
    u64 scal, q, r0, r1, pre;   // pre (0 = none): X is multiplied by it first (the BGV forms' factor t)
    double n, ninv;             // ArithF64 limbs: the tail in exact FP64 arithmetic (about 15 operations per word instead of ~80 integer ones)
    const u64 *add_limb;        // galois != 0: the addend is sigma_k(add) -- read through the Galois map from the limb's first word
    u32 pos0, galois;           // (index of the tile's first word inside its limb; a rotation's sigma(c0), RowEpiArgs::galois)
    int logn;
    const u64 *acc_limb;        // != nullptr: `acc` too is read through the Galois map, from this limb base (RowEpiArgs::galois_a)
    u64 scal2;                  // 0 = none: the finished word is multiplied by it (RowEpiArgs::scal2 -- the rescale after a mod-down)
    template <class E, class C> FHE_D void in(u32, E, const C &) {}
    // integer form (ArithU64 limbs; any 64-bit words)
    FHE_D u64 one_int(u64 a, u64 x, u64 t, bool has_add) const
    {
        const u64 av = a < q ? a : barrett128(a, 0, q, r0, r1);
        if (pre) x = barrett128(x * pre, mulhi64(x, pre), q, r0, r1);
        const u64 d = av >= x ? av - x : av + q - x;
        u64 v = barrett128(d * scal, mulhi64(d, scal), q, r0, r1);
        if (has_add) {
            v += t < q ? t : barrett128(t, 0, q, r0, r1);
            v = v >= q ? v - q : v;
        }
        if (scal2) v = barrett128(v * scal2, mulhi64(v, scal2), q, r0, r1);
        return v;
    }
    // FP64 form: every word canonical (x is the pass's own output; a and t are checked by the caller)
    FHE_D u64 one_f64(u64 a, u64 x, u64 t, bool has_add) const
    {
        const typename ArithF64::Ctx c{n, ninv, q};
        double xv = ArithF64::from_canonical(x);
        if (pre) {
            const double pw = ArithF64::from_canonical(pre);          // residues below 2^50: exact, and no conversion instruction
            xv = ArithF64::mulmod_w(xv, pw, pw * ninv, c);              // |.| < 0.9 q
        }
        const double d = ArithF64::from_canonical(a) - xv;             // |d| < 1.9 q
        const double sw = ArithF64::from_canonical(scal);
        double v = ArithF64::mulmod_w(d, sw, sw * ninv, c);            // |v| < 0.9 q
        if (has_add) v += ArithF64::from_canonical(t);
        if (scal2) {
            const double s2 = ArithF64::from_canonical(scal2);
            v = ArithF64::mulmod_w(v, s2, s2 * ninv, c);               // |v| < 1.9 q going in
        }
        return ArithF64::canonical(v, c);
    }
    FHE_D void store(u32 idx, u64 x0, u64 x1)
    {
        ulonglong2 a;
        if (acc_limb) a = ulonglong2{acc_limb[galois_slot(pos0 + idx, logn, galois)], acc_limb[galois_slot(pos0 + idx + 1, logn, galois)]};
        else a = *reinterpret_cast<const ulonglong2 *>(acc + idx);
        ulonglong2 t = ulonglong2{0, 0};
        if (add) {
            if (galois) t = ulonglong2{add_limb[galois_slot(pos0 + idx, logn, galois)], add_limb[galois_slot(pos0 + idx + 1, logn, galois)]};
            else t = *reinterpret_cast<const ulonglong2 *>(add + idx);
        }
        ulonglong2 r;
        bool fp = A::PATH == PATH_F64;
        if (A::PATH == PATH_F64) fp = !__builtin_expect((a.x >= q) | (a.y >= q) | (t.x >= q) | (t.y >= q), 0);     // out-of-range words: integer form
        if (fp) {
            r.x = one_f64(a.x, x0, t.x, add != nullptr);
            r.y = one_f64(a.y, x1, t.y, add != nullptr);
        } else {
            r.x = one_int(a.x, x0, t.x, add != nullptr);
            r.y = one_int(a.y, x1, t.y, add != nullptr);
        }
        *reinterpret_cast<ulonglong2 *>(out + idx) = r;
    }
};

// Column pass of that transform with a second input (ColAddSrc): every word it loads becomes x + w_l * y (mod q_l), y read from the
// part's ONE extra limb at the same position.  The sum is folded back right away, so the pass's lazy-range schedule sees an input no
// larger than a canonical one.
template <class A>
struct AddSrcTap {
    static constexpr bool ACTIVE = true;
    static constexpr bool MID = false;
    static constexpr bool STORES = false;
    static constexpr bool PRELOAD = true;
    const u64 *y;               // the tile's first word inside the part's extra limb
    Tw w;
    FHE_D u64 load(u32 idx) const { return y[idx]; }
    // one cold branch for the whole register set (FP64 limbs under an extra limb of the integer path: words up to 2^61)
    template <int R> FHE_D void prepare(u64 (&e)[R], const typename A::Ctx &c) const
    {
        if constexpr (A::PATH == PATH_F64) {
            bool big = false;
#pragma unroll
            for (int r = 0; r < R; r++) big |= (e[r] >> 52) != 0;
            if (__builtin_expect(big, 0)) {
#pragma unroll
                for (int r = 0; r < R; r++) e[r] = (e[r] >> 52) ? reduce_any_u64(e[r], c.q) : e[r];
            }
        }
    }
    FHE_D void apply(typename A::elem &x, u64 v, const typename A::Ctx &c) const
    {
        if constexpr (A::PATH == PATH_F64) {
            x += A::mulmod(A::from_canonical(v), w, c);                               // v < 2^52: exact; |.| < 1.6 q
            A::reduce(x, c);
        } else {
            x += A::mulmod(v, w, c);                                                   // [0, 3q): inside the forward butterflies' [0, 4q)
        }
    }
};

template <class PASS, int LOGN>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_col_addsrc(PassArgs a, ColAddSrc s)
{
    typedef typename PASS::Arith A;
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS > 0 ? PASS::LDS_ELEMS : 1];
    u32 limb;
    u64 *base = col_tile<PASS, LOGN>(blockIdx.x, a, limb);
    const u32 unit = blockIdx.x / PASS::TILES, tile = blockIdx.x % PASS::TILES, polys = a.units / a.limbs;
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.fwd);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    AddSrcTap<A> tap{s.y + (size_t)(unit % polys) * s.y_stride + (size_t)tile * PASS::TCOLS, s.w[limb]};
    const u64 *from = pass_source<PASS, LOGN, true>(a, base, 0u);
    PASS::template phase<0>(tid, base, lds, tw, 0u, ctx, inv_n, &tap, from);
    if constexpr (PASS::NPHASE > 1) {
        __syncthreads();
        PASS::template phase<1>(tid, base, lds, tw, 0u, ctx, inv_n);
    }
    if constexpr (PASS::NPHASE > 2) {
        __syncthreads();
        PASS::template phase<2>(tid, base, lds, tw, 0u, ctx, inv_n);
    }
}

template <class PASS, int LOGN>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_row_subscale(PassArgs a, RowEpiArgs ep)
{
    typedef typename PASS::Arith A;
    static_assert(PASS::STAGE_OUT, "the epilogue sits in the staged copy-out phase");
    __shared__ __attribute__((aligned(16))) typename PASS::elem lds[PASS::LDS_ELEMS];
    u32 limb, row0 = 0;
    u64 *base = row_tile<PASS, LOGN>(blockIdx.x, a, limb, row0);
    const u32 unit = blockIdx.x / PASS::TILES, polys = a.units / a.limbs, l = unit / polys, h = unit % polys;
    const LimbParams &p = a.lp[limb];
    const typename A::Ctx ctx = A::make_ctx(p);
    const TwPtr tw = as_global(p.fwd);
    const Tw inv_n = p.inv_n;
    const int tid = threadIdx.x;
    const size_t eoff = ((size_t)l << LOGN) + (size_t)row0 * PASS::NPTS;      // element offset inside part h
    SubScaleTap<A> tap{ep.a + (size_t)h * ep.a_stride + eoff, ep.add[h] ? ep.add[h] + eoff : nullptr, ep.out[h] + eoff, ep.scal[l], p.q, p.barrett_lo, p.barrett_hi,
                       ep.pre ? ep.pre[l] : 0, p.n, p.ninv, ep.add[h] ? ep.add[h] + ((size_t)l << LOGN) : nullptr, row0 * (u32)PASS::NPTS, ep.galois, LOGN,
                       ep.galois && ep.galois_a ? ep.a + (size_t)h * ep.a_stride + ((size_t)l << LOGN) : nullptr, ep.scal2 ? ep.scal2[l] : 0};
    const u64 *from = pass_source<PASS, LOGN, false>(a, base, row0);
    PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n, &tap, from);
    if constexpr (PASS::NPHASE > 1) {
        __syncthreads();
        PASS::template phase<1>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    if constexpr (PASS::NPHASE > 2) {
        __syncthreads();
        PASS::template phase<2>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
    if constexpr (PASS::NPHASE > 3) {
        __syncthreads();
        PASS::template phase<3>(tid, base, lds, tw, row0, ctx, inv_n, &tap);
    }
}

template <class A, int LOGN>
static hipError_t launch_subscale_t(hipStream_t st, const PassArgs &a, const RowEpiArgs &ep, const ColAddSrc *add_src)
{
    constexpr int GEO = LOGN >= 13 ? 1 : 0;
    typedef Passes<A, LOGN, false, GEO> PS;
    if constexpr (!PS::G::TWO_PASS) {
        if (add_src) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_ntt_row_subscale<typename PS::Single, LOGN>), dim3(a.units * PS::Single::TILES), dim3(NTT_THREADS), 0, st, a, ep);
    } else {
        hipError_t e;
        if (add_src) {
            hipLaunchKernelGGL((k_ntt_col_addsrc<typename PS::Col, LOGN>), dim3(a.units * PS::Col::TILES), dim3(NTT_THREADS), 0, st, a, *add_src);
            e = hipGetLastError();
        } else {
            e = launch_pass<typename PS::Col, LOGN, false, true>(st, a);
        }
        if (e != hipSuccess) return e;
        PassArgs second = a;
        second.src = nullptr;
        hipLaunchKernelGGL((k_ntt_row_subscale<typename PS::Row, LOGN>), dim3(a.units * PS::Row::TILES), dim3(NTT_THREADS), 0, st, second, ep);
    }
    return hipGetLastError();
}

bool ntt_subscale_supported(int logn) { return logn >= 5 && logn <= NTT_MAX_LOGN; }

hipError_t launch_ntt_subscale(hipStream_t st, const PassArgs &a, const RowEpiArgs &ep, int logn, int path, const ColAddSrc *add_src)
{
    if (a.units == 0) return hipSuccess;
    if (a.map || !ntt_subscale_supported(logn) || a.units / a.limbs > 3) return hipErrorInvalidValue;
    switch (logn) {
#define FHE_CASE(L) \
    case L: return path == PATH_F64 ? launch_subscale_t<ArithF64, L>(st, a, ep, add_src) : launch_subscale_t<ArithU64, L>(st, a, ep, add_src);
        FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13)
        FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// Key-switch inner product fused with the last pass of the forward transform of the extended digits (MULTEVK with the NTT
// that precedes it in the reference's trace, profile_framewk/build/data/ckks/16384_4:471-452).  One workgroup = one owned
// limb (row jj of ext / acc / the key) x one row tile.  For every digit d: the limb's row-pass steps run on ext[d][jj]'s
// tile in LDS (the digit's own limbs take the NTT-form input c instead), each thread then picks up its 16-byte pairs of the
// finished tile where the copy-out phase would, multiplies by the two key halves and adds into registers.  After the last
// digit the two sums are written once.  Exact integer arithmetic in both paths: the result equals k_ks_mac's word for word.
// The 64 registers of sums put the kernel at two workgroups per CU; a variant with one key half per workgroup (32 registers of
// sums, the row passes run twice) measured 15-30 % slower at every shape (profiles/r02_ks_variants.txt) and is not kept.
// ---------------------------------------------------------------------------
template <class A, int LOGN, int GEO>
__global__ __launch_bounds__(NTT_THREADS, 2) void k_ks_rowmac(KsMacArgs a)
{
    typedef MidPasses<A, LOGN, GEO> MP;
    // (2^16: the row pass in three register steps of 8 / 8 / 4 points instead of two of 16 -- one more exchange through LDS, but its
    // registers no longer compete with the 64 registers of sums: with the two-step form the kernel sat at 256 VGPRs with 49 spilled)
    typedef typename std::conditional<LOGN == 16 && GEO == 1,
                                      RowPass<A, Steps<3, 3, 2>, LOGN, MP::TR, NTT_THREADS, false, IO_LAZY, IO_LAZY, MP::F::RED_SECOND, MP::SB>,
                                      typename MP::Fwd>::type FR;
    static_assert(FR::STAGED, "fused inner product needs the staged row pass");
    typedef typename FR::elem elem;
    constexpr int PAIRS = FR::TROWS * FR::NPTS / 2;
    constexpr int PER = (PAIRS + NTT_THREADS - 1) / NTT_THREADS;
    __shared__ __attribute__((aligned(16))) elem lds[FR::LDS_ELEMS];
    const u32 jj = blockIdx.x / FR::TILES, tile = blockIdx.x % FR::TILES;
    const u32 tl = jj < a.cn ? a.clo + jj : jj + a.sp_shift;         // table limb of this row
    const LimbParams &p = a.lp[tl];
    if (p.path != A::PATH) return;                                   // the other instantiation's limb (uniform per workgroup)
    const typename A::Ctx ctx = A::make_ctx(p);
    const Tw inv_n = p.inv_n;
    const TwPtr tw = as_global(p.fwd);
    const int tid = threadIdx.x;
    const u32 row0 = tile * FR::TROWS;
    const size_t toff = (size_t)row0 * FR::NPTS;                      // tile offset inside a limb
    elem s0[PER][2], s1[PER][2];
#pragma unroll
    for (int k = 0; k < PER; k++) s0[k][0] = s0[k][1] = s1[k][0] = s1[k][1] = elem(0);
    int terms = 0;
    for (u32 d = 0; d < a.dnum; d++) {
        const u32 lo = d * a.alpha, hi = lo + a.alpha < a.L ? lo + a.alpha : a.L;
        const bool own = jj < a.cn && tl >= lo && tl < hi;
        if (!own) {
            u64 *src = const_cast<u64 *>(a.ext) + (((size_t)d * a.M + jj) << LOGN) + toff;
            if (d) __syncthreads();                                  // the previous digit's pairs have been read
            fwd_steps<FR>(tid, src, lds, tw, row0, ctx, inv_n);
            __syncthreads();
        }
        const u64 *k0 = a.evk + ((((size_t)d * 2) * a.M + jj) << LOGN) + toff, *k1 = k0 + ((size_t)a.M << LOGN);
        ++terms;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int i = tid + k * NTT_THREADS;
            if (PAIRS % NTT_THREADS == 0 || i < PAIRS) {
                const u32 row = (u32)i / (FR::NPTS / 2), g = ((u32)i % (FR::NPTS / 2)) * 2;
                const size_t off = (size_t)row * FR::NPTS + g;
                ulonglong2 cw = ulonglong2{0, 0};
                if (own) cw = *reinterpret_cast<const ulonglong2 *>(a.c + ((size_t)jj << LOGN) + toff + off);
                // key words are touched once per call and the shapes that come here carry hundreds of MiB of them: non-temporal
                // loads, so that they do not push the digits' tiles and the sums out of the caches
                typedef u64 u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(k0 + off)), v1 = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(k1 + off));
                ulonglong2 w0 = ulonglong2{v0.x, v0.y}, w1 = ulonglong2{v1.x, v1.y};
                // words outside [0, q) (bit-flipped inputs, reliability_test/dotprod_test.cu:38-61) take ONE cold branch; no calls on the hot path
                if (__builtin_expect((cw.x >= p.q) | (cw.y >= p.q) | (w0.x >= p.q) | (w0.y >= p.q) | (w1.x >= p.q) | (w1.y >= p.q), 0)) {
                    cw.x = barrett128(cw.x, 0, p.q, p.barrett_lo, p.barrett_hi);
                    cw.y = barrett128(cw.y, 0, p.q, p.barrett_lo, p.barrett_hi);
                    w0.x = barrett128(w0.x, 0, p.q, p.barrett_lo, p.barrett_hi);
                    w0.y = barrett128(w0.y, 0, p.q, p.barrett_lo, p.barrett_hi);
                    w1.x = barrett128(w1.x, 0, p.q, p.barrett_lo, p.barrett_hi);
                    w1.y = barrett128(w1.y, 0, p.q, p.barrett_lo, p.barrett_hi);
                }
                elem x0, x1;
                if (own) {
                    x0 = A::from_canonical(cw.x);
                    x1 = A::from_canonical(cw.y);
                } else {
                    const elem *src = lds + row * FR::ROW_LDS + row_pad(g);
                    x0 = src[0];
                    x1 = src[1];
                }
                const elem y00 = A::from_canonical(w0.x), y01 = A::from_canonical(w0.y), y10 = A::from_canonical(w1.x), y11 = A::from_canonical(w1.y);
                if constexpr (A::PATH == PATH_F64) {
                    A::lazy_acc(s0[k][0], A::mulvar_lazy(x0, y00, ctx), terms, ctx);
                    A::lazy_acc(s0[k][1], A::mulvar_lazy(x1, y01, ctx), terms, ctx);
                    A::lazy_acc(s1[k][0], A::mulvar_lazy(x0, y10, ctx), terms, ctx);
                    A::lazy_acc(s1[k][1], A::mulvar_lazy(x1, y11, ctx), terms, ctx);
                } else {
                    A::lazy_acc(s0[k][0], A::mulvar_lazy(x0, y00, p), terms, ctx);
                    A::lazy_acc(s0[k][1], A::mulvar_lazy(x1, y01, p), terms, ctx);
                    A::lazy_acc(s1[k][0], A::mulvar_lazy(x0, y10, p), terms, ctx);
                    A::lazy_acc(s1[k][1], A::mulvar_lazy(x1, y11, p), terms, ctx);
                }
            }
#if defined(__HIP_DEVICE_COMPILE__)
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep at most four pairs' loads in flight: the sums already hold 64 registers
#endif
        }
    }
    u64 *o0 = a.acc + ((size_t)jj << LOGN) + toff, *o1 = o0 + ((size_t)a.M << LOGN);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * NTT_THREADS;
        if (PAIRS % NTT_THREADS == 0 || i < PAIRS) {
            const u32 row = (u32)i / (FR::NPTS / 2), g = ((u32)i % (FR::NPTS / 2)) * 2;
            const size_t off = (size_t)row * FR::NPTS + g;
            ulonglong2 r0, r1;
            r0.x = A::canonical(s0[k][0], ctx);
            r0.y = A::canonical(s0[k][1], ctx);
            r1.x = A::canonical(s1[k][0], ctx);
            r1.y = A::canonical(s1[k][1], ctx);
            *reinterpret_cast<ulonglong2 *>(o0 + off) = r0;
            *reinterpret_cast<ulonglong2 *>(o1 + off) = r1;
        }
    }
}

// ---------------------------------------------------------------------------
// The fused inner product with the rows' twiddle factors kept in LDS (FP64 limbs, two-launch sizes).  Every digit's extension of a
// limb runs the SAME row pass on the same rows, and a row pass reads 16 bytes of table per point -- twice its data.  Round 2's kernel
// fetched them again for every digit (PMC: 65 % of the wave-cycles waiting on memory, 0.27 of the vector issue rate, no LDS pressure).
// Here a workgroup stages its tile's factors once, 8 bytes each (the quotient w/q is formed as w * (1/q) on the fly, ntt_core.hpp
// RowTwLds), and runs all the digits from them: 32 KiB of factors next to the 33 KiB image, still two workgroups per CU.
// The looser quotient estimate costs one more range reduction per pass (every fourth stage, from the pass's first stage on).
// ---------------------------------------------------------------------------
template <int LOGN>
struct RowMacLt {
    typedef ArithF64 A;
    typedef MidPasses<A, LOGN, 1> MP;
    typedef typename MP::PL PL;
    static constexpr int PR = PL::Row::P, NPTS = 1 << PR, S0 = LOGN - PR, TR = MP::TR;
    typedef typename std::conditional<PR == 8, Steps<3, 3, 2>, typename PL::Row>::type ST;
    static constexpr u32 red_every4()
    {
        u32 m = 0;
        for (int u = 0; u < PR; u += 4) m |= 1u << u;
        return m;
    }
    // STAGE_BOTH: the register steps take the tile from the LDS image (the kernel copies it there itself, one digit ahead)
    typedef RowPass<A, ST, LOGN, TR, NTT_THREADS, false, IO_LAZY, IO_LAZY, red_every4(), MP::SB, 0, false, true> FR;
    static constexpr int TW_PER_ROW = NPTS - 1, TW_ELEMS = TR * TW_PER_ROW;
    static constexpr size_t LDS_BYTES = ((size_t)FR::LDS_ELEMS + TW_ELEMS) * 8;
    static_assert(TR * NPTS == 16 * NTT_THREADS, "sixteen points per thread");
};

template <class FR, int E, class TWS>
FHE_D void lt_steps(int tid, typename FR::elem *lds, const TWS &tw, u32 row0, const typename FR::Arith::Ctx &ctx, const Tw &inv_n)
{
    if constexpr (E <= FR::NSTEP) {
        FR::template phase<E, NoTap, TWS>(tid, nullptr, lds, tw, row0, ctx, inv_n);
        __syncthreads();
        lt_steps<FR, E + 1, TWS>(tid, lds, tw, row0, ctx, inv_n);
    }
}

// Software pipeline of a workgroup (one owned limb x one row tile, all digits): every request to memory is made a whole phase
// before its words are needed, in one batch, with no branch between the requests --
//   tile of digit d+1  : requested before digit d's butterflies, copied into the LDS image after digit d's products;
//   key words of digit d: requested before digit d's butterflies, consumed after them;
//   twiddle factors    : staged once, sixteen requests per thread in flight together.
// (Round 2's form asked for a digit's tile, its factors and its key words one after the other, each behind the previous one's
// arithmetic and a cold-path branch per pair: the three parts of a digit ran back to back, 37 + 30 us of 125.)
template <int LOGN>
__global__ __launch_bounds__(NTT_THREADS, 2) void k_ks_rowmac_lt(KsMacArgs a)
{
    typedef RowMacLt<LOGN> RM;
    typedef typename RM::FR FR;
    typedef ArithF64 A;
    typedef double elem;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    constexpr int PAIRS = FR::TROWS * FR::NPTS / 2;
    constexpr int PER = PAIRS / NTT_THREADS;
    static_assert(PER == 8, "sixteen points per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char rowmac_lds[];
    elem *lds = reinterpret_cast<elem *>(rowmac_lds);
    double *twl = lds + FR::LDS_ELEMS;
    const u32 jj = blockIdx.x / FR::TILES, tile = blockIdx.x % FR::TILES;
    const u32 tl = jj < a.cn ? a.clo + jj : jj + a.sp_shift;         // table limb of this row
    const LimbParams &p = a.lp[tl];
    if (p.path != PATH_F64) return;                                  // integer-path limbs: launch_ks_rowmac runs k_ks_rowmac for them
    const A::Ctx ctx = A::make_ctx(p);
    const Tw inv_n = p.inv_n;
    const TwPtr tw = as_global(p.fwd);
    const int tid = threadIdx.x;
    const u32 row0 = tile * FR::TROWS;
    const size_t toff = (size_t)row0 * FR::NPTS;                      // tile offset inside a limb
    // the digit whose own limb this row is (its "extension" is the input c itself: no tile, no transform); dnum = none
    const u32 own_d = jj < a.cn ? tl / a.alpha : a.dnum;
    // offsets of this thread's pairs inside a tile (in words) and inside the LDS image (in elements)
    u32 goff[PER], loff[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u32 i = (u32)tid + (u32)k * NTT_THREADS, row = i / (FR::NPTS / 2), g = (i % (FR::NPTS / 2)) * 2;
        goff[k] = row * FR::NPTS + g;
        loff[k] = row * FR::ROW_LDS + row_pad(g);
    }
    auto tile_of = [&](u32 d) { return a.ext + (((size_t)d * a.M + jj) << LOGN) + toff; };
    // first tile: in flight under the staging of the factors
    u64x2 tv[PER];
    u32 d_tile = own_d == 0 ? 1u : 0u;                                // digit whose tile tv holds (>= dnum: none)
    if (d_tile < a.dnum) {
        const u64 *src = tile_of(d_tile);
#pragma unroll
        for (int k = 0; k < PER; k++) tv[k] = *reinterpret_cast<const u64x2 *>(src + goff[k]);
    }
    // the tile's factors: stage S0 + t, blocks (row0 + r) * 2^t + j, r < TR, j < 2^t.  Lanes run along the table's stored order
    // (ntt_core.hpp tw_index: natural below stage SBLK, blocked from there on -- 2^NB - 1 runs of RUN consecutive entries, then the
    // natural stages' RUN - TR entries), so the reads are whole lines; sixteen requests per lane, all made before the first is used.
    {
        constexpr int SB = RM::MP::SB, S0 = RM::S0, PR = RM::PR, TR = RM::TR;
        constexpr int NB = LOGN - SB, NN = PR - NB;                    // blocked / natural stages of the pass
        constexpr u32 RUN = (u32)TR << NN, BLK = ((1u << NB) - 1u) * RUN, TOTAL = BLK + RUN - TR;
        static_assert((RUN & (RUN - 1)) == 0 && (TR & (TR - 1)) == 0, "powers of two");
        constexpr int LOG_RUN = __builtin_ctz(RUN), LOG_TR = __builtin_ctz((u32)TR);
        double wv[16];
        u32 dst[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            u32 f = (u32)tid + (u32)k * NTT_THREADS;
            f = f < TOTAL ? f : TOTAL - 1u;                             // (the last sixteen lanes repeat the last entry: no branch)
            const bool blk = f < BLK;
            // blocked stages
            const u32 rr = (f >> LOG_RUN) + 1u, pp = f & (RUN - 1u);
            const u32 u = 31u - (u32)__builtin_clz(rr | 1u), bb = rr - (1u << u), tb = (u32)NN + u;
            const u32 p0 = ((u32)row0 << tb) >> u;
            const u32 ib = ((p0 + pp) << u) | bb, srcb = (((1u << u) + bb) << SB) + p0 + pp;
            // natural stages: stage t holds TR * 2^t entries
            const u32 e = f - BLK;
            const u32 tn = 31u - (u32)__builtin_clz(((e >> LOG_TR) + 1u) | 1u);
            const u32 in = ((u32)row0 << tn) + (e - (((1u << tn) - 1u) << LOG_TR)), srcn = (1u << (S0 + tn)) + in;
            const u32 t = blk ? tb : tn, i = blk ? ib : in, src = blk ? srcb : srcn;
            const u32 r = (i >> t) - row0, j = i & ((1u << t) - 1u);
            dst[k] = r * RM::TW_PER_ROW + ((1u << t) - 1u) + j;
            wv[k] = u64_bits_to_double(tw[src].a);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) twl[dst[k]] = wv[k];
    }
    const RowTwLds tws{twl, ctx.ninv, row0, RM::S0, RM::TW_PER_ROW};
    elem s0[PER][2], s1[PER][2];
#pragma unroll
    for (int k = 0; k < PER; k++) s0[k][0] = s0[k][1] = s1[k][0] = s1[k][1] = 0.0;
    int terms = 0;
    const u64 q = p.q;
    for (u32 d = 0; d < a.dnum; d++) {
        const bool own = d == own_d;
        const u64 *k0 = a.evk + ((((size_t)d * 2) * a.M + jj) << LOGN) + toff, *k1 = k0 + ((size_t)a.M << LOGN);
        // key words: touched once per call, hundreds of MiB of them -- non-temporal, so that they do not push the tiles out of the caches
        u64x2 kv0[PER], kv1[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            kv0[k] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(k0 + goff[k]));
            kv1[k] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(k1 + goff[k]));
        }
        elem x[PER][2];
        if (own) {
            const u64 *src = a.c + ((size_t)jj << LOGN) + toff;
            u64x2 cw[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) cw[k] = *reinterpret_cast<const u64x2 *>(src + goff[k]);
            bool bad = false;
#pragma unroll
            for (int k = 0; k < PER; k++) bad |= (cw[k].x >= q) | (cw[k].y >= q);
            if (__builtin_expect(bad, 0)) {
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    cw[k].x = barrett128(cw[k].x, 0, q, p.barrett_lo, p.barrett_hi);
                    cw[k].y = barrett128(cw[k].y, 0, q, p.barrett_lo, p.barrett_hi);
                }
            }
#pragma unroll
            for (int k = 0; k < PER; k++) {
                x[k][0] = A::from_canonical(cw[k].x);
                x[k][1] = A::from_canonical(cw[k].y);
            }
        } else {
            // tv holds this digit's tile (requested a digit ago): into the image, the next one's request goes out, then the steps
            __syncthreads();                                         // the previous image's pairs have been read (the factors are staged)
#pragma unroll
            for (int k = 0; k < PER; k++) *reinterpret_cast<u64x2 *>(lds + loff[k]) = tv[k];
            u32 nd = d + 1;
            nd += nd == own_d ? 1u : 0u;
            if (nd < a.dnum) {
                const u64 *src = tile_of(nd);
#pragma unroll
                for (int k = 0; k < PER; k++) tv[k] = *reinterpret_cast<const u64x2 *>(src + goff[k]);
            }
            __syncthreads();
            lt_steps<FR, 1, RowTwLds>(tid, lds, tws, row0, ctx, inv_n);
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const double2 v = *reinterpret_cast<const double2 *>(lds + loff[k]);
                x[k][0] = v.x;
                x[k][1] = v.y;
            }
        }
        // words outside [0, q) (bit-flipped inputs, reliability_test/dotprod_test.cu:38-61): ONE cold branch per digit
        {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < PER; k++) bad |= (kv0[k].x >= q) | (kv0[k].y >= q) | (kv1[k].x >= q) | (kv1[k].y >= q);
            if (__builtin_expect(bad, 0)) {
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    kv0[k].x = barrett128(kv0[k].x, 0, q, p.barrett_lo, p.barrett_hi);
                    kv0[k].y = barrett128(kv0[k].y, 0, q, p.barrett_lo, p.barrett_hi);
                    kv1[k].x = barrett128(kv1[k].x, 0, q, p.barrett_lo, p.barrett_hi);
                    kv1[k].y = barrett128(kv1[k].y, 0, q, p.barrett_lo, p.barrett_hi);
                }
            }
        }
        ++terms;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            A::lazy_acc(s0[k][0], A::mulvar_lazy(x[k][0], A::from_canonical(kv0[k].x), ctx), terms, ctx);
            A::lazy_acc(s0[k][1], A::mulvar_lazy(x[k][1], A::from_canonical(kv0[k].y), ctx), terms, ctx);
            A::lazy_acc(s1[k][0], A::mulvar_lazy(x[k][0], A::from_canonical(kv1[k].x), ctx), terms, ctx);
            A::lazy_acc(s1[k][1], A::mulvar_lazy(x[k][1], A::from_canonical(kv1[k].y), ctx), terms, ctx);
        }
    }
    u64 *o0 = a.acc + ((size_t)jj << LOGN) + toff, *o1 = o0 + ((size_t)a.M << LOGN);
#pragma unroll
    for (int k = 0; k < PER; k++) {
        *reinterpret_cast<u64x2 *>(o0 + goff[k]) = u64x2{A::canonical(s0[k][0], ctx), A::canonical(s0[k][1], ctx)};
        *reinterpret_cast<u64x2 *>(o1 + goff[k]) = u64x2{A::canonical(s1[k][0], ctx), A::canonical(s1[k][1], ctx)};
    }
}

template <int LOGN>
static hipError_t launch_rowmac_lt(hipStream_t st, const KsMacArgs &a)
{
    typedef RowMacLt<LOGN> RM;
    static_assert(RM::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ks_rowmac_lt<LOGN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RM::LDS_BYTES);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL((k_ks_rowmac_lt<LOGN>), dim3(a.M * RM::FR::TILES), dim3(NTT_THREADS), RM::LDS_BYTES, st, a);
    return hipSuccess;
}

bool ks_rowmac_supported(int logn) { return logn >= 5 && logn <= NTT_MAX_LOGN; }

template <class A, int LOGN>
static void launch_rowmac_t(hipStream_t st, const KsMacArgs &a)
{
    constexpr int GEO = LOGN >= 13 ? 1 : 0;
    typedef typename MidPasses<A, LOGN, GEO>::Fwd FR;
    // FHE_KS_ROWMAC (read once): 1 (default) = FP64 limbs of the two-launch sizes run the pipelined form with the tile's twiddle
    // factors in LDS (k_ks_rowmac_lt); 0 = round 2's kernel everywhere
    static const int variant = getenv("FHE_KS_ROWMAC") ? atoi(getenv("FHE_KS_ROWMAC")) : 1;
    if constexpr (LOGN >= 13 && A::PATH == PATH_F64) {
        if (variant) {
            (void)launch_rowmac_lt<LOGN>(st, a);
            return;
        }
    }
    hipLaunchKernelGGL((k_ks_rowmac<A, LOGN, GEO>), dim3(a.M * FR::TILES), dim3(NTT_THREADS), 0, st, a);
}

hipError_t launch_ks_rowmac(hipStream_t st, const KsMacArgs &a, bool has_f64, bool has_u64)
{
    if (!a.M) return hipSuccess;
    if (!ks_rowmac_supported(a.logn)) return hipErrorInvalidValue;
    switch (a.logn) {
#define FHE_CASE(L)                                              \
    case L:                                                      \
        if (has_f64) launch_rowmac_t<ArithF64, L>(st, a);        \
        if (has_u64) launch_rowmac_t<ArithU64, L>(st, a);        \
        break;
        FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8) FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13)
        FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// true when launch_ntt would use PassArgs::scratch for this transform (the caller then provides packed_scratch_words per unit)
bool ntt_packed_supported(int logn, bool inverse, int path)
{
    return logn == 16 && !inverse && path == PATH_F64 && packed_ok<ArithF64, 16, false, 1>();
}
size_t ntt_packed_scratch_words() { return (size_t)256 * PK_BLOCK_WORDS; }

// (Batches larger than the Infinity Cache are cut into sub-batches by the caller, capi.cpp ntt_batch: one launch pair each.)
hipError_t launch_ntt(hipStream_t st, const PassArgs &a, int logn, bool inverse, int path, int geo, int which, bool resident)
{
    if (a.units == 0) return hipSuccess;
    if (a.src || a.tmp) {      // out-of-place: the plain launches only (no packed hand-off, no resident pass)
        if (a.scratch) return hipErrorInvalidValue;
        geo = 1;
        resident = false;
    }
    return path == PATH_F64 ? launch_size<ArithF64>(st, a, logn, inverse, geo, which, resident)
                            : launch_size<ArithU64>(st, a, logn, inverse, geo, which, resident);
}

} // namespace fhe
