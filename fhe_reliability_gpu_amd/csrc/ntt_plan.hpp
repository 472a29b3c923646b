// ntt_plan.hpp -- compile-time decomposition of a 2^LOGN transform into passes and
// register steps, plus the block-level glue shared by the HIP kernels
// (ntt_kernels.hip) and the CPU emulation used by the tests (tests/emu).
#pragma once
#include "ntt_core.hpp"

namespace fhe {

constexpr int NTT_THREADS = 256;
constexpr int NTT_MAX_LOGN = 20;

// Stage split per size.  Row pass = stages inside contiguous rows of 2^PR points
// (always present); column pass = the PC leading stages (absent for LOGN <= 12:
// the whole limb fits one workgroup's LDS and is read and written exactly once).
template <int LOGN> struct Plan;
#define FHE_PLAN(LOGN_, C0, C1, C2, R0, R1, R2)                      \
    template <> struct Plan<LOGN_> {                                 \
        typedef Steps<C0, C1, C2> Col;                               \
        typedef Steps<R0, R1, R2> Row;                               \
        static_assert(Col::P + Row::P == LOGN_, "bad plan");         \
    };
FHE_PLAN(1, 0, 0, 0, 1, 0, 0)
FHE_PLAN(2, 0, 0, 0, 2, 0, 0)
FHE_PLAN(3, 0, 0, 0, 3, 0, 0)
FHE_PLAN(4, 0, 0, 0, 4, 0, 0)
FHE_PLAN(5, 0, 0, 0, 3, 2, 0)
FHE_PLAN(6, 0, 0, 0, 3, 3, 0)
FHE_PLAN(7, 0, 0, 0, 4, 3, 0)
FHE_PLAN(8, 0, 0, 0, 4, 4, 0)
FHE_PLAN(9, 0, 0, 0, 3, 3, 3)
FHE_PLAN(10, 0, 0, 0, 4, 3, 3)
FHE_PLAN(11, 0, 0, 0, 4, 4, 3)
FHE_PLAN(12, 0, 0, 0, 4, 4, 4)
FHE_PLAN(13, 3, 2, 0, 4, 4, 0)
FHE_PLAN(14, 3, 3, 0, 4, 4, 0)
FHE_PLAN(15, 4, 3, 0, 4, 4, 0)
FHE_PLAN(16, 4, 4, 0, 4, 4, 0)
FHE_PLAN(17, 4, 4, 0, 3, 3, 3)
FHE_PLAN(18, 4, 4, 0, 4, 3, 3)
FHE_PLAN(19, 4, 4, 0, 4, 4, 3)
FHE_PLAN(20, 4, 4, 0, 4, 4, 4)
#undef FHE_PLAN


// First stage whose table entries are stored blocked (ntt_core.hpp tw_index): the stages of the
// row plan's last register step.  One value per size, used by every kernel variant and by the host
// code that lays the tables out.
template <int LOGN> struct BlkStage {
    typedef typename Plan<LOGN>::Row R;
    static constexpr int value = LOGN - R::k(R::NSTEP - 1);
};
inline int blk_stage_rt(int logn)
{
    switch (logn) {
#define FHE_B(L) case L: return BlkStage<L>::value;
        FHE_B(1) FHE_B(2) FHE_B(3) FHE_B(4) FHE_B(5) FHE_B(6) FHE_B(7) FHE_B(8) FHE_B(9) FHE_B(10)
        FHE_B(11) FHE_B(12) FHE_B(13) FHE_B(14) FHE_B(15) FHE_B(16) FHE_B(17) FHE_B(18) FHE_B(19) FHE_B(20)
#undef FHE_B
    default: return logn;
    }
}
// natural table position k (= 2^sigma + i) -> position in the stored layout
inline u32 tw_stored_index(int logn, u32 k)
{
    if (k == 0) return 0;
    const int sigma = 31 - __builtin_clz(k);
    return tw_index(blk_stage_rt(logn), sigma, k - (1u << sigma));
}

// GEO 0: largest column tile that fits 64 KiB of LDS (256-byte row segments at 2^16);
// GEO 1: 16-column tiles (128-byte segments, ~35 KiB of LDS, one register set per thread) so
// that four workgroups share a CU.
template <int LOGN, int GEO = 0> struct PlanGeom {
    typedef Plan<LOGN> PL;
    static constexpr int PC = PL::Col::P, PR = PL::Row::P;
    static constexpr bool TWO_PASS = PC > 0;
    static constexpr int TC = !TWO_PASS ? 1 : GEO == 0 ? cmin(64, 8192 >> PC) : (PC <= 8 ? 16 : 8);
    static constexpr int TR = !TWO_PASS ? 1 : PR <= 8 ? 16 : PR == 9 ? 8 : PR == 10 ? 4 : PR == 11 ? 2 : 1;   // 4096 points per row tile
};

// Entry of an explicit unit list: where the limb-polynomial starts (in units of N words from `data`) and
// which table limb it belongs to.
struct UnitRef {
    u32 off, limb;
    u32 src = 0xFFFFFFFFu;  // optional: the unit's FIRST launch reads limb `src` (units of N words from PassArgs::src) instead of its own
                            // data -- a one-limb base extension (x mod q_limb) riding on the load, no converted copy in memory
};

// What a pass needs to know about the launch.
struct PassArgs {
    u64 *data;              // [n_poly][limbs][N]
    const LimbParams *lp;   // table of all limbs of the table set
    u32 limb0;              // first modulus index (start_modulus_idx of the Phantom call)
    u32 limbs;              // limbs of each polynomial handled by this launch (same arithmetic path)
    u32 units;              // n_poly * limbs
    u32 poly_stride;        // distance between polynomials in units of one limb (>= limbs)
    const UnitRef *map = nullptr;     // optional (two-launch path): units[] given explicitly instead of the [poly][limb] grid --
                            // e.g. "every limb of every key-switch digit except the digit's own" in one launch
    u64 *scratch = nullptr; // optional: packed hand-off area, PK_BLOCK_WORDS * 256 words per unit (ntt_core.hpp)
    u64 *tmp = nullptr;     // optional (two-launch sizes): hand-off buffer of the same layout -- the first launch writes there, the second
                            // reads it and writes data ("ping-pong": both launches out of place, the result still lands in data)
    u64 src_bcast = 0;      // with src: every limb of polynomial p reads the ONE source limb at src + p * src_bcast words (its residues modulo
                            // each limb's prime are taken on the load): the one-limb conversions of a rescale / a K = 1 mod-down
    const u64 *src = nullptr; // optional: the transform's FIRST launch reads its input from here (same layout as data) -- an
                            // out-of-place transform with no copy; the natural-order transforms (launch_ntt_gs) require it
    u32 stream_hint = 0;    // two-launch sizes: the external side of the two launches -- the first one's loads, the second one's stores --
                            // uses non-temporal accesses, the hand-off side stays cacheable (pieces of a batch that streams from HBM)
    u32 src_stride = 0;     // with src: distance between the SOURCE's polynomials in limbs when it differs from poly_stride (0 = the same)
    u32 tmp_stride = 0;     // with tmp: the hand-off buffer's polynomial stride (0 = poly_stride) -- a compact scratch for a window of limbs
    u32 galois = 0;         // with src, inverse transforms whose first launch stages its tile (N >= 2^5): the input is sigma_k(src) -- the
                            // NTT-domain Galois map applied on the load (galois_slot, ntt_core.hpp) -- instead of src: a rotation's
                            // automorphism rides on the opening INTT of its key switch
    u64 *galois_copy = nullptr; // optional: sigma_k(src) itself is also written there (same layout; the key switch's inner product reads it)
};

template <class A, int LOGN, bool INVERSE, int GEO = 0>
struct Passes {
    typedef PlanGeom<LOGN, GEO> G;
    typedef typename G::PL PL;
    // inverse transforms of the FP64 path: lazy range tracked per register (ntt_core.hpp INV_LAZY): the first launch starts from canonical
    // words (bound q; the fused product's inverse from |.| < 0.6 q), the second from whatever the first one's last step hands over
    static constexpr bool LAZY_INV = INVERSE && A::PATH == PATH_F64;
    static constexpr u32 RED_FIRST = LAZY_INV ? (INV_LAZY | 8u)
                                     : INVERSE ? reduce_mask(0, G::PR, A::INV_FIRST, A::INV_NEXT)
                                               : reduce_mask(0, G::TWO_PASS ? G::PC : G::PR, A::FWD_FIRST, A::FWD_NEXT);
    static constexpr u32 RED_SECOND = LAZY_INV ? (INV_LAZY | (u32)inv_lazy_pass_out8<typename PL::Row>(8))
                                      : INVERSE ? reduce_mask(G::PR, G::PC, A::INV_FIRST, A::INV_NEXT)
                                                : reduce_mask(G::PC, G::PR, A::FWD_FIRST, A::FWD_NEXT);
    // single pass
    static constexpr int SB = BlkStage<LOGN>::value;
    typedef RowPass<A, typename PL::Row, LOGN, 1, NTT_THREADS, INVERSE, IO_CANONICAL, IO_CANONICAL, RED_FIRST, SB> Single;
    typedef RowPass<A, typename PL::Row, LOGN, 1, NTT_THREADS, INVERSE, IO_CANONICAL, IO_CANONICAL, RED_FIRST, SB, 0, true> SingleNt;   // non-temporal loads and stores
    // forward: column pass then row pass; inverse: row pass then column pass
    typedef ColPass<A, typename PL::Col, LOGN, 0, G::TC, NTT_THREADS, INVERSE, INVERSE ? IO_LAZY : IO_CANONICAL,
                    INVERSE ? IO_CANONICAL : IO_LAZY, INVERSE ? RED_SECOND : RED_FIRST, SB> Col;
    typedef RowPass<A, typename PL::Row, LOGN, G::TR, NTT_THREADS, INVERSE, INVERSE ? IO_CANONICAL : IO_LAZY,
                    INVERSE ? IO_LAZY : IO_CANONICAL, INVERSE ? RED_FIRST : RED_SECOND, SB> Row;
    // the same passes with non-temporal accesses on the external (canonical) side
    typedef ColPass<A, typename PL::Col, LOGN, 0, G::TC, NTT_THREADS, INVERSE, INVERSE ? IO_LAZY : IO_CANONICAL,
                    INVERSE ? IO_CANONICAL : IO_LAZY, INVERSE ? RED_SECOND : RED_FIRST, SB, 0, true> ColNt;
    typedef RowPass<A, typename PL::Row, LOGN, G::TR, NTT_THREADS, INVERSE, INVERSE ? IO_CANONICAL : IO_LAZY,
                    INVERSE ? IO_LAZY : IO_CANONICAL, INVERSE ? RED_FIRST : RED_SECOND, SB, 0, true> RowNt;
};

// LDS-resident single pass for the sizes just above the 64 KiB static limit (opt-in, "ntt_resident"): a 2^13 / 2^14
// limb (64 / 128 KiB) still fits one CU's 160 KiB of LDS on gfx950, so the whole transform can run in one workgroup
// and the limb crosses the fabric once in and once out instead of twice.  Radix-32 register steps (32 points per
// thread) keep the number of LDS exchanges at two; 2^14 takes a CU's LDS alone (512 threads), 2^13 shares it between
// two workgroups.  Measured: with one or two workgroups per CU nothing overlaps a workgroup's
// load, arithmetic and store phases, and the halved traffic does not pay for that except for the 2^13 inverse
// (0.081 vs 0.105 ms per 128 MiB) -- hence not the default.
template <int LOGN> struct ResidentPlan {
    static constexpr bool OK = false;
    typedef Steps<1, 0, 0> S;
    static constexpr int THREADS = NTT_THREADS;
};
template <> struct ResidentPlan<13> {
    static constexpr bool OK = true;
    typedef Steps<5, 4, 4> S;
    static constexpr int THREADS = 256;
};
template <> struct ResidentPlan<14> {
    static constexpr bool OK = true;
    typedef Steps<5, 5, 4> S;
    static constexpr int THREADS = 512;
};
template <class A, int LOGN, bool INVERSE>
struct ResidentPass {
    typedef ResidentPlan<LOGN> RP;
    static_assert(!RP::OK || LOGN - RP::S::k(RP::S::NSTEP - 1) == BlkStage<LOGN>::value, "last register step must match the table layout");
    static constexpr u32 RED = INVERSE ? reduce_mask(0, LOGN, A::INV_FIRST, A::INV_NEXT) : reduce_mask(0, LOGN, A::FWD_FIRST, A::FWD_NEXT);
    typedef RowPass<A, typename RP::S, LOGN, 1, RP::THREADS, INVERSE, IO_CANONICAL, IO_CANONICAL, RED, BlkStage<LOGN>::value> Pass;
};

// Block -> work mapping.  One block = one tile of one unit (unit = one limb of one
// polynomial); the tiles of a unit are adjacent block indices.
template <class CP, int LOGN>
FHE_D u64 *col_tile_of(u32 unit, u32 tile, const PassArgs &a, u32 &limb)
{
    const u32 poly = unit / a.limbs, l = unit % a.limbs;
    limb = a.limb0 + l;
    return a.data + (((size_t)poly * a.poly_stride + l) << LOGN) + (size_t)tile * CP::TCOLS;
}
template <class RP, int LOGN>
FHE_D u64 *row_tile_of(u32 unit, u32 tile, const PassArgs &a, u32 &limb, u32 &row0)
{
    const u32 poly = unit / a.limbs, l = unit % a.limbs;
    limb = a.limb0 + l;
    row0 = tile * RP::TROWS;
    return a.data + (((size_t)poly * a.poly_stride + l) << LOGN) + (size_t)row0 * RP::NPTS;
}
// Launch order of the two-launch path is limb-major (all polynomials of limb 0, then limb 1, ...): the
// blocks in flight at any moment share one or two twiddle tables, which then live in the XCDs' L2
// instead of being streamed from the Infinity Cache once per polynomial.
template <class CP, int LOGN>
FHE_D u64 *col_tile(u32 block, const PassArgs &a, u32 &limb)
{
    const u32 unit = block / CP::TILES, tile = block % CP::TILES;
    if (a.map) {
        const UnitRef m = a.map[unit];
        limb = m.limb;
        return a.data + ((size_t)m.off << LOGN) + (size_t)tile * CP::TCOLS;
    }
    const u32 polys = a.units / a.limbs;
    const u32 l = unit / polys, poly = unit % polys;
    limb = a.limb0 + l;
    return a.data + (((size_t)poly * a.poly_stride + l) << LOGN) + (size_t)tile * CP::TCOLS;
}
template <class RP, int LOGN>
FHE_D u64 *row_tile(u32 block, const PassArgs &a, u32 &limb, u32 &row0)
{
    const u32 unit = block / RP::TILES, tile = block % RP::TILES;
    row0 = tile * RP::TROWS;
    if (a.map) {
        const UnitRef m = a.map[unit];
        limb = m.limb;
        return a.data + ((size_t)m.off << LOGN) + (size_t)row0 * RP::NPTS;
    }
    const u32 polys = a.units / a.limbs;
    const u32 l = unit / polys, poly = unit % polys;
    limb = a.limb0 + l;
    return a.data + (((size_t)poly * a.poly_stride + l) << LOGN) + (size_t)row0 * RP::NPTS;
}

} // namespace fhe
