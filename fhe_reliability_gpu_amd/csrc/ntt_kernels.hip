// ntt_kernels.hip -- gfx950 kernels for the batched forward / inverse NTT.
// One workgroup (256 threads = 4 waves of 64) owns one tile of one limb:
//   column pass: TC adjacent columns x 2^PC points, lanes along the columns;
//   row pass   : TR contiguous rows of 2^PR points, lanes along the row.
// The butterfly code lives in ntt_core.hpp; this file only binds it to the grid.
#include "ntt_launch.hpp"
#include "ntt_plan.hpp"

namespace fhe {

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
    PASS::template phase<0>(tid, base, lds, tw, row0, ctx, inv_n);
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

template <class PASS, int LOGN, bool INV, bool IS_COL>
static hipError_t launch_pass(hipStream_t st, const PassArgs &a)
{
    const u32 blocks = a.units * PASS::TILES;
    hipLaunchKernelGGL((k_ntt_pass<PASS, LOGN, INV, IS_COL>), dim3(blocks), dim3(NTT_THREADS), 0, st, a);
    return hipGetLastError();
}

// which: -1 = whole transform, 0 / 1 = only the first / second launch of a two-pass size (used by
// the fault-injection hook that corrupts the intermediate between the passes)
template <class A, int LOGN, bool INV, int GEO>
static hipError_t launch_transform(hipStream_t st, const PassArgs &a, int which)
{
    typedef Passes<A, LOGN, INV, GEO> PS;
    if constexpr (!PS::G::TWO_PASS) {
        if (which == 1) return hipSuccess;
        return launch_pass<typename PS::Single, LOGN, INV, false>(st, a);
    } else if constexpr (!INV) {
        hipError_t e = which == 1 ? hipSuccess : launch_pass<typename PS::Col, LOGN, INV, true>(st, a);
        if (e != hipSuccess || which == 0) return e;
        return launch_pass<typename PS::Row, LOGN, INV, false>(st, a);
    } else {
        hipError_t e = which == 1 ? hipSuccess : launch_pass<typename PS::Row, LOGN, INV, false>(st, a);
        if (e != hipSuccess || which == 0) return e;
        return launch_pass<typename PS::Col, LOGN, INV, true>(st, a);
    }
}

template <class A>
static hipError_t launch_size(hipStream_t st, const PassArgs &a, int logn, bool inverse, int geo, int which)
{
    if (logn == 16 && geo == 0)   // tuning: the wide-tile geometry is kept for 2^16 only
        return inverse ? launch_transform<A, 16, true, 0>(st, a, which) : launch_transform<A, 16, false, 0>(st, a, which);
    switch (logn) {
#define FHE_CASE(L)                                                        \
    case L:                                                                \
        return inverse ? launch_transform<A, L, true, (L >= 13 ? 1 : 0)>(st, a, which) : launch_transform<A, L, false, (L >= 13 ? 1 : 0)>(st, a, which);
        FHE_CASE(1) FHE_CASE(2) FHE_CASE(3) FHE_CASE(4) FHE_CASE(5) FHE_CASE(6) FHE_CASE(7) FHE_CASE(8)
        FHE_CASE(9) FHE_CASE(10) FHE_CASE(11) FHE_CASE(12) FHE_CASE(13) FHE_CASE(14) FHE_CASE(15) FHE_CASE(16)
        FHE_CASE(17) FHE_CASE(18) FHE_CASE(19) FHE_CASE(20)
#undef FHE_CASE
    default:
        return hipErrorInvalidValue;
    }
}

// (Chunking large batches so that the second launch would find the first launch's output in the
// Infinity Cache was measured and brings nothing: a 512 MiB batch runs at the HBM-streaming rate
// either way, and splitting costs launches.  One launch pair per call.)
hipError_t launch_ntt(hipStream_t st, const PassArgs &a, int logn, bool inverse, int path, int geo, int which)
{
    if (a.units == 0) return hipSuccess;
    return path == PATH_F64 ? launch_size<ArithF64>(st, a, logn, inverse, geo, which) : launch_size<ArithU64>(st, a, logn, inverse, geo, which);
}

} // namespace fhe
