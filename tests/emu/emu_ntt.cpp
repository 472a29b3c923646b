// emu_ntt.cpp -- CPU emulation of the HIP NTT passes (TEST INFRASTRUCTURE ONLY).
//
// Compiles fhe_reliability_gpu_amd/csrc/ntt_core.hpp / ntt_plan.hpp with g++ and
// runs every workgroup step as a loop over thread ids, so the tile indexing, the
// twiddle indexing, the LDS exchange pattern and the FP64 lazy-range schedule can be
// checked against the oracle without a GPU.  It is NOT a product path: the library
// never links this file.
//
//   g++ -O2 -std=c++17 -ffp-contract=off -shared -fPIC -I<csrc> emu_ntt.cpp -o libemu_ntt.so
#include "ntt_fused.hpp"

#include <array>
#include <cmath>
#include <vector>

using namespace fhe;

namespace {

// largest |value| (in units of q) any FP64 register held, to validate the lazy schedule
double g_max_ratio = 0.0;

template <class PASS, int LOGN, bool INV, bool IS_COL>
void emu_pass(const PassArgs &a)
{
    typedef typename PASS::Arith A;
    const u32 blocks = a.units * PASS::TILES;
    std::vector<typename PASS::elem> lds(PASS::LDS_ELEMS > 0 ? PASS::LDS_ELEMS : 1);
    for (u32 b = 0; b < blocks; b++) {
        u32 limb, row0 = 0;
        u64 *base;
        if constexpr (IS_COL) base = col_tile<PASS, LOGN>(b, a, limb);
        else base = row_tile<PASS, LOGN>(b, a, limb, row0);
        const LimbParams &p = a.lp[limb];
        auto ctx = A::make_ctx(p);
        const TwPtr tw = as_global(INV ? p.inv : p.fwd);
        for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<0>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
        if constexpr (PASS::NPHASE > 1)
            for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<1>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
        if constexpr (PASS::NPHASE > 2)
            for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<2>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
        if constexpr (PASS::NPHASE > 3)
            for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<3>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
        if constexpr (PASS::NPHASE > 4)
            for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<4>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
        if constexpr (A::PATH == PATH_F64 && false) {
            for (auto v : lds) {
                double r = std::fabs((double)v) / p.n;
                if (r > g_max_ratio) g_max_ratio = r;
            }
        }
    }
}

template <class A, int LOGN, bool INV>
void emu_transform(const PassArgs &a)
{
    typedef Passes<A, LOGN, INV, (LOGN >= 13 ? 1 : 0)> PS;   // the geometry launch_ntt uses by default
    if constexpr (!PS::G::TWO_PASS) {
        emu_pass<typename PS::Single, LOGN, INV, false>(a);
    } else if constexpr (!INV) {
        emu_pass<typename PS::Col, LOGN, INV, true>(a);
        emu_pass<typename PS::Row, LOGN, INV, false>(a);
    } else {
        emu_pass<typename PS::Row, LOGN, INV, false>(a);
        emu_pass<typename PS::Col, LOGN, INV, true>(a);
    }
}

// forward transform with the packed hand-off (ntt_core.hpp): the kernels' phase order with the per-thread registers that
// live across a barrier (chunk / x) kept per thread id
template <class A, int LOGN>
int emu_packed(const PassArgs &a)
{
    typedef Passes<A, LOGN, false, 1> PS;
    typedef typename PS::Col CP;
    typedef typename PS::Row RP;
    if constexpr (PS::G::TWO_PASS && A::PATH == PATH_F64) {
        if constexpr (CP::PACKABLE && RP::PACKABLE) {
            std::vector<u64> scratch((size_t)a.units * 256 * PK_BLOCK_WORDS, 0xDEADBEEFDEADBEEFull);
            std::vector<typename A::elem> lds(cmax(CP::LDS_ELEMS, RP::LDS_ELEMS));
            u64 *stage = reinterpret_cast<u64 *>(lds.data());
            for (u32 b = 0; b < a.units * CP::TILES; b++) {
                u32 limb;
                u64 *base = col_tile<CP, LOGN>(b, a, limb);
                const u32 unit = b / CP::TILES, tile = b % CP::TILES;
                const LimbParams &p = a.lp[limb];
                auto ctx = A::make_ctx(p);
                const TwPtr tw = as_global(p.fwd);
                for (int tid = 0; tid < CP::THREADS; tid++) CP::template phase<0>(tid, base, lds.data(), tw, 0u, ctx, p.inv_n);
                std::vector<std::array<u64, PK_WORDS>> chunk(CP::THREADS);
                for (int tid = 0; tid < CP::THREADS; tid++) {
                    u64 c[PK_WORDS];
                    CP::phase_last_packed(tid, lds.data(), tw, 0u, ctx, c);
                    for (int j = 0; j < PK_WORDS; j++) chunk[tid][j] = c[j];
                }
                for (int tid = 0; tid < CP::THREADS; tid++) {
                    u64 c[PK_WORDS];
                    for (int j = 0; j < PK_WORDS; j++) c[j] = chunk[tid][j];
                    CP::pack_stage(tid, stage, c);
                }
                for (int tid = 0; tid < CP::THREADS; tid++)
                    CP::pack_copy_out(tid, stage, scratch.data() + ((size_t)unit * 256 + (size_t)tile * 16) * PK_BLOCK_WORDS);
            }
            for (u32 b = 0; b < a.units * RP::TILES; b++) {
                u32 limb, row0 = 0;
                u64 *base = row_tile<RP, LOGN>(b, a, limb, row0);
                const u32 unit = b / RP::TILES, tile = b % RP::TILES;
                const LimbParams &p = a.lp[limb];
                auto ctx = A::make_ctx(p);
                const TwPtr tw = as_global(p.fwd);
                for (int tid = 0; tid < RP::THREADS; tid++) RP::unpack_copy_in(tid, stage, scratch.data() + (size_t)unit * 256 * PK_BLOCK_WORDS, tile);
                std::vector<std::array<typename A::elem, 16>> xs(RP::THREADS);
                for (int tid = 0; tid < RP::THREADS; tid++) {
                    typename A::elem x[16];
                    RP::phase_first_packed(tid, stage, tw, row0, ctx, x);
                    for (int r = 0; r < 16; r++) xs[tid][r] = x[r];
                }
                for (int tid = 0; tid < RP::THREADS; tid++) {
                    typename A::elem x[16];
                    for (int r = 0; r < 16; r++) x[r] = xs[tid][r];
                    RP::phase_first_store(tid, lds.data(), x);
                }
                for (int tid = 0; tid < RP::THREADS; tid++) RP::template phase<1>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
                for (int tid = 0; tid < RP::THREADS; tid++) RP::template phase<2>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
            }
            return 0;
        }
    }
    return -1;
}

// LDS-resident single pass of 2^13 / 2^14 (ntt_plan.hpp ResidentPlan)
template <class A, int LOGN>
int emu_resident(const PassArgs &a, int inverse)
{
    if constexpr (ResidentPlan<LOGN>::OK) {
        if (inverse) emu_pass<typename ResidentPass<A, LOGN, true>::Pass, LOGN, true, false>(a);
        else emu_pass<typename ResidentPass<A, LOGN, false>::Pass, LOGN, false, false>(a);
        return 0;
    }
    return -1;
}

template <class A, int LOGN>
void emu_dir(const PassArgs &a, int inverse)
{
    if (inverse) emu_transform<A, LOGN, true>(a);
    else emu_transform<A, LOGN, false>(a);
}

template <class A>
int emu_size(const PassArgs &a, int logn, int inverse)
{
    switch (logn) {
#define CASE(L) case L: emu_dir<A, L>(a, inverse); return 0;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10)
        CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20)
#undef CASE
    default: return -1;
    }
}

// the fused kernel's ticket schedule, run by ONE sequential "team": every wait is already
// satisfied when its ticket comes up, which is exactly the progress argument of ntt_fused.hpp
template <class PASS, int LOGN, bool INV, bool IS_COL>
void emu_tile(const PassArgs &a, u32 unit, u32 tile, std::vector<typename PASS::elem> &lds)
{
    typedef typename PASS::Arith A;
    u32 limb, row0 = 0;
    u64 *base;
    if constexpr (IS_COL) base = col_tile_of<PASS, LOGN>(unit, tile, a, limb);
    else base = row_tile_of<PASS, LOGN>(unit, tile, a, limb, row0);
    const LimbParams &p = a.lp[limb];
    auto ctx = A::make_ctx(p);
    const TwPtr tw = as_global(INV ? p.inv : p.fwd);
    for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<0>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
    if constexpr (PASS::NPHASE > 1)
        for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<1>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
    if constexpr (PASS::NPHASE > 2)
        for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<2>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
    if constexpr (PASS::NPHASE > 3)
        for (int tid = 0; tid < PASS::THREADS; tid++) PASS::template phase<3>(tid, base, lds.data(), tw, row0, ctx, p.inv_n);
}

template <class A, int LOGN, bool INV>
int emu_fused(const PassArgs &a, u32 dist)
{
    typedef FusedPasses<A, LOGN, INV> FP;
    std::vector<typename A::elem> lds(FP::LDS_ELEMS);
    // teams run one after the other here; on the GPU they run concurrently and independently
    for (u32 x = 0; x < FUSED_TEAMS; x++) {
        const u32 my_limbs = fused_team_limbs(a.units, x), last_group = my_limbs + dist;
        std::vector<u32> done(my_limbs + 1, 0);
        for (u32 t = 0;; t++) {
            const FusedTicket k = fused_decode(t, FP::T1, FP::T2, dist);
            if (k.group >= last_group) break;
            if (k.phase == 0 || k.slot >= my_limbs) continue;
            const u32 unit = x + FUSED_TEAMS * k.slot;
            if (k.phase == 1) {
                if constexpr (INV) emu_tile<typename FP::Row, LOGN, INV, false>(a, unit, k.tile, lds);
                else emu_tile<typename FP::Col, LOGN, INV, true>(a, unit, k.tile, lds);
                done[k.slot]++;
            } else {
                if (done[k.slot] != FP::T1) return -4;   // a wait that would block: schedule bug
                if constexpr (INV) emu_tile<typename FP::Col, LOGN, INV, true>(a, unit, k.tile, lds);
                else emu_tile<typename FP::Row, LOGN, INV, false>(a, unit, k.tile, lds);
            }
        }
    }
    return 0;
}

template <class A>
int emu_fused_size(const PassArgs &a, int logn, int inverse, u32 dist)
{
    switch (logn) {
#define CASE(L) case L: return inverse ? emu_fused<A, L, true>(a, dist) : emu_fused<A, L, false>(a, dist);
        CASE(13) CASE(14) CASE(15) CASE(16) CASE(17)
#undef CASE
    default: return -1;
    }
}

u64 invmod(u64 a, u64 m)
{
    __int128 t = 0, nt = 1, r = m, nr = a % m;
    while (nr != 0) {
        __int128 q = r / nr, tmp;
        tmp = t - q * nt; t = nt; nt = tmp;
        tmp = r - q * nr; r = nr; nr = tmp;
    }
    if (t < 0) t += m;
    return (u64)t;
}

} // namespace

// data: [n_poly][limbs][N] in place.  q: limbs moduli.  rp: limbs x N forward tables
// (canonical residues, entry k = psi^bitrev(k)).  path: 0 = ArithF64, 1 = ArithU64.
extern "C" int emu_ntt(u64 *data, int logn, int inverse, int n_poly, int limbs, const u64 *q, const u64 *rp, int path,
                       int fused_dist)
{
    const size_t N = (size_t)1 << logn;
    std::vector<LimbParams> lp(limbs);
    std::vector<std::vector<Tw>> fwd(limbs), inv(limbs);
    for (int l = 0; l < limbs; l++) {
        fwd[l].resize(N);
        inv[l].resize(N);
        for (size_t k = 0; k < N; k++) {
            u64 w = rp[(size_t)l * N + k], wi = invmod(w, q[l]);
            const u32 at = tw_stored_index(logn, (u32)k);   // same layout the library uploads
            fwd[l][at] = path == PATH_F64 ? ArithF64::encode(w, q[l]) : ArithU64::encode(w, q[l]);
            inv[l][at] = path == PATH_F64 ? ArithF64::encode(wi, q[l]) : ArithU64::encode(wi, q[l]);
        }
        {   // inverse entry 0: N^-1 times the last inverse stage's twiddle (as capi.cpp build_tables lays it out)
            const u64 ni0 = invmod(N % q[l], q[l]), w1 = invmod(rp[(size_t)l * N + (N > 1 ? 1 : 0)], q[l]);
            const u64 wn = (u64)((unsigned __int128)ni0 * w1 % q[l]);
            inv[l][0] = path == PATH_F64 ? ArithF64::encode(wn, q[l]) : ArithU64::encode(wn, q[l]);
        }
        LimbParams &p = lp[l];
        p.q = q[l];
        p.two_q = 2 * q[l];
        p.n = (double)q[l];
        p.ninv = 1.0 / p.n;
        u64 ni = invmod(N % q[l], q[l]);
        p.inv_n = path == PATH_F64 ? ArithF64::encode(ni, q[l]) : ArithU64::encode(ni, q[l]);
        p.fwd = fwd[l].data();
        p.inv = inv[l].data();
        p.path = path;
    }
    PassArgs a{data, lp.data(), 0u, (u32)limbs, (u32)(n_poly * limbs), (u32)limbs};
    // fused_dist == -1: the same batch addressed through an explicit unit list (PassArgs::map), in reverse order
    std::vector<UnitRef> map;
    if (fused_dist == -1) {
        for (int u = n_poly * limbs - 1; u >= 0; u--) map.push_back(UnitRef{(u32)u, (u32)(u % limbs)});
        a.map = map.data();
    }
    g_max_ratio = 0.0;
    if (fused_dist == -3) {          // forward transform with the packed hand-off (2^16, FP64 path)
        if (logn == 16 && path == PATH_F64 && !inverse) return emu_packed<ArithF64, 16>(a);
        return -1;
    }
    if (fused_dist == -2) {          // resident pass
        if (logn == 13) return path == PATH_F64 ? emu_resident<ArithF64, 13>(a, inverse) : emu_resident<ArithU64, 13>(a, inverse);
        if (logn == 14) return path == PATH_F64 ? emu_resident<ArithF64, 14>(a, inverse) : emu_resident<ArithU64, 14>(a, inverse);
        return -1;
    }
    if (fused_dist > 0)
        return path == PATH_F64 ? emu_fused_size<ArithF64>(a, logn, inverse, (u32)fused_dist)
                                : emu_fused_size<ArithU64>(a, logn, inverse, (u32)fused_dist);
    return path == PATH_F64 ? emu_size<ArithF64>(a, logn, inverse) : emu_size<ArithU64>(a, logn, inverse);
}

extern "C" double emu_max_ratio() { return g_max_ratio; }

// the per-register lazy-range plan of K inverse stages (ntt_core.hpp inv_lazy_plan), as the kernels' templates evaluate it
template <int K> static void dump_plan(int in8, int exit8, int fold, uint32_t *before, uint32_t *at_exit, int *out8)
{
    const InvLazyPlan<K> p = inv_lazy_plan<K>(in8, exit8, fold != 0);
    for (int v = 0; v < K; v++) before[v] = p.before[v];
    *at_exit = p.at_exit;
    *out8 = p.out8;
}
extern "C" int emu_inv_lazy_plan(int K, int in8, int exit8, int fold, uint32_t *before, uint32_t *at_exit, int *out8)
{
    switch (K) {
    case 1: dump_plan<1>(in8, exit8, fold, before, at_exit, out8); return 0;
    case 2: dump_plan<2>(in8, exit8, fold, before, at_exit, out8); return 0;
    case 3: dump_plan<3>(in8, exit8, fold, before, at_exit, out8); return 0;
    case 4: dump_plan<4>(in8, exit8, fold, before, at_exit, out8); return 0;
    case 5: dump_plan<5>(in8, exit8, fold, before, at_exit, out8); return 0;
    default: return 1;
    }
}
// entry bound of the second launch of a two-launch inverse and of every step, as Passes<> derives them (2^16: Steps<4,4> twice)
extern "C" int emu_inv_lazy_chain(int k0, int k1, int k2, int in8, int se)
{
    if (k2) return k1 == 3 ? inv_lazy_step_in8<Steps<3, 3, 3>>(in8, se) : inv_lazy_step_in8<Steps<4, 4, 4>>(in8, se);
    if (k0 == 4 && k1 == 4) return inv_lazy_step_in8<Steps<4, 4, 0>>(in8, se);
    if (k0 == 4 && k1 == 3) return inv_lazy_step_in8<Steps<4, 3, 0>>(in8, se);
    return -1;
}
