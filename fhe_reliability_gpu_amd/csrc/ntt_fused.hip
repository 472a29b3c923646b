// ntt_fused.hip -- fused forward / inverse NTT for two-pass sizes: persistent workgroups,
// static XCD teams, per-team ticket queues, first-pass -> second-pass hand-off through the
// team's L2, plus the completeness fix-up launch (protocol in ntt_fused.hpp).
#include "ntt_fused.hpp"
#include "ntt_launch.hpp"

namespace fhe {

__device__ __forceinline__ u32 ld_ctl(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ctl(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 add_ctl(u32 *p, u32 v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifndef FUSED_WAVES_PER_SIMD
#define FUSED_WAVES_PER_SIMD 3
#endif
constexpr u32 FUSED_SPIN_LIMIT = 1u << 22;   // x ~0.3 us per poll: seconds, far beyond any legitimate wait

template <class PASS, int LOGN, bool INV, bool IS_COL>
__device__ __forceinline__ void run_tile(const PassArgs &a, u32 unit, u32 tile, typename PASS::elem *lds)
{
    typedef typename PASS::Arith A;
    u32 limb, row0 = 0;
    u64 *base;
    if constexpr (IS_COL) base = col_tile_of<PASS, LOGN>(unit, tile, a, limb);
    else base = row_tile_of<PASS, LOGN>(unit, tile, a, limb, row0);
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

template <class FP, int LOGN, bool INV>
__device__ __forceinline__ void first_pass_tile(const PassArgs &a, u32 unit, u32 tile, typename FP::Col::elem *lds)
{
    if constexpr (INV) run_tile<typename FP::Row, LOGN, INV, false>(a, unit, tile, lds);
    else run_tile<typename FP::Col, LOGN, INV, true>(a, unit, tile, lds);
}
template <class FP, int LOGN, bool INV>
__device__ __forceinline__ void second_pass_tile(const PassArgs &a, u32 unit, u32 tile, typename FP::Col::elem *lds)
{
    if constexpr (INV) run_tile<typename FP::Col, LOGN, INV, true>(a, unit, tile, lds);
    else run_tile<typename FP::Row, LOGN, INV, false>(a, unit, tile, lds);
}

template <class A, int LOGN, bool INV, int HANDOFF, bool STREAM, bool FAT>
__global__ __launch_bounds__((FusedPasses<A, LOGN, INV, HANDOFF, STREAM, FAT>::NT), (FAT ? 4 : FUSED_WAVES_PER_SIMD)) void k_ntt_fused(FusedArgs f)
{
    typedef FusedPasses<A, LOGN, INV, HANDOFF, STREAM, FAT> FP;
    __shared__ __attribute__((aligned(16))) typename A::elem lds[FP::LDS_ELEMS];
    __shared__ u32 sh[8];

    u32 xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= FUSED_TEAMS - 1;
    if ((f.skip_teams >> xcc) & 1u) return;
    u32 *ctl = f.ctl;
    u32 *err = ctl + fused_error_word();
    u32 *ticket = ctl + fused_ticket_word(xcc);
    u32 *done = ctl + fused_base() + xcc * f.maxg;
    u32 *fin = ctl + fused_base() + FUSED_TEAMS * f.maxg;
    const u32 units = f.pa.units;
    const u32 my_limbs = fused_team_limbs(units, xcc);       // limbs x, x+8, ... of this team
    const u32 last_group = my_limbs + f.dist;                 // first group with nothing left
    constexpr u32 NONE = 0xFFFFFFFFu;

    // lane 0 draws tickets one ahead: the atomic for the next ticket is in flight while this tile runs
    u32 t_next = 0;
    if (threadIdx.x == 0) t_next = add_ctl(ticket, 1u);

    for (;;) {
        if (threadIdx.x == 0) {
            const u32 t = t_next;
            const FusedTicket k = fused_decode(t, FP::T1, FP::T2, f.dist);
            u32 unit = NONE, stop = 0;
            if (k.group >= last_group) {
                stop = 1;
            } else {
                if (k.phase != 0 && k.slot < my_limbs) {
                    unit = xcc + FUSED_TEAMS * k.slot;
                    if (k.phase == 2) {
                        // all first-pass tiles of this limb must have been stored and drained
                        bool ready = false;
                        for (u32 i = 0; i < FUSED_SPIN_LIMIT && !ready; i++) {
                            ready = ld_ctl(done + k.slot) == FP::T1;
                            if (!ready) __builtin_amdgcn_s_sleep(2);
                        }
                        if (!ready) {
                            st_ctl(err, 1u);
                            stop = 1;
                        }
                    }
                }
                // draw the next ticket now: its round trip overlaps this tile (issued after the poll,
                // because memory operations of a wave return in order)
                t_next = add_ctl(ticket, 1u);
            }
            sh[0] = (u32)k.phase;
            sh[1] = unit;
            sh[2] = k.tile;
            sh[3] = stop;
            sh[4] = k.slot;
        }
        __syncthreads();
        // the decoded ticket is the same for every lane: keep it in SGPRs so that limb parameters and
        // twiddle base pointers become scalar values
        const u32 phase = __builtin_amdgcn_readfirstlane(sh[0]), unit = __builtin_amdgcn_readfirstlane(sh[1]),
                  tile = __builtin_amdgcn_readfirstlane(sh[2]), stop = __builtin_amdgcn_readfirstlane(sh[3]),
                  slot = __builtin_amdgcn_readfirstlane(sh[4]);
        if (stop) break;
        if (unit != NONE) {
            if (phase == 1) {
                first_pass_tile<FP, LOGN, INV>(f.pa, unit, tile, lds);
                // publish: every storing wave drains its stores to L2, the workgroup meets, one lane counts the tile
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (threadIdx.x == 0) add_ctl(done + slot, 1u);
            } else {
                if constexpr (HANDOFF == HANDOFF_ACQUIRE) {
                    // drop this CU's L1 so plain loads are served by the team's L2
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                second_pass_tile<FP, LOGN, INV>(f.pa, unit, tile, lds);
                if (threadIdx.x == 0) add_ctl(fin + unit, 1u);
            }
        }
        __syncthreads();
    }
}

// Completeness fix-up: any limb without a single finished second-pass tile was owned by a team
// that has no workgroup (an XCD the dispatcher did not use).  Transform it inside one workgroup:
// all first-pass tiles, drain + barrier + L1 invalidate, then all second-pass tiles.
template <class A, int LOGN, bool INV>
__global__ __launch_bounds__(NTT_THREADS, FUSED_WAVES_PER_SIMD) void k_ntt_fixup(FusedArgs f)
{
    typedef FusedPasses<A, LOGN, INV, HANDOFF_ACQUIRE, false> FP;
    __shared__ __attribute__((aligned(16))) typename A::elem lds[FP::LDS_ELEMS];
    u32 *ctl = f.ctl;
    const u32 *fin = ctl + fused_base() + FUSED_TEAMS * f.maxg;
    for (u32 unit = blockIdx.x; unit < f.pa.units; unit += gridDim.x) {
        const u32 n = __builtin_amdgcn_readfirstlane(ld_ctl(fin + unit));
        if (n == FP::T2) continue;
        if (n != 0) {                                   // partially transformed: cannot happen, report it
            if (threadIdx.x == 0) st_ctl(ctl + fused_error_word(), 2u);
            continue;
        }
        for (u32 tile = 0; tile < FP::T1; tile++) {
            first_pass_tile<FP, LOGN, INV>(f.pa, unit, tile, lds);
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (u32 tile = 0; tile < FP::T2; tile++) {
            second_pass_tile<FP, LOGN, INV>(f.pa, unit, tile, lds);
            __syncthreads();
        }
    }
}

template <class A, int LOGN, bool INV, int HANDOFF, bool STREAM, bool FAT = false>
static hipError_t launch_variant(hipStream_t st, const FusedArgs &f, u32 wgs)
{
    typedef FusedPasses<A, LOGN, INV, HANDOFF, STREAM, FAT> FP;
    const u64 tickets = ((u64)f.pa.units + (u64)FUSED_TEAMS * f.dist) * (FP::T1 + FP::T2);
    const u32 grid = (u32)(tickets < wgs ? tickets : wgs);
    hipLaunchKernelGGL((k_ntt_fused<A, LOGN, INV, HANDOFF, STREAM, FAT>), dim3(grid), dim3(FP::NT), 0, st, f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const u32 fix = f.pa.units < 64 ? f.pa.units : 64;
    hipLaunchKernelGGL((k_ntt_fixup<A, LOGN, INV>), dim3(fix), dim3(NTT_THREADS), 0, st, f);
    return hipGetLastError();
}

// variant = handoff * 2 + stream; the full matrix exists for 2^16 (tuning), other sizes use the default
template <class A, int LOGN, bool INV>
static hipError_t launch_fused_t(hipStream_t st, const FusedArgs &f, u32 wgs, int variant)
{
    if constexpr (LOGN == 16) {
        switch (variant) {
        case HANDOFF_SC1 * 2: return launch_variant<A, LOGN, INV, HANDOFF_SC1, false>(st, f, wgs);
        case HANDOFF_SC1 * 2 + 1: return launch_variant<A, LOGN, INV, HANDOFF_SC1, true>(st, f, wgs);
        case HANDOFF_NT * 2: return launch_variant<A, LOGN, INV, HANDOFF_NT, false>(st, f, wgs);
        case HANDOFF_NT * 2 + 1: return launch_variant<A, LOGN, INV, HANDOFF_NT, true>(st, f, wgs);
        case HANDOFF_ACQUIRE * 2: return launch_variant<A, LOGN, INV, HANDOFF_ACQUIRE, false>(st, f, wgs);
        case 8 + HANDOFF_SC1 * 2: return launch_variant<A, LOGN, INV, HANDOFF_SC1, false, true>(st, f, wgs);       // fat tiles
        case 8 + HANDOFF_NT * 2: return launch_variant<A, LOGN, INV, HANDOFF_NT, false, true>(st, f, wgs);
        case 8 + HANDOFF_ACQUIRE * 2: return launch_variant<A, LOGN, INV, HANDOFF_ACQUIRE, false, true>(st, f, wgs);
        default: return launch_variant<A, LOGN, INV, HANDOFF_ACQUIRE, true>(st, f, wgs);
        }
    } else {
        return launch_variant<A, LOGN, INV, HANDOFF_ACQUIRE, true>(st, f, wgs);
    }
}

template <class A>
static hipError_t launch_fused_size(hipStream_t st, const FusedArgs &f, int logn, bool inverse, u32 wgs, int variant)
{
    switch (logn) {
#define FHE_CASE(L) \
    case L:         \
        return inverse ? launch_fused_t<A, L, true>(st, f, wgs, variant) : launch_fused_t<A, L, false>(st, f, wgs, variant);
        FHE_CASE(13) FHE_CASE(14) FHE_CASE(15) FHE_CASE(16) FHE_CASE(17)
#undef FHE_CASE
    default:
        return hipErrorInvalidValue;
    }
}

bool fused_supported(int logn) { return logn >= 13 && logn <= 17; }

size_t fused_ctl_bytes(u32 units) { return fused_ctl_words(units) * sizeof(u32); }

// ctl must hold fused_ctl_bytes(units) bytes; its counters are zeroed on the stream first.  The error word is NOT: it stays
// set from the launch whose bounded wait ran out until the host reads and clears it (fhe_ctx_check), so a later launch on
// the same stream cannot hide an earlier failure.  (The owner zeroes the whole block once when it allocates it.)
hipError_t launch_ntt_fused(hipStream_t st, const PassArgs &a, int logn, bool inverse, int path, u32 *ctl, u32 dist, u32 wgs,
                            int variant, u32 skip_teams)
{
    if (a.units == 0) return hipSuccess;
    if (!fused_supported(logn) || dist < 1) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(ctl, 0, (size_t)fused_error_word() * sizeof(u32), st);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ctl + fused_base(), 0, fused_ctl_bytes(a.units) - (size_t)fused_base() * sizeof(u32), st);
    if (e != hipSuccess) return e;
    FusedArgs f{a, ctl, fused_maxg(a.units), dist, skip_teams};
    return path == PATH_F64 ? launch_fused_size<ArithF64>(st, f, logn, inverse, wgs, variant)
                            : launch_fused_size<ArithU64>(st, f, logn, inverse, wgs, variant);
}

} // namespace fhe
