// ntt_fused.hpp -- schedule of the fused (single main launch) two-pass NTT.
//
// Why: two separate launches move every residue through the fabric twice per pass
// (measured: FETCH+WRITE = 512 MiB for a 128 MiB batch), which caps a 2^16 transform
// near 40 % of the 8 TB/s roofline whatever the ALU does.  The fused kernel hands a
// limb from its first pass to its second pass inside ONE XCD: the workgroups that run
// on XCD x (read from HW_REG_XCC_ID at run time, never inferred from blockIdx) form
// team x, and team x processes the limbs u = x, x+8, x+16, ... -- first-pass tiles,
// and D limbs later the second-pass tiles, so the second pass can be served by the L2
// the first pass wrote through.
//
// Schedule (per team): tickets drawn with one atomic add each (prefetched one ahead).
//   group g = ticket / (T1 + T2);   r = ticket % (T1 + T2)
//   r <  T1 : first-pass tile r of the team's limb number g
//   r >= T1 : second-pass tile r - T1 of the team's limb number g - D
// A team stops at group (its limb count) + D.
//
// Progress: a second-pass ticket waits only for first-pass tickets of the same limb,
// which have smaller ticket numbers and wait for nothing, so the lowest unfinished
// ticket can always run; no co-residency assumption, every spin is bounded.
//
// Completeness: correctness must not depend on where the dispatcher places workgroups.
// Every finished second-pass tile is counted per limb; a tiny follow-up launch
// (k_ntt_fixup) transforms any limb whose count is still zero -- which happens only if
// an XCD received no workgroup at all -- inside a single workgroup.
//
// Visibility (producer and consumer share an XCD by construction): first-pass stores
// are plain (write-through L1, line stays in the team's L2), each storing wave drains
// them (s_waitcnt vmcnt(0)), the workgroup barriers, one lane adds to the limb's done
// counter (agent-scope atomic).  The consumer polls the counter with coherent loads,
// then reads the limb with loads that cannot be served by its CU's L1 (flavour chosen
// by HANDOFF: sc1 loads, nt loads, or an agent-scope acquire followed by plain loads).
#pragma once
#include "ntt_plan.hpp"

namespace fhe {

constexpr u32 FUSED_TEAMS = 8;          // XCDs of an MI355X; HW_REG_XCC_ID & 7
constexpr u32 FUSED_LINE = 32;          // u32 words per 128-byte line

enum { HANDOFF_SC1 = 1, HANDOFF_NT = 2, HANDOFF_ACQUIRE = 3 };

// Control block layout (u32 words); the counters are zeroed before every launch, the error flag only by the host:
//   [LINE*x]                 ticket counter of team x
//   [LINE*TEAMS]             error flag (a bounded spin ran out / inconsistent state), sticky until fhe_ctx_check reads it
//   done[x][g] : finished first-pass tiles of team x's limb g   at  base + x*maxg + g
//   fin[u]     : finished second-pass tiles of limb u           at  base + TEAMS*maxg + u
FHE_HD constexpr u32 fused_ticket_word(u32 x) { return FUSED_LINE * x; }
FHE_HD constexpr u32 fused_error_word() { return FUSED_LINE * FUSED_TEAMS; }
FHE_HD constexpr u32 fused_base() { return FUSED_LINE * (FUSED_TEAMS + 1); }
FHE_HD u32 fused_maxg(u32 units) { return (units + FUSED_TEAMS - 1) / FUSED_TEAMS + 1; }
FHE_HD size_t fused_ctl_words(u32 units) { return fused_base() + (size_t)FUSED_TEAMS * fused_maxg(units) + units; }

struct FusedArgs {
    PassArgs pa;
    u32 *ctl;
    u32 maxg;
    u32 dist;      // D
    u32 skip_teams; // test hook: bit x set = team x behaves as if its XCD had received no workgroup
};

struct FusedTicket {
    u32 group, slot, tile;
    int phase;     // 1 or 2; 0 = no work for this ticket
};

FHE_HD FusedTicket fused_decode(u32 ticket, u32 t1, u32 t2, u32 dist)
{
    FusedTicket k;
    const u32 G = t1 + t2;
    k.group = ticket / G;
    const u32 r = ticket % G;
    if (r < t1) {
        k.phase = 1;
        k.slot = k.group;
        k.tile = r;
    } else {
        k.tile = r - t1;
        if (k.group >= dist) {
            k.phase = 2;
            k.slot = k.group - dist;
        } else {
            k.phase = 0;
            k.slot = 0;
        }
    }
    return k;
}

// limbs of team x: u = x + 8 g, g = 0 .. count-1
FHE_HD u32 fused_team_limbs(u32 units, u32 x) { return units > x ? (units - x + FUSED_TEAMS - 1) / FUSED_TEAMS : 0u; }

// Geometry of the fused kernel: smaller tiles than the two-launch path so that three to
// four workgroups fit one CU's LDS and register file.
template <int LOGN> struct FusedGeom {
    typedef Plan<LOGN> PL;
    static constexpr int PC = PL::Col::P, PR = PL::Row::P;
    static constexpr int TC = PC <= 8 ? 16 : 8;
    static constexpr int TR = PR <= 8 ? 16 : PR == 9 ? 8 : 4;
};

// FAT: 512 threads per tile and radix-8 register steps (8 points per thread) at 2^16 -- twice the
// waves per byte of in-flight tile, i.e. half the L2 footprint for the same number of resident waves.
template <int LOGN, bool FAT> struct FusedSteps {
    typedef typename Plan<LOGN>::Col Col;
    typedef typename Plan<LOGN>::Row Row;
    static constexpr int NT = NTT_THREADS;
};
template <> struct FusedSteps<16, true> {
    typedef Steps<3, 3, 2> Col;
    typedef Steps<3, 3, 2> Row;
    static constexpr int NT = 512;
};

template <class A, int LOGN, bool INVERSE, int HANDOFF = HANDOFF_SC1, bool STREAM = false, bool FAT = false>
struct FusedPasses {
    typedef FusedGeom<LOGN> G;
    struct PL {
        typedef typename FusedSteps<LOGN, FAT>::Col Col;
        typedef typename FusedSteps<LOGN, FAT>::Row Row;
    };
    static constexpr int NT = FusedSteps<LOGN, FAT>::NT;
    static constexpr u32 RED_FIRST = INVERSE ? reduce_mask(0, G::PR, A::INV_FIRST, A::INV_NEXT)
                                             : reduce_mask(0, G::PC, A::FWD_FIRST, A::FWD_NEXT);
    static constexpr u32 RED_SECOND = INVERSE ? reduce_mask(G::PR, G::PC, A::INV_FIRST, A::INV_NEXT)
                                              : reduce_mask(G::PC, G::PR, A::FWD_FIRST, A::FWD_NEXT);
    static constexpr int LOADS = HANDOFF == HANDOFF_ACQUIRE ? 0 : HANDOFF;   // how the second pass reads the hand-off
    typedef ColPass<A, typename PL::Col, LOGN, 0, G::TC, NT, INVERSE, INVERSE ? IO_LAZY : IO_CANONICAL,
                    INVERSE ? IO_CANONICAL : IO_LAZY, INVERSE ? RED_SECOND : RED_FIRST, BlkStage<LOGN>::value, INVERSE ? LOADS : 0, STREAM> Col;
    typedef RowPass<A, typename PL::Row, LOGN, G::TR, NT, INVERSE, INVERSE ? IO_CANONICAL : IO_LAZY,
                    INVERSE ? IO_LAZY : IO_CANONICAL, INVERSE ? RED_FIRST : RED_SECOND, BlkStage<LOGN>::value, INVERSE ? 0 : LOADS, STREAM> Row;
    static constexpr u32 T1 = INVERSE ? Row::TILES : Col::TILES;   // tiles of the pass that runs first
    static constexpr u32 T2 = INVERSE ? Col::TILES : Row::TILES;
    static constexpr int LDS_ELEMS = cmax(Col::LDS_ELEMS, Row::LDS_ELEMS);
};

} // namespace fhe
