// capi_keyswitch.cpp -- Galois automorphisms, hybrid key switching, rotation (part of the C ABI of include/fhe_mi355x.h; shared pieces in capi_internal.hpp)
//
// One plan type serves a single device and a rank of a limb-sharded job (BASELINE configs 4-5: "RNS limbs sharded across
// 8 x MI355X + RCCL all-gather"): a rank OWNS a slab of the L ciphertext primes and a slab of the K special primes
// (ks_shard_layout), keeps its limbs of the input, of every key digit and of the result, and runs
//   begin  : INTT of the owned ciphertext limbs into its slot of gather buffer 1            -> all-gather 1 (host, RCCL)
//   inner  : per digit exact extension to the owned limbs, NTT, inner product with the key,
//            INTT of the owned special limbs into its slot of gather buffer 2                -> all-gather 2
//   finish : mod-down to the owned ciphertext limbs
// With world = 1 the owner has everything, the slots are the whole buffers and fhe_keyswitch_apply runs the three
// phases back to back with no collective.
#include "capi_internal.hpp"

void ks_shard_layout(int L, int K, int world, int rank, KsShard *s)
{
    s->world = world;
    s->rank = rank;
    const int cb = L / world, ce = L % world;
    s->clo = rank * cb + std::min(rank, ce);
    s->cn = cb + (rank < ce ? 1 : 0);
    s->cmax = cb + (ce ? 1 : 0);
    // special limbs are handed out from the last rank backwards: the ranks with the smaller ciphertext slabs get them first
    const int rr = world - 1 - rank, sb = K / world, se = K % world;
    s->slo = L + rr * sb + std::min(rr, se);
    s->sn = sb + (rr < se ? 1 : 0);
    s->smax = sb + (se ? 1 : 0);
}

// row of ciphertext limb l in gather buffer 1 ([world][cmax][N]) / of special limb k (0-based in P), half h, in gather buffer 2 ([world][2][smax][N])
static u32 g1_row(int L, int K, int world, int l)
{
    for (int r = 0; r < world; r++) {
        KsShard s;
        ks_shard_layout(L, K, world, r, &s);
        if (l >= s.clo && l < s.clo + s.cn) return (u32)(r * s.cmax + (l - s.clo));
    }
    return 0;
}
static u32 g2_row(int L, int K, int world, int k, int h)
{
    for (int r = 0; r < world; r++) {
        KsShard s;
        ks_shard_layout(L, K, world, r, &s);
        if (L + k >= s.slo && L + k < s.slo + s.sn) return (u32)((r * 2 + h) * s.smax + (L + k - s.slo));
    }
    return 0;
}

static int ks_create(fhe_ctx *ctx, const fhe_ntt_tables *t, int L, int K, int dnum, int world, int rank, uint64_t *d_gather1,
                     uint64_t *d_gather2, uint64_t *d_bcast, fhe_keyswitch **out)
{
    if (!ctx || !t || !out || L < 1 || K < 1 || dnum < 1 || dnum > L || L + K > t->count)
        return fail(FHE_ERR_INVALID, "bad key-switch shape");
    if (world < 1 || rank < 0 || rank >= world) return fail(FHE_ERR_INVALID, "bad world / rank");
    if (world > 1 && (!d_gather1 || !d_gather2)) return fail(FHE_ERR_INVALID, "a sharded plan needs the two gather buffers");
    const bool sharded = d_gather1 && d_gather2;      // world = 1 with buffers: the phase path with (trivial) gathers, as a 1-rank job runs it
    std::unique_ptr<fhe_keyswitch> p(new fhe_keyswitch);
    p->ctx = ctx;
    p->t = t;
    p->L = L;
    p->K = K;
    p->dnum = dnum;
    p->alpha = (L + dnum - 1) / dnum;
    p->log_n = t->log_n;
    p->sharded = sharded;
    ks_shard_layout(L, K, world, rank, &p->sh);
    const KsShard &sh = p->sh;
    const size_t N = (size_t)1 << t->log_n, MO = (size_t)sh.cn + sh.sn;   // owned limbs: rows of ext / acc / the key
    p->m_own = (int)MO;
    for (size_t jj = 0; jj < MO; jj++) p->own_path[t->path[jj < (size_t)sh.cn ? (size_t)sh.clo + jj : (size_t)sh.slo + (jj - sh.cn)]] = true;
    auto limb_of = [&](size_t jj) { return jj < (size_t)sh.cn ? (size_t)sh.clo + jj : (size_t)sh.slo + (jj - sh.cn); };
    HIP_TRY(hipSetDevice(ctx->device));
    // per digit: where the digit's limbs sit in gather buffer 1, which owned rows it skips, and the conversion to the rest
    std::vector<u32> up_rows((size_t)dnum * p->alpha, 0);
    std::vector<std::pair<u32, u32>> gaps(dnum);
    for (int d = 0; d < dnum; d++) {
        const int lo = d * p->alpha, hi = std::min(L, lo + p->alpha);
        if (lo >= hi) return fail(FHE_ERR_INVALID, "dnum leaves an empty digit");
        for (int l = lo; l < hi; l++) up_rows[(size_t)d * p->alpha + (l - lo)] = g1_row(L, K, world, l);
        const int ga = std::max(lo, sh.clo) - sh.clo, gb = std::min(hi, sh.clo + sh.cn) - sh.clo;     // owned rows inside the digit
        gaps[d] = gb > ga ? std::make_pair((u32)ga, (u32)(gb - ga)) : std::make_pair(0xFFFFFFFFu, 0u);
        std::vector<u64> in(t->q.begin() + lo, t->q.begin() + hi), other;
        for (size_t jj = 0; jj < MO; jj++) {
            const size_t tl = limb_of(jj);
            if ((int)tl < lo || (int)tl >= hi) other.push_back(t->q[tl]);
        }
        fhe_baseconv *bc = nullptr;
        if (!other.empty()) {
            int rc = fhe_baseconv_create(ctx, in.data(), (int)in.size(), other.data(), (int)other.size(), &bc);
            if (rc) return rc;
        }
        p->up.push_back(bc);
    }
    if (sh.cn > 0) {
        std::vector<u64> P(t->q.begin() + L, t->q.begin() + L + K), Q(t->q.begin() + sh.clo, t->q.begin() + sh.clo + sh.cn);
        int rc = fhe_baseconv_create(ctx, P.data(), K, Q.data(), sh.cn, &p->down);
        if (rc) return rc;
        std::vector<u64> pinv(sh.cn);
        for (int j = 0; j < sh.cn; j++) {
            u64 pm = 1 % Q[j];
            for (u64 pk : P) pm = host::mul_mod(pm, pk % Q[j], Q[j]);
            pinv[j] = host::inv_mod(pm, Q[j]);
            if (!pinv[j]) return fail(FHE_ERR_INVALID, "special primes must be coprime to the ciphertext primes");
        }
        HIP_TRY(p->pinv.upload(pinv));
    }
    {
        // P mod q_j as a twiddle of limb j's arithmetic, indexed by TABLE limb (ColAddSrc::w), every ciphertext limb
        std::vector<Tw> pq(L);
        for (int j = 0; j < L; j++) {
            const u64 qj = t->q[j];
            u64 pm = 1 % qj;
            for (int k = 0; k < K; k++) pm = host::mul_mod(pm, t->q[L + k] % qj, qj);
            pq[j] = t->path[j] == PATH_F64 ? ArithF64::encode(pm, qj) : ArithU64::encode(pm, qj);
        }
        HIP_TRY(p->pq_tw.upload(pq));
    }
    p->up_trivial = p->alpha == 1 && t->log_n >= 13;
    // every owned limb of every digit's extension except the digit's own limbs, one list per arithmetic path
    for (int path = 0; path < 2; path++) {
        std::vector<UnitRef> map;
        for (int d = 0; d < dnum; d++) {
            const int lo = d * p->alpha, hi = std::min(L, lo + p->alpha);
            for (size_t jj = 0; jj < MO; jj++) {
                const size_t tl = limb_of(jj);
                if (((int)tl < lo || (int)tl >= hi) && t->path[tl] == path)
                    map.push_back(UnitRef{(u32)(d * MO + jj), (u32)tl, p->up_trivial ? up_rows[(size_t)d * p->alpha] : 0xFFFFFFFFu});
            }
        }
        p->ext_units[path] = (u32)map.size();
        if (!map.empty()) HIP_TRY(p->ext_map[path].upload(map));
    }
    if (!sharded) {
        HIP_TRY(p->coef.alloc((size_t)L * N * 8));
        p->g1 = p->coef.as<u64>();
        p->g2 = nullptr;   // the special limbs are converted where they stand, inside acc
    } else {
        p->g1 = d_gather1;
        p->g2 = d_gather2;
    }
    HIP_TRY(p->ext.alloc((size_t)dnum * MO * N * 8));
    HIP_TRY(p->acc.alloc(2 * MO * N * 8));
    HIP_TRY(p->conv.alloc(2 * (size_t)sh.cn * N * 8));
    HIP_TRY(p->rot.alloc(3 * (size_t)sh.cn * N * 8));
    HIP_TRY(p->up_rows.upload(up_rows));
    {
        std::vector<u32> down_rows((size_t)2 * K);
        for (int h = 0; h < 2; h++)
            for (int k = 0; k < K; k++) down_rows[(size_t)h * K + k] = !sharded ? (u32)(h * MO + L + k) : g2_row(L, K, world, k, h);
        HIP_TRY(p->down_rows.upload(down_rows));
        p->down_src_row = down_rows[0];
        p->down_src_stride = down_rows[K] - down_rows[0];
        std::vector<BcJob> up, down;
        p->up_batched = true;
        int first = -1;
        for (int d = 0; d < dnum; d++) {
            if (!p->up[d]) continue;
            const BaseConvPlanDev &pl = p->up[d]->dev;
            if (first < 0) first = d;
            up.push_back(BcJob{pl, p->g1, p->ext.as<u64>() + (size_t)d * MO * N, gaps[d].first, gaps[d].second, p->up_rows.as<u32>() + (size_t)d * p->alpha});
            p->up_max_m = std::max(p->up_max_m, pl.m);
            if (pl.m <= 16) p->up_m_mask |= 1u << (pl.m - 1);
            p->up_max_k = std::max(p->up_max_k, pl.k);
            p->up_batched = p->up_batched && pl.f64 == p->up[first]->dev.f64;
        }
        p->up_f64 = first >= 0 && p->up[first]->dev.f64 != 0;
        p->n_up_jobs = (u32)up.size();
        if (p->down)
            for (int h = 0; h < 2; h++)
                down.push_back(BcJob{p->down->dev, !sharded ? p->acc.as<u64>() : p->g2, p->conv.as<u64>() + (size_t)h * sh.cn * N, 0xFFFFFFFFu, 0u,
                                     p->down_rows.as<u32>() + (size_t)h * K});
        if (!up.empty()) HIP_TRY(p->up_jobs.upload(up));
        if (!down.empty()) HIP_TRY(p->down_jobs.upload(down));
        p->up_host = up;
    }
    if (!sharded) {
        HIP_TRY(p->hm.alloc(3 * (size_t)L * N * 8));
        HIP_TRY(p->hm_pre.alloc(2 * (size_t)L * N * 8));
    }
    if (L >= 2 && (!sharded || d_bcast)) {
        // rescale / mod-switch to the next level: drop q_{L-1}.  Its owner turns the last limb of every part to coefficient
        // form; every rank needs it (one device: it is simply there; sharded: ONE broadcast of n_parts x N words into d_bcast)
        // and forms (c - delta) / q_last on the ciphertext limbs below L-1 that it owns.
        const u64 ql = t->q[L - 1];
        p->own_last = sh.clo <= L - 1 && L - 1 < sh.clo + sh.cn;
        p->rs_n = std::max(0, std::min(sh.clo + sh.cn, L - 1) - sh.clo);
        if (!sharded) {
            HIP_TRY(p->rs_last.alloc(3 * N * 8));
            p->rs_bc = p->rs_last.as<u64>();
        } else {
            p->rs_bc = d_bcast;
        }
        if (p->rs_n > 0) {
            int rc = fhe_baseconv_create(ctx, &ql, 1, t->q.data() + sh.clo, p->rs_n, &p->last);
            if (rc) return rc;
            std::vector<u64> qinv(p->rs_n);
            for (int j = 0; j < p->rs_n; j++) {
                const u64 qj = t->q[sh.clo + j];
                qinv[j] = host::inv_mod(ql % qj, qj);
                if (!qinv[j]) return fail(FHE_ERR_INVALID, "ciphertext primes must be pairwise coprime");
            }
            HIP_TRY(p->qlast_inv.upload(qinv));
            HIP_TRY(p->rs_delta.alloc(3 * (size_t)p->rs_n * N * 8));
            std::vector<BcJob> jobs;
            for (int part = 0; part < 3; part++)
                jobs.push_back(BcJob{p->last->dev, p->rs_bc + (size_t)part * N, p->rs_delta.as<u64>() + (size_t)part * p->rs_n * N, 0xFFFFFFFFu, 0u});
            HIP_TRY(p->rs_jobs.upload(jobs));
        }
    }
    *out = p.release();
    return FHE_OK;
}

// ---------------------------------------------------------------- the phases
// INTT of the owned ciphertext limbs (the "3 INTT" that open KEYSWITCH in the L = 4 trace, 16384_4:468-470) into this
// rank's slot of gather buffer 1
// galois != 0 (a rotation): the input is sigma_k(d_c), taken on the INTT's load; d_sig receives sigma_k(d_c) itself for the inner product
static int ks_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c, hipStream_t st, u32 galois = 0, u64 *d_sig = nullptr)
{
    const KsShard &sh = p->sh;
    if (!sh.cn) return FHE_OK;
    const size_t N = (size_t)1 << p->log_n;
    u64 *slot = p->g1 + (size_t)sh.rank * sh.cmax * N;
    return ntt_batch(ctx, slot, p->t, 1, sh.cn, sh.clo, st, true, d_c, galois, d_sig);     // out of place: the input is the caller's, no copy
}

// base extension of each digit to every other owned prime (MODREDUCTION, 16384_4:471-452), their transforms, and the inner
// product with the key (MULTEVK)
// with the fused inner product only the FIRST launch of the extended limbs' forward transform runs on its own (the column
// pass; nothing at all for single-pass sizes): the row pass happens inside the MULTEVK launch, tile by tile
// (by shape: the fused launch has m_own x tiles workgroups, each looping over the digits -- it pays once that fills the chip)
static bool ks_use_fused(const fhe_ctx *ctx, const fhe_keyswitch *p)
{
    const size_t MO = p->m_own;
    const bool want = ctx->ks_fused < 0 ? (MO << (p->log_n > 12 ? p->log_n - 12 : 0)) >= 640 : ctx->ks_fused != 0;
    return want && ks_rowmac_supported(p->log_n) && ctx->fault_idx < 0;
}

// Hoisted rotations share the extended digits: with two or more Galois elements the digits' row passes run ONCE (plain forward
// transform of the extensions) and every element costs a word-wise inner product that streams digits and key at the fabric's rate,
// instead of the fused launch repeating all row passes per element (config 5, eight elements: 201 -> 175 us per rotation).  A batch of
// one, and an explicit ks_fused setting, keep ks_use_fused()'s choice; the sharded phases do not know the batch size and take the
// shared-row-pass form.
static bool ks_hoist_fused(const fhe_ctx *ctx, const fhe_keyswitch *p, size_t n_rot)
{
    if (ctx->ks_fused < 0 && n_rot != 1) return false;
    return ks_use_fused(ctx, p);
}

// the digits' extensions and the part of their forward transform that does not ride on the inner product (everything a hoisted
// rotation shares between its Galois elements)
static int ks_extend(fhe_ctx *ctx, fhe_keyswitch *p, hipStream_t st, bool fused)
{
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *ext = p->ext.as<u64>();
    hipError_t e;
    if (!MO) return FHE_OK;
    {
        // (one trace line for the phase; the nested NTT line precedes it, as the reference's tools expect of nested costs)
        TraceScope tr_mr(ctx, st, "MODREDUCTION");
        const bool trivial = p->up_trivial && ctx->mode == 0;
        if (trivial) {
            // one-limb digits: nothing to launch, the column pass below reads the digit's limb and reduces on the load
        } else if (p->up_batched) {
            e = launch_baseconv_exact_jobs(st, p->up_jobs.as<BcJob>(), p->n_up_jobs, p->up_max_m, p->up_max_k, p->up_f64, N, p->up_max_m <= 16 ? p->up_m_mask : 0u);
            if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact_jobs");
        } else {
            for (const BcJob &j : p->up_host) {
                e = launch_baseconv_exact(st, j.out, j.in, j.pl, N, j.gap_at, j.gap, j.in_rows);
                if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact");
            }
        }
        TraceScope tr_ntt(ctx, st, "NTT");
        for (int path = 0; path < 2; path++) {
            if (!p->ext_units[path] || (fused && p->log_n <= 12)) continue;
            PassArgs a{ext, lp, 0u, 1u, p->ext_units[path], 1u, p->ext_map[path].as<UnitRef>()};
            if (trivial) a.src = p->g1;
            if ((e = launch_ntt(st, a, p->log_n, false, path, 1, fused ? 0 : -1)) != hipSuccess) return hip_fail(e, "launch_ntt");
        }
    }
    return FHE_OK;
}

// multiply-accumulate with the evaluation key (MULTEVK): all digits, both halves, one launch
static int ks_mac(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c, const uint64_t *d_evk, hipStream_t st, bool fused)
{
    const fhe_ntt_tables *t = p->t;
    const KsShard &sh = p->sh;
    const size_t MO = p->m_own;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *ext = p->ext.as<u64>(), *acc = p->acc_cur();
    hipError_t e;
    if (!MO) return FHE_OK;
    TraceScope tr_mk(ctx, st, "MULTEVK");
    const KsMacArgs ka{acc, ext, d_c, d_evk, lp, (u32)p->L, (u32)MO, (u32)p->dnum, (u32)p->alpha, p->log_n, (u32)sh.cn, (u32)sh.clo, (u32)(sh.slo - sh.cn)};
    e = fused ? launch_ks_rowmac(st, ka, p->own_path[0], p->own_path[1]) : launch_ks_mac(st, ka);
    if (e != hipSuccess) return hip_fail(e, "launch_ks_mac");
    return FHE_OK;
}

static int ks_extend_mac(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c, const uint64_t *d_evk, hipStream_t st)
{
    const bool fused = ks_use_fused(ctx, p);
    int rc = ks_extend(ctx, p, st, fused);
    return rc ? rc : ks_mac(ctx, p, d_c, d_evk, st, fused);
}

// opening of the mod-down (MODSWITCH, 16384_4:454-463): the owned special limbs of both halves to coefficient form -- in
// place inside acc ([2][MO][N]) on one device, in this rank's slot of gather buffer 2 ([2][smax][N]) when sharded
// galois != 0 (a hoisted rotation on a sharded plan): the sums are still in the un-rotated frame, sigma is taken on this INTT's load
// with_last (one device, ks_finish_rescale): the LAST ciphertext limb goes along -- its sums become acc P^-1 + add first (the limb
// after the mod-down, up to the converted part that is subtracted in coefficient form later) and row L-1 of acc ends in coefficient form
static int ks_special_intt(fhe_ctx *ctx, fhe_keyswitch *p, hipStream_t st, u32 galois = 0, bool with_last = false, const uint64_t *d_add0 = nullptr,
                           const uint64_t *d_add1 = nullptr)
{
    const fhe_ntt_tables *t = p->t;
    const KsShard &sh = p->sh;
    if (!sh.sn) return FHE_OK;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *acc = p->acc.as<u64>(), *sp = acc + (size_t)sh.cn * N;
    u32 stride = (u32)MO;
    int rc;
    size_t first = sh.slo, count = sh.sn;
    if (with_last) {
        const size_t R = (size_t)p->L - 1;
        const SubScaleArgs sa{acc + R * N, acc + (MO + R) * N, acc + R * N, nullptr, d_add0 ? d_add0 + R * N : nullptr, p->pinv.as<u64>() + R, (u64)(MO * N), 0, lp, (u32)R, 1u,
                              p->log_n, d_add1 ? d_add1 + R * N : nullptr};
        hipError_t e = launch_sub_scale(st, sa);
        if (e != hipSuccess) return hip_fail(e, "launch_sub_scale");
        sp -= N;
        first -= 1;
        count += 1;
    }
    if (galois && (!p->sharded || p->log_n < 5)) return fail(FHE_ERR_UNSUPPORTED, "the Galois map on the special limbs' INTT needs the out-of-place (sharded) form and N >= 2^5");
    // sharded: out of place, from acc straight into this rank's slot of gather buffer 2 (round 2 copied the rows there first)
    const u64 *from = nullptr;
    if (p->sharded) {
        from = sp;
        sp = p->g2 + (size_t)sh.rank * 2 * sh.smax * N;
        stride = (u32)sh.smax;
    }
    {
        TraceScope tr_ntt(ctx, st, "NTT");
        rc = for_each_run(t, count, first, [&](size_t off, size_t len, int path) -> int {
            PassArgs a{sp + off * N, lp, (u32)(first + off), (u32)len, (u32)(2 * len), stride, nullptr};
            if (from) {
                a.src = from + off * N;
                a.src_stride = (u32)MO;
                a.galois = galois;
            }
            hipError_t e2 = launch_ntt(st, a, p->log_n, true, path, 1);
            return e2 == hipSuccess ? FHE_OK : hip_fail(e2, "launch_ntt");
        });
        if (rc) return rc;
    }
    // BGV: remove delta = t * [acc * t^-1]_P instead of [acc]_P, so that delta = 0 mod t
    if (p->plain_modulus)
        for (int h = 0; h < 2; h++)
            if ((rc = fhe_scalar_affine(ctx, sp + (size_t)h * stride * N, sp + (size_t)h * stride * N, p->t_inv_P.data() + (sh.slo - p->L), nullptr, t, 1, sh.sn,
                                        sh.slo, st)))
                return rc;
    return FHE_OK;
}

// the mod-down's tail rides on the last pass of the converted limbs' forward transform (ks_finish) unless a test hook or an
// experimental transform variant is active
static bool ks_fast_path(const fhe_ctx *ctx, const fhe_keyswitch *p)
{
    return ntt_subscale_supported(p->log_n) && ctx->mode == 0 && ctx->fault_idx < 0 && !ctx->packed_on && ctx->only_pass < 0 && !ctx->resident;
}

// rest of the mod-down, to the owned ciphertext limbs: conversion of the (gathered) special limbs to Q, NTT, subtract, times P^-1
// (galois != 0: the addends are sigma_k(d_add*), read through the Galois map by the fused tail -- ks_fast_path() shapes only)
// sp_hoist != nullptr (a hoisted rotation): the special limbs in coefficient form sit there ([2][K][N]) and the sums in acc are still in
// the un-rotated frame -- the tail reads them through the Galois map as well
static int ks_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_add0, const uint64_t *d_add1,
                     hipStream_t st, u32 galois = 0, const u64 *sp_hoist = nullptr, bool galois_acc = false)
{
    galois_acc = galois_acc || sp_hoist != nullptr;
    const fhe_ntt_tables *t = p->t;
    const KsShard &sh = p->sh;
    if (!sh.cn) return FHE_OK;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *acc = p->acc_cur(), *conv = p->conv_cur();
    int rc;
    hipError_t e;
    const bool plain = !ks_fast_path(ctx, p);
    if (plain && galois) return fail(FHE_ERR_INVALID, "the Galois map on the addends belongs to the fused tail");
    // one special prime at a two-launch size: the conversion x mod q_j rides on the converted limbs' column pass
    const bool trivial = p->K == 1 && p->log_n >= 13 && !plain;
    if (galois_acc && plain) return fail(FHE_ERR_UNSUPPORTED, "hoisted rotations need the fused mod-down tail");
    if (!trivial) {
        e = launch_baseconv_exact_jobs(st, (sp_hoist ? (p->cur ? p->hdown_jobs2 : p->hdown_jobs) : p->down_jobs).as<BcJob>(), 2, p->down->dev.m, p->down->dev.k, p->down->dev.f64 != 0, N, p->down->dev.m <= 16 ? 1u << (p->down->dev.m - 1) : 0u);
        if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact_jobs");
        if (p->plain_modulus && (rc = fhe_scalar_affine(ctx, conv, conv, p->t_mod_Q.data() + sh.clo, nullptr, t, 2, sh.cn, sh.clo, st))) return rc;
    }
    if (plain) {
        if ((rc = ntt_batch(ctx, conv, t, 2, sh.cn, sh.clo, st, false))) return rc;
        const SubScaleArgs sa{d_out0, d_out1, acc, conv, d_add0, p->pinv.as<u64>(), (u64)(MO * N), (u64)((size_t)sh.cn * N), lp, (u32)sh.clo, (u32)sh.cn, p->log_n, d_add1};
        if ((e = launch_sub_scale(st, sa)) != hipSuccess) return hip_fail(e, "launch_sub_scale");
        return FHE_OK;
    }
    // the converted limbs' forward transform with the tail -- subtract from acc, times P^-1, plus the optional addends -- riding on
    // its last pass: neither the transformed limbs nor a separate tail launch exist
    TraceScope tr_ntt(ctx, st, "NTT");
    return for_each_run(t, sh.cn, sh.clo, [&](size_t off, size_t len, int path) -> int {
        PassArgs a{conv + off * N, lp, (u32)(sh.clo + off), (u32)len, (u32)(2 * len), (u32)sh.cn};
        RowEpiArgs ep{{d_out0 + off * N, d_out1 + off * N, nullptr}, {d_add0 ? d_add0 + off * N : nullptr, d_add1 ? d_add1 + off * N : nullptr, nullptr},
                      acc + off * N, (u64)(MO * N), p->pinv.as<u64>() + off};
        ep.galois = galois;
        ep.galois_a = galois_acc ? 1u : 0u;
        if (trivial) {
            a.src = sp_hoist ? sp_hoist : (p->sharded ? p->g2 : acc) + (size_t)p->down_src_row * N;
            a.src_bcast = sp_hoist ? (u64)p->K * N : (u64)p->down_src_stride * N;
            if (p->plain_modulus) ep.pre = p->d_t_mod_Q.as<u64>() + sh.clo + off;     // BGV: delta = t * [acc t^-1]_P
        }
        hipError_t e2 = launch_ntt_subscale(st, a, ep, p->log_n, path);
        return e2 == hipSuccess ? FHE_OK : hip_fail(e2, "launch_ntt_subscale");
    });
}

// Mod-down and rescale behind ONE forward transform (fhe_hmult with rescale, one device, CKKS form).  The sequence of the reference --
// relinearize_inplace, then mod_switch_to_next_inplace (reliability_test/dotprod_test.cu:114-115) -- transforms the converted special
// limbs X_j = NTT(conv_j), forms v_j = (acc_j - X_j) P^-1 + d_j, turns v_{L-1} to coefficient form y, transforms y mod q_j for every
// remaining prime (Delta_j) and forms (v_j - Delta_j) q_last^-1.  The transform is linear: NTT(conv_j + P y) = X_j + P Delta_j and
//   ((acc_j - NTT(conv_j + P y)) P^-1 + d_j) q_last^-1 = (v_j - Delta_j) q_last^-1,
// the same residues, hence the same canonical words.  y itself needs no forward transform either: INTT(X_{L-1}) = conv_{L-1}, so
//   y = INTT(acc_{L-1} P^-1 + d_{L-1}) - P^-1 conv_{L-1}  (mod q_{L-1}),
// whose first term rides along with the special limbs' INTT (ks_special_intt, with_last).  What is left after the conversion: one
// word-wise launch for y, ONE column pass that loads conv_j + (P mod q_j) (y mod q_j) (ColAddSrc) and ONE row pass whose tail ends with
// the second factor (RowEpiArgs::scal2).  The full-width transform of the y residues, the L-limb intermediate and its re-read disappear.
bool ks_rescale_fusable(const fhe_ctx *ctx, const fhe_keyswitch *p)
{
    return ctx->hmult_fused_rescale && !p->plain_modulus && !ctx->trace_on && ks_fast_path(ctx, p) && p->log_n >= 13 && p->K >= 2 && p->L >= 2 && p->rs_bc != nullptr &&
           p->pq_tw.bytes != 0;
}

// the last limb's part of y, INTT(acc P^-1 + d), in place in row L-1 of acc -- for a plan whose special limbs' INTT does not take it along
// (sharded: that launch writes into the gather buffer)
static int ks_rescale_last(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_add0, const uint64_t *d_add1, hipStream_t st)
{
    if (!p->own_last) return FHE_OK;
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own, R = (size_t)p->L - 1, lr = R - p->sh.clo;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *acc = p->acc_cur();
    const SubScaleArgs sa{acc + lr * N, acc + (MO + lr) * N, acc + lr * N, nullptr, d_add0 ? d_add0 + lr * N : nullptr, p->pinv.as<u64>() + lr, (u64)(MO * N), 0, lp, (u32)R, 1u, p->log_n,
                          d_add1 ? d_add1 + lr * N : nullptr};
    hipError_t e = launch_sub_scale(st, sa);
    if (e != hipSuccess) return hip_fail(e, "launch_sub_scale");
    PassArgs a{acc + lr * N, lp, (u32)R, 1u, 2u, (u32)MO};
    if ((e = launch_ntt(st, a, p->log_n, true, t->path[R], 1)) != hipSuccess) return hip_fail(e, "launch_ntt");
    return FHE_OK;
}

// conversion of the (gathered) special limbs to the owned ciphertext limbs; the owner of limb L-1 then forms y (rs_bc: [2][N])
static int ks_finish_rescale_begin(fhe_ctx *ctx, fhe_keyswitch *p, hipStream_t st)
{
    const fhe_ntt_tables *t = p->t;
    const KsShard &sh = p->sh;
    if (!sh.cn) return FHE_OK;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own, R = (size_t)p->L - 1, lr = R - sh.clo;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *acc = p->acc_cur(), *conv = p->conv_cur();
    hipError_t e = launch_baseconv_exact_jobs(st, p->down_jobs.as<BcJob>(), 2, p->down->dev.m, p->down->dev.k, p->down->dev.f64 != 0, N,
                                              p->down->dev.m <= 16 ? 1u << (p->down->dev.m - 1) : 0u);
    if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact_jobs");
    if (p->own_last) {
        // y = INTT(v_{L-1}) = INTT(acc P^-1 + d) - P^-1 conv_{L-1}: the first term sits in row L-1 of acc
        const SubScaleArgs sa{p->rs_bc, p->rs_bc + N, nullptr, conv + lr * N, acc + lr * N, p->pinv.as<u64>() + lr, 0, (u64)((size_t)sh.cn * N), lp, (u32)R, 1u, p->log_n,
                              acc + (MO + lr) * N};
        if ((e = launch_sub_scale(st, sa)) != hipSuccess) return hip_fail(e, "launch_sub_scale");
    }
    return FHE_OK;
}

// ONE forward transform of conv_j + P y on the owned limbs below L-1, its tail ending with q_last^-1; outputs: rs_n rows per part
static int ks_finish_rescale_end(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_add0, const uint64_t *d_add1, hipStream_t st)
{
    const fhe_ntt_tables *t = p->t;
    const KsShard &sh = p->sh;
    if (!p->rs_n) return FHE_OK;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *acc = p->acc_cur(), *conv = p->conv_cur();
    const ColAddSrc as{p->rs_bc, (u64)N, p->pq_tw.as<Tw>()};
    return for_each_run(t, p->rs_n, sh.clo, [&](size_t off, size_t len, int path) -> int {
        PassArgs a{conv + off * N, lp, (u32)(sh.clo + off), (u32)len, (u32)(2 * len), (u32)sh.cn};
        RowEpiArgs ep{{d_out0 + off * N, d_out1 + off * N, nullptr}, {d_add0 ? d_add0 + off * N : nullptr, d_add1 ? d_add1 + off * N : nullptr, nullptr},
                      acc + off * N, (u64)(MO * N), p->pinv.as<u64>() + off};
        ep.scal2 = p->qlast_inv.as<u64>() + off;
        hipError_t e2 = launch_ntt_subscale(st, a, ep, p->log_n, path, &as);
        return e2 == hipSuccess ? FHE_OK : hip_fail(e2, "launch_ntt_subscale");
    });
}

// Hybrid RNS key switching on one device, operation order of the reference's SEAL trace (profile_framewk/build/data/ckks/16384_4:466-539)
// with the launches batched.  d_add0 / d_add1 (optional, L x N): added to the output parts -- a rotation passes sigma(c0), a
// relinearisation d0 and d1.
int keyswitch_core(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c, const uint64_t *d_evk,
                   const uint64_t *d_add0, const uint64_t *d_add1, void *stream, bool rescale)
{
    if (!ctx || !p || !d_out0 || !d_out1 || !d_c || !d_evk) return fail(FHE_ERR_INVALID, "null argument");
    if (p->sharded)
        return fail(FHE_ERR_INVALID, "a sharded plan runs through fhe_keyswitch_shard_begin / _inner / _finish with the all-gathers between them");
    if (rescale && !ks_rescale_fusable(ctx, p)) return fail(FHE_ERR_UNSUPPORTED, "this plan rescales in a separate step");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    int rc;
    TraceScope tr_ks(ctx, st, "KEYSWITCH");
    if ((rc = ks_begin(ctx, p, d_c, st))) return rc;
    if ((rc = ks_extend_mac(ctx, p, d_c, d_evk, st))) return rc;
    TraceScope tr_ms(ctx, st, "MODSWITCH");
    if ((rc = ks_special_intt(ctx, p, st, 0, rescale, d_add0, d_add1))) return rc;
    if (rescale) {
        if ((rc = ks_finish_rescale_begin(ctx, p, st))) return rc;
        return ks_finish_rescale_end(ctx, p, d_out0, d_out1, d_add0, d_add1, st);
    }
    return ks_finish(ctx, p, d_out0, d_out1, d_add0, d_add1, st);
}

extern "C" {

// ---------------------------------------------------------------- rotation / key switching
int fhe_automorphism(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, const fhe_ntt_tables *t, uint32_t galois_elt,
                     size_t n_poly, size_t limbs, size_t start_idx, void *stream)
{
    if (!ctx || !d_dst || !d_src || d_dst == d_src || !(galois_elt & 1)) return fail(FHE_ERR_INVALID, "bad automorphism arguments");
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_automorphism(pick(ctx, stream), d_dst, d_src, t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs,
                                       (u32)(n_poly * limbs), t->log_n, galois_elt);
    if (e != hipSuccess) return hip_fail(e, "launch_automorphism");
    return FHE_OK;
}

int fhe_automorphism_ntt(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, int log_n, uint32_t galois_elt, size_t n_units,
                         void *stream)
{
    if (!ctx || !d_dst || !d_src || d_dst == d_src || !(galois_elt & 1) || log_n < 1 || log_n > 30)
        return fail(FHE_ERR_INVALID, "bad automorphism arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_automorphism_ntt(pick(ctx, stream), d_dst, d_src, (u32)n_units, log_n, galois_elt);
    if (e != hipSuccess) return hip_fail(e, "launch_automorphism_ntt");
    return FHE_OK;
}

int fhe_keyswitch_create(fhe_ctx *ctx, const fhe_ntt_tables *t, int L, int K, int dnum, fhe_keyswitch **out)
{
    return ks_create(ctx, t, L, K, dnum, 1, 0, nullptr, nullptr, nullptr, out);
}

int fhe_keyswitch_create_sharded(fhe_ctx *ctx, const fhe_ntt_tables *t, int L, int K, int dnum, int world, int rank, uint64_t *d_gather1,
                                 uint64_t *d_gather2, uint64_t *d_bcast, fhe_keyswitch **out)
{
    return ks_create(ctx, t, L, K, dnum, world, rank, d_gather1, d_gather2, d_bcast, out);
}

int fhe_keyswitch_shard_layout(int L, int K, int world, int rank, int out[6])
{
    if (!out || L < 1 || K < 1 || world < 1 || rank < 0 || rank >= world) return fail(FHE_ERR_INVALID, "bad shard arguments");
    KsShard s;
    ks_shard_layout(L, K, world, rank, &s);
    out[0] = s.clo;
    out[1] = s.cn;
    out[2] = s.slo;
    out[3] = s.sn;
    out[4] = s.cmax;
    out[5] = s.smax;
    return FHE_OK;
}

int fhe_keyswitch_set_plain_modulus(fhe_keyswitch *p, uint64_t plain_modulus)
{
    if (!p) return fail(FHE_ERR_INVALID, "null plan");
    p->plain_modulus = plain_modulus;
    p->t_inv_P.clear();
    p->t_mod_Q.clear();
    if (plain_modulus) {
        for (int k = 0; k < p->K; k++) {
            const u64 pk = p->t->q[p->L + k], inv = host::inv_mod(plain_modulus % pk, pk);
            if (!inv) return fail(FHE_ERR_INVALID, "plain modulus must be coprime to the special primes");
            p->t_inv_P.push_back(inv);
        }
        for (int j = 0; j < p->L; j++) p->t_mod_Q.push_back(plain_modulus % p->t->q[j]);
        (void)hipSetDevice(p->ctx->device);
        HIP_TRY(p->d_t_mod_Q.upload(p->t_mod_Q));
        if (p->L >= 2) {
            const u64 ql = p->t->q[p->L - 1];
            p->t_inv_qlast = host::inv_mod(plain_modulus % ql, ql);
            if (!p->t_inv_qlast) return fail(FHE_ERR_INVALID, "plain modulus must be coprime to the ciphertext primes");
        }
    }
    return FHE_OK;
}

int fhe_keyswitch_destroy(fhe_keyswitch *p)
{
    if (p) {
        (void)hipSetDevice(p->ctx->device);
        delete p;
    }
    return FHE_OK;
}

int fhe_keyswitch_apply(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c,
                        const uint64_t *d_evk, void *stream)
{
    return keyswitch_core(ctx, p, d_out0, d_out1, d_c, d_evk, nullptr, nullptr, stream);
}

int fhe_keyswitch_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c_local, void *stream)
{
    if (!ctx || !p || (!d_c_local && p->sh.cn)) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_begin(ctx, p, d_c_local, pick(ctx, stream));
}

int fhe_keyswitch_shard_inner(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c_local, const uint64_t *d_evk_local, void *stream)
{
    if (!ctx || !p || (!d_c_local && p->sh.cn) || !d_evk_local) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    int rc = ks_extend_mac(ctx, p, d_c_local, d_evk_local, st);
    return rc ? rc : ks_special_intt(ctx, p, st);
}

int fhe_keyswitch_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local, const uint64_t *d_add0_local,
                               const uint64_t *d_add1_local, void *stream)
{
    if (!ctx || !p || ((!d_out0_local || !d_out1_local) && p->sh.cn)) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_finish(ctx, p, d_out0_local, d_out1_local, d_add0_local, d_add1_local, pick(ctx, stream));
}

// Sharded hmult with the mod-down and the rescale behind one transform (ks_finish_rescale_*): after the second all-gather
//   fhe_hmult_shard_finish_begin  (owner of limb L-1: y into the broadcast buffer)  ->  ONE broadcast of 2 x N words  ->
//   fhe_hmult_shard_finish_end    (the owned limbs below L-1 of both parts, rs_n rows each)
// instead of fhe_keyswitch_shard_finish + fhe_rescale_shard_begin / _finish: same words, same three collectives per hmult.
int fhe_hmult_shard_fusable(const fhe_ctx *ctx, const fhe_keyswitch *p)
{
    return ctx && p && p->sharded && ks_rescale_fusable(ctx, p) ? 1 : 0;
}

int fhe_hmult_shard_finish_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_add0_local, const uint64_t *d_add1_local, void *stream)
{
    if (!ctx || !p) return fail(FHE_ERR_INVALID, "null argument");
    if (!p->sharded || !ks_rescale_fusable(ctx, p)) return fail(FHE_ERR_UNSUPPORTED, "this plan rescales in a separate step (fhe_hmult_shard_fusable)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    int rc;
    if ((rc = ks_rescale_last(ctx, p, d_add0_local, d_add1_local, st))) return rc;
    return ks_finish_rescale_begin(ctx, p, st);
}

int fhe_hmult_shard_finish_end(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local, const uint64_t *d_add0_local,
                               const uint64_t *d_add1_local, void *stream)
{
    if (!ctx || !p) return fail(FHE_ERR_INVALID, "null argument");
    if (!p->sharded || !ks_rescale_fusable(ctx, p)) return fail(FHE_ERR_UNSUPPORTED, "this plan rescales in a separate step (fhe_hmult_shard_fusable)");
    if (p->rs_n && (!d_out0_local || !d_out1_local)) return fail(FHE_ERR_INVALID, "null argument");
    if (p->rs_n && d_out0_local == d_out1_local) return fail(FHE_ERR_INVALID, "the two output parts must be distinct buffers");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_finish_rescale_end(ctx, p, d_out0_local, d_out1_local, d_add0_local, d_add1_local, pick(ctx, stream));
}

// the last phase of a rotation: the mod-down with sigma(c0) added to the first part -- through the Galois map on the fused tail's
// load, or (test hooks, experimental transform variants, N < 2^5) from a permuted copy made by a launch of its own
static int rotate_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c0, u32 galois, hipStream_t st)
{
    if (ks_fast_path(ctx, p) || !p->sh.cn) return ks_finish(ctx, p, d_out0, d_out1, d_c0, nullptr, st, galois);
    const size_t N = (size_t)1 << p->log_n;
    u64 *sig0 = p->rot.as<u64>() + (size_t)p->sh.cn * N;
    hipError_t e = launch_automorphism_ntt(st, sig0, d_c0, (u32)p->sh.cn, p->log_n, galois);
    if (e != hipSuccess) return hip_fail(e, "launch_automorphism_ntt");
    return ks_finish(ctx, p, d_out0, d_out1, sig0, nullptr, st);
}

int fhe_rotate(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c0, const uint64_t *d_c1,
               uint32_t galois_elt, const uint64_t *d_galois_key, void *stream)
{
    if (!ctx || !p || !d_out0 || !d_out1 || !d_c0 || !d_c1 || !d_galois_key) return fail(FHE_ERR_INVALID, "null argument");
    if (d_out0 == d_c0 || d_out1 == d_c1 || d_out0 == d_c1 || d_out1 == d_c0 || d_out0 == d_out1) return fail(FHE_ERR_INVALID, "rotate is out of place");
    if (!(galois_elt & 1)) return fail(FHE_ERR_INVALID, "Galois elements are odd");
    if (p->sharded)
        return fail(FHE_ERR_INVALID, "a sharded plan runs through fhe_rotate_shard_begin / _inner / _finish with the all-gathers between them");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    TraceScope tr(ctx, st, "ROTATE", true);
    // sigma(c1) is a ciphertext part under sigma(s): switch it back to s with the Galois key; sigma(c0) is added to the first part.
    // Both automorphisms (in the NTT domain: permutations of the slots) ride on loads of the key switch's own launches: the opening
    // INTT reads c1 through the map (and leaves sigma(c1) in the plan's buffer for the inner product), the mod-down's tail reads c0
    // through it.
    int rc;
    TraceScope tr_ks(ctx, st, "KEYSWITCH");
    u64 *sig1 = p->rot.as<u64>();
    if ((rc = ks_begin(ctx, p, d_c1, st, galois_elt, sig1))) return rc;
    if ((rc = ks_extend_mac(ctx, p, sig1, d_galois_key, st))) return rc;
    TraceScope tr_ms(ctx, st, "MODSWITCH");
    if ((rc = ks_special_intt(ctx, p, st))) return rc;
    return rotate_finish(ctx, p, d_out0, d_out1, d_c0, galois_elt, st);
}

// ---------------------------------------------------------------- hoisted rotations
// Rotations of ONE ciphertext by several Galois elements (the baby steps of a BSGS matrix-vector product,
// profile_framewk/src/matmul_ckks.cpp:45-113; the rotate-and-sum of reliability_test/dotprod_test.cu:143-148): the decomposition of c1
// -- INTT, digit extension, the extended limbs' forward transform up to the launch the inner product rides on -- does not depend on
// the Galois element and is done once.  sigma is a ring automorphism, so
//     sum_d sigma(ext_d) * key_d  =  sigma( sum_d ext_d * sigma^-1(key_d) ):
// with the key stored in the un-rotated frame (fhe_galois_key_prepare: sigma^-1 of every key row, once per key) the inner product
// runs on the shared digits exactly as a plain key switch's, and sigma is taken on loads the mod-down makes anyway: the special limbs'
// INTT reads its sums through the Galois map, the fused tail reads the ciphertext limbs' sums and c0 through it.
// The result is the rotation with ext_d = sigma(extension of c1's digit) where fhe_rotate uses the extension of sigma(c1)'s digit: the
// two differ by multiples of the digit modulus where sigma flips a sign (the exact extension lifts to [0, P_d), which negation does not
// preserve), so the words differ while both are valid key switches of sigma(c1) (tests/: oracle rotate_hoisted_ref word for word, and
// the decryption of both).
static int hoist_buffers(fhe_ctx *ctx, fhe_keyswitch *p)
{
    if (p->hsp.p) return FHE_OK;
    const size_t N = (size_t)1 << p->log_n, K = p->K;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(p->hsp.alloc(2 * K * N * 8));
    std::vector<u32> rows(2 * K);
    for (size_t i = 0; i < 2 * K; i++) rows[i] = (u32)i;
    HIP_TRY(p->hdown_rows.upload(rows));
    std::vector<BcJob> down;
    for (int h = 0; h < 2; h++)
        down.push_back(BcJob{p->down->dev, p->hsp.as<u64>(), p->conv.as<u64>() + (size_t)h * p->sh.cn * N, 0xFFFFFFFFu, 0u, p->hdown_rows.as<u32>() + (size_t)h * K});
    HIP_TRY(p->hdown_jobs.upload(down));
    // the second set (rotations on the side stream)
    HIP_TRY(p->acc2.alloc(p->acc.bytes));
    HIP_TRY(p->conv2.alloc(p->conv.bytes));
    HIP_TRY(p->hsp2.alloc(2 * K * N * 8));
    std::vector<BcJob> down2;
    for (int h = 0; h < 2; h++)
        down2.push_back(BcJob{p->down->dev, p->hsp2.as<u64>(), p->conv2.as<u64>() + (size_t)h * p->sh.cn * N, 0xFFFFFFFFu, 0u, p->hdown_rows.as<u32>() + (size_t)h * K});
    HIP_TRY(p->hdown_jobs2.upload(down2));
    // the sums of a group of up to four rotations (launch_ks_mac_multi)
    HIP_TRY(p->acc_multi.alloc(4 * p->acc.bytes));
    return FHE_OK;
}

extern "C" int fhe_galois_key_prepare(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_key_out, const uint64_t *d_key_in, uint32_t galois_elt, void *stream)
{
    if (!ctx || !p || !d_key_out || !d_key_in || d_key_out == d_key_in || !(galois_elt & 1)) return fail(FHE_ERR_INVALID, "bad arguments");
    // sigma_k^-1 = sigma_{k^-1 mod 2N}
    const u64 two_n = (u64)2 << p->log_n;
    const u64 kinv = host::inv_mod(galois_elt % two_n, two_n);
    if (!kinv) return fail(FHE_ERR_INVALID, "Galois elements are odd");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_automorphism_ntt(pick(ctx, stream), d_key_out, d_key_in, (u32)((size_t)p->dnum * 2 * p->m_own), p->log_n, (u32)kinv);
    if (e != hipSuccess) return hip_fail(e, "launch_automorphism_ntt");
    return FHE_OK;
}

extern "C" int fhe_rotate_hoisted(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *const *d_out0, uint64_t *const *d_out1, const uint64_t *d_c0,
                                  const uint64_t *d_c1, const uint32_t *galois_elts, const uint64_t *const *d_prepared_keys, size_t n_rot, void *stream)
{
    if (!ctx || !p || !d_c0 || !d_c1 || (n_rot && (!d_out0 || !d_out1 || !galois_elts || !d_prepared_keys))) return fail(FHE_ERR_INVALID, "null argument");
    if (p->sharded) return fail(FHE_ERR_UNSUPPORTED, "hoisted rotations run on one device");
    if (p->log_n < 5 || !ks_fast_path(ctx, p)) return fail(FHE_ERR_UNSUPPORTED, "hoisted rotations need N >= 2^5 and the default transform path");
    for (size_t r = 0; r < n_rot; r++) {
        if (!d_out0[r] || !d_out1[r] || !d_prepared_keys[r]) return fail(FHE_ERR_INVALID, "null argument");
        if (!(galois_elts[r] & 1)) return fail(FHE_ERR_INVALID, "Galois elements are odd");
        if (d_out0[r] == d_c0 || d_out1[r] == d_c0 || d_out0[r] == d_c1 || d_out1[r] == d_c1 || d_out0[r] == d_out1[r])
            return fail(FHE_ERR_INVALID, "rotate is out of place");
    }
    if (!n_rot) return FHE_OK;
    int rc = hoist_buffers(ctx, p);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, MO = p->m_own, L = p->L, K = p->K;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    const bool fused = ks_hoist_fused(ctx, p, n_rot);
    // shared: decomposition of c1 (no automorphism yet)
    {
        TraceScope tr(ctx, st, "HOIST");
        if ((rc = ks_begin(ctx, p, d_c1, st))) return rc;
        if ((rc = ks_extend(ctx, p, st, fused))) return rc;
    }
    // Rotations alternate between the caller's stream and the context's side stream (fork / join by events; not inside a stream capture,
    // not while tracing): a rotation's conversions (FP64 issue-bound) and small transforms run under the next one's inner product
    // (memory-bound).  Each stream has its own sums / special limbs / converted limbs; the shared digits are read-only.
    fhe_ctx::Side *sd = nullptr;
    if (n_rot > 1 && ctx->split != 0 && !ctx->trace_on) {
        if ((rc = side_stream(ctx, st, &sd))) return rc;
        if (sd) {
            HIP_TRY(hipEventRecord(sd->fork, st));
            HIP_TRY(hipStreamWaitEvent(sd->s, sd->fork, 0));
        }
    }
    rc = FHE_OK;
    // the mod-down of one rotation on stream s, its sums in p->acc_cur()
    auto mod_down = [&](size_t r, hipStream_t s) -> int {
        u64 *acc = p->acc_cur(), *hsp = p->hsp_cur();
        const u32 g = galois_elts[r];
        TraceScope tr_ms(ctx, s, "MODSWITCH");
        // INTT of sigma(special limbs of the sums), out of place: acc ([2][M][N], special limbs from row L) -> hsp ([2][K][N])
        {
            TraceScope tr_ntt(ctx, s, "NTT");
            int rc2 = for_each_run(t, K, L, [&](size_t off, size_t len, int path) -> int {
                PassArgs a{hsp + off * N, lp, (u32)(L + off), (u32)len, (u32)(2 * len), (u32)K, nullptr};
                a.src = acc + (L + off) * N;
                a.src_stride = (u32)MO;
                a.galois = g;
                hipError_t e2 = launch_ntt(s, a, p->log_n, true, path, 1);
                return e2 == hipSuccess ? FHE_OK : hip_fail(e2, "launch_ntt");
            });
            if (rc2) return rc2;
        }
        if (p->plain_modulus)
            for (int h = 0; h < 2; h++)
                if (int rc2 = fhe_scalar_affine(ctx, hsp + (size_t)h * K * N, hsp + (size_t)h * K * N, p->t_inv_P.data(), nullptr, t, 1, K, L, s)) return rc2;
        return ks_finish(ctx, p, d_out0[r], d_out1[r], d_c0, nullptr, s, g, hsp);
    };
    if (!fused && n_rot > 1 && !ctx->trace_on && !sd) {
        // ONE STREAM (inside a stream capture, or "ntt_split" off): groups of up to four rotations, ONE pass over the shared digits forms the
        // sums of the whole group (launch_ks_mac_multi: the digits are read once per group, not once per key): 220 -> 192 us per rotation at
        // config 5.  With the side stream the per-rotation inner products already run under the other stream's mod-down and the grouped form
        // measured the same (169 against 170 us; 73 against 68 us at N = 2^16, L = 16): it is kept for the one-stream case only.
        const size_t slot = 2 * MO * N;      // (acc_multi: hoist_buffers)
        const KsShard &sh = p->sh;
        for (size_t g0 = 0; g0 < n_rot && !rc; g0 += 4) {
            const size_t n = n_rot - g0 < 4 ? n_rot - g0 : 4;
            if (g0 && sd) {     // the side stream's mod-downs of the previous group still read their sums
                HIP_TRY(hipEventRecord(sd->join, sd->s));
                HIP_TRY(hipStreamWaitEvent(st, sd->join, 0));
            }
            KsMacMultiArgs m{KsMacArgs{nullptr, p->ext.as<u64>(), d_c1, nullptr, lp, (u32)p->L, (u32)MO, (u32)p->dnum, (u32)p->alpha, p->log_n, (u32)sh.cn, (u32)sh.clo,
                                       (u32)(sh.slo - sh.cn)},
                             (u32)n, {nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
            for (size_t i = 0; i < n; i++) {
                m.evk[i] = d_prepared_keys[g0 + i];
                m.acc[i] = p->acc_multi.as<u64>() + i * slot;
            }
            hipError_t e = launch_ks_mac_multi(st, m);
            if (e != hipSuccess) {
                rc = hip_fail(e, "launch_ks_mac_multi");
                break;
            }
            if (sd) {
                HIP_TRY(hipEventRecord(sd->fork, st));
                HIP_TRY(hipStreamWaitEvent(sd->s, sd->fork, 0));
            }
            for (size_t i = 0; i < n && !rc; i++) {
                p->cur = sd ? (int)(i & 1) : 0;
                p->acc_ovr = p->acc_multi.as<u64>() + i * slot;
                rc = mod_down(g0 + i, p->cur ? sd->s : st);
            }
        }
        p->acc_ovr = nullptr;
    } else {
        for (size_t r = 0; r < n_rot && !rc; r++) {
            p->cur = sd ? (int)(r & 1) : 0;
            hipStream_t s = p->cur ? sd->s : st;
            TraceScope tr(ctx, s, "ROTATE", true);
            if ((rc = ks_mac(ctx, p, d_c1, d_prepared_keys[r], s, fused))) break;
            rc = mod_down(r, s);
        }
    }
    p->cur = 0;
    if (sd) {       // join even after a failed launch: the side stream must not be left forked
        HIP_TRY(hipEventRecord(sd->join, sd->s));
        HIP_TRY(hipStreamWaitEvent(st, sd->join, 0));
    }
    return rc;
}

// Hoisted rotations on a limb-sharded plan: the input's all-gather (gather 1) and the digit extension are shared by all Galois elements,
// per element only the inner product, the special limbs' all-gather (gather 2) and the mod-down run -- ONE collective per rotation
// instead of two.
//   fhe_rotate_hoisted_shard_begin   INTT of the owned limbs of c1 (un-rotated) into gather buffer 1     -- all-gather 1 (once) --
//   fhe_rotate_hoisted_shard_extend  extension of every digit to the owned limbs + column pass            (once)
//   fhe_rotate_hoisted_shard_inner   per element: inner product with the owned rows of the prepared key, INTT of sigma(owned special
//                                    limbs of the sums) into gather buffer 2                              -- all-gather 2 --
//   fhe_rotate_hoisted_shard_finish  per element: mod-down to the owned limbs, sums and c0 read through the Galois map
extern "C" int fhe_rotate_hoisted_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c1_local, void *stream)
{
    if (!ctx || !p || (!d_c1_local && p->sh.cn)) return fail(FHE_ERR_INVALID, "null argument");
    if (!p->sharded) return fail(FHE_ERR_INVALID, "a plan without gather buffers runs fhe_rotate_hoisted");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_begin(ctx, p, d_c1_local, pick(ctx, stream));
}

extern "C" int fhe_rotate_hoisted_shard_extend(fhe_ctx *ctx, fhe_keyswitch *p, void *stream)
{
    if (!ctx || !p) return fail(FHE_ERR_INVALID, "null argument");
    if (!p->sharded) return fail(FHE_ERR_INVALID, "a plan without gather buffers runs fhe_rotate_hoisted");
    if (p->log_n < 5 || !ks_fast_path(ctx, p)) return fail(FHE_ERR_UNSUPPORTED, "hoisted rotations need N >= 2^5 and the default transform path");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_extend(ctx, p, pick(ctx, stream), ks_hoist_fused(ctx, p, 0));
}

extern "C" int fhe_rotate_hoisted_shard_inner(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c1_local, const uint64_t *d_prepared_key_local,
                                              uint32_t galois_elt, void *stream)
{
    if (!ctx || !p || (!d_c1_local && p->sh.cn) || !d_prepared_key_local || !(galois_elt & 1)) return fail(FHE_ERR_INVALID, "bad rotation arguments");
    if (!p->sharded) return fail(FHE_ERR_INVALID, "a plan without gather buffers runs fhe_rotate_hoisted");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    int rc = ks_mac(ctx, p, d_c1_local, d_prepared_key_local, st, ks_hoist_fused(ctx, p, 0));
    return rc ? rc : ks_special_intt(ctx, p, st, galois_elt);
}

extern "C" int fhe_rotate_hoisted_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local,
                                               const uint64_t *d_c0_local, uint32_t galois_elt, void *stream)
{
    if (!ctx || !p || ((!d_out0_local || !d_out1_local || !d_c0_local) && p->sh.cn) || !(galois_elt & 1)) return fail(FHE_ERR_INVALID, "bad rotation arguments");
    if (!p->sharded) return fail(FHE_ERR_INVALID, "a plan without gather buffers runs fhe_rotate_hoisted");
    if (p->sh.cn && (d_out0_local == d_c0_local || d_out1_local == d_c0_local)) return fail(FHE_ERR_INVALID, "rotate is out of place");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_finish(ctx, p, d_out0_local, d_out1_local, d_c0_local, nullptr, pick(ctx, stream), galois_elt, nullptr, true);
}

// ---------------------------------------------------------------- baby-step / giant-step matrix-vector product
// y = sum_{g < n2} sigma_{G_g}( sum_{b < n1} diag[g][b] * sigma_{B_b}(x) ),  sigma_{B_0} = sigma_{G_0} = identity
// (profile_framewk/src/matmul_ckks.cpp:45-113 in the arrangement that needs n1 + n2 - 2 rotations instead of n1 n2 - 1; its plaintext
// block form is motivation/bsgs.py:39-52): the n1 - 1 baby rotations of x are HOISTED (one decomposition of x), the n2 inner sums are one
// launch each (k_diag_mac), the n2 - 1 giant rotations are plain rotations of the inner sums, accumulated into the result.
extern "C" int fhe_bsgs_matvec(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c0, const uint64_t *d_c1,
                               const uint64_t *d_diags, size_t n1, size_t n2, const uint32_t *baby_elts, const uint64_t *const *d_baby_keys_prepared,
                               const uint32_t *giant_elts, const uint64_t *const *d_giant_keys, void *stream)
{
    if (!ctx || !p || !d_out0 || !d_out1 || !d_c0 || !d_c1 || !d_diags || n1 < 1 || n2 < 1 || n1 > 4096 || n2 > 4096) return fail(FHE_ERR_INVALID, "bad arguments");
    if ((n1 > 1 && (!baby_elts || !d_baby_keys_prepared)) || (n2 > 1 && (!giant_elts || !d_giant_keys))) return fail(FHE_ERR_INVALID, "null argument");
    if (d_out0 == d_c0 || d_out0 == d_c1 || d_out1 == d_c0 || d_out1 == d_c1 || d_out0 == d_out1) return fail(FHE_ERR_INVALID, "the product is out of place");
    if (p->sharded) return fail(FHE_ERR_UNSUPPORTED, "the BSGS product runs on one device");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, L = p->L, part = L * N;
    // scratch: baby rotations [n1 - 1][2][L][N], inner sum [2][L][N], one rotated inner sum [2][L][N]
    const size_t need = ((n1 - 1) * 2 + 4) * part * 8;
    if (p->bsgs.bytes < need) {
        HIP_TRY(hipStreamSynchronize(st));        // (growing frees the old block)
        HIP_TRY(p->bsgs.alloc(need));
    }
    u64 *rot = p->bsgs.as<u64>(), *inner = rot + (n1 - 1) * 2 * part, *tmp = inner + 2 * part;
    int rc;
    if (n1 > 1) {
        std::vector<u64 *> o0(n1 - 1), o1(n1 - 1);
        for (size_t b = 1; b < n1; b++) {
            o0[b - 1] = rot + (b - 1) * 2 * part;
            o1[b - 1] = o0[b - 1] + part;
        }
        if ((rc = fhe_rotate_hoisted(ctx, p, o0.data(), o1.data(), d_c0, d_c1, baby_elts, d_baby_keys_prepared, n1 - 1, stream))) return rc;
    }
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    for (size_t g = 0; g < n2; g++) {
        // inner sum of giant step g: straight into the result for g = 0
        u64 *s0 = g ? inner : d_out0, *s1 = g ? inner + part : d_out1;
        const DiagMacArgs da{s0, s1, d_diags + g * n1 * part, d_c0, d_c1, rot, lp, 0u, (u32)L, (u32)n1, p->log_n};
        hipError_t e = launch_diag_mac(st, da);
        if (e != hipSuccess) return hip_fail(e, "launch_diag_mac");
        if (!g) continue;
        if ((rc = fhe_rotate(ctx, p, tmp, tmp + part, inner, inner + part, giant_elts[g - 1], d_giant_keys[g - 1], stream))) return rc;
        if ((rc = fhe_modadd(ctx, d_out0, d_out0, tmp, t, 1, L, 0, stream))) return rc;
        if ((rc = fhe_modadd(ctx, d_out1, d_out1, tmp + part, t, 1, L, 0, stream))) return rc;
    }
    return FHE_OK;
}

// The three phases of a rotation on a limb-sharded plan (the joins between them are the key switch's, fhe_keyswitch_shard_*): the
// automorphism permutes slots inside each limb, so every rank applies it to its own rows -- on the loads of its launches.
int fhe_rotate_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_c1_local, uint32_t galois_elt, void *stream)
{
    if (!ctx || !p || (!d_c1_local && p->sh.cn) || !(galois_elt & 1)) return fail(FHE_ERR_INVALID, "bad rotation arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    return ks_begin(ctx, p, d_c1_local, pick(ctx, stream), galois_elt, p->rot.as<u64>());
}

int fhe_rotate_shard_inner(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_galois_key_local, void *stream)
{
    if (!ctx || !p || !d_galois_key_local) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    int rc = ks_extend_mac(ctx, p, p->rot.as<u64>(), d_galois_key_local, st);
    return rc ? rc : ks_special_intt(ctx, p, st);
}

int fhe_rotate_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0_local, uint64_t *d_out1_local, const uint64_t *d_c0_local,
                            uint32_t galois_elt, void *stream)
{
    if (!ctx || !p || ((!d_out0_local || !d_out1_local || !d_c0_local) && p->sh.cn) || !(galois_elt & 1)) return fail(FHE_ERR_INVALID, "bad rotation arguments");
    // the tail reads c0 through the Galois permutation while it writes out0: in place, later tiles would read words already overwritten
    if (p->sh.cn && (d_out0_local == d_c0_local || d_out1_local == d_c0_local)) return fail(FHE_ERR_INVALID, "rotate is out of place");
    HIP_TRY(hipSetDevice(ctx->device));
    return rotate_finish(ctx, p, d_out0_local, d_out1_local, d_c0_local, galois_elt, pick(ctx, stream));
}

} // extern "C"
