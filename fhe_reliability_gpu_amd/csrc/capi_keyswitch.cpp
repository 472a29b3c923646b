// capi_keyswitch.cpp -- Galois automorphisms, hybrid key switching, rotation (part of the C ABI of include/fhe_mi355x.h; shared pieces in capi_internal.hpp)
#include "capi_internal.hpp"

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
    if (!ctx || !t || !out || L < 1 || K < 1 || dnum < 1 || dnum > L || L + K > t->count)
        return fail(FHE_ERR_INVALID, "bad key-switch shape");
    std::unique_ptr<fhe_keyswitch> p(new fhe_keyswitch);
    p->ctx = ctx;
    p->t = t;
    p->L = L;
    p->K = K;
    p->dnum = dnum;
    p->alpha = (L + dnum - 1) / dnum;
    p->log_n = t->log_n;
    const size_t N = (size_t)1 << t->log_n, M = (size_t)L + K;
    for (int d = 0; d < dnum; d++) {
        const int lo = d * p->alpha, hi = std::min(L, lo + p->alpha);
        if (lo >= hi) return fail(FHE_ERR_INVALID, "dnum leaves an empty digit");
        std::vector<u64> in(t->q.begin() + lo, t->q.begin() + hi), other;
        for (size_t j = 0; j < M; j++)
            if ((int)j < lo || (int)j >= hi) other.push_back(t->q[j]);
        fhe_baseconv *bc = nullptr;
        int rc = fhe_baseconv_create(ctx, in.data(), (int)in.size(), other.data(), (int)other.size(), &bc);
        if (rc) return rc;
        p->up.push_back(bc);
    }
    {
        std::vector<u64> P(t->q.begin() + L, t->q.begin() + M), Q(t->q.begin(), t->q.begin() + L);
        int rc = fhe_baseconv_create(ctx, P.data(), K, Q.data(), L, &p->down);
        if (rc) return rc;
        std::vector<u64> pinv(L);
        for (int j = 0; j < L; j++) {
            u64 pm = 1 % Q[j];
            for (u64 pk : P) pm = host::mul_mod(pm, pk % Q[j], Q[j]);
            pinv[j] = host::inv_mod(pm, Q[j]);
            if (!pinv[j]) return fail(FHE_ERR_INVALID, "special primes must be coprime to the ciphertext primes");
        }
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(p->pinv.upload(pinv));
    }
    // every limb of every digit's extension except the digit's own limbs, one list per arithmetic path
    for (int path = 0; path < 2; path++) {
        std::vector<UnitRef> map;
        for (int d = 0; d < dnum; d++) {
            const int lo = d * p->alpha, hi = std::min(L, lo + p->alpha);
            for (size_t j = 0; j < M; j++)
                if (((int)j < lo || (int)j >= hi) && t->path[j] == path) map.push_back(UnitRef{(u32)(d * M + j), (u32)j});
        }
        p->ext_units[path] = (u32)map.size();
        if (!map.empty()) HIP_TRY(p->ext_map[path].upload(map));
    }
    HIP_TRY(p->coef.alloc(L * N * 8));
    HIP_TRY(p->ext.alloc((size_t)dnum * M * N * 8));
    HIP_TRY(p->acc.alloc(2 * M * N * 8));
    HIP_TRY(p->conv.alloc(2 * (size_t)L * N * 8));
    HIP_TRY(p->rot.alloc(3 * (size_t)L * N * 8));
    {
        std::vector<BcJob> up, down;
        p->up_batched = true;
        for (int d = 0; d < dnum; d++) {
            const size_t lo = (size_t)d * p->alpha, hi = std::min((size_t)L, lo + (size_t)p->alpha);
            const BaseConvPlanDev &pl = p->up[d]->dev;
            up.push_back(BcJob{pl, p->coef.as<u64>() + lo * N, p->ext.as<u64>() + (size_t)d * M * N, (u32)lo, (u32)(hi - lo)});
            p->up_max_m = std::max(p->up_max_m, pl.m);
            p->up_max_k = std::max(p->up_max_k, pl.k);
            p->up_batched = p->up_batched && pl.f64 == p->up[0]->dev.f64;
        }
        for (int h = 0; h < 2; h++)
            down.push_back(BcJob{p->down->dev, p->acc.as<u64>() + ((size_t)h * M + L) * N, p->conv.as<u64>() + (size_t)h * L * N, 0xFFFFFFFFu, 0u});
        HIP_TRY(p->up_jobs.upload(up));
        HIP_TRY(p->down_jobs.upload(down));
    }
    HIP_TRY(p->hm.alloc(3 * (size_t)L * N * 8));
    HIP_TRY(p->hm_pre.alloc(2 * (size_t)L * N * 8));
    if (L >= 2) {
        // rescale / mod-switch to the next level: drop q_{L-1}
        const u64 ql = t->q[L - 1];
        int rc = fhe_baseconv_create(ctx, &ql, 1, t->q.data(), L - 1, &p->last);
        if (rc) return rc;
        std::vector<u64> qinv(L - 1);
        for (int j = 0; j + 1 < L; j++) {
            qinv[j] = host::inv_mod(ql % t->q[j], t->q[j]);
            if (!qinv[j]) return fail(FHE_ERR_INVALID, "ciphertext primes must be pairwise coprime");
        }
        HIP_TRY(p->qlast_inv.upload(qinv));
        HIP_TRY(p->rs_last.alloc(3 * N * 8));
        HIP_TRY(p->rs_delta.alloc(3 * (size_t)(L - 1) * N * 8));
        std::vector<BcJob> jobs;
        for (int part = 0; part < 3; part++)
            jobs.push_back(BcJob{p->last->dev, p->rs_last.as<u64>() + (size_t)part * N, p->rs_delta.as<u64>() + (size_t)part * (L - 1) * N, 0xFFFFFFFFu, 0u});
        HIP_TRY(p->rs_jobs.upload(jobs));
    }
    *out = p.release();
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

// Hybrid RNS key switching, operation order of the reference's SEAL trace (profile_framewk/build/data/ckks/16384_4:466-539)
// with the launches batched: one INTT, one base extension per digit written straight into the [dnum][M][N]
// layout, ONE forward transform over every extended limb of every digit (unit list), ONE inner-product launch
// for all digits and both key halves, and a mod-down that handles both halves per launch where the layout allows.
int fhe_keyswitch_apply(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c,
                        const uint64_t *d_evk, void *stream)
{
    return keyswitch_core(ctx, p, d_out0, d_out1, d_c, d_evk, nullptr, nullptr, stream);
}

} // extern "C"

// d_add0 / d_add1 (optional, L x N): added to the output parts -- a rotation passes sigma(c0), a relinearisation d0 and d1
int keyswitch_core(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c, const uint64_t *d_evk,
                   const uint64_t *d_add0, const uint64_t *d_add1, void *stream)
{
    if (!ctx || !p || !d_out0 || !d_out1 || !d_c || !d_evk) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, L = p->L, K = p->K, M = L + K;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *coef = p->coef.as<u64>(), *ext = p->ext.as<u64>(), *acc = p->acc.as<u64>(), *conv = p->conv.as<u64>();
    int rc;
    hipError_t e;
    TraceScope tr_ks(ctx, st, "KEYSWITCH");
    // INTT of the L input limbs (the "3 INTT" that open KEYSWITCH in the L=4 trace, 16384_4:468-470)
    HIP_TRY(hipMemcpyAsync(coef, d_c, L * N * 8, hipMemcpyDeviceToDevice, st));
    if ((rc = ntt_batch(ctx, coef, t, 1, L, 0, st, true))) return rc;
    {
        // base extension of each digit to every other prime (MODREDUCTION, 16384_4:471-452), then their transforms
        // (one trace line for the phase; the nested NTT line precedes it, as the reference's tools expect of nested costs)
        TraceScope tr_mr(ctx, st, "MODREDUCTION");
        if (p->up_batched) {
            e = launch_baseconv_exact_jobs(st, p->up_jobs.as<BcJob>(), (u32)p->dnum, p->up_max_m, p->up_max_k, p->up[0]->dev.f64 != 0, N);
            if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact_jobs");
        } else {
            for (int d = 0; d < p->dnum; d++) {
                const size_t lo = (size_t)d * p->alpha, hi = std::min(L, lo + (size_t)p->alpha);
                e = launch_baseconv_exact(st, ext + (size_t)d * M * N, coef + lo * N, p->up[d]->dev, N, (u32)lo, (u32)(hi - lo));
                if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact");
            }
        }
        TraceScope tr_ntt(ctx, st, "NTT");
        for (int path = 0; path < 2; path++) {
            if (!p->ext_units[path]) continue;
            PassArgs a{ext, lp, 0u, 1u, p->ext_units[path], 1u, p->ext_map[path].as<UnitRef>()};
            if ((e = launch_ntt(st, a, p->log_n, false, path, 1)) != hipSuccess) return hip_fail(e, "launch_ntt");
        }
    }
    {
        // multiply-accumulate with the evaluation key (MULTEVK): all digits, both halves, one launch
        TraceScope tr_mk(ctx, st, "MULTEVK");
        const KsMacArgs ka{acc, ext, d_c, d_evk, lp, (u32)L, (u32)M, (u32)p->dnum, (u32)p->alpha, p->log_n};
        if ((e = launch_ks_mac(st, ka)) != hipSuccess) return hip_fail(e, "launch_ks_mac");
    }
    // mod-down by P (MODSWITCH, 16384_4:454-463): INTT of the special limbs, conversion to Q, NTT, subtract, times P^-1
    TraceScope tr_ms(ctx, st, "MODSWITCH");
    {
        // the K special limbs of both halves, in place inside acc ([2][M][N], polynomial stride M)
        TraceScope tr_ntt(ctx, st, "NTT");
        rc = for_each_run(t, K, L, [&](size_t off, size_t len, int path) -> int {
            PassArgs a{acc + (L + off) * N, lp, (u32)(L + off), (u32)len, (u32)(2 * len), (u32)M, nullptr};
            hipError_t e2 = launch_ntt(st, a, p->log_n, true, path, 1);
            return e2 == hipSuccess ? FHE_OK : hip_fail(e2, "launch_ntt");
        });
        if (rc) return rc;
    }
    for (int h = 0; h < 2; h++) {
        u64 *tP = acc + ((size_t)h * M + L) * N;
        // BGV: remove delta = t * [acc * t^-1]_P instead of [acc]_P, so that delta = 0 mod t
        if (p->plain_modulus && (rc = fhe_scalar_affine(ctx, tP, tP, p->t_inv_P.data(), nullptr, t, 1, K, L, st))) return rc;
    }
    e = launch_baseconv_exact_jobs(st, p->down_jobs.as<BcJob>(), 2, p->down->dev.m, p->down->dev.k, p->down->dev.f64 != 0, N);
    if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact_jobs");
    if (p->plain_modulus && (rc = fhe_scalar_affine(ctx, conv, conv, p->t_mod_Q.data(), nullptr, t, 2, L, 0, st))) return rc;
    if ((rc = ntt_batch(ctx, conv, t, 2, L, 0, st, false))) return rc;
    const SubScaleArgs sa{d_out0, d_out1, acc, conv, d_add0, p->pinv.as<u64>(), (u64)(M * N), (u64)(L * N), lp, 0u, (u32)L, p->log_n, d_add1};
    if ((e = launch_sub_scale(st, sa)) != hipSuccess) return hip_fail(e, "launch_sub_scale");
    return FHE_OK;
}

extern "C" {

int fhe_rotate(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c0, const uint64_t *d_c1,
               uint32_t galois_elt, const uint64_t *d_galois_key, void *stream)
{
    if (!ctx || !p || !d_out0 || !d_out1 || !d_c0 || !d_c1 || !d_galois_key) return fail(FHE_ERR_INVALID, "null argument");
    if (d_out0 == d_c0 || d_out1 == d_c1) return fail(FHE_ERR_INVALID, "rotate is out of place");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    TraceScope tr(ctx, st, "ROTATE", true);
    const size_t L = p->L, N = (size_t)1 << p->log_n;
    u64 *sig0 = p->rot.as<u64>(), *sig1 = sig0 + L * N;
    // sigma on both parts (in the NTT domain a permutation of the slots), one launch
    hipError_t e = launch_automorphism_ntt(st, sig0, d_c0, (u32)L, p->log_n, galois_elt, sig1, d_c1);
    if (e != hipSuccess) return hip_fail(e, "launch_automorphism_ntt");
    // sigma(c1) is a ciphertext part under sigma(s): switch it back to s with the Galois key; the mod-down's last
    // launch adds sigma(c0) to the first part and writes both parts where the caller wants them
    return keyswitch_core(ctx, p, d_out0, d_out1, sig1, d_galois_key, sig0, nullptr, st);
}

} // extern "C"
