// capi_abft.cpp -- ABFT detector around the forward transform (part of the C ABI of include/fhe_mi355x.h; shared pieces in capi_internal.hpp)
#include "capi_internal.hpp"

extern "C" {

// ---------------------------------------------------------------- ABFT detector
int fhe_ctx_inject_fault(fhe_ctx *ctx, long long idx, int bit)
{
    if (!ctx || bit < 0 || bit > 63) return fail(FHE_ERR_INVALID, "bad fault");
    ctx->fault_idx = idx;
    ctx->fault_bit = bit;
    return FHE_OK;
}

int fhe_ctx_inject_fault_in_pass(fhe_ctx *ctx, int pass, uint32_t workgroup, uint32_t lds_word, int bit)
{
    if (!ctx || bit < 0 || bit > 63 || pass > 1) return fail(FHE_ERR_INVALID, "bad fault");
    ctx->pfault_pass = pass < 0 ? -1 : pass;
    ctx->pfault_block = workgroup;
    ctx->pfault_word = lds_word;
    ctx->pfault_bit = bit;
    return FHE_OK;
}

int fhe_abft_create(fhe_ctx *ctx, const fhe_ntt_tables *t, fhe_abft **out)
{
    if (!ctx || !t || !out) return fail(FHE_ERR_INVALID, "null argument");
    std::unique_ptr<fhe_abft> a(new fhe_abft);
    a->ctx = ctx;
    a->t = t;
    const size_t N = (size_t)1 << t->log_n;
    const u64 p = (u64)1 << (t->log_n / 2);
    std::vector<u64> w((size_t)t->count * N), u((size_t)t->count * N), ninv(t->count);
    for (int l = 0; l < t->count; l++) {
        const u64 q = t->q[l];
        u64 *wl = w.data() + (size_t)l * N, *ul = u.data() + (size_t)l * N;
        for (size_t i = 0; i < N; i++) wl[i] = ((i % p + 1) + (i / p + 1)) % q;      // generate_weights, negaclic_ntt.py:7-13
        // w_hat = V^-T w = N^-1 * NTT(u),  u = (w_0, -w_{N-1}, ..., -w_1)   (psi^N = -1)
        ul[0] = wl[0];
        for (size_t i = 1; i < N; i++) ul[i] = wl[N - i] ? q - wl[N - i] : 0;
        ninv[l] = host::inv_mod((u64)(N % q), q);
        if (!ninv[l]) return fail(FHE_ERR_INVALID, "N is not invertible modulo a table modulus");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(a->w.upload(w));
    HIP_TRY(a->what.upload(u));
    HIP_TRY(a->ninv.upload(ninv));
    int rc = ntt_batch(ctx, a->what.as<u64>(), t, 1, t->count, 0, nullptr, false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    {
        // twiddle-style encodings for the checksums that ride on the transform's passes (ntt_kernels.hip k_ntt_pass_abft)
        std::vector<u64> what((size_t)t->count * N);
        HIP_TRY(hipMemcpy(what.data(), a->what.p, what.size() * 8, hipMemcpyDeviceToHost));
        std::vector<Tw> ein(what.size()), eout(what.size());
        std::vector<u64> out8(what.size());
        for (int l = 0; l < t->count; l++) {
            const u64 q = t->q[l];
            for (size_t i = 0; i < N; i++) {
                const size_t k = (size_t)l * N + i;
                const u64 wo = host::mul_mod(what[k], ninv[l], q);
                ein[k] = t->path[l] == PATH_F64 ? ArithF64::encode(w[k], q) : ArithU64::encode(w[k], q);
                eout[k] = t->path[l] == PATH_F64 ? ArithF64::encode(wo, q) : ArithU64::encode(wo, q);
                out8[k] = wo;
            }
        }
        HIP_TRY(a->win.upload(ein));
        HIP_TRY(a->wout.upload(eout));
        HIP_TRY(a->wout8.upload(out8));
    }
    if (ntt_phases_supported(t->log_n)) {
        // Per-phase detector: weights on the words the column pass hands to the row pass.  The column pass is the same
        // length-2^PC negacyclic transform C (root phi = psi^(2^PR), bit-reversed output) on every column, so
        // u = P1^-T w has columns C^-T w_c = 2^-PC * NTT_{phi^-1}(w_c), and NTT_{phi^-1} is NTT_phi read at the
        // complemented position: u[k][c] = 2^-PC * (column pass of w)[~k][c].  The device's own column pass produces it.
        DevBuf wb;
        HIP_TRY(wb.upload(w));
        hipStream_t st = ctx->stream;
        rc = for_each_run(t, t->count, 0, [&](size_t off, size_t len, int path) -> int {
            PassArgs pa{wb.as<u64>() + off * N, t->d_lp.as<LimbParams>(), (u32)off, (u32)len, (u32)len, (u32)t->count};
            hipError_t e = launch_ntt(st, pa, t->log_n, false, path, 1, 0);
            return e == hipSuccess ? FHE_OK : hip_fail(e, "launch_ntt(column pass of the weights)");
        });
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<u64> y(w.size());
        HIP_TRY(hipMemcpy(y.data(), wb.p, y.size() * 8, hipMemcpyDeviceToHost));
        const int pc = ntt_column_stages(t->log_n), pr = t->log_n - pc;
        const size_t n1 = (size_t)1 << pc, n2 = (size_t)1 << pr;
        std::vector<Tw> eu(w.size());
        std::vector<u64> u8(w.size());
        for (int l = 0; l < t->count; l++) {
            const u64 q = t->q[l], n1inv = host::inv_mod((u64)(n1 % q), q);
            for (size_t k = 0; k < n1; k++)
                for (size_t c = 0; c < n2; c++) {
                    const u64 raw = y[(size_t)l * N + (n1 - 1 - k) * n2 + c];
                    u64 v;
                    if (t->path[l] == PATH_F64) {          // the hand-off holds raw FP64 bits of an exact integer in the lazy range
                        const long long sv = (long long)u64_bits_to_double(raw);
                        v = (u64)(((sv % (long long)q) + (long long)q) % (long long)q);
                    } else {
                        v = raw % q;                       // [0, 4q)
                    }
                    v = host::mul_mod(v, n1inv, q);
                    const size_t idx = (size_t)l * N + k * n2 + c;
                    eu[idx] = t->path[l] == PATH_F64 ? ArithF64::encode(v, q) : ArithU64::encode(v, q);
                    u8[idx] = v;
                }
        }
        HIP_TRY(a->umid.upload(eu));
        HIP_TRY(a->umid8.upload(u8));
    }
    *out = a.release();
    return FHE_OK;
}

int fhe_abft_destroy(fhe_abft *a)
{
    if (a) {
        (void)hipSetDevice(a->ctx->device);
        delete a;
    }
    return FHE_OK;
}

int fhe_abft_checksum(fhe_ctx *ctx, const fhe_abft *a, int side, const uint64_t *d_data, uint64_t *d_out, size_t n_poly,
                      size_t limbs, size_t start_idx, void *stream)
{
    if (!ctx || !a || !d_data || !d_out || (side != 0 && side != 1)) return fail(FHE_ERR_INVALID, "bad checksum arguments");
    int rc = check_range(a->t, n_poly, limbs, start_idx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_weighted_checksum(pick(ctx, stream), d_out, d_data, side ? a->what.as<u64>() : a->w.as<u64>(),
                                            side ? a->ninv.as<u64>() : nullptr, a->t->d_lp.as<LimbParams>(), (u32)start_idx,
                                            (u32)limbs, (u32)(n_poly * limbs), (u32)limbs, a->t->log_n);
    if (e != hipSuccess) return hip_fail(e, "launch_weighted_checksum");
    return FHE_OK;
}

int fhe_ntt_forward_checked(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, const fhe_abft *a, size_t n_poly,
                            size_t limbs, size_t start_idx, uint32_t *d_flags, void *stream)
{
    if (!ctx || !a || a->t != t || !d_flags) return fail(FHE_ERR_INVALID, "bad checked-transform arguments");
    const size_t units = n_poly * limbs;
    fhe_abft *m = const_cast<fhe_abft *>(a);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    u32 tin = 1, tout = 1;
    ntt_checked_tiles(t->log_n, &tin, &tout);
    if (m->sum_in.bytes < units * 8 * tin || m->sum_out.bytes < units * 8 * tout) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(m->sum_in.alloc(units * 16 * tin));
        HIP_TRY(m->sum_out.alloc(units * 16 * tout));
    }
    int rc;
    if (ctx->mode == 0 && ntt_checked_supported(t->log_n)) {
        // checksums fused into the transform's passes: no extra sweep over the data
        if ((rc = check_range(t, n_poly, limbs, start_idx))) return rc;
        if (!units) return FHE_OK;
        const size_t N = (size_t)1 << t->log_n;
        const bool hook = ctx->fault_idx >= 0 && t->log_n >= 13;
        rc = for_each_run(t, limbs, start_idx, [&](size_t off, size_t len, int path) -> int {
            PassArgs pa{d_data + off * N, t->d_lp.as<LimbParams>(), (u32)(start_idx + off), (u32)len, (u32)(n_poly * len), (u32)limbs, nullptr};
            u64 *si = m->sum_in.as<u64>() + off * tin, *so = m->sum_out.as<u64>() + off * tout;
            hipError_t e;
            if (hook) {
                // fault-injection hook: corrupt the intermediate between the two launches (one shot, after the last run's first pass)
                e = launch_ntt_checked(st, pa, a->win.as<Tw>(), a->wout.as<Tw>(), a->wout8.as<u64>(), si, so, t->log_n, path, 0);
                if (e == hipSuccess && off + len == limbs) e = launch_flip_bit(st, d_data, (u64)ctx->fault_idx, ctx->fault_bit);
                if (e != hipSuccess) return hip_fail(e, "launch_ntt_checked");
                return FHE_OK;
            }
            // batches that stream from HBM run as sub-batches, like the unchecked transform (capi.cpp ntt_batch); checksum slots are per unit
            if (const size_t per = sub_batch_polys(ctx, t->log_n, n_poly, len)) {
                u64 *pp = nullptr;
                const size_t pp_bytes = per * limbs * N * 8;
                // (scratch hand-off together with the non-temporal accesses on the pieces' external side, as the unchecked transform:
                // 430 us on the 512 MiB batch against 440 us with neither and 485-500 us with only one of the two, profiles/r02_abft_sweep.txt)
                if (ctx->pingpong > 0 || (ctx->pingpong < 0 && ctx->stream_hint != 0)) HIP_TRY(handoff_scratch(ctx, st, pp_bytes, &pp));
                return for_sub_batches(ctx, st, n_poly, per, pp ? pp_bytes : 0, [&](hipStream_t s, size_t p0, size_t cnt, u64 *side_tmp) {
                    PassArgs c = pa;
                    c.data = pa.data + p0 * limbs * N;
                    c.units = (u32)(cnt * len);
                    c.tmp = side_tmp ? side_tmp + off * N : pp ? pp + off * N : nullptr;
                    c.stream_hint = ctx->stream_hint != 0;
                    return launch_ntt_checked(s, c, a->win.as<Tw>(), a->wout.as<Tw>(), a->wout8.as<u64>(), si + p0 * limbs * tin, so + p0 * limbs * tout, t->log_n, path);
                });
            }
            e = launch_ntt_checked(st, pa, a->win.as<Tw>(), a->wout.as<Tw>(), a->wout8.as<u64>(), si, so, t->log_n, path);
            return e == hipSuccess ? FHE_OK : hip_fail(e, "launch_ntt_checked");
        });
        if (rc) return rc;
        if (hook) {
            ctx->fault_idx = -1;
            rc = for_each_run(t, limbs, start_idx, [&](size_t off, size_t len, int path) -> int {
                PassArgs pa{d_data + off * N, t->d_lp.as<LimbParams>(), (u32)(start_idx + off), (u32)len, (u32)(n_poly * len), (u32)limbs, nullptr};
                hipError_t e = launch_ntt_checked(st, pa, a->win.as<Tw>(), a->wout.as<Tw>(), a->wout8.as<u64>(), m->sum_in.as<u64>() + off * tin, m->sum_out.as<u64>() + off * tout,
                                                  t->log_n, path, 1);
                return e == hipSuccess ? FHE_OK : hip_fail(e, "launch_ntt_checked");
            });
            if (rc) return rc;
        }
        hipError_t e = launch_compare_sums(st, d_flags, m->sum_in.as<u64>(), tin, m->sum_out.as<u64>(), tout, t->d_lp.as<LimbParams>(),
                                           (u32)start_idx, (u32)limbs, (u32)units);
        if (e != hipSuccess) return hip_fail(e, "launch_compare_sums");
        return FHE_OK;
    }
    // separate reduction launches (tiny sizes, fused-NTT mode)
    if ((rc = fhe_abft_checksum(ctx, a, 0, d_data, m->sum_in.as<u64>(), n_poly, limbs, start_idx, st))) return rc;
    if ((rc = ntt_batch(ctx, d_data, t, n_poly, limbs, start_idx, st, false))) return rc;
    if ((rc = fhe_abft_checksum(ctx, a, 1, d_data, m->sum_out.as<u64>(), n_poly, limbs, start_idx, st))) return rc;
    hipError_t e = launch_compare_flags(st, d_flags, m->sum_in.as<u64>(), m->sum_out.as<u64>(), (u32)units);
    if (e != hipSuccess) return hip_fail(e, "launch_compare_flags");
    return FHE_OK;
}

int fhe_ntt_forward_checked_phases(fhe_ctx *ctx, uint64_t *d_data, const fhe_ntt_tables *t, const fhe_abft *a, size_t n_poly, size_t limbs,
                                   size_t start_idx, uint32_t *d_flags, void *stream)
{
    if (!ctx || !a || a->t != t || !d_flags || !d_data) return fail(FHE_ERR_INVALID, "bad checked-transform arguments");
    if (!ntt_phases_supported(t->log_n) || ctx->mode != 0)
        return fail(FHE_ERR_UNSUPPORTED, "per-phase checks belong to the two-launch transform (N >= 2^13); single-launch sizes have one phase: use fhe_ntt_forward_checked");
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    const size_t units = n_poly * limbs, N = (size_t)1 << t->log_n;
    if (!units) return FHE_OK;
    fhe_abft *m = const_cast<fhe_abft *>(a);
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    u32 tc = 1, tr = 1;
    ntt_checked_tiles(t->log_n, &tc, &tr);
    if (m->sum_in.bytes < units * 8 * tc || m->sum_out.bytes < units * 8 * tr || m->sum_mid1.bytes < units * 8 * tc || m->sum_mid2.bytes < units * 8 * tr) {
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(m->sum_in.alloc(units * 16 * tc));
        HIP_TRY(m->sum_mid1.alloc(units * 16 * tc));
        HIP_TRY(m->sum_mid2.alloc(units * 16 * tr));
        HIP_TRY(m->sum_out.alloc(units * 16 * tr));
    }
    // one-shot test hooks: a flip between the launches (fhe_ctx_inject_fault) and / or inside one pass (fhe_ctx_inject_fault_in_pass)
    const bool between = ctx->fault_idx >= 0;
    const int fpass = ctx->pfault_pass;
    auto run = [&](int which) -> int {
        return for_each_run(t, limbs, start_idx, [&](size_t off, size_t len, int path) -> int {
            PassArgs pa{d_data + off * N, t->d_lp.as<LimbParams>(), (u32)(start_idx + off), (u32)len, (u32)(n_poly * len), (u32)limbs, nullptr};
            PhaseArgs p1{a->win.as<Tw>(), a->umid.as<Tw>(), a->wout.as<Tw>(), a->umid8.as<u64>(), a->wout8.as<u64>(), t->log_n / 2,
                         m->sum_in.as<u64>() + off * tc, m->sum_mid1.as<u64>() + off * tc, fpass, ctx->pfault_block, ctx->pfault_word, ctx->pfault_bit};
            PhaseArgs p2 = p1;
            p2.sum_a = m->sum_mid2.as<u64>() + off * tr;
            p2.sum_b = m->sum_out.as<u64>() + off * tr;
            if (const size_t per = which < 0 && fpass < 0 ? sub_batch_polys(ctx, t->log_n, n_poly, len) : 0) {
                u64 *pp = nullptr;
                const size_t pp_bytes = per * limbs * N * 8;
                // (scratch hand-off together with the non-temporal accesses on the pieces' external side, as the unchecked transform:
                // 430 us on the 512 MiB batch against 440 us with neither and 485-500 us with only one of the two, profiles/r02_abft_sweep.txt)
                if (ctx->pingpong > 0 || (ctx->pingpong < 0 && ctx->stream_hint != 0)) HIP_TRY(handoff_scratch(ctx, st, pp_bytes, &pp));
                return for_sub_batches(ctx, st, n_poly, per, pp ? pp_bytes : 0, [&](hipStream_t s, size_t p0, size_t cnt, u64 *side_tmp) {
                    PassArgs c = pa;
                    c.data = pa.data + p0 * limbs * N;
                    c.units = (u32)(cnt * len);
                    c.tmp = side_tmp ? side_tmp + off * N : pp ? pp + off * N : nullptr;
                    c.stream_hint = ctx->stream_hint != 0;
                    PhaseArgs c1 = p1, c2 = p2;
                    c1.sum_a += p0 * limbs * tc, c1.sum_b += p0 * limbs * tc;
                    c2.sum_a += p0 * limbs * tr, c2.sum_b += p0 * limbs * tr;
                    return launch_ntt_phases(s, c, c1, c2, t->log_n, path, -1);
                });
            }
            hipError_t e = launch_ntt_phases(st, pa, p1, p2, t->log_n, path, which);
            return e == hipSuccess ? FHE_OK : hip_fail(e, "launch_ntt_phases");
        });
    };
    if (between) {
        if ((rc = run(0))) return rc;
        hipError_t e = launch_flip_bit(st, d_data, (u64)ctx->fault_idx, ctx->fault_bit);
        if (e != hipSuccess) return hip_fail(e, "launch_flip_bit");
        ctx->fault_idx = -1;
        if ((rc = run(1))) return rc;
    } else if ((rc = run(-1))) {
        return rc;
    }
    ctx->pfault_pass = -1;
    hipError_t e = launch_compare_phases(st, d_flags, m->sum_in.as<u64>(), m->sum_mid1.as<u64>(), tc, m->sum_mid2.as<u64>(), m->sum_out.as<u64>(), tr,
                                         t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs, (u32)units);
    if (e != hipSuccess) return hip_fail(e, "launch_compare_phases");
    return FHE_OK;
}

} // extern "C"
