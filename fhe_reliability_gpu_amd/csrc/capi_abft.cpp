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

} // extern "C"
