// capi_hmult.cpp -- homomorphic multiply on top of the key switch: tensor product, relinearisation, rescale
// (part of the C ABI of include/fhe_mi355x.h; shared pieces in capi_internal.hpp).  This is BASELINE config 4's
// composite ("hmult with baseConv"): phantom::multiply + relinearize_inplace + mod_switch_to_next_inplace of
// reliability_test/dotprod_test.cu:113-115, frontends MULTIPLY_CKKS / RELIN of the reference's SEAL traces
// (profile_framewk/build/data/ckks/16384_4:388-451).
#include "capi_internal.hpp"

extern "C" {

int fhe_tensor_product(fhe_ctx *ctx, uint64_t *d_d0, uint64_t *d_d1, uint64_t *d_d2, const uint64_t *d_a0, const uint64_t *d_a1,
                       const uint64_t *d_b0, const uint64_t *d_b1, const fhe_ntt_tables *t, size_t limbs, size_t start_idx, void *stream)
{
    if (!ctx || !d_d0 || !d_d1 || !d_d2 || !d_a0 || !d_a1 || !d_b0 || !d_b1) return fail(FHE_ERR_INVALID, "null argument");
    int rc = check_range(t, 1, limbs, start_idx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    TraceScope tr(ctx, st, "MULTIPLY_CKKS", true);
    const TensorArgs ta{d_d0, d_d1, d_d2, d_a0, d_a1, d_b0, d_b1, t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs, t->log_n};
    hipError_t e = launch_tensor(st, ta);
    if (e != hipSuccess) return hip_fail(e, "launch_tensor");
    return FHE_OK;
}

int fhe_relinearize(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_d0, const uint64_t *d_d1,
                    const uint64_t *d_d2, const uint64_t *d_relin_key, void *stream)
{
    if (!ctx || !p || !d_d0 || !d_d1 || !d_d2) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    TraceScope tr(ctx, pick(ctx, stream), "RELIN", true);
    return keyswitch_core(ctx, p, d_out0, d_out1, d_d2, d_relin_key, d_d0, d_d1, stream);
}

} // extern "C"

// (c - [c]_{q_last}) / q_last on every part: INTT of the last limb, its residues modulo the remaining primes, NTT,
// subtract, times q_last^-1.  BGV (plain modulus t set on the plan): the removed part is t * [c t^-1]_{q_last}.
// d_in = [n_parts][cn][N] (the rows this rank owns; one device: cn = L).  Two phases around the one exchange a sharded job
// needs: the owner of limb L-1 produces the last limbs in coefficient form (rs_bc), everybody consumes them.
static int rescale_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_in, size_t n_parts, hipStream_t st)
{
    if (!p->own_last) return FHE_OK;
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, cn = p->sh.cn, R = (size_t)p->L - 1, lr = R - p->sh.clo;   // lr: the last limb's row in this rank's slab
    int rc;
    {
        // out of place: the parts' last limbs (cn rows apart in the caller's buffer) straight into rs_bc ([n_parts][N]), no copy
        TraceScope tr_ntt(ctx, st, "NTT");
        PassArgs a{p->rs_bc, t->d_lp.as<LimbParams>(), (u32)R, 1u, (u32)n_parts, 1u};
        a.src = d_in + lr * N;
        a.src_stride = (u32)cn;
        hipError_t e = launch_ntt(st, a, p->log_n, true, t->path[R], 1);
        if (e != hipSuccess) return hip_fail(e, "launch_ntt");
    }
    if (p->plain_modulus && (rc = fhe_scalar_affine(ctx, p->rs_bc, p->rs_bc, &p->t_inv_qlast, nullptr, t, n_parts, 1, R, st))) return rc;
    return FHE_OK;
}

static int rescale_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *const *outs, const uint64_t *d_in, size_t n_parts, hipStream_t st)
{
    if (!p->rs_n) return FHE_OK;
    const fhe_ntt_tables *t = p->t;
    const size_t N = (size_t)1 << p->log_n, cn = p->sh.cn, R = p->rs_n, lo = p->sh.clo;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    u64 *delta = p->rs_delta.as<u64>();
    int rc;
    hipError_t e;
    const bool plain = !ntt_subscale_supported(p->log_n) || ctx->mode != 0 || ctx->fault_idx >= 0 || ctx->packed_on || ctx->only_pass >= 0 || ctx->resident;
    // a rescale converts ONE limb: at the two-launch sizes x mod q_j rides on the column pass of the residues' transform
    const bool trivial = p->log_n >= 13 && !plain;
    if (!trivial) {
        e = launch_baseconv_exact_jobs(st, p->rs_jobs.as<BcJob>(), (u32)n_parts, 1, (int)R, p->last->dev.f64 != 0, N, 1u);
        if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact_jobs");
        if (p->plain_modulus && (rc = fhe_scalar_affine(ctx, delta, delta, p->t_mod_Q.data() + lo, nullptr, t, n_parts, R, lo, st))) return rc;
    }
    if (plain) {
        if ((rc = ntt_batch(ctx, delta, t, n_parts, R, lo, st, false))) return rc;
        for (size_t part = 0; part < n_parts; part += 2) {
            const bool two = part + 1 < n_parts;
            const SubScaleArgs sa{outs[part], two ? outs[part + 1] : nullptr, d_in + part * cn * N, delta + part * R * N, nullptr,
                                  p->qlast_inv.as<u64>(), (u64)(cn * N), (u64)(R * N), lp, (u32)lo, (u32)R, p->log_n, nullptr};
            if ((e = launch_sub_scale(st, sa)) != hipSuccess) return hip_fail(e, "launch_sub_scale");
        }
        return FHE_OK;
    }
    // forward transform of the residues with (c - delta) / q_last riding on its last pass
    TraceScope tr_ntt(ctx, st, "NTT");
    return for_each_run(t, R, lo, [&](size_t off, size_t len, int path) -> int {
        PassArgs a{delta + off * N, lp, (u32)(lo + off), (u32)len, (u32)(n_parts * len), (u32)R};
        RowEpiArgs ep{{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, d_in + off * N, (u64)(cn * N), p->qlast_inv.as<u64>() + off};
        for (size_t i = 0; i < n_parts; i++) ep.out[i] = outs[i] + off * N;
        if (trivial) {
            a.src = p->rs_bc;
            a.src_bcast = N;
            if (p->plain_modulus) ep.pre = p->d_t_mod_Q.as<u64>() + lo + off;          // BGV: delta = t * [c t^-1]_{q_last}
        }
        hipError_t e2 = launch_ntt_subscale(st, a, ep, p->log_n, path);
        return e2 == hipSuccess ? FHE_OK : hip_fail(e2, "launch_ntt_subscale");
    });
}

static int rescale_check(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_in, size_t n_parts)
{
    if (!ctx || !p || (!d_in && p->sh.cn)) return fail(FHE_ERR_INVALID, "null argument");
    if (n_parts < 1 || n_parts > 3) return fail(FHE_ERR_INVALID, "a ciphertext has 1 to 3 parts");
    if (p->L < 2) return fail(FHE_ERR_INVALID, "no prime left to drop");
    if (!p->rs_bc) return fail(FHE_ERR_INVALID, "this sharded plan was created without a broadcast buffer for the rescale");
    return FHE_OK;
}

// Input parts are cn rows apart, output parts rs_n rows: in place the strides differ and later parts would be read after they were
// overwritten.  Any overlap of an output part with the input is refused.
static int rescale_overlap(const fhe_keyswitch *p, uint64_t *const *outs, const uint64_t *d_in, size_t n_parts)
{
    const size_t N = (size_t)1 << p->log_n, in_words = n_parts * (size_t)p->sh.cn * N, out_words = (size_t)p->rs_n * N;
    for (size_t i = 0; i < n_parts; i++)
        if (outs[i] && outs[i] < d_in + in_words && d_in < outs[i] + out_words) return fail(FHE_ERR_INVALID, "rescale is out of place");
    return FHE_OK;
}

static int rescale_core(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *const *outs, const uint64_t *d_in, size_t n_parts, void *stream)
{
    int rc = rescale_check(ctx, p, d_in, n_parts);
    if (rc) return rc;
    if (p->sharded) return fail(FHE_ERR_INVALID, "a sharded plan rescales through fhe_rescale_shard_begin / _finish with the broadcast between them");
    for (size_t i = 0; i < n_parts; i++)
        if (!outs[i]) return fail(FHE_ERR_INVALID, "null argument");
    if (int rc2 = rescale_overlap(p, outs, d_in, n_parts)) return rc2;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    TraceScope tr(ctx, st, "RESCALE", true);
    if ((rc = rescale_begin(ctx, p, d_in, n_parts, st))) return rc;
    return rescale_finish(ctx, p, outs, d_in, n_parts, st);
}

extern "C" {

int fhe_rescale(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out, const uint64_t *d_in, size_t n_parts, void *stream)
{
    if (!p || !d_out || p->L < 2) return fail(FHE_ERR_INVALID, !p || !d_out ? "null argument" : "no prime left to drop");
    const size_t step = (size_t)(p->L - 1) << p->log_n;
    uint64_t *outs[3] = {d_out, d_out + step, d_out + 2 * step};
    return rescale_core(ctx, p, outs, d_in, n_parts, stream);
}

int fhe_rescale_shard_begin(fhe_ctx *ctx, fhe_keyswitch *p, const uint64_t *d_in_local, size_t n_parts, void *stream)
{
    int rc = rescale_check(ctx, p, d_in_local, n_parts);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    return rescale_begin(ctx, p, d_in_local, n_parts, pick(ctx, stream));
}

int fhe_rescale_shard_finish(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out_local, const uint64_t *d_in_local, size_t n_parts, void *stream)
{
    int rc = rescale_check(ctx, p, d_in_local, n_parts);
    if (rc) return rc;
    if (!d_out_local && p->rs_n) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t step = (size_t)p->rs_n << p->log_n;
    uint64_t *outs[3] = {d_out_local, d_out_local + step, d_out_local + 2 * step};
    if (p->rs_n)
        if (int rc2 = rescale_overlap(p, outs, d_in_local, n_parts)) return rc2;
    return rescale_finish(ctx, p, outs, d_in_local, n_parts, pick(ctx, stream));
}

int fhe_rescale_shard_info(const fhe_keyswitch *p, int *owns_last, int *out_rows)
{
    if (!p) return fail(FHE_ERR_INVALID, "null plan");
    if (owns_last) *owns_last = p->own_last ? 1 : 0;
    if (out_rows) *out_rows = p->rs_n;
    return FHE_OK;
}

int fhe_hmult(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_a0, const uint64_t *d_a1,
              const uint64_t *d_b0, const uint64_t *d_b1, const uint64_t *d_relin_key, int rescale, void *stream)
{
    if (!ctx || !p || !d_out0 || !d_out1 || !d_relin_key) return fail(FHE_ERR_INVALID, "null argument");
    if (rescale && p->L < 2) return fail(FHE_ERR_INVALID, "no prime left to drop");
    if (p->sharded) return fail(FHE_ERR_UNSUPPORTED, "fhe_hmult runs on one device; a sharded job runs fhe_tensor_product, the fhe_keyswitch_shard_* phases and fhe_rescale_shard_* on its rows");
    // (the operands are read by the first launch only, the outputs written by the last: an output may reuse an operand's buffer; the
    // two outputs must be distinct)
    if (d_out0 == d_out1) return fail(FHE_ERR_INVALID, "the two output parts must be distinct buffers");
    const size_t N = (size_t)1 << p->log_n, L = p->L;
    u64 *d0 = p->hm.as<u64>(), *d1 = d0 + L * N, *d2 = d1 + L * N, *pre = p->hm_pre.as<u64>();
    int rc;
    if ((rc = fhe_tensor_product(ctx, d0, d1, d2, d_a0, d_a1, d_b0, d_b1, p->t, L, 0, stream))) return rc;
    if (!rescale) return fhe_relinearize(ctx, p, d_out0, d_out1, d0, d1, d2, d_relin_key, stream);
    if (ks_rescale_fusable(ctx, p)) {
        // the mod-down and the rescale share one forward transform (capi_keyswitch.cpp ks_finish_rescale): same words as the two steps below
        HIP_TRY(hipSetDevice(ctx->device));
        return keyswitch_core(ctx, p, d_out0, d_out1, d2, d_relin_key, d0, d1, stream, true);
    }
    if ((rc = fhe_relinearize(ctx, p, pre, pre + L * N, d0, d1, d2, d_relin_key, stream))) return rc;
    uint64_t *outs[3] = {d_out0, d_out1, nullptr};
    return rescale_core(ctx, p, outs, pre, 2, stream);
}

} // extern "C"
