// capi.cpp -- the C ABI declared in include/fhe_mi355x.h: handle management, table
// construction and launch sequencing.  No arithmetic on residues happens on the host
// here; there is no CPU fallback -- every transform call ends in a HIP kernel launch
// or an error.
#include "capi_internal.hpp"

// Build device tables from forward tables in canonical residues (count x N).
// gs_scale (optional, one per limb): "natural-order" table set for launch_ntt_gs -- the INVERSE slot receives the given rows
// themselves (the inverse-structured network then computes the transposed transform: bit-reversed in, natural out, same
// twiddles) with entry 0 = scale * rows[1] and inv_n = scale, so that the last stage multiplies the result by `scale`
// (1 for a forward cyclic transform, n^-1 for motivation/bsgs.py:31-36's intt).  No twiddle needs an inverse: any modulus.
int build_tables(fhe_ctx *ctx, int log_n, const u64 *q, int count, const u64 *fwd_rows, bool want_inverse, int force_path,
                 const u64 *psi_or_null, fhe_ntt_tables **out, const u64 *gs_scale)
{
    if (!ctx || !q || !out || count < 1 || log_n < 1 || log_n > NTT_MAX_LOGN) return fail(FHE_ERR_INVALID, "bad table arguments");
    const size_t N = (size_t)1 << log_n;
    std::unique_ptr<fhe_ntt_tables> t(new fhe_ntt_tables);
    t->ctx = ctx;
    t->log_n = log_n;
    t->count = count;
    t->q.assign(q, q + count);
    t->psi.assign(count, 0);
    t->path.resize(count);
    t->h_lp.resize(count);
    t->has_inverse = want_inverse;
    std::vector<Tw> tw((size_t)count * 2 * N);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(t->d_tw.alloc(tw.size() * sizeof(Tw)));
    std::vector<u64> inv;
    for (int l = 0; l < count; l++) {
        const int path = force_path >= 0 ? force_path : path_for(q[l]);
        if (path < 0 || (path == PATH_F64 && q[l] >= ((u64)1 << 50)) || q[l] >= ((u64)1 << 61) || q[l] < 2)
            return fail(FHE_ERR_UNSUPPORTED, "modulus must satisfy 2 <= q < 2^61 (FP64 path: q < 2^50)");
        t->path[l] = path;
        const u64 *row = fwd_rows + (size_t)l * N;
        Tw *f = tw.data() + (size_t)l * 2 * N, *b = f + N;
        for (size_t k = 0; k < N; k++) f[tw_stored_index(log_n, (u32)k)] = encode(path, row[k] % q[l], q[l]);
        bool inv_ok = false;
        if (gs_scale) {
            const u64 sc = gs_scale[l] % q[l];
            for (size_t k = 1; k < N; k++) b[tw_stored_index(log_n, (u32)k)] = encode(path, row[k] % q[l], q[l]);
            b[tw_stored_index(log_n, 0)] = encode(path, host::mul_mod(sc, row[1 % N] % q[l], q[l]), q[l]);
        } else if (want_inverse) inv_ok = batch_inverse(row, N, q[l], inv);
        if (gs_scale) {
        } else if (inv_ok) {
            // entry 0 is not a twiddle of the network: it carries N^-1 times the last inverse stage's twiddle (entry 1),
            // which that stage uses to scale while it multiplies (ntt_core.hpp radix_inv FOLD)
            inv[0] = host::mul_mod(host::inv_mod((u64)(N % q[l]), q[l]), inv[1 % N], q[l]);
            for (size_t k = 0; k < N; k++) b[tw_stored_index(log_n, (u32)k)] = encode(path, inv[k], q[l]);
        } else {
            for (size_t k = 0; k < N; k++) b[k] = f[k];
            t->has_inverse = false;
        }
        LimbParams &p = t->h_lp[l];
        std::memset(&p, 0, sizeof p);
        p.q = q[l];
        p.two_q = 2 * q[l];
        p.n = (double)q[l];
        p.ninv = 1.0 / p.n;
        const u64 ninv = gs_scale ? gs_scale[l] % q[l] : host::inv_mod((u64)(N % q[l]), q[l]);
        if (!ninv && want_inverse) t->has_inverse = false;
        p.inv_n = encode(path, ninv, q[l]);
        p.fwd = t->d_tw.as<Tw>() + (size_t)l * 2 * N;
        p.inv = p.fwd + N;
        u64 cr[3];
        host::const_ratio(q[l], cr);
        p.barrett_lo = cr[0];
        p.barrett_hi = cr[1];
        p.path = path;
        if (psi_or_null) t->psi[l] = psi_or_null[l];
    }
    HIP_TRY(hipMemcpy(t->d_tw.p, tw.data(), tw.size() * sizeof(Tw), hipMemcpyHostToDevice));
    HIP_TRY(t->d_lp.upload(t->h_lp));
    *out = t.release();
    return FHE_OK;
}

// Sub-batch policy of the two-launch transforms ("ntt_chunk_mib"): batches that cannot stay in the 256 MiB Infinity Cache between
// the two launches are cut into sub-batches that can (only those: cutting a 128 MiB batch costs 15 %).  A piece is `pc` polynomials
// x `lc` limbs of the run: as many polynomials of as few limbs as fit, so that a piece works with few twiddle tables (a table is
// fetched once per piece and XCD: 16 limbs x 8 polynomials per piece cost 0.27 of the roofline where 2 limbs x 64 cost 0.30).
// pc = 0: one launch pair for the whole batch.
// bufs: buffers of the batch's shape a piece works on (a negacyclic product: both operands and the result).
SubBatchCut sub_batch_cut(const fhe_ctx *ctx, int log_n, size_t n_poly, size_t len, size_t bufs)
{
    // (a product's pieces run four launches each: they pay from twice the size and only on batches well past the cache --
    // profiles/r02_polymul_sweep.txt: 384 MiB of operands + result 0.46 whole / 0.45 cut, 1.5 GiB 0.47 whole / 0.51 in 128 MiB pieces)
    const size_t unit_bytes = ((size_t)8 << log_n) * bufs, total = n_poly * len * unit_bytes, chunk_bytes = ((size_t)ctx->chunk_mib << 20) * (bufs > 1 ? 2 : 1);
    const size_t floor_bytes = ((size_t)ctx->chunk_floor_mib << 20) * (bufs > 1 ? 3 : 1);
    if (!chunk_bytes || log_n < 13 || n_poly * len < 2 || total <= std::max(chunk_bytes + (chunk_bytes >> 1), floor_bytes)) return SubBatchCut{0, 0};
    const size_t pc = std::min(n_poly, std::max<size_t>(1, chunk_bytes / unit_bytes));
    const size_t lc = std::min(len, std::max<size_t>(1, chunk_bytes / (unit_bytes * pc)));
    if (pc == n_poly && lc == len) return SubBatchCut{0, 0};
    return SubBatchCut{pc, lc};
}
// the same policy for callers that cut along the polynomials only (whole limbs of the run in every piece)
size_t sub_batch_polys(const fhe_ctx *ctx, int log_n, size_t n_poly, size_t len)
{
    const size_t unit_bytes = (size_t)8 << log_n, total = n_poly * len * unit_bytes, chunk_bytes = (size_t)ctx->chunk_mib << 20;
    if (!chunk_bytes || log_n < 13 || n_poly < 2 || total <= std::max(chunk_bytes + (chunk_bytes >> 1), (size_t)ctx->chunk_floor_mib << 20)) return 0;
    const size_t per = std::max<size_t>(1, chunk_bytes / (unit_bytes * len));
    return per < n_poly ? per : 0;
}

// Per-stream hand-off scratch of the ping-pong transforms ("ntt_pingpong"); *out = null inside a stream capture that would have to
// allocate (the launches then hand over in place).
hipError_t handoff_scratch(fhe_ctx *ctx, hipStream_t st, size_t bytes, u64 **out)
{
    DevBuf *b;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto &slot = ctx->pp_tmp[st];
        if (!slot) slot.reset(new DevBuf);
        b = slot.get();
    }
    *out = nullptr;
    if (b->bytes < bytes) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) return hipSuccess;
        hipError_t e = b->p ? hipStreamSynchronize(st) : hipSuccess;      // growing frees the old block
        if (e == hipSuccess) e = b->alloc(bytes);
        if (e != hipSuccess) return e;
    }
    *out = b->as<u64>();
    return hipSuccess;
}

// the side stream the context keeps for work forked off `st` (created on first use); *out = nullptr inside a stream capture
int side_stream(fhe_ctx *ctx, hipStream_t st, fhe_ctx::Side **out)
{
    *out = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return FHE_OK;
    fhe_ctx::Side *sd;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto &slot = ctx->side[st];
        if (!slot) slot.reset(new fhe_ctx::Side);
        sd = slot.get();
    }
    if (!sd->s) {
        HIP_TRY(hipStreamCreateWithFlags(&sd->s, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sd->join, hipEventDisableTiming));
    }
    *out = sd;
    return FHE_OK;
}

// fn(stream, piece, side scratch) over the pieces of a call; with "ntt_split" they alternate between the caller's stream and the
// context's side stream for it (fork / join by events; side scratch = side_tmp_bytes of the side stream's own, null on the caller's
// stream), so that one piece's row pass runs under the next one's column pass.
int for_pieces(fhe_ctx *ctx, hipStream_t st, size_t n_pieces, size_t side_tmp_bytes, const std::function<hipError_t(hipStream_t, size_t, u64 *)> &fn)
{
    fhe_ctx::Side *sd = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ctx->split && n_pieces > 1 && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
        {
            std::lock_guard<std::mutex> lock(ctx->mu);
            auto &slot = ctx->side[st];
            if (!slot) slot.reset(new fhe_ctx::Side);
            sd = slot.get();
        }
        if (!sd->s) {
            HIP_TRY(hipStreamCreateWithFlags(&sd->s, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&sd->fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sd->join, hipEventDisableTiming));
        }
        if (sd->tmp.bytes < side_tmp_bytes) {
            HIP_TRY(hipStreamSynchronize(sd->s));
            HIP_TRY(sd->tmp.alloc(side_tmp_bytes));
        }
        HIP_TRY(hipEventRecord(sd->fork, st));
        HIP_TRY(hipStreamWaitEvent(sd->s, sd->fork, 0));
    }
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < n_pieces && e == hipSuccess; ++i) {
        const bool on_side = sd && (i & 1);
        e = fn(on_side ? sd->s : st, i, on_side && side_tmp_bytes ? sd->tmp.as<u64>() : nullptr);
    }
    if (sd) {   // join even after a failed launch: the side stream must not be left forked
        HIP_TRY(hipEventRecord(sd->join, sd->s));
        HIP_TRY(hipStreamWaitEvent(st, sd->join, 0));
    }
    return e == hipSuccess ? FHE_OK : hip_fail(e, "sub-batch launch");
}
// pieces of `per` polynomials: fn(stream, first polynomial, count, side scratch)
int for_sub_batches(fhe_ctx *ctx, hipStream_t st, size_t n_poly, size_t per, size_t side_tmp_bytes,
                    const std::function<hipError_t(hipStream_t, size_t, size_t, u64 *)> &fn)
{
    return for_pieces(ctx, st, (n_poly + per - 1) / per, side_tmp_bytes,
                      [&](hipStream_t s, size_t i, u64 *side_tmp) { return fn(s, i * per, std::min(per, n_poly - i * per), side_tmp); });
}

// d_src (optional): out-of-place -- the input is read from there (same layout), nothing is copied (PassArgs::src)
int ntt_batch(fhe_ctx *ctx, u64 *d, const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx, void *stream,
              bool inverse, const u64 *d_src, u32 galois, u64 *d_galois_copy)
{
    if (!ctx || !d) return fail(FHE_ERR_INVALID, "null argument");
    if (galois) {
        // the input is sigma_k(d_src) (NTT-domain Galois map): on the load of the inverse transform's first launch where that launch
        // stages its tile (PassArgs::galois), as a launch of its own otherwise
        if (!inverse || !d_src || d_src == d || !(galois & 1)) return fail(FHE_ERR_INVALID, "the Galois map rides on an out-of-place inverse transform");
        if (ctx->mode == 1 || ctx->fault_idx >= 0 || ctx->packed_on || ctx->only_pass >= 0 || ctx->resident || t->log_n < 5) {
            HIP_TRY(hipSetDevice(ctx->device));
            u64 *to = d_galois_copy ? d_galois_copy : d;
            hipError_t e = launch_automorphism_ntt(pick(ctx, stream), to, d_src, (u32)(n_poly * limbs), t->log_n, galois);
            if (e != hipSuccess) return hip_fail(e, "launch_automorphism_ntt");
            d_src = d_galois_copy;        // (null: sigma(src) already sits in d, transform in place)
            galois = 0;
            d_galois_copy = nullptr;
        }
    }
    if (d_src && (ctx->mode == 1 || ctx->fault_idx >= 0 || ctx->packed_on || ctx->only_pass >= 0 || ctx->resident)) {
        // the experimental variants and the test hooks work in place: copy first
        HIP_TRY(hipSetDevice(ctx->device));
        if (d_src != d) HIP_TRY(hipMemcpyAsync(d, d_src, (n_poly * limbs << t->log_n) * 8, hipMemcpyDeviceToDevice, pick(ctx, stream)));
        d_src = nullptr;
    }
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    if (inverse && !t->has_inverse) return fail(FHE_ERR_UNSUPPORTED, "table set has no inverse (twiddle or N not invertible)");
    if (!n_poly || !limbs) return FHE_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const size_t N = (size_t)1 << t->log_n;
    TraceScope tr(ctx, st, "NTT");
    return for_each_run(t, limbs, start_idx, [&](size_t off, size_t len, int path) -> int {
        PassArgs a{d + off * N, t->d_lp.as<LimbParams>(), (u32)(start_idx + off), (u32)len, (u32)(n_poly * len), (u32)limbs};
        if (d_src) a.src = d_src + off * N;
        a.galois = galois;
        if (d_galois_copy) a.galois_copy = d_galois_copy + off * N;
        hipError_t e;
        if (ctx->mode == 1 && fused_supported(t->log_n)) {
            DevBuf *ctl;
            {
                std::lock_guard<std::mutex> lock(ctx->mu);
                auto &slot = ctx->fused_ctl[st];
                if (!slot) slot.reset(new DevBuf);
                ctl = slot.get();
            }
            const size_t need = fused_ctl_bytes(a.units);
            if (ctl->bytes < need) {
                // growing frees the old block: make sure no launch on this stream still uses it
                HIP_TRY(hipStreamSynchronize(st));
                u32 carried = 0;       // an error word set by an earlier launch moves to the new block
                if (ctl->p) HIP_TRY(hipMemcpy(&carried, ctl->as<u32>() + fused_error_word(), sizeof carried, hipMemcpyDeviceToHost));
                HIP_TRY(ctl->alloc(need * 2));
                HIP_TRY(hipMemset(ctl->p, 0, need * 2));
                if (carried) HIP_TRY(hipMemcpy(ctl->as<u32>() + fused_error_word(), &carried, sizeof carried, hipMemcpyHostToDevice));
            }
            e = launch_ntt_fused(st, a, t->log_n, inverse, path, ctl->as<u32>(), ctx->fused_dist, ctx->fused_wgs, ctx->fused_variant, ctx->fused_skip_teams);
        } else if (ctx->fault_idx >= 0 && t->log_n >= 13) {
            // fault-injection hook: corrupt the intermediate between the two launches (one shot)
            e = launch_ntt(st, a, t->log_n, inverse, path, ctx->geo, 0);
            if (e == hipSuccess) e = launch_flip_bit(st, d, (u64)ctx->fault_idx, ctx->fault_bit);
            if (e == hipSuccess) e = launch_ntt(st, a, t->log_n, inverse, path, ctx->geo, 1);
            ctx->fault_idx = -1;
        } else {
            if (ctx->packed_on && ctx->only_pass < 0 && ctx->geo == 1 && ntt_packed_supported(t->log_n, inverse, path)) {
                DevBuf *sc;
                {
                    std::lock_guard<std::mutex> lock(ctx->mu);
                    auto &slot = ctx->packed[st];
                    if (!slot) slot.reset(new DevBuf);
                    sc = slot.get();
                }
                const size_t need = (size_t)a.units * ntt_packed_scratch_words() * 8;
                if (sc->bytes < need) {
                    HIP_TRY(hipStreamSynchronize(st));     // growing frees the old block
                    HIP_TRY(sc->alloc(need));
                }
                a.scratch = sc->as<u64>();
            }
            // Batches that cannot stay in the 256 MiB Infinity Cache between the two launches run as sub-batches that can: a sub-batch's
            // second launch then finds the first one's output on-die (512 MiB / 2 GiB batches: 0.30 -> 0.34 of the roofline,
            // profiles/r02_chunk_sweep.txt, r02_split_sweep.txt).  Limb-major launch order keeps whole limbs together.
            const bool plain = t->log_n >= 13 && ctx->only_pass < 0 && !a.scratch;
            const SubBatchCut cut = plain ? sub_batch_cut(ctx, t->log_n, n_poly, len) : SubBatchCut{0, 0};
            u64 *pp = nullptr;
            // hand-off through a per-stream scratch (both launches out of place); by default only for calls that are sub-batched, i.e.
            // stream from HBM anyway: for a batch that fits the Infinity Cache the scratch would double the footprint and push it out
            const bool want_pp = !galois && (ctx->pingpong < 0 ? cut.pc != 0 : ctx->pingpong != 0);
            // one piece, compact ([pc][lc][N]), or the whole call in the data's own layout
            const size_t pp_bytes = (cut.pc ? cut.pc * cut.lc : n_poly * limbs) * N * 8;
            if (want_pp && plain) HIP_TRY(handoff_scratch(ctx, st, pp_bytes, &pp));
            if (cut.pc) {
                const size_t npp = (n_poly + cut.pc - 1) / cut.pc, npl = (len + cut.lc - 1) / cut.lc;
                return for_pieces(ctx, st, npp * npl, pp ? pp_bytes : 0, [&](hipStream_t s, size_t i, u64 *side_tmp) {
                    // (pieces of one limb window follow each other: its tables stay in the L2s)
                    const size_t l0 = (i / npp) * cut.lc, p0 = (i % npp) * cut.pc;
                    const size_t lc = std::min(cut.lc, len - l0), pc = std::min(cut.pc, n_poly - p0);
                    PassArgs c = a;
                    c.data = a.data + (p0 * limbs + l0) * N;
                    if (a.src) c.src = a.src + (p0 * limbs + l0) * N;
                    if (a.galois_copy) c.galois_copy = a.galois_copy + (p0 * limbs + l0) * N;
                    c.limb0 = a.limb0 + (u32)l0;
                    c.limbs = (u32)lc;
                    c.units = (u32)(pc * lc);
                    c.tmp = side_tmp ? side_tmp : pp;
                    c.tmp_stride = (u32)lc;
                    c.stream_hint = ctx->stream_hint != 0;       // a piece of a batch that streams from HBM
                    return launch_ntt(s, c, t->log_n, inverse, path, ctx->geo, -1, false);
                });
            } else {
                a.tmp = pp ? pp + off * N : nullptr;
                // single-launch sizes have nothing to hand over and are never cut: a batch of theirs that cannot stay in the Infinity
                // Cache is touched once in, once out -- non-temporal accesses, as for the pieces above
                const bool streams = t->log_n < 13 && t->log_n >= 5 && ctx->stream_hint < 0 && ctx->only_pass < 0 &&
                                     (size_t)a.units * N * 8 > ((size_t)ctx->chunk_floor_mib << 20);
                a.stream_hint = ctx->stream_hint > 0 || streams;
                e = launch_ntt(st, a, t->log_n, inverse, path, ctx->geo, ctx->only_pass, ctx->resident);
            }
        }
        if (e != hipSuccess) return hip_fail(e, "launch_ntt");
        return FHE_OK;
    });
}

int pointwise(fhe_ctx *ctx, u64 *c, const u64 *a, const u64 *b, const fhe_ntt_tables *t, size_t n_poly, size_t limbs,
              size_t start_idx, void *stream, bool acc)
{
    if (!ctx || !c || !a || !b) return fail(FHE_ERR_INVALID, "null argument");
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    PointwiseArgs p{c, a, b, t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs, (u32)(n_poly * limbs), (u32)limbs, t->log_n};
    hipError_t e = launch_modmul(pick(ctx, stream), p, acc);
    if (e != hipSuccess) return hip_fail(e, "launch_modmul");
    return FHE_OK;
}

// natural-order table set (GS mode) of a cyclic transform, cached per context
int cyclic_tables(fhe_ctx *ctx, int log_n, u64 mod, u64 root, int convention, u64 scale, fhe_ntt_tables **out)
{
    std::lock_guard<std::mutex> lock(ctx->mu);
    auto key = std::make_tuple(log_n, mod, root, convention * 2 + 1, scale);
    auto it = ctx->cyclic.find(key);
    if (it == ctx->cyclic.end()) {
        const u64 n = (u64)1 << log_n;
        std::vector<u64> tw(n);
        host::cyclic_table(mod, log_n, root, convention == 1, tw.data());
        fhe_ntt_tables *nt = nullptr;
        int rc = build_tables(ctx, log_n, &mod, 1, tw.data(), false, -1, nullptr, &nt, &scale);
        if (rc) return rc;
        it = ctx->cyclic.emplace(key, std::unique_ptr<fhe_ntt_tables>(nt)).first;
    }
    *out = it->second.get();
    return FHE_OK;
}

// lengths below 2^5: the forward network with a cyclic table, then the bit-reversal gather (three tiny launches)
static int cyclic_small(fhe_ctx *ctx, u64 *d_data, u64 *d_scratch, int log_n, size_t n_vec, u64 mod, u64 root, int convention, u64 scale, bool do_scale,
                        void *stream)
{
    fhe_ntt_tables *t = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto key = std::make_tuple(log_n, mod, root, convention * 2, (u64)0);
        auto it = ctx->cyclic.find(key);
        if (it == ctx->cyclic.end()) {
            std::vector<u64> tw((size_t)1 << log_n);
            host::cyclic_table(mod, log_n, root, convention == 1, tw.data());
            fhe_ntt_tables *nt = nullptr;
            int rc = build_tables(ctx, log_n, &mod, 1, tw.data(), false, -1, nullptr, &nt);
            if (rc) return rc;
            it = ctx->cyclic.emplace(key, std::unique_ptr<fhe_ntt_tables>(nt)).first;
        }
        t = it->second.get();
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    PassArgs a{d_data, t->d_lp.as<LimbParams>(), 0u, 1u, (u32)n_vec, 1u};
    hipError_t e = launch_ntt(st, a, log_n, false, t->path[0]);
    if (e != hipSuccess) return hip_fail(e, "launch_ntt(cyclic)");
    e = launch_bitrev_scale(st, d_scratch, d_data, log_n, (u32)n_vec, mod_const(mod), scale, do_scale);
    if (e != hipSuccess) return hip_fail(e, "launch_bitrev_scale");
    HIP_TRY(hipMemcpyAsync(d_data, d_scratch, (n_vec << log_n) * sizeof(u64), hipMemcpyDeviceToDevice, st));
    return FHE_OK;
}

extern "C" {

int fhe_version(void) { return 100; }
const char *fhe_last_error(void) { return g_err.c_str(); }

int fhe_ctx_create(int device, fhe_ctx **out)
{
    if (!out) return fail(FHE_ERR_INVALID, "null out");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(FHE_ERR_INVALID, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<fhe_ctx> c(new fhe_ctx);
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    // tuning knobs: FHE_NTT_MODE=twopass|fused, FHE_FUSED_DIST, FHE_FUSED_WGS, FHE_NTT_RESIDENT=0|1
    if (const char *m = getenv("FHE_NTT_MODE")) c->mode = std::strcmp(m, "fused") == 0 ? 1 : 0;
    if (const char *v = getenv("FHE_FUSED_DIST")) c->fused_dist = (unsigned)std::max(1, atoi(v));
    if (const char *v = getenv("FHE_FUSED_WGS")) c->fused_wgs = (unsigned)std::max(1, atoi(v));
    if (const char *v = getenv("FHE_NTT_RESIDENT")) c->resident = atoi(v) != 0;
    if (const char *v = getenv("FHE_NTT_PACKED")) c->packed_on = atoi(v) != 0;
    if (const char *v = getenv("FHE_NTT_PINGPONG")) c->pingpong = atoi(v) < 0 ? -1 : atoi(v) ? 1 : 0;
    if (const char *v = getenv("FHE_NTT_SPLIT")) c->split = atoi(v) < 0 ? -1 : atoi(v) ? 1 : 0;
    if (const char *v = getenv("FHE_NTT_CHUNK_MIB")) c->chunk_mib = (unsigned)std::max(0, atoi(v));
    if (const char *v = getenv("FHE_KS_FUSED")) c->ks_fused = atoi(v) < 0 ? -1 : atoi(v) ? 1 : 0;
    if (const char *v = getenv("FHE_HMULT_FUSED_RESCALE")) c->hmult_fused_rescale = atoi(v) != 0;
    *out = c.release();
    return FHE_OK;
}

int fhe_ctx_destroy(fhe_ctx *ctx)
{
    if (!ctx) return FHE_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->cyclic.clear();
    ctx->garner.clear();
    ctx->fused_ctl.clear();
    ctx->packed.clear();
    ctx->pp_tmp.clear();
    for (auto &kv : ctx->side)
        if (kv.second && kv.second->s) {
            (void)hipStreamSynchronize(kv.second->s);
            (void)hipEventDestroy(kv.second->fork);
            (void)hipEventDestroy(kv.second->join);
            (void)hipStreamDestroy(kv.second->s);
        }
    ctx->side.clear();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FHE_OK;
}

int fhe_ctx_set_option(fhe_ctx *ctx, const char *name, long value)
{
    if (!ctx || !name) return fail(FHE_ERR_INVALID, "null argument");
    if (!std::strcmp(name, "ntt_mode")) ctx->mode = value ? 1 : 0;
    else if (!std::strcmp(name, "fused_dist")) ctx->fused_dist = (unsigned)std::max(1l, value);
    else if (!std::strcmp(name, "fused_wgs")) ctx->fused_wgs = (unsigned)std::max(1l, value);
    else if (!std::strcmp(name, "fused_variant")) ctx->fused_variant = (int)value;
    else if (!std::strcmp(name, "tile_geo")) ctx->geo = value ? 1 : 0;
    else if (!std::strcmp(name, "ntt_resident")) ctx->resident = value != 0;
    else if (!std::strcmp(name, "ntt_packed")) ctx->packed_on = value != 0;
    else if (!std::strcmp(name, "ntt_stream")) ctx->stream_hint = value < 0 ? -1 : value ? 1 : 0;
    else if (!std::strcmp(name, "ntt_chunk_floor_mib")) ctx->chunk_floor_mib = (unsigned)std::max(0l, value);
    else if (!std::strcmp(name, "ntt_split")) ctx->split = value < 0 ? -1 : value ? 1 : 0;
    else if (!std::strcmp(name, "ntt_pingpong")) ctx->pingpong = value < 0 ? -1 : value ? 1 : 0;
    else if (!std::strcmp(name, "ntt_chunk_mib")) ctx->chunk_mib = (unsigned)std::max(0l, value);
    else if (!std::strcmp(name, "ks_fused")) ctx->ks_fused = value < 0 ? -1 : value ? 1 : 0;
    else if (!std::strcmp(name, "hmult_fused_rescale")) ctx->hmult_fused_rescale = value != 0;
    else if (!std::strcmp(name, "ntt_only_pass")) ctx->only_pass = value == 0 ? 0 : value == 1 ? 1 : -1;   // bench.py times each kernel with it
    else if (!std::strcmp(name, "fused_skip_teams")) ctx->fused_skip_teams = (unsigned)value;   // test hook
    else return fail(FHE_ERR_INVALID, "unknown option");
    return FHE_OK;
}

int fhe_ctx_check(fhe_ctx *ctx)
{
    if (!ctx) return fail(FHE_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(ctx->mu);
    for (auto &kv : ctx->fused_ctl) {
        if (!kv.second || !kv.second->p) continue;
        HIP_TRY(hipStreamSynchronize(kv.first));
        u32 flag = 0;
        HIP_TRY(hipMemcpy(&flag, kv.second->as<u32>() + fused_error_word(), sizeof flag, hipMemcpyDeviceToHost));
        if (flag) {
            HIP_TRY(hipMemset(kv.second->as<u32>() + fused_error_word(), 0, sizeof flag));      // reported once
            return fail(FHE_ERR_HIP, "fused NTT: a bounded wait ran out in a launch since the last check (its results are invalid)");
        }
    }
    return FHE_OK;
}

int fhe_ctx_stream(fhe_ctx *ctx, void **stream_out)
{
    if (!ctx || !stream_out) return fail(FHE_ERR_INVALID, "null argument");
    *stream_out = ctx->stream;
    return FHE_OK;
}

int fhe_sync(fhe_ctx *ctx, void *stream)
{
    if (!ctx) return fail(FHE_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(pick(ctx, stream)));
    return FHE_OK;
}

int fhe_alloc(fhe_ctx *ctx, size_t bytes, void **dptr)
{
    if (!ctx || !dptr) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) return fail(FHE_ERR_NOMEM, "hipMalloc: out of device memory");
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return FHE_OK;
}

int fhe_free(fhe_ctx *ctx, void *dptr)
{
    if (!ctx) return fail(FHE_ERR_INVALID, "null ctx");
    if (!dptr) return FHE_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipFree(dptr));
    return FHE_OK;
}

int fhe_h2d(fhe_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream)
{
    if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, pick(ctx, stream)));
    return FHE_OK;
}
int fhe_d2h(fhe_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream)
{
    if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, pick(ctx, stream)));
    return FHE_OK;
}
int fhe_d2d(fhe_ctx *ctx, void *dst, const void *src, size_t bytes, void *stream)
{
    if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, pick(ctx, stream)));
    return FHE_OK;
}
int fhe_memset(fhe_ctx *ctx, void *dst, int byte, size_t bytes, void *stream)
{
    if (!ctx || (!dst && bytes)) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemsetAsync(dst, byte, bytes, pick(ctx, stream)));
    return FHE_OK;
}

// ---------------------------------------------------------------- host tables
int fhe_moduli_create(uint64_t N, const int *bits, int count, uint64_t *out_q)
{
    if (!bits || !out_q || count < 1 || N < 1 || (N & (N - 1))) return fail(FHE_ERR_INVALID, "bad arguments");
    for (int i = 0; i < count; i++)
        if (bits[i] < 2 || bits[i] > 61) return fail(FHE_ERR_UNSUPPORTED, "bit sizes must be in [2, 61]");
    if (!host::create_moduli(N, bits, count, out_q)) return fail(FHE_ERR_INVALID, "not enough primes of the requested size");
    return FHE_OK;
}

int fhe_modulus_const_ratio(uint64_t q, uint64_t out[3])
{
    if (q < 2 || !out) return fail(FHE_ERR_INVALID, "bad arguments");
    host::const_ratio(q, out);
    return FHE_OK;
}

int fhe_min_primitive_root(uint64_t q, uint64_t order, uint64_t *out)
{
    if (!out || !host::min_primitive_root(q, order, *out)) return fail(FHE_ERR_INVALID, "no primitive root of that order");
    return FHE_OK;
}

int fhe_root_powers(uint64_t q, int log_n, uint64_t *rp, uint64_t *rp_shoup)
{
    if (log_n < 1 || log_n > 30 || q < 2) return fail(FHE_ERR_INVALID, "bad arguments");
    u64 psi;
    if (!host::min_primitive_root(q, (u64)2 << log_n, psi)) return fail(FHE_ERR_INVALID, "q is not 1 mod 2N (or not prime)");
    const size_t N = (size_t)1 << log_n;
    std::vector<u64> t(N);
    host::root_powers(q, log_n, psi, t.data());
    for (size_t k = 0; k < N; k++) {
        if (rp) rp[k] = t[k];
        if (rp_shoup) rp_shoup[k] = (u64)(((unsigned __int128)t[k] << 64) / q);
    }
    return FHE_OK;
}

// ---------------------------------------------------------------- device tables
int fhe_ntt_tables_create(fhe_ctx *ctx, int log_n, const uint64_t *q, int count, fhe_ntt_tables **out)
{
    if (!ctx || !q || !out || count < 1 || log_n < 1 || log_n > NTT_MAX_LOGN) return fail(FHE_ERR_INVALID, "bad table arguments");
    const size_t N = (size_t)1 << log_n;
    std::vector<u64> rows((size_t)count * N), psi(count);
    for (int l = 0; l < count; l++) {
        if (q[l] < 2 || q[l] >= ((u64)1 << 61)) return fail(FHE_ERR_UNSUPPORTED, "modulus " + std::to_string(q[l]) + " outside [2, 2^61)");
        if (!host::min_primitive_root(q[l], (u64)2 * N, psi[l]))
            return fail(FHE_ERR_INVALID, "modulus " + std::to_string(q[l]) + " has no primitive 2N-th root");
        host::root_powers(q[l], log_n, psi[l], rows.data() + (size_t)l * N);
    }
    return build_tables(ctx, log_n, q, count, rows.data(), true, -1, psi.data(), out);
}

int fhe_ntt_tables_create_from_roots(fhe_ctx *ctx, int log_n, const uint64_t *q, int count, const uint64_t *root_powers,
                                     int force_path, fhe_ntt_tables **out)
{
    if (!root_powers) return fail(FHE_ERR_INVALID, "null root powers");
    if (force_path < -1 || force_path > 1) return fail(FHE_ERR_INVALID, "bad force_path");
    std::vector<u64> psi(count > 0 ? count : 0);
    if (log_n >= 1 && count >= 1)
        for (int l = 0; l < count; l++) psi[l] = root_powers[((size_t)l << log_n) + ((size_t)1 << (log_n - 1))];
    return build_tables(ctx, log_n, q, count, root_powers, true, force_path, psi.data(), out);
}

int fhe_ntt_tables_destroy(fhe_ntt_tables *t)
{
    if (t) {
        (void)hipSetDevice(t->ctx->device);
        delete t;
    }
    return FHE_OK;
}

int fhe_ntt_tables_info(const fhe_ntt_tables *t, int *log_n, int *count, int *out_path, uint64_t *out_psi)
{
    if (!t) return fail(FHE_ERR_INVALID, "null tables");
    if (log_n) *log_n = t->log_n;
    if (count) *count = t->count;
    for (int l = 0; l < t->count; l++) {
        if (out_path) out_path[l] = t->path[l];
        if (out_psi) out_psi[l] = t->psi[l];
    }
    return FHE_OK;
}

// ---------------------------------------------------------------- transforms
int fhe_ntt_forward_inplace(fhe_ctx *ctx, uint64_t *d, const fhe_ntt_tables *t, size_t limbs, size_t start_idx, void *stream)
{
    return ntt_batch(ctx, d, t, 1, limbs, start_idx, stream, false);
}
int fhe_ntt_inverse_inplace(fhe_ctx *ctx, uint64_t *d, const fhe_ntt_tables *t, size_t limbs, size_t start_idx, void *stream)
{
    return ntt_batch(ctx, d, t, 1, limbs, start_idx, stream, true);
}
int fhe_ntt_forward_batch(fhe_ctx *ctx, uint64_t *d, const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx,
                          void *stream)
{
    return ntt_batch(ctx, d, t, n_poly, limbs, start_idx, stream, false);
}
int fhe_ntt_inverse_batch(fhe_ctx *ctx, uint64_t *d, const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx,
                          void *stream)
{
    return ntt_batch(ctx, d, t, n_poly, limbs, start_idx, stream, true);
}

int fhe_bitrev_permute(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, int log_n, size_t n_vec, void *stream)
{
    if (!ctx || !d_dst || !d_src || d_dst == d_src || log_n < 0 || log_n > 30) return fail(FHE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_bitrev_scale(pick(ctx, stream), d_dst, d_src, log_n, (u32)n_vec, ModConst{1, 0, 0}, 1, false);
    if (e != hipSuccess) return hip_fail(e, "launch_bitrev_scale");
    return FHE_OK;
}

// Natural-order transform of n_vec vectors (two launches through a hand-off buffer, launch_ntt_gs).  Batches that cannot stay in the
// Infinity Cache run as sub-batches on alternating streams like the negacyclic transforms (ntt_batch), each stream handing over through
// its own scratch of one sub-batch; d_tmp_full (whole-batch hand-off buffer, may be null when the batch is cut) serves the others.
static int gs_batch(fhe_ctx *ctx, hipStream_t st, u64 *d_dst, const u64 *d_src, const fhe_ntt_tables *t, int log_n, size_t n_vec, u64 *d_tmp_full)
{
    const size_t N = (size_t)1 << log_n;
    const LimbParams *lp = t->d_lp.as<LimbParams>();
    const int path = t->path[0];
    if (const size_t per = sub_batch_polys(ctx, log_n, n_vec, 1)) {
        u64 *pp = nullptr;
        HIP_TRY(handoff_scratch(ctx, st, per * N * 8, &pp));
        if (pp)
            return for_sub_batches(ctx, st, n_vec, per, per * N * 8, [&](hipStream_t s, size_t p0, size_t cnt, u64 *side_tmp) {
                PassArgs a{d_dst + p0 * N, lp, 0u, 1u, (u32)cnt, 1u};
                a.src = d_src + p0 * N;
                a.stream_hint = ctx->stream_hint != 0;
                return launch_ntt_gs(s, a, side_tmp ? side_tmp : pp, log_n, path);
            });
    }
    if (!d_tmp_full) return fail(FHE_ERR_INVALID, "no hand-off buffer");
    PassArgs a{d_dst, lp, 0u, 1u, (u32)n_vec, 1u};
    a.src = d_src;
    hipError_t e = launch_ntt_gs(st, a, d_tmp_full, log_n, path);
    if (e != hipSuccess) return hip_fail(e, "launch_ntt_gs");
    return FHE_OK;
}

int fhe_ntt_cyclic(fhe_ctx *ctx, uint64_t *d_data, uint64_t *d_scratch, int log_n, size_t n_vec, uint64_t mod, uint64_t root,
                   int convention, int inverse, void *stream)
{
    if (!ctx || !d_data || !d_scratch) return fail(FHE_ERR_INVALID, "null argument");
    if (log_n < 1 || log_n > NTT_MAX_LOGN || mod < 2 || (convention != 0 && convention != 1))
        return fail(FHE_ERR_INVALID, "bad cyclic NTT arguments");
    if (!n_vec) return FHE_OK;
    const u64 n = (u64)1 << log_n;
    u64 use_root = root % mod, scale = 1;
    if (inverse) {
        // motivation/bsgs.py:31-36: inv_root = root^(mod-2), inv_n = n^(mod-2)
        use_root = host::pow_mod(root, mod - 2, mod);
        scale = host::pow_mod(n % mod, mod - 2, mod);
    }
    if (!ntt_gs_supported(log_n)) return cyclic_small(ctx, d_data, d_scratch, log_n, n_vec, mod, use_root, convention, scale, inverse != 0, stream);
    fhe_ntt_tables *t = nullptr;
    int rc = cyclic_tables(ctx, log_n, mod, use_root, convention, scale, &t);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    // two launches, the bit reversal folded into the first one's loads, the scale into the last stage: d_data -> d_scratch -> d_data
    return gs_batch(ctx, st, d_data, d_data, t, log_n, n_vec, d_scratch);
}

// ---------------------------------------------------------------- four-step
// four_step_ntt (reliability_test/four_step_ntt_prot.py:71-109): column transforms, twiddle, row transforms, natural-order
// output, equal to the direct DFT with w = g^((mod-1)/N) (:244-245).  The engine's natural-order transform IS that flow --
// launch 1 = the batch of "column" transforms read straight from the input's columns (no transpose pass), launch 2 = the batch
// of "row" transforms with the twiddle w^(k2 t1) folded into their butterflies, written in natural order -- at the engine's own
// split of N (ntt_plan.hpp); the result does not depend on the split, so any power-of-two (n1, n2) is accepted, n1 != n2 included.
int fhe_fourstep_create(fhe_ctx *ctx, uint64_t n1, uint64_t n2, uint64_t mod, uint64_t g, fhe_fourstep **out)
{
    if (!ctx || !out) return fail(FHE_ERR_INVALID, "null argument");
    const int l1 = ilog2_exact(n1), l2 = ilog2_exact(n2);
    if (l1 < 1 || l2 < 1 || l1 > NTT_MAX_LOGN || l2 > NTT_MAX_LOGN || l1 + l2 > 26 || mod < 2)
        return fail(FHE_ERR_INVALID, "n1 and n2 must be powers of two, 2 <= n1, n2 <= 2^20, n1 * n2 <= 2^26");
    const u64 N = n1 * n2;
    if ((mod - 1) % N) return fail(FHE_ERR_INVALID, "N must divide mod - 1");
    if (mod >= ((u64)1 << 61)) return fail(FHE_ERR_UNSUPPORTED, "modulus must be below 2^61");
    std::unique_ptr<fhe_fourstep> p(new fhe_fourstep);
    p->ctx = ctx;
    p->n1 = n1;
    p->n2 = n2;
    p->mod = mod;
    p->g = g % mod;
    p->log_n = l1 + l2;
    p->log1 = l1;
    p->log2 = l2;
    if (p->log_n > NTT_MAX_LOGN) {
        // past the largest single plan: the composition of four_step_ntt_prot.py:71-109 itself, each factor through the natural-order
        // transform of its length; w^(k2 t1) from two tables of 2^lo_bits and 2^(log_n - lo_bits) entries
        p->big = true;
        p->lo_bits = (p->log_n + 1) / 2;
        const u64 w = host::pow_mod(p->g, (mod - 1) / N, mod);
        std::vector<u64> lo((size_t)1 << p->lo_bits), hi((size_t)1 << (p->log_n - p->lo_bits));
        u64 cur = 1 % mod;
        for (auto &x : lo) {
            x = cur;
            cur = host::mul_mod(cur, w, mod);
        }
        const u64 step = cur;           // w^(2^lo_bits)
        cur = 1 % mod;
        for (auto &x : hi) {
            x = cur;
            cur = host::mul_mod(cur, step, mod);
        }
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(p->tw_lo.upload(lo));
        HIP_TRY(p->tw_hi.upload(hi));
        *out = p.release();
        return FHE_OK;
    }
    if (ntt_gs_supported(p->log_n)) {
        int rc = cyclic_tables(ctx, p->log_n, mod, p->g, 0, 1 % mod, &p->t);
        if (rc) return rc;
    }
    *out = p.release();
    return FHE_OK;
}

int fhe_fourstep_destroy(fhe_fourstep *p)
{
    if (p) {
        (void)hipSetDevice(p->ctx->device);
        delete p;
    }
    return FHE_OK;
}

int fhe_fourstep_ntt_batch(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, fhe_fourstep *p, size_t n_vec, void *stream)
{
    if (!ctx || !d_dst || !d_src || !p) return fail(FHE_ERR_INVALID, "null argument");
    if (!n_vec) return FHE_OK;
    if (n_vec > ((size_t)1 << 24)) return fail(FHE_ERR_INVALID, "batch too large for one launch (max 2^24 vectors)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    const size_t words = n_vec << p->log_n;
    if (p->big) {
        if (n_vec > 4096 || (n_vec << (p->log1 > p->log2 ? p->log1 : p->log2)) > ((size_t)1 << 24)) return fail(FHE_ERR_INVALID, "batch too large for the large-N four-step");
        if (p->buf0.bytes < words * 8) {
            HIP_TRY(hipStreamSynchronize(st));      // growing frees the old blocks
            HIP_TRY(p->buf0.alloc(words * 8));
            HIP_TRY(p->buf1.alloc(words * 8));
        }
        u64 *b0 = p->buf0.as<u64>(), *b1 = p->buf1.as<u64>();
        const u32 n1 = (u32)p->n1, n2 = (u32)p->n2;
        const ModConst mc = mod_const(p->mod);
        hipError_t e;
        int rc;
        // A[t2][t1] = a[t1 + n1 t2] (:81)  ->  b0[t1][t2]
        if ((e = launch_transpose_tw(st, b0, d_src, n2, n1, (u32)n_vec, mc, nullptr, nullptr, 0)) != hipSuccess) return hip_fail(e, "launch_transpose_tw");
        // n1 transforms of length n2 along t2, natural order in and out (:84-90)
        if ((rc = fhe_ntt_cyclic(ctx, b0, b1, p->log2, n_vec * n1, p->mod, p->g, 0, 0, st))) return rc;
        // C[t1][k2] = B[t1][k2] w^(k2 t1) (:93) on the way to b1[k2][t1]
        if ((e = launch_transpose_tw(st, b1, b0, n1, n2, (u32)n_vec, mc, p->tw_lo.as<u64>(), p->tw_hi.as<u64>(), p->lo_bits)) != hipSuccess) return hip_fail(e, "launch_transpose_tw");
        // n2 transforms of length n1 along t1 (:96-102)
        if ((rc = fhe_ntt_cyclic(ctx, b1, b0, p->log1, n_vec * n2, p->mod, p->g, 0, 0, st))) return rc;
        // y[k1 n2 + k2] = Y[k2][k1] (:105-108)
        if ((e = launch_transpose_tw(st, d_dst, b1, n2, n1, (u32)n_vec, mc, nullptr, nullptr, 0)) != hipSuccess) return hip_fail(e, "launch_transpose_tw");
        return FHE_OK;
    }
    const bool cut = p->t && sub_batch_polys(ctx, p->log_n, n_vec, 1) != 0;      // (then the hand-off goes through per-stream scratch)
    if (!cut && p->tmp.bytes < words * 8) {
        HIP_TRY(hipStreamSynchronize(st));      // growing frees the old block
        HIP_TRY(p->tmp.alloc(words * 8));
    }
    if (!p->t) {
        // N < 32: the small cyclic path works in place with a scratch buffer
        if (d_dst != d_src) HIP_TRY(hipMemcpyAsync(d_dst, d_src, words * 8, hipMemcpyDeviceToDevice, st));
        return fhe_ntt_cyclic(ctx, d_dst, p->tmp.as<u64>(), p->log_n, n_vec, p->mod, p->g, 0, 0, st);
    }
    if (cut && p->tmp.bytes < words * 8) {
        // (inside a stream capture the scratch may be unavailable: the whole-batch buffer is the fallback and has to exist)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        if (cap != hipStreamCaptureStatusNone) return fail(FHE_ERR_INVALID, "first call of this batch size inside a stream capture: run it once outside");
    }
    return gs_batch(ctx, st, d_dst, d_src, p->t, p->log_n, n_vec, p->tmp.bytes >= words * 8 ? p->tmp.as<u64>() : nullptr);
}

int fhe_fourstep_ntt(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, fhe_fourstep *p, void *stream)
{
    return fhe_fourstep_ntt_batch(ctx, d_dst, d_src, p, 1, stream);
}

// ---------------------------------------------------------------- pointwise
int fhe_modmul(fhe_ctx *ctx, uint64_t *c, const uint64_t *a, const uint64_t *b, const fhe_ntt_tables *t, size_t n_poly,
               size_t limbs, size_t start_idx, void *stream)
{
    return pointwise(ctx, c, a, b, t, n_poly, limbs, start_idx, stream, false);
}
int fhe_modmul_acc(fhe_ctx *ctx, uint64_t *c, const uint64_t *a, const uint64_t *b, const fhe_ntt_tables *t, size_t n_poly,
                   size_t limbs, size_t start_idx, void *stream)
{
    return pointwise(ctx, c, a, b, t, n_poly, limbs, start_idx, stream, true);
}

int fhe_modadd(fhe_ctx *ctx, uint64_t *c, const uint64_t *a, const uint64_t *b, const fhe_ntt_tables *t, size_t n_poly,
               size_t limbs, size_t start_idx, void *stream)
{
    if (!ctx || !c || !a || !b) return fail(FHE_ERR_INVALID, "null argument");
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    PointwiseArgs p{c, a, b, t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs, (u32)(n_poly * limbs), (u32)limbs, t->log_n};
    hipError_t e = launch_modadd(pick(ctx, stream), p);
    if (e != hipSuccess) return hip_fail(e, "launch_modadd");
    return FHE_OK;
}

int fhe_modsub(fhe_ctx *ctx, uint64_t *c, const uint64_t *a, const uint64_t *b, const fhe_ntt_tables *t, size_t n_poly,
               size_t limbs, size_t start_idx, void *stream)
{
    if (!ctx || !c || !a || !b) return fail(FHE_ERR_INVALID, "null argument");
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    PointwiseArgs p{c, a, b, t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs, (u32)(n_poly * limbs), (u32)limbs, t->log_n};
    hipError_t e = launch_modsub(pick(ctx, stream), p);
    if (e != hipSuccess) return hip_fail(e, "launch_modsub");
    return FHE_OK;
}

int fhe_scalar_affine(fhe_ctx *ctx, uint64_t *c, const uint64_t *a, const uint64_t *mul, const uint64_t *add, const fhe_ntt_tables *t,
                      size_t n_poly, size_t limbs, size_t start_idx, void *stream)
{
    if (!ctx || !c || !a) return fail(FHE_ERR_INVALID, "null argument");
    if (limbs > SCALAR_MAX_LIMBS) return fail(FHE_ERR_UNSUPPORTED, "at most 64 limbs per scalar call");
    int rc = check_range(t, n_poly, limbs, start_idx);
    if (rc) return rc;
    ScalarVec m{}, ad{};
    for (size_t l = 0; l < limbs; l++) {
        const u64 q = t->q[start_idx + l];
        m.v[l] = mul ? mul[l] % q : 1 % q;
        ad.v[l] = add ? add[l] % q : 0;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    PointwiseArgs p{c, a, a, t->d_lp.as<LimbParams>(), (u32)start_idx, (u32)limbs, (u32)(n_poly * limbs), (u32)limbs, t->log_n};
    hipError_t e = launch_scalar_affine(pick(ctx, stream), p, m, ad);
    if (e != hipSuccess) return hip_fail(e, "launch_scalar_affine");
    return FHE_OK;
}

int fhe_polymul(fhe_ctx *ctx, uint64_t *c, uint64_t *a, uint64_t *b, const fhe_ntt_tables *t, size_t n_poly, size_t limbs,
                size_t start_idx, void *stream)
{
    int rc;
    if (!ctx || !c || !a || !b) return fail(FHE_ERR_INVALID, "null argument");
    if ((rc = check_range(t, n_poly, limbs, start_idx))) return rc;
    if (!t->has_inverse) return fail(FHE_ERR_UNSUPPORTED, "table set has no inverse (twiddle or N not invertible)");
    if (!n_poly || !limbs) return FHE_OK;
    if (ctx->mode == 0 && ctx->fault_idx < 0 && polymul_fused_supported(t->log_n)) {
        HIP_TRY(hipSetDevice(ctx->device));
        hipStream_t st = pick(ctx, stream);
        const size_t N = (size_t)1 << t->log_n;
        TraceScope tr(ctx, st, "POLYMUL");
        return for_each_run(t, limbs, start_idx, [&](size_t off, size_t len, int path) -> int {
            PassArgs pa{a + off * N, t->d_lp.as<LimbParams>(), (u32)(start_idx + off), (u32)len, (u32)(n_poly * len), (u32)limbs};
            // operands + result that cannot stay in the Infinity Cache together: pieces that can (limb windows x polynomial ranges,
            // alternating between the caller's stream and the side stream), so that the middle launch and the inverse column pass
            // find their inputs on-die
            const SubBatchCut cut = ctx->only_pass < 0 ? sub_batch_cut(ctx, t->log_n, n_poly, len, a == b ? 2 : 3) : SubBatchCut{0, 0};
            if (cut.pc) {
                const size_t npp = (n_poly + cut.pc - 1) / cut.pc, npl = (len + cut.lc - 1) / cut.lc;
                return for_pieces(ctx, st, npp * npl, 0, [&](hipStream_t s, size_t i, u64 *) {
                    const size_t l0 = (i / npp) * cut.lc, p0 = (i % npp) * cut.pc, o = (p0 * limbs + l0) * N;
                    const size_t lc = std::min(cut.lc, len - l0), pc = std::min(cut.pc, n_poly - p0);
                    PassArgs w = pa;
                    w.data = pa.data + o;
                    w.limb0 = pa.limb0 + (u32)l0;
                    w.limbs = (u32)lc;
                    w.units = (u32)(pc * lc);
                    // (no non-temporal accesses here: measured slower on the product's pieces, 0.48 against 0.51, profiles/r02_polymul_sweep.txt)
                    return launch_polymul(s, w, b + off * N + o, c + off * N + o, t->log_n, path);
                });
            }
            hipError_t e = launch_polymul(st, pa, b + off * N, c + off * N, t->log_n, path);
            if (e != hipSuccess) return hip_fail(e, "launch_polymul");
            return FHE_OK;
        });
    }
    // unfused sequence (tiny sizes, fused-NTT mode, fault-injection hook)
    if ((rc = ntt_batch(ctx, a, t, n_poly, limbs, start_idx, stream, false))) return rc;
    if (b != a && (rc = ntt_batch(ctx, b, t, n_poly, limbs, start_idx, stream, false))) return rc;
    if ((rc = pointwise(ctx, c, a, b, t, n_poly, limbs, start_idx, stream, false))) return rc;
    return ntt_batch(ctx, c, t, n_poly, limbs, start_idx, stream, true);
}

int fhe_ctx_trace(fhe_ctx *ctx, int enable)
{
    if (!ctx) return fail(FHE_ERR_INVALID, "null ctx");
    ctx->trace_on = enable != 0;
    if (enable) ctx->trace.clear();
    return FHE_OK;
}

int fhe_ctx_trace_read(fhe_ctx *ctx, char *buf, size_t cap, size_t *len)
{
    if (!ctx || (!buf && cap)) return fail(FHE_ERR_INVALID, "null argument");
    if (len) *len = ctx->trace.size();
    if (cap) {
        const size_t n = std::min(cap - 1, ctx->trace.size());
        std::memcpy(buf, ctx->trace.data(), n);
        buf[n] = 0;
    }
    return FHE_OK;
}

} // extern "C"
