// capi_baseconv.cpp -- RNS base conversion, Garner CRT, BSGS Hadamard, bit flip (part of the C ABI of include/fhe_mi355x.h; shared pieces in capi_internal.hpp)
#include "capi_internal.hpp"

extern "C" {

// ---------------------------------------------------------------- base conversion
int fhe_baseconv_create(fhe_ctx *ctx, const uint64_t *mod_in, int m, const uint64_t *mod_out, int k, fhe_baseconv **out)
{
    if (!ctx || !mod_in || !mod_out || !out || m < 1 || k < 1 || m > 64 || k > 64) return fail(FHE_ERR_INVALID, "bad base conversion arguments");
    std::unique_ptr<fhe_baseconv> p(new fhe_baseconv);
    p->m = m;
    p->k = k;
    std::vector<u64> mi(mod_in, mod_in + m), mo(mod_out, mod_out + k), fc((size_t)m * k), fs((size_t)m * k);
    u64 maxq = 0, maxall = 0;
    for (int j = 0; j < m; j++) {
        if (mi[j] < 2 || mi[j] >= ((u64)1 << 62)) return fail(FHE_ERR_UNSUPPORTED, "input modulus out of range");
        maxall = std::max(maxall, mi[j]);
    }
    for (int o = 0; o < k; o++) {
        if (mo[o] < 2 || mo[o] >= ((u64)1 << 62)) return fail(FHE_ERR_UNSUPPORTED, "output modulus out of range");
        maxq = std::max(maxq, mo[o]);
    }
    maxall = std::max(maxall, maxq);
    // constants of the exact conversion (aux_kernels.hip k_baseconv_exact), encoded for the arithmetic path
    const bool f64 = maxall < ((u64)1 << 50);
    auto enc = [&](u64 w, u64 q) { return f64 ? ArithF64::encode(w, q) : ArithU64::encode(w, q); };
    auto fp = [](u64 q) { return Tw{double_to_u64_bits((double)q), double_to_u64_bits(1.0 / (double)q)}; };
    std::vector<Tw> dig((size_t)m * m, Tw{0, 0}), hor((size_t)m * k), fpi(m), fpo(k);
    for (int j = 0; j < m; j++) {
        fpi[j] = fp(mi[j]);
        // D_lj = (p_l ... p_{j-1})^-1 mod p_j for l = j-1 .. 0; A_j = D_0j (1 for j = 0)
        u64 prod = 1 % mi[j];
        for (int l = j - 1; l >= 0; l--) {
            prod = host::mul_mod(prod, mi[l] % mi[j], mi[j]);
            const u64 inv = host::inv_mod(prod, mi[j]);
            if (!inv) return fail(FHE_ERR_INVALID, "input moduli must be pairwise coprime");
            dig[(size_t)l * m + j] = enc(inv, mi[j]);
            if (l == 0) dig[(size_t)j * m + j] = enc(inv, mi[j]);
        }
        if (j == 0) dig[0] = enc(1 % mi[0], mi[0]);
    }
    for (int o = 0; o < k; o++) {
        fpo[o] = fp(mo[o]);
        u64 prod = 1 % mo[o];
        for (int l = 0; l < m; l++) {
            hor[(size_t)l * k + o] = enc(prod, mo[o]);
            prod = host::mul_mod(prod, mi[l] % mo[o], mo[o]);
        }
    }
    // rfhe_framewk/src/baseConv.py:17-18: hat_p[j] = P // p_j, inv_hat_p[j] = hat_p[j]^-1 mod p_j
    for (int j = 0; j < m; j++) {
        u64 hat_pj = 1 % mi[j];
        for (int l = 0; l < m; l++)
            if (l != j) hat_pj = host::mul_mod(hat_pj, mi[l] % mi[j], mi[j]);
        const u64 inv = host::inv_mod(hat_pj, mi[j]);
        for (int o = 0; o < k; o++) {
            u64 hat_q = 1 % mo[o];
            for (int l = 0; l < m; l++)
                if (l != j) hat_q = host::mul_mod(hat_q, mi[l] % mo[o], mo[o]);
            const u64 coef = host::mul_mod(hat_q, inv % mo[o], mo[o]);
            fc[(size_t)j * k + o] = coef;
            fs[(size_t)j * k + o] = (u64)(((unsigned __int128)coef << 64) / mo[o]);
        }
    }
    p->fast_ok = (unsigned __int128)maxq * (u64)m < ((unsigned __int128)1 << 64);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(p->mod_in.upload(mi));
    HIP_TRY(p->mod_out.upload(mo));
    HIP_TRY(p->dig.upload(dig));
    HIP_TRY(p->hor.upload(hor));
    HIP_TRY(p->fp_in.upload(fpi));
    HIP_TRY(p->fp_out.upload(fpo));
    HIP_TRY(p->fast_coef.upload(fc));
    HIP_TRY(p->fast_shoup.upload(fs));
    if (m <= 16) {
        std::vector<Tw> head((size_t)m * m + m + (m + 1) / 2, Tw{0, 0}), outs((size_t)k * (m + 2), Tw{0, 0});
        for (int i = 0; i < m * m; i++) head[i] = dig[i];
        for (int j = 0; j < m; j++) {
            head[(size_t)m * m + j] = fpi[j];
            Tw &e = head[(size_t)m * m + m + j / 2];
            (j & 1 ? e.b : e.a) = mi[j];
        }
        for (int o = 0; o < k; o++) {
            // FP64 plans: the output constants as two halves of 25 bits, E = e1 2^25 + e0 with e0 centred (the fixed-size kernels'
            // split products); integer plans: the Shoup pairs
            u64 prod = 1 % mo[o];
            for (int l = 0; l < m; l++) {
                if (f64) {
                    const double E = (double)prod, e1 = __builtin_rint(E * 0x1p-25), e0 = E - e1 * 0x1p25;
                    outs[(size_t)o * (m + 2) + l] = Tw{double_to_u64_bits(e1), double_to_u64_bits(e0)};
                } else {
                    outs[(size_t)o * (m + 2) + l] = hor[(size_t)l * k + o];
                }
                prod = host::mul_mod(prod, mi[l] % mo[o], mo[o]);
            }
            outs[(size_t)o * (m + 2) + m] = fpo[o];
            outs[(size_t)o * (m + 2) + m + 1] = Tw{mo[o], 0};
        }
        HIP_TRY(p->img_head.upload(head));
        HIP_TRY(p->img_out.upload(outs));
        std::vector<u32> ident(64);
        for (u32 i = 0; i < 64; i++) ident[i] = i;
        HIP_TRY(p->rows_id.upload(ident));
    }
    p->dev = BaseConvPlanDev{m, k, p->mod_in.as<u64>(), p->mod_out.as<u64>(), p->dig.as<Tw>(), p->hor.as<Tw>(), p->fp_in.as<Tw>(),
                             p->fp_out.as<Tw>(), f64 ? 1 : 0, p->fast_coef.as<u64>(), p->fast_shoup.as<u64>(), p->img_head.as<Tw>(), p->img_out.as<Tw>(), p->rows_id.as<u32>()};
    *out = p.release();
    return FHE_OK;
}

int fhe_baseconv_destroy(fhe_baseconv *p)
{
    delete p;
    return FHE_OK;
}

int fhe_baseconv_exact(fhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, const fhe_baseconv *p, size_t N, void *stream)
{
    if (!ctx || !d_out || !d_in || !p) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_baseconv_exact(pick(ctx, stream), d_out, d_in, p->dev, N);
    if (e != hipSuccess) return hip_fail(e, "launch_baseconv_exact");
    return FHE_OK;
}

int fhe_baseconv_fast(fhe_ctx *ctx, uint64_t *d_out, const uint64_t *d_in, const fhe_baseconv *p, size_t N, void *stream)
{
    if (!ctx || !d_out || !d_in || !p) return fail(FHE_ERR_INVALID, "null argument");
    if (!p->fast_ok) return fail(FHE_ERR_UNSUPPORTED, "unreduced sum would exceed 64 bits (m * max q >= 2^64)");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_bconv_fast(pick(ctx, stream), d_out, d_in, p->dev, N);
    if (e != hipSuccess) return hip_fail(e, "launch_bconv_fast");
    return FHE_OK;
}

int fhe_crt_garner(fhe_ctx *ctx, uint64_t *d_x_lo, uint64_t *d_x_hi, const uint64_t *d_residues, const uint64_t *moduli, int m,
                   size_t N, void *stream)
{
    if (!ctx || !d_x_lo || !d_x_hi || !d_residues || !moduli) return fail(FHE_ERR_INVALID, "null argument");
    if (m < 1 || m > 16) return fail(FHE_ERR_UNSUPPORTED, "crt_garner supports 1..16 limbs");
    GarnerTables *g = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        std::vector<u64> key(moduli, moduli + m);
        auto it = ctx->garner.find(key);
        if (it == ctx->garner.end()) {
            // rfhe_framewk/src/baseConv.cu:157-169
            std::vector<u64> ratio(2 * m), plo(m), phi(m), inv(m, 0);
            unsigned __int128 pref = 1;
            for (int j = 0; j < m; j++) {
                if (moduli[j] < 2 || moduli[j] >= ((u64)1 << 62)) return fail(FHE_ERR_UNSUPPORTED, "modulus out of range");
                u64 cr[3];
                host::const_ratio(moduli[j], cr);
                ratio[2 * j] = cr[0];
                ratio[2 * j + 1] = cr[1];
                plo[j] = (u64)pref;
                phi[j] = (u64)(pref >> 64);
                if (j >= 1) {
                    inv[j] = host::inv_mod((u64)(pref % moduli[j]), moduli[j]);
                    if (!inv[j]) return fail(FHE_ERR_INVALID, "prefix product not invertible modulo p_j");
                }
                pref *= moduli[j];
            }
            std::unique_ptr<GarnerTables> nt(new GarnerTables);
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(nt->mod.upload(key));
            HIP_TRY(nt->ratio.upload(ratio));
            HIP_TRY(nt->pref_lo.upload(plo));
            HIP_TRY(nt->pref_hi.upload(phi));
            HIP_TRY(nt->inv_pref.upload(inv));
            it = ctx->garner.emplace(key, std::move(nt)).first;
        }
        g = it->second.get();
    }
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_crt_garner(pick(ctx, stream), d_x_lo, d_x_hi, d_residues, g->mod.as<u64>(), g->ratio.as<u64>(),
                                     g->pref_lo.as<u64>(), g->pref_hi.as<u64>(), g->inv_pref.as<u64>(), m, N);
    if (e != hipSuccess) return hip_fail(e, "launch_crt_garner");
    return FHE_OK;
}

int fhe_bsgs_hadamard(fhe_ctx *ctx, uint64_t *d_y, const uint64_t *d_M_blocks, const uint64_t *d_v, int k, int block_size,
                      uint64_t mod, void *stream)
{
    if (!ctx || !d_y || !d_M_blocks || !d_v || k < 1 || block_size < 1) return fail(FHE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    ModConst mc = mod >= 2 ? mod_const(mod) : ModConst{1, 0, 0};
    hipError_t e = launch_bsgs_hadamard(pick(ctx, stream), d_y, d_M_blocks, d_v, k, block_size, mod >= 2 ? &mc : nullptr);
    if (e != hipSuccess) return hip_fail(e, "launch_bsgs_hadamard");
    return FHE_OK;
}

int fhe_flip_bit(fhe_ctx *ctx, uint64_t *d_data, uint64_t idx, int bit, void *stream)
{
    if (!ctx || !d_data || bit < 0 || bit > 63) return fail(FHE_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = launch_flip_bit(pick(ctx, stream), d_data, idx, bit);
    if (e != hipSuccess) return hip_fail(e, "launch_flip_bit");
    return FHE_OK;
}

} // extern "C"
