// capi.cpp -- the C ABI declared in include/fhe_mi355x.h: handle management, table
// construction and launch sequencing.  No arithmetic on residues happens on the host
// here; there is no CPU fallback -- every transform call ends in a HIP kernel launch
// or an error.
#include "../../include/fhe_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "host_math.hpp"
#include "ntt_fused.hpp"
#include "ntt_launch.hpp"

using namespace fhe;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return FHE_ERR_HIP;
}
#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);      \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n)
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = n;
        return hipMalloc(&p, n ? n : 16);
    }
    template <class T> hipError_t upload(const std::vector<T> &v)
    {
        hipError_t e = alloc(v.size() * sizeof(T));
        if (e != hipSuccess) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

} // namespace

struct fhe_ntt_tables {
    fhe_ctx *ctx = nullptr;
    int log_n = 0, count = 0;
    std::vector<u64> q, psi;
    std::vector<int> path;
    std::vector<LimbParams> h_lp;
    DevBuf d_lp, d_tw;
    bool has_inverse = true;
};

struct fhe_baseconv {
    int m = 0, k = 0;
    bool fast_ok = true;
    DevBuf mod_in, mod_out, dig, hor, fp_in, fp_out, fast_coef, fast_shoup;
    BaseConvPlanDev dev{};
};

struct GarnerTables {
    DevBuf mod, ratio, pref_lo, pref_hi, inv_pref;
};

struct fhe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // fused-NTT control blocks, one per stream the caller launches on (zeroed on that stream per launch)
    std::map<hipStream_t, std::unique_ptr<DevBuf>> fused_ctl;
    int mode = 0;          // 0 = two launches per transform (default), 1 = fused launch (experimental)
    unsigned fused_dist = 4, fused_wgs = 768;
    unsigned fused_skip_teams = 0;
    bool trace_on = false;
    std::string trace;          // collected trace text (fhe_ctx_trace)
    long long fault_idx = -1;   // one-shot mid-transform bit flip (fhe_ctx_inject_fault)
    int fault_bit = 0;
    int geo = 1;           // column-tile geometry of the two-launch path (ntt_launch.hpp)
    bool resident = false; // 2^13 / 2^14: one LDS-resident pass instead of two launches (opt-in, see ntt_plan.hpp)
    int only_pass = -1;    // measurement hook: 0 / 1 = launch only the first / second pass of a two-pass size
    int fused_variant = 7;   // handoff*2 + stream hint (ntt_launch.hpp); 7 = acquire + nt streaming
    // cyclic tables keyed by (log_n, mod, root, convention)
    std::map<std::tuple<int, u64, u64, int>, std::unique_ptr<fhe_ntt_tables>> cyclic;
    std::map<std::vector<u64>, std::unique_ptr<GarnerTables>> garner;
};

struct fhe_abft {
    fhe_ctx *ctx = nullptr;
    const fhe_ntt_tables *t = nullptr;
    DevBuf w, what, ninv;       // count x N weights (input side / output side), N^-1 per limb
    DevBuf win, wout, wout8;    // weights for the fused checksums: twiddle-encoded (ArithU64 limbs) and, for the output side,
                                // as residues (ArithF64 limbs); N^-1 is folded into the output-side weights
    DevBuf sum_in, sum_out;     // scratch checksums (grown on demand)
};

struct fhe_keyswitch {
    fhe_ctx *ctx = nullptr;
    const fhe_ntt_tables *t = nullptr;
    int L = 0, K = 0, dnum = 0, alpha = 0, log_n = 0;
    u64 plain_modulus = 0;              // BGV: delta must vanish modulo this (0 = CKKS-style flooring)
    std::vector<u64> t_inv_P, t_mod_Q;  // plain_modulus^-1 mod p_k, plain_modulus mod q_j
    std::vector<fhe_baseconv *> up;     // per digit: digit primes -> every other prime (ascending index)
    fhe_baseconv *down = nullptr;       // P -> Q
    DevBuf pinv;                        // P^-1 mod q_j, j < L
    DevBuf coef, ext, acc, conv, rot;  // coef [L][N], ext [dnum][M][N], acc [2][M][N], conv [2][L][N]
    DevBuf up_jobs, down_jobs;         // device job lists: all digit extensions / both mod-down conversions in one launch each
    int up_max_m = 0, up_max_k = 0;
    bool up_batched = false;           // every digit plan on the same arithmetic path
    DevBuf ext_map[2];                 // per arithmetic path: the limbs of ext the forward transform covers
    u32 ext_units[2] = {0, 0};
    ~fhe_keyswitch()
    {
        for (auto *b : up) fhe_baseconv_destroy(b);
        fhe_baseconv_destroy(down);
    }
};

struct fhe_fourstep {
    fhe_ctx *ctx = nullptr;
    u64 n1 = 0, n2 = 0, mod = 0;
    int log1 = 0, log2 = 0;
    fhe_ntt_tables *t1 = nullptr, *t2 = nullptr; // sub-transform tables of length n1 / n2
    DevBuf tw, buf0, buf1;
    ModConst mc{};
};

namespace {

hipStream_t pick(fhe_ctx *ctx, void *stream) { return stream ? static_cast<hipStream_t>(stream) : ctx->stream; }

int path_for(u64 q)
{
    if (q < 2) return -1;
    if (q < ((u64)1 << 50)) return PATH_F64;
    if (q < ((u64)1 << 61)) return PATH_U64;
    return -1;
}

Tw encode(int path, u64 w, u64 q) { return path == PATH_F64 ? ArithF64::encode(w, q) : ArithU64::encode(w, q); }

// entry-wise inverses of t[1..n-1] mod q by Montgomery's batch trick; false when an
// entry is not a unit
bool batch_inverse(const u64 *t, size_t n, u64 q, std::vector<u64> &out)
{
    out.assign(n, 0);
    if (n < 2) return true;
    std::vector<u64> pre(n);
    u64 acc = 1 % q;
    for (size_t i = 1; i < n; i++) {
        pre[i] = acc;
        acc = host::mul_mod(acc, t[i] % q, q);
    }
    u64 inv = host::inv_mod(acc, q);
    if (!inv && q != 1) return false;
    for (size_t i = n - 1; i >= 1; i--) {
        out[i] = host::mul_mod(inv, pre[i], q);
        inv = host::mul_mod(inv, t[i] % q, q);
    }
    return true;
}

// Build device tables from forward tables in canonical residues (count x N).
int build_tables(fhe_ctx *ctx, int log_n, const u64 *q, int count, const u64 *fwd_rows, bool want_inverse, int force_path,
                 const u64 *psi_or_null, fhe_ntt_tables **out)
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
        if (want_inverse) inv_ok = batch_inverse(row, N, q[l], inv);
        if (inv_ok) {
            inv[0] = 1 % q[l];
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
        const u64 ninv = host::inv_mod((u64)(N % q[l]), q[l]);
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

// one launch per maximal run of limbs that share an arithmetic path
template <class F> int for_each_run(const fhe_ntt_tables *t, size_t limbs, size_t start_idx, F f)
{
    size_t i = 0;
    while (i < limbs) {
        size_t j = i + 1;
        while (j < limbs && t->path[start_idx + j] == t->path[start_idx + i]) j++;
        int rc = f(i, j - i, t->path[start_idx + i]);
        if (rc != FHE_OK) return rc;
        i = j;
    }
    return FHE_OK;
}

// Trace scope: when tracing is on, synchronises the stream at both ends and appends one line.
struct TraceScope {
    fhe_ctx *ctx;
    hipStream_t st;
    const char *tag;
    bool frontend;
    std::chrono::steady_clock::time_point t0;
    TraceScope(fhe_ctx *c, hipStream_t s, const char *t, bool fe = false) : ctx(c), st(s), tag(t), frontend(fe)
    {
        if (!ctx->trace_on) return;
        (void)hipStreamSynchronize(st);
        if (frontend) ctx->trace += std::string("frontend: ") + tag + "\n";
        t0 = std::chrono::steady_clock::now();
    }
    ~TraceScope()
    {
        if (!ctx->trace_on) return;
        (void)hipStreamSynchronize(st);
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        char line[128];
        // enclosing scopes use SEAL's own "<layer>: TAG[n microseconds]" spelling, which the reference's
        // tools skip; only leaf steps are "[TAG] total cost" lines (as in profile_framewk/build/sample.txt)
        if (frontend) std::snprintf(line, sizeof line, "frontend: %s[%lld microseconds]\n", tag, us);
        else if (!std::strcmp(tag, "KEYSWITCH")) std::snprintf(line, sizeof line, "evaluator: %s[%lld microseconds]\n", tag, us);
        else std::snprintf(line, sizeof line, "[%s] total cost %lld \xC2\xB5s\n", tag, us);
        ctx->trace += line;
    }
};

int check_range(const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx)
{
    if (!t) return fail(FHE_ERR_INVALID, "null tables");
    if (start_idx + limbs > (size_t)t->count) return fail(FHE_ERR_INVALID, "limb range exceeds the table set");
    if (n_poly * limbs > ((size_t)1 << 24)) return fail(FHE_ERR_INVALID, "batch too large for one launch (max 2^24 limb-polynomials)");
    return FHE_OK;
}

int ntt_batch(fhe_ctx *ctx, u64 *d, const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx, void *stream,
              bool inverse)
{
    if (!ctx || !d) return fail(FHE_ERR_INVALID, "null argument");
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
                HIP_TRY(ctl->alloc(need * 2));
            }
            e = launch_ntt_fused(st, a, t->log_n, inverse, path, ctl->as<u32>(), ctx->fused_dist, ctx->fused_wgs, ctx->fused_variant, ctx->fused_skip_teams);
        } else if (ctx->fault_idx >= 0 && t->log_n >= 13) {
            // fault-injection hook: corrupt the intermediate between the two launches (one shot)
            e = launch_ntt(st, a, t->log_n, inverse, path, ctx->geo, 0);
            if (e == hipSuccess) e = launch_flip_bit(st, d, (u64)ctx->fault_idx, ctx->fault_bit);
            if (e == hipSuccess) e = launch_ntt(st, a, t->log_n, inverse, path, ctx->geo, 1);
            ctx->fault_idx = -1;
        } else {
            e = launch_ntt(st, a, t->log_n, inverse, path, ctx->geo, ctx->only_pass, ctx->resident);
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

ModConst mod_const(u64 q)
{
    u64 cr[3];
    host::const_ratio(q, cr);
    return ModConst{q, cr[0], cr[1]};
}

int ilog2_exact(u64 v)
{
    if (!v || (v & (v - 1))) return -1;
    return 63 - __builtin_clzll(v);
}

} // namespace

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
        if (flag) return fail(FHE_ERR_HIP, "fused NTT: a bounded wait ran out (results of that launch are invalid)");
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
    fhe_ntt_tables *t = nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        auto key = std::make_tuple(log_n, (u64)mod, use_root, convention);
        auto it = ctx->cyclic.find(key);
        if (it == ctx->cyclic.end()) {
            std::vector<u64> tw(n);
            host::cyclic_table(mod, log_n, use_root, convention == 1, tw.data());
            fhe_ntt_tables *nt = nullptr;
            u64 q = mod;
            int rc = build_tables(ctx, log_n, &q, 1, tw.data(), false, -1, nullptr, &nt);
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
    e = launch_bitrev_scale(st, d_scratch, d_data, log_n, (u32)n_vec, mod_const(mod), scale, inverse != 0);
    if (e != hipSuccess) return hip_fail(e, "launch_bitrev_scale");
    HIP_TRY(hipMemcpyAsync(d_data, d_scratch, n_vec * n * sizeof(u64), hipMemcpyDeviceToDevice, st));
    return FHE_OK;
}

// ---------------------------------------------------------------- four-step
int fhe_fourstep_create(fhe_ctx *ctx, uint64_t n1, uint64_t n2, uint64_t mod, uint64_t g, fhe_fourstep **out)
{
    if (!ctx || !out) return fail(FHE_ERR_INVALID, "null argument");
    const int l1 = ilog2_exact(n1), l2 = ilog2_exact(n2);
    if (l1 < 1 || l2 < 1 || l1 > NTT_MAX_LOGN || l2 > NTT_MAX_LOGN || l1 + l2 > 26 || mod < 2)
        return fail(FHE_ERR_INVALID, "n1 and n2 must be powers of two >= 2");
    const u64 N = n1 * n2;
    if ((mod - 1) % N) return fail(FHE_ERR_INVALID, "N must divide mod - 1");
    std::unique_ptr<fhe_fourstep> p(new fhe_fourstep);
    p->ctx = ctx;
    p->n1 = n1;
    p->n2 = n2;
    p->mod = mod;
    p->log1 = l1;
    p->log2 = l2;
    p->mc = mod_const(mod);
    // sub-transforms: length n2 with root w^n1 and length n1 with root w^n2 are both
    // "generator" transforms of g (four_step_ntt_prot.py:76-78): wlen(len) = g^((mod-1)/len)
    std::vector<u64> tw2(n2), tw1(n1);
    host::cyclic_table(mod, l2, g, false, tw2.data());
    host::cyclic_table(mod, l1, g, false, tw1.data());
    u64 q = mod;
    int rc = build_tables(ctx, l2, &q, 1, tw2.data(), false, -1, nullptr, &p->t2);
    if (rc) return rc;
    rc = build_tables(ctx, l1, &q, 1, tw1.data(), false, -1, nullptr, &p->t1);
    if (rc) {
        fhe_ntt_tables_destroy(p->t2);
        return rc;
    }
    // C[t1][k2] = B[t1][k2] * w^(k2*t1)  (four_step_ntt_prot.py:93)
    const u64 w = host::pow_mod(g, (mod - 1) / N, mod);
    std::vector<u64> tw(N);
    for (u64 t1 = 0; t1 < n1; t1++) {
        const u64 step = host::pow_mod(w, t1, mod);
        u64 cur = 1 % mod;
        for (u64 k2 = 0; k2 < n2; k2++) {
            tw[t1 * n2 + k2] = cur;
            cur = host::mul_mod(cur, step, mod);
        }
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(p->tw.upload(tw));
    HIP_TRY(p->buf0.alloc(N * sizeof(u64)));
    HIP_TRY(p->buf1.alloc(N * sizeof(u64)));
    *out = p.release();
    return FHE_OK;
}

int fhe_fourstep_destroy(fhe_fourstep *p)
{
    if (p) {
        (void)hipSetDevice(p->ctx->device);
        fhe_ntt_tables_destroy(p->t1);
        fhe_ntt_tables_destroy(p->t2);
        delete p;
    }
    return FHE_OK;
}

int fhe_fourstep_ntt(fhe_ctx *ctx, uint64_t *d_dst, const uint64_t *d_src, fhe_fourstep *p, void *stream)
{
    if (!ctx || !d_dst || !d_src || !p) return fail(FHE_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = pick(ctx, stream);
    u64 *b0 = p->buf0.as<u64>(), *b1 = p->buf1.as<u64>();
    const u32 n1 = (u32)p->n1, n2 = (u32)p->n2;
    hipError_t e;
    // A[t2][t1] = a[t1 + n1 t2] (:81) -> b0[t1][t2]
    if ((e = launch_transpose(st, b0, d_src, n2, n1)) != hipSuccess) return hip_fail(e, "transpose");
    // step 1 (:84-90): n1 transforms of length n2 along t2; output bit-reversed in k2
    PassArgs a1{b0, p->t2->d_lp.as<LimbParams>(), 0u, 1u, n1, 1u};
    if ((e = launch_ntt(st, a1, p->log2, false, p->t2->path[0])) != hipSuccess) return hip_fail(e, "four-step column transforms");
    // step 2 (:93) fused with the reordering: b1[k2][t1] = b0[t1][bitrev(k2)] * w^(k2 t1)
    if ((e = launch_fourstep_mid(st, b1, b0, n1, n2, p->log2, p->tw.as<u64>(), p->mc, true)) != hipSuccess)
        return hip_fail(e, "four-step twiddle");
    // step 3 (:96-102): n2 transforms of length n1 along t1
    PassArgs a2{b1, p->t1->d_lp.as<LimbParams>(), 0u, 1u, n2, 1u};
    if ((e = launch_ntt(st, a2, p->log1, false, p->t1->path[0])) != hipSuccess) return hip_fail(e, "four-step row transforms");
    // step 4 (:105-108): y[k1*n2 + k2] = Y[k1][k2] = b1[k2][bitrev(k1)]
    if ((e = launch_fourstep_mid(st, d_dst, b1, n2, n1, p->log1, nullptr, p->mc, false)) != hipSuccess)
        return hip_fail(e, "four-step output reorder");
    return FHE_OK;
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
static int keyswitch_core(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c, const uint64_t *d_evk,
                          const uint64_t *d_add0, void *stream);

int fhe_keyswitch_apply(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c,
                        const uint64_t *d_evk, void *stream)
{
    return keyswitch_core(ctx, p, d_out0, d_out1, d_c, d_evk, nullptr, stream);
}

// d_add0 (optional, L x N): added to the first output part -- a rotation passes sigma(c0) here
static int keyswitch_core(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c, const uint64_t *d_evk,
                          const uint64_t *d_add0, void *stream)
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
    const SubScaleArgs sa{d_out0, d_out1, acc, conv, d_add0, p->pinv.as<u64>(), (u64)(M * N), (u64)(L * N), lp, 0u, (u32)L, p->log_n};
    if ((e = launch_sub_scale(st, sa)) != hipSuccess) return hip_fail(e, "launch_sub_scale");
    return FHE_OK;
}

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
    return keyswitch_core(ctx, p, d_out0, d_out1, sig1, d_galois_key, sig0, st);
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
    p->dev = BaseConvPlanDev{m, k, p->mod_in.as<u64>(), p->mod_out.as<u64>(), p->dig.as<Tw>(), p->hor.as<Tw>(), p->fp_in.as<Tw>(),
                             p->fp_out.as<Tw>(), f64 ? 1 : 0, p->fast_coef.as<u64>(), p->fast_shoup.as<u64>()};
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
