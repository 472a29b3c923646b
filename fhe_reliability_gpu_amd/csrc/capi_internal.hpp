// capi_internal.hpp -- shared by the translation units that implement the C ABI (capi*.cpp): the opaque handle
// types, error reporting, device buffers, trace scopes and the few helpers more than one unit needs.
#pragma once
#include "../../include/fhe_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
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


// ---- error reporting (one message per host thread, read through fhe_last_error) ----

inline thread_local std::string g_err;

inline int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
inline int hip_fail(hipError_t e, const char *what)
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
    DevBuf mod_in, mod_out, dig, hor, fp_in, fp_out, fast_coef, fast_shoup, img_head, img_out, rows_id;
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
    // packed hand-off area of the forward 2^16 transform, one per stream the caller launches on
    std::map<hipStream_t, std::unique_ptr<DevBuf>> packed;
    bool packed_on = false;   // "ntt_packed"
    int mode = 0;          // 0 = two launches per transform (default), 1 = fused launch (experimental)
    unsigned fused_dist = 4, fused_wgs = 768;
    bool hmult_fused_rescale = true;   // fhe_hmult: mod-down and rescale behind one forward transform (FHE_HMULT_FUSED_RESCALE=0: the two steps apart)
    unsigned fused_skip_teams = 0;
    bool trace_on = false;
    std::string trace;          // collected trace text (fhe_ctx_trace)
    long long fault_idx = -1;   // one-shot mid-transform bit flip (fhe_ctx_inject_fault)
    int fault_bit = 0;
    int pfault_pass = -1, pfault_bit = 0;       // one-shot bit flip INSIDE a pass of the per-phase checked transform
    u32 pfault_block = 0, pfault_word = 0;      // (fhe_ctx_inject_fault_in_pass): workgroup and LDS word
    int geo = 1;           // column-tile geometry of the two-launch path (ntt_launch.hpp)
    bool resident = false; // 2^13 / 2^14: one LDS-resident pass instead of two launches (opt-in, see ntt_plan.hpp)
    int pingpong = -1;        // "ntt_pingpong": two-launch transforms hand over through a per-stream scratch buffer (both launches out of
                              // place: +6 % on batches that stream from HBM, profiles/r02_variant_sweep.txt); -1 = for calls that are sub-batched
    std::map<hipStream_t, std::unique_ptr<DevBuf>> pp_tmp;
    // "ntt_split": sub-batches of one call alternate between the caller's stream and a side stream the context owns (fork / join by
    // events), so that one sub-batch's row pass overlaps the next one's column pass; -1 = when the call is sub-batched
    struct Side { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; DevBuf tmp; };
    std::map<hipStream_t, std::unique_ptr<Side>> side;
    int split = -1;
    int stream_hint = -1;  // "ntt_stream": non-temporal accesses on the external side of sub-batched transforms (-1 = when the call is cut, 0 / 1)
    unsigned chunk_floor_mib = 192;   // batches up to this size (and up to 1.5 sub-batches) are never cut
    unsigned chunk_mib = 96;  // two-launch transforms of larger batches run as sub-batches of this size (0 = off), capi.cpp ntt_batch
    int ks_fused = -1;     // key-switch inner product fused with the extended limbs' row pass: -1 = by shape, 0 = never, 1 = always (where supported)
    int only_pass = -1;    // measurement hook: 0 / 1 = launch only the first / second pass of a two-pass size
    int fused_variant = 7;   // handoff*2 + stream hint (ntt_launch.hpp); 7 = acquire + nt streaming
    // cyclic tables keyed by (log_n, mod, root, convention * 2 + natural-order form, scale folded into the last stage)
    std::map<std::tuple<int, u64, u64, int, u64>, std::unique_ptr<fhe_ntt_tables>> cyclic;
    std::map<std::vector<u64>, std::unique_ptr<GarnerTables>> garner;
};

struct fhe_abft {
    fhe_ctx *ctx = nullptr;
    const fhe_ntt_tables *t = nullptr;
    DevBuf w, what, ninv;       // count x N weights (input side / output side), N^-1 per limb
    DevBuf win, wout, wout8;    // weights for the fused checksums: twiddle-encoded (ArithU64 limbs) and, for the output side,
                                // as residues (ArithF64 limbs); N^-1 is folded into the output-side weights
    DevBuf sum_in, sum_out;     // scratch checksums (grown on demand)
    DevBuf umid, umid8;         // per-phase detector: weights on the hand-off between the two launches, u = P1^-T w (twiddle-encoded / residues)
    DevBuf sum_mid1, sum_mid2;  // hand-off checksums as stored by the column pass / as loaded by the row pass
};

// Which limbs a rank of a limb-sharded key switch owns: ciphertext limbs [clo, clo + cn) and special limbs
// [slo, slo + sn) (table indices, slo >= L); cmax / smax = the largest slab of any rank (rows per rank in the two
// gather buffers).  world = 1: everything.
struct KsShard {
    int world = 1, rank = 0, clo = 0, cn = 0, slo = 0, sn = 0, cmax = 0, smax = 0;
};
void ks_shard_layout(int L, int K, int world, int rank, KsShard *s);

struct fhe_keyswitch {
    fhe_ctx *ctx = nullptr;
    const fhe_ntt_tables *t = nullptr;
    int L = 0, K = 0, dnum = 0, alpha = 0, log_n = 0;
    KsShard sh;                         // owned limbs (capi_keyswitch.cpp)
    bool sharded = false;               // caller-owned gather buffers + the three phases (always when world > 1)
    int m_own = 0;                      // cn + sn: rows of ext / acc / the key on this rank
    u64 *g1 = nullptr, *g2 = nullptr;   // gather buffers: [world][cmax][N] coefficient-form input, [world][2][smax][N] special limbs
    DevBuf up_rows, down_rows;          // row of each conversion input limb inside g1 / g2 (or acc on one device)
    bool up_f64 = false;
    bool own_path[2] = {false, false};  // arithmetic paths present among the owned limbs
    bool up_trivial = false;            // one-limb digits (dnum = L, SEAL's form) at a two-launch size: the extension x mod q_j rides on the
                                        // extended limbs' column pass (UnitRef::src), no conversion launch and no converted copy
    u32 down_src_row = 0, down_src_stride = 0;   // K = 1: row of the special limb of half 0 in acc / gather buffer 2, rows between the halves
    DevBuf d_t_mod_Q;                   // plain_modulus mod q_j on the device (BGV factor of the fused tails)
    u32 n_up_jobs = 0;
    std::vector<BcJob> up_host;         // host copy of the digit jobs (mixed arithmetic paths: one launch per digit)
    u64 plain_modulus = 0;              // BGV: delta must vanish modulo this (0 = CKKS-style flooring)
    std::vector<u64> t_inv_P, t_mod_Q;  // plain_modulus^-1 mod p_k, plain_modulus mod q_j
    std::vector<fhe_baseconv *> up;     // per digit: digit primes -> every other owned prime (ascending row; nullptr: nothing to extend to)
    fhe_baseconv *down = nullptr;       // P -> owned ciphertext primes
    DevBuf pinv;                        // P^-1 mod q_j, owned j
    DevBuf coef, ext, acc, conv, rot;  // coef [L][N] (one device: = g1), ext [dnum][m_own][N], acc [2][m_own][N], conv [2][cn][N]
    DevBuf up_jobs, down_jobs;         // device job lists: all digit extensions / both mod-down conversions in one launch each
    int up_max_m = 0, up_max_k = 0;
    u32 up_m_mask = 0;                 // bit (m - 1): a digit of m limbs exists (fixed-size conversion kernels, one launch per size)
    bool up_batched = false;           // every digit plan on the same arithmetic path
    DevBuf ext_map[2];                 // per arithmetic path: the limbs of ext the forward transform covers
    u32 ext_units[2] = {0, 0};
    // homomorphic multiply / rescale on top of the key switch (capi_hmult.cpp); rescale needs L >= 2
    fhe_baseconv *last = nullptr;      // q_{L-1} -> the owned ciphertext primes below it
    int rs_n = 0;                      // how many those are (one device: L-1); rescale outputs have rs_n rows per part
    bool own_last = false;             // this rank owns limb L-1 (it feeds the broadcast)
    u64 *rs_bc = nullptr;              // [3][N] last limbs in coefficient form: plan-owned on one device, the caller's broadcast buffer when sharded
    DevBuf qlast_inv;                  // q_{L-1}^-1 mod q_j, owned j < L-1
    DevBuf rs_last, rs_delta, rs_jobs; // rs_last backs rs_bc on one device; [3][rs_n][N] residues; job list (3 parts)
    DevBuf hm, hm_pre;                 // [3][L][N] tensor product, [2][L][N] relinearised product before the rescale
    DevBuf pq_tw;                      // one device: P mod q_j as a twiddle of limb j's arithmetic, j < L (mod-down and rescale sharing one transform)
    DevBuf bsgs;                       // fhe_bsgs_matvec: baby rotations, inner sum, one rotated inner sum (grown on demand)
    // second set of per-rotation buffers: hoisted rotations alternate between the caller's stream and a side stream (one rotation's
    // conversions run under the next one's inner product); `cur` selects the set the host-side helpers address (launch arguments are
    // taken at launch time, so flipping it between launches is safe under the one-plan-one-caller rule)
    DevBuf acc2, conv2, hsp2, hdown_jobs2;
    int cur = 0;
    DevBuf acc_multi;                  // hoisted batches: the sums of up to four rotations formed in one pass over the shared digits ([4][2][MO][N])
    u64 *acc_ovr = nullptr;            // != nullptr: the sums the mod-down in flight works on (one slot of acc_multi)
    u64 *acc_cur() const { return acc_ovr ? acc_ovr : (cur ? acc2 : acc).as<u64>(); }
    u64 *conv_cur() const { return (cur ? conv2 : conv).as<u64>(); }
    u64 *hsp_cur() const { return (cur ? hsp2 : hsp).as<u64>(); }
    DevBuf hsp, hdown_rows, hdown_jobs; // hoisted rotations (allocated on first use): [2][K][N] special limbs of sigma(sums) in coefficient form, the mod-down jobs that read them
    u64 t_inv_qlast = 0;               // plain_modulus^-1 mod q_{L-1} (BGV)
    ~fhe_keyswitch()
    {
        for (auto *b : up) fhe_baseconv_destroy(b);
        fhe_baseconv_destroy(down);
        fhe_baseconv_destroy(last);
    }
};

struct fhe_fourstep {
    fhe_ctx *ctx = nullptr;
    u64 n1 = 0, n2 = 0, mod = 0, g = 0;
    int log_n = 0;
    fhe_ntt_tables *t = nullptr;   // natural-order table set of length n1 * n2 (owned by the context's cache)
    DevBuf tmp;                    // hand-off buffer between the two launches, grown to the largest batch seen
    // n1 * n2 past the largest single plan (2^21 .. 2^26): the reference's own composition -- transpose, n1 transforms of length n2,
    // twiddle w^(k2 t1) on the way through the second transpose, n2 transforms of length n1, transpose
    bool big = false;
    int log1 = 0, log2 = 0, lo_bits = 0;
    DevBuf tw_lo, tw_hi, buf0, buf1;
};



inline hipStream_t pick(fhe_ctx *ctx, void *stream) { return stream ? static_cast<hipStream_t>(stream) : ctx->stream; }

inline int path_for(u64 q)
{
    if (q < 2) return -1;
    if (q < ((u64)1 << 50)) return PATH_F64;
    if (q < ((u64)1 << 61)) return PATH_U64;
    return -1;
}

inline Tw encode(int path, u64 w, u64 q) { return path == PATH_F64 ? ArithF64::encode(w, q) : ArithU64::encode(w, q); }

// entry-wise inverses of t[1..n-1] mod q by Montgomery's batch trick; false when an
// entry is not a unit
inline bool batch_inverse(const u64 *t, size_t n, u64 q, std::vector<u64> &out)
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

inline int check_range(const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx)
{
    if (!t) return fail(FHE_ERR_INVALID, "null tables");
    if (start_idx + limbs > (size_t)t->count) return fail(FHE_ERR_INVALID, "limb range exceeds the table set");
    if (n_poly * limbs > ((size_t)1 << 24)) return fail(FHE_ERR_INVALID, "batch too large for one launch (max 2^24 limb-polynomials)");
    return FHE_OK;
}



inline ModConst mod_const(u64 q)
{
    u64 cr[3];
    host::const_ratio(q, cr);
    return ModConst{q, cr[0], cr[1]};
}

inline int ilog2_exact(u64 v)
{
    if (!v || (v & (v - 1))) return -1;
    return 63 - __builtin_clzll(v);
}


// defined in capi_keyswitch.cpp: the key switch of d_c with d_add0 / d_add1 (optional, L x N) added to the two output parts;
// rescale (only where ks_rescale_fusable() says so): the outputs are the (L-1)-limb parts after dropping q_{L-1} as well
int keyswitch_core(fhe_ctx *ctx, fhe_keyswitch *p, uint64_t *d_out0, uint64_t *d_out1, const uint64_t *d_c, const uint64_t *d_evk,
                   const uint64_t *d_add0, const uint64_t *d_add1, void *stream, bool rescale = false);
bool ks_rescale_fusable(const fhe_ctx *ctx, const fhe_keyswitch *p);
// defined in capi.cpp
int build_tables(fhe_ctx *ctx, int log_n, const fhe::u64 *q, int count, const fhe::u64 *fwd_rows, bool want_inverse, int force_path,
                 const fhe::u64 *psi_or_null, fhe_ntt_tables **out, const fhe::u64 *gs_scale = nullptr);
int cyclic_tables(fhe_ctx *ctx, int log_n, fhe::u64 mod, fhe::u64 root, int convention, fhe::u64 scale, fhe_ntt_tables **out);
struct SubBatchCut {
    size_t pc, lc;      // polynomials x limbs per piece; pc = 0: no cut
};
SubBatchCut sub_batch_cut(const fhe_ctx *ctx, int log_n, size_t n_poly, size_t len, size_t bufs = 1);
size_t sub_batch_polys(const fhe_ctx *ctx, int log_n, size_t n_poly, size_t len);
int side_stream(fhe_ctx *ctx, hipStream_t st, fhe_ctx::Side **out);
int for_pieces(fhe_ctx *ctx, hipStream_t st, size_t n_pieces, size_t side_tmp_bytes, const std::function<hipError_t(hipStream_t, size_t, fhe::u64 *)> &fn);
hipError_t handoff_scratch(fhe_ctx *ctx, hipStream_t st, size_t bytes, fhe::u64 **out);
int for_sub_batches(fhe_ctx *ctx, hipStream_t st, size_t n_poly, size_t per, size_t side_tmp_bytes,
                    const std::function<hipError_t(hipStream_t, size_t, size_t, fhe::u64 *)> &fn);
int ntt_batch(fhe_ctx *ctx, fhe::u64 *d, const fhe_ntt_tables *t, size_t n_poly, size_t limbs, size_t start_idx, void *stream, bool inverse,
              const fhe::u64 *d_src = nullptr, fhe::u32 galois = 0, fhe::u64 *d_galois_copy = nullptr);
int pointwise(fhe_ctx *ctx, fhe::u64 *c, const fhe::u64 *a, const fhe::u64 *b, const fhe_ntt_tables *t, size_t n_poly, size_t limbs,
              size_t start_idx, void *stream, bool acc);

