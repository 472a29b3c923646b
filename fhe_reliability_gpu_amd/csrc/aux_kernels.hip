// aux_kernels.hip -- HBM-bound helpers around the transforms: coefficient-wise
// modular products, bit-reversal gather, four-step glue, bit flips, base conversion,
// Garner CRT and the BSGS Hadamard accumulate.  All integer work, one element (or one
// coefficient column) per lane, 64-bit coalesced accesses along N.
#include "ntt_launch.hpp"

#include <cstdlib>
#include <type_traits>

namespace fhe {

__device__ __forceinline__ u64 mulmod_b(u64 a, u64 b, u64 q, u64 r0, u64 r1)
{
    return barrett128(a * b, __umul64hi(a, b), q, r0, r1);
}
// Shoup product with full correction: exact (a*w mod q) for ANY 64-bit a, w < q
__device__ __forceinline__ u64 mulmod_shoup(u64 a, u64 w, u64 ws, u64 q)
{
    u64 r = a * w - __umul64hi(a, ws) * q;
    return r >= q ? r - q : r;
}

// ---------------------------------------------------------------------------
// NT: operands and result past the Infinity Cache (touched once): non-temporal accesses
template <bool ACC, bool NT>
__global__ __launch_bounds__(256) void k_modmul(PointwiseArgs p)
{
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const u64 n = (u64)1 << p.logn;
    const u64 total = (u64)p.units << p.logn;
    for (u64 i_ = (blockIdx.x * (u64)blockDim.x + threadIdx.x) * 2; i_ < total; i_ += (u64)gridDim.x * blockDim.x * 2) {
        const u32 unit = (u32)(i_ >> p.logn);
        const u32 poly = unit / p.limbs, l = unit % p.limbs;
        const LimbParams &lp = p.lp[p.limb0 + l];
        const u64 q = lp.q, r0 = lp.barrett_lo, r1 = lp.barrett_hi;
        {
            const u64 i = (((u64)poly * p.poly_stride + l) << p.logn) + (i_ & (n - 1));
            ulonglong2 a, b;
            if (NT) {
                const u64x2 va = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(p.a + i)), vb = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(p.b + i));
                a = ulonglong2{va.x, va.y};
                b = ulonglong2{vb.x, vb.y};
            } else {
                a = *reinterpret_cast<const ulonglong2 *>(p.a + i);
                b = *reinterpret_cast<const ulonglong2 *>(p.b + i);
            }
            ulonglong2 c;
            c.x = mulmod_b(a.x, b.x, q, r0, r1);
            c.y = mulmod_b(a.y, b.y, q, r0, r1);
            if (ACC) {
                const ulonglong2 o = *reinterpret_cast<const ulonglong2 *>(p.c + i);
                u64 x = c.x + barrett128(o.x, 0, q, r0, r1), y = c.y + barrett128(o.y, 0, q, r0, r1);
                c.x = x >= q ? x - q : x;
                c.y = y >= q ? y - q : y;
            }
            if (NT) __builtin_nontemporal_store(u64x2{c.x, c.y}, reinterpret_cast<u64x2 *>(p.c + i));
            else *reinterpret_cast<ulonglong2 *>(p.c + i) = c;
        }
    }
}

// c = (a + b) mod q per limb
__global__ __launch_bounds__(256) void k_modadd(PointwiseArgs p)
{
    const u64 n = (u64)1 << p.logn;
    const u64 total = (u64)p.units << p.logn;
    for (u64 i_ = blockIdx.x * (u64)blockDim.x + threadIdx.x; i_ < total; i_ += (u64)gridDim.x * blockDim.x) {
        const u32 unit = (u32)(i_ >> p.logn);
        const u32 poly = unit / p.limbs, l = unit % p.limbs;
        const LimbParams &lp = p.lp[p.limb0 + l];
        const u64 q = lp.q, r0 = lp.barrett_lo, r1 = lp.barrett_hi;
        const u64 i = (((u64)poly * p.poly_stride + l) << p.logn) + (i_ & (n - 1));
        const u64 s = barrett128(p.a[i], 0, q, r0, r1) + barrett128(p.b[i], 0, q, r0, r1);
        p.c[i] = s >= q ? s - q : s;
    }
}

hipError_t launch_modadd(hipStream_t st, const PointwiseArgs &p)
{
    const u64 total = (u64)p.units << p.logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_modadd, dim3((u32)(want > 8192 ? 8192 : want)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// c = (a - b) mod q per limb
__global__ __launch_bounds__(256) void k_modsub(PointwiseArgs p)
{
    const u64 n = (u64)1 << p.logn;
    const u64 total = (u64)p.units << p.logn;
    for (u64 i_ = blockIdx.x * (u64)blockDim.x + threadIdx.x; i_ < total; i_ += (u64)gridDim.x * blockDim.x) {
        const u32 unit = (u32)(i_ >> p.logn);
        const u32 poly = unit / p.limbs, l = unit % p.limbs;
        const LimbParams &lp = p.lp[p.limb0 + l];
        const u64 q = lp.q, r0 = lp.barrett_lo, r1 = lp.barrett_hi;
        const u64 i = (((u64)poly * p.poly_stride + l) << p.logn) + (i_ & (n - 1));
        const u64 x = barrett128(p.a[i], 0, q, r0, r1), y = barrett128(p.b[i], 0, q, r0, r1);
        p.c[i] = x >= y ? x - y : x + q - y;
    }
}

hipError_t launch_modsub(hipStream_t st, const PointwiseArgs &p)
{
    const u64 total = (u64)p.units << p.logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_modsub, dim3((u32)(want > 8192 ? 8192 : want)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// c = a * s_l + t_l mod q_l with per-limb scalars passed by value (scalar multiply / scalar add)
__global__ __launch_bounds__(256) void k_scalar_affine(PointwiseArgs p, ScalarVec mul, ScalarVec add)
{
    const u64 n = (u64)1 << p.logn;
    const u64 total = (u64)p.units << p.logn;
    for (u64 i_ = blockIdx.x * (u64)blockDim.x + threadIdx.x; i_ < total; i_ += (u64)gridDim.x * blockDim.x) {
        const u32 unit = (u32)(i_ >> p.logn);
        const u32 poly = unit / p.limbs, l = unit % p.limbs;
        const LimbParams &lp = p.lp[p.limb0 + l];
        const u64 q = lp.q, r0 = lp.barrett_lo, r1 = lp.barrett_hi;
        const u64 i = (((u64)poly * p.poly_stride + l) << p.logn) + (i_ & (n - 1));
        u64 v = mulmod_b(p.a[i], mul.v[l], q, r0, r1) + add.v[l];
        p.c[i] = v >= q ? v - q : v;
    }
}

hipError_t launch_scalar_affine(hipStream_t st, const PointwiseArgs &p, const ScalarVec &mul, const ScalarVec &add)
{
    const u64 total = (u64)p.units << p.logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_scalar_affine, dim3((u32)(want > 8192 ? 8192 : want)), dim3(256), 0, st, p, mul, add);
    return hipGetLastError();
}

hipError_t launch_modmul(hipStream_t st, const PointwiseArgs &p, bool accumulate)
{
    const u64 total = (u64)p.units << p.logn;
    if (!total) return hipSuccess;
    u64 want = (total / 2 + 255) / 256;
    const u32 blocks = (u32)(want < 1 ? 1 : want > 8192 ? 8192 : want);
    const bool nt = total * 24 > ((u64)192 << 20);      // three buffers of the batch's size
    if (accumulate) hipLaunchKernelGGL((k_modmul<true, false>), dim3(blocks), dim3(256), 0, st, p);
    else if (nt) hipLaunchKernelGGL((k_modmul<false, true>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_modmul<false, false>), dim3(blocks), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ void k_flip_bit(u64 *data, u64 idx, int bit) { data[idx] ^= (u64)1 << bit; }

hipError_t launch_flip_bit(hipStream_t st, u64 *data, u64 idx, int bit)
{
    hipLaunchKernelGGL(k_flip_bit, dim3(1), dim3(1), 0, st, data, idx, bit);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bitrev_scale(u64 *dst, const u64 *src, int logn, u64 total, ModConst mc, u64 sc,
                                                      bool do_scale)
{
    const u64 q = mc.q, r0 = mc.r0, r1 = mc.r1;
    const u64 mask = ((u64)1 << logn) - 1;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 lo = i & mask;
        const u64 j = logn ? (__brevll(lo) >> (64 - logn)) : 0;
        u64 v = src[(i & ~mask) | j];
        if (do_scale) v = mulmod_b(v, sc, q, r0, r1);
        dst[i] = v;
    }
}

hipError_t launch_bitrev_scale(hipStream_t st, u64 *dst, const u64 *src, int logn, u32 units, const ModConst &mc, u64 scale,
                               bool do_scale)
{
    const u64 total = (u64)units << logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_bitrev_scale, dim3((u32)(want > 8192 ? 8192 : want)), dim3(256), 0, st, dst, src, logn, total, mc,
                       scale, do_scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// dst[v][c][r] = src[v][r][c] (* w^(r c) mod q): the transposes of the four-step flow for N past the largest single plan
// (reliability_test/four_step_ntt_prot.py:81, 93, 105-108), 32 x 32 tiles through LDS so that both sides move whole 256-byte
// segments.  w^(r c) from two tables: w^e = tlo[e & (2^lo_bits - 1)] * thi[e >> lo_bits].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose_tw(u64 *dst, const u64 *src, u32 rows, u32 cols, ModConst mc, const u64 *tlo, const u64 *thi, int lo_bits)
{
    __shared__ u64 tile[32][33];
    const u64 q = mc.q, r0 = mc.r0, r1 = mc.r1;
    const size_t vec = (size_t)blockIdx.z * rows * cols;
    const u32 c0 = blockIdx.x * 32, rr0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (u32 k = ty; k < 32; k += 8) {
        const u32 r = rr0 + k, c = c0 + tx;
        if (r < rows && c < cols) {
            u64 v = src[vec + (size_t)r * cols + c];
            if (tlo) {
                const u64 e = (u64)r * c;
                const u64 w = mulmod_b(tlo[e & (((u64)1 << lo_bits) - 1)], thi[e >> lo_bits], q, r0, r1);
                v = mulmod_b(v < q ? v : barrett128(v, 0, q, r0, r1), w, q, r0, r1);
            }
            tile[k][tx] = v;
        }
    }
    __syncthreads();
    for (u32 k = ty; k < 32; k += 8) {
        const u32 c = c0 + k, r = rr0 + tx;
        if (r < rows && c < cols) dst[vec + (size_t)c * rows + r] = tile[tx][k];
    }
}

hipError_t launch_transpose_tw(hipStream_t st, u64 *dst, const u64 *src, u32 rows, u32 cols, u32 n_vec, const ModConst &mc, const u64 *tlo, const u64 *thi,
                               int lo_bits)
{
    if (!rows || !cols || !n_vec) return hipSuccess;
    if (n_vec > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_transpose_tw, dim3((cols + 31) / 32, (rows + 31) / 32, n_vec), dim3(256), 0, st, dst, src, rows, cols, mc, tlo, thi, lo_bits);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Base conversion.  One lane = one coefficient; residues are read at stride N
// (coalesced across lanes), mixed-radix digits stay in registers.
// ---------------------------------------------------------------------------
constexpr int BC_MAX_LIMBS = 64;

// Exact conversion (motivation/baseConv.py:67-83): mixed-radix digits of x over (p_0, p_1, ...),
// x = c_0 + c_1 p_0 + c_2 p_0 p_1 + ..., then x mod q_o.  The reference's nested recurrence
// t = (t - c_l) p_l^-1 is unrolled into independent constant products
//   c_j = r_j A_j - sum_{l<j} c_l D_lj (mod p_j),   x mod q_o = sum_l c_l E_lo (mod q_o)
// (constants in BaseConvPlanDev), every product one lazy multiply of the limb's arithmetic path: no
// dependent multiply chain and no 128-bit Barrett step.  Same canonical digits, same outputs.
struct BcF64 {
    typedef double acc_t;
    static __device__ __forceinline__ double load(u64 raw, u64 q) { return ArithF64::from_canonical(raw < q ? raw : reduce_any_u64(raw, q)); }
    // the same without a function call (a call site makes every live register of an unrolled kernel a spill candidate): words
    // outside [0, q) -- any 64-bit pattern -- are folded in exact FP64 arithmetic, hi * (2^32 mod q) + lo; result lazy, |.| <= q/2 + 1
    static __device__ __forceinline__ bool out_of_range(u64 raw, u64 q) { return raw >= q; }
    static __device__ __forceinline__ double from_word(u64 raw) { return ArithF64::from_canonical(raw); }
    // any 64-bit word -> canonical residue, exact FP64 arithmetic, no call
    static __device__ __forceinline__ u64 fold(u64 raw, u64 q, const ArithF64::Ctx &c)
    {
        if (raw < q) return raw;
        double hi = (double)(u32)(raw >> 32), lo = (double)(u32)raw, w = 4294967296.0;
        ArithF64::reduce(hi, c);
        ArithF64::reduce(w, c);
        double r = ArithF64::mulmod_w(hi, w, w * c.ninv, c) + lo;
        return ArithF64::canonical(r, c);
    }
    static __device__ __forceinline__ double load_inline(u64 raw, u64 q, const ArithF64::Ctx &c)
    {
        if (__builtin_expect(raw < q, 1)) return ArithF64::from_canonical(raw);
        double hi = (double)(u32)(raw >> 32), lo = (double)(u32)raw, w = 4294967296.0;
        ArithF64::reduce(hi, c);
        ArithF64::reduce(w, c);
        double r = ArithF64::mulmod_w(hi, w, w * c.ninv, c) + lo;
        ArithF64::reduce(r, c);
        return r;
    }
    static __device__ __forceinline__ ArithF64::Ctx ctx(u64 q, const Tw &fp) { return ArithF64::Ctx{u64_bits_to_double(fp.a), u64_bits_to_double(fp.b), q}; }
    static __device__ __forceinline__ double mul(double a, const Tw &t, const ArithF64::Ctx &c) { return ArithF64::mulmod(a, t, c); }
    static __device__ __forceinline__ double sub(double a, double b, const ArithF64::Ctx &) { return a - b; }
    static __device__ __forceinline__ double add(double a, double b, const ArithF64::Ctx &) { return a + b; }
    // |product| < 0.75 q: eight of them stay below 2^53
    static __device__ __forceinline__ void relax(double &a, int terms, const ArithF64::Ctx &c)
    {
        if ((terms & 7) == 7) ArithF64::reduce(a, c);
    }
    static __device__ __forceinline__ double digit(double a, const ArithF64::Ctx &c)
    {
        ArithF64::reduce(a, c);
        return a < 0.0 ? a + c.n : a;
    }
    static __device__ __forceinline__ u64 out(double a, const ArithF64::Ctx &c) { return ArithF64::canonical(a, c); }
};
struct BcU64 {
    typedef u64 acc_t;
    struct Ctx {
        u64 q;
    };
    static __device__ __forceinline__ u64 load(u64 raw, u64) { return raw; }   // the Shoup product takes any 64-bit word
    static __device__ __forceinline__ u64 load_inline(u64 raw, u64, const Ctx &) { return raw; }
    static __device__ __forceinline__ bool out_of_range(u64, u64) { return false; }
    static __device__ __forceinline__ u64 from_word(u64 raw) { return raw; }
    static __device__ __forceinline__ u64 fold(u64 raw, u64, const Ctx &) { return raw; }
    static __device__ __forceinline__ Ctx ctx(u64 q, const Tw &) { return Ctx{q}; }
    static __device__ __forceinline__ u64 mul(u64 a, const Tw &t, const Ctx &c) { return mulmod_shoup(a, t.a, t.b, c.q); }
    static __device__ __forceinline__ u64 sub(u64 a, u64 b, const Ctx &c) { return a >= b ? a - b : a + c.q - b; }
    static __device__ __forceinline__ u64 add(u64 a, u64 b, const Ctx &c)
    {
        const u64 s = a + b;
        return s >= c.q ? s - c.q : s;
    }
    static __device__ __forceinline__ void relax(u64 &, int, const Ctx &) {}
    static __device__ __forceinline__ u64 digit(u64 a, const Ctx &) { return a; }
    static __device__ __forceinline__ u64 out(u64 a, const Ctx &) { return a; }
};

template <int MAXM, class B, bool STAGE_>
__device__ __forceinline__ void bc_exact_body(const BcJob &job, u64 N, u32 oc)
{
    typedef typename B::acc_t T;
    constexpr int UNR = MAXM <= 16 ? MAXM : 1;   // bases up to 16 limbs (key-switch digits): digits in registers, loops unrolled; larger ones index a scratch array
    // Short launches of small bases (N = 2^16: a single wave of workgroups) stage their constants in LDS once per
    // workgroup: read through scalar loads inside the loops they were bound by those loads' round trips.  Long
    // launches keep the scalar loads (operands in SGPRs cost nothing once other waves hide the latency).
    constexpr bool STAGE = STAGE_ && MAXM <= 16;
    constexpr int OCMAX = 64;
    __shared__ Tw s_dig[STAGE ? MAXM * MAXM : 1], s_hor[STAGE ? MAXM * OCMAX : 1], s_fpi[STAGE ? MAXM : 1], s_fpo[STAGE ? OCMAX : 1];
    __shared__ u64 s_pi[STAGE ? MAXM : 1], s_qo[STAGE ? OCMAX : 1];
    const BaseConvPlanDev &pl = job.pl;
    const u64 *__restrict__ in = job.in;
    u64 *__restrict__ out = job.out;
    const int m = pl.m, k = pl.k;
    const u32 FHE_CONSTANT *rows = (const u32 FHE_CONSTANT *)(__UINTPTR_TYPE__)job.in_rows;
    // the plan's tables are never written by a kernel: constant address space, so uniform reads become scalar loads
    // whatever the compiler can or cannot prove about `out`
    const Tw FHE_CONSTANT *dig = (const Tw FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.dig, *hor = (const Tw FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.hor;
    const Tw FHE_CONSTANT *fp_in = (const Tw FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.fp_in, *fp_out = (const Tw FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.fp_out;
    const u64 FHE_CONSTANT *mod_in = (const u64 FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.mod_in, *mod_out = (const u64 FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.mod_out;
    // blockIdx.y selects a slice of the outputs (the digits are cheap to recompute; small N needs the extra workgroups)
    const int o0 = (int)blockIdx.y * (int)oc, o1 = o0 + (int)oc < k ? o0 + (int)oc : k;
    if (STAGE) {
        const int cnt = o1 - o0;
        for (int t = threadIdx.x; t < m * m; t += blockDim.x) s_dig[(t / m) * MAXM + t % m] = dig[t];
        for (int t = threadIdx.x; t < m * cnt; t += blockDim.x) s_hor[(t / cnt) * OCMAX + t % cnt] = hor[(t / cnt) * k + o0 + t % cnt];
        for (int t = threadIdx.x; t < m; t += blockDim.x) {
            s_fpi[t] = fp_in[t];
            s_pi[t] = mod_in[t];
        }
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            s_fpo[t] = fp_out[o0 + t];
            s_qo[t] = mod_out[o0 + t];
        }
        __syncthreads();
    }
    auto c_dig = [&](int l, int j) -> Tw { if (STAGE) return s_dig[l * MAXM + j]; const Tw t = dig[l * m + j]; return t; };
    auto c_hor = [&](int l, int o) -> Tw { if (STAGE) return s_hor[l * OCMAX + o - o0]; const Tw t = hor[l * k + o]; return t; };
    auto c_fpi = [&](int j) -> Tw { if (STAGE) return s_fpi[j]; const Tw t = fp_in[j]; return t; };
    auto c_fpo = [&](int o) -> Tw { if (STAGE) return s_fpo[o - o0]; const Tw t = fp_out[o]; return t; };
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        T c[MAXM];
#pragma unroll UNR
        for (int j = 0; j < MAXM; j++) {
            if (j < m) {
                const u64 pj = STAGE ? s_pi[j] : mod_in[j];
                const auto cx = B::ctx(pj, c_fpi(j));
                const u64 row = rows ? (u64)rows[j] : (u64)j;
                T t = B::mul(B::load(in[row * N + i], pj), c_dig(j, j), cx);
#pragma unroll UNR
                for (int l = 0; l < j; l++) {
                    t = B::sub(t, B::mul(c[l], c_dig(l, j), cx), cx);
                    B::relax(t, l + 1, cx);
                }
                c[j] = B::digit(t, cx);
            }
        }
        for (int o = o0; o < o1; o++) {
            const u64 q = STAGE ? s_qo[o - o0] : mod_out[o];
            const auto cx = B::ctx(q, c_fpo(o));
            T acc = B::mul(c[0], c_hor(0, o), cx);
#pragma unroll UNR
            for (int l = 1; l < MAXM; l++) {
                if (l < m) {
                    acc = B::add(acc, B::mul(c[l], c_hor(l, o), cx), cx);
                    B::relax(acc, l, cx);
                }
            }
            out[(u64)((u32)o < job.gap_at ? o : o + job.gap) * N + i] = B::out(acc, cx);
        }
    }
}

// ---------------------------------------------------------------------------
// Bases of up to 16 limbs (every key-switch digit, every mod-down): the same arithmetic with the number of input limbs a
// compile-time constant, so that the whole conversion of a coefficient is straight-line code -- no branch per product, every
// constant's LDS read issued ahead of its use, OU outputs (x CPT coefficients) accumulated side by side.  The runtime-m body
// above compiled to one basic block per product (ds_read, wait, seven dependent FP64 instructions, branch): 40 % of the
// vector issue rate at N = 2^16, L = 44, alpha = 11 (51 us per launch).  Constants: the digit table and this workgroup's
// slice of the output table are staged in LDS once per workgroup ([output][limb], one limb's 16 bytes after the other).
// ---------------------------------------------------------------------------
// nothing moves across: neither in the optimiser (the memory clobber) nor in the instruction scheduler
__device__ __forceinline__ void bc_sched_fence()
{
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// the constants in LDS are invariant in the coefficient loop: without this the compiler hoists every read out of it (hundreds of registers)
__device__ __forceinline__ void bc_no_hoist() { __asm__ volatile("" ::: "memory"); }

template <int M, class B, int CPT, int OU>
__device__ __forceinline__ void bc_exact_fixed(const BcJob &job, u64 N, u32 oc)
{
    typedef typename B::acc_t T;
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    constexpr int OCMAX = 64, NH = M * M + M + (M + 1) / 2, REC = M + 2;
    // LDS image = the plan's img_head followed by this workgroup's slice of img_out (ntt_launch.hpp BaseConvPlanDev)
    __shared__ __attribute__((aligned(16))) Tw s_img[NH + OCMAX * REC];
    const Tw *s_dig = s_img, *s_fpi = s_img + M * M;
    const u64 *s_pi = reinterpret_cast<const u64 *>(s_img + M * M + M);
    const Tw *s_rec = s_img + NH;                         // per output: hor[0..M-1], fp_out, {mod_out, 0}
    const BaseConvPlanDev &pl = job.pl;
    const u64 FHE_GLOBAL *in = (const u64 FHE_GLOBAL *)job.in;
    u64 FHE_GLOBAL *out = (u64 FHE_GLOBAL *)job.out;
    const int k = pl.k;
    const u32 FHE_CONSTANT *rows = (const u32 FHE_CONSTANT *)(__UINTPTR_TYPE__)job.in_rows;
    const int o0 = (int)blockIdx.y * (int)oc, o1 = o0 + (int)oc < k ? o0 + (int)oc : k, cnt = o1 - o0;
    if (cnt <= 0) return;
    // word offset of input limb j (uniform).  One unconditional batch of scalar loads: the plan carries an identity row table for
    // jobs without a row map (a select per limb put every row's load behind its own branch: eleven dependent trips to the scalar cache,
    // and the workgroup size read from the dispatch packet one more, before the first residue could be requested)
    const u32 FHE_CONSTANT *rowtab = rows ? rows : (const u32 FHE_CONSTANT *)(__UINTPTR_TYPE__)pl.rows_identity;
    u64 roff[M];
#pragma unroll
    for (int j = 0; j < M; j++) roff[j] = (u64)rowtab[j] * N;
    const u64 stride = (u64)gridDim.x * 256 * CPT;            // (launched with 256 threads, launch_fixed_m)
    u64 i = (blockIdx.x * (u64)256 + threadIdx.x) * CPT;
    // the first coefficient's residues are requested BEFORE the constants are staged: one trip to memory for both
    u64 raw[M][CPT];
    auto load_raw = [&](u64 at) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            const u64 FHE_GLOBAL *src = in + roff[j] + at;
            if constexpr (CPT == 2) {
                const u64x2 v = *reinterpret_cast<const u64x2 FHE_GLOBAL *>(src);
                raw[j][0] = v.x;
                raw[j][1] = v.y;
            } else {
                raw[j][0] = *src;
            }
        }
    };
    if (i < N) load_raw(i);
    {
        // constants: ONE batch of 16-byte requests per thread (round 3's first form staged four tables in four dependent loops:
        // four trips to memory before the first product)
        typedef u64x2 FHE_GLOBAL const *gvec;
        const gvec gh = (gvec)pl.img_head, go = (gvec)pl.img_out + (size_t)o0 * REC;
        const int total = NH + cnt * REC;
        constexpr int ROUNDS = (NH + OCMAX * REC + 255) / 256;
        u64x2 v[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            int t = (int)threadIdx.x + r * 256;
            t = t < total ? t : total - 1;                  // (the tail repeats the last entry: no branch between the requests)
            v[r] = t < NH ? gh[t] : go[t - NH];
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; r++) {
            int t = (int)threadIdx.x + r * 256;
            t = t < total ? t : total - 1;
            *reinterpret_cast<u64x2 *>(s_img + t) = v[r];
        }
    }
    __syncthreads();
    for (; i < N; i += stride) {
        bc_no_hoist();
        // words outside [0, q) (the fault-injection harnesses feed them): ONE cold branch for the whole coefficient
        T x[M][CPT];
        bool bad = false;
#pragma unroll
        for (int j = 0; j < M; j++) {
#pragma unroll
            for (int e = 0; e < CPT; e++) bad |= B::out_of_range(raw[j][e], s_pi[j]);
        }
        if (__builtin_expect(bad, 0)) {
#pragma unroll
            for (int j = 0; j < M; j++) {
                const auto cx = B::ctx(s_pi[j], s_fpi[j]);
#pragma unroll
                for (int e = 0; e < CPT; e++) raw[j][e] = B::fold(raw[j][e], s_pi[j], cx);
            }
        }
#pragma unroll
        for (int j = 0; j < M; j++) {
#pragma unroll
            for (int e = 0; e < CPT; e++) x[j][e] = B::from_word(raw[j][e]);
        }
        bc_sched_fence();
        // (the fences bound how far ahead the constants' LDS reads are issued: left alone the compiler issues all of them first and
        // spills hundreds of registers)
        T c[M][CPT];
#pragma unroll
        for (int j = 0; j < M; j++) {
            const u64 pj = s_pi[j];
            const auto cx = B::ctx(pj, s_fpi[j]);
            const Tw a = s_dig[j * M + j];
            T t[CPT];
#pragma unroll
            for (int e = 0; e < CPT; e++) t[e] = B::mul(x[j][e], a, cx);
#pragma unroll
            for (int l = 0; l < j; l++) {
                const Tw w = s_dig[l * M + j];
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    t[e] = B::sub(t[e], B::mul(c[l][e], w, cx), cx);
                    B::relax(t[e], l + 1, cx);
                }
                if ((l & 3) == 3) bc_sched_fence();
            }
#pragma unroll
            for (int e = 0; e < CPT; e++) c[j][e] = B::digit(t[e], cx);
            bc_sched_fence();
        }
        // FP64 limbs: the digits' halves for the split products of the output loop
        double c1[M][CPT], c0[M][CPT];
        if constexpr (std::is_same<B, BcF64>::value) {
#pragma unroll
            for (int l = 0; l < M; l++)
#pragma unroll
                for (int e = 0; e < CPT; e++) {
                    c1[l][e] = __builtin_rint(c[l][e] * 0x1p-25);
                    c0[l][e] = __builtin_fma(c1[l][e], -0x1p25, c[l][e]);
                }
        }
#pragma unroll 1
        for (int ob = 0; ob < cnt; ob += OU) {
            bc_no_hoist();
            T acc[OU][CPT];
            double S2[OU][CPT], S1[OU][CPT], S0[OU][CPT];          // (FP64 limbs)
            int oo[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) oo[u] = ob + u < cnt ? ob + u : cnt - 1;     // the block's tail recomputes the last output (no branch)
            Tw w[2][OU];
            decltype(B::ctx(0, Tw{})) cxu[OU];
#pragma unroll
            for (int u = 0; u < OU; u++) {
                w[0][u] = s_rec[oo[u] * REC];
                cxu[u] = B::ctx(s_rec[oo[u] * REC + M + 1].a, s_rec[oo[u] * REC + M]);
            }
#pragma unroll
            for (int l = 0; l < M; l++) {
                if (l + 1 < M) {
#pragma unroll
                    for (int u = 0; u < OU; u++) w[(l + 1) & 1][u] = s_rec[oo[u] * REC + l + 1];      // next limb's constants: in flight under this limb's products
                }
                if constexpr (std::is_same<B, BcF64>::value) {
                    // FP64 limbs: digit and constant are split into halves of 25 bits, c = c1 2^25 + c0, E = e1 2^25 + e0 (c0, e0 centred), so that
                    // every partial product is an exact integer below 2^50 and a term costs FOUR fused multiply-adds into three exact sums
                    //   S2 += c1 e1,  S1 += c1 e0 + c0 e1,  S0 += c0 e0      (x mod q = S2 2^50 + S1 2^25 + S0 mod q, formed once per output)
                    // instead of the seven instructions of a modular product and its accumulation.  Eight terms of at most 2^50 stay
                    // exact below 2^53; longer bases fold S2 and S1 modulo q before the ninth and the sixteenth term.
                    if (l == 8 || l == 15) {
#pragma unroll
                        for (int u = 0; u < OU; u++)
#pragma unroll
                            for (int e = 0; e < CPT; e++) {
                                ArithF64::reduce(S2[u][e], cxu[u]);
                                ArithF64::reduce(S1[u][e], cxu[u]);
                            }
                    }
#pragma unroll
                    for (int u = 0; u < OU; u++)
#pragma unroll
                        for (int e = 0; e < CPT; e++) S2[u][e] = l == 0 ? c1[l][e] * u64_bits_to_double(w[l & 1][u].a) : __builtin_fma(c1[l][e], u64_bits_to_double(w[l & 1][u].a), S2[u][e]);
#pragma unroll
                    for (int u = 0; u < OU; u++)
#pragma unroll
                        for (int e = 0; e < CPT; e++) S1[u][e] = l == 0 ? c1[l][e] * u64_bits_to_double(w[l & 1][u].b) : __builtin_fma(c1[l][e], u64_bits_to_double(w[l & 1][u].b), S1[u][e]);
#pragma unroll
                    for (int u = 0; u < OU; u++)
#pragma unroll
                        for (int e = 0; e < CPT; e++) S0[u][e] = l == 0 ? c0[l][e] * u64_bits_to_double(w[l & 1][u].b) : __builtin_fma(c0[l][e], u64_bits_to_double(w[l & 1][u].b), S0[u][e]);
#pragma unroll
                    for (int u = 0; u < OU; u++)
#pragma unroll
                        for (int e = 0; e < CPT; e++) S1[u][e] = __builtin_fma(c0[l][e], u64_bits_to_double(w[l & 1][u].a), S1[u][e]);
                } else {
#pragma unroll
                    for (int u = 0; u < OU; u++) {
#pragma unroll
                        for (int e = 0; e < CPT; e++) {
                            if (l == 0) acc[u][e] = B::mul(c[0][e], w[0][u], cxu[u]);
                            else {
                                acc[u][e] = B::add(acc[u][e], B::mul(c[l][e], w[l & 1][u], cxu[u]), cxu[u]);
                                B::relax(acc[u][e], l, cxu[u]);
                            }
                        }
                    }
                }
                bc_sched_fence();
            }
            // (unconditional stores: a block's tail writes its last output again -- same thread, same value -- instead of branching,
            // which would let the compiler sink each output's arithmetic into its own conditional block)
#pragma unroll
            for (int u = 0; u < OU; u++) {
                const int o = o0 + oo[u];
                u64 FHE_GLOBAL *dst = out + (u64)((u32)o < job.gap_at ? o : o + job.gap) * N + i;
                if constexpr (std::is_same<B, BcF64>::value) {
                    // S2 2^50 + S1 2^25 + S0 mod q: the two scaled sums are exact doubles (powers of two), each reduced by one quotient estimate
#pragma unroll
                    for (int e = 0; e < CPT; e++) {
                        const double A = S2[u][e] * 0x1p50, Bv = S1[u][e] * 0x1p25;
                        const double ka = __builtin_rint(A * cxu[u].ninv), kb = __builtin_rint(Bv * cxu[u].ninv);
                        double r = __builtin_fma(-ka, cxu[u].n, A) + __builtin_fma(-kb, cxu[u].n, Bv);
                        ArithF64::reduce(r, cxu[u]);
                        acc[u][e] = r + S0[u][e];
                    }
                }
                if constexpr (CPT == 2) *reinterpret_cast<u64x2 FHE_GLOBAL *>(dst) = u64x2{B::out(acc[u][0], cxu[u]), B::out(acc[u][1], cxu[u])};
                else *dst = B::out(acc[u][0], cxu[u]);
            }
        }
        if (i + stride < N) load_raw(i + stride);       // (grids cover N in one step except for N beyond 2^22)
    }
}

template <int M, class B, int CPT, int OU>
__global__ __launch_bounds__(256, 4) void k_bc_exact_fixed(const BcJob *jobs, BcJob one, u64 N, u32 oc)
{
    if (jobs) {
        const BcJob job = jobs[blockIdx.z];
        if (job.pl.m != M) return;              // a job list may mix digit sizes (the last digit of a key switch can be shorter): one launch per size
        bc_exact_fixed<M, B, CPT, OU>(job, N, oc);
    } else {
        bc_exact_fixed<M, B, CPT, OU>(one, N, oc);
    }
}

// one conversion, job passed by value
template <int MAXM, class B, bool STAGE>
__global__ __launch_bounds__(256) void k_baseconv_exact(BcJob job, u64 N, u32 oc)
{
    bc_exact_body<MAXM, B, STAGE>(job, N, oc);
}
// several conversions of the same shape class in one launch (key-switch digits, the two halves of a mod-down):
// blockIdx.z picks the job from a device-resident list
template <int MAXM, class B, bool STAGE>
__global__ __launch_bounds__(256) void k_baseconv_exact_jobs(const BcJob *jobs, u64 N, u32 oc)
{
    const BcJob job = jobs[blockIdx.z];
    bc_exact_body<MAXM, B, STAGE>(job, N, oc);
}

template <class B>
static void launch_exact(hipStream_t st, dim3 grid, const BcJob *dev_jobs, const BcJob &job, int maxm, u64 N, u32 oc)
{
#define FHE_BC(MM)                                                                                                    \
    do {                                                                                                              \
        const bool stage = MM <= 16 && (u64)grid.x * grid.y * grid.z <= 8192;                                          \
        if (dev_jobs && stage) hipLaunchKernelGGL((k_baseconv_exact_jobs<MM, B, true>), grid, dim3(256), 0, st, dev_jobs, N, oc);  \
        else if (dev_jobs) hipLaunchKernelGGL((k_baseconv_exact_jobs<MM, B, false>), grid, dim3(256), 0, st, dev_jobs, N, oc);     \
        else if (stage) hipLaunchKernelGGL((k_baseconv_exact<MM, B, true>), grid, dim3(256), 0, st, job, N, oc);        \
        else hipLaunchKernelGGL((k_baseconv_exact<MM, B, false>), grid, dim3(256), 0, st, job, N, oc);                  \
    } while (0)
    if (maxm <= 4) FHE_BC(4);
    else if (maxm <= 8) FHE_BC(8);
    else if (maxm <= 12) FHE_BC(12);
    else if (maxm <= 16) FHE_BC(16);
    else if (maxm <= 32) FHE_BC(32);
    else FHE_BC(64);
#undef FHE_BC
}

// aim at >= `target` workgroups: slice the k outputs over blockIdx.y while a slice stays at least as large as the
// digit computation it repeats (m/2 products per digit on average)
static u32 bc_slices(u32 gx, u32 jobs, int m, int k, u32 target = 2048)
{
    u32 slices = 1;
    while (gx * jobs * slices < target && slices * 2 <= (u32)k && (u32)k / (slices * 2) >= (u32)(m + 1) / 2) slices *= 2;
    return slices;
}

// tuning knobs of the fixed-size form (read once): FHE_BC_VARIANT = 0 selects the runtime-m kernels, FHE_BC_WGS = workgroups to aim at
static int bc_env(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <class B, int CPT, int OU>
static void launch_fixed_m(hipStream_t st, dim3 grid, const BcJob *dev_jobs, const BcJob &job, int m, u64 N, u32 oc)
{
    switch (m) {
#define FHE_BCM(MM) case MM: hipLaunchKernelGGL((k_bc_exact_fixed<MM, B, CPT, OU>), grid, dim3(256), 0, st, dev_jobs, job, N, oc); break;
        FHE_BCM(1) FHE_BCM(2) FHE_BCM(3) FHE_BCM(4) FHE_BCM(5) FHE_BCM(6) FHE_BCM(7) FHE_BCM(8)
        FHE_BCM(9) FHE_BCM(10) FHE_BCM(11) FHE_BCM(12) FHE_BCM(13) FHE_BCM(14) FHE_BCM(15) FHE_BCM(16)
#undef FHE_BCM
    default: break;
    }
}

// m_mask: bit (m - 1) set for every input size present among the jobs (one launch per size; a launch's workgroups of the other
// sizes leave at once).  One coefficient per thread; four outputs side by side where the digits leave room for it in 128 registers
// (two coefficients per thread measured 10 % slower at N = 2^16, L = 44: half the workgroups, or the digits computed twice).
template <class B>
static void launch_fixed(hipStream_t st, const BcJob *dev_jobs, const BcJob &job, u32 n_jobs, u32 m_mask, int max_k, u64 N)
{
    static const int target = bc_env("FHE_BC_WGS", 1024);
    for (int m = 1; m <= 16; m++) {
        if (!((m_mask >> (m - 1)) & 1)) continue;
        const u64 want = (N + 255) / 256;
        const u32 gx = (u32)(want > 16384 ? 16384 : want);
        const u32 slices = bc_slices(gx, n_jobs, m, max_k, (u32)target), oc = ((u32)max_k + slices - 1) / slices;
        const dim3 grid(gx, ((u32)max_k + oc - 1) / oc, n_jobs);
        static const int ou_env = bc_env("FHE_BC_OU", 0);
        const bool wide = ou_env ? ou_env == 4 : std::is_same<B, BcF64>::value ? m <= 11 : m <= 11;
        if (wide) launch_fixed_m<B, 1, 4>(st, grid, dev_jobs, job, m, N, oc);
        else launch_fixed_m<B, 1, 2>(st, grid, dev_jobs, job, m, N, oc);
    }
}
// FHE_BC_VARIANT=0: the runtime-m kernels everywhere (A/B runs)
static bool bc_fixed_on()
{
    static const int variant = bc_env("FHE_BC_VARIANT", 1);
    return variant != 0;
}

hipError_t launch_baseconv_exact(hipStream_t st, u64 *out, const u64 *in, const BaseConvPlanDev &pl, u64 N, u32 gap_at, u32 gap, const u32 *in_rows)
{
    if (pl.m > BC_MAX_LIMBS) return hipErrorInvalidValue;
    const BcJob job{pl, in, out, gap_at, gap, in_rows};
    if (pl.m <= 16 && bc_fixed_on()) {
        if (pl.f64) launch_fixed<BcF64>(st, nullptr, job, 1, 1u << (pl.m - 1), pl.k, N);
        else launch_fixed<BcU64>(st, nullptr, job, 1, 1u << (pl.m - 1), pl.k, N);
        return hipGetLastError();
    }
    u64 want = (N + 255) / 256;
    const u32 gx = (u32)(want > 16384 ? 16384 : want);
    const u32 slices = bc_slices(gx, 1, pl.m, pl.k), oc = ((u32)pl.k + slices - 1) / slices;
    const dim3 grid(gx, ((u32)pl.k + oc - 1) / oc);
    if (pl.f64) launch_exact<BcF64>(st, grid, nullptr, job, pl.m, N, oc);
    else launch_exact<BcU64>(st, grid, nullptr, job, pl.m, N, oc);
    return hipGetLastError();
}

// jobs: device array of n_jobs entries that share the arithmetic path (f64) ; max_m / max_k = largest m / k among them;
// m_mask: bit (m - 1) set for every input size m <= 16 present among the jobs (0 = unknown: the runtime-m kernels)
hipError_t launch_baseconv_exact_jobs(hipStream_t st, const BcJob *dev_jobs, u32 n_jobs, int max_m, int max_k, bool f64, u64 N, u32 m_mask)
{
    if (!n_jobs) return hipSuccess;
    if (max_m > BC_MAX_LIMBS || n_jobs > 65535) return hipErrorInvalidValue;
    const BcJob none{};
    if (m_mask && max_m <= 16 && bc_fixed_on()) {
        if (f64) launch_fixed<BcF64>(st, dev_jobs, none, n_jobs, m_mask, max_k, N);
        else launch_fixed<BcU64>(st, dev_jobs, none, n_jobs, m_mask, max_k, N);
        return hipGetLastError();
    }
    u64 want = (N + 255) / 256;
    const u32 gx = (u32)(want > 16384 ? 16384 : want);
    const u32 slices = bc_slices(gx, n_jobs, max_m, max_k), oc = ((u32)max_k + slices - 1) / slices;
    const dim3 grid(gx, ((u32)max_k + oc - 1) / oc, n_jobs);
    if (f64) launch_exact<BcF64>(st, grid, dev_jobs, none, max_m, N, oc);
    else launch_exact<BcU64>(st, grid, dev_jobs, none, max_m, N, oc);
    return hipGetLastError();
}

// rfhe_framewk/src/baseConv.py:10-40: out[o][i] = sum_j ((r_j * Phat_j * inv_j) mod q_o), sum NOT reduced
__global__ __launch_bounds__(256) void k_bconv_fast(u64 *out, const u64 *in, BaseConvPlanDev pl, u64 N)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        for (int o = 0; o < pl.k; o++) {
            const u64 q = pl.mod_out[o];
            u64 total = 0;
            for (int j = 0; j < pl.m; j++)
                total += mulmod_shoup(in[(u64)j * N + i], pl.fast_coef[j * pl.k + o], pl.fast_coef_shoup[j * pl.k + o], q);
            out[(u64)o * N + i] = total;
        }
    }
}

hipError_t launch_bconv_fast(hipStream_t st, u64 *out, const u64 *in, const BaseConvPlanDev &pl, u64 N)
{
    u64 want = (N + 255) / 256;
    hipLaunchKernelGGL(k_bconv_fast, dim3((u32)(want > 4096 ? 4096 : want)), dim3(256), 0, st, out, in, pl, N);
    return hipGetLastError();
}

// rfhe_framewk/src/baseConv.cu:85-120 (crt_kernel) re-expressed for gfx950: same
// Garner recurrence and the same 128-bit wrap-around of c_k * pref_k and of the sum;
// the 128 % 64 steps use the Barrett ratio instead of a software division.
constexpr int GARNER_MAX = 16;
__global__ __launch_bounds__(256) void k_crt_garner(u64 *x_lo, u64 *x_hi, const u64 *res, const u64 *mod, const u64 *ratio,
                                                    const u64 *pref_lo, const u64 *pref_hi, const u64 *inv_pref, int m, u64 N)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        u64 c[GARNER_MAX];
        c[0] = res[i];
        for (int j = 1; j < m; j++) {
            const u64 pj = mod[j], r0 = ratio[2 * j], r1 = ratio[2 * j + 1];
            u64 t = barrett128(res[(u64)j * N + i], 0, pj, r0, r1);
            for (int k = 0; k < j; k++) {
                // prod = c_k * pref_k wrapped to 128 bits
                const u64 lo = c[k] * pref_lo[k];
                const u64 hi = __umul64hi(c[k], pref_lo[k]) + c[k] * pref_hi[k];
                const u64 r = barrett128(lo, hi, pj, r0, r1);
                t = t >= r ? t - r : t + pj - r;
            }
            c[j] = mulmod_b(t, inv_pref[j], pj, r0, r1);
        }
        u64 lo = 0, hi = 0;
        for (int k = 0; k < m; k++) {
            const u64 pl = c[k] * pref_lo[k];
            const u64 ph = __umul64hi(c[k], pref_lo[k]) + c[k] * pref_hi[k];
            const u64 nl = lo + pl;
            hi += ph + (nl < lo);
            lo = nl;
        }
        x_lo[i] = lo;
        x_hi[i] = hi;
    }
}

hipError_t launch_crt_garner(hipStream_t st, u64 *x_lo, u64 *x_hi, const u64 *residues, const u64 *moduli, const u64 *ratios,
                             const u64 *pref_lo, const u64 *pref_hi, const u64 *inv_pref, int m, u64 N)
{
    if (m < 1 || m > GARNER_MAX) return hipErrorInvalidValue;
    u64 want = (N + 255) / 256;
    hipLaunchKernelGGL(k_crt_garner, dim3((u32)(want > 4096 ? 4096 : want)), dim3(256), 0, st, x_lo, x_hi, residues, moduli,
                       ratios, pref_lo, pref_hi, inv_pref, m, N);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Key-switch helpers (shape of profile_framewk/build/data/ckks/16384_4:466-539)
// ---------------------------------------------------------------------------
// out = (a - b) * s_l mod q_l per limb: the mod-down tail (subtract the converted special-prime part,
// multiply by P^-1)
// Last step of the key-switch mod-down for both halves in one launch (blockIdx.y = half):
//   out_h = (a_h - b_h) * scal[l] (+ add_h) mod q_l,   a_h = a + h * a_stride, b_h = b + h * b_stride
// `add0` / `add1` let a rotation fold its sigma(c0) term (a relinearisation: d0 and d1) in instead of a separate add and copy.
__global__ __launch_bounds__(256) void k_sub_scale(SubScaleArgs p)
{
    const u32 h = blockIdx.y;
    u64 *out = h ? p.out1 : p.out0;
    const u64 *a = p.a ? p.a + (u64)h * p.a_stride : nullptr, *b = p.b ? p.b + (u64)h * p.b_stride : nullptr, *add = h ? p.add1 : p.add0;
    const u64 total = (u64)p.limbs << p.logn;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u32 l = (u32)(i >> p.logn);
        const LimbParams &lp = p.lp[p.limb0 + l];
        const u64 q = lp.q, r0 = lp.barrett_lo, r1 = lp.barrett_hi;
        const u64 x = a ? barrett128(a[i], 0, q, r0, r1) : 0, y = b ? barrett128(b[i], 0, q, r0, r1) : 0;     // (a or b absent: zero)
        const u64 d = x >= y ? x - y : x + q - y;
        u64 v = mulmod_b(d, p.scal[l], q, r0, r1);
        if (add) {
            v += barrett128(add[i], 0, q, r0, r1);
            v = v >= q ? v - q : v;
        }
        out[i] = v;
    }
}

hipError_t launch_sub_scale(hipStream_t st, const SubScaleArgs &p)
{
    const u64 total = (u64)p.limbs << p.logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_sub_scale, dim3((u32)(want > 8192 ? 8192 : want), p.out1 ? 2 : 1), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Key-switch inner product (MULTEVK): acc_h[j] = sum_d ext_d[j] * evk[d][h][j] mod q_j for both key halves in one
// pass: every extended digit is read once, nothing is read-modify-written.  ext_d[j] is the digit's own
// NTT-form input limb when j belongs to digit d, the base-extended + transformed limb otherwise.
// ---------------------------------------------------------------------------
struct KsMacF64 {
    typedef double acc_t;
    static __device__ __forceinline__ double zero() { return 0.0; }
    static __device__ __forceinline__ void mac(double &s, u64 x, u64 y, int term, const LimbParams &p)
    {
        const ArithF64::Ctx c = ArithF64::make_ctx(p);
        const double a = ArithF64::from_canonical(x < p.q ? x : reduce_any_u64(x, p.q));
        const double b = ArithF64::from_canonical(y < p.q ? y : reduce_any_u64(y, p.q));
        const double h = a * b;
        const double k = __builtin_rint(a * (b * c.ninv));
        const double l = __builtin_fma(a, b, -h);
        s += __builtin_fma(-k, c.n, h) + l;          // |term| < 0.875 q
        if ((term & 7) == 7) ArithF64::reduce(s, c);
    }
    static __device__ __forceinline__ u64 out(double s, const LimbParams &p) { return ArithF64::canonical(s, ArithF64::make_ctx(p)); }
};
struct KsMacU64 {
    struct acc_t {
        u64 lo, hi;
    };
    static __device__ __forceinline__ acc_t zero() { return acc_t{0, 0}; }
    static __device__ __forceinline__ void mac(acc_t &s, u64 x, u64 y, int term, const LimbParams &p)
    {
        x = x < p.q ? x : reduce_any_u64(x, p.q);
        y = y < p.q ? y : reduce_any_u64(y, p.q);
        const u64 lo = x * y, hi = __umul64hi(x, y);      // < 2^124
        s.lo += lo;
        s.hi += hi + (s.lo < lo);
        if ((term & 7) == 7) s = acc_t{barrett128(s.lo, s.hi, p.q, p.barrett_lo, p.barrett_hi), 0};
    }
    static __device__ __forceinline__ u64 out(const acc_t &s, const LimbParams &p) { return barrett128(s.lo, s.hi, p.q, p.barrett_lo, p.barrett_hi); }
};

template <class K>
__device__ __forceinline__ void ks_mac_limb(const KsMacArgs &a, u32 j, u64 i, const LimbParams &p)
{
    const u64 N = (u64)1 << a.logn;
    typename K::acc_t s0 = K::zero(), s1 = K::zero();
    const u32 tl = j < a.cn ? a.clo + j : 0xFFFFFFFFu;     // table limb when the row is a ciphertext limb
    for (u32 d = 0; d < a.dnum; d++) {
        const u32 lo = d * a.alpha, hi = lo + a.alpha < a.L ? lo + a.alpha : a.L;
        const u64 x = (tl >= lo && tl < hi) ? a.c[(u64)j * N + i] : a.ext[((u64)d * a.M + j) * N + i];
        const u64 *key = a.evk + ((u64)d * 2 * a.M + j) * N + i;
        K::mac(s0, x, key[0], (int)d, p);
        K::mac(s1, x, key[(u64)a.M * N], (int)d, p);
    }
    a.acc[(u64)j * N + i] = K::out(s0, p);
    a.acc[((u64)a.M + j) * N + i] = K::out(s1, p);
}

__global__ __launch_bounds__(256) void k_ks_mac(KsMacArgs a)
{
    const u64 total = (u64)a.M << a.logn;
    for (u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 j = (u32)(e >> a.logn);
        const u64 i = e & (((u64)1 << a.logn) - 1);
        const LimbParams &p = a.lp[j < a.cn ? a.clo + j : j + a.sp_shift];
        if (p.path == PATH_F64) ks_mac_limb<KsMacF64>(a, j, i, p);
        else ks_mac_limb<KsMacU64>(a, j, i, p);
    }
}

// The same inner product for up to four keys at once on SHARED digits (hoisted rotations, fhe_rotate_hoisted): every extended digit is
// read once for the group instead of once per key -- 220 + n (440 + 110) limb sweeps instead of n (220 + 440 + 110) at config 5.
template <class K, int R>
__device__ __forceinline__ void ks_mac_limb_multi(const KsMacMultiArgs &m, u32 j, u64 i, const LimbParams &p)
{
    const KsMacArgs &a = m.a;
    const u64 N = (u64)1 << a.logn;
    typename K::acc_t s0[R], s1[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        s0[r] = K::zero();
        s1[r] = K::zero();
    }
    const u32 tl = j < a.cn ? a.clo + j : 0xFFFFFFFFu;
    for (u32 d = 0; d < a.dnum; d++) {
        const u32 lo = d * a.alpha, hi = lo + a.alpha < a.L ? lo + a.alpha : a.L;
        const u64 x = (tl >= lo && tl < hi) ? a.c[(u64)j * N + i] : a.ext[((u64)d * a.M + j) * N + i];
        const u64 off = ((u64)d * 2 * a.M + j) * N + i;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u64 *key = m.evk[r] + off;
            K::mac(s0[r], x, key[0], (int)d, p);
            K::mac(s1[r], x, key[(u64)a.M * N], (int)d, p);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        m.acc[r][(u64)j * N + i] = K::out(s0[r], p);
        m.acc[r][((u64)a.M + j) * N + i] = K::out(s1[r], p);
    }
}

template <int R>
__global__ __launch_bounds__(256) void k_ks_mac_multi(KsMacMultiArgs m)
{
    const KsMacArgs &a = m.a;
    const u64 total = (u64)a.M << a.logn;
    for (u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const u32 j = (u32)(e >> a.logn);
        const u64 i = e & (((u64)1 << a.logn) - 1);
        const LimbParams &p = a.lp[j < a.cn ? a.clo + j : j + a.sp_shift];
        if (p.path == PATH_F64) ks_mac_limb_multi<KsMacF64, R>(m, j, i, p);
        else ks_mac_limb_multi<KsMacU64, R>(m, j, i, p);
    }
}

hipError_t launch_ks_mac_multi(hipStream_t st, const KsMacMultiArgs &m)
{
    const u64 total = (u64)m.a.M << m.a.logn;
    if (!total || !m.n) return hipSuccess;
    const u64 want = (total + 255) / 256;
    const dim3 grid((u32)(want > 16384 ? 16384 : want));
    switch (m.n) {
    case 1: hipLaunchKernelGGL(k_ks_mac_multi<1>, grid, dim3(256), 0, st, m); break;
    case 2: hipLaunchKernelGGL(k_ks_mac_multi<2>, grid, dim3(256), 0, st, m); break;
    case 3: hipLaunchKernelGGL(k_ks_mac_multi<3>, grid, dim3(256), 0, st, m); break;
    case 4: hipLaunchKernelGGL(k_ks_mac_multi<4>, grid, dim3(256), 0, st, m); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_ks_mac(hipStream_t st, const KsMacArgs &a)
{
    const u64 total = (u64)a.M << a.logn;
    if (!total) return hipSuccess;
    const u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_ks_mac, dim3((u32)(want > 16384 ? 16384 : want)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Tensor product of two two-part ciphertexts (phantom::multiply, reliability_test/dotprod_test.cu:113), NTT domain:
// d0 = a0 b0, d1 = a0 b1 + a1 b0, d2 = a1 b1 per limb.  Four reads and three writes per coefficient (56 N bytes per
// limb) instead of the 88 N of four separate products; the cross term is summed lazily and reduced once.
// ---------------------------------------------------------------------------
template <class K>
__device__ __forceinline__ void tensor_elem(const TensorArgs &t, u64 i, const LimbParams &p)
{
    const u64 a0 = t.a0[i], a1 = t.a1[i], b0 = t.b0[i], b1 = t.b1[i];
    typename K::acc_t s0 = K::zero(), s1 = K::zero(), s2 = K::zero();
    K::mac(s0, a0, b0, 0, p);
    K::mac(s1, a0, b1, 0, p);
    K::mac(s1, a1, b0, 1, p);
    K::mac(s2, a1, b1, 0, p);
    t.d0[i] = K::out(s0, p);
    t.d1[i] = K::out(s1, p);
    t.d2[i] = K::out(s2, p);
}

__global__ __launch_bounds__(256) void k_tensor(TensorArgs t)
{
    const u64 total = (u64)t.limbs << t.logn;
    for (u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const LimbParams &p = t.lp[t.limb0 + (u32)(e >> t.logn)];
        if (p.path == PATH_F64) tensor_elem<KsMacF64>(t, e, p);
        else tensor_elem<KsMacU64>(t, e, p);
    }
}

hipError_t launch_tensor(hipStream_t st, const TensorArgs &p)
{
    const u64 total = (u64)p.limbs << p.logn;
    if (!total) return hipSuccess;
    const u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_tensor, dim3((u32)(want > 16384 ? 16384 : want)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Inner sum of a baby-step / giant-step matrix-vector product (profile_framewk/src/matmul_ckks.cpp:45-113, multiply_plain + add over
// the baby steps; motivation/bsgs.py:44-51 is the same accumulate on plaintext blocks): for both parts h of the rotated ciphertexts
//   out_h[l][i] = sum_{b < n1} diag[b][l][i] * R_b,h[l][i]  mod q_l,      R_0 = (x0, x1), R_b = rot[b-1] for b >= 1
// one pass: every diagonal and every rotated part is read once, the sums stay in registers (lazy, reduced every eighth term).
// ---------------------------------------------------------------------------
template <class K>
__device__ __forceinline__ void diag_mac_elem(const DiagMacArgs &a, u64 e, const LimbParams &p)
{
    const u64 part = (u64)a.limbs << a.logn;
    typename K::acc_t s0 = K::zero(), s1 = K::zero();
    for (u32 b = 0; b < a.n1; b++) {
        const u64 d = a.diag[(u64)b * part + e];
        const u64 *r = b ? a.rot + (u64)(b - 1) * 2 * part : nullptr;
        const u64 y0 = b ? r[e] : a.x0[e], y1 = b ? r[part + e] : a.x1[e];
        K::mac(s0, d, y0, (int)b, p);
        K::mac(s1, d, y1, (int)b, p);
    }
    a.out0[e] = K::out(s0, p);
    a.out1[e] = K::out(s1, p);
}

__global__ __launch_bounds__(256) void k_diag_mac(DiagMacArgs a)
{
    const u64 total = (u64)a.limbs << a.logn;
    for (u64 e = blockIdx.x * (u64)blockDim.x + threadIdx.x; e < total; e += (u64)gridDim.x * blockDim.x) {
        const LimbParams &p = a.lp[a.limb0 + (u32)(e >> a.logn)];
        if (p.path == PATH_F64) diag_mac_elem<KsMacF64>(a, e, p);
        else diag_mac_elem<KsMacU64>(a, e, p);
    }
}

hipError_t launch_diag_mac(hipStream_t st, const DiagMacArgs &a)
{
    const u64 total = (u64)a.limbs << a.logn;
    if (!total || !a.n1) return hipSuccess;
    const u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_diag_mac, dim3((u32)(want > 16384 ? 16384 : want)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ABFT: weighted checksum sum_i w_i x_i mod q (rfhe_framewk/src/negaclic_ntt.py:143-144), one
// workgroup per limb-polynomial, lanes stride through the limb (coalesced), LDS tree reduction.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_weighted_checksum(u64 *out, const u64 *x, const u64 *w, const u64 *scal,
                                                           const LimbParams *lp, u32 limb0, u32 limbs, u32 poly_stride, int logn)
{
    __shared__ u64 part[256];
    const u32 unit = blockIdx.x, poly = unit / limbs, l = unit % limbs;
    const LimbParams &p = lp[limb0 + l];
    const u64 q = p.q, r0 = p.barrett_lo, r1 = p.barrett_hi;
    const u64 *xs = x + (((u64)poly * poly_stride + l) << logn), *ws = w + ((u64)(limb0 + l) << logn);
    u64 acc = 0;
    for (u32 i = threadIdx.x; i < (1u << logn); i += 256) {
        acc += mulmod_b(xs[i], ws[i], q, r0, r1);
        acc = acc >= q ? acc - q : acc;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            u64 v = part[threadIdx.x] + part[threadIdx.x + s];
            part[threadIdx.x] = v >= q ? v - q : v;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[unit] = scal ? mulmod_b(part[0], scal[limb0 + l], q, r0, r1) : part[0];
}

hipError_t launch_weighted_checksum(hipStream_t st, u64 *out, const u64 *x, const u64 *w, const u64 *scal, const LimbParams *lp,
                                    u32 limb0, u32 limbs, u32 units, u32 poly_stride, int logn)
{
    if (!units) return hipSuccess;
    hipLaunchKernelGGL(k_weighted_checksum, dim3(units), dim3(256), 0, st, out, x, w, scal, lp, limb0, limbs, poly_stride, logn);
    return hipGetLastError();
}

__global__ void k_compare_flags(u32 *flags, const u64 *a, const u64 *b, u32 units)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < units) flags[i] = a[i] != b[i];
}

// flags[unit] = (sum of a[unit][*] mod q) != (sum of b[unit][*] mod q): per-tile partial sums of the fused checksums
__global__ void k_compare_sums(u32 *flags, const u64 *a, u32 ta, const u64 *b, u32 tb, const LimbParams *lp, u32 limb0, u32 limbs, u32 units)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < units) {
        const u64 q = lp[limb0 + i % limbs].q;
        u64 sa = 0, sb = 0;
        for (u32 t = 0; t < ta; t++) {
            sa += a[(u64)i * ta + t];      // partials are canonical (< q < 2^62)
            sa = sa >= q ? sa - q : sa;
        }
        for (u32 t = 0; t < tb; t++) {
            sb += b[(u64)i * tb + t];
            sb = sb >= q ? sb - q : sb;
        }
        flags[i] = sa != sb;
    }
}

hipError_t launch_compare_sums(hipStream_t st, u32 *flags, const u64 *a, u32 ta, const u64 *b, u32 tb, const LimbParams *lp, u32 limb0,
                               u32 limbs, u32 units)
{
    if (!units) return hipSuccess;
    hipLaunchKernelGGL(k_compare_sums, dim3((units + 255) / 256), dim3(256), 0, st, flags, a, ta, b, tb, lp, limb0, limbs, units);
    return hipGetLastError();
}

__global__ void k_compare_phases(u32 *flags, const u64 *s_in, const u64 *s_mid1, u32 tc, const u64 *s_mid2, const u64 *s_out, u32 tr, const LimbParams *lp,
                                 u32 limb0, u32 limbs, u32 units)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= units) return;
    const u64 q = lp[limb0 + i % limbs].q;
    auto total = [&](const u64 *v, u32 n) {
        u64 s = 0;
        for (u32 t = 0; t < n; t++) {
            s += v[(u64)i * n + t];
            s = s >= q ? s - q : s;
        }
        return s;
    };
    const u64 a = total(s_in, tc), b = total(s_mid1, tc), c = total(s_mid2, tr), d = total(s_out, tr);
    flags[3 * i] = a != b;
    flags[3 * i + 1] = b != c;
    flags[3 * i + 2] = c != d;
}

hipError_t launch_compare_phases(hipStream_t st, u32 *flags, const u64 *s_in, const u64 *s_mid1, u32 tc, const u64 *s_mid2, const u64 *s_out, u32 tr,
                                 const LimbParams *lp, u32 limb0, u32 limbs, u32 units)
{
    if (!units) return hipSuccess;
    hipLaunchKernelGGL(k_compare_phases, dim3((units + 255) / 256), dim3(256), 0, st, flags, s_in, s_mid1, tc, s_mid2, s_out, tr, lp, limb0, limbs, units);
    return hipGetLastError();
}

hipError_t launch_compare_flags(hipStream_t st, u32 *flags, const u64 *a, const u64 *b, u32 units)
{
    if (!units) return hipSuccess;
    hipLaunchKernelGGL(k_compare_flags, dim3((units + 255) / 256), dim3(256), 0, st, flags, a, b, units);
    return hipGetLastError();
}

// Galois automorphism x -> x^k (k odd) on coefficient-domain limbs: dst[(i k) mod N] = +-src[i]
// (the index map behind phantom::rotate_inplace, reliability_test/dotprod_test.cu:146)
__global__ __launch_bounds__(256) void k_automorphism(u64 *dst, const u64 *src, const LimbParams *lp, u32 limb0, u32 limbs,
                                                      u32 units, int logn, u32 k)
{
    const u64 total = (u64)units << logn;
    const u32 n = 1u << logn, mask2 = 2 * n - 1;
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u32 unit = (u32)(g >> logn), i = (u32)g & (n - 1);
        const LimbParams &p = lp[limb0 + unit % limbs];
        const u64 q = p.q;
        const u32 j = (u32)(((u64)i * k) & mask2);
        u64 v = src[g];
        if (v >= q) v = barrett128(v, 0, q, p.barrett_lo, p.barrett_hi);      // out-of-range words only: no 64-bit division on the hot path
        if (j >= n) v = v ? q - v : 0;
        dst[((u64)unit << logn) + (j & (n - 1))] = v;
    }
}

hipError_t launch_automorphism(hipStream_t st, u64 *dst, const u64 *src, const LimbParams *lp, u32 limb0, u32 limbs, u32 units,
                               int logn, u32 k)
{
    const u64 total = (u64)units << logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_automorphism, dim3((u32)(want > 8192 ? 8192 : want)), dim3(256), 0, st, dst, src, lp, limb0, limbs, units,
                       logn, k);
    return hipGetLastError();
}

// The same map on NTT-domain limbs (bit-reversed order): slot j evaluates at psi^(2 bitrev(j) + 1), and
// (sigma_k f)(x) = f(x^k), so dst[j] = src[j'] with 2 bitrev(j') + 1 = (2 bitrev(j) + 1) k mod 2N.
__global__ __launch_bounds__(256) void k_automorphism_ntt(u64 *dst, const u64 *src, u64 *dst_b, const u64 *src_b, u32 units, int logn, u32 k)
{
    if (blockIdx.y) {          // second (source, destination) pair of the same shape: both parts of a ciphertext in one launch
        dst = dst_b;
        src = src_b;
    }
    const u64 total = (u64)units << logn;
    const u32 n = 1u << logn, mask2 = 2 * n - 1;
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < total; g += (u64)gridDim.x * blockDim.x) {
        const u32 j = (u32)g & (n - 1);
        const u32 e = 2 * (__brev(j) >> (32 - logn)) + 1;
        const u32 e2 = (u32)(((u64)e * k) & mask2);
        const u32 j2 = __brev((e2 - 1) >> 1) >> (32 - logn);
        dst[g] = src[(g & ~(u64)(n - 1)) | j2];
    }
}

hipError_t launch_automorphism_ntt(hipStream_t st, u64 *dst, const u64 *src, u32 units, int logn, u32 k, u64 *dst_b, const u64 *src_b)
{
    const u64 total = (u64)units << logn;
    if (!total) return hipSuccess;
    u64 want = (total + 255) / 256;
    hipLaunchKernelGGL(k_automorphism_ntt, dim3((u32)(want > 8192 ? 8192 : want), dst_b ? 2 : 1), dim3(256), 0, st, dst, src, dst_b, src_b, units,
                       logn, k);
    return hipGetLastError();
}

// motivation/bsgs.py:39-52: y_i = sum_j M[(j - i) mod k] (.) v_j.  lp == nullptr keeps
// the reference's int64 wrap-around (NumPy, no reduction); otherwise mod q.
__global__ __launch_bounds__(256) void k_bsgs_hadamard(u64 *y, const u64 *M, const u64 *v, int k, int bs, ModConst mc, bool lp)
{
    const u64 total = (u64)k * bs;
    for (u64 o = blockIdx.x * (u64)blockDim.x + threadIdx.x; o < total; o += (u64)gridDim.x * blockDim.x) {
        const int i = (int)(o / bs), e = (int)(o % bs);
        u64 acc = 0;
        if (lp) {
            const u64 q = mc.q, r0 = mc.r0, r1 = mc.r1;
            for (int j = 0; j < k; j++) {
                int b = j - i;
                b = b < 0 ? b + k : b;
                acc += mulmod_b(M[(u64)b * bs + e], v[(u64)j * bs + e], q, r0, r1);
                acc = acc >= q ? acc - q : acc;
            }
        } else {
            for (int j = 0; j < k; j++) {
                int b = j - i;
                b = b < 0 ? b + k : b;
                acc += M[(u64)b * bs + e] * v[(u64)j * bs + e];
            }
        }
        y[o] = acc;
    }
}

hipError_t launch_bsgs_hadamard(hipStream_t st, u64 *y, const u64 *M_blocks, const u64 *v, int k, int bs, const ModConst *mc)
{
    u64 want = ((u64)k * bs + 255) / 256;
    hipLaunchKernelGGL(k_bsgs_hadamard, dim3((u32)(want > 4096 ? 4096 : want)), dim3(256), 0, st, y, M_blocks, v, k, bs,
                       mc ? *mc : ModConst{1, 0, 0}, mc != nullptr);
    return hipGetLastError();
}

} // namespace fhe
