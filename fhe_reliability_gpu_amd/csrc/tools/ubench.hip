// ubench.hip -- instruction-rate and memory-bandwidth probes for gfx950 that the
// NTT kernel design depends on (DESIGN.md "ALU ceiling").  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 ubench.hip -o ubench && ./ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef uint64_t u64;
typedef uint32_t u32;

constexpr int ITER = 4096;
constexpr int ILP = 8;

enum Op { MUL_LO, MUL_HI, MAD64, ADD64, MUL64LO, MULHI64, FMA64, MULF64, ADDF64, RNDNE, FMAF32, CNDMASK64, SHOUP, FPMULMOD, CVT_U64_F64, DPP64 };

template <int OP>
__global__ __launch_bounds__(256) void k_alu(u64 *out, u64 seed, double dseed)
{
    u64 x[ILP];
    double d[ILP];
    for (int i = 0; i < ILP; i++) { x[i] = seed + threadIdx.x * 977 + i * 131 + blockIdx.x; d[i] = dseed + threadIdx.x + i; }
    const u64 q = 1125899903107073ull, w = seed | 12345, wp = seed * 3 + 7;
    const double n = 1125899903107073.0, ninv = 1.0 / n, dw = dseed * 1e15, dwp = dw * ninv;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if constexpr (OP == MUL_LO) { u32 a = (u32)x[i]; a = a * (u32)w + 1; x[i] = a; }
            else if constexpr (OP == MUL_HI) { u32 a = (u32)x[i]; a = __umulhi(a, (u32)w) + 3; x[i] = a; }
            else if constexpr (OP == MAD64) { x[i] = (u64)(u32)x[i] * (u32)w + x[i]; }
            else if constexpr (OP == ADD64) { x[i] = x[i] + w; }
            else if constexpr (OP == MUL64LO) { x[i] = x[i] * w + 1; }
            else if constexpr (OP == MULHI64) { x[i] = __umul64hi(x[i], w) + 1; }
            else if constexpr (OP == FMA64) { d[i] = __builtin_fma(d[i], dw, dwp); }
            else if constexpr (OP == MULF64) { d[i] = d[i] * dwp; }
            else if constexpr (OP == ADDF64) { d[i] = d[i] + dwp; }
            else if constexpr (OP == RNDNE) { d[i] = __builtin_rint(d[i]) + 0.0; }
            else if constexpr (OP == FMAF32) { float f = __builtin_bit_cast(float, (u32)x[i]); f = __builtin_fmaf(f, 1.0001f, 0.5f); x[i] = __builtin_bit_cast(u32, f); }
            else if constexpr (OP == CNDMASK64) { x[i] = (x[i] >= q) ? x[i] - q : x[i] + w; }
            else if constexpr (OP == SHOUP) {
                u64 qh = __umul64hi(x[i], wp);
                u64 r = x[i] * w - qh * q;
                x[i] = r >= q ? r - q : r;
            } else if constexpr (OP == FPMULMOD) {
                double a = d[i];
                double h = a * dw;
                double qq = __builtin_rint(a * dwp);
                double l = __builtin_fma(a, dw, -h);
                double r = __builtin_fma(-qq, n, h);
                d[i] = r + l;
            } else if constexpr (OP == CVT_U64_F64) {
                // magic-number u64(<2^52) <-> f64 round trip
                double t = __builtin_bit_cast(double, x[i] | 0x4330000000000000ull) - 4503599627370496.0;
                t = t + 4503599627370497.0;
                x[i] = __builtin_bit_cast(u64, t) & 0xFFFFFFFFFFFFFull;
            } else if constexpr (OP == DPP64) {
                u32 lo = (u32)x[i], hi = (u32)(x[i] >> 32);
                lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); // quad_perm [1,0,3,2]
                hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false);
                x[i] = ((u64)hi << 32 | lo) + 1;
            }
        }
    }
    u64 acc = 0;
    for (int i = 0; i < ILP; i++) acc += x[i] + __builtin_bit_cast(u64, d[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// ---- memory probes ---------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256) void k_copy(V *__restrict__ dst, const V *__restrict__ src, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}
template <typename V>
__global__ __launch_bounds__(256) void k_rmw(V *__restrict__ buf, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { V v = buf[i]; v.x += 1; buf[i] = v; }
}
// tile-local RMW: each block owns a contiguous 64 KiB tile (like an NTT pass)
__global__ __launch_bounds__(256) void k_rmw_tile(ulonglong2 *__restrict__ buf, size_t tiles)
{
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        ulonglong2 *p = buf + t * 4096;
        ulonglong2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = p[k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 16; k++) { v[k].x += 1; p[k * 256 + threadIdx.x] = v[k]; }
    }
}
// strided-column RMW: block owns 16 columns (128 B) x 256 rows at stride 2 KiB inside a 512 KiB limb
__global__ __launch_bounds__(256) void k_rmw_cols(u64 *__restrict__ buf, size_t limbs)
{
    size_t ntile = limbs * 16;
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        u64 *p = buf + (t / 16) * 65536 + (t % 16) * 16;
        int c = threadIdx.x & 15, r0 = threadIdx.x >> 4;
        u64 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = p[(size_t)(r0 + 16 * k) * 256 + c];
#pragma unroll
        for (int k = 0; k < 16; k++) p[(size_t)(r0 + 16 * k) * 256 + c] = v[k] + 1;
    }
}
__global__ __launch_bounds__(256) void k_rmw_cols32(u64 *__restrict__ buf, size_t limbs)
{
    // 32 columns (256 B) x 256 rows, 512 threads
    size_t ntile = limbs * 8;
    for (size_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        u64 *p = buf + (t / 8) * 65536 + (t % 8) * 32;
        int c = threadIdx.x & 31, r0 = threadIdx.x >> 5; // 256 threads: r0 in 0..7, 32 elems each
        u64 v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = p[(size_t)(r0 + 8 * k) * 256 + c];
#pragma unroll
        for (int k = 0; k < 32; k++) p[(size_t)(r0 + 8 * k) * 256 + c] = v[k] + 1;
    }
}

template <typename F>
static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int OP>
static void run_alu(const char *name, u64 *out, double ops_per_iter)
{
    int blocks = 256 * 8;
    float ms = time_ms([&] { k_alu<OP><<<blocks, 256>>>(out, 0x9e3779b97f4a7c15ull, 1.25); }, 5);
    double total = (double)blocks * 256 * ITER * ILP * ops_per_iter;
    double per_s = total / (ms * 1e-3);
    // cycles per wave64-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    double cyc = 1024.0 * 2.4e9 / (per_s / 64.0);
    printf("%-12s %8.3f ms  %9.2f Gop/s   %6.2f cyc/wave-op/SIMD @2.4GHz (per listed op, %g per iter)\n", name, ms, per_s * 1e-9, cyc, ops_per_iter);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    u64 *out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    run_alu<FMAF32>("fma_f32", out, 1);
    run_alu<MUL_LO>("mul_lo_u32", out, 1);
    run_alu<MUL_HI>("mul_hi_u32", out, 1);
    run_alu<MAD64>("mad_u64_u32", out, 1);
    run_alu<ADD64>("add_u64", out, 1);
    run_alu<MUL64LO>("mul64_lo", out, 1);
    run_alu<MULHI64>("mulhi64", out, 1);
    run_alu<CNDMASK64>("condsub64", out, 1);
    run_alu<FMA64>("fma_f64", out, 1);
    run_alu<MULF64>("mul_f64", out, 1);
    run_alu<ADDF64>("add_f64", out, 1);
    run_alu<RNDNE>("rndne+add", out, 1);
    run_alu<CVT_U64_F64>("cvt_rt", out, 1);
    run_alu<DPP64>("dpp64+add", out, 1);
    run_alu<SHOUP>("shoup_mulmod", out, 1);
    run_alu<FPMULMOD>("fp_mulmod", out, 1);

    // memory
    for (size_t mib : {16, 64, 2048}) {
        size_t bytes = mib << 20;
        char *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        int reps = mib >= 1024 ? 5 : 50;
        float ms;
        ms = time_ms([&] { k_copy<ulonglong2><<<2048, 256>>>((ulonglong2 *)b, (ulonglong2 *)a, bytes / 16); }, reps);
        printf("copy16B   %5zu MiB: %8.3f ms  %7.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        ms = time_ms([&] { k_copy<ulonglong1><<<2048, 256>>>((ulonglong1 *)b, (ulonglong1 *)a, bytes / 8); }, reps);
        printf("copy8B    %5zu MiB: %8.3f ms  %7.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        ms = time_ms([&] { k_rmw<ulonglong2><<<2048, 256>>>((ulonglong2 *)a, bytes / 16); }, reps);
        printf("rmw16B    %5zu MiB: %8.3f ms  %7.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        ms = time_ms([&] { k_rmw_tile<<<2048, 256>>>((ulonglong2 *)a, bytes / 65536); }, reps);
        printf("rmw_tile  %5zu MiB: %8.3f ms  %7.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        ms = time_ms([&] { k_rmw_cols<<<2048, 256>>>((u64 *)a, bytes / 524288); }, reps);
        printf("rmw_cols16 %4zu MiB: %8.3f ms  %7.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        ms = time_ms([&] { k_rmw_cols32<<<2048, 256>>>((u64 *)a, bytes / 524288); }, reps);
        printf("rmw_cols32 %4zu MiB: %8.3f ms  %7.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        // two dependent passes over the same buffer (tile then cols): does pass 2 hit MALL/L2?
        ms = time_ms([&] { k_rmw_cols<<<2048, 256>>>((u64 *)a, bytes / 524288); k_rmw_tile<<<2048, 256>>>((ulonglong2 *)a, bytes / 65536); }, reps);
        printf("cols+tile %5zu MiB: %8.3f ms  %7.1f GB/s (algorithmic r+w once)\n", mib, ms, 2.0 * bytes / ms * 1e-6);
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
