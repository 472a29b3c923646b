// Measurement tool: issue rate of the integer vector instructions the ArithU64 (61-bit) butterflies are made of (gfx950), next to v_fma_f64.
// hipcc -O3 --offload-arch=gfx950 int_rates.hip -o int_rates && ./int_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef uint32_t u32;

template <int OP>
__global__ __launch_bounds__(256) void k(u64 *out, u64 a, u64 b, int iters)
{
    u64 x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 977 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) x[i] = (u64)((u32)x[i] * (u32)b) | (x[i] & 0xFFFFFFFF00000000ull);              // v_mul_lo_u32
            if constexpr (OP == 1) x[i] = (u64)__umulhi((u32)x[i], (u32)b) | (x[i] & 0xFFFFFFFF00000000ull);       // v_mul_hi_u32
            if constexpr (OP == 2) x[i] = (u64)(u32)x[i] * (u32)b + x[i];                                          // v_mad_u64_u32
            if constexpr (OP == 3) x[i] = x[i] + b;                                                                // 64-bit add (2 instr)
            if constexpr (OP == 4) x[i] = x[i] * b;                                                                // 64-bit mul lo
            if constexpr (OP == 5) x[i] = __umul64hi(x[i], b);                                                     // 64-bit mul hi
            if constexpr (OP == 6) {   // Shoup product as in modarith.hpp mulmod_shoup: q^ = mulhi(x, w'), r = x*w - q^*q, one conditional subtraction
                const u64 q = 2305843009211596801ull, qh = __umul64hi(x[i], b);
                u64 r = x[i] * a - qh * q;
                x[i] = r >= q ? r - q : r;
            }
        }
    }
    u64 s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name)
{
    const int blocks = 256 * 8, iters = 2048;
    u64 *out;
    (void)hipMalloc(&out, blocks * 256 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 0x123456789abcdefull, 0xfedcba987654321ull, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 0x123456789abcdefull, 0xfedcba987654321ull, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters * 8;
    printf("%-34s %8.3f ms  %7.2f T lane-results/s\n", name, ms, n / ms / 1e9);
    (void)hipFree(out);
}

int main()
{
    run<0>("v_mul_lo_u32 (+ and/or)");
    run<1>("v_mul_hi_u32 (+ and/or)");
    run<2>("v_mad_u64_u32");
    run<3>("64-bit add");
    run<4>("64-bit mul lo");
    run<5>("64-bit mul hi");
    run<6>("Shoup product, 61-bit prime");
    return 0;
}
