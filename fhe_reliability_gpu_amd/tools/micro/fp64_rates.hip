// Measurement tool: issue rate of the FP64 vector instructions the ArithF64 butterflies are made of (gfx950).
// hipcc -O3 --offload-arch=gfx950 fp64_rates.hip -o fp64_rates && ./fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters)
{
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x * 0.001 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (OP == 0) x[i] = x[i] + b;
            if constexpr (OP == 1) x[i] = x[i] * b;
            if constexpr (OP == 2) x[i] = __builtin_fma(x[i], b, a);
            if constexpr (OP == 3) x[i] = __builtin_rint(x[i]) + b;          // rint + add
            if constexpr (OP == 4) x[i] = (x[i] + 6755399441055744.0) - b;   // add + add (magic rounding shape)
            if constexpr (OP == 5) {   // mulmod as in ArithF64::mulmod
                double h = x[i] * a, kq = __builtin_rint(x[i] * b), l = __builtin_fma(x[i], a, -h), r = __builtin_fma(-kq, 1125899903107073.0, h);
                x[i] = r + l;
            }
            if constexpr (OP == 6) {   // mulmod with the quotient rounded by a magic-constant fma
                double h = x[i] * a, kq = __builtin_fma(x[i], b, 6755399441055744.0) - 6755399441055744.0, l = __builtin_fma(x[i], a, -h),
                       r = __builtin_fma(-kq, 1125899903107073.0, h);
                x[i] = r + l;
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, int ops_per_elem)
{
    const int blocks = 256 * 8, iters = 4096;
    double *out;
    hipMalloc(&out, blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1.000001, 0.999999, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 1.000001, 0.999999, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters * 8;
    printf("%-28s %8.3f ms  %7.2f T lane-results/s  (%d instr each -> %7.2f T lane-instr/s)\n", name, ms, n / ms / 1e9, ops_per_elem, n * ops_per_elem / ms / 1e9);
    hipFree(out);
}

int main()
{
    run<0>("v_add_f64", 1);
    run<1>("v_mul_f64", 1);
    run<2>("v_fma_f64", 1);
    run<3>("v_rndne_f64 + v_add_f64", 2);
    run<4>("v_add_f64 + v_add_f64", 2);
    run<5>("mulmod (rint)", 6);
    run<6>("mulmod (magic fma)", 6);
    return 0;
}
