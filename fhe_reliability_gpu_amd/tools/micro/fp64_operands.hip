// Measurement tool (round 3): does the FP64 product of ArithF64::mulmod issue slower when its constant factors sit in VECTOR registers
// (as the base-conversion kernel has them after reading them from LDS) than when they are wave-uniform SCALAR operands?
// hipcc -O3 --offload-arch=gfx950 -ffp-contract=off fp64_operands.hip -o fp64_operands && ./fp64_operands
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int WAVES8>
__global__ __launch_bounds__(256) void k(double *out, const double *tab, double a, double b, double n, int iters)
{
    double x[4];
    for (int i = 0; i < 4; ++i) x[i] = a + threadIdx.x * 0.001 + i;
    double va = a, vb = b, vn = n;
    if (MODE == 1) {                   // constants in vector registers (the compiler cannot prove them uniform)
        va = tab[threadIdx.x & 1];
        vb = tab[2 + (threadIdx.x & 1)];
        vn = tab[4 + (threadIdx.x & 1)];
    }
    for (int it = 0; it < iters; ++it) {
        // four independent products side by side, step by step (the base-conversion kernel's order)
        double h[4], kq[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = x[i] * va;
#pragma unroll
        for (int i = 0; i < 4; ++i) kq[i] = x[i] * vb;
#pragma unroll
        for (int i = 0; i < 4; ++i) kq[i] = __builtin_rint(kq[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = __builtin_fma(x[i], va, -h[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = __builtin_fma(-kq[i], vn, h[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = h[i] + l[i];
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int blocks)
{
    const int iters = 8192;
    double *out, *tab;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipMalloc(&tab, 64);
    const double h[6] = {1.000001, 1.000001, 0.999999, 0.999999, 1125899903107073.0, 1125899903107073.0};
    hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, 0><<<blocks, 256>>>(out, tab, 1.000001, 0.999999, 1125899903107073.0, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, 0><<<blocks, 256>>>(out, tab, 1.000001, 0.999999, 1125899903107073.0, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 256 * iters * 4 * 6;
    printf("%-44s %4d workgroups (%d waves/SIMD) %8.3f ms  %7.2f T lane-instr/s\n", name, blocks, blocks / 256, ms, n / ms / 1e9);
    hipFree(out);
    hipFree(tab);
}

int main()
{
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("product, constants scalar (kernel arguments)", blocks);
        run<1>("product, constants in vector registers", blocks);
    }
    return 0;
}
