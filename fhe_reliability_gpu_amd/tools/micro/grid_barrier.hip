// Measurement tool: cost of a grid-wide barrier among co-resident workgroups (device-scope atomic counter + spin) against the
// cost of a dependent kernel launch on the same stream (gfx950).  Every spin is bounded: a workgroup that does not see the
// barrier complete within the bound sets an error word and leaves.
// hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, unsigned *err)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000u) {
                atomicExch(err, 1u);
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_barriers(unsigned *counter, unsigned *err, double *data, int rounds)
{
    double x = data[blockIdx.x * 256 + threadIdx.x];
    for (int r = 0; r < rounds; ++r) {
        x = x * 1.0000001 + 1.0;
        data[blockIdx.x * 256 + threadIdx.x] = x;      // something for the release to publish
        if (!grid_barrier(counter, (unsigned)(r + 1) * gridDim.x, err)) return;
        x += data[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];   // read a neighbour's word written before the barrier
    }
    data[blockIdx.x * 256 + threadIdx.x] = x;
}

__global__ __launch_bounds__(256) void k_small(double *data)
{
    double x = data[blockIdx.x * 256 + threadIdx.x];
    data[blockIdx.x * 256 + threadIdx.x] = x * 1.0000001 + 1.0;
}

int main()
{
    unsigned *counter, *err;
    double *data;
    hipMalloc(&counter, 4);
    hipMalloc(&err, 4);
    hipMalloc(&data, 2048 * 256 * 8);
    hipMemset(data, 0, 2048 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int per_cu = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_barriers, 256, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("CUs %d, co-resident workgroups per CU %d\n", prop.multiProcessorCount, per_cu);
    for (int grid : {64, 256, 512, 1024}) {
        if (grid > prop.multiProcessorCount * per_cu) continue;
        for (int rounds : {10, 100}) {
            hipMemset(counter, 0, 4);
            hipMemset(err, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k_barriers<<<grid, 256>>>(counter, err, data, rounds);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            unsigned h = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost);
            printf("grid %4d, %3d barriers in one launch: %8.1f us total, %6.2f us per barrier%s\n", grid, rounds, ms * 1e3, ms * 1e3 / rounds, h ? "  [SPIN BOUND HIT]" : "");
        }
    }
    for (int grid : {64, 256, 1024}) {
        for (int i = 0; i < 20; ++i) k_small<<<grid, 256>>>(data);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) k_small<<<grid, 256>>>(data);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("grid %4d, 200 dependent launches: %6.2f us per launch\n", grid, ms * 1e3 / 200);
    }
    return 0;
}
