// pack_ubench.hip -- what would a 50-bit packed hand-off between the two NTT passes buy?  
// Memory patterns and pack / unpack arithmetic of the two passes at N = 2^16 without the butterflies:
//   plain : K1 reads a column tile (16 columns x 256 points), writes it back in place (8 B words);
//           K2 reads a row tile (16 rows x 256 points), writes it back in place               -> 4 sweeps of 8 B words
//   packed: K1 reads a column tile, canonicalises + packs 50-bit residues into 16x16 blocks of 1664 B staged through LDS,
//           writes 26 KiB per tile to a scratch; K2 reads the 16 blocks of its row tile through LDS, unpacks, writes the tile
//                                                                                              -> 2 sweeps of 8 B, 2 of 6.5 B
// hipcc -O3 --offload-arch=gfx950 pack_ubench.hip -o pack_ubench && ./pack_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
typedef uint64_t u64;
typedef uint32_t u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int BLK_BYTES = 1664, CHUNK_DW = 26;          // 16 chunks of 104 B (25 dwords of payload) per 16x16 block
constexpr u64 MASK50 = (1ull << 50) - 1;

__global__ __launch_bounds__(256) void k1_plain(u64 *__restrict__ buf, u32 limbs)
{
    const u32 t = blockIdx.x;
    u64 *p = buf + (size_t)(t / 16) * 65536 + (t % 16) * 16;
    const int c = threadIdx.x & 15, a = threadIdx.x >> 4;
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = p[(size_t)(a + 16 * k) * 256 + c];
#pragma unroll
    for (int k = 0; k < 16; k++) p[(size_t)(a * 16 + k) * 256 + c] = v[k] + 1;     // second step owns 16 consecutive points
}
__global__ __launch_bounds__(256) void k2_plain(ulonglong2 *__restrict__ buf, u32 limbs)
{
    ulonglong2 *p = buf + (size_t)blockIdx.x * 2048;                               // 16 rows x 256 points = 32 KiB
#pragma unroll
    for (int k = 0; k < 8; k++) {
        ulonglong2 v = p[threadIdx.x + 256 * k];
        v.x += 1;
        v.y += 1;
        p[threadIdx.x + 256 * k] = v;
    }
}

__global__ __launch_bounds__(256) void k1_packed(const u64 *__restrict__ in, u32 *__restrict__ scratch, double n, double ninv)
{
    __shared__ __attribute__((aligned(16))) u32 st[16 * BLK_BYTES / 4];
    const u32 t = blockIdx.x, unit = t / 16, C = t % 16;
    const u64 *p = in + (size_t)unit * 65536 + C * 16;
    const int c = threadIdx.x & 15, a = threadIdx.x >> 4;
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        double x = __builtin_bit_cast(double, p[(size_t)(a * 16 + k) * 256 + c]);
        double q = __builtin_rint(x * ninv);                                         // canonicalise the lazy value
        x = __builtin_fma(-q, n, x);
        x = x < 0.0 ? x + n : x;
        v[k] = (__builtin_bit_cast(u64, x + 4503599627370496.0)) & MASK50;
    }
    u32 *dst = st + a * (BLK_BYTES / 4) + c * CHUNK_DW;
    u64 acc = 0;
    int have = 0, w = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {                                                   // 16 x 50 bits -> 25 dwords
        acc |= v[k] << have;
        const u64 spill = have > 14 ? v[k] >> (64 - have) : 0;
        have += 50;
        while (have >= 32) {
            dst[w++] = (u32)acc;
            acc = (acc >> 32) | (have > 64 ? spill << 32 : 0);
            have -= 32;
        }
    }
    __syncthreads();
    uint4 *g = reinterpret_cast<uint4 *>(scratch + ((size_t)unit * 256 + (size_t)C * 16) * (BLK_BYTES / 4));
    const uint4 *s4 = reinterpret_cast<const uint4 *>(st);
    for (int i = threadIdx.x; i < 16 * BLK_BYTES / 16; i += 256) g[i] = s4[i];       // 26,624 B contiguous per tile
}

__global__ __launch_bounds__(256) void k2_packed(const u32 *__restrict__ scratch, u64 *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) u32 st[16 * BLK_BYTES / 4];
    const u32 t = blockIdx.x, unit = t / 16, P = t % 16;
    uint4 *s4 = reinterpret_cast<uint4 *>(st);
    for (int i = threadIdx.x; i < 16 * BLK_BYTES / 16; i += 256) {                   // block (P, C) for C = 0..15
        const int C = i / (BLK_BYTES / 16), o = i % (BLK_BYTES / 16);
        s4[i] = reinterpret_cast<const uint4 *>(scratch + ((size_t)unit * 256 + (size_t)C * 16 + P) * (BLK_BYTES / 4))[o];
    }
    __syncthreads();
    const int cc = threadIdx.x & 15, r = threadIdx.x >> 4;
    u64 *row = out + (size_t)unit * 65536 + (size_t)(P * 16 + r) * 256;
    const int bit = 50 * r, d = bit >> 5, s = bit & 31;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 *ch = st + k * (BLK_BYTES / 4) + cc * CHUNK_DW + d;
        const u64 lo = (u64)ch[0] | ((u64)ch[1] << 32), hi = ch[2];
        u64 val = (lo >> s) | (s > 14 ? hi << (64 - s) : 0);
        val &= MASK50;
        const double x = __builtin_bit_cast(double, val | 0x4330000000000000ull) - 4503599627370496.0;
        row[cc + 16 * k] = __builtin_bit_cast(u64, x);
    }
}

int main()
{
    const u32 limbs = 256;
    const size_t bytes = (size_t)limbs * 65536 * 8, sbytes = (size_t)limbs * 256 * BLK_BYTES;
    u64 *a;
    u32 *sc;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&sc, sbytes));
    CK(hipMemset(a, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double n = 1125899903107073.0, ninv = 1.0 / n;
    for (int round = 0; round < 2; round++) {
        for (int variant = 0; variant < 2; variant++) {
            for (int i = 0; i < 300; i++) {
                if (variant == 0) { k1_plain<<<limbs * 16, 256>>>(a, limbs); k2_plain<<<limbs * 16, 256>>>((ulonglong2 *)a, limbs); }
                else { k1_packed<<<limbs * 16, 256>>>(a, sc, n, ninv); k2_packed<<<limbs * 16, 256>>>(sc, a); }
            }
            CK(hipEventRecord(e0));
            const int reps = 1500;
            for (int i = 0; i < reps; i++) {
                if (variant == 0) { k1_plain<<<limbs * 16, 256>>>(a, limbs); k2_plain<<<limbs * 16, 256>>>((ulonglong2 *)a, limbs); }
                else { k1_packed<<<limbs * 16, 256>>>(a, sc, n, ninv); k2_packed<<<limbs * 16, 256>>>(sc, a); }
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s hand-off: %.4f ms per 256-limb pass pair (%s)\n", variant ? "packed" : "plain ", ms / reps,
                   variant ? "8 B in, 6.5 B scratch, 6.5 B back, 8 B out" : "4 sweeps of 8 B words");
        }
    }
    CK(hipGetLastError());
    return 0;
}
