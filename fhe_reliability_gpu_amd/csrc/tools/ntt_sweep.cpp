// ntt_sweep.cpp -- timing sweeps of the NTT launch variants through the C ABI
// (tuning tool, not part of the library):
//   ./ntt_sweep <log_n> <limbs> <polys> <bits> [reps]
// For every (mode, dist, wgs) setting: checks the fused result against the two-launch
// result word for word, then times `reps` forward transforms with HIP events on the
// context's stream.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/fhe_mi355x.h"

#define OK(x) do { int rc_ = (x); if (rc_) { printf("FAIL %s -> %d: %s\n", #x, rc_, fhe_last_error()); return 1; } } while (0)

static uint64_t splitmix(uint64_t &s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    int log_n = argc > 1 ? atoi(argv[1]) : 16, limbs = argc > 2 ? atoi(argv[2]) : 1, polys = argc > 3 ? atoi(argv[3]) : 256;
    int bits = argc > 4 ? atoi(argv[4]) : 50, reps = argc > 5 ? atoi(argv[5]) : 20;
    int inverse = argc > 6 ? atoi(argv[6]) : 0;
    const size_t N = (size_t)1 << log_n, words = N * limbs * polys;
    fhe_ctx *ctx;
    OK(fhe_ctx_create(0, &ctx));
    std::vector<int> b(limbs, bits);
    std::vector<uint64_t> q(limbs);
    OK(fhe_moduli_create(N, b.data(), limbs, q.data()));
    fhe_ntt_tables *t;
    OK(fhe_ntt_tables_create(ctx, log_n, q.data(), limbs, &t));
    std::vector<uint64_t> h(words), ref(words), got(words);
    uint64_t seed = 2025;
    for (int p = 0; p < polys; p++)
        for (int l = 0; l < limbs; l++)
            for (size_t i = 0; i < N; i++) h[((size_t)p * limbs + l) * N + i] = splitmix(seed) % q[l];
    void *d, *d0, *stream;
    OK(fhe_alloc(ctx, words * 8, &d));
    OK(fhe_alloc(ctx, words * 8, &d0));
    OK(fhe_ctx_stream(ctx, &stream));
    OK(fhe_h2d(ctx, d0, h.data(), words * 8, nullptr));
    OK(fhe_sync(ctx, nullptr));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](void) { return inverse ? fhe_ntt_inverse_batch(ctx, (uint64_t *)d, t, polys, limbs, 0, nullptr)
                                          : fhe_ntt_forward_batch(ctx, (uint64_t *)d, t, polys, limbs, 0, nullptr); };
    struct Cfg { int mode, dist, wgs, nt; };
    std::vector<Cfg> cfgs = {{0, 0, 0, 0}, {0, 1, 0, 0}};   // two-launch path: tile_geo 0 and 1 (in the dist column)
    for (int nt : {2, 10, 12, 14})
        for (int dist : {2, 3, 4})
            for (int wgs : {256, 384, 512, 768}) cfgs.push_back({1, dist, wgs, nt});
    if (const char *e = getenv("SWEEP_WGS")) {
        cfgs.resize(2);
        std::vector<int> variants = {2, 3, 4, 5, 6, 7};
        if (const char *v = getenv("SWEEP_VARIANTS")) {          // comma-separated fused_variant values
            variants.clear();
            for (const char *p = v; *p;) {
                variants.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p) p++;
            }
        }
        for (int nt : variants)
            for (int dist : {getenv("SWEEP_DIST") ? atoi(getenv("SWEEP_DIST")) : 3}) cfgs.push_back({1, dist, atoi(e), nt});
    }
    const double alg_bytes = 16.0 * N * limbs * polys;
    printf("# log_n=%d limbs=%d polys=%d bits=%d inverse=%d  (%.1f MiB in place)\n", log_n, limbs, polys, bits, inverse, words * 8 / 1048576.0);
    for (const Cfg &c : cfgs) {
        OK(fhe_ctx_set_option(ctx, "ntt_mode", c.mode));
        if (!c.mode) OK(fhe_ctx_set_option(ctx, "tile_geo", c.dist));
        if (c.mode) {
            OK(fhe_ctx_set_option(ctx, "fused_dist", c.dist));
            OK(fhe_ctx_set_option(ctx, "fused_wgs", c.wgs));
            OK(fhe_ctx_set_option(ctx, "fused_variant", c.nt));
        }
        OK(fhe_d2d(ctx, d, d0, words * 8, nullptr));
        OK(run());
        OK(fhe_d2h(ctx, got.data(), d, words * 8, nullptr));
        OK(fhe_sync(ctx, nullptr));
        OK(fhe_ctx_check(ctx));
        size_t bad = 0;
        if (c.mode == 0 && c.dist == 0) ref = got;
        else
            for (size_t i = 0; i < words; i++) bad += got[i] != ref[i];
        for (int w = 0; w < 3; w++) OK(run());
        hipEventRecord(e0, (hipStream_t)stream);
        for (int r = 0; r < reps; r++) OK(run());
        hipEventRecord(e1, (hipStream_t)stream);
        OK(fhe_sync(ctx, nullptr));
        OK(fhe_ctx_check(ctx));
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double per = ms / reps;
        printf("mode=%d dist=%d wgs=%4d variant=%d : %8.4f ms/step  %7.3f M NTT/s  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)  mismatches=%zu\n", c.mode,
               c.dist, c.wgs, c.nt, per, limbs * polys / per * 1e-3, alg_bytes / per * 1e-6, alg_bytes / per * 1e-6 / 80.0, bad);
        fflush(stdout);
    }
    return 0;
}
