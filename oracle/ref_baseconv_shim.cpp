// ref_baseconv_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the reference's own rfhe_framewk/src/baseConv.cpp (found through -I,
// never copied into this repository) into oracle/_ref/libref_baseconv.so and
// exposes its crt_reconstruct() (baseConv.cpp:146-173) through a C entry point so
// the oracle's CRT code can be cross-checked against the reference itself.
// The reference file's main() is renamed so the translation unit links as a library.
#define main ref_baseconv_main
#include "baseConv.cpp"
#undef main

extern "C" int ref_crt_reconstruct(const uint64_t *residues, const uint64_t *moduli, int m, int N,
                                   uint64_t *x_lo, uint64_t *x_hi)
{
    std::vector<std::vector<uint64>> res(m, std::vector<uint64>(N));
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < N; ++i) res[j][i] = residues[(size_t)j * N + i];
    std::vector<uint64> mod(moduli, moduli + m);
    std::vector<uint128> x = crt_reconstruct(res, mod);
    for (int i = 0; i < N; ++i) {
        x_lo[i] = (uint64_t)x[i];
        x_hi[i] = (uint64_t)(x[i] >> 64);
    }
    return 0;
}
