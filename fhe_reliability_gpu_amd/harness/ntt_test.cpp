// ntt_test / ntt_real_test for MI355X -- drop-in for the reference's
// reliability_test/ntt_test.cu and ntt_real_test.cu command lines and text protocol, so
// that run_bench_test.sh, run_real_test.sh and test_scripts/gen_errorimpact.py keep
// working:
//
//   ntt_test <log_dim> <batch_size> <num_flips> <num_target_symbols>
//        flip <num_flips> distinct bits in each of <num_target_symbols> distinct words
//        (ntt_test.cu:201-219)
//   ntt_test <log_dim> <batch_size> <num_flips>            (legacy form of run_bench_test.sh:9)
//        flip <num_flips> distinct bits anywhere in the buffer (reliability_test/exp_log.txt:5)
//   ntt_real_test <log_dim> <batch_size> <num_flips>       (build with -DREAL_TEST)
//        no software flip: two identical runs must agree unless the hardware faults
//        (ntt_real_test.cu:108-117, run_real_test.sh:24)
//
// stdout / stderr lines are byte-compatible with the reference (ntt_test.cu:125-127,175-198);
// gen_errorimpact.py:28-29 parses the "[FAULT DETECTED]" line with a regex.
#include <cinttypes>
#include <cstring>
#include <iostream>
#include <random>
#include <set>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/phantom_shim.hpp"

using phantom::arith::CoeffModulus;
using phantom::arith::NTT;

namespace {

struct Args {
    size_t log_dim = 0, batch = 0;
    int flips = 0, symbols = 0;   // symbols == 0: legacy / real form
};

void die(const std::string &msg)
{
    std::cerr << msg << "\n";
    std::exit(1);
}

// transform the host buffer on the device and bring the result back
void transform(const std::vector<uint64_t> &in, std::vector<uint64_t> &out, uint64_t *dev, const DNTTTable &tables, size_t batch,
               const hipStream_t &s)
{
    const size_t bytes = in.size() * sizeof(uint64_t);
    if (hipMemcpyAsync(dev, in.data(), bytes, hipMemcpyHostToDevice, s) != hipSuccess) die("ERROR: host-to-device copy failed");
    nwt_2d_radix8_forward_inplace(dev, tables, batch, 0, s);
    if (hipMemcpyAsync(out.data(), dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) die("ERROR: device-to-host copy failed");
    if (hipStreamSynchronize(s) != hipSuccess) die("ERROR: stream synchronisation failed");
}

int run(const Args &a)
{
    const size_t dim = (size_t)1 << a.log_dim;
    const uint64_t total = (uint64_t)a.batch * dim, total_bits = total * 64;

#ifndef REAL_TEST
    if (a.symbols > 0) {
        if ((uint64_t)a.flips > 64) die("ERROR: num_flips (" + std::to_string(a.flips) + ") exceeds bits per symbol (64).");
        if ((uint64_t)a.symbols > total)
            die("ERROR: num_target_symbols (" + std::to_string(a.symbols) + ") exceeds total symbols (" + std::to_string(total) + ").");
    } else
#endif
        if ((uint64_t)a.flips > total_bits)
        die("ERROR: num_flips (" + std::to_string(a.flips) + ") exceeds total possible bits (" + std::to_string(total_bits) + ").");

    phantom::util::cuda_stream_wrapper stream_wrapper;
    const auto &s = stream_wrapper.get_stream();

    // moduli: one 50-bit prime per batch entry, then their device-side constants and twiddles
    const auto moduli = CoeffModulus::Create(dim, std::vector<int>(a.batch, 50));
    auto dmod = phantom::util::make_cuda_auto_ptr<DModulus>(a.batch, s);
    DNTTTable tables;
    tables.init(dim, a.batch, s);
    for (size_t i = 0; i < a.batch; i++) {
        dmod.get()[i].set(moduli[i].value(), moduli[i].const_ratio()[0], moduli[i].const_ratio()[1]);
        NTT host_tables((int)a.log_dim, moduli[i]);
        tables.set(&dmod.get()[i], host_tables.get_from_root_powers().data(), host_tables.get_from_root_powers_shoup().data(),
                   nullptr, nullptr, 0, 0, i, s);
    }

    // uniform residues per limb
    std::random_device rd;
    std::mt19937_64 rng(rd());
    std::vector<uint64_t> data(total), clean(total), faulty(total);
    for (size_t i = 0; i < a.batch; i++) {
        std::uniform_int_distribution<uint64_t> pick(0, moduli[i].value() - 1);
        for (size_t j = 0; j < dim; j++) data[i * dim + j] = pick(rng);
    }

    auto dev = phantom::util::make_cuda_auto_ptr<uint64_t>(total, s);
    transform(data, clean, dev.get(), tables, a.batch, s);

#ifndef REAL_TEST
    // choose the bits to flip on the host copy
    std::set<uint64_t> bits;   // global bit indices
    if (a.symbols > 0) {
        std::unordered_set<uint64_t> words;
        std::uniform_int_distribution<uint64_t> pick_word(0, total - 1);
        std::uniform_int_distribution<int> pick_bit(0, 63);
        while ((int)words.size() < a.symbols) words.insert(pick_word(rng));
        for (uint64_t w : words) {
            std::unordered_set<int> pos;
            while ((int)pos.size() < a.flips) pos.insert(pick_bit(rng));
            for (int b : pos) bits.insert(w * 64 + (uint64_t)b);
        }
        std::cout << "[2D] Flipping " << bits.size() << " bits across " << words.size() << " symbols...\n";
    } else {
        std::uniform_int_distribution<uint64_t> pick(0, total_bits - 1);
        while ((int)bits.size() < a.flips) bits.insert(pick(rng));
        std::cout << "[2D] Flipping " << a.flips << " unique random bit" << (a.flips == 1 ? "" : "s") << " in the data buffer...\n";
    }
    for (uint64_t g : bits) data[g / 64] ^= (uint64_t)1 << (g % 64);
#endif

    transform(data, faulty, dev.get(), tables, a.batch, s);

    // Hamming distance and affected symbols
    size_t hamming = 0, mismatched = 0;
    std::vector<size_t> where;
    for (size_t i = 0; i < total; i++) {
        const uint64_t diff = clean[i] ^ faulty[i];
        if (diff) {
            if (++mismatched <= 128) where.push_back(i);
            hamming += (size_t)__builtin_popcountll(diff);
        }
    }
    if (hamming) {
        const double ber = (double)hamming / (double)total_bits, ser = (double)mismatched / (double)total;
        std::cout << "ERROR! Total bitwise Hamming distance = " << hamming << " (bit error rate = " << ber << ")\n";
        std::cout << "       Affected symbols = " << mismatched << "/" << total << " (symbol error rate = " << ser << ")\n";
        std::fprintf(stderr, "[FAULT DETECTED] Bit error: %zu/%zu = %.6f, Symbol error: %zu/%zu = %.6f\n", hamming,
                     (size_t)total_bits, ber, mismatched, (size_t)total, ser);
    } else {
        std::cout << "ALL CORRECT\n";
    }
#ifndef REAL_TEST
    if (a.symbols > 0 && mismatched > 0 && mismatched <= 128) {
        std::cout << "[Debug] Symbol mismatch indices:\n";
        for (size_t i : where) std::cout << "  - Index " << i << "\n";
    }
#endif
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
#ifdef REAL_TEST
    const int need = 4;
    const char *usage = " <log_dim> <batch_size> <num_flips>\n";
#else
    const int need = 4;   // the legacy 3-argument form is accepted as well as the 4-argument one
    const char *usage = " <log_dim> <batch_size> <num_flips> <num_target_symbols>\n";
#endif
    if (argc < need) {
        std::cerr << "Usage: " << argv[0] << usage;
        return 1;
    }
    Args a;
    try {
        a.log_dim = std::stoul(argv[1]);
        a.batch = std::stoul(argv[2]);
        a.flips = std::stoi(argv[3]);
#ifndef REAL_TEST
        if (argc >= 5) a.symbols = std::stoi(argv[4]);
#endif
    } catch (const std::exception &) {
        std::cerr << "Usage: " << argv[0] << usage;
        return 1;
    }
#ifdef REAL_TEST
    if (a.flips < 1) {
        std::cerr << "Error: <num_flips> must be >= 1\n";
        return 1;
    }
#else
    if (a.flips < 1 || (argc >= 5 && a.symbols < 1)) {
        std::cerr << "Error: <num_flips> and <num_target_symbols> must be >= 1\n";
        return 1;
    }
#endif
    if (a.log_dim < 1 || a.log_dim > 20 || a.batch < 1) {
        std::cerr << "Error: <log_dim> must be in [1, 20] and <batch_size> >= 1\n";
        return 1;
    }
    try {
        return run(a);
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        return 1;
    }
}
