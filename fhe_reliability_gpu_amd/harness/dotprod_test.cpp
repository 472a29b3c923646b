// Encrypted dot-product harness for the MI355X engine.  Three command lines are built from this file:
//
//   dotprod_test <bits_per_symbol> <num_symbols>   one encrypted dot product with software bit flips in x
//   dotprod_real_test                              (-DREAL_TEST) the same without flips: any mismatch is hardware
//   naive_gemm_test                                (-DGEMM_LOOP) 100 encrypted dot products, keys reused, no report
//
// They stand in for reliability_test/{dotprod_test,dotprod_real_test,naive_gemm_test}.cu: the sweep scripts
// (run_dotprod_simu.sh:13-38) only append stdout/stderr to log files, so the contract is the command line
// (dotprod_test.cu:190-196), the parameter set (BGV, N = 16384, 6 x 50-bit primes, 2 special, 20-bit batching
// modulus, :198-204) and the text of the report lines as logged in reliability_test/data/bits1-16_num1.txt:4-37.
// Everything numeric happens in the engine through include/phantom_bgv_shim.hpp; this file is a driver:
// it draws inputs, runs the pipeline stage by stage and formats the report.
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/phantom_bgv_shim.hpp"

namespace {

using Slots = std::vector<uint64_t>;

enum class Variant { Inject, Real, Loop };
#if defined(GEMM_LOOP)
constexpr Variant kVariant = Variant::Loop;
#elif defined(REAL_TEST)
constexpr Variant kVariant = Variant::Real;
#else
constexpr Variant kVariant = Variant::Inject;
#endif

// the ring and the plaintext space every variant uses
phantom::EncryptionParameters make_parameters()
{
    constexpr size_t degree = 16384;
    phantom::EncryptionParameters p(phantom::scheme_type::bgv);
    p.set_poly_modulus_degree(degree);
    p.set_coeff_modulus(phantom::arith::CoeffModulus::Create(degree, std::vector<int>(6, 50)));
    p.set_special_modulus_size(2);
    p.set_plain_modulus(phantom::arith::PlainModulus::Batching(degree, 20));
    return p;
}

// keys + encoder, generated once per process
struct Party {
    PhantomContext &ctx;
    PhantomSecretKey sk;
    PhantomPublicKey pk;
    PhantomRelinKey rlk;
    PhantomBatchEncoder codec;
    explicit Party(PhantomContext &c) : ctx(c), sk(c), pk(sk.gen_publickey(c)), rlk(sk.gen_relinkey(c)), codec(c) {}

    PhantomCiphertext seal(const Slots &v)
    {
        PhantomCiphertext ct;
        pk.encrypt_asymmetric(ctx, codec.encode(ctx, v), ct);
        return ct;
    }
    Slots open(const PhantomCiphertext &ct) { return codec.decode(ctx, sk.decrypt(ctx, ct)); }
};

Slots draw_slots(size_t n)
{
    Slots v(n);
    for (auto &x : v) x = (uint64_t)(rand() % 100);
    return v;
}

// the two input vectors are drawn interleaved (x0, y0, x1, y1, ...), as the logged runs were
void draw_pair(Slots &x, Slots &y)
{
    for (size_t i = 0; i < x.size(); ++i) {
        x[i] = (uint64_t)(rand() % 100);
        y[i] = (uint64_t)(rand() % 100);
    }
}

// XOR `per_word` random bit positions into each of `words` random ciphertext words, on the device
void corrupt(PhantomCiphertext &ct, int per_word, int words)
{
    const size_t span = ct.size() * ct.coeff_modulus_size() * ct.poly_modulus_degree();
    for (int w = 0; w < words; ++w) {
        const size_t at = (size_t)rand() % span;
        for (int k = 0; k < per_word; ++k) {
            const size_t bit = (size_t)rand() % 64;
            phantom::detail::must(fhe_flip_bit(phantom::detail::engine(), ct.data(), at, (int)bit, nullptr), "flip");
            phantom::detail::must(fhe_sync(phantom::detail::engine(), nullptr), "sync");
            std::cerr << "Injected bitflip @ idx=" << at << ", bit=" << bit << "\n";
        }
    }
}

// slot-wise product x (.) y, relinearised and switched down one level
PhantomCiphertext hadamard(Party &who, const PhantomCiphertext &x, const PhantomCiphertext &y)
{
    PhantomCiphertext z = phantom::multiply(who.ctx, x, y);
    phantom::relinearize_inplace(who.ctx, z, who.rlk);
    phantom::mod_switch_to_next_inplace(who.ctx, z);
    return z;
}

// log2(row) rotate-and-add steps: every slot of a row ends up holding the row's sum
void fold_rows(Party &who, PhantomCiphertext &z, const PhantomGaloisKey &gk, size_t row)
{
    for (size_t shift = 1; shift < row; shift *= 2) {
        PhantomCiphertext moved = z;
        phantom::rotate_inplace(who.ctx, moved, (int)shift, gk);
        phantom::add_inplace(who.ctx, z, moved);
    }
}

struct Mismatch {
    size_t slots = 0, bits = 0;
};
Mismatch compare(const Slots &want, const Slots &got)
{
    Mismatch m;
    for (size_t i = 0; i < want.size(); ++i) {
        const uint64_t x = want[i] ^ got[i];
        m.slots += x != 0;
        m.bits += (size_t)__builtin_popcountll(x);
    }
    return m;
}

void show(const char *label, const Slots &v)
{
    std::cout << label;
    print_vector(v, 3, 7);
}

int run_single(PhantomContext &ctx, uint64_t t, int per_word, int words)
{
    using std::cout;
    using std::endl;
    cout << "Example: BGV HomMul test" << endl;
    Party who(ctx);
    const size_t slots = who.codec.slot_count(), row = slots / 2;
    cout << "Plaintext matrix row size: " << row << endl;

    Slots x(slots), y(slots), want(slots);
    draw_pair(x, y);
    show("Input vector 1: ", x);
    show("Input vector 2: ", y);
    uint64_t want_sum = 0;
    for (size_t i = 0; i < slots; ++i) {
        want[i] = x[i] * y[i] % t;
        want_sum = (want_sum + want[i]) % t;
    }

    PhantomCiphertext cx = who.seal(x), cy = who.seal(y);
    if (kVariant == Variant::Inject) corrupt(cx, per_word, words);

    cout << "Compute x * y homomorphically..." << endl;
    PhantomCiphertext cz = hadamard(who, cx, cy);
    {
        const Slots got = who.open(cz);
        const Mismatch m = compare(want, got);
        show("Raw product vector: ", got);
        show("CPU baseline      : ", want);
        cout << "Elementwise symbol errors: " << m.slots << " / " << slots << endl;
        cout << "Elementwise Hamming distance (bit errors): " << m.bits << endl;
    }

    const PhantomGaloisKey gk = who.sk.create_galois_keys(ctx);
    fold_rows(who, cz, gk, row);
    const Slots folded = who.open(cz);
    const uint64_t got_sum = (folded[0] + folded[row]) % t;   // the two rows hold their own sums

    const uint64_t gap = got_sum > want_sum ? got_sum - want_sum : want_sum - got_sum;
    cout << "Decrypted dot product = " << got_sum << endl;
    cout << "Expected (CPU)         = " << want_sum << endl;
    cout << "Dot product bit errors (Hamming distance): " << __builtin_popcountll(got_sum ^ want_sum) << endl;
    cout << "Absolute difference   = " << gap << endl;
    if (want_sum == 0) cout << "Percentage error      = undefined (expected is zero)" << endl;
    else cout << "Percentage error      = " << 100.0 * ((double)gap / (double)want_sum) << (kVariant == Variant::Real ? " %" : "%") << endl;
    cout << (got_sum == want_sum ? "✔ Dot product matches CPU result." : "✖ MISMATCH detected!") << endl;
    return 0;
}

// timing body without a report: fresh inputs every round, keys made once
int run_loop(PhantomContext &ctx, int rounds)
{
    Party who(ctx);
    const PhantomGaloisKey gk = who.sk.create_galois_keys(ctx);
    const size_t slots = who.codec.slot_count();
    for (int r = 0; r < rounds; ++r) {
        Slots x(slots), y(slots);
        draw_pair(x, y);
        PhantomCiphertext cz = hadamard(who, who.seal(x), who.seal(y));
        fold_rows(who, cz, gk, slots / 2);
        (void)who.open(cz);
    }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    srand((unsigned)time(nullptr));
    int per_word = 0, words = 0;
    if (kVariant == Variant::Inject) {
        if (argc != 3) {
            std::cerr << "Usage: " << argv[0] << " <bits_per_symbol> <num_symbols>\n";
            return 1;
        }
        per_word = atoi(argv[1]);
        words = atoi(argv[2]);
    }
    try {
        const phantom::EncryptionParameters parms = make_parameters();
        PhantomContext ctx(parms);
        print_parameters(ctx);
        std::cout << std::endl;
        return kVariant == Variant::Loop ? run_loop(ctx, 100) : run_single(ctx, parms.plain_modulus().value(), per_word, words);
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << "\n";
        return 1;
    }
}
