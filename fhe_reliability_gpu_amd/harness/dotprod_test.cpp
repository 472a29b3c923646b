// dotprod_test for MI355X -- drop-in for reliability_test/dotprod_test.cu: same command line
//   dotprod_test <bits_per_symbol> <num_symbols>                       (dotprod_test.cu:190-196)
// same parameters (BGV, N = 16384, six 50-bit primes of which two special, 20-bit batching plain
// modulus, :198-204) and the same printed lines (:57-58,70,81,90-91,110,134-139,164-184; logged
// sample reliability_test/data/bits1-16_num1.txt:4-37), so run_dotprod_simu.sh keeps working.
// Build with -DGEMM_LOOP for the naive_gemm_test.cu variant (100 encrypted dot products, keys reused),
// with -DREAL_TEST for dotprod_real_test.cu: no arguments and no software flip (dotprod_real_test.cu:95-96,185),
// so any mismatch is a hardware fault.
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <vector>

#include "../../include/phantom_bgv_shim.hpp"

using namespace std;
using namespace phantom;
using namespace phantom::arith;

static EncryptionParameters parms(scheme_type::bgv);
#ifdef REAL_TEST
static const char *const PCT = " %";   // dotprod_real_test.cu:173
#else
static const char *const PCT = "%";    // dotprod_test.cu:176
#endif

// flip bits_per_symbol random bits in each of num_symbols random words of the ciphertext, on the device
static void inject_bitflip_ciphertext(PhantomCiphertext &ct, int bits_per_symbol, int num_symbols)
{
    const size_t total = ct.size() * ct.coeff_modulus_size() * ct.poly_modulus_degree();
    for (int s = 0; s < num_symbols; ++s) {
        const size_t idx = (size_t)rand() % total;
        for (int b = 0; b < bits_per_symbol; ++b) {
            const size_t bit = (size_t)rand() % 64;
            phantom::detail::must(fhe_flip_bit(phantom::detail::engine(), ct.data(), idx, (int)bit, nullptr), "flip");
            phantom::detail::must(fhe_sync(phantom::detail::engine(), nullptr), "sync");
            cerr << "Injected bitflip @ idx=" << idx << ", bit=" << bit << "\n";
        }
    }
}

static void dot_product_test(PhantomContext &context, int bits_per_symbol, int num_symbols)
{
    cout << "Example: BGV HomMul test" << endl;

    PhantomSecretKey secret_key(context);
    PhantomPublicKey public_key = secret_key.gen_publickey(context);
    PhantomRelinKey relin_keys = secret_key.gen_relinkey(context);

    PhantomBatchEncoder batch_encoder(context);
    const size_t slot_count = batch_encoder.slot_count(), row_size = slot_count / 2;
    cout << "Plaintext matrix row size: " << row_size << endl;

    vector<uint64_t> input1(slot_count), input2(slot_count);
    for (size_t i = 0; i < slot_count; i++) {
        input1[i] = (uint64_t)(rand() % 100);
        input2[i] = (uint64_t)(rand() % 100);
    }
    cout << "Input vector 1: ";
    print_vector(input1, 3, 7);
    cout << "Input vector 2: ";
    print_vector(input2, 3, 7);

    const uint64_t mod = parms.plain_modulus().value();
    vector<uint64_t> baseline(slot_count);
    for (size_t i = 0; i < slot_count; ++i) baseline[i] = (input1[i] * input2[i]) % mod;

    PhantomPlaintext x_plain = batch_encoder.encode(context, input1), y_plain = batch_encoder.encode(context, input2);
    PhantomCiphertext x_cipher, y_cipher;
    public_key.encrypt_asymmetric(context, x_plain, x_cipher);
    public_key.encrypt_asymmetric(context, y_plain, y_cipher);

#ifndef REAL_TEST
    inject_bitflip_ciphertext(x_cipher, bits_per_symbol, num_symbols);
#else
    (void)bits_per_symbol, (void)num_symbols, (void)inject_bitflip_ciphertext;
#endif

    cout << "Compute x * y homomorphically..." << endl;
    PhantomCiphertext xy_cipher = multiply(context, x_cipher, y_cipher);
    relinearize_inplace(context, xy_cipher, relin_keys);
    mod_switch_to_next_inplace(context, xy_cipher);

    {
        PhantomPlaintext xy_plain = secret_key.decrypt(context, xy_cipher);
        vector<uint64_t> prod = batch_encoder.decode(context, xy_plain);
        size_t symbol_errors = 0, bit_errors = 0;
        for (size_t i = 0; i < slot_count; ++i)
            if (baseline[i] != prod[i]) {
                ++symbol_errors;
                bit_errors += (size_t)__builtin_popcountll(baseline[i] ^ prod[i]);
            }
        cout << "Raw product vector: ";
        print_vector(prod, 3, 7);
        cout << "CPU baseline      : ";
        print_vector(baseline, 3, 7);
        cout << "Elementwise symbol errors: " << symbol_errors << " / " << slot_count << endl;
        cout << "Elementwise Hamming distance (bit errors): " << bit_errors << endl;
    }

    PhantomGaloisKey gal_keys = secret_key.create_galois_keys(context);
    for (size_t step = 1; step < row_size; step <<= 1) {
        PhantomCiphertext rotated = xy_cipher;
        rotate_inplace(context, rotated, (int)step, gal_keys);
        add_inplace(context, xy_cipher, rotated);
    }

    PhantomPlaintext dp_plain = secret_key.decrypt(context, xy_cipher);
    vector<uint64_t> result = batch_encoder.decode(context, dp_plain);
    const uint64_t result_full = (result[0] + result[row_size]) % mod;

    uint64_t expected = 0;
    for (size_t i = 0; i < slot_count; ++i) expected = (expected + baseline[i]) % mod;

    const size_t dp_bit_errors = (size_t)__builtin_popcountll(result_full ^ expected);
    cout << "Decrypted dot product = " << result_full << endl;
    cout << "Expected (CPU)         = " << expected << endl;
    cout << "Dot product bit errors (Hamming distance): " << dp_bit_errors << endl;
    const uint64_t abs_diff = result_full > expected ? result_full - expected : expected - result_full;
    cout << "Absolute difference   = " << abs_diff << endl;
    if (expected != 0) cout << "Percentage error      = " << (double)abs_diff / (double)expected * 100.0 << PCT << endl;
    else cout << "Percentage error      = undefined (expected is zero)" << endl;
    if (result_full == expected) cout << "✔ Dot product matches CPU result." << endl;
    else cout << "✖ MISMATCH detected!" << endl;
}

#ifdef GEMM_LOOP
// naive_gemm_test.cu:25-66,94-100: the same encrypted dot product 100 times with the keys reused
static vector<uint64_t> encrypted_dot_product(PhantomContext &context, PhantomSecretKey &secret_key, PhantomPublicKey &public_key,
                                              PhantomRelinKey &relin_keys, PhantomGaloisKey &gal_keys, PhantomBatchEncoder &enc)
{
    const size_t slot_count = enc.slot_count(), row_size = slot_count / 2;
    vector<uint64_t> input1(slot_count), input2(slot_count);
    for (size_t i = 0; i < slot_count; ++i) {
        input1[i] = (uint64_t)(rand() % 100);
        input2[i] = (uint64_t)(rand() % 100);
    }
    PhantomCiphertext x, y;
    public_key.encrypt_asymmetric(context, enc.encode(context, input1), x);
    public_key.encrypt_asymmetric(context, enc.encode(context, input2), y);
    PhantomCiphertext xy = multiply(context, x, y);
    relinearize_inplace(context, xy, relin_keys);
    mod_switch_to_next_inplace(context, xy);
    for (size_t step = 1; step < row_size; step <<= 1) {
        PhantomCiphertext tmp = xy;
        rotate_inplace(context, tmp, (int)step, gal_keys);
        add_inplace(context, xy, tmp);
    }
    return enc.decode(context, secret_key.decrypt(context, xy));
}
#endif

int main(int argc, char *argv[])
{
    srand((unsigned)time(NULL));
#if !defined(GEMM_LOOP) && !defined(REAL_TEST)
    if (argc != 3) {
        cerr << "Usage: " << argv[0] << " <bits_per_symbol> <num_symbols>\n";
        return 1;
    }
    const int bits_per_symbol = atoi(argv[1]), num_symbols = atoi(argv[2]);
#elif defined(REAL_TEST)
    (void)argc;
    (void)argv;
    const int bits_per_symbol = 0, num_symbols = 0;
#else
    (void)argc;
    (void)argv;
#endif
    const size_t poly_modulus_degree = 16384;
    parms.set_poly_modulus_degree(poly_modulus_degree);
    parms.set_coeff_modulus(CoeffModulus::Create(poly_modulus_degree, {50, 50, 50, 50, 50, 50}));
    parms.set_special_modulus_size(2);
    parms.set_plain_modulus(PlainModulus::Batching(poly_modulus_degree, 20));
    try {
        PhantomContext context(parms);
        print_parameters(context);
        cout << endl;
#ifndef GEMM_LOOP
        dot_product_test(context, bits_per_symbol, num_symbols);
#else
        PhantomSecretKey secret_key(context);
        PhantomPublicKey public_key = secret_key.gen_publickey(context);
        PhantomRelinKey relin_keys = secret_key.gen_relinkey(context);
        PhantomGaloisKey gal_keys = secret_key.create_galois_keys(context);
        PhantomBatchEncoder enc(context);
        for (int i = 0; i < 100; ++i) (void)encrypted_dot_product(context, secret_key, public_key, relin_keys, gal_keys, enc);
#endif
    } catch (const std::exception &e) {
        cerr << "ERROR: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
