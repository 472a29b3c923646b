// phantom_bgv_shim.hpp -- the Phantom scheme-level names reliability_test/dotprod_test.cu and
// naive_gemm_test.cu use ("scheme level"), as a minimal BGV layer whose every
// polynomial operation runs through the C ABI of libfhe_mi355x.so:
//
//   EncryptionParameters, scheme_type::bgv, PlainModulus::Batching            dotprod_test.cu:25,199-204
//   PhantomContext, print_parameters                                           :207-210
//   PhantomSecretKey{gen_publickey, gen_relinkey, create_galois_keys, decrypt} :73-76,119,143,152
//   PhantomPublicKey::encrypt_asymmetric                                       :104-106
//   PhantomBatchEncoder{slot_count, encode, decode}                            :78-80,100-101,120
//   PhantomPlaintext, PhantomCiphertext{size, coeff_modulus_size, poly_modulus_degree, data}  :41-46
//   multiply, relinearize_inplace, mod_switch_to_next_inplace, rotate_inplace, add_inplace    :113-115,146-147
//
// Scheme: textbook RNS-BGV.  c0 + c1 s = m + t e (mod Q); hybrid key switching with one prime per
// digit and the special primes as P (fhe_keyswitch, BGV mod-down); modulus switching keeps the
// plaintext up to the factor q_last^-1 mod t, tracked per ciphertext (correction factor); batching
// uses SEAL's slot order (generator 3, two rows).  Phantom's own ciphertext bytes are not reproduced
// (its source is absent and its sampling is random); what carries over is the behaviour the
// harness checks: without a fault the decrypted dot product equals the plaintext one.
#pragma once
#include <algorithm>
#include <iostream>
#include <map>
#include <random>

#include "phantom_shim.hpp"

namespace phantom {

enum class scheme_type { none, bgv };

namespace arith {
struct PlainModulus {
    // largest prime below 2^bits that is 1 mod 2N (SEAL's PlainModulus::Batching)
    static Modulus Batching(size_t poly_modulus_degree, int bit_size) { return CoeffModulus::Create(poly_modulus_degree, {bit_size})[0]; }
};
} // namespace arith

class EncryptionParameters {
public:
    explicit EncryptionParameters(scheme_type s = scheme_type::bgv) : scheme_(s) {}
    void set_poly_modulus_degree(size_t n) { n_ = n; }
    void set_coeff_modulus(const std::vector<arith::Modulus> &m) { coeff_ = m; }
    void set_special_modulus_size(size_t k) { special_ = k; }
    void set_plain_modulus(const arith::Modulus &t) { plain_ = t; }
    scheme_type scheme() const { return scheme_; }
    size_t poly_modulus_degree() const { return n_; }
    const std::vector<arith::Modulus> &coeff_modulus() const { return coeff_; }
    size_t special_modulus_size() const { return special_; }
    const arith::Modulus &plain_modulus() const { return plain_; }

private:
    scheme_type scheme_;
    size_t n_ = 0, special_ = 1;
    std::vector<arith::Modulus> coeff_;
    arith::Modulus plain_;
};

namespace bgv_detail {

typedef unsigned __int128 u128;
inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t m) { return (uint64_t)((u128)a * b % m); }
inline uint64_t powmod(uint64_t b, uint64_t e, uint64_t m)
{
    uint64_t r = 1 % m;
    for (b %= m; e; e >>= 1, b = mulmod(b, b, m))
        if (e & 1) r = mulmod(r, b, m);
    return r;
}
inline uint64_t invmod_prime(uint64_t a, uint64_t p) { return powmod(a % p, p - 2, p); }

// little bignum (base 2^32) for floor(Q/2) mod small moduli
struct Big {
    std::vector<uint32_t> d{1};
    void mul(uint64_t m)
    {
        std::vector<uint32_t> out(d.size() + 2, 0);
        const uint32_t m0 = (uint32_t)m, m1 = (uint32_t)(m >> 32);
        for (size_t i = 0; i < d.size(); i++) {
            uint64_t c = (uint64_t)d[i] * m0 + out[i];
            out[i] = (uint32_t)c;
            uint64_t c2 = (uint64_t)d[i] * m1 + out[i + 1] + (c >> 32);
            out[i + 1] = (uint32_t)c2;
            size_t k = i + 2;
            uint64_t carry = c2 >> 32;
            while (carry) {
                uint64_t s = (uint64_t)out[k] + carry;
                out[k] = (uint32_t)s;
                carry = s >> 32;
                k++;
            }
        }
        while (out.size() > 1 && out.back() == 0) out.pop_back();
        d = out;
    }
    void half()
    {
        uint32_t carry = 0;
        for (size_t i = d.size(); i-- > 0;) {
            uint32_t nc = d[i] & 1;
            d[i] = (d[i] >> 1) | (carry << 31);
            carry = nc;
        }
    }
    uint64_t mod(uint64_t m) const
    {
        u128 r = 0;
        for (size_t i = d.size(); i-- > 0;) r = ((r << 32) | d[i]) % m;
        return (uint64_t)r;
    }
};

inline void must(int rc, const char *what) { phantom::detail::must(rc, what); }

// device polynomial block: [parts][limbs][N] words
struct DevPoly {
    uint64_t *p = nullptr;
    size_t words = 0;
    DevPoly() = default;
    explicit DevPoly(size_t w) { alloc(w); }
    DevPoly(const DevPoly &o) { *this = o; }
    DevPoly(DevPoly &&o) noexcept : p(o.p), words(o.words) { o.p = nullptr; o.words = 0; }
    DevPoly &operator=(DevPoly &&o) noexcept
    {
        if (this != &o) {
            release();
            p = o.p;
            words = o.words;
            o.p = nullptr;
            o.words = 0;
        }
        return *this;
    }
    DevPoly &operator=(const DevPoly &o)
    {
        if (this != &o) {
            alloc(o.words);
            if (words) must(fhe_d2d(phantom::detail::engine(), p, o.p, words * 8, nullptr), "copy");
        }
        return *this;
    }
    ~DevPoly() { release(); }
    void alloc(size_t w)
    {
        release();
        words = w;
        if (w) {
            void *q = nullptr;
            must(fhe_alloc(phantom::detail::engine(), w * 8, &q), "alloc");
            p = static_cast<uint64_t *>(q);
        }
    }
    void release()
    {
        if (p) {
            fhe_sync(phantom::detail::engine(), nullptr);
            fhe_free(phantom::detail::engine(), p);
        }
        p = nullptr;
        words = 0;
    }
    void upload(const std::vector<uint64_t> &h)
    {
        if (words != h.size()) alloc(h.size());
        must(fhe_h2d(phantom::detail::engine(), p, h.data(), h.size() * 8, nullptr), "h2d");
        must(fhe_sync(phantom::detail::engine(), nullptr), "sync");
    }
    std::vector<uint64_t> download() const
    {
        std::vector<uint64_t> h(words);
        must(fhe_d2h(phantom::detail::engine(), h.data(), p, words * 8, nullptr), "d2h");
        must(fhe_sync(phantom::detail::engine(), nullptr), "sync");
        return h;
    }
};

} // namespace bgv_detail
} // namespace phantom

class PhantomContext;
class PhantomCiphertext;

class PhantomPlaintext {
public:
    std::vector<uint64_t> coeffs;   // N coefficients in [0, t)
};

class PhantomCiphertext {
public:
    size_t size() const { return size_; }
    size_t coeff_modulus_size() const { return limbs_; }
    size_t poly_modulus_degree() const { return n_; }
    uint64_t *data() { return buf_.p; }
    const uint64_t *data() const { return buf_.p; }
    uint64_t *part(size_t i) { return buf_.p + i * limbs_ * n_; }
    const uint64_t *part(size_t i) const { return buf_.p + i * limbs_ * n_; }
    void resize(size_t size, size_t limbs, size_t n)
    {
        size_ = size;
        limbs_ = limbs;
        n_ = n;
        buf_.alloc(size * limbs * n);
    }
    uint64_t correction = 1;        // decrypted plaintext must be multiplied by this (mod t)

private:
    size_t size_ = 0, limbs_ = 0, n_ = 0;
    phantom::bgv_detail::DevPoly buf_;
};

// one key-switching key: [L][2][L+K][N] at the top level, repacked per level on demand
struct PhantomKSwitchKey {
    phantom::bgv_detail::DevPoly full;
    mutable std::map<size_t, phantom::bgv_detail::DevPoly> per_level;
};
struct PhantomRelinKey {
    PhantomKSwitchKey key;
};
struct PhantomGaloisKey {
    std::map<uint32_t, PhantomKSwitchKey> keys;   // by Galois element
};

class PhantomContext {
public:
    explicit PhantomContext(const phantom::EncryptionParameters &parms) : parms_(parms)
    {
        using namespace phantom::bgv_detail;
        if (parms.scheme() != phantom::scheme_type::bgv) throw std::invalid_argument("only scheme_type::bgv is provided");
        n_ = parms.poly_modulus_degree();
        while (((size_t)1 << log_n_) < n_) log_n_++;
        K_ = parms.special_modulus_size();
        const auto &cm = parms.coeff_modulus();
        if (cm.size() <= K_) throw std::invalid_argument("coeff_modulus must hold data and special primes");
        L_ = cm.size() - K_;
        for (auto &m : cm) primes_.push_back(m.value());
        t_ = parms.plain_modulus().value();
        auto *ctx = phantom::detail::engine();
        must(fhe_ntt_tables_create(ctx, log_n_, primes_.data(), (int)primes_.size(), &full_), "tables");
        must(fhe_ntt_tables_create(ctx, log_n_, &t_, 1, &plain_), "plain tables");
        // SEAL's slot order: generator 3, two rows of N/2 slots
        const size_t row = n_ / 2, m = 2 * n_;
        index_map_.resize(n_);
        uint64_t pos = 1;
        for (size_t i = 0; i < row; i++) {
            index_map_[i] = bitrev((pos - 1) >> 1);
            index_map_[row + i] = bitrev((m - pos - 1) >> 1);
            pos = pos * 3 % m;
        }
    }
    PhantomContext(const PhantomContext &) = delete;
    ~PhantomContext()
    {
        for (auto &kv : levels_) {
            fhe_keyswitch_destroy(kv.second.ks);
            fhe_baseconv_destroy(kv.second.to_plain);
            if (kv.second.tables != full_) fhe_ntt_tables_destroy(kv.second.tables);
        }
        fhe_ntt_tables_destroy(plain_);
        fhe_ntt_tables_destroy(full_);
    }

    struct Level {
        fhe_ntt_tables *tables = nullptr;     // [q_0 .. q_{l-1}, p_0 .. p_{K-1}]
        fhe_keyswitch *ks = nullptr;
        fhe_baseconv *to_plain = nullptr;     // Q_l -> {t}
        std::vector<uint64_t> half_mod_q;     // floor(Q_l / 2) mod q_j
        uint64_t half_mod_t = 0;
    };
    const Level &level(size_t l) const
    {
        using namespace phantom::bgv_detail;
        auto it = levels_.find(l);
        if (it != levels_.end()) return it->second;
        Level lv;
        auto *ctx = phantom::detail::engine();
        std::vector<uint64_t> ps(primes_.begin(), primes_.begin() + l);
        ps.insert(ps.end(), primes_.begin() + L_, primes_.end());
        if (l == L_) lv.tables = full_;
        else must(fhe_ntt_tables_create(ctx, log_n_, ps.data(), (int)ps.size(), &lv.tables), "level tables");
        must(fhe_keyswitch_create(ctx, lv.tables, (int)l, (int)K_, (int)l, &lv.ks), "keyswitch plan");
        must(fhe_keyswitch_set_plain_modulus(lv.ks, t_), "keyswitch plain modulus");
        must(fhe_baseconv_create(ctx, primes_.data(), (int)l, &t_, 1, &lv.to_plain), "decrypt conversion");
        Big Q;
        for (size_t j = 0; j < l; j++) Q.mul(primes_[j]);
        Q.half();
        for (size_t j = 0; j < l; j++) lv.half_mod_q.push_back(Q.mod(primes_[j]));
        lv.half_mod_t = Q.mod(t_);
        return levels_.emplace(l, std::move(lv)).first->second;
    }

    const phantom::EncryptionParameters &parms() const { return parms_; }
    size_t n() const { return n_; }
    int log_n() const { return log_n_; }
    size_t L() const { return L_; }
    size_t K() const { return K_; }
    uint64_t t() const { return t_; }
    const std::vector<uint64_t> &primes() const { return primes_; }
    fhe_ntt_tables *full_tables() const { return full_; }
    fhe_ntt_tables *plain_tables() const { return plain_; }
    const std::vector<uint32_t> &index_map() const { return index_map_; }
    std::mt19937_64 &rng() const { return rng_; }

    // ---- sampling helpers (host) -> NTT-domain device polynomials over limbs [0, limbs) of `tables`
    // small signed coefficients, the same integer in every limb
    void upload_small(const std::vector<int> &v, phantom::bgv_detail::DevPoly &dst, const std::vector<uint64_t> &ps, fhe_ntt_tables *tab) const
    {
        std::vector<uint64_t> h(ps.size() * n_);
        for (size_t j = 0; j < ps.size(); j++)
            for (size_t i = 0; i < n_; i++) h[j * n_ + i] = v[i] >= 0 ? (uint64_t)v[i] : ps[j] - (uint64_t)(-v[i]);
        dst.upload(h);
        phantom::bgv_detail::must(fhe_ntt_forward_inplace(phantom::detail::engine(), dst.p, tab, ps.size(), 0, nullptr), "ntt");
    }
    std::vector<int> sample_ternary() const
    {
        std::vector<int> v(n_);
        for (auto &x : v) x = (int)(rng_() % 3) - 1;
        return v;
    }
    std::vector<int> sample_error() const
    {
        // centred binomial, variance 10.5 (standard deviation ~3.2, as SEAL/Phantom use)
        std::vector<int> v(n_);
        for (auto &x : v) {
            uint64_t r = rng_();
            x = __builtin_popcountll(r & 0x1FFFFF) - __builtin_popcountll((r >> 21) & 0x1FFFFF);
        }
        return v;
    }
    void sample_uniform(phantom::bgv_detail::DevPoly &dst, const std::vector<uint64_t> &ps) const
    {
        std::vector<uint64_t> h(ps.size() * n_);
        for (size_t j = 0; j < ps.size(); j++) {
            std::uniform_int_distribution<uint64_t> d(0, ps[j] - 1);
            for (size_t i = 0; i < n_; i++) h[j * n_ + i] = d(rng_);
        }
        dst.upload(h);   // uniform residues are uniform in either domain
    }

private:
    uint32_t bitrev(uint64_t x) const
    {
        uint32_t r = 0;
        for (int i = 0; i < log_n_; i++, x >>= 1) r = (r << 1) | (uint32_t)(x & 1);
        return r;
    }
    phantom::EncryptionParameters parms_;
    size_t n_ = 0, L_ = 0, K_ = 0;
    int log_n_ = 0;
    uint64_t t_ = 0;
    std::vector<uint64_t> primes_;
    fhe_ntt_tables *full_ = nullptr, *plain_ = nullptr;
    std::vector<uint32_t> index_map_;
    mutable std::map<size_t, Level> levels_;
    mutable std::mt19937_64 rng_{std::random_device{}()};
};

inline void print_parameters(const PhantomContext &context)
{
    // layout of the block the reference logs at reliability_test/data/bits1-16_num1.txt:4-12
    const auto &cm = context.parms().coeff_modulus();
    size_t total = 0;
    std::cout << "/\n| Encryption parameters :\n|   scheme: BGV\n|   poly_modulus_degree: " << context.n() << "\n|   coeff_modulus size: ";
    std::string parts;
    for (size_t i = 0; i < cm.size(); i++) {
        int bits = 64 - __builtin_clzll(cm[i].value());
        total += bits;
        parts += std::to_string(bits) + (i + 1 < cm.size() ? " + " : "");
    }
    std::cout << total << " (" << parts << ") bits\n\n";
    for (auto &m : cm) std::cout << m.value() << " ,  ";
    std::cout << "\n\n\\\n";
}

template <class T> inline void print_vector(const std::vector<T> &vec, size_t print_size = 4, int /*prec*/ = 3)
{
    const size_t n = vec.size();
    std::cout << std::endl << "    [";
    if (n <= 2 * print_size) {
        for (size_t i = 0; i < n; i++) std::cout << " " << vec[i] << (i + 1 < n ? "," : " ]\n");
    } else {
        for (size_t i = 0; i < print_size; i++) std::cout << " " << vec[i] << ",";
        std::cout << " ...,";
        for (size_t i = n - print_size; i < n; i++) std::cout << " " << vec[i] << (i + 1 < n ? "," : " ]\n");
    }
    std::cout << std::endl;
}

class PhantomBatchEncoder {
public:
    explicit PhantomBatchEncoder(const PhantomContext &c) : n_(c.n()) {}
    size_t slot_count() const { return n_; }
    PhantomPlaintext encode(const PhantomContext &c, const std::vector<uint64_t> &values) const
    {
        using namespace phantom::bgv_detail;
        std::vector<uint64_t> slots(n_, 0);
        for (size_t i = 0; i < values.size() && i < n_; i++) slots[c.index_map()[i]] = values[i] % c.t();
        DevPoly d;
        d.upload(slots);
        must(fhe_ntt_inverse_inplace(phantom::detail::engine(), d.p, c.plain_tables(), 1, 0, nullptr), "encode");
        PhantomPlaintext p;
        p.coeffs = d.download();
        return p;
    }
    std::vector<uint64_t> decode(const PhantomContext &c, const PhantomPlaintext &p) const
    {
        using namespace phantom::bgv_detail;
        DevPoly d;
        d.upload(p.coeffs);
        must(fhe_ntt_forward_inplace(phantom::detail::engine(), d.p, c.plain_tables(), 1, 0, nullptr), "decode");
        std::vector<uint64_t> slots = d.download(), out(n_);
        for (size_t i = 0; i < n_; i++) out[i] = slots[c.index_map()[i]];
        return out;
    }

private:
    size_t n_;
};

class PhantomPublicKey {
public:
    // c0 = b u + t e0 + m,  c1 = a u + t e1   over the L data primes
    void encrypt_asymmetric(const PhantomContext &c, const PhantomPlaintext &plain, PhantomCiphertext &ct) const
    {
        using namespace phantom::bgv_detail;
        auto *ctx = phantom::detail::engine();
        const size_t L = c.L(), N = c.n();
        std::vector<uint64_t> q(c.primes().begin(), c.primes().begin() + L);
        fhe_ntt_tables *tab = c.full_tables();
        ct.resize(2, L, N);
        ct.correction = 1;
        DevPoly u, e, m;
        c.upload_small(c.sample_ternary(), u, q, tab);
        std::vector<int> mm(N);
        for (size_t i = 0; i < N; i++) mm[i] = (int)plain.coeffs[i];
        c.upload_small(mm, m, q, tab);
        std::vector<uint64_t> tmul(L, c.t());
        for (int h = 0; h < 2; h++) {
            c.upload_small(c.sample_error(), e, q, tab);
            must(fhe_scalar_affine(ctx, e.p, e.p, tmul.data(), nullptr, tab, 1, L, 0, nullptr), "t*e");
            must(fhe_modmul(ctx, ct.part(h), (h ? a_ : b_).p, u.p, tab, 1, L, 0, nullptr), "pk*u");
            must(fhe_modadd(ctx, ct.part(h), ct.part(h), e.p, tab, 1, L, 0, nullptr), "+te");
        }
        must(fhe_modadd(ctx, ct.part(0), ct.part(0), m.p, tab, 1, L, 0, nullptr), "+m");
        must(fhe_sync(ctx, nullptr), "sync");
    }
    phantom::bgv_detail::DevPoly b_, a_;   // [L][N] NTT domain
};

class PhantomSecretKey {
public:
    explicit PhantomSecretKey(const PhantomContext &c)
    {
        s_coef_ = c.sample_ternary();
        c.upload_small(s_coef_, s_, c.primes(), c.full_tables());   // all L+K limbs, NTT domain
    }
    PhantomPublicKey gen_publickey(const PhantomContext &c) const
    {
        using namespace phantom::bgv_detail;
        auto *ctx = phantom::detail::engine();
        const size_t L = c.L();
        std::vector<uint64_t> q(c.primes().begin(), c.primes().begin() + L), tmul(L, c.t());
        PhantomPublicKey pk;
        c.sample_uniform(pk.a_, q);
        DevPoly e;
        c.upload_small(c.sample_error(), e, q, c.full_tables());
        must(fhe_scalar_affine(ctx, e.p, e.p, tmul.data(), nullptr, c.full_tables(), 1, L, 0, nullptr), "t*e");
        pk.b_.alloc(L * c.n());
        must(fhe_modmul(ctx, pk.b_.p, pk.a_.p, s_.p, c.full_tables(), 1, L, 0, nullptr), "a*s");
        must(fhe_modsub(ctx, pk.b_.p, e.p, pk.b_.p, c.full_tables(), 1, L, 0, nullptr), "b = te - as");
        must(fhe_sync(ctx, nullptr), "sync");
        return pk;
    }
    // key that switches from s_from (NTT domain, all L+K limbs) to this secret:
    // digit d (one prime):  b_d = -a_d s + t e_d + [limb d only] P s_from,   a_d uniform
    PhantomKSwitchKey make_kswitch_key(const PhantomContext &c, const phantom::bgv_detail::DevPoly &s_from) const
    {
        using namespace phantom::bgv_detail;
        auto *ctx = phantom::detail::engine();
        const size_t L = c.L(), K = c.K(), M = L + K, N = c.n();
        fhe_ntt_tables *tab = c.full_tables();
        std::vector<uint64_t> tmul(M, c.t());
        PhantomKSwitchKey key;
        key.full.alloc(L * 2 * M * N);
        DevPoly a, e, ps(N);
        for (size_t d = 0; d < L; d++) {
            uint64_t *b = key.full.p + (d * 2 + 0) * M * N, *ad = key.full.p + (d * 2 + 1) * M * N;
            c.sample_uniform(a, c.primes());
            must(fhe_d2d(ctx, ad, a.p, M * N * 8, nullptr), "a_d");
            c.upload_small(c.sample_error(), e, c.primes(), tab);
            must(fhe_scalar_affine(ctx, e.p, e.p, tmul.data(), nullptr, tab, 1, M, 0, nullptr), "t*e");
            must(fhe_modmul(ctx, b, ad, s_.p, tab, 1, M, 0, nullptr), "a*s");
            must(fhe_modsub(ctx, b, e.p, b, tab, 1, M, 0, nullptr), "te - as");
            uint64_t pmod = 1;
            for (size_t k = 0; k < K; k++) pmod = mulmod(pmod, c.primes()[L + k] % c.primes()[d], c.primes()[d]);
            must(fhe_scalar_affine(ctx, ps.p, s_from.p + d * N, &pmod, nullptr, tab, 1, 1, d, nullptr), "P*s'");
            must(fhe_modadd(ctx, b + d * N, b + d * N, ps.p, tab, 1, 1, d, nullptr), "+P s'");
            must(fhe_sync(ctx, nullptr), "sync");
        }
        return key;
    }
    PhantomRelinKey gen_relinkey(const PhantomContext &c) const
    {
        using namespace phantom::bgv_detail;
        const size_t M = c.L() + c.K();
        DevPoly s2(M * c.n());
        must(fhe_modmul(phantom::detail::engine(), s2.p, s_.p, s_.p, c.full_tables(), 1, M, 0, nullptr), "s^2");
        PhantomRelinKey rk;
        rk.key = make_kswitch_key(c, s2);
        return rk;
    }
    // keys for the row rotations by powers of two (what dotprod_test.cu:143-148 uses)
    PhantomGaloisKey create_galois_keys(const PhantomContext &c) const
    {
        using namespace phantom::bgv_detail;
        const size_t M = c.L() + c.K(), N = c.n();
        PhantomGaloisKey gk;
        for (size_t step = 1; step < N / 2; step <<= 1) {
            const uint32_t elt = galois_elt_from_step(step, N);
            DevPoly sg(M * N);
            must(fhe_automorphism_ntt(phantom::detail::engine(), sg.p, s_.p, c.log_n(), elt, M, nullptr), "sigma(s)");
            gk.keys.emplace(elt, make_kswitch_key(c, sg));
        }
        return gk;
    }
    static uint32_t galois_elt_from_step(size_t step, size_t n)
    {
        uint64_t elt = 1;
        for (size_t i = 0; i < step; i++) elt = elt * 3 % (2 * n);
        return (uint32_t)elt;
    }
    PhantomPlaintext decrypt(const PhantomContext &c, const PhantomCiphertext &ct) const
    {
        using namespace phantom::bgv_detail;
        auto *ctx = phantom::detail::engine();
        const size_t l = ct.coeff_modulus_size(), N = c.n();
        const PhantomContext::Level &lv = c.level(l);
        fhe_ntt_tables *tab = c.full_tables();   // limbs 0..l-1 of the full set are the data primes of this level
        // y = c0 + c1 s (+ c2 s^2)
        DevPoly y(l * N), tmp(l * N), sp;
        must(fhe_d2d(ctx, y.p, ct.part(0), l * N * 8, nullptr), "c0");
        sp = DevPoly(l * N);
        must(fhe_d2d(ctx, sp.p, s_.p, l * N * 8, nullptr), "s");
        for (size_t i = 1; i < ct.size(); i++) {
            must(fhe_modmul_acc(ctx, y.p, ct.part(i), sp.p, tab, 1, l, 0, nullptr), "c_i s^i");
            if (i + 1 < ct.size()) must(fhe_modmul(ctx, sp.p, sp.p, s_.p, tab, 1, l, 0, nullptr), "s^(i+1)");
        }
        must(fhe_ntt_inverse_inplace(ctx, y.p, tab, l, 0, nullptr), "intt");
        // centre: (x + floor(Q/2)) mod Q, convert to t, subtract floor(Q/2) mod t, undo the mod-switch factor
        must(fhe_scalar_affine(ctx, y.p, y.p, nullptr, lv.half_mod_q.data(), tab, 1, l, 0, nullptr), "+Q/2");
        DevPoly mt(N);
        must(fhe_baseconv_exact(ctx, mt.p, y.p, lv.to_plain, N, nullptr), "mod t");
        std::vector<uint64_t> v = mt.download();
        PhantomPlaintext p;
        p.coeffs.resize(N);
        const uint64_t t = c.t();
        for (size_t i = 0; i < N; i++) p.coeffs[i] = mulmod((v[i] + t - lv.half_mod_t) % t, ct.correction % t, t);
        return p;
    }
    const phantom::bgv_detail::DevPoly &s() const { return s_; }

private:
    std::vector<int> s_coef_;
    phantom::bgv_detail::DevPoly s_;   // [L+K][N] NTT domain
};

namespace phantom {

// level-l view of a top-level key: digits < l, limbs {0..l-1} and the special primes
inline const uint64_t *key_at_level(const PhantomContext &c, const PhantomKSwitchKey &key, size_t l)
{
    using namespace bgv_detail;
    if (l == c.L()) return key.full.p;
    auto it = key.per_level.find(l);
    if (it != key.per_level.end()) return it->second.p;
    const size_t L = c.L(), K = c.K(), M = L + K, Ml = l + K, N = c.n();
    DevPoly out(l * 2 * Ml * N);
    auto *ctx = phantom::detail::engine();
    for (size_t d = 0; d < l; d++)
        for (int h = 0; h < 2; h++) {
            const uint64_t *src = key.full.p + (d * 2 + h) * M * N;
            uint64_t *dst = out.p + (d * 2 + h) * Ml * N;
            must(fhe_d2d(ctx, dst, src, l * N * 8, nullptr), "key limbs");
            must(fhe_d2d(ctx, dst + l * N, src + L * N, K * N * 8, nullptr), "key special limbs");
        }
    must(fhe_sync(ctx, nullptr), "sync");
    return key.per_level.emplace(l, std::move(out)).first->second.p;
}

inline PhantomCiphertext multiply(const PhantomContext &c, const PhantomCiphertext &x, const PhantomCiphertext &y)
{
    using namespace bgv_detail;
    if (x.size() != 2 || y.size() != 2 || x.coeff_modulus_size() != y.coeff_modulus_size()) throw std::invalid_argument("multiply: two fresh ciphertexts of one level");
    auto *ctx = phantom::detail::engine();
    const size_t l = x.coeff_modulus_size();
    fhe_ntt_tables *tab = c.full_tables();
    PhantomCiphertext r;
    r.resize(3, l, c.n());
    // one pass over the four parts: d0 = x0 y0, d1 = x0 y1 + x1 y0, d2 = x1 y1
    must(fhe_tensor_product(ctx, r.part(0), r.part(1), r.part(2), x.part(0), x.part(1), y.part(0), y.part(1), tab, l, 0, nullptr), "tensor product");
    r.correction = mulmod(x.correction, y.correction, c.t());
    return r;
}

inline void relinearize_inplace(const PhantomContext &c, PhantomCiphertext &ct, const PhantomRelinKey &rk)
{
    using namespace bgv_detail;
    if (ct.size() != 3) return;
    auto *ctx = phantom::detail::engine();
    const size_t l = ct.coeff_modulus_size(), N = c.n();
    const PhantomContext::Level &lv = c.level(l);
    PhantomCiphertext r;
    r.resize(2, l, N);
    r.correction = ct.correction;
    // key switch of the degree-2 part; d0 and d1 are added by the key switch's last launch
    must(fhe_relinearize(ctx, lv.ks, r.part(0), r.part(1), ct.part(0), ct.part(1), ct.part(2), key_at_level(c, rk.key, l), nullptr), "relinearize");
    must(fhe_sync(ctx, nullptr), "sync");
    ct = std::move(r);
}

// drop the last data prime: c' = (c - delta) / q_last, delta = c mod q_last, delta = 0 mod t
inline void mod_switch_to_next_inplace(const PhantomContext &c, PhantomCiphertext &ct)
{
    using namespace bgv_detail;
    auto *ctx = phantom::detail::engine();
    const size_t l = ct.coeff_modulus_size(), N = c.n(), last = l - 1;
    if (l < 2) throw std::invalid_argument("no prime left to drop");
    const PhantomContext::Level &lv = c.level(l);
    const uint64_t ql = c.primes()[last], t = c.t();
    PhantomCiphertext r;
    r.resize(ct.size(), last, N);
    // every part in one call: INTT of the last limbs, their residues (times t [. t^-1]: BGV), NTT with the subtraction and
    // the division by q_last riding on its last pass
    must(fhe_rescale(ctx, lv.ks, r.data(), ct.data(), ct.size(), nullptr), "mod switch");
    must(fhe_sync(ctx, nullptr), "sync");
    r.correction = mulmod(ct.correction, ql % t, t);   // plaintext became m q_last^-1: undo at decryption
    ct = std::move(r);
}

inline void rotate_inplace(const PhantomContext &c, PhantomCiphertext &ct, int step, const PhantomGaloisKey &gk)
{
    using namespace bgv_detail;
    if (ct.size() != 2) throw std::invalid_argument("rotate needs a relinearised ciphertext");
    auto *ctx = phantom::detail::engine();
    const size_t l = ct.coeff_modulus_size(), N = c.n();
    const uint32_t elt = PhantomSecretKey::galois_elt_from_step((size_t)step, N);
    auto it = gk.keys.find(elt);
    if (it == gk.keys.end()) throw std::invalid_argument("no Galois key for this step");
    const PhantomContext::Level &lv = c.level(l);
    PhantomCiphertext r;
    r.resize(2, l, N);
    r.correction = ct.correction;
    must(fhe_rotate(ctx, lv.ks, r.part(0), r.part(1), ct.part(0), ct.part(1), elt, key_at_level(c, it->second, l), nullptr), "rotate");
    must(fhe_sync(ctx, nullptr), "sync");
    ct = std::move(r);
}

inline void add_inplace(const PhantomContext &c, PhantomCiphertext &a, const PhantomCiphertext &b)
{
    using namespace bgv_detail;
    if (a.size() != b.size() || a.coeff_modulus_size() != b.coeff_modulus_size() || a.correction != b.correction)
        throw std::invalid_argument("add: mismatched ciphertexts");
    must(fhe_modadd(phantom::detail::engine(), a.data(), a.data(), b.data(), c.full_tables(), a.size(), a.coeff_modulus_size(), 0, nullptr), "add");
}

} // namespace phantom
