// phantom_shim.hpp -- the handful of Phantom-FHE names the reference's NTT harnesses use
// (reliability_test/ntt_test.cu, ntt_real_test.cu), implemented over the C ABI of
// libfhe_mi355x.so (fhe_mi355x.h).  Symbol list: SURVEY.md section 8 b2, i.e. what
// `nm -D reliability_test/build/ntt_test` imports from libPhantom.so plus the
// header-inline types those translation units instantiate:
//
//   phantom::util::cuda_stream_wrapper{ctor, get_stream}      ntt_test.cu:40-41
//   phantom::util::make_cuda_auto_ptr<T>(n, stream), .get()   ntt_test.cu:47,88
//   phantom::arith::Modulus{value, const_ratio}               ntt_test.cu:49-53
//   phantom::arith::CoeffModulus::Create(N, {bits})           ntt_test.cu:44
//   phantom::arith::NTT(log_n, Modulus), get_from_root_powers[_shoup]   ntt_test.cu:60-64
//   DModulus::set(value, ratio0, ratio1)                      ntt_test.cu:49-53
//   DNTTTable::init(n, size, stream) / set(...)               ntt_test.cu:57-69
//   nwt_2d_radix8_forward_inplace(data, table, size, start, stream)     ntt_test.cu:95,144
//
// HIP only: streams are hipStream_t.  A harness written against Phantom keeps its
// structure; only the runtime calls change (cudaMemcpyAsync -> hipMemcpyAsync, ...),
// see INTEGRATION.md.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "fhe_mi355x.h"

namespace phantom {

namespace detail {
inline void must(int rc, const char *what)
{
    if (rc != FHE_OK) throw std::runtime_error(std::string(what) + ": " + fhe_last_error());
}
// one engine context per process, device 0 (the harnesses are single-GPU, ntt_test.cu:40)
inline fhe_ctx *engine()
{
    static fhe_ctx *ctx = [] {
        fhe_ctx *c = nullptr;
        must(fhe_ctx_create(0, &c), "fhe_ctx_create");
        return c;
    }();
    return ctx;
}
} // namespace detail

namespace arith {

class Modulus {
public:
    Modulus() = default;
    explicit Modulus(uint64_t v) { set_value(v); }
    void set_value(uint64_t v)
    {
        value_ = v;
        uint64_t r[3];
        detail::must(fhe_modulus_const_ratio(v, r), "fhe_modulus_const_ratio");
        ratio_ = {r[0], r[1], r[2]};
    }
    uint64_t value() const { return value_; }
    const std::array<uint64_t, 3> &const_ratio() const { return ratio_; }

private:
    uint64_t value_ = 0;
    std::array<uint64_t, 3> ratio_{};
};

struct CoeffModulus {
    static std::vector<Modulus> Create(size_t poly_modulus_degree, const std::vector<int> &bit_sizes)
    {
        std::vector<uint64_t> q(bit_sizes.size());
        detail::must(fhe_moduli_create(poly_modulus_degree, bit_sizes.data(), (int)bit_sizes.size(), q.data()),
                     "CoeffModulus::Create");
        std::vector<Modulus> out;
        for (uint64_t v : q) out.emplace_back(v);
        return out;
    }
};

class NTT {
public:
    NTT(int log_n, const Modulus &m) : rp_((size_t)1 << log_n), rps_((size_t)1 << log_n)
    {
        detail::must(fhe_root_powers(m.value(), log_n, rp_.data(), rps_.data()), "NTT tables");
    }
    const std::vector<uint64_t> &get_from_root_powers() const { return rp_; }
    const std::vector<uint64_t> &get_from_root_powers_shoup() const { return rps_; }

private:
    std::vector<uint64_t> rp_, rps_;
};

} // namespace arith

namespace util {

class cuda_stream_wrapper {
public:
    cuda_stream_wrapper()
    {
        void *s = nullptr;
        detail::must(fhe_ctx_stream(detail::engine(), &s), "fhe_ctx_stream");
        stream_ = static_cast<hipStream_t>(s);
    }
    const hipStream_t &get_stream() const { return stream_; }

private:
    hipStream_t stream_ = nullptr;
};

// Device buffer with RAII release.  The harness WRITES DModulus entries through the
// pointer from the host (ntt_test.cu:48-54), so non-uint64_t payloads live in pinned
// host memory that the device can read; raw residue buffers are plain device memory.
template <class T> class cuda_auto_ptr {
public:
    cuda_auto_ptr() = default;
    cuda_auto_ptr(T *p, bool pinned) : p_(p), pinned_(pinned) {}
    cuda_auto_ptr(cuda_auto_ptr &&o) noexcept : p_(o.p_), pinned_(o.pinned_) { o.p_ = nullptr; }
    cuda_auto_ptr &operator=(cuda_auto_ptr &&o) noexcept
    {
        reset();
        p_ = o.p_;
        pinned_ = o.pinned_;
        o.p_ = nullptr;
        return *this;
    }
    cuda_auto_ptr(const cuda_auto_ptr &) = delete;
    cuda_auto_ptr &operator=(const cuda_auto_ptr &) = delete;
    ~cuda_auto_ptr() { reset(); }
    T *get() const { return p_; }

private:
    void reset()
    {
        if (!p_) return;
        if (pinned_) (void)hipHostFree(p_);
        else (void)fhe_free(detail::engine(), p_);
        p_ = nullptr;
    }
    T *p_ = nullptr;
    bool pinned_ = false;
};

template <class T> cuda_auto_ptr<T> make_cuda_auto_ptr(size_t n, const hipStream_t &)
{
    if constexpr (std::is_same<T, uint64_t>::value) {
        void *p = nullptr;
        detail::must(fhe_alloc(detail::engine(), n * sizeof(T), &p), "make_cuda_auto_ptr");
        return cuda_auto_ptr<T>(static_cast<T *>(p), false);
    } else {
        void *p = nullptr;
        if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
        return cuda_auto_ptr<T>(static_cast<T *>(p), true);
    }
}

} // namespace util
} // namespace phantom

// Phantom keeps these two in the global namespace.
struct DModulus {
    void set(uint64_t value, uint64_t ratio0, uint64_t ratio1)
    {
        value_ = value;
        const_ratio_[0] = ratio0;
        const_ratio_[1] = ratio1;
    }
    uint64_t value() const { return value_; }
    const uint64_t *const_ratio() const { return const_ratio_; }

private:
    uint64_t value_ = 0, const_ratio_[2] = {0, 0};
};

class DNTTTable {
public:
    DNTTTable() = default;
    DNTTTable(const DNTTTable &) = delete;
    DNTTTable &operator=(const DNTTTable &) = delete;
    ~DNTTTable()
    {
        if (tables_) fhe_ntt_tables_destroy(tables_);
    }
    void init(size_t n, size_t size, const hipStream_t &)
    {
        n_ = n;
        size_ = size;
        log_n_ = 0;
        while (((size_t)1 << log_n_) < n) log_n_++;
        q_.assign(size, 0);
        roots_.assign(size * n, 0);
        if (tables_) fhe_ntt_tables_destroy(tables_);
        tables_ = nullptr;
    }
    // twiddle_shoup / inverse tables / n_inv are recomputed by the engine in its own
    // layout; the forward root powers and the modulus define the transform.
    void set(const DModulus *modulus, const uint64_t *twiddle, const uint64_t * /*twiddle_shoup*/,
             const uint64_t * /*itwiddle*/, const uint64_t * /*itwiddle_shoup*/, uint64_t /*n_inv*/,
             uint64_t /*n_inv_shoup*/, size_t index, const hipStream_t &) const
    {
        if (index >= size_) throw std::out_of_range("DNTTTable::set index");
        q_[index] = modulus->value();
        std::copy(twiddle, twiddle + n_, roots_.begin() + index * n_);
        if (tables_) {
            fhe_ntt_tables_destroy(tables_);
            tables_ = nullptr;
        }
    }
    size_t n() const { return n_; }
    size_t size() const { return size_; }
    // device tables are built on first use
    fhe_ntt_tables *handle() const
    {
        if (!tables_)
            phantom::detail::must(fhe_ntt_tables_create_from_roots(phantom::detail::engine(), log_n_, q_.data(), (int)size_,
                                                                   roots_.data(), -1, &tables_),
                                  "DNTTTable");
        return tables_;
    }

private:
    size_t n_ = 0, size_ = 0;
    int log_n_ = 0;
    mutable std::vector<uint64_t> q_, roots_;
    mutable fhe_ntt_tables *tables_ = nullptr;
};

inline void nwt_2d_radix8_forward_inplace(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                          size_t start_modulus_idx, const hipStream_t &stream)
{
    phantom::detail::must(fhe_ntt_forward_inplace(phantom::detail::engine(), inout, ntt_tables.handle(), coeff_modulus_size,
                                                  start_modulus_idx, stream),
                          "nwt_2d_radix8_forward_inplace");
}

inline void nwt_2d_radix8_backward_inplace(uint64_t *inout, const DNTTTable &ntt_tables, size_t coeff_modulus_size,
                                           size_t start_modulus_idx, const hipStream_t &stream)
{
    phantom::detail::must(fhe_ntt_inverse_inplace(phantom::detail::engine(), inout, ntt_tables.handle(), coeff_modulus_size,
                                                  start_modulus_idx, stream),
                          "nwt_2d_radix8_backward_inplace");
}
