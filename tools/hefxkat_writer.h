// hefxkat_writer.h -- writer of the "HEFXKAT1" known-answer files (format: tools/gen_seal_vectors.cpp; loader:
// tests/seal_vectors.py), shared by tools/gen_composite_vectors.cpp and drivers/xcheck_lr.cpp.  Uses only the public
// SEAL 3.4.5 API, so it compiles against Microsoft SEAL and against include/seal/seal.h alike.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "seal/seal.h"

namespace hefxkat {
using namespace seal;

struct Writer {
    FILE *f;
    explicit Writer(const std::string &path) : f(std::fopen(path.c_str(), "wb"))
    {
        if (!f) throw std::runtime_error("cannot open " + path);
    }
    ~Writer()
    {
        if (f) std::fclose(f);
    }
    void raw(const void *p, std::size_t n)
    {
        if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error("short write");
    }
    void u32(std::uint32_t v) { raw(&v, 4); }
    void u64(std::uint64_t v) { raw(&v, 8); }
    void f64(double v) { raw(&v, 8); }
    void fixed(const std::string &s, std::size_t n)
    {
        std::vector<char> b(n, 0);
        std::memcpy(b.data(), s.data(), s.size() < n ? s.size() : n - 1);
        raw(b.data(), n);
    }
    void record(const std::string &tag, std::uint32_t kind, std::uint32_t size, std::uint32_t rows, std::uint32_t aux,
                double scale, const std::uint64_t *w, std::uint64_t nwords)
    {
        fixed(tag, 24);
        u32(kind);
        u32(size);
        u32(rows);
        u32(aux);
        f64(scale);
        u64(nwords);
        raw(w, nwords * 8);
    }
};

inline void put_ct(Writer &w, const std::string &tag, const Ciphertext &c, std::uint32_t aux = 0)
{
    const std::size_t n = c.poly_modulus_degree(), rows = c.coeff_mod_count();
    if (!c.is_ntt_form()) throw std::logic_error("CKKS ciphertext not in NTT form");
    w.record(tag, 1, (std::uint32_t)c.size(), (std::uint32_t)rows, aux, c.scale(),
             reinterpret_cast<const std::uint64_t *>(c.data()), (std::uint64_t)c.size() * rows * n);
}

inline void put_pt(Writer &w, const std::string &tag, const Plaintext &p, std::size_t n, std::uint32_t aux = 0)
{
    const std::size_t rows = p.coeff_count() / n;
    w.record(tag, 2, 1, (std::uint32_t)rows, aux, p.scale(), reinterpret_cast<const std::uint64_t *>(p.data()),
             (std::uint64_t)rows * n);
}

inline void put_key(Writer &w, const std::string &tag, const std::vector<PublicKey> &key, std::uint32_t elt)
{
    std::vector<std::uint64_t> all;
    std::size_t rows = 0;
    for (const PublicKey &pk : key) {
        const Ciphertext &c = pk.data();
        rows = c.coeff_mod_count();
        if (c.size() != 2) throw std::logic_error("key component is not a size-2 ciphertext");
        const std::uint64_t *d = reinterpret_cast<const std::uint64_t *>(c.data());
        all.insert(all.end(), d, d + c.size() * rows * c.poly_modulus_degree());
    }
    w.record(tag, 3, (std::uint32_t)key.size(), (std::uint32_t)rows, elt, 1.0, all.data(), all.size());
}

inline std::uint32_t elt_from_step(long long step, std::size_t n)
{
    const std::uint64_t m = 2 * n;
    std::uint64_t pos = step > 0 ? (std::uint64_t)step : (std::uint64_t)((long long)(n / 2) + step);
    std::uint64_t r = 1, b = 3;
    for (; pos; pos >>= 1) {
        if (pos & 1) r = (r * b) & (m - 1);
        b = (b * b) & (m - 1);
    }
    return (std::uint32_t)r;
}

// file header for the key-level parameters of `context`
inline void put_header(Writer &w, const std::shared_ptr<SEALContext> &context, const char *producer)
{
    const auto &parms = context->key_context_data()->parms();
    const auto &cm = parms.coeff_modulus();
    w.raw("HEFXKAT1", 8);
    w.u32(1);
    w.u32((std::uint32_t)parms.poly_modulus_degree());
    w.u32((std::uint32_t)cm.size());
    w.u32(0);
    for (const auto &q : cm) w.u64(q.value());
    w.fixed(producer, 64);
}

}  // namespace hefxkat
