// shim_io.h -- what SEAL's save / load members need below the classes of seal/seal.h: SHA3-256 (SEAL's parms_id is the
// SHA3-256 of the parameter words, util/hash.h) and little-endian stream helpers.
//
// FORMAT UNPINNED: the byte layouts written by the save() members of seal/seal.h restate SEAL 3.4.5's uncompressed
// streams (3.4.x has no compression and no SEALHeader; those arrived with 3.5) from its published sources -- no SEAL
// binary was available to check a single byte (SEAL is neither vendored by the reference nor installable offline;
// DESIGN.md section 2).  tools/gen_seal_vectors.cpp also saves one ciphertext and one Galois key, so that a box with a
// real SEAL pins the format together with the arithmetic.  The reference itself never calls save / load
// (/root/reference/CMakeLists.txt:7-23 lists every target): this is SURVEY 8(f) rank 4, completeness of the class surface.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace seal {
namespace shim {

// Keccak-f[1600] and SHA3-256 (FIPS 202: rate 136 bytes, domain byte 0x06); known answers in drivers/shim_selftest.cpp
// and tests/test_shim_host_cpu.py (against Python's hashlib)
inline void keccak_f1600(std::uint64_t st[25])
{
    static const std::uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int PIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    auto rotl = [](std::uint64_t x, int s) { return (x << s) | (x >> (64 - s)); };
    for (int round = 0; round < 24; ++round) {
        std::uint64_t bc[5], t;
        for (int i = 0; i < 5; ++i) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; ++i) {
            t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; ++i) {
            const int j = PIL[i];
            bc[0] = st[j];
            st[j] = rotl(t, ROT[i]);
            t = bc[0];
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; ++i) bc[i] = st[j + i];
            for (int i = 0; i < 5; ++i) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}

inline std::array<std::uint8_t, 32> sha3_256(const std::uint8_t *in, std::size_t len)
{
    constexpr std::size_t RATE = 136;
    std::uint64_t st[25] = {};
    std::uint8_t block[RATE];
    auto absorb = [&](const std::uint8_t *b) {
        for (std::size_t i = 0; i < RATE / 8; ++i) {
            std::uint64_t w = 0;
            for (int k = 0; k < 8; ++k) w |= (std::uint64_t)b[8 * i + k] << (8 * k);
            st[i] ^= w;
        }
        keccak_f1600(st);
    };
    while (len >= RATE) {
        absorb(in);
        in += RATE;
        len -= RATE;
    }
    std::memset(block, 0, RATE);
    std::memcpy(block, in, len);
    block[len] ^= 0x06;
    block[RATE - 1] ^= 0x80;
    absorb(block);
    std::array<std::uint8_t, 32> out{};
    for (int i = 0; i < 32; ++i) out[i] = (std::uint8_t)(st[i / 8] >> (8 * (i % 8)));
    return out;
}

// SEAL's HashFunction::sha3_hash: the digest of `count` uint64 words (their little-endian bytes) as four uint64 words
inline std::array<std::uint64_t, 4> sha3_words(const std::vector<std::uint64_t> &words)
{
    std::vector<std::uint8_t> bytes(words.size() * 8);
    for (std::size_t i = 0; i < words.size(); ++i)
        for (int k = 0; k < 8; ++k) bytes[8 * i + k] = (std::uint8_t)(words[i] >> (8 * k));
    const auto d = sha3_256(bytes.data(), bytes.size());
    std::array<std::uint64_t, 4> out{};
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 8; ++k) out[i] |= (std::uint64_t)d[8 * i + k] << (8 * k);
    return out;
}

// ---- stream helpers: SEAL writes native little-endian PODs with ostream::write and throws on failbit / badbit
struct StreamGuard {  // exceptions on for the duration of a save / load, the caller's mask restored afterwards
    std::ios &s;
    std::ios::iostate old;
    explicit StreamGuard(std::ios &st) : s(st), old(st.exceptions()) { s.exceptions(std::ios_base::badbit | std::ios_base::failbit); }
    ~StreamGuard()
    {
        try {
            s.exceptions(old);
        } catch (...) {
        }
    }
};
inline void put_u64(std::ostream &o, std::uint64_t v)
{
    char b[8];
    for (int k = 0; k < 8; ++k) b[k] = (char)(v >> (8 * k));
    o.write(b, 8);
}
inline std::uint64_t get_u64(std::istream &i)
{
    unsigned char b[8];
    i.read(reinterpret_cast<char *>(b), 8);
    std::uint64_t v = 0;
    for (int k = 0; k < 8; ++k) v |= (std::uint64_t)b[k] << (8 * k);
    return v;
}
inline void put_u8(std::ostream &o, std::uint8_t v) { o.write(reinterpret_cast<const char *>(&v), 1); }
inline std::uint8_t get_u8(std::istream &i)
{
    char c;
    i.read(&c, 1);
    return (std::uint8_t)c;
}
inline void put_f64(std::ostream &o, double d)
{
    std::uint64_t v;
    std::memcpy(&v, &d, 8);
    put_u64(o, v);
}
inline double get_f64(std::istream &i)
{
    const std::uint64_t v = get_u64(i);
    double d;
    std::memcpy(&d, &v, 8);
    return d;
}
inline void put_id(std::ostream &o, const std::array<std::uint64_t, 4> &id)
{
    for (auto w : id) put_u64(o, w);
}
inline std::array<std::uint64_t, 4> get_id(std::istream &i)
{
    std::array<std::uint64_t, 4> id{};
    for (auto &w : id) w = get_u64(i);
    return id;
}
// SEAL's IntArray<uint64_t>::save / load: the element count as uint64, then the words
inline void put_words(std::ostream &o, const std::uint64_t *w, std::size_t n)
{
    put_u64(o, (std::uint64_t)n);
    std::vector<char> b(n * 8);
    for (std::size_t i = 0; i < n; ++i)
        for (int k = 0; k < 8; ++k) b[8 * i + k] = (char)(w[i] >> (8 * k));
    if (n) o.write(b.data(), (std::streamsize)b.size());
}
inline std::vector<std::uint64_t> get_words(std::istream &i, std::uint64_t max_words)
{
    const std::uint64_t n = get_u64(i);
    if (n > max_words) throw std::invalid_argument("loaded data is too large for the encryption parameters");
    std::vector<unsigned char> b((std::size_t)n * 8);
    if (n) i.read(reinterpret_cast<char *>(b.data()), (std::streamsize)b.size());
    std::vector<std::uint64_t> w((std::size_t)n, 0);
    for (std::size_t j = 0; j < w.size(); ++j)
        for (int k = 0; k < 8; ++k) w[j] |= (std::uint64_t)b[8 * j + k] << (8 * k);
    return w;
}

}  // namespace shim
}  // namespace seal
