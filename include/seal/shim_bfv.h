// shim_bfv.h -- host-side arithmetic of the BFV half of the seal/seal.h shim.
//
// BFV is NOT on the hot path of the reference (CKKS is); it appears in two drivers whose main() runs a BFV demo before
// the CKKS one (vector_ops.cpp:101-195 bfvOps, 5_rotation.cpp:88-165 bfvRotation).  This header gives the shim what
// those demos need so that the drivers run unchanged end to end.  Division of labour:
//   * everything that is a key switch or an NTT-domain ring operation (key generation, encryption of zero, the
//     relinearisation and rotation key switches, decryption's c0 + c1 s + c2 s^2) runs on the GPU engine through the
//     same hefx_* entry points as CKKS -- a BFV ciphertext is kept in coefficient form, so those calls are bracketed
//     by hefx_ntt_forward / hefx_ntt_inverse; NTTs are exact and linear, so the residues are the ones a
//     coefficient-domain implementation produces;
//   * what is specific to BFV is plain multi-precision integer work on the host: Delta*m scaling, the
//     round(t*x/Q) of decryption, the noise budget, the tensor product scaled by t/Q (exact integers through an
//     auxiliary RNS basis on the GPU, CRT-composed here), and BatchEncoder's NTT modulo the plain modulus.
// It follows the textbook BFV definition (Fan-Vercauteren with Delta = floor(Q/t)); SEAL's BEHZ RNS variant computes the
// same rounded quantities up to its documented approximation error in the noise, so decrypted results agree while
// ciphertext bits need not -- BFV is outside the bit-exact contract.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace seal {
namespace shim {
namespace bfv {

typedef unsigned __int128 u128;

// ---- fixed-width unsigned integers, 8 x 64 bits (every quantity here is below 2^460)
struct Big {
    static constexpr int W = 8;
    std::uint64_t w[W];
    Big() { std::memset(w, 0, sizeof w); }
    explicit Big(std::uint64_t v)
    {
        std::memset(w, 0, sizeof w);
        w[0] = v;
    }
    bool is_zero() const
    {
        for (int i = 0; i < W; ++i)
            if (w[i]) return false;
        return true;
    }
    int bits() const
    {
        for (int i = W - 1; i >= 0; --i)
            if (w[i]) return 64 * i + 64 - __builtin_clzll(w[i]);
        return 0;
    }
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1; }
};
inline int cmp(const Big &a, const Big &b)
{
    for (int i = Big::W - 1; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1;
    return 0;
}
inline Big add(const Big &a, const Big &b)
{
    Big r;
    u128 c = 0;
    for (int i = 0; i < Big::W; ++i) {
        c += (u128)a.w[i] + b.w[i];
        r.w[i] = (std::uint64_t)c;
        c >>= 64;
    }
    return r;
}
inline Big sub(const Big &a, const Big &b)  // a >= b
{
    Big r;
    std::uint64_t borrow = 0;
    for (int i = 0; i < Big::W; ++i) {
        const std::uint64_t t = a.w[i] - b.w[i], b1 = a.w[i] < b.w[i], t2 = t - borrow, b2 = t < borrow;
        r.w[i] = t2;
        borrow = b1 | b2;
    }
    return r;
}
inline Big mul_small(const Big &a, std::uint64_t m)
{
    Big r;
    u128 c = 0;
    for (int i = 0; i < Big::W; ++i) {
        c += (u128)a.w[i] * m;
        r.w[i] = (std::uint64_t)c;
        c >>= 64;
    }
    return r;
}
inline std::uint64_t mod_small(const Big &a, std::uint64_t m)
{
    u128 r = 0;
    for (int i = Big::W - 1; i >= 0; --i) r = ((r << 64) | a.w[i]) % m;
    return (std::uint64_t)r;
}
inline Big shl1(const Big &a)
{
    Big r;
    std::uint64_t c = 0;
    for (int i = 0; i < Big::W; ++i) {
        r.w[i] = (a.w[i] << 1) | c;
        c = a.w[i] >> 63;
    }
    return r;
}
inline Big shr1(const Big &a)
{
    Big r;
    for (int i = 0; i < Big::W; ++i) r.w[i] = (a.w[i] >> 1) | (i + 1 < Big::W ? a.w[i + 1] << 63 : 0);
    return r;
}
// binary long division: num = quo * den + rem
inline void divrem(const Big &num, const Big &den, Big &quo, Big &rem)
{
    quo = Big();
    rem = Big();
    for (int i = num.bits() - 1; i >= 0; --i) {
        rem = shl1(rem);
        if (num.bit(i)) rem.w[0] |= 1;
        if (cmp(rem, den) >= 0) {
            rem = sub(rem, den);
            quo.w[i >> 6] |= (std::uint64_t)1 << (i & 63);
        }
    }
}

inline std::uint64_t mulmod64(std::uint64_t a, std::uint64_t b, std::uint64_t q) { return (std::uint64_t)(((u128)a * b) % q); }
inline std::uint64_t powmod64(std::uint64_t a, std::uint64_t e, std::uint64_t q)
{
    std::uint64_t r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod64(r, a, q);
        a = mulmod64(a, a, q);
    }
    return r;
}

// ---- an RNS basis with its CRT composition: x = sum_j [r_j * (M/m_j)^-1]_{m_j} * (M/m_j)  mod M
struct Basis {
    std::vector<std::uint64_t> m;
    Big M, half;                      // product and floor(M/2)
    std::vector<Big> punct;           // M / m_j
    std::vector<std::uint64_t> inv;   // (M / m_j)^-1 mod m_j
    void init(const std::vector<std::uint64_t> &moduli)
    {
        m = moduli;
        M = Big(1);
        for (auto q : m) M = mul_small(M, q);
        half = shr1(M);
        punct.clear();
        inv.clear();
        for (std::size_t j = 0; j < m.size(); ++j) {
            Big p(1);
            for (std::size_t i = 0; i < m.size(); ++i)
                if (i != j) p = mul_small(p, m[i]);
            punct.push_back(p);
            inv.push_back(powmod64(mod_small(p, m[j]), m[j] - 2, m[j]));
        }
    }
    // residues r[j] (row stride `stride`, element `idx`) -> the integer in [0, M)
    Big compose(const std::uint64_t *rows, std::size_t stride, std::size_t idx) const
    {
        Big x;
        for (std::size_t j = 0; j < m.size(); ++j) {
            x = add(x, mul_small(punct[j], mulmod64(rows[j * stride + idx], inv[j], m[j])));
            // each addend is < M and there are at most 8 of them: reduce lazily, M < 2^450 leaves the headroom
        }
        while (cmp(x, M) >= 0) x = sub(x, M);  // at most m.size() - 1 times
        return x;
    }
};

// ---- negacyclic NTT modulo the plain modulus t (BatchEncoder); same conventions as the engine's: psi minimal primitive
// 2N-th root, forward = natural in -> bit-reversed out, out[bitrev(j)] = a(psi^(2j+1))
struct PlainNtt {
    std::uint64_t t = 0;
    int logn = 0;
    std::vector<std::uint64_t> tw, itw;  // tw[bitrev(i)] = psi^i
    std::uint64_t ninv = 0;
    static std::uint32_t bitrev(std::uint32_t x, int bits)
    {
        std::uint32_t r = 0;
        for (int i = 0; i < bits; ++i) r = (r << 1) | ((x >> i) & 1);
        return r;
    }
    bool init(std::uint64_t t_, std::size_t n)
    {
        t = t_;
        logn = 0;
        while (((std::size_t)1 << logn) < n) ++logn;
        const std::uint64_t two_n = 2 * n;
        if (t < 2 || (t - 1) % two_n) return false;
        std::uint64_t root = 0;
        for (std::uint64_t g = 2; g < 2000 && !root; ++g) {
            const std::uint64_t c = powmod64(g, (t - 1) / two_n, t);
            if (powmod64(c, n, t) == t - 1) root = c;
        }
        if (!root) return false;
        const std::uint64_t sq = mulmod64(root, root, t);  // minimal root, like the engine (any one works for batching)
        std::uint64_t best = root, cur = root;
        for (std::size_t i = 0; i < n; ++i) {
            if (cur < best) best = cur;
            cur = mulmod64(cur, sq, t);
        }
        const std::uint64_t psi = best, ipsi = powmod64(psi, t - 2, t);
        tw.assign(n, 0);
        itw.assign(n, 0);
        std::uint64_t p = 1, ip = 1;
        for (std::size_t i = 0; i < n; ++i) {
            tw[bitrev((std::uint32_t)i, logn)] = p;
            itw[bitrev((std::uint32_t)i, logn)] = ip;
            p = mulmod64(p, psi, t);
            ip = mulmod64(ip, ipsi, t);
        }
        ninv = powmod64(n % t, t - 2, t);
        return true;
    }
    void forward(std::vector<std::uint64_t> &a) const  // Cooley-Tukey
    {
        const std::size_t n = a.size();
        std::size_t len = n / 2, m = 1;
        for (; m < n; m <<= 1, len >>= 1)
            for (std::size_t i = 0; i < m; ++i) {
                const std::uint64_t w = tw[m + i];
                for (std::size_t j = 2 * i * len; j < 2 * i * len + len; ++j) {
                    const std::uint64_t u = a[j], v = mulmod64(a[j + len], w, t);
                    a[j] = (u + v) % t;
                    a[j + len] = (u + t - v) % t;
                }
            }
    }
    void inverse(std::vector<std::uint64_t> &a) const  // Gentleman-Sande
    {
        const std::size_t n = a.size();
        std::size_t len = 1, m = n / 2;
        for (; m >= 1; m >>= 1, len <<= 1)
            for (std::size_t i = 0; i < m; ++i) {
                const std::uint64_t w = itw[m + i];
                for (std::size_t j = 2 * i * len; j < 2 * i * len + len; ++j) {
                    const std::uint64_t u = a[j], v = a[j + len];
                    a[j] = (u + v) % t;
                    a[j + len] = mulmod64((u + t - v) % t, w, t);
                }
            }
        for (auto &x : a) x = mulmod64(x, ninv, t);
    }
};

}  // namespace bfv
}  // namespace shim
}  // namespace seal
