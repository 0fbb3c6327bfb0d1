// gen_seal_vectors.cpp -- known-answer vectors for the hot path, produced by REAL Microsoft SEAL.
//
// Why this file exists: the arithmetic of the reference's hot path lives in Microsoft SEAL 3.4.5, which is neither
// vendored in the reference nor installable in the build container, so the repository's oracle
// is "parity unpinned".  This program is the way out: anyone with a SEAL install compiles it against the REAL library,
// runs it once and drops the files it writes into tests/golden/seal/ -- `pytest tests/test_seal_vectors.py` then checks
// BOTH the CPU oracle (oracle/ckks_oracle.c) and the HIP engine (libhefx.so, through the C-ABI) against SEAL's own
// uint64 RNS words, bit for bit, and reports which rescale division (floor / round) that SEAL version uses.
//
//   g++ -O2 -std=c++17 tools/gen_seal_vectors.cpp -o gen_seal_vectors $(pkg-config --cflags --libs seal)   # or:
//   g++ -O2 -std=c++17 tools/gen_seal_vectors.cpp -I<prefix>/include/SEAL-3.4 -L<prefix>/lib -lseal-3.4 -pthread
//   ./gen_seal_vectors tests/golden/seal toy c2            # sets: toy c2 c3 c4 cfg1 rot5 (default: toy c2)
//
// Written against the SEAL 3.4.5 API (the version the reference pins, /root/reference/README.md:6); -DSEAL_API_36
// switches to the 3.6+ spellings (scheme_type::ckks, SEALContext by value, create_*_keys, coeff_modulus_size).
// Keys and encryption randomness come from SEAL's own PRNG: they are INPUTS of the vectors (stored in the file), the
// evaluator results are the known answers.
//
// File format ("HEFXKAT1", little endian; loader: tests/seal_vectors.py):
//   header : char magic[8] = "HEFXKAT1"; u32 version = 1; u32 N; u32 k; u32 reserved; u64 primes[k];
//            char producer[64]  (e.g. "Microsoft SEAL 3.4.5")
//   record*: char tag[24]; u32 kind (1 ciphertext, 2 plaintext, 3 key-switching key); u32 size; u32 rows; u32 aux;
//            f64 scale; u64 nwords; u64 words[nwords]
//     ciphertext : size polys x rows RNS rows x N words, SEAL's layout data[(p*rows + j)*N + i], NTT form
//     plaintext  : size = 1, rows x N words (CKKS plaintexts are in NTT form)
//     key        : size = decomposition count (k-1), rows = k; words = [digit][component 0/1][row][N]
//                  = the concatenation of PublicKey::data() of KSwitchKeys::data()[index]; aux = Galois element (0 = relin)
//     stream (kind 4): the bytes a save() member wrote, padded with zeros to whole words; aux = byte count.  Pins the
//                  SERIALISATION format (include/seal/shim_io.h is "format unpinned" until a real SEAL has written these):
//                  tags parms_stream (EncryptionParameters::Save of the key-level parameters), ct_stream (ct.save),
//                  gk1_stream (a GaloisKeys holding only the key of step 1, .save)
//   tags: inputs  ct, ct_b, pt, gk (one per element, aux = element), rk
//         outputs rot1 = rotate_vector(ct, 1)            rot1_mulpt = multiply_plain(rot1, pt)
//                 rotm1 = rotate_vector(ct, -1)          rot3_naf = rotate_vector(ct, 3) with power-of-two keys only
//                 conj = apply_galois(ct, 2N-1)          mulpt = multiply_plain(ct, pt)        add = add(ct, ct_b)
//                 mul = multiply(ct, ct_b) (size 3)      sq = square(ct) (size 3)              relin = relinearize(mul)
//                 rescale = rescale_to_next(relin)       rescale3 = rescale_to_next(mul)
//                 modsw = mod_switch_to_next(ct)         rot1_low = rotate_vector(modsw, 1)    addpl = add_plain(ct, pt)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "seal/seal.h"

using namespace seal;

#ifdef SEAL_API_36
#define CKKS_SCHEME scheme_type::ckks
#else
#define CKKS_SCHEME scheme_type::CKKS
#endif

#ifndef PRODUCER
#if defined(SEAL_VERSION)
#define PRODUCER "Microsoft SEAL " SEAL_VERSION
#elif defined(SEAL_VERSION_STRING)
#define PRODUCER "Microsoft SEAL " SEAL_VERSION_STRING
#else
#define PRODUCER "Microsoft SEAL (version macro not found; pass -DPRODUCER='\"Microsoft SEAL x.y.z\"')"
#endif
#endif

namespace {

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
        if (std::fwrite(p, 1, n, f) != n) throw std::runtime_error("short write");
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

std::size_t rows_of(const Ciphertext &c)
{
#ifdef SEAL_API_36
    return c.coeff_modulus_size();
#else
    return c.coeff_mod_count();
#endif
}

void put_ct(Writer &w, const std::string &tag, const Ciphertext &c, std::uint32_t aux = 0)
{
    const std::size_t n = c.poly_modulus_degree(), rows = rows_of(c);
    if (!c.is_ntt_form()) throw std::logic_error("CKKS ciphertext not in NTT form");
    w.record(tag, 1, (std::uint32_t)c.size(), (std::uint32_t)rows, aux, c.scale(),
             reinterpret_cast<const std::uint64_t *>(c.data()), (std::uint64_t)c.size() * rows * n);
}

void put_pt(Writer &w, const std::string &tag, const Plaintext &p, std::size_t n)
{
    const std::size_t rows = p.coeff_count() / n;
    w.record(tag, 2, 1, (std::uint32_t)rows, 0, p.scale(), reinterpret_cast<const std::uint64_t *>(p.data()),
             (std::uint64_t)rows * n);
}

// one key-switching key = vector<PublicKey>, each a size-2 key-level ciphertext
void put_key(Writer &w, const std::string &tag, const std::vector<PublicKey> &key, std::uint32_t elt)
{
    std::vector<std::uint64_t> all;
    std::size_t rows = 0;
    for (const PublicKey &pk : key) {
        const Ciphertext &c = pk.data();
        rows = rows_of(c);
        const std::size_t words = c.size() * rows * c.poly_modulus_degree();
        if (c.size() != 2) throw std::logic_error("key component is not a size-2 ciphertext");
        const std::uint64_t *d = reinterpret_cast<const std::uint64_t *>(c.data());
        all.insert(all.end(), d, d + words);
    }
    w.record(tag, 3, (std::uint32_t)key.size(), (std::uint32_t)rows, elt, 1.0, all.data(), all.size());
}

// the bytes of a save() call as a record of kind 4 (aux = byte count)
void put_stream(Writer &w, const std::string &tag, const std::string &bytes)
{
    std::vector<std::uint64_t> words((bytes.size() + 7) / 8, 0);
    std::memcpy(words.data(), bytes.data(), bytes.size());
    w.record(tag, 4, 1, 0, (std::uint32_t)bytes.size(), 1.0, words.data(), words.size());
}

std::uint32_t elt_from_step(int step, std::size_t n)
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

struct Set {
    const char *name;
    std::size_t n;
    std::vector<int> bits;  // empty: BFVDefault(n)
    double scale;  // <= 0: sqrt of the last coeff modulus prime
};

void generate(const std::string &dir, const Set &s)
{
    EncryptionParameters parms(CKKS_SCHEME);
    parms.set_poly_modulus_degree(s.n);
    parms.set_coeff_modulus(s.bits.empty() ? CoeffModulus::BFVDefault(s.n) : CoeffModulus::Create(s.n, s.bits));
#ifdef SEAL_API_36
    SEALContext context_obj(parms);
    SEALContext *context_p = &context_obj;
#define CTX context_obj
    const auto &key_parms = context_p->key_context_data()->parms();
#else
    auto context = SEALContext::Create(parms);
#define CTX context
    const auto &key_parms = context->key_context_data()->parms();
#endif
    const auto &cm = key_parms.coeff_modulus();
    const std::size_t k = cm.size(), n = s.n;
    const double scale = s.scale > 0 ? s.scale : std::sqrt((double)cm.back().value());

    KeyGenerator keygen(CTX);
#ifdef SEAL_API_36
    PublicKey pk;
    keygen.create_public_key(pk);
    RelinKeys rk;
    keygen.create_relin_keys(rk);
    GaloisKeys gk;
    keygen.create_galois_keys(gk);
#else
    PublicKey pk = keygen.public_key();
    RelinKeys rk = keygen.relin_keys();
    GaloisKeys gk = keygen.galois_keys();  // default set: 3^(+-2^i) and 2N-1 -- rotate_vector(ct, 3) must take the NAF path
#endif
    Encryptor encryptor(CTX, pk);
    Evaluator ev(CTX);
    CKKSEncoder encoder(CTX);

    std::vector<double> va(encoder.slot_count()), vb(encoder.slot_count()), vp(encoder.slot_count());
    for (std::size_t i = 0; i < va.size(); ++i) {
        va[i] = 0.001 * (double)(i % 1000) - 0.25;
        vb[i] = 1.0 / (1.0 + (double)(i % 17));
        vp[i] = (double)((i % 5) + 1) * 0.5;
    }
    Plaintext pa, pb, pt;
    encoder.encode(va, scale, pa);
    encoder.encode(vb, scale, pb);
    encoder.encode(vp, scale, pt);
    Ciphertext ct, ct_b;
    encryptor.encrypt(pa, ct);
    encryptor.encrypt(pb, ct_b);

    Writer w(dir + "/seal_" + s.name + ".bin");
    w.raw("HEFXKAT1", 8);
    w.u32(1);
    w.u32((std::uint32_t)n);
    w.u32((std::uint32_t)k);
    w.u32(0);
    for (const auto &q : cm) w.u64(q.value());
    w.fixed(PRODUCER, 64);

    // ---- inputs
    put_ct(w, "ct", ct);
    put_ct(w, "ct_b", ct_b);
    put_pt(w, "pt", pt, n);
    const std::uint32_t e1 = elt_from_step(1, n), em1 = elt_from_step(-1, n), e4 = elt_from_step(4, n),
                        econj = (std::uint32_t)(2 * n - 1);
    for (std::uint32_t e : {e1, em1, e4, econj}) put_key(w, "gk", gk.key(e), e);
    put_key(w, "rk", rk.key(2), 0);

    // ---- serialised forms (SEAL >= 3.5 prefixes a SEALHeader and may compress: compr_mode_type::none keeps the body plain)
    {
        std::ostringstream ps, cs, gs;
#ifdef SEAL_API_36
        key_parms.save(ps, compr_mode_type::none);
        ct.save(cs, compr_mode_type::none);
        GaloisKeys gk1;
        keygen.create_galois_keys(std::vector<int>{1}, gk1);
        gk1.save(gs, compr_mode_type::none);
#else
        EncryptionParameters::Save(key_parms, ps);
        ct.save(cs);
        GaloisKeys gk1 = keygen.galois_keys(std::vector<int>{1});
        gk1.save(gs);
#endif
        put_stream(w, "parms_stream", ps.str());
        put_stream(w, "ct_stream", cs.str());
        put_stream(w, "gk1_stream", gs.str());
    }

    // ---- known answers
    Ciphertext rot1, rotm1, rot3, conj, mulpt, add, mul, sq, relin, rescale, rescale3, modsw, rot1_low, rot1_mulpt, addpl;
    ev.rotate_vector(ct, 1, gk, rot1);
    put_ct(w, "rot1", rot1, e1);
    ev.multiply_plain(rot1, pt, rot1_mulpt);
    put_ct(w, "rot1_mulpt", rot1_mulpt);
    ev.rotate_vector(ct, -1, gk, rotm1);
    put_ct(w, "rotm1", rotm1, em1);
    ev.rotate_vector(ct, 3, gk, rot3);  // NAF(3) = [-1, 4] (least significant term first)
    put_ct(w, "rot3_naf", rot3);
    ev.apply_galois(ct, econj, gk, conj);
    put_ct(w, "conj", conj, econj);
    ev.multiply_plain(ct, pt, mulpt);
    put_ct(w, "mulpt", mulpt);
    ev.add(ct, ct_b, add);
    put_ct(w, "add", add);
    ev.add_plain(ct, pt, addpl);
    put_ct(w, "addpl", addpl);
    ev.multiply(ct, ct_b, mul);
    put_ct(w, "mul", mul);
    ev.square(ct, sq);
    put_ct(w, "sq", sq);
    relin = mul;
    ev.relinearize_inplace(relin, rk);
    put_ct(w, "relin", relin);
    if (k >= 3) {  // at least two data primes
        rescale = relin;
        ev.rescale_to_next_inplace(rescale);
        put_ct(w, "rescale", rescale);
        rescale3 = mul;
        ev.rescale_to_next_inplace(rescale3);
        put_ct(w, "rescale3", rescale3);
        modsw = ct;
        ev.mod_switch_to_next_inplace(modsw);
        put_ct(w, "modsw", modsw);
        ev.rotate_vector(modsw, 1, gk, rot1_low);
        put_ct(w, "rot1_low", rot1_low, e1);
    }
    std::printf("%s/seal_%s.bin: N=%zu k=%zu (%s)\n", dir.c_str(), s.name, n, k, PRODUCER);
}

}  // namespace

int main(int argc, char **argv)
{
    const std::vector<Set> sets = {
        {"toy", 4096, {36, 36, 37}, std::pow(2.0, 30)},                              // small enough to commit
        {"c2", 8192, {60, 40, 40, 60}, std::pow(2.0, 40)},                           // BASELINE config 2
        {"c3", 16384, {60, 40, 40, 40, 40, 60}, std::pow(2.0, 40)},                  // config 3 (headline metric)
        {"c4", 16384, {60, 40, 40, 40, 40, 40, 40, 40, 60}, std::pow(2.0, 40)},      // config 4
        {"cfg1", 8192, {}, -1.0},                                                    // config 1: BFVDefault(8192), scale = sqrt(last prime) (vector_ops.cpp:251)
        {"rot5", 8192, {40, 40, 40, 40, 40}, std::pow(2.0, 40)},                     // 5_rotation.cpp CKKS half
    };
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <output dir> [toy c2 c3 c4 cfg1 rot5 ...]\n", argv[0]);
        return 2;
    }
    std::vector<std::string> want;
    for (int i = 2; i < argc; ++i) want.push_back(argv[i]);
    if (want.empty()) want = {"toy", "c2"};
    try {
        for (const auto &wname : want) {
            bool found = false;
            for (const Set &s : sets)
                if (wname == s.name) {
                    generate(argv[1], s);
                    found = true;
                }
            if (!found) throw std::invalid_argument("unknown set " + wname);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "gen_seal_vectors: %s\n", e.what());
        return 1;
    }
    return 0;
}
