// xcheck_lr.cpp -- the reference's OWN logistic-regression composites as known answers, for
// tests/test_gpu_xcheck.py.  Companion of tools/gen_composite_vectors.cpp (rows a1-a8), same file format; this one is
// for THIS repository's shim only, because Tree_cipher, Horner_cipher and predict_cipher_weights encrypt a constant
// INSIDE the function (logistic_regression_ckks.cpp:102, :162): the result is a function of recordable inputs only if
// the Encryptor's randomness is known.  With SEAL_SHIM_SEED set, include/seal/seal.h draws every sampler key from
// std::mt19937_64(seed) in construction order -- KeyGenerator first, then each Encryptor -- and numbers an Encryptor's
// streams 1, 2, ...; the test re-derives the keys from the seed and hands them to seal.py's Encryptor, whose stream
// numbering is the same.  The program refuses to run without the seed.
//
// It #includes /root/reference/logistic_regression_ckks.cpp (which includes helper.h) with main() renamed: the
// composition under test is the reference's source, compiled from where it lies.
//
// Records of lr_c4.bin (N = 16384, {60, 40 x 7, 60}, scale 2^40: config 4's parameters):
//   pk (a size-2 key-level ciphertext), rk, gk for the steps 1 and -8 (the rotations of cipher_dot_product at size 8; the
//   program generates the reference's default key set and records the two keys a replay needs)
//   poly_ct; tree = Tree_cipher(poly_ct, 3, ...)   [Encryptor 1, stream 1]
//            horner = Horner_cipher(poly_ct, 3, ...) [Encryptor 2, stream 1]
//   feat[6], weights; predict = predict_cipher_weights(feat, weights, 8, ...)   [Encryptor 3, stream 1]
// Sampler-key order under SEAL_SHIM_SEED: draw 0 KeyGenerator, 1 the Encryptor of the inputs, 2..4 Encryptors 1..3.
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "seal/seal.h"
#include "hefxkat_writer.h"

#define main reference_logistic_regression_main
#include "logistic_regression_ckks.cpp"  // the reference's file, from -I<reference>
#undef main

using namespace hefxkat;

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: SEAL_SHIM_SEED=<n> %s <output dir>\n", argv[0]);
        return 2;
    }
    if (!std::getenv("SEAL_SHIM_SEED")) {
        std::fprintf(stderr, "xcheck_lr: SEAL_SHIM_SEED is not set (the replay needs the Encryptors' randomness)\n");
        return 2;
    }
    try {
        const std::size_t n = POLY_MOD_DEGREE;
        EncryptionParameters params(scheme_type::CKKS);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::Create(n, {60, 40, 40, 40, 40, 40, 40, 40, 60}));  // logistic_regression_ckks.cpp:421
        auto context = SEALContext::Create(params);
        const double scale = std::pow(2.0, 40);
        KeyGenerator keygen(context);  // sampler key: draw 0
        PublicKey pk = keygen.public_key();
        RelinKeys rk = keygen.relin_keys();
        GaloisKeys gk = keygen.galois_keys();
        Encryptor enc_inputs(context, pk);  // draw 1
        Evaluator evaluator(context);
        CKKSEncoder encoder(context);
        Writer w(std::string(argv[1]) + "/lr_c4.bin");
        put_header(w, context, "hefx seal.h shim of this repository (NOT Microsoft SEAL: pins nothing)");
        put_ct(w, "pk", pk.data());
        put_key(w, "rk", rk.key(2), 0);
        for (int step : {1, -8}) {
            const std::uint32_t e = elt_from_step(step, n);
            put_key(w, "gk", gk.key(e), e);
        }
        const std::vector<double> coeffs = {0.5, 1.20069, 0.00001, -0.81562};  // :247 (DEGREE 3)
        {
            std::vector<double> x(8);
            for (int i = 0; i < 8; ++i) x[i] = -1.0 + 2.0 * i / 7.0;
            Plaintext px;
            encoder.encode(x, scale, px);
            Ciphertext cx;
            enc_inputs.encrypt(px, cx);
            put_ct(w, "poly_ct", cx);
            Encryptor e1(context, pk);  // draw 2
            put_ct(w, "tree", Tree_cipher(cx, 3, scale, coeffs, encoder, evaluator, e1, rk, params));
            Encryptor e2(context, pk);  // draw 3
            put_ct(w, "horner", Horner_cipher(cx, 3, coeffs, encoder, scale, evaluator, e2, rk, params));
        }
        {
            const int rows = 6, nw = 8;
            std::vector<Ciphertext> features(rows);
            std::uint64_t s = 0x5EA1C0DEull;
            auto next = [&]() {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                return (double)((s >> 11) & ((1ull << 40) - 1)) / (double)(1ull << 39) - 1.0;
            };
            for (int i = 0; i < rows; ++i) {
                std::vector<double> f(nw);
                for (double &x : f) x = next();
                Plaintext p;
                encoder.encode(f, scale, p);
                enc_inputs.encrypt(p, features[i]);
                put_ct(w, "feat", features[i], (std::uint32_t)i);
            }
            std::vector<double> wv(nw);
            for (double &x : wv) x = 0.5 * next();
            Plaintext pw;
            encoder.encode(wv, scale, pw);
            Ciphertext cw;
            enc_inputs.encrypt(pw, cw);
            put_ct(w, "weights", cw);
            Encryptor e3(context, pk);  // draw 4
            put_ct(w, "predict", predict_cipher_weights(features, cw, nw, scale, evaluator, encoder, gk, rk, e3, params));
        }
        std::printf("%s/lr_c4.bin written\n", argv[1]);
    } catch (const std::exception &ex) {
        std::fprintf(stderr, "xcheck_lr: %s\n", ex.what());
        return 1;
    }
    return 0;
}
