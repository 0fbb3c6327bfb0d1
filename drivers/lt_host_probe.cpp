// Where the wall time of the reference's Linear_Transform_Plain (helper.h:237-262) goes when it runs through the shim:
// the pieces of the unchanged function timed one by one at N = 8192, d = 1000, default (power-of-two) Galois keys --
// argument copies, SEALContext construction, recording the d rotations and products, add_many (the submission) and the
// wait for the device.  Development tool (make -C drivers _ref/lt_host_probe); prints microseconds per piece.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "seal/seal.h"
using namespace std;
using namespace seal;
static double now() { return chrono::duration<double, micro>(chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int d = argc > 1 ? atoi(argv[1]) : 1000, reps = argc > 2 ? atoi(argv[2]) : 5;
    EncryptionParameters params(scheme_type::CKKS);
    params.set_poly_modulus_degree(8192);
    params.set_coeff_modulus(CoeffModulus::Create(8192, {60, 40, 40, 60}));
    auto context = SEALContext::Create(params);
    KeyGenerator keygen(context);
    GaloisKeys gk = keygen.galois_keys();
    Encryptor encryptor(context, keygen.public_key());
    Decryptor decryptor(context, keygen.secret_key());
    CKKSEncoder encoder(context);
    const double scale = pow(2.0, 40);
    vector<Plaintext> diags(d);
    for (int l = 0; l < d; l++) {
        vector<double> v(d);
        for (int i = 0; i < d; i++) v[i] = 0.001 * ((i + 3 * l) % 17);
        encoder.encode(v, scale, diags[l]);
    }
    Plaintext pv;
    encoder.encode(vector<double>(d, 0.5), scale, pv);
    Ciphertext ct;
    encryptor.encrypt(pv, ct);
    for (int rep = 0; rep < reps; rep++) {
        hefx_stream_sync(context->engine()->live(), nullptr);
        const double t0 = now();
        vector<Plaintext> U = diags;  // by-value arguments of the reference's signature
        GaloisKeys g2 = gk;
        const double t1 = now();
        SEALContext c2(params);
        Evaluator evaluator(c2);
        const double t2 = now();
        Ciphertext ct_rot, ct_new;
        evaluator.rotate_vector(ct, -d, g2, ct_rot);
        evaluator.add(ct, ct_rot, ct_new);
        vector<Ciphertext> res(d);
        evaluator.multiply_plain(ct_new, U[0], res[0]);
        const double t3 = now();
        for (int l = 1; l < d; l++) {
            Ciphertext tmp;
            evaluator.rotate_vector(ct_new, l, g2, tmp);
            evaluator.multiply_plain(tmp, U[l], res[l]);
        }
        const double t4 = now();
        Ciphertext out;
        evaluator.add_many(res, out);
        const double t5 = now();
        hefx_stream_sync(context->engine()->live(), nullptr);
        Plaintext pd;
        decryptor.decrypt(out, pd);
        const double t6 = now();
        printf("d=%d: copies %.0f  context %.0f  head %.0f  record loop %.0f  add_many %.0f  wait+decrypt %.0f  total %.0f us\n", d,
               t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0);
    }
    return 0;
}
