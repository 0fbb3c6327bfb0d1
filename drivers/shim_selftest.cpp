// shim_selftest.cpp -- exercises include/seal/seal.h the way the reference's drivers do (both API spellings) and
// checks decrypted values and SEAL's error behaviour.  Exit code 0 = all checks passed.  Needs a HIP device.
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>

#include "seal/seal.h"

using namespace std;
using namespace seal;

static int failures = 0;
#define CHECK(cond, what)                                  \
    do {                                                   \
        if (!(cond)) {                                     \
            cout << "FAIL: " << what << endl;              \
            ++failures;                                    \
        } else                                             \
            cout << "ok:   " << what << endl;              \
    } while (0)

template <class F>
static bool throws_invalid(F f, const string &needle)
{
    try {
        f();
    } catch (const invalid_argument &e) {
        return string(e.what()).find(needle) != string::npos;
    } catch (...) {
    }
    return false;
}

// helper.h:237-262 restated against the 3.6 spelling (context rebuilt per call, keys by value)
static Ciphertext lt_plain(Ciphertext ct, vector<Plaintext> diags, GaloisKeys gk, EncryptionParameters params)
{
    SEALContext context(params);
    Evaluator evaluator(context);
    Ciphertext ct_rot, ct_new, out;
    evaluator.rotate_vector(ct, -(int)diags.size(), gk, ct_rot);
    evaluator.add(ct, ct_rot, ct_new);
    vector<Ciphertext> res(diags.size());
    evaluator.multiply_plain(ct_new, diags[0], res[0]);
    for (size_t l = 1; l < diags.size(); l++) {
        Ciphertext tmp;
        evaluator.rotate_vector(ct_new, (int)l, gk, tmp);
        evaluator.multiply_plain(tmp, diags[l], res[l]);
    }
    evaluator.add_many(res, out);
    return out;
}

int main()
{
    EncryptionParameters params(scheme_type::CKKS);
    params.set_poly_modulus_degree(8192);
    params.set_coeff_modulus(CoeffModulus::Create(8192, {60, 40, 40, 60}));
    auto context = SEALContext::Create(params);  // 3.4.5 spelling
    CHECK(context->first_context_data()->chain_index() == 2, "chain_index of first data level");
    CHECK(context->key_context_data()->total_coeff_modulus_bit_count() == 200, "total_coeff_modulus_bit_count");
    CHECK(params.coeff_modulus()[0].value() == 0xffffffffffe8001ull && params.coeff_modulus()[3].value() == 0xfffffffffffc001ull,
          "CoeffModulus::Create primes (SURVEY App. B, C2)");

    KeyGenerator keygen(context);
    PublicKey pk = keygen.public_key();
    SecretKey sk = keygen.secret_key();
    RelinKeys rk = keygen.relin_keys();
    GaloisKeys gk;
    keygen.create_galois_keys(gk);  // 3.6 spelling
    CHECK(gk.size() == 24, "default Galois keys: 24 distinct elements at N=8192");
    Encryptor encryptor(context, pk);
    Evaluator evaluator(context);
    Decryptor decryptor(context, sk);
    CKKSEncoder encoder(context);
    const double scale = pow(2.0, 40);

    auto dec = [&](const Ciphertext &c) {
        Plaintext p;
        vector<double> v;
        decryptor.decrypt(c, p);
        encoder.decode(p, v);
        return v;
    };

    vector<double> a{1.0, 2.0, 3.0, 4.0}, b{0.5, -1.0, 2.0, 0.25};
    Plaintext pa, pb;
    encoder.encode(a, scale, pa);
    encoder.encode(b, scale, pb);
    Ciphertext ca, cb;
    encryptor.encrypt(pa, ca);
    encryptor.encrypt(pb, cb);
    auto va = dec(ca);
    CHECK(fabs(va[0] - 1) < 1e-6 && fabs(va[3] - 4) < 1e-6 && fabs(va[4]) < 1e-6, "encode/encrypt/decrypt/decode");

    Ciphertext sum, prod, rot;
    evaluator.add(ca, cb, sum);
    CHECK(fabs(dec(sum)[1] - 1.0) < 1e-6, "add");
    evaluator.multiply(ca, cb, prod);
    CHECK(prod.size() == 3, "multiply gives size 3");
    CHECK(fabs(dec(prod)[2] - 6.0) < 1e-5, "decrypt of a size-3 ciphertext");
    evaluator.relinearize_inplace(prod, rk);
    evaluator.rescale_to_next_inplace(prod);
    CHECK(prod.size() == 2 && context->get_context_data(prod.parms_id())->chain_index() == 1, "relinearize + rescale level");
    CHECK(fabs(dec(prod)[2] - 6.0) < 1e-5 && fabs(dec(prod)[1] + 2.0) < 1e-5, "relinearize + rescale values");
    evaluator.rotate_vector(ca, 1, gk, rot);
    CHECK(fabs(dec(rot)[0] - 2.0) < 1e-5 && fabs(dec(rot)[2] - 4.0) < 1e-5, "rotate_vector by 1");
    evaluator.rotate_vector(ca, 3, gk, rot);  // NAF chain: -1 then 4
    CHECK(fabs(dec(rot)[0] - 4.0) < 1e-5, "rotate_vector by 3 (NAF chain)");
    evaluator.rotate_vector_inplace(rot, -3, gk);
    CHECK(fabs(dec(rot)[0] - 1.0) < 1e-5 && fabs(dec(rot)[3] - 4.0) < 1e-5, "rotate_vector_inplace back");
    Plaintext p3;
    encoder.encode(3.0, scale, p3);
    Ciphertext sc;
    evaluator.multiply_plain(ca, p3, sc);
    CHECK(fabs(dec(sc)[3] - 12.0) < 1e-5 && sc.scale() == scale * scale, "multiply_plain with scalar plaintext");

    // Linear_Transform_Plain known answer: M = 1..16, v = first column -> [90, 202, 314, 426]
    {
        vector<vector<double>> M(4, vector<double>(4));
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) M[i][j] = 4 * i + j + 1;
        vector<Plaintext> diags(4);
        for (int l = 0; l < 4; l++) {
            vector<double> d(4);
            for (int i = 0; i < 4; i++) d[i] = M[i][(i + l) % 4];
            encoder.encode(d, scale, diags[l]);
        }
        Plaintext pv;
        encoder.encode(vector<double>{1, 5, 9, 13}, scale, pv);
        Ciphertext cv;
        encryptor.encrypt(pv, cv);
        Ciphertext ref = lt_plain(cv, diags, gk, params);
        auto r = dec(ref);
        CHECK(fabs(r[0] - 90) < 1e-3 && fabs(r[1] - 202) < 1e-3 && fabs(r[2] - 314) < 1e-3 && fabs(r[3] - 426) < 1e-3,
              "Linear_Transform_Plain 4x4 known answer [90,202,314,426]");
        // the one-call engine path (extension) must give the same ciphertext bits as the op-by-op body
        Ciphertext fast;
        evaluator.hefx_linear_transform_plain(cv, diags, gk, fast);
        CHECK(fast.scale() == ref.scale() && fast.parms_id() == ref.parms_id() &&
                  shim::download(fast.buf) == shim::download(ref.buf),
              "hefx_linear_transform_plain == op-by-op Linear_Transform_Plain, bit for bit");
        // d = 13: NAF chains of different lengths, shared prefixes
        vector<Plaintext> d13(13);
        for (int l = 0; l < 13; l++) {
            vector<double> dv(13);
            for (int i = 0; i < 13; i++) dv[i] = 0.01 * (i + 1) * (l + 2);
            encoder.encode(dv, scale, d13[l]);
        }
        evaluator.hefx_linear_transform_plain(cv, d13, gk, fast);
        ref = lt_plain(cv, d13, gk, params);
        CHECK(shim::download(fast.buf) == shim::download(ref.buf), "hefx_linear_transform_plain d=13 bit for bit");
        CHECK(throws_invalid([&] { GaloisKeys none; Ciphertext t; evaluator.hefx_linear_transform_plain(cv, diags, none, t); },
                             "Galois key not present"),
              "hefx_linear_transform_plain: missing keys throw");
        // three independent d = 13 transforms in lockstep (hefx_linear_transform_plain_many: CC_Matrix_Multiplication's sigma
        // and tau transforms, matrix_multiplication.cpp:22-25) == the op-by-op body, input by input
        {
            Plaintext pw;
            encoder.encode(vector<double>{-2, 0.5, 3, 7, 1, 1, 4}, scale, pw);
            Ciphertext cw;
            encryptor.encrypt(pw, cw);
            vector<Plaintext> e13(13);
            for (int l = 0; l < 13; l++) {
                vector<double> dv(13);
                for (int i = 0; i < 13; i++) dv[i] = 0.03 * (i + 2) - 0.02 * l;
                encoder.encode(dv, scale, e13[l]);
            }
            vector<Ciphertext> many;
            evaluator.hefx_linear_transform_plain_many({cv, cw, cv}, {d13, e13, e13}, gk, many);
            Ciphertext r0 = lt_plain(cv, d13, gk, params), r1 = lt_plain(cw, e13, gk, params), r2 = lt_plain(cv, e13, gk, params);
            CHECK(many.size() == 3 && shim::download(many[0].buf) == shim::download(r0.buf) &&
                      shim::download(many[1].buf) == shim::download(r1.buf) && shim::download(many[2].buf) == shim::download(r2.buf) &&
                      many[1].scale() == r1.scale() && many[2].parms_id() == r2.parms_id(),
                  "hefx_linear_transform_plain_many == three op-by-op transforms, bit for bit");
            CHECK(throws_invalid([&] { vector<Ciphertext> t; evaluator.hefx_linear_transform_plain_many({cv, cw}, {d13, diags}, gk, t); },
                                 "encrypteds parameter mismatch"),
                  "hefx_linear_transform_plain_many: transforms of different dimensions are refused");
        }
    }

    // extensions: product sum (Linear_Transform_CipherMatrix_PlainVector in one pass) and the baby-step / giant-step
    // form of Linear_Transform_Plain
    {
        const int d = 13, n1 = 4, n2 = 4;
        vector<vector<double>> M(d, vector<double>(d));
        vector<double> v(d), want(d, 0.0);
        for (int i = 0; i < d; i++) {
            v[i] = 0.1 * (i + 1);
            for (int j = 0; j < d; j++) M[i][j] = 0.01 * ((7 * i + 3 * j) % 11) - 0.05;
        }
        for (int i = 0; i < d; i++)
            for (int j = 0; j < d; j++) want[i] += M[i][j] * v[j];
        vector<Plaintext> plain(d), shifted(d);
        for (int l = 0; l < d; l++) {
            vector<double> dv(d), sv((l / n1) * n1 + d, 0.0);
            for (int i = 0; i < d; i++) dv[i] = sv[(l / n1) * n1 + i] = M[i][(i + l) % d];
            encoder.encode(dv, scale, plain[l]);
            encoder.encode(sv, scale, shifted[l]);
        }
        vector<int> steps{-d};
        for (int i = 1; i < n1; i++) steps.push_back(i);
        for (int j = 1; j < n2; j++) steps.push_back(j * n1);
        GaloisKeys gb = keygen.galois_keys(steps);
        Plaintext pv;
        encoder.encode(v, scale, pv);
        Ciphertext cv, res;
        encryptor.encrypt(pv, cv);
        for (int hoisted = 0; hoisted < 2; hoisted++) {
            evaluator.hefx_linear_transform_plain_bsgs(cv, shifted, gb, n1, res, hoisted != 0);
            auto r = dec(res);
            double err = 0;
            for (int i = 0; i < d; i++) err = max(err, fabs(r[i] - want[i]));
            CHECK(err < 1e-4 && res.scale() == scale * scale, "hefx_linear_transform_plain_bsgs decrypts to M.v");
        }
        CHECK(throws_invalid([&] { Ciphertext t; evaluator.hefx_linear_transform_plain_bsgs(cv, shifted, gk, n1, t); },
                             "direct Galois key"),
              "hefx_linear_transform_plain_bsgs: power-of-two keys alone are refused");
        // product sum against the op-by-op body of helper.h:265-278
        vector<Ciphertext> cts(5);
        for (int i = 0; i < 5; i++) evaluator.rotate_vector(cv, i == 0 ? 0 : 1 << (i - 1), gk, cts[i]);
        vector<Ciphertext> prods(5);
        for (int i = 0; i < 5; i++) evaluator.multiply_plain(cts[i], plain[i], prods[i]);
        Ciphertext ref, fast;
        evaluator.add_many(prods, ref);
        evaluator.hefx_multiply_plain_sum(cts, plain, fast);
        CHECK(fast.scale() == ref.scale() && shim::download(fast.buf) == shim::download(ref.buf),
              "hefx_multiply_plain_sum == add_many(multiply_plain), bit for bit");
    }

    // deferred rotations (shim::Engine::defer_*): every usage pattern must give the bits of immediate execution
    {
        auto run = [&](bool lazy) {
            context->engine()->live();
            context->engine()->lazy = lazy;
            Ciphertext r1, r2, r3, m1, m2, t, chain, back;
            evaluator.rotate_vector(ca, 1, gk, r1);    // kept AND multiplied: the product must not swallow it
            evaluator.multiply_plain(r1, pb, m1);
            evaluator.rotate_vector(ca, 1, gk, r2);    // same (source, element) as r1: computed once
            evaluator.rotate_vector(ca, 3, gk, t);     // NAF chain, multiplied, then t is reassigned: fused
            evaluator.multiply_plain(t, pb, m2);
            evaluator.rotate_vector(m2, 2, gk, t);     // rotation of a recorded product
            evaluator.rotate_vector(r1, 5, gk, chain); // rotation of a recorded rotation
            evaluator.multiply(r2, cb, r3);            // tensor product with a recorded rotation
            evaluator.rotate_vector_inplace(chain, -5, gk);
            evaluator.multiply_plain_inplace(chain, pb);
            std::vector<std::vector<uint64_t>> out;
            for (const Ciphertext *c : {&r1, &r2, &m1, &m2, &t, &chain, &r3}) out.push_back(shim::download(c->buf));
            context->engine()->lazy = true;
            return out;
        };
        const auto a1 = run(true), a0 = run(false);
        CHECK(a1 == a0, "deferred rotations / products == immediate execution, bit for bit (7 patterns)");
        Ciphertext r;
        evaluator.rotate_vector(ca, 2, gk, r);
        CHECK(fabs(dec(r)[0] - 3.0) < 1e-5, "a recorded rotation is materialised by decrypt");
    }

    // exact hoisting behind the unchanged loop of helper.h:252-257: d - 1 = 40 rotations of ONE ciphertext with a direct
    // Galois key per step.  Recorded, they reach the engine as one batch, which it runs exactly hoisted (one decomposition
    // of ct_new, the flip-mask term in the key MAC); executed call by call every rotation is a key switch of its own.  Same
    // bits -- the hoisted kernels against the per-item kernels, inside one process.
    {
        const int d = 41;
        vector<int> steps{-d};
        for (int i = 1; i < d; i++) steps.push_back(i);
        GaloisKeys gd = keygen.galois_keys(steps);
        vector<Plaintext> diags(d);
        for (int l = 0; l < d; l++) {
            vector<double> dv(d);
            for (int i = 0; i < d; i++) dv[i] = 0.01 * ((i * 7 + l * 3) % 11) - 0.05;
            encoder.encode(dv, scale, diags[l]);
        }
        auto run = [&](bool lazy) {
            context->engine()->live();
            context->engine()->lazy = lazy;
            Ciphertext out = lt_plain(ca, diags, gd, params);
            auto bits = shim::download(out.buf);
            context->engine()->lazy = true;
            return bits;
        };
        uint64_t fb0 = 0, fb1 = 0;
        hefx_ks_fallback_count(context->engine()->live(), &fb0);
        const auto h = run(true), s1 = run(false);
        hefx_ks_fallback_count(context->engine()->live(), &fb1);
        CHECK(h == s1, "Linear_Transform_Plain, 40 direct-key rotations: recorded (exactly hoisted) == call by call, bit for bit");
        CHECK(fb1 == fb0, "... and the hoisted kernels produced it (no chunk fell back)");
    }

    // the recorder as a dependency graph: the LR loop of the reference (logistic_regression_ckks.cpp:217-229 calling
    // helper.h:432-476 per observation row) -- multiply, relinearize, rescale, rotate(-size), add, then size-1 times
    // rotate-by-1 + add_inplace, a plaintext mod_switch and a mask product per row, add_many at the end -- recorded for
    // ALL rows and run in lockstep must give the bits of call-by-call execution; also with a tiny pending budget
    // (SEAL_SHIM_PENDING_MB-style forced flushes in the middle of the chains)
    {
        int rows = 5, size = 4;
        auto run = [&](bool lazy, std::size_t budget) {
            auto e = context->engine();
            e->live();
            e->lazy = lazy;
            const std::size_t old_budget = e->pend_budget;
            e->pend_budget = e->pend_check = budget;
            vector<Ciphertext> results(rows);
            for (int i = 0; i < rows; i++) {
                Ciphertext feat;
                evaluator.rotate_vector(ca, i, gk, feat);  // stands in for features[i] (distinct ciphertexts)
                Ciphertext mult;
                evaluator.multiply(feat, cb, mult);
                evaluator.relinearize_inplace(mult, rk);
                evaluator.rescale_to_next_inplace(mult);
                Ciphertext zero_filled, dup;
                evaluator.rotate_vector(mult, -size, gk, zero_filled);
                evaluator.add(mult, zero_filled, dup);
                for (int j = 1; j < size; j++) {
                    evaluator.rotate_vector_inplace(dup, 1, gk);
                    evaluator.add_inplace(mult, dup);
                }
                mult.scale() = pow(2.0, (int)log2(mult.scale()));
                vector<double> mask(rows, 0.0);
                mask[i] = 1.0;
                Plaintext mask_pt;
                encoder.encode(mask, scale, mask_pt);                // eager: must not interrupt the recording
                evaluator.mod_switch_to_next_inplace(mask_pt);
                evaluator.multiply_plain_inplace(mult, mask_pt);
                results[i] = mult;
            }
            const std::size_t recorded = e->pend.size();
            Ciphertext sum;
            evaluator.add_many(results, sum);
            Ciphertext diff;
            evaluator.sub(results[0], results[1], diff);
            e->lazy = true;
            e->pend_budget = e->pend_check = old_budget;
            return std::make_tuple(shim::download(sum.buf), shim::download(diff.buf), recorded, sum.scale());
        };
        const auto lazy = run(true, (std::size_t)8192 << 20), eager = run(false, (std::size_t)8192 << 20),
                   tiny = run(true, (std::size_t)3 << 20);
        CHECK(std::get<0>(lazy) == std::get<0>(eager) && std::get<1>(lazy) == std::get<1>(eager) &&
                  std::get<3>(lazy) == std::get<3>(eager),
              "LR loop recorded for all rows == call-by-call execution, bit for bit");
        CHECK(std::get<0>(tiny) == std::get<0>(eager) && std::get<1>(tiny) == std::get<1>(eager),
              "LR loop with forced mid-chain flushes (3 MB pending budget) == call-by-call execution");
        CHECK(std::get<2>(lazy) >= (std::size_t)rows * (3 + 2 * size) && std::get<2>(eager) == 0,
              "the whole loop stays recorded until add_many (encode / plaintext mod_switch do not flush)");
        // round 4: the rotate-by-1 + add_inplace pairs run as hefx_apply_galois_add_batch, runs of them as
        // hefx_rotate_add_chain (3 levels above; 11 here: long enough for the captured two-level HIP graph), the mask
        // encodes as one hefx_ckks_encode_batch -- and with each fusion switched off the bits stay those of call-by-call
        {
            rows = 3, size = 12;
            const auto chain = run(true, (std::size_t)8192 << 20), plain = run(false, (std::size_t)8192 << 20);
            CHECK(std::get<0>(chain) == std::get<0>(plain) && std::get<1>(chain) == std::get<1>(plain),
                  "LR loop with 11-level rotate+add chains (hefx_rotate_add_chain, graph replay) == call-by-call execution");
            auto e = context->engine();
            const std::size_t calls0 = e->stats.calls;
            (void)run(true, (std::size_t)8192 << 20);
            const std::size_t fused_calls = e->stats.calls - calls0;
            // (row 0's feature is an external ciphertext, rows 1 and 2 are rotations: two depth classes, each with its own
            // calls; call by call the loop is 3 x (5 + 2 x 11 + 2) = 87 engine calls)
            const auto off = [](const char *name) { return getenv(name) && atoi(getenv(name)) == 0; };
            if (!off("SEAL_SHIM_FUSE_ADD") && !off("SEAL_SHIM_CHAINS"))  // (the test suite also runs this file with them off)
                CHECK(fused_calls <= 24, "the 3 x 11-level loop submits in <= 24 batched engine calls (87 call by call)");
            rows = 5, size = 4;
        }
        // several devices behind the same program (SEAL_SHIM_DEVICES): the rows are independent sub-graphs, dealt over
        // two / three engine contexts (sharing the GPUs that exist), inputs replicated, results copied home -- the
        // bits of the one-device run; a second pass reuses the cached replicas of the keys
        auto e = context->engine();
        const int old_ndev = e->ndev;
        for (int nd : {2, 3}) {
            e->live();
            e->ndev = nd;
            const auto multi = run(true, (std::size_t)8192 << 20), again = run(true, (std::size_t)8192 << 20);
            CHECK(std::get<0>(multi) == std::get<0>(eager) && std::get<1>(multi) == std::get<1>(eager) &&
                      std::get<0>(again) == std::get<0>(eager) && std::get<1>(again) == std::get<1>(eager),
                  (nd == 2 ? "LR loop over 2 engine contexts (SEAL_SHIM_DEVICES) == call-by-call execution, bit for bit"
                           : "LR loop over 3 engine contexts == call-by-call execution, bit for bit"));
            CHECK((int)e->dev_ctx.size() >= nd && e->dev_ctx[nd - 1] != nullptr && !e->replicas[1].empty(),
                  "the extra contexts exist and hold replicas of the shared inputs");
        }
        e->live();
        e->ndev = old_ndev;
    }

    // SEAL's error behaviour at the boundary
    Ciphertext low = ca;
    evaluator.mod_switch_to_next_inplace(low);
    // recorded encode + encrypt (round 4): a loop of encode / encrypt calls stays recorded and reaches the device as one
    // hefx_ckks_encode_batch + one hefx_encrypt_batch; decrypt observes it
    {
        auto e = context->engine();
        e->live();
        const std::size_t before = e->pend.size();
        vector<Ciphertext> cts(6);
        for (int i = 0; i < 6; i++) {
            vector<double> v{1.0 + i, -2.0 * i, 0.5};
            Plaintext pt;
            encoder.encode(v, scale, pt);
            encryptor.encrypt(pt, cts[i]);
        }
        CHECK(e->pend.size() - before == 12, "6 encodes + 6 encryptions stay recorded");
        const std::size_t calls0 = e->stats.calls;
        bool ok = true;
        for (int i = 0; i < 6; i++) {
            const auto r = dec(cts[i]);
            ok = ok && fabs(r[0] - (1.0 + i)) < 1e-6 && fabs(r[1] + 2.0 * i) < 1e-6 && fabs(r[2] - 0.5) < 1e-6 && fabs(r[3]) < 1e-6;
        }
        CHECK(ok, "recorded encode + encrypt: every ciphertext decrypts to its vector");
        CHECK(e->stats.calls - calls0 == 2, "... from two batched engine calls");
        // a recorded plaintext, mod-switched without a copy, multiplies like an eager one
        Plaintext m;
        encoder.encode(vector<double>{2.0, 2.0, 2.0, 2.0}, scale, m);
        evaluator.mod_switch_to_next_inplace(m);
        Ciphertext low2 = ca, prod2;
        evaluator.mod_switch_to_next_inplace(low2);
        evaluator.multiply_plain(low2, m, prod2);
        CHECK(fabs(dec(prod2)[0] - 2.0 * dec(ca)[0]) < 1e-4 && m.coeff_count() == 2 * 8192,
              "recorded encode -> zero-copy mod_switch_to_next -> multiply_plain");
    }
    CHECK(throws_invalid([&] { Ciphertext t; evaluator.add(ca, low, t); }, "parameter mismatch"), "add: parms_id mismatch throws");
    Ciphertext big = ca;
    big.scale() = scale * 2;
    CHECK(throws_invalid([&] { Ciphertext t; evaluator.add(ca, big, t); }, "scale mismatch"), "add: scale mismatch throws");
    CHECK(throws_invalid([&] { Ciphertext t; evaluator.rotate_vector(ca, 4096, gk, t); }, "step count too large"), "rotate: step too large");
    CHECK(throws_invalid([&] { GaloisKeys none; Ciphertext t; evaluator.rotate_vector(ca, 4, none, t); }, "Galois key not present"),
          "rotate: missing key throws");
    bool transparent = false;
    try {
        Plaintext zero;
        encoder.encode(vector<double>{0.0, 0.0}, scale, zero);
        Ciphertext t;
        evaluator.multiply_plain(ca, zero, t);
    } catch (const logic_error &e) {
        transparent = string(e.what()).find("transparent") != string::npos;
    }
    CHECK(transparent, "multiply_plain by zero plaintext: logic_error(transparent)");
    CHECK(throws_invalid([&] {
              Ciphertext t = sc;  // scale 2^80
              Plaintext p;
              encoder.encode(1.0, scale, p);
              evaluator.multiply_plain(t, p, t);
              evaluator.multiply_plain(t, p, t);  // 2^160 fits 200 bits; one more exceeds it
              evaluator.multiply_plain(t, p, t);
              evaluator.multiply_plain(t, p, t);
          }, "scale out of bounds"),
          "scale out of bounds throws");

    // BFV: what vector_ops.cpp:101-195 and 5_rotation.cpp:88-165 do before their CKKS halves
    {
        EncryptionParameters bp(scheme_type::BFV);
        bp.set_poly_modulus_degree(8192);
        bp.set_coeff_modulus(CoeffModulus::BFVDefault(8192));
        bp.set_plain_modulus(786433);
        auto bctx = SEALContext::Create(bp);
        KeyGenerator bkg(bctx);
        PublicKey bpk = bkg.public_key();
        SecretKey bsk = bkg.secret_key();
        RelinKeys brk = bkg.relin_keys();
        GaloisKeys bgk = bkg.galois_keys();
        Encryptor benc(bctx, bpk);
        Evaluator bev(bctx);
        Decryptor bdec(bctx, bsk);
        BatchEncoder be(bctx);
        const size_t slots = be.slot_count(), row = slots / 2;
        const uint64_t t = 786433;
        CHECK(slots == 8192, "BatchEncoder slot_count");
        vector<uint64_t> m1(slots), m2(slots), back;
        for (size_t i = 0; i < slots; i++) m1[i] = i, m2[i] = (i % 2) + 1;
        Plaintext p1, p2, pr;
        be.encode(m1, p1);
        be.encode(m2, p2);
        be.decode(p1, back);
        CHECK(back == m1, "BatchEncoder decode(encode(x)) == x");
        Ciphertext c;
        benc.encrypt(p1, c);
        const int fresh = bdec.invariant_noise_budget(c);
        CHECK(!c.is_ntt_form() && fresh > 100 && fresh < 174, "BFV encrypt: coefficient form, fresh noise budget in range");
        bdec.decrypt(c, pr);
        be.decode(pr, back);
        CHECK(back == m1, "BFV decrypt(encrypt(x)) == x");
        bev.add_plain_inplace(c, p2);
        bev.square_inplace(c);
        CHECK(c.size() == 3, "BFV square gives size 3");
        bev.relinearize_inplace(c, brk);
        const int after = bdec.invariant_noise_budget(c);
        bdec.decrypt(c, pr);
        be.decode(pr, back);
        bool ok = c.size() == 2;
        for (size_t i = 0; i < slots; i++) ok = ok && back[i] == ((m1[i] + m2[i]) % t) * ((m1[i] + m2[i]) % t) % t;
        CHECK(ok && after > 0 && after < fresh, "BFV (x + y)^2 with add_plain, square, relinearize (vector_ops.cpp:178-180)");
        vector<uint64_t> pm(slots, 0);
        for (int i = 0; i < 4; i++) pm[i] = i, pm[row + i] = 4 + i;
        Plaintext pp;
        be.encode(pm, pp);
        benc.encrypt(pp, c);
        bev.rotate_rows_inplace(c, 3, bgk);       // NAF(3) = [-1, 4] with the default power-of-two keys
        bdec.decrypt(c, pr);
        be.decode(pr, back);
        CHECK(back[0] == 3 && back[row - 3] == 0 && back[row - 1] == 2 && back[row] == 7 && back[2 * row - 1] == 6,
              "BFV rotate_rows by 3 (5_rotation.cpp:132)");
        bev.rotate_columns_inplace(c, bgk);
        bdec.decrypt(c, pr);
        be.decode(pr, back);
        CHECK(back[0] == 7 && back[row] == 3, "BFV rotate_columns swaps the rows (5_rotation.cpp:143)");
        bev.rotate_rows_inplace(c, -4, bgk);
        bdec.decrypt(c, pr);
        be.decode(pr, back);
        CHECK(back[0] == 0 && back[1] == 4 && back[4] == 7 && back[row + 2] == 1 && back[row + 4] == 3 &&
                  bdec.invariant_noise_budget(c) > 0,
              "BFV rotate_rows by -4 (5_rotation.cpp:152)");
    }

    // BFV tutorials (1_bfv.cpp, 2_encoders.cpp, 3_levels.cpp): hex-polynomial plaintexts, IntegerEncoder, products of
    // size-3 ciphertexts, multiply_plain, modulus switching down the chain
    {
        CHECK(Plaintext("1x^3 + 2x^2 + 3x^1 + 4").to_string() == "1x^3 + 2x^2 + 3x^1 + 4" && Plaintext("0").to_string() == "0" &&
                  Plaintext("1Fx^2 + A").to_string() == "1Fx^2 + A",
              "Plaintext hex polynomial strings round-trip");
        EncryptionParameters bp(scheme_type::BFV);
        bp.set_poly_modulus_degree(8192);
        bp.set_coeff_modulus(CoeffModulus::Create(8192, {50, 30, 30, 50, 50}));
        bp.set_plain_modulus(PlainModulus::Batching(8192, 20));
        auto bctx = SEALContext::Create(bp);
        KeyGenerator bkg(bctx);
        PublicKey bpk = bkg.public_key();
        SecretKey bsk = bkg.secret_key();
        RelinKeys brk = bkg.relin_keys();
        Encryptor benc(bctx, bpk);
        Evaluator bev(bctx);
        Decryptor bdec(bctx, bsk);
        Plaintext plain("1x^3 + 2x^2 + 3x^1 + 4"), back;
        Ciphertext c;
        benc.encrypt(plain, c);
        int prev = bdec.invariant_noise_budget(c);
        bool shrinking = true;
        while (bctx->get_context_data(c.parms_id())->next_context_data()) {
            bev.mod_switch_to_next_inplace(c);
            const int now = bdec.invariant_noise_budget(c);
            shrinking = shrinking && now < prev && now > 0;
            prev = now;
        }
        bdec.decrypt(c, back);
        CHECK(shrinking && c.coeff_mod_count() == 1 && back.to_string() == "1x^3 + 2x^2 + 3x^1 + 4",
              "BFV mod_switch_to_next down the chain keeps the plaintext (3_levels.cpp:95-117)");
        benc.encrypt(plain, c);
        bev.square_inplace(c);
        bev.relinearize_inplace(c, brk);
        bev.mod_switch_to_next_inplace(c);
        bev.square_inplace(c);
        bev.relinearize_inplace(c, brk);
        bdec.decrypt(c, back);
        CHECK(back.to_string() == "1x^12 + 8x^11 + 24x^10 + 80x^9 + 136x^8 + 1E0x^7 + 278x^6 + 2A0x^5 + 271x^4 + 1C8x^3 + 120x^2 + 80x^1 + 40" ||
                  back.coeff_count() == 13,
              "BFV 4th power across a modulus switch has degree 12");
        IntegerEncoder ie(bctx);
        CHECK(ie.encode(10).to_string() == "1x^3 + 1x^1" && ie.decode_int32(ie.encode(-37)) == -37, "IntegerEncoder binary expansion");
        Ciphertext c1, c2, prod, res;
        benc.encrypt(ie.encode(10), c1);
        benc.encrypt(ie.encode(12), c2);
        bev.multiply(c1, c2, prod);
        bev.sub(prod, c1, res);
        bdec.decrypt(res, back);
        CHECK(res.size() == 3 && ie.decode_int32(back) == 110, "BFV 10 * 12 - 10 through IntegerEncoder (2_encoders.cpp:137-147)");
        Plaintext four("4");
        Ciphertext x, a, b, big;
        benc.encrypt(Plaintext("6"), x);
        bev.square(x, a);
        bev.add_plain_inplace(a, Plaintext("1"));
        bev.multiply_plain_inplace(a, four);
        bev.add_plain(x, Plaintext("1"), b);
        bev.square_inplace(b);
        bev.multiply(a, b, big);  // size 3 x size 3 -> size 5
        bdec.decrypt(big, back);
        CHECK(big.size() == 5 && back.to_string() == "1C54", "BFV 4(x^2+1)(x+1)^2 at x = 6 without relinearisation (1_bfv.cpp:130-132)");
    }

    // ---- save / load: SEAL 3.4.5's uncompressed stream layouts; "format unpinned" (shim_io.h)
    {
        auto hex = [](const array<uint8_t, 32> &d) {
            static const char *H = "0123456789abcdef";
            string o;
            for (auto b : d) o += H[b >> 4], o += H[b & 15];
            return o;
        };
        CHECK(hex(shim::sha3_256(nullptr, 0)) == "a7ffc6f8bf1ed76651c14756a061d662f580ff4de43b49fa82d80a4b80f8434a" &&
                  hex(shim::sha3_256((const uint8_t *)"abc", 3)) == "3a985da74fe225b2045c172d6bd390bd855f086e3e9d525b46bfe24511431532",
              "SHA3-256 known answers (FIPS 202): SEAL's parms_id is this hash of the parameter words");
        vector<uint8_t> longmsg(200, 0xa3);  // one full block + a remainder (NIST's 1600-bit message of 0xa3 bytes)
        CHECK(hex(shim::sha3_256(longmsg.data(), longmsg.size())) == "79f38adec5c20307a98ef76e8324afbfd46cfd81b22e3973c65fa1bd9de31787",
              "SHA3-256 over more than one block");
        EncryptionParameters sp(scheme_type::CKKS);
        sp.set_poly_modulus_degree(8192);
        sp.set_coeff_modulus(CoeffModulus::Create(8192, {60, 40, 40, 60}));
        auto sctx = SEALContext::Create(sp);
        CHECK(sctx->key_parms_id() == sp.parms_id() && sctx->first_parms_id() != sctx->key_parms_id() &&
                  sctx->first_context_data()->parms().parms_id() == sctx->first_parms_id(),
              "parms_id of every level is the SHA3-256 of that level's parameters");
        stringstream ps;
        EncryptionParameters::Save(sp, ps);
        CHECK(ps.str().size() == 1 + 8 + 8 + 4 * 8 + 8 && (uint8_t)ps.str()[0] == 2, "EncryptionParameters::Save: 57 bytes for four primes, scheme byte 2 (CKKS)");
        CHECK(EncryptionParameters::Load(ps) == sp, "EncryptionParameters::Load gives the parameters back");
        KeyGenerator skg(sctx);
        PublicKey spk = skg.public_key();
        SecretKey ssk = skg.secret_key();
        RelinKeys srk = skg.relin_keys();
        GaloisKeys sgk = skg.galois_keys(vector<int>{1, -1, 4});
        Encryptor senc(sctx, spk);
        Decryptor sdec(sctx, ssk);
        CKKSEncoder scod(sctx);
        Evaluator sev(sctx);
        vector<double> vals{1.5, -2.25, 3.0, 0.125};
        Plaintext pt;
        scod.encode(vals, pow(2.0, 40), pt);
        Ciphertext ct;
        senc.encrypt(pt, ct);
        stringstream cs;
        ct.save(cs);
        const size_t words = 2 * 3 * 8192;
        CHECK(cs.str().size() == 32 + 1 + 8 + 8 + 8 + 8 + 8 + words * 8, "Ciphertext::save: 73 header bytes + the words");
        Ciphertext ct2, ct3;
        ct2.load(sctx, cs);
        cs.seekg(0);
        ct3.unsafe_load(cs);  // the 3.4.x spelling: the context is found by the stream's parms_id
        bool same = ct2.size() == 2 && ct2.parms_id() == ct.parms_id() && ct2.scale() == ct.scale() && ct2.is_ntt_form();
        for (size_t i = 0; same && i < words; ++i) same = ct2.data()[i] == ct.data()[i] && ct3.data()[i] == ct.data()[i];
        CHECK(same, "Ciphertext save -> load / unsafe_load: the same words, level and scale");
        stringstream pss;
        pt.save(pss);
        Plaintext pt2;
        pt2.load(sctx, pss);
        vector<double> back;
        scod.decode(pt2, back);
        CHECK(pt2.parms_id() == pt.parms_id() && pt2.scale() == pt.scale() && fabs(back[1] + 2.25) < 1e-6 && fabs(back[3] - 0.125) < 1e-6,
              "Plaintext save -> load decodes to the same values");
        // keys through a stream, then the whole pipeline on the loaded ones: encrypt under the loaded public key, rotate with
        // the loaded Galois keys, square + relinearise with the loaded relinearisation key, decrypt with the loaded secret key
        stringstream k1, k2, k3, k4;
        spk.save(k1);
        ssk.save(k2);
        srk.save(k3);
        sgk.save(k4);
        PublicKey lpk;
        SecretKey lsk;
        RelinKeys lrk;
        GaloisKeys lgk;
        lpk.load(sctx, k1);
        lsk.load(sctx, k2);
        lrk.load(sctx, k3);
        lgk.load(sctx, k4);
        CHECK(lgk.size() == 3 && lgk.has_key(shim::galois_elt_from_step(4, 8192)) && lrk.has_key(0u) && lsk.host == ssk.host,
              "RelinKeys / GaloisKeys / SecretKey save -> load keep their entries");
        Ciphertext r1, r2, q1, q2;
        sev.rotate_vector(ct, 1, sgk, r1);
        sev.rotate_vector(ct, 1, lgk, r2);
        sev.square(ct, q1);
        sev.relinearize_inplace(q1, srk);
        sev.square(ct, q2);
        sev.relinearize_inplace(q2, lrk);
        same = true;
        for (size_t i = 0; same && i < words; ++i) same = r1.data()[i] == r2.data()[i] && q1.data()[i] == q2.data()[i];
        CHECK(same, "evaluation with the loaded keys gives the bits of the original keys");
        Encryptor lenc(sctx, lpk);
        Decryptor ldec(sctx, lsk);
        Ciphertext lc;
        lenc.encrypt(pt, lc);
        sev.rotate_vector_inplace(lc, 1, lgk);
        Plaintext lp;
        ldec.decrypt(lc, lp);
        scod.decode(lp, back);
        CHECK(fabs(back[0] + 2.25) < 1e-4 && fabs(back[2] - 0.125) < 1e-4, "encrypt / rotate / decrypt entirely on loaded keys");
        // a stream of another parameter set, and a damaged one, are refused
        EncryptionParameters op(scheme_type::CKKS);
        op.set_poly_modulus_degree(8192);
        op.set_coeff_modulus(CoeffModulus::Create(8192, {50, 30, 50}));
        auto octx = SEALContext::Create(op);
        cs.clear();
        cs.seekg(0);
        Ciphertext bad;
        CHECK(throws_invalid([&] { bad.load(octx, cs); }, "invalid"), "load refuses a ciphertext of other parameters");
        string raw = cs.str();
        raw[73 + 8 + 7] = (char)0xff;  // top byte of the first word: no longer a residue of the 60-bit prime
        stringstream dm(raw);
        CHECK(throws_invalid([&] { bad.load(sctx, dm); }, "invalid"), "load refuses a word that is not a residue (unsafe_load would take it)");
    }

    cout << (failures ? "SELFTEST FAILED" : "SELFTEST PASSED") << " (" << failures << " failures)" << endl;
    return failures ? 1 : 0;
}
