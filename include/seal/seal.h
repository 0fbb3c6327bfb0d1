// seal/seal.h -- header-only shim that offers the part of Microsoft SEAL's C++ API used by
// MarwanNour/SEAL-FYP-Logistic-Regression (the union of every seal:: call in the
// reference, in BOTH its SEAL 3.4.5 spelling -- SEALContext::Create, scheme_type::CKKS, keygen.galois_keys() --
// and its 3.6 spelling -- SEALContext context(parms), scheme_type::ckks, keygen.create_galois_keys(gk)) on top
// of the hefx C-ABI (include/hefx.h, gfx950 HIP kernels).  The reference's drivers compile against this header
// unchanged and link with -lhefx; see INTEGRATION.md.
//
// What runs where: every homomorphic operation (Evaluator::*) and every NTT dispatches HIP kernels through
// hefx_*; ciphertext / plaintext / key payloads live in device memory (immutable buffers shared by value-
// semantic handles, so SEAL's copy-heavy call style -- helper.h:237 passes GaloisKeys BY VALUE -- costs
// nothing).  Host-side: parameter bookkeeping (parms_id chain, scale), SEAL's validity checks and exception
// types/messages, NAF decomposition of rotation steps (App. A.7), random sampling and the complex FFT of
// CKKSEncoder.  BFV-only classes exist so that the drivers compile; they throw std::logic_error when used.
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <chrono>
#include <ctime>
#include <cstdio>
#include <initializer_list>
#include <functional>
#include <map>
#include <unordered_map>
#include <tuple>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../hefx.h"
#include "shim_bfv.h"
#include "shim_io.h"

namespace seal {

// ------------------------------------------------------------------------------------------------
// basic types
// ------------------------------------------------------------------------------------------------
enum class scheme_type : std::uint8_t { none = 0, BFV = 1, bfv = 1, CKKS = 2, ckks = 2 };
using parms_id_type = std::array<std::uint64_t, 4>;
static const parms_id_type parms_id_zero = {0, 0, 0, 0};

class SmallModulus {
public:
    SmallModulus(std::uint64_t v = 0) : value_(v) {}
    std::uint64_t value() const { return value_; }
    int bit_count() const
    {
        int b = 0;
        for (std::uint64_t v = value_; v; v >>= 1) ++b;
        return b;
    }
    bool is_zero() const { return value_ == 0; }
    // SEAL 3.4.5 SmallModulus::save / load: the value, one uint64 (format notes: shim_io.h)
    void save(std::ostream &stream) const
    {
        shim::StreamGuard g(stream);
        shim::put_u64(stream, value_);
    }
    void load(std::istream &stream)
    {
        shim::StreamGuard g(stream);
        value_ = shim::get_u64(stream);
    }
    bool operator==(const SmallModulus &o) const { return value_ == o.value_; }
    bool operator!=(const SmallModulus &o) const { return value_ != o.value_; }

private:
    std::uint64_t value_;
};
using Modulus = SmallModulus;  // SEAL >= 3.5 name

namespace shim {

typedef unsigned __int128 u128;
inline std::uint64_t mulmod(std::uint64_t a, std::uint64_t b, std::uint64_t q) { return (std::uint64_t)(((u128)a * b) % q); }
inline std::uint64_t powmod(std::uint64_t a, std::uint64_t e, std::uint64_t q)
{
    std::uint64_t r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = mulmod(r, a, q);
        a = mulmod(a, a, q);
    }
    return r;
}
inline bool is_prime(std::uint64_t n)
{
    if (n < 2) return false;
    static const std::uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (auto p : bases) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    std::uint64_t d = n - 1;
    int r = 0;
    while (!(d & 1)) d >>= 1, ++r;
    for (auto a : bases) {
        std::uint64_t x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < r && comp; ++i) {
            x = mulmod(x, x, n);
            if (x == n - 1) comp = false;
        }
        if (comp) return false;
    }
    return true;
}

[[noreturn]] inline void raise(int rc)
{
    const std::string msg = hefx_last_error();
    if (rc == HEFX_ERR_INVALID) throw std::invalid_argument(msg);
    if (rc == HEFX_ERR_TRANSPARENT) throw std::logic_error(msg);
    throw std::runtime_error("hefx: " + msg);
}
// SEAL_SHIM_SYNC=1: wait for the device after every engine call, so that a caller's own wall-clock timers (the
// reference brackets its calls with chrono) measure completed work, as they would with SEAL's synchronous CPU code.
// Default off: calls return as soon as their launches are queued.
inline hefx_context *&sync_target()
{
    static thread_local hefx_context *ctx = nullptr;
    return ctx;
}
// wall-clock origin of the shim's timeline prints (SEAL_SHIM_STATS=2): the first time anything of the shim runs
inline std::chrono::steady_clock::time_point process_epoch()
{
    static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    return t0;
}
inline bool sync_mode()
{
    static const bool on = [] {
        const char *s = std::getenv("SEAL_SHIM_SYNC");
        return s && *s && *s != '0';
    }();
    return on;
}
// (inside a submission the wait is held back: its batches are ordered on the device anyway, and Engine::flush waits once
// when the last of them is queued -- the caller's timer sees the same completed work, without the host and the device
// taking turns after every batch: the reference's linear_transformation.cpp at d = 1000 3.8 -> 2.9 ms in its own timer)
inline int &sync_hold()
{
    static thread_local int depth = 0;
    return depth;
}
inline void check(int rc)
{
    if (rc != HEFX_OK) raise(rc);
    if (sync_mode() && sync_target() && sync_hold() == 0) (void)hefx_stream_sync(sync_target(), nullptr);
}

// One engine context per distinct (N, primes); created on first use and kept for the life of the process, so the
// reference's habit of rebuilding SEALContext + Evaluator inside every Linear_Transform_Plain call
// (helper.h:239-240) costs a map lookup.
struct Buf;
using BufPtr = std::shared_ptr<Buf>;

struct Engine {
    hefx_context *ctx_raw = nullptr;
    std::uint32_t n = 0;
    std::vector<std::uint64_t> primes;
    std::mutex mu;

    // ---- several devices behind one SEAL program (SEAL_SHIM_DEVICES=n) ------------------------------------------
    // The reference's callers are single-process C++ whose loops are embarrassingly parallel: 2000 independent
    // observation rows (logistic_regression_ckks.cpp:217-229), 2(n-1) independent linear transforms
    // (matrix_mult_benchmark.cpp:41-43).  The recorder below already holds them as a dependency graph, so a submission
    // cuts the graph into its connected sub-graphs, deals them over `ndev` engine contexts (one per GPU; context 0 is
    // the program's home device where every payload lives), replicates the external inputs a sub-graph reads (keys and
    // shared ciphertexts once, kept while their owner lives), runs all devices' batches concurrently and copies back
    // the results somebody outside the graph still holds (hefx_copy_peer = hipMemcpyPeerAsync).  Every node is a
    // deterministic function of its inputs: same bits on any device.  With fewer GPUs than SEAL_SHIM_DEVICES the extra
    // contexts share the existing ones (d mod hefx_device_count()) -- how the path is exercised on a one-GPU box.
    struct Node;
    int ndev = 1;
    std::vector<hefx_context *> dev_ctx;  // [ndev], dev_ctx[0] == ctx_raw; created on first multi-device submission
    struct Replica {
        std::weak_ptr<Buf> owner;  // the home buffer; a recycled address with another owner is a miss
        std::uint64_t *p;
        std::size_t words;
        std::uint64_t used;        // submission counter of the last use (least recently used replicas go first)
    };
    std::vector<std::map<const std::uint64_t *, Replica>> replicas;  // per device > 0: home pointer -> copy
    std::vector<std::size_t> replica_bytes;                          // per device: bytes held by its replicas
    std::size_t replica_budget = (std::size_t)16384 << 20;           // per device (SEAL_SHIM_REPLICA_MB)
    std::uint64_t submission = 0;
    // frees replicas of device d, least recently used first, until `need` more bytes fit the budget (or, with
    // everything = true, all that this submission does not use); returns the bytes released
    inline std::size_t evict_replicas(int d, std::size_t need, bool everything);
    ~Engine()
    {
        for (std::size_t d = 1; d < dev_ctx.size(); ++d) {
            if (!dev_ctx[d]) continue;
            (void)hefx_stream_sync(dev_ctx[d], nullptr);
            if (d < replicas.size())
                for (auto &r : replicas[d]) (void)hefx_free(dev_ctx[d], r.second.p);
            hefx_context_destroy(dev_ctx[d]);
        }
        // the home context outlives the payload buffers that still point at it (process-lifetime registry): not destroyed here
    }
    inline hefx_context *device_context(int d);
    struct Fusion;
    inline void flush_multi(std::vector<Node> &K, const Fusion &fz, int max_depth);

    // ---- deferred evaluation ------------------------------------------------------------------------------
    // The reference issues its ciphertext operations one call at a time -- helper.h:252-257 (rotate_vector,
    // multiply_plain, next diagonal ...), and logistic_regression_ckks.cpp:217-229 -> helper.h:432-476 (per observation
    // row: multiply, relinearize, rescale, rotate, add, then `size` times rotate-by-1 + add) -- which would keep every
    // launch at batch size 1 (70-100 us of dependent-kernel latency per key switch).  The evaluator members that make
    // up these loops are therefore RECORDED as nodes of a small dependency graph: the result buffer exists at once
    // (payloads are immutable shared device buffers), only its contents are late; level / scale / size bookkeeping and
    // every SEAL validity check happen eagerly, at the call, with SEAL's exceptions.  The graph runs when a result is
    // observed (decrypt, add_many, a download, any non-recorded member reading a recorded result) or when the recorded
    // results to be stored exceed SEAL_SHIM_PENDING_MB (default 8192): nodes go to the device by dependency depth, all nodes of one
    // depth, kind and level as ONE batched C-ABI call (hefx_apply_galois_batch, hefx_rotate_multiply_plain_batch,
    // hefx_multiply_batch, hefx_relinearize_batch, hefx_rescale_to_next_batch, hefx_add_batch, ...) -- the 2000 dot
    // products of the LR loop advance in lockstep.  Equal (source, Galois element, key) rotations are computed once; a
    // multiply_plain is fused into the key switch that feeds it when nobody else holds the rotated input.  Same bits
    // as immediate execution: every node is a deterministic function of its inputs.  SEAL_SHIM_LAZY=0 turns it off.
    struct Node {
        // (ENCODE and ENCRYPT first: the groups of one depth are submitted in this order, producers of plaintexts ahead
        // of the products that read them)
        enum Kind { ENCODE, ENCRYPT, ROT, MULPT, MULCT, RELIN, RESCALE, ADD, SUB };
        Kind kind;
        BufPtr a, b, dst;     // a: ciphertext (ENCRYPT: the plaintext; ENCODE: none); b: ciphertext (MULCT/ADD/SUB),
                              // plaintext (MULPT), key (ROT/RELIN) or public key (ENCRYPT)
        std::uint32_t elt;    // ROT
        int L, size;          // rows and size of the input ciphertext(s); ENCODE: size = number of slot values
        int depth;
        int consumers;        // recorded nodes that read dst
        int mulpt_consumer;   // a MULPT node reading dst (fusion candidate), -1 if none
        // ENCODE: the slot values (copied at the call: the caller's vector may change) and the scale;
        // ENCRYPT: the encryptor's sampler key and the stream id this encryption drew
        std::shared_ptr<std::vector<double>> host;
        double scale = 0;
        std::shared_ptr<std::array<std::uint8_t, 32>> skey;
        std::uint64_t stream_id = 0;
    };
    std::vector<Node> pend;
    struct CseKey {
        const Buf *a, *b;
        std::uint32_t elt;
        bool operator==(const CseKey &o) const { return a == o.a && b == o.b && elt == o.elt; }
    };
    struct CseHash {
        std::size_t operator()(const CseKey &k) const
        {
            return (std::size_t)(((std::uintptr_t)k.a >> 4) * 0x9E3779B97F4A7C15ull ^ ((std::uintptr_t)k.b >> 4) * 0xC2B2AE3D27D4EB4Full ^
                                 (std::uint64_t)k.elt * 0x165667B19E3779F9ull);
        }
    };
    struct PtrHash {
        std::size_t operator()(const Buf *p) const { return (std::size_t)(((std::uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull); }
    };
    // result buffer -> node: the index rides in the Buf itself (Buf::pend_idx, valid while Buf::pend_epoch == epoch_ of its
    // engine) -- a hash-map insert and lookup per recorded call was a fifth of the recording time of a 1000-step transform
    std::uint32_t epoch_ = 1;
    inline int pend_index(const Buf *b) const;
    std::unordered_map<CseKey, int, CseHash> pend_cse;               // rotations: (source, element, key) -> node
    std::size_t pend_bytes = 0, pend_budget = (std::size_t)8192 << 20, pend_check = (std::size_t)8192 << 20;
    std::string failed;  // a batched call of flush() failed: results recorded with it are garbage
    struct Stats {       // SEAL_SHIM_STATS=1 prints them when the process ends
        std::size_t flushes = 0, nodes = 0, calls = 0, levels = 0;
        double seconds = 0;
        // SEAL_SHIM_STATS=3: where a submission's host time goes (seconds): fusion plan, grouping, result buffers, engine calls
        double t_plan = 0, t_group = 0, t_alloc = 0, t_calls = 0;
    } stats;
    bool lazy = true;
    bool pending(const Buf *b) const;
    // the context with everything recorded so far submitted
    hefx_context *live()
    {
        if (!pend.empty()) flush();
        if (!failed.empty()) throw std::runtime_error("a deferred evaluator operation failed earlier: " + failed);
        sync_target() = ctx_raw;
        return ctx_raw;
    }
    // the context for an operation that reads only `bufs`: nothing is submitted unless one of them is a recorded result
    // (encode, encrypt, key generation, plaintext mod_switch ... do not interrupt a lockstep batch)
    hefx_context *ready(std::initializer_list<const Buf *> bufs)
    {
        for (const Buf *b : bufs)
            if (b && pending(b)) return live();
        if (!failed.empty()) throw std::runtime_error("a deferred evaluator operation failed earlier: " + failed);
        sync_target() = ctx_raw;
        return ctx_raw;
    }
    BufPtr record(Node::Kind kind, const BufPtr &a, const BufPtr &b, std::uint32_t elt, int L, int size,
                  std::size_t out_words, const std::shared_ptr<Engine> &self, const Node *extra = nullptr);
    inline void flush();

    // What a submission fuses (all of it invisible in the bits: every fused call runs the same key switches and sums):
    //   rot_mul   ROT + the MULPT that alone reads it            -> hefx_rotate_multiply_plain_batch
    //   rot_add   ROT + an ADD of its result (helper.h:474-475)  -> hefx_apply_galois_add_batch (sum in the epilogue)
    //   chains    runs of rot_add pairs, each feeding the next, whose intermediate rotations and sums nobody outside the
    //             graph holds (helper.h:472-476: the loop reassigns `dup` and `mult`)  -> hefx_rotate_add_chain, all
    //             chains of one start depth, level and length in lockstep (the LR gradient's eight, 2000 levels each)
    struct Fusion {
        std::vector<int> mul_of, add_of;    // per ROT node: the fused MULPT / ADD node, or -1
        std::vector<char> skip;             // the node runs inside another node's call
        std::vector<char> unwritten;        // ... and its result buffer is never written (nobody may read it)
        struct Chain {
            int first_rot, last_rot, last_add, steps;
            const Buf *ct_in, *acc_in;
        };
        std::vector<Chain> chains;
    };
    inline Fusion plan_fusion(const std::vector<Node> &K) const;

    // payload buffers come from the engine's pooled allocator (hefx_malloc / hefx_free: slab-carved, no hipFree and so no
    // device synchronisation on the hot path; all work is ordered on the default stream)
    std::uint64_t *alloc(std::size_t words)
    {
        void *p = nullptr;
        const int rc = hefx_malloc(ctx_raw, words * sizeof(std::uint64_t), &p);
        if (rc != HEFX_OK) raise(rc);
        return static_cast<std::uint64_t *>(p);
    }
    void release(std::uint64_t *p, std::size_t) { (void)hefx_free(ctx_raw, p); }
};

inline std::shared_ptr<Engine> get_engine(std::uint32_t n, const std::vector<std::uint64_t> &primes)
{
    static std::mutex mu;
    (void)process_epoch();
    static auto *registry = new std::map<std::pair<std::uint32_t, std::vector<std::uint64_t>>, std::shared_ptr<Engine>>();
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(n, primes);
    auto it = registry->find(key);
    if (it != registry->end()) return it->second;
    auto e = std::make_shared<Engine>();
    e->n = n;
    e->primes = primes;
    int dev = 0;
    if (const char *d = std::getenv("HEFX_DEVICE")) dev = std::atoi(d);
    check(hefx_context_create(n, primes.data(), (int)primes.size(), dev, &e->ctx_raw));
    if (std::getenv("SEAL_SHIM_STATS") && std::atoi(std::getenv("SEAL_SHIM_STATS")) > 1)
        std::fprintf(stderr, "[seal shim] wall +%.3f s (cpu %.3f s) engine context for N=%u, %zu primes ready\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - process_epoch()).count(),
                     (double)std::clock() / CLOCKS_PER_SEC, n, primes.size());
    if (const char *l = std::getenv("SEAL_SHIM_LAZY")) e->lazy = std::atoi(l) != 0;
    if (const char *nd = std::getenv("SEAL_SHIM_DEVICES")) e->ndev = std::max(1, std::min(64, std::atoi(nd)));
    // the one SEAL semantic that could not be verified offline: rescale_to_next divides with
    // round-to-nearest by default (the engine's default since round 6: Evaluator::mod_switch_scale_to_next calling
    // BaseConverter::round_last_coeff_modulus_ntt_inplace); SEAL_SHIM_RESCALE=floor selects the
    // floor division App. A.9 reads 3.4.x as
    if (const char *r = std::getenv("SEAL_SHIM_RESCALE"))
        check(hefx_set_rescale_mode(e->ctx_raw, std::string(r) == "floor" ? HEFX_RESCALE_FLOOR : HEFX_RESCALE_ROUND));
    {   // results a submission will STORE may reach 8 GiB (a quarter of the device on a small one) before it is forced;
        // elided results do not count (record()).  profiles/r04/lr_driver_pending_budget.txt has the measurements that led
        // here: with every recorded result counted and allocated, 8 GiB cut the LR driver's eight gradient chains across
        // two submissions (0.94 s), 16 GiB kept them together (0.69 s) but made every second back-to-back run wait 1.4 s
        // for the driver to scrub the previous process's 20 GB, and 64 GiB spent 2.5 s allocating.
        std::size_t fr = 0, tot = 0;
        if (hefx_device_memory(e->ctx_raw, &fr, &tot) == HEFX_OK && tot)
            e->pend_budget = std::min<std::size_t>((std::size_t)8 << 30, std::max<std::size_t>((std::size_t)1 << 30, tot / 4));
        e->pend_check = e->pend_budget;
    }
    if (const char *m = std::getenv("SEAL_SHIM_PENDING_MB")) e->pend_check = e->pend_budget = (std::size_t)std::strtoull(m, nullptr, 10) << 20;
    if (const char *m = std::getenv("SEAL_SHIM_REPLICA_MB")) e->replica_budget = (std::size_t)std::strtoull(m, nullptr, 10) << 20;
    if (const char *st = std::getenv("SEAL_SHIM_STATS")) {
        if (std::atoi(st)) {
            static std::vector<std::shared_ptr<Engine>> *watched = new std::vector<std::shared_ptr<Engine>>();
            if (watched->empty())
                std::atexit([] {
                    for (auto &w : *watched)
                        std::fprintf(stderr,
                                     "[seal shim] N=%u: %zu recorded operations ran in %zu submissions (%zu dependency levels, "
                                     "%zu batched C-ABI calls), %.3f s of host time submitting\n",
                                     w->n, w->stats.nodes, w->stats.flushes, w->stats.levels, w->stats.calls, w->stats.seconds);
                });
            watched->push_back(e);
        }
    }
    (*registry)[key] = e;
    return e;
}

// immutable device payload.  The device memory behind it is allocated on FIRST USE of `p` (round 4): a recorded node's
// result buffer exists as a handle from the moment of the call, but nodes that a submission runs inside another call and
// never stores -- the rotation inside a fused product, the 2 x 1998 intermediate rotations and sums of every gradient chain
// of the LR driver (16 GB at N = 16384) -- never ask for their address and so never take pool memory.
struct Buf {
    std::shared_ptr<Engine> eng;
    std::size_t words = 0;
    struct Ptr {
        Buf *b;
        operator std::uint64_t *() const { return b->get(); }
        std::uint64_t *operator+(std::size_t off) const { return b->get() + off; }
    } p;
    Buf(std::shared_ptr<Engine> e, std::size_t w) : eng(std::move(e)), words(w), p{this} {}
    ~Buf()
    {
        if (addr_) eng->release(addr_, words);
    }
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    std::uint64_t *get()
    {
        if (!addr_) addr_ = eng->alloc(words);
        return addr_;
    }
    bool allocated() const { return addr_ != nullptr; }
    // the recorder's back-pointer: node index of the pending operation that produces this buffer (Engine::pend_index)
    int pend_idx = -1;
    std::uint32_t pend_epoch = 0;

private:
    std::uint64_t *addr_ = nullptr;
};
inline int Engine::pend_index(const Buf *b) const { return b->eng.get() == this && b->pend_epoch == epoch_ ? b->pend_idx : -1; }
inline BufPtr new_buf(const std::shared_ptr<Engine> &e, std::size_t words) { return std::make_shared<Buf>(e, words); }

inline bool Engine::pending(const Buf *b) const { return pend_index(b) >= 0; }

inline BufPtr Engine::record(Node::Kind kind, const BufPtr &a, const BufPtr &b, std::uint32_t elt, int L, int size,
                             std::size_t out_words, const std::shared_ptr<Engine> &self, const Node *extra)
{
    if (kind == Node::ROT) {  // the same rotation of the same buffer with the same key: computed once
        auto hit = pend_cse.find(CseKey{a.get(), b.get(), elt});
        if (hit != pend_cse.end() && pend[hit->second].L == L) return pend[hit->second].dst;
    }
    Node nd{kind, a, b, new_buf(self, out_words), elt, L, size, 0, 0, -1};
    if (extra) {
        nd.host = extra->host;
        nd.scale = extra->scale;
        nd.skey = extra->skey;
        nd.stream_id = extra->stream_id;
    }
    const int idx = (int)pend.size();
    auto link = [&](const BufPtr &in, bool ct_input) {
        if (!in) return;
        const int pi = pend_index(in.get());
        if (pi < 0) return;
        Node &src = pend[pi];
        nd.depth = std::max(nd.depth, src.depth + 1);
        ++src.consumers;
        if (kind == Node::MULPT && ct_input) src.mulpt_consumer = idx;
    };
    link(a, true);
    if (kind == Node::MULCT || kind == Node::ADD || kind == Node::SUB) {
        if (b.get() != a.get()) link(b, true);
    } else if (kind == Node::MULPT) {
        link(b, false);  // the plaintext may be a recorded encode
    }
    BufPtr out = nd.dst;
    pend.push_back(std::move(nd));
    out->pend_idx = idx;
    out->pend_epoch = epoch_;
    if (kind == Node::ROT) pend_cse[CseKey{a.get(), b.get(), elt}] = idx;
    pend_bytes += out_words * 8;
    // bounded memory: run what is recorded once the results it will STORE exceed the budget.  Result buffers take memory
    // only when a submission writes them (Buf is lazy), and the nodes a submission elides -- chain intermediates, rotations
    // inside fused products -- never do: the eight 2000-level gradient chains of the LR driver record 16 GB of results and
    // store 8 ciphertexts.  So when the recorded total passes the mark, the fusion plan says what would really be written.
    if (pend_bytes > pend_check || pend.size() > 400000) {
        std::size_t need = 0;
        if (pend.size() <= 400000) {
            const Fusion fz = plan_fusion(pend);
            for (std::size_t i = 0; i < pend.size(); ++i)
                if (!fz.unwritten[i]) need += pend[i].dst->words * 8;
        }
        if (need > pend_budget || pend.size() > 400000)
            flush();
        else
            pend_check = pend_bytes + pend_budget / 4;  // look again after another quarter budget of recording
    }
    // SEAL_SHIM_SYNC=1 promises a caller's chrono timers completed work: only the rotations and the products of the
    // linear-transform loops (helper.h:216-229, 252-257) stay recorded there -- their add_many observes them inside the
    // timed region; relinearize / rescale / add run (and are waited for) at the call
    if (sync_mode() && kind != Node::ROT && kind != Node::MULPT && kind != Node::MULCT && !pend.empty()) {
        flush();
        (void)hefx_stream_sync(ctx_raw, nullptr);
    }
    return out;
}

// One device's share of a submission: its nodes of K (by dependency depth), depths depth_first .. depth_last, all nodes
// of one depth, kind and level as ONE batched C-ABI call on `cx`.  in(p) translates an input pointer of the home device
// to this device, out(i) is where node i's result goes.
template <class In, class Out>
inline void submit_nodes(hefx_context *cx, Engine::Stats &stats, const std::vector<Engine::Node> &K,
                         const std::vector<std::vector<int>> &by_depth, const Engine::Fusion &fz, int depth_first,
                         int depth_last, const In &in, const Out &out);
inline std::vector<std::vector<int>> nodes_by_depth(const std::vector<Engine::Node> &K, const std::vector<int> &ids, int max_depth)
{
    std::vector<std::vector<int>> by_depth(max_depth + 1);
    for (int i : ids) by_depth[K[i].depth].push_back(i);
    return by_depth;
}

inline void Engine::flush()
{
    std::vector<Node> K;
    K.swap(pend);
    pend.reserve(std::min<std::size_t>(K.capacity(), (std::size_t)1 << 17));  // the next recording does not regrow from nothing
    ++epoch_;  // nothing is pending any more
    pend_cse.clear();
    pend_bytes = 0;
    pend_check = pend_budget;
    if (K.empty()) return;
    const auto t_start = std::chrono::steady_clock::now();
    int max_depth = 0;
    for (const Node &k : K) max_depth = std::max(max_depth, k.depth);
    ++stats.flushes;
    stats.nodes += K.size();
    stats.levels += (std::size_t)max_depth + 1;
    const Fusion fz = plan_fusion(K);
    stats.t_plan += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    struct SyncHold {  // SEAL_SHIM_SYNC: one wait at the end of the submission instead of one per engine call (check())
        SyncHold() { ++sync_hold(); }
        ~SyncHold() { --sync_hold(); }
    };
    try {
        SyncHold hold;
        if (ndev > 1)
            flush_multi(K, fz, max_depth);
        else {
            std::vector<int> all(K.size());
            for (std::size_t i = 0; i < K.size(); ++i) all[i] = (int)i;
            submit_nodes(ctx_raw, stats, K, nodes_by_depth(K, all, max_depth), fz, 0, max_depth,
                         [](const Buf *b_) -> const std::uint64_t * { return const_cast<Buf *>(b_)->get(); },
                   [&](int i) -> std::uint64_t * { return K[i].dst->get(); });
        }
    } catch (const std::exception &ex) {
        // results recorded with the failed batch (and everything after it) were never computed: every later use of
        // this engine reports it instead of handing out garbage
        failed = ex.what();
        throw;
    }
    if (sync_mode() && sync_hold() == 0) {
        (void)hefx_stream_sync(ctx_raw, nullptr);
        for (std::size_t d = 1; d < dev_ctx.size(); ++d)
            if (dev_ctx[d]) (void)hefx_stream_sync(dev_ctx[d], nullptr);
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    stats.seconds += dt;
    static const bool verbose = std::getenv("SEAL_SHIM_STATS") && std::atoi(std::getenv("SEAL_SHIM_STATS")) > 1;
    if (verbose)
        std::fprintf(stderr, "[seal shim] wall +%.3f s (cpu %.3f s) submission %zu: %zu operations, %d dependency levels, %.1f ms host time\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - process_epoch()).count(),
                     (double)std::clock() / CLOCKS_PER_SEC, stats.flushes, K.size(), max_depth + 1, dt * 1e3);
    if (verbose && std::atoi(std::getenv("SEAL_SHIM_STATS")) > 2)
        std::fprintf(stderr, "[seal shim]   so far: fusion plan %.2f ms, grouping + operand lists %.2f ms, result buffers %.2f ms, engine calls %.2f ms\n",
                     stats.t_plan * 1e3, stats.t_group * 1e3, stats.t_alloc * 1e3, stats.t_calls * 1e3);
}
template <class In, class Out>
inline void submit_nodes(hefx_context *cx, Engine::Stats &stats, const std::vector<Engine::Node> &K,
                         const std::vector<std::vector<int>> &by_depth, const Engine::Fusion &fz, int depth_first,
                         int depth_last, const In &in, const Out &out)
{
    using Node = Engine::Node;
    std::vector<const std::uint64_t *> va, vb, vc;
    std::vector<std::uint64_t *> vo, vo2;
    std::vector<std::uint32_t> ve;
    static const bool timing = std::getenv("SEAL_SHIM_STATS") && std::atoi(std::getenv("SEAL_SHIM_STATS")) > 2;
    static const bool merge_rot = !(std::getenv("SEAL_SHIM_MERGE_ROT") && std::atoi(std::getenv("SEAL_SHIM_MERGE_ROT")) == 0);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    for (int depth = depth_first; depth <= depth_last; ++depth) {
        if (by_depth[depth].empty()) continue;
        auto t_g0 = now();
        // groups of this depth: (kind, fusion, L, size, shared key); fusion of a ROT: 0 none, 1 + multiply_plain, 2 + add
        std::map<std::tuple<int, int, int, int, const Buf *>, std::vector<int>> groups;
        // chains that start at this depth, in lockstep per (level, length): (L, steps) -> chain indices
        std::map<std::pair<int, int>, std::vector<int>> chain_groups;
        for (int i : by_depth[depth]) {
            const Node &k = K[i];
            if (fz.skip[i]) continue;  // runs inside another node's call
            const Buf *shared = (k.kind == Node::RELIN || k.kind == Node::ENCRYPT) ? k.b.get() : nullptr;
            int f = 0;
            if (k.kind == Node::ROT) f = fz.mul_of[i] >= 0 ? 1 : (fz.add_of[i] >= 0 ? 2 : 0);
            // the plain rotations of a depth ride in the batch of its fused products, as items without a plaintext
            // (hefx_rotate_multiply_plain_batch takes null entries): one launch sequence per depth of a NAF forest
            // instead of two
            if (k.kind == Node::ROT && f == 0 && merge_rot) f = 1;
            groups[std::make_tuple((int)k.kind, f, k.L, k.size, shared)].push_back(i);
        }
        for (std::size_t ci = 0; ci < fz.chains.size(); ++ci) {
            const auto &ch = fz.chains[ci];
            if (K[ch.first_rot].depth != depth) continue;
            // flush_multi hands every device only its own nodes: a chain belongs to the device of its first rotation
            if (std::find(by_depth[depth].begin(), by_depth[depth].end(), ch.first_rot) == by_depth[depth].end()) continue;
            chain_groups[{K[ch.first_rot].L, ch.steps}].push_back((int)ci);
        }
        for (auto &g : chain_groups) {
            const int L = g.first.first, steps = g.first.second, n = (int)g.second.size();
            va.clear(), vb.clear(), vc.clear(), vo.clear(), vo2.clear(), ve.clear();
            ++stats.calls;
            for (int ci : g.second) {
                const auto &ch = fz.chains[ci];
                va.push_back(in(ch.ct_in));
                vb.push_back(in(K[ch.first_rot].b.get()));
                ve.push_back(K[ch.first_rot].elt);
                vc.push_back(in(ch.acc_in));
                vo.push_back(out(ch.last_rot));
                vo2.push_back(out(ch.last_add));
            }
            check(hefx_rotate_add_chain(cx, L, n, va.data(), ve.data(), vb.data(), vc.data(), vo2.data(), vo.data(), steps, nullptr));
        }
        if (timing) stats.t_group += secs(t_g0, now());
        for (auto &g : groups) {
            const int kind = std::get<0>(g.first), f = std::get<1>(g.first), L = std::get<2>(g.first),
                      size = std::get<3>(g.first);
            const std::vector<int> &gi = g.second;
            const int n = (int)gi.size();
            va.clear(), vb.clear(), vc.clear(), vo.clear(), vo2.clear(), ve.clear();
            ++stats.calls;
            auto t_l0 = now();
            if (timing)  // result buffers first, so that their allocation is timed apart from the operand lists
                for (int i : gi) {
                    if (kind == Node::ROT && f == 1) (void)out(fz.mul_of[i] >= 0 ? fz.mul_of[i] : i);
                    else if (kind == Node::ROT && f == 2) (void)out(i), (void)out(fz.add_of[i]);
                    else (void)out(i);
                }
            auto t_l1 = now();
            if (timing) stats.t_alloc += secs(t_l0, t_l1);
            for (int i : gi) {
                const Node &k = K[i];
                if (kind == Node::ENCODE || kind == Node::ENCRYPT) continue;  // gather their own operands below
                va.push_back(in(k.a.get()));
                if (k.b) vb.push_back(in(k.b.get()));
                ve.push_back(k.elt);
                if (kind == Node::ROT && f == 1 && fz.mul_of[i] < 0) {  // a plain rotation in the fused batch
                    vc.push_back(nullptr);
                    vo.push_back(out(i));
                } else if (kind == Node::ROT && f == 1) {
                    const Node &m = K[fz.mul_of[i]];
                    vc.push_back(in(m.b.get()));
                    vo.push_back(out(fz.mul_of[i]));
                } else if (kind == Node::ROT && f == 2) {
                    const Node &ad = K[fz.add_of[i]];
                    vc.push_back(in(ad.a.get() == k.dst.get() ? ad.b.get() : ad.a.get()));  // the sum's other operand
                    vo.push_back(out(i));
                    vo2.push_back(out(fz.add_of[i]));
                } else {
                    vo.push_back(out(i));
                }
            }
            auto t_c0 = now();
            if (timing) stats.t_group += secs(t_l1, t_c0);
            struct CallTimer {
                bool on;
                double &acc;
                std::chrono::steady_clock::time_point t0;
                int kind, f, n, depth;
                ~CallTimer()
                {
                    if (!on) return;
                    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    acc += dt;
                    static const bool each = std::atoi(std::getenv("SEAL_SHIM_STATS")) > 3;
                    if (each) std::fprintf(stderr, "[seal shim]     depth %d kind %d fusion %d, %d operations: engine call %.1f us\n", depth, kind, f, n, dt * 1e6);
                }
            } call_timer{timing, stats.t_calls, t_c0, kind, f, n, depth};
            switch (kind) {
                case Node::ENCODE: {  // all vectors of one (level, length), scale by scale: one engine call each
                    std::map<double, std::vector<int>> by_scale;
                    for (int i : gi) by_scale[K[i].scale].push_back(i);
                    for (auto &bs : by_scale) {
                        const std::size_t nv = (std::size_t)size;
                        std::vector<double> vals(nv * bs.second.size());
                        std::vector<std::uint64_t *> outs;
                        for (std::size_t t = 0; t < bs.second.size(); ++t) {
                            std::memcpy(vals.data() + t * nv, K[bs.second[t]].host->data(), nv * sizeof(double));
                            outs.push_back(out(bs.second[t]));
                        }
                        check(hefx_ckks_encode_batch(cx, L, vals.data(), nullptr, size, (int)bs.second.size(), bs.first,
                                                     outs.data(), nullptr));
                    }
                    break;
                }
                case Node::ENCRYPT: {  // runs of consecutive stream ids of one encryptor: one engine call each
                    std::vector<int> order(gi);
                    std::sort(order.begin(), order.end(), [&](int x, int y) {
                        if (*K[x].skey != *K[y].skey) return *K[x].skey < *K[y].skey;
                        return K[x].stream_id < K[y].stream_id;
                    });
                    for (std::size_t t0 = 0; t0 < order.size();) {
                        std::size_t t1 = t0 + 1;
                        while (t1 < order.size() && *K[order[t1]].skey == *K[order[t0]].skey &&
                               K[order[t1]].stream_id == K[order[t0]].stream_id + (t1 - t0))
                            ++t1;
                        std::vector<const std::uint64_t *> plains;
                        std::vector<std::uint64_t *> outs;
                        for (std::size_t t = t0; t < t1; ++t) {
                            plains.push_back(K[order[t]].a ? in(K[order[t]].a.get()) : nullptr);
                            outs.push_back(out(order[t]));
                        }
                        check(hefx_encrypt_batch(cx, L, (int)(t1 - t0), in(std::get<4>(g.first)), plains.data(),
                                                 K[order[t0]].skey->data(), K[order[t0]].stream_id, outs.data(), nullptr));
                        t0 = t1;
                    }
                    break;
                }
                case Node::ROT:
                    if (f == 1)
                        check(hefx_rotate_multiply_plain_batch(cx, L, n, va.data(), ve.data(), vb.data(), vc.data(), vo.data(),
                                                               nullptr));
                    else if (f == 2)
                        check(hefx_apply_galois_add_batch(cx, L, n, va.data(), ve.data(), vb.data(), vc.data(), vo2.data(),
                                                          vo.data(), nullptr));
                    else
                        check(hefx_apply_galois_batch(cx, L, n, va.data(), ve.data(), vb.data(), vo.data(), nullptr));
                    break;
                case Node::MULPT:
                    check(hefx_multiply_plain_batch(cx, L, size, n, va.data(), vb.data(), vo.data(), nullptr));
                    break;
                case Node::MULCT:
                    check(hefx_multiply_batch(cx, L, n, va.data(), vb.data(), vo.data(), nullptr));
                    break;
                case Node::RELIN:
                    check(hefx_relinearize_batch(cx, L, n, va.data(), in(std::get<4>(g.first)), vo.data(), nullptr));
                    break;
                case Node::RESCALE:
                    check(hefx_rescale_to_next_batch(cx, L, size, n, va.data(), vo.data(), nullptr));
                    break;
                case Node::ADD:
                    check(hefx_add_batch(cx, L, size, n, va.data(), vb.data(), vo.data(), nullptr));
                    break;
                case Node::SUB:
                    check(hefx_sub_batch(cx, L, size, n, va.data(), vb.data(), vo.data(), nullptr));
                    break;
            }
        }
    }
}

// The fusion plan of a submission (see Engine::Fusion).  O(nodes log nodes) on the host.
inline Engine::Fusion Engine::plan_fusion(const std::vector<Node> &K) const
{
    const int nk = (int)K.size();
    Fusion fz;
    fz.mul_of.assign(nk, -1);
    fz.add_of.assign(nk, -1);
    fz.skip.assign(nk, 0);
    fz.unwritten.assign(nk, 0);
    static const bool fuse_add = !(std::getenv("SEAL_SHIM_FUSE_ADD") && std::atoi(std::getenv("SEAL_SHIM_FUSE_ADD")) == 0);
    static const bool fuse_chain = !(std::getenv("SEAL_SHIM_CHAINS") && std::atoi(std::getenv("SEAL_SHIM_CHAINS")) == 0);
    // a multiply_plain rides in its rotation's epilogue when the rotated ciphertext is visible to nobody else: the
    // rotation node and the product node are its only holders
    for (int i = 0; i < nk; ++i) {
        const Node &k = K[i];
        if (k.kind != Node::ROT || k.consumers != 1 || k.mulpt_consumer < 0 || k.dst.use_count() != 2) continue;
        const Node &m = K[k.mulpt_consumer];
        if (m.kind != Node::MULPT || m.a.get() != k.dst.get() || m.size != 2) continue;
        fz.mul_of[i] = k.mulpt_consumer;
        fz.skip[k.mulpt_consumer] = 1;
        fz.unwritten[i] = 1;  // the rotation inside a fused product is never stored
    }
    if (!fuse_add) return fz;
    std::unordered_map<const Buf *, int, PtrHash> producer;
    producer.reserve((std::size_t)nk * 2);
    for (int i = 0; i < nk; ++i) producer[K[i].dst.get()] = i;
    // the depth at which a node's result EXISTS: its own, or -- for a sum fused into a rotation -- the rotation's
    std::vector<int> eff_depth(nk);
    for (int i = 0; i < nk; ++i) eff_depth[i] = K[i].depth;
    auto depth_of = [&](const BufPtr &b) {  // -1: not produced by this submission
        auto p = producer.find(b.get());
        return p == producer.end() ? -1 : eff_depth[p->second];
    };
    // rot_add: the first ADD (size 2, same level) that reads a rotation's result; its other operand must exist when the
    // rotation runs, i.e. come from outside or from a shallower (effective) depth.  Nodes are recorded after their
    // inputs, so by the time ADD j is looked at every producer of its operands has been decided.
    std::vector<int> rot_of_add(nk, -1);
    for (int j = 0; j < nk; ++j) {
        const Node &ad = K[j];
        if (ad.kind != Node::ADD || ad.size != 2 || ad.a.get() == ad.b.get()) continue;
        for (int side = 0; side < 2; ++side) {
            const BufPtr &rin = side ? ad.b : ad.a, &other = side ? ad.a : ad.b;
            auto p = producer.find(rin.get());
            if (p == producer.end()) continue;
            const int i = p->second;
            const Node &r = K[i];
            if (r.kind != Node::ROT || r.L != ad.L || fz.mul_of[i] >= 0 || fz.add_of[i] >= 0) continue;
            if (depth_of(other) >= r.depth) continue;
            fz.add_of[i] = j;
            rot_of_add[j] = i;
            fz.skip[j] = 1;
            eff_depth[j] = r.depth;
            break;
        }
    }
    if (!fuse_chain) return fz;
    // chains: pair (r', a') continues pair (r, a) when r' rotates r's result with the same element and key, a' adds r' to
    // a's sum, and nobody but these nodes holds r's result (node + a + r' = 3 references) or a's sum (node + a' = 2)
    std::vector<int> next_pair(nk, -1), has_prev(nk, 0);
    std::map<const Buf *, int> rot_reading;  // buffer -> a paired ROT that rotates it (unique or -2)
    for (int i = 0; i < nk; ++i)
        if (K[i].kind == Node::ROT && fz.add_of[i] >= 0) {
            auto ins = rot_reading.emplace(K[i].a.get(), i);
            if (!ins.second) ins.first->second = -2;
        }
    for (int i = 0; i < nk; ++i) {
        if (K[i].kind != Node::ROT || fz.add_of[i] < 0) continue;
        const Node &r = K[i], &a = K[fz.add_of[i]];
        auto nx = rot_reading.find(r.dst.get());
        if (nx == rot_reading.end() || nx->second < 0) continue;
        const int i2 = nx->second;
        const Node &r2 = K[i2], &a2 = K[fz.add_of[i2]];
        if (r2.elt != r.elt || r2.b.get() != r.b.get() || r2.L != r.L) continue;
        const BufPtr &other2 = a2.a.get() == r2.dst.get() ? a2.b : a2.a;
        if (other2.get() != a.dst.get()) continue;
        if (r.dst.use_count() != 3 || a.dst.use_count() != 2) continue;
        next_pair[i] = i2;
        has_prev[i2] = 1;
    }
    for (int i = 0; i < nk; ++i) {
        if (K[i].kind != Node::ROT || fz.add_of[i] < 0 || has_prev[i] || next_pair[i] < 0) continue;
        int steps = 1, last = i;
        while (next_pair[last] >= 0) last = next_pair[last], ++steps;
        if (steps < 3) continue;  // two levels are two fused calls either way
        const Node &a1 = K[fz.add_of[i]];
        Fusion::Chain ch{i, last, fz.add_of[last], steps, K[i].a.get(),
                         (a1.a.get() == K[i].dst.get() ? a1.b : a1.a).get()};
        fz.chains.push_back(ch);
        for (int r = i;; r = next_pair[r]) {  // every node of the chain runs inside the one call of its first rotation
            fz.skip[r] = 1;
            fz.skip[fz.add_of[r]] = 1;
            if (r != last) fz.unwritten[r] = fz.unwritten[fz.add_of[r]] = 1;
            if (r == last) break;
        }
    }
    return fz;
}

inline hefx_context *Engine::device_context(int d)
{
    if (dev_ctx.empty()) {
        dev_ctx.assign((std::size_t)ndev, nullptr);
        dev_ctx[0] = ctx_raw;
        replicas.resize((std::size_t)ndev);
        replica_bytes.assign((std::size_t)ndev, 0);
    }
    if ((int)dev_ctx.size() < ndev) {  // ndev was raised after the first submission (tests)
        dev_ctx.resize((std::size_t)ndev, nullptr);
        replicas.resize((std::size_t)ndev);
        replica_bytes.resize((std::size_t)ndev, 0);
    }
    if (!dev_ctx[d]) {
        const int have = std::max(1, hefx_device_count());
        const int home = hefx_context_device(ctx_raw);
        hefx_context *cx = nullptr;
        check(hefx_context_create(n, primes.data(), (int)primes.size(), (home + d) % have, &cx));
        check(hefx_set_rescale_mode(cx, hefx_get_rescale_mode(ctx_raw)));
        dev_ctx[d] = cx;
    }
    return dev_ctx[d];
}

inline std::size_t Engine::evict_replicas(int d, std::size_t need, bool everything)
{
    auto &rep = replicas[d];
    std::size_t freed = 0;
    while (!rep.empty() && (everything || replica_bytes[d] + need > replica_budget)) {
        auto victim = rep.end();
        for (auto it = rep.begin(); it != rep.end(); ++it)
            if (it->second.used != submission && (victim == rep.end() || it->second.used < victim->second.used)) victim = it;
        if (victim == rep.end()) break;  // everything left is in use by the submission being prepared
        (void)hefx_free(dev_ctx[d], victim->second.p);
        freed += victim->second.words * 8;
        replica_bytes[d] -= victim->second.words * 8;
        rep.erase(victim);
    }
    return freed;
}

inline void Engine::flush_multi(std::vector<Node> &K, const Fusion &fz, int max_depth)
{
    const int nk = (int)K.size();
    // connected sub-graphs of the recorded dependencies (shared EXTERNAL inputs -- keys, the weight ciphertext -- do not
    // connect: they are replicated)
    std::unordered_map<const Buf *, int, PtrHash> producer;
    producer.reserve((std::size_t)nk * 2);
    for (int i = 0; i < nk; ++i) producer[K[i].dst.get()] = i;
    std::vector<int> root(nk);
    for (int i = 0; i < nk; ++i) root[i] = i;
    std::function<int(int)> find = [&](int x) { return root[x] == x ? x : root[x] = find(root[x]); };
    auto unite = [&](int a_, int b_) { root[find(a_)] = find(b_); };
    for (int i = 0; i < nk; ++i)
        for (const BufPtr *inp : {&K[i].a, &K[i].b}) {
            if (!*inp) continue;
            auto p_ = producer.find(inp->get());
            if (p_ != producer.end() && p_->second != i) unite(i, p_->second);
        }
    // cost of a sub-graph = its key switches (everything else is cheap); heaviest first onto the least loaded device
    std::map<int, std::pair<long, std::vector<int>>> comps;
    for (int i = 0; i < nk; ++i) {
        auto &c_ = comps[find(i)];
        c_.first += (K[i].kind == Node::ROT || K[i].kind == Node::RELIN) ? 16 : 1;
        c_.second.push_back(i);
    }
    std::vector<std::pair<long, int>> order;
    for (auto &c_ : comps) order.emplace_back(-c_.second.first, c_.first);
    std::sort(order.begin(), order.end());
    std::vector<long> load((std::size_t)ndev, 0);
    std::vector<std::vector<int>> share((std::size_t)ndev);
    std::vector<int> dev_of(nk, 0);
    for (auto &o : order) {
        const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        load[d] += -o.first;
        for (int i : comps[o.second].second) {
            share[d].push_back(i);
            dev_of[i] = d;
        }
    }
    // references from inside the graph: a result nobody else holds is not copied home
    std::vector<int> inner(nk, 0);
    for (int i = 0; i < nk; ++i)
        for (const BufPtr *inp : {&K[i].a, &K[i].b}) {
            if (!*inp) continue;
            auto p_ = producer.find(inp->get());
            if (p_ != producer.end()) ++inner[p_->second];
        }
    // per device > 0: result buffers of its nodes, replicas of the external inputs they read
    std::vector<std::uint64_t *> tmp(nk, nullptr);
    struct Cleanup {
        Engine *e;
        std::vector<std::uint64_t *> &tmp;
        std::vector<int> &dev_of;
        ~Cleanup()
        {
            for (std::size_t i = 0; i < tmp.size(); ++i)
                if (tmp[i]) (void)hefx_free(e->dev_ctx[dev_of[i]], tmp[i]);
        }
    } cleanup{this, tmp, dev_of};
    (void)device_context(0);
    ++submission;
    bool copied_in = false;
    // a device that cannot hold a sub-graph's buffers (its results, the replicas of its external inputs) hands its whole
    // share back to the home device, where every input already lives: the submission runs, only less spread out
    auto give_back = [&](int d) {
        for (int i : share[d]) {
            if (tmp[i]) (void)hefx_free(dev_ctx[d], tmp[i]);
            tmp[i] = nullptr;
            dev_of[i] = 0;
            share[0].push_back(i);
        }
        share[d].clear();
    };
    for (int d = 1; d < ndev; ++d) {
        if (share[d].empty()) continue;
        hefx_context *cx = device_context(d);
        auto &rep = replicas[d];
        for (auto it = rep.begin(); it != rep.end();) {  // replicas whose home buffer is gone
            if (it->second.owner.expired()) {
                (void)hefx_free(cx, it->second.p);
                replica_bytes[d] -= it->second.words * 8;
                it = rep.erase(it);
            } else
                ++it;
        }
        auto dev_alloc = [&](std::size_t bytes, bool is_replica) -> std::uint64_t * {
            if (is_replica) (void)evict_replicas(d, bytes, false);  // stay inside SEAL_SHIM_REPLICA_MB per device
            void *p_ = nullptr;
            if (hefx_malloc(cx, bytes, &p_) == HEFX_OK) return static_cast<std::uint64_t *>(p_);
            (void)hefx_stream_sync(cx, nullptr);
            if (evict_replicas(d, 0, true) && hefx_malloc(cx, bytes, &p_) == HEFX_OK) return static_cast<std::uint64_t *>(p_);
            return nullptr;
        };
        bool fits = true;
        for (int i : share[d]) {
            if (!fz.unwritten[i] && !(tmp[i] = dev_alloc(K[i].dst->words * 8, false))) {
                fits = false;
                break;
            }
            for (const BufPtr *inp : {&K[i].a, &K[i].b}) {
                if (!*inp || producer.count(inp->get())) continue;
                const std::uint64_t *home = (*inp)->get();  // an external input: it has content, hence an address
                auto hit = rep.find(home);
                if (hit != rep.end() && hit->second.owner.lock().get() == inp->get()) {
                    hit->second.used = submission;
                    continue;
                }
                if (hit != rep.end()) {
                    (void)hefx_free(cx, hit->second.p);
                    replica_bytes[d] -= hit->second.words * 8;
                    rep.erase(hit);
                }
                std::uint64_t *r_ = dev_alloc((*inp)->words * 8, true);
                if (!r_) {
                    fits = false;
                    break;
                }
                check(hefx_copy_peer(cx, r_, ctx_raw, home, (*inp)->words * 8, nullptr));  // on the home stream
                rep[home] = Replica{*inp, r_, (*inp)->words, submission};
                replica_bytes[d] += (*inp)->words * 8;
                copied_in = true;
            }
            if (!fits) break;
        }
        if (!fits) give_back(d);
    }
    // the replicas were copied on the home device's stream, behind whatever produced them
    if (copied_in) check(hefx_stream_sync(ctx_raw, nullptr));
    // all devices advance depth by depth; every call is asynchronous on its own device
    std::vector<std::vector<std::vector<int>>> by_depth((std::size_t)ndev);
    for (int d = 0; d < ndev; ++d) by_depth[d] = nodes_by_depth(K, share[d], max_depth);
    for (int depth = 0; depth <= max_depth; ++depth)
        for (int d = 0; d < ndev; ++d) {
            if (share[d].empty()) continue;
            hefx_context *cx = dev_ctx[d];
            auto &rep = replicas[d];
            if (d == 0)
                submit_nodes(cx, stats, K, by_depth[d], fz, depth, depth,
                             [](const Buf *b_) -> const std::uint64_t * { return const_cast<Buf *>(b_)->get(); },
                             [&](int i) -> std::uint64_t * { return K[i].dst->get(); });
            else
                submit_nodes(cx, stats, K, by_depth[d], fz, depth, depth,
                             [&](const Buf *b_) -> const std::uint64_t * {
                                 auto pr = producer.find(b_);
                                 if (pr != producer.end()) return tmp[pr->second];
                                 return rep.at(const_cast<Buf *>(b_)->get()).p;
                             },
                             [&](int i) { return tmp[i]; });
        }
    // Results that are visible outside the graph go home.  The copy is submitted on the HOME device's stream, after the
    // host has seen the producing device finish: a home buffer comes from the pooled allocator, which hands out blocks
    // whose previous owner's work may still be queued on the home stream (hefx_malloc's contract: same-stream order
    // only) -- a copy issued on the producer's stream would race with it on a real second GPU.  The home
    // stream is then waited for once, because the producers' scratch results are recycled when this function returns.
    bool copied_back = false;
    for (int d = 1; d < ndev; ++d) {
        if (share[d].empty()) continue;
        check(hefx_stream_sync(dev_ctx[d], nullptr));
        for (int i : share[d]) {
            if (fz.unwritten[i]) continue;  // inside a fused product or a chain: never written
            if (K[i].dst.use_count() - 1 - inner[i] <= 0) continue;
            check(hefx_copy_peer_to(ctx_raw, K[i].dst->p, dev_ctx[d], tmp[i], K[i].dst->words * 8, nullptr));
            copied_back = true;
        }
    }
    if (copied_back) check(hefx_stream_sync(ctx_raw, nullptr));
}

inline BufPtr upload(const std::shared_ptr<Engine> &e, const std::vector<std::uint64_t> &h)
{
    BufPtr b = new_buf(e, h.size());
    check(hefx_upload(e->ready({}), b->p, h.data(), h.size() * 8, nullptr));
    return b;
}
inline std::vector<std::uint64_t> download(const BufPtr &b, std::size_t words = 0)
{
    std::vector<std::uint64_t> h(words ? words : b->words);
    check(hefx_download(b->eng->ready({b.get()}), h.data(), b->p, h.size() * 8, nullptr));
    return h;
}

inline std::mt19937_64 &rng()
{
    static std::mt19937_64 g = [] {
        if (const char *s = std::getenv("SEAL_SHIM_SEED")) return std::mt19937_64(std::strtoull(s, nullptr, 0));
        std::random_device rd;
        return std::mt19937_64(((std::uint64_t)rd() << 32) ^ rd());
    }();
    return g;
}

// 32-byte key + stream counter of the engine's counter-mode sampler (hefx_sample_* / hefx_encrypt): one per
// KeyGenerator / Encryptor.  Key bytes come from std::random_device (the OS entropy source), or -- tests only --
// from the deterministic generator above when SEAL_SHIM_SEED is set.
struct SamplerState {
    std::array<std::uint8_t, 32> key{};
    mutable std::uint64_t next = 0;
    SamplerState()
    {
        if (std::getenv("SEAL_SHIM_SEED")) {
            for (auto &b : key) b = (std::uint8_t)rng()();
        } else {
            std::random_device rd;
            for (std::size_t i = 0; i < key.size(); i += 4) {
                const std::uint32_t v = rd();
                for (int b = 0; b < 4; ++b) key[i + b] = (std::uint8_t)(v >> (8 * b));
            }
        }
    }
    // a copy must never replay its source's streams (u / e reuse across messages): copies re-key themselves
    SamplerState(const SamplerState &) : SamplerState() {}
    SamplerState &operator=(const SamplerState &) { return *this; }
    std::uint64_t stream() const { return ++next; }
};

inline std::uint32_t galois_elt_from_step(int step, std::size_t n)
{
    const std::uint32_t m = (std::uint32_t)(2 * n);
    if (step == 0) return m - 1;
    std::size_t pos;
    if (step < 0) {
        if ((std::size_t)(-step) >= n / 2) throw std::invalid_argument("step count too large");
        pos = n / 2 - (std::size_t)(-step);
    } else {
        if ((std::size_t)step >= n / 2) throw std::invalid_argument("step count too large");
        pos = (std::size_t)step;
    }
    // 3^pos mod 2N by squaring (the loop over pos multiplications was the largest single cost of RECORDING a rotation:
    // the negative terms of a NAF chain have pos ~ N/2, ~4 us each at N = 8192)
    std::uint64_t e = 1, b = 3;
    for (std::size_t x = pos; x; x >>= 1) {
        if (x & 1) e = (e * b) & (m - 1);
        b = (b * b) & (m - 1);
    }
    return (std::uint32_t)e;
}

// non-adjacent form, least significant term first; out holds at most 33 terms.  (No allocation: this runs once per
// recorded rotate_vector.)
inline int naf_terms(int value, int (&out)[34])
{
    int cnt = 0;
    const bool sign = value < 0;
    value = std::abs(value);
    for (int i = 0; value; ++i) {
        const int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) out[cnt++] = (sign ? -zi : zi) * (1 << i);
    }
    return cnt;
}
inline std::vector<int> naf(int value)
{
    int t[34];
    const int cnt = naf_terms(value, t);
    return std::vector<int>(t, t + cnt);
}

inline std::uint32_t bitrev(std::uint32_t x, int bits)
{
    std::uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r = (r << 1) | ((x >> i) & 1);
    return r;
}

}  // namespace shim

// ------------------------------------------------------------------------------------------------
// parameters / context
// ------------------------------------------------------------------------------------------------
class CoeffModulus {
public:
    static int MaxBitCount(std::size_t n)
    {
        switch (n) {
            case 1024: return 27;
            case 2048: return 54;
            case 4096: return 109;
            case 8192: return 218;
            case 16384: return 438;
            case 32768: return 881;
            default: return 0;
        }
    }
    // SEAL CoeffModulus::Create
    static std::vector<SmallModulus> Create(std::size_t n, std::vector<int> bit_sizes)
    {
        std::map<int, int> need;
        for (int b : bit_sizes) {
            if (b < 2 || b > 60) throw std::invalid_argument("bit_sizes is invalid");
            ++need[b];
        }
        std::map<int, std::vector<std::uint64_t>> table;
        for (auto &kv : need) {
            std::uint64_t v = ((std::uint64_t)1 << kv.first) - 2 * n + 1, lower = (std::uint64_t)1 << (kv.first - 1);
            auto &vec = table[kv.first];
            while ((int)vec.size() < kv.second && v > lower) {
                if (shim::is_prime(v)) vec.push_back(v);
                v -= 2 * n;
            }
            if ((int)vec.size() < kv.second) throw std::logic_error("failed to find enough qualifying primes");
        }
        std::vector<SmallModulus> out;
        for (int b : bit_sizes) {
            out.emplace_back(table[b].back());
            table[b].pop_back();
        }
        return out;
    }
    // SEAL's hard-coded 128-bit-security defaults (util/globals.cpp), the sizes the reference uses
    static std::vector<SmallModulus> BFVDefault(std::size_t n)
    {
        switch (n) {
            case 4096: return {0xffffee001ull, 0xffffc4001ull, 0x1ffffe0001ull};
            case 8192: return {0x7fffffd8001ull, 0x7fffffc8001ull, 0xfffffffc001ull, 0xffffff6c001ull, 0xfffffebc001ull};
            case 16384:
                return {0xfffffffd8001ull,  0xfffffffa0001ull,  0xfffffff00001ull,  0x1fffffff68001ull, 0x1fffffff50001ull,
                        0x1ffffffee8001ull, 0x1ffffffea0001ull, 0x1ffffffe88001ull, 0x1ffffffe48001ull};
            default: throw std::invalid_argument("poly_modulus_degree is not supported by BFVDefault in this shim");
        }
    }
};

class PlainModulus {
public:
    static SmallModulus Batching(std::size_t n, int bit_size) { return CoeffModulus::Create(n, {bit_size})[0]; }
};

class EncryptionParameters {
public:
    EncryptionParameters(scheme_type s = scheme_type::ckks) : scheme_(s) {}
    EncryptionParameters(std::uint8_t s) : scheme_((scheme_type)s) {}
    void set_poly_modulus_degree(std::size_t n) { n_ = n; }
    void set_coeff_modulus(const std::vector<SmallModulus> &q) { q_ = q; }
    void set_plain_modulus(const SmallModulus &t) { t_ = t; }
    void set_plain_modulus(std::uint64_t t) { t_ = SmallModulus(t); }
    std::size_t poly_modulus_degree() const { return n_; }
    const std::vector<SmallModulus> &coeff_modulus() const { return q_; }
    const SmallModulus &plain_modulus() const { return t_; }
    scheme_type scheme() const { return scheme_; }
    // SEAL's parms_id (EncryptionParameters::compute_parms_id): SHA3-256 over the uint64 words scheme,
    // poly_modulus_degree, the coeff_modulus values and the plain_modulus value
    parms_id_type parms_id() const
    {
        std::vector<std::uint64_t> w{(std::uint64_t)scheme_, (std::uint64_t)n_};
        for (auto &q : q_) w.push_back(q.value());
        w.push_back(t_.value());
        return shim::sha3_words(w);
    }
    bool operator==(const EncryptionParameters &o) const { return scheme_ == o.scheme_ && n_ == o.n_ && q_ == o.q_ && t_ == o.t_; }
    bool operator!=(const EncryptionParameters &o) const { return !(*this == o); }
    // SEAL 3.4.5: static Save / Load -- scheme (one byte), poly_modulus_degree, coeff_mod_count, the coeff_modulus
    // values, the plain_modulus value (saved for both schemes).  Members save / load: the >= 3.5 spelling, same bytes here.
    static void Save(const EncryptionParameters &parms, std::ostream &stream)
    {
        shim::StreamGuard g(stream);
        shim::put_u8(stream, (std::uint8_t)parms.scheme_);
        shim::put_u64(stream, (std::uint64_t)parms.n_);
        shim::put_u64(stream, (std::uint64_t)parms.q_.size());
        for (auto &q : parms.q_) q.save(stream);
        parms.t_.save(stream);
    }
    static EncryptionParameters Load(std::istream &stream)
    {
        shim::StreamGuard g(stream);
        const std::uint8_t scheme = shim::get_u8(stream);
        if (scheme != (std::uint8_t)scheme_type::bfv && scheme != (std::uint8_t)scheme_type::ckks)
            throw std::invalid_argument("unsupported scheme");
        EncryptionParameters p(scheme);
        const std::uint64_t n = shim::get_u64(stream), cnt = shim::get_u64(stream);
        if (n > 32768 || cnt > 62) throw std::invalid_argument("coeff_modulus is invalid");  // SEAL_COEFF_MOD_COUNT_MAX
        p.n_ = (std::size_t)n;
        p.q_.resize((std::size_t)cnt);
        for (auto &q : p.q_) q.load(stream);
        p.t_.load(stream);
        return p;
    }
    void save(std::ostream &stream) const { Save(*this, stream); }
    void load(std::istream &stream) { *this = Load(stream); }

private:
    scheme_type scheme_;
    std::size_t n_ = 0;
    std::vector<SmallModulus> q_;
    SmallModulus t_;
};

struct EncryptionParameterQualifiers {
    bool parameters_set = true, using_fft = true, using_ntt = true, using_batching = false, using_fast_plain_lift = false,
         using_descending_modulus_chain = true;
    int sec_level = 128;
};

class SEALContext {
public:
    class ContextData {
    public:
        const EncryptionParameters &parms() const { return parms_; }
        const parms_id_type &parms_id() const { return id_; }
        std::size_t chain_index() const { return chain_index_; }
        int total_coeff_modulus_bit_count() const { return bits_; }
        EncryptionParameterQualifiers qualifiers() const { return quals_; }
        std::shared_ptr<const ContextData> next_context_data() const { return next_; }

    private:
        friend class SEALContext;
        EncryptionParameters parms_;
        parms_id_type id_{};
        std::size_t chain_index_ = 0;
        int bits_ = 0;
        EncryptionParameterQualifiers quals_;
        std::shared_ptr<const ContextData> next_;
    };

    // SEAL 3.6 spelling: public, copyable (helper.h:239, logistic_regression_ckks.cpp:59-60)
    SEALContext(const EncryptionParameters &parms, bool = true, int = 128) { init(parms); }
    // SEAL 3.4.5 spelling (linear_transformation2.cpp:235, matrix_multiplication.cpp:149)
    static std::shared_ptr<SEALContext> Create(const EncryptionParameters &parms, bool = true, int = 128)
    {
        return std::make_shared<SEALContext>(parms);
    }

    std::shared_ptr<const ContextData> key_context_data() const { return levels_.back(); }
    std::shared_ptr<const ContextData> first_context_data() const { return levels_[first_level()]; }
    std::shared_ptr<const ContextData> last_context_data() const { return levels_[0]; }
    std::shared_ptr<const ContextData> get_context_data(const parms_id_type &id) const
    {
        for (auto &l : levels_)
            if (l->id_ == id) return l;
        return nullptr;
    }
    const parms_id_type &key_parms_id() const { return levels_.back()->id_; }
    const parms_id_type &first_parms_id() const { return levels_[first_level()]->id_; }
    const parms_id_type &last_parms_id() const { return levels_[0]->id_; }
    bool parameters_set() const { return true; }
    bool using_keyswitching() const { return k_ > 1; }

    // ---- shim internals
    const std::shared_ptr<shim::Engine> &engine() const
    {
        if (!eng_) eng_ = shim::get_engine((std::uint32_t)n_, primes_);  // lazily: BFV drivers never reach the GPU
        return eng_;
    }
    std::size_t n() const { return n_; }
    int k() const { return k_; }
    const std::vector<std::uint64_t> &primes() const { return primes_; }
    bool is_ckks() const { return scheme_ == scheme_type::ckks; }
    std::uint64_t plain_modulus_value() const { return t_; }
    // ---- BFV (shim_bfv.h): everything derived from (Q = q_0..q_(L-1), t), built on first use.  BFV ciphertexts live at
    // the first data level only (the demos never mod-switch them).
    struct BfvTools {
        int L = 0;                                  // data primes
        std::uint64_t t = 0;
        shim::bfv::Basis data, aux;                 // Q-basis and the auxiliary basis of the tensor product
        shim::bfv::Big delta;                       // floor(Q / t)
        std::vector<std::uint64_t> delta_mod;       // delta mod q_j
        std::shared_ptr<shim::Engine> aux_engine;   // hefx context over the auxiliary primes (NTT + dyadic products)
        shim::bfv::PlainNtt pntt;
        bool batching = false;
    };
    int bfv_top_level() const { return k_ > 1 ? k_ - 1 : 1; }
    // tools of the level with `L` data primes (0: the first data level); levels share the auxiliary basis and engine
    const BfvTools &bfv(int L = 0) const
    {
        if (L <= 0) L = bfv_top_level();
        auto hit = bfv_.find(L);
        if (hit != bfv_.end()) return *hit->second;
        if (is_ckks()) throw std::logic_error("not a BFV context");
        if (t_ < 2) throw std::invalid_argument("plain_modulus is not set");
        if (L > bfv_top_level()) throw std::invalid_argument("encrypted is not valid for encryption parameters");
        auto b = std::make_shared<BfvTools>();
        b->L = L;
        b->t = t_;
        b->data.init(std::vector<std::uint64_t>(primes_.begin(), primes_.begin() + b->L));
        shim::bfv::Big rem;
        shim::bfv::divrem(b->data.M, shim::bfv::Big(t_), b->delta, rem);
        for (int j = 0; j < b->L; ++j) b->delta_mod.push_back(shim::bfv::mod_small(b->delta, primes_[j]));
        // tensor-product coefficients are sums of N products of centred residues, up to three of them per output
        // polynomial: |x| < 3 N (Q/2)^2 at the top level; one auxiliary basis of that width serves every level
        if (bfv_.empty()) {
            shim::bfv::Basis top;
            top.init(std::vector<std::uint64_t>(primes_.begin(), primes_.begin() + bfv_top_level()));
            int logn = 0;
            while (((std::size_t)1 << logn) < n_) ++logn;
            const int need = 2 * top.M.bits() + logn + 4;
            std::vector<std::uint64_t> aux;
            for (std::uint64_t v = ((std::uint64_t)1 << 60) - 2 * n_ + 1; (int)aux.size() * 59 < need; v -= 2 * n_)
                if (shim::is_prime(v) && std::find(primes_.begin(), primes_.end(), v) == primes_.end()) aux.push_back(v);
            if (aux.size() > 7) throw std::invalid_argument("coeff_modulus too large for the BFV demo path of this shim");
            bfv_aux_.init(aux);
            bfv_aux_engine_ = shim::get_engine((std::uint32_t)n_, aux);
        }
        b->aux = bfv_aux_;
        b->aux_engine = bfv_aux_engine_;
        b->batching = b->pntt.init(t_, n_);
        bfv_[L] = b;
        return *b;
    }
    // number of RNS rows of a level, from its parms_id (0 if unknown)
    int rows_of(const parms_id_type &id) const
    {
        for (std::size_t i = 0; i < levels_.size(); ++i)
            if (levels_[i]->id_ == id) return (int)i + 1;
        return 0;
    }
    const parms_id_type &id_of_rows(int rows) const { return levels_[rows - 1]->id_; }

private:
    std::size_t first_level() const { return k_ > 1 ? (std::size_t)k_ - 2 : 0; }
    void init(const EncryptionParameters &parms)
    {
        n_ = parms.poly_modulus_degree();
        scheme_ = parms.scheme();
        k_ = (int)parms.coeff_modulus().size();
        t_ = parms.plain_modulus().value();
        if (k_ < 1) throw std::invalid_argument("coeff_modulus is not set");
        if (n_ < 1024 || (n_ & (n_ - 1))) throw std::invalid_argument("poly_modulus_degree is not valid");
        for (auto &q : parms.coeff_modulus()) primes_.push_back(q.value());
        levels_.resize(k_);
        for (int rows = 1; rows <= k_; ++rows) {
            auto cd = std::make_shared<ContextData>();
            EncryptionParameters p(scheme_);
            p.set_poly_modulus_degree(n_);
            p.set_coeff_modulus(std::vector<SmallModulus>(parms.coeff_modulus().begin(), parms.coeff_modulus().begin() + rows));
            p.set_plain_modulus(parms.plain_modulus());
            cd->parms_ = p;
            cd->id_ = p.parms_id();  // SEAL's own value (SHA3-256 of the level's parameters): what save() writes
            // chain_index: key level = k-1 ... last data level = 0 (App. A.2)
            cd->chain_index_ = (std::size_t)(rows - 1);
            shim::u128 dummy = 0;
            (void)dummy;
            int bits = 0;
            {  // bit count of the product of the first `rows` primes
                long double lg = 0;
                for (int j = 0; j < rows; ++j) lg += std::log2((long double)primes_[j]);
                bits = (int)std::floor(lg) + 1;
            }
            cd->bits_ = bits;
            cd->quals_.using_batching = scheme_ == scheme_type::bfv;
            levels_[rows - 1] = cd;
        }
        for (int rows = 2; rows <= k_; ++rows) levels_[rows - 1]->next_ = levels_[rows - 2];
        // every parameter set a context was built for is remembered: the context-free unsafe_load(stream) of SEAL 3.4.x
        // finds the engine of a stream's parms_id here (known())
        std::lock_guard<std::mutex> lk(known_mutex());
        bool seen = false;
        for (auto &c : known_list()) seen = seen || c->levels_.back()->id_ == levels_.back()->id_;
        if (!seen) known_list().push_back(std::make_shared<SEALContext>(*this));
    }
    static std::mutex &known_mutex()
    {
        static std::mutex m;
        return m;
    }
    static std::vector<std::shared_ptr<SEALContext>> &known_list()
    {
        static auto *v = new std::vector<std::shared_ptr<SEALContext>>();
        return *v;
    }

public:
    // a context of this process one of whose levels carries `id` (nullptr: none was ever created)
    static std::shared_ptr<SEALContext> known(const parms_id_type &id)
    {
        std::lock_guard<std::mutex> lk(known_mutex());
        for (auto &c : known_list())
            if (c->rows_of(id)) return c;
        return nullptr;
    }

private:

    std::size_t n_ = 0;
    int k_ = 0;
    scheme_type scheme_ = scheme_type::ckks;
    std::vector<std::uint64_t> primes_;
    std::vector<std::shared_ptr<ContextData>> levels_;  // index = rows-1
    mutable std::shared_ptr<shim::Engine> eng_;
    std::uint64_t t_ = 0;
    mutable std::map<int, std::shared_ptr<BfvTools>> bfv_;
    mutable shim::bfv::Basis bfv_aux_;
    mutable std::shared_ptr<shim::Engine> bfv_aux_engine_;
};

namespace shim {
// the two context spellings the reference passes to constructors
inline std::shared_ptr<SEALContext> as_ptr(const std::shared_ptr<SEALContext> &c)
{
    if (!c) throw std::invalid_argument("invalid context");
    return c;
}
inline std::shared_ptr<SEALContext> as_ptr(const SEALContext &c) { return std::make_shared<SEALContext>(c); }

// Delta * m modulo every data prime, [L][N] (coefficient form): what BFV encryption and add_plain add to c0
inline std::vector<std::uint64_t> bfv_scaled_plain(const SEALContext &ctx, const std::vector<std::uint64_t> &coeffs,
                                                    int L = 0)
{
    const auto &B = ctx.bfv(L);
    const std::size_t n = ctx.n();
    std::vector<std::uint64_t> out((std::size_t)B.L * n, 0);
    for (int j = 0; j < B.L; ++j) {
        const std::uint64_t q = ctx.primes()[j], d = B.delta_mod[j];
        for (std::size_t i = 0; i < coeffs.size() && i < n; ++i) out[(std::size_t)j * n + i] = bfv::mulmod64(d, coeffs[i] % q, q);
    }
    return out;
}
// a BFV plaintext as a ring element modulo every data prime, [L][N] coefficient form: coefficients above t/2 stand for
// negative numbers (SEAL's multiply_plain lift), so products stay small
inline std::vector<std::uint64_t> bfv_lifted_plain(const SEALContext &ctx, const std::vector<std::uint64_t> &coeffs, int L)
{
    const std::size_t n = ctx.n();
    const std::uint64_t t = ctx.plain_modulus_value();
    std::vector<std::uint64_t> out((std::size_t)L * n, 0);
    for (int j = 0; j < L; ++j) {
        const std::uint64_t q = ctx.primes()[j];
        for (std::size_t i = 0; i < coeffs.size() && i < n; ++i) {
            const std::uint64_t m = coeffs[i];
            out[(std::size_t)j * n + i] = m > t / 2 ? q - ((t - m) % q) : m % q;
        }
    }
    return out;
}
}  // namespace shim

// ------------------------------------------------------------------------------------------------
// payload carriers
// ------------------------------------------------------------------------------------------------
class Plaintext {
public:
    Plaintext() = default;
    // SEAL's hexadecimal polynomial notation, e.g. "1x^3 + 2x^2 + 3x^1 + 4", "6", "0" (1_bfv.cpp:45, 3_levels.cpp:88)
    explicit Plaintext(const std::string &hex_poly)
    {
        std::size_t pos = 0;
        const std::string &h = hex_poly;
        while (pos < h.size()) {
            while (pos < h.size() && (h[pos] == ' ' || h[pos] == '+')) ++pos;
            if (pos >= h.size()) break;
            std::size_t end = pos;
            while (end < h.size() && std::isxdigit((unsigned char)h[end])) ++end;
            if (end == pos) throw std::invalid_argument("unable to parse hex_poly");
            const std::uint64_t coeff = std::stoull(h.substr(pos, end - pos), nullptr, 16);
            std::size_t exp = 0;
            pos = end;
            if (pos < h.size() && h[pos] == 'x') {
                if (pos + 1 >= h.size() || h[pos + 1] != '^') throw std::invalid_argument("unable to parse hex_poly");
                pos += 2;
                end = pos;
                while (end < h.size() && std::isdigit((unsigned char)h[end])) ++end;
                if (end == pos) throw std::invalid_argument("unable to parse hex_poly");
                exp = (std::size_t)std::stoull(h.substr(pos, end - pos));
                pos = end;
            }
            if (bfv.size() <= exp) bfv.resize(exp + 1, 0);
            bfv[exp] = coeff;
        }
        if (bfv.empty()) bfv.assign(1, 0);
    }
    double &scale() { return scale_; }
    const double &scale() const { return scale_; }
    parms_id_type &parms_id() { return id_; }
    const parms_id_type &parms_id() const { return id_; }
    bool is_ntt_form() const { return bfv.empty(); }
    bool is_zero() const { return zero_; }
    // SEAL: number of uint64 words of the plaintext (CKKS: rows * N in NTT form; BFV: coefficients modulo t)
    std::size_t coeff_count() const { return bfv.empty() && buf ? (view_words_ ? view_words_ : buf->words) : bfv.size(); }
    // read access to the words (SEAL's Plaintext::data()); CKKS payloads live on the device: a host mirror is fetched
    const std::uint64_t *data() const
    {
        if (!bfv.empty()) return bfv.data();
        if (!buf) return nullptr;
        if (!mirror_ || mirror_of_ != buf->p) {
            mirror_ = std::make_shared<std::vector<std::uint64_t>>(shim::download(buf, view_words_));
            mirror_of_ = buf->p;
        }
        return mirror_->data();
    }
    std::string to_string() const
    {
        if (bfv.empty() && buf) throw std::invalid_argument("cannot convert NTT transformed plaintext to string");
        std::string out;
        char tmp[32];
        for (std::size_t i = bfv.size(); i-- > 0;) {
            if (!bfv[i]) continue;
            if (!out.empty()) out += " + ";
            std::snprintf(tmp, sizeof tmp, "%llX", (unsigned long long)bfv[i]);
            out += tmp;
            if (i) out += "x^" + std::to_string(i);
        }
        return out.empty() ? "0" : out;
    }
    // ---- SEAL 3.4.5 Plaintext::save: parms_id (4 x uint64), scale (double), then the IntArray -- coefficient count and the
    // words (CKKS: rows * N words in NTT form; BFV: coefficients modulo t under parms_id_zero).  Format notes: shim_io.h.
    void save(std::ostream &stream) const
    {
        shim::StreamGuard g(stream);
        shim::put_id(stream, id_);
        shim::put_f64(stream, scale_);
        const bool on_device = bfv.empty() && buf;
        shim::put_words(stream, on_device ? data() : bfv.data(), on_device ? coeff_count() : bfv.size());
    }
    // unsafe_load: the stream's words as they are (structure checked only as far as the payload needs a home)
    template <class Ctx>
    void unsafe_load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        shim::StreamGuard g(stream);
        const parms_id_type id = shim::get_id(stream);
        const double scale = shim::get_f64(stream);
        Plaintext p;
        if (id == parms_id_zero) {  // BFV form
            p.bfv = shim::get_words(stream, (std::uint64_t)ctx->n());
            if (p.bfv.empty()) p.bfv.assign(1, 0);
        } else {
            const int rows = ctx->rows_of(id);
            if (!rows) throw std::invalid_argument("plaintext data is invalid: parms_id is not of this context");
            auto w = shim::get_words(stream, (std::uint64_t)rows * ctx->n());
            if (w.size() != (std::size_t)rows * ctx->n()) throw std::invalid_argument("plaintext data is invalid");
            p.zero_ = std::all_of(w.begin(), w.end(), [](std::uint64_t x) { return x == 0; });
            p.buf = shim::upload(ctx->engine(), w);
            p.rows = rows;
        }
        p.id_ = id;
        p.scale_ = scale;
        *this = std::move(p);
    }
    // load: unsafe_load + SEAL's is_valid_for (every word a residue of its row's prime / of the plain modulus)
    template <class Ctx>
    void load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        Plaintext p;
        p.unsafe_load(ctx, stream);
        const std::uint64_t *w = p.data();
        if (p.id_ == parms_id_zero) {
            const std::uint64_t t = ctx->plain_modulus_value();
            for (std::size_t i = 0; t && i < p.bfv.size(); ++i)
                if (p.bfv[i] >= t) throw std::invalid_argument("plaintext data is invalid");
        } else {
            for (int j = 0; j < p.rows; ++j)
                for (std::size_t i = 0; i < ctx->n(); ++i)
                    if (w[(std::size_t)j * ctx->n() + i] >= ctx->primes()[j]) throw std::invalid_argument("plaintext data is invalid");
        }
        *this = std::move(p);
    }
    // SEAL 3.4.x spelling without a context: the parameter set is found among the contexts this process has created
    inline void unsafe_load(std::istream &stream);
    // shim internals
    shim::BufPtr buf;                 // CKKS: [rows][N] NTT form on the device
    std::vector<std::uint64_t> bfv;   // BFV: N coefficients modulo the plain modulus, on the host
    int rows = 0;
    bool zero_ = false;
    std::size_t view_words_ = 0;      // non-zero: the plaintext is the first view_words_ words of buf (after a mod switch)
    mutable std::shared_ptr<std::vector<std::uint64_t>> mirror_;
    mutable const std::uint64_t *mirror_of_ = nullptr;

private:
    double scale_ = 1.0;
    parms_id_type id_ = parms_id_zero;
};

class Ciphertext {
public:
    Ciphertext() = default;
    double &scale() { return scale_; }
    const double &scale() const { return scale_; }
    parms_id_type &parms_id() { return id_; }
    const parms_id_type &parms_id() const { return id_; }
    std::size_t size() const { return size_; }
    std::size_t coeff_mod_count() const { return (std::size_t)rows; }
    bool is_ntt_form() const { return ntt_form_; }
    std::size_t poly_modulus_degree() const { return buf && size_ && rows ? buf->words / (size_ * (std::size_t)rows) : 0; }
    // read access to the words in SEAL's layout (SEAL's Ciphertext::data()): a host mirror of the device payload,
    // fetched on first use (this is how tools/gen_seal_vectors.cpp reads results; the evaluator never needs it)
    const std::uint64_t *data() const
    {
        if (!buf) return nullptr;
        if (!mirror_ || mirror_of_ != buf->p) {
            mirror_ = std::make_shared<std::vector<std::uint64_t>>(shim::download(buf));
            mirror_of_ = buf->p;
        }
        return mirror_->data();
    }
    const std::uint64_t *data(std::size_t poly) const { return data() + poly * (std::size_t)rows * poly_modulus_degree(); }
    // ---- SEAL 3.4.5 Ciphertext::save: parms_id, is_ntt_form (one byte), size, poly_modulus_degree, coeff_mod_count (uint64
    // each), scale (double), then the IntArray (count + words, [size][coeff_mod_count][N]).  Format notes: shim_io.h.
    void save(std::ostream &stream) const
    {
        shim::StreamGuard g(stream);
        shim::put_id(stream, id_);
        shim::put_u8(stream, ntt_form_ ? 1 : 0);
        shim::put_u64(stream, (std::uint64_t)size_);
        shim::put_u64(stream, (std::uint64_t)poly_modulus_degree());
        shim::put_u64(stream, (std::uint64_t)rows);
        shim::put_f64(stream, scale_);
        shim::put_words(stream, buf ? data() : nullptr, buf ? buf->words : 0);
    }
    template <class Ctx>
    void unsafe_load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        shim::StreamGuard g(stream);
        Ciphertext c;
        c.id_ = shim::get_id(stream);
        c.ntt_form_ = shim::get_u8(stream) != 0;
        const std::uint64_t size = shim::get_u64(stream), n = shim::get_u64(stream), rows = shim::get_u64(stream);
        c.scale_ = shim::get_f64(stream);
        if (size > 16 || rows > (std::uint64_t)ctx->k() || (size && (n != ctx->n() || !rows)))
            throw std::invalid_argument("ciphertext data is invalid");
        auto w = shim::get_words(stream, size * rows * n);
        if (w.size() != (std::size_t)(size * rows * n)) throw std::invalid_argument("ciphertext data is invalid");
        c.size_ = (std::size_t)size;
        c.rows = (int)rows;
        if (!w.empty()) c.buf = shim::upload(ctx->engine(), w);
        *this = std::move(c);
    }
    // load: unsafe_load + SEAL's is_valid_for (a level of this context, the size that level's payload has, the form the
    // scheme keeps its ciphertexts in, every word a residue of its row's prime)
    template <class Ctx>
    void load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        Ciphertext c;
        c.unsafe_load(ctx, stream);
        if (c.size_) {
            // (rows_of() is 0 for a parms_id this context does not know: rejected as is_valid_for does, even when the stream
            // also says "zero rows")
            const int known_rows = ctx->rows_of(c.id_);
            if (!known_rows || known_rows != c.rows || c.size_ < 2 || c.ntt_form_ != ctx->is_ckks())
                throw std::invalid_argument("ciphertext data is invalid");
            const std::uint64_t *w = c.data();
            for (std::size_t p = 0; p < c.size_; ++p)
                for (int j = 0; j < c.rows; ++j)
                    for (std::size_t i = 0; i < ctx->n(); ++i)
                        if (w[(p * c.rows + j) * ctx->n() + i] >= ctx->primes()[j]) throw std::invalid_argument("ciphertext data is invalid");
        }
        *this = std::move(c);
    }
    inline void unsafe_load(std::istream &stream);  // SEAL 3.4.x spelling, see Plaintext::unsafe_load
    // shim internals
    shim::BufPtr buf;
    int rows = 0;
    std::size_t size_ = 0;
    mutable std::shared_ptr<std::vector<std::uint64_t>> mirror_;
    mutable const std::uint64_t *mirror_of_ = nullptr;
    bool ntt_form_ = true;  // CKKS ciphertexts are always NTT form, BFV ones coefficient form (as in SEAL)
    void set(shim::BufPtr b, std::size_t size, int r, const parms_id_type &id, double scale)
    {
        buf = std::move(b);
        size_ = size;
        rows = r;
        id_ = id;
        scale_ = scale;
    }

private:
    double scale_ = 1.0;
    parms_id_type id_ = parms_id_zero;
};

class SecretKey {
public:
    std::vector<std::uint64_t> host;  // [k][N], NTT form
    shim::BufPtr buf;
    parms_id_type &parms_id() { return id_; }
    const parms_id_type &parms_id() const { return id_; }
    // SEAL: a secret key is a key-level Plaintext in NTT form (scale 1)
    void save(std::ostream &stream) const
    {
        shim::StreamGuard g(stream);
        shim::put_id(stream, id_);
        shim::put_f64(stream, 1.0);
        shim::put_words(stream, host.data(), host.size());
    }
    template <class Ctx>
    void unsafe_load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        Plaintext p;
        p.unsafe_load(ctx, stream);
        adopt(*ctx, p);
    }
    template <class Ctx>
    void load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        Plaintext p;
        p.load(ctx, stream);
        adopt(*ctx, p);
    }

private:
    void adopt(const SEALContext &ctx, const Plaintext &p)
    {
        if (p.parms_id() != ctx.key_parms_id() || !p.buf) throw std::invalid_argument("SecretKey data is invalid");
        buf = p.buf;
        host = shim::download(buf);
        id_ = p.parms_id();
    }
    parms_id_type id_ = parms_id_zero;
};
class PublicKey {
public:
    shim::BufPtr buf;  // [2][k][N]
    // SEAL: a public key (and every component of a key-switching key) is a size-2 key-level ciphertext
    const Ciphertext &data() const
    {
        if (!view_.buf || view_.buf != buf) view_.set(buf, 2, rows_, id_, 1.0);
        return view_;
    }
    int rows_ = 0;
    parms_id_type &parms_id() { return id_; }
    const parms_id_type &parms_id() const { return id_; }
    void save(std::ostream &stream) const { data().save(stream); }
    template <class Ctx>
    void unsafe_load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        Ciphertext c;
        c.unsafe_load(ctx, stream);
        adopt(*ctx, c);
    }
    template <class Ctx>
    void load(const Ctx &context, std::istream &stream)
    {
        auto ctx = shim::as_ptr(context);
        Ciphertext c;
        c.load(ctx, stream);
        adopt(*ctx, c);
    }

private:
    void adopt(const SEALContext &ctx, const Ciphertext &c)
    {
        if (c.size() != 2 || c.rows != ctx.k() || c.parms_id() != ctx.key_parms_id() || !c.buf)
            throw std::invalid_argument("PublicKey data is invalid");
        buf = c.buf;
        rows_ = c.rows;
        id_ = c.parms_id();
    }
    parms_id_type id_ = parms_id_zero;
    mutable Ciphertext view_;
};
class KSwitchKeys {
public:
    virtual ~KSwitchKeys() = default;
    bool has_key(std::uint32_t elt) const { return keys.count(elt) != 0; }
    std::map<std::uint32_t, shim::BufPtr> keys;  // Galois element -> [k-1][2][k][N]; relin key under element 0
    std::size_t size() const { return keys.size(); }
    parms_id_type &parms_id() { return id_; }
    const parms_id_type &parms_id() const { return id_; }
    // ---- SEAL 3.4.5 KSwitchKeys::save: parms_id, the outer dimension, then per index its component count followed by that
    // many PublicKeys (key-level size-2 ciphertexts: component i = digit i = [2][k][N]).  RelinKeys: index = key_power - 2
    // (one index); GaloisKeys: index = (galois_elt - 1) / 2 over N indices, absent elements with zero components.
    void save(std::ostream &stream) const
    {
        shim::StreamGuard g(stream);
        shim::put_id(stream, id_);
        const std::uint64_t dim1 = keys.empty() ? 0 : (galois_indexing() ? (std::uint64_t)keys.begin()->second->eng->n : 1);
        shim::put_u64(stream, dim1);
        for (std::uint64_t index = 0; index < dim1; ++index) {
            const std::uint32_t elt = galois_indexing() ? (std::uint32_t)(2 * index + 1) : 0u;
            if (!has_key(elt)) {
                shim::put_u64(stream, 0);
                continue;
            }
            const std::vector<PublicKey> &comps = components(elt);
            shim::put_u64(stream, (std::uint64_t)comps.size());
            for (const PublicKey &c : comps) {
                Ciphertext view;
                view.set(c.buf, 2, c.rows_, id_, 1.0);
                view.save(stream);
            }
        }
    }
    template <class Ctx>
    void unsafe_load(const Ctx &context, std::istream &stream)
    {
        load_impl(*shim::as_ptr(context), stream, false);
    }
    template <class Ctx>
    void load(const Ctx &context, std::istream &stream)
    {
        load_impl(*shim::as_ptr(context), stream, true);
    }

protected:
    virtual bool galois_indexing() const { return true; }  // how an index of SEAL's outer vector maps to an element
    void load_impl(const SEALContext &ctx, std::istream &stream, bool validate)
    {
        shim::StreamGuard g(stream);
        const parms_id_type id = shim::get_id(stream);
        const std::uint64_t dim1 = shim::get_u64(stream);
        const std::size_t n = ctx.n(), k = (std::size_t)ctx.k(), slice = 2 * k * n;
        if (dim1 > (galois_indexing() ? (std::uint64_t)n : 1) || (validate && id != ctx.key_parms_id()))
            throw std::invalid_argument("KSwitchKeys data is invalid");
        std::map<std::uint32_t, shim::BufPtr> loaded;
        auto ctxp = std::make_shared<SEALContext>(ctx);
        for (std::uint64_t index = 0; index < dim1; ++index) {
            const std::uint64_t dim2 = shim::get_u64(stream);
            if (!dim2) continue;
            if (dim2 != k - 1) throw std::invalid_argument("KSwitchKeys data is invalid");
            std::vector<std::uint64_t> words(dim2 * slice);
            for (std::uint64_t i = 0; i < dim2; ++i) {
                Ciphertext c;
                if (validate)
                    c.load(ctxp, stream);
                else
                    c.unsafe_load(ctxp, stream);
                if (c.size() != 2 || (std::size_t)c.rows != k || !c.buf) throw std::invalid_argument("KSwitchKeys data is invalid");
                std::memcpy(words.data() + i * slice, c.data(), slice * 8);
            }
            loaded[galois_indexing() ? (std::uint32_t)(2 * index + 1) : 0u] = shim::upload(ctx.engine(), words);
        }
        keys = std::move(loaded);
        views_.clear();
        id_ = id;
    }
    parms_id_type id_ = parms_id_zero;
    // SEAL's view of one key: vector<PublicKey>, component i = digit i = [2][k][N] (a copy of that slice)
    const std::vector<PublicKey> &components(std::uint32_t elt) const
    {
        auto hit = views_.find(elt);
        if (hit != views_.end()) return hit->second;
        const shim::BufPtr &key = keys.at(elt);
        // [k-1][2][k][N] words: solve words = (k-1) * 2 * k * N for k with N from the engine
        const std::size_t n = key->eng->n;
        std::size_t k = 2;
        while ((k - 1) * 2 * k * n < key->words) ++k;
        std::vector<PublicKey> comps(k - 1);
        const std::size_t slice = 2 * k * n;
        for (std::size_t i = 0; i + 1 < k; ++i) {
            comps[i].buf = shim::new_buf(key->eng, slice);
            shim::check(hefx_copy(key->eng->ready({}), comps[i].buf->p, key->p + i * slice, slice * 8, nullptr));
            comps[i].rows_ = (int)k;
        }
        return views_[elt] = std::move(comps);
    }
    mutable std::map<std::uint32_t, std::vector<PublicKey>> views_;
};
class RelinKeys : public KSwitchKeys {
protected:
    bool galois_indexing() const override { return false; }

public:
    static std::size_t get_index(std::size_t key_power) { return key_power - 2; }
    // shim-internal lookups use has_key(0u) (the relinearisation key is stored under element 0); SEAL's has_key(key_power)
    // is has_power
    bool has_power(std::size_t key_power) const { return key_power == 2 && keys.count(0) != 0; }
    const std::vector<PublicKey> &key(std::size_t key_power) const
    {
        if (key_power != 2) throw std::invalid_argument("key_power is not valid");
        return components(0);
    }
};
class GaloisKeys : public KSwitchKeys {
public:
    static std::size_t get_index(std::uint64_t galois_elt) { return (std::size_t)((galois_elt - 1) >> 1); }
    const std::vector<PublicKey> &key(std::uint64_t galois_elt) const { return components((std::uint32_t)galois_elt); }
};

// SEAL 3.4.x's unsafe_load(stream) has no context argument; the payload still needs a device to live on, so the stream's
// parms_id is looked up among the contexts this process has created (SEALContext::known).  The 16-byte look-ahead is
// undone with seekg: these members need a seekable stream (files, stringstreams).
inline void Plaintext::unsafe_load(std::istream &stream)
{
    const auto pos = stream.tellg();
    parms_id_type id;
    {
        shim::StreamGuard g(stream);
        id = shim::get_id(stream);
        stream.seekg(pos);
    }
    std::shared_ptr<SEALContext> ctx = id == parms_id_zero ? nullptr : SEALContext::known(id);
    if (id == parms_id_zero) {  // BFV form carries no parameters: any context's ring bounds it
        Plaintext p;
        shim::StreamGuard g(stream);
        (void)shim::get_id(stream);
        p.scale() = shim::get_f64(stream);
        p.bfv = shim::get_words(stream, 32768);
        if (p.bfv.empty()) p.bfv.assign(1, 0);
        *this = std::move(p);
        return;
    }
    if (!ctx) throw std::invalid_argument("no SEALContext of this process has the parameters of the stream (parms_id)");
    unsafe_load(ctx, stream);
}
inline void Ciphertext::unsafe_load(std::istream &stream)
{
    const auto pos = stream.tellg();
    parms_id_type id;
    {
        shim::StreamGuard g(stream);
        id = shim::get_id(stream);
        stream.seekg(pos);
    }
    auto ctx = SEALContext::known(id);
    if (!ctx) throw std::invalid_argument("no SEALContext of this process has the parameters of the stream (parms_id)");
    unsafe_load(ctx, stream);
}

// ------------------------------------------------------------------------------------------------
// KeyGenerator (App. A.11): sampling (hefx_sample_*) and arithmetic on the GPU, key assembly on the host
// ------------------------------------------------------------------------------------------------
class KeyGenerator {
public:
    template <class Ctx>
    explicit KeyGenerator(const Ctx &context) : ctx_(shim::as_ptr(context))
    {
        // the keys of BFV and CKKS are the same objects (App. A.11): one code path
        const int k = ctx_->k();
        auto &e = ctx_->engine();
        sk_.buf = shim::new_buf(e, (std::size_t)k * ctx_->n());
        shim::check(hefx_sample_ternary(e->ready({}), rnd_.key.data(), rnd_.stream(), 1, k, 0, sk_.buf->p, nullptr));
        shim::check(hefx_ntt_forward(e->ready({}), sk_.buf->p, 1, k, 0, nullptr));
        sk_.host = shim::download(sk_.buf);
        sk_.parms_id() = ctx_->key_parms_id();
    }

    const SecretKey &secret_key() const { return sk_; }
    PublicKey public_key()
    {
        if (pk_.buf) return pk_;  // generated once, like SEAL's KeyGenerator
        PublicKey &pk = pk_;
        auto z = encrypt_zero(1, ctx_->k());
        const std::size_t w = (std::size_t)ctx_->k() * ctx_->n();
        std::vector<std::uint64_t> h(2 * w);
        auto c0 = shim::download(z.first);
        std::copy(c0.begin(), c0.end(), h.begin());
        std::copy(z.second.begin(), z.second.end(), h.begin() + w);
        pk.buf = shim::upload(ctx_->engine(), h);
        pk.rows_ = ctx_->k();
        pk.parms_id() = ctx_->key_parms_id();
        return pk;
    }
    void create_public_key(PublicKey &pk) { pk = public_key(); }

    RelinKeys relin_keys(std::size_t = 1)
    {
        RelinKeys rk;
        const int k = ctx_->k();
        auto &e = ctx_->engine();
        auto s2 = shim::new_buf(e, (std::size_t)k * ctx_->n());
        shim::check(hefx_multiply_plain(e->ready({}), k, 1, 1, sk_.buf->p, sk_.buf->p, s2->p, nullptr));
        rk.keys[0] = kswitch_key(s2);
        rk.parms_id() = ctx_->key_parms_id();
        return rk;
    }
    void create_relin_keys(RelinKeys &rk) { rk = relin_keys(); }

    // default: 3^(+-2^i) and 2N-1 (power-of-two steps only, App. A.7)
    GaloisKeys galois_keys()
    {
        const std::size_t n = ctx_->n();
        int logn = 0;
        while (((std::size_t)1 << logn) < n) ++logn;
        std::vector<std::uint32_t> elts{(std::uint32_t)(2 * n - 1)};
        for (int i = 0; i < logn - 1; ++i) {
            elts.push_back(shim::galois_elt_from_step(1 << i, n));
            elts.push_back(shim::galois_elt_from_step(-(1 << i), n));
        }
        return galois_keys_for(elts);
    }
    GaloisKeys galois_keys(const std::vector<int> &steps)
    {
        std::vector<std::uint32_t> elts;
        for (int s : steps) elts.push_back(shim::galois_elt_from_step(s, ctx_->n()));
        return galois_keys_for(elts);
    }
    void create_galois_keys(GaloisKeys &gk) { gk = galois_keys(); }
    void create_galois_keys(const std::vector<int> &steps, GaloisKeys &gk) { gk = galois_keys(steps); }

private:
    GaloisKeys galois_keys_for(const std::vector<std::uint32_t> &elts)
    {
        GaloisKeys gk;
        auto &e = ctx_->engine();
        const int k = ctx_->k();
        for (std::uint32_t g : elts) {
            if (gk.has_key(g)) continue;
            auto sp = shim::new_buf(e, (std::size_t)k * ctx_->n());  // s(X^g), NTT domain, on the device
            shim::check(hefx_galois_permute(e->ready({}), g, sk_.buf->p, k, sp->p, nullptr));
            gk.keys[g] = kswitch_key(sp);
        }
        gk.parms_id() = ctx_->key_parms_id();
        return gk;
    }

    // npoly fresh symmetric encryptions of zero over the first `rows` primes: (device c0, host c1) -- public key
    std::pair<shim::BufPtr, std::vector<std::uint64_t>> encrypt_zero(int npoly, int rows)
    {
        auto &e = ctx_->engine();
        const std::size_t n = ctx_->n();
        const std::size_t words = (std::size_t)npoly * rows * n;
        auto da = shim::new_buf(e, words), de = shim::new_buf(e, words);
        shim::check(hefx_sample_uniform(e->ready({}), rnd_.key.data(), rnd_.stream(), npoly, rows, 0, da->p, nullptr));
        shim::check(hefx_sample_noise(e->ready({}), rnd_.key.data(), rnd_.stream(), npoly, rows, 0, de->p, nullptr));
        const std::vector<std::uint64_t> a = shim::download(da, words);
        shim::check(hefx_ntt_forward(e->ready({}), de->p, npoly, rows, 0, nullptr));
        auto t = shim::new_buf(e, a.size());
        shim::check(hefx_multiply_plain(e->ready({}), rows, npoly, 1, da->p, sk_.buf->p, t->p, nullptr));
        shim::check(hefx_add(e->ready({}), rows, npoly, 1, t->p, de->p, t->p, nullptr));
        shim::check(hefx_negate(e->ready({}), rows, npoly, 1, t->p, t->p, nullptr));
        return {t, a};
    }

    // key-switching key for new_sk (device, [k][N] NTT form) under sk: sampling, NTT and assembly on the GPU
    shim::BufPtr kswitch_key(const shim::BufPtr &new_sk)
    {
        auto &e = ctx_->engine();
        const int k = ctx_->k();
        if (k < 2) throw std::logic_error("keyswitching is not supported by the context");
        auto key = shim::new_buf(e, (std::size_t)(k - 1) * 2 * k * ctx_->n());
        shim::check(hefx_keygen_kswitch(e->ready({}), sk_.buf->p, new_sk->p, rnd_.key.data(), 0x40000000ull + rnd_.stream(),
                                        key->p, nullptr));
        return key;
    }

    std::shared_ptr<SEALContext> ctx_;
    SecretKey sk_;
    PublicKey pk_;
    shim::SamplerState rnd_;
};

// ------------------------------------------------------------------------------------------------
// Encryptor / Decryptor
// ------------------------------------------------------------------------------------------------
class Encryptor {
public:
    template <class Ctx>
    Encryptor(const Ctx &context, const PublicKey &pk) : ctx_(shim::as_ptr(context)), pk_(pk) {}

    // (pk0*u + e0 + m, pk1*u + e1) over the plaintext's level
    void encrypt(const Plaintext &plain, Ciphertext &dest) const
    {
        if (!ctx_->is_ckks()) return encrypt_bfv(plain, dest);
        if (!plain.buf) throw std::invalid_argument("plain is not valid for encryption parameters");
        auto &e = ctx_->engine();
        const int L = plain.rows;
        if (e->lazy && !shim::sync_mode()) {  // recorded: consecutive encryptions of one Encryptor run as hefx_encrypt_batch
            shim::Engine::Node x{};
            if (!key_) key_ = std::make_shared<std::array<std::uint8_t, 32>>(rnd_.key);
            x.skey = key_;
            x.stream_id = rnd_.stream();
            dest.set(e->record(shim::Engine::Node::ENCRYPT, plain.buf, pk_.buf, 0, L, 2, (std::size_t)2 * L * ctx_->n(), e, &x), 2,
                     L, plain.parms_id(), plain.scale());
            return;
        }
        auto c = shim::new_buf(e, (std::size_t)2 * L * ctx_->n());
        // sampling (u ternary, e0/e1 clipped normal), NTT and the dyadic arithmetic: one engine call
        shim::check(hefx_encrypt(e->ready({plain.buf.get()}), L, pk_.buf->p, plain.buf->p, rnd_.key.data(), rnd_.stream(), c->p, nullptr));
        dest.set(c, 2, L, plain.parms_id(), plain.scale());
    }

private:
    // BFV: (pk0*u + e0 + Delta*m, pk1*u + e1) at the first data level, coefficient form (vector_ops.cpp:155)
    void encrypt_bfv(const Plaintext &plain, Ciphertext &dest) const
    {
        const auto &B = ctx_->bfv();
        if (plain.bfv.empty() || plain.bfv.size() > ctx_->n()) throw std::invalid_argument("plain is not valid for encryption parameters");
        for (std::uint64_t m : plain.bfv)
            if (m >= B.t) throw std::invalid_argument("plain is not valid for encryption parameters");
        auto &e = ctx_->engine();
        const int L = B.L;
        const std::size_t n = ctx_->n();
        auto c = shim::new_buf(e, (std::size_t)2 * L * n);
        shim::check(hefx_encrypt(e->ready({}), L, pk_.buf->p, nullptr, rnd_.key.data(), rnd_.stream(), c->p, nullptr));
        shim::check(hefx_ntt_inverse(e->ready({}), c->p, 2, L, 0, nullptr));
        auto dm = shim::upload(e, shim::bfv_scaled_plain(*ctx_, plain.bfv));
        shim::check(hefx_add_plain(e->ready({}), L, 2, c->p, dm->p, c->p, nullptr));
        dest.set(c, 2, L, ctx_->id_of_rows(L), 1.0);
        dest.ntt_form_ = false;
    }

    std::shared_ptr<SEALContext> ctx_;
    PublicKey pk_;
    shim::SamplerState rnd_;
    mutable std::shared_ptr<std::array<std::uint8_t, 32>> key_;  // rnd_.key, shared with the recorded encryptions
};

class Decryptor {
public:
    template <class Ctx>
    Decryptor(const Ctx &context, const SecretKey &sk) : ctx_(shim::as_ptr(context)), sk_(sk) {}

    // c0 + c1 s (+ c2 s^2 ...): handles size-3 ciphertexts (matrix_multiplication.cpp:419)
    void decrypt(const Ciphertext &ct, Plaintext &dest) const
    {
        if (!ct.buf) throw std::invalid_argument("encrypted is not valid for encryption parameters");
        if (!ctx_->is_ckks()) {
            int budget = 0;
            decrypt_bfv(ct, dest, budget);
            return;
        }
        auto &e = ctx_->engine();
        const int L = ct.rows;
        auto acc = shim::new_buf(e, (std::size_t)L * ctx_->n());
        shim::check(hefx_decrypt(e->ready({ct.buf.get()}), L, (int)ct.size(), ct.buf->p, sk_.buf->p, acc->p, nullptr));
        dest.buf = acc;
        dest.rows = L;
        dest.view_words_ = 0;
        dest.parms_id() = ct.parms_id();
        dest.scale() = ct.scale();
        dest.zero_ = false;
    }
    // bits left before decryption fails: log2(Q) - log2(|| t * [ct(s)]_Q centred mod Q ||_inf) - 1 (vector_ops.cpp:157)
    int invariant_noise_budget(const Ciphertext &ct) const
    {
        if (ctx_->is_ckks()) throw std::invalid_argument("unsupported scheme");
        if (!ct.buf) throw std::invalid_argument("encrypted is not valid for encryption parameters");
        Plaintext tmp;
        int budget = 0;
        decrypt_bfv(ct, tmp, budget);
        return budget;
    }

private:
    // BFV: x = [c0 + c1 s (+ c2 s^2)]_Q on the GPU (NTT domain), then per coefficient t*x = quo*Q + rem on the host:
    // m = round(t*x/Q) mod t, and |rem centred| is the invariant noise
    void decrypt_bfv(const Ciphertext &ct, Plaintext &dest, int &budget) const
    {
        namespace bf = shim::bfv;
        const auto &B = ctx_->bfv(ct.rows);
        auto &e = ctx_->engine();
        const int L = B.L;
        const std::size_t n = ctx_->n();
        auto tmp = shim::new_buf(e, ct.buf->words), acc = shim::new_buf(e, (std::size_t)L * n);
        shim::check(hefx_copy(e->ready({ct.buf.get()}), tmp->p, ct.buf->p, ct.buf->words * 8, nullptr));
        shim::check(hefx_ntt_forward(e->ready({}), tmp->p, (int)ct.size(), L, 0, nullptr));
        shim::check(hefx_decrypt(e->ready({}), L, (int)ct.size(), tmp->p, sk_.buf->p, acc->p, nullptr));
        shim::check(hefx_ntt_inverse(e->ready({}), acc->p, 1, L, 0, nullptr));
        const std::vector<std::uint64_t> x = shim::download(acc);
        dest = Plaintext();
        dest.bfv.assign(n, 0);
        dest.parms_id() = parms_id_zero;
        bf::Big worst;
        for (std::size_t i = 0; i < n; ++i) {
            const bf::Big xi = B.data.compose(x.data(), n, i);
            bf::Big quo, rem;
            bf::divrem(bf::mul_small(xi, B.t), B.data.M, quo, rem);
            std::uint64_t m = bf::mod_small(quo, B.t);
            if (bf::cmp(rem, B.data.half) > 0) {  // round up; the noise is the distance to the next multiple of Q
                m = (m + 1) % B.t;
                rem = bf::sub(B.data.M, rem);
            }
            dest.bfv[i] = m;
            if (bf::cmp(rem, worst) > 0) worst = rem;
        }
        budget = std::max(0, B.data.M.bits() - worst.bits() - 1);
        while (dest.bfv.size() > 1 && dest.bfv.back() == 0) dest.bfv.pop_back();  // SEAL keeps the significant coefficients
    }

    std::shared_ptr<SEALContext> ctx_;
    SecretKey sk_;
};

// ------------------------------------------------------------------------------------------------
// CKKSEncoder (App. A.12): canonical embedding, slot i <-> root zeta^(3^i); encode and decode run on the GPU
// (hefx_ckks_encode / hefx_ckks_decode); the host FFT below remains for wide coefficients and SEAL_SHIM_HOST_ENCODE=1
// ------------------------------------------------------------------------------------------------
class CKKSEncoder {
public:
    template <class Ctx>
    explicit CKKSEncoder(const Ctx &context) : ctx_(shim::as_ptr(context))
    {
        if (!ctx_->is_ckks()) throw std::invalid_argument("unsupported scheme");
        const std::size_t n = ctx_->n();
        r1_.resize(n / 2);
        r2_.resize(n / 2);
        std::uint64_t pos = 1;
        for (std::size_t i = 0; i < n / 2; ++i) {
            r1_[i] = (std::size_t)((pos - 1) >> 1);
            r2_[i] = (std::size_t)((2 * n - pos - 1) >> 1);
            pos = (pos * 3) & (2 * n - 1);
        }
        zeta_.resize(n);
        const double pi = std::acos(-1.0);
        for (std::size_t i = 0; i < n; ++i) zeta_[i] = std::polar(1.0, pi * (double)i / (double)n);
        tw_.resize(n / 2);
        for (std::size_t i = 0; i < n / 2; ++i) tw_[i] = std::polar(1.0, -2.0 * pi * (double)i / (double)n);
    }
    std::size_t slot_count() const { return ctx_->n() / 2; }

    void encode(const std::vector<double> &values, parms_id_type id, double scale, Plaintext &dest) const
    {
        const std::size_t n = ctx_->n();
        if (values.size() > n / 2) throw std::invalid_argument("values has invalid size");
        const int L = rows_checked(id, scale);
        if (encode_on_device(values, L, id, scale, dest)) return;
        std::vector<std::complex<double>> A(n, 0.0);
        for (std::size_t i = 0; i < values.size(); ++i) {
            A[r1_[i]] = values[i];
            A[r2_[i]] = values[i];
        }
        fft(A, false);  // a_k = (1/N) sum_r A_r e^{-2 pi i r k / N}
        std::vector<std::uint64_t> rows((std::size_t)L * n);
        bool any = false;
        for (std::size_t i = 0; i < n; ++i) {
            const double co = std::round((A[i] * std::conj(zeta_[i])).real() / (double)n * scale);
            any = any || co != 0.0;
            put(rows, i, co, L);
        }
        auto &e = ctx_->engine();
        dest.buf = shim::upload(e, rows);
        shim::check(hefx_ntt_forward(e->ready({}), dest.buf->p, 1, L, 0, nullptr));
        finish(dest, L, id, scale, !any);
    }
    void encode(const std::vector<double> &values, double scale, Plaintext &dest) const
    {
        encode(values, ctx_->first_parms_id(), scale, dest);
    }
    // every NTT slot = round(value*scale): no FFT
    void encode(double value, parms_id_type id, double scale, Plaintext &dest) const
    {
        const std::size_t n = ctx_->n();
        const int L = rows_checked(id, scale);
        const double co = std::round(value * scale);
        std::vector<std::uint64_t> one((std::size_t)L), rows((std::size_t)L * n);
        std::vector<std::uint64_t> tmp((std::size_t)L * n, 0);
        put(tmp, 0, co, L);  // residue of the constant per row
        for (int j = 0; j < L; ++j) std::fill(rows.begin() + (std::size_t)j * n, rows.begin() + (std::size_t)(j + 1) * n, tmp[(std::size_t)j * n]);
        dest.buf = shim::upload(ctx_->engine(), rows);
        finish(dest, L, id, scale, co == 0.0);
    }
    void encode(double value, double scale, Plaintext &dest) const { encode(value, ctx_->first_parms_id(), scale, dest); }

    void decode(const Plaintext &plain, std::vector<double> &dest) const
    {
        if (!plain.buf) throw std::invalid_argument("plain is not valid for encryption parameters");
        auto &e = ctx_->engine();
        const std::size_t n = ctx_->n();
        const int L = plain.rows;
        static const bool host_only = [] {
            const char *s = std::getenv("SEAL_SHIM_HOST_ENCODE");
            return s && *s && *s != '0';
        }();
        if (!host_only && L <= 16 && n >= 1024) {  // inverse NTT, CRT, centring and the slot-root FFT on the GPU
            dest.resize(n / 2);
            shim::check(hefx_ckks_decode(e->ready({plain.buf.get()}), L, plain.buf->p, 1, plain.scale(), dest.data(), nullptr, nullptr));
            return;
        }
        auto tmp = shim::new_buf(e, (std::size_t)L * n);
        shim::check(hefx_copy(e->ready({plain.buf.get()}), tmp->p, plain.buf->p, (std::size_t)L * n * 8, nullptr));
        shim::check(hefx_ntt_inverse(e->ready({}), tmp->p, 1, L, 0, nullptr));
        const std::vector<std::uint64_t> co = shim::download(tmp);
        // CRT compose (Garner mixed radix -> little-endian limbs), centre, scale
        const auto &q = ctx_->primes();
        std::vector<std::vector<std::uint64_t>> inv(L, std::vector<std::uint64_t>(L, 0));
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < i; ++j) inv[j][i] = shim::powmod(q[j] % q[i], q[i] - 2, q[i]);
        const int NL = L + 1;
        std::vector<std::uint64_t> Q(NL, 0), half(NL, 0);
        Q[0] = 1;
        for (int j = 0; j < L; ++j) mul_add(Q, q[j], 0);
        for (int i = 0; i < NL; ++i) half[i] = (Q[i] >> 1) | (i + 1 < NL ? Q[i + 1] << 63 : 0);
        std::vector<std::complex<double>> A(n);
        std::vector<std::uint64_t> dig(L), X(NL), Y(NL);
        for (std::size_t a = 0; a < n; ++a) {
            for (int i = 0; i < L; ++i) {
                std::uint64_t t = co[(std::size_t)i * n + a];
                for (int j = 0; j < i; ++j) {
                    const std::uint64_t dj = dig[j] % q[i];
                    t = shim::mulmod(t >= dj ? t - dj : t + q[i] - dj, inv[j][i], q[i]);
                }
                dig[i] = t;
            }
            std::fill(X.begin(), X.end(), 0);
            X[0] = dig[L - 1];
            for (int i = L - 2; i >= 0; --i) mul_add(X, q[i], dig[i]);
            double v;
            if (cmp(X, half) > 0) {
                sub(Y, Q, X);
                v = -to_double(Y);
            } else
                v = to_double(X);
            A[a] = (v / plain.scale()) * zeta_[a];
        }
        fft(A, true);  // P(zeta^(2r+1)) = sum_k p_k zeta^k e^{+2 pi i r k / N}
        dest.resize(n / 2);
        for (std::size_t i = 0; i < n / 2; ++i) dest[i] = A[r1_[i]].real();
    }

private:
    int rows_checked(const parms_id_type &id, double scale) const
    {
        const int L = ctx_->rows_of(id);
        if (L == 0) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (scale <= 0 || (int)std::log2(scale) >= ctx_->get_context_data(id)->total_coeff_modulus_bit_count())
            throw std::invalid_argument("scale out of bounds");
        return L;
    }
    // hefx_ckks_encode (FFT + rounding + RNS + NTT on the GPU) when N is in the kernel's range, every coefficient
    // provably fits 62 bits (|p_k| <= max|v|) and zero-ness follows from norms (Parseval: max|p_k| >=
    // sqrt(2 sum v^2)/N); otherwise the host FFT below.  SEAL_SHIM_HOST_ENCODE=1 forces the host path.
    bool encode_on_device(const std::vector<double> &values, int L, const parms_id_type &id, double scale,
                          Plaintext &dest) const
    {
        const std::size_t n = ctx_->n();
        if (n < 1024 || n > 32768 || values.empty()) return false;
        static const bool host_only = [] {
            const char *s = std::getenv("SEAL_SHIM_HOST_ENCODE");
            return s && *s && *s != '0';
        }();
        if (host_only) return false;
        double mx = 0, ss = 0;
        for (double v : values) {
            if (!std::isfinite(v)) return false;
            mx = std::max(mx, std::fabs(v));
            ss += v * v;
        }
        if (mx * scale >= 4611686018427387904.0) return false;
        const bool zero = mx * scale < 0.499, nonzero = std::sqrt(2.0 * ss) / (double)n * scale > 0.501;
        if (!zero && !nonzero) return false;
        auto &e = ctx_->engine();
        if (e->lazy && !shim::sync_mode()) {
            // recorded: the encodes of a loop (2000 one-hot masks in logistic_regression_ckks.cpp:222-225, 4018 vectors in
            // front of the training loop) go to the device as one hefx_ckks_encode_batch per (level, length, scale)
            shim::Engine::Node x{};
            x.host = std::make_shared<std::vector<double>>(values);
            x.scale = scale;
            dest.buf = e->record(shim::Engine::Node::ENCODE, nullptr, nullptr, 0, L, (int)values.size(), (std::size_t)L * n, e, &x);
            dest.view_words_ = 0;
            finish(dest, L, id, scale, zero);
            return true;
        }
        dest.buf = shim::new_buf(e, (std::size_t)L * n);
        dest.view_words_ = 0;
        shim::check(hefx_ckks_encode(e->ready({}), L, values.data(), nullptr, (int)values.size(), 1, scale, dest.buf->p, nullptr));
        finish(dest, L, id, scale, zero);
        return true;
    }
    void finish(Plaintext &dest, int L, const parms_id_type &id, double scale, bool zero) const
    {
        dest.rows = L;
        dest.view_words_ = 0;
        dest.parms_id() = id;
        dest.scale() = scale;
        dest.zero_ = zero;
    }
    void put(std::vector<std::uint64_t> &rows, std::size_t i, double co, int L) const
    {
        const std::size_t n = ctx_->n();
        const bool neg = co < 0;
        const shim::u128 mag = (shim::u128)std::fabs(co);
        for (int j = 0; j < L; ++j) {
            const std::uint64_t q = ctx_->primes()[j], r = (std::uint64_t)(mag % q);
            rows[(std::size_t)j * n + i] = neg ? (r ? q - r : 0) : r;
        }
    }
    // in-place radix-2 DFT of size N; inverse=false: e^{-i}, scaled by nothing (caller divides); inverse=true: e^{+i}
    void fft(std::vector<std::complex<double>> &v, bool positive) const
    {
        const std::size_t n = v.size();
        for (std::size_t i = 1, j = 0; i < n; ++i) {
            std::size_t bit = n >> 1;
            for (; j >= bit; bit >>= 1) j -= bit;
            j += bit;
            if (i < j) std::swap(v[i], v[j]);
        }
        for (std::size_t len = 2; len <= n; len <<= 1) {
            const std::size_t h = len >> 1, step = n / len;
            for (std::size_t i = 0; i < n; i += len)
                for (std::size_t j = 0; j < h; ++j) {
                    const std::complex<double> w = positive ? std::conj(tw_[j * step]) : tw_[j * step];
                    const std::complex<double> a = v[i + j], b = v[i + j + h] * w;
                    v[i + j] = a + b;
                    v[i + j + h] = a - b;
                }
        }
    }
    static void mul_add(std::vector<std::uint64_t> &x, std::uint64_t m, std::uint64_t a)
    {
        shim::u128 carry = a;
        for (auto &limb : x) {
            const shim::u128 t = (shim::u128)limb * m + carry;
            limb = (std::uint64_t)t;
            carry = t >> 64;
        }
    }
    static int cmp(const std::vector<std::uint64_t> &a, const std::vector<std::uint64_t> &b)
    {
        for (int i = (int)a.size() - 1; i >= 0; --i)
            if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
        return 0;
    }
    static void sub(std::vector<std::uint64_t> &r, const std::vector<std::uint64_t> &a, const std::vector<std::uint64_t> &b)
    {
        std::uint64_t borrow = 0;
        for (std::size_t i = 0; i < a.size(); ++i) {
            const std::uint64_t t = a[i] - b[i], b1 = a[i] < b[i], t2 = t - borrow, b2 = t < borrow;
            r[i] = t2;
            borrow = b1 | b2;
        }
    }
    static double to_double(const std::vector<std::uint64_t> &a)
    {
        double r = 0;
        for (int i = (int)a.size() - 1; i >= 0; --i) r = r * 18446744073709551616.0 + (double)a[i];
        return r;
    }

    std::shared_ptr<SEALContext> ctx_;
    std::vector<std::size_t> r1_, r2_;
    std::vector<std::complex<double>> zeta_, tw_;
};

// BatchEncoder (BFV; vector_ops.cpp:127-193, 5_rotation.cpp:108-164): N slots as a 2 x N/2 matrix, slot i <-> the
// evaluation point psi^(3^i) (row 0) / psi^(-3^i) (row 1), so that X -> X^(3^k) rotates the rows by k and X -> X^(2N-1)
// swaps them.  Host arithmetic modulo the plain modulus (shim_bfv.h PlainNtt).
class BatchEncoder {
public:
    template <class Ctx>
    explicit BatchEncoder(const Ctx &context) : ctx_(shim::as_ptr(context))
    {
        if (ctx_->is_ckks()) throw std::invalid_argument("unsupported scheme");
        if (!ctx_->bfv().batching) throw std::invalid_argument("encryption parameters are not valid for batching");
        const std::size_t n = ctx_->n();
        int logn = 0;
        while (((std::size_t)1 << logn) < n) ++logn;
        index_.resize(n);
        std::uint64_t pos = 1;
        for (std::size_t i = 0; i < n / 2; ++i) {
            index_[i] = shim::bfv::PlainNtt::bitrev((std::uint32_t)((pos - 1) >> 1), logn);
            index_[n / 2 + i] = shim::bfv::PlainNtt::bitrev((std::uint32_t)((2 * n - pos - 1) >> 1), logn);
            pos = (pos * 3) & (2 * n - 1);
        }
    }
    std::size_t slot_count() const { return ctx_->n(); }
    void encode(const std::vector<std::uint64_t> &values, Plaintext &dest) const
    {
        const std::size_t n = ctx_->n();
        const std::uint64_t t = ctx_->plain_modulus_value();
        if (values.size() > n) throw std::invalid_argument("values has invalid size");
        dest = Plaintext();
        dest.bfv.assign(n, 0);
        for (std::size_t i = 0; i < values.size(); ++i) {
            if (values[i] >= t) throw std::invalid_argument("input value is larger than plain_modulus");
            dest.bfv[index_[i]] = values[i];
        }
        ctx_->bfv().pntt.inverse(dest.bfv);
    }
    void encode(const std::vector<std::int64_t> &values, Plaintext &dest) const
    {
        const std::int64_t t = (std::int64_t)ctx_->plain_modulus_value();
        std::vector<std::uint64_t> u(values.size());
        for (std::size_t i = 0; i < values.size(); ++i) {
            if (values[i] > t / 2 || values[i] < -(t / 2)) throw std::invalid_argument("input value is larger than plain_modulus");
            u[i] = (std::uint64_t)(values[i] < 0 ? values[i] + t : values[i]);
        }
        encode(u, dest);
    }
    void decode(const Plaintext &plain, std::vector<std::uint64_t> &dest) const
    {
        const std::size_t n = ctx_->n();
        if (plain.bfv.empty() || plain.bfv.size() > n) throw std::invalid_argument("plain is not valid for encryption parameters");
        std::vector<std::uint64_t> tmp(plain.bfv);
        tmp.resize(n, 0);
        ctx_->bfv().pntt.forward(tmp);
        dest.resize(n);
        for (std::size_t i = 0; i < n; ++i) dest[i] = tmp[index_[i]];
    }
    void decode(const Plaintext &plain, std::vector<std::int64_t> &dest) const
    {
        std::vector<std::uint64_t> u;
        decode(plain, u);
        const std::uint64_t t = ctx_->plain_modulus_value();
        dest.resize(u.size());
        for (std::size_t i = 0; i < u.size(); ++i) dest[i] = u[i] > t / 2 ? (std::int64_t)u[i] - (std::int64_t)t : (std::int64_t)u[i];
    }

private:
    std::shared_ptr<SEALContext> ctx_;
    std::vector<std::size_t> index_;
};
// IntegerEncoder (BFV; 2_encoders.cpp:113-147): an integer as the polynomial of its binary digits, negative numbers with
// coefficients t-1; decoding evaluates the polynomial at 2 with centred coefficients.
class IntegerEncoder {
public:
    template <class Ctx>
    explicit IntegerEncoder(const Ctx &context) : ctx_(shim::as_ptr(context))
    {
        if (ctx_->is_ckks()) throw std::invalid_argument("unsupported scheme");
        if (ctx_->plain_modulus_value() < 2) throw std::invalid_argument("plain_modulus must be at least 2");
    }
    Plaintext encode(std::uint64_t value) const
    {
        Plaintext p;
        p.bfv.assign(1, 0);
        for (int i = 0; value; ++i, value >>= 1) {
            if ((int)p.bfv.size() <= i) p.bfv.resize(i + 1, 0);
            p.bfv[i] = value & 1;
        }
        return p;
    }
    Plaintext encode(std::int64_t value) const
    {
        if (value >= 0) return encode((std::uint64_t)value);
        Plaintext p = encode((std::uint64_t)(-(value + 1)) + 1);
        const std::uint64_t minus_one = ctx_->plain_modulus_value() - 1;
        for (auto &c : p.bfv)
            if (c) c = minus_one;
        return p;
    }
    Plaintext encode(std::int32_t value) const { return encode((std::int64_t)value); }
    Plaintext encode(std::uint32_t value) const { return encode((std::uint64_t)value); }
    template <class Int>
    void encode(Int value, Plaintext &dest) const { dest = encode(value); }
    std::int64_t decode_int64(const Plaintext &p) const
    {
        const std::uint64_t t = ctx_->plain_modulus_value();
        __int128 acc = 0;
        for (std::size_t i = p.bfv.size(); i-- > 0;) {
            const std::uint64_t c = p.bfv[i];
            if (c >= t) throw std::invalid_argument("plain does not represent a valid plaintext polynomial");
            acc = acc * 2 + (c > t / 2 ? (__int128)c - (__int128)t : (__int128)c);
            if (acc > (__int128)INT64_MAX || acc < (__int128)INT64_MIN) throw std::invalid_argument("output out of range");
        }
        return (std::int64_t)acc;
    }
    std::int32_t decode_int32(const Plaintext &p) const
    {
        const std::int64_t v = decode_int64(p);
        if (v > INT32_MAX || v < INT32_MIN) throw std::invalid_argument("output out of range");
        return (std::int32_t)v;
    }
    std::uint64_t decode_uint64(const Plaintext &p) const
    {
        const std::int64_t v = decode_int64(p);
        if (v < 0) throw std::invalid_argument("output out of range");
        return (std::uint64_t)v;
    }
    std::uint32_t decode_uint32(const Plaintext &p) const { return (std::uint32_t)decode_uint64(p); }

private:
    std::shared_ptr<SEALContext> ctx_;
};

// ------------------------------------------------------------------------------------------------
// Evaluator: every method is a hefx_* dispatch (GPU); SEAL's checks and messages stay on the host
// ------------------------------------------------------------------------------------------------
class Evaluator {
public:
    template <class Ctx>
    explicit Evaluator(const Ctx &context) : ctx_(shim::as_ptr(context))
    {
    }

    // ---- add / sub / negate (helper.h:219,247,259,464,475; logistic_regression_ckks.cpp:288,341-342)
    void add(const Ciphertext &a, const Ciphertext &b, Ciphertext &dest) const { addsub(a, b, dest, false); }
    void add_inplace(Ciphertext &a, const Ciphertext &b) const { addsub(a, b, a, false); }
    void sub(const Ciphertext &a, const Ciphertext &b, Ciphertext &dest) const { addsub(a, b, dest, true); }
    void sub_inplace(Ciphertext &a, const Ciphertext &b) const { addsub(a, b, a, true); }
    void add_many(const std::vector<Ciphertext> &cts, Ciphertext &dest) const
    {
        if (cts.empty()) throw std::invalid_argument("encrypteds cannot be empty");
        bool uniform = true;
        for (auto &c : cts) {
            check_ct(c);
            uniform = uniform && c.size() == cts[0].size();
            if (c.parms_id() != cts[0].parms_id()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
            if (!close(c.scale(), cts[0].scale())) throw std::invalid_argument("scale mismatch");
        }
        if (!uniform) {  // SEAL's definition: dest = cts[0]; add_inplace the rest
            Ciphertext acc = cts[0];
            for (std::size_t i = 1; i < cts.size(); ++i) add_inplace(acc, cts[i]);
            dest = acc;
            return;
        }
        auto &e = eng();
        const int L = cts[0].rows;
        std::vector<const std::uint64_t *> ptrs;
        for (auto &c : cts) ptrs.push_back(c.buf->p);
        auto out = shim::new_buf(e, words(cts[0].size(), L));
        shim::check(hefx_add_many(e->live(), L, (int)cts[0].size(), (int)cts.size(), ptrs.data(), out->p, nullptr));
        dest.set(out, cts[0].size(), L, cts[0].parms_id(), cts[0].scale());
    }
    void negate(const Ciphertext &a, Ciphertext &dest) const
    {
        check_ct(a);
        auto out = shim::new_buf(eng(), a.buf->words);
        shim::check(hefx_negate(eng()->ready({a.buf.get()}), a.rows, (int)a.size(), 1, a.buf->p, out->p, nullptr));
        dest.set(out, a.size(), a.rows, a.parms_id(), a.scale());
    }
    void negate_inplace(Ciphertext &a) const { negate(a, a); }
    void add_plain(const Ciphertext &a, const Plaintext &p, Ciphertext &dest) const
    {
        check_ct(a);
        if (!ctx_->is_ckks()) {  // BFV: c0 += Delta * m (vector_ops.cpp:178)
            if (p.bfv.empty()) throw std::invalid_argument("plain is not valid for encryption parameters");
            auto dm = shim::upload(eng(), shim::bfv_scaled_plain(*ctx_, p.bfv, a.rows));
            auto out = shim::new_buf(eng(), a.buf->words);
            shim::check(hefx_add_plain(eng()->ready({a.buf.get()}), a.rows, (int)a.size(), a.buf->p, dm->p, out->p, nullptr));
            dest.set(out, a.size(), a.rows, a.parms_id(), a.scale());
            dest.ntt_form_ = false;
            return;
        }
        check_pt(a, p);
        if (!close(a.scale(), p.scale())) throw std::invalid_argument("scale mismatch");
        auto out = shim::new_buf(eng(), a.buf->words);
        shim::check(hefx_add_plain(eng()->ready({a.buf.get(), p.buf.get()}), a.rows, (int)a.size(), a.buf->p, p.buf->p, out->p, nullptr));
        dest.set(out, a.size(), a.rows, a.parms_id(), a.scale());
    }
    void add_plain_inplace(Ciphertext &a, const Plaintext &p) const { add_plain(a, p, a); }

    // ---- multiply (helper.h:222,228,250,256,432; matrix_multiplication.cpp:104,127; vector_ops.cpp:269)
    void multiply_plain(const Ciphertext &a, const Plaintext &p, Ciphertext &dest) const
    {
        check_ct(a);
        if (!ctx_->is_ckks()) {  // BFV (1_bfv.cpp:131): ring product with the lifted plaintext, through the NTT domain
            if (p.bfv.empty()) throw std::invalid_argument("plain is not valid for encryption parameters");
            bool zero = true;
            for (std::uint64_t m : p.bfv) zero = zero && m == 0;
            if (zero) throw std::logic_error("result ciphertext is transparent");
            auto &e = eng();
            const int L = a.rows;
            auto pt = shim::upload(e, shim::bfv_lifted_plain(*ctx_, p.bfv, L));
            auto tmp = shim::new_buf(e, a.buf->words), out = shim::new_buf(e, a.buf->words);
            shim::check(hefx_ntt_forward(e->ready({}), pt->p, 1, L, 0, nullptr));
            shim::check(hefx_copy(e->ready({a.buf.get()}), tmp->p, a.buf->p, a.buf->words * 8, nullptr));
            shim::check(hefx_ntt_forward(e->ready({}), tmp->p, (int)a.size(), L, 0, nullptr));
            shim::check(hefx_multiply_plain(e->ready({}), L, (int)a.size(), 1, tmp->p, pt->p, out->p, nullptr));
            shim::check(hefx_ntt_inverse(e->ready({}), out->p, (int)a.size(), L, 0, nullptr));
            dest.set(out, a.size(), L, a.parms_id(), a.scale());
            dest.ntt_form_ = false;
            return;
        }
        check_pt(a, p);
        const double ns = a.scale() * p.scale();
        check_scale(ns, a.parms_id());
        // a valid ciphertext's c1 is uniformly random, so the product is transparent exactly when the plaintext
        // is zero -- known on the host since encode time; no device sync needed (why the reference adds 1e-8).
        if (p.is_zero()) throw std::logic_error("result ciphertext is transparent");
        if (eng()->lazy) {  // recorded; a product of a recorded rotation rides in that key switch's epilogue
            const std::size_t sz = a.size();
            const int rows = a.rows;
            const parms_id_type id = a.parms_id();
            dest.set(eng()->record(shim::Engine::Node::MULPT, a.buf, p.buf, 0, rows, (int)sz, a.buf->words, eng()), sz,
                     rows, id, ns);
            return;
        }
        auto out = shim::new_buf(eng(), a.buf->words);
        shim::check(hefx_multiply_plain(eng()->live(), a.rows, (int)a.size(), 1, a.buf->p, p.buf->p, out->p, nullptr));
        dest.set(out, a.size(), a.rows, a.parms_id(), ns);
    }
    void multiply_plain_inplace(Ciphertext &a, const Plaintext &p) const { multiply_plain(a, p, a); }
    void multiply(const Ciphertext &a, const Ciphertext &b, Ciphertext &dest) const
    {
        check_ct(a);
        check_ct(b);
        if (a.parms_id() != b.parms_id()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (!ctx_->is_ckks()) return multiply_bfv(a, b, dest);  // any sizes (1_bfv.cpp:132 multiplies two size-3 ciphertexts)
        if (a.size() != 2 || b.size() != 2)
            throw std::invalid_argument("multiply: only size-2 operands are built (every reference call site)");
        const double ns = a.scale() * b.scale();
        check_scale(ns, a.parms_id());
        if (eng()->lazy) {  // recorded (helper.h:227-228, 432): the products of all rows / diagonals go out as one batch
            const int rows = a.rows;
            const parms_id_type id = a.parms_id();
            dest.set(eng()->record(shim::Engine::Node::MULCT, a.buf, b.buf, 0, rows, 2, words(3, rows), eng()), 3, rows, id,
                     ns);
            return;
        }
        auto out = shim::new_buf(eng(), words(3, a.rows));
        if (a.buf == b.buf)
            shim::check(hefx_square(eng()->live(), a.rows, a.buf->p, out->p, nullptr));
        else
            shim::check(hefx_multiply(eng()->live(), a.rows, a.buf->p, b.buf->p, out->p, nullptr));
        dest.set(out, 3, a.rows, a.parms_id(), ns);
    }
    void multiply_inplace(Ciphertext &a, const Ciphertext &b) const { multiply(a, b, a); }
    void square(const Ciphertext &a, Ciphertext &dest) const { multiply(a, a, dest); }
    void square_inplace(Ciphertext &a) const { multiply(a, a, a); }

    // ---- relinearize / rescale / mod switch (helper.h:440-441; matrix_multiplication.cpp:71-72,112)
    void relinearize_inplace(Ciphertext &a, const RelinKeys &rk) const
    {
        check_ct(a);
        if (a.size() == 2) return;  // SEAL: nothing to do
        if (a.size() != 3) throw std::invalid_argument("encrypted size must be 2 or 3");
        if (!rk.has_key(0)) throw std::invalid_argument("not enough relinearization keys");
        if (!ctx_->is_ckks()) {  // BFV: the same key switch, bracketed by NTTs (the ciphertext is in coefficient form)
            auto &e = eng();
            auto tmp = shim::new_buf(e, a.buf->words), out = shim::new_buf(e, words(2, a.rows));
            shim::check(hefx_copy(e->ready({a.buf.get()}), tmp->p, a.buf->p, a.buf->words * 8, nullptr));
            shim::check(hefx_ntt_forward(e->ready({}), tmp->p, 3, a.rows, 0, nullptr));
            shim::check(hefx_relinearize(e->ready({}), a.rows, tmp->p, rk.keys.at(0)->p, out->p, nullptr));
            shim::check(hefx_ntt_inverse(e->ready({}), out->p, 2, a.rows, 0, nullptr));
            a.set(out, 2, a.rows, a.parms_id(), a.scale());
            a.ntt_form_ = false;
            return;
        }
        if (eng()->lazy) {
            a.set(eng()->record(shim::Engine::Node::RELIN, a.buf, rk.keys.at(0), 0, a.rows, 3, words(2, a.rows), eng()), 2,
                  a.rows, a.parms_id(), a.scale());
            return;
        }
        auto out = shim::new_buf(eng(), words(2, a.rows));
        shim::check(hefx_relinearize(eng()->live(), a.rows, a.buf->p, rk.keys.at(0)->p, out->p, nullptr));
        a.set(out, 2, a.rows, a.parms_id(), a.scale());
    }
    void relinearize(const Ciphertext &a, const RelinKeys &rk, Ciphertext &dest) const
    {
        dest = a;
        relinearize_inplace(dest, rk);
    }
    void rescale_to_next(const Ciphertext &a, Ciphertext &dest) const
    {
        check_ct(a);
        if (a.rows <= 1) throw std::invalid_argument("end of modulus switching chain reached");
        const std::size_t sz = a.size();
        const int rows = a.rows;
        const double ns = a.scale() / (double)ctx_->primes()[rows - 1];
        if (eng()->lazy) {
            dest.set(eng()->record(shim::Engine::Node::RESCALE, a.buf, nullptr, 0, rows, (int)sz, words(sz, rows - 1), eng()),
                     sz, rows - 1, ctx_->id_of_rows(rows - 1), ns);
            return;
        }
        auto out = shim::new_buf(eng(), words(sz, rows - 1));
        shim::check(hefx_rescale_to_next(eng()->live(), rows, (int)sz, 1, a.buf->p, out->p, nullptr));
        dest.set(out, sz, rows - 1, ctx_->id_of_rows(rows - 1), ns);
    }
    void rescale_to_next_inplace(Ciphertext &a) const { rescale_to_next(a, a); }
    void mod_switch_to_inplace(Ciphertext &a, const parms_id_type &id) const
    {
        check_ct(a);
        const int L = target_rows(a.rows, id);
        if (L == a.rows) return;
        if (!ctx_->is_ckks()) {  // BFV (3_levels.cpp:103): divide by the dropped primes and round, one prime at a time
            auto &e = eng();
            shim::BufPtr cur = shim::new_buf(e, a.buf->words);
            shim::check(hefx_copy(e->ready({a.buf.get()}), cur->p, a.buf->p, a.buf->words * 8, nullptr));
            shim::check(hefx_ntt_forward(e->ready({}), cur->p, (int)a.size(), a.rows, 0, nullptr));
            for (int rows = a.rows; rows > L; --rows) {
                auto out = shim::new_buf(e, words(a.size(), rows - 1));
                shim::check(hefx_rescale_to_next_mode(e->ready({}), rows, (int)a.size(), 1, cur->p, out->p, HEFX_RESCALE_ROUND, nullptr));
                cur = out;
            }
            shim::check(hefx_ntt_inverse(e->ready({}), cur->p, (int)a.size(), L, 0, nullptr));
            a.set(cur, a.size(), L, id, a.scale());
            a.ntt_form_ = false;
            return;
        }
        auto out = shim::new_buf(eng(), words(a.size(), L));
        shim::check(hefx_mod_drop(eng()->ready({a.buf.get()}), a.rows, L, (int)a.size(), a.buf->p, out->p, nullptr));
        a.set(out, a.size(), L, id, a.scale());
    }
    void mod_switch_to_inplace(Plaintext &p, const parms_id_type &id) const
    {
        if (!p.buf) throw std::invalid_argument("plain is not valid for encryption parameters");
        const int L = target_rows(p.rows, id);
        if (L == p.rows) return;
        // a CKKS plaintext is [rows][N]: dropping its last primes is the PREFIX of the same payload -- no copy, no engine
        // call, and a recorded encode (logistic_regression_ckks.cpp:225-227: encode, mod_switch_to_next, multiply_plain per
        // observation row) stays recorded
        p.rows = L;
        p.view_words_ = words(1, L);
        p.parms_id() = id;
    }
    void mod_switch_to_next_inplace(Ciphertext &a) const
    {
        if (a.rows <= 1) throw std::invalid_argument("end of modulus switching chain reached");
        mod_switch_to_inplace(a, ctx_->id_of_rows(a.rows - 1));
    }
    void mod_switch_to_next_inplace(Plaintext &p) const
    {
        if (p.rows <= 1) throw std::invalid_argument("end of modulus switching chain reached");
        mod_switch_to_inplace(p, ctx_->id_of_rows(p.rows - 1));
    }
    void mod_switch_to_next(const Ciphertext &a, Ciphertext &dest) const
    {
        dest = a;
        mod_switch_to_next_inplace(dest);
    }
    void mod_switch_to(const Ciphertext &a, const parms_id_type &id, Ciphertext &dest) const
    {
        dest = a;
        mod_switch_to_inplace(dest, id);
    }

    // ---- rotations (helper.h:216,227,244,255,316,352,455,474; 5_rotation.cpp:215): rotate_internal of
    //      App. A.7 -- direct key if present, otherwise the NAF chain, least significant term first
    void rotate_vector(const Ciphertext &a, int steps, const GaloisKeys &gk, Ciphertext &dest) const
    {
        check_ct(a);
        if (!ctx_->is_ckks()) throw std::logic_error("unsupported scheme");
        if (a.size() != 2) throw std::invalid_argument("encrypted size must be 2");
        static thread_local std::vector<std::uint32_t> plan;  // reused: no allocation per recorded rotation
        plan.clear();
        rotation_plan(steps, gk, plan);
        shim::BufPtr cur = a.buf;
        for (std::uint32_t elt : plan) {
            if (eng()->lazy) {  // recorded; runs batched with its siblings when a result is observed
                cur = eng()->record(shim::Engine::Node::ROT, cur, gk.keys.at(elt), elt, a.rows, 2, words(2, a.rows), eng());
                continue;
            }
            auto out = shim::new_buf(eng(), words(2, a.rows));
            shim::check(hefx_apply_galois(eng()->live(), a.rows, cur->p, elt, gk.keys.at(elt)->p, out->p, nullptr));
            cur = out;
        }
        dest.set(cur, 2, a.rows, a.parms_id(), a.scale());
    }
    void rotate_vector_inplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { rotate_vector(a, steps, gk, a); }
    void complex_conjugate(const Ciphertext &a, const GaloisKeys &gk, Ciphertext &dest) const
    {
        check_ct(a);
        const std::uint32_t elt = (std::uint32_t)(2 * ctx_->n() - 1);
        if (!gk.has_key(elt)) throw std::invalid_argument("Galois key not present");
        if (eng()->lazy) {
            dest.set(eng()->record(shim::Engine::Node::ROT, a.buf, gk.keys.at(elt), elt, a.rows, 2, words(2, a.rows), eng()), 2,
                     a.rows, a.parms_id(), a.scale());
            return;
        }
        auto out = shim::new_buf(eng(), words(2, a.rows));
        shim::check(hefx_apply_galois(eng()->live(), a.rows, a.buf->p, elt, gk.keys.at(elt)->p, out->p, nullptr));
        dest.set(out, 2, a.rows, a.parms_id(), a.scale());
    }
    void complex_conjugate_inplace(Ciphertext &a, const GaloisKeys &gk) const { complex_conjugate(a, gk, a); }
    // SEAL 3.4's public apply_galois: one automorphism + key switch with a directly keyed element
    void apply_galois(const Ciphertext &a, std::uint64_t galois_elt, const GaloisKeys &gk, Ciphertext &dest) const
    {
        check_ct(a);
        if (a.size() != 2) throw std::invalid_argument("encrypted size must be 2");
        const std::uint32_t elt = (std::uint32_t)galois_elt;
        if (!(elt & 1) || elt >= 2 * ctx_->n()) throw std::invalid_argument("Galois element is not valid");
        if (!gk.has_key(elt)) throw std::invalid_argument("Galois key not present");
        if (!ctx_->is_ckks()) return apply_galois_bfv(a, {elt}, gk, dest);
        auto out = shim::new_buf(eng(), words(2, a.rows));
        shim::check(hefx_apply_galois(eng()->ready({a.buf.get()}), a.rows, a.buf->p, elt, gk.keys.at(elt)->p, out->p, nullptr));
        dest.set(out, 2, a.rows, a.parms_id(), a.scale());
    }
    void apply_galois_inplace(Ciphertext &a, std::uint64_t galois_elt, const GaloisKeys &gk) const { apply_galois(a, galois_elt, gk, a); }
    // BFV only (5_rotation.cpp:130-160): the same Galois key switches as rotate_vector / complex_conjugate, bracketed
    // by NTTs because BFV ciphertexts are kept in coefficient form
    void rotate_rows(const Ciphertext &a, int steps, const GaloisKeys &gk, Ciphertext &dest) const
    {
        if (ctx_->is_ckks()) throw std::logic_error("unsupported scheme");
        std::vector<std::uint32_t> plan;
        rotation_plan(steps, gk, plan);
        apply_galois_bfv(a, plan, gk, dest);
    }
    void rotate_rows_inplace(Ciphertext &a, int steps, const GaloisKeys &gk) const { rotate_rows(a, steps, gk, a); }
    void rotate_columns(const Ciphertext &a, const GaloisKeys &gk, Ciphertext &dest) const
    {
        if (ctx_->is_ckks()) throw std::logic_error("unsupported scheme");
        const std::uint32_t elt = (std::uint32_t)(2 * ctx_->n() - 1);
        if (!gk.has_key(elt)) throw std::invalid_argument("Galois key not present");
        apply_galois_bfv(a, {elt}, gk, dest);
    }
    void rotate_columns_inplace(Ciphertext &a, const GaloisKeys &gk) const { rotate_columns(a, gk, a); }

    void rotation_plan(int steps, const GaloisKeys &gk, std::vector<std::uint32_t> &plan) const
    {
        if (steps == 0) return;
        const std::size_t n = ctx_->n();
        const std::uint32_t elt = shim::galois_elt_from_step(steps, n);
        if (gk.has_key(elt)) {
            plan.push_back(elt);
            return;
        }
        int terms[34];
        const int nt = shim::naf_terms(steps, terms);
        if (nt == 1) throw std::invalid_argument("Galois key not present");
        for (int i = 0; i < nt; ++i) {
            if ((std::size_t)std::abs(terms[i]) == n / 2) continue;
            rotation_plan(terms[i], gk, plan);
        }
    }

    // ---- EXTENSION (not part of SEAL): the reference's Linear_Transform_Plain (helper.h:237-262) as one engine
    // call, hefx_linear_transform_plain -- same checks, exceptions and result bits as the op-by-op body in helper.h,
    // with the rotations of one NAF depth batched.  A maintainer switches helper.h to it with one line:
    //     evaluator.hefx_linear_transform_plain(ct, U_diagonals, gal_keys, ct_prime); return ct_prime;
    void hefx_linear_transform_plain(const Ciphertext &ct, const std::vector<Plaintext> &diags, const GaloisKeys &gk,
                                     Ciphertext &dest) const
    {
        check_ct(ct);
        if (ct.size() != 2) throw std::invalid_argument("encrypted size must be 2");
        if (diags.empty()) throw std::invalid_argument("encrypteds cannot be empty");
        const int d = (int)diags.size();
        double ns = 0;
        std::vector<const std::uint64_t *> pts;
        for (const Plaintext &p : diags) {
            check_pt(ct, p);
            const double s = ct.scale() * p.scale();
            check_scale(s, ct.parms_id());
            if (!pts.empty() && !close(ns, s)) throw std::invalid_argument("scale mismatch");
            if (pts.empty()) ns = s;
            if (p.is_zero()) throw std::logic_error("result ciphertext is transparent");
            pts.push_back(p.buf->p);
        }
        std::vector<std::uint32_t> plan;  // SEAL's exceptions for missing keys / too large steps
        rotation_plan(-d, gk, plan);
        for (int l = 1; l < d; ++l) rotation_plan(l, gk, plan);
        std::vector<std::uint32_t> elts;
        std::vector<const std::uint64_t *> keys;
        for (const auto &kv : gk.keys) {
            elts.push_back(kv.first);
            keys.push_back(kv.second->p);
        }
        auto out = shim::new_buf(eng(), words(2, ct.rows));
        shim::check(::hefx_linear_transform_plain(eng()->live(), ct.rows, ct.buf->p, d, pts.data(), (int)keys.size(),
                                                  elts.data(), keys.data(), out->p, nullptr));
        dest.set(out, 2, ct.rows, ct.parms_id(), ns);
    }

    // ---- EXTENSION: several INDEPENDENT Linear_Transform_Plain calls of one dimension in lockstep
    // (hefx_linear_transform_plain_many): dests[t] = Linear_Transform_Plain(cts[t], diag_sets[t], gk), e.g. the sigma transform
    // of ctA and the tau transform of ctB in CC_Matrix_Multiplication (matrix_multiplication.cpp:22-25) -- every dependent
    // launch sequence carries the items of all inputs; checks, exceptions and result bits of the one-transform form.
    void hefx_linear_transform_plain_many(const std::vector<Ciphertext> &cts, const std::vector<std::vector<Plaintext>> &diag_sets,
                                          const GaloisKeys &gk, std::vector<Ciphertext> &dests) const
    {
        if (cts.empty() || cts.size() != diag_sets.size() || cts.size() > 64) throw std::invalid_argument("encrypteds cannot be empty");
        const int d = (int)diag_sets[0].size();
        if (d < 1) throw std::invalid_argument("encrypteds cannot be empty");
        std::vector<const std::uint64_t *> in, pts;
        std::vector<double> ns(cts.size(), 0.0);
        for (std::size_t t = 0; t < cts.size(); ++t) {
            const Ciphertext &ct = cts[t];
            check_ct(ct);
            if (ct.size() != 2) throw std::invalid_argument("encrypted size must be 2");
            if (ct.parms_id() != cts[0].parms_id() || (int)diag_sets[t].size() != d)
                throw std::invalid_argument("encrypteds parameter mismatch");
            bool first = true;
            for (const Plaintext &p : diag_sets[t]) {
                check_pt(ct, p);
                const double s = ct.scale() * p.scale();
                check_scale(s, ct.parms_id());
                if (!first && !close(ns[t], s)) throw std::invalid_argument("scale mismatch");
                if (first) ns[t] = s;
                first = false;
                if (p.is_zero()) throw std::logic_error("result ciphertext is transparent");
                pts.push_back(p.buf->p);
            }
            in.push_back(ct.buf->p);
        }
        std::vector<std::uint32_t> plan;  // SEAL's exceptions for missing keys / too large steps
        rotation_plan(-d, gk, plan);
        for (int l = 1; l < d; ++l) rotation_plan(l, gk, plan);
        std::vector<std::uint32_t> elts;
        std::vector<const std::uint64_t *> keys;
        for (const auto &kv : gk.keys) {
            elts.push_back(kv.first);
            keys.push_back(kv.second->p);
        }
        std::vector<shim::BufPtr> outs;
        std::vector<std::uint64_t *> op;
        for (std::size_t t = 0; t < cts.size(); ++t) {
            outs.push_back(shim::new_buf(eng(), words(2, cts[0].rows)));
            op.push_back(outs.back()->p);
        }
        shim::check(::hefx_linear_transform_plain_many(eng()->live(), cts[0].rows, (int)cts.size(), in.data(), d, pts.data(),
                                                       (int)keys.size(), elts.data(), keys.data(), op.data(), nullptr));
        dests.resize(cts.size());
        for (std::size_t t = 0; t < cts.size(); ++t) dests[t].set(outs[t], 2, cts[0].rows, cts[0].parms_id(), ns[t]);
    }

    // ---- EXTENSION: add_many(multiply_plain(cts[i], pts[i])) in one pass over the operands
    // (hefx_multiply_plain_sum) -- the body of Linear_Transform_CipherMatrix_PlainVector (helper.h:265-278) with the
    // checks, exceptions and result bits of its op-by-op form.
    void hefx_multiply_plain_sum(const std::vector<Ciphertext> &cts, const std::vector<Plaintext> &pts,
                                 Ciphertext &dest) const
    {
        if (cts.empty() || cts.size() > pts.size()) throw std::invalid_argument("encrypteds cannot be empty");
        const Ciphertext &c0 = cts[0];
        double ns = 0;
        std::vector<const std::uint64_t *> cp, pp;
        for (std::size_t i = 0; i < cts.size(); ++i) {
            check_ct(cts[i]);
            if (cts[i].parms_id() != c0.parms_id() || cts[i].size() != c0.size())
                throw std::invalid_argument("encrypteds parameter mismatch");
            check_pt(cts[i], pts[i]);
            const double s = cts[i].scale() * pts[i].scale();
            check_scale(s, c0.parms_id());
            if (i && !close(ns, s)) throw std::invalid_argument("scale mismatch");
            if (!i) ns = s;
            if (pts[i].is_zero()) throw std::logic_error("result ciphertext is transparent");
            cp.push_back(cts[i].buf->p);
            pp.push_back(pts[i].buf->p);
        }
        auto out = shim::new_buf(eng(), words(c0.size(), c0.rows));
        std::uint64_t *op = out->p;
        shim::check(::hefx_multiply_plain_sum(eng()->live(), c0.rows, (int)c0.size(), (int)cp.size(), (int)cp.size(),
                                              cp.data(), pp.data(), &op, nullptr));
        dest.set(out, c0.size(), c0.rows, c0.parms_id(), ns);
    }

    // ---- EXTENSION: Linear_Transform_Plain in baby-step / giant-step form (hefx_linear_transform_plain_bsgs):
    // `shifted_diags[l]` encodes diagonal l shifted right by (l / n1) * n1 slots; gk holds direct keys for -d (or its
    // NAF terms), 1..n1-1 and n1, 2*n1, ...  n1 + ceil(d/n1) - 2 key switches instead of d - 1; a fast mode like the
    // hoisted ones: the plaintext result of helper.h:237-262, not its noise bits.
    void hefx_linear_transform_plain_bsgs(const Ciphertext &ct, const std::vector<Plaintext> &shifted_diags,
                                          const GaloisKeys &gk, int n1, Ciphertext &dest, bool hoisted_baby = true) const
    {
        check_ct(ct);
        if (ct.size() != 2) throw std::invalid_argument("encrypted size must be 2");
        if (shifted_diags.empty()) throw std::invalid_argument("encrypteds cannot be empty");
        const int d = (int)shifted_diags.size();
        double ns = 0;
        std::vector<const std::uint64_t *> pts;
        for (const Plaintext &p : shifted_diags) {
            check_pt(ct, p);
            const double s = ct.scale() * p.scale();
            check_scale(s, ct.parms_id());
            if (!pts.empty() && !close(ns, s)) throw std::invalid_argument("scale mismatch");
            if (pts.empty()) ns = s;
            if (p.is_zero()) throw std::logic_error("result ciphertext is transparent");
            pts.push_back(p.buf->p);
        }
        std::vector<std::uint32_t> elts;
        std::vector<const std::uint64_t *> keys;
        for (const auto &kv : gk.keys) {
            elts.push_back(kv.first);
            keys.push_back(kv.second->p);
        }
        auto out = shim::new_buf(eng(), words(2, ct.rows));
        shim::check(::hefx_linear_transform_plain_bsgs(eng()->live(), ct.rows, ct.buf->p, d, n1, pts.data(),
                                                       (int)keys.size(), elts.data(), keys.data(), hoisted_baby ? 1 : 0,
                                                       out->p, nullptr));
        dest.set(out, 2, ct.rows, ct.parms_id(), ns);
    }

private:
    // BFV tensor product scaled by t/Q (vector_ops.cpp:179 square_inplace): the exact integer products
    // c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1 of the centred operands are formed in an auxiliary RNS basis wide
    // enough to hold them (dyadic products in the NTT domain on the GPU), CRT-composed on the host, multiplied by t,
    // divided by Q with rounding and reduced into the data basis -- the textbook Fan-Vercauteren multiplication.
    void multiply_bfv(const Ciphertext &a, const Ciphertext &b, Ciphertext &dest) const
    {
        namespace bf = shim::bfv;
        const auto &B = ctx_->bfv(a.rows);
        const std::size_t n = ctx_->n();
        const int L = B.L, A = (int)B.aux.m.size();
        const int sa = (int)a.size(), sb = (int)b.size(), sr = sa + sb - 1;
        if (sa < 2 || sb < 2 || sr > 6) throw std::invalid_argument("encrypted1 or encrypted2 has an unsupported size");
        auto lift = [&](const Ciphertext &c) {  // [size][L][N] residues -> centred integers -> [size][A][N] residues
            const std::vector<std::uint64_t> h = shim::download(c.buf);
            const int sz = (int)c.size();
            std::vector<std::uint64_t> out((std::size_t)sz * A * n);
            for (int p = 0; p < sz; ++p)
                for (std::size_t i = 0; i < n; ++i) {
                    bf::Big x = B.data.compose(h.data() + (std::size_t)p * L * n, n, i);
                    const bool neg = bf::cmp(x, B.data.half) > 0;
                    if (neg) x = bf::sub(B.data.M, x);
                    for (int j = 0; j < A; ++j) {
                        const std::uint64_t r = bf::mod_small(x, B.aux.m[j]);
                        out[((std::size_t)p * A + j) * n + i] = neg && r ? B.aux.m[j] - r : r;
                    }
                }
            return out;
        };
        auto &ae = B.aux_engine;
        auto da = shim::upload(ae, lift(a));
        auto db = a.buf == b.buf ? da : shim::upload(ae, lift(b));
        shim::check(hefx_ntt_forward(ae->ready({}), da->p, sa, A, 0, nullptr));
        if (db != da) shim::check(hefx_ntt_forward(ae->ready({}), db->p, sb, A, 0, nullptr));
        const std::size_t poly = (std::size_t)A * n;
        auto prod = shim::new_buf(ae, (std::size_t)sr * poly), tmp = shim::new_buf(ae, poly);
        if (sa == 2 && sb == 2) {
            shim::check(hefx_multiply(ae->ready({}), A, da->p, db->p, prod->p, nullptr));
        } else {  // c_k = sum_{i+j=k} a_i b_j, polynomial by polynomial (dyadic products in the NTT domain)
            shim::check(hefx_memset_zero(ae->ready({}), prod->p, (std::size_t)sr * poly * 8, nullptr));
            for (int i = 0; i < sa; ++i)
                for (int j = 0; j < sb; ++j) {
                    shim::check(hefx_multiply_plain(ae->ready({}), A, 1, 1, da->p + i * poly, db->p + j * poly, tmp->p, nullptr));
                    shim::check(hefx_add(ae->ready({}), A, 1, 1, prod->p + (i + j) * poly, tmp->p, prod->p + (i + j) * poly, nullptr));
                }
        }
        shim::check(hefx_ntt_inverse(ae->ready({}), prod->p, sr, A, 0, nullptr));
        const std::vector<std::uint64_t> hp = shim::download(prod);
        std::vector<std::uint64_t> res((std::size_t)sr * L * n);
        for (int p = 0; p < sr; ++p)
            for (std::size_t i = 0; i < n; ++i) {
                bf::Big x = B.aux.compose(hp.data() + (std::size_t)p * A * n, n, i);
                const bool neg = bf::cmp(x, B.aux.half) > 0;
                if (neg) x = bf::sub(B.aux.M, x);
                bf::Big quo, rem;
                bf::divrem(bf::mul_small(x, B.t), B.data.M, quo, rem);  // round(t |x| / Q)
                if (bf::cmp(rem, B.data.half) > 0) quo = bf::add(quo, bf::Big(1));
                for (int j = 0; j < L; ++j) {
                    const std::uint64_t q = ctx_->primes()[j], r = bf::mod_small(quo, q);
                    res[((std::size_t)p * L + j) * n + i] = neg && r ? q - r : r;
                }
            }
        dest.set(shim::upload(eng(), res), (std::size_t)sr, L, a.parms_id(), a.scale() * b.scale());
        dest.ntt_form_ = false;
    }
    void apply_galois_bfv(const Ciphertext &a, const std::vector<std::uint32_t> &plan, const GaloisKeys &gk, Ciphertext &dest) const
    {
        check_ct(a);
        if (a.size() != 2) throw std::invalid_argument("encrypted size must be 2");
        auto &e = eng();
        auto cur = shim::new_buf(e, a.buf->words);
        shim::check(hefx_copy(e->ready({a.buf.get()}), cur->p, a.buf->p, a.buf->words * 8, nullptr));
        shim::check(hefx_ntt_forward(e->ready({}), cur->p, 2, a.rows, 0, nullptr));
        for (std::uint32_t elt : plan) {
            auto out = shim::new_buf(e, a.buf->words);
            shim::check(hefx_apply_galois(e->ready({}), a.rows, cur->p, elt, gk.keys.at(elt)->p, out->p, nullptr));
            cur = out;
        }
        shim::check(hefx_ntt_inverse(e->ready({}), cur->p, 2, a.rows, 0, nullptr));
        dest.set(cur, 2, a.rows, a.parms_id(), a.scale());
        dest.ntt_form_ = false;
    }
    const std::shared_ptr<shim::Engine> &eng() const { return ctx_->engine(); }
    std::size_t words(std::size_t size, int rows) const { return size * (std::size_t)rows * ctx_->n(); }
    static bool close(double a, double b) { return a == b || std::fabs(a - b) <= std::max(std::fabs(a), std::fabs(b)) * 9.094947017729282e-13; }
    void check_ct(const Ciphertext &c) const
    {
        if (!c.buf || ctx_->rows_of(c.parms_id()) != c.rows) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    }
    void check_pt(const Ciphertext &c, const Plaintext &p) const
    {
        if (!p.buf) throw std::invalid_argument("plain is not valid for encryption parameters");
        if (c.parms_id() != p.parms_id()) throw std::invalid_argument("encrypted_ntt and plain_ntt parameter mismatch");
    }
    void check_scale(double s, const parms_id_type &id) const
    {
        if (s <= 0 || (int)std::log2(s) >= ctx_->get_context_data(id)->total_coeff_modulus_bit_count())
            throw std::invalid_argument("scale out of bounds");
    }
    int target_rows(int rows, const parms_id_type &id) const
    {
        const int L = ctx_->rows_of(id);
        if (L == 0) throw std::invalid_argument("parms_id is not valid for encryption parameters");
        if (L > rows) throw std::invalid_argument("cannot switch to higher level modulus");
        return L;
    }
    void addsub(const Ciphertext &a, const Ciphertext &b, Ciphertext &dest, bool sub) const
    {
        check_ct(a);
        check_ct(b);
        if (a.parms_id() != b.parms_id()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        if (!close(a.scale(), b.scale())) throw std::invalid_argument("scale mismatch");
        auto &e = eng();
        const int L = a.rows;
        const std::size_t mx = std::max(a.size(), b.size()), mn = std::min(a.size(), b.size());
        const bool ntt = a.is_ntt_form();
        struct Mark {  // element-wise: the result has the operands' form (BFV: coefficient form)
            Ciphertext &d;
            bool v;
            ~Mark() { d.ntt_form_ = v; }
        } mark{dest, ntt};
        if (e->lazy && mx == mn) {  // the adds of helper.h:464,475 stay in the lockstep batch
            const parms_id_type id = a.parms_id();
            const double sc = a.scale();
            dest.set(e->record(sub ? shim::Engine::Node::SUB : shim::Engine::Node::ADD, a.buf, b.buf, 0, L, (int)mx,
                               words(mx, L), e),
                     mx, L, id, sc);
            return;
        }
        auto out = shim::new_buf(e, words(mx, L));
        auto f = sub ? hefx_sub : hefx_add;
        shim::check(f(e->live(), L, (int)mn, 1, a.buf->p, b.buf->p, out->p, nullptr));
        if (mx > mn) {  // result size = max; extra polys are copied (negated when they come from b in a sub)
            const Ciphertext &big = a.size() > b.size() ? a : b;
            const std::size_t off = words(mn, L), cnt = words(mx - mn, L);
            if (sub && &big == &b)
                shim::check(hefx_negate(e->live(), L, (int)(mx - mn), 1, b.buf->p + off, out->p + off, nullptr));
            else
                shim::check(hefx_copy(e->live(), out->p + off, big.buf->p + off, cnt * 8, nullptr));
        }
        dest.set(out, mx, L, a.parms_id(), a.scale());
    }

    std::shared_ptr<SEALContext> ctx_;
};

}  // namespace seal
