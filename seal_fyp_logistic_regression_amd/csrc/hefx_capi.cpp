// hefx_capi.cpp -- host side of the C-ABI declared in include/hefx.h: context (prime validation,
// minimal primitive roots, twiddle/constant tables uploaded once to HBM), Galois gather-table cache,
// scratch management and the chunked launch sequences.  No CPU arithmetic fallback exists: every
// compute entry point dispatches HIP kernels from hefx_kernels.hip or fails.
#include "../../include/hefx.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <functional>
#include <vector>

#include <dlfcn.h>
#include <link.h>

#include "hefx_internal.h"

using namespace hefx;
typedef unsigned __int128 u128;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
static int hipfail(hipError_t e, const char *what)
{
    return fail(HEFX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(expr)                                  \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) return hipfail(_e, #expr); \
    } while (0)

// Scoped device selection: every entry point runs with the context's device current and puts the caller's device
// back on return -- a process may hold contexts on several GPUs, and torch (or any other HIP user of the thread) may
// have switched the current device since hefx_context_create.  hipGetDevice is a thread-local read; hipSetDevice is
// only called when the devices differ.
struct DevGuard {
    int prev = -1;
    bool changed = false;
    explicit DevGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard()
    {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};
#define CTXCHK(c)                                                \
    do {                                                         \
        if (!(c)) return fail(HEFX_ERR_INVALID, "null context"); \
    } while (0);                                                 \
    DevGuard _devguard((c)->device)

// ---------------------------------------------------------------------------------------------
// host number theory (context creation only; independent of oracle/)
// ---------------------------------------------------------------------------------------------
static inline u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static u64 h_powmod(u64 a, u64 e, u64 q)
{
    u64 r = 1 % q;
    a %= q;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mulmod(r, a, q);
        a = h_mulmod(a, a, q);
    }
    return r;
}
static inline u64 h_invmod(u64 a, u64 q) { return h_powmod(a, q - 2, q); }
static inline u64 h_shoup(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }

static bool h_is_prime(u64 n)
{
    if (n < 2) return false;
    static const u64 bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : bases) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int r = 0;
    while (!(d & 1)) d >>= 1, ++r;
    for (u64 a : bases) {
        u64 x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int i = 1; i < r && composite; ++i) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) composite = false;
        }
        if (composite) return false;
    }
    return true;
}

// smallest element of exact order 2N (SEAL try_minimal_primitive_root)
static u64 h_min_primitive_root(u64 two_n, u64 q)
{
    const u64 cof = (q - 1) / two_n;
    u64 root = 0;
    for (u64 g = 2; g < 1000 && !root; ++g) {
        u64 c = h_powmod(g, cof, q);
        if (h_powmod(c, two_n / 2, q) == q - 1) root = c;
    }
    if (!root) return 0;
    const u64 sq = h_mulmod(root, root, q);
    u64 best = root, cur = root;
    for (u64 i = 0; i < two_n / 2; ++i) {
        if (cur < best) best = cur;
        cur = h_mulmod(cur, sq, q);
    }
    return best;
}

static inline uint32_t h_bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r = (r << 1) | ((x >> i) & 1);
    return r;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct hefx_context {
    int device = 0;
    uint32_t n = 0;
    int logn = 0;
    int k = 0;
    std::vector<u64> primes, psi;
    DevTables T{};
    void *d_tables = nullptr;  // one allocation holding tw, itw, mods, invmod, halfmod
    std::mutex mu;
    std::unordered_map<uint32_t, uint32_t *> perm;  // Galois element -> device gather table
    // exact hoisting (ks_mac_exact_kernel): Galois element -> NTT_m(flip mask) rows [k][N], built on first use, several
    // elements per allocation; q_i mod q_m [k][k]; one gate word per descriptor-ring slot + the count of chunks redone
    std::unordered_map<uint32_t, u64 *> flipw;
    std::vector<void *> flipw_slabs;
    size_t flipw_bytes = 0, flipw_cap = (size_t)8 << 30;  // tables are kept for the context's life: bounded (HEFX_FLIPW_MB)
    u64 *d_qmod = nullptr;
    uint32_t *d_gate = nullptr;
    uint32_t gate_seq = 0;
    // scratch (grown on demand, reused across calls so it stays cache-resident)
    u64 *scratch = nullptr;
    size_t scratch_words = 0;
    std::vector<u64 *> scratch_retired;  // outgrown scratch buffers, freed with the context (ensure_scratch)
    int chunk = 0;  // items per launch sequence; 0 = sized from the scratch budget (HEFX_CHUNK overrides)
    int *d_flag = nullptr;  // [0] transparent count, [1 + i] "non-zero seen beyond c0" of ciphertext i of the call
    int flag_cap = 1 + 4096;
    void *comm = nullptr;  // ncclComm_t of hefx_comm_init (RCCL, resolved at run time), one per context = per rank
    int comm_world = 0, comm_rank = 0;
    int rescale_mode = HEFX_RESCALE_ROUND;  // default of hefx_rescale_to_next (HEFX_RESCALE=floor|round presets it)
    static constexpr int MAX_STREAMS = 4;
    hipStream_t streams[MAX_STREAMS] = {};  // internal streams for chunk pipelining
    hipEvent_t ev_fork = nullptr, ev_join[MAX_STREAMS] = {};
    int nstreams = 2;
    bool use_streams = true;
    // CKKS encode: tables built on first use, value staging buffer
    void *d_enc_tables = nullptr;
    EncodeTables E{};
    double *d_vals = nullptr;
    size_t vals_cap = 0;
    // pinned staging ring for small encodes (one vector): the caller's array is copied here and the call returns
    // without waiting for the H2D copy
    static constexpr int STAGE_SLOTS = 8;
    double *h_stage = nullptr;
    hipEvent_t stage_ev[STAGE_SLOTS] = {};
    bool stage_busy[STAGE_SLOTS] = {};
    unsigned stage_next = 0;
    // ... and two pinned buffers for large encodes (many vectors in one call), alternating
    double *h_big[2] = {};
    size_t big_cap[2] = {};
    hipEvent_t big_ev[2] = {};
    bool big_busy[2] = {};
    unsigned big_next = 0;
    NoiseTable noise{};  // inverse-CDF thresholds of the clipped normal (sigma 3.2, bound 19.2, truncated)
    // linear-transform workspace (rotated copies and products of one hefx_linear_transform_plain call)
    u64 *lt_ws = nullptr;
    size_t lt_cap = 0;
    // hefx_malloc / hefx_free pool: freed blocks are kept (keyed by their rounded size) and handed out again without
    // a hipFree -- which synchronises the whole device -- or a hipMalloc.  See the contract at hefx_malloc.
    struct PoolSlab {
        void *base;
        size_t block;       // rounded block size
        int blocks, parked;  // blocks carved from it / currently in the free list
    };
    struct PoolBlock {
        size_t size;
        int slab;
    };
    std::vector<PoolSlab> pool_slabs;                             // one hipMalloc each; blocks are carved from them
    std::unordered_map<void *, PoolBlock> pool_block;             // live + parked blocks
    std::unordered_map<size_t, std::vector<void *>> pool_free;    // rounded size -> parked blocks
    std::unordered_map<size_t, int> pool_next;                    // rounded size -> blocks of its next slab (doubles)
    size_t pool_cached = 0, pool_cap = (size_t)64 << 30;          // bytes parked / allowed to stay parked (HEFX_POOL_MB)
    // descriptor ring: pinned host mirror + device copy + "slot free" events
    KsItem *h_items = nullptr, *d_items = nullptr;
    u64 *lt_head = nullptr;  // ping / pong / ct_new / product 0 of hefx_linear_transform_plain (lt_ws: one ciphertext per node)
    size_t lt_head_cap = 0;
    KsItem *chain_items = nullptr;  // device descriptors of hefx_rotate_add_chain's two replayed levels
    hipEvent_t ring_ev[KS_RING] = {};
    bool ring_busy[KS_RING] = {};
    unsigned ring_next = 0;
    // profiling session (hefx_profile_begin/end): one event per launch, serial on the caller's stream
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;
    std::vector<int> prof_stage;
    KsProf prof{nullptr, nullptr, 0, 0};
    size_t prof_chunks = 0;
    int sub = 0;  // items per K2+MAC sub-chunk; 0 = auto (HEFX_SUB overrides)
    // hybrid fused digit-NTT + MAC kernel (ks_ntt_macf_kernel): the FP64-policy target moduli accumulate their key
    // products in registers, only the integer-policy targets' digit x modulus products travel through scratch x.
    // HEFX_FUSED=0/1 overrides the default.
    bool fused = false;
    std::vector<uint8_t> is_f64;  // per prime: FP64 policy in use (mirror of T.modsf[j].q != 0)
    // hefx_ks_stats: key switches submitted / of them exactly hoisted / launch sequences / batched calls (host-side counts)
    uint64_t stat_ks_items = 0, stat_ks_hoisted = 0, stat_ks_chunks = 0, stat_ks_calls = 0;
};

// The scratch buffer grows geometrically and WITHOUT waiting for the device: a buffer that became too small is retired, not
// freed (hipFree synchronises the device, and work already submitted may still be using it) -- retired buffers go with
// the context, and being a geometric series they hold less than the live one.  (Until round 5 every growth was
// hipDeviceSynchronize + hipFree + hipMalloc, ~0.2 ms of host time each plus the drained queue: a one-shot caller like the
// reference's linear_transformation.cpp, whose batches get wider level by level, paid it three times per transform --
// 0.6 of the 1.3 ms its d = 100 case spent submitting.)
// (the key switch's scratch and the linear transforms' node buffers grow this way)
static int grow_retiring(hefx_context *c, u64 **buf, size_t *cap, size_t words, size_t floor_words, const char *what)
{
    if (*cap >= words) return HEFX_OK;
    static const bool dbg = getenv("HEFX_DEBUG") && atoi(getenv("HEFX_DEBUG")) >= 2;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t want = std::max(std::max(words, 2 * *cap), floor_words);
    u64 *fresh = nullptr;
    hipError_t e = hipMalloc((void **)&fresh, want * sizeof(u64));
    size_t got = want;
    if (e != hipSuccess && want > words) {  // no room for the generous size: the exact one
        (void)hipGetLastError();
        got = words;
        e = hipMalloc((void **)&fresh, got * sizeof(u64));
    }
    if (e != hipSuccess) {  // still none: give back what is retired (which needs the device idle), then once more
        (void)hipGetLastError();
        HIPCHK(hipDeviceSynchronize());
        for (u64 *p : c->scratch_retired) (void)hipFree(p);
        c->scratch_retired.clear();
        if (*buf) (void)hipFree(*buf);
        *buf = nullptr;
        *cap = 0;
        HIPCHK(hipMalloc((void **)&fresh, got * sizeof(u64)));
    }
    if (*buf) c->scratch_retired.push_back(*buf);
    *buf = fresh;
    *cap = got;
    if (dbg)
        fprintf(stderr, "[hefx] %s grown to %zu MiB in %.1f us (%zu retired buffers)\n", what, (got * sizeof(u64)) >> 20,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), c->scratch_retired.size());
    return HEFX_OK;
}
static int ensure_scratch(hefx_context *c, size_t words)
{
    // (at least 64 MiB, so that a caller that starts small does not climb a ladder of tiny buffers)
    return grow_retiring(c, &c->scratch, &c->scratch_words, words, (size_t)8 << 20, "scratch");
}

extern "C" const char *hefx_last_error(void) { return g_err.c_str(); }
extern "C" const char *hefx_version(void) { return "hefx 0.1 (gfx950, HIP)"; }

extern "C" int hefx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int hefx_context_create(uint32_t poly_degree, const uint64_t *primes, int k, int device,
                                   hefx_context **out)
{
    if (!out || !primes) return fail(HEFX_ERR_INVALID, "null argument");
    *out = nullptr;
    int logn = 0;
    while ((1u << logn) < poly_degree) ++logn;
    if ((1u << logn) != poly_degree) return fail(HEFX_ERR_INVALID, "poly_degree must be a power of two");
    if (logn < 10 || logn > 15)
        return fail(HEFX_ERR_UNSUPPORTED, "poly_degree must be in [1024, 32768] in this build");
    if (k < 1 || k > 62) return fail(HEFX_ERR_INVALID, "prime count out of range");
    const u64 two_n = 2ull * poly_degree;
    for (int j = 0; j < k; ++j) {
        const u64 q = primes[j];
        if (q >> 61) return fail(HEFX_ERR_INVALID, "primes must be below 2^61");
        if (q % two_n != 1) return fail(HEFX_ERR_INVALID, "prime is not 1 mod 2N");
        if (!h_is_prime(q)) return fail(HEFX_ERR_INVALID, "coeff modulus is not prime");
        for (int i = 0; i < j; ++i)
            if (primes[i] == q) return fail(HEFX_ERR_INVALID, "primes must be distinct");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(HEFX_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(HEFX_ERR_INVALID, "bad device index");
    DevGuard devguard(device);  // tables, streams and events are created on `device`; the caller's device is restored
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device) return fail(HEFX_ERR_HIP, "hipSetDevice failed");
    }

    // HEFX_DEBUG=1: where context creation spends its time (stderr), for tools/first_call_costs2.py
    const bool dbg = getenv("HEFX_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hefx] context_create: %-44s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    lap("device selection (HIP runtime start-up)");
    hefx_context *c = new hefx_context();
    c->device = device;
    c->n = poly_degree;
    c->logn = logn;
    c->k = k;
    c->primes.assign(primes, primes + k);
    c->psi.resize(k);
    if (const char *e = getenv("HEFX_CHUNK")) {
        int v = atoi(e);
        if (v >= 1 && v <= KS_MAX_CHUNK) c->chunk = v;
    }

    const size_t n = poly_degree;
    const size_t tw_bytes = sizeof(ulonglong2) * n * k;
    const size_t mods_bytes = sizeof(ModConst) * k;
    const size_t inv_bytes = sizeof(ulonglong2) * k * k;
    const size_t half_bytes = (sizeof(u64) * k * k + 15) & ~(size_t)15;  // padded: 16-byte records follow
    const size_t modsf_bytes = sizeof(ModConstF) * k;
    const size_t invf_bytes = sizeof(double2) * k * k;
    const size_t twf_bytes = sizeof(double) * n * k;  // FP64 policy: 8-byte twiddles (the quotient estimate is h * 1/q)
    const size_t total = 2 * tw_bytes + 2 * twf_bytes + mods_bytes + inv_bytes + half_bytes + modsf_bytes + invf_bytes;
    std::vector<unsigned char> host(total);
    ulonglong2 *tw = reinterpret_cast<ulonglong2 *>(host.data());
    ulonglong2 *itw = tw + n * k;
    ModConst *mods = reinterpret_cast<ModConst *>(itw + n * k);
    ulonglong2 *invmod = reinterpret_cast<ulonglong2 *>(mods + k);
    u64 *halfmod = reinterpret_cast<u64 *>(invmod + (size_t)k * k);
    double *twf = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(halfmod) + half_bytes);
    double *itwf = twf + n * k;
    ModConstF *modsf = reinterpret_cast<ModConstF *>(itwf + n * k);
    double2 *invmodf = reinterpret_cast<double2 *>(modsf + k);

    for (int j = 0; j < k; ++j) {
        const u64 q = primes[j];
        const u64 psi = h_min_primitive_root(two_n, q);
        if (!psi) {
            delete c;
            return fail(HEFX_ERR_INVALID, "no primitive 2N-th root found");
        }
        c->psi[j] = psi;
        const u64 ipsi = h_invmod(psi, q);
        u64 p = 1, ip = 1;
        for (size_t i = 0; i < n; ++i) {
            const uint32_t r = h_bitrev((uint32_t)i, logn);
            tw[(size_t)j * n + r] = make_ulonglong2(p, h_shoup(p, q));
            itw[(size_t)j * n + r] = make_ulonglong2(ip, h_shoup(ip, q));
            p = h_mulmod(p, psi, q);
            ip = h_mulmod(ip, ipsi, q);
        }
        ModConst &m = mods[j];
        m.q = q;
        const u128 ratio = ~(u128)0 / q;  // floor((2^128-1)/q) == floor(2^128/q) since q is odd > 1
        m.r0 = (u64)ratio;
        m.r1 = (u64)(ratio >> 64);
        m.ninv = h_invmod(n % q, q);
        m.ninv_s = h_shoup(m.ninv, q);
        m.ilw = h_mulmod(itw[(size_t)j * n + 1].x, m.ninv, q);
        m.ilw_s = h_shoup(m.ilw, q);
        m.nq = 0 - q;
        // FP64 policy (exact FMA modmul) for primes below 2^41; HEFX_NO_FP64=1 forces the integer policy
        ModConstF &f = modsf[j];
        memset(&f, 0, sizeof f);
        c->is_f64.resize((size_t)k);
        c->is_f64[(size_t)j] = 0;
        if ((q >> 41) == 0 && !getenv("HEFX_NO_FP64")) {
            c->is_f64[(size_t)j] = 1;
            const double qd = (double)q;
            f.q = qd;
            f.qinv = 1.0 / qd;
            f.ninv = (double)m.ninv;
            f.ninv_r = (double)m.ninv / qd;
            f.ilw = (double)m.ilw;
            f.ilw_r = (double)m.ilw / qd;
            f.c32 = (double)(((u64)1 << 32) % q);
            const u64 c40 = ((u64)1 << 40) % q;
            f.c40 = (q >> 39) == 1 && c40 < ((u64)1 << 23) ? (double)c40 : 0.0;  // only for 2^39 < q < 2^40 close to 2^40
            for (size_t i = 0; i < n; ++i) {
                twf[(size_t)j * n + i] = (double)tw[(size_t)j * n + i].x;
                itwf[(size_t)j * n + i] = (double)itw[(size_t)j * n + i].x;
            }
        }
    }
    for (int l = 0; l < k; ++l)
        for (int j = 0; j < k; ++j) {
            if (l == j) {
                invmod[(size_t)l * k + j] = make_ulonglong2(0, 0);
                invmodf[(size_t)l * k + j] = make_double2(0.0, 0.0);
                halfmod[(size_t)l * k + j] = 0;
                continue;
            }
            const u64 q = primes[j];
            const u64 inv = h_invmod(primes[l] % q, q);
            invmod[(size_t)l * k + j] = make_ulonglong2(inv, h_shoup(inv, q));
            invmodf[(size_t)l * k + j] = make_double2((double)inv, (double)inv / (double)q);  // used only if q < 2^41
            halfmod[(size_t)l * k + j] = (primes[l] >> 1) % q;
        }

    {  // same formula as the CPU checker (orc_noise_thresholds under oracle/)
        const double sigma = 3.2, clip = 19.2, rs2 = 1.0 / (sigma * 1.4142135623730951);
        const double tail = erfc(clip * rs2), tot = 1.0 - tail;
        for (int i = 0; i < 39; ++i) {
            const int kk = -19 + i;
            const double edge = kk >= 0 ? (double)(kk + 1) : (double)kk;
            const double below = edge >= clip ? tot : 0.5 * erfc(-edge * rs2) - 0.5 * tail;
            const double cdf = below / tot;
            if (cdf >= 1.0 || i == 38)
                c->noise.t[i] = ~(u64)0;
            else
                c->noise.t[i] = (u64)((long double)cdf * 18446744073709551616.0L);
        }
    }
    lap("twiddle / constant tables on the host");
    hipError_t e = hipMalloc(&c->d_tables, total);
    if (e == hipSuccess) e = hipMemcpy(c->d_tables, host.data(), total, hipMemcpyHostToDevice);
    lap("first hipMalloc + table upload");
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_flag, c->flag_cap * sizeof(int));
    if (e == hipSuccess) e = hipMemset(c->d_flag, 0, c->flag_cap * sizeof(int));
    if (const char *ev = getenv("HEFX_STREAMS")) {
        const int v = atoi(ev);
        c->use_streams = v != 0;
        if (v >= 2 && v <= hefx_context::MAX_STREAMS) c->nstreams = v;
    }
    // only the internal streams that will be used (two unless HEFX_STREAMS asks for more): a stream is a hardware queue,
    // 8-20 ms each to create -- a visible part of a short process like the reference's drivers
    for (int s = 0; s < c->nstreams && e == hipSuccess; ++s) {
        e = hipStreamCreateWithFlags(&c->streams[s], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join[s], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    lap("internal streams + events");
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_items, sizeof(KsItem) * KS_RING * KS_MAX_CHUNK, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_items, sizeof(KsItem) * KS_RING * KS_MAX_CHUNK);
    for (int s = 0; s < KS_RING && e == hipSuccess; ++s) e = hipEventCreateWithFlags(&c->ring_ev[s], hipEventDisableTiming);
    if (e == hipSuccess) {
        std::vector<u64> qm((size_t)k * k);
        for (int i = 0; i < k; ++i)
            for (int m = 0; m < k; ++m) qm[(size_t)i * k + m] = c->primes[(size_t)i] % c->primes[(size_t)m];
        e = hipMalloc((void **)&c->d_qmod, qm.size() * sizeof(u64));
        if (e == hipSuccess) e = hipMemcpy(c->d_qmod, qm.data(), qm.size() * sizeof(u64), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&c->d_gate, sizeof(uint32_t) * (KS_RING + 1));
        if (e == hipSuccess) e = hipMemset(c->d_gate, 0, sizeof(uint32_t) * (KS_RING + 1));
    }
    lap("descriptor ring (pinned + device), gate words");
    if (const char *sv = getenv("HEFX_SUB")) c->sub = atoi(sv);
    if (const char *fv = getenv("HEFX_FLIPW_MB")) c->flipw_cap = (size_t)strtoull(fv, nullptr, 10) << 20;
    if (const char *pv = getenv("HEFX_POOL_MB")) c->pool_cap = (size_t)strtoull(pv, nullptr, 10) << 20;
    if (const char *fv = getenv("HEFX_FUSED")) c->fused = atoi(fv) != 0;
    if (const char *rv = getenv("HEFX_RESCALE")) c->rescale_mode = !strcmp(rv, "floor") ? HEFX_RESCALE_FLOOR : HEFX_RESCALE_ROUND;
    if (e != hipSuccess) {
        hefx_context_destroy(c);  // frees whatever exists so far (tables, flags, streams, events, ring, gate words)
        return hipfail(e, "context table upload");
    }
    // load every code object now (first-launch cost), not inside the caller's first timed operation
    if (warm_kernels(nullptr) != hipSuccess || warm_keyswitch(nullptr) != hipSuccess || warm_encode(nullptr) != hipSuccess ||
        warm_sample(nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        hefx_context_destroy(c);
        return fail(HEFX_ERR_HIP, "kernel code objects failed to load");
    }
    lap("code objects of the four kernel files (warm launches)");
    unsigned char *base = static_cast<unsigned char *>(c->d_tables);
    c->T.tw = reinterpret_cast<const ulonglong2 *>(base);
    c->T.itw = reinterpret_cast<const ulonglong2 *>(base + tw_bytes);
    c->T.mods = reinterpret_cast<const ModConst *>(base + 2 * tw_bytes);
    c->T.invmod = reinterpret_cast<const ulonglong2 *>(base + 2 * tw_bytes + mods_bytes);
    c->T.halfmod = reinterpret_cast<const u64 *>(base + 2 * tw_bytes + mods_bytes + inv_bytes);
    c->T.twf = reinterpret_cast<const double *>(base + 2 * tw_bytes + mods_bytes + inv_bytes + half_bytes);
    c->T.itwf = c->T.twf + n * k;
    const size_t f_off = 2 * tw_bytes + 2 * twf_bytes + mods_bytes + inv_bytes + half_bytes;
    c->T.modsf = reinterpret_cast<const ModConstF *>(base + f_off);
    c->T.invmodf = reinterpret_cast<const double2 *>(base + f_off + modsf_bytes);
    c->T.k = k;
    c->T.logn = logn;
    *out = c;
    return HEFX_OK;
}

extern "C" int hefx_comm_destroy(hefx_context *c);
extern "C" void hefx_context_destroy(hefx_context *c)
{
    if (!c) return;
    DevGuard devguard(c->device);
    (void)hipDeviceSynchronize();
    (void)hefx_comm_destroy(c);
    for (auto &sl : c->pool_slabs)  // every slab, parked or not: the context's memory ends with the context
        if (sl.base) (void)hipFree(sl.base);
    for (auto &kv : c->perm) (void)hipFree(kv.second);
    for (void *p : c->flipw_slabs) (void)hipFree(p);
    if (c->d_qmod) (void)hipFree(c->d_qmod);
    if (c->d_gate) (void)hipFree(c->d_gate);
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);
    for (int s = 0; s < hefx_context::MAX_STREAMS; ++s) {
        if (c->streams[s]) (void)hipStreamDestroy(c->streams[s]);
        if (c->ev_join[s]) (void)hipEventDestroy(c->ev_join[s]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int s = 0; s < KS_RING; ++s)
        if (c->ring_ev[s]) (void)hipEventDestroy(c->ring_ev[s]);
    if (c->h_items) (void)hipHostFree(c->h_items);
    if (c->d_items) (void)hipFree(c->d_items);
    if (c->chain_items) (void)hipFree(c->chain_items);
    if (c->lt_head) (void)hipFree(c->lt_head);
    if (c->scratch) (void)hipFree(c->scratch);
    for (u64 *p : c->scratch_retired) (void)hipFree(p);  // outgrown scratch / lt_head / lt_ws buffers (grow_retiring)
    c->scratch_retired.clear();
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_enc_tables) (void)hipFree(c->d_enc_tables);
    if (c->d_vals) (void)hipFree(c->d_vals);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (int b = 0; b < 2; ++b) {
        if (c->h_big[b]) (void)hipHostFree(c->h_big[b]);
        if (c->big_ev[b]) (void)hipEventDestroy(c->big_ev[b]);
    }
    for (auto &ev : c->stage_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->lt_ws) (void)hipFree(c->lt_ws);
    delete c;
}

extern "C" uint32_t hefx_poly_degree(const hefx_context *c) { return c ? c->n : 0; }
extern "C" int hefx_prime_count(const hefx_context *c) { return c ? c->k : 0; }
extern "C" uint64_t hefx_prime(const hefx_context *c, int j) { return (c && j >= 0 && j < c->k) ? c->primes[j] : 0; }
extern "C" uint64_t hefx_psi(const hefx_context *c, int j) { return (c && j >= 0 && j < c->k) ? c->psi[j] : 0; }

// ---------------------------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------------------------

// Pooled device memory.  Every evaluator call of the shim / seal.py allocates its result; hipMalloc costs ~4 us (a
// NAF-expanded 1000-diagonal transform allocates 3800 intermediate ciphertexts: 13 ms of hipMalloc around 4.5 ms of
// GPU work) and hipFree waits for the whole device, which serialises an otherwise asynchronous call sequence.
// hefx_malloc therefore carves blocks from slabs -- one hipMalloc for 1, 2, 4, .. 128 blocks of a size class (at most
// 256 MiB) -- and hefx_free parks the block for the next hefx_malloc of that size.  Slabs whose blocks are all parked
// are returned to the driver when more than HEFX_POOL_MB (default 65536) is parked, on out-of-memory and at
// hefx_context_destroy.  No synchronisation is involved in malloc / free: a recycled block may still be read or
// written by work submitted BEFORE the free, and the new owner's work is submitted AFTER the malloc -- correct
// whenever both are ordered on the device (one stream, the model of the shim and of seal.py, or streams the caller has
// ordered with events before freeing).  HEFX_POOL_MB=0 restores plain hipMalloc / hipFree.
static void pool_release(hefx_context *c)  // returns every fully parked slab to the driver (device-synchronising)
{
    (void)hipDeviceSynchronize();
    for (size_t si = 0; si < c->pool_slabs.size(); ++si) {
        hefx_context::PoolSlab &sl = c->pool_slabs[si];
        if (!sl.base || sl.parked != sl.blocks) continue;
        std::vector<void *> &fl = c->pool_free[sl.block];
        for (int b = 0; b < sl.blocks; ++b) {
            void *p = static_cast<char *>(sl.base) + (size_t)b * sl.block;
            c->pool_block.erase(p);
            for (size_t t = 0; t < fl.size(); ++t)
                if (fl[t] == p) {
                    fl[t] = fl.back();
                    fl.pop_back();
                    break;
                }
        }
        c->pool_cached -= (size_t)sl.blocks * sl.block;
        (void)hipFree(sl.base);
        sl.base = nullptr;
    }
}
extern "C" int hefx_malloc(hefx_context *c, size_t bytes, void **d_ptr)
{
    CTXCHK(c);
    if (!d_ptr) return fail(HEFX_ERR_INVALID, "null out pointer");
    const size_t rounded = ((bytes ? bytes : 8) + 255) & ~(size_t)255;
    if (!c->pool_cap) {
        HIPCHK(hipMalloc(d_ptr, rounded));
        return HEFX_OK;
    }
    std::lock_guard<std::mutex> lk(c->mu);
    std::vector<void *> &fl = c->pool_free[rounded];
    if (fl.empty()) {  // a new slab for this size class
        int &next = c->pool_next[rounded];
        if (next < 1) next = 1;
        int n = next;
        while (n > 1 && (size_t)n * rounded > ((size_t)256 << 20)) n >>= 1;
        void *base = nullptr;
        const auto t_alloc = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&base, (size_t)n * rounded);
        {   // a slab normally takes ~20 us; a driver that makes the caller wait (seen on this pool: seconds, with the process
            // asleep) is worth a line on stderr -- it is not the engine's time
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count();
            static const bool dbg = getenv("HEFX_DEBUG") && atoi(getenv("HEFX_DEBUG")) >= 2;
            if (ms > 100.0 || dbg)
                fprintf(stderr, "[hefx] hipMalloc of a %zu MiB pool slab took %.3f ms (pool holds %zu MiB in %zu slabs)\n",
                        ((size_t)n * rounded) >> 20, ms, c->pool_cached >> 20, c->pool_slabs.size());
        }
        if (e != hipSuccess) {  // out of memory: give back what is parked, then ask for a single block
            (void)hipGetLastError();
            pool_release(c);
            n = 1;
            e = hipMalloc(&base, rounded);
        }
        if (e != hipSuccess) return hipfail(e, "hipMalloc");
        if (next < 128) next *= 2;
        const int si = (int)c->pool_slabs.size();
        c->pool_slabs.push_back(hefx_context::PoolSlab{base, rounded, n, n});
        for (int b = n - 1; b >= 0; --b) {
            void *p = static_cast<char *>(base) + (size_t)b * rounded;
            c->pool_block[p] = hefx_context::PoolBlock{rounded, si};
            fl.push_back(p);
        }
        c->pool_cached += (size_t)n * rounded;
    }
    *d_ptr = fl.back();
    fl.pop_back();
    c->pool_cached -= rounded;
    --c->pool_slabs[c->pool_block[*d_ptr].slab].parked;
    return HEFX_OK;
}
extern "C" int hefx_free(hefx_context *c, void *d_ptr)
{
    CTXCHK(c);
    if (!d_ptr) return HEFX_OK;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->pool_block.find(d_ptr);
        if (it != c->pool_block.end()) {
            c->pool_free[it->second.size].push_back(d_ptr);
            c->pool_cached += it->second.size;
            ++c->pool_slabs[it->second.slab].parked;
            if (c->pool_cached > c->pool_cap) pool_release(c);
            return HEFX_OK;
        }
    }
    HIPCHK(hipFree(d_ptr));  // not one of ours (or the pool is off)
    return HEFX_OK;
}
extern "C" int hefx_upload(hefx_context *c, void *d_dst, const void *h_src, size_t bytes, void *stream)
{
    CTXCHK(c);
    HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));  // the host buffer may be pageable / transient
    return HEFX_OK;
}
extern "C" int hefx_download(hefx_context *c, void *h_dst, const void *d_src, size_t bytes, void *stream)
{
    CTXCHK(c);
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_copy(hefx_context *c, void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    CTXCHK(c);
    HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_copy_peer(hefx_context *dst, void *d_dst, hefx_context *src, const void *d_src, size_t bytes,
                              void *stream)
{
    if (!dst) return fail(HEFX_ERR_INVALID, "null destination context");
    CTXCHK(src);  // the copy is submitted on the source device
    if (!d_dst || !d_src) return fail(HEFX_ERR_INVALID, "null pointer");
    if (dst->device == src->device)
        HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else
        HIPCHK(hipMemcpyPeerAsync(d_dst, dst->device, d_src, src->device, bytes, (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_copy_peer_to(hefx_context *dst, void *d_dst, hefx_context *src, const void *d_src, size_t bytes,
                                 void *dst_stream)
{
    if (!src) return fail(HEFX_ERR_INVALID, "null source context");
    CTXCHK(dst);  // the copy is submitted on the destination device
    if (!d_dst || !d_src) return fail(HEFX_ERR_INVALID, "null pointer");
    if (dst->device == src->device)
        HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)dst_stream));
    else
        HIPCHK(hipMemcpyPeerAsync(d_dst, dst->device, d_src, src->device, bytes, (hipStream_t)dst_stream));
    return HEFX_OK;
}
extern "C" int hefx_context_device(const hefx_context *c) { return c ? c->device : -1; }
extern "C" int hefx_device_memory(hefx_context *c, size_t *free_bytes, size_t *total_bytes)
{
    CTXCHK(c);
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return HEFX_OK;
}
extern "C" int hefx_ks_fallback_count(hefx_context *c, uint64_t *chunks)
{
    CTXCHK(c);
    if (!chunks) return fail(HEFX_ERR_INVALID, "null pointer");
    uint32_t v = 0;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(&v, c->d_gate + KS_RING, sizeof(v), hipMemcpyDeviceToHost));
    *chunks = v;
    return HEFX_OK;
}
extern "C" int hefx_ks_stats(hefx_context *c, uint64_t out[4])
{
    CTXCHK(c);
    if (!out) return fail(HEFX_ERR_INVALID, "null pointer");
    out[0] = c->stat_ks_items, out[1] = c->stat_ks_hoisted, out[2] = c->stat_ks_chunks, out[3] = c->stat_ks_calls;
    return HEFX_OK;
}
extern "C" int hefx_memset_zero(hefx_context *c, void *d_dst, size_t bytes, void *stream)
{
    CTXCHK(c);
    HIPCHK(hipMemsetAsync(d_dst, 0, bytes, (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_stream_sync(hefx_context *c, void *stream)
{
    CTXCHK(c);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return HEFX_OK;
}

// ---------------------------------------------------------------------------------------------
// argument checks
// ---------------------------------------------------------------------------------------------
static int check_level(const hefx_context *c, int L)
{
    // data levels use q_0..q_{L-1}; the special prime (index k-1) is never a data prime unless k == 1
    if (L < 1 || L > c->k) return fail(HEFX_ERR_INVALID, "level L out of range");
    return HEFX_OK;
}
static int check_ks_level(const hefx_context *c, int L)
{
    if (c->logn < 11) return fail(HEFX_ERR_UNSUPPORTED, "key switching / rescale need poly_degree >= 2048");
    if (c->k < 2) return fail(HEFX_ERR_INVALID, "key switching needs a special prime (k >= 2)");
    if (L < 1 || L > c->k - 1) return fail(HEFX_ERR_INVALID, "key-switch level L must be in [1, k-1]");
    return HEFX_OK;
}

// ---------------------------------------------------------------------------------------------
// NTT
// ---------------------------------------------------------------------------------------------
static int ntt_common(hefx_context *c, bool inv, uint64_t *d, int npoly, int nrows, int mod_first, void *stream)
{
    CTXCHK(c);
    if (!d || npoly < 1 || nrows < 1 || mod_first < 0 || mod_first + nrows > c->k)
        return fail(HEFX_ERR_INVALID, "bad NTT arguments");
    if (c->logn == 15) {  // split kernels are out of place: transform into scratch, copy back
        const size_t words = (size_t)c->n * npoly * nrows;
        if (int rc = ensure_scratch(c, words)) return rc;
        HIPCHK(launch_ntt_split15(c->T, inv, (const u64 *)d, c->scratch, npoly, nrows, mod_first, (hipStream_t)stream));
        HIPCHK(hipMemcpyAsync(d, c->scratch, words * sizeof(u64), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return HEFX_OK;
    }
    HIPCHK(launch_ntt(c->T, inv, (u64 *)d, npoly, nrows, mod_first, (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_ntt_forward(hefx_context *c, uint64_t *d, int npoly, int nrows, int mod_first, void *stream)
{
    return ntt_common(c, false, d, npoly, nrows, mod_first, stream);
}
extern "C" int hefx_ntt_inverse(hefx_context *c, uint64_t *d, int npoly, int nrows, int mod_first, void *stream)
{
    return ntt_common(c, true, d, npoly, nrows, mod_first, stream);
}

// ---------------------------------------------------------------------------------------------
// element-wise
// ---------------------------------------------------------------------------------------------
static int ew_common(hefx_context *c, EwOp op, int L, int size, int count, const uint64_t *a, const uint64_t *b,
                     uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (size < 1 || count < 1 || !a || !out) return fail(HEFX_ERR_INVALID, "bad element-wise arguments");
    if ((op == EW_ADD || op == EW_SUB || op == EW_MULPLAIN || op == EW_ADDPLAIN) && !b)
        return fail(HEFX_ERR_INVALID, "missing second operand");
    HIPCHK(launch_elementwise(c->T, op, L, size, count, (const u64 *)a, (const u64 *)b, (u64 *)out, c->d_flag,
                              (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_add(hefx_context *c, int L, int size, int count, const uint64_t *a, const uint64_t *b,
                        uint64_t *out, void *stream)
{
    return ew_common(c, EW_ADD, L, size, count, a, b, out, stream);
}
extern "C" int hefx_sub(hefx_context *c, int L, int size, int count, const uint64_t *a, const uint64_t *b,
                        uint64_t *out, void *stream)
{
    return ew_common(c, EW_SUB, L, size, count, a, b, out, stream);
}
extern "C" int hefx_negate(hefx_context *c, int L, int size, int count, const uint64_t *a, uint64_t *out,
                           void *stream)
{
    return ew_common(c, EW_NEG, L, size, count, a, nullptr, out, stream);
}
extern "C" int hefx_add_plain(hefx_context *c, int L, int size, const uint64_t *ct, const uint64_t *pt,
                              uint64_t *out, void *stream)
{
    return ew_common(c, EW_ADDPLAIN, L, size, 1, ct, pt, out, stream);
}
extern "C" int hefx_reduce_canonical(hefx_context *c, int L, int size, uint64_t *d, int addends, void *stream)
{
    (void)addends;  // barrett64 handles any 64-bit word; callers guarantee the sum did not wrap
    return ew_common(c, EW_REDUCE, L, size, 1, d, nullptr, d, stream);
}

// flag[1 + i] == 0 after a multiply_plain over `count` ciphertexts means nothing beyond c0 of ciphertext i was
// non-zero: count it in flag[0]; the per-ciphertext marks are cleared for the next call
__global__ void transparent_finalize_kernel(int *flag, int count)
{
    int zeros = 0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        if (flag[1 + i] == 0) ++zeros;
        flag[1 + i] = 0;
    }
    if (zeros) atomicAdd(&flag[0], zeros);
}

extern "C" int hefx_multiply_plain(hefx_context *c, int L, int size, int count, const uint64_t *ct,
                                   const uint64_t *pt, uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (count >= 1 && size > 1 && c->flag_cap < 1 + count) {  // [0] transparent count, [1..] one mark per ciphertext
        int *nf = nullptr;
        HIPCHK(hipMalloc((void **)&nf, sizeof(int) * (size_t)(1 + count)));
        HIPCHK(hipMemsetAsync(nf, 0, sizeof(int) * (size_t)(1 + count), (hipStream_t)stream));
        HIPCHK(hipMemcpyAsync(nf, c->d_flag, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        HIPCHK(hipFree(c->d_flag));
        c->d_flag = nf;
        c->flag_cap = 1 + count;
    }
    int rc = ew_common(c, EW_MULPLAIN, L, size, count, ct, pt, out, stream);
    if (rc) return rc;
    if (size > 1) {
        hipLaunchKernelGGL(transparent_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, c->d_flag, count);
        HIPCHK(hipGetLastError());
    }
    return HEFX_OK;
}

extern "C" int hefx_check_transparent(hefx_context *c, void *stream)
{
    CTXCHK(c);
    int h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (h[0]) {
        HIPCHK(hipMemsetAsync(c->d_flag, 0, sizeof(int), (hipStream_t)stream));
        return fail(HEFX_ERR_TRANSPARENT, "result ciphertext is transparent");
    }
    return HEFX_OK;
}

// pt0 (engine-internal; only for n <= 2 * ADD_MANY_GROUP, the sums that skip the table level): in[0] enters the sum
// multiplied by that plaintext
static int add_many_impl(hefx_context *c, int L, int size, int n, const uint64_t *const *in, const uint64_t *pt0, uint64_t *out,
                         void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (n < 1 || size < 1 || !in || !out) return fail(HEFX_ERR_INVALID, "bad add_many arguments");
    if (pt0 && n > 2 * ADD_MANY_GROUP) return fail(HEFX_ERR_INVALID, "internal: a fused first product needs a direct sum");
    // wide sums: one launch reduces groups of 16 through a device pointer table (a ring slot of the key-switch
    // descriptors doubles as the table), then the partials are summed below
    constexpr int TABLE_GROUP = 16;
    constexpr int TABLE_MAX = (int)(sizeof(KsItem) * KS_MAX_CHUNK / sizeof(void *));
    std::vector<const uint64_t *> partial_ptrs;
    if (n > 2 * ADD_MANY_GROUP && n <= TABLE_MAX) {
        for (int i = 0; i < n; ++i)
            if (!in[i]) return fail(HEFX_ERR_INVALID, "null ciphertext in add_many");
        const int groups = (n + TABLE_GROUP - 1) / TABLE_GROUP;
        const size_t words = (size_t)size * L * c->n;
        if (int rc = ensure_scratch(c, words * groups)) return rc;
        const unsigned slot = c->ring_next++ % KS_RING;
        if (c->ring_busy[slot]) HIPCHK(hipEventSynchronize(c->ring_ev[slot]));
        const uint64_t **hp = reinterpret_cast<const uint64_t **>(c->h_items + (size_t)slot * KS_MAX_CHUNK);
        const u64 *const *dp = reinterpret_cast<const u64 *const *>(c->d_items + (size_t)slot * KS_MAX_CHUNK);
        for (int i = 0; i < n; ++i) hp[i] = in[i];
        hipStream_t s = (hipStream_t)stream;
        HIPCHK(hipMemcpyAsync((void *)dp, hp, sizeof(void *) * n, hipMemcpyHostToDevice, s));
        HIPCHK(launch_add_many_table(c->T, L, size, dp, n, TABLE_GROUP, c->scratch, s));
        HIPCHK(hipEventRecord(c->ring_ev[slot], s));
        c->ring_busy[slot] = true;
        partial_ptrs.resize(groups);
        for (int g = 0; g < groups; ++g) partial_ptrs[g] = reinterpret_cast<const uint64_t *>(c->scratch + (size_t)g * words);
        in = partial_ptrs.data();
        n = groups;
    }
    for (int base = 0; base < n; base += ADD_MANY_GROUP) {
        PtrGroup g{};
        const int cnt = (n - base < ADD_MANY_GROUP) ? n - base : ADD_MANY_GROUP;
        for (int i = 0; i < cnt; ++i) {
            if (!in[base + i]) return fail(HEFX_ERR_INVALID, "null ciphertext in add_many");
            g.p[i] = (const u64 *)in[base + i];
        }
        HIPCHK(launch_add_many(c->T, L, size, g, cnt, base > 0, (u64 *)out, (hipStream_t)stream, base == 0 ? (const u64 *)pt0 : nullptr));
    }
    return HEFX_OK;
}
extern "C" int hefx_add_many(hefx_context *c, int L, int size, int n, const uint64_t *const *in, uint64_t *out,
                             void *stream)
{
    return add_many_impl(c, L, size, n, in, nullptr, out, stream);
}

extern "C" int hefx_multiply_plain_sum(hefx_context *c, int L, int size, int n, int group,
                                       const uint64_t *const *cts, const uint64_t *const *pts, uint64_t *const *outs,
                                       void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (n < 1 || size < 1 || group < 1 || !cts || !pts || !outs)
        return fail(HEFX_ERR_INVALID, "bad multiply_plain_sum arguments");
    if (group > n) group = n;
    const int groups = (n + group - 1) / group;
    for (int i = 0; i < n; ++i)
        if (!cts[i] || !pts[i]) return fail(HEFX_ERR_INVALID, "null operand in multiply_plain_sum");
    for (int g = 0; g < groups; ++g) {
        if (!outs[g]) return fail(HEFX_ERR_INVALID, "null output in multiply_plain_sum");
        for (int i = g * group; i < n && i < (g + 1) * group; ++i)
            if (cts[i] == outs[g]) return fail(HEFX_ERR_INVALID, "multiply_plain_sum output aliases an input");
    }
    // the pointer table travels through a ring slot of the key-switch descriptors; whole groups per slice
    constexpr int TABLE_MAX = (int)(sizeof(KsItem) * KS_MAX_CHUNK / sizeof(void *));
    if (2 * group + 1 > TABLE_MAX) {  // a group longer than one table slice: partial sums in scratch, then add_many
        const int part = (TABLE_MAX - 1) / 2;
        const size_t words = (size_t)size * L * c->n;
        for (int g = 0; g < groups; ++g) {
            const int i0 = g * group, len = (n - i0 < group) ? n - i0 : group, nparts = (len + part - 1) / part;
            if (int rc = ensure_scratch(c, words * nparts)) return rc;
            std::vector<uint64_t *> pp(nparts);
            for (int t = 0; t < nparts; ++t) pp[t] = reinterpret_cast<uint64_t *>(c->scratch + (size_t)t * words);
            if (int rc = hefx_multiply_plain_sum(c, L, size, len, part, cts + i0, pts + i0, pp.data(), stream)) return rc;
            if (int rc = hefx_add_many(c, L, size, nparts, pp.data(), outs[g], stream)) return rc;
        }
        return HEFX_OK;
    }
    const int gps = TABLE_MAX / (2 * group + 1);  // groups per slice
    hipStream_t s = (hipStream_t)stream;
    for (int g0 = 0; g0 < groups; g0 += gps) {
        const int ng = groups - g0 < gps ? groups - g0 : gps;
        const int i0 = g0 * group, cnt = (n - i0 < ng * group) ? n - i0 : ng * group;
        const unsigned slot = c->ring_next++ % KS_RING;
        if (c->ring_busy[slot]) HIPCHK(hipEventSynchronize(c->ring_ev[slot]));
        const uint64_t **hp = reinterpret_cast<const uint64_t **>(c->h_items + (size_t)slot * KS_MAX_CHUNK);
        const u64 *const *dp = reinterpret_cast<const u64 *const *>(c->d_items + (size_t)slot * KS_MAX_CHUNK);
        for (int i = 0; i < cnt; ++i) {
            hp[i] = cts[i0 + i];
            hp[cnt + i] = pts[i0 + i];
        }
        for (int g = 0; g < ng; ++g) hp[2 * cnt + g] = outs[g0 + g];
        HIPCHK(hipMemcpyAsync((void *)dp, hp, sizeof(void *) * (2 * (size_t)cnt + ng), hipMemcpyHostToDevice, s));
        HIPCHK(launch_mulplain_sum(c->T, L, size, dp, cnt, group, s));
        HIPCHK(hipEventRecord(c->ring_ev[slot], s));
        c->ring_busy[slot] = true;
    }
    return HEFX_OK;
}

extern "C" int hefx_multiply(hefx_context *c, int L, const uint64_t *a, const uint64_t *b, uint64_t *out3,
                             void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (!a || !b || !out3) return fail(HEFX_ERR_INVALID, "null operand");
    HIPCHK(launch_multiply(c->T, L, (const u64 *)a, (const u64 *)b, (u64 *)out3, (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_multiply_batch(hefx_context *c, int L, int n, const uint64_t *const *a, const uint64_t *const *b,
                                   uint64_t *const *out3, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (n < 1 || !a || !b || !out3) return fail(HEFX_ERR_INVALID, "bad multiply batch arguments");
    for (int i = 0; i < n; ++i)
        if (!a[i] || !b[i] || !out3[i] || out3[i] == a[i] || out3[i] == b[i])
            return fail(HEFX_ERR_INVALID, "null or aliasing operand in multiply batch");
    constexpr int SLICE = (int)(sizeof(KsItem) * KS_MAX_CHUNK / sizeof(void *)) / 3;  // pointer-table ring slot
    hipStream_t s = (hipStream_t)stream;
    for (int i0 = 0; i0 < n; i0 += SLICE) {
        const int cnt = n - i0 < SLICE ? n - i0 : SLICE;
        const unsigned slot = c->ring_next++ % KS_RING;
        if (c->ring_busy[slot]) HIPCHK(hipEventSynchronize(c->ring_ev[slot]));
        const uint64_t **hp = reinterpret_cast<const uint64_t **>(c->h_items + (size_t)slot * KS_MAX_CHUNK);
        const u64 *const *dp = reinterpret_cast<const u64 *const *>(c->d_items + (size_t)slot * KS_MAX_CHUNK);
        for (int i = 0; i < cnt; ++i) {
            hp[i] = a[i0 + i];
            hp[cnt + i] = b[i0 + i];
            hp[2 * cnt + i] = out3[i0 + i];
        }
        HIPCHK(hipMemcpyAsync((void *)dp, hp, sizeof(void *) * 3 * (size_t)cnt, hipMemcpyHostToDevice, s));
        HIPCHK(launch_multiply_table(c->T, L, dp, cnt, s));
        HIPCHK(hipEventRecord(c->ring_ev[slot], s));
        c->ring_busy[slot] = true;
    }
    return HEFX_OK;
}
extern "C" int hefx_square(hefx_context *c, int L, const uint64_t *a, uint64_t *out3, void *stream)
{
    return hefx_multiply(c, L, a, a, out3, stream);
}

// ---------------------------------------------------------------------------------------------
// Galois tables: out[i] = in[ bitrev(((elt*(2*bitrev(i)+1)) mod 2N - 1)/2) ]
// ---------------------------------------------------------------------------------------------
static int get_perm(hefx_context *c, uint32_t elt, const uint32_t **out)
{
    if (!(elt & 1) || elt >= 2 * c->n) return fail(HEFX_ERR_INVALID, "Galois element must be odd and < 2N");
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->perm.find(elt);
    if (it != c->perm.end()) {
        *out = it->second;
        return HEFX_OK;
    }
    std::vector<uint32_t> tab(c->n);
    const uint32_t mask = 2 * c->n - 1;
    for (uint32_t i = 0; i < c->n; ++i) {
        const uint32_t raw = (uint32_t)(((uint64_t)elt * (2 * h_bitrev(i, c->logn) + 1)) & mask);
        tab[i] = h_bitrev((raw - 1) >> 1, c->logn);
    }
    uint32_t *d = nullptr;
    HIPCHK(hipMalloc((void **)&d, sizeof(uint32_t) * c->n));
    hipError_t e = hipMemcpy(d, tab.data(), sizeof(uint32_t) * c->n, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return hipfail(e, "Galois table upload");
    }
    c->perm[elt] = d;
    *out = d;
    return HEFX_OK;
}

static uint32_t inv_mod_2n(uint32_t elt, uint32_t n)
{
    uint32_t x = elt;  // elt odd: x = elt is its inverse mod 8; Newton doubles the correct bits
    for (int r = 0; r < 5; ++r) x *= 2u - elt * x;
    return x & (2u * n - 1u);
}
// exact hoisting's per-element tables (ks_mac_exact_kernel: W_g[m] = NTT_m(F_g)), for every element of `elts` that has none
// yet: one allocation, one mask launch, one transform launch for all of them, then a wait -- a cached table may be read
// from any stream later.  First use only; a linear transform's d tables cost about what its d gather tables do.
// out[i] = the table of elts[i], resolved under the context's lock (the map may grow under another thread's call).
static int ensure_flipw(hefx_context *c, const uint32_t *elts, int n, hipStream_t s, std::vector<const u64 *> &out)
{
    out.assign((size_t)n, nullptr);
    std::vector<uint32_t> miss;
    auto resolve = [&] {  // caller holds c->mu
        miss.clear();
        for (int i = 0; i < n; ++i) {
            auto it = c->flipw.find(elts[i] ? elts[i] : 1u);
            if (it != c->flipw.end())
                out[(size_t)i] = it->second;
            else
                miss.push_back(elts[i] ? elts[i] : 1u);
        }
    };
    {
        std::lock_guard<std::mutex> lk(c->mu);
        resolve();
    }
    if (miss.empty()) return HEFX_OK;
    std::sort(miss.begin(), miss.end());
    miss.erase(std::unique(miss.begin(), miss.end()), miss.end());
    const size_t per = (size_t)c->k * c->n;
    if (c->flipw_bytes + per * miss.size() * sizeof(u64) > c->flipw_cap) {  // over the budget: this batch runs unhoisted (same bits)
        out.clear();
        return HEFX_OK;
    }
    std::vector<uint32_t> ginv(miss.size());
    for (size_t i = 0; i < miss.size(); ++i) ginv[i] = inv_mod_2n(miss[i], c->n);
    u64 *rows = nullptr;
    uint32_t *d_ginv = nullptr;
    HIPCHK(hipMalloc((void **)&rows, per * miss.size() * sizeof(u64) + ginv.size() * sizeof(uint32_t)));
    d_ginv = reinterpret_cast<uint32_t *>(rows + per * miss.size());
    hipError_t e = hipMemcpy(d_ginv, ginv.data(), ginv.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    int rc = HEFX_OK;
    // (slices: grid.z <= 1024; at N = 32768 the transform is out of place through the context's scratch buffer, which never
    // shrinks -- 64 tables at a time keep that loan at 100 MB instead of GBs)
    const size_t slice = c->logn == 15 ? 64 : 1024;
    for (size_t i0 = 0; i0 < miss.size() && e == hipSuccess && rc == HEFX_OK; i0 += slice) {
        const int cnt = (int)std::min<size_t>(slice, miss.size() - i0);
        e = launch_flip_rows(c->T, d_ginv + i0, cnt, rows + per * i0, s);
        if (e == hipSuccess) rc = ntt_common(c, false, (uint64_t *)(rows + per * i0), cnt, c->k, 0, (void *)s);
    }
    if (e == hipSuccess && rc == HEFX_OK) e = hipStreamSynchronize(s);
    if (e != hipSuccess || rc != HEFX_OK) {
        (void)hipFree(rows);
        return rc != HEFX_OK ? rc : hipfail(e, "flip-mask tables");
    }
    std::lock_guard<std::mutex> lk(c->mu);
    c->flipw_slabs.push_back(rows);
    c->flipw_bytes += per * miss.size() * sizeof(u64);
    for (size_t i = 0; i < miss.size(); ++i) c->flipw.emplace(miss[i], rows + per * i);  // (keeps another thread's earlier table)
    resolve();
    return HEFX_OK;
}

// ---------------------------------------------------------------------------------------------
// key switching
// ---------------------------------------------------------------------------------------------
static size_t ks_words_per_item(const hefx_context *c, int L)
{
    // d: L, acc: 2(L+1), u: 2   (units of N words); x is per sub-chunk (ks_x_words), alias copies only when needed
    return (size_t)c->n * ((size_t)L + 2 * (size_t)(L + 1) + 2);
}
static size_t ks_x_words(const hefx_context *c, int L, int sub) { return (size_t)c->n * sub * L * (L + 1); }

// Workgroup shapes of a SMALL chunk (descriptors in the kernel arguments, one launch sequence; KS_Q_* mask).
// Pair path (round 5, ks_pair_*): four launches with two transform phases instead of five with four -- taken while its
// widest grid (4 L^2 quarter workgroups per item) still gets a CU per workgroup (ks_pair_digits runs at 190 VGPRs: one
// 512-thread workgroup per CU), i.e. for what the latency path is for: lone rotations of a NAF chain, the lockstep chains
// of a few dot products.  Measured over n = 1..32, L = 2..8 (profiles/r05/small_batch_pair_sweep.txt): it wins by 8-11 us up
// to 256 workgroups (n = 16 at L = 2, 4 at L = 4, 1 at L = 8) and loses beyond (n = 24, L = 2: 82.8 against 75.2 us).
// HEFX_PAIR=0/1 forces it off / on, HEFX_PAIR_MAX=<workgroups> moves the bound.
// Otherwise per launch: quarter rows where the quarter grid (4 workgroups per row) still gets a CU per workgroup -- measured
// with clock stamps over n = 1..8, L = 2..8 (profiles/r03/quarter_mask_sweep.txt) and end to end up to n = 32: the
// inverse launches up to 256 / 192 quarter workgroups (256 of the mod-down inverse at n = 32 measured +27 us), the
// digit transforms up to 256 (beyond that 2 split workgroups per row win: 14 us for 256 of them against 19 us for
// 512 quarters), the mod-down finish up to 320.  HEFX_QUARTER=0/1 forces none / all, HEFX_QMASK=<bits> any combination.
static int ks_small_shape(int n, int L)
{
    static const int quarter_force = getenv("HEFX_QUARTER") ? atoi(getenv("HEFX_QUARTER")) : -1;
    static const int qmask_force = getenv("HEFX_QMASK") ? atoi(getenv("HEFX_QMASK")) : -1;
    static const int pair_force = getenv("HEFX_PAIR") ? atoi(getenv("HEFX_PAIR")) : -1;
    static const int pair_max = getenv("HEFX_PAIR_MAX") ? atoi(getenv("HEFX_PAIR_MAX")) : 256;
    if (qmask_force >= 0) return qmask_force & (KS_Q_ALL | KS_Q_PAIR);
    if (pair_force > 0 || (pair_force < 0 && quarter_force < 0 && n * L * L * 4 <= pair_max)) return KS_Q_PAIR;
    if (quarter_force >= 0) return quarter_force ? KS_Q_ALL : 0;
    // (round 6: all 16 masks over n = 6..32, L = 3..8 -- tools/qmask_sweep.py, profiles/r06/qmask_sweep.txt: this rule is within
    // 1-2 us of the best mask at every point; the mod-down inverse on quarter rows up to 256 workgroups, i.e. n = 32, now
    // measures 3 us better than split rows there)
    return (n * L * 4 <= 256 ? KS_Q_INTT : 0) | (n * L * L * 4 <= 256 ? KS_Q_NTT : 0) | (n * 2 * 4 <= 256 ? KS_Q_MDI : 0) |
           (n * 2 * L * 4 <= 320 ? KS_Q_FIN : 0);
}

// Chunks of a batch alternate between two internal streams (each with its own scratch half) so that the
// small tail launches of one chunk (2 workgroups per item in the mod-down INTT) overlap the wide launches of
// the next; the caller's stream is forked before and joined after.
// acc_in / acc_out (both or neither; rotations without a fused plaintext only): acc_out[i] = acc_in[i] + ct_out[i]
// trusted (engine-internal callers whose inputs and outputs are engine-owned workspace, disjoint by construction --
// hefx_linear_transform_plain's node buffers): the O(n log n) byte-range independence check is skipped (40 us of host time
// in front of the first launch of a 511-rotation batch)
static int ks_run(hefx_context *c, int L, int n, bool relin, const uint64_t *const *ct_in, const uint32_t *elts,
                  const uint64_t *const *keys, const uint64_t *single_key, const uint64_t *const *pts,
                  uint64_t *const *ct_out, void *stream, bool hoist = false, const uint64_t *const *acc_in = nullptr,
                  uint64_t *const *acc_out = nullptr, bool trusted = false, size_t scratch_off = 0)
{
    // scratch_off (words; engine-internal, hefx_linear_transform_plain's second lane): this call's scratch starts there, so that
    // it may run on another stream beside a call that uses the front of the buffer.  One chunk only (the chunk pipeline's
    // internal streams and scratch halves are one set per context), and the caller has sized the buffer for both.
    CTXCHK(c);
    static const bool ksdbg = getenv("HEFX_DEBUG") && atoi(getenv("HEFX_DEBUG")) >= 2;  // host time of the call's phases
    const auto ks_t0 = std::chrono::steady_clock::now();
    auto ks_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ks_t0).count(); };
    double t_valid = 0, t_tables = 0;
    if (int rc = check_ks_level(c, L)) return rc;
    if (n < 1 || !ct_in || !ct_out) return fail(HEFX_ERR_INVALID, "bad key-switch batch arguments");
    if (!relin && (!elts || !keys)) return fail(HEFX_ERR_INVALID, "missing Galois elements / keys");
    if (relin && !single_key) return fail(HEFX_ERR_INVALID, "missing relinearization key");
    if ((acc_in != nullptr) != (acc_out != nullptr) || (acc_out && (relin || pts || hoist)))
        return fail(HEFX_ERR_INVALID, "accumulate: rotations only, input and output sums together, no fused plaintext");
    const size_t per = ks_words_per_item(c, L);
    int chunk = c->chunk;
    if (chunk <= 0) {  // auto: about 1 GiB of scratch per in-flight chunk, multiple of 8, at most KS_AUTO_CHUNK
        chunk = (int)(((size_t)1 << 30) / (per * sizeof(u64)));
        // ... and at most 256 items from N = 16384 on (2 workgroups per CU x 256 CUs: 256 items fill the chip in whole rounds
        // in every launch; 384 / 512 measure the same at C3 and C5 and 3 % less at C4), 512 below (N = 8192: three or four
        // 256-thread workgroups per CU, and five launches of ~60 us each per 256 items are mostly ramp and tail: 788 k ->
        // 921 k ops/s at C2 with 512, 904 k / 910 k with 768 / 1024; profiles/r04/chunk_size_by_ring.txt)
        const int cap = c->logn <= 13 ? 2 * KS_AUTO_CHUNK : KS_AUTO_CHUNK;
        chunk = chunk > cap ? cap : (chunk < 16 ? 16 : chunk & ~7);
    }
    const int nchunks = (n + chunk - 1) / chunk;
    const bool two = nchunks > 1 && c->use_streams && !c->profiling;
    int sub = c->sub;
    // HEFX_SUB > 0 restricts the digit x modulus scratch to `sub` items at a time (K2 + MAC per sub-chunk).
    // Measured neutral on MI355X (the scratch traffic is not what binds), so the default is one sub-chunk.
    if (sub <= 0 || sub > chunk) sub = chunk;
    // fused digit-NTT + MAC kernel (no x scratch at all) when enabled and N <= 16384 (longer rows do not fit the
    // 8-coefficient-per-thread workgroup); the launcher takes sub < 0 as "fused"
    // (never for a hoisted batch: the sources' digit x modulus products live in the x scratch the fused kernel does without)
    bool fused = c->fused && c->logn <= 14 && !hoist;
    int fused_code = 0;
    if (fused) {
        int nf = 0;
        for (int j = 0; j < L; ++j) nf += c->is_f64[(size_t)j];
        const bool special_f = c->is_f64[(size_t)c->k - 1];
        nf += special_f;
        fused_code = nf | ((special_f ? 0 : 1) << 8);
        if (nf == 0) fused = false;  // nothing to fuse: every target modulus is integer-policy
    }
    // hoisting asked for explicitly (hefx_*_hoisted): every item rotates the same source, which no item may overwrite.
    // Since the hoisted form is exact (same words as the per-item sequence) the flag no longer selects an algorithm: the
    // rule below picks the hoisted kernels for more than 32 items and the latency path for fewer, where it measures the
    // same or better (baby-step/giant-step transforms with 9..31 baby rotations: 178 / 215 / 430 us against 202 / 237 /
    // 445 at N = 8192, d = 10 / 100 / 1000; profiles/r04/ab_exact_hoisting.txt)
    if (hoist) {
        if (relin) return fail(HEFX_ERR_INVALID, "hoisting applies to rotations only");
        for (int i = 0; i < n; ++i)
            if (ct_in[i] != ct_in[0] || ct_out[i] == ct_in[0])
                return fail(HEFX_ERR_INVALID, "hoisted batch: one shared source, never overwritten");
    }
    // Validate the WHOLE batch and resolve every gather table before anything is submitted: an argument error must
    // leave no chunk in flight, no forked stream unjoined and no ring slot marked busy.
    std::vector<const uint32_t *> perms(relin ? 0 : (size_t)n);
    bool any_alias = false;  // rotate_vector_inplace (helper.h:474): the input is copied to scratch before it is overwritten
    for (int i = 0; i < n; ++i) {
        if (!ct_in[i] || !ct_out[i]) return fail(HEFX_ERR_INVALID, "null ciphertext pointer in batch");
        if (relin) {
            if (ct_in[i] == ct_out[i]) return fail(HEFX_ERR_INVALID, "relinearize input and output must not alias");
        } else {
            if (!keys[i]) return fail(HEFX_ERR_INVALID, "null key pointer in batch");
            if (int rc = get_perm(c, elts[i], &perms[i])) return rc;
            any_alias |= ct_in[i] == ct_out[i];
        }
    }
    // Items run in key-grouped order on several internal streams, so no item may read another item's output: a
    // dependent chain handed over as ONE batch would silently give wrong bits.  Refused here, before anything is
    // submitted, on BYTE RANGES (callers hand out views of one allocation: big.view(...) slices): no two outputs may
    // overlap, and no input or plaintext may overlap another item's output -- nor its own, except the exact in-place
    // rotation c_in == c_out, which the kernels serve from a scratch copy.  O(n log n) on the host.
    if (!trusted && (n > 1 || pts || acc_out || relin)) {  // (a lone relinearisation too: its three input polys against its two output polys)
        const size_t row = (size_t)c->n * sizeof(u64);
        const size_t out_b = 2 * (size_t)L * row, in_b = (relin ? 3 : 2) * (size_t)L * row, pt_b = (size_t)L * row;
        std::vector<std::pair<uintptr_t, int>> outs((size_t)n);
        for (int i = 0; i < n; ++i) outs[(size_t)i] = {(uintptr_t)ct_out[i], i};
        std::sort(outs.begin(), outs.end());
        for (int i = 1; i < n; ++i)
            if (outs[(size_t)i - 1].first + out_b > outs[(size_t)i].first)
                return fail(HEFX_ERR_INVALID, "two items of a key-switch batch write overlapping outputs");
        // the output (if any) whose range meets [p, p + bytes): outputs are disjoint, so at most two candidates
        auto hits = [&](const void *p, size_t bytes, int self, bool inplace_ok) -> bool {
            const uintptr_t a = (uintptr_t)p;
            auto it = std::upper_bound(outs.begin(), outs.end(), std::make_pair(a, n));
            for (int step = 0; step < 2; ++step) {
                if (step == 0) {
                    if (it == outs.begin()) continue;
                    const auto &o = *(it - 1);  // starts at or before a
                    if (o.first + out_b > a && !(inplace_ok && o.second == self && o.first == a)) return true;
                } else if (it != outs.end() && it->first < a + bytes)  // starts inside the range
                    return true;
            }
            return false;
        };
        for (int j = 0; j < n; ++j) {
            if (hits(ct_in[j], in_b, j, !relin))
                return fail(HEFX_ERR_INVALID, "key-switch batch items must be independent: one item's input overlaps an item's output");
            if (pts && pts[j] && hits(pts[j], pt_b, j, false))
                return fail(HEFX_ERR_INVALID, "key-switch batch items must be independent: a plaintext overlaps an item's output");
            if (acc_out && (hits(acc_in[j], out_b, -1, false) || hits(acc_out[j], out_b, -1, false)))
                return fail(HEFX_ERR_INVALID, "accumulate: a sum overlaps a rotation output of the batch");
        }
    }
    if (acc_out) {  // the sums: pairwise disjoint, none overlapping any input; acc_in[i] == acc_out[i] (in place) is fine
        const size_t out_b = 2 * (size_t)L * (size_t)c->n * sizeof(u64);
        std::vector<std::pair<uintptr_t, int>> sums((size_t)n);
        for (int i = 0; i < n; ++i) {
            if (!acc_in[i] || !acc_out[i]) return fail(HEFX_ERR_INVALID, "null accumulator pointer in batch");
            sums[(size_t)i] = {(uintptr_t)acc_out[i], i};
        }
        std::sort(sums.begin(), sums.end());
        for (int i = 1; i < n; ++i)
            if (sums[(size_t)i - 1].first + out_b > sums[(size_t)i].first)
                return fail(HEFX_ERR_INVALID, "accumulate: two items write overlapping sums");
        auto meets = [&](const void *p, int self, bool same_ok) {
            const uintptr_t a = (uintptr_t)p;
            auto it = std::upper_bound(sums.begin(), sums.end(), std::make_pair(a, n));
            if (it != sums.begin()) {
                const auto &o = *(it - 1);
                if (o.first + out_b > a && !(same_ok && o.second == self && o.first == a)) return true;
            }
            return it != sums.end() && it->first < a + out_b;
        };
        for (int j = 0; j < n; ++j)
            if (meets(ct_in[j], j, false) || meets(acc_in[j], j, true))
                return fail(HEFX_ERR_INVALID, "accumulate: an input overlaps another item's sum");
    }
    t_valid = ks_us();
    // The items of a batch are independent, so they are PROCESSED grouped by key (stable order inside a group): items
    // that share a Galois key become neighbours -- the MAC loads the key once for two neighbours, and a chunk touches
    // few keys, which then stay in L2 / Infinity Cache.  A linear transform's rotations arrive step by step, i.e. keys
    // round-robin: without the grouping every neighbour pair differs (16 keys round-robin: 183 k against 214 k ops/s).
    std::vector<int> ord;
    if (!relin && n > 2) {
        bool mixed = false, grouped = true;  // grouped: equal keys are already neighbours (non-decreasing addresses)
        for (int i = 1; i < n; ++i) {
            mixed = mixed || keys[i] != keys[0];
            grouped = grouped && !std::less<const void *>()(keys[i], keys[i - 1]);
        }
        if (mixed && !grouped) {
            ord.resize((size_t)n);
            for (int i = 0; i < n; ++i) ord[(size_t)i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return std::less<const void *>()(keys[a], keys[b]); });
        }
    }
    auto src = [&](int i) { return ord.empty() ? i : ord[(size_t)i]; };
    const int cmax = n < chunk ? n : chunk;
    const int ns = two ? (nchunks < c->nstreams ? nchunks : c->nstreams) : 1;
    // EXACT HOISTING (ks_mac_exact_kernel): when at most a third as many DISTINCT source ciphertexts as items are rotated
    // -- the d-1 rotations of a linear transform rotate ONE (helper.h:252-257) -- every distinct source is decomposed and
    // extended to the key moduli once per chunk and the items run the gathered key MAC with the flip-mask correction:
    // SEAL's bits at (L+1)(L+2) -> 2 + 2L transforms per item (profiles/r04/ab_exact_hoisting.txt).  Not for the
    // small-batch latency path (more launches), in-place rotations (their sources are per-item scratch copies), the
    // fused-transform path, relinearisations.  HEFX_SHARE_SRC=0 switches it off.
    static const bool share_ok = !(getenv("HEFX_SHARE_SRC") && atoi(getenv("HEFX_SHARE_SRC")) == 0);
    // items per distinct source from which a chunk is hoisted: 3 (HEFX_SHARE_RATIO).  The direct-key transform has d - 1 per
    // source; the levels of a NAF forest have 2..20, and there 2 and 3 measure 2-3 % under 4 (3.85 / 3.85-3.96 / 3.98-4.01 ms
    // at C3, d = 512; 2.76-2.79 / 2.70-2.74 / 2.78-2.83 ms at N = 8192, d = 1000)
    static const size_t share_ratio = getenv("HEFX_SHARE_RATIO") ? (size_t)std::max(1, atoi(getenv("HEFX_SHARE_RATIO"))) : 3;
    bool share = share_ok && !relin && !fused && !any_alias && n > ks_small_max() && sub >= cmax;
    bool one_source = share;  // the common case -- a linear transform's rotations of ct_new -- needs no hash set
    for (int i = 1; i < n && one_source; ++i) one_source = ct_in[i] == ct_in[0];
    if (share && !one_source) {
        std::unordered_set<const void *> distinct;
        for (int i = 0; i < n && distinct.size() * share_ratio <= (size_t)n; ++i) distinct.insert((const void *)ct_in[i]);
        share = distinct.size() * share_ratio <= (size_t)n;
    }
    std::vector<const u64 *> flips;
    if (share) {
        if (int rc = ensure_flipw(c, elts, n, (hipStream_t)stream, flips)) return rc;
        if (flips.empty()) share = false;  // the tables' memory budget is spent
    }
    // (a hoisted chunk's fallback runs the digit transforms of all its items at once: x for the whole chunk)
    const size_t x_words = ks_x_words(c, L, fused || share || sub > cmax ? cmax : sub);
    const size_t half_words = per * (size_t)cmax + x_words + (any_alias ? (size_t)cmax * 2 * L * c->n : 0);
    t_tables = ks_us();
    if (scratch_off && (two || c->scratch_words < scratch_off + half_words))
        return fail(HEFX_ERR_INVALID, "internal: a laned key-switch batch must be one chunk inside the pre-sized scratch");
    if (int rc = ensure_scratch(c, scratch_off + half_words * (size_t)ns)) return rc;
    hipStream_t user = (hipStream_t)stream;
    if (two) {
        HIPCHK(hipEventRecord(c->ev_fork, user));
        for (int s = 0; s < ns; ++s) HIPCHK(hipStreamWaitEvent(c->streams[s], c->ev_fork, 0));
    }
    const size_t N = c->n;
    // from here on only HIP runtime failures can occur; they stop the submission, and the internal streams are still
    // joined to the caller's stream below so that whatever was launched stays ordered before the caller's next work
    hipError_t herr = hipSuccess;
    const char *hwhat = "";
#define KS_TRY(expr)                       \
    if (herr == hipSuccess) {              \
        herr = (expr);                     \
        if (herr != hipSuccess) hwhat = #expr; \
    }
    int ci = 0;
    ++c->stat_ks_calls;
    for (int base = 0; base < n && herr == hipSuccess; base += chunk, ++ci) {
        const int cnt = (n - base < chunk) ? n - base : chunk;
        const unsigned slot = c->ring_next++ % KS_RING;
        if (c->ring_busy[slot]) {  // its previous chunk has drained
            KS_TRY(hipEventSynchronize(c->ring_ev[slot]));
            c->ring_busy[slot] = false;
        }
        KsItem *hb = c->h_items + (size_t)slot * KS_MAX_CHUNK, *db = c->d_items + (size_t)slot * KS_MAX_CHUNK;
        KsScratch S{};
        S.d = c->scratch + scratch_off + (two ? (size_t)(ci % ns) * half_words : 0);
        S.acc = S.d + (size_t)cnt * L * N;
        S.u = S.acc + (size_t)cnt * 2 * (L + 1) * N;
        S.x = S.u + (size_t)cnt * 2 * N;
        S.alias = S.x + ks_x_words(c, L, fused || share || sub > cnt ? cnt : sub);
        S.qmod = c->d_qmod;
        S.gate = c->d_gate + slot;
        S.gate_hits = c->d_gate + KS_RING;
        if (++c->gate_seq == 0) c->gate_seq = 1;
        S.gate_tag = c->gate_seq;
        S.gate_mode = 0;
        bool chunk_alias = false;
        int nsrc = 0;  // > 0: this chunk runs exactly hoisted over that many distinct sources
        std::unordered_map<const void *, uint32_t> src_of;
        if (share && cnt > ks_small_max() && one_source && cnt + 1 <= KS_MAX_CHUNK) {  // (room for the source descriptor in the ring slot)
            src_of.emplace((const void *)ct_in[0], 0u);
            nsrc = 1;
        } else if (share && cnt > ks_small_max()) {
            for (int i = 0; i < cnt && src_of.size() * share_ratio <= (size_t)cnt; ++i)
                src_of.emplace((const void *)ct_in[src(base + i)], (uint32_t)src_of.size());
            if (src_of.size() * share_ratio <= (size_t)cnt && cnt + (int)src_of.size() <= KS_MAX_CHUNK) nsrc = (int)src_of.size();
        }
        for (int i = 0; i < cnt; ++i) {
            KsItem &it = hb[i];
            const int j = src(base + i);
            it.c_in = (const u64 *)ct_in[j];
            it.c_out = (u64 *)ct_out[j];
            it.pt = pts ? (const u64 *)pts[j] : nullptr;
            it.key = relin ? (const u64 *)single_key : (const u64 *)keys[j];
            it.perm = relin ? nullptr : perms[j];
            it.elt = relin ? 1u : elts[j];
            it.flags = 0;
            it.acc_in = acc_out ? (const u64 *)acc_in[j] : nullptr;
            it.acc_out = acc_out ? (u64 *)acc_out[j] : nullptr;
            it.dsrc = it.pad_ = 0;
            it.flipw = nullptr;
            if (nsrc) {
                it.dsrc = one_source ? 0u : src_of[(const void *)ct_in[j]];
                it.flipw = flips[(size_t)j];
            }
            if (!relin && it.c_in == it.c_out) {  // in place: the kernels read a scratch copy (ks_alias_copy_kernel)
                it.c_in = S.alias + (size_t)i * 2 * L * N;
                it.flags = KS_ALIASED;
                chunk_alias = true;
            }
        }
        KsProf *prof = nullptr;
        if (c->profiling) {
            const size_t need = (size_t)c->prof.used + 8 + 2 * ((size_t)cnt / sub + 1);
            while (c->prof_events.size() < need && herr == hipSuccess) {
                hipEvent_t e;
                KS_TRY(hipEventCreate(&e));
                if (herr == hipSuccess) c->prof_events.push_back(e);
            }
            if (herr != hipSuccess) break;
            c->prof_stage.resize(c->prof_events.size());
            c->prof.ev = c->prof_events.data();
            c->prof.stage = c->prof_stage.data();
            c->prof.cap = (int)c->prof_events.size();
            prof = &c->prof;
            ++c->prof_chunks;
        }
        hipStream_t cs = two ? c->streams[ci % ns] : user;
        // small chunks carry their descriptors in the first launch's kernel arguments (HEFX_SMALL=0 restores the copy)
        static const bool small_ok = !(getenv("HEFX_SMALL") && atoi(getenv("HEFX_SMALL")) == 0);
        const bool small = small_ok && cnt <= ks_small_max() && !nsrc && !chunk_alias;
        // ... and run on quarter-row workgroups when split-2 workgroups (2 L (L+1) per item in the widest launch) would
        // leave CUs idle: HEFX_QUARTER=0/1 overrides the size test
        // per launch: quarter rows where the quarter grid (4 workgroups per row) still gets a CU per workgroup -- measured
        // with clock stamps over n = 1..8, L = 2..8 (profiles/r03/quarter_mask_sweep.txt) and end to end up to n = 32: the
        // inverse launches up to 256 / 192 quarter workgroups (256 of the mod-down inverse at n = 32 measured +27 us), the
        // digit transforms up to 256 (beyond that 2 split workgroups per row win: 14 us for 256 of them against 19 us for
        // 512 quarters), the mod-down finish up to 320.  HEFX_QUARTER=0/1 forces none / all, HEFX_QMASK=<bits> any
        // combination (KS_Q_*)
        const int quarter = (small && !fused && nchunks == 1) ? ks_small_shape(cnt, L) : 0;
        for (const auto &so : src_of) {  // the distinct sources, behind the items: only c_in and elt are read (noperm)
            if (!nsrc) break;
            KsItem &sd = hb[cnt + (int)so.second];
            sd = KsItem{};
            sd.c_in = (const u64 *)so.first;
            sd.elt = 1u;
        }
        c->stat_ks_items += (uint64_t)cnt, c->stat_ks_hoisted += nsrc ? (uint64_t)cnt : 0, ++c->stat_ks_chunks;
        if (!small) KS_TRY(hipMemcpyAsync(db, hb, sizeof(KsItem) * (cnt + nsrc), hipMemcpyHostToDevice, cs));
        KS_TRY(launch_keyswitch_chunk(c->T, L, cnt, db, relin, S, fused ? -1 - fused_code : sub, chunk_alias, small ? hb : nullptr,
                                      quarter, cs, prof, nsrc));
        if (herr == hipSuccess && hipEventRecord(c->ring_ev[slot], cs) == hipSuccess) c->ring_busy[slot] = true;
    }
#undef KS_TRY
    if (two) {
        for (int s = 0; s < ns; ++s) {
            hipError_t e = hipEventRecord(c->ev_join[s], c->streams[s]);
            if (e == hipSuccess) e = hipStreamWaitEvent(user, c->ev_join[s], 0);
            if (e != hipSuccess && herr == hipSuccess) herr = e, hwhat = "join of the internal streams";
        }
    }
    if (herr != hipSuccess) return hipfail(herr, hwhat);
    if (ksdbg)
        fprintf(stderr, "[hefx] ks_run n=%d L=%d chunks=%d share=%d: checks %.1f us, order/tables %.1f us, scratch+submit %.1f us\n", n, L,
                nchunks, (int)share, t_valid, t_tables - t_valid, ks_us() - t_tables);
    return HEFX_OK;
}

extern "C" int hefx_apply_galois(hefx_context *c, int L, const uint64_t *ct_in, uint32_t elt, const uint64_t *key,
                                 uint64_t *ct_out, void *stream)
{
    return ks_run(c, L, 1, false, &ct_in, &elt, &key, nullptr, nullptr, &ct_out, stream);
}
extern "C" int hefx_apply_galois_batch(hefx_context *c, int L, int n, const uint64_t *const *ct_in,
                                       const uint32_t *elts, const uint64_t *const *keys, uint64_t *const *ct_out,
                                       void *stream)
{
    return ks_run(c, L, n, false, ct_in, elts, keys, nullptr, nullptr, ct_out, stream);
}
extern "C" int hefx_rotate_multiply_plain_batch(hefx_context *c, int L, int n, const uint64_t *const *ct_in,
                                                const uint32_t *elts, const uint64_t *const *keys,
                                                const uint64_t *const *pts, uint64_t *const *ct_out, void *stream)
{
    if (!pts) return fail(HEFX_ERR_INVALID, "missing plaintexts");
    // (a null ENTRY is a plain rotation: hefx.h)
    return ks_run(c, L, n, false, ct_in, elts, keys, nullptr, pts, ct_out, stream);
}
extern "C" int hefx_apply_galois_add_batch(hefx_context *c, int L, int n, const uint64_t *const *ct_in,
                                           const uint32_t *elts, const uint64_t *const *keys,
                                           const uint64_t *const *acc_in, uint64_t *const *acc_out,
                                           uint64_t *const *ct_out, void *stream)
{
    if (!acc_in || !acc_out) return fail(HEFX_ERR_INVALID, "missing accumulators");
    return ks_run(c, L, n, false, ct_in, elts, keys, nullptr, nullptr, ct_out, stream, false, acc_in, acc_out);
}
// ---------------------------------------------------------------------------------------------
// helper.h:472-476 as ONE call:  for (s = 1 .. steps) { rotate_vector_inplace(dup, step, gal_keys); add_inplace(mult, dup); }
// for n ciphertext pairs in lockstep (the eight weight chains of the LR gradient, logistic_regression_ckks.cpp:295-300:
// 2000 levels each).  t_0 = ct_in, t_s = apply_galois(t_(s-1)), a_s = a_(s-1) + t_s; ct_out = t_steps, acc_out = a_steps;
// the inputs are never written.  Every level is the key switch the op-by-op sequence runs (same descriptors, same
// kernels, hence the same bits); what the call removes is the host: the intermediate rotations ping-pong between two
// buffer sets, so the levels are two alternating launch sequences issued from one loop -- no validation, allocation or
// bookkeeping per level (HEFX_CHAIN_GRAPH=1: captured once as a HIP graph of two levels and replayed; measured slower).
// ---------------------------------------------------------------------------------------------
namespace {
struct ChainLevel {
    const uint64_t *const *in;
    uint64_t *const *out;
};
// one level on `s`, descriptors in the first launch's kernel arguments (n <= ks_small_max()); nothing is validated here
hipError_t chain_level(hefx_context *c, int L, int n, const uint64_t *const *in, uint64_t *const *out,
                       const uint64_t *const *acc_in, uint64_t *const *acc_out, const uint32_t *elts,
                       const uint64_t *const *keys, const uint32_t *const *perms, const KsScratch &S,
                       KsItem *db, int quarter, hipStream_t s)
{
    KsItem hb[64];
    for (int i = 0; i < n; ++i) {
        KsItem &it = hb[i];
        it.c_in = (const u64 *)in[i];
        it.c_out = (u64 *)out[i];
        it.pt = nullptr;
        it.key = (const u64 *)keys[i];
        it.perm = perms[i];
        it.elt = elts[i];
        it.flags = 0;
        it.acc_in = (const u64 *)acc_in[i];
        it.acc_out = (u64 *)acc_out[i];
        it.dsrc = it.pad_ = 0;
        it.flipw = nullptr;
    }
    c->stat_ks_items += (uint64_t)n, ++c->stat_ks_chunks;
    return launch_keyswitch_chunk(c->T, L, n, db, false, S, n, false, hb, quarter, s, nullptr);
}
}  // namespace

extern "C" int hefx_rotate_add_chain(hefx_context *c, int L, int n, const uint64_t *const *ct_in, const uint32_t *elts,
                                     const uint64_t *const *keys, const uint64_t *const *acc_in,
                                     uint64_t *const *acc_out, uint64_t *const *ct_out, int steps, void *stream)
{
    CTXCHK(c);
    if (steps < 1) return fail(HEFX_ERR_INVALID, "a rotate-and-add chain needs at least one step");
    if (!acc_in || !acc_out) return fail(HEFX_ERR_INVALID, "missing accumulators");
    if (steps == 1)
        return ks_run(c, L, n, false, ct_in, elts, keys, nullptr, nullptr, ct_out, stream, false, acc_in, acc_out);
    if (int rc = check_ks_level(c, L)) return rc;
    if (n < 1 || !ct_in || !ct_out || !elts || !keys) return fail(HEFX_ERR_INVALID, "bad chain arguments");
    // The levels after the first are launched without ks_run's checks, so what they write is checked HERE, before anything
    // is submitted and whatever n is (the wide path's ks_run calls would refuse the same arguments, only later): every
    // final rotation and every sum non-null and all 2n of them pairwise disjoint in bytes.  (ct_out[i] == ct_in[i] is
    // fine: the inputs are read by level 1 only.)
    {
        const size_t out_b = 2 * (size_t)L * (size_t)c->n * sizeof(u64);
        std::vector<uintptr_t> w;
        w.reserve(2 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            if (!ct_out[i] || !acc_out[i] || !acc_in[i] || !ct_in[i] || !keys[i])
                return fail(HEFX_ERR_INVALID, "null pointer in a rotate-and-add chain");
            w.push_back((uintptr_t)ct_out[i]);
            w.push_back((uintptr_t)acc_out[i]);
        }
        std::sort(w.begin(), w.end());
        for (size_t i = 1; i < w.size(); ++i)
            if (w[i - 1] + out_b > w[i])
                return fail(HEFX_ERR_INVALID, "rotate-and-add chain: final rotations and sums must not overlap one another");
    }
    // the two intermediate buffer sets (pooled; parked again when the call returns -- later users are ordered behind
    // this call's work on the caller's stream, hefx_malloc's contract)
    const size_t ctw = 2 * (size_t)L * c->n;
    void *ws = nullptr;
    if (int rc = hefx_malloc(c, 2 * (size_t)n * ctw * sizeof(u64), &ws)) return rc;
    struct Parked {
        hefx_context *c;
        void *p;
        ~Parked() { (void)hefx_free(c, p); }
    } parked{c, ws};
    std::vector<uint64_t *> A((size_t)n), B((size_t)n);
    for (int i = 0; i < n; ++i) {
        A[(size_t)i] = (uint64_t *)ws + (size_t)i * ctw;
        B[(size_t)i] = (uint64_t *)ws + ((size_t)n + i) * ctw;
    }
    // level 1 through the validating front door: ct_in -> A, acc_in -> acc_out; from then on the sums are in place
    if (int rc = ks_run(c, L, n, false, ct_in, elts, keys, nullptr, nullptr, A.data(), stream, false, acc_in, acc_out)) return rc;
    const int mid = steps - 2;  // levels 2 .. steps-1 alternate A -> B, B -> A; the last level writes ct_out
    const uint64_t *const *lastsrc = (mid & 1) ? B.data() : A.data();
    hipStream_t user = (hipStream_t)stream;
    // HIP-graph replay of the two alternating levels: opt-in (HEFX_CHAIN_GRAPH=1).  Measured on MI355X / ROCm 7.2
    // (tools/chain_latency.py, profiles/r04/chain_latency.txt): 60.3 us per level replayed against 53.9 us with plain
    // launches at n = 8, L = 2 (52.4 / 46.6 at n = 1) -- the graph's kernel nodes are dispatched with larger gaps than
    // back-to-back stream launches, and the host is not the bottleneck once the levels are issued from one C loop.
    static const bool graph_ok = getenv("HEFX_CHAIN_GRAPH") && atoi(getenv("HEFX_CHAIN_GRAPH")) != 0;
    if (n > ks_small_max() || n > 64) {  // wide chains: the regular batched path per level (descriptor ring, chunks)
        for (int sidx = 0; sidx < mid; ++sidx) {
            const bool ab = (sidx & 1) == 0;
            if (int rc = ks_run(c, L, n, false, ab ? A.data() : B.data(), elts, keys, nullptr, nullptr, ab ? B.data() : A.data(),
                                stream, false, acc_out, acc_out))
                return rc;
        }
        return ks_run(c, L, n, false, lastsrc, elts, keys, nullptr, nullptr, ct_out, stream, false, acc_out, acc_out);
    }
    std::vector<const uint32_t *> perms((size_t)n);
    for (int i = 0; i < n; ++i)
        if (int rc = get_perm(c, elts[i], &perms[(size_t)i])) return rc;
    // scratch as ks_run lays it out for one chunk on the caller's stream (level 1 has already sized it)
    const size_t per = ks_words_per_item(c, L);
    if (int rc = ensure_scratch(c, per * (size_t)n + ks_x_words(c, L, n))) return rc;
    const size_t N = c->n;
    KsScratch S{};
    S.d = c->scratch;
    S.acc = S.d + (size_t)n * L * N;
    S.u = S.acc + (size_t)n * 2 * (L + 1) * N;
    S.x = S.u + (size_t)n * 2 * N;
    S.alias = S.x + ks_x_words(c, L, n);
    const int quarter = ks_small_shape(n, L);
    // the kernels after the first read the descriptors from device memory: a slot of their own per direction (and lane), so
    // a replayed level never finds another call's descriptors there
    constexpr int MAX_LANES = 1 + hefx_context::MAX_STREAMS;
    if (!c->chain_items) HIPCHK(hipMalloc((void **)&c->chain_items, sizeof(KsItem) * 2 * 64 * MAX_LANES));
    KsItem *dbA = c->chain_items, *dbB = c->chain_items + 64;
    auto level = [&](bool ab, hipStream_t s) {
        return chain_level(c, L, n, ab ? A.data() : B.data(), ab ? B.data() : A.data(), acc_out, acc_out, elts, keys, perms.data(), S,
                           ab ? dbA : dbB, quarter, s);
    };
    const int pairs = mid / 2;
    hipError_t herr = hipSuccess;
    if (graph_ok && pairs >= 4) {
        // legacy default streams cannot be captured: the chain runs on an internal stream, forked from and joined to the caller's
        hipStream_t cs = c->streams[0];
        HIPCHK(hipEventRecord(c->ev_fork, user));
        HIPCHK(hipStreamWaitEvent(cs, c->ev_fork, 0));
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        herr = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (herr == hipSuccess) {
            hipError_t e1 = level(true, cs);
            hipError_t e2 = e1 == hipSuccess ? level(false, cs) : e1;
            herr = hipStreamEndCapture(cs, &graph);
            if (e2 != hipSuccess) herr = e2;
        }
        if (herr == hipSuccess) herr = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        for (int r = 0; r < pairs && herr == hipSuccess; ++r) herr = hipGraphLaunch(exec, cs);
        if (herr == hipSuccess && (mid & 1)) herr = level(true, cs);
        if (herr == hipSuccess)
            herr = chain_level(c, L, n, lastsrc, ct_out, acc_out, acc_out, elts, keys, perms.data(), S, dbA, quarter, cs);
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        hipError_t ej = hipEventRecord(c->ev_join[0], cs);
        if (ej == hipSuccess) ej = hipStreamWaitEvent(user, c->ev_join[0], 0);
        if (herr == hipSuccess) herr = ej;
    } else {
        // LANES (round 5).  Chains are independent of one another, so many of them can be dealt over the caller's stream and
        // the internal ones (every lane its own slice of the scratch and its own descriptor slots; internal streams forked from
        // and joined to the caller's) -- same launches per chain, same bits.  Measured (profiles/r05/chain_lanes.txt): a level
        // is a chain of DEPENDENT launches, so lanes do not shorten it, and launches on several hardware queues start later
        // than on one (n = 8, L = 2: 43.8 us per level on one lane, 51.4 on three); they pay once a lane still carries a
        // full small batch -- n = 24: 79.5 -> 62.9 us per level.  Hence one lane per 8 chains.  HEFX_CHAIN_LANES=<k> forces k.
        static const int lanes_env = getenv("HEFX_CHAIN_LANES") ? atoi(getenv("HEFX_CHAIN_LANES")) : 0;
        int nl = c->use_streams && !c->profiling ? 1 + c->nstreams : 1;
        if (lanes_env > 0)
            nl = lanes_env < nl ? lanes_env : nl;
        else if (nl > n / 8)
            nl = n / 8 > 0 ? n / 8 : 1;
        if (nl > n) nl = n;
        struct Lane {
            int lo, cnt, quarter;
            KsScratch S;
            KsItem *dbA, *dbB;
            hipStream_t s;
        };
        Lane lane[MAX_LANES];
        size_t off = 0;
        for (int l = 0; l < nl; ++l) {
            Lane &ln = lane[l];
            ln.lo = (int)((long long)n * l / nl);
            ln.cnt = (int)((long long)n * (l + 1) / nl) - ln.lo;
            ln.quarter = ks_small_shape(ln.cnt, L);
            ln.S = KsScratch{};
            ln.S.d = c->scratch + off;
            ln.S.acc = ln.S.d + (size_t)ln.cnt * L * N;
            ln.S.u = ln.S.acc + (size_t)ln.cnt * 2 * (L + 1) * N;
            ln.S.x = ln.S.u + (size_t)ln.cnt * 2 * N;
            ln.S.alias = ln.S.x + ks_x_words(c, L, ln.cnt);
            off += per * (size_t)ln.cnt + ks_x_words(c, L, ln.cnt);
            ln.dbA = c->chain_items + (size_t)l * 128;
            ln.dbB = ln.dbA + 64;
            ln.s = l == 0 ? user : c->streams[l - 1];
        }
        if (nl > 1) {
            herr = hipEventRecord(c->ev_fork, user);
            for (int l = 1; l < nl && herr == hipSuccess; ++l) herr = hipStreamWaitEvent(lane[l].s, c->ev_fork, 0);
        }
        auto lane_level = [&](const Lane &ln, const uint64_t *const *in, uint64_t *const *out, KsItem *db) {
            return chain_level(c, L, ln.cnt, in + ln.lo, out + ln.lo, acc_out + ln.lo, acc_out + ln.lo, elts + ln.lo, keys + ln.lo,
                               perms.data() + ln.lo, ln.S, db, ln.quarter, ln.s);
        };
        // level by level over the lanes, so that every queue is fed from the first microsecond on
        for (int sidx = 0; sidx < mid && herr == hipSuccess; ++sidx) {
            const bool ab = (sidx & 1) == 0;
            for (int l = 0; l < nl && herr == hipSuccess; ++l)
                herr = lane_level(lane[l], ab ? A.data() : B.data(), ab ? B.data() : A.data(), ab ? lane[l].dbA : lane[l].dbB);
        }
        for (int l = 0; l < nl && herr == hipSuccess; ++l) herr = lane_level(lane[l], lastsrc, ct_out, lane[l].dbA);
        for (int l = 1; l < nl; ++l) {  // joined on every path: whatever was launched stays ordered before the caller's next work
            hipError_t ej = hipEventRecord(c->ev_join[l - 1], lane[l].s);
            if (ej == hipSuccess) ej = hipStreamWaitEvent(user, c->ev_join[l - 1], 0);
            if (herr == hipSuccess) herr = ej;
        }
    }
    if (herr != hipSuccess) return hipfail(herr, "rotate-and-add chain");
    return HEFX_OK;
}

extern "C" int hefx_relinearize(hefx_context *c, int L, const uint64_t *ct3, const uint64_t *key, uint64_t *ct2,
                                void *stream)
{
    return ks_run(c, L, 1, true, &ct3, nullptr, nullptr, key, nullptr, &ct2, stream);
}
extern "C" int hefx_relinearize_batch(hefx_context *c, int L, int n, const uint64_t *const *ct3,
                                      const uint64_t *key, uint64_t *const *ct2, void *stream)
{
    return ks_run(c, L, n, true, ct3, nullptr, nullptr, key, nullptr, ct2, stream);
}

// ---------------------------------------------------------------------------------------------
// rescale / mod drop
// ---------------------------------------------------------------------------------------------
static int rescale_common(hefx_context *c, int L, int size, int count, const uint64_t *in, uint64_t *out, int mode,
                          void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (L < 2) return fail(HEFX_ERR_INVALID, "cannot rescale at the last level");
    if (size < 1 || count < 1 || !in || !out) return fail(HEFX_ERR_INVALID, "bad rescale arguments");
    if (in == out) return fail(HEFX_ERR_INVALID, "rescale input and output must not alias");
    if (mode != HEFX_RESCALE_FLOOR && mode != HEFX_RESCALE_ROUND) return fail(HEFX_ERR_INVALID, "bad rescale mode");
    if (c->logn < 11) return fail(HEFX_ERR_UNSUPPORTED, "key switching / rescale need poly_degree >= 2048");
    if (int rc = ensure_scratch(c, (size_t)c->n * size * count)) return rc;
    HIPCHK(launch_rescale(c->T, L, size, count, (const u64 *)in, (u64 *)out, nullptr, c->scratch,
                          mode == HEFX_RESCALE_ROUND, (hipStream_t)stream));
    return HEFX_OK;
}

// Pointer tables of the *_batch entries travel through a ring slot of the key-switch descriptors (pinned host mirror
// -> device copy, one async copy per slice).  fill(hp, i0, cnt) writes the slice's table; run(dp, cnt) launches on it.
template <class Fill, class Run>
static int table_slices(hefx_context *c, int n, int ptrs_per_item, hipStream_t s, Fill fill, Run run, int max_slice = 0)
{
    int SLICE = (int)(sizeof(KsItem) * KS_MAX_CHUNK / sizeof(void *)) / ptrs_per_item;
    if (max_slice > 0 && SLICE > max_slice) SLICE = max_slice;
    for (int i0 = 0; i0 < n; i0 += SLICE) {
        const int cnt = n - i0 < SLICE ? n - i0 : SLICE;
        const unsigned slot = c->ring_next++ % KS_RING;
        if (c->ring_busy[slot]) HIPCHK(hipEventSynchronize(c->ring_ev[slot]));
        const uint64_t **hp = reinterpret_cast<const uint64_t **>(c->h_items + (size_t)slot * KS_MAX_CHUNK);
        const u64 *const *dp = reinterpret_cast<const u64 *const *>(c->d_items + (size_t)slot * KS_MAX_CHUNK);
        fill(hp, i0, cnt);
        HIPCHK(hipMemcpyAsync((void *)dp, hp, sizeof(void *) * (size_t)ptrs_per_item * cnt, hipMemcpyHostToDevice, s));
        HIPCHK(run(dp, i0, cnt));
        HIPCHK(hipEventRecord(c->ring_ev[slot], s));
        c->ring_busy[slot] = true;
    }
    return HEFX_OK;
}

extern "C" int hefx_rescale_to_next_batch(hefx_context *c, int L, int size, int n, const uint64_t *const *in,
                                          uint64_t *const *out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (L < 2) return fail(HEFX_ERR_INVALID, "cannot rescale at the last level");
    if (size < 1 || n < 1 || !in || !out) return fail(HEFX_ERR_INVALID, "bad rescale arguments");
    if (c->logn < 11) return fail(HEFX_ERR_UNSUPPORTED, "key switching / rescale need poly_degree >= 2048");
    for (int i = 0; i < n; ++i)
        if (!in[i] || !out[i] || in[i] == out[i])
            return fail(HEFX_ERR_INVALID, "null or aliasing operand in rescale batch");
    const int SLICE = (int)(sizeof(KsItem) * KS_MAX_CHUNK / sizeof(void *)) / 2;
    if (int rc = ensure_scratch(c, (size_t)c->n * size * (n < SLICE ? n : SLICE))) return rc;
    const bool rounded = c->rescale_mode == HEFX_RESCALE_ROUND;
    return table_slices(
        c, n, 2, (hipStream_t)stream,
        [&](const uint64_t **hp, int i0, int cnt) {
            for (int i = 0; i < cnt; ++i) hp[i] = in[i0 + i], hp[cnt + i] = out[i0 + i];
        },
        [&](const u64 *const *dp, int, int cnt) {
            return launch_rescale(c->T, L, size, cnt, nullptr, nullptr, dp, c->scratch, rounded, (hipStream_t)stream);
        });
}

static int addsub_batch(hefx_context *c, bool sub, int L, int size, int n, const uint64_t *const *a,
                        const uint64_t *const *b, uint64_t *const *out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (n < 1 || size < 1 || !a || !b || !out) return fail(HEFX_ERR_INVALID, "bad element-wise batch arguments");
    for (int i = 0; i < n; ++i)
        if (!a[i] || !b[i] || !out[i]) return fail(HEFX_ERR_INVALID, "null operand in element-wise batch");
    return table_slices(
        c, n, 3, (hipStream_t)stream,
        [&](const uint64_t **hp, int i0, int cnt) {
            for (int i = 0; i < cnt; ++i) hp[i] = a[i0 + i], hp[cnt + i] = b[i0 + i], hp[2 * cnt + i] = out[i0 + i];
        },
        [&](const u64 *const *dp, int, int cnt) {
            return launch_addsub_table(c->T, sub, L, size, dp, cnt, (hipStream_t)stream);
        });
}
extern "C" int hefx_add_batch(hefx_context *c, int L, int size, int n, const uint64_t *const *a,
                              const uint64_t *const *b, uint64_t *const *out, void *stream)
{
    return addsub_batch(c, false, L, size, n, a, b, out, stream);
}
extern "C" int hefx_sub_batch(hefx_context *c, int L, int size, int n, const uint64_t *const *a,
                              const uint64_t *const *b, uint64_t *const *out, void *stream)
{
    return addsub_batch(c, true, L, size, n, a, b, out, stream);
}
extern "C" int hefx_multiply_plain_batch(hefx_context *c, int L, int size, int n, const uint64_t *const *cts,
                                         const uint64_t *const *pts, uint64_t *const *outs, void *stream)
{
    // n one-term sums: the same kernel, the same canonical products (transparency is the caller's check, as there)
    return hefx_multiply_plain_sum(c, L, size, n, 1, cts, pts, outs, stream);
}
extern "C" int hefx_rescale_to_next(hefx_context *c, int L, int size, int count, const uint64_t *in, uint64_t *out,
                                    void *stream)
{
    CTXCHK(c);
    return rescale_common(c, L, size, count, in, out, c->rescale_mode, stream);
}
extern "C" int hefx_rescale_to_next_mode(hefx_context *c, int L, int size, int count, const uint64_t *in,
                                         uint64_t *out, int mode, void *stream)
{
    return rescale_common(c, L, size, count, in, out, mode, stream);
}
extern "C" int hefx_set_rescale_mode(hefx_context *c, int mode)
{
    CTXCHK(c);
    if (mode != HEFX_RESCALE_FLOOR && mode != HEFX_RESCALE_ROUND) return fail(HEFX_ERR_INVALID, "bad rescale mode");
    c->rescale_mode = mode;
    return HEFX_OK;
}
extern "C" int hefx_get_rescale_mode(const hefx_context *c) { return c ? c->rescale_mode : HEFX_ERR_INVALID; }

extern "C" int hefx_mod_drop(hefx_context *c, int L_in, int L_out, int npoly, const uint64_t *in, uint64_t *out,
                             void *stream)
{
    CTXCHK(c);
    if (L_out < 1 || L_out > L_in || L_in > c->k || npoly < 1 || !in || !out)
        return fail(HEFX_ERR_INVALID, "bad mod_drop arguments");
    const size_t row = (size_t)c->n * sizeof(u64);
    HIPCHK(hipMemcpy2DAsync(out, row * L_out, in, row * L_in, row * L_out, npoly, hipMemcpyDeviceToDevice,
                            (hipStream_t)stream));
    return HEFX_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU exchange behind the boundary (SURVEY 8b hefx_allreduce_sum, 8e): RCCL over xGMI, one communicator per
// context (= per rank, one rank per GPU).  librccl is resolved at run time with dlopen -- libhefx.so has no link-time
// dependency on it, single-GPU users never load it, and inside a torch process the SONAME resolves to the copy
// torch.distributed already loaded.  Types and constants below are RCCL's public ABI (rccl.h: ncclUniqueId is 128
// opaque bytes, ncclUint64 = 5, ncclSum = 0, ncclSuccess = 0).
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclId {
    char internal[128];
};
struct RcclApi {
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
RcclApi &rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        void *h = nullptr;
        // The copy the process already uses, if any -- found by walking the loaded objects, because a host such as
        // torch maps its own build by path / RPATH (torch/lib/librccl.so), which a bare-name RTLD_NOLOAD lookup misses;
        // two RCCL builds in one process is what this avoids.
        std::string mapped;
        dl_iterate_phdr(
            [](struct dl_phdr_info *info, size_t, void *out) -> int {
                const char *nm = info->dlpi_name;
                if (nm && strstr(nm, "librccl.so")) {
                    *static_cast<std::string *>(out) = nm;
                    return 1;
                }
                return 0;
            },
            &mapped);
        if (!mapped.empty()) h = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD);
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        }
        // none loaded yet: load one privately (RTLD_LOCAL: its symbols must not interpose on a copy loaded later)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        }
        if (!h) {
            a.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
            return a;
        }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.GetErrorString;
        if (!a.ok) a.why = "librccl lacks the expected symbols";
        return a;
    }();
    return api;
}
int rcclfail(int rc, const char *what)
{
    return fail(HEFX_ERR_HIP, std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(rc) : "RCCL error"));
}
}  // namespace

extern "C" int hefx_comm_unique_id(uint8_t *id128)
{
    if (!id128) return fail(HEFX_ERR_INVALID, "null id buffer");
    if (!rccl().ok) return fail(HEFX_ERR_UNSUPPORTED, rccl().why);
    RcclId id;
    if (int rc = rccl().GetUniqueId(&id)) return rcclfail(rc, "ncclGetUniqueId");
    memcpy(id128, id.internal, sizeof id.internal);
    return HEFX_OK;
}
extern "C" int hefx_comm_init(hefx_context *c, int world, int rank, const uint8_t *id128)
{
    CTXCHK(c);
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(HEFX_ERR_INVALID, "bad communicator arguments");
    if (world > 8) return fail(HEFX_ERR_INVALID, "the wrap-free uint64 sum holds for at most 8 ranks (residues < 2^61)");
    if (c->comm) return fail(HEFX_ERR_INVALID, "the context already has a communicator");
    if (!rccl().ok) return fail(HEFX_ERR_UNSUPPORTED, rccl().why);
    RcclId id;
    memcpy(id.internal, id128, sizeof id.internal);
    void *comm = nullptr;
    if (int rc = rccl().CommInitRank(&comm, world, id, rank)) return rcclfail(rc, "ncclCommInitRank");
    c->comm = comm;
    c->comm_world = world;
    c->comm_rank = rank;
    return HEFX_OK;
}
extern "C" int hefx_comm_destroy(hefx_context *c)
{
    CTXCHK(c);
    if (c->comm) {
        (void)rccl().CommDestroy(c->comm);
        c->comm = nullptr;
        c->comm_world = c->comm_rank = 0;
    }
    return HEFX_OK;
}
extern "C" int hefx_comm_world(const hefx_context *c) { return c ? c->comm_world : 0; }
extern "C" int hefx_comm_rank(const hefx_context *c) { return c ? c->comm_rank : 0; }

// sum of every rank's partial ciphertext, in place, canonical: one all-reduce(SUM, uint64) of size*L*N words -- exact,
// at most 8 addends below 2^61 cannot wrap -- and the local reduction mod q_j.  Modular addition is associative, so
// the bits equal a serial add_many over the ranks' partials (helper.h:259).
extern "C" int hefx_allreduce_sum(hefx_context *c, int L, int size, uint64_t *d_ct, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (size < 1 || !d_ct) return fail(HEFX_ERR_INVALID, "bad all-reduce arguments");
    if (!c->comm) return fail(HEFX_ERR_INVALID, "no communicator: call hefx_comm_init first");
    const size_t words = (size_t)size * L * c->n;
    if (int rc = rccl().AllReduce(d_ct, d_ct, words, /*ncclUint64*/ 5, /*ncclSum*/ 0, c->comm, (hipStream_t)stream))
        return rcclfail(rc, "ncclAllReduce");
    return hefx_reduce_canonical(c, L, size, d_ct, c->comm_world, stream);
}

// ---------------------------------------------------------------------------------------------
// timing helpers: HIP events on the stream the kernels are launched on
// ---------------------------------------------------------------------------------------------
extern "C" int hefx_event_create(hefx_context *c, void **ev)
{
    CTXCHK(c);
    if (!ev) return fail(HEFX_ERR_INVALID, "null out pointer");
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    *ev = (void *)e;
    return HEFX_OK;
}
extern "C" int hefx_event_destroy(hefx_context *c, void *ev)
{
    CTXCHK(c);
    if (ev) HIPCHK(hipEventDestroy((hipEvent_t)ev));
    return HEFX_OK;
}
extern "C" int hefx_event_record(hefx_context *c, void *ev, void *stream)
{
    CTXCHK(c);
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_event_elapsed_ms(hefx_context *c, void *ev_start, void *ev_stop, float *ms)
{
    CTXCHK(c);
    if (!ms) return fail(HEFX_ERR_INVALID, "null out pointer");
    HIPCHK(hipEventSynchronize((hipEvent_t)ev_stop));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return HEFX_OK;
}

extern "C" int hefx_profile_begin(hefx_context *c)
{
    CTXCHK(c);
    HIPCHK(hipDeviceSynchronize());
    c->profiling = true;
    c->prof.used = 0;
    c->prof_chunks = 0;
    return HEFX_OK;
}
extern "C" int hefx_profile_end(hefx_context *c, double *stage_ms, uint64_t *launches)
{
    CTXCHK(c);
    if (!stage_ms || !launches) return fail(HEFX_ERR_INVALID, "null out pointer");
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < KS_STAGES; ++k) stage_ms[k] = 0.0;
    *launches = c->prof_chunks;
    for (int i = 0; i + 1 < c->prof.used; ++i) {
        const int st = c->prof_stage[i];
        if (st < 0 || st >= KS_STAGES) continue;
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_events[i], c->prof_events[i + 1]));
        stage_ms[st] += ms;
    }
    c->profiling = false;
    c->prof.used = 0;
    return HEFX_OK;
}
extern "C" const char *hefx_profile_stage_name(int k)
{
    static const char *names[KS_STAGES] = {"ks_alias_copy_kernel",   "ks_intt_digits_kernel",
                                           "ks_ntt_digits_kernel",   "ks_mac_kernel",
                                           "ks_moddown_intt_kernel", "ks_moddown_finish_kernel",
                                           "ks_ntt_mac_kernel",      "gated fallback launches"};
    static_assert(KS_STAGES == HEFX_PROFILE_STAGES, "profile stage count");
    return (k >= 0 && k < KS_STAGES) ? names[k] : "";
}

// ---------------------------------------------------------------------------------------------
// CKKS encode on the GPU
// ---------------------------------------------------------------------------------------------
static int ensure_encode_tables(hefx_context *c)
{
    if (c->d_enc_tables) return HEFX_OK;
    const size_t n = c->n, half = n / 2;
    const size_t slot_b = sizeof(int) * half, w_b = sizeof(double2) * (n / 4), pre_b = sizeof(double2) * half,
                 post_b = sizeof(double2) * n;
    const size_t slot_pad = (slot_b + 15) & ~(size_t)15;
    std::vector<unsigned char> host(slot_pad + w_b + pre_b + post_b);
    int *slot = reinterpret_cast<int *>(host.data());
    double2 *w = reinterpret_cast<double2 *>(host.data() + slot_pad);
    double2 *pre = w + n / 4;
    double2 *post = pre + half;
    for (size_t r = 0; r < half; ++r) slot[r] = -1;
    u64 pos = 1;
    for (size_t i = 0; i < half; ++i) {
        const size_t r1 = (size_t)((pos - 1) >> 1);
        if (r1 < half)
            slot[r1] = (int)(i << 1);
        else
            slot[n - 1 - r1] = (int)((i << 1) | 1);
        pos = (pos * 3) & (2 * n - 1);
    }
    const double pi = 3.14159265358979323846264338327950288;
    for (size_t m = 0; m < n / 4; ++m) w[m] = make_double2(cos(-2.0 * pi * (double)m / (double)half), sin(-2.0 * pi * (double)m / (double)half));
    for (size_t r = 0; r < half; ++r) pre[r] = make_double2(cos(-2.0 * pi * (double)r / (double)n), sin(-2.0 * pi * (double)r / (double)n));
    for (size_t k = 0; k < n; ++k) post[k] = make_double2(cos(-pi * (double)k / (double)n), sin(-pi * (double)k / (double)n));
    HIPCHK(hipMalloc(&c->d_enc_tables, host.size()));
    HIPCHK(hipMemcpy(c->d_enc_tables, host.data(), host.size(), hipMemcpyHostToDevice));
    unsigned char *b = static_cast<unsigned char *>(c->d_enc_tables);
    c->E.slot = reinterpret_cast<const int *>(b);
    c->E.wfft = reinterpret_cast<const double2 *>(b + slot_pad);
    c->E.pre = c->E.wfft + n / 4;
    c->E.post = c->E.pre + half;
    return HEFX_OK;
}

extern "C" int hefx_ckks_encode(hefx_context *c, int L, const double *h_re, const double *h_im, int nvalues,
                                int count, double scale, uint64_t *d_out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (c->logn < 10 || c->logn > 15) return fail(HEFX_ERR_UNSUPPORTED, "GPU encode is built for poly_degree in [1024, 32768]");
    if (!h_re || !d_out || count < 1 || nvalues < 1 || (size_t)nvalues > c->n / 2)
        return fail(HEFX_ERR_INVALID, "values has invalid size");
    if (!(scale > 0)) return fail(HEFX_ERR_INVALID, "scale out of bounds");
    if (int rc = ensure_encode_tables(c)) return rc;
    const size_t nv = (size_t)nvalues * count, need = nv * (h_im ? 2 : 1);
    if (c->vals_cap < need) {
        if (c->d_vals) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipFree(c->d_vals));
        }
        HIPCHK(hipMalloc((void **)&c->d_vals, need * sizeof(double)));
        c->vals_cap = need;
    }
    hipStream_t s = (hipStream_t)stream;
    const size_t slot_doubles = c->n;  // one vector of N/2 complex values
    if (need <= slot_doubles) {  // small: through the pinned ring, no wait for the copy
        if (!c->h_stage) {
            HIPCHK(hipHostMalloc((void **)&c->h_stage, sizeof(double) * slot_doubles * hefx_context::STAGE_SLOTS,
                                 hipHostMallocDefault));
            for (auto &ev : c->stage_ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        const unsigned slot = c->stage_next++ % hefx_context::STAGE_SLOTS;
        if (c->stage_busy[slot]) HIPCHK(hipEventSynchronize(c->stage_ev[slot]));
        double *h = c->h_stage + (size_t)slot * slot_doubles;
        memcpy(h, h_re, nv * sizeof(double));
        if (h_im) memcpy(h + nv, h_im, nv * sizeof(double));
        HIPCHK(hipMemcpyAsync(c->d_vals, h, need * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(hipEventRecord(c->stage_ev[slot], s));
        c->stage_busy[slot] = true;
    } else {
        // large: through one of two pinned buffers sized for the call (grown on demand), so that the call neither copies
        // from pageable memory (1-3 GB/s through the runtime's bounce buffers) nor waits for the STREAM -- until round 5
        // it ended in hipStreamSynchronize because the host arrays may be transient, i.e. every batch of encodes waited
        // for all the work queued before it: 30 ms per 546 one-hot masks inside the reference's prediction loop
        // (logistic_regression_ckks.cpp:222-225), a quarter of that driver's run
        const unsigned b = c->big_next++ & 1u;
        if (c->big_busy[b]) HIPCHK(hipEventSynchronize(c->big_ev[b]));
        c->big_busy[b] = false;
        if (c->big_cap[b] < need) {
            if (c->h_big[b]) HIPCHK(hipHostFree(c->h_big[b]));
            c->h_big[b] = nullptr;
            c->big_cap[b] = 0;
            HIPCHK(hipHostMalloc((void **)&c->h_big[b], need * sizeof(double), hipHostMallocDefault));
            c->big_cap[b] = need;
        }
        if (!c->big_ev[b]) HIPCHK(hipEventCreateWithFlags(&c->big_ev[b], hipEventDisableTiming));
        double *h = c->h_big[b];
        memcpy(h, h_re, nv * sizeof(double));
        if (h_im) memcpy(h + nv, h_im, nv * sizeof(double));
        HIPCHK(hipMemcpyAsync(c->d_vals, h, need * sizeof(double), hipMemcpyHostToDevice, s));
        HIPCHK(hipEventRecord(c->big_ev[b], s));
        c->big_busy[b] = true;
    }
    if (c->logn == 15) {  // the N = 32768 transform is out of place: coefficients into scratch, NTT into d_out
        const size_t words = (size_t)count * L * c->n;
        if (int rc = ensure_scratch(c, words)) return rc;
        HIPCHK(launch_encode(c->T, c->E, c->d_vals, h_im ? c->d_vals + nv : nullptr, nvalues, count, scale, L, c->scratch, s));
        HIPCHK(launch_ntt_split15(c->T, false, c->scratch, (u64 *)d_out, count, L, 0, s));
        return HEFX_OK;
    }
    HIPCHK(launch_encode(c->T, c->E, c->d_vals, h_im ? c->d_vals + nv : nullptr, nvalues, count, scale, L, (u64 *)d_out, s));
    HIPCHK(launch_ntt(c->T, false, (u64 *)d_out, count, L, 0, s));
    return HEFX_OK;
}

// the same for `count` vectors whose plaintexts are separately allocated (pointer table): slices of vectors are encoded
// into engine scratch in one pass each and handed to their owners by one scatter launch -- the words hefx_ckks_encode
// writes for each vector.  What the shim's recorder calls for the encodes it has collected (2000 one-hot masks inside the
// reference's prediction loop, logistic_regression_ckks.cpp:222-225; 4018 in front of it).
extern "C" int hefx_ckks_encode_batch(hefx_context *c, int L, const double *h_re, const double *h_im, int nvalues,
                                      int count, double scale, uint64_t *const *d_outs, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (!d_outs || count < 1) return fail(HEFX_ERR_INVALID, "values has invalid size");
    if (c->logn == 15) {  // out-of-place transform: vector by vector
        for (int i = 0; i < count; ++i)
            if (int rc = hefx_ckks_encode(c, L, h_re + (size_t)i * nvalues, h_im ? h_im + (size_t)i * nvalues : nullptr, nvalues, 1,
                                          scale, d_outs[i], stream))
                return rc;
        return HEFX_OK;
    }
    for (int i = 0; i < count; ++i)
        if (!d_outs[i]) return fail(HEFX_ERR_INVALID, "null output pointer in batch");
    const size_t words = (size_t)L * c->n;
    const int cap = 256;  // vectors per slice: 256 x L x N words of scratch
    hipStream_t s = (hipStream_t)stream;
    return table_slices(
        c, count, 1, s, [&](const uint64_t **hp, int i0, int cnt) { for (int i = 0; i < cnt; ++i) hp[i] = d_outs[i0 + i]; },
        [&](const u64 *const *dp, int i0, int cnt) -> hipError_t {
            // (hefx_ckks_encode sizes its own staging; the scratch region is only this call's output area)
            if (ensure_scratch(c, words * (size_t)cnt) != HEFX_OK) return hipErrorOutOfMemory;
            if (hefx_ckks_encode(c, L, h_re + (size_t)i0 * nvalues, h_im ? h_im + (size_t)i0 * nvalues : nullptr, nvalues, cnt, scale,
                                 (uint64_t *)c->scratch, s) != HEFX_OK)
                return hipErrorUnknown;
            return launch_scatter_rows(c->scratch, dp, cnt, words, s);
        },
        cap);
}

// ---------------------------------------------------------------------------------------------
// Linear_Transform_Plain (helper.h:237-262 = linear_transformation2.cpp:149-174) as ONE call: the rotation plans
// (SEAL rotate_internal: the direct key if present, else the NAF terms of the step, App. A.7), their batching and
// the final sum all stay on this side of the C-ABI.  The result is bit-identical to the reference's sequence of
// rotate_vector / multiply_plain / add_many calls: every key switch is deterministic, so (a) a (source, Galois
// element) pair shared by several plans is computed once, (b) independent key switches of one depth run as one
// batch, (c) the last key switch of each plan is fused with its multiply_plain.
// ---------------------------------------------------------------------------------------------
namespace {
// Galois element -> key payload.  A sorted vector (callers hand the elements over in increasing order: one pass, no
// allocation per key), looked up by binary search: building an unordered_map of 512 keys was 30 us in front of the first
// launch of every direct-key linear transform.
struct LtKeys {
    std::vector<std::pair<uint32_t, const uint64_t *>> v;
    void set(int n, const uint32_t *elts, const uint64_t *const *keys)
    {
        v.resize((size_t)n);
        bool sorted = true;
        for (int i = 0; i < n; ++i) {
            v[(size_t)i] = {elts[i], keys[i]};
            sorted = sorted && (i == 0 || elts[i - 1] < elts[i]);
        }
        if (!sorted) {  // later entries of an element win, like repeated assignment to a map
            std::stable_sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            size_t w = 0;
            for (size_t i = 0; i < v.size(); ++i) {
                if (i + 1 < v.size() && v[i + 1].first == v[i].first) continue;
                v[w++] = v[i];
            }
            v.resize(w);
        }
    }
    const uint64_t *at(uint32_t e) const
    {
        auto it = std::lower_bound(v.begin(), v.end(), e, [](const auto &a, uint32_t x) { return a.first < x; });
        return it != v.end() && it->first == e ? it->second : nullptr;
    }
    bool has(uint32_t e) const { return at(e) != nullptr; }
};
uint32_t lt_elt_from_step(int step, size_t n)
{
    const uint64_t m = 2 * n;
    if (step == 0) return (uint32_t)(m - 1);
    uint64_t pos = step > 0 ? (uint64_t)step : (uint64_t)((long long)(n / 2) + step);
    uint64_t r = 1, b = 3;
    while (pos) {
        if (pos & 1) r = (r * b) & (m - 1);
        b = (b * b) & (m - 1);
        pos >>= 1;
    }
    return (uint32_t)r;
}
void lt_naf(int value, std::vector<int> &out)  // SEAL util::naf: least significant term first, signed like value
{
    const bool neg = value < 0;
    unsigned v = (unsigned)(neg ? -value : value);
    for (int i = 0; v; ++i) {
        const int zi = (v & 1) ? 2 - (int)(v & 3) : 0;
        v = (unsigned)((int)v - zi) >> 1;
        if (zi) out.push_back((neg ? -zi : zi) * (1 << i));
    }
}
// Galois elements rotate_vector(steps) applies, in order; nullptr on success, else SEAL's exception text
const char *lt_plan(int steps, size_t n, const LtKeys &keys, std::vector<uint32_t> &plan)
{
    if (steps == 0) return nullptr;
    if ((size_t)(steps < 0 ? -steps : steps) >= n / 2) return "step count too large";
    const uint32_t e = lt_elt_from_step(steps, n);
    if (keys.has(e)) {
        plan.push_back(e);
        return nullptr;
    }
    std::vector<int> terms;
    lt_naf(steps, terms);
    if (terms.size() == 1) return "Galois key not present";
    for (int t : terms) {
        if ((size_t)(t < 0 ? -t : t) == n / 2) continue;
        if (const char *err = lt_plan(t, n, keys, plan)) return err;
    }
    return nullptr;
}
}  // namespace

// A forest of key switches: node i rotates its parent's result (parent < 0: the external ciphertext in_ext) by `elt` with
// `key`, optionally multiplied by the plaintext `pt` in the epilogue, into `out`.  Parents precede their children; `depth`
// is the distance to the root.  Runs depth by depth, the subtrees on lanes (below).  Shared by hefx_linear_transform_plain
// and hefx_apply_galois_forest.
struct ForestNode {
    int parent;
    uint32_t elt;
    int depth;
    const uint64_t *in_ext, *key, *pt;
    uint64_t *out;
};
static int run_forest(hefx_context *c, int L, const std::vector<ForestNode> &nodes, int max_depth, void *stream, bool hoisted)
{
    std::vector<const uint64_t *> in, kk, pp;
    std::vector<uint64_t *> oo;
    std::vector<uint32_t> ee;
    // One batch per depth: the nodes that end in a diagonal product (leaves) and the ones that only feed deeper nodes are
    // independent of one another, so they go out together -- larger batches, i.e. more two-chunk submissions whose phases
    // overlap on the two internal streams, and half the latency-bound calls at the shallow depths (round 5; until then two
    // batches per depth.  HEFX_LT_MERGE=0 restores that).  A null entry of `pp` = no fused product for that item.
    static const bool merge = !(getenv("HEFX_LT_MERGE") && atoi(getenv("HEFX_LT_MERGE")) == 0);
    // TWO LANES (round 5) for a forest of chains (the reference's default keys: every rotation a NAF chain, five depths at
    // d = 512): a depth is a barrier only inside a subtree, so the subtrees below ct_new are dealt onto two lanes of about
    // equal size and each lane's depth batches run on a stream of their own -- the ramp and tail of one lane's five launches
    // under the other lane's wide ones.  Each lane's batches must be single chunks (ks_run's chunk pipeline is one per
    // context); lane 1 works in the back half of the scratch buffer, sized here for both.  HEFX_LT_LANES=0 switches it off.
    // From 96 nodes on (HEFX_LT_LANES_MIN): a small forest is a handful of latency-bound batches, and launches on two hardware
    // queues start later than on one (N = 8192, d = 10: 0.23 against 0.20 ms; d = 100: 0.55 against 0.57; d = 1000: 2.37
    // against 2.48; C3, d = 512: 3.25 against 3.43 -- profiles/r05/lt_naf_two_lanes.txt).
    static const bool lanes_ok = !(getenv("HEFX_LT_LANES") && atoi(getenv("HEFX_LT_LANES")) == 0);
    static const size_t lanes_min = getenv("HEFX_LT_LANES_MIN") ? (size_t)atoi(getenv("HEFX_LT_LANES_MIN")) : 96;
    constexpr int MAXL = 1 + hefx_context::MAX_STREAMS;
    // (HEFX_LT_NLANES: three lanes measure like two -- 3.31-3.33 against 3.29-3.33 ms at C3, d = 512; 2.30 both at N = 8192, d = 1000)
    static const int want_env = getenv("HEFX_LT_NLANES") ? atoi(getenv("HEFX_LT_NLANES")) : 2;
    const int want = std::max(2, std::min(want_env, 1 + c->nstreams));
    std::vector<uint8_t> lane_of(nodes.size(), 0);
    size_t lane1_off = 0;
    int nlanes = 1;
    if (lanes_ok && merge && max_depth >= 2 && nodes.size() >= lanes_min && c->use_streams && !c->profiling && c->sub <= 0 &&
        !c->fused) {
        std::vector<int> root_of(nodes.size()), weight(nodes.size(), 0);
        for (size_t i = 0; i < nodes.size(); ++i) {  // (a node's parent precedes it)
            root_of[i] = nodes[i].parent < 0 ? (int)i : root_of[(size_t)nodes[i].parent];
            ++weight[(size_t)root_of[i]];
        }
        std::vector<int> roots;
        for (size_t i = 0; i < nodes.size(); ++i)
            if (nodes[i].parent < 0) roots.push_back((int)i);
        std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return weight[(size_t)a] > weight[(size_t)b]; });
        int load[MAXL] = {};
        std::vector<uint8_t> root_lane(nodes.size(), 0);
        for (int r : roots) {  // heaviest subtree first, each onto the lightest lane
            int ln = 0;
            for (int q = 1; q < want; ++q)
                if (load[q] < load[ln]) ln = q;
            root_lane[(size_t)r] = (uint8_t)ln;
            load[ln] += weight[(size_t)r];
        }
        std::vector<int> cnt((size_t)max_depth * MAXL, 0);
        int widest = 0;
        for (size_t i = 0; i < nodes.size(); ++i) {
            lane_of[i] = root_lane[(size_t)root_of[i]];
            widest = std::max(widest, ++cnt[(size_t)nodes[i].depth * MAXL + lane_of[i]]);
        }
        bool all_loaded = true;
        for (int q = 0; q < want; ++q) all_loaded = all_loaded && load[q] > 0;
        int chunk = c->chunk;  // ks_run's rule for the items of one launch sequence
        if (chunk <= 0) {
            chunk = (int)(((size_t)1 << 30) / (ks_words_per_item(c, L) * sizeof(u64)));
            const int cap = c->logn <= 13 ? 2 * KS_AUTO_CHUNK : KS_AUTO_CHUNK;
            chunk = chunk > cap ? cap : (chunk < 16 ? 16 : chunk & ~7);
        }
        if (all_loaded && widest <= chunk) {
            lane1_off = ks_words_per_item(c, L) * (size_t)widest + ks_x_words(c, L, widest);
            if (int rc = ensure_scratch(c, (size_t)want * lane1_off)) return rc;
            nlanes = want;
            // whatever a lane's batches would build on first use is built HERE, on the caller's stream, before the lanes
            // part: the flip-mask tables of exact hoisting (their transform borrows the front of the scratch at N = 32768)
            // and the gather tables -- inside the lanes every lookup then hits
            std::vector<uint32_t> all_elts(nodes.size());
            for (size_t i = 0; i < nodes.size(); ++i) all_elts[i] = nodes[i].elt;
            std::sort(all_elts.begin(), all_elts.end());
            all_elts.erase(std::unique(all_elts.begin(), all_elts.end()), all_elts.end());
            std::vector<const u64 *> unused;
            if (int rc = ensure_flipw(c, all_elts.data(), (int)all_elts.size(), (hipStream_t)stream, unused)) return rc;
            for (uint32_t e : all_elts) {
                const uint32_t *perm = nullptr;
                if (int rc = get_perm(c, e, &perm)) return rc;
            }
        } else
            std::fill(lane_of.begin(), lane_of.end(), 0);
    }
    hipStream_t lane_stream[MAXL] = {(hipStream_t)stream};
    for (int q = 1; q < MAXL; ++q) lane_stream[q] = c->streams[q - 1];
    if (nlanes > 1) {
        HIPCHK(hipEventRecord(c->ev_fork, lane_stream[0]));
        for (int q = 1; q < nlanes; ++q) HIPCHK(hipStreamWaitEvent(lane_stream[q], c->ev_fork, 0));
    }
    int lane_rc = HEFX_OK;
    for (int depth = 0; depth < max_depth && lane_rc == HEFX_OK; ++depth)
        for (int pass = 0; pass < (merge ? nlanes : 2) && lane_rc == HEFX_OK; ++pass) {
            const int ln = nlanes > 1 ? pass : 0;
            in.clear(), kk.clear(), pp.clear(), oo.clear(), ee.clear();
            bool any_pt = false;
            for (size_t i = 0; i < nodes.size(); ++i) {
                const ForestNode &nd = nodes[i];
                if (nd.depth != depth || (!merge && (nd.pt != nullptr) != (pass != 0)) || (nlanes > 1 && lane_of[i] != ln)) continue;
                in.push_back((nd.parent < 0 ? nd.in_ext : nodes[(size_t)nd.parent].out));
                ee.push_back(nd.elt);
                kk.push_back(nd.key);
                oo.push_back(nd.out);
                pp.push_back(nd.pt);
                any_pt = any_pt || nd.pt != nullptr;
            }
            if (in.empty()) continue;
            // (node buffers are this context's workspace, disjoint by construction; the diagonals are the caller's and
            // cannot reach into it: the batch is trusted)
            // (an error leaves the loop, not the function: lane 1's stream must be joined to the caller's whatever happened)
            lane_rc = ks_run(c, L, (int)in.size(), false, in.data(), ee.data(), kk.data(), nullptr, any_pt ? pp.data() : nullptr,
                             oo.data(), lane_stream[ln], hoisted, nullptr, nullptr, true, (size_t)ln * lane1_off);
        }
    for (int q = 1; q < nlanes; ++q) {
        hipError_t ej = hipEventRecord(c->ev_join[q - 1], lane_stream[q]);
        if (ej == hipSuccess) ej = hipStreamWaitEvent(lane_stream[0], c->ev_join[q - 1], 0);
        if (ej != hipSuccess && lane_rc == HEFX_OK) lane_rc = hipfail(ej, "join of the linear transform's lanes");
    }
    return lane_rc;
}

// `count` transforms with the same dimension and key set in lockstep (count = 1: hefx_linear_transform_plain): transform t
// maps cts[t] with the diagonals diag_pts[t * d .. t * d + d) to outs[t].  The rotation plans are those of ONE transform;
// every launch sequence -- the -d rotation, each depth of the rotation forest -- carries the items of all `count` inputs
// (round 6: CC_Matrix_Multiplication's sigma / tau transforms of ctA / ctB, matrix_multiplication.cpp:22-25, are
// independent of one another: half the dependent launch sequences, each twice as wide).  Per input the operations and
// their order are those of the single transform: same bits.
static int lt_impl(hefx_context *c, int L, int count, const uint64_t *const *cts, int d, const uint64_t *const *diag_pts, int nkeys,
                   const uint32_t *key_elts, const uint64_t *const *keys, uint64_t *const *outs, void *stream, bool hoisted)
{
    CTXCHK(c);
    if (int rc = check_ks_level(c, L)) return rc;
    if (count < 1 || count > 64 || !cts || !outs || d < 1 || !diag_pts || nkeys < 0 || (nkeys && (!key_elts || !keys)))
        return fail(HEFX_ERR_INVALID, "bad linear-transform arguments");
    for (int t = 0; t < count; ++t)
        if (!cts[t] || !outs[t]) return fail(HEFX_ERR_INVALID, "bad linear-transform arguments");
    for (int i = 0; i < count * d; ++i)
        if (!diag_pts[i]) return fail(HEFX_ERR_INVALID, "null diagonal plaintext");
    static const bool dbg = getenv("HEFX_DEBUG") != nullptr;  // host time of the call's phases on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hefx] linear_transform(d=%d x %d): %-36s %7.1f us\n", d, count, what, std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = now;
    };
    LtKeys K;
    for (int i = 0; i < nkeys; ++i)
        if (!keys[i]) return fail(HEFX_ERR_INVALID, "null Galois key");
    K.set(nkeys, key_elts, keys);
    const size_t N = c->n, ctw = 2 * (size_t)L * N;
    std::vector<uint32_t> first;
    if (const char *err = lt_plan(-d, N, K, first)) return fail(HEFX_ERR_INVALID, err);
    // a missing key or a step too large must be reported before anything runs, like the op-by-op sequence would: plans of
    // every step first (cheap), the node forest later -- while the GPU is already at work on ct_new
    std::vector<std::vector<uint32_t>> plans((size_t)d);
    for (int l = 1; l < d; ++l) {
        if (const char *err = lt_plan(l, N, K, plans[(size_t)l])) return fail(HEFX_ERR_INVALID, err);
        if (hoisted && plans[(size_t)l].size() != 1)
            return fail(HEFX_ERR_INVALID, "hoisted linear transform needs a direct Galois key for every step 1..d-1");
    }
    lap("key map + rotation plans");
    // ---- workspace, part 1, per input: ping/pong for the first rotation chain, ct_new, product 0
    if (int rc = grow_retiring(c, &c->lt_head, &c->lt_head_cap, (size_t)count * 4 * ctw, 0, "linear-transform head")) return rc;
    uint64_t *head = reinterpret_cast<uint64_t *>(c->lt_head);
    auto ping = [&](int t) { return head + (size_t)t * 4 * ctw; };
    auto pong = [&](int t) { return ping(t) + ctw; };
    auto ct_new = [&](int t) { return ping(t) + 2 * ctw; };
    auto prod0 = [&](int t) { return ping(t) + 3 * ctw; };
    // ---- ct_new = ct + rotate(ct, -d)      (helper.h:244-247): launched FIRST, so that the planning below (0.1 ms of host
    // time at d = 512) runs beside it instead of in front of it.  The sum rides in the epilogue of the chain's last key
    // switch (acc_out = acc_in + rotation, the kernels of hefx_apply_galois_add_batch): one launch less than rotate + add
    {
        std::vector<const uint64_t *> src((size_t)count), kk((size_t)count), ain((size_t)count);
        std::vector<uint64_t *> dst((size_t)count), aout((size_t)count);
        std::vector<uint32_t> ee((size_t)count);
        for (int t = 0; t < count; ++t) src[(size_t)t] = cts[t], ain[(size_t)t] = cts[t], aout[(size_t)t] = ct_new(t);
        for (size_t s = 0; s < first.size(); ++s) {
            const bool last = s + 1 == first.size();
            for (int t = 0; t < count; ++t) {
                dst[(size_t)t] = (s & 1) ? pong(t) : ping(t);
                kk[(size_t)t] = K.at(first[s]);
                ee[(size_t)t] = first[s];
            }
            if (int rc = ks_run(c, L, count, false, src.data(), ee.data(), kk.data(), nullptr, nullptr, dst.data(), stream, false,
                                last ? ain.data() : nullptr, last ? aout.data() : nullptr))
                return rc;
            for (int t = 0; t < count; ++t) src[(size_t)t] = dst[(size_t)t];
        }
        if (first.empty())  // d == 0 mod N/2 cannot happen (lt_plan refuses it); kept for completeness: ct_new = ct + ct
            for (int t = 0; t < count; ++t)
                if (int rc = hefx_add(c, L, 2, 1, cts[t], cts[t], ct_new(t), stream)) return rc;
    }
    // ---- res[0] = ct_new * diag[0]         (helper.h:250): up to 96 diagonals the product is formed inside the final sum
    // (add_many_impl's pt0: one launch and the transparency bookkeeping of hefx_multiply_plain -- 15 us between the -d
    // rotation and the forest at d = 16, profiles/r06/lt_naf_d16_timeline_before.txt -- off the critical path); wider
    // transforms, whose sum goes through the table level, keep the launch
    const bool fuse0 = d <= 2 * ADD_MANY_GROUP;
    for (int t = 0; t < count && !fuse0; ++t)
        if (int rc = hefx_multiply_plain(c, L, 2, 1, ct_new(t), diag_pts[(size_t)t * d], prod0(t), stream)) return rc;
    lap("head submitted (rotate -d + add, product 0)");
    // ---- plans -> a forest of key-switch nodes rooted at ct_new, deduplicated per (parent, element, fused diagonal)
    struct Node {
        int parent;  // -1: ct_new
        uint32_t elt;
        int fused;   // diagonal index whose plaintext multiplies this node's output, or -1
        int depth;
    };
    std::vector<Node> nodes;
    // (parent node + 1, element, fused diagonal + 1) packed into 64 bits: parent < 2^22 nodes, element < 2^16 (N <= 32768),
    // diagonal < 2^22
    std::unordered_map<uint64_t, int> index;
    index.reserve((size_t)d * 2);
    nodes.reserve((size_t)d * 2);
    std::vector<int> leaf(d, -1);
    int max_depth = 0;
    for (int l = 1; l < d; ++l) {
        const std::vector<uint32_t> &plan = plans[(size_t)l];
        int cur = -1;
        for (size_t t = 0; t < plan.size(); ++t) {
            const int fused = t + 1 == plan.size() ? l : -1;
            const uint64_t key = ((uint64_t)(cur + 1) << 42) | ((uint64_t)plan[t] << 24) | (uint64_t)(fused + 1);
            auto it = index.find(key);
            if (it == index.end()) {
                nodes.push_back(Node{cur, plan[t], fused, (int)t});
                it = index.emplace(key, (int)nodes.size() - 1).first;
                if ((int)t + 1 > max_depth) max_depth = (int)t + 1;
            }
            cur = it->second;
        }
        leaf[l] = cur;
    }
    lap("forest");
    // ---- workspace, part 2: one ciphertext per node and input
    const size_t nn = nodes.size();
    const size_t need = ctw * nn * (size_t)count;
    if (int rc = grow_retiring(c, &c->lt_ws, &c->lt_cap, need ? need : 1, 0, "linear-transform nodes")) return rc;
    uint64_t *node0 = reinterpret_cast<uint64_t *>(c->lt_ws);
    auto node_ptr = [&](int t, int i) { return i < 0 ? ct_new(t) : node0 + ((size_t)t * nn + (size_t)i) * ctw; };
    // ---- res[l] = rotate(ct_new, l) * diag[l], depth by depth on lanes   (helper.h:252-257; run_forest): the forests of
    // the inputs side by side, input t's nodes at t * nn ..
    if (nn) {
        std::vector<ForestNode> fn(nn * (size_t)count);
        for (int t = 0; t < count; ++t)
            for (size_t i = 0; i < nn; ++i) {
                const Node &nd = nodes[i];
                fn[(size_t)t * nn + i] = ForestNode{nd.parent < 0 ? -1 : (int)((size_t)t * nn) + nd.parent, nd.elt, nd.depth, ct_new(t),
                                                    K.at(nd.elt), nd.fused >= 0 ? diag_pts[(size_t)t * d + nd.fused] : nullptr,
                                                    node_ptr(t, (int)i)};
            }
        if (int rc = run_forest(c, L, fn, max_depth, stream, hoisted)) return rc;
    }
    lap("key-switch batches submitted");
    // ---- out = add_many(res)               (helper.h:259)
    std::vector<const uint64_t *> res(d);
    int rc = HEFX_OK;
    for (int t = 0; t < count && rc == HEFX_OK; ++t) {
        res[0] = fuse0 ? ct_new(t) : prod0(t);
        for (int l = 1; l < d; ++l) res[l] = node_ptr(t, leaf[l]);
        rc = add_many_impl(c, L, 2, d, res.data(), fuse0 ? diag_pts[(size_t)t * d] : nullptr, outs[t], stream);
    }
    lap("add_many submitted");
    return rc;
}

extern "C" int hefx_apply_galois_forest(hefx_context *c, int L, int n, const int32_t *parent, const uint64_t *const *ext_in,
                                        const uint32_t *elts, const uint64_t *const *keys, const uint64_t *const *pts,
                                        uint64_t *const *outs, void *stream)
{
    CTXCHK(c);
    if (int rc = check_ks_level(c, L)) return rc;
    if (n < 1 || !parent || !ext_in || !elts || !keys || !outs) return fail(HEFX_ERR_INVALID, "bad rotation-forest arguments");
    std::vector<ForestNode> nodes((size_t)n);
    int max_depth = 0;
    for (int i = 0; i < n; ++i) {
        if (parent[i] >= i) return fail(HEFX_ERR_INVALID, "rotation forest: a node's parent must precede it");
        if (!keys[i] || !outs[i] || (parent[i] < 0 && !ext_in[i])) return fail(HEFX_ERR_INVALID, "null pointer in rotation forest");
        const int depth = parent[i] < 0 ? 0 : nodes[(size_t)parent[i]].depth + 1;
        nodes[(size_t)i] = ForestNode{parent[i] < 0 ? -1 : parent[i], elts[i], depth, parent[i] < 0 ? ext_in[i] : nullptr, keys[i],
                                      pts ? pts[i] : nullptr, outs[i]};
        max_depth = std::max(max_depth, depth + 1);
    }
    // outputs pairwise disjoint; no external input or plaintext reaches into an output (byte ranges, like ks_run's check --
    // the per-depth batches below then run trusted)
    {
        const size_t row = (size_t)c->n * sizeof(u64), out_b = 2 * (size_t)L * row, pt_b = (size_t)L * row;
        std::vector<uintptr_t> o((size_t)n);
        for (int i = 0; i < n; ++i) o[(size_t)i] = (uintptr_t)outs[i];
        std::sort(o.begin(), o.end());
        for (int i = 1; i < n; ++i)
            if (o[(size_t)i - 1] + out_b > o[(size_t)i]) return fail(HEFX_ERR_INVALID, "rotation forest: two nodes write overlapping outputs");
        auto hits = [&](const void *p, size_t bytes) {
            const uintptr_t a = (uintptr_t)p;
            auto it = std::upper_bound(o.begin(), o.end(), a);
            if (it != o.begin() && *(it - 1) + out_b > a) return true;
            return it != o.end() && *it < a + bytes;
        };
        for (int i = 0; i < n; ++i) {
            if (parent[i] < 0 && hits(ext_in[i], out_b))
                return fail(HEFX_ERR_INVALID, "rotation forest: an external input overlaps a node's output");
            if (pts && pts[i] && hits(pts[i], pt_b)) return fail(HEFX_ERR_INVALID, "rotation forest: a plaintext overlaps a node's output");
        }
    }
    return run_forest(c, L, nodes, max_depth, stream, false);
}

extern "C" int hefx_linear_transform_plain(hefx_context *c, int L, const uint64_t *ct, int d,
                                           const uint64_t *const *diag_pts, int nkeys, const uint32_t *key_elts,
                                           const uint64_t *const *keys, uint64_t *out, void *stream)
{
    return lt_impl(c, L, 1, &ct, d, diag_pts, nkeys, key_elts, keys, &out, stream, false);
}
extern "C" int hefx_linear_transform_plain_many(hefx_context *c, int L, int count, const uint64_t *const *cts, int d,
                                                const uint64_t *const *diag_pts, int nkeys, const uint32_t *key_elts,
                                                const uint64_t *const *keys, uint64_t *const *outs, void *stream)
{
    return lt_impl(c, L, count, cts, d, diag_pts, nkeys, key_elts, keys, outs, stream, false);
}
extern "C" int hefx_linear_transform_plain_hoisted(hefx_context *c, int L, const uint64_t *ct, int d,
                                                   const uint64_t *const *diag_pts, int nkeys,
                                                   const uint32_t *key_elts, const uint64_t *const *keys,
                                                   uint64_t *out, void *stream)
{
    return lt_impl(c, L, 1, &ct, d, diag_pts, nkeys, key_elts, keys, &out, stream, true);
}
// Baby-step / giant-step Linear_Transform_Plain: with l = j*n1 + i,
//   sum_l diag_l (.) rot_l(ct_new) = sum_j rot_(j*n1)( sum_i diag'_l (.) rot_i(ct_new) ),  diag'_l = diag_l shifted
// right by j*n1 slots in the clear (the caller encodes them that way).  n1-1 baby rotations of ct_new (one hoisted
// batch, or regular key switches), the inner sums in one pass (hefx_multiply_plain_sum), n2-1 giant rotations (one
// regular batch), one add_many.  Direct Galois keys for 1..n1-1 and n1, 2*n1, ..; rotate(-d) may use a NAF chain.
extern "C" int hefx_linear_transform_plain_bsgs(hefx_context *c, int L, const uint64_t *ct, int d, int n1,
                                                const uint64_t *const *shifted_diag_pts, int nkeys,
                                                const uint32_t *key_elts, const uint64_t *const *keys, int hoisted_baby,
                                                uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_ks_level(c, L)) return rc;
    if (!ct || !out || d < 1 || n1 < 1 || n1 > d || !shifted_diag_pts || nkeys < 0 || (nkeys && (!key_elts || !keys)))
        return fail(HEFX_ERR_INVALID, "bad linear-transform arguments");
    const int n2 = (d + n1 - 1) / n1;
    if ((size_t)d + (size_t)n1 * n2 > c->n / 2)
        return fail(HEFX_ERR_INVALID, "baby-step/giant-step transform: dimension too large for the slot count");
    for (int i = 0; i < d; ++i)
        if (!shifted_diag_pts[i]) return fail(HEFX_ERR_INVALID, "null diagonal plaintext");
    static const bool dbg = getenv("HEFX_DEBUG") != nullptr;  // host time of the call's phases on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hefx] linear_transform(d=%d): %-36s %7.1f us\n", d, what, std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = now;
    };
    LtKeys K;
    for (int i = 0; i < nkeys; ++i)
        if (!keys[i]) return fail(HEFX_ERR_INVALID, "null Galois key");
    K.set(nkeys, key_elts, keys);
    const size_t N = c->n, ctw = 2 * (size_t)L * N;
    std::vector<uint32_t> first, plan, belt, gelt;
    if (const char *err = lt_plan(-d, N, K, first)) return fail(HEFX_ERR_INVALID, err);
    for (int t = 1; t < n1 + n2 - 1; ++t) {  // baby steps 1..n1-1, then giant steps n1, 2*n1, ...
        const int step = t < n1 ? t : (t - n1 + 1) * n1;
        plan.clear();
        if (const char *err = lt_plan(step, N, K, plan)) return fail(HEFX_ERR_INVALID, err);
        if (plan.size() != 1)
            return fail(HEFX_ERR_INVALID,
                        "baby-step/giant-step transform needs a direct Galois key for every baby and giant step");
        (t < n1 ? belt : gelt).push_back(plan[0]);
    }
    // ---- workspace: ping / pong / ct_new, n1-1 baby rotations, n2 inner sums, n2-1 rotated inner sums
    const size_t need = ctw * (size_t)(3 + (n1 - 1) + n2 + (n2 - 1));
    if (int rc = grow_retiring(c, &c->lt_ws, &c->lt_cap, need, 0, "linear-transform workspace")) return rc;
    uint64_t *ping = reinterpret_cast<uint64_t *>(c->lt_ws), *pong = ping + ctw, *ct_new = pong + ctw,
             *rots = ct_new + ctw, *inner = rots + (size_t)(n1 - 1) * ctw, *grot = inner + (size_t)n2 * ctw;
    // ---- ct_new = ct + rotate(ct, -d)      (helper.h:244-247)
    const uint64_t *src = ct;
    for (size_t t = 0; t < first.size(); ++t) {
        uint64_t *dst = (t & 1) ? pong : ping;
        const uint64_t *key = K.at(first[t]);
        if (int rc = ks_run(c, L, 1, false, &src, &first[t], &key, nullptr, nullptr, &dst, stream)) return rc;
        src = dst;
    }
    if (int rc = hefx_add(c, L, 2, 1, ct, src, ct_new, stream)) return rc;
    std::vector<const uint64_t *> in, kk;
    std::vector<uint64_t *> oo;
    // ---- baby steps: rots[i-1] = rotate(ct_new, i)
    if (n1 > 1) {
        for (int i = 1; i < n1; ++i) {
            in.push_back(ct_new);
            kk.push_back(K.at(belt[i - 1]));
            oo.push_back(rots + (size_t)(i - 1) * ctw);
        }
        if (int rc = ks_run(c, L, n1 - 1, false, in.data(), belt.data(), kk.data(), nullptr, nullptr, oo.data(), stream,
                            hoisted_baby != 0))
            return rc;
    }
    // ---- inner[j] = sum_i rotate(ct_new, i) (.) diag'[j*n1 + i]
    std::vector<const uint64_t *> cc(d);
    std::vector<uint64_t *> io(n2);
    for (int l = 0; l < d; ++l) cc[l] = (l % n1) ? rots + (size_t)(l % n1 - 1) * ctw : ct_new;
    for (int j = 0; j < n2; ++j) io[j] = inner + (size_t)j * ctw;
    if (int rc = hefx_multiply_plain_sum(c, L, 2, d, n1, cc.data(), shifted_diag_pts, io.data(), stream)) return rc;
    // ---- giant steps and the final sum
    std::vector<const uint64_t *> res(n2);
    res[0] = inner;
    if (n2 > 1) {
        in.clear(), kk.clear(), oo.clear();
        for (int j = 1; j < n2; ++j) {
            in.push_back(inner + (size_t)j * ctw);
            kk.push_back(K.at(gelt[j - 1]));
            oo.push_back(grot + (size_t)(j - 1) * ctw);
            res[j] = oo.back();
        }
        if (int rc = ks_run(c, L, n2 - 1, false, in.data(), gelt.data(), kk.data(), nullptr, nullptr, oo.data(), stream))
            return rc;
    }
    return hefx_add_many(c, L, 2, n2, res.data(), out, stream);
}

// Double-hoisted Linear_Transform_Plain (see lt2_mac_kernel): top data level, direct keys for 1..d-1, diagonals
// encoded at the KEY level ([k][N]: data primes then the special prime).
// terms: term 0 is the unrotated one (steps[0] == 0), term i >= 1 rotates ct_new by steps[i]; d only enters through
// the duplication rotate(ct, -d).  The dense transform is steps = 0, 1, .., d-1.
static int lt2_impl(hefx_context *c, int L, const uint64_t *ct, int d, int nterms, const int *steps,
                    const uint64_t *const *diag_pts_keylevel, int nkeys, const uint32_t *key_elts,
                    const uint64_t *const *keys, uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_ks_level(c, L)) return rc;
    if (L != c->k - 1) return fail(HEFX_ERR_UNSUPPORTED, "double hoisting is built for the top data level");
    if (!ct || !out || d < 1 || nterms < 1 || !steps || steps[0] != 0 || !diag_pts_keylevel || nkeys < 0 ||
        (nkeys && (!key_elts || !keys)))
        return fail(HEFX_ERR_INVALID, "bad linear-transform arguments");
    for (int i = 0; i < nterms; ++i)
        if (!diag_pts_keylevel[i]) return fail(HEFX_ERR_INVALID, "null diagonal plaintext");
    static const bool dbg = getenv("HEFX_DEBUG") != nullptr;  // host time of the call's phases on stderr
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hefx] linear_transform(d=%d): %-36s %7.1f us\n", d, what, std::chrono::duration<double, std::micro>(now - t_last).count());
        t_last = now;
    };
    LtKeys K;
    for (int i = 0; i < nkeys; ++i)
        if (!keys[i]) return fail(HEFX_ERR_INVALID, "null Galois key");
    K.set(nkeys, key_elts, keys);
    const size_t N = c->n, ctw = 2 * (size_t)L * N;
    const int k = c->k, nrot = nterms - 1, chunks = nrot > 0 ? (nrot + lt2_chunk() - 1) / lt2_chunk() : 0;
    std::vector<uint32_t> first, plan;
    if (const char *err = lt_plan(-d, N, K, first)) return fail(HEFX_ERR_INVALID, err);
    std::vector<KsItem> items((size_t)nterms + 1);  // [0]: source (ct_new); [1..nterms-1]: rotations; last: the mod-down
    // ---- workspace
    const size_t ws_words = 4 * ctw + (size_t)chunks * (2 * (size_t)k + L) * N + (items.size() * sizeof(KsItem) + 7) / 8;
    if (int rc = grow_retiring(c, &c->lt_ws, &c->lt_cap, ws_words, 0, "linear-transform workspace")) return rc;
    uint64_t *ping = reinterpret_cast<uint64_t *>(c->lt_ws), *pong = ping + ctw, *ct_new = pong + ctw, *cbuf = ct_new + ctw;
    u64 *partial_s = reinterpret_cast<u64 *>(cbuf + ctw), *partial_c0 = partial_s + (size_t)chunks * 2 * k * N;
    KsItem *d_items = reinterpret_cast<KsItem *>(partial_c0 + (size_t)chunks * L * N);
    for (int l = 1; l < nterms; ++l) {
        plan.clear();
        if (steps[l] == 0) return fail(HEFX_ERR_INVALID, "only term 0 may be unrotated");
        if (const char *err = lt_plan(steps[l], N, K, plan)) return fail(HEFX_ERR_INVALID, err);
        if (plan.size() != 1)
            return fail(HEFX_ERR_INVALID, "hoisted linear transform needs a direct Galois key for every step 1..d-1");
        KsItem &it = items[l];
        it.c_in = nullptr;
        it.c_out = nullptr;
        it.key = (const u64 *)K.at(plan[0]);
        it.pt = (const u64 *)diag_pts_keylevel[l];
        it.elt = plan[0];
        it.flags = 0;
        if (int rc = get_perm(c, plan[0], &it.perm)) return rc;
    }
    items[0] = KsItem{(const u64 *)ct_new, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1u, 0u};
    items[nterms] = KsItem{(const u64 *)cbuf, nullptr, nullptr, nullptr, (u64 *)out, nullptr, nullptr, 1u, 0u};
    // ---- ct_new = ct + rotate(ct, -d); cbuf = ct_new * diag_0
    const uint64_t *src = ct;
    for (size_t t = 0; t < first.size(); ++t) {
        uint64_t *dst = (t & 1) ? pong : ping;
        const uint64_t *key = K.at(first[t]);
        if (int rc = ks_run(c, L, 1, false, &src, &first[t], &key, nullptr, nullptr, &dst, stream)) return rc;
        src = dst;
    }
    if (int rc = hefx_add(c, L, 2, 1, ct, src, ct_new, stream)) return rc;
    if (int rc = hefx_multiply_plain(c, L, 2, 1, ct_new, diag_pts_keylevel[0], cbuf, stream)) return rc;
    if (nrot == 0) {
        HIPCHK(hipMemcpyAsync(out, cbuf, ctw * sizeof(u64), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return HEFX_OK;
    }
    // ---- decomposition of ct_new (once), gathered MACs of all rotations, one mod-down
    const size_t per = ks_words_per_item(c, L) + ks_x_words(c, L, 1);
    if (int rc = ensure_scratch(c, per)) return rc;
    KsScratch S{};
    S.d = c->scratch;
    S.acc = S.d + (size_t)L * N;
    S.u = S.acc + (size_t)2 * (L + 1) * N;
    S.x = S.u + (size_t)2 * N;
    S.alias = nullptr;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(d_items, items.data(), sizeof(KsItem) * items.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));  // `items` is a local
    HIPCHK(launch_lt2_decompose(c->T, L, d_items, d_items + 1, nrot, S, (const u64 *)ct_new, partial_s, partial_c0,
                                (u64 *)cbuf, s));
    HIPCHK(launch_lt2_moddown(c->T, L, d_items + nterms, S, s));
    return HEFX_OK;
}

extern "C" int hefx_linear_transform_plain_hoisted2(hefx_context *c, int L, const uint64_t *ct, int d,
                                                    const uint64_t *const *diag_pts_keylevel, int nkeys,
                                                    const uint32_t *key_elts, const uint64_t *const *keys,
                                                    uint64_t *out, void *stream)
{
    if (d < 1) return fail(HEFX_ERR_INVALID, "bad linear-transform arguments");
    std::vector<int> steps(d);
    for (int l = 0; l < d; ++l) steps[l] = l;
    return lt2_impl(c, L, ct, d, d, steps.data(), diag_pts_keylevel, nkeys, key_elts, keys, out, stream);
}
// The same over a subset of the diagonals: term i multiplies rotate(ct_new, steps[i]) by d_diag_pts_keylevel[i];
// steps[0] must be 0 (the unrotated term), the others non-zero with a direct Galois key each.
extern "C" int hefx_linear_transform_plain_hoisted2_sparse(hefx_context *c, int L, const uint64_t *ct, int d, int nterms,
                                                           const int *steps, const uint64_t *const *diag_pts_keylevel,
                                                           int nkeys, const uint32_t *key_elts,
                                                           const uint64_t *const *keys, uint64_t *out, void *stream)
{
    return lt2_impl(c, L, ct, d, nterms, steps, diag_pts_keylevel, nkeys, key_elts, keys, out, stream);
}

extern "C" int hefx_rotate_hoisted_batch(hefx_context *c, int L, const uint64_t *ct_in, int n, const uint32_t *elts,
                                         const uint64_t *const *keys, const uint64_t *const *pts,
                                         uint64_t *const *ct_out, void *stream)
{
    if (n < 1 || !ct_in) return fail(HEFX_ERR_INVALID, "bad hoisted batch arguments");
    if (pts)
        for (int i = 0; i < n; ++i)
            if (!pts[i]) return fail(HEFX_ERR_INVALID, "null plaintext pointer in batch");
    std::vector<const uint64_t *> in((size_t)n, ct_in);
    return ks_run(c, L, n, false, in.data(), elts, keys, nullptr, pts, ct_out, stream, true);
}

// ---------------------------------------------------------------------------------------------
// Randomness, encrypt, decrypt on the GPU
// ---------------------------------------------------------------------------------------------
static SampleKey sample_key(const uint8_t *key32)
{
    SampleKey k;
    for (int i = 0; i < 8; ++i)
        k.w[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) |
                 ((uint32_t)key32[4 * i + 3] << 24);
    return k;
}

static int sample_common(hefx_context *c, int mode, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                         int mod_first, uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (!key32 || !out || npoly < 1 || nrows < 1 || mod_first < 0 || mod_first + nrows > c->k)
        return fail(HEFX_ERR_INVALID, "bad sampling arguments");
    if ((size_t)npoly * nrows >= ((size_t)1 << 32)) return fail(HEFX_ERR_INVALID, "too many rows in one sampling call");
    HIPCHK(launch_sample(c->T, mode, sample_key(key32), c->noise, stream_id, npoly, nrows, mod_first, (u64 *)out,
                         (hipStream_t)stream));
    return HEFX_OK;
}
extern "C" int hefx_sample_uniform(hefx_context *c, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                                   int mod_first, uint64_t *out, void *stream)
{
    return sample_common(c, SAMPLE_UNIFORM, key32, stream_id, npoly, nrows, mod_first, out, stream);
}
extern "C" int hefx_sample_ternary(hefx_context *c, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                                   int mod_first, uint64_t *out, void *stream)
{
    return sample_common(c, SAMPLE_TERNARY, key32, stream_id, npoly, nrows, mod_first, out, stream);
}
extern "C" int hefx_sample_noise(hefx_context *c, const uint8_t *key32, uint64_t stream_id, int npoly, int nrows,
                                 int mod_first, uint64_t *out, void *stream)
{
    return sample_common(c, SAMPLE_NOISE, key32, stream_id, npoly, nrows, mod_first, out, stream);
}

extern "C" int hefx_encrypt(hefx_context *c, int L, const uint64_t *pk, const uint64_t *plain, const uint8_t *key32,
                            uint64_t stream_id, uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (!pk || !key32 || !out) return fail(HEFX_ERR_INVALID, "bad encrypt arguments");
    if (stream_id >> 62) return fail(HEFX_ERR_INVALID, "stream id must be below 2^62");
    const size_t N = c->n, rowsz = (size_t)L * N;
    const bool split = c->logn == 15;  // the N = 32768 transform is out of place
    if (int rc = ensure_scratch(c, (split ? 6 : 3) * rowsz)) return rc;
    u64 *u = c->scratch, *e = u + rowsz;
    hipStream_t s = (hipStream_t)stream;
    const SampleKey k = sample_key(key32);
    // sub-streams 4*id + {0: u, 1: e0, 2: e1}
    HIPCHK(launch_sample(c->T, SAMPLE_TERNARY, k, c->noise, 4 * stream_id + 0, 1, L, 0, u, s));
    HIPCHK(launch_sample(c->T, SAMPLE_NOISE, k, c->noise, 4 * stream_id + 1, 1, L, 0, e, s));
    HIPCHK(launch_sample(c->T, SAMPLE_NOISE, k, c->noise, 4 * stream_id + 2, 1, L, 0, e + rowsz, s));
    if (split) {
        HIPCHK(launch_ntt_split15(c->T, false, u, u + 3 * rowsz, 3, L, 0, s));
        u += 3 * rowsz;
        e += 3 * rowsz;
    } else {
        HIPCHK(launch_ntt(c->T, false, u, 3, L, 0, s));
    }
    HIPCHK(launch_encrypt_combine(c->T, L, (const u64 *)pk, u, e, (const u64 *)plain, (u64 *)out, s));
    return HEFX_OK;
}

// n encryptions under one public key and sampler key, item i with stream id first_stream_id + i: what n hefx_encrypt
// calls with those ids produce, word for word (the sampler is counter mode: item i's polynomials are polynomial 0 of the
// sub-streams 4 (first + i) + {0, 1, 2}), as three sampling launches, one transform launch and one combine launch per
// slice of items instead of five launches per item.  d_plains[i] may be NULL (encryption of zero); outputs through a
// pointer table.  The 4018 + 2013 encode / encrypt calls in front of the reference's LR training loop
// (logistic_regression_ckks.cpp:560-640) arrive here as a handful of calls (include/seal/seal.h records them).
extern "C" int hefx_encrypt_batch(hefx_context *c, int L, int n, const uint64_t *pk, const uint64_t *const *plains,
                                  const uint8_t *key32, uint64_t first_stream_id, uint64_t *const *outs, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (!pk || !key32 || !outs || n < 1) return fail(HEFX_ERR_INVALID, "bad encrypt arguments");
    if ((first_stream_id + (uint64_t)n) >> 62) return fail(HEFX_ERR_INVALID, "stream id must be below 2^62");
    for (int i = 0; i < n; ++i)
        if (!outs[i]) return fail(HEFX_ERR_INVALID, "null output pointer in batch");
    if (c->logn == 15) {  // the N = 32768 transform is out of place: item by item (no caller batches there yet)
        for (int i = 0; i < n; ++i)
            if (int rc = hefx_encrypt(c, L, pk, plains ? plains[i] : nullptr, key32, first_stream_id + (uint64_t)i, outs[i], stream))
                return rc;
        return HEFX_OK;
    }
    const size_t N = c->n, rowsz = (size_t)L * N;
    hipStream_t s = (hipStream_t)stream;
    const SampleKey k = sample_key(key32);
    // slices of at most 256 items (3 polynomials of scratch each) and at most one pointer-table ring slot
    const int cap = 256;
    if (int rc = ensure_scratch(c, 3 * rowsz * (size_t)(n < cap ? n : cap))) return rc;
    return table_slices(
        c, n, 2, s,
        [&](const uint64_t **hp, int i0, int cnt) {
            for (int i = 0; i < cnt; ++i) {
                hp[i] = plains ? plains[i0 + i] : nullptr;
                hp[cnt + i] = outs[i0 + i];
            }
        },
        [&](const u64 *const *dp, int i0, int cnt) -> hipError_t {  // cnt <= cap: one slice, one set of launches
            u64 *u = c->scratch, *ee = u + (size_t)cnt * rowsz;
            const uint64_t sid = 4 * (first_stream_id + (uint64_t)i0);
            hipError_t e = launch_sample(c->T, SAMPLE_TERNARY, k, c->noise, sid + 0, cnt, L, 0, u, s, 4);
            if (e == hipSuccess) e = launch_sample(c->T, SAMPLE_NOISE, k, c->noise, sid + 1, cnt, L, 0, ee, s, 4);
            if (e == hipSuccess)
                e = launch_sample(c->T, SAMPLE_NOISE, k, c->noise, sid + 2, cnt, L, 0, ee + (size_t)cnt * rowsz, s, 4);
            if (e == hipSuccess) e = launch_ntt(c->T, false, u, 3 * cnt, L, 0, s);
            if (e == hipSuccess) e = launch_encrypt_combine_table(c->T, L, cnt, (const u64 *)pk, u, ee, dp, s);
            return e;
        },
        cap);
}

extern "C" int hefx_decrypt(hefx_context *c, int L, int size, const uint64_t *ct, const uint64_t *sk, uint64_t *out,
                            void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (!ct || !sk || !out || size < 1) return fail(HEFX_ERR_INVALID, "bad decrypt arguments");
    HIPCHK(launch_decrypt(c->T, L, size, (const u64 *)ct, (const u64 *)sk, (u64 *)out, (hipStream_t)stream));
    return HEFX_OK;
}

extern "C" int hefx_keygen_kswitch(hefx_context *c, const uint64_t *sk, const uint64_t *new_sk, const uint8_t *key32,
                                   uint64_t stream_id, uint64_t *out, void *stream)
{
    CTXCHK(c);
    if (c->k < 2) return fail(HEFX_ERR_UNSUPPORTED, "keyswitching is not supported by the context");
    if (!sk || !new_sk || !key32 || !out) return fail(HEFX_ERR_INVALID, "bad key generation arguments");
    if (stream_id >> 62) return fail(HEFX_ERR_INVALID, "stream id must be below 2^62");
    const size_t N = c->n, words = (size_t)(c->k - 1) * c->k * N;
    const bool split = c->logn == 15;
    if (int rc = ensure_scratch(c, (split ? 3 : 2) * words)) return rc;
    u64 *a = c->scratch, *e = a + words;
    hipStream_t s = (hipStream_t)stream;
    const SampleKey k = sample_key(key32);
    // sub-streams 2*id: uniform a, 2*id+1: noise e
    HIPCHK(launch_sample(c->T, SAMPLE_UNIFORM, k, c->noise, 2 * stream_id, c->k - 1, c->k, 0, a, s));
    HIPCHK(launch_sample(c->T, SAMPLE_NOISE, k, c->noise, 2 * stream_id + 1, c->k - 1, c->k, 0, e, s));
    if (split) {
        HIPCHK(launch_ntt_split15(c->T, false, e, e + words, c->k - 1, c->k, 0, s));
        e += words;
    } else {
        HIPCHK(launch_ntt(c->T, false, e, c->k - 1, c->k, 0, s));
    }
    HIPCHK(launch_keygen_combine(c->T, (const u64 *)sk, (const u64 *)new_sk, a, e, (u64 *)out, s));
    return HEFX_OK;
}

extern "C" int hefx_galois_permute(hefx_context *c, uint32_t galois_elt, const uint64_t *in, int rows, uint64_t *out,
                                   void *stream)
{
    CTXCHK(c);
    if (!in || !out || rows < 1 || in == out) return fail(HEFX_ERR_INVALID, "bad permutation arguments");
    const uint32_t *perm = nullptr;
    if (int rc = get_perm(c, galois_elt, &perm)) return rc;
    HIPCHK(launch_galois_permute(c->T, perm, (const u64 *)in, rows, (u64 *)out, (hipStream_t)stream));
    return HEFX_OK;
}

// ---------------------------------------------------------------------------------------------
// CKKS decode on the GPU
// ---------------------------------------------------------------------------------------------
// mixed-radix digits of floor(Q/2) for Q = q_0 ... q_(L-1): little-endian big integer on the host
static DecodeTables decode_tables(const hefx_context *c, int L)
{
    std::vector<u64> Q{1};
    for (int j = 0; j < L; ++j) {  // Q *= q_j
        u128 carry = 0;
        for (auto &limb : Q) {
            const u128 t = (u128)limb * c->primes[j] + carry;
            limb = (u64)t;
            carry = t >> 64;
        }
        if (carry) Q.push_back((u64)carry);
    }
    for (size_t i = 0; i < Q.size(); ++i)  // Q >>= 1 (Q is odd: floor)
        Q[i] = (Q[i] >> 1) | (i + 1 < Q.size() ? Q[i + 1] << 63 : 0);
    DecodeTables D{};
    for (int j = 0; j < L; ++j) {  // digit j = Q mod q_j; Q /= q_j
        u128 rem = 0;
        for (size_t i = Q.size(); i-- > 0;) {
            const u128 cur = (rem << 64) | Q[i];
            Q[i] = (u64)(cur / c->primes[j]);
            rem = cur % c->primes[j];
        }
        D.half[j] = (u64)rem;
    }
    return D;
}

extern "C" int hefx_ckks_decode(hefx_context *c, int L, const uint64_t *d_pt, int count, double scale, double *h_re,
                                double *h_im, void *stream)
{
    CTXCHK(c);
    if (int rc = check_level(c, L)) return rc;
    if (c->logn < 10 || c->logn > 15) return fail(HEFX_ERR_UNSUPPORTED, "GPU decode is built for poly_degree in [1024, 32768]");
    if (L > 16) return fail(HEFX_ERR_UNSUPPORTED, "GPU decode handles at most 16 primes");
    if (!d_pt || !h_re || count < 1) return fail(HEFX_ERR_INVALID, "plain is not valid for encryption parameters");
    if (!(scale > 0)) return fail(HEFX_ERR_INVALID, "scale out of bounds");
    if (int rc = ensure_encode_tables(c)) return rc;
    const size_t N = c->n, words = (size_t)count * L * N;
    const bool split = c->logn == 15;
    if (int rc = ensure_scratch(c, (split ? 2 : 1) * words)) return rc;
    const size_t ndbl = (size_t)count * N * 2;  // p [count][N] | re [count][N/2] | im [count][N/2]
    if (c->vals_cap < ndbl) {
        if (c->d_vals) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipFree(c->d_vals));
            c->d_vals = nullptr;
            c->vals_cap = 0;
        }
        HIPCHK(hipMalloc((void **)&c->d_vals, ndbl * sizeof(double)));
        c->vals_cap = ndbl;
    }
    hipStream_t s = (hipStream_t)stream;
    u64 *coef = c->scratch;
    if (split) {
        HIPCHK(launch_ntt_split15(c->T, true, (const u64 *)d_pt, coef, count, L, 0, s));
    } else {
        HIPCHK(hipMemcpyAsync(coef, d_pt, words * sizeof(u64), hipMemcpyDeviceToDevice, s));
        HIPCHK(launch_ntt(c->T, true, coef, count, L, 0, s));
    }
    double *p = c->d_vals, *re = p + (size_t)count * N, *im = re + (size_t)count * (N / 2);
    HIPCHK(launch_decode(c->T, c->E, decode_tables(c, L), L, coef, count, scale, p, re, h_im ? im : nullptr, s));
    HIPCHK(hipMemcpyAsync(h_re, re, sizeof(double) * count * (N / 2), hipMemcpyDeviceToHost, s));
    if (h_im) HIPCHK(hipMemcpyAsync(h_im, im, sizeof(double) * count * (N / 2), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return HEFX_OK;
}
