// hefx_kernels.hip -- hand-written gfx950 kernels for the CKKS evaluator hot path.
//
// Kernel families: K1/K2 NTT, K3/K4/K10/K11 dyadic element-wise, K5 Galois gather (fused
// into the key-switch loads), K6/K7 key switch (digit INTT -> per-modulus NTT -> 128-bit MAC with the key
// -> mod-down by the special prime, fused with the Galois/relin add-in and an optional multiply_plain),
// K8 rescale.  Pure 64-bit modular integer work: no MFMA.  Bounds: the NTT kernels are integer-VALU
// bound (64-bit modmul emulated with v_mad_u64_u32), the element-wise and MAC kernels are HBM/L2 bound.
#include "hefx_internal.h"
#include "hefx_ntt.cuh"

namespace hefx {

// ------------------------------------------------------------------------------------------------
// K1/K2: stand-alone NTT, one workgroup per RNS row, in place.
// ------------------------------------------------------------------------------------------------
template <int LOGN, bool INV>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void ntt_rows_kernel(DevTables T, u64 *data, int nrows, int mod_first)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x;
    const int row = blockIdx.x, poly = blockIdx.y;
    const int m = mod_first + row;
    u64 *p = data + ((size_t)poly * nrows + row) * C::N;
    const ModConst mc = T.mods[m];
    u64 v[16];
    if (!INV) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = p[C::idx_nat(t, r)];
        ntt_fwd_core<LOGN>(v, lds, T.tw + (size_t)m * C::N, mc.q, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[C::idx_out(t, r)] = v[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = p[C::idx_out(t, r)];
        ntt_inv_core<LOGN>(v, lds, T.itw + (size_t)m * C::N, mc, t);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[C::idx_nat(t, r)] = v[r];
    }
}

template <int LOGN>
static hipError_t launch_ntt_t(const DevTables &T, bool inverse, u64 *data, int npoly, int nrows, int mod_first,
                               hipStream_t s)
{
    using C = NttCfg<LOGN>;
    dim3 grid(nrows, npoly), block(C::T);
    size_t lds = sizeof(u64) * C::LDS_WORDS;
    if (inverse) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_rows_kernel<LOGN, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ntt_rows_kernel<LOGN, true>), grid, block, lds, s, T, data, nrows, mod_first);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ntt_rows_kernel<LOGN, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ntt_rows_kernel<LOGN, false>), grid, block, lds, s, T, data, nrows, mod_first);
    }
    return hipGetLastError();
}

#define HEFX_DISPATCH_LOGN(logn, CALL)     \
    switch (logn) {                        \
        case 10: return CALL(10);          \
        case 11: return CALL(11);          \
        case 12: return CALL(12);          \
        case 13: return CALL(13);          \
        case 14: return CALL(14);          \
        default: return hipErrorInvalidValue; \
    }

hipError_t launch_ntt(const DevTables &T, bool inverse, u64 *data, int npoly, int nrows, int mod_first,
                      hipStream_t s)
{
#define CALL(LN) launch_ntt_t<LN>(T, inverse, data, npoly, nrows, mod_first, s)
    HEFX_DISPATCH_LOGN(T.logn, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------
// K3/K4/K10: element-wise ops, two words (16 B) per lane per step, grid-stride.
// word w of a ciphertext batch: row = w / N, modulus j = row % L.  Plaintext operand: [L][N].
// ------------------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void elementwise_kernel(DevTables T, int L, int size, size_t total_pairs,
                                                          const ulonglong2 *__restrict__ a,
                                                          const ulonglong2 *__restrict__ b,
                                                          ulonglong2 *__restrict__ out, int *flag)
{
    const int logn = T.logn;
    const size_t pairs_per_row = (size_t)1 << (logn - 1);
    bool nonzero_beyond_c0 = false;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total_pairs;
         w += (size_t)gridDim.x * blockDim.x) {
        const size_t row = w >> (logn - 1);
        const int j = (int)(row % (size_t)L);
        const ModConst mc = T.mods[j];
        ulonglong2 x = a[w], r;
        if (OP == EW_ADD) {
            ulonglong2 y = b[w];
            r.x = addmod(x.x, y.x, mc.q);
            r.y = addmod(x.y, y.y, mc.q);
        } else if (OP == EW_SUB) {
            ulonglong2 y = b[w];
            r.x = submod(x.x, y.x, mc.q);
            r.y = submod(x.y, y.y, mc.q);
        } else if (OP == EW_NEG) {
            r.x = negmod(x.x, mc.q);
            r.y = negmod(x.y, mc.q);
        } else if (OP == EW_REDUCE) {
            r.x = barrett64(x.x, mc.q, mc.r1);
            r.y = barrett64(x.y, mc.q, mc.r1);
        } else {
            // plaintext operand: poly index within ciphertext is irrelevant, only (j, i)
            const size_t in_row = w & (pairs_per_row - 1);
            ulonglong2 y = b[(size_t)j * pairs_per_row + in_row];
            const int poly = (int)((row / (size_t)L) % (size_t)size);
            if (OP == EW_MULPLAIN) {
                r.x = mulmod(x.x, y.x, mc);
                r.y = mulmod(x.y, y.y, mc);
                if (poly > 0 && (r.x | r.y)) nonzero_beyond_c0 = true;
            } else {  // EW_ADDPLAIN: only c0 gets the plaintext
                if (poly == 0) {
                    r.x = addmod(x.x, y.x, mc.q);
                    r.y = addmod(x.y, y.y, mc.q);
                } else {
                    r = x;
                }
            }
        }
        out[w] = r;
    }
    if (OP == EW_MULPLAIN) {
        // transparent-ciphertext detection (SEAL is_transparent): flag[1] is set if ANY word of a poly
        // beyond c0 is non-zero; all writers store the same value.
        if (__any(nonzero_beyond_c0) && (threadIdx.x & 63) == 0) flag[1] = 1;
    }
}

hipError_t launch_elementwise(const DevTables &T, EwOp op, int L, int size, int count, const u64 *a,
                              const u64 *b, u64 *out, int *flag, hipStream_t s)
{
    const size_t n = (size_t)1 << T.logn;
    const size_t total_pairs = (size_t)count * size * L * n / 2;
    int blocks = (int)((total_pairs + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    const ulonglong2 *pa = reinterpret_cast<const ulonglong2 *>(a);
    const ulonglong2 *pb = reinterpret_cast<const ulonglong2 *>(b);
    ulonglong2 *po = reinterpret_cast<ulonglong2 *>(out);
#define LAUNCH(OPC) \
    hipLaunchKernelGGL((elementwise_kernel<OPC>), dim3(blocks), dim3(256), 0, s, T, L, size, total_pairs, pa, pb, po, flag)
    switch (op) {
        case EW_ADD: LAUNCH(EW_ADD); break;
        case EW_SUB: LAUNCH(EW_SUB); break;
        case EW_NEG: LAUNCH(EW_NEG); break;
        case EW_MULPLAIN: LAUNCH(EW_MULPLAIN); break;
        case EW_ADDPLAIN: LAUNCH(EW_ADDPLAIN); break;
        case EW_REDUCE: LAUNCH(EW_REDUCE); break;
    }
#undef LAUNCH
    return hipGetLastError();
}

// add_many: out = (accumulate ? out : 0) + sum_{i<n} in[i]   (n <= ADD_MANY_GROUP pointers by value)
__global__ __launch_bounds__(256) void add_many_kernel(DevTables T, int L, size_t total_pairs, PtrGroup g, int n,
                                                       int accumulate, ulonglong2 *__restrict__ out)
{
    const int logn = T.logn;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total_pairs;
         w += (size_t)gridDim.x * blockDim.x) {
        const size_t row = w >> (logn - 1);
        const u64 q = T.mods[(int)(row % (size_t)L)].q;
        ulonglong2 acc = accumulate ? out[w] : make_ulonglong2(0, 0);
        for (int i = 0; i < n; ++i) {
            ulonglong2 x = reinterpret_cast<const ulonglong2 *>(g.p[i])[w];
            acc.x = addmod(acc.x, x.x, q);
            acc.y = addmod(acc.y, x.y, q);
        }
        out[w] = acc;
    }
}

hipError_t launch_add_many(const DevTables &T, int L, int size, const PtrGroup &g, int n, bool accumulate,
                           u64 *out, hipStream_t s)
{
    const size_t total_pairs = (size_t)size * L * ((size_t)1 << T.logn) / 2;
    int blocks = (int)((total_pairs + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(add_many_kernel, dim3(blocks), dim3(256), 0, s, T, L, total_pairs, g, n, accumulate ? 1 : 0,
                       reinterpret_cast<ulonglong2 *>(out));
    return hipGetLastError();
}

// K3/K11: size-2 x size-2 tensor product: c0=a0b0, c1=a0b1+a1b0, c2=a1b1 (b==a gives square)
__global__ __launch_bounds__(256) void multiply_kernel(DevTables T, int L, size_t pairs_per_poly,
                                                       const ulonglong2 *__restrict__ a,
                                                       const ulonglong2 *__restrict__ b, ulonglong2 *__restrict__ out)
{
    const int logn = T.logn;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < pairs_per_poly;
         w += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(w >> (logn - 1));
        const ModConst mc = T.mods[j];
        ulonglong2 a0 = a[w], a1 = a[w + pairs_per_poly], b0 = b[w], b1 = b[w + pairs_per_poly];
        ulonglong2 c0, c1, c2;
        c0.x = mulmod(a0.x, b0.x, mc);
        c0.y = mulmod(a0.y, b0.y, mc);
        c2.x = mulmod(a1.x, b1.x, mc);
        c2.y = mulmod(a1.y, b1.y, mc);
        {
            u64 lo = 0, hi = 0;
            mac128(lo, hi, a0.x, b1.x);
            mac128(lo, hi, a1.x, b0.x);
            c1.x = barrett128(lo, hi, mc);
            lo = hi = 0;
            mac128(lo, hi, a0.y, b1.y);
            mac128(lo, hi, a1.y, b0.y);
            c1.y = barrett128(lo, hi, mc);
        }
        out[w] = c0;
        out[w + pairs_per_poly] = c1;
        out[w + 2 * pairs_per_poly] = c2;
    }
}

hipError_t launch_multiply(const DevTables &T, int L, const u64 *a, const u64 *b, u64 *out3, hipStream_t s)
{
    const size_t pairs = (size_t)L * ((size_t)1 << T.logn) / 2;
    int blocks = (int)((pairs + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(multiply_kernel, dim3(blocks), dim3(256), 0, s, T, L, pairs,
                       reinterpret_cast<const ulonglong2 *>(a), reinterpret_cast<const ulonglong2 *>(b),
                       reinterpret_cast<ulonglong2 *>(out3));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// K5/K6/K7: key switch, five launches per chunk of items.
// ------------------------------------------------------------------------------------------------

// (1) digit i of item b: Galois permutation -> keep NTT copy in x[b][i][i] -> INTT mod q_i -> d[b][i].
// The permutation is done through LDS (coalesced global read, LDS scatter by the table of g^-1, conflict-
// free LDS read) instead of an 8-byte global gather that touches one cache line per lane.  For rotations the
// extra blocks bx >= L write perm(c0[j]) into scratch p0[b][j]; the mod-down epilogue adds it in, so a rotation
// may run in place (c_out == c_in): c_in is fully consumed by this kernel.
template <int LOGN>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void ks_intt_digits_kernel(DevTables T, KsBatch B, int L, int relin,
                                                                         KsScratch S)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x, bx = blockIdx.x, b = blockIdx.y;
    const KsItem it = B.it[b];
    u64 v[16];
    if (bx >= L) {  // rotation only: p0[b][j] = perm_g(c_in[0][j])
        const int j = bx - L;
        const u64 *src = it.c_in + (size_t)j * C::N;
        u64 *dst = S.p0 + ((size_t)b * L + j) * C::N;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = src[C::idx_nat(t, r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[C::phys((int)it.perm[C::idx_nat(t, r)])] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[C::idx_nat(t, r)] = lds[C::phys(C::idx_nat(t, r))];
        return;
    }
    const int i = bx;
    const u64 *src = it.c_in + ((size_t)(relin ? 2 * L : L) + i) * C::N;
    if (it.perm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = src[C::idx_nat(t, r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[C::phys((int)it.perm[C::idx_nat(t, r)])] = v[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = lds[C::phys(C::idx_out(t, r))];
        // no barrier needed: the INTT core's first LDS writes go to exactly the words this thread just read
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = src[C::idx_out(t, r)];
    }
    u64 *xd = S.x + (((size_t)b * L + i) * (L + 1) + i) * C::N;
#pragma unroll
    for (int r = 0; r < 16; ++r) xd[C::idx_out(t, r)] = v[r];
    ntt_inv_core<LOGN>(v, lds, T.itw + (size_t)i * C::N, T.mods[i], t);
    u64 *dd = S.d + ((size_t)b * L + i) * C::N;
#pragma unroll
    for (int r = 0; r < 16; ++r) dd[C::idx_nat(t, r)] = v[r];
}

// (2) digit i -> modulus slot jj != i: x[b][i][jj] = NTT_m(d[b][i] mod m)
template <int LOGN>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void ks_ntt_digits_kernel(DevTables T, int L, KsScratch S)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x, b = blockIdx.y;
    const int i = blockIdx.x / L;
    int jj = blockIdx.x % L;
    if (jj >= i) ++jj;  // skip the diagonal; jj in [0, L], jj == L is the special prime
    const int m = jj < L ? jj : T.k - 1;
    const ModConst mc = T.mods[m];
    const u64 qi = T.mods[i].q;
    const u64 *dd = S.d + ((size_t)b * L + i) * C::N;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = dd[C::idx_nat(t, r)];
    if (qi > mc.q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = barrett64(v[r], mc.q, mc.r1);
    }
    ntt_fwd_core<LOGN>(v, lds, T.tw + (size_t)m * C::N, mc.q, t);
    u64 *xd = S.x + (((size_t)b * L + i) * (L + 1) + jj) * C::N;
#pragma unroll
    for (int r = 0; r < 16; ++r) xd[C::idx_out(t, r)] = v[r];
}

// (3) acc[b][c][jj] = sum_i x[b][i][jj] * key[i][c][m]  (128-bit lazy accumulation, one Barrett at the end)
__global__ __launch_bounds__(256) void ks_mac_kernel(DevTables T, KsBatch B, int L, KsScratch S)
{
    const int logn = T.logn;
    const size_t n = (size_t)1 << logn;
    const int jj = blockIdx.y, b = blockIdx.z;
    const int m = jj < L ? jj : T.k - 1;
    const ModConst mc = T.mods[m];
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // pair index within the row
    const KsItem it = B.it[b];
    u64 a0xl = 0, a0xh = 0, a0yl = 0, a0yh = 0, a1xl = 0, a1xh = 0, a1yl = 0, a1yh = 0;
    for (int i = 0; i < L; ++i) {
        const ulonglong2 x =
            reinterpret_cast<const ulonglong2 *>(S.x + (((size_t)b * L + i) * (L + 1) + jj) * n)[w];
        const u64 *kbase = it.key + ((size_t)i * 2 * T.k + m) * n;
        const ulonglong2 k0 = reinterpret_cast<const ulonglong2 *>(kbase)[w];
        const ulonglong2 k1 = reinterpret_cast<const ulonglong2 *>(kbase + (size_t)T.k * n)[w];
        mac128(a0xl, a0xh, x.x, k0.x);
        mac128(a0yl, a0yh, x.y, k0.y);
        mac128(a1xl, a1xh, x.x, k1.x);
        mac128(a1yl, a1yh, x.y, k1.y);
    }
    ulonglong2 r0, r1;
    r0.x = barrett128(a0xl, a0xh, mc);
    r0.y = barrett128(a0yl, a0yh, mc);
    r1.x = barrett128(a1xl, a1xh, mc);
    r1.y = barrett128(a1yl, a1yh, mc);
    reinterpret_cast<ulonglong2 *>(S.acc + (((size_t)b * 2 + 0) * (L + 1) + jj) * n)[w] = r0;
    reinterpret_cast<ulonglong2 *>(S.acc + (((size_t)b * 2 + 1) * (L + 1) + jj) * n)[w] = r1;
}

// (4) u[b][c] = (INTT_P(acc[b][c][P]) + floor(P/2)) mod P
template <int LOGN>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void ks_moddown_intt_kernel(DevTables T, int L, KsScratch S)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x, c = blockIdx.x, b = blockIdx.y;
    const int sp = T.k - 1;
    const ModConst mc = T.mods[sp];
    const u64 *src = S.acc + (((size_t)b * 2 + c) * (L + 1) + L) * C::N;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = src[C::idx_out(t, r)];
    ntt_inv_core<LOGN>(v, lds, T.itw + (size_t)sp * C::N, mc, t);
    const u64 half = mc.q >> 1;
    u64 *ud = S.u + ((size_t)b * 2 + c) * C::N;
#pragma unroll
    for (int r = 0; r < 16; ++r) ud[C::idx_nat(t, r)] = csub(v[r] + half, mc.q);
}

// (5) out[b][c][j] = (acc[b][c][j] - NTT_j((u mod q_j) - (P/2 mod q_j))) * P^-1  + add-in, optionally * pt
template <int LOGN>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void ks_moddown_finish_kernel(DevTables T, KsBatch B, int L,
                                                                            int relin, KsScratch S)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x, b = blockIdx.y;
    const int c = blockIdx.x / L, j = blockIdx.x % L;
    const int sp = T.k - 1;
    const ModConst mc = T.mods[j];
    const u64 q = mc.q;
    const u64 P = T.mods[sp].q;
    const u64 half_j = T.halfmod[(size_t)sp * T.k + j];
    const ulonglong2 pinv = T.invmod[(size_t)sp * T.k + j];
    const KsItem it = B.it[b];
    const u64 *ud = S.u + ((size_t)b * 2 + c) * C::N;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = ud[C::idx_nat(t, r)];  // all 16 loads in flight before any use
    (void)P;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = submod(barrett64(v[r], q, mc.r1), half_j, q);  // exact for P <= q too
    ntt_fwd_core<LOGN>(v, lds, T.tw + (size_t)j * C::N, q, t);
    // relinearisation adds (c0,c1) of the input; a rotation adds perm(c0), which kernel (1) left in S.p0.
    // All operand loads of a half (8 coefficients) are issued before any store so their latency overlaps.
    const u64 *__restrict__ acc = S.acc + (((size_t)b * 2 + c) * (L + 1) + j) * C::N;
    const u64 *__restrict__ addsrc =
        relin ? it.c_in + ((size_t)c * L + j) * C::N : S.p0 + ((size_t)b * L + j) * C::N;
    const bool has_add = relin || c == 0;
    const u64 *__restrict__ pt = it.pt ? it.pt + (size_t)j * C::N : nullptr;
    u64 *__restrict__ dst = it.c_out + ((size_t)c * L + j) * C::N;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        u64 a[8], sadd[8], pp[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = C::idx_out(t, 8 * h + r);
            a[r] = acc[idx];
            sadd[r] = has_add ? addsrc[idx] : 0;
            pp[r] = pt ? pt[idx] : 0;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = C::idx_out(t, 8 * h + r);
            u64 x = submod(a[r], v[8 * h + r], q);
            x = csub(shoup_lazy(x, pinv.x, pinv.y, q), q);
            x = addmod(x, sadd[r], q);
            if (pt) x = mulmod(x, pp[r], mc);
            dst[idx] = x;
        }
    }
}

template <int LOGN>
static hipError_t launch_keyswitch_chunk_t(const DevTables &T, int L, int n, const KsBatch &batch, bool relin,
                                           const KsScratch &scr, hipStream_t s, hipEvent_t *ev)
{
    using C = NttCfg<LOGN>;
    const size_t lds = sizeof(u64) * C::LDS_WORDS;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks_intt_digits_kernel<LOGN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks_ntt_digits_kernel<LOGN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks_moddown_intt_kernel<LOGN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ks_moddown_finish_kernel<LOGN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int rl = relin ? 1 : 0;
    // optional profiling: ev[0..5] bracket the five launches (hefx_profile_*), recorded on the same stream
#define HEFX_EV(i) \
    if (ev) (void)hipEventRecord(ev[i], s)
    HEFX_EV(0);
    hipLaunchKernelGGL((ks_intt_digits_kernel<LOGN>), dim3(relin ? L : 2 * L, n), dim3(C::T), lds, s, T, batch, L, rl, scr);
    HEFX_EV(1);
    hipLaunchKernelGGL((ks_ntt_digits_kernel<LOGN>), dim3(L * L, n), dim3(C::T), lds, s, T, L, scr);
    HEFX_EV(2);
    hipLaunchKernelGGL(ks_mac_kernel, dim3(C::N / 2 / 256, L + 1, n), dim3(256), 0, s, T, batch, L, scr);
    HEFX_EV(3);
    hipLaunchKernelGGL((ks_moddown_intt_kernel<LOGN>), dim3(2, n), dim3(C::T), lds, s, T, L, scr);
    HEFX_EV(4);
    hipLaunchKernelGGL((ks_moddown_finish_kernel<LOGN>), dim3(2 * L, n), dim3(C::T), lds, s, T, batch, L, rl, scr);
    HEFX_EV(5);
#undef HEFX_EV
    return hipGetLastError();
}

hipError_t launch_keyswitch_chunk(const DevTables &T, int L, int n, const KsBatch &batch, bool relin,
                                  const KsScratch &scr, hipStream_t s, hipEvent_t *ev)
{
#define CALL(LN) launch_keyswitch_chunk_t<LN>(T, L, n, batch, relin, scr, s, ev)
    HEFX_DISPATCH_LOGN(T.logn, CALL)
#undef CALL
}

// ------------------------------------------------------------------------------------------------
// K8: rescale_to_next, SEAL 3.4.x floor variant (App. A.9): per poly 1 INTT + (L-1) NTT.
// ------------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void rs_intt_kernel(DevTables T, int L, const u64 *in, u64 *d)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x;
    const size_t poly = blockIdx.x;  // over count*size polys
    const u64 *src = in + (poly * L + (L - 1)) * C::N;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = src[C::idx_out(t, r)];
    ntt_inv_core<LOGN>(v, lds, T.itw + (size_t)(L - 1) * C::N, T.mods[L - 1], t);
    u64 *dd = d + poly * C::N;
#pragma unroll
    for (int r = 0; r < 16; ++r) dd[C::idx_nat(t, r)] = v[r];
}

template <int LOGN>
__global__ __launch_bounds__(NttCfg<LOGN>::T, 4) void rs_finish_kernel(DevTables T, int L, const u64 *in,
                                                                    const u64 *d, u64 *out)
{
    using C = NttCfg<LOGN>;
    extern __shared__ __align__(16) u64 lds[];
    const int t = threadIdx.x, j = blockIdx.x;
    const size_t poly = blockIdx.y;
    const ModConst mc = T.mods[j];
    const u64 q = mc.q, ql = T.mods[L - 1].q;
    const ulonglong2 qinv = T.invmod[(size_t)(L - 1) * T.k + j];
    const u64 *dd = d + poly * C::N;
    u64 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = dd[C::idx_nat(t, r)];
    if (ql > q) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = barrett64(v[r], q, mc.r1);
    }
    ntt_fwd_core<LOGN>(v, lds, T.tw + (size_t)j * C::N, q, t);
    const u64 *src = in + (poly * L + j) * C::N;
    u64 *dst = out + (poly * (L - 1) + j) * C::N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int idx = C::idx_out(t, r);
        u64 x = submod(src[idx], v[r], q);
        dst[idx] = csub(shoup_lazy(x, qinv.x, qinv.y, q), q);
    }
}

template <int LOGN>
static hipError_t launch_rescale_t(const DevTables &T, int L, int size, int count, const u64 *in, u64 *out,
                                   u64 *scratch_d, hipStream_t s)
{
    using C = NttCfg<LOGN>;
    const size_t lds = sizeof(u64) * C::LDS_WORDS;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rs_intt_kernel<LOGN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rs_finish_kernel<LOGN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int polys = size * count;
    hipLaunchKernelGGL((rs_intt_kernel<LOGN>), dim3(polys), dim3(C::T), lds, s, T, L, in, scratch_d);
    hipLaunchKernelGGL((rs_finish_kernel<LOGN>), dim3(L - 1, polys), dim3(C::T), lds, s, T, L, in, scratch_d, out);
    return hipGetLastError();
}

hipError_t launch_rescale(const DevTables &T, int L, int size, int count, const u64 *in, u64 *out, u64 *scratch_d,
                          hipStream_t s)
{
#define CALL(LN) launch_rescale_t<LN>(T, L, size, count, in, out, scratch_d, s)
    HEFX_DISPATCH_LOGN(T.logn, CALL)
#undef CALL
}

}  // namespace hefx
