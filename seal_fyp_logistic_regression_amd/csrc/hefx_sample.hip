// hefx_sample.hip -- randomness and the encrypt / decrypt arithmetic on the GPU (reference
// call sites linear_transformation2.cpp:344-350, logistic_regression_ckks.cpp:362-381 -- the per-iteration
// decrypt -> re-encrypt refresh of the LR loop).
//
// Sampling is counter mode: every random 64-bit word is a pure function of (256-bit key, stream id, position)
// taken from the ChaCha20 keystream (RFC 7539 block function; 64-bit block counter in state words 12-13, 64-bit
// stream id in words 14-15), so threads draw independently and the CPU oracle (the checker under oracle/,
// orc_sample_*) reproduces the same bits in a plain loop.  SEAL seeds its own generator from random_device, so
// random draws are inputs of the parity chain, not outputs; what is kept from SEAL 3.4.5 (App. A.11) are the
// DISTRIBUTIONS: uniform mod q by rejection, ternary {-1,0,1}, clipped normal sigma 3.2 / bound 19.2 truncated
// toward zero (drawn here by exact inverse-CDF on a 64-bit word against a 39-entry threshold table).
//   position -> block counter = attempt << 48 | row << 16 | (index >> 3), word (index & 7) of that block.
// One thread = one ChaCha block = 8 consecutive coefficients.
#include "hefx_internal.h"

namespace hefx {

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

#define HEFX_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);

__device__ __forceinline__ void chacha20_block(const SampleKey &key, u64 counter, u64 nonce, u64 (&out)[8])
{
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.w[0], key.w[1], key.w[2], key.w[3],
                      key.w[4], key.w[5], key.w[6], key.w[7], (uint32_t)counter, (uint32_t)(counter >> 32),
                      (uint32_t)nonce, (uint32_t)(nonce >> 32)};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#pragma unroll 2
    for (int r = 0; r < 10; ++r) {
        HEFX_QR(x[0], x[4], x[8], x[12]) HEFX_QR(x[1], x[5], x[9], x[13])
        HEFX_QR(x[2], x[6], x[10], x[14]) HEFX_QR(x[3], x[7], x[11], x[15])
        HEFX_QR(x[0], x[5], x[10], x[15]) HEFX_QR(x[1], x[6], x[11], x[12])
        HEFX_QR(x[2], x[7], x[8], x[13]) HEFX_QR(x[3], x[4], x[9], x[14])
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) out[w] = (u64)(x[2 * w] + s[2 * w]) | ((u64)(x[2 * w + 1] + s[2 * w + 1]) << 32);
}

// words r[0..7] of block (row, blk) where every word satisfies r < bound; rejected words are redrawn from the
// blocks of attempt 1, 2, ... at the same position
__device__ __forceinline__ void draw8(const SampleKey &key, u64 stream, u64 row, u64 blk, u64 bound, u64 (&r)[8])
{
    chacha20_block(key, (row << 16) | blk, stream, r);
    bool bad = false;
#pragma unroll
    for (int w = 0; w < 8; ++w) bad |= r[w] >= bound;
    for (u64 attempt = 1; bad; ++attempt) {
        u64 t[8];
        chacha20_block(key, (attempt << 48) | (row << 16) | blk, stream, t);
        bad = false;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            if (r[w] >= bound) r[w] = t[w];
            bad |= r[w] >= bound;
        }
    }
}

__device__ __forceinline__ void store8(u64 *dst, const u64 (&v)[8])
{
#pragma unroll
    for (int w = 0; w < 8; w += 2) *reinterpret_cast<ulonglong2 *>(dst + w) = make_ulonglong2(v[w], v[w + 1]);
}

// grid (N/8/256, rows): out[row][8*blk + w] uniform in [0, q_row)
__global__ __launch_bounds__(256) void sample_uniform_kernel(DevTables T, SampleKey key, u64 stream, int nrows,
                                                             int mod_first, u64 *out)
{
    const size_t n = (size_t)1 << T.logn;
    const u64 blk = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n / 8) return;
    const u64 row = blockIdx.y;
    const ModConst mc = T.mods[mod_first + (int)(row % nrows)];
    const u64 bound = (~(u64)0 / mc.q) * mc.q;
    u64 r[8];
    draw8(key, stream, row, blk, bound, r);
#pragma unroll
    for (int w = 0; w < 8; ++w) r[w] = barrett64(r[w], mc.q, mc.r1);
    store8(out + row * n + blk * 8, r);
}

// grid (N/8/256, npoly): one signed draw per coefficient, written to every row as its residue.
// NOISE = false: ternary; true: clipped normal via the threshold table
// stride != 0 (hefx_encrypt_batch): polynomial p is polynomial 0 of sub-stream `stream + p * stride` -- the words n single
// calls with those stream ids would draw
template <bool NOISE>
__global__ __launch_bounds__(256) void sample_small_kernel(DevTables T, SampleKey key, NoiseTable tab, u64 stream0,
                                                           u64 stride, int nrows, int mod_first, u64 *out)
{
    const size_t n = (size_t)1 << T.logn;
    const u64 blk = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n / 8) return;
    const u64 slot = blockIdx.y;  // where the polynomial is stored
    const u64 stream = stream0 + slot * stride, poly = stride ? 0 : slot;
    u64 r[8];
    int v[8];
    if (NOISE) {
        chacha20_block(key, (poly << 16) | blk, stream, r);  // inverse CDF: every word maps to a value, no rejection
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            int k = -19;
            for (int e = 0; e < 38; ++e) k += r[w] >= tab.t[e];
            v[w] = k;
        }
    } else {
        draw8(key, stream, poly, blk, 0xFFFFFFFFFFFFFFFFull, r);
#pragma unroll
        for (int w = 0; w < 8; ++w) v[w] = (int)(r[w] % 3) - 1;
    }
    for (int j = 0; j < nrows; ++j) {
        const u64 q = T.mods[mod_first + j].q;
        u64 o[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) o[w] = v[w] < 0 ? q - (u64)(-v[w]) : (u64)v[w];
        store8(out + (slot * nrows + j) * n + blk * 8, o);
    }
}

// out[c][j] = pk[c][j] * u[j] + e[c][j] (+ plain[j] for c = 0); pk rows have stride k, everything NTT form
__global__ __launch_bounds__(256) void encrypt_combine_kernel(DevTables T, int L, const u64 *__restrict__ pk,
                                                              const u64 *__restrict__ u, const u64 *__restrict__ e,
                                                              const u64 *__restrict__ plain, u64 *__restrict__ out)
{
    const size_t n = (size_t)1 << T.logn;
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int j = blockIdx.y, c = blockIdx.z;
    const ModConst mc = T.mods[j];
    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(pk + ((size_t)c * T.k + j) * n + w);
    const ulonglong2 uu = *reinterpret_cast<const ulonglong2 *>(u + (size_t)j * n + w);
    const ulonglong2 ee = *reinterpret_cast<const ulonglong2 *>(e + ((size_t)c * L + j) * n + w);
    ulonglong2 r;
    r.x = addmod(mulmod(a.x, uu.x, mc), ee.x, mc.q);
    r.y = addmod(mulmod(a.y, uu.y, mc), ee.y, mc.q);
    if (c == 0 && plain) {
        const ulonglong2 p = *reinterpret_cast<const ulonglong2 *>(plain + (size_t)j * n + w);
        r.x = addmod(r.x, p.x, mc.q);
        r.y = addmod(r.y, p.y, mc.q);
    }
    *reinterpret_cast<ulonglong2 *>(out + ((size_t)c * L + j) * n + w) = r;
}

// m encryptions at once: u [m][L][N], e [2][m][L][N], tab = m plaintext pointers (null: encryption of zero) | m outputs
__global__ __launch_bounds__(256) void encrypt_combine_table_kernel(DevTables T, int L, int m, const u64 *__restrict__ pk,
                                                                    const u64 *__restrict__ u, const u64 *__restrict__ e,
                                                                    const u64 *const *__restrict__ tab)
{
    const size_t n = (size_t)1 << T.logn;
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int j = blockIdx.y, item = blockIdx.z >> 1, c = blockIdx.z & 1;
    const ModConst mc = T.mods[j];
    const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(pk + ((size_t)c * T.k + j) * n + w);
    const ulonglong2 uu = *reinterpret_cast<const ulonglong2 *>(u + ((size_t)item * L + j) * n + w);
    const ulonglong2 ee = *reinterpret_cast<const ulonglong2 *>(e + (((size_t)c * m + item) * L + j) * n + w);
    ulonglong2 r;
    r.x = addmod(mulmod(a.x, uu.x, mc), ee.x, mc.q);
    r.y = addmod(mulmod(a.y, uu.y, mc), ee.y, mc.q);
    const u64 *plain = tab[item];
    if (c == 0 && plain) {
        const ulonglong2 p = gld16(plain + (size_t)j * n + w);
        r.x = addmod(r.x, p.x, mc.q);
        r.y = addmod(r.y, p.y, mc.q);
    }
    gst16(const_cast<u64 *>(tab[m + item]) + ((size_t)c * L + j) * n + w, r);
}

// out[j] = sum_p ct[p][j] * s[j]^p (Horner from the top), any size >= 1
__global__ __launch_bounds__(256) void decrypt_kernel(DevTables T, int L, int size, const u64 *__restrict__ ct,
                                                      const u64 *__restrict__ sk, u64 *__restrict__ out)
{
    const size_t n = (size_t)1 << T.logn;
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int j = blockIdx.y;
    const ModConst mc = T.mods[j];
    const ulonglong2 s = *reinterpret_cast<const ulonglong2 *>(sk + (size_t)j * n + w);
    ulonglong2 acc = *reinterpret_cast<const ulonglong2 *>(ct + ((size_t)(size - 1) * L + j) * n + w);
    for (int p = size - 2; p >= 0; --p) {
        const ulonglong2 cp = *reinterpret_cast<const ulonglong2 *>(ct + ((size_t)p * L + j) * n + w);
        acc.x = addmod(mulmod(acc.x, s.x, mc), cp.x, mc.q);
        acc.y = addmod(mulmod(acc.y, s.y, mc), cp.y, mc.q);
    }
    *reinterpret_cast<ulonglong2 *>(out + (size_t)j * n + w) = acc;
}

// Key-switching key for the secret new_sk under sk (App. A.11), digit i, key-level row m:
//   out[i][1][m] = a_i[m] (uniform),  out[i][0][m] = -(a_i[m]*sk[m] + e_i[m]) + [m == i] (P mod q_i) * new_sk[i]
// a: uniform [k-1][k][N]; e: noise [k-1][k][N] already in NTT form; everything NTT form, SEAL's key layout.
__global__ __launch_bounds__(256) void keygen_combine_kernel(DevTables T, const u64 *__restrict__ sk,
                                                             const u64 *__restrict__ new_sk,
                                                             const u64 *__restrict__ a, const u64 *__restrict__ e,
                                                             u64 *__restrict__ out)
{
    const size_t n = (size_t)1 << T.logn;
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const int m = blockIdx.y, i = blockIdx.z, k = T.k;
    const ModConst mc = T.mods[m];
    const ulonglong2 av = *reinterpret_cast<const ulonglong2 *>(a + ((size_t)i * k + m) * n + w);
    const ulonglong2 ev = *reinterpret_cast<const ulonglong2 *>(e + ((size_t)i * k + m) * n + w);
    const ulonglong2 sv = *reinterpret_cast<const ulonglong2 *>(sk + (size_t)m * n + w);
    ulonglong2 c0;
    c0.x = negmod(addmod(mulmod(av.x, sv.x, mc), ev.x, mc.q), mc.q);
    c0.y = negmod(addmod(mulmod(av.y, sv.y, mc), ev.y, mc.q), mc.q);
    if (m == i) {
        const u64 f = barrett64(T.mods[k - 1].q, mc.q, mc.r1);
        const ulonglong2 ns = *reinterpret_cast<const ulonglong2 *>(new_sk + (size_t)m * n + w);
        c0.x = addmod(c0.x, mulmod(ns.x, f, mc), mc.q);
        c0.y = addmod(c0.y, mulmod(ns.y, f, mc), mc.q);
    }
    *reinterpret_cast<ulonglong2 *>(out + (((size_t)i * 2 + 0) * k + m) * n + w) = c0;
    *reinterpret_cast<ulonglong2 *>(out + (((size_t)i * 2 + 1) * k + m) * n + w) = av;
}

// out[p][j][w] = in[p][j][perm[w]]: the NTT-domain automorphism of plain polynomials (s(X^g) for Galois keys)
__global__ __launch_bounds__(256) void galois_permute_kernel(DevTables T, const uint32_t *__restrict__ perm,
                                                             const u64 *__restrict__ in, u64 *__restrict__ out)
{
    const size_t n = (size_t)1 << T.logn;
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    const size_t row = blockIdx.y;
    const uint2 pi = *reinterpret_cast<const uint2 *>(perm + w);
    const u64 *__restrict__ src = in + row * n;
    *reinterpret_cast<ulonglong2 *>(out + row * n + w) = make_ulonglong2(src[pi.x], src[pi.y]);
}

hipError_t launch_keygen_combine(const DevTables &T, const u64 *sk, const u64 *new_sk, const u64 *a, const u64 *e,
                                 u64 *out, hipStream_t s)
{
    const int n2 = (1 << T.logn) / 2;
    hipLaunchKernelGGL(keygen_combine_kernel, dim3((n2 + 255) / 256, T.k, T.k - 1), dim3(256), 0, s, T, sk, new_sk, a,
                       e, out);
    return hipGetLastError();
}

hipError_t launch_galois_permute(const DevTables &T, const uint32_t *perm, const u64 *in, int rows, u64 *out,
                                 hipStream_t s)
{
    const int n2 = (1 << T.logn) / 2;
    hipLaunchKernelGGL(galois_permute_kernel, dim3((n2 + 255) / 256, rows), dim3(256), 0, s, T, perm, in, out);
    return hipGetLastError();
}

hipError_t launch_sample(const DevTables &T, int mode, const SampleKey &key, const NoiseTable &tab, u64 stream,
                         int npoly, int nrows, int mod_first, u64 *out, hipStream_t s, u64 stream_stride)
{
    const int n8 = (1 << T.logn) / 8;
    const dim3 block(256);
    if (mode == SAMPLE_UNIFORM)
        hipLaunchKernelGGL(sample_uniform_kernel, dim3((n8 + 255) / 256, npoly * nrows), block, 0, s, T, key, stream,
                           nrows, mod_first, out);
    else if (mode == SAMPLE_TERNARY)
        hipLaunchKernelGGL((sample_small_kernel<false>), dim3((n8 + 255) / 256, npoly), block, 0, s, T, key, tab,
                           stream, stream_stride, nrows, mod_first, out);
    else
        hipLaunchKernelGGL((sample_small_kernel<true>), dim3((n8 + 255) / 256, npoly), block, 0, s, T, key, tab, stream,
                           stream_stride, nrows, mod_first, out);
    return hipGetLastError();
}

hipError_t launch_encrypt_combine(const DevTables &T, int L, const u64 *pk, const u64 *u, const u64 *e,
                                  const u64 *plain, u64 *out, hipStream_t s)
{
    const int n2 = (1 << T.logn) / 2;
    hipLaunchKernelGGL(encrypt_combine_kernel, dim3((n2 + 255) / 256, L, 2), dim3(256), 0, s, T, L, pk, u, e, plain,
                       out);
    return hipGetLastError();
}

hipError_t launch_encrypt_combine_table(const DevTables &T, int L, int m, const u64 *pk, const u64 *u, const u64 *e,
                                        const u64 *const *d_tab, hipStream_t s)
{
    const int n2 = (1 << T.logn) / 2;
    hipLaunchKernelGGL(encrypt_combine_table_kernel, dim3((n2 + 255) / 256, L, 2 * m), dim3(256), 0, s, T, L, m, pk, u, e,
                       d_tab);
    return hipGetLastError();
}

hipError_t launch_decrypt(const DevTables &T, int L, int size, const u64 *ct, const u64 *sk, u64 *out, hipStream_t s)
{
    const int n2 = (1 << T.logn) / 2;
    hipLaunchKernelGGL(decrypt_kernel, dim3((n2 + 255) / 256, L), dim3(256), 0, s, T, L, size, ct, sk, out);
    return hipGetLastError();
}

// HIP loads a translation unit's code object at its first kernel launch (milliseconds); hefx_context_create pays
// that once, up front, instead of the first encode / rotation / encryption of a program.
__global__ void warm_sample_kernel() {}
hipError_t warm_sample(hipStream_t s)
{
    hipLaunchKernelGGL(warm_sample_kernel, dim3(1), dim3(64), 0, s);
    return hipGetLastError();
}

}  // namespace hefx
