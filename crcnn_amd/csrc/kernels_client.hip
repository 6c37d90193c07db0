// kernels_client.hip -- BFV public-key encryption on gfx950 (the step in front of the evaluation path).
//
// Encryptor::encrypt (SEAL encryptor.cpp:71-134) per plaintext m:   u <- uniform {-1,0,1}^n,  e1, e2 <- clipped normal (sigma 3.19,
// cut at 6 sigma, truncated to an integer: util/globals.cpp:13-15, util/clipnormal.cpp),
//     c0 = pk0 * u + e1 + Delta*m (+ q mod t on the upper half, evaluator.cpp:1168-1191),   c1 = pk1 * u + e2.
// 784 of these per image dominate the client's latency in the reference (2.74 s/image); here the polynomial products are
// the row NTT of kernels.hip and the sampling is ChaCha20 in counter mode (chacha.h), one (ciphertext, coefficient) stream per lane.
// The reference draws from std::random_device, so there are no reference bits to match: the ciphertexts are checked by
// decrypting them (tests/test_gpu_ops.py) and by their noise budget against the CPU encryptor's.
#include "kernels.h"
#include "chacha.h"

// One lane = one (ciphertext, coefficient): its own ChaCha20 stream, nonce = (ciphertext stream id, domain | coefficient).  Block 0
// serves the ternary sample (2-bit fields of its first 64 bits, rejecting 3: all 32 fields equal to 3 has probability 2^-64 and
// falls back to 0) and the first try of both noise terms (4 words each); a rejected normal (|g| > 6 sigma, p = 2e-9) draws
// from block 1, 2, ...
__device__ __forceinline__ double unit53(u32 lo, u32 hi) { return ((double)((((u64)hi << 32) | lo) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// U: [count][k][n] ternary polynomial in RNS form (coefficient domain);  E: [count][2][n] signed noise
__global__ void __launch_bounds__(256) enc_sample_kernel(u64 *U, signed char *E, const ModParams *mods, int n, int k, ChaChaKey key, u64 stream_base)
{
    const int sblocks = n / blockDim.x;
    const size_t m = blockIdx.x / sblocks;
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const u64 sid = stream_base + m;
    const u32 n0 = (u32)sid, n1 = (u32)(sid >> 32), n2 = ((u32)CHACHA_DOM_ENC_DEV << 24) | (u32)s;
    u32 b[16];
    chacha20_block(key, 0, n0, n1, n2, b);
    u32 v = 0;
    {
        const u64 w = (u64)b[0] | ((u64)b[1] << 32);
        for (int j = 0; j < 32; j++) { const u32 f = (u32)(w >> (2 * j)) & 3u; if (f != 3u) { v = f; break; } }
    }
    for (int i = 0; i < k; i++) U[(m * k + i) * (size_t)n + s] = v == 0 ? 0 : (v == 1 ? 1 : mods[i].q - 1);
    const double sigma = 3.19, lim = 6 * sigma;
    for (int p = 0; p < 2; p++) {
        double g = sigma * sqrt(-2.0 * log(unit53(b[2 + 4 * p], b[3 + 4 * p]))) * cos(6.283185307179586 * unit53(b[4 + 4 * p], b[5 + 4 * p]));
        for (u32 ctr = 1; fabs(g) > lim; ctr++) {
            u32 r[16];
            chacha20_block(key, ctr, n0, n1, n2, r);
            g = sigma * sqrt(-2.0 * log(unit53(r[4 * p], r[4 * p + 1]))) * cos(6.283185307179586 * unit53(r[4 * p + 2], r[4 * p + 3]));
        }
        E[(m * 2 + p) * (size_t)n + s] = (signed char)(int)g;
    }
}

// ct[m][p][i][s] = U_ntt[m][i][s] * pk[p][i][s]
__global__ void __launch_bounds__(256) enc_mulpk_kernel(const u64 *U, const u64 *pk, u64 *ct, const ModParams *mods, int n, int k)
{
    const size_t row = blockIdx.x;                // m*2k + p*k + i
    const int i = (int)(row % k), p = (int)((row / k) % 2);
    const size_t m = row / (2 * (size_t)k);
    const ModParams md = mods[i];
    const u64 *u = U + (m * k + i) * (size_t)n, *key = pk + ((size_t)p * k + i) * n;
    u64 *dst = ct + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) dst[s] = mulmod(u[s], key[s], md);
}

// ct[m][p][i][s] += e_p[s]  (+ Delta*m[s] on poly 0)
__global__ void __launch_bounds__(256) enc_finish_kernel(u64 *ct, const signed char *E, const u64 *plain, const ModParams *mods, int n, int k, PlainParams pp)
{
    const size_t row = blockIdx.x;
    const int i = (int)(row % k), p = (int)((row / k) % 2);
    const size_t m = row / (2 * (size_t)k);
    const ModParams md = mods[i];
    const signed char *e = E + (m * 2 + p) * (size_t)n;
    const u64 *pl = plain + m * (size_t)n;
    u64 *dst = ct + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const int ev = e[s];
        u64 v = addmod(dst[s], ev >= 0 ? (u64)ev : md.q - (u64)(-ev), md.q);
        if (p == 0) {
            const u64 c = pl[s];
            u64 lo, hi; mul64wide(pp.delta[i], c, lo, hi);
            if (c >= pp.threshold) { const u64 l2 = lo + pp.uhi[i]; hi += (l2 < lo); lo = l2; }
            v = addmod(v, barrett128(lo, hi, md), md.q);
        }
        dst[s] = v;
    }
}

size_t k_encrypt_work_words(const crc_ctx *c, size_t cnt)
{
    return cnt * (size_t)c->n * c->k + (cnt * 2 * (size_t)c->n + 7) / 8;       // U + E
}

int k_encrypt(crc_ctx *c, const u64 *pk, const u64 *plain, size_t cnt, const ChaChaKey &key, u64 stream_base, u64 *ct, u64 *work, hipStream_t st)
{
    if (cnt == 0) return CRC_OK;
    const int n = c->n, k = c->k;
    u64 *U = work; signed char *E = reinterpret_cast<signed char *>(U + cnt * (size_t)n * k);
    const int threads = n < 256 ? n : 256, sblocks = n / threads;
    hipLaunchKernelGGL(enc_sample_kernel, dim3((unsigned)(cnt * sblocks)), dim3(threads), 0, st, U, E, c->d_mods, n, k, key, stream_base);
    HIPCHK(hipGetLastError());
    int rc;
    if ((rc = k_ntt_ct(c, false, U, U, cnt, 1, false, st, nullptr, 0, 0, 0))) return rc;
    hipLaunchKernelGGL(enc_mulpk_kernel, dim3((unsigned)(cnt * 2 * k)), dim3(256), 0, st, U, pk, ct, c->d_mods, n, k);
    HIPCHK(hipGetLastError());
    if ((rc = k_ntt_ct(c, true, ct, ct, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc;
    hipLaunchKernelGGL(enc_finish_kernel, dim3((unsigned)(cnt * 2 * k)), dim3(256), 0, st, ct, E, plain, c->d_mods, n, k, c->plain);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
