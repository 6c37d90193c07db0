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
#include <cmath>

// One lane = one (ciphertext, coefficient PAIR): its own ChaCha20 stream, nonce = (ciphertext stream id, domain | even coefficient index), ONE block (round 5;
// one block per coefficient and two Box-Muller draws before -- sampling was 40 % of an encryption).  Per coefficient six words: 64 bits for the ternary sample
// (2-bit fields, rejecting 3: all 32 fields equal to 3 has probability 2^-64 and falls back to 0) and 64 bits for the magnitude of each noise term; the noise
// signs are bits of word 12.
//
// The noise law, exactly: SEAL draws g ~ N(0, 3.19^2), redraws while |g| > 6 sigma = 19.14 (util/clipnormal.h) and keeps static_cast<int64_t>(g)
// (encryptor.cpp:237-240) -- an integer in [-19, 19] with  P(0) = P(|g| < 1) / Z,  P(+-a) = P(a <= g < a + 1) / Z (a < 19),  P(+-19) = P(19 <= g <= 19.14) / Z,
// Z = P(|g| <= 19.14).  Sampling that integer directly needs no logarithm, cosine or rejection loop: the magnitude is the number of thresholds
// T_a = floor(2^64 P(|e| <= a)) (a = 0..18; enc_cdt, computed once on the host in long double) that a uniform 64-bit word reaches -- the law above to 2^-63.
struct EncCdt { u64 t[19]; };
__device__ __forceinline__ int cdt_noise(u32 lo, u32 hi, u32 sign, const EncCdt &T)
{
    const u64 x = ((u64)hi << 32) | lo;
    int a = 0;
#pragma unroll
    for (int j = 0; j < 19; j++) a += x >= T.t[j] ? 1 : 0;
    return sign ? -a : a;
}
__device__ __forceinline__ u32 ternary_field(u32 lo, u32 hi)
{
    const u64 w = (u64)lo | ((u64)hi << 32);
    u32 v = 0;
    for (int j = 0; j < 32; j++) { const u32 f = (u32)(w >> (2 * j)) & 3u; if (f != 3u) { v = f; break; } }
    return v;
}

// U: [count][k][n] ternary polynomial in RNS form (coefficient domain).  ROWS = false: E [count][2][n] signed noise bytes (the coefficient-form pipeline adds
// them after its inverse transforms); ROWS = true: the rows e1 + Delta m (+ q mod t on the upper half, evaluator.cpp:1168-1191) and e2 of every ciphertext, as
// residues in ct [count][2][k][n] -- what the NTT-form pipeline transforms next
// COMPACT: `plain` holds the 96-word compact plaintexts of the fractional encoder (crc_plain_expand's layout) instead of dense rows
template <bool ROWS, bool COMPACT = false>
__global__ void __launch_bounds__(256) enc_sample_kernel(u64 *U, signed char *E, u64 *ct, const u64 *plain, const ModParams *mods, int n, int k, ChaChaKey key,
                                                         u64 stream_base, EncCdt cdt, PlainParams pp)
{
    const int pairs = n >> 1, pblocks = (pairs + (int)blockDim.x - 1) / (int)blockDim.x;
    const size_t m = blockIdx.x / pblocks;
    const int pr = (blockIdx.x % pblocks) * blockDim.x + threadIdx.x;
    if (pr >= pairs) return;
    const int s = 2 * pr;
    const u64 sid = stream_base + m;
    u32 b[16];
    chacha20_block(key, 0, (u32)sid, (u32)(sid >> 32), ((u32)CHACHA_DOM_ENC_DEV << 24) | (u32)s, b);
    u32 tv[2]; int e[2][2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const u32 *w = b + 6 * c;
        tv[c] = ternary_field(w[0], w[1]);
        e[c][0] = cdt_noise(w[2], w[3], (b[12] >> (2 * c)) & 1u, cdt);
        e[c][1] = cdt_noise(w[4], w[5], (b[12] >> (2 * c + 1)) & 1u, cdt);
    }
    for (int i = 0; i < k; i++) {
        const u64 qm1 = mods[i].q - 1;
        *reinterpret_cast<ulonglong2 *>(U + (m * k + i) * (size_t)n + s) = ulonglong2{tv[0] == 0 ? 0 : (tv[0] == 1 ? 1 : qm1), tv[1] == 0 ? 0 : (tv[1] == 1 ?
            1 : qm1)};
    }
    if (!ROWS) {
#pragma unroll
        for (int p = 0; p < 2; p++) *reinterpret_cast<char2 *>(E + (m * 2 + p) * (size_t)n + s) = char2{(signed char)e[0][p], (signed char)e[1][p]};
    } else {
        ulonglong2 pl = make_ulonglong2(0, 0);
        if (!COMPACT) pl = *reinterpret_cast<const ulonglong2 *>(plain + m * (size_t)n + s);
        else if (s < CRC_PLAIN_COMPACT_LOW) pl = *reinterpret_cast<const ulonglong2 *>(plain + m * (size_t)CRC_PLAIN_COMPACT_WORDS + s);
        else if (s >= n - CRC_PLAIN_COMPACT_HIGH)
            pl = *reinterpret_cast<const ulonglong2 *>(plain + m * (size_t)CRC_PLAIN_COMPACT_WORDS + CRC_PLAIN_COMPACT_LOW + (s - (n - CRC_PLAIN_COMPACT_HIGH)));
        const u64 pc[2] = {pl.x, pl.y};
        for (int i = 0; i < k; i++) {
            const ModParams md = mods[i];
            u64 r[2][2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
#pragma unroll
                for (int p = 0; p < 2; p++) r[c][p] = e[c][p] >= 0 ? (u64)e[c][p] : md.q - (u64)(-e[c][p]);
                u64 lo, hi; mul64wide(pp.delta[i], pc[c], lo, hi);
                if (pc[c] >= pp.threshold) { const u64 l2 = lo + pp.uhi[i]; hi += (l2 < lo); lo = l2; }
                r[c][0] = addmod(r[c][0], barrett128(lo, hi, md), md.q);
            }
#pragma unroll
            for (int p = 0; p < 2; p++) *reinterpret_cast<ulonglong2 *>(ct + ((m * 2 + p) * k + i) * (size_t)n + s) = ulonglong2{r[0][p], r[1][p]};
        }
    }
}

// NTT-form pipeline: ct[m][p][i][s] (= the transformed noise rows) += U_ntt[m][i][s] * pk[p][i][s]; two neighbouring slots per thread, both polys
__global__ void __launch_bounds__(256) enc_fma_kernel(const u64 *U, const u64 *pk, u64 *ct, const ModParams *mods, int n, int k)
{
    const size_t row = blockIdx.x;                // m*k + i
    const int i = (int)(row % k);
    const size_t m = row / k;
    const ModParams md = mods[i];
    const u64 *u = U + row * (size_t)n;
    for (int s = 2 * threadIdx.x; s < n; s += 2 * blockDim.x) {
        const ulonglong2 uv = *reinterpret_cast<const ulonglong2 *>(u + s);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(pk + ((size_t)p * k + i) * n + s);
            u64 *dst = ct + ((m * 2 + p) * k + i) * (size_t)n + s;
            const ulonglong2 cv = *reinterpret_cast<const ulonglong2 *>(dst);
            *reinterpret_cast<ulonglong2 *>(dst) = ulonglong2{addmod(cv.x, mulmod(uv.x, kv.x, md), md.q), addmod(cv.y, mulmod(uv.y, kv.y, md), md.q)};
        }
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

// thresholds of the noise magnitude (enc_sample_kernel): P(|e| <= a) for the clipped, truncated normal of the reference, in long double (64-bit mantissa)
static const EncCdt &enc_cdt()
{
    static const EncCdt T = [] {
        EncCdt t{};
        const long double sigma = 3.19L, lim = 6.0L * sigma, r2 = sqrtl(2.0L) * sigma;
        const long double Z = erfl(lim / r2);                                  // P(|g| <= 6 sigma)
        for (int a = 0; a < 19; a++) {
            // P(|e| > a) = P(a + 1 <= |g| <= lim) / Z, from the complementary error function (no cancellation in the tail)
            const long double tail = (erfcl((long double)(a + 1) / r2) - erfcl(lim / r2)) / Z;
            t.t[a] = (u64)floorl((1.0L - tail) * 18446744073709551616.0L);
        }
        return t;
    }();
    return T;
}
void k_encrypt_cdt(u64 *out19) { const EncCdt &T = enc_cdt(); for (int a = 0; a < 19; a++) out19[a] = T.t[a]; }

int k_encrypt(crc_ctx *c, const u64 *pk, const u64 *plain, size_t cnt, const ChaChaKey &key, u64 stream_base, u64 *ct, u64 *work, hipStream_t st, bool out_ntt,
              bool plain_compact)
{
    if (cnt == 0) return CRC_OK;
    if (plain_compact && !out_ntt) return CRC_ERR_INVALID_ARGUMENT;       // (the coefficient-form pipeline adds Delta m from dense rows: expand first)
    const int n = c->n, k = c->k;
    u64 *U = work; signed char *E = reinterpret_cast<signed char *>(U + cnt * (size_t)n * k);
    const int pairs = n / 2, threads = pairs < 256 ? pairs : 256, pblocks = (pairs + threads - 1) / threads;
    if (cnt * (size_t)pblocks > 0x7fffffffULL || cnt * 2 * (size_t)k > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    int rc;
    if (out_ntt) {
        // c_p = NTT(e_p (+ Delta m)) + pk_p . NTT(u): three forward transforms per modulus and no inverse one -- the same residues as transforming the
        // coefficient form
        if (plain_compact)
            hipLaunchKernelGGL((enc_sample_kernel<true, true>), dim3((unsigned)(cnt * pblocks)), dim3(threads), 0, st, U, E, ct, plain, c->d_mods, n, k, key,
                               stream_base, enc_cdt(), c->plain);
        else
            hipLaunchKernelGGL((enc_sample_kernel<true, false>), dim3((unsigned)(cnt * pblocks)), dim3(threads), 0, st, U, E, ct, plain, c->d_mods, n, k, key,
                               stream_base, enc_cdt(), c->plain);
        HIPCHK(hipGetLastError());
        if ((rc = k_ntt_ct(c, false, U, U, cnt, 1, false, st, nullptr, 0, 0, 0))) return rc;
        // (the product joins in the transform's last loop where the ring has the wave-local kernel; else as a pass of its own)
        rc = k_ntt_ct_fwd_fma(c, ct, cnt, U, pk, st);
        if (rc != CRC_ERR_UNSUPPORTED) return rc;
        if ((rc = k_ntt_ct(c, false, ct, ct, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc;
        hipLaunchKernelGGL(enc_fma_kernel, dim3((unsigned)(cnt * k)), dim3(n / 2 < 256 ? n / 2 : 256), 0, st, U, pk, ct, c->d_mods, n, k);
        HIPCHK(hipGetLastError());
        return CRC_OK;
    }
    hipLaunchKernelGGL((enc_sample_kernel<false, false>), dim3((unsigned)(cnt * pblocks)), dim3(threads), 0, st, U, E, ct, plain, c->d_mods, n, k, key, stream_base,
        enc_cdt(),
                       c->plain);
    HIPCHK(hipGetLastError());
    if ((rc = k_ntt_ct(c, false, U, U, cnt, 1, false, st, nullptr, 0, 0, 0))) return rc;
    hipLaunchKernelGGL(enc_mulpk_kernel, dim3((unsigned)(cnt * 2 * k)), dim3(256), 0, st, U, pk, ct, c->d_mods, n, k);
    HIPCHK(hipGetLastError());
    if ((rc = k_ntt_ct(c, true, ct, ct, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc;
    hipLaunchKernelGGL(enc_finish_kernel, dim3((unsigned)(cnt * 2 * k)), dim3(256), 0, st, ct, E, plain, c->d_mods, n, k, c->plain);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
