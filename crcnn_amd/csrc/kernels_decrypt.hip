// kernels_decrypt.hip -- BFV decryption, fractional decoding and re-encoding on gfx950: the client-side "refresh" of CrCNN's Network::forward
// (CrCNN/src/network.cpp:30-34: decryptImage -> encryptImage in front of layer 6; globals.cpp:144-157, 207-230) as kernels on the launch stream.
//
// Decryptor::decrypt (SEAL decryptor.cpp:107-236) per ciphertext:  v = c0 + c1 s (+ c2 s^2) mod q in the NTT domain, one inverse transform per residue,
// then per coefficient the BEHZ correction with the auxiliary prime gamma (baseconverter.cpp:744-797 fastbconv_plain_gamma, decryptor.cpp:193-215):
//     y_i = v_i t gamma (q/q_i)^-1 mod q_i;   r_m = -(sum_i y_i (q/q_i) mod m) q^-1 mod m  for m in {t, gamma};   r_gamma centred;
//     m = (r_t - r_gamma) gamma^-1 mod t.
// Exact integer arithmetic: the plaintext polynomial is the reference's, bit for bit (tests: SEAL's own ref_dec_* vectors and the oracle's decrypt).
//
// FractionalEncoder::decode / encode (SEAL encoder.cpp:1226-1270, 1013-1076, 408-481; CrCNN instantiates 64 integer + 32 fractional coefficients, base 3) are
// the double-precision loops of encoder.cpp (host) restated per ciphertext -- IEEE additions, divisions by 3 and multiplications by 3 in the SAME order with
// contraction switched off, so a value decoded and re-encoded here is the one the reference's client computes.
#include "kernels.h"

struct DecParams {
    int k;
    u64 yc[CRC_MAXK], yc_s[CRC_MAXK];            // t gamma (q/q_i)^-1 mod q_i and its Shoup companion: the two constant products of the reference in one
    u64 qhat_t[CRC_MAXK], qhat_g[CRC_MAXK];       // (q/q_i) mod t, mod gamma
    ModParams tmod, gmod;
    u64 ninv_q_t, ninv_q_g, inv_gamma_t;          // (-q)^-1 mod t, (-q)^-1 mod gamma, gamma^-1 mod t
};

// V[m][i][s] = sum_p ct[m][p][i][s] sk[i][s]^p  (NTT domain; size 2 or 3), two slots per lane
__global__ void __launch_bounds__(256) dec_dot_kernel(const u64 *ct, const u64 *sk, u64 *V, const ModParams *mods, int n, int k, int size)
{
    const size_t row = blockIdx.x;                // m*k + i
    const int i = (int)(row % k);
    const size_t m = row / k;
    const ModParams md = mods[i];
    const u64 *base = ct + (m * size * k + i) * (size_t)n, *sr = sk + (size_t)i * n;
    u64 *dst = V + row * (size_t)n;
    const size_t pstride = (size_t)k * n;
    for (int s = 2 * threadIdx.x; s < n; s += 2 * blockDim.x) {
        const ulonglong2 sv = *reinterpret_cast<const ulonglong2 *>(sr + s);
        ulonglong2 acc = *reinterpret_cast<const ulonglong2 *>(base + (size_t)(size - 1) * pstride + s);
        for (int p = size - 2; p >= 0; p--) {     // Horner in s
            const ulonglong2 cv = *reinterpret_cast<const ulonglong2 *>(base + (size_t)p * pstride + s);
            acc.x = addmod(mulmod(acc.x, sv.x, md), cv.x, md.q);
            acc.y = addmod(mulmod(acc.y, sv.y, md), cv.y, md.q);
        }
        *reinterpret_cast<ulonglong2 *>(dst + s) = acc;
    }
}

__device__ __forceinline__ void acc128(u64 &lo, u64 &hi, u64 a, u64 b)
{
    u64 pl, ph; mul64wide(a, b, pl, ph);
    const u64 l2 = lo + pl; hi += ph + (l2 < lo); lo = l2;
}

// one plaintext coefficient from the k coefficient-form residues v[i n] of c0 + c1 s: the gamma-corrected scaling by t / q
__device__ __forceinline__ u64 dec_gamma_one(const u64 *v, const ModParams *mods, int n, const DecParams &dp)
{
    u64 tl = 0, th = 0, gl = 0, gh = 0;
    for (int i = 0; i < dp.k; i++) {
        const u64 y = mulmod_shoup(v[(size_t)i * n], dp.yc[i], dp.yc_s[i], mods[i].q);
        acc128(tl, th, y, dp.qhat_t[i]); acc128(gl, gh, y, dp.qhat_g[i]);         // k <= 8 terms below 2^62 2^61: no overflow
    }
    const u64 t = dp.tmod.q, g = dp.gmod.q;
    const u64 rt = mulmod(barrett128(tl, th, dp.tmod), dp.ninv_q_t, dp.tmod), rg = mulmod(barrett128(gl, gh, dp.gmod), dp.ninv_q_g, dp.gmod);
    // centred correction (decryptor.cpp:193-215)
    const u64 w = rg > (g >> 1) ? addmod(rt, barrett128(g - rg, 0, dp.tmod), t) : submod(rt, barrett128(rg, 0, dp.tmod), t);
    return mulmod(w, dp.inv_gamma_t, dp.tmod);
}

// plain[m][s] for every coefficient of every ciphertext
__global__ void __launch_bounds__(256) dec_gamma_kernel(const u64 *V, u64 *plain, const ModParams *mods, int n, DecParams dp)
{
    const size_t m = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    plain[m * (size_t)n + s] = dec_gamma_one(V + m * (size_t)dp.k * n + s, mods, n, dp);
}

// ---- FractionalEncoder(t, x^n + 1, 64, 32, base 3) -------------------------------------------------------------------
__device__ __forceinline__ long long dec_centred(u64 c, u64 t, u64 thr) { return c >= thr ? -(long long)(t - c) : (long long)c; }

// decode (encoder.cpp:1226-1270 as restated in encoder.cpp's crc_decode): lo = coefficients 0..63, hi = coefficients n-32..n-1
__device__ double dev_fra_decode(const u64 *lo, const u64 *hi, u64 t)
{
#pragma clang fp contract(off)
    const u64 thr = (t + 1) >> 1;
    int top = CRC_PLAIN_COMPACT_LOW - 1;
    while (top >= 0 && lo[top] == 0) top--;
    unsigned long long r = 0;                     // two's-complement Horner (wraps like the host's int64 on garbage; exact on encodings)
    for (int i = top; i >= 0; i--) r = r * 3ull + (unsigned long long)dec_centred(lo[i], t, thr);
    double frac = 0;
    for (int i = 0; i < CRC_PLAIN_COMPACT_HIGH; i++) { frac += (double)dec_centred(hi[i], t, thr); frac /= 3.0; }
    return (double)(long long)r - frac;
}

// encode (encoder.cpp:1013-1076 encode_odd, :408-481): out = 96 words in the compact layout (words 0..63 = coefficients 0..63, 64..95 = n-32..n-1)
__device__ void dev_fra_encode(double value, u64 t, u64 *out)
{
#pragma clang fp contract(off)
    for (int i = 0; i < CRC_PLAIN_COMPACT_WORDS; i++) out[i] = 0;
    const double rounded = round(value);          // half away from zero, as std::round
    const long long whole = (long long)rounded;
    unsigned long long u = whole >= 0 ? (unsigned long long)whole : 0ull - (unsigned long long)whole;
    const bool neg = whole < 0;
    for (int i = 0; u && i < CRC_PLAIN_COMPACT_LOW; i++) {
        const unsigned r = (unsigned)(u % 3ull);
        out[i] = r == 0 ? 0 : ((r == 1) != neg ? 1 : t - 1);
        u = (u + 1) / 3ull;
    }
    double f = value - (double)whole;
    if (f == 0) return;
    for (int m = 1; m <= CRC_PLAIN_COMPACT_HIGH; m++) {
        f = f * 3.0;
        const double mag = ceil(fabs(f) - 0.5);   // ties toward zero
        const long long d = f >= 0 ? (long long)mag : -(long long)mag;
        f = f - (double)d;
        out[CRC_PLAIN_COMPACT_LOW + CRC_PLAIN_COMPACT_HIGH - m] = d == 0 ? 0 : (d > 0 ? t - (u64)d : (u64)(-d));
    }
}

// one lane per plaintext: plain [count][n] -> doubles
__global__ void __launch_bounds__(64) fra_decode_kernel(const u64 *plain, double *out, size_t count, int n, u64 t)
{
    const size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= count) return;
    const u64 *p = plain + m * (size_t)n;
    out[m] = dev_fra_decode(p, p + n - CRC_PLAIN_COMPACT_HIGH, t);
}

// one workgroup per plaintext: the value (MODE 0: float, 1: double, 2: decoded from the plaintext `src` [count][n], rounded to float as decryptImage keeps
// it -- globals.cpp:221 -- and left in vals_out when that is not NULL) is encoded by lane 0 into LDS and the workgroup writes the dense row
template <int MODE>
__global__ void __launch_bounds__(256) fra_encode_kernel(const void *src, u64 *plain, float *vals_out, int n, u64 t)
{
    __shared__ u64 enc[CRC_PLAIN_COMPACT_WORDS];
    const size_t m = blockIdx.x;
    if (threadIdx.x == 0) {
        double value;
        if (MODE == 0) value = (double)static_cast<const float *>(src)[m];
        else if (MODE == 1) value = static_cast<const double *>(src)[m];
        else {
            const u64 *p = static_cast<const u64 *>(src) + m * (size_t)n;
            const float fv = (float)dev_fra_decode(p, p + n - CRC_PLAIN_COMPACT_HIGH, t);
            if (vals_out) vals_out[m] = fv;
            value = (double)fv;
        }
        dev_fra_encode(value, t, enc);
    }
    __syncthreads();
    u64 *dst = plain + m * (size_t)n;
    for (int s = threadIdx.x * 2; s < n; s += blockDim.x * 2) {
        ulonglong2 v = make_ulonglong2(0, 0);
        if (s < CRC_PLAIN_COMPACT_LOW) { v.x = enc[s]; v.y = enc[s + 1]; }
        else if (s >= n - CRC_PLAIN_COMPACT_HIGH) { v.x = enc[CRC_PLAIN_COMPACT_LOW + s - (n - CRC_PLAIN_COMPACT_HIGH)];
            v.y = enc[CRC_PLAIN_COMPACT_LOW + s + 1 - (n - CRC_PLAIN_COMPACT_HIGH)]; }
        *reinterpret_cast<ulonglong2 *>(dst + s) = v;
    }
}

// The middle of a refresh, one workgroup (128 lanes) per ciphertext: FractionalEncoder::decode reads coefficients 0..63 and n-32..n-1 only, so only those 96 are
// scaled (96 lanes, one each, into LDS); lane 0 decodes, rounds to float (globals.cpp:221) and encodes again; the workgroup writes the new plaintext in the compact
// form (96 words: crc_plain_expand's layout), which the encryptor's sampling kernel reads directly -- no dense plaintext row is written or read
__global__ void __launch_bounds__(128) dec_recode_kernel(const u64 *V, u64 *compact, float *vals_out, const ModParams *mods, int n, DecParams dp)
{
    __shared__ u64 co[CRC_PLAIN_COMPACT_WORDS], enc[CRC_PLAIN_COMPACT_WORDS];
    const size_t m = blockIdx.x;
    const int l = threadIdx.x;
    if (l < CRC_PLAIN_COMPACT_WORDS) {
        const int s = l < CRC_PLAIN_COMPACT_LOW ? l : n - CRC_PLAIN_COMPACT_HIGH + (l - CRC_PLAIN_COMPACT_LOW);
        co[l] = dec_gamma_one(V + m * (size_t)dp.k * n + s, mods, n, dp);
    }
    __syncthreads();
    if (l == 0) {
        const float fv = (float)dev_fra_decode(co, co + CRC_PLAIN_COMPACT_LOW, dp.tmod.q);
        if (vals_out) vals_out[m] = fv;
        dev_fra_encode((double)fv, dp.tmod.q, enc);
    }
    __syncthreads();
    if (l < CRC_PLAIN_COMPACT_WORDS) compact[m * (size_t)CRC_PLAIN_COMPACT_WORDS + l] = enc[l];
}

static DecParams dec_params(const crc_ctx *c)
{
    DecParams dp{};
    dp.k = c->k;
    for (int i = 0; i < c->k; i++) {
        dp.yc[i] = h_mulmod(c->tgamma_mod_q[i], c->behz.inv_qhat[i], c->q[i]);
        dp.yc_s[i] = (u64)(((unsigned __int128)dp.yc[i] << 64) / c->q[i]);
        dp.qhat_t[i] = c->qhat_mod_tg[0][i]; dp.qhat_g[i] = c->qhat_mod_tg[1][i];
    }
    dp.tmod = c->tmod; dp.gmod = c->gmod;
    dp.ninv_q_t = c->neg_inv_q_mod_tg[0]; dp.ninv_q_g = c->neg_inv_q_mod_tg[1]; dp.inv_gamma_t = c->inv_gamma_mod_t;
    return dp;
}

// work: V [cnt][k][n], and for coefficient-form input the transformed ciphertexts [cnt][size][k][n] behind it
size_t k_decrypt_work_words(const crc_ctx *c, size_t cnt, int size, bool in_ntt)
{
    return cnt * (size_t)c->k * c->n * (in_ntt ? 1 : 1 + (size_t)size);
}

// V = work [cnt][k][n]: c0 + c1 s (+ c2 s^2) in coefficient form
static int decrypt_rows(crc_ctx *c, const u64 *sk, const u64 *ct, size_t cnt, int size, bool in_ntt, u64 *work, hipStream_t st)
{
    const int n = c->n, k = c->k;
    if (size < 2 || size > 3 || cnt * (size_t)k > 0x7fffffffULL || cnt > 65535u * 4096ull) return CRC_ERR_INVALID_ARGUMENT;
    u64 *V = work;
    int rc;
    const u64 *hat = ct;
    if (!in_ntt) {
        u64 *tmp = V + cnt * (size_t)k * n;
        if ((rc = k_ntt_ct(c, false, ct, tmp, cnt, size, false, st, nullptr, 0, 0, 0))) return rc;
        hat = tmp;
    }
    hipLaunchKernelGGL(dec_dot_kernel, dim3((unsigned)(cnt * k)), dim3(n / 2 < 256 ? n / 2 : 256), 0, st, hat, sk, V, c->d_mods, n, k, size);
    HIPCHK(hipGetLastError());
    return k_ntt_ct(c, true, V, V, cnt, 1, false, st, nullptr, 0, 0, 0);
}

int k_decrypt(crc_ctx *c, const u64 *sk, const u64 *ct, size_t cnt, int size, bool in_ntt, u64 *plain, u64 *work, hipStream_t st)
{
    if (cnt == 0) return CRC_OK;
    const int n = c->n, k = c->k;
    int rc;
    if ((rc = decrypt_rows(c, sk, ct, cnt, size, in_ntt, work, st))) return rc;
    u64 *V = work;
    const DecParams dp = dec_params(c);
    const int threads = n < 256 ? n : 256;
    for (size_t o = 0; o < cnt; o += 65535) {     // grid.y limit
        const size_t ch = cnt - o < 65535 ? cnt - o : 65535;
        hipLaunchKernelGGL(dec_gamma_kernel, dim3((unsigned)((n + threads - 1) / threads), (unsigned)ch), dim3(threads), 0, st, V + o * (size_t)k * n,
                           plain + o * (size_t)n, c->d_mods, n, dp);
        HIPCHK(hipGetLastError());
    }
    return CRC_OK;
}

int k_fra_decode(crc_ctx *c, const u64 *plain, size_t cnt, double *out, hipStream_t st)
{
    if (cnt == 0) return CRC_OK;
    if ((cnt + 63) / 64 > 0x7fffffffULL || c->n <= CRC_PLAIN_COMPACT_WORDS) return CRC_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(fra_decode_kernel, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st, plain, out, cnt, c->n, c->t);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// mode 0: float values, 1: double values, 2: plaintexts [cnt][n] decoded, rounded to float (-> vals_out when not NULL) and encoded again
int k_fra_encode(crc_ctx *c, const void *src, int mode, size_t cnt, u64 *plain, float *vals_out, hipStream_t st)
{
    if (cnt == 0) return CRC_OK;
    if (cnt > 0x7fffffffULL || c->n <= CRC_PLAIN_COMPACT_WORDS || mode < 0 || mode > 2) return CRC_ERR_INVALID_ARGUMENT;
    const dim3 g((unsigned)cnt), b(256);
    if (mode == 0) hipLaunchKernelGGL(fra_encode_kernel<0>, g, b, 0, st, src, plain, vals_out, c->n, c->t);
    else if (mode == 1) hipLaunchKernelGGL(fra_encode_kernel<1>, g, b, 0, st, src, plain, vals_out, c->n, c->t);
    else hipLaunchKernelGGL(fra_encode_kernel<2>, g, b, 0, st, src, plain, vals_out, c->n, c->t);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// decrypt -> decode -> float -> encode for the refresh: compact plaintexts [cnt][96] (and the floats) out; work as k_decrypt
int k_decrypt_recode(crc_ctx *c, const u64 *sk, const u64 *ct, size_t cnt, bool in_ntt, u64 *compact, float *vals_out, u64 *work, hipStream_t st)
{
    if (cnt == 0) return CRC_OK;
    if (cnt > 0x7fffffffULL || c->n <= CRC_PLAIN_COMPACT_WORDS) return CRC_ERR_INVALID_ARGUMENT;
    int rc;
    if ((rc = decrypt_rows(c, sk, ct, cnt, 2, in_ntt, work, st))) return rc;
    hipLaunchKernelGGL(dec_recode_kernel, dim3((unsigned)cnt), dim3(128), 0, st, work, compact, vals_out, c->d_mods, c->n, dec_params(c));
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
