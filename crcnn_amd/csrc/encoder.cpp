// encoder.cpp -- host-side balanced base-3 fractional encoding of model weights / pixels.
//
// Produces the same plaintext polynomials as the reference's `fraencoder->encode(x)` with
// FractionalEncoder(t, x^n+1, 64 integer coeffs, 32 fractional coeffs, base 3) (CrCNN/src/globals.cpp:52,
// SEAL encoder.cpp:1013-1076 encode_odd, :408-481 BalancedEncoder::encode, :1226-1270 decode):
//   value = I + f,  I = round-half-away(value);  I in balanced ternary on coefficients 0,1,2,...  (digit -1 -> t-1);
//   f expanded to 32 balanced ternary digits d_1..d_32 (weights 3^-m, ties toward zero), digit d_m stored on
//   coefficient n-m with the opposite sign (x^n = -1 turns x^(n-m) into -x^-m).
#include "ctx.h"
#include "host_parallel.h"
#include <cmath>
#include <cstring>

namespace {
constexpr int kFrac = 32, kInt = 64;

// balanced ternary digits of an integer, little endian; returns SEAL's Plaintext coeff_count for it
int encode_integer(u64 t, int64_t v, u64 *dst, int cap)
{
    int cc;
    if (v >= 0) {
        u64 u = (u64)v; int bits = u ? 64 - __builtin_clzll(u) : 0;
        cc = (int)(std::ceil((double)bits / std::log2(3.0)) + 1);            // encoder.cpp:411-413
        for (int i = 0; i < cc && i < cap; i++) dst[i] = 0;
        for (int i = 0; u; i++) { u64 r = u % 3; dst[i] = r == 0 ? 0 : (r == 1 ? 1 : t - 1); u = (u + 1) / 3; }
    } else {
        u64 u = (u64)(-v);
        cc = (int)(std::ceil(64.0 / std::log2(3.0)) + 1);                    // bit count of a negative int64 is 64 (:438-440)
        for (int i = 0; i < cc && i < cap; i++) dst[i] = 0;
        for (int i = 0; u; i++) { u64 r = u % 3; dst[i] = r == 0 ? 0 : (r == 1 ? t - 1 : 1); u = (u + 1) / 3; }
    }
    return cc;
}

int encode_one(const crc_ctx *c, double value, u64 *co)
{
    const int n = c->n; const u64 t = c->t;
    std::memset(co, 0, sizeof(u64) * (size_t)n);
    u64 ip[80];
    const int64_t whole = (int64_t)std::round(value);
    const int icc = encode_integer(t, whole, ip, 80);
    double f = value - (double)whole;
    if (f == 0) { for (int i = 0; i < icc && i < n; i++) co[i] = ip[i]; return icc; }
    for (int m = 1; m <= kFrac; m++) {
        f *= 3;
        const double mag = std::ceil(std::fabs(f) - 0.5);                    // ties toward zero
        const int64_t d = f >= 0 ? (int64_t)mag : -(int64_t)mag;
        f -= (double)d;
        co[n - m] = d == 0 ? 0 : (d > 0 ? t - (u64)d : (u64)(-d));           // sign flipped
    }
    for (int i = 0; i < icc; i++) co[i] = ip[i];
    return n + 1;
}

// the same plaintext as encode_one in compact form: words 0..63 = coefficients 0..63, words 64..95 = coefficients n-32..n-1 (every other coefficient is zero)
int encode_one_compact(const crc_ctx *c, double value, u64 *cp)
{
    const u64 t = c->t;
    std::memset(cp, 0, sizeof(u64) * CRC_PLAIN_COMPACT_WORDS);
    u64 ip[80];
    const int64_t whole = (int64_t)std::round(value);
    const int icc = encode_integer(t, whole, ip, 80);
    double f = value - (double)whole;
    for (int i = 0; i < icc && i < CRC_PLAIN_COMPACT_LOW; i++) cp[i] = ip[i];
    if (f == 0) return icc;
    for (int m = 1; m <= kFrac; m++) {
        f *= 3;
        const double mag = std::ceil(std::fabs(f) - 0.5);
        const int64_t d = f >= 0 ? (int64_t)mag : -(int64_t)mag;
        f -= (double)d;
        cp[CRC_PLAIN_COMPACT_LOW + kFrac - m] = d == 0 ? 0 : (d > 0 ? t - (u64)d : (u64)(-d));
    }
    return c->n + 1;
}

int64_t balanced_value(u64 t, const u64 *cf, int cnt)
{
    const u64 thr = (t + 1) >> 1;
    int top = cnt - 1; while (top >= 0 && cf[top] == 0) top--;
    int64_t r = 0;
    for (int i = top; i >= 0; i--) r = r * 3 + (cf[i] >= thr ? -(int64_t)(t - cf[i]) : (int64_t)cf[i]);
    return r;
}
}  // namespace

extern "C" int crc_encode_f64(const crc_ctx *c, const double *v, size_t count, uint64_t *plain, int32_t *cc)
{
    if (!c || !v || !plain) return CRC_ERR_INVALID_ARGUMENT;
    if (kInt + kFrac >= c->n + 1) return CRC_ERR_INVALID_ARGUMENT;            // encoder.cpp:993-996
    crc_host::parallel_for(count, 256, [&](size_t b, size_t e) { for (size_t i = b; i < e; i++) { int r = encode_one(c, v[i], plain + i * (size_t)c->n); if (cc) cc[i] = r; } });
    return CRC_OK;
}
extern "C" int crc_encode_f32(const crc_ctx *c, const float *v, size_t count, uint64_t *plain, int32_t *cc)
{
    if (!c || !v || !plain) return CRC_ERR_INVALID_ARGUMENT;
    if (kInt + kFrac >= c->n + 1) return CRC_ERR_INVALID_ARGUMENT;
    crc_host::parallel_for(count, 256, [&](size_t b, size_t e) { for (size_t i = b; i < e; i++) { int r = encode_one(c, (double)v[i], plain + i * (size_t)c->n); if (cc) cc[i] = r; } });
    return CRC_OK;
}
extern "C" int crc_encode_f32_compact(const crc_ctx *c, const float *v, size_t count, uint64_t *compact, int32_t *cc)
{
    static_assert(CRC_PLAIN_COMPACT_LOW == kInt && CRC_PLAIN_COMPACT_HIGH == kFrac && CRC_PLAIN_COMPACT_WORDS == kInt + kFrac, "compact plaintext layout");
    if (!c || !v || !compact) return CRC_ERR_INVALID_ARGUMENT;
    if (kInt + kFrac >= c->n + 1) return CRC_ERR_INVALID_ARGUMENT;
    crc_host::parallel_for(count, 1024, [&](size_t b, size_t e) { for (size_t i = b; i < e; i++) { int r = encode_one_compact(c, (double)v[i], compact + i * (size_t)CRC_PLAIN_COMPACT_WORDS); if (cc) cc[i] = r; } });
    return CRC_OK;
}
extern "C" double crc_decode(const crc_ctx *c, const uint64_t *plain)
{
    const int n = c->n;
    const int64_t ipart = balanced_value(c->t, plain, kInt);
    double frac = 0;
    for (int i = 0; i < kFrac; i++) { frac += (double)balanced_value(c->t, plain + n - kFrac + i, 1); frac /= 3; }
    return (double)ipart - frac;
}
extern "C" int crc_bn_invstd_f32(const float *var, size_t count, float *out)
{
    if (!var || !out) return CRC_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < count; i++) out[i] = (float)(1 / std::sqrt((double)var[i] + 0.00001));   // cnnBuilder.cpp:100-102
    return CRC_OK;
}
