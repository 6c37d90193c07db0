// client.cpp -- client-side BFV on the host CPU: key generation, encryption, decryption, noise budget.
//
// Outside the accelerated path (needs the secret key) but part of what a CrCNN user calls around it:
// setParameters / encryptImage / decryptImage (CrCNN/src/globals.cpp:25-56,127-157,207-230) over SEAL's KeyGenerator
// (keygenerator.cpp:96-282), Encryptor (encryptor.cpp:71-134) and Decryptor (decryptor.cpp:107-236, BEHZ gamma rounding).
// Randomness: ChaCha20 keystreams (chacha.h) under a 256-bit key -- crc_random_key() draws it from the OS (getrandom(2)); the
// uint64 "seed" entry points expand a public seed and exist for tests / bench / goldens only.  SEAL draws from std::random_device,
// so the reference defines sampling laws, not bits.
#include "ctx.h"
#include "host_parallel.h"
#include "chacha.h"
#include <cmath>
#include <cstring>
#include <vector>
#include <cerrno>
#include <sys/random.h>

typedef unsigned __int128 u128;

namespace {
typedef ChaChaStream Rng;                                 // 64-bit words of a ChaCha20 keystream (chacha.h)
ChaChaKey load_key(const uint8_t *key) { return chacha_load_key(key); }
ChaChaKey seed_key(u64 seed) { return chacha_seed_key(seed); }

void ternary(const crc_ctx *c, Rng &r, u64 *p)            // uniform over {-1,0,1}^n in RNS form
{
    for (int s = 0; s < c->n; s++) {
        u64 v; do { v = r.next() >> 62; } while (v == 3);
        for (int i = 0; i < c->k; i++) p[(size_t)i * c->n + s] = v == 0 ? 0 : (v == 1 ? 1 : c->q[i] - 1);
    }
}
void gauss(const crc_ctx *c, Rng &r, u64 *p)              // sigma = 3.19 clipped at 6 sigma, truncated to integer
{
    const double sigma = 3.19, lim = 6 * sigma;           // util/globals.cpp:13-15
    for (int s = 0; s < c->n; s++) {
        double v;
        do { v = sigma * std::sqrt(-2.0 * std::log(r.unit())) * std::cos(6.283185307179586 * r.unit()); } while (std::fabs(v) > lim);
        const int64_t e = (int64_t)v;
        for (int i = 0; i < c->k; i++) p[(size_t)i * c->n + s] = e >= 0 ? (u64)e : c->q[i] - (u64)(-e);
    }
}
void uniform(const crc_ctx *c, Rng &r, u64 *p)
{
    for (int i = 0; i < c->k; i++) for (int s = 0; s < c->n; s++) { u128 z = ((u128)r.next() << 64) | r.next(); p[(size_t)i * c->n + s] = (u64)(z % c->q[i]); }
}
u64 delta_times(const crc_ctx *c, int i, u64 m)           // Delta*m (+ q mod t for the upper half), evaluator.cpp:1168-1191
{
    u128 z = (u128)c->plain.delta[i] * m;
    if (m >= c->plain.threshold) z += c->plain.uhi[i];
    return (u64)(z % c->q[i]);
}
// v = c0 + c1 s + c2 s^2 ... in coefficient form
void dot_secret(const crc_ctx *c, const u64 *sk, const u64 *ct, int size, u64 *v)
{
    const int n = c->n, k = c->k;
    std::vector<u64> tmp(n), sp(n);
    for (int i = 0; i < k; i++) {
        const u64 q = c->q[i]; u64 *o = v + (size_t)i * n;
        std::memset(o, 0, 8 * (size_t)n);
        std::memcpy(sp.data(), sk + (size_t)i * n, 8 * (size_t)n);
        for (int p = 1; p < size; p++) {
            std::memcpy(tmp.data(), ct + ((size_t)p * k + i) * n, 8 * (size_t)n);
            h_ntt_fwd(c->tabs[i], tmp.data(), n);
            for (int s = 0; s < n; s++) { o[s] = addmod(o[s], h_mulmod(tmp[s], sp[s], q), q); sp[s] = h_mulmod(sp[s], sk[(size_t)i * n + s], q); }
        }
        h_ntt_inv(c->tabs[i], o, n);
        for (int s = 0; s < n; s++) o[s] = addmod(o[s], ct[(size_t)i * n + s], q);
    }
}
}  // namespace

// known-answer access to the generator (RFC 8439 section 2.3.2 test vector; tests/test_client_rng.py)
extern "C" int crc_chacha20_block(const uint8_t *key, uint32_t counter, const uint8_t *nonce, uint8_t *out)
{
    if (!key || !nonce || !out) return CRC_ERR_INVALID_ARGUMENT;
    u32 nw[3], o[16];
    for (int i = 0; i < 3; i++) nw[i] = (u32)nonce[4 * i] | ((u32)nonce[4 * i + 1] << 8) | ((u32)nonce[4 * i + 2] << 16) | ((u32)nonce[4 * i + 3] << 24);
    chacha20_block(load_key(key), counter, nw[0], nw[1], nw[2], o);
    for (int i = 0; i < 16; i++) { out[4 * i] = (uint8_t)o[i]; out[4 * i + 1] = (uint8_t)(o[i] >> 8); out[4 * i + 2] = (uint8_t)(o[i] >> 16); out[4 * i + 3] = (uint8_t)(o[i] >> 24); }
    return CRC_OK;
}

extern "C" int crc_random_key(uint8_t *key)
{
    if (!key) return CRC_ERR_INVALID_ARGUMENT;
    size_t got = 0;
    while (got < CRC_KEY_BYTES) {
        const ssize_t r = getrandom(key + got, CRC_KEY_BYTES - got, 0);
        if (r < 0) { if (errno == EINTR) continue; return CRC_ERR_IO; }
        got += (size_t)r;
    }
    return CRC_OK;
}

static int keygen_impl(const crc_ctx *c, const ChaChaKey &key, uint64_t *sk, uint64_t *pk)
{
    const int n = c->n, k = c->k;
    Rng r(key, 0, 0, (u32)CHACHA_DOM_KEYGEN << 24);
    std::vector<u64> e((size_t)k * n);
    ternary(c, r, sk); uniform(c, r, pk + (size_t)k * n); gauss(c, r, e.data());
    for (int i = 0; i < k; i++) {
        const u64 q = c->q[i];
        h_ntt_fwd(c->tabs[i], sk + (size_t)i * n, n); h_ntt_fwd(c->tabs[i], pk + ((size_t)k + i) * n, n); h_ntt_fwd(c->tabs[i], e.data() + (size_t)i * n, n);
        for (int s = 0; s < n; s++) {             // pk0 = -(a s + e), pk1 = a   (keygenerator.cpp:112-150), NTT form
            const size_t o = (size_t)i * n + s;
            pk[o] = negmod(addmod(h_mulmod(sk[o], pk[(size_t)k * n + o], q), e[o], q), q);
        }
    }
    return CRC_OK;
}
extern "C" int crc_keygen_key(const crc_ctx *c, const uint8_t *key, uint64_t *sk, uint64_t *pk)
{
    if (!c || !key || !sk || !pk) return CRC_ERR_INVALID_ARGUMENT;
    return keygen_impl(c, load_key(key), sk, pk);
}
extern "C" int crc_keygen(const crc_ctx *c, uint64_t seed, uint64_t *sk, uint64_t *pk)
{
    if (!c || !sk || !pk) return CRC_ERR_INVALID_ARGUMENT;
    return keygen_impl(c, seed_key(seed), sk, pk);
}

static int gen_evk_impl(const crc_ctx *c, const ChaChaKey &ckey, const uint64_t *sk, int dbc, uint64_t *evk)
{
    if (dbc < 1 || dbc > 60) return CRC_ERR_INVALID_ARGUMENT;
    const int n = c->n, k = c->k;
    Rng r(ckey, 0, 0, (u32)CHACHA_DOM_EVK << 24);
    std::vector<u64> s2((size_t)k * n), e((size_t)k * n);
    for (int j = 0; j < k; j++) for (int s = 0; s < n; s++) { const size_t o = (size_t)j * n + s; s2[o] = h_mulmod(sk[o], sk[o], c->q[j]); }
    u64 *key = evk;
    for (int l = 0; l < k; l++) {
        u64 factor = 1;                            // (q/q_l) mod q_l, then times 2^(dbc*d)   keygenerator.cpp:652-698
        for (int j = 0; j < k; j++) if (j != l) factor = h_mulmod(factor, c->q[j] % c->q[l], c->q[l]);
        const int L = evk_digits(c->q[l], dbc);
        for (int d = 0; d < L; d++) {
            u64 *first = key + (size_t)(2 * d) * k * n, *second = first + (size_t)k * n;
            uniform(c, r, second); gauss(c, r, e.data());
            for (int j = 0; j < k; j++) {
                const u64 q = c->q[j];
                h_ntt_fwd(c->tabs[j], second + (size_t)j * n, n); h_ntt_fwd(c->tabs[j], e.data() + (size_t)j * n, n);
                for (int s = 0; s < n; s++) {
                    const size_t o = (size_t)j * n + s;
                    u64 v = negmod(addmod(h_mulmod(second[o], sk[o], q), e[o], q), q);
                    if (j == l) v = addmod(v, h_mulmod(s2[o], factor, q), q);
                    first[o] = v;
                }
            }
            factor = h_mulmod(factor, (1ULL << dbc) % c->q[l], c->q[l]);
        }
        key += (size_t)2 * L * k * n;
    }
    return CRC_OK;
}
extern "C" int crc_gen_evk_key(const crc_ctx *c, const uint8_t *key, const uint64_t *sk, int dbc, uint64_t *evk)
{
    if (!c || !key || !sk || !evk) return CRC_ERR_INVALID_ARGUMENT;
    return gen_evk_impl(c, load_key(key), sk, dbc, evk);
}
extern "C" int crc_gen_evk(const crc_ctx *c, uint64_t seed, const uint64_t *sk, int dbc, uint64_t *evk)
{
    if (!c || !sk || !evk) return CRC_ERR_INVALID_ARGUMENT;
    return gen_evk_impl(c, seed_key(seed), sk, dbc, evk);
}

static int encrypt_impl(const crc_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t count, const ChaChaKey &key, uint64_t stream_base, uint64_t *ct)
{
    const int n = c->n, k = c->k;
    // one keystream per ciphertext: ranges of ciphertexts on the host threads, the same bits however they are split
    crc_host::parallel_for(count, 8, [&](size_t m0, size_t m1) {
    std::vector<u64> u((size_t)k * n), e((size_t)k * n);
    for (size_t m = m0; m < m1; m++) {
        const u64 sid = stream_base + m;
        Rng r(key, (u32)sid, (u32)(sid >> 32), (u32)CHACHA_DOM_ENC_HOST << 24);
        u64 *o = ct + m * 2 * (size_t)k * n; const u64 *pl = plain + m * (size_t)n;
        ternary(c, r, u.data());
        for (int i = 0; i < k; i++) {
            const u64 q = c->q[i];
            h_ntt_fwd(c->tabs[i], u.data() + (size_t)i * n, n);
            for (int s = 0; s < n; s++) { const size_t x = (size_t)i * n + s; o[x] = h_mulmod(u[x], pk[x], q); o[(size_t)k * n + x] = h_mulmod(u[x], pk[(size_t)k * n + x], q); }
            h_ntt_inv(c->tabs[i], o + (size_t)i * n, n); h_ntt_inv(c->tabs[i], o + ((size_t)k + i) * n, n);
        }
        for (int p = 0; p < 2; p++) {
            gauss(c, r, e.data());
            for (int i = 0; i < k; i++) for (int s = 0; s < n; s++) {
                const size_t x = (size_t)i * n + s; u64 v = addmod(o[(size_t)p * k * n + x], e[x], c->q[i]);
                if (p == 0) v = addmod(v, delta_times(c, i, pl[s]), c->q[i]);
                o[(size_t)p * k * n + x] = v;
            }
        }
    }
    });
    return CRC_OK;
}
extern "C" int crc_encrypt_key(const crc_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t count, const uint8_t *key, uint64_t stream_base, uint64_t *ct)
{
    if (!c || !pk || !plain || !ct || !key) return CRC_ERR_INVALID_ARGUMENT;
    return encrypt_impl(c, pk, plain, count, load_key(key), stream_base, ct);
}
extern "C" int crc_encrypt(const crc_ctx *c, const uint64_t *pk, const uint64_t *plain, size_t count, uint64_t seed, uint64_t *ct)
{
    if (!c || !pk || !plain || !ct) return CRC_ERR_INVALID_ARGUMENT;
    return encrypt_impl(c, pk, plain, count, seed_key(seed), 0, ct);
}

extern "C" int crc_decrypt(const crc_ctx *c, const uint64_t *sk, const uint64_t *ct, size_t count, int size, uint64_t *plain)
{
    if (!c || !sk || !ct || !plain || size < 2) return CRC_ERR_INVALID_ARGUMENT;
    const int n = c->n, k = c->k; const u64 t = c->t, gamma = c->gmod.q;
    std::vector<u64> v((size_t)k * n);
    for (size_t m = 0; m < count; m++) {
        dot_secret(c, sk, ct + m * (size_t)size * k * n, size, v.data());
        u64 *out = plain + m * (size_t)n;
        for (int s = 0; s < n; s++) {
            u128 at = 0, ag = 0;                   // fastbconv_plain_gamma of (t gamma v), baseconverter.cpp:744-797
            for (int i = 0; i < k; i++) {
                const u64 y = h_mulmod(h_mulmod(v[(size_t)i * n + s], c->tgamma_mod_q[i], c->q[i]), c->behz.inv_qhat[i], c->q[i]);
                at += (u128)y * c->qhat_mod_tg[0][i]; ag += (u128)y * c->qhat_mod_tg[1][i];
            }
            const u64 rt = h_mulmod((u64)(at % t), c->neg_inv_q_mod_tg[0], t), rg = h_mulmod((u64)(ag % gamma), c->neg_inv_q_mod_tg[1], gamma);
            const u64 w = rg > (gamma >> 1) ? addmod(rt, (gamma - rg) % t, t) : submod(rt, rg % t, t);     // centred correction, decryptor.cpp:193-215
            out[s] = h_mulmod(w, c->inv_gamma_mod_t, t);
        }
    }
    return CRC_OK;
}

extern "C" int crc_noise_budget(const crc_ctx *c, const uint64_t *sk, const uint64_t *ct, int size)
{
    // invariant noise budget = bits(q) - bits(|| t (c0 + c1 s + ...) mod q ||_inf centred) - 1   (decryptor.cpp:295-403)
    if (!c || !sk || !ct || size < 2) return CRC_ERR_INVALID_ARGUMENT;
    const int n = c->n, k = c->k;
    std::vector<u64> v((size_t)k * n);
    dot_secret(c, sk, ct, size, v.data());
    auto cmp = [&](const u64 *a, const u64 *b) { for (int l = k - 1; l >= 0; l--) if (a[l] != b[l]) return a[l] < b[l] ? -1 : 1; return 0; };
    auto sub = [&](u64 *a, const u64 *b) { u64 br = 0; for (int l = 0; l < k; l++) { u128 z = (u128)a[l] - b[l] - br; a[l] = (u64)z; br = (u64)(z >> 64) & 1; } };
    std::vector<std::vector<u64>> qhat(k, std::vector<u64>(k, 0));
    for (int i = 0; i < k; i++) { qhat[i][0] = 1; for (int j = 0; j < k; j++) if (j != i) { u64 cy = 0; for (int l = 0; l < k; l++) { u128 z = (u128)qhat[i][l] * c->q[j] + cy; qhat[i][l] = (u64)z; cy = (u64)(z >> 64); } } }
    std::vector<u64> half(c->qbig), norm(k, 0), acc(k), term(k);
    { u64 cy = 0; for (int l = k - 1; l >= 0; l--) { u64 nc = half[l] & 1; half[l] = (half[l] >> 1) | (cy << 63); cy = nc; } }
    for (int s = 0; s < n; s++) {
        std::fill(acc.begin(), acc.end(), 0);
        for (int i = 0; i < k; i++) {
            const u64 x = h_mulmod(h_mulmod(v[(size_t)i * n + s], c->t % c->q[i], c->q[i]), c->behz.inv_qhat[i], c->q[i]);
            u64 cy = 0; for (int l = 0; l < k; l++) { u128 z = (u128)qhat[i][l] * x + cy; term[l] = (u64)z; cy = (u64)(z >> 64); }
            cy = 0; for (int l = 0; l < k; l++) { u128 z = (u128)acc[l] + term[l] + cy; acc[l] = (u64)z; cy = (u64)(z >> 64); }
            if (cmp(acc.data(), c->qbig.data()) >= 0) sub(acc.data(), c->qbig.data());
        }
        if (cmp(acc.data(), half.data()) > 0) { std::vector<u64> tq(c->qbig); sub(tq.data(), acc.data()); acc = tq; }
        if (cmp(acc.data(), norm.data()) > 0) norm = acc;
    }
    int nb = 0; for (int l = k - 1; l >= 0; l--) if (norm[l]) { nb = 64 * l + 64 - __builtin_clzll(norm[l]); break; }
    const int b = c->total_bits - nb - 1;
    return b > 0 ? b : 0;
}
