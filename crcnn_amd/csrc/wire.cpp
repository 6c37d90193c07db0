// wire.cpp -- SEAL 2.3.1 wire formats for the objects that cross the evaluation path's boundary (host side).
//
// So that artefacts interchange with an unmodified CrCNN: ciphertexts (`Ciphertext::save`, SEAL/ciphertext.cpp:103-130),
// evaluation keys (`EvaluationKeys::save`, evaluationkeys.cpp:8-39), public / secret keys (publickey.h:81-98, secretkey.h:86-107 over
// BigPolyArray::save bigpolyarray.cpp:131-141 / BigPoly::save bigpoly.cpp:467-476).  Every object starts with the 32-byte parameter
// hash: SHA3-256 over the little-endian words [x^n+1 coefficients (n+1), q_i..., t, sigma, 6 sigma] (encryptionparams.cpp:69-100).
// Engine layout [..][k][n]  <->  SEAL layout [..][k][n+1] (dead pad word, always 0).
#include "ctx.h"
#include <cstring>
#include <vector>

namespace {
// Keccak-f[1600] / SHA3-256 from FIPS 202
inline u64 rotl64(u64 x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }
void keccak_f(u64 st[25])
{
    static const u64 RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL, 0x0000000080000001ULL,
                               0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                               0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL,
                               0x000000000000800aULL, 0x800000008000000aULL, 0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // lane x + 5y
    for (int r = 0; r < 24; r++) {
        u64 C[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) { const u64 D = C[(x + 4) % 5] ^ rotl64(C[(x + 1) % 5], 1); for (int y = 0; y < 25; y += 5) st[x + y] ^= D; }
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(st[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 25; y += 5) for (int x = 0; x < 5; x++) st[x + y] = B[x + y] ^ (~B[(x + 1) % 5 + y] & B[(x + 2) % 5 + y]);
        st[0] ^= RC[r];
    }
}
void sha3_256(const uint8_t *msg, size_t len, uint8_t out[32])
{
    u64 st[25] = {0};
    const size_t rate = 136;
    std::vector<uint8_t> buf(msg, msg + len);
    buf.push_back(0x06);
    while (buf.size() % rate) buf.push_back(0);
    buf.back() |= 0x80;
    for (size_t o = 0; o < buf.size(); o += rate) {
        for (size_t i = 0; i < rate / 8; i++) { u64 w; std::memcpy(&w, &buf[o + 8 * i], 8); st[i] ^= w; }
        keccak_f(st);
    }
    std::memcpy(out, st, 32);
}
struct Writer { uint8_t *p; size_t cap, off = 0; bool ok = true;
    void put(const void *s, size_t n) { if (off + n > cap) { ok = false; off += n; return; } std::memcpy(p + off, s, n); off += n; }
    void i32(int32_t v) { put(&v, 4); } };
struct Reader { const uint8_t *p; size_t len, off = 0; bool ok = true;
    void get(void *d, size_t n) { if (off + n > len) { ok = false; return; } std::memcpy(d, p + off, n); off += n; }
    int32_t i32() { int32_t v = 0; get(&v, 4); return v; } };

void put_rows(Writer &w, const crc_ctx *c, const u64 *rows, size_t nrows)      // [nrows][n] -> [nrows][n+1]
{
    const u64 zero = 0;
    for (size_t r = 0; r < nrows; r++) { w.put(rows + r * (size_t)c->n, 8 * (size_t)c->n); w.put(&zero, 8); }
}
bool get_rows(Reader &r, const crc_ctx *c, u64 *rows, size_t nrows)
{
    for (size_t i = 0; i < nrows; i++) { u64 pad = 1; r.get(rows + i * (size_t)c->n, 8 * (size_t)c->n); r.get(&pad, 8); if (!r.ok || pad != 0) return false; }
    return true;
}
void ct_save(Writer &w, const crc_ctx *c, const uint8_t hash[32], const u64 *ct, int size)
{
    w.put(hash, 32); w.i32(size); w.i32(c->n + 1); w.i32(c->k);
    put_rows(w, c, ct, (size_t)size * c->k);
}
bool ct_load(Reader &r, const crc_ctx *c, const uint8_t hash[32], u64 *ct, int max_size, int *size)
{
    uint8_t h[32]; r.get(h, 32);
    const int sz = r.i32(), pc = r.i32(), km = r.i32();
    if (!r.ok || std::memcmp(h, hash, 32) || pc != c->n + 1 || km != c->k || sz < 0 || sz > max_size) return false;   // "not valid for encryption parameters"
    *size = sz;
    return get_rows(r, c, ct, (size_t)sz * c->k);
}
}  // namespace

extern "C" int crc_params_hash(const crc_ctx *c, uint64_t out[4])
{
    if (!c || !out) return CRC_ERR_INVALID_ARGUMENT;
    std::vector<u64> words((size_t)c->n + 1, 0);
    words[0] = 1; words[c->n] = 1;
    for (u64 q : c->q) words.push_back(q);
    words.push_back(c->t);
    const double sigma = 3.19, maxdev = 6 * 3.19;          // util/globals.cpp:13-15
    u64 bits; std::memcpy(&bits, &sigma, 8); words.push_back(bits); std::memcpy(&bits, &maxdev, 8); words.push_back(bits);
    sha3_256(reinterpret_cast<const uint8_t *>(words.data()), words.size() * 8, reinterpret_cast<uint8_t *>(out));
    return CRC_OK;
}

extern "C" size_t crc_seal_ct_bytes(const crc_ctx *c, int size) { return 32 + 12 + (size_t)size * c->k * (c->n + 1) * 8; }
extern "C" size_t crc_seal_evk_bytes(const crc_ctx *c, int dbc)
{
    size_t b = 32 + 4 + 4 + 4;
    for (int l = 0; l < c->k; l++) b += crc_seal_ct_bytes(c, 2 * evk_digits(c->q[l], dbc));
    return b;
}
extern "C" size_t crc_seal_pk_bytes(const crc_ctx *c) { return 32 + 12 + (size_t)2 * c->k * (c->n + 1) * 8; }
extern "C" size_t crc_seal_sk_bytes(const crc_ctx *c) { return 32 + 8 + (size_t)c->k * (c->n + 1) * 8; }

extern "C" int crc_seal_ct_save(const crc_ctx *c, const uint64_t *h_ct, int size, void *buf, size_t cap, size_t *written)
{
    if (!c || !h_ct || !buf || size < 1) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Writer w{(uint8_t *)buf, cap};
    ct_save(w, c, hash, h_ct, size);
    if (written) *written = w.off;
    return w.ok ? CRC_OK : CRC_ERR_INVALID_ARGUMENT;
}
extern "C" int crc_seal_ct_load(const crc_ctx *c, const void *buf, size_t bytes, uint64_t *h_ct, int max_size, int *size, size_t *consumed)
{
    if (!c || !buf || !h_ct || !size) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Reader r{(const uint8_t *)buf, bytes};
    if (!ct_load(r, c, hash, h_ct, max_size, size)) return CRC_ERR_INVALID_ARGUMENT;
    if (consumed) *consumed = r.off;
    return CRC_OK;
}
extern "C" int crc_seal_evk_save(const crc_ctx *c, const uint64_t *h_evk, int dbc, void *buf, size_t cap, size_t *written)
{
    if (!c || !h_evk || !buf) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Writer w{(uint8_t *)buf, cap};
    w.put(hash, 32); w.i32(dbc); w.i32(1); w.i32(c->k);              // keys_.size() = 1 (count), keys_[0].size() = k
    const u64 *src = h_evk;
    for (int l = 0; l < c->k; l++) { const int L = evk_digits(c->q[l], dbc); ct_save(w, c, hash, src, 2 * L); src += (size_t)2 * L * c->k * c->n; }
    if (written) *written = w.off;
    return w.ok ? CRC_OK : CRC_ERR_INVALID_ARGUMENT;
}
extern "C" int crc_seal_evk_load(const crc_ctx *c, const void *buf, size_t bytes, uint64_t *h_evk, int *dbc)
{
    if (!c || !buf || !h_evk || !dbc) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32], h[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Reader r{(const uint8_t *)buf, bytes};
    r.get(h, 32); const int d = r.i32(), dim1 = r.i32();
    if (!r.ok || std::memcmp(h, hash, 32) || dim1 < 1 || d < 1 || d > 60) return CRC_ERR_INVALID_ARGUMENT;
    if (r.i32() != c->k) return CRC_ERR_INVALID_ARGUMENT;
    u64 *dst = h_evk;
    for (int l = 0; l < c->k; l++) { const int L = evk_digits(c->q[l], d); int sz = 0; if (!ct_load(r, c, hash, dst, 2 * L, &sz) || sz != 2 * L) return CRC_ERR_INVALID_ARGUMENT; dst += (size_t)2 * L * c->k * c->n; }
    *dbc = d;
    return CRC_OK;
}
extern "C" int crc_seal_pk_save(const crc_ctx *c, const uint64_t *h_pk, void *buf, size_t cap, size_t *written)
{
    if (!c || !h_pk || !buf) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Writer w{(uint8_t *)buf, cap};
    w.put(hash, 32); w.i32(2); w.i32(c->n + 1); w.i32(c->k * 64); put_rows(w, c, h_pk, (size_t)2 * c->k);
    if (written) *written = w.off;
    return w.ok ? CRC_OK : CRC_ERR_INVALID_ARGUMENT;
}
extern "C" int crc_seal_pk_load(const crc_ctx *c, const void *buf, size_t bytes, uint64_t *h_pk)
{
    if (!c || !buf || !h_pk) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32], h[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Reader r{(const uint8_t *)buf, bytes};
    r.get(h, 32);
    if (r.i32() != 2 || r.i32() != c->n + 1 || r.i32() != c->k * 64 || !r.ok || std::memcmp(h, hash, 32)) return CRC_ERR_INVALID_ARGUMENT;
    return get_rows(r, c, h_pk, (size_t)2 * c->k) ? CRC_OK : CRC_ERR_INVALID_ARGUMENT;
}
extern "C" int crc_seal_sk_save(const crc_ctx *c, const uint64_t *h_sk, void *buf, size_t cap, size_t *written)
{
    if (!c || !h_sk || !buf) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Writer w{(uint8_t *)buf, cap};
    w.put(hash, 32); w.i32(c->n + 1); w.i32(c->k * 64); put_rows(w, c, h_sk, (size_t)c->k);
    if (written) *written = w.off;
    return w.ok ? CRC_OK : CRC_ERR_INVALID_ARGUMENT;
}
extern "C" int crc_seal_sk_load(const crc_ctx *c, const void *buf, size_t bytes, uint64_t *h_sk)
{
    if (!c || !buf || !h_sk) return CRC_ERR_INVALID_ARGUMENT;
    uint8_t hash[32], h[32]; crc_params_hash(c, reinterpret_cast<uint64_t *>(hash));
    Reader r{(const uint8_t *)buf, bytes};
    r.get(h, 32);
    if (r.i32() != c->n + 1 || r.i32() != c->k * 64 || !r.ok || std::memcmp(h, hash, 32)) return CRC_ERR_INVALID_ARGUMENT;
    return get_rows(r, c, h_sk, (size_t)c->k) ? CRC_OK : CRC_ERR_INVALID_ARGUMENT;
}
