// chacha.h -- ChaCha20 block function (RFC 8439 layout) shared by the host client code (client.cpp) and the device encryptor
// (kernels_client.hip): the generator behind every secret the client side draws (secret key s, encryption sample u, noise e).
//
// The reference (SEAL 2.3.1 KeyGenerator / Encryptor) draws from std::random_device; there are no reference bits to match, only
// laws (uniform ternary, clipped normal sigma 3.19 cut at 6 sigma: util/globals.cpp:13-15).  A 256-bit key -- from the OS
// (crc_random_key -> getrandom(2)) in normal use, expanded from a 64-bit seed only in the explicitly deterministic test / bench
// entry points -- plus a 96-bit nonce that names the stream (domain, ciphertext index, coefficient) gives every sample its own
// keystream block(s); the 32-bit block counter extends a stream when rejection sampling runs past one block.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CHACHA_HD __host__ __device__ __forceinline__
#else
#define CHACHA_HD inline
#endif

struct ChaChaKey { uint32_t w[8]; };

CHACHA_HD uint32_t chacha_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = chacha_rotl(d, 16); c += d; b ^= c; b = chacha_rotl(b, 12); \
    a += b; d ^= a; d = chacha_rotl(d, 8);  c += d; b ^= c; b = chacha_rotl(b, 7);

// out[16] = ChaCha20 block(key, counter, nonce[3])
CHACHA_HD void chacha20_block(const ChaChaKey &key, uint32_t counter, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t out[16])
{
    uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
    uint32_t x4 = key.w[0], x5 = key.w[1], x6 = key.w[2], x7 = key.w[3], x8 = key.w[4], x9 = key.w[5], x10 = key.w[6], x11 = key.w[7];
    uint32_t x12 = counter, x13 = n0, x14 = n1, x15 = n2;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int r = 0; r < 10; r++) {
        CHACHA_QR(x0, x4, x8, x12) CHACHA_QR(x1, x5, x9, x13) CHACHA_QR(x2, x6, x10, x14) CHACHA_QR(x3, x7, x11, x15)
        CHACHA_QR(x0, x5, x10, x15) CHACHA_QR(x1, x6, x11, x12) CHACHA_QR(x2, x7, x8, x13) CHACHA_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + 0x61707865u; out[1] = x1 + 0x3320646eu; out[2] = x2 + 0x79622d32u; out[3] = x3 + 0x6b206574u;
    out[4] = x4 + key.w[0]; out[5] = x5 + key.w[1]; out[6] = x6 + key.w[2]; out[7] = x7 + key.w[3];
    out[8] = x8 + key.w[4]; out[9] = x9 + key.w[5]; out[10] = x10 + key.w[6]; out[11] = x11 + key.w[7];
    out[12] = x12 + counter; out[13] = x13 + n0; out[14] = x14 + n1; out[15] = x15 + n2;
}

// keystream reader: 64-bit words of the stream named by (n0, n1, n2) under `key`
struct ChaChaStream {
    ChaChaKey key; uint32_t n0, n1, n2, counter; uint32_t buf[16]; int pos;
    CHACHA_HD ChaChaStream(const ChaChaKey &k, uint32_t a, uint32_t b, uint32_t c) : key(k), n0(a), n1(b), n2(c), counter(0), pos(16) {}
    CHACHA_HD uint64_t next()
    {
        if (pos >= 16) { chacha20_block(key, counter++, n0, n1, n2, buf); pos = 0; }
        const uint64_t v = (uint64_t)buf[pos] | ((uint64_t)buf[pos + 1] << 32);
        pos += 2;
        return v;
    }
    CHACHA_HD double unit() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }   // uniform in (0, 1)
};

// deterministic entry points (tests, bench, goldens): the 64-bit seed in front of a fixed tag.  NOT secure -- the seed is public
inline ChaChaKey chacha_seed_key(uint64_t seed)
{
    ChaChaKey k; k.w[0] = (uint32_t)seed; k.w[1] = (uint32_t)(seed >> 32);
    const char tag[25] = "crcnn-seeded-determinist";
    for (int i = 0; i < 6; i++) k.w[2 + i] = (uint32_t)(uint8_t)tag[4 * i] | ((uint32_t)(uint8_t)tag[4 * i + 1] << 8) | ((uint32_t)(uint8_t)tag[4 * i + 2] << 16) | ((uint32_t)(uint8_t)tag[4 * i + 3] << 24);
    return k;
}
inline ChaChaKey chacha_load_key(const uint8_t *key)
{
    ChaChaKey k;
    for (int i = 0; i < 8; i++) k.w[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    return k;
}

// stream domains (nonce word 2 carries the domain in its top byte)
enum { CHACHA_DOM_KEYGEN = 1, CHACHA_DOM_EVK = 2, CHACHA_DOM_ENC_HOST = 3, CHACHA_DOM_ENC_DEV = 4 };
