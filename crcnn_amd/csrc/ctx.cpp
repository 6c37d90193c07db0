// ctx.cpp -- builds the context tables on the host and uploads them to HBM.
#include "ctx.h"
#include "host_parallel.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <string>

typedef unsigned __int128 u128;

static thread_local int g_last_hip = 0;
int crc_set_hip_error(hipError_t e) { g_last_hip = (int)e; return CRC_ERR_HIP; }

extern "C" int crc_last_hip_error(void) { return g_last_hip; }
extern "C" int crc_version(void) { return 100; }
extern "C" const char *crc_strerror(int s)
{
    switch (s) {
    case CRC_OK: return "ok";
    case CRC_ERR_INVALID_ARGUMENT: return "invalid argument";
    case CRC_ERR_PARAMETERS: return "encryption parameters are not set correctly";
    case CRC_ERR_HIP: return "HIP runtime error";
    case CRC_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case CRC_ERR_IO: return "I/O error";
    case CRC_ERR_NOT_FOUND: return "not found";
    case CRC_ERR_COMM: return "RCCL error";
    }
    return "unknown status";
}

// ---- small host number theory -------------------------------------------------------------------------------
u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }
u64 h_powmod(u64 a, u64 e, u64 q)
{
    u64 r = 1 % q; a %= q;
    for (; e; e >>= 1) { if (e & 1) r = h_mulmod(r, a, q); a = h_mulmod(a, a, q); }
    return r;
}
u64 h_invmod(u64 a, u64 q)
{
    __int128 t = 0, nt = 1, r = q, nr = a % q;
    while (nr) { __int128 d = r / nr, x = t - d * nt; t = nt; nt = x; x = r - d * nr; r = nr; nr = x; }
    return (u64)(t < 0 ? t + q : t);
}
static int sigbits(u64 v) { return v ? 64 - __builtin_clzll(v) : 0; }

ModParams make_mod(u64 q, bool no_fold)
{
    ModParams m{};
    u128 all = ~(u128)0, quo = all / q;
    if (all % q + 1 == q) quo += 1;                // q | 2^128 (m~ = 2^32)
    m.q = q; m.r0 = (u64)quo; m.r1 = (u64)(quo >> 64); m.two_q = 2 * q; m.bits = (u32)sigbits(q);
    m.fold = no_fold ? 0 : fold_constant(q, m.bits);
    return m;
}

int evk_digits(u64 q, int dbc) { int L = 0; while (q) { L++; q >>= dbc; } return L; }

static bool is_prime(u64 n)
{
    if (n < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (n % p == 0) return n == p; }
    u64 d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int r = 1; r < s; r++) { x = h_mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}

// numerically smallest primitive 2n-th root of unity mod q (what SEAL's try_minimal_primitive_root converges to,
// util/uintarithsmallmod.cpp:83-108): every primitive root is an odd power of any one of them.
static u64 minimal_primitive_root(u64 two_n, u64 q)
{
    if ((q - 1) % two_n) return 0;
    u64 e = (q - 1) / two_n, root = 0;
    for (u64 g = 2; g < 4096 && !root; g++) {
        u64 c = h_powmod(g, e, q);
        if (h_powmod(c, two_n >> 1, q) == q - 1) root = c;
    }
    if (!root) return 0;
    u64 sq = h_mulmod(root, root, q), cur = root, best = root;
    for (u64 i = 0; i < two_n / 2; i++) { if (cur < best) best = cur; cur = h_mulmod(cur, sq, q); }
    return best;
}

static u32 bitrev(u32 x, int bits) { u32 r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

static bool build_ntt(HostNtt &T, int logn, u64 q, bool no_fold)
{
    int n = 1 << logn;
    T.m = make_mod(q, no_fold);
    T.root = minimal_primitive_root(2 * (u64)n, q);
    if (!T.root) return false;
    u64 iroot = h_invmod(T.root, q);
    T.inv_n = h_invmod((u64)n, q);
    T.rp.assign(n, 0); T.srp.assign(n, 0); T.irp2.assign(n, 0); T.sirp2.assign(n, 0); T.irp.assign(n, 0); T.sirp.assign(n, 0);
    u64 p = 1, ip = 1;
    for (int i = 0; i < n; i++) {
        u32 j = bitrev((u32)i, logn);
        T.rp[j] = p;
        T.irp2[j] = (ip & 1) ? (u64)(((u128)ip + q) >> 1) : ip >> 1;       // psi^-i / 2 mod q
        T.irp[j] = ip;
        p = h_mulmod(p, T.root, q); ip = h_mulmod(ip, iroot, q);
    }
    for (int i = 0; i < n; i++) {
        T.srp[i] = (u64)(((u128)T.rp[i] << 64) / q);
        T.sirp2[i] = (u64)(((u128)T.irp2[i] << 64) / q);
        T.sirp[i] = (u64)(((u128)T.irp[i] << 64) / q);
    }
    return true;
}

// host reference transforms (client side + table self-checks); same ordering as the device kernels
void h_ntt_fwd(const HostNtt &T, u64 *a, int n)
{
    u64 q = T.m.q;
    for (int m = 1, t = n >> 1; m < n; m <<= 1, t >>= 1)
        for (int i = 0; i < m; i++) {
            u64 w = T.rp[m + i];
            for (int j = 2 * i * t; j < 2 * i * t + t; j++) {
                u64 v = h_mulmod(a[j + t], w, q), u = a[j];
                a[j] = addmod(u, v, q); a[j + t] = submod(u, v, q);
            }
        }
}
void h_ntt_inv(const HostNtt &T, u64 *a, int n)
{
    u64 q = T.m.q;
    for (int m = n, t = 1; m > 1; m >>= 1, t <<= 1) {
        int h = m >> 1;
        for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) {
            u64 w = T.irp2[h + i];
            for (int j = j1; j < j1 + t; j++) {
                u64 u = a[j], v = a[j + t];
                u64 s = addmod(u, v, q);
                a[j] = (s & 1) ? (u64)(((u128)s + q) >> 1) : s >> 1;
                a[j + t] = h_mulmod(submod(u, v, q), w, q);
            }
        }
    }
}

// SEAL's internal BEHZ moduli (util/globals.cpp:321-367): 61-bit primes = 1 mod 2^18
static const u64 kAuxMods[] = {
    0x1fffffffffb40001ULL, 0x1fffffffff500001ULL, 0x1fffffffff380001ULL, 0x1fffffffff000001ULL, 0x1ffffffffef00001ULL,
    0x1ffffffffee80001ULL, 0x1ffffffffeb40001ULL, 0x1ffffffffe780001ULL, 0x1ffffffffe600001ULL, 0x1ffffffffe4c0001ULL };
static const u64 kMsk = 0x1fffffffffe00001ULL, kMtilde = 1ULL << 32, kGamma = 0x1fffffffffc80001ULL;

static u64 prod_mod(const u64 *v, int cnt, int skip, u64 m)
{
    u64 r = 1 % m;
    for (int j = 0; j < cnt; j++) if (j != skip) r = h_mulmod(r, v[j] % m, m);
    return r;
}

// the only place the engine reads its environment: once per context, before any kernel can be launched on it
static void read_tuning(CrcTuning &t)
{
    auto geti = [](const char *name, long long dflt) { const char *e = getenv(name); return e && *e ? atoll(e) : dflt; };
    t.no_fold = getenv("CRC_NO_FOLD") ? 1 : 0;
    t.ntt_inv61_loose = getenv("CRC_NTT_INV61_LOOSE") ? 1 : 0;
    t.mfma_order = (int)geti("CRC_MFMA_ORDER", -1);
    t.mfma_variant = (int)geti("CRC_MFMA_VARIANT", 2);
    { const int v = (int)geti("CRC_MFMA_RING", 0); t.mfma_ring = v >= 3 && v <= 5 ? v : 0; }
    { const int v = (int)geti("CRC_CONV1_WAVES", 0); t.conv1_waves = v == 8 || v == 12 ? v : 0; }
    t.conv1_narrow = (int)geti("CRC_CONV1_NARROW", 1);
    { const long long v = geti("CRC_CONV1_PASS_BYTES", 0); t.conv1_pass_bytes = v > 0 ? v : 0; }
    { const int v = (int)geti("CRC_LIMB_PACK_GROUP", 1); t.limb_pack_group = v > 0 ? v : 1; }
    t.mac2_cfg = (int)geti("CRC_MAC2_CFG", 0);
    t.mac_order = (int)geti("CRC_MAC_ORDER", -1);
    t.mac_regstage = (int)geti("CRC_MAC_REGSTAGE", 0);
    t.mac_stream = (int)geti("CRC_MAC_STREAM", 1);
    t.mac2_dbg = (int)geti("CRC_MAC2_DBG", 0);
    t.relin_path = (int)geti("CRC_RELIN_PATH", 0);
    t.sq_path = (int)geti("CRC_SQ_PATH", 0);
    t.sq_chunk = (int)geti("CRC_SQ_CHUNK", 0);
    t.sq_fuse = (int)geti("CRC_SQ_FUSE", -1);
    t.f64_radix = (int)geti("CRC_F64_RADIX", 0);
    t.f64_wave = (int)geti("CRC_F64_WAVE", -1);
    t.ntt_wave = (int)geti("CRC_NTT_WAVE", -1);
    t.mfma_min_steps = (int)geti("CRC_MFMA_MIN_STEPS", 0);
    t.relin_mac_ct = (int)geti("CRC_RELIN_MAC_CT", 0);
    t.ntt_split = (int)geti("CRC_NTT_SPLIT", 1);
}
extern "C" int crc_ctx_set_tuning(crc_ctx *c, const char *name, long long value)
{
    if (!c || !name) return CRC_ERR_INVALID_ARGUMENT;
    const std::string s(name);
    CrcTuning &t = c->tune;
    if (s == "mfma_order") t.mfma_order = (int)value;
    else if (s == "mfma_variant") t.mfma_variant = (int)value;
    else if (s == "mfma_ring") t.mfma_ring = value >= 3 && value <= 5 ? (int)value : 0;
    else if (s == "conv1_waves") t.conv1_waves = value == 8 || value == 12 ? (int)value : 0;
    else if (s == "conv1_narrow") t.conv1_narrow = (int)value;
    else if (s == "conv1_pass_bytes") t.conv1_pass_bytes = value > 0 ? value : 0;
    else if (s == "limb_pack_group") t.limb_pack_group = value > 0 ? (int)value : 1;
    else if (s == "mac2_cfg") t.mac2_cfg = (int)value;
    else if (s == "mac_order") t.mac_order = (int)value;
    else if (s == "mac_regstage") t.mac_regstage = (int)value;
    else if (s == "mac_stream") t.mac_stream = (int)value;
    else if (s == "ntt_inv61_loose") t.ntt_inv61_loose = value ? 1 : 0;
    else if (s == "relin_path") t.relin_path = (int)value;
    else if (s == "sq_path") t.sq_path = (int)value;
    else if (s == "sq_chunk") t.sq_chunk = (int)value;
    else if (s == "sq_fuse") t.sq_fuse = (int)value;
    else if (s == "f64_radix") t.f64_radix = (int)value;
    else if (s == "f64_wave") t.f64_wave = (int)value;
    else if (s == "ntt_wave") t.ntt_wave = (int)value;
    else if (s == "mfma_min_steps") t.mfma_min_steps = (int)value;
    else if (s == "relin_mac_ct") t.relin_mac_ct = (int)value;
    else if (s == "ntt_split") t.ntt_split = (int)value;
    else return CRC_ERR_NOT_FOUND;             // (no_fold is baked into the tables at creation)
    return CRC_OK;
}

extern "C" int crc_default_coeff_modulus_128(int n, uint64_t *q, int cap)
{
    static const u64 m1024[] = {0x7e00001}, m2048[] = {0x3fffffff000001}, m4096[] = {0x7fffffff380001, 0x3fffffff000001},
        m8192[] = {0x7fffffff380001, 0x7ffffffef00001, 0x3fffffff000001, 0x3ffffffef40001},
        m16384[] = {0x7fffffff380001, 0x7ffffffef00001, 0x7ffffffeac0001, 0x7ffffffe700001, 0x7ffffffe600001, 0x7ffffffe4c0001, 0x3fffffff000001,
            0x3ffffffef40001};
    const u64 *src; int cnt;
    switch (n) {
    case 1024: src = m1024; cnt = 1; break;
    case 2048: src = m2048; cnt = 1; break;
    case 4096: src = m4096; cnt = 2; break;
    case 8192: src = m8192; cnt = 4; break;
    case 16384: src = m16384; cnt = 8; break;
    default: return CRC_ERR_INVALID_ARGUMENT;
    }
    for (int i = 0; i < cnt && i < cap; i++) q[i] = src[i];
    return cnt;
}

extern "C" int crc_ctx_create(int n, const uint64_t *q, int k, uint64_t t, int device, crc_ctx **out)
{
    if (!out || !q) return CRC_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    // SEALContext::validate, context.cpp:15-169
    if (k < 1 || k > CRC_MAXK) return CRC_ERR_PARAMETERS;
    if (n < 64 || n > 32768 || (n & (n - 1))) return CRC_ERR_PARAMETERS;
    if (t < 2 || (t >> 60)) return CRC_ERR_PARAMETERS;
    for (int i = 0; i < k; i++) {
        if (q[i] < 2 || (q[i] >> 60)) return CRC_ERR_PARAMETERS;
        if (!is_prime(q[i]) || (q[i] - 1) % (2 * (u64)n)) return CRC_ERR_PARAMETERS;      // enable_ntt
        for (int j = 0; j < i; j++) if (q[i] == q[j]) return CRC_ERR_PARAMETERS;
        if (t % q[i] == 0) return CRC_ERR_PARAMETERS;
    }
    crc_ctx *c = new crc_ctx();
    c->n = n; c->k = k; c->t = t; c->device = device;
    c->logn = 0; while ((1 << c->logn) < n) c->logn++;
    c->q.assign(q, q + k);

    // aux base size rule, baseconverter.cpp:47-56
    int total = 0; for (int i = 0; i < k; i++) total += sigbits(q[i]);
    c->ka = k + ((32 + sigbits(t) + total >= 61 * k + 61) ? 1 : 0);
    c->kb = c->ka + 1;
    u64 bsk[CRC_MAXB];
    for (int j = 0; j < c->ka; j++) bsk[j] = kAuxMods[j];
    bsk[c->ka] = kMsk;

    read_tuning(c->tune);
    c->tabs.resize(k + c->kb);
    for (int i = 0; i < k; i++) if (!build_ntt(c->tabs[i], c->logn, q[i], c->tune.no_fold)) { delete c; return CRC_ERR_PARAMETERS; }
    for (int j = 0; j < c->kb; j++) if (!build_ntt(c->tabs[k + j], c->logn, bsk[j], c->tune.no_fold)) { delete c; return CRC_ERR_PARAMETERS; }

    // floor(q/t) mod q_i and (q mod t) mod q_i, evaluator.cpp:66-105 -- long division of the multi-limb q by t
    {
        std::vector<u64> big(k, 0); big[0] = 1;
        for (int i = 0; i < k; i++) { u64 carry = 0; for (int l = 0; l < k; l++) { u128 z = (u128)big[l] * q[i] + carry; big[l] = (u64)z;
            carry = (u64)(z >> 64); } }
        c->qbig = big;
        c->total_bits = 0; for (int l = k - 1; l >= 0; l--) if (big[l]) { c->total_bits = 64 * l + sigbits(big[l]); break; }
        std::vector<u64> quo(big); u64 rem = 0;
        for (int l = k - 1; l >= 0; l--) { u128 z = ((u128)rem << 64) | quo[l]; quo[l] = (u64)(z / t); rem = (u64)(z % t); }
        memset(&c->plain, 0, sizeof c->plain);
        c->plain.threshold = (t + 1) >> 1;
        c->plain.fast = 1; for (int i = 0; i < k; i++) if (q[i] <= t) c->plain.fast = 0;
        for (int i = 0; i < k; i++) {
            u64 r = 0; for (int l = k - 1; l >= 0; l--) r = (u64)((((u128)r << 64) | quo[l]) % q[i]);
            c->plain.delta[i] = r; c->plain.uhi[i] = rem % q[i]; c->plain.inc[i] = q[i] - t % q[i];
        }
    }
    // BEHZ tables, baseconverter.cpp:104-353
    BehzParams &b = c->behz; memset(&b, 0, sizeof b);
    b.k = k; b.ka = c->ka; b.kb = c->kb; b.t = t; b.m_tilde = kMtilde; b.m_sk = kMsk;
    for (int i = 0; i < k; i++) {
        b.inv_qhat[i] = h_invmod(prod_mod(q, k, i, q[i]), q[i]);
        b.mt_inv_qhat[i] = h_mulmod(b.inv_qhat[i], kMtilde % q[i], q[i]);
        b.qhat_mod_mt[i] = prod_mod(q, k, i, kMtilde);
        for (int j = 0; j < c->kb; j++) b.qhat_mod_bsk[j][i] = prod_mod(q, k, i, bsk[j]);
        for (int j = 0; j < c->ka; j++) b.mhat_mod_q[i][j] = prod_mod(bsk, c->ka, j, q[i]);
        b.M_mod_q[i] = prod_mod(bsk, c->ka, -1, q[i]);
    }
    for (int j = 0; j < c->kb; j++) {
        b.q_mod_bsk[j] = prod_mod(q, k, -1, bsk[j]);
        b.inv_q_mod_bsk[j] = h_invmod(b.q_mod_bsk[j], bsk[j]);
        b.inv_mt_mod_bsk[j] = h_invmod(kMtilde % bsk[j], bsk[j]);
    }
    for (int j = 0; j < c->ka; j++) {
        b.inv_mhat[j] = h_invmod(prod_mod(bsk, c->ka, j, bsk[j]), bsk[j]);
        b.mhat_mod_msk[j] = prod_mod(bsk, c->ka, j, kMsk);
    }
    b.inv_M_mod_msk = h_invmod(prod_mod(bsk, c->ka, -1, kMsk), kMsk);
    b.inv_q_mod_mt = h_invmod(prod_mod(q, k, -1, kMtilde), kMtilde);
    auto shoup = [](u64 w, u64 m) { return (u64)(((u128)w << 64) / m); };
    for (int i = 0; i < k; i++) {
        b.mt_inv_qhat_s[i] = shoup(b.mt_inv_qhat[i], q[i]); b.inv_qhat_s[i] = shoup(b.inv_qhat[i], q[i]);
        b.t_inv_qhat[i] = h_mulmod(t % q[i], b.inv_qhat[i], q[i]); b.t_inv_qhat_s[i] = shoup(b.t_inv_qhat[i], q[i]);
    }
    for (int j = 0; j < c->kb; j++) {
        b.t_mod_bsk[j] = t % bsk[j]; b.t_mod_bsk_s[j] = shoup(b.t_mod_bsk[j], bsk[j]);
        b.inv_mt_mod_bsk_s[j] = shoup(b.inv_mt_mod_bsk[j], bsk[j]); b.inv_q_mod_bsk_s[j] = shoup(b.inv_q_mod_bsk[j], bsk[j]);
    }
    for (int j = 0; j < c->kb; j++) {
        b.lift_r[j] = h_mulmod(b.q_mod_bsk[j], b.inv_mt_mod_bsk[j], bsk[j]);
        b.floor_x[j] = h_mulmod(b.t_mod_bsk[j], b.inv_q_mod_bsk[j], bsk[j]);
        for (int i = 0; i < k; i++) {
            b.lift_c[j][i] = h_mulmod(b.qhat_mod_bsk[j][i], b.inv_mt_mod_bsk[j], bsk[j]);
            b.floor_c[j][i] = negmod(h_mulmod(b.qhat_mod_bsk[j][i], b.inv_q_mod_bsk[j], bsk[j]), bsk[j]);
        }
    }
    for (int j = 0; j < c->ka; j++) b.inv_mhat_s[j] = shoup(b.inv_mhat[j], bsk[j]);
    b.inv_M_mod_msk_s = shoup(b.inv_M_mod_msk, kMsk);
    // decrypt-only constants
    c->tmod = make_mod(t); c->gmod = make_mod(kGamma); c->mtmod = make_mod(kMtilde);
    for (int i = 0; i < k; i++) {
        c->qhat_mod_tg[0][i] = prod_mod(q, k, i, t); c->qhat_mod_tg[1][i] = prod_mod(q, k, i, kGamma);
        c->tgamma_mod_q[i] = h_mulmod(t % q[i], kGamma % q[i], q[i]);
    }
    c->neg_inv_q_mod_tg[0] = h_invmod(negmod(prod_mod(q, k, -1, t), t), t);
    c->neg_inv_q_mod_tg[1] = h_invmod(negmod(prod_mod(q, k, -1, kGamma), kGamma), kGamma);
    c->inv_gamma_mod_t = h_invmod(kGamma % t, t);

    // fp64 NTT primes: the largest primes below 2^47 that are 1 mod 2^16 (so that every ring degree up to 32768 has its 2n-th roots)
    std::vector<std::vector<double>> f64rp(CRC_NF64A), f64irp(CRC_NF64A);
    {
        int found = 0;
        for (u64 cand = ((u64)1 << CRC_F64_PRIME_BITS) - 65536 + 1; found < CRC_NF64A; cand -= 65536) if (is_prime(cand)) c->f64_primes[found++] = cand;
        F64Params &f = c->f64; memset(&f, 0, sizeof f);
        auto centred = [](u64 v, u64 p) { return v > p / 2 ? (double)((long long)v - (long long)p) : (double)v; };
        auto quot = [](double w, u64 p) { return (double)((long double)w / (long double)p); };
        // the square's auxiliary base: the fewest primes with prod p_j >= 4 n t q (1 + 2^-27 + 2^-40) (ctx.h Sq64Params; every p_j > 2^46.9999); transforms on
        // an LDS image of n doubles
        Sq64Params &sq = c->sq64; memset(&sq, 0, sizeof sq);
        if (n <= 16384) {
            // exact, in multi-limb integers: X = 4 n t q (1 + 2^-27 + 2^-40) -- twice the bound 2 n t q (1 + k 2^-32)^2 + k on |floor(t P / q)| of SEAL 2.3.1's
            // non-centred mont_rq, plus fastbconv_sk's own 2 B (1 + #B) -- must not exceed prod p_j
            auto mul = [](std::vector<u64> a, u64 m) { u64 carry = 0; for (auto &l : a) { u128 z = (u128)l * m + carry; l = (u64)z; carry = (u64)(z >> 64);
                } if (carry) a.push_back(carry); return a; };
            auto shr = [](const std::vector<u64> &a, int sh) { std::vector<u64> r(a.size(), 0); const int w = sh / 64, b = sh % 64;
                for (size_t i = w; i < a.size(); i++) { r[i - w] = a[i] >> b; if (b && i + 1 < a.size()) r[i - w] |= a[i + 1] << (64 - b); } return r; };
            auto add = [](std::vector<u64> a, const std::vector<u64> &b) { a.resize(std::max(a.size(), b.size()) + 1, 0); u64 carry = 0;
                for (size_t i = 0; i < a.size(); i++) { u128 z = (u128)a[i] + (i < b.size() ? b[i] : 0) + carry; a[i] = (u64)z; carry = (u64)(z >> 64);
                    } return a; };
            auto geq = [](std::vector<u64> a, std::vector<u64> b) { const size_t m = std::max(a.size(), b.size()); a.resize(m, 0); b.resize(m, 0);
                for (size_t i = m; i-- > 0;) if (a[i] != b[i]) return a[i] > b[i]; return true; };
            std::vector<u64> X = mul(mul(mul(c->qbig, t), (u64)n), 4);
            X = add(add(X, shr(X, 27)), shr(X, 40));
            std::vector<u64> prod{1};
            for (int kf = 1; kf <= CRC_NF64A; kf++) {
                prod = mul(prod, c->f64_primes[kf - 1]);
                if (kf >= 3 && geq(prod, X)) { sq.kf = kf; break; }
            }
        }
        c->nf64 = sq.kf > CRC_NF64 ? sq.kf : CRC_NF64;
        for (int m = 0; m < c->nf64; m++) {
            const u64 p = c->f64_primes[m];
            if (m < CRC_NF64) {
                f.m[m].p = (double)p; f.m[m].pinv = 1.0 / (double)p;
                f.ninv[m] = centred(h_invmod((u64)n, p), p); f.ninv_q[m] = quot(f.ninv[m], p);
            }
            if (device >= 0) {
                const u64 psi = minimal_primitive_root(2 * (u64)n, p);
                if (!psi) { delete c; return CRC_ERR_PARAMETERS; }
                const u64 ipsi = h_invmod(psi, p);
                f64rp[m].assign((size_t)n, 0.0); f64irp[m].assign((size_t)n, 0.0);
                u64 a = 1, b = 1;
                for (int i = 0; i < n; i++) {
                    const u32 j = bitrev((u32)i, c->logn);
                    f64rp[m][j] = centred(a, p); f64irp[m][j] = centred(b, p);
                    a = h_mulmod(a, psi, p); b = h_mulmod(b, ipsi, p);
                }
            }
        }
        f.inv_p0_p1 = centred(h_invmod(c->f64_primes[0] % c->f64_primes[1], c->f64_primes[1]), c->f64_primes[1]);
        f.inv_p0_p1_q = quot(f.inv_p0_p1, c->f64_primes[1]);
        for (int i = 0; i < k; i++) f.p0_mod_q[i] = c->f64_primes[0] % q[i];
        if (sq.kf) {
            const int kf = sq.kf, kB = kf - 1;
            const u64 *P = c->f64_primes, msk = P[kB];
            auto put = [&](double *dst, u64 v, u64 p) { dst[0] = centred(v, p); dst[1] = quot(dst[0], p); };
            const u64 two32 = (u64)1 << 32;
            for (int j = 0; j < kf; j++) {
                const u64 p = P[j];
                sq.m[j].p = (double)p; sq.m[j].pinv = 1.0 / (double)p;
                const u64 q_p = prod_mod(q, k, -1, p), inv_q = h_invmod(q_p, p), inv_mt = h_invmod(kMtilde % p, p);
                put(sq.lift_r[j], h_mulmod(q_p, inv_mt, p), p);
                put(sq.floor_x[j], h_mulmod(h_mulmod(t % p, inv_q, p), h_invmod((u64)n, p), p), p);
                for (int i = 0; i < k; i++) {
                    const u64 qhat = prod_mod(q, k, i, p);
                    const u64 lc = h_mulmod(qhat, inv_mt, p), fc = negmod(h_mulmod(qhat, inv_q, p), p);
                    put(sq.lift_c[j][i], lc, p); put(sq.lift_c[j][i] + 2, h_mulmod(lc, two32 % p, p), p);
                    put(sq.floor_c[j][i], fc, p); put(sq.floor_c[j][i] + 2, h_mulmod(fc, two32 % p, p), p);
                }
                if (j < kB) {
                    put(sq.inv_mhat[j], h_invmod(prod_mod(P, kB, j, p), p), p);
                    put(sq.mhat_msk[j], prod_mod(P, kB, j, msk), msk);
                    for (int i = 0; i < k; i++) sq.mhat_q[i][j] = prod_mod(P, kB, j, q[i]);
                }
            }
            put(sq.inv_B_msk, h_invmod(prod_mod(P, kB, -1, msk), msk), msk);
            for (int i = 0; i < k; i++) sq.B_q[i] = prod_mod(P, kB, -1, q[i]);
        }
    }

    // upload (device < 0: host-only context for encode / client-side use and CPU-only tests of the tables)
    if (device >= 0) {
        int rc = CRC_OK;
        auto fail = [&](hipError_t e) { rc = crc_set_hip_error(e); };
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { fail(e); delete c; return rc; }
        int nm = k + c->kb; size_t tw = (size_t)nm * n * 16;      // {twiddle, Shoup companion} interleaved: one 16-byte load per butterfly
        std::vector<ModParams> mods(nm);
        std::vector<u64> rp((size_t)nm * n * 2), irp2(rp.size()), irp(rp.size());
        for (int m = 0; m < nm; m++) {
            mods[m] = c->tabs[m].m;
            for (int i = 0; i < n; i++) {
                rp[((size_t)m * n + i) * 2] = c->tabs[m].rp[i]; rp[((size_t)m * n + i) * 2 + 1] = c->tabs[m].srp[i];
                irp2[((size_t)m * n + i) * 2] = c->tabs[m].irp2[i]; irp2[((size_t)m * n + i) * 2 + 1] = c->tabs[m].sirp2[i];
                irp[((size_t)m * n + i) * 2] = c->tabs[m].irp[i]; irp[((size_t)m * n + i) * 2 + 1] = c->tabs[m].sirp[i];
            }
        }
        if ((e = hipMalloc(&c->d_mods, sizeof(ModParams) * nm)) != hipSuccess || (e = hipMalloc(&c->d_rp, tw)) != hipSuccess ||
            (e = hipMalloc(&c->d_irp2, tw)) != hipSuccess || (e = hipMalloc(&c->d_irp, tw)) != hipSuccess || (e = hipMalloc(&c->d_behz, sizeof(BehzParams))) != hipSuccess ||
            (e = hipMalloc(&c->d_zero, 8192)) != hipSuccess || (e = hipMemset(c->d_zero, 0, 8192)) != hipSuccess ||
            (e = hipMemcpy(c->d_mods, mods.data(), sizeof(ModParams) * nm, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(c->d_rp, rp.data(), tw, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(c->d_irp2, irp2.data(), tw, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(c->d_irp, irp.data(), tw, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemcpy(c->d_behz, &c->behz, sizeof(BehzParams), hipMemcpyHostToDevice)) != hipSuccess) {
            fail(e); crc_ctx_destroy(c); return rc;
        }
        const size_t ftw = (size_t)n * 8;
        if ((e = hipMalloc(&c->d_f64_rp, ftw * c->nf64)) != hipSuccess || (e = hipMalloc(&c->d_f64_irp, ftw * c->nf64)) != hipSuccess ||
            (e = hipMalloc(&c->d_sq64, sizeof(Sq64Params))) != hipSuccess || (e = hipMemcpy(c->d_sq64, &c->sq64, sizeof(Sq64Params),
                hipMemcpyHostToDevice)) != hipSuccess) { fail(e); crc_ctx_destroy(c); return rc; }
        for (int m = 0; m < c->nf64; m++)
            if ((e = hipMemcpy((char *)c->d_f64_rp + m * ftw, f64rp[m].data(), ftw, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy((char *)c->d_f64_irp + m * ftw, f64irp[m].data(), ftw, hipMemcpyHostToDevice)) != hipSuccess) { fail(e); crc_ctx_destroy(c);
                    return rc; }
        c->d_scratch = c->d_zero + 512;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
    }
    *out = c;
    return CRC_OK;
}

extern "C" void crc_ctx_destroy(crc_ctx *c)
{
    if (!c) return;
    if (c->device >= 0) {
        (void)hipFree(c->d_mods); (void)hipFree(c->d_rp); (void)hipFree(c->d_irp2); (void)hipFree(c->d_irp); (void)hipFree(c->d_behz); (void)hipFree(c->d_zero);
        (void)hipFree(c->d_f64_rp); (void)hipFree(c->d_f64_irp); (void)hipFree(c->d_sq64);
    }
    delete c;
}

// raise a kernel's dynamic-LDS limit (needed above 64 KiB) once per (context = device, kernel)
int crc_ctx_ensure_lds(crc_ctx *c, const void *kernel, size_t lds_bytes)
{
    if (lds_bytes <= 64 * 1024) return CRC_OK;
    std::lock_guard<std::mutex> g(c->attr_mu);
    auto it = c->lds_attr.find(kernel);
    if (it != c->lds_attr.end() && it->second >= lds_bytes) return CRC_OK;
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    c->lds_attr[kernel] = lds_bytes;
    return CRC_OK;
}

extern "C" int crc_ctx_n(const crc_ctx *c) { return c->n; }
extern "C" int crc_ctx_k(const crc_ctx *c) { return c->k; }
extern "C" int crc_ctx_kbsk(const crc_ctx *c) { return c->kb; }
extern "C" int crc_ctx_device(const crc_ctx *c) { return c->device; }
extern "C" size_t crc_ct_words(const crc_ctx *c, int size) { return (size_t)size * c->k * c->n; }
extern "C" size_t crc_evk_words(const crc_ctx *c, int dbc)
{
    size_t w = 0;
    for (int l = 0; l < c->k; l++) w += (size_t)2 * evk_digits(c->q[l], dbc) * c->k * c->n;
    return w;
}

extern "C" int crc_ctx_table(const crc_ctx *c, const char *name, uint64_t *out, int cap)
{
    std::vector<u64> v; std::string s(name);
    int k = c->k, kb = c->kb;
    if (s == "q") for (int i = 0; i < k; i++) v.push_back(c->q[i]);
    else if (s == "root") for (int i = 0; i < k; i++) v.push_back(c->tabs[i].root);
    else if (s == "const_ratio") for (int i = 0; i < k; i++) { v.push_back(c->tabs[i].m.r0); v.push_back(c->tabs[i].m.r1); }
    else if (s == "delta") for (int i = 0; i < k; i++) v.push_back(c->plain.delta[i]);
    else if (s == "upper_half_increment") for (int i = 0; i < k; i++) v.push_back(c->plain.uhi[i]);
    else if (s == "bsk") for (int j = 0; j < kb; j++) v.push_back(c->tabs[k + j].m.q);
    else if (s == "bsk_root") for (int j = 0; j < kb; j++) v.push_back(c->tabs[k + j].root);
    else if (s == "f64_primes") for (int m = 0; m < CRC_NF64; m++) v.push_back(c->f64_primes[m]);
    else if (s == "sq64_primes") for (int m = 0; m < c->sq64.kf; m++) v.push_back(c->f64_primes[m]);
    else if (s.rfind("root_powers:", 0) == 0) { int mi = atoi(name + 12); if (mi < 0 || mi >= k + kb) return CRC_ERR_INVALID_ARGUMENT; v = c->tabs[mi].rp; }
    else if (s.rfind("inv_root_powers_div_two:", 0) == 0) { int mi = atoi(name + 24); if (mi < 0 || mi >= k + kb) return CRC_ERR_INVALID_ARGUMENT;
        v = c->tabs[mi].irp2; }
    else return CRC_ERR_NOT_FOUND;
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}

// ---- device memory helpers ----------------------------------------------------------------------------------
extern "C" int crc_mem_info(crc_ctx *c, size_t *free_bytes, size_t *total_bytes)
{
    if (!c || c->device < 0 || !free_bytes || !total_bytes) return CRC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device)); HIPCHK(hipMemGetInfo(free_bytes, total_bytes)); return CRC_OK;
}
extern "C" int crc_malloc(crc_ctx *c, size_t bytes, void **p)
{
    HIPCHK(hipSetDevice(c->device));
    if (hipMalloc(p, bytes) != hipSuccess) {
        (void)hipGetLastError();        // a caller may free something and try again: do not leave the failure for the next launch check to find
        *p = nullptr;
        return CRC_ERR_HIP;
    }
    return CRC_OK;
}
extern "C" int crc_free(crc_ctx *c, void *p) { (void)c; HIPCHK(hipFree(p)); return CRC_OK; }
extern "C" int crc_memcpy_h2d(crc_ctx *c, void *d, const void *h, size_t b, void *s) { (void)c; HIPCHK(hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice,
    (hipStream_t)s)); return CRC_OK; }
extern "C" int crc_memcpy_d2h(crc_ctx *c, void *h, const void *d, size_t b, void *s) { (void)c; HIPCHK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost,
    (hipStream_t)s)); return CRC_OK; }
extern "C" int crc_memcpy_d2d(crc_ctx *c, void *d, const void *s0, size_t b, void *s) { (void)c; HIPCHK(hipMemcpyAsync(d, s0, b, hipMemcpyDeviceToDevice,
    (hipStream_t)s)); return CRC_OK; }
extern "C" int crc_memset(crc_ctx *c, void *d, int v, size_t b, void *s) { (void)c; HIPCHK(hipMemsetAsync(d, v, b, (hipStream_t)s)); return CRC_OK; }
extern "C" int crc_stream_sync(crc_ctx *c, void *s) { (void)c; HIPCHK(hipStreamSynchronize((hipStream_t)s)); return CRC_OK; }
// Streams of the caller's own: non-blocking ones (no implicit ordering against the default stream), so that a host can put the next chunk's upload beside the
// current chunk's kernels; ordered with events (crc_event_record on one, crc_stream_wait_event on the other)
extern "C" int crc_stream_create(crc_ctx *c, void **stream)
{
    if (!c || !stream || c->device < 0) return CRC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s; HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); *stream = (void *)s; return CRC_OK;
}
extern "C" int crc_stream_destroy(crc_ctx *c, void *stream) { (void)c; if (!stream) return CRC_OK; HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return CRC_OK; }
extern "C" int crc_stream_wait_event(crc_ctx *c, void *stream, void *ev)
{
    (void)c; if (!ev) return CRC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); return CRC_OK;
}
// page-locked host memory: what an asynchronous upload needs to run beside kernels at the link's rate
extern "C" int crc_host_alloc(crc_ctx *c, size_t bytes, void **h_ptr)
{
    if (!c || !h_ptr || c->device < 0) return CRC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    if (hipHostMalloc(h_ptr, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *h_ptr = nullptr; return CRC_ERR_HIP; }
    return CRC_OK;
}
extern "C" int crc_host_free(crc_ctx *c, void *h_ptr) { (void)c; if (!h_ptr) return CRC_OK; HIPCHK(hipHostFree(h_ptr)); return CRC_OK; }
// threads the host-side item loops of this process use (csrc/host_parallel.h: CRC_HOST_THREADS, else the hardware's, at most 16, divided by the ranks of the
// node)
extern "C" int crc_host_thread_limit(void) { return crc_host::thread_limit(); }
// HIP events for hosts that do not link HIP themselves (the C++ host classes time their layers with them: on the stream the kernels are launched on, no
// synchronisation between layers)
extern "C" int crc_event_create(crc_ctx *c, void **ev) { if (!c || !ev || c->device < 0) return CRC_ERR_INVALID_ARGUMENT; hipEvent_t e;
    HIPCHK(hipEventCreate(&e)); *ev = (void *)e; return CRC_OK; }
extern "C" int crc_event_destroy(crc_ctx *c, void *ev) { (void)c; if (!ev) return CRC_OK; HIPCHK(hipEventDestroy((hipEvent_t)ev)); return CRC_OK; }
extern "C" int crc_event_record(crc_ctx *c, void *ev, void *s) { (void)c; if (!ev) return CRC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); return CRC_OK; }
extern "C" int crc_event_elapsed_ms(crc_ctx *c, void *ev0, void *ev1, float *ms)
{
    (void)c; if (!ev0 || !ev1 || !ms) return CRC_ERR_INVALID_ARGUMENT;
    HIPCHK(hipEventSynchronize((hipEvent_t)ev1));
    HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)ev0, (hipEvent_t)ev1));
    return CRC_OK;
}

extern "C" int crc_import_seal(const crc_ctx *c, const uint64_t *seal, int size, uint64_t *out)
{
    for (int r = 0; r < size * c->k; r++) {
        if (seal[(size_t)r * (c->n + 1) + c->n] != 0) return CRC_ERR_INVALID_ARGUMENT;
        memcpy(out + (size_t)r * c->n, seal + (size_t)r * (c->n + 1), 8 * (size_t)c->n);
    }
    return CRC_OK;
}
extern "C" int crc_export_seal(const crc_ctx *c, const uint64_t *in, int size, uint64_t *seal)
{
    for (int r = 0; r < size * c->k; r++) {
        memcpy(seal + (size_t)r * (c->n + 1), in + (size_t)r * c->n, 8 * (size_t)c->n);
        seal[(size_t)r * (c->n + 1) + c->n] = 0;
    }
    return CRC_OK;
}
