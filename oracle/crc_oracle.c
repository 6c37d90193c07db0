/*
 * crc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See crc_oracle.h for the rules.
 *
 * Restatement in plain C of the reference's encrypted-CNN evaluation path: SEAL 2.3.1 BFV (full-RNS / BEHZ)
 * evaluator primitives and the CrCNN layer loops, in the reference's own operation order.  Every function cites
 * the reference file:line it follows (paths relative to /root/reference; SEAL = SEAL_2.3.1/SEAL/seal).
 * Nothing here is copied: the algorithms are re-expressed on flat uint64 arrays without SEAL's classes or pools.
 */
#include "crc_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------------------------
 * scalar modular arithmetic
 * ---------------------------------------------------------------------------------------------------------- */

/* SEAL/smallmodulus.cpp:42-76 : const_ratio = floor(2^128 / q) (two words) and 2^128 mod q (third word). */
void orc_const_ratio(u64 q, u64 ratio[3])
{
    u128 all = ~(u128)0;               /* 2^128 - 1 */
    u128 quo = all / q;
    u128 rem = all % q + 1;            /* remainder of 2^128 */
    if (rem == q) { quo += 1; rem = 0; }
    ratio[0] = (u64)quo;
    ratio[1] = (u64)(quo >> 64);
    ratio[2] = (u64)rem;
}

typedef struct { u64 q; u64 r0, r1; int bits; } mod_t;

static int sig_bits(u64 v) { int b = 0; while (v) { b++; v >>= 1; } return b; }

static mod_t mod_make(u64 q)
{
    mod_t m; u64 r[3];
    orc_const_ratio(q, r);
    m.q = q; m.r0 = r[0]; m.r1 = r[1]; m.bits = sig_bits(q);
    return m;
}

/* SEAL/util/uintarithsmallmod.h:137-176 barrett_reduce_128: base-2^64 Barrett on a 128-bit input, one correction. */
static inline u64 barrett128(u64 lo, u64 hi, const mod_t *m)
{
    u64 carry, tmp1, tmp3;
    u128 t2;
    /* round 1 */
    carry = (u64)(((u128)lo * m->r0) >> 64);
    t2 = (u128)lo * m->r1;
    tmp1 = (u64)t2 + carry;
    tmp3 = (u64)(t2 >> 64) + (tmp1 < carry);
    /* round 2 */
    t2 = (u128)hi * m->r0;
    {
        u64 s = tmp1 + (u64)t2;
        carry = (u64)(t2 >> 64) + (s < tmp1);
    }
    /* only the low word of the quotient estimate matters */
    tmp1 = hi * m->r1 + tmp3 + carry;
    tmp3 = lo - tmp1 * m->q;
    return tmp3 - (m->q & (u64)(-(int64_t)(tmp3 >= m->q)));
}

static inline u64 mulmod_m(u64 a, u64 b, const mod_t *m)   /* uintarithsmallmod.h:178-190 */
{
    u128 z = (u128)a * b;
    return barrett128((u64)z, (u64)(z >> 64), m);
}
static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s - (q & (u64)(-(int64_t)(s >= q))); }   /* :92-112 */
static inline u64 submod(u64 a, u64 b, u64 q) { u64 d = a - b; return d + (q & (u64)(-(int64_t)(a < b))); }    /* :114-135 */
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

u64 orc_barrett_reduce_128(u64 lo, u64 hi, u64 q) { mod_t m = mod_make(q); return barrett128(lo, hi, &m); }
u64 orc_mulmod(u64 a, u64 b, u64 q) { mod_t m = mod_make(q); return mulmod_m(a, b, &m); }

static u64 powmod(u64 a, u64 e, const mod_t *m)              /* uintarithsmallmod.cpp:110-150 */
{
    u64 r = 1; a %= m->q;
    while (e) { if (e & 1) r = mulmod_m(r, a, m); a = mulmod_m(a, a, m); e >>= 1; }
    return r;
}

/* modular inverse by extended Euclid (util/numth / try_invert_uint_mod); works for any modulus coprime to a */
static u64 invmod(u64 a, u64 q)
{
    __int128 t = 0, nt = 1, r = q, nr = a % q;
    while (nr) { __int128 qq = r / nr, tmp;
        tmp = t - qq * nt; t = nt; nt = tmp;
        tmp = r - qq * nr; r = nr; nr = tmp; }
    if (t < 0) t += q;
    return (u64)t;
}

/* SEAL/util/uintarithsmallmod.cpp:13-108.  SEAL draws a random primitive root, then walks all `degree` odd powers
 * and keeps the numerically smallest (:83-108); the result does not depend on the starting root, so we start from
 * a deterministic one. */
u64 orc_min_primitive_root(u64 degree, u64 q)
{
    mod_t m = mod_make(q);
    if ((q - 1) % degree) return 0;
    u64 quot = (q - 1) / degree, root = 0;
    for (u64 g = 2; g < 1000; g++) {
        u64 cand = powmod(g, quot, &m);
        if (powmod(cand, degree >> 1, &m) == q - 1) { root = cand; break; }   /* is_primitive_root :13-38 */
    }
    if (!root) return 0;
    u64 gen_sq = mulmod_m(root, root, &m), cur = root, best = root;
    for (u64 i = 0; i < degree; i++) {
        if (cur < best) best = cur;
        cur = mulmod_m(cur, gen_sq, &m);
    }
    return best;
}

/* ------------------------------------------------------------------------------------------------------------
 * NTT tables and transforms
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    mod_t m; int logn, n; u64 root;
    u64 *rp, *srp;        /* root_powers (bit-reversed) and Shoup-scaled copies      smallntt.cpp:66-68,162-184 */
    u64 *irp2, *sirp2;    /* inv_root_powers_div_two and scaled                      smallntt.cpp:74-79          */
    u64 inv_n;
} ntt_t;

static uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static u64 shoup(u64 x, u64 q) { return (u64)(((u128)x << 64) / q); }   /* smallntt.cpp:175-184 */

static int ntt_make(ntt_t *t, int logn, u64 q)
{
    memset(t, 0, sizeof *t);
    t->m = mod_make(q); t->logn = logn; t->n = 1 << logn;
    int n = t->n;
    t->root = orc_min_primitive_root(2 * (u64)n, q);
    if (!t->root) return -1;
    u64 iroot = invmod(t->root, q);
    t->rp = malloc(8 * n); t->srp = malloc(8 * n); t->irp2 = malloc(8 * n); t->sirp2 = malloc(8 * n);
    u64 *irp = malloc(8 * n);
    /* smallntt.cpp:162-173: powers stored at bit-reversed positions */
    u64 p = 1, ip = 1;
    t->rp[0] = 1; irp[0] = 1;
    for (int i = 1; i < n; i++) {
        p = mulmod_m(p, t->root, &t->m); ip = mulmod_m(ip, iroot, &t->m);
        t->rp[bitrev(i, logn)] = p; irp[bitrev(i, logn)] = ip;
    }
    for (int i = 0; i < n; i++) {
        u64 v = irp[i];                                   /* div2_uint_mod, uintarithsmallmod.h:68-90 */
        t->irp2[i] = (v & 1) ? (u64)(((u128)v + q) >> 1) : v >> 1;
        t->srp[i] = shoup(t->rp[i], q);
        t->sirp2[i] = shoup(t->irp2[i], q);
    }
    free(irp);
    t->inv_n = invmod((u64)n, q);
    return 0;
}
static void ntt_free(ntt_t *t) { free(t->rp); free(t->srp); free(t->irp2); free(t->sirp2); }

/* smallntt.cpp:195-273 : Harvey lazy forward butterflies, values stay in [0,4q) */
static void ntt_fwd_lazy(u64 *a, const ntt_t *T)
{
    u64 q = T->m.q, q2 = 2 * q;
    int n = T->n, t = n >> 1;
    for (int m = 1; m < n; m <<= 1) {
        for (int i = 0; i < m; i++) {
            int j1 = 2 * i * t, j2 = j1 + t;
            u64 W = T->rp[m + i], Wp = T->srp[m + i];
            for (int j = j1; j < j2; j++) {
                u64 X = a[j], Y = a[j + t];
                u64 cx = X - (q2 & (u64)(-(int64_t)(X >= q2)));
                u64 Q = (u64)(((u128)Wp * Y) >> 64);
                Q = Y * W - Q * q;
                a[j] = cx + Q;
                a[j + t] = cx + (q2 - Q);
            }
        }
        t >>= 1;
    }
}
/* smallntt.h:210-234 */
static void ntt_fwd(u64 *a, const ntt_t *T)
{
    ntt_fwd_lazy(a, T);
    u64 q = T->m.q, q2 = 2 * q;
    for (int i = 0; i < T->n; i++) { if (a[i] >= q2) a[i] -= q2; if (a[i] >= q) a[i] -= q; }
}
/* smallntt.cpp:276-375 : Gentleman-Sande with n^-1 folded in through the /2 tables, values in [0,2q) */
static void ntt_inv_lazy(u64 *a, const ntt_t *T)
{
    u64 q = T->m.q, q2 = 2 * q;
    int n = T->n, t = 1;
    for (int m = n; m > 1; m >>= 1) {
        int j1 = 0, h = m >> 1;
        for (int i = 0; i < h; i++) {
            int j2 = j1 + t;
            u64 W = T->irp2[h + i], Wp = T->sirp2[h + i];
            for (int j = j1; j < j2; j++) {
                u64 U = a[j], V = a[j + t];
                u64 Tt = q2 - V + U;
                u64 cu = U + V - (q2 & (u64)(-(int64_t)((U << 1) >= Tt)));
                a[j] = (cu + (q & (u64)(-(int64_t)(Tt & 1)))) >> 1;
                u64 H = (u64)(((u128)Wp * Tt) >> 64);
                a[j + t] = Tt * W - H * q;
            }
            j1 += t << 1;
        }
        t <<= 1;
    }
}
/* smallntt.h:239-258 */
static void ntt_inv(u64 *a, const ntt_t *T)
{
    ntt_inv_lazy(a, T);
    u64 q = T->m.q;
    for (int i = 0; i < T->n; i++) if (a[i] >= q) a[i] -= q;
}

/* ------------------------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------------------------- */
#define MAXK 16
#define MAXB (MAXK + 2)

/* SEAL/util/globals.cpp:321-367 (constants) */
static const u64 AUX_MODS[] = {
    0x1fffffffffb40001ULL, 0x1fffffffff500001ULL, 0x1fffffffff380001ULL, 0x1fffffffff000001ULL,
    0x1ffffffffef00001ULL, 0x1ffffffffee80001ULL, 0x1ffffffffeb40001ULL, 0x1ffffffffe780001ULL,
    0x1ffffffffe600001ULL, 0x1ffffffffe4c0001ULL, 0x1ffffffffdf40001ULL, 0x1ffffffffdac0001ULL,
    0x1ffffffffda40001ULL, 0x1ffffffffc680001ULL, 0x1ffffffffc000001ULL, 0x1ffffffffb880001ULL,
    0x1ffffffffb7c0001ULL };
static const u64 M_SK = 0x1fffffffffe00001ULL, M_TILDE = 1ULL << 32, GAMMA = 0x1fffffffffc80001ULL;

struct orc_ctx {
    int n, logn, k; u64 t; int t_bits;
    ntt_t qn[MAXK];
    /* Evaluator ctor constants, evaluator.cpp:66-105 */
    u64 delta[MAXK];      /* floor(q/t) mod q_i           coeff_div_plain_modulus_        */
    u64 uhi[MAXK];        /* (q - t*floor(q/t)) mod q_i   upper_half_increment_           */
    u64 inc[MAXK];        /* q_i - t                      plain_upper_half_increment_array_ (fast plain lift only) */
    int fast_lift;        /* every q_i > t                context.cpp:156-165 */
    u64 inc_big[MAXK];    /* q - t, multi-word            plain_upper_half_increment_ (evaluator.cpp:74-77) */
    u64 threshold;        /* (t+1)>>1 */
    int total_bits;       /* significant bits of q */
    u64 qbig[MAXK];       /* q as little-endian limbs */
    /* BaseConverter, baseconverter.cpp:20-353 */
    int ka, kb;                       /* aux_base_mod_count_, bsk_base_mod_count_ = ka+1 */
    ntt_t bn[MAXB];                   /* Bsk NTT tables (aux..., m_sk) */
    mod_t msk, mtilde, gamma, tmod;
    u64 inv_qhat[MAXK];               /* (q/q_i)^-1 mod q_i                   inv_coeff_base_products_mod_coeff_array_ */
    u64 mt_inv_qhat[MAXK];            /* m~ * (q/q_i)^-1 mod q_i              mtilde_inv_coeff_base_products_mod_coeff_array_ */
    u64 qhat_mod_bsk[MAXB][MAXK];     /* (q/q_i) mod Bsk_j                    coeff_base_products_mod_aux_bsk_array_ */
    u64 qhat_mod_mt[MAXK];            /* (q/q_i) mod m~                       coeff_base_products_mod_mtilde_array_ */
    u64 inv_q_mod_mt;                 /* q^-1 mod m~                          inv_coeff_products_mod_mtilde_ */
    u64 q_mod_bsk[MAXB];              /* q mod Bsk_j                          coeff_products_all_mod_bsk_array_ */
    u64 inv_mt_mod_bsk[MAXB];         /* m~^-1 mod Bsk_j                      inv_mtilde_mod_bsk_array_ */
    u64 inv_q_mod_bsk[MAXB];          /* q^-1 mod Bsk_j                       inv_coeff_products_all_mod_aux_bsk_array_ */
    u64 inv_mhat[MAXB];               /* (M/m_j)^-1 mod m_j                   inv_aux_base_products_mod_aux_array_ */
    u64 mhat_mod_q[MAXK][MAXB];       /* (M/m_j) mod q_i                      aux_base_products_mod_coeff_array_ */
    u64 mhat_mod_msk[MAXB];           /* (M/m_j) mod m_sk                     aux_base_products_mod_msk_array_ */
    u64 inv_M_mod_msk;                /* M^-1 mod m_sk                        inv_aux_products_mod_msk_ */
    u64 M_mod_q[MAXK];                /* M mod q_i                            aux_products_all_mod_coeff_array_ */
    /* decryption (plain, gamma) */
    u64 qhat_mod_tg[2][MAXK];         /* (q/q_i) mod {t, gamma}               coeff_products_mod_plain_gamma_array_ */
    u64 neg_inv_q_mod_tg[2];          /* (-q)^-1 mod {t, gamma}               neg_inv_coeff_products_all_mod_plain_gamma_array_ */
    u64 inv_gamma_mod_t;              /* gamma^-1 mod t */
    u64 tgamma_mod_q[MAXK];           /* t*gamma mod q_i                      plain_gamma_product_mod_coeff_array_ */
};

static u64 prod_mod_except(const u64 *v, int cnt, int skip, const mod_t *m)
{
    u64 r = 1 % m->q;
    for (int j = 0; j < cnt; j++) if (j != skip) r = mulmod_m(r, v[j] % m->q, m);
    return r;
}

/* little-endian multi-limb helpers (only used for floor(q/t) and the noise budget) */
static void big_mul_u64(u64 *a, int limbs, u64 b)
{
    u64 carry = 0;
    for (int i = 0; i < limbs; i++) { u128 z = (u128)a[i] * b + carry; a[i] = (u64)z; carry = (u64)(z >> 64); }
}
static u64 big_divmod_u64(u64 *a, int limbs, u64 d)        /* a /= d, returns remainder */
{
    u64 rem = 0;
    for (int i = limbs - 1; i >= 0; i--) { u128 z = ((u128)rem << 64) | a[i]; a[i] = (u64)(z / d); rem = (u64)(z % d); }
    return rem;
}
static u64 big_mod_u64(const u64 *a, int limbs, u64 d)
{
    u64 rem = 0;
    for (int i = limbs - 1; i >= 0; i--) { u128 z = ((u128)rem << 64) | a[i]; rem = (u64)(z % d); }
    return rem;
}
static int big_bits(const u64 *a, int limbs)
{
    for (int i = limbs - 1; i >= 0; i--) if (a[i]) return 64 * i + sig_bits(a[i]);
    return 0;
}
static int big_cmp(const u64 *a, const u64 *b, int limbs)
{
    for (int i = limbs - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; }
    return 0;
}
static void big_add(u64 *a, const u64 *b, int limbs)
{
    u64 c = 0;
    for (int i = 0; i < limbs; i++) { u128 z = (u128)a[i] + b[i] + c; a[i] = (u64)z; c = (u64)(z >> 64); }
}
static void big_sub(u64 *a, const u64 *b, int limbs)
{
    u64 br = 0;
    for (int i = 0; i < limbs; i++) { u128 z = (u128)a[i] - b[i] - br; a[i] = (u64)z; br = (u64)((z >> 64) & 1); }
}

orc_ctx *orc_ctx_create(int n, const u64 *q, int k, u64 t)
{
    if (k < 1 || k > MAXK - 1 || n < 2 || (n & (n - 1))) return NULL;
    orc_ctx *c = calloc(1, sizeof *c);
    c->n = n; c->k = k; c->t = t; c->t_bits = sig_bits(t);
    while ((1 << c->logn) < n) c->logn++;
    for (int i = 0; i < k; i++) {
        if (ntt_make(&c->qn[i], c->logn, q[i])) { free(c); return NULL; }
    }
    /* q, floor(q/t), q mod t  -- evaluator.cpp:66-105 */
    u64 big[MAXK] = {0}, quo[MAXK];
    big[0] = 1;
    for (int i = 0; i < k; i++) big_mul_u64(big, k, q[i]);
    memcpy(c->qbig, big, sizeof big);
    c->total_bits = big_bits(big, k);
    memcpy(quo, big, sizeof big);
    u64 q_mod_t = big_divmod_u64(quo, k, t);              /* upper_half_increment = q - t*floor(q/t) = q mod t */
    for (int i = 0; i < k; i++) {
        c->delta[i] = big_mod_u64(quo, k, q[i]);
        c->uhi[i] = q_mod_t % q[i];
        c->inc[i] = q[i] - t;                              /* only meaningful (and only used) when q_i > t */
    }
    c->fast_lift = 1;
    for (int i = 0; i < k; i++) if (q[i] <= t) c->fast_lift = 0;                        /* context.cpp:156-165 */
    { u64 tb[MAXK] = {0}; tb[0] = t; memcpy(c->inc_big, big, sizeof big); big_sub(c->inc_big, tb, k); }     /* evaluator.cpp:74-77 */
    c->threshold = (t + 1) >> 1;

    /* BaseConverter ctor -- baseconverter.cpp:20-353 */
    int total = 0;
    for (int i = 0; i < k; i++) total += c->qn[i].m.bits;
    c->ka = k;
    if (32 + c->t_bits + total >= 61 * k + 61) c->ka++;     /* :47-56 */
    c->kb = c->ka + 1;
    c->msk = mod_make(M_SK); c->mtilde = mod_make(M_TILDE); c->gamma = mod_make(GAMMA); c->tmod = mod_make(t);
    u64 bsk[MAXB];
    for (int j = 0; j < c->ka; j++) bsk[j] = AUX_MODS[j];
    bsk[c->ka] = M_SK;
    for (int j = 0; j < c->kb; j++) if (ntt_make(&c->bn[j], c->logn, bsk[j])) { free(c); return NULL; }
    for (int i = 0; i < k; i++) {
        const mod_t *mi = &c->qn[i].m;
        c->inv_qhat[i] = invmod(prod_mod_except(q, k, i, mi), q[i]);
        c->mt_inv_qhat[i] = mulmod_m(c->inv_qhat[i], M_TILDE, mi);
        c->qhat_mod_mt[i] = prod_mod_except(q, k, i, &c->mtilde);
        for (int j = 0; j < c->kb; j++) c->qhat_mod_bsk[j][i] = prod_mod_except(q, k, i, &c->bn[j].m);
        for (int j = 0; j < c->ka; j++) c->mhat_mod_q[i][j] = prod_mod_except(bsk, c->ka, j, mi);
        c->M_mod_q[i] = prod_mod_except(bsk, c->ka, -1, mi);
        c->qhat_mod_tg[0][i] = prod_mod_except(q, k, i, &c->tmod);
        c->qhat_mod_tg[1][i] = prod_mod_except(q, k, i, &c->gamma);
        c->tgamma_mod_q[i] = mulmod_m(t, GAMMA, mi);
    }
    for (int j = 0; j < c->kb; j++) {
        const mod_t *mj = &c->bn[j].m;
        c->q_mod_bsk[j] = prod_mod_except(q, k, -1, mj);
        c->inv_q_mod_bsk[j] = invmod(c->q_mod_bsk[j], mj->q);
        c->inv_mt_mod_bsk[j] = invmod(M_TILDE % mj->q, mj->q);
    }
    for (int j = 0; j < c->ka; j++) {
        c->inv_mhat[j] = invmod(prod_mod_except(bsk, c->ka, j, &c->bn[j].m), bsk[j]);
        c->mhat_mod_msk[j] = prod_mod_except(bsk, c->ka, j, &c->msk);
    }
    c->inv_M_mod_msk = invmod(prod_mod_except(bsk, c->ka, -1, &c->msk), M_SK);
    c->inv_q_mod_mt = invmod(prod_mod_except(q, k, -1, &c->mtilde), M_TILDE);
    {
        u64 qt = prod_mod_except(q, k, -1, &c->tmod), qg = prod_mod_except(q, k, -1, &c->gamma);
        c->neg_inv_q_mod_tg[0] = invmod(negmod(qt, t), t);
        c->neg_inv_q_mod_tg[1] = invmod(negmod(qg, GAMMA), GAMMA);
        c->inv_gamma_mod_t = invmod(GAMMA % t, t);
    }
    return c;
}

void orc_ctx_destroy(orc_ctx *c)
{
    if (!c) return;
    for (int i = 0; i < c->k; i++) ntt_free(&c->qn[i]);
    for (int j = 0; j < c->kb; j++) ntt_free(&c->bn[j]);
    free(c);
}
int orc_ctx_n(const orc_ctx *c) { return c->n; }
int orc_ctx_k(const orc_ctx *c) { return c->k; }
int orc_ctx_kbsk(const orc_ctx *c) { return c->kb; }

static const ntt_t *tab(const orc_ctx *c, int mi) { return mi < c->k ? &c->qn[mi] : &c->bn[mi - c->k]; }

int orc_ctx_table(const orc_ctx *c, const char *name, u64 *out, int cap)
{
    u64 tmp[4 * MAXB * MAXK]; int cnt = 0;
#define PUT(v) tmp[cnt++] = (v)
    if (!strcmp(name, "q")) for (int i = 0; i < c->k; i++) PUT(c->qn[i].m.q);
    else if (!strcmp(name, "const_ratio")) for (int i = 0; i < c->k; i++) { PUT(c->qn[i].m.r0); PUT(c->qn[i].m.r1); }
    else if (!strcmp(name, "root")) for (int i = 0; i < c->k; i++) PUT(c->qn[i].root);
    else if (!strcmp(name, "bsk")) for (int j = 0; j < c->kb; j++) PUT(c->bn[j].m.q);
    else if (!strcmp(name, "bsk_root")) for (int j = 0; j < c->kb; j++) PUT(c->bn[j].root);
    else if (!strcmp(name, "delta")) for (int i = 0; i < c->k; i++) PUT(c->delta[i]);
    else if (!strcmp(name, "upper_half_increment")) for (int i = 0; i < c->k; i++) PUT(c->uhi[i]);
    else if (!strcmp(name, "inv_qhat")) for (int i = 0; i < c->k; i++) PUT(c->inv_qhat[i]);
    else if (!strcmp(name, "inv_q_mod_bsk")) for (int j = 0; j < c->kb; j++) PUT(c->inv_q_mod_bsk[j]);
    else if (!strcmp(name, "inv_n")) for (int i = 0; i < c->k; i++) PUT(c->qn[i].inv_n);
    else if (!strncmp(name, "root_powers", 11) || !strncmp(name, "inv_root_powers_div_two", 23)) {
        /* "root_powers:<mod_index>" */
        const char *p = strchr(name, ':'); int mi = p ? atoi(p + 1) : 0;
        const ntt_t *T = tab(c, mi);
        const u64 *src = name[0] == 'r' ? T->rp : T->irp2;
        int m = T->n < cap ? T->n : cap;
        memcpy(out, src, 8 * (size_t)m);
        return T->n;
    } else return -1;
#undef PUT
    memcpy(out, tmp, 8 * (size_t)(cnt < cap ? cnt : cap));
    return cnt;
}

void orc_ntt_fwd(const orc_ctx *c, int mi, u64 *p) { ntt_fwd(p, tab(c, mi)); }
void orc_ntt_inv(const orc_ctx *c, int mi, u64 *p) { ntt_inv(p, tab(c, mi)); }
/* polyarithsmallmod.h:401-465 dyadic_product_coeffmod */
void orc_dyadic(const orc_ctx *c, int mi, const u64 *a, const u64 *b, u64 *out)
{
    const ntt_t *T = tab(c, mi);
    for (int s = 0; s < c->n; s++) out[s] = mulmod_m(a[s], b[s], &T->m);
}

/* ------------------------------------------------------------------------------------------------------------
 * fractional encoder  (CrCNN/src/globals.cpp:52 instantiates FractionalEncoder(t, x^n+1, 64, 32, 3))
 * ---------------------------------------------------------------------------------------------------------- */
#define ENC_BASE 3
#define ENC_FRAC 32
#define ENC_INT 64

/* BalancedEncoder::encode(int64) for an odd base: encoder.cpp:408-481.  Returns the Plaintext coeff_count. */
static int balanced_encode_int(u64 t, int64_t value, u64 *dst, int cap)
{
    int cc, idx = 0;
    if (value < 0) {
        u64 pos = (u64)(-value);
        cc = (int)(ceil(64.0 / log2((double)ENC_BASE)) + 1);          /* get_significant_bit_count of a negative int64 = 64 (:438-440) */
        for (int i = 0; i < cc && i < cap; i++) dst[i] = 0;
        while (pos) {
            u64 rem = pos % ENC_BASE;
            if (0 < rem && rem <= (ENC_BASE - 1) / 2) dst[idx] = t - rem;
            else if (rem > (ENC_BASE - 1) / 2) dst[idx] = ENC_BASE - rem;
            pos = (pos + ((ENC_BASE - 1) / 2)) / ENC_BASE;
            idx++;
        }
    } else {
        u64 v = (u64)value;
        cc = (int)(ceil((double)sig_bits(v) / log2((double)ENC_BASE)) + 1);  /* :411-413 */
        for (int i = 0; i < cc && i < cap; i++) dst[i] = 0;
        while (v) {
            u64 rem = v % ENC_BASE;
            if (0 < rem && rem <= (ENC_BASE - 1) / 2) dst[idx] = rem;
            else if (rem > (ENC_BASE - 1) / 2) dst[idx] = t - ENC_BASE + rem;
            v = (v + ENC_BASE / 2) / ENC_BASE;
            idx++;
        }
    }
    return cc;
}

/* BalancedFractionalEncoder::encode_odd, encoder.cpp:1013-1076 */
int orc_encode(const orc_ctx *c, double value, u64 *coeffs)
{
    int n = c->n;
    u64 t = c->t;
    u64 ip[80];
    memset(coeffs, 0, 8 * (size_t)n);
    int64_t vi = (int64_t)round(value);
    int icc = balanced_encode_int(t, vi, ip, 80);
    value -= (double)vi;
    if (value == 0) {
        for (int i = 0; i < icc && i < n; i++) coeffs[i] = ip[i];
        return icc;
    }
    /* digit m (weight 3^-m), m=1..32, lands on coefficient n-m with the sign flipped (:1030-1066) */
    for (int i = 0; i < ENC_FRAC; i++) {
        value *= ENC_BASE;
        int sign = (value >= 0 ? 1 : -1);
        int64_t d = (int64_t)(sign * ceil(fabs(value) - 0.5));
        value -= (double)d;
        u64 enc;
        if (d < 0) enc = (u64)(-d);
        else enc = d ? t - (u64)d : 0;
        coeffs[n - 1 - i] = enc;
    }
    for (int i = 0; i < icc; i++) coeffs[i] = ip[i];              /* set_uint_uint(encoded_int ...) :1073 */
    return n + 1;
}

/* BalancedEncoder::decode_int64 on the significant coefficients (encoder.cpp:576-645), overflow checks dropped */
static int64_t balanced_decode(u64 t, const u64 *cf, int count)
{
    u64 neg_thr = (t + 1) >> 1;
    int64_t r = 0;
    int top = count - 1;
    while (top >= 0 && cf[top] == 0) top--;
    for (int i = top; i >= 0; i--) {
        int64_t v = cf[i] >= neg_thr ? -(int64_t)(t - cf[i]) : (int64_t)cf[i];
        r = r * ENC_BASE + v;
    }
    return r;
}
/* BalancedFractionalEncoder::decode, encoder.cpp:1226-1270 */
double orc_decode(const orc_ctx *c, const u64 *coeffs)
{
    int n = c->n;
    int64_t ipart = balanced_decode(c->t, coeffs, ENC_INT);
    double frac = 0;
    for (int i = 0; i < ENC_FRAC; i++) {
        u64 one = coeffs[n - ENC_FRAC + i];
        frac += (double)balanced_decode(c->t, &one, 1);
        frac /= ENC_BASE;
    }
    return (double)ipart - frac;
}

/* ------------------------------------------------------------------------------------------------------------
 * evaluator ops
 * ---------------------------------------------------------------------------------------------------------- */
#define PN(c) ((size_t)(c)->n)
#define CTW(c, size) ((size_t)(size) * (c)->k * (c)->n)

/* evaluator.cpp:1418-1493 (fast-plain-lift branch :1465-1486) */
void orc_plain_to_ntt(const orc_ctx *c, const u64 *plain, u64 *out)
{
    for (int i = 0; i < c->k; i++) {
        u64 *o = out + i * PN(c);
        if (c->fast_lift)
            for (int s = 0; s < c->n; s++) o[s] = plain[s] >= c->threshold ? plain[s] + c->inc[i] : plain[s];
        else                                               /* !enable_fast_plain_lift, evaluator.cpp:1447-1463: multi-word add of q - t, then decompose */
            for (int s = 0; s < c->n; s++) {
                u64 wide[MAXK] = {0}; wide[0] = plain[s];
                if (plain[s] >= c->threshold) big_add(wide, c->inc_big, c->k);
                o[s] = big_mod_u64(wide, c->k, c->qn[i].m.q);
            }
        ntt_fwd(o, &c->qn[i]);
    }
}
void orc_ct_to_ntt(const orc_ctx *c, u64 *ct, int size)      /* evaluator.cpp:1495-1516 */
{
    for (int p = 0; p < size; p++) for (int i = 0; i < c->k; i++) ntt_fwd(ct + (p * c->k + i) * PN(c), &c->qn[i]);
}
void orc_ct_from_ntt(const orc_ctx *c, u64 *ct, int size)    /* evaluator.cpp:1518-1539 */
{
    for (int p = 0; p < size; p++) for (int i = 0; i < c->k; i++) ntt_inv(ct + (p * c->k + i) * PN(c), &c->qn[i]);
}
void orc_multiply_plain_ntt(const orc_ctx *c, u64 *ct, int size, const u64 *w)   /* evaluator.cpp:1541-1585 */
{
    for (int p = 0; p < size; p++) for (int i = 0; i < c->k; i++) {
        u64 *a = ct + (p * c->k + i) * PN(c); const u64 *b = w + i * PN(c);
        for (int s = 0; s < c->n; s++) a[s] = mulmod_m(a[s], b[s], &c->qn[i].m);
    }
}
void orc_add(const orc_ctx *c, u64 *acc, const u64 *b, int size)                 /* evaluator.cpp:254-294 */
{
    for (int p = 0; p < size; p++) for (int i = 0; i < c->k; i++) {
        u64 q = c->qn[i].m.q; size_t o = (p * c->k + i) * PN(c);
        for (int s = 0; s < c->n; s++) acc[o + s] = addmod(acc[o + s], b[o + s], q);
    }
}
/* Delta*m term shared by add_plain / sub_plain / Encryptor::preencrypt (evaluator.cpp:1168-1191, encryptor.cpp:136-166) */
static inline u64 scaled_plain_coeff(const orc_ctx *c, int i, u64 m)
{
    const mod_t *mi = &c->qn[i].m;
    if (m >= c->threshold) {
        u128 z = (u128)c->delta[i] * m + c->uhi[i];
        return barrett128((u64)z, (u64)(z >> 64), mi);
    }
    return mulmod_m(c->delta[i], m, mi);
}
void orc_add_plain(const orc_ctx *c, u64 *ct, const u64 *plain)                  /* evaluator.cpp:1145-1192 */
{
    for (int i = 0; i < c->k; i++) { u64 q = c->qn[i].m.q;
        for (int s = 0; s < c->n; s++) ct[i * PN(c) + s] = addmod(ct[i * PN(c) + s], scaled_plain_coeff(c, i, plain[s]), q); }
}
void orc_sub_plain(const orc_ctx *c, u64 *ct, const u64 *plain)                  /* evaluator.cpp:1194-1241 */
{
    for (int i = 0; i < c->k; i++) { u64 q = c->qn[i].m.q;
        for (int s = 0; s < c->n; s++) ct[i * PN(c) + s] = submod(ct[i * PN(c) + s], scaled_plain_coeff(c, i, plain[s]), q); }
}
/* evaluator.cpp:1343-1415 generic path (lift, NTT(plain) once, then per ct poly: lazy NTT, dyadic, INTT).  The
 * coeff_count==1 scalar path (:1279-1341) multiplies by the same lifted constant and is the same ring element. */
void orc_multiply_plain(const orc_ctx *c, u64 *ct, int size, const u64 *plain)
{
    u64 *w = malloc(8 * CTW(c, 1));
    orc_plain_to_ntt(c, plain, w);
    for (int p = 0; p < size; p++) for (int i = 0; i < c->k; i++) {
        u64 *a = ct + (p * c->k + i) * PN(c); const u64 *b = w + i * PN(c);
        ntt_fwd_lazy(a, &c->qn[i]);
        for (int s = 0; s < c->n; s++) a[s] = mulmod_m(a[s], b[s], &c->qn[i].m);
        ntt_inv(a, &c->qn[i]);
    }
    free(w);
}

/* ---- BEHZ pieces, on one polynomial: input x[k][n] etc.  All follow util/baseconverter.cpp ---- */

/* fastbconv_mtilde :663-742  in: x[k][n] (base q)  out: y[kb+1][n] (Bsk then m~) */
static void fastbconv_mtilde(const orc_ctx *c, const u64 *x, u64 *y)
{
    int n = c->n, k = c->k;
    u64 *tr = malloc(8 * (size_t)n * k);
    for (int i = 0; i < k; i++) for (int s = 0; s < n; s++)
        tr[(size_t)s * k + i] = mulmod_m(x[i * PN(c) + s], c->mt_inv_qhat[i], &c->qn[i].m);
    for (int j = 0; j < c->kb; j++) for (int s = 0; s < n; s++) {
        u128 acc = 0;
        for (int i = 0; i < k; i++) acc += (u128)tr[(size_t)s * k + i] * c->qhat_mod_bsk[j][i];
        y[j * PN(c) + s] = barrett128((u64)acc, (u64)(acc >> 64), &c->bn[j].m);
    }
    for (int s = 0; s < n; s++) {
        u128 acc = 0;
        for (int i = 0; i < k; i++) acc += (u128)tr[(size_t)s * k + i] * c->qhat_mod_mt[i];
        y[c->kb * PN(c) + s] = barrett128((u64)acc, (u64)(acc >> 64), &c->mtilde);
    }
    free(tr);
}
/* mont_rq :581-622  in: y[kb+1][n]  out: z[kb][n] */
static void mont_rq(const orc_ctx *c, const u64 *y, u64 *z)
{
    int n = c->n;
    const u64 *ymt = y + c->kb * PN(c);
    for (int j = 0; j < c->kb; j++) {
        const mod_t *mj = &c->bn[j].m;
        for (int s = 0; s < n; s++) {
            u64 r = mulmod_m(ymt[s], c->inv_q_mod_mt, &c->mtilde);
            r = negmod(r, M_TILDE);
            u128 tmp = (u128)c->q_mod_bsk[j] * r + y[j * PN(c) + s];
            u64 v = barrett128((u64)tmp, (u64)(tmp >> 64), mj);
            z[j * PN(c) + s] = mulmod_m(v, c->inv_mt_mod_bsk[j], mj);
        }
    }
}
/* fastbconv :388-446  in: x[k][n] (base q)  out: y[kb][n] */
static void fastbconv(const orc_ctx *c, const u64 *x, u64 *y)
{
    int n = c->n, k = c->k;
    u64 *tr = malloc(8 * (size_t)n * k);
    for (int i = 0; i < k; i++) for (int s = 0; s < n; s++)
        tr[(size_t)s * k + i] = mulmod_m(x[i * PN(c) + s], c->inv_qhat[i], &c->qn[i].m);
    for (int j = 0; j < c->kb; j++) for (int s = 0; s < n; s++) {
        u128 acc = 0;
        for (int i = 0; i < k; i++) acc += (u128)tr[(size_t)s * k + i] * c->qhat_mod_bsk[j][i];
        y[j * PN(c) + s] = barrett128((u64)acc, (u64)(acc >> 64), &c->bn[j].m);
    }
    free(tr);
}
/* fast_floor :624-661  in: xq[k][n], xb[kb][n]  out: y[kb][n] */
static void fast_floor(const orc_ctx *c, const u64 *xq, const u64 *xb, u64 *y)
{
    fastbconv(c, xq, y);
    for (int j = 0; j < c->kb; j++) {
        const mod_t *mj = &c->bn[j].m;
        for (int s = 0; s < c->n; s++) {
            size_t o = j * PN(c) + s;
            y[o] = mulmod_m(xb[o] + mj->q - y[o], c->inv_q_mod_bsk[j], mj);
        }
    }
}
/* fastbconv_sk :448-579  in: x[kb][n] (Bsk)  out: y[k][n] (base q) */
static void fastbconv_sk(const orc_ctx *c, const u64 *x, u64 *y)
{
    int n = c->n, k = c->k, ka = c->ka;
    u64 *tr = malloc(8 * (size_t)n * ka);
    for (int j = 0; j < ka; j++) for (int s = 0; s < n; s++)
        tr[(size_t)s * ka + j] = mulmod_m(x[j * PN(c) + s], c->inv_mhat[j], &c->bn[j].m);
    for (int i = 0; i < k; i++) for (int s = 0; s < n; s++) {
        u128 acc = 0;
        for (int j = 0; j < ka; j++) acc += (u128)tr[(size_t)s * ka + j] * c->mhat_mod_q[i][j];
        y[i * PN(c) + s] = barrett128((u64)acc, (u64)(acc >> 64), &c->qn[i].m);
    }
    u64 *alpha = malloc(8 * (size_t)n);
    const u64 *xsk = x + ka * PN(c);
    for (int s = 0; s < n; s++) {
        u128 acc = 0;
        for (int j = 0; j < ka; j++) acc += (u128)tr[(size_t)s * ka + j] * c->mhat_mod_msk[j];
        u64 v = barrett128((u64)acc, (u64)(acc >> 64), &c->msk);
        alpha[s] = mulmod_m(v + (M_SK - xsk[s]), c->inv_M_mod_msk, &c->msk);      /* :533-539 */
    }
    u64 half = M_SK >> 1;
    for (int i = 0; i < k; i++) {
        const mod_t *mi = &c->qn[i].m;
        for (int s = 0; s < n; s++) {
            size_t o = i * PN(c) + s;
            u128 z;
            if (alpha[s] > half) z = (u128)c->M_mod_q[i] * (M_SK - alpha[s]) + y[o];     /* :553-559 */
            else z = (u128)(mi->q - c->M_mod_q[i]) * alpha[s] + y[o];                     /* :561-569 */
            y[o] = barrett128((u64)z, (u64)(z >> 64), mi);
        }
    }
    free(alpha); free(tr);
}

/* Evaluator::square, evaluator.cpp:702-884 (size-2 input) */
void orc_square(const orc_ctx *c, const u64 *ct2, u64 *ct3)
{
    int n = c->n, k = c->k, kb = c->kb;
    size_t N = PN(c);
    u64 *ymt = malloc(8 * N * (kb + 1));
    u64 *cq = malloc(8 * N * k * 2), *cb = malloc(8 * N * kb * 2);     /* inputs in q and Bsk, to be NTT'd */
    u64 *dq = malloc(8 * N * k * 3), *db = malloc(8 * N * kb * 3);     /* products */
    /* steps 0+1 :745-751 */
    for (int p = 0; p < 2; p++) {
        fastbconv_mtilde(c, ct2 + p * k * N, ymt);
        mont_rq(c, ymt, cb + p * kb * N);
    }
    memcpy(cq, ct2, 8 * N * k * 2);
    /* :769-779 lazy forward NTTs */
    for (int p = 0; p < 2; p++) {
        for (int i = 0; i < k; i++) ntt_fwd_lazy(cq + (p * k + i) * N, &c->qn[i]);
        for (int j = 0; j < kb; j++) ntt_fwd_lazy(cb + (p * kb + j) * N, &c->bn[j]);
    }
    /* :783-834 c0^2, c1^2, 2*c0*c1 */
    for (int i = 0; i < k; i++) {
        const mod_t *m = &c->qn[i].m; const u64 *a = cq + i * N, *b = cq + (k + i) * N;
        for (int s = 0; s < n; s++) {
            dq[(0 * k + i) * N + s] = mulmod_m(a[s], a[s], m);
            dq[(2 * k + i) * N + s] = mulmod_m(b[s], b[s], m);
            u64 x = mulmod_m(a[s], b[s], m);
            dq[(1 * k + i) * N + s] = addmod(x, x, m->q);
        }
    }
    for (int j = 0; j < kb; j++) {
        const mod_t *m = &c->bn[j].m; const u64 *a = cb + j * N, *b = cb + (kb + j) * N;
        for (int s = 0; s < n; s++) {
            db[(0 * kb + j) * N + s] = mulmod_m(a[s], a[s], m);
            db[(2 * kb + j) * N + s] = mulmod_m(b[s], b[s], m);
            u64 x = mulmod_m(a[s], b[s], m);
            db[(1 * kb + j) * N + s] = addmod(x, x, m->q);
        }
    }
    /* :837-848 lazy inverse NTTs, :856-871 multiply by t */
    for (int p = 0; p < 3; p++) {
        for (int i = 0; i < k; i++) { u64 *a = dq + (p * k + i) * N; ntt_inv_lazy(a, &c->qn[i]);
            for (int s = 0; s < n; s++) a[s] = mulmod_m(a[s], c->t, &c->qn[i].m); }
        for (int j = 0; j < kb; j++) { u64 *a = db + (p * kb + j) * N; ntt_inv_lazy(a, &c->bn[j]);
            for (int s = 0; s < n; s++) a[s] = mulmod_m(a[s], c->t, &c->bn[j].m); }
    }
    /* :875-883 fast_floor then fastbconv_sk */
    u64 *fl = malloc(8 * N * kb);
    for (int p = 0; p < 3; p++) {
        fast_floor(c, dq + p * k * N, db + p * kb * N, fl);
        fastbconv_sk(c, fl, ct3 + p * k * N);
    }
    free(fl); free(ymt); free(cq); free(cb); free(dq); free(db);
}

/* number of dbc-bit digits of q_i: keygenerator.cpp:683-691 */
static int evk_digits(const orc_ctx *c, int i, int dbc) { int L = 0; u64 v = c->qn[i].m.q; while (v) { L++; v >>= dbc; } return L; }
/* evk blob layout: for l in 0..k-1: [2*L_l][k][n]   (= evk.data()[0][l] polys, pad word dropped) */
int orc_ctx_evk_words(const orc_ctx *c, int dbc)
{
    size_t w = 0;
    for (int l = 0; l < c->k; l++) w += (size_t)2 * evk_digits(c, l, dbc) * c->k * c->n;
    return (int)w;
}

/* Evaluator::relinearize_one_step for size 3 -> 2, evaluator.cpp:934-1069 */
void orc_relinearize(const orc_ctx *c, const u64 *ct3, const u64 *evk, int dbc, u64 *ct2)
{
    int n = c->n, k = c->k;
    size_t N = PN(c);
    u128 *w0 = calloc(N * k, sizeof(u128)), *w1 = calloc(N * k, sizeof(u128));
    u64 *e = malloc(8 * N), *dig = malloc(8 * N), *tmp = malloc(8 * N);
    const u64 *c2 = ct3 + 2 * k * N;
    const u64 *key = evk;
    u64 mask = (1ULL << dbc) - 1;
    for (int i = 0; i < k; i++) {
        for (int s = 0; s < n; s++) e[s] = mulmod_m(c2[i * N + s], c->inv_qhat[i], &c->qn[i].m);      /* :984-985 */
        int L = evk_digits(c, i, dbc), shift = 0;
        for (int d = 0; d < L; d++) {
            const u64 *k0 = key + (size_t)(2 * d) * k * N, *k1 = key + (size_t)(2 * d + 1) * k * N;
            for (int s = 0; s < n; s++) dig[s] = (e[s] >> shift) & mask;                               /* :997-1001 */
            for (int j = 0; j < k; j++) {
                memcpy(tmp, dig, 8 * N);
                ntt_fwd_lazy(tmp, &c->qn[j]);                                                          /* :1011 */
                for (int s = 0; s < n; s++) {                                                          /* :1015-1030 */
                    w0[j * N + s] += (u128)tmp[s] * k0[j * N + s];
                    w1[j * N + s] += (u128)tmp[s] * k1[j * N + s];
                }
            }
            shift += dbc;
        }
        key += (size_t)2 * L * k * N;
    }
    memcpy(ct2, ct3, 8 * N * k * 2);
    for (int p = 0; p < 2; p++) {
        u128 *w = p ? w1 : w0;
        for (int i = 0; i < k; i++) {                                                                  /* :1041-1068 */
            for (int s = 0; s < n; s++) tmp[s] = barrett128((u64)w[i * N + s], (u64)(w[i * N + s] >> 64), &c->qn[i].m);
            ntt_inv(tmp, &c->qn[i]);
            u64 *dst = ct2 + (p * k + i) * N; u64 q = c->qn[i].m.q;
            for (int s = 0; s < n; s++) dst[s] = addmod(dst[s], tmp[s], q);
        }
    }
    free(w0); free(w1); free(e); free(dig); free(tmp);
}

/* ------------------------------------------------------------------------------------------------------------
 * client side: keygen / encrypt / decrypt.  Same algorithms as SEAL, own seeded RNG (SEAL's default RNG is
 * std::random_device -- randomgen.cpp:7 -- so no bit pattern is pinned by the reference here; ref_harness checks that
 * SEAL's Decryptor accepts our ciphertexts and keys).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct { u64 s; } rng_t;
static u64 rng_next(rng_t *r) { u64 z = (r->s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static double rng_unit(rng_t *r) { return ((double)(rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

/* R_3 sample in RNS form: encryptor.cpp:168-206 / keygenerator.cpp:467-505 */
static void sample_ternary(const orc_ctx *c, rng_t *r, u64 *p)
{
    for (int s = 0; s < c->n; s++) {
        int v = (int)(rng_next(r) % 3) - 1;
        for (int i = 0; i < c->k; i++) p[i * PN(c) + s] = v == 1 ? 1 : (v == -1 ? c->qn[i].m.q - 1 : 0);
    }
}
/* clipped normal sigma=3.19, |x| <= 6 sigma, truncated toward zero: encryptor.cpp:228-270, util/globals.cpp:13-15 */
static void sample_noise(const orc_ctx *c, rng_t *r, u64 *p)
{
    const double sigma = 3.19, maxdev = 6 * 3.19;
    for (int s = 0; s < c->n; s++) {
        double v;
        do { double u1 = rng_unit(r), u2 = rng_unit(r); v = sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2); } while (fabs(v) > maxdev);
        int64_t e = (int64_t)v;
        for (int i = 0; i < c->k; i++) p[i * PN(c) + s] = e >= 0 ? (u64)e : c->qn[i].m.q - (u64)(-e);
    }
}
static void sample_uniform(const orc_ctx *c, rng_t *r, u64 *p)    /* keygenerator.cpp:544-572 (uniform mod q_i) */
{
    for (int i = 0; i < c->k; i++) for (int s = 0; s < c->n; s++) {
        u128 z = ((u128)rng_next(r) << 64) | rng_next(r);
        p[i * PN(c) + s] = (u64)(z % c->qn[i].m.q);
    }
}

/* KeyGenerator::generate, keygenerator.cpp:96-164: sk in NTT form, pk = (-(a s + e), a) in NTT form */
void orc_keygen(const orc_ctx *c, u64 seed, u64 *sk, u64 *pk)
{
    rng_t r = { seed };
    size_t N = PN(c); int k = c->k;
    u64 *e = malloc(8 * N * k);
    sample_ternary(c, &r, sk);
    sample_uniform(c, &r, pk + k * N);
    sample_noise(c, &r, e);
    for (int i = 0; i < k; i++) {
        const mod_t *m = &c->qn[i].m;
        ntt_fwd(sk + i * N, &c->qn[i]); ntt_fwd(pk + (k + i) * N, &c->qn[i]); ntt_fwd(e + i * N, &c->qn[i]);
        for (int s = 0; s < c->n; s++) {
            u64 v = mulmod_m(sk[i * N + s], pk[(k + i) * N + s], m);
            pk[i * N + s] = negmod(addmod(v, e[i * N + s], m->q), m->q);
        }
    }
    free(e);
}

/* KeyGenerator::generate_evaluation_keys(dbc, count=1), keygenerator.cpp:166-282 + :652-698 */
void orc_gen_evk(const orc_ctx *c, u64 seed, const u64 *sk, int dbc, u64 *evk)
{
    rng_t r = { seed ^ 0xE7A1ULL };
    size_t N = PN(c); int k = c->k, n = c->n;
    u64 *s2 = malloc(8 * N * k), *e = malloc(8 * N * k);
    for (int j = 0; j < k; j++) for (int s = 0; s < n; s++) s2[j * N + s] = mulmod_m(sk[j * N + s], sk[j * N + s], &c->qn[j].m);
    u64 *key = evk;
    for (int l = 0; l < k; l++) {
        const mod_t *ml = &c->qn[l].m;
        u64 factor = 1;                                           /* hat-q_l mod q_l :664-676 */
        for (int j = 0; j < k; j++) if (j != l) factor = mulmod_m(factor, c->qn[j].m.q % ml->q, ml);
        int L = evk_digits(c, l, dbc);
        for (int d = 0; d < L; d++) {
            u64 *first = key + (size_t)(2 * d) * k * N, *second = key + (size_t)(2 * d + 1) * k * N;
            sample_uniform(c, &r, second);
            sample_noise(c, &r, e);
            for (int j = 0; j < k; j++) {
                const mod_t *mj = &c->qn[j].m;
                ntt_fwd(second + j * N, &c->qn[j]); ntt_fwd(e + j * N, &c->qn[j]);
                u64 f = (j == l) ? factor : 0;
                for (int s = 0; s < n; s++) {
                    u64 v = mulmod_m(second[j * N + s], sk[j * N + s], mj);
                    v = negmod(addmod(v, e[j * N + s], mj->q), mj->q);
                    first[j * N + s] = addmod(v, mulmod_m(s2[j * N + s], f, mj), mj->q);
                }
            }
            factor = mulmod_m(factor, 1ULL << dbc, ml);
        }
        key += (size_t)2 * L * k * N;
    }
    free(s2); free(e);
}

/* Encryptor::encrypt, encryptor.cpp:71-134 */
void orc_encrypt(const orc_ctx *c, const u64 *pk, const u64 *plain, u64 seed, u64 *ct)
{
    rng_t r = { seed };
    size_t N = PN(c); int k = c->k, n = c->n;
    u64 *u = malloc(8 * N * k), *e = malloc(8 * N * k);
    sample_ternary(c, &r, u);
    for (int i = 0; i < k; i++) {
        const mod_t *m = &c->qn[i].m;
        ntt_fwd(u + i * N, &c->qn[i]);
        for (int s = 0; s < n; s++) {
            ct[i * N + s] = mulmod_m(u[i * N + s], pk[i * N + s], m);
            ct[(k + i) * N + s] = mulmod_m(u[i * N + s], pk[(k + i) * N + s], m);
        }
        ntt_inv(ct + i * N, &c->qn[i]); ntt_inv(ct + (k + i) * N, &c->qn[i]);
    }
    orc_add_plain(c, ct, plain);                                 /* preencrypt :136-166 */
    for (int p = 0; p < 2; p++) {
        sample_noise(c, &r, e);
        for (int i = 0; i < k; i++) { u64 q = c->qn[i].m.q;
            for (int s = 0; s < n; s++) ct[(p * k + i) * N + s] = addmod(ct[(p * k + i) * N + s], e[i * N + s], q); }
    }
    free(u); free(e);
}

/* c0 + c1 s + c2 s^2 (mod q_i), coefficient form: decryptor.cpp:140-172 */
static void dot_secret(const orc_ctx *c, const u64 *sk, const u64 *ct, int size, u64 *out)
{
    size_t N = PN(c); int k = c->k, n = c->n;
    u64 *tmp = malloc(8 * N), *spow = malloc(8 * N);
    for (int i = 0; i < k; i++) {
        const mod_t *m = &c->qn[i].m;
        u64 *o = out + i * N;
        memset(o, 0, 8 * N);
        memcpy(spow, sk + i * N, 8 * N);
        for (int p = 1; p < size; p++) {
            memcpy(tmp, ct + (p * k + i) * N, 8 * N);
            ntt_fwd(tmp, &c->qn[i]);
            for (int s = 0; s < n; s++) o[s] = addmod(o[s], mulmod_m(tmp[s], spow[s], m), m->q);
            for (int s = 0; s < n; s++) spow[s] = mulmod_m(spow[s], sk[i * N + s], m);
        }
        ntt_inv(o, &c->qn[i]);
        for (int s = 0; s < n; s++) o[s] = addmod(o[s], ct[i * N + s], m->q);
    }
    free(tmp); free(spow);
}

/* Decryptor::decrypt, decryptor.cpp:107-236 (BEHZ gamma-corrected rounding) */
void orc_decrypt(const orc_ctx *c, const u64 *sk, const u64 *ct, int size, u64 *plain)
{
    size_t N = PN(c); int k = c->k, n = c->n;
    u64 *v = malloc(8 * N * k), *tr = malloc(8 * N * k);
    dot_secret(c, sk, ct, size, v);
    for (int i = 0; i < k; i++) for (int s = 0; s < n; s++) {
        u64 x = mulmod_m(v[i * N + s], c->tgamma_mod_q[i], &c->qn[i].m);                /* :172-174 */
        tr[(size_t)s * k + i] = mulmod_m(x, c->inv_qhat[i], &c->qn[i].m);              /* fastbconv_plain_gamma :744-797 */
    }
    const mod_t *tg[2] = { &c->tmod, &c->gamma };
    u64 half = GAMMA >> 1;
    for (int s = 0; s < n; s++) {
        u64 r[2];
        for (int j = 0; j < 2; j++) {
            u128 acc = 0;
            for (int i = 0; i < k; i++) acc += (u128)tr[(size_t)s * k + i] * c->qhat_mod_tg[j][i];
            r[j] = barrett128((u64)acc, (u64)(acc >> 64), tg[j]);
            r[j] = mulmod_m(r[j], c->neg_inv_q_mod_tg[j], tg[j]);                       /* :184-189 */
        }
        u64 w;
        if (r[1] > half) w = addmod(r[0], (GAMMA - r[1]) % c->t, c->t);                 /* :198-206 */
        else w = submod(r[0], r[1] % c->t, c->t);                                       /* :208-214 */
        plain[s] = mulmod_m(w, c->inv_gamma_mod_t, &c->tmod);                           /* :223-235 */
    }
    free(v); free(tr);
}

/* Decryptor::invariant_noise_budget, decryptor.cpp:295-403 (+ compose :262-293) */
int orc_noise_budget(const orc_ctx *c, const u64 *sk, const u64 *ct, int size)
{
    size_t N = PN(c); int k = c->k, n = c->n;
    u64 *v = malloc(8 * N * k);
    dot_secret(c, sk, ct, size, v);
    u64 qhat[MAXK][MAXK];                        /* q/q_i as limbs */
    for (int i = 0; i < k; i++) { memset(qhat[i], 0, sizeof qhat[i]); qhat[i][0] = 1;
        for (int j = 0; j < k; j++) if (j != i) big_mul_u64(qhat[i], k, c->qn[j].m.q); }
    u64 half[MAXK]; memcpy(half, c->qbig, sizeof half);
    { u64 carry = 0; for (int i = k - 1; i >= 0; i--) { u64 nc = half[i] & 1; half[i] = (half[i] >> 1) | (carry << 63); carry = nc; } }
    u64 norm[MAXK] = {0};
    for (int s = 0; s < n; s++) {
        u64 acc[MAXK] = {0};
        for (int i = 0; i < k; i++) {
            u64 x = mulmod_m(v[i * N + s], c->t, &c->qn[i].m);
            x = mulmod_m(x, c->inv_qhat[i], &c->qn[i].m);
            u64 term[MAXK]; memcpy(term, qhat[i], sizeof term);
            big_mul_u64(term, k, x);
            big_add(acc, term, k);
            if (big_cmp(acc, c->qbig, k) >= 0) big_sub(acc, c->qbig, k);
        }
        if (big_cmp(acc, half, k) > 0) { u64 tq[MAXK]; memcpy(tq, c->qbig, sizeof tq); big_sub(tq, acc, k); memcpy(acc, tq, sizeof acc); }
        if (big_cmp(acc, norm, k) > 0) memcpy(norm, acc, sizeof norm);
    }
    free(v);
    int b = c->total_bits - big_bits(norm, k) - 1;
    return b > 0 ? b : 0;
}

/* ------------------------------------------------------------------------------------------------------------
 * CrCNN layers, reference operation order, threaded like the reference (std::thread -> pthread)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct job { void (*fn)(struct job *, int, int); int from, to; void *arg; } job_t;
static void *job_tramp(void *p) { job_t *j = p; j->fn(j, j->from, j->to); return NULL; }
/* split [0,total) into `threads` contiguous chunks, remainder to the last one: convolutionalLayer.cpp:177-191 */
static void run_split(void (*fn)(job_t *, int, int), void *arg, int begin, int end, int threads)
{
    int total = end - begin;
    if (threads > total) threads = total;
    if (threads <= 1) { job_t j = { fn, begin, end, arg }; fn(&j, begin, end); return; }
    pthread_t *th = malloc(sizeof(pthread_t) * threads); job_t *jobs = malloc(sizeof(job_t) * threads);
    int per = total / threads, to = begin;
    for (int i = 0; i < threads; i++) {
        int from = to; to += per; if (i == threads - 1) to += total % threads;
        jobs[i] = (job_t){ fn, from, to, arg };
        pthread_create(&th[i], NULL, job_tramp, &jobs[i]);
    }
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
}

/* Layer::computeBoundaries, layer.cpp:12-26 */
static void boundaries(int xd, int yd, int xs, int ys, int xf, int yf, int *xl, int *yl)
{
    *xl = xf > xs ? xd - xf + 1 : xd - xs + 1;
    *yl = yf > ys ? yd - yf + 1 : yd - ys + 1;
}

typedef struct {
    const orc_ctx *c; const u64 *x; u64 *xn; int zd, xd, yd, xs, ys, xf, yf, nf; const u64 *w, *bias; u64 *y; int fast;
} conv_arg;

static void conv_ntt_job(job_t *j, int from, int to)     /* transform_input_to_ntt, convolutionalLayer.cpp:95-148 */
{
    conv_arg *a = j->arg; size_t ctw = CTW(a->c, 2);
    for (size_t i = from; i < (size_t)to; i++) { memcpy(a->xn + i * ctw, a->x + i * ctw, 8 * ctw); orc_ct_to_ntt(a->c, a->xn + i * ctw, 2); }
}
static void conv_job(job_t *j, int from, int to)         /* convolution3d, convolutionalLayer.cpp:56-93 */
{
    conv_arg *a = j->arg; const orc_ctx *c = a->c; size_t ctw = CTW(c, 2);
    int xo = (a->xd - a->xf) / a->xs + 1, yo = (a->yd - a->yf) / a->ys + 1, xl, yl;
    boundaries(a->xd, a->yd, a->xs, a->ys, a->xf, a->yf, &xl, &yl);
    u64 *prod = malloc(8 * ctw), *acc = malloc(8 * ctw);
    for (int f = from; f < to; f++) {
        const u64 *wf = a->w + (size_t)f * a->zd * a->xf * a->yf * c->k * c->n;
        for (int i = 0; i < xl; i += a->xs) for (int jj = 0; jj < yl; jj += a->ys) {
            int p = 0;
            u64 *out = a->y + (((size_t)f * xo + i / a->xs) * yo + jj / a->ys) * ctw;
            if (a->fast) memset(acc, 0, 8 * ctw);
            for (int z = 0; z < a->zd; z++) for (int kx = 0; kx < a->xf; kx++) for (int ky = 0; ky < a->yf; ky++, p++) {
                const u64 *src = a->xn + (((size_t)z * a->xd + i + kx) * a->yd + jj + ky) * ctw;
                const u64 *wt = wf + (size_t)p * c->k * c->n;
                memcpy(prod, src, 8 * ctw);
                orc_multiply_plain_ntt(c, prod, 2, wt);                     /* :77-79 */
                if (a->fast) { orc_add(c, acc, prod, 2); continue; }
                orc_ct_from_ntt(c, prod, 2);                                /* :81 */
                if (p == 0) { orc_add_plain(c, prod, a->bias + (size_t)f * c->n); memcpy(acc, prod, 8 * ctw); }   /* :87 */
                else orc_add(c, acc, prod, 2);                              /* add_many :88 = left fold, evaluator.cpp:296-308 */
            }
            if (a->fast) { orc_ct_from_ntt(c, acc, 2); orc_add_plain(c, acc, a->bias + (size_t)f * c->n); }
            memcpy(out, acc, 8 * ctw);
        }
    }
    free(prod); free(acc);
}
static void conv_run(const orc_ctx *c, const u64 *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
                     const u64 *w, const u64 *bias, u64 *y, int threads, int fb, int fe, int fast)
{
    conv_arg a = { c, x, NULL, zd, xd, yd, xs, ys, xf, yf, nf, w, bias, y, fast };
    a.xn = malloc(8 * CTW(c, 2) * zd * xd * yd);
    run_split(conv_ntt_job, &a, 0, zd * xd * yd, threads);
    run_split(conv_job, &a, fb, fe, threads);
    free(a.xn);
}
void orc_conv_forward(const orc_ctx *c, const u64 *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
                      const u64 *w, const u64 *bias, u64 *y, int threads, int fb, int fe)
{ conv_run(c, x, zd, xd, yd, xs, ys, xf, yf, nf, w, bias, y, threads, fb, fe, 0); }
void orc_conv_forward_fast(const orc_ctx *c, const u64 *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
                      const u64 *w, const u64 *bias, u64 *y, int threads)
{ conv_run(c, x, zd, xd, yd, xs, ys, xf, yf, nf, w, bias, y, threads, 0, nf, 1); }

/* FullyConnectedLayer::forward, fullyConnectedLayer.cpp:113-168 (input already flattened z,x,y row-major = memory order) */
void orc_fc_forward(const orc_ctx *c, const u64 *x, int in_dim, int out_dim, const u64 *w, const u64 *bias, u64 *y,
                    int threads, int rb, int re)
{
    (void)out_dim;
    /* an FC layer is a 1x1 "convolution" over in_dim channels with out_dim filters */
    conv_run(c, x, in_dim, 1, 1, 1, 1, 1, 1, out_dim, w, bias, y, threads, rb, re, 0);
}

typedef struct { const orc_ctx *c; const u64 *x; u64 *y; int zd, xd, yd, xs, ys, xf, yf; const u64 *div; } pool_arg;
static void pool_job(job_t *j, int from, int to)         /* poolingLayer.cpp:22-44, avgPoolingLayer.cpp:16-45 */
{
    pool_arg *a = j->arg; const orc_ctx *c = a->c; size_t ctw = CTW(c, 2);
    int xo = (a->xd - a->xf) / a->xs + 1, yo = (a->yd - a->yf) / a->ys + 1, xl, yl;
    boundaries(a->xd, a->yd, a->xs, a->ys, a->xf, a->yf, &xl, &yl);
    for (int z = from; z < to; z++) for (int i = 0; i < xl; i += a->xs) for (int jj = 0; jj < yl; jj += a->ys) {
        u64 *out = a->y + (((size_t)z * xo + i / a->xs) * yo + jj / a->ys) * ctw;
        int p = 0;
        for (int kx = 0; kx < a->xf; kx++) for (int ky = 0; ky < a->yf; ky++, p++) {
            const u64 *src = a->x + (((size_t)z * a->xd + i + kx) * a->yd + jj + ky) * ctw;
            if (p == 0) memcpy(out, src, 8 * ctw); else orc_add(c, out, src, 2);
        }
        if (a->div) orc_multiply_plain(c, out, 2, a->div);
    }
}
void orc_pool_forward(const orc_ctx *c, const u64 *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf,
                      const u64 *div, u64 *y, int threads)
{
    pool_arg a = { c, x, y, zd, xd, yd, xs, ys, xf, yf, div };
    run_split(pool_job, &a, 0, zd, threads);
}

typedef struct { const orc_ctx *c; u64 *x; int zd, xd, yd; const u64 *mean, *invstd; } bn_arg;
static void bn_job(job_t *j, int from, int to)           /* batchNormLayer.cpp:29-40 */
{
    bn_arg *a = j->arg; const orc_ctx *c = a->c; size_t ctw = CTW(c, 2);
    for (int z = from; z < to; z++) for (int i = 0; i < a->xd * a->yd; i++) {
        u64 *ct = a->x + ((size_t)z * a->xd * a->yd + i) * ctw;
        orc_sub_plain(c, ct, a->mean + (size_t)z * c->n);
        orc_multiply_plain(c, ct, 2, a->invstd + (size_t)z * c->n);
    }
}
void orc_bn_forward(const orc_ctx *c, u64 *x, int zd, int xd, int yd, const u64 *mean, const u64 *invstd, int threads)
{
    bn_arg a = { c, x, zd, xd, yd, mean, invstd };
    run_split(bn_job, &a, 0, zd, threads);
}

typedef struct { const orc_ctx *c; const u64 *x; u64 *y; const u64 *evk; int dbc; } sq_arg;
static void sq_job(job_t *j, int from, int to)           /* squareLayer.cpp:22-45 */
{
    sq_arg *a = j->arg; const orc_ctx *c = a->c; size_t ctw = CTW(c, 2);
    u64 *c3 = malloc(8 * CTW(c, 3));
    for (size_t i = from; i < (size_t)to; i++) {
        orc_square(c, a->x + i * ctw, c3);
        orc_relinearize(c, c3, a->evk, a->dbc, a->y + i * ctw);
    }
    free(c3);
}
void orc_square_forward(const orc_ctx *c, const u64 *x, size_t count, const u64 *evk, int dbc, u64 *y, int threads)
{
    sq_arg a = { c, x, y, evk, dbc };
    run_split(sq_job, &a, 0, (int)count, threads);
}
