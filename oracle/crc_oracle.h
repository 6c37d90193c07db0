/*
 * crc_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the reference's hot path (CrCNN layers over SEAL 2.3.1 BFV, full-RNS "BEHZ" variant),
 * written from the algorithm, each function citing the reference file:line it follows.  It exists so that the HIP
 * path can be checked bit-for-bit on a machine where /root/reference is absent.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * (crcnn_amd/, include/crcnn_hip.h) never links, imports or falls back to it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against (a) SEAL's own known-answer unit-test
 * values (SEALTest/util/{smallntt,uintarithsmallmod,polyarithsmallmod}.cpp, cited per vector in tests/golden) and
 * (b) outputs of the compiled reference itself (oracle/_ref/ref_harness, built by oracle/Makefile from the
 * sources under /root/reference) on identical keys and ciphertexts, committed as tests/golden/ (.bin files) together with the
 * generating script oracle/make_golden.py.
 *
 * Data layout used everywhere in the oracle (and on the device): a polynomial residue is n uint64 in [0,q_i);
 * a ciphertext of `size` polys is uint64[size][k][n]  (SEAL stores [size][k][n+1] with a dead, always-zero pad
 * word per residue -- ciphertext.cpp:103-130 -- which we drop; ref_harness re-inserts it at the boundary).
 */
#ifndef CRC_ORACLE_H
#define CRC_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* ---- context (SEALContext + Evaluator ctor + BaseConverter ctor: context.cpp:15-169, evaluator.cpp:19-121,
 *      util/baseconverter.cpp:20-353) ---- */
orc_ctx *orc_ctx_create(int n, const uint64_t *q, int k, uint64_t t);
void     orc_ctx_destroy(orc_ctx *c);
/* named table read-out for golden tests: returns the number of words the table has (copies min(count,cap)). */
int      orc_ctx_table(const orc_ctx *c, const char *name, uint64_t *out, int cap);
int      orc_ctx_n(const orc_ctx *c);
int      orc_ctx_k(const orc_ctx *c);
int      orc_ctx_kbsk(const orc_ctx *c);          /* |Bsk| = k_aux + 1 */
int      orc_ctx_evk_words(const orc_ctx *c, int dbc); /* words in an evaluation-key blob for this ctx */

/* ---- scalar modular arithmetic (util/uintarithsmallmod.h:92-190, smallmodulus.cpp:42-76) ---- */
void     orc_const_ratio(uint64_t q, uint64_t ratio[3]);
uint64_t orc_barrett_reduce_128(uint64_t lo, uint64_t hi, uint64_t q);
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t orc_min_primitive_root(uint64_t degree, uint64_t q);   /* uintarithsmallmod.cpp:83-108 */

/* ---- NTT (util/smallntt.cpp:195-375, smallntt.h:210-258); mod_index <k: q_i, >=k: Bsk[mod_index-k] ---- */
void orc_ntt_fwd(const orc_ctx *c, int mod_index, uint64_t *poly);
void orc_ntt_inv(const orc_ctx *c, int mod_index, uint64_t *poly);
void orc_dyadic(const orc_ctx *c, int mod_index, const uint64_t *a, const uint64_t *b, uint64_t *out); /* polyarithsmallmod.h:401-465 */

/* ---- FractionalEncoder(t, x^n+1, 64, 32, base 3): encoder.cpp:1013-1076, 1226-1270, 408-481 ---- */
/* writes n coefficients (zero-extended), returns SEAL's coeff_count for the Plaintext (1..n+1) */
int    orc_encode(const orc_ctx *c, double value, uint64_t *coeffs);
double orc_decode(const orc_ctx *c, const uint64_t *coeffs);

/* ---- Evaluator ops on ciphertexts uint64[size][k][n] ---- */
void orc_plain_to_ntt(const orc_ctx *c, const uint64_t *plain /*[n]*/, uint64_t *out /*[k][n]*/);     /* evaluator.cpp:1418-1493 */
void orc_ct_to_ntt(const orc_ctx *c, uint64_t *ct, int size);                                          /* :1495-1516 */
void orc_ct_from_ntt(const orc_ctx *c, uint64_t *ct, int size);                                        /* :1518-1539 */
void orc_multiply_plain_ntt(const orc_ctx *c, uint64_t *ct, int size, const uint64_t *w_ntt /*[k][n]*/); /* :1541-1585 */
void orc_add(const orc_ctx *c, uint64_t *acc, const uint64_t *b, int size);                            /* :254-294 */
void orc_add_plain(const orc_ctx *c, uint64_t *ct, const uint64_t *plain /*[n]*/);                     /* :1145-1192 */
void orc_sub_plain(const orc_ctx *c, uint64_t *ct, const uint64_t *plain /*[n]*/);                     /* :1194-1241 */
void orc_multiply_plain(const orc_ctx *c, uint64_t *ct, int size, const uint64_t *plain /*[n]*/);      /* :1243-1416 */
void orc_square(const orc_ctx *c, const uint64_t *ct2 /*[2][k][n]*/, uint64_t *ct3 /*[3][k][n]*/);     /* :702-884 */
void orc_relinearize(const orc_ctx *c, const uint64_t *ct3, const uint64_t *evk, int dbc, uint64_t *ct2); /* :886-1069 */

/* ---- client side (keygenerator.cpp:96-282, encryptor.cpp:71-134, decryptor.cpp:107-236); own seeded RNG ---- */
void orc_keygen(const orc_ctx *c, uint64_t seed, uint64_t *sk_ntt /*[k][n]*/, uint64_t *pk /*[2][k][n] NTT form*/);
void orc_gen_evk(const orc_ctx *c, uint64_t seed, const uint64_t *sk_ntt, int dbc, uint64_t *evk);
void orc_encrypt(const orc_ctx *c, const uint64_t *pk, const uint64_t *plain /*[n]*/, uint64_t seed, uint64_t *ct /*[2][k][n]*/);
void orc_decrypt(const orc_ctx *c, const uint64_t *sk_ntt, const uint64_t *ct, int size, uint64_t *plain /*[n]*/);
int  orc_noise_budget(const orc_ctx *c, const uint64_t *sk_ntt, const uint64_t *ct, int size);        /* decryptor.cpp:295-403 */

/* ---- CrCNN layers in the reference's own operation order (CPU baseline + parity oracle) ----
 * tensors: x[zd][xd][yd] of ct(2) ; weights w_ntt[nf][zd][xf][yf][k][n] (already transform_to_ntt'ed) ;
 * bias/mean/invstd/div: plaintext coefficient arrays [..][n].  `threads` splits like the reference does. */
void orc_conv_forward(const orc_ctx *c, const uint64_t *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
                      const uint64_t *w_ntt, const uint64_t *bias_plain, uint64_t *y, int threads,
                      int f_begin, int f_end);     /* convolutionalLayer.cpp:56-93,159-197; [f_begin,f_end) = filter slice */
void orc_fc_forward(const orc_ctx *c, const uint64_t *x, int in_dim, int out_dim, const uint64_t *w_ntt,
                    const uint64_t *bias_plain, uint64_t *y, int threads, int r_begin, int r_end); /* fullyConnectedLayer.cpp:113-168 */
void orc_pool_forward(const orc_ctx *c, const uint64_t *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf,
                      const uint64_t *div_plain /* NULL = sum pool */, uint64_t *y, int threads);   /* poolingLayer.cpp:22-44, avgPoolingLayer.cpp:16-45 */
void orc_bn_forward(const orc_ctx *c, uint64_t *x, int zd, int xd, int yd, const uint64_t *mean_plain,
                    const uint64_t *invstd_plain, int threads);                                     /* batchNormLayer.cpp:29-40 */
void orc_square_forward(const orc_ctx *c, const uint64_t *x, size_t count, const uint64_t *evk, int dbc,
                        uint64_t *y, int threads);                                                  /* squareLayer.cpp:22-74 */
/* same conv/fc result computed with NTT-domain accumulation (one INTT per output): used only to speed up big
 * parity cases in tests; bit-identical to the functions above by linearity (checked in tests). */
void orc_conv_forward_fast(const orc_ctx *c, const uint64_t *x, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
                      const uint64_t *w_ntt, const uint64_t *bias_plain, uint64_t *y, int threads);

#ifdef __cplusplus
}
#endif
#endif
