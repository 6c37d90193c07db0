// kernels.h -- launchers implemented in kernels.hip / kernels_square.hip (internal; the public surface is crcnn_hip.h)
#pragma once
#include "ctx.h"

// grid of a launch whose workgroups are mapped by xcd_group (ntt_device.h): `groups` groups of G workgroups, each group on one XCD
static inline unsigned xcd_grid(size_t groups, unsigned G) { return (unsigned)((groups + 7) / 8 * 8 * G); }

int k_ntt_ct(crc_ctx *c, bool inv, const u64 *src, u64 *dst, size_t count, int size, bool bsk, hipStream_t st,
             const u64 *addend, int add_sign, size_t add_group, int add_mod = 0, int pack_out = 0);
int k_ntt_ct_fwd_mul(crc_ctx *c, u64 *ct, size_t count, const u64 *w, size_t group, hipStream_t st);
int k_ntt_ct_fwd_fma(crc_ctx *c, u64 *ct, size_t count, const u64 *u, const u64 *key, hipStream_t st);
int k_ntt_ct_addct(crc_ctx *c, const u64 *src, u64 *dst, size_t count, const u64 *addct, int add_size, hipStream_t st);
int k_ntt_ct_head_add(crc_ctx *c, const u64 *src, int src_size, u64 *dst, size_t count, const u64 *addrows, hipStream_t st);
int k_spread_ntt(crc_ctx *c, const u64 *src, size_t items, u64 *dst, hipStream_t st);
int k_digit_ntt(crc_ctx *c, const u64 *src, int src_size, int src_poly, size_t count, int D, const unsigned char *dig_i, const unsigned char *dig_shift, int dbc,
                u64 *dst, hipStream_t st, int pack_out = 0);
int k_square_intt(crc_ctx *c, const u64 *src, u64 *dst, size_t count, bool bsk, hipStream_t st, const u64 *opt_mul = nullptr, bool *applied = nullptr);
int k_ntt_ct_inv_scaled(crc_ctx *c, const u64 *src, u64 *dst, size_t count, int size, const u64 *mul, const u64 *mul_s, hipStream_t st);
int k_plain_ntt(crc_ctx *c, const u64 *d_plain, size_t count, int mode, bool do_ntt, u64 *d_out, hipStream_t st);
int k_plain_expand(crc_ctx *c, const u64 *d_compact, size_t count, u64 *d_plain, hipStream_t st);
int k_rowwise(crc_ctx *c, u64 *acc, const u64 *b, size_t count, int size, int op, int sign, size_t group, size_t gmod, hipStream_t st);
int k_pool(crc_ctx *c, const u64 *x, u64 *y, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, const u64 *mul, hipStream_t st, int pack_out = 0);
int k_bn_ntt(crc_ctx *c, u64 *x, int B, int zd, int hw, const u64 *mean, const u64 *invstd, hipStream_t st);
int k_mac(crc_ctx *c, const u64 *x, const u64 *w, u64 *y, const int *d_xoff, const int *d_toff, int B, int P, int F, int T, int in_cts,
          const u64 *bias_ntt, hipStream_t st, int xp = 0, int wp = 0, int yp = 0);
int k_pack28(crc_ctx *c, u64 *rows, size_t nrows, bool unpack, hipStream_t st);
int k_conv_offsets(crc_ctx *c, int *xoff, int *toff, unsigned *toffw, int P, int T, int in_cts, int xd, int yd, int xs, int ys, int xf, int yf, int yo, hipStream_t st);
size_t k_square_work_words(const crc_ctx *c, size_t cnt);
size_t k_relin_work_words(const crc_ctx *c, size_t cnt, int dbc);
size_t k_relin_keys_words(const crc_ctx *c, int dbc);
// kernels_relin64.hip: key switching over the two fp64 primes
bool   k_relin64_supported(const crc_ctx *c, int dbc);
size_t k_relin64_keys_words(const crc_ctx *c, int dbc);
size_t k_relin64_work_words(const crc_ctx *c, size_t cnt, int dbc);
int k_relin64_prepare_keys(crc_ctx *c, const u64 *evk, int dbc, u64 *kp, u64 *scratch, hipStream_t st);
// a window of the sum pooling that follows a Square layer: the key switch is linear in the digit polynomials, so the digits of a window's c2's are summed before
// they are transformed and ONE key switch serves the pooled ciphertext (kernels_relin64.hip)
struct PoolGeom { int xd, yd, xs, ys, xf, yf, xo, yo; };
bool k_relin64_pool_supported(const crc_ctx *c, int dbc, int window);
int k_relinearize64(crc_ctx *c, const u64 *src, int src_size, int src_poly, const u64 *x3, int add_size, size_t cnt, int dbc, u64 *y, u64 *work, const u64 *kp,
                    hipStream_t st, bool out_ntt, const PoolGeom *pool = nullptr, const u64 *mul = nullptr);
int k_square(crc_ctx *c, const u64 *x, size_t cnt, u64 *y3, u64 *work, hipStream_t st, bool in_ntt = false, bool premul_c2 = false);
// kernels_square64.hip: the square's auxiliary base over the engine's fp64 primes
bool k_square64_supported(const crc_ctx *c);
int k_square64(crc_ctx *c, const u64 *x, size_t cnt, u64 *y3, u64 *work, hipStream_t st, bool in_ntt, bool premul_c2);
int k_relinearize(crc_ctx *c, const u64 *x3, size_t cnt, const u64 *evk, int dbc, u64 *y, u64 *work, u64 *kp, hipStream_t st, bool out_ntt = false,
                  bool c2_premul = false, bool keys_ready = false);
int k_mac2(crc_ctx *c, const u64 *x, const u64 *w, u64 *y, const int *d_xoff, const int *d_toff, int B, int P, int F, int T, int in_cts,
           const u64 *bias_ntt, int gxd, int gyd, int gxf, int gyf, const unsigned *d_toffw, hipStream_t st, int xp = 0, int wp = 0, int yp = 0);
int k_fold_pool(crc_ctx *c, const u64 *w, const u64 *bias, const u64 *div, u64 *wout, u64 *bout, int nf, int zd, int xf, int yf, int cxs, int cys,
                int pxf, int pyf, hipStream_t st);
size_t k_encrypt_work_words(const crc_ctx *c, size_t cnt);
struct ChaChaKey;
int k_encrypt(crc_ctx *c, const u64 *pk, const u64 *plain, size_t cnt, const ChaChaKey &key, u64 stream_base, u64 *ct, u64 *work, hipStream_t st, bool out_ntt = false,
              bool plain_compact = false);
void k_encrypt_cdt(u64 *out19);                 // the 19 thresholds of the device encryptor's noise magnitudes (tests)
// kernels_decrypt.hip: Decryptor::decrypt and the fractional encoder on the device (the refresh of Network::forward)
size_t k_decrypt_work_words(const crc_ctx *c, size_t cnt, int size, bool in_ntt);
int k_decrypt(crc_ctx *c, const u64 *sk, const u64 *ct, size_t cnt, int size, bool in_ntt, u64 *plain, u64 *work, hipStream_t st);
int k_decrypt_recode(crc_ctx *c, const u64 *sk, const u64 *ct, size_t cnt, bool in_ntt, u64 *compact, float *vals_out, u64 *work, hipStream_t st);
int k_fra_decode(crc_ctx *c, const u64 *plain, size_t cnt, double *out, hipStream_t st);
int k_fra_encode(crc_ctx *c, const void *src, int mode, size_t cnt, u64 *plain, float *vals_out, hipStream_t st);

// kernels_mfma.hip: conv / dense multiply-accumulate as an int8 limb GEMM on the matrix cores (operand form CRC_NTTL)
bool   k_limb_supported(const crc_ctx *c, int T);
size_t k_limb_tensor_bytes(const crc_ctx *c, int B, int zd, int npos);
size_t k_limb_weights_bytes(const crc_ctx *c, int nf, int zd, int xf, int yf);
int    k_limb_flat_zdc(int zd);                        // channel bytes per position of the flat form (layers of fewer than 32 channels), 0: blocked form
int    k_limb_steps(int zd, int xf, int yf);           // 32-term reduction steps of a layer
size_t k_limb_result_words(const crc_ctx *c, int B, int nf, int P);
int k_limb_pack_tensor(crc_ctx *c, const u64 *x, signed char *xl, int B, int zd, int npos, bool packed, hipStream_t st, int Btot = 0, int b0 = 0);
int k_limb_pack_weights(crc_ctx *c, const u64 *w, signed char *wl, int nf, int zd, int xf, int yf, hipStream_t st, int f0 = 0, int ft = -1);
int k_limb_result_to_rows(crc_ctx *c, const u64 *ys, u64 *y, size_t rows, bool pack_out, hipStream_t st);
int k_limb_result_to_limb(crc_ctx *c, const u64 *ys, signed char *xl, int B, int zd, hipStream_t st);
bool k_limb_direct_dense(int P);
int k_limb_mac(crc_ctx *c, const signed char *xl, const signed char *wl, u64 *ys, signed char *xl_out, const u64 *bias_ntt, int B, int zd, int xd, int yd, int xs, int ys_, int xf, int yf, int nf,
               hipStream_t st);
// kernels_mfma1.hip: one-channel convolutions (conv1 [+ pool1]) on the matrix cores (weight form CRC_NTTL1)
bool   k_limb_conv1_shape(const crc_ctx *c, int zd, int xd, int yd, int xs, int ys_, int xf, int yf, int nf);
size_t k_limb_conv1_weights_bytes(const crc_ctx *c);
size_t k_limb_conv1_image_bytes(const crc_ctx *c, int B, int xd);
int k_limb_conv1_pack_weights(crc_ctx *c, const u64 *w, signed char *wl, int nf, int xf, int yf, hipStream_t st);
int k_limb_conv1(crc_ctx *c, const u64 *x, bool packed, signed char *xr, const signed char *wl, u64 *ys, signed char *xl_out, int Bout, int b0, const u64 *bias_ntt, int B, int xd, int yd,
                 int xs, int ys_, int xf, int yf, int nf, hipStream_t st);
