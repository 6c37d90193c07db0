// abi.hip -- extern "C" entry points of include/crcnn_hip.h that drive the kernels (layers and Evaluator ops).
#include <cstdlib>
#include "kernels.h"
#include "chacha.h"

// every device entry point makes its context's GPU current for the calling thread (a no-op in the one-process-per-GPU flow; in a one-process, many-GPU
// application the launch must not land on whatever device the thread used last)
#define CHECK_CTX(c) do { if (!(c) || (c)->device < 0) return CRC_ERR_INVALID_ARGUMENT; \
        int dev_ = -1; if (hipGetDevice(&dev_) != hipSuccess || dev_ != (c)->device) HIPCHK(hipSetDevice((c)->device)); } while (0)
#define RUN(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
static inline hipStream_t S(void *s) { return (hipStream_t)s; }
static inline bool form_ok(int f) { return f == CRC_COEFF || f == CRC_NTT; }
static inline bool nform_ok(int f) { return f == CRC_COEFF || f == CRC_NTT || f == CRC_NTTP; }
static inline bool lform_ok(int f) { return nform_ok(f) || f == CRC_NTTL; }

extern "C" int crc_plain_to_ntt(crc_ctx *c, const uint64_t *d_plain, size_t count, uint64_t *d_out, void *stream)
{
    CHECK_CTX(c); if (!d_plain || !d_out) return CRC_ERR_INVALID_ARGUMENT;
    return k_plain_ntt(c, d_plain, count, 1, true, d_out, S(stream));
}
extern "C" int crc_plain_expand(crc_ctx *c, const uint64_t *d_compact, size_t count, uint64_t *d_plain, void *stream)
{
    CHECK_CTX(c); if (!d_compact || !d_plain) return CRC_ERR_INVALID_ARGUMENT;
    return k_plain_expand(c, d_compact, count, d_plain, S(stream));
}
extern "C" int crc_plain_to_delta(crc_ctx *c, const uint64_t *d_plain, size_t count, int form, uint64_t *d_out, void *stream)
{
    CHECK_CTX(c); if (!d_plain || !d_out || !form_ok(form)) return CRC_ERR_INVALID_ARGUMENT;
    return k_plain_ntt(c, d_plain, count, 2, form == CRC_NTT, d_out, S(stream));
}

extern "C" int crc_ntt_fwd(crc_ctx *c, uint64_t *d_ct, size_t count, int size, void *stream)
{
    CHECK_CTX(c); if (!d_ct || size < 1) return CRC_ERR_INVALID_ARGUMENT;
    return k_ntt_ct(c, false, d_ct, d_ct, count, size, false, S(stream), nullptr, 0, 0);
}
extern "C" int crc_ntt_inv(crc_ctx *c, uint64_t *d_ct, size_t count, int size, void *stream)
{
    CHECK_CTX(c); if (!d_ct || size < 1) return CRC_ERR_INVALID_ARGUMENT;
    return k_ntt_ct(c, true, d_ct, d_ct, count, size, false, S(stream), nullptr, 0, 0);
}
extern "C" int crc_ntt_fwd_bsk(crc_ctx *c, uint64_t *d_rows, size_t count, void *stream)
{
    CHECK_CTX(c); if (!d_rows) return CRC_ERR_INVALID_ARGUMENT;
    return k_ntt_ct(c, false, d_rows, d_rows, count, 1, true, S(stream), nullptr, 0, 0);
}
extern "C" int crc_ntt_inv_bsk(crc_ctx *c, uint64_t *d_rows, size_t count, void *stream)
{
    CHECK_CTX(c); if (!d_rows) return CRC_ERR_INVALID_ARGUMENT;
    return k_ntt_ct(c, true, d_rows, d_rows, count, 1, true, S(stream), nullptr, 0, 0);
}

extern "C" int crc_add(crc_ctx *c, uint64_t *d_acc, const uint64_t *d_b, size_t count, int size, void *stream)
{
    CHECK_CTX(c); if (!d_acc || !d_b || size < 1) return CRC_ERR_INVALID_ARGUMENT;
    return k_rowwise(c, d_acc, d_b, count, size, 0, 1, 1, 0, S(stream));
}
extern "C" int crc_add_plain(crc_ctx *c, uint64_t *d_ct, const uint64_t *d_delta, size_t count, size_t group, int sign, void *stream)
{
    CHECK_CTX(c); if (!d_ct || !d_delta || (sign != 1 && sign != -1)) return CRC_ERR_INVALID_ARGUMENT;
    return k_rowwise(c, d_ct, d_delta, count, 2, 1, sign, group, 0, S(stream));
}
extern "C" int crc_multiply_plain_ntt(crc_ctx *c, uint64_t *d_ct, const uint64_t *d_w, size_t count, size_t group, int size, void *stream)
{
    CHECK_CTX(c); if (!d_ct || !d_w || size < 1) return CRC_ERR_INVALID_ARGUMENT;
    return k_rowwise(c, d_ct, d_w, count, size, 2, 1, group, 0, S(stream));
}
extern "C" int crc_multiply_plain(crc_ctx *c, uint64_t *d_ct, const uint64_t *d_w, size_t count, size_t group, void *stream)
{
    CHECK_CTX(c); if (!d_ct || !d_w) return CRC_ERR_INVALID_ARGUMENT;
    // (the dyadic product in the last loop of the forward transform where the ring has the wave-local kernel, else as a pass of its own)
    const int rc = k_ntt_ct_fwd_mul(c, d_ct, count, d_w, group, S(stream));
    if (rc == CRC_ERR_UNSUPPORTED) {
        RUN(k_ntt_ct(c, false, d_ct, d_ct, count, 2, false, S(stream), nullptr, 0, 0));
        RUN(k_rowwise(c, d_ct, d_w, count, 2, 2, 1, group, 0, S(stream)));
    } else if (rc) return rc;
    return k_ntt_ct(c, true, d_ct, d_ct, count, 2, false, S(stream), nullptr, 0, 0);
}

// ---- convolution / dense ------------------------------------------------------------------------------------------
static bool conv_shape_ok(int xd, int yd, int xs, int ys, int xf, int yf)
{
    if (xd < 1 || yd < 1 || xs < 1 || ys < 1 || xf < 1 || yf < 1 || xf > xd || yf > yd) return false;
    // the reference iterates i in [0, xd - max(xf,xs) + 1) step xs (Layer::computeBoundaries, layer.cpp:12-26) but sizes
    // its result (xd-xf)/xs+1: when the stride exceeds the window the two disagree and trailing outputs stay empty
    // Ciphertexts -- reject those shapes instead of inventing values.
    const int xl = xd - (xf > xs ? xf : xs) + 1, yl = yd - (yf > ys ? yf : ys) + 1;
    if (xl < 1 || yl < 1) return false;
    return (xl + xs - 1) / xs == (xd - xf) / xs + 1 && (yl + ys - 1) / ys == (yd - yf) / ys + 1;
}
static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" size_t crc_conv2d_work_bytes(const crc_ctx *c, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int in_form)
{
    (void)nf;
    if (!c || !conv_shape_ok(xd, yd, xs, ys, xf, yf)) return 0;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
    size_t b = align256(sizeof(int) * (size_t)xo * yo) + 2 * align256(sizeof(int) * ((size_t)zd * xf * yf + 8));
    if (in_form == CRC_COEFF) b += (size_t)B * zd * xd * yd * crc_ct_words(c, 2) * 8;
    return b + 256;
}

// ---- limb form (CRC_NTTL): the layer on the matrix cores (kernels_mfma.hip) ------------------------------------------------------------
extern "C" int crc_limb_supported(const crc_ctx *c, int zd, int xf, int yf)
{
    if (!c || zd < 1 || xf < 1 || yf < 1) return 0;
    return k_limb_supported(c, k_limb_steps(zd, xf, yf) * 32) ? 1 : 0;
}
extern "C" size_t crc_limb_tensor_bytes(const crc_ctx *c, int B, int zd, int xd, int yd) { return c ? k_limb_tensor_bytes(c, B, zd, xd * yd) : 0; }
extern "C" size_t crc_limb_weights_bytes(const crc_ctx *c, int nf, int zd, int xf, int yf) { return c ? k_limb_weights_bytes(c, nf, zd, xf, yf) : 0; }
extern "C" int crc_limb_pack_weights(crc_ctx *c, const uint64_t *d_w_ntt, int nf, int zd, int xf, int yf, void *d_wl, void *stream)
{
    CHECK_CTX(c); if (!d_w_ntt || !d_wl || nf < 1 || zd < 1 || xf < 1 || yf < 1) return CRC_ERR_INVALID_ARGUMENT;
    if (!crc_limb_supported(c, zd, xf, yf)) return CRC_ERR_UNSUPPORTED;
    return k_limb_pack_weights(c, d_w_ntt, (signed char *)d_wl, nf, zd, xf, yf, S(stream));
}
extern "C" int crc_limb_pack_weights_tile(crc_ctx *c, const uint64_t *d_w_tile_ntt, int nf, int f0, int ft, int zd, int xf, int yf, void *d_wl, void *stream)
{
    CHECK_CTX(c); if (!d_w_tile_ntt || !d_wl || nf < 1 || zd < 1 || xf < 1 || yf < 1 || f0 < 0 || ft < 1 || f0 + ft > nf) return CRC_ERR_INVALID_ARGUMENT;
    if (!crc_limb_supported(c, zd, xf, yf)) return CRC_ERR_UNSUPPORTED;
    return k_limb_pack_weights(c, d_w_tile_ntt, (signed char *)d_wl, nf, zd, xf, yf, S(stream), f0, ft);
}
extern "C" int crc_limb_pack_tensor(crc_ctx *c, const uint64_t *d_x, int in_form, int B, int zd, int xd, int yd, void *d_xl, void *stream)
{
    CHECK_CTX(c); if (!d_x || !d_xl || B < 0 || zd < 1 || xd < 1 || yd < 1 || (in_form != CRC_NTT && in_form != CRC_NTTP)) return CRC_ERR_INVALID_ARGUMENT;
    if (!crc_limb_supported(c, zd, 1, 1)) return CRC_ERR_UNSUPPORTED;
    return k_limb_pack_tensor(c, d_x, (signed char *)d_xl, B, zd, xd * yd, in_form == CRC_NTTP, S(stream));
}
extern "C" int crc_limb_pack_tensor_at(crc_ctx *c, const uint64_t *d_x, int in_form, int B, int zd, int xd, int yd, void *d_xl, int Btot, int b0, void *stream)
{
    CHECK_CTX(c); if (!d_x || !d_xl || B < 0 || zd < 1 || xd < 1 || yd < 1 || Btot < B || b0 < 0 || b0 + B > Btot || (in_form != CRC_NTT &&
        in_form != CRC_NTTP)) return CRC_ERR_INVALID_ARGUMENT;
    if (!crc_limb_supported(c, zd, 1, 1)) return CRC_ERR_UNSUPPORTED;
    return k_limb_pack_tensor(c, d_x, (signed char *)d_xl, B, zd, xd * yd, in_form == CRC_NTTP, S(stream), Btot, b0);
}
// one-channel convolutions (CRC_NTTL1, kernels_mfma1.hip)
extern "C" int crc_limb_conv1_supported(const crc_ctx *c, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf)
{
    if (!c || nf < 1 || !conv_shape_ok(xd, yd, xs, ys, xf, yf)) return 0;
    return k_limb_conv1_shape(c, zd, xd, yd, xs, ys, xf, yf, nf) ? 1 : 0;
}
extern "C" size_t crc_limb_conv1_weights_bytes(const crc_ctx *c) { return c ? k_limb_conv1_weights_bytes(c) : 0; }
extern "C" int crc_limb_conv1_pack_weights(crc_ctx *c, const uint64_t *d_w_ntt, int nf, int xf, int yf, void *d_wl, void *stream)
{
    CHECK_CTX(c); if (!d_w_ntt || !d_wl || nf < 1 || nf > 32 || xf < 1 || yf < 1 || xf > 8 || yf > 8) return CRC_ERR_INVALID_ARGUMENT;
    if (!k_limb_supported(c, 64)) return CRC_ERR_UNSUPPORTED;
    return k_limb_conv1_pack_weights(c, d_w_ntt, (signed char *)d_wl, nf, xf, yf, S(stream));
}
// images per internal pass: the y-expanded limb images and the slot-major result of a pass stay below ~16 GiB of work space (a dense consumer's flattened
// limb tensor is converted from the whole batch's result at once: no sub-batching there)
static int conv1_sub_batch(const crc_ctx *c, int B, int xd, int yo, int nf, int P, int out_form)
{
    if (out_form == CRC_NTTL || B <= 1) return B;
    const size_t per = k_limb_conv1_image_bytes(c, 1, xd) + (out_form == CRC_NTTLC ? 0 : 8 * k_limb_result_words(c, 1, nf, P));
    const long long ev = c->tune.conv1_pass_bytes;                   // (the tests shrink it to cover the multi-pass path at small sizes: crc_ctx_set_tuning)
    const size_t cap = ev > 0 ? (size_t)ev : (size_t)16 << 30;
    const size_t fit = cap / (per ? per : 1);
    return (int)(fit < 1 ? 1 : fit > (size_t)B ? (size_t)B : fit);
}
static int conv2d_limb1(crc_ctx *c, const uint64_t *d_x, const void *d_wl, const uint64_t *d_bias, int B, int xd, int yd, int xs, int ys, int xf, int yf,
    int nf,
                        int in_form, int out_form, uint64_t *d_y, void *d_work, hipStream_t st)
{
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1, P = xo * yo, in_cts = xd * yd;
    // a 1 x 1 result is a dense layer's input: the K-blocked form (kernels_mfma.hip), made from the slot-major result
    if (out_form == CRC_NTTLC && P == 1) out_form = CRC_NTTL;
    const int Bs = conv1_sub_batch(c, B, xd, yo, nf, P, out_form);
    const size_t ctw = crc_ct_words(c, 2);
    char *w = (char *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    signed char *Xr = (signed char *)w; w += align256(k_limb_conv1_image_bytes(c, Bs, xd));
    u64 *Ys = nullptr;
    if (out_form != CRC_NTTLC) { Ys = (u64 *)w; w += align256(8 * k_limb_result_words(c, Bs, nf, P)); }
    u64 *buf = (u64 *)w;                                           // NTT copy of a coefficient-form sub-batch
    for (int b0 = 0; b0 < B; b0 += Bs) {
        const int Bn = B - b0 < Bs ? B - b0 : Bs;
        const u64 *xn = d_x + (size_t)b0 * in_cts * ctw; bool packed = in_form == CRC_NTTP;
        if (in_form == CRC_COEFF) { RUN(k_ntt_ct(c, false, xn, buf, (size_t)Bn * in_cts, 2, false, st, nullptr, 0, 0, 0, 0)); xn = buf; packed = false; }
        RUN(k_limb_conv1(c, xn, packed, Xr, (const signed char *)d_wl, Ys, out_form == CRC_NTTLC ? (signed char *)d_y : nullptr, B, b0,
            out_form != CRC_COEFF ? d_bias : nullptr,
                         Bn, xd, yd, xs, ys, xf, yf, nf, st));
        if (out_form == CRC_NTTLC) continue;
        if (out_form == CRC_NTTL) return k_limb_result_to_limb(c, Ys, (signed char *)d_y, B, nf * P, st);     // (Bs == B)
        RUN(k_limb_result_to_rows(c, Ys, d_y + (size_t)b0 * nf * P * ctw, (size_t)Bn * nf * P * 2, out_form == CRC_NTTP, st));
    }
    if (out_form == CRC_COEFF) RUN(k_ntt_ct(c, true, d_y, d_y, (size_t)B * nf * P, 2, false, st, d_bias, 1, (size_t)P, nf));
    return CRC_OK;
}
extern "C" size_t crc_conv2d_forms_work_bytes(const crc_ctx *c, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int in_form,
    int w_form, int out_form)
{
    if (w_form == CRC_NTTL1) {
        if (!c || !conv_shape_ok(xd, yd, xs, ys, xf, yf)) return 0;
        const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
        if (out_form == CRC_NTTLC && xo * yo == 1) out_form = CRC_NTTL;
        const int Bs = conv1_sub_batch(c, B, xd, yo, nf, xo * yo, out_form);
        size_t b = align256(k_limb_conv1_image_bytes(c, Bs, xd));
        if (out_form != CRC_NTTLC) b += align256(8 * k_limb_result_words(c, Bs, nf, xo * yo));
        if (in_form == CRC_COEFF) b += align256((size_t)Bs * xd * yd * crc_ct_words(c, 2) * 8);
        return b + 256;
    }
    if (w_form != CRC_NTTL) return crc_conv2d_work_bytes(c, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form);
    if (!c || !conv_shape_ok(xd, yd, xs, ys, xf, yf)) return 0;
    (void)out_form;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
    size_t b = align256(8 * k_limb_result_words(c, B, nf, xo * yo));                                   // Ys
    if (in_form != CRC_NTTL) b += align256(k_limb_tensor_bytes(c, B, zd, xd * yd));                      // Xl
    if (in_form == CRC_COEFF) b += align256((size_t)B * zd * xd * yd * crc_ct_words(c, 2) * 8);          // NTT copy of the input
    return b + 256;
}
static int conv2d_limb(crc_ctx *c, const uint64_t *d_x, const void *d_wl, const uint64_t *d_bias, int B, int zd, int xd, int yd, int xs, int ys, int xf,
    int yf, int nf,
                       int in_form, int out_form, uint64_t *d_y, void *d_work, hipStream_t st)
{
    if (!crc_limb_supported(c, zd, xf, yf)) return CRC_ERR_UNSUPPORTED;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1, P = xo * yo, in_cts = zd * xd * yd;
    char *w = (char *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    u64 *Ys = (u64 *)w; w += align256(8 * k_limb_result_words(c, B, nf, P));
    const signed char *xl = (const signed char *)d_x;
    if (in_form != CRC_NTTL) {
        signed char *Xl = (signed char *)w; w += align256(k_limb_tensor_bytes(c, B, zd, xd * yd));
        const u64 *xn = d_x; bool packed = in_form == CRC_NTTP;
        if (in_form == CRC_COEFF) { u64 *buf = (u64 *)w; RUN(k_ntt_ct(c, false, d_x, buf, (size_t)B * in_cts, 2, false, st, nullptr, 0, 0, 0, 0)); xn = buf;
            packed = false; }
        RUN(k_limb_pack_tensor(c, xn, Xl, B, zd, xd * yd, packed, st));
        xl = Xl;
    }
    // bias joins in the NTT domain unless the result goes back to coefficient form (then add_plain(bias) rides on the inverse transform's store)
    if (out_form == CRC_NTTL && k_limb_direct_dense(P))         // hand-over to a dense layer (channels = (f, px, py) flattened), written by the kernel itself
        return k_limb_mac(c, xl, (const signed char *)d_wl, Ys, (signed char *)d_y, d_bias, B, zd, xd, yd, xs, ys, xf, yf, nf, st);
    RUN(k_limb_mac(c, xl, (const signed char *)d_wl, Ys, nullptr, out_form != CRC_COEFF ? d_bias : nullptr, B, zd, xd, yd, xs, ys, xf, yf, nf, st));
    if (out_form == CRC_NTTL) return k_limb_result_to_limb(c, Ys, (signed char *)d_y, B, nf * P, st);     // ... or re-limbed from the slot-major result
    RUN(k_limb_result_to_rows(c, Ys, d_y, (size_t)B * nf * P * 2, out_form == CRC_NTTP, st));
    if (out_form == CRC_COEFF) RUN(k_ntt_ct(c, true, d_y, d_y, (size_t)B * nf * P, 2, false, st, d_bias, 1, (size_t)P, nf));
    return CRC_OK;
}

extern "C" int crc_conv2d_forms(crc_ctx *c, const uint64_t *d_x, const uint64_t *d_w, int w_form, const uint64_t *d_bias, int B, int zd, int xd, int yd,
                                int xs, int ys, int xf, int yf, int nf, int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream)
{
    CHECK_CTX(c);
    if (w_form == CRC_NTTL1) {
        if (!d_x || !d_w || !d_y || !d_work || B < 0 || nf < 1 || !nform_ok(in_form) || !(lform_ok(out_form) || out_form == CRC_NTTLC) || !conv_shape_ok(xd,
            yd, xs, ys, xf, yf))
            return CRC_ERR_INVALID_ARGUMENT;
        if (!k_limb_conv1_shape(c, zd, xd, yd, xs, ys, xf, yf, nf)) return CRC_ERR_UNSUPPORTED;
        if (B == 0) return CRC_OK;
        return conv2d_limb1(c, d_x, d_w, d_bias, B, xd, yd, xs, ys, xf, yf, nf, in_form, out_form, d_y, d_work, S(stream));
    }
    if (w_form == CRC_NTTL) {
        if (!d_x || !d_w || !d_y || !d_work || B < 0 || zd < 1 || nf < 1 || !lform_ok(in_form) || !lform_ok(out_form) || !conv_shape_ok(xd, yd, xs, ys, xf,
            yf)) return CRC_ERR_INVALID_ARGUMENT;
        if (B == 0) return CRC_OK;
        return conv2d_limb(c, d_x, d_w, d_bias, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, out_form, d_y, d_work, S(stream));
    }
    if (!d_x || !d_w || !d_y || !d_work || B < 0 || zd < 1 || nf < 1 || !nform_ok(in_form) || !nform_ok(out_form) || (w_form != CRC_NTT && w_form != CRC_NTTP))
        return CRC_ERR_INVALID_ARGUMENT;
    if (!conv_shape_ok(xd, yd, xs, ys, xf, yf)) return CRC_ERR_INVALID_ARGUMENT;
    if (B == 0) return CRC_OK;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1, P = xo * yo, T = zd * xf * yf, in_cts = zd * xd * yd;
    hipStream_t st = S(stream);
    char *w = (char *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    int *xoff = (int *)w; w += align256(sizeof(int) * (size_t)P);
    int *toff = (int *)w; w += align256(sizeof(int) * ((size_t)T + 8));
    unsigned *toffw = (unsigned *)w; w += align256(sizeof(int) * ((size_t)T + 8));
    RUN(k_conv_offsets(c, xoff, toff, toffw, P, T, in_cts, xd, yd, xs, ys, xf, yf, yo, st));
    const u64 *xn = d_x;
    int xp = in_form == CRC_NTTP;
    if (in_form == CRC_COEFF) {                   // transform_input_to_ntt, convolutionalLayer.cpp:95-148 (out of place: x is const)
        u64 *buf = (u64 *)w;
        int maxbits = 0; for (int i = 0; i < c->k; i++) if ((int)c->tabs[i].m.bits > maxbits) maxbits = c->tabs[i].m.bits;
        xp = maxbits <= 55;                       // the private copy goes straight into the MAC kernels' operand form
        RUN(k_ntt_ct(c, false, d_x, buf, (size_t)B * in_cts, 2, false, st, nullptr, 0, 0, 0, xp));
        xn = buf;
    }
    // sum of products in the NTT domain; bias joins here when the output stays NTT-resident
    RUN(k_mac2(c, xn, d_w, d_y, xoff, toff, B, P, nf, T, in_cts, out_form != CRC_COEFF ? d_bias : nullptr, xd, yd, xf, yf, toffw, st, xp, w_form == CRC_NTTP,
        out_form == CRC_NTTP));
    if (out_form == CRC_COEFF)                    // one inverse NTT per output ciphertext, add_plain(bias) fused into its store
        RUN(k_ntt_ct(c, true, d_y, d_y, (size_t)B * nf * P, 2, false, st, d_bias, 1, (size_t)P, nf));
    return CRC_OK;
}
extern "C" int crc_conv2d(crc_ctx *c, const uint64_t *d_x, const uint64_t *d_w, const uint64_t *d_bias, int B, int zd, int xd, int yd,
                          int xs, int ys, int xf, int yf, int nf, int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream)
{
    if (!form_ok(in_form) || !form_ok(out_form)) return CRC_ERR_INVALID_ARGUMENT;
    return crc_conv2d_forms(c, d_x, d_w, CRC_NTT, d_bias, B, zd, xd, yd, xs, ys, xf, yf, nf, in_form, out_form, d_y, d_work, stream);
}
extern "C" int crc_dense_forms(crc_ctx *c, const uint64_t *d_x, const uint64_t *d_w, int w_form, const uint64_t *d_bias, int B, int in_dim, int out_dim,
                               int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream)
{
    return crc_conv2d_forms(c, d_x, d_w, w_form, d_bias, B, in_dim, 1, 1, 1, 1, 1, 1, out_dim, in_form, out_form, d_y, d_work, stream);
}
extern "C" int crc_pack28(crc_ctx *c, uint64_t *d_rows, size_t rows, int unpack, void *stream)
{
    CHECK_CTX(c); if (!d_rows) return CRC_ERR_INVALID_ARGUMENT;
    return k_pack28(c, d_rows, rows, unpack != 0, S(stream));
}

extern "C" size_t crc_dense_work_bytes(const crc_ctx *c, int B, int in_dim, int out_dim, int in_form)
{
    return crc_conv2d_work_bytes(c, B, in_dim, 1, 1, 1, 1, 1, 1, out_dim, in_form);
}
extern "C" int crc_dense(crc_ctx *c, const uint64_t *d_x, const uint64_t *d_w, const uint64_t *d_bias, int B, int in_dim, int out_dim,
                         int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream)
{
    // a dense layer is the 1x1 convolution of an in_dim-channel 1x1 image (reshapeInput, fullyConnectedLayer.cpp:38-56,
    // flattens z,x,y row-major, which is the tensor's memory order)
    return crc_conv2d(c, d_x, d_w, d_bias, B, in_dim, 1, 1, 1, 1, 1, 1, out_dim, in_form, out_form, d_y, d_work, stream);
}

extern "C" int crc_conv2d_fold_pool(crc_ctx *c, const uint64_t *d_w, const uint64_t *d_bias_ntt, const uint64_t *d_div_ntt, int nf, int zd, int xf, int yf,
                                    int cxs, int cys, int pxf, int pyf, uint64_t *d_w_out, uint64_t *d_bias_out, void *stream)
{
    CHECK_CTX(c);
    if (!d_w || !d_bias_ntt || !d_w_out || !d_bias_out || nf < 1 || zd < 1 || xf < 1 || yf < 1 || cxs < 1 || cys < 1 || pxf < 1 ||
        pyf < 1) return CRC_ERR_INVALID_ARGUMENT;
    return k_fold_pool(c, d_w, d_bias_ntt, d_div_ntt, d_w_out, d_bias_out, nf, zd, xf, yf, cxs, cys, pxf, pyf, S(stream));
}

// ---- pooling / batch-norm -----------------------------------------------------------------------------------------
extern "C" int crc_pool(crc_ctx *c, const uint64_t *d_x, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf,
                        const uint64_t *d_div, int form, uint64_t *d_y, void *stream)
{
    // form = CRC_NTTP: NTT-form input, result written in the MAC kernels' operand form (the layer behind is a conv / dense layer)
    CHECK_CTX(c);
    if (!d_x || !d_y || B < 0 || zd < 1 || !nform_ok(form) || !conv_shape_ok(xd, yd, xs, ys, xf, yf)) return CRC_ERR_INVALID_ARGUMENT;
    if (B == 0) return CRC_OK;
    hipStream_t st = S(stream);
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
    if (form == CRC_NTTP) return k_pool(c, d_x, d_y, B, zd, xd, yd, xs, ys, xf, yf, d_div, st, 1);
    if (form == CRC_NTT || !d_div) return k_pool(c, d_x, d_y, B, zd, xd, yd, xs, ys, xf, yf, form == CRC_NTT ? d_div : nullptr, st);
    // coefficient form average pooling: add_many then multiply_plain(div_factor) (avgPoolingLayer.cpp:37-38)
    RUN(k_pool(c, d_x, d_y, B, zd, xd, yd, xs, ys, xf, yf, nullptr, st));
    const size_t cnt = (size_t)B * zd * xo * yo;
    RUN(k_ntt_ct(c, false, d_y, d_y, cnt, 2, false, st, nullptr, 0, 0));
    RUN(k_rowwise(c, d_y, d_div, cnt, 2, 2, 1, cnt, 0, st));
    return k_ntt_ct(c, true, d_y, d_y, cnt, 2, false, st, nullptr, 0, 0);
}

extern "C" int crc_batchnorm(crc_ctx *c, uint64_t *d_x, int B, int zd, int xd, int yd, const uint64_t *d_mean, const uint64_t *d_invstd,
                             int form, void *stream)
{
    CHECK_CTX(c);
    if (!d_x || !d_mean || !d_invstd || B < 0 || zd < 1 || xd < 1 || yd < 1 || !form_ok(form)) return CRC_ERR_INVALID_ARGUMENT;
    if (B == 0) return CRC_OK;
    hipStream_t st = S(stream);
    const size_t hw = (size_t)xd * yd, cnt = (size_t)B * zd * hw;
    if (form == CRC_NTT) return k_bn_ntt(c, d_x, B, zd, (int)hw, d_mean, d_invstd, st);
    // sub_plain(mean[z]) then multiply_plain(var[z])  (batchNormLayer.cpp:36-37)
    RUN(k_rowwise(c, d_x, d_mean, cnt, 2, 1, -1, hw, (size_t)zd, st));
    RUN(k_ntt_ct(c, false, d_x, d_x, cnt, 2, false, st, nullptr, 0, 0));
    RUN(k_rowwise(c, d_x, d_invstd, cnt, 2, 2, 1, hw, (size_t)zd, st));
    return k_ntt_ct(c, true, d_x, d_x, cnt, 2, false, st, nullptr, 0, 0);
}

// ---- kernel selection: ONE statement of the policy for every host (netrun.py, crcnn_amd/host) ------------------------------------------------------
// which multiply-accumulate kernel a conv / dense layer (dense: xd = yd = xf = yf = xs = ys = 1, zd = in_dim, nf = out_dim) runs on when launched on B images
extern "C" int crc_plan_mac(const crc_ctx *c, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int B, int matrix_cores, int *w_form)
{
    if (!c || !w_form || zd < 1 || nf < 1 || !conv_shape_ok(xd, yd, xs, ys, xf, yf)) return CRC_ERR_INVALID_ARGUMENT;
    bool packable = true;
    for (int i = 0; i < c->k; i++) if (c->tabs[i].m.bits > 55) packable = false;
    *w_form = packable ? CRC_NTTP : CRC_NTT;                      // mac3_kernel / mac2_kernel on 28-bit limb pairs (canonical residues above 55 bits)
    if (!matrix_cores || !packable) return CRC_OK;
    // one-channel convolutions (conv1, alone or with its pooling layer folded in) have their own matrix-core kernel
    if (zd == 1) { if (k_limb_conv1_shape(c, zd, xd, yd, xs, ys, xf, yf, nf)) *w_form = CRC_NTTL1; return CRC_OK; }
    // the limb GEMM pays from 8 reduction steps of 32 channels on (below that its fixed costs per output tile and the channel padding eat the gain), and only
    // with at least half a 64-row tile of rows = (image, pixel, poly) per launch: with fewer, most of every MFMA is padding and every slot's weights are
    // streamed for a handful of rows (PlainModelWoPad at 6 images per launch: fc4 0.23 ms per image on mac3_kernel against 1.59)
    const int min_steps = c->tune.mfma_min_steps > 0 ? c->tune.mfma_min_steps : 8;
    const long long P = (long long)((xd - xf) / xs + 1) * ((yd - yf) / ys + 1);
    // ... a full tile of rows for layers of fewer than 24 filters: the limb form pads the filters to 64, so a 10-filter layer -- CrCNN's fc4 -- spends 6x its
    // canonical bytes and 5/6 of its MFMAs on zeros (PlainModelWoPad's fc4 at 24 images: 14 GiB of weights, 0.16 against 0.11 ms per image on the vector-ALU
    // kernel; with 64 rows and more -- PlainModelTiny at 128 images, ApproxPlainModel at 32 -- it still wins, not least because the layer in front hands its
    // tensor over in limb form)
    // -- and only on the smaller rings (n k <= 32768), where those 6x are a few GB (28 GiB at n = 16384 with eight primes)
    const long long rows = (long long)B * 2 * P, min_rows = nf >= 24 ? 32 : 64;
    const bool few_filters_ok = nf >= 24 || (long long)c->n * c->k <= 32768;
    if (zd >= 16 && few_filters_ok && (long long)((zd + 31) / 32) * xf * yf >= min_steps && crc_limb_supported(c, zd, xf, yf) && (B <= 0 ||
        rows >= min_rows)) *w_form = CRC_NTTL;
    return CRC_OK;
}
// should a (sum / average) pooling layer be folded into the convolution in front of it (crc_conv2d_fold_pool: exact)?  Cost in units of one multiply-accumulate
// term per output ciphertext: the MAC kernels pay ~24 terms of prologue / epilogue per output and take filters in multiples of 8; a pooling pass moves (window
// + 1) ciphertexts per output at HBM rate, ~10 term-times each.  Folding wins whenever it removes MACs (decimating pools) and narrowly for CrCNN's stride-1
// pools.
extern "C" int crc_plan_fold_pool(const crc_ctx *c, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int pxs, int pys, int pxf, int pyf,
    int *fold)
{
    if (!c || !fold || zd < 1 || nf < 1 || !conv_shape_ok(xd, yd, xs, ys, xf, yf) || pxs < 1 || pys < 1 || pxf < 1 || pyf < 1) return CRC_ERR_INVALID_ARGUMENT;
    *fold = 0;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
    const int xf2 = (pxf - 1) * xs + xf, yf2 = (pyf - 1) * ys + yf, xs2 = xs * pxs, ys2 = ys * pys;
    if (xf2 > xd || yf2 > yd || pxf > xo || pyf > yo) return CRC_OK;
    const int xo2 = (xd - xf2) / xs2 + 1, yo2 = (yd - yf2) / ys2 + 1;
    if (xo2 != (xo - pxf) / pxs + 1 || yo2 != (yo - pyf) / pys + 1) return CRC_OK;           // the folded convolution must produce exactly the pooled tensor
    const long long fpad = (nf + 7) / 8 * 8, T1 = (long long)zd * xf * yf, T2 = (long long)zd * xf2 * yf2;
    const long long cost_sep = fpad * xo * yo * (T1 + 24) + (long long)nf * xo2 * yo2 * 10 * (pxf * pyf + 1);
    const long long cost_fused = fpad * xo2 * yo2 * (T2 + 24);
    *fold = cost_fused < cost_sep ? 1 : 0;
    return CRC_OK;
}

// ---- square + relinearize -----------------------------------------------------------------------------------------
// ciphertexts per internal pass of square + relinearise: bounds the scratch footprint at ~10 GB for every ring size -- 1024 up to n k = 32768, 512 up to 65536,
// 256 at n = 16384 with all eight primes.  (Round 5: 1024 instead of 512 on the small rings -- the eight kernels of a pass each end in a partly filled last
// wave of workgroups; at (8192, 3) twice the pass is 2 % faster, at (16384, 4) it changes nothing: profiles/r05_square_chunk_sweep.txt)
static size_t square_chunk(const crc_ctx *c)
{
    if (c->tune.sq_chunk > 0) return (size_t)c->tune.sq_chunk;
    const size_t nk = (size_t)c->n * c->k;
    return nk <= 32768 ? 1024 : nk <= 65536 ? 512 : nk <= 131072 ? 256 : 128;
}

extern "C" size_t crc_square_relin_work_bytes(const crc_ctx *c, size_t count, int dbc)
{
    if (!c) return 0;
    const size_t kSquareChunk = square_chunk(c); const size_t ch = count < kSquareChunk ? count : kSquareChunk;
    const size_t sq = k_square_work_words(c, ch), rl = k_relin_work_words(c, ch, dbc);
    // [packed keys][size-3 intermediates of one pass][scratch of the square, then of the relinearisation]
    return 8 * (k_relin_keys_words(c, dbc) + (sq > rl ? sq : rl) + ch * crc_ct_words(c, 3)) + 256;
}
extern "C" size_t crc_encrypt_dev_work_bytes(const crc_ctx *c, size_t count) { return c ? 8 * k_encrypt_work_words(c, count) + 256 : 0; }
extern "C" int crc_encrypt_dev_key(crc_ctx *c, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, const uint8_t *key, uint64_t stream_base,
                                   uint64_t *d_ct, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_pk || !d_plain || !d_ct || !d_work || !key) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    return k_encrypt(c, d_pk, d_plain, count, chacha_load_key(key), stream_base, d_ct, w, S(stream));
}
extern "C" int crc_encrypt_dev(crc_ctx *c, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, uint64_t seed, uint64_t *d_ct, void *d_work,
    void *stream)
{
    CHECK_CTX(c); if (!d_pk || !d_plain || !d_ct || !d_work) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    return k_encrypt(c, d_pk, d_plain, count, chacha_seed_key(seed), 0, d_ct, w, S(stream));
}
extern "C" int crc_encrypt_dev_key_forms(crc_ctx *c, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, const uint8_t *key, uint64_t stream_base,
                                         int out_form, uint64_t *d_ct, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_pk || !d_plain || !d_ct || !d_work || !key || (out_form != CRC_COEFF && out_form != CRC_NTT)) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    return k_encrypt(c, d_pk, d_plain, count, chacha_load_key(key), stream_base, d_ct, w, S(stream), out_form == CRC_NTT);
}
extern "C" int crc_encrypt_dev_forms(crc_ctx *c, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, uint64_t seed, int out_form, uint64_t *d_ct,
                                     void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_pk || !d_plain || !d_ct || !d_work || (out_form != CRC_COEFF && out_form != CRC_NTT)) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    return k_encrypt(c, d_pk, d_plain, count, chacha_seed_key(seed), 0, d_ct, w, S(stream), out_form == CRC_NTT);
}
extern "C" void crc_encrypt_dev_noise_thresholds(uint64_t *h_out19) { if (h_out19) k_encrypt_cdt(h_out19); }

// ---- Decryptor::decrypt, FractionalEncoder and the refresh of Network::forward on the device (kernels_decrypt.hip) ----
static bool ct_form_ok(int f) { return f == CRC_COEFF || f == CRC_NTT; }
extern "C" size_t crc_decrypt_dev_work_bytes(const crc_ctx *c, size_t count, int size, int in_form)
{
    return c && ct_form_ok(in_form) ? 8 * k_decrypt_work_words(c, count, size, in_form == CRC_NTT) + 256 : 0;
}
extern "C" int crc_decrypt_dev(crc_ctx *c, const uint64_t *d_sk, const uint64_t *d_ct, size_t count, int size, int in_form, uint64_t *d_plain, void *d_work,
                               void *stream)
{
    CHECK_CTX(c); if (!d_sk || !d_ct || !d_plain || !d_work || !ct_form_ok(in_form)) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    return k_decrypt(c, d_sk, d_ct, count, size, in_form == CRC_NTT, d_plain, w, S(stream));
}
extern "C" int crc_decode_dev(crc_ctx *c, const uint64_t *d_plain, size_t count, double *d_out, void *stream)
{
    CHECK_CTX(c); if (!d_plain || !d_out) return CRC_ERR_INVALID_ARGUMENT;
    return k_fra_decode(c, d_plain, count, d_out, S(stream));
}
extern "C" int crc_encode_dev_f32(crc_ctx *c, const float *d_values, size_t count, uint64_t *d_plain, void *stream)
{
    CHECK_CTX(c); if (!d_values || !d_plain) return CRC_ERR_INVALID_ARGUMENT;
    return k_fra_encode(c, d_values, 0, count, d_plain, nullptr, S(stream));
}
extern "C" int crc_encode_dev_f64(crc_ctx *c, const double *d_values, size_t count, uint64_t *d_plain, void *stream)
{
    CHECK_CTX(c); if (!d_values || !d_plain) return CRC_ERR_INVALID_ARGUMENT;
    return k_fra_encode(c, d_values, 1, count, d_plain, nullptr, S(stream));
}
// work of a refresh: [compact plaintexts [count][96]][dense plaintexts [count][n]: coefficient-form results only][the decryptor's rows, then the encryptor's samples]
extern "C" size_t crc_refresh_dev_work_bytes(const crc_ctx *c, size_t count, int in_form)
{
    if (!c || !ct_form_ok(in_form)) return 0;
    const size_t dec = k_decrypt_work_words(c, count, 2, in_form == CRC_NTT), enc = k_encrypt_work_words(c, count);
    return 8 * (count * ((size_t)c->n + CRC_PLAIN_COMPACT_WORDS) + (dec > enc ? dec : enc)) + 256;
}
static int refresh_impl(crc_ctx *c, const u64 *d_sk, const u64 *d_pk, const u64 *d_in, size_t count, int in_form, const ChaChaKey &key, u64 stream_base,
                        int out_form, u64 *d_out, float *d_vals, void *d_work, hipStream_t st)
{
    u64 *compact = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255), *dense = compact + count * (size_t)CRC_PLAIN_COMPACT_WORDS;
    u64 *w = dense + count * (size_t)c->n;
    // decrypt -> decode -> float -> encode leaves the 96-word compact plaintexts (only the 96 coefficients the decoder reads are ever scaled) ...
    RUN(k_decrypt_recode(c, d_sk, d_in, count, in_form == CRC_NTT, compact, d_vals, w, st));
    // ... which the NTT-form encryptor reads as they are; the coefficient-form one adds Delta m from dense rows
    if (out_form == CRC_NTT) return k_encrypt(c, d_pk, compact, count, key, stream_base, d_out, w, st, true, true);
    RUN(k_plain_expand(c, compact, count, dense, st));
    return k_encrypt(c, d_pk, dense, count, key, stream_base, d_out, w, st, false);
}
extern "C" int crc_refresh_dev(crc_ctx *c, const uint64_t *d_sk, const uint64_t *d_pk, const uint64_t *d_ct_in, size_t count, int in_form, uint64_t seed,
                               int out_form, uint64_t *d_ct_out, float *d_values_out, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_sk || !d_pk || !d_ct_in || !d_ct_out || !d_work || !ct_form_ok(in_form) || !ct_form_ok(out_form)) return CRC_ERR_INVALID_ARGUMENT;
    return refresh_impl(c, d_sk, d_pk, d_ct_in, count, in_form, chacha_seed_key(seed), 0, out_form, d_ct_out, d_values_out, d_work, S(stream));
}
extern "C" int crc_refresh_dev_key(crc_ctx *c, const uint64_t *d_sk, const uint64_t *d_pk, const uint64_t *d_ct_in, size_t count, int in_form,
                                   const uint8_t *key, uint64_t stream_base, int out_form, uint64_t *d_ct_out, float *d_values_out, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_sk || !d_pk || !d_ct_in || !d_ct_out || !d_work || !key || !ct_form_ok(in_form) || !ct_form_ok(out_form))
        return CRC_ERR_INVALID_ARGUMENT;
    return refresh_impl(c, d_sk, d_pk, d_ct_in, count, in_form, chacha_load_key(key), stream_base, out_form, d_ct_out, d_values_out, d_work, S(stream));
}
extern "C" int crc_square(crc_ctx *c, const uint64_t *d_x, size_t count, uint64_t *d_y3, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_x || !d_y3 || !d_work) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    const size_t kSquareChunk = square_chunk(c);
    for (size_t o = 0; o < count; o += kSquareChunk) {
        const size_t ch = count - o < kSquareChunk ? count - o : kSquareChunk;
        RUN(k_square(c, d_x + o * crc_ct_words(c, 2), ch, d_y3 + o * crc_ct_words(c, 3), w, S(stream)));
    }
    return CRC_OK;
}
extern "C" int crc_relinearize(crc_ctx *c, const uint64_t *d_x3, size_t count, const uint64_t *d_evk, int dbc, uint64_t *d_y, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_x3 || !d_y || !d_evk || !d_work) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    const size_t kSquareChunk = square_chunk(c);
    for (size_t o = 0; o < count; o += kSquareChunk) {
        const size_t ch = count - o < kSquareChunk ? count - o : kSquareChunk;
        RUN(k_relinearize(c, d_x3 + o * crc_ct_words(c, 3), ch, d_evk, dbc, d_y + o * crc_ct_words(c, 2), w + k_relin_keys_words(c, dbc), w, S(stream), false,
            false, o != 0));
    }
    return CRC_OK;
}
extern "C" int crc_square_relin_forms(crc_ctx *c, const uint64_t *d_x, int in_form, size_t count, const uint64_t *d_evk, int dbc, uint64_t *d_y, int out_form,
                                      void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_x || !d_y || !d_evk || !d_work || !form_ok(in_form) || !form_ok(out_form)) return CRC_ERR_INVALID_ARGUMENT;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    const size_t kSquareChunk = square_chunk(c);
    const size_t ch0 = count < kSquareChunk ? count : kSquareChunk;
    for (size_t o = 0; o < count; o += kSquareChunk) {
        const size_t ch = count - o < kSquareChunk ? count - o : kSquareChunk;
        // [packed keys (filled by pass 0)][size-3 intermediates][scratch]
        u64 *kp = w, *y3 = kp + k_relin_keys_words(c, dbc), *rest = y3 + ch0 * crc_ct_words(c, 3);
        RUN(k_square(c, d_x + o * crc_ct_words(c, 2), ch, y3, rest, S(stream), in_form == CRC_NTT, true));
        RUN(k_relinearize(c, y3, ch, d_evk, dbc, d_y + o * crc_ct_words(c, 2), rest, kp, S(stream), out_form == CRC_NTT, true, o != 0));
    }
    return CRC_OK;
}
// Square + relinearise + sum pooling as ONE key switch per pooled ciphertext (kernels_relin64.hip: relin_digits_pool_f64_kernel).  Internal passes take whole
// channel planes (a window never leaves its plane): [packed keys][size-3 squares of a pass][scratch]
static size_t sqpool_planes(const crc_ctx *c, int xd, int yd) { const size_t per = (size_t)xd * yd, ch = square_chunk(c); return ch / per ? ch / per : 1; }
extern "C" int crc_square_pool_relin_supported(const crc_ctx *c, int dbc, int xf, int yf)
{
    return c && c->tune.sq_path != 1 && c->tune.relin_path != 1 && k_relin64_pool_supported(c, dbc, xf * yf) ? 1 : 0;
}
extern "C" size_t crc_square_pool_relin_work_bytes(const crc_ctx *c, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int dbc)
{
    if (!c || xd < xf || yd < yf || xs < 1 || ys < 1) return 0;
    const size_t planes = (size_t)B * zd, pp = planes < sqpool_planes(c, xd, yd) ? planes : sqpool_planes(c, xd, yd);
    const size_t xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1, cin = pp * xd * yd, cout = pp * xo * yo;
    const size_t sq = k_square_work_words(c, cin), rl = k_relin_work_words(c, cout, dbc);
    return 8 * (k_relin_keys_words(c, dbc) + cin * crc_ct_words(c, 3) + (sq > rl ? sq : rl)) + 256;
}
extern "C" int crc_square_pool_relin_forms(crc_ctx *c, const uint64_t *d_x, int in_form, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf,
                                           const uint64_t *d_evk, int dbc, const uint64_t *d_div_ntt, uint64_t *d_y, int out_form, void *d_work, void *stream)
{
    CHECK_CTX(c); if (!d_x || !d_y || !d_evk || !d_work || !form_ok(in_form) || !form_ok(out_form)) return CRC_ERR_INVALID_ARGUMENT;
    if (d_div_ntt && out_form != CRC_NTT) return CRC_ERR_INVALID_ARGUMENT;           // the divisor multiplies slot-wise
    if (B < 0 || zd < 1 || xd < xf || yd < yf || xs < 1 || ys < 1 || xf < 1 || yf < 1) return CRC_ERR_INVALID_ARGUMENT;
    if (!crc_square_pool_relin_supported(c, dbc, xf, yf)) return CRC_ERR_UNSUPPORTED;
    u64 *w = (u64 *)(((uintptr_t)d_work + 255) & ~(uintptr_t)255);
    const PoolGeom pg{xd, yd, xs, ys, xf, yf, (xd - xf) / xs + 1, (yd - yf) / ys + 1};
    const size_t planes = (size_t)B * zd, step = sqpool_planes(c, xd, yd), pp0 = planes < step ? planes : step;
    const size_t pin = (size_t)xd * yd, pout = (size_t)pg.xo * pg.yo;
    u64 *kp = w, *y3 = kp + k_relin_keys_words(c, dbc), *rest = y3 + pp0 * pin * crc_ct_words(c, 3);
    for (size_t o = 0; o < planes; o += step) {
        const size_t pp = planes - o < step ? planes - o : step, cin = pp * pin, cout = pp * pout;
        RUN(k_square(c, d_x + o * pin * crc_ct_words(c, 2), cin, y3, rest, S(stream), in_form == CRC_NTT, true));
        if (o == 0) RUN(k_relin64_prepare_keys(c, d_evk, dbc, kp, rest, S(stream)));
        // (the (c0, c1) of a window are added up where the key switch's result meets them: relin_inv_crt_kernel)
        RUN(k_relinearize64(c, y3, 3, 2, y3, 3, cout, dbc, d_y + o * pout * crc_ct_words(c, 2), rest, kp, S(stream), out_form == CRC_NTT, &pg, d_div_ntt));
    }
    return CRC_OK;
}
extern "C" int crc_square_relin(crc_ctx *c, const uint64_t *d_x, size_t count, const uint64_t *d_evk, int dbc, uint64_t *d_y, void *d_work, void *stream)
{
    return crc_square_relin_forms(c, d_x, CRC_COEFF, count, d_evk, dbc, d_y, CRC_COEFF, d_work, stream);
}
