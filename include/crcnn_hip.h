/*
 * crcnn_hip.h -- C ABI of the MI355X-native encrypted-CNN evaluation engine (libcrcnn_hip.so).
 *
 * Drop-in boundary for the reference's hot path.  The reference (CrCNN) has no FFI layer: its
 * evaluation path is `Layer::forward(ciphertext3D)` (CrCNN/src/layer.h:10-31) calling ten SEAL 2.3.1 `Evaluator`
 * methods (SEAL/seal/evaluator.cpp).  Each entry point below names the reference function(s) it replaces.  The C++
 * classes in crcnn_amd/host/ (Layer, ConvolutionalLayer, ..., Network, CnnBuilder) mirror the reference's interface
 * one-to-one and are implemented purely on top of this header.
 *
 * Conventions
 *   - plain C types only; every pointer named d_* is a DEVICE pointer (HBM), h_* is a host pointer.
 *   - a ciphertext of `size` polynomials is uint64[size][k][n], residues canonical in [0,q_i)  (SEAL keeps
 *     [size][k][n+1] with a dead zero pad word, ciphertext.cpp:103-130: crc_import/export_seal convert).
 *   - a ciphertext tensor is [B][C][H][W] of ciphertexts, row-major (CrCNN's ciphertext3D is [z][x][y]; B = image batch).
 *   - plaintext polynomials are uint64[n] coefficient vectors in [0,t); "ntt form" weights are uint64[k][n];
 *     "delta form" plaintexts (pre-scaled for add_plain/sub_plain) are uint64[k][n].
 *   - `form` flags say whether a ciphertext tensor is in coefficient form (CRC_COEFF, what the reference's layers
 *     exchange) or in NTT form (CRC_NTT, SEAL's transform_to_ntt ordering: slot j holds a(psi^(2*bitrev(j)+1))).
 *   - all functions return 0 on success or a negative crc_status; nothing throws across the ABI.  The reference
 *     signals errors by C++ exceptions (std::invalid_argument, evaluator.cpp:1549-1556) -- the C++ host classes turn a
 *     non-zero status back into std::invalid_argument / std::runtime_error.
 *   - a context is immutable after creation (crc_ctx_set_tuning excepted: tools and tests only, on a context nobody is launching on) and may be used
 *     from several host threads at once, provided every concurrent call has its own stream and its own `d_work`: every launch goes to the HIP
 *     stream passed in (`stream` is a hipStream_t cast to void*, NULL = default stream), no entry point keeps state between calls, and the one
 *     piece of context scratch (crc_checksum64's accumulators) is handed out per call (tests/test_gpu_threads.py: two host threads, two streams,
 *     a convolution on one and square + relinearise on the other, both against the reference's goldens).  No entry point allocates, frees or
 *     synchronises unless its name says so; scratch memory is passed in (`d_work`, sized by *_work_bytes).  The C++ host classes
 *     (crcnn_amd/host/crcnn_host.h) are NOT thread-safe: like the reference they keep the context, the keys and a work buffer in globals
 *     (CrCNN/src/globals.h:18-26) and launch on the default stream.
 *   - host-side calls that work item by item (crc_encode_f32 / _f64 / _compact, crc_encrypt / crc_encrypt_key) spread their items over up to CRC_HOST_THREADS
 *     std::threads (default: the hardware's, at most 16) drawn from one budget per process, so callers that are themselves pool threads do not multiply
 *     thread counts; results do not depend on the split (one keystream per ciphertext, one weight per plaintext)
 */
#ifndef CRCNN_HIP_H
#define CRCNN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crc_ctx crc_ctx;

typedef enum {
    CRC_OK = 0,
    CRC_ERR_INVALID_ARGUMENT = -1,   /* bad shape / null pointer / unsupported parameter (std::invalid_argument in the reference) */
    CRC_ERR_PARAMETERS = -2,         /* (n, q[], t) rejected: SEALContext::validate, context.cpp:15-169 */
    CRC_ERR_HIP = -3,                /* a HIP runtime call failed; crc_last_hip_error() has the code */
    CRC_ERR_UNSUPPORTED = -4,        /* valid in the reference but not implemented here (e.g. more than 48 relinearisation digits) */
    CRC_ERR_IO = -5,                 /* file could not be read / not an HDF5 file we understand */
    CRC_ERR_NOT_FOUND = -6,          /* dataset name missing in the model file */
    CRC_ERR_COMM = -7                /* an RCCL call failed; crc_last_comm_error() has the ncclResult_t */
} crc_status;

enum { CRC_COEFF = 0, CRC_NTT = 1,
       /* NTT form with every residue stored as its two 28-bit limbs (lo | hi << 32): the operand form of the multiply-accumulate
        * kernels (no 64x64 multiplier on gfx950; 3 v_mad_u64_u32 per product on 28-bit limbs).  Only crc_conv2d_forms /
        * crc_dense_forms take or produce it (weights and the tensors that travel between conv / dense layers); crc_pack28 converts.
        * Needs coefficient moduli below 2^56. */
       CRC_NTTP = 2,
       /* "limb form": the operand form of the matrix-core multiply-accumulate (kernels_mfma.hip).  Every residue as the seven balanced base-256 digits of
        * its centred representative (int8), SLOT-MAJOR: tensors [k][n][B][7][positions][2 polys][channels rounded up to 32] (a dense layer's input, one
        * position, is
        * K-blocked instead: [k][n][7][channels / 32][B * 2 rows = (image, poly)][32]), weights
        * [k][n][tap][channel block][7][filters rounded up to 64][32] (times 2^64 mod q: the kernel's Montgomery reduction divides it out).  Exact integer
        * arithmetic on v_mfma_i32_16x16x64_i8 (49 limb products per modular
        * multiply, int32 accumulators, one reduction per output): the same ciphertexts as every other form, about 4x the throughput of the vector-ALU
        * kernel on long reductions.  crc_limb_pack_weights makes the weights; crc_conv2d_forms / crc_dense_forms take w_form = CRC_NTTL, convert a
        * CRC_COEFF / CRC_NTT / CRC_NTTP input themselves and produce any form (out_form = CRC_NTTL hands the tensor to a DENSE layer: channels =
        * (filter, x, y) flattened, 1 x 1 positions).  Needs coefficient moduli below 2^55 (at most 55 significant bits: |centred residue| < 2^54 fits seven
        * balanced
        * bytes) and reductions of at most 18 000 terms; crc_limb_supported answers for a context and shape. */
       CRC_NTTL = 3,
       /* weights of a ONE-CHANNEL convolution (CrCNN's conv1, alone or fused with its pooling layer: window <= 8 x 8, <= 32 filters) for the matrix-core
        * kernel specialised for it (kernels_mfma1.hip): [k][n][2][7][32 filters][32 window taps], tap = 8 kx + ky.  w_form only (crc_limb_conv1_*). */
       CRC_NTTL1 = 4,
       /* out_form only, with w_form = CRC_NTTL1: the result as the limb tensor a CONVOLUTION reads with in_form = CRC_NTTL
        * ([k][n][B][7][xo*yo][2][32 channels]; CRC_NTTL as out_form flattens for a dense consumer instead) */
       CRC_NTTLC = 5 };

const char *crc_strerror(int status);
int         crc_last_hip_error(void);
int         crc_version(void);

/* ---------------------------------------------------------------------------------------------------------------
 * context   replaces: SEALContext (context.cpp:15-169) + Evaluator ctor tables (evaluator.cpp:19-121) + BaseConverter
 *           ctor (util/baseconverter.cpp:20-353) + SmallNTTTables (util/smallntt.cpp:37-92) as built by
 *           CrCNN setParameters (CrCNN/src/globals.cpp:25-56).  q may be any explicit list (coeff_modulus_128(n) or a
 *           prefix of it, as BASELINE.json's configs ask).
 * ------------------------------------------------------------------------------------------------------------- */
int  crc_ctx_create(int n, const uint64_t *q, int k, uint64_t t, int device, crc_ctx **out);
void crc_ctx_destroy(crc_ctx *ctx);
/* SEAL's default 128-bit-security moduli (util/globals.cpp:25-90); returns count, copies min(count,cap) */
int  crc_default_coeff_modulus_128(int n, uint64_t *q, int cap);
int  crc_ctx_n(const crc_ctx *ctx);
int  crc_ctx_k(const crc_ctx *ctx);
int  crc_ctx_kbsk(const crc_ctx *ctx);                 /* |Bsk| */
int  crc_ctx_device(const crc_ctx *ctx);
size_t crc_ct_words(const crc_ctx *ctx, int size);     /* size*k*n */
size_t crc_evk_words(const crc_ctx *ctx, int dbc);     /* words of an evaluation-key blob: sum_l 2*L_l*k*n */
/* named host-side table read-out (tests): "root","const_ratio","delta","upper_half_increment","bsk","bsk_root",
 * "root_powers:<i>","inv_root_powers_div_two:<i>", "f64_primes" (the two fp64 primes of relinearisation's key switching), "sq64_primes" (the fp64 primes that
 * carry the square's auxiliary base: B' = all but the last, m_sk' = the last; empty when the parameters do not fit twelve of them); returns word count */
int  crc_ctx_table(const crc_ctx *ctx, const char *name, uint64_t *h_out, int cap);
/* Tuning switches of tools/ and the tests (none is needed for normal use).  The engine reads its environment (CRC_MFMA_VARIANT, CRC_CONV1_PASS_BYTES, ...)
 * exactly
 * once, inside crc_ctx_create; this call changes one switch of a context nobody is launching on: "mfma_variant", "mfma_order", "mfma_ring", "conv1_waves",
 * "conv1_narrow" (0: a one-channel convolution with 17-20 filters runs its second filter group like a full one), "conv1_pass_bytes", "limb_pack_group", "mac2_cfg", "mac_order", "mac_regstage", "ntt_inv61_loose", "ntt_split", "mfma_min_steps", "f64_radix",
 * "relin_mac_ct",
 * "relin_path" (1: key switching over the coefficient moduli, as the reference does it, instead of over two fp64 primes), "sq_path" (1: the square's auxiliary
 * base is SEAL's 61-bit
 * one instead of the engine's fp64 primes; 2: force the latter), "sq_chunk" (ciphertexts per internal pass of square + relinearise; changes
 * crc_square_relin_work_bytes),
 * "sq_fuse" (1: an NTT-resident square lifts to its auxiliary base inside the forward transforms, 0: in a kernel of its own, -1: by the number of moduli),
 * "f64_wave" (bit mask of the fp64 row kernels that run with one workgroup barrier per transform at n = 8192 / 16384: 1 sq64_inv, 2 the digit kernel, 4 K3, 8
 * the lifting
 * forward kernel, 16 K3's 64-bit forward transform; -1: the measured choice, 0: the round-4 kernels), "ntt_wave" (the same for the 64-bit row transforms: 1 n =
 * 8192, 2 n = 4096, 4 n = 16384, 8 n = 16384 with the square's prologues, 16 inverse butterflies that halve at every stage as in round 4 instead of scaling once
 * at the end; -1: the measured choice = 15).
 * Every path gives the same ciphertexts.  CRC_ERR_NOT_FOUND for anything else. */
int  crc_ctx_set_tuning(crc_ctx *ctx, const char *name, long long value);

/* thin device-memory helpers so that C / C++ / ctypes callers need not link HIP themselves */
int crc_mem_info(crc_ctx *ctx, size_t *free_bytes, size_t *total_bytes);      /* hipMemGetInfo of the context's device */
int crc_malloc(crc_ctx *ctx, size_t bytes, void **d_ptr);
int crc_free(crc_ctx *ctx, void *d_ptr);
int crc_memcpy_h2d(crc_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, void *stream);
int crc_memcpy_d2h(crc_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, void *stream);
int crc_memcpy_d2d(crc_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream);
int crc_memset(crc_ctx *ctx, void *d_dst, int value, size_t bytes, void *stream);
int crc_stream_sync(crc_ctx *ctx, void *stream);
/* Streams of the caller's own (hipStream_t behind void*, created non-blocking: no implicit ordering against the default stream) and page-locked host memory:
 * what a host needs to upload the next chunk of encrypted images while the current one is evaluated -- the reference's driver encrypts, evaluates and decrypts
 * one image after the other (CrCNN/src/mainparams.cpp:85-112).  crc_stream_wait_event makes `stream` wait for an event recorded on another one. */
int crc_stream_create(crc_ctx *ctx, void **stream);
int crc_stream_destroy(crc_ctx *ctx, void *stream);
int crc_stream_wait_event(crc_ctx *ctx, void *stream, void *event);
int crc_host_alloc(crc_ctx *ctx, size_t bytes, void **h_ptr);
int crc_host_free(crc_ctx *ctx, void *h_ptr);
/* threads the host-side item loops of this process use (the reference's th_count fan-out, convolutionalLayer.cpp:177-191, has no process-wide cap):
 * CRC_HOST_THREADS,
 * else the hardware's, at most 16, divided by the ranks that share the node (LOCAL_WORLD_SIZE / CRC_LOCAL_WORLD) */
int crc_host_thread_limit(void);
/* HIP events (hipEvent_t behind void*): record on the stream the kernels go to, read the time between two of them (waits for the second) */
int crc_event_create(crc_ctx *ctx, void **event);
int crc_event_destroy(crc_ctx *ctx, void *event);
int crc_event_record(crc_ctx *ctx, void *event, void *stream);
int crc_event_elapsed_ms(crc_ctx *ctx, void *event_start, void *event_end, float *ms);

/* ---------------------------------------------------------------------------------------------------------------
 * encoding (host)   replaces: FractionalEncoder(t, x^n+1, 64, 32, 3)::encode/decode (encoder.cpp:1013-1076,
 *           1226-1270; instantiated at CrCNN/src/globals.cpp:52) as used by CnnBuilder::build*Layer (cnnBuilder.cpp:25-105)
 * ------------------------------------------------------------------------------------------------------------- */
/* values are widened float32 -> double exactly as `fraencoder->encode(weights[w])` does.  h_coeff_count (optional)
 * receives SEAL's Plaintext::coeff_count() for each value (needed only for wire-format compatibility). */
int    crc_encode_f32(const crc_ctx *ctx, const float *h_values, size_t count, uint64_t *h_plain /*[count][n]*/, int32_t *h_coeff_count);
int    crc_encode_f64(const crc_ctx *ctx, const double *h_values, size_t count, uint64_t *h_plain, int32_t *h_coeff_count);
double crc_decode(const crc_ctx *ctx, const uint64_t *h_plain /*[n]*/);
/* the same plaintexts in COMPACT form: the encoder only ever sets coefficients 0..63 (integer part: at most ceil(64 / log2 3) + 1 = 42 digits) and n-32..n-1
 * (fraction), so a weight travels as CRC_PLAIN_COMPACT_WORDS words -- words 0..63 = coefficients 0..63, words 64..95 = coefficients n-32..n-1 -- and
 * crc_plain_expand (below) zero-extends it on the device.  PlainModelWoPad's fc3 at n = 16384: 0.3 GB over PCIe instead of 52 GB. */
#define CRC_PLAIN_COMPACT_LOW   64
#define CRC_PLAIN_COMPACT_HIGH  32
#define CRC_PLAIN_COMPACT_WORDS 96
int    crc_encode_f32_compact(const crc_ctx *ctx, const float *h_values, size_t count, uint64_t *h_compact /*[count][96]*/, int32_t *h_coeff_count);
/* batch-norm parameters: invstd = float(1/sqrt(double(var)+0.00001))  (cnnBuilder.cpp:100-102) */
int    crc_bn_invstd_f32(const float *h_var, size_t count, float *h_invstd);

/* ---------------------------------------------------------------------------------------------------------------
 * plaintext preparation (device)
 *   crc_plain_to_ntt     replaces Evaluator::transform_to_ntt(Plaintext&) (evaluator.cpp:1418-1493): lift to each q_i
 *                        (c >= (t+1)/2 ? c + q_i - t : c) and forward NTT.     d_plain [count][n] -> d_out [count][k][n]
 *   crc_plain_to_delta   the Delta*m term of add_plain/sub_plain (evaluator.cpp:1168-1191): floor(q/t)*c (+ q mod t for
 *                        "negative" c) mod q_i.  form=CRC_NTT additionally NTTs it (for NTT-resident tensors).
 * ------------------------------------------------------------------------------------------------------------- */
int crc_plain_to_ntt(crc_ctx *ctx, const uint64_t *d_plain, size_t count, uint64_t *d_out, void *stream);
/* compact plaintexts on the device (crc_encode_f32_compact, copied down as they are) -> dense [count][n] for the two calls around this one */
int crc_plain_expand(crc_ctx *ctx, const uint64_t *d_compact /*[count][96]*/, size_t count, uint64_t *d_plain /*[count][n]*/, void *stream);
int crc_plain_to_delta(crc_ctx *ctx, const uint64_t *d_plain, size_t count, int form, uint64_t *d_out, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * NTT   replaces Evaluator::transform_to_ntt(Ciphertext&) / transform_from_ntt (evaluator.cpp:1495-1539) ->
 *       ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey (util/smallntt.h:210-258, smallntt.cpp:195-375).
 *       In place on `count` ciphertexts of `size` polys each; canonical output.
 * ------------------------------------------------------------------------------------------------------------- */
int crc_ntt_fwd(crc_ctx *ctx, uint64_t *d_ct, size_t count, int size, void *stream);
int crc_ntt_inv(crc_ctx *ctx, uint64_t *d_ct, size_t count, int size, void *stream);
/* same over the Bsk moduli (rows are [count][kbsk][n]); exposed for unit tests of the Square pipeline */
int crc_ntt_fwd_bsk(crc_ctx *ctx, uint64_t *d_rows, size_t count, void *stream);
int crc_ntt_inv_bsk(crc_ctx *ctx, uint64_t *d_rows, size_t count, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * element-wise Evaluator ops on `count` size-2 ciphertexts
 *   crc_add              Evaluator::add (evaluator.cpp:254-294)                      d_acc += d_b
 *   crc_add_plain        Evaluator::add_plain / sub_plain (:1145-1241) with a pre-scaled delta-form plaintext
 *                        shared by `group` consecutive ciphertexts (d_delta [count/group][k][n]); sign=+1/-1
 *   crc_multiply_plain_ntt  Evaluator::multiply_plain_ntt (:1541-1585): NTT-form ct times NTT-form plaintext
 *   crc_multiply_plain   Evaluator::multiply_plain generic path (:1343-1415): coefficient-form ct in, coefficient-form
 *                        out, NTT-form plaintext given (the reference re-NTTs it on every call)
 * ------------------------------------------------------------------------------------------------------------- */
int crc_add(crc_ctx *ctx, uint64_t *d_acc, const uint64_t *d_b, size_t count, int size, void *stream);
int crc_add_plain(crc_ctx *ctx, uint64_t *d_ct, const uint64_t *d_delta, size_t count, size_t group, int sign, void *stream);
int crc_multiply_plain_ntt(crc_ctx *ctx, uint64_t *d_ct, const uint64_t *d_w_ntt, size_t count, size_t group, int size, void *stream);
int crc_multiply_plain(crc_ctx *ctx, uint64_t *d_ct, const uint64_t *d_w_ntt, size_t count, size_t group, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * layers (batched over B images).  in_form / out_form: CRC_COEFF reproduces the reference layer exactly (coefficient
 * form in and out); CRC_NTT keeps tensors NTT-resident between layers (bit-identical after the final INTT).
 *
 *   crc_conv2d   ConvolutionalLayer::forward / convolution3d (CrCNN/src/convolutionalLayer.cpp:159-197, 56-93):
 *                y[b][f][i][j] = sum_{z,kx,ky} x[b][z][i*xs+kx][j*ys+ky] (*) w[f][z][kx][ky] + Delta*bias[f],
 *                valid padding, xo=(xd-xf)/xs+1, yo=(yd-yf)/ys+1.   d_w_ntt [nf][zd][xf][yf][k][n],
 *                d_bias_delta [nf][k][n] in the form of the OUTPUT (coeff for CRC_COEFF, NTT for CRC_NTT).
 *   crc_dense    FullyConnectedLayer::forward (fullyConnectedLayer.cpp:113-168) incl. reshapeInput (:38-56):
 *                x is [B][in_dim] cts in z,x,y row-major order; d_w_ntt [out_dim][in_dim][k][n].
 *   crc_pool     PoolingLayer::forward (poolingLayer.cpp:22-44) and AvgPoolingLayer::forward (avgPoolingLayer.cpp:16-45):
 *                window sum; if d_div_ntt != NULL multiply by that NTT-form plaintext (encode(1./(xf*yf))).
 *                form = CRC_NTTP: NTT-form input, output in the packed operand form (hand-over to a conv / dense layer).
 *   crc_batchnorm BatchNormLayer::forward (batchNormLayer.cpp:29-40): (x - Delta*mean[z]) (*) invstd[z].
 *                d_mean_delta [C][k][n] in the form of the INPUT, d_invstd_ntt [C][k][n].
 *   crc_square_relin  SquareLayer::forward (squareLayer.cpp:22-74) = Evaluator::square (evaluator.cpp:702-884) +
 *                relinearize (:886-1069) with decomposition-bit-count `dbc` keys.  Coefficient form in and out.
 *                d_evk: for l<k: [2*L_l][k][n] (= evaluation_keys.data()[0][l], pad words dropped), values may be
 *                SEAL's lazy non-canonical residues.  The result is the reference's ciphertext bit for bit; HOW it is computed differs where
 *                BFV leaves the evaluator a choice of moduli: BEHZ's auxiliary base (baseconverter.cpp:47-56 takes 61-bit primes) and the key-switching
 *                inner products (evaluator.cpp:997-1030 transforms every digit under every q_j) run over 47-bit primes of the engine's own in exact fp64
 *                arithmetic; crc_ctx_set_tuning "sq_path" / "relin_path" = 1 select kernels that follow the reference step by step.
 * ------------------------------------------------------------------------------------------------------------- */
size_t crc_conv2d_work_bytes(const crc_ctx *ctx, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int in_form);
int crc_conv2d(crc_ctx *ctx, const uint64_t *d_x, const uint64_t *d_w_ntt, const uint64_t *d_bias_delta,
               int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
               int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream);
/* Algebraic fusion of a convolution with the sum/average pooling that follows it (ConvolutionalLayer + PoolingLayer /
 * AvgPoolingLayer, convolutionalLayer.cpp:159-197 + poolingLayer.cpp:22-44 / avgPoolingLayer.cpp:16-45): both are linear over Z_q,
 * so pool(conv_w(x) + b) = conv_w'(x) + b' exactly, with w'[f][z][u][v] = div * sum_{a,b} w[f][z][u-a*cxs][v-b*cys], b' = div*pxf*pyf*b.
 * Produces the pooled kernel [nf][zd][xf'][yf'][k][n] (xf' = (pxf-1)*cxs + xf) and bias [nf][k][n], all in NTT form; run it with
 * crc_conv2d(..., xs = cxs*pxs, ys = cys*pys, xf', yf', out_form = CRC_NTT).  The final network output is bit-identical; only the
 * (unobservable in NTT-resident mode) intermediate tensor disappears.  d_div_ntt = NULL for sum pooling. */
int crc_conv2d_fold_pool(crc_ctx *ctx, const uint64_t *d_w_ntt, const uint64_t *d_bias_delta_ntt, const uint64_t *d_div_ntt, int nf, int zd, int xf, int yf,
                         int cxs, int cys, int pxf, int pyf, uint64_t *d_w_out, uint64_t *d_bias_out, void *stream);
/* the same two layers with the packed operand form: in_form / out_form may also be CRC_NTTP, w_form says how d_w_ntt is stored
 * (CRC_NTT canonical, CRC_NTTP packed by crc_pack28).  Same ciphertexts; nothing is re-split inside the kernels. */
int crc_conv2d_forms(crc_ctx *ctx, const uint64_t *d_x, const uint64_t *d_w_ntt, int w_form, const uint64_t *d_bias_delta,
                     int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf,
                     int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream);
int crc_dense_forms(crc_ctx *ctx, const uint64_t *d_x, const uint64_t *d_w_ntt, int w_form, const uint64_t *d_bias_delta,
                    int B, int in_dim, int out_dim, int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream);
/* limb form (CRC_NTTL): sizes, weight conversion (from CRC_NTT canonical weights; d_wl: crc_limb_weights_bytes), and the work space of
 * crc_conv2d_forms / crc_dense_forms when w_form = CRC_NTTL (crc_conv2d_work_bytes covers the other weight forms) */
int    crc_limb_supported(const crc_ctx *ctx, int zd, int xf, int yf);
size_t crc_limb_tensor_bytes(const crc_ctx *ctx, int B, int zd, int xd, int yd);
size_t crc_limb_weights_bytes(const crc_ctx *ctx, int nf, int zd, int xf, int yf);
int    crc_limb_pack_weights(crc_ctx *ctx, const uint64_t *d_w_ntt, int nf, int zd, int xf, int yf, void *d_wl, void *stream);
/* the same a filter tile at a time: d_w_tile_ntt holds filters f0 .. f0 + ft of the layer's nf ([ft][zd][xf][yf][k][n]); the tiles must be packed in order from
 * f0 = 0 (that call zeroes the padding of d_wl).  A layer whose canonical NTT-form weights and limb copy do not fit in HBM together (PlainModelWoPad's fc3 at
 * n = 16384: 202 + 177 GiB) is built this way straight from its plaintexts: encode -> crc_plain_to_ntt -> (batch-norm fold) -> tile, the canonical tile being
 * scratch */
int    crc_limb_pack_weights_tile(crc_ctx *ctx, const uint64_t *d_w_tile_ntt, int nf, int f0, int ft, int zd, int xf, int yf, void *d_wl, void *stream);
/* Kernel selection -- the ONE statement of the policy, asked by every host (crcnn_amd/netrun.py and the C++ classes of crcnn_amd/host):
 *   crc_plan_mac        the weight form (= kernel) of a conv / dense layer launched on B images (B <= 0: do not apply the rows-per-launch guard):
 *                       CRC_NTTL1 one-channel convolution on the matrix cores, CRC_NTTL limb GEMM (>= 8 reduction steps of 32 channels and >= 32 rows = images
 *                       x
 *                       2 polys x output pixels per launch), CRC_NTTP the vector-ALU kernel on 28-bit limb pairs, CRC_NTT canonical (moduli above 55 bits).
 *                       A dense layer is the 1 x 1 convolution zd = in_dim, nf = out_dim.  matrix_cores = 0 keeps everything on the vector ALU.
 *   crc_plan_fold_pool  whether folding a pooling layer into the convolution in front of it (crc_conv2d_fold_pool) pays, by the cost model of DESIGN.md section
 *   4 */
int    crc_plan_mac(const crc_ctx *ctx, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int B, int matrix_cores, int *w_form);
int    crc_plan_fold_pool(const crc_ctx *ctx, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int pxs, int pys, int pxf, int pyf, int *fold);
/* an NTT-form tensor (CRC_NTT canonical or CRC_NTTP) -> limb form; crc_conv2d_forms does this itself for such inputs, the separate entry point lets a
 * caller convert once and reuse (d_xl: crc_limb_tensor_bytes) */
int    crc_limb_pack_tensor(crc_ctx *ctx, const uint64_t *d_x, int in_form, int B, int zd, int xd, int yd, void *d_xl, void *stream);
/* the same for images b0 .. b0 + B of a limb tensor of Btot images (d_xl: crc_limb_tensor_bytes(Btot, ...)): several chunks assemble the input of one
 * dense-layer launch
 * (a dense layer streams all of its weights per launch, so it is run on as many images as fit: netrun's / Network's two-level chunking) */
int    crc_limb_pack_tensor_at(crc_ctx *ctx, const uint64_t *d_x, int in_form, int B, int zd, int xd, int yd, void *d_xl, int Btot, int b0, void *stream);
/* one-channel convolutions on the matrix cores (w_form = CRC_NTTL1): eligibility of a shape, size of the weights, conversion from CRC_NTT weights */
int    crc_limb_conv1_supported(const crc_ctx *ctx, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf);
size_t crc_limb_conv1_weights_bytes(const crc_ctx *ctx);
int    crc_limb_conv1_pack_weights(crc_ctx *ctx, const uint64_t *d_w_ntt, int nf, int xf, int yf, void *d_wl, void *stream);
size_t crc_conv2d_forms_work_bytes(const crc_ctx *ctx, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int nf, int in_form, int w_form,
int out_form);
/* in-place CRC_NTT <-> CRC_NTTP conversion of `rows` residue rows (unpack = 0: pack, 1: unpack) */
int crc_pack28(crc_ctx *ctx, uint64_t *d_rows, size_t rows, int unpack, void *stream);
size_t crc_dense_work_bytes(const crc_ctx *ctx, int B, int in_dim, int out_dim, int in_form);
int crc_dense(crc_ctx *ctx, const uint64_t *d_x, const uint64_t *d_w_ntt, const uint64_t *d_bias_delta,
              int B, int in_dim, int out_dim, int in_form, int out_form, uint64_t *d_y, void *d_work, void *stream);
int crc_pool(crc_ctx *ctx, const uint64_t *d_x, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf,
             const uint64_t *d_div_ntt /* NULL = sum pool */, int form, uint64_t *d_y, void *stream);
int crc_batchnorm(crc_ctx *ctx, uint64_t *d_x, int B, int zd, int xd, int yd, const uint64_t *d_mean_delta,
                  const uint64_t *d_invstd_ntt, int form, void *stream);
size_t crc_square_relin_work_bytes(const crc_ctx *ctx, size_t count, int dbc);
int crc_square_relin(crc_ctx *ctx, const uint64_t *d_x, size_t count, const uint64_t *d_evk, int dbc,
                     uint64_t *d_y, void *d_work, void *stream);
/* the same layer between NTT-resident neighbours: with in_form = CRC_NTT the given NTT values feed the products
 * and one inverse transform supplies the coefficients for the base extension; with out_form = CRC_NTT the tail adds
 * NTT(c0, c1) to the key-switched c2 instead of transforming back.  Identical ciphertexts in the requested form; 4k row
 * transforms fewer per ciphertext than converting outside. */
int crc_square_relin_forms(crc_ctx *ctx, const uint64_t *d_x, int in_form, size_t count, const uint64_t *d_evk, int dbc,
                           uint64_t *d_y, int out_form, void *d_work, void *stream);
/* Square followed by a SUM pooling (CrCNN's act1 -> pool2; Network::fuse pairs them): relinearisation is linear in the digit polynomials of c2, so the digits
 * of
 * a window's ciphertexts are added and ONE key switch serves the pooled ciphertext --
 *     Sum_w relin(ct_w) = Sum_w (c0, c1)_w + Sum_g (Sum_w digit_g(c2'_w)) (*) key_g
 * -- the same element of Z_q as squaring, relinearising and pooling one after the other (evaluator.cpp:934-1069, poolingLayer.cpp:22-44), hence the same bits,
 * with xo yo / (xd yd) of the key switch's transforms and inner products (16 / 25 for CrCNN's 5 x 5 -> 4 x 4 pool2).  d_x: [B][zd][xd][yd] ciphertexts, d_y:
 * [B][zd][xo][yo].  d_div_ntt: an average pooling's divisor (NTT-form plaintext [k][n], as crc_pool takes it), multiplied in while an NTT-form result leaves
 * the
 * last kernel (out_form must be CRC_NTT then).  crc_square_pool_relin_supported: the key switch over
 * the fp64 primes must hold the window's larger integers (n D W 2^dbc q at most 2^92, a quarter of p_0 p_1) and a residue at most four digits. */
int    crc_square_pool_relin_supported(const crc_ctx *ctx, int dbc, int xf, int yf);
size_t crc_square_pool_relin_work_bytes(const crc_ctx *ctx, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, int dbc);
int    crc_square_pool_relin_forms(crc_ctx *ctx, const uint64_t *d_x, int in_form, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf,
                                   const uint64_t *d_evk, int dbc, const uint64_t *d_div_ntt /* NULL = sum pool */, uint64_t *d_y, int out_form, void *d_work,
                                   void *stream);
/* the two halves separately (unit tests): square -> size-3 ciphertexts; relinearize -> size 2 */
int crc_square(crc_ctx *ctx, const uint64_t *d_x, size_t count, uint64_t *d_y3, void *d_work, void *stream);
int crc_relinearize(crc_ctx *ctx, const uint64_t *d_x3, size_t count, const uint64_t *d_evk, int dbc, uint64_t *d_y,
                    void *d_work, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * SEAL memory/wire layout <-> engine layout (host)   ciphertext.cpp:103-130 ([size][k][n+1], pad word zero)
 * ------------------------------------------------------------------------------------------------------------- */
int crc_import_seal(const crc_ctx *ctx, const uint64_t *h_seal, int size, uint64_t *h_out);
int crc_export_seal(const crc_ctx *ctx, const uint64_t *h_in, int size, uint64_t *h_seal);
/* SEAL 2.3.1 wire formats (byte-compatible with Ciphertext::save ciphertext.cpp:103-113, EvaluationKeys::save
 * evaluationkeys.cpp:8-39, PublicKey/SecretKey::save publickey.h:81-98 / secretkey.h:86-107), incl. the 32-byte SHA3-256
 * parameter hash (encryptionparams.cpp:69-100) that SEAL checks on every object.  *_load rejects a wrong hash / shape with
 * CRC_ERR_INVALID_ARGUMENT ("not valid for encryption parameters").  Buffers are host memory. */
int    crc_params_hash(const crc_ctx *ctx, uint64_t out[4]);
size_t crc_seal_ct_bytes(const crc_ctx *ctx, int size);
size_t crc_seal_evk_bytes(const crc_ctx *ctx, int dbc);
size_t crc_seal_pk_bytes(const crc_ctx *ctx);
size_t crc_seal_sk_bytes(const crc_ctx *ctx);
int crc_seal_ct_save(const crc_ctx *ctx, const uint64_t *h_ct, int size, void *buf, size_t cap, size_t *written);
int crc_seal_ct_load(const crc_ctx *ctx, const void *buf, size_t bytes, uint64_t *h_ct, int max_size, int *size, size_t *consumed);
int crc_seal_evk_save(const crc_ctx *ctx, const uint64_t *h_evk, int dbc, void *buf, size_t cap, size_t *written);
int crc_seal_evk_load(const crc_ctx *ctx, const void *buf, size_t bytes, uint64_t *h_evk, int *dbc);
int crc_seal_pk_save(const crc_ctx *ctx, const uint64_t *h_pk, void *buf, size_t cap, size_t *written);
int crc_seal_pk_load(const crc_ctx *ctx, const void *buf, size_t bytes, uint64_t *h_pk);
int crc_seal_sk_save(const crc_ctx *ctx, const uint64_t *h_sk_ntt, void *buf, size_t cap, size_t *written);
int crc_seal_sk_load(const crc_ctx *ctx, const void *buf, size_t bytes, uint64_t *h_sk_ntt);

/* ---------------------------------------------------------------------------------------------------------------
 * model loader (host)   replaces LoadH5::getDataVfloat (CrCNN/src/H5Easy.cpp:584-644) as used by
 *           CnnBuilder::getPretrained (cnnBuilder.cpp:20-23): flat float32 read of dataset `name` (e.g.
 *           "pool1_features.conv1.weight") from an HDF5 file written by PlainModel/ToH5.py.
 *           Two readers behind these entry points: a built-in one for the files the reference ships (superblock v0, contiguous little-endian
 *           float32 datasets; no dependency), and -- for every other layout: newer superblocks, chunked / compressed datasets, other float
 *           types -- libhdf5 itself, the library the reference links, loaded with dlopen when the machine has it (CRC_LIBHDF5 names one;
 *           CRC_H5_BACKEND=lite|hdf5 forces a reader).  A file neither can read gives CRC_ERR_IO.
 * ------------------------------------------------------------------------------------------------------------- */
int crc_h5_backend_available(void);                               /* 1 when libhdf5 (>= 1.10) could be loaded */
int crc_h5_dataset_count(const char *path, const char *name, size_t *count);
int crc_h5_read_f32(const char *path, const char *name, float *h_out, size_t cap, size_t *count);
int crc_h5_list(const char *path, char *h_names, size_t cap);   /* newline-separated dataset names */

/* ---------------------------------------------------------------------------------------------------------------
 * client side (host CPU; SURVEY 8f-2, outside the accelerated path): keygen / encrypt / decrypt so that a user of
 * CrCNN's globals.cpp (setParameters, encryptImage, decryptImage: globals.cpp:25-56,127-157,207-230) finds them.
 *
 * Randomness.  SEAL 2.3.1 draws from std::random_device (randomgen.cpp:7), so the reference fixes sampling LAWS (uniform
 * ternary secret / encryption sample, clipped normal sigma 3.19 cut at 6 sigma: util/globals.cpp:13-15), not bits.  Here every
 * sample comes from ChaCha20 keystreams under a 256-bit key:
 *   *_key entry points   take the key (CRC_KEY_BYTES bytes).  Draw it with crc_random_key (getrandom(2)) -- that is the
 *                        secure way and what the C++ host classes do on every setParameters().  One key may serve keygen,
 *                        evaluation keys and any number of encryptions: each use has its own stream (domain, `stream_base` +
 *                        ciphertext index, coefficient).  NEVER encrypt two different batches under the same (key, stream_base).
 *   uint64 seed variants expand a PUBLIC 64-bit seed into the key.  Deterministic by design: tests, bench, golden vectors.
 *                        NOT SECURE -- anyone who knows the seed can decrypt.
 * ------------------------------------------------------------------------------------------------------------- */
#define CRC_KEY_BYTES 32
int crc_random_key(uint8_t *h_key /*[CRC_KEY_BYTES]*/);
/* one ChaCha20 block (RFC 8439: 32-byte key, 32-bit block counter, 12-byte nonce -> 64 bytes): known-answer access to the generator */
int crc_chacha20_block(const uint8_t *h_key, uint32_t counter, const uint8_t *h_nonce /*[12]*/, uint8_t *h_out /*[64]*/);
int crc_keygen_key(const crc_ctx *ctx, const uint8_t *h_key, uint64_t *h_sk_ntt /*[k][n]*/, uint64_t *h_pk /*[2][k][n]*/);
int crc_gen_evk_key(const crc_ctx *ctx, const uint8_t *h_key, const uint64_t *h_sk_ntt, int dbc, uint64_t *h_evk);
int crc_encrypt_key(const crc_ctx *ctx, const uint64_t *h_pk, const uint64_t *h_plain, size_t count, const uint8_t *h_key, uint64_t stream_base,
                    uint64_t *h_ct /*[count][2][k][n]*/);
int crc_keygen(const crc_ctx *ctx, uint64_t seed, uint64_t *h_sk_ntt /*[k][n]*/, uint64_t *h_pk /*[2][k][n]*/);
int crc_gen_evk(const crc_ctx *ctx, uint64_t seed, const uint64_t *h_sk_ntt, int dbc, uint64_t *h_evk);
int crc_encrypt(const crc_ctx *ctx, const uint64_t *h_pk, const uint64_t *h_plain, size_t count, uint64_t seed, uint64_t *h_ct /*[count][2][k][n]*/);
int crc_decrypt(const crc_ctx *ctx, const uint64_t *h_sk_ntt, const uint64_t *h_ct, size_t count, int size, uint64_t *h_plain /*[count][n]*/);
int crc_noise_budget(const crc_ctx *ctx, const uint64_t *h_sk_ntt, const uint64_t *h_ct, int size);

/* Encryptor::encrypt (encryptor.cpp:71-134) on the device, for the 784 encryptions per image that dominate the client's
 * latency in the reference: d_pk = the public key of crc_keygen copied to the device ([2][k][n], NTT form), d_plain =
 * [count][n] plaintext coefficients (< t), d_ct = [count][2][k][n] coefficient form.  Sampling (ternary u, clipped-normal
 * e1/e2) is ChaCha20 in counter mode, one stream (one block) per (ciphertext, coefficient pair): same laws as crc_encrypt, different bits.
 * d_work: crc_encrypt_dev_work_bytes(count). */
size_t crc_encrypt_dev_work_bytes(const crc_ctx *ctx, size_t count);
int crc_encrypt_dev_key(crc_ctx *ctx, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, const uint8_t *h_key, uint64_t stream_base,
                        uint64_t *d_ct, void *d_work, void *stream);
int crc_encrypt_dev(crc_ctx *ctx, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, uint64_t seed, uint64_t *d_ct, void *d_work, void *stream);
/* The same with the form of the result chosen: CRC_COEFF (as above) or CRC_NTT -- c_p = NTT(e_p (+ Delta m)) + pk_p . NTT(u), three forward transforms per
 * modulus and no inverse one; the residues are those of crc_ntt_fwd applied to the coefficient-form result of the same (seed / key, stream) -- for a network
 * whose first layer takes NTT-form inputs (every convolution here does: the transform it would run on a coefficient-form image is skipped). */
int crc_encrypt_dev_key_forms(crc_ctx *ctx, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, const uint8_t *h_key, uint64_t stream_base, int out_form,
                              uint64_t *d_ct, void *d_work, void *stream);
int crc_encrypt_dev_forms(crc_ctx *ctx, const uint64_t *d_pk, const uint64_t *d_plain, size_t count, uint64_t seed, int out_form, uint64_t *d_ct, void *d_work,
                          void *stream);
/* The device encryptor samples its noise integers e in [-19, 19] directly from their law (the reference's N(0, 3.19^2) clipped at 6 sigma and truncated,
 * encryptor.cpp:237-240): |e| = the number of these 19 thresholds T_a = floor(2^64 P(|e| <= a)) that a uniform 64-bit word reaches.  For tests. */
void crc_encrypt_dev_noise_thresholds(uint64_t *h_out19);

/* Decryptor::decrypt (SEAL decryptor.cpp:107-236) on the device: d_sk_ntt = the secret key of crc_keygen copied to the device ([k][n], NTT form), d_ct =
 * [count][size][k][n] ciphertexts of `size` 2 or 3 in `in_form` CRC_COEFF or CRC_NTT (an NTT-resident tensor is decrypted as it stands: c0 + c1 s is formed in
 * the NTT domain and ONE inverse transform per residue follows), d_plain = [count][n] plaintext coefficients below t -- the polynomial crc_decrypt and the
 * reference produce, bit for bit (dot product with the secret key, inverse transform, BEHZ correction with the auxiliary prime gamma:
 * util/baseconverter.cpp:744-797).  d_work: crc_decrypt_dev_work_bytes(count, size, in_form).  Asynchronous on `stream`. */
size_t crc_decrypt_dev_work_bytes(const crc_ctx *ctx, size_t count, int size, int in_form);
int crc_decrypt_dev(crc_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_ct, size_t count, int size, int in_form, uint64_t *d_plain, void *d_work,
                    void *stream);
/* FractionalEncoder::decode / encode (encoder.cpp:1226-1270, 1013-1076; 64 integer + 32 fractional coefficients, base 3: CrCNN/src/globals.cpp:52) on the
 * device: the doubles crc_decode returns for d_plain [count][n], and the dense plaintexts [count][n] crc_encode_f32 / _f64 make of the values -- the same IEEE
 * operations in the same order as the host encoder, contraction off. */
int crc_decode_dev(crc_ctx *ctx, const uint64_t *d_plain, size_t count, double *d_out, void *stream);
int crc_encode_dev_f32(crc_ctx *ctx, const float *d_values, size_t count, uint64_t *d_plain, void *stream);
int crc_encode_dev_f64(crc_ctx *ctx, const double *d_values, size_t count, uint64_t *d_plain, void *stream);
/* The client-side refresh of Network::forward (CrCNN/src/network.cpp:30-34: `floatCube image = decryptImage(input); input = encryptImage(image)`,
 * globals.cpp:207-230 and 144-157) for `count` ciphertexts at once, entirely on `stream`: decrypt -> decode -> float (globals.cpp:221 keeps floats) ->
 * encode -> Encryptor::encrypt with fresh randomness (seed / key + stream_base as crc_encrypt_dev[_key]_forms).  in_form / out_form: CRC_COEFF or CRC_NTT.
 * d_values_out (may be NULL): the `count` floats the client saw.  d_ct_out may be d_ct_in.  d_work: crc_refresh_dev_work_bytes(count, in_form). */
size_t crc_refresh_dev_work_bytes(const crc_ctx *ctx, size_t count, int in_form);
int crc_refresh_dev(crc_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_pk, const uint64_t *d_ct_in, size_t count, int in_form, uint64_t seed,
                    int out_form, uint64_t *d_ct_out, float *d_values_out, void *d_work, void *stream);
int crc_refresh_dev_key(crc_ctx *ctx, const uint64_t *d_sk_ntt, const uint64_t *d_pk, const uint64_t *d_ct_in, size_t count, int in_form,
                        const uint8_t *h_key, uint64_t stream_base, int out_form, uint64_t *d_ct_out, float *d_values_out, void *d_work, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * multi-GPU (SURVEY 8e / 8b `crc_broadcast_weights`).  The reference has no analogue: its only parallelism is the
 * std::thread fan-out inside a layer (convolutionalLayer.cpp:177-191).  Here a batch of encrypted images shards over the
 * GPUs of a node with NO data-path collective; the ONE collective is the start-up broadcast of the encoded (NTT-form)
 * weights and evaluation keys from the rank that built them, over RCCL (ncclBroadcast; xGMI inside a node).
 *
 *   crc_comm_unique_id    the root makes the 128-byte rendezvous id and hands it to the other ranks out of band
 *                         (file, socket, MPI, torch.distributed ...)
 *   crc_comm_create       one communicator per (process, GPU): rank `rank` of `world` on the context's device
 *   crc_comm_create_all   single-process alternative: one communicator per context / device (ncclCommInitAll); use the
 *                         *_all broadcast below, or one host thread per communicator
 *   crc_broadcast_weights in-place broadcast of `words` uint64 from `root` in <= 1 GiB pieces on `stream` (asynchronous:
 *                         ordered with the kernels of that stream, no host sync)
 *   crc_comm_allgather_u64  small host-to-host all-gather (per-rank checksums, timings); synchronises `stream`
 *   crc_checksum64        position-sensitive checksum of a device buffer: h_out[0] = xor of all words, h_out[1] =
 *                         sum_i w_i * (2i+1) mod 2^64; synchronises `stream`.  Every rank checks what it received against the
 *                         root's pair.
 * Rehearsal on one GPU: RCCL refuses two ranks on the same device.  With CRC_COMM_TRANSPORT=shm in the environment crc_comm_unique_id makes a POSIX
 * shared-memory segment (an unguessable name, created exclusively, mode 0600) instead of an RCCL rendezvous, names it in the id, and the same calls stage their
 * bytes through it (hipMemcpy, a process-shared barrier): the multi-rank HOST code above this header runs unchanged with several processes on one device.
 * A transport for tests only; never the default.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct crc_comm crc_comm;
#define CRC_COMM_ID_BYTES 128
int  crc_comm_unique_id(uint8_t *h_id /*[CRC_COMM_ID_BYTES]*/);
int  crc_comm_create(crc_ctx *ctx, int world, int rank, const uint8_t *h_id, crc_comm **out);
int  crc_comm_create_all(crc_ctx *const *ctxs, int ndev, crc_comm **out /*[ndev]*/);
void crc_comm_destroy(crc_comm *comm);
int  crc_comm_rank(const crc_comm *comm);
int  crc_comm_world(const crc_comm *comm);
int  crc_last_comm_error(void);                        /* the ncclResult_t of the last failed RCCL call (CRC_ERR_COMM) */
int  crc_broadcast_weights(crc_comm *comm, uint64_t *d_w, size_t words, int root, void *stream);
int  crc_broadcast_weights_all(crc_comm *const *comms, int ndev, uint64_t *const *d_w, size_t words, int root, void *const *streams);
int  crc_comm_allgather_u64(crc_comm *comm, const uint64_t *h_in, size_t words, uint64_t *h_out /*[world][words]*/, void *stream);
int  crc_checksum64(crc_ctx *ctx, const uint64_t *d_words, size_t words, uint64_t *h_out /*[2]*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif
