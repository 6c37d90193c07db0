// ctx.h -- engine context: BFV parameter set, host-side tables, their device copies.
//
// Mirrors what the reference spreads over SEALContext (context.cpp:15-169), SmallNTTTables (util/smallntt.cpp:37-92),
// the Evaluator ctor (evaluator.cpp:19-121) and the BaseConverter ctor (util/baseconverter.cpp:20-353), built
// MI355X-first: one flat table set in HBM, indexed by "modulus index" mi (0..k-1 = q_i, k..k+kb-1 = Bsk_j).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "modarith.h"
#include "f64mod.h"
#include "../../include/crcnn_hip.h"

#define CRC_MAXK 8            // coeff moduli (SEAL's largest default set, n=16384, has 8)
#define CRC_MAXB (CRC_MAXK + 2)

// constants of add_plain / transform_to_ntt(Plaintext) -- passed to kernels by value
struct PlainParams {
    u64 threshold;            // (t+1)>>1                          evaluator.cpp:73
    u64 inc[CRC_MAXK];        // (q - t) mod q_i = q_i - (t mod q_i); = q_i - t when q_i > t          evaluator.cpp:74-87
    int fast;                 // every q_i > t: c + inc needs no reduction (enable_fast_plain_lift, context.cpp:156-165); otherwise the lift of
                              // evaluator.cpp:1447-1463 (multi-word c + (q - t), then decompose) is computed as (c mod q_i) + inc mod q_i
    u64 delta[CRC_MAXK];      // floor(q/t) mod q_i                evaluator.cpp:66-70,96-100
    u64 uhi[CRC_MAXK];        // (q mod t) mod q_i                 evaluator.cpp:89-105
};

// BEHZ constants used by the Square pipeline -- lives in device memory (one copy per context)
struct BehzParams {
    int k, ka, kb;
    u64 t;
    u64 m_tilde, m_sk;
    u64 inv_qhat[CRC_MAXK];                  // (q/q_i)^-1 mod q_i
    u64 mt_inv_qhat[CRC_MAXK];               // m~ (q/q_i)^-1 mod q_i
    u64 qhat_mod_bsk[CRC_MAXB][CRC_MAXK];    // (q/q_i) mod Bsk_j
    u64 qhat_mod_mt[CRC_MAXK];               // (q/q_i) mod m~
    u64 inv_q_mod_mt;                        // q^-1 mod m~
    u64 q_mod_bsk[CRC_MAXB];                 // q mod Bsk_j
    u64 inv_mt_mod_bsk[CRC_MAXB];            // m~^-1 mod Bsk_j
    u64 inv_q_mod_bsk[CRC_MAXB];             // q^-1 mod Bsk_j
    u64 inv_mhat[CRC_MAXB];                  // (M/m_j)^-1 mod m_j
    u64 mhat_mod_q[CRC_MAXK][CRC_MAXB];      // (M/m_j) mod q_i
    u64 mhat_mod_msk[CRC_MAXB];              // (M/m_j) mod m_sk
    u64 inv_M_mod_msk;                       // M^-1 mod m_sk
    u64 M_mod_q[CRC_MAXK];                   // M mod q_i
    // Shoup companions floor(c 2^64 / modulus) of the per-modulus constants the base-conversion kernels multiply by, and the two
    // products that fold consecutive constant multiplications of the reference into one (same residue)
    u64 mt_inv_qhat_s[CRC_MAXK];
    u64 t_inv_qhat[CRC_MAXK], t_inv_qhat_s[CRC_MAXK];          // t (q/q_i)^-1 mod q_i   (evaluator.cpp:856-871 then baseconverter.cpp:413-423)
    u64 t_mod_bsk[CRC_MAXB], t_mod_bsk_s[CRC_MAXB];            // t mod Bsk_j
    u64 inv_mt_mod_bsk_s[CRC_MAXB], inv_q_mod_bsk_s[CRC_MAXB], inv_mhat_s[CRC_MAXB];
    u64 inv_M_mod_msk_s;
    // products of consecutive constant multiplications inside one modulus, so that a base conversion is ONE lazy 128-bit sum and ONE reduction
    // per target residue (same residue, hence the same bits):
    u64 lift_c[CRC_MAXB][CRC_MAXK];          // (q/q_i) m~^-1 mod Bsk_j          fastbconv_mtilde then mont_rq's final multiply (baseconverter.cpp:698-718, 619)
    u64 lift_r[CRC_MAXB];                    // q m~^-1 mod Bsk_j                 the r q term of mont_rq (:614-618)
    u64 floor_x[CRC_MAXB];                   // t q^-1 mod Bsk_j                  x t (evaluator.cpp:856-871) then fast_floor's q^-1 (:646-660)
    u64 floor_c[CRC_MAXB][CRC_MAXK];         // -(q/q_i) q^-1 mod Bsk_j           minus fastbconv(x) times q^-1
    u64 inv_qhat_s[CRC_MAXK];                // Shoup companion of inv_qhat (relinearisation digits are cut out of c2 (q/q_i)^-1, evaluator.cpp:984-985)
};

// The engine's own fp64 NTT primes (f64mod.h): the largest primes below 2^47 that are 1 mod 2^16.  Relinearisation's key-switching inner products
// sum_g digit_g (*) key_g are computed modulo these two and lifted by CRT (kernels_relin64.hip) -- passed to kernels by value
#define CRC_NF64 2
struct F64Params {
    F64Mod m[CRC_NF64];
    double inv_p0_p1, inv_p0_p1_q;                // p_0^-1 mod p_1 (centred) and its quotient companion (f64_mulmod_const)
    double ninv[CRC_NF64], ninv_q[CRC_NF64];      // n^-1 mod p_m (centred) + companion: folded into the keys, so the inverse transforms do not scale
    u64 p0_mod_q[CRC_MAXK];                       // p_0 mod q_j
};

// The auxiliary base of the ciphertext square over the engine's fp64 primes (kernels_square64.hip).  BEHZ's results do not depend on WHICH auxiliary base
// carries the intermediate integers as long as it is large enough (fastbconv_sk is exact once |floor(t P / q)| / B + #B < m_sk / 2): instead of SEAL's k (+1)
// 61-bit primes and m_sk the engine takes kf of its own 47-bit primes -- the fewest with prod p_j >= 4 n t q (1 + 2^-27 + 2^-40), decided in exact integers
// (ctx.cpp) --, p_0 .. p_{kf-2} as B and p_{kf-1} as m_sk, whose transforms and base conversions are fp64 arithmetic. Constants are {centred residue, residue /
// p} pairs for f64_mulmod_const; a 55..60-bit operand enters as its two 32-bit halves, hence the "x 2^32" twins.
#define CRC_NF64A 12
struct Sq64Params {
    int kf;                                       // primes in use (0: the parameters do not fit 12 primes -- the 61-bit base is used)
    F64Mod m[CRC_NF64A];
    double lift_c[CRC_NF64A][CRC_MAXK][4];        // (q/q_i) m~^-1 mod p_j: {w, w/p} and {2^32 w, 2^32 w / p}        (BehzParams.lift_c)
    double lift_r[CRC_NF64A][2];                  // q m~^-1 mod p_j                                                   (lift_r)
    double floor_x[CRC_NF64A][2];                 // t q^-1 n^-1 mod p_j: the inverse transforms do not scale           (floor_x)
    double floor_c[CRC_NF64A][CRC_MAXK][4];       // -(q/q_i) q^-1 mod p_j and its 2^32 twin                            (floor_c)
    double inv_mhat[CRC_NF64A][2];                // (B/p_j)^-1 mod p_j, j < kf - 1                                     (inv_mhat)
    double mhat_msk[CRC_NF64A][2];                // (B/p_j) mod m_sk                                                   (mhat_mod_msk)
    double inv_B_msk[2];                          // B^-1 mod m_sk                                                      (inv_M_mod_msk)
    u64 mhat_q[CRC_MAXK][CRC_NF64A];              // (B/p_j) mod q_i                                                    (mhat_mod_q)
    u64 B_q[CRC_MAXK];                            // B mod q_i                                                          (M_mod_q)
};

struct HostNtt {               // one modulus
    ModParams m;
    u64 root, inv_n;
    std::vector<u64> rp, srp, irp2, sirp2;   // bit-reversed powers + Shoup companions (layout as SEAL's tables)
    std::vector<u64> irp, sirp;              // the inverse powers NOT halved (round 5: the inverse transforms that scale once, at the end)
};

// Tuning switches of tools/ and the tests.  The environment is read ONCE, inside crc_ctx_create (a host application that calls setenv from another thread
// cannot race a launch); crc_ctx_set_tuning changes one value on a quiescent context.  None is needed for normal use.
struct CrcTuning {
    int no_fold = 0;              // CRC_NO_FOLD=1: generic Barrett reduction instead of the folding one
    int ntt_inv61_loose = 0;      // CRC_NTT_INV61_LOOSE=1: do not hold the 61-bit inverse transform to 64 VGPRs
    int mfma_order = -1;          // CRC_MFMA_ORDER=0|1: force the tile walk order of the limb GEMM (-1: by shape)
    int mfma_variant = 2;         // CRC_MFMA_VARIANT=2: two workgroups per CU (mfma_mac2w_kernel), 1: mfma_mac_kernel
    int mfma_ring = 0;            // CRC_MFMA_RING=4|5: LDS ring slots of mfma_mac_kernel
    int conv1_waves = 0;          // CRC_CONV1_WAVES=8|12
    int conv1_narrow = 1;         // CRC_CONV1_NARROW=0: 17-20 filters run the second filter group like a full one (round 4) instead of the packed form
    long long conv1_pass_bytes = 0;   // CRC_CONV1_PASS_BYTES: work-space cap per internal pass of a one-channel convolution (0: 16 GiB)
    int limb_pack_group = 1;      // CRC_LIMB_PACK_GROUP
    int mac2_cfg = 0;             // CRC_MAC2_CFG=16|8: force a tile shape (mac2_kernel)
    int mac_order = -1;           // CRC_MAC_ORDER=0|1
    int mac_regstage = 0;         // CRC_MAC_REGSTAGE=1: register-staged mac2_kernel instead of the LDS-DMA mac3_kernel
    int mac_stream = 1;           // CRC_MAC_STREAM=0: batch-1 dense layers on mac3_kernel (round 5) instead of the weight-stream kernel; 1..4: its shape (FT, SL)
    int mac2_dbg = 0;             // CRC_MAC2_DBG (only in -DCRC_TUNING builds)
    int ntt_split = 1;            // CRC_NTT_SPLIT=0: rows of n = 16384 as one 128-KiB LDS image (one workgroup per CU) instead of two 64-KiB halves
    int relin_mac_ct = 0;         // CRC_RELIN_MAC_CT=8: eight ciphertexts per thread in relin_mac_f64_kernel for k >= 4 (default 4)
    int mfma_min_steps = 0;       // CRC_MFMA_MIN_STEPS: reduction steps of 32 channels from which a conv / dense layer goes to the limb GEMM (0: 8)
    int f64_radix = 0;            // CRC_F64_RADIX=3|4|5: butterfly stages per LDS pass of the fp64 transforms (0: default)
    int sq_chunk = 0;             // CRC_SQ_CHUNK: ciphertexts per internal pass of square + relinearise (0: by ring size)
    // CRC_SQ_FUSE=1: an NTT-resident square lifts inside its forward fp64 transforms, 0: in a kernel of its own (round 3), -1: by k (fused up to k = 4)
    int sq_fuse = -1;
    // CRC_NTT_WAVE: the lazy 64-bit row transforms with one workgroup barrier per transform (ntt_rows_wave_kernel): bit 0 n = 8192, 1 n = 4096,
    // 2 n = 16384, 3 n = 16384 with the Square prologues, 4 inverse butterflies that halve per stage (round 4) instead of scaling once, 5 n = 2048 (round 6);
    // -1: measured (47 = bits 0 to 3 and 5)
    int ntt_wave = -1;
    int f64_wave = -1;
                                  // 2 K3, 3 the lifting forward kernel, 4 K3's 64-bit forward transform (with bit 2); -1: what measured faster
                                  // (profiles/r05_square_pool_wave_local_*.txt): 7; 0: round-4 kernels
    // CRC_SQ_PATH=0: by parameters, 1: the square's auxiliary base is SEAL's 61-bit one (round-2 kernels), 2: the engine's fp64 primes
    int sq_path = 0;
    int relin_path = 0;           // CRC_RELIN_PATH=0: by parameters, 1: key switching over the coefficient moduli (round-2 path), 2: over the two fp64 primes
};

struct crc_ctx {
    int n, logn, k, ka, kb, device;
    CrcTuning tune;
    u64 t;
    int total_bits;
    std::vector<u64> q;
    std::vector<HostNtt> tabs;               // k + kb entries
    PlainParams plain;
    BehzParams behz;
    std::vector<u64> qbig;                   // q as little-endian limbs (noise budget)
    // decrypt-only constants (host client side)
    ModParams tmod, gmod, mtmod;
    u64 qhat_mod_tg[2][CRC_MAXK], neg_inv_q_mod_tg[2], inv_gamma_mod_t, tgamma_mod_q[CRC_MAXK];
    // device copies
    ModParams *d_mods = nullptr;             // [k+kb]
    u64 *d_rp = nullptr, *d_irp2 = nullptr;   // [(k+kb)][n][2]: {bit-reversed root power, its Shoup companion} (forward / inverse-div-2)
    u64 *d_irp = nullptr;                     // the same layout, inverse powers not halved (ntt_rows_wave_kernel's unscaled inverse butterflies)
    BehzParams *d_behz = nullptr;
    u64 f64_primes[CRC_NF64A] = {0};
    int nf64 = CRC_NF64;                     // fp64 primes with transform tables: max(CRC_NF64, sq64.kf)
    F64Params f64;
    Sq64Params sq64;
    Sq64Params *d_sq64 = nullptr;
    double *d_f64_rp = nullptr, *d_f64_irp = nullptr;   // [nf64][n]: bit-reversed powers of psi_m (centred residues) forward; of psi_m^-1 inverse
    u64 *d_zero = nullptr;                   // 4 KiB of zeros (source row of reduction terms past T in mac3_kernel) + 4 KiB context scratch
    u64 *d_scratch = nullptr;                // = d_zero + 512 words: 256 two-word accumulator slots of crc_checksum64
    std::atomic<unsigned> scratch_next{0};
    int cus = 256;                           // compute units of THIS context's device
    // kernels whose dynamic-LDS limit has been raised on this context's device (hipFuncSetAttribute is a driver call: once, not per launch)
    std::mutex attr_mu;
    std::unordered_map<const void *, size_t> lds_attr;
};
int crc_ctx_ensure_lds(crc_ctx *c, const void *kernel, size_t lds_bytes);

int  crc_set_hip_error(hipError_t e);
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return crc_set_hip_error(e_); } while (0)

// host helpers shared by ctx.cpp / client.cpp / encoder.cpp
u64  h_mulmod(u64 a, u64 b, u64 q);
u64  h_powmod(u64 a, u64 e, u64 q);
u64  h_invmod(u64 a, u64 q);
void h_ntt_fwd(const HostNtt &T, u64 *a, int n);      // canonical in/out
void h_ntt_inv(const HostNtt &T, u64 *a, int n);
ModParams make_mod(u64 q, bool no_fold = false);
int  evk_digits(u64 q, int dbc);
