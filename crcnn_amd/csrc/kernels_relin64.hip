// kernels_relin64.hip -- relinearisation's key switching over TWO fp64 NTT primes instead of the k coefficient moduli.
//
// Reference: Evaluator::relinearize_one_step (evaluator.cpp:934-1069).  For every target modulus q_j and output polynomial the reference forms
//     R_j = sum_{i < k} sum_{d < L_i}  e_{i,d} (*) key_{i,d}[j]  mod q_j  ((*) = negacyclic product, e_{i,d} = digit d of c2 (q/q_i)^-1 mod q_i) with 4 k^2
//     forward transforms over the 55-bit primes (every digit polynomial under every q_j), 2 k inverse ones, and adds c0, c1.  The digits are below 2^dbc (16
//     bits) and the key residues below q_j, so over the INTEGERS |R_j| <= n D 2^dbc q_j / 2 (keys taken as centred residues) < 2^90: R_j is a fixed integer
//     polynomial, and its residue mod q_j -- all the reference needs -- can be had from its residues modulo ANY primes whose product exceeds 2 |R_j|.  We take
//     the two fp64 primes of the context (p_0 p_1 ~ 2^94, f64mod.h): 2 D forward transforms of 16-bit polynomials and 2 * 2k inverse ones per ciphertext in
//     6-flop fp64 arithmetic, the keys re-expressed once per call (inverse transform mod q_j, centre, reduce mod p_m, forward transform mod p_m, times n^-1), a
//     CRT lift per coefficient.  Same element of Z_q, hence the same bits (goldens: tests/golden/ops_*.npz ref_relin*, layers / nets).
//   K1 relin_digits_f64_kernel : one workgroup per (ciphertext, i): the L_i digits of row c2'_i, each transformed under p_0 and p_1 in LDS  -> E  [ct][g][m][n]
//   K2 relin_mac_f64_kernel  : per slot: A[ct][poly][j][m] = sum_g E[ct][g][m] Kf[g][poly][j][m]  (lazy sum of reduced products, < 42 p)  -> A  [ct][2k][m][n]
//   K3 relin_inv_crt_kernel    : one workgroup per (ciphertext, poly, j): both inverse transforms, CRT lift, mod q_j, + c_poly; coefficient form out, or the
//                                forward transform over q_j in the same LDS image for an NTT-resident result
#include "kernels.h"
#include "ntt_device.h"

#include "ntt_f64.h"


struct Relin64Tab { unsigned char L[CRC_MAXK], g0[CRC_MAXK]; };

// ---- key preparation (once per call) ------------------------------------------------------------------------------------------------------------------------
// evaluation keys as SEAL hands
// them over (NTT form over q_j, residues possibly lazy / non-canonical) -> canonical
__global__ void __launch_bounds__(256) evk_canon_kernel(const u64 *evk, u64 *out, const ModParams *mods, int n, int k)
{
    const size_t row = blockIdx.x;
    const ModParams m = mods[row % k];
    const u64 *src = evk + row * (size_t)n; u64 *dst = out + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) dst[s] = barrett128(src[s], 0, m);
}
// kc: the keys in coefficient form over q_j, rows [2 g + poly][j] (blob order)  ->  Kf [g][poly k + j][m][n]: centred residue mod p_m, forward transform, times
// n^-1
template <int RB>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024) relin_keys_f64_kernel(const u64 *kc, double *Kf, const ModParams *mods, const double *Wf,
    F64Params fp, int n, int logn, int k)
{
    extern __shared__ double smd[];
    const int m = blockIdx.x % CRC_NF64; const size_t row = blockIdx.x / CRC_NF64;       // row = (2 g + poly) k + j
    const int j = (int)(row % k); const size_t gp = row / k; const size_t g = gp >> 1; const int poly = (int)(gp & 1);
    const u64 q = mods[j].q;
    const F64Mod md = fp.m[m];
    const double *W = Wf + (size_t)m * n;
    const u64 *src = kc + row * (size_t)n;
    auto cen = [&](u64 v) { return f64_from_i64(v > (q >> 1) ? (long long)v - (long long)q : (long long)v, md); };
    for (int s = 2 * threadIdx.x; s < n; s += 2 * blockDim.x) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + s);
        sm_store_pair<RB>(smd, s, cen(v.x), cen(v.y));
    }
    __syncthreads();
    ntt_row_passes_f64<false, RB>(smd, W, n, logn, md);
    double *dst = Kf + (((g * 2 * k + (size_t)poly * k + j) * CRC_NF64) + m) * (size_t)n;
    for (int s = 2 * threadIdx.x; s < n; s += 2 * blockDim.x) {
        const d2 v = f64_stage_out<false, RB>(sm_load_pair<RB>(smd, s), W, n, logn, s, md);
        *reinterpret_cast<d2 *>(dst + s) = d2{f64_reduce(f64_mulmod_const(v.x, fp.ninv[m], fp.ninv_q[m], md.p), md), f64_reduce(f64_mulmod_const(v.y,
            fp.ninv[m], fp.ninv_q[m], md.p), md)};
    }
}

// ---- K1: digits of c2' under both primes --------------------------------------------------------------------------------------------------------------------
// src: size-`src_size` ciphertexts, poly `src_poly` = c2 (q/q_i)^-1 mod q_i (evaluator.cpp:984-985); E [ct][g][m][n], unreduced (|.| < 14 p) The source row is
// read ONCE and waits in registers (NPT points per thread) through the 2 L_i transforms cut from it: round 3 read it again in front of every transform, 8 times
// at 16-bit digits, and the L2 had long been swept by then (20 row reads per ciphertext for 3 rows at (8192, 3))
template <int RB, int NPT>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024, 4) relin_digits_f64_kernel(const u64 *src, int src_size, int src_poly, double *E, const double *Wf,
    F64Params fp, int n, int logn,
                                                                                                         int k, int D, int dbc, Relin64Tab tab)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / k; const int i = blockIdx.x % k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const u64 *row = src + ((ct * src_size + src_poly) * k + i) * (size_t)n;
    const u64 mask = (1ULL << dbc) - 1;
    const int L = tab.L[i], g0 = tab.g0[i];
    u64 r[NPT];
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) {
        const int s = 2 * (tid + u * nt);
        if (s < n) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(row + s); r[2 * u] = v.x; r[2 * u + 1] = v.y; }
    }
    for (int d = 0; d < L; d++) {
        const int sh = d * dbc;
        for (int m = 0; m < CRC_NF64; m++) {
            double *dst = E + ((ct * D + g0 + d) * CRC_NF64 + m) * (size_t)n;
#pragma unroll
            for (int u = 0; u < NPT / 2; u++) {
                const int s = 2 * (tid + u * nt);
                if (s < n) sm_store_pair<RB>(smd, s, (double)(u32)((r[2 * u] >> sh) & mask), (double)(u32)((r[2 * u + 1] >> sh) & mask));
            }
            __syncthreads();
            ntt_row_passes_f64<false, RB>(smd, Wf + (size_t)m * n, n, logn, fp.m[m]);
            f64_drain<false, RB, NPT / 4>(smd, Wf + (size_t)m * n, n, logn, fp.m[m], [&](int s, d2 v) { *reinterpret_cast<d2 *>(dst + s) = v; });
            __syncthreads();
        }
    }
}

// K1 for a Square layer with a sum pooling behind it: one workgroup per (POOLED ciphertext, i).  Sum_w relin(ct_w) = Sum_w c0_w + Sum_g (Sum_w digit_g(c2'_w))
// (*) key_g: the digit polynomials of the window's ciphertexts are added (integers below W 2^dbc) and transformed once -- the same element of Z_q as
// relinearising every ciphertext and adding the results (evaluator.cpp:934-1069 + poolingLayer.cpp:22-44), hence the same bits, with xo yo / (xd yd) of the
// transforms, inner products and inverse transforms.  Up to four digits per residue (16-bit digits of a 55..64-bit modulus), two 32-bit fields per register
// word. ONE: all digit sums of a value in one word -- fields of F = dbc + log2 W bits (the top digit of a 55-bit residue has 7 bits: 3 x 18 + 9 = 63 bits for a
// 2 x 2 window) -- so the window's sums take the registers the unpooled kernel spends on its one source row; otherwise two words of two 32-bit fields each
template <int RB, int NPT, bool ONE>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024, 4) relin_digits_pool_f64_kernel(const u64 *src, int src_size, int src_poly, double *E,
    const double *Wf, F64Params fp, int n, int logn,
                                                                                                              int k, int D, int dbc, Relin64Tab tab,
                                                                                                                  PoolGeom pg, int F)
{
    extern __shared__ double smd[];
    const size_t o = blockIdx.x / k; const int i = blockIdx.x % k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t per = (size_t)pg.xo * pg.yo, plane = o / per; const int rem = (int)(o % per), ox = rem / pg.yo, oy = rem % pg.yo;
    const size_t ct0 = (plane * pg.xd + (size_t)ox * pg.xs) * pg.yd + (size_t)oy * pg.ys;
    const u64 mask = (1ULL << dbc) - 1;
    const int L = tab.L[i], g0 = tab.g0[i];
    u64 lo[NPT], hi[ONE ? 2 : NPT];
#pragma unroll
    for (int u = 0; u < NPT; u++) { lo[u] = 0; if (!ONE) hi[u] = 0; }
    // (a fourth field exists only where 3 F < 64: with three digits per residue the host's test bounds (L - 1) F alone, and a shift by 64 or more is undefined)
    const bool four = 3 * F < 64;
    auto spread = [&](u64 v) {
        u64 w = (v & mask) | (((v >> dbc) & mask) << F) | (((v >> (2 * dbc)) & mask) << (2 * F));
        if (four) w |= (v >> (3 * dbc)) << (3 * F);
        return w;
    };
    for (int kx = 0; kx < pg.xf; kx++) for (int ky = 0; ky < pg.yf; ky++) {
        const u64 *row = src + (((ct0 + (size_t)kx * pg.yd + ky) * src_size + src_poly) * k + i) * (size_t)n;
#pragma unroll
        for (int u = 0; u < NPT / 2; u++) {
            const int s = 2 * (tid + u * nt);
            if (s < n) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(row + s);
                if (ONE) { lo[2 * u] += spread(v.x); lo[2 * u + 1] += spread(v.y); }
                else {
                    lo[2 * u] += (v.x & mask) | (((v.x >> dbc) & mask) << 32);          hi[(2 * u) % (ONE ? 2 :
                        NPT)] += ((v.x >> (2 * dbc)) & mask) | (((v.x >> (3 * dbc)) & mask) << 32);
                    lo[2 * u + 1] += (v.y & mask) | (((v.y >> dbc) & mask) << 32);      hi[(2 * u + 1) % (ONE ? 2 :
                        NPT)] += ((v.y >> (2 * dbc)) & mask) | (((v.y >> (3 * dbc)) & mask) << 32);
                }
            }
        }
    }
    const u64 fmask = (1ULL << F) - 1;
    for (int d = 0; d < L; d++) {
        for (int m = 0; m < CRC_NF64; m++) {
            double *dst = E + ((o * D + g0 + d) * CRC_NF64 + m) * (size_t)n;
#pragma unroll
            for (int u = 0; u < NPT / 2; u++) {
                const int s = 2 * (tid + u * nt);
                if (s < n) {
                    if (ONE) sm_store_pair<RB>(smd, s, (double)(u32)((lo[2 * u] >> (d * F)) & fmask), (double)(u32)((lo[2 * u + 1] >> (d * F)) & fmask));
                    else {
                        const u64 f0 = d < 2 ? lo[2 * u] : hi[(2 * u) % (ONE ? 2 : NPT)], f1 = d < 2 ? lo[2 * u + 1] : hi[(2 * u + 1) % (ONE ? 2 : NPT)];
                        sm_store_pair<RB>(smd, s, (double)(u32)(d & 1 ? f0 >> 32 : f0), (double)(u32)(d & 1 ? f1 >> 32 : f1));
                    }
                }
            }
            __syncthreads();
            ntt_row_passes_f64<false, RB>(smd, Wf + (size_t)m * n, n, logn, fp.m[m]);
            f64_drain<false, RB, NPT / 4>(smd, Wf + (size_t)m * n, n, logn, fp.m[m], [&](int s, d2 v) { *reinterpret_cast<d2 *>(dst + s) = v; });
            __syncthreads();
        }
    }
}

// K1 with ONE workgroup barrier per transform (round 5; ntt_f64.h, wave-local passes): n = 8192 (CS = 3) or 16384 (CS = 4), n / 16 threads.  The source row --
// or the digit sums of a pooling window, all four of a value in one word (relin_digits_pool_f64_kernel<.., ONE>) -- is held in the cross layout, every digit is
// cut straight into the registers the cross pass works on (no fill), and the transformed block leaves through the wave's own drain.  POOL: one workgroup per
// (pooled ciphertext, i); otherwise per (ciphertext, i).
template <int CS, bool POOL>
__global__ void __launch_bounds__(CS == 2 ? 256 : CS == 3 ? 512 : 1024, 4) relin_digits_wave_kernel(const u64 *src, int src_size, int src_poly, double *E, const double *Wf,
    F64Params fp, int n,
                                                                                int k, int D, int dbc, Relin64Tab tab, PoolGeom pg, int F)
{
    extern __shared__ double smd[];
    const size_t o = blockIdx.x / k; const int i = blockIdx.x % k;
    const u64 mask = (1ULL << dbc) - 1;
    const int L = tab.L[i], g0 = tab.g0[i];
    u64 r[16];
    auto load_row = [&](const u64 *row, bool first) {
        if (CS == 2 || CS == 3) {                                   // pairs: 16-byte loads (CS = 2: two pairs per block offset)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(row + f64_cross_point<CS == 2 ? 2 : 3>(2 * c));
                if (!POOL) { r[2 * c] = v.x; r[2 * c + 1] = v.y; continue; }
                const bool four = 3 * F < 64;
                auto spread = [&](u64 x) {
                    u64 w = (x & mask) | (((x >> dbc) & mask) << F) | (((x >> (2 * dbc)) & mask) << (2 * F));
                    if (four) w |= (x >> (3 * dbc)) << (3 * F);
                    return w;
                };
                r[2 * c] = (first ? 0 : r[2 * c]) + spread(v.x); r[2 * c + 1] = (first ? 0 : r[2 * c + 1]) + spread(v.y);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const u64 x = row[f64_cross_point<4>(c)];
                if (!POOL) { r[c] = x; continue; }
                const bool four = 3 * F < 64;
                u64 w = (x & mask) | (((x >> dbc) & mask) << F) | (((x >> (2 * dbc)) & mask) << (2 * F));
                if (four) w |= (x >> (3 * dbc)) << (3 * F);
                r[c] = (first ? 0 : r[c]) + w;
                if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four loads in flight, not sixteen: the sums already take 32 registers)
            }
        }
    };
    if (POOL) {
        const size_t per = (size_t)pg.xo * pg.yo, plane = o / per; const int rem = (int)(o % per), ox = rem / pg.yo, oy = rem % pg.yo;
        const size_t ct0 = (plane * pg.xd + (size_t)ox * pg.xs) * pg.yd + (size_t)oy * pg.ys;
        for (int kx = 0; kx < pg.xf; kx++) for (int ky = 0; ky < pg.yf; ky++)
            load_row(src + (((ct0 + (size_t)kx * pg.yd + ky) * src_size + src_poly) * k + i) * (size_t)n, kx == 0 && ky == 0);
    } else load_row(src + ((o * src_size + src_poly) * k + i) * (size_t)n, true);
    const int fw = POOL ? F : dbc;                           // width of a digit field in the held words
    const u64 fmask = (1ULL << fw) - 1;
    for (int d = 0; d < L; d++) {
        for (int m = 0; m < CRC_NF64; m++) {
            double *dst = E + ((o * D + g0 + d) * CRC_NF64 + m) * (size_t)n;
            const double *W = Wf + (size_t)m * n;
            f64_wave_forward<CS, 3>(smd, W, n, fp.m[m], [&](int q) { return (double)(u32)((r[q] >> (d * fw)) & fmask); });
            f64_local_drain<3>(smd, W, n, fp.m[m], [&](int s, d2 pr) { *reinterpret_cast<d2 *>(dst + s) = pr; });
            __syncthreads();
        }
    }
}

// ---- K2: slot-wise inner products ---------------------------------------------------------------------------------------------------------------------------
// A[ct][pj][m][s] = sum_g Kf[g][pj][m][s] E[ct][g][m][s]: every product reduced below 0.875 p, the sum of D <= 48 of them stays below 2^53 (exact); CT
// ciphertexts share every key value a thread loads A thread owns one slot of one prime for CT ciphertexts x PJ of the 2k key columns (PJS = 2k / PJ thread
// groups share the E values through L2): per digit CT + PJ loads feed CT PJ products -- the kernel is bound by L2 bandwidth on the key and digit values, not by
// the 7 flops per product
template <int K, int CT, int PJ>
__global__ void __launch_bounds__(256, CT * PJ <= 36 ? 4 : 1) relin_mac_f64_kernel(const double *E, const double *Kf, double *A, F64Params fp, int n, int D,
    size_t cnt)
{
    constexpr int PJS = 2 * K / PJ;
    const int sblocks = n / blockDim.x;
    // ciphertext group fastest: the workgroups that are resident together work on the SAME key tile (D x PJ values for 256 slots, 0.1-0.5 MiB: one L2 fill per
    // XCD) and stream only their own digit values -- with the slot block fastest every ciphertext group pulled the whole key set (9-17 MB, more than an L2)
    // again
    const unsigned groups = (unsigned)((cnt + CT - 1) / CT);
    unsigned b = blockIdx.x;
    const size_t ct0 = (size_t)(b % groups) * CT; b /= groups;
    const int s = (b % sblocks) * blockDim.x + threadIdx.x; b /= sblocks;
    const int pj0 = (b % PJS) * PJ; b /= PJS;
    const int m = b % CRC_NF64;
    const F64Mod md = fp.m[m];
    double acc[CT][PJ];
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
        for (int pj = 0; pj < PJ; pj++) acc[c][pj] = 0.0;
    const size_t nn = (size_t)n;
    // (round 5) the digit values of g + 1 -- the operand that comes from HBM; the key tile is L2-resident -- are requested before the CT x PJ products of digit
    // g are formed: the kernel waits on memory half of its time (profiles/r05_pmc_square_pool_8192_issue.json) and every iteration started with an exposed
    // round trip
    constexpr bool AHEAD = CT * PJ <= 36;                 // (the wider tiles have no registers to spare)
    double e[CT];
    auto fetch = [&](int g, double (&ev)[CT]) {
#pragma unroll
        for (int c = 0; c < CT; c++) ev[c] = ct0 + c < cnt ? E[(((ct0 + c) * D + g) * CRC_NF64 + m) * nn + s] : 0.0;
    };
    fetch(0, e);
    for (int g = 0; g < D; g++) {
        double en[CT];
        if (AHEAD && g + 1 < D) fetch(g + 1, en);
        const double *kr = Kf + (((size_t)g * 2 * K + pj0) * CRC_NF64 + m) * nn + s;
#pragma unroll
        for (int pj = 0; pj < PJ; pj++) {
            const double kv = kr[(size_t)pj * CRC_NF64 * nn];
#pragma unroll
            for (int c = 0; c < CT; c++) acc[c][pj] += f64_mulmod(kv, e[c], md);
        }
        if (g + 1 < D) {
            if (!AHEAD) fetch(g + 1, en);
#pragma unroll
            for (int c = 0; c < CT; c++) e[c] = en[c];
        }
    }
#pragma unroll
    for (int c = 0; c < CT; c++)
        if (ct0 + c < cnt)
#pragma unroll
            for (int pj = 0; pj < PJ; pj++) A[(((ct0 + c) * 2 * K + pj0 + pj) * CRC_NF64 + m) * nn + s] = acc[c][pj];
}

// ---- K3: inverse transforms, CRT lift, mod q_j, + (c0, c1) --------------------------------------------------------------------------------------------------
// A [ct][poly k + j][m][n]; x3: size-`add_size` ciphertexts whose polys 0, 1 are added (coefficient form); y [ct][2][k][n].  The first prime's result waits in
// registers (NPT points per thread, reduced) while the image serves the second transform -- round 3 parked it in its own row of A (6 rows written and read back
// per ciphertext at (8192, 3), and 11 spilled registers: the twiddle companions took the room) Workgroups of n / 16 threads, 16 points each: at n = 8192 two
// 512-thread workgroups per CU, four waves per SIMD and 128 registers -- with 1024 threads and 8 points the 64-register line of the second workgroup left no
// room for the held row (18 spilled registers, 16.6 instead of 12.2 MB of traffic per ciphertext and 5.55 instead of 5.19 us at (8192, 3):
// profiles/r04_square_relin_ab_step1.txt)
template <int RB, int NPT, bool OUT_NTT, bool LAZY>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024, 4) relin_inv_crt_kernel(const double *A, const u64 *x3, int add_size, u64 *y, const ModParams *mods,
    const double *Wi,
                                                                                                                     const ulonglong2 *Wq, F64Params fp,
                                                                                                                         int n, int logn, int k,
                                                                                                                         const u64 *mul, PoolGeom pg)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / (2 * k); const int pj = blockIdx.x % (2 * k), poly = pj / k, j = pj % k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double *a0row = A + ((ct * 2 * k + pj) * CRC_NF64) * (size_t)n;
    double a0[NPT];
    for (int m = 0; m < CRC_NF64; m++) {
        const double *src = a0row + (size_t)m * n;
        f64_fill_inv<RB, NPT / 2>(smd, Wi + (size_t)m * n, n, logn, fp.m[m], [&](int s) { return *reinterpret_cast<const d2 *>(src + s); });
        __syncthreads();
        // (A holds lazy sums below 2^52.4: the fused first stage reduces them while it fills the image, otherwise the first pass does)
        ntt_row_passes_f64<true, RB>(smd, Wi + (size_t)m * n, n, logn, fp.m[m], !f64_fused_stage<RB>(logn));
        if (m == 0) {
#pragma unroll
            for (int u = 0; u < NPT / 2; u++) {
                const int s = 2 * (tid + u * nt);
                if (s < n) { const d2 v = sm_load_pair<RB>(smd, s); a0[2 * u] = f64_reduce(v.x, fp.m[0]); a0[2 * u + 1] = f64_reduce(v.y, fp.m[0]); }
            }
            __syncthreads();
        }
    }
    // x = a0 + p0 t,  t = (a1 - a0) p0^-1 mod p1 centred: |x| < p0 p1 / 2, and the true value is below a quarter of that, so t is nowhere near +- p1 / 2
    const ModParams mq = mods[j];
    const u64 q = mq.q, p0q = fp.p0_mod_q[j];
    // pooled key switch (pg.xf > 0): `ct` counts pooled ciphertexts and the (c0, c1) of the window's ciphertexts in x3 are added up here -- no pooling pass, no
    // pooled copy
    size_t actw = ct;
    if (pg.xf > 0) {
        const size_t per = (size_t)pg.xo * pg.yo, plane = ct / per; const int rem = (int)(ct % per), ox = rem / pg.yo, oy = rem % pg.yo;
        actw = (plane * pg.xd + (size_t)ox * pg.xs) * pg.yd + (size_t)oy * pg.ys;
    }
    const u64 *add = x3 + ((actw * add_size + poly) * k + j) * (size_t)n;
    u64 *dst = y + ((ct * 2 + poly) * k + j) * (size_t)n;
    u64 *sm = reinterpret_cast<u64 *>(smd);           // (a thread reads and rewrites only its own slot of the image here: no barrier in between)
    auto lift = [&](double a0v, double a1r, u64 addv) {
        const double a1 = f64_reduce(a1r, fp.m[1]);
        const double t = f64_reduce(f64_mulmod_const(a1 - a0v, fp.inv_p0_p1, fp.inv_p0_p1_q, fp.m[1].p), fp.m[1]);
        const long long ti = (long long)t, a0i = (long long)a0v;
        u64 lo, hi; mul64wide((u64)(ti < 0 ? -ti : ti), p0q, lo, hi);
        u64 r = barrett128(lo, hi, mq);
        if (ti < 0) r = negmod(r, q);
        u64 a0m = (u64)(a0i < 0 ? -a0i : a0i);                    // |a0| < 2^46: below q for the 54..60-bit coefficient moduli, not for SEAL's 40-bit ones
        if (a0m >= q) a0m = barrett128(a0m, 0, mq);
        r = addmod(r, a0i < 0 ? negmod(a0m, q) : a0m, q);
        return addmod(r, addv, q);
    };
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) {
        const int s = 2 * (tid + u * nt);
        if (s < n) {
            const d2 a1 = sm_load_pair<RB>(smd, s);
            ulonglong2 av = *reinterpret_cast<const ulonglong2 *>(add + s);
            if (pg.xf > 0) {
                for (int w = 1; w < pg.xf * pg.yf; w++) {
                    const int kx = w / pg.yf, ky = w - kx * pg.yf;
                    const ulonglong2 bv = *reinterpret_cast<const ulonglong2 *>(add + ((size_t)kx * pg.yd + ky) * add_size * k * n + s);
                    av.x = addmod(av.x, bv.x, q); av.y = addmod(av.y, bv.y, q);
                }
            }
            const u64 r0 = lift(a0[2 * u], a1.x, av.x), r1 = lift(a0[2 * u + 1], a1.y, av.y);
            if (OUT_NTT) {
                const int a = swz<RB>(s);
                *reinterpret_cast<ulonglong2 *>(sm + (a & ~1)) = (a & 1) ? ulonglong2{r1, r0} : ulonglong2{r0, r1};
            } else *reinterpret_cast<ulonglong2 *>(dst + s) = ulonglong2{r0, r1};
        }
    }
    if (!OUT_NTT) return;
    // NTT-resident result: forward transform over q_j of (c_poly + R) in the same LDS image, same layout (ntt_device.h's radix-8 passes on the fp64 image's
    // swizzle)
    __syncthreads();
    ntt_row_passes<false, LAZY, RB, true>(sm, Wq + (size_t)j * n, n, logn, q, mq.two_q);
    const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
    // (the gap-1 stage of the forward transform is applied here, while the image is drained: ntt_device.h)
    const bool fuse1 = ntt_fused_stage(logn);
    for (int s = 2 * tid; s < n; s += 2 * nt) {
        const int a = swz<RB>(s);
        ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(sm + (a & ~1));
        if (a & 1) { const u64 t = v.x; v.x = v.y; v.y = t; }
        if (fuse1) fwd_pair_stage<LAZY>(v, Wq[(size_t)j * n + (n >> 1) + (s >> 1)], q, mq.two_q);
        if (LAZY) { v.x = reduce_small(v.x, q, mq.two_q, rq); v.y = reduce_small(v.y, q, mq.two_q, rq); }
        else { v.x = v.x >= mq.two_q ? v.x - mq.two_q : v.x; v.x = v.x >= q ? v.x - q : v.x; v.y = v.y >= mq.two_q ? v.y - mq.two_q : v.y;
            v.y = v.y >= q ? v.y - q : v.y; }
        if (mul) {          // the divisor of an average pooling behind the Square layer (an NTT-form plaintext [k][n]): slot-wise, while the result leaves
            const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(mul + (size_t)j * n + s);
            v.x = mulmod(v.x, w.x, mq); v.y = mulmod(v.y, w.y, mq);
        }
        *reinterpret_cast<ulonglong2 *>(dst + s) = v;
    }
}

// K3 with ONE workgroup barrier per transform (round 5; ntt_f64.h, wave-local passes): n = 8192 (CS = 3) or 16384 (CS = 4), n / 16 threads.  Both inverse
// transforms end in the cross pass, i.e. in registers: the first prime's result simply stays there (no LDS read, no parking), the second meets it for the CRT
// lift, the (c0, c1) rows are read in the same cross layout, and the canonical sums go into the image for the forward transform over q_j (or straight to
// memory). U64W: the forward transform over q_j that follows the CRT runs wave-locally too -- its cross pass works on the CRT's results where they are made, in
// registers
template <int CS, bool OUT_NTT, bool LAZY, bool U64W>
__global__ void __launch_bounds__(CS == 2 ? 256 : CS == 3 ? 512 : 1024, 4) relin_inv_crt_wave_kernel(const double *A, const u64 *x3, int add_size, u64 *y,
    const ModParams *mods, const double *Wi,
                                                                                 const ulonglong2 *Wq, F64Params fp, int n, int logn, int k, const u64 *mul,
                                                                                     PoolGeom pg)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / (2 * k); const int pj = blockIdx.x % (2 * k), poly = pj / k, j = pj % k;
    const int tid = threadIdx.x, nt = blockDim.x;
    const double *a0row = A + ((ct * 2 * k + pj) * CRC_NF64) * (size_t)n;
    // (two explicit blocks, not a loop over the primes: a loop keeps BOTH result arrays alive through both transforms)
    double a0[16];
    {
        // (A holds lazy sums below 2^52.4: the fill reduces them)
        f64_local_fill<3>(smd, Wi, n, fp.m[0], [&](int, int s) { return *reinterpret_cast<const d2 *>(a0row + s); });
        f64_wave_inverse<CS, 3>(smd, Wi, n, fp.m[0], [&](int q, double x) { a0[q] = f64_reduce(x, fp.m[0]); });
        __syncthreads();                                      // the image is filled again
    }
    double a1[16];
    {
        const double *src = a0row + n, *W = Wi + n;
        f64_local_fill<3>(smd, W, n, fp.m[1], [&](int, int s) { return *reinterpret_cast<const d2 *>(src + s); });
        f64_wave_inverse<CS, 3>(smd, W, n, fp.m[1], [&](int q, double x) { a1[q] = x; });
    }
    const ModParams mq = mods[j];
    const u64 q = mq.q, p0q = fp.p0_mod_q[j];
    size_t actw = ct;
    if (pg.xf > 0) {
        const size_t per = (size_t)pg.xo * pg.yo, plane = ct / per; const int rem = (int)(ct % per), ox = rem / pg.yo, oy = rem % pg.yo;
        actw = (plane * pg.xd + (size_t)ox * pg.xs) * pg.yd + (size_t)oy * pg.ys;
    }
    const u64 *add = x3 + ((actw * add_size + poly) * k + j) * (size_t)n;
    u64 *dst = y + ((ct * 2 + poly) * k + j) * (size_t)n;
    // x = a0 + p0 t,  t = (a1 - a0) p0^-1 mod p1 centred (relin_inv_crt_kernel)
    auto lift = [&](double a0v, double a1r, u64 addv) {
        const double a1v = f64_reduce(a1r, fp.m[1]);
        const double t = f64_reduce(f64_mulmod_const(a1v - a0v, fp.inv_p0_p1, fp.inv_p0_p1_q, fp.m[1].p), fp.m[1]);
        const long long ti = (long long)t, a0i = (long long)a0v;
        u64 lo, hi; mul64wide((u64)(ti < 0 ? -ti : ti), p0q, lo, hi);
        u64 r = barrett128(lo, hi, mq);
        if (ti < 0) r = negmod(r, q);
        u64 a0m = (u64)(a0i < 0 ? -a0i : a0i);
        if (a0m >= q) a0m = barrett128(a0m, 0, mq);
        r = addmod(r, a0i < 0 ? negmod(a0m, q) : a0m, q);
        return addmod(r, addv, q);
    };
    const int nw = pg.xf > 0 ? pg.xf * pg.yf : 1;
    u64 *sm = reinterpret_cast<u64 *>(smd);
    if (OUT_NTT) __syncthreads();                             // every wave has read its part of the image in the cross pass: the canonical sums go into it now
    if constexpr (CS == 2 || CS == 3) {                        // pairs (CS = 2: two per block offset; the cross layout's registers 2 c, 2 c + 1 are neighbours either way)
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int s = f64_cross_point<CS == 2 ? 2 : 3>(2 * c);
            ulonglong2 av = *reinterpret_cast<const ulonglong2 *>(add + s);
            for (int w = 1; w < nw; w++) {
                const int kx = w / pg.yf, ky = w - kx * pg.yf;
                const ulonglong2 bv = *reinterpret_cast<const ulonglong2 *>(add + ((size_t)kx * pg.yd + ky) * add_size * k * n + s);
                av.x = addmod(av.x, bv.x, q); av.y = addmod(av.y, bv.y, q);
            }
            const u64 r0 = lift(a0[2 * c], a1[2 * c], av.x), r1 = lift(a0[2 * c + 1], a1[2 * c + 1], av.y);
            if (OUT_NTT) {
                const int a = swz<3>(s);
                *reinterpret_cast<ulonglong2 *>(sm + (a & ~1)) = (a & 1) ? ulonglong2{r1, r0} : ulonglong2{r0, r1};
            } else *reinterpret_cast<ulonglong2 *>(dst + s) = ulonglong2{r0, r1};
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int s = f64_cross_point<4>(c);
            u64 av = add[s];
            for (int w = 1; w < nw; w++) { const int kx = w / pg.yf, ky = w - kx * pg.yf; av = addmod(av, add[((size_t)kx * pg.yd + ky) * add_size * k * n +
                s], q); }
            const u64 r0 = lift(a0[c], a1[c], av);
            if (OUT_NTT) sm[swz<3>(s)] = r0; else dst[s] = r0;
        }
    }
    if (!OUT_NTT) return;
    const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
    if constexpr (U64W) {
        // the forward transform wave-locally too.  The points a thread has just written are the cross groups it owns (two of eight at n = 8192, one of sixteen
        // at 16384): the cross stages read them back in program order -- no barrier --, one barrier later every wave finishes its own block.  (Round 5, first
        // form: the cross stages on the lifts' results in registers -- with both held transforms still alive that spilled 190-400 bytes per lane and lost 4 %.)
        const ulonglong2 *W = Wq + (size_t)j * n;
        constexpr int E = 16 >> CS, C = 1 << CS;
        // (nothing of the transform is scheduled into the lifts above: they fill the register file as it is)
        __builtin_amdgcn_sched_barrier(0);
        int t2 = tid;
        // (... nor are the sixteen image addresses of the lifts kept alive for it: they are computed again)
        asm volatile("" : "+v"(t2));
        auto own = [&](int c, int e) { return swz<3>(CS == 2 ? 4 * t2 + e + CRC_F64_BLOCK * c : CS == 3 ? 2 * t2 + e + CRC_F64_BLOCK * c : t2 + CRC_F64_BLOCK * c); };
#pragma unroll
        for (int e = 0; e < E; e++) {
            u64 x[C];
#pragma unroll
            for (int c = 0; c < C; c++) x[c] = sm[own(c, e)];
            fwd_stages<CS, LAZY>(x, W, 1, 0, q, mq.two_q);
#pragma unroll
            for (int c = 0; c < C; c++) sm[own(c, e)] = x[c];
        }
        __syncthreads();
        u64_local_passes_fwd<LAZY>(sm, W, n, q, mq.two_q);
#pragma unroll 2
        for (int u = 0; u < 8; u++) {                         // block-local drain through the gap-1 stage
            const int s = f64_local_pair(u), a = swz<3>(s);
            ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(sm + (a & ~1));
            if (a & 1) { const u64 t = v.x; v.x = v.y; v.y = t; }
            fwd_pair_stage<LAZY>(v, W[(n >> 1) + (s >> 1)], q, mq.two_q);
            if (LAZY) { v.x = reduce_small(v.x, q, mq.two_q, rq); v.y = reduce_small(v.y, q, mq.two_q, rq); }
            else {
                v.x = v.x >= mq.two_q ? v.x - mq.two_q : v.x; v.x = v.x >= q ? v.x - q : v.x;
                v.y = v.y >= mq.two_q ? v.y - mq.two_q : v.y; v.y = v.y >= q ? v.y - q : v.y;
            }
            if (mul) {
                const ulonglong2 wv = *reinterpret_cast<const ulonglong2 *>(mul + (size_t)j * n + s);
                v.x = mulmod(v.x, wv.x, mq); v.y = mulmod(v.y, wv.y, mq);
            }
            *reinterpret_cast<ulonglong2 *>(dst + s) = v;
        }
        return;
    }
    // NTT-resident result: forward transform over q_j of (c_poly + R) in the same LDS image (as relin_inv_crt_kernel)
    __syncthreads();
    ntt_row_passes<false, LAZY, 3, true>(sm, Wq + (size_t)j * n, n, logn, q, mq.two_q);
    const bool fuse1 = ntt_fused_stage(logn);
    for (int s = 2 * tid; s < n; s += 2 * nt) {
        const int a = swz<3>(s);
        ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(sm + (a & ~1));
        if (a & 1) { const u64 t = v.x; v.x = v.y; v.y = t; }
        if (fuse1) fwd_pair_stage<LAZY>(v, Wq[(size_t)j * n + (n >> 1) + (s >> 1)], q, mq.two_q);
        if (LAZY) { v.x = reduce_small(v.x, q, mq.two_q, rq); v.y = reduce_small(v.y, q, mq.two_q, rq); }
        else {
            v.x = v.x >= mq.two_q ? v.x - mq.two_q : v.x; v.x = v.x >= q ? v.x - q : v.x;
            v.y = v.y >= mq.two_q ? v.y - mq.two_q : v.y; v.y = v.y >= q ? v.y - q : v.y;
        }
        if (mul) {
            const ulonglong2 w = *reinterpret_cast<const ulonglong2 *>(mul + (size_t)j * n + s);
            v.x = mulmod(v.x, w.x, mq); v.y = mulmod(v.y, w.y, mq);
        }
        *reinterpret_cast<ulonglong2 *>(dst + s) = v;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------------------------------
// can this context / key set take the fp64 path?  2 |R_j| <= n D 2^dbc q_max must stay below p_0 p_1 / 2 (a factor 2 of slack for the floating CRT), the row
// must fit the LDS image the transforms work on, and at most 48 products may be summed lazily
bool k_relin64_supported(const crc_ctx *c, int dbc)
{
    if (c->n < 64 || c->n > 16384 || dbc < 1 || dbc > 32) return false;
    int D = 0, qbits = 0;
    for (int i = 0; i < c->k; i++) { D += evk_digits(c->q[i], dbc); if ((int)c->tabs[i].m.bits > qbits) qbits = c->tabs[i].m.bits; }
    if (D > 48) return false;
    int dbits = 0; while ((1 << dbits) < D) dbits++;
    // log2(n D 2^dbc q_max) <= logn + dbits + dbc + qbits  must be <= 2 * 47 - 2 (p_m > 2^46.99)
    return c->logn + dbits + dbc + qbits <= 2 * CRC_F64_PRIME_BITS - 3;
}
// the pooled form: the integer inner products are `window` times larger, a residue has at most four digits (two 32-bit fields per word hold their sums)
bool k_relin64_pool_supported(const crc_ctx *c, int dbc, int window)
{
    if (!k_relin64_supported(c, dbc) || window < 1 || window > 64 || dbc > 20) return false;
    int D = 0, qbits = 0;
    for (int i = 0; i < c->k; i++) { const int L = evk_digits(c->q[i], dbc); if (L > 4) return false; D += L;
        if ((int)c->tabs[i].m.bits > qbits) qbits = c->tabs[i].m.bits; }
    int dbits = 0; while ((1 << dbits) < D) dbits++;
    int wbits = 0; while ((1 << wbits) < window) wbits++;
    // 2 |R| <= n D W 2^dbc q_max <= 2^92 leaves |R| <= 2^91 < p_0 p_1 / 2 / 3.9: the CRT's t = (a1 - a0) p_0^-1 mod p_1 stays below p_1 / 3.9 in magnitude,
    // well inside the centred range it is reduced to (the unpooled test keeps one bit more; this one admits n = 16384 with all eight primes, D = 32, W = 4)
    return c->logn + dbits + wbits + dbc + qbits <= 2 * CRC_F64_PRIME_BITS - 2;
}
size_t k_relin64_keys_words(const crc_ctx *c, int dbc) { return (size_t)CRC_NF64 * crc_evk_words(c, dbc); }
// scratch words: E [cnt][D][2][n] + A [cnt][2k][2][n] (+ PM [cnt][k][n] when the caller's c2 is not premultiplied); the key preparation borrows the same space
size_t k_relin64_work_words(const crc_ctx *c, size_t cnt, int dbc)
{
    size_t D = 0; for (int i = 0; i < c->k; i++) D += evk_digits(c->q[i], dbc);
    const size_t n = c->n, k = c->k;
    const size_t run = cnt * n * (k + CRC_NF64 * D + CRC_NF64 * 2 * k), prep = crc_evk_words(c, dbc);
    return run > prep ? run : prep;
}

// threads per workgroup for passes of 2^RB values per thread
static int f64_threads(const crc_ctx *c, int RB) { int nt = c->n >> RB; if (nt < 64) nt = 64; if (nt > (RB == 5 ? 512 : 1024)) nt = RB == 5 ? 512 : 1024;
    return nt; }
// 3 stages (8 values per thread) per LDS pass by default: measured on (8192, 3) the digit kernel runs 0.90 / 1.08 / 1.34 us per ciphertext with 3 / 4 / 5 --
// the wider passes save LDS round trips and barriers but cost occupancy (76 / 134 registers), and the kernel is bound by instruction issue, not by LDS
// (profiles/r03_square_relin.txt)
static int f64_radix(const crc_ctx *c) { const int r = c->tune.f64_radix; return r >= 3 && r <= 5 ? r : 3; }
// threads of the kernels that keep a row in registers
// (16 points per thread at radix 8 and 16, 32 at radix 32: n / 16 threads, at least a wave, at most 1024)
static int f64_hold_threads(const crc_ctx *c, int RB) { int nt = c->n >> (RB == 5 ? 5 : 4); if (nt < 64) nt = 64;
    if (nt > (RB == 5 ? 512 : 1024)) nt = RB == 5 ? 512 : 1024; return nt; }

// kp: k_relin64_keys_words; scratch: crc_evk_words (k_relin64_work_words covers it)
int k_relin64_prepare_keys(crc_ctx *c, const u64 *evk, int dbc, u64 *kp, u64 *scratch, hipStream_t st)
{
    if (!k_relin64_supported(c, dbc)) return CRC_ERR_UNSUPPORTED;
    double *Kf = reinterpret_cast<double *>(kp);
    const size_t rows = crc_evk_words(c, dbc) / c->n;                 // (2 g + poly) k + j
    hipLaunchKernelGGL(evk_canon_kernel, dim3((unsigned)rows), dim3(256), 0, st, evk, scratch, c->d_mods, c->n, c->k);
    HIPCHK(hipGetLastError());
    int rc;
    if ((rc = k_ntt_ct(c, true, scratch, scratch, rows / c->k, 1, false, st, nullptr, 0, 0, 0))) return rc;
    const size_t lds = (size_t)c->n * 8;
    const int RB = f64_radix(c);
    auto kern = RB == 3 ? relin_keys_f64_kernel<3> : RB == 4 ? relin_keys_f64_kernel<4> : relin_keys_f64_kernel<5>;
    { const int r2 = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (r2) return r2; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(rows * CRC_NF64)), dim3(f64_threads(c, RB)), lds, st, scratch, Kf, c->d_mods, c->d_f64_rp, c->f64, c->n, c->logn,
        c->k);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

template <int K, int BIGCT = 4>
static int relin64_mac(crc_ctx *c, const double *E, const double *Kf, double *A, int D, size_t cnt, hipStream_t st)
{
    // accumulators: CT x PJ doubles per thread (32..48)
    constexpr int PJ = 2 * K <= 8 ? 2 * K : (2 * K) % 8 == 0 ? 8 : (2 * K) % 6 == 0 ? 6 : (2 * K) % 5 == 0 ? 5 : 7, CT = PJ <= 4 ? 8 : PJ <= 6 ? 6 : BIGCT,
        PJS = 2 * K / PJ;
    static_assert(PJ * PJS == 2 * K, "key columns must split evenly");
    const int threads = c->n < 256 ? c->n : 256, sblocks = c->n / threads;
    const size_t groups = (cnt + CT - 1) / CT;
    hipLaunchKernelGGL((relin_mac_f64_kernel<K, CT, PJ>), dim3((unsigned)(groups * CRC_NF64 * PJS * sblocks)), dim3(threads), 0, st, E, Kf, A, c->f64, c->n,
        D, cnt);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// the wave-local transforms (one workgroup barrier per transform) serve the two rings the bench configurations use; CRC_F64_WAVE=0 keeps the round-4 kernels
// (bit 1: the digit kernel, bit 2: K3.  Measured, same box: K1 -17 % at n = 8192, -16 % at 16384; K3 +3 % / +14 % in its first form, which parked the first
// prime's result in scratch, -0.3 % / -1.5 % of the whole sequence now -- its 64-bit forward transform is untouched: profiles/r05_square_pool_wave_local_*.txt)
static bool f64_wave_path(const crc_ctx *c, int RB, int bit)
{
    if (RB != 3 || (c->logn != 12 && c->logn != 13 && c->logn != 14)) return false;      // (n = 4096 since round 6: CS = 2)
    const int sel = c->tune.f64_wave < 0 ? 7 : c->tune.f64_wave;
    return (sel >> bit) & 1;
}

template <int RB, int NPT>
static int relin64_tail(crc_ctx *c, const double *A, const u64 *x3, int add_size, u64 *y, size_t cnt, bool out_ntt, hipStream_t st, const u64 *mul,
    const PoolGeom *pool)
{
    bool lazy = true;
    for (int i = 0; i < c->k; i++) if (c->tabs[i].m.bits > 57 || c->tabs[i].m.bits < 45) lazy = false;
    const size_t lds = (size_t)c->n * 8;
    if (f64_wave_path(c, RB, 2)) {
        const int cs = c->logn - 10;
        const bool u64w = f64_wave_path(c, RB, 4);           // (bit 4: the 64-bit forward transform behind the CRT wave-local as well)
#define K3W(A, B, C) (cs == 2 ? relin_inv_crt_wave_kernel<2, A, B, C> : cs == 3 ? relin_inv_crt_wave_kernel<3, A, B, C> : relin_inv_crt_wave_kernel<4, A, B, C>)
        auto kern = !out_ntt ? K3W(false, false, false) : lazy ? (u64w ? K3W(true, true, true) : K3W(true, true, false)) : K3W(true, false, false);
#undef K3W
        { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * 2 * c->k)), dim3(c->n / 16), lds, st, A, x3, add_size, y, c->d_mods, c->d_f64_irp,
                           reinterpret_cast<const ulonglong2 *>(c->d_rp), c->f64, c->n, c->logn, c->k, mul, pool ? *pool : PoolGeom{0, 0, 0, 0, 0, 0, 0, 0});
        HIPCHK(hipGetLastError());
        return CRC_OK;
    }
    auto kern = !out_ntt ? relin_inv_crt_kernel<RB, NPT, false, false> : lazy ? relin_inv_crt_kernel<RB, NPT, true, true> : relin_inv_crt_kernel<RB, NPT,
        true, false>;
    { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * 2 * c->k)), dim3(f64_hold_threads(c, RB)), lds, st, A, x3, add_size, y, c->d_mods, c->d_f64_irp,
                       reinterpret_cast<const ulonglong2 *>(c->d_rp), c->f64, c->n, c->logn, c->k, mul, pool ? *pool : PoolGeom{0, 0, 0, 0, 0, 0, 0, 0});
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// src / src_size / src_poly: where c2 (q/q_i)^-1 lives; x3 / add_size: the ciphertexts whose (c0, c1) are added; kp: the keys as k_relin64_prepare_keys left
// them; work: cnt n (2 D + 4 k) words
int k_relinearize64(crc_ctx *c, const u64 *src, int src_size, int src_poly, const u64 *x3, int add_size, size_t cnt, int dbc, u64 *y, u64 *work, const u64 *kp,
                    hipStream_t st, bool out_ntt, const PoolGeom *pool, const u64 *mul)
{
    if (cnt == 0) return CRC_OK;
    if (!k_relin64_supported(c, dbc)) return CRC_ERR_UNSUPPORTED;
    // (cnt counts POOLED ciphertexts then; src holds the unpooled ones)
    if (pool && !k_relin64_pool_supported(c, dbc, pool->xf * pool->yf)) return CRC_ERR_UNSUPPORTED;
    const size_t n = c->n, k = c->k;
    Relin64Tab tab{};
    int D = 0;
    for (int i = 0; i < c->k; i++) { tab.L[i] = (unsigned char)evk_digits(c->q[i], dbc); tab.g0[i] = (unsigned char)D; D += tab.L[i]; }
    const double *Kf = reinterpret_cast<const double *>(kp);
    int rc;
    double *E = reinterpret_cast<double *>(work), *A = E + cnt * D * CRC_NF64 * n;
    const size_t lds = n * 8;
    const int RB = f64_radix(c);
    if (pool) {
        // all digit sums of a value in one word when they fit: fields of F = dbc + log2 W bits, the top digit of residue i has bits_i - (L_i - 1) dbc (+ log2
        // W) bits
        int wbits = 0; while ((1 << wbits) < pool->xf * pool->yf) wbits++;
        const int F = dbc + wbits;
        bool one = true;
        for (int i = 0; i < c->k; i++) { const int L = tab.L[i]; if ((L - 1) * F + ((int)c->tabs[i].m.bits - (L - 1) * dbc + wbits) > 64 || F > 32) one =
            false; }
        if (one && f64_wave_path(c, RB, 1)) {
            auto kw = c->logn == 12 ? relin_digits_wave_kernel<2, true> : c->logn == 13 ? relin_digits_wave_kernel<3, true> : relin_digits_wave_kernel<4, true>;
            const int r3 = crc_ctx_ensure_lds(c, (const void *)kw, lds); if (r3) return r3;
            hipLaunchKernelGGL(kw, dim3((unsigned)(cnt * k)), dim3(c->n / 16), lds, st, src, src_size, src_poly, E, c->d_f64_rp, c->f64, c->n, c->k, D, dbc,
                tab, *pool, F);
            HIPCHK(hipGetLastError());
        } else {
        auto kern = RB == 3 ? (one ? relin_digits_pool_f64_kernel<3, 16, true> : relin_digits_pool_f64_kernel<3, 16, false>)
                  : RB == 4 ? (one ? relin_digits_pool_f64_kernel<4, 16, true> : relin_digits_pool_f64_kernel<4, 16, false>)
                            : (one ? relin_digits_pool_f64_kernel<5, 32, true> : relin_digits_pool_f64_kernel<5, 32, false>);
        const int r2 = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (r2) return r2;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * k)), dim3(f64_hold_threads(c, RB)), lds, st, src, src_size, src_poly, E, c->d_f64_rp, c->f64, c->n,
            c->logn, c->k, D, dbc, tab, *pool, F);
        HIPCHK(hipGetLastError());
        }
    } else if (f64_wave_path(c, RB, 1)) {
        auto kw = c->logn == 12 ? relin_digits_wave_kernel<2, false> : c->logn == 13 ? relin_digits_wave_kernel<3, false> : relin_digits_wave_kernel<4, false>;
        const int r3 = crc_ctx_ensure_lds(c, (const void *)kw, lds); if (r3) return r3;
        hipLaunchKernelGGL(kw, dim3((unsigned)(cnt * k)), dim3(c->n / 16), lds, st, src, src_size, src_poly, E, c->d_f64_rp, c->f64, c->n, c->k, D, dbc, tab,
                           PoolGeom{0, 0, 0, 0, 0, 0, 0, 0}, dbc);
        HIPCHK(hipGetLastError());
    } else {
        auto kern = RB == 3 ? relin_digits_f64_kernel<3, 16> : RB == 4 ? relin_digits_f64_kernel<4, 16> : relin_digits_f64_kernel<5, 32>;
        const int r2 = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (r2) return r2;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * k)), dim3(f64_hold_threads(c, RB)), lds, st, src, src_size, src_poly, E, c->d_f64_rp, c->f64, c->n,
            c->logn, c->k, D, dbc, tab);
        HIPCHK(hipGetLastError());
    }
    switch (c->k) {
#define MACK(KV) case KV: rc = c->tune.relin_mac_ct == 8 ? relin64_mac<KV, 8>(c, E, Kf, A, D, cnt, st) : relin64_mac<KV, 4>(c, E, Kf, A, D, cnt, st); break;
    MACK(1) MACK(2) MACK(3) MACK(4) MACK(5) MACK(6) MACK(7) MACK(8)
#undef MACK
    default: return CRC_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    if (mul && !out_ntt) return CRC_ERR_INVALID_ARGUMENT;
    // (pooled: x3 / add_size are the UNPOOLED ciphertexts whose (c0, c1) K3 adds up window by window)
    return RB == 3 ? relin64_tail<3, 16>(c, A, x3, add_size, y, cnt, out_ntt, st, mul, pool)
         : RB == 4 ? relin64_tail<4, 16>(c, A, x3, add_size, y, cnt, out_ntt, st, mul, pool) : relin64_tail<5, 32>(c, A, x3, add_size, y, cnt, out_ntt, st,
             mul, pool);
}
