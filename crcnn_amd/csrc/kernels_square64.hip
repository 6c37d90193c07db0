// kernels_square64.hip -- the auxiliary-base half of the BFV ciphertext square over the engine's own fp64 primes.
//
// Reference: Evaluator::square (evaluator.cpp:702-884) with BaseConverter::fastbconv_mtilde / mont_rq / fast_floor / fastbconv_sk (util/baseconverter.cpp:388-742).
// BEHZ carries the tensor product in TWO bases: q (the coefficient moduli) and an auxiliary base B U {m_sk} that only has to be large enough to hold the integers
//     c' = (S + q r) / m~  (|c'| < q (1 + k 2^-32)),   P = c'_a (*) c'_b   (|P| <= 2 n q^2),   R = (t P - fastbconv_q(t P)) / q   (|R| <= 2 n t q + k)
// exactly: fastbconv_q's sums S are integers that do not depend on where they are reduced, and fastbconv_sk returns R mod q_i exactly as soon as
// |R| / B + #B + 1 < m_sk / 2.  The reference takes k (+1) 61-bit primes and a 61-bit m_sk; ANY base of that size gives the same R, hence the same residues mod
// q_i (tests: the oracle's restatement with SEAL's base, and every golden of the reference itself).  The engine takes kf of its own 47-bit fp64 primes (ctx.h
// Sq64Params: the fewest with prod p_j >= 4 n t q): 5 instead of 4 rows per polynomial at (8192, k = 3), 6 instead of 5 at (16384, 4), 11 instead of 9 at
// (16384, 8) -- but a row transform in 6-flop fp64 arithmetic costs half of one over a 61-bit modulus, which has no lazy form (profiles/r03_square_relin_*:
// 0.041 against 0.069-0.088 us per row at n = 8192, 0.090 against 0.16-0.21 at 16384).  The q half (dyadic products and inverse transforms over the coefficient
// moduli) stays in kernels.hip.
//   sq64_lift_kernel  : q -> {p_j}: fastbconv_mtilde + mont_rq, residues written as centred doubles                                  LB [ct][2][kf][n]
//   sq64_fwd_kernel   : forward transform of every LB row in place
//   sq64_liftfwd_kernel (round 4, NTT-resident callers): the two above in one -- the inverse transform that brings x to coefficient form leaves it multiplied
//   by
//                       m~ (q/q_i)^-1 (NttArgs prologue 5), and the forward transform under p_j lifts its row while it stages it: the kf workgroups of a
//                       polynomial
//                       share their k source rows through one XCD's L2, LB is written once and never read back untransformed (-20 of 260 row transfers per
//                       ciphertext at (8192, 3), one kernel less)
//   sq64_inv_kernel   : a^2, 2ab, b^2 formed while a row is staged, inverse transform (unscaled: n^-1 sits in floor_x)                   DB [ct][3][kf][n]
//   sq64_floor_kernel : x t, fast_floor, fastbconv_sk back to q (one lazy 128-bit sum + one reduction per q_i), optional (q/q_i)^-1 on c2 for relinearisation
#include "kernels.h"
#include "ntt_f64.h"

// every (k, kf) a context can have is not known at compile time (kf depends on t): instances for k = 1..8 with the kf range 3..12 the size rule can produce for
// 40-60-bit q_i
#define CRC_FOR_ALL_K_KF(X) \
    X(1, 3) X(1, 4) X(2, 3) X(2, 4) X(2, 5) X(3, 4) X(3, 5) X(3, 6) X(4, 5) X(4, 6) X(4, 7) X(5, 6) X(5, 7) X(5, 8) X(5, 9) X(6, 7) X(6, 8) X(6, 9) X(6, 10) \
    X(7, 8) X(7, 9) X(7, 10) X(7, 11) X(8, 9) X(8, 10) X(8, 11) X(8, 12)

namespace {
struct acc128 { u64 lo, hi; };
__device__ __forceinline__ void acc_mad(acc128 &a, u64 x, u64 y)
{
    u64 pl, ph; mul64wide(x, y, pl, ph);
    const u64 nl = a.lo + pl; a.hi += ph + (nl < pl); a.lo = nl;
}
// (hi 2^32 + lo) w mod p from the constant's two pairs {w, w/p, 2^32 w, 2^32 w / p}: both halves are below 2^32, each product comes back below 0.875 p
__device__ __forceinline__ double mul_split(double hi, double lo, const double *c, double p)
{
    return f64_mulmod_const(lo, c[0], c[1], p) + f64_mulmod_const(hi, c[2], c[3], p);
}
}

// x: [count][2][K][n] coefficient form over q  ->  out: [count][2][KF][n] doubles, |.| <= (p_j + 1) / 2.  Two neighbouring coefficients per thread: 16-byte
// loads and stores
template <int K, int KF>
__global__ void __launch_bounds__(256) sq64_lift_kernel(const u64 *x, double *out, const ModParams *mods, const BehzParams *bp, const Sq64Params *sp, int n)
{
    const BehzParams &b = *bp; const Sq64Params &f = *sp;
    const int sblocks = n / (2 * blockDim.x);
    const size_t poly = blockIdx.x / sblocks;                       // ct*2 + p
    const int s = ((blockIdx.x % sblocks) * blockDim.x + threadIdx.x) * 2;
    const u64 *src = x + poly * (size_t)K * n + s;
    double th[K][2], tl[K][2];
    u32 xm[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < K; i++) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + (size_t)i * n);
        const u64 tr0 = mulmod_shoup(v.x, b.mt_inv_qhat[i], b.mt_inv_qhat_s[i], mods[i].q);                     // baseconverter.cpp:686-696
        const u64 tr1 = mulmod_shoup(v.y, b.mt_inv_qhat[i], b.mt_inv_qhat_s[i], mods[i].q);
        xm[0] += (u32)tr0 * (u32)b.qhat_mod_mt[i]; xm[1] += (u32)tr1 * (u32)b.qhat_mod_mt[i];                   // residue mod m~ = 2^32 (:720-741)
        th[i][0] = (double)(u32)(tr0 >> 32); tl[i][0] = (double)(u32)tr0;
        th[i][1] = (double)(u32)(tr1 >> 32); tl[i][1] = (double)(u32)tr1;
    }
    // r = -(x_m~ q^-1) mod m~ in [0, m~): mont_rq :604-612 (not centred in SEAL 2.3.1)
    const double r0 = (double)(0u - xm[0] * (u32)b.inv_q_mod_mt), r1 = (double)(0u - xm[1] * (u32)b.inv_q_mod_mt);
    double *dst = out + poly * (size_t)KF * n + s;
#pragma unroll
    for (int j = 0; j < KF; j++) {
        // (sum_i tr_i (q/q_i) + q r) m~^-1 mod p_j (:698-718, 614-619), m~^-1 folded into the constants: 2K + 1 products below 0.875 p each, one reduction
        const double p = f.m[j].p;
        double a0 = f64_mulmod_const(r0, f.lift_r[j][0], f.lift_r[j][1], p), a1 = f64_mulmod_const(r1, f.lift_r[j][0], f.lift_r[j][1], p);
#pragma unroll
        for (int i = 0; i < K; i++) { a0 += mul_split(th[i][0], tl[i][0], f.lift_c[j][i], p); a1 += mul_split(th[i][1], tl[i][1], f.lift_c[j][i], p); }
        *reinterpret_cast<double2 *>(dst + (size_t)j * n) = double2{f64_reduce(a0, f.m[j]), f64_reduce(a1, f.m[j])};
    }
}

// rows: [count * 2 * kf][n] doubles, transformed in place under p_(row % kf); results reduced
template <int RB>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024) sq64_fwd_kernel(double *rows, const double *Wf, const Sq64Params *sp, int n, int logn, int kf)
{
    extern __shared__ double smd[];
    const int j = blockIdx.x % kf;
    const F64Mod md = sp->m[j];
    double *row = rows + (size_t)blockIdx.x * n;
    for (int s = 2 * threadIdx.x; s < n; s += 2 * blockDim.x) { const d2 v = *reinterpret_cast<const d2 *>(row + s); sm_store_pair<RB>(smd, s, v.x, v.y); }
    __syncthreads();
    ntt_row_passes_f64<false, RB>(smd, Wf + (size_t)j * n, n, logn, md);
    f64_drain<false, RB, 4>(smd, Wf + (size_t)j * n, n, logn, md, [&](int s, d2 v) { *reinterpret_cast<d2 *>(row + s) = d2{f64_reduce(v.x, md),
        f64_reduce(v.y, md)}; });
}

// tr: [count][2][K][n] = x m~ (q/q_i)^-1 mod q_i in coefficient form (k_ntt_ct_inv_scaled)  ->  out [count][2][kf][n]: the lifted polynomial under p_j,
// transformed, reduced. One workgroup per (polynomial, prime); the kf workgroups of a polynomial sit on one XCD (xcd_group) and read the same k rows.  The lift
// is sq64_lift_kernel's (baseconverter.cpp:663-742, 581-622) for ONE target prime, computed while the row is staged into the image
template <int K, int RB>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024, RB == 3 ? 8 : 4) sq64_liftfwd_kernel(const u64 *tr, double *out, const double *Wf,
    const BehzParams *bp, const Sq64Params *sp,
                                                                                           int n, int logn, int kf, size_t polys)
{
    extern __shared__ double smd[];
    size_t poly; unsigned j;
    if (!xcd_group(blockIdx.x, (unsigned)kf, polys, poly, j)) return;
    const BehzParams &b = *bp; const Sq64Params &f = *sp;
    const F64Mod md = f.m[j];
    const u64 *src = tr + poly * (size_t)K * n;
    const u32 iq = (u32)b.inv_q_mod_mt;
    for (int s = 2 * threadIdx.x; s < n; s += 2 * blockDim.x) {
        u32 xm0 = 0, xm1 = 0;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < K; i++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + (size_t)i * n + s);
            xm0 += (u32)v.x * (u32)b.qhat_mod_mt[i]; xm1 += (u32)v.y * (u32)b.qhat_mod_mt[i];                        // residue mod m~ = 2^32 (:720-741)
            a0 += mul_split((double)(u32)(v.x >> 32), (double)(u32)v.x, f.lift_c[j][i], md.p);
            a1 += mul_split((double)(u32)(v.y >> 32), (double)(u32)v.y, f.lift_c[j][i], md.p);
        }
        // r = -(x_m~ q^-1) mod m~ in [0, m~): mont_rq :604-612 (not centred in SEAL 2.3.1); (sum_i tr_i (q/q_i) + q r) m~^-1 mod p_j: 2K + 1 products below
        // 0.875 p each
        a0 += f64_mulmod_const((double)(0u - xm0 * iq), f.lift_r[j][0], f.lift_r[j][1], md.p);
        a1 += f64_mulmod_const((double)(0u - xm1 * iq), f.lift_r[j][0], f.lift_r[j][1], md.p);
        sm_store_pair<RB>(smd, s, f64_reduce(a0, md), f64_reduce(a1, md));
    }
    __syncthreads();
    ntt_row_passes_f64<false, RB>(smd, Wf + (size_t)j * n, n, logn, md);
    double *row = out + (poly * kf + j) * (size_t)n;
    f64_drain<false, RB, 4>(smd, Wf + (size_t)j * n, n, logn, md, [&](int s, d2 v) { *reinterpret_cast<d2 *>(row + s) = d2{f64_reduce(v.x, md),
        f64_reduce(v.y, md)}; });
}

// in: [count][2][kf][n] transformed rows (a, b)  ->  out: [count][3][kf][n]: n (a^2, 2ab, b^2) in coefficient form, reduced.  One workgroup per (ciphertext,
// prime): the three products share the two source rows, each of which is read ONCE and waits in registers (NPT = points per thread) while the transform in
// front of its second use runs -- round 3 read them again, and by then the L2 had been swept by the other workgroups' rows (20 row reads per ciphertext for 10
// rows at (8192, 3))
template <int RB, int NPT>
__global__ void __launch_bounds__(RB == 5 ? 512 : 1024, 4) sq64_inv_kernel(const double *in, double *out, const double *Wi, const Sq64Params *sp, int n,
    int logn, int kf)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / kf; const int j = blockIdx.x % kf;
    const F64Mod md = sp->m[j];
    const int tid = threadIdx.x, nt = blockDim.x;
    const double *a = in + ((ct * 2 + 0) * kf + j) * (size_t)n, *b = in + ((ct * 2 + 1) * kf + j) * (size_t)n;
    const double *W = Wi + (size_t)j * n;
    auto put = [&](int s, double x, double y) { const d2 v = f64_stage_in<true, RB>(d2{x, y}, W, n, logn, s, md); sm_store_pair<RB>(smd, s, v.x, v.y); };
    double r[NPT];
    auto transform_store = [&](int o) {
        __syncthreads();
        ntt_row_passes_f64<true, RB>(smd, W, n, logn, md, false);           // (the image holds reduced products / one fused stage of them: below 1.75 p)
        double *dst = out + ((ct * 3 + o) * kf + j) * (size_t)n;
        f64_drain<true, RB, NPT / 2>(smd, W, n, logn, md, [&](int s, d2 v) { *reinterpret_cast<d2 *>(dst + s) = d2{f64_reduce(v.x, md), f64_reduce(v.y,
            md)}; });
        __syncthreads();
    };
    // a^2 (|.| < 0.875 p: the first pass reduces on load); a stays in r
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) {
        const int s = 2 * (tid + u * nt);
        if (s < n) { const d2 v = *reinterpret_cast<const d2 *>(a + s); r[2 * u] = v.x; r[2 * u + 1] = v.y;
            put(s, f64_mulmod(v.x, v.x, md), f64_mulmod(v.y, v.y, md)); }
    }
    transform_store(0);
    // 2ab; b replaces a in r
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) {
        const int s = 2 * (tid + u * nt);
        if (s < n) {
            const d2 v = *reinterpret_cast<const d2 *>(b + s);
            put(s, 2.0 * f64_mulmod(r[2 * u], v.x, md), 2.0 * f64_mulmod(r[2 * u + 1], v.y, md));
            r[2 * u] = v.x; r[2 * u + 1] = v.y;
        }
    }
    transform_store(1);
    // b^2
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) {
        const int s = 2 * (tid + u * nt);
        if (s < n) put(s, f64_mulmod(r[2 * u], r[2 * u], md), f64_mulmod(r[2 * u + 1], r[2 * u + 1], md));
    }
    transform_store(2);
}

// ---- the two row kernels with ONE workgroup barrier per transform (round 5; ntt_f64.h, wave-local passes): n = 8192 (CS = 3) or 16384 (CS = 4), n / 16
// threads ---- sq64_liftfwd_kernel: the lifted values are computed straight into the cross layout (the k source rows are read in it: 16-byte pairs for CS = 3),
// the cross pass runs on them in registers, the wave drains its own block.
template <int K, int CS>
__global__ void __launch_bounds__(CS == 3 ? 512 : 1024, 4) sq64_liftfwd_wave_kernel(const u64 *tr, double *out, const double *Wf, const BehzParams *bp,
    const Sq64Params *sp,
                                                                                int n, int kf, size_t polys)
{
    extern __shared__ double smd[];
    size_t poly; unsigned j;
    if (!xcd_group(blockIdx.x, (unsigned)kf, polys, poly, j)) return;
    const BehzParams &b = *bp; const Sq64Params &f = *sp;
    const F64Mod md = f.m[j];
    const u64 *src = tr + poly * (size_t)K * n;
    const u32 iq = (u32)b.inv_q_mod_mt;
    // (sum_i tr_i (q/q_i) + q r) m~^-1 mod p_j of one point from its K residues (sq64_liftfwd_kernel; baseconverter.cpp:663-742, 581-622)
    double v[16];
    if constexpr (CS == 3) {
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int s = f64_cross_point<3>(2 * c);
            u32 xm0 = 0, xm1 = 0; double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int i = 0; i < K; i++) {
                const ulonglong2 x = *reinterpret_cast<const ulonglong2 *>(src + (size_t)i * n + s);
                xm0 += (u32)x.x * (u32)b.qhat_mod_mt[i]; xm1 += (u32)x.y * (u32)b.qhat_mod_mt[i];
                a0 += mul_split((double)(u32)(x.x >> 32), (double)(u32)x.x, f.lift_c[j][i], md.p);
                a1 += mul_split((double)(u32)(x.y >> 32), (double)(u32)x.y, f.lift_c[j][i], md.p);
            }
            a0 += f64_mulmod_const((double)(0u - xm0 * iq), f.lift_r[j][0], f.lift_r[j][1], md.p);
            a1 += f64_mulmod_const((double)(0u - xm1 * iq), f.lift_r[j][0], f.lift_r[j][1], md.p);
            v[2 * c] = f64_reduce(a0, md); v[2 * c + 1] = f64_reduce(a1, md);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int s = f64_cross_point<4>(c);
            u32 xm0 = 0; double a0 = 0.0;
#pragma unroll
            for (int i = 0; i < K; i++) {
                const u64 x = src[(size_t)i * n + s];
                xm0 += (u32)x * (u32)b.qhat_mod_mt[i];
                a0 += mul_split((double)(u32)(x >> 32), (double)(u32)x, f.lift_c[j][i], md.p);
            }
            a0 += f64_mulmod_const((double)(0u - xm0 * iq), f.lift_r[j][0], f.lift_r[j][1], md.p);
            v[c] = f64_reduce(a0, md);
        }
    }
    const double *W = Wf + (size_t)j * n;
    f64_wave_forward<CS, 3>(smd, W, n, md, [&](int q) { return v[q]; });
    double *row = out + (poly * kf + j) * (size_t)n;
    f64_local_drain<3>(smd, W, n, md, [&](int s, d2 pr) { *reinterpret_cast<d2 *>(row + s) = d2{f64_reduce(pr.x, md), f64_reduce(pr.y, md)}; });
}
// sq64_inv_kernel: a / b are held in the block-local layout the fills read them in; every product's transform ends in the cross pass and leaves from registers.
template <int CS>
__global__ void __launch_bounds__(CS == 2 ? 256 : CS == 3 ? 512 : 1024, 4) sq64_inv_wave_kernel(const double *in, double *out, const double *Wi, const Sq64Params *sp, int n,
    int kf)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / kf; const int j = blockIdx.x % kf;
    const F64Mod md = sp->m[j];
    const double *a = in + ((ct * 2 + 0) * kf + j) * (size_t)n, *b = in + ((ct * 2 + 1) * kf + j) * (size_t)n;
    const double *W = Wi + (size_t)j * n;
    double r[16], res[16];
    auto finish = [&](int o) {
        f64_wave_inverse<CS, 3>(smd, W, n, md, [&](int q, double x) { res[q] = f64_reduce(x, md); });
        f64_cross_store<CS>(out + ((ct * 3 + o) * kf + j) * (size_t)n, res);
        __syncthreads();                                      // the image is filled again
    };
    // a^2; a stays in r
    f64_local_fill<3>(smd, W, n, md, [&](int u, int s) {
        const d2 v = *reinterpret_cast<const d2 *>(a + s); r[2 * u] = v.x; r[2 * u + 1] = v.y;
        return d2{f64_mulmod(v.x, v.x, md), f64_mulmod(v.y, v.y, md)};
    });
    finish(0);
    // 2ab; b replaces a in r
    f64_local_fill<3>(smd, W, n, md, [&](int u, int s) {
        const d2 v = *reinterpret_cast<const d2 *>(b + s);
        const d2 pr{2.0 * f64_mulmod(r[2 * u], v.x, md), 2.0 * f64_mulmod(r[2 * u + 1], v.y, md)};
        r[2 * u] = v.x; r[2 * u + 1] = v.y;
        return pr;
    });
    finish(1);
    // b^2
    f64_local_fill<3>(smd, W, n, md, [&](int u, int) { return d2{f64_mulmod(r[2 * u], r[2 * u], md), f64_mulmod(r[2 * u + 1], r[2 * u + 1], md)}; });
    finish(2);
}

// dq: [count][3][K][n] u64 (coefficient form over q, scaled), db: [count][3][KF][n] doubles (n times the coefficient, reduced) -> y3: [count][3][K][n]. (One
// coefficient per thread: two, as in the lift kernel, cost more in occupancy than the 16-byte accesses returned -- 0.77 against 0.64 us per ciphertext at
// (8192, 3))
template <int K, int KF>
__global__ void __launch_bounds__(256) sq64_floor_kernel(const u64 *dq, const double *db, u64 *y3, const ModParams *mods, const BehzParams *bp,
    const Sq64Params *sp, int n, int premul_c2, int dq_scaled)
{
    const BehzParams &b = *bp; const Sq64Params &f = *sp;
    constexpr int KB = KF - 1;
    const int sblocks = n / blockDim.x;
    const size_t poly = blockIdx.x / sblocks;                       // ct*3 + p
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const u64 *xq = dq + poly * (size_t)K * n + s;
    const double *xb = db + poly * (size_t)KF * n + s;
    double th[K], tl[K];
    // x t (evaluator.cpp:856-871) and the (q/q_i)^-1 of fastbconv (:413-423) are one constant
#pragma unroll
    for (int i = 0; i < K; i++) {
        // (dq_scaled: the inverse transform that made the row closed with this very multiplication -- k_square_intt, round 5)
        const u64 xv = xq[(size_t)i * n], tr = dq_scaled ? xv : mulmod_shoup(xv, b.t_inv_qhat[i], b.t_inv_qhat_s[i], mods[i].q);
        th[i] = (double)(u32)(tr >> 32); tl[i] = (double)(u32)tr;
    }
    // fast_floor (:646-660): (x_p t - fastbconv(x_q t)) q^-1 mod p_j, with q^-1 (and the inverse transform's n^-1) folded into floor_x / floor_c
    double fl[KF];
#pragma unroll
    for (int j = 0; j < KF; j++) {
        const double p = f.m[j].p;
        double a = f64_mulmod_const(xb[(size_t)j * n], f.floor_x[j][0], f.floor_x[j][1], p);
#pragma unroll
        for (int i = 0; i < K; i++) a += mul_split(th[i], tl[i], f.floor_c[j][i], p);                            // :425-445
        fl[j] = f64_reduce(a, f.m[j]);
    }
    // fastbconv_sk (:448-579): z_j = fl_j (B/p_j)^-1 mod p_j in [0, p_j)
    const F64Mod msk = f.m[KB];
    double z[KB], v = 0.0;
#pragma unroll
    for (int j = 0; j < KB; j++) {
        double zz = f64_reduce(f64_mulmod_const(fl[j], f.inv_mhat[j][0], f.inv_mhat[j][1], f.m[j].p), f.m[j]);
        zz = zz < 0.0 ? zz + f.m[j].p : zz;
        z[j] = zz;
        v += f64_mulmod_const(zz, f.mhat_msk[j][0], f.mhat_msk[j][1], msk.p);
    }
    // alpha = (sum_j z_j (B/p_j) - R) / B, from its residue mod m_sk: a small integer (|alpha| <= KB + |R| / B, far inside the centred range), exact in the
    // double
    const double ad = f64_reduce(f64_mulmod_const(f64_reduce(v - fl[KB], msk), f.inv_B_msk[0], f.inv_B_msk[1], msk.p), msk);
    const long long alpha = (long long)ad;
    const bool neg = alpha < 0;
    const u64 am = (u64)(neg ? -alpha : alpha);                      // (up to |R| / B + KB: 37 bits at n = 256, t = 2^41 with two 40-bit moduli)
    const u32 am0 = (u32)am & 0xfffffffu, am1 = (u32)(am >> 28);
    // sum_j z_j (B/p_j) - alpha B mod q_i (:553-569), lazily: z_j (47 bits) and the constants (up to 60) in 28-bit pieces, so that every partial product is ONE
    // v_mad_u64_u32 into one of three 64-bit sums that cannot overflow (z0 y0 < 2^56, z0 y1 + z1 y0 < 2^60 + 2^47, z1 y1 < 2^51; at most 12 terms, |alpha|
    // among them) -- 4 instructions per term instead of the 14 of a 64 x 64 -> 128 product and its carry chain; the sums meet in 128 bits once, before the one
    // reduction
    u32 z0[KB], z1[KB];
#pragma unroll
    for (int j = 0; j < KB; j++) { z1[j] = (u32)(z[j] * 0x1p-28); z0[j] = (u32)(z[j] - (double)z1[j] * 0x1p28); }
    u64 *dst = y3 + poly * (size_t)K * n + s;
#pragma unroll
    for (int i = 0; i < K; i++) {
        const ModParams mi = mods[i];
        u64 P0 = 0, P1 = 0, P2 = 0;
#pragma unroll
        for (int j = 0; j < KB; j++) {
            const u64 y = f.mhat_q[i][j]; const u32 y0 = (u32)y & 0xfffffffu, y1 = (u32)(y >> 28);
            P0 += (u64)z0[j] * y0; P1 += (u64)z0[j] * y1; P1 += (u64)z1[j] * y0; P2 += (u64)z1[j] * y1;
        }
        {                                                                                                                             // - alpha B
            const u64 y = neg ? f.B_q[i] : mi.q - f.B_q[i]; const u32 y0 = (u32)y & 0xfffffffu, y1 = (u32)(y >> 28);
            P0 += (u64)am0 * y0; P1 += (u64)am0 * y1; P1 += (u64)am1 * y0; P2 += (u64)am1 * y1;
        }
        const u64 l1 = P0 + (P1 << 28), t2 = P2 << 56, lo = l1 + t2;
        const u64 hi = (P1 >> 36) + (l1 < P0) + (P2 >> 8) + (lo < t2);
        u64 r = barrett128(lo, hi, mi);
        if (premul_c2 && poly % 3 == 2) r = mulmod_shoup(r, b.inv_qhat[i], b.inv_qhat_s[i], mi.q);
        dst[(size_t)i * n] = r;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------------------------------
static int sq64_threads(const crc_ctx *c, int RB) { int nt = c->n >> RB; if (nt < 64) nt = 64; if (nt > (RB == 5 ? 512 : 1024)) nt = RB == 5 ? 512 : 1024;
    return nt; }
// the wave-local transforms (one workgroup barrier per transform) serve the two rings the bench configurations use; CRC_F64_WAVE=0 keeps the round-4 kernels
// (bit 0: sq64_inv_kernel -- -25 % at both rings --, bit 3: the lifting forward kernel -- +2 % / +31 %: it has no row to hold and the round-4 form runs it at
// eight waves per SIMD; off by default)
static bool sq64_wave_path(const crc_ctx *c, int RB, int bit)
{
    // (n = 4096 since round 6: CS = 2; the lifting forward kernel -- bit 3, off by default -- exists for n = 8192 / 16384 only)
    if (RB != 3 || (c->logn != 12 && c->logn != 13 && c->logn != 14) || (bit == 3 && c->logn == 12)) return false;
    const int sel = c->tune.f64_wave < 0 ? 7 : c->tune.f64_wave;
    return (sel >> bit) & 1;
}
static int sq64_radix(const crc_ctx *c) { const int r = c->tune.f64_radix; return r >= 3 && r <= 5 ? r : 3; }

bool k_square64_supported(const crc_ctx *c)
{
    if (c->sq64.kf < 3 || c->n < 64 || c->n > 16384) return false;
#define HAVE(KV, KFV) if (c->k == KV && c->sq64.kf == KFV) return true;
    CRC_FOR_ALL_K_KF(HAVE)
#undef HAVE
    return false;
}

// threads of the kernel that keeps a row in registers: 16 points per thread (32 at radix 32) -- kernels_relin64.hip relin_inv_crt_kernel
static int sq64_hold_threads(const crc_ctx *c, int RB) { int nt = c->n >> (RB == 5 ? 5 : 4); if (nt < 64) nt = 64;
    if (nt > (RB == 5 ? 512 : 1024)) nt = RB == 5 ? 512 : 1024; return nt; }

// work: QN [2k] | LB [2 kf] | DQ [3k] | DB [3 kf]   (k_square_work_words sizes the rows by max(kb, kf))
int k_square64(crc_ctx *c, const u64 *x, size_t cnt, u64 *y3, u64 *work, hipStream_t st, bool in_ntt, bool premul_c2)
{
    if (cnt == 0) return CRC_OK;
    if (!k_square64_supported(c)) return CRC_ERR_UNSUPPORTED;
    const size_t n = c->n, k = c->k, kf = c->sq64.kf;
    u64 *QN = work, *LBw = QN + cnt * 2 * k * n, *DQ = LBw + cnt * 2 * kf * n, *DBw = DQ + cnt * 3 * k * n;
    double *LB = reinterpret_cast<double *>(LBw), *DB = reinterpret_cast<double *>(DBw);
    const int lthreads = c->n < 512 ? c->n / 2 : 256, lblocks = c->n / (2 * lthreads);         // (the lift kernel: two coefficients per thread)
    int rc;
    const size_t lds = n * 8;
    const int RB = sq64_radix(c), nt = sq64_threads(c, RB);
    const double *Wf = c->d_f64_rp, *Wi = c->d_f64_irp;
    const u64 *xn = QN;
    // (fused up to k = 4: every one of the kf workgroups of a polynomial reads its k source rows, and at k = 8, kf = 11 that costs more than the lift kernel's
    // round trip -- 35.7 against 34.3 us per ciphertext at (16384, 8), 14.6 against 14.75 at (16384, 4), 5.55 against 5.69 at (8192, 3):
    // profiles/r04_square_relin_ab_step1.txt)
    if (in_ntt && (c->tune.sq_fuse < 0 ? c->k <= 4 : c->tune.sq_fuse != 0)) {
        // NTT-resident caller: the coefficient form exists only for the lift, so it is made premultiplied and lifted inside the forward transforms
        // (sq64_liftfwd_kernel)
        if ((rc = k_ntt_ct_inv_scaled(c, x, QN, cnt, 2, c->behz.mt_inv_qhat, c->behz.mt_inv_qhat_s, st))) return rc;
        xn = x;
        bool launched = false;
        const unsigned grid = xcd_grid(cnt * 2, (unsigned)kf);
        if (sq64_wave_path(c, RB, 3)) {
#define LIFTW(KV) if (c->k == KV) { \
            auto kern = c->logn == 13 ? sq64_liftfwd_wave_kernel<KV, 3> : sq64_liftfwd_wave_kernel<KV, 4>; \
            if ((rc = crc_ctx_ensure_lds(c, (const void *)kern, lds))) return rc; \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(c->n / 16), lds, st, QN, LB, Wf, c->d_behz, c->d_sq64, c->n, (int)kf, cnt * 2); launched = true; }
            LIFTW(1) LIFTW(2) LIFTW(3) LIFTW(4)
#undef LIFTW
        }
#define LIFTFWD(KV) if (!launched && c->k == KV) { \
            auto kern = RB == 3 ? sq64_liftfwd_kernel<KV, 3> : RB == 4 ? sq64_liftfwd_kernel<KV, 4> : sq64_liftfwd_kernel<KV, 5>; \
            if ((rc = crc_ctx_ensure_lds(c, (const void *)kern, lds))) return rc; \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), lds, st, QN, LB, Wf, c->d_behz, c->d_sq64, c->n, c->logn, (int)kf, cnt * 2); launched = true; }
        LIFTFWD(1) LIFTFWD(2) LIFTFWD(3) LIFTFWD(4) LIFTFWD(5) LIFTFWD(6) LIFTFWD(7) LIFTFWD(8)
#undef LIFTFWD
        if (!launched) return CRC_ERR_UNSUPPORTED;
        HIPCHK(hipGetLastError());
    } else {
        const u64 *xc = x;
        if (in_ntt) { if ((rc = k_ntt_ct(c, true, x, QN, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc; xc = QN; xn = x; }
        {
            const dim3 grid((unsigned)(cnt * 2 * lblocks)), blk(lthreads);
            bool launched = false;
#define LIFT(KV, KFV) if (c->k == KV && c->sq64.kf == KFV) { hipLaunchKernelGGL((sq64_lift_kernel<KV, KFV>), grid, blk, 0, st, xc, LB, c->d_mods, c->d_behz, c->d_sq64, c->n); launched = true; }
            CRC_FOR_ALL_K_KF(LIFT)
#undef LIFT
            if (!launched) return CRC_ERR_UNSUPPORTED;
            HIPCHK(hipGetLastError());
        }
        if (!in_ntt && (rc = k_ntt_ct(c, false, x, QN, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc;
        auto kern = RB == 3 ? sq64_fwd_kernel<3> : RB == 4 ? sq64_fwd_kernel<4> : sq64_fwd_kernel<5>;
        if ((rc = crc_ctx_ensure_lds(c, (const void *)kern, lds))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * 2 * kf)), dim3(nt), lds, st, LB, Wf, c->d_sq64, c->n, c->logn, (int)kf);
        HIPCHK(hipGetLastError());
    }
    // a^2, 2ab, b^2 over q are formed while the inverse transforms load their rows (kernels.hip); over the fp64 primes in sq64_inv_kernel
    // (x t (q/q_i)^-1, the first thing the floor kernel does to these rows, goes into the transform's closing multiplication where it has one)
    bool dq_scaled = false;
    if ((rc = k_square_intt(c, xn, DQ, cnt, false, st, c->behz.t_inv_qhat, &dq_scaled))) return rc;
    if (sq64_wave_path(c, RB, 0)) {
        auto kern = c->logn == 12 ? sq64_inv_wave_kernel<2> : c->logn == 13 ? sq64_inv_wave_kernel<3> : sq64_inv_wave_kernel<4>;
        if ((rc = crc_ctx_ensure_lds(c, (const void *)kern, lds))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * kf)), dim3(c->n / 16), lds, st, LB, DB, Wi, c->d_sq64, c->n, (int)kf);
        HIPCHK(hipGetLastError());
    } else {
        auto kern = RB == 3 ? sq64_inv_kernel<3, 16> : RB == 4 ? sq64_inv_kernel<4, 16> : sq64_inv_kernel<5, 32>;
        if ((rc = crc_ctx_ensure_lds(c, (const void *)kern, lds))) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * kf)), dim3(sq64_hold_threads(c, RB)), lds, st, LB, DB, Wi, c->d_sq64, c->n, c->logn, (int)kf);
        HIPCHK(hipGetLastError());
    }
    {
        const int threads = c->n < 256 ? c->n : 256;
        const dim3 grid((unsigned)(cnt * 3 * (c->n / threads))), blk(threads);
        bool launched = false;
#define FLOOR(KV, KFV) if (c->k == KV && c->sq64.kf == KFV) { hipLaunchKernelGGL((sq64_floor_kernel<KV, KFV>), grid, blk, 0, st, DQ, DB, y3, c->d_mods, c->d_behz, c->d_sq64, c->n, premul_c2 ? 1 : 0, dq_scaled ? 1 : 0); launched = true; }
        CRC_FOR_ALL_K_KF(FLOOR)
#undef FLOOR
        if (!launched) return CRC_ERR_UNSUPPORTED;
        HIPCHK(hipGetLastError());
    }
    return CRC_OK;
}
