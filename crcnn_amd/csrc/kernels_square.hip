// kernels_square.hip -- BFV ciphertext x ciphertext square (full-RNS "BEHZ") and relinearisation on gfx950.
//
// Pipeline per ciphertext (reference: Evaluator::square evaluator.cpp:702-884, relinearize_one_step :934-1069,
// BaseConverter routines util/baseconverter.cpp:388-742).  Per-coefficient base conversions are lane-per-coefficient
// kernels (coalesced over s, tables in scalar registers); everything between them is the row NTT of kernels.hip.
//   lift     : q -> Bsk U {m~} (fastbconv_mtilde) and small Montgomery reduction (mont_rq)        [sq_lift_kernel]
//   NTT      : 2 polys in q and in Bsk                                                             [ntt_rows_kernel]
//   products : c0^2, 2 c0 c1, c1^2 in both bases                                                   [sq_dyadic_kernel]
//   INTT     : 3 polys in both bases
//   floor    : x t, fast_floor (q U Bsk -> Bsk), fastbconv_sk (Bsk -> q)                           [sq_floor_kernel]
//   relin    : digits of c2 (q/q_i)^-1, NTT per digit and modulus, 128-bit MAC against the keys,
//              Barrett, INTT, add into (c0, c1)                                  [relin digit prologue + relin_mac_kernel]
#include "kernels.h"

// 128-bit accumulate helper
struct RelinTab { long long keyoff[48]; unsigned char dig_i[48]; unsigned char dig_shift[48]; };   // passed by value (kernarg)
struct acc128 { u64 lo, hi; };
__device__ __forceinline__ void acc_mad(acc128 &a, u64 x, u64 y)
{
    u64 pl, ph; mul64wide(x, y, pl, ph);
    const u64 nl = a.lo + pl; a.hi += ph + (nl < pl); a.lo = nl;
}

// x: [count][2][k][n] (coefficient form, base q)  ->  out: [count][2][kb][n] in Bsk
__global__ void __launch_bounds__(256) sq_lift_kernel(const u64 *x, u64 *out, const ModParams *mods, const BehzParams *bp, int n)
{
    const BehzParams &b = *bp;
    const int k = b.k, kb = b.kb;
    const int sblocks = n / blockDim.x;
    const size_t poly = blockIdx.x / sblocks;                       // ct*2 + p
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const u64 *src = x + poly * (size_t)k * n + s;
    u64 tr[CRC_MAXK];
    for (int i = 0; i < k; i++) tr[i] = mulmod_shoup(src[(size_t)i * n], b.mt_inv_qhat[i], b.mt_inv_qhat_s[i], mods[i].q);      // baseconverter.cpp:686-696
    // residue mod m~ = 2^32 (:720-741) and r = -(x_m~ q^-1) mod m~ (mont_rq :604-612)
    u64 xm = 0;
    for (int i = 0; i < k; i++) xm += tr[i] * b.qhat_mod_mt[i];
    xm &= 0xffffffffULL;
    u64 r = (xm * b.inv_q_mod_mt) & 0xffffffffULL;
    r = (0 - r) & 0xffffffffULL;
    u64 *dst = out + poly * (size_t)kb * n + s;
    for (int j = 0; j < kb; j++) {
        const ModParams mj = mods[k + j];
        acc128 a{0, 0};
        for (int i = 0; i < k; i++) acc_mad(a, tr[i], b.qhat_mod_bsk[j][i]);                        // :698-718
        acc_mad(a, b.q_mod_bsk[j], r);                                                              // mont_rq :614-618 (sum stays < 2^128)
        const u64 v = barrett128(a.lo, a.hi, mj);
        dst[(size_t)j * n] = mulmod_shoup(v, b.inv_mt_mod_bsk[j], b.inv_mt_mod_bsk_s[j], mj.q);     // :619
    }
}

// in: [count][2][K][n] NTT form, out: [count][3][K][n] = (a^2, 2ab, b^2); rows use moduli mod_base + (row % K)
__global__ void __launch_bounds__(256) sq_dyadic_kernel(const u64 *in, u64 *out, const ModParams *mods, int n, int K, int mod_base)
{
    const size_t r = blockIdx.x;                 // ct*K + j
    const size_t ct = r / K; const int j = (int)(r % K);
    const ModParams m = mods[mod_base + j];
    const u64 *a = in + ((ct * 2 + 0) * K + j) * (size_t)n, *b = in + ((ct * 2 + 1) * K + j) * (size_t)n;
    u64 *d0 = out + ((ct * 3 + 0) * K + j) * (size_t)n, *d1 = out + ((ct * 3 + 1) * K + j) * (size_t)n, *d2 = out + ((ct * 3 + 2) * K + j) * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const u64 av = a[s], bv = b[s];
        d0[s] = mulmod(av, av, m);
        d2[s] = mulmod(bv, bv, m);
        const u64 x = mulmod(av, bv, m);
        d1[s] = addmod(x, x, m.q);
    }
}

// dq: [count][3][k][n], db: [count][3][kb][n] (coefficient form, before the multiplication by t) -> y3: [count][3][k][n]
__global__ void __launch_bounds__(256) sq_floor_kernel(const u64 *dq, const u64 *db, u64 *y3, const ModParams *mods, const BehzParams *bp, int n)
{
    const BehzParams &b = *bp;
    const int k = b.k, kb = b.kb, ka = b.ka;
    const int sblocks = n / blockDim.x;
    const size_t poly = blockIdx.x / sblocks;                       // ct*3 + p
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const u64 *xq = dq + poly * (size_t)k * n + s, *xb = db + poly * (size_t)kb * n + s;
    u64 tr[CRC_MAXK];
    // x t (evaluator.cpp:856-871) and the (q/q_i)^-1 of fastbconv (:413-423) are one constant
    for (int i = 0; i < k; i++) tr[i] = mulmod_shoup(xq[(size_t)i * n], b.t_inv_qhat[i], b.t_inv_qhat_s[i], mods[i].q);
    u64 fl[CRC_MAXB];
    for (int j = 0; j < kb; j++) {
        const ModParams mj = mods[k + j];
        acc128 a{0, 0};
        for (int i = 0; i < k; i++) acc_mad(a, tr[i], b.qhat_mod_bsk[j][i]);                        // :425-445
        const u64 conv = barrett128(a.lo, a.hi, mj);
        const u64 xv = mulmod_shoup(xb[(size_t)j * n], b.t_mod_bsk[j], b.t_mod_bsk_s[j], mj.q);
        fl[j] = mulmod_shoup(xv + mj.q - conv, b.inv_q_mod_bsk[j], b.inv_q_mod_bsk_s[j], mj.q);    // fast_floor :646-660
    }
    // fastbconv_sk :448-579
    u64 z[CRC_MAXB];
    for (int j = 0; j < ka; j++) z[j] = mulmod_shoup(fl[j], b.inv_mhat[j], b.inv_mhat_s[j], mods[k + j].q);
    const ModParams msk = mods[k + ka];
    acc128 as{0, 0};
    for (int j = 0; j < ka; j++) acc_mad(as, z[j], b.mhat_mod_msk[j]);
    const u64 vsk = barrett128(as.lo, as.hi, msk);
    const u64 alpha = mulmod_shoup(vsk + (b.m_sk - fl[ka]), b.inv_M_mod_msk, b.inv_M_mod_msk_s, msk.q);
    const bool neg = alpha > (b.m_sk >> 1);
    u64 *dst = y3 + poly * (size_t)k * n + s;
    for (int i = 0; i < k; i++) {
        const ModParams mi = mods[i];
        acc128 a{0, 0};
        for (int j = 0; j < ka; j++) acc_mad(a, z[j], b.mhat_mod_q[i][j]);
        if (neg) acc_mad(a, b.M_mod_q[i], b.m_sk - alpha);                                          // :553-559
        else acc_mad(a, mi.q - b.M_mod_q[i], alpha);                                                // :561-569
        dst[(size_t)i * n] = barrett128(a.lo, a.hi, mi);
    }
}

// E: [count][D][k][n] NTT-form digits; keys: evk blob; out: [count][2][k][n] NTT form, Barrett-reduced
__global__ void __launch_bounds__(256) relin_mac_kernel(const u64 *E, const u64 *evk, u64 *out, const ModParams *mods, int n, int k, int D,
                                                        RelinTab tab)
{
    const int sblocks = n / blockDim.x;
    const size_t cj = blockIdx.x / sblocks;                          // ct*k + j
    const size_t ct = cj / k; const int j = (int)(cj % k);
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const ModParams m = mods[j];
    acc128 a0{0, 0}, a1{0, 0};
    for (int g = 0; g < D; g++) {
        const u64 e = E[((ct * D + g) * k + j) * (size_t)n + s];
        const u64 *key = evk + tab.keyoff[g] + (size_t)j * n + s;
        acc_mad(a0, e, key[0]);
        acc_mad(a1, e, key[(size_t)k * n]);
    }
    out[((ct * 2 + 0) * k + j) * (size_t)n + s] = barrett128(a0.lo, a0.hi, m);
    out[((ct * 2 + 1) * k + j) * (size_t)n + s] = barrett128(a1.lo, a1.hi, m);
}

// c2 (third poly of x3 [count][3][k][n]) -> digit polynomials [count][D][n] (not yet spread over the k target moduli)
__global__ void __launch_bounds__(256) relin_digits_kernel(const u64 *x3, u64 *dig, const ModParams *mods, const BehzParams *bp, int n, int k,
                                                           int D, RelinTab tab, int dbc)
{
    const size_t r = blockIdx.x;                 // ct*D + g
    const size_t ct = r / D; const int g = (int)(r % D);
    const int i = tab.dig_i[g], sh = tab.dig_shift[g];
    const ModParams m = mods[i];
    const u64 inv = bp->inv_qhat[i];
    const u64 mask = (1ULL << dbc) - 1;
    const u64 *src = x3 + ((ct * 3 + 2) * k + i) * (size_t)n;
    u64 *dst = dig + r * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const u64 e = mulmod(src[s], inv, m);                        // evaluator.cpp:984-985
        dst[s] = (e >> sh) & mask;                                   // :997-1001
    }
}

size_t k_square_work_words(const crc_ctx *c, size_t cnt)
{
    const size_t n = c->n, k = c->k, kb = c->kb;
    // QN[2k] BS[2kb] DQ[3k] DB[3kb]
    return cnt * n * (2 * k + 2 * kb + 3 * k + 3 * kb);
}
size_t k_relin_work_words(const crc_ctx *c, size_t cnt, int dbc)
{
    const size_t n = c->n, k = c->k;
    size_t D = 0; for (int i = 0; i < c->k; i++) D += evk_digits(c->q[i], dbc);
    // DIG[D] E[D*k] R[2k] + tables
    return cnt * n * (D + D * k + 2 * k);
}

int k_square(crc_ctx *c, const u64 *x, size_t cnt, u64 *y3, u64 *work, hipStream_t st, bool in_ntt)
{
    if (cnt == 0) return CRC_OK;
    const size_t n = c->n, k = c->k, kb = c->kb;
    u64 *QN = work, *BS = QN + cnt * 2 * k * n, *DQ = BS + cnt * 2 * kb * n, *DB = DQ + cnt * 3 * k * n;
    const int threads = c->n < 256 ? c->n : 256, sblocks = c->n / threads;
    int rc;
    // the square needs x in both forms: coefficients for the base extension, NTT values (base q) for the products.  An NTT-resident
    // caller hands over the latter, so one inverse transform replaces the forward one (and the caller's own conversion disappears)
    const u64 *xc = x, *xn = QN;
    if (in_ntt) { if ((rc = k_ntt_ct(c, true, x, QN, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc; xc = QN; xn = x; }
    hipLaunchKernelGGL(sq_lift_kernel, dim3((unsigned)(cnt * 2 * sblocks)), dim3(threads), 0, st, xc, BS, c->d_mods, c->d_behz, c->n);
    HIPCHK(hipGetLastError());
    if (!in_ntt && (rc = k_ntt_ct(c, false, x, QN, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc;
    if ((rc = k_ntt_ct(c, false, BS, BS, cnt, 2, true, st, nullptr, 0, 0, 0))) return rc;
    hipLaunchKernelGGL(sq_dyadic_kernel, dim3((unsigned)(cnt * k)), dim3(256), 0, st, xn, DQ, c->d_mods, c->n, (int)k, 0);
    hipLaunchKernelGGL(sq_dyadic_kernel, dim3((unsigned)(cnt * kb)), dim3(256), 0, st, BS, DB, c->d_mods, c->n, (int)kb, (int)k);
    HIPCHK(hipGetLastError());
    if ((rc = k_ntt_ct(c, true, DQ, DQ, cnt, 3, false, st, nullptr, 0, 0, 0))) return rc;
    if ((rc = k_ntt_ct(c, true, DB, DB, cnt, 3, true, st, nullptr, 0, 0, 0))) return rc;
    hipLaunchKernelGGL(sq_floor_kernel, dim3((unsigned)(cnt * 3 * sblocks)), dim3(threads), 0, st, DQ, DB, y3, c->d_mods, c->d_behz, c->n);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

int k_relinearize(crc_ctx *c, const u64 *x3, size_t cnt, const u64 *evk, int dbc, u64 *y, u64 *work, hipStream_t st, bool out_ntt)
{
    if (cnt == 0) return CRC_OK;
    if (dbc < 1 || dbc > 60) return CRC_ERR_INVALID_ARGUMENT;
    const size_t n = c->n, k = c->k;
    RelinTab tab{};
    int D = 0; long long off = 0;
    for (int i = 0; i < c->k; i++) {
        int L = evk_digits(c->q[i], dbc);
        for (int d = 0; d < L; d++) { if (D >= 48) return CRC_ERR_UNSUPPORTED; tab.dig_i[D] = (unsigned char)i; tab.dig_shift[D] = (unsigned char)(d * dbc); tab.keyoff[D] = off + (long long)(2 * d) * k * n; D++; }
        off += (long long)2 * L * k * n;
    }
    u64 *DIG = work, *E = DIG + cnt * D * n, *R = E + cnt * D * k * n;
    hipLaunchKernelGGL(relin_digits_kernel, dim3((unsigned)(cnt * D)), dim3(256), 0, st, x3, DIG, c->d_mods, c->d_behz, c->n, c->k, D, tab, dbc);
    HIPCHK(hipGetLastError());
    // forward NTT of every digit polynomial under every q_j: rows [cnt*D][k][n], source row = digit polynomial (plain prologue without lift)
    int rc;
    if ((rc = k_spread_ntt(c, DIG, cnt * D, E, st))) return rc;
    const int threads = c->n < 256 ? c->n : 256, sblocks = c->n / threads;
    hipLaunchKernelGGL(relin_mac_kernel, dim3((unsigned)(cnt * k * sblocks)), dim3(threads), 0, st, E, evk, R, c->d_mods, c->n, c->k, D, tab);
    HIPCHK(hipGetLastError());
    // INTT and add (c0, c1) of the size-3 input   (evaluator.cpp:1041-1068); for an NTT-form result the sum is formed on the other
    // side of the (linear) transform: NTT(c0, c1) + R
    if (out_ntt) return k_ntt_ct_head_add(c, x3, 3, y, cnt, R, st);
    return k_ntt_ct_addct(c, R, y, cnt, x3, 3, st);
}
