// kernels_square.hip -- BFV ciphertext x ciphertext square (full-RNS "BEHZ") and relinearisation on gfx950.
//
// Pipeline per ciphertext (reference: Evaluator::square evaluator.cpp:702-884, relinearize_one_step :934-1069,
// BaseConverter routines util/baseconverter.cpp:388-742).  Per-coefficient base conversions are lane-per-coefficient
// kernels (coalesced over s, tables in scalar registers); everything between them is the row NTT of kernels.hip.
//   lift     : q -> Bsk U {m~} (fastbconv_mtilde) and small Montgomery reduction (mont_rq)        [sq_lift_kernel]
//   NTT      : 2 polys in q and in Bsk                                                             [ntt_rows_kernel]
//   products : c0^2, 2 c0 c1, c1^2 in both bases, formed while the inverse transform loads its row [ntt_rows_kernel, prologue 4]
//   INTT     : 3 polys in both bases
//   floor    : x t, fast_floor (q U Bsk -> Bsk), fastbconv_sk (Bsk -> q); the third polynomial leaves
//              premultiplied by (q/q_i)^-1 for the digit decomposition                             [sq_floor_kernel]
//   relin    : digit g of c2 (q/q_i)^-1 cut out while the forward transform loads its row          [ntt_rows_kernel, prologue 3]
//              lazy 28-bit-limb MAC against the (packed) keys, one folding reduction               [relin_mac28_kernel]
//              INTT + (c0, c1), or NTT(c0, c1) + that for an NTT-resident result
// Row transfers through HBM per ciphertext at (k, kb, D) = (3, 4, 12): 2k+2k | 2k+2kb | 2kb+2kb | 4k+3k, 4kb+3kb | 3k+3kb+3k | D+Dk | Dk+2k | 4k+2k = 198
// (round 1: 296 -- the dyadic products, the digit polynomials and their re-reads are gone)
#include "kernels.h"

// every (k, ka) a context can have: k = 1..CRC_MAXK coefficient moduli, auxiliary base of k or k+1 primes (baseconverter.cpp:47-56)
#define CRC_FOR_ALL_K_KA(X) X(1, 1) X(1, 2) X(2, 2) X(2, 3) X(3, 3) X(3, 4) X(4, 4) X(4, 5) X(5, 5) X(5, 6) X(6, 6) X(6, 7) X(7, 7) X(7, 8) X(8, 8) X(8, 9)

// 128-bit accumulate helper
struct RelinTab { long long keyoff[48]; unsigned char dig_i[48]; unsigned char dig_shift[48]; };   // passed by value (kernarg)
struct acc128 { u64 lo, hi; };
__device__ __forceinline__ void acc_mad(acc128 &a, u64 x, u64 y)
{
    u64 pl, ph; mul64wide(x, y, pl, ph);
    const u64 nl = a.lo + pl; a.hi += ph + (nl < pl); a.lo = nl;
}

// x: [count][2][k][n] (coefficient form, base q)  ->  out: [count][2][kb][n] in Bsk
// (K, KB compile-time: the loops unroll completely, tr[] / constants live in registers / SGPRs -- with run-time trip counts the kernel was a chain of
// dependent scalar loads: profiles/r02_square_relin.txt)
template <int K, int KB>
__global__ void __launch_bounds__(256) sq_lift_kernel(const u64 *x, u64 *out, const ModParams *mods, const BehzParams *bp, int n)
{
    const BehzParams &b = *bp;
    constexpr int k = K, kb = KB;
    const int sblocks = n / blockDim.x;
    const size_t poly = blockIdx.x / sblocks;                       // ct*2 + p
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const u64 *src = x + poly * (size_t)k * n + s;
    u64 tr[K];
#pragma unroll
    for (int i = 0; i < k; i++) tr[i] = mulmod_shoup(src[(size_t)i * n], b.mt_inv_qhat[i], b.mt_inv_qhat_s[i], mods[i].q);      // baseconverter.cpp:686-696
    // residue mod m~ = 2^32 (:720-741) and r = -(x_m~ q^-1) mod m~ (mont_rq :604-612): 32-bit arithmetic
    u32 xm = 0;
#pragma unroll
    for (int i = 0; i < k; i++) xm += (u32)tr[i] * (u32)b.qhat_mod_mt[i];
    const u32 r = 0u - xm * (u32)b.inv_q_mod_mt;
    u64 *dst = out + poly * (size_t)kb * n + s;
#pragma unroll
    for (int j = 0; j < kb; j++) {
        // (sum_i tr_i (q/q_i) + q r) m~^-1  mod Bsk_j  (:698-718, 614-619) with m~^-1 folded into the constants: one lazy sum (< 2^119), one reduction
        acc128 a{0, 0};
#pragma unroll
        for (int i = 0; i < k; i++) acc_mad(a, tr[i], b.lift_c[j][i]);
        acc_mad(a, b.lift_r[j], (u64)r);
        dst[(size_t)j * n] = barrett128(a.lo, a.hi, mods[k + j]);
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
// premul_c2: the third polynomial is stored as c2 (q/q_i)^-1 mod q_i, the form relinearisation cuts its digits from (evaluator.cpp:984-985)
template <int K, int KA>
__global__ void __launch_bounds__(256) sq_floor_kernel(const u64 *dq, const u64 *db, u64 *y3, const ModParams *mods, const BehzParams *bp, int n, int premul_c2)
{
    const BehzParams &b = *bp;
    constexpr int k = K, ka = KA, kb = KA + 1;
    const int sblocks = n / blockDim.x;
    const size_t poly = blockIdx.x / sblocks;                       // ct*3 + p
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const u64 *xq = dq + poly * (size_t)k * n + s, *xb = db + poly * (size_t)kb * n + s;
    u64 tr[K];
    // x t (evaluator.cpp:856-871) and the (q/q_i)^-1 of fastbconv (:413-423) are one constant
#pragma unroll
    for (int i = 0; i < k; i++) tr[i] = mulmod_shoup(xq[(size_t)i * n], b.t_inv_qhat[i], b.t_inv_qhat_s[i], mods[i].q);
    u64 fl[KA + 1];
#pragma unroll
    for (int j = 0; j < kb; j++) {
        // fast_floor (:646-660): (x_bsk t - fastbconv(x_q t)) q^-1 mod Bsk_j with q^-1 folded into the constants (floor_x = t q^-1, floor_c = -(q/q_i) q^-1):
        // one lazy sum of k+1 products (< 2^124), one reduction -- the same residue as the reference's three separate steps
        acc128 a{0, 0};
        acc_mad(a, xb[(size_t)j * n], b.floor_x[j]);
#pragma unroll
        for (int i = 0; i < k; i++) acc_mad(a, tr[i], b.floor_c[j][i]);                             // :425-445
        fl[j] = barrett128(a.lo, a.hi, mods[k + j]);
    }
    // fastbconv_sk :448-579
    u64 z[KA];
#pragma unroll
    for (int j = 0; j < ka; j++) z[j] = mulmod_shoup(fl[j], b.inv_mhat[j], b.inv_mhat_s[j], mods[k + j].q);
    const ModParams msk = mods[k + ka];
    acc128 as{0, 0};
#pragma unroll
    for (int j = 0; j < ka; j++) acc_mad(as, z[j], b.mhat_mod_msk[j]);
    const u64 vsk = barrett128(as.lo, as.hi, msk);
    const u64 alpha = mulmod_shoup(vsk + (b.m_sk - fl[ka]), b.inv_M_mod_msk, b.inv_M_mod_msk_s, msk.q);
    const bool neg = alpha > (b.m_sk >> 1);
    u64 *dst = y3 + poly * (size_t)k * n + s;
#pragma unroll
    for (int i = 0; i < k; i++) {
        const ModParams mi = mods[i];
        acc128 a{0, 0};
#pragma unroll
        for (int j = 0; j < ka; j++) acc_mad(a, z[j], b.mhat_mod_q[i][j]);
        if (neg) acc_mad(a, b.M_mod_q[i], b.m_sk - alpha);                                          // :553-559
        else acc_mad(a, mi.q - b.M_mod_q[i], alpha);                                                // :561-569
        u64 v = barrett128(a.lo, a.hi, mi);
        if (premul_c2 && poly % 3 == 2) v = mulmod_shoup(v, b.inv_qhat[i], b.inv_qhat_s[i], mi.q);
        dst[(size_t)i * n] = v;
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

// lazy 28-bit-limb variant of relin_mac_kernel (moduli 2^b - d, 52 <= b <= 55; the arithmetic of mac3_kernel): E arrives packed from the digit
// transform, the keys packed and canonical from evk_pack_kernel; 3 v_mad_u64_u32 per product, D <= 48 terms stay below 2^63 per limb sum
__global__ void __launch_bounds__(256) relin_mac28_kernel(const u64 *E, const u64 *keyp, u64 *out, const ModParams *mods, int n, int k, int D, RelinTab tab)
{
    const int sblocks = n / blockDim.x;
    const size_t cj = blockIdx.x / sblocks;                          // ct*k + j
    const size_t ct = cj / k; const int j = (int)(cj % k);
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const ModParams m = mods[j];
    u64 A0[2] = {0, 0}, A1[2] = {0, 0}, A2[2] = {0, 0};
    const u64 *e = E + (ct * D * k + j) * (size_t)n + s;
    const size_t kn = (size_t)k * n;
#pragma unroll 4
    for (int g = 0; g < D; g++) {
        const u64 ev = e[(size_t)g * kn];
        const u32 e0 = (u32)ev, e1 = (u32)(ev >> 32), es = e0 + e1;
        const u64 *key = keyp + tab.keyoff[g] + (size_t)j * n + s;
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const u64 kv = key[(size_t)p * kn];
            const u32 k0 = (u32)kv, k1 = (u32)(kv >> 32);
            A0[p] += (u64)e0 * k0; A2[p] += (u64)e1 * k1; A1[p] += (u64)es * (k0 + k1);
        }
    }
    out[((ct * 2 + 0) * k + j) * (size_t)n + s] = mac_reduce_fold(A0[0], A1[0], A2[0], 0, m);
    out[((ct * 2 + 1) * k + j) * (size_t)n + s] = mac_reduce_fold(A0[1], A1[1], A2[1], 0, m);
}
// evaluation keys (SEAL hands them over with lazy, possibly non-canonical residues) -> canonical, 28-bit limb pairs; rows alternate over the k moduli
__global__ void __launch_bounds__(256) evk_pack_kernel(const u64 *evk, u64 *out, const ModParams *mods, int n, int k)
{
    const size_t row = blockIdx.x;
    const ModParams m = mods[row % k];
    const u64 *src = evk + row * (size_t)n; u64 *dst = out + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const u64 v = barrett128(src[s], 0, m);
        dst[s] = (v & 0x0fffffffULL) | ((v >> 28) << 32);
    }
}

// c2 (third poly of x3 [count][3][k][n]) -> c2 (q/q_i)^-1 mod q_i, [count][k][n]: what the digit transform reads when the caller's size-3
// ciphertexts are plain BFV ones (crc_relinearize); the fused square + relinearise path gets this form from sq_floor_kernel directly
__global__ void __launch_bounds__(256) relin_premul_kernel(const u64 *x3, u64 *pm, const ModParams *mods, const BehzParams *bp, int n, int k)
{
    const size_t r = blockIdx.x;                 // ct*k + i
    const size_t ct = r / k; const int i = (int)(r % k);
    const u64 q = mods[i].q, inv = bp->inv_qhat[i], invs = bp->inv_qhat_s[i];
    const u64 *src = x3 + ((ct * 3 + 2) * k + i) * (size_t)n;
    u64 *dst = pm + r * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) dst[s] = mulmod_shoup(src[s], inv, invs, q);      // evaluator.cpp:984-985
}

size_t k_square_work_words(const crc_ctx *c, size_t cnt)
{
    const size_t n = c->n, k = c->k, kb = k_square64_supported(c) && (size_t)c->sq64.kf > (size_t)c->kb ? c->sq64.kf : c->kb;
    // QN[2k] BS[2kb] DQ[3k] DB[3kb]   (kb: whichever auxiliary base has more rows -- SEAL's 61-bit one or the engine's fp64 primes, kernels_square64.hip)
    return cnt * n * (2 * k + 2 * kb + 3 * k + 3 * kb);
}
size_t k_relin_work_words(const crc_ctx *c, size_t cnt, int dbc)
{
    const size_t n = c->n, k = c->k;
    size_t D = 0; for (int i = 0; i < c->k; i++) D += evk_digits(c->q[i], dbc);
    // PM[k] E[D*k] R[2k]   (the prepared keys live in front of the caller's work space: k_relin_keys_words more); either path must fit
    const size_t a = cnt * n * (k + D * k + 2 * k), b = k_relin64_work_words(c, cnt, dbc);
    return a > b ? a : b;
}
size_t k_relin_keys_words(const crc_ctx *c, int dbc) { return k_relin64_keys_words(c, dbc); }      // (>= crc_evk_words: the 28-bit packed keys of the other path fit too)

int k_square(crc_ctx *c, const u64 *x, size_t cnt, u64 *y3, u64 *work, hipStream_t st, bool in_ntt, bool premul_c2)
{
    if (cnt == 0) return CRC_OK;
    // the auxiliary base over the engine's fp64 primes whenever the parameters fit (every set of the reference's does); tune.sq_path = 1 keeps SEAL's 61-bit base
    if (c->tune.sq_path != 1 && k_square64_supported(c)) return k_square64(c, x, cnt, y3, work, st, in_ntt, premul_c2);
    const size_t n = c->n, k = c->k, kb = c->kb;
    u64 *QN = work, *BS = QN + cnt * 2 * k * n, *DQ = BS + cnt * 2 * kb * n, *DB = DQ + cnt * 3 * k * n;
    const int threads = c->n < 256 ? c->n : 256, sblocks = c->n / threads;
    int rc;
    // the square needs x in both forms: coefficients for the base extension, NTT values (base q) for the products.  An NTT-resident
    // caller hands over the latter, so one inverse transform replaces the forward one (and the caller's own conversion disappears)
    const u64 *xc = x, *xn = QN;
    if (in_ntt) { if ((rc = k_ntt_ct(c, true, x, QN, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc; xc = QN; xn = x; }
    {
        const dim3 grid((unsigned)(cnt * 2 * sblocks)), blk(threads);
        bool launched = false;
#define LIFT(KV, KAV) if (c->k == KV && c->ka == KAV) { hipLaunchKernelGGL((sq_lift_kernel<KV, KAV + 1>), grid, blk, 0, st, xc, BS, c->d_mods, c->d_behz, c->n); launched = true; }
        CRC_FOR_ALL_K_KA(LIFT)
#undef LIFT
        if (!launched) return CRC_ERR_UNSUPPORTED;
        HIPCHK(hipGetLastError());
    }
    if (!in_ntt && (rc = k_ntt_ct(c, false, x, QN, cnt, 2, false, st, nullptr, 0, 0, 0))) return rc;
    if ((rc = k_ntt_ct(c, false, BS, BS, cnt, 2, true, st, nullptr, 0, 0, 0))) return rc;
    // a^2, 2ab, b^2 are formed while the inverse transforms load their rows (no product tensors in memory)
    if ((rc = k_square_intt(c, xn, DQ, cnt, false, st))) return rc;
    if ((rc = k_square_intt(c, BS, DB, cnt, true, st))) return rc;
    {
        const dim3 grid((unsigned)(cnt * 3 * sblocks)), blk(threads);
        bool launched = false;
#define FLOOR(KV, KAV) if (c->k == KV && c->ka == KAV) { hipLaunchKernelGGL((sq_floor_kernel<KV, KAV>), grid, blk, 0, st, DQ, DB, y3, c->d_mods, c->d_behz, c->n, premul_c2 ? 1 : 0); launched = true; }
        CRC_FOR_ALL_K_KA(FLOOR)
#undef FLOOR
        if (!launched) return CRC_ERR_UNSUPPORTED;
        HIPCHK(hipGetLastError());
    }
    return CRC_OK;
}

// c2_premul: the third polynomial of x3 already holds c2 (q/q_i)^-1 (k_square with premul_c2).  kp: crc_evk_words of space for the packed keys;
// keys_ready: a previous call with the same evk has filled it
int k_relinearize(crc_ctx *c, const u64 *x3, size_t cnt, const u64 *evk, int dbc, u64 *y, u64 *work, u64 *kp, hipStream_t st, bool out_ntt, bool c2_premul, bool keys_ready)
{
    if (cnt == 0) return CRC_OK;
    if (dbc < 1 || dbc > 60) return CRC_ERR_INVALID_ARGUMENT;
    const size_t n = c->n, k = c->k;
    // key switching over the context's two fp64 primes (kernels_relin64.hip) whenever the inner products fit below p_0 p_1 / 4 -- every parameter set of the reference
    // with 16-bit digits does; tune.relin_path = 1 keeps the transforms over the coefficient moduli (the round-2 path below)
    if (c->tune.relin_path != 1 && k_relin64_supported(c, dbc)) {
        int rc;
        if (!keys_ready && (rc = k_relin64_prepare_keys(c, evk, dbc, kp, work, st))) return rc;        // (borrows the scratch: before anything else is put there)
        const u64 *src = x3; int src_size = 3, src_poly = 2;
        if (!c2_premul) {
            hipLaunchKernelGGL(relin_premul_kernel, dim3((unsigned)(cnt * k)), dim3(256), 0, st, x3, work, c->d_mods, c->d_behz, c->n, c->k);
            HIPCHK(hipGetLastError());
            src = work; src_size = 1; src_poly = 0;
        }
        return k_relinearize64(c, src, src_size, src_poly, x3, 3, cnt, dbc, y, c2_premul ? work : work + cnt * k * n, kp, st, out_ntt);
    }
    RelinTab tab{};
    int D = 0; long long off = 0;
    for (int i = 0; i < c->k; i++) {
        int L = evk_digits(c->q[i], dbc);
        for (int d = 0; d < L; d++) { if (D >= 48) return CRC_ERR_UNSUPPORTED; tab.dig_i[D] = (unsigned char)i; tab.dig_shift[D] = (unsigned char)(d * dbc); tab.keyoff[D] = off + (long long)(2 * d) * k * n; D++; }
        off += (long long)2 * L * k * n;
    }
    // the limb MAC (and its folding reduction) needs moduli 2^b - d with 52 <= b <= 55; anything else keeps 128-bit accumulators
    bool limb = true;
    for (int i = 0; i < c->k; i++) if (!c->tabs[i].m.fold || c->tabs[i].m.bits > 55) limb = false;
    u64 *PM = work, *E = PM + cnt * k * n, *R = E + cnt * D * k * n, *KP = kp;
    int rc;
    if (!c2_premul) {
        hipLaunchKernelGGL(relin_premul_kernel, dim3((unsigned)(cnt * k)), dim3(256), 0, st, x3, PM, c->d_mods, c->d_behz, c->n, c->k);
        HIPCHK(hipGetLastError());
    }
    // forward NTT of every digit polynomial under every q_j: rows [cnt*D][k][n]; the digit is cut out of the premultiplied c2 on load
    if ((rc = k_digit_ntt(c, c2_premul ? x3 : PM, c2_premul ? 3 : 1, c2_premul ? 2 : 0, cnt, D, tab.dig_i, tab.dig_shift, dbc, E, st, limb ? 1 : 0))) return rc;
    const int threads = c->n < 256 ? c->n : 256, sblocks = c->n / threads;
    if (limb) {
        if (!keys_ready) {
            hipLaunchKernelGGL(evk_pack_kernel, dim3((unsigned)(crc_evk_words(c, dbc) / n)), dim3(256), 0, st, evk, KP, c->d_mods, c->n, c->k);
            HIPCHK(hipGetLastError());
        }
        hipLaunchKernelGGL(relin_mac28_kernel, dim3((unsigned)(cnt * k * sblocks)), dim3(threads), 0, st, E, KP, R, c->d_mods, c->n, c->k, D, tab);
    } else
        hipLaunchKernelGGL(relin_mac_kernel, dim3((unsigned)(cnt * k * sblocks)), dim3(threads), 0, st, E, evk, R, c->d_mods, c->n, c->k, D, tab);
    HIPCHK(hipGetLastError());
    // INTT and add (c0, c1) of the size-3 input   (evaluator.cpp:1041-1068); for an NTT-form result the sum is formed on the other
    // side of the (linear) transform: NTT(c0, c1) + R
    if (out_ntt) return k_ntt_ct_head_add(c, x3, 3, y, cnt, R, st);
    return k_ntt_ct_addct(c, R, y, cnt, x3, 3, st);
}
