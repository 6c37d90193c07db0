// kernels_relin64.hip -- relinearisation's key switching over TWO fp64 NTT primes instead of the k coefficient moduli.
//
// Reference: Evaluator::relinearize_one_step (evaluator.cpp:934-1069).  For every target modulus q_j and output polynomial the reference forms
//     R_j = sum_{i < k} sum_{d < L_i}  e_{i,d} (*) key_{i,d}[j]      mod q_j         ((*) = negacyclic product, e_{i,d} = digit d of c2 (q/q_i)^-1 mod q_i)
// with 4 k^2 forward transforms over the 55-bit primes (every digit polynomial under every q_j), 2 k inverse ones, and adds c0, c1.  The digits are below 2^dbc (16
// bits) and the key residues below q_j, so over the INTEGERS |R_j| <= n D 2^dbc q_j / 2 (keys taken as centred residues) < 2^90: R_j is a fixed integer polynomial,
// and its residue mod q_j -- all the reference needs -- can be had from its residues modulo ANY primes whose product exceeds 2 |R_j|.  We take the two fp64 primes of
// the context (p_0 p_1 ~ 2^94, f64mod.h): 2 D forward transforms of 16-bit polynomials and 2 * 2k inverse ones per ciphertext in 6-flop fp64 arithmetic, the keys
// re-expressed once per call (inverse transform mod q_j, centre, reduce mod p_m, forward transform mod p_m, times n^-1), a CRT lift per coefficient.  Same element of
// Z_q, hence the same bits (goldens: tests/golden/ops_*.npz ref_relin*, layers / nets).
//   K1 relin_digits_f64_kernel : one workgroup per (ciphertext, i): the L_i digits of row c2'_i, each transformed under p_0 and p_1 in LDS     -> E  [ct][g][m][n]
//   K2 relin_mac_f64_kernel    : per slot: A[ct][poly][j][m] = sum_g E[ct][g][m] Kf[g][poly][j][m]  (lazy sum of reduced products, < 42 p)         -> A  [ct][2k][m][n]
//   K3 relin_inv_crt_kernel    : one workgroup per (ciphertext, poly, j): both inverse transforms, CRT lift, mod q_j, + c_poly; coefficient form out, or the
//                                forward transform over q_j in the same LDS image for an NTT-resident result
#include "kernels.h"
#include "ntt_device.h"

typedef double2 d2;

// ---- fp64 butterflies (the index arithmetic of ntt_device.h's fwd_stages / inv_stages; arithmetic of f64mod.h) ---------------------------------------------------
// forward: values grow by at most 0.875 p per stage: 16-bit inputs stay below 14 p < 2^51 through 15 stages -- no reduction anywhere
template <int R>
__device__ __forceinline__ void fwd_stages_f64(double (&v)[1 << R], const d2 *W, int m, int blk, double p)
{
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << (R - 1 - st);
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const d2 tw = W[(m << st) + (blk << st) + (c >> (R - st))];
            const double X = v[c], T = f64_mulmod_const(v[c + half], tw.x, tw.y, p);
            v[c] = X + T; v[c + half] = X - T;
        }
    }
}
// inverse (Gentleman-Sande, no halving: n^-1 sits in the keys): sums double per stage, so a pass starts from reduced values (|x| <= p/2 -> below 4 p after three stages)
template <int R>
__device__ __forceinline__ void inv_stages_f64(double (&v)[1 << R], const d2 *W, int h, int blk, double p)
{
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << st;
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const d2 tw = W[(h >> st) + (blk << (R - 1 - st)) + (c >> (st + 1))];
            const double U = v[c], V = v[c + half];
            v[c] = U + V; v[c + half] = f64_mulmod_const(U - V, tw.x, tw.y, p);
        }
    }
}
template <bool INV, int R>
__device__ __forceinline__ void ntt_pass_f64(double *sm, const d2 *W, int n, int s, int tabidx, const F64Mod md, bool reduce_in)
{
    const int groups = n >> R;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        const int blk = g / s, l = g - blk * s;
        const int base = blk * (s << R) + l;
        double v[1 << R];
#pragma unroll
        for (int c = 0; c < (1 << R); c++) { v[c] = sm[lpad(base + c * s)]; if (INV && reduce_in) v[c] = f64_reduce(v[c], md); }
        if (INV) inv_stages_f64<R>(v, W, tabidx, blk, md.p); else fwd_stages_f64<R>(v, W, tabidx, blk, md.p);
#pragma unroll
        for (int c = 0; c < (1 << R); c++) sm[lpad(base + c * s)] = v[c];
    }
    __syncthreads();
}
// all passes of one row on the LDS image (lpad-swizzled); caller has synchronised after filling it, returns synchronised.  Inverse: the image holds values below
// 2^52 (lazy sums of up to 48 products); every pass reduces on load.
template <bool INV>
__device__ __forceinline__ void ntt_row_passes_f64(double *sm, const d2 *W, int n, int logn, const F64Mod md)
{
    const int full = logn / 3, rem = logn - 3 * full;
    if (!INV) {
        int t = n >> 1;
        for (int p = 0; p < full; p++, t >>= 3) ntt_pass_f64<false, 3>(sm, W, n, t >> 2, n / (2 * t), md, false);
        if (rem == 2) ntt_pass_f64<false, 2>(sm, W, n, t >> 1, n / (2 * t), md, false);
        else if (rem == 1) ntt_pass_f64<false, 1>(sm, W, n, t, n / (2 * t), md, false);
    } else {
        int t = 1;
        for (int p = 0; p < full; p++, t <<= 3) ntt_pass_f64<true, 3>(sm, W, n, t, n / (2 * t), md, true);
        if (rem == 2) ntt_pass_f64<true, 2>(sm, W, n, t, n / (2 * t), md, true);
        else if (rem == 1) ntt_pass_f64<true, 1>(sm, W, n, t, n / (2 * t), md, true);
    }
}

struct Relin64Tab { unsigned char L[CRC_MAXK], g0[CRC_MAXK]; };

// ---- key preparation (once per call) ------------------------------------------------------------------------------------------------------------------------------
// evaluation keys as SEAL hands them over (NTT form over q_j, residues possibly lazy / non-canonical) -> canonical
__global__ void __launch_bounds__(256) evk_canon_kernel(const u64 *evk, u64 *out, const ModParams *mods, int n, int k)
{
    const size_t row = blockIdx.x;
    const ModParams m = mods[row % k];
    const u64 *src = evk + row * (size_t)n; u64 *dst = out + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) dst[s] = barrett128(src[s], 0, m);
}
// kc: the keys in coefficient form over q_j, rows [2 g + poly][j] (blob order)  ->  Kf [g][poly k + j][m][n]: centred residue mod p_m, forward transform, times n^-1
__global__ void __launch_bounds__(1024) relin_keys_f64_kernel(const u64 *kc, double *Kf, const ModParams *mods, const d2 *Wf, F64Params fp, int n, int logn, int k)
{
    extern __shared__ double smd[];
    const int m = blockIdx.x % CRC_NF64; const size_t row = blockIdx.x / CRC_NF64;       // row = (2 g + poly) k + j
    const int j = (int)(row % k); const size_t gp = row / k; const size_t g = gp >> 1; const int poly = (int)(gp & 1);
    const u64 q = mods[j].q;
    const F64Mod md = fp.m[m];
    const u64 *src = kc + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        const u64 v = src[s];
        smd[lpad(s)] = f64_from_i64(v > (q >> 1) ? (long long)v - (long long)q : (long long)v, md);
    }
    __syncthreads();
    ntt_row_passes_f64<false>(smd, Wf + (size_t)m * n, n, logn, md);
    double *dst = Kf + (((g * 2 * k + (size_t)poly * k + j) * CRC_NF64) + m) * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) dst[s] = f64_reduce(f64_mulmod_const(smd[lpad(s)], fp.ninv[m], fp.ninv_q[m], md.p), md);
}

// ---- K1: digits of c2' under both primes ---------------------------------------------------------------------------------------------------------------------------
// src: size-`src_size` ciphertexts, poly `src_poly` = c2 (q/q_i)^-1 mod q_i (evaluator.cpp:984-985); E [ct][g][m][n], unreduced (|.| < 14 p)
__global__ void __launch_bounds__(1024) relin_digits_f64_kernel(const u64 *src, int src_size, int src_poly, double *E, const d2 *Wf, F64Params fp, int n, int logn, int k, int D,
                                                                int dbc, Relin64Tab tab)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / k; const int i = blockIdx.x % k;
    const u64 *row = src + ((ct * src_size + src_poly) * k + i) * (size_t)n;
    const u64 mask = (1ULL << dbc) - 1;
    const int L = tab.L[i], g0 = tab.g0[i];
    for (int d = 0; d < L; d++) {
        const int sh = d * dbc;
        for (int m = 0; m < CRC_NF64; m++) {
            for (int s = threadIdx.x; s < n; s += blockDim.x) smd[lpad(s)] = (double)(u32)((row[s] >> sh) & mask);
            __syncthreads();
            ntt_row_passes_f64<false>(smd, Wf + (size_t)m * n, n, logn, fp.m[m]);
            double *dst = E + ((ct * D + g0 + d) * CRC_NF64 + m) * (size_t)n;
            for (int s = threadIdx.x; s < n; s += blockDim.x) dst[s] = smd[lpad(s)];
            __syncthreads();
        }
    }
}

// ---- K2: slot-wise inner products ------------------------------------------------------------------------------------------------------------------------------------
// A[ct][pj][m][s] = sum_g Kf[g][pj][m][s] E[ct][g][m][s]: every product reduced below 0.875 p, the sum of D <= 48 of them stays below 2^53 (exact); CT ciphertexts share
// every key value a thread loads
template <int K, int CT>
__global__ void __launch_bounds__(256) relin_mac_f64_kernel(const double *E, const double *Kf, double *A, F64Params fp, int n, int D, size_t cnt)
{
    const int sblocks = n / blockDim.x;
    const int s = (blockIdx.x % sblocks) * blockDim.x + threadIdx.x;
    const int m = (blockIdx.x / sblocks) % CRC_NF64;
    const size_t ct0 = (size_t)(blockIdx.x / (sblocks * CRC_NF64)) * CT;
    const F64Mod md = fp.m[m];
    double acc[CT][2 * K];
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
        for (int pj = 0; pj < 2 * K; pj++) acc[c][pj] = 0.0;
    const size_t nn = (size_t)n;
    for (int g = 0; g < D; g++) {
        double e[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) e[c] = ct0 + c < cnt ? E[(((ct0 + c) * D + g) * CRC_NF64 + m) * nn + s] : 0.0;
        const double *kr = Kf + (((size_t)g * 2 * K) * CRC_NF64 + m) * nn + s;
#pragma unroll
        for (int pj = 0; pj < 2 * K; pj++) {
            const double kv = kr[(size_t)pj * CRC_NF64 * nn];
#pragma unroll
            for (int c = 0; c < CT; c++) acc[c][pj] += f64_mulmod(kv, e[c], md);
        }
    }
#pragma unroll
    for (int c = 0; c < CT; c++)
        if (ct0 + c < cnt)
#pragma unroll
            for (int pj = 0; pj < 2 * K; pj++) A[(((ct0 + c) * 2 * K + pj) * CRC_NF64 + m) * nn + s] = acc[c][pj];
}

// ---- K3: inverse transforms, CRT lift, mod q_j, + (c0, c1) -------------------------------------------------------------------------------------------------------------
// A [ct][poly k + j][m][n]; x3: size-`add_size` ciphertexts whose polys 0, 1 are added (coefficient form); y [ct][2][k][n].
// NPT = n / blockDim.x values per thread keep the first prime's result in registers while the LDS image serves the second transform.
template <int NPT, bool OUT_NTT, bool LAZY>
__global__ void __launch_bounds__(1024) relin_inv_crt_kernel(const double *A, const u64 *x3, int add_size, u64 *y, const ModParams *mods, const d2 *Wi, const ulonglong2 *Wq, F64Params fp,
                                                             int n, int logn, int k)
{
    extern __shared__ double smd[];
    const size_t ct = blockIdx.x / (2 * k); const int pj = blockIdx.x % (2 * k), poly = pj / k, j = pj % k;
    const int tid = threadIdx.x, nt = blockDim.x;
    double r0[NPT];
    for (int m = 0; m < CRC_NF64; m++) {
        const double *src = A + ((ct * 2 * k + pj) * CRC_NF64 + m) * (size_t)n;
#pragma unroll
        for (int u = 0; u < NPT; u++) smd[lpad(tid + u * nt)] = src[tid + u * nt];
        __syncthreads();
        ntt_row_passes_f64<true>(smd, Wi + (size_t)m * n, n, logn, fp.m[m]);
        if (m == 0) {
#pragma unroll
            for (int u = 0; u < NPT; u++) r0[u] = f64_reduce(smd[lpad(tid + u * nt)], fp.m[0]);
            __syncthreads();
        }
    }
    // x = a0 + p0 t,  t = (a1 - a0) p0^-1 mod p1 centred: |x| < p0 p1 / 2, and the true value is below a quarter of that, so t is nowhere near +- p1 / 2
    const ModParams mq = mods[j];
    const u64 q = mq.q, p0q = fp.p0_mod_q[j];
    const u64 *add = x3 + ((ct * add_size + poly) * k + j) * (size_t)n;
    u64 *dst = y + ((ct * 2 + poly) * k + j) * (size_t)n;
    u64 *sm = reinterpret_cast<u64 *>(smd);           // (a thread reads and rewrites only its own positions of the image here: no barrier in between)
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        const int s = tid + u * nt;
        const double a1 = f64_reduce(smd[lpad(s)], fp.m[1]);
        const double t = f64_reduce(f64_mulmod_const(a1 - r0[u], fp.inv_p0_p1, fp.inv_p0_p1_q, fp.m[1].p), fp.m[1]);
        const long long ti = (long long)t, a0i = (long long)r0[u];
        u64 lo, hi; mul64wide((u64)(ti < 0 ? -ti : ti), p0q, lo, hi);
        u64 r = barrett128(lo, hi, mq);
        if (ti < 0) r = negmod(r, q);
        u64 a0m = (u64)(a0i < 0 ? -a0i : a0i);                    // |a0| < 2^46: below q for the 54..60-bit coefficient moduli, not for SEAL's 40-bit ones
        if (a0m >= q) a0m = barrett128(a0m, 0, mq);
        r = addmod(r, a0i < 0 ? negmod(a0m, q) : a0m, q);
        r = addmod(r, add[s], q);
        if (OUT_NTT) sm[lpad(s)] = r; else dst[s] = r;
    }
    if (!OUT_NTT) return;
    // NTT-resident result: forward transform over q_j of (c_poly + R) in the same LDS image (ntt_device.h)
    __syncthreads();
    ntt_row_passes<false, LAZY>(sm, Wq + (size_t)j * n, n, logn, q, mq.two_q);
    const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
#pragma unroll
    for (int u = 0; u < NPT; u++) {
        u64 v = sm[lpad(tid + u * nt)];
        if (LAZY) v = reduce_small(v, q, mq.two_q, rq);
        else { v = v >= mq.two_q ? v - mq.two_q : v; v = v >= q ? v - q : v; }
        dst[tid + u * nt] = v;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------------------------------------------------------
// can this context / key set take the fp64 path?  2 |R_j| <= n D 2^dbc q_max must stay below p_0 p_1 / 2 (a factor 2 of slack for the floating CRT), the row must fit
// the LDS image the transforms work on, and at most 48 products may be summed lazily
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
size_t k_relin64_keys_words(const crc_ctx *c, int dbc) { return (size_t)CRC_NF64 * crc_evk_words(c, dbc); }
// scratch words: E [cnt][D][2][n] + A [cnt][2k][2][n] (+ PM [cnt][k][n] when the caller's c2 is not premultiplied); the key preparation borrows the same space
size_t k_relin64_work_words(const crc_ctx *c, size_t cnt, int dbc)
{
    size_t D = 0; for (int i = 0; i < c->k; i++) D += evk_digits(c->q[i], dbc);
    const size_t n = c->n, k = c->k;
    const size_t run = cnt * n * (k + CRC_NF64 * D + CRC_NF64 * 2 * k), prep = crc_evk_words(c, dbc);
    return run > prep ? run : prep;
}

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
    int nt = c->n / 8; if (nt < 64) nt = 64; if (nt > 1024) nt = 1024;
    const size_t lds = (size_t)c->n * 8;
    { const int r2 = crc_ctx_ensure_lds(c, (const void *)relin_keys_f64_kernel, lds); if (r2) return r2; }
    hipLaunchKernelGGL(relin_keys_f64_kernel, dim3((unsigned)(rows * CRC_NF64)), dim3(nt), lds, st, scratch, Kf, c->d_mods, reinterpret_cast<const d2 *>(c->d_f64_rp), c->f64, c->n, c->logn, c->k);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

template <int K>
static int relin64_mac(crc_ctx *c, const double *E, const double *Kf, double *A, int D, size_t cnt, hipStream_t st)
{
    constexpr int CT = K <= 2 ? 4 : K <= 4 ? 2 : 1;
    const int threads = c->n < 256 ? c->n : 256, sblocks = c->n / threads;
    const size_t groups = (cnt + CT - 1) / CT;
    hipLaunchKernelGGL((relin_mac_f64_kernel<K, CT>), dim3((unsigned)(groups * CRC_NF64 * sblocks)), dim3(threads), 0, st, E, Kf, A, c->f64, c->n, D, cnt);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

template <int NPT>
static int relin64_tail(crc_ctx *c, const double *A, const u64 *x3, int add_size, u64 *y, size_t cnt, bool out_ntt, hipStream_t st)
{
    bool lazy = true;
    for (int i = 0; i < c->k; i++) if (c->tabs[i].m.bits > 57 || c->tabs[i].m.bits < 45) lazy = false;
    const int nt = c->n / NPT;
    const size_t lds = (size_t)c->n * 8;
    auto kern = !out_ntt ? relin_inv_crt_kernel<NPT, false, false> : lazy ? relin_inv_crt_kernel<NPT, true, true> : relin_inv_crt_kernel<NPT, true, false>;
    { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)(cnt * 2 * c->k)), dim3(nt), lds, st, A, x3, add_size, y, c->d_mods, reinterpret_cast<const d2 *>(c->d_f64_irp),
                       reinterpret_cast<const ulonglong2 *>(c->d_rp), c->f64, c->n, c->logn, c->k);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// src / src_size / src_poly: where c2 (q/q_i)^-1 lives; x3 / add_size: the ciphertexts whose (c0, c1) are added; kp: the keys as k_relin64_prepare_keys left them;
// work: cnt n (2 D + 4 k) words
int k_relinearize64(crc_ctx *c, const u64 *src, int src_size, int src_poly, const u64 *x3, int add_size, size_t cnt, int dbc, u64 *y, u64 *work, const u64 *kp,
                    hipStream_t st, bool out_ntt)
{
    if (cnt == 0) return CRC_OK;
    if (!k_relin64_supported(c, dbc)) return CRC_ERR_UNSUPPORTED;
    const size_t n = c->n, k = c->k;
    Relin64Tab tab{};
    int D = 0;
    for (int i = 0; i < c->k; i++) { tab.L[i] = (unsigned char)evk_digits(c->q[i], dbc); tab.g0[i] = (unsigned char)D; D += tab.L[i]; }
    const double *Kf = reinterpret_cast<const double *>(kp);
    int rc;
    double *E = reinterpret_cast<double *>(work), *A = E + cnt * D * CRC_NF64 * n;
    int nt = c->n / 8; if (nt < 64) nt = 64; if (nt > 1024) nt = 1024;
    const size_t lds = n * 8;
    { const int r2 = crc_ctx_ensure_lds(c, (const void *)relin_digits_f64_kernel, lds); if (r2) return r2; }
    hipLaunchKernelGGL(relin_digits_f64_kernel, dim3((unsigned)(cnt * k)), dim3(nt), lds, st, src, src_size, src_poly, E, reinterpret_cast<const d2 *>(c->d_f64_rp), c->f64, c->n, c->logn,
                       c->k, D, dbc, tab);
    HIPCHK(hipGetLastError());
    switch (c->k) {
#define MACK(KV) case KV: rc = relin64_mac<KV>(c, E, Kf, A, D, cnt, st); break;
    MACK(1) MACK(2) MACK(3) MACK(4) MACK(5) MACK(6) MACK(7) MACK(8)
#undef MACK
    default: return CRC_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    switch (c->n / nt) {
    case 1: return relin64_tail<1>(c, A, x3, add_size, y, cnt, out_ntt, st);
    case 2: return relin64_tail<2>(c, A, x3, add_size, y, cnt, out_ntt, st);
    case 4: return relin64_tail<4>(c, A, x3, add_size, y, cnt, out_ntt, st);
    case 8: return relin64_tail<8>(c, A, x3, add_size, y, cnt, out_ntt, st);
    case 16: return relin64_tail<16>(c, A, x3, add_size, y, cnt, out_ntt, st);
    }
    return CRC_ERR_UNSUPPORTED;
}
