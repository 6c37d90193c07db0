// tools/mfma_rns.hip -- PROTOTYPE (not part of the product; round 6): the ct x pt multiply-accumulate of a convolution as a MULTI-MODULAR int8-MFMA
// GEMM -- 16 int8 products per modular multiply-add where the product's limb GEMM (kernels_mfma.hip; its round-2 prototype is tools/mfma_mac.hip) spends 49.
//
// The limb GEMM writes a 55-bit residue as seven balanced base-256 digits, so x w = sum_{l,m} a_l b_m 256^(l+m) costs 7 x 7 = 49 digit products.  But the quantity the
// layer needs is the INTEGER  V = sum_t x_t w_t  over centred representatives (|V| <= T (q/2)^2 < 2^119 for T = 1152, q < 2^55), reduced mod q ONCE per output -- and an
// integer of 119 bits is determined by its residues modulo pairwise coprime small moduli whose product exceeds 2^120.  With moduli m_j <= 256 the balanced residues of
// x_t and w_t are int8 values, their products sum exactly in int32 (T 2^14 < 2^25), and sixteen moduli suffice:
//     { 256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193 },   product M = 2^125.4
// so ONE v_mfma_i32_32x32x32_i8 per modulus and reduction step forms  D_j = sum_t (x_t mod m_j)(w_t mod m_j)  ==  V (mod m_j):  16 MFMAs instead of 49.  The epilogue
// lifts the sixteen residues back by the Chinese remainder theorem directly modulo q:
//     y_j = D_j (M/m_j)^-1 mod m_j  in [0, m_j);   V = sum_j y_j (M/m_j) - alpha M  with  alpha = round(sum_j y_j / m_j)   (|V| < M / 2^6: the sum is within 0.02 of an
//     integer, a float decides it);   V mod q = sum_j y_j C_j + alpha (q - D)  mod q,   C_j = (M/m_j) mod q,  D = M mod q   -- one 128-bit sum, ONE reduction.
// Exact integer arithmetic throughout: the same element of Z_q as mac3_kernel and the reference, hence the same bits (tools/bench_mfma_rns.py compares them).
// What it costs: sixteen operand planes instead of seven (2.3 x the operand bytes from L2 / LDS per step, no reuse of a fragment across several MFMAs), and a pack
// step that takes 16 small remainders of every activation instead of 7 digits.
//
// Same shape, layouts and staging as tools/mfma_mac.hip (CrCNN's conv2 + pool2: 32 channels, 12 x 12, 6 x 6 window, stride 2, 64 filters, T = 1152), with NPL = 16
// planes and a two-slot LDS ring (2 x 64 KiB).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint64_t u64; typedef uint32_t u32; typedef int32_t i32;
typedef i32 v4i __attribute__((ext_vector_type(4)));
typedef i32 v16i __attribute__((ext_vector_type(16)));

#define NPL 16           // residue planes = small moduli
#define ZD 32
#define XD 12
#define WF 6
#define STR 2
#define NPOS (XD * XD)
#define NF 64
#define IMG_BYTES (NPL * NPOS * 2 * ZD)          // 147456
#define TAPS (WF * WF)
#define TILE_BYTES (NPL * 64 * ZD)               // 32768: one operand tile of a reduction step (A: 64 rows, W: 64 filters)
#define WSLOT_BYTES (TAPS * TILE_BYTES)

__host__ __device__ constexpr int small_mod(int j)
{
    constexpr int M[NPL] = {256, 255, 253, 251, 247, 241, 239, 233, 229, 227, 223, 217, 211, 199, 197, 193};
    return M[j];
}

struct Mod {
    u64 q, half; u32 bits, d;           // q = 2^bits - d
    u32 qm[NPL];                        // q mod m_j
    u32 inv[NPL];                       // (M/m_j)^-1 mod m_j
    u32 bias[NPL];                      // a multiple of m_j above 2^25: makes D_j non-negative
    u32 clo[NPL], chi[NPL];             // C_j = (M/m_j) mod q as two 32-bit words
    u64 qmd;                            // q - (M mod q)
    float rm[NPL];                      // 1 / m_j
};

// canonical residue -> balanced residues of its centred representative modulo the sixteen small moduli
template <int J> struct PackRes {
    static __device__ __forceinline__ void run(u64 r, bool neg, const Mod &m, signed char *dst, size_t stride)
    {
        constexpr int mj = small_mod(J);
        u32 u = (u32)(r % (u64)mj);                         // r mod m_j (compile-time divisor)
        if (neg) { u += mj - m.qm[J]; u = u >= (u32)mj ? u - mj : u; }      // (r - q) mod m_j
        const int b = (int)u > (mj - 1) / 2 ? (int)u - mj : (int)u;        // balanced: [-128, 127] for 256, [-(m-1)/2, (m-1)/2] for odd m
        dst[(size_t)J * stride] = (signed char)b;
        PackRes<J + 1>::run(r, neg, m, dst, stride);
    }
};
template <> struct PackRes<NPL> { static __device__ __forceinline__ void run(u64, bool, const Mod &, signed char *, size_t) {} };

__global__ void __launch_bounds__(256) pack_x_kernel(const u64 *x, signed char *xp, const Mod *mods, int n, int k, int B)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)B * ZD * NPOS * 2 * k;
    if (e >= rows * n) return;
    const int s = (int)(e % n); size_t r = e / n;
    const int i = (int)(r % k); r /= k; const int c = (int)(r % 2); r /= 2;
    const int pos = (int)(r % NPOS); r /= NPOS; const int z = (int)(r % ZD); const int b = (int)(r / ZD);
    const u64 v = x[e];
    signed char *dst = xp + (((size_t)i * n + s) * B + b) * IMG_BYTES + ((size_t)pos * 2 + c) * ZD + z;
    PackRes<0>::run(v, v > mods[i].half, mods[i], dst, (size_t)NPOS * 2 * ZD);
}
__global__ void __launch_bounds__(256) pack_w_kernel(const u64 *w, signed char *wp, const Mod *mods, int n, int k)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)NF * ZD * TAPS * k;
    if (e >= rows * n) return;
    const int s = (int)(e % n); size_t r = e / n;
    const int i = (int)(r % k); r /= k;
    const int tap = (int)(r % TAPS); r /= TAPS; const int z = (int)(r % ZD); const int f = (int)(r / ZD);
    const u64 v = w[e];
    signed char *dst = wp + ((size_t)i * n + s) * WSLOT_BYTES + (size_t)tap * TILE_BYTES + (size_t)f * ZD + z;
    PackRes<0>::run(v, v > mods[i].half, mods[i], dst, (size_t)NF * ZD);
}
__global__ void __launch_bounds__(256) unpack_y_kernel(const u64 *ys, u64 *y, int n, int k, int B)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)B * NF * 16 * 2 * k;
    if (e >= rows * n) return;
    const int s = (int)(e % n); size_t r = e / n;
    const int i = (int)(r % k); r /= k;
    const size_t b = r / (NF * 32), rest = r % (NF * 32);
    y[e] = ys[(((size_t)i * n + s) * B + b) * (NF * 32) + rest];
}

// the sixteen accumulators of one output -> V mod q, canonical
template <int J> struct Crt {
    static __device__ __forceinline__ void run(const i32 (&D)[NPL], const Mod &m, u64 &slo, u64 &shi, float &fs)
    {
        constexpr u32 mj = (u32)small_mod(J);
        const u32 r = ((u32)D[J] + m.bias[J]) % mj;            // D_j mod m_j in [0, m_j)   (compile-time divisor)
        const u32 y = (r * m.inv[J]) % mj;                     // y_j
        slo += (u64)y * m.clo[J]; shi += (u64)y * m.chi[J];    // sum_j y_j C_j as two lazy word sums (16 terms of 8 + 32 bits)
        fs = __builtin_fmaf((float)y, m.rm[J], fs);
        Crt<J + 1>::run(D, m, slo, shi, fs);
    }
};
template <> struct Crt<NPL> { static __device__ __forceinline__ void run(const i32 (&)[NPL], const Mod &, u64 &, u64 &, float &) {} };

__device__ __forceinline__ u64 reduce_crt(const i32 (&D)[NPL], const Mod &m)
{
    u64 slo = 0, shi = 0; float fs = 0.f;
    Crt<0>::run(D, m, slo, shi, fs);
    const u32 alpha = (u32)__builtin_rintf(fs);
    // U = shi 2^32 + slo + alpha (q - M mod q)  <  2^68
    u64 lo = slo + (shi << 32), hi = (shi >> 32) + (lo < slo);
    { u64 tl = (u64)alpha * (u32)m.qmd, th = (u64)alpha * (u32)(m.qmd >> 32);          // alpha < 17, qmd < 2^55
      const u64 add = tl + (th << 32); const u64 nl = lo + add; hi += (th >> 32) + (nl < lo); lo = nl; }
    // fold twice (q = 2^b - d, d < 2^26, b >= 52): U < 2^68 -> below 2^b + 2^42, then canonical
    const u32 b = m.bits; const u64 d = m.d, mask = ((u64)1 << b) - 1;
    const u64 h1 = (lo >> b) | (hi << (64 - b));                                         // < 2^16
    u64 r = h1 * d + (lo & mask);
    r = (r >> b) * d + (r & mask);
    return r >= m.q ? r - m.q : r;
}

// grid: one workgroup per (residue, slot, image pair); 256 threads = 4 waves in a 2 x 2 arrangement of 32 x 32 tiles; MODE 2 = no MFMAs (ablation, wrong results)
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mfma_rns_kernel(const signed char *xp, const signed char *wp, u64 *ys, const Mod *mods, int n, int k, int B)
{
    extern __shared__ __attribute__((aligned(16))) signed char lds[];          // ring of 2 x (A tile | W tile) = 2 x 65536 B
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    const int pairs = B / 2, slots = n * k;
    int g = blockIdx.x, slot, bp;
    if ((slots & 7) == 0) { const int xcd = g & 7, r = g >> 3; slot = xcd * (slots >> 3) + r / pairs; bp = r % pairs; }
    else { slot = g / pairs; bp = g % pairs; }
    const int i = slot / n;
    const signed char *ximg = xp + ((size_t)slot * B + 2 * bp) * IMG_BYTES;
    const signed char *wsl = wp + (size_t)slot * WSLOT_BYTES;

    // staging: 64 LDS-DMA pieces of 1 KiB per step (pieces 0..31 of A, 32..63 of W), 16 per wave
    constexpr int NST = 2, PCS = 16;
    u32 src_off[PCS];
#pragma unroll
    for (int j = 0; j < PCS; j++) {
        const int pc = wave + 4 * j;
        if (pc < 32) {
            const int c16 = pc * 64 + lane;
            const int plane = c16 >> 7, row = (c16 >> 1) & 63, half = c16 & 1;
            const int img = row >> 5, p = (row >> 1) & 15, c = row & 1;
            const int ox = p >> 2, oy = p & 3;
            src_off[j] = (u32)(img * IMG_BYTES + plane * (NPOS * 2 * ZD) + (((ox * STR) * XD + oy * STR) * 2 + c) * ZD + half * 16);
        } else src_off[j] = (u32)((pc - 32) * 1024 + lane * 16);
    }
    auto issue = [&](int tap) {
        const int kx = tap / WF, ky = tap % WF;
        const u32 tapoff = (u32)((kx * XD + ky) * 2 * ZD);
        signed char *dst = lds + (tap % NST) * (2 * TILE_BYTES);
        const signed char *wt = wsl + (size_t)tap * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < PCS; j++) {
            const int pc = wave + 4 * j;
            const signed char *src = pc < 32 ? ximg + src_off[j] + tapoff : wt + src_off[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
        }
    };

    v16i acc[NPL];
#pragma unroll
    for (int d = 0; d < NPL; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[d][e] = 0;

    issue(0);
    const int fragA = (wm * 32 + (lane & 31)) * ZD + (lane >> 5) * 16, fragW = (wn * 32 + (lane & 31)) * ZD + (lane >> 5) * 16;
    for (int tap = 0; tap < TAPS; tap++) {
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));    // this wave's pieces of step `tap` have landed ...
        __syncthreads();                                          // ... and everybody's; the other ring slot, read in step tap - 1, is free
        if (tap + 1 < TAPS) issue(tap + 1);
        const signed char *tA = lds + (tap % NST) * (2 * TILE_BYTES), *tW = tA + TILE_BYTES;
        // one product per modulus: sixteen independent accumulators (no dependent back-to-back MFMAs)
#pragma unroll
        for (int l = 0; l < NPL; l++) {
            const v4i a = *reinterpret_cast<const v4i *>(tA + l * (64 * ZD) + fragA);
            const v4i w = *reinterpret_cast<const v4i *>(tW + l * (64 * ZD) + fragW);
            if (MODE == 2) { acc[l][0] += a[0] + w[0]; continue; }
            acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, acc[l], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of the 32 x 32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const Mod m = mods[i];
    const int f = wn * 32 + (lane & 31);
    u64 *dst = ys + ((size_t)slot * B + 2 * bp + wm) * (NF * 32) + (size_t)f * 32;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        i32 D[NPL];
#pragma unroll
        for (int d = 0; d < NPL; d++) D[d] = acc[d][reg];
        dst[row] = reduce_crt(D, m);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- host entry points (ctypes): device pointers, default stream ------------------------------------------------------------------
static u64 powmod_small(u64 a, u64 e, u64 m) { u64 r = 1 % m; a %= m; while (e) { if (e & 1) r = r * a % m; a = a * a % m; e >>= 1; } return r; }
static int set_mods(Mod *d_mods, const u64 *q, int k)
{
    typedef unsigned __int128 u128;
    Mod h[8];
    for (int i = 0; i < k; i++) {
        Mod &m = h[i];
        u32 bits = 64 - __builtin_clzll(q[i]);
        m.q = q[i]; m.half = q[i] >> 1; m.bits = bits; m.d = (u32)(((u64)1 << bits) - q[i]);
        u64 Mq = 1;                                                  // M mod q
        for (int j = 0; j < NPL; j++) Mq = (u64)((u128)Mq * (u64)small_mod(j) % q[i]);
        m.qmd = q[i] - Mq;
        for (int j = 0; j < NPL; j++) {
            const u64 mj = (u64)small_mod(j);
            m.qm[j] = (u32)(q[i] % mj);
            u64 hat_m = 1, hat_q = 1;                                // (M/m_j) mod m_j, mod q
            for (int l = 0; l < NPL; l++) if (l != j) { hat_m = hat_m * ((u64)small_mod(l) % mj) % mj; hat_q = (u64)((u128)hat_q * (u64)small_mod(l) % q[i]); }
            // inverse modulo m_j (m_j need not be prime: search; m_j <= 256)
            u32 inv = 0; for (u32 c = 1; c < mj; c++) if (hat_m * c % mj == 1) { inv = c; break; }
            if (!inv) return -5;
            m.inv[j] = inv;
            m.bias[j] = (u32)(((1u << 25) / mj + 1) * mj);
            m.clo[j] = (u32)hat_q; m.chi[j] = (u32)(hat_q >> 32);
            m.rm[j] = 1.0f / (float)mj;
        }
    }
    (void)powmod_small;
    return hipMemcpy(d_mods, h, sizeof(Mod) * k, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
extern "C" size_t rr_mod_bytes(int k) { return sizeof(Mod) * (size_t)k; }
extern "C" size_t rr_xp_bytes(int n, int k, int B) { return (size_t)n * k * B * IMG_BYTES; }
extern "C" size_t rr_wp_bytes(int n, int k) { return (size_t)n * k * WSLOT_BYTES; }
extern "C" size_t rr_ys_bytes(int n, int k, int B) { return (size_t)n * k * B * NF * 32 * 8; }
extern "C" int rr_pack_w(const u64 *w, signed char *wp, const u64 *q, int n, int k, void *d_mods)
{
    const int rc = set_mods((Mod *)d_mods, q, k);
    if (rc) return rc;
    const size_t e = (size_t)NF * ZD * TAPS * k * n;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, w, wp, (const Mod *)d_mods, n, k);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int rr_pack_x(const u64 *x, signed char *xp, int n, int k, int B, void *d_mods)
{
    const size_t e = (size_t)B * ZD * NPOS * 2 * k * n;
    hipLaunchKernelGGL(pack_x_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, x, xp, (const Mod *)d_mods, n, k, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int rr_conv(const signed char *xp, const signed char *wp, u64 *ys, int n, int k, int B, void *d_mods, int mode)
{
    if (B % 2) return -3;
    const size_t lds = 4 * TILE_BYTES;                    // 2-slot ring of (A tile | W tile): 128 KiB
    auto kern = mode == 2 ? mfma_rns_kernel<2> : mfma_rns_kernel<0>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -4;
    const size_t grid = (size_t)n * k * (B / 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, 0, xp, wp, ys, (const Mod *)d_mods, n, k, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int rr_unpack_y(const u64 *ys, u64 *y, int n, int k, int B)
{
    const size_t e = (size_t)B * NF * 32 * k * n;
    hipLaunchKernelGGL(unpack_y_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, ys, y, n, k, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
