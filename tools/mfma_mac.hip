// tools/mfma_mac.hip -- PROTOTYPE (not part of the product; VERDICT r1 task 6): the ct x pt multiply-accumulate of a convolution as an int8-MFMA
// limb GEMM, to measure what the matrix cores would buy over the v_mad_u64_u32 roofline that mac3_kernel sits on.
//
// Per residue i and slot s the layer is a GEMM over Z_q:  Y[m][f] = sum_t A[m][t] W[t][f],  m = (image, pixel, poly), t = (kx, ky, z).
// Every 55-bit residue r is written as the centred representative r' in (-q/2, q/2] (same class mod q) in balanced base 256:
//     r' = sum_{l<7} d_l 256^l,  d_l in [-128, 127]   (|r'| < 2^54  =>  |d_6| <= 64)
// so x w = sum_{l,m} a_l b_m 256^(l+m): 49 int8 products that land on 13 diagonals l+m.  One v_mfma_i32_32x32x32_i8 forms a 32 x 32 tile of
// 32-term dot products of one (l, m) pair and adds it to diagonal l+m's int32 accumulator -- exact: |D| <= T * 7 * 128^2 < 2^31 up to T = 18 000.
// After the reduction loop  V = sum_d D_d 2^(8d)  is reduced mod q once per output.  Bit-exact against mac3_kernel (tools/bench_mfma.py).
//
// Specialised to CrCNN's conv2+pool2 shape (the dominant layer of PlainModelTiny: 32 channels, 12 x 12 input, 6 x 6 window, stride 2, 4 x 4
// outputs, 64 filters, T = 1152).  Operands come in a slot-major limb layout (pack kernels below; in a product the producing kernels would
// write it directly, as they do CRC_NTTP today):
//     Xp [k][n][B][7 planes][144 positions][2 polys][32 channels]  int8        64 512 B per (slot, image)
//     Wp [k][n][36 taps (kx,ky)][7 planes][64 filters][32 channels] int8       516 096 B per slot
//     Ys [k][n][B][64 filters][16 pixels][2 polys] u64                         slot-major result, canonical residues
// Workgroup = one slot, two images (64 rows) x 64 filters; 4 waves, one per SIMD, a 32 x 32 output tile each with 13 x 16 int32 accumulators.
// Per reduction step (one tap, 32 channels) the workgroup needs 64 x 32 B x 7 planes of A (implicit im2col: gathered from the two image blocks by
// LDS-DMA with per-lane addresses) and as much of W (contiguous), double-buffered in 56 KiB of LDS; 49 MFMAs per wave and step.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint64_t u64; typedef uint32_t u32; typedef int32_t i32;
typedef i32 v4i __attribute__((ext_vector_type(4)));
typedef i32 v16i __attribute__((ext_vector_type(16)));

#define NPL 7            // limb planes
#define ZD 32
#define XD 12
#define WF 6             // window
#define STR 2
#define XO 4
#define NPOS (XD * XD)
#define NF 64
#define IMG_BYTES (NPL * NPOS * 2 * ZD)          // 64512
#define TAPS (WF * WF)
#define WSLOT_BYTES (TAPS * NPL * NF * ZD)       // 516096
#define TILE_BYTES (NPL * 64 * ZD)               // 14336: one operand tile of a reduction step (A: 64 rows, W: 64 filters)

struct Mod { u64 q; u64 half; u32 bits; u32 d; u64 clo, chi; };     // q = 2^bits - d; (chi, clo) = a multiple of q above 2^125 (makes V positive)

// ---- pack: canonical residue -> 7 balanced base-256 digits of its centred representative
__device__ __forceinline__ void digits7(u64 r, const Mod &m, signed char out[NPL])
{
    long long v = r > m.half ? (long long)r - (long long)m.q : (long long)r;
#pragma unroll
    for (int l = 0; l < NPL; l++) { const int d = (int)(signed char)(v & 0xff); out[l] = (signed char)d; v = (v - d) >> 8; }
}

// x: [B][ZD*NPOS cts][2][k][n] canonical (NTT form)  ->  Xp.  One thread per (b, ct, poly, residue, slot); lanes run over slots (coalesced reads).
__global__ void __launch_bounds__(256) pack_x_kernel(const u64 *x, signed char *xp, const Mod *mods, int n, int k, int B)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)B * ZD * NPOS * 2 * k;
    if (e >= rows * n) return;
    const int s = (int)(e % n); size_t r = e / n;
    const int i = (int)(r % k); r /= k; const int c = (int)(r % 2); r /= 2;
    const int pos = (int)(r % NPOS); r /= NPOS; const int z = (int)(r % ZD); const int b = (int)(r / ZD);
    signed char d[NPL]; digits7(x[e], mods[i], d);
    signed char *dst = xp + (((size_t)i * n + s) * B + b) * IMG_BYTES + ((size_t)pos * 2 + c) * ZD + z;
#pragma unroll
    for (int l = 0; l < NPL; l++) dst[(size_t)l * NPOS * 2 * ZD] = d[l];
}
// w: [NF][ZD][WF][WF][k][n] canonical (NTT form)  ->  Wp
__global__ void __launch_bounds__(256) pack_w_kernel(const u64 *w, signed char *wp, const Mod *mods, int n, int k)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)NF * ZD * TAPS * k;
    if (e >= rows * n) return;
    const int s = (int)(e % n); size_t r = e / n;
    const int i = (int)(r % k); r /= k;
    const int tap = (int)(r % TAPS); r /= TAPS; const int z = (int)(r % ZD); const int f = (int)(r / ZD);
    signed char d[NPL]; digits7(w[e], mods[i], d);
    signed char *dst = wp + ((size_t)i * n + s) * WSLOT_BYTES + (size_t)tap * TILE_BYTES + (size_t)f * ZD + z;
#pragma unroll
    for (int l = 0; l < NPL; l++) dst[(size_t)l * NF * ZD] = d[l];
}
// Ys [k][n][B][NF][16][2] -> y [B][NF][16][2][k][n]  (the product's tensor layout)
__global__ void __launch_bounds__(256) unpack_y_kernel(const u64 *ys, u64 *y, int n, int k, int B)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows = (size_t)B * NF * 16 * 2 * k;
    if (e >= rows * n) return;
    const int s = (int)(e % n); size_t r = e / n;
    const int i = (int)(r % k); r /= k;                  // r = ((b*NF + f)*16 + p)*2 + c
    const size_t b = r / (NF * 32), rest = r % (NF * 32);
    y[e] = ys[(((size_t)i * n + s) * B + b) * (NF * 32) + rest];
}

// V = sum_d D_d 2^(8d) (signed)  ->  V mod q, canonical
__device__ __forceinline__ u64 reduce_diagonals(const i32 (&D)[13], const Mod &m)
{
    // four signed 64-bit groups of four diagonals: G_g = D_4g + D_4g+1 2^8 + D_4g+2 2^16 + D_4g+3 2^24   (|G| < 2^52)
    long long G[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        long long a = D[4 * g];
        if (4 * g + 1 < 13) a += (long long)D[4 * g + 1] * 256;
        if (4 * g + 2 < 13) a += (long long)D[4 * g + 2] * 65536;
        if (4 * g + 3 < 13) a += (long long)D[4 * g + 3] * 16777216;
        G[g] = a;
    }
    // V = G0 + G1 2^32 + G2 2^64 + G3 2^96 as a two's-complement 128-bit value, plus the positive multiple of q
    u64 lo = (u64)G[0], hi = (u64)(G[0] >> 63);                   // sign extension
    { const u64 t = (u64)G[1] << 32; const u64 nl = lo + t; hi += (u64)(G[1] >> 32) + (nl < lo); lo = nl; }
    hi += (u64)G[2] + ((u64)G[3] << 32);
    { const u64 nl = lo + m.clo; hi += m.chi + (nl < lo); lo = nl; }
    // fold three times (q = 2^b - d, d < 2^26, 52 <= b <= 62): the same reduction as the product's fold128
    const u32 b = m.bits; const u64 d = m.d, mask = ((u64)1 << b) - 1;
    const u64 h1l = (lo >> b) | (hi << (64 - b)), h1h = hi >> b;
    u64 pl = h1l * d, ph = __umul64hi(h1l, d);
    u64 x1l = pl + (lo & mask), x1h = ph + h1h * d + (x1l < pl);
    const u64 h2 = (x1l >> b) | (x1h << (64 - b));
    pl = h2 * d; ph = __umul64hi(h2, d);
    u64 x2l = pl + (x1l & mask), x2h = ph + (x2l < pl);
    const u64 h3 = (x2l >> b) | (x2h << (64 - b));
    u64 r = h3 * d + (x2l & mask);
    return r >= m.q ? r - m.q : r;
}

// grid: one workgroup per (residue, slot, image pair); 256 threads = 4 waves (wm, wn) in a 2 x 2 arrangement of 32 x 32 tiles
// MODE (ablation, wrong results): 1 = no operand loads after the first three steps, 2 = no MFMAs, 3 = no reduction mod q in the epilogue
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mfma_conv_kernel(const signed char *xp, const signed char *wp, u64 *ys, const Mod *mods, int n, int k, int B)
{
    constexpr int mode = MODE;
    extern __shared__ __attribute__((aligned(16))) signed char lds[];          // ring of 4 x (A tile | W tile) = 4 x 28672 B
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    const int pairs = B / 2, slots = n * k;
    // XCD-aware decode (workgroups are dealt round-robin over the 8 XCDs): the image pairs of one slot run on one XCD at about the same time,
    // so the slot's 504 KiB of weight limbs are fetched from HBM once and then served by that XCD's L2
    int g = blockIdx.x, slot, bp;
    if ((slots & 7) == 0) { const int xcd = g & 7, r = g >> 3; slot = xcd * (slots >> 3) + r / pairs; bp = r % pairs; }
    else { slot = g / pairs; bp = g % pairs; }
    const int i = slot / n;
    const Mod m = mods[i];
    const signed char *ximg = xp + ((size_t)slot * B + 2 * bp) * IMG_BYTES;       // two consecutive image blocks
    const signed char *wsl = wp + (size_t)slot * WSLOT_BYTES;

    // staging: 28 LDS-DMA pieces of 1 KiB per step (pieces 0..13 of A, 14..27 of W), 7 per wave (pieces wave, wave+4, ...); a piece = 64 lanes x 16 B,
    // landing lane-linear.  A piece c16 (0..895) -> (plane, row, half): row = (image, pixel, poly)
    constexpr int NST = 4;                                  // LDS ring: the loads run three reduction steps ahead of their use
    u32 src_off[7];                                         // this lane's source offset per piece (A: inside `ximg`, without the tap term; W: inside the tap's tile)
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const int pc = wave + 4 * j;
        if (pc < 14) {
            const int c16 = pc * 64 + lane;
            const int plane = c16 >> 7, row = (c16 >> 1) & 63, half = c16 & 1;
            const int img = row >> 5, p = (row >> 1) & 15, c = row & 1;
            const int ox = p >> 2, oy = p & 3;
            src_off[j] = (u32)(img * IMG_BYTES + plane * (NPOS * 2 * ZD) + (((ox * STR) * XD + oy * STR) * 2 + c) * ZD + half * 16);
        } else src_off[j] = (u32)((pc - 14) * 1024 + lane * 16);
    }
    auto issue = [&](int tap) {
        const int kx = tap / WF, ky = tap % WF;
        const u32 tapoff = (u32)((kx * XD + ky) * 2 * ZD);
        signed char *dst = lds + (tap % NST) * (2 * TILE_BYTES);
        const signed char *wt = wsl + (size_t)tap * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int pc = wave + 4 * j;
            const signed char *src = pc < 14 ? ximg + src_off[j] + tapoff : wt + src_off[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
        }
    };

    v16i acc[13];
#pragma unroll
    for (int d = 0; d < 13; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[d][e] = 0;

    issue(0); issue(1); issue(2);
    const int fragA = (wm * 32 + (lane & 31)) * ZD + (lane >> 5) * 16, fragW = (wn * 32 + (lane & 31)) * ZD + (lane >> 5) * 16;
    for (int tap = 0; tap < TAPS; tap++) {
        // this wave's pieces of step `tap` have landed (two younger steps = 14 loads may still be in flight) ...
        if (mode == 1) __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        else if (tap + 2 < TAPS) __builtin_amdgcn_s_waitcnt(14 | (7 << 4) | (15 << 8));
        else if (tap + 1 < TAPS) __builtin_amdgcn_s_waitcnt(7 | (7 << 4) | (15 << 8));
        else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        __syncthreads();                                          // ... and everybody's; ring slot (tap + 3) % 4 = the one read in step tap - 1 is free
        if (tap + 3 < TAPS && mode != 1) issue(tap + 3);
        const signed char *tA = lds + (tap % NST) * (2 * TILE_BYTES), *tW = tA + TILE_BYTES;
        v4i w[NPL];
#pragma unroll
        for (int l = 0; l < NPL; l++) w[l] = *reinterpret_cast<const v4i *>(tW + l * (64 * ZD) + fragW);
        // 49 limb products; the seven MFMAs of one A plane go to seven different diagonals (no back-to-back dependent accumulators)
#pragma unroll
        for (int l = 0; l < NPL; l++) {
            const v4i a = *reinterpret_cast<const v4i *>(tA + l * (64 * ZD) + fragA);
            if (mode == 2) { acc[l][0] += a[0] + w[l][0]; continue; }
#pragma unroll
            for (int mm = 0; mm < NPL; mm++)
                acc[l + mm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w[mm], acc[l + mm], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of the 32 x 32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int f = wn * 32 + (lane & 31);
    u64 *dst = ys + ((size_t)slot * B + 2 * bp + wm) * (NF * 32) + (size_t)f * 32;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        i32 D[13];
#pragma unroll
        for (int d = 0; d < 13; d++) D[d] = acc[d][reg];
        if (mode == 3) { u64 sum = 0; for (int d = 0; d < 13; d++) sum += (u64)(u32)D[d] << d; dst[row] = sum; }     // keeps every accumulator alive, drops the reduction
        else dst[row] = reduce_diagonals(D, m);
        __builtin_amdgcn_sched_barrier(0);                 // one output at a time: sixteen interleaved reductions would spill
    }
}

// Variant with the workgroup's two image blocks RESIDENT in LDS (129 KiB) and only the weight tile streamed (2-slot ring, 28 KiB): every A byte crosses L2 -> LDS once
// instead of once per tap that touches it (~9x), halving the per-step LDS-DMA fill that bounds the streaming kernel.  A fragments are gathered from the resident blocks.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mfma_conv_resident_kernel(const signed char *xp, const signed char *wp, u64 *ys, const Mod *mods, int n, int k, int B)
{
    extern __shared__ __attribute__((aligned(16))) signed char lds[];          // [2 image blocks][2 x W tile]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    const int pairs = B / 2, slots = n * k;
    int g = blockIdx.x, slot, bp;
    if ((slots & 7) == 0) { const int xcd = g & 7, r = g >> 3; slot = xcd * (slots >> 3) + r / pairs; bp = r % pairs; }
    else { slot = g / pairs; bp = g % pairs; }
    const int i = slot / n;
    const Mod m = mods[i];
    const signed char *ximg = xp + ((size_t)slot * B + 2 * bp) * IMG_BYTES;
    const signed char *wsl = wp + (size_t)slot * WSLOT_BYTES;
    signed char *ldsW = lds + 2 * IMG_BYTES;
    constexpr int APIECES = 2 * IMG_BYTES / 1024;            // 126
    // W pieces of a step: 14, waves take pieces wave, wave + 4, ... (4, 4, 3, 3)
    auto issue_w = [&](int tap) {
        signed char *dst = ldsW + (tap & 1) * TILE_BYTES;
        const signed char *wt = wsl + (size_t)tap * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int pc = wave + 4 * j;
            if (pc < 14)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wt + pc * 1024 + lane * 16), (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
        }
    };
    for (int pc = wave; pc < APIECES; pc += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ximg + pc * 1024 + lane * 16), (__attribute__((address_space(3))) void *)(lds + pc * 1024), 16, 0, 0);
    issue_w(0);

    v16i acc[13];
#pragma unroll
    for (int d = 0; d < 13; d++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[d][e] = 0;

    const int r = lane & 31, p = r >> 1, c = r & 1, ox = p >> 2, oy = p & 3;
    const int fragA = wm * IMG_BYTES + (((ox * STR) * XD + oy * STR) * 2 + c) * ZD + (lane >> 5) * 16;
    const int fragW = (wn * 32 + (lane & 31)) * ZD + (lane >> 5) * 16;
    for (int tap = 0; tap < TAPS; tap++) {
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));    // this wave's pieces of step `tap` (and, at tap 0, of the image blocks) have landed ...
        __syncthreads();                                          // ... and everybody's; the other W slot, read in step tap - 1, is free
        if (tap + 1 < TAPS) issue_w(tap + 1);
        const int kx = tap / WF, ky = tap % WF;
        const signed char *tA = lds + fragA + (kx * XD + ky) * 2 * ZD, *tW = ldsW + (tap & 1) * TILE_BYTES;
        v4i w[NPL];
#pragma unroll
        for (int l = 0; l < NPL; l++) w[l] = *reinterpret_cast<const v4i *>(tW + l * (64 * ZD) + fragW);
#pragma unroll
        for (int l = 0; l < NPL; l++) {
            const v4i a = *reinterpret_cast<const v4i *>(tA + l * (NPOS * 2 * ZD));
#pragma unroll
            for (int mm = 0; mm < NPL; mm++)
                acc[l + mm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w[mm], acc[l + mm], 0, 0, 0);
        }
    }
    const int f = wn * 32 + (lane & 31);
    u64 *dst = ys + ((size_t)slot * B + 2 * bp + wm) * (NF * 32) + (size_t)f * 32;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        i32 D[13];
#pragma unroll
        for (int d = 0; d < 13; d++) D[d] = acc[d][reg];
        dst[row] = reduce_diagonals(D, m);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- host entry points (ctypes): device pointers, default stream ------------------------------------------------------------------
static int set_mods(Mod *d_mods, const u64 *q, int k)
{
    Mod h[8];
    for (int i = 0; i < k; i++) {
        u32 bits = 64 - __builtin_clzll(q[i]);
        h[i].q = q[i]; h[i].half = q[i] >> 1; h[i].bits = bits; h[i].d = (u32)(((u64)1 << bits) - q[i]);
        const unsigned __int128 c = (unsigned __int128)q[i] << (126 - bits);           // in (2^125, 2^126): |V| < 2^125 always
        h[i].clo = (u64)c; h[i].chi = (u64)(c >> 64);
    }
    return hipMemcpy(d_mods, h, sizeof(Mod) * k, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
extern "C" size_t mm_xp_bytes(int n, int k, int B) { return (size_t)n * k * B * IMG_BYTES; }
extern "C" size_t mm_wp_bytes(int n, int k) { return (size_t)n * k * WSLOT_BYTES; }
extern "C" size_t mm_ys_bytes(int n, int k, int B) { return (size_t)n * k * B * NF * 32 * 8; }
extern "C" int mm_pack_w(const u64 *w, signed char *wp, const u64 *q, int n, int k, void *d_mods)
{
    if (set_mods((Mod *)d_mods, q, k)) return -1;
    const size_t e = (size_t)NF * ZD * TAPS * k * n;
    hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, w, wp, (const Mod *)d_mods, n, k);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mm_pack_x(const u64 *x, signed char *xp, int n, int k, int B, void *d_mods)
{
    const size_t e = (size_t)B * ZD * NPOS * 2 * k * n;
    hipLaunchKernelGGL(pack_x_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, x, xp, (const Mod *)d_mods, n, k, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mm_conv(const signed char *xp, const signed char *wp, u64 *ys, int n, int k, int B, void *d_mods, int mode)
{
    if (B % 2) return -3;
    if (mode == 4) {
        const size_t ldsr = 2 * IMG_BYTES + 2 * TILE_BYTES;
        if (hipFuncSetAttribute((const void *)mfma_conv_resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr) != hipSuccess) return -4;
        hipLaunchKernelGGL(mfma_conv_resident_kernel, dim3((unsigned)((size_t)n * k * (B / 2))), dim3(256), ldsr, 0, xp, wp, ys, (const Mod *)d_mods, n, k, B);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const size_t lds = 8 * TILE_BYTES;                    // 4-slot ring of (A tile | W tile): 112 KiB
    auto kern = mode == 1 ? mfma_conv_kernel<1> : mode == 2 ? mfma_conv_kernel<2> : mode == 3 ? mfma_conv_kernel<3> : mfma_conv_kernel<0>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -4;
    const size_t grid = (size_t)n * k * (B / 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, 0, xp, wp, ys, (const Mod *)d_mods, n, k, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int mm_unpack_y(const u64 *ys, u64 *y, int n, int k, int B)
{
    const size_t e = (size_t)B * NF * 32 * k * n;
    hipLaunchKernelGGL(unpack_y_kernel, dim3((unsigned)((e + 255) / 256)), dim3(256), 0, 0, ys, y, n, k, B);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
