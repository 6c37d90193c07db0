// kernels_mfma1.hip -- one-channel convolutions (CrCNN's conv1, alone or fused with its pooling layer: 6 x 6 / 7 x 7 windows over the 28 x 28 encrypted image,
// 32 / 20 filters) on the matrix cores.
//
// With one input channel the reduction is xf*yf = 36..49 terms.  As a generic limb GEMM (kernels_mfma.hip) that is two reduction steps per output tile and the
// per-output work -- recombining 13 diagonals and reducing mod q -- dominates; on the vector ALU (mac3_kernel) the layer runs at 3.5 T modmul/s and, once conv2
// and fc3 moved to the matrix cores, is a third of PlainModelTiny.  This kernel is built around the two things such a layer offers:
//   reuse     per slot the window taps are K = 8 rows x 8 columns = 64 (zero weights beyond the window): ONE v_mfma_i32_16x16x64_i8 per limb pair covers the
//             whole reduction.  The 7 weight fragments of a wave's 16 filters live in registers for the whole workgroup, and a workgroup walks ALL images of
//             the chunk for its slot (image b+1 arrives by LDS-DMA while image b is multiplied);
//   a cheap   the accumulators start at per-diagonal biases B_d (so every diagonal stays in [0, 2^24) and their sum is a multiple of q: no signs, no
//   epilogue  correction term), pairs of diagonals pack into 32-bit words without carries, the 128-bit value is two word vectors added once.  Round 4: the
//   value is
//             reduced by FOLDING (q = 2^b - f: three folds, 3 + 1 multiplies: limbred.h diag_fold_short_centred) instead of a Montgomery step (7 quarter-rate
//             multiplies: 112 of ~370 issue cycles per output), and the MFMA operands are swapped -- weights as the A operand, image windows as B -- so that a
//             lane's four
//             accumulator registers are FOUR CONSECUTIVE FILTERS of one output row: their digit bytes transpose in registers (v_perm_b32) into one dword per
//             limb plane,
//             stored with 7 ds_write_b32 per four outputs instead of 28 byte stores (the byte-wise staging was most of the 23 % of CU cycles lost to LDS bank
//             conflicts, profiles/r03_pmc_conv1_issue.json; the two polys' image blocks now sit 8 banks apart for the window reads).
// 8 or 12 waves per workgroup (2-3 per SIMD: one wave's epilogue overlaps another's MFMAs); a wave owns 16 filters and every (waves/2)-th 16-row tile.
//   per image  the limb image [plane][poly][row + 1 pad][32 columns] (13 KiB for 28 x 28) in LDS, double-buffered; a lane's A fragment is two 8-byte LDS reads
//   per
//               plane (window rows kx, kx+1): the three words around byte offset oy * stride, byte-aligned in registers;
//   output      either the limb tensor of a following convolution, staged in LDS and written out as one contiguous [plane][pixel][poly][32 channels] block
//               (no limb_pack_tensor pass in front of conv2), or slot-major u64 for the generic conversions. Exact integer arithmetic throughout: the same
//               element of Z_q, hence the same bits, as mac3_kernel and the reference (convolutionalLayer.cpp:56-93).
#include "kernels.h"
#include "limbred.h"
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef signed char i8;
#define NPL 7
#define MAXK 8                                   // coefficient moduli (kernel argument tables)

struct Conv1Args {
    const i8 *xr; const i8 *wl; u64 *ys; i8 *xl_out; const ModParams *mods; const u64 *bias;
    int n, k, B, Bout, b0, xd, yo, xs, ystr, P, F, mtiles;   // B images in this launch; the limb result is image b0 + b of Bout
    // per (slot, image): 7 planes x 2 polys x xd rows x 32 bytes (+ 8: the last window may read past its row), rounded up to 1 KiB
    unsigned img_stride, plane_bytes, poly_bytes;
    // limb output, bytes per image: 7 * P * 2 * 32, or the flat form's 7 * 2 * P * zdc rounded up to 16 (kernels_mfma.hip)
    unsigned out_img_bytes;
    // flat form (fewer than 32 filters = channels of the next convolution): channel bytes per position, else 0
    int out_zdc;
    // 16 < F <= 20: the second filter group has at most four filters -- its waves pack (filter, weight limb) into the 16 rows of
    int narrow;
                                                       // the MFMA's A operand: 14 MFMAs and ONE output per lane and tile instead of 49 and four
                                                       // (mfma_conv1_kernel)
    int acc0[MAXK][13];                                // initial value of the 13 diagonal accumulators, per modulus (conv1_tables)
    u32 qbits[MAXK], qfold[MAXK];                      // q = 2^qbits - qfold (limbred.h conv1_fold_ok)
};

// canonical residue -> 7 balanced base-256 digits of its centred representative
__device__ __forceinline__ void limb_digits1(u64 r, u64 q, int (&d)[NPL])
{
    long long v = r > (q >> 1) ? (long long)r - (long long)q : (long long)r;
#pragma unroll
    for (int l = 0; l < NPL; l++) { d[l] = (int)(signed char)(v & 0xff); v = (v - d[l]) >> 8; }
}

__global__ void __launch_bounds__(768) mfma_conv1_kernel(Conv1Args a)
{
    extern __shared__ __attribute__((aligned(16))) i8 lds[];                  // [2 image buffers][output staging]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int slot = blockIdx.x, i = slot / a.n, s = slot % a.n;
    const u64 q = a.mods[i].q;
    const u32 qbits = a.qbits[i], qfold = a.qfold[i];
    // wave roles.  Usually wave w takes filter group w & 1 and row tiles w >> 1, w >> 1 + waves / 2, ...  With a NARROW second group (a.narrow) its waves cost
    // less than half of the others', and waves w, w + 4, w + 8 share a SIMD: groups alternate in blocks of four waves so that every SIMD hosts both kinds
    const int g = lane >> 4, r16 = lane & 15;
    const bool narrow = a.narrow != 0;
    const int nt = narrow ? (wave >> 2) & 1 : wave & 1;
    const int tile0 = !narrow ? wave >> 1 : nt ? wave - 4 : (wave < 4 ? wave : wave - 4), tile_step = !narrow ? nwaves >> 1 : nt ? 4 : nwaves - 4;
    const bool nar = narrow && nt;                          // this wave runs the packed form
    // operand A of the MFMA = the weights: this lane's filter nt * 16 + r16, K group g (window rows 2g, 2g+1): resident for the whole workgroup
    v4i wv[NPL];
    {
        const i8 *ws = a.wl + (size_t)slot * (NPL * 32 * 64) + (nt * 16 + r16) * 64 + g * 16;
        if (!nar) {
#pragma unroll
            for (int l = 0; l < NPL; l++) wv[l] = *reinterpret_cast<const v4i *>(ws + l * (32 * 64));
        } else {
            // packed A operand: row r16 = 4 f + m <-> filter 16 + f, weight limb 4 grp + m (grp = 0, 1: wv[0], wv[1]; limb 7 does not exist: zero rows)
            const i8 *wp = a.wl + (size_t)slot * (NPL * 32 * 64) + (16 + (r16 >> 2)) * 64 + g * 16;
#pragma unroll
            for (int grp = 0; grp < 2; grp++) {
                const int pl = 4 * grp + (r16 & 3);
                wv[grp] = pl < NPL ? *reinterpret_cast<const v4i *>(wp + pl * (32 * 64)) : v4i{0, 0, 0, 0};
            }
#pragma unroll
            for (int l = 2; l < NPL; l++) wv[l] = v4i{0, 0, 0, 0};
        }
    }
    // the accumulators start at zero (an inline constant of the MFMAs); the per-diagonal biases join pair by pair in the reduction (limbred.h)
    u32 PB[7];
#pragma unroll
    for (int j = 0; j < 6; j++) PB[j] = (u32)a.acc0[i][2 * j] + ((u32)a.acc0[i][2 * j + 1] << 8);
    PB[6] = (u32)a.acc0[i][12];
    // output pixel -> (row, column) of the output map with one multiply: p < 1024, yo <= 32, so (p rcp) >> 16 is exact
    const u32 rcp_yo = (65536u + (u32)a.yo - 1) / (u32)a.yo;
    // C/D layout of a 16 x 16 tile: col = lane & 15, row = 4 (lane >> 4) + reg.  With the weights as A the ROW is the filter and the COLUMN the output row:
    // this lane owns output row (pixel, poly) r16 of every tile and filters f0 .. f0 + 3, four consecutive channels of the next layer.  Rows alternate poly 0 /
    // poly 1 and tiles start at multiples of 16, so a lane's poly is lane & 1 for good: the bias (poly 0 only), centred, is a per-lane constant
    const int f0 = nt * 16 + 4 * g;
    long long bvc[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int f = f0 + reg;
        const u64 bv = (a.bias && f < a.F && !(lane & 1)) ? a.bias[((size_t)f * a.k + i) * a.n + s] : 0;
        bvc[reg] = bv > (q >> 1) ? (long long)(bv - q) : (long long)bv;
    }
    // (packed form: the lane's ONE filter is 16 + g)
    long long bvn = 0;
    if (nar) {
        const int f = 16 + g;
        const u64 bv = (a.bias && f < a.F && !(lane & 1)) ? a.bias[((size_t)f * a.k + i) * a.n + s] : 0;
        bvn = bv > (q >> 1) ? (long long)(bv - q) : (long long)bv;
    }
    i8 *stage = lds + 2 * (size_t)a.img_stride;
    const i8 *ximg = a.xr + (size_t)slot * a.B * a.img_stride;
    const int pieces = a.img_stride / 1024;
    auto issue_img = [&](int b) {
        i8 *dst = lds + (b & 1) * (size_t)a.img_stride;
        const i8 *src = ximg + (size_t)b * a.img_stride + lane * 16;
        for (int pc = wave; pc < pieces; pc += nwaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + pc * 1024),
                (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
    };
    // (flat limb result: the bytes between the last position and the image's 16-byte end leave with every image -- zero, like the tensor the pack kernel makes)
    if (a.xl_out && a.out_zdc && threadIdx.x < 16) { const unsigned e = a.out_img_bytes - 16 + threadIdx.x;
        if (e >= (unsigned)(NPL * 2 * a.P * a.out_zdc)) stage[e] = 0; }
    issue_img(0);
    const int chan_bytes = a.out_zdc ? a.out_zdc : 32;          // channel bytes per (pixel, poly) position of a limb result
    for (int b = 0; b < a.B; b++) {
        // this wave's pieces of image b have landed (and its stores of image b-1's staging copy have left) ...
        __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        __syncthreads();                                          // ... everybody's; the other image buffer and the staging area are free
        if (b + 1 < a.B) issue_img(b + 1);
        const i8 *img = lds + (b & 1) * (size_t)a.img_stride;
        for (int mt = tile0; mt < a.mtiles; mt += tile_step) {
            v4i acc[13];
#pragma unroll
            for (int d = 0; d < 13; d++) acc[d] = v4i{0, 0, 0, 0};
            // operand B = this lane's output row: pixel p, poly c (rows past 2P re-read the last one and are never stored)
            const int mm = mt * 16 + r16, mrow = min(mm, 2 * a.P - 1), p = mrow >> 1, c = mrow & 1, ox = (int)(((u32)p * rcp_yo) >> 16), oy = p - ox * a.yo;
            // k = 16 g + j  <->  window row kx = 2 g + (j >> 3), window column ky = j & 7: per row the three words around the window's first column, cut to its
            // 8 bytes with a byte alignment (rows clamped, and columns past the window may belong to the next row: their weights are zero)
            const int r0 = min(ox * a.xs + 2 * g, a.xd - 1), r1 = min(ox * a.xs + 2 * g + 1, a.xd - 1);
            const int off = oy * a.ystr, sh = off & 3;
            const i8 *p0 = img + c * a.poly_bytes + (off & ~3) + r0 * 32, *p1 = img + c * a.poly_bytes + (off & ~3) + r1 * 32;
            // image limb plane l of this lane's window rows, cut to the window's 8 bytes
            auto window = [&](int l) {
                const u32 *q0 = reinterpret_cast<const u32 *>(p0 + l * a.plane_bytes), *q1 = reinterpret_cast<const u32 *>(p1 + l * a.plane_bytes);
                const u32 a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2];
                return v4i{(int)__builtin_amdgcn_alignbyte(a1, a0, sh), (int)__builtin_amdgcn_alignbyte(a2, a1, sh), (int)__builtin_amdgcn_alignbyte(b1, b0,
                    sh),
                           (int)__builtin_amdgcn_alignbyte(b2, b1, sh)};
            };
            if (nar) {
                // packed second filter group: acc[l] = image limb l x weight limbs 0-3, acc[7 + l] (hi6 for l = 6) x limbs 4-6, of filter 16 + g; register m =
                // limb
                v4i hi6;
#pragma unroll
                for (int l = 0; l < NPL; l++) {
                    const v4i av = window(l);
                    acc[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[0], av, v4i{0, 0, 0, 0}, 0, 0, 0);
                    if (l < 6) acc[7 + l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[1], av, v4i{0, 0, 0, 0}, 0, 0, 0);
                    else hi6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[1], av, v4i{0, 0, 0, 0}, 0, 0, 0);
                }
                // the 13 diagonals of this lane's one output: D_d = sum over l + m = d of (image limb l) x (weight limb m); then the same reduction as below
                int D[13];
#pragma unroll
                for (int d = 0; d < 13; d++) {
                    int sum = 0;
#pragma unroll
                    for (int l = 0; l < NPL; l++) {
                        const int m = d - l;
                        if (m < 0 || m >= NPL) continue;
                        sum += m < 4 ? acc[l][m] : (l < 6 ? acc[7 + l][m - 4] : hi6[m - 4]);
                    }
                    D[d] = sum;
                }
                const long long cvn = diag_fold_short_centred(D, q, qbits, qfold, bvn, PB);
                const int f = 16 + g;
                if (mm >= 2 * a.P) continue;
                if (a.xl_out) {
                    if (f >= chan_bytes) continue;
                    const u64 dg = f < a.F ? centred_digit_bytes(cvn) : 0;
                    i8 *sp; unsigned pstride;
                    if (a.out_zdc) { sp = stage + ((mm & 1) * a.P + (mm >> 1)) * a.out_zdc + f; pstride = 2 * a.P * a.out_zdc; }
                    else { sp = stage + mm * (NPL * 32) + f; pstride = 32; }
#pragma unroll
                    for (int l = 0; l < NPL; l++) sp[l * pstride] = (i8)(dg >> (8 * l));
                } else if (f < a.F) a.ys[(((size_t)slot * a.B + b) * a.F + f) * (2 * a.P) + mm] = (u64)(cvn + ((cvn >> 63) & (long long)q));
                continue;
            }
#pragma unroll
            for (int l = 0; l < NPL; l++) {
                const v4i av = window(l);
#pragma unroll
                for (int m2 = 0; m2 < NPL; m2++)
                    acc[l + m2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[m2], av, acc[l + m2], 0, 0, 0);
            }
            // epilogue: four consecutive filters of output row mm
            long long cv[4];
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int D[13];
#pragma unroll
                for (int d = 0; d < 13; d++) D[d] = acc[d][reg];
                // one pass from the diagonals to the centred representative, bias included
                cv[reg] = diag_fold_short_centred(D, q, qbits, qfold, bvc[reg], PB);
            }
            if (mm >= 2 * a.P) continue;
            // limb tensor of the next convolution, staged in LDS as it leaves: [row][plane][32 channels], or the flat form [plane][poly][pixel][zdc]
            if (a.xl_out) {
                if (f0 >= chan_bytes) continue;                                      // (flat form: channel padding is a multiple of 4, this group is past it)
                u32 lo[4], hi[4];
#pragma unroll
                for (int reg = 0; reg < 4; reg++) { const u64 dg = f0 + reg < a.F ? centred_digit_bytes(cv[reg]) : 0; lo[reg] = (u32)dg;
                    hi[reg] = (u32)(dg >> 32); }
                // 4 x 7 byte transpose: dword l = digit l of the four filters (v_perm_b32: result byte = selector byte picks from {second operand: 0-3, first:
                // 4-7})
                const u32 t0 = __builtin_amdgcn_perm(lo[1], lo[0], 0x05010400u), t1 = __builtin_amdgcn_perm(lo[1], lo[0], 0x07030602u);
                const u32 u0 = __builtin_amdgcn_perm(lo[3], lo[2], 0x05010400u), u1 = __builtin_amdgcn_perm(lo[3], lo[2], 0x07030602u);
                const u32 t2 = __builtin_amdgcn_perm(hi[1], hi[0], 0x05010400u), t3 = __builtin_amdgcn_perm(hi[1], hi[0], 0x07030602u);
                const u32 u2 = __builtin_amdgcn_perm(hi[3], hi[2], 0x05010400u), u3 = __builtin_amdgcn_perm(hi[3], hi[2], 0x07030602u);
                u32 pl[NPL];
                pl[0] = __builtin_amdgcn_perm(u0, t0, 0x05040100u); pl[1] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                pl[2] = __builtin_amdgcn_perm(u1, t1, 0x05040100u); pl[3] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                pl[4] = __builtin_amdgcn_perm(u2, t2, 0x05040100u); pl[5] = __builtin_amdgcn_perm(u2, t2, 0x07060302u);
                pl[6] = __builtin_amdgcn_perm(u3, t3, 0x05040100u);
                i8 *sp; unsigned pstride;
                if (a.out_zdc) { sp = stage + ((mm & 1) * a.P + (mm >> 1)) * a.out_zdc + f0; pstride = 2 * a.P * a.out_zdc; }
                else { sp = stage + mm * (NPL * 32) + f0; pstride = 32; }
#pragma unroll
                for (int l = 0; l < NPL; l++) *reinterpret_cast<u32 *>(sp + l * pstride) = pl[l];
            } else {
#pragma unroll
                for (int reg = 0; reg < 4; reg++)
                    // canonical
                    if (f0 + reg < a.F) a.ys[(((size_t)slot * a.B + b) * a.F + f0 + reg) * (2 * a.P) + mm] = (u64)(cv[reg] + ((cv[reg] >> 63) & (long long)q));
            }
        }
        if (a.xl_out) {
            __syncthreads();
            i8 *dst = a.xl_out + ((size_t)slot * a.Bout + a.b0 + b) * a.out_img_bytes;
            if (a.out_zdc) {              // the staging area IS the image's byte layout
                for (unsigned o = threadIdx.x; o < a.out_img_bytes / 16; o += blockDim.x)
                    *reinterpret_cast<uint4 *>(dst + (size_t)o * 16) = *reinterpret_cast<const uint4 *>(stage + (size_t)o * 16);
            } else {
                const unsigned per_plane = a.out_img_bytes / (NPL * 16);      // 16-byte pieces of one plane: (row, half)
                for (unsigned o = threadIdx.x; o < NPL * per_plane; o += blockDim.x) {
                    const unsigned l = o / per_plane, rem = o - l * per_plane;
                    *reinterpret_cast<uint4 *>(dst + (size_t)o * 16) = *reinterpret_cast<const uint4 *>(stage + (rem >> 1) * (NPL * 32) + l * 32 +
                        (rem & 1) * 16);
                }
            }
        }
    }
}

// ---- operand preparation -----------------------------------------------------------------------------------------------------------------
// NTT-form image x [B][xd*yd cts][2][k][n] (one channel; canonical or 28-bit packed) -> Xr [slot][B][plane][poly][row][32 columns] (yd <= 32, zero padded). A
// workgroup = 64 consecutive slots x (image, poly, group of RG rows): thread (slot lane, row q of the group) reads its row's columns -- lanes run over the
// slots, so every load is a coalesced 512-byte segment of one ciphertext row -- and stages the seven planes' 32 bytes in LDS; then the workgroup writes the
// staged block out so that the RG * 32 = 128 contiguous bytes a (slot, plane) owns in Xr leave as ONE full line from eight adjacent lanes.  (Round 2's form --
// every thread storing its own 16-byte pieces, neighbouring lanes 1.6 MB apart -- moved 26 GB at 1.7 TB/s; a forward transform that writes this layout itself
// would have to hold 32 columns of a row at once, 32 polynomials per workgroup: the image layout is slot-major because the convolution's workgroup walks one
// slot's image, the transform's row is slot-minor, and the transpose between them is this pass.)
#define RG 4
#define RSL 32
// (32 slots per workgroup, thread = (slot, row of the group, 16-column half): 28 KiB of staging and five workgroups per CU, so that the loads of one overlap
// the digit arithmetic and the stores of the others -- with 64 slots and two workgroups per CU the three ran one after the other: 2.5 TB/s, the same
// restructuring took the weight pack of kernels_mfma.hip from 1.0 to 3.7 TB/s)
__global__ void __launch_bounds__(256) limb_pack_rows1_kernel(const u64 *x, i8 *xr, const ModParams *mods, int n, int k, int B, int xd, int yd, int packed,
                                                              unsigned img_stride, unsigned plane_bytes, unsigned poly_bytes)
{
    __shared__ __attribute__((aligned(16))) i8 st[RSL * NPL * RG * 32];          // [slot][plane][row of the group][32 columns]
    const int sblocks = n / RSL;
    const int sb = blockIdx.x % (sblocks * k), i = sb / sblocks, s0 = (sb % sblocks) * RSL;
    const int rgs = (xd + RG - 1) / RG;
    size_t r = blockIdx.x / (sblocks * k);                       // (b*2 + c)*rgs + row group
    const int rg = (int)(r % rgs); r /= rgs; const int c = (int)(r % 2); const int b = (int)(r / 2);
    const u64 q = mods[i].q;
    const int lane = threadIdx.x & (RSL - 1), qrow = (threadIdx.x >> 5) & (RG - 1), h = threadIdx.x >> 7, row = rg * RG + qrow;
    {
        u32 pl[NPL][4];
#pragma unroll
        for (int l = 0; l < NPL; l++)
#pragma unroll
            for (int wv = 0; wv < 4; wv++) pl[l][wv] = 0;
        if (row < xd) {
            const u64 *src = x + ((((size_t)b * xd * yd + (size_t)row * yd + h * 16) * 2 + c) * k + i) * (size_t)n + s0 + lane;
#pragma unroll
            for (int colx = 0; colx < 16; colx++)
                if (h * 16 + colx < yd) {
                    u64 v = src[(size_t)colx * 2 * k * n];
                    if (packed) v = (v & 0xffffffffULL) | ((v >> 32) << 28);
                    const u64 dg = balanced_digit_bytes(v, q);         // the 7 balanced digits, one per byte
#pragma unroll
                    for (int l = 0; l < NPL; l++) pl[l][colx >> 2] |= (u32)((dg >> (8 * l)) & 0xff) << (8 * (colx & 3));
                }
        }
        i8 *sp = st + (size_t)lane * (NPL * RG * 32) + qrow * 32 + h * 16;
#pragma unroll
        for (int l = 0; l < NPL; l++) *reinterpret_cast<uint4 *>(sp + l * (RG * 32)) = make_uint4(pl[l][0], pl[l][1], pl[l][2], pl[l][3]);
    }
    __syncthreads();
    // RSL slots x 7 planes runs of RG * 32 bytes, 16 bytes per lane: eight adjacent lanes write one run (rows past xd of a ragged last group are not stored)
    const int pieces_per_run = RG * 2, rows_here = min(RG, xd - rg * RG);
    for (int o = threadIdx.x; o < RSL * NPL * pieces_per_run; o += 256) {
        const int run = o / pieces_per_run, part = o - run * pieces_per_run, sl = run / NPL, l = run - sl * NPL;
        if ((part >> 1) >= rows_here) continue;
        i8 *dst = xr + (((size_t)i * n + s0 + sl) * B + b) * img_stride + (size_t)l * plane_bytes + (size_t)c * poly_bytes + (size_t)(rg * RG) * 32 + part * 16;
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(st + (size_t)run * (RG * 32) + part * 16);
    }
}
// NTT-form weights w [F][1][xf][yf][k][n] -> Wl1 [slot][7 planes][32 filters][64 taps], tap = kx*8 + ky (pre-zeroed).  (No 2^64 factor since round 4: the
// kernel's reduction folds, it no longer divides by 2^64.)
__global__ void __launch_bounds__(64) limb_pack_w1_kernel(const u64 *w, i8 *wl, const ModParams *mods, int n, int k, int F, int xf, int yf)
{
    const int sblocks = n / 64;
    const int sb = blockIdx.x % (sblocks * k), i = sb / sblocks, s = (sb % sblocks) * 64 + threadIdx.x;
    size_t r = blockIdx.x / (sblocks * k);                       // (f*xf + kx)*yf + ky
    const int ky = (int)(r % yf); r /= yf; const int kx = (int)(r % xf); const int f = (int)(r / xf);
    const ModParams m = mods[i];
    int d[NPL]; limb_digits1(w[((((size_t)f * xf + kx) * yf + ky) * k + i) * (size_t)n + s], m.q, d);
    i8 *dst = wl + ((size_t)i * n + s) * (NPL * 32 * 64) + f * 64 + kx * 8 + ky;
#pragma unroll
    for (int l = 0; l < NPL; l++) dst[(size_t)l * (32 * 64)] = (i8)d[l];
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------------------
// per (slot, image): 7 planes x 2 polys x (xd rows + 1) x 32 bytes: the extra row puts the two polys' blocks 8 LDS banks apart (xd = 28: 896-byte blocks would
// be 224 dwords = 0 banks apart, and the window reads of a pixel's two polys -- neighbouring lanes -- a two-way conflict each); it also takes the last window's
// read-ahead
static inline unsigned conv1_poly_bytes(int xd) { return ((unsigned)xd + 1) * 32; }
static inline unsigned conv1_img_stride(int xd) { const unsigned b = NPL * 2 * conv1_poly_bytes(xd); return (b + 1023) / 1024 * 1024; }
bool k_limb_conv1_shape(const crc_ctx *c, int zd, int xd, int yd, int xs, int ys_, int xf, int yf, int nf)
{
    if (zd != 1 || xf > 8 || yf > 8 || nf > 32 || yd > 32 || c->n < 64 || c->k > MAXK) return false;
    // the epilogue's bounds: the folding reduction wants q = 2^b - f with 53 <= b <= 55 and a small f (limbred.h conv1_fold_ok); 7 balanced digits with |top
    // digit| <= 64 need q < 2^55
    for (int i = 0; i < c->k; i++) if (!conv1_fold_ok(c->tabs[i].m.q, c->tabs[i].m.bits, fold_constant(c->tabs[i].m.q, c->tabs[i].m.bits))) return false;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys_ + 1;
    if (xo * yo > 1024) return false;                          // (the kernel's reciprocal division of a pixel index by yo <= 32 is exact below 2048)
    // two image buffers + the staging area of a limb result (at most 32 channel bytes per position) must fit the 160 KiB of LDS
    return 2 * (size_t)conv1_img_stride(xd) + (size_t)NPL * xo * yo * 2 * 32 <= 160 * 1024;
}
size_t k_limb_conv1_weights_bytes(const crc_ctx *c) { return (size_t)c->n * c->k * NPL * 32 * 64; }
size_t k_limb_conv1_image_bytes(const crc_ctx *c, int B, int xd) { return (size_t)c->n * c->k * B * conv1_img_stride(xd); }

static void conv1_tables(const crc_ctx *c, Conv1Args &a)      // limbred.h: accumulator biases and q^-1 mod 2^64 per modulus
{
    for (int i = 0; i < c->k; i++) { conv1_bias_table(c->tabs[i].m.q, a.acc0[i]); a.qbits[i] = c->tabs[i].m.bits;
        a.qfold[i] = fold_constant(c->tabs[i].m.q, c->tabs[i].m.bits); }
}

int k_limb_conv1_pack_weights(crc_ctx *c, const u64 *w, i8 *wl, int nf, int xf, int yf, hipStream_t st)
{
    HIPCHK(hipMemsetAsync(wl, 0, k_limb_conv1_weights_bytes(c), st));
    const size_t blocks = (size_t)(c->n / 64) * c->k * nf * xf * yf;
    hipLaunchKernelGGL(limb_pack_w1_kernel, dim3((unsigned)blocks), dim3(64), 0, st, w, wl, c->d_mods, c->n, c->k, nf, xf, yf);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
// x: B NTT-form one-channel images; xr: k_limb_conv1_image_bytes of scratch; result either images b0 .. b0 + B of a limb tensor of Bout images (xl_out,
// [slot][Bout][7][P][2][32]) or slot-major u64 (ys, [slot][B][F][P][2])
int k_limb_conv1(crc_ctx *c, const u64 *x, bool packed, i8 *xr, const i8 *wl, u64 *ys, i8 *xl_out, int Bout, int b0, const u64 *bias_ntt, int B, int xd,
    int yd, int xs, int ys_,
                 int xf, int yf, int nf, hipStream_t st)
{
    if (B == 0) return CRC_OK;
    if (!k_limb_conv1_shape(c, 1, xd, yd, xs, ys_, xf, yf, nf)) return CRC_ERR_UNSUPPORTED;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys_ + 1;
    Conv1Args a{};
    a.xr = xr; a.wl = wl; a.ys = ys; a.xl_out = xl_out; a.mods = c->d_mods; a.bias = bias_ntt;
    a.n = c->n; a.k = c->k; a.B = B; a.Bout = Bout; a.b0 = b0; a.xd = xd; a.yo = yo; a.xs = xs; a.ystr = ys_; a.P = xo * yo; a.F = nf;
        a.mtiles = (2 * a.P + 15) / 16;
    a.poly_bytes = conv1_poly_bytes(xd); a.plane_bytes = 2 * a.poly_bytes; a.img_stride = conv1_img_stride(xd);
    // (the limb tensor of the convolution behind: its layout follows ITS channel count = this layer's filters)
    a.out_zdc = a.P > 1 ? k_limb_flat_zdc(nf) : 0;
    a.out_img_bytes = a.out_zdc ? (unsigned)((NPL * 2 * a.P * a.out_zdc + 15) / 16 * 16) : (unsigned)(NPL * a.P * 2 * 32);
    a.narrow = c->tune.conv1_narrow != 0 && nf > 16 && nf <= 20 ? 1 : 0;
    conv1_tables(c, a);
    {
        const size_t blocks = (size_t)(c->n / RSL) * c->k * B * 2 * ((xd + RG - 1) / RG);
        if (blocks > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(limb_pack_rows1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, xr, c->d_mods, c->n, c->k, B, xd, yd, packed ? 1 : 0,
            a.img_stride, a.plane_bytes,
                           a.poly_bytes);
        HIPCHK(hipGetLastError());
    }
    // waves per workgroup (8 or 12; wave w runs on SIMD w % 4 and takes row tiles w/2, w/2 + waves/2, ...): the count that loads the busiest SIMD least, 12 on
    // a tie (a third wave per SIMD hides more of the epilogue behind the other waves' MFMAs)
    auto busiest = [&](int nw) {
        int load[4] = {0, 0, 0, 0}, worst = 0;
        for (int w = 0; w < nw; w++) load[w & 3] += (a.mtiles - (w >> 1) + nw / 2 - 1) / (nw / 2);
        for (int sd = 0; sd < 4; sd++) worst = load[sd] > worst ? load[sd] : worst;
        return worst;
    };
    const int forced = c->tune.conv1_waves;     // tuning (tools/)
    // (narrow second group: four of its waves + four or eight of the others, one or two heavy waves and one light wave per SIMD either way: 12, for the
    // overlap)
    const int nwaves = forced ? forced : a.narrow ? 12 : busiest(12) <= busiest(8) ? 12 : 8;
    const size_t lds = 2 * (size_t)a.img_stride + (xl_out ? a.out_img_bytes : 0);
    { const int rc = crc_ctx_ensure_lds(c, (const void *)mfma_conv1_kernel, lds); if (rc) return rc; }
    hipLaunchKernelGGL(mfma_conv1_kernel, dim3((unsigned)(c->n * c->k)), dim3(64 * nwaves), lds, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
