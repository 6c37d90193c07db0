// ntt_device.h -- device-side pieces of the row NTT (ntt_rows_kernel in kernels.hip): LDS swizzle, lazy Shoup multiplication,
// register-resident butterfly stages, one LDS pass, all passes of a row.
#pragma once
#include "modarith.h"
#include <hip/hip_runtime.h>

// Workgroups that read the same rows -- the kf transforms that lift one polynomial, the three products of one (ciphertext, modulus) pair -- should share an L2.
// Blocks b and b + 8 run on the same XCD (dispatch deals blocks round-robin over the eight XCDs: observed, not promised, and only speed depends on it), so the
// G members of a group take consecutive slots of ONE XCD: the rows come from memory once and from that XCD's L2 afterwards.  Launch xcd_grid(groups, G) blocks
// (kernels.h); blocks past the last group return at once.
__device__ __forceinline__ bool xcd_group(unsigned b, unsigned G, size_t groups, size_t &grp, unsigned &member)
{
    const unsigned xcd = b & 7u, slot = b >> 3;
    member = slot % G; grp = (size_t)(slot / G) * 8 + xcd;
    return grp < groups;
}

// LDS index swizzle (XOR, no padding): conflict-free ds_read/write_b64 for the contiguous staging accesses AND for every strided
// register-tile pattern of the radix-8 passes at n = 4096 (at most 2-way in the short tail pass of n = 8192 / 16384); found by
// enumerating the access patterns (bank = index mod 32 per 32-lane group)
#ifndef CRC_SWZ_KEEP_BIT0
__device__ __forceinline__ int lpad(int i) { return i ^ ((i >> 3) & 7) ^ (((i >> 6) & 3) << 3); }
#else
// Experiment (round 6, -DCRC_SWZ_KEEP_BIT0): a swizzle that leaves bit 0 alone -- bits 5..7 into bits 1..3, bit 7 also into bit 4 -- so that the pair (s, s + 1), s
// even, always sits in its 16-byte slot in order and the four v_cndmask of every pair access fold away.  Conflict-free for the radix-8 passes with element strides
// 2, 16, 128 and 1024 (enumerated like the original); the 8-byte accesses of the wave-local kernels' cross passes (index 2 t + e) become two-way conflicts.
__device__ __forceinline__ int lpad(int i) { return i ^ (((i >> 5) & 7) << 1) ^ (((i >> 7) & 1) << 4); }
#endif
// the same idea for passes of 16 / 32 values per thread (the fp64 transforms of kernels_relin64.hip): XOR of higher index bits into the low four = the sixteen
// 8-byte slots of a 128-byte LDS row; found by enumerating every pass's access pattern in groups of 16 lanes (forward and inverse, n = 4096 / 8192 / 16384):
// conflict-free for their own radix, at most 2-way in one pass of the radix-8 transforms below when those run on an image laid out this way (s, s + 1), s even,
// share one aligned 16-byte slot of the image (the swizzle only XORs higher index bits into bit 0), in either order: rows move between memory and the image two
// points per lane -- one 16-byte global access and one ds_*_b128
__device__ __forceinline__ void sm_store_pair64(u64 *sm, int s, u64 x, u64 y)
{
    const int a = lpad(s);
    *reinterpret_cast<ulonglong2 *>(sm + (a & ~1)) = (a & 1) ? ulonglong2{y, x} : ulonglong2{x, y};
}
__device__ __forceinline__ ulonglong2 sm_load_pair64(const u64 *sm, int s)
{
    const int a = lpad(s);
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(sm + (a & ~1));
    return (a & 1) ? ulonglong2{v.y, v.x} : v;
}
__device__ __forceinline__ ulonglong2 ld2(const u64 *p) { return *reinterpret_cast<const ulonglong2 *>(p); }
__device__ __forceinline__ void st2(u64 *p, u64 x, u64 y) { *reinterpret_cast<ulonglong2 *>(p) = ulonglong2{x, y}; }
template <int SW> __device__ __forceinline__ int swz(int i);
template <> __device__ __forceinline__ int swz<3>(int i) { return lpad(i); }
template <> __device__ __forceinline__ int swz<4>(int i) { return i ^ ((i >> 2) & 1) ^ ((i >> 4) & 15); }
template <> __device__ __forceinline__ int swz<5>(int i) { return i ^ ((i >> 4) & 15) ^ ((i >> 5) & 15); }

// Lazy variant (LAZY = true, moduli of at most 57 bits): there are 7+ spare bits above q in a 64-bit word, so butterflies never
// correct their inputs and use a 3-product estimate of the Shoup quotient (result in [0, 4q) instead of [0, 2q)); the forward
// transform lets values grow by 4q per stage (<= (1 + 4*14) q < 2^63), the inverse one stays below 11q because every stage halves.
// One float-estimated reduction per coefficient at the end makes the output canonical, so the bits are the reference's.
__device__ __forceinline__ u64 shoup_lazy4(u64 a, u64 w, u64 wp, u64 q)
{
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
    const u64 h = (u64)a1 * p1 + __umulhi(a1, p0) + __umulhi(a0, p1);          // floor(a*wp / 2^64) - {0,1,2}
    return a * w - h * q;
}
// v < 128 q  ->  v mod q.  rq = 1 / float((q >> 32) + 1): the quotient estimate never exceeds floor(v/q) and is at most 2 short.
__device__ __forceinline__ u64 reduce_small(u64 v, u64 q, u64 q2, float rq)
{
    u32 est = (u32)((float)(u32)(v >> 32) * rq);
    est = est ? est - 1 : 0;
    v -= (u64)est * q;
    v = v >= q2 ? v - q2 : v;
    return v >= q ? v - q : v;
}

// R butterfly stages on 2^R register-resident values.  Forward (Cooley-Tukey): first stage pairs c with c + 2^(R-1);
// inverse (Gentleman-Sande): first stage pairs c with c + 1.
template <int R, bool LAZY>
__device__ __forceinline__ void fwd_stages(u64 (&v)[1 << R], const ulonglong2 *W, int m, int blk, u64 q, u64 q2)
{
    const u64 q4 = q2 + q2;
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << (R - 1 - st);
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const int wi = (m << st) + (blk << st) + (c >> (R - st));
            const ulonglong2 tw = W[wi]; const u64 w = tw.x, wp = tw.y;
            u64 X = v[c]; const u64 Y = v[c + half];
            if (LAZY) {
                const u64 Q = shoup_lazy4(Y, w, wp, q);
                v[c] = X + Q;
                v[c + half] = X + (q4 - Q);
            } else {
                X = X >= q2 ? X - q2 : X;
                const u64 Q = mulmod_shoup_lazy(Y, w, wp, q);
                v[c] = X + Q;
                v[c + half] = X + (q2 - Q);
            }
        }
    }
}
template <int R, bool LAZY>
__device__ __forceinline__ void inv_stages(u64 (&v)[1 << R], const ulonglong2 *W, int h, int blk, u64 q, u64 q2)
{
    const u64 q16 = q2 << 3;
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << st;
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const int wi = (h >> st) + (blk << (R - 1 - st)) + (c >> (st + 1));
            const ulonglong2 tw = W[wi]; const u64 w = tw.x, wp = tw.y;
            const u64 U = v[c], V = v[c + half];
            if (LAZY) {                                  // U, V < Bq  ->  U' < (B + 1/2) q, V' < 4q: never above (4 + 13/2) q < 16q
                const u64 T = q16 - V + U;
                const u64 cu = U + V;
                v[c] = (cu + ((cu & 1) ? q : 0)) >> 1;
                v[c + half] = shoup_lazy4(T, w, wp, q);
            } else {
                const u64 T = q2 - V + U;
                u64 cu = U + V; cu = cu >= q2 ? cu - q2 : cu;
                v[c] = (cu + ((cu & 1) ? q : 0)) >> 1;
                v[c + half] = mulmod_shoup_lazy(T, w, wp, q);
            }
        }
    }
}

// Inverse stages that do NOT halve (round 5; moduli below 2^55, tables of the plain inverse powers): the sum side of a Gentleman-Sande butterfly is U + V and
// nothing else, the factor n^-1 is one constant in the multiplication every result ends in anyway.  The sums double per stage instead: with inputs below B q a
// group of 2^R values leaves one value (index 0, the sum of all) below 2^R B q, R more below 2^(R-1) ... 16 q, the rest below 4 q; the caller reduces index 0
// when another pass follows.  kq >= the bound of any difference operand V (2^(R-1) B q); everything stays below 2^9 q < 2^64.  Six instructions fewer per
// butterfly.
template <int R>
__device__ __forceinline__ void inv_stages_unscaled(u64 (&v)[1 << R], const ulonglong2 *W, int h, int blk, u64 q, u64 kq)
{
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << st;
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const int wi = (h >> st) + (blk << (R - 1 - st)) + (c >> (st + 1));
            const ulonglong2 tw = W[wi];
            const u64 U = v[c], V = v[c + half];
            v[c] = U + V;
            v[c + half] = shoup_lazy4(kq - V + U, tw.x, tw.y, q);
        }
    }
}
__device__ __forceinline__ void inv_pair_stage_unscaled(ulonglong2 &v, const ulonglong2 tw, u64 q, u64 kq)
{
    const u64 U = v.x, V = v.y;
    v.x = U + V; v.y = shoup_lazy4(kq - V + U, tw.x, tw.y, q);
}

// The stage with gap 1 -- the last of a forward transform, the first of an inverse one -- pairs the two points (s, s + 1), s even, that share one 16-byte slot
// of the image, i.e. what one lane moves between memory and the image.  When log2 n = 3 m + 1 (n = 8192, and the halves of n = 16384) that stage would be an
// LDS pass of its own with one butterfly per thread; instead the loops that fill / drain the image apply it in registers (ntt_fused_stage: four passes and
// barriers per row instead of five).  Same butterflies in the same order on the same values, hence the same results.  Twiddle of the pair: table index n/2 +
// s/2 (both directions).
__device__ __forceinline__ bool ntt_fused_stage(int logn) { return logn > 3 && logn % 3 == 1; }
template <bool LAZY>
__device__ __forceinline__ void fwd_pair_stage(ulonglong2 &v, const ulonglong2 tw, u64 q, u64 q2)
{
    u64 X = v.x; const u64 Y = v.y;
    if (LAZY) { const u64 Q = shoup_lazy4(Y, tw.x, tw.y, q); v.x = X + Q; v.y = X + (q2 + q2 - Q); }
    else { X = X >= q2 ? X - q2 : X; const u64 Q = mulmod_shoup_lazy(Y, tw.x, tw.y, q); v.x = X + Q; v.y = X + (q2 - Q); }
}
template <bool LAZY>
__device__ __forceinline__ void inv_pair_stage(ulonglong2 &v, const ulonglong2 tw, u64 q, u64 q2)
{
    const u64 U = v.x, V = v.y;
    if (LAZY) { const u64 T = (q2 << 3) - V + U, cu = U + V; v.x = (cu + ((cu & 1) ? q : 0)) >> 1; v.y = shoup_lazy4(T, tw.x, tw.y, q); }
    else { const u64 T = q2 - V + U; u64 cu = U + V; cu = cu >= q2 ? cu - q2 : cu; v.x = (cu + ((cu & 1) ? q : 0)) >> 1; v.y = mulmod_shoup_lazy(T, tw.x, tw.y, q); }
}

// one pass over the whole row: every thread takes groups of 2^R values that interact in the next R stages
template <bool INV, int R, bool LAZY, int SW = 3>
__device__ __forceinline__ void ntt_pass(u64 *sm, const ulonglong2 *W, int n, int s /*element stride inside a group*/, int tabidx, u64 q, u64 q2)
{
    const int groups = n >> R;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        const int blk = g / s, l = g - blk * s;
        const int base = blk * (s << R) + l;
        u64 v[1 << R];
#pragma unroll
        for (int c = 0; c < (1 << R); c++) v[c] = sm[swz<SW>(base + c * s)];
        if (INV) inv_stages<R, LAZY>(v, W, tabidx, blk, q, q2); else fwd_stages<R, LAZY>(v, W, tabidx, blk, q, q2);
#pragma unroll
        for (int c = 0; c < (1 << R); c++) sm[swz<SW>(base + c * s)] = v[c];
    }
    __syncthreads();
}


// all passes of one row transform on the LDS image `sm` (n values, lpad-swizzled); the caller has synchronised after filling it and
// the function returns synchronised.  Forward: gaps n/2, n/4, ... (radix-8 passes, then a radix-4 / radix-2 pass when log2 n is not
// a multiple of 3); inverse: gaps 1, 2, 4, ...
// FUSE1: the caller applies the gap-1 stage itself while it fills (inverse) / drains (forward) the image whenever ntt_fused_stage(logn) says so
template <bool INV, bool LAZY, int SW = 3, bool FUSE1 = false>
__device__ __forceinline__ void ntt_row_passes(u64 *sm, const ulonglong2 *W, int n, int logn, u64 q, u64 q2)
{
    const bool fused = FUSE1 && ntt_fused_stage(logn);
    const int full = logn / 3, rem = fused ? 0 : logn - 3 * full;
    if (!INV) {
        int t = n >> 1;
        for (int p = 0; p < full; p++, t >>= 3) ntt_pass<false, 3, LAZY, SW>(sm, W, n, t >> 2, n / (2 * t), q, q2);
        if (rem == 2) ntt_pass<false, 2, LAZY, SW>(sm, W, n, t >> 1, n / (2 * t), q, q2);
        else if (rem == 1) ntt_pass<false, 1, LAZY, SW>(sm, W, n, t, n / (2 * t), q, q2);
    } else {
        int t = fused ? 2 : 1;
        for (int p = 0; p < full; p++, t <<= 3) ntt_pass<true, 3, LAZY, SW>(sm, W, n, t, n / (2 * t), q, q2);
        if (rem == 2) ntt_pass<true, 2, LAZY, SW>(sm, W, n, t, n / (2 * t), q, q2);
        else if (rem == 1) ntt_pass<true, 1, LAZY, SW>(sm, W, n, t, n / (2 * t), q, q2);
    }
}
