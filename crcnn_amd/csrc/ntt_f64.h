// ntt_f64.h -- negacyclic row transforms modulo the engine's own fp64 primes (f64mod.h) on an LDS image: the device functions shared by relinearisation's key
// switching (kernels_relin64.hip) and the auxiliary-base half of the ciphertext square (kernels_square64.hip).
#pragma once
#include "kernels.h"
#include "ntt_device.h"

typedef double2 d2;

// Two neighbouring points of a row (even s, s + 1) sit in ONE aligned 16-byte slot of the swizzled image -- every swizzle only XORs higher index bits into bit
// 0 -- in either order: rows move between memory and the image in 16-byte pieces per lane (one global access and one ds_*_b128 for two points; 8-byte accesses
// ran the base conversion kernels at 2.6 TB/s against 4.1 with 16-byte ones, profiles/r03_square_relin_*)
template <int SW> __device__ __forceinline__ void sm_store_pair(double *sm, int s, double x, double y)
{
    const int a = swz<SW>(s);
    *reinterpret_cast<d2 *>(sm + (a & ~1)) = (a & 1) ? d2{y, x} : d2{x, y};
}
template <int SW> __device__ __forceinline__ d2 sm_load_pair(const double *sm, int s)
{
    const int a = swz<SW>(s);
    const d2 v = *reinterpret_cast<const d2 *>(sm + (a & ~1));
    return (a & 1) ? d2{v.y, v.x} : v;
}

// ---- fp64 butterflies (the index arithmetic of ntt_device.h's fwd_stages / inv_stages; arithmetic of f64mod.h) ----------------------------------------------
// forward: values grow by at most 0.875 p per stage: 16-bit inputs stay below 14 p < 2^51 through 15 stages -- no reduction anywhere.
// The 2^R - 1 twiddles of a thread's R stages are fetched up front (tw[(1 << st) - 1 + j] = twiddle j of stage st), in front of the LDS reads of the pass: one
// exposed memory latency per pass instead of one per stage (the compiler keeps loads where they are written and waits right in front of the first use).
// A twiddle is ONE double (the centred power of psi): the quotient of a butterfly product is estimated from the product itself (f64_mulmod: fl(h / p) instead
// of y (w / p) -- the same six flops, one more link in the dependency chain), not from a precomputed companion w / p.  Half the table bytes and, what matters
// more, 7 instead of 14 live twiddle registers per radix-8 pass: the registers returned hold a row across its transforms (relinearisation's source row, the
// square's a / b rows, the first prime's result in front of the CRT step) instead of re-reading it from memory or parking it there.
template <int R>
__device__ __forceinline__ void load_tw_fwd(double (&tw)[(1 << R) - 1], const double *W, int m, int blk)
{
#pragma unroll
    for (int st = 0; st < R; st++)
#pragma unroll
        for (int j = 0; j < (1 << st); j++) tw[(1 << st) - 1 + j] = W[(m << st) + (blk << st) + j];
}
template <int R>
__device__ __forceinline__ void load_tw_inv(double (&tw)[(1 << R) - 1], const double *W, int h, int blk)
{
#pragma unroll
    for (int st = 0; st < R; st++)
#pragma unroll
        for (int j = 0; j < (1 << (R - 1 - st)); j++) tw[(1 << R) - (1 << (R - st)) + j] = W[(h >> st) + (blk << (R - 1 - st)) + j];
}
template <int R>
__device__ __forceinline__ void fwd_stages_f64(double (&v)[1 << R], const double (&tw)[(1 << R) - 1], const F64Mod md)
{
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << (R - 1 - st);
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const double X = v[c], T = f64_mulmod(tw[(1 << st) - 1 + (c >> (R - st))], v[c + half], md);
            v[c] = X + T; v[c + half] = X - T;
        }
    }
}
// inverse (Gentleman-Sande, no halving: n^-1 sits in the keys): sums double per stage, so a pass starts from reduced values (|x| <= p/2 -> below 4 p after
// three stages)
template <int R>
__device__ __forceinline__ void inv_stages_f64(double (&v)[1 << R], const double (&tw)[(1 << R) - 1], const F64Mod md)
{
#pragma unroll
    for (int st = 0; st < R; st++) {
        const int half = 1 << st;
#pragma unroll
        for (int c = 0; c < (1 << R); c++) {
            if (c & half) continue;
            const double U = v[c], V = v[c + half];
            v[c] = U + V; v[c + half] = f64_mulmod(tw[(1 << R) - (1 << (R - st)) + (c >> (st + 1))], U - V, md);
        }
    }
}
// s = 2^ls: element stride inside a group.  The swizzles are XORs of shifted index bits, i.e. linear over GF(2), and (c << ls) occupies bits that are zero in
// `base`: swz(base + c s) = swz(base) ^ swz(c s) -- one vector XOR per element against a wave-uniform constant instead of the whole index arithmetic
//
// Inverse passes and their reductions (round 5).  A Gentleman-Sande stage leaves a sum (magnitudes add) and a product (back below m = 0.875 p), so after the
// three
// stages of a radix-8 group whose inputs are bounded by B the outputs are bounded by [8 B, 4 m, 2 m, 2 m, m, m, m, m]: only TWO of the eight values grow.  With
// B = 1.75 p the largest difference a stage multiplies is 8 B = 14 p < 2^51 (f64_mulmod's operand range) and the largest sum 14 p < 2^53 (exact), and reducing
// just v[0] and v[1] on the way out restores the bound for the next pass: 6 instead of 24 reduction flops per group (120 -> 102 flops, -14 % measured on the
// whole transform: tools/f64_row_timeline.hip, profiles/r05_f64_row_timeline.txt).  reduce_in = true keeps the old form (every input reduced on load) for a
// first
// pass whose inputs may be lazy sums up to 2^52.4 and for the tail passes.
template <bool INV, int R, int RB>
__device__ __forceinline__ void ntt_pass_f64(double *sm, const double *W, int n, int ls, int tabidx, const F64Mod md, bool reduce_in)
{
    const unsigned groups = (unsigned)n >> R;
    for (unsigned g = threadIdx.x; g < groups; g += blockDim.x) {
        const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
        const int a0 = swz<RB>((int)((blk << (ls + R)) + l));
        double tw[(1 << R) - 1];
        if (INV) load_tw_inv<R>(tw, W, tabidx, (int)blk); else load_tw_fwd<R>(tw, W, tabidx, (int)blk);
        double v[1 << R];
#pragma unroll
        for (int c = 0; c < (1 << R); c++) { v[c] = sm[a0 ^ swz<RB>(c << ls)]; if (INV && reduce_in) v[c] = f64_reduce(v[c], md); }
        if (INV) inv_stages_f64<R>(v, tw, md); else fwd_stages_f64<R>(v, tw, md);
        if (INV && !reduce_in) { v[0] = f64_reduce(v[0], md); if (R >= 2) v[1] = f64_reduce(v[1], md); }
#pragma unroll
        for (int c = 0; c < (1 << R); c++) sm[a0 ^ swz<RB>(c << ls)] = v[c];
    }
    __syncthreads();
}
template <bool INV, int RB>
__device__ __forceinline__ void ntt_tail_pass_f64(int rem, double *sm, const double *W, int n, int ls, int tabidx, const F64Mod md)
{
    if (rem == 1) ntt_pass_f64<INV, 1, RB>(sm, W, n, ls, tabidx, md, INV);
    else if (rem == 2) ntt_pass_f64<INV, 2, RB>(sm, W, n, ls, tabidx, md, INV);
    else if (RB > 3 && rem == 3) ntt_pass_f64<INV, 3, RB>(sm, W, n, ls, tabidx, md, INV);
    else if (RB > 4 && rem == 4) ntt_pass_f64<INV, 4, RB>(sm, W, n, ls, tabidx, md, INV);
}
// The single leftover stage of a transform whose log2 n is not a multiple of RB plus one... is not a pass: when log2 n = RB m + 1 (n = 8192 at radix 8) the
// leftover stage is the one with gap 1 -- the last of a forward transform, the first of an inverse one -- and its two points are the two halves of ONE 16-byte
// slot of the image, i.e. exactly what a lane moves between memory and the image.  The loops that fill and drain the image apply it in registers (f64_stage_in
// / f64_stage_out below): four LDS passes and barriers per row instead of five.  Twiddle of the pair (s, s + 1): table index n/2 + s/2, both ways.
template <int RB> __device__ __forceinline__ bool f64_fused_stage(int logn) { return logn > RB && logn % RB == 1; }
// what goes INTO the image for points (s, s + 1) holding v.  Inverse with a fused first stage: inputs may be lazy sums below 2^52 -- reduced first, as a pass
// does on load
template <bool INV, int RB>
__device__ __forceinline__ d2 f64_stage_in(d2 v, const double *W, int n, int logn, int s, const F64Mod md)
{
    if (!INV || !f64_fused_stage<RB>(logn)) return v;
    const double U = f64_reduce(v.x, md), V = f64_reduce(v.y, md);
    return d2{U + V, f64_mulmod(W[(n >> 1) + (s >> 1)], U - V, md)};
}
// what comes OUT of the image for points (s, s + 1) holding v (forward with a fused last stage: values below 13 p in, below 14 p out -- unreduced, like a pass
// leaves them)
template <bool INV, int RB>
__device__ __forceinline__ d2 f64_stage_out(d2 v, const double *W, int n, int logn, int s, const F64Mod md)
{
    if (INV || !f64_fused_stage<RB>(logn)) return v;
    const double T = f64_mulmod(W[(n >> 1) + (s >> 1)], v.y, md);
    return d2{v.x + T, v.x - T};
}
// Draining the image (round 5).  A loop that reads a pair from the image, fetches the fused stage's twiddle, multiplies and stores runs one memory latency per
// iteration -- a quarter of a whole forward transform in the kernels that keep 16 points per thread (tools/f64_row_timeline.hip: 5200 of 20 400 cycles).  Here
// the LDS reads and twiddle loads of CH pairs are issued together before the first of them is used.  store(s, v) gets the pair (s, s + 1) after the fused
// stage.
template <bool INV, int RB, int CH, class F>
__device__ __forceinline__ void f64_drain(const double *sm, const double *W, int n, int logn, const F64Mod md, F &&store)
{
    const bool fused = !INV && f64_fused_stage<RB>(logn);
    const int step = 2 * (int)blockDim.x;
    for (int s0 = 2 * (int)threadIdx.x; s0 < n; s0 += step * CH) {
        d2 v[CH]; double tw[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int s = s0 + c * step;
            if (s < n) { v[c] = sm_load_pair<RB>(sm, s); if (fused) tw[c] = W[(n >> 1) + (s >> 1)]; }
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int s = s0 + c * step;
            if (s < n) {
                if (fused) { const double T = f64_mulmod(tw[c], v[c].y, md); v[c] = d2{v[c].x + T, v[c].x - T}; }
                store(s, v[c]);
            }
        }
    }
}
// ... and filling it for an inverse transform from pairs the caller provides: load(s) returns the pair (s, s + 1) (a global load, a product of held values);
// the fused first stage's twiddles are fetched beside the caller's loads
template <int RB, int CH, class F>
__device__ __forceinline__ void f64_fill_inv(double *sm, const double *W, int n, int logn, const F64Mod md, F &&load)
{
    const bool fused = f64_fused_stage<RB>(logn);
    const int step = 2 * (int)blockDim.x;
    for (int s0 = 2 * (int)threadIdx.x; s0 < n; s0 += step * CH) {
        d2 v[CH]; double tw[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int s = s0 + c * step;
            if (s < n) { v[c] = load(s); if (fused) tw[c] = W[(n >> 1) + (s >> 1)]; }
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const int s = s0 + c * step;
            if (s < n) {
                if (fused) { const double U = f64_reduce(v[c].x, md), V = f64_reduce(v[c].y, md); v[c] = d2{U + V, f64_mulmod(tw[c], U - V, md)}; }
                sm_store_pair<RB>(sm, s, v[c].x, v[c].y);
            }
        }
    }
}
// all LDS passes of one row on the image (swizzled); the caller has filled it through f64_stage_in and synchronised, the function returns synchronised, the
// caller drains it through f64_stage_out.  Inverse: the image holds values below 2^52 (lazy sums of up to 48 products) when first_reduce is set, below 1.75 p
// otherwise; what comes out is below 14 p (the caller reduces while it drains). (RB = stages per pass: 2^RB values per thread in registers between two LDS
// round trips.  The fp64 butterfly is a third of the 64-bit integer one's issue cycles, so the LDS passes, their barriers and the twiddle loads weigh more here
// than in ntt_device.h: fewer, wider passes)
template <bool INV, int RB>
__device__ __forceinline__ void ntt_row_passes_f64(double *sm, const double *W, int n, int logn, const F64Mod md, bool first_reduce = true)
{
    const bool fused = f64_fused_stage<RB>(logn);
    const int full = logn / RB, rem = fused ? 0 : logn - RB * full;
    if (!INV) {
        // gaps n/2, n/4, ...: a pass of R stages starting at gap 2^lt works on groups of stride 2^(lt - R + 1); twiddle block index n / 2^(lt + 1)
        int lt = logn - 1;
        for (int p = 0; p < full; p++, lt -= RB) ntt_pass_f64<false, RB, RB>(sm, W, n, lt - RB + 1, n >> (lt + 1), md, false);
        if (rem) ntt_tail_pass_f64<false, RB>(rem, sm, W, n, lt - rem + 1, n >> (lt + 1), md);
    } else {
        // radix-8 passes reduce lazily (two outputs per group, see ntt_pass_f64): the image must hold values below 1.75 p when the first pass starts -- callers
        // whose image holds lazy sums (relin_inv_crt_kernel without a fused first stage) ask for first_reduce, which reduces every input of the first pass
        // instead. The wider passes keep the reduction on load: their sums grow past the multiplier's operand range within a pass.
        int lt = fused ? 1 : 0;
        for (int p = 0; p < full; p++, lt += RB) ntt_pass_f64<true, RB, RB>(sm, W, n, lt, n >> (lt + 1), md, RB != 3 || (p == 0 && first_reduce));
        if (rem) ntt_tail_pass_f64<true, RB>(rem, sm, W, n, lt, n >> (lt + 1), md);
    }
}

// ---- ONE workgroup barrier per transform: wave-local passes (round 5) ---------------------------------------------------------------------------------------
// A workgroup of n / 16 threads has n / 1024 waves; give wave w the 1024-point block [1024 w, 1024 w + 1024).  Only the CS = log2 n - 10 stages with gaps of
// 1024 and more cross the blocks (round 6: also CS = 2, n = 4096 -- the ring of the reference's published ApproxPlainModel run).  A thread that owns, at every block offset c 1024, the same position(s) inside the block -- the PAIR (2 t, 2 t + 1) for n =
// 8192 (CS = 3: 8 offsets x 2 points), the single point t for n = 16384 (CS = 4: 16 offsets) -- runs those stages in registers, straight from (forward) or to
// (inverse) memory: the fill and the first pass, or the last pass and the drain, are one step without an LDS round trip.  The other ten stages never leave a
// block: three radix-8 passes in which wave w works on the 128 groups of ITS block (in every pass with a gap below 1024 the block of group g is g >> 7) and the
// gap-1 stage, applied while the block is filled (inverse) or drained (forward).  A wave's LDS operations execute in order, so these passes need no workgroup
// barrier: the image is synchronised ONCE per transform (between the cross pass and the local ones) and once more before it is reused, instead of five to six
// times -- the waves of a workgroup stop waiting for the slowest of them after every pass (a quarter of a transform's cycles: tools/f64_row_timeline.hip,
// profiles/r05_f64_row_timeline_one_barrier.txt: 29.5 -> 26.4 ns forward, 30.4 -> 25.1 ns inverse, same results bit for bit).
//   cross layout of a thread's 16 points:  CS = 3: v[2 c + e] = point 2 t + e + 1024 c;  CS = 4: v[c] = point t + 1024 c
//   block-local layout (fill / drain):     pair u of a thread = points 1024 w + 2 lane + 128 u and the next one, u < 8
#define CRC_F64_BLOCK 1024
__device__ __forceinline__ bool f64_wave_geometry(int n, int logn) { return (logn == 12 || logn == 13 || logn == 14) && (int)blockDim.x * 16 == n; }
__device__ __forceinline__ void f64_wave_sync()
{
    // the compiler must not move this wave's LDS reads above its own earlier LDS writes (the hardware never does)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (round 6: CS = 2, n = 4096 on 256 threads: v[4 c + e] = point 4 t + e + 1024 c -- four block offsets x four points, two 16-byte accesses per offset)
template <int CS> __device__ __forceinline__ int f64_cross_point(int i)
{
    return CS == 2 ? 4 * (int)threadIdx.x + (i & 3) + CRC_F64_BLOCK * (i >> 2)
         : CS == 3 ? 2 * (int)threadIdx.x + (i & 1) + CRC_F64_BLOCK * (i >> 1) : (int)threadIdx.x + CRC_F64_BLOCK * i;
}
__device__ __forceinline__ int f64_local_pair(int u) { return CRC_F64_BLOCK * (int)(threadIdx.x >> 6) + 2 * (int)(threadIdx.x & 63) + 128 * u; }
// a row of doubles in memory <-> the cross layout (16-byte accesses for CS = 3, 8-byte ones -- 512 contiguous bytes per wave instruction -- for CS = 4)
template <int CS> __device__ __forceinline__ void f64_cross_load(const double *row, double (&v)[16])
{
    if (CS == 2) {
#pragma unroll
        for (int h = 0; h < 8; h++) { const d2 x = *reinterpret_cast<const d2 *>(row + f64_cross_point<2>(2 * h)); v[2 * h] = x.x; v[2 * h + 1] = x.y; }
    } else if (CS == 3) {
#pragma unroll
        for (int c = 0; c < 8; c++) { const d2 x = *reinterpret_cast<const d2 *>(row + f64_cross_point<3>(2 * c)); v[2 * c] = x.x; v[2 * c + 1] = x.y; }
    } else {
#pragma unroll
        for (int c = 0; c < 16; c++) v[c] = row[f64_cross_point<4>(c)];
    }
}
template <int CS> __device__ __forceinline__ void f64_cross_store(double *row, const double (&v)[16])
{
    if (CS == 2) {
#pragma unroll
        for (int h = 0; h < 8; h++) *reinterpret_cast<d2 *>(row + f64_cross_point<2>(2 * h)) = d2{v[2 * h], v[2 * h + 1]};
    } else if (CS == 3) {
#pragma unroll
        for (int c = 0; c < 8; c++) *reinterpret_cast<d2 *>(row + f64_cross_point<3>(2 * c)) = d2{v[2 * c], v[2 * c + 1]};
    } else {
#pragma unroll
        for (int c = 0; c < 16; c++) row[f64_cross_point<4>(c)] = v[c];
    }
}
// the CS stages that cross the blocks, in registers, between a generator / consumer of the cross layout's registers and the image (8-byte LDS accesses: a
// pair's two points belong to different radix-8 groups here, and keeping both groups alive for a 16-byte access costs 16 registers the row-holding kernels do
// not have). Every group has block index 0 in these stages: the 2^CS - 1 twiddles are the same for the whole grid. Forward (the first stages of the transform,
// gaps n/2 ... 1024): get(i) -> register i of the cross layout; results go to the image. Inverse (its last ones): inputs from the image, below 1.75 p (the lazy
// local passes leave that) -- the radix-16 form reduces them first: four stages of sums would carry a 28 p difference into a multiplier that takes 16 p --;
// put(i, value) receives register i, unreduced, below 14 p. (a value every lane holds alike, moved to scalar registers: the cross passes' twiddles are the same
// for the whole grid, and 14 / 30 vector registers of them are what the row-holding kernels spill)
__device__ __forceinline__ double f64_uniform(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
template <int CS, int RB, class G>
__device__ __forceinline__ void f64_cross_fwd_to_image(double *sm, const double *W, const F64Mod md, G &&get)
{
    double tw[(1 << CS) - 1];
    load_tw_fwd<CS>(tw, W, 1, 0);
#pragma unroll
    for (int i = 0; i < (1 << CS) - 1; i++) tw[i] = f64_uniform(tw[i]);
    if constexpr (CS == 2) {
        // two radix-4 groups at a time: the points (4 t + 2 h, 4 t + 2 h + 1) at the four block offsets -- a pair per offset, stored with ONE 16-byte access
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double x0[4], x1[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { x0[c] = get(4 * c + 2 * h); x1[c] = get(4 * c + 2 * h + 1); }
            fwd_stages_f64<2>(x0, tw, md); fwd_stages_f64<2>(x1, tw, md);
#pragma unroll
            for (int c = 0; c < 4; c++) sm_store_pair<RB>(sm, f64_cross_point<2>(4 * c + 2 * h), x0[c], x1[c]);
        }
    } else if constexpr (CS == 3) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            double x[8];
#pragma unroll
            for (int c = 0; c < 8; c++) x[c] = get(2 * c + e);
            fwd_stages_f64<3>(x, tw, md);
#pragma unroll
            for (int c = 0; c < 8; c++) sm[swz<RB>(f64_cross_point<3>(2 * c + e))] = x[c];
        }
    } else {
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; c++) x[c] = get(c);
        fwd_stages_f64<4>(x, tw, md);
#pragma unroll
        for (int c = 0; c < 16; c++) sm[swz<RB>(f64_cross_point<4>(c))] = x[c];
    }
}
template <int CS, int RB, class P>
__device__ __forceinline__ void f64_cross_inv_from_image(const double *sm, const double *W, int n, const F64Mod md, P &&put)
{
    double tw[(1 << CS) - 1];
    load_tw_inv<CS>(tw, W, n >> 11, 0);                        // the first of these stages has gap 1024: table index n / 2048
#pragma unroll
    for (int i = 0; i < (1 << CS) - 1; i++) tw[i] = f64_uniform(tw[i]);
    if constexpr (CS == 2) {
        // (two stages of sums on inputs below 1.75 p: differences below 7 p, inside the multiplier's range -- no reduction on the way in)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double x0[4], x1[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { const d2 v = sm_load_pair<RB>(sm, f64_cross_point<2>(4 * c + 2 * h)); x0[c] = v.x; x1[c] = v.y; }
            inv_stages_f64<2>(x0, tw, md); inv_stages_f64<2>(x1, tw, md);
#pragma unroll
            for (int c = 0; c < 4; c++) { put(4 * c + 2 * h, x0[c]); put(4 * c + 2 * h + 1, x1[c]); }
        }
    } else if constexpr (CS == 3) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            double x[8];
#pragma unroll
            for (int c = 0; c < 8; c++) x[c] = sm[swz<RB>(f64_cross_point<3>(2 * c + e))];
            inv_stages_f64<3>(x, tw, md);
#pragma unroll
            for (int c = 0; c < 8; c++) put(2 * c + e, x[c]);
        }
    } else {
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; c++) x[c] = f64_reduce(sm[swz<RB>(f64_cross_point<4>(c))], md);
        inv_stages_f64<4>(x, tw, md);
#pragma unroll
        for (int c = 0; c < 16; c++) put(c, x[c]);
    }
}
// the three wave-local radix-8 passes (nine stages: gaps 512 ... 2 forward, 2 ... 512 inverse) on this wave's block of the image
template <bool INV, int RB>
__device__ __forceinline__ void f64_local_passes(double *sm, const double *W, int n, const F64Mod md)
{
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int p = 0; p < 3; p++) {
        const int lt = INV ? 1 + 3 * p : 9 - 3 * p, ls = INV ? lt : lt - 2, tabidx = n >> (lt + 1);
#pragma unroll 1
        for (unsigned u = 0; u < 2; u++) {
            const unsigned g = (w << 7) + lane + 64 * u;
            const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
            const int a0 = swz<RB>((int)((blk << (ls + 3)) + l));
            double tw[7], v[8];
            if (INV) load_tw_inv<3>(tw, W, tabidx, (int)blk); else load_tw_fwd<3>(tw, W, tabidx, (int)blk);
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = sm[a0 ^ swz<RB>(c << ls)];
            if (INV) { inv_stages_f64<3>(v, tw, md); v[0] = f64_reduce(v[0], md); v[1] = f64_reduce(v[1], md); }      // lazy reduction (ntt_pass_f64)
            else fwd_stages_f64<3>(v, tw, md);
#pragma unroll
            for (int c = 0; c < 8; c++) sm[a0 ^ swz<RB>(c << ls)] = v[c];
        }
        f64_wave_sync();
    }
}
// forward: drain this wave's block through the gap-1 stage; store(s, pair) gets the unreduced pair (below 14 p).  No barrier in front: the block is the wave's
// own
template <int RB, class F>
__device__ __forceinline__ void f64_local_drain(const double *sm, const double *W, int n, const F64Mod md, F &&store)
{
#pragma unroll 1
    for (int h = 0; h < 2; h++) {                             // two batches of four pairs: eight at once cost the row-holding kernels their registers
        d2 v[4]; double tw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int s = f64_local_pair(4 * h + u); v[u] = sm_load_pair<RB>(sm, s); tw[u] = W[(n >> 1) + (s >> 1)]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const double T = f64_mulmod(tw[u], v[u].y, md); store(f64_local_pair(4 * h + u), d2{v[u].x + T, v[u].x - T}); }
    }
}
// inverse: fill this wave's block through the gap-1 stage; load(u, s) returns the pair (s, s + 1) = pair u of the thread, any values below 2^52 (reduced here).
// Ends with the wave-level fence the local passes need, no barrier
template <int RB, class F>
__device__ __forceinline__ void f64_local_fill(double *sm, const double *W, int n, const F64Mod md, F &&load)
{
#pragma unroll
    for (int h = 0; h < 2; h++) {                             // (unrolled: load(u, .) indexes the caller's registers)
        d2 v[4]; double tw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int s = f64_local_pair(4 * h + u); v[u] = load(4 * h + u, s); tw[u] = W[(n >> 1) + (s >> 1)]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double U = f64_reduce(v[u].x, md), V = f64_reduce(v[u].y, md);
            sm_store_pair<RB>(sm, f64_local_pair(4 * h + u), U + V, f64_mulmod(tw[u], U - V, md));
        }
    }
    f64_wave_sync();
}
// whole transforms on top of these.  FORWARD: get(i) (register i of the cross layout; any values below 2^35 or so, e.g. digits) -> the image, transformed; the
// caller drains with f64_local_drain and must __syncthreads() before the image is written again.
template <int CS, int RB, class G>
__device__ __forceinline__ void f64_wave_forward(double *sm, const double *W, int n, const F64Mod md, G &&get)
{
    f64_cross_fwd_to_image<CS, RB>(sm, W, md, get);
    __syncthreads();
    f64_local_passes<false, RB>(sm, W, n, md);
}
// INVERSE after f64_local_fill: local passes, the barrier, the cross pass -> put(i, value) for the 16 registers of the cross layout, unreduced (below 14 p).
// The caller must __syncthreads() before the image is written again.
template <int CS, int RB, class P>
__device__ __forceinline__ void f64_wave_inverse(double *sm, const double *W, int n, const F64Mod md, P &&put)
{
    f64_local_passes<true, RB>(sm, W, n, md);
    __syncthreads();
    f64_cross_inv_from_image<CS, RB>(sm, W, n, md, put);
}

// ---- the same scheme for the 64-bit transforms over a coefficient modulus (ntt_device.h's lazy butterflies, the block ownership above) ---------------------
// the three wave-local radix-8 passes of a FORWARD transform (gaps 512 ... 2) on this wave's 1024-point block of the image
template <bool LAZY>
__device__ __forceinline__ void u64_local_passes_fwd(u64 *sm, const ulonglong2 *W, int n, u64 q, u64 q2)
{
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int p = 0; p < 3; p++) {
        const int lt = 9 - 3 * p, ls = lt - 2, tabidx = n >> (lt + 1);
#pragma unroll 1
        for (unsigned u = 0; u < 2; u++) {
            const unsigned g = (w << 7) + lane + 64 * u;
            const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
            const int a0 = swz<3>((int)((blk << (ls + 3)) + l));
            u64 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = sm[a0 ^ swz<3>(c << ls)];
            fwd_stages<3, LAZY>(v, W, tabidx, (int)blk, q, q2);
#pragma unroll
            for (int c = 0; c < 8; c++) sm[a0 ^ swz<3>(c << ls)] = v[c];
        }
        f64_wave_sync();
    }
}
// ... and of an INVERSE one (gaps 2 ... 512; the gap-1 stage is applied while the block is filled)
template <bool LAZY>
__device__ __forceinline__ void u64_local_passes_inv(u64 *sm, const ulonglong2 *W, int n, u64 q, u64 q2)
{
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int p = 0; p < 3; p++) {
        const int ls = 1 + 3 * p, tabidx = n >> (ls + 1);
#pragma unroll 1
        for (unsigned u = 0; u < 2; u++) {
            const unsigned g = (w << 7) + lane + 64 * u;
            const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
            const int a0 = swz<3>((int)((blk << (ls + 3)) + l));
            u64 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = sm[a0 ^ swz<3>(c << ls)];
            inv_stages<3, LAZY>(v, W, tabidx, (int)blk, q, q2);
#pragma unroll
            for (int c = 0; c < 8; c++) sm[a0 ^ swz<3>(c << ls)] = v[c];
        }
        f64_wave_sync();
    }
}
// ... the same with the butterflies that do not halve (ntt_device.h inv_stages_unscaled): pass inputs below 16 q, the one sum-of-all value of every group of
// eight (below 128 q) reduced on the way out
__device__ __forceinline__ void u64_local_passes_inv_unscaled(u64 *sm, const ulonglong2 *W, int n, u64 q, u64 q2, float rq)
{
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int p = 0; p < 3; p++) {
        const int ls = 1 + 3 * p, tabidx = n >> (ls + 1);
#pragma unroll 1
        for (unsigned u = 0; u < 2; u++) {
            const unsigned g = (w << 7) + lane + 64 * u;
            const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
            const int a0 = swz<3>((int)((blk << (ls + 3)) + l));
            u64 v[8];
#pragma unroll
            for (int c = 0; c < 8; c++) v[c] = sm[a0 ^ swz<3>(c << ls)];
            inv_stages_unscaled<3>(v, W, tabidx, (int)blk, q, q << 6);
            v[0] = reduce_small(v[0], q, q2, rq);
#pragma unroll
            for (int c = 0; c < 8; c++) sm[a0 ^ swz<3>(c << ls)] = v[c];
        }
        f64_wave_sync();
    }
}
