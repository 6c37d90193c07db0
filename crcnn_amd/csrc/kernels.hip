// kernels.hip -- gfx950 kernels of the encrypted-CNN evaluation path + their launchers.
//
// Data model: every tensor is an array of "rows"; a row is one polynomial residue = n uint64 in [0,q_mi).  A size-S
// ciphertext is S*k consecutive rows ([S][k][n]); row r of a ciphertext array belongs to modulus index r % k.
// Integer modular arithmetic only -- no MFMA.  Lane <-> coefficient/slot, so every global access is a coalesced
// 512 B (8 B/lane) or 1 KiB (16 B/lane) wave transaction.
#include "kernels.h"
#include <type_traits>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------------------
// NTT: one workgroup per row, whole polynomial staged in LDS (32 KiB @ n=4096 ... 128 KiB @ n=16384).
//   forward = Cooley-Tukey, natural -> bit-reversed, Harvey lazy butterflies   (reference: util/smallntt.cpp:195-273)
//   inverse = Gentleman-Sande, bit-reversed -> natural, n^-1 folded into the /2 twiddles   (smallntt.cpp:276-375)
// Output canonical.  Optional fused epilogue for the inverse transform: add/subtract a delta-form plaintext row to
// poly 0 (the add_plain(bias) of convolution3d, convolutionalLayer.cpp:87) and fused prologue for the forward one:
// plaintext lift (transform_to_ntt(Plaintext), evaluator.cpp:1465-1486) or delta scaling.
// ---------------------------------------------------------------------------------------------------------------
struct NttArgs {
    const u64 *src; u64 *dst;
    const ModParams *mods; const ulonglong2 *w;               // twiddle table [(k+kb)][n] of {w, Shoup(w)} (forward: root powers, inverse: psi^-i / 2)
    int n, logn;
    int mod_base, mod_count;                                   // row r -> modulus index mod_base + r % mod_count
    int src_rows_per_item;                                     // 0: src row = dst row;  >0: plain prologue, src row = r / mod_count
    size_t rows;                                               // total rows of the launch (prefetch variant)
    int pack_out;                                              // forward only: store the result as 28-bit limb pairs (operand form of the MAC kernels)
    // both > 0: dst row r = (ct, j < dst_ct_rows) reads src row ct*src_ct_rows + j (leading polys of wider cts)
    int src_ct_rows, dst_ct_rows;
    // 0 none, 1 plain lift, 2 delta scale, 3 relinearisation digit, 4 square products, 5 (inverse) scaled result
    int prologue;
    // prologue 3 (forward): row = ((ct*D + g)*k + j); the source row is the premultiplied third polynomial c2 (q/q_i)^-1 mod q_i of ciphertext ct under
    //   modulus i = dig_i[g] (src = size-`src_size` ciphertexts, poly `src_poly`); the value fed to the transform is its digit (v >> dig_shift[g]) & dig_mask
    //   (relinearize_one_step, evaluator.cpp:984-1001) -- the digit polynomials never exist in memory
    // prologue 4 (inverse): row = ((ct*3 + p)*mod_count + j); sources are the NTT-form rows a, b of polys 0 and 1 of ciphertext ct in `src`
    //   ([ct][2][mod_count][n]); the value fed to the transform is a^2, 2ab or b^2 (Evaluator::square's dyadic products, evaluator.cpp:798-852).
    //   The three products of one (ciphertext, modulus) pair run on ONE XCD, blockIdx -> row through xcd_group; `pairs` = ciphertexts x mod_count
    // prologue 5 (inverse): the result leaves multiplied by the per-modulus constant post_mul (Shoup companion post_mul_s) -- the square's lift wants
    //   x m~ (q/q_i)^-1 mod q_i (baseconverter.cpp:686-696), and the Shoup multiplication takes the lazy value in place of the final reduction
    int D, src_size, src_poly; unsigned long long dig_mask;
    size_t pairs;
    u64 post_mul[CRC_MAXK], post_mul_s[CRC_MAXK];
    // forward transforms of size-2 ciphertexts that leave as  NTT(row) + fma_u[ct][i] . fma_k[p][i]  (the device encryptor: noise rows + pk . NTT(u));
    // null: off
    const u64 *fma_u, *fma_k;
    unsigned long long fma_group;                              // fma_u null, fma_k set: the rows leave multiplied by the plaintext row fma_k[ct / fma_group][i]
    // host side only, prologue 4: 1 = post_mul is OFFERED -- a kernel that closes with a multiplication anyway takes it in (and says so: 2), the others
    // ignore it
    int opt_mul;
    unsigned char dig_i[48], dig_shift[48];
    const u64 *addend; int add_sign; int rows_per_ct; long long add_group;   // epilogue (inverse only)
    int add_mod;                                               // plaintext index = (ct / add_group) % add_mod (0: no modulo)
    int add_mode, add_size;                                    // 1: delta plaintext on poly 0 (shared by add_group cts); 2: rows of a size-add_size ct array
    PlainParams pp;
};

// transform_to_ntt(Plaintext)'s lift of a plaintext coefficient c < t to its residue mod q_i (evaluator.cpp:1447-1486): c, or c + (q - t) for the "negative"
// upper half.  With every q_i > t that is c + (q_i - t) without reduction (the fast branch); otherwise the reference adds the multi-word q - t and decomposes,
// i.e. (c mod q_i) + ((q - t) mod q_i) mod q_i -- the same residue
__device__ __forceinline__ u64 plain_lift(u64 c, const PlainParams &pp, int i, const ModParams &m)
{
    if (pp.fast) return c >= pp.threshold ? c + pp.inc[i] : c;
    const u64 r = barrett128(c, 0, m);
    return c >= pp.threshold ? addmod(r, pp.inc[i], m.q) : r;
}

#include "ntt_device.h"  // lpad, shoup_lazy4, reduce_small, fwd_stages / inv_stages, ntt_pass
#include "ntt_f64.h"     // the wave-local scheme: block ownership, cross layout, u64_local_passes_*
__device__ __forceinline__ u64 split28v(u64 v) { return (v & 0x0fffffffULL) | ((v >> 28) << 32); }      // = split28 further down

// PRO: load prologue compiled into this instance -- 0: a.prologue in {0 none, 1 plain lift, 2 delta scale}; 3: relinearisation digit; 4: square products
// (separate instances keep every variant at <= 64 VGPRs = two resident 1024-thread workgroups per CU at n = 8192)
template <bool INV, bool LAZY, int PRO>
__device__ __forceinline__ void ntt_rows_body(const NttArgs &a)
{
    extern __shared__ u64 sm[];
    const int n = a.n, logn = a.logn, tid = threadIdx.x, nt = blockDim.x;
    size_t row = blockIdx.x;
    if (INV && PRO == 4) {
        size_t pair; unsigned p;
        if (!xcd_group(blockIdx.x, 3, a.pairs, pair, p)) return;
        row = ((pair / a.mod_count) * 3 + p) * a.mod_count + pair % a.mod_count;
    }
    const int mloc = (int)(row % a.mod_count);
    const int mi = a.mod_base + mloc;
    const ModParams m = a.mods[mi];
    const u64 q = m.q, q2 = m.two_q;
    const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
    const ulonglong2 *W = a.w + (size_t)mi * n;
    u64 *dst = a.dst + row * (size_t)n;
    // the gap-1 stage is applied here, in the loops that fill (inverse) / drain (forward) the image, when it would otherwise be a pass of its own
    // (ntt_device.h)
    const bool fuse1 = ntt_fused_stage(logn);
    const ulonglong2 *W1 = W + (n >> 1);
    auto put = [&](int s, u64 v0, u64 v1) { ulonglong2 v{v0, v1}; if (INV && fuse1) inv_pair_stage<LAZY>(v, W1[s >> 1], q, q2);
        sm_store_pair64(sm, s, v.x, v.y); };
    if (INV && PRO == 4) {
        const size_t ct = row / (3 * (size_t)a.mod_count); const int p = (int)((row / a.mod_count) % 3);
        const u64 *pa = a.src + ((ct * 2 + (p == 2 ? 1 : 0)) * a.mod_count + mloc) * (size_t)n;
        const u64 *pb = a.src + ((ct * 2 + (p == 0 ? 0 : 1)) * a.mod_count + mloc) * (size_t)n;
        for (int s = 2 * tid; s < n; s += 2 * nt) {
            const ulonglong2 av = ld2(pa + s), bv = ld2(pb + s);
            u64 v0 = mulmod(av.x, bv.x, m), v1 = mulmod(av.y, bv.y, m);
            if (p == 1) { v0 = addmod(v0, v0, q); v1 = addmod(v1, v1, q); }
            put(s, v0, v1);
        }
    } else if (!INV && PRO == 3) {
        const size_t item = row / a.mod_count, ct = item / a.D; const int g = (int)(item % a.D);
        const u64 *src = a.src + ((ct * a.src_size + a.src_poly) * a.mod_count + a.dig_i[g]) * (size_t)n;
        const int sh = a.dig_shift[g];
        for (int s = 2 * tid; s < n; s += 2 * nt) { const ulonglong2 v = ld2(src + s); sm_store_pair64(sm, s, (v.x >> sh) & a.dig_mask,
            (v.y >> sh) & a.dig_mask); }
    } else {
        const size_t srow = a.dst_ct_rows ? (row / a.dst_ct_rows) * a.src_ct_rows + row % a.dst_ct_rows : (a.src_rows_per_item ? (row / a.mod_count) : row);
        const u64 *src = a.src + srow * (size_t)n;
        auto pro = [&](u64 v) -> u64 {
            if (a.prologue == 1) v = plain_lift(v, a.pp, mloc, m);
            else if (a.prologue == 2) {
                u64 lo, hi; mul64wide(a.pp.delta[mloc], v, lo, hi);
                if (v >= a.pp.threshold) { u64 l2 = lo + a.pp.uhi[mloc]; hi += (l2 < lo); lo = l2; }
                v = barrett128(lo, hi, m);
            }
            return v;
        };
        for (int s = 2 * tid; s < n; s += 2 * nt) { const ulonglong2 v = ld2(src + s); put(s, pro(v.x), pro(v.y)); }
    }
    __syncthreads();

    if (!INV) {
        // gaps n/2, n/4, ...: radix-8 passes first, a radix-4 / radix-2 pass finishes when log2(n) is not a multiple of 3
        ntt_row_passes<false, LAZY, 3, true>(sm, W, n, logn, q, q2);
        const u64 *add = a.addend ? a.addend + row * (size_t)n : nullptr;      // forward epilogue: + an NTT-form row of the same index
        auto fin = [&](u64 v) -> u64 {
            if (LAZY) return reduce_small(v, q, q2, rq);
            v = v >= q2 ? v - q2 : v; return v >= q ? v - q : v;
        };
        for (int s = 2 * tid; s < n; s += 2 * nt) {
            ulonglong2 v = sm_load_pair64(sm, s);
            if (fuse1) fwd_pair_stage<LAZY>(v, W1[s >> 1], q, q2);
            v.x = fin(v.x); v.y = fin(v.y);
            if (add) { const ulonglong2 ad = ld2(add + s); v.x = addmod(v.x, ad.x, q); v.y = addmod(v.y, ad.y, q); }
            if (a.pack_out) st2(dst + s, split28v(v.x), split28v(v.y)); else st2(dst + s, v.x, v.y);
        }
    } else {
        // gaps 1, 2, 4, ...
        ntt_row_passes<true, LAZY, 3, true>(sm, W, n, logn, q, q2);
        const u64 *add = nullptr;
        if (a.addend) {
            const size_t ct = row / a.rows_per_ct; const int p = (int)((row % a.rows_per_ct) / a.mod_count);
            if (a.add_mode == 2) add = a.addend + ((ct * a.add_size + p) * a.mod_count + mloc) * (size_t)n;
            else if (p == 0) { size_t g = ct / a.add_group; if (a.add_mod) g %= a.add_mod; add = a.addend + (g * a.mod_count + mloc) * (size_t)n; }
        }
        const u64 pm = PRO == 5 ? a.post_mul[mloc] : 0, pms = PRO == 5 ? a.post_mul_s[mloc] : 0;
        auto fin = [&](u64 v) -> u64 {
            if (PRO == 5) return mulmod_shoup(v, pm, pms, q);                 // (any 64-bit value in, canonical out)
            if (LAZY) return reduce_small(v, q, q2, rq);
            v = v >= q2 ? v - q2 : v; return v >= q ? v - q : v;
        };
        for (int s = 2 * tid; s < n; s += 2 * nt) {
            ulonglong2 v = sm_load_pair64(sm, s);
            v.x = fin(v.x); v.y = fin(v.y);
            if (add) {
                const ulonglong2 ad = ld2(add + s);
                if (a.add_sign > 0) { v.x = addmod(v.x, ad.x, q); v.y = addmod(v.y, ad.y, q); } else { v.x = submod(v.x, ad.x, q); v.y = submod(v.y, ad.y, q); }
            }
            st2(dst + s, v.x, v.y);
        }
    }
}

template <bool INV, bool LAZY, int PRO>
__global__ void __launch_bounds__(1024) ntt_rows_kernel(NttArgs a) { ntt_rows_body<INV, LAZY, PRO>(a); }

// ONE workgroup barrier per transform (round 5; ntt_f64.h: wave-local passes) for the lazy 64-bit transforms at n = 4096 / 8192 / 16384: n / 16 threads, wave w
// owns the 1024-point block w.  Forward: 16-byte loads in the cross layout -> the three cross stages in registers -> image | barrier | three wave-local passes
// -> block-local drain through the gap-1 stage.  Inverse: block-local fill through the gap-1 stage (PRO 4: the square's products formed on the way) -> three
// wave-local passes | barrier | the cross stages from the image to registers -> final reduction (PRO 5: the scaled result) -> 16-byte stores.  Same butterflies
// on the same values as ntt_rows_body, hence the same results.  Prologues 0 / 4 / 5 without an addend; everything else stays with ntt_rows_kernel.
template <bool INV, int PRO, int CS, bool UNS, int FMA = 0>
__global__ void __launch_bounds__(CS == 1 ? 128 : CS == 2 ? 256 : CS == 3 ? 512 : 1024, 4) ntt_rows_wave_kernel(NttArgs a)
{
    // CS = log2 n - 10 stages cross the 1024-point blocks (n = 2048 / 4096 / 8192 / 16384: 1 / 2 / 3 / 4; n = 2048 -- the ring of the reference's published
    // PlainModelTiny run -- since round 6); a thread owns E = 16 >> CS neighbouring points of every block
    constexpr int E = 16 >> CS, C = 1 << CS;
    extern __shared__ u64 sm[];
    const int n = a.n, tid = threadIdx.x;
    size_t row = blockIdx.x;
    if (INV && PRO == 4) {
        size_t pair; unsigned p;
        if (!xcd_group(blockIdx.x, 3, a.pairs, pair, p)) return;
        row = ((pair / a.mod_count) * 3 + p) * a.mod_count + pair % a.mod_count;
    }
    const int mloc = (int)(row % a.mod_count), mi = a.mod_base + mloc;
    const ModParams m = a.mods[mi];
    const u64 q = m.q, q2 = m.two_q;
    const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
    const ulonglong2 *W = a.w + (size_t)mi * n, *W1 = W + (n >> 1);
    u64 *dst = a.dst + row * (size_t)n;
    auto pt = [&](int c, int e) { return E * tid + e + 1024 * c; };          // point e of this thread in block c
    if (!INV) {
        const size_t srow = a.dst_ct_rows ? (row / a.dst_ct_rows) * a.src_ct_rows + row % a.dst_ct_rows : (a.src_rows_per_item ? (row / a.mod_count) : row);
        const u64 *src = a.src + srow * (size_t)n;
        u64 x[16];
#pragma unroll
        for (int c = 0; c < C; c++) {
            if constexpr (E == 1) x[c] = src[pt(c, 0)];
            else {
#pragma unroll
                for (int e = 0; e < E; e += 2) { const ulonglong2 v = ld2(src + pt(c, e)); x[c * E + e] = v.x; x[c * E + e + 1] = v.y; }
            }
        }
#pragma unroll
        for (int e = 0; e < E; e++) {
            u64 y[C];
#pragma unroll
            for (int c = 0; c < C; c++) y[c] = x[c * E + e];
            fwd_stages<CS, true>(y, W, 1, 0, q, q2);
#pragma unroll
            for (int c = 0; c < C; c++) sm[swz<3>(pt(c, e))] = y[c];
        }
        __syncthreads();
        const bool fma_lazy = m.fold != 0 && m.bits >= 53;
        u64_local_passes_fwd<true>(sm, W, n, q, q2);
#pragma unroll 2
        for (int u = 0; u < 8; u++) {
            const int s = f64_local_pair(u);
            ulonglong2 v = sm_load_pair64(sm, s);
            fwd_pair_stage<true>(v, W1[s >> 1], q, q2);
            if (FMA == 2) {     // . plaintext row (crc_multiply_plain: the dyadic product in the transform's last loop instead of a pass of its own)
                const size_t ctm = row / (2 * (size_t)a.mod_count);
                const ulonglong2 kv = ld2(a.fma_k + ((ctm / a.fma_group) * a.mod_count + mloc) * (size_t)n + s);
                v.x = mulmod(reduce_small(v.x, q, q2, rq), kv.x, m); v.y = mulmod(reduce_small(v.y, q, q2, rq), kv.y, m);
                st2(dst + s, v.x, v.y);
                continue;
            }
            if (FMA == 1) {     // + u . key, the product lazily (two folds: below 2q) into the value the one reduction takes anyway (below 60 q + 2 q < 128 q)
                const size_t ctm = row / (2 * (size_t)a.mod_count); const int pp = (int)((row / a.mod_count) & 1);
                const ulonglong2 uv = ld2(a.fma_u + (ctm * a.mod_count + mloc) * (size_t)n + s), kv = ld2(a.fma_k + ((size_t)pp * a.mod_count + mloc) * n + s);
                v.x += fma_lazy ? mulmod_fold2_lazy(uv.x, kv.x, m) : mulmod(uv.x, kv.x, m); v.y += fma_lazy ? mulmod_fold2_lazy(uv.y, kv.y, m) : mulmod(uv.y,
                    kv.y, m);
            }
            v.x = reduce_small(v.x, q, q2, rq); v.y = reduce_small(v.y, q, q2, rq);
            if (a.pack_out) st2(dst + s, split28v(v.x), split28v(v.y)); else st2(dst + s, v.x, v.y);
        }
    } else {
        if (PRO == 4) {
            const size_t ct = row / (3 * (size_t)a.mod_count); const int p = (int)((row / a.mod_count) % 3);
            const u64 *pa = a.src + ((ct * 2 + (p == 2 ? 1 : 0)) * a.mod_count + mloc) * (size_t)n;
            const u64 *pb = a.src + ((ct * 2 + (p == 0 ? 0 : 1)) * a.mod_count + mloc) * (size_t)n;
            const bool lazy_prod = m.fold != 0 && m.bits >= 53;
#pragma unroll 4
            for (int u = 0; u < 8; u++) {
                const int s = f64_local_pair(u);
                const ulonglong2 av = ld2(pa + s), bv = ld2(pb + s);
                ulonglong2 v;
                // below 2q each, 4q doubled: the unscaled butterflies take that (inv_stages_unscaled: sums below 8q, v0 of pass 1 below 64q)
                if (UNS && lazy_prod) {
                    v = ulonglong2{mulmod_fold2_lazy(av.x, bv.x, m), mulmod_fold2_lazy(av.y, bv.y, m)};
                    if (p == 1) { v.x += v.x; v.y += v.y; }
                } else {
                    v = ulonglong2{mulmod(av.x, bv.x, m), mulmod(av.y, bv.y, m)};
                    if (p == 1) { v.x = addmod(v.x, v.x, q); v.y = addmod(v.y, v.y, q); }
                }
                if (UNS) inv_pair_stage_unscaled(v, W1[s >> 1], q, q2 + q2); else inv_pair_stage<true>(v, W1[s >> 1], q, q2);
                sm_store_pair64(sm, s, v.x, v.y);
            }
        } else {
            const size_t srow = a.dst_ct_rows ? (row / a.dst_ct_rows) * a.src_ct_rows + row % a.dst_ct_rows : (a.src_rows_per_item ? (row / a.mod_count) : row);
            const u64 *src = a.src + srow * (size_t)n;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s = f64_local_pair(u);
                ulonglong2 v = ld2(src + s);
                if (UNS) inv_pair_stage_unscaled(v, W1[s >> 1], q, q2 + q2); else inv_pair_stage<true>(v, W1[s >> 1], q, q2);
                sm_store_pair64(sm, s, v.x, v.y);
            }
        }
        f64_wave_sync();
        if (UNS) u64_local_passes_inv_unscaled(sm, W, n, q, q2, rq); else u64_local_passes_inv<true>(sm, W, n, q, q2);
        __syncthreads();
        // (UNS: every prologue ends in the multiplication -- by n^-1, times the caller's constant at prologue 5; the host put the product into post_mul)
        const u64 pm = PRO == 5 || UNS ? a.post_mul[mloc] : 0, pms = PRO == 5 || UNS ? a.post_mul_s[mloc] : 0;
        u64 x[16];
#pragma unroll
        for (int e = 0; e < E; e++) {
            u64 y[C];
#pragma unroll
            for (int c = 0; c < C; c++) y[c] = sm[swz<3>(pt(c, e))];
            if (UNS) inv_stages_unscaled<CS>(y, W, n >> 11, 0, q, q << (CS + 3)); else inv_stages<CS, true>(y, W, n >> 11, 0, q, q2);
#pragma unroll
            for (int c = 0; c < C; c++) x[c * E + e] = PRO == 5 || UNS ? mulmod_shoup(y[c], pm, pms, q) : reduce_small(y[c], q, q2, rq);
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            if constexpr (E == 1) dst[pt(c, 0)] = x[c];
            else {
#pragma unroll
                for (int e = 0; e < E; e += 2) st2(dst + pt(c, e), x[c * E + e], x[c * E + e + 1]);
            }
        }
    }
}

// n = 16384: a whole row is 128 KiB of LDS = ONE resident workgroup per CU, and nothing overlaps its loads, barriers and stores.  Split form: the row as two
// halves of n/2 values that go through the SAME 64-KiB image one after the other (two workgroups per CU again).
//   forward  stage 0 pairs j with j + n/2 (one twiddle, W[1]): it is applied while the half is staged -- half h takes X + W Y (h = 0) or X - W Y (h = 1), both
//   source
//            values read for each half (the second time from L2) -- and the remaining stages are the transform of n/2 points whose twiddle block index is (2 +
//            h) m
//            instead of m (global block I = h m' + i' of a stage with 2 m' blocks sits at table index 2 m' + I);
//   inverse  the stages up to gap n/4 are two independent half transforms (again block index (2 + H) h'); the first half's result is parked in the destination
//   row
//            (written and re-read by the same threads), the last stage (one twiddle, W[1]) combines it with the second half straight out of the image.
// Value ranges, prologues and epilogues are those of ntt_rows_body: the same butterflies in the same order, hence the same (canonical) results.
template <bool INV, bool LAZY, int PRO>
__device__ __forceinline__ void ntt_rows_split_body(const NttArgs &a)
{
    extern __shared__ u64 sm[];
    const int n = a.n, n2 = n >> 1, logn2 = a.logn - 1, tid = threadIdx.x, nt = blockDim.x;
    size_t row = blockIdx.x;
    if (INV && PRO == 4) {
        size_t pair; unsigned p;
        if (!xcd_group(blockIdx.x, 3, a.pairs, pair, p)) return;
        row = ((pair / a.mod_count) * 3 + p) * a.mod_count + pair % a.mod_count;
    }
    const int mloc = (int)(row % a.mod_count);
    const int mi = a.mod_base + mloc;
    const ModParams m = a.mods[mi];
    const u64 q = m.q, q2 = m.two_q, q4 = q2 + q2;
    const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
    const ulonglong2 *W = a.w + (size_t)mi * n;
    u64 *dst = a.dst + row * (size_t)n;
    // ---- the value the transform reads at index s (the load prologues of ntt_rows_body)
    const u64 *pa = nullptr, *pb = nullptr, *src = nullptr; int prod = 0, sh = 0;
    if (INV && PRO == 4) {
        const size_t ct = row / (3 * (size_t)a.mod_count); prod = (int)((row / a.mod_count) % 3);
        pa = a.src + ((ct * 2 + (prod == 2 ? 1 : 0)) * a.mod_count + mloc) * (size_t)n;
        pb = a.src + ((ct * 2 + (prod == 0 ? 0 : 1)) * a.mod_count + mloc) * (size_t)n;
    } else if (!INV && PRO == 3) {
        const size_t item = row / a.mod_count, ct = item / a.D; const int g = (int)(item % a.D);
        src = a.src + ((ct * a.src_size + a.src_poly) * a.mod_count + a.dig_i[g]) * (size_t)n; sh = a.dig_shift[g];
    } else {
        const size_t srow = a.dst_ct_rows ? (row / a.dst_ct_rows) * a.src_ct_rows + row % a.dst_ct_rows : (a.src_rows_per_item ? (row / a.mod_count) : row);
        src = a.src + srow * (size_t)n;
    }
    auto pro = [&](u64 v) -> u64 {
        if (a.prologue == 1) v = plain_lift(v, a.pp, mloc, m);
        else if (a.prologue == 2) {
            u64 lo, hi; mul64wide(a.pp.delta[mloc], v, lo, hi);
            if (v >= a.pp.threshold) { u64 l2 = lo + a.pp.uhi[mloc]; hi += (l2 < lo); lo = l2; }
            v = barrett128(lo, hi, m);
        }
        return v;
    };
    auto load2 = [&](int s) -> ulonglong2 {        // points s, s + 1 (s even): 16-byte accesses throughout
        if (INV && PRO == 4) {
            const ulonglong2 av = ld2(pa + s), bv = ld2(pb + s);
            ulonglong2 v{mulmod(av.x, bv.x, m), mulmod(av.y, bv.y, m)};
            if (prod == 1) { v.x = addmod(v.x, v.x, q); v.y = addmod(v.y, v.y, q); }
            return v;
        }
        const ulonglong2 v = ld2(src + s);
        if (!INV && PRO == 3) return ulonglong2{(v.x >> sh) & a.dig_mask, (v.y >> sh) & a.dig_mask};
        return ulonglong2{pro(v.x), pro(v.y)};
    };
    // (the gap-1 stage of a half transform -- its last stage forward, its first inverse -- is applied by the loops that drain / fill the image: ntt_device.h)
    const bool fuse1 = ntt_fused_stage(logn2);
    auto passes = [&](int h) {          // the half transform: n/2 points, twiddle block index (2 + h) m
        const int full = logn2 / 3, rem = fuse1 ? 0 : logn2 - 3 * full, tm = 2 + h;
        if (!INV) {
            int t = n2 >> 1;
            for (int p = 0; p < full; p++, t >>= 3) ntt_pass<false, 3, LAZY>(sm, W, n2, t >> 2, tm * (n2 / (2 * t)), q, q2);
            if (rem == 2) ntt_pass<false, 2, LAZY>(sm, W, n2, t >> 1, tm * (n2 / (2 * t)), q, q2);
            else if (rem == 1) ntt_pass<false, 1, LAZY>(sm, W, n2, t, tm * (n2 / (2 * t)), q, q2);
        } else {
            int t = fuse1 ? 2 : 1;
            for (int p = 0; p < full; p++, t <<= 3) ntt_pass<true, 3, LAZY>(sm, W, n2, t, tm * (n2 / (2 * t)), q, q2);
            if (rem == 2) ntt_pass<true, 2, LAZY>(sm, W, n2, t, tm * (n2 / (2 * t)), q, q2);
            else if (rem == 1) ntt_pass<true, 1, LAZY>(sm, W, n2, t, tm * (n2 / (2 * t)), q, q2);
        }
    };
    auto pair_tw = [&](int h, int s) { return W[(2 + h) * (n2 >> 1) + (s >> 1)]; };       // twiddle of the pair (s, s + 1) of half h in its gap-1 stage
    auto canon = [&](u64 v) -> u64 {
        if (LAZY) return reduce_small(v, q, q2, rq);
        v = v >= q2 ? v - q2 : v; return v >= q ? v - q : v;
    };
    const ulonglong2 tw = W[1];
    if (!INV) {
        const u64 *add = a.addend ? a.addend + row * (size_t)n : nullptr;
        auto bf = [&](u64 X, u64 Y, int h) -> u64 {
            if (LAZY) { const u64 Q = shoup_lazy4(Y, tw.x, tw.y, q); return h ? X + (q4 - Q) : X + Q; }
            X = X >= q2 ? X - q2 : X; const u64 Q = mulmod_shoup_lazy(Y, tw.x, tw.y, q); return h ? X + (q2 - Q) : X + Q;
        };
        auto stage0 = [&](int h) {
            for (int s = 2 * tid; s < n2; s += 2 * nt) {
                const ulonglong2 X = load2(s), Y = load2(s + n2);
                sm_store_pair64(sm, s, bf(X.x, Y.x, h), bf(X.y, Y.y, h));
            }
            __syncthreads();
        };
        auto put = [&](int s, u64 v0, u64 v1) {        // s: index inside the whole row
            if (add) { const ulonglong2 ad = ld2(add + s); v0 = addmod(v0, ad.x, q); v1 = addmod(v1, ad.y, q); }
            if (a.pack_out) st2(dst + s, split28v(v0), split28v(v1)); else st2(dst + s, v0, v1);
        };
        // the transform may run in place (src == dst): nothing is stored before the second half has read its inputs -- the first half's result waits in
        // registers
        constexpr int NPT = 8;                          // n/2 = 8192 values on 1024 threads (the split form serves n = 16384 only)
        u64 r0[NPT];
        stage0(0);
        passes(0);
#pragma unroll
        for (int u = 0; u < NPT / 2; u++) {
            const int s = 2 * (tid + u * nt);
            if (s < n2) { ulonglong2 v = sm_load_pair64(sm, s); if (fuse1) fwd_pair_stage<LAZY>(v, pair_tw(0, s), q, q2); r0[2 * u] = canon(v.x);
                r0[2 * u + 1] = canon(v.y); }
        }
        __syncthreads();
        stage0(1);
#pragma unroll
        for (int u = 0; u < NPT / 2; u++) { const int s = 2 * (tid + u * nt); if (s < n2) put(s, r0[2 * u], r0[2 * u + 1]); }
        passes(1);
        for (int s = 2 * tid; s < n2; s += 2 * nt) { ulonglong2 v = sm_load_pair64(sm, s); if (fuse1) fwd_pair_stage<LAZY>(v, pair_tw(1, s), q, q2);
            put(n2 + s, canon(v.x), canon(v.y)); }
    } else {
        const u64 *add = nullptr;
        if (a.addend) {
            const size_t ct = row / a.rows_per_ct; const int p = (int)((row % a.rows_per_ct) / a.mod_count);
            if (a.add_mode == 2) add = a.addend + ((ct * a.add_size + p) * a.mod_count + mloc) * (size_t)n;
            else if (p == 0) { size_t g = ct / a.add_group; if (a.add_mod) g %= a.add_mod; add = a.addend + (g * a.mod_count + mloc) * (size_t)n; }
        }
        for (int s = 2 * tid; s < n2; s += 2 * nt) { ulonglong2 v = load2(s); if (fuse1) inv_pair_stage<LAZY>(v, pair_tw(0, s), q, q2);
            sm_store_pair64(sm, s, v.x, v.y); }
        __syncthreads();
        passes(0);
        // parked (lazy, below 2^63); read back by this very thread below
        for (int s = 2 * tid; s < n2; s += 2 * nt) { const ulonglong2 v = sm_load_pair64(sm, s); st2(dst + s, v.x, v.y); }
        __syncthreads();
        for (int s = 2 * tid; s < n2; s += 2 * nt) { ulonglong2 v = load2(s + n2); if (fuse1) inv_pair_stage<LAZY>(v, pair_tw(1, s), q, q2);
            sm_store_pair64(sm, s, v.x, v.y); }
        __syncthreads();
        passes(1);
        const u64 q16 = q2 << 3;
        auto last = [&](u64 U, u64 V, u64 &lo, u64 &hi) {
            if (LAZY) {
                const u64 T = q16 - V + U, cu = U + V;
                lo = (cu + ((cu & 1) ? q : 0)) >> 1; hi = shoup_lazy4(T, tw.x, tw.y, q);
            } else {
                const u64 T = q2 - V + U; u64 cu = U + V; cu = cu >= q2 ? cu - q2 : cu;
                lo = (cu + ((cu & 1) ? q : 0)) >> 1; hi = mulmod_shoup_lazy(T, tw.x, tw.y, q);
            }
            if (PRO == 5) { lo = mulmod_shoup(lo, a.post_mul[mloc], a.post_mul_s[mloc], q); hi = mulmod_shoup(hi, a.post_mul[mloc], a.post_mul_s[mloc], q); }
            else { lo = canon(lo); hi = canon(hi); }
        };
        for (int s = 2 * tid; s < n2; s += 2 * nt) {
            const ulonglong2 U = ld2(dst + s), V = sm_load_pair64(sm, s);
            u64 lo0, hi0, lo1, hi1;
            last(U.x, V.x, lo0, hi0); last(U.y, V.y, lo1, hi1);
            if (add) {
                const ulonglong2 al = ld2(add + s), ah = ld2(add + s + n2);
                if (a.add_sign > 0) { lo0 = addmod(lo0, al.x, q); lo1 = addmod(lo1, al.y, q); hi0 = addmod(hi0, ah.x, q); hi1 = addmod(hi1, ah.y, q); }
                else { lo0 = submod(lo0, al.x, q); lo1 = submod(lo1, al.y, q); hi0 = submod(hi0, ah.x, q); hi1 = submod(hi1, ah.y, q); }
            }
            st2(dst + s, lo0, lo1); st2(dst + s + n2, hi0, hi1);
        }
    }
}
template <bool INV, bool LAZY, int PRO>
__global__ void __launch_bounds__(1024, 8) ntt_rows_split_kernel(NttArgs a) { ntt_rows_split_body<INV, LAZY, PRO>(a); }
// the inverse transform over the 61-bit auxiliary base (Harvey form, full 64 x 64 high products) allocates 68 VGPRs on its own: one 1024-thread workgroup
// per CU at n = 8192.  Held to 64 (8 waves per SIMD = two workgroups per CU) it spills five dwords outside the butterfly loops and runs faster.
template <int PRO>
__global__ void __launch_bounds__(1024, 8) ntt_rows_inv61_kernel(NttArgs a) { ntt_rows_body<true, false, PRO>(a); }

// n = 16384: 128 KiB of LDS per row leaves ONE resident workgroup per CU, so the HBM read of a row sits exposed in front of its butterfly
// passes.  This variant keeps the workgroup resident over rows blockIdx, blockIdx + gridDim, ... and fetches the next row into registers
// (16 coefficients per thread) while the passes of the current one run: +5 % at n = 16384.  (At n <= 8192 the extra registers cost a
// resident workgroup and the plain kernel above is faster -- profiles/r01_ntt_experiments.txt.)
template <bool INV, bool LAZY>
__global__ void __launch_bounds__(1024) ntt_rows_prefetch_kernel(NttArgs a)
{
    constexpr int PF = 16;
    extern __shared__ u64 sm[];
    const int n = a.n, logn = a.logn, tid = threadIdx.x, nt = blockDim.x;
    auto src_of = [&](size_t row) {
        const size_t srow = a.dst_ct_rows ? (row / a.dst_ct_rows) * a.src_ct_rows + row % a.dst_ct_rows : (a.src_rows_per_item ? (row / a.mod_count) : row);
        return a.src + srow * (size_t)n;
    };
    u64 pre[PF];
    {
        const u64 *src = src_of(blockIdx.x);
#pragma unroll
        for (int j = 0; j < PF; j++) pre[j] = src[tid + j * nt];
    }
    for (size_t row = blockIdx.x; row < a.rows; row += gridDim.x) {
        const int mloc = (int)(row % a.mod_count);
        const int mi = a.mod_base + mloc;
        const ModParams m = a.mods[mi];
        const u64 q = m.q, q2 = m.two_q;
        const float rq = 1.0f / (float)((u32)(q >> 32) + 1);
        const ulonglong2 *W = a.w + (size_t)mi * n;
        u64 *dst = a.dst + row * (size_t)n;
#pragma unroll
        for (int j = 0; j < PF; j++) {
            u64 v = pre[j];
            if (a.prologue == 1) v = plain_lift(v, a.pp, mloc, m);
            else if (a.prologue == 2) {
                u64 lo, hi; mul64wide(a.pp.delta[mloc], v, lo, hi);
                if (v >= a.pp.threshold) { u64 l2 = lo + a.pp.uhi[mloc]; hi += (l2 < lo); lo = l2; }
                v = barrett128(lo, hi, m);
            }
            sm[lpad(tid + j * nt)] = v;
        }
        if (row + gridDim.x < a.rows) {
            const u64 *nsrc = src_of(row + gridDim.x);
#pragma unroll
            for (int j = 0; j < PF; j++) pre[j] = nsrc[tid + j * nt];
        }
        __syncthreads();
        ntt_row_passes<INV, LAZY>(sm, W, n, logn, q, q2);
        const u64 *add = nullptr;
        if (INV) {
            if (a.addend) {
                const size_t ct = row / a.rows_per_ct; const int p = (int)((row % a.rows_per_ct) / a.mod_count);
                if (a.add_mode == 2) add = a.addend + ((ct * a.add_size + p) * a.mod_count + mloc) * (size_t)n;
                else if (p == 0) { size_t g = ct / a.add_group; if (a.add_mod) g %= a.add_mod; add = a.addend + (g * a.mod_count + mloc) * (size_t)n; }
            }
        } else if (a.addend) add = a.addend + row * (size_t)n;
        for (int s = tid; s < n; s += nt) {
            u64 v = sm[lpad(s)];
            if (LAZY) v = reduce_small(v, q, q2, rq);
            else { v = v >= q2 ? v - q2 : v; v = v >= q ? v - q : v; }
            if (add) v = (INV && a.add_sign < 0) ? submod(v, add[s], q) : addmod(v, add[s], q);
            dst[s] = (!INV && a.pack_out) ? split28v(v) : v;
        }
        __syncthreads();                             // the next row's staging overwrites the LDS image
    }
}

static int ntt_launch(crc_ctx *c, bool inv, NttArgs &a, size_t rows, hipStream_t st)
{
    if (rows == 0) return CRC_OK;
    a.mods = c->d_mods; a.n = c->n; a.logn = c->logn;
    a.w = reinterpret_cast<const ulonglong2 *>(inv ? c->d_irp2 : c->d_rp);
    int nt = c->n / 8; if (nt < 64) nt = 64; if (nt > 1024) nt = 1024;
    size_t lds = (size_t)c->n * 8;
    // lazy butterflies need (1 + 4 log2 n) q < 2^64 (and the float quotient estimate a q of 45+ bits): every coefficient modulus of the
    // reference's parameter sets qualifies (54..55 bits), the 61-bit auxiliary base of Square does not
    bool lazy = true;
    for (int i = a.mod_base; i < a.mod_base + a.mod_count; i++) if (c->tabs[i].m.bits > 57 || c->tabs[i].m.bits < 45) lazy = false;
    const int cus = c->cus;
    if ((a.prologue == 3 && inv) || ((a.prologue == 4 || a.prologue == 5) && !inv)) return CRC_ERR_INVALID_ARGUMENT;
    if (a.prologue == 4) rows = xcd_grid(a.pairs, 3);                  // (the three products of a pair on one XCD: ntt_rows_body)
    // wave-local passes (ntt_rows_wave_kernel): CRC_NTT_WAVE bit 0 n = 8192, bit 1 n = 4096, bit 2 n = 16384 (plain transforms), bit 3 n = 16384 with the
    // Square prologues (slower than the split kernel with the halving butterflies, 11.9 against 11.7 us per squared ciphertext, faster with the ones that do
    // not halve: 11.27 against 11.6 -- profiles/r05_ntt_u64_wave_local_ab.txt, r05_ntt_inverse_unscaled_ab.txt), bit 4: keep the halving butterflies; -1: bits
    // 0 to 3 and 5
    {
        // (bit 5, round 6: n = 2048 -- one cross stage, two waves per workgroup)
        const int sel = c->tune.ntt_wave < 0 ? 47 : c->tune.ntt_wave;
        const int bit = c->n == 8192 ? 0 : c->n == 4096 ? 1 : c->n == 16384 ? (a.prologue ? 3 : 2) : c->n == 2048 ? 5 : -1;
        if (bit >= 0 && ((sel >> bit) & 1) && lazy && !a.addend && (a.prologue == 0 || a.prologue == 4 || a.prologue == 5)) {
            lds = (size_t)c->n * 8;
            // inverse transforms over moduli below 2^55 take the butterflies that do not halve (inv_stages_unscaled; CRC_NTT_WAVE bit 4 switches them off): the
            // table of plain inverse powers, and n^-1 in the constant of the closing multiplication
            bool uns = inv && !(sel & 16);
            for (int i = 0; i < a.mod_count; i++) uns = uns && c->tabs[a.mod_base + i].m.q < (1ull << 55);
            if (uns) {
                a.w = reinterpret_cast<const ulonglong2 *>(c->d_irp);
                for (int i = 0; i < a.mod_count; i++) {
                    const HostNtt &T = c->tabs[a.mod_base + i]; const u64 q = T.m.q;
                    const u64 pm = a.prologue == 5 || a.opt_mul ? (u64)(((unsigned __int128)a.post_mul[i] * T.inv_n) % q) : T.inv_n;
                    a.post_mul[i] = pm; a.post_mul_s[i] = (u64)(((unsigned __int128)pm << 64) / q);
                }
                if (a.opt_mul) a.opt_mul = 2;
            }
            const int cs = c->logn - 10;
#define WAVEK(CSV, U) (a.prologue == 4 ? ntt_rows_wave_kernel<true, 4, CSV, U> : a.prologue == 5 ? ntt_rows_wave_kernel<true, 5, CSV, U> \
                       : inv ? ntt_rows_wave_kernel<true, 0, CSV, U> : ntt_rows_wave_kernel<false, 0, CSV, false>)
#define WAVECS(U) (cs == 1 ? WAVEK(1, U) : cs == 2 ? WAVEK(2, U) : cs == 3 ? WAVEK(3, U) : WAVEK(4, U))
#define WAVEF(FV) (cs == 1 ? ntt_rows_wave_kernel<false, 0, 1, false, FV> : cs == 2 ? ntt_rows_wave_kernel<false, 0, 2, false, FV> \
                   : cs == 3 ? ntt_rows_wave_kernel<false, 0, 3, false, FV> : ntt_rows_wave_kernel<false, 0, 4, false, FV>)
            auto kw = uns ? WAVECS(true) : WAVECS(false);
            if (a.fma_u || a.fma_k) {
                if (inv || a.prologue || a.pack_out) return CRC_ERR_INVALID_ARGUMENT;
                kw = a.fma_u ? WAVEF(1) : WAVEF(2);
            }
#undef WAVEF
#undef WAVECS
#undef WAVEK
            { const int rc = crc_ctx_ensure_lds(c, (const void *)kw, lds); if (rc) return rc; }
            hipLaunchKernelGGL(kw, dim3((unsigned)rows), dim3(c->n / 16), lds, st, a);
            HIPCHK(hipGetLastError());
            return CRC_OK;
        }
    }
    // n = 16384: the row as two halves through a 64-KiB image -- two workgroups per CU (ntt_rows_split_body)
    if (a.fma_u || a.fma_k) return CRC_ERR_UNSUPPORTED;   // (only the wave-local kernel above has those epilogues; the caller runs the product as a pass of its own)
    if (c->n == 16384 && nt == 1024 && c->tune.ntt_split != 0) {
        lds /= 2;
        auto ks = a.prologue == 4 ? (lazy ? ntt_rows_split_kernel<true, true, 4> : ntt_rows_split_kernel<true, false, 4>)
                : a.prologue == 5 ? (lazy ? ntt_rows_split_kernel<true, true, 5> : ntt_rows_split_kernel<true, false, 5>)
                : a.prologue == 3 ? (lazy ? ntt_rows_split_kernel<false, true, 3> : ntt_rows_split_kernel<false, false, 3>)
                : inv ? (lazy ? ntt_rows_split_kernel<true, true, 0> : ntt_rows_split_kernel<true, false, 0>) : (lazy ? ntt_rows_split_kernel<false, true,
                    0> : ntt_rows_split_kernel<false, false, 0>);
        hipLaunchKernelGGL(ks, dim3((unsigned)rows), dim3(nt), lds, st, a);
        HIPCHK(hipGetLastError());
        return CRC_OK;
    }
    if (c->n / nt == 16 && rows > (size_t)cus && a.prologue < 3) {     // n = 16384: one resident workgroup per CU, prefetching the next row
        a.rows = rows;
        auto pk = inv ? (lazy ? ntt_rows_prefetch_kernel<true, true> : ntt_rows_prefetch_kernel<true, false>) : (lazy ? ntt_rows_prefetch_kernel<false,
            true> : ntt_rows_prefetch_kernel<false, false>);
        { const int rc = crc_ctx_ensure_lds(c, (const void *)pk, lds); if (rc) return rc; }
        hipLaunchKernelGGL(pk, dim3((unsigned)cus), dim3(nt), lds, st, a);
        HIPCHK(hipGetLastError());
        return CRC_OK;
    }
    if ((a.prologue == 3 && inv) || (a.prologue == 4 && !inv)) return CRC_ERR_INVALID_ARGUMENT;
    const bool strict61 = !c->tune.ntt_inv61_loose;
    auto kern = a.prologue == 4 ? (lazy ? ntt_rows_kernel<true, true, 4> : strict61 ? ntt_rows_inv61_kernel<4> : ntt_rows_kernel<true, false, 4>)
              : a.prologue == 5 ? (lazy ? ntt_rows_kernel<true, true, 5> : ntt_rows_kernel<true, false, 5>)
              : a.prologue == 3 ? (lazy ? ntt_rows_kernel<false, true, 3> : ntt_rows_kernel<false, false, 3>)
              : inv ? (lazy ? ntt_rows_kernel<true, true, 0> : strict61 ? ntt_rows_inv61_kernel<0> : ntt_rows_kernel<true, false, 0>) : (lazy ?
                  ntt_rows_kernel<false, true, 0> : ntt_rows_kernel<false, false, 0>);
    { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(nt), lds, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

int k_ntt_ct(crc_ctx *c, bool inv, const u64 *src, u64 *dst, size_t count, int size, bool bsk, hipStream_t st,
             const u64 *addend, int add_sign, size_t add_group, int add_mod, int pack_out)
{
    NttArgs a{};
    a.pack_out = inv ? 0 : pack_out;
    a.src = src; a.dst = dst;
    a.mod_base = bsk ? c->k : 0; a.mod_count = bsk ? c->kb : c->k;
    a.addend = addend; a.add_sign = add_sign; a.rows_per_ct = size * a.mod_count; a.add_group = (long long)(add_group ? add_group : 1);
    a.add_mode = 1; a.add_mod = add_mod;
    return ntt_launch(c, inv, a, count * size * a.mod_count, st);
}

// forward NTT in place of size-2 ciphertexts that leave as NTT(row) + u[ct][i] . key[p][i] (u: [count][k][n], key: [2][k][n], both NTT form) -- the device
// encryptor's last step.  CRC_ERR_UNSUPPORTED where the ring has no wave-local kernel: the caller then transforms and multiplies in two passes
int k_ntt_ct_fwd_fma(crc_ctx *c, u64 *ct, size_t count, const u64 *u, const u64 *key, hipStream_t st)
{
    NttArgs a{};
    a.src = ct; a.dst = ct; a.mod_base = 0; a.mod_count = c->k; a.rows_per_ct = 2 * c->k; a.add_group = 1; a.add_mode = 1;
    a.fma_u = u; a.fma_k = key;
    return ntt_launch(c, false, a, count * 2 * c->k, st);
}

// forward NTT in place of size-2 ciphertexts, both polys leaving multiplied by the NTT-form plaintext row w[ct / group] ([.][k][n]) -- the first two of
// multiply_plain's three steps (evaluator.cpp:1193-1323) in one kernel.  CRC_ERR_UNSUPPORTED where the ring has no wave-local kernel
int k_ntt_ct_fwd_mul(crc_ctx *c, u64 *ct, size_t count, const u64 *w, size_t group, hipStream_t st)
{
    NttArgs a{};
    a.src = ct; a.dst = ct; a.mod_base = 0; a.mod_count = c->k; a.rows_per_ct = 2 * c->k; a.add_group = 1; a.add_mode = 1;
    a.fma_k = w; a.fma_group = group ? group : 1;
    return ntt_launch(c, false, a, count * 2 * c->k, st);
}

// inverse NTT of size-2 ciphertexts src -> dst, adding polys 0,1 of a size-`add_size` ciphertext array (relinearize tail)
int k_ntt_ct_addct(crc_ctx *c, const u64 *src, u64 *dst, size_t count, const u64 *addct, int add_size, hipStream_t st)
{
    NttArgs a{};
    a.src = src; a.dst = dst; a.mod_base = 0; a.mod_count = c->k;
    a.addend = addct; a.add_sign = 1; a.rows_per_ct = 2 * c->k; a.add_group = 1; a.add_mode = 2; a.add_size = add_size;
    return ntt_launch(c, true, a, count * 2 * c->k, st);
}

// forward NTT of polys 0,1 of size-`src_size` ciphertexts src -> size-2 dst, adding the NTT-form rows `addrows` ([count][2][k][n])
// (relinearize tail when the result stays in NTT form: NTT(c0, c1) + key-switched c2)
int k_ntt_ct_head_add(crc_ctx *c, const u64 *src, int src_size, u64 *dst, size_t count, const u64 *addrows, hipStream_t st)
{
    NttArgs a{};
    a.src = src; a.dst = dst; a.mod_base = 0; a.mod_count = c->k;
    a.src_ct_rows = src_size * c->k; a.dst_ct_rows = 2 * c->k;
    a.addend = addrows; a.add_sign = 1; a.rows_per_ct = 2 * c->k; a.add_group = 1;
    return ntt_launch(c, false, a, count * 2 * c->k, st);
}

// forward NTT of `items` polynomials under every q_j: src [items][n] -> dst [items][k][n]
int k_spread_ntt(crc_ctx *c, const u64 *src, size_t items, u64 *dst, hipStream_t st)
{
    NttArgs a{};
    a.src = src; a.dst = dst; a.mod_base = 0; a.mod_count = c->k; a.src_rows_per_item = 1; a.prologue = 0;
    a.rows_per_ct = c->k; a.add_group = 1;
    return ntt_launch(c, false, a, items * c->k, st);
}

// relinearisation digits: forward NTT of digit g of c2 (premultiplied by (q/q_i)^-1, poly `src_poly` of the size-`src_size` ciphertexts `src`) under every q_j:
// dst [count][D][k][n]; the digit is cut out of the source word while the row is loaded (NttArgs prologue 3)
int k_digit_ntt(crc_ctx *c, const u64 *src, int src_size, int src_poly, size_t count, int D, const unsigned char *dig_i, const unsigned char *dig_shift,
    int dbc,
                u64 *dst, hipStream_t st, int pack_out)
{
    if (D > 48) return CRC_ERR_UNSUPPORTED;
    NttArgs a{};
    a.src = src; a.dst = dst; a.mod_base = 0; a.mod_count = c->k; a.prologue = 3; a.pack_out = pack_out;
    a.D = D; a.src_size = src_size; a.src_poly = src_poly; a.dig_mask = (1ULL << dbc) - 1;
    for (int g = 0; g < D; g++) { a.dig_i[g] = dig_i[g]; a.dig_shift[g] = dig_shift[g]; }
    a.rows_per_ct = c->k; a.add_group = 1;
    return ntt_launch(c, false, a, count * D * c->k, st);
}

// inverse NTT of the three dyadic products (a^2, 2ab, b^2) of size-2 NTT-form ciphertexts src [count][2][K][n] -> dst [count][3][K][n] (coefficient form),
// K = k moduli of q or kb moduli of Bsk; the products are formed while the row is loaded (NttArgs prologue 4)
// opt_mul (per modulus, may be null): constants the consumer of dst would multiply the rows by first thing; *applied says whether the transform took them into
// its closing multiplication (the kernels that end in one: ntt_rows_wave_kernel's unscaled inverse) or left the rows as they are
int k_square_intt(crc_ctx *c, const u64 *src, u64 *dst, size_t count, bool bsk, hipStream_t st, const u64 *opt_mul, bool *applied)
{
    NttArgs a{};
    a.src = src; a.dst = dst; a.mod_base = bsk ? c->k : 0; a.mod_count = bsk ? c->kb : c->k; a.prologue = 4;
    a.rows_per_ct = 3 * a.mod_count; a.add_group = 1; a.pairs = count * a.mod_count;
    if (opt_mul) { a.opt_mul = 1; for (int i = 0; i < a.mod_count; i++) a.post_mul[i] = opt_mul[i]; }
    const int rc = ntt_launch(c, true, a, count * 3 * a.mod_count, st);
    if (applied) *applied = a.opt_mul == 2;
    return rc;
}

// inverse NTT of size-`size` ciphertexts over q whose result leaves multiplied by mul[i] mod q_i (mul_s: Shoup companions) -- NttArgs prologue 5
int k_ntt_ct_inv_scaled(crc_ctx *c, const u64 *src, u64 *dst, size_t count, int size, const u64 *mul, const u64 *mul_s, hipStream_t st)
{
    NttArgs a{};
    a.src = src; a.dst = dst; a.mod_base = 0; a.mod_count = c->k; a.prologue = 5;
    a.rows_per_ct = size * c->k; a.add_group = 1;
    for (int i = 0; i < c->k; i++) { a.post_mul[i] = mul[i]; a.post_mul_s[i] = mul_s[i]; }
    return ntt_launch(c, true, a, count * size * c->k, st);
}

int k_plain_ntt(crc_ctx *c, const u64 *d_plain, size_t count, int mode, bool do_ntt, u64 *d_out, hipStream_t st);

// plaintext [count][n] -> [count][k][n]: lift (mode 1) or delta-scale (mode 2), then forward NTT (unless !do_ntt)
__global__ void plain_prep_kernel(const u64 *plain, u64 *out, const ModParams *mods, int n, int k, int mode, PlainParams pp)
{
    const size_t row = blockIdx.x; const int i = (int)(row % k);
    const ModParams m = mods[i];
    const u64 *src = plain + (row / k) * (size_t)n; u64 *dst = out + row * (size_t)n;
    for (int s = threadIdx.x; s < n; s += blockDim.x) {
        u64 v = src[s];
        if (mode == 1) v = plain_lift(v, pp, i, m);
        else {
            u64 lo, hi; mul64wide(pp.delta[i], v, lo, hi);
            if (v >= pp.threshold) { u64 l2 = lo + pp.uhi[i]; hi += (l2 < lo); lo = l2; }
            v = barrett128(lo, hi, m);
        }
        dst[s] = v;
    }
}

int k_plain_ntt(crc_ctx *c, const u64 *d_plain, size_t count, int mode, bool do_ntt, u64 *d_out, hipStream_t st)
{
    if (count == 0) return CRC_OK;
    if (do_ntt) {
        NttArgs a{};
        a.src = d_plain; a.dst = d_out; a.mod_base = 0; a.mod_count = c->k; a.src_rows_per_item = 1; a.prologue = mode; a.pp = c->plain;
        a.rows_per_ct = c->k; a.add_group = 1;
        return ntt_launch(c, false, a, count * c->k, st);
    }
    hipLaunchKernelGGL(plain_prep_kernel, dim3((unsigned)(count * c->k)), dim3(256), 0, st, d_plain, d_out, c->d_mods, c->n, c->k, mode, c->plain);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// compact plaintexts [count][CRC_PLAIN_COMPACT_WORDS] -> dense [count][n]: the fractional encoder only ever sets coefficients 0..63 (integer part) and
// n-32..n-1 (fraction), so the host ships 96 words per weight instead of n (PlainModelWoPad's fc3 at n = 16384: 0.3 GB over PCIe instead of 52 GB)
__global__ void __launch_bounds__(256) plain_expand_kernel(const u64 *compact, u64 *out, int n)
{
    const size_t row = blockIdx.x;
    const u64 *src = compact + row * (size_t)CRC_PLAIN_COMPACT_WORDS; u64 *dst = out + row * (size_t)n;
    for (int s = threadIdx.x * 2; s < n; s += blockDim.x * 2) {
        ulonglong2 v = make_ulonglong2(0, 0);
        if (s < CRC_PLAIN_COMPACT_LOW) { v.x = src[s]; v.y = src[s + 1]; }
        else if (s >= n - CRC_PLAIN_COMPACT_HIGH) { v.x = src[CRC_PLAIN_COMPACT_LOW + s - (n - CRC_PLAIN_COMPACT_HIGH)];
            v.y = src[CRC_PLAIN_COMPACT_LOW + s + 1 - (n - CRC_PLAIN_COMPACT_HIGH)]; }
        *reinterpret_cast<ulonglong2 *>(dst + s) = v;
    }
}
int k_plain_expand(crc_ctx *c, const u64 *d_compact, size_t count, u64 *d_plain, hipStream_t st)
{
    if (count == 0) return CRC_OK;
    if (count > 0x7fffffffu) return CRC_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(plain_expand_kernel, dim3((unsigned)count), dim3(256), 0, st, d_compact, d_plain, c->n);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// element-wise row kernels (HBM-bound): 16 B per lane
// ---------------------------------------------------------------------------------------------------------------
// op 0: acc = acc + b            (Evaluator::add, evaluator.cpp:254-294)
// op 1: acc = acc +/- plain      (add_plain/sub_plain on poly 0 only; b = delta-form plaintext shared by `group` cts)
// op 2: acc = acc * w            (multiply_plain_ntt, evaluator.cpp:1541-1585; w shared by `group` cts, both polys)
__global__ void __launch_bounds__(256) rowwise_kernel(u64 *acc, const u64 *b, const ModParams *mods, int n, int k, int size,
                                                      int op, int sign, unsigned long long group, unsigned long long gmod)
{
    const size_t row = blockIdx.x;
    const int i = (int)(row % k);
    const ModParams m = mods[i];
    const size_t ct = row / ((size_t)size * k);
    const int p = (int)((row / k) % size);
    u64 *x = acc + row * (size_t)n;
    const u64 *y;
    if (op == 0) y = b + row * (size_t)n;
    else { if (op == 1 && p != 0) return; size_t g = ct / group; if (gmod) g %= gmod; y = b + (g * k + i) * (size_t)n; }
    for (int s = threadIdx.x * 2; s < n; s += blockDim.x * 2) {
        ulonglong2 xv = *reinterpret_cast<const ulonglong2 *>(x + s);
        const ulonglong2 yv = *reinterpret_cast<const ulonglong2 *>(y + s);
        if (op == 0 || (op == 1 && sign > 0)) { xv.x = addmod(xv.x, yv.x, m.q); xv.y = addmod(xv.y, yv.y, m.q); }
        else if (op == 1) { xv.x = submod(xv.x, yv.x, m.q); xv.y = submod(xv.y, yv.y, m.q); }
        else { xv.x = mulmod(xv.x, yv.x, m); xv.y = mulmod(xv.y, yv.y, m); }
        *reinterpret_cast<ulonglong2 *>(x + s) = xv;
    }
}

// in-place conversion of NTT-form rows between canonical residues and 28-bit limb pairs
__global__ void __launch_bounds__(256) pack28_kernel(u64 *rows, size_t words, int unpack)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * 2;             // grid-stride: weight tensors exceed 2^32 threads' worth of words
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < words; i += stride) {
        ulonglong2 v = *reinterpret_cast<ulonglong2 *>(rows + i);
        if (unpack) { v.x = (v.x & 0xffffffffULL) | ((v.x >> 32) << 28); v.y = (v.y & 0xffffffffULL) | ((v.y >> 32) << 28); }
        else { v.x = (v.x & 0x0fffffffULL) | ((v.x >> 28) << 32); v.y = (v.y & 0x0fffffffULL) | ((v.y >> 28) << 32); }
        *reinterpret_cast<ulonglong2 *>(rows + i) = v;
    }
}
int k_pack28(crc_ctx *c, u64 *rows, size_t nrows, bool unpack, hipStream_t st)
{
    if (nrows == 0) return CRC_OK;
    for (int i = 0; i < c->k; i++) if (c->tabs[i].m.bits > 56) return CRC_ERR_UNSUPPORTED;
    const size_t words = nrows * (size_t)c->n;
    size_t blocks = (words / 2 + 255) / 256; if (blocks > (1u << 20)) blocks = 1u << 20;
    hipLaunchKernelGGL(pack28_kernel, dim3((unsigned)blocks), dim3(256), 0, st, rows, words, unpack ? 1 : 0);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

int k_rowwise(crc_ctx *c, u64 *acc, const u64 *b, size_t count, int size, int op, int sign, size_t group, size_t gmod, hipStream_t st)
{
    if (count == 0) return CRC_OK;
    size_t rows = count * size * c->k;
    hipLaunchKernelGGL(rowwise_kernel, dim3((unsigned)rows), dim3(256), 0, st, acc, b, c->d_mods, c->n, c->k, size, op, sign,
                       (unsigned long long)(group ? group : 1), (unsigned long long)gmod);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// window sum (PoolingLayer::forward, poolingLayer.cpp:22-44) with optional dyadic multiply by an NTT-form plaintext
// (only meaningful when the tensor is NTT-resident) and optional per-channel affine (batch-norm in NTT form).
__global__ void __launch_bounds__(256) pool_kernel(const u64 *x, u64 *y, const ModParams *mods, int n, int k,
                                                   int zd, int xd, int yd, int xs, int ys, int xf, int yf, int xo, int yo, const u64 *mul, int pack_out)
{
    // one block per output row: row = (((b*zd + z)*xo + ox)*yo + oy)*2k + p*k + i
    const size_t row = blockIdx.x;
    const int i = (int)(row % k); const int p = (int)((row / k) % 2);
    size_t ct = row / (2 * (size_t)k);
    const int oy = (int)(ct % yo); ct /= yo; const int ox = (int)(ct % xo); ct /= xo;       // ct = b*zd + z
    const ModParams m = mods[i];
    const u64 *base = x + ((ct * xd + (size_t)ox * xs) * yd + (size_t)oy * ys) * (2 * (size_t)k * n) + ((size_t)p * k + i) * n;
    u64 *dst = y + row * (size_t)n;
    const u64 *w = mul ? mul + (size_t)i * n : nullptr;
    for (int s = threadIdx.x * 2; s < n; s += blockDim.x * 2) {
        ulonglong2 acc = make_ulonglong2(0, 0);
        for (int kx = 0; kx < xf; kx++) for (int ky = 0; ky < yf; ky++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(base + ((size_t)kx * yd + ky) * (2 * (size_t)k * n) + s);
            acc.x = addmod(acc.x, v.x, m.q); acc.y = addmod(acc.y, v.y, m.q);
        }
        if (w) { const ulonglong2 wv = *reinterpret_cast<const ulonglong2 *>(w + s); acc.x = mulmod(acc.x, wv.x, m); acc.y = mulmod(acc.y, wv.y, m); }
        if (pack_out) { acc.x = split28v(acc.x); acc.y = split28v(acc.y); }          // hand-over to a conv / dense layer in its operand form
        *reinterpret_cast<ulonglong2 *>(dst + s) = acc;
    }
}

int k_pool(crc_ctx *c, const u64 *x, u64 *y, int B, int zd, int xd, int yd, int xs, int ys, int xf, int yf, const u64 *mul, hipStream_t st, int pack_out)
{
    int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys + 1;
    size_t rows = (size_t)B * zd * xo * yo * 2 * c->k;
    if (rows == 0) return CRC_OK;
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)rows), dim3(256), 0, st, x, y, c->d_mods, c->n, c->k, zd, xd, yd, xs, ys, xf, yf, xo, yo, mul, pack_out);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// batch-norm on an NTT-resident tensor: x = (x - mean_delta_ntt[z]) * invstd_ntt[z]   (poly 0 gets the subtraction)
__global__ void __launch_bounds__(256) bn_ntt_kernel(u64 *x, const u64 *mean, const u64 *invstd, const ModParams *mods, int n, int k,
                                                     int zd, int hw)
{
    const size_t row = blockIdx.x;
    const int i = (int)(row % k); const int p = (int)((row / k) % 2);
    const size_t ct = row / (2 * (size_t)k);
    const int z = (int)((ct / hw) % zd);
    const ModParams m = mods[i];
    u64 *d = x + row * (size_t)n;
    const u64 *mu = mean + ((size_t)z * k + i) * n, *w = invstd + ((size_t)z * k + i) * n;
    for (int s = threadIdx.x * 2; s < n; s += blockDim.x * 2) {
        ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(d + s);
        const ulonglong2 wv = *reinterpret_cast<const ulonglong2 *>(w + s);
        if (p == 0) { const ulonglong2 mv = *reinterpret_cast<const ulonglong2 *>(mu + s); v.x = submod(v.x, mv.x, m.q); v.y = submod(v.y, mv.y, m.q); }
        v.x = mulmod(v.x, wv.x, m); v.y = mulmod(v.y, wv.y, m);
        *reinterpret_cast<ulonglong2 *>(d + s) = v;
    }
}

int k_bn_ntt(crc_ctx *c, u64 *x, int B, int zd, int hw, const u64 *mean, const u64 *invstd, hipStream_t st)
{
    size_t rows = (size_t)B * zd * hw * 2 * c->k;
    if (rows == 0) return CRC_OK;
    hipLaunchKernelGGL(bn_ntt_kernel, dim3((unsigned)rows), dim3(256), 0, st, x, mean, invstd, c->d_mods, c->n, c->k, zd, hw);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ct x pt multiply-accumulate in the NTT domain (the hot loop of convolution3d / FullyConnectedLayer::forward):
//   y[b][f][p][poly][i][s] = sum_T  x[b][xoff[p] + toff[T]][poly][i][s] * w[f][T][i][s]   (mod q_i)
// The reference computes INTT(x*w) per product and adds in coefficient form (convolutionalLayer.cpp:73-88); summing
// in the NTT domain and transforming once is the same element of Z_q[x]/(x^n+1), hence the same bits.
// One lane = one slot s; a thread keeps a PT x 2(polys) x FT register tile of 128-bit lazy accumulators and does a
// single Barrett reduction per output (products < 2^120 for T < 2^10..2^18 terms).
// ---------------------------------------------------------------------------------------------------------------
// gather tables of a valid-padding strided convolution (device-built: no host data on the launch path)
__global__ void conv_offsets_kernel(int *xoff, int *toff, unsigned *toffw, unsigned ctw, int P, int T, int xd, int yd, int xs, int ys, int xf, int yf, int yo)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < P) { const int ox = idx / yo, oy = idx % yo; xoff[idx] = (ox * xs) * yd + oy * ys; }
    if (idx < T + 8) {      // 8 padding entries repeat the last term (the staged weights are zeroed there)
        const int t = min(idx, T - 1);
        const int z = t / (xf * yf), kx = (t / yf) % xf, ky = t % yf; const int o = (z * xd + kx) * yd + ky;
        toff[idx] = o; toffw[idx] = (unsigned)o * ctw;
    }
}
int k_conv_offsets(crc_ctx *c, int *xoff, int *toff, unsigned *toffw, int P, int T, int in_cts, int xd, int yd, int xs, int ys, int xf, int yf, int yo,
    hipStream_t st)
{
    const int m = P > T + 8 ? P : T + 8;
    const size_t ctw = 2 * (size_t)c->k * c->n;
    if ((size_t)in_cts * ctw > 0xffffffffULL) return CRC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv_offsets_kernel, dim3((m + 255) / 256), dim3(256), 0, st, xoff, toff, toffw, (unsigned)ctw, P, T, xd, yd, xs, ys, xf, yf, yo);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

struct MacArgs {
    const u64 *x; const u64 *w; u64 *y; const ModParams *mods;
    const int *xoff; const int *toff;        // device tables: output pixel -> base ct index, term -> ct offset
    const unsigned *toffw;                   // term -> element offset toff*2kn, padded by 8 entries (mac2)
    int n, k, B, P, F, T, in_cts;            // P output pixels per image, in_cts input cts per image
    const u64 *bias; int bias_sign;          // optional NTT-form delta bias [F][k][n] added to poly 0
    int mt_fastest;                          // tile walk order (see the kernels)
    int xp, wp, yp;                          // x / w arrive packed (28-bit limb pairs), y leaves packed
#ifdef CRC_TUNING
    int dbg;                                 // -DCRC_TUNING builds only (make tuning; tools/bench_mac.py): ablation switches of mac2_kernel that give
                                             // WRONG results (1 = skip operand staging, 2 = skip barriers too, ...).  The shipped library has none.
#endif
    const u64 *zero;                         // >= 1 KiB of zeros (mac3: source of the terms past T)
    int gxd, gyd, gxf, gyf;                  // window geometry: toff[t] = (z*gxd + kx)*gyd + ky for t = (z*gxf + kx)*gyf + ky
};

// x (< 2^56) -> {low dword = x & (2^28-1), high dword = x >> 28}: two 32-bit VALU ops (64-bit shifts are slow on gfx950)
__device__ __forceinline__ u64 split28(u64 r)
{
    const u32 lo = (u32)r, hi = (u32)(r >> 32);
    const u32 x0 = lo & 0x0fffffffu, x1 = __builtin_amdgcn_alignbit(hi, lo, 28);
    return (u64)x0 | ((u64)x1 << 32);
}
// inverse: the canonical residue of a packed value.  "Packed" (CRC_NTTP) is how the MAC kernels want their operands: weights and the
// NTT-resident tensors that travel from one conv / dense layer to the next are kept in this form so that nobody has to split them again
__device__ __forceinline__ u64 unsplit28(u64 p) { return (p & 0xffffffffULL) | ((p >> 32) << 28); }

#ifdef CRC_TUNING
#define MAC_DBG (a.dbg)
#else
#define MAC_DBG 0
#endif

template <int PT, int FT>
__global__ void __launch_bounds__(256) mac_kernel(MacArgs a)
{
    const int n = a.n, k = a.k;
    const int sblocks = n / blockDim.x;
    const int rs = blockIdx.x % (sblocks * k);
    const int i = rs / sblocks;                                           // residue
    const int s = (rs % sblocks) * blockDim.x + threadIdx.x;              // slot
    const int ptiles = (a.P + PT - 1) / PT;
    const int bp = blockIdx.x / (sblocks * k);
    const int b = bp / ptiles, p0 = (bp % ptiles) * PT;
    const int f0 = blockIdx.y * FT;
    const ModParams m = a.mods[i];
    const size_t ctw = 2 * (size_t)k * n, rown = (size_t)i * n + s;

    u64 lo[PT][2][FT], hi[PT][2][FT];
#pragma unroll
    for (int pp = 0; pp < PT; pp++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int ff = 0; ff < FT; ff++) { lo[pp][c][ff] = 0; hi[pp][c][ff] = 0; }

    const u64 *xb[PT];
#pragma unroll
    for (int pp = 0; pp < PT; pp++) {
        const int p = min(p0 + pp, a.P - 1);
        xb[pp] = a.x + ((size_t)b * a.in_cts + a.xoff[p]) * ctw + rown;
    }
    const u64 *wb = a.w + (size_t)f0 * a.T * k * n + rown;
    const size_t wstride_f = (size_t)a.T * k * n, wstride_t = (size_t)k * n;

    for (int t = 0; t < a.T; t++) {
        const size_t to = (size_t)a.toff[t] * ctw;
        u64 xv[PT][2], wv[FT];
#pragma unroll
        for (int pp = 0; pp < PT; pp++) { xv[pp][0] = xb[pp][to]; xv[pp][1] = xb[pp][to + (size_t)k * n]; if (a.xp) { xv[pp][0] = unsplit28(xv[pp][0]);
            xv[pp][1] = unsplit28(xv[pp][1]); } }
#pragma unroll
        for (int ff = 0; ff < FT; ff++) { wv[ff] = (f0 + ff < a.F) ? wb[ff * wstride_f + t * wstride_t] : 0; if (a.wp) wv[ff] = unsplit28(wv[ff]); }
#pragma unroll
        for (int pp = 0; pp < PT; pp++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int ff = 0; ff < FT; ff++) {
                    u64 pl, ph; mul64wide(xv[pp][c], wv[ff], pl, ph);
                    const u64 nl = lo[pp][c][ff] + pl;
                    hi[pp][c][ff] += ph + (nl < pl);
                    lo[pp][c][ff] = nl;
                }
    }
#pragma unroll
    for (int pp = 0; pp < PT; pp++) {
        const int p = p0 + pp;
        if (p >= a.P) continue;
#pragma unroll
        for (int ff = 0; ff < FT; ff++) {
            const int f = f0 + ff;
            if (f >= a.F) continue;
            u64 *dst = a.y + (((size_t)b * a.F + f) * a.P + p) * ctw + rown;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                u64 v = barrett128(lo[pp][c][ff], hi[pp][c][ff], m);
                if (c == 0 && a.bias) { const u64 bv = a.bias[(size_t)f * k * n + rown]; v = a.bias_sign > 0 ? addmod(v, bv, m.q) : submod(v, bv, m.q); }
                dst[(size_t)c * k * n] = a.yp ? split28(v) : v;
            }
        }
    }
}

int k_mac(crc_ctx *c, const u64 *x, const u64 *w, u64 *y, const int *d_xoff, const int *d_toff, int B, int P, int F, int T, int in_cts,
          const u64 *bias_ntt, hipStream_t st, int xp, int wp, int yp)
{
    if (B == 0 || P == 0 || F == 0) return CRC_OK;
    // 128-bit lazy accumulation bound: T * q^2 < 2^128
    int maxbits = 0; for (int i = 0; i < c->k; i++) if ((int)c->tabs[i].m.bits > maxbits) maxbits = c->tabs[i].m.bits;
    int tb = 0; while ((1LL << tb) < T) tb++;
    if (2 * maxbits + tb > 127) return CRC_ERR_UNSUPPORTED;
    MacArgs a{};
    a.x = x; a.w = w; a.y = y; a.mods = c->d_mods; a.xoff = d_xoff; a.toff = d_toff;
    a.n = c->n; a.k = c->k; a.B = B; a.P = P; a.F = F; a.T = T; a.in_cts = in_cts; a.bias = bias_ntt; a.bias_sign = 1; a.xp = xp; a.wp = wp; a.yp = yp;
    const int threads = c->n < 256 ? c->n : 256;
    constexpr int PT = 2, FT = 4;
    const size_t gx = (size_t)(c->n / threads * c->k) * B * ((P + PT - 1) / PT);
    if (gx > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    dim3 grid((unsigned)gx, (unsigned)((F + FT - 1) / FT));
    hipLaunchKernelGGL((mac_kernel<PT, FT>), grid, dim3(threads), 0, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// mac2: the production ct x pt multiply-accumulate kernel (moduli up to 55 bits; wider ones take mac_kernel above).
//
// Arithmetic.  gfx950 has no 64x64 multiplier; v_mad_u64_u32 (32x32+64) issues at quarter rate (measured 17.3 T/s,
// profiles/r01_microbench_valu.txt) and is the roofline of this kernel.  Operands are split into 28-bit limbs
// x = x1*2^28 + x0 (x1 < 2^27 for q < 2^55) and products are formed Karatsuba-style with THREE mads per MAC:
//     A0 += x0*w0     A2 += x1*w1     A1 += (x0+x1)*(w0+w1)          (each term < 2^57.2, carry-free for 32 terms)
// Every 32 terms the top bit of each 64-bit accumulator is moved into a packed overflow counter, so the sum over up to
// 2^15 terms is exact; one 128-bit recombination  V = A0 + (A1-A0-A2)*2^28 + A2*2^56  and ONE Barrett reduction per
// output give the canonical residue -- the same element of Z_q as the reference's per-product reductions.
//
// Data movement.  A 512-thread workgroup owns 64 consecutive slots of one residue and a (2*PX*WM rows) x (FT*WN filters)
// output tile; lane = slot, the 8 waves form a WM x WN grid of PX*2 x FT register tiles.  Per reduction step the
// workgroup needs only 2*PX*WM + FT*WN operand vectors (512 B each); they are fetched once (coalesced 8 B/lane),
// double-buffered through LDS (S steps per stage, one barrier per stage) and re-read by the waves that share them.
// Workgroups are numbered so that the 32 CUs of an XCD work on the same slot block and neighbouring tiles at the same
// time: their shared operands are served by that XCD's L2 instead of HBM.
// ---------------------------------------------------------------------------------------------------------------

template <int PX, int FT, int WM, int WN, int S, int DEPTH = 2>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) mac2_kernel(MacArgs a)
{
    constexpr int NW = WM * WN, MT = PX * WM, ROWS = 2 * MT, FW = FT * WN;
    constexpr int VEC = S * (ROWS + FW), NPAIR = VEC / 2, RLOAD = 2 * ((NPAIR + NW - 1) / NW);      // operand vectors are staged in (even, odd) pairs
    static_assert(FT % 2 == 0 && (ROWS + FW) % 2 == 0, "pairs");
    extern __shared__ __attribute__((aligned(16))) u64 smem[];      // 2 x [S][(ROWS + FW)/2 pairs][64 lanes][2]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave / WN, wn = wave % WN;
    const int n = a.n, k = a.k;
    const int M = a.B * a.P;
    const int mtiles = (M + MT - 1) / MT, ftiles = (a.F + FW - 1) / FW, sbt = (n >> 6) * k;
    // XCD-aware decode: consecutive workgroup ids go to different XCDs (round-robin dispatch), so give every XCD its own
    // slot blocks and walk (f tile fastest, then m tile) inside one slot block
    int g = blockIdx.x, sb, tile;
    if ((sbt & 7) == 0) { const int xcd = g & 7, r = g >> 3, per = sbt >> 3; sb = xcd * per + r / (mtiles * ftiles); tile = r % (mtiles * ftiles); }
    else { sb = g / (mtiles * ftiles); tile = g % (mtiles * ftiles); }
    // walk order inside a slot block: filter tiles fastest (neighbouring workgroups share the pixel tile's x rows) or, for layers with
    // few pixel tiles (dense), pixel tiles fastest (neighbours share the filter tile's weight rows, which are then read from HBM once)
    const int ft = a.mt_fastest ? tile / mtiles : tile % ftiles, mt = a.mt_fastest ? tile % mtiles : tile / ftiles;
    const int i = sb / (n >> 6), s = ((sb % (n >> 6)) << 6) + lane;
    const int m0 = mt * MT, f0 = ft * FW;
    const size_t rown = (size_t)i * n + s, kn = (size_t)k * n, ctw = 2 * kn;
    const ModParams m = a.mods[i];

    // wave-uniform base pointer of every operand vector this wave stages (kept in SGPRs; lanes add (i*n+s)*8)
    const u64 *vbase[RLOAD]; int vstep[RLOAD]; bool visx[RLOAD];
#pragma unroll
    for (int j = 0; j < RLOAD; j++) {
        const int v = 2 * (wave + (j >> 1) * NW) + (j & 1);
        const int vv = min(v, VEC - 1);                  // surplus slots (NPAIR not a multiple of NW) re-load the last vector, never stored
        const int e = vv % (ROWS + FW);
        vstep[j] = vv / (ROWS + FW);
        if (e < ROWS) {
            const int mm = min(m0 + (e >> 1), M - 1), b = mm / a.P, p = mm % a.P;
            vbase[j] = a.x + ((size_t)b * a.in_cts + a.xoff[p]) * ctw + (size_t)(e & 1) * kn; visx[j] = true;
        } else {
            const int f = min(f0 + (e - ROWS), a.F - 1);     // filters past F are computed on a clamped copy and never stored
            vbase[j] = a.w + (size_t)f * a.T * kn; visx[j] = false;
        }
    }

    // wide staging loads: ONE global_load_dwordx4 fetches two vectors of a pair -- lanes 0..31 take slots (2l, 2l+1) of the even
    // vector, lanes 32..63 the same slots of the odd one (the cost of staging is per memory INSTRUCTION, see DESIGN.md ablation)
    const int half = lane >> 5, l2 = (lane & 31) * 2;
    const u32 rown2 = (u32)((size_t)i * n + (s - lane) + l2);
    const u32 hmask = half ? 0xffffffffu : 0u;
    u32 vdelta[RLOAD / 2];                                   // element distance even -> odd vector of a pair (wave-uniform)
#pragma unroll
    for (int jj = 0; jj < RLOAD / 2; jj++) vdelta[jj] = (u32)(vbase[2 * jj + 1] - vbase[2 * jj]);

    u64 A0[PX * 2][FT], A1[PX * 2][FT], A2[PX * 2][FT]; u32 OV[PX * 2][FT];
#pragma unroll
    for (int r = 0; r < PX * 2; r++)
#pragma unroll
        for (int f = 0; f < FT; f++) { A0[r][f] = 0; A1[r][f] = 0; A2[r][f] = 0; OV[r][f] = 0; }

    ulonglong2 regA[RLOAD / 2], regB[RLOAD / 2];    // two register stages: operand loads run TWO pipeline stages ahead of their use
    // per-term x offsets: the padded toffw table is copied into LDS once (scalar loads inside the loop would share lgkmcnt
    // with the LDS operand reads and stall them); lanes read the same word (broadcast)
    u32 *tw = reinterpret_cast<u32 *>(smem + (size_t)2 * VEC * 64);
    for (int t = threadIdx.x; t < a.T + 8; t += blockDim.x) tw[t] = a.toffw[t];
    __syncthreads();                                 // the table is read by every wave from the first load_stage on
    const u32 kn32 = (u32)kn;
    auto load_one = [&](int st, int jj, ulonglong2 (&reg)[RLOAD / 2]) {
        const int t = min(st * S + vstep[2 * jj], a.T - 1);
        const u32 off = visx[2 * jj] ? tw[t] : (u32)t * kn32;
        // consumed only by store_pair
        reg[jj] = *reinterpret_cast<const ulonglong2 *>(reinterpret_cast<const char *>(vbase[2 * jj]) + ((size_t)(off + rown2 + (hmask & vdelta[jj])) << 3));
    };
    auto store_pair = [&](int st, int jj, ulonglong2 (&reg)[RLOAD / 2]) {
        u64 *dst = smem + (size_t)(st & 1) * VEC * 64;
        const int pr = wave + jj * NW;
        const bool dead = !visx[2 * jj] && st * S + vstep[2 * jj] >= a.T;                 // weights past the last term are zero (x may be anything valid)
        ulonglong2 v;
        const bool packed = visx[2 * jj] ? a.xp : a.wp;                                  // operands that arrive packed are already split
        v.x = dead ? 0 : reg[jj].x; v.y = dead ? 0 : reg[jj].y;
        if (!packed) { v.x = split28(v.x); v.y = split28(v.y); }                          // pre-split once: low dword = x0 (28 bit), high dword = x1
        if (pr < NPAIR) *reinterpret_cast<ulonglong2 *>(dst + (2 * pr + half) * 64 + l2) = v;     // one ds_write_b128: two adjacent slots of one vector
    };
    auto load_stage = [&](int st, ulonglong2 (&reg)[RLOAD / 2]) {
        if (st > 1 && (MAC_DBG == 1 || MAC_DBG == 2 || MAC_DBG == 4)) return;
#pragma unroll
        for (int jj = 0; jj < RLOAD / 2; jj++) { if (MAC_DBG == 7 && st > 1 && (jj & 1)) continue; load_one(st, jj, reg); }
    };
    auto store_stage = [&](int st, ulonglong2 (&reg)[RLOAD / 2]) {
        if (st > 1 && (MAC_DBG == 1 || MAC_DBG == 2 || MAC_DBG == 3)) return;
#pragma unroll
        for (int jj = 0; jj < RLOAD / 2; jj++) store_pair(st, jj, reg);
    };
    // the staging traffic of a stage (LDS writes of stage st+1, global loads of stage st+2) is issued INSIDE the compute of stage
    // st, at a different reduction step in the two waves that share a SIMD (waves w and w+4): while one wave issues its memory
    // instructions the other one has multiply-adds to issue, instead of all eight waves staging at the same moment
    const int io_sel = MAC_DBG == 8 ? (wave & 1) : MAC_DBG == 9 ? 0 : ((wave >> 2) & 1);     // dbg 8/9: tuning controls (tools/bench_mac.py)
    const int io_step = io_sel ? S / 2 : 0;
    auto compute_stage = [&](int st, auto &&io) {
        const u64 *buf = smem + (size_t)(st & 1) * VEC * 64;
#pragma unroll 1
        for (int step = 0; step < S; step++) {
            if (step == io_step) io();
            const u64 *sv = buf + step * (ROWS + FW) * 64 + lane;
            u32 w0[FT], w1[FT], ws[FT];
#pragma unroll
            for (int f = 0; f < FT; f++) { const u64 wv = sv[(ROWS + wn * FT + f) * 64]; w0[f] = (u32)wv; w1[f] = (u32)(wv >> 32); ws[f] = w0[f] + w1[f]; }
#pragma unroll
            for (int r = 0; r < PX * 2; r++) {
                const u64 xv = sv[(wm * PX * 2 + r) * 64];
                const u32 x0 = (u32)xv, x1 = (u32)(xv >> 32), xs = x0 + x1;
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    A0[r][f] += (u64)x0 * w0[f];
                    A2[r][f] += (u64)x1 * w1[f];
                    A1[r][f] += (u64)xs * ws[f];
                }
            }
        }
        if ((((st + 1) * S) & 31) < S) {              // every 32 reduction steps: park bit 63 of each accumulator in the overflow word
#pragma unroll
            for (int r = 0; r < PX * 2; r++)
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    OV[r][f] += (u32)(A0[r][f] >> 63) + ((u32)(A1[r][f] >> 63) << 10) + ((u32)(A2[r][f] >> 63) << 20);
                    A0[r][f] &= ~(1ULL << 63); A1[r][f] &= ~(1ULL << 63); A2[r][f] &= ~(1ULL << 63);
                }
        }
    };

    const int nstages = (a.T + S - 1) / S;
    load_stage(0, regA); store_stage(0, regA);
    if (DEPTH == 2) {
        if (nstages > 1) load_stage(1, regA);
        __syncthreads();
        for (int st = 0; st < nstages; st += 2) {
            compute_stage(st, [&] { if (st + 1 < nstages) store_stage(st + 1, regA); if (st + 2 < nstages) load_stage(st + 2, regB); });
            __syncthreads();
            if (st + 1 < nstages) {
                compute_stage(st + 1, [&] { if (st + 2 < nstages) store_stage(st + 2, regB); if (st + 3 < nstages) load_stage(st + 3, regA); });
                __syncthreads();
            }
        }
    } else {
        __syncthreads();
        for (int st = 0; st < nstages; st++) {
            if (st + 1 < nstages) load_stage(st + 1, regA);
            compute_stage(st, [] {});
            if (st + 1 < nstages) store_stage(st + 1, regA);
            __syncthreads();
        }
    }

    // recombine (mod 2^128, exact because the true sum is < 2^128), reduce once, add the bias, store
#pragma unroll
    for (int r = 0; r < PX * 2; r++) {
        const int mm = m0 + wm * PX + (r >> 1), c = r & 1;
        if (mm >= M) continue;
        const int b = mm / a.P, p = mm % a.P;
#pragma unroll
        for (int f = 0; f < FT; f++) {
            const int ff = f0 + wn * FT + f;
            if (ff >= a.F) continue;
            const u32 ov = OV[r][f];
            u64 v;
            {
            // a_j = A_j + ov_j * 2^63  as (lo, hi)
            u64 o0 = ov & 1023, o1 = (ov >> 10) & 1023, o2 = (ov >> 20) & 1023;
            u64 a0l = A0[r][f] + (o0 << 63), a0h = (o0 >> 1) + (a0l < A0[r][f]);
            u64 a1l = A1[r][f] + (o1 << 63), a1h = (o1 >> 1) + (a1l < A1[r][f]);
            u64 a2l = A2[r][f] + (o2 << 63), a2h = (o2 >> 1) + (a2l < A2[r][f]);
            // mid = a1 - a0 - a2
            u64 ml = a1l - a0l, mh = a1h - a0h - (a1l < a0l);
            u64 ml2 = ml - a2l; mh = mh - a2h - (ml < a2l); ml = ml2;
            // V = a0 + mid << 28 + a2 << 56
            u64 vl = a0l, vh = a0h;
            u64 tl = ml << 28, th = (mh << 28) | (ml >> 36);
            u64 nl = vl + tl; vh += th + (nl < vl); vl = nl;
            tl = a2l << 56; th = (a2h << 56) | (a2l >> 8);
            nl = vl + tl; vh += th + (nl < vl); vl = nl;
            v = barrett128(vl, vh, m);
            }
            if (c == 0 && a.bias) { const u64 bv = a.bias[(size_t)ff * kn + rown]; v = addmod(v, bv, m.q); }
            a.y[(((size_t)b * a.F + ff) * a.P + p) * ctw + (size_t)c * kn + rown] = a.yp ? split28(v) : v;
        }
    }
}

// mac3_kernel (the default): mac2_kernel with the operand staging done by LDS-DMA (global_load_lds_dwordx4: global -> LDS, no VGPR
// hop, no ds_write).  The LDS image holds the raw residues (the 28-bit split moves to the consumer side); the ~45 VGPRs this frees
// pay for longer stages: S = 4 reduction steps per stage in two LDS buffers (115-128 KiB), i.e. half the barriers per MAC, with the
// loads of stage st+1 in flight during stage st.  A wave waits for its own loads (s_waitcnt vmcnt(0)) before the one barrier per
// stage.  Terms past T read a row of zeros.  Same arithmetic, same output as mac2_kernel; +7 % on the large layers.
template <int PX, int FT, int WM, int WN, int S, bool XP, bool WP>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) mac3_kernel(MacArgs a)
{
    constexpr int NW = WM * WN, MT = PX * WM, ROWS = 2 * MT, FW = FT * WN;
    constexpr int VEC = S * (ROWS + FW), NPAIR = VEC / 2, RLOAD = 2 * ((NPAIR + NW - 1) / NW);      // operand vectors are staged in (even, odd) pairs
    static_assert(FT % 2 == 0 && (ROWS + FW) % 2 == 0, "pairs");
    extern __shared__ __attribute__((aligned(16))) u64 smem[];      // 2 x [S][ROWS + FW vectors][64 slots] raw residues, then the term table, then 1 KiB dump
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave / WN, wn = wave % WN;
    const int n = a.n, k = a.k;
    const int M = a.B * a.P;
    const int mtiles = (M + MT - 1) / MT, ftiles = (a.F + FW - 1) / FW, sbt = (n >> 6) * k;
    // XCD-aware decode: consecutive workgroup ids go to different XCDs (round-robin dispatch), so give every XCD its own
    // slot blocks and walk (f tile fastest, then m tile) inside one slot block
    int g = blockIdx.x, sb, tile;
    if ((sbt & 7) == 0) { const int xcd = g & 7, r = g >> 3, per = sbt >> 3; sb = xcd * per + r / (mtiles * ftiles); tile = r % (mtiles * ftiles); }
    else { sb = g / (mtiles * ftiles); tile = g % (mtiles * ftiles); }
    // walk order inside a slot block: filter tiles fastest (neighbouring workgroups share the pixel tile's x rows) or, for layers with
    // few pixel tiles (dense), pixel tiles fastest (neighbours share the filter tile's weight rows, which are then read from HBM once)
    const int ft = a.mt_fastest ? tile / mtiles : tile % ftiles, mt = a.mt_fastest ? tile % mtiles : tile / ftiles;
    const int i = sb / (n >> 6), s = ((sb % (n >> 6)) << 6) + lane;
    const int m0 = mt * MT, f0 = ft * FW;
    const size_t rown = (size_t)i * n + s, kn = (size_t)k * n, ctw = 2 * kn;
    const ModParams m = a.mods[i];

    // wave-uniform base pointer of every operand vector this wave stages (kept in SGPRs; lanes add (i*n+s)*8)
    const u64 *vbase[RLOAD]; int vstep[RLOAD]; bool visx[RLOAD];
#pragma unroll
    for (int j = 0; j < RLOAD; j++) {
        const int v = 2 * (wave + (j >> 1) * NW) + (j & 1);
        const int vv = min(v, VEC - 1);                  // surplus slots (NPAIR not a multiple of NW) re-load the last vector, never stored
        const int e = vv % (ROWS + FW);
        vstep[j] = vv / (ROWS + FW);
        if (e < ROWS) {
            const int mm = min(m0 + (e >> 1), M - 1), b = mm / a.P, p = mm % a.P;
            vbase[j] = a.x + ((size_t)b * a.in_cts + a.xoff[p]) * ctw + (size_t)(e & 1) * kn; visx[j] = true;
        } else {
            const int f = min(f0 + (e - ROWS), a.F - 1);     // filters past F are computed on a clamped copy and never stored
            vbase[j] = a.w + (size_t)f * a.T * kn; visx[j] = false;
        }
    }

    // wide staging loads: ONE global_load_dwordx4 fetches two vectors of a pair -- lanes 0..31 take slots (2l, 2l+1) of the even
    // vector, lanes 32..63 the same slots of the odd one (the cost of staging is per memory INSTRUCTION, see DESIGN.md ablation)
    const int half = lane >> 5, l2 = (lane & 31) * 2;
    const u32 rown2 = (u32)((size_t)i * n + (s - lane) + l2);
    const u32 hmask = half ? 0xffffffffu : 0u;
    u32 vdelta[RLOAD / 2];                                   // element distance even -> odd vector of a pair (wave-uniform)
#pragma unroll
    for (int jj = 0; jj < RLOAD / 2; jj++) vdelta[jj] = (u32)(vbase[2 * jj + 1] - vbase[2 * jj]);

    u64 A0[PX * 2][FT], A1[PX * 2][FT], A2[PX * 2][FT]; u32 OV[PX * 2][FT];
#pragma unroll
    for (int r = 0; r < PX * 2; r++)
#pragma unroll
        for (int f = 0; f < FT; f++) { A0[r][f] = 0; A1[r][f] = 0; A2[r][f] = 0; OV[r][f] = 0; }

    constexpr int NBUF = 2, NLD = RLOAD / 2;                                 // every wave issues NLD loads per stage (surplus ones land in the dump)
    // per-term x offsets: the padded toffw table is copied into LDS once (scalar loads inside the loop would share lgkmcnt
    // with the LDS operand reads and stall them); lanes read the same word (broadcast)
    u32 *tw = reinterpret_cast<u32 *>(smem + (size_t)NBUF * VEC * 64);
    for (int t = threadIdx.x; t < a.T + 8; t += blockDim.x) tw[t] = a.toffw[t];
    u64 *dump = smem + (size_t)NBUF * VEC * 64 + ((a.T + 8 + 3) / 4) * 2;
    __syncthreads();                                 // the table is read by every wave from the first issue on
    const u32 kn32 = (u32)kn;
    const char *zrow = reinterpret_cast<const char *>(a.zero) + lane * 16;
    auto issue_one = [&](int st, int jj) {
        const int t = min(st * S + vstep[2 * jj], a.T - 1);
        const u32 off = visx[2 * jj] ? tw[t] : (u32)t * kn32;
        const bool dead = !visx[2 * jj] && st * S + vstep[2 * jj] >= a.T;                 // weights past the last term are zero (x may be anything valid)
        const char *src = dead ? zrow : reinterpret_cast<const char *>(vbase[2 * jj]) + ((size_t)(off + rown2 + (hmask & vdelta[jj])) << 3);
        const int pr = wave + jj * NW;
        u64 *dst = pr < NPAIR ? smem + ((size_t)(st % NBUF) * VEC + 2 * pr) * 64 : dump;  // lanes 0..31 -> vector 2pr, lanes 32..63 -> vector 2pr+1
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };
    auto issue_stage = [&](int st) {
#pragma unroll
        for (int jj = 0; jj < NLD; jj++) issue_one(st, jj);
    };
    // the staging traffic of a stage (LDS writes of stage st+1, global loads of stage st+2) is issued INSIDE the compute of stage
    // st, at a different reduction step in the two waves that share a SIMD (waves w and w+4): while one wave issues its memory
    // instructions the other one has multiply-adds to issue, instead of all eight waves staging at the same moment
    const int io_sel = (wave >> 2) & 1;
    const int io_step = io_sel ? S / 2 : 0;
    auto compute_stage = [&](int st, auto &&io) {
        const u64 *buf = smem + (size_t)(st % NBUF) * VEC * 64;
#pragma unroll
        for (int step = 0; step < S; step++) {
            if (step == io_step) io();
            const u64 *sv = buf + step * (ROWS + FW) * 64 + lane;
            u32 w0[FT], w1[FT], ws[FT];
#pragma unroll
            for (int f = 0; f < FT; f++) { const u64 wr = sv[(ROWS + wn * FT + f) * 64]; const u64 wv = WP ? wr : split28(wr); w0[f] = (u32)wv;
                w1[f] = (u32)(wv >> 32); ws[f] = w0[f] + w1[f]; }
#pragma unroll
            for (int r = 0; r < PX * 2; r++) {
                const u64 xr = sv[(wm * PX * 2 + r) * 64]; const u64 xv = XP ? xr : split28(xr);
                const u32 x0 = (u32)xv, x1 = (u32)(xv >> 32), xs = x0 + x1;
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    A0[r][f] += (u64)x0 * w0[f];
                    A2[r][f] += (u64)x1 * w1[f];
                    A1[r][f] += (u64)xs * ws[f];
                }
            }
        }
        if ((st + 1) % (32 / S) == 0) {               // at most every 32 reduction steps: park bit 63 of each accumulator in the overflow word
#pragma unroll
            for (int r = 0; r < PX * 2; r++)
#pragma unroll
                for (int f = 0; f < FT; f++) {
                    OV[r][f] += (u32)(A0[r][f] >> 63) + ((u32)(A1[r][f] >> 63) << 10) + ((u32)(A2[r][f] >> 63) << 20);
                    A0[r][f] &= ~(1ULL << 63); A1[r][f] &= ~(1ULL << 63); A2[r][f] &= ~(1ULL << 63);
                }
        }
    };

    const int nstages = (a.T + S - 1) / S;
    constexpr int WAIT_VM0 = (7 << 4) | (15 << 8);  // s_waitcnt vmcnt(0) only (gfx9 encoding: expcnt / lgkmcnt unconstrained)
    issue_stage(0);
    for (int st = 0; st < nstages; st++) {
        __builtin_amdgcn_s_waitcnt(WAIT_VM0);        // this wave's loads for stage st have landed ...
        __builtin_amdgcn_s_barrier();                // ... and everybody else's; all waves are also done reading the other buffer
        compute_stage(st, [&] { if (st + 1 < nstages) issue_stage(st + 1); });
    }

    // recombine (mod 2^128, exact because the true sum is < 2^128), reduce once, add the bias, store
#pragma unroll
    for (int r = 0; r < PX * 2; r++) {
        const int mm = m0 + wm * PX + (r >> 1), c = r & 1;
        if (mm >= M) continue;
        const int b = mm / a.P, p = mm % a.P;
#pragma unroll
        for (int f = 0; f < FT; f++) {
            const int ff = f0 + wn * FT + f;
            if (ff >= a.F) continue;
            const u32 ov = OV[r][f];
            // every modulus this kernel is launched for has the form 2^b - d (k_mac2 checks): fold every limb sum first (modarith.h)
            u64 v = mac_reduce_fold(A0[r][f], A1[r][f], A2[r][f], ov, m);
            if (c == 0 && a.bias) { const u64 bv = a.bias[(size_t)ff * kn + rown]; v = addmod(v, bv, m.q); }
            a.y[(((size_t)b * a.F + ff) * a.P + p) * ctw + (size_t)c * kn + rown] = a.yp ? split28(v) : v;
        }
    }
}

template <int PX, int FT, int WM, int WN, int S, bool XP, bool WP>
static int mac3_launch(crc_ctx *c, MacArgs &a, hipStream_t st)
{
    constexpr int MT = PX * WM, FW = FT * WN, VEC = S * (2 * MT + FW);
    const int M = a.B * a.P;
    const size_t grid = (size_t)((M + MT - 1) / MT) * ((a.F + FW - 1) / FW) * (size_t)((c->n >> 6) * c->k);
    if (grid > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    if (a.T + 8 > 16384) return CRC_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * VEC * 64 * 8 + (size_t)((a.T + 8 + 3) / 4) * 16 + 1024;
    if (lds > 160 * 1024) return CRC_ERR_UNSUPPORTED;
    auto kern = mac3_kernel<PX, FT, WM, WN, S, XP, WP>;
    { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WM * WN), lds, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

// mac_stream_kernel (round 6): ONE image, ONE output position -- a dense layer at batch 1, the reference's whole usage model (mainparams.cpp:85-112 evaluates one
// image at a time).  Two rows (c0, c1) per weight: every weight residue is read for two multiply-adds, so the layer is a weight STREAM (PlainModelTiny's fc3 at
// n = 4096: 34 GB for 1024 outputs), the one place where north_star's HBM roofline is the bound.  mac3_kernel serves it with 64-slot workgroups, eleven of whose
// twelve pixel rows are padding, at 0.42 of 8 TB/s; here a workgroup owns SL * 256 consecutive slots of one residue row and FT filters, every lane SL adjacent
// slots (16-byte loads at SL = 2), and walks the T terms with the next term's operands requested before the current ones are multiplied.  The x rows -- the same
// for every filter group -- are shared through L2: the filter groups of a slot block run on ONE XCD next to each other (blockIdx -> (XCD, slot block, group)).
// Same limb products, same lazy sums and the same folding reduction as mac3_kernel, hence the same residues.
template <int FT, int SL, bool XP, bool WP>
__global__ void __launch_bounds__(256) mac_stream_kernel(MacArgs a)
{
    typedef typename std::conditional<SL == 2, ulonglong2, u64>::type vec_t;
    const int n = a.n, k = a.k;
    const int per_row = n / (256 * SL), sbt = per_row * k, nfg = (a.F + FT - 1) / FT;
    int g = blockIdx.x, sb, fg;
    if ((sbt & 7) == 0) { const int xcd = g & 7, r = g >> 3, per = sbt >> 3; sb = xcd * per + r / nfg; fg = r % nfg; }
    else { sb = g / nfg; fg = g % nfg; }
    const int i = sb / per_row, s = (sb % per_row) * 256 * SL + threadIdx.x * SL;
    const size_t rown = (size_t)i * n + s, kn = (size_t)k * n, ctw = 2 * kn;
    const ModParams m = a.mods[i];
    const int f0 = fg * FT;
    const u64 *xb = a.x + (size_t)a.xoff[0] * ctw + rown;
    const u64 *wb[FT];
#pragma unroll
    for (int f = 0; f < FT; f++) wb[f] = a.w + (size_t)min(f0 + f, a.F - 1) * a.T * kn + rown;        // filters past F: a clamped copy, never stored

    u64 A0[SL][2][FT], A1[SL][2][FT], A2[SL][2][FT]; u32 OV[SL][2][FT];
#pragma unroll
    for (int e = 0; e < SL; e++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int f = 0; f < FT; f++) { A0[e][c][f] = 0; A1[e][c][f] = 0; A2[e][c][f] = 0; OV[e][c][f] = 0; }

    auto load = [&](int t, vec_t (&xv)[2], vec_t (&wv)[FT]) {
        const u64 *xt = xb + a.toffw[t];
        xv[0] = *reinterpret_cast<const vec_t *>(xt); xv[1] = *reinterpret_cast<const vec_t *>(xt + kn);
#pragma unroll
        for (int f = 0; f < FT; f++) wv[f] = *reinterpret_cast<const vec_t *>(wb[f] + (size_t)t * kn);
    };
    auto lane_of = [](const vec_t &v, int e) -> u64 { if constexpr (SL == 2) return e ? v.y : v.x; else return v; };
    auto compute = [&](const vec_t (&xv)[2], const vec_t (&wv)[FT]) {
#pragma unroll
        for (int e = 0; e < SL; e++) {
            u32 w0[FT], w1[FT], ws[FT];
#pragma unroll
            for (int f = 0; f < FT; f++) { const u64 wr = lane_of(wv[f], e); const u64 wp = WP ? wr : split28(wr); w0[f] = (u32)wp; w1[f] = (u32)(wp >> 32);
                ws[f] = w0[f] + w1[f]; }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const u64 xr = lane_of(xv[c], e); const u64 xp = XP ? xr : split28(xr);
                const u32 x0 = (u32)xp, x1 = (u32)(xp >> 32), xs = x0 + x1;
#pragma unroll
                for (int f = 0; f < FT; f++) { A0[e][c][f] += (u64)x0 * w0[f]; A2[e][c][f] += (u64)x1 * w1[f]; A1[e][c][f] += (u64)xs * ws[f]; }
            }
        }
    };
    vec_t xc[2], wc[FT], xn[2], wn[FT];
    load(0, xc, wc);
    for (int t = 0; t < a.T; t++) {
        if (t + 1 < a.T) load(t + 1, xn, wn);
        compute(xc, wc);
        if ((t & 31) == 31) {                            // at most every 32 terms: park bit 63 of each accumulator in the overflow word (as mac3_kernel)
#pragma unroll
            for (int e = 0; e < SL; e++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int f = 0; f < FT; f++) {
                        OV[e][c][f] += (u32)(A0[e][c][f] >> 63) + ((u32)(A1[e][c][f] >> 63) << 10) + ((u32)(A2[e][c][f] >> 63) << 20);
                        A0[e][c][f] &= ~(1ULL << 63); A1[e][c][f] &= ~(1ULL << 63); A2[e][c][f] &= ~(1ULL << 63);
                    }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) xc[c] = xn[c];
#pragma unroll
        for (int f = 0; f < FT; f++) wc[f] = wn[f];
    }
#pragma unroll
    for (int f = 0; f < FT; f++) {
        const int ff = f0 + f;
        if (ff >= a.F) continue;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            u64 v[SL];
#pragma unroll
            for (int e = 0; e < SL; e++) {
                v[e] = mac_reduce_fold(A0[e][c][f], A1[e][c][f], A2[e][c][f], OV[e][c][f], m);
                if (c == 0 && a.bias) v[e] = addmod(v[e], a.bias[(size_t)ff * kn + rown + e], m.q);
                if (a.yp) v[e] = split28(v[e]);
            }
            u64 *dst = a.y + (size_t)ff * ctw + (size_t)c * kn + rown;       // B = P = 1: y[f][c][i][s]
            if constexpr (SL == 2) *reinterpret_cast<ulonglong2 *>(dst) = ulonglong2{v[0], v[1]}; else *dst = v[0];
        }
    }
}

template <int FT, int SL, bool XP, bool WP>
static int mac_stream_launch(crc_ctx *c, MacArgs &a, hipStream_t st)
{
    const size_t grid = (size_t)(c->n / (256 * SL)) * c->k * (size_t)((a.F + FT - 1) / FT);
    if (grid > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL((mac_stream_kernel<FT, SL, XP, WP>), dim3((unsigned)grid), dim3(256), 0, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

template <int PX, int FT, int WM, int WN, int S, int DEPTH = 2>
static int mac2_launch(crc_ctx *c, MacArgs &a, hipStream_t st)
{
    constexpr int MT = PX * WM, FW = FT * WN, VEC = S * (2 * MT + FW);
    const int M = a.B * a.P;
    const size_t grid = (size_t)((M + MT - 1) / MT) * ((a.F + FW - 1) / FW) * (size_t)((c->n >> 6) * c->k);
    if (grid > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    if (a.T + 8 > 16384) return CRC_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * VEC * 64 * 8 + (size_t)(a.T + 8) * 4;
    auto kern = mac2_kernel<PX, FT, WM, WN, S, DEPTH>;
    { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * WM * WN), lds, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}

int k_mac2(crc_ctx *c, const u64 *x, const u64 *w, u64 *y, const int *d_xoff, const int *d_toff, int B, int P, int F, int T, int in_cts,
           const u64 *bias_ntt, int gxd, int gyd, int gxf, int gyf, const unsigned *d_toffw, hipStream_t st, int xp, int wp, int yp)
{
    if (B == 0 || P == 0 || F == 0) return CRC_OK;
    int maxbits = 0; for (int i = 0; i < c->k; i++) if ((int)c->tabs[i].m.bits > maxbits) maxbits = c->tabs[i].m.bits;
    // operands are addressed as wave-uniform base + 32-bit ELEMENT offset (term offset + half-wave pair offset + slot), widened to bytes per lane
    const size_t kn1 = (size_t)c->k * c->n;
    const bool off32 = ((size_t)2 * T + 3) * kn1 < (1ull << 32) && ((size_t)2 * in_cts + 3) * kn1 < (1ull << 32);
    if (maxbits > 55 || T > 16000 || c->n < 64 || !off32) return k_mac(c, x, w, y, d_xoff, d_toff, B, P, F, T, in_cts, bias_ntt, st, xp, wp, yp);
    MacArgs a{};
    a.x = x; a.w = w; a.y = y; a.mods = c->d_mods; a.xoff = d_xoff; a.toff = d_toff; a.toffw = d_toffw;
    a.n = c->n; a.k = c->k; a.B = B; a.P = P; a.F = F; a.T = T; a.in_cts = in_cts; a.bias = bias_ntt; a.bias_sign = 1;
    a.gxd = gxd; a.gyd = gyd; a.gxf = gxf; a.gyf = gyf; a.xp = xp; a.wp = wp; a.yp = yp;
#ifdef CRC_TUNING
    a.dbg = c->tune.mac2_dbg;
#endif
    // tile configuration <PX, FT, WM, WN, S>: a workgroup covers PX*WM pixels x FT*WN filters (6 x 16 or 12 x 8), 24 accumulators
    // per wave either way; pick the shape that wastes fewer multiply-adds on filter/pixel padding (F = 50 -> 56 instead of 64,
    // F = 20 -> 24 instead of 32; measured equal on full tiles).  CRC_MAC2_CFG forces one (tools/bench_mac.py)
    const int cfg = c->tune.mac2_cfg;
    int pick = cfg;
    if (pick == 0) {
        const long long M = (long long)B * P;
        auto padded = [&](int mt, int fw) { return ((M + mt - 1) / mt * mt) * (((long long)F + fw - 1) / fw * fw); };
        pick = padded(12, 8) * 100 < padded(6, 16) * 98 ? 8 : 16;
    }
    a.zero = c->d_zero;
    { const int ord = c->tune.mac_order;
      const long long mtl = ((long long)B * P + 5) / 6;
      a.mt_fastest = ord >= 0 ? ord : (mtl <= 16); }
    // default: LDS-DMA staging with 4-step stages; the register-staged mac2_kernel remains for reductions whose term table does not
    // fit beside the two stage buffers (T > ~7000) and as the tuning reference (CRC_MAC_REGSTAGE=1, CRC_MAC2_CFG).  (Short reductions
    // that are not a multiple of 4 pay a padded last stage, T = 25: 28 steps; with packed operands mac3 still wins.)
    const int regstage = c->tune.mac_regstage;
    bool foldable = true;                          // mac3's epilogue is the folding reduction: needs q = 2^b - d, 50 <= b <= 55
    for (int i = 0; i < c->k; i++) if (!c->tabs[i].m.fold || c->tabs[i].m.bits < 50 || c->tabs[i].m.bits > 55) foldable = false;
    // one image, one output position (a dense layer at batch 1): the weight stream (mac_stream_kernel).  CRC_MAC_STREAM=0 keeps mac3_kernel; 1..4 pick a shape
    if (!regstage && !cfg && foldable && (long long)B * P == 1 && T >= 8 && c->tune.mac_stream != 0 && c->n >= 512) {
        // measured (profiles/r06_ab_mac_stream.txt): four filters x two slots per lane streams fc3 at 6.1-6.2 TB/s; a layer of a few filters (fc4: 10) fills the chip
        // better with two filters per workgroup
        const int shape = c->tune.mac_stream > 1 ? c->tune.mac_stream : (F >= 16 ? 1 : 2);
#define STREAM_GO(FTV, SLV) (a.xp && a.wp ? mac_stream_launch<FTV, SLV, true, true>(c, a, st) : a.xp ? mac_stream_launch<FTV, SLV, true, false>(c, a, st) \
                             : a.wp ? mac_stream_launch<FTV, SLV, false, true>(c, a, st) : mac_stream_launch<FTV, SLV, false, false>(c, a, st))
        switch (shape) {                               // (CRC_MAC_STREAM=5 forces the default shape for every F)
        case 2: return STREAM_GO(2, 2);
        case 3: return STREAM_GO(4, 1);
        case 4: return STREAM_GO(8, 1);
        default: return STREAM_GO(4, 2);
        }
#undef STREAM_GO
    }
    if (!regstage && !cfg && foldable) {
        int rc;
#define MAC3_GO(XPV, WPV) (pick == 8 ? mac3_launch<3, 4, 4, 2, 4, XPV, WPV>(c, a, st) : mac3_launch<3, 4, 2, 4, 4, XPV, WPV>(c, a, st))
        if (a.xp && a.wp) rc = MAC3_GO(true, true); else if (a.xp) rc = MAC3_GO(true, false); else if (a.wp) rc = MAC3_GO(false, true);
            else rc = MAC3_GO(false, false);
#undef MAC3_GO
        if (rc != CRC_ERR_UNSUPPORTED) return rc;
    }
    switch (pick) {
    case 8: return mac2_launch<3, 4, 4, 2, 2>(c, a, st);
    default: return mac2_launch<3, 4, 2, 4, 2>(c, a, st);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Algebraic fusion conv -> sum/avg pool (both linear over Z_q, so this is exact): pool(conv_w(x) + b) == conv_w'(x) + b'
// with the pooled kernel  w'[f][z][u][v] = div * sum_{a<pxf, b<pyf} w[f][z][u - a*cxs][v - b*cys]   (indices in range),
// b' = div * (pxf*pyf) * b, window xf' = (pxf-1)*cxs + xf, stride cxs*pxs.  For a decimating pool (stride = window) the
// fused layer has (xf'*yf')/(xf*yf) more terms but pxs*pys fewer outputs: 2.8x fewer MACs for CrCNN's 5x5 conv + 2x2/2 pool.
// All inputs/outputs in NTT form.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold_pool_kernel(const u64 *w, const u64 *bias, const u64 *div, u64 *wout, u64 *bout, const ModParams *mods, int n,
    int k,
                                                        int nf, int zd, int xf, int yf, int cxs, int cys, int pxf, int pyf, int xf2, int yf2)
{
    // rows: first nf*zd*xf2*yf2*k weight rows, then nf*k bias rows
    const size_t row = blockIdx.x;
    const size_t wrows = (size_t)nf * zd * xf2 * yf2 * k;
    const int i = (int)(row % k);
    const ModParams m = mods[i];
    const u64 *dv = div ? div + (size_t)i * n : nullptr;
    if (row < wrows) {
        size_t e = row / k;
        const int v = (int)(e % yf2); e /= yf2; const int u = (int)(e % xf2); e /= xf2;     // e = f*zd + z
        u64 *dst = wout + row * (size_t)n;
        for (int s = threadIdx.x; s < n; s += blockDim.x) {
            u64 acc = 0;
            for (int a = 0; a < pxf; a++) { const int kx = u - a * cxs; if (kx < 0 || kx >= xf) continue;
                for (int b = 0; b < pyf; b++) { const int ky = v - b * cys; if (ky < 0 || ky >= yf) continue;
                    acc = addmod(acc, w[(((e * xf + kx) * yf + ky) * k + i) * (size_t)n + s], m.q); } }
            dst[s] = dv ? mulmod(acc, dv[s], m) : acc;
        }
    } else {
        const size_t br = row - wrows;                   // f*k + i
        const u64 *src = bias + br * (size_t)n; u64 *dst = bout + br * (size_t)n;
        const u64 cnt = (u64)(pxf * pyf) % m.q;
        for (int s = threadIdx.x; s < n; s += blockDim.x) {
            u64 x = mulmod(src[s], cnt, m);
            dst[s] = dv ? mulmod(x, dv[s], m) : x;
        }
    }
}

int k_fold_pool(crc_ctx *c, const u64 *w, const u64 *bias, const u64 *div, u64 *wout, u64 *bout, int nf, int zd, int xf, int yf, int cxs, int cys,
                int pxf, int pyf, hipStream_t st)
{
    const int xf2 = (pxf - 1) * cxs + xf, yf2 = (pyf - 1) * cys + yf;
    const size_t rows = ((size_t)nf * zd * xf2 * yf2 + nf) * c->k;
    hipLaunchKernelGGL(fold_pool_kernel, dim3((unsigned)rows), dim3(256), 0, st, w, bias, div, wout, bout, c->d_mods, c->n, c->k, nf, zd, xf, yf, cxs, cys,
        pxf, pyf, xf2, yf2);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
