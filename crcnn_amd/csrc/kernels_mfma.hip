// kernels_mfma.hip -- the ct x pt multiply-accumulate of convolution / dense layers as an int8 limb GEMM on the matrix cores.
//
// Why.  mac3_kernel (kernels.hip) sits at 88 % of the v_mad_u64_u32 rate: gfx950 has no 64 x 64 multiplier and the vector ALU gives 5.7 T modular
// multiply-adds per second, 3.5 % of what HBM could feed.  Per residue i and slot s a layer is a GEMM over Z_q,
//       Y[m][f] = sum_t A[m][t] W[t][f]        m = (image, output pixel, poly),  t = (tap, channel),
// and the integer MFMA unit does exact int8 x int8 -> int32 dot products 49x faster than that.  Every residue r < q < 2^55 is written as its centred
// representative r' in (-q/2, q/2] (same class mod q) in balanced base 256,  r' = sum_{l<7} d_l 256^l, d_l in [-128, 127] (|d_6| <= 64).  Then
// x w = sum_{l,m} a_l b_m 256^(l+m): 49 limb products on 13 diagonals l+m.  One v_mfma_i32_16x16x64_i8 forms a 16 x 16 tile of 64-term dot products
// of one (l, m) pair into diagonal l+m's int32 accumulator -- exact while T 7 128^2 < 2^31 (T <= 18 000).  After the reduction loop
// V = sum_d D_d 2^(8d) is reduced mod q ONCE per output (diag_reduce: biased accumulators, one Montgomery step; the weights carry 2^64 mod q).  Exact integer
// arithmetic, the same element of Z_q, hence the same bits as the reference (convolutionalLayer.cpp:56-93 / fullyConnectedLayer.cpp:113-168) and as mac3_kernel;
// measured 4x faster on CrCNN's conv2+pool2 (profiles/r02_*).
//
// Layouts (CRC_NTTL, "limb form"; slot = i*n + s).  The GEMMs of different slots share nothing, so operands are SLOT-MAJOR here (the rest of the engine
// is slot-minor: one row = n slots of one residue):
//     tensor   Xl [slot][B][7 planes][positions][2 polys][zdp]          int8, zdp = channels rounded up to 32 (zero padded); a DENSE layer's input (one position) is
//                 K-blocked instead: [slot][7 planes][zdp / 32][B * 2 rows = (image, poly)][32], so that a reduction step's 64 rows are 2 KiB contiguous per plane
//                 (channel-innermost rows 1 KiB apart filled a quarter of every cache line the LDS-DMA touched: fc3 ran 20 % below the convolution's rate)
//     weights  Wl [slot][reduction step = (tap, 32-channel block)][7 planes][Fp][32]   int8, Fp = filters rounded up to 64 (zero padded), an odd number of
//                 steps rounded up to even with a zero step; every weight times 2^64 mod q
//     FLAT form (round 4; layers of fewer than 32 channels, CrCNN's second convolution has 20): padding every tap's channels to 32 made such a layer's GEMM 20/32
//                 useful work at best (ApproxPlainModel's conv2: 10 steps for 180 terms, 44 % with the filter padding).  There the tensor is
//                 Xl [slot][B][7 planes][2 polys][positions][zdc], zdc = channels rounded up to 4: for one window row kx the (ky, channel) terms of an output pixel are
//                 yf zdc CONTIGUOUS bytes, and the reduction runs over that flattened run in 32-byte steps -- step = (kx, 32-byte piece of the run), S = ceil(yf zdc / 32)
//                 steps per window row (3 x 20 = 60 bytes: 2 steps per row, 6 instead of 10 for that layer).  The piece that sticks out of the run reads the next
//                 position's bytes and meets zero weights.  Weights: same container, steps in that order.  The LDS-DMA pieces are then only 4-byte aligned in memory
//                 (tools/lds_dma_unaligned.hip).  limb_flat() is the one rule every producer and consumer asks.
//     result   Ys [slot][B][F][P][2] u64 canonical (internal), then transposed to the slot-minor tensor layout or re-limbed for a dense consumer
// Workgroup = one slot, 64 rows x 64 filters: 4 waves (one per SIMD), a 32 x 32 tile = 2 x 2 sub-tiles x 13 diagonals x 4 int32 accumulators (208 AGPRs) each.
// Per 32-term reduction step the workgroup stages 64 x 32 B x 7 planes of A -- implicit im2col: each lane's LDS-DMA piece reads its own (pixel + tap) address, no
// patch matrix exists -- and as much of W (contiguous) into a 5-slot LDS ring (140 KiB), three steps ahead of use; an MFMA spans two steps: 196 per wave and barrier.
#include "kernels.h"
#include "limbred.h"
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef signed char i8;

#define NPL 7                                    // limb planes
#define TILE_B (NPL * 64 * 32)                   // 14336: one operand tile of a reduction step (A: 64 rows, W: 64 filters), [plane][row][32 B]

struct MfmaArgs {
    const i8 *xl; const i8 *wl; u64 *ys; const ModParams *mods; const u64 *bias;     // bias: NTT-form delta rows [F][k][n] added to poly 0, or null
    int xcdmap;                                        // grid x carries the XCD in its low three bits (the slot count is a multiple of 8)
    int mfast;                                         // tile order inside a slot: row tiles fastest (neighbouring workgroups share the weight tile) instead of filter tiles fastest
    i8 *xl_out; int lp2; unsigned zdp_out;             // direct limb result for a dense consumer (2P = 2^lp2 divides 64): [slot][B][7][2][zdp_out], channel = f P + p
    int n, k, B, zdp, npos, yd, xs, ys_, yf, yo, P, F, Fp, zblks, ksteps, M, mtiles, ntiles;      // ksteps: rounded up to even (the weights carry a zero step)
    unsigned img_bytes; unsigned long long wslot_bytes; int ksteps_real;
    int flat, S, zdc; unsigned fplane;                 // flat form: S steps per window row, zdc channel bytes per position, fplane = 2 npos zdc bytes per plane
    int acc0[8][13];                                   // initial value of the 13 diagonal accumulators, per modulus (limb_tables)
    u64 qinv[8];                                       // q^-1 mod 2^64
};

// canonical residue -> 7 balanced base-256 digits of its centred representative
__device__ __forceinline__ void limb_digits(u64 r, u64 q, int (&d)[NPL])
{
    long long v = r > (q >> 1) ? (long long)r - (long long)q : (long long)r;
#pragma unroll
    for (int l = 0; l < NPL; l++) { d[l] = (int)(signed char)(v & 0xff); v = (v - d[l]) >> 8; }
}

// v_mfma_i32_16x16x64_i8: a wave's 32 x 32 tile as 2 x 2 sub-tiles, two 32-term reduction steps per instruction.  Same cycles per product as v_mfma_i32_32x32x32_i8
// (this kernel's first form: tools/mfma_mac.hip), but the chip holds a higher clock on this shape under random operands (tools/mfma_shape.hip: 3.6 vs 3.3 Pop/s in a
// bare loop), and there is twice the work between barriers.  A K group g = lane / 16 of an operand fragment is half (g & 1) of ring slot 2d + (g >> 1): staging stays
// one slot per 32-term step.
template <int NST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) mfma_mac_kernel(MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) i8 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    const int slots = a.n * a.k, per = a.mtiles * a.ntiles;
    int g = blockIdx.x, slot, tile;
    if ((slots & 7) == 0) { const int xcd = g & 7, r = g >> 3; slot = xcd * (slots >> 3) + r / per; tile = r % per; }
    else { slot = g / per; tile = g % per; }
    const int mt = a.mfast ? tile % a.mtiles : tile / a.ntiles, nt = a.mfast ? tile / a.mtiles : tile % a.ntiles;
    const int i = slot / a.n, s = slot % a.n;
    const int m0 = mt * 64, f0 = nt * 64;
    const ModParams m = a.mods[i];
    const u64 qinv = a.qinv[i];
    const i8 *xs = a.xl + (size_t)slot * a.B * a.img_bytes;
    const i8 *ws = a.wl + (size_t)slot * a.wslot_bytes + (size_t)f0 * 32;
    u32 src_off[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
        const int pc = wave + 4 * j;
        if (pc < 14) {
            const int c16 = pc * 64 + lane;
            const int plane = c16 >> 7, row = (c16 >> 1) & 63, half = c16 & 1;
            const int mm = min(m0 + row, a.M - 1);
            const int b = mm / (2 * a.P), p = (mm >> 1) % a.P, c = mm & 1;
            const int ox = p / a.yo, oy = p % a.yo;
            src_off[j] = a.npos == 1 ? (u32)(plane * a.zblks * (2 * a.B) + mm) * 32 + half * 16        // dense input: K-blocked rows (mm = image * 2 + poly)
                       : a.flat ? (u32)b * a.img_bytes + (u32)(plane * a.fplane + (c * a.npos + (ox * a.xs) * a.yd + oy * a.ys_) * a.zdc + half * 16)
                                : (u32)b * a.img_bytes + (u32)(plane * (a.npos * 2 * a.zdp) + (((ox * a.xs) * a.yd + oy * a.ys_) * 2 + c) * a.zdp + half * 16);
        } else src_off[j] = (u32)(((pc - 14) >> 1) * (a.Fp * 32) + ((pc - 14) & 1) * 1024 + lane * 16);
    }
    const int kreal = a.ksteps_real;
    auto issue = [&](int ks) {
        const int ka = min(ks, kreal - 1);                        // the padding step multiplies by zero weights: any valid rows will do
        const int tap = ka / a.zblks, zb = ka - tap * a.zblks;
        const int kx = tap / a.yf, ky = tap - kx * a.yf;
        const int fkx = ka / a.S;                                    // flat form: window row, piece of its run
        const u32 delta = a.npos == 1 ? (u32)zb * (2 * a.B) * 32 : a.flat ? (u32)(fkx * a.yd * a.zdc + (ka - fkx * a.S) * 32) : (u32)((kx * a.yd + ky) * 2 * a.zdp + zb * 32);
        i8 *dst = lds + (ks % NST) * (2 * TILE_B);
        const i8 *wt = ws + (size_t)ks * (NPL * a.Fp * 32);
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int pc = wave + 4 * j;
            const i8 *src = pc < 14 ? xs + src_off[j] + delta : wt + src_off[j];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(dst + pc * 1024), 16, 0, 0);
        }
    };
    v4i acc[2][2][13];
#pragma unroll
    for (int rs = 0; rs < 2; rs++)
#pragma unroll
        for (int cs = 0; cs < 2; cs++)
#pragma unroll
            for (int d = 0; d < 13; d++) { const int b0 = a.acc0[i][d]; acc[rs][cs][d] = v4i{b0, b0, b0, b0}; }
    const int K = a.ksteps;                                       // even
#pragma unroll
    for (int j = 0; j < NST - 2; j++) if (j < K) issue(j);
    const int kg = lane >> 4, r16 = lane & 15;
    const int fragA = (wm * 32 + r16) * 32 + (kg & 1) * 16, fragW = (wn * 32 + r16) * 32 + (kg & 1) * 16;
    for (int ks = 0; ks < K; ks += 2) {
        // steps ks, ks + 1 have landed; up to NST - 4 younger steps (7 loads each) may still be in flight
        const int younger = min(NST - 4, K - 2 - ks);
        if (younger >= 1) __builtin_amdgcn_s_waitcnt(7 | (7 << 4) | (15 << 8));
        else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);                       // (the wait immediates count loads in issue order: keep the two steps' pieces apart and in order)
        if (ks + NST - 2 < K) { issue(ks + NST - 2); __builtin_amdgcn_sched_barrier(0); if (ks + NST - 1 < K) issue(ks + NST - 1); }
        __builtin_amdgcn_sched_barrier(0);
        const i8 *tA = lds + ((ks + (kg >> 1)) % NST) * (2 * TILE_B), *tW = tA + TILE_B;
        // the weight fragments and the first two A fragments are requested up front, then A fragment g + 2 behind the MFMAs of fragment g: a read has two groups of
        // 14 MFMAs to land (read just in time, each fragment's LDS latency sat exposed in front of its MFMAs; all 28 up front overflow the 4-bit lgkm counter and the
        // compiler waits for everything).  The scheduling barriers pin this order.
        v4i w[2][NPL], av[2 * NPL];
        auto read_a = [&](int gi) { return *reinterpret_cast<const v4i *>(tA + (gi % NPL) * (64 * 32) + fragA + (gi / NPL) * (16 * 32)); };
#pragma unroll
        for (int cs = 0; cs < 2; cs++)
#pragma unroll
            for (int l = 0; l < NPL; l++) w[cs][l] = *reinterpret_cast<const v4i *>(tW + l * (64 * 32) + fragW + cs * (16 * 32));
        av[0] = read_a(0); av[1] = read_a(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < 2 * NPL; gi++) {
            const int rs = gi / NPL, l = gi % NPL;
#pragma unroll
            for (int cs = 0; cs < 2; cs++)
#pragma unroll
                for (int mm = 0; mm < NPL; mm++)
                    acc[rs][cs][l + mm] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[gi], w[cs][mm], acc[rs][cs][l + mm], 0, 0, 0);
            if (gi + 2 < 2 * NPL) av[gi + 2] = read_a(gi + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: C/D layout of a 16 x 16 tile: col = lane & 15, row = 4 (lane >> 4) + reg.  Row mm = image b, pixel p, poly c (= reg & 1: the row bases are multiples
    // of 4) lands at Ys[((slot B + b) F + f) 2P + (mm - b 2P)] = Ys[slot B F 2P + b (F - 1) 2P + f 2P + mm]: one division per group of four rows.
    // With a dense consumer behind (xl_out; 2P = 2^lp2 divides 64, so the tile is 64 / 2P whole images) the balanced digits go straight into that layer's limb tensor
    // -- channel = f P + p: per image, plane and poly the tile owns one run of 64 P bytes -- staged in the (now idle) ring as [image][plane][poly][64 P] and copied out
    // in 16-byte pieces: no slot-major u64 result, no conversion kernel (9 % of the conv2+pool2 layer call)
    u64 *yslot = a.ys + (size_t)slot * a.B * a.F * (2 * a.P);
    const u32 P2 = 2 * a.P;
    const u32 RL = 64 * a.P;                                      // bytes of one run
    if (a.xl_out) __syncthreads();                                // everybody has read its last fragments: the ring becomes the staging area
#pragma unroll
    for (int rs = 0; rs < 2; rs++) {
        const u32 mbase = m0 + wm * 32 + rs * 16 + 4 * kg, bb = mbase / P2, rem = mbase - bb * P2;
#pragma unroll
        for (int cs = 0; cs < 2; cs++) {
            const int fl = wn * 32 + cs * 16 + r16, f = f0 + fl;
            const u64 bv = (a.bias && f < a.F) ? a.bias[((size_t)f * a.k + i) * a.n + s] : 0;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                int D[13];
#pragma unroll
                for (int d = 0; d < 13; d++) D[d] = acc[rs][cs][d][reg];
                u64 v = diag_reduce_w(D, m, qinv);
                if ((reg & 1) == 0) v = addmod(v, bv, m.q);
                if (a.xl_out) {
                    const u32 rt = wm * 32 + rs * 16 + 4 * kg + reg, bl = rt >> a.lp2, qq = rt & (P2 - 1);        // row in the tile -> image in the tile, (pixel, poly)
                    const u64 dg = f < a.F ? balanced_digit_bytes(v, m.q) : 0;          // the 7 balanced digits, one per byte; filters past F: zero padding of the consumer's channels
                    i8 *sp = lds + (size_t)bl * (NPL * 2 * RL) + (qq & 1) * RL + fl * a.P + (qq >> 1);
#pragma unroll
                    for (int l = 0; l < NPL; l++) sp[(size_t)l * (2 * RL)] = (i8)(dg >> (8 * l));
                } else {
                    const u32 r = rem + reg, b = bb + (r >= P2) + (r >= 2 * P2);
                    if (mbase + reg < (u32)a.M && f < a.F) yslot[b * (u32)(a.F - 1) * P2 + (u32)f * P2 + mbase + reg] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (a.xl_out) {
        __syncthreads();
        const u32 imgs = 64 >> a.lp2, b0 = (u32)m0 >> a.lp2, per_run = RL / 16, pieces = imgs * NPL * 2 * per_run;
        const u32 ch0 = (u32)f0 * a.P;                            // first channel of this tile's runs
        for (u32 o = threadIdx.x; o < pieces; o += 256) {
            const u32 run = o / per_run, off = (o - run * per_run) * 16, bl = run / (NPL * 2), lc = run - bl * (NPL * 2);       // lc = plane * 2 + poly
            const u32 ch = ch0 + off;                                 // consumer layout: [plane][channel block][row = image * 2 + poly][32]
            if (b0 + bl < (u32)a.B && ch < a.zdp_out)
                *reinterpret_cast<uint4 *>(a.xl_out + (size_t)slot * ((size_t)NPL * 2 * a.B * a.zdp_out)
                                           + ((size_t)((lc >> 1) * (a.zdp_out / 32) + (ch >> 5)) * (2 * a.B) + (b0 + bl) * 2 + (lc & 1)) * 32 + (ch & 31)) =
                    *reinterpret_cast<const uint4 *>(lds + (size_t)run * RL + off);
        }
    }
}

// Two workgroups per CU.  With one workgroup per CU and one wave per SIMD (above) nothing overlaps a workgroup's start-up, its barrier skew or its epilogue: the matrix
// pipe idles for half of the cycles.  Here a wave's tile is 32 rows x 16 filters (2 x 13 x 4 = 104 accumulators), a workgroup's 64 x 32, the weight fragments come
// straight from L2 into registers (a lane's fragment is 16 contiguous bytes of Wl; they are prefetched one double step ahead) and only A goes through LDS: 5 x 14 KiB per
// workgroup, <= 256 registers per wave, so TWO workgroups share a CU and one's stalls are the other's issue slots.
#define TILE_A (NPL * 64 * 32)
// MODE: the tensor's addressing -- 0 a dense layer's K-blocked tensor (one position), 1 the flat form of a convolution with fewer than 32 channels, 2 32-channel blocks
template <int NST, int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) mfma_mac2w_kernel(MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) i8 lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    // 3-D grid (round 4: no divisions on the way to the tile): x = 8 * (fast tile index) + XCD -- the dispatcher deals consecutive workgroups to the eight XCDs in
    // turn and gridDim.x is a multiple of 8 -- y = the slow tile index, z = the slot within the XCD's share.  The workgroups of an XCD walk the tiles of one slot, then
    // of the next: its L2 holds that slot's operands meanwhile.  (Rings whose slot count is not a multiple of 8 -- none of CrCNN's -- put the slot in z as it is.)
    const int slots = a.n * a.k;
    const int fast = a.xcdmap ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, slow = (int)blockIdx.y;
    const int slot = a.xcdmap ? (int)(blockIdx.x & 7) * (slots >> 3) + (int)blockIdx.z : (int)blockIdx.z;
    const int mt = a.mfast ? fast : slow, nt = a.mfast ? slow : fast;
    const int lgn = __builtin_ctz(a.n), i = slot >> lgn, s = slot & (a.n - 1);          // (n is a power of two)
    const int m0 = mt * 64, f0 = nt * 32;
    if (f0 >= a.F) return;                                        // a 32-filter tile that is all padding (filters are padded to 64): nothing to compute or store
    const ModParams m = a.mods[i];
    const u64 qinv = a.qinv[i];
    const i8 *xs = a.xl + (size_t)slot * a.B * a.img_bytes;
    const int kg = lane >> 4, r16 = lane & 15;
    // this lane's weight fragment: filter f0 + 16 wn + r16, K group kg = half (kg & 1) of step ks + (kg >> 1)
    const i8 *wlane = a.wl + (size_t)slot * a.wslot_bytes + (size_t)(f0 + wn * 16 + r16) * 32 + (kg & 1) * 16 + (size_t)(kg >> 1) * (NPL * a.Fp * 32);
    const int wplane = a.Fp * 32, wstep = NPL * a.Fp * 32;
    // A staging: 14 LDS-DMA pieces of 1 KiB per 32-term step, pieces wave, wave + 4, ...; waves 2 and 3 load pieces 10 / 11 twice so that every wave issues four
    // loads per step (uniform counts: the waits below are immediates)
    // (round 4: a thread's four pieces are the same tile row -- piece parity = wave parity -- in four planes, so the row's image / pixel / window origin is divided
    // out ONCE, not per piece: the start-up of a tile was 690 vector + 900 scalar instructions, most of them these divisions and the ones of the step addresses below)
    u32 src_off[4]; int pcs[4];
    {
        const int row = ((wave & 1) << 5) + (lane >> 1), half = lane & 1;
        const int mm = min(m0 + row, a.M - 1);
        const int b = mm / (2 * a.P), rem = mm - b * (2 * a.P), p = rem >> 1, c = rem & 1;
        const int ox = p / a.yo, oy = p - ox * a.yo;
        const u32 rowpart = MODE == 0 ? (u32)mm * 32 + half * 16
                          : MODE == 1 ? (u32)b * a.img_bytes + (u32)((c * a.npos + (ox * a.xs) * a.yd + oy * a.ys_) * a.zdc + half * 16)
                                   : (u32)b * a.img_bytes + (u32)((((ox * a.xs) * a.yd + oy * a.ys_) * 2 + c) * a.zdp + half * 16);
        const u32 pstride = MODE == 0 ? (u32)(a.zblks * (2 * a.B)) * 32 : MODE == 1 ? a.fplane : (u32)(a.npos * 2 * a.zdp);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            pcs[j] = wave + 4 * j < 14 ? wave + 4 * j : wave + 8;
            src_off[j] = rowpart + (u32)(pcs[j] >> 1) * pstride;          // plane = piece / 2
        }
    }
    const int kreal = a.ksteps_real;
    // A(ks) is issued for ks = 0, 1, 2, ... in order (the first loads, then two per double step), so the step's address offset and ring buffer are running counters
    // instead of three divisions per step; the padding step of an odd reduction re-reads the last real one (its weights are zero)
    int iss = 0, ibuf = 0, zb = 0, ky = 0, kx = 0;
    u32 delta = 0;
    auto issue_a = [&](int ks) {
        (void)ks;                                                     // == iss
        i8 *dst = lds + ibuf * TILE_A;
#pragma unroll
        for (int j = 0; j < 4; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xs + src_off[j] + delta), (__attribute__((address_space(3))) void *)(dst + pcs[j] * 1024), 16, 0, 0);
        iss++; ibuf = ibuf + 1 == NST ? 0 : ibuf + 1;
        if (iss < kreal) {
            if (MODE == 0) delta += (u32)(2 * a.B) * 32;                                                           // next 32-channel block
            else if (MODE == 1) { if (++zb == a.S) { zb = 0; kx++; } delta = (u32)(kx * a.yd * a.zdc + zb * 32); }    // next piece of the window row's run, or next window row
            else { if (++zb == a.zblks) { zb = 0; if (++ky == a.yf) { ky = 0; kx++; } } delta = (u32)((kx * a.yd + ky) * 2 * a.zdp + zb * 32); }
        }
    };
    auto load_w = [&](int ks, v4i (&w)[NPL]) {
        const i8 *pw = wlane + (size_t)ks * wstep;
#pragma unroll
        for (int l = 0; l < NPL; l++) w[l] = *reinterpret_cast<const v4i *>(pw + (size_t)l * wplane);
    };
    v4i acc[2][13];
#pragma unroll
    for (int rs = 0; rs < 2; rs++)
#pragma unroll
        for (int d = 0; d < 13; d++) { const int b0 = a.acc0[i][d]; acc[rs][d] = v4i{b0, b0, b0, b0}; }
    const int K = a.ksteps;                                       // even
    const int fragA = (wm * 32 + r16) * 32 + (kg & 1) * 16;
    // one double step: steps ks, ks + 1 of A and the weights `wuse` (double step ks) have landed.  Issue order inside a double step: A(ks + 3), W(ks + 4) -> wload,
    // A(ks + 4); at the top of the next one everything up to A(ks + 3) is needed, so 7 + 4 loads may stay in flight: the weights are fetched TWO double steps ahead
    // (dense layers stream them from HBM), A three steps ahead.  Three weight buffers rotate through a loop unrolled by three (no register copies).
    auto dstep = [&](int ks, v4i (&wuse)[NPL], v4i (&wload)[NPL]) {
        if (ks + 2 < K) __builtin_amdgcn_s_waitcnt(11 | (7 << 4) | (15 << 8));
        else __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
        __syncthreads();
        // the immediates above count loads in ISSUE order: the scheduling barriers pin A(ks + 3), W(ks + 4), A(ks + 4) (LDS-DMA writes do not alias the register loads,
        // so nothing else stops the compiler from permuting them -- a permutation would let A(ks + 3) still be in flight when it is read)
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 3 < K) issue_a(ks + 3);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 4 < K) { load_w(ks + 4, wload); __builtin_amdgcn_sched_barrier(0); issue_a(ks + 4); }
        __builtin_amdgcn_sched_barrier(0);
        const i8 *tA = lds + ((ks + (kg >> 1)) % NST) * TILE_A;
        v4i av[2 * NPL];
        auto read_a = [&](int gi) { return *reinterpret_cast<const v4i *>(tA + (gi % NPL) * (64 * 32) + fragA + (gi / NPL) * (16 * 32)); };
        av[0] = read_a(0); av[1] = read_a(1); av[2] = read_a(2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gi = 0; gi < 2 * NPL; gi++) {
            const int rs = gi / NPL, l = gi % NPL;
#pragma unroll
            for (int mm = 0; mm < NPL; mm++)
                acc[rs][l + mm] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av[gi], wuse[mm], acc[rs][l + mm], 0, 0, 0);
            if (gi + 3 < 2 * NPL) av[gi + 3] = read_a(gi + 3);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    v4i w0[NPL], w1[NPL], w2[NPL];
    // prologue in the steady-state order: A(0), A(1), W(0), A(2), W(2)  -- the first wait leaves W(2) + A(2)'s four pieces ... at most 11 loads in flight
    issue_a(0); if (1 < K) issue_a(1);
    __builtin_amdgcn_sched_barrier(0);
    load_w(0, w0);
    __builtin_amdgcn_sched_barrier(0);
    if (2 < K) { issue_a(2); __builtin_amdgcn_sched_barrier(0); load_w(2, w1); }
    __builtin_amdgcn_sched_barrier(0);
    for (int ks = 0; ks < K; ks += 6) {
        dstep(ks, w0, w2);
        if (ks + 2 < K) dstep(ks + 2, w1, w0);
        if (ks + 4 < K) dstep(ks + 4, w2, w1);
    }
    // epilogue: 8 outputs per lane (diag_reduce_w: the reduction in 32-bit words)
    u64 *yslot = a.ys + (size_t)slot * a.B * a.F * (2 * a.P);
    const u32 P2 = 2 * a.P;
    const u32 RL = 32 * a.P;                                      // bytes of one run of a direct limb result (this tile's 32 filters)
    if (a.xl_out) __syncthreads();
    const int fl = wn * 16 + r16, f = f0 + fl;
    const u64 bv = (a.bias && f < a.F) ? a.bias[((size_t)f * a.k + i) * a.n + s] : 0;
    const u32 img_jump = (u32)(a.F - 1) * P2;                     // Ys [image][filter][2P]: rows run on inside an image's filter block, the next image's is F - 1 blocks further
#pragma unroll
    for (int rs = 0; rs < 2; rs++) {
        const u32 mbase = m0 + wm * 32 + rs * 16 + 4 * kg, bb = mbase / P2, rem = mbase - bb * P2;
        const u32 idx0 = bb * img_jump + (u32)f * P2 + mbase;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            int D[13];
#pragma unroll
            for (int d = 0; d < 13; d++) D[d] = acc[rs][d][reg];
            u64 v = diag_reduce_w(D, m, qinv);
            if ((reg & 1) == 0) v = addmod(v, bv, m.q);
            if (a.xl_out) {
                const u32 rt = wm * 32 + rs * 16 + 4 * kg + reg, bl = rt >> a.lp2, qq = rt & (P2 - 1);
                const u64 dg = f < a.F ? balanced_digit_bytes(v, m.q) : 0;
                i8 *sp = lds + (size_t)bl * (NPL * 2 * RL) + (qq & 1) * RL + fl * a.P + (qq >> 1);
#pragma unroll
                for (int l = 0; l < NPL; l++) sp[(size_t)l * (2 * RL)] = (i8)(dg >> (8 * l));
            } else {
                const u32 r = rem + reg, jump = (r >= P2 ? img_jump : 0u) + (r >= 2 * P2 ? img_jump : 0u);
                if (mbase + reg < (u32)a.M && f < a.F) yslot[idx0 + reg + jump] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (a.xl_out) {
        __syncthreads();
        const u32 imgs = 64 >> a.lp2, b0 = (u32)m0 >> a.lp2, per_run = RL / 16, pieces = imgs * NPL * 2 * per_run;
        const u32 ch0 = (u32)f0 * a.P;
        for (u32 o = threadIdx.x; o < pieces; o += 256) {
            const u32 run = o / per_run, off = (o - run * per_run) * 16, bl = run / (NPL * 2), lc = run - bl * (NPL * 2);
            const u32 ch = ch0 + off;                                 // consumer layout: [plane][channel block][row = image * 2 + poly][32]
            if (b0 + bl < (u32)a.B && ch < a.zdp_out)
                *reinterpret_cast<uint4 *>(a.xl_out + (size_t)slot * ((size_t)NPL * 2 * a.B * a.zdp_out)
                                           + ((size_t)((lc >> 1) * (a.zdp_out / 32) + (ch >> 5)) * (2 * a.B) + (b0 + bl) * 2 + (lc & 1)) * 32 + (ch & 31)) =
                    *reinterpret_cast<const uint4 *>(lds + (size_t)run * RL + off);
        }
    }
}

// ---- layout conversions ------------------------------------------------------------------------------------------------------------------
// slot-minor NTT-form tensor x [B][zd*npos cts][2][k][n] (canonical residues, or 28-bit limb pairs when `packed`) -> Xl.  One thread per (slot, image,
// position, poly, 32-channel block); lanes run over 64 consecutive slots, so every read is a coalesced 512-B row segment; every lane writes 7 x 32 B into its
// own slot block (1.4x write amplification at the memory side, profiles/r02_pmc_traffic.json).  This form serves the CONVOLUTION layout (several positions) only, which
// no bench configuration reaches any more (conv1 writes conv2's limb tensor itself); a dense layer's tensor goes through limb_pack_dense_kernel below.  (Round 2 tried an
// LDS transpose with 64 slots per workgroup here and dropped it -- 48.1 vs 46.4 ms; what it lacked was occupancy, see limb_pack_weights_kernel.)
// (Btot, b0: the B images are images b0 .. b0 + B of a tensor of Btot -- a group of chunks assembling one dense layer's input; Btot = B, b0 = 0 otherwise)
__global__ void __launch_bounds__(64) limb_pack_tensor_kernel(const u64 *x, i8 *xl, const ModParams *mods, int n, int k, int B, int zd, int zdp, int npos, int packed, int group,
                                                              int Btot, int b0, int flat_zdc, unsigned flat_img)
{
    const int sblocks = n / 64;
    const int sb = blockIdx.x % (sblocks * k), i = sb / sblocks, s = (sb % sblocks) * 64 + threadIdx.x;
    const int zblks = zdp / 32;
    const size_t items = (size_t)B * npos * 2 * zblks;           // item = ((b*npos + pos)*2 + c)*zblks + zb
    const u64 q = mods[i].q;
    // `group` consecutive items per thread: with one channel block they are adjacent 32-B pieces of the same slot block, written back to back so that
    // the L2 can merge them into whole lines before they go to HBM
    for (size_t r0 = (size_t)(blockIdx.x / (sblocks * k)) * group, u = 0; u < (size_t)group && r0 + u < items; u++) {
        size_t r = r0 + u;
        int zb, c, pos, b;
        if (npos == 1) { const size_t rows = (size_t)B * 2; zb = (int)(r / rows); r %= rows; c = (int)(r & 1); b = (int)(r >> 1); pos = 0; }      // K-blocked: a thread's items are neighbouring rows of one channel block
        else { zb = (int)(r % zblks); r /= zblks; c = (int)(r % 2); r /= 2; pos = (int)(r % npos); b = (int)(r / npos); }
        u32 pl[NPL][8];
#pragma unroll
        for (int l = 0; l < NPL; l++)
#pragma unroll
            for (int wv = 0; wv < 8; wv++) pl[l][wv] = 0;
#pragma unroll
        for (int z = 0; z < 32; z++) {
            const int zz = zb * 32 + z;
            if (zz < zd) {
                u64 v = x[((((size_t)b * zd * npos + (size_t)zz * npos + pos) * 2 + c) * k + i) * (size_t)n + s];
                if (packed) v = (v & 0xffffffffULL) | ((v >> 32) << 28);
                int d[NPL]; limb_digits(v, q, d);
#pragma unroll
                for (int l = 0; l < NPL; l++) pl[l][z >> 2] |= (u32)(d[l] & 0xff) << (8 * (z & 3));
            }
        }
        if (flat_zdc) {                 // flat form (fewer than 32 channels): [plane][poly][position][zdc], zdc bytes per piece, 4-byte aligned
            i8 *dstf = xl + (((size_t)i * n + s) * Btot + b0 + b) * (size_t)flat_img + ((size_t)c * npos + pos) * flat_zdc;
#pragma unroll
            for (int l = 0; l < NPL; l++) {
                u32 *o = reinterpret_cast<u32 *>(dstf + (size_t)l * 2 * npos * flat_zdc);
#pragma unroll
                for (int wv = 0; wv < 8; wv++) if (wv * 4 < flat_zdc) o[wv] = pl[l][wv];
            }
            continue;
        }
        // one position (a dense layer's input): K-blocked [plane][channel block][row = image * 2 + poly][32]
        i8 *dst = npos == 1 ? xl + ((size_t)i * n + s) * ((size_t)NPL * 2 * Btot * zdp) + ((size_t)zb * (2 * Btot) + (b0 + b) * 2 + c) * 32
                            : xl + (((size_t)i * n + s) * Btot + b0 + b) * ((size_t)NPL * npos * 2 * zdp) + ((size_t)pos * 2 + c) * zdp + zb * 32;
        const size_t plane_stride = npos == 1 ? (size_t)zblks * (2 * Btot) * 32 : (size_t)npos * 2 * zdp;
#pragma unroll
        for (int l = 0; l < NPL; l++) {
            uint4 *o = reinterpret_cast<uint4 *>(dst + (size_t)l * plane_stride);
            o[0] = make_uint4(pl[l][0], pl[l][1], pl[l][2], pl[l][3]); o[1] = make_uint4(pl[l][4], pl[l][5], pl[l][6], pl[l][7]);
        }
    }
}
// The same conversion for ONE position (a dense layer's K-blocked tensor [plane][channel block][row = image * 2 + poly][32]) as an LDS-staged transpose, the structure of
// limb_pack_weights_kernel below: a workgroup = 32 slots x 4 neighbouring rows of one channel block; thread (slot, row, 16-channel half) reads its channel values (lanes
// over slots: coalesced row segments), stages the seven planes' bytes, and the workgroup writes every (slot, plane)'s 4 x 32 = 128 contiguous bytes as one line.
#define TRG 4
#define TSL 32
__global__ void __launch_bounds__(256) limb_pack_dense_kernel(const u64 *x, i8 *xl, const ModParams *mods, int n, int k, int B, int zd, int zdp, int packed, int Btot, int b0)
{
    __shared__ __attribute__((aligned(16))) i8 st[TSL * NPL * TRG * 32];         // [slot][plane][row of the group][32 channels]
    const int sblocks = n / TSL, zblks = zdp / 32, rgs = (2 * B + TRG - 1) / TRG;
    const int sb = blockIdx.x % (sblocks * k), i = sb / sblocks, s0 = (sb % sblocks) * TSL;
    size_t r = blockIdx.x / (sblocks * k);                       // zb * rgs + row group
    const int rg = (int)(r % rgs), zb = (int)(r / rgs);
    const u64 q = mods[i].q;
    const int lane = threadIdx.x & (TSL - 1), qrow = (threadIdx.x >> 5) & (TRG - 1), h = threadIdx.x >> 7, row = rg * TRG + qrow;      // row = image * 2 + poly inside this call
    {
        u32 pl[NPL][4];
#pragma unroll
        for (int l = 0; l < NPL; l++)
#pragma unroll
            for (int wv = 0; wv < 4; wv++) pl[l][wv] = 0;
        if (row < 2 * B) {
            const int b = row >> 1, c = row & 1, z0 = zb * 32 + h * 16;
            const u64 *src = x + ((((size_t)b * zd + z0) * 2 + c) * k + i) * (size_t)n + s0 + lane;
#pragma unroll
            for (int z = 0; z < 16; z++)
                if (z0 + z < zd) {
                    u64 v = src[(size_t)z * 2 * k * n];
                    if (packed) v = (v & 0xffffffffULL) | ((v >> 32) << 28);
                    int d[NPL]; limb_digits(v, q, d);
#pragma unroll
                    for (int l = 0; l < NPL; l++) pl[l][z >> 2] |= (u32)(d[l] & 0xff) << (8 * (z & 3));
                }
        }
        i8 *sp = st + (size_t)lane * (NPL * TRG * 32) + qrow * 32 + h * 16;
#pragma unroll
        for (int l = 0; l < NPL; l++) *reinterpret_cast<uint4 *>(sp + l * (TRG * 32)) = make_uint4(pl[l][0], pl[l][1], pl[l][2], pl[l][3]);
    }
    __syncthreads();
    const int pieces_per_run = TRG * 2, rows_here = min(TRG, 2 * B - rg * TRG);
    const size_t plane_stride = (size_t)zblks * (2 * Btot) * 32, slot_stride = (size_t)NPL * plane_stride;
    for (int o = threadIdx.x; o < TSL * NPL * pieces_per_run; o += 256) {
        const int run = o / pieces_per_run, part = o - run * pieces_per_run, sl = run / NPL, l = run - sl * NPL;
        if ((part >> 1) >= rows_here) continue;
        i8 *dst = xl + ((size_t)i * n + s0 + sl) * slot_stride + (size_t)l * plane_stride + ((size_t)zb * (2 * Btot) + (size_t)b0 * 2 + rg * TRG) * 32 + part * 16;
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(st + (size_t)run * (TRG * 32) + part * 16);
    }
}
// NTT-form weights w [F][zd][taps][k][n] (canonical) -> Wl (pre-zeroed: channel / filter padding).  A transpose: the source rows are slot-minor, Wl is slot-major with
// the 32-byte piece of a (filter, tap, channel block) as its unit.  A workgroup = 64 consecutive slots x 4 neighbouring filters of one (tap, channel block): thread
// (slot lane, filter) reads its 32 channel values -- lanes run over the slots, every load a coalesced 512-byte row segment -- and stages the seven planes' 32 bytes in
// LDS; then the workgroup writes the staged block so that the 4 x 32 = 128 contiguous bytes a (slot, plane) owns leave as ONE whole line from eight adjacent lanes.
// (Round 2's form -- one thread per piece, every lane storing 32 bytes into its own slot block 3.5 MB from its neighbour's -- ran at 1.0 TB/s; a streamed layer packs
// its weights inside every forward: PlainModelWoPad's fc3 with all eight primes spent 23 of its 40 ms per image here.)
// w may be a TILE of the layer's filters: filter f of w is filter f0 + f of Wl (whose filter stride Fp belongs to the whole layer)
#define WFG 4
#define WSL 32
// (32 slots per workgroup: 28 KiB of staging, five workgroups per CU -- the loads of one overlap the digit arithmetic and the stores of the others; with 64 slots and two
// workgroups per CU loads, arithmetic and stores ran one after the other: 8 + 8 + 11 ms of a 27-ms call.)  Fz >= F: filters F .. Fz-1 are written as zeros (the filter
// padding of the layer's last tile), so nothing has to be cleared beforehand
__global__ void __launch_bounds__(256) limb_pack_weights_kernel(const u64 *w, i8 *wl, const ModParams *mods, int n, int k, int F, int Fz, int Fp, int zd, int zblks, int xf, int yf, int f0,
                                                                int flat_zdc, int S)
{
    __shared__ __attribute__((aligned(16))) i8 st[WSL * NPL * WFG * 32];         // [slot][plane][filter of the group][32 channels]
    const int sblocks = n / WSL, taps = xf * yf, steps = flat_zdc ? xf * S : taps * zblks;
    const int sb = blockIdx.x % (sblocks * k), i = sb / sblocks, s0 = (sb % sblocks) * WSL;
    size_t r = blockIdx.x / (sblocks * k);                       // fg * steps + step, fg = group of WFG filters
    const int step = (int)(r % steps); const int fg = (int)(r / steps);
    const ModParams m = mods[i];
    const u64 R = barrett128(0, 1, m);                           // 2^64 mod q: the factor the kernel's Montgomery reduction divides out
    const int lane = threadIdx.x & (WSL - 1), fq = (threadIdx.x >> 5) & (WFG - 1), h = threadIdx.x >> 7, f = fg * WFG + fq;      // thread = (slot, filter, 16-term half of the step)
    {
        u32 pl[NPL][4];
#pragma unroll
        for (int l = 0; l < NPL; l++)
#pragma unroll
            for (int wv = 0; wv < 4; wv++) pl[l][wv] = 0;
        if (f < F) {
            // term e = 16 h + z of the step  ->  (tap, channel).  Blocked form: step = (tap, 32-channel block).  Flat form: step = (window row kx, 32-byte piece of the
            // row's run of (ky, channel) terms, zdc bytes per ky); bytes past yf zdc and the channel padding are zero weights
            int tap, ch, ky = 0, kx = 0;
            if (flat_zdc) { kx = step / S; const int j0 = (step - kx * S) * 32 + h * 16; ky = j0 / flat_zdc; ch = j0 - ky * flat_zdc; tap = kx * yf + ky; }
            else { tap = step / zblks; ch = (step - tap * zblks) * 32 + h * 16; }
#pragma unroll
            for (int z = 0; z < 16; z++) {
                if (ch < zd && (!flat_zdc || ky < yf)) {
                    const u64 v = mulmod(w[((((size_t)f * zd + ch) * taps + tap) * k + i) * (size_t)n + s0 + lane], R, m);
                    int d[NPL]; limb_digits(v, m.q, d);
#pragma unroll
                    for (int l = 0; l < NPL; l++) pl[l][z >> 2] |= (u32)(d[l] & 0xff) << (8 * (z & 3));
                }
                ch++;
                if (flat_zdc && ch == flat_zdc) { ch = 0; ky++; tap++; }
            }
        }
        i8 *sp = st + (size_t)lane * (NPL * WFG * 32) + fq * 32 + h * 16;
#pragma unroll
        for (int l = 0; l < NPL; l++) *reinterpret_cast<uint4 *>(sp + l * (WFG * 32)) = make_uint4(pl[l][0], pl[l][1], pl[l][2], pl[l][3]);
    }
    __syncthreads();
    // WSL slots x 7 planes runs of WFG * 32 bytes, 16 bytes per lane: eight adjacent lanes write one run (filters past Fz of a ragged last group are not stored)
    const int pieces_per_run = WFG * 2, filters_here = min(WFG, Fz - fg * WFG);
    const size_t slot_stride = (size_t)((steps + 1) & ~1) * NPL * Fp * 32;
    for (int o = threadIdx.x; o < WSL * NPL * pieces_per_run; o += 256) {
        const int run = o / pieces_per_run, part = o - run * pieces_per_run, sl = run / NPL, l = run - sl * NPL;
        if ((part >> 1) >= filters_here) continue;
        i8 *dst = wl + ((size_t)i * n + s0 + sl) * slot_stride + (size_t)step * (NPL * Fp * 32) + (size_t)l * Fp * 32 + (size_t)(f0 + fg * WFG) * 32 + part * 16;
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(st + (size_t)run * (WFG * 32) + part * 16);
    }
}
// Ys [slot][rows] (rows = B*F*P*2 canonical u64, slot-major) -> y [rows][k][n] (the engine's slot-minor tensor layout): 64 x 64 tile transpose through LDS
__global__ void __launch_bounds__(256) slotmajor_to_rows_kernel(const u64 *ys, u64 *y, int n, int k, size_t rows, int pack_out)
{
    __shared__ u64 tile[64][65];
    const int sblocks = n / 64;
    const int sb = blockIdx.x % (sblocks * k), i = sb / sblocks, s0 = (sb % sblocks) * 64;
    const size_t e0 = (size_t)(blockIdx.x / (sblocks * k)) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int sl = r * 4 + ty;
        const size_t e = e0 + tx;
        tile[sl][tx] = e < rows ? ys[((size_t)i * n + s0 + sl) * rows + e] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int el = r * 4 + ty;
        const size_t e = e0 + el;
        if (e < rows) { u64 v = tile[tx][el]; if (pack_out) v = (v & 0x0fffffffULL) | ((v >> 28) << 32); y[(e * k + i) * (size_t)n + s0 + tx] = v; }
    }
}
// Ys [slot][B][zd'*2] (zd' = F*P flattened channels, poly innermost) -> Xl' [slot][7][zdp'/32][B*2][32] for a dense consumer.  One thread per 16 channels.
__global__ void __launch_bounds__(256) slotmajor_to_limb_kernel(const u64 *ys, i8 *xl, const ModParams *mods, int n, int B, int zd, int zdp, size_t total)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;              // ((slot*B + b)*2 + c)*(zdp/16) + zg
    if (t >= total) return;
    const int zgs = zdp / 16;
    const int zg = (int)(t % zgs); size_t r = t / zgs; const int c = (int)(r % 2); r /= 2; const int b = (int)(r % B); const size_t slot = r / B;
    const u64 q = mods[slot / n].q;
    u32 pl[NPL][4];
#pragma unroll
    for (int l = 0; l < NPL; l++)
#pragma unroll
        for (int wv = 0; wv < 4; wv++) pl[l][wv] = 0;
    const u64 *src = ys + (slot * B + b) * (size_t)zd * 2 + c;
#pragma unroll
    for (int z = 0; z < 16; z++) {
        const int zz = zg * 16 + z;
        if (zz < zd) {
            int d[NPL]; limb_digits(src[(size_t)zz * 2], q, d);
#pragma unroll
            for (int l = 0; l < NPL; l++) pl[l][z >> 2] |= (u32)(d[l] & 0xff) << (8 * (z & 3));
        }
    }
    i8 *dst = xl + slot * ((size_t)NPL * 2 * B * zdp) + ((size_t)(zg >> 1) * (2 * B) + b * 2 + c) * 32 + (zg & 1) * 16;      // K-blocked dense input (header)
#pragma unroll
    for (int l = 0; l < NPL; l++) *reinterpret_cast<uint4 *>(dst + (size_t)l * (zdp / 32) * (2 * B) * 32) = make_uint4(pl[l][0], pl[l][1], pl[l][2], pl[l][3]);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
int k_limb_flat_zdc(int zd);
static void limb_tables(const crc_ctx *c, int T, int (*acc0)[13], u64 *qinv)      // limbred.h: accumulator biases and q^-1 mod 2^64 per modulus
{
    for (int i = 0; i < c->k; i++) { limb_bias_table(c->tabs[i].m.q, T, acc0[i]); qinv[i] = inverse_mod_2_64(c->tabs[i].m.q); }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------------------
bool k_limb_supported(const crc_ctx *c, int T)
{
    if (c->n < 64 || T > 18000) return false;
    for (int i = 0; i < c->k; i++) if (c->tabs[i].m.bits > 55) return false;          // 7 balanced bytes hold |r'| < 2^54
    return true;
}
// the flat form (header): layers of fewer than 32 channels.  zdc = channel bytes per position; a dense layer (one position) keeps its K-blocked tensor, its one
// reduction step is the same bytes either way
int k_limb_flat_zdc(int zd) { return zd < 32 ? round_up(zd, 4) : 0; }
int k_limb_steps(int zd, int xf, int yf) { const int zdc = k_limb_flat_zdc(zd); return zdc ? xf * ((yf * zdc + 31) / 32) : xf * yf * (round_up(zd, 32) / 32); }
// bytes of one image of a convolution's limb tensor (npos > 1)
static size_t limb_img_bytes(int zd, int npos) { const int zdc = k_limb_flat_zdc(zd); return zdc ? (size_t)round_up(NPL * 2 * npos * zdc, 16) : (size_t)NPL * npos * 2 * round_up(zd, 32); }
size_t k_limb_tensor_bytes(const crc_ctx *c, int B, int zd, int npos)
{
    if (npos == 1) return (size_t)c->n * c->k * B * NPL * 2 * round_up(zd, 32);
    return (size_t)c->n * c->k * B * limb_img_bytes(zd, npos) + 64;                     // (+ 64: the last 32-byte piece of a flat run may read past the last position)
}
size_t k_limb_weights_bytes(const crc_ctx *c, int nf, int zd, int xf, int yf) { return (size_t)c->n * c->k * round_up(k_limb_steps(zd, xf, yf), 2) * NPL * round_up(nf, 64) * 32; }      // (an odd number of reduction steps gets a zero step)
size_t k_limb_result_words(const crc_ctx *c, int B, int nf, int P) { return (size_t)c->n * c->k * B * nf * P * 2; }

int k_limb_pack_tensor(crc_ctx *c, const u64 *x, i8 *xl, int B, int zd, int npos, bool packed, hipStream_t st, int Btot, int b0)
{
    if (Btot <= 0) { Btot = B; b0 = 0; }
    if (b0 < 0 || b0 + B > Btot) return CRC_ERR_INVALID_ARGUMENT;
    const int zdp = round_up(zd, 32);
    // pieces per thread: a dense layer's K-blocked tensor (one position) takes 4 neighbouring rows of a channel block -- 4 x 32 B = one whole line per plane, written back to
    // back; for the convolution layout 4 and 8 adjacent pieces per thread measured 3-7 % slower
    const int group = c->tune.limb_pack_group > 1 ? c->tune.limb_pack_group : (npos == 1 && (B * 2) % 4 == 0 ? 4 : 1);
    const size_t items = (size_t)B * npos * 2 * (zdp / 32);
    if (npos == 1 && c->tune.limb_pack_group <= 1) {       // a dense layer's tensor: the LDS-staged transpose (whole-line stores)
        const size_t blocks = (size_t)(c->n / TSL) * c->k * (zdp / 32) * ((2 * (size_t)B + TRG - 1) / TRG);
        if (blocks == 0) return CRC_OK;
        if (blocks > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(limb_pack_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, xl, c->d_mods, c->n, c->k, B, zd, zdp, packed ? 1 : 0, Btot, b0);
        HIPCHK(hipGetLastError());
        return CRC_OK;
    }
    const size_t blocks = (size_t)(c->n / 64) * c->k * ((items + group - 1) / group);
    if (blocks == 0) return CRC_OK;
    if (blocks > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    const int fz = npos > 1 ? k_limb_flat_zdc(zd) : 0;
    hipLaunchKernelGGL(limb_pack_tensor_kernel, dim3((unsigned)blocks), dim3(64), 0, st, x, xl, c->d_mods, c->n, c->k, B, zd, zdp, npos, packed ? 1 : 0, group, Btot, b0, fz,
                       (unsigned)limb_img_bytes(zd, npos));
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
// w: filters f0 .. f0 + ft of the layer's nf (ft = nf, f0 = 0: the whole layer); the padding of Wl is zeroed when the first tile (f0 = 0) is packed
int k_limb_pack_weights(crc_ctx *c, const u64 *w, i8 *wl, int nf, int zd, int xf, int yf, hipStream_t st, int f0, int ft)
{
    if (ft < 0) ft = nf;
    if (f0 < 0 || ft < 1 || f0 + ft > nf) return CRC_ERR_INVALID_ARGUMENT;
    const int zblks = round_up(zd, 32) / 32, Fp = round_up(nf, 64), zdc = k_limb_flat_zdc(zd), S = zdc ? (yf * zdc + 31) / 32 : 0;
    // the kernel writes every term of every step of every filter it is given, zeros for the padding; the call that holds the layer's last filter also writes the filter
    // padding up to Fp as zeros.  What no call writes is the zero step that evens out an odd number of reduction steps: cleared with the first tile
    const size_t step = (size_t)NPL * Fp * 32, steps = (size_t)k_limb_steps(zd, xf, yf), slot = (size_t)round_up((int)steps, 2) * step;
    if (f0 == 0 && (steps & 1)) HIPCHK(hipMemset2DAsync(wl + steps * step, slot, 0, step, (size_t)c->n * c->k, st));
    const int fz = f0 + ft == nf ? Fp - f0 : ft;
    const size_t blocks = (size_t)(c->n / WSL) * c->k * ((fz + WFG - 1) / WFG) * steps;
    if (blocks > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(limb_pack_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, wl, c->d_mods, c->n, c->k, ft, fz, Fp, zd, zblks, xf, yf, f0, zdc, S);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
int k_limb_result_to_rows(crc_ctx *c, const u64 *ys, u64 *y, size_t rows, bool pack_out, hipStream_t st)
{
    const size_t blocks = (size_t)(c->n / 64) * c->k * ((rows + 63) / 64);
    if (blocks == 0) return CRC_OK;
    if (blocks > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(slotmajor_to_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ys, y, c->n, c->k, rows, pack_out ? 1 : 0);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
int k_limb_result_to_limb(crc_ctx *c, const u64 *ys, i8 *xl, int B, int zd, hipStream_t st)
{
    const int zdp = round_up(zd, 32);
    const size_t total = (size_t)c->n * c->k * B * 2 * (zdp / 16);
    if (total == 0) return CRC_OK;
    hipLaunchKernelGGL(slotmajor_to_limb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ys, xl, c->d_mods, c->n, B, zd, zdp, total);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
// the layer: Xl (B images of zd x xd x yd) * Wl -> Ys [slot][B][nf][P][2], + NTT-form bias on poly 0
// can the layer write a dense consumer's limb tensor itself?  (the output tile must be whole images: 2P a power of two dividing 64)
bool k_limb_direct_dense(int P) { const int p2 = 2 * P; return p2 <= 64 && (p2 & (p2 - 1)) == 0; }
int k_limb_mac(crc_ctx *c, const i8 *xl, const i8 *wl, u64 *ys, i8 *xl_out, const u64 *bias_ntt, int B, int zd, int xd, int yd, int xs, int ys_, int xf, int yf, int nf, hipStream_t st)
{
    if (B == 0) return CRC_OK;
    const int xo = (xd - xf) / xs + 1, yo = (yd - yf) / ys_ + 1;
    MfmaArgs a{};
    a.xl = xl; a.wl = wl; a.ys = ys; a.mods = c->d_mods; a.bias = bias_ntt;
    a.xl_out = nullptr; a.lp2 = -1; a.zdp_out = 0; a.mfast = 0;
    a.n = c->n; a.k = c->k; a.B = B; a.zdp = round_up(zd, 32); a.npos = xd * yd; a.yd = yd; a.xs = xs; a.ys_ = ys_; a.yf = yf; a.yo = yo; a.P = xo * yo;
    a.F = nf; a.Fp = round_up(nf, 64); a.zblks = a.zdp / 32; a.ksteps_real = k_limb_steps(zd, xf, yf); a.M = B * a.P * 2;
    a.mtiles = (a.M + 63) / 64; a.ntiles = a.Fp / 64;
    a.zdc = a.npos > 1 ? k_limb_flat_zdc(zd) : 0; a.flat = a.zdc ? 1 : 0; a.S = a.zdc ? (yf * a.zdc + 31) / 32 : 1; a.fplane = (unsigned)(2 * a.npos * a.zdc);
    const size_t img = a.npos == 1 ? (size_t)NPL * 2 * a.zdp : limb_img_bytes(zd, a.npos);
    if (img * B > 0xffffffffULL || !k_limb_supported(c, a.ksteps_real * 32)) return CRC_ERR_UNSUPPORTED;
    if (xl_out) {
        if (!k_limb_direct_dense(a.P)) return CRC_ERR_INVALID_ARGUMENT;
        a.xl_out = xl_out; a.zdp_out = (unsigned)round_up(nf * a.P, 32);
        for (a.lp2 = 0; (1 << a.lp2) < 2 * a.P; a.lp2++) {}
    }
    a.img_bytes = (unsigned)img; a.wslot_bytes = (unsigned long long)round_up(a.ksteps_real, 2) * NPL * a.Fp * 32;
    const size_t grid = (size_t)c->n * c->k * a.mtiles * a.ntiles;
    if (grid > 0x7fffffffULL) return CRC_ERR_INVALID_ARGUMENT;
    if (c->k > 8) return CRC_ERR_UNSUPPORTED;
    limb_tables(c, a.ksteps_real * 32, a.acc0, a.qinv);
    const int ring = c->tune.mfma_ring;     // tuning (tools/)
    a.ksteps = round_up(a.ksteps_real, 2);
    // layers with few row tiles and many filter tiles (dense layers) stream their weights: walk the row tiles of one filter tile back to back, so that the weight tile is
    // fetched from HBM once and the (small) tensor stays in L2; convolutions keep filter tiles fastest (the big tensor tile is shared, the weights sit in L2)
    a.mfast = c->tune.mfma_order >= 0 ? c->tune.mfma_order : (a.mtiles < a.Fp / 32 ? 1 : 0);
    const int variant = c->tune.mfma_variant;                    // 2 (default): two workgroups per CU (mfma_mac2w_kernel); 1: mfma_mac_kernel (the tests run both: crc_ctx_set_tuning)
    if (variant == 2) {
        const int slots = c->n * c->k, ntiles2 = a.Fp / 32;
        a.xcdmap = (slots & 7) == 0 ? 1 : 0;
        const unsigned gfast = (unsigned)(a.mfast ? a.mtiles : ntiles2) * (a.xcdmap ? 8u : 1u), gslow = (unsigned)(a.mfast ? ntiles2 : a.mtiles), gz = (unsigned)(a.xcdmap ? slots >> 3 : slots);
        if (gslow > 65535u || gz > 65535u) return CRC_ERR_INVALID_ARGUMENT;
        const size_t lds = (size_t)5 * TILE_A;
        auto kern2 = a.npos == 1 ? mfma_mac2w_kernel<5, 0> : a.flat ? mfma_mac2w_kernel<5, 1> : mfma_mac2w_kernel<5, 2>;
        { const int rc = crc_ctx_ensure_lds(c, (const void *)kern2, lds); if (rc) return rc; }
        hipLaunchKernelGGL(kern2, dim3(gfast, gslow, gz), dim3(256), lds, st, a);
        HIPCHK(hipGetLastError());
        return CRC_OK;
    }
    const int nst = ring == 4 ? 4 : 5;
    const size_t lds = (size_t)nst * 2 * TILE_B;
    auto kern = nst == 4 ? mfma_mac_kernel<4> : mfma_mac_kernel<5>;
    { const int rc = crc_ctx_ensure_lds(c, (const void *)kern, lds); if (rc) return rc; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, st, a);
    HIPCHK(hipGetLastError());
    return CRC_OK;
}
