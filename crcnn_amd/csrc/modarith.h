// modarith.h -- 64-bit modular arithmetic shared by host table builders and gfx950 kernels.
//
// Everything in the engine is exact arithmetic in Z_q for word-sized q (54..61 bit), so any exact algorithm produces
// the reference's bits.  On the device there is no 64x64 multiplier: products are built from
// 32-bit v_mad_u64_u32 / v_mul_hi_u32 pieces by the compiler (__umul64hi, unsigned __int128 are not used on device).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CRC_HD __host__ __device__ __forceinline__
#else
#define CRC_HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;

// per-modulus constants (host builds them, kernels read them from a small device array)
struct ModParams {
    u64 q;          // modulus
    u64 r0, r1;     // floor(2^128 / q) low / high word  (Barrett, same constant SEAL calls const_ratio)
    u64 two_q;      // 2q
    u32 bits;       // significant bits of q
    u32 fold;       // fold_constant(q, bits): 2^bits - q when the folding reduction is valid for this modulus, else 0
};

// The folding reduction fold128 below is exact for ANY 128-bit input only when q = 2^b - d with 52 <= b <= 62 and d < 2^26: after three
// folds the value is below 2^(206-3b) + 2^b, which must stay under 2q (b >= 52), and the intermediate high parts must fit a word
// (b >= 48).  Every coefficient modulus SEAL's default sets use (54-55 bits) and the 61-bit auxiliary primes qualify; smaller primes of
// the same shape (SEAL's small_mods_40bit / 50bit) take the generic Barrett path.
CRC_HD u32 fold_constant(u64 q, u32 bits)
{
    if (bits < 52 || bits > 62) return 0;
    const u64 d = ((u64)1 << bits) - q;
    return d && d < ((u64)1 << 26) ? (u32)d : 0;
}

// 64 x 64 -> 128 from four 32 x 32 + 64 multiply-adds (v_mad_u64_u32 on gfx950: there is no wider multiplier, and the compiler's
// expansion of __umul64hi next to a separate a*b costs about twice as many multiplies).  No step can overflow: (2^32-1)^2 + 2 (2^32-1) < 2^64.
CRC_HD void mul64wide_mad(u64 a, u64 b, u64 &lo, u64 &hi)
{
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    const u64 p10 = (u64)a1 * b0 + (u32)p01;
    hi = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    lo = (u64)(u32)p00 | (p10 << 32);
}

CRC_HD u64 mulhi64(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__) || defined(CRC_FORCE_MAD_MUL)
    u64 lo, hi; mul64wide_mad(a, b, lo, hi); return hi;
#else
    return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

CRC_HD void mul64wide(u64 a, u64 b, u64 &lo, u64 &hi)
{
#if defined(__HIP_DEVICE_COMPILE__) || defined(CRC_FORCE_MAD_MUL)
    mul64wide_mad(a, b, lo, hi);
#else
    lo = a * b;
    hi = mulhi64(a, b);
#endif
}

CRC_HD u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
CRC_HD u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
CRC_HD u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

// x = hi*2^64 + lo  ->  x mod q, canonical, for q = 2^b - d with a small d (m.fold): 2^b = d (mod q), so the part of x above bit b
// folds down as (x >> b) d + (x mod 2^b).  Three folds take any 128-bit x below 2^b + 2^(206-3b) < 2q; 6 word multiplies instead of the
// 18 of the generic Barrett reduction below.  Requires 52 <= b <= 62, d < 2^26 (fold_constant above; m.fold is 0 otherwise).
CRC_HD u64 fold128(u64 lo, u64 hi, const ModParams &m)
{
    const u32 b = m.bits; const u64 d = m.fold, mask = ((u64)1 << b) - 1;
    // fold 1: h1 = x >> b (up to 128 - b <= 76 bits), x1 = h1 d + (x mod 2^b) < 2^102 + 2^b
    const u64 h1l = (lo >> b) | (hi << (64 - b)), h1h = hi >> b;
    u64 pl, ph; mul64wide(h1l, d, pl, ph);
    u64 x1l = pl + (lo & mask); u64 x1h = ph + h1h * d + (x1l < pl);
    // fold 2: h2 = x1 >> b (<= 154 - 2b <= 50 bits)
    const u64 h2 = (x1l >> b) | (x1h << (64 - b));
    mul64wide(h2, d, pl, ph);
    u64 x2l = pl + (x1l & mask); u64 x2h = ph + (x2l < pl);
    // fold 3: h3 = x2 >> b (<= 180 - 3b <= 24 bits), x3 = h3 d + (x2 mod 2^b) < 2^50 + 2^b
    const u64 h3 = (x2l >> b) | (x2h << (64 - b));
    u64 r = h3 * d + (x2l & mask);
    return r >= m.q ? r - m.q : r;
}

// a b mod q for CANONICAL a, b and q = 2^b - d with b >= 53 (m.fold != 0, m.bits >= 53), lazily: two folds leave a value below 2^b + 2^52 < 2q -- the third
// fold and the conditional subtraction of fold128 serve inputs this product cannot have.  (x < 2^2b: h1 = x >> b < 2^b, x1 = h1 d + (x mod 2^b) < 2^(b+26); h2
// = x1 >> b < 2^27, x2 = h2 d + (x1 mod 2^b) < 2^53 + 2^b.)  For the transforms whose first butterfly takes lazy operands (ntt_rows_wave_kernel, prologue 4).
CRC_HD u64 mulmod_fold2_lazy(u64 a, u64 b, const ModParams &m)
{
    u64 lo, hi; mul64wide(a, b, lo, hi);
    const u32 bt = m.bits; const u64 d = m.fold, mask = ((u64)1 << bt) - 1;
    const u64 h1 = (lo >> bt) | (hi << (64 - bt));
    u64 pl, ph; mul64wide(h1, d, pl, ph);
    const u64 x1l = pl + (lo & mask), x1h = ph + (x1l < pl);
    const u32 h2 = (u32)((x1l >> bt) | (x1h << (64 - bt)));
    return (u64)h2 * m.fold + (x1l & mask);
}

// The MAC kernels' epilogue for q = 2^b - d (m.fold, 50 <= b <= 55):  a0 + (a1 - a0 - a2) 2^28 + a2 2^56  mod q  from the three lazy limb
// accumulators a_j = A_j + o_j 2^63 (o_j < 1024 packed in ov as 3 x 10 bits), canonical.  Every accumulator is folded below 2^b + 2^50
// first (one word multiply each), so the 2^28 / 2^56 weights never need 128-bit arithmetic: 9 word multiplies in all, about a third of
// the instructions of recombining into 128 bits and reducing that.
CRC_HD u64 mac_reduce_fold(u64 A0, u64 A1, u64 A2, u32 ov, const ModParams &m)
{
    const u32 b = m.bits, d = m.fold;
    const u64 mask = ((u64)1 << b) - 1;
    const u32 os = 63 - b;
    // a_j = (A_j >> b) 2^b + (A_j mod 2^b) + o_j 2^(63-b) 2^b  ==  h_j d + l_j,   h_j < 2^24
    const u32 h0 = (u32)(A0 >> b) + ((ov & 1023u) << os), h1 = (u32)(A1 >> b) + (((ov >> 10) & 1023u) << os), h2 = (u32)(A2 >> b) + (((ov >> 20) & 1023u) << os);
    const u64 a0 = (u64)h0 * d + (A0 & mask), a1 = (u64)h1 * d + (A1 & mask);
    u64 a2 = (u64)h2 * d + (A2 & mask);                                   // each < 2^b + 2^50
    // mid = a1 - a0 - a2 + 8q  in (0, 2^(b+3.2)), folded once: < 2^b + 2^30
    u64 mid = a1 + (m.q << 3) - a0 - a2;
    mid = (u64)(u32)(mid >> b) * d + (mid & mask);
    // mid 2^28  ==  (mid >> (b-28)) d + (mid mod 2^(b-28)) 2^28
    const u32 lb = b - 28;
    const u64 M = (u64)(u32)(mid >> lb) * d + ((mid & (((u64)1 << lb) - 1)) << 28);          // < 2^54.1 + 2^b
    // a2 2^56  ==  a2 c,  c = d 2^(56-b) < 2^32,  a2 first brought below 2^b
    a2 = (a2 >> b) ? (a2 & mask) + d : a2;
    const u32 c = d << (56 - b);
    const u64 p0 = (u64)(u32)a2 * c, p1 = (u64)(u32)(a2 >> 32) * c;
    const u64 t = p1 + (p0 >> 32);                                        // a2 c = t 2^32 + lo32(p0) < 2^(b+32)
    const u32 hb = b - 32;
    const u64 N = (u64)(u32)(t >> hb) * d + (((t & (((u64)1 << hb) - 1)) << 32) | (u32)p0);  // < 2^58 + 2^b
    const u64 sum = a0 + M + N;                                           // < 2^58.3
    const u64 r = (u64)(u32)(sum >> b) * d + (sum & mask);                // < 2^b + 2^35 < 2q
    return r >= m.q ? r - m.q : r;
}

// x = hi*2^64 + lo  ->  x mod q, canonical.  Exact for any 128-bit x (quotient estimate is off by at most one).
CRC_HD u64 barrett128(u64 lo, u64 hi, const ModParams &m)
{
    if (m.fold) return fold128(lo, hi, m);
    // floor(x * r / 2^128) mod 2^64, r = r1*2^64 + r0
    u64 c = mulhi64(lo, m.r0);
    u64 t_lo, t_hi;
    mul64wide(lo, m.r1, t_lo, t_hi);
    u64 s1 = t_lo + c;
    u64 acc = t_hi + (s1 < c);
    mul64wide(hi, m.r0, t_lo, t_hi);
    u64 s2 = s1 + t_lo;
    u64 carry = t_hi + (s2 < s1);
    u64 quot = hi * m.r1 + acc + carry;
    u64 r = lo - quot * m.q;
    return r >= m.q ? r - m.q : r;
}

// a*w mod q, canonical, for a constant w with its exact Shoup companion wp = floor(w 2^64 / q); any 64-bit a, q < 2^63
CRC_HD u64 mulmod_shoup(u64 a, u64 w, u64 wp, u64 q)
{
    const u64 r = a * w - mulhi64(a, wp) * q;
    return r >= q ? r - q : r;
}

CRC_HD u64 mulmod(u64 a, u64 b, const ModParams &m)
{
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    return barrett128(lo, hi, m);
}

// Shoup multiplication by a constant w with precomputed wp = floor(w * 2^64 / q):  returns a*w mod q in [0, 2q)
// for any 64-bit a (lazy), provided q < 2^63.
CRC_HD u64 mulmod_shoup_lazy(u64 a, u64 w, u64 wp, u64 q)
{
    u64 h = mulhi64(wp, a);
    return a * w - h * q;
}
