// limbred.h -- the once-per-output reduction of the matrix-core kernels (kernels_mfma.hip, kernels_mfma1.hip) and the tables it needs; host + device, so that
// tests/cpp/limbred_check.cpp can run the very same code against 128-bit integer arithmetic on the CPU.
//
// A modular multiply-add chain  V = sum_t x_t w_t  over centred representatives in balanced base 256 arrives as 13 int32 diagonals D_d = sum_t sum_{l+m=d} a_l b_m,
// V = sum_d D_d 2^(8d).  The kernels start every accumulator at a bias B_d instead of 0:
//   * B_d > max |D_d|, so D'_d = B_d + D_d is a NON-NEGATIVE word and U = sum_d D'_d 2^(8d) can be assembled as unsigned multi-word numbers (no sign handling);
//   * K* = sum_d B_d 2^(8d) is a multiple of q (the low digits of B are adjusted by the balanced digits of the distance to the nearest multiple), so U == V (mod q)
//     without a correction term.
// The weights carry the factor 2^64 mod q; one Montgomery step in its subtractive form -- m = U_lo q^-1 mod 2^64 makes U - m q divisible by 2^64, the low halves cancel
// without a borrow, t = U_hi - hi64(m q) in (U 2^-64 - q, U 2^-64] -- divides it out again.
#pragma once
#include "modarith.h"

CRC_HD u32 crc_alignbit(u32 hi, u32 lo, u32 sh)      // low word of (hi:lo) >> sh, 0 <= sh < 32
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (u32)((((u64)hi << 32) | lo) >> sh);
#endif
}

// generic form: any 32-bit D'_d (reductions of up to 18 000 terms), 0 < U < 2^127.  The diagonals d = r (mod 4) are the words of one number each (no overlap); U is
// their sum at byte offsets 0..3.  After the Montgomery step t + q is in (0, 2^63 + q): folded (q = 2^b - f) or Barrett-reduced to the canonical residue.
CRC_HD u64 diag_reduce(const int (&D)[13], const ModParams &m, u64 qinv)
{
    typedef unsigned __int128 u128;
    const u128 s0 = ((u128)(((u64)(u32)D[12] << 32) | (u32)D[8]) << 64) | (((u64)(u32)D[4] << 32) | (u32)D[0]);
    const u128 s1 = ((u128)(u32)D[9] << 64) | (((u64)(u32)D[5] << 32) | (u32)D[1]);
    const u128 s2 = ((u128)(u32)D[10] << 64) | (((u64)(u32)D[6] << 32) | (u32)D[2]);
    const u128 s3 = ((u128)(u32)D[11] << 64) | (((u64)(u32)D[7] << 32) | (u32)D[3]);
    const u128 U = s0 + (s1 << 8) + (s2 << 16) + (s3 << 24);
    const u64 ulo = (u64)U, uhi = (u64)(U >> 64);
    const u64 mq = ulo * qinv;
    const u64 t = uhi - mulhi64(mq, m.q) + m.q;                  // in (0, 2^63 + q)
    if (m.fold) {                                                 // 2^b = f (mod q): (t >> b) < 2^12, f < 2^26
        const u64 r = (t & (((u64)1 << m.bits) - 1)) + (u64)(u32)(t >> m.bits) * (u32)m.fold;        // one 32 x 32 multiply
        return r >= m.q ? r - m.q : r;
    }
    return barrett128(t, 0, m);
}

// The same value from pre-shifted 32-bit words (round 4).  The compiler lowers the 128-bit shifts of s1 << 8, s2 << 16, s3 << 24 above to chains of 64-bit shifts and
// ors (a quarter of the epilogue's instructions); here every shifted word is one v_alignbit_b32, the four numbers add as two carry chains, and the fold after the
// Montgomery step works on the upper word (bits >= 32 for every modulus with a fold constant).  Same contract, same result as diag_reduce.
CRC_HD u64 diag_reduce_w(const int (&D)[13], const ModParams &m, u64 qinv)
{
    typedef unsigned __int128 u128;
    const u32 a0 = (u32)D[1] << 8,  a1 = crc_alignbit((u32)D[5], (u32)D[1], 24), a2 = crc_alignbit((u32)D[9], (u32)D[5], 24),  a3 = (u32)D[9] >> 24;
    const u32 b0 = (u32)D[2] << 16, b1 = crc_alignbit((u32)D[6], (u32)D[2], 16), b2 = crc_alignbit((u32)D[10], (u32)D[6], 16), b3 = (u32)D[10] >> 16;
    const u32 c0 = (u32)D[3] << 24, c1 = crc_alignbit((u32)D[7], (u32)D[3], 8),  c2 = crc_alignbit((u32)D[11], (u32)D[7], 8),  c3 = (u32)D[11] >> 8;
    auto w128 = [](u32 w3, u32 w2, u32 w1, u32 w0) { return ((u128)(((u64)w3 << 32) | w2) << 64) | (((u64)w1 << 32) | w0); };
    const u128 U = w128((u32)D[12], (u32)D[8], (u32)D[4], (u32)D[0]) + w128(a3, a2, a1, a0) + w128(b3, b2, b1, b0) + w128(c3, c2, c1, c0);
    const u64 ulo = (u64)U, uhi = (u64)(U >> 64);
    const u64 mq = ulo * qinv;
    const u64 t = uhi - mulhi64(mq, m.q) + m.q;                  // in (0, 2^63 + q)
    if (m.fold) {                                                 // 2^b = f (mod q), b >= 52: (t >> b) < 2^12 lives in the upper word
        const u32 th = (u32)(t >> 32), sb = m.bits - 32;
        const u64 r = (((u64)(th & ((1u << sb) - 1)) << 32) | (u32)t) + (u64)(th >> sb) * (u32)m.fold;
        return r >= m.q ? r - m.q : r;
    }
    return barrett128(t, 0, m);
}

// short form (kernels_mfma1.hip: 64-term reductions, 2^52 < q < 2^55): all D'_d < 2^24 with D'_odd < 0.94 2^24, 0 < U < 2^115.6.  PAIRS of diagonals
// P_j = D'_2j + D'_2j+1 2^8 < 2^32 sit at bit 16 j: even pairs are the words of one 128-bit number, odd pairs of another, 16 bits up.  t in (-q, 2^51.6]: one
// conditional add, no second reduction.
CRC_HD u64 diag_reduce_short(const int (&D)[13], u64 q, u64 qinv)
{
    u32 P[7];
    for (int j = 0; j < 6; j++) P[j] = (u32)D[2 * j] + ((u32)D[2 * j + 1] << 8);
    P[6] = (u32)D[12];
    const u32 o0 = P[1] << 16, o1 = (P[3] << 16) | (P[1] >> 16), o2 = (P[5] << 16) | (P[3] >> 16), o3 = P[5] >> 16;          // (v_alignbit_b32)
    const u64 elo = ((u64)P[2] << 32) | P[0], ehi = ((u64)P[6] << 32) | P[4], olo = ((u64)o1 << 32) | o0, ohi = ((u64)o3 << 32) | o2;
    const u64 ulo = elo + olo, uhi = ehi + ohi + (ulo < elo);
    const u64 mq = ulo * qinv;
    const long long t = (long long)(uhi - mulhi64(mq, q));
    return (u64)(t + ((t >> 63) & (long long)q));
}

// the same, returning the CENTRED representative of (U 2^-64 + bias) mod q in [-(q-1)/2, (q-1)/2] -- what the limb form of the next layer is made of -- in one pass:
// t in (-q, 2^51.6] plus a centred bias in [-(q-1)/2, (q-1)/2] lies in (-1.5 q, q), so one conditional +q and (only when a bias was added) one conditional -q
// replace the three corrections of canonicalise / add the bias mod q / centre
template <bool HAS_BIAS>
CRC_HD long long diag_reduce_short_centred(const int (&D)[13], u64 q, u64 qinv, long long bias_centred)
{
    typedef unsigned __int128 u128;
    u32 P[7];
    for (int j = 0; j < 6; j++) P[j] = (u32)D[2 * j] + ((u32)D[2 * j + 1] << 8);
    P[6] = (u32)D[12];
    const u128 E = ((u128)(((u64)P[6] << 32) | P[4]) << 64) | (((u64)P[2] << 32) | P[0]);
    const u128 O = ((u128)P[5] << 64) | (((u64)P[3] << 32) | P[1]);
    const u128 U = E + (O << 16);
    const u64 mq = (u64)U * qinv;
    long long t = (long long)((u64)(U >> 64) - mulhi64(mq, q));
    const long long h = (long long)(q >> 1);
    if (HAS_BIAS) t += bias_centred;
    t += t < -h ? (long long)q : 0;
    if (HAS_BIAS) t -= t > h ? (long long)q : 0;
    return t;
}
// Round 4: the same value WITHOUT the Montgomery step.  For q = 2^b - f (every modulus the one-channel kernel accepts: 53 <= b <= 55, f < 2^26) the 116-bit U folds
// down as  U = Uh 2^b + Ul  ==  Ul + Uh f:  three folds (Uh < 2^63, then < 2^32, then < 32) leave a value below 2^b + 2^31 < 2q.  Multiplies: two 32 x 32 -> 64 and
// one more for the second fold, one 32-bit low product for the third -- against seven 32 x 32 products of m = U_lo q^-1 and hi64(m q); v_mad_u64_u32 / v_mul_lo_u32 are
// quarter rate, so they were 112 of the ~370 issue cycles of an output.  The weights then carry NO 2^64 factor (limb_pack_w1_kernel).  Returns the centred
// representative of (U + bias) mod q, bias centred: in [-(q-1)/2, (q-1)/2].  conv1_fold_ok() is the precondition (checked on the host).
CRC_HD bool conv1_fold_ok(u64 q, u32 bits, u32 fold)
{
    typedef unsigned __int128 u128;
    // U < 2^115.6 < 0xC3 2^108 (conv1_bias_table); the second fold wants (Ul + (U >> b) f) >> b below 2^32; the third multiplies a value below 2^5 by f < 2^26
    if (bits < 53 || bits > 55 || fold == 0 || fold >= (1u << 26) || (((u64)1 << bits) - fold) != q) return false;
    const u128 uh_max = (((u128)0xC3) << 108) >> bits;
    return uh_max * fold + ((u128)1 << bits) < ((u128)1 << (bits + 32));
}
// (written in 32-bit words: b >= 53 puts every shift by b inside the upper words -- v_alignbit_b32 -- and every multiply-add is one v_mad_u64_u32 with its 64-bit addend;
// the compiler's own lowering of the same arithmetic on 64-bit values spent 22 64-bit shifts and 9 64-bit adds per output)
// PB: the accumulator biases of conv1_bias_table pair by pair, PB_j = B_2j + B_2j+1 2^8 (j < 6), PB_6 = B_12 -- null when the diagonals arrive biased already.  (The
// kernel starts its accumulators at ZERO -- an inline constant of the first MFMA that touches a diagonal, instead of 13 x 4 register moves per tile -- and adds the biases
// here, one word add per pair: D + B is the same word either way.)
// the two halves of the reduction, so that a kernel can move the four words of U between lanes before it folds them (kernels_mfma1.hip: the 4 leftover filters of a
// 20-filter layer sit in a quarter of a wave's lanes; the words of four tiles are gathered into one full wave before the expensive half runs)
CRC_HD void diag_pack_words(const int (&D)[13], const u32 *PB, u32 (&u)[4])
{
    u32 P[7];
    for (int j = 0; j < 6; j++) P[j] = (u32)D[2 * j] + ((u32)D[2 * j + 1] << 8);
    P[6] = (u32)D[12];
    if (PB) for (int j = 0; j < 7; j++) P[j] += PB[j];
    // U = E + (O << 16), E = words (P0, P2, P4, P6), O = words (P1, P3, P5): four words u0..u3 (u3 < 2^20)
    const u32 o0 = P[1] << 16, o1 = crc_alignbit(P[3], P[1], 16), o2 = crc_alignbit(P[5], P[3], 16), o3 = P[5] >> 16;
    const u64 lo64 = (((u64)P[2] << 32) | P[0]) + (((u64)o1 << 32) | o0);
    const u64 hi64 = (((u64)P[6] << 32) | P[4]) + (((u64)o3 << 32) | o2) + (lo64 < (((u64)P[2] << 32) | P[0]));
    u[0] = (u32)lo64; u[1] = (u32)(lo64 >> 32); u[2] = (u32)hi64; u[3] = (u32)(hi64 >> 32);
}
CRC_HD long long fold_words_centred(u32 u0, u32 u1, u32 u2, u32 u3, u64 q, u32 bits, u32 fold, long long bias_centred)
{
    const u32 sb = bits - 32, m1 = (1u << sb) - 1;
    // fold 1: Uh = U >> b = (h1:h0) < 2^63;  x1 = Ul + h0 f + (h1 f << 32)  as words (A.lo, B.lo, B.hi)
    const u32 h0 = crc_alignbit(u2, u1, sb), h1 = crc_alignbit(u3, u2, sb);
    const u64 A = (u64)h0 * fold + (((u64)(u1 & m1) << 32) | u0);      // < 2^58 + 2^55
    const u64 B = (u64)h1 * fold + (A >> 32);                           // < 2^57
    // fold 2: Uh1 = x1 >> b < 2^32
    const u32 uh1 = crc_alignbit((u32)(B >> 32), (u32)B, sb);
    const u64 x2 = (u64)uh1 * fold + (((u64)((u32)B & m1) << 32) | (u32)A);      // < 2^b + 2^58
    // fold 3: x2 >> b < 2^5
    u64 r = (u64)((u32)(x2 >> 32) >> sb) * fold + (((u64)((u32)(x2 >> 32) & m1) << 32) | (u32)x2);      // < 2^b + 2^31 < 2 q
    // centred representative of r + bias: r - q + bias lies in (-1.5 q, q/2 + 2^31): one conditional + q, then (the bias may have pushed it past q/2) one conditional - q
    long long t = (long long)(r - q) + bias_centred;
    const long long h = (long long)(q >> 1);
    t += t < -h ? (long long)q : 0;
    t -= t > h ? (long long)q : 0;
    return t;
}
CRC_HD long long diag_fold_short_centred(const int (&D)[13], u64 q, u32 bits, u32 fold, long long bias_centred, const u32 *PB = nullptr)
{
    u32 u[4];
    diag_pack_words(D, PB, u);
    return fold_words_centred(u[0], u[1], u[2], u[3], q, bits, fold, bias_centred);
}
CRC_HD u64 centred_digit_bytes(long long cv) { return ((u64)cv + 0x0080808080808080ULL) ^ 0x0080808080808080ULL; }

// the 7 balanced base-256 digits of a canonical residue's centred representative, one per byte: the bytes of (centred value + 0x80...80) with their top bits flipped
CRC_HD u64 balanced_digit_bytes(u64 r, u64 q)
{
    const long long cv = r > (q >> 1) ? (long long)(r - q) : (long long)r;
    return ((u64)cv + 0x0080808080808080ULL) ^ 0x0080808080808080ULL;
}

// ---- tables (host) ------------------------------------------------------------------------------------------------------------------------------------------
inline u64 inverse_mod_2_64(u64 q) { u64 inv = q; for (int it = 0; it < 6; it++) inv *= 2 - q * inv; return inv; }      // Newton: q odd, q q = 1 (mod 8)

// move K0 = sum B_d 2^(8d) to the nearest multiple of q by adding the balanced digits of the distance (|.| <= q/2 < 2^62: at most 8 digits of at most 128) to B_0..7
inline void bias_to_multiple(u64 q, long long (&B)[13])
{
    typedef unsigned __int128 u128;
    u128 K0 = 0;
    for (int d = 0; d < 13; d++) K0 += (u128)(u64)B[d] << (8 * d);
    const u64 rem = (u64)(K0 % q);
    long long delta = rem > q / 2 ? (long long)(q - rem) : -(long long)rem;
    for (int d = 0; d < 8 && delta; d++) { const long long dg = (long long)(signed char)(delta & 0xff); B[d] += dg; delta = (delta - dg) >> 8; }
}
// generic form, reductions of T terms: diagonal d collects np_d = min(d, 12 - d) + 1 products of every term, |D_d| <= T np_d 2^14; B_d = the power of two above that
// bound (+ 256: room for the digit adjustment) keeps D'_d positive and below 2^32 for T <= 18 000
inline void limb_bias_table(u64 q, int T, int (&out)[13])
{
    long long B[13];
    for (int d = 0; d < 13; d++) {
        const u64 bound = (u64)T * (u64)((d < 12 - d ? d : 12 - d) + 1) * 16384 + 256;
        u64 b = 1; while (b < bound) b <<= 1;
        B[d] = (long long)b;
    }
    bias_to_multiple(q, B);
    for (int d = 0; d < 13; d++) out[d] = (int)(u32)(u64)B[d];                                   // (B_d up to 2^31: the accumulators are words mod 2^32)
}
// short form, 64 terms: digits are in [-128, 127], the top ones (l, m = 6) in [-64, 64] because |centred residue| < 2^54:  |D_d| <= 64 * 7 * 2^14 < 0.877 2^23
// (d <= 10), |D_11| <= 2^20, |D_12| <= 2^18.  B = 2^23 (d <= 10), 2^21, 2^19 keep every D'_d inside (0, 0.94 2^24) and a pair D'_2j + D'_2j+1 2^8 inside 32 bits;
// K0 ~ 2^115.02 exceeds |V| <= 64 (q/2)^2 <= 2^114, so U > 0, and U < 2^115.6
inline void conv1_bias_table(u64 q, int (&out)[13])
{
    long long B[13];
    for (int d = 0; d < 13; d++) B[d] = d <= 10 ? 1 << 23 : d == 11 ? 1 << 21 : 1 << 19;
    bias_to_multiple(q, B);
    for (int d = 0; d < 13; d++) out[d] = (int)B[d];
}
