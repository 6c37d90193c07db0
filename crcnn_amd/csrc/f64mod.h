// f64mod.h -- exact modular arithmetic on integers held in doubles (host + device: tests/cpp/f64mod_check.cpp runs the same code against 128-bit integers).
//
// gfx950 has no 64 x 64 integer multiplier -- a 55-bit modular multiplication costs ~50 issue cycles per wave (3 half-rate 32 x 32 high products, two 64-bit low
// products) -- but v_fma_f64 issues at full rate and gives an exact 53 x 53 -> 106-bit product in two instructions (h = a b rounded, l = fma(a, b, -h) = the rounding
// error, exactly).  Wherever the engine may choose its own moduli -- the key-switching inner products of relinearisation, whose digit polynomials have 16-bit
// coefficients, so that the result is an integer the engine can compute modulo ANY primes whose product exceeds its size -- it computes modulo primes p < 2^47 in
// fp64: 6 flops per multiplication (profiles/r03_microbench_valu_f64.txt: 24.6 vs 49.6 cycles), and two such primes replace k coefficient moduli.
//
// Conventions.  Values are integers stored exactly in doubles, |x| < 2^52 (every integer below 2^53 is representable).  A constant multiplier w is the centred residue
// (|w| <= p/2) together with wq = w / p rounded to double.  Proofs of the bounds are in the comments; the CPU test checks them on random and extreme operands.
#pragma once
#include "modarith.h"

#define CRC_F64_PRIME_BITS 47          // primes just below 2^47, = 1 mod 2^16 (negacyclic NTTs up to n = 32768)

struct F64Mod { double p, pinv; };     // pinv = 1 / p rounded

// w y mod p for |y| < 2^52, |w| <= p/2 < 2^46:  returns T == w y (mod p) with |T| < 0.875 p, exactly.
//   h = fl(w y), l = w y - h exactly (an integer, |l| <= ulp(h)/2 <= 2^45); c = rint(fl(y wq)) differs from w y / p by at most 0.5 + |y| 2^-55 + 2^-2 <= 0.875;
//   h - c p is an integer below 2^48, so the fma that forms it rounds nothing, and adding l stays far below 2^53.
CRC_HD double f64_mulmod_const(double y, double w, double wq, double p)
{
    const double h = w * y;
    const double l = __builtin_fma(w, y, -h);
    const double c = __builtin_rint(y * wq);
    return __builtin_fma(-c, p, h) + l;
}
// a b mod p for |a| <= (p + 1)/2 (a twiddle or a residue as f64_reduce leaves it), |b| < 2^51 (no precomputed quotient: fl(h pinv) carries three roundings of a value below 2^50, so c = rint(.) differs from a b / p by
// at most 0.5 + 3 2^-3): |result| < 0.875 p
CRC_HD double f64_mulmod(double a, double b, const F64Mod &m)
{
    const double h = a * b;
    const double l = __builtin_fma(a, b, -h);
    const double c = __builtin_rint(h * m.pinv);
    return __builtin_fma(-c, m.p, h) + l;
}
// |x| < 2^53 (every such integer is exact in a double: relinearisation's lazy sums of 48 products reach 42 p = 2^52.4)  ->  a residue with |result| <= (p + 1) / 2; for |x| <= 8 p it is exactly the centred one (x / p is at least 1/(2p) = 2^-48 away from a half-integer and
// fl(x pinv) is off by at most |x / p| 2^-52).  Further out a value within 2^-48 |x / p| of a half-integer may land on the other side: (p + 1)/2 instead of
// -(p - 1)/2 -- still the right class, one past the centred range, which is all the transforms and the CRT step need (their true values are nowhere near p/2).
CRC_HD double f64_reduce(double x, const F64Mod &m)
{
    return __builtin_fma(-__builtin_rint(x * m.pinv), m.p, x);
}
// signed 64-bit integer (|v| < 2^62) -> centred residue mod p as a double.  The double nearest to v is off by at most 2^9, so the quotient estimate is off by at most
// one: the remainder, computed in integers, is below 2 p and fits a double exactly; one more reduction centres it.
CRC_HD double f64_from_i64(long long v, const F64Mod &m)
{
    const long long qe = (long long)__builtin_rint((double)v * m.pinv);
    const long long r = v - qe * (long long)m.p;
    return f64_reduce((double)r, m);
}
