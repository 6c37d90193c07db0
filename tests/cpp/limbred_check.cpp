// Host-side check of crcnn_amd/csrc/limbred.h -- the once-per-output reduction of the matrix-core kernels and its bias tables -- against 128-bit integer arithmetic:
// random and extreme centred residues are split into balanced base-256 digits exactly as the pack kernels do, the 13 diagonals are accumulated in 32-bit words
// starting from the bias table (as the MFMA accumulators are), and the reduced value must be sum_t x_t w_t mod q when the weights carry the factor 2^64 mod q.
#include "limbred.h"
#include <cstdio>
#include <cstdlib>
typedef unsigned __int128 u128;
static ModParams make(u64 q, bool allow_fold)
{
    ModParams m{}; u128 all = ~(u128)0, quo = all / q;
    m.q = q; m.r0 = (u64)quo; m.r1 = (u64)(quo >> 64); m.two_q = 2 * q; m.bits = 64 - __builtin_clzll(q);
    m.fold = allow_fold ? fold_constant(q, m.bits) : 0;
    return m;
}
static void digits(u64 r, u64 q, int (&d)[7])
{
    const u64 b = balanced_digit_bytes(r, q);
    long long back = 0;
    for (int l = 6; l >= 0; l--) { d[l] = (int)(signed char)(b >> (8 * l)); back = back * 256 + d[l]; }
    const long long cv = r > (q >> 1) ? (long long)(r - q) : (long long)r;
    if (back != cv || (b >> 56)) { printf("balanced_digit_bytes wrong for %llx mod %llx\n", (unsigned long long)r, (unsigned long long)q); exit(1); }
}
int main()
{
    // the coefficient moduli of the reference's parameter sets (54-55 bits: both forms), smaller ones of the same shape and one without the 2^b - d form (generic form only)
    const u64 qs[] = {0x7fffffff380001ULL, 0x3fffffff000001ULL, 0x7ffffffef00001ULL, 0x3ffffffef40001ULL, 0x7ffffffeac0001ULL, 0x7ffffffe700001ULL, 0x7ffffffe600001ULL,
                      0x7ffffffe4c0001ULL, 0x1fffffffd80001ULL /* 53 bits */, 0xffffffff00001ULL /* 52 */, 0x3ffffffb80001ULL /* 50 */, 0xffffe80001ULL /* 40 */,
                      18014398509481951ULL /* 2^54 - 33: not of the SEAL shape; odd */};
    u64 x = 88172645463325252ULL; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long checked = 0, folded_checked = 0;
    for (u64 q : qs) for (int fold = 0; fold < 2; fold++) {
        const ModParams m = make(q, fold);
        if (fold && !m.fold) continue;
        const u64 qinv = inverse_mod_2_64(q), R = (u64)((((u128)1) << 64) % q);
        if (q * qinv != 1) { printf("inverse_mod_2_64 wrong\n"); return 1; }
        const u64 edge[] = {0, 1, q - 1, q / 2, q / 2 + 1, q / 2 - 1, 0x7f7f7f7f7f7f7fULL % q, 0x80808080808080ULL % q, (q / 2) & ~0xffffffffffULL};
        for (int Tsel = 0; Tsel < 6; Tsel++) {
            const int T = (const int[]){64, 64, 320, 1152, 4096, 18000}[Tsel];
            const bool shortform = Tsel == 0 && m.bits >= 53 && m.bits <= 55;             // kernels_mfma1.hip's form and table
            if (Tsel == 0 && !shortform) continue;
            int B[13];
            if (shortform) conv1_bias_table(q, B); else limb_bias_table(q, T, B);
            {   // the table is a multiple of q
                u128 K = 0; for (int d = 0; d < 13; d++) K += (u128)(u32)B[d] << (8 * d);
                if (K % q) { printf("bias table is not a multiple of q=%llx\n", (unsigned long long)q); return 1; }
            }
            const int reps = T > 2000 ? 12 : 600;
            for (int rep = 0; rep < reps; rep++) {
                u32 D[13]; for (int d = 0; d < 13; d++) D[d] = (u32)B[d];
                u128 want = 0;
                const int mode = rep % 6;            // 0-1 random, 2 all extreme same sign, 3 extreme alternating, 4 edge values, 5 zeros
                for (int t = 0; t < T; t++) {
                    u64 xv, sv;                      // sv: the weight as the pack kernels store it (w 2^64 mod q); the extreme modes choose ITS digits
                    if (mode <= 1) { xv = rnd() % q; sv = rnd() % q; }
                    else if (mode == 2) { xv = q / 2; sv = q / 2; }
                    else if (mode == 3) { xv = (t & 1) ? q / 2 : q / 2 + 1; sv = (t & 2) ? q / 2 : q / 2 + 1; }
                    else if (mode == 4) { xv = edge[rnd() % 9]; sv = edge[rnd() % 9]; }
                    else { xv = 0; sv = 0; }
                    int a[7], b[7]; digits(xv, q, a); digits(sv, q, b);
                    for (int l = 0; l < 7; l++) for (int mm = 0; mm < 7; mm++) D[l + mm] += (u32)(a[l] * b[mm]);      // int32 accumulators, mod 2^32
                    want = (want + (u128)xv * sv) % q;
                }
                // the kernels return sum x (sv 2^-64) mod q
                u64 Rinv; { __int128 r0 = q, r1 = R % q, t0 = 0, t1 = 1; while (r1) { const __int128 qq = r0 / r1, r2 = r0 - qq * r1, t2 = t0 - qq * t1; r0 = r1; r1 = r2; t0 = t1; t1 = t2; } Rinv = (u64)((t0 % (__int128)q + q) % q); }   // extended Euclid (q odd: gcd(2^64, q) = 1)
                const u64 expect = (u64)(want * (u128)Rinv % q);
                int Ds[13]; for (int d = 0; d < 13; d++) Ds[d] = (int)D[d];
                if (shortform) {
                    for (int d = 0; d < 13; d++) if (D[d] >= (1u << 24)) { printf("short form: diagonal %d out of range q=%llx\n", d, (unsigned long long)q); return 1; }
                    if (diag_reduce_short(Ds, q, qinv) != expect) { printf("diag_reduce_short mismatch q=%llx mode=%d\n", (unsigned long long)q, mode); return 1; }
                    // the one-pass centred form, without and with a bias
                    const long long hq = (long long)(q >> 1);
                    auto centre = [&](u64 r) { return r > (q >> 1) ? (long long)(r - q) : (long long)r; };
                    if (diag_reduce_short_centred<false>(Ds, q, qinv, 0) != centre(expect)) { printf("diag_reduce_short_centred mismatch q=%llx mode=%d\n", (unsigned long long)q, mode); return 1; }
                    const u64 bias = mode == 4 ? edge[rnd() % 9] : rnd() % q;
                    const long long got = diag_reduce_short_centred<true>(Ds, q, qinv, centre(bias));
                    if (got != centre((u64)(((u128)expect + bias) % q)) || got < -hq || got > hq) { printf("diag_reduce_short_centred (bias) mismatch q=%llx mode=%d\n", (unsigned long long)q, mode); return 1; }
                    if (centred_digit_bytes(got) != balanced_digit_bytes((u64)(((u128)expect + bias) % q), q)) { printf("centred_digit_bytes mismatch\n"); return 1; }
                    // round 4's form: the folding reduction of the same diagonals gives sum x sv mod q itself (no 2^64 factor to divide out), centred, bias included
                    if (conv1_fold_ok(q, m.bits, fold_constant(q, m.bits))) {
                        const u32 f = fold_constant(q, m.bits);
                        const u64 plain = (u64)want;
                        if (diag_fold_short_centred(Ds, q, m.bits, f, 0) != centre(plain)) { printf("diag_fold_short_centred mismatch q=%llx mode=%d\n", (unsigned long long)q, mode); return 1; }
                        {   // the kernel's form: accumulators started at zero, the biases added pair by pair inside the reduction
                            int Dz[13]; u32 PB[7];
                            for (int d = 0; d < 13; d++) Dz[d] = (int)(D[d] - (u32)B[d]);
                            for (int j = 0; j < 6; j++) PB[j] = (u32)B[2 * j] + ((u32)B[2 * j + 1] << 8);
                            PB[6] = (u32)B[12];
                            if (diag_fold_short_centred(Dz, q, m.bits, f, 0, PB) != centre(plain)) { printf("diag_fold_short_centred (zero start) mismatch q=%llx mode=%d\n", (unsigned long long)q, mode); return 1; }
                        }
                        const long long g2 = diag_fold_short_centred(Ds, q, m.bits, f, centre(bias));
                        if (g2 != centre((u64)(((u128)plain + bias) % q)) || g2 < -hq || g2 > hq) { printf("diag_fold_short_centred (bias) mismatch q=%llx mode=%d\n", (unsigned long long)q, mode); return 1; }
                        folded_checked++;
                    }
                }
                {
                    if (shortform) { int Bg[13]; limb_bias_table(q, T, Bg); for (int d = 0; d < 13; d++) Ds[d] = (int)(D[d] - (u32)B[d] + (u32)Bg[d]); }
                    if (diag_reduce_w(Ds, m, qinv) != expect) { printf("diag_reduce_w mismatch q=%llx T=%d mode=%d fold=%d\n", (unsigned long long)q, T, mode, fold); return 1; }
                    if (diag_reduce(Ds, m, qinv) != expect) { printf("diag_reduce mismatch q=%llx T=%d mode=%d fold=%d\n", (unsigned long long)q, T, mode, fold); return 1; }
                }
                checked++;
            }
        }
    }
    if (folded_checked < 1000) { printf("the folding short form was not exercised (%ld)\n", folded_checked); return 1; }
    printf("ok %ld\n", checked);
    return 0;
}
