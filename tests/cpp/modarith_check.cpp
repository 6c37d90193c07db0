// Host-side check of crcnn_amd/csrc/modarith.h (shared by the host table builders and the gfx950 kernels): barrett128 / fold128 /
// mulmod / Shoup multiplications against unsigned __int128 arithmetic for every prime the reference's parameter sets use
// (coeff_modulus_128, SEAL util/globals.cpp:25-90; auxiliary 61-bit primes :321-367) plus a prime without the 2^b - d form.
#include "modarith.h"
#include <cstdio>
typedef unsigned __int128 u128;
static ModParams make(u64 q, bool allow_fold)
{
    ModParams m{}; u128 all = ~(u128)0, quo = all / q;
    m.q = q; m.r0 = (u64)quo; m.r1 = (u64)(quo >> 64); m.two_q = 2 * q; m.bits = 64 - __builtin_clzll(q);
    m.fold = allow_fold ? fold_constant(q, m.bits) : 0;        // the product's own rule (ctx.cpp make_mod)
    return m;
}
int main()
{
    const u64 qs[] = {0x7fffffff380001ULL, 0x3fffffff000001ULL, 0x7ffffffef00001ULL, 0x3ffffffef40001ULL, 0x7ffffffeac0001ULL, 0x7ffffffe700001ULL,
                      0x7ffffffe600001ULL, 0x7ffffffe4c0001ULL, 0x1fffffffffe00001ULL, 0x1fffffffffc80001ULL, 0x1fffffffffb40001ULL, 0x1fffffffff500001ULL,
                      1152921504606584833ULL /* 2^60 - 2^18 + 1 */, 4611686018326724609ULL /* 62 bits */,
                      // primes of the same 2^b - c 2^s + 1 shape below the folding bound: SEAL's small_mods_40bit (util/globals.cpp) and 45..53-bit ones.
                      // Below 52 bits fold_constant must refuse them (three folds no longer reach [0, 2q)) and the generic Barrett path must be exact
                      0xffffe80001ULL, 0xffffc40001ULL, 0x1fffff980001ULL, 0xfffffff00001ULL, 0x3ffffffb80001ULL, 0x3fffffec80001ULL, 0x7ffffff9c0001ULL,
                      0xffffffff00001ULL, 0xfffffffe40001ULL, 0x1fffffffd80001ULL};
    // known answers the reference's authors wrote down (SEAL 2.3.1 SEALTest/util/uintarithsmallmod.cpp:143-213: BarrettReduce128, MultiplyUIntUIntSmallMod), on
    // the product's barrett128 / mulmod -- data only; tests/test_seal_known_answers.py replays them on the oracle
    {
        const u64 ALL = ~0ULL, M62 = 4611686018427289601ULL;
        const u64 br[][4] = {{0, 0, 2, 0}, {1, 0, 2, 1}, {ALL, ALL, 2, 1}, {0, 0, 3, 0}, {1, 0, 3, 1}, {123, 456, 3, 0}, {ALL, ALL, 3, 0}, {0, 0, 13131313131313ULL, 0},
                             {1, 0, 13131313131313ULL, 1}, {123, 456, 13131313131313ULL, 8722750765283ULL}, {24242424242424ULL, 79797979797979ULL, 13131313131313ULL, 1010101010101ULL}};
        for (auto &v : br) { const ModParams m = make(v[2], true); if (barrett128(v[0], v[1], m) != v[3]) { printf("SEALTest BarrettReduce128 vector failed: mod %llu\n", (unsigned long long)v[2]); return 1; } }
        const u64 mm[][4] = {{0, 0, 2, 0}, {0, 1, 2, 0}, {1, 0, 2, 0}, {1, 1, 2, 1}, {0, 0, 10, 0}, {0, 1, 10, 0}, {1, 0, 10, 0}, {1, 1, 10, 1}, {7, 7, 10, 9}, {6, 7, 10, 2}, {7, 6, 10, 2},
                             {0, 0, M62, 0}, {0, 1, M62, 0}, {1, 0, M62, 0}, {1, 1, M62, 1}, {2305843009213644800ULL, 2305843009213644801ULL, M62, 1152921504606822400ULL},
                             {2305843009213644801ULL, 2305843009213644800ULL, M62, 1152921504606822400ULL}, {2305843009213644801ULL, 2305843009213644801ULL, M62, 3458764513820467201ULL},
                             {4611686018427289600ULL, 4611686018427289600ULL, M62, 1}};
        for (auto &v : mm) { const ModParams m = make(v[2], true); if (mulmod(v[0], v[1], m) != v[3]) { printf("SEALTest MultiplyUIntUIntSmallMod vector failed: mod %llu\n", (unsigned long long)v[2]); return 1; } }
    }
    u64 x = 88172645463325252ULL; auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long checked = 0;
    for (u64 q : qs) for (int fold = 0; fold < 2; fold++) {
        const ModParams m = make(q, fold);
        if (fold && !m.fold) continue;
        if (m.fold && m.bits < 52) { printf("fold enabled for a %u-bit modulus\n", m.bits); return 1; }
        const u128 qq = (u128)q * q - 1;
        for (long it = 0; it < 400000; it++) {
            u64 lo = rnd(), hi = rnd();
            if (it % 7 == 0) hi = ~0ULL; if (it % 11 == 0) lo = ~0ULL; if (it % 13 == 0) hi = 0; if (it % 17 == 0) { hi = (u64)(qq >> 64); lo = (u64)qq; } if (it % 19 == 0) { hi = 0; lo = q - 1 + (it & 1); }
            const u64 e = (u64)(((((u128)hi) << 64) | lo) % q);
            if (barrett128(lo, hi, m) != e) { printf("barrett128 mismatch q=%llx fold=%d hi=%llx lo=%llx\n", (unsigned long long)q, fold, (unsigned long long)hi, (unsigned long long)lo); return 1; }
            {   // what sq_floor / sq_lift feed it: sums of k products of a 61-bit value and a residue of this modulus
                u128 acc = 0; for (int j = 0; j < 3; j++) acc += (u128)(rnd() >> 3) * (rnd() % q);
                if (barrett128((u64)acc, (u64)(acc >> 64), m) != (u64)(acc % q)) { printf("barrett128 (sum of products) mismatch q=%llx fold=%d\n", (unsigned long long)q, fold); return 1; }
            }
            const u64 a = rnd() % q, b = rnd() % q, wp = (u64)(((u128)b << 64) / q);
            if (mulmod(a, b, m) != (u64)((u128)a * b % q)) { printf("mulmod mismatch q=%llx\n", (unsigned long long)q); return 1; }
            const u64 any = rnd();
            if (mulmod_shoup(any, b, wp, q) != (u64)((u128)any * b % q)) { printf("mulmod_shoup mismatch q=%llx\n", (unsigned long long)q); return 1; }
            const u64 lz = mulmod_shoup_lazy(any, b, wp, q);
            if (lz >= 2 * q || lz % q != (u64)((u128)any * b % q)) { printf("mulmod_shoup_lazy mismatch q=%llx\n", (unsigned long long)q); return 1; }
            if (m.fold && m.bits >= 50 && m.bits <= 55) {                  // the MAC epilogue: limbs up to their lazy bounds, overflow counters up to 1023
                u64 A0 = rnd(), A1 = rnd(), A2 = rnd(); u32 ov = (u32)rnd() & 0x3fffffffu;
                if (it % 5 == 0) { A0 = ~0ULL; A1 = ~0ULL; A2 = ~0ULL; ov = 0x3fffffffu; } if (it % 9 == 0) { A0 = A1 = A2 = 0; } if (it % 3 == 0) ov = 0;
                const u128 a0 = (u128)A0 + ((u128)(ov & 1023) << 63), a1 = (u128)A1 + ((u128)((ov >> 10) & 1023) << 63), a2 = (u128)A2 + ((u128)((ov >> 20) & 1023) << 63);
                const u64 r0 = (u64)(a0 % q), r1 = (u64)(a1 % q), r2 = (u64)(a2 % q);
                const u64 midv = (u64)(((u128)r1 + 2 * (u128)q - r0 - r2) % q);
                const u64 want = (u64)(((u128)r0 + (u128)midv * (u64)(((u128)1 << 28) % q) % q + (u128)r2 * (u64)(((u128)1 << 56) % q) % q) % q);
                if (mac_reduce_fold(A0, A1, A2, ov, m) != want) { printf("mac_reduce_fold mismatch q=%llx A=%llx %llx %llx ov=%x\n", (unsigned long long)q, (unsigned long long)A0, (unsigned long long)A1, (unsigned long long)A2, ov); return 1; }
            }
            checked++;
        }
    }
    printf("ok %ld\n", checked);
    return 0;
}
