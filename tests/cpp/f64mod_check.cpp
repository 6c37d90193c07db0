// Host-side check of crcnn_amd/csrc/f64mod.h -- exact modular arithmetic on integers held in doubles, the arithmetic of the fp64-prime NTTs of relinearisation --
// against 128-bit integer arithmetic: the bounds the header states (|result| < 0.875 p, exactness for |y| < 2^52) on random and extreme operands, the centring
// reduction at its boundaries, the integer -> residue conversion, and a whole forward / inverse negacyclic transform pair without intermediate reductions
// (values grow by at most 0.875 p per forward stage) against an O(n^2) evaluation in integers.
#include "f64mod.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __int128 i128;
typedef unsigned __int128 u128;
static u64 mulm(u64 a, u64 b, u64 p) { return (u64)((u128)a * b % p); }
static u64 powm(u64 a, u64 e, u64 p) { u64 r = 1; for (; e; e >>= 1) { if (e & 1) r = mulm(r, a, p); a = mulm(a, a, p); } return r; }
static long long centre(long long v, long long p) { v %= p; if (v < 0) v += p; return v > p / 2 ? v - p : v; }
static bool is_prime(u64 n)
{
    if (n < 2) return false;
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) { if (n % a == 0) return n == a; }
    u64 d = n - 1; int s = 0; while (!(d & 1)) { d >>= 1; s++; }
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = powm(a, d, n); if (x == 1 || x == n - 1) continue;
        bool comp = true; for (int r = 1; r < s; r++) { x = mulm(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}
int main()
{
    u64 st = 88172645463325252ULL; auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    long checked = 0;
    // the primes the engine picks: the largest ones below 2^47 that are 1 mod 2^16
    std::vector<u64> primes;
    for (u64 c = ((u64)1 << CRC_F64_PRIME_BITS) - 65536 + 1; primes.size() < 4; c -= 65536) if (is_prime(c)) primes.push_back(c);
    for (u64 pu : primes) {
        const long long p = (long long)pu;
        F64Mod m{(double)p, 1.0 / (double)p};
        if ((long long)m.p != p) { printf("prime not representable\n"); return 1; }
        const long long half = p / 2;
        const long long wedge[] = {0, 1, -1, half, -half, half - 1, 12345, -(half - 7)};
        const long long yedge[] = {0, 1, -1, ((long long)1 << 52) - 1, -(((long long)1 << 52) - 1), half, p, -p, 14 * p, -14 * p + 3, ((long long)1 << 51) + 1, 65535};
        for (int it = 0; it < 400000; it++) {
            long long w = it < 64 ? wedge[it % 8] : centre((long long)(rnd() >> 2), p);
            long long y = it < 96 ? yedge[it % 12] : ((long long)(rnd() >> 11) - ((long long)1 << 52));       // uniform in (-2^52, 2^52)
            if (y >= ((long long)1 << 52) || y <= -((long long)1 << 52)) continue;
            const double wq = (double)((long double)w / (long double)p);
            const double T = f64_mulmod_const((double)y, (double)w, wq, m.p);
            const i128 exact = (i128)w * y;
            if (T != std::floor(T) || std::fabs(T) >= 0.875 * m.p || (long long)((exact - (i128)(long long)T) % p) != 0) { printf("f64_mulmod_const wrong: w %lld y %lld -> %.1f\n", w, y, T); return 1; }
            if (std::llabs(y) < ((long long)1 << 51)) {
                const double U = f64_mulmod((double)w, (double)y, m);
                if (U != std::floor(U) || std::fabs(U) >= 0.875 * m.p || (long long)((exact - (i128)(long long)U) % p) != 0) { printf("f64_mulmod wrong: %lld %lld -> %.1f\n", w, y, U); return 1; }
            }
            const double r = f64_reduce((double)y, m);
            if (std::fabs(r) > (double)(half + 1) || centre((long long)r, p) != centre(y, p) || (std::llabs(y) <= 8 * p && (long long)r != centre(y, p))) { printf("f64_reduce wrong: %lld -> %.1f (want %lld)\n", y, r, centre(y, p)); return 1; }
            {   // the widened contracts: f64_reduce up to 2^53 -- relinearisation's lazy sums of 48 products reach 42 p = 2^52.4 -- and f64_mulmod with |a| = (p + 1) / 2,
                // what f64_reduce may leave one past the centred range
                const long long y2 = it % 3 == 0 ? 42 * p - (long long)(rnd() % 1000) : it % 3 == 1 ? -(((long long)1 << 53) - 1 - (long long)(rnd() % 1000)) : (long long)(rnd() >> 12) + ((long long)1 << 52) - 1;
                const double r2 = f64_reduce((double)y2, m);
                if (r2 != std::floor(r2) || std::fabs(r2) > (double)(half + 1) || centre((long long)r2, p) != centre(y2, p)) { printf("f64_reduce (wide) wrong: %lld -> %.1f\n", y2, r2); return 1; }
                const long long a2 = (it & 1) ? half + 1 : -(half + 1);
                if (std::llabs(y) < ((long long)1 << 51)) {
                    const double U2 = f64_mulmod((double)a2, (double)y, m);
                    if (U2 != std::floor(U2) || std::fabs(U2) >= 0.875 * m.p || (long long)(((i128)a2 * y - (i128)(long long)U2) % p) != 0) { printf("f64_mulmod (|a| = (p+1)/2) wrong: %lld %lld -> %.1f\n", a2, y, U2); return 1; }
                }
                checked += 2;
            }
            const long long v = (long long)(rnd() >> 2) - ((long long)1 << 61);
            const double fv = f64_from_i64(v, m);
            if ((long long)fv != centre(v, p)) { printf("f64_from_i64 wrong: %lld -> %.1f\n", v, fv); return 1; }
            checked += 4;
        }
        // reduction right at the centring boundary: (p - 1)/2 and (p + 1)/2 plus multiples of p
        for (long long mult = -30; mult <= 30; mult++) for (long long off : {half, half + 1, -half, -half - 1}) {
            const long long y = mult * p + off;
            if (std::llabs(y) >= ((long long)1 << 52)) continue;
            const double r = f64_reduce((double)y, m);
            if (std::fabs(r) > (double)(half + 1) || centre((long long)r, p) != centre(y, p) || (std::llabs(mult) <= 7 && (long long)r != centre(y, p))) { printf("boundary reduction wrong at %lld\n", y); return 1; }
            checked++;
        }
        // a transform pair of size 64 with the lazy ranges of the kernels: forward Cooley-Tukey without reductions, inverse Gentleman-Sande reducing once per three stages
        const int n = 64, logn = 6;
        u64 psi = 0;
        for (u64 g = 2; !psi; g++) { u64 c = powm(g, (pu - 1) / (2 * n), pu); if (powm(c, n, pu) == pu - 1) psi = c; }
        const u64 ipsi = powm(psi, pu - 2, pu);
        auto brev = [&](int i) { int r = 0; for (int b = 0; b < logn; b++) r |= ((i >> b) & 1) << (logn - 1 - b); return r; };
        std::vector<double> rp(n), rpq(n), irp(n), irpq(n);
        { u64 a = 1, b = 1; for (int i = 0; i < n; i++) { const int j = brev(i);
            const long long ca = centre((long long)a, p), cb = centre((long long)b, p);
            rp[j] = (double)ca; rpq[j] = (double)((long double)ca / (long double)p); irp[j] = (double)cb; irpq[j] = (double)((long double)cb / (long double)p);
            a = mulm(a, psi, pu); b = mulm(b, ipsi, pu); } }
        for (int rep = 0; rep < 50; rep++) {
            std::vector<long long> a(n); std::vector<double> x(n);
            for (int i = 0; i < n; i++) { a[i] = rep == 0 ? 65535 : (long long)(rnd() & 0xffff); x[i] = (double)a[i]; }
            for (int mm = 1, t = n >> 1; mm < n; mm <<= 1, t >>= 1)
                for (int i = 0; i < mm; i++) for (int j = 2 * i * t; j < 2 * i * t + t; j++) {
                    const double T = f64_mulmod_const(x[j + t], rp[mm + i], rpq[mm + i], m.p), X = x[j];
                    x[j] = X + T; x[j + t] = X - T;
                }
            // slot j holds a(psi^(2 bitrev(j) + 1))
            for (int j = 0; j < n; j++) {
                const u64 pt = powm(psi, 2 * (u64)brev(j) + 1, pu); u64 acc = 0, pw = 1;
                for (int i = 0; i < n; i++) { acc = (acc + mulm((u64)a[i], pw, pu)) % pu; pw = mulm(pw, pt, pu); }
                if (std::fabs(x[j]) >= 0.875 * logn * m.p + 65536 || centre((long long)x[j], p) != centre((long long)acc, p)) { printf("forward transform wrong at slot %d\n", j); return 1; }
            }
            int stage = 0;
            for (int mm = n, t = 1; mm > 1; mm >>= 1, t <<= 1, stage++) {
                if (stage % 3 == 0) for (int i = 0; i < n; i++) x[i] = f64_reduce(x[i], m);
                const int h = mm >> 1;
                for (int i = 0, j1 = 0; i < h; i++, j1 += 2 * t) for (int j = j1; j < j1 + t; j++) {
                    const double U = x[j], V = x[j + t];
                    x[j] = U + V; x[j + t] = f64_mulmod_const(U - V, irp[h + i], irpq[h + i], m.p);
                }
            }
            const u64 ninv = powm(n, pu - 2, pu);
            for (int i = 0; i < n; i++) {
                const u64 got = mulm((u64)(centre((long long)x[i], p) + p) % pu, ninv, pu);
                if (got != (u64)a[i]) { printf("inverse transform wrong at %d\n", i); return 1; }
            }
            checked += 2 * n;
        }
    }
    printf("ok %ld primes", checked);
    for (u64 q : primes) printf(" %llu", (unsigned long long)q);
    printf("\n");
    return 0;
}
