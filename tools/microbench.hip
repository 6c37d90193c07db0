// microbench.hip -- VALU integer throughput probes on gfx950 (which primitive should the modular MAC be built from?)
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64; typedef uint32_t u32;
#define ITERS 4096
#define NACC 8

template <int OP> __global__ void __launch_bounds__(256) probe(u64 *out, u32 a0, u32 b0)
{
    u64 acc[NACC]; u32 a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    double dacc[NACC]; double da = (double)a * 1e-9, db = (double)b * 1e-9;
    for (int i = 0; i < NACC; i++) { acc[i] = i + threadIdx.x; dacc[i] = i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (OP == 0) acc[i] += (u64)(a + i) * (u32)(b ^ (u32)acc[i]);                 // v_mad_u64_u32 (dependent on acc lo only through xor)
            if (OP == 1) acc[i] = (u64)((u32)acc[i] * (a + i) + b);                       // v_mul_lo_u32 + add  (v_mad_u32_u24? no: full 32-bit)
            if (OP == 2) acc[i] = (u64)(__umulhi((u32)acc[i], a + i) + b);                // v_mul_hi_u32
            if (OP == 3) acc[i] = (u64)(__umul24((u32)acc[i], a + i) + b);                // v_mul_u32_u24 / v_mad_u32_u24
            if (OP == 4) dacc[i] = __builtin_fma(dacc[i], da, db);                        // v_fma_f64
            if (OP == 5) acc[i] += ((u64)a << 32 | b) ^ (u64)i;                           // 64-bit add (v_add_co + v_addc)
            if (OP == 6) acc[i] = (u64)((u32)acc[i] + a + i);                             // v_add_u32 / v_add3
            if (OP == 7) { u64 x = acc[i]; acc[i] = __umul64hi(x, ((u64)a << 32) | b) + x * (((u64)b << 32) | a); }   // full 64x64 hi + lo
            if (OP == 8) { u64 x = acc[i]; acc[i] = x * ((((u64)a << 32) | b) + i); }    // 64x64 low
            if (OP == 9) dacc[i] = __builtin_rint(dacc[i] * da) + db;                     // v_mul_f64 + v_rndne_f64 + v_add_f64
            if (OP == 10) {                                                               // fp64 modular multiplication by a constant (6 flops): exact for p < 2^50
                const double w = 123456789012345.0 + i, wq = w / 140737488355333.0, P = 140737488355333.0, y = dacc[i];
                const double h = w * y, l = __builtin_fma(w, y, -h), c = __builtin_rint(y * wq);
                dacc[i] = __builtin_fma(-c, P, h) + l;
            }
            if (OP == 11) {                                                               // the row NTT's lazy 64-bit Shoup product (3-product quotient estimate)
                const u64 w = (((u64)a << 23) | b) + i, wp = ~w, q = 0x7fffffff380001ULL, y = acc[i];
                const u32 y0 = (u32)y, y1 = (u32)(y >> 32), p0 = (u32)wp, p1 = (u32)(wp >> 32);
                const u64 hq = (u64)y1 * p1 + __umulhi(y1, p0) + __umulhi(y0, p1);
                acc[i] = y * w - hq * q;
            }
        }
    }
    u64 s = 0; for (int i = 0; i < NACC; i++) s += acc[i] + (u64)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> double run(const char *name, double ops_per_iter)
{
    const int blocks = 256 * 16, threads = 256;
    u64 *out; hipMalloc(&out, sizeof(u64) * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<OP><<<blocks, threads>>>(out, 12345, 6789); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) probe<OP><<<blocks, threads>>>(out, 12345 + r, 6789);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double total = (double)blocks * threads * ITERS * NACC * ops_per_iter;
    double rate = total / (ms * 1e-3) / 1e12;
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
    double cyc = 2.4e9 * 1024 / (rate * 1e12 / 64);
    printf("%-28s %8.3f ms  %8.2f Tops/s  ~%5.1f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, rate, cyc);
    hipFree(out); return rate;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("mad_u64_u32", 1); run<1>("mul_lo_u32(+add)", 1); run<2>("mul_hi_u32(+add)", 1); run<3>("mul_u32_u24(+add)", 1);
    run<4>("fma_f64", 1); run<5>("add_u64", 1); run<6>("add_u32", 1); run<7>("umul64hi + mul64lo", 1); run<8>("mul64 lo", 1);
    run<9>("mul+rndne+add f64", 3); run<10>("fp64 modmul (6 flops) as 1", 1); run<11>("u64 lazy Shoup modmul as 1", 1);
    return 0;
}
