// issue cost of the fp64 vector instructions f64mod.h is built from (gfx950): one wave set of 4 / 8 waves per SIMD, 8 independent chains per lane
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(1024) k(double *out, int iters, double seed)
{
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = seed + i + threadIdx.x * 1e-3;
    const double c = 1.0000001, M = 6755399441055744.0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) x[i] = x[i] + c;
                if (OP == 1) x[i] = x[i] * c;
                if (OP == 2) x[i] = __builtin_fma(x[i], c, c);
                if (OP == 3) x[i] = __builtin_rint(x[i]) + 0.0 * c == 7.0 ? 1.0 : __builtin_rint(x[i] * 1.0);      // (kept simple below)
                if (OP == 4) x[i] = (x[i] + M) - M;
            }
    }
    double s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
__global__ void __launch_bounds__(1024) krint(double *out, int iters, double seed)
{
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = seed + i + threadIdx.x * 1e-3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) { double y; asm volatile("v_rndne_f64 %0, %1" : "=v"(y) : "v"(x[i])); x[i] = y; }
    }
    double s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K> static float run(K kern, int threads, int iters, double *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, 10, 1.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double *d; hipMalloc(&d, 256 * 1024 * 8);
    const int iters = 20000;
    for (int threads = 256; threads <= 1024; threads *= 2) {
        const double per = 64.0 * iters * (threads / 256);            // instructions per SIMD
        float t[5] = {run(k<0>, threads, iters, d), run(k<1>, threads, iters, d), run(k<2>, threads, iters, d), run(krint<0>, threads, iters, d), run(k<4>, threads, iters, d)};
        const char *nm[5] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_rndne_f64", "(x + 1.5 2^52) - 1.5 2^52 (two v_add_f64)"};
        for (int m = 0; m < 5; m++) printf("%d waves per SIMD  %-44s %8.3f ms  %6.2f ns per instruction (or pair) and SIMD\n", threads / 256, nm[m], t[m], t[m] * 1e6 / per);
    }
    return 0;
}
