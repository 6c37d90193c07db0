// Does a wave's own vector work hide behind its own MFMAs?  Per iteration 49 v_mfma_i32_16x16x64_i8 (13 accumulators, as mfma_conv1_kernel issues them) and 245
// dependent-free v_mad_u32_u24: (a) all MFMAs, then all VALU; (b) braided 1 : 5 (pinned with sched_barrier); (c) MFMAs only; (d) VALU only.  1 / 2 / 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(768) k(int *out, int iters, int seed)
{
    v4i acc[13], a[7], b[7];
    for (int d = 0; d < 13; d++) acc[d] = v4i{0, 0, 0, 0};
    for (int l = 0; l < 7; l++) { a[l] = v4i{seed + l, seed * 3 + l, seed * 5 + l, seed * 7 + l}; b[l] = v4i{seed - l, seed * 11 + l, seed ^ l, seed + 9 * l}; }
    unsigned x[5] = {(unsigned)threadIdx.x, (unsigned)seed, 3u, 5u, 7u};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int l = 0; l < 7; l++)
#pragma unroll
                for (int m = 0; m < 7; m++) acc[l + m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[l], acc[l + m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int e = 0; e < 49; e++)
#pragma unroll
                for (int v = 0; v < 5; v++) x[v] = (x[v] ^ x[(v + 1) % 5]) + x[(v + 2) % 5];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int l = 0; l < 7; l++)
#pragma unroll
                for (int m = 0; m < 7; m++) {
                    acc[l + m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[l], acc[l + m], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < 5; v++) x[v] = (x[v] ^ x[(v + 1) % 5]) + x[(v + 2) % 5];
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    int s = 0;
    for (int d = 0; d < 13; d++) s += acc[d][0] + acc[d][1] + acc[d][2] + acc[d][3];
    for (int v = 0; v < 5; v++) s += (int)x[v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> static float run(int waves_per_simd, int iters, int *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 10, 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    int *d; hipMalloc(&d, 256 * 768 * 4);
    const int iters = 20000;
    const char *nm[4] = {"49 MFMA then 245 VALU", "braided 1 MFMA : 5 VALU", "49 MFMA only", "245 VALU only"};
    for (int w = 1; w <= 3; w++) {
        float t[4] = {run<0>(w, iters, d), run<1>(w, iters, d), run<2>(w, iters, d), run<3>(w, iters, d)};
        for (int m = 0; m < 4; m++) printf("%d wave(s) per SIMD  %-26s %8.3f ms  %7.1f ns per iteration and wave set\n", w, nm[m], t[m], t[m] * 1e6 / iters);
    }
    return 0;
}
