// Which int8 MFMA shape holds the higher clock on random operands?  (MI355X_MICROARCH.md, 'DVFS give-back' item 7.)  A bare loop: per wave a 32 x 32 output tile,
// 13 accumulators per sub-tile, 49 limb-pair MFMAs per K = 64 -- as v_mfma_i32_32x32x32_i8 (2 x 49 per K = 64) or v_mfma_i32_16x16x64_i8 (4 sub-tiles x 49).
// Operands in registers, one wave per SIMD, every CU busy.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip ; run: ./mfma_shape [zeros]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k32(const v4i *in, int *out, int iters)
{
    v4i a[7], w[7];
    for (int l = 0; l < 7; l++) { a[l] = in[(threadIdx.x * 14 + l) % 4096]; w[l] = in[(threadIdx.x * 14 + 7 + l) % 4096]; }
    v16i acc[13];
    for (int d = 0; d < 13; d++) for (int e = 0; e < 16; e++) acc[d][e] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int l = 0; l < 7; l++)
#pragma unroll
                for (int m = 0; m < 7; m++) acc[l + m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[l], w[m], acc[l + m], 0, 0, 0);
    }
    int s = 0;
    for (int d = 0; d < 13; d++) for (int e = 0; e < 16; e++) s += acc[d][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k16(const v4i *in, int *out, int iters)
{
    v4i a[2][7], w[2][7];
    for (int l = 0; l < 7; l++) for (int h = 0; h < 2; h++) { a[h][l] = in[(threadIdx.x * 28 + h * 14 + l) % 4096]; w[h][l] = in[(threadIdx.x * 28 + h * 14 + 7 + l) % 4096]; }
    v4i acc[4][13];
    for (int t = 0; t < 4; t++) for (int d = 0; d < 13; d++) for (int e = 0; e < 4; e++) acc[t][d][e] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rs = 0; rs < 2; rs++)
#pragma unroll
            for (int l = 0; l < 7; l++)
#pragma unroll
                for (int cs = 0; cs < 2; cs++)
#pragma unroll
                    for (int m = 0; m < 7; m++) acc[rs * 2 + cs][l + m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[rs][l], w[cs][m], acc[rs * 2 + cs][l + m], 0, 0, 0);
    }
    int s = 0;
    for (int t = 0; t < 4; t++) for (int d = 0; d < 13; d++) for (int e = 0; e < 4; e++) s += acc[t][d][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
    std::vector<int> h(4096 * 4);
    srand(1); for (auto &v : h) v = zeros ? 0 : (int)(((unsigned)rand() << 16) ^ (unsigned)rand());
    v4i *din; int *dout; hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 1024 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 1024;
    for (int which = 0; which < 2; which++)
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, din, dout, iters); else hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, din, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ops = (double)grid * 4 * iters * 98 * 65536.0;       // per wave and iteration: 98 x 32x32x32 = 196 x 16x16x64
            printf("%s %s: %.2f ms  %.2f Pop/s\n", which ? "16x16x64" : "32x32x32", zeros ? "zeros" : "random", ms, ops / ms / 1e12);
        }
    return 0;
}
