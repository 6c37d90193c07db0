// lds_dma_unaligned.hip -- does global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) accept global addresses that are only 4-byte aligned, and what does it cost?
// (the K-flattened limb tensor of kernels_mfma.hip keeps channels unpadded: a pixel's 20 channel bytes start at multiples of 20)
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_dma_unaligned.hip -o tools/lds_dma_unaligned ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef signed char i8;
__global__ void __launch_bounds__(256) probe(const i8 *src, i8 *out, int stride, int iters, int check)
{
    extern __shared__ __attribute__((aligned(16))) i8 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i8 *base = src + (size_t)blockIdx.x * 256 * stride + (size_t)(wave * 64 + lane) * stride;
    for (int it = 0; it < iters; it++) {
        // every wave fills its own 1 KiB of LDS: lane l's 16 bytes land at 16 l (the DMA writes lane-linear)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (size_t)it * 4096 * stride), (__attribute__((address_space(3))) void *)(lds + wave * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
    __syncthreads();
    if (check) *reinterpret_cast<uint4 *>(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16) = *reinterpret_cast<const uint4 *>(lds + threadIdx.x * 16);
}
int main()
{
    const int blocks = 2048, iters = 64;
    const size_t bytes = (size_t)blocks * 256 * 32 + (size_t)iters * 4096 * 32 + 64;
    std::vector<i8> h(bytes); for (size_t i = 0; i < bytes; i++) h[i] = (i8)((i * 131 + (i >> 9)) & 0xff);
    i8 *d, *o; hipMalloc(&d, bytes); hipMalloc(&o, (size_t)blocks * 256 * 16); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    std::vector<i8> r((size_t)blocks * 256 * 16);
    for (int stride : {16, 20, 24, 28, 32, 12}) {
        hipMemset(o, 0, r.size());
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 4096, 0, d, o, stride, 1, 1);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("stride %d: launch failed: %s\n", stride, hipGetErrorString(e)); return 1; }
        hipMemcpy(r.data(), o, r.size(), hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t t = 0; t < (size_t)blocks * 256; t++) for (int j = 0; j < 16; j++) if (r[t * 16 + j] != h[t * stride + j]) bad++;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 4096, 0, d, o, stride, iters, 0);
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 4096, 0, d, o, stride, iters, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("stride %2d B (lane addresses %s16-byte aligned): %zu wrong bytes; %.3f ms per launch, %.2f TB/s of DMA payload\n", stride, stride % 16 ? "not " : "", bad, ms / 5,
               (double)blocks * 256 * 16 * iters / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
