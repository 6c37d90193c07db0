// ntt_butterfly_ab.hip -- the two ways to multiply by a twiddle in the lazy 64-bit butterflies of csrc/ntt_device.h, issue cost per butterfly on gfx950:
//   A  Shoup (the product's shoup_lazy4): quotient estimate from the companion word wp = floor(w 2^64 / q) -- 1 v_mad_u64_u32 + 2 v_mul_hi_u32 -- then the low
//      64 bits of a w and of h q (3 multiplies each): 9 quarter-rate multiplies, ~6 adds, two table words per twiddle
//   B  fold (q = 2^b - f, the shape of every coefficient modulus SEAL ships): the full 128-bit product a w (4 v_mad_u64_u32), then 2^b == f three times over
//      (2 + 1 + 1 multiplies, the last a 32-bit low product): 8 multiplies and ~20 shifts / masks / adds on 64- and 96-bit values, one table word per twiddle
// Both return a value congruent to a w mod q below 4 q for a < 2^59 (the lazy range of the transforms); the tool checks B against A modulo q on every lane.
// Loop: 8 independent butterflies (X, Y) -> (X + T, X - T + 4q) per thread and iteration, twiddles in registers, 4 and 8 waves per SIMD on every CU.
//   ntt_butterfly_ab            -> ns per butterfly and SIMD for both forms
#include <hip/hip_runtime.h>
#include "../crcnn_amd/csrc/ntt_device.h"
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ u64 fold_lazy(u64 a, u64 w, u32 b, u32 f)
{
    u64 lo, hi; mul64wide_mad(a, w, lo, hi);                    // a < 2^59, w < 2^55: below 2^114
    const u64 mask = ((u64)1 << b) - 1;
    // fold 1: h1 = x >> b < 2^(114 - b) <= 2^62
    const u64 h1 = (lo >> b) | (hi << (64 - b));
    const u64 p0 = (u64)(u32)h1 * f, p1 = (u64)(u32)(h1 >> 32) * f + (p0 >> 32);       // h1 f = p1 2^32 + lo32(p0) < 2^88
    const u64 x1l = (lo & mask) + (((p1 & 0xffffffffu) << 32) | (u32)p0);               // may carry into x1h
    const u64 x1h = (p1 >> 32) + (x1l < (lo & mask));
    // fold 2: h2 = x1 >> b < 2^(85 - b + 1) = 2^31 at b = 55: one word
    const u32 h2 = (u32)((x1l >> b) | (x1h << (64 - b)));
    const u64 x2 = (x1l & mask) + (u64)h2 * f;                   // < 2^b + 2^57: one more fold, of at most three bits
    return (x2 & mask) + (u64)((u32)(x2 >> b) * f);              // < 2^b + 2^29 < 2 q
}

template <int FORM>
__global__ void __launch_bounds__(1024) bf_kernel(u64 *out, const u64 *in, const u64 *W, const u64 *WP, int iters, u64 q, u32 b, u32 f)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    u64 X[8], Y[8], w[8], wp[8];
    for (int i = 0; i < 8; i++) { X[i] = in[(tid * 8 + i) % 4096] % q; Y[i] = in[(tid * 8 + i + 1) % 4096] % q; w[i] = W[(tid + 17 * i) % 4096];
        wp[i] = WP[(tid + 17 * i) % 4096]; }
    const u64 q4 = 4 * q;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 T = FORM == 0 ? shoup_lazy4(Y[i], w[i], wp[i], q) : fold_lazy(Y[i], w[i], b, f);
            const u64 nx = X[i] + T, ny = X[i] + (q4 - T);
            // (keep the values in the lazy range without changing the instruction mix much: one conditional subtraction of 8 q)
            X[i] = nx >= 8 * q ? nx - 8 * q : nx; Y[i] = ny >= 8 * q ? ny - 8 * q : ny;
        }
    }
    u64 s = 0; for (int i = 0; i < 8; i++) s += X[i] % q + Y[i] % q;
    out[tid] = s;
}

// one application of both forms to the same operands (the host compares both with a w mod q in 128-bit arithmetic)
__global__ void check_kernel(const u64 *in, const u64 *W, const u64 *WP, u64 *res, u64 q, u32 b, u32 f)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const u64 a = in[tid] & (((u64)1 << 59) - 1);
    res[2 * tid] = shoup_lazy4(a, W[tid], WP[tid], q); res[2 * tid + 1] = fold_lazy(a, W[tid], b, f);
}

int main()
{
    const u64 q = 0x7fffffff380001ULL; const u32 b = 55, f = (u32)(((u64)1 << 55) - q);
    std::vector<u64> h(4096); u64 x = 88172645463325252ULL;
    for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
    std::vector<u64> hw(4096), hwp(4096);
    for (int i = 0; i < 4096; i++) { hw[i] = h[(i * 7 + 3) % 4096] % q; hwp[i] = (u64)(((unsigned __int128)hw[i] << 64) / q); }
    u64 *in, *out, *W, *WP, *res;
    CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&W, 4096 * 8)); CK(hipMalloc(&WP, 4096 * 8)); CK(hipMalloc(&res, 8192 * 8));
        CK(hipMalloc(&out, (size_t)256 * 4 * 1024 * 8));
    CK(hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), 4096 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(WP, hwp.data(), 4096 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3(16), dim3(256), 0, 0, in, W, WP, res, q, b, f);
    std::vector<u64> hr(8192); CK(hipMemcpy(hr.data(), res, 8192 * 8, hipMemcpyDeviceToHost));
    int nbad = 0;
    for (int i = 0; i < 4096; i++) {
        const u64 a = h[i] & (((u64)1 << 59) - 1), want = (u64)(((unsigned __int128)a * hw[i]) % q);
        if (hr[2 * i] % q != want || hr[2 * i + 1] % q != want || hr[2 * i + 1] >= 4 * q || hr[2 * i] >= 4 * q) nbad++;
    }
    printf("q = 2^55 - %u: %s (4096 random operand pairs, a < 2^59)\n", f, nbad ? "FORMS DISAGREE" : "both forms agree with a w mod q, results below 4 q");
    const int iters = 4000;
    for (int threads = 256; threads <= 512; threads *= 2) {            // x 4 workgroups per CU below = 4 / 8 waves per SIMD
        for (int form = 0; form < 2; form++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&](int n) { if (form == 0) hipLaunchKernelGGL(bf_kernel<0>, dim3(256 * 4), dim3(threads), 0, 0, out, in, W, WP, n, q, b, f);
                else hipLaunchKernelGGL(bf_kernel<1>, dim3(256 * 4), dim3(threads), 0, 0, out, in, W, WP, n, q, b, f); };
            launch(10); CK(hipEventRecord(e0)); launch(iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double per_simd = 8.0 * iters * (threads * 4 / 256);      // butterflies per SIMD lane-group: waves per SIMD x 8 x iters
            printf("%d waves per SIMD  %-28s %8.3f ms  %6.2f ns per butterfly (wave) and SIMD\n", threads * 4 / 256, form == 0 ? "A Shoup (9 multiplies)" :
                "B fold (8 multiplies)", ms, ms * 1e6 / per_simd);
        }
    }
    return 0;
}
