// f64_row_timeline.hip -- where does a row transform over an fp64 prime (csrc/ntt_f64.h) spend its time?  A sandbox for the pass structure of the kernels that
// hold a row in registers (relin_digits*_f64_kernel, sq64_inv_kernel, relin_inv_crt_kernel: 512 threads, 16 points per thread, two workgroups per CU).
//
//   f64_row_timeline [n] [rows] [reps]
//
// One workgroup per source row: `reps` transforms of that row (as the digit kernel cuts 8 transforms from one row), each = fill the LDS image, the radix-8
// passes,
// drain + 16-byte stores.  Variants of the pass (same arithmetic, same results -- the tool checks every variant against variant 0 bit for bit):
//   0  the product's pass (ntt_pass_f64): per group twiddle loads, LDS reads, butterflies, LDS writes; groups one after the other
//   1  both groups of a thread loaded before the first is computed (software pipelining across the groups of a pass)
//   2  variant 1 + twiddles through the scalar cache in the passes whose twiddle block index is wave-uniform (stride 2^ls >= 64 groups)
//   3  (inverse only) variant 2 + lazy reduction: a pass reduces the two outputs that grew (8 B and 3.5 p) instead of all eight inputs
//   4 / 5 / 6  the product's pass with scalar twiddles / scalar twiddles + lazy reduction / lazy reduction only (no second group in flight: fewer registers)
//   7  the product's pass (lazy reduction when inverse) with the 7 twiddles of a (pass, block) stored as 8 consecutive doubles: one address and four 16-byte "
    "loads
//      instead of seven addresses and seven 8-byte loads; scalar-cache loads where the block index is wave-uniform
// (all variants drain the image with the product's batched f64_drain since round 5's second half)
//   Throughput: ns per row transform over `rows` rows (HIP events).  Timeline: s_memtime stamps of wave 0 of every workgroup around the phases of the LAST rep,
//   median over workgroups, in shader cycles (a diagnostic build of the same code: the stamps cost a few percent, the throughput numbers come from the build
//   without).
#include "../crcnn_amd/csrc/ntt_f64.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static constexpr int RB = 3, NPT = 16, NSTAMP = 16;
struct Grouped { const double *g[4]; };           // variant 7: the 7 twiddles of a (pass, block) as 8 consecutive doubles

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

// twiddle fetch: per lane from global (vector loads), or -- when the block index is the same for the whole wave -- through the scalar cache
template <bool INV, bool SCALAR>
__device__ __forceinline__ void fetch_tw(double (&tw)[7], const double *W, int tabidx, unsigned blk)
{
    if (SCALAR) blk = (unsigned)__builtin_amdgcn_readfirstlane((int)blk);
    if (INV) load_tw_inv<RB>(tw, W, tabidx, (int)blk); else load_tw_fwd<RB>(tw, W, tabidx, (int)blk);
}

// inverse stages with lazy reduction: inputs |x| <= 1.75 p, outputs [14 p, 3.5 p, 1.75 p, 1.75 p, m, m, m, m] (m = 0.875 p) with the first two reduced
__device__ __forceinline__ void inv_stages_lazy(double (&v)[8], const double (&tw)[7], const F64Mod md)
{
    inv_stages_f64<RB>(v, tw, md);
    v[0] = f64_reduce(v[0], md); v[1] = f64_reduce(v[1], md);
}

template <bool INV, int VAR, bool STAMP>
// st: this workgroup's stamp slots in global memory
__device__ __forceinline__ void pass(double *sm, const double *W, int n, int ls, int tabidx, const F64Mod md, bool reduce_in, unsigned long long *st, int &si,
                                     const double *G)
{
    const unsigned groups = (unsigned)n >> RB;
    const bool uniform = (VAR == 2 || VAR == 3) && ls >= 6;
    if (VAR == 0 || VAR >= 4) {
        const bool scal = (VAR == 4 || VAR == 5) && ls >= 6, lazy = INV && (VAR == 5 || VAR == 6 || VAR == 7) && !reduce_in;
        for (unsigned g = threadIdx.x; g < groups; g += blockDim.x) {
            const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
            const int a0 = swz<RB>((int)((blk << (ls + RB)) + l));
            double tw[7], v[8];
            if (VAR == 7) {            // one address, four 16-byte loads (scalar-cache loads where the block index is wave-uniform)
                const unsigned bb = ls >= 6 ? (unsigned)__builtin_amdgcn_readfirstlane((int)blk) : blk;
                const d2 *gp = reinterpret_cast<const d2 *>(G + (size_t)bb * 8);
                const d2 t0 = gp[0], t1 = gp[1], t2 = gp[2], t3 = gp[3];
                tw[0] = t0.x; tw[1] = t0.y; tw[2] = t1.x; tw[3] = t1.y; tw[4] = t2.x; tw[5] = t2.y; tw[6] = t3.x;
            } else if (scal) fetch_tw<INV, true>(tw, W, tabidx, blk); else fetch_tw<INV, false>(tw, W, tabidx, blk);
#pragma unroll
            for (int c = 0; c < 8; c++) { v[c] = sm[a0 ^ swz<RB>(c << ls)]; if (INV && !lazy) v[c] = f64_reduce(v[c], md); }
            if (INV) { if (lazy) inv_stages_lazy(v, tw, md); else inv_stages_f64<RB>(v, tw, md); } else fwd_stages_f64<RB>(v, tw, md);
#pragma unroll
            for (int c = 0; c < 8; c++) sm[a0 ^ swz<RB>(c << ls)] = v[c];
        }
    } else {
        // two groups per thread (n = 16 points per thread x blockDim): everything the pass reads is requested before the first butterfly
        const unsigned g0 = threadIdx.x, g1 = threadIdx.x + blockDim.x;
        const unsigned blk0 = g0 >> ls, l0 = g0 & ((1u << ls) - 1), blk1 = g1 >> ls, l1 = g1 & ((1u << ls) - 1);
        const int a0 = swz<RB>((int)((blk0 << (ls + RB)) + l0)), a1 = swz<RB>((int)((blk1 << (ls + RB)) + l1));
        double tw0[7], tw1[7], v0[8], v1[8];
        if (uniform) { fetch_tw<INV, true>(tw0, W, tabidx, blk0); fetch_tw<INV, true>(tw1, W, tabidx, blk1); }
        else { fetch_tw<INV, false>(tw0, W, tabidx, blk0); fetch_tw<INV, false>(tw1, W, tabidx, blk1); }
#pragma unroll
        for (int c = 0; c < 8; c++) v0[c] = sm[a0 ^ swz<RB>(c << ls)];
#pragma unroll
        for (int c = 0; c < 8; c++) v1[c] = sm[a1 ^ swz<RB>(c << ls)];
        if (STAMP && threadIdx.x == 0) st[si++] = now();
        const bool lazy = INV && VAR == 3 && !reduce_in;
        if (INV && !lazy) {
#pragma unroll
            for (int c = 0; c < 8; c++) v0[c] = f64_reduce(v0[c], md);
        }
        if (INV) { if (lazy) inv_stages_lazy(v0, tw0, md); else inv_stages_f64<RB>(v0, tw0, md); } else fwd_stages_f64<RB>(v0, tw0, md);
#pragma unroll
        for (int c = 0; c < 8; c++) sm[a0 ^ swz<RB>(c << ls)] = v0[c];
        if (INV && !lazy) {
#pragma unroll
            for (int c = 0; c < 8; c++) v1[c] = f64_reduce(v1[c], md);
        }
        if (INV) { if (lazy) inv_stages_lazy(v1, tw1, md); else inv_stages_f64<RB>(v1, tw1, md); } else fwd_stages_f64<RB>(v1, tw1, md);
#pragma unroll
        for (int c = 0; c < 8; c++) sm[a1 ^ swz<RB>(c << ls)] = v1[c];
    }
    if (STAMP && threadIdx.x == 0) st[si++] = now();
    __syncthreads();
    if (STAMP && threadIdx.x == 0) st[si++] = now();
}

template <bool INV, int VAR, bool STAMP>
__global__ void __launch_bounds__(512, 4) rows_kernel(const double *src, double *dst, const double *Wt, F64Mod md, int n, int logn, int reps,
                                                      unsigned long long *stamps, Grouped gr)
{
    extern __shared__ double smd[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const double *row = src + (size_t)blockIdx.x * n;
    unsigned long long *st = stamps + (size_t)blockIdx.x * NSTAMP; int si = 0;
    double r[NPT];
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) { const int s = 2 * (tid + u * nt); const d2 v = *reinterpret_cast<const d2 *>(row + s); r[2 * u] = v.x;
        r[2 * u + 1] = v.y; }
    for (int rep = 0; rep < reps; rep++) {
        const bool last = STAMP && rep == reps - 1;
        si = 0;
        if (last && tid == 0) st[si++] = now();
        const double sc = (double)(rep + 1);
#pragma unroll
        for (int u = 0; u < NPT / 2; u++) {
            const int s = 2 * (tid + u * nt);
            // (the held row gives every rep different operands, as the digit kernel's shifts do)
            const d2 v = f64_stage_in<INV, RB>(d2{r[2 * u] * sc, r[2 * u + 1] * sc}, Wt, n, logn, s, md);
            sm_store_pair<RB>(smd, s, v.x, v.y);
        }
        __syncthreads();
        if (last && tid == 0) st[si++] = now();
        const int full = logn / RB;                                  // (n = 8192: 13 = 4 x 3 + the fused gap-1 stage)
        if (!INV) {
            int lt = logn - 1;
            for (int p = 0; p < full; p++, lt -= RB) {
                if (last) pass<false, VAR, true>(smd, Wt, n, lt - RB + 1, n >> (lt + 1), md, false, st, si, gr.g[p]);
                else pass<false, VAR, false>(smd, Wt, n, lt - RB + 1, n >> (lt + 1), md, false, st, si, gr.g[p]);
            }
        } else {
            int lt = 1;
            for (int p = 0; p < full; p++, lt += RB) {
                if (last) pass<true, VAR, true>(smd, Wt, n, lt, n >> (lt + 1), md, false, st, si, gr.g[p]);
                else pass<true, VAR, false>(smd, Wt, n, lt, n >> (lt + 1), md, false, st, si, gr.g[p]);
            }
        }
        double *out = dst + ((size_t)blockIdx.x * reps + rep) * n;
        f64_drain<INV, RB, NPT / 4>(smd, Wt, n, logn, md, [&](int s, d2 v) { *reinterpret_cast<d2 *>(out + s) = d2{f64_reduce(v.x, md), f64_reduce(v.y,
            md)}; });
        if (last && tid == 0) st[si++] = now();
        __syncthreads();
        if (last && tid == 0) st[si++] = now();
    }
    if (STAMP && tid == 0) for (int i = si; i < NSTAMP; i++) st[i] = 0;
}

// ---- variant 8: ONE workgroup barrier per transform -------------------------------------------------------------------------------------------------------
// n = 8192 on 512 threads = 8 waves x 1024 points.  The three stages with gaps 1024 / 2048 / 4096 are the only ones that cross the 1024-point blocks;
    a thread that
// owns the PAIR (l, l +
    1) at the eight block offsets c 1024 can run them in registers straight from (forward) / to (inverse) the 16-byte memory accesses -- the fill
// and the first pass, or the last pass and the drain,
    become one step without an LDS round trip -- and the other ten stages stay inside a block.  Give block w to wave w
// and those passes need no workgroup barrier at all: a wave's LDS operations execute in order, so its own writes are visible to its later reads. Forward:
// registers -> cross pass -> image | BARRIER | three wave-local passes -> block-local drain (fused last stage) -> memory | BARRIER (the image is reused).
// Inverse: block-local fill (fused first stage) -> three wave-local passes | BARRIER | cross pass from the image -> registers -> memory | BARRIER.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool INV>
__device__ __forceinline__ void local_pass(double *sm, const double *W, int ls, int tabidx, const F64Mod md, bool lazy)
{
    const unsigned w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (unsigned u = 0; u < 2; u++) {
        const unsigned g = (w << 7) + lane + 64 * u;                 // the 128 groups of block w (in every pass with a gap below 1024: block = g >> 7)
        const unsigned blk = g >> ls, l = g & ((1u << ls) - 1);
        const int a0 = swz<RB>((int)((blk << (ls + RB)) + l));
        double tw[7], v[8];
        if (INV) load_tw_inv<RB>(tw, W, tabidx, (int)blk); else load_tw_fwd<RB>(tw, W, tabidx, (int)blk);
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = sm[a0 ^ swz<RB>(c << ls)];
        if (INV) { inv_stages_f64<RB>(v, tw, md); if (lazy) { v[0] = f64_reduce(v[0], md); v[1] = f64_reduce(v[1], md); } } else fwd_stages_f64<RB>(v, tw, md);
#pragma unroll
        for (int c = 0; c < 8; c++) sm[a0 ^ swz<RB>(c << ls)] = v[c];
    }
    wave_sync();
}
template <bool INV>
__global__ void __launch_bounds__(512, 4) rows_kernel_wave(const double *src, double *dst, const double *Wt, F64Mod md, int n, int logn, int reps)
{
    extern __shared__ double smd[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const double *row = src + (size_t)blockIdx.x * n;
    // the held row: forward in the cross pass's layout (pairs 2 tid + 1024 c), inverse block-local (pairs 1024 w + 2 lane + 128 u)
    double r[NPT];
#pragma unroll
    for (int u = 0; u < NPT / 2; u++) {
        const int s = INV ? 1024 * w + 2 * lane + 128 * u : 2 * tid + 1024 * u;
        const d2 v = *reinterpret_cast<const d2 *>(row + s); r[2 * u] = v.x; r[2 * u + 1] = v.y;
    }
    for (int rep = 0; rep < reps; rep++) {
        const double sc = (double)(rep + 1);
        double *out = dst + ((size_t)blockIdx.x * reps + rep) * n;
        if (!INV) {
            double tw[7];
            load_tw_fwd<RB>(tw, Wt, 1, 0);                            // gaps 4096, 2048, 1024: block index 0 for every group -- seven wave-uniform twiddles
#pragma unroll
            for (int e = 0; e < 2; e++) {
                double v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = r[2 * c + e] * sc;
                fwd_stages_f64<RB>(v, tw, md);
#pragma unroll
                // (back into the pair slots: stored as pairs below; r is re-made from the source next rep)
                for (int c = 0; c < 8; c++) r[2 * c + e] = v[c];
            }
#pragma unroll
            for (int c = 0; c < 8; c++) sm_store_pair<RB>(smd, 2 * tid + 1024 * c, r[2 * c], r[2 * c + 1]);
            __syncthreads();
            int lt = logn - 1 - RB;
            for (int p = 1; p < logn / RB; p++, lt -= RB) local_pass<false>(smd, Wt, lt - RB + 1, n >> (lt + 1), md, false);
            // block-local drain with the fused gap-1 stage: wave w owns points [1024 w, 1024 w + 1024)
            d2 v[8]; double t1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int s = 1024 * w + 2 * lane + 128 * u; v[u] = sm_load_pair<RB>(smd, s); t1[u] = Wt[(n >> 1) + (s >> 1)]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s = 1024 * w + 2 * lane + 128 * u;
                const double T = f64_mulmod(t1[u], v[u].y, md);
                *reinterpret_cast<d2 *>(out + s) = d2{f64_reduce(v[u].x + T, md), f64_reduce(v[u].x - T, md)};
            }
            __syncthreads();
            // (the sandbox re-reads its source row; the digit kernel re-cuts its held words)
#pragma unroll
            for (int u = 0; u < NPT / 2; u++) { const d2 x = *reinterpret_cast<const d2 *>(row + 2 * tid + 1024 * u); r[2 * u] = x.x; r[2 * u + 1] = x.y; }
        } else {
            // block-local fill with the fused gap-1 stage
            double t1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t1[u] = Wt[(n >> 1) + ((1024 * w + 2 * lane + 128 * u) >> 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int s = 1024 * w + 2 * lane + 128 * u;
                const double U = f64_reduce(r[2 * u] * sc, md), V = f64_reduce(r[2 * u + 1] * sc, md);
                sm_store_pair<RB>(smd, s, U + V, f64_mulmod(t1[u], U - V, md));
            }
            wave_sync();
            int lt = 1;
            for (int p = 0; p + 1 < logn / RB; p++, lt += RB) local_pass<true>(smd, Wt, lt, n >> (lt + 1), md, true);
            __syncthreads();
            // cross pass (gaps 1024, 2048, 4096) from the image to registers to memory
            double tw[7];
            load_tw_inv<RB>(tw, Wt, n >> (lt + 1), 0);
            d2 x[8];
#pragma unroll
            for (int c = 0; c < 8; c++) x[c] = sm_load_pair<RB>(smd, 2 * tid + 1024 * c);
#pragma unroll
            for (int e = 0; e < 2; e++) {
                double v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = e ? x[c].y : x[c].x;
                inv_stages_f64<RB>(v, tw, md);
#pragma unroll
                for (int c = 0; c < 8; c++) { if (e) x[c].y = f64_reduce(v[c], md); else x[c].x = f64_reduce(v[c], md); }
            }
#pragma unroll
            for (int c = 0; c < 8; c++) *reinterpret_cast<d2 *>(out + 2 * tid + 1024 * c) = x[c];
            __syncthreads();
        }
    }
}

template <bool INV, int VAR>
static void run(const char *name, const double *src, double *dst, const double *W, F64Mod md, int n, int logn, int rows, int reps, unsigned long long *d_st,
                std::vector<double> *ref, Grouped gr = Grouped{})
{
    const size_t lds = (size_t)n * 8;
    auto k0 = rows_kernel<INV, VAR, false>; auto k1 = rows_kernel<INV, VAR, true>;
    CK(hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k0, dim3(rows), dim3(n / NPT), lds, 0, src, dst, W, md, n, logn, reps, d_st, gr);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k0, dim3(rows), dim3(n / NPT), lds, 0, src, dst, W, md, n, logn, reps, d_st, gr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // results of the first 4 rows against variant 0
    std::vector<double> h((size_t)4 * reps * n);
    CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
    bool same = true;
    if (ref->empty()) *ref = h; else same = h == *ref;
    hipLaunchKernelGGL(k1, dim3(rows), dim3(n / NPT), lds, 0, src, dst, W, md, n, logn, reps, d_st, gr);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)rows * NSTAMP);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    printf("%-58s %7.2f ns per row transform   %s\n", name, ms / 3 * 1e6 / ((double)rows * reps), same ? "results = variant 0" : "RESULTS DIFFER");
    // phase medians
    int ns = 0; while (ns < NSTAMP && st[ns]) ns++;
    printf("    cycles of wave 0, last transform, median over workgroups:");
    for (int i = 1; i < ns; i++) {
        std::vector<long long> d(rows);
        for (int b = 0; b < rows; b++) d[b] = (long long)(st[(size_t)b * NSTAMP + i] - st[(size_t)b * NSTAMP + i - 1]);
        std::nth_element(d.begin(), d.begin() + rows / 2, d.end());
        printf(" %lld", d[rows / 2]);
    }
    { std::vector<long long> d(rows); for (int b = 0; b < rows; b++) d[b] = (long long)(st[(size_t)b * NSTAMP + ns - 1] - st[(size_t)b * NSTAMP]);
        std::nth_element(d.begin(), d.begin() + rows / 2, d.end()); printf("   | whole %lld\n", d[rows / 2]); }
}

template <bool INV>
static void run_wave(const char *name, const double *src, double *dst, const double *W, F64Mod md, int n, int logn, int rows, int reps,
    const std::vector<double> &ref)
{
    const size_t lds = (size_t)n * 8;
    auto k0 = rows_kernel_wave<INV>;
    CK(hipFuncSetAttribute((const void *)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k0, dim3(rows), dim3(n / NPT), lds, 0, src, dst, W, md, n, logn, reps);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k0, dim3(rows), dim3(n / NPT), lds, 0, src, dst, W, md, n, logn, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> h((size_t)4 * reps * n);
    CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%-58s %7.2f ns per row transform   %s\n", name, ms / 3 * 1e6 / ((double)rows * reps), h == ref ? "results = variant 0" : "RESULTS DIFFER");
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192, rows = argc > 2 ? atoi(argv[2]) : 4096, reps = argc > 3 ? atoi(argv[3]) : 4;
    int logn = 0; while ((1 << logn) < n) logn++;
    if ((1 << logn) != n || logn % RB != 1 || n / NPT > 512) { fprintf(stderr, "n must be 2^(3m+1) with n / 16 <= 512 threads (8192)\n"); return 1; }
    const double p = 140737488273409.0;           // just below 2^47 (timing only: primality is irrelevant here, exactness of the arithmetic is not)
    F64Mod md{p, 1.0 / p};
    std::vector<double> hs((size_t)rows * n), hw(n);
    unsigned long long x = 88172645463325252ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto &v : hs) v = (double)(long long)(rnd() % 65536);
    for (auto &v : hw) v = (double)((long long)(rnd() % (unsigned long long)p) - (long long)(p / 2));
    double *src, *dst, *W; unsigned long long *d_st;
    CK(hipMalloc(&src, hs.size() * 8)); CK(hipMalloc(&dst, hs.size() * 8 * reps)); CK(hipMalloc(&W, hw.size() * 8));
        CK(hipMalloc(&d_st, (size_t)rows * NSTAMP * 8));
    CK(hipMemcpy(src, hs.data(), hs.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 8, hipMemcpyHostToDevice));
    printf("n = %d, %d workgroups (one source row each), %d transforms per workgroup, 512 threads x 16 points, 64 KiB LDS image (two workgroups per CU)\n", n,
        rows, reps);
    printf("stamp order: fill+barrier | per pass: [loads issued ->] butterflies+stores -> barrier | drain+stores issued | barrier\n");
    // variant 7: per pass and block the 7 twiddles of a radix-8 group as 8 consecutive doubles (the same values the product's table gives, regrouped)
    auto grouped = [&](bool inv) {
        Grouped gr{};
        const int full = logn / RB;
        int lt = inv ? 1 : logn - 1;
        for (int p = 0; p < full; p++, lt += inv ? RB : -RB) {
            const int tab = n >> (lt + 1), blocks = inv ? n >> (lt + RB) : tab;
            std::vector<double> g((size_t)blocks * 8, 0.0);
            for (int blk = 0; blk < blocks; blk++)
                for (int st = 0; st < RB; st++) {
                    if (!inv) for (int j = 0; j < (1 << st); j++) g[(size_t)blk * 8 + (1 << st) - 1 + j] = hw[(tab << st) + (blk << st) + j];
                    else for (int j = 0; j < (1 << (RB - 1 - st)); j++) g[(size_t)blk * 8 + (1 << RB) - (1 << (RB - st)) + j] = hw[(tab >> st) + (blk << (RB -
                        1 - st)) + j];
                }
            double *d; CK(hipMalloc(&d, g.size() * 8)); CK(hipMemcpy(d, g.data(), g.size() * 8, hipMemcpyHostToDevice));
            gr.g[p] = d;
        }
        return gr;
    };
    std::vector<double> ref;
    run<false, 0>("forward, product pass", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<false, 1>("forward, both groups loaded first", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<false, 2>("forward, + scalar twiddles in wave-uniform passes", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<false, 4>("forward, product pass + scalar twiddles", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<false, 7>("forward, grouped twiddle table (4 x 16-byte loads)", src, dst, W, md, n, logn, rows, reps, d_st, &ref, grouped(false));
    run_wave<false>("forward, ONE workgroup barrier (wave-local passes)", src, dst, W, md, n, logn, rows, reps, ref);
    ref.clear();
    run<true, 0>("inverse, product pass", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 1>("inverse, both groups loaded first", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 2>("inverse, + scalar twiddles in wave-uniform passes", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 3>("inverse, + lazy reduction (2 of 8 values per pass)", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 4>("inverse, product pass + scalar twiddles", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 5>("inverse, product pass + scalar twiddles + lazy reduction", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 6>("inverse, product pass + lazy reduction", src, dst, W, md, n, logn, rows, reps, d_st, &ref);
    run<true, 7>("inverse, lazy reduction + grouped twiddle table", src, dst, W, md, n, logn, rows, reps, d_st, &ref, grouped(true));
    run_wave<true>("inverse, ONE workgroup barrier (wave-local passes, lazy)", src, dst, W, md, n, logn, rows, reps, ref);
    return 0;
}
