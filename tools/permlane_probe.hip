// what v_permlane16_swap_b32 / v_permlane32_swap_b32 move (gfx950): lanes print (a, b) = (lane, 100 + lane) after each swap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out) {
    const unsigned l = threadIdx.x;
    v2u r = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    out[l] = r.x; out[64 + l] = r.y;
    v2u s = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    out[128 + l] = s.x; out[192 + l] = s.y;
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[4] = {"permlane16_swap .x (vdst)", "permlane16_swap .y (src)", "permlane32_swap .x (vdst)", "permlane32_swap .y (src)"};
    for (int v = 0; v < 4; v++) { printf("%s:", nm[v]); for (int l = 0; l < 64; l += 16) printf("  lanes %2d-%2d = %u..%u", l, l + 15, h[v * 64 + l], h[v * 64 + l + 15]); printf("\n"); }
    return 0;
}
