// Lab: what v_permlane16_swap / v_permlane32_swap return on gfx950 (both results), for vdst = lane id, src = 100 + lane id.  Not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const auto a = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
    out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap r[0] (vdst)", "permlane16_swap r[1] (src) ", "permlane32_swap r[0] (vdst)", "permlane32_swap r[1] (src) "};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int l = 0; l < 64; l += 8) printf(" [%2d]=%3u", l, h[r * 64 + l]); printf("\n"); }
    return 0;
}
