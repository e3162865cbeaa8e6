// Lab: do a wave's VALU instructions overlap with MFMAs (its own, or another wave's on the same SIMD)?  Not product code.
// Each wave loops over [NM x v_mfma_f32_32x32x16_bf16 on two independent accumulators] + [NV independent v_fma_f32 (+ NE v_exp_f32)].
// Reported: cycles per iteration per SIMD for MFMA only, VALU only, both, at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NM, int NV, int NE>
__global__ void k(float* out, int iters, long long* cyc) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x16 c0 = {}, c1 = {};
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; m += 2) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) v[n & 15] = __builtin_fmaf(v[n & 15], 1.0001f, 0.5f);
#pragma unroll
        for (int n = 0; n < NE; ++n) v[n & 15] = __builtin_amdgcn_exp2f(v[n & 15]);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i] + c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int NE>
static void run(const char* name, int waves_per_simd) {
    float* out; long long* cyc; CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int threads = 256 * waves_per_simd;          // one workgroup per CU, 4 SIMDs
    hipLaunchKernelGGL((k<NM, NV, NE>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<NM, NV, NE>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-28s waves/SIMD %d: %8.1f ns per iteration per SIMD (%.0f clock64 ticks per iteration of one wave)\n", name, waves_per_simd, ms * 1e6 / iters, (double)c / iters);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    for (int w : {1, 2, 4}) {
        run<8, 0, 0>("8 MFMA", w);
        run<0, 64, 0>("64 FMA", w);
        run<0, 0, 16>("16 EXP", w);
        run<0, 64, 16>("64 FMA + 16 EXP", w);
        run<8, 64, 16>("8 MFMA + 64 FMA + 16 EXP", w);
    }
    return 0;
}
