// Lab: one wave per SIMD issuing v_mfma_f32_32x32x16_bf16 back to back with filler instructions in the gaps.
// How much MFMA throughput survives (a) an LDS-DMA every N MFMAs, (b) ds_read_b128 every M MFMAs, (c) both?  Not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// DMA_EVERY: one global_load_lds_dwordx4 per DMA_EVERY MFMAs (0 = none); RD_PER4: ds_read_b128 per 4 MFMAs (0..8)
template <int NW, int DMA_EVERY, int RD_PER4, int NACC>
__global__ __launch_bounds__(NW * 64, 1) void k(const char* base, size_t panel_bytes, int npanels, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* panel = base + (size_t)(blockIdx.x % npanels) * panel_bytes;
    const int row = w * 8 + (lane >> 3); const int c = (lane & 7) ^ ((row >> 1) & 7);
    const char* src = panel + (size_t)row * 16384 + c * 16;
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8_t fa[8], fb[2];
    for (int i = 0; i < 8; ++i) fa[i] = *(const bf16x8_t*)(smem + lane * 16 + i * 1024);
    fb[0] = fa[0]; fb[1] = fa[1];
    const char* rd = smem + 32768 + (lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) << 4);
    int kq = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {          // 16 groups of 4 MFMAs = 64 MFMAs per iteration (one K-step of a 128x128 wave tile)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int idx = (g * 4 + m) % NACC;
                acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(g * 4 + m) & 7], fb[m & 1], acc[idx], 0, 0, 0);
                if constexpr (DMA_EVERY > 0) {
                    if ((g * 4 + m) % DMA_EVERY == DMA_EVERY - 1) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(kq & 63) * 128),
                                                         (__attribute__((address_space(3))) void*)(smem + 65536 + (w * 16 + (kq & 15)) * 1024), 16, 0, 0);
                        ++kq;
                    }
                }
            }
            if constexpr (RD_PER4 > 0) {
#pragma unroll
                for (int r = 0; r < RD_PER4; ++r) fa[(g * RD_PER4 + r) & 7] = *(const bf16x8_t*)(rd + ((g * RD_PER4 + r) & 7) * 4096);
            }
            if constexpr (DMA_EVERY > 0) { if (g == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) sink[0] = s;
}

template <int NW, int DMA_EVERY, int RD_PER4, int NACC>
void run(const char* name, const char* buf, size_t panel_bytes, int npanels, float* sink) {
    const int iters = 400, grid = 256;
    CK(hipFuncSetAttribute((const void*)k<NW, DMA_EVERY, RD_PER4, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<NW, DMA_EVERY, RD_PER4, NACC>), dim3(grid), dim3(NW * 64), 163840, 0, buf, panel_bytes, npanels, iters, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double mfma = (double)grid * NW * iters * 64;
    const double tf = mfma * 32768.0 / best / 1e9;
    printf("%-34s NW=%d dma/%d rd%d nacc=%d: %7.3f ms  %7.1f TF/s  %5.1f ns/MFMA/SIMD\n", name, NW, DMA_EVERY, RD_PER4, NACC, best, tf,
           best * 1e6 / (iters * 64.0 * (NW > 4 ? NW / 4 : 1)));
}

int main() {
    const size_t panel_bytes = 256 * (size_t)16384;
    const int npanels = 8;
    char* buf; CK(hipMalloc(&buf, panel_bytes * npanels + 65536));
    CK(hipMemset(buf, 0x3c, panel_bytes * npanels));
    float* sink; CK(hipMalloc(&sink, 4));
    run<4, 0, 0, 16>("mfma only", buf, panel_bytes, npanels, sink);
    run<4, 0, 0, 4>("mfma only, 4 accumulators", buf, panel_bytes, npanels, sink);
    run<4, 0, 0, 2>("mfma only, 2 accumulators", buf, panel_bytes, npanels, sink);
    run<4, 0, 2, 16>("2 ds_read_b128 / 4 mfma", buf, panel_bytes, npanels, sink);
    run<4, 0, 4, 16>("4 ds_read_b128 / 4 mfma", buf, panel_bytes, npanels, sink);
    run<4, 8, 0, 16>("dma / 8 mfma", buf, panel_bytes, npanels, sink);
    run<4, 4, 0, 16>("dma / 4 mfma (GEMM 128x128 rate)", buf, panel_bytes, npanels, sink);
    run<4, 2, 0, 16>("dma / 2 mfma", buf, panel_bytes, npanels, sink);
    run<4, 4, 2, 16>("dma/4 + 2 reads/4 (full G mix)", buf, panel_bytes, npanels, sink);
    run<4, 4, 3, 16>("dma/4 + 3 reads/4", buf, panel_bytes, npanels, sink);
    run<8, 0, 0, 8>("8 waves mfma only", buf, panel_bytes, npanels, sink);
    run<8, 4, 2, 8>("8 waves dma/4 + 2 reads/4", buf, panel_bytes, npanels, sink);
    return 0;
}
