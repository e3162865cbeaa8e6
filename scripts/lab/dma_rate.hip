// Lab: sustained L2->LDS (global_load_lds_dwordx4) and L2->VGPR rates per CU by access pattern.  Not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// Each workgroup (NW waves) sweeps a [256 rows x K] 16-bit panel in 64-column steps, like a GEMM operand panel:
// per K step 256 rows x 128 B = 32 KB -> NW waves x (32/NW) DMA instructions of 1 KB.
// MODE 0: 8 rows x 128 B per instruction, chunk order XOR-swizzled inside the line (gemm.hip pattern)
// MODE 1: 8 rows x 128 B, linear chunk order
// MODE 2: 16 rows x 64 B per instruction (half lines)
// MODE 3: panel stored tile-contiguous: every instruction reads 1 KB contiguous
// MODE 4: as MODE 1 but plain global_load_dwordx4 into VGPRs (no LDS)
// MODE 5: 4 rows x 256 B per instruction (BK = 128 style)
template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void dma_kernel(const char* base, size_t panel_bytes, int npanels, int ld_bytes, int ksteps, int reps, int depth, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* panel = base + (size_t)(blockIdx.x % npanels) * panel_bytes;
    constexpr int J = 32 / NW;                 // instructions per wave per K step
    const char* src[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int inst = j * NW + w;           // 0..31
        if (MODE == 0 || MODE == 1 || MODE == 4) {
            const int row = inst * 8 + (lane >> 3); int c = lane & 7;
            if (MODE == 0) c ^= (row >> 1) & 7;
            src[j] = panel + (size_t)row * ld_bytes + c * 16;
        } else if (MODE == 2) {
            const int row = (inst >> 1) * 16 + (lane >> 2); const int c = (inst & 1) * 4 + (lane & 3);
            src[j] = panel + (size_t)row * ld_bytes + c * 16;
        } else if (MODE == 3) {
            src[j] = panel + (size_t)inst * 1024 + lane * 16;      // + kstep * 32 KB
        } else {
            const int row = inst * 4 + (lane >> 4); const int c = lane & 15;     // 128 rows x 256 B
            src[j] = panel + (size_t)row * ld_bytes + c * 16;
        }
    }
    const size_t kstride = (MODE == 3) ? 32768 : (MODE == 5 ? 256 : 128);
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        for (int k = 0; k < ksteps; ++k) {
            char* dst = smem + (k & 1) * 32768;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const char* s = src[j] + (size_t)k * kstride;
                if (MODE == 4) {
                    float4 v = *(const float4*)s;
                    acc += v.x;
                } else {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                                     (__attribute__((address_space(3))) void*)(dst + (j * NW + w) * 1024), 16, 0, 0);
                }
            }
            if (MODE != 4) {
                if (depth == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (J >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int NW>
void run(const char* name, const char* buf, size_t panel_bytes, int npanels, int ld_bytes, int ksteps, int depth, float* sink, int grid, int reps = 20) {
    CK(hipFuncSetAttribute((const void*)dma_kernel<MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((dma_kernel<MODE, NW>), dim3(grid), dim3(NW * 64), 65536, 0, buf, panel_bytes, npanels, ld_bytes, ksteps, reps, depth, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    const double bytes = (double)grid * reps * ksteps * 32768.0;
    printf("%-28s NW=%d grid=%d ld=%d depth=%d: %7.3f ms  %7.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU@2.4\n", name, NW, grid, ld_bytes, depth, best,
           bytes / best / 1e9, bytes / best / 1e6 / 256, bytes / best / 1e6 / 256 / 2.4);
}

int main() {
    const int K = 8192;                      // panel = 256 rows x 8192 cols x 2 B = 4 MB
    const size_t panel_bytes = 256 * (size_t)K * 2;
    const int npanels = 8;                    // 32 MB total: L2-resident-ish per XCD (each XCD touches 1 panel: block%8 == xcd)
    char* buf; CK(hipMalloc(&buf, panel_bytes * npanels + 65536));
    CK(hipMemset(buf, 1, panel_bytes * npanels));
    float* sink; CK(hipMalloc(&sink, 4));
    const int ld = K * 2, ks = K / 64;
    for (int grid : {256, 512}) {
        for (int depth : {0, 1}) {
            run<0, 8>("8x128B swizzled", buf, panel_bytes, npanels, ld, ks, depth, sink, grid);
            run<1, 8>("8x128B linear", buf, panel_bytes, npanels, ld, ks, depth, sink, grid);
            run<2, 8>("16x64B", buf, panel_bytes, npanels, ld, ks, depth, sink, grid);
            run<3, 8>("1KB contiguous", buf, panel_bytes, npanels, ld, ks, depth, sink, grid);
            run<5, 8>("4x256B", buf, panel_bytes, npanels, ld, ks / 2, depth, sink, grid);
        }
        run<4, 8>("8x128B linear -> VGPR", buf, panel_bytes, npanels, ld, ks, 1, sink, grid);
        run<0, 4>("8x128B swizzled", buf, panel_bytes, npanels, ld, ks, 1, sink, grid);
        run<3, 4>("1KB contiguous", buf, panel_bytes, npanels, ld, ks, 1, sink, grid);
        run<0, 2>("8x128B swizzled", buf, panel_bytes, npanels, ld, ks, 1, sink, grid);
        run<0, 1>("8x128B swizzled", buf, panel_bytes, npanels, ld, ks, 1, sink, grid);
    }
    // short rows (ViT K=1024): ld = 2048 B
    run<0, 8>("8x128B swizzled, ld 2KB", buf, 256 * 2048, 64, 2048, 16, 1, sink, 256, 200);
    run<1, 8>("8x128B linear, ld 2KB", buf, 256 * 2048, 64, 2048, 16, 1, sink, 256, 200);
    return 0;
}
