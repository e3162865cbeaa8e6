// Lab: does a kernel that requests the NEXT kernel's first weight bytes into its XCD's L2 (while it runs its own epilogue) shorten the next kernel's ramp?
// A chain of streaming kernels (256 workgroups x 8 waves, each workgroup streams its own contiguous share of a 61.8 MB matrix -- the average
// producer matrix of the 7B decoder -- with a two-buffer pipeline of 4 x 1 KiB wave-loads, then an LDS reduce + a 1 KiB store like a GEMV epilogue),
// HBM-cold (16 matrices in rotation), captured in a hipGraph.  PF = 1: before its epilogue every wave also requests the first PFB KiB that the wave of
// the same (workgroup, wave) index will read in the next kernel, default cache policy, never consumed.  Not product code.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tail_prefetch scripts/lab/tail_prefetch.hip && /tmp/tail_prefetch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u4;

template <int PF, int PFB, int EARLY>
__global__ __launch_bounds__(512) void chain_kernel(const char* cur, const char* nxt, size_t bytes, float* out) {
    __shared__ u4 red[8][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_wg = bytes / gridDim.x;
    const char* p = cur + (size_t)blockIdx.x * per_wg + lane * 16;
    const char* q = nxt + (size_t)blockIdx.x * per_wg + lane * 16;
    const int nblk = (int)(per_wg / 1024);
    const int nb = (nblk - w + 7) / 8;                  // blocks of this wave: w, w + 8, ...
    u4 acc = {0, 0, 0, 0};
    u4 a[4], b[4];
    auto load = [&](u4 (&v)[4], int i0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const u4*)(p + (size_t)(w + 8 * min(i0 + u, nb - 1)) * 1024));
    };
    auto use = [&](u4 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
    };
    // never read: the wave ends with these in flight.  hipcc does not know that the asm's result arrives later, so the registers are kept
    // reserved up to the wave's end by `keep` (otherwise they are reused and the late data lands in live values -- the first version faulted).
    u4 pf[PFB];
    auto prefetch = [&]() {
        if (PF) {
#pragma unroll
            for (int u = 0; u < PFB; ++u)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pf[u]) : "v"(q + (size_t)(w + 8 * u) * 1024) : "memory");
        }
    };
    auto keep = [&]() {
        if (PF) {
#pragma unroll
            for (int u = 0; u < PFB; ++u) asm volatile("" :: "v"(pf[u]));
        }
    };
    load(a, 0);
    int i = 0;
    for (; i + 8 < nb; i += 8) {
        load(b, i + 4);
        __builtin_amdgcn_sched_barrier(0);
        use(a);
        __builtin_amdgcn_sched_barrier(0);
        load(a, i + 8);
        __builtin_amdgcn_sched_barrier(0);
        use(b);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (EARLY) prefetch();                               // behind the last requests of the own stream
    load(b, i + 4);
    use(a);
    use(b);
    if (!EARLY) prefetch();                              // after the own stream has landed, in front of the epilogue
    red[w][lane] = acc;
    __syncthreads();
    if (w != 0) { keep(); return; }
    u4 t = red[0][lane];
#pragma unroll
    for (int k = 1; k < 8; ++k) t ^= red[k][lane];
    out[(size_t)blockIdx.x * 64 + lane] = (float)(t[0] ^ t[1] ^ t[2] ^ t[3]);
    keep();
}

template <int PF, int PFB, int EARLY>
static void run(const std::vector<char*>& mats, size_t bytes, float* out, const char* name) {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 64;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL((chain_kernel<PF, PFB, EARLY>), dim3(256), dim3(512), 0, s, mats[i % mats.size()], mats[(i + 1) % mats.size()], bytes, out);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r > 0) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s %6.2f us per kernel (best %6.2f)   %.2f TB/s\n", name, sum / 5 * 1e3 / n, best * 1e3 / n, bytes / (sum / 5 * 1e-3 / n) / 1e12);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
    for (size_t mb : {34, 62, 100}) {
        const size_t bytes = (mb << 20) / (256 * 8192) * (256 * 8192);
        std::vector<char*> mats(16);
        for (auto& m : mats) { CK(hipMalloc(&m, bytes)); CK(hipMemset(m, 1, bytes)); }
        float* out; CK(hipMalloc(&out, 256 * 64 * 4));
        printf("== %zu MB per kernel ==\n", mb);
        run<0, 4, 0>(mats, bytes, out, "no prefetch");
        run<1, 4, 0>(mats, bytes, out, "4 KiB per wave in front of the epilogue");
        run<1, 4, 1>(mats, bytes, out, "4 KiB per wave behind the last own requests");
        run<1, 8, 1>(mats, bytes, out, "8 KiB per wave behind the last own requests");
        run<0, 4, 0>(mats, bytes, out, "no prefetch (again)");
        for (auto& m : mats) CK(hipFree(m));
        CK(hipFree(out));
    }
    return 0;
}
