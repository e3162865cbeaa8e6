// Lab: what bounds the re-read of the activation operand by the wide decode GEMVs?  Not product code.
// G workgroups of 8 waves all read the SAME x image (blocked layout: k-block kb = CT KiB contiguous, CT = 4 -> 512 KB at K = 4096), wave w the
// k-blocks 2 (w + 8 j) + h like gemv_mfma_kernel, with D k-blocks (4 x 1 KiB wave-loads each) in flight per wave.  If the time falls as 1 / D the
// re-read is latency-bound (deeper prefetch pays); if it does not move, it is a bandwidth / hot-spot limit.  ROT = 1: workgroup b starts its sweep
// at step b mod nsteps (tests the lock-step hot-spot idea; the product cannot do this: the order of additions is fixed).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/x_depth scripts/lab/x_depth.hip && /tmp/x_depth
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f4;

template <int D, int ROT, int PRIV>
__global__ __launch_bounds__(512) void xread(const char* x, const char* wpriv, int nsteps, int reps, float* sink) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f4 buf[D][4];
    f4 wbuf[D][PRIV ? 6 : 1];
    float acc = 0.f;
    const int rot = ROT ? (int)(blockIdx.x % nsteps) : 0;
    const char* wp = wpriv + (size_t)blockIdx.x * 6 * 128 * 1024;     // six private "row blocks" of 128 KB (the weights of gate/up with three pairs)
    auto issue = [&](int s, f4 (&b)[4], f4 (&wb)[PRIV ? 6 : 1]) {
        int j = s + rot; if (j >= nsteps) j -= nsteps;
        const int kb = 2 * (w + 8 * (j >> 1)) + (j & 1);
        const char* p = x + (size_t)kb * 4096 + lane * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) b[c] = *(const f4*)(p + c * 1024);
        if (PRIV) {
#pragma unroll
            for (int t = 0; t < 6; ++t) wb[t] = __builtin_nontemporal_load((const f4*)(wp + (size_t)t * 131072 + (size_t)kb * 1024 + lane * 16));
        }
    };
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, buf[d], wbuf[d]);
        for (int s = 0; s < nsteps; s += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc += buf[d][c][0] + buf[d][c][3];
                if (PRIV) {
#pragma unroll
                    for (int t = 0; t < 6; ++t) acc += wbuf[d][t][1];
                }
                const int nx = s + d + D;
                issue(nx < nsteps ? nx : nsteps - 1, buf[d], wbuf[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc += buf[d][c][1];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static const size_t kCopy = (size_t)256 * 6 * 131072;
template <int D, int ROT, int PRIV>
static void run(const char* x, const char* wpriv, int G, float* sink) {
    const int nsteps = 16, reps = 1;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((xread<D, ROT, PRIV>), dim3(G), dim3(512), 0, 0, x, wpriv, nsteps, reps, sink);
    CK(hipEventRecord(a));
    const int N = 20;
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL((xread<D, ROT, PRIV>), dim3(G), dim3(512), 0, 0, x, wpriv + (size_t)(i % 4) * kCopy, nsteps, reps, sink);   // HBM-cold weights
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / N;
    const double xb = 512.0 * 1024 * (1 + 1.0 * (D - 0) / 16);     // (the prologue + clamped re-reads add D / 16)
    printf("G=%3d depth=%2d rot=%d weights=%d: %7.2f us per launch   x %.1f GB/s per CU%s\n", G, D, ROT, PRIV, us, 512.0 * 1024 / (us * 1e-6) / 1e9,
           PRIV ? "  (+ 768 KB private weights per workgroup, HBM)" : "");
    (void)xb;
}

int main() {
    char *x, *wpriv; float* sink;
    CK(hipMalloc(&x, 1 << 20)); CK(hipMemset(x, 0, 1 << 20));
    CK(hipMalloc(&wpriv, 4 * kCopy + (1 << 20))); CK(hipMemset(wpriv, 0, 4 * kCopy + (1 << 20)));
    CK(hipMalloc(&sink, 64));
    for (int G : {230, 256, 512}) {
        run<1, 0, 0>(x, wpriv, G, sink); run<2, 0, 0>(x, wpriv, G, sink); run<4, 0, 0>(x, wpriv, G, sink); run<8, 0, 0>(x, wpriv, G, sink);
        run<2, 1, 0>(x, wpriv, G, sink); run<8, 1, 0>(x, wpriv, G, sink);
    }
    for (int G : {230}) {
        run<1, 0, 1>(x, wpriv, G, sink); run<2, 0, 1>(x, wpriv, G, sink); run<4, 0, 1>(x, wpriv, G, sink);
        run<2, 1, 1>(x, wpriv, G, sink);
    }
    return 0;
}
