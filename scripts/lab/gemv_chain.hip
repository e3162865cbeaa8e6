// Lab: the decode GEMV kernels (through the C ABI) vs a bare streaming-read kernel of the same bytes, as plain launches and as a hipGraph.
// Not product code.  Separates the fixed per-launch cost of the GEMV (prologue, LDS reduce, store) from the dispatch cost.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/pgv.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define PK(x) do { int e = (x); if (e) { printf("%s: %s\n", #x, pgv_last_error()); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// internal launcher of libpgv (exported C++ symbol): reaches the folded-RMSNorm producer / consumer modes the public pgv_gemv does not
struct GemvNorm {
    const float* ssq_in = nullptr; int nparts_in = 0; int hidden = 1; float eps = 0.f;
    const float* gamma = nullptr; void* xg = nullptr; float* ssq_out = nullptr;
    float* amax_val = nullptr; int* amax_idx = nullptr;
    int ssq_ts = 0, amax_ts = 0;
};
int pgv_launch_gemv(pgv_ctx* ctx, int dtype, int mode, const void* W, const void* x, int ldx, void* out, int ldo, int N, int K, int B, hipStream_t s,
                    const float* wscale, const GemvNorm* norm);

template <int EPI>
__global__ __launch_bounds__(1024) void stream_kernel(const char* base, size_t bytes, const float* side, float* out) {
    __shared__ u32x4 red[16][64];
    const size_t per_wg = bytes / gridDim.x;
    const char* p = base + (size_t)blockIdx.x * per_wg;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t nblk = per_wg / 1024;
    u32x4 acc = {0, 0, 0, 0};
    float s = 0.f;
    if (EPI >= 2) s = side[threadIdx.x] + side[threadIdx.x + 1024] + side[threadIdx.x + 2048];      // L2-hit side loads, like the ssq partials
    size_t j = w;
    for (; j + 16 * 3 < nblk; j += 16 * 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)(p + (j + 16 * u) * 1024 + lane * 16));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
    for (; j < nblk; j += 16) acc ^= __builtin_nontemporal_load((const u32x4*)(p + j * 1024 + lane * 16));
    if (EPI == 0) { if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345677u) out[0] = 1; return; }
    red[w][lane] = acc;
    __syncthreads();
    if (w != 0) return;
    u32x4 t = red[0][lane];
#pragma unroll
    for (int i = 1; i < 16; ++i) t ^= red[i][lane];
    out[(size_t)blockIdx.x * 64 + lane] = (float)(t[0] ^ t[1] ^ t[2] ^ t[3]) + s;
}

template <typename F>
static void time_chain(const char* name, hipStream_t s, int n, double bytes, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int m = 0; m < n; ++m) launch(m);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int m = 0; m < n; ++m) launch(m);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float bestg = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < bestg) bestg = ms;
    }
    printf("%-34s plain %6.2f us  graph %6.2f us  (%.1f MB: %.2f / %.2f TB/s)\n", name, best * 1e3 / n, bestg * 1e3 / n, bytes / 1e6, bytes / (best * 1e-3 / n) / 1e12,
           bytes / (bestg * 1e-3 / n) / 1e12);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    pgv_ctx* ctx; PK(pgv_ctx_create(0, &ctx));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NM = 24;
    const size_t slot = (size_t)32016 * 4096 * 2;            // largest matrix (lm_head)
    char* W; CK(hipMalloc(&W, slot * NM)); CK(hipMemset(W, 0x11, slot * NM));     // small finite 16-bit values
    char* x; CK(hipMalloc(&x, 64 * 11008 * 2)); CK(hipMemset(x, 0x11, 64 * 11008 * 2));
    char* out; CK(hipMalloc(&out, 64 * 32016 * 4)); CK(hipMemset(out, 0, 64 * 32016 * 4));
    float* side; CK(hipMalloc(&side, 4096 * 4)); CK(hipMemset(side, 0, 4096 * 4));
    struct { const char* name; int mode, N, K; } cases[] = {{"qkv      (mode 0, 12288x4096)", 0, 12288, 4096}, {"gate/up  (mode 2, 22016x4096)", 2, 22016, 4096},
                                                            {"down     (mode 1, 4096x11008)", 1, 4096, 11008}, {"o_proj   (mode 1, 4096x4096)", 1, 4096, 4096},
                                                            {"lm_head  (mode 3, 32003x4096)", 3, 32003, 4096}};
    for (auto& c : cases) {
        const double bytes = 2.0 * ((c.N + 15) / 16 * 16) * c.K;
        for (int B : {8, 16, 32, 64}) {
            char nm[96]; snprintf(nm, sizeof nm, "%s B=%d", c.name, B);
            time_chain(nm, s, NM, bytes, [&](int m) {
                PK(pgv_gemv(ctx, PGV_BF16, c.mode, W + slot * m, x, c.K, out, c.mode == 2 ? c.N / 2 : c.N, c.N, c.K, B, s));
            });
        }
        time_chain("  bare stream, no epilogue", s, NM, bytes, [&](int m) { hipLaunchKernelGGL((stream_kernel<0>), dim3(256), dim3(1024), 0, s, W + slot * m, (size_t)bytes, side, (float*)out); });
        time_chain("  + LDS reduce + store", s, NM, bytes, [&](int m) { hipLaunchKernelGGL((stream_kernel<1>), dim3(256), dim3(1024), 0, s, W + slot * m, (size_t)bytes, side, (float*)out); });
        time_chain("  + side loads", s, NM, bytes, [&](int m) { hipLaunchKernelGGL((stream_kernel<2>), dim3(256), dim3(1024), 0, s, W + slot * m, (size_t)bytes, side, (float*)out); });
    }
    // the folded-norm modes and a whole decoder layer (without attention), 8 sequences
    const int H = 4096, I = 11008, B = 8;
    float* resid; CK(hipMalloc(&resid, 16 * H * 4)); CK(hipMemset(resid, 0, 16 * H * 4));
    float* gamma; CK(hipMalloc(&gamma, H * 4)); CK(hipMemset(gamma, 0, H * 4));
    char* xg; CK(hipMalloc(&xg, 16 * H * 2)); CK(hipMemset(xg, 0x11, 16 * H * 2));
    float* ssq; CK(hipMalloc(&ssq, (H / 16) * 16 * 4)); CK(hipMemset(ssq, 0, (H / 16) * 16 * 4));
    char* act; CK(hipMalloc(&act, 16 * I * 2)); CK(hipMemset(act, 0x11, 16 * I * 2));
    char* qkv; CK(hipMalloc(&qkv, 16 * 3 * H * 2));
    GemvNorm cons; cons.ssq_in = ssq; cons.nparts_in = H / 16; cons.hidden = H; cons.eps = 1e-5f;
    GemvNorm prod; prod.gamma = gamma; prod.xg = xg; prod.ssq_out = ssq; prod.hidden = H;
    const size_t o_qkv = 0, o_o = (size_t)3 * H * H * 2, o_gu = o_o + (size_t)H * H * 2, o_dn = o_gu + (size_t)2 * I * H * 2;     // 404.75 MB per layer < 2 slots
    auto L = [&](int m) { return W + (size_t)m * 2 * slot; };
    const int NL = NM / 2;
    time_chain("qkv consumer (folded norm)", s, NL, 2.0 * 3 * H * H, [&](int m) { PK(pgv_launch_gemv(ctx, PGV_BF16, 0, L(m) + o_qkv, xg, H, qkv, 3 * H, 3 * H, H, B, s, nullptr, &cons)); });
    time_chain("o_proj producer (16 waves)", s, NL, 2.0 * H * H, [&](int m) { PK(pgv_launch_gemv(ctx, PGV_BF16, 5, L(m) + o_o, qkv, H, resid, H, H, H, B, s, nullptr, &prod)); });
    time_chain("gate/up consumer", s, NL, 4.0 * I * H, [&](int m) { PK(pgv_launch_gemv(ctx, PGV_BF16, 2, L(m) + o_gu, xg, H, act, I, 2 * I, H, B, s, nullptr, &cons)); });
    time_chain("down producer (16 waves)", s, NL, 2.0 * I * H, [&](int m) { PK(pgv_launch_gemv(ctx, PGV_BF16, 5, L(m) + o_dn, act, I, resid, H, H, I, B, s, nullptr, &prod)); });
    time_chain("down as plain resid (8 waves)", s, NL, 2.0 * I * H, [&](int m) { PK(pgv_launch_gemv(ctx, PGV_BF16, 1, L(m) + o_dn, act, I, resid, H, H, I, B, s, nullptr, nullptr)); });
    time_chain("whole layer (4 GEMVs)", s, NL, 2.0 * (4.0 * H * H + 3.0 * I * H), [&](int m) {
        PK(pgv_launch_gemv(ctx, PGV_BF16, 0, L(m) + o_qkv, xg, H, qkv, 3 * H, 3 * H, H, B, s, nullptr, &cons));
        PK(pgv_launch_gemv(ctx, PGV_BF16, 5, L(m) + o_o, qkv, H, resid, H, H, H, B, s, nullptr, &prod));
        PK(pgv_launch_gemv(ctx, PGV_BF16, 2, L(m) + o_gu, xg, H, act, I, 2 * I, H, B, s, nullptr, &cons));
        PK(pgv_launch_gemv(ctx, PGV_BF16, 5, L(m) + o_dn, act, I, resid, H, H, I, B, s, nullptr, &prod));
    });
    return 0;
}
