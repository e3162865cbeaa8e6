// MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue), 16-bit in, fp32 accumulate.
//
// This is the workhorse of the path: every Linear of the CLIP tower (HF:clip/modeling_clip.py
// CLIPAttention q/k/v/out_proj :290-293, CLIPMLP fc1/fc2 :343-344, patch conv :151-157), the
// mm_projector (video_chatgpt/model/video_chatgpt.py:51-55,105) and the LLaMA prefill projections
// (HF:llama/modeling_llama.py LlamaAttention/LlamaMLP) is `x @ W.T (+ b)` with x and W both
// K-contiguous, so one NT kernel serves all of them.
//
// Design (CDNA4): 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), each wave a
// 128x64 sub-tile = 4x2 v_mfma_f32_32x32x16 accumulators), BK=64, one workgroup per CU.
// Global->LDS staging uses global_load_lds_dwordx4 (no VGPR round trip).  The LDS image of a tile is
// [rows][64] 16-bit = 128 B rows; a DMA'd wave-instruction fills 8 rows linearly, so the bank-conflict
// swizzle is applied on the SOURCE chunk index (chunk ^= (row>>1)&7) and again on the ds_read_b128
// address: the 16 lanes a ds_read_b128 services together then hit 16 distinct 16-B slots.
// Operands are fed swapped (MFMA A-operand = W fragment, B-operand = A fragment) so each lane ends up
// with 4 consecutive output columns of one row -> 8-byte (16-bit out) / 16-byte (fp32 residual) stores.
// Workgroups are renumbered so that the tiles an XCD works on concurrently share A row-panels (per-XCD L2).
#include "pgv_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A + W
constexpr int LDS_BYTES = 2 * STAGE_BYTES;     // double buffered: 128 KiB

struct KArgs {
    const char* A; const char* W; const float* bias; char* C;
    int lda, ldw, ldc;      // elements
    int M, N, K;            // logical sizes (N = output columns before SWIGLU halving)
    int ntm, ntn;
    int wblk;               // W is in the fragment-blocked layout (weights.h) instead of row-major
};

__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_256(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    // ---- XCD-aware tile id (block b runs on XCD b%8; give each XCD a contiguous run of tiles) ----
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int tm = t / p.ntn, tn = t - tm * p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave w, instruction j covers tile rows (j*8+w)*8 .. +8, lane -> (row, slot) ----
    const int srow = lane >> 3;                                       // row within the 8-row group
    const int sw_src = ((lane >> 4) + 4 * (w & 1)) & 7;                // (row>>1)&7 of the tile row
    const int chunk = (lane & 7) ^ sw_src;                            // source 16-B chunk landing in slot lane&7
    const char* ga[4];
    const char* gw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int row = (j * 8 + w) * 8 + srow;
        int ra = min(m0 + row, p.M - 1);
        int rw = min(n0 + row, p.N - 1);
        ga[j] = p.A + ((size_t)ra * p.lda + chunk * 8) * 2;
        if (p.wblk)   // block (rw/16, k/32): chunk c of the 64-wide K tile lives in block c/4 at slot (c%4)*16 + rw%16
            gw[j] = p.W + (((size_t)(rw >> 4) * (p.K >> 5) + (chunk >> 2)) * 512 + (((chunk & 3) << 4) + (rw & 15)) * 8) * 2;
        else
            gw[j] = p.W + ((size_t)rw * p.ldw + chunk * 8) * 2;
    }
    const size_t wstep = p.wblk ? 2048 : (size_t)BK * 2;     // bytes per K tile on the W side
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
        const size_t koff = (size_t)kt * BK * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[j] + koff),
                                             (__attribute__((address_space(3))) void*)(base + (j * 8 + w) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[j] + (size_t)kt * wstep),
                                             (__attribute__((address_space(3))) void*)(base + TILE_BYTES + (j * 8 + w) * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes) ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (lane >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const int a_row_off = (wr * 128 + l31) * 128;                     // + i*32*128
    const int w_row_off = TILE_BYTES + (wc * 64 + l31) * 128;         // + j*32*128

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const char* sb = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            typename T::v8 af[4], wf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const typename T::v8*)(sb + a_row_off + i * 4096 + koffs[kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *(const typename T::v8*)(sb + w_row_off + j * 4096 + koffs[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma32(wf[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of tile kt+1 has landed ...
        __syncthreads();                                   // ... for every wave, and reads of `buf` are done before it is restaged
    }

    // ---- epilogue: lane holds, for row m = ..+l31, columns n = ..+8g+4hi+{0..3} (g = 0..3) of each 32x32 block ----
    const int mbase = m0 + wr * 128 + l31;
    const int nbase = n0 + wc * 64 + 4 * hi;
    if constexpr (EPI == PGV_EPI_SWIGLU) {
        // W rows interleaved per 64: [32 gate | 32 up]; output column = (n0 + wc*64)/2 + 8g + 4hi + e
        const int obase = (n0 + wc * 64) / 2 + 4 * hi;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + i * 32;
            if (m < p.M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(acc[i][0][g * 4 + e]) * acc[i][1][g * 4 + e];
                    const int n = obase + 8 * g;
                    if (n < p.N / 2) *(u32x2_t*)(p.C + ((size_t)m * p.ldc + n) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else {
        f32x4_t bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + j * 32 + 8 * g;
                if (p.bias != nullptr && n < p.N) bv[j][g] = *(const f32x4_t*)(p.bias + n);
                else bv[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + i * 32;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nbase + j * 32 + 8 * g;
                        if (n >= p.N) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e] + bv[j][g][e];
                        if constexpr (EPI == PGV_EPI_BIAS_QGELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
                        }
                        if constexpr (EPI == PGV_EPI_BIAS_GELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                        }
                        if constexpr (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID) {
                            f32x4_t* rp = (f32x4_t*)(p.C + ((size_t)m * p.ldc + n) * 4);
                            f32x4_t o = *rp;
                            o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
                            *rp = o;
                        } else if constexpr (EPI == PGV_EPI_F32) {
                            *(f32x4_t*)(p.C + ((size_t)m * p.ldc + n) * 4) = f32x4_t{v[0], v[1], v[2], v[3]};
                        } else {
                            *(u32x2_t*)(p.C + ((size_t)m * p.ldc + n) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
                        }
                    }
            }
        }
    }
}

template <typename T, int EPI>
int launch(const KArgs& k, int grid, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_256<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) { pgv_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e)); return PGV_EHIP; }
        configured = true;
    }
    hipLaunchKernelGGL((gemm_nt_256<T, EPI>), dim3(grid), dim3(512), LDS_BYTES, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pgv_set_error("gemm launch: %s", hipGetErrorString(e)); return PGV_EHIP; }
    return PGV_OK;
}

template <typename T>
int dispatch_epi(int epi, const KArgs& k, int grid, hipStream_t s) {
    switch (epi) {
        case PGV_EPI_NONE:       // bias pointer is null -> same code path as BIAS
        case PGV_EPI_BIAS:       return launch<T, PGV_EPI_BIAS>(k, grid, s);
        case PGV_EPI_BIAS_QGELU: return launch<T, PGV_EPI_BIAS_QGELU>(k, grid, s);
        case PGV_EPI_BIAS_GELU:  return launch<T, PGV_EPI_BIAS_GELU>(k, grid, s);
        case PGV_EPI_RESID:
        case PGV_EPI_BIAS_RESID: return launch<T, PGV_EPI_BIAS_RESID>(k, grid, s);
        case PGV_EPI_SWIGLU:     return launch<T, PGV_EPI_SWIGLU>(k, grid, s);
        case PGV_EPI_F32:        return launch<T, PGV_EPI_F32>(k, grid, s);
    }
    pgv_set_error("unknown epilogue %d", epi);
    return PGV_EINVAL;
}

}  // namespace

int pgv_launch_gemm(pgv_ctx* ctx, int dtype, const GemmArgs& a, hipStream_t s) {
    PGV_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    PGV_CHECK(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    PGV_CHECK(a.N % 8 == 0, "gemm: N=%d must be a multiple of 8", a.N);
    PGV_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements");
    PGV_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: A/W must be 16-byte aligned");
    const bool out32 = (a.epi == PGV_EPI_RESID || a.epi == PGV_EPI_BIAS_RESID || a.epi == PGV_EPI_F32);
    PGV_CHECK(((uintptr_t)a.C & (out32 ? 15 : 7)) == 0 && a.ldc % 4 == 0, "gemm: C misaligned (ptr/ldc)");
    KArgs k;
    k.A = (const char*)a.A; k.W = (const char*)a.W; k.bias = a.bias; k.C = (char*)a.C;
    k.lda = a.lda; k.ldw = a.ldw; k.ldc = a.ldc; k.M = a.M; k.N = a.N; k.K = a.K;
    k.ntm = (a.M + BM - 1) / BM; k.ntn = (a.N + BN - 1) / BN;
    k.wblk = a.w_blocked ? 1 : 0;
    PGV_CHECK(!a.w_blocked || a.N % 16 == 0, "gemm: blocked weights need N %% 16 == 0");
    const int grid = k.ntm * k.ntn;
    pgv_prof_begin(ctx, 0, s);
    int rc;
    if (dtype == PGV_F16) rc = dispatch_epi<TF16>(a.epi, k, grid, s);
    else if (dtype == PGV_BF16) rc = dispatch_epi<TBF16>(a.epi, k, grid, s);
    else { pgv_set_error("gemm: unsupported dtype %d", dtype); rc = PGV_EINVAL; }
    const double out_cols = (a.epi == PGV_EPI_SWIGLU) ? a.N / 2.0 : (double)a.N;
    pgv_prof_end(ctx, 0, s, 2.0 * a.M * (double)a.N * a.K,
                 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (out32 ? 8.0 : 2.0) * a.M * out_cols);
    return rc;
}

extern "C" int pgv_gemm(pgv_ctx* ctx, int dtype, int epi, const void* d_A, int lda, const void* d_W, int ldw, const float* d_bias,
                        void* d_C, int ldc, int M, int N, int K, void* stream) {
    PGV_CHECK(ctx != nullptr, "pgv_gemm: null ctx");
    GemmArgs a{};
    a.A = d_A; a.lda = lda; a.W = d_W; a.ldw = ldw; a.bias = d_bias; a.C = d_C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.epi = epi;
    return pgv_launch_gemm(ctx, dtype, a, (hipStream_t)stream);
}
