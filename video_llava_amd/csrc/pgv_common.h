// Internal helpers shared by the libpgv translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/pgv.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void pgv_set_error(const char* fmt, ...);

#define PGV_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            pgv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return PGV_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

#define PGV_CHECK(cond, ...)                                                                   \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            pgv_set_error(__VA_ARGS__);                                                        \
            return PGV_EINVAL;                                                                 \
        }                                                                                      \
    } while (0)

#define PGV_TRY(expr)                                                                          \
    do {                                                                                       \
        int _r = (expr);                                                                       \
        if (_r != PGV_OK) return _r;                                                           \
    } while (0)

// ---------------------------------------------------------------------------------------------
// context: device + bump-allocated workspace arena + optional per-family event timers
// ---------------------------------------------------------------------------------------------
struct pgv_prof_family {
    std::vector<hipEvent_t> ev;   // start/stop pairs
    size_t used = 0;              // events used
    double flops = 0, bytes = 0;
    int64_t launches = 0;
};

constexpr int PGV_ZERO_BIAS_LEN = 1 << 16;

struct pgv_ctx {
    int device = 0;
    int num_cu = 256;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    size_t ws_off = 0;
    bool prof = false;
    pgv_prof_family fam[PGV_NFAMILY];
    hipStream_t cap_stream = nullptr;   // used only to capture decode graphs
    hipStream_t ws_stream = nullptr;    // stream of the last call that carved the arena: compared by value only, never used after that call
    bool ws_stream_valid = false;
    hipEvent_t ws_event = nullptr;      // recorded on the user's stream at the end of every arena-using call (pgv_ws_release)
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};   // further lanes of the multi-lane ViT pass (vit.hip): forked from / joined into the caller's stream by events
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    float* zero_bias = nullptr;         // PGV_ZERO_BIAS_LEN zeros: stands in for a null bias so the persistent GEMM epilogue is branch-free
};

// Reserve the arena for a call enqueued on `s` (may hipMalloc: never call inside graph capture).  The arena is reused by every call, so
// the only ordering between two users is stream order: every user ends with pgv_ws_release (an event recorded on its own stream), and when
// `s` differs from the stream of the previous user, `s` first waits for that event -- two torch streams on one context stay correct (if
// serialised) and a stream handle is never used after the call it was passed to.
int pgv_ws_reserve(pgv_ctx* ctx, size_t bytes, hipStream_t s);
int pgv_ws_release(pgv_ctx* ctx, hipStream_t s);
// Bump allocate (256-B aligned) from the reserved arena; nullptr if exhausted.
void* pgv_ws_alloc(pgv_ctx* ctx, size_t bytes);
inline void pgv_ws_reset(pgv_ctx* ctx) { ctx->ws_off = 0; }
inline size_t pgv_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// RAII-free profiling scope helpers
void pgv_prof_begin(pgv_ctx* ctx, int family, hipStream_t s);
void pgv_prof_end(pgv_ctx* ctx, int family, hipStream_t s, double flops, double bytes);

// ---------------------------------------------------------------------------------------------
// device-side dtype traits
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;     // result of ds_read_b64_tr_b16
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

struct TF16 {
    using elem = _Float16;
    using v8 = half8_t;
    using v4 = half4_t;
    using v2 = half2_t;
    static constexpr int id = PGV_F16;
    __device__ static inline f32x16_t mfma32(v8 a, v8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    __device__ static inline f32x4_t mfma16(v8 a, v8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    __device__ static inline elem from_f32(float x) { return (_Float16)x; }
    __device__ static inline float to_f32(elem x) { return (float)x; }
};

struct TBF16 {
    using elem = __bf16;
    using v8 = bf16x8_t;
    using v4 = bf16x4_t;
    using v2 = bf16x2_t;
    static constexpr int id = PGV_BF16;
    __device__ static inline f32x16_t mfma32(v8 a, v8 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static inline f32x4_t mfma16(v8 a, v8 b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static inline elem from_f32(float x) { return (__bf16)x; }
    __device__ static inline float to_f32(elem x) { return (float)x; }
};

// pack 4 floats -> 4 x 16-bit (8 bytes)
template <typename T>
__device__ inline u32x2_t pack4(float a, float b, float c, float d) {
    typename T::v4 v;
    v[0] = T::from_f32(a); v[1] = T::from_f32(b); v[2] = T::from_f32(c); v[3] = T::from_f32(d);
    return __builtin_bit_cast(u32x2_t, v);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// dispatch on runtime dtype
#define PGV_DISPATCH_DTYPE(dtype, T, ...)                       \
    do {                                                        \
        if ((dtype) == PGV_F16) { using T = TF16; __VA_ARGS__; } \
        else if ((dtype) == PGV_BF16) { using T = TBF16; __VA_ARGS__; } \
        else { pgv_set_error("unsupported dtype %d", (int)(dtype)); return PGV_EINVAL; } \
    } while (0)

// ---------------------------------------------------------------------------------------------
// internal launchers (defined in the kernel translation units)
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
    const void* A; int lda;       // [M,K] 16-bit
    const void* W; int ldw;       // [N,K] 16-bit (N padded to 256 rows, K to 64)
    const float* bias;            // [N] fp32 or null
    void* C; int ldc;             // 16-bit / fp32 out, or fp32 residual (RESID epilogues)
    int M, N, K;
    int epi;
    bool w_blocked = false;       // W in the fragment-blocked layout (weights.h) -- the decoder's packed weights
    // folded LayerNorm (gemm.hip, EPI_LN_*): consumer side rowstat [M][2] + colsum [N]; producer side gnext [N], x16 [M][ldx16], stats_part [N/64][M][2], rowmean [M]
    const float* rowstat = nullptr; const float* colsum = nullptr;
    const float* gnext = nullptr; void* x16 = nullptr; int ldx16 = 0; float* stats_part = nullptr;
    const float* rowmean = nullptr;      // producer: per-row mean before the sublayer (null: centre 0)
    const float* cshift = nullptr;       // producer: device scalar mean(bias), added to rowmean to form the centre
};
enum { PGV_EPI_LN_BIAS = 8, PGV_EPI_LN_BIAS_QGELU = 9, PGV_EPI_BIAS_RESID_LNOUT = 10 };     // internal epilogues (gemm.hip)
int pgv_launch_gemm(pgv_ctx* ctx, int dtype, const GemmArgs& a, hipStream_t s);

// Folded-RMSNorm arguments of the decode GEMVs (gemv.hip, GemvArgs).  Consumer modes (store16 / swiglu / f32): ssq_in [nparts_in][16]
// + hidden + eps (null ssq_in = plain GEMV).  Producer mode (residual + norm): out = fp32 residual, gamma / xg / ssq_out.
struct GemvNorm {
    const float* ssq_in = nullptr; int nparts_in = 0; int hidden = 1; float eps = 0.f;
    const float* gamma = nullptr; void* xg = nullptr; float* ssq_out = nullptr;
    float* amax_val = nullptr; int* amax_idx = nullptr;      // f32 mode: per-workgroup greedy candidates [ceil(N / 16)][16]
    void* k8_part = nullptr;                                 // producer mode, batches beyond 16: scratch of the 8-phase form, [N / 16][8][B tiles][64] float4; null = 16-row kernel
    int ssq_ts = 0, amax_ts = 0;                             // batches beyond 16: the side arrays are tile-major [B / 16][...][16] with these tile strides (elements)
    // batches beyond 16: the normalised 16-bit operand (consumer: x; producer: xg) is in the fragment-blocked ACTIVATION layout
    // [K / 32][column tiles][4 k-groups][16 sequences][8 elements] (gv_xblk_offset): a consumer's MFMA B fragment is one contiguous 1 KiB wave-load
    bool x_blocked = false;
};

// Byte offset of activation element (sequence b, feature n) in the fragment-blocked activation layout with `ct` column tiles of 16 sequences.
// Four consecutive n (n % 4 == 0) of one sequence are 8 contiguous bytes, eight (n % 8 == 0) are 16.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t gv_xblk_offset(int b, int n, int ct) {
    return ((((size_t)(n >> 5) * ct + (b >> 4)) * 4 + ((n & 31) >> 3)) * 16 + (b & 15)) * 16 + (size_t)(n & 7) * 2;
}
