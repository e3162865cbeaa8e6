// HBM-bound kernels of the path: LayerNorm / RMSNorm (fp32 residual stream in, 16-bit out), CLIP
// embedding assembly, frame preprocessing, im2col for the patch conv, casts, spatio-temporal pooling.
// All are one-pass, 16-byte vectorised, wave64 shuffle reductions, fp32 arithmetic.
#include "pgv_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: one wave per row, row kept in registers (NV float4 per lane)
//   LayerNorm: nn.LayerNorm (HF:clip/modeling_clip.py:357-358 layer_norm1/2, :605 pre_layrnorm), biased var
//   RMSNorm:   LlamaRMSNorm.forward (HF:llama/modeling_llama.py:61-67)
// ---------------------------------------------------------------------------------------------
template <typename T, int NV, bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, char* __restrict__ y, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int cols = NV * 256;
    const f32x4_t* xr = (const f32x4_t*)(x + (size_t)row * cols);
    f32x4_t v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = xr[i * 64 + lane];
    float mean = 0.f;
    if constexpr (!RMS) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        mean = wave_sum(s) * (1.0f / cols);
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; ss += d * d; }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / cols) + eps);
    u32x2_t* yr = (u32x2_t*)(y + (size_t)row * cols * 2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4_t g = ((const f32x4_t*)gamma)[i * 64 + lane];
        f32x4_t b = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!RMS) b = ((const f32x4_t*)beta)[i * 64 + lane];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        yr[i * 64 + lane] = pack4<T>(o[0], o[1], o[2], o[3]);
    }
}

template <typename T, bool RMS>
int launch_norm(const float* x, const float* g, const float* b, float eps, void* y, int rows, int cols, hipStream_t s) {
    const int grid = (rows + 3) / 4;
#define PGV_NORM_CASE(NV_) \
    case NV_: hipLaunchKernelGGL((norm_kernel<T, NV_, RMS>), dim3(grid), dim3(256), 0, s, x, g, b, eps, (char*)y, rows); break;
    switch (cols / 256) {
        PGV_NORM_CASE(1) PGV_NORM_CASE(2) PGV_NORM_CASE(3) PGV_NORM_CASE(4) PGV_NORM_CASE(8)
        PGV_NORM_CASE(16) PGV_NORM_CASE(20)
        default:
            pgv_set_error("norm: unsupported width %d", cols);
            return PGV_EINVAL;
    }
#undef PGV_NORM_CASE
    return PGV_OK;
}

// ---------------------------------------------------------------------------------------------
// CLIP embedding assembly + pre_layrnorm  (CLIPVisionEmbeddings.forward HF:clip/modeling_clip.py:200-218,
// pre_layrnorm :642): row (t, n) = (n == 0 ? class_embedding : patch_embed[t, n-1]) + position[n], LayerNorm'ed
// -> fp32 residual stream (= hidden_states[0]).
// ---------------------------------------------------------------------------------------------
// When `gnext` is given (the tower runs at least one layer) the kernel is also the first PRODUCER of the folded LayerNorm (gemm.hip, EPI_LN_*):
// x16 = round16((out - mean) * gnext), rowstat = (0, rstd) and rowmean = mean of the OUTPUT row (the centred form: here the centre IS the row's
// own mean, so the consumer's mean correction is zero), and layer 0's qkv GEMM needs no LayerNorm launch.
template <typename T, int NV>
__global__ __launch_bounds__(256) void embed_ln_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float* __restrict__ out,
                                                       int rows, int tokens, const float* __restrict__ gnext, char* __restrict__ x16,
                                                       float* __restrict__ rowstat, float* __restrict__ rowmean) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int cols = NV * 256;
    const int t = row / tokens, n = row - t * tokens;
    const f32x4_t* src = (n == 0) ? (const f32x4_t*)cls : (const f32x4_t*)(pe + ((size_t)t * (tokens - 1) + (n - 1)) * cols);
    const f32x4_t* pr = (const f32x4_t*)(pos + (size_t)n * cols);
    f32x4_t v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = src[i * 64 + lane] + pr[i * 64 + lane];
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) * (1.0f / cols);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { float d = v[i][e] - mean; ss += d * d; }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / cols) + eps);
    f32x4_t* o = (f32x4_t*)(out + (size_t)row * cols);
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4_t g = ((const f32x4_t*)gamma)[i * 64 + lane], b = ((const f32x4_t*)beta)[i * 64 + lane];
        f32x4_t r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
        o[i * 64 + lane] = r;
        v[i] = r;
        s1 += (r[0] + r[1]) + (r[2] + r[3]);
    }
    if (gnext == nullptr) return;
    const float m2 = wave_sum(s1) * (1.0f / cols);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - m2; q += d * d; }
    const float rstd2 = rsqrtf(wave_sum(q) * (1.0f / cols) + eps);
    if (lane == 0) { *(f32x2_t*)(rowstat + (size_t)row * 2) = f32x2_t{0.f, rstd2}; rowmean[row] = m2; }
    u32x2_t* xr = (u32x2_t*)(x16 + (size_t)row * cols * 2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4_t g = ((const f32x4_t*)gnext)[i * 64 + lane];
        xr[i * 64 + lane] = pack4<T>((v[i][0] - m2) * g[0], (v[i][1] - m2) * g[1], (v[i][2] - m2) * g[2], (v[i][3] - m2) * g[3]);
    }
}

// Folded LayerNorm, between producer and consumer.  The producer GEMM works on the row CENTRED at c = rowmean[row] (the row's mean before this
// sublayer's update): per-row partial (sum, sum of squares) of (x - c) over NP 64-column pieces ([NP][rows][2]) -> delta = mean(x) - c, rstd;
// rowstat = (delta, rstd) for the consumer GEMM, rowmean <- c + delta for the next producer.  var = E[(x-c)^2] - delta^2 in fp32: the
// cancellation involves only how far the mean MOVED in one sublayer, not the mean itself, so rows whose mean dominates their spread are as
// accurate as zero-mean rows (tests/test_gpu_vision.py::test_folded_layernorm_mean_dominated_rows).
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ part, float* __restrict__ rowstat, float* __restrict__ rowmean,
                                                       const float* __restrict__ cshift, int rows, int np, float inv_cols, float eps) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const f32x2_t* p = (const f32x2_t*)part + row;               // piece-major [np][rows][2]: coalesced across the rows of a wave
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < np; ++i) { const f32x2_t v = p[(size_t)i * rows]; s1 += v[0]; s2 += v[1]; }
    const float delta = s1 * inv_cols;
    const float var = fmaxf(s2 * inv_cols - delta * delta, 0.f);
    *(f32x2_t*)(rowstat + (size_t)row * 2) = f32x2_t{delta, rsqrtf(var + eps)};
    rowmean[row] += delta + (cshift ? *cshift : 0.f);       // the producer centred at rowmean + cshift
}

// mean of a vector (fp32) -> one device scalar (the data-independent shift of the row mean a biased GEMM applies; fixed-order reduction)
__global__ __launch_bounds__(256) void vec_mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (float)n;
}

// Folded LayerNorm, load time: for a consumer weight W [N, K] (row-major 16-bit) with bias b and the LayerNorm parameters (gamma, beta) in front of it:
//   colsum[n] = sum_k gamma_k W[n,k],   bias2[n] = b[n] + sum_k beta_k W[n,k]      (fp32).  One wave per output row.
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_kernel(const typename T::elem* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ colsum, float* __restrict__ bias2, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float sg = 0.f, sb = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const typename T::v8 w = *(const typename T::v8*)(W + (size_t)n * K + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) { sg += (float)w[e] * gamma[k + e]; sb += (float)w[e] * beta[k + e]; }
    }
    sg = wave_sum(sg); sb = wave_sum(sb);
    if (lane == 0) { colsum[n] = sg; bias2[n] = bias[n] + sb; }
}

// fp32 -> 16-bit cast, 4 elements per thread
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const f32x4_t* __restrict__ x, u32x2_t* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4_t v = x[i];
        y[i] = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// frame ingest: uint8 [T,H,W,3] at native resolution -> nearest resize to S x S -> (x/255 - mean)/std, NCHW 16-bit.
//   resize   = load_video (video_chatgpt/eval/model_utils.py:38-43): .float(), F.interpolate(size=...) in its default mode 'nearest',
//              back to uint8.  PyTorch's index rule (ATen UpSample.h nearest_neighbor_compute_source_index): with the fp32 scale
//              (float)in / out, src = min((int)floorf(dst * scale), in - 1); nearest sampling copies bytes, so the float round trip is exact.
//   normalise = CLIPImageProcessor.preprocess for crop-sized frames + .half() (video_chatgpt/inference.py:86-89).
// The scales are computed on the host with the same fp32 division and passed in, the per-pixel product dst * scale is one exactly
// rounded fp32 multiply on both sides.  H == W == S is the identity map (scale 1).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ingest_kernel(const uint8_t* __restrict__ in, typename T::elem* __restrict__ out, int total_px, int S, int H, int W,
                                                     float scale_h, float scale_w) {
    const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
    const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
    const int hw = S * S;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total_px; i += gridDim.x * 256) {
        const int t = i / hw, p = i - t * hw;
        const int y = p / S, x = p - y * S;
        const int sy = min((int)floorf((float)y * scale_h), H - 1), sx = min((int)floorf((float)x * scale_w), W - 1);
        const uint8_t* px = in + (((size_t)t * H + sy) * W + sx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)px[c] * (1.0f / 255.0f);
            v = (v - mean[c]) / stdv[c];
            out[((size_t)t * 3 + c) * hw + p] = T::from_f32(v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// im2col for the stride-14 patch conv (HF:clip/modeling_clip.py:151-157,208): pixels [T,3,S,S] 16-bit ->
// A0 [T*g*g, Kp] with column (c*p + i)*p + j, zero padded from 3*p*p up to Kp.  One thread per (row, c, i)
// copies one 14-pixel run (28 B, 4-byte aligned on both sides because p is even); the thread with
// ci == 3*p zero-fills the K padding.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const uint16_t* __restrict__ pix, uint16_t* __restrict__ a0, int T, int S,
                                                     int g, int p, int Kp) {
    const int per_row = 3 * p + 1;
    const long long total = (long long)T * g * g * per_row;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ci = (int)(idx % per_row);
        const long long row = idx / per_row;
        uint32_t* dst_row = (uint32_t*)(a0 + row * Kp);
        if (ci == 3 * p) {
            for (int k = (3 * p * p) / 2; k < Kp / 2; ++k) dst_row[k] = 0u;
            continue;
        }
        const int c = ci / p, i = ci - c * p;
        const int t = (int)(row / (g * g)), pr = (int)(row - (long long)t * g * g);
        const int py = pr / g, px = pr - py * g;
        const uint32_t* src = (const uint32_t*)(pix + (((size_t)t * 3 + c) * S + (py * p + i)) * S + px * p);
        uint32_t* dst = dst_row + (ci * p) / 2;
        for (int k = 0; k < p / 2; ++k) dst[k] = src[k];
    }
}

// ---------------------------------------------------------------------------------------------
// spatio-temporal pooling (video_chatgpt/inference.py:13-44; scripts/save_spatio_temporal_clip_features.py:46-57)
//   blocks [0, n_temporal*CB)        : temporal row t  = mean over P patches of frame t (zeros for t >= T)
//   blocks [n_temporal*CB, +P*CB)    : spatial  row p  = mean over T frames of patch p
// CB = C/256 channel chunks; 256 threads = 8 row-groups x 32 lanes x 8 channels (16-B loads); the 8 partial sums
// are combined through LDS in a fixed order, so the result is deterministic.  The second read of the feature
// tensor (52 MB for 100 frames) is served by L2/Infinity Cache.
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void st_pool_kernel(const char* __restrict__ feats, int T, int P, int C, long long frame_stride,
                                                      int n_temporal, char* __restrict__ out) {
    __shared__ float red[8][256];
    const int CB = C / 256;
    const int b = blockIdx.x;
    const int cb = b % CB;
    const int r = b / CB;
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int c0 = cb * 256 + l * 8;
    const bool temporal = r < n_temporal;
    const int fixed = temporal ? r : r - n_temporal;          // frame index (temporal) or patch index (spatial)
    const int count = temporal ? P : T;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!(temporal && fixed >= T)) {
        for (int i = grp; i < count; i += 8) {
            const long long elem = temporal ? ((long long)fixed * frame_stride + (long long)i * C + c0)
                                            : ((long long)i * frame_stride + (long long)fixed * C + c0);
            const typename TI::v8 v = *(const typename TI::v8*)(feats + elem * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[grp][l * 8 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll
    for (int gidx = 0; gidx < 8; ++gidx) s += red[gidx][c];
    s *= 1.0f / (float)count;
    if (temporal && fixed >= T) s = 0.f;
    ((typename TO::elem*)out)[(size_t)r * C + cb * 256 + c] = TO::from_f32(s);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
int pgv_launch_layernorm(int dtype, const float* x, const float* g, const float* b, float eps, void* y, int rows, int cols, hipStream_t s) {
    PGV_DISPATCH_DTYPE(dtype, T, return (launch_norm<T, false>(x, g, b, eps, y, rows, cols, s)));
    return PGV_OK;
}
int pgv_launch_rmsnorm(int dtype, const float* x, const float* g, float eps, void* y, int rows, int cols, hipStream_t s) {
    PGV_DISPATCH_DTYPE(dtype, T, return (launch_norm<T, true>(x, g, nullptr, eps, y, rows, cols, s)));
    return PGV_OK;
}
int pgv_launch_embed_ln(int dtype, const float* pe, const float* cls, const float* pos, const float* g, const float* b, float eps, float* out,
                        int rows, int tokens, int cols, const float* gnext, void* x16, float* rowstat, float* rowmean, hipStream_t s) {
    PGV_CHECK(cols == 1024, "embed_ln: CLIP width must be 1024 (got %d)", cols);
    PGV_CHECK(gnext == nullptr || (x16 && rowstat && rowmean), "embed_ln: the folded-LayerNorm producer needs x16 / rowstat / rowmean");
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((embed_ln_kernel<T, 4>), dim3((rows + 3) / 4), dim3(256), 0, s, pe, cls, pos, g, b, eps, out, rows, tokens,
                                                    gnext, (char*)x16, rowstat, rowmean));
    return PGV_OK;
}
int pgv_launch_ln_stats(const float* part, float* rowstat, float* rowmean, const float* cshift, int rows, int np, int cols, float eps, hipStream_t s) {
    hipLaunchKernelGGL(ln_stats_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, rowstat, rowmean, cshift, rows, np, 1.0f / (float)cols, eps);
    return PGV_OK;
}
int pgv_launch_vec_mean(const float* x, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(vec_mean_kernel, dim3(1), dim3(256), 0, s, x, n, out);
    return PGV_OK;
}
int pgv_launch_ln_fold(int dtype, const void* W, const float* bias, const float* gamma, const float* beta, float* colsum, float* bias2, int N, int K, hipStream_t s) {
    PGV_CHECK(K % 512 == 0, "ln_fold: K=%d must be a multiple of 512", K);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((ln_fold_kernel<T>), dim3((N + 3) / 4), dim3(256), 0, s, (const typename T::elem*)W, bias, gamma, beta, colsum,
                                                    bias2, N, K));
    return PGV_OK;
}
int pgv_launch_cast(int dtype, const float* x, void* y, size_t n, hipStream_t s) {
    PGV_CHECK(n % 4 == 0, "cast: element count must be a multiple of 4");
    const size_t n4 = n / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((cast_kernel<T>), dim3(grid), dim3(256), 0, s, (const f32x4_t*)x, (u32x2_t*)y, n4));
    return PGV_OK;
}
int pgv_launch_im2col(const void* pix, void* a0, int T, int S, int g, int p, int Kp, hipStream_t s) {
    PGV_CHECK(p % 2 == 0 && Kp % 2 == 0 && (S % 2) == 0, "im2col: patch/image size must be even");
    const long long total = (long long)T * g * g * (3 * p + 1);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, s, (const uint16_t*)pix, (uint16_t*)a0, T, S, g, p, Kp);
    return PGV_OK;
}

extern "C" int pgv_layernorm(pgv_ctx* ctx, int dtype, const float* d_x, const float* d_gamma, const float* d_beta, float eps, void* d_y,
                             int rows, int cols, void* stream) {
    PGV_CHECK(ctx && d_x && d_gamma && d_beta && d_y && rows > 0, "pgv_layernorm: bad arguments");
    return pgv_launch_layernorm(dtype, d_x, d_gamma, d_beta, eps, d_y, rows, cols, (hipStream_t)stream);
}
extern "C" int pgv_rmsnorm(pgv_ctx* ctx, int dtype, const float* d_x, const float* d_gamma, float eps, void* d_y, int rows, int cols,
                           void* stream) {
    PGV_CHECK(ctx && d_x && d_gamma && d_y && rows > 0, "pgv_rmsnorm: bad arguments");
    return pgv_launch_rmsnorm(dtype, d_x, d_gamma, eps, d_y, rows, cols, (hipStream_t)stream);
}

extern "C" int pgv_ingest_u8(pgv_ctx* ctx, const uint8_t* d_frames, int T, int H, int W, int image, int dtype, void* d_pixels, void* stream) {
    PGV_CHECK(ctx && d_frames && d_pixels && T > 0 && H > 0 && W > 0 && image > 0, "pgv_ingest_u8: bad arguments");
    const long long total = (long long)T * image * image;
    PGV_CHECK(total < (1ll << 31), "pgv_ingest_u8: too many pixels");
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    // compute_scales_value<float>(nullopt, in, out) of ATen: static_cast<float>(in) / out, an fp32 division
    const float scale_h = (float)H / (float)image, scale_w = (float)W / (float)image;
    PGV_DISPATCH_DTYPE(dtype, Tt, hipLaunchKernelGGL((ingest_kernel<Tt>), dim3(grid), dim3(256), 0, (hipStream_t)stream, d_frames,
                                                     (typename Tt::elem*)d_pixels, (int)total, image, H, W, scale_h, scale_w));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

extern "C" int pgv_preprocess_u8(pgv_ctx* ctx, const uint8_t* d_frames, int T, int image, int dtype, void* d_pixels, void* stream) {
    return pgv_ingest_u8(ctx, d_frames, T, image, image, image, dtype, d_pixels, stream);
}

extern "C" int pgv_st_pool(pgv_ctx* ctx, const void* d_feats, int in_dtype, int T, int P, int C, int64_t frame_stride, int n_temporal,
                           void* d_out, int out_dtype, void* stream) {
    PGV_CHECK(ctx && d_feats && d_out, "pgv_st_pool: null pointer");
    PGV_CHECK(T > 0 && P > 0 && C > 0 && C % 256 == 0, "pgv_st_pool: need T,P > 0 and C a multiple of 256 (T=%d P=%d C=%d)", T, P, C);
    PGV_CHECK(T <= n_temporal, "pgv_st_pool: T=%d exceeds the %d temporal tokens (the reference never truncates; cap frames upstream)", T, n_temporal);
    PGV_CHECK(frame_stride >= (int64_t)P * C && frame_stride % 8 == 0 && ((uintptr_t)d_feats & 15) == 0,
              "pgv_st_pool: frame stride / alignment unsupported");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (n_temporal + P) * (C / 256);
    pgv_prof_begin(ctx, 5, s);
#define PGV_POOL(TI_, TO_) \
    hipLaunchKernelGGL((st_pool_kernel<TI_, TO_>), dim3(grid), dim3(256), 0, s, (const char*)d_feats, T, P, C, (long long)frame_stride, n_temporal, (char*)d_out)
    if (in_dtype == PGV_F16 && out_dtype == PGV_F16) PGV_POOL(TF16, TF16);
    else if (in_dtype == PGV_BF16 && out_dtype == PGV_BF16) PGV_POOL(TBF16, TBF16);
    else if (in_dtype == PGV_BF16 && out_dtype == PGV_F16) PGV_POOL(TBF16, TF16);
    else if (in_dtype == PGV_F16 && out_dtype == PGV_BF16) PGV_POOL(TF16, TBF16);
    else { pgv_set_error("pgv_st_pool: unsupported dtypes %d -> %d", in_dtype, out_dtype); return PGV_EINVAL; }
#undef PGV_POOL
    pgv_prof_end(ctx, 5, s, 0.0, 2.0 * ((double)T * P * C + (double)(n_temporal + P) * C));
    return PGV_OK;
}
