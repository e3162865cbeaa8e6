// fp8 (OCP e4m3fn) weight path for the HBM-bound decode GEMVs (BASELINE config 5: 13B with fp8 weights).
//
// A matrix W[N,K] is quantised per OUTPUT ROW with a power-of-two scale:  s[n] = 2^ceil(log2(amax_n / 448)),
// q[n,k] = e4m3(W[n,k] / s[n]) (round to nearest even).  Because s is a power of two and e4m3 has 3 mantissa bits,
// q * s is exactly representable in bf16 and in fp16 (|q| <= 448): the 16-bit copy of the matrix is overwritten with the dequantised
// values, so the prefill GEMM (16-bit weights), the decode GEMV (fp8 weights, scale applied to the fp32 accumulator) and the CPU
// oracle (same dequantised values in fp32) all compute with the SAME weights -- the fp8 GEMV equals the 16-bit GEMV bit for bit.
//
// fp8 blocked layout (what the GEMV streams): block (n/16, k/64) = 1 KiB = one 16-byte load per lane of a wave;
// lane = ((k % 32) / 8) * 16 + n % 16, byte = ((k / 32) % 2) * 8 + k % 8: the two halves of a lane's 16 bytes are the A fragments of two
// consecutive v_mfma_f32_16x16x32 k-blocks.
#include "pgv_common.h"
#include "weights.h"

namespace {

__device__ __forceinline__ size_t fp8_blocked_offset(long long row, long long col, long long K) {
    return (size_t)((row >> 4) * (K >> 6) + (col >> 6)) * 1024 + (size_t)((((col & 31) >> 3) << 4) + (row & 15)) * 16 + (((col >> 5) & 1) << 3) + (col & 7);
}

// one workgroup per block of 16 rows (all K columns): absmax per row -> scale; quantise; write fp8 + dequantised 16-bit
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(typename T::elem* __restrict__ w16, unsigned char* __restrict__ w8, float* __restrict__ scales,
                                                           long long K) {
    __shared__ float amax_s[16][17];
    __shared__ float scale_s[16];
    const int tid = threadIdx.x;
    const long long rb = blockIdx.x;
    const int r = tid & 15, cg = tid >> 4;            // 16 threads per column group, 16 column groups
    // pass 1: absmax per row.  Element (row, col) of the 16-bit blocked layout
    float am = 0.f;
    for (long long c = cg; c < K; c += 16) am = fmaxf(am, fabsf((float)w16[pgv_blocked_offset(rb * 16 + r, c, K)]));
    amax_s[r][cg] = am;
    __syncthreads();
    if (tid < 16) {
        float m = 0.f;
        for (int i = 0; i < 16; ++i) m = fmaxf(m, amax_s[tid][i]);
        float s = 1.0f;
        if (m > 0.f) {
            int e;
            const float fr = frexpf(m / 448.0f, &e);                 // m/448 = fr * 2^e, fr in [0.5, 1)
            s = ldexpf(1.0f, fr == 0.5f ? e - 1 : e);                // smallest power of two >= m / 448
        }
        scale_s[tid] = s;
        scales[rb * 16 + tid] = s;
    }
    __syncthreads();
    const float inv = 1.0f / scale_s[r];                            // exact (power of two)
    for (long long c = cg * 2; c < K; c += 32) {                    // two adjacent columns per thread: one packed conversion
        const long long row = rb * 16 + r;
        const size_t o0 = pgv_blocked_offset(row, c, K), o1 = pgv_blocked_offset(row, c + 1, K);
        const float a = (float)w16[o0] * inv, b = (float)w16[o1] * inv;
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
        const f32x2_t back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
        w8[fp8_blocked_offset(row, c, K)] = (unsigned char)(pk & 0xff);
        w8[fp8_blocked_offset(row, c + 1, K)] = (unsigned char)((pk >> 8) & 0xff);
        w16[o0] = T::from_f32(back[0] * scale_s[r]);
        w16[o1] = T::from_f32(back[1] * scale_s[r]);
    }
}

// blocked 16-bit -> row-major fp32 (test / export helper)
template <typename T>
__global__ __launch_bounds__(256) void unpack_blocked_kernel(const typename T::elem* __restrict__ src, float* __restrict__ dst, long long N, long long K,
                                                             long long row_blk, long long blk_stride, long long row_off, long long rows) {
    const long long total = rows * K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / K, c = i - r * K;
        const long long pr = (row_blk > 0 ? (r / row_blk) * blk_stride + r % row_blk : r) + row_off;
        dst[i] = (float)src[pgv_blocked_offset(pr, c, K)];
    }
}

}  // namespace

int pgv_launch_quantize_fp8(int dtype, void* w16_blocked, void* w8_blocked, float* scales, long long N, long long K, hipStream_t s) {
    PGV_CHECK(N % 16 == 0 && K % 64 == 0, "quantize_fp8: need N %% 16 == 0 and K %% 64 == 0 (got %lld x %lld)", N, K);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((quantize_fp8_kernel<T>), dim3((unsigned)(N / 16)), dim3(256), 0, s, (typename T::elem*)w16_blocked,
                                                    (unsigned char*)w8_blocked, scales, K));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int pgv_launch_unpack_blocked(int dtype, const void* src, float* dst, long long N, long long K, long long row_blk, long long blk_stride, long long row_off,
                              long long rows, hipStream_t s) {
    const long long total = rows * K;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((unpack_blocked_kernel<T>), dim3(grid), dim3(256), 0, s, (const typename T::elem*)src, dst, N, K, row_blk,
                                                    blk_stride, row_off, rows));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// C ABI building blocks (tests): quantise a blocked 16-bit matrix in place + produce its fp8 copy and scales
extern "C" int pgv_quantize_fp8_blocked(pgv_ctx* ctx, int dtype, void* d_w16_blocked, void* d_w8_blocked, float* d_scales, int N, int K, void* stream) {
    PGV_CHECK(ctx && d_w16_blocked && d_w8_blocked && d_scales, "pgv_quantize_fp8_blocked: null argument");
    return pgv_launch_quantize_fp8(dtype, d_w16_blocked, d_w8_blocked, d_scales, N, K, (hipStream_t)stream);
}

extern "C" int pgv_unpack_blocked(pgv_ctx* ctx, int dtype, const void* d_src_blocked, float* d_dst, int N, int K, void* stream) {
    PGV_CHECK(ctx && d_src_blocked && d_dst && N > 0 && K > 0 && N % 16 == 0 && K % 32 == 0, "pgv_unpack_blocked: bad arguments");
    return pgv_launch_unpack_blocked(dtype, d_src_blocked, d_dst, N, K, 0, 0, 0, N, (hipStream_t)stream);
}
