#include "weights.h"

namespace {

template <typename S>
__device__ inline float load_as_f32(const void* p, long long i);
template <>
__device__ inline float load_as_f32<float>(const void* p, long long i) { return ((const float*)p)[i]; }
template <>
__device__ inline float load_as_f32<_Float16>(const void* p, long long i) { return (float)((const _Float16*)p)[i]; }
template <>
__device__ inline float load_as_f32<__bf16>(const void* p, long long i) { return (float)((const __bf16*)p)[i]; }

template <typename S, typename D>
__global__ __launch_bounds__(256) void pack_kernel(const void* __restrict__ src, D* __restrict__ dst, long long rows, long long cols,
                                                   long long dst_stride, long long row_blk, long long blk_stride, long long row_off, int blocked) {
    const long long total = rows * dst_stride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / dst_stride, c = i - r * dst_stride;
        const long long dr = (row_blk > 0 ? (r / row_blk) * blk_stride + r % row_blk : r) + row_off;
        const float v = c < cols ? load_as_f32<S>(src, r * cols + c) : 0.0f;
        dst[blocked ? pgv_blocked_offset(dr, c, dst_stride) : dr * dst_stride + c] = (D)v;
    }
}

template <typename S>
int launch_pack(const PackDst& d, const void* src, hipStream_t s) {
    const long long total = d.rows * d.dst_stride;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    switch (d.dst_dtype) {
        case PGV_F32:
            hipLaunchKernelGGL((pack_kernel<S, float>), dim3(grid), dim3(256), 0, s, src, (float*)d.ptr, d.rows, d.cols, d.dst_stride, d.row_blk, d.blk_stride, d.row_off, d.blocked ? 1 : 0);
            break;
        case PGV_F16:
            hipLaunchKernelGGL((pack_kernel<S, _Float16>), dim3(grid), dim3(256), 0, s, src, (_Float16*)d.ptr, d.rows, d.cols, d.dst_stride, d.row_blk, d.blk_stride, d.row_off, d.blocked ? 1 : 0);
            break;
        case PGV_BF16:
            hipLaunchKernelGGL((pack_kernel<S, __bf16>), dim3(grid), dim3(256), 0, s, src, (__bf16*)d.ptr, d.rows, d.cols, d.dst_stride, d.row_blk, d.blk_stride, d.row_off, d.blocked ? 1 : 0);
            break;
        default:
            pgv_set_error("pack: bad destination dtype %d", d.dst_dtype);
            return PGV_EINVAL;
    }
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

__global__ __launch_bounds__(256) void zero_rows_blocked_kernel(uint16_t* __restrict__ dst, long long row0, long long nrows, long long cols) {
    const long long total = nrows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols, c = i - r * cols;
        dst[pgv_blocked_offset(row0 + r, c, cols)] = 0;
    }
}

}  // namespace

int pgv_zero_rows_blocked(void* blocked16, long long row0, long long nrows, long long cols, hipStream_t s) {
    if (nrows <= 0) return PGV_OK;
    const long long total = nrows * cols;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(zero_rows_blocked_kernel, dim3(grid), dim3(256), 0, s, (uint16_t*)blocked16, row0, nrows, cols);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

int pgv_pack_tensor(const PackDst& d, const void* data, int src_dtype, int on_device, hipStream_t s) {
    PGV_CHECK(data != nullptr && d.ptr != nullptr && d.rows > 0 && d.cols > 0, "pack: bad arguments");
    PGV_CHECK(!d.blocked || (d.dst_stride % 32 == 0), "pack: blocked layout needs a multiple-of-32 width");
    PGV_CHECK(src_dtype == PGV_F32 || src_dtype == PGV_F16 || src_dtype == PGV_BF16, "pack: bad source dtype %d", src_dtype);
    const void* src = data;
    void* tmp = nullptr;
    if (!on_device) {
        const size_t bytes = (size_t)d.rows * d.cols * pgv_dtype_size(src_dtype);
        PGV_HIP(hipMalloc(&tmp, bytes));
        hipError_t e = hipMemcpy(tmp, data, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(tmp); pgv_set_error("pack: H2D copy failed: %s", hipGetErrorString(e)); return PGV_EHIP; }
        src = tmp;
    }
    int rc;
    if (src_dtype == PGV_F32) rc = launch_pack<float>(d, src, s);
    else if (src_dtype == PGV_F16) rc = launch_pack<_Float16>(d, src, s);
    else rc = launch_pack<__bf16>(d, src, s);
    if (tmp) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(tmp);
    }
    return rc;
}
