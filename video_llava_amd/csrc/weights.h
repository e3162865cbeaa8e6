// Weight packing helpers shared by vit.hip and llm.hip: convert one checkpoint tensor (fp32/fp16/bf16,
// host or device) into the library's packed device storage (16-bit GEMM operands, fp32 biases / norm
// parameters), optionally re-mapping rows (fused qkv, interleaved gate/up) and zero-padding columns.
#pragma once
#include <set>
#include <string>

#include "pgv_common.h"

struct PackDst {
    void* ptr = nullptr;      // destination base
    int dst_dtype = PGV_F32;  // PGV_F16 / PGV_BF16 / PGV_F32
    long long rows = 0, cols = 0;   // logical source shape (rows x cols, row-major contiguous)
    long long dst_stride = 0;       // destination row stride (elements) >= cols; [cols, dst_stride) is zero-filled
    long long row_blk = 0;          // 0: dst_row = r + row_off;  else dst_row = (r / row_blk) * blk_stride + r % row_blk + row_off
    long long blk_stride = 0;
    long long row_off = 0;
    // blocked = true: destination is the MFMA-fragment-blocked layout [rows/16][cols/32][kgroup 4][row 16][8 elems]
    // (one 1 KiB block = exactly what one wave loads for a 16x16x32 A fragment); cols must be a multiple of 32.
    bool blocked = false;
};
// element offset of (row, col) in the blocked layout of a [rows, cols] matrix
inline __host__ __device__ long long pgv_blocked_offset(long long row, long long col, long long cols) {
    return ((row >> 4) * (cols >> 5) + (col >> 5)) * 512 + ((((col & 31) >> 3) << 4) + (row & 15)) * 8 + (col & 7);
}

// Enqueue the conversion on `s`.  Host sources are staged through a temporary device buffer (synchronous).
int pgv_pack_tensor(const PackDst& d, const void* data, int src_dtype, int on_device, hipStream_t s);

// Zero rows [row0, row0 + nrows) of a 16-bit matrix in the blocked layout (all-zero bits are +0 in fp16 and bf16).
int pgv_zero_rows_blocked(void* blocked16, long long row0, long long nrows, long long cols, hipStream_t s);

inline size_t pgv_dtype_size(int dt) { return dt == PGV_F32 ? 4 : 2; }
