// Internal interface of the LLaMA decoder translation units (llm.hip = host drivers; llm_prefill.hip, gemv.hip, decode_attn.hip, sampling.hip =
// kernels + launchers): launcher prototypes, mode enums and the few device helpers more than one of them uses.
#pragma once
#include "pgv_common.h"

constexpr int kHD = 128;             // LLaMA head_dim
constexpr int kMaxBatch = 64;        // decode GEMVs: up to 4 MFMA column tiles of 16 sequences per weight fragment
enum { GV_STORE16 = 0, GV_RESID = 1, GV_SWIGLU = 2, GV_F32 = 3, GV_RESIDNORM = 5 };     // gemv.hip epilogues
enum { AM_INC_POS = 1, AM_RECORD = 2, AM_SAMPLE = 4 };                                  // sampling.hip bookkeeping flags
constexpr int kDattnSplitMax = 8, kDattnPart = kHD + 2;                                 // decode_attn.hip: context splits, floats per partial state

// vit_attn.hip
int pgv_vit_attn_configure(pgv_ctx* ctx);  // per-device dynamic-LDS opt-in of the attention kernels: called once per context
// elementwise.hip
int pgv_launch_rmsnorm(int dtype, const float* x, const float* g, float eps, void* y, int rows, int cols, hipStream_t s);
// llm_prefill.hip
int pgv_launch_embed_splice(int dtype, const int* row_src, const void* embed, const void* video, float* resid, int M, int H, hipStream_t s);
int pgv_launch_gather_rows(const float* src, const int* rows, float* dst, int B, int H, hipStream_t s);
int pgv_launch_rope_kv_write(int dtype, void* qkv, const int* row_b, const int* row_pos, const void* rope, void* Kc, void* Vc, int M, int H,
                             int heads, int max_seq, hipStream_t s);
int pgv_launch_prefill_attn(pgv_ctx* ctx, int dtype, const void* qkv, void* out, const void* Kc, const void* Vc, const int* cu, const int* koff, int B, int max_len,
                            int H, int heads, int max_seq, double flops, hipStream_t s);
// gemv.hip
int pgv_launch_gemv(pgv_ctx* ctx, int dtype, int mode, const void* W, const void* x, int ldx, void* out, int ldo, int N, int K, int B, hipStream_t s,
                    const float* wscale = nullptr, const GemvNorm* norm = nullptr);
int pgv_gemv_configure(pgv_ctx* ctx);      // per-device function attributes of the 8-phase producers: called once per context, outside graph capture
int pgv_launch_embed_tok_norm(int dtype, const int* tok, const void* embed, float* resid, const float* gamma, void* xg, float* ssq, int B, int H, hipStream_t s, bool x_blocked);
int pgv_launch_final_prep(int dtype, const float* resid, const float* gamma, void* xg, float* ssq, int B, int H, hipStream_t s, bool x_blocked);
int pgv_gemv_xblk_tiles(int B);            // column tiles of the blocked activation layout at B sequences (0: row-major)
// fp8.hip
int pgv_launch_quantize_fp8(int dtype, void* w16_blocked, void* w8_blocked, float* scales, long long N, long long K, hipStream_t s);
int pgv_launch_unpack_blocked(int dtype, const void* src, float* dst, long long N, long long K, long long row_blk, long long blk_stride, long long row_off,
                              long long rows, hipStream_t s);
// decode_attn.hip
int pgv_launch_decode_attn(pgv_ctx* ctx, int dtype, const void* qkv, const int* pos, const void* rope, void* Kc, void* Vc, void* out, int B, int H,
                           int heads, int max_seq, double bytes, hipStream_t s, float* part, unsigned* ticket);
// sampling.hip
int pgv_launch_sample(const float* logits, int V, int B, float temperature, int top_k, const float* u, int u_stride, int u_by_step, int* next, int* pos,
                      int* step, int* hist, int hist_stride, int* done, int eos, int advance, hipStream_t s);
int pgv_launch_argmax_parts(const float* val, const int* idx, int nblk, int amax_ts, int V, int B, int* next, int* pos, int* step, int* hist, int hist_stride, int* done,
                            int eos, int advance, hipStream_t s);

#if defined(__HIPCC__)
// sum over each aligned group of 16 lanes with DPP (VALU latency) instead of ds_bpermute shuffles (LDS crossbar latency):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8 -> every lane of the row ends up with the row total.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    return v;
}

// Buffer descriptor over `bytes` bytes at `base` from provably wave-uniform inputs (cdna_hip_programming.md T20): lanes whose voffset lies outside
// the range are dropped by the hardware -- no memory request, the result is 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gv_make_rsrc(const void* base, unsigned bytes) {
    const uintptr_t b = (uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Cross-row sums with gfx950's lane-swap instructions instead of ds_bpermute round trips.  Only the FIRST result of the swap is used (vdst:
// even rows / the low half keep their value, odd rows / the high half receive the partner's -- checked on hardware, scripts/lab/permlane_swap.hip):
// v + swap(v, v) is the pair sum in the odd rows / the high half, so after both steps the total of the four rows sits in ROW 3 (lanes 48..63).
// (The second result would give the sum everywhere, but with one value on both operands hipcc of ROCm 7.2 reads it from the wrong register.)
__device__ __forceinline__ float rows_sum_to_row3(float v) {
    const unsigned a = __builtin_bit_cast(unsigned, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_permlane16_swap(a, a, false, false)[0]);          // rows 1, 3: v1 + v0, v3 + v2
    const unsigned b = __builtin_bit_cast(unsigned, v);
    return v + __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(b, b, false, false)[0]);    // row 3: (v3 + v2) + (v1 + v0)
}
#endif
