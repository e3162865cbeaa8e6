// LLaMA decoder kernels for gfx950 (HF:llama/modeling_llama.py; reference glue video_chatgpt/model/video_chatgpt.py):
//   embed_splice   : embed_tokens gather + replacement of the <vid_patch> run by projected video rows (:100-168)
//   rope_kv_write  : rotate-half RoPE on q,k (:129-160) + KV-cache append (prefill)
//   prefill_attn   : causal flash attention, head_dim 128, K/V streamed from the cache through LDS, MFMA
//   gemv_mfma      : decode projections y[B,N] = x[B,K] W[N,K]^T for B <= 16 on v_mfma_f32_16x16x32 (HBM-bound:
//                    every weight byte is read exactly once, split-K across the 8 waves of a workgroup)
//   decode_attn    : RoPE + cache append + single-query attention over the cache (flash-decoding style)
//   argmax         : greedy token + per-sequence bookkeeping (position, step, EOS stickiness) on the device
#include <stdlib.h>
#include <type_traits>

#include "pgv_common.h"

namespace {

constexpr int HD = 128;   // LLaMA head_dim

// sum over each aligned group of 16 lanes with DPP (VALU latency) instead of ds_bpermute shuffles (LDS crossbar latency):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8 -> every lane of the row ends up with the row total.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    return v;
}

// ---------------------------------------------------------------------------------------------
// embedding gather + video splice -> fp32 residual stream.  row_src[r] >= 0: token id; < 0: -(video row + 1)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_splice_kernel(const int* __restrict__ row_src, const typename T::elem* __restrict__ embed,
                                                           const typename T::elem* __restrict__ video, float* __restrict__ resid, int H) {
    const int r = blockIdx.x;
    const int src = row_src[r];
    const typename T::elem* p = src >= 0 ? embed + (size_t)src * H : video + (size_t)(-src - 1) * H;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        const typename T::v8 v = *(const typename T::v8*)(p + c);
        f32x4_t a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (float)v[e]; b[e] = (float)v[4 + e]; }
        *(f32x4_t*)(resid + (size_t)r * H + c) = a;
        *(f32x4_t*)(resid + (size_t)r * H + c + 4) = b;
    }
}

// gather rows (last token of every sequence) of the fp32 residual
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows, float* __restrict__ dst, int H) {
    const int b = blockIdx.x;
    const f32x4_t* s = (const f32x4_t*)(src + (size_t)rows[b] * H);
    f32x4_t* d = (f32x4_t*)(dst + (size_t)b * H);
    for (int c = threadIdx.x; c < H / 4; c += 256) d[c] = s[c];
}

// ---------------------------------------------------------------------------------------------
// prefill: RoPE on q (in place in the qkv buffer) and k, append k/v to the cache.
// rope table: [max_pos][64] (cos, sin) fp32.  One block per token row.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_write_kernel(typename T::elem* __restrict__ qkv, const int* __restrict__ row_b,
                                                            const int* __restrict__ row_pos, const float2* __restrict__ rope,
                                                            typename T::elem* __restrict__ Kc, typename T::elem* __restrict__ Vc, int H, int heads, int max_seq) {
    const int r = blockIdx.x;
    const int b = row_b[r], pos = row_pos[r];
    typename T::elem* q = qkv + (size_t)r * 3 * H;
    typename T::elem* k = q + H;
    typename T::elem* v = q + 2 * H;
    for (int idx = threadIdx.x; idx < heads * 64; idx += 256) {
        const int h = idx >> 6, j = idx & 63;
        const float2 cs = rope[(size_t)pos * 64 + j];
        const float q1 = (float)q[h * HD + j], q2 = (float)q[h * HD + j + 64];
        q[h * HD + j] = T::from_f32(q1 * cs.x - q2 * cs.y);
        q[h * HD + j + 64] = T::from_f32(q2 * cs.x + q1 * cs.y);
        const float k1 = (float)k[h * HD + j], k2 = (float)k[h * HD + j + 64];
        typename T::elem* kd = Kc + (((size_t)b * heads + h) * max_seq + pos) * HD;
        kd[j] = T::from_f32(k1 * cs.x - k2 * cs.y);
        kd[j + 64] = T::from_f32(k2 * cs.x + k1 * cs.y);
        typename T::elem* vd = Vc + (((size_t)b * heads + h) * max_seq + pos) * HD;
        vd[j] = v[h * HD + j];
        vd[j + 64] = v[h * HD + j + 64];
    }
}

// ---------------------------------------------------------------------------------------------
// prefill causal attention.  Grid (qtiles, heads, B); 4 waves, wave w owns queries q0 + 32w + (lane&31).
// Per 64-key chunk: K rows DMA'd into LDS ([64][256 B], 16-B chunks XOR-swizzled by row&15 on the source side),
// V transposed through registers into V^T [128][144 B] with key bits 2/3 swapped (see vit_attn.hip); then
// S^T = K Q^T, online softmax (lane-local + one cross-half shuffle), O^T += V^T P^T, all on 32x32x16 MFMA.
// ---------------------------------------------------------------------------------------------
struct PrefillAttnArgs {
    const char* qkv;      // [M, 3H]: rotated q at cols [0,H)
    char* out;            // [M, H]
    const char* Kc;       // [B, heads, max_seq, 128]
    const char* Vc;
    const int* cu;        // [B+1] row offsets
    int H, heads, max_seq;
    float scale_log2e;
};

template <typename T>
__global__ __launch_bounds__(256) void prefill_attn_kernel(PrefillAttnArgs p) {
    __shared__ __attribute__((aligned(16))) char Ks[64 * 256];
    __shared__ __attribute__((aligned(16))) char Vs[64 * 256];   // row-major like K; 16-B chunk index ^ ((row & 3) << 2)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int r0 = p.cu[b], S = p.cu[b + 1] - r0;
    const int q0 = blockIdx.x * 128;
    if (q0 >= S) return;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qi = q0 + w * 32 + l31;                 // this lane's query index within the sequence
    const int qrow = r0 + min(qi, S - 1);
    const char* kbase = p.Kc + ((size_t)b * p.heads + h) * p.max_seq * HD * 2;
    const char* vbase = p.Vc + ((size_t)b * p.heads + h) * p.max_seq * HD * 2;

    typename T::v8 qf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const typename T::v8*)(p.qkv + ((size_t)qrow * 3 * p.H + h * HD + kk * 16 + hi * 8) * 2);

    f32x16_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
    const float NEG = -1e30f;
    float mrun = NEG, lrun = 0.f;
    int koffs[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) koffs[kk] = ((kk * 2 + hi) ^ (l31 & 15)) << 4;

    // transposing V reads: lane i of 16-lane group g addresses key (4 hi + i/4) of a 16-key slice, d columns j*32 + g*16 + (i%4)*4 .. +3
    const int li = lane & 15, lg = (lane >> 4) & 1;
    const int voff0 = (4 * hi + (li >> 2)) * 256 + (((lg * 2 + ((li & 3) >> 1)) ^ ((li >> 2) << 2)) << 4) + (li & 1) * 8;   // d block j: ^ (j * 64)

    const int kend = min(S, q0 + 128);
    for (int kc0 = 0; kc0 < kend; kc0 += 64) {
        __syncthreads();
        // stage K: wave-instruction = 4 rows x 256 B
        {
            const int srow = lane >> 4, slot = lane & 15;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int g = w * 4 + g4;                  // 16 groups of 4 rows
                const int row = g * 4 + srow;
                const int chunk = slot ^ (row & 15);
                const int kr = min(kc0 + row, S - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + ((size_t)kr * HD + chunk * 8) * 2),
                                                 (__attribute__((address_space(3))) void*)(Ks + g * 1024), 16, 0, 0);
            }
        }
        // stage V the same way (row-major; the V^T MFMA fragments come out of it through ds_read_b64_tr_b16, see vit_attn.hip).  The 32 lanes of
        // an LDS cycle of those reads touch 4 consecutive keys x 64 B: chunk ^ ((row & 3) << 2) puts them in the four 64-B quarters of a bank row.
        // (The first version transposed V through registers with ds_write_b16: SQ_LDS_BANK_CONFLICT was 84 % of SQ_LDS_IDX_ACTIVE.)
        {
            const int srow = lane >> 4, slot = lane & 15;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int g = w * 4 + g4;
                const int row = g * 4 + srow;
                const int chunk = slot ^ ((row & 3) << 2);
                const int vr = min(kc0 + row, S - 1);      // rows past S: finite duplicates, their probabilities are exactly 0
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + ((size_t)vr * HD + chunk * 8) * 2),
                                                 (__attribute__((address_space(3))) void*)(Vs + g * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kc0 > q0 + w * 32 + 31) continue;           // chunk entirely in this wave's future: nothing to add

        f32x16_t s[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[c][e] = 0.f;
            const char* kr = Ks + (c * 32 + l31) * 256;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const typename T::v8 kf = *(const typename T::v8*)(kr + koffs[kk]);
                s[c] = T::mfma32(kf, qf[kk], s[c]);
            }
        }
        float cmax = NEG;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kc0 + c * 32 + 4 * hi + (e & 3) + 8 * (e >> 2);
                float v = s[c][e] * p.scale_log2e;
                v = (key <= qi && key < S) ? v : NEG;
                s[c][e] = v;
                cmax = fmaxf(cmax, v);
            }
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
        const float mnew = fmaxf(mrun, cmax);
        const float alpha = exp2f(mrun - mnew);
        mrun = mnew;
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                // a fully masked row (possible only for padded queries) keeps p = exp2(NEG - NEG) = 1; harmless, never stored
                const float pv = exp2f(s[c][e] - mnew);
                s[c][e] = pv;
                psum += pv;
            }
        lrun = lrun * alpha + psum;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename T::v8 pa;
#pragma unroll
                for (int e = 0; e < 8; ++e) pa[e] = T::from_f32(s[c][ks * 8 + e]);
                const char* vb = Vs + (c * 32 + ks * 16) * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const char* a0 = vb + (voff0 ^ (j * 64));
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a0);
                    const s16x4_t up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 8 * 256));
                    o[j] = T::mfma32(__builtin_bit_cast(typename T::v8, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7)), pa, o[j]);
                }
            }
    }
    const float ltot = lrun + __shfl_xor(lrun, 32, 64);
    const float inv = 1.0f / ltot;
    if (qi < S) {
        char* orow = p.out + ((size_t)(r0 + qi) * p.H + h * HD) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * hi;
                *(u32x2_t*)(orow + d * 2) = pack4<T>(o[j][g * 4] * inv, o[j][g * 4 + 1] * inv, o[j][g * 4 + 2] * inv, o[j][g * 4 + 3] * inv);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// decode GEMV on MFMA: y[b, n] = sum_k x[b, k] W[n, k], B <= 16.
// W is stored in the fragment-blocked layout (weights.h): block (n/16, k/32) is the 1 KiB a wave loads as ONE
// v_mfma_f32_16x16x32 A fragment, so every wave-load is a single contiguous, fully coalesced 1 KiB burst.
// Workgroup = NW waves, owns TL row blocks; wave w takes the 64-column groups w, w+NW, ... (adjacent 2 KiB of the same row block).
// B operand = x fragment (lane: batch l&15, k (l>>4)*8..+8) served by L2.  Partial 16x16 tiles are reduced through LDS.
// ---------------------------------------------------------------------------------------------
enum { GV_STORE16 = 0, GV_RESID = 1, GV_SWIGLU = 2, GV_F32 = 3, GV_RESIDNORM = 5 };

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

// RMSNorm is folded into the GEMVs around it (decode only; no launch of its own):
//   producer (GV_RESIDNORM: o_proj, down_proj; also the embedding gather): r = resid + y is written back in fp32, xg = round16(r * gamma) --
//     gamma of the norm that FOLLOWS, known statically -- is written as the 16-bit operand of the next GEMV, and the sum of r^2 over the
//     workgroup's 16 rows goes to ssq_out[workgroup][b];
//   consumer (qkv, gate/up, lm_head): y = (W xg) * rstd[b] with rstd = rsqrt(sum_parts ssq / H + eps), applied to the fp32 accumulators.
// W (r rstd gamma) = rstd W (r gamma) exactly in real arithmetic; in floating point the 16-bit rounding now happens before the scale by rstd
// instead of after it -- the same relative error, no weight is modified, so the fp8 path keeps its bit-equality with the 16-bit path.
struct GemvArgs {
    const char* W; const char* x; char* out;
    int N, K, B, ldx, ldo;
    const float* wscale;   // W8 = true: per-row power-of-two scales of the fp8 matrix (fp8.hip)
    // consumer side of the folded RMSNorm (null ssq_in: plain GEMV)
    const float* ssq_in; int nparts_in; float inv_h, eps;      // ssq_in [nparts_in][16]
    // producer side (GV_RESIDNORM): out = fp32 residual [B][ldo] (read-modify-write)
    const float* gamma; char* xg; float* ssq_out;              // xg [B][ldo] 16-bit, ssq_out [gridDim.x][16]
    // GV_F32 (lm_head): per-workgroup greedy candidates -- the largest logit of the workgroup's 16 rows and its (smallest) index per batch
    // column -- so the token pick scans N / 16 candidates instead of N logits (null: not wanted)
    float* amax_val; int* amax_idx;                            // [gridDim.x][16]
    // batches beyond one MFMA tile (CT column tiles of 16 sequences, B <= 16 CT): every per-batch-column side array above is tile-major,
    // [CT][...][16], with these tile strides in elements
    int ssq_ts, amax_ts;
    unsigned lds_bytes;                                        // gemv_k8_kernel: bytes of LDS for the x slice of one pass
    // A8 (the fp8 x fp8 MFMA form, PGV_FP8_MFMA=1): x is the hi / lo e4m3 image written by quant_hilo_kernel, xscale [B][2] its per-token scales
    const float* xscale;
};

// W8 = true streams the fp8 (e4m3) blocked copy of the matrix: one 16-byte load per lane carries the A fragments of TWO consecutive
// k-blocks, the codes are widened to the activation dtype in registers (exact) and the per-row scale multiplies the fp32 result, so the
// output is bit-identical to the 16-bit kernel on the dequantised matrix while the weight stream is half as long.
// 8 e4m3 codes -> 8 x 16-bit: v_cvt_scalef32_pk_{f16,bf16}_fp8 widens two codes per instruction (scale 1.0; e4m3 values are exactly
// representable in fp16 and in bf16), 4 VALU per MFMA operand.  (Round 1 went through fp32 -- v_cvt_pk_f32_fp8 + a 16-bit pack, 16 VALU per
// operand -- and the fp8 GEMVs were VALU-bound at 3.5 TB/s of fp8 bytes.)
template <typename T>
__device__ __forceinline__ typename T::v8 fp8x8_to_v8(unsigned lo, unsigned hi) {
    typename T::v2 a, b, c, d;
    if constexpr (T::id == PGV_F16) {
        a = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(lo, 1.0f, false); b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(lo, 1.0f, true);
        c = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(hi, 1.0f, false); d = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(hi, 1.0f, true);
    } else {
        a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false); b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
        c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false); d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    }
    typename T::v8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1]; r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
    return r;
}

// NW = waves per workgroup (the K split inside a workgroup): 8.  The residual producers cannot split K across workgroups (the workgroup that
// owns 16 output rows must see their complete sums to emit xg and the sum of squares), so o_proj / down_proj run 256 workgroups = one per CU
// with 8 waves; the pipelined weight stream below keeps enough loads in flight for that (down_proj 16.4 us; a 16-wave variant, needed
// before the stream was pipelined, takes 19.3 us: twice the LDS partials and 12 % dead tail loads at 10.75 groups per wave).
// TL = row blocks of 16 per workgroup.  Every row block of a workgroup multiplies the SAME x fragments, so the activation loads -- which are
// re-read by every workgroup and cost as many requests on the CU's load path as the weights of ONE row block -- are shared: 2 for gate/up
// (the SiLU pair), 3 for qkv when the row-block count divides (7B: 768 = 3 x 256 workgroups, one per CU), 1 otherwise.
// PU = 64-column groups per register buffer.  The weight stream is software-pipelined over two buffers: the loads of batch i+1 are issued
// before the MFMAs of batch i, so a wave always has one or two batches in flight (no round trip with an empty memory pipe between batches,
// which is what the launches with ONE workgroup per CU -- the producers, qkv with TL = 3 -- cannot hide behind another workgroup).
// X2 (B <= 8 only): the x fragments of BOTH k-blocks of a 64-column group come from one wave-load.  An MFMA B operand has 16 batch columns;
// with at most 8 sequences the lanes of columns 8..15 are free, so they fetch the second k-block of columns 0..7 -- per batch row the wave
// then reads one full 128-byte line instead of two half lines in two instructions -- and a row_ror:8 DPP move hands it to the lanes that feed
// the second MFMA.  (Columns >= 8 of either operand then hold the other half's data: they only reach output columns that are never stored.)
// The activations are re-read from L2 by every workgroup: this halves their requests on the CU's load path and in the L2, which is what the
// fp8 GEMVs (half the weight requests, the same x requests) were bound by: 3.5 -> 4.4 TB/s of fp8 bytes; 16-bit weights: +0.7 %.
// CT (round 4): column tiles of 16 sequences per weight fragment, B <= 16 CT (decode batches up to 64).  The weights are still streamed once;
// every tile is a separate pass of the 16-column arithmetic (its own accumulators, its own sum-of-squares reduction, its own epilogue), so a
// sequence's results are BITWISE the same whether it is decoded alone or next to 63 others.
// A8 (round 4; BASELINE configs[4] names an "fp8 MFMA weight path"): the weights' e4m3 codes go into v_mfma_f32_16x16x32_fp8_fp8 as they are
// (no widening VALU) and the activation enters as TWO e4m3 operands, x ~ s_hi hi + s_lo lo (quant_hilo_kernel below: per-token power-of-two
// scales from the exact amax of x and of x - s_hi hi), one MFMA each into separate accumulators that the epilogue combines.  Products of
// e4m3 pairs are exact in fp32, so the only error against the weight-only path is the 7 - 8 bits the pair keeps of every activation:
// oracle/a8_study.py measures 1.8e-2 on the 40-layer 13B logits (plain e4m3 activations: 5.9e-1; bf16 activations: 6.7e-2).  The x image
// has the 16-bit operand's footprint (16 B per 8 values: 8 hi codes, 8 lo codes), so every load above is unchanged.
template <typename T, int MODE, bool W8, int NW, int TL, int PU, bool X2, int CT = 1, bool A8 = false>
__global__ __launch_bounds__(NW * 64) void gemv_mfma_kernel(GemvArgs p) {
    constexpr int TILES = TL;
    static_assert(!X2 || CT == 1, "the merged x load uses the lanes of columns 8..15");
    static_assert(!A8 || (W8 && CT == 1), "the fp8 x fp8 form streams the fp8 weights; batches up to 16");
    __shared__ f32x4_t red[NW][TILES * CT][64];
    __shared__ f32x4_t red2[A8 ? NW : 1][A8 ? TILES : 1][64];
    __shared__ f32x4_t ssq_red[NW][4 * CT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int kblocks = p.K >> 5;
    int rb[TILES];
    if constexpr (MODE == GV_SWIGLU) {
        static_assert(TILES % 2 == 0, "SwiGLU: (gate, up) row-block pairs");
#pragma unroll
        for (int q = 0; q < TILES / 2; ++q) {                    // TILES / 2 output column blocks per workgroup, each a (gate, up) pair of row blocks
            const int i0 = ((int)blockIdx.x * (TILES / 2) + q) * 16;
            const int base = (i0 >> 5) * 64 + (i0 & 31);         // packed gate row (multiple of 16)
            rb[2 * q] = base >> 4;                               // gate row block
            rb[2 * q + 1] = (base + 32) >> 4;                    // matching up row block
        }
    } else {
#pragma unroll
        for (int t = 0; t < TILES; ++t) rb[t] = blockIdx.x * TILES + t;
    }
    // folded RMSNorm, consumer side: this thread's share of the sum-of-squares partials (L2 hits), requested before the weight stream
    // (three independent loads whose first USE is after the weight loop: summing them here would park the wave on an L2 round trip
    // before its first weight load -- measured +1.4 us on the 18 us qkv GEMV)
    constexpr int SSQ_LD = 3;                                   // 3 x NW x 64 float4 >= 4 x hidden / 16 up to hidden 6144
    f32x4_t ssq_ld[CT][SSQ_LD];
    const bool scaled = (MODE == GV_STORE16 || MODE == GV_SWIGLU || MODE == GV_F32) && p.ssq_in != nullptr;
    if (scaled) {
        const int n4 = p.nparts_in * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < SSQ_LD; ++q) ssq_ld[ct][q] = ((const f32x4_t*)(p.ssq_in + (size_t)ct * p.ssq_ts))[min(tid + q * NW * 64, n4 - 1)];   // chunk (= 4 batch columns) index & 3 is fixed per thread
    }
    // producer side: the old residual and gamma are requested up front as well (by every wave; wave 0 consumes them in the epilogue)
    f32x4_t r_old[CT], g_nx = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == GV_RESIDNORM) {
        const int n0p = blockIdx.x * 16 + kg * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) r_old[ct] = *(const f32x4_t*)(p.out + ((size_t)min(ct * 16 + l15, p.B - 1) * p.ldo + n0p) * 4);
        g_nx = *(const f32x4_t*)(p.gamma + n0p);
    }
    // x fragment (MFMA B operand: lane = batch column l15, 8 consecutive k): an MFMA tile has 16 batch columns; the lanes of the columns
    // >= B point outside the buffer descriptor, so they cost no request on the load path (with B = 8 half of every 1 KiB wave-load; the
    // activations are re-read by every workgroup and the per-CU load path, not HBM, is what the fp8 GEMVs run into).  Their zeros only
    // feed output columns that are never stored.
    const __amdgpu_buffer_rsrc_t xrs = gv_make_rsrc(p.x, (unsigned)(((size_t)(p.B - 1) * p.ldx + p.K) * 2));
    unsigned xvo[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) xvo[ct] = ct * 16 + l15 < p.B ? (unsigned)(((size_t)(ct * 16 + l15) * p.ldx + kg * 8) * 2) : 0x80000000u;
    auto xload = [&](size_t kb, int ct) -> typename T::v8 {
        return __builtin_bit_cast(typename T::v8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[ct] + (unsigned)(kb * 64), 0, 0));
    };
    f32x4_t acc[TILES][CT];
    f32x4_t acc2[A8 ? TILES : 1];                                // A8: the lo operand's accumulators
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[t][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (A8 ? TILES : 1); ++t) acc2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {
        // groups (64 columns) j_lo + w + NW * g, g = 0 .. : batch b holds g = b * PU .. b * PU + PU - 1.  Every batch but the last is complete
        // for every wave (the waves' group counts differ by at most one); in the last batch, entries past the end re-read the last valid
        // block (an L2 hit) against an all-zero x fragment, which adds exact zeros -- no predicated loads (predicated loads would make hipcc serialise the whole batch: a vmcnt(0) per load).
        const int kb_end = kblocks;
        const int j_end = (kb_end + 1) >> 1;                      // the last group may hold a single 32-block (16-bit weights; fp8 needs K % 64 == 0)
        constexpr int j_lo = 0;
        const int gpw = (j_end - j_lo + NW - 1) / NW;
        const int nb = (gpw + PU - 1) / PU;
        using wreg_t = typename std::conditional<W8, u32x4_t, typename T::v8>::type;
        constexpr int WH = W8 ? 1 : 2;                            // weight loads per group and row block
        const char* wp[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) wp[t] = p.W + ((size_t)rb[t] * (W8 ? (p.K >> 6) : kblocks)) * 1024 + lane * 16;
        constexpr int XH = X2 ? 1 : 2;                            // x loads per group
        const unsigned xvo2 = (l15 & 7) < p.B ? (unsigned)(((size_t)(l15 & 7) * p.ldx + (l15 >> 3) * 32 + kg * 8) * 2) : 0x80000000u;
        auto load = [&](wreg_t (&wf)[PU][WH][TILES], typename T::v8 (&xf)[PU][XH][CT], int b) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int g = j_lo + w + NW * (b * PU + u);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int kb = min(2 * g + h, kb_end - 1);
                    if (!W8 || h == 0) {
#pragma unroll
                        for (int t = 0; t < TILES; ++t)
                            wf[u][W8 ? 0 : h][t] = __builtin_nontemporal_load((const wreg_t*)(wp[t] + (size_t)(W8 ? min(g, j_end - 1) : kb) * 1024));
                    }
                    if constexpr (!X2) {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) xf[u][h][ct] = xload((size_t)kb, ct);
                    } else if (h == 0)
                        xf[u][0][0] = __builtin_bit_cast(typename T::v8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo2 + (unsigned)min(g, j_end - 1) * 128u, 0, 0));
                }
            }
        };
        auto mma = [&](wreg_t (&wf)[PU][WH][TILES], typename T::v8 (&xf)[PU][XH][CT], int b, bool last) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int g = j_lo + w + NW * (b * PU + u);
                typename T::v8 xv[2][CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xv[0][ct] = xf[u][0][ct];
                if constexpr (X2) {
                    const u32x4_t r = __builtin_bit_cast(u32x4_t, xf[u][0][0]);
                    u32x4_t q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r[e], 0x128, 0xF, 0xF, true);   // row_ror:8
                    xv[1][0] = __builtin_bit_cast(typename T::v8, q);
                } else {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) xv[1][ct] = xf[u][XH - 1][ct];
                }
                if (last) {
                    const typename T::v8 z = {};
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) xv[h][ct] = (2 * g + h >= kb_end) ? z : xv[h][ct];
                }
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if constexpr (A8) {
                            const long wq = (long)(((unsigned long)wf[u][0][t][2 * h + 1] << 32) | (unsigned long)wf[u][0][t][2 * h]);
                            const u32x4_t xr = __builtin_bit_cast(u32x4_t, xv[h][0]);
                            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wq, (long)(((unsigned long)xr[1] << 32) | (unsigned long)xr[0]), acc[t][0], 0, 0, 0);
                            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wq, (long)(((unsigned long)xr[3] << 32) | (unsigned long)xr[2]), acc2[t], 0, 0, 0);
                        } else {
                            typename T::v8 wv;
                            if constexpr (W8) wv = fp8x8_to_v8<T>(wf[u][0][t][2 * h], wf[u][0][t][2 * h + 1]);      // widened once, used by every column tile
                            else wv = wf[u][h][t];
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) acc[t][ct] = T::mfma16(wv, xv[h][ct], acc[t][ct]);
                        }
                    }
            }
        };
        wreg_t wa[PU][WH][TILES], wb[PU][WH][TILES];
        typename T::v8 xa[PU][XH][CT], xb[PU][XH][CT];
        load(wa, xa, 0);
        int i = 0;
        for (; i + 2 < nb; i += 2) {
            load(wb, xb, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(wa, xa, i, false);
            __builtin_amdgcn_sched_barrier(0);
            load(wa, xa, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(wb, xb, i + 1, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nb - i == 2) {
            load(wb, xb, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(wa, xa, i, false);
            mma(wb, xb, i + 1, true);
        } else {
            mma(wa, xa, i, true);
        }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) red[w][t * CT + ct][lane] = acc[t][ct];
    if constexpr (A8) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) red2[w][t][lane] = acc2[t];
    }
    if (scaled) {
      const int n4 = p.nparts_in * 4;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4_t ssq_acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < SSQ_LD; ++q)
            if (tid + q * NW * 64 < n4) ssq_acc += ssq_ld[ct][q];
        // lanes with equal (lane & 3) hold the same 4 batch columns (NW * 64 is a multiple of 4): fold the wave, lanes 48..51 publish.  DPP rotations
        // inside the rows of 16 and the gfx950 row / half swaps across them: no LDS round trips (four dependent ds_bpermute rounds sat on the
        // tail of every consumer launch).  Fixed association per lane: deterministic.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = ssq_acc[e];
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));      // row_ror:4
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));      // row_ror:8
            ssq_acc[e] = rows_sum_to_row3(v);
        }
        if (lane >= 48 && lane < 52) ssq_red[w][ct * 4 + lane - 48] = ssq_acc;
      }
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {                             // one pass of the 16-column epilogue per column tile
    f32x4_t tot[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        tot[t] = red[0][t * CT + ct][lane];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) tot[t] += red[ww][t * CT + ct][lane];
        if constexpr (A8) {                                       // y = s_hi (W hi) + s_lo (W lo), per batch column
            f32x4_t lo = red2[0][t][lane];
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) lo += red2[ww][t][lane];
            const float sh = l15 < p.B ? p.xscale[l15 * 2] : 0.f, sl = l15 < p.B ? p.xscale[l15 * 2 + 1] : 0.f;
            tot[t] = tot[t] * sh + lo * sl;
        }
    }
    if constexpr (W8) {                                           // per-row power-of-two scale: exact in fp32
#pragma unroll
        for (int t = 0; t < TILES; ++t) tot[t] *= *(const f32x4_t*)(p.wscale + rb[t] * 16 + kg * 4);
    }
    // lane holds D[n = kg*4 + r][b = ct*16 + l15]
    const int b = ct * 16 + l15;
    if (scaled) {
        float ss = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) ss += ssq_red[ww][ct * 4 + (l15 >> 2)][l15 & 3];       // fixed order: deterministic
        const float rstd = rsqrtf(ss * p.inv_h + p.eps);
#pragma unroll
        for (int t = 0; t < TILES; ++t) tot[t] *= rstd;
    }
    const int n0 = (MODE == GV_SWIGLU ? (int)blockIdx.x : rb[0]) * 16 + kg * 4;
    if constexpr (MODE == GV_RESIDNORM) {
        // r = resid + y (fp32, written back); xg = round16(r * gamma); sum of r^2 over this workgroup's 16 rows per batch column
        float sq = 0.f;
        if (b < p.B) {
            f32x4_t* rp = (f32x4_t*)(p.out + ((size_t)b * p.ldo + n0) * 4);
            const f32x4_t r = r_old[ct] + tot[0];
            *rp = r;
            const f32x4_t g = g_nx;
            *(u32x2_t*)(p.xg + ((size_t)b * p.ldo + n0) * 2) = pack4<T>(r[0] * g[0], r[1] * g[1], r[2] * g[2], r[3] * g[3]);
            sq = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
        }
        {   // sum over the four rows of 16 lanes (the 16 output rows of this workgroup) without LDS round trips
            sq = rows_sum_to_row3(sq);
        }
        if (kg == 3) p.ssq_out[(size_t)ct * p.ssq_ts + (size_t)blockIdx.x * 16 + l15] = sq;               // columns >= B carry 0
        continue;
    }
    if constexpr (MODE == GV_F32) {
        if (p.amax_val != nullptr) {                              // wave-uniform
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float v = tot[0][r]; if (n0 + r < p.N && v > bv) { bv = v; bi = n0 + r; } }      // ascending n: the first maximum wins, NaN never does
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (kg == 0) { p.amax_val[(size_t)ct * p.amax_ts + (size_t)blockIdx.x * 16 + l15] = bv; p.amax_idx[(size_t)ct * p.amax_ts + (size_t)blockIdx.x * 16 + l15] = bi; }
        }
    }
    if (b >= p.B) continue;
    if constexpr (MODE == GV_SWIGLU) {
#pragma unroll
        for (int q = 0; q < TILES / 2; ++q) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float g = tot[2 * q][r]; v[r] = g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.4426950408889634f)) * tot[2 * q + 1][r]; }   // same SiLU as gemm.hip
            *(u32x2_t*)(p.out + ((size_t)b * p.ldo + ((int)blockIdx.x * (TILES / 2) + q) * 16 + kg * 4) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
        }
    } else if constexpr (MODE == GV_STORE16) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            *(u32x2_t*)(p.out + ((size_t)b * p.ldo + n0 + t * 16) * 2) = pack4<T>(tot[t][0], tot[t][1], tot[t][2], tot[t][3]);
    } else if constexpr (MODE == GV_RESID) {
        f32x4_t* rp = (f32x4_t*)(p.out + ((size_t)b * p.ldo + n0) * 4);
        *rp = *rp + tot[0];
    } else {
        float* op = (float*)p.out + (size_t)b * p.ldo + n0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n0 + r < p.N) op[r] = tot[0][r];
    }
    }   // column tiles
}

// ---------------------------------------------------------------------------------------------
// Residual producers (o_proj, down_proj) at decode batches beyond 16: a 2-D decomposition (round 4).
// These matrices are narrow (N = hidden: 256 / 320 row blocks), so the 16-row kernel above puts ONE row block on a CU and every CU reads
// the whole activation operand -- at 32 sequences twice the lines of its weights (down_proj 7B: 32.0 us at B = 32, 53.0 at 64, against
// 19.2 at 16; LAB.md).  Here workgroup (grp, j) covers EIGHT row blocks (one per wave) x the K phase j of 8: wave p accumulates, for row
// block 8 grp + p, exactly the 64-column groups j, j + 8, j + 16, ... in that order -- what wave j of the 16-row kernel accumulates for
// that row block -- against x columns the whole workgroup shares through LDS (1/8 of x per workgroup instead of all of it).  The eight
// phase tiles of a row block go to global memory and gemv_k8_finish_kernel adds them in phase order (the 16-row kernel's LDS reduce over
// its waves 0..7) and runs the same epilogue: BITWISE the 16-row kernel's result, so a sequence is still independent of its batch.
// NWB row blocks (= waves) per workgroup: 8.
// ---------------------------------------------------------------------------------------------
template <typename T, bool W8, int CT, int NWB>
__global__ __launch_bounds__(NWB * 64) void gemv_k8_kernel(GemvArgs p, f32x4_t* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char xs[];    // [groups of this pass][B rows][128 B], 16-B chunks XORed with (row & 7)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int j = blockIdx.x & 7, rb = ((int)blockIdx.x >> 3) * NWB + w;
    const int kblocks = p.K >> 5, j_end = (kblocks + 1) >> 1;
    const int gpw = (j_end - j + 7) >> 3;                        // groups of phase j: j, j + 8, ...
    const int Brows = CT * 16;
    const int gpp = max(1, min(gpw, (int)(p.lds_bytes / (Brows * 128))));   // groups per pass
    using wreg_t = typename std::conditional<W8, u32x4_t, typename T::v8>::type;
    constexpr int WH = W8 ? 1 : 2;
    const char* wp = p.W + ((size_t)rb * (W8 ? (p.K >> 6) : kblocks)) * 1024 + lane * 16;
    f32x4_t acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PU = 4;                                        // groups per register buffer
    auto wload = [&](wreg_t (&wf)[PU][WH], int gi0) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int g = min(j + 8 * (gi0 + u), j_end - 1);
#pragma unroll
            for (int h = 0; h < WH; ++h)
                wf[u][h] = __builtin_nontemporal_load((const wreg_t*)(wp + (size_t)(W8 ? g : min(2 * g + h, kblocks - 1)) * 1024));
        }
    };
    for (int pass0 = 0; pass0 < gpw; pass0 += gpp) {
        const int ng = min(gpp, gpw - pass0);
        wreg_t wa[PU][WH], wb[PU][WH];
        wload(wa, pass0);                                        // the first weight batch is in flight while the x slice is staged
        __syncthreads();                                         // the previous pass has been read by every wave
        for (int c = tid; c < ng * Brows * 8; c += NWB * 64) {   // 16-byte chunks: (group, row, chunk)
            const int ch = c & 7, row = (c >> 3) % Brows, gi = (c >> 3) / Brows;
            const int g = j + 8 * (pass0 + gi), kb = 2 * g + (ch >> 2);
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (row < p.B && kb < kblocks) v = *(const u32x4_t*)(p.x + ((size_t)row * p.ldx + (size_t)g * 64 + ch * 8) * 2);
            *(u32x4_t*)(xs + ((size_t)(gi * Brows + row) * 128 + ((ch ^ (row & 7)) << 4))) = v;
        }
        __syncthreads();
        auto mma = [&](wreg_t (&wf)[PU][WH], int gi0) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int gi = gi0 + u - pass0;
                if (gi < ng) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        typename T::v8 wv;
                        if constexpr (W8) wv = fp8x8_to_v8<T>(wf[u][0][2 * h], wf[u][0][2 * h + 1]);
                        else wv = wf[u][h];
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) {
                            const int row = ct * 16 + l15;
                            const typename T::v8 xv = *(const typename T::v8*)(xs + ((size_t)(gi * Brows + row) * 128 + (((h * 4 + kg) ^ (row & 7)) << 4)));
                            acc[ct] = T::mfma16(wv, xv, acc[ct]);
                        }
                    }
                }
            }
        };
        int gi0 = pass0;
        for (; gi0 + 2 * PU < pass0 + ng; gi0 += 2 * PU) {
            wload(wb, gi0 + PU);
            __builtin_amdgcn_sched_barrier(0);
            mma(wa, gi0);
            __builtin_amdgcn_sched_barrier(0);
            wload(wa, gi0 + 2 * PU);
            __builtin_amdgcn_sched_barrier(0);
            mma(wb, gi0 + PU);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (gi0 + PU < pass0 + ng) {
            wload(wb, gi0 + PU);
            __builtin_amdgcn_sched_barrier(0);
            mma(wa, gi0);
            mma(wb, gi0 + PU);
        } else {
            mma(wa, gi0);
        }
    }
    f32x4_t* dst = part + (((size_t)rb * 8 + j) * CT) * 64 + lane;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) dst[ct * 64] = acc[ct];
}

// one wave per row block: the eight phase tiles in phase order, then the epilogue of gemv_mfma_kernel<GV_RESIDNORM> statement for statement
template <typename T, bool W8, int CT>
__global__ __launch_bounds__(64) void gemv_k8_finish_kernel(GemvArgs p, const f32x4_t* __restrict__ part) {
    const int lane = threadIdx.x, l15 = lane & 15, kg = lane >> 4;
    const int rb = blockIdx.x;
    const int n0 = rb * 16 + kg * 4;
    f32x4_t r_old[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) r_old[ct] = *(const f32x4_t*)(p.out + ((size_t)min(ct * 16 + l15, p.B - 1) * p.ldo + n0) * 4);
    const f32x4_t g_nx = *(const f32x4_t*)(p.gamma + n0);
    const f32x4_t* src = part + ((size_t)rb * 8 * CT) * 64 + lane;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x4_t tot = src[(0 * CT + ct) * 64];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) tot += src[(ww * CT + ct) * 64];
        if constexpr (W8) tot *= *(const f32x4_t*)(p.wscale + rb * 16 + kg * 4);
        const int b = ct * 16 + l15;
        float sq = 0.f;
        if (b < p.B) {
            f32x4_t* rp = (f32x4_t*)(p.out + ((size_t)b * p.ldo + n0) * 4);
            const f32x4_t r = r_old[ct] + tot;
            *rp = r;
            const f32x4_t g = g_nx;
            *(u32x2_t*)(p.xg + ((size_t)b * p.ldo + n0) * 2) = pack4<T>(r[0] * g[0], r[1] * g[1], r[2] * g[2], r[3] * g[3]);
            sq = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
        }
        sq = rows_sum_to_row3(sq);
        if (kg == 3) p.ssq_out[(size_t)ct * p.ssq_ts + (size_t)rb * 16 + l15] = sq;
    }
}

// Activation image of the fp8 x fp8 form: x [B][K] 16-bit -> x8 [B][K / 8][16 B] (8 hi codes, 8 lo codes per 8 consecutive k: the 16-bit
// operand's footprint, so the GEMV's x loads are unchanged) + scales [B][2] = (s_hi, s_lo), each 2^ceil(log2(amax / 448)) of what it scales
// (the rule of the weight quantiser, fp8.hip): hi = e4m3(x / s_hi), lo = e4m3((x - s_hi hi) / s_lo), round to nearest even.  One workgroup
// per sequence; x is read three times from L2 (amax, residual amax, codes).
__device__ __forceinline__ float pow2_scale_448(float m) {
    if (!(m > 0.f)) return 1.0f;
    int e;
    const float fr = frexpf(m / 448.0f, &e);
    return ldexpf(1.0f, fr == 0.5f ? e - 1 : e);
}

template <typename T>
__global__ __launch_bounds__(256) void quant_hilo_kernel(const typename T::elem* __restrict__ x, int ldx, int K, unsigned char* __restrict__ x8, float* __restrict__ scales) {
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const typename T::elem* xr = x + (size_t)b * ldx;
    auto block_max = [&](float v) -> float {
        v = wave_max(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    };
    float am = 0.f;
    for (int c = tid * 8; c < K; c += 2048) {
        const typename T::v8 v = *(const typename T::v8*)(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf((float)v[e]));
    }
    const float s_hi = pow2_scale_448(block_max(am)), inv_hi = 1.0f / s_hi;
    auto hi_codes = [&](const typename T::v8& v, unsigned (&code)[2], float (&resid)[8]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = (float)v[2 * q], c2 = (float)v[2 * q + 1];
            const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a * inv_hi, c2 * inv_hi, 0, false);
            const f32x2_t back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
            resid[2 * q] = a - back[0] * s_hi; resid[2 * q + 1] = c2 - back[1] * s_hi;     // exact: both terms on the grid of a's last bit or coarser
            if (q & 1) code[q >> 1] |= (unsigned)(pk & 0xffff) << 16; else code[q >> 1] = (unsigned)(pk & 0xffff);
        }
    };
    float am2 = 0.f;
    for (int c = tid * 8; c < K; c += 2048) {
        const typename T::v8 v = *(const typename T::v8*)(xr + c);
        unsigned code[2]; float r[8];
        hi_codes(v, code, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) am2 = fmaxf(am2, fabsf(r[e]));
    }
    const float s_lo = pow2_scale_448(block_max(am2)), inv_lo = 1.0f / s_lo;
    for (int c = tid * 8; c < K; c += 2048) {
        const typename T::v8 v = *(const typename T::v8*)(xr + c);
        unsigned hi[2], lo[2]; float r[8];
        hi_codes(v, hi, r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(r[2 * q] * inv_lo, r[2 * q + 1] * inv_lo, 0, false);
            if (q & 1) lo[q >> 1] |= (unsigned)(pk & 0xffff) << 16; else lo[q >> 1] = (unsigned)(pk & 0xffff);
        }
        *(u32x4_t*)(x8 + ((size_t)b * K + c) * 2) = u32x4_t{hi[0], hi[1], lo[0], lo[1]};
    }
    if (tid == 0) { scales[b * 2] = s_hi; scales[b * 2 + 1] = s_lo; }
}

// decode: resid[b] = embed[tok[b]] plus the producer side of the folded RMSNorm (see GemvArgs): xg = round16(resid * gamma of layer 0's
// input norm), ssq[0][b] = sum resid^2 (one partial per sequence).
template <typename T>
__global__ __launch_bounds__(256) void embed_tok_norm_kernel(const int* __restrict__ tok, const typename T::elem* __restrict__ embed, float* __restrict__ resid,
                                                             const float* __restrict__ gamma, typename T::elem* __restrict__ xg, float* __restrict__ ssq, int H, int ssq_ts) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const typename T::elem* p = embed + (size_t)tok[b] * H;
    float ss = 0.f;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
        const typename T::v8 v = *(const typename T::v8*)(p + c);
        f32x4_t a, bb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (float)v[e]; bb[e] = (float)v[4 + e]; }
        *(f32x4_t*)(resid + (size_t)b * H + c) = a;
        *(f32x4_t*)(resid + (size_t)b * H + c + 4) = bb;
        const f32x4_t g0 = *(const f32x4_t*)(gamma + c), g1 = *(const f32x4_t*)(gamma + c + 4);
        *(u32x2_t*)((char*)xg + ((size_t)b * H + c) * 2) = pack4<T>(a[0] * g0[0], a[1] * g0[1], a[2] * g0[2], a[3] * g0[3]);
        *(u32x2_t*)((char*)xg + ((size_t)b * H + c + 4) * 2) = pack4<T>(bb[0] * g1[0], bb[1] * g1[1], bb[2] * g1[2], bb[3] * g1[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += a[e] * a[e] + bb[e] * bb[e];
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) ssq[(size_t)(b >> 4) * ssq_ts + (b & 15)] = (red[0] + red[1]) + (red[2] + red[3]);     // tile-major [B / 16][...][16]
}

// Producer side of the folded RMSNorm for rows that already sit in the fp32 residual (the last prompt position of every sequence
// before lm_head): xg = round16(resid * gamma), ssq[b] = sum resid^2.  One workgroup per row.
template <typename T>
__global__ __launch_bounds__(256) void resid_norm_prep_kernel(const float* __restrict__ resid, const float* __restrict__ gamma, typename T::elem* __restrict__ xg,
                                                              float* __restrict__ ssq, int H, int ssq_ts) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float ss = 0.f;
    for (int c = threadIdx.x * 4; c < H; c += 256 * 4) {
        const f32x4_t r = *(const f32x4_t*)(resid + (size_t)b * H + c), g = *(const f32x4_t*)(gamma + c);
        *(u32x2_t*)((char*)xg + ((size_t)b * H + c) * 2) = pack4<T>(r[0] * g[0], r[1] * g[1], r[2] * g[2], r[3] * g[3]);
        ss += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) ssq[(size_t)(b >> 4) * ssq_ts + (b & 15)] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------
// decode attention: one workgroup of 8 waves per (sequence, head); lane = (key slot 0..3, 16-B d chunk 0..15), so a
// wave-load covers 4 consecutive cache rows = 1 KiB contiguous.  Each lane keeps TWO independent online-softmax states
// (keys k and k+32 of every 64-key round) with DEPTH rounds of loads in flight.
// The kernel is latency-bound (60-76 MB of KV per launch over 256 workgroups, one per CU), so the fixed parts are kept off the
// critical path: every lane builds its own slice of the rotated query straight from the qkv buffer (no LDS round trip, no
// barrier before the key loop); the fresh token's k/v are appended to the cache by wave 0 on the side and enter the softmax from
// registers (the loop only streams keys [0, pos)), so nothing waits for that store; partial states are merged inside each wave
// with shuffles before 8 (not 64) states meet in LDS.
// ---------------------------------------------------------------------------------------------
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void decode_attn_kernel(const typename T::elem* __restrict__ qkv, const int* __restrict__ pos_arr,
                                                          const float2* __restrict__ rope, typename T::elem* __restrict__ Kc,
                                                          typename T::elem* __restrict__ Vc, typename T::elem* __restrict__ out, int H, int heads,
                                                          int max_seq, float scale_log2e) {
    __shared__ float st_m[NW], st_l[NW];
    __shared__ float st_o[NW][HD];
    constexpr int KPR = NW * 8;                        // keys per round: NW waves x 4 slots x 2 states
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const typename T::elem* q = qkv + (size_t)b * 3 * H + h * HD;
    const typename T::elem* k = q + H;
    const typename T::elem* v = q + 2 * H;
    typename T::elem* kcache = Kc + ((size_t)b * heads + h) * max_seq * HD;
    typename T::elem* vcache = Vc + ((size_t)b * heads + h) * max_seq * HD;
    const int slot = lane >> 4, dc = lane & 15;

    // The first DEPTH rounds of K / V loads go out before anything else -- before `pos` has even arrived: their addresses depend only on
    // kernel arguments (rows up to max_seq - 1 exist; rows >= pos hold stale or unwritten data and are masked in update()), so the stream
    // starts without the L2 round trip of the position load in front of it, and the query's RoPE below -- two more round trips for the rope
    // table and q -- runs while they are in flight.  The cache is streamed once per token (2.4 GB per token step at 8 sequences: no reuse
    // in L2 / Infinity Cache), hence non-temporal loads like the GEMVs' weights.
    const int key0 = w * 4 + slot;
    // DEPTH rounds of KPR keys are kept in flight per workgroup (DEPTH x 4 x 16-B loads per lane)
    constexpr int DEPTH = NW == 8 ? 4 : 2;
    typename T::v8 kq[DEPTH][2], vq[DEPTH][2];
    auto load_row = [&](int kc, typename T::v8& kk_, typename T::v8& vv_) {
        kk_ = __builtin_nontemporal_load((const typename T::v8*)(kcache + (size_t)kc * HD + dc * 8));
        vv_ = __builtin_nontemporal_load((const typename T::v8*)(vcache + (size_t)kc * HD + dc * 8));
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        load_row(min(key0 + KPR * d, max_seq - 1), kq[d][0], vq[d][0]);
        load_row(min(key0 + KPR * d + KPR / 2, max_seq - 1), kq[d][1], vq[d][1]);
    }
    const typename T::v8 q_own = *(const typename T::v8*)(q + dc * 8), q_oth = *(const typename T::v8*)(q + (dc ^ 8) * 8);     // needs no position either
    __builtin_amdgcn_sched_barrier(0);                 // keeps the (scalar) position load and everything that hangs off it behind the loads above
    const int pos = pos_arr[b];
    const int n_keys = pos;                            // cached keys; the fresh key (index pos) is handled from registers below
    auto load = [&](int key, typename T::v8& kk_, typename T::v8& vv_) { load_row(max(0, min(key, n_keys - 1)), kk_, vv_); };   // clamped rows are masked below

    // rotate-half RoPE on this lane's 8 dims d = dc*8 + e of a 128-wide row x: d < 64: x[d] c[d] - x[d+64] s[d];  d >= 64: x[d] c[d-64] + x[d-64] s[d-64]
    // (values rounded to the activation dtype like the prefill path writes them)
    const int j0 = (dc & 7) * 8;                       // rope index of e = 0
    float cs_c[8], cs_s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float2 cs = rope[(size_t)pos * 64 + j0 + e]; cs_c[e] = cs.x; cs_s[e] = dc < 8 ? -cs.y : cs.y; }
    auto rotate = [&](const typename T::v8& own, const typename T::v8& oth, float (&r)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (float)T::from_f32((float)own[e] * cs_c[e] + (float)oth[e] * cs_s[e]);
    };
    auto rotated = [&](const typename T::elem* x, float (&r)[8]) {
        rotate(*(const typename T::v8*)(x + dc * 8), *(const typename T::v8*)(x + (dc ^ 8) * 8), r);
    };
    float qr[8];
    rotate(q_own, q_oth, qr);
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] *= scale_log2e;

    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f}, o[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[u][e] = 0.f;
    auto update = [&](int u, const typename T::v8& kf, const typename T::v8& vf, bool valid) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qr[e] * (float)kf[e];
        s = row16_sum(s);
        if (valid) {
            const float mn = fmaxf(m[u], s);
            const float alpha = __builtin_amdgcn_exp2f(m[u] - mn), pv = __builtin_amdgcn_exp2f(s - mn);     // arguments <= 0: the raw v_exp_f32 is exact enough and flushes to 0
            m[u] = mn;
            l[u] = l[u] * alpha + pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[u][e] = o[u][e] * alpha + pv * (float)vf[e];
        }
    };

    // fresh token (overlaps the first loads): wave 0 appends the rotated k and v to the cache; its slot-0 lanes also keep them for the softmax
    typename T::v8 knew, vnew;
    if (w == 0) {
        float kr[8];
        rotated(k, kr);
#pragma unroll
        for (int e = 0; e < 8; ++e) knew[e] = T::from_f32(kr[e]);
        vnew = *(const typename T::v8*)(v + dc * 8);
        if (slot == 0) {
            *(typename T::v8*)(kcache + (size_t)pos * HD + dc * 8) = knew;
            *(typename T::v8*)(vcache + (size_t)pos * HD + dc * 8) = vnew;
        }
    }

    for (int base = key0; base < n_keys; base += KPR * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int u = 0; u < 2; ++u) update(u, kq[d][u], vq[d][u], base + KPR * d + (KPR / 2) * u < n_keys);
            // refill this ring slot with the round DEPTH ahead (clamped loads past the end are harmless and masked)
            load(base + KPR * (d + DEPTH), kq[d][0], vq[d][0]);
            load(base + KPR * (d + DEPTH) + KPR / 2, kq[d][1], vq[d][1]);
        }
    }
    if (w == 0) update(0, knew, vnew, slot == 0);      // the fresh key, once (wave 0, slot 0)

    // merge: the lane's two states, then the four slots of the wave (lanes with equal dc), then the eight waves through LDS
    auto combine = [&](float& ma, float& la, float (&oa)[8], float mb, float lb, const float (&ob)[8]) {
        const float mn = fmaxf(ma, mb);
        const float fa = exp2f(ma - mn), fb = exp2f(mb - mn);
        ma = mn; la = la * fa + lb * fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) oa[e] = oa[e] * fa + ob[e] * fb;
    };
    combine(m[0], l[0], o[0], m[1], l[1], o[1]);
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        float ob[8];
        const float mb = __shfl_xor(m[0], sh, 64), lb = __shfl_xor(l[0], sh, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = __shfl_xor(o[0][e], sh, 64);
        combine(m[0], l[0], o[0], mb, lb, ob);
    }
    if (slot == 0) {
        if (dc == 0) { st_m[w] = m[0]; st_l[w] = l[0]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) st_o[w][dc * 8 + e] = o[0][e];
    }
    __syncthreads();
    if (tid < HD) {
        float M = st_m[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) M = fmaxf(M, st_m[i]);
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float f = exp2f(st_m[i] - M);
            L += st_l[i] * f;
            acc += st_o[i][tid] * f;
        }
        out[(size_t)b * H + h * HD + tid] = T::from_f32(acc / L);
    }
}

// ---------------------------------------------------------------------------------------------
// decode attention, context-split variant (round 4): SPLIT workgroups per (sequence, head), for launches whose (sequence, head) units do
// not fill the chip evenly -- 13B at 8 sequences: 320 units on 256 CUs = 1 1/4 rounds (21.8 us per layer against ~15 balanced); one
// sequence: 32 / 40 units on 256 CUs.  The context is dealt out in GROUPS OF 16 KEYS round-robin (split sp owns groups sp, sp + SPLIT, ...),
// so every load address depends only on kernel arguments and the block index -- the stream starts before `pos` has arrived, like the
// unsplit kernel -- and the splits' key counts differ by at most 16.
// Cross-workgroup merge WITHOUT fences: round 3 built the same split with an agent-scope release / acquire around an atomic ticket and
// measured 61 us instead of 18.8 -- on an 8-XCD part that pair is `buffer_wbl2 sc1` + `buffer_inv sc1`, a walk of the XCD's whole L2.
// Here every byte that crosses workgroups is itself moved by AGENT-SCOPE RELAXED ATOMICS (global_store / global_load with sc1: written
// through to / read from the memory side, which is coherent across the XCDs), ordered by `s_waitcnt vmcnt(0)` + the workgroup barrier
// before the ticket and by the control dependency on the ticket's return value after it: no cache maintenance at all.  The last arriver
// merges the SPLIT partial states in split order (the result does not depend on who arrives last: deterministic) and re-arms the ticket.
// ---------------------------------------------------------------------------------------------
constexpr int DSPLIT_MAX = 8;
constexpr int DPART = HD + 2;                            // floats per partial state: o[128] (unnormalised, relative to m), m, l

template <typename T, int SPLIT>
__global__ __launch_bounds__(512) void decode_attn_split_kernel(const typename T::elem* __restrict__ qkv, const int* __restrict__ pos_arr,
                                                                const float2* __restrict__ rope, typename T::elem* __restrict__ Kc,
                                                                typename T::elem* __restrict__ Vc, typename T::elem* __restrict__ out, int H, int heads,
                                                                int max_seq, float scale_log2e, float* __restrict__ part, unsigned* __restrict__ ticket) {
    constexpr int NW = 8, DEPTH = 2;
    __shared__ float st_m[NW], st_l[NW];
    __shared__ float st_o[NW][HD];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y, sp = blockIdx.z;
    const typename T::elem* q = qkv + (size_t)b * 3 * H + h * HD;
    const typename T::elem* k = q + H;
    const typename T::elem* v = q + 2 * H;
    typename T::elem* kcache = Kc + ((size_t)b * heads + h) * max_seq * HD;
    typename T::elem* vcache = Vc + ((size_t)b * heads + h) * max_seq * HD;
    const int slot = lane >> 4, dc = lane & 15;
    // key of (workgroup round r, state u) for this lane: 16 (SPLIT (4 r + w / 4 + 2 u) + sp) + (w % 4) * 4 + slot
    const int kin = (w & 3) * 4 + slot, jw = w >> 2;
    auto key_of = [&](int r, int u) { return 16 * (SPLIT * (4 * r + jw + 2 * u) + sp) + kin; };
    typename T::v8 kq[DEPTH][2], vq[DEPTH][2];
    auto load_row = [&](int kc, typename T::v8& kk_, typename T::v8& vv_) {
        kk_ = __builtin_nontemporal_load((const typename T::v8*)(kcache + (size_t)kc * HD + dc * 8));
        vv_ = __builtin_nontemporal_load((const typename T::v8*)(vcache + (size_t)kc * HD + dc * 8));
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int u = 0; u < 2; ++u) load_row(min(key_of(d, u), max_seq - 1), kq[d][u], vq[d][u]);
    const typename T::v8 q_own = *(const typename T::v8*)(q + dc * 8), q_oth = *(const typename T::v8*)(q + (dc ^ 8) * 8);
    __builtin_amdgcn_sched_barrier(0);
    const int pos = pos_arr[b];
    const int n_keys = pos;
    const int j0 = (dc & 7) * 8;
    float cs_c[8], cs_s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float2 cs = rope[(size_t)pos * 64 + j0 + e]; cs_c[e] = cs.x; cs_s[e] = dc < 8 ? -cs.y : cs.y; }
    auto rotate = [&](const typename T::v8& own, const typename T::v8& oth, float (&r)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (float)T::from_f32((float)own[e] * cs_c[e] + (float)oth[e] * cs_s[e]);
    };
    float qr[8];
    rotate(q_own, q_oth, qr);
#pragma unroll
    for (int e = 0; e < 8; ++e) qr[e] *= scale_log2e;
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f}, o[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[u][e] = 0.f;
    auto update = [&](int u, const typename T::v8& kf, const typename T::v8& vf, bool valid) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qr[e] * (float)kf[e];
        s = row16_sum(s);
        if (valid) {
            const float mn = fmaxf(m[u], s);
            const float alpha = __builtin_amdgcn_exp2f(m[u] - mn), pv = __builtin_amdgcn_exp2f(s - mn);
            m[u] = mn;
            l[u] = l[u] * alpha + pv;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[u][e] = o[u][e] * alpha + pv * (float)vf[e];
        }
    };
    // fresh token: split 0, wave 0 appends the rotated k and v to the cache and feeds them to the softmax from registers
    typename T::v8 knew, vnew;
    const bool fresh = (sp == 0 && w == 0);
    if (fresh) {
        float kr[8];
        rotate(*(const typename T::v8*)(k + dc * 8), *(const typename T::v8*)(k + (dc ^ 8) * 8), kr);
#pragma unroll
        for (int e = 0; e < 8; ++e) knew[e] = T::from_f32(kr[e]);
        vnew = *(const typename T::v8*)(v + dc * 8);
        if (slot == 0) {
            *(typename T::v8*)(kcache + (size_t)pos * HD + dc * 8) = knew;
            *(typename T::v8*)(vcache + (size_t)pos * HD + dc * 8) = vnew;
        }
    }
    for (int r0 = 0; 16 * (SPLIT * 4 * r0 + sp) < n_keys; r0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int u = 0; u < 2; ++u) update(u, kq[d][u], vq[d][u], key_of(r0 + d, u) < n_keys);
#pragma unroll
            for (int u = 0; u < 2; ++u) load_row(max(0, min(key_of(r0 + d + DEPTH, u), n_keys - 1)), kq[d][u], vq[d][u]);     // clamped rows are masked
        }
    }
    if (fresh) update(0, knew, vnew, slot == 0);
    auto combine = [&](float& ma, float& la, float (&oa)[8], float mb, float lb, const float (&ob)[8]) {
        const float mn = fmaxf(ma, mb);
        const float fa = exp2f(ma - mn), fb = exp2f(mb - mn);
        ma = mn; la = la * fa + lb * fb;
#pragma unroll
        for (int e = 0; e < 8; ++e) oa[e] = oa[e] * fa + ob[e] * fb;
    };
    combine(m[0], l[0], o[0], m[1], l[1], o[1]);
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
        float ob[8];
        const float mb = __shfl_xor(m[0], sh, 64), lb = __shfl_xor(l[0], sh, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = __shfl_xor(o[0][e], sh, 64);
        combine(m[0], l[0], o[0], mb, lb, ob);
    }
    if (slot == 0) {
        if (dc == 0) { st_m[w] = m[0]; st_l[w] = l[0]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) st_o[w][dc * 8 + e] = o[0][e];
    }
    __syncthreads();
    // this workgroup's state (M, L, acc[tid]) in threads tid < HD
    float M = -1e30f, L = 0.f, acc = 0.f;
    if (tid < HD) {
        M = st_m[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) M = fmaxf(M, st_m[i]);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const float f = exp2f(st_m[i] - M);
            L += st_l[i] * f;
            acc += st_o[i][tid] * f;
        }
    }
    const int unit = b * heads + h;
    float* mine = part + ((size_t)unit * SPLIT + sp) * DPART;
    if (tid < HD) {
        __hip_atomic_store(mine + tid, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(mine + HD, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + HD + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's write-through stores have been acknowledged ...
    __syncthreads();                                         // ... and so have everyone's in this workgroup
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ticket + unit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != SPLIT - 1) return;                       // not the last arriver
    if (tid == 0) __hip_atomic_store(ticket + unit, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-arm (the next launch is ordered by the kernel boundary)
    if (tid < HD) {
        const float* base = part + (size_t)unit * SPLIT * DPART;
        float ms[SPLIT], ls[SPLIT], os[SPLIT];
#pragma unroll
        for (int s_ = 0; s_ < SPLIT; ++s_) {
            os[s_] = __hip_atomic_load(base + s_ * DPART + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ms[s_] = __hip_atomic_load(base + s_ * DPART + HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ls[s_] = __hip_atomic_load(base + s_ * DPART + HD + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        float MM = ms[0];
#pragma unroll
        for (int s_ = 1; s_ < SPLIT; ++s_) MM = fmaxf(MM, ms[s_]);
        float LL = 0.f, A = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < SPLIT; ++s_) {                 // fixed order: independent of the arrival order
            const float f = exp2f(ms[s_] - MM);
            LL += ls[s_] * f;
            A += os[s_] * f;
        }
        out[(size_t)b * H + h * HD + tid] = T::from_f32(A / LL);
    }
}

// ---------------------------------------------------------------------------------------------
// greedy pick (first index wins ties, like torch.argmax on CPU) + bookkeeping: advance bit 0 does pos[b]++, bit 1 does
// hist[b][step[b]++] = token and EOS stickiness.
// ---------------------------------------------------------------------------------------------
// greedy pick from the lm_head GEMV's per-workgroup candidates (GemvArgs::amax_*): same result as a scan of the full logits
// (largest value, smallest index on ties, NaN never), 16x fewer values to scan.
__global__ __launch_bounds__(256) void argmax_parts_kernel(const float* __restrict__ val, const int* __restrict__ idx, int nblk, int amax_ts, int V, int* __restrict__ next,
                                                           int* __restrict__ pos, int* __restrict__ step, int* __restrict__ hist, int hist_stride,
                                                           int* __restrict__ done, int eos, int advance) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    val += (size_t)(b >> 4) * amax_ts; idx += (size_t)(b >> 4) * amax_ts;          // tile-major [B / 16][nblk][16]
    for (int i = tid; i < nblk; i += 256) {
        const float v = val[(size_t)i * 16 + (b & 15)];
        const int j = idx[(size_t)i * 16 + (b & 15)];
        if (v > best || (v == best && j < bi)) { best = v; bi = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; ++i)
            if (sv[i] > best || (sv[i] == best && si[i] < bi)) { best = sv[i]; bi = si[i]; }
        int tok = bi < V ? bi : 0;
        // flags (llm.hip AM_*): 1 = advance the position, 2 = record (token history + EOS stickiness).  A plain pgv_llm_decode step passes 1
        // only: it must neither append to the history nor look at a `done` flag a previous decode_greedy / decode_sample run left set
        // (tok = eos = -1 would index embed[-H] on the next step).
        if (advance & 2) {
            if (done[b]) tok = eos;
            else if (eos >= 0 && tok == eos) done[b] = 1;
            hist[(size_t)b * hist_stride + step[b]] = tok;
            step[b] += 1;
        }
        if (advance & 1) pos[b] += 1;
        next[b] = tok;
    }
}

// ---------------------------------------------------------------------------------------------
// Sampling pick (the reference's default decode mode, video_chatgpt/inference.py:106-112: do_sample=True, temperature=0.2; HF's
// sample loop = logits / temperature -> top-k mask (k = 50 from the default GenerationConfig; `scores < kth` keeps ties) -> softmax ->
// multinomial).  The multinomial draw is an inverse-CDF pick with a caller-supplied uniform u: the first vocabulary index whose
// cumulative (unnormalised) weight exceeds u * total.  One workgroup of 16 waves per sequence; wave w owns the contiguous index range
// [w * seg, (w + 1) * seg) so the cumulative sum can be walked hierarchically in vocabulary order: 16 wave totals -> the rounds of 64
// inside the selected wave -> an inclusive scan of the selected round.  Everything is a fixed-order reduction: same logits + same u ->
// same token, on any launch.  The k-th largest logit comes from a 4-pass radix select on order-preserving integer keys (LDS histogram).
// The logits (128 KB per sequence) are re-read from L2 per pass instead of living in registers.
// ---------------------------------------------------------------------------------------------
struct SampleArgs {
    const float* logits; int V;
    float c;                 // log2(e) / temperature
    int top_k;
    const float* u; int u_stride; int u_by_step;     // uniform of sequence b: u[(u_by_step ? step[b] : 0) * u_stride + b]
    int* next; int* pos; int* step; int* hist; int hist_stride; int* done; int eos; int advance;
};

__device__ __forceinline__ unsigned float_key(float x) {          // larger float <-> larger key; NaN -> 0 (below every number)
    if (x != x) return 0u;
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int SAMPLE_MAXR = 64;     // rounds of 64 per wave: V <= 16 * 64 * 64 = 65536

__global__ __launch_bounds__(1024) void sample_kernel(SampleArgs p) {
    __shared__ int hist[256];
    __shared__ float R[16][SAMPLE_MAXR];
    __shared__ float Wt[16];
    __shared__ float red[16];
    __shared__ int sel_i[4];        // chosen bin / wave / round / pick-last flag
    __shared__ float sel_f[1];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int V = p.V;
    const float* lg = p.logits + (size_t)b * V;
    const int seg = (((V + 15) / 16) + 63) / 64 * 64, rounds = seg / 64;
    const int base = w * seg;
    // ---- pass 1: maximum over the finite entries --------------------------------------------------
    float mx = -INFINITY;
    for (int r = 0; r < rounds; ++r) {
        const int i = base + r * 64 + lane;
        const float x = i < V ? lg[i] : -INFINITY;
        mx = fmaxf(mx, x);                       // fmaxf drops NaN operands
    }
    mx = wave_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    float M = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) M = fmaxf(M, red[i]);
    // ---- k-th largest key (radix select, most significant byte first) -----------------------------------
    unsigned thr = 0u;                           // keep keys >= thr
    if (p.top_k > 0 && p.top_k < V) {
        unsigned prefix = 0u, mask = 0u;
        int k = p.top_k;
        for (int pass = 3; pass >= 0; --pass) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int r = 0; r < rounds; ++r) {
                const int i = base + r * 64 + lane;
                if (i < V) {
                    const unsigned key = float_key(lg[i]);
                    if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                int cum = 0, bin = 0;
                for (int q = 255; q >= 0; --q) {
                    const int h = hist[q];
                    if (cum + h >= k) { bin = q; k -= cum; break; }
                    cum += h;
                }
                sel_i[0] = bin; sel_i[1] = k;
            }
            __syncthreads();
            prefix |= (unsigned)sel_i[0] << (8 * pass);
            mask |= 0xffu << (8 * pass);
            k = sel_i[1];
            __syncthreads();
        }
        thr = prefix;
    }
    // ---- weights e_i = exp2((x_i - M) c) of the kept entries; round and wave totals -------------------------
    auto weight = [&](int i) -> float {
        if (i >= V) return 0.f;
        const float x = lg[i];
        const unsigned key = float_key(x);
        return (key >= thr && key != 0u) ? __builtin_amdgcn_exp2f((x - M) * p.c) : 0.f;
    };
    for (int r = 0; r < rounds; ++r) {
        const float e = weight(base + r * 64 + lane);
        const float t = wave_sum(e);
        if (lane == 0) R[w][r] = t;
    }
    __syncthreads();
    if (tid < 16) {
        float t = 0.f;
        for (int r = 0; r < rounds; ++r) t += R[tid][r];
        Wt[tid] = t;
    }
    __syncthreads();
    if (tid == 0) {
        float S = 0.f;
        for (int i = 0; i < 16; ++i) S += Wt[i];
        const float uu = p.u[(size_t)(p.u_by_step ? p.step[b] : 0) * p.u_stride + b];
        float T = uu * S;
        int ws = -1, last = 0; float acc = 0.f;
        for (int i = 0; i < 16; ++i) {
            if (Wt[i] > 0.f) { last = i; if (T < acc + Wt[i]) { ws = i; break; } acc += Wt[i]; }
        }
        int pick_last = 0;
        if (ws < 0) { ws = last; pick_last = 1; }            // u * S rounded past the total: the last kept entry
        T -= acc;
        int rs = -1, lastr = 0; acc = 0.f;
        for (int r = 0; r < rounds; ++r) {
            const float t = R[ws][r];
            if (t > 0.f) { lastr = r; if (!pick_last && T < acc + t) { rs = r; break; } acc += t; }
        }
        if (rs < 0) { rs = lastr; pick_last = 1; }
        sel_i[1] = ws; sel_i[2] = rs; sel_i[3] = pick_last; sel_f[0] = T - acc;
    }
    __syncthreads();
    if (w != sel_i[1]) return;
    const int i0 = base + sel_i[2] * 64;
    const float e = weight(i0 + lane);
    float cs = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(cs, o, 64);
        if (lane >= o) cs += t;
    }
    const unsigned long long hit = __ballot(e > 0.f && cs > sel_f[0]);
    const unsigned long long any = __ballot(e > 0.f);
    int sl = 0;
    if (hit != 0ull && !sel_i[3]) sl = __builtin_ctzll(hit);
    else if (any != 0ull) sl = 63 - __builtin_clzll(any);
    if (lane == 0) {
        int tok = (any != 0ull) ? i0 + sl : 0;               // all-NaN / empty rows: token 0 rather than an out-of-range id
        if (p.advance & 2) {                                  // AM_RECORD (see argmax_parts_kernel)
            if (p.done[b]) tok = p.eos;
            else if (p.eos >= 0 && tok == p.eos) p.done[b] = 1;
            p.hist[(size_t)b * p.hist_stride + p.step[b]] = tok;
            p.step[b] += 1;
        }
        if (p.advance & 1) p.pos[b] += 1;
        p.next[b] = tok;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int pgv_launch_embed_splice(int dtype, const int* row_src, const void* embed, const void* video, float* resid, int M, int H, hipStream_t s) {
    PGV_CHECK(H % 8 == 0, "embed: hidden must be a multiple of 8");
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((embed_splice_kernel<T>), dim3(M), dim3(256), 0, s, row_src, (const typename T::elem*)embed,
                                                    (const typename T::elem*)video, resid, H));
    return PGV_OK;
}
int pgv_launch_gather_rows(const float* src, const int* rows, float* dst, int B, int H, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, s, src, rows, dst, H);
    return PGV_OK;
}
int pgv_launch_rope_kv_write(int dtype, void* qkv, const int* row_b, const int* row_pos, const void* rope, void* Kc, void* Vc, int M, int H,
                             int heads, int max_seq, hipStream_t s) {
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((rope_kv_write_kernel<T>), dim3(M), dim3(256), 0, s, (typename T::elem*)qkv, row_b, row_pos,
                                                    (const float2*)rope, (typename T::elem*)Kc, (typename T::elem*)Vc, H, heads, max_seq));
    return PGV_OK;
}
int pgv_launch_prefill_attn(pgv_ctx* ctx, int dtype, const void* qkv, void* out, const void* Kc, const void* Vc, const int* cu, int B, int max_len,
                            int H, int heads, int max_seq, double flops, hipStream_t s) {
    PrefillAttnArgs a;
    a.qkv = (const char*)qkv; a.out = (char*)out; a.Kc = (const char*)Kc; a.Vc = (const char*)Vc; a.cu = cu;
    a.H = H; a.heads = heads; a.max_seq = max_seq;
    a.scale_log2e = 0.08838834764831845f * 1.4426950408889634f;    // 128^-0.5 * log2(e)
    dim3 grid((max_len + 127) / 128, heads, B);
    pgv_prof_begin(ctx, 2, s);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((prefill_attn_kernel<T>), grid, dim3(256), 0, s, a));
    pgv_prof_end(ctx, 2, s, flops, 0.0);
    return PGV_OK;
}

int pgv_launch_final_prep(int dtype, const float* resid, const float* gamma, void* xg, float* ssq, int B, int H, hipStream_t s) {
    PGV_CHECK(H % 4 == 0, "final_prep: hidden %d unsupported", H);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((resid_norm_prep_kernel<T>), dim3(B), dim3(256), 0, s, resid, gamma, (typename T::elem*)xg, ssq, H, H));
    return PGV_OK;
}

int pgv_launch_embed_tok_norm(int dtype, const int* tok, const void* embed, float* resid, const float* gamma, void* xg, float* ssq, int B, int H, hipStream_t s) {
    PGV_CHECK(H % 8 == 0, "embed: hidden must be a multiple of 8");
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((embed_tok_norm_kernel<T>), dim3(B), dim3(256), 0, s, tok, (const typename T::elem*)embed, resid, gamma,
                                                    (typename T::elem*)xg, ssq, H, H));
    return PGV_OK;
}

int pgv_launch_quant_hilo(int dtype, const void* x, int ldx, int K, int B, void* x8, float* scales, hipStream_t s) {
    PGV_CHECK(K % 8 == 0 && B >= 1, "quant_hilo: K=%d must be a multiple of 8", K);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((quant_hilo_kernel<T>), dim3(B), dim3(256), 0, s, (const typename T::elem*)x, ldx, K, (unsigned char*)x8, scales));
    return PGV_OK;
}

// The 8-phase form of a residual producer (gemv_k8_kernel + gemv_k8_finish_kernel) with CT column tiles; `a` is complete except lds_bytes.
template <bool W8, int CT>
static int launch_k8(int dtype, GemvArgs a, int grid, void* k8_part, hipStream_t s) {
    const int gpw_max = ((a.K / 32 + 1) / 2 + 7) / 8;
    const unsigned budget = 96u * 1024u, per_group = (unsigned)CT * 16u * 128u;
    unsigned lds = (unsigned)gpw_max * per_group;
    if (lds > budget) lds = budget / per_group * per_group;
    a.lds_bytes = lds;
    f32x4_t* part = (f32x4_t*)k8_part;
    static bool cfg_done = false;
    if (!cfg_done) {
        PGV_HIP(hipFuncSetAttribute((const void*)gemv_k8_kernel<TF16, W8, CT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        PGV_HIP(hipFuncSetAttribute((const void*)gemv_k8_kernel<TBF16, W8, CT, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        cfg_done = true;
    }
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_k8_kernel<T, W8, CT, 8>), dim3(grid), dim3(512), lds, s, a, part));
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_k8_finish_kernel<T, W8, CT>), dim3(grid), dim3(64), 0, s, a, (const f32x4_t*)part));
    return PGV_OK;
}
static bool k8_enabled() {
    static int k8 = -1;
    if (k8 < 0) { const char* e = getenv("PGV_GEMV_K8"); k8 = (e && e[0] == '0') ? 0 : 1; }
    return k8 == 1;
}

int pgv_launch_gemv(pgv_ctx* ctx, int dtype, int mode, const void* W, const void* x, int ldx, void* out, int ldo, int N, int K, int B, hipStream_t s,
                    const float* wscale, const GemvNorm* norm, const float* xscale) {
    PGV_CHECK(B >= 1 && B <= 64, "gemv: batch %d outside [1,64]", B);
    PGV_CHECK(K % 32 == 0, "gemv: K=%d must be a multiple of 32", K);
    const bool w8 = wscale != nullptr;                  // W is the fp8 blocked copy (fp8.hip) with per-row scales
    GemvArgs a;
    a.W = (const char*)W; a.x = (const char*)x; a.out = (char*)out; a.N = N; a.K = K; a.B = B; a.ldx = ldx; a.ldo = ldo; a.wscale = wscale;
    a.ssq_in = nullptr; a.nparts_in = 0; a.inv_h = 0.f; a.eps = 0.f; a.gamma = nullptr; a.xg = nullptr; a.ssq_out = nullptr; a.amax_val = nullptr; a.amax_idx = nullptr;
    a.ssq_ts = 0; a.amax_ts = 0; a.xscale = xscale; a.lds_bytes = 0;
    PGV_CHECK(xscale == nullptr || (w8 && B <= 16 && mode != GV_RESID), "gemv: the fp8 x fp8 form needs fp8 weights, at most 16 sequences and a folded-norm mode");
    PGV_CHECK(B <= 16 || norm == nullptr || (norm->ssq_ts > 0 && (norm->amax_val == nullptr || norm->amax_ts > 0)), "gemv: batches beyond 16 need the tile strides of the side arrays");
    if (norm) {
        a.ssq_ts = norm->ssq_ts; a.amax_ts = norm->amax_ts;
        a.ssq_in = norm->ssq_in; a.nparts_in = norm->nparts_in; a.inv_h = 1.0f / (float)norm->hidden; a.eps = norm->eps;
        a.gamma = norm->gamma; a.xg = (char*)norm->xg; a.ssq_out = norm->ssq_out; a.amax_val = norm->amax_val; a.amax_idx = norm->amax_idx;
        PGV_CHECK(norm->nparts_in * 4 <= 3 * 8 * 64, "gemv: %d sum-of-squares partials exceed what a consumer workgroup loads (hidden <= 6144)", norm->nparts_in);
    }
    PGV_CHECK(mode != GV_RESIDNORM || (a.gamma && a.xg && a.ssq_out), "gemv: the residual+norm producer needs gamma / xg / ssq_out");
    if (w8) PGV_CHECK(K % 64 == 0, "gemv fp8: K=%d must be a multiple of 64", K);
    int grid;
    // W must be in the fragment-blocked layout with its row count padded to a multiple of 16 (zero rows)
    if (mode == GV_SWIGLU) { PGV_CHECK(N % 64 == 0, "gemv swiglu: N=%d must be a multiple of 64", N); grid = N / 32; }
    else { if (mode != GV_F32) PGV_CHECK(N % 16 == 0, "gemv: N=%d must be a multiple of 16", N); grid = (N + 15) / 16; }
    pgv_prof_begin(ctx, 3, s);
    // register-buffer depth: an fp8 group is one 16-byte load per lane and row block, a 16-bit group two -- the fp8 variants buffer twice as
    // many groups for the same bytes in flight
#define PGV_GEMV_X(MODE_, NW_, TL_, PU16_, PU8_, X2_, GRID_) do { \
        if (w8) PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE_, true, NW_, TL_, PU8_, X2_>), dim3(GRID_), dim3(NW_ * 64), 0, s, a)); \
        else PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE_, false, NW_, TL_, PU16_, X2_>), dim3(GRID_), dim3(NW_ * 64), 0, s, a)); } while (0)
#define PGV_GEMV(MODE_, NW_, TL_, PU16_, PU8X_, PU8_, GRID_) do { \
        if (x2) PGV_GEMV_X(MODE_, NW_, TL_, PU16_, PU8X_, true, GRID_); else PGV_GEMV_X(MODE_, NW_, TL_, PU16_, PU8_, false, GRID_); } while (0)
    // batches beyond one MFMA tile: CT column tiles per weight fragment, one 64-column group per register buffer, no merged x load
#define PGV_GEMV_W(MODE_, TL_, CT_, GRID_) do { \
        if (w8) PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE_, true, 8, TL_, 1, false, CT_>), dim3(GRID_), dim3(512), 0, s, a)); \
        else PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE_, false, 8, TL_, 1, false, CT_>), dim3(GRID_), dim3(512), 0, s, a)); } while (0)
#define PGV_GEMV_WIDE(MODE_, TL_, GRID_) do { if (B <= 32) PGV_GEMV_W(MODE_, TL_, 2, GRID_); else PGV_GEMV_W(MODE_, TL_, 4, GRID_); } while (0)
    if (B > 16) {
        // Wide batches re-read 2 - 4 x the activation lines per workgroup, so the row blocks that share them matter more: qkv 3, gate/up two
        // (gate, up) pairs when the block counts divide.  Same K partition and reduction order as the narrow kernels: same bits per column.
        switch (mode) {
            case GV_STORE16: if (grid % 3 == 0) PGV_GEMV_WIDE(GV_STORE16, 3, grid / 3); else PGV_GEMV_WIDE(GV_STORE16, 1, grid); break;
            case GV_RESID: PGV_GEMV_WIDE(GV_RESID, 1, grid); break;
            case GV_SWIGLU: if (grid % 2 == 0 && B <= 32) PGV_GEMV_W(GV_SWIGLU, 4, 2, grid / 2); else PGV_GEMV_WIDE(GV_SWIGLU, 2, grid); break;   // (4 row blocks x 4 column tiles spill)
            case GV_F32: PGV_GEMV_WIDE(GV_F32, 1, grid); break;
            case GV_RESIDNORM: {
                // narrow matrices: 8 row blocks x K phase per workgroup + a finish launch (gemv_k8_kernel) when the scratch is there
                // where it pays (kernel trace at 32 clips, gpurun_out/r4q): down_proj 32.0 -> 22.4 + 4.9 us (finish), o_proj 13.3 -> 9.6 + 4.9: the
                // short-K matrix only gains once the batch spans four column tiles
                // (4 row blocks per workgroup, two workgroups per CU: 29.6 us per launch against 28.0 at 32 clips, gpurun_out/r4p)
                if (k8_enabled() && norm && norm->k8_part && grid % 8 == 0 && (K / 64) >= 16 && (B > 32 || K >= 8192)) {
                    if (w8) { if (B <= 32) PGV_TRY((launch_k8<true, 2>(dtype, a, grid, norm->k8_part, s))); else PGV_TRY((launch_k8<true, 4>(dtype, a, grid, norm->k8_part, s))); }
                    else { if (B <= 32) PGV_TRY((launch_k8<false, 2>(dtype, a, grid, norm->k8_part, s))); else PGV_TRY((launch_k8<false, 4>(dtype, a, grid, norm->k8_part, s))); }
                } else PGV_GEMV_WIDE(GV_RESIDNORM, 1, grid);
                break;
            }
            default: pgv_set_error("gemv: bad mode %d", mode); return PGV_EINVAL;
        }
        pgv_prof_end(ctx, 3, s, 2.0 * B * (double)N * K, (w8 ? 1.0 : 2.0) * (double)N * K);
        return PGV_OK;
    }
    if (xscale) {                                       // fp8 x fp8 MFMA form (flag path)
#define PGV_GEMV_A8(MODE_, TL_, PU_, GRID_) do { \
        if (B <= 8) PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE_, true, 8, TL_, PU_, true, 1, true>), dim3(GRID_), dim3(512), 0, s, a)); \
        else PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE_, true, 8, TL_, PU_, false, 1, true>), dim3(GRID_), dim3(512), 0, s, a)); } while (0)
        switch (mode) {
            case GV_STORE16: if (grid % 3 == 0 && grid / 3 <= 2 * ctx->num_cu) PGV_GEMV_A8(GV_STORE16, 3, 2, grid / 3); else PGV_GEMV_A8(GV_STORE16, 1, 4, grid); break;
            case GV_SWIGLU: PGV_GEMV_A8(GV_SWIGLU, 2, 2, grid); break;
            case GV_F32: PGV_GEMV_A8(GV_F32, 1, 4, grid); break;
            case GV_RESIDNORM: PGV_GEMV_A8(GV_RESIDNORM, 1, 4, grid); break;
            default: pgv_set_error("gemv: mode %d has no fp8 x fp8 form", mode); return PGV_EINVAL;
        }
#undef PGV_GEMV_A8
        pgv_prof_end(ctx, 3, s, 2.0 * B * (double)N * K, (double)N * K);
        return PGV_OK;
    }
    static int tl3 = -1, x2env = -1;
    if (tl3 < 0) { const char* e = getenv("PGV_GEMV_TL3"); tl3 = (e && e[0] == '0') ? 0 : 1; }
    if (x2env < 0) { const char* e = getenv("PGV_GEMV_X2"); x2env = (e && e[0] == '0') ? 0 : 1; }
    const bool x2 = x2env && B <= 8;
    switch (mode) {
        case GV_STORE16:
            // three row blocks per workgroup when that puts at most ~one workgroup on every CU and nothing is left over (7B qkv: 768 -> 256)
            // (13B: 960 -> 320 workgroups, two resident per CU: with fp8 weights a 16-row workgroup requests as many activation lines as weight
            // lines -- qkv 20.7 -> 18.7 us in the chain lab, gpurun_out/r4e; with 16-bit weights the same launch shape LOSES 4 us per layer: 13B bf16
            // 4.75 -> 4.64 videos/s, gpurun_out/r4final1 -- so the relaxed bound is for the fp8 stream only)
            if (tl3 && grid % 3 == 0 && grid / 3 <= (w8 ? 2 : 1) * ctx->num_cu && grid / 3 >= ctx->num_cu / 2) PGV_GEMV(GV_STORE16, 8, 3, 1, 2, 2, grid / 3);
            else PGV_GEMV(GV_STORE16, 8, 1, 2, 4, 4, grid);
            break;
        case GV_RESID: PGV_GEMV(GV_RESID, 8, 1, 2, 4, 4, grid); break;
        case GV_SWIGLU: PGV_GEMV(GV_SWIGLU, 8, 2, 1, 2, 2, grid); break;
        case GV_F32: PGV_GEMV(GV_F32, 8, 1, 2, 4, 4, grid); break;
        case GV_RESIDNORM: {
            // fp8 down_proj of the 13B shapes (K = 13 824, 320 row blocks): a 16-row workgroup requests as many activation lines as weight lines
            // (LAB.md "what the activation operand costs"); the 8-phase form reads 1/8 of x per workgroup and is bitwise the same result
            static int mink = -1;
            if (mink < 0) { const char* e = getenv("PGV_GEMV_K8_NARROW_MINK"); mink = e ? atoi(e) : 12288; }
            if (w8 && k8_enabled() && norm && norm->k8_part && grid % 8 == 0 && K >= mink) PGV_TRY((launch_k8<true, 1>(dtype, a, grid, norm->k8_part, s)));
            else PGV_GEMV(GV_RESIDNORM, 8, 1, 2, 4, 4, grid);
            break;
        }
        default: pgv_set_error("gemv: bad mode %d", mode); return PGV_EINVAL;
    }
#undef PGV_GEMV
#undef PGV_GEMV_X
#undef PGV_GEMV_WIDE
#undef PGV_GEMV_W
    pgv_prof_end(ctx, 3, s, 2.0 * B * (double)N * K, (w8 ? 1.0 : 2.0) * (double)N * K);
    return PGV_OK;
}

// Workgroups per (sequence, head) unit.  The split is a function of the MODEL (its head count), never of the batch: a sequence's attention is
// then computed with the same partition and the same merge order whether it is decoded alone or next to 15 others -- results stay bitwise
// batch-invariant (tests/test_gpu_llm.py).  Head counts whose 8-sequence launch fills the chip's CUs in whole rounds (7B: 32 heads x 8 = 256)
// keep the unsplit kernel; the others (13B: 40 heads -> 320 units = 1 1/4 rounds) are cut in 2 (640 workgroups, all resident at once).
// Measured (gpurun_out/r4d, 13B fp8, 8 sequences, us per layer): unsplit 18.7, 2 parts 18.1, 4 parts 22.6, 8 parts 31.4 -- 7B (256 units):
// 12.8 / 14.2 / 18.8 / 26.4.  Every extra hand-off through the memory side (write-through stores -> acknowledged -> ticket -> loads) adds
// ~4 us to a workgroup's life, which more resident workgroups only partly hide: a cut in 2 is the only one that pays, and only where the
// unsplit launch leaves a ragged round.  PGV_DATTN_SPLIT=1/2/4/8 forces a value (A/B and the parity tests of every variant; bitwise
// invariance then holds only among runs with the same setting).
static int decode_attn_split(int heads, int num_cu) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PGV_DATTN_SPLIT"); forced = e ? atoi(e) : 0; }
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    return (heads * 8) % num_cu == 0 ? 1 : 2;
}

int pgv_launch_decode_attn(pgv_ctx* ctx, int dtype, const void* qkv, const int* pos, const void* rope, void* Kc, void* Vc, void* out, int B, int H,
                           int heads, int max_seq, double bytes, hipStream_t s, float* part, unsigned* ticket) {
    const float sc = 0.08838834764831845f * 1.4426950408889634f;
    static int nw = -1;
    if (nw < 0) { const char* e = getenv("PGV_DATTN_WAVES"); nw = (e && atoi(e) == 16) ? 16 : 8; }
    const int split = (part && ticket) ? decode_attn_split(heads, ctx->num_cu) : 1;
    pgv_prof_begin(ctx, 4, s);
#define PGV_DATTN_SPLIT_LAUNCH(S_) \
        PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((decode_attn_split_kernel<T, S_>), dim3(heads, B, S_), dim3(512), 0, s, (const typename T::elem*)qkv, pos, \
                                                        (const float2*)rope, (typename T::elem*)Kc, (typename T::elem*)Vc, (typename T::elem*)out, H, heads, \
                                                        max_seq, sc, part, ticket))
    if (split == 2) PGV_DATTN_SPLIT_LAUNCH(2);
    else if (split == 4) PGV_DATTN_SPLIT_LAUNCH(4);
    else if (split == 8) PGV_DATTN_SPLIT_LAUNCH(8);
#undef PGV_DATTN_SPLIT_LAUNCH
    else if (nw == 16)
        PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((decode_attn_kernel<T, 16>), dim3(heads, B), dim3(1024), 0, s, (const typename T::elem*)qkv, pos,
                                                        (const float2*)rope, (typename T::elem*)Kc, (typename T::elem*)Vc, (typename T::elem*)out, H, heads,
                                                        max_seq, sc));
    else
        PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((decode_attn_kernel<T, 8>), dim3(heads, B), dim3(512), 0, s, (const typename T::elem*)qkv, pos,
                                                        (const float2*)rope, (typename T::elem*)Kc, (typename T::elem*)Vc, (typename T::elem*)out, H, heads,
                                                        max_seq, sc));
    pgv_prof_end(ctx, 4, s, 0.0, bytes);
    return PGV_OK;
}

int pgv_launch_argmax_parts(const float* val, const int* idx, int nblk, int amax_ts, int V, int B, int* next, int* pos, int* step, int* hist, int hist_stride, int* done,
                            int eos, int advance, hipStream_t s) {
    hipLaunchKernelGGL(argmax_parts_kernel, dim3(B), dim3(256), 0, s, val, idx, nblk, amax_ts, V, next, pos, step, hist, hist_stride, done, eos, advance);
    return PGV_OK;
}

int pgv_launch_sample(const float* logits, int V, int B, float temperature, int top_k, const float* u, int u_stride, int u_by_step, int* next, int* pos,
                      int* step, int* hist, int hist_stride, int* done, int eos, int advance, hipStream_t s) {
    PGV_CHECK(V >= 1 && V <= 16 * 64 * SAMPLE_MAXR, "sample: vocabulary %d outside [1, %d]", V, 16 * 64 * SAMPLE_MAXR);
    PGV_CHECK(temperature > 0.f, "sample: temperature must be positive (got %g); use the greedy path for temperature 0", (double)temperature);
    SampleArgs a;
    a.logits = logits; a.V = V; a.c = 1.4426950408889634f / temperature; a.top_k = top_k; a.u = u; a.u_stride = u_stride; a.u_by_step = u_by_step;
    a.next = next; a.pos = pos; a.step = step; a.hist = hist; a.hist_stride = hist_stride; a.done = done; a.eos = eos; a.advance = advance;
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(1024), 0, s, a);
    return PGV_OK;
}

extern "C" int pgv_sample_logits(pgv_ctx* ctx, const float* d_logits, int V, int B, float temperature, int top_k, const float* d_u, int32_t* d_next,
                                 void* stream) {
    PGV_CHECK(ctx && d_logits && d_u && d_next && B >= 1, "pgv_sample_logits: bad arguments");
    PGV_TRY(pgv_launch_sample(d_logits, V, B, temperature, top_k, d_u, B, 0, d_next, nullptr, nullptr, nullptr, 0, nullptr, -1, 0, (hipStream_t)stream));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

extern "C" int pgv_gemv(pgv_ctx* ctx, int dtype, int mode, const void* d_W, const void* d_x, int ldx, void* d_out, int ldo, int N, int K, int B,
                        void* stream) {
    PGV_CHECK(ctx && d_W && d_x && d_out, "pgv_gemv: null argument");
    PGV_CHECK(mode >= 0 && mode <= 3, "pgv_gemv: mode %d outside [0,3]", mode);
    return pgv_launch_gemv(ctx, dtype, mode, d_W, d_x, ldx, d_out, ldo, N, K, B, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

#include "weights.h"
extern "C" int pgv_pack_blocked(pgv_ctx* ctx, int dtype, const void* d_src, int rows, int cols, void* d_dst, void* stream) {
    PGV_CHECK(ctx && d_src && d_dst && rows > 0 && cols > 0 && cols % 32 == 0, "pgv_pack_blocked: bad arguments");
    PackDst d;
    d.ptr = d_dst; d.dst_dtype = dtype; d.rows = rows; d.cols = cols; d.dst_stride = cols; d.blocked = true;
    return pgv_pack_tensor(d, d_src, dtype, 1, (hipStream_t)stream);
}

extern "C" int pgv_gemv_fp8(pgv_ctx* ctx, int dtype, int mode, const void* d_W8, const float* d_scales, const void* d_x, int ldx, void* d_out, int ldo, int N,
                            int K, int B, void* stream) {
    PGV_CHECK(ctx && d_W8 && d_scales && d_x && d_out, "pgv_gemv_fp8: null argument");
    PGV_CHECK(mode >= GV_STORE16 && mode <= GV_F32, "pgv_gemv_fp8: mode %d outside [0,3]", mode);
    return pgv_launch_gemv(ctx, dtype, mode, d_W8, d_x, ldx, d_out, ldo, N, K, B, (hipStream_t)stream, d_scales, nullptr, nullptr);
}

extern "C" int pgv_quantize_act_hilo(pgv_ctx* ctx, int dtype, const void* d_x, int ldx, int B, int K, void* d_x8, float* d_xscales, void* stream) {
    PGV_CHECK(ctx && d_x && d_x8 && d_xscales && B >= 1 && B <= 64, "pgv_quantize_act_hilo: bad arguments");
    PGV_TRY(pgv_launch_quant_hilo(dtype, d_x, ldx, K, B, d_x8, d_xscales, (hipStream_t)stream));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

extern "C" int pgv_gemv_fp8_a8(pgv_ctx* ctx, int dtype, int mode, const void* d_W8, const float* d_wscales, const void* d_x8, const float* d_xscales, void* d_out,
                               int ldo, int N, int K, int B, void* stream) {
    PGV_CHECK(ctx && d_W8 && d_wscales && d_x8 && d_xscales && d_out, "pgv_gemv_fp8_a8: null argument");
    PGV_CHECK(mode == GV_STORE16 || mode == GV_SWIGLU || mode == GV_F32, "pgv_gemv_fp8_a8: mode %d outside {0, 2, 3}", mode);
    return pgv_launch_gemv(ctx, dtype, mode, d_W8, d_x8, K, d_out, ldo, N, K, B, (hipStream_t)stream, d_wscales, nullptr, d_xscales);
}
