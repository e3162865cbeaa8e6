// LLaMA decoder, prompt side (HF:llama/modeling_llama.py; reference glue video_chatgpt/model/video_chatgpt.py):
//   embed_splice   : embed_tokens gather + replacement of the <vid_patch> run by projected video rows (:100-168)
//   rope_kv_write  : rotate-half RoPE on q,k (:129-160) + KV-cache append
//   prefill_attn   : causal flash attention, head_dim 128, K/V streamed from the cache through LDS, MFMA; with a per-sequence key offset the
//                    new rows attend to the cached prefix as well (pgv_llm_prefill_append: forward() with past_key_values and S > 1)
// (the prompt's projections run on the persistent GEMM of gemm.hip)
#include "llm_internal.h"

namespace {

constexpr int HD = kHD;

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
    const int* koff;      // [B] tokens already in the cache in front of this call's rows (null: 0) -- query i sits at position koff + i
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
    const int off = p.koff ? p.koff[b] : 0;           // cached prefix: keys [0, off) precede this call's rows; Tk = off + S keys exist
    const int Tk = off + S;
    const int l31 = lane & 31, hi = lane >> 5;
    const int qi = q0 + w * 32 + l31;                 // this lane's query index within this call's rows of the sequence
    const int qpos = off + qi;                        // ... and its position
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

    // Key chunks are aligned to ABSOLUTE positions (multiples of 64) whatever the offset and the query tile, a chunk a query cannot see adds
    // exact zeros (alpha = 1, p = 0): a row's result does not depend on how the prompt was cut into calls -- an appended row is bitwise the row
    // of one full prefill.
    const int kend = min(Tk, off + q0 + 128);
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
                const int kr = min(kc0 + row, Tk - 1);
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
                const int vr = min(kc0 + row, Tk - 1);      // rows past Tk: finite duplicates, their probabilities are exactly 0
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + ((size_t)vr * HD + chunk * 8) * 2),
                                                 (__attribute__((address_space(3))) void*)(Vs + g * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kc0 > off + q0 + w * 32 + 31) continue;     // chunk entirely in this wave's future: nothing to add

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
                v = (key <= qpos && key < Tk) ? v : NEG;
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
int pgv_launch_prefill_attn(pgv_ctx* ctx, int dtype, const void* qkv, void* out, const void* Kc, const void* Vc, const int* cu, const int* koff, int B, int max_len,
                            int H, int heads, int max_seq, double flops, hipStream_t s) {
    PrefillAttnArgs a;
    a.qkv = (const char*)qkv; a.out = (char*)out; a.Kc = (const char*)Kc; a.Vc = (const char*)Vc; a.cu = cu; a.koff = koff;
    a.H = H; a.heads = heads; a.max_seq = max_seq;
    a.scale_log2e = 0.08838834764831845f * 1.4426950408889634f;    // 128^-0.5 * log2(e)
    dim3 grid((max_len + 127) / 128, heads, B);
    pgv_prof_begin(ctx, 2, s);
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((prefill_attn_kernel<T>), grid, dim3(256), 0, s, a));
    pgv_prof_end(ctx, 2, s, flops, 0.0);
    return PGV_OK;
}
