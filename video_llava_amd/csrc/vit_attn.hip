// CLIP multi-head self-attention for batched video frames (HF:clip/modeling_clip.py CLIPAttention :280-335,
// eager math :259-277): per frame and head, softmax(q k^T * d^-0.5) v with N = patches+1 tokens, d = 64, no mask,
// softmax in fp32.
//
// One workgroup per (frame, head).  The whole K [Npad,64] and V^T [64,Npad] of that head live in LDS
// (73 KiB at N=257 -> two workgroups per CU), each wave owns 32-query blocks.  Both products run on
// v_mfma_f32_32x32x16 with *swapped* operands so that a lane owns one query column throughout:
//   S^T[key, q]  = K[key,:] . Q[q,:]      (A = K fragment from LDS, B = Q fragment held in registers)
//   O^T[d, q]   += V^T[d, keys] . P^T[keys, q]   (A = V^T fragment from LDS, B = P packed from the S registers)
// so row max / row sum are lane-local plus one cross-half shuffle, and the probabilities never leave registers.
// The S accumulator's key order inside a 16-key group is {0-3, 8-11 | 4-7, 12-15} per half-wave; V^T is written to
// LDS with key bits 2 and 3 swapped so each lane's 8 k-slots are one contiguous ds_read_b128.
// K rows are DMA'd with global_load_lds (XOR-swizzled via the source address, as in gemm.hip); V is transposed
// through registers with ds_write_b16 (1/9 of a workgroup's LDS traffic).  Keys are processed in chunks of 96 with an
// online softmax so the same kernel covers N=257 (224 px) and N=577 (336 px).
#include "pgv_common.h"

namespace {

constexpr int HD = 64;        // CLIP-L head_dim
constexpr int CB = 3;         // key blocks (of 32) per online-softmax chunk

struct AttnArgs {
    const char* qkv;   // [T*N, ld] 16-bit: q at col 0, k at col C, v at col 2C (fused qkv GEMM output)
    char* out;         // [T*N, ldo] 16-bit, head h at columns h*64
    int ld, ldo;
    int N, C, heads;
    int nkb;           // ceil(N/32)
    int vt_stride;     // bytes per V^T row in LDS (Npad*2 + 16: odd multiple of 16 -> conflict-free b128 reads)
    float scale_log2e; // d^-0.5 * log2(e)
};

template <typename T>
__global__ __launch_bounds__(256) void vit_attn_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int t = blockIdx.x / p.heads, h = blockIdx.x - t * p.heads;
    const int N = p.N, npad = p.nkb * 32;
    char* Ks = smem;                               // [npad][128 B], chunk-swizzled
    char* Vt = smem + (size_t)npad * 128;          // [64][vt_stride]
    const size_t row0 = (size_t)t * N;
    const char* kbase = p.qkv + ((size_t)p.C + h * HD) * 2;
    const char* vbase = p.qkv + ((size_t)2 * p.C + h * HD) * 2;
    const char* qbase = p.qkv + ((size_t)h * HD) * 2;

    // ---- stage K: one wave-instruction = 8 rows x 128 B, destination linear, source chunk pre-swizzled ----
    {
        const int srow = lane >> 3, slot = lane & 7;
        for (int g = w; g < npad / 8; g += nw) {
            const int row = g * 8 + srow;
            const int chunk = slot ^ ((row >> 1) & 7);
            const int rr = min(row, N - 1);
            const char* src = kbase + ((row0 + rr) * p.ld + chunk * 8) * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Ks + g * 1024), 16, 0, 0);
        }
    }
    // ---- stage V transposed: thread takes (key, 8-wide d chunk), writes 8 halfwords to Vt[d][pos(key)] ----
    for (int idx = tid; idx < npad * 8; idx += blockDim.x) {
        const int key = idx >> 3, dc = idx & 7;
        typename T::v8 v;
        if (key < N) v = *(const typename T::v8*)(vbase + ((row0 + key) * p.ld + dc * 8) * 2);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (typename T::elem)0.0f;
        }
        const int pos = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) *(typename T::elem*)(Vt + (size_t)(dc * 8 + e) * p.vt_stride + pos * 2) = v[e];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (lane >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const float NEG = -1e30f;

    for (int qb = w; qb < p.nkb; qb += nw) {
        // Q fragments (B operand): lane (q = l31, hi) holds Q[q][kk*16 + hi*8 .. +8]
        const int qrow = min(qb * 32 + l31, N - 1);
        typename T::v8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const typename T::v8*)(qbase + ((row0 + qrow) * p.ld + kk * 16 + hi * 8) * 2);

        f32x16_t o[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
        float mrun = NEG, lrun = 0.f;

        for (int kb0 = 0; kb0 < p.nkb; kb0 += CB) {
            f32x16_t s[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[c][e] = 0.f;
                if (kb0 + c < p.nkb) {
                    const char* kr = Ks + (size_t)((kb0 + c) * 32 + l31) * 128;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const typename T::v8 kf = *(const typename T::v8*)(kr + koffs[kk]);
                        s[c] = T::mfma32(kf, qf[kk], s[c]);
                    }
                }
            }
            // scale into log2 domain, mask the padded keys, chunk max
            float cmax = NEG;
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int kbase_idx = (kb0 + c) * 32 + 4 * hi;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kbase_idx + (e & 3) + 8 * (e >> 2);
                    float v = s[c][e] * p.scale_log2e;
                    v = (key < N) ? v : NEG;     // also covers kb0+c >= nkb (key >= npad >= N)
                    s[c][e] = v;
                    cmax = fmaxf(cmax, v);
                }
            }
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            const float mnew = fmaxf(mrun, cmax);
            const float alpha = exp2f(mrun - mnew);
            mrun = mnew;
            float psum = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = exp2f(s[c][e] - mnew);
                    s[c][e] = pv;
                    psum += pv;
                }
            lrun = lrun * alpha + psum;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
            // O^T += V^T . P^T
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                if (kb0 + c < p.nkb) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        typename T::v8 pa;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pa[e] = T::from_f32(s[c][ks * 8 + e]);
                        const int col = ((kb0 + c) * 32 + ks * 16 + hi * 8) * 2;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const typename T::v8 vf = *(const typename T::v8*)(Vt + (size_t)(j * 32 + l31) * p.vt_stride + col);
                            o[j] = T::mfma32(vf, pa, o[j]);
                        }
                    }
                }
            }
        }
        const float ltot = lrun + __shfl_xor(lrun, 32, 64);
        const float inv = 1.0f / ltot;
        const int q = qb * 32 + l31;
        if (q < N) {
            char* orow = p.out + ((row0 + q) * p.ldo + h * HD) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = j * 32 + 8 * g + 4 * hi;
                    *(u32x2_t*)(orow + d * 2) = pack4<T>(o[j][g * 4 + 0] * inv, o[j][g * 4 + 1] * inv, o[j][g * 4 + 2] * inv, o[j][g * 4 + 3] * inv);
                }
        }
    }
}

}  // namespace

int pgv_launch_vit_attn(pgv_ctx* ctx, int dtype, const void* qkv, int ld, void* out, int ldo, int T, int N, int C, int heads, hipStream_t s) {
    PGV_CHECK(C == heads * HD, "vit_attn: head_dim must be 64 (hidden %d, heads %d)", C, heads);
    AttnArgs a;
    a.qkv = (const char*)qkv; a.out = (char*)out; a.ld = ld; a.ldo = ldo; a.N = N; a.C = C; a.heads = heads;
    a.nkb = (N + 31) / 32;
    const int npad = a.nkb * 32;
    a.vt_stride = npad * 2 + 16;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
    const size_t lds = (size_t)npad * 128 + (size_t)HD * a.vt_stride;
    PGV_CHECK(lds <= 160 * 1024, "vit_attn: %d tokens per frame need %zu B of LDS (> 160 KiB)", N, lds);
    const int nw = (a.nkb % 3 == 0) ? 3 : 4;
    pgv_prof_begin(ctx, 1, s);
    if (dtype == PGV_F16) {
        static bool cfg = false;
        if (!cfg) { PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); cfg = true; }
        hipLaunchKernelGGL((vit_attn_kernel<TF16>), dim3(T * heads), dim3(nw * 64), lds, s, a);
    } else if (dtype == PGV_BF16) {
        static bool cfg = false;
        if (!cfg) { PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); cfg = true; }
        hipLaunchKernelGGL((vit_attn_kernel<TBF16>), dim3(T * heads), dim3(nw * 64), lds, s, a);
    } else {
        pgv_set_error("vit_attn: unsupported dtype %d", dtype);
        return PGV_EINVAL;
    }
    PGV_HIP(hipGetLastError());
    pgv_prof_end(ctx, 1, s, 4.0 * (double)T * heads * (double)N * N * HD, 2.0 * 4.0 * (double)T * N * C);
    return PGV_OK;
}

extern "C" int pgv_vit_attention(pgv_ctx* ctx, int dtype, const void* d_qkv, void* d_out, int T, int N, int C, int heads, void* stream) {
    PGV_CHECK(ctx && d_qkv && d_out && T > 0 && N > 0, "pgv_vit_attention: bad arguments");
    return pgv_launch_vit_attn(ctx, dtype, d_qkv, 3 * C, d_out, C, T, N, C, heads, (hipStream_t)stream);
}
