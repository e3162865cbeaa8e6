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
#include <stdlib.h>

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
    int abl;           // diagnostic ablation (PGV_ATTN_ABLATE): 1 = no V staging, 2 = no query loop, 4 = no K staging
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
    int* qctr = (int*)(Vt + (size_t)HD * p.vt_stride);   // next unassigned query block
    if (tid == 0) *qctr = nw;
    const size_t row0 = (size_t)t * N;
    const char* kbase = p.qkv + ((size_t)p.C + h * HD) * 2;
    const char* vbase = p.qkv + ((size_t)2 * p.C + h * HD) * 2;
    const char* qbase = p.qkv + ((size_t)h * HD) * 2;

    // ---- stage K: one wave-instruction = 8 rows x 128 B, destination linear, source chunk pre-swizzled ----
    if (!(p.abl & 4)) {
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
    // All global loads of a batch are issued before the first LDS write: one exposed round trip per batch instead of one per
    // element (the loop was 12 dependent load -> 8 x ds_write_b16 rounds and dominated the kernel).
    if (!(p.abl & 1)) {
        constexpr int VB = 10;                                     // loads in flight per thread (40 VGPRs)
        const int total = npad * 8;
        for (int base = tid; base < total; base += VB * blockDim.x) {
            typename T::v8 vv[VB];
#pragma unroll
            for (int u = 0; u < VB; ++u) {
                const int idx = base + u * blockDim.x;
                const int key = min(idx >> 3, N - 1), dc = idx & 7;
                vv[u] = *(const typename T::v8*)(vbase + ((row0 + key) * p.ld + dc * 8) * 2);
            }
#pragma unroll
            for (int u = 0; u < VB; ++u) {
                const int idx = base + u * blockDim.x;
                if (idx < total) {
                    const int key = idx >> 3, dc = idx & 7;
                    const int pos = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        *(typename T::elem*)(Vt + (size_t)(dc * 8 + e) * p.vt_stride + pos * 2) = (key < N) ? vv[u][e] : (typename T::elem)0.0f;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (lane >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const float NEG = -1e30f;            // raw-score domain: NEG * scale is still hugely negative, exp2 -> 0

    // Query blocks of 32 are handed out dynamically through an LDS counter: the kernel is bound by the softmax VALU work, a CU runs
    // two workgroups (8 waves on 4 SIMDs) and a static split of 9 blocks over the waves left whole SIMDs idle at the tail.
    // Q fragments (B operand): lane (q = l31, hi) holds Q[q][kk*16 + hi*8 .. +8]; the next block's are fetched during this one's work.
    auto load_q = [&](int qb, typename T::v8 (&dst)[4]) {
        const int qrow = min(qb * 32 + l31, N - 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dst[kk] = *(const typename T::v8*)(qbase + ((row0 + qrow) * p.ld + kk * 16 + hi * 8) * 2);
    };
    auto next_block = [&]() -> int {
        int v = 0;
        if (lane == 0) v = atomicAdd(qctr, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    const int nblocks = (p.abl & 2) ? 0 : p.nkb;
    typename T::v8 qnext[4];
    int qb = w, qb_next = next_block();
    load_q(qb, qnext);
    for (; qb < nblocks; qb = qb_next, qb_next = next_block()) {
        typename T::v8 qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) qf[kk] = qnext[kk];
        load_q(qb_next, qnext);

        f32x16_t o[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
        float mrun = NEG, lrun = 0.f;

        for (int kb0 = 0; kb0 < p.nkb; kb0 += CB) {
            // A key block past the end re-reads the last one (valid LDS) and is masked below: no branches, so the 12 fragment reads
            // of a chunk are issued together and the MFMAs of the three independent accumulators interleave (the first version
            // ran ds_read -> wait -> dependent MFMA one at a time and was latency-bound at ~180 cycles per MFMA).
            int kb[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) kb[c] = min(kb0 + c, p.nkb - 1);
            typename T::v8 kf[CB][4];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const char* kr = Ks + (size_t)(kb[c] * 32 + l31) * 128;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kf[c][kk] = *(const typename T::v8*)(kr + koffs[kk]);
            }
            f32x16_t s[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[c][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int c = 0; c < CB; ++c) s[c] = T::mfma32(kf[c][kk], qf[kk], s[c]);
            // V^T fragments of the chunk: issued now, their LDS latency hides under the softmax arithmetic
            typename T::v8 vf[CB][2][2];
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        vf[c][ks][j] = *(const typename T::v8*)(Vt + (size_t)(j * 32 + l31) * p.vt_stride + (kb[c] * 32 + ks * 16 + hi * 8) * 2);
            // chunk max on the raw scores (scale > 0), padded keys masked only in the chunk that holds them (wave-uniform test)
            if ((kb0 + CB) * 32 > N) {
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const int kbase_idx = (kb0 + c) * 32 + 4 * hi;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kbase_idx + (e & 3) + 8 * (e >> 2);
                        s[c][e] = (key < N) ? s[c][e] : NEG;     // also covers kb0+c >= nkb (key >= npad >= N)
                    }
                }
            }
            float cmax = NEG;
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) cmax = fmaxf(cmax, s[c][e]);
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
            const float mnew = fmaxf(mrun, cmax * p.scale_log2e);
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            mrun = mnew;
            // p = exp2(s * scale - m): one packed FMA per two scores, packed partial sums
            const f32x2_t sc2 = {p.scale_log2e, p.scale_log2e}, nm2 = {-mnew, -mnew};
            f32x2_t psum2 = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2_t x = f32x2_t{s[c][e], s[c][e + 1]} * sc2 + nm2;
                    const f32x2_t pv = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    s[c][e] = pv[0]; s[c][e + 1] = pv[1];
                    psum2 += pv;
                }
            lrun = lrun * alpha + (psum2[0] + psum2[1]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
            // O^T += V^T . P^T
#pragma unroll
            for (int c = 0; c < CB; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    typename T::v8 pa;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pa[e] = T::from_f32(s[c][ks * 8 + e]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) o[j] = T::mfma32(vf[c][ks][j], pa, o[j]);
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
    { static int abl = -1; if (abl < 0) { const char* e = getenv("PGV_ATTN_ABLATE"); abl = e ? atoi(e) : 0; } a.abl = abl; }
    const size_t lds = (size_t)npad * 128 + (size_t)HD * a.vt_stride + 16;    // + the query-block counter
    PGV_CHECK(lds <= 160 * 1024, "vit_attn: %d tokens per frame need %zu B of LDS (> 160 KiB)", N, lds);
    const int nw = 4;
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
