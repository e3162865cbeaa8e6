// CLIP multi-head self-attention for batched video frames (HF:clip/modeling_clip.py CLIPAttention :280-335,
// eager math :259-277): per frame and head, softmax(q k^T * d^-0.5) v with N = patches+1 tokens, d = 64, no mask,
// softmax in fp32.
//
// One workgroup per (frame, head).  The whole K and V [Npad,64] of that head live in LDS, both ROW-major and both staged by
// LDS-DMA (global_load_lds, 16-B chunks XOR-swizzled through the source address; 72 KiB at N=257 -> two workgroups per CU);
// each wave owns 32-query blocks.  Both products run on v_mfma_f32_32x32x16 with *swapped* operands so that a lane owns one
// query column throughout:
//   S^T[key, q]  = K[key,:] . Q[q,:]      (A = K fragment from LDS, B = Q fragment held in registers)
//   O^T[d, q]   += V^T[d, keys] . P^T[keys, q]   (A = V^T fragment, B = P packed from the S registers)
// so row max / row sum are lane-local plus one cross-half shuffle, and the probabilities never leave registers.
// The V^T fragments come from the row-major V image through gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group
// reads a [4 keys][16 d] block, every lane supplying the address of 4 consecutive d of one key and receiving 4 consecutive keys
// of one d).  The S accumulator's key order inside a 16-key group is {0-3, 8-11 | 4-7, 12-15} per half-wave, which is exactly
// two such 4-key reads per lane.  (Round 1 first transposed V through registers with ds_write_b16: 8-way bank conflicts on the
// writes -- SQ_LDS_BANK_CONFLICT was 48 % of SQ_LDS_IDX_ACTIVE -- and 40 VGPRs of staging buffers.)
// Keys are processed in chunks of 96 with an online softmax so the same kernel covers N=257 (224 px) and N=577 (336 px).
#include <stdlib.h>

#include <type_traits>

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
    float scale_log2e; // d^-0.5 * log2(e)
#ifdef PGV_LAB
    int abl;           // lab builds only (-DPGV_LAB, PGV_ATTN_ABLATE): 1 = no V staging, 2 = no query loop, 4 = no K staging,
                       // 8 = no exp (copy), 16 = no PV MFMAs, 32 = no S MFMAs, 64 = no max (wrong results by design: timing ablations)
#endif
};
#ifdef PGV_LAB
#define PGV_ATTN_ABL(p) ((p).abl)
#else
#define PGV_ATTN_ABL(p) 0          // the release library has no garbage-producing ablation switch (documented A/B switches: INTEGRATION.md)
#endif

// cross-half (lane l <-> l + 32) maximum / sum with gfx950's half swap: one VALU op instead of a ds_bpermute round trip through the LDS crossbar.
// v_permlane32_swap x, y exchanges x's upper half with y's lower half: lanes 0-31 end up with (own, partner's), lanes 32-63 with (partner's, own).
// Inline asm, because the builtin cannot be used here: with one value on both operands hipcc 7.2 reads the second result from the wrong
// register (gemv.hip rows_sum_to_row3), and with a second SSA copy of the value it folds max(r0, r1) to r0 (observed in the ISA).  The
// `s_nop 1` is the wait state the compiler itself places between a VALU write of an operand and the swap (the first round-5 version had none:
// wrong maxima at N = 5); nothing is needed behind it.
__device__ __forceinline__ void half_swap(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_swap_max(float v) {
    float a = v, b = v;
    half_swap(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float half_swap_sum(float v) {
    float a = v, b = v;
    half_swap(a, b);
    return a + b;
}

// Lazy rescale of the online softmax (round 5): the running maximum m only has to BOUND the scores loosely -- p = exp2(s - m) stays exact in
// fp32 and representable in 16 bits as long as s - m <= LAZY_TH -- so the accumulator rescale (32 multiplies + an exp2 per chunk, and a
// dependency between the softmax and the P.V MFMAs) runs only when some query of the wave saw a score more than LAZY_TH (log2 units) above
// its running maximum: after the first chunk practically never.  Mathematically identical (every term carries the same factor exp2(-m), which
// cancels against the row sum); P <= 2^LAZY_TH = 64 is far inside fp16 / bf16 range, relative rounding unchanged.
#ifndef PGV_LAB_ATTN_LAZY_TH
#define PGV_LAB_ATTN_LAZY_TH 6.0f      // lab: -DPGV_LAB_ATTN_LAZY_TH=-1.0f restores the rescale on every chunk
#endif

// NWAVES = waves per workgroup: 4 where two workgroups share a CU (K + V of a head <= 80 KiB: 224 px), 8 where only one fits (336 px: 152 KiB)
// so that every SIMD still has two waves whose MFMA and softmax phases overlap.
template <typename T, int ABL, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void vit_attn_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = blockDim.x >> 6;
    const int t = blockIdx.x / p.heads, h = blockIdx.x - t * p.heads;
    const int N = p.N, npad = p.nkb * 32;
    char* Ks = smem;                               // [npad][128 B], chunk-swizzled
    char* Vs = smem + (size_t)npad * 128;          // [npad][128 B], chunk bit 2 flipped on rows with (row >> 1) & 1
    int* qctr = (int*)(Vs + (size_t)npad * 128);   // next unassigned query block
    if (tid == 0) *qctr = nw;
    const size_t row0 = (size_t)t * N;
    const char* kbase = p.qkv + ((size_t)p.C + h * HD) * 2;
    const char* vbase = p.qkv + ((size_t)2 * p.C + h * HD) * 2;
    const char* qbase = p.qkv + ((size_t)h * HD) * 2;

    const int l31 = lane & 31, hi = lane >> 5;
    // Query blocks of 32 are handed out dynamically through an LDS counter: a CU runs two workgroups (8 waves on 4 SIMDs) and a static
    // split of 9 blocks over the waves left whole SIMDs idle at the tail.
    // Q fragments (B operand): lane (q = l31, hi) holds Q[q][kk*16 + hi*8 .. +8].  The first block's are requested before the K/V
    // staging and every later block's one block ahead, and each block ends with an explicit vmcnt(0) BEFORE its output stores are
    // issued: hipcc sizes the vmcnt in front of the first MFMA for the loop-entry path (4 younger loads), which on the back edge also
    // drained the 8 output stores issued in between.
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
    typename T::v8 qnext[4];
    load_q(w, qnext);

    // ---- stage K and V: one wave-instruction = 8 rows x 128 B, destination linear, source chunk pre-swizzled.  K: chunk ^ ((row >> 1) & 7)
    // (conflict-free ds_read_b128 of one row per lane); V: chunk ^ (((row >> 1) & 1) << 2) (conflict-free transposing reads: the 32 lanes of
    // an LDS cycle touch 4 consecutive keys x 64 B, and rows two apart share banks).  Rows past N re-read the last row: their
    // probabilities are exactly 0 and the duplicated V values are finite.
    // The DMA is issued by inline asm (M0 = LDS destination of the wave-instruction; nothing else in this kernel uses M0: gfx9 DS instructions
    // do not read it).  With __builtin_amdgcn_global_load_lds hipcc puts s_waitcnt vmcnt(0) in front of the V fragment reads of EVERY chunk while
    // a DMA may be pending (it cannot tell the rows a pending DMA writes from the rows a chunk reads), which would end the overlap below at the
    // first chunk and drain the Q prefetch and the output stores in every later one.  The waits that order DMA and reads are explicit.
#ifndef PGV_LAB_ATTN_PRIO
#define PGV_LAB_ATTN_PRIO 0            // lab A/B: 1 = s_setprio(1) around the MFMA clusters, 2 = static priority for the younger half of the waves
#endif
#ifndef PGV_LAB_ATTN_STAGGER
#define PGV_LAB_ATTN_STAGGER 0         // lab A/B: s_sleep N (x 64 cycles) for the upper half of the waves before the query loop: de-phases the two waves of a SIMD
#endif
#ifndef PGV_LAB_ATTN_SPLIT
#define PGV_LAB_ATTN_SPLIT 3           // lab A/B: chunks of the first staging part where one workgroup owns the CU; 0 = one part
#endif
#ifndef PGV_LAB_ATTN_ASMDMA
#define PGV_LAB_ATTN_ASMDMA 1          // lab A/B: 0 = __builtin_amdgcn_global_load_lds (only sensible with one staging part)
#endif
    auto dma16 = [&](const char* src, const char* dst) {
#if !PGV_LAB_ATTN_ASMDMA
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        return;
#endif
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)dst) : "memory");
    };
    auto stage = [&](int g_lo, int g_hi) {
        const int srow = lane >> 3, slot = lane & 7;
        for (int g = g_lo + w; g < g_hi; g += nw) {
            const int row = g * 8 + srow;
            const int rr = min(row, N - 1);
            if (!(PGV_ATTN_ABL(p) & 4)) dma16(kbase + ((row0 + rr) * p.ld + (slot ^ ((row >> 1) & 7)) * 8) * 2, Ks + g * 1024);
            if (!(PGV_ATTN_ABL(p) & 1)) dma16(vbase + ((row0 + rr) * p.ld + (slot ^ (((row >> 1) & 1) << 2)) * 8) * 2, Vs + g * 1024);
        }
    };
    // Where ONE workgroup owns the CU (336 px: K + V of a head fill its LDS) nobody's compute hides the staging of a unit, so it is cut in two
    // (round 6): the keys of the first SPLIT chunks are staged and waited for, the rest is requested and lands behind the first chunks of every
    // wave's FIRST query block, which stops for it once (`tail_pending`).  With two workgroups per CU (224 px) one's staging already runs under
    // the other's compute: one part.
    constexpr int SPLIT = PGV_LAB_ATTN_SPLIT;        // chunks of the first part: 288 keys
    const int g_all = npad / 8;
    const int g_first = (SPLIT > 0 && NWAVES == 8 && p.nkb > SPLIT * CB) ? SPLIT * CB * 4 : g_all;
    stage(0, g_first);
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the (first part of the) K / V DMA and the first Q fragments
    __syncthreads();
    bool tail_pending = g_first < g_all;             // wave-uniform
    if (tail_pending) stage(g_first, g_all);
    auto tail_arrive = [&](int kb0) {                // before the first chunk that reads keys of the second part
        if (tail_pending && kb0 >= SPLIT * CB) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            tail_pending = false;
        }
    };

    const int sw = (lane >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    // transposing V reads: lane i of a 16-lane group g addresses key (4 hi + i/4) of the 16-key slice, d columns jj*32 + g*16 + (i%4)*4 .. +3
    const int li = lane & 15, lg = (lane >> 4) & 1;
    const int voff0 = (4 * hi + (li >> 2)) * 128 + (((lg * 2 + ((li & 3) >> 1)) ^ (((li >> 3) & 1) << 2)) << 4) + (li & 1) * 8;   // jj = 1: ^ 64
    const float NEG = -1e30f;            // raw-score domain: NEG * scale is still hugely negative, exp2 -> 0

    // S^T of one chunk: key blocks past the end re-read the last one (valid LDS, masked in the softmax) so there are no branches, the 12
    // fragment reads are issued together and the MFMAs of the three independent accumulators interleave.
    // NB (compile time) = key blocks of this chunk: CB for every full chunk; the LAST chunk runs with exactly the blocks that exist (1..CB), so
    // N = 577 (19 blocks = 6 chunks + 1 block) no longer pays for two dummy blocks of MFMAs and exponentials per query block.
    auto scores = [&](int kb0, const typename T::v8 (&qf)[4], f32x16_t (&s)[CB], auto nb_tag) __attribute__((always_inline)) {
        constexpr int NB = decltype(nb_tag)::value;
        typename T::v8 kf[NB][4];
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const char* kr = Ks + (size_t)((kb0 + c) * 32 + l31) * 128;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) kf[c][kk] = *(const typename T::v8*)(kr + koffs[kk]);
        }
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[c][e] = 0.f;
        if constexpr ((PGV_LAB_ATTN_PRIO & 1) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if constexpr ((ABL & 32) != 0) { s[c][kk] += (float)kf[c][kk][0] * (float)qf[kk][0]; continue; }
                s[c] = T::mfma32(kf[c][kk], qf[kk], s[c]);
            }
        if constexpr ((PGV_LAB_ATTN_PRIO & 1) != 0) __builtin_amdgcn_s_setprio(0);
    };
    // One chunk of keys: scores, online-softmax update, O^T += V^T . P^T.  MASK: the chunk may hold keys >= N.
    // Every chunk but a query block's first is absorbed SPECULATIVELY (round 6) -- no chunk maximum, no rescale: p = exp2(s * scale - m) with the
    // running m as it stands.  The row sums that are computed anyway tell whether that was legitimate: all p >= 0, so a lane's partial sum bounds
    // each of its p, and `sum <= SPEC_LIMIT` guarantees p <= 2^12 (exact in fp32, representable in fp16 / bf16, same relative rounding as any
    // other p).  The wave-uniform test replaces the 24 v_max3 + 6 v_max + cross-half swap of a chunk; when it fails (a score more than 2^12 above the
    // running maximum of the chunks before: after a block's first chunk practically never) nothing has been accumulated yet: the scores are
    // recomputed and the chunk takes the exact path (maximum, lazy rescale).  Mathematically the same softmax either way.
#ifndef PGV_LAB_ATTN_SPEC
#define PGV_LAB_ATTN_SPEC 1            // lab A/B: 0 = chunk maximum + lazy rescale on every chunk (round 5)
#endif
    constexpr float SPEC_LIMIT = 4096.0f;
    auto chunk = [&](int kb0, const typename T::v8 (&qf)[4], f32x16_t (&s)[CB], f32x16_t (&o)[2], float& mrun, float& lrun, auto mask_tag, auto nb_tag) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr int NB = decltype(nb_tag)::value;
        typename T::v8 vf[NB][2][2];
        // V^T fragments of the chunk: issued first, their LDS latency hides under the softmax arithmetic
        auto load_v = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const char* vb = Vs + (size_t)(kb0 + c) * 4096;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const char* a0 = vb + (voff0 ^ (j * 64)) + ks * 2048;
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a0);
                        const s16x4_t up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 1024));
                        vf[c][ks][j] = __builtin_bit_cast(typename T::v8, __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
            }
        };
        auto mask = [&]() __attribute__((always_inline)) {
            if constexpr (MASK) {
                constexpr int c = NB - 1;                            // only the last existing block can hold keys >= N
                if ((kb0 + c + 1) * 32 > N) {                        // wave-uniform
                    const int kbase_idx = (kb0 + c) * 32 + 4 * hi;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kbase_idx + (e & 3) + 8 * (e >> 2);
                        s[c][e] = (key < N) ? s[c][e] : NEG;
                    }
                }
            }
        };
        // p = exp2(s * scale - m) over the chunk (one packed FMA per two scores), in place; returns the lane's partial row sum
        auto probs = [&](float m) __attribute__((always_inline)) -> float {
            const f32x2_t sc2 = {p.scale_log2e, p.scale_log2e}, nm2 = {-m, -m};
            f32x2_t psum2 = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NB; ++c)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2_t x = f32x2_t{s[c][e], s[c][e + 1]} * sc2 + nm2;
                    const f32x2_t pv = ((ABL & 8) != 0) ? x : f32x2_t{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    s[c][e] = pv[0]; s[c][e + 1] = pv[1];
                    psum2 += pv;
                }
            return psum2[0] + psum2[1];
        };
        scores(kb0, qf, s, nb_tag);
        load_v();
        mask();
        bool exact = !(PGV_LAB_ATTN_SPEC && kb0 > 0);                // wave-uniform
        float psum = 0.f;
        if (!exact) {
            psum = probs(mrun);
            if (__builtin_amdgcn_ballot_w64(!(psum <= SPEC_LIMIT)) != 0ull) {      // NaN / inf fail too
                exact = true;                                        // nothing accumulated yet: start the chunk over
                scores(kb0, qf, s, nb_tag);
                load_v();
                mask();
            }
        }
        if (exact) {
            float cmax = NEG;
            if constexpr ((ABL & 64) != 0) cmax = s[0][0];
            else {
#pragma unroll
                for (int c = 0; c < NB; ++c)
#pragma unroll
                    for (int e = 0; e < 16; ++e) cmax = fmaxf(cmax, s[c][e]);
                cmax = half_swap_max(cmax);
            }
            const float cm = cmax * p.scale_log2e;
            // Lazy rescale (round 5, see PGV_LAB_ATTN_LAZY_TH): only when some query of the wave saw a score more than LAZY_TH above its running maximum
            if (__builtin_amdgcn_ballot_w64(cm > mrun + PGV_LAB_ATTN_LAZY_TH) != 0ull) {      // wave-uniform; always taken on a block's first chunk (mrun = NEG)
                const float mnew = fmaxf(mrun, cm);
                const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                mrun = mnew;
                lrun *= alpha;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
            }
            psum = probs(mrun);
        }
        lrun += psum;
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                typename T::v8 pa;
#pragma unroll
                for (int e = 0; e < 8; ++e) pa[e] = T::from_f32(s[c][ks * 8 + e]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr ((ABL & 16) != 0) { o[j][ks] += (float)vf[c][ks][j][0] * (float)pa[0]; continue; }
                    o[j] = T::mfma32(vf[c][ks][j], pa, o[j]);
                }
            }
    };

    if constexpr ((PGV_LAB_ATTN_PRIO & 2) != 0) { if (w >= NWAVES / 2) __builtin_amdgcn_s_setprio(1); }
    if constexpr (PGV_LAB_ATTN_STAGGER > 0) { if (w >= NWAVES / 2) __builtin_amdgcn_s_sleep(PGV_LAB_ATTN_STAGGER); }
    const int nblocks = (PGV_ATTN_ABL(p) & 2) ? 0 : p.nkb;
    int qb = w, qb_next = next_block();
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

        // (Issuing the scores of chunk i+1 beside the softmax of chunk i was tried: same time, 48 more VGPRs.)
        f32x16_t s[CB];
        int kb0 = 0;
        for (; kb0 + CB < p.nkb; kb0 += CB) {
            tail_arrive(kb0);
            chunk(kb0, qf, s, o, mrun, lrun, std::false_type{}, std::integral_constant<int, CB>{});
        }
        // the last chunk holds 1..CB blocks (wave-uniform) and is the only one that can hold keys >= N
        const int rem = p.nkb - kb0;
        tail_arrive(kb0);
        if (rem == 1) chunk(kb0, qf, s, o, mrun, lrun, std::true_type{}, std::integral_constant<int, 1>{});
        else if (rem == 2) chunk(kb0, qf, s, o, mrun, lrun, std::true_type{}, std::integral_constant<int, 2>{});
        else chunk(kb0, qf, s, o, mrun, lrun, std::true_type{}, std::integral_constant<int, CB>{});

        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the next block's Q (requested a whole block ago); see load_q
        const float ltot = half_swap_sum(lrun);
        const float inv = 1.0f / ltot;
        const int q = qb * 32 + l31;
        // Output row q lives in the lane pair (q, q + 32): 4 consecutive d per accumulator group, alternating between the two lanes.  Two
        // v_permlane32_swap per group pair hand each lane 8 consecutive d, so a row is written as 32 contiguous bytes per store instruction
        // (4 stores of 16 B per lane instead of 8 of 8 B: the 16-byte pieces were a quarter of the kernel's time on the L2 write path).
        u32x4_t piece[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const u32x2_t a = pack4<T>(o[j][8 * m + 0] * inv, o[j][8 * m + 1] * inv, o[j][8 * m + 2] * inv, o[j][8 * m + 3] * inv);      // group 2m:     d = 16m + 4hi + 0..3
                const u32x2_t b = pack4<T>(o[j][8 * m + 4] * inv, o[j][8 * m + 5] * inv, o[j][8 * m + 6] * inv, o[j][8 * m + 7] * inv);      // group 2m + 1: d = 16m + 8 + 4hi + 0..3
                const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                piece[j][m] = u32x4_t{r0[0], r1[0], r0[1], r1[1]};           // lanes 0-31: d = 16m .. 16m+7; lanes 32-63: d = 16m+8 .. 16m+15
            }
        if (q < N) {
            char* orow = p.out + ((row0 + q) * p.ldo + h * HD) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m) *(u32x4_t*)(orow + (j * 32 + 16 * m + 8 * hi) * 2) = piece[j][m];
        }
    }
    tail_arrive(p.nkb);                              // a wave that got no query block still owes the workgroup its barrier (and its DMAs their wait)
}

}  // namespace

int pgv_launch_vit_attn(pgv_ctx* ctx, int dtype, const void* qkv, int ld, void* out, int ldo, int T, int N, int C, int heads, hipStream_t s) {
    PGV_CHECK(C == heads * HD, "vit_attn: head_dim must be 64 (hidden %d, heads %d)", C, heads);
    AttnArgs a;
    a.qkv = (const char*)qkv; a.out = (char*)out; a.ld = ld; a.ldo = ldo; a.N = N; a.C = C; a.heads = heads;
    a.nkb = (N + 31) / 32;
    const int npad = a.nkb * 32;
    a.scale_log2e = 0.125f * 1.4426950408889634f;
#ifdef PGV_LAB
    { static int abl = -1; if (abl < 0) { const char* e = getenv("PGV_ATTN_ABLATE"); abl = e ? atoi(e) : 0; } a.abl = abl; }
#endif
    const size_t lds = (size_t)npad * 256 + 16;    // K + V images + the query-block counter
    PGV_CHECK(lds <= 160 * 1024, "vit_attn: %d tokens per frame need %zu B of LDS (> 160 KiB)", N, lds);
    static_assert(CB == 3, "the last-chunk dispatch above enumerates 1..3 blocks");
    bool one_wg_per_cu = lds > 80 * 1024;            // 336 px: K + V of a head fill the CU's LDS -> 8 waves so that every SIMD still holds two
#ifdef PGV_LAB
    { static int f = -1; if (f < 0) { const char* e = getenv("PGV_ATTN_NW4"); f = e ? atoi(e) : 0; } if (f) one_wg_per_cu = false; }     // lab A/B: 4 waves everywhere
#endif
    pgv_prof_begin(ctx, 1, s);
#define PGV_ATTN_LAUNCH(T_, ABL_) do { \
        if (one_wg_per_cu) hipLaunchKernelGGL((vit_attn_kernel<T_, ABL_, 8>), dim3(T * heads), dim3(512), lds, s, a); \
        else hipLaunchKernelGGL((vit_attn_kernel<T_, ABL_, 4>), dim3(T * heads), dim3(256), lds, s, a); } while (0)
    if (dtype == PGV_F16) {
        PGV_ATTN_LAUNCH(TF16, 0);
    } else if (dtype == PGV_BF16) {
#ifndef PGV_LAB
        PGV_ATTN_LAUNCH(TBF16, 0);
#else
        switch (a.abl & 0x78) {                      // fine-grained ablations exist for bf16 only
            case 0: PGV_ATTN_LAUNCH(TBF16, 0); break;
            case 8: PGV_ATTN_LAUNCH(TBF16, 8); break;
            case 16: PGV_ATTN_LAUNCH(TBF16, 16); break;
            case 32: PGV_ATTN_LAUNCH(TBF16, 32); break;
            case 64: PGV_ATTN_LAUNCH(TBF16, 64); break;
            case 48: PGV_ATTN_LAUNCH(TBF16, 48); break;
            case 72: PGV_ATTN_LAUNCH(TBF16, 72); break;
            case 120: PGV_ATTN_LAUNCH(TBF16, 120); break;
            default: pgv_set_error("vit_attn: unsupported ablation %d", a.abl); return PGV_EINVAL;
        }
#endif
    } else {
        pgv_set_error("vit_attn: unsupported dtype %d", dtype);
        return PGV_EINVAL;
    }
#undef PGV_ATTN_LAUNCH
    PGV_HIP(hipGetLastError());
    pgv_prof_end(ctx, 1, s, 4.0 * (double)T * heads * (double)N * N * HD, 2.0 * 4.0 * (double)T * N * C);
    return PGV_OK;
}

// Dynamic-LDS opt-in of every instantiation, per device, at context creation (not lazily under a process-wide flag: ADVICE r4 on gemv.hip).
int pgv_vit_attn_configure(pgv_ctx*) {
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TF16, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TF16, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef PGV_LAB
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 48, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 72, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 120, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 64, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 48, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 72, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PGV_HIP(hipFuncSetAttribute((const void*)vit_attn_kernel<TBF16, 120, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
    return PGV_OK;
}

extern "C" int pgv_vit_attention(pgv_ctx* ctx, int dtype, const void* d_qkv, void* d_out, int T, int N, int C, int heads, void* stream) {
    PGV_CHECK(ctx && d_qkv && d_out && T > 0 && N > 0, "pgv_vit_attention: bad arguments");
    return pgv_launch_vit_attn(ctx, dtype, d_qkv, 3 * C, d_out, C, T, N, C, heads, (hipStream_t)stream);
}
