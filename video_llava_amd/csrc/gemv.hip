// LLaMA decoder, decode projections: y[B,N] = x[B,K] W[N,K]^T for 1..64 sequences on v_mfma_f32_16x16x32 with the RMSNorm around them folded
// in (HBM-bound: every weight byte is read exactly once, split-K across the 8 waves of a workgroup).  Replaces the nn.Linear calls of
// LlamaAttention / LlamaMLP / lm_head and LlamaRMSNorm at q_len == 1 (HF:llama/modeling_llama.py:53-67,96-176,347-418).
#include <stdlib.h>
#include <type_traits>

#include "llm_internal.h"

namespace {

// ---------------------------------------------------------------------------------------------
// decode GEMV on MFMA: y[b, n] = sum_k x[b, k] W[n, k], B <= 16.
// W is stored in the fragment-blocked layout (weights.h): block (n/16, k/32) is the 1 KiB a wave loads as ONE
// v_mfma_f32_16x16x32 A fragment, so every wave-load is a single contiguous, fully coalesced 1 KiB burst.
// Workgroup = NW waves, owns TL row blocks; wave w takes the 64-column groups w, w+NW, ... (adjacent 2 KiB of the same row block).
// B operand = x fragment (lane: batch l&15, k (l>>4)*8..+8) served by L2.  Partial 16x16 tiles are reduced through LDS.
// ---------------------------------------------------------------------------------------------
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
    // Batches beyond 16 (round 6): the activation operand of the CONSUMERS (qkv, gate/up, lm_head) is in the fragment-blocked activation layout
    // [K / 32][CT][4 k-groups][16 sequences][8 elements] (gv_xblk_offset): an MFMA B fragment is one contiguous 1 KiB wave-load, like a weight
    // fragment, instead of 16 rows x 64 B.  The activation operand is re-read by every workgroup and at 64 sequences it is MORE bytes on a CU's
    // load path than its weights (512 KB against 393 KB for qkv); with row-major rows the loads alone took 26 us of the 38 us qkv launch
    // (x-only ablation, LAB.md), blocked 19.  xblk: x is blocked; xgblk: the producer writes xg blocked.
    int xblk, xgblk;
#ifdef PGV_LAB
    int abl;           // lab builds only (PGV_GEMV_ABLATE; results are garbage): 1 = no x loads in the K loop, 2 = no MFMA, 4 = no weight loads
#endif
};
#ifdef PGV_LAB
#define GV_ABL(p, bit) (((p).abl & (bit)) != 0)
#else
#define GV_ABL(p, bit) false
#endif

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

// NW = kGemvWaves = waves per workgroup (the K split inside a workgroup): 8.  The residual producers cannot split K across workgroups (the workgroup that
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
// (An fp8 x fp8 MFMA form -- e4m3 weight codes straight into v_mfma_f32_16x16x32_fp8_fp8 against a hi + lo e4m3 image of the activations -- was
// built and measured in round 4: slower (one more launch per GEMV) and 2.6 x less accurate than this weight-only form; removed in round 5,
// the study is in LAB.md.)
// register-buffer depth (64-column groups per buffer): one row block per workgroup buffers 2 groups, 2 - 3 row blocks 1; an fp8 group is one 16-byte
// load per lane and row block where a 16-bit group is two, so the fp8 variants buffer twice as many groups for the same bytes in flight; batches
// beyond one MFMA tile keep one group per buffer (the x fragments of CT tiles fill the registers)
constexpr int gemv_pu(int tl, bool w8, int ct) { return (ct > 1 || tl >= 6) ? 1 : (tl == 1 ? 2 : 1) * (w8 ? 2 : 1); }
constexpr int kGemvWaves = 8;          // waves per workgroup = K phases of a row block (the arithmetic of every launch shape is defined on these 8 chains)

template <typename T, int MODE, bool W8, int TL, bool X2, int CT = 1>
__global__ __launch_bounds__(kGemvWaves * 64) void gemv_mfma_kernel(GemvArgs p) {
    constexpr int NW = kGemvWaves, PU = gemv_pu(TL, W8, CT);
    constexpr int TILES = TL;
    static_assert(!X2 || CT == 1, "the merged x load uses the lanes of columns 8..15");
    // HS (six row blocks and more): the weight stream is pipelined in HALF groups (one 32-column k-block per register buffer) -- a whole group of
    // six row blocks x four column tiles would not fit the register file next to its 96 accumulator registers.  Same order of the MFMAs.
    constexpr bool HS = (TL >= 6) && !W8;
    static_assert(!HS || (!W8 && PU == 1 && !X2), "half-group pipelining: 16-bit weights, one group per buffer");
    // HS8: the same for fp8 weights.  An fp8 group (64 columns) is ONE 16-byte load per lane and row block, so the weights are buffered per group and
    // only the x fragments per k-block: 2 x (TL + CT) x 4 buffer registers instead of 2 x (TL + 2 CT) x 4.  Same order of the MFMAs.
    constexpr bool HS8 = (TL >= 6) && W8;
    static_assert(!HS8 || (PU == 1 && !X2), "half-group x pipelining: one group per buffer");
    static_assert(MODE != GV_RESIDNORM || TL == 1, "a residual producer owns one row block");
    // The partial tiles of the NW waves meet in LDS, at most 16 tiles (128 KB) at a time: CTR column tiles per round.
    constexpr int CTR = (TILES * CT <= 16) ? CT : (16 / TILES);
    static_assert(CTR >= 1 && CT % CTR == 0, "column tiles per reduction round");
    constexpr int RND = CT / CTR;
    __shared__ f32x4_t red[NW][TILES * CTR][64];
    __shared__ f32x4_t ssq_red[NW][4 * CT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int kblocks = p.K >> 5;
    // row block of tile t.  SwiGLU: TILES / 2 output column blocks per workgroup, each a (gate, up) pair of row blocks; a ragged last workgroup
    // (pair count not a multiple of TILES / 2) re-reads the last pair and stores nothing for it.
    const int npairs = p.N >> 5;
    auto rb_of = [&](int t) -> int {
        if constexpr (MODE == GV_SWIGLU) {
            static_assert(TILES % 2 == 0, "SwiGLU: (gate, up) row-block pairs");
            const int i0 = min((int)blockIdx.x * (TILES / 2) + (t >> 1), npairs - 1) * 16;
            const int base = (i0 >> 5) * 64 + (i0 & 31);         // packed gate row (multiple of 16)
            return (base + (t & 1) * 32) >> 4;                   // gate row block / matching up row block
        } else {
            return (int)blockIdx.x * TILES + t;                  // may lie past the matrix in a ragged last workgroup (lm_head with 8 row blocks)
        }
    };
    const int nrb = (p.N + 15) >> 4;
    int rb[TILES];                                               // the row block whose weights tile t STREAMS: clamped, a tile past the matrix stores nothing
#pragma unroll
    for (int t = 0; t < TILES; ++t) rb[t] = min(rb_of(t), nrb - 1);
    // folded RMSNorm, consumer side: this thread's share of the sum-of-squares partials (L2 hits), requested before the weight stream
    // (three independent loads whose first USE is after the weight loop: summing them here would park the wave on an L2 round trip
    // before its first weight load -- measured +1.4 us on the 18 us qkv GEMV)
    constexpr int SSQ_LD = 3;                                   // 3 x NW x 64 float4 >= 4 x hidden / 16 up to hidden 6144
    // ... except in the widest shapes (4 row blocks x 4 column tiles): there the 48 registers the partials would occupy across the weight loop
    // are what decides between 256 VGPRs and a spill, and a 60 us launch does not notice the round trip: requested after the loop.
    constexpr bool SSQ_LATE = (CT * TILES >= 16);
    f32x4_t ssq_ld[CT][SSQ_LD];
    const bool scaled = (MODE == GV_STORE16 || MODE == GV_SWIGLU || MODE == GV_F32) && p.ssq_in != nullptr;
    if (!SSQ_LATE && scaled) {
        const int n4 = p.nparts_in * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < SSQ_LD; ++q) ssq_ld[ct][q] = ((const f32x4_t*)(p.ssq_in + (size_t)ct * p.ssq_ts))[min(tid + q * NW * 64, n4 - 1)];   // chunk (= 4 batch columns) index & 3 is fixed per thread
    }
    // producer side: the old residual and gamma are requested up front as well (by every wave; wave 0 consumes them in the epilogue)
    // (wave w finishes column tile w in the epilogue: it requests that tile's old residual only)
    f32x4_t r_old = {0.f, 0.f, 0.f, 0.f}, g_nx = {0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == GV_RESIDNORM) {
        const int n0p = blockIdx.x * 16 + kg * 4;
        r_old = *(const f32x4_t*)(p.out + ((size_t)min(min(w, CT - 1) * 16 + l15, p.B - 1) * p.ldo + n0p) * 4);
        g_nx = *(const f32x4_t*)(p.gamma + n0p);
    }
    // x fragment (MFMA B operand: lane = batch column l15, 8 consecutive k): an MFMA tile has 16 batch columns; the lanes of the columns
    // >= B point outside the buffer descriptor, so they cost no request on the load path (with B = 8 half of every 1 KiB wave-load; the
    // activations are re-read by every workgroup and the per-CU load path, not HBM, is what the fp8 GEMVs run into).  Their zeros only
    // feed output columns that are never stored.
    const bool xblk = p.xblk != 0;
    const __amdgpu_buffer_rsrc_t xrs = gv_make_rsrc(p.x, xblk ? (unsigned)((size_t)kblocks * CT * 1024) : (unsigned)(((size_t)(p.B - 1) * p.ldx + p.K) * 2));
    unsigned xvo[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        xvo[ct] = xblk ? (ct * 16 < p.B ? (unsigned)((ct * 64 + lane) * 16) : 0x80000000u)          // whole tiles beyond B: outside the descriptor, no request
                       : (ct * 16 + l15 < p.B ? (unsigned)(((size_t)(ct * 16 + l15) * p.ldx + kg * 8) * 2) : 0x80000000u);
    const unsigned xkb = xblk ? CT * 1024u : 64u;              // bytes between consecutive k-blocks of one column tile
    auto xload = [&](size_t kb, int ct) -> typename T::v8 {
        return __builtin_bit_cast(typename T::v8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[ct] + (unsigned)kb * xkb, 0, 0));
    };
    f32x4_t acc[TILES][CT];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[t][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (HS) {
        // steps i = 0, 1, ...: step i is k-block h = i & 1 of this wave's group w + NW * (i >> 1); every wave runs 2 * ceil(groups / NW) steps, a
        // step past the end of K re-reads the last k-block against an all-zero x fragment (exact zeros, as below).
        const int kb_end = kblocks, j_end = (kb_end + 1) >> 1;
        const int nsteps = 2 * ((j_end + NW - 1) / NW);
        const char* wp[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) wp[t] = p.W + ((size_t)rb[t] * kblocks) * 1024 + lane * 16;
        auto loadh = [&](typename T::v8 (&wf)[TILES], typename T::v8 (&xf)[CT], int i) {
            const int kb = min(2 * (w + NW * (i >> 1)) + (i & 1), kb_end - 1);
            if (!GV_ABL(p, 4)) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) wf[t] = __builtin_nontemporal_load((const typename T::v8*)(wp[t] + (size_t)kb * 1024));
            }
            if (!GV_ABL(p, 1)) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xf[ct] = xload((size_t)kb, ct);
            }
        };
        auto mmah = [&](typename T::v8 (&wf)[TILES], typename T::v8 (&xf)[CT], int i, bool last) {
            typename T::v8 xv[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) xv[ct] = xf[ct];
            if (last) {
                const typename T::v8 z = {};
                const bool past = 2 * (w + NW * (i >> 1)) + (i & 1) >= kb_end;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xv[ct] = past ? z : xv[ct];
            }
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                if (GV_ABL(p, 2)) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[t][ct][0] += (float)wf[t][0] + (float)xv[ct][0];
                    continue;
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[t][ct] = T::mfma16(wf[t], xv[ct], acc[t][ct]);
            }
        };
        typename T::v8 wa[TILES] = {}, wb[TILES] = {}, xa[CT] = {}, xb[CT] = {};
        loadh(wa, xa, 0);
        int i = 0;
        for (; i + 2 < nsteps; i += 2) {                          // the last two steps are the only ones that can lie past the end of K
            loadh(wb, xb, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            mmah(wa, xa, i, false);
            __builtin_amdgcn_sched_barrier(0);
            loadh(wa, xa, i + 2);
            __builtin_amdgcn_sched_barrier(0);
            mmah(wb, xb, i + 1, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        loadh(wb, xb, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        mmah(wa, xa, i, true);
        mmah(wb, xb, i + 1, true);
    } else if constexpr (HS8) {
        // group gi of this wave = 64 columns w + NW * gi, gi = 0 .. ng - 1 (every wave runs ng = ceil(groups / NW) of them); step 2 gi + h multiplies
        // k-block h of the group.  A group past the end of K re-reads the last one against x fragments requested OUTSIDE the buffer descriptor
        // (all-zero, no request): exact zeros, no predicated loads, no select in front of the MFMAs.
        const int j_end = p.K >> 6;                               // fp8 weights: K % 64 == 0
        const int ng = ((j_end + NW - 1) / NW + 1) & ~1;         // groups per wave, rounded up to whole PAIRS (hidden 4096 / 5120: 8 / 10, nothing added)
        // (row block, group) addresses are wave-uniform: scalar base + one shared 32-bit lane offset per load, no per-tile pointer registers
        const unsigned lane_off = (unsigned)lane * 16u;
        auto loadw = [&](u32x4_t (&wf)[TILES], int gi) {
            const int g = min(w + NW * gi, j_end - 1);
            if (!GV_ABL(p, 4)) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const char* sp = p.W + ((size_t)rb[t] * j_end + g) * 1024;
                    wf[t] = __builtin_nontemporal_load((const u32x4_t*)(sp + lane_off));
                }
            }
        };
        auto loadx = [&](typename T::v8 (&xf)[CT], int gi, int h) {
            const int g = w + NW * gi;
            const bool past = g >= j_end;                         // wave-uniform
            if (!GV_ABL(p, 1)) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    xf[ct] = __builtin_bit_cast(typename T::v8, __builtin_amdgcn_raw_buffer_load_b128(xrs, past ? 0x80000000u : xvo[ct] + (unsigned)(2 * g + h) * xkb, 0, 0));
            }
        };
        auto mma8 = [&](u32x4_t (&wf)[TILES], typename T::v8 (&xf)[CT], int h) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const typename T::v8 wv = fp8x8_to_v8<T>(wf[t][2 * h], wf[t][2 * h + 1]);      // widened once, used by every column tile
                if (GV_ABL(p, 2)) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[t][ct][0] += (float)wv[0] + (float)xf[ct][0];
                    continue;
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[t][ct] = T::mfma16(wv, xf[ct], acc[t][ct]);
            }
        };
        u32x4_t wa[TILES] = {}, wb[TILES] = {};
        typename T::v8 xa[CT] = {}, xb[CT] = {};
        loadw(wa, 0);
        loadx(xa, 0, 0);
        int gi = 0;
        for (; gi + 2 < ng; gi += 2) {
            loadx(xb, gi, 1); loadw(wb, gi + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma8(wa, xa, 0);
            __builtin_amdgcn_sched_barrier(0);
            loadx(xa, gi + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma8(wa, xb, 1);
            __builtin_amdgcn_sched_barrier(0);
            loadx(xb, gi + 1, 1); loadw(wa, gi + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma8(wb, xa, 0);
            __builtin_amdgcn_sched_barrier(0);
            loadx(xa, gi + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma8(wb, xb, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the last two groups: wa / xa hold group gi, k-block 0
        loadx(xb, gi, 1); loadw(wb, gi + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma8(wa, xa, 0);
        __builtin_amdgcn_sched_barrier(0);
        loadx(xa, gi + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma8(wa, xb, 1);
        __builtin_amdgcn_sched_barrier(0);
        loadx(xb, gi + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma8(wb, xa, 0);
        mma8(wb, xb, 1);
    } else {
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
                    if ((!W8 || h == 0) && !GV_ABL(p, 4)) {
#pragma unroll
                        for (int t = 0; t < TILES; ++t)
                            wf[u][W8 ? 0 : h][t] = __builtin_nontemporal_load((const wreg_t*)(wp[t] + (size_t)(W8 ? min(g, j_end - 1) : kb) * 1024));
                    }
                    if (GV_ABL(p, 1)) continue;
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
                        typename T::v8 wv;
                        if constexpr (W8) wv = fp8x8_to_v8<T>(wf[u][0][t][2 * h], wf[u][0][t][2 * h + 1]);      // widened once, used by every column tile
                        else wv = wf[u][h][t];
                        if (GV_ABL(p, 2)) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) acc[t][ct][0] += (float)wv[0] + (float)xv[h][ct][0];
                            continue;
                        }
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[t][ct] = T::mfma16(wv, xv[h][ct], acc[t][ct]);
                    }
            }
        };
        wreg_t wa[PU][WH][TILES] = {}, wb[PU][WH][TILES] = {};
        typename T::v8 xa[PU][XH][CT] = {}, xb[PU][XH][CT] = {};
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
    if (SSQ_LATE && scaled) {
        const int n4 = p.nparts_in * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < SSQ_LD; ++q) ssq_ld[ct][q] = ((const f32x4_t*)(p.ssq_in + (size_t)ct * p.ssq_ts))[min(tid + q * NW * 64, n4 - 1)];
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
    // Cross-wave reduction + epilogue (round 6: spread over the waves).  Every (tile, column tile) unit -- a (gate, up) tile pair for SwiGLU -- is
    // summed over the waves in wave order and finished by ONE wave, unit u by wave u % NW (until round 5 wave 0 did all of them, up to 16 tiles of
    // 8 KB, while the other seven waves had already left: a serial tail on every launch).  The order of the additions inside a unit is unchanged.
    constexpr int UPT = (MODE == GV_SWIGLU) ? TILES / 2 : TILES;       // units per column tile
    auto tile_sum = [&](int t, int ctl) -> f32x4_t {
        f32x4_t tot = red[0][t * CTR + ctl][lane];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) tot += red[ww][t * CTR + ctl][lane];
        return tot;
    };
#pragma unroll
    for (int r = 0; r < RND; ++r) {
        if (r > 0) __syncthreads();                               // the previous round's tiles have been read
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int ctl = 0; ctl < CTR; ++ctl) red[w][t * CTR + ctl][lane] = acc[t][r * CTR + ctl];
        __syncthreads();
        for (int u = w; u < UPT * CTR; u += NW) {                 // wave-uniform
            const int ut = u / CTR, ctl = u - ut * CTR, ct = r * CTR + ctl;
            const int b = ct * 16 + l15;                          // lane holds D[n = kg*4 + r][b]
            float rstd = 1.f;
            if (scaled) {
                float ss = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) ss += ssq_red[ww][ct * 4 + (l15 >> 2)][l15 & 3];       // fixed order: deterministic
                rstd = rsqrtf(ss * p.inv_h + p.eps);
            }
            if constexpr (MODE == GV_SWIGLU) {
                const int pair = (int)blockIdx.x * (TILES / 2) + ut;
                f32x4_t tg = tile_sum(2 * ut, ctl), tu = tile_sum(2 * ut + 1, ctl);
                if constexpr (W8) {                               // per-row power-of-two scale: exact in fp32
                    tg *= *(const f32x4_t*)(p.wscale + rb_of(2 * ut) * 16 + kg * 4);
                    tu *= *(const f32x4_t*)(p.wscale + rb_of(2 * ut + 1) * 16 + kg * 4);
                }
                if (scaled) { tg *= rstd; tu *= rstd; }
                if (b < p.B && pair < npairs) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float g = tg[e]; v[e] = g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.4426950408889634f)) * tu[e]; }   // same SiLU as gemm.hip
                    *(u32x2_t*)(p.out + ((size_t)b * p.ldo + pair * 16 + kg * 4) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            } else {
                const int rbt = rb_of(ut);
                if (rbt >= nrb) continue;                         // wave-uniform: a tile past the matrix (ragged last workgroup)
                const int n0 = rbt * 16 + kg * 4;
                f32x4_t tot = tile_sum(ut, ctl);
                if constexpr (W8) tot *= *(const f32x4_t*)(p.wscale + rbt * 16 + kg * 4);
                if (scaled) tot *= rstd;
                if constexpr (MODE == GV_RESIDNORM) {
                    // r = resid + y (fp32, written back); xg = round16(r * gamma); sum of r^2 over this workgroup's 16 rows per batch column
                    float sq = 0.f;
                    if (b < p.B) {
                        f32x4_t* rp = (f32x4_t*)(p.out + ((size_t)b * p.ldo + n0) * 4);
                        const f32x4_t rr = r_old + tot;           // (wave w == unit ct: its r_old is this tile's)
                        *rp = rr;
                        const f32x4_t g = g_nx;
                        *(u32x2_t*)(p.xg + (p.xgblk ? gv_xblk_offset(b, n0, CT) : ((size_t)b * p.ldo + n0) * 2)) = pack4<T>(rr[0] * g[0], rr[1] * g[1], rr[2] * g[2], rr[3] * g[3]);
                        sq = (rr[0] * rr[0] + rr[1] * rr[1]) + (rr[2] * rr[2] + rr[3] * rr[3]);
                    }
                    sq = rows_sum_to_row3(sq);                    // the four rows of 16 lanes (the 16 output rows of this workgroup), no LDS round trips
                    if (kg == 3) p.ssq_out[(size_t)ct * p.ssq_ts + (size_t)blockIdx.x * 16 + l15] = sq;               // columns >= B carry 0
                    continue;
                }
                if constexpr (MODE == GV_F32) {
                    if (p.amax_val != nullptr) {                  // wave-uniform
                        float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float v = tot[e]; if (n0 + e < p.N && v > bv) { bv = v; bi = n0 + e; } }      // ascending n: the first maximum wins, NaN never does
#pragma unroll
                        for (int o = 16; o <= 32; o <<= 1) {
                            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                        }
                        if (kg == 0) { p.amax_val[(size_t)ct * p.amax_ts + (size_t)rbt * 16 + l15] = bv; p.amax_idx[(size_t)ct * p.amax_ts + (size_t)rbt * 16 + l15] = bi; }
                    }
                }
                if (b >= p.B) continue;
                if constexpr (MODE == GV_STORE16) {
                    *(u32x2_t*)(p.out + ((size_t)b * p.ldo + n0) * 2) = pack4<T>(tot[0], tot[1], tot[2], tot[3]);
                } else if constexpr (MODE == GV_RESID) {
                    f32x4_t* rp = (f32x4_t*)(p.out + ((size_t)b * p.ldo + n0) * 4);
                    *rp = *rp + tot;
                } else if constexpr (MODE == GV_F32) {
                    float* op = (float*)p.out + (size_t)b * p.ldo + n0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n0 + e < p.N) op[e] = tot[e];
                }
            }
        }
    }
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
// Sum of the eight phase tiles of row block rb in phase order + the epilogue of gemv_mfma_kernel<GV_RESIDNORM>, statement for statement (one wave).
template <typename T, bool W8, int CT>
__device__ __forceinline__ void k8_finish_row_block(const GemvArgs& p, const f32x4_t* __restrict__ part, int rb, int lane, const f32x4_t (&r_old)[CT], f32x4_t g_nx) {
    const int l15 = lane & 15, kg = lane >> 4;
    const int n0 = rb * 16 + kg * 4;
    const f32x4_t* src = part + ((size_t)rb * 8 * CT) * 64 + lane;
    f32x4_t tile[CT][8];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) tile[ct][ww] = src[(ww * CT + ct) * 64];      // all loads in flight before the first add
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x4_t tot = tile[ct][0];
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) tot += tile[ct][ww];
        if constexpr (W8) tot *= *(const f32x4_t*)(p.wscale + rb * 16 + kg * 4);
        const int b = ct * 16 + l15;
        float sq = 0.f;
        if (b < p.B) {
            f32x4_t* rp = (f32x4_t*)(p.out + ((size_t)b * p.ldo + n0) * 4);
            const f32x4_t r = r_old[ct] + tot;
            *rp = r;
            const f32x4_t g = g_nx;
            *(u32x2_t*)(p.xg + (p.xgblk ? gv_xblk_offset(b, n0, CT) : ((size_t)b * p.ldo + n0) * 2)) = pack4<T>(r[0] * g[0], r[1] * g[1], r[2] * g[2], r[3] * g[3]);
            sq = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
        }
        sq = rows_sum_to_row3(sq);
        if (kg == 3) p.ssq_out[(size_t)ct * p.ssq_ts + (size_t)rb * 16 + l15] = sq;
    }
}

// (Round 5 also built a FUSED form -- no finish launch: phase tiles published with agent-scope relaxed atomic stores, a ticket per row group,
// the eighth arriver runs the finish -- bitwise equal, but the hand-off through the memory side costs more than the 4.8 us finish launch:
// 13B fp8 6.53 -> 6.14 videos/s, 7B at 32 / 64 clips 16.5 -> 15.9 / 19.3 -> 17.7.  Removed; numbers in LAB.md.)
template <typename T, bool W8, int CT, int NWB>
__global__ __launch_bounds__(NWB * 64) void gemv_k8_kernel(GemvArgs p, f32x4_t* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char xs[];    // [groups of this pass][B rows][128 B], 16-B chunks XORed with (row & 7)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    // (row group, phase) of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (id & 7); when the row-group count is a multiple
    // of 8 the eight phases of a row group are given ids that share an XCD, so their x slices and the fused hand-off stay inside one L2's reach.
    const int R = (int)gridDim.x >> 3;
    int j, rg;
    if ((R & 7) == 0) { const int q = (int)blockIdx.x >> 3; rg = ((int)blockIdx.x & 7) + 8 * (q >> 3); j = q & 7; }
    else { j = (int)blockIdx.x & 7; rg = (int)blockIdx.x >> 3; }
    const int rb = rg * NWB + w;
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
    constexpr int PU = 4;                                        // groups per register buffer (3 / 4 / 6 at 64 sequences: 14.7 / 15.3 / 15.5 us per launch, gpurun_out/r6n: not the lever)
    auto wload = [&](wreg_t (&wf)[PU][WH], int gi0) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int g = min(j + 8 * (gi0 + u), j_end - 1);
#pragma unroll
            for (int h = 0; h < WH; ++h)
                if (!GV_ABL(p, 4)) wf[u][h] = __builtin_nontemporal_load((const wreg_t*)(wp + (size_t)(W8 ? g : min(2 * g + h, kblocks - 1)) * 1024));
        }
    };
    for (int pass0 = 0; pass0 < gpw; pass0 += gpp) {
        const int ng = min(gpp, gpw - pass0);
        wreg_t wa[PU][WH] = {}, wb[PU][WH] = {};
        wload(wa, pass0);                                        // the first weight batch is in flight while the x slice is staged
        __syncthreads();                                         // the previous pass has been read by every wave
        if (GV_ABL(p, 1)) {
        } else if ((kblocks & 1) == 0) {
            // LDS-DMA (round 6): one wave-instruction lands 8 rows x 128 B = 1 KiB linearly in LDS, every lane fetching the 16-byte chunk that
            // belongs at its position (the XOR swizzle is applied on the SOURCE side); all instructions of a pass are in flight together.  (The
            // first version loaded 16 bytes per thread into a register and stored it, an L2 round trip per loop iteration: 12 dependent
            // iterations per pass at four column tiles -- 18.5 -> 15.2 us per launch at 64 sequences.)  Rows >= B re-read the last sequence:
            // their tile columns are never stored.  (Staging the next pass behind the running one -- two half buffers, the DMA issued by inline
            // asm so that hipcc does not drain the weight prefetch in front of every fragment read -- was built and measured: no gain, 15.3 us.)
            const int srow = lane >> 3, pos = lane & 7;
            for (int q = w; q < ng * Brows / 8; q += NWB) {
                const int fr = q * 8 + srow, gi = fr / Brows, row = fr - gi * Brows;
                const int g = j + 8 * (pass0 + gi), ch = pos ^ (row & 7);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.x + ((size_t)min(row, p.B - 1) * p.ldx + (size_t)g * 64 + ch * 8) * 2),
                                                 (__attribute__((address_space(3))) void*)(xs + (size_t)q * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {                                                 // K % 64 == 32: the last group holds one 32-block, its other half must read as zeros
            for (int c = tid; c < ng * Brows * 8; c += NWB * 64) {   // 16-byte chunks: (group, row, chunk)
                const int ch = c & 7, row = (c >> 3) % Brows, gi = (c >> 3) / Brows;
                const int g = j + 8 * (pass0 + gi), kb = 2 * g + (ch >> 2);
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (row < p.B && kb < kblocks) v = *(const u32x4_t*)(p.x + ((size_t)row * p.ldx + (size_t)g * 64 + ch * 8) * 2);
                *(u32x4_t*)(xs + ((size_t)(gi * Brows + row) * 128 + ((ch ^ (row & 7)) << 4))) = v;
            }
        }
        __syncthreads();
        auto mma = [&](wreg_t (&wf)[PU][WH], int gi0) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int gi = gi0 + u - pass0;
                if (gi < ng && !GV_ABL(p, 2)) {
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
    for (int ct = 0; ct < CT; ++ct)
        if (!GV_ABL(p, 8)) dst[ct * 64] = acc[ct];
}

// the finish launch: one wave per row block
template <typename T, bool W8, int CT>
__global__ __launch_bounds__(64) void gemv_k8_finish_kernel(GemvArgs p, const f32x4_t* __restrict__ part) {
    const int lane = threadIdx.x, l15 = lane & 15, kg = lane >> 4;
    const int rb = blockIdx.x;
    const int n0 = rb * 16 + kg * 4;
    f32x4_t r_old[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) r_old[ct] = *(const f32x4_t*)(p.out + ((size_t)min(ct * 16 + l15, p.B - 1) * p.ldo + n0) * 4);
    const f32x4_t g_nx = *(const f32x4_t*)(p.gamma + n0);
    k8_finish_row_block<T, W8, CT>(p, part, rb, lane, r_old, g_nx);
}

// decode: resid[b] = embed[tok[b]] plus the producer side of the folded RMSNorm (see GemvArgs): xg = round16(resid * gamma of layer 0's
// input norm), ssq[0][b] = sum resid^2 (one partial per sequence).
// xct > 0: xg is written in the fragment-blocked activation layout with xct column tiles (gv_xblk_offset); 0: row-major [B][H].
template <typename T>
__global__ __launch_bounds__(256) void embed_tok_norm_kernel(const int* __restrict__ tok, const typename T::elem* __restrict__ embed, float* __restrict__ resid,
                                                             const float* __restrict__ gamma, typename T::elem* __restrict__ xg, float* __restrict__ ssq, int H, int ssq_ts, int xct) {
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
        char* xo = (char*)xg + (xct ? gv_xblk_offset(b, c, xct) : ((size_t)b * H + c) * 2);       // 8 consecutive features: 16 contiguous bytes in either layout
        *(u32x2_t*)xo = pack4<T>(a[0] * g0[0], a[1] * g0[1], a[2] * g0[2], a[3] * g0[3]);
        *(u32x2_t*)(xo + 8) = pack4<T>(bb[0] * g1[0], bb[1] * g1[1], bb[2] * g1[2], bb[3] * g1[3]);
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
                                                              float* __restrict__ ssq, int H, int ssq_ts, int xct) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float ss = 0.f;
    for (int c = threadIdx.x * 4; c < H; c += 256 * 4) {
        const f32x4_t r = *(const f32x4_t*)(resid + (size_t)b * H + c), g = *(const f32x4_t*)(gamma + c);
        *(u32x2_t*)((char*)xg + (xct ? gv_xblk_offset(b, c, xct) : ((size_t)b * H + c) * 2)) = pack4<T>(r[0] * g[0], r[1] * g[1], r[2] * g[2], r[3] * g[3]);
        ss += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) ssq[(size_t)(b >> 4) * ssq_ts + (b & 15)] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

// Column tiles of the blocked activation layout for a batch of B sequences (0: row-major -- up to 8 sequences the merged row loads of the
// narrow kernels, one full line per sequence and group, are cheaper).  The ONE place that decides it: producers and consumers must agree.
int pgv_gemv_xblk_tiles(int B) { return B <= 8 ? 0 : (B <= 16 ? 1 : (B <= 32 ? 2 : 4)); }

int pgv_launch_final_prep(int dtype, const float* resid, const float* gamma, void* xg, float* ssq, int B, int H, hipStream_t s, bool x_blocked) {
    PGV_CHECK(H % 32 == 0, "final_prep: hidden %d unsupported", H);
    const int xct = x_blocked ? pgv_gemv_xblk_tiles(B) : 0;
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((resid_norm_prep_kernel<T>), dim3(B), dim3(256), 0, s, resid, gamma, (typename T::elem*)xg, ssq, H, H, xct));
    return PGV_OK;
}

int pgv_launch_embed_tok_norm(int dtype, const int* tok, const void* embed, float* resid, const float* gamma, void* xg, float* ssq, int B, int H, hipStream_t s, bool x_blocked) {
    PGV_CHECK(H % 32 == 0, "embed: hidden must be a multiple of 32");
    const int xct = x_blocked ? pgv_gemv_xblk_tiles(B) : 0;
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((embed_tok_norm_kernel<T>), dim3(B), dim3(256), 0, s, tok, (const typename T::elem*)embed, resid, gamma,
                                                    (typename T::elem*)xg, ssq, H, H, xct));
    return PGV_OK;
}

// ---------------------------------------------------------------------------------------------
// launch-shape selection: ONE table of instantiations keyed on (epilogue, row blocks per workgroup, column tiles) x (fp8 weights, merged x load),
// and one function that picks the key from (mode, w8, B, N, K, CUs).  Nothing here depends on anything but the model shape and the batch's
// column-tile count, and every shape computes bit-identical results per batch column (tests/test_gpu_llm.py), so the choice is performance only.
// ---------------------------------------------------------------------------------------------
// Launch-shape A/B switches exist in the lab library only (-DPGV_LAB, never loaded by the product): PGV_GEMV_TL3, PGV_GEMV_X2, PGV_GEMV_K8,
// PGV_GEMV_K8_NARROW_MINK.  The release library uses the defaults.
static int lab_switch(const char* name, int dflt) {
#ifdef PGV_LAB
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

using gemv_launch_fn = int (*)(int dtype, int grid, const GemvArgs& a, hipStream_t s);

template <int MODE, bool W8, int TL, bool X2, int CT>
static int launch_variant(int dtype, int grid, const GemvArgs& a, hipStream_t s) {
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_mfma_kernel<T, MODE, W8, TL, X2, CT>), dim3(grid), dim3(kGemvWaves * 64), 0, s, a));
    return PGV_OK;
}

template <int MODE, bool W8, int TL, int CT>
constexpr gemv_launch_fn x2_variant() {          // the merged x load (B <= 8) uses the lanes of batch columns 8..15: one column tile only, not the half-group pipeline
    if constexpr (CT == 1 && TL < 6) return &launch_variant<MODE, W8, TL, true, 1>;
    else return nullptr;
}

struct GemvShape {
    int mode, tl, ct;
    gemv_launch_fn fn[2][2];                     // [fp8 weights][merged x load]
};
template <int MODE, int TL, int CT>
constexpr GemvShape gemv_shape() {
    if constexpr (TL >= 6)                       // six row blocks and more: no merged x load (half-group pipelining for 16-bit weights; an fp8 group is ONE load per lane: whole groups)
        return {MODE, TL, CT, {{&launch_variant<MODE, false, TL, false, CT>, nullptr}, {&launch_variant<MODE, true, TL, false, CT>, nullptr}}};
    else
        return {MODE, TL, CT, {{&launch_variant<MODE, false, TL, false, CT>, x2_variant<MODE, false, TL, CT>()},
                               {&launch_variant<MODE, true, TL, false, CT>, x2_variant<MODE, true, TL, CT>()}}};
}
static const GemvShape kGemvShapes[] = {
    // one MFMA tile of sequences (B <= 16)
    gemv_shape<GV_STORE16, 1, 1>(), gemv_shape<GV_STORE16, 3, 1>(), gemv_shape<GV_STORE16, 4, 1>(), gemv_shape<GV_RESID, 1, 1>(), gemv_shape<GV_SWIGLU, 2, 1>(),
    gemv_shape<GV_F32, 1, 1>(), gemv_shape<GV_F32, 8, 1>(), gemv_shape<GV_RESIDNORM, 1, 1>(),
    // 2 / 4 column tiles (B <= 32 / 64)
    gemv_shape<GV_STORE16, 1, 2>(), gemv_shape<GV_STORE16, 3, 2>(), gemv_shape<GV_STORE16, 4, 2>(), gemv_shape<GV_RESID, 1, 2>(), gemv_shape<GV_SWIGLU, 2, 2>(),
    gemv_shape<GV_SWIGLU, 4, 2>(), gemv_shape<GV_SWIGLU, 6, 2>(), gemv_shape<GV_SWIGLU, 8, 2>(), gemv_shape<GV_F32, 1, 2>(), gemv_shape<GV_F32, 8, 2>(),
    gemv_shape<GV_RESIDNORM, 1, 2>(),
    gemv_shape<GV_STORE16, 1, 4>(), gemv_shape<GV_STORE16, 3, 4>(), gemv_shape<GV_STORE16, 4, 4>(), gemv_shape<GV_RESID, 1, 4>(), gemv_shape<GV_SWIGLU, 2, 4>(),
    gemv_shape<GV_SWIGLU, 4, 4>(), gemv_shape<GV_SWIGLU, 6, 4>(), gemv_shape<GV_SWIGLU, 8, 4>(), gemv_shape<GV_F32, 1, 4>(), gemv_shape<GV_F32, 8, 4>(),
    gemv_shape<GV_RESIDNORM, 1, 4>(),
};

// The 8-phase form of a residual producer with CT column tiles and NWB row blocks (= waves) per workgroup; `a` is complete except lds_bytes.
// nrb = row blocks of the matrix (N / 16, a multiple of NWB): (nrb / NWB) row groups x 8 phases workgroups, then the finish launch.
template <bool W8, int CT, int NWB>
static int launch_k8(int dtype, GemvArgs a, int nrb, void* k8_part, hipStream_t s) {
    const int gpw_max = ((a.K / 32 + 1) / 2 + 7) / 8;
    const unsigned budget = 96u * 1024u, per_group = (unsigned)CT * 16u * 128u;
    unsigned lds = (unsigned)gpw_max * per_group;
    if (lds > budget) lds = budget / per_group * per_group;
    a.lds_bytes = lds;
    f32x4_t* part = (f32x4_t*)k8_part;
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_k8_kernel<T, W8, CT, NWB>), dim3(nrb / NWB * 8), dim3(NWB * 64), lds, s, a, part));
    if (GV_ABL(a, 16)) return PGV_OK;
    PGV_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((gemv_k8_finish_kernel<T, W8, CT>), dim3(nrb), dim3(64), 0, s, a, (const f32x4_t*)part));
    return PGV_OK;
}
using k8_launch_fn = int (*)(int dtype, GemvArgs a, int nrb, void* k8_part, hipStream_t s);
static const k8_launch_fn kK8[2][3][2] = {      // [fp8 weights][column tiles 1 / 2 / 4][row blocks per workgroup 8 / 10]
    {{&launch_k8<false, 1, 8>, &launch_k8<false, 1, 10>}, {&launch_k8<false, 2, 8>, &launch_k8<false, 2, 10>}, {&launch_k8<false, 4, 8>, &launch_k8<false, 4, 10>}},
    {{&launch_k8<true, 1, 8>, &launch_k8<true, 1, 10>}, {&launch_k8<true, 2, 8>, &launch_k8<true, 2, 10>}, {&launch_k8<true, 4, 8>, &launch_k8<true, 4, 10>}}};

// The 8-phase kernels stage up to 96 KB of x in dynamic LDS: the opt-in attribute is per function AND per device, so it is set for every
// instantiation when a context is created on a device (pgv_ctx_create), never lazily inside a launch (which may sit in a graph capture).
template <bool W8, int CT, int NWB>
static int configure_k8() {
    PGV_HIP(hipFuncSetAttribute((const void*)gemv_k8_kernel<TF16, W8, CT, NWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    PGV_HIP(hipFuncSetAttribute((const void*)gemv_k8_kernel<TBF16, W8, CT, NWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    return PGV_OK;
}
template <bool W8, int CT>
static int configure_k8_row() { PGV_TRY((configure_k8<W8, CT, 8>())); PGV_TRY((configure_k8<W8, CT, 10>())); return PGV_OK; }
int pgv_gemv_configure(pgv_ctx*) {
    PGV_TRY((configure_k8_row<false, 1>())); PGV_TRY((configure_k8_row<false, 2>())); PGV_TRY((configure_k8_row<false, 4>()));
    PGV_TRY((configure_k8_row<true, 1>())); PGV_TRY((configure_k8_row<true, 2>())); PGV_TRY((configure_k8_row<true, 4>()));
    return PGV_OK;
}

struct GemvChoice { int tl, ct; bool x2, k8; int nwb; };

// TL = row blocks per workgroup (they share the x fragments), CT = column tiles, x2 = merged x load, k8 = the 8-phase residual producer.
static GemvChoice choose_gemv(int mode, bool w8, int B, int grid, int K, int num_cu, bool have_k8_scratch) {
    GemvChoice c{1, B <= 16 ? 1 : (B <= 32 ? 2 : 4), B <= 8 && lab_switch("PGV_GEMV_X2", 1) != 0, false, 8};
    // row blocks (= waves) per 8-phase workgroup: 8, or 10 where that makes (row groups x 8 phases) exactly one workgroup per CU (13B: 320 row
    // blocks -> 32 groups x 8 = 256 workgroups of 10 waves instead of 320 of 8 = 1 1/4 rounds).  The (row block, phase) partition and the phase
    // order of the sums do not depend on it: bitwise the same result.
    // (13B fp8, 8 sequences, same box: 6.53 -> 6.64 videos/s with down_proj alone in that shape, 6.71 with o_proj as well.)
    if (lab_switch("PGV_GEMV_K8_NWB10", 1) != 0 && grid % 10 == 0 && grid / 10 * 8 == num_cu) c.nwb = 10;
    const bool k8_ok = have_k8_scratch && lab_switch("PGV_GEMV_K8", 1) != 0 && grid % c.nwb == 0;
    if (B > 16) {
        // Wide batches re-read 2 - 4 x the activation lines per workgroup, so the row blocks that share them matter more: qkv 3, gate/up two
        // (gate, up) pairs when the block counts divide.
        if (mode == GV_STORE16 && grid % 3 == 0) c.tl = 3;
        // ... four where three leave a ragged second round and four fill one (13B qkv: 960 row blocks -> 320 workgroups = 1 1/4 rounds with three,
        // 240 with four -- the choice the narrow batches make below, for the same reason)
        if (mode == GV_STORE16 && lab_switch("PGV_GEMV_WIDE_QKV4", 1) != 0 && grid % 4 == 0 && grid / 4 <= num_cu && (grid % 3 != 0 || grid / 3 > num_cu)) c.tl = 4;
        // (four column tiles: 228 VGPRs once the sum-of-squares partials are requested after the weight loop, SSQ_LATE)
        if (mode == GV_SWIGLU) c.tl = (grid % 2 == 0 && (B <= 32 || lab_switch("PGV_GEMV_WIDE_TL4", 1) != 0)) ? 4 : 2;
        // A wide workgroup holds a CU to itself (its registers and the LDS of its partial tiles), so the launch runs in ROUNDS of num_cu workgroups
        // and every workgroup moves the whole activation operand through its CU: 7B gate/up = 688 (gate, up) pairs -> 344 workgroups of two pairs =
        // two rounds (88 CUs run a second workgroup, the others wait: 50.9 us at 64 sequences), 230 workgroups of THREE pairs = one round and a third
        // fewer reads of x (LAB.md round 6).  Taken when three pairs per workgroup fit one round and two do not.
        if (mode == GV_SWIGLU && (!w8 || lab_switch("PGV_GEMV_WIDE_FP8", 1) != 0) && lab_switch("PGV_GEMV_WIDE_TL6", 1) != 0 && (grid + 2) / 3 <= num_cu && (grid + 1) / 2 > num_cu) c.tl = 6;
        // ... four pairs where three still need a second round (13B gate/up: 864 pairs -> 288 workgroups of three, 216 of four)
        else if (mode == GV_SWIGLU && (!w8 || lab_switch("PGV_GEMV_WIDE_FP8", 1) != 0) && lab_switch("PGV_GEMV_WIDE_TL8", 1) != 0 && (grid + 3) / 4 <= num_cu && (grid + 1) / 2 > num_cu) c.tl = 8;
        // lm_head: 2001 row blocks; one per workgroup means 2001 passes over the activation operand (512 KB each against 128 KB of weights: 102 us at
        // 64 sequences).  Eight per workgroup: 251 workgroups, one round.
        if (mode == GV_F32 && (!w8 || lab_switch("PGV_GEMV_WIDE_FP8", 1) != 0) && lab_switch("PGV_GEMV_WIDE_HEAD8", 1) != 0 && (grid + 7) / 8 <= num_cu) c.tl = 8;
        // narrow matrices (o_proj, down_proj): 8 row blocks x K phase per workgroup + a finish launch where it pays (kernel trace at 32 clips:
        // down_proj 32.0 -> 22.4 + 4.9 us (finish), o_proj 13.3 -> 9.6 + 4.9: the short-K matrix only gains once the batch spans four tiles)
        if (mode == GV_RESIDNORM) c.k8 = k8_ok && (K / 64) >= 16 && (B > 32 || K >= 8192);
        return c;
    }
    switch (mode) {
        case GV_STORE16:
            // three row blocks per workgroup when that puts at most ~one workgroup on every CU and nothing is left over (7B qkv: 768 -> 256;
            // 13B: 960 -> 320 workgroups, two resident per CU -- with fp8 weights a 16-row workgroup requests as many activation lines as weight
            // lines: qkv 20.7 -> 18.7 us; with 16-bit weights the same launch shape LOSES 4 us per layer, so the relaxed bound is for fp8 only)
            if (lab_switch("PGV_GEMV_TL3", 1) != 0 && grid % 3 == 0 && grid / 3 <= (w8 ? 2 : 1) * num_cu && grid / 3 >= num_cu / 2) c.tl = 3;
            // ... unless three blocks leave a ragged second round and FOUR give one round that nearly fills the chip (13B qkv: 960 row blocks ->
            // 320 workgroups = 1 1/4 rounds with three, 240 workgroups on 256 CUs with four: same box, fp8 weights 6.68 -> 6.92 videos/s and the
            // GEMV family 0.539 -> 0.570 of 8 TB/s; 16-bit weights 4.79 -> 4.84)
            if (lab_switch("PGV_GEMV_TL4", 1) != 0 && grid % 4 == 0 && grid / 4 <= num_cu && grid / 4 >= num_cu - num_cu / 8 &&
                !(grid % 3 == 0 && grid / 3 == num_cu)) c.tl = 4;
            break;
        case GV_SWIGLU: c.tl = 2; break;
        case GV_F32:
            // lm_head at 9 .. 16 sequences: 2001 workgroups of one row block each read the whole activation operand (128 KB against 128 KB of
            // weights); eight row blocks per workgroup as at the wide batches (251 workgroups, one round)
            if ((!w8 || lab_switch("PGV_GEMV_WIDE_FP8", 1) != 0) && B > 8 && lab_switch("PGV_GEMV_WIDE_HEAD8", 1) != 0 && (grid + 7) / 8 <= num_cu) c.tl = 8;
            break;
        case GV_RESIDNORM:
            // fp8 down_proj of the 13B shapes (K = 13 824, 320 row blocks): a 16-row workgroup requests as many activation lines as weight lines
            // (LAB.md "what the activation operand costs"); the 8-phase form reads 1/8 of x per workgroup and is bitwise the same result
            // With 10 row blocks per workgroup (13B) the short-K producer (o_proj, K = 5120) gains too: one workgroup per CU, 1/8 of x each.
            c.k8 = w8 && k8_ok && K >= lab_switch("PGV_GEMV_K8_NARROW_MINK", c.nwb == 10 ? 4096 : 12288);
            break;
        default: break;
    }
    return c;
}

int pgv_launch_gemv(pgv_ctx* ctx, int dtype, int mode, const void* W, const void* x, int ldx, void* out, int ldo, int N, int K, int B, hipStream_t s,
                    const float* wscale, const GemvNorm* norm) {
    PGV_CHECK(B >= 1 && B <= 64, "gemv: batch %d outside [1,64]", B);
    PGV_CHECK(K % 32 == 0, "gemv: K=%d must be a multiple of 32", K);
    PGV_CHECK(mode == GV_STORE16 || mode == GV_RESID || mode == GV_SWIGLU || mode == GV_F32 || mode == GV_RESIDNORM, "gemv: bad mode %d", mode);
    const bool w8 = wscale != nullptr;                  // W is the fp8 blocked copy (fp8.hip) with per-row scales
    GemvArgs a;
    a.W = (const char*)W; a.x = (const char*)x; a.out = (char*)out; a.N = N; a.K = K; a.B = B; a.ldx = ldx; a.ldo = ldo; a.wscale = wscale;
    a.ssq_in = nullptr; a.nparts_in = 0; a.inv_h = 0.f; a.eps = 0.f; a.gamma = nullptr; a.xg = nullptr; a.ssq_out = nullptr; a.amax_val = nullptr; a.amax_idx = nullptr;
    a.ssq_ts = 0; a.amax_ts = 0; a.lds_bytes = 0; a.xblk = 0; a.xgblk = 0;
#ifdef PGV_LAB
    a.abl = lab_switch("PGV_GEMV_ABLATE", 0);
    if (B > 16 && lab_switch("PGV_GEMV_XBLK", 0) != 0) a.xblk = 1;      // lab: the caller hands over x in the blocked layout (scripts/microbench.py gemvwide)
#endif
    PGV_CHECK(B <= 16 || norm == nullptr || (norm->ssq_ts > 0 && (norm->amax_val == nullptr || norm->amax_ts > 0)), "gemv: batches beyond 16 need the tile strides of the side arrays");
    if (norm) {
        a.ssq_ts = norm->ssq_ts; a.amax_ts = norm->amax_ts;
        a.ssq_in = norm->ssq_in; a.nparts_in = norm->nparts_in; a.inv_h = 1.0f / (float)norm->hidden; a.eps = norm->eps;
        a.gamma = norm->gamma; a.xg = (char*)norm->xg; a.ssq_out = norm->ssq_out; a.amax_val = norm->amax_val; a.amax_idx = norm->amax_idx;
        if (norm->x_blocked && pgv_gemv_xblk_tiles(B) > 0) {          // consumers read x blocked, producers write xg blocked
            if (mode == GV_RESIDNORM) { PGV_CHECK(N % 32 == 0, "gemv: a blocked xg needs N in multiples of 32"); a.xgblk = 1; }
            else a.xblk = 1;                                       // (K % 32 == 0 is checked above)
        }
        PGV_CHECK(norm->nparts_in * 4 <= 3 * 8 * 64, "gemv: %d sum-of-squares partials exceed what a consumer workgroup loads (hidden <= 6144)", norm->nparts_in);
    }
    PGV_CHECK(mode != GV_RESIDNORM || (a.gamma && a.xg && a.ssq_out), "gemv: the residual+norm producer needs gamma / xg / ssq_out");
    if (w8) PGV_CHECK(K % 64 == 0, "gemv fp8: K=%d must be a multiple of 64", K);
    int grid;
    // W must be in the fragment-blocked layout with its row count padded to a multiple of 16 (zero rows)
    if (mode == GV_SWIGLU) { PGV_CHECK(N % 64 == 0, "gemv swiglu: N=%d must be a multiple of 64", N); grid = N / 32; }
    else { if (mode != GV_F32) PGV_CHECK(N % 16 == 0, "gemv: N=%d must be a multiple of 16", N); grid = (N + 15) / 16; }
    const GemvChoice c = choose_gemv(mode, w8, B, grid, K, ctx->num_cu, norm && norm->k8_part);
    pgv_prof_begin(ctx, 3, s);
    if (c.k8) {
#ifdef PGV_LAB
        a.abl = lab_switch("PGV_K8_ABLATE", 0);     // lab: 1 = no x staging, 2 = no MFMA / fragment reads, 4 = no weight loads, 8 = no partial-tile stores, 16 = no finish launch (garbage results)
#endif
        PGV_TRY(kK8[w8][c.ct == 1 ? 0 : (c.ct == 2 ? 1 : 2)][c.nwb == 10](dtype, a, grid, norm->k8_part, s));
    } else {
        const GemvShape* shape = nullptr;
        for (const GemvShape& g : kGemvShapes)
            if (g.mode == mode && g.tl == c.tl && g.ct == c.ct) { shape = &g; break; }
        PGV_CHECK(shape != nullptr, "gemv: no kernel for mode %d with %d row blocks x %d column tiles", mode, c.tl, c.ct);
        const int tiles_per_wg = mode == GV_SWIGLU ? c.tl / 2 : c.tl;       // SwiGLU: the grid counts (gate, up) pairs
        gemv_launch_fn fn = shape->fn[w8][c.x2 && c.ct == 1];
        PGV_CHECK(fn != nullptr, "gemv: mode %d with %d row blocks x %d column tiles has no %s kernel", mode, c.tl, c.ct, w8 ? "fp8" : "16-bit");
        PGV_TRY(fn(dtype, (grid + tiles_per_wg - 1) / tiles_per_wg, a, s));     // (only the three-pair SwiGLU shape can leave a ragged last workgroup)
    }
    pgv_prof_end(ctx, 3, s, 2.0 * B * (double)N * K, (w8 ? 1.0 : 2.0) * (double)N * K);
    return PGV_OK;
}

extern "C" int pgv_gemv(pgv_ctx* ctx, int dtype, int mode, const void* d_W, const void* d_x, int ldx, void* d_out, int ldo, int N, int K, int B,
                        void* stream) {
    PGV_CHECK(ctx && d_W && d_x && d_out, "pgv_gemv: null argument");
    PGV_CHECK(mode >= 0 && mode <= 3, "pgv_gemv: mode %d outside [0,3]", mode);
    return pgv_launch_gemv(ctx, dtype, mode, d_W, d_x, ldx, d_out, ldo, N, K, B, (hipStream_t)stream, nullptr, nullptr);
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
    return pgv_launch_gemv(ctx, dtype, mode, d_W8, d_x, ldx, d_out, ldo, N, K, B, (hipStream_t)stream, d_scales, nullptr);
}
