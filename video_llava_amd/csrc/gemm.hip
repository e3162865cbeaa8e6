// MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue), 16-bit in, fp32 accumulate.
//
// This is the workhorse of the path: every Linear of the CLIP tower (HF:clip/modeling_clip.py
// CLIPAttention q/k/v/out_proj :290-293, CLIPMLP fc1/fc2 :343-344, patch conv :151-157), the
// mm_projector (video_chatgpt/model/video_chatgpt.py:51-55,105) and the LLaMA prefill projections
// (HF:llama/modeling_llama.py LlamaAttention/LlamaMLP) is `x @ W.T (+ b)` with x and W both
// K-contiguous, so one NT kernel serves all of them.
//
// Design (CDNA4), details at the kernel: persistent 256x256x64 tiles, four waves per workgroup (one per SIMD, 128x128 each,
// accumulators in the AGPR file), operands staged by LDS-DMA (buffer_load ... lds) into a five-slot LDS ring whose 16-B chunks are
// XOR-swizzled via the DMA *source* address and again on the ds_read_b128 address (conflict-free fragment reads), operands fed
// swapped (MFMA A-operand = W fragment) so a lane owns 4 consecutive output columns, epilogue transposed through LDS so every
// global access is a whole 128-byte line, XCD-aware tile order (tiles that share an A row-panel run on one XCD / L2).
// What was measured on the way here (scripts/lab/, profiles/): a 2-waves-per-SIMD ping-pong schedule with 8 barriers per K-step
// reached 1.25 PF on 8192^3, this one 1.4 PF on random data and 1.8-1.9 PF on constant data: the chip clocks to its power budget,
// so the random-data number is the honest one.
#include <stdlib.h>
#include <type_traits>

#include "pgv_common.h"

namespace {

struct KArgs {
    const char* A; const char* W; const float* bias; char* C;
    int lda, ldw, ldc;      // elements
    int M, N, K;            // logical sizes (N = output columns before SWIGLU halving)
    int ntm, ntn;
    int wblk;               // W is in the fragment-blocked layout (weights.h) instead of row-major
    // LayerNorm folded into the GEMMs around it (CLIP tower; see the EPI_LN_* epilogues)
    const float* rowstat;   // consumer: [M][2] = (mean, rstd) of the residual row
    const float* colsum;    // consumer: [N] s_n = sum_k gamma_k W[n,k]
    const float* gnext;     // producer: [N] gamma of the LayerNorm that follows
    char* x16; int ldx16;   // producer: 16-bit copy round16(resid * gnext), row stride ldx16 elements
    float* stats_part;      // producer: [N/64][M][2] partial (sum, sum of squares) of the CENTRED new residual row over each 64-column piece
    const float* rowmean;   // producer: [M] mean of each row before this sublayer (null: centre 0 -- the plain 16-bit copy of the last layer)
    const float* cshift;    // producer: device scalar added to rowmean to form the centre: mean(bias) of this GEMM, the data-independent part of
                            // how far the sublayer moves the row mean (null = 0)
};

// Internal epilogues of the CLIP tower (not part of the public pgv_epi enum).  LayerNorm has no launch of its own:
//   LN(x) W^T + b = rstd (x gamma) W^T - rstd mu (W gamma) + (b + W beta)
// The GEMM that COMPLETES a residual row block (out_proj, fc2; EPI_BIAS_RESID_LNOUT) writes, next to the fp32 residual, the 16-bit operand
// x16 = round16(resid * gamma_next) of the next GEMM and per-row partial sums (sum, sum of squares) over 64-column pieces; a small kernel
// turns the partials into (mean, rstd) per row; the consumer GEMM (qkv, fc1; EPI_LN_BIAS / EPI_LN_BIAS_QGELU) finishes in its epilogue:
//   out[m,n] = rstd_m (acc[m,n] - mean_m s_n) + b'_n,   s_n = sum_k gamma_k W[n,k],  b'_n = b_n + sum_k beta_k W[n,k]  (fp32, at load time).
// No weight is modified (gamma rides on the activation side), the statistics come from the fp32 residual; the 16-bit rounding moves from
// after the normalisation to before it (same relative size).  Saves the read of the fp32 residual + the launch of every LayerNorm.
// CENTRED form (round 3): the identity holds for x - c with any per-row c,
//   LN(x) W^T + b = rstd ((x - c) gamma) W^T - rstd (mu - c) (W gamma) + (b + W beta),
// and the producer uses c = the row's mean BEFORE its own update (rowmean, maintained by ln_stats_kernel) + mean(bias of this GEMM) (the
// data-independent part of the update; what remains, a_row . mean_n W[n,:], is small): x16 = round16((resid - c) gamma_next),
// partial sums of (resid - c); the consumer is unchanged with rowstat = (mu - c, rstd).  The operand that is rounded to 16 bits now has the
// magnitude of the row's SPREAD whatever its mean, so mean-dominated rows (|mu| >> sigma) lose nothing -- the uncentred form carried
// |mu| / sigma times the rounding error of the normalised operand and was only safe because random-init CLIP rows are zero-mean.
enum { EPI_LN_BIAS = 8, EPI_LN_BIAS_QGELU = 9, EPI_BIAS_RESID_LNOUT = 10 };

// Sum over each aligned group of 16 lanes, every lane receiving a BIT-IDENTICAL total: a butterfly of exchanges (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror) adds the same two partial sums on both sides of every exchange, and a + b == b + a.
// (A rotate-based reduction -- row_ror:4, row_ror:8 -- associates differently in different quads: (q0+q1)+(q2+q3) vs (q1+q2)+(q3+q0);
// with the lane that reports a row depending on the row's position, LayerNorm statistics became position dependent in the last bit.)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// x * sigmoid(a x) as v_mul, v_exp, v_add, v_rcp, v_mul.  The IEEE division `x / (1 + expf(..))` expands to ~10 more VALU per element
// (div_scale / fma chain / div_fixup): SQ_INSTS_VALU per MFMA was 5.2 in the fc1 GEMM against 1.8 with the plain bias epilogue.
// v_rcp_f32 and v_exp_f32 are 1-ulp fp32 approximations; the result is rounded to 16 bits right after.
__device__ __forceinline__ float sigmoid_mul_f(float x, float neg_a_log2e) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * neg_a_log2e)); }
__device__ __forceinline__ float quick_gelu_f(float x) { return sigmoid_mul_f(x, -1.702f * 1.4426950408889634f); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return sigmoid_mul_f(x, -1.4426950408889634f); }

// Buffer descriptor over `bytes` bytes at `base`, built from provably wave-uniform inputs (cdna_hip_programming.md T20):
// out-of-range lanes of a raw buffer load/store are dropped by the hardware, so edge tiles need no exec-masked branches
// (an exec-masked VMEM op inside the persistent loop makes hipcc drain vmcnt(0) at the loop header and kills the pipeline).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const uintptr_t b = (uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr unsigned OOB = 0x80000000u;     // voffset of a lane whose store/load must be dropped

// Tile order (band order; wide outputs use the W-resident order inside tile_coords_v): block b runs on XCD b%8, so each XCD gets a contiguous run of tile ids; inside that run tiles are walked
// in bands of GM tile-rows, column-major inside a band, so the ~32 workgroups an XCD runs concurrently form a GM x (32/GM)
// patch that shares GM A-panels and 32/GM W-panels through the XCD's L2 (instead of 1 A-panel and 32 W-panels).
// vb = virtual block id (blockIdx.x + i * gridDim.x for a persistent workgroup), nwg = total number of tiles.
__device__ __forceinline__ void tile_coords_v(int vb, int nwg, int ntm, int ntn, int& tm, int& tn) {
#ifndef PGV_LAB_BAND_ORDER
    // W-resident order for wide outputs (qkv: 12 tile columns, fc1: 16): the N tiles are cut into 4 column groups, XCD x owns group x % 4 for
    // the tile rows of parity x / 4 and walks them row-major (its 32 concurrent tiles = 8 rows x 4 columns).  Its W panels (ntn / 4 of them,
    // 1.5 - 2 MB at K = 1024) stay in its L2 for the whole launch and every A panel is streamed by 4 XCDs at the same moment, instead of every
    // XCD re-streaming ALL W panels once per band (fc1: 1.6 of its 2.7 GB of L2-miss traffic was W).  Same tiles, same values; vision bench
    // 126.6 -> 125.3 ms per 800 frames (A/B on one box, gpurun_out/r3m).  Narrow outputs (out_proj, fc2: 4 tile columns) keep the band
    // order below: there the roles are reversed (A would be streamed four times).
    if ((ntn & 3) == 0 && ntn >= 8 && (ntm & 1) == 0) {
        const int wn = ntn >> 2, xcd = vb & 7, loc = vb >> 3;
        const int r = loc / wn, c = loc - r * wn;
        tm = 2 * r + (xcd >> 2);
        tn = (xcd & 3) * wn + c;
        return;
    }
#endif
#ifdef PGV_LAB_GM8
    constexpr int GM = 8;
#else
    constexpr int GM = 4;
#endif
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, loc = vb >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int band = t / (GM * ntn);
    const int idx = t - band * GM * ntn;
    const int rows = min(GM, ntm - band * GM);
    tn = idx / rows;
    tm = band * GM + (idx - tn * rows);
}

// =================================================================================================
// "w4" kernel: persistent 256x256x64 tiles with FOUR waves per workgroup, one per SIMD, each owning a 128x128 sub-tile
// (4x4 v_mfma_f32_32x32x16 accumulators = 256 registers of the 512 a lone wave on a SIMD may use).
// Measured basis (scripts/lab/mfma_fill.hip): a lone wave issues MFMAs back to back at the pipe rate, ds_read_b128 in the
// gaps are free and one LDS-DMA per 4 MFMAs costs ~14 %, so no second wave / ping-pong barrier protocol is needed to
// overlap loads with MFMAs; what remains is ONE workgroup barrier per K-step.
// Per K-step a wave runs four groups of 16 independent MFMAs (one 16-deep k slice each) with the fragments of the next
// group being read meanwhile (two fragment register sets):
//   G0: read (s,1) | DMA 2nd half of step s+1      G1: read (s,2)      G2: read (s,3)
//   -- lgkmcnt(0), vmcnt(0), s_barrier: step s+1 has landed for everyone, step s is fully read --
//   G3: read (s+1,0) | DMA 1st half of step s+2    [last K-step of a tile: epilogue]
// 0.5 ds_read_b128 and 0.25 DMA instructions per MFMA.  LDS: 2 x 64 KiB operand buffers + 32 KiB epilogue staging.
// Operand DMA uses buffer_load ... lds with per-lane voffsets (row clamped inside the tile) and the k offset in soffset,
// so there is no per-instruction address arithmetic.
// =================================================================================================
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_w4(const KArgs& p, f32x16_t (&acc)[4][4], char* stg, int m0w, int n0w, int lane) {
    // This wave's 128x128 sub-tile starts at (m0w, n0w); stg = 8 KiB private LDS.  Every epilogue stages fp32 pieces
    // (32 rows x 64 columns, row stride 256 B, 16-B chunk index XORed with row & 15) and finishes on the READ side, where a
    // lane owns 4 consecutive columns of one row: bias, activation, residual add and the 16-bit pack happen there, so only
    // 8 bias registers are live and every global access covers whole 128-byte lines.  Edge tiles are branch-free: rows
    // beyond M fall outside the buffer descriptor, columns beyond N get an out-of-range voffset.
    const int l31 = lane & 31, hi = lane >> 5;
    const int rrow = lane >> 4, rc = lane & 15;                // read side: 4 rows x 16 chunks of 16 B per instruction
    constexpr bool OUT32 = (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID || EPI == PGV_EPI_F32 || EPI == EPI_BIAS_RESID_LNOUT);
    constexpr bool RMW = (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID || EPI == EPI_BIAS_RESID_LNOUT);
    constexpr bool SWIGLU = (EPI == PGV_EPI_SWIGLU);
    constexpr bool LN_IN = (EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_QGELU);      // consumer of the folded LayerNorm
    constexpr bool LN_OUT = (EPI == EPI_BIAS_RESID_LNOUT);                        // producer
    constexpr int ES = OUT32 ? 4 : 2;
    const int rows_valid = max(0, min(128, p.M - m0w));
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.C + (size_t)m0w * p.ldc * ES, (unsigned)rows_valid * (unsigned)p.ldc * ES);
    const unsigned rowpitch = (unsigned)p.ldc * ES;
    unsigned voff[2];
    f32x4_t bv[2];
    f32x4_t sv[2];                                             // LN_IN: column sums s_n;  LN_OUT: gamma of the next LayerNorm
    unsigned voffx[2];                                         // LN_OUT: offsets into the x16 copy
    __amdgpu_buffer_rsrc_t rs_aux, rs_aux2, rs_cm;             // LN_IN: rowstat;  LN_OUT: x16, the partial statistics and the row centres
    if constexpr (LN_IN) rs_aux = make_rsrc(p.rowstat + (size_t)m0w * 2, (unsigned)rows_valid * 8u);
    if constexpr (LN_OUT) {
        rs_aux = make_rsrc(p.x16 + (size_t)m0w * p.ldx16 * 2, (unsigned)rows_valid * (unsigned)p.ldx16 * 2u);
        // partial statistics, piece-major [N/64][M][2]: the descriptor spans all pieces from this sub-tile's first row; rows >= M of the last
        // tile row land in the next piece's first rows... so they are dropped by the row test below instead of the range check
        rs_aux2 = make_rsrc(p.stats_part + (size_t)m0w * 2, (unsigned)(((size_t)(p.N >> 6) - 1) * p.M * 8 + (size_t)rows_valid * 8));
        // centres: a null rowmean gives a zero-size descriptor, whose loads return 0 (c = 0 without a branch)
        rs_cm = make_rsrc(p.rowmean ? p.rowmean + m0w : (const float*)p.C, p.rowmean ? (unsigned)rows_valid * 4u : 0u);
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        if constexpr (LN_IN || LN_OUT) {
            const int col = n0w + jp * 64 + rc * 4;
            sv[jp] = *(const f32x4_t*)((LN_IN ? p.colsum : p.gnext) + min(col, p.N - 4));
            if constexpr (LN_OUT) voffx[jp] = (col < p.N) ? (unsigned)rrow * (unsigned)p.ldx16 * 2u + (unsigned)col * 2u : OOB;
        }
        if constexpr (SWIGLU) {
            // W rows interleaved per 64 as [32 gate | 32 up]: piece jp holds 32 gate + 32 up columns -> 32 output columns;
            // lanes 0..7 of a row group own output columns 4 (rc & 7) .. +3 (rc >= 8 lanes idle on the store)
            const int col = n0w / 2 + jp * 32 + (rc & 7) * 4;
            voff[jp] = (rc < 8 && col < p.N / 2) ? (unsigned)rrow * rowpitch + (unsigned)col * 2 : OOB;
            bv[jp] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        } else {
            const int col = n0w + jp * 64 + rc * 4;
            voff[jp] = (col < p.N) ? (unsigned)rrow * rowpitch + (unsigned)col * ES : OOB;
            bv[jp] = *(const f32x4_t*)(p.bias + min(col, p.N - 4));
        }
    }
#ifndef PGV_LAB_OUT16_AUX
#define PGV_LAB_OUT16_AUX 2             // cache-policy bits of the 16-bit output stores of the LayerNorm-consumer GEMMs (qkv, fc1): non-temporal -- outputs of
#endif                                  // 0.6 - 0.8 GB per lane that the next kernel streams once; A/B (gpurun_out/r4m): 125.26 / 125.52 -> 124.89 / 125.28 ms.
                                        // (The A-operand DMA with the same hint: 132.2 / 133.4 ms -- the panels ARE re-read by the other tile columns.)
#ifndef PGV_LAB_EPI_AUX
#define PGV_LAB_EPI_AUX 2               // cache-policy bits of the fp32-residual read-modify-write: 2 = non-temporal.  The residual (842 MB per 400-frame
#endif                                  // lane) is touched once per producer GEMM and is far larger than L2 + Infinity Cache: streaming it past the L2 keeps the
                                        // A / W panels of the K loop resident.  A/B on one box (gpurun_out/r4l): vision bench 121.92 / 122.06 -> 121.28 / 121.45 ms
    u32x2_t rst[8];                                            // LN_IN: (mean - centre, rstd) of the 8 rows
    float cm[8];                                               // LN_OUT: centre of the 8 rows
    float csh = 0.f;
    if constexpr (LN_OUT) { if (p.rowmean && p.cshift) csh = *p.cshift; }
    float keep1 = 0.f, keep2 = 0.f;                            // LN_OUT: (sum, sum of squares) of the row this lane reports
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) {
        const int i = piece >> 1, jp = piece & 1;
        if constexpr (LN_IN) {
            if (jp == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) rst[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_aux, (unsigned)(i * 32 + r * 4 + rrow) * 8u, 0, 0);
            }
        }
        if constexpr (LN_OUT) {
            if (jp == 0) {
#pragma unroll
                for (int r = 0; r < 8; ++r) cm[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_cm, (unsigned)(i * 32 + r * 4 + rrow) * 4u, 0, 0)) + csh;
            }
        }
        u32x4_t old[8];
        if constexpr (RMW) {
#pragma unroll
            for (int r = 0; r < 8; ++r) old[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[jp] + (i * 32 + r * 4) * rowpitch, 0, PGV_LAB_EPI_AUX);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][2 * jp + jj][g * 4 + e];
                *(f32x4_t*)(stg + l31 * 256 + (((8 * jj + 2 * g + hi) ^ (l31 & 15)) << 4)) = v;
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = r * 4 + rrow;
            const unsigned o = voff[jp] + (i * 32 + r * 4) * rowpitch;
            f32x4_t d = *(const f32x4_t*)(stg + row * 256 + ((rc ^ (row & 15)) << 4));
            if constexpr (SWIGLU) {
                const f32x4_t u = *(const f32x4_t*)(stg + row * 256 + (((rc ^ 8) ^ (row & 15)) << 4));     // partner chunk: up (for rc < 8)
                __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(silu_f(d[0]) * u[0], silu_f(d[1]) * u[1], silu_f(d[2]) * u[2], silu_f(d[3]) * u[3]), rsrc, o, 0, 0);
            } else {
                if constexpr (LN_IN) {
                    const f32x2_t ms = __builtin_bit_cast(f32x2_t, rst[r]);            // (mean, rstd)
                    const float a = ms[1], c = -ms[1] * ms[0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = a * d[e] + (c * sv[jp][e] + bv[jp][e]);       // rstd (acc - mean s_n) + b'_n
                    if constexpr (EPI == EPI_LN_BIAS_QGELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] = quick_gelu_f(d[e]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d[e] += bv[jp][e];
                        if constexpr (EPI == PGV_EPI_BIAS_QGELU) d[e] = quick_gelu_f(d[e]);
                        if constexpr (EPI == PGV_EPI_BIAS_GELU) d[e] = gelu_erf_f(d[e]);
                    }
                }
                if constexpr (RMW) {
                    const f32x4_t q = __builtin_bit_cast(f32x4_t, old[r]);
                    d[0] += q[0]; d[1] += q[1]; d[2] += q[2]; d[3] += q[3];
                }
                if constexpr (OUT32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, d), rsrc, o, 0, RMW ? PGV_LAB_EPI_AUX : 0);
                else __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(d[0], d[1], d[2], d[3]), rsrc, o, 0, LN_IN ? PGV_LAB_OUT16_AUX : 0);
                if constexpr (LN_OUT) {
                    d[0] -= cm[r]; d[1] -= cm[r]; d[2] -= cm[r]; d[3] -= cm[r];           // centred from here on (the fp32 residual above is not)
                    __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(d[0] * sv[jp][0], d[1] * sv[jp][1], d[2] * sv[jp][2], d[3] * sv[jp][3]), rs_aux,
                                                          voffx[jp] + (unsigned)(i * 32 + r * 4) * (unsigned)p.ldx16 * 2u, 0, 0);
                    // the 16 lanes of a DPP row hold the 64 columns of this piece of one residual row: fold them (every lane gets the total);
                    // lane rc == r keeps it, so after the 8 rows lane (rrow, rc < 8) owns row rc * 4 + rrow of the piece's 32
                    const float t1 = row16_sum((d[0] + d[1]) + (d[2] + d[3])), t2 = row16_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                    if (rc == r) { keep1 = t1; keep2 = t2; }
                }
            }
        }
        if constexpr (LN_OUT) {
            // one coalesced store per piece: 32 consecutive rows x 8 bytes of the piece-major partial array [N/64][M][2]
            const unsigned so = (rc < 8 && i * 32 + rc * 4 + rrow < rows_valid) ? (unsigned)(i * 32 + rc * 4 + rrow) * 8u : OOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, f32x2_t{keep1, keep2}), rs_aux2, so, (unsigned)((n0w >> 6) + jp) * (unsigned)p.M * 8u, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// -------------------------------------------------------------------------------------------------
// Direct epilogue of the CLIP tower's three GEMMs (LayerNorm consumers qkv / fc1, producers out_proj / fc2): no LDS transpose.
// For these instances the kernel feeds the MFMA unswapped (A fragment = activation rows), so a lane holds COLUMN lane & 31 of each 32-column
// block and 16 rows (8 g + 4 hi + e) per accumulator tuple, and the W tile is DMAed with its rows permuted (LDS row 32 j + c of a wave's
// 128-column slab <- W row 4 c + j; the permutation lives in the DMA source offsets and costs nothing): block j of lane c is output
// column 4 c + j, so for one row the four tuples acc[i][0..3] give a lane 4 CONSECUTIVE columns and a half-wave 128 consecutive columns --
// every store is 16 B (fp32) or 8 B (16-bit) per lane over whole 128-byte lines, straight from the accumulators.  Per-row values (LayerNorm
// statistics, centres) are wave-uniform per (register, hi): loaded one K-step ahead (epi_direct_prefetch), parked in 1 KiB of the free
// ring slot and read back as broadcast ds_read_b128.  Same arithmetic, element for element, as the staged epilogue above (the 16 lanes of a
// DPP row still hold the 64 columns of a statistics piece in the same order): results are bitwise those of the staged form.
// -------------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epi_direct_prefetch(const KArgs& p, int m0w, int lane, u32x2_t (&pf)[2]) {
    const int rows_valid = max(0, min(128, p.M - m0w));
    if constexpr (EPI == EPI_BIAS_RESID_LNOUT) {
        // centres of rows lane and lane + 64 (a null rowmean gives a zero-size descriptor: loads return 0)
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.rowmean ? p.rowmean + m0w : (const float*)p.C, p.rowmean ? (unsigned)rows_valid * 4u : 0u);
        pf[0][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)lane * 4u, 0, 0);
        pf[1][0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(lane + 64) * 4u, 0, 0);
        pf[0][1] = 0; pf[1][1] = 0;
    } else {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.rowstat + (size_t)m0w * 2, (unsigned)rows_valid * 8u);
        pf[0] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)lane * 8u, 0, 0);
        pf[1] = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(lane + 64) * 8u, 0, 0);
    }
}

template <typename T, int EPI>
__device__ __forceinline__ void epilogue_direct(const KArgs& p, f32x16_t (&acc)[4][4], char* stg, int m0w, int n0w, int lane, const u32x2_t (&pf)[2]) {
    constexpr bool LN_OUT = (EPI == EPI_BIAS_RESID_LNOUT);
    constexpr int ES = LN_OUT ? 4 : 2;
    const int l31 = lane & 31, hi = lane >> 5;
    if constexpr (LN_OUT) { *(unsigned*)(stg + lane * 4) = pf[0][0]; *(unsigned*)(stg + 256 + lane * 4) = pf[1][0]; }
    else { *(u32x2_t*)(stg + lane * 8) = pf[0]; *(u32x2_t*)(stg + 512 + lane * 8) = pf[1]; }
    const int rows_valid = max(0, min(128, p.M - m0w));
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.C + (size_t)m0w * p.ldc * ES, (unsigned)rows_valid * (unsigned)p.ldc * ES);
    const unsigned rowpitch = (unsigned)p.ldc * ES;
    const int col = n0w + 4 * l31;
    const bool colok = col < p.N;
    const unsigned voff = colok ? (unsigned)(4 * hi) * rowpitch + (unsigned)col * ES : OOB;
    const f32x4_t bv = *(const f32x4_t*)(p.bias + min(col, p.N - 4));
    const f32x4_t sv = *(const f32x4_t*)((LN_OUT ? p.gnext : p.colsum) + min(col, p.N - 4));      // LN_IN: column sums s_n;  LN_OUT: gamma of the next LayerNorm
    __amdgpu_buffer_rsrc_t rs_x, rs_part;
    unsigned voffx = 0, xpitch = 0;
    float csh = 0.f;
    if constexpr (LN_OUT) {
        xpitch = (unsigned)p.ldx16 * 2u;
        rs_x = make_rsrc(p.x16 + (size_t)m0w * p.ldx16 * 2, (unsigned)rows_valid * xpitch);
        rs_part = make_rsrc(p.stats_part + (size_t)m0w * 2, (unsigned)(((size_t)(p.N >> 6) - 1) * p.M * 8 + (size_t)rows_valid * 8));
        voffx = colok ? (unsigned)(4 * hi) * xpitch + (unsigned)col * 2u : OOB;
        if (p.rowmean && p.cshift) csh = *p.cshift;
    }
    __builtin_amdgcn_wave_barrier();
    const char* srd = stg + hi * (LN_OUT ? 16 : 32);
    u32x4_t old[8];                                            // LN_OUT: rolling window of fp32 residual rows, 8 rows ahead
    if constexpr (LN_OUT) {
#pragma unroll
        for (int r = 0; r < 8; ++r) old[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)((r >> 2) * 8 + (r & 3)) * rowpitch, 0, PGV_LAB_EPI_AUX);
    }
    float keep1 = 0.f, keep2 = 0.f;
#pragma unroll
    for (int ig = 0; ig < 16; ++ig) {
        const int i = ig >> 2, g = ig & 3;
        const int rb = i * 32 + g * 8;                         // rows rb + 4 hi + e
        __builtin_amdgcn_sched_barrier(0);                     // hipcc otherwise hoists the accumulator reads of two row blocks (132 registers) to the top
        f32x4_t st01, st23, cmv;
        if constexpr (LN_OUT) {
            cmv = *(const f32x4_t*)(srd + rb * 4);
        } else {
            st01 = *(const f32x4_t*)(srd + rb * 8);            // (mean - centre, rstd) of rows e = 0, 1
            st23 = *(const f32x4_t*)(srd + rb * 8 + 16);       // ... e = 2, 3
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned ro = (unsigned)(rb + e);
            f32x4_t d;
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(d[j]) : "a"(acc[i][j][g * 4 + e]));
            if constexpr (!LN_OUT) {
                const float mean = (e < 2 ? st01 : st23)[(e & 1) * 2], rstd = (e < 2 ? st01 : st23)[(e & 1) * 2 + 1];
                const float a = rstd, c = -rstd * mean;
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = a * d[j] + (c * sv[j] + bv[j]);               // rstd (acc - mean s_n) + b'_n
                if constexpr (EPI == EPI_LN_BIAS_QGELU) {
                    // quick_gelu_f on pairs: the scale and the + 1 as v_pk_mul_f32 / v_pk_add_f32 (same IEEE operations, half the instructions)
                    const f32x2_t k2 = {-1.702f * 1.4426950408889634f, -1.702f * 1.4426950408889634f}, one2 = {1.0f, 1.0f};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2_t y = {d[2 * h], d[2 * h + 1]};
                        const f32x2_t z = y * k2;
                        const f32x2_t t = f32x2_t{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])} + one2;
                        const f32x2_t o = y * f32x2_t{__builtin_amdgcn_rcpf(t[0]), __builtin_amdgcn_rcpf(t[1])};
                        d[2 * h] = o[0]; d[2 * h + 1] = o[1];
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(d[0], d[1], d[2], d[3]), rsrc, voff + ro * rowpitch, 0, PGV_LAB_OUT16_AUX);
            } else {
                const int slot = (g & 1) * 4 + e;
                const f32x4_t q = __builtin_bit_cast(f32x4_t, old[slot]);
                if (ig + 2 < 16) {
                    const int ig2 = ig + 2;
                    old[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (unsigned)((ig2 >> 2) * 32 + (ig2 & 3) * 8 + e) * rowpitch, 0, PGV_LAB_EPI_AUX);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { d[j] += bv[j]; d[j] += q[j]; }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, d), rsrc, voff + ro * rowpitch, 0, PGV_LAB_EPI_AUX);
                const float cm = cmv[e] + csh;
                d[0] -= cm; d[1] -= cm; d[2] -= cm; d[3] -= cm;                                   // centred from here on (the fp32 residual above is not)
                __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(d[0] * sv[0], d[1] * sv[1], d[2] * sv[2], d[3] * sv[3]), rs_x, voffx + ro * xpitch, 0, 0);
                // the 16 lanes of a DPP row hold the 64 columns of one statistics piece of this row: fold them (every lane gets the total);
                // lane (l31 & 15) == 4 g + e keeps it, so after the 16 rows of block i lane (hi, piece, k) owns row i 32 + 8 (k >> 2) + 4 hi + (k & 3)
                const float t1 = row16_sum((d[0] + d[1]) + (d[2] + d[3])), t2 = row16_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]));
                if ((l31 & 15) == g * 4 + e) { keep1 = t1; keep2 = t2; }
            }
        }
        if constexpr (LN_OUT) {
            if (g == 3) {
                const int k = l31 & 15, row = i * 32 + 8 * (k >> 2) + 4 * hi + (k & 3);
                const unsigned so = (row < rows_valid) ? (unsigned)row * 8u : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, f32x2_t{keep1, keep2}), rs_part, so, (unsigned)((n0w >> 6) + (l31 >> 4)) * (unsigned)p.M * 8u, 0);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS = ring of five 32 KiB slots; half-step h (2s = the A tile of K-step s, 2s+1 = its W tile) lives in slot h % 5.
    // While step s is computed, step s+1 is resident and A(s+2) is landing in the fifth slot, so only HALF of a K-step's bytes
    // (W(s+2), issued right after the barrier that frees A(s)'s slot) has the one-K-step deadline; the other half has two.
    constexpr int BM = 256, BN = 256, SLOT = 32768;
#ifdef PGV_LAB_STAGED_EPI
    constexpr bool DIRECT = false;
#else
    constexpr bool DIRECT = (EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_QGELU || EPI == EPI_BIAS_RESID_LNOUT);     // epilogue_direct: unswapped MFMA, permuted W rows
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int total = p.ntm * p.ntn, G = gridDim.x;
    const int nk = p.K >> 6;

    // ---- DMA cursors (one for A tiles, one for W tiles): instruction j of wave w fills tile rows (j*4 + w)*8 .. +8 ----
    const int srow = lane >> 3, slot = lane & 7;
    const int chunk = slot ^ ((4 * w + (srow >> 1)) & 7);              // (row >> 1) & 7 for row = 32 j + 8 w + srow
    unsigned voffA[8], voffW[8];
    __amdgpu_buffer_rsrc_t rsA, rsW;
    int a_kt = 0, a_vb = blockIdx.x, a_slot = 0, w_kt = 0, w_vb = blockIdx.x, w_slot = 1;
    unsigned a_k = 0, w_k = 0;                                          // byte offset of the cursor's K-step inside a row
    const unsigned wkstep = p.wblk ? 2048u : 128u;
    auto rebaseA = [&](int vb) __attribute__((always_inline)) {
        if (vb >= total) { rsA = make_rsrc(p.A, 0u); return; }          // past the last tile: zero-size descriptor, the DMA writes zeros nobody reads
        int tm, tn;
        tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);
        const int m0 = tm * BM;
        const int rows = min(BM, p.M - m0);
        rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)rows * (unsigned)p.lda * 2u);
#pragma unroll
        for (int j = 0; j < 8; ++j) voffA[j] = (unsigned)min(32 * j + 8 * w + srow, rows - 1) * (unsigned)p.lda * 2u + chunk * 16;
    };
    auto rebaseW = [&](int vb) __attribute__((always_inline)) {
        if (vb >= total) { rsW = make_rsrc(p.W, 0u); return; }
        int tm, tn;
        tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);
        const int n0 = tn * BN;
        const int rows = min(BN, p.N - n0);
        if (p.wblk) {
            const size_t bytes = (size_t)((p.N + 15) & ~15) * p.K * 2;
            rsW = make_rsrc(p.W, bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes);
        } else {
            rsW = make_rsrc(p.W + (size_t)n0 * p.ldw * 2, (unsigned)rows * (unsigned)p.ldw * 2u);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int lrow = 32 * j + 8 * w + srow;                            // LDS row of the tile = 128 slab + 32 jb + c
            if constexpr (DIRECT) lrow = 4 * (8 * w + srow) + (j >> 2) * 128 + (j & 3);           // ... holds W row 128 slab + 4 c + jb (epilogue_direct)
            const int row = min(lrow, rows - 1);
            if (p.wblk) {
                const int rw = n0 + row;                                 // 1 KiB block (rw/16, k/32); inside: slot ((k%32)/8)*16 + rw%16
                voffW[j] = (unsigned)((((size_t)(rw >> 4) * (p.K >> 5) + (chunk >> 2)) * 512 + (((chunk & 3) << 4) + (rw & 15)) * 8) * 2);
            } else {
                voffW[j] = (unsigned)row * (unsigned)p.ldw * 2u + chunk * 16;
            }
        }
    };
    auto dma_one = [&](auto which_c, int j) __attribute__((always_inline)) {      // which: 0 = A cursor, 1 = W cursor
        constexpr int which = decltype(which_c)::value;
        if constexpr (!(ABL & 1)) {
            char* dst = smem + (which == 0 ? a_slot : w_slot) * SLOT + (j * 4 + w) * 1024;
#ifndef PGV_LAB_DMA_AUX
#define PGV_LAB_DMA_AUX 0               // cache-policy bits of the operand DMA (lab A/B: 1 = sc0, 2 = sc1, 8 = nt ...)
#endif
#ifndef PGV_LAB_DMA_A_LNIN_AUX
#define PGV_LAB_DMA_A_LNIN_AUX 0            // ... of the A operand of the LayerNorm-consumer GEMMs (qkv, fc1: W-resident order, every A panel read once per XCD)
#endif
#ifndef PGV_LAB_DMA_A_AUX
#define PGV_LAB_DMA_A_AUX PGV_LAB_DMA_AUX   // ... of the A operand alone
#endif
            __builtin_amdgcn_raw_ptr_buffer_load_lds(which == 0 ? rsA : rsW, (__attribute__((address_space(3))) void*)dst, 16,
                                                     which == 0 ? voffA[j] : voffW[j], which == 0 ? a_k : w_k, 0,
                                                     which == 0 ? ((EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_QGELU) ? PGV_LAB_DMA_A_LNIN_AUX : PGV_LAB_DMA_A_AUX) : PGV_LAB_DMA_AUX);
        }
    };
    auto advance = [&](auto which_c) __attribute__((always_inline)) {             // cursor -> the same tile kind of the next K-step
        constexpr int which = decltype(which_c)::value;
        if constexpr (which == 0) {
            a_slot += 2; if (a_slot >= 5) a_slot -= 5;
            a_k += 128;
            if (__builtin_expect(++a_kt == nk, 0)) { a_kt = 0; a_k = 0; a_vb += G; rebaseA(a_vb); }
        } else {
            w_slot += 2; if (w_slot >= 5) w_slot -= 5;
            w_k += wkstep;
            if (__builtin_expect(++w_kt == nk, 0)) { w_kt = 0; w_k = 0; w_vb += G; rebaseW(w_vb); }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using NoDma [[maybe_unused]] = std::integral_constant<int, -1>;
    using First = std::true_type;
    using Later = std::false_type;
    rebaseA(blockIdx.x); rebaseW(blockIdx.x);

    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const int a_off = (wr * 128 + l31) * 128;
    const int w_off = (wc * 128 + l31) * 128;

    f32x16_t acc[4][4];                                                   // written first by the C = 0 MFMAs of each tile
    typename T::v8 fa[2][4], fw[2][4];
    // The accumulators live in the AGPR half of the register file ("a" constraints): hipcc's own allocation of 256 accumulator
    // registers shuffles them between the files and spills, so the MFMA is issued from an asm statement.  volatile keeps the
    // hand-written MFMA / ds_read / DMA interleave below in source order.
    auto mfma = [&](auto first_c, f32x16_t& c, const typename T::v8& wfrag, const typename T::v8& afrag) __attribute__((always_inline)) {
        constexpr bool first = decltype(first_c)::value;      // first k slice of an output tile: C = 0 (no accumulator zeroing pass)
        if constexpr (ABL & 4) { if constexpr (first) asm volatile("" : "=a"(c) : "v"(wfrag), "v"(afrag)); else asm volatile("" : "+a"(c) : "v"(wfrag), "v"(afrag)); }
        else if constexpr (T::id == PGV_BF16) {
            if constexpr (first) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(wfrag), "v"(afrag));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(wfrag), "v"(afrag));
        } else {
            if constexpr (first) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(wfrag), "v"(afrag));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(wfrag), "v"(afrag));
        }
    };
    // One group: 16 MFMAs on fragment set SET; after every second MFMA one ds_read_b128 of the NEXT fragment set (k slice rkk of the
    // A tile at ra / the W tile at rw); after every fourth MFMA one LDS-DMA instruction (DMA = 0/1: first/second half of the A
    // cursor's tile, 2/3: of the W cursor's tile, -1: none).  16 DMA instructions per wave per K-step = one per 4 MFMAs, spread
    // evenly: the texture path takes ~30 cycles per instruction per CU, bunching them stalls the (in-order) MFMA issue behind them.
    auto group = [&](auto first_c, auto set_c, const char* ra, const char* rw, int rkk, auto dma_c) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value, dmasel = decltype(dma_c)::value;
        using Which = std::integral_constant<int, (dmasel >> 1)>;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if constexpr (DIRECT) {
                mfma(first_c, acc[(2 * t) & 3][(2 * t) >> 2], fa[set][(2 * t) & 3], fw[set][(2 * t) >> 2]);
                mfma(first_c, acc[(2 * t + 1) & 3][(2 * t + 1) >> 2], fa[set][(2 * t + 1) & 3], fw[set][(2 * t + 1) >> 2]);
            } else {
                mfma(first_c, acc[(2 * t) & 3][(2 * t) >> 2], fw[set][(2 * t) >> 2], fa[set][(2 * t) & 3]);
                mfma(first_c, acc[(2 * t + 1) & 3][(2 * t + 1) >> 2], fw[set][(2 * t + 1) >> 2], fa[set][(2 * t + 1) & 3]);
            }
            if constexpr (!(ABL & 2)) {
                if (t < 4) fa[set ^ 1][t] = *(const typename T::v8*)(ra + a_off + t * 4096 + koffs[rkk]);
                else fw[set ^ 1][t - 4] = *(const typename T::v8*)(rw + w_off + (t - 4) * 4096 + koffs[rkk]);
            }
            if constexpr (dmasel >= 0) { if (t & 1) dma_one(Which{}, (dmasel & 1) * 4 + (t >> 1)); }
        }
        if constexpr (dmasel >= 0 && (dmasel & 1)) advance(Which{});
    };
    using DmaA0 = std::integral_constant<int, 0>;
    using DmaA1 = std::integral_constant<int, 1>;
    using DmaW0 = std::integral_constant<int, 2>;
    using DmaW1 = std::integral_constant<int, 3>;

    // prologue: A0 W0 A1 and the first half of W1 (slots 0 1 2 3); the loop continues with W1's second half
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_one(S0{}, j);
    advance(S0{});
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_one(S1{}, j);
    advance(S1{});
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_one(S0{}, j);
    advance(S0{});
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_one(S1{}, j);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // K-step 0 landed (this wave's share)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = *(const typename T::v8*)(smem + a_off + i * 4096 + koffs[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fw[0][j] = *(const typename T::v8*)(smem + SLOT + w_off + j * 4096 + koffs[0]);

    // The DMA cursors need no end-of-work guards: past the last K-step they run on zero-size descriptors into free slots.
    int ca = 0, cw = 1;                                                   // slots of the K-step being computed
#define PGV_W4_KSTEP(FIRST)                                                                                                \
    {                                                                                                                       \
        const char* ra = smem + ca * SLOT;                                                                                  \
        const char* rw = smem + cw * SLOT;                                                                                  \
        int na = ca + 2; if (na >= 5) na -= 5;                                                                              \
        int nw = cw + 2; if (nw >= 5) nw -= 5;                                                                              \
        group(FIRST{}, S0{}, ra, rw, 1, DmaW1{});                    /* + 2nd half of W(s+1) (slot of A(s-1))              */ \
        group(Later{}, S1{}, ra, rw, 2, DmaA0{});                    /* + A(s+2) into the slot W(s-1) left at the last barrier */ \
        group(Later{}, S0{}, ra, rw, 3, DmaA1{});                                                                           \
        if constexpr (ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); /* step s+1 landed; only A(s+2) may still fly */   \
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();     /* ... for everyone, and step s is fully read */       \
        group(Later{}, S1{}, smem + na * SLOT, smem + nw * SLOT, 0, DmaW0{}); /* + 1st half of W(s+2) into the slot of A(s) */ \
        stg_slot = cw; ca = na; cw = nw;                                                                                    \
    }
    int stg_slot = 1;
    for (int vb = blockIdx.x; vb < total; vb += G) {
        int tm, tn;
        tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);
        if constexpr (DIRECT) {
            // the per-row values of the epilogue are fetched one K-step ahead (K >= 128 is checked at launch)
            u32x2_t pf[2];
            PGV_W4_KSTEP(First)
            for (int kt = 1; kt < nk - 1; ++kt) PGV_W4_KSTEP(Later)
            epi_direct_prefetch<EPI>(p, tm * BM + wr * 128, lane, pf);
            PGV_W4_KSTEP(Later)
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");         // last MFMA result -> first accumulator read
            epilogue_direct<T, EPI>(p, acc, smem + stg_slot * SLOT + w * 8192, tm * BM + wr * 128, tn * BN + wc * 128, lane, pf);
        } else {
            PGV_W4_KSTEP(First)                                       // C = 0 form on the first k slice: no zeroing pass
            for (int kt = 1; kt < nk; ++kt) PGV_W4_KSTEP(Later)
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");         // last MFMA result -> first accumulator read
            // staging = the slot of the W tile just consumed: free until the next K-step's G0 refills it
            if constexpr (!(ABL & 32)) epilogue_w4<T, EPI>(p, acc, smem + stg_slot * SLOT + w * 8192, tm * BM + wr * 128, tn * BN + wc * 128, lane);
        }
        __builtin_amdgcn_s_barrier();                                 // the next G0 refills the staging slot: every wave must be done with it
    }
#undef PGV_W4_KSTEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the cursors' trailing DMAs must land before the LDS is released
}

template <typename T, int EPI, int ABL>
int launch_w4_inst(KArgs k, hipStream_t s, int num_cu) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_w4<T, EPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) { pgv_set_error("hipFuncSetAttribute(gemm_w4): %s", hipGetErrorString(e)); return PGV_EHIP; }
        configured = true;
    }
    k.ntm = (k.M + 255) / 256; k.ntn = (k.N + 255) / 256;
    const int total = k.ntm * k.ntn;
    const int grid = total < num_cu ? total : num_cu;
    hipLaunchKernelGGL((gemm_w4<T, EPI, ABL>), dim3(grid), dim3(256), 163840, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pgv_set_error("gemm_w4 launch: %s", hipGetErrorString(e)); return PGV_EHIP; }
    return PGV_OK;
}

template <typename T, int EPI>
int launch_w4(const KArgs& k, hipStream_t s, int num_cu) {
#ifdef PGV_LAB
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("PGV_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    // Lab builds only (-DPGV_LAB): timing ablations (results are garbage), plain BIAS epilogue in bf16 only -- PGV_GEMM_ABLATE bits: 1 no DMA,
    // 2 no fragment reads, 4 no MFMA, 8 no counted vmcnt wait, 16 no barrier, 32 no epilogue (scripts/microbench.py ablate,
    // scripts/lab/gemm_epi_decomp.py).
    // The release library has no garbage-producing ablation switch (its documented A/B switches, INTEGRATION.md, all pass the parity tests).
    if constexpr (EPI == PGV_EPI_BIAS && T::id == PGV_BF16) {
        switch (abl) {
            case 1: return launch_w4_inst<T, EPI, 1>(k, s, num_cu);
            case 3: return launch_w4_inst<T, EPI, 3>(k, s, num_cu);
            case 6: return launch_w4_inst<T, EPI, 6>(k, s, num_cu);
            case 8: return launch_w4_inst<T, EPI, 8>(k, s, num_cu);
            case 12: return launch_w4_inst<T, EPI, 12>(k, s, num_cu);
            case 14: return launch_w4_inst<T, EPI, 14>(k, s, num_cu);
            case 24: return launch_w4_inst<T, EPI, 24>(k, s, num_cu);
            case 32: return launch_w4_inst<T, EPI, 32>(k, s, num_cu);
            default: break;
        }
    }
#endif
    return launch_w4_inst<T, EPI, 0>(k, s, num_cu);
}

template <typename T>
int dispatch_epi(int epi, const KArgs& k, hipStream_t s, int num_cu) {
    switch (epi) {
        case PGV_EPI_NONE:       // bias pointer is the context's zero vector -> same code path as BIAS
        case PGV_EPI_BIAS:       return launch_w4<T, PGV_EPI_BIAS>(k, s, num_cu);
        case PGV_EPI_BIAS_QGELU: return launch_w4<T, PGV_EPI_BIAS_QGELU>(k, s, num_cu);
        case PGV_EPI_BIAS_GELU:  return launch_w4<T, PGV_EPI_BIAS_GELU>(k, s, num_cu);
        case PGV_EPI_RESID:
        case PGV_EPI_BIAS_RESID: return launch_w4<T, PGV_EPI_BIAS_RESID>(k, s, num_cu);
        case PGV_EPI_SWIGLU:     return launch_w4<T, PGV_EPI_SWIGLU>(k, s, num_cu);
        case PGV_EPI_F32:        return launch_w4<T, PGV_EPI_F32>(k, s, num_cu);
        case EPI_LN_BIAS:        return launch_w4<T, EPI_LN_BIAS>(k, s, num_cu);
        case EPI_LN_BIAS_QGELU:  return launch_w4<T, EPI_LN_BIAS_QGELU>(k, s, num_cu);
        case EPI_BIAS_RESID_LNOUT: return launch_w4<T, EPI_BIAS_RESID_LNOUT>(k, s, num_cu);
    }
    pgv_set_error("unknown epilogue %d", epi);
    return PGV_EINVAL;
}

}  // namespace

int pgv_launch_gemm(pgv_ctx* ctx, int dtype, const GemmArgs& a, hipStream_t s) {
    PGV_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    PGV_CHECK(a.K % 64 == 0, "gemm: K=%d must be a multiple of 64", a.K);
    PGV_CHECK(a.N % 8 == 0, "gemm: N=%d must be a multiple of 8", a.N);
    PGV_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm: lda/ldw must be multiples of 8 elements");
    PGV_CHECK(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: A/W must be 16-byte aligned");
    const bool out32 = (a.epi == PGV_EPI_RESID || a.epi == PGV_EPI_BIAS_RESID || a.epi == PGV_EPI_F32 || a.epi == EPI_BIAS_RESID_LNOUT);
    if (a.epi == EPI_LN_BIAS || a.epi == EPI_LN_BIAS_QGELU || a.epi == EPI_BIAS_RESID_LNOUT) PGV_CHECK(!a.w_blocked && a.K >= 128, "gemm: the LayerNorm epilogues take row-major weights and K >= 128");
    if (a.epi == EPI_LN_BIAS || a.epi == EPI_LN_BIAS_QGELU) PGV_CHECK(a.rowstat && a.colsum && a.bias, "gemm: the LayerNorm-consumer epilogue needs rowstat / colsum / bias");
    if (a.epi == EPI_BIAS_RESID_LNOUT) PGV_CHECK(a.gnext && a.x16 && a.stats_part && a.N % 64 == 0 && a.ldx16 % 4 == 0, "gemm: the LayerNorm-producer epilogue needs gnext / x16 / stats_part and N %% 64 == 0");
    PGV_CHECK(((uintptr_t)a.C & (out32 ? 15 : 7)) == 0 && a.ldc % 4 == 0, "gemm: C misaligned (ptr/ldc)");
    PGV_CHECK((size_t)256 * a.lda * 2 < 0xffffffffull && (size_t)256 * a.ldw * 2 < 0xffffffffull && (size_t)128 * a.ldc * 4 < 0x7fffffffull,
              "gemm: leading dimension too large for the 32-bit buffer offsets");
    KArgs k;
    k.A = (const char*)a.A; k.W = (const char*)a.W; k.bias = a.bias; k.C = (char*)a.C;
    k.lda = a.lda; k.ldw = a.ldw; k.ldc = a.ldc; k.M = a.M; k.N = a.N; k.K = a.K;
    k.ntm = 0; k.ntn = 0;
    k.wblk = a.w_blocked ? 1 : 0;
    k.rowstat = a.rowstat; k.colsum = a.colsum; k.gnext = a.gnext; k.x16 = (char*)a.x16; k.ldx16 = a.ldx16; k.stats_part = a.stats_part; k.rowmean = a.rowmean; k.cshift = a.cshift;
    PGV_CHECK(!a.w_blocked || a.N % 16 == 0, "gemm: blocked weights need N %% 16 == 0");
    if (k.bias == nullptr) {        // the branch-free epilogue always reads a bias vector
        PGV_CHECK(a.N <= PGV_ZERO_BIAS_LEN, "gemm: N=%d exceeds the zero-bias vector", a.N);
        k.bias = ctx->zero_bias;
    }
    pgv_prof_begin(ctx, 0, s);
    int rc;
    if (dtype == PGV_F16) rc = dispatch_epi<TF16>(a.epi, k, s, ctx->num_cu);
    else if (dtype == PGV_BF16) rc = dispatch_epi<TBF16>(a.epi, k, s, ctx->num_cu);
    else { pgv_set_error("gemm: unsupported dtype %d", dtype); rc = PGV_EINVAL; }
    const double out_cols = (a.epi == PGV_EPI_SWIGLU) ? a.N / 2.0 : (double)a.N;
    pgv_prof_end(ctx, 0, s, 2.0 * a.M * (double)a.N * a.K,
                 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + (out32 ? 8.0 : 2.0) * a.M * out_cols);
    return rc;
}

extern "C" int pgv_gemm(pgv_ctx* ctx, int dtype, int epi, const void* d_A, int lda, const void* d_W, int ldw, const float* d_bias,
                        void* d_C, int ldc, int M, int N, int K, void* stream) {
    PGV_CHECK(ctx != nullptr, "pgv_gemm: null ctx");
    GemmArgs a{};
    a.A = d_A; a.lda = lda; a.W = d_W; a.ldw = ldw; a.bias = d_bias; a.C = d_C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.epi = epi;
    return pgv_launch_gemm(ctx, dtype, a, (hipStream_t)stream);
}
