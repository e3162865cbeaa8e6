// MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue), 16-bit in, fp32 accumulate.
//
// This is the workhorse of the path: every Linear of the CLIP tower (HF:clip/modeling_clip.py
// CLIPAttention q/k/v/out_proj :290-293, CLIPMLP fc1/fc2 :343-344, patch conv :151-157), the
// mm_projector (video_chatgpt/model/video_chatgpt.py:51-55,105) and the LLaMA prefill projections
// (HF:llama/modeling_llama.py LlamaAttention/LlamaMLP) is `x @ W.T (+ b)` with x and W both
// K-contiguous, so one NT kernel serves all of them.
//
// Design (CDNA4): 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), each wave a
// 128x64 sub-tile = 4x2 v_mfma_f32_32x32x16 accumulators), BK=64, one workgroup per CU.
// Global->LDS staging uses global_load_lds_dwordx4 (no VGPR round trip).  The LDS image of a tile is
// [rows][64] 16-bit = 128 B rows; a DMA'd wave-instruction fills 8 rows linearly, so the bank-conflict
// swizzle is applied on the SOURCE chunk index (chunk ^= (row>>1)&7) and again on the ds_read_b128
// address: the 16 lanes a ds_read_b128 services together then hit 16 distinct 16-B slots.
// Operands are fed swapped (MFMA A-operand = W fragment, B-operand = A fragment) so each lane ends up
// with 4 consecutive output columns of one row -> 8-byte (16-bit out) / 16-byte (fp32 residual) stores.
// Workgroups are renumbered so that the tiles an XCD works on concurrently share A row-panels (per-XCD L2).
#include <stdlib.h>

#include "pgv_common.h"

namespace {

struct KArgs {
    const char* A; const char* W; const float* bias; char* C;
    int lda, ldw, ldc;      // elements
    int M, N, K;            // logical sizes (N = output columns before SWIGLU halving)
    int ntm, ntn;
    int wblk;               // W is in the fragment-blocked layout (weights.h) instead of row-major
};

__device__ __forceinline__ float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }


// ---- shared epilogue: lane holds, for row m = mbase + 32 i, columns nbase + 32 j + 8 g + {0..3} (i < 4, j < 2, g < 4) ----
template <typename T, int EPI, bool FULL>
__device__ __forceinline__ void epilogue_store(const KArgs& p, const f32x16_t (&acc)[4][2], int mbase, int nbase, int obase) {
    if constexpr (EPI == PGV_EPI_SWIGLU) {
        // W rows interleaved per 64: [32 gate | 32 up]; output column = obase + 8g + e
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + i * 32;
            if (FULL || m < p.M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(acc[i][0][g * 4 + e]) * acc[i][1][g * 4 + e];
                    const int n = obase + 8 * g;
                    if (FULL || n < p.N / 2) *(u32x2_t*)(p.C + ((size_t)m * p.ldc + n) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            }
        }
    } else {
        f32x4_t bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + j * 32 + 8 * g;
                if (p.bias != nullptr && (FULL || n < p.N)) bv[j][g] = *(const f32x4_t*)(p.bias + n);
                else bv[j][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mbase + i * 32;
            if (FULL || m < p.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = nbase + j * 32 + 8 * g;
                        if (!FULL && n >= p.N) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e] + bv[j][g][e];
                        if constexpr (EPI == PGV_EPI_BIAS_QGELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
                        }
                        if constexpr (EPI == PGV_EPI_BIAS_GELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                        }
                        if constexpr (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID) {
                            f32x4_t* rp = (f32x4_t*)(p.C + ((size_t)m * p.ldc + n) * 4);
                            f32x4_t o = *rp;
                            o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
                            *rp = o;
                        } else if constexpr (EPI == PGV_EPI_F32) {
                            *(f32x4_t*)(p.C + ((size_t)m * p.ldc + n) * 4) = f32x4_t{v[0], v[1], v[2], v[3]};
                        } else {
                            *(u32x2_t*)(p.C + ((size_t)m * p.ldc + n) * 2) = pack4<T>(v[0], v[1], v[2], v[3]);
                        }
                    }
            }
        }
    }
}

template <typename T, int EPI>
__device__ __forceinline__ void epilogue(const KArgs& p, const f32x16_t (&acc)[4][2], int m0, int n0, int bm, int bn, int wr, int wc, int l31, int hi) {
    const int mbase = m0 + wr * 128 + l31;
    const int nbase = n0 + wc * 64 + 4 * hi;
    const int obase = (n0 + wc * 64) / 2 + 4 * hi;
    if (m0 + bm <= p.M && n0 + bn <= p.N) epilogue_store<T, EPI, true>(p, acc, mbase, nbase, obase);     // block-uniform fast path
    else epilogue_store<T, EPI, false>(p, acc, mbase, nbase, obase);
}


// Tile order: block b runs on XCD b%8, so each XCD gets a contiguous run of tile ids; inside that run tiles are walked
// in bands of GM tile-rows, column-major inside a band, so the ~32 workgroups an XCD runs concurrently form a GM x (32/GM)
// patch that shares GM A-panels and 32/GM W-panels through the XCD's L2 (instead of 1 A-panel and 32 W-panels).
__device__ __forceinline__ void tile_coords(int ntm, int ntn, int& tm, int& tn) {
    constexpr int GM = 4;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int band = t / (GM * ntn);
    const int idx = t - band * GM * ntn;
    const int rows = min(GM, ntm - band * GM);
    tn = idx / rows;
    tm = band * GM + (idx - tn * rows);
}

// Tile configuration: WR x WC waves, each wave owns a 128x64 sub-tile (4x2 32x32x16 accumulators) -> BM = 128*WR,
// BN = 64*WC; BK = 64 or 32.  LDS rows are BK*2 bytes; the 16-B chunk index is XOR-swizzled with
//   BK=64: (row>>1)&7      BK=32: (row>>2)&3
// (each makes the 16 lanes a ds_read_b128 services together land on 16 distinct 16-B slots of the 256-B bank row).
template <int WR_, int WC_, int BK_>
struct Cfg {
    static constexpr int WR = WR_, WC = WC_, BK = BK_;
    static constexpr int NW = WR * WC, NT = NW * 64;
    static constexpr int BM = 128 * WR, BN = 64 * WC;
    static constexpr int RB = BK * 2;                 // bytes per LDS row
    static constexpr int CPR = RB / 16;               // 16-B chunks per row
    static constexpr int RPI = 64 / CPR;              // tile rows covered by one wave-wide DMA instruction
    static constexpr int A_BYTES = BM * RB, W_BYTES = BN * RB, STAGE = A_BYTES + W_BYTES, LDS = 2 * STAGE;
    static constexpr int AJ = BM / RPI / NW, WJ = BN / RPI / NW;   // DMA instructions per wave per operand
    static constexpr int KK = BK / 16;                // MFMA k-steps per tile
    __device__ static __forceinline__ int swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};

template <typename T, int EPI, typename CF>
__global__ __launch_bounds__(CF::NT, 2) void gemm_nt(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = CF::BM, BN = CF::BN, BK = CF::BK, RB = CF::RB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w / CF::WC, wc = w - wr * CF::WC;

    int tm, tn;
    tile_coords(p.ntm, p.ntn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: DMA instruction j of wave w covers tile rows (j*NW + w)*RPI .. +RPI; lane -> (row, slot) ----
    const int srow = lane / CF::CPR, slot = lane % CF::CPR;
    const char* ga[CF::AJ];
    const char* gw[CF::WJ];
#pragma unroll
    for (int j = 0; j < CF::AJ; ++j) {
        const int row = (j * CF::NW + w) * CF::RPI + srow;
        const int chunk = slot ^ CF::swz(row);
        const int ra = min(m0 + row, p.M - 1);
        ga[j] = p.A + ((size_t)ra * p.lda + chunk * 8) * 2;
    }
#pragma unroll
    for (int j = 0; j < CF::WJ; ++j) {
        const int row = (j * CF::NW + w) * CF::RPI + srow;
        const int chunk = slot ^ CF::swz(row);
        const int rw = min(n0 + row, p.N - 1);
        if (p.wblk)   // 1 KiB block (rw/16, k/32); inside: slot ((k%32)/8)*16 + rw%16
            gw[j] = p.W + (((size_t)(rw >> 4) * (p.K >> 5) + (chunk >> 2)) * 512 + (((chunk & 3) << 4) + (rw & 15)) * 8) * 2;
        else
            gw[j] = p.W + ((size_t)rw * p.ldw + chunk * 8) * 2;
    }
    const size_t wstep = p.wblk ? (size_t)(BK / 32) * 1024 : (size_t)BK * 2;     // bytes per K tile on the W side
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * CF::STAGE;
        const size_t koff = (size_t)kt * BK * 2;
#pragma unroll
        for (int j = 0; j < CF::AJ; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[j] + koff),
                                             (__attribute__((address_space(3))) void*)(base + (j * CF::NW + w) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < CF::WJ; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[j] + (size_t)kt * wstep),
                                             (__attribute__((address_space(3))) void*)(base + CF::A_BYTES + (j * CF::NW + w) * 1024), 16, 0, 0);
    };

    // ---- fragment read offsets (bytes) ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = CF::swz(l31);
    int koffs[CF::KK];
#pragma unroll
    for (int kk = 0; kk < CF::KK; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const int a_row_off = (wr * 128 + l31) * RB;                          // + i*32*RB
    const int w_row_off = CF::A_BYTES + (wc * 64 + l31) * RB;             // + j*32*RB

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const char* sb = smem + buf * CF::STAGE;
#pragma unroll
        for (int kk = 0; kk < CF::KK; ++kk) {
            typename T::v8 af[4], wf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const typename T::v8*)(sb + a_row_off + i * 32 * RB + koffs[kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *(const typename T::v8*)(sb + w_row_off + j * 32 * RB + koffs[kk]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma32(wf[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of tile kt+1 has landed ...
        __syncthreads();                                   // ... for every wave, and reads of `buf` are done before it is restaged
    }

    epilogue<T, EPI>(p, acc, m0, n0, BM, BN, wr, wc, l31, hi);
}

// =================================================================================================
// 8-phase schedule (256x256x64 tile, 8 waves as 2(M) x 4(N)).  Per K tile each wave runs 4 phases, each
// phase = LOAD segment | barrier | COMPUTE segment (8 x v_mfma_f32_32x32x16 = one 64x32 quadrant x K=64) | barrier.
// The wave rows are offset by one barrier, so while one wave of a SIMD is in its MFMA segment (priority 1) its
// partner is in its LOAD segment (ds_read_b128 of the next fragments + 2 DMA instructions of the NEXT tile).
// Operand tiles are staged as four 16 KiB "parts" per K tile, ordered by first use:
//   A-part mh = rows {wr*128 + mh*64 .. +64}, W-part nh = rows {wc*64 + nh*32 .. +32};
//   phase 1 needs A0,W0; phase 2 W1; phase 3 A1; phase 4 nothing (both W halves stay in registers).
// Part i of tile t+1 is issued in phase i of tile t and awaited with a COUNTED s_waitcnt vmcnt(4) (two parts stay
// in flight across every barrier; vmcnt(0) only in the last tile) at the end of the LOAD segment that precedes its
// first reader by a full barrier for both wave groups.
// =================================================================================================
// ABL (diagnostic ablation, normally 0): bit0 = no DMA inside the main loop, bit1 = no ds_reads, bit2 = no MFMA.
template <typename T, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_8ph(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256;
    constexpr int PART = 16384, BUF = 4 * PART;          // per buffer: A0 | A1 | W0 | W1
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;

    int tm, tn;
    tile_coords(p.ntm, p.ntn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: a part = 128 LDS rows x 128 B; DMA instruction j (0,1) of wave w fills rows (j*8+w)*8 .. +8 ----
    const int srow = lane >> 3, slot = lane & 7;
    const char* gsrc[4][2];                      // [part: A0 A1 W0 W1][j]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = (j * 8 + w) * 8 + srow;                    // LDS row inside the part
        const int chunk = slot ^ ((lr >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (lr >> 6) * 128 + h * 64 + (lr & 63);                 // A-part h: (wr = lr/64, r64)
            const int ra = min(m0 + arow, p.M - 1);
            gsrc[h][j] = p.A + ((size_t)ra * p.lda + chunk * 8) * 2;
            const int wrow = (lr >> 5) * 64 + h * 32 + (lr & 31);                  // W-part h: (wc = lr/32, r32)
            const int rw = min(n0 + wrow, p.N - 1);
            if (p.wblk)
                gsrc[2 + h][j] = p.W + (((size_t)(rw >> 4) * (p.K >> 5) + (chunk >> 2)) * 512 + (((chunk & 3) << 4) + (rw & 15)) * 8) * 2;
            else
                gsrc[2 + h][j] = p.W + ((size_t)rw * p.ldw + chunk * 8) * 2;
        }
    }
    const size_t astep = 128, wstep = p.wblk ? 2048 : 128;      // bytes per K tile
    auto issue = [&](int part, int kt) {                         // 2 DMA instructions: part `part` of K tile kt
        char* base = smem + (kt & 1) * BUF + part * PART;
        const size_t off = (size_t)kt * (part < 2 ? astep : wstep);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc[part][j] + off),
                                             (__attribute__((address_space(3))) void*)(base + (j * 8 + w) * 1024), 16, 0, 0);
    };

    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const int a_off = (wr * 64 + l31) * 128;            // inside an A part (+ i*32*128)
    const int w_off = (wc * 32 + l31) * 128;            // inside a W part

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typename T::v8 af[2][4], wf[2][4];                  // A: [m-tile in half][kk]; W: [n-half][kk]
    if constexpr (ABL != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { af[i][kk] = *(const typename T::v8*)(smem + a_off + i * 4096 + koffs[kk]); wf[i][kk] = af[i][kk]; }
    }
    auto read_a = [&](const char* buf, int mh) {
        if constexpr (ABL & 2) { asm volatile("" : "+v"(af[0][0])); return; }
        const char* base = buf + mh * PART + a_off;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) af[i][kk] = *(const typename T::v8*)(base + i * 4096 + koffs[kk]);
    };
    auto read_w = [&](const char* buf, int nh) {
        if constexpr (ABL & 2) { asm volatile("" : "+v"(wf[0][0])); return; }
        const char* base = buf + (2 + nh) * PART + w_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wf[nh][kk] = *(const typename T::v8*)(base + koffs[kk]);
    };
    auto compute = [&](int mh, int nh) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (ABL & 4) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) { asm volatile("" :: "v"(wf[nh][kk]), "v"(af[i][kk])); }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[mh * 2 + i][nh] = T::mfma32(wf[nh][kk], af[i][kk], acc[mh * 2 + i][nh]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };

    const int nk = p.K / 64;
    // prologue: all four parts of tile 0
    issue(0, 0); issue(2, 0); issue(3, 0); issue(1, 0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // A0(0), W0(0) landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // wave row 1 runs one barrier behind wave row 0

#define PGV_PHASE(ISSUE, READS, MH, NH, WAITASM)                                       \
    {                                                                                    \
        ISSUE;                                                                           \
        READS;                                                                           \
        asm volatile(WAITASM ::: "memory");                                              \
        __builtin_amdgcn_s_barrier();                                                    \
        compute(MH, NH);                                                                 \
        __builtin_amdgcn_s_barrier();                                                    \
    }
#define PGV_ISSUE_NEXT { if (!(ABL & 1)) { issue(0, kt + 1); issue(2, kt + 1); issue(3, kt + 1); issue(1, kt + 1); } }
    // The buffer of tile kt+1 was last read in phase 3 of tile kt-1, so all 8 DMA instructions of tile kt+1 are issued at
    // the top of tile kt (order of first use A0, W0, W1, A1).  In-flight accounting per wave (2 instructions per part):
    //   end of phase 1 (W1(kt) must have landed)      : younger = A1(kt) + 4 parts of kt+1 -> vmcnt(10)
    //   end of phase 2 (A1(kt))                        : younger = 4 parts of kt+1          -> vmcnt(8)
    //   end of phase 4 (A0, W0 of kt+1)                : younger = W1, A1 of kt+1           -> vmcnt(4)
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
        const char* buf = smem + (kt & 1) * BUF;
        PGV_PHASE(PGV_ISSUE_NEXT, { read_a(buf, 0); read_w(buf, 0); }, 0, 0, "s_waitcnt vmcnt(10)")
        PGV_PHASE({}, { read_w(buf, 1); }, 0, 1, "s_waitcnt vmcnt(8)")
        PGV_PHASE({}, { read_a(buf, 1); }, 1, 1, "s_waitcnt vmcnt(8)")
        PGV_PHASE({}, { }, 1, 0, "s_waitcnt vmcnt(4)")
    }
    {   // last K tile: nothing left to issue; W1, A1 of this tile are the only DMA still in flight
        const char* buf = smem + (kt & 1) * BUF;
        PGV_PHASE({}, { read_a(buf, 0); read_w(buf, 0); }, 0, 0, "s_waitcnt vmcnt(2)")
        PGV_PHASE({}, { read_w(buf, 1); }, 0, 1, "s_waitcnt vmcnt(0)")
        PGV_PHASE({}, { read_a(buf, 1); }, 1, 1, "s_waitcnt vmcnt(0)")
        PGV_PHASE({}, { }, 1, 0, "s_waitcnt vmcnt(0)")
    }
#undef PGV_ISSUE_NEXT
#undef PGV_PHASE
    if (wr == 0) __builtin_amdgcn_s_barrier();           // rebalance the barrier count of the two wave rows

    epilogue<T, EPI>(p, acc, m0, n0, BM, BN, wr, wc, l31, hi);
}

template <typename T, int EPI, int ABL>
int launch_8ph_abl(KArgs k, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_8ph<T, EPI, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e != hipSuccess) { pgv_set_error("hipFuncSetAttribute(gemm8): %s", hipGetErrorString(e)); return PGV_EHIP; }
        configured = true;
    }
    k.ntm = (k.M + 255) / 256; k.ntn = (k.N + 255) / 256;
    hipLaunchKernelGGL((gemm_nt_8ph<T, EPI, ABL>), dim3(k.ntm * k.ntn), dim3(512), 131072, s, k);
    return PGV_OK;
}

template <typename T, int EPI>
int launch_8ph(KArgs k, hipStream_t s) {
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("PGV_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if constexpr (EPI == PGV_EPI_BIAS) {      // diagnostic ablations exist for the plain epilogue only
        switch (abl) {
            case 1: return launch_8ph_abl<T, EPI, 1>(k, s);
            case 2: return launch_8ph_abl<T, EPI, 2>(k, s);
            case 3: return launch_8ph_abl<T, EPI, 3>(k, s);
            case 4: return launch_8ph_abl<T, EPI, 4>(k, s);
            case 5: return launch_8ph_abl<T, EPI, 5>(k, s);
            case 6: return launch_8ph_abl<T, EPI, 6>(k, s);
            default: break;
        }
    }
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_8ph<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e != hipSuccess) { pgv_set_error("hipFuncSetAttribute(gemm8): %s", hipGetErrorString(e)); return PGV_EHIP; }
        configured = true;
    }
    k.ntm = (k.M + 255) / 256; k.ntn = (k.N + 255) / 256;
    hipLaunchKernelGGL((gemm_nt_8ph<T, EPI>), dim3(k.ntm * k.ntn), dim3(512), 131072, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pgv_set_error("gemm8 launch: %s", hipGetErrorString(e)); return PGV_EHIP; }
    return PGV_OK;
}

using CfgA = Cfg<2, 4, 64>;    // 256x256, 8 waves, BK 64, 128 KiB LDS: one workgroup per CU
using CfgB = Cfg<2, 2, 32>;    // 256x128, 4 waves, BK 32,  48 KiB LDS: two independent workgroups per CU
using CfgC = Cfg<1, 4, 32>;    // 128x256, 4 waves, BK 32,  48 KiB LDS

template <typename T, int EPI, typename CF>
int launch(KArgs k, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt<T, EPI, CF>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS);
        if (e != hipSuccess) { pgv_set_error("hipFuncSetAttribute(gemm): %s", hipGetErrorString(e)); return PGV_EHIP; }
        configured = true;
    }
    k.ntm = (k.M + CF::BM - 1) / CF::BM; k.ntn = (k.N + CF::BN - 1) / CF::BN;
    hipLaunchKernelGGL((gemm_nt<T, EPI, CF>), dim3(k.ntm * k.ntn), dim3(CF::NT), CF::LDS, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pgv_set_error("gemm launch: %s", hipGetErrorString(e)); return PGV_EHIP; }
    return PGV_OK;
}

int gemm_cfg_override() {
    static int v = -2;
    if (v == -2) { const char* e = getenv("PGV_GEMM_CFG"); v = e ? atoi(e) : -1; }
    return v;
}

template <typename T, int EPI>
int launch_cfg(const KArgs& k, hipStream_t s) {
    int cfg = gemm_cfg_override();
    if (cfg < 0) cfg = 0;
    switch (cfg) {
        case 1: return launch<T, EPI, CfgB>(k, s);
        case 2: return launch<T, EPI, CfgC>(k, s);
        case 3: return launch_8ph<T, EPI>(k, s);
        default: return launch<T, EPI, CfgA>(k, s);
    }
}

template <typename T>
int dispatch_epi(int epi, const KArgs& k, hipStream_t s) {
    switch (epi) {
        case PGV_EPI_NONE:       // bias pointer is null -> same code path as BIAS
        case PGV_EPI_BIAS:       return launch_cfg<T, PGV_EPI_BIAS>(k, s);
        case PGV_EPI_BIAS_QGELU: return launch_cfg<T, PGV_EPI_BIAS_QGELU>(k, s);
        case PGV_EPI_BIAS_GELU:  return launch_cfg<T, PGV_EPI_BIAS_GELU>(k, s);
        case PGV_EPI_RESID:
        case PGV_EPI_BIAS_RESID: return launch_cfg<T, PGV_EPI_BIAS_RESID>(k, s);
        case PGV_EPI_SWIGLU:     return launch_cfg<T, PGV_EPI_SWIGLU>(k, s);
        case PGV_EPI_F32:        return launch_cfg<T, PGV_EPI_F32>(k, s);
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
    const bool out32 = (a.epi == PGV_EPI_RESID || a.epi == PGV_EPI_BIAS_RESID || a.epi == PGV_EPI_F32);
    PGV_CHECK(((uintptr_t)a.C & (out32 ? 15 : 7)) == 0 && a.ldc % 4 == 0, "gemm: C misaligned (ptr/ldc)");
    KArgs k;
    k.A = (const char*)a.A; k.W = (const char*)a.W; k.bias = a.bias; k.C = (char*)a.C;
    k.lda = a.lda; k.ldw = a.ldw; k.ldc = a.ldc; k.M = a.M; k.N = a.N; k.K = a.K;
    k.ntm = 0; k.ntn = 0;
    k.wblk = a.w_blocked ? 1 : 0;
    PGV_CHECK(!a.w_blocked || a.N % 16 == 0, "gemm: blocked weights need N %% 16 == 0");
    pgv_prof_begin(ctx, 0, s);
    int rc;
    if (dtype == PGV_F16) rc = dispatch_epi<TF16>(a.epi, k, s);
    else if (dtype == PGV_BF16) rc = dispatch_epi<TBF16>(a.epi, k, s);
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
