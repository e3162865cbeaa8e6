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
#include <type_traits>

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


// =================================================================================================
// Persistent ping-pong kernel ("pp"): the 8-phase schedule above with three changes.
//  1. Persistent: gridDim.x = min(#tiles, #CUs) workgroups walk the XCD-ordered tile list with stride gridDim.x and
//     treat all their K tiles as ONE stream of K-steps, so the DMA pipeline never drains between output tiles: the
//     operands of the next tile are already landing in LDS while the current tile's epilogue runs.
//  2. One 16 KiB part (2 DMA instructions per wave) is issued per phase, each part as early as its LDS slot allows
//     (2 phases after its last reader), i.e. 5-6 phases (~1.5 K-steps) ahead of its first reader:
//        phase 1 of step s: W1(s+1)   phase 2: A1(s+1)   phase 3: A0(s+2)   phase 4: W0(s+2)
//     Per wave 10-12 DMA instructions are in flight at any time; the counted wait at the end of a load segment is
//     always vmcnt(8) (four younger parts), vmcnt(0) only in the last two K-steps of the workgroup.
//  3. Epilogue through LDS: a wave transposes its 128x64 sub-tile in 4 KiB pieces through a private staging area (LDS
//     bytes 128K..160K) so that every global store instruction writes 8 full 128-byte lines (row-contiguous) instead of
//     32 partial lines; the fp32 residual read-modify-write uses the same full-line shape.
// LDS: 2 x 64 KiB operand buffers + 32 KiB staging = 160 KiB, one workgroup per CU.
// =================================================================================================
// Buffer descriptor over `bytes` bytes at `base`, built from provably wave-uniform inputs (cdna_hip_programming.md T20):
// out-of-range lanes of a raw buffer load/store are dropped by the hardware, so edge tiles need no exec-masked branches
// (an exec-masked VMEM op inside the persistent loop makes hipcc drain vmcnt(0) at the loop header and kills the pipeline).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const uintptr_t b = (uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr unsigned OOB = 0x80000000u;     // voffset of a lane whose store/load must be dropped

template <typename T, int EPI>
__device__ __forceinline__ void epilogue_pp(const KArgs& p, const f32x16_t (&acc)[4][2], char* stg, int m0w, int n0w, int lane) {
    // m0w/n0w: first row / column of this wave's 128x64 sub-tile.  stg: 4 KiB private LDS.  Branch-free for edge tiles.
    const int l31 = lane & 31, hi = lane >> 5;
    const int rrow = lane >> 3, rc = lane & 7;                 // read side: 8 rows x 8 chunks of 16 B per instruction
    constexpr bool OUT32 = (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID || EPI == PGV_EPI_F32);
    constexpr int ES = OUT32 ? 4 : 2;
    const int rows_valid = max(0, min(128, p.M - m0w));
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.C + (size_t)m0w * p.ldc * ES, (unsigned)rows_valid * (unsigned)p.ldc * ES);
    const unsigned rowpitch = (unsigned)p.ldc * ES;
    if constexpr (EPI == PGV_EPI_SWIGLU) {
        // W rows interleaved per 64: [32 gate | 32 up]; this wave's 64 accumulator columns -> 32 output columns
        const int ocol = n0w / 2 + 4 * hi;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = silu_f(acc[i][0][g * 4 + e]) * acc[i][1][g * 4 + e];
                const int n = ocol + 8 * g;
                const unsigned off = (n < p.N / 2) ? (unsigned)(i * 32 + l31) * rowpitch + (unsigned)n * 2 : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(v[0], v[1], v[2], v[3]), rsrc, off, 0, 0);
            }
    } else if constexpr (!OUT32) {
        f32x4_t bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = min(n0w + j * 32 + 8 * g + 4 * hi, p.N - 4);
                bv[j][g] = *(const f32x4_t*)(p.bias + n);     // never null here (ctx->zero_bias stands in)
            }
        const int col = n0w + rc * 8;
        const unsigned voff = (col < p.N) ? (unsigned)rrow * rowpitch + (unsigned)col * 2 : OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // piece = rows 32i..32i+31 x 64 columns x 2 B: row stride 128 B, 16-B chunk index XORed with (row & 7)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][g * 4 + e] + bv[j][g][e];
                        if constexpr (EPI == PGV_EPI_BIAS_QGELU) v[e] = quick_gelu_f(v[e]);
                        if constexpr (EPI == PGV_EPI_BIAS_GELU) v[e] = gelu_erf_f(v[e]);
                    }
                    *(u32x2_t*)(stg + l31 * 128 + (((4 * j + g) ^ (l31 & 7)) << 4) + 8 * hi) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r * 8 + rrow;
                const u32x4_t d = *(const u32x4_t*)(stg + row * 128 + ((rc ^ (row & 7)) << 4));
                __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voff + (i * 32 + r * 8) * rowpitch, 0, 0);   // soffset is not range-checked
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        constexpr bool RMW = (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID);
        // piece (i, j) = rows 32i.. x columns 32j.. x 4 B: row stride 128 B (32 fp32), chunk = 4 floats
        unsigned voff[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0w + j * 32 + rc * 4;
            voff[j] = (col < p.N) ? (unsigned)rrow * rowpitch + (unsigned)col * 4 : OOB;
        }
        f32x4_t bv[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = min(n0w + j * 32 + 8 * g + 4 * hi, p.N - 4);
                bv[j][g] = *(const f32x4_t*)(p.bias + n);     // never null here (ctx->zero_bias stands in)
            }
        u32x4_t cur[4], nxt[4];
        if constexpr (RMW) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[0] + (r * 8) * rowpitch, 0, 0);
        }
#pragma unroll
        for (int piece = 0; piece < 8; ++piece) {
            const int i = piece >> 1, j = piece & 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][g * 4 + e] + bv[j][g][e];
                *(f32x4_t*)(stg + l31 * 128 + (((2 * g + hi) ^ (l31 & 7)) << 4)) = v;
            }
            if constexpr (RMW) {
                if (piece < 7) {
                    const int i2 = (piece + 1) >> 1, j2 = (piece + 1) & 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) nxt[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[j2] + (i2 * 32 + r * 8) * rowpitch, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r * 8 + rrow;
                f32x4_t d = *(const f32x4_t*)(stg + row * 128 + ((rc ^ (row & 7)) << 4));
                if constexpr (RMW) {
                    const f32x4_t o = __builtin_bit_cast(f32x4_t, cur[r]);
                    d[0] += o[0]; d[1] += o[1]; d[2] += o[2]; d[3] += o[3];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, d), rsrc, voff[j] + (i * 32 + r * 8) * rowpitch, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (RMW) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
            }
        }
    }
}

__device__ __forceinline__ void tile_coords_v(int vb, int nwg, int ntm, int ntn, int& tm, int& tn) {
    constexpr int GM = 4;
    const int q = nwg >> 3, r = nwg & 7, xcd = vb & 7, loc = vb >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int band = t / (GM * ntn);
    const int idx = t - band * GM * ntn;
    const int rows = min(GM, ntm - band * GM);
    tn = idx / rows;
    tm = band * GM + (idx - tn * rows);
}

// OPT: reserved for A/B experiments.  ABL as in gemm_nt_8ph.
template <int P> using PartC = std::integral_constant<int, P>;
using Guarded = std::true_type;
using Steady = std::false_type;

template <typename T, int EPI, int OPT = 1, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_pp(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256;
    constexpr int PART = 16384, BUF = 4 * PART, STG = 2 * BUF;      // buffer: A0 | A1 | W0 | W1
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int total = p.ntm * p.ntn, G = gridDim.x;
    const int nk = p.K >> 6;
    const int S = ((total - (int)blockIdx.x + G - 1) / G) * nk;        // K-steps of this workgroup

    // ---- DMA issue state: one cursor per part (A0 A1 W0 W1) ----
    const int srow = lane >> 3, slot = lane & 7;
    const char* ptr[4][2];
    int stepi[4], kleft[4], vbn[4];
    const size_t wstep = p.wblk ? 2048 : 128;
    auto rebase = [&](auto part_c, int vb) __attribute__((always_inline)) {
        constexpr int part = decltype(part_c)::value;
        int tm, tn;
        tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);
        const int h = part & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lr = (j * 8 + w) * 8 + srow;
            const int chunk = slot ^ ((lr >> 1) & 7);
            if (part < 2) {
                const int arow = (lr >> 6) * 128 + h * 64 + (lr & 63);
                const int ra = min(tm * BM + arow, p.M - 1);
                ptr[part][j] = p.A + ((size_t)ra * p.lda + chunk * 8) * 2;
            } else {
                const int wrow = (lr >> 5) * 64 + h * 32 + (lr & 31);
                const int rw = min(tn * BN + wrow, p.N - 1);
                if (p.wblk)
                    ptr[part][j] = p.W + (((size_t)(rw >> 4) * (p.K >> 5) + (chunk >> 2)) * 512 + (((chunk & 3) << 4) + (rw & 15)) * 8) * 2;
                else
                    ptr[part][j] = p.W + ((size_t)rw * p.ldw + chunk * 8) * 2;
            }
        }
    };
    auto issue = [&](auto guard, auto part_c) __attribute__((always_inline)) {
        constexpr int part = decltype(part_c)::value;
        if (!decltype(guard)::value || stepi[part] < S) {
            char* base = smem + (stepi[part] & 1) * BUF + part * PART;
            if constexpr (!(ABL & 1)) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ptr[part][j],
                                                     (__attribute__((address_space(3))) void*)(base + (j * 8 + w) * 1024), 16, 0, 0);
            }
            const size_t adv = part < 2 ? (size_t)128 : wstep;
            ptr[part][0] += adv; ptr[part][1] += adv;
            ++stepi[part];
            if (__builtin_expect(--kleft[part] == 0, 0)) {
                kleft[part] = nk;
                vbn[part] += G;
                if (stepi[part] < S) rebase(part_c, vbn[part]);
            }
        }
    };
#pragma unroll
    for (int part = 0; part < 4; ++part) { stepi[part] = 0; kleft[part] = nk; vbn[part] = blockIdx.x; }
    rebase(PartC<0>{}, blockIdx.x); rebase(PartC<1>{}, blockIdx.x); rebase(PartC<2>{}, blockIdx.x); rebase(PartC<3>{}, blockIdx.x);

    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const int a_off = (wr * 64 + l31) * 128;
    const int w_off = (wc * 32 + l31) * 128;

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    typename T::v8 af[2][4], wf[2][4];
    if constexpr (ABL != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { af[i][kk] = *(const typename T::v8*)(smem + a_off + i * 4096 + koffs[kk]); wf[i][kk] = af[i][kk]; }
    }
    auto read_a = [&](const char* buf, int mh) __attribute__((always_inline)) {
        if constexpr (ABL & 2) { asm volatile("" : "+v"(af[0][0])); return; }
        const char* base = buf + mh * PART + a_off;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) af[i][kk] = *(const typename T::v8*)(base + i * 4096 + koffs[kk]);
    };
    auto read_w = [&](const char* buf, int nh) __attribute__((always_inline)) {
        if constexpr (ABL & 2) { asm volatile("" : "+v"(wf[0][0])); return; }
        const char* base = buf + (2 + nh) * PART + w_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wf[nh][kk] = *(const typename T::v8*)(base + koffs[kk]);
    };
    auto compute = [&](int mh, int nh) __attribute__((always_inline)) {
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

    // prologue: all of K-step 0 and the first two parts of K-step 1 (order of first use)
    issue(Guarded{}, PartC<0>{}); issue(Guarded{}, PartC<2>{}); issue(Guarded{}, PartC<3>{}); issue(Guarded{}, PartC<1>{}); issue(Guarded{}, PartC<0>{}); issue(Guarded{}, PartC<2>{});
    if (S < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // A0(0), W0(0) landed (this wave's share)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                        // wave row 1 runs one barrier behind wave row 0

    int dbg_i = 0;
    unsigned long long tts[12];
    if constexpr ((ABL & 8) && OPT == 2) { for (int i = 0; i < 12; ++i) tts[i] = 0; }
    auto stamp = [&](int s) __attribute__((always_inline)) {                                         // ABL bit3: s_memtime stamps of K-steps 8..11 of block 0
        if constexpr ((ABL & 8) && OPT != 2) {
            if (blockIdx.x == 0 && s >= 8 && s < 12) {
                const unsigned long long t = __builtin_readcyclecounter();
                if (lane == 0) *(unsigned long long*)(smem + STG + w * 4096 + dbg_i * 8) = t;
                ++dbg_i;
            }
        }
    };
#define PGV_PP_PHASE(TAG, READS, PART_TO_ISSUE, MH, NH, WAIT, ENDBAR)                   \
    {                                                                                    \
        stamp(s);                                                                        \
        READS;                                                                           \
        issue(TAG{}, PartC<PART_TO_ISSUE>{});                                                   \
        if (WAIT) {                                                                      \
            if (TAG::value) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             \
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                        \
        }                                                                                \
        __builtin_amdgcn_s_barrier();                                                    \
        stamp(s);                                                                        \
        compute(MH, NH);                                                                 \
        stamp(s);                                                                        \
        if (ENDBAR) __builtin_amdgcn_s_barrier();                                        \
        if (ENDBAR) stamp(s);                                                            \
    }
    // one K-step: TAG = Steady (s + 2 < S: every part exists, counted waits) or Guarded (last two K-steps: drained waits)
#define PGV_PP_KSTEP(TAG)                                                                                                  \
    {                                                                                                                       \
        const char* buf = smem + (s & 1) * BUF;                                                                             \
        if constexpr ((ABL & 8) && OPT == 2) { if (s >= nk - 3 && s < nk + 5) tts[s - (nk - 3)] = __builtin_readcyclecounter(); }        \
        PGV_PP_PHASE(TAG, { read_a(buf, 0); read_w(buf, 0); }, 3, 0, 0, true, true)   /* issues W1(s+1); wait: W1(s)           */ \
        PGV_PP_PHASE(TAG, { read_w(buf, 1); }, 1, 0, 1, true, true)                     /* issues A1(s+1); wait: A1(s)           */ \
        PGV_PP_PHASE(TAG, { read_a(buf, 1); }, 0, 1, 1, false, true)                    /* issues A0(s+2)                        */ \
        PGV_PP_PHASE(TAG, { }, 2, 1, 0, true, false)                                     /* issues W0(s+2); wait: A0, W0 of s+1   */ \
        /* The end barrier of phase 4 is taken by wave row 0 BEFORE the tile-end block and by wave row 1 (which runs one      \
           barrier behind) AFTER it, so both rows run their epilogues in the same barrier interval instead of back to back. */ \
        if (wr == 0) __builtin_amdgcn_s_barrier();                                                                          \
        if (++kt == nk) {                                                                                                   \
            kt = 0;                                                                                                         \
            int tm, tn;                                                                                                     \
            tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);                                                                 \
            if constexpr ((ABL & 8) && OPT == 2) { if (s == nk - 1) tts[8] = __builtin_readcyclecounter(); }                \
            if constexpr (!(ABL & 8) || OPT == 2) epilogue_pp<T, EPI>(p, acc, smem + STG + w * 4096, tm * BM + wr * 128, tn * BN + wc * 64, lane); \
            if constexpr ((ABL & 8) && OPT == 2) { if (s == nk - 1) tts[9] = __builtin_readcyclecounter(); }                \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                               \
                    _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;                                      \
            vb += G;                                                                                                        \
        }                                                                                                                   \
        if (wr == 1) __builtin_amdgcn_s_barrier();                                                                          \
        stamp(s);                                                                                                           \
    }
    int kt = 0, vb = blockIdx.x, s = 0;
    for (; s + 2 < S; ++s) PGV_PP_KSTEP(Steady)
    for (; s < S; ++s) PGV_PP_KSTEP(Guarded)
#undef PGV_PP_KSTEP
#undef PGV_PP_PHASE
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if constexpr (ABL & 8) {                     // dump the stamps of block 0 (8 waves x 64) into the first 4 KiB of C
        if (blockIdx.x == 0 && lane == 0) {
            if constexpr (OPT == 2) { for (int i = 0; i < 12; ++i) ((unsigned long long*)p.C)[w * 64 + i] = tts[i]; }
            else for (int i = 0; i < 64; ++i) ((unsigned long long*)p.C)[w * 64 + i] = *(unsigned long long*)(smem + STG + w * 4096 + i * 8);
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));
        }
    }
}

template <typename T, int EPI, int OPT, int ABL>
int launch_pp_inst(KArgs k, hipStream_t s, int num_cu) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_pp<T, EPI, OPT, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if (e != hipSuccess) { pgv_set_error("hipFuncSetAttribute(gemm_pp): %s", hipGetErrorString(e)); return PGV_EHIP; }
        configured = true;
    }
    k.ntm = (k.M + 255) / 256; k.ntn = (k.N + 255) / 256;
    const int total = k.ntm * k.ntn;
    const int grid = total < num_cu ? total : num_cu;
    hipLaunchKernelGGL((gemm_pp<T, EPI, OPT, ABL>), dim3(grid), dim3(512), 163840, s, k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pgv_set_error("gemm_pp launch: %s", hipGetErrorString(e)); return PGV_EHIP; }
    return PGV_OK;
}

template <typename T, int EPI>
int launch_pp(const KArgs& k, hipStream_t s, int num_cu, int opt) {
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("PGV_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if constexpr (EPI == PGV_EPI_BIAS) {
        switch (abl) {
            case 1: return launch_pp_inst<T, EPI, 1, 1>(k, s, num_cu);
            case 3: return launch_pp_inst<T, EPI, 1, 3>(k, s, num_cu);
            case 4: return launch_pp_inst<T, EPI, 1, 4>(k, s, num_cu);
            case 6: return launch_pp_inst<T, EPI, 1, 6>(k, s, num_cu);
            case 7: return launch_pp_inst<T, EPI, 1, 7>(k, s, num_cu);
            case 8: return launch_pp_inst<T, EPI, 1, 8>(k, s, num_cu);
            case 9: return launch_pp_inst<T, EPI, 2, 8>(k, s, num_cu);
            default: break;
        }
    }
    (void)opt;
    return launch_pp_inst<T, EPI, 1, 0>(k, s, num_cu);
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
    constexpr bool OUT32 = (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID || EPI == PGV_EPI_F32);
    constexpr bool RMW = (EPI == PGV_EPI_RESID || EPI == PGV_EPI_BIAS_RESID);
    constexpr bool SWIGLU = (EPI == PGV_EPI_SWIGLU);
    constexpr int ES = OUT32 ? 4 : 2;
    const int rows_valid = max(0, min(128, p.M - m0w));
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.C + (size_t)m0w * p.ldc * ES, (unsigned)rows_valid * (unsigned)p.ldc * ES);
    const unsigned rowpitch = (unsigned)p.ldc * ES;
    unsigned voff[2];
    f32x4_t bv[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
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
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) {
        const int i = piece >> 1, jp = piece & 1;
        u32x4_t old[8];
        if constexpr (RMW) {
#pragma unroll
            for (int r = 0; r < 8; ++r) old[r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[jp] + (i * 32 + r * 4) * rowpitch, 0, 0);
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
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d[e] += bv[jp][e];
                    if constexpr (EPI == PGV_EPI_BIAS_QGELU) d[e] = quick_gelu_f(d[e]);
                    if constexpr (EPI == PGV_EPI_BIAS_GELU) d[e] = gelu_erf_f(d[e]);
                }
                if constexpr (RMW) {
                    const f32x4_t q = __builtin_bit_cast(f32x4_t, old[r]);
                    d[0] += q[0]; d[1] += q[1]; d[2] += q[2]; d[3] += q[3];
                }
                if constexpr (OUT32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, d), rsrc, o, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(d[0], d[1], d[2], d[3]), rsrc, o, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T, int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 1) void gemm_w4(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = 256, ABYTES = 32768, BUF = 65536, STG = 2 * BUF;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int total = p.ntm * p.ntn, G = gridDim.x;
    const int nk = p.K >> 6;
    const int S = ((total - (int)blockIdx.x + G - 1) / G) * nk;        // K-steps of this workgroup

    // ---- DMA cursor: instruction jj (0..7: A, 8..15: W) of wave w fills tile rows ((jj&7)*4 + w)*8 .. +8 ----
    const int srow = lane >> 3, slot = lane & 7;
    const int chunk = slot ^ ((4 * w + (srow >> 1)) & 7);              // (row >> 1) & 7 for row = 32 j + 8 w + srow
    unsigned voffA[8], voffW[8];
    __amdgpu_buffer_rsrc_t rsA, rsW;
    int d_step = 0, d_kt = 0, d_vb = blockIdx.x;
    unsigned d_ka = 0, d_kw = 0;                                        // byte offsets of the cursor's K-step inside a row
    const unsigned wkstep = p.wblk ? 2048u : 128u;
    auto rebase = [&](int vb) __attribute__((always_inline)) {
        int tm, tn;
        tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        const int rowsA = min(BM, p.M - m0), rowsW = min(BN, p.N - n0);
        if (vb >= total) {          // cursor ran past this workgroup's last tile: zero-size descriptors, the DMA writes zeros nobody reads
            rsA = make_rsrc(p.A, 0u); rsW = make_rsrc(p.W, 0u);
            return;
        }
        rsA = make_rsrc(p.A + (size_t)m0 * p.lda * 2, (unsigned)rowsA * (unsigned)p.lda * 2u);
        if (p.wblk) rsW = make_rsrc(p.W, (unsigned)(((size_t)((p.N + 15) & ~15) * p.K * 2) > 0xffffffffull ? 0xffffffffu : (size_t)((p.N + 15) & ~15) * p.K * 2));
        else rsW = make_rsrc(p.W + (size_t)n0 * p.ldw * 2, (unsigned)rowsW * (unsigned)p.ldw * 2u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 32 * j + 8 * w + srow;
            voffA[j] = (unsigned)min(row, rowsA - 1) * (unsigned)p.lda * 2u + chunk * 16;
            if (p.wblk) {
                const int rw = n0 + min(row, rowsW - 1);      // 1 KiB block (rw/16, k/32); inside: slot ((k%32)/8)*16 + rw%16
                voffW[j] = (unsigned)((((size_t)(rw >> 4) * (p.K >> 5) + (chunk >> 2)) * 512 + (((chunk & 3) << 4) + (rw & 15)) * 8) * 2);
            } else {
                voffW[j] = (unsigned)min(row, rowsW - 1) * (unsigned)p.ldw * 2u + chunk * 16;
            }
        }
    };
    auto dma_half = [&](auto half_c) __attribute__((always_inline)) {   // half 0: jj 0..7 (A), half 1: jj 8..15 (W), of K-step d_step
        constexpr int half = decltype(half_c)::value;
        char* base = smem + (d_step & 1) * BUF + half * ABYTES + w * 1024;
        if constexpr (!(ABL & 1)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(half == 0 ? rsA : rsW, (__attribute__((address_space(3))) void*)(base + j * 4096), 16,
                                                         half == 0 ? voffA[j] : voffW[j], half == 0 ? d_ka : d_kw, 0, 0);
        }
        if constexpr (half == 1) {                                       // cursor moves on after the second half
            ++d_step; d_ka += 128; d_kw += wkstep;
            if (__builtin_expect(++d_kt == nk, 0)) {
                d_kt = 0; d_ka = 0; d_kw = 0; d_vb += G;
                rebase(d_vb);
            }
        }
    };
    rebase(blockIdx.x);

    const int l31 = lane & 31, hi = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    int koffs[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koffs[kk] = ((kk * 2 + hi) ^ sw) << 4;
    const int a_off = (wr * 128 + l31) * 128;
    const int w_off = ABYTES + (wc * 128 + l31) * 128;

    f32x16_t acc[4][4];                                                   // written first by the C = 0 MFMAs of each tile

    typename T::v8 fa[2][4], fw[2][4];
    auto read_frags = [&](const char* buf, int kk, auto set_c) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *(const typename T::v8*)(buf + a_off + i * 4096 + koffs[kk]);
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[set][j] = *(const typename T::v8*)(buf + w_off + j * 4096 + koffs[kk]);
    };
    // The accumulators live in the AGPR half of the register file ("+a"): hipcc's own allocation of 256 accumulator
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
    // One group: 16 MFMAs on fragment set SET; between every two MFMAs one ds_read_b128 of the NEXT fragment set (k slice rkk of
    // rbuf) and, when DMA >= 0, one LDS-DMA instruction of half DMA of the cursor's K-step.
    auto group = [&](auto first_c, auto set_c, const char* rbuf, int rkk, auto dma_c) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value, dmah = decltype(dma_c)::value;
        char* dbase = smem + (d_step & 1) * BUF + (dmah > 0 ? ABYTES : 0) + w * 1024;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            mfma(first_c, acc[(2 * t) & 3][(2 * t) >> 2], fw[set][(2 * t) >> 2], fa[set][(2 * t) & 3]);
            mfma(first_c, acc[(2 * t + 1) & 3][(2 * t + 1) >> 2], fw[set][(2 * t + 1) >> 2], fa[set][(2 * t + 1) & 3]);
            if constexpr (!(ABL & 2)) {
                if (t < 4) fa[set ^ 1][t] = *(const typename T::v8*)(rbuf + a_off + t * 4096 + koffs[rkk]);
                else fw[set ^ 1][t - 4] = *(const typename T::v8*)(rbuf + w_off + (t - 4) * 4096 + koffs[rkk]);
            }
            if constexpr (dmah >= 0 && !(ABL & 1)) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dmah == 0 ? rsA : rsW, (__attribute__((address_space(3))) void*)(dbase + t * 4096), 16,
                                                         dmah == 0 ? voffA[t] : voffW[t], dmah == 0 ? d_ka : d_kw, 0, 0);
            }
        }
        if constexpr (dmah == 1) {                                        // cursor moves on after the second half
            ++d_step; d_ka += 128; d_kw += wkstep;
            if (__builtin_expect(++d_kt == nk, 0)) {
                d_kt = 0; d_ka = 0; d_kw = 0; d_vb += G;
                rebase(d_vb);
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using NoDma = std::integral_constant<int, -1>;
    using First = std::true_type;
    using Later = std::false_type;

    // prologue: K-step 0 completely, then the first half of K-step 1
    dma_half(S0{}); dma_half(S1{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dma_half(S0{});
    read_frags(smem, 0, S0{});

    // The DMA cursor needs no end-of-work guards: past the last K-step it runs on zero-size descriptors into free buffers.
    int s = 0;
#define PGV_W4_KSTEP(FIRST)                                                                                                \
    {                                                                                                                       \
        const char* buf = smem + (s & 1) * BUF;                                                                             \
        group(FIRST{}, S0{}, buf, 1, S1{});                                                                                 \
        group(Later{}, S1{}, buf, 2, NoDma{});                                                                              \
        group(Later{}, S0{}, buf, 3, NoDma{});                                                                              \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                         \
        __builtin_amdgcn_s_barrier();                                                                                       \
        group(Later{}, S1{}, smem + ((s + 1) & 1) * BUF, 0, S0{});                                                          \
        ++s;                                                                                                                \
    }
    for (int vb = blockIdx.x; vb < total; vb += G) {
        PGV_W4_KSTEP(First)                                           // C = 0 form on the first k slice: no zeroing pass
        for (int kt = 1; kt < nk; ++kt) PGV_W4_KSTEP(Later)
        int tm, tn;
        tile_coords_v(vb, total, p.ntm, p.ntn, tm, tn);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");             // last MFMA result -> first accumulator read
        epilogue_w4<T, EPI>(p, acc, smem + STG + w * 8192, tm * BM + wr * 128, tn * BN + wc * 128, lane);
    }
#undef PGV_W4_KSTEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the cursor's trailing DMAs must land before the LDS is released
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
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("PGV_GEMM_ABLATE"); abl = e ? atoi(e) : 0; }
    if constexpr (EPI == PGV_EPI_BIAS) {
        switch (abl) {
            case 1: return launch_w4_inst<T, EPI, 1>(k, s, num_cu);
            case 3: return launch_w4_inst<T, EPI, 3>(k, s, num_cu);
            case 4: return launch_w4_inst<T, EPI, 4>(k, s, num_cu);
            case 6: return launch_w4_inst<T, EPI, 6>(k, s, num_cu);
            default: break;
        }
    }
    return launch_w4_inst<T, EPI, 0>(k, s, num_cu);
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
int launch_cfg(const KArgs& k, hipStream_t s, int num_cu) {
    int cfg = gemm_cfg_override();
    if (cfg < 0) cfg = 0;
    switch (cfg) {
        case 4: return launch_pp<T, EPI>(k, s, num_cu, 1);
        case 6: return launch_w4<T, EPI>(k, s, num_cu);
        case 1: return launch<T, EPI, CfgB>(k, s);
        case 2: return launch<T, EPI, CfgC>(k, s);
        case 3: return launch_8ph<T, EPI>(k, s);
        default: return launch<T, EPI, CfgA>(k, s);
    }
}

template <typename T>
int dispatch_epi(int epi, const KArgs& k, hipStream_t s, int num_cu) {
    switch (epi) {
        case PGV_EPI_NONE:       // bias pointer is null -> same code path as BIAS
        case PGV_EPI_BIAS:       return launch_cfg<T, PGV_EPI_BIAS>(k, s, num_cu);
        case PGV_EPI_BIAS_QGELU: return launch_cfg<T, PGV_EPI_BIAS_QGELU>(k, s, num_cu);
        case PGV_EPI_BIAS_GELU:  return launch_cfg<T, PGV_EPI_BIAS_GELU>(k, s, num_cu);
        case PGV_EPI_RESID:
        case PGV_EPI_BIAS_RESID: return launch_cfg<T, PGV_EPI_BIAS_RESID>(k, s, num_cu);
        case PGV_EPI_SWIGLU:     return launch_cfg<T, PGV_EPI_SWIGLU>(k, s, num_cu);
        case PGV_EPI_F32:        return launch_cfg<T, PGV_EPI_F32>(k, s, num_cu);
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
    if (k.bias == nullptr && (gemm_cfg_override() == 4 || gemm_cfg_override() == 6)) {      // persistent kernel: branch-free epilogue reads a real vector
        PGV_CHECK(a.N <= PGV_ZERO_BIAS_LEN, "gemm: N=%d exceeds the zero-bias vector", a.N);
        k.bias = ctx->zero_bias;
    }
    PGV_CHECK(!a.w_blocked || a.N % 16 == 0, "gemm: blocked weights need N %% 16 == 0");
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
