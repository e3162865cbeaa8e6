// LLaMA decoder, token pick on the device: greedy (argmax over the lm_head GEMV's per-workgroup candidates) and the reference's sampling mode
// (temperature, top-k 50, multinomial: video_chatgpt/inference.py:106-112 -> HF generation/logits_process.py:238-302,542-593), plus the
// per-sequence bookkeeping (position, step, token history, EOS stickiness) that keeps a token step's kernel arguments constant.
#include "llm_internal.h"

namespace {

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
