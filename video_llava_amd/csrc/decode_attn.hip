// LLaMA decoder, decode attention: RoPE + cache append + single-query attention over the cache (flash-decoding style), one workgroup per
// (sequence, head) or -- head counts that leave a ragged round -- per (sequence, head, context part) with a fence-free cross-workgroup merge.
// Replaces LlamaAttention.forward with a KV cache at q_len == 1 (HF:llama/modeling_llama.py:191-326).
#include <stdlib.h>

#include "llm_internal.h"

namespace {

constexpr int HD = kHD;

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

#ifndef PGV_LAB_DATTN_WAVES_PER_EU
#define PGV_LAB_DATTN_WAVES_PER_EU 6      // lab A/B: -DPGV_LAB_DATTN_WAVES_PER_EU=5 restores two workgroups per CU
#endif
template <typename T, int SPLIT>
__global__ __launch_bounds__(512, PGV_LAB_DATTN_WAVES_PER_EU) void decode_attn_split_kernel(const typename T::elem* __restrict__ qkv, const int* __restrict__ pos_arr,
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
    const bool fresh = (sp == 0 && w == 0);
    for (int r0 = 0; 16 * (SPLIT * 4 * r0 + sp) < n_keys; r0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int u = 0; u < 2; ++u) update(u, kq[d][u], vq[d][u], key_of(r0 + d, u) < n_keys);
#pragma unroll
            for (int u = 0; u < 2; ++u) load_row(max(0, min(key_of(r0 + d + DEPTH, u), n_keys - 1)), kq[d][u], vq[d][u]);     // clamped rows are masked
        }
    }
    // fresh token: split 0, wave 0 appends the rotated k and v to the cache and feeds them to the softmax from registers -- AFTER the key loop
    // (round 5), when the ring of prefetched keys is dead: rotating k up front kept 96 VGPRs live, 5 waves per SIMD = two workgroups per CU, so
    // 13B's 640 workgroups ran as 512 + 128; with the block here the kernel fits 80 VGPRs = three workgroups per CU, all 640 resident at once.
    // Same values in the same order (the fresh key always entered the softmax last): bitwise the former result.
    if (fresh) {
        typename T::v8 knew, vnew;
        float cs_c[8], cs_s[8], kr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float2 cs = rope[(size_t)pos * 64 + j0 + e]; cs_c[e] = cs.x; cs_s[e] = dc < 8 ? -cs.y : cs.y; }
        const typename T::v8 k_own = *(const typename T::v8*)(k + dc * 8), k_oth = *(const typename T::v8*)(k + (dc ^ 8) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) kr[e] = (float)T::from_f32((float)k_own[e] * cs_c[e] + (float)k_oth[e] * cs_s[e]);
#pragma unroll
        for (int e = 0; e < 8; ++e) knew[e] = T::from_f32(kr[e]);
        vnew = *(const typename T::v8*)(v + dc * 8);
        if (slot == 0) {
            *(typename T::v8*)(kcache + (size_t)pos * HD + dc * 8) = knew;
            *(typename T::v8*)(vcache + (size_t)pos * HD + dc * 8) = vnew;
        }
        update(0, knew, vnew, slot == 0);
    }
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

}  // namespace

// Workgroups per (sequence, head) unit.  The split is a function of the MODEL (its head count), never of the batch: a sequence's attention is
// then computed with the same partition and the same merge order whether it is decoded alone or next to 15 others -- results stay bitwise
// batch-invariant (tests/test_gpu_llm.py).  Head counts whose 8-sequence launch fills the chip's CUs in whole rounds (7B: 32 heads x 8 = 256)
// keep the unsplit kernel; the others (13B: 40 heads -> 320 units = 1 1/4 rounds) are cut in 2 (640 workgroups, all resident at once).
// Measured (gpurun_out/r4d, 13B fp8, 8 sequences, us per layer): unsplit 18.7, 2 parts 18.1, 4 parts 22.6, 8 parts 31.4 -- 7B (256 units):
// 12.8 / 14.2 / 18.8 / 26.4.  Every extra hand-off through the memory side (write-through stores -> acknowledged -> ticket -> loads) adds
// ~4 us to a workgroup's life, which more resident workgroups only partly hide: a cut in 2 is the only one that pays, and only where the
// unsplit launch leaves a ragged round.  PGV_DATTN_SPLIT=1/2/4/8 forces a value (A/B and the parity tests of every variant; bitwise
// invariance then holds only among runs with the same setting).  PGV_DATTN_SPLIT=1 is also the SAFE FALLBACK: the split kernel's cross-workgroup
// merge relies on how gfx950 lowers relaxed agent-scope atomics (sc1 write-through stores / sc1 loads -- checked against the ISA at build time,
// video_llava_amd/build.py check_isa) rather than on the HIP memory model; the unsplit kernel has no cross-workgroup traffic at all.
static int decode_attn_split(int heads, int num_cu) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PGV_DATTN_SPLIT"); forced = e ? atoi(e) : 0; }
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
    return (heads * 8) % num_cu == 0 ? 1 : 2;
}

int pgv_launch_decode_attn(pgv_ctx* ctx, int dtype, const void* qkv, const int* pos, const void* rope, void* Kc, void* Vc, void* out, int B, int H,
                           int heads, int max_seq, double bytes, hipStream_t s, float* part, unsigned* ticket) {
    const float sc = 0.08838834764831845f * 1.4426950408889634f;
    int nw = 8;                                          // 16-wave workgroups: a lab A/B switch (-DPGV_LAB), never the release default
#ifdef PGV_LAB
    { const char* e = getenv("PGV_DATTN_WAVES"); if (e && atoi(e) == 16) nw = 16; }
#endif
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
