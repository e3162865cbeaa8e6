// Lab: what would ONE persistent launch per token step buy over a hipGraph chain of kernels?  Not product code.
// The same streaming phases as tail_prefetch.hip (256 workgroups x 8 waves, a two-buffer pipeline of 4 x 1 KiB non-temporal wave-loads over the
// workgroup's share of a matrix, LDS reduce, 1 KiB store), 64 phases over 16 HBM-cold matrices.  Every phase first reads 1 KiB that ANOTHER workgroup
// wrote in the previous phase (the activation dependency of a decoder: no phase can start before the previous one is complete everywhere).
//   chain      : one kernel per phase in a hipGraph (the product's structure)
//   persistent : one launch, a grid barrier (agent-scope atomic counter) between phases
//   persistent + prefetch : the first two weight buffers of the NEXT phase are requested before the barrier (they do not depend on the activations)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/persistent_chain scripts/lab/persistent_chain.hip && /tmp/persistent_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
constexpr int NMAT = 16;
struct Mats { const char* m[NMAT]; };

struct Phase {
    const char* p; int nb, w, lane;
    __device__ void load(u4 (&v)[4], int i0) const {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load((const u4*)(p + (size_t)(w + 8 * min(i0 + u, nb - 1)) * 1024));
    }
};

// the body of one phase; a / b may arrive already requested (PRE)
template <bool PRE>
__device__ inline u4 stream_phase(const Phase& ph, u4 (&a)[4], u4 (&b)[4], u4 acc) {
    auto use = [&](u4 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
    };
    if (!PRE) { ph.load(a, 0); }
    int i = 0;
    if (PRE) {                                            // b = blocks 4..7 is in flight as well
        __builtin_amdgcn_sched_barrier(0);
        use(a);
        __builtin_amdgcn_sched_barrier(0);
        ph.load(a, 8);
        __builtin_amdgcn_sched_barrier(0);
        use(b);
        __builtin_amdgcn_sched_barrier(0);
        i = 8;
    }
    for (; i + 8 < ph.nb; i += 8) {
        ph.load(b, i + 4);
        __builtin_amdgcn_sched_barrier(0);
        use(a);
        __builtin_amdgcn_sched_barrier(0);
        ph.load(a, i + 8);
        __builtin_amdgcn_sched_barrier(0);
        use(b);
        __builtin_amdgcn_sched_barrier(0);
    }
    ph.load(b, i + 4);
    use(a);
    use(b);
    return acc;
}

template <bool FF = false>
__device__ inline void epilogue(u4 (*red)[64], u4 acc, int w, int lane, float* out) {
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0) {
        u4 t = red[0][lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) t ^= red[k][lane];
        const float v = (float)(t[0] ^ t[1] ^ t[2] ^ t[3]);
        if (FF) __hip_atomic_store(out + (size_t)blockIdx.x * 64 + lane, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1) store, no fence
        else out[(size_t)blockIdx.x * 64 + lane] = v;
    }
}

__global__ __launch_bounds__(512) void chain_kernel(const char* cur, size_t bytes, const float* prev, float* out) {
    __shared__ u4 red[8][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_wg = bytes / gridDim.x;
    const int nblk = (int)(per_wg / 1024);
    Phase ph{cur + (size_t)blockIdx.x * per_wg + lane * 16, (nblk - w + 7) / 8, w, lane};
    u4 a[4], b[4];
    u4 acc = {0, 0, 0, 0};
    ph.load(a, 0);                                           // the weights do not wait for the activations
    acc[0] = __float_as_uint(prev[(size_t)((blockIdx.x + 1) % gridDim.x) * 64 + lane]);
    u4 a2[4] = {a[0], a[1], a[2], a[3]};
    // (PRE = false would re-request a; keep the request above and enter the loop by hand)
    auto use = [&](u4 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u];
    };
    int i = 0;
    for (; i + 8 < ph.nb; i += 8) {
        ph.load(b, i + 4);
        __builtin_amdgcn_sched_barrier(0);
        use(a2);
        __builtin_amdgcn_sched_barrier(0);
        ph.load(a2, i + 8);
        __builtin_amdgcn_sched_barrier(0);
        use(b);
        __builtin_amdgcn_sched_barrier(0);
    }
    ph.load(b, i + 4);
    use(a2);
    use(b);
    epilogue(red, acc, w, lane, out);
}

template <int PF, bool FF>
__global__ __launch_bounds__(512) void persistent_kernel(Mats mats, int nphase, size_t bytes, float* buf0, float* buf1, unsigned* counter) {
    __shared__ u4 red[8][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t per_wg = bytes / gridDim.x;
    const int nblk = (int)(per_wg / 1024);
    const size_t off = (size_t)blockIdx.x * per_wg + lane * 16;
    u4 a[4], b[4];
    if (PF) {
        Phase p0{mats.m[0] + off, (nblk - w + 7) / 8, w, lane};
        p0.load(a, 0); p0.load(b, 4);
    }
    for (int i = 0; i < nphase; ++i) {
        const float* prev = (i & 1) ? buf0 : buf1;
        float* out = (i & 1) ? buf1 : buf0;
        Phase ph{mats.m[i % NMAT] + off, (nblk - w + 7) / 8, w, lane};
        u4 acc = {0, 0, 0, 0};
        // written by another workgroup in the previous phase
        if (FF) acc[0] = __float_as_uint(__hip_atomic_load(prev + (size_t)((blockIdx.x + 1) % gridDim.x) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else acc[0] = __float_as_uint(__builtin_nontemporal_load(prev + (size_t)((blockIdx.x + 1) % gridDim.x) * 64 + lane));
        acc = PF ? stream_phase<true>(ph, a, b, acc) : stream_phase<false>(ph, a, b, acc);
        const bool more = PF && i + 1 < nphase;
        Phase pn{mats.m[(i + 1) % NMAT] + off, ph.nb, w, lane};
        if (more && (!FF || w != 0)) { pn.load(a, 0); pn.load(b, 4); }     // the next phase's first two buffers go out before the barrier
        epilogue<FF>(red, acc, w, lane, out);
        // grid barrier: every workgroup's stores of this phase are visible before anyone starts the next one
        const unsigned want = (unsigned)(i + 1) * gridDim.x;
        if (FF) {
            // fence-free (the protocol of decode_attn_split_kernel): the results were stored write-through by wave 0; once they are acknowledged
            // (vmcnt(0) -- which is why wave 0 requests its prefetch only afterwards) a relaxed ticket suffices, and the readers use sc1 loads
            if (w == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
                }
                if (more) { pn.load(a, 0); pn.load(b, 4); }
            }
        } else {
            __syncthreads();
            if (threadIdx.x == 0) {
                __threadfence();
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
}

int main() {
    for (size_t mb : {34, 62, 100, 180}) {
        const size_t bytes = (mb << 20) / (256 * 8192) * (256 * 8192);
        std::vector<char*> mats(NMAT);
        Mats mm;
        for (int i = 0; i < NMAT; ++i) { CK(hipMalloc(&mats[i], bytes)); CK(hipMemset(mats[i], 1, bytes)); mm.m[i] = mats[i]; }
        float *buf0, *buf1; unsigned* counter;
        CK(hipMalloc(&buf0, 256 * 64 * 4)); CK(hipMalloc(&buf1, 256 * 64 * 4)); CK(hipMalloc(&counter, 4));
        CK(hipMemset(buf0, 0, 256 * 64 * 4)); CK(hipMemset(buf1, 0, 256 * 64 * 4));
        hipStream_t s; CK(hipStreamCreate(&s));
        const int n = 64;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("== %zu MB per phase ==\n", mb);
        {   // chain
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int i = 0; i < n; ++i)
                hipLaunchKernelGGL(chain_kernel, dim3(256), dim3(512), 0, s, mats[i % NMAT], bytes, (i & 1) ? buf0 : buf1, (i & 1) ? buf1 : buf0);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            float sum = 0;
            for (int r = 0; r < 6; ++r) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0) sum += ms;
            }
            printf("chain of kernels (hipGraph)                   %6.2f us per phase   %.2f TB/s\n", sum / 5 * 1e3 / n, bytes / (sum / 5 * 1e-3 / n) / 1e12);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        for (int v = 0; v < 4; ++v) {
            float sum = 0;
            for (int r = 0; r < 6; ++r) {
                CK(hipMemsetAsync(counter, 0, 4, s));
                CK(hipEventRecord(e0, s));
                if (v == 0) hipLaunchKernelGGL((persistent_kernel<0, false>), dim3(256), dim3(512), 0, s, mm, n, bytes, buf0, buf1, counter);
                if (v == 1) hipLaunchKernelGGL((persistent_kernel<1, false>), dim3(256), dim3(512), 0, s, mm, n, bytes, buf0, buf1, counter);
                if (v == 2) hipLaunchKernelGGL((persistent_kernel<0, true>), dim3(256), dim3(512), 0, s, mm, n, bytes, buf0, buf1, counter);
                if (v == 3) hipLaunchKernelGGL((persistent_kernel<1, true>), dim3(256), dim3(512), 0, s, mm, n, bytes, buf0, buf1, counter);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 0) sum += ms;
            }
            const char* names[4] = {"persistent, fenced grid barrier             ", "persistent, fenced grid barrier + prefetch  ", "persistent, fence-free barrier              ", "persistent, fence-free barrier + prefetch   "};
            printf("%s %6.2f us per phase   %.2f TB/s\n", names[v], sum / 5 * 1e3 / n, bytes / (sum / 5 * 1e-3 / n) / 1e12);
        }
        for (auto m : mats) CK(hipFree(m));
        CK(hipFree(buf0)); CK(hipFree(buf1)); CK(hipFree(counter)); CK(hipStreamDestroy(s));
    }
    return 0;
}
