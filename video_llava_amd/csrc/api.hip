// libpgv core: error reporting, context, workspace arena, per-family device timers.
#include <stdarg.h>
#include <string.h>

#include "llm_internal.h"

static thread_local char g_err[1024] = "";

void pgv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int pgv_version(void) { return PGV_VERSION; }
extern "C" const char* pgv_last_error(void) { return g_err; }

extern "C" int pgv_ctx_create(int device, pgv_ctx** out) {
    PGV_CHECK(out != nullptr, "pgv_ctx_create: null out");
    int n = 0;
    PGV_HIP(hipGetDeviceCount(&n));
    PGV_CHECK(device >= 0 && device < n, "pgv_ctx_create: device %d out of range (%d visible)", device, n);
    PGV_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    PGV_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pgv_set_error("pgv_ctx_create: device %d is %s; libpgv is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return PGV_ESTATE;
    }
    pgv_ctx* c = new pgv_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    if (hipMalloc((void**)&c->zero_bias, PGV_ZERO_BIAS_LEN * sizeof(float)) != hipSuccess ||
        hipMemset(c->zero_bias, 0, PGV_ZERO_BIAS_LEN * sizeof(float)) != hipSuccess) {
        pgv_set_error("pgv_ctx_create: cannot allocate the zero-bias vector");
        delete c;
        return PGV_ENOMEM;
    }
    int rc_cfg = pgv_vit_attn_configure(c);
    if (int rc = rc_cfg != PGV_OK ? rc_cfg : pgv_gemv_configure(c); rc != PGV_OK) {      // per-device function attributes (vit_attn.hip, gemv.hip), outside any graph capture
        (void)hipFree(c->zero_bias);
        delete c;
        return rc;
    }
    *out = c;
    return PGV_OK;
}

extern "C" void pgv_ctx_destroy(pgv_ctx* ctx) {
    if (!ctx) return;
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->zero_bias) (void)hipFree(ctx->zero_bias);
    if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
    if (ctx->ws_event) (void)hipEventDestroy(ctx->ws_event);
    for (auto st : ctx->aux_stream) if (st) (void)hipStreamDestroy(st);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    for (auto ev : ctx->ev_join) if (ev) (void)hipEventDestroy(ev);
    for (auto& f : ctx->fam)
        for (auto e : f.ev) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" size_t pgv_ctx_workspace_bytes(const pgv_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

int pgv_ws_reserve(pgv_ctx* ctx, size_t bytes, hipStream_t s) {
    ctx->ws_off = 0;
    // Ordering against the previous user of the arena: it recorded ws_event on ITS stream when it finished enqueueing (pgv_ws_release), so a
    // call on another stream only waits on that event -- the old stream handle is never touched again (it may have been destroyed since).
    if (ctx->ws_stream_valid && ctx->ws_stream != s && ctx->ws_event) PGV_HIP(hipStreamWaitEvent(s, ctx->ws_event, 0));
    ctx->ws_stream = s; ctx->ws_stream_valid = true;
    if (bytes <= ctx->ws_bytes) return PGV_OK;
    // grow: wait for queued work that may still read the old arena
    PGV_HIP(hipDeviceSynchronize());
    if (ctx->ws) PGV_HIP(hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = pgv_align(bytes + (bytes >> 3), 1 << 20);
    hipError_t e = hipMalloc((void**)&ctx->ws, want);
    if (e != hipSuccess) {
        pgv_set_error("workspace hipMalloc(%zu MiB) failed: %s", want >> 20, hipGetErrorString(e));
        return PGV_ENOMEM;
    }
    ctx->ws_bytes = want;
    return PGV_OK;
}

int pgv_ws_release(pgv_ctx* ctx, hipStream_t s) {
    if (!ctx->ws_event) PGV_HIP(hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming));
    PGV_HIP(hipEventRecord(ctx->ws_event, s));
    return PGV_OK;
}

void* pgv_ws_alloc(pgv_ctx* ctx, size_t bytes) {
    size_t b = pgv_align(bytes);
    if (ctx->ws_off + b > ctx->ws_bytes) return nullptr;
    void* p = ctx->ws + ctx->ws_off;
    ctx->ws_off += b;
    return p;
}

// ---------------------------------------------------------------------------------------------
// profiling: hipEvent pairs recorded on the launch stream around every launch of a kernel family
// ---------------------------------------------------------------------------------------------
extern "C" int pgv_prof_enable(pgv_ctx* ctx, int on) {
    PGV_CHECK(ctx != nullptr, "pgv_prof_enable: null ctx");
    ctx->prof = on != 0;
    return PGV_OK;
}

extern "C" int pgv_prof_reset(pgv_ctx* ctx) {
    PGV_CHECK(ctx != nullptr, "pgv_prof_reset: null ctx");
    for (auto& f : ctx->fam) { f.used = 0; f.flops = 0; f.bytes = 0; f.launches = 0; }
    return PGV_OK;
}

void pgv_prof_begin(pgv_ctx* ctx, int family, hipStream_t s) {
    if (!ctx->prof) return;
    auto& f = ctx->fam[family];
    if (f.used + 2 > f.ev.size()) {
        size_t old = f.ev.size();
        f.ev.resize(old + 512);
        for (size_t i = old; i < f.ev.size(); ++i) (void)hipEventCreate(&f.ev[i]);
    }
    (void)hipEventRecord(f.ev[f.used], s);
}

void pgv_prof_end(pgv_ctx* ctx, int family, hipStream_t s, double flops, double bytes) {
    if (!ctx->prof) return;
    auto& f = ctx->fam[family];
    (void)hipEventRecord(f.ev[f.used + 1], s);
    f.used += 2;
    f.flops += flops;
    f.bytes += bytes;
    f.launches += 1;
}

extern "C" int pgv_prof_get(pgv_ctx* ctx, int family, int64_t* launches, double* ms, double* flops, double* bytes) {
    PGV_CHECK(ctx != nullptr && family >= 0 && family < PGV_NFAMILY, "pgv_prof_get: bad arguments");
    auto& f = ctx->fam[family];
    double total = 0;
    for (size_t i = 0; i + 1 < f.used; i += 2) {
        PGV_HIP(hipEventSynchronize(f.ev[i + 1]));
        float t = 0;
        PGV_HIP(hipEventElapsedTime(&t, f.ev[i], f.ev[i + 1]));
        total += t;
    }
    if (launches) *launches = f.launches;
    if (ms) *ms = total;
    if (flops) *flops = f.flops;
    if (bytes) *bytes = f.bytes;
    return PGV_OK;
}

// Mean elapsed time of an EMPTY hipEvent pair on `stream` (nothing enqueued between the two records): the fixed cost every
// per-launch measurement of pgv_prof_get carries.  bench.py subtracts it from the per-launch averages of short kernels.
extern "C" int pgv_prof_calibrate(pgv_ctx* ctx, void* stream, int n, double* ms_per_pair) {
    PGV_CHECK(ctx != nullptr && ms_per_pair != nullptr && n >= 1 && n <= 4096, "pgv_prof_calibrate: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(2 * (size_t)n);
    for (auto& e : ev) PGV_HIP(hipEventCreate(&e));
    for (int i = 0; i < n; ++i) { PGV_HIP(hipEventRecord(ev[2 * i], s)); PGV_HIP(hipEventRecord(ev[2 * i + 1], s)); }
    PGV_HIP(hipEventSynchronize(ev.back()));
    double tot = 0;
    for (int i = 0; i < n; ++i) { float t = 0; PGV_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1])); tot += t; }
    for (auto& e : ev) (void)hipEventDestroy(e);
    *ms_per_pair = tot / n;
    return PGV_OK;
}
