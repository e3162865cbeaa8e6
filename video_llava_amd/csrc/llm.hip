// TEMPORARY stubs (replaced by the real decoder in the next commit).
#include "pgv_common.h"
#define NOTYET(name) pgv_set_error(name ": not implemented yet"); return PGV_ESTATE
extern "C" int pgv_llm_create(pgv_ctx*, const pgv_llm_config*, int, pgv_llm**) { NOTYET("pgv_llm_create"); }
extern "C" void pgv_llm_destroy(pgv_llm*) {}
extern "C" int pgv_llm_load_tensor(pgv_llm*, const char*, const void*, int, int, void*) { NOTYET("pgv_llm_load_tensor"); }
extern "C" int pgv_llm_missing(const pgv_llm*) { return -1; }
extern "C" int pgv_kv_create(pgv_ctx*, pgv_llm*, int, int, pgv_kv**) { NOTYET("pgv_kv_create"); }
extern "C" void pgv_kv_destroy(pgv_kv*) {}
extern "C" int pgv_kv_len(const pgv_kv*, int) { return -1; }
extern "C" int pgv_llm_prefill(pgv_ctx*, pgv_llm*, pgv_kv*, const int32_t*, const int32_t*, int, const void*, int, const int32_t*, float*, int32_t*, void*) { NOTYET("pgv_llm_prefill"); }
extern "C" int pgv_llm_decode(pgv_ctx*, pgv_llm*, pgv_kv*, const int32_t*, float*, int32_t*, void*) { NOTYET("pgv_llm_decode"); }
extern "C" int pgv_llm_decode_greedy(pgv_ctx*, pgv_llm*, pgv_kv*, const int32_t*, int, int, int32_t*, void*) { NOTYET("pgv_llm_decode_greedy"); }

extern "C" int pgv_projector(pgv_ctx* ctx, int dtype, int depth, const void* const* d_weights, const float* const* d_biases, int mm_hidden,
                             int hidden, const void* d_x, int rows, void* d_y, void* stream) {
    PGV_CHECK(ctx && d_weights && d_biases && d_x && d_y, "pgv_projector: null argument");
    PGV_CHECK(depth >= 1 && depth <= 8, "pgv_projector: depth %d unsupported (identity has nothing to run)", depth);
    PGV_CHECK(rows > 0, "pgv_projector: rows must be positive");
    hipStream_t s = (hipStream_t)stream;
    void* tmp[2] = {nullptr, nullptr};
    if (depth > 1) {
        PGV_TRY(pgv_ws_reserve(ctx, 2 * pgv_align((size_t)rows * hidden * 2)));
        tmp[0] = pgv_ws_alloc(ctx, (size_t)rows * hidden * 2);
        tmp[1] = pgv_ws_alloc(ctx, (size_t)rows * hidden * 2);
    }
    const void* in = d_x;
    int K = mm_hidden;
    for (int i = 0; i < depth; ++i) {
        const bool last = (i == depth - 1);
        GemmArgs g{};
        g.A = in; g.lda = K; g.W = d_weights[i]; g.ldw = K; g.bias = d_biases[i];
        g.C = last ? d_y : tmp[i & 1]; g.ldc = hidden; g.M = rows; g.N = hidden; g.K = K;
        g.epi = last ? PGV_EPI_BIAS : PGV_EPI_BIAS_GELU;     // GELU sits between Linear i and Linear i+1 (builder.py:42-45)
        PGV_TRY(pgv_launch_gemm(ctx, dtype, g, s));
        in = g.C; K = hidden;
    }
    return PGV_OK;
}
