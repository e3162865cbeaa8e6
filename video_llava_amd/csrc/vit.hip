// CLIP vision tower: packed weights + forward driver.
// Replaces `vision_tower(image_tensor, output_hidden_states=True).hidden_states[k]`
// (video_chatgpt/inference.py:93-94; chat.py:140-143; scripts/save_spatio_temporal_clip_features.py:116-120),
// i.e. HF CLIPVisionTransformer (HF:clip/modeling_clip.py:594-658) up to encoder layer k.
#include <stdlib.h>
#include <string.h>

#include "pgv_common.h"
#include "weights.h"

int pgv_launch_layernorm(int dtype, const float* x, const float* g, const float* b, float eps, void* y, int rows, int cols, hipStream_t s);
int pgv_launch_embed_ln(int dtype, const float* pe, const float* cls, const float* pos, const float* g, const float* b, float eps, float* out,
                        int rows, int tokens, int cols, const float* gnext, void* x16, float* rowstat, float* rowmean, hipStream_t s);
int pgv_launch_ln_stats(const float* part, float* rowstat, float* rowmean, const float* cshift, int rows, int np, int cols, float eps, hipStream_t s);
int pgv_launch_vec_mean(const float* x, int n, float* out, hipStream_t s);
int pgv_launch_ln_fold(int dtype, const void* W, const float* bias, const float* gamma, const float* beta, float* colsum, float* bias2, int N, int K, hipStream_t s);
int pgv_launch_cast(int dtype, const float* x, void* y, size_t n, hipStream_t s);
int pgv_launch_im2col(const void* pix, void* a0, int T, int S, int g, int p, int Kp, hipStream_t s);
int pgv_launch_vit_attn(pgv_ctx* ctx, int dtype, const void* qkv, int ld, void* out, int ldo, int T, int N, int C, int heads, hipStream_t s);

struct VitLayer {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    void *wqkv, *wo, *w1, *w2;
    float *bqkv, *bo, *b1, *b2;
    // folded LayerNorm (gemm.hip EPI_LN_*), derived from the loaded tensors before the first forward: column sums sum_k gamma_k W[n,k]
    // and biases b + W beta of the two GEMMs that consume a LayerNorm (qkv after layer_norm1, fc1 after layer_norm2)
    float *s_qkv, *b2_qkv, *s_fc1, *b2_fc1;
    float *m_bo, *m_b2;            // device scalars: mean of the out_proj / fc2 bias (centre shift of the folded-LayerNorm producers)
};

struct pgv_vit {
    pgv_ctx* ctx;
    pgv_vit_config cfg;
    int dtype;
    int grid, patches, tokens, Kp;
    char* blob = nullptr;
    size_t blob_bytes = 0;
    void* patch_w;
    float *cls, *pos, *pre_g, *pre_b;
    float* ones;                   // [hidden] of 1.0f: the "next gamma" of the last executed layer, whose 16-bit operand copy IS the hidden state returned
    std::vector<VitLayer> layers;
    std::set<std::string> loaded;
    int expected = 0;
    bool folded = false;           // the derived vectors above are current (reset by every tensor load)
    int max_chunk_frames = 1024;   // frames per pass (workspace 5.8 GB at 224 px): one pass for the 800-frame bench batch -> 3.75 % fewer GEMM tile rounds than 2 x 400
};

extern "C" int pgv_vit_create(pgv_ctx* ctx, const pgv_vit_config* cfg, int dtype, pgv_vit** out) {
    PGV_CHECK(ctx && cfg && out, "pgv_vit_create: null argument");
    PGV_CHECK(dtype == PGV_F16 || dtype == PGV_BF16, "pgv_vit_create: dtype must be PGV_F16 or PGV_BF16");
    PGV_CHECK(cfg->hidden == 1024, "pgv_vit_create: CLIP width must be 1024 (video_chatgpt/model/video_chatgpt.py:106 hard-codes it); got %d", cfg->hidden);
    PGV_CHECK(cfg->heads * 64 == cfg->hidden, "pgv_vit_create: head_dim must be 64");
    PGV_CHECK(cfg->inter % 256 == 0 && cfg->inter > 0, "pgv_vit_create: intermediate size must be a multiple of 256");
    PGV_CHECK(cfg->patch > 0 && cfg->patch % 2 == 0 && cfg->image % cfg->patch == 0, "pgv_vit_create: image %d / patch %d unsupported", cfg->image, cfg->patch);
    PGV_CHECK(cfg->layers > 0, "pgv_vit_create: layers must be positive");
    pgv_vit* v = new pgv_vit();
    v->ctx = ctx; v->cfg = *cfg; v->dtype = dtype;
    v->grid = cfg->image / cfg->patch; v->patches = v->grid * v->grid; v->tokens = v->patches + 1;
    v->Kp = (3 * cfg->patch * cfg->patch + 63) / 64 * 64;
    const size_t C = cfg->hidden, I = cfg->inter, L = cfg->layers;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += pgv_align(bytes); return o; };
    const size_t o_patch = take(C * v->Kp * 2), o_cls = take(C * 4), o_pos = take((size_t)v->tokens * C * 4), o_pg = take(C * 4), o_pb = take(C * 4), o_ones = take(C * 4);
    struct LO { size_t ln1g, ln1b, ln2g, ln2b, wqkv, wo, w1, w2, bqkv, bo, b1, b2, sqkv, b2qkv, sfc1, b2fc1, mbo, mb2; };
    std::vector<LO> lo(L);
    for (size_t i = 0; i < L; ++i) {
        lo[i].ln1g = take(C * 4); lo[i].ln1b = take(C * 4); lo[i].ln2g = take(C * 4); lo[i].ln2b = take(C * 4);
        lo[i].wqkv = take(3 * C * C * 2); lo[i].wo = take(C * C * 2); lo[i].w1 = take(I * C * 2); lo[i].w2 = take(C * I * 2);
        lo[i].bqkv = take(3 * C * 4); lo[i].bo = take(C * 4); lo[i].b1 = take(I * 4); lo[i].b2 = take(C * 4);
        lo[i].sqkv = take(3 * C * 4); lo[i].b2qkv = take(3 * C * 4); lo[i].sfc1 = take(I * 4); lo[i].b2fc1 = take(I * 4);
        lo[i].mbo = take(4); lo[i].mb2 = take(4);
    }
    hipError_t e = hipMalloc((void**)&v->blob, off);
    if (e != hipSuccess) { delete v; pgv_set_error("pgv_vit_create: hipMalloc(%zu MiB): %s", off >> 20, hipGetErrorString(e)); return PGV_ENOMEM; }
    v->blob_bytes = off;
    char* b = v->blob;
    v->patch_w = b + o_patch; v->cls = (float*)(b + o_cls); v->pos = (float*)(b + o_pos); v->pre_g = (float*)(b + o_pg); v->pre_b = (float*)(b + o_pb); v->ones = (float*)(b + o_ones);
    {
        std::vector<float> one(C, 1.0f);
        e = hipMemcpy(v->ones, one.data(), C * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(v->blob); delete v; pgv_set_error("pgv_vit_create: upload: %s", hipGetErrorString(e)); return PGV_EHIP; }
    }
    v->layers.resize(L);
    for (size_t i = 0; i < L; ++i) {
        VitLayer& l = v->layers[i];
        l.ln1_g = (float*)(b + lo[i].ln1g); l.ln1_b = (float*)(b + lo[i].ln1b); l.ln2_g = (float*)(b + lo[i].ln2g); l.ln2_b = (float*)(b + lo[i].ln2b);
        l.wqkv = b + lo[i].wqkv; l.wo = b + lo[i].wo; l.w1 = b + lo[i].w1; l.w2 = b + lo[i].w2;
        l.bqkv = (float*)(b + lo[i].bqkv); l.bo = (float*)(b + lo[i].bo); l.b1 = (float*)(b + lo[i].b1); l.b2 = (float*)(b + lo[i].b2);
        l.s_qkv = (float*)(b + lo[i].sqkv); l.b2_qkv = (float*)(b + lo[i].b2qkv); l.s_fc1 = (float*)(b + lo[i].sfc1); l.b2_fc1 = (float*)(b + lo[i].b2fc1);
        l.m_bo = (float*)(b + lo[i].mbo); l.m_b2 = (float*)(b + lo[i].mb2);
    }
    v->expected = 5 + 16 * (int)L;   // post_layernorm is accepted but not needed for hidden_states
    *out = v;
    return PGV_OK;
}

extern "C" void pgv_vit_destroy(pgv_vit* vit) {
    if (!vit) return;
    if (vit->blob) (void)hipFree(vit->blob);
    delete vit;
}

extern "C" int pgv_vit_missing(const pgv_vit* vit) { return vit ? vit->expected - (int)vit->loaded.size() : -1; }

extern "C" int pgv_vit_load_tensor(pgv_vit* v, const char* name_in, const void* data, int src_dtype, int on_device, int64_t numel, void* stream) {
    PGV_CHECK(v && name_in && data, "pgv_vit_load_tensor: null argument");
    std::string name(name_in);
    if (name.rfind("vision_model.", 0) == 0) name = name.substr(13);
    const long long C = v->cfg.hidden, I = v->cfg.inter;
    PackDst d;
    bool counted = true;
    if (name == "embeddings.class_embedding") { d.ptr = v->cls; d.rows = 1; d.cols = C; }
    else if (name == "embeddings.patch_embedding.weight") { d.ptr = v->patch_w; d.dst_dtype = v->dtype; d.rows = C; d.cols = 3 * v->cfg.patch * v->cfg.patch; d.dst_stride = v->Kp; }
    else if (name == "embeddings.position_embedding.weight") { d.ptr = v->pos; d.rows = v->tokens; d.cols = C; }
    else if (name == "embeddings.position_ids") return PGV_OK;
    else if (name == "pre_layrnorm.weight") { d.ptr = v->pre_g; d.rows = 1; d.cols = C; }
    else if (name == "pre_layrnorm.bias") { d.ptr = v->pre_b; d.rows = 1; d.cols = C; }
    else if (name == "post_layernorm.weight" || name == "post_layernorm.bias") return PGV_OK;
    else if (name.rfind("encoder.layers.", 0) == 0) {
        const char* p = name.c_str() + 15;
        char* end = nullptr;
        long li = strtol(p, &end, 10);
        if (end == p || *end != '.' || li < 0 || li >= v->cfg.layers) { pgv_set_error("pgv_vit_load_tensor: bad layer index in '%s'", name_in); return PGV_ENAME; }
        std::string rest(end + 1);
        VitLayer& l = v->layers[li];
        auto vec = [&](float* ptr, long long n) { d.ptr = ptr; d.rows = 1; d.cols = n; };
        auto mat = [&](void* ptr, long long r, long long c, long long roff) { d.ptr = ptr; d.dst_dtype = v->dtype; d.rows = r; d.cols = c; d.row_off = roff; };
        if (rest == "self_attn.q_proj.weight") mat(l.wqkv, C, C, 0);
        else if (rest == "self_attn.k_proj.weight") mat(l.wqkv, C, C, C);
        else if (rest == "self_attn.v_proj.weight") mat(l.wqkv, C, C, 2 * C);
        else if (rest == "self_attn.q_proj.bias") vec(l.bqkv, C);
        else if (rest == "self_attn.k_proj.bias") vec(l.bqkv + C, C);
        else if (rest == "self_attn.v_proj.bias") vec(l.bqkv + 2 * C, C);
        else if (rest == "self_attn.out_proj.weight") mat(l.wo, C, C, 0);
        else if (rest == "self_attn.out_proj.bias") vec(l.bo, C);
        else if (rest == "layer_norm1.weight") vec(l.ln1_g, C);
        else if (rest == "layer_norm1.bias") vec(l.ln1_b, C);
        else if (rest == "layer_norm2.weight") vec(l.ln2_g, C);
        else if (rest == "layer_norm2.bias") vec(l.ln2_b, C);
        else if (rest == "mlp.fc1.weight") mat(l.w1, I, C, 0);
        else if (rest == "mlp.fc1.bias") vec(l.b1, I);
        else if (rest == "mlp.fc2.weight") mat(l.w2, C, I, 0);
        else if (rest == "mlp.fc2.bias") vec(l.b2, C);
        else { pgv_set_error("pgv_vit_load_tensor: unexpected key '%s'", name_in); return PGV_ENAME; }
    } else { pgv_set_error("pgv_vit_load_tensor: unexpected key '%s'", name_in); return PGV_ENAME; }
    if (d.dst_stride == 0) d.dst_stride = d.cols;
    PGV_CHECK(numel == d.rows * d.cols, "pgv_vit_load_tensor: size mismatch for '%s': %lld elements given, the model expects %lld x %lld", name_in,
              (long long)numel, d.rows, d.cols);
    PGV_TRY(pgv_pack_tensor(d, data, src_dtype, on_device, (hipStream_t)stream));
    if (counted) v->loaded.insert(name);
    v->folded = false;
    return PGV_OK;
}

static bool ln_fold_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PGV_VIT_LN_FOLD"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}

constexpr int MAX_LANES = 4;
static int vit_lanes() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PGV_VIT_LANES"); v = e ? atoi(e) : 2; if (v < 1) v = 1; if (v > MAX_LANES) v = MAX_LANES; }   // A/B switch: results are identical for any value
    return v;
}

extern "C" int pgv_vit_forward(pgv_ctx* ctx, pgv_vit* v, const void* d_pixels, int T, int n_layers, void* d_hidden, void* stream) {
    PGV_CHECK(ctx && v && d_pixels && d_hidden, "pgv_vit_forward: null argument");
    PGV_CHECK(T > 0, "pgv_vit_forward: T must be positive (got %d)", T);
    PGV_CHECK(n_layers >= 0 && n_layers <= v->cfg.layers, "pgv_vit_forward: n_layers %d outside [0, %d]", n_layers, v->cfg.layers);
    if (pgv_vit_missing(v) != 0) { pgv_set_error("pgv_vit_forward: %d weight tensors not loaded", pgv_vit_missing(v)); return PGV_ESTATE; }
    hipStream_t s = (hipStream_t)stream;
    const int C = v->cfg.hidden, I = v->cfg.inter, N = v->tokens, P = v->patches, S = v->cfg.image;
    const bool fold = ln_fold_enabled();
    if (fold && !v->folded) {          // derived vectors of the folded LayerNorm, once per (re)load
        for (auto& l : v->layers) {
            PGV_TRY(pgv_launch_ln_fold(v->dtype, l.wqkv, l.bqkv, l.ln1_g, l.ln1_b, l.s_qkv, l.b2_qkv, 3 * C, C, s));
            PGV_TRY(pgv_launch_ln_fold(v->dtype, l.w1, l.b1, l.ln2_g, l.ln2_b, l.s_fc1, l.b2_fc1, I, C, s));
            PGV_TRY(pgv_launch_vec_mean(l.bo, C, l.m_bo, s));
            PGV_TRY(pgv_launch_vec_mean(l.b2, C, l.m_b2, s));
        }
        v->folded = true;
    }
    // ---- lanes ----------------------------------------------------------------------------------------------------------------------
    // A pass is a chain of ~160 dependent launches, each a persistent grid whose last round fills only part of the chip (N = 1024 GEMMs at
    // 800 frames: 3216 tiles = 12.56 rounds of 256 CUs; one 100-frame clip: 404 tiles = 1.58 rounds).  With several lanes (two by default) -- the frames split in
    // equal parts, each part its own chain on its own stream and its own workspace -- the idle CUs of one lane's tail run the head of the other
    // lane's next kernel.  Per-frame results do not depend on how frames are batched (bitwise; tests/test_gpu_vision.py), so the split is
    // invisible in the output.  The second stream is forked from and joined back into the caller's stream with events.
    const int lanes = (!ctx->prof && T >= 8 * vit_lanes()) ? vit_lanes() : 1;     // per-launch profiling wants kernels alone on the chip
    int lane_t0[MAX_LANES] = {0}, lane_T[MAX_LANES] = {T};
    for (int k = 0, at = 0; k < lanes; ++k) { lane_t0[k] = at; lane_T[k] = (T - at + (lanes - k) - 1) / (lanes - k); at += lane_T[k]; }
    int chunk_frames = v->max_chunk_frames;
#ifdef PGV_LAB
    { static int c = -1; if (c < 0) { const char* e = getenv("PGV_VIT_CHUNK"); c = e ? atoi(e) : 0; } if (c > 0) chunk_frames = c; }   // lab: frames per lane and pass
#endif
    // the arena holds `lanes` workspaces: a lane's pass is capped at chunk_frames / lanes so the total stays what one lane of chunk_frames took
    // (with T > max_chunk_frames the arena would otherwise double next to a 13B model + KV cache)
    const int lane_cap = chunk_frames / lanes > 0 ? chunk_frames / lanes : 1;
    const int Tc_max = lane_T[0] < lane_cap ? lane_T[0] : lane_cap;
    const size_t Mmax = (size_t)Tc_max * N;
    const int NP = C / 64;             // 64-column pieces of a residual row (partial statistics of the folded LayerNorm)
    const size_t b_resid = pgv_align(Mmax * C * 4), b_xn = pgv_align(Mmax * C * 2), b_qkv = pgv_align(Mmax * 3 * C * 2), b_ao = pgv_align(Mmax * C * 2),
                 b_h = pgv_align(Mmax * I * 2 > (size_t)Tc_max * P * ((size_t)v->Kp * 2 + C * 4) + 512 ? Mmax * I * 2 : (size_t)Tc_max * P * ((size_t)v->Kp * 2 + C * 4) + 512),
                 b_part = pgv_align(Mmax * NP * 8), b_stat = pgv_align(Mmax * 8), b_mean = pgv_align(Mmax * 4);
    PGV_TRY(pgv_ws_reserve(ctx, (size_t)lanes * (b_resid + b_xn + b_qkv + b_ao + b_h + b_part + b_stat + b_mean), s));
    struct Lane { hipStream_t s; float* resid; char *xn, *qkv, *ao, *hbuf; float *part, *rowstat, *rowmean; int t0, T, done; } L[MAX_LANES];
    for (int k = 0; k < lanes; ++k) {
        L[k].resid = (float*)pgv_ws_alloc(ctx, b_resid);
        L[k].xn = (char*)pgv_ws_alloc(ctx, b_xn);
        L[k].qkv = (char*)pgv_ws_alloc(ctx, b_qkv);
        L[k].ao = (char*)pgv_ws_alloc(ctx, b_ao);
        L[k].hbuf = (char*)pgv_ws_alloc(ctx, b_h);
        L[k].part = (float*)pgv_ws_alloc(ctx, b_part);
        L[k].rowstat = (float*)pgv_ws_alloc(ctx, b_stat);
        L[k].rowmean = (float*)pgv_ws_alloc(ctx, b_mean);
        PGV_CHECK(L[k].resid && L[k].xn && L[k].qkv && L[k].ao && L[k].hbuf && L[k].part && L[k].rowstat && L[k].rowmean, "pgv_vit_forward: workspace exhausted");
        L[k].s = s; L[k].t0 = lane_t0[k]; L[k].T = lane_T[k]; L[k].done = 0;
    }
    if (lanes > 1) {
        if (!ctx->ev_fork) PGV_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        PGV_HIP(hipEventRecord(ctx->ev_fork, s));
        for (int k = 1; k < lanes; ++k) {
            if (!ctx->aux_stream[k - 1]) PGV_HIP(hipStreamCreateWithFlags(&ctx->aux_stream[k - 1], hipStreamNonBlocking));
            if (!ctx->ev_join[k - 1]) PGV_HIP(hipEventCreateWithFlags(&ctx->ev_join[k - 1], hipEventDisableTiming));
            PGV_HIP(hipStreamWaitEvent(ctx->aux_stream[k - 1], ctx->ev_fork, 0));
            L[k].s = ctx->aux_stream[k - 1];
        }
    }

    // one chunk (<= Tc_max frames) of a lane, phase by phase so that the two lanes' launches are enqueued alternately
    auto embed = [&](Lane& ln, int t0, int Tc) -> int {
        const int M = Tc * N;
        const char* pix = (const char*)d_pixels + (size_t)t0 * 3 * S * S * 2;
        // patch conv as GEMM: [Tc*P, Kp] x [C, Kp]^T -> fp32 [Tc*P, C]
        char* a0 = ln.hbuf;
        float* pe = (float*)(ln.hbuf + pgv_align((size_t)Tc * P * v->Kp * 2));
        PGV_TRY(pgv_launch_im2col(pix, a0, Tc, S, v->grid, v->cfg.patch, v->Kp, ln.s));
        GemmArgs g{};
        g.A = a0; g.lda = v->Kp; g.W = v->patch_w; g.ldw = v->Kp; g.bias = nullptr; g.C = pe; g.ldc = C; g.M = Tc * P; g.N = C; g.K = v->Kp; g.epi = PGV_EPI_F32;
        PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, ln.s));
        const bool f0 = fold && n_layers > 0;
        PGV_TRY(pgv_launch_embed_ln(v->dtype, pe, v->cls, v->pos, v->pre_g, v->pre_b, v->cfg.eps, ln.resid, M, N, C, f0 ? v->layers[0].ln1_g : nullptr, ln.xn, ln.rowstat, ln.rowmean, ln.s));
        return PGV_OK;
    };
    auto layer = [&](Lane& ln, int t0, int Tc, int li) -> int {
        const int M = Tc * N;
        const VitLayer& l = v->layers[li];
        float* resid = ln.resid; char *xn = ln.xn, *qkv = ln.qkv, *ao = ln.ao, *hbuf = ln.hbuf; float *part = ln.part, *rowstat = ln.rowstat, *rowmean = ln.rowmean;
        hipStream_t s = ln.s;
        GemmArgs g{};
        if (fold) {
            // LayerNorm has no launch of its own (gemm.hip, EPI_LN_*): xn = round16(resid * ln1_g), rowstat = (mean, rstd) of resid
            g = GemmArgs{}; g.A = xn; g.lda = C; g.W = l.wqkv; g.ldw = C; g.bias = l.b2_qkv; g.C = qkv; g.ldc = 3 * C; g.M = M; g.N = 3 * C; g.K = C;
            g.epi = PGV_EPI_LN_BIAS; g.rowstat = rowstat; g.colsum = l.s_qkv;
            PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
            PGV_TRY(pgv_launch_vit_attn(ctx, v->dtype, qkv, 3 * C, ao, C, Tc, N, C, v->cfg.heads, s));
            g = GemmArgs{}; g.A = ao; g.lda = C; g.W = l.wo; g.ldw = C; g.bias = l.bo; g.C = resid; g.ldc = C; g.M = M; g.N = C; g.K = C;
            g.epi = PGV_EPI_BIAS_RESID_LNOUT; g.gnext = l.ln2_g; g.x16 = xn; g.ldx16 = C; g.stats_part = part; g.rowmean = rowmean; g.cshift = l.m_bo;
            PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
            PGV_TRY(pgv_launch_ln_stats(part, rowstat, rowmean, l.m_bo, M, NP, C, v->cfg.eps, s));
            g = GemmArgs{}; g.A = xn; g.lda = C; g.W = l.w1; g.ldw = C; g.bias = l.b2_fc1; g.C = hbuf; g.ldc = I; g.M = M; g.N = I; g.K = C;
            g.epi = PGV_EPI_LN_BIAS_QGELU; g.rowstat = rowstat; g.colsum = l.s_fc1;
            PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
            g = GemmArgs{}; g.A = hbuf; g.lda = I; g.W = l.w2; g.ldw = I; g.bias = l.b2; g.C = resid; g.ldc = C; g.M = M; g.N = C; g.K = I;
            if (li + 1 < n_layers) {
                g.epi = PGV_EPI_BIAS_RESID_LNOUT; g.gnext = v->layers[li + 1].ln1_g; g.x16 = xn; g.ldx16 = C; g.stats_part = part; g.rowmean = rowmean; g.cshift = l.m_b2;
                PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
                PGV_TRY(pgv_launch_ln_stats(part, rowstat, rowmean, l.m_b2, M, NP, C, v->cfg.eps, s));
            } else {
                // the last executed layer feeds no LayerNorm: its "operand copy" with gamma = 1 is round16(resid) -- the hidden state itself,
                // written straight into the caller's buffer (no cast pass over the fp32 residual); the statistics are discarded
                g.epi = PGV_EPI_BIAS_RESID_LNOUT; g.gnext = v->ones; g.x16 = (char*)d_hidden + (size_t)t0 * N * C * 2; g.ldx16 = C; g.stats_part = part;
                PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
            }
            return PGV_OK;
        }
        PGV_TRY(pgv_launch_layernorm(v->dtype, resid, l.ln1_g, l.ln1_b, v->cfg.eps, xn, M, C, s));
        g = GemmArgs{}; g.A = xn; g.lda = C; g.W = l.wqkv; g.ldw = C; g.bias = l.bqkv; g.C = qkv; g.ldc = 3 * C; g.M = M; g.N = 3 * C; g.K = C; g.epi = PGV_EPI_BIAS;
        PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
        PGV_TRY(pgv_launch_vit_attn(ctx, v->dtype, qkv, 3 * C, ao, C, Tc, N, C, v->cfg.heads, s));
        g = GemmArgs{}; g.A = ao; g.lda = C; g.W = l.wo; g.ldw = C; g.bias = l.bo; g.C = resid; g.ldc = C; g.M = M; g.N = C; g.K = C; g.epi = PGV_EPI_BIAS_RESID;
        PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
        PGV_TRY(pgv_launch_layernorm(v->dtype, resid, l.ln2_g, l.ln2_b, v->cfg.eps, xn, M, C, s));
        g = GemmArgs{}; g.A = xn; g.lda = C; g.W = l.w1; g.ldw = C; g.bias = l.b1; g.C = hbuf; g.ldc = I; g.M = M; g.N = I; g.K = C; g.epi = PGV_EPI_BIAS_QGELU;
        PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
        g = GemmArgs{}; g.A = hbuf; g.lda = I; g.W = l.w2; g.ldw = I; g.bias = l.b2; g.C = resid; g.ldc = C; g.M = M; g.N = C; g.K = I; g.epi = PGV_EPI_BIAS_RESID;
        PGV_TRY(pgv_launch_gemm(ctx, v->dtype, g, s));
        return PGV_OK;
    };
    auto tail = [&](Lane& ln, int t0, int Tc) -> int {
        if (!(fold && n_layers > 0)) PGV_TRY(pgv_launch_cast(v->dtype, ln.resid, (char*)d_hidden + (size_t)t0 * N * C * 2, (size_t)Tc * N * C, ln.s));
        return PGV_OK;
    };
    auto pending = [&]() { for (int k = 0; k < lanes; ++k) if (L[k].done < L[k].T) return true; return false; };
    auto run = [&]() -> int {
        while (pending()) {
            int ct0[MAX_LANES], cT[MAX_LANES];
            for (int k = 0; k < lanes; ++k) {
                ct0[k] = L[k].t0 + L[k].done;
                cT[k] = (L[k].T - L[k].done) < Tc_max ? (L[k].T - L[k].done) : Tc_max;
            }
            for (int k = 0; k < lanes; ++k) if (cT[k] > 0) PGV_TRY(embed(L[k], ct0[k], cT[k]));
            for (int li = 0; li < n_layers; ++li)
                for (int k = 0; k < lanes; ++k) if (cT[k] > 0) PGV_TRY(layer(L[k], ct0[k], cT[k], li));
            for (int k = 0; k < lanes; ++k) if (cT[k] > 0) { PGV_TRY(tail(L[k], ct0[k], cT[k])); L[k].done += cT[k]; }
        }
        return PGV_OK;
    };
    // Common exit: whatever run() returned, every forked lane is joined back into the caller's stream and the arena is released
    // (its event recorded) -- a failed pass must not leave aux-lane kernels unordered against the next user of the shared arena.
    int rc = run();
    for (int k = 1; k < lanes; ++k) {
        hipError_t e = hipEventRecord(ctx->ev_join[k - 1], ctx->aux_stream[k - 1]);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, ctx->ev_join[k - 1], 0);
        if (e != hipSuccess && rc == PGV_OK) { pgv_set_error("pgv_vit_forward: lane join: %s", hipGetErrorString(e)); rc = PGV_EHIP; }
    }
    const int rel = pgv_ws_release(ctx, s);
    if (rc != PGV_OK) return rc;
    PGV_TRY(rel);
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}
