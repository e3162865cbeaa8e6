// mm_projector + LLaMA decoder host side: packed weights, KV cache, prefill / decode drivers.
// Replaces VideoChatGPTLlamaForCausalLM.forward (video_chatgpt/model/video_chatgpt.py:193-251 -> :82-175 -> HF LlamaModel)
// as driven by model.generate (video_chatgpt/inference.py:105-112).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "llm_internal.h"
#include "weights.h"

constexpr int kMaxPos = 4096;   // max_position_embeddings of LLaVA-1.5 / Vicuna-1.5

struct LlmLayer {
    float *in_g, *post_g;
    void *wqkv, *wo, *wgu, *wdown;
    // fp8 decode copies (pgv_llm_quantize_fp8): e4m3 blocked matrices + per-row power-of-two scales; null until quantised
    void *q_wqkv = nullptr, *q_wo = nullptr, *q_wgu = nullptr, *q_wdown = nullptr;
    float *s_wqkv = nullptr, *s_wo = nullptr, *s_wgu = nullptr, *s_wdown = nullptr;
};

struct pgv_llm {
    pgv_ctx* ctx;
    pgv_llm_config cfg;
    int dtype;
    char* blob = nullptr;
    size_t blob_bytes = 0;
    void *embed, *lm_head;
    float* norm_g;
    float* rope;   // float2 [kMaxPos][64]
    std::vector<LlmLayer> layers;
    std::set<std::string> loaded;
    int expected = 0;
    int vocab_cap = 0;   // allocated rows of embed / lm_head (cfg.vocab at creation + 64 spare)
    char* blob8 = nullptr;   // fp8 copies + scales (decode weight stream halves; the 16-bit copies hold the dequantised values)
    void* q_head = nullptr; float* s_head = nullptr;
    bool fp8 = false;
    int generation = 0;      // bumped by every call that changes what a captured decode graph baked in (vocabulary size, weight pointers)
};

struct pgv_kv {
    pgv_llm* llm;
    int B, max_seq;
    char* blob = nullptr;
    std::vector<void*> Kc, Vc;          // per layer [B][heads][max_seq][128]
    // fixed-address decode buffers (so a decode step can be captured into a hipGraph)
    float* resid; void* xn; void* qkv; void* ao; void* act; float* logits;
    float* amax_val; int* amax_idx;     // greedy candidates of the lm_head GEMV, tile-major: [ceil(B / 16)][vocab_cap / 16][16]
    float* ssq;                         // sum-of-squares partials of the folded RMSNorm, tile-major: [ceil(B / 16)][hidden / 16][16] (see GemvArgs in gemv.hip)
    int ssq_ts, amax_ts;                // their tile strides in elements: hidden, vocab_cap
    void* k8_part;                      // 8-phase residual producers (gemv.hip gemv_k8_kernel): phase tiles [hidden / 16][8][column tiles][64] float4
    float* dattn_part; unsigned* dattn_ticket;   // context-split decode attention: partial states [B * heads][8][130], arrival tickets [B * heads] (zero between launches)
    int *d_pos, *d_cur, *d_step, *d_done, *d_hist;
    std::vector<int> h_len;
    std::vector<int> h_meta;           // staging for prefill row maps (kept alive across the async copy)
    int active = 0;                     // sequences of the last prefill
    // one decode step captured as a hipGraph (all kernel arguments are fixed device addresses; positions, current
    // tokens and step counters live in device memory and are advanced by the argmax kernel)
    hipGraphExec_t gexec[2] = {nullptr, nullptr};      // [0]: one token step, [1]: kGraphSteps consecutive steps
    int g_B = 0, g_eos = 0, g_flags = 0, g_gen = -1, g_topk = 0;
    float g_temp = 0.f;
    float* d_u = nullptr;               // uniforms of a sampled decode run: [max_seq][B], indexed by the device-side step counter
    float s_temp = 1.f; int s_topk = 0; // sampling parameters of the run in flight (AM_SAMPLE)
    bool warmed = false;
};

extern "C" int pgv_llm_create(pgv_ctx* ctx, const pgv_llm_config* cfg, int dtype, pgv_llm** out) {
    PGV_CHECK(ctx && cfg && out, "pgv_llm_create: null argument");
    PGV_CHECK(dtype == PGV_F16 || dtype == PGV_BF16, "pgv_llm_create: dtype must be PGV_F16 or PGV_BF16");
    PGV_CHECK(cfg->heads * kHD == cfg->hidden, "pgv_llm_create: head_dim must be 128 (hidden %d, heads %d)", cfg->hidden, cfg->heads);
    PGV_CHECK(cfg->hidden % 256 == 0, "pgv_llm_create: hidden must be a multiple of 256");
    PGV_CHECK(cfg->inter % 64 == 0, "pgv_llm_create: intermediate size must be a multiple of 64");
    PGV_CHECK(cfg->vocab > 0 && cfg->layers >= 0, "pgv_llm_create: bad vocab/layers");
    pgv_llm* m = new pgv_llm();
    m->ctx = ctx; m->cfg = *cfg; m->dtype = dtype;
    m->vocab_cap = (cfg->vocab + 64 + 15) / 16 * 16;   // lm_head is fragment-blocked: rows padded to 16 (zero rows)
    const size_t H = cfg->hidden, I = cfg->inter, V = m->vocab_cap, L = cfg->layers;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += pgv_align(bytes); return o; };
    const size_t o_embed = take(V * H * 2), o_head = take(V * H * 2), o_norm = take(H * 4), o_rope = take((size_t)kMaxPos * 64 * 8);
    struct LO { size_t in_g, post_g, wqkv, wo, wgu, wdown; };
    std::vector<LO> lo(L);
    for (size_t i = 0; i < L; ++i) {
        lo[i].in_g = take(H * 4); lo[i].post_g = take(H * 4);
        lo[i].wqkv = take(3 * H * H * 2); lo[i].wo = take(H * H * 2); lo[i].wgu = take(2 * I * H * 2); lo[i].wdown = take(H * I * 2);
    }
    hipError_t e = hipMalloc((void**)&m->blob, off);
    if (e != hipSuccess) { delete m; pgv_set_error("pgv_llm_create: hipMalloc(%zu MiB): %s", off >> 20, hipGetErrorString(e)); return PGV_ENOMEM; }
    m->blob_bytes = off;
    (void)hipMemset(m->blob + o_head, 0, V * H * 2);     // padded lm_head rows must be finite
    char* b = m->blob;
    m->embed = b + o_embed; m->lm_head = b + o_head; m->norm_g = (float*)(b + o_norm); m->rope = (float*)(b + o_rope);
    m->layers.resize(L);
    for (size_t i = 0; i < L; ++i) {
        LlmLayer& l = m->layers[i];
        l.in_g = (float*)(b + lo[i].in_g); l.post_g = (float*)(b + lo[i].post_g);
        l.wqkv = b + lo[i].wqkv; l.wo = b + lo[i].wo; l.wgu = b + lo[i].wgu; l.wdown = b + lo[i].wdown;
    }
    // RoPE table exactly as LlamaRotaryEmbedding (HF:llama/modeling_llama.py:96-126): fp32 inv_freq, fp32 angle, fp32 cos/sin
    std::vector<float> tab((size_t)kMaxPos * 64 * 2);
    for (int j = 0; j < 64; ++j) {
        const float inv = 1.0f / powf(cfg->rope_theta, (float)(2 * j) / (float)kHD);
        for (int p = 0; p < kMaxPos; ++p) {
            const float ang = (float)p * inv;
            tab[((size_t)p * 64 + j) * 2] = cosf(ang);
            tab[((size_t)p * 64 + j) * 2 + 1] = sinf(ang);
        }
    }
    e = hipMemcpy(m->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->blob); delete m; pgv_set_error("pgv_llm_create: rope upload: %s", hipGetErrorString(e)); return PGV_EHIP; }
    m->expected = 3 + 9 * (int)L;
    *out = m;
    return PGV_OK;
}

extern "C" void pgv_llm_destroy(pgv_llm* llm) {
    if (!llm) return;
    if (llm->blob) (void)hipFree(llm->blob);
    if (llm->blob8) (void)hipFree(llm->blob8);
    delete llm;
}

extern "C" int pgv_llm_missing(const pgv_llm* llm) { return llm ? llm->expected - (int)llm->loaded.size() : -1; }
extern "C" int pgv_llm_vocab(const pgv_llm* llm) { return llm ? llm->cfg.vocab : -1; }

extern "C" int pgv_llm_resize_vocab(pgv_llm* m, int new_vocab, void* stream) {
    PGV_CHECK(m != nullptr, "pgv_llm_resize_vocab: null model");
    PGV_CHECK(new_vocab >= 1 && new_vocab <= m->vocab_cap, "pgv_llm_resize_vocab: %d outside [1,%d] (64 spare rows are allocated at creation)", new_vocab, m->vocab_cap);
    if (new_vocab == m->cfg.vocab) return PGV_OK;
    if (m->fp8) { pgv_set_error("pgv_llm_resize_vocab: the weights are already quantised to fp8; resize before pgv_llm_quantize_fp8"); return PGV_ESTATE; }
    const size_t H = m->cfg.hidden;
    if (new_vocab > m->cfg.vocab) {
        const size_t off = (size_t)m->cfg.vocab * H * 2, bytes = (size_t)(new_vocab - m->cfg.vocab) * H * 2;
        PGV_HIP(hipMemsetAsync((char*)m->embed + off, 0, bytes, (hipStream_t)stream));
        // lm_head is fragment-blocked (rows interleaved inside 1 KiB blocks): zero the new rows in place (a shrink followed by a
        // grow must not resurrect old rows)
        PGV_TRY(pgv_zero_rows_blocked(m->lm_head, m->cfg.vocab, new_vocab - m->cfg.vocab, (long long)H, (hipStream_t)stream));
    }
    m->cfg.vocab = new_vocab;
    m->generation += 1;           // captured decode graphs have the old vocabulary baked into lm_head / the token pick
    return PGV_OK;
}

extern "C" int pgv_llm_load_rows(pgv_llm* m, const char* name, const void* data, int src_dtype, int on_device, int row0, int nrows, int64_t numel, void* stream) {
    PGV_CHECK(m && name && data, "pgv_llm_load_rows: null argument");
    PGV_CHECK(row0 >= 0 && nrows >= 1 && row0 + nrows <= m->cfg.vocab, "pgv_llm_load_rows: rows [%d,%d) outside the vocabulary (%d)", row0, row0 + nrows, m->cfg.vocab);
    PGV_CHECK(numel == (int64_t)nrows * m->cfg.hidden, "pgv_llm_load_rows: size mismatch for '%s': %lld elements given for %d rows of %d", name, (long long)numel,
              nrows, m->cfg.hidden);
    if (m->fp8) { pgv_set_error("pgv_llm_load_rows: the weights are quantised to fp8 (the fp8 copies would go stale); load into a fresh model"); return PGV_ESTATE; }
    PackDst d;
    d.dst_dtype = m->dtype; d.rows = nrows; d.cols = m->cfg.hidden; d.dst_stride = d.cols; d.row_off = row0;
    if (!strcmp(name, "model.embed_tokens.weight")) d.ptr = m->embed;
    else if (!strcmp(name, "lm_head.weight")) { d.ptr = m->lm_head; d.blocked = true; }
    else { pgv_set_error("pgv_llm_load_rows: '%s' is not a vocabulary matrix", name); return PGV_ENAME; }
    PGV_TRY(pgv_pack_tensor(d, data, src_dtype, on_device, (hipStream_t)stream));
    m->loaded.insert(name);
    return PGV_OK;
}

extern "C" int pgv_llm_load_tensor(pgv_llm* m, const char* name_in, const void* data, int src_dtype, int on_device, int64_t numel, void* stream) {
    PGV_CHECK(m && name_in && data, "pgv_llm_load_tensor: null argument");
    if (m->fp8) { pgv_set_error("pgv_llm_load_tensor: the weights are quantised to fp8 (the fp8 copies would go stale); load into a fresh model"); return PGV_ESTATE; }
    std::string name(name_in);
    const long long H = m->cfg.hidden, I = m->cfg.inter, V = m->cfg.vocab;
    PackDst d;
    auto vec = [&](float* ptr, long long n) { d.ptr = ptr; d.rows = 1; d.cols = n; };
    auto mat = [&](void* ptr, long long r, long long c, long long roff) { d.ptr = ptr; d.dst_dtype = m->dtype; d.rows = r; d.cols = c; d.row_off = roff; };
    if (name == "model.embed_tokens.weight") mat(m->embed, V, H, 0);
    else if (name == "lm_head.weight") { mat(m->lm_head, V, H, 0); d.blocked = true; }
    else if (name == "model.norm.weight") vec(m->norm_g, H);
    else if (name.rfind("model.layers.", 0) == 0) {
        const char* p = name.c_str() + 13;
        char* end = nullptr;
        long li = strtol(p, &end, 10);
        if (end == p || *end != '.' || li < 0 || li >= m->cfg.layers) { pgv_set_error("pgv_llm_load_tensor: bad layer index in '%s'", name_in); return PGV_ENAME; }
        std::string rest(end + 1);
        LlmLayer& l = m->layers[li];
        if (rest == "self_attn.q_proj.weight") { mat(l.wqkv, H, H, 0); d.blocked = true; }
        else if (rest == "self_attn.k_proj.weight") { mat(l.wqkv, H, H, H); d.blocked = true; }
        else if (rest == "self_attn.v_proj.weight") { mat(l.wqkv, H, H, 2 * H); d.blocked = true; }
        else if (rest == "self_attn.o_proj.weight") { mat(l.wo, H, H, 0); d.blocked = true; }
        else if (rest == "mlp.gate_proj.weight") { mat(l.wgu, I, H, 0); d.row_blk = 32; d.blk_stride = 64; d.blocked = true; }       // [32 gate | 32 up] per 64 rows
        else if (rest == "mlp.up_proj.weight") { mat(l.wgu, I, H, 32); d.row_blk = 32; d.blk_stride = 64; d.blocked = true; }
        else if (rest == "mlp.down_proj.weight") { mat(l.wdown, H, I, 0); d.blocked = true; }
        else if (rest == "input_layernorm.weight") vec(l.in_g, H);
        else if (rest == "post_attention_layernorm.weight") vec(l.post_g, H);
        else if (rest == "self_attn.rotary_emb.inv_freq") return PGV_OK;
        else { pgv_set_error("pgv_llm_load_tensor: unexpected key '%s'", name_in); return PGV_ENAME; }
    } else { pgv_set_error("pgv_llm_load_tensor: unexpected key '%s'", name_in); return PGV_ENAME; }
    d.dst_stride = d.cols;
    PGV_CHECK(numel == d.rows * d.cols, "pgv_llm_load_tensor: size mismatch for '%s': %lld elements given, the model expects %lld x %lld", name_in,
              (long long)numel, d.rows, d.cols);
    PGV_TRY(pgv_pack_tensor(d, data, src_dtype, on_device, (hipStream_t)stream));
    m->loaded.insert(name);
    return PGV_OK;
}

// ---------------------------------------------------------------------------------------------
// fp8 weight path (BASELINE config 5): quantise every decoder matrix + lm_head once, after loading
// ---------------------------------------------------------------------------------------------
extern "C" int pgv_llm_quantize_fp8(pgv_ctx* ctx, pgv_llm* m, void* stream) {
    PGV_CHECK(ctx && m, "pgv_llm_quantize_fp8: null argument");
    if (pgv_llm_missing(m) != 0) { pgv_set_error("pgv_llm_quantize_fp8: %d weight tensors not loaded", pgv_llm_missing(m)); return PGV_ESTATE; }
    if (m->fp8) return PGV_OK;
    PGV_CHECK(m->cfg.hidden % 64 == 0 && m->cfg.inter % 64 == 0, "pgv_llm_quantize_fp8: hidden and intermediate sizes must be multiples of 64");
    hipStream_t s = (hipStream_t)stream;
    const size_t H = m->cfg.hidden, I = m->cfg.inter, V = m->vocab_cap, L = m->cfg.layers;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += pgv_align(bytes); return o; };
    struct LO { size_t qkv, o, gu, down, sqkv, so, sgu, sdown; };
    std::vector<LO> lo(L);
    const size_t o_head = take(V * H), o_shead = take(V * 4);
    for (size_t i = 0; i < L; ++i) {
        lo[i].qkv = take(3 * H * H); lo[i].o = take(H * H); lo[i].gu = take(2 * I * H); lo[i].down = take(H * I);
        lo[i].sqkv = take(3 * H * 4); lo[i].so = take(H * 4); lo[i].sgu = take(2 * I * 4); lo[i].sdown = take(H * 4);
    }
    hipError_t e = hipMalloc((void**)&m->blob8, off);
    if (e != hipSuccess) { pgv_set_error("pgv_llm_quantize_fp8: hipMalloc(%zu MiB): %s", off >> 20, hipGetErrorString(e)); return PGV_ENOMEM; }
    char* b = m->blob8;
    m->q_head = b + o_head; m->s_head = (float*)(b + o_shead);
    PGV_TRY(pgv_launch_quantize_fp8(m->dtype, m->lm_head, m->q_head, m->s_head, (long long)V, (long long)H, s));
    for (size_t i = 0; i < L; ++i) {
        LlmLayer& l = m->layers[i];
        l.q_wqkv = b + lo[i].qkv; l.q_wo = b + lo[i].o; l.q_wgu = b + lo[i].gu; l.q_wdown = b + lo[i].down;
        l.s_wqkv = (float*)(b + lo[i].sqkv); l.s_wo = (float*)(b + lo[i].so); l.s_wgu = (float*)(b + lo[i].sgu); l.s_wdown = (float*)(b + lo[i].sdown);
        PGV_TRY(pgv_launch_quantize_fp8(m->dtype, l.wqkv, l.q_wqkv, l.s_wqkv, 3 * (long long)H, (long long)H, s));
        PGV_TRY(pgv_launch_quantize_fp8(m->dtype, l.wo, l.q_wo, l.s_wo, (long long)H, (long long)H, s));
        PGV_TRY(pgv_launch_quantize_fp8(m->dtype, l.wgu, l.q_wgu, l.s_wgu, 2 * (long long)I, (long long)H, s));
        PGV_TRY(pgv_launch_quantize_fp8(m->dtype, l.wdown, l.q_wdown, l.s_wdown, (long long)H, (long long)I, s));
    }
    PGV_HIP(hipStreamSynchronize(s));
    m->fp8 = true;
    m->generation += 1;           // decode graphs captured before now stream the 16-bit matrices
    return PGV_OK;
}

extern "C" int pgv_llm_is_fp8(const pgv_llm* m) { return (m && m->fp8) ? 1 : 0; }

// Read one decoder matrix back as row-major fp32 [rows, cols] under its HF key (after quantisation: the dequantised values the
// whole path computes with) -- what a parity test or a checkpoint writer needs.
extern "C" int pgv_llm_get_weight(pgv_ctx* ctx, pgv_llm* m, const char* name_in, float* d_out, void* stream) {
    PGV_CHECK(ctx && m && name_in && d_out, "pgv_llm_get_weight: null argument");
    std::string name(name_in);
    const long long H = m->cfg.hidden, I = m->cfg.inter, V = m->cfg.vocab;
    hipStream_t s = (hipStream_t)stream;
    if (name == "lm_head.weight") return pgv_launch_unpack_blocked(m->dtype, m->lm_head, d_out, m->vocab_cap, H, 0, 0, 0, V, s);
    if (name.rfind("model.layers.", 0) != 0) { pgv_set_error("pgv_llm_get_weight: unsupported key '%s'", name_in); return PGV_ENAME; }
    const char* p = name.c_str() + 13;
    char* end = nullptr;
    long li = strtol(p, &end, 10);
    if (end == p || *end != '.' || li < 0 || li >= m->cfg.layers) { pgv_set_error("pgv_llm_get_weight: bad layer index in '%s'", name_in); return PGV_ENAME; }
    std::string rest(end + 1);
    const LlmLayer& l = m->layers[li];
    if (rest == "self_attn.q_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wqkv, d_out, 3 * H, H, 0, 0, 0, H, s);
    if (rest == "self_attn.k_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wqkv, d_out, 3 * H, H, 0, 0, H, H, s);
    if (rest == "self_attn.v_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wqkv, d_out, 3 * H, H, 0, 0, 2 * H, H, s);
    if (rest == "self_attn.o_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wo, d_out, H, H, 0, 0, 0, H, s);
    if (rest == "mlp.gate_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wgu, d_out, 2 * I, H, 32, 64, 0, I, s);
    if (rest == "mlp.up_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wgu, d_out, 2 * I, H, 32, 64, 32, I, s);
    if (rest == "mlp.down_proj.weight") return pgv_launch_unpack_blocked(m->dtype, l.wdown, d_out, H, I, 0, 0, 0, H, s);
    pgv_set_error("pgv_llm_get_weight: unsupported key '%s'", name_in);
    return PGV_ENAME;
}

// ---------------------------------------------------------------------------------------------
// KV cache
// ---------------------------------------------------------------------------------------------
extern "C" int pgv_kv_create(pgv_ctx* ctx, pgv_llm* llm, int batch, int max_seq, pgv_kv** out) {
    PGV_CHECK(ctx && llm && out, "pgv_kv_create: null argument");
    PGV_CHECK(batch >= 1 && batch <= kMaxBatch, "pgv_kv_create: batch %d outside [1,%d] (the decode GEMVs tile up to 4 x 16 sequences)", batch, kMaxBatch);
    PGV_CHECK(max_seq >= 1 && max_seq <= kMaxPos, "pgv_kv_create: max_seq %d outside [1,%d]", max_seq, kMaxPos);
    pgv_kv* kv = new pgv_kv();
    kv->llm = llm; kv->B = batch; kv->max_seq = max_seq;
    const size_t H = llm->cfg.hidden, I = llm->cfg.inter, V = llm->vocab_cap, L = llm->cfg.layers, B = batch;
    const size_t per = pgv_align(B * llm->cfg.heads * (size_t)max_seq * kHD * 2);
    const size_t CTn = B <= 16 ? 1 : (B <= 32 ? 2 : 4);      // column tiles the decode GEMVs run with (1 / 2 / 4): every tile's slice of the side arrays is written, also the empty ones
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += pgv_align(bytes); return o; };
    const size_t o_cache = take(per * 2 * L);
    const size_t o_resid = take(B * H * 4), o_xn = take(CTn * 16 * H * 2),      // xn: row-major [B][H] up to 8 sequences, fragment-blocked [H / 32][CTn][4][16][8] beyond (gv_xblk_offset)
                 o_qkv = take(B * 3 * H * 2), o_ao = take(B * H * 2), o_act = take(B * I * 2),
                 o_logits = take(B * V * 4), o_ssq = take(CTn * H * 4), o_av = take(CTn * V * 4), o_ai = take(CTn * V * 4), o_ints = take((4 * B + B * (size_t)max_seq) * 4),
                 o_u = take(B * (size_t)max_seq * 4), o_dpart = take(B * llm->cfg.heads * (size_t)kDattnSplitMax * kDattnPart * 4), o_dtick = take(B * llm->cfg.heads * 4),
                 o_k8 = take((H / 16) * 8 * CTn * 64 * 16);
    hipError_t e = hipMalloc((void**)&kv->blob, off);
    if (e != hipSuccess) { delete kv; pgv_set_error("pgv_kv_create: hipMalloc(%zu MiB): %s", off >> 20, hipGetErrorString(e)); return PGV_ENOMEM; }
    char* b = kv->blob;
    for (size_t i = 0; i < L; ++i) { kv->Kc.push_back(b + o_cache + per * 2 * i); kv->Vc.push_back(b + o_cache + per * (2 * i + 1)); }
    kv->resid = (float*)(b + o_resid); kv->xn = b + o_xn; kv->qkv = b + o_qkv; kv->ao = b + o_ao; kv->act = b + o_act; kv->logits = (float*)(b + o_logits); kv->ssq = (float*)(b + o_ssq); kv->amax_val = (float*)(b + o_av); kv->amax_idx = (int*)(b + o_ai);
    int* ints = (int*)(b + o_ints);
    kv->d_pos = ints; kv->d_cur = ints + B; kv->d_step = ints + 2 * B; kv->d_done = ints + 3 * B; kv->d_hist = ints + 4 * B;
    kv->d_u = (float*)(b + o_u);
    kv->dattn_part = (float*)(b + o_dpart); kv->dattn_ticket = (unsigned*)(b + o_dtick);
    kv->k8_part = (void*)(b + o_k8);
    e = hipMemset(ints, 0, (4 * B + B * (size_t)max_seq) * 4);
    if (e == hipSuccess) e = hipMemset(kv->dattn_ticket, 0, B * llm->cfg.heads * 4);
    if (e == hipSuccess) e = hipMemset(kv->ssq, 0, CTn * H * 4);
    if (e == hipSuccess) e = hipMemset(kv->xn, 0, CTn * 16 * H * 2);            // the columns of absent sequences are read (never stored): finite
    kv->ssq_ts = (int)H; kv->amax_ts = (int)V;
    if (e != hipSuccess) { (void)hipFree(kv->blob); delete kv; pgv_set_error("pgv_kv_create: memset: %s", hipGetErrorString(e)); return PGV_EHIP; }
    kv->h_len.assign(batch, 0);
    *out = kv;
    return PGV_OK;
}

extern "C" void pgv_kv_destroy(pgv_kv* kv) {
    if (!kv) return;
    for (auto g : kv->gexec) if (g) (void)hipGraphExecDestroy(g);
    if (kv->blob) (void)hipFree(kv->blob);
    delete kv;
}

extern "C" int pgv_kv_len(const pgv_kv* kv, int b) { return (kv && b >= 0 && b < kv->B) ? kv->h_len[b] : -1; }

// ---------------------------------------------------------------------------------------------
// shared tail: final norm + lm_head + argmax on kv->resid [B,H]
// ---------------------------------------------------------------------------------------------
// kv->xn holds round16(resid * norm_g) and kv->ssq the `nparts` sum-of-squares partials of resid (folded final RMSNorm): lm_head scales its
// fp32 accumulators by rstd.
// PGV_LLM_NORM_FOLD=0: decode with a standalone RMSNorm launch in front of every consumer GEMV (HF's order of operations: normalise, round
// to 16 bits, multiply) instead of the folded form -- a bisect switch for real-checkpoint regressions, not a fallback: both paths
// are tested against the same goldens (tests/test_gpu_llm.py::test_unfolded_decoder_path_matches_goldens).
static bool norm_fold_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PGV_LLM_NORM_FOLD"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}

// One decode GEMV on the fp8 weights (weight-only form: 16-bit x, e4m3 codes widened in registers; gemv.hip).
static int gemv8(pgv_ctx* ctx, pgv_llm* m, pgv_kv*, int mode, const void* w8, const float* ws, const void* x, int K, void* out, int ldo, int N, int B,
                 hipStream_t s, const GemvNorm* nm) {
    return pgv_launch_gemv(ctx, m->dtype, mode, w8, x, K, out, ldo, N, K, B, s, ws, nm);
}

static int lm_head_and_pick(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, int B, int eos, int flags, hipStream_t s, int nparts) {
    const int H = m->cfg.hidden, V = m->cfg.vocab;
    GemvNorm nm; nm.ssq_in = nparts > 0 ? kv->ssq : nullptr; nm.nparts_in = nparts; nm.hidden = H; nm.eps = m->cfg.eps;     // nparts 0: kv->xn is already normalised
    nm.ssq_ts = kv->ssq_ts; nm.amax_ts = kv->amax_ts;
    nm.x_blocked = nparts > 0;                       // the folded path hands kv->xn over blocked at batches beyond 16 (pgv_gemv_xblk_tiles)
    const bool greedy = !(flags & AM_SAMPLE);
    if (greedy) { nm.amax_val = kv->amax_val; nm.amax_idx = kv->amax_idx; }
    if (m->fp8 && nparts > 0) PGV_TRY(gemv8(ctx, m, kv, GV_F32, m->q_head, m->s_head, kv->xn, H, kv->logits, V, V, B, s, &nm));
    else PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_F32, m->fp8 ? m->q_head : m->lm_head, kv->xn, H, kv->logits, V, V, H, B, s, m->fp8 ? m->s_head : nullptr, &nm));
    pgv_prof_begin(ctx, 6, s);
    if (flags & AM_SAMPLE)
        PGV_TRY(pgv_launch_sample(kv->logits, V, B, kv->s_temp, kv->s_topk, kv->d_u, B, 1, kv->d_cur, kv->d_pos, kv->d_step, kv->d_hist, kv->max_seq, kv->d_done,
                                  eos, flags & (AM_INC_POS | AM_RECORD), s));
    else
        PGV_TRY(pgv_launch_argmax_parts(kv->amax_val, kv->amax_idx, (V + 15) / 16, kv->amax_ts, V, B, kv->d_cur, kv->d_pos, kv->d_step, kv->d_hist, kv->max_seq, kv->d_done, eos,
                                        flags, s));
    pgv_prof_end(ctx, 6, s, 0.0, 0.0);
    return PGV_OK;
}

// ---------------------------------------------------------------------------------------------
// prefill
// ---------------------------------------------------------------------------------------------
// `append`: the rows continue the sequences already in the cache (row p of sequence b sits at position h_len[b] + p and attends to the cached
// prefix as well): VideoChatGPTLlamaForCausalLM.forward with past_key_values and input_ids.shape[1] > 1 (video_chatgpt/model/video_chatgpt.py:193-251;
// the splice still runs when the new ids carry a placeholder run, :103).  Same kernels, same per-row arithmetic: an appended row is bitwise the
// row of one full prefill over prefix + new tokens.
static int prefill_impl(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, const int32_t* h_ids, const int32_t* h_seq_lens, int B, const void* d_video,
                        int Vt, const int32_t* h_vid_pos, float* d_logits, int32_t* d_next, float* d_all_logits, int ld_all, void* stream, bool append) {
    const char* who = append ? "pgv_llm_prefill_append" : "pgv_llm_prefill";
    PGV_CHECK(ctx && m && kv && h_ids && h_seq_lens, "%s: null argument", who);
    PGV_CHECK(kv->llm == m, "%s: kv cache belongs to another model", who);
    PGV_CHECK(B >= 1 && B <= kv->B, "%s: batch %d outside [1,%d]", who, B, kv->B);
    if (pgv_llm_missing(m) != 0) { pgv_set_error("%s: %d weight tensors not loaded", who, pgv_llm_missing(m)); return PGV_ESTATE; }
    if (append && kv->active != B) { pgv_set_error("pgv_llm_prefill_append: %d sequences for a cache that holds %d prefilled ones", B, kv->active); return PGV_ESTATE; }
    hipStream_t s = (hipStream_t)stream;
    const int H = m->cfg.hidden, I = m->cfg.inter, heads = m->cfg.heads, vocab = m->cfg.vocab;
    // arrival tickets of the context-split decode attention: a launch re-arms its own, but an ABORTED decode launch (device fault, process kill
    // between launches) would leave one non-zero for the next user of this cache -- every prefill starts from zeros
    PGV_HIP(hipMemsetAsync(kv->dattn_ticket, 0, (size_t)kv->B * heads * sizeof(unsigned), s));
    int M = 0, max_len = 0;
    for (int b = 0; b < B; ++b) {
        const int off = append ? kv->h_len[b] : 0;
        PGV_CHECK(h_seq_lens[b] >= 1 && off + h_seq_lens[b] <= kv->max_seq, "%s: sequence %d has %d + %d tokens (cache holds %d)", who, b, off, h_seq_lens[b], kv->max_seq);
        M += h_seq_lens[b];
        if (h_seq_lens[b] > max_len) max_len = h_seq_lens[b];
    }
    // ---- host-built row maps: [row_src | row_b | row_pos | cu (B+1) | last_rows (B) | lens (B) = positions after this call | offs (B)] ----
    std::vector<int>& meta = kv->h_meta;
    meta.assign((size_t)3 * M + 4 * B + 1, 0);
    int* row_src = meta.data(); int* row_b = row_src + M; int* row_pos = row_b + M; int* cu = row_pos + M; int* last = cu + B + 1; int* lens = last + B; int* offs = lens + B;
    int r = 0;
    for (int b = 0; b < B; ++b) {
        cu[b] = r;
        const int off = append ? kv->h_len[b] : 0;
        const int vp = (d_video && h_vid_pos) ? h_vid_pos[b] : -1;
        if (vp >= 0) PGV_CHECK(vp + Vt + 1 < h_seq_lens[b], "%s: video run of sequence %d (start %d, %d rows) overruns its %d tokens", who, b, vp, Vt, h_seq_lens[b]);
        for (int p = 0; p < h_seq_lens[b]; ++p, ++r) {
            const int id = h_ids[r];
            const bool vid = vp >= 0 && p > vp && p <= vp + Vt;
            if (!vid) PGV_CHECK(id >= 0 && id < vocab, "%s: token id %d at row %d outside the vocabulary (%d)", who, id, r, vocab);
            row_src[r] = vid ? -(b * Vt + (p - vp - 1) + 1) : id;
            row_b[r] = b; row_pos[r] = off + p;
        }
        last[b] = r - 1; lens[b] = off + h_seq_lens[b]; offs[b] = off;
    }
    cu[B] = r;

    const size_t b_meta = pgv_align(meta.size() * 4), b_resid = pgv_align((size_t)M * H * 4), b_xn = pgv_align((size_t)M * H * 2),
                 b_qkv = pgv_align((size_t)M * 3 * H * 2), b_ao = pgv_align((size_t)M * H * 2), b_act = pgv_align((size_t)M * I * 2);
    PGV_TRY(pgv_ws_reserve(ctx, b_meta + b_resid + b_xn + b_qkv + b_ao + b_act, s));
    int* d_meta = (int*)pgv_ws_alloc(ctx, b_meta);
    float* resid = (float*)pgv_ws_alloc(ctx, b_resid);
    char* xn = (char*)pgv_ws_alloc(ctx, b_xn);
    char* qkv = (char*)pgv_ws_alloc(ctx, b_qkv);
    char* ao = (char*)pgv_ws_alloc(ctx, b_ao);
    char* act = (char*)pgv_ws_alloc(ctx, b_act);
    PGV_CHECK(d_meta && resid && xn && qkv && ao && act, "pgv_llm_prefill: workspace exhausted");
    PGV_HIP(hipMemcpyAsync(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, s));
    const int* d_row_src = d_meta; const int* d_row_b = d_meta + M; const int* d_row_pos = d_row_b + M; const int* d_cu = d_row_pos + M;
    const int* d_last = d_cu + B + 1; const int* d_lens = d_last + B; const int* d_offs = append ? d_lens + B : nullptr;
    PGV_HIP(hipMemcpyAsync(kv->d_pos, d_lens, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipMemsetAsync(kv->d_step, 0, (size_t)2 * kv->B * 4, s));      // step + done

    PGV_TRY(pgv_launch_embed_splice(m->dtype, d_row_src, m->embed, d_video, resid, M, H, s));
    double attn_flops = 0;
    for (int b = 0; b < B; ++b) attn_flops += 2.0 * (double)h_seq_lens[b] * (h_seq_lens[b] + 2.0 * offs[b]) * H;   // causal: 4 * S * (S / 2 + prefix) * H
    for (int li = 0; li < m->cfg.layers; ++li) {
        const LlmLayer& l = m->layers[li];
        PGV_TRY(pgv_launch_rmsnorm(m->dtype, resid, l.in_g, m->cfg.eps, xn, M, H, s));
        GemmArgs g{};
        g.A = xn; g.lda = H; g.W = l.wqkv; g.ldw = H; g.bias = nullptr; g.C = qkv; g.ldc = 3 * H; g.M = M; g.N = 3 * H; g.K = H; g.epi = PGV_EPI_NONE; g.w_blocked = true;
        PGV_TRY(pgv_launch_gemm(ctx, m->dtype, g, s));
        PGV_TRY(pgv_launch_rope_kv_write(m->dtype, qkv, d_row_b, d_row_pos, m->rope, kv->Kc[li], kv->Vc[li], M, H, heads, kv->max_seq, s));
        PGV_TRY(pgv_launch_prefill_attn(ctx, m->dtype, qkv, ao, kv->Kc[li], kv->Vc[li], d_cu, d_offs, B, max_len, H, heads, kv->max_seq, attn_flops, s));
        g = GemmArgs{}; g.A = ao; g.lda = H; g.W = l.wo; g.ldw = H; g.C = resid; g.ldc = H; g.M = M; g.N = H; g.K = H; g.epi = PGV_EPI_RESID; g.w_blocked = true;
        PGV_TRY(pgv_launch_gemm(ctx, m->dtype, g, s));
        PGV_TRY(pgv_launch_rmsnorm(m->dtype, resid, l.post_g, m->cfg.eps, xn, M, H, s));
        g = GemmArgs{}; g.A = xn; g.lda = H; g.W = l.wgu; g.ldw = H; g.C = act; g.ldc = I; g.M = M; g.N = 2 * I; g.K = H; g.epi = PGV_EPI_SWIGLU; g.w_blocked = true;
        PGV_TRY(pgv_launch_gemm(ctx, m->dtype, g, s));
        g = GemmArgs{}; g.A = act; g.lda = I; g.W = l.wdown; g.ldw = I; g.C = resid; g.ldc = H; g.M = M; g.N = H; g.K = I; g.epi = PGV_EPI_RESID; g.w_blocked = true;
        PGV_TRY(pgv_launch_gemm(ctx, m->dtype, g, s));
    }
    if (d_all_logits) {
        // the reference's forward output: final norm + lm_head over ALL positions (video_chatgpt.py:225-226) -- on request only
        const int n16 = (vocab + 15) & ~15;
        PGV_CHECK(ld_all >= n16 && ld_all % 4 == 0 && n16 <= m->vocab_cap, "%s: ld_all %d must be a multiple of 4 and at least %d", who, ld_all, n16);
        PGV_TRY(pgv_launch_rmsnorm(m->dtype, resid, m->norm_g, m->cfg.eps, xn, M, H, s));
        GemmArgs g{};
        g.A = xn; g.lda = H; g.W = m->lm_head; g.ldw = H; g.C = d_all_logits; g.ldc = ld_all; g.M = M; g.N = n16; g.K = H; g.epi = PGV_EPI_F32; g.w_blocked = true;
        PGV_TRY(pgv_launch_gemm(ctx, m->dtype, g, s));
    }
    // lm_head only on the last position of every sequence (the reference computes all S positions, video_chatgpt.py:226)
    PGV_TRY(pgv_launch_gather_rows(resid, d_last, kv->resid, B, H, s));
    if (norm_fold_enabled()) {
        PGV_TRY(pgv_launch_final_prep(m->dtype, kv->resid, m->norm_g, kv->xn, kv->ssq, B, H, s, true));
        PGV_TRY(lm_head_and_pick(ctx, m, kv, B, -1, 0, s, 1));
    } else {
        PGV_TRY(pgv_launch_rmsnorm(m->dtype, kv->resid, m->norm_g, m->cfg.eps, kv->xn, B, H, s));
        PGV_TRY(lm_head_and_pick(ctx, m, kv, B, -1, 0, s, 0));
    }
    if (d_logits) PGV_HIP(hipMemcpyAsync(d_logits, kv->logits, (size_t)B * vocab * 4, hipMemcpyDeviceToDevice, s));
    if (d_next) PGV_HIP(hipMemcpyAsync(d_next, kv->d_cur, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    for (int b = 0; b < kv->B; ++b) kv->h_len[b] = b < B ? lens[b] : 0;
    kv->active = B;
    PGV_HIP(hipGetLastError());
    PGV_TRY(pgv_ws_release(ctx, s));
    return PGV_OK;
}

extern "C" int pgv_llm_prefill(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, const int32_t* h_ids, const int32_t* h_seq_lens, int B, const void* d_video,
                               int Vt, const int32_t* h_vid_pos, float* d_logits, int32_t* d_next, float* d_all_logits, int ld_all, void* stream) {
    return prefill_impl(ctx, m, kv, h_ids, h_seq_lens, B, d_video, Vt, h_vid_pos, d_logits, d_next, d_all_logits, ld_all, stream, false);
}

extern "C" int pgv_llm_prefill_append(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, const int32_t* h_ids, const int32_t* h_seq_lens, int B, const void* d_video,
                                      int Vt, const int32_t* h_vid_pos, float* d_logits, int32_t* d_next, float* d_all_logits, int ld_all, void* stream) {
    return prefill_impl(ctx, m, kv, h_ids, h_seq_lens, B, d_video, Vt, h_vid_pos, d_logits, d_next, d_all_logits, ld_all, stream, true);
}

// Drop the tail of sequence b: the next pgv_llm_prefill_append / decode call continues at position `len` (the entries beyond it are simply
// overwritten).  What a chat turn needs when the new prompt shares only a prefix with the cached conversation (surplus decode steps past a stop
// string, a re-tokenised answer).
extern "C" int pgv_kv_truncate(pgv_kv* kv, int b, int len, void* stream) {
    PGV_CHECK(kv != nullptr, "pgv_kv_truncate: null cache");
    PGV_CHECK(b >= 0 && b < kv->active, "pgv_kv_truncate: sequence %d outside the %d prefilled ones", b, kv->active);
    PGV_CHECK(len >= 1 && len <= kv->h_len[b], "pgv_kv_truncate: length %d outside [1,%d]", len, kv->h_len[b]);
    kv->h_len[b] = len;
    PGV_HIP(hipMemcpyAsync(kv->d_pos + b, &kv->h_len[b], 4, hipMemcpyHostToDevice, (hipStream_t)stream));
    PGV_HIP(hipStreamSynchronize((hipStream_t)stream));          // the source is host bookkeeping that the next call may change
    return PGV_OK;
}

// ---------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------
static int decode_enqueue_unfolded(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, int eos, int flags, hipStream_t s) {
    const int H = m->cfg.hidden, I = m->cfg.inter, heads = m->cfg.heads, B = kv->active, L = m->cfg.layers;
    pgv_prof_begin(ctx, 6, s);
    PGV_TRY(pgv_launch_embed_tok_norm(m->dtype, kv->d_cur, m->embed, kv->resid, L > 0 ? m->layers[0].in_g : m->norm_g, kv->xn, kv->ssq, B, H, s, false));   // resid is what matters here
    pgv_prof_end(ctx, 6, s, 0.0, 0.0);
    double kv_bytes = 0;
    for (int b = 0; b < B; ++b) kv_bytes += 2.0 * 2.0 * (double)(kv->h_len[b] + 1) * H;
    for (int li = 0; li < L; ++li) {
        const LlmLayer& l = m->layers[li];
        const bool q8 = m->fp8;
        PGV_TRY(pgv_launch_rmsnorm(m->dtype, kv->resid, l.in_g, m->cfg.eps, kv->xn, B, H, s));
        PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_STORE16, q8 ? l.q_wqkv : l.wqkv, kv->xn, H, kv->qkv, 3 * H, 3 * H, H, B, s, q8 ? l.s_wqkv : nullptr, nullptr));
        PGV_TRY(pgv_launch_decode_attn(ctx, m->dtype, kv->qkv, kv->d_pos, m->rope, kv->Kc[li], kv->Vc[li], kv->ao, B, H, heads, kv->max_seq, kv_bytes, s, kv->dattn_part, kv->dattn_ticket));
        PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_RESID, q8 ? l.q_wo : l.wo, kv->ao, H, kv->resid, H, H, H, B, s, q8 ? l.s_wo : nullptr, nullptr));
        PGV_TRY(pgv_launch_rmsnorm(m->dtype, kv->resid, l.post_g, m->cfg.eps, kv->xn, B, H, s));
        PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_SWIGLU, q8 ? l.q_wgu : l.wgu, kv->xn, H, kv->act, I, 2 * I, H, B, s, q8 ? l.s_wgu : nullptr, nullptr));
        PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_RESID, q8 ? l.q_wdown : l.wdown, kv->act, I, kv->resid, H, H, I, B, s, q8 ? l.s_wdown : nullptr, nullptr));
    }
    PGV_TRY(pgv_launch_rmsnorm(m->dtype, kv->resid, m->norm_g, m->cfg.eps, kv->xn, B, H, s));
    PGV_TRY(lm_head_and_pick(ctx, m, kv, B, eos, flags, s, 0));
    return PGV_OK;
}

static int decode_enqueue(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, int eos, int flags, hipStream_t s) {
    if (!norm_fold_enabled()) return decode_enqueue_unfolded(ctx, m, kv, eos, flags, s);
    const int H = m->cfg.hidden, I = m->cfg.inter, heads = m->cfg.heads, B = kv->active, L = m->cfg.layers;
    const int nparts = H / 16;                     // one sum-of-squares partial per 16-row workgroup of a residual producer
    // RMSNorm has no launch of its own in decode (GemvArgs, gemv.hip): the kernel that completes the residual also writes
    // xg = round16(resid * gamma_next) and sum-of-squares partials; the consumer GEMV scales its accumulators by rstd.
    pgv_prof_begin(ctx, 6, s);
    PGV_TRY(pgv_launch_embed_tok_norm(m->dtype, kv->d_cur, m->embed, kv->resid, L > 0 ? m->layers[0].in_g : m->norm_g, kv->xn, kv->ssq, B, H, s, true));
    pgv_prof_end(ctx, 6, s, 0.0, 0.0);
    double kv_bytes = 0;
    for (int b = 0; b < B; ++b) kv_bytes += 2.0 * 2.0 * (double)(kv->h_len[b] + 1) * H;
    int parts_in = 1;                              // the embedding kernel leaves one partial per sequence
    for (int li = 0; li < L; ++li) {
        const LlmLayer& l = m->layers[li];
        const bool q8 = m->fp8;
        GemvNorm cons; cons.ssq_in = kv->ssq; cons.nparts_in = parts_in; cons.hidden = H; cons.eps = m->cfg.eps; cons.ssq_ts = kv->ssq_ts; cons.x_blocked = true;
        if (q8) PGV_TRY(gemv8(ctx, m, kv, GV_STORE16, l.q_wqkv, l.s_wqkv, kv->xn, H, kv->qkv, 3 * H, 3 * H, B, s, &cons));
        else PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_STORE16, l.wqkv, kv->xn, H, kv->qkv, 3 * H, 3 * H, H, B, s, nullptr, &cons));
        PGV_TRY(pgv_launch_decode_attn(ctx, m->dtype, kv->qkv, kv->d_pos, m->rope, kv->Kc[li], kv->Vc[li], kv->ao, B, H, heads, kv->max_seq, kv_bytes, s, kv->dattn_part, kv->dattn_ticket));
        GemvNorm prod; prod.gamma = l.post_g; prod.xg = kv->xn; prod.ssq_out = kv->ssq; prod.ssq_ts = kv->ssq_ts; prod.k8_part = kv->k8_part; prod.x_blocked = true;
        if (q8) PGV_TRY(gemv8(ctx, m, kv, GV_RESIDNORM, l.q_wo, l.s_wo, kv->ao, H, kv->resid, H, H, B, s, &prod));
        else PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_RESIDNORM, l.wo, kv->ao, H, kv->resid, H, H, H, B, s, nullptr, &prod));
        cons.nparts_in = nparts;
        if (q8) PGV_TRY(gemv8(ctx, m, kv, GV_SWIGLU, l.q_wgu, l.s_wgu, kv->xn, H, kv->act, I, 2 * I, B, s, &cons));
        else PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_SWIGLU, l.wgu, kv->xn, H, kv->act, I, 2 * I, H, B, s, nullptr, &cons));
        prod.gamma = li + 1 < L ? m->layers[li + 1].in_g : m->norm_g;
        if (q8) PGV_TRY(gemv8(ctx, m, kv, GV_RESIDNORM, l.q_wdown, l.s_wdown, kv->act, I, kv->resid, H, H, B, s, &prod));
        else PGV_TRY(pgv_launch_gemv(ctx, m->dtype, GV_RESIDNORM, l.wdown, kv->act, I, kv->resid, H, H, I, B, s, nullptr, &prod));
        parts_in = nparts;
    }
    PGV_TRY(lm_head_and_pick(ctx, m, kv, B, eos, flags, s, parts_in));
    return PGV_OK;
}

static bool graphs_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PGV_NO_GRAPH"); v = (e && e[0] == '1') ? 0 : 1; }
    return v == 1;
}

// Decode steps: eager the first time (loads code objects), then replayed from captured hipGraphs -- one graph holding kGraphSteps
// consecutive token steps (positions, current tokens and step counters live in device memory and are advanced by the pick kernel, so
// a graph is valid for any starting position) and one holding a single step for the remainder.  Per-family event timers need eager
// launches, so profiling disables the graphs.
constexpr int kGraphSteps = 8;

static int decode_graph(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, int eos, int flags, int which, int steps, hipStream_t s) {
    const int B = kv->active;
    if (kv->g_B != B || kv->g_eos != eos || kv->g_flags != flags || kv->g_gen != m->generation ||
        ((flags & AM_SAMPLE) && (kv->g_temp != kv->s_temp || kv->g_topk != kv->s_topk))) {
        for (auto& g : kv->gexec) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
        kv->g_B = B; kv->g_eos = eos; kv->g_flags = flags; kv->g_gen = m->generation; kv->g_temp = kv->s_temp; kv->g_topk = kv->s_topk;
    }
    if (!kv->gexec[which]) {
        // capture on a library-owned stream (the caller's may be the legacy default stream, which cannot be
        // captured); capture executes nothing, the instantiated graph is then launched on the caller's stream
        hipGraph_t graph = nullptr;
        if (!ctx->cap_stream) PGV_HIP(hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
        PGV_HIP(hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeRelaxed));
        int rc = PGV_OK;
        for (int i = 0; i < steps && rc == PGV_OK; ++i) rc = decode_enqueue(ctx, m, kv, eos, flags, ctx->cap_stream);
        hipError_t e = hipStreamEndCapture(ctx->cap_stream, &graph);
        if (rc != PGV_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess) { pgv_set_error("decode graph capture failed: %s", hipGetErrorString(e)); return PGV_EHIP; }
        e = hipGraphInstantiate(&kv->gexec[which], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e != hipSuccess) { kv->gexec[which] = nullptr; pgv_set_error("decode graph instantiate failed: %s", hipGetErrorString(e)); return PGV_EHIP; }
    }
    PGV_HIP(hipGraphLaunch(kv->gexec[which], s));
    return PGV_OK;
}

static int decode_steps(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, int eos, int flags, int n, hipStream_t s) {
    const int B = kv->active;
    while (n > 0) {
        int k = 1;
        if (ctx->prof || !graphs_enabled() || !kv->warmed) {
            kv->warmed = true;
            PGV_TRY(decode_enqueue(ctx, m, kv, eos, flags, s));
        } else if (n >= kGraphSteps) {
            k = kGraphSteps;
            PGV_TRY(decode_graph(ctx, m, kv, eos, flags, 1, kGraphSteps, s));
        } else {
            PGV_TRY(decode_graph(ctx, m, kv, eos, flags, 0, 1, s));
        }
        for (int b = 0; b < B; ++b) kv->h_len[b] += k;
        n -= k;
    }
    return PGV_OK;
}

static int decode_step(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, int eos, int flags, hipStream_t s) { return decode_steps(ctx, m, kv, eos, flags, 1, s); }

static int check_decode(pgv_llm* m, pgv_kv* kv, int steps, const char* who) {
    PGV_CHECK(m && kv && kv->llm == m, "%s: bad model / cache", who);
    if (kv->active < 1) { pgv_set_error("%s: no prefilled sequences in this cache", who); return PGV_ESTATE; }
    for (int b = 0; b < kv->active; ++b)
        PGV_CHECK(kv->h_len[b] + steps <= kv->max_seq, "%s: sequence %d would grow to %d tokens (cache holds %d)", who, b, kv->h_len[b] + steps, kv->max_seq);
    return PGV_OK;
}

extern "C" int pgv_llm_decode(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, const int32_t* d_last, float* d_logits, int32_t* d_next, void* stream) {
    PGV_CHECK(ctx && d_last, "pgv_llm_decode: null argument");
    PGV_TRY(check_decode(m, kv, 1, "pgv_llm_decode"));
    hipStream_t s = (hipStream_t)stream;
    const int B = kv->active;
    if (d_last != kv->d_cur) PGV_HIP(hipMemcpyAsync(kv->d_cur, d_last, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    PGV_TRY(decode_step(ctx, m, kv, -1, AM_INC_POS, s));
    if (d_logits) PGV_HIP(hipMemcpyAsync(d_logits, kv->logits, (size_t)B * m->cfg.vocab * 4, hipMemcpyDeviceToDevice, s));
    if (d_next) PGV_HIP(hipMemcpyAsync(d_next, kv->d_cur, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

extern "C" int pgv_llm_decode_greedy(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, const int32_t* d_first, int n, int eos_id, int32_t* d_tokens, void* stream) {
    PGV_CHECK(ctx && d_first && d_tokens, "pgv_llm_decode_greedy: null argument");
    PGV_CHECK(n >= 1, "pgv_llm_decode_greedy: n must be positive");
    PGV_TRY(check_decode(m, kv, n, "pgv_llm_decode_greedy"));
    hipStream_t s = (hipStream_t)stream;
    const int B = kv->active;
    if (d_first != kv->d_cur) PGV_HIP(hipMemcpyAsync(kv->d_cur, d_first, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipMemsetAsync(kv->d_step, 0, (size_t)kv->B * 4, s));
    PGV_TRY(decode_steps(ctx, m, kv, eos_id, AM_INC_POS | AM_RECORD, n, s));
    PGV_HIP(hipMemcpy2DAsync(d_tokens, (size_t)n * 4, kv->d_hist, (size_t)kv->max_seq * 4, (size_t)n * 4, B, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

extern "C" int pgv_llm_sample(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, float temperature, int top_k, const float* d_u, int32_t* d_next, void* stream) {
    PGV_CHECK(ctx && d_u, "pgv_llm_sample: null argument");
    PGV_CHECK(m && kv && kv->llm == m, "pgv_llm_sample: bad model / cache");
    if (kv->active < 1) { pgv_set_error("pgv_llm_sample: no prefilled sequences in this cache"); return PGV_ESTATE; }
    hipStream_t s = (hipStream_t)stream;
    const int B = kv->active;
    PGV_TRY(pgv_launch_sample(kv->logits, m->cfg.vocab, B, temperature, top_k, d_u, B, 0, kv->d_cur, kv->d_pos, kv->d_step, kv->d_hist, kv->max_seq, kv->d_done,
                              -1, 0, s));
    if (d_next) PGV_HIP(hipMemcpyAsync(d_next, kv->d_cur, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

extern "C" int pgv_llm_decode_sample(pgv_ctx* ctx, pgv_llm* m, pgv_kv* kv, const int32_t* d_first, int n, int eos_id, float temperature, int top_k,
                                     const float* d_u, int32_t* d_tokens, void* stream) {
    PGV_CHECK(ctx && d_first && d_tokens && d_u, "pgv_llm_decode_sample: null argument");
    PGV_CHECK(n >= 1, "pgv_llm_decode_sample: n must be positive");
    PGV_CHECK(temperature > 0.f, "pgv_llm_decode_sample: temperature must be positive (got %g)", (double)temperature);
    PGV_TRY(check_decode(m, kv, n, "pgv_llm_decode_sample"));
    hipStream_t s = (hipStream_t)stream;
    const int B = kv->active;
    if (d_first != kv->d_cur) PGV_HIP(hipMemcpyAsync(kv->d_cur, d_first, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipMemsetAsync(kv->d_step, 0, (size_t)kv->B * 4, s));
    PGV_HIP(hipMemcpyAsync(kv->d_u, d_u, (size_t)n * B * 4, hipMemcpyDeviceToDevice, s));     // n <= max_seq (check_decode)
    kv->s_temp = temperature; kv->s_topk = top_k;
    PGV_TRY(decode_steps(ctx, m, kv, eos_id, AM_INC_POS | AM_RECORD | AM_SAMPLE, n, s));
    PGV_HIP(hipMemcpy2DAsync(d_tokens, (size_t)n * 4, kv->d_hist, (size_t)kv->max_seq * 4, (size_t)n * 4, B, hipMemcpyDeviceToDevice, s));
    PGV_HIP(hipGetLastError());
    return PGV_OK;
}

// ---------------------------------------------------------------------------------------------
// mm_projector
// ---------------------------------------------------------------------------------------------
extern "C" int pgv_projector(pgv_ctx* ctx, int dtype, int depth, const void* const* d_weights, const float* const* d_biases, int mm_hidden,
                             int hidden, const void* d_x, int rows, void* d_y, void* stream) {
    PGV_CHECK(ctx && d_weights && d_biases && d_x && d_y, "pgv_projector: null argument");
    PGV_CHECK(depth >= 1 && depth <= 8, "pgv_projector: depth %d unsupported (identity has nothing to run)", depth);
    PGV_CHECK(rows > 0, "pgv_projector: rows must be positive");
    hipStream_t s = (hipStream_t)stream;
    void* tmp[2] = {nullptr, nullptr};
    if (depth > 1) {
        PGV_TRY(pgv_ws_reserve(ctx, 2 * pgv_align((size_t)rows * hidden * 2), s));
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
    if (depth > 1) PGV_TRY(pgv_ws_release(ctx, s));
    return PGV_OK;
}
