/*
 * pgv.h -- C ABI of libpgv.so: the MI355X (gfx950) hot path of PG-Video-LLaVA.
 *
 * The reference (mbzuai-oryx/Video-LLaVA, package `video_chatgpt`) has no FFI layer: its
 * boundary is Python objects handed between `initialize_model` and `video_chatgpt_infer`
 * (SURVEY.md 8b).  This header is the C boundary placed *under* those Python contracts; each
 * entry point cites the reference call site (path:line under /root/reference) it replaces.
 * The Python mirror in video_llava_amd/ binds these symbols with ctypes.
 *
 * Conventions
 *  - plain C, no C++/torch types; every pointer named `d_*` is a DEVICE pointer owned by the
 *    caller (the library never frees caller memory); `h_*` is a HOST pointer.
 *  - every compute call is enqueued on the caller's `stream` (a hipStream_t passed as void*,
 *    e.g. torch.cuda.current_stream().cuda_stream) and returns without synchronising.
 *  - return value: 0 = ok, otherwise a PGV_E* code; pgv_last_error() gives the message for the
 *    calling thread.  No exceptions cross the ABI.
 *  - one pgv_ctx per GPU/rank; a ctx is not thread-safe (one host thread per rank).
 *  - activations/weights are 16-bit (PGV_F16 or PGV_BF16, chosen per model handle), all
 *    accumulation, normalisation statistics, softmax and the residual stream are fp32.
 */
#ifndef PGV_H
#define PGV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGV_VERSION 310

enum { PGV_OK = 0, PGV_EINVAL = 1, PGV_EHIP = 2, PGV_ENOMEM = 3, PGV_ESTATE = 4, PGV_ENAME = 5 };
enum { PGV_F16 = 0, PGV_BF16 = 1, PGV_F32 = 2 };

typedef struct pgv_ctx pgv_ctx;   /* device + workspace arena                         */
typedef struct pgv_vit pgv_vit;   /* packed CLIP vision tower weights                 */
typedef struct pgv_llm pgv_llm;   /* packed mm_projector + LLaMA decoder weights      */
typedef struct pgv_kv pgv_kv;     /* KV cache + decode state for a batch of sequences */

int pgv_version(void);
const char *pgv_last_error(void);

/* ---- context -------------------------------------------------------------------------------- */
int pgv_ctx_create(int device, pgv_ctx **out);
void pgv_ctx_destroy(pgv_ctx *ctx);
/* bytes currently held by the workspace arena (grows on demand, never inside a captured region) */
size_t pgv_ctx_workspace_bytes(const pgv_ctx *ctx);

/* Per-kernel-family device timers (hipEvent pairs around every launch of the family on the
 * caller's stream).  family: 0 = MFMA GEMM, 1 = ViT attention, 2 = LLM prefill attention,
 * 3 = decode GEMV, 4 = decode attention, 5 = pooling, 6 = the short decode kernels (residual+RMSNorm, embedding, argmax).
 * Off by default. */
#define PGV_NFAMILY 7
int pgv_prof_enable(pgv_ctx *ctx, int on);
int pgv_prof_reset(pgv_ctx *ctx);
/* synchronises the recorded events; fills launches / total ms / algorithmic flops / algorithmic bytes */
int pgv_prof_get(pgv_ctx *ctx, int family, int64_t *launches, double *ms, double *flops, double *bytes);
/* mean elapsed ms of an EMPTY event pair on `stream`: the fixed per-launch cost included in pgv_prof_get's totals */
int pgv_prof_calibrate(pgv_ctx *ctx, void *stream, int n, double *ms_per_pair);

/* ---- CLIP vision tower ------------------------------------------------------------------------
 * Replaces `vision_tower(image_tensor, output_hidden_states=True)` -- video_chatgpt/inference.py:93,
 * video_chatgpt/chat.py:140, scripts/save_spatio_temporal_clip_features.py:116 (HF CLIPVisionModel). */
typedef struct {
    int hidden;      /* 1024 */
    int inter;       /* 4096 */
    int layers;      /* 24   */
    int heads;       /* 16 (head_dim must be 64) */
    int image;       /* 224 or 336 */
    int patch;       /* 14 */
    float eps;       /* 1e-5 */
} pgv_vit_config;

int pgv_vit_create(pgv_ctx *ctx, const pgv_vit_config *cfg, int dtype, pgv_vit **out);
void pgv_vit_destroy(pgv_vit *vit);
/* Load one tensor by its HF state-dict name ("vision_model.encoder.layers.3.mlp.fc1.weight" ...;
 * the "vision_model." prefix is optional).  `data` is contiguous in `src_dtype`, on the device if
 * `on_device` else on the host.  The library converts/re-lays it out into its packed storage
 * (fused qkv, K-padded patch filter, fp32 biases and norm parameters).  `numel` is the element count of `data`: a tensor whose size
 * disagrees with the shape the config implies is rejected with PGV_EINVAL (torch's "size mismatch"), never read past its end.  Replaces
 * CLIPVisionModel.from_pretrained(...) at video_chatgpt/eval/model_utils.py:134-136. */
int pgv_vit_load_tensor(pgv_vit *vit, const char *name, const void *data, int src_dtype, int on_device, int64_t numel, void *stream);
/* number of tensors still missing (0 = ready) */
int pgv_vit_missing(const pgv_vit *vit);

/* Frame preprocessing fused for the GPU: uint8 RGB frames [T, image, image, 3] (already at the crop
 * size) -> (x/255 - mean)/std in NCHW 16-bit, i.e. CLIPImageProcessor.preprocess(...).half()
 * (video_chatgpt/inference.py:86-89) for crop-sized input. */
int pgv_preprocess_u8(pgv_ctx *ctx, const uint8_t *d_frames, int T, int image, int dtype, void *d_pixels, void *stream);
/* Frame ingest at native resolution: uint8 RGB frames [T, H, W, 3] as the video decoder delivers them -> nearest
 * resize to image x image exactly as load_video does it (video_chatgpt/eval/model_utils.py:38-43: permute, .float(),
 * F.interpolate(size) with the default mode 'nearest' -> source index min(floorf(dst * (float)in / out), in - 1) per axis, cast back
 * to uint8) -> CLIPImageProcessor normalisation -> NCHW 16-bit, in one pass.  H == W == image reduces to pgv_preprocess_u8. */
int pgv_ingest_u8(pgv_ctx *ctx, const uint8_t *d_frames, int T, int H, int W, int image, int dtype, void *d_pixels, void *stream);

/* hidden_states[k] of the tower for k = n_layers (k=0: pre-LayerNorm'ed embeddings):
 * d_pixels [T,3,image,image] 16-bit NCHW -> d_hidden [T, patches+1, hidden] 16-bit.
 * The reference selects hidden_states[-2] => n_layers = layers-1 (inference.py:94); only the
 * layers that feed the selected state are executed. */
int pgv_vit_forward(pgv_ctx *ctx, pgv_vit *vit, const void *d_pixels, int T, int n_layers, void *d_hidden, void *stream);

/* ---- spatio-temporal pooling -------------------------------------------------------------------
 * Replaces get_spatio_temporal_features_torch (video_chatgpt/inference.py:13-44; chat.py:77-87) and
 * the numpy twin get_spatio_temporal_features (scripts/save_spatio_temporal_clip_features.py:46-57).
 * d_feats: T frames of P patch rows x C channels, 16-bit, row stride C, frame stride
 * `frame_stride` elements (so the `[:, 1:]` view of a [T,P+1,C] tensor is consumed without a copy).
 * d_out [n_temporal + P, C]: rows [0,T) = per-frame mean over patches, rows [T,n_temporal) = 0,
 * rows [n_temporal, n_temporal+P) = per-patch mean over frames.  fp32 accumulation, one rounding.
 * T > n_temporal is rejected (the reference never truncates; its callers cap T at 100). */
int pgv_st_pool(pgv_ctx *ctx, const void *d_feats, int in_dtype, int T, int P, int C, int64_t frame_stride,
                int n_temporal, void *d_out, int out_dtype, void *stream);

/* ---- mm_projector ------------------------------------------------------------------------------
 * Replaces `self.mm_projector(video_spatio_temporal_features)` (video_chatgpt/model/video_chatgpt.py:105):
 * a bare nn.Linear(mm_hidden, hidden) for 224-px towers (:52-53) or build_vision_projector's
 * Linear + (depth-1) x [GELU(erf), Linear] (multimodal_projector/builder.py:33-50).  The parameters stay
 * owned by the caller's nn.Module (so `mm_projector.bin` loads by name, eval/model_utils.py:122-127):
 * d_weights[i] is layer i's [out, in] 16-bit row-major weight, d_biases[i] its fp32 bias.
 * d_x [rows, mm_hidden] -> d_y [rows, hidden]; depth 0 (identity) is rejected (nothing to run). */
int pgv_projector(pgv_ctx *ctx, int dtype, int depth, const void *const *d_weights, const float *const *d_biases,
                  int mm_hidden, int hidden, const void *d_x, int rows, void *d_y, void *stream);

/* ---- LLaMA decoder --------------------------------------------------------------------------------
 * Replaces VideoChatGPTLlamaForCausalLM (video_chatgpt/model/video_chatgpt.py:178-325) as driven by
 * model.generate at video_chatgpt/inference.py:105-112 / chat.py:148-154. */
typedef struct {
    int vocab;            /* 32003 */
    int hidden;           /* 4096 / 5120 */
    int inter;            /* 11008 / 13824 */
    int layers;           /* 32 / 40 */
    int heads;            /* 32 / 40 (MHA; head_dim must be 128) */
    float eps;            /* rms_norm_eps */
    float rope_theta;     /* 10000 */
} pgv_llm_config;

int pgv_llm_create(pgv_ctx *ctx, const pgv_llm_config *cfg, int dtype, pgv_llm **out);
void pgv_llm_destroy(pgv_llm *llm);
/* HF names: "model.embed_tokens.weight", "model.layers.N.self_attn.q_proj.weight", ...,
 * "model.norm.weight", "lm_head.weight".  `model.embed_tokens.weight` may be re-loaded later (the
 * projector checkpoint carries the rows of the added video tokens, train/llava_trainer.py:34). */
int pgv_llm_load_tensor(pgv_llm *llm, const char *name, const void *data, int src_dtype, int on_device, int64_t numel, void *stream);
int pgv_llm_missing(const pgv_llm *llm);
/* Load `nrows` rows starting at `row0` of "model.embed_tokens.weight" or "lm_head.weight" (checkpoints whose
 * vocabulary is smaller than the resized model, eval/model_utils.py:119-127). */
int pgv_llm_load_rows(pgv_llm *llm, const char *name, const void *data, int src_dtype, int on_device, int row0, int nrows, int64_t numel, void *stream);
/* model.resize_token_embeddings(n) (eval/model_utils.py:119): the handle is allocated with 64 spare vocabulary
 * rows; growing within them zero-fills the new embed / lm_head rows (the projector checkpoint then overwrites them). */
int pgv_llm_resize_vocab(pgv_llm *llm, int new_vocab, void *stream);
int pgv_llm_vocab(const pgv_llm *llm);
/* fp8 weight path (BASELINE config 5, "13B, fp8 weight path"): after every tensor is loaded, quantise all decoder matrices and
 * lm_head to OCP e4m3 with one power-of-two scale per output row.  The decode GEMVs then stream the fp8 copies (half the
 * bytes per token); the 16-bit copies are overwritten with the dequantised values (exactly representable), so prefill, decode
 * and a CPU oracle fed with pgv_llm_get_weight() all compute with the same weights.  Replaces nothing in the reference (it
 * has no quantised path); idempotent. */
int pgv_llm_quantize_fp8(pgv_ctx *ctx, pgv_llm *llm, void *stream);
int pgv_llm_is_fp8(const pgv_llm *llm);
/* Read a decoder matrix ("model.layers.N.{self_attn.{q,k,v,o}_proj,mlp.{gate,up,down}_proj}.weight", "lm_head.weight") back as
 * row-major fp32 [rows, cols] in its HF orientation: the values the path computes with (after quantisation: dequantised). */
int pgv_llm_get_weight(pgv_ctx *ctx, pgv_llm *llm, const char *name, float *d_out, void *stream);

int pgv_kv_create(pgv_ctx *ctx, pgv_llm *llm, int batch, int max_seq, pgv_kv **out);
void pgv_kv_destroy(pgv_kv *kv);

/* Prefill (the `input_ids.shape[1] != 1` branch, video_chatgpt/model/video_chatgpt.py:100-175 and
 * :225-226): embed + splice projected video rows + decoder stack + lm_head on the LAST position of
 * each sequence.  Sequences are ragged: h_seq_lens[b] tokens each, concatenated in h_ids (host: the
 * tokenizer output never lives on the device, and the row map of the splice is built from it).
 * d_video [B, V, hidden] = mm_projector output, 16-bit (or NULL); h_vid_pos[b] = index of <vid_start>
 * in sequence b, or -1 for a text-only sample (rows (pos, pos+V] are replaced -- the caller has already
 * validated the placeholder run on the host, raising the reference's ValueErrors).
 * Outputs (any may be NULL): d_logits [B, vocab] fp32, d_next [B] int32 greedy argmax, and -- what
 * VideoChatGPTLlamaForCausalLM.forward returns (:225-226, `logits = self.lm_head(hidden_states)` over ALL positions) --
 * d_all_logits [sum of h_seq_lens, ld_all] fp32 with ld_all >= vocab rounded up to a multiple of 16 (row r = token r of the
 * concatenated batch; columns >= vocab hold the zero spare rows of lm_head). */
int pgv_llm_prefill(pgv_ctx *ctx, pgv_llm *llm, pgv_kv *kv, const int32_t *h_ids, const int32_t *h_seq_lens, int B,
                    const void *d_video, int V, const int32_t *h_vid_pos, float *d_logits, int32_t *d_next, float *d_all_logits,
                    int ld_all, void *stream);

/* Prefill of NEW tokens behind the ones already in the cache: VideoChatGPTLlamaForCausalLM.forward called with `past_key_values` and
 * input_ids.shape[1] > 1 (video_chatgpt/model/video_chatgpt.py:193-251 accepts any input_ids next to a cache; :103 still splices when the new
 * ids carry a placeholder run) -- what a second chat turn is (video_chatgpt/chat.py:108-160 re-tokenises and re-runs the WHOLE conversation
 * every turn although the prefix is unchanged).  Same arguments as pgv_llm_prefill; B must equal the batch of the prefill that filled `kv`;
 * row p of sequence b takes position pgv_kv_len(kv, b) + p, attends to the cached prefix and the new rows before it, and is appended.
 * h_vid_pos is relative to the NEW rows.  Outputs as pgv_llm_prefill (d_all_logits: the new rows only).  Given the same cache contents an
 * appended row is bitwise the row of one full prefill over prefix + new tokens (a prefix that pgv_llm_prefill wrote reproduces the one-call
 * result exactly; entries written by decode steps carry that path's rounding). */
int pgv_llm_prefill_append(pgv_ctx *ctx, pgv_llm *llm, pgv_kv *kv, const int32_t *h_ids, const int32_t *h_seq_lens, int B,
                           const void *d_video, int V, const int32_t *h_vid_pos, float *d_logits, int32_t *d_next, float *d_all_logits,
                           int ld_all, void *stream);
/* Forget the cache entries of sequence b from position `len` on (1 <= len <= pgv_kv_len): the next append / decode call continues there.
 * Used when a new chat turn shares only a prefix with what the cache holds (surplus decode steps past a stop string,
 * video_chatgpt/model/utils.py:6-26). */
int pgv_kv_truncate(pgv_kv *kv, int b, int len, void *stream);

/* One decode step for all B sequences (the `input_ids.shape[1] == 1` branch, :103): consumes
 * d_last [B] token ids, appends to the KV cache, writes d_logits [B, vocab] and/or d_next [B]. */
int pgv_llm_decode(pgv_ctx *ctx, pgv_llm *llm, pgv_kv *kv, const int32_t *d_last, float *d_logits, int32_t *d_next, void *stream);

/* n greedy steps without host round trips: step i consumes the previous argmax (step 0 consumes
 * d_first [B]); token of step i of sequence b is written to d_tokens[b * n + i].  A sequence that
 * emits eos_id (>= 0) keeps emitting eos_id.  Replaces the per-token loop of GenerationMixin for
 * do_sample=False. */
int pgv_llm_decode_greedy(pgv_ctx *ctx, pgv_llm *llm, pgv_kv *kv, const int32_t *d_first, int n, int eos_id,
                          int32_t *d_tokens, void *stream);

/* Sampling (the reference's DEFAULT decode mode: model.generate(do_sample=True, temperature=0.2), video_chatgpt/inference.py:106-112;
 * HF's sample loop = TemperatureLogitsWarper, TopKLogitsWarper(top_k=50 from the default GenerationConfig), softmax, multinomial).
 * The multinomial draw is an inverse-CDF pick with a caller-supplied uniform u in [0,1): token = the first vocabulary index whose
 * cumulative probability exceeds u; probabilities = softmax(logits / temperature) over the top_k largest logits (top_k <= 0: all of
 * them; ties at the k-th value are kept, as HF's `scores < kth` mask does).  The token stays on the device.
 * pgv_llm_sample: pick from the logits of the last prefill / decode call of `kv` with d_u [B]; the result replaces the greedy pick
 * as the cache's current token and is written to d_next [B] (may be NULL). */
int pgv_llm_sample(pgv_ctx *ctx, pgv_llm *llm, pgv_kv *kv, float temperature, int top_k, const float *d_u, int32_t *d_next, void *stream);
/* n sampled steps without host round trips (twin of pgv_llm_decode_greedy): step i draws with d_u[i * B + b]. */
int pgv_llm_decode_sample(pgv_ctx *ctx, pgv_llm *llm, pgv_kv *kv, const int32_t *d_first, int n, int eos_id, float temperature, int top_k,
                          const float *d_u, int32_t *d_tokens, void *stream);
/* building block: the same pick on caller logits [B, V] fp32 (row stride V) */
int pgv_sample_logits(pgv_ctx *ctx, const float *d_logits, int V, int B, float temperature, int top_k, const float *d_u, int32_t *d_next,
                      void *stream);

/* current length (tokens in cache) of sequence b, host-side bookkeeping */
int pgv_kv_len(const pgv_kv *kv, int b);

/* ---- building blocks exported for unit parity tests (same kernels the calls above use) -------- */
/* C[M,N] = A[M,K] * W[N,K]^T (+bias) with epilogue `epi` (see pgv_epi); 16-bit in, fp32 accumulate. */
enum pgv_epi {
    PGV_EPI_NONE = 0,        /* C16 = acc                        */
    PGV_EPI_BIAS = 1,        /* C16 = acc + bias                 */
    PGV_EPI_BIAS_QGELU = 2,  /* C16 = quick_gelu(acc + bias)     */
    PGV_EPI_BIAS_GELU = 3,   /* C16 = gelu_erf(acc + bias)       */
    PGV_EPI_RESID = 4,       /* R32 += acc                       */
    PGV_EPI_BIAS_RESID = 5,  /* R32 += acc + bias                */
    PGV_EPI_SWIGLU = 6,      /* C16[:, n/2] = silu(gate)*up, W rows interleaved gate/up in blocks of 32 */
    PGV_EPI_F32 = 7          /* C32 = acc (+bias if given)       */
};
int pgv_gemm(pgv_ctx *ctx, int dtype, int epi, const void *d_A, int lda, const void *d_W, int ldw, const float *d_bias,
             void *d_C, int ldc, int M, int N, int K, void *stream);
/* CLIP self-attention on a fused qkv buffer [T*N, 3C] (q | k | v column blocks) -> [T*N, C]; head_dim 64, no mask
 * (HF CLIPAttention eager math, HF:clip/modeling_clip.py:259-277). */
int pgv_vit_attention(pgv_ctx *ctx, int dtype, const void *d_qkv, void *d_out, int T, int N, int C, int heads, void *stream);
/* Decode-time projection y[B,N] = x[B,K] W[N,K]^T for B <= 64 (weights streamed once; 16 sequences per MFMA column tile).  mode: 0 = 16-bit out,
 * 1 = fp32 residual accumulate, 2 = SwiGLU (W rows interleaved [32 gate | 32 up], out [B, N/2]), 3 = fp32 out.
 * d_W is in the fragment-blocked layout produced by pgv_pack_blocked (rows padded to a multiple of 16). */
int pgv_gemv(pgv_ctx *ctx, int dtype, int mode, const void *d_W, const void *d_x, int ldx, void *d_out, int ldo, int N, int K,
             int B, void *stream);
/* Re-lay a row-major [rows, cols] 16-bit matrix (cols % 32 == 0) into the fragment-blocked layout
 * [rows/16][cols/32][4 k-groups][16 rows][8 elems]: each 1 KiB block is one v_mfma_f32_16x16x32 A fragment in lane order.
 * d_dst must hold ceil(rows/16)*16 * cols elements (zero it first when rows % 16 != 0). */
int pgv_pack_blocked(pgv_ctx *ctx, int dtype, const void *d_src, int rows, int cols, void *d_dst, void *stream);
/* Inverse of pgv_pack_blocked: blocked 16-bit [N, K] (N % 16 == 0, K % 32 == 0) -> row-major fp32. */
int pgv_unpack_blocked(pgv_ctx *ctx, int dtype, const void *d_src_blocked, float *d_dst, int N, int K, void *stream);
/* Quantise a blocked 16-bit matrix [N, K] (N % 16 == 0, K % 64 == 0) to e4m3 with per-row power-of-two scales: writes the fp8
 * blocked copy (N*K bytes: block (n/16, k/64) = 1 KiB, lane ((k%32)/8)*16 + n%16, byte ((k/32)%2)*8 + k%8), the scales [N], and
 * overwrites the 16-bit matrix with the dequantised values. */
int pgv_quantize_fp8_blocked(pgv_ctx *ctx, int dtype, void *d_w16_blocked, void *d_w8_blocked, float *d_scales, int N, int K, void *stream);
/* pgv_gemv on the fp8 copy: bit-identical to pgv_gemv on the dequantised 16-bit matrix, half the weight bytes.  modes 0..3. */
int pgv_gemv_fp8(pgv_ctx *ctx, int dtype, int mode, const void *d_W8, const float *d_scales, const void *d_x, int ldx, void *d_out, int ldo,
                 int N, int K, int B, void *stream);
int pgv_layernorm(pgv_ctx *ctx, int dtype, const float *d_x, const float *d_gamma, const float *d_beta, float eps,
                  void *d_y, int rows, int cols, void *stream);
int pgv_rmsnorm(pgv_ctx *ctx, int dtype, const float *d_x, const float *d_gamma, float eps, void *d_y, int rows, int cols,
                void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PGV_H */
