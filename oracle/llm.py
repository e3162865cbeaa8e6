"""Oracle: VideoChatGPTLlamaForCausalLM forward (splice + LLaMA decoder) and greedy decode.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain torch on CPU in fp32 or fp64.
"""
from __future__ import annotations

import numpy as np
import torch

from .synth import LlamaCfg
from .vision import mm_projector


def _t(w: dict, key: str, dtype) -> torch.Tensor:
    v = w[key]
    return (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).to(dtype)


class _WeightCache(dict):
    """Lazily converted copy of a state dict (see LlamaOracle.__init__)."""

    def __init__(self, src: dict, dtype):
        super().__init__()
        self._src, self._dtype = src, dtype

    def __missing__(self, key):
        v = self._src[key]
        t = (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).to(self._dtype)
        self[key] = t
        return t

    def __contains__(self, key):
        return key in self._src


def rms_norm(x: torch.Tensor, g: torch.Tensor, eps: float) -> torch.Tensor:
    """LlamaRMSNorm.forward (HF:llama/modeling_llama.py:61-67): x * rsqrt(mean(x^2) + eps) * weight."""
    var = x.pow(2).mean(-1, keepdim=True)
    return g * (x * torch.rsqrt(var + eps))


def rope_cos_sin(positions: torch.Tensor, head_dim: int, theta: float, dtype) -> tuple[torch.Tensor, torch.Tensor]:
    """LlamaRotaryEmbedding (HF:llama/modeling_llama.py:96-126): inv_freq_i = theta^(-2i/d) for
    i < d/2, angle = pos * inv_freq computed in fp32, the d/2 angles repeated twice along the
    feature axis (cat(freqs, freqs)), cos/sin then cast to the activation dtype."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = positions.to(torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([ang, ang], dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """apply_rotary_pos_emb + rotate_half (HF:llama/modeling_llama.py:129-160): half-split pairing
    (feature j pairs with j + d/2), x*cos + cat(-x2, x1)*sin.  x [heads, S, d]; cos/sin [S, d]."""
    d = x.shape[-1]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    return x * cos + torch.cat([-x2, x1], dim=-1) * sin


def splice_video_embeddings(input_ids: torch.Tensor, embeds: torch.Tensor, video_feats: torch.Tensor,
                            vid_start: int, vid_end: int, vid_patch: int) -> torch.Tensor:
    """The use_vid_start_end branch of VideoChatGPTLlamaModel.forward
    (video_chatgpt/model/video_chatgpt.py:110-146) for one sample: rows (pos, pos+V] after the
    single <vid_start> at `pos` are replaced by the V projected video rows; <vid_end> must sit at
    pos+V+1 and the start/end counts must agree, else ValueError with the reference's messages.
    A sample without any <vid_patch> is returned unchanged (:113-118, the dummy term is 0)."""
    if int((input_ids == vid_patch).sum()) == 0:
        return embeds
    if vid_start is None:
        # use_vid_start_end = False (video_chatgpt/model/video_chatgpt.py:147-167): the <vid_patch> run itself is replaced; its length
        # must equal the number of video rows and it must be consecutive, else ValueError with the reference's messages.
        V = video_feats.shape[0]
        if int((input_ids == vid_patch).sum()) != V:
            raise ValueError("The number of video patch tokens should be the same as the number of video patches.")
        idx = torch.where(input_ids == vid_patch)[0]
        s0 = int(idx[0])
        if bool((idx != torch.arange(s0, s0 + V)).any()):
            raise ValueError("The video patch tokens should be consecutive.")
        return torch.cat([embeds[:s0], video_feats.to(embeds.dtype), embeds[s0 + V:]], dim=0)
    if int((input_ids == vid_start).sum()) != int((input_ids == vid_end).sum()):
        raise ValueError("The number of video start tokens and video end tokens should be the same.")
    starts = torch.where(input_ids == vid_start)[0]
    V = video_feats.shape[0]
    out = embeds
    for pos in starts.tolist():
        if pos + V + 1 >= input_ids.shape[0] or int(input_ids[pos + V + 1]) != vid_end:
            raise ValueError("The video end token should follow the video start token.")
        out = torch.cat([embeds[: pos + 1], video_feats.to(embeds.dtype), embeds[pos + V + 1:]], dim=0)
    return out


class LlamaOracle:
    """Single-sequence KV-cached decoder.  `prefill` == VideoChatGPTLlamaForCausalLM.forward on
    the full prompt (video_chatgpt/model/video_chatgpt.py:193-251 -> :82-175 ->
    HF LlamaModel.forward, HF:llama/modeling_llama.py:347-418); `step` == the same forward with
    input_ids.shape[1]==1, which skips the splice (:103)."""

    def __init__(self, w: dict, cfg: LlamaCfg, dtype=torch.float32, cache_weights: bool = False):
        """`w` may hold fp32 numpy arrays or 16-bit torch tensors (full-size checkpoints): every use converts to `dtype`;
        cache_weights=True keeps the converted copies (7B fp32 = 27 GB) instead of converting per forward."""
        self.w, self.cfg, self.dtype = (_WeightCache(w, dtype) if cache_weights else w), cfg, dtype
        self.k: list[torch.Tensor] = []
        self.v: list[torch.Tensor] = []
        self.pos = 0

    def reset(self):
        self.k, self.v, self.pos = [], [], 0

    def _layer(self, i: int, x: torch.Tensor, cos, sin) -> torch.Tensor:
        cfg, w, dt = self.cfg, self.w, self.dtype
        q = f"model.layers.{i}."
        S = x.shape[0]
        nh, hd = cfg.heads, cfg.head_dim
        h = rms_norm(x, _t(w, q + "input_layernorm.weight", dt), cfg.eps)
        qs = (h @ _t(w, q + "self_attn.q_proj.weight", dt).t()).view(S, nh, hd).transpose(0, 1)
        ks = (h @ _t(w, q + "self_attn.k_proj.weight", dt).t()).view(S, nh, hd).transpose(0, 1)
        vs = (h @ _t(w, q + "self_attn.v_proj.weight", dt).t()).view(S, nh, hd).transpose(0, 1)
        qs, ks = apply_rope(qs, cos, sin), apply_rope(ks, cos, sin)
        if len(self.k) <= i:
            self.k.append(ks); self.v.append(vs)
        else:
            self.k[i] = torch.cat([self.k[i], ks], dim=1); self.v[i] = torch.cat([self.v[i], vs], dim=1)
        K, V = self.k[i], self.v[i]
        Skv = K.shape[1]
        # eager_attention_forward (HF:llama/modeling_llama.py:191-214): scale, causal mask, softmax.
        sc = (qs @ K.transpose(-1, -2)) * (hd ** -0.5)
        qpos = torch.arange(Skv - S, Skv)[:, None]
        kpos = torch.arange(Skv)[None, :]
        sc = sc.masked_fill(kpos > qpos, float("-inf"))
        att = torch.softmax(sc, dim=-1)
        o = (att @ V).transpose(0, 1).reshape(S, nh * hd)
        x = x + o @ _t(w, q + "self_attn.o_proj.weight", dt).t()
        h = rms_norm(x, _t(w, q + "post_attention_layernorm.weight", dt), cfg.eps)
        g = h @ _t(w, q + "mlp.gate_proj.weight", dt).t()
        u = h @ _t(w, q + "mlp.up_proj.weight", dt).t()
        # LlamaMLP (HF:llama/modeling_llama.py:174-176): down(silu(gate) * up)
        x = x + (torch.nn.functional.silu(g) * u) @ _t(w, q + "mlp.down_proj.weight", dt).t()
        return x

    def _forward_embeds(self, x: torch.Tensor, all_logits: bool = False) -> torch.Tensor:
        cfg, w, dt = self.cfg, self.w, self.dtype
        S = x.shape[0]
        cos, sin = rope_cos_sin(torch.arange(self.pos, self.pos + S), cfg.head_dim, cfg.rope_theta, dt)
        for i in range(cfg.layers):
            x = self._layer(i, x, cos, sin)
        self.pos += S
        x = rms_norm(x, _t(w, "model.norm.weight", dt), cfg.eps)
        if all_logits is True:
            pass
        elif all_logits:                                   # an int n: the last n positions only (a long causal pass judged at its tail)
            x = x[-int(all_logits):]
        else:
            x = x[-1:]
        return x @ _t(w, "lm_head.weight", dt).t()

    def prefill(self, input_ids, video_feats: torch.Tensor | None, vid_start: int, vid_end: int,
                vid_patch: int, all_logits: bool = False, n_prompt: int | None = None) -> torch.Tensor:
        """input_ids [S] ints; video_feats [V, 1024] pooled CLIP features or None.
        Returns logits [1, vocab] for the last position (all_logits=True: [S, vocab]; all_logits=n: the last n positions).
        n_prompt: only the first n_prompt ids are the PROMPT (checked and spliced); the rest are tokens that followed it one at a time -- the
        reference's forward with input_ids.shape[1] == 1 skips the splice and the placeholder checks (model/video_chatgpt.py:103), so a generated
        <vid_start> is an ordinary embedding row there -- evaluated here in the same causal pass."""
        ids = torch.as_tensor(input_ids, dtype=torch.long)
        x = _t(self.w, "model.embed_tokens.weight", self.dtype)[ids]
        n = ids.shape[0] if n_prompt is None else int(n_prompt)
        if video_feats is not None and n != 1:
            proj = mm_projector(video_feats.to(self.dtype), self.w, self.cfg.projector)
            x = torch.cat([splice_video_embeddings(ids[:n], x[:n], proj, vid_start, vid_end, vid_patch), x[n:]], dim=0)
        return self._forward_embeds(x, all_logits)

    def step(self, token: int) -> torch.Tensor:
        x = _t(self.w, "model.embed_tokens.weight", self.dtype)[torch.tensor([int(token)])]
        return self._forward_embeds(x)


def greedy_generate(w: dict, cfg: LlamaCfg, input_ids, video_feats, vid_start: int, vid_end: int, vid_patch: int,
                    max_new_tokens: int, eos_id: int | None = None, dtype=torch.float32,
                    return_margins: bool = False, cache_weights: bool = False, return_logits: bool = False):
    """Greedy decode by driving forward (SURVEY.md 8c oracle recipe; replaces model.generate of
    video_chatgpt/inference.py:105-112 with do_sample=False).  Returns the new token ids (and the
    top-1/top-2 logit gap of each step when `return_margins`)."""
    m = LlamaOracle(w, cfg, dtype, cache_weights=cache_weights)
    logits = m.prefill(input_ids, video_feats, vid_start, vid_end, vid_patch)
    toks, margins, all_logits = [], [], []
    for i in range(max_new_tokens):
        top2 = torch.topk(logits[0], 2)
        tok = int(top2.indices[0])
        toks.append(tok)
        margins.append(float(top2.values[0] - top2.values[1]))
        if return_logits:
            all_logits.append(logits[0].clone())
        if (eos_id is not None and tok == eos_id) or i + 1 == max_new_tokens:
            break
        logits = m.step(tok)
    if return_logits:
        return toks, margins, torch.stack(all_logits)
    return (toks, margins) if return_margins else toks


def sample_cdf(logits, temperature: float, top_k: int = 50, dtype=torch.float64):
    """The distribution HF's sample loop draws from, as a CDF over the vocabulary: TemperatureLogitsWarper (scores / temperature,
    HF:generation/logits_process.py:302), TopKLogitsWarper (`scores < topk(scores, k)[..., -1]` -> -inf, :593; k = 50 is HF's default
    GenerationConfig value, HF:generation/configuration_utils.py:617, which the reference inherits at video_chatgpt/inference.py:106-112),
    softmax (HF:generation/utils.py:2921).  Returns cumsum(probs) [B, V] in `dtype`."""
    x = torch.as_tensor(logits).to(dtype) / temperature
    if top_k and 0 < top_k < x.shape[-1]:
        kth = torch.topk(x, top_k, dim=-1).values[..., -1, None]
        x = x.masked_fill(x < kth, float("-inf"))
    return torch.cumsum(torch.softmax(x, dim=-1), dim=-1)


def sample_pick(logits, u, temperature: float, top_k: int = 50, dtype=torch.float64):
    """Inverse-CDF draw: the first vocabulary index whose cumulative probability exceeds u (u [B] uniforms in [0, 1)) -- what
    torch.multinomial(probs, 1) (HF:generation/utils.py:2923) does with its own uniform.  Returns (tokens [B], distance of u to the
    nearest CDF step [B]): a HIP pick is required to match only where that distance is above summation-order noise."""
    cdf = sample_cdf(logits, temperature, top_k, dtype)
    uu = torch.as_tensor(u).to(dtype)[:, None]
    tok = (cdf <= uu).sum(-1).clamp_max(cdf.shape[-1] - 1)
    return tok, (cdf - uu).abs().min(-1).values


def quantize_e4m3_rows(w) -> "torch.Tensor":
    """CPU twin of csrc/fp8.hip: per-row power-of-two scale s = 2^ceil(log2(amax / 448)), q = e4m3fn(w / s) (round to nearest even),
    returns the dequantised matrix q * s (fp32).  There is no reference implementation of this step (the reference has no
    quantised path, BASELINE config 5 is this repo's own extension); the HIP quantiser is tested for bit equality with this twin."""
    w = torch.as_tensor(w, dtype=torch.float32)
    amax = w.abs().amax(dim=1, keepdim=True)
    mant, exp = torch.frexp(amax / 448.0)                       # amax/448 = mant * 2^exp, mant in [0.5, 1)
    e = torch.where(mant == 0.5, exp - 1, exp)
    s = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), e), torch.ones_like(amax))
    q = (w / s).to(torch.float8_e4m3fn).to(torch.float32)
    return q * s


FP8_KEYS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def quantize_llama_weights_fp8(w: dict, round16=None) -> dict:
    """State dict with every decoder matrix and lm_head replaced by its dequantised fp8 version.  `round16`: the model's 16-bit
    storage dtype (weights are rounded to it BEFORE quantisation, as the library quantises its packed 16-bit copy).
    q/k/v share one fused matrix in the library but scales are per row, so quantising them separately is identical."""
    out = dict(w)
    for k, v in w.items():
        if k == "lm_head.weight" or (k.endswith(".weight") and any(f".{n}." in k for n in FP8_KEYS)):
            t = torch.as_tensor(v, dtype=torch.float32)
            if round16 is not None:
                t = t.to(round16).to(torch.float32)
            out[k] = quantize_e4m3_rows(t).numpy()
    return out
