"""Oracle: frame preprocessing, CLIP ViT forward, spatio-temporal pooling, mm_projector.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain torch on CPU; `dtype` selects
fp32 (the reference-on-CPU arithmetic) or fp64 (ground truth for error budgets).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .synth import ClipCfg

# CLIPImageProcessor defaults = OpenAI CLIP statistics (call sites: video_chatgpt/inference.py:86,
# video_chatgpt/chat.py:67, scripts/save_spatio_temporal_clip_features.py:105).
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def get_seq_frames(total_num_frames: int, desired_num_frames: int) -> list[int]:
    """Uniform frame sampling (video_chatgpt/eval/model_utils.py:55-79; duplicate at
    scripts/save_spatio_temporal_clip_features.py:35-43).  Segment length (n-1)/k, the
    chosen index is the integer midpoint of the *rounded* segment ends; np.round is
    round-half-to-even, which matters when seg*i lands on .5."""
    seg = float(total_num_frames - 1) / desired_num_frames
    out = []
    for i in range(desired_num_frames):
        lo = int(np.round(seg * i))
        hi = int(np.round(seg * (i + 1)))
        out.append((lo + hi) // 2)
    return out


def clip_preprocess(frames_u8: np.ndarray, dtype=torch.float32) -> torch.Tensor:
    """CLIPImageProcessor.preprocess for frames already at the crop size (video_chatgpt/
    inference.py:86).  For HxW == crop the resize (shortest edge, bicubic) and the centre
    crop are identities, leaving rescale 1/255 then (x-mean)/std, NHWC->NCHW (SURVEY.md 8a F3,
    verified bit-equal against HF in oracle/gen_golden.py).  The HF processor computes
    `x * (1/255)` in fp32 then `(x - mean) / std` in fp32."""
    assert frames_u8.dtype == np.uint8 and frames_u8.ndim == 4 and frames_u8.shape[-1] == 3
    x = torch.from_numpy(frames_u8.astype(np.float32)) * np.float32(1.0 / 255.0)
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32)
    std = torch.tensor(CLIP_STD, dtype=torch.float32)
    x = (x - mean) / std
    return x.permute(0, 3, 1, 2).contiguous().to(dtype)


def _t(w: dict, key: str, dtype) -> torch.Tensor:
    v = w[key]
    return (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).to(dtype)


def _layer_norm(x, g, b, eps):
    # nn.LayerNorm: biased variance over the last dim (HF:clip/modeling_clip.py:357-358,605)
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def quick_gelu(x):
    """QuickGELU x*sigmoid(1.702x) (HF:activations.py QuickGELUActivation; CLIP-L/14 hidden_act)."""
    return x * torch.sigmoid(1.702 * x)


def clip_embeddings(pixels: torch.Tensor, w: dict, cfg: ClipCfg) -> torch.Tensor:
    """CLIPVisionEmbeddings.forward (HF:clip/modeling_clip.py:200-218): stride-14 conv without
    bias == per-patch dot product with the [C, 3*14*14] filter bank (channel-major, then
    row, then column inside a patch), CLS prepended, learned position table added."""
    dtype = pixels.dtype
    T = pixels.shape[0]
    g, p = cfg.grid, cfg.patch
    patches = pixels.reshape(T, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(T, g * g, 3 * p * p)
    filt = _t(w, "vision_model.embeddings.patch_embedding.weight", dtype).reshape(cfg.hidden, -1)
    x = patches @ filt.t()
    cls = _t(w, "vision_model.embeddings.class_embedding", dtype).expand(T, 1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + _t(w, "vision_model.embeddings.position_embedding.weight", dtype)


def clip_encoder_layer(x: torch.Tensor, w: dict, cfg: ClipCfg, i: int) -> torch.Tensor:
    """CLIPEncoderLayer.forward (HF:clip/modeling_clip.py:364-384) with eager attention
    (:259-277): scores scaled by head_dim**-0.5, softmax in fp32-or-wider, no mask."""
    dtype = x.dtype
    q = f"vision_model.encoder.layers.{i}."
    T, N, C = x.shape
    H, D = cfg.heads, C // cfg.heads
    h = _layer_norm(x, _t(w, q + "layer_norm1.weight", dtype), _t(w, q + "layer_norm1.bias", dtype), cfg.eps)

    def lin(name, v):
        return v @ _t(w, q + name + ".weight", dtype).t() + _t(w, q + name + ".bias", dtype)

    qs = lin("self_attn.q_proj", h).view(T, N, H, D).transpose(1, 2)
    ks = lin("self_attn.k_proj", h).view(T, N, H, D).transpose(1, 2)
    vs = lin("self_attn.v_proj", h).view(T, N, H, D).transpose(1, 2)
    att = torch.softmax((qs @ ks.transpose(-1, -2)) * (D ** -0.5), dim=-1)
    o = (att @ vs).transpose(1, 2).reshape(T, N, C)
    x = x + lin("self_attn.out_proj", o)
    h = _layer_norm(x, _t(w, q + "layer_norm2.weight", dtype), _t(w, q + "layer_norm2.bias", dtype), cfg.eps)
    h = lin("mlp.fc2", quick_gelu(lin("mlp.fc1", h)))
    return x + h


def clip_hidden_states(pixels: torch.Tensor, w: dict, cfg: ClipCfg, upto: int | None = None) -> list[torch.Tensor]:
    """`vision_tower(pixels, output_hidden_states=True).hidden_states` (call site
    video_chatgpt/inference.py:93): entry 0 is the pre-LayerNorm'ed embedding
    (HF:clip/modeling_clip.py:642), entry i the output of layer i.  post_layernorm is never
    applied to hidden_states.  `upto` limits the number of layers evaluated."""
    dtype = pixels.dtype
    x = clip_embeddings(pixels, w, cfg)
    x = _layer_norm(x, _t(w, "vision_model.pre_layrnorm.weight", dtype),
                    _t(w, "vision_model.pre_layrnorm.bias", dtype), cfg.eps)
    hs = [x]
    n = cfg.layers if upto is None else upto
    for i in range(n):
        x = clip_encoder_layer(x, w, cfg, i)
        hs.append(x)
    return hs


def clip_select_features(pixels: torch.Tensor, w: dict, cfg: ClipCfg, select_layer: int = -2) -> torch.Tensor:
    """`hidden_states[-2][:, 1:]` (video_chatgpt/inference.py:94; chat.py:141-143;
    save_spatio_temporal_clip_features.py:118-120): output of layer L-1 (=23 of 24), CLS dropped.
    Only the layers that feed the selected state are evaluated (the reference also runs
    layer 24 and discards it -- SURVEY.md 8a V6)."""
    n_states = cfg.layers + 1
    idx = select_layer if select_layer >= 0 else n_states + select_layer
    return clip_hidden_states(pixels, w, cfg, upto=idx)[idx][:, 1:]


def spatio_temporal_pool_torch(features: torch.Tensor, num_temporal_tokens: int = 100) -> torch.Tensor:
    """get_spatio_temporal_features_torch (video_chatgpt/inference.py:13-44; private copy at
    chat.py:77-87).  features [T, P, C].  Temporal tokens = mean over the P patches of each
    frame, zero-padded to 100 rows when T < 100 (never truncated when T > 100); spatial tokens
    = mean over the T frames; rows = [temporal; spatial]; result cast to fp16.  torch.mean on an
    fp16 tensor accumulates in fp32 and rounds once; here the mean is taken in the input dtype
    of the oracle (fp32/fp64) and rounded once, which is the same value up to that rounding."""
    T, P, C = features.shape
    temporal = features.mean(dim=1)
    if num_temporal_tokens - T > 0:
        temporal = torch.cat([temporal, torch.zeros(num_temporal_tokens - T, C, dtype=temporal.dtype)], dim=0)
    spatial = features.mean(dim=0)
    return torch.cat([temporal, spatial], dim=0).half()


def spatio_temporal_pool_numpy(features: np.ndarray, num_temporal_tokens: int = 100) -> np.ndarray:
    """get_spatio_temporal_features (scripts/save_spatio_temporal_clip_features.py:46-57): the
    offline-extraction twin.  The caller hands it an fp16 array (:123), np.mean of fp16 sums
    pairwise in fp32 and returns fp16; padding by np.pad zeros; np.concatenate."""
    T, P, C = features.shape
    temporal = np.mean(features, axis=1)
    if num_temporal_tokens - T > 0:
        temporal = np.pad(temporal, ((0, num_temporal_tokens - T), (0, 0)), mode="constant")
    spatial = np.mean(features, axis=0)
    return np.concatenate([temporal, spatial], axis=0)


def gelu_erf(x):
    """nn.GELU() default = exact erf form (multimodal_projector/builder.py:43)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mm_projector(x: torch.Tensor, w: dict, kind: str = "linear", prefix: str = "model.mm_projector") -> torch.Tensor:
    """mm_projector forward.  224-px towers get a bare nn.Linear(1024, H)
    (video_chatgpt/model/video_chatgpt.py:52-53); otherwise build_vision_projector
    (multimodal_projector/builder.py:33-50): 'linear', 'mlp{N}x_gelu' = Linear + (N-1) x [GELU, Linear]
    with Sequential indices 0,2,4.., or 'identity'."""
    dtype = x.dtype
    if kind == "identity":
        return x
    if kind == "linear":
        return x @ _t(w, prefix + ".weight", dtype).t() + _t(w, prefix + ".bias", dtype)
    if kind.startswith("mlp") and kind.endswith("x_gelu"):
        depth = int(kind[3:-6])
        y = x @ _t(w, prefix + ".0.weight", dtype).t() + _t(w, prefix + ".0.bias", dtype)
        for d in range(1, depth):
            y = gelu_erf(y)
            y = y @ _t(w, f"{prefix}.{2 * d}.weight", dtype).t() + _t(w, f"{prefix}.{2 * d}.bias", dtype)
        return y
    raise ValueError(f"Unknown projector type: {kind}")
