"""Deterministic synthetic weights / inputs shared by the oracle, the golden
generator and the parity tests (TEST INFRASTRUCTURE -- see oracle/__init__.py).

There are no model checkpoints offline, so every parity case runs
on seeded random weights.  numpy's `default_rng` (PCG64) is bit-reproducible
across machines, so the weights generated here in the build container, on the
GPU box and inside `oracle/gen_golden.py` are identical; golden fixtures store
only inputs' seeds and the reference's outputs.

State-dict key names are the HuggingFace ones (HF:clip/modeling_clip.py
CLIPVisionModel, HF:llama/modeling_llama.py LlamaForCausalLM) plus the
reference's `model.mm_projector.*` (video_chatgpt/model/video_chatgpt.py:51-55;
train/llava_trainer.py:34), so the same dict loads into the reference modules
and into the HIP weight packer.
"""
from __future__ import annotations

import dataclasses

import numpy as np


@dataclasses.dataclass(frozen=True)
class ClipCfg:
    """CLIP vision tower shape (defaults = openai/clip-vit-large-patch14)."""
    hidden: int = 1024
    inter: int = 4096
    layers: int = 24
    heads: int = 16
    image: int = 224
    patch: int = 14
    eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image // self.patch

    @property
    def patches(self) -> int:
        return self.grid * self.grid

    @property
    def tokens(self) -> int:
        return self.patches + 1


@dataclasses.dataclass(frozen=True)
class LlamaCfg:
    """Decoder shape (defaults = LLaVA/Vicuna-7B + 3 video tokens)."""
    vocab: int = 32003
    hidden: int = 4096
    inter: int = 11008
    layers: int = 32
    heads: int = 32
    head_dim: int = 128
    eps: float = 1e-5
    rope_theta: float = 10000.0
    mm_hidden: int = 1024          # CLIP width feeding mm_projector (video_chatgpt.py:106 hard-codes 1024)
    projector: str = "linear"      # 'linear' | 'mlp2x_gelu' | 'identity' (multimodal_projector/builder.py:33-50)


CLIP_L14_224 = ClipCfg()
CLIP_L14_336 = ClipCfg(image=336)
CLIP_TINY = ClipCfg(hidden=1024, inter=512, layers=3, heads=16, image=56, patch=14)   # width must stay 1024
LLAMA_7B = LlamaCfg()
LLAMA_13B = LlamaCfg(hidden=5120, inter=13824, layers=40, heads=40)
LLAMA_TINY = LlamaCfg(vocab=515, hidden=512, inter=768, layers=2, heads=4)


def _normal(rng, shape, std):
    return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)


def make_clip_weights(cfg: ClipCfg, seed: int = 0, std: float = 0.02) -> dict[str, np.ndarray]:
    """Seeded CLIPVisionModel state dict (fp32 numpy).  Norm gains are 1+N(0,0.1) and all
    biases non-zero so that a dropped bias / gain shows up in parity."""
    rng = np.random.default_rng(seed)
    C, I = cfg.hidden, cfg.inter
    w: dict[str, np.ndarray] = {}
    p = "vision_model."
    w[p + "embeddings.class_embedding"] = _normal(rng, (C,), std)
    w[p + "embeddings.patch_embedding.weight"] = _normal(rng, (C, 3, cfg.patch, cfg.patch), std)
    w[p + "embeddings.position_embedding.weight"] = _normal(rng, (cfg.tokens, C), std)
    w[p + "pre_layrnorm.weight"] = 1.0 + _normal(rng, (C,), 0.1)
    w[p + "pre_layrnorm.bias"] = _normal(rng, (C,), 0.1)
    for i in range(cfg.layers):
        q = f"{p}encoder.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w[q + f"self_attn.{name}.weight"] = _normal(rng, (C, C), std)
            w[q + f"self_attn.{name}.bias"] = _normal(rng, (C,), std)
        w[q + "layer_norm1.weight"] = 1.0 + _normal(rng, (C,), 0.1)
        w[q + "layer_norm1.bias"] = _normal(rng, (C,), 0.1)
        w[q + "mlp.fc1.weight"] = _normal(rng, (I, C), std)
        w[q + "mlp.fc1.bias"] = _normal(rng, (I,), std)
        w[q + "mlp.fc2.weight"] = _normal(rng, (C, I), std)
        w[q + "mlp.fc2.bias"] = _normal(rng, (C,), std)
        w[q + "layer_norm2.weight"] = 1.0 + _normal(rng, (C,), 0.1)
        w[q + "layer_norm2.bias"] = _normal(rng, (C,), 0.1)
    w[p + "post_layernorm.weight"] = 1.0 + _normal(rng, (C,), 0.1)
    w[p + "post_layernorm.bias"] = _normal(rng, (C,), 0.1)
    return w


def make_llama_weights(cfg: LlamaCfg, seed: int = 0, std: float = 0.02,
                       head_std: float | None = None) -> dict[str, np.ndarray]:
    """Seeded VideoChatGPTLlamaForCausalLM state dict (fp32 numpy).

    `head_std` scales lm_head/embeddings separately: with std 0.02 everywhere greedy
    margins of a random model sit at fp16 noise level (SURVEY.md 7 'hard parts'); tests
    that demand token-exact decode pass a larger `head_std` and assert the oracle margin."""
    rng = np.random.default_rng(seed)
    H, I, V = cfg.hidden, cfg.inter, cfg.vocab
    hs = std if head_std is None else head_std
    w: dict[str, np.ndarray] = {}
    w["model.embed_tokens.weight"] = _normal(rng, (V, H), hs)
    for i in range(cfg.layers):
        q = f"model.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            w[q + f"self_attn.{name}.weight"] = _normal(rng, (H, H), std)
        w[q + "mlp.gate_proj.weight"] = _normal(rng, (I, H), std)
        w[q + "mlp.up_proj.weight"] = _normal(rng, (I, H), std)
        w[q + "mlp.down_proj.weight"] = _normal(rng, (H, I), std)
        w[q + "input_layernorm.weight"] = 1.0 + _normal(rng, (H,), 0.1)
        w[q + "post_attention_layernorm.weight"] = 1.0 + _normal(rng, (H,), 0.1)
    w["model.norm.weight"] = 1.0 + _normal(rng, (H,), 0.1)
    w["lm_head.weight"] = _normal(rng, (V, H), hs)
    if cfg.projector == "linear":
        w["model.mm_projector.weight"] = _normal(rng, (H, cfg.mm_hidden), std)
        w["model.mm_projector.bias"] = _normal(rng, (H,), std)
    elif cfg.projector.startswith("mlp"):
        depth = int(cfg.projector[3:cfg.projector.index("x")])
        w["model.mm_projector.0.weight"] = _normal(rng, (H, cfg.mm_hidden), std)
        w["model.mm_projector.0.bias"] = _normal(rng, (H,), std)
        for d in range(1, depth):
            w[f"model.mm_projector.{2 * d}.weight"] = _normal(rng, (H, H), std)
            w[f"model.mm_projector.{2 * d}.bias"] = _normal(rng, (H,), std)
    return w


def llama_tensor_specs(cfg: LlamaCfg, std: float = 0.02, head_std: float | None = None):
    """[(key, shape, std, offset)] of a VideoChatGPTLlamaForCausalLM state dict in a fixed order (value = offset + N(0, std))."""
    H, I, V = cfg.hidden, cfg.inter, cfg.vocab
    hs = std if head_std is None else head_std
    specs = [("model.embed_tokens.weight", (V, H), hs, 0.0)]
    for i in range(cfg.layers):
        q = f"model.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            specs.append((q + f"self_attn.{name}.weight", (H, H), std, 0.0))
        specs += [(q + "mlp.gate_proj.weight", (I, H), std, 0.0), (q + "mlp.up_proj.weight", (I, H), std, 0.0),
                  (q + "mlp.down_proj.weight", (H, I), std, 0.0), (q + "input_layernorm.weight", (H,), 0.1, 1.0),
                  (q + "post_attention_layernorm.weight", (H,), 0.1, 1.0)]
    specs += [("model.norm.weight", (H,), 0.1, 1.0), ("lm_head.weight", (V, H), hs, 0.0)]
    if cfg.projector == "linear":
        specs += [("model.mm_projector.weight", (H, cfg.mm_hidden), std, 0.0), ("model.mm_projector.bias", (H,), std, 0.0)]
    elif cfg.projector.startswith("mlp"):
        depth = int(cfg.projector[3:cfg.projector.index("x")])
        specs += [("model.mm_projector.0.weight", (H, cfg.mm_hidden), std, 0.0), ("model.mm_projector.0.bias", (H,), std, 0.0)]
        for d in range(1, depth):
            specs += [(f"model.mm_projector.{2 * d}.weight", (H, H), std, 0.0), (f"model.mm_projector.{2 * d}.bias", (H,), std, 0.0)]
    return specs


def make_llama_weights_16bit(cfg: LlamaCfg, seed: int = 0, std: float = 0.02, head_std: float | None = None, dtype: str = "float16",
                             workers: int | None = None) -> dict:
    """Full-size (7B / 13B) seeded state dict as 16-bit TORCH tensors -- what a released checkpoint is.  Every tensor has its own
    PCG64 stream (seeded with [seed, index]) and is filled by a thread pool (numpy releases the GIL), so 6.7e9 weights take seconds on
    a many-core host instead of minutes, with bit-identical values on any machine and any worker count."""
    import concurrent.futures
    import os
    import torch
    tdt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[dtype]
    specs = llama_tensor_specs(cfg, std, head_std)

    def fill(job):
        idx, (key, shape, s, off) = job
        rng = np.random.default_rng([seed, idx])
        out = torch.empty(shape, dtype=tdt)
        rows = shape[0]
        step = max(1, (1 << 24) // max(1, int(np.prod(shape[1:], dtype=np.int64))))        # ~64 MB fp32 pieces
        for r0 in range(0, rows, step):
            n = min(step, rows - r0)
            a = rng.standard_normal((n,) + tuple(shape[1:]), dtype=np.float32)
            a *= np.float32(s)
            if off:
                a += np.float32(off)
            out[r0:r0 + n] = torch.from_numpy(a).to(tdt)
        return key, out

    workers = workers or min(32, os.cpu_count() or 8)
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as ex:
        return dict(ex.map(fill, enumerate(specs)))


def make_frames(n_frames: int, size: int = 224, seed: int = 0) -> np.ndarray:
    """Synthetic clip: uint8 [T, size, size, 3] (default_rng(seed).integers(0,256))."""
    return np.random.default_rng(seed).integers(0, 256, (n_frames, size, size, 3), dtype=np.uint8)


def quantize_weights(w: dict, dtype: str) -> dict:
    """Round every tensor to the checkpoint dtype ('float16' / 'bfloat16') and return it as fp32 again.  Released
    PG-Video-LLaVA / CLIP checkpoints ARE 16-bit, so the reference on CPU and the HIP path see the same weight
    values; parity cases at 7B shapes use this so that weight rounding is not misread as kernel error."""
    import torch
    dt = {"float16": torch.float16, "bfloat16": torch.bfloat16}[dtype]
    return {k: torch.from_numpy(v).to(dt).float().numpy() for k, v in w.items()}
