"""Random-weight checkpoints generated directly on the GPU (there are no model files offline): used by bench.py and
__graft_entry__.smoke() to instantiate ViT-L/14 and LLaMA-7B/13B-shaped models with the HF state-dict names the
loaders expect.  Distributions follow HF's initializer_range=0.02 convention; norm gains are ~1.
"""
from __future__ import annotations

import torch


def iter_clip_tensors(hidden=1024, inter=4096, layers=24, image=224, patch=14, device="cuda", dtype=torch.float16, seed=0, std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)

    def n(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    tokens = (image // patch) ** 2 + 1
    p = "vision_model."
    yield p + "embeddings.class_embedding", n(hidden)
    yield p + "embeddings.patch_embedding.weight", n(hidden, 3, patch, patch)
    yield p + "embeddings.position_embedding.weight", n(tokens, hidden)
    yield p + "pre_layrnorm.weight", 1 + n(hidden, s=0.1)
    yield p + "pre_layrnorm.bias", n(hidden, s=0.1)
    for i in range(layers):
        q = f"{p}encoder.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield q + f"self_attn.{name}.weight", n(hidden, hidden)
            yield q + f"self_attn.{name}.bias", n(hidden)
        yield q + "layer_norm1.weight", 1 + n(hidden, s=0.1)
        yield q + "layer_norm1.bias", n(hidden, s=0.1)
        yield q + "mlp.fc1.weight", n(inter, hidden)
        yield q + "mlp.fc1.bias", n(inter)
        yield q + "mlp.fc2.weight", n(hidden, inter)
        yield q + "mlp.fc2.bias", n(hidden)
        yield q + "layer_norm2.weight", 1 + n(hidden, s=0.1)
        yield q + "layer_norm2.bias", n(hidden, s=0.1)


def iter_llama_tensors(vocab=32003, hidden=4096, inter=11008, layers=32, mm_hidden=1024, projector="linear", device="cuda",
                       dtype=torch.float16, seed=0, std=0.02, head_std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)

    def n(*shape, s=std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * s).to(dtype)

    yield "model.embed_tokens.weight", n(vocab, hidden, s=head_std)
    for i in range(layers):
        q = f"model.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj", "o_proj"):
            yield q + f"self_attn.{name}.weight", n(hidden, hidden)
        yield q + "mlp.gate_proj.weight", n(inter, hidden)
        yield q + "mlp.up_proj.weight", n(inter, hidden)
        yield q + "mlp.down_proj.weight", n(hidden, inter)
        yield q + "input_layernorm.weight", 1 + n(hidden, s=0.1)
        yield q + "post_attention_layernorm.weight", 1 + n(hidden, s=0.1)
    yield "model.norm.weight", 1 + n(hidden, s=0.1)
    yield "lm_head.weight", n(vocab, hidden, s=head_std)
    if projector == "linear":
        yield "model.mm_projector.weight", n(hidden, mm_hidden)
        yield "model.mm_projector.bias", n(hidden)
    elif projector == "mlp2x_gelu":
        yield "model.mm_projector.0.weight", n(hidden, mm_hidden)
        yield "model.mm_projector.0.bias", n(hidden)
        yield "model.mm_projector.2.weight", n(hidden, hidden)
        yield "model.mm_projector.2.bias", n(hidden)


def load_streaming(module, tensor_iter):
    """Feed tensors one at a time (so a 13 GB decoder never exists twice on the device)."""
    for name, t in tensor_iter:
        module.load_state_dict({name: t}, strict=False)
        del t
