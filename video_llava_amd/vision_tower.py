"""`vision_tower` drop-in: the object `initialize_model` returns in slot 2 and `video_chatgpt_infer`
calls as `vision_tower(image_tensor, output_hidden_states=True).hidden_states[-2][:, 1:]`
(reference: video_chatgpt/eval/model_utils.py:134-136, video_chatgpt/inference.py:92-94,
video_chatgpt/chat.py:140-143, scripts/save_spatio_temporal_clip_features.py:84-92,116-120).

Same call contract as HF `CLIPVisionModel`, computed by libpgv's hand-written gfx950 kernels.  Only the
layers that feed the hidden state the caller actually indexes are executed (the reference runs layer 24
and then discards it).
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib


class CLIPVisionTowerConfig:
    """Subset of HF CLIPVisionConfig the path reads (image_size / patch_size / hidden_size are read by
    video_chatgpt/model/video_chatgpt.py:44-49)."""

    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                 image_size=224, patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu", **_ignored):
        if hidden_act != "quick_gelu":
            raise ValueError(f"only quick_gelu CLIP towers are supported (got {hidden_act})")
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.image_size, self.patch_size, self.layer_norm_eps = image_size, patch_size, layer_norm_eps
        self.hidden_act = hidden_act

    @classmethod
    def from_pretrained(cls, path: str) -> "CLIPVisionTowerConfig":
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        cfg = cfg.get("vision_config", cfg)
        return cls(**cfg)


class _LazyHiddenStates:
    """Sequence of the tower's L+1 hidden states; entry k is computed on first access by running k layers."""

    def __init__(self, tower: "CLIPVisionTower", pixels: torch.Tensor):
        self._tower, self._pixels, self._cache = tower, pixels, {}

    def __len__(self):
        return self._tower.config.num_hidden_layers + 1

    def __getitem__(self, idx):
        n = len(self)
        if isinstance(idx, slice):
            return [self[i] for i in range(*idx.indices(n))]
        if idx < 0:
            idx += n
        if not 0 <= idx < n:
            raise IndexError(idx)
        if idx not in self._cache:
            self._cache[idx] = self._tower.hidden_state(self._pixels, idx)
        return self._cache[idx]


class VisionTowerOutput:
    def __init__(self, hidden_states):
        self.hidden_states = hidden_states


class CLIPVisionTower:
    def __init__(self, config: CLIPVisionTowerConfig | None = None, torch_dtype: torch.dtype = torch.float16, device=None):
        self.config = config or CLIPVisionTowerConfig()
        self.dtype = torch_dtype
        self.ctx = _lib.Context.get(device)
        self.device = torch.device("cuda", self.ctx.device)
        c = self.config
        vc = _lib.VitConfig(c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads,
                            c.image_size, c.patch_size, c.layer_norm_eps)
        h = C.c_void_p()
        _lib.check(self.ctx.lib.pgv_vit_create(self.ctx.handle, C.byref(vc), _lib.dtype_code(torch_dtype), C.byref(h)), "pgv_vit_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.pgv_vit_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- the HF module surface the reference touches ---------------------------------------------
    def cuda(self, *a, **k):
        return self

    def eval(self):
        return self

    def half(self):
        return self

    def to(self, *a, **k):
        return self

    @property
    def num_patches(self) -> int:
        return (self.config.image_size // self.config.patch_size) ** 2

    def _expected_shape(self, key: str):
        """Shape HF's CLIPVisionModel holds under `key` for this config (None: not a key of the tower)."""
        c = self.config
        C_, I, p = c.hidden_size, c.intermediate_size, c.patch_size
        k = key[len("vision_model."):] if key.startswith("vision_model.") else key
        fixed = {"embeddings.class_embedding": (C_,), "embeddings.patch_embedding.weight": (C_, 3, p, p),
                 "embeddings.position_embedding.weight": (self.num_patches + 1, C_), "pre_layrnorm.weight": (C_,), "pre_layrnorm.bias": (C_,),
                 "post_layernorm.weight": (C_,), "post_layernorm.bias": (C_,)}
        if k in fixed:
            return fixed[k]
        if not k.startswith("encoder.layers."):
            return None
        rest = k.split(".", 3)[-1]
        per_layer = {"mlp.fc1.weight": (I, C_), "mlp.fc1.bias": (I,), "mlp.fc2.weight": (C_, I), "mlp.fc2.bias": (C_,)}
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            per_layer[f"self_attn.{n}.weight"] = (C_, C_)
            per_layer[f"self_attn.{n}.bias"] = (C_,)
        for n in ("layer_norm1", "layer_norm2"):
            per_layer[f"{n}.weight"] = (C_,)
            per_layer[f"{n}.bias"] = (C_,)
        return per_layer.get(rest)

    def load_state_dict(self, sd: dict, strict: bool = True):
        """Accepts HF CLIPVisionModel keys (with or without the `vision_model.` prefix); values may be torch
        tensors (any device, fp32/fp16/bf16) or numpy arrays.  A tensor whose shape disagrees with the config raises
        torch's "size mismatch" RuntimeError (the C ABI re-checks the element count and never reads past the buffer)."""
        lib = self.ctx.lib
        for k, v in sd.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            v = v.detach()
            if v.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                if "position_ids" in k:
                    continue
                v = v.float()
            v = v.contiguous()
            exp = self._expected_shape(k)
            if exp is not None and tuple(v.shape) != exp:
                raise RuntimeError(f"Error(s) in loading state_dict for CLIPVisionTower:\n\tsize mismatch for {k}: copying a param with shape "
                                   f"{tuple(v.shape)} from checkpoint, the shape in current model is {exp}.")
            if v.is_cuda:
                torch.cuda.current_stream(v.device).synchronize()
            rc = lib.pgv_vit_load_tensor(self.handle, k.encode(), v.data_ptr(), _lib.dtype_code(v.dtype), 1 if v.is_cuda else 0,
                                         v.numel(), _lib.stream_ptr(self.device))
            if rc == _lib.PGV_ENAME and not strict:
                continue
            _lib.check(rc, f"load {k}")
        torch.cuda.synchronize(self.device)
        missing = lib.pgv_vit_missing(self.handle)
        if strict and missing:
            raise RuntimeError(f"CLIPVisionTower.load_state_dict: {missing} tensors missing")
        return missing

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype: torch.dtype = torch.float16, low_cpu_mem_usage: bool = True, device=None):
        """Load a local HF CLIP checkpoint directory (config.json + model.safetensors or pytorch_model.bin)."""
        cfg = CLIPVisionTowerConfig.from_pretrained(path)
        tower = cls(cfg, torch_dtype, device)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        sd = {k: v for k, v in sd.items() if k.startswith("vision_model.") or not k.startswith(("text_model.", "logit_scale", "text_projection", "visual_projection"))}
        tower.load_state_dict(sd, strict=False)
        if tower.ctx.lib.pgv_vit_missing(tower.handle):
            raise RuntimeError(f"{path}: vision tower checkpoint is incomplete")
        return tower

    # ---- compute ---------------------------------------------------------------------------------
    def hidden_state(self, pixel_values: torch.Tensor, k: int) -> torch.Tensor:
        """hidden_states[k]: [T, patches+1, hidden] in the tower dtype."""
        px = pixel_values
        if not px.is_cuda:
            px = px.to(self.device)
        if px.dtype != self.dtype:
            px = px.to(self.dtype)
        px = px.contiguous()
        T = px.shape[0]
        S = self.config.image_size
        if tuple(px.shape[1:]) != (3, S, S):
            raise ValueError(f"Input image size ({px.shape[2]}*{px.shape[3]}) doesn't match model ({S}*{S}).")
        out = torch.empty(T, self.num_patches + 1, self.config.hidden_size, dtype=self.dtype, device=px.device)
        _lib.check(self.ctx.lib.pgv_vit_forward(self.ctx.handle, self.handle, px.data_ptr(), T, k, out.data_ptr(),
                                                _lib.stream_ptr(px.device)), "pgv_vit_forward")
        return out

    def __call__(self, pixel_values: torch.Tensor, output_hidden_states: bool = True, **_kw) -> VisionTowerOutput:
        if not output_hidden_states:
            raise ValueError("the PG-Video-LLaVA path always reads hidden_states; call with output_hidden_states=True")
        return VisionTowerOutput(_LazyHiddenStates(self, pixel_values))

    forward = __call__
