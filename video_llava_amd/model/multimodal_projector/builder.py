"""`mm_projector` factory -- same names, config keys and state-dict layout as the reference's
video_chatgpt/model/multimodal_projector/builder.py:33-50 (`linear` | `mlp{N}x_gelu` | `identity`), with the
forward pass on libpgv's MFMA GEMM (bias / exact-erf GELU fused in the epilogue).

The modules stay `nn.Module`s whose parameters are named exactly like the reference's
(`weight`/`bias`, or `0.weight`, `0.bias`, `2.weight`, ... for the MLP) so `mm_projector.bin` files written by
train/llava_trainer.py:34-46 load through `load_state_dict` unchanged (eval/model_utils.py:122-127).
"""
from __future__ import annotations

import ctypes as C
import re

import torch
import torch.nn as nn

from ... import _lib


class IdentityMap(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x

    @property
    def config(self):
        return {"mm_projector_type": "identity"}


class HipLinear(nn.Module):
    """nn.Linear-compatible parameters; forward = pgv_projector(depth=1)."""

    def __init__(self, in_features: int, out_features: int, dtype=torch.float16, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_features, dtype=dtype, device=device), requires_grad=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _run_projector([self], x)


class _GELUMarker(nn.Module):
    """Occupies the odd Sequential slots (nn.GELU in the reference); the activation itself is fused into the
    preceding GEMM's epilogue."""

    def forward(self, x):
        return torch.nn.functional.gelu(x)


class HipProjectorMLP(nn.Sequential):
    """Linear + (depth-1) x [GELU, Linear] with Sequential indices 0, 2, 4, ... (builder.py:39-46)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _run_projector([m for m in self if isinstance(m, HipLinear)], x)


def _run_projector(linears, x: torch.Tensor) -> torch.Tensor:
    w0 = linears[0].weight
    if not w0.is_cuda:
        raise RuntimeError("mm_projector lives on the CPU: move the model to the GPU (there is no CPU fallback)")
    ctx = _lib.Context.get(w0.device)
    dtype = w0.dtype
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).to(device=w0.device, dtype=dtype).contiguous()
    rows = x2.shape[0]
    hidden = linears[-1].out_features
    y = torch.empty(rows, hidden, dtype=dtype, device=w0.device)
    ws = [l.weight.data.contiguous() for l in linears]
    bs = [l.bias.data.float().contiguous() for l in linears]        # the GEMM epilogue takes fp32 biases
    wp = (C.c_void_p * len(ws))(*[t.data_ptr() for t in ws])
    bp = (C.c_void_p * len(bs))(*[t.data_ptr() for t in bs])
    _lib.check(ctx.lib.pgv_projector(ctx.handle, _lib.dtype_code(dtype), len(ws), wp, bp, linears[0].in_features, hidden,
                                     x2.data_ptr(), rows, y.data_ptr(), _lib.stream_ptr(w0.device)), "pgv_projector")
    return y.reshape(*lead, hidden)


def build_vision_projector(config, delay_load=False, dtype=torch.float16, device=None, **kwargs):
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "linear":
        return HipLinear(config.mm_hidden_size, config.hidden_size, dtype, device)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        depth = int(m.group(1))
        mods = [HipLinear(config.mm_hidden_size, config.hidden_size, dtype, device)]
        for _ in range(1, depth):
            mods.append(_GELUMarker())
            mods.append(HipLinear(config.hidden_size, config.hidden_size, dtype, device))
        return HipProjectorMLP(*mods)
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")
