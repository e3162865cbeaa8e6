#!/usr/bin/env python3
"""Lab: how much of each ViT GEMM is epilogue?  The 800-frame shapes through the public epilogues of the LAB library, run once per
PGV_GEMM_ABLATE setting (0 = full kernel, 32 = no epilogue, 1 = no DMA, 4 = no MFMA; BIAS epilogue, bf16 only).
    python scripts/lab/gemm_epi_decomp.py            (under PGV_GEMM_ABLATE=...)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_llava_amd import _lib  # noqa: E402

if os.environ.get("PGV_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PGV_LAB_LIB"])
else:
    _lib.use_lab_build()
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from microbench import timeit, DEV  # noqa: E402

ctx = _lib.Context.get(0)
abl = os.environ.get("PGV_GEMM_ABLATE", "0")
M = 800 * 257
dtype = torch.bfloat16
EPIS = {"bias": _lib.EPI_BIAS, "qgelu": _lib.EPI_BIAS_QGELU, "resid": _lib.EPI_BIAS_RESID, "f32": _lib.EPI_F32}
for (n, k, name) in ((3072, 1024, "qkv"), (4096, 1024, "fc1"), (1024, 1024, "out_proj"), (1024, 4096, "fc2")):
    a = torch.randn(M, k, device=DEV).to(dtype)
    w = (torch.randn(n, k, device=DEV) * 0.02).to(dtype)
    bias = torch.randn(n, device=DEV)
    for ename, epi in EPIS.items():
        if abl != "0" and ename != "bias":
            continue
        out = torch.zeros(M, n, device=DEV, dtype=torch.float32 if ename in ("resid", "f32") else dtype)
        med, mn = timeit(lambda: ctx.gemm(a, w, bias, epi, out=out), iters=12)
        tiles = ((M + 255) // 256) * (n // 256)
        rounds = -(-tiles // 256)
        print(f"{os.path.basename(_lib.LIB_PATH):18s} ABLATE={abl:3s} {name:9s} N={n:5d} K={k:5d} epi={ename:6s}: {med * 1e3:8.1f} us (min {mn * 1e3:8.1f})  {2.0 * M * n * k / med / 1e9:7.1f} TF/s  "
              f"{med * 1e3 / rounds:6.2f} us/tile-round ({rounds} rounds)", flush=True)
        del out
    del a, w
