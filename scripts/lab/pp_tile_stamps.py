#!/usr/bin/env python3
"""Lab: tile-level s_memtime stamps of the pp GEMM (PGV_GEMM_CFG=4 PGV_GEMM_ABLATE=9): K-step starts around the first epilogue."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_llava_amd import _lib
ctx = _lib.Context.get(0)
M, N, K = 102800, 3072, 1024
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ctx.gemm(a, w, bias, _lib.EPI_BIAS, out=out)
torch.cuda.synchronize()
raw = out.view(-1)[:8 * 64 * 4].view(torch.int16).cpu().numpy().view(np.uint64).reshape(8, 64).astype(np.int64)[:, :12]
t0 = raw[:, :8].min()
for wv in range(8):
    r = raw[wv] - t0
    ks = r[:8]
    print(f"wave {wv}: kstep starts (nk-3..nk+4): {ks.tolist()}  deltas {np.diff(ks).tolist()}  epilogue begin {r[8]} end {r[9]} (len {r[9]-r[8]})")
