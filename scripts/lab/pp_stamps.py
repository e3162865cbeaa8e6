#!/usr/bin/env python3
"""Lab: decode the s_memtime stamps of the pp GEMM (PGV_GEMM_CFG=4 PGV_GEMM_ABLATE=8): per-phase segment cycles per wave."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_llava_amd import _lib
ctx = _lib.Context.get(0)
M = N = 4096; K = 8192
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ctx.gemm(a, w, bias, _lib.EPI_BIAS, out=out)
torch.cuda.synchronize()
raw = out.view(-1)[:8 * 64 * 4].view(torch.int16).cpu().numpy().view(np.uint64).reshape(8, 64).astype(np.int64)
t0 = raw.min()
st = raw.reshape(8, 4, 4, 4) - t0          # wave, kstep, phase, stamp(T0 start, T2 after mid barrier, T3 after compute, T4 after end barrier)
np.set_printoptions(linewidth=200)
for wv in range(8):
    print(f"wave {wv} (row {wv >> 2}):")
    for ks in range(4):
        segs = []
        for ph in range(4):
            s = st[wv, ks, ph]
            segs.append(f"[load+bar {s[1]-s[0]:4d} | mfma {s[2]-s[1]:4d} | bar {s[3]-s[2]:4d}]")
        nxt = st[wv, ks + 1, 0, 0] if ks < 3 else None
        print(f"  kstep {ks}: start {st[wv, ks, 0, 0]:6d} " + " ".join(segs) + (f"  total {nxt - st[wv, ks, 0, 0]}" if nxt is not None else ""))
