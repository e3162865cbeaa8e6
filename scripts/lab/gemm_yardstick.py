#!/usr/bin/env python3
"""Yardstick: what does the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reach at the ViT GEMM shapes of the bench,
next to this repo's gemm_w4 with its plain bias epilogue?  Measurement only -- the product path never calls a vendor GEMM.
Usage (GPU box): python scripts/lab/gemm_yardstick.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_llava_amd import _lib  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ctx = _lib.Context.get(0)
    print("shape (M, N, K)                 vendor matmul      gemm_w4 (+bias)    ratio")
    for dtype in (torch.bfloat16, torch.float16):
        for name, M, N, K in (("vit qkv", 205600, 3072, 1024), ("vit out_proj", 205600, 1024, 1024), ("vit fc1", 205600, 4096, 1024), ("vit fc2", 205600, 1024, 4096),
                              ("vit qkv 400f", 102800, 3072, 1024), ("llm gate/up prefill", 3632, 22016, 4096), ("square 8192", 8192, 8192, 8192)):
            a = torch.randn(M, K, device=DEV).to(dtype)
            w = (torch.randn(N, K, device=DEV) * 0.02).to(dtype)
            bias = torch.randn(N, device=DEV)
            out = torch.empty(M, N, device=DEV, dtype=dtype)
            wt = w.t()
            tv = timeit(lambda: torch.matmul(a, wt, out=out))
            to = timeit(lambda: ctx.gemm(a, w, bias, _lib.EPI_BIAS, out=out))
            fl = 2.0 * M * N * K
            print(f"{str(dtype)[6:]:9s}{name:20s} {M:7d} {N:6d} {K:6d}: {tv:8.3f} ms {fl / tv / 1e9:7.1f} TF/s   {to:8.3f} ms {fl / to / 1e9:7.1f} TF/s   {tv / to:5.2f}x")
            del a, w, out


if __name__ == "__main__":
    main()
