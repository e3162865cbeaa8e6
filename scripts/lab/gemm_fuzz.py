#!/usr/bin/env python3
"""Random-shape sweep of pgv_gemm against torch fp32 (all epilogues, fp16 and bf16): M in [1, 1400], N multiple of 8 (64 for SwiGLU) up to 1100,
K multiple of 64 up to 1024; seeded.  Prints the worst case per epilogue; exit code 1 on any violation.  Run on an MI355X:
python scripts/lab/gemm_fuzz.py [cases]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from video_llava_amd import _lib  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main(cases):
    ctx = _lib.Context.get(torch.device(DEV))
    g = torch.Generator().manual_seed(1234)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst, bad = {}, 0
    for c in range(cases):
        dtype = (torch.float16, torch.bfloat16)[c & 1]
        epi = ri(0, 7)
        M, K = ri(1, 1400), 64 * ri(1, 16)
        N = 64 * ri(1, 17) if epi == _lib.EPI_SWIGLU else 8 * ri(1, 137)
        a = torch.randn(M, K, generator=g).to(dtype).to(DEV)
        w = (torch.randn(N, K, generator=g) * 0.06).to(dtype).to(DEV)
        b = torch.randn(N, generator=g).to(DEV)
        pre = a.float() @ w.float().t()
        tol = 1.5e-3 if dtype == torch.float16 else 8e-3
        if epi == _lib.EPI_NONE:
            out, ref = ctx.gemm(a, w, None, epi), pre
        elif epi == _lib.EPI_BIAS:
            out, ref = ctx.gemm(a, w, b, epi), pre + b
        elif epi == _lib.EPI_BIAS_QGELU:
            out, ref = ctx.gemm(a, w, b, epi), (pre + b) * torch.sigmoid(1.702 * (pre + b))
        elif epi == _lib.EPI_BIAS_GELU:
            out, ref = ctx.gemm(a, w, b, epi), torch.nn.functional.gelu(pre + b)
        elif epi in (_lib.EPI_RESID, _lib.EPI_BIAS_RESID):
            r0 = torch.randn(M, N, generator=g).to(DEV)
            out = r0.clone()
            ctx.gemm(a, w, b if epi == _lib.EPI_BIAS_RESID else None, epi, out=out)
            ref, tol = r0 + pre + (b if epi == _lib.EPI_BIAS_RESID else 0), 2e-5
        elif epi == _lib.EPI_SWIGLU:
            x = pre.view(M, N // 64, 2, 32)
            out, ref = ctx.gemm(a, w, None, epi), (torch.nn.functional.silu(x[:, :, 0]) * x[:, :, 1]).reshape(M, N // 2)
        else:
            out, ref, tol = ctx.gemm(a, w, b, epi), pre + b, 2e-5
        torch.cuda.synchronize()
        e = rel(out, ref)
        key = (epi, str(dtype))
        if e > worst.get(key, (0,))[0]:
            worst[key] = (e, M, N, K)
        if not (e <= tol) or not torch.isfinite(out.float()).all():
            bad += 1
            print(f"VIOLATION epi {epi} {dtype} M={M} N={N} K={K}: rel {e:.3e} > {tol}")
    for k in sorted(worst):
        print(k, "worst rel %.3e at M=%d N=%d K=%d" % worst[k])
    print(f"{cases} cases, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300))
