#!/usr/bin/env python3
"""Kernel microbenchmarks at the BASELINE shapes (run on the MI355X box): per-kernel TFLOP/s or GB/s with HIP events,
interleaved rounds, median/min.  Usage: python scripts/microbench.py [gemm] [attn] [gemv] [dattn] [norm]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from video_llava_amd import _lib  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def bench_gemm(ctx, dtype=torch.bfloat16):
    print("== gemm (M,N,K,epi) median/min ms, TF/s(median) ==")
    frames = 400
    M = frames * 257
    shapes = [(M, 3072, 1024, _lib.EPI_BIAS, "vit qkv"), (M, 1024, 1024, _lib.EPI_BIAS_RESID, "vit out_proj"),
              (M, 4096, 1024, _lib.EPI_BIAS_QGELU, "vit fc1"), (M, 1024, 4096, _lib.EPI_BIAS_RESID, "vit fc2"),
              (25700, 3072, 1024, _lib.EPI_BIAS, "vit qkv 1 clip"), (25700, 1024, 4096, _lib.EPI_BIAS_RESID, "vit fc2 1 clip"),
              (3632, 12288, 4096, _lib.EPI_NONE, "llm qkv B=8"), (3632, 22016, 4096, _lib.EPI_SWIGLU, "llm gate/up"),
              (3632, 4096, 11008, _lib.EPI_RESID, "llm down"), (8192, 8192, 8192, _lib.EPI_NONE, "8192^3")]
    for (m, n, k, epi, name) in shapes:
        a = torch.randn(m, k, device=DEV).to(dtype)
        w = (torch.randn(n, k, device=DEV) * 0.02).to(dtype)
        bias = torch.randn(n, device=DEV) if epi in (_lib.EPI_BIAS, _lib.EPI_BIAS_RESID, _lib.EPI_BIAS_QGELU) else None
        if epi in (_lib.EPI_RESID, _lib.EPI_BIAS_RESID):
            out = torch.zeros(m, n, device=DEV)
        elif epi == _lib.EPI_SWIGLU:
            out = torch.empty(m, n // 2, device=DEV, dtype=dtype)
        else:
            out = torch.empty(m, n, device=DEV, dtype=dtype)
        med, mn = timeit(lambda: ctx.gemm(a, w, bias, epi, out=out))
        print(f"{name:18s} {m:7d} {n:6d} {k:6d} epi{epi}: {med:8.3f} / {mn:8.3f} ms  {2.0 * m * n * k / med / 1e9:8.1f} TF/s")
        del a, w, out


def bench_gemm_ablate(ctx, dtype=torch.bfloat16):
    """gemm_w4 with PGV_GEMM_ABLATE set by the caller (bit0 no DMA, bit1 no ds_reads, bit2 no MFMA, bit3 no counted wait, bit4 no barrier;
    plain BIAS epilogue only): timing ablations, results are garbage."""
    print("== gemm ablation (ABLATE=%s) ==" % os.environ.get("PGV_GEMM_ABLATE", "0"))
    for (m, n, k, name) in ((205600, 3072, 1024, "vit qkv"), (205600, 1024, 4096, "fc2 shape"), (8192, 8192, 8192, "8192^3"), (16384, 4096, 16384, "long K")):
        a = torch.randn(m, k, device=DEV).to(dtype)
        w = (torch.randn(n, k, device=DEV) * 0.02).to(dtype)
        bias = torch.randn(n, device=DEV)
        out = torch.empty(m, n, device=DEV, dtype=dtype)
        med, mn = timeit(lambda: ctx.gemm(a, w, bias, _lib.EPI_BIAS, out=out))
        print(f"{name:10s} {m:7d} {n:6d} {k:6d}: {med:8.3f} / {mn:8.3f} ms  {2.0 * m * n * k / med / 1e9:8.1f} TF/s-equivalent")


def bench_vit_attn(ctx, dtype=torch.bfloat16):
    import ctypes as C
    print("== vit attention ==")
    lib = ctx.lib
    for T, N in ((400, 257), (100, 257), (100, 577)):
        qkv = torch.randn(T * N, 3072, device=DEV).to(dtype)
        out = torch.empty(T * N, 1024, device=DEV, dtype=dtype)
        f = lambda: _lib.check(lib.pgv_vit_attention(ctx.handle, _lib.dtype_code(dtype), qkv.data_ptr(), out.data_ptr(), T, N, 1024, 16, _lib.stream_ptr()))
        med, mn = timeit(f)
        fl = 4.0 * T * 16 * N * N * 64
        print(f"T={T} N={N}: {med:8.3f} / {mn:8.3f} ms  {fl / med / 1e9:8.1f} TF/s")


def bench_decode(ctx, dtype=torch.bfloat16):
    import ctypes as C
    lib = ctx.lib
    print("== decode gemv (mode, N, K, B) median/min us, GB/s ==")
    # rotate through several weight copies so the 256 MB Infinity Cache cannot serve the stream
    for (mode, N, K, name) in ((0, 12288, 4096, "qkv"), (1, 4096, 4096, "o_proj"), (2, 22016, 4096, "gate/up"), (1, 4096, 11008, "down"),
                               (3, 32003, 4096, "lm_head")):
        for B in (8, 16):
            ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
            ws = []
            Np = (N + 15) // 16 * 16
            for _ in range(ncopy):
                src = (torch.randn(N, K, device=DEV) * 0.02).to(dtype)
                dst = torch.zeros(Np, K, device=DEV, dtype=dtype)
                _lib.check(lib.pgv_pack_blocked(ctx.handle, _lib.dtype_code(dtype), src.data_ptr(), N, K, dst.data_ptr(), _lib.stream_ptr()))
                ws.append(dst)
            torch.cuda.synchronize()
            x = torch.randn(B, K, device=DEV).to(dtype)
            if mode == 1:
                out = torch.zeros(B, N, device=DEV)
            elif mode == 3:
                out = torch.empty(B, N, device=DEV)
            elif mode == 2:
                out = torch.empty(B, N // 2, device=DEV, dtype=dtype)
            else:
                out = torch.empty(B, N, device=DEV, dtype=dtype)
            state = {"i": 0}

            def f():
                w = ws[state["i"] % ncopy]; state["i"] += 1
                _lib.check(lib.pgv_gemv(ctx.handle, _lib.dtype_code(dtype), mode, w.data_ptr(), x.data_ptr(), K, out.data_ptr(),
                                            out.shape[1], N, K, B, _lib.stream_ptr()))
            med, mn = timeit(f, iters=40)
            print(f"{name:8s} mode{mode} N={N:6d} K={K:6d} B={B:2d}: {med * 1e3:8.1f} / {mn * 1e3:8.1f} us  {N * K * 2 / med / 1e6:8.1f} GB/s")
            del ws


def bench_decode_wide(ctx, dtype=torch.bfloat16):
    """The consumer GEMVs (qkv, gate/up, lm_head) at 8 .. 64 sequences, HBM-cold weights (rotating copies): us per launch and TB/s of weight bytes.
    With the lab library (PGV_LIB=lab) PGV_GEMV_ABLATE = 1 (no x loads) / 2 (no MFMA) / 4 (no weight loads) attributes the time."""
    lib = ctx.lib
    print(f"== decode gemv, wide batches (PGV_GEMV_ABLATE={os.environ.get('PGV_GEMV_ABLATE', '0')} PGV_XPAD={os.environ.get('PGV_XPAD', '0')} PGV_GEMV_XBLK={os.environ.get('PGV_GEMV_XBLK', '0')}) ==")
    shapes = ((0, 12288, 4096, "qkv"), (2, 22016, 4096, "gate/up"), (3, 32003, 4096, "lm_head"))
    if os.environ.get("PGV_WIDE_13B"):
        shapes = ((0, 15360, 5120, "qkv13"), (2, 27648, 5120, "gate/up13"))
    if os.environ.get("PGV_WIDE_PROD"):                    # the residual producers on the 16-row kernel (mode 1: resid += W x; no 8-phase form): what does blocked x buy them?
        shapes = ((1, 4096, 4096, "o_proj"), (1, 4096, 11008, "down"), (1, 5120, 5120, "o_proj13"), (1, 5120, 13824, "down13"))
    for (mode, N, K, name) in shapes:
        ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
        Np = (N + 15) // 16 * 16
        ws = []
        for _ in range(ncopy):
            src = (torch.randn(N, K, device=DEV) * 0.02).to(dtype)
            dst = torch.zeros(Np, K, device=DEV, dtype=dtype)
            _lib.check(lib.pgv_pack_blocked(ctx.handle, _lib.dtype_code(dtype), src.data_ptr(), N, K, dst.data_ptr(), _lib.stream_ptr()))
            ws.append(dst)
        torch.cuda.synchronize()
        xpad = int(os.environ.get("PGV_XPAD", "0"))
        xblk = os.environ.get("PGV_GEMV_XBLK", "0") != "0"
        for B in (8, 16, 32, 48, 64):
            xbuf = torch.randn(B, K + xpad, device=DEV).to(dtype)
            x = xbuf[:, :K]
            ldx = K + xpad
            if xblk and B > 16:                              # [K/32][CT][4 k-groups][16 sequences][8]: one contiguous 1 KiB wave-load per B fragment
                CT = 2 if B <= 32 else 4
                xp = torch.zeros(CT * 16, K, device=DEV, dtype=dtype)
                xp[:B] = x
                x = xp.view(CT, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()
                ldx = K
            out = torch.empty(B, N, device=DEV) if mode in (1, 3) else torch.empty(B, N // 2 if mode == 2 else N, device=DEV, dtype=dtype)
            if mode == 1:
                out.zero_()
            state = {"i": 0}

            def f():
                w = ws[state["i"] % ncopy]; state["i"] += 1
                _lib.check(lib.pgv_gemv(ctx.handle, _lib.dtype_code(dtype), mode, w.data_ptr(), x.data_ptr(), ldx, out.data_ptr(), out.shape[1], N, K, B, _lib.stream_ptr()))
            med, mn = timeit(f, iters=40)
            print(f"{name:9s} N={N:6d} K={K:5d} B={B:2d}: {med * 1e3:7.1f} / {mn * 1e3:7.1f} us  {N * K * 2 / med / 1e9:6.2f} TB/s", flush=True)
        del ws


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"gemm", "attn", "gemv"}
    if os.environ.get("PGV_LIB") == "lab":
        _lib.use_lab_build()
    if "ablate" in which or os.environ.get("PGV_ATTN_ABLATE"):
        _lib.use_lab_build()               # the ablation switches exist only in libpgv_lab.so (-DPGV_LAB)
    ctx = _lib.Context.get(0)
    if "gemm" in which:
        bench_gemm(ctx)
    if "ablate" in which:
        bench_gemm_ablate(ctx)
    if "attn" in which:
        bench_vit_attn(ctx)
    if "gemv" in which:
        bench_decode(ctx)
    if "gemvwide" in which:
        bench_decode_wide(ctx)


def bench_gemm_pad(ctx, dtype=torch.bfloat16):
    """Does a non-power-of-two leading dimension change the DMA-bound rate?  (L2 channel camping test)"""
    print("== gemm leading-dimension padding test ==")
    for (m, n, k, name) in ((102800, 3072, 1024, "vit qkv"), (102800, 4096, 1024, "vit fc1"), (102800, 1024, 4096, "vit fc2-shape"),
                            (8192, 8192, 8192, "8192^3")):
        for pad_a, pad_w in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (128, 128)):
            abuf = torch.randn(m, k + pad_a, device=DEV).to(dtype)
            wbuf = (torch.randn(n, k + pad_w, device=DEV) * 0.02).to(dtype)
            a, w = abuf[:, :k], wbuf[:, :k]
            bias = torch.randn(n, device=DEV)
            out = torch.empty(m, n, device=DEV, dtype=dtype)
            med, mn = timeit(lambda: ctx.gemm(a, w, bias, _lib.EPI_BIAS, out=out))
            print(f"{name:14s} padA={pad_a:3d} padW={pad_w:3d}: {med:8.3f} / {mn:8.3f} ms  {2.0 * m * n * k / med / 1e9:8.1f} TF/s")
            del abuf, wbuf, out


if "pad" in sys.argv[1:]:
    bench_gemm_pad(_lib.Context.get(0))


def bench_gemm_data(ctx, dtype=torch.bfloat16):
    """DVFS check: the same GEMM on zeros / small integers / random data (MI355X clocks to its power budget)."""
    print("== gemm data-dependence (8192^3 and vit qkv) ==")
    for (m, n, k, name) in ((8192, 8192, 8192, "8192^3"), (102800, 3072, 1024, "vit qkv")):
        for fill in ("zeros", "ones", "randn", "uniform"):
            if fill == "zeros":
                a = torch.zeros(m, k, device=DEV, dtype=dtype); w = torch.zeros(n, k, device=DEV, dtype=dtype)
            elif fill == "ones":
                a = torch.ones(m, k, device=DEV, dtype=dtype); w = torch.ones(n, k, device=DEV, dtype=dtype)
            elif fill == "randn":
                a = torch.randn(m, k, device=DEV).to(dtype); w = (torch.randn(n, k, device=DEV) * 0.02).to(dtype)
            else:
                a = (torch.rand(m, k, device=DEV) * 2 - 1).to(dtype); w = (torch.rand(n, k, device=DEV) * 2 - 1).to(dtype)
            bias = torch.zeros(n, device=DEV)
            out = torch.empty(m, n, device=DEV, dtype=dtype)
            med, mn = timeit(lambda: ctx.gemm(a, w, bias, _lib.EPI_BIAS, out=out))
            print(f"{name:10s} {fill:8s}: {med:8.3f} / {mn:8.3f} ms  {2.0 * m * n * k / med / 1e9:8.1f} TF/s")
            del a, w, out


if "data" in sys.argv[1:]:
    bench_gemm_data(_lib.Context.get(0))


def bench_prefetch(ctx, dtype=torch.bfloat16):
    """Does a weight matrix that was just touched (L2 / Infinity Cache resident) stream faster through the decode GEMV than a cold one?
    Rotates over enough copies to be HBM-cold; 'touch' = a torch reduction over the first `mb` MB right before the GEMV."""
    import ctypes as C
    lib = ctx.lib
    print("== gemv after prefetch (qkv shape N=12288 K=4096 B=8) ==")
    N, K, B = 12288, 4096, 8
    ncopy = 8
    ws = []
    for _ in range(ncopy):
        src = (torch.randn(N, K, device=DEV) * 0.02).to(dtype)
        dst = torch.zeros(N, K, device=DEV, dtype=dtype)
        _lib.check(lib.pgv_pack_blocked(ctx.handle, _lib.dtype_code(dtype), src.data_ptr(), N, K, dst.data_ptr(), _lib.stream_ptr()))
        ws.append(dst)
    x = torch.randn(B, K, device=DEV).to(dtype)
    out = torch.empty(B, N, device=DEV, dtype=dtype)
    for mb in (0, 8, 32, 100):
        ts = []
        for it in range(24):
            w = ws[it % ncopy]
            if mb:
                n_el = mb * 1024 * 1024 // 2
                w.view(-1)[:n_el].view(torch.int16).max()          # reads the bytes through the caches
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.check(lib.pgv_gemv(ctx.handle, _lib.dtype_code(dtype), 0, w.data_ptr(), x.data_ptr(), K, out.data_ptr(), N, N, K, B, _lib.stream_ptr()))
            b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        print(f"prefetched {mb:3d} MB: gemv median {ts[len(ts) // 2] * 1e3:7.1f} us  min {ts[0] * 1e3:7.1f} us")


if "prefetch" in sys.argv[1:]:
    bench_prefetch(_lib.Context.get(0))
