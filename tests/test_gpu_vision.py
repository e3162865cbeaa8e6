"""GPU parity: HIP vision path (GEMM, norms, ViT attention, tower, pooling, projector) vs the CPU oracle,
the committed golden fixtures (generated from the reference) and plain torch fp32 references.

Tolerances (north star: 1e-3 relative in fp16): normwise relative error ||y - ref|| / ||ref||.
  fp16 tower/projector/pool end to end : <= 1e-3
  bf16 (perf dtype, 8 mantissa bits)   : <= 8e-3   (reported, looser by the 8x larger unit roundoff)
"""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle import vision as ovis

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# --------------------------------------------------------------------------------------------------
# GEMM + epilogues
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 256, 64), (300, 512, 128), (257, 1024, 640), (2570, 3072, 1024), (515, 264, 192)])
def test_gemm_bias(ctx, dtype, M, N, K):
    from video_llava_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    ref = a.float() @ w.float().t() + b
    out = ctx.gemm(a, w, b, _lib.EPI_BIAS)
    assert out.shape == (M, N) and out.dtype == dtype
    tol = 1.5e-3 if dtype == torch.float16 else 8e-3
    assert rel(out, ref) < tol
    # asymmetric check: a transposed / row-swapped write would be caught elementwise
    assert (out.float() - ref).abs().max() < 0.05 * ref.abs().max()
    out2 = ctx.gemm(a, w, None, _lib.EPI_NONE)
    assert rel(out2, a.float() @ w.float().t()) < tol
    out32 = ctx.gemm(a, w, b, _lib.EPI_F32)
    assert out32.dtype == torch.float32 and rel(out32, ref) < 2e-5 + (0 if dtype == torch.float16 else 0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_activation_epilogues(ctx, dtype):
    from video_llava_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(3)
    M, N, K = 777, 768, 256
    a = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) * 0.08).to(dtype).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    pre = a.float() @ w.float().t() + b
    tol = 1.5e-3 if dtype == torch.float16 else 8e-3
    assert rel(ctx.gemm(a, w, b, _lib.EPI_BIAS_QGELU), pre * torch.sigmoid(1.702 * pre)) < tol
    assert rel(ctx.gemm(a, w, b, _lib.EPI_BIAS_GELU), torch.nn.functional.gelu(pre)) < tol
    # residual accumulate (fp32 stream)
    r0 = torch.randn(M, N, generator=g).to(DEV)
    r = r0.clone()
    ctx.gemm(a, w, b, _lib.EPI_BIAS_RESID, out=r)
    assert rel(r, r0 + pre) < 1e-5
    r = r0.clone()
    ctx.gemm(a, w, None, _lib.EPI_RESID, out=r)
    assert rel(r, r0 + (pre - b)) < 1e-5
    # SwiGLU: rows interleaved [32 gate | 32 up] per 64
    I = N // 2
    gate = (torch.randn(I, K, generator=g) * 0.08).to(dtype)
    up = (torch.randn(I, K, generator=g) * 0.08).to(dtype)
    wi = torch.stack([gate.view(I // 32, 32, K), up.view(I // 32, 32, K)], dim=1).reshape(N, K).to(DEV)
    ref = torch.nn.functional.silu(a.float() @ gate.float().t().to(DEV)) * (a.float() @ up.float().t().to(DEV))
    out = ctx.gemm(a, wi, None, _lib.EPI_SWIGLU)
    assert out.shape == (M, I) and rel(out, ref) < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4400, 4352, 64), (4609, 4104, 192), (9000, 2304, 320), (70000, 1024, 128)])
def test_gemm_many_tiles(ctx, dtype, M, N, K):
    """More output tiles than CUs (a persistent workgroup walks several tiles; K-steps 1, 3, 5, 2 per tile), ragged M and N
    edges, every store shape (16-bit, fp32, fp32 read-modify-write).  Reference: torch fp32 on the same device."""
    from video_llava_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.1).to(dtype)
    b = torch.randn(N, generator=g, device=DEV)
    ref = a.float() @ w.float().t() + b
    tol = 1.5e-3 if dtype == torch.float16 else 8e-3
    out = ctx.gemm(a, w, b, _lib.EPI_BIAS)
    assert rel(out, ref) < tol
    err = (out.float() - ref).abs()
    assert float(err.max()) < 0.05 * float(ref.abs().max()), f"worst element at {np.unravel_index(int(err.argmax()), err.shape)}"
    r0 = torch.randn(M, N, generator=g, device=DEV)
    r = r0.clone()
    ctx.gemm(a, w, b, _lib.EPI_BIAS_RESID, out=r)
    assert float((r - (r0 + ref)).abs().max()) < 1e-3
    out32 = ctx.gemm(a, w, None, _lib.EPI_F32)
    assert float((out32 - (ref - b)).abs().max()) < 1e-3
    # rows beyond M / columns beyond N of a padded output buffer must stay untouched
    pad = torch.full((M + 3, N + 8), 7.0, dtype=dtype, device=DEV)
    ctx.gemm(a, w, b, _lib.EPI_BIAS, out=pad[:M, :N])
    assert bool((pad[M:] == 7).all()) and bool((pad[:, N:] == 7).all()) and rel(pad[:M, :N], ref) < tol


def test_gemm_rejects_bad_shapes(ctx):
    a = torch.zeros(4, 60, dtype=torch.float16, device=DEV)
    w = torch.zeros(8, 60, dtype=torch.float16, device=DEV)
    with pytest.raises(ValueError):
        ctx.gemm(a, w)          # K not a multiple of 64


# --------------------------------------------------------------------------------------------------
# norms
# --------------------------------------------------------------------------------------------------
# --------------------------------------------------------------------------------------------------
# ViT attention building block (CLIPAttention eager math, HF:clip/modeling_clip.py:259-277) through the C ABI
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("T,N,heads", [(3, 1, 2), (2, 5, 16), (2, 31, 2), (2, 32, 2), (3, 33, 2), (2, 96, 4), (2, 97, 4), (5, 257, 16), (2, 577, 16)])
def test_vit_attention_block(ctx, dtype, tol, T, N, heads):
    """Token counts around every boundary of the kernel: a single key, one partial 32-key block, exact block / chunk multiples (32, 96) and
    one past them (33, 97: the single-valid-key block both CLIP sizes end with), 257 (224 px) and 577 (336 px).  Random q/k/v with distinct
    rows make the check sensitive to any transposition or key-order slip in the LDS images."""
    from video_llava_amd import _lib
    C = heads * 64
    g = torch.Generator().manual_seed(1000 * N + T)
    qkv = (torch.randn(T * N, 3 * C, generator=g) * 1.5).to(dtype)
    out = torch.full((T * N, C), float("nan"), dtype=dtype, device=DEV)
    qd = qkv.to(DEV)
    _lib.check(ctx.lib.pgv_vit_attention(ctx.handle, _lib.dtype_code(dtype), qd.data_ptr(), out.data_ptr(), T, N, C, heads, _lib.stream_ptr()), "pgv_vit_attention")
    torch.cuda.synchronize()
    x = qkv.float().view(T, N, 3, heads, 64)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))            # [T, heads, N, 64]
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(T * N, C)
    assert torch.isfinite(out.float()).all()
    e = rel(out, ref)
    assert e <= tol, f"N={N}: rel err {e:.3e}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cols", [512, 1024, 4096, 5120])
def test_layernorm_rmsnorm(ctx, dtype, cols):
    g = torch.Generator(device="cpu").manual_seed(cols)
    x = (torch.randn(37, cols, generator=g) * 3 + 0.7).to(DEV)
    gam = (1 + 0.1 * torch.randn(cols, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(cols, generator=g)).to(DEV)
    ref = torch.nn.functional.layer_norm(x, (cols,), gam, bet, 1e-5)
    tol = 6e-4 if dtype == torch.float16 else 5e-3
    assert rel(ctx.layernorm(x, gam, bet, 1e-5, dtype), ref) < tol
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * gam
    assert rel(ctx.rmsnorm(x, gam, 1e-6, dtype), ref) < tol


# --------------------------------------------------------------------------------------------------
# spatio-temporal pooling
# --------------------------------------------------------------------------------------------------
def test_pool_golden(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "pool.npz"))
    for name, (T, P) in {"t8_p16": (8, 16), "t100_p16": (100, 16), "t3_p4": (3, 4)}.items():
        rng = np.random.default_rng(int(g[name + "_seed"]))
        f16 = (rng.standard_normal((T, P, 1024), dtype=np.float32) * 1.5).astype(np.float16)
        out = ctx.st_pool(torch.from_numpy(f16).to(DEV)).cpu().numpy()
        assert out.shape == (100 + P, 1024) and out.dtype == np.float16
        for ref in (g[name + "_torch"], g[name + "_numpy"]):
            d = np.abs(out.astype(np.float32) - ref.astype(np.float32))
            assert d.max() <= 2 ** -10 * max(1.0, np.abs(ref).max()), name          # <= 1 fp16 ulp (sum order)
            assert (d == 0).mean() > 0.98, name                                     # and almost always identical
        if T < 100:
            assert not out[T:100].any()


def test_pool_full_size_and_strided_view(ctx):
    """[100, 257, 1024] hidden state, CLS dropped through a strided view (inference.py:94-95)."""
    rng = np.random.default_rng(1)
    hid = torch.from_numpy(rng.standard_normal((100, 257, 1024), dtype=np.float32).astype(np.float16)).to(DEV)
    view = hid[:, 1:]
    out = ctx.st_pool(view)
    ref = ovis.spatio_temporal_pool_torch(view.float().cpu())
    assert out.shape == (356, 1024)
    d = (out.float().cpu() - ref.float()).abs()
    assert d.max() <= 2 ** -10 and float((d == 0).float().mean()) > 0.98
    # linearity property at full size: pool(2x) == 2*pool(x) exactly (power-of-two scaling is exact in fp16/fp32)
    out2 = ctx.st_pool((view * 2).contiguous())
    normal = out.abs() > 2 ** -13                      # doubling commutes with rounding outside fp16's subnormal range
    assert torch.equal(out2[normal], (out * 2)[normal])
    with pytest.raises(ValueError):
        ctx.st_pool(torch.zeros(101, 4, 1024, dtype=torch.float16, device=DEV))   # reference never truncates; we refuse


def test_preprocess_u8(ctx):
    frames = synth.make_frames(3, 224, seed=4)
    ref = ovis.clip_preprocess(frames)
    out = ctx.preprocess_u8(torch.from_numpy(frames).to(DEV), torch.float16)
    assert out.shape == (3, 3, 224, 224)
    assert torch.equal(out.cpu(), ref.half()) or (out.float().cpu() - ref).abs().max() <= 2e-3


# --------------------------------------------------------------------------------------------------
# CLIP tower
# --------------------------------------------------------------------------------------------------
def _tower(cfg: synth.ClipCfg, w: dict, dtype):
    from video_llava_amd.vision_tower import CLIPVisionTower, CLIPVisionTowerConfig
    tc = CLIPVisionTowerConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                               num_attention_heads=cfg.heads, image_size=cfg.image, patch_size=cfg.patch, layer_norm_eps=cfg.eps)
    t = CLIPVisionTower(tc, dtype)
    t.load_state_dict(w)
    return t


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_vit_tiny_golden(ctx, golden_dir, dtype, tol):
    g = np.load(os.path.join(golden_dir, "clip_tiny.npz"))
    cfg = synth.CLIP_TINY
    w = synth.make_clip_weights(cfg, seed=int(g["weight_seed"]))
    tower = _tower(cfg, w, dtype)
    px = ovis.clip_preprocess(synth.make_frames(int(g["n_frames"]), cfg.image, seed=int(g["frame_seed"])))
    out = tower(px.to(dtype).to(DEV), output_hidden_states=True)
    assert len(out.hidden_states) == cfg.layers + 1
    assert rel(out.hidden_states[0], torch.from_numpy(g["hs0"])) < tol
    assert rel(out.hidden_states[1], torch.from_numpy(g["hs1"])) < tol
    feat = out.hidden_states[-2][:, 1:]
    assert feat.shape == (5, cfg.patches, 1024)
    assert rel(feat, torch.from_numpy(g["feat"])) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_vit_l14_config1_vs_oracle(ctx, golden_dir, dtype, tol):
    """BASELINE config 1 shape: ViT-L/14@224, 8 frames, hidden_states[-2][:,1:] -> pool, vs the CPU oracle and the
    HF-generated pooled fixture."""
    cfg = synth.CLIP_L14_224
    w = synth.make_clip_weights(cfg, seed=0)
    tower = _tower(cfg, w, dtype)
    frames = synth.make_frames(8, 224, seed=0)
    px = ovis.clip_preprocess(frames)
    feat = tower(px.to(dtype).to(DEV), output_hidden_states=True).hidden_states[-2][:, 1:]
    ref = ovis.clip_select_features(px, w, cfg)
    e = rel(feat, ref)
    # yardstick (oracle/gen_yardstick.py vit): HF CLIPVisionModel itself, run in this dtype on the host for the same frames and weights, sits at
    # 1.65e-3 (fp16) / 1.32e-2 (bf16) from the fp32 oracle -- the reference's own 16-bit run does not meet the 1e-3 bar this path is held to
    e_ref = float(np.load(os.path.join(golden_dir, "yardstick.npz"))["vit_l14_8f_fp16_ref_err" if dtype == torch.float16 else "vit_l14_8f_bf16_ref_err"])
    print(f"ViT-L/14 8 frames {dtype}: normwise rel err vs fp32 oracle = {e:.3e}; HF CLIPVisionModel in the same dtype: {e_ref:.3e}")
    assert e < tol and e < 1.25 * e_ref
    pooled = ctx.st_pool(feat)
    gold = np.load(os.path.join(golden_dir, "vit_l14_8f_pooled.npz"))["pooled"]
    assert rel(pooled, torch.from_numpy(gold.astype(np.float32))) < tol
    assert not pooled[8:100].any()
    # also through the fused uint8 preprocessing kernel
    px2 = ctx.preprocess_u8(torch.from_numpy(frames).to(DEV), dtype)
    feat2 = tower(px2, output_hidden_states=True).hidden_states[-2][:, 1:]
    assert rel(feat2, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_vit_l14_336_vs_oracle(ctx, dtype, tol):
    """The tower the released PG-Video-LLaVA weights use: ViT-L/14@336, N = 577 tokens per frame, 3 frames; features and the
    pooled [100 + 576, 1024] video tokens vs the CPU oracle."""
    cfg = synth.CLIP_L14_336
    w = synth.make_clip_weights(cfg, seed=3)
    tower = _tower(cfg, w, dtype)
    frames = synth.make_frames(3, 336, seed=4)
    px = ovis.clip_preprocess(frames)
    feat = tower(ctx.preprocess_u8(torch.from_numpy(frames).to(DEV), dtype), output_hidden_states=True).hidden_states[-2][:, 1:]
    assert tuple(feat.shape) == (3, 576, 1024)
    ref = ovis.clip_select_features(px, w, cfg)
    e = rel(feat, ref)
    print(f"ViT-L/14-336 3 frames {dtype}: normwise rel err vs fp32 oracle = {e:.3e}")
    assert e < tol
    pooled = ctx.st_pool(feat)
    assert tuple(pooled.shape) == (676, 1024) and not pooled[3:100].any()
    assert rel(pooled, ovis.spatio_temporal_pool_torch(ref)) < tol


def test_vit_100_frames_properties(ctx):
    """BASELINE config 2 size (100 frames): size-independent properties -- frames are independent, so a batch
    split (the reference's infer_batch=32 loop, save_spatio_temporal_clip_features.py:108-121) or a frame permutation
    must reproduce the same per-frame features bit for bit."""
    cfg = synth.CLIP_L14_224
    w = synth.make_clip_weights(cfg, seed=0)
    tower = _tower(cfg, w, torch.float16)
    px = ctx.preprocess_u8(torch.from_numpy(synth.make_frames(100, 224, seed=0)).to(DEV), torch.float16)
    full = tower(px).hidden_states[-2]
    assert full.shape == (100, 257, 1024) and torch.isfinite(full).all()
    parts = torch.cat([tower(px[i:i + 32]).hidden_states[-2] for i in range(0, 100, 32)])
    assert torch.equal(full, parts)
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(0)).to(DEV)
    assert torch.equal(tower(px[perm]).hidden_states[-2], full[perm])
    # first 8 frames agree with the oracle (same inputs as config 1)
    ref = ovis.clip_select_features(ovis.clip_preprocess(synth.make_frames(100, 224, seed=0)[:8]), w, cfg)
    assert rel(full[:8, 1:], ref) < 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_vit_800_frames_benched_shape(ctx, dtype, tol):
    """The shape `bench.py` times: 8 clips x 100 frames in ONE tower call = two lanes of 400 frames -> 402 tile rows per
    lane, an EVEN row count, so the qkv / fc1 GEMMs take the W-resident tile order of csrc/gemm.hip tile_coords_v with 19 - 25 persistent
    rounds per workgroup -- a branch no 100-frame test enters (2 x 50 frames -> 51 rows, odd -> band order).  Frames are independent, so the
    800-frame pass must equal the eight 100-frame passes (the reference's per-clip call, video_chatgpt/inference.py:93; batching loop
    scripts/save_spatio_temporal_clip_features.py:108-121) BIT FOR BIT; the profiled pass of the bench (pgv_prof_enable forces ONE lane:
    804 tile rows, also even) must too; and the first 8 frames are held to the fp32 oracle like BASELINE config 1."""
    cfg = synth.CLIP_L14_224
    w = synth.make_clip_weights(cfg, seed=0)
    tower = _tower(cfg, w, dtype)
    frames = np.concatenate([synth.make_frames(100, 224, seed=k) for k in range(8)])            # 8 distinct clips; clip 0 = the config-1 frames
    px = ctx.preprocess_u8(torch.from_numpy(frames).to(DEV), dtype)
    full = tower(px).hidden_states[-2]
    assert full.shape == (800, 257, 1024) and torch.isfinite(full).all()
    for k in range(8):
        part = tower(px[100 * k:100 * (k + 1)]).hidden_states[-2]
        assert torch.equal(full[100 * k:100 * (k + 1)], part), f"clip {k}: the 800-frame pass differs from its 100-frame pass"
    ctx.prof_enable(True)                                                                       # the bench's profiled pass: one lane, 804 tile rows
    try:
        single = tower(px).hidden_states[-2]
    finally:
        ctx.prof_enable(False)
        ctx.prof_reset()
    assert torch.equal(single, full)
    ref = ovis.clip_select_features(ovis.clip_preprocess(frames[:8]), w, cfg)
    e = rel(full[:8, 1:], ref)
    print(f"800-frame pass {dtype}: first 8 frames vs fp32 oracle {e:.3e}")
    assert e < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_vit_336_even_row_count(ctx, dtype, tol):
    """336 px (N = 577): 14 frames in one lane = 8078 rows -> 32 tile rows (even -> W-resident order for qkv / fc1); the same frames in
    passes of 3 (1731 rows -> 7 tile rows, odd -> band order) must give the same bits, and the first 3 frames agree with the fp32 oracle."""
    cfg = synth.CLIP_L14_336
    w = synth.make_clip_weights(cfg, seed=3)
    tower = _tower(cfg, w, dtype)
    frames = np.concatenate([synth.make_frames(3, 336, seed=4), synth.make_frames(11, 336, seed=5)])
    px = ctx.preprocess_u8(torch.from_numpy(frames).to(DEV), dtype)
    full = tower(px).hidden_states[-2]
    assert full.shape == (14, 577, 1024)
    parts = torch.cat([tower(px[i:i + 3]).hidden_states[-2] for i in range(0, 14, 3)])
    assert torch.equal(full, parts)
    ref = ovis.clip_select_features(ovis.clip_preprocess(frames[:3]), w, cfg)
    assert rel(full[:3, 1:], ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_vit_336_100_frames_production_shape(ctx, dtype, tol):
    """The PRODUCTION tower of the released PG-Video-LLaVA weights at the size a clip has: ViT-L/14-336, 100 frames x 577
    tokens in one call (two lanes of 50 frames; the attention kernel runs its 8-wave form: K + V of a head fill a CU's LDS).  The pass must equal
    the reference's infer_batch = 32 loop (scripts/save_spatio_temporal_clip_features.py:108-121) bit for bit, a frame permutation must
    permute the output, the first 3 frames are held to the fp32 oracle, and the pooled [100 + 576, 1024] video tokens to the oracle's pooling
    of the same features."""
    cfg = synth.CLIP_L14_336
    w = synth.make_clip_weights(cfg, seed=3)
    tower = _tower(cfg, w, dtype)
    frames = np.concatenate([synth.make_frames(3, 336, seed=4), synth.make_frames(97, 336, seed=6)])
    px = ctx.preprocess_u8(torch.from_numpy(frames).to(DEV), dtype)
    full = tower(px).hidden_states[-2]
    assert full.shape == (100, 577, 1024) and torch.isfinite(full).all()
    parts = torch.cat([tower(px[i:i + 32]).hidden_states[-2] for i in range(0, 100, 32)])
    assert torch.equal(full, parts)
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(1)).to(DEV)
    assert torch.equal(tower(px[perm]).hidden_states[-2], full[perm])
    ref = ovis.clip_select_features(ovis.clip_preprocess(frames[:3]), w, cfg)
    e = rel(full[:3, 1:], ref)
    print(f"336 px, 100 frames {dtype}: first 3 frames vs fp32 oracle {e:.3e}")
    assert e < tol
    pooled = ctx.st_pool(full[:, 1:])
    assert tuple(pooled.shape) == (676, 1024)
    assert rel(pooled[100:], full[:, 1:].float().mean(0)) < 2e-3          # spatial tokens = mean over the 100 frames (fp16 output rounding)
    assert rel(pooled[:3], ref.mean(1)) < tol                             # temporal tokens of the oracle-checked frames


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N", [(102800, 3072), (102800, 4096), (4096, 3072), (4096, 4096), (205600, 3072)])
def test_gemm_w_resident_tile_order(ctx, dtype, M, N):
    """pgv_gemm at the tile grids of the benched ViT pass (K = 1024): M = 102 800 rows = 402 tile rows (one lane of 400 frames), 205 600 =
    804 (the single-lane profiled pass), 4096 = 16; N = 3072 (qkv, 12 tile columns) and 4096 (fc1, 16) -> the W-resident order
    (csrc/gemm.hip tile_coords_v: even row count, >= 8 tile columns) with many row pairs per XCD.  Checked against torch fp32, and against
    the ODD-row-count twin of the same problem (one tile row fewer -> band order): the two orders visit the same tiles with the same K
    loop, so the shared rows must be bitwise equal."""
    from video_llava_amd import _lib
    K = 1024
    g = torch.Generator(device=DEV).manual_seed(M + N)
    a = torch.randn(M, K, generator=g, device=DEV).to(dtype)
    w = (torch.randn(N, K, generator=g, device=DEV) * 0.05).to(dtype)
    b = torch.randn(N, generator=g, device=DEV)
    out = ctx.gemm(a, w, b, _lib.EPI_BIAS)
    tol = 1.5e-3 if dtype == torch.float16 else 8e-3
    worst = 0.0
    for r0 in range(0, M, 25700):                                  # reference in row slabs (fp32 [M, N] at once would be 3.4 GB)
        ref = a[r0:r0 + 25700].float() @ w.float().t() + b
        got = out[r0:r0 + 25700].float()
        assert rel(got, ref) < tol
        worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
    assert worst < 0.05
    M_odd = M - 256                                                # 401 / 803 / 15 tile rows: the band order
    assert ((M + 255) // 256) % 2 == 0 and ((M_odd + 255) // 256) % 2 == 1
    twin = ctx.gemm(a[:M_odd], w, b, _lib.EPI_BIAS)
    assert torch.equal(twin, out[:M_odd])
    q = ctx.gemm(a, w, b, _lib.EPI_BIAS_QGELU)                     # fc1's activation on the same order
    q_twin = ctx.gemm(a[:M_odd], w, b, _lib.EPI_BIAS_QGELU)
    assert torch.equal(q_twin, q[:M_odd])


# --------------------------------------------------------------------------------------------------
# mm_projector (C ABI level; the nn.Module mirror is tested in test_gpu_llm.py)
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("kind,rows", [("linear", 356), ("mlp2x_gelu", 676)])
def test_projector(ctx, dtype, tol, kind, rows):
    import ctypes as C
    from video_llava_amd import _lib
    cfg = synth.LlamaCfg(hidden=4096, projector=kind, layers=0)
    w = synth.make_llama_weights(cfg, seed=9)
    x = torch.randn(rows, 1024, generator=torch.Generator().manual_seed(2))
    ref = ovis.mm_projector(x, w, kind)
    names = ["model.mm_projector"] if kind == "linear" else ["model.mm_projector.0", "model.mm_projector.2"]
    ws = [torch.from_numpy(w[n + ".weight"]).to(dtype).to(DEV) for n in names]
    bs = [torch.from_numpy(w[n + ".bias"]).float().to(DEV) for n in names]
    y = torch.empty(rows, 4096, dtype=dtype, device=DEV)
    wp = (C.c_void_p * len(ws))(*[t.data_ptr() for t in ws])
    bp = (C.c_void_p * len(bs))(*[t.data_ptr() for t in bs])
    xd = x.to(dtype).to(DEV)
    _lib.check(ctx.lib.pgv_projector(ctx.handle, _lib.dtype_code(dtype), len(ws), wp, bp, 1024, 4096, xd.data_ptr(), rows,
                                     y.data_ptr(), _lib.stream_ptr()))
    assert rel(y, ref) < tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("h,w,s", [(360, 640, 224), (360, 640, 336), (20, 20, 56), (481, 853, 224), (224, 224, 224), (112, 112, 224), (1080, 1920, 336)])
def test_ingest_native_resolution_bit_exact(ctx, dtype, h, w, s):
    """pgv_ingest_u8 = load_video's nearest resize (reference eval/model_utils.py:38-43: F.interpolate default mode on the
    float tensor, cast back to uint8) fused with CLIPImageProcessor's normalisation.  Integer-valued inputs, so the bar is BIT-exact
    against the two-step path: torch's own interpolate on the host, then the (already oracle-checked) crop-size preprocessing kernel --
    for down-sampling, up-sampling, non-integer ratios and the identity."""
    from video_llava_amd.feature_extraction import resize_nearest
    frames = np.random.default_rng(h * 3 + s).integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    fused = ctx.ingest_u8(torch.from_numpy(frames).to(DEV), s, dtype)
    two_step = ctx.preprocess_u8(torch.from_numpy(np.ascontiguousarray(resize_nearest(frames, (s, s)))).to(DEV), dtype)
    assert fused.shape == (3, 3, s, s) and torch.equal(fused, two_step)
    # and against the oracle's CLIPImageProcessor restatement (fp32) within one rounding of the 16-bit output
    ref = ovis.clip_preprocess(resize_nearest(frames, (s, s)))
    assert float((fused.float().cpu() - ref).abs().max()) <= (2e-3 if dtype == torch.float16 else 1.6e-2)


def test_native_frames_route_through_ingest(ctx, tmp_path):
    """load_video(..., device_resize=True) -> NativeFrames -> inference.frames_to_pixels takes the fused path and equals the host-resized
    route bit for bit; a NativeFrames whose target is not the tower's crop size falls back to the caller's image_processor."""
    from video_llava_amd import feature_extraction as fx
    from video_llava_amd.inference import frames_to_pixels
    from video_llava_amd.vision_tower import CLIPVisionTower, CLIPVisionTowerConfig
    ccfg = synth.CLIP_TINY
    tower = CLIPVisionTower(CLIPVisionTowerConfig(hidden_size=ccfg.hidden, intermediate_size=ccfg.inter, num_hidden_layers=ccfg.layers,
                                                  num_attention_heads=ccfg.heads, image_size=ccfg.image, patch_size=ccfg.patch), torch.float16)
    clip = np.random.default_rng(4).integers(0, 256, (130, 90, 160, 3), dtype=np.uint8)
    np.save(tmp_path / "clip.npy", clip)
    native = fx.load_video(str(tmp_path / "clip.npy"), shape=(ccfg.image, ccfg.image), device_resize=True)
    assert isinstance(native, fx.NativeFrames) and native.array.shape == (100, 90, 160, 3)
    host = fx.load_video(str(tmp_path / "clip.npy"), shape=(ccfg.image, ccfg.image))
    assert host.shape == (100, ccfg.image, ccfg.image, 3)
    a = frames_to_pixels(native, None, tower)
    b = frames_to_pixels(host, None, tower)
    assert torch.equal(a, b)


@pytest.mark.parametrize("shift,outliers", [(0.0, False), (1.0, False), (5.0, False), (20.0, False), (0.0, True), (5.0, True)])
def test_folded_layernorm_mean_dominated_rows(ctx, shift, outliers):
    """The LayerNorm folded into the CLIP GEMMs (csrc/gemm.hip EPI_LN_*) rounds its operand to 16 bits BEFORE the normalisation.  In the
    centred form the rounded operand is (x - c) gamma with c tracking the row mean, so rows whose mean dominates their spread (|mu| / sigma =
    `shift`) or that carry a few massive channels (as released ViT-L checkpoints do) must come out as accurate as the zero-mean rows of a
    random-init tower.  Construction: layer 0's out_proj bias moves every residual row by +shift (and +80 on four channels with
    `outliers`), the last layer's fc2 bias moves it back, so 11 of the 12 LayerNorms of a 6-layer tower see the shifted rows while the
    16-bit OUTPUT keeps its usual magnitude (otherwise the output rounding itself would swamp the comparison).  fp16, <= 1e-3 normwise vs
    the fp32 oracle, and no worse than 1.5 x the unshifted tower."""
    cfg = synth.ClipCfg(hidden=1024, inter=512, layers=6, heads=16, image=56, patch=14)
    base = synth.make_clip_weights(cfg, seed=5)
    frames = synth.make_frames(7, cfg.image, seed=6)
    px = ovis.clip_preprocess(frames)

    def run(sh, outl):
        w = {k: v.copy() for k, v in base.items()}
        delta = np.full(1024, sh, np.float32)
        if outl:
            delta[[7, 300, 301, 900]] += 80.0
        w["vision_model.encoder.layers.0.self_attn.out_proj.bias"] = w["vision_model.encoder.layers.0.self_attn.out_proj.bias"] + delta
        w[f"vision_model.encoder.layers.{cfg.layers - 1}.mlp.fc2.bias"] = w[f"vision_model.encoder.layers.{cfg.layers - 1}.mlp.fc2.bias"] - delta
        tower = _tower(cfg, w, torch.float16)
        got = tower(px.half().to(DEV), output_hidden_states=True).hidden_states[cfg.layers]
        ref = ovis.clip_hidden_states(px, w, cfg)[cfg.layers]
        mid = ovis.clip_hidden_states(px, w, cfg, upto=2)[2]
        ratio = float((mid.mean(-1).abs() / mid.std(-1)).median())
        return rel(got, ref), ratio

    e0, r0 = run(0.0, False)
    e, r = run(shift, outliers)
    print(f"folded LayerNorm, shift {shift} outliers {outliers}: |mu|/sigma of the rows the LayerNorms see = {r:.2f} (unshifted {r0:.2f}); "
          f"tower error {e:.3e} (unshifted {e0:.3e})")
    if shift > 0 and not outliers:
        assert r > 0.5 * shift
    assert e < 1e-3 and e < 1.5 * e0 + 1e-5
