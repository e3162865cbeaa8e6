"""GPU: decoder parity at BASELINE size and FULL DEPTH against the fp32 CPU oracle (reference
video_chatgpt/model/video_chatgpt.py:193-251 driven over 32 / 40 layers by the greedy loop of video_chatgpt/inference.py:105-112).

  * 7B (config 3): PG-Video-LLaVA-7B shapes, 32 layers, fp16 (the reference's dtype), one 100-frame-shaped prompt (356 video rows, 441
    tokens): prefill logits <= TOL_7B normwise, then 16 FREE-RUNNING greedy tokens token-exact.  The oracle's top-1/top-2 margin of
    every step is asserted above the floor first (seeds searched with `python -m oracle.fulldepth search 7b`), so the token comparison
    can never be skipped.
  * 13B fp8 (config 5): 40 layers, e4m3 weights with per-row power-of-two scales, fp16 activations; the oracle runs on the DEQUANTISED
    weights read back from the library (pgv_llm_get_weight: what prefill and decode actually multiply with; one matrix is also checked
    against the CPU twin of the quantiser).  Same bar as 7B: logits bounded, 16 free-running tokens exact, margin floor asserted.
  * bf16 (the dtype bench.py's headline runs in; 7B 32 layers, and 13B 40 layers with fp8 weights = config 5 as benched): bf16 carries 8x the
    rounding, so a free-running 16-token comparison hits near-ties whatever the seed.  Instead 96 / 160 seeded continuation tokens are
    TEACHER-FORCED, every visited position is compared with the fp32 oracle's logits (one causal pass), the argmax is asserted wherever the
    oracle's margin exceeds 6 sigma of the measured per-logit noise, and at least 12 such positions must exist (26 / 20 do).
  * Tolerances are tied to a YARDSTICK: the reference's own VideoChatGPTLlamaForCausalLM run in 16 bits on the host for the same case
    (tests/golden/yardstick.npz, oracle/gen_yardstick.py); this path must stay within 1.25 x the reference's own distance from the fp32
    oracle (measured: 0.59 x fp16, 0.56 x bf16 at 7B; 0.39 x / 0.51 x at 13B fp8), plus regression pins at 1.2 x what it measured.

  * END TO END AT THE BENCHED SIZE (the chain of video_chatgpt/inference.py:47-124 in one piece): 100 synthetic uint8
    frames -> preprocessing -> 23-layer ViT-L/14 -> spatio-temporal pool -> linear projector -> splice -> 32-layer 7B (40-layer 13B fp8)
    -> 256 FREE-RUNNING greedy tokens (the bench's horizon: prompt + 256), with the fp32 oracle chain (ovis.clip_select_features on all 100
    frames -> pool -> ollm.LlamaOracle) evaluated at EVERY visited position in one causal pass over prompt + the HIP path's own tokens.
    By induction, as long as the HIP token is the oracle's argmax the oracle's own free run is the same sequence; `exact prefix` is that
    length (asserted >= 128 in fp16; measured 243 of 256 tokens at 7B and all 256 at 13B fp8, with 255 / 256 resp. 256 / 256 tokens equal to
    the oracle's argmax given the same prefix; in bf16, the benched dtype, 236 / 256 and 239 / 256).  Beyond the
    first near-tie flip the comparison stays meaningful because the oracle is evaluated on the HIP path's prefix: at every position whose
    oracle margin exceeds 6 sigma of the measured per-logit noise the token must be the oracle's argmax, everywhere the chosen token's
    oracle logit must be within that bound of the best, and the logits error is bounded at all 256 positions (every KV position bench.py
    touches).  Why not 64 tokens with every margin above a fixed floor: a random-init model's top-1/top-2 gaps are exponentially
    distributed (mean 0.6 here, 16-bit logit noise 0.03), P(gap > 0.1) = 0.85 per step, 0.85^64 = 3e-5 per seed pair -- not searchable
    with a 60 s oracle; the 16-token searched case above stays as the strict form.
  * THE BENCH'S OWN CALL: bench.Workload.step -- `generate(list of 8 prompts, chunk=64)` on the pooled features of ONE 800-frame tower pass
    -- on 8 distinct clips equals the 8 single-clip runs through video_features + generate(batch 1) bit for bit, and clip 0 equals the
    free-running case above.
  * THE PRODUCTION CONFIGURATION (round 6): 336 px / ViT-L/14-336 / 676 video tokens / mlp2x_gelu end to end through the 32-layer 7B, fp16 (exact
    prefix >= 128, then teacher-forced on to 1024 generated positions = the reference's default max_new_tokens, cache to 1794) and bf16 (bounded),
    and bench.py's `side.image336` call bit for bit against single-clip runs.  The 16-token searched cases judge the HIP tokens with ONE causal
    oracle pass (the oracle's free run by induction).
  * The reference's OWN bf16 teacher-forced argmax agreement (tests/golden/yardstick.npz `*_tf_*`, oracle/gen_yardstick.py tf) is the bar for
    the unfiltered agreement count of the bf16 cases.

The oracle keeps the 16-bit checkpoint tensors and converts per use (or caches fp32 copies when the host has the memory).
"""
import argparse
import gc
import os
import time

import numpy as np
import pytest
import torch

from oracle import fulldepth as fd
from oracle import llm as ollm
from oracle import synth
from oracle import vision as ovis

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Tolerances.  The YARDSTICK is the reference itself: tests/golden/yardstick.npz (oracle/gen_yardstick.py) holds the last-position logits of
# the reference's own VideoChatGPTLlamaForCausalLM run in 16 bits on the host for the same full-depth case, and its normwise error against
# the fp32 oracle.  The HIP path must stay within YARD_FACTOR x the reference's own distance from exact arithmetic.  The REGRESSION pins
# (1.2 x what this path measured on MI355X) stay as a second, tighter bar so that a lost digit cannot hide under the yardstick.
YARD_FACTOR = 1.25
TOL_7B_FP16 = 8.1e-3        # regression pin: 1.2 x 6.72e-3 (2 layers: 1.84e-3; x sqrt(16): independent 16-bit roundings per layer, fp32 residual)
TOL_13B_FP8_FP16 = 1.33e-2  # regression pin: 1.2 x the worst decode step measured (1.105e-2; prefill 6.98e-3)
# end-to-end cases: worst logits error over the 256 free-running positions (1.2 x measured on MI355X) and the exact prefix of the fp16 runs
# (measured, gpurun_out/r5a: 7B fp16 worst 2.75e-3, exact prefix 243 of 256 tokens, 255 / 256 == the oracle's argmax; 7B bf16 worst 2.86e-2, 236 / 256;
#  13B fp8 fp16 worst 4.85e-3, exact prefix 256 = the whole run; 13B fp8 bf16 worst 4.97e-2, 239 / 256)
E2E_PIN_7B_FP16 = float(os.environ.get("PGV_E2E_PIN", 3.3e-3))
E2E_PIN_7B_BF16 = float(os.environ.get("PGV_E2E_PIN", 3.5e-2))
E2E_PIN_13B_FP16 = float(os.environ.get("PGV_E2E_PIN", 5.9e-3))
E2E_PIN_13B_BF16 = float(os.environ.get("PGV_E2E_PIN", 6.0e-2))
E2E_PREFIX_7B_FP16 = int(os.environ.get("PGV_E2E_PREFIX", 128))     # >= 64 is the bar; 243 measured: the first 16-bit near-tie flip of this clip / prompt pair
E2E_PREFIX_13B_FP16 = int(os.environ.get("PGV_E2E_PREFIX", 128))    # 256 measured


def _yard(golden_dir, case, truth_logits):
    """(the reference's own 16-bit error, recomputed against THIS run's oracle logits; must reproduce the stored figure)."""
    import os
    y = np.load(os.path.join(golden_dir, "yardstick.npz"))
    ref = torch.from_numpy(y[f"{case}_ref_logits"])
    assert torch.allclose(torch.as_tensor(truth_logits).float().cpu()[::97], torch.from_numpy(y[f"{case}_truth_sub"]), rtol=1e-3, atol=2e-3), \
        "the fixture does not belong to this case (oracle logits differ)"
    e = rel(ref, truth_logits)
    assert abs(e - float(y[f"{case}_ref_err"])) < 0.05 * e + 1e-4, (e, float(y[f"{case}_ref_err"]))
    return e


def rel(a, b) -> float:
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _host_can_cache(n_params: float) -> bool:
    import psutil
    return psutil.virtual_memory().available > 4.0 * n_params * 1.6


_CASE: dict = {}


def _case(name, dtype, fp8=False, image=224):
    """(c, cfg, w, m) of a full-depth case, ONE alive at a time: consecutive tests of the same case share the seeded weights, the loaded model
    and -- fp8 -- the dequantised read-back (the oracle's weights); asking for another case frees the previous one first (13B: 26 GB of host
    tensors + 39 GB of device memory).  `state` carries results from one test of the case to the next (the free-running tokens)."""
    key = (name, dtype, fp8, image)
    if _CASE.get("key") != key:
        _CASE.clear()
        gc.collect()
        torch.cuda.empty_cache()
        c, cfg, w, m = _build(name, dtype, image)
        if fp8:
            m.quantize_weights_fp8()
            assert m.is_fp8
            t0 = time.time()
            for k in list(w):
                if k == "lm_head.weight" or any(f".{n}." in k for n in ollm.FP8_KEYS):
                    deq = m.get_weight(k).to(dtype).cpu()
                    if dtype == torch.float16 and k.endswith("layers.17.mlp.down_proj.weight"):     # one matrix against the CPU twin of the quantiser
                        assert torch.equal(deq.float(), ollm.quantize_e4m3_rows(w[k].float()))
                        _CASE["twin_checked"] = True
                    w[k] = deq
            gc.collect()
            print(f"[{name}] dequantised weights read back in {time.time() - t0:.0f}s")
        _CASE.update(key=key, val=(c, cfg, w, m), state={})
    return _CASE["val"] + (_CASE["state"],)


_VIS: dict = {}


def _vision_case(image=224):
    """The vision side shared by every end-to-end case: ViT-L/14 weights at `image` px (seed 0, rounded to fp16 -- released CLIP checkpoints are
    16-bit), 8 distinct 100-frame uint8 clips (clip 0 = synth.make_frames(100, image, seed=0)), and the fp32 ORACLE chain on all 100 frames of clip
    0: hidden_states[-2][:, 1:] [100, 256 | 576, 1024] and the pooled [356 | 676, 1024] features (video_chatgpt/inference.py:86-95).  One size alive
    at a time (the 336-px clips are 271 MB, the raw oracle states 236 MB)."""
    if _VIS.get("image") != image:
        _VIS.clear()
        gc.collect()
        ccfg = synth.CLIP_L14_224 if image == 224 else synth.CLIP_L14_336
        cw = synth.quantize_weights(synth.make_clip_weights(ccfg, seed=0), "float16")
        clips = [synth.make_frames(100, image, seed=k) for k in range(8)]
        t0 = time.time()
        with torch.no_grad():
            px = ovis.clip_preprocess(clips[0])
            raw = torch.cat([ovis.clip_select_features(px[i:i + 25], cw, ccfg) for i in range(0, 100, 25)])
            pooled = ovis.spatio_temporal_pool_torch(raw)
        print(f"[vision oracle] 100 frames of {image} px x 23 layers in fp32: {time.time() - t0:.0f}s ({torch.get_num_threads()} threads)")
        _VIS.update(image=image, ccfg=ccfg, cw=cw, clips=clips, raw=raw, pooled=pooled, towers={})
    return _VIS


def _tower(dtype, image=224):
    from helpers import make_tower
    v = _vision_case(image)
    if dtype not in v["towers"]:
        v["towers"][dtype] = make_tower(v["ccfg"], v["cw"], dtype)
    return v["towers"][dtype]


def _bench_prompts(image=224):
    from video_llava_amd import benchlib
    return benchlib.make_prompts(8, 32003, 100 + (image // 14) ** 2, seed=5)            # what benchlib.Workload builds for rank 0 of 1


def _build(name, dtype, image=224):
    """image = 336: the PRODUCTION configuration of the released weights (reference docs/1-CLI_DEMO.md:27-44): VisionConfig(frame_size=336) ->
    build_vision_projector(mlp2x_gelu) (multimodal_projector/builder.py:39-46), 676 video tokens.  The decoder tensors are the 224-px case's
    (every tensor has its own seeded stream; the projector's come last)."""
    import dataclasses
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    c = fd.CASES[name]
    cfg = c["cfg"] if image == 224 else dataclasses.replace(c["cfg"], projector="mlp2x_gelu")
    t0 = time.time()
    w = synth.make_llama_weights_16bit(cfg, seed=c["weight_seed"], head_std=c["head_std"], dtype="float16" if dtype == torch.float16 else "bfloat16")
    t_gen = time.time() - t0
    hc = VideoChatGPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                            num_attention_heads=cfg.heads, rms_norm_eps=cfg.eps, rope_theta=cfg.rope_theta, eos_token_id=None,
                            mm_projector_type=cfg.projector)
    m = VideoChatGPTLlamaForCausalLM(hc, VisionConfig(frame_size=image), dtype, torch.device(DEV))
    t0 = time.time()
    for k, v in w.items():
        m.load_state_dict({k: v}, strict=False)
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1, True
    print(f"[{name}] weights generated in {t_gen:.0f}s, loaded in {time.time() - t0:.0f}s")
    return c, cfg, w, m


# ---------------------------------------------------------------------------------------------------------------------------------
# end-to-end helpers
# ---------------------------------------------------------------------------------------------------------------------------------
N_FREE = 256                                   # the bench's decode horizon (BASELINE configs[2..4]: 256 generated tokens)
SIGMAS = 6.0


def _hip_features(dtype, clip_u8, image=224):
    """frames -> pooled [356 | 676, 1024] fp16 through the product's per-clip chain (video_llava_amd.inference.video_features ==
    video_chatgpt/inference.py:86-95): fused uint8 preprocessing, 23-layer tower, hidden_states[-2][:, 1:], spatio-temporal pool."""
    from video_llava_amd.inference import video_features
    return video_features(torch.from_numpy(clip_u8).to(DEV), _tower(dtype, image), None)


def _free_run_check(tag, m, cfg, w, ids, feats_hip, feats_ref, dtype, *, min_decisive, min_exact_prefix, pin_worst, sigma_cap=None, n_tail=0, tail_seed=0):
    """N_FREE free-running greedy tokens of the HIP path (its own argmax fed back, logits kept at every step) -- then, with n_tail > 0, n_tail more
    positions TEACHER-FORCED on seeded tokens (the reference's default horizon, max_new_tokens = 1024, video_chatgpt/inference.py:111: N_FREE + n_tail
    = 1024 generated positions; decode attention, RoPE and the cache far beyond the bench's 256) -- then the fp32 oracle chain in ONE causal pass over
    prompt + all of those tokens with the ORACLE's features.  Returns (free-running tokens, stats)."""
    n_all = N_FREE + n_tail
    kv, nxt, lg = m.prefill([ids], feats_hip, len(ids) + n_all + 8, want_logits=True)
    toks, L = [int(nxt[0])], [lg[0].clone()]
    for _ in range(N_FREE - 1):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        toks.append(int(nxt[0]))
        L.append(lg[0].clone())
    fed = list(toks[:-1])                                  # what followed the prompt in the cache so far
    if n_tail:
        tail = np.random.default_rng([777, tail_seed]).integers(3, 32000, n_tail - 1).tolist()
        step_in = [toks[-1]] + tail                        # n_tail further steps: the last free token, then the seeded ones
        dev_in = torch.tensor(step_in, dtype=torch.int32, device=DEV)
        for i in range(n_tail):
            _n, lg = m.decode_step(kv, dev_in[i:i + 1], want_logits=True)
            L.append(lg[0].clone())
        fed += step_in
    L = torch.stack(L).float().cpu()
    assert L.shape[0] == n_all and len(fed) == n_all - 1 and m.ctx.lib.pgv_kv_len(kv, 0) == len(ids) + n_all - 1
    # the same run through generate() (device-side greedy loop replayed from hipGraphs, chunks of 32): prompt echoed, same 256 tokens
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats_hip[None], do_sample=False, max_new_tokens=N_FREE, eos_token_id=None)
    assert out[0, :len(ids)].tolist() == list(ids) and out[0, len(ids):].tolist() == toks, f"[{tag}] generate() differs from the stepwise free run"
    t0 = time.time()
    lg_ref, margins, arg_ref = fd.teacher_forced_reference(w, cfg, ids, feats_ref.float(), fed)
    print(f"[{tag}] fp32 oracle chain, one causal pass over {len(ids) + n_all - 1} tokens: {time.time() - t0:.0f}s")
    assert lg_ref.shape[0] == n_all
    d = L - lg_ref
    errs = (d.double().norm(dim=-1) / lg_ref.double().norm(dim=-1)).tolist()
    sigma = d.std(dim=-1).tolist()
    got = toks + L[N_FREE:].argmax(-1).tolist()            # free run: the token the path emitted; tail: the argmax of its logits
    decisive = agree = 0
    prefix = None
    worst_deficit = 0.0
    for i in range(n_all):
        thr = SIGMAS * sigma[i]
        ok = got[i] == arg_ref[i]
        agree += int(ok) if i < N_FREE else 0
        if not ok and prefix is None and i < N_FREE:
            prefix = i
        if margins[i] > thr:
            decisive += 1
            assert ok, f"[{tag}] step {i}: token {got[i]} != oracle argmax {arg_ref[i]} at margin {margins[i]:.3f} > {SIGMAS} sigma = {thr:.3f}"
        deficit = float(lg_ref[i, arg_ref[i]] - lg_ref[i, got[i]])
        worst_deficit = max(worst_deficit, deficit / max(thr, 1e-9))
        assert deficit <= thr, f"[{tag}] step {i}: the chosen token sits {deficit:.3f} below the oracle's best, more than {SIGMAS} sigma = {thr:.3f}"
    prefix = N_FREE if prefix is None else prefix
    tail_agree = sum(int(got[i] == arg_ref[i]) for i in range(N_FREE, n_all))
    print(f"[{tag}] {N_FREE} free-running tokens{f' + {n_tail} teacher-forced positions (cache to {len(ids) + n_all - 1})' if n_tail else ''}: logits rel err first {errs[0]:.3e} / "
          f"worst {max(errs):.3e} (position {int(np.argmax(errs))}){f', worst of the tail {max(errs[N_FREE:]):.3e}' if n_tail else ''}; per-logit noise "
          f"sigma median {float(np.median(sigma)):.4f} max {max(sigma):.4f}; oracle margin median {float(np.median(margins)):.3f}; decisive positions "
          f"(margin > {SIGMAS} sigma) {decisive}/{n_all}, all exact; token == oracle argmax at {agree}/{N_FREE}{f' (tail: {tail_agree}/{n_tail})' if n_tail else ''}; exact prefix "
          f"(== the oracle's own free run) {prefix} tokens; worst chosen-token deficit {worst_deficit:.2f} of the bound")
    assert decisive >= min_decisive * n_all // N_FREE, f"[{tag}] only {decisive} decisive positions: the measured noise is too high for the comparison to mean anything"
    if sigma_cap is not None:
        assert max(sigma) < sigma_cap, (max(sigma), sigma_cap)
    assert prefix >= min_exact_prefix, f"[{tag}] the free run leaves the oracle's own greedy sequence after {prefix} tokens"
    assert max(errs) < pin_worst, (max(errs), pin_worst)
    return toks, dict(errs=errs, prefix=prefix, decisive=decisive, agree=agree, tail_agree=tail_agree)


def _e2e(tag, name, dtype, fp8, image=224, **bars):
    c, cfg, w, m, state = _case(name, dtype, fp8, image)
    v = _vision_case(image)
    ids = _bench_prompts(image)[0]
    feats_hip = _hip_features(dtype, v["clips"][0], image)
    assert tuple(feats_hip.shape) == (100 + (image // 14) ** 2, 1024)
    e = rel(feats_hip, v["pooled"])
    print(f"[{tag}] pooled features of 100 frames vs the fp32 oracle chain: {e:.3e}")
    assert e < (1e-3 if dtype == torch.float16 else 8e-3)
    toks, st = _free_run_check(tag, m, cfg, w, ids, feats_hip, v["pooled"], dtype, **bars)
    state["free_run_tokens"] = toks
    return st


def _searched_16_token_case(tag, name, dtype, fp8, golden_dir, yard_case, pin):
    """The strict form: 16 FREE-RUNNING tokens, token-exact, every oracle margin above the searched floor.  The HIP path runs first; the fp32 oracle
    then judges its tokens in ONE causal pass (as long as each token is the oracle's argmax the oracle's own free run is the same sequence, so
    `argmax == tokens` at all 16 positions IS token-exactness against the oracle's greedy decode -- round 5 ran the oracle's 16 sequential steps
    instead, twice the host time for the same statement)."""
    c, cfg, w, m, _ = _case(name, dtype, fp8)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    kv, nxt, lg = m.prefill([ids], feats.to(dtype), 512, want_logits=True)
    toks, L = [int(nxt[0])], [lg[0].clone()]
    for _ in range(fd.N_NEW - 1):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        toks.append(int(nxt[0]))
        L.append(lg[0].clone())
    kv, nxt, _ = m.prefill([ids], feats.to(dtype), 512)
    assert [int(nxt[0])] + m.decode_greedy(kv, nxt, fd.N_NEW - 1)[0].tolist() == toks              # the device-side loop, same tokens
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats.to(dtype)[None], do_sample=False, max_new_tokens=fd.N_NEW)
    assert out[0, :len(ids)].tolist() == ids and out[0, len(ids):].tolist() == toks
    t0 = time.time()
    lg_ref, margins, arg_ref = fd.teacher_forced_reference(w, cfg, ids, feats, toks[:-1])
    print(f"[{tag}] oracle (fp32, {torch.get_num_threads()} threads), one causal pass: {time.time() - t0:.0f}s; margins {[round(x, 3) for x in margins]}")
    assert min(margins) > c["floor"], f"oracle margin {min(margins)} below the floor {c['floor']}: pick another seed (oracle/fulldepth.py)"
    assert toks == arg_ref, (toks, arg_ref, margins)
    e = rel(L[0], lg_ref[0])
    e_ref = _yard(golden_dir, yard_case, lg_ref[0])
    worst = max(rel(L[i], lg_ref[i]) for i in range(fd.N_NEW))
    print(f"[{tag}] prefill logits rel err: {e:.3e}; the reference's own 16-bit run: {e_ref:.3e}; worst decode-step logits rel err: {worst:.3e}")
    assert e < YARD_FACTOR * e_ref, (e, e_ref)          # the yardstick: no further from exact arithmetic than the reference itself (x 1.25)
    assert e < pin and worst < pin                      # regression pin


# ---------------------------------------------------------------------------------------------------------------------------------
# 7B fp16 (config 3 in the reference's dtype)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_vision_chain_100_frames_all_vs_oracle(ctx):
    """All 100 frames of the benched clip against the fp32 oracle (until round 5 only frames 0-7 had met fp32; the rest was covered by
    bitwise split invariance): per-frame hidden_states[-2][:, 1:] and the pooled features, fp16 <= 1e-3 and bf16 <= 8e-3."""
    v = _vision_case()
    for dtype, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        tower = _tower(dtype)
        px = ctx.preprocess_u8(torch.from_numpy(v["clips"][0]).to(DEV), dtype)
        feat = tower(px, output_hidden_states=True).hidden_states[-2][:, 1:]
        per_frame = [rel(feat[t], v["raw"][t]) for t in range(100)]
        print(f"100 frames {dtype}: per-frame rel err vs fp32 oracle max {max(per_frame):.3e} (frame {int(np.argmax(per_frame))}), all {rel(feat, v['raw']):.3e}")
        assert max(per_frame) < tol
        assert rel(ctx.st_pool(feat), v["pooled"]) < tol


def test_7b_full_depth_fp16_token_exact(ctx, golden_dir):
    _searched_16_token_case("7b", "7b", torch.float16, False, golden_dir, "7b_fp16", TOL_7B_FP16)


def test_7b_config3_end_to_end_fp16_256_free_running_tokens(ctx):
    """BASELINE config 3 as video_chatgpt/inference.py:47-124 runs it, in the reference's dtype: 100 uint8 frames -> ... -> 256 free-running greedy
    tokens; see the module docstring for the criterion."""
    st = _e2e("7b fp16 e2e", "7b", torch.float16, False, min_decisive=200, min_exact_prefix=E2E_PREFIX_7B_FP16, pin_worst=E2E_PIN_7B_FP16, sigma_cap=0.02)
    assert st["agree"] >= 250        # 255 measured


def _bench_call_check(tag, name, dtype, fp8, image=224):
    """benchlib.Workload.step -- the call bench.py times -- on this case's seeded weights and 8 distinct clips."""
    from video_llava_amd import benchlib
    from video_llava_amd.inference import video_features
    c, cfg, w, m, state = _case(name, dtype, fp8, image)
    v = _vision_case(image)
    a = argparse.Namespace(dtype="fp16" if dtype == torch.float16 else "bf16", llm=name, image=image, weights="fp8" if fp8 else "16bit", workload="full",
                           clips_per_gpu=8, frames=100, new_tokens=N_FREE)
    tower = _tower(dtype, image)
    wl = benchlib.Workload(a, torch.device(DEV), 0, 1, tower=tower, model=m)
    assert wl.prompts == _bench_prompts(image) and wl.video_rows == 100 + (image // 14) ** 2 and wl.projector == ("linear" if image == 224 else "mlp2x_gelu")
    wl.frames = torch.from_numpy(np.concatenate(v["clips"])).to(DEV)                      # 8 distinct clips instead of the device-RNG frames
    toks = wl.step(N_FREE, collate=False)                                                  # ingest -> ONE 800-frame tower pass -> pool -> generate(8 prompts, chunk=64)
    assert tuple(toks.shape) == (8, N_FREE)
    toks = toks.cpu().tolist()
    assert len({tuple(t) for t in toks}) == 8, "8 distinct clips / prompts must give 8 distinct answers"
    for b in range(8):                                                                     # the reference-shaped single-clip call (inference.py:86-112)
        feats_b = video_features(wl.frames[100 * b:100 * (b + 1)], tower, None)
        ids = wl.prompts[b]
        out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats_b.unsqueeze(0), do_sample=False, max_new_tokens=N_FREE, eos_token_id=None)
        assert out[0, len(ids):].tolist() == toks[b], f"[{tag}] clip {b}: the batched bench call differs from the single-clip run"
    if "free_run_tokens" in state:
        assert toks[0] == state["free_run_tokens"], f"[{tag}] clip 0 of the bench call differs from the oracle-checked free run"
    else:
        print(f"[{tag}] (the end-to-end free-run test of this case did not run before: clip 0 not cross-checked)")
    print(f"[{tag}] bench.Workload.step on 8 clips == 8 single-clip runs, bit for bit; clip 0 == the oracle-checked free run")


def test_7b_bench_call_fp16_equals_single_clip_runs(ctx):
    _bench_call_check("7b fp16 bench call", "7b", torch.float16, False)


# ---------------------------------------------------------------------------------------------------------------------------------
# 7B bf16 (the benched dtype)
# ---------------------------------------------------------------------------------------------------------------------------------
def _teacher_forced_check(tag, m, cfg, w, ids, feats, cont, dtype, golden_dir, yard_case, pin_prefill, pin_worst, n_min=12, sigmas=6.0):
    """Teacher-force `cont` through the HIP model and compare every visited position with the fp32 oracle's logits (one causal pass):
    logits error bounded by the yardstick, and at every position whose oracle top-1/top-2 margin exceeds `sigmas` x the MEASURED per-logit
    noise of this run (std over the vocabulary of hip - oracle at that position) the argmax must agree.  At least n_min positions must
    qualify -- so the check cannot pass vacuously: a broken kernel inflates the noise, fewer positions qualify, the count assert fails.
    The UNFILTERED agreement count must reach the reference's own (its 16-bit forward over the same tokens, tests/golden/yardstick.npz
    `<case>_tf_ref_argmax`, recounted against this run's oracle argmax)."""
    t0 = time.time()
    lg_ref, margins, arg_ref = fd.teacher_forced_reference(w, cfg, ids, feats, cont)
    print(f"[{tag}] fp32 oracle, one causal pass over {len(ids) + len(cont)} tokens: {time.time() - t0:.0f}s")
    kv, nxt, lg = m.prefill([ids], feats.to(dtype), len(ids) + len(cont) + 8, want_logits=True)
    errs, checked, agree_all = [], 0, 0
    worst = 0.0
    for i in range(len(cont) + 1):
        d = lg[0].float().cpu() - lg_ref[i]
        e = float(d.double().norm() / lg_ref[i].double().norm())
        sigma = float(d.std())
        errs.append(e)
        got = int(lg[0].argmax())
        agree_all += int(got == arg_ref[i])
        if margins[i] > sigmas * sigma:
            checked += 1
            assert got == arg_ref[i], (tag, i, got, arg_ref[i], margins[i], sigma)
        worst = max(worst, e)
        if i < len(cont):
            nxt, lg = m.decode_step(kv, torch.tensor([cont[i]], dtype=torch.int32, device=DEV), want_logits=True)
    e_ref = _yard(golden_dir, yard_case, lg_ref[0])
    y = np.load(os.path.join(golden_dir, "yardstick.npz"))
    ref_arg = y[f"{yard_case}_tf_ref_argmax"]
    assert len(ref_arg) == len(arg_ref) and (y[f"{yard_case}_tf_truth_argmax"] == np.asarray(arg_ref)).mean() > 0.97, "the fixture does not belong to this case"
    ref_agree = int((ref_arg == np.asarray(arg_ref)).sum())
    print(f"[{tag}] prefill logits rel err {errs[0]:.3e}, worst of {len(errs)} teacher-forced positions {worst:.3e}; the reference's own "
          f"16-bit run (prefill): {e_ref:.3e}; argmax checked at {checked} positions (margin > {sigmas} sigma), all agree; "
          f"unfiltered agreement {agree_all}/{len(errs)}; the reference's own 16-bit run: {ref_agree}/{len(errs)}")
    assert checked >= n_min, f"only {checked} positions had a margin above {sigmas} sigma of the measured noise"
    assert errs[0] < YARD_FACTOR * e_ref, (errs[0], e_ref)          # the yardstick (prefill position: the one the reference run covers)
    assert errs[0] < pin_prefill and worst < pin_worst, (errs[0], worst)          # regression pins: 1.2 x measured on MI355X
    assert agree_all >= ref_agree, f"[{tag}] unfiltered argmax agreement {agree_all} is below the reference's own {ref_agree}"
    return errs[0], worst, e_ref


def test_7b_full_depth_bf16_teacher_forced(ctx, golden_dir):
    """The BENCHED dtype at full depth (bench.py's headline line is bf16): 32 layers, bf16 weights and activations, 441-token prompt with 356
    video rows, then 96 teacher-forced positions against the fp32 oracle on the same bf16-valued weights.  bf16 carries 8x fp16's rounding, so
    a free-running 16-token comparison would hit near-ties whatever the seed; instead every position is compared and the argmax is
    asserted wherever the oracle's margin clears 6 sigma of the measured noise, at least 12 such positions required."""
    c, cfg, w, m, _ = _case("7b", torch.bfloat16)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    cont = fd.teacher_tokens(cfg, c["prompt_seed"], fd.N_TEACHER["7b"])
    # measured on MI355X: prefill 5.45e-2 (the reference's own bf16 run: 9.81e-2), worst of the 97 positions 6.54e-2, 26 positions checked
    _teacher_forced_check("7b bf16", m, cfg, w, ids, feats, cont, torch.bfloat16, golden_dir, "7b_bf16", pin_prefill=6.6e-2, pin_worst=7.9e-2)


def test_7b_config3_end_to_end_bf16_256_free_running_tokens(ctx):
    """Config 3 end to end in the BENCHED dtype: bf16 noise is 8x fp16's, so fewer positions are decisive; same criterion."""
    st = _e2e("7b bf16 e2e", "7b", torch.bfloat16, False, min_decisive=100, min_exact_prefix=0, pin_worst=E2E_PIN_7B_BF16, sigma_cap=0.15)
    assert st["agree"] >= 220        # 236 measured (the reference's own bf16 run agrees at 80 / 97 = 82 % of teacher-forced positions; this is 92 %)


def test_7b_bench_call_bf16_equals_single_clip_runs(ctx):
    """The headline configuration of bench.py itself (bf16, 7B, 8 clips x 100 frames, 256 tokens, chunk 64)."""
    _bench_call_check("7b bf16 bench call", "7b", torch.bfloat16, False)


# ---------------------------------------------------------------------------------------------------------------------------------
# 13B fp8 weights (config 5)
# ---------------------------------------------------------------------------------------------------------------------------------
def test_13b_full_depth_fp8_weights_fp16_token_exact(ctx, golden_dir):
    """BASELINE config 5 at full depth: 13B shapes, 40 layers, e4m3 weights (per-row power-of-two scales), fp16 activations (the reference's
    dtype); oracle = fp32 on the dequantised weights the library reports.  Free-running, token-exact, margin floor asserted."""
    _case("13b", torch.float16, fp8=True)
    assert _CASE.get("twin_checked")
    _searched_16_token_case("13b fp8", "13b", torch.float16, True, golden_dir, "13b_fp8_fp16", TOL_13B_FP8_FP16)


def test_13b_config5_end_to_end_fp8_fp16_256_free_running_tokens(ctx):
    """Config 5 end to end (13B, 40 layers, e4m3 weights, fp16 activations): 100 frames -> 256 free-running tokens vs the fp32 oracle chain on the
    dequantised weights."""
    st = _e2e("13b fp8 fp16 e2e", "13b", torch.float16, True, min_decisive=190, min_exact_prefix=E2E_PREFIX_13B_FP16, pin_worst=E2E_PIN_13B_FP16, sigma_cap=0.035)
    assert st["agree"] >= 250        # 256 measured


def test_13b_full_depth_fp8_weights_bf16_teacher_forced(ctx, golden_dir):
    """BASELINE config 5 as bench.py runs it: 13B shapes, 40 layers, e4m3 weights, BF16 activations.  Oracle = fp32 on the dequantised weights
    the library reports; same noise-aware teacher-forced check as the 7B bf16 case over 160 positions."""
    c, cfg, w, m, _ = _case("13b", torch.bfloat16, fp8=True)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    cont = fd.teacher_tokens(cfg, c["prompt_seed"], fd.N_TEACHER["13b"])
    # measured on MI355X: prefill 6.73e-2 (reference bf16 on the same dequantised weights: 1.33e-1), worst of 161 positions 9.44e-2, 20 checked
    _teacher_forced_check("13b fp8 bf16", m, cfg, w, ids, feats, cont, torch.bfloat16, golden_dir, "13b_fp8_bf16", pin_prefill=8.1e-2, pin_worst=1.14e-1)


def test_13b_config5_end_to_end_fp8_bf16_256_free_running_tokens(ctx):
    """Config 5 end to end as bench.py's side line runs it (bf16 activations)."""
    st = _e2e("13b fp8 bf16 e2e", "13b", torch.bfloat16, True, min_decisive=60, min_exact_prefix=0, pin_worst=E2E_PIN_13B_BF16, sigma_cap=0.3)
    assert st["agree"] >= 220        # 239 measured (the reference's own bf16 run on these weights: 107 / 161 = 66 %; this is 93 %)


def test_13b_bench_call_fp8_bf16_equals_single_clip_runs(ctx):
    """bench.py's `side.cfg5_13b_fp8` call: 13B, fp8 weights, bf16, 8 clips, 256 tokens."""
    _bench_call_check("13b fp8 bf16 bench call", "13b", torch.bfloat16, True)


# ---------------------------------------------------------------------------------------------------------------------------------
# The PRODUCTION configuration of the released weights, end to end (VERDICT r5 #1a; reference docs/1-CLI_DEMO.md:27-44,
# video_chatgpt/model/multimodal_projector/builder.py:39-46, video_chatgpt/inference.py:86-112): 100 x 336^2 uint8 frames -> ingest -> ViT-L/14-336
# (23 layers, 577 tokens per frame) -> pool [676, 1024] -> mlp2x_gelu -> splice (771-token prompt) -> 32-layer 7B -> 256 free-running tokens,
# against the fp32 oracle chain (ovis.clip_select_features at CLIP_L14_336 on all 100 frames, ovis.mm_projector("mlp2x_gelu"), ollm.LlamaOracle).
# The fp16 case goes on, teacher-forced, to the reference's DEFAULT generation length (max_new_tokens = 1024, inference.py:111): cache positions
# up to 771 + 1023 = 1794 -- decode attention's key loop, RoPE angles and the cache far past the bench's horizon (VERDICT r5 #1b).
# These run last: they swap the shared vision case to 336 px.
# ---------------------------------------------------------------------------------------------------------------------------------
# measured on MI355X (gpurun_out/r6s, profiles/r06_s_fulldepth_end_to_end.log): fp16 pooled features 1.84e-4, logits first 1.76e-3 / worst 2.38e-3 over all 1024
# positions (cache to 1794), 968 / 1024 decisive positions all exact, 255 / 256 free tokens == the oracle's argmax (tail 759 / 768), exact prefix 188; bf16
# pooled 3.48e-3, logits worst 2.44e-2, 138 / 256 decisive, 243 / 256, prefix 40.  Pins = 1.2 x measured.
E2E_PIN_336_FP16 = float(os.environ.get("PGV_E2E_PIN", 2.9e-3))
E2E_PIN_336_BF16 = float(os.environ.get("PGV_E2E_PIN", 3.0e-2))


def test_7b_production_336_end_to_end_fp16_256_free_running_then_teacher_forced_to_1024(ctx):
    st = _e2e("7b 336 fp16 e2e", "7b", torch.float16, False, image=336, min_decisive=180, min_exact_prefix=E2E_PREFIX_7B_FP16, pin_worst=E2E_PIN_336_FP16,
              sigma_cap=0.03, n_tail=1024 - N_FREE, tail_seed=336)
    assert st["agree"] >= 245 and st["tail_agree"] >= 730          # of 256 / 768


def test_7b_production_336_bench_call_fp16_equals_single_clip_runs(ctx):
    _bench_call_check("7b 336 fp16 bench call", "7b", torch.float16, False, image=336)


def test_7b_production_336_end_to_end_bf16_256_free_running_tokens(ctx):
    """`side.image336` of bench.py is timed in bf16: same chain, the benched dtype's criterion (bounded logits, argmax where the oracle is decisive)."""
    st = _e2e("7b 336 bf16 e2e", "7b", torch.bfloat16, False, image=336, min_decisive=90, min_exact_prefix=0, pin_worst=E2E_PIN_336_BF16, sigma_cap=0.2)
    assert st["agree"] >= 215


def test_7b_production_336_bench_call_bf16_equals_single_clip_runs(ctx):
    """bench.py's `side.image336` call itself: Workload(image=336).step on 8 distinct clips == 8 single-clip runs, clip 0 == the oracle-checked run."""
    _bench_call_check("7b 336 bf16 bench call", "7b", torch.bfloat16, False, image=336)
    _CASE.clear()
    _VIS.clear()
    gc.collect()
