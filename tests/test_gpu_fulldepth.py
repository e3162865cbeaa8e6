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

The oracle keeps the 16-bit checkpoint tensors and converts per use (or caches fp32 copies when the host has the memory).
"""
import gc
import time

import numpy as np
import pytest
import torch

from oracle import fulldepth as fd
from oracle import llm as ollm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# Tolerances.  The YARDSTICK is the reference itself: tests/golden/yardstick.npz (oracle/gen_yardstick.py) holds the last-position logits of
# the reference's own VideoChatGPTLlamaForCausalLM run in 16 bits on the host for the same full-depth case, and its normwise error against
# the fp32 oracle.  The HIP path must stay within YARD_FACTOR x the reference's own distance from exact arithmetic.  The REGRESSION pins
# (1.2 x what this path measured on MI355X) stay as a second, tighter bar so that a lost digit cannot hide under the yardstick.
YARD_FACTOR = 1.25
TOL_7B_FP16 = 8.1e-3        # regression pin: 1.2 x 6.72e-3 (2 layers: 1.84e-3; x sqrt(16): independent 16-bit roundings per layer, fp32 residual)
TOL_13B_FP8_FP16 = 1.33e-2  # regression pin: 1.2 x the worst decode step measured (1.105e-2; prefill 6.98e-3)


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


def _build(name, dtype):
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    c = fd.CASES[name]
    cfg = c["cfg"]
    t0 = time.time()
    w = fd.make_weights(name, "float16" if dtype == torch.float16 else "bfloat16")
    t_gen = time.time() - t0
    hc = VideoChatGPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                            num_attention_heads=cfg.heads, rms_norm_eps=cfg.eps, rope_theta=cfg.rope_theta, eos_token_id=None)
    m = VideoChatGPTLlamaForCausalLM(hc, VisionConfig(frame_size=224), dtype, torch.device(DEV))
    t0 = time.time()
    for k, v in w.items():
        m.load_state_dict({k: v}, strict=False)
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1, True
    print(f"[{name}] weights generated in {t_gen:.0f}s, loaded in {time.time() - t0:.0f}s")
    return c, cfg, w, m


def test_7b_full_depth_fp16_token_exact(ctx, golden_dir):
    c, cfg, w, m = _build("7b", torch.float16)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    t0 = time.time()
    toks_ref, margins, logits_ref = fd.run_oracle(w, cfg, ids, feats, cache_weights=_host_can_cache(6.8e9))
    print(f"[7b] oracle (fp32, {torch.get_num_threads()} threads): {time.time() - t0:.0f}s; margins {[round(x, 3) for x in margins]}")
    assert min(margins) > c["floor"], f"oracle margin {min(margins)} below the floor {c['floor']}: pick another seed (oracle/fulldepth.py)"
    kv, nxt, lg = m.prefill([ids], feats.half(), 512, want_logits=True)
    e = rel(lg[0], logits_ref[0])
    e_ref = _yard(golden_dir, "7b_fp16", logits_ref[0])
    print(f"[7b] 32-layer prefill logits rel err (fp16): {e:.3e}; the reference's own fp16 run: {e_ref:.3e}")
    assert e < YARD_FACTOR * e_ref, (e, e_ref)          # the yardstick: no further from exact arithmetic than the reference itself (x 1.25)
    assert e < TOL_7B_FP16                              # regression pin
    toks = [int(nxt[0])] + m.decode_greedy(kv, nxt, fd.N_NEW - 1)[0].tolist()
    assert toks == toks_ref, (toks, toks_ref, margins)
    # the same prompt through generate(): prompt echoed, same tokens
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats.half()[None], do_sample=False, max_new_tokens=fd.N_NEW)
    assert out[0, :len(ids)].tolist() == ids and out[0, len(ids):].tolist() == toks_ref
    # per-step logits of the free run stay within the bound too (teacher forcing is implied: the tokens are identical)
    kv, nxt, lg = m.prefill([ids], feats.half(), 512, want_logits=True)
    worst = 0.0
    for i in range(1, fd.N_NEW):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        worst = max(worst, rel(lg[0], logits_ref[i]))
    print(f"[7b] worst decode-step logits rel err: {worst:.3e}")
    assert worst < TOL_7B_FP16


def test_13b_full_depth_fp8_weights_fp16_token_exact(ctx, golden_dir):
    """BASELINE config 5 at full depth: 13B shapes, 40 layers, e4m3 weights (per-row power-of-two scales), fp16 activations (the reference's
    dtype); oracle = fp32 on the dequantised weights the library reports.  Free-running, token-exact, margin floor asserted."""
    c, cfg, w, m = _build("13b", torch.float16)
    m.quantize_weights_fp8()
    assert m.is_fp8
    # the oracle's weights: dequantised matrices as the kernels see them (exactly representable in 16 bits), everything else as loaded
    t0 = time.time()
    twin_checked = False
    for k in list(w):
        if k == "lm_head.weight" or any(f".{n}." in k for n in ollm.FP8_KEYS):
            deq = m.get_weight(k).to(torch.float16).cpu()
            if not twin_checked and k.endswith("layers.17.mlp.down_proj.weight"):          # one matrix against the CPU twin of the quantiser
                assert torch.equal(deq.float(), ollm.quantize_e4m3_rows(w[k].float()))
                twin_checked = True
            w[k] = deq
    gc.collect()
    assert twin_checked
    print(f"[13b] dequantised weights read back in {time.time() - t0:.0f}s")
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    t0 = time.time()
    toks_ref, margins, logits_ref = fd.run_oracle(w, cfg, ids, feats, cache_weights=_host_can_cache(13.2e9))
    print(f"[13b] oracle (fp32 on dequantised weights): {time.time() - t0:.0f}s; margins {[round(x, 3) for x in margins]}")
    assert min(margins) > c["floor"], f"oracle margin {min(margins)} below the floor {c['floor']}: pick another seed (oracle/fulldepth.py)"
    kv, nxt, lg = m.prefill([ids], feats.half(), 512, want_logits=True)
    e = rel(lg[0], logits_ref[0])
    e_ref = _yard(golden_dir, "13b_fp8_fp16", logits_ref[0])
    print(f"[13b fp8] 40-layer prefill logits rel err (fp16): {e:.3e}; the reference's own fp16 run on the same dequantised weights: {e_ref:.3e}")
    assert e < YARD_FACTOR * e_ref, (e, e_ref)
    assert e < TOL_13B_FP8_FP16
    toks = [int(nxt[0])] + m.decode_greedy(kv, nxt, fd.N_NEW - 1)[0].tolist()
    assert toks == toks_ref, (toks, toks_ref, margins)
    kv, nxt, lg = m.prefill([ids], feats.half(), 512, want_logits=True)
    worst = 0.0
    for i in range(1, fd.N_NEW):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        worst = max(worst, rel(lg[0], logits_ref[i]))
    print(f"[13b fp8] worst decode-step logits rel err: {worst:.3e}")
    assert worst < TOL_13B_FP8_FP16


def _teacher_forced_check(tag, m, cfg, w, ids, feats, cont, dtype, golden_dir, yard_case, pin_prefill, pin_worst, n_min=12, sigmas=6.0):
    """Teacher-force `cont` through the HIP model and compare every visited position with the fp32 oracle's logits (one causal pass):
    logits error bounded by the yardstick, and at every position whose oracle top-1/top-2 margin exceeds `sigmas` x the MEASURED per-logit
    noise of this run (std over the vocabulary of hip - oracle at that position) the argmax must agree.  At least n_min positions must
    qualify -- so the check cannot pass vacuously: a broken kernel inflates the noise, fewer positions qualify, the count assert fails."""
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
    print(f"[{tag}] prefill logits rel err {errs[0]:.3e}, worst of {len(errs)} teacher-forced positions {worst:.3e}; the reference's own "
          f"16-bit run (prefill): {e_ref:.3e}; argmax checked at {checked} positions (margin > {sigmas} sigma), all agree; "
          f"unfiltered agreement {agree_all}/{len(errs)}")
    assert checked >= n_min, f"only {checked} positions had a margin above {sigmas} sigma of the measured noise"
    assert errs[0] < YARD_FACTOR * e_ref, (errs[0], e_ref)          # the yardstick (prefill position: the one the reference run covers)
    assert errs[0] < pin_prefill and worst < pin_worst, (errs[0], worst)          # regression pins: 1.2 x measured on MI355X
    return errs[0], worst, e_ref


def test_7b_full_depth_bf16_teacher_forced(ctx, golden_dir):
    """The BENCHED dtype at full depth (bench.py's headline line is bf16): 32 layers, bf16 weights and activations, 441-token prompt with 356
    video rows, then 96 teacher-forced positions against the fp32 oracle on the same bf16-valued weights.  bf16 carries 8x fp16's rounding, so
    a free-running 16-token comparison would hit near-ties whatever the seed; instead every position is compared and the argmax is
    asserted wherever the oracle's margin clears 6 sigma of the measured noise, at least 12 such positions required."""
    c, cfg, w, m = _build("7b", torch.bfloat16)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    cont = fd.teacher_tokens(cfg, c["prompt_seed"], fd.N_TEACHER["7b"])
    # measured on MI355X: prefill 5.45e-2 (the reference's own bf16 run: 9.81e-2), worst of the 97 positions 6.54e-2, 26 positions checked
    _teacher_forced_check("7b bf16", m, cfg, w, ids, feats, cont, torch.bfloat16, golden_dir, "7b_bf16", pin_prefill=6.6e-2, pin_worst=7.9e-2)


def test_13b_full_depth_fp8_weights_bf16_teacher_forced(ctx, golden_dir):
    """BASELINE config 5 as bench.py runs it: 13B shapes, 40 layers, e4m3 weights, BF16 activations.  Oracle = fp32 on the dequantised weights
    the library reports; same noise-aware teacher-forced check as the 7B bf16 case over 160 positions."""
    c, cfg, w, m = _build("13b", torch.bfloat16)
    m.quantize_weights_fp8()
    for k in list(w):
        if k == "lm_head.weight" or any(f".{n}." in k for n in ollm.FP8_KEYS):
            w[k] = m.get_weight(k).to(torch.bfloat16).cpu()
    gc.collect()
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    cont = fd.teacher_tokens(cfg, c["prompt_seed"], fd.N_TEACHER["13b"])
    # measured on MI355X: prefill 6.73e-2 (reference bf16 on the same dequantised weights: 1.33e-1), worst of 161 positions 9.44e-2, 20 checked
    _teacher_forced_check("13b fp8 bf16", m, cfg, w, ids, feats, cont, torch.bfloat16, golden_dir, "13b_fp8_bf16", pin_prefill=8.1e-2, pin_worst=1.14e-1)
