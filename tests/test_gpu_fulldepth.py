"""GPU: decoder parity at BASELINE size and FULL DEPTH against the fp32 CPU oracle (reference
video_chatgpt/model/video_chatgpt.py:193-251 driven over 32 / 40 layers by the greedy loop of video_chatgpt/inference.py:105-112).

  * 7B (config 3): PG-Video-LLaVA-7B shapes, 32 layers, fp16 (the reference's dtype), one 100-frame-shaped prompt (356 video rows, 441
    tokens): prefill logits <= TOL_7B normwise, then 16 FREE-RUNNING greedy tokens token-exact.  The oracle's top-1/top-2 margin of
    every step is asserted above the floor first (seeds searched with `python -m oracle.fulldepth search 7b`), so the token comparison
    can never be skipped.
  * 13B fp8 (config 5): 40 layers, e4m3 weights with per-row power-of-two scales, fp16 activations; the oracle runs on the DEQUANTISED
    weights read back from the library (pgv_llm_get_weight: what prefill and decode actually multiply with; one matrix is also checked
    against the CPU twin of the quantiser).  Same bar as 7B: logits bounded, 16 free-running tokens exact, margin floor asserted.
    (bf16 activations -- the bench dtype -- carry 8x the rounding: measured 7.4e-2 normwise on the logits at 40 layers against 1.6e-2 at 2
    layers, i.e. the same sqrt(depth) growth; at that noise a random model's 16-step greedy run has near-ties whatever the seed, so the
    token-exact bar is stated in fp16 and bf16 is bounded at 2 layers in tests/test_gpu_llm.py.)

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

TOL_7B_FP16 = 8.1e-3      # normwise logits error: 1.2 x the 6.72e-3 measured on MI355X (2 layers: 1.84e-3; sixteen-fold depth -> x sqrt(16) = 4:
                          # independent 16-bit roundings of xn / qkv / P / attention out / SwiGLU act per layer; the residual stream itself is fp32)
TOL_13B_FP8_FP16 = 1.33e-2  # 1.2 x the worst decode-step error measured on MI355X (1.105e-2; prefill 6.98e-3)


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


def test_7b_full_depth_fp16_token_exact(ctx):
    c, cfg, w, m = _build("7b", torch.float16)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    t0 = time.time()
    toks_ref, margins, logits_ref = fd.run_oracle(w, cfg, ids, feats, cache_weights=_host_can_cache(6.8e9))
    print(f"[7b] oracle (fp32, {torch.get_num_threads()} threads): {time.time() - t0:.0f}s; margins {[round(x, 3) for x in margins]}")
    assert min(margins) > c["floor"], f"oracle margin {min(margins)} below the floor {c['floor']}: pick another seed (oracle/fulldepth.py)"
    kv, nxt, lg = m.prefill([ids], feats.half(), 512, want_logits=True)
    e = rel(lg[0], logits_ref[0])
    print(f"[7b] 32-layer prefill logits rel err (fp16): {e:.3e}")
    assert e < TOL_7B_FP16
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


def test_13b_full_depth_fp8_weights_fp16_token_exact(ctx):
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
    print(f"[13b fp8] 40-layer prefill logits rel err (fp16): {e:.3e}")
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
