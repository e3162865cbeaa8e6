#!/usr/bin/env python3
"""The REFERENCE's own 16-bit error at full depth -- the yardstick of tests/test_gpu_fulldepth.py.  TEST INFRASTRUCTURE (oracle/__init__.py).

Runs in the build container only (imports /root/reference + the installed HF transformers on CPU, shims of oracle/gen_golden.py).
For each case it builds the reference's `VideoChatGPTLlamaForCausalLM` (video_chatgpt/model/video_chatgpt.py:177-251) at BASELINE size
in the 16-bit dtype the reference loads checkpoints in (`torch_dtype=torch.float16`, video_chatgpt/eval/model_utils.py:104-105; bf16 as
the variant bench.py runs), feeds the full-depth case of oracle/fulldepth.py (same seeded 16-bit weights, same 441-token prompt with 356
video rows) through its `forward`, and compares the last-position logits with the fp32 oracle on the same weights:

    yardstick = || logits_reference_16bit - logits_oracle_fp32 || / || logits_oracle_fp32 ||

i.e. how far the reference itself sits from exact arithmetic when it runs in 16 bits.  tests/golden/yardstick.npz keeps the
reference's 16-bit logits and that number per case; the GPU test recomputes the fp32 oracle, checks the stored number against it and
requires the HIP path's own error to be <= 1.25 x the reference's.

Regenerating reproduces the error FIGURES (to ~1e-3 relative), not the bytes: the host's 16-bit matmul reduction order depends on its thread
count and ISA; the GPU test recomputes the figure from the stored logits against its own oracle run and accepts 5 %.

Usage: python oracle/gen_yardstick.py [7b_fp16 7b_bf16 13b_fp8_fp16 13b_fp8_bf16]   (13B: ~50 GB of host memory)
       python oracle/gen_yardstick.py vit                                            (HF CLIPVisionModel fp16 / bf16, 8 frames)
       python oracle/gen_yardstick.py tf [7b_bf16 13b_fp8_bf16]                      (the reference's own teacher-forced argmax agreement)
"""
from __future__ import annotations

import gc
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "yardstick.npz")

from oracle import fulldepth as fd      # noqa: E402
from oracle import gen_golden as gg     # noqa: E402
from oracle import llm as ollm          # noqa: E402
from oracle import synth                # noqa: E402

CASES = {"7b_fp16": ("7b", "float16", False), "7b_bf16": ("7b", "bfloat16", False),
         "13b_fp8_fp16": ("13b", "float16", True), "13b_fp8_bf16": ("13b", "bfloat16", True)}


def case_weights(name: str, dtype: str, fp8: bool) -> dict:
    """The weights the HIP path multiplies with: the seeded 16-bit checkpoint; for fp8 every decoder matrix + lm_head replaced by its
    e4m3-dequantised value (exactly representable in 16 bits; CPU twin of csrc/fp8.hip, bit-equal to the library's read-back)."""
    w = fd.make_weights(name, dtype)
    if fp8:
        tdt = torch.float16 if dtype == "float16" else torch.bfloat16
        for k in list(w):
            if k == "lm_head.weight" or any(f".{n}." in k for n in ollm.FP8_KEYS):
                w[k] = ollm.quantize_e4m3_rows(w[k].float()).to(tdt)
    return w


def reference_logits(cfg: synth.LlamaCfg, w: dict, ids, feats, tdt, first: int | None = None) -> torch.Tensor:
    """Logits of the reference's own forward in `tdt`: the last position [vocab], or (first given) positions first.. [S - first, vocab]."""
    from transformers import CLIPVisionConfig
    import transformers.modeling_utils as mu
    from video_chatgpt.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM
    with tempfile.TemporaryDirectory() as tmp:
        CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224, patch_size=14,
                         hidden_act="quick_gelu", layer_norm_eps=1e-5).save_pretrained(tmp)
        hc = VideoChatGPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                                num_attention_heads=cfg.heads, num_key_value_heads=cfg.heads, max_position_embeddings=4096, rms_norm_eps=cfg.eps,
                                mm_vision_tower=tmp, use_mm_proj=True, mm_hidden_size=cfg.mm_hidden, attn_implementation="eager")
        # no random init of 7e9 parameters (the reference's own disable_torch_init, video_chatgpt/utils.py), parameters created in 16 bits
        saved = (torch.nn.Linear.reset_parameters, torch.nn.Embedding.reset_parameters, mu.PreTrainedModel.init_weights)
        torch.nn.Linear.reset_parameters = lambda self: None
        torch.nn.Embedding.reset_parameters = lambda self: None
        mu.PreTrainedModel.init_weights = lambda self: None
        old = torch.get_default_dtype()
        torch.set_default_dtype(tdt)
        try:
            model = VideoChatGPTLlamaForCausalLM(hc).eval()
        finally:
            torch.set_default_dtype(old)
            torch.nn.Linear.reset_parameters, torch.nn.Embedding.reset_parameters, mu.PreTrainedModel.init_weights = saved
    missing, unexpected = model.load_state_dict(w, strict=False, assign=True)
    missing = [k for k in missing if "rotary" not in k and "inv_freq" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    # NOT model.to(tdt): that would also round the rotary inv_freq buffer to 16 bits (10x the logits error).  `from_pretrained(torch_dtype=
    # float16)` (video_chatgpt/eval/model_utils.py:104) builds the module under a 16-bit default dtype and leaves the explicitly-fp32
    # rotary buffers alone; parameters are already 16-bit here.
    assert all(p.dtype == tdt for p in model.parameters())
    rot = [b for n, b in model.named_buffers() if "inv_freq" in n]
    assert rot and all(b.dtype == torch.float32 for b in rot), [b.dtype for b in rot]
    vc = model.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1, True
    with torch.no_grad():
        o = model(input_ids=torch.tensor([ids]), video_spatio_temporal_features=feats.to(tdt)[None], use_cache=False)
    return o.logits[0, -1].float() if first is None else o.logits[0, first:].float()


def teacher_forced_yardstick(names):
    """`tf` mode: the reference's OWN teacher-forced argmax agreement in 16 bits.  The seeded continuation of
    tests/test_gpu_fulldepth.py (fd.teacher_tokens, fd.N_TEACHER positions) goes through the reference's forward in the case's dtype in one
    causal pass; stored per case: the reference's argmax at every visited position (`<case>_tf_ref_argmax`), the fp32 oracle's argmax and
    top-1/top-2 margin there, and the agreement count.  The GPU test recounts the agreement against its own oracle run and requires the HIP
    path's unfiltered count to be at least the reference's."""
    gg._import_reference()
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        case, dtype, fp8 = CASES[name]
        tdt = torch.float16 if dtype == "float16" else torch.bfloat16
        c = fd.CASES[case]
        cfg = c["cfg"]
        t0 = time.time()
        w = case_weights(case, dtype, fp8)
        ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
        cont = fd.teacher_tokens(cfg, c["prompt_seed"], fd.N_TEACHER[case])
        print(f"[{name}] weights {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        lg_truth, margins, arg_truth = fd.teacher_forced_reference(w, cfg, ids, feats, cont)
        print(f"[{name}] fp32 oracle, one causal pass over {len(ids) + len(cont)} tokens: {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        ref = reference_logits(cfg, w, list(ids) + list(cont), feats, tdt, first=len(ids) - 1)
        print(f"[{name}] reference forward in {dtype}: {time.time() - t0:.0f}s", flush=True)
        arg_ref = ref.argmax(-1)
        agree = int((arg_ref == torch.tensor(arg_truth)).sum())
        errs = ((ref.double() - lg_truth.double()).norm(dim=-1) / lg_truth.double().norm(dim=-1))
        print(f"[{name}] reference {dtype} teacher-forced: argmax agrees at {agree}/{len(arg_truth)} positions; normwise error "
              f"first {float(errs[0]):.3e} worst {float(errs.max()):.3e}", flush=True)
        out[f"{name}_tf_ref_argmax"] = arg_ref.numpy().astype(np.int32)
        out[f"{name}_tf_truth_argmax"] = np.asarray(arg_truth, dtype=np.int32)
        out[f"{name}_tf_truth_margin"] = np.asarray(margins, dtype=np.float32)
        out[f"{name}_tf_agree"] = np.int64(agree)
        out[f"{name}_tf_ref_err_worst"] = np.float64(float(errs.max()))
        del w, ref, lg_truth
        gc.collect()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: int(v) for k, v in out.items() if k.endswith("_tf_agree")})


def main():
    if sys.argv[1:] == ["vit"]:
        return vit_yardstick()
    if sys.argv[1:2] == ["tf"]:
        return teacher_forced_yardstick(sys.argv[2:] or ["7b_bf16", "13b_fp8_bf16"])
    names = sys.argv[1:] or ["7b_fp16", "7b_bf16"]
    gg._import_reference()
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        case, dtype, fp8 = CASES[name]
        tdt = torch.float16 if dtype == "float16" else torch.bfloat16
        c = fd.CASES[case]
        cfg = c["cfg"]
        t0 = time.time()
        w = case_weights(case, dtype, fp8)
        ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
        print(f"[{name}] weights {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        with torch.no_grad():
            truth = ollm.LlamaOracle(w, cfg).prefill(ids, feats, cfg.vocab - 2, cfg.vocab - 1, cfg.vocab - 3)[0]
        print(f"[{name}] fp32 oracle prefill {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        ref = reference_logits(cfg, w, ids, feats, tdt)
        print(f"[{name}] reference forward in {dtype} {time.time() - t0:.0f}s", flush=True)
        err = float((ref.double() - truth.double()).norm() / truth.double().norm())
        agree = int(ref.argmax()) == int(truth.argmax())
        print(f"[{name}] reference {dtype} vs fp32 oracle: normwise {err:.4e}; argmax agrees: {agree}", flush=True)
        out[f"{name}_ref_logits"] = ref.numpy().astype(np.float32)
        out[f"{name}_ref_err"] = np.float64(err)
        out[f"{name}_truth_sub"] = truth[::97].numpy().astype(np.float32)       # a thin slice of the oracle's logits: pins the case itself
        del w, ref, truth
        gc.collect()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: float(v) for k, v in out.items() if k.endswith("_err")})




def vit_yardstick():
    """The vision side of the yardstick: HF CLIPVisionModel (the module `initialize_model` builds, video_chatgpt/eval/model_utils.py:134) run in
    fp16 / bf16 on the host for BASELINE config 1 (ViT-L/14, 8 synthetic frames, the seeds of tests/test_gpu_vision.py), hidden_states[-2][:, 1:]
    against the fp32 oracle -> tests/golden/yardstick.npz keys vit_l14_8f_{fp16,bf16}_ref_err."""
    from oracle import vision as ovis
    gg._import_reference()
    cfg = synth.CLIP_L14_224
    w = synth.make_clip_weights(cfg, seed=0)
    px = ovis.clip_preprocess(synth.make_frames(8, 224, seed=0))
    truth = ovis.clip_select_features(px, w, cfg)
    out = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name, tdt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        model, _ = gg._hf_clip(cfg, w)
        model = model.to(tdt)
        t0 = time.time()
        with torch.no_grad():
            feat = model(px.to(tdt), output_hidden_states=True).hidden_states[-2][:, 1:].float()
        err = float((feat.double() - truth.double()).norm() / truth.double().norm())
        print(f"[vit_l14_8f_{name}] HF CLIPVisionModel in {name} vs fp32 oracle: normwise {err:.4e} ({time.time() - t0:.0f}s)", flush=True)
        out[f"vit_l14_8f_{name}_ref_err"] = np.float64(err)
    np.savez_compressed(OUT, **out)


if __name__ == "__main__":
    main()
