"""Tiny synthetic checkpoint tree in the ON-DISK formats the reference's loader reads.  TEST INFRASTRUCTURE (oracle/__init__.py).

There are no released checkpoints offline, so the loader path -- reference video_chatgpt/eval/model_utils.py:82-150
(`initialize_model`), fed by the files video_chatgpt/train/llava_trainer.py:24-46 writes -- is exercised on a tree generated
from seeds:

    <root>/clip/      config.json (HF CLIP config with a vision_config block), preprocessor_config.json, model.safetensors
    <root>/llm/       config.json (LLaMA fields + mm_vision_tower -> <root>/clip, mm_hidden_size, use_mm_proj),
                      model-0000x-of-00002.safetensors + model.safetensors.index.json  (base vocabulary, NO video tokens),
                      tokenizer.json / tokenizer_config.json / special_tokens_map.json  (byte-level, <s>=1, </s>=2)
    <root>/mm_projector.bin   torch.save({k: v}) of the keys the trainer keeps ('mm_projector' | 'embed_tokens' | 'embed_in',
                      llava_trainer.py:33-36): the projector and the embedding table AFTER the three video tokens were added

Both the real reference (oracle/gen_golden.py, build container only) and this repo's `initialize_model` (tests, GPU box) load the
same tree; everything is bit-reproducible from the seeds (numpy PCG64, oracle/synth.py).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import synth

BASE_VOCAB = 512          # vocabulary of the "language-only" checkpoint; initialize_model grows it by the 3 video tokens


def build_tokenizer(vocab_size: int = BASE_VOCAB):
    """Byte-level tokenizer with LLaMA's special ids (<unk>=0, <s>=1, </s>=2) and a BOS-prepending post-processor: exposes exactly
    the calls the path makes (tokenizer([prompt]).input_ids, tokenizer(str).input_ids, batch_decode(skip_special_tokens=True),
    add_tokens(special_tokens=True), convert_tokens_to_ids, len)."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for ch in alphabet:
        vocab[ch] = len(vocab)
    i = 0
    while len(vocab) < vocab_size:                       # filler entries so the table has the checkpoint's row count
        vocab[f"<filler_{i}>"] = len(vocab)
        i += 1
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[], unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)
    tok.decoder = decoders.ByteLevel()
    tok.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    return PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", unk_token="<unk>")


def write_checkpoint_tree(root: str, lcfg: synth.LlamaCfg | None = None, ccfg: synth.ClipCfg | None = None, clip_seed: int = 61,
                          llm_seed: int = 62, head_std: float = 0.08, projector_dtype=torch.float16) -> dict:
    """Write the tree described in the module docstring.  `lcfg.vocab` is the FINAL vocabulary (base + 3 video tokens).
    Returns {"llm": dir, "clip": dir, "projector": file, "weights": full-vocabulary state dict (fp32 numpy, values rounded to fp16
    exactly as stored), "clip_weights": ...} for the oracle side of a test."""
    from safetensors.torch import save_file
    lcfg = lcfg or synth.LLAMA_TINY
    ccfg = ccfg or synth.CLIP_TINY
    assert lcfg.vocab == BASE_VOCAB + 3
    clip_dir, llm_dir = os.path.join(root, "clip"), os.path.join(root, "llm")
    os.makedirs(clip_dir, exist_ok=True)
    os.makedirs(llm_dir, exist_ok=True)

    # ---- CLIP directory (openai/clip-vit-large-patch14 layout: the vision tower's fields sit under "vision_config") ----------
    vis = dict(hidden_size=ccfg.hidden, intermediate_size=ccfg.inter, num_hidden_layers=ccfg.layers, num_attention_heads=ccfg.heads,
               image_size=ccfg.image, patch_size=ccfg.patch, hidden_act="quick_gelu", layer_norm_eps=ccfg.eps, model_type="clip_vision_model")
    with open(os.path.join(clip_dir, "config.json"), "w") as f:
        json.dump({"model_type": "clip", "projection_dim": 64, "vision_config": vis,
                   "text_config": {"model_type": "clip_text_model", "hidden_size": 32, "intermediate_size": 64, "num_hidden_layers": 1,
                                   "num_attention_heads": 2, "vocab_size": 64, "max_position_embeddings": 16}}, f, indent=1)
    with open(os.path.join(clip_dir, "preprocessor_config.json"), "w") as f:
        json.dump({"image_processor_type": "CLIPImageProcessor", "do_resize": True, "size": {"shortest_edge": ccfg.image}, "resample": 3,
                   "do_center_crop": True, "crop_size": {"height": ccfg.image, "width": ccfg.image}, "do_rescale": True,
                   "rescale_factor": 1 / 255, "do_normalize": True, "image_mean": [0.48145466, 0.4578275, 0.40821073],
                   "image_std": [0.26862954, 0.26130258, 0.27577711], "do_convert_rgb": True}, f, indent=1)
    cw = synth.quantize_weights(synth.make_clip_weights(ccfg, seed=clip_seed), "float16")
    save_file({k: torch.from_numpy(v).half().contiguous() for k, v in cw.items()}, os.path.join(clip_dir, "model.safetensors"),
              metadata={"format": "pt"})

    # ---- language checkpoint: base vocabulary only, two shards + index ------------------------------------------------------
    w = synth.quantize_weights(synth.make_llama_weights(lcfg, seed=llm_seed, head_std=head_std), "float16")   # full (final) vocabulary
    base = {k: v for k, v in w.items() if not k.startswith("model.mm_projector.")}
    for k in ("model.embed_tokens.weight", "lm_head.weight"):
        base[k] = w[k][:BASE_VOCAB]
    keys = sorted(base)
    half = len(keys) // 2
    shards = {"model-00001-of-00002.safetensors": keys[:half], "model-00002-of-00002.safetensors": keys[half:]}
    weight_map = {}
    for fname, ks in shards.items():
        save_file({k: torch.from_numpy(base[k]).half().contiguous() for k in ks}, os.path.join(llm_dir, fname), metadata={"format": "pt"})
        weight_map.update({k: fname for k in ks})
    with open(os.path.join(llm_dir, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {"total_size": int(sum(base[k].size * 2 for k in keys))}, "weight_map": weight_map}, f, indent=1)
    with open(os.path.join(llm_dir, "config.json"), "w") as f:
        json.dump({"model_type": "VideoChatGPT", "architectures": ["VideoChatGPTLlamaForCausalLM"], "vocab_size": BASE_VOCAB,
                   "hidden_size": lcfg.hidden, "intermediate_size": lcfg.inter, "num_hidden_layers": lcfg.layers,
                   "num_attention_heads": lcfg.heads, "num_key_value_heads": lcfg.heads, "max_position_embeddings": 2048,
                   "rms_norm_eps": lcfg.eps, "rope_theta": lcfg.rope_theta, "hidden_act": "silu", "bos_token_id": 1, "eos_token_id": 2,
                   "pad_token_id": 0, "tie_word_embeddings": False, "torch_dtype": "float16", "use_cache": True,
                   "mm_vision_tower": clip_dir, "use_mm_proj": True, "mm_hidden_size": lcfg.mm_hidden,
                   "mm_projector_type": lcfg.projector, "attn_implementation": "eager"}, f, indent=1)
    build_tokenizer(BASE_VOCAB).save_pretrained(llm_dir)

    # ---- mm_projector.bin: the trainer's key filter (llava_trainer.py:33-36) over the trained model's state dict ----------------
    keys_to_match = ["mm_projector", "embed_tokens", "embed_in"]
    proj = {k: torch.from_numpy(v).to(projector_dtype) for k, v in w.items() if any(m in k for m in keys_to_match)}
    proj_path = os.path.join(root, "mm_projector.bin")
    torch.save(proj, proj_path)
    # what the loaded model must compute with: checkpoint rows + the projector file's embedding table; lm_head rows of the three
    # added tokens are whatever resize_token_embeddings initialises them to -- the reference (transformers >= 4.3x) uses the MEAN of
    # the old rows, this repo zero-fills.  Neither file carries them, and a video token is never a sensible prediction; the
    # effective table for parity is therefore reported per loader by the caller.
    return {"llm": llm_dir, "clip": clip_dir, "projector": proj_path, "weights": w, "clip_weights": cw}
