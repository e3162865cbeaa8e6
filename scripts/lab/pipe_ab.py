"""A/B check for GEMV kernel variants selected by an environment variable (PGV_GEMV_PIPE, ...): 7B-shaped 4-layer random model, 8 ragged prompts,
prefill + 6 decode steps with 16-bit and with fp8 weights; all logits go to argv[1] (.npy).  Run once per variant, compare the files with cmp."""
import sys

import numpy as np
import torch

from video_llava_amd import random_init as ri
from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig

DEV = "cuda:0"
vocab, V = 32003, 356
out = []
for dtype in (torch.bfloat16, torch.float16):
    cfg = VideoChatGPTConfig(vocab_size=vocab, hidden_size=4096, intermediate_size=11008, num_hidden_layers=4, num_attention_heads=32, eos_token_id=None)
    m = VideoChatGPTLlamaForCausalLM(cfg, VisionConfig(frame_size=224), dtype, torch.device(DEV))
    ri.load_streaming(m, ri.iter_llama_tensors(vocab=vocab, hidden=4096, inter=11008, layers=4, device=DEV, dtype=dtype, seed=5))
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = vocab - 3, vocab - 2, vocab - 1, True
    rng = np.random.default_rng(1)
    prompts = [[1] + rng.integers(3, vocab - 3, 60 + 3 * i).tolist() + [vocab - 2] + [vocab - 3] * V + [vocab - 1] + rng.integers(3, vocab - 3, 6).tolist()
               for i in range(8)]
    feats = torch.from_numpy(rng.standard_normal((8, V, 1024)).astype(np.float32) * 0.5).to(torch.float16).to(DEV)
    for weights in ("16", "fp8"):
        if weights == "fp8":
            m.quantize_weights_fp8()
        kv, nxt, lg = m.prefill(prompts, feats, 512, want_logits=True)
        out.append(lg.float().cpu().numpy())
        for _ in range(6):
            nxt, lg = m.decode_step(kv, nxt, want_logits=True)
            out.append(lg.float().cpu().numpy())
        del kv
    del m
np.save(sys.argv[1], np.stack(out))
print("saved", sys.argv[1], np.stack(out).shape, float(np.abs(np.stack(out)).max()))
