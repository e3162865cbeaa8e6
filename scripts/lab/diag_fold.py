"""Diagnostic: is the folded-LayerNorm tower deterministic and position-independent?  (run on the GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import synth
from video_llava_amd import _lib
from video_llava_amd.vision_tower import CLIPVisionTower, CLIPVisionTowerConfig
DEV = "cuda:0"
ctx = _lib.Context.get(0)
cfg = synth.CLIP_L14_224
w = synth.make_clip_weights(cfg, seed=0)
tower = CLIPVisionTower(CLIPVisionTowerConfig(), torch.float16)
tower.load_state_dict(w)
px = ctx.preprocess_u8(torch.from_numpy(synth.make_frames(100, 224, seed=0)).to(DEV), torch.float16)
for nl in (1, 2, 23):
    a = tower.hidden_state(px, nl); b = tower.hidden_state(px, nl)
    print("layers", nl, "run-to-run equal:", bool(torch.equal(a, b)))
    perm = torch.randperm(100, generator=torch.Generator().manual_seed(0)).to(DEV)
    c = tower.hidden_state(px[perm], nl)
    diff = (c != a[perm])
    print("  perm mismatches:", int(diff.sum()), "of", diff.numel(), "frames with mismatch:", int(diff.any(-1).any(-1).sum()),
          "rows (token idx) with mismatch:", torch.nonzero(diff.any(-1).any(0))[:12, 0].tolist(), "max abs", float((c.float() - a[perm].float()).abs().max()))
    # which positions (in the permuted batch) mismatch
    fr = torch.nonzero(diff.any(-1).any(-1))[:, 0].tolist()
    print("  permuted positions with mismatch:", fr[:20])
