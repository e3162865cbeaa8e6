"""Which rows differ between a one-call forward and the same prompt cut into two calls (scripts/sessions/r6_b.sh)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import synth
from helpers import make_model
g = np.load(os.path.join(ROOT, "tests", "golden", "llama_tiny.npz"))
cfg = synth.LLAMA_TINY
w = synth.make_llama_weights(cfg, seed=int(g["lin_weight_seed"]), head_std=float(g["head_std"]))
m = make_model(cfg, w, torch.float16, 224)
ids = g["lin_ids"].tolist(); S = len(ids)
feats = torch.from_numpy(g["lin_feats"]).half()[None]
full = m(input_ids=torch.tensor([ids]), video_spatio_temporal_features=feats).logits[0].clone()
full2 = m(input_ids=torch.tensor([ids]), video_spatio_temporal_features=feats).logits[0].clone()
print("S", S, "full == full again:", torch.equal(full, full2))
start = ids.index(cfg.vocab - 2); end = ids.index(cfg.vocab - 1)
print("start", start, "end", end)
for cut in (1, 2, 16, 32, 33, start, 64, 65, 100, 128, 129, 200, 256, end + 1, end + 3, S - 1):
    o1 = m(input_ids=torch.tensor([ids[:cut]]), video_spatio_temporal_features=feats, max_length=S + 16)
    try:
        o2 = m(input_ids=torch.tensor([ids[cut:]]), video_spatio_temporal_features=feats, past_key_values=o1.past_key_values)
    except Exception as e:
        print("cut", cut, "second call:", type(e).__name__, str(e)[:80]); continue
    got = torch.cat([o1.logits[0], o2.logits[0]])
    bad = (got != full).any(dim=1).nonzero().flatten().tolist()
    print("cut", cut, "rows differing:", len(bad), "first", bad[:6], "last", bad[-3:], "max abs", float((got - full).abs().max()))
# text-only model path: no video at all
tids = [1] + list(range(5, 5 + 300))
fullt = m(input_ids=torch.tensor([tids])).logits[0].clone()
for cut in (1, 31, 32, 64, 100, 128, 200, 299):
    o1 = m(input_ids=torch.tensor([tids[:cut]]), max_length=400)
    o2 = m(input_ids=torch.tensor([tids[cut:]]), past_key_values=o1.past_key_values)
    got = torch.cat([o1.logits[0], o2.logits[0]])
    bad = (got != fullt).any(dim=1).nonzero().flatten().tolist()
    print("text cut", cut, "rows differing:", len(bad), "first", bad[:6], "last", bad[-3:])
