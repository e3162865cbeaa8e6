"""Shared builders of the GPU tests (imported as `from helpers import ...`; pytest puts tests/ on sys.path)."""
import torch

from oracle import synth


def make_model(cfg: synth.LlamaCfg, w: dict, dtype, image=224):
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    hc = VideoChatGPTConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                            num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, rms_norm_eps=cfg.eps,
                            rope_theta=cfg.rope_theta, mm_hidden_size=cfg.mm_hidden, mm_projector_type=cfg.projector,
                            eos_token_id=None)
    m = VideoChatGPTLlamaForCausalLM(hc, VisionConfig(frame_size=image), dtype)
    st = m.load_state_dict(w)
    assert not st.unexpected_keys and not st.missing_keys
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1, True
    return m


def make_tower(ccfg, cw, dtype=torch.float16):
    from video_llava_amd.vision_tower import CLIPVisionTower, CLIPVisionTowerConfig
    t = CLIPVisionTower(CLIPVisionTowerConfig(hidden_size=ccfg.hidden, intermediate_size=ccfg.inter, num_hidden_layers=ccfg.layers,
                                              num_attention_heads=ccfg.heads, image_size=ccfg.image, patch_size=ccfg.patch), dtype)
    t.load_state_dict(cw)
    return t


class SynthTokenizer:
    """Synthetic tokenizer exposing exactly the calls the path makes: ids are byte values + 3, the three video tokens
    sit at the top of the vocabulary."""

    def __init__(self, vocab):
        self.vocab = vocab
        self.special = {"<vid_patch>": vocab - 3, "<vid_start>": vocab - 2, "<vid_end>": vocab - 1}

    def _encode(self, s):
        ids, i = [1], 0
        while i < len(s):
            for name, tid in self.special.items():
                if s.startswith(name, i):
                    ids.append(tid); i += len(name)
                    break
            else:
                ids.append(3 + (ord(s[i]) % (self.vocab - 8))); i += 1
        return ids

    def __call__(self, x):
        class R: pass
        r = R()
        r.input_ids = [self._encode(t) for t in x] if isinstance(x, (list, tuple)) else self._encode(x)
        return r

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(t)) for t in row) for row in ids]


