"""`VideoChatGPTLlamaForCausalLM` drop-in: the object `initialize_model` returns in slot 1 and
`video_chatgpt_infer` drives through `model.get_model().vision_config`, `model.get_model().mm_projector` and
`model.generate(input_ids, video_spatio_temporal_features=..., do_sample, temperature, max_new_tokens,
stopping_criteria)` (reference: video_chatgpt/model/video_chatgpt.py:16-325; call sites
video_chatgpt/inference.py:67,105-112, chat.py:148-154, eval/model_utils.py:104-144).

The decoder stack, the video splice, lm_head and greedy picking run in libpgv (hand-written gfx950 kernels);
this file is host orchestration only: placeholder validation (same ValueErrors as the reference), the
token loop, stopping criteria and sampling glue.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..constants import CLIP_WIDTH
from .multimodal_projector.builder import HipLinear, IdentityMap, build_vision_projector

DEFAULT_VIDEO_TOKEN = "<video>"
DEFAULT_VIDEO_PATCH_TOKEN = "<vid_patch>"
DEFAULT_VID_START_TOKEN = "<vid_start>"
DEFAULT_VID_END_TOKEN = "<vid_end>"


class VisionConfig:
    """Same fields as the reference's VisionConfig (model/video_chatgpt.py:16-30)."""

    def __init__(self, frame_size=224, patch_size=14, hidden_size=1024):
        self.frame_size = frame_size
        self.patch_size = patch_size
        self.hidden_size = hidden_size
        self.use_vid_start_end = None
        self.vid_start_token = None
        self.vid_end_token = None
        self.vid_patch_token = None


class VideoChatGPTConfig:
    """The LlamaConfig fields the path reads + the multimodal keys of config.json
    (`mm_vision_tower`, `use_mm_proj`, `mm_hidden_size`, `mm_projector_type`; model/video_chatgpt.py:43-55)."""
    model_type = "VideoChatGPT"

    def __init__(self, vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                 num_attention_heads=32, num_key_value_heads=None, max_position_embeddings=4096, rms_norm_eps=1e-5,
                 rope_theta=10000.0, mm_vision_tower=None, use_mm_proj=True, mm_hidden_size=1024,
                 mm_projector_type="linear", eos_token_id=2, bos_token_id=1, pad_token_id=None, **extra):
        if num_key_value_heads not in (None, num_attention_heads):
            raise ValueError("grouped-query checkpoints are outside the PG-Video-LLaVA path (7B/13B are MHA)")
        self.vocab_size, self.hidden_size, self.intermediate_size = vocab_size, hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.max_position_embeddings, self.rms_norm_eps, self.rope_theta = max_position_embeddings, rms_norm_eps, rope_theta
        self.mm_vision_tower, self.use_mm_proj, self.mm_hidden_size = mm_vision_tower, use_mm_proj, mm_hidden_size
        self.mm_projector_type = mm_projector_type
        self.eos_token_id, self.bos_token_id, self.pad_token_id = eos_token_id, bos_token_id, pad_token_id
        self.extra = extra

    @classmethod
    def from_pretrained(cls, path: str) -> "VideoChatGPTConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls(**json.load(f))


class VideoChatGPTLlamaModel(nn.Module):
    """`model.get_model()`: carries vision_config and mm_projector like the reference's inner LlamaModel subclass."""

    def __init__(self, config: VideoChatGPTConfig, vision_config: VisionConfig, dtype, device):
        super().__init__()
        self.config = config
        self.vision_config = vision_config
        if config.use_mm_proj:
            if vision_config.frame_size == 224:      # LLaVA-Lightning style bare Linear (model/video_chatgpt.py:52-53)
                self.mm_projector = HipLinear(config.mm_hidden_size, config.hidden_size, dtype, device)
            else:                                     # LLaVA-1.5: by config.mm_projector_type (:54-55)
                self.mm_projector = build_vision_projector(config, dtype=dtype, device=device)


class PastKeyValues:
    """What forward() hands back as `past_key_values`: the library's KV-cache handle (the cache lives in HBM inside libpgv; HF's tuple of
    tensors / DynamicCache has no counterpart).  Truthy once it holds tokens, like the tuple the reference tests at model/video_chatgpt.py:256."""

    def __init__(self, owner, kv, batch: int, seq_len: int, max_seq: int, epoch: int):
        self.owner, self.kv, self.batch, self.seq_len = owner, kv, batch, seq_len
        self.max_seq = max_seq          # capacity of the cache behind `kv`
        self.epoch = epoch              # the model's cache epoch when this object was handed out: any later prefill / generate makes it stale

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.seq_len

    def __bool__(self):
        return self.seq_len > 0

    def __len__(self):
        return self.owner.config.num_hidden_layers


class CausalLMOutputWithPast:
    """`.logits` / `.past_key_values`, also indexable like HF's ModelOutput ([0] = logits, [1] = past_key_values)."""

    def __init__(self, logits, past_key_values):
        self.logits, self.past_key_values = logits, past_key_values
        self.loss = self.hidden_states = self.attentions = None

    def __getitem__(self, i):
        return (self.logits, self.past_key_values)[i]


class VideoChatGPTLlamaForCausalLM(nn.Module):
    def __init__(self, config: VideoChatGPTConfig, vision_config: Optional[VisionConfig] = None,
                 torch_dtype: torch.dtype = torch.float16, device=None):
        super().__init__()
        self.config = config
        self.dtype_ = torch_dtype
        self.ctx = _lib.Context.get(device)
        self.device_ = torch.device("cuda", self.ctx.device)
        if vision_config is None:
            vision_config = _vision_config_from(config.mm_vision_tower)
        if vision_config.hidden_size != CLIP_WIDTH:
            raise ValueError(f"CLIP width must be {CLIP_WIDTH} (the reference hard-codes it at model/video_chatgpt.py:106)")
        self.model = VideoChatGPTLlamaModel(config, vision_config, torch_dtype, self.device_)
        lc = _lib.LlmConfig(config.vocab_size, config.hidden_size, config.intermediate_size, config.num_hidden_layers,
                            config.num_attention_heads, config.rms_norm_eps, config.rope_theta)
        h = C.c_void_p()
        _lib.check(self.ctx.lib.pgv_llm_create(self.ctx.handle, C.byref(lc), _lib.dtype_code(torch_dtype), C.byref(h)), "pgv_llm_create")
        self.handle = h
        self._kv: dict = {}
        self._kv_epoch = 0              # bumped by every call that (re)fills a cache from position 0: what PastKeyValues objects are checked against
        self._reuse = None              # generate(kv_reuse_key=...): (key, kv handle, max_seq, ids whose keys / values sit in the cache)

    def __del__(self):
        try:
            for kv in getattr(self, "_kv", {}).values():
                self.ctx.lib.pgv_kv_destroy(kv)
            if getattr(self, "handle", None):
                self.ctx.lib.pgv_llm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- HF-module surface the reference touches ----------------------------------------------------
    def get_model(self) -> VideoChatGPTLlamaModel:
        return self.model

    def eval(self):
        return self

    def cuda(self, *a, **k):
        return self

    def half(self):
        return self

    @property
    def device(self):
        return self.device_

    @property
    def vocab_size(self) -> int:
        return self.ctx.lib.pgv_llm_vocab(self.handle)

    # ---- fp8 weight path (BASELINE config 5) ------------------------------------------------------------
    def quantize_weights_fp8(self):
        """Quantise every decoder matrix and lm_head to e4m3 with per-row power-of-two scales (pgv_llm_quantize_fp8): decode then
        streams half the weight bytes per token; prefill and decode both compute with the dequantised values, which
        `get_weight(key)` returns for a checkpoint writer or a parity oracle.  Call after load_state_dict."""
        _lib.check(self.ctx.lib.pgv_llm_quantize_fp8(self.ctx.handle, self.handle, _lib.stream_ptr(self.device_)), "pgv_llm_quantize_fp8")
        return self

    @property
    def is_fp8(self) -> bool:
        return bool(self.ctx.lib.pgv_llm_is_fp8(self.handle))

    def get_weight(self, key: str) -> torch.Tensor:
        """fp32 [out, in] copy of a decoder matrix / lm_head under its HF key, as the kernels see it (dequantised after fp8)."""
        c = self.config
        H, I = c.hidden_size, c.intermediate_size
        if key == "lm_head.weight":
            shape = (self.vocab_size, H)
        elif key.endswith(("gate_proj.weight", "up_proj.weight")):
            shape = (I, H)
        elif key.endswith("down_proj.weight"):
            shape = (H, I)
        else:
            shape = (H, H)
        out = torch.empty(shape, dtype=torch.float32, device=self.device_)
        _lib.check(self.ctx.lib.pgv_llm_get_weight(self.ctx.handle, self.handle, key.encode(), out.data_ptr(), _lib.stream_ptr(self.device_)),
                   "pgv_llm_get_weight")
        return out

    def resize_token_embeddings(self, new_num_tokens: int):
        """eval/model_utils.py:119: grow embed_tokens / lm_head for the three video tokens (new rows are zero until a checkpoint fills
        them; HF initialises them from the old rows' statistics -- neither is ever a trained value)."""
        _lib.check(self.ctx.lib.pgv_llm_resize_vocab(self.handle, int(new_num_tokens), _lib.stream_ptr(self.device_)),
                   "resize_token_embeddings")
        self.config.vocab_size = int(new_num_tokens)
        for kv in self._kv.values():                    # logits buffers / captured decode graphs carry the old vocabulary size
            self.ctx.lib.pgv_kv_destroy(kv)
        self._kv.clear()
        self._kv_epoch += 1
        self._reuse = None

    # ---- weights ----------------------------------------------------------------------------------------
    def _expected_shape(self, key: str):
        """Shape HF's LlamaForCausalLM holds under `key` for this config and the CURRENT vocabulary (None: not a decoder key)."""
        c = self.config
        H, I, V = c.hidden_size, c.intermediate_size, self.vocab_size
        if key in ("model.embed_tokens.weight", "lm_head.weight"):
            return (V, H)
        if key == "model.norm.weight":
            return (H,)
        if not key.startswith("model.layers."):
            return None
        rest = key.split(".", 3)[-1]
        return {"self_attn.q_proj.weight": (H, H), "self_attn.k_proj.weight": (H, H), "self_attn.v_proj.weight": (H, H),
                "self_attn.o_proj.weight": (H, H), "mlp.gate_proj.weight": (I, H), "mlp.up_proj.weight": (I, H),
                "mlp.down_proj.weight": (H, I), "input_layernorm.weight": (H,), "post_attention_layernorm.weight": (H,)}.get(rest)

    def load_state_dict(self, sd: dict, strict: bool = True):
        """Routes `model.mm_projector.*` to the projector module and every other key to the packed device weights.
        Returns an object with `.missing_keys` / `.unexpected_keys` like torch (eval/model_utils.py:124-126); a tensor whose shape
        disagrees with the config (or, for the vocabulary matrices, with the current vocabulary size) raises torch's "size mismatch"
        RuntimeError -- also under strict=False, as torch does."""
        lib = self.ctx.lib
        unexpected: List[str] = []
        proj_sd = {}
        for k, v in sd.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            if k.startswith("model.mm_projector."):
                proj_sd[k[len("model.mm_projector."):]] = v
                continue
            if "rotary_emb.inv_freq" in k:
                continue
            v = v.detach()
            if v.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                v = v.float()
            v = v.contiguous()
            exp = self._expected_shape(k)
            if exp is not None and tuple(v.shape) != exp:
                raise RuntimeError(f"Error(s) in loading state_dict for VideoChatGPTLlamaForCausalLM:\n\tsize mismatch for {k}: copying a param "
                                   f"with shape {tuple(v.shape)} from checkpoint, the shape in current model is {exp}.")
            if v.is_cuda:
                torch.cuda.current_stream(v.device).synchronize()
            dev = 1 if v.is_cuda else 0
            if k in ("model.embed_tokens.weight", "lm_head.weight"):
                rc = lib.pgv_llm_load_rows(self.handle, k.encode(), v.data_ptr(), _lib.dtype_code(v.dtype), dev, 0, v.shape[0], v.numel(),
                                           _lib.stream_ptr(self.device_))
            else:
                rc = lib.pgv_llm_load_tensor(self.handle, k.encode(), v.data_ptr(), _lib.dtype_code(v.dtype), dev, v.numel(),
                                             _lib.stream_ptr(self.device_))
            if rc == _lib.PGV_ENAME:
                unexpected.append(k)
                continue
            _lib.check(rc, f"load {k}")
        if proj_sd:
            mp = self.model.mm_projector
            res = mp.load_state_dict({k: v.to(self.dtype_) for k, v in proj_sd.items()}, strict=False)
            unexpected += ["model.mm_projector." + k for k in res.unexpected_keys]
        torch.cuda.synchronize(self.device_)
        missing = lib.pgv_llm_missing(self.handle)
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: {missing} decoder tensors missing, unexpected keys {unexpected}")

        class _Status:
            pass
        st = _Status()
        st.missing_keys = [f"<{missing} decoder tensors>"] if missing else []
        st.unexpected_keys = unexpected
        return st

    @classmethod
    def from_pretrained(cls, path: str, low_cpu_mem_usage: bool = True, torch_dtype: torch.dtype = torch.float16,
                        use_cache: bool = True, device=None, vision_config: Optional[VisionConfig] = None):
        """Load a local HF checkpoint directory (config.json + *.safetensors or pytorch_model*.bin shards)."""
        cfg = VideoChatGPTConfig.from_pretrained(path)
        model = cls(cfg, vision_config, torch_dtype, device)
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if files:
            from safetensors.torch import load_file
            for f in files:
                model.load_state_dict(load_file(os.path.join(path, f)), strict=False)
        else:
            for f in sorted(f for f in os.listdir(path) if f.startswith("pytorch_model") and f.endswith(".bin")):
                model.load_state_dict(torch.load(os.path.join(path, f), map_location="cpu"), strict=False)
        if model.ctx.lib.pgv_llm_missing(model.handle):
            raise RuntimeError(f"{path}: decoder checkpoint is incomplete")
        return model

    # ---- prompt validation (host): same checks and messages as model/video_chatgpt.py:113-128,150-157 ----
    def _video_positions(self, seqs: Sequence[Sequence[int]], num_video_rows: int) -> List[int]:
        vc = self.model.vision_config
        out = []
        for ids in seqs:
            a = np.asarray(ids)
            if int((a == vc.vid_patch_token).sum()) == 0:
                out.append(-1)                          # text-only sample (:113-118)
                continue
            if vc.use_vid_start_end:
                starts = np.nonzero(a == vc.vid_start_token)[0]
                if len(starts) != int((a == vc.vid_end_token).sum()):
                    raise ValueError("The number of video start tokens and video end tokens should be the same.")
                pos = -1
                for p in starts.tolist():
                    if p + num_video_rows + 1 >= len(a) or a[p + num_video_rows + 1] != vc.vid_end_token:
                        raise ValueError("The video end token should follow the video start token.")
                    pos = p                              # with several runs the reference keeps the last splice (:123-146)
                out.append(pos)
            else:
                idx = np.nonzero(a == vc.vid_patch_token)[0]
                if len(idx) != num_video_rows:
                    raise ValueError("The number of video patch tokens should be the same as the number of video patches.")
                if (idx != np.arange(idx[0], idx[0] + num_video_rows)).any():
                    raise ValueError("The video patch tokens should be consecutive.")
                out.append(int(idx[0]) - 1)              # rows (pos, pos+V] are replaced
        return out

    # ---- KV cache handles --------------------------------------------------------------------------------
    def _get_kv(self, batch: int, max_seq: int):
        key = (batch, max_seq)
        if key not in self._kv:
            for old in self._kv.values():               # keep one cache alive (they are GBs)
                self.ctx.lib.pgv_kv_destroy(old)
            self._kv.clear()
            self._reuse = None
            h = C.c_void_p()
            _lib.check(self.ctx.lib.pgv_kv_create(self.ctx.handle, self.handle, batch, max_seq, C.byref(h)), "pgv_kv_create")
            self._kv[key] = h
        return self._kv[key]

    def _reusable_prefix(self, key, ids: Sequence[int], max_new_tokens: int) -> int:
        """Positions of the cache kept by the last generate(kv_reuse_key=key) that the prompt `ids` can start from (0: none -- full prefill)."""
        r = self._reuse
        if r is None or r[0] != key or self._kv.get((1, r[2])) is not r[1] or len(ids) + max_new_tokens > r[2]:
            return 0
        cached = r[3]
        n = min(len(cached), len(ids) - 1)                 # at least one new token has to run: its logits pick the first answer token
        a, b = np.asarray(cached[:n]), np.asarray(ids[:n])
        diff = np.flatnonzero(a != b)
        L = int(diff[0]) if diff.size else n
        vc = self.model.vision_config
        video_ids = [t for t in (vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token) if t is not None]
        if L < 1 or np.isin(np.asarray(ids[L:]), video_ids).any():
            return 0                                       # the placeholder run is (partly) in the new tokens: it needs the splice of a full prefill
        return L

    # ---- low-level steps (used by generate and by the parity tests) -------------------------------------
    def prefill(self, seqs: Sequence[Sequence[int]], video_spatio_temporal_features: Optional[torch.Tensor], max_seq: int,
                want_logits: bool = False, want_all_logits: bool = False, append_to=None):
        """Run the prompt(s); returns (kv handle, next-token ids [B] int32 on device, logits [B, vocab] or None) -- with want_all_logits a fourth
        entry: the logits of EVERY position, [sum of lengths, vocab] fp32 (what the reference's forward returns, model/video_chatgpt.py:225-226).
        `append_to` = a kv handle this model filled before: the rows continue its sequences (pgv_llm_prefill_append; `max_seq` is ignored)."""
        B = len(seqs)
        lens = [len(s) for s in seqs]
        feats = video_spatio_temporal_features
        proj = None
        V = 0
        vid_pos = [-1] * B
        if feats is not None and not (max(lens) == 1):     # the reference skips the splice when input_ids.shape[1]==1 (:103)
            if feats.dim() == 2:
                feats = feats.unsqueeze(0)
            proj = self.model.mm_projector(feats.to(self.device_))         # [B, V, hidden]  (:105)
            if isinstance(self.model.mm_projector, IdentityMap):
                proj = proj.to(self.dtype_)
            proj = proj.contiguous()
            V = proj.shape[1]
            if proj.shape[0] != B:
                raise ValueError(f"{proj.shape[0]} video feature sets for {B} prompts")
            vid_pos = self._video_positions(seqs, V)
        if append_to is None:
            kv = self._get_kv(max(B, 1), max_seq)
            self._kv_epoch += 1
            self._reuse = None
        else:
            kv = append_to
        flat = np.concatenate([np.asarray(s, dtype=np.int32) for s in seqs])
        h_lens = (C.c_int32 * B)(*lens)
        h_pos = (C.c_int32 * B)(*vid_pos)
        nxt = torch.empty(B, dtype=torch.int32, device=self.device_)
        logits = torch.empty(B, self.vocab_size, dtype=torch.float32, device=self.device_) if want_logits else None
        ld_all = (self.vocab_size + 15) // 16 * 16
        all_logits = torch.empty(int(flat.size), ld_all, dtype=torch.float32, device=self.device_) if want_all_logits else None
        fn, who = (self.ctx.lib.pgv_llm_prefill, "pgv_llm_prefill") if append_to is None else (self.ctx.lib.pgv_llm_prefill_append, "pgv_llm_prefill_append")
        _lib.check(fn(self.ctx.handle, self.handle, kv, flat.ctypes.data_as(C.c_void_p), h_lens, B,
                      proj.data_ptr() if proj is not None else None, V, h_pos,
                      logits.data_ptr() if logits is not None else None, nxt.data_ptr(),
                      all_logits.data_ptr() if all_logits is not None else None, ld_all,
                      _lib.stream_ptr(self.device_)), who)
        if want_all_logits:
            return kv, nxt, logits, all_logits[:, :self.vocab_size]
        return kv, nxt, logits

    def decode_step(self, kv, last: torch.Tensor, want_logits: bool = False):
        B = last.shape[0]
        nxt = torch.empty(B, dtype=torch.int32, device=self.device_)
        logits = torch.empty(B, self.vocab_size, dtype=torch.float32, device=self.device_) if want_logits else None
        _lib.check(self.ctx.lib.pgv_llm_decode(self.ctx.handle, self.handle, kv, last.data_ptr(),
                                               logits.data_ptr() if logits is not None else None, nxt.data_ptr(),
                                               _lib.stream_ptr(self.device_)), "pgv_llm_decode")
        return nxt, logits

    def decode_greedy(self, kv, first: torch.Tensor, n: int, eos_id: int = -1) -> torch.Tensor:
        B = first.shape[0]
        toks = torch.empty(B, n, dtype=torch.int32, device=self.device_)
        _lib.check(self.ctx.lib.pgv_llm_decode_greedy(self.ctx.handle, self.handle, kv, first.data_ptr(), n, eos_id, toks.data_ptr(),
                                                      _lib.stream_ptr(self.device_)), "pgv_llm_decode_greedy")
        return toks

    def sample_last(self, kv, u: torch.Tensor, temperature: float, top_k: int = 50) -> torch.Tensor:
        """Draw the next token of every sequence from the logits of the last prefill / decode call (pgv_llm_sample): u [B] uniforms."""
        u = u.to(device=self.device_, dtype=torch.float32).contiguous()
        nxt = torch.empty(u.numel(), dtype=torch.int32, device=self.device_)
        _lib.check(self.ctx.lib.pgv_llm_sample(self.ctx.handle, self.handle, kv, float(temperature), int(top_k), u.data_ptr(), nxt.data_ptr(),
                                               _lib.stream_ptr(self.device_)), "pgv_llm_sample")
        return nxt

    def decode_sample(self, kv, first: torch.Tensor, n: int, eos_id: int, temperature: float, top_k: int, u: torch.Tensor) -> torch.Tensor:
        """n sampled steps on the device (pgv_llm_decode_sample): u [n, B] uniforms, step i draws with u[i]."""
        B = first.shape[0]
        u = u.to(device=self.device_, dtype=torch.float32).contiguous()
        assert u.shape == (n, B)
        toks = torch.empty(B, n, dtype=torch.int32, device=self.device_)
        _lib.check(self.ctx.lib.pgv_llm_decode_sample(self.ctx.handle, self.handle, kv, first.data_ptr(), n, eos_id, float(temperature), int(top_k),
                                                      u.data_ptr(), toks.data_ptr(), _lib.stream_ptr(self.device_)), "pgv_llm_decode_sample")
        return toks

    # ---- forward: the reference's own public entry (model/video_chatgpt.py:193-251) --------------------------------
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, video_spatio_temporal_features: Optional[torch.Tensor] = None,
                return_dict=None, max_length: Optional[int] = None, **_unused):
        """`model(input_ids=ids, video_spatio_temporal_features=f, use_cache=True)` -> object with `.logits` [B, S, vocab] and `.past_key_values`;
        `model(input_ids=tok [B, 1], past_key_values=out.past_key_values, ...)` -> the next step (`input_ids.shape[1] == 1` skips the splice,
        :103); `input_ids [B, S > 1]` next to a cache appends S tokens per sequence (a later chat turn) and returns their S logit rows.
        `max_length` (first call): positions the cache is allocated for (default: prompt + 256).
        This is the recipe SURVEY.md 8c drives the reference with (prefill + cached single-token steps); `generate()` is the fast path
        and what the reference's own callers use.  Inference only: `labels`, `inputs_embeds`, attention / hidden-state outputs and a padded
        `attention_mask` have no counterpart in the eval path and raise; logits are fp32 (HF up-casts them for the loss the same way)."""
        if labels is not None or inputs_embeds is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("forward(): labels / inputs_embeds / output_attentions / output_hidden_states belong to the training path, "
                                      "which is outside this package (SURVEY.md 2 OUT-OF-SCOPE)")
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        ids = torch.as_tensor(input_ids)
        if ids.dim() == 1:
            ids = ids[None]
        B, S = ids.shape
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).to(torch.bool).all()):
            raise NotImplementedError("forward(): padded batches are not supported; pass equal-length prompts or use generate() with a list of prompts")
        limit = min(int(self.config.max_position_embeddings), 4096)
        if past_key_values is None:
            # cache capacity: the prompt + `max_length` headroom (default 256 further positions, rounded up to 64) -- NOT the whole context window
            # (2 GiB per 7B sequence at 4096 positions, and a second cache shape evicts the one generate() was using)
            max_seq = min(limit, (int(max_length) if max_length is not None else S + 256) + 63 & ~63)
            if S > max_seq:
                raise ValueError(f"prompt of {S} tokens exceeds {'max_length' if max_length is not None and S <= limit else 'max_position_embeddings'} {max_seq}")
            kv, _nxt, _lg, all_lg = self.prefill([row.tolist() for row in ids.cpu()], video_spatio_temporal_features, max_seq, want_all_logits=True)
            return CausalLMOutputWithPast(all_lg.reshape(B, S, -1), PastKeyValues(self, kv, B, S, max_seq, self._kv_epoch))
        if not isinstance(past_key_values, PastKeyValues) or past_key_values.owner is not self:
            raise ValueError("past_key_values must be the object a previous forward() of this model returned")
        if past_key_values.batch != B:
            raise ValueError(f"{B} sequences for a cache of {past_key_values.batch}")
        # one KV cache is kept alive per model and a later prefill of the SAME shape refills it in place: the epoch, not the handle, tells
        # whether this object still describes what the cache holds
        if past_key_values.epoch != self._kv_epoch or self._kv.get((past_key_values.batch, past_key_values.max_seq)) is not past_key_values.kv:
            raise RuntimeError("past_key_values is stale: the model has run another prefill / generate since (one KV cache is kept alive per model)")
        if past_key_values.seq_len + S > past_key_values.max_seq:
            raise ValueError(f"{past_key_values.seq_len} cached + {S} new tokens exceed the cache's {past_key_values.max_seq} positions "
                             f"(pass max_length= to the first forward(); the model's limit is {limit})")
        if S != 1:
            # any number of new tokens next to a cache (model/video_chatgpt.py:193-251; a later chat turn): prefill them behind the cached ones
            _kv, _nxt, _lg, all_lg = self.prefill([row.tolist() for row in ids.cpu()], video_spatio_temporal_features, 0, want_all_logits=True,
                                                  append_to=past_key_values.kv)
            past_key_values.seq_len += S
            return CausalLMOutputWithPast(all_lg.reshape(B, S, -1), past_key_values)
        _nxt, lg = self.decode_step(past_key_values.kv, ids[:, 0].to(device=self.device_, dtype=torch.int32).contiguous(), want_logits=True)
        past_key_values.seq_len += 1
        return CausalLMOutputWithPast(lg[:, None, :], past_key_values)

    # ---- generate ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids, video_spatio_temporal_features: Optional[torch.Tensor] = None, do_sample: bool = False,
                 temperature: float = 1.0, max_new_tokens: int = 1024, stopping_criteria=None, eos_token_id="config",
                 chunk: int = 32, generator: Optional[torch.Generator] = None, top_k: int = 50,
                 stop_strings: Optional[Sequence[Optional[str]]] = None, tokenizer=None, timings: Optional[dict] = None,
                 kv_reuse_key=None, **_unused):
        """Returns LongTensor [B, S + n_new] on the device, prompt echoed (checked by inference.py:115-117).

        input_ids: LongTensor [B, S] (equal-length prompts, the reference's case) or a list of id lists (ragged batch).
        Tokens never leave the device inside a chunk of `chunk` steps, in both modes:
          * greedy: pgv_llm_decode_greedy;
          * do_sample=True (the reference's default, temperature 0.2, inference.py:109-110): HF's sample loop is logits / temperature ->
            top-k mask (top_k = 50 is HF's default GenerationConfig value, which the reference inherits) -> softmax -> multinomial.  The
            multinomial draw is an inverse-CDF pick in a HIP kernel (pgv_llm_decode_sample) fed with uniforms drawn up front from
            `generator` (a device torch.Generator; one uniform per sequence and step, so a run is reproducible from the seed and
            independent of the chunk size).
        Stopping criteria (B == 1, as in the reference) are evaluated on the host after each chunk, token by token in generation
        order, and the output is cut at the first hit: the same ids as a per-token loop, the surplus steps of the chunk are discarded.
        `stop_strings` (one per sequence, None = no stop string; needs `tokenizer`): the BATCHED form of the reference's
        KeywordsStoppingCriteria (model/utils.py:6-26, passed at inference.py:101-102 with batch size 1): after every chunk each live
        sequence's new ids are checked with `first_stop_length` (the exact per-token criterion, scanned incrementally); a sequence that
        fired is cut there and finished, and the loop ends when every sequence has finished -- at most one chunk past the last stop.
        `timings` (a dict) receives prefill_s / decode_s / steps of this call (host clock; the chunk boundaries are D2H syncs anyway).
        `kv_reuse_key` (B == 1; any hashable naming the conversation's video, None = off): keep the KV cache of this call for the next one with
        the same key -- a later chat turn whose prompt extends what the cache holds prefills only the NEW tokens behind the common prefix
        (pgv_kv_truncate + pgv_llm_prefill_append; the reference re-runs the whole conversation every turn, chat.py:108-160).  The video run
        must lie inside the common prefix (it was spliced when the cache was filled); otherwise the call falls back to a full prefill.  The
        appended rows are bitwise what one prefill over the same cache contents computes; cache entries that DECODE steps wrote (the previous
        answer) carry the decode path's 16-bit rounding instead of the prefill path's -- as with any KV cache that is kept across calls.
        `timings["reused_tokens"]` reports the prefix."""
        if torch.is_tensor(input_ids):
            seqs = [row.tolist() for row in input_ids.cpu()]
        else:
            seqs = [list(s) for s in input_ids]
        B = len(seqs)
        eos = self.config.eos_token_id if eos_token_id == "config" else eos_token_id
        eos_i = -1 if eos is None else int(eos)
        max_seq = max(len(s) for s in seqs) + max_new_tokens
        if max_seq > self.config.max_position_embeddings:
            raise ValueError(f"prompt + max_new_tokens = {max_seq} exceeds max_position_embeddings {self.config.max_position_embeddings}")
        max_seq = (max_seq + 63) // 64 * 64
        if kv_reuse_key is not None and B == 1:
            max_seq = min(int(self.config.max_position_embeddings), 4096)      # one cache for the whole conversation: later turns grow into it
        criteria = list(stopping_criteria) if stopping_criteria else []
        if criteria and B != 1:
            raise ValueError("stopping_criteria are evaluated on one sequence (the reference passes them with batch size 1 only)")
        if do_sample and not float(temperature) > 0.0:
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float; use do_sample=False for greedy decoding")
        if criteria:
            chunk = min(chunk, 16)                         # at most 15 surplus steps past a stop string
        stops = list(stop_strings) if stop_strings is not None else None
        if stops is not None:
            if len(stops) != B or tokenizer is None:
                raise ValueError("stop_strings needs one entry per sequence and the tokenizer that decodes them")
            if not any(stops):
                stops = None
        scanned = [2] * B                                  # stop strings: prefixes shorter than this are known not to fire
        final = [False] * B                                # ... and sequences whose stop check is complete
        import time as _time
        _t0 = _time.perf_counter()

        reused = 0
        if kv_reuse_key is not None and B == 1:
            reused = self._reusable_prefix(kv_reuse_key, seqs[0], max_new_tokens)
        if reused:
            kv = self._reuse[1]
            if reused < len(self._reuse[3]):
                _lib.check(self.ctx.lib.pgv_kv_truncate(kv, 0, reused, _lib.stream_ptr(self.device_)), "pgv_kv_truncate")
            _kv, nxt, _ = self.prefill([seqs[0][reused:]], None, 0, append_to=kv)
        else:
            kv, nxt, _ = self.prefill(seqs, video_spatio_temporal_features, max_seq)
        raw: List[int] = []                                # B == 1: every token the device emitted, stop cuts ignored (= what the cache holds, shifted by one)
        u = None
        if do_sample and max_new_tokens >= 1:
            u = torch.rand(max_new_tokens, B, device=self.device_, dtype=torch.float32, generator=generator)
            nxt = self.sample_last(kv, u[0], temperature, top_k)
        new = [[] for _ in range(B)]
        done = [False] * B
        prompt_t = [torch.tensor(s, dtype=torch.long) for s in seqs]

        def absorb(tok_host) -> bool:
            """Append one step's tokens; True = generation is over (HF: EOS on every sequence, or a criterion fires on the new ids)."""
            for b in range(B):
                if not done[b]:
                    new[b].append(int(tok_host[b]))
                    if eos_i >= 0 and int(tok_host[b]) == eos_i:
                        done[b] = True
            if all(done):
                return True
            if criteria:
                cur = torch.cat([prompt_t[0], torch.tensor(new[0], dtype=torch.long)])[None]
                if any(c(cur, None) for c in criteria):
                    return True
            return False

        def check_stops() -> bool:
            """Per-chunk stop-string check of the live sequences; True = every sequence has finished."""
            from .utils import first_stop_length
            for b in range(B):
                if final[b] or not stops[b] or len(new[b]) < 2:
                    continue
                n = first_stop_length(new[b], tokenizer, [stops[b]], start=scanned[b])
                if n is not None:
                    del new[b][n:]                         # a stop string ahead of an EOS of the same chunk wins, as in the per-token loop
                    done[b] = final[b] = True
                elif done[b]:
                    final[b] = True                        # finished by EOS, no stop string before it
                else:
                    scanned[b] = len(new[b]) + 1
            return all(done)

        n_gen = 0
        _t1 = _t0
        if max_new_tokens >= 1:
            first = nxt.cpu()
            raw.append(int(first[0]))
            stop = absorb(first)
            _t1 = _time.perf_counter()
            n_gen = 1
            while not stop and n_gen < max_new_tokens:
                n = min(chunk, max_new_tokens - n_gen)
                if do_sample:
                    toks = self.decode_sample(kv, nxt, n, eos_i, temperature, top_k, u[n_gen:n_gen + n])
                else:
                    toks = self.decode_greedy(kv, nxt, n, eos_i)
                host = toks.cpu()
                raw.extend(host[0].tolist())
                if criteria:
                    for i in range(n):
                        stop = absorb(host[:, i])
                        n_gen += 1
                        if stop:
                            break
                else:
                    # the same bookkeeping per SEQUENCE instead of per step (a chunk of 64 steps x 8 sequences cost ~0.6 ms of Python between two
                    # decode launches, with the GPU idle): every live sequence takes its tokens up to and including its first EOS
                    h = host.numpy()
                    last = 0
                    for b in range(B):
                        if done[b]:
                            continue
                        k = n
                        if eos_i >= 0:
                            hit = np.flatnonzero(h[b] == eos_i)
                            if hit.size:
                                k, done[b] = int(hit[0]) + 1, True
                        new[b].extend(h[b, :k].tolist())
                        last = max(last, k)
                    stop = all(done)
                    n_gen += last if stop else n
                if stops is not None:                   # also when the chunk ended with every sequence at EOS: a stop string AHEAD of that EOS still cuts
                    stop = check_stops() or stop
                nxt = toks[:, n - 1].contiguous()
        if timings is not None:
            _t2 = _time.perf_counter()
            timings.update(prefill_s=_t1 - _t0, decode_s=_t2 - _t1, steps=n_gen, batch=B, reused_tokens=reused)
        if kv_reuse_key is not None and B == 1:
            # the cache holds the prompt and every emitted token that was fed back (all but the last one)
            self._reuse = (kv_reuse_key, kv, max_seq, list(seqs[0]) + raw[:-1])
            assert self.ctx.lib.pgv_kv_len(kv, 0) == len(self._reuse[3]), (self.ctx.lib.pgv_kv_len(kv, 0), len(self._reuse[3]))
        width = max(len(s) + len(n_) for s, n_ in zip(seqs, new))
        pad = eos_i if eos_i >= 0 else 0
        out = torch.full((B, width), pad, dtype=torch.long)
        for b in range(B):
            row = seqs[b] + new[b]
            out[b, :len(row)] = torch.tensor(row, dtype=torch.long)
        return out.to(self.device_)


def _vision_config_from(mm_vision_tower) -> VisionConfig:
    """VisionConfig from the CLIP config the checkpoint names (model/video_chatgpt.py:43-49).  Accepts a local
    directory with a CLIP config.json, a CLIPVisionTowerConfig-like object, or the two published hub names."""
    if mm_vision_tower is None:
        return VisionConfig()
    if hasattr(mm_vision_tower, "image_size"):
        c = mm_vision_tower
        return VisionConfig(frame_size=c.image_size, patch_size=c.patch_size, hidden_size=c.hidden_size)
    if isinstance(mm_vision_tower, str) and os.path.isdir(mm_vision_tower):
        from ..vision_tower import CLIPVisionTowerConfig
        c = CLIPVisionTowerConfig.from_pretrained(mm_vision_tower)
        return VisionConfig(frame_size=c.image_size, patch_size=c.patch_size, hidden_size=c.hidden_size)
    known = {"openai/clip-vit-large-patch14": 224, "openai/clip-vit-large-patch14-336": 336}
    if mm_vision_tower in known:
        return VisionConfig(frame_size=known[mm_vision_tower], patch_size=14, hidden_size=1024)
    raise ValueError(f"cannot resolve mm_vision_tower={mm_vision_tower!r} offline: pass a local CLIP directory")
