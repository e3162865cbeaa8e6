"""Single-video QA entry points with the reference's names and signatures
(video_chatgpt/inference.py:13-44 `get_spatio_temporal_features_torch`, :47-124 `video_chatgpt_infer`).

Device work -- CLIP tower, pooling, projector, decoder -- runs in libpgv (HIP, gfx950); this module is the host
orchestration: prompt assembly, tokenisation, preprocessing, stop handling, detokenisation.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .constants import (DEFAULT_TRANSCRIPT_START, DEFAULT_VID_END_TOKEN, DEFAULT_VID_START_TOKEN, DEFAULT_VIDEO_PATCH_TOKEN,
                        DEFAULT_VIDEO_TOKEN, NUM_TEMPORAL_TOKENS)
from .model.utils import KeywordsStoppingCriteria
from .video_conversation import SeparatorStyle, conv_templates


def get_spatio_temporal_features_torch(features: torch.Tensor) -> torch.Tensor:
    """[T, P, C] frame features -> [100 + P, C] fp16: per-frame means (zero padded to 100 rows) followed by per-patch
    means over time (reference video_chatgpt/inference.py:13-44).  One HIP pass; the `[:, 1:]` view of the hidden
    state is consumed in place."""
    if not features.is_cuda:
        raise RuntimeError("get_spatio_temporal_features_torch: features must live on the GPU (no CPU fallback in this package)")
    if features.stride(2) != 1 or features.stride(1) != features.shape[2]:
        features = features.contiguous()
    if features.dtype not in (torch.float16, torch.bfloat16):
        features = features.half()
    return _lib.Context.get(features.device).st_pool(features, NUM_TEMPORAL_TOKENS, torch.float16)


def build_prompt(question: str, conv_mode: str, video_token_len: int, use_vid_start_end: bool, transcript=None):
    """Prompt string + stop string exactly as video_chatgpt_infer assembles them (inference.py:66-80,101)."""
    if use_vid_start_end:
        qs = question + "\n" + DEFAULT_VID_START_TOKEN + DEFAULT_VIDEO_PATCH_TOKEN * video_token_len + DEFAULT_VID_END_TOKEN
    else:
        qs = question + "\n" + DEFAULT_VIDEO_PATCH_TOKEN * video_token_len
    if transcript:
        qs = f'{qs}\n{DEFAULT_TRANSCRIPT_START}\n"{transcript}"'
    conv = conv_templates[conv_mode].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    stop_str = conv.sep if conv.sep_style != SeparatorStyle.TWO else conv.sep2
    return conv.get_prompt(), stop_str


def frames_to_pixels(video_frames, image_processor, vision_tower) -> torch.Tensor:
    """Frames -> normalised NCHW tensor on the tower's device.  uint8 arrays already at the crop size take the fused
    HIP path (pgv_preprocess_u8 == CLIPImageProcessor for crop-sized frames); anything else goes through
    the caller's image_processor exactly like the reference (inference.py:86-89)."""
    S = vision_tower.config.image_size
    from .feature_extraction import NativeFrames
    if isinstance(video_frames, NativeFrames):
        # frames at the decoder's resolution: load_video's nearest resize + CLIPImageProcessor in one HIP pass (pgv_ingest_u8)
        if video_frames.shape == (S, S):
            dev = vision_tower.device
            return _lib.Context.get(dev).ingest_u8(torch.from_numpy(video_frames.array).to(dev), S, vision_tower.dtype)
        video_frames = video_frames.resized()                # not the tower's crop size: host resize, then the caller's image_processor
        video_frames = [video_frames[i] for i in range(video_frames.shape[0])]
    if isinstance(video_frames, np.ndarray) and video_frames.dtype == np.uint8 and video_frames.shape[1:] == (S, S, 3):
        dev = vision_tower.device
        return _lib.Context.get(dev).preprocess_u8(torch.from_numpy(video_frames).to(dev), vision_tower.dtype)
    if torch.is_tensor(video_frames) and video_frames.dtype == torch.uint8 and tuple(video_frames.shape[1:]) == (S, S, 3):
        dev = vision_tower.device
        return _lib.Context.get(dev).preprocess_u8(video_frames.to(dev).contiguous(), vision_tower.dtype)
    px = image_processor.preprocess(video_frames, return_tensors="pt")["pixel_values"]
    return px.to(vision_tower.dtype).to(vision_tower.device)


def video_features(video_frames, vision_tower, image_processor) -> torch.Tensor:
    """frames -> [100 + P, 1024] fp16 pooled CLIP features (inference.py:86-95)."""
    image_tensor = frames_to_pixels(video_frames, image_processor, vision_tower)
    with torch.no_grad():
        out = vision_tower(image_tensor, output_hidden_states=True)
        frame_features = out.hidden_states[-2][:, 1:]          # second-to-last layer, CLS dropped
    return get_spatio_temporal_features_torch(frame_features)


GROUP_UPLOAD_MAX_BYTES = 512 << 20      # native-resolution bytes uploaded and ingested together (a run of same-sized clips is cut at this budget)


def video_features_batch(clips, vision_tower, image_processor, timings=None):
    """Several clips (ragged frame counts) -> list of [100 + P, 1024] fp16 pooled features with ONE tower pass over the concatenated frames.
    The tower is bitwise batch-split invariant (tests/test_gpu_vision.py::test_vit_100_frames_properties), so every entry equals
    `video_features(clip)` bit for bit; what changes is the launch shape: 8 clips x 100 frames fill the persistent GEMM's 256 CUs with
    3216 + tiles per launch instead of 404, and there is one launch sequence instead of eight.  Clips that share a native resolution are
    uploaded and ingested (nearest resize + CLIP normalisation, pgv_ingest_u8) together, in runs of at most GROUP_UPLOAD_MAX_BYTES
    native bytes copied into ONE destination buffer (no per-clip device tensors + torch.cat: the transient device memory of a group is the
    run's bytes once, not twice -- the per-clip DEVICE_RESIZE_MAX_BYTES bound of load_video is not undone by grouping).
    `timings` (a dict) receives upload_ingest_s / tower_pool_s from events on the launch stream (one synchronisation at the end)."""
    from .feature_extraction import NativeFrames
    S = vision_tower.config.image_size
    dev = vision_tower.device
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timings is not None else None
    if ev:
        ev[0].record()
    px_parts, counts = [], []
    i = 0
    while i < len(clips):
        c = clips[i]
        j = i + 1
        if isinstance(c, NativeFrames) and c.shape == (S, S):
            # run of clips decoded at the same native resolution: one destination buffer, one ingest launch
            run_bytes = c.array.nbytes
            while (j < len(clips) and isinstance(clips[j], NativeFrames) and clips[j].shape == (S, S) and clips[j].array.shape[1:] == c.array.shape[1:]
                   and run_bytes + clips[j].array.nbytes <= GROUP_UPLOAD_MAX_BYTES):
                run_bytes += clips[j].array.nbytes
                j += 1
            host = [torch.as_tensor(x.array) for x in clips[i:j]]
            if len(host) == 1:
                up = host[0].to(dev, non_blocking=True)
            else:
                up = torch.empty((sum(int(h.shape[0]) for h in host),) + tuple(host[0].shape[1:]), dtype=torch.uint8, device=dev)
                o = 0
                for h in host:
                    up[o:o + h.shape[0]].copy_(h, non_blocking=True)
                    o += int(h.shape[0])
            px_parts.append(_lib.Context.get(dev).ingest_u8(up.contiguous(), S, vision_tower.dtype))
            del up
            counts.extend(int(h.shape[0]) for h in host)
        else:
            px = frames_to_pixels(c, image_processor, vision_tower)
            px_parts.append(px)
            counts.append(int(px.shape[0]))
        i = j
    px = torch.cat(px_parts) if len(px_parts) > 1 else px_parts[0]
    del px_parts
    if ev:
        ev[1].record()
    with torch.no_grad():
        hid = vision_tower(px, output_hidden_states=True).hidden_states[-2]
    out, off = [], 0
    for t in counts:
        out.append(get_spatio_temporal_features_torch(hid[off:off + t, 1:]))
        off += t
    if ev:
        ev[2].record()
        ev[2].synchronize()
        timings.update(upload_ingest_s=ev[0].elapsed_time(ev[1]) * 1e-3, tower_pool_s=ev[1].elapsed_time(ev[2]) * 1e-3, frames=int(sum(counts)))
    return out


def video_chatgpt_infer_ids(video_frames, question, conv_mode, model, vision_tower, tokenizer, image_processor, video_token_len,
                            transcript=None, do_sample=True, temperature=0.2, max_new_tokens=1024, features=None):
    """`video_chatgpt_infer` up to (not including) the detokenisation: -> (generated token ids [n] as a list, stop string).  The runners'
    sampling path collates these ids through the same fixed-shape all-gather as the greedy path (parallel.gather_answers) and decodes them
    on every rank.  `features`: pooled [100 + P, 1024] features computed earlier for this clip (skips the tower)."""
    prompt, stop_str = build_prompt(question, conv_mode, video_token_len, model.get_model().vision_config.use_vid_start_end, transcript)
    inputs = tokenizer([prompt])
    feats = features if features is not None else video_features(video_frames, vision_tower, image_processor)
    input_ids = torch.as_tensor(inputs.input_ids)
    stopping_criteria = KeywordsStoppingCriteria([stop_str], tokenizer, input_ids)
    with torch.inference_mode():
        output_ids = model.generate(input_ids, video_spatio_temporal_features=feats.unsqueeze(0), do_sample=do_sample,
                                    temperature=temperature, max_new_tokens=max_new_tokens, stopping_criteria=[stopping_criteria])
    n_in = input_ids.shape[1]
    n_diff = int((input_ids.to(output_ids.device) != output_ids[:, :n_in]).sum())
    if n_diff > 0:
        print(f"[Warning] {n_diff} output_ids are not the same as the input_ids")
    return output_ids[0, n_in:].tolist(), stop_str


def video_chatgpt_infer(video_frames, question, conv_mode, model, vision_tower, tokenizer, image_processor, video_token_len,
                        transcript=None, do_sample=True, temperature=0.2, max_new_tokens=1024):
    """Answer `question` about one clip.  Same positional signature and defaults as the reference
    (video_chatgpt/inference.py:47; do_sample=True, temperature=0.2, max_new_tokens=1024 at :109-111); pass
    do_sample=False for deterministic greedy decoding."""
    new_ids, stop_str = video_chatgpt_infer_ids(video_frames, question, conv_mode, model, vision_tower, tokenizer, image_processor, video_token_len,
                                                transcript, do_sample, temperature, max_new_tokens)
    outputs = tokenizer.batch_decode([new_ids], skip_special_tokens=True)[0]
    return outputs.strip().rstrip(stop_str).strip()
