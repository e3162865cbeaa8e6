"""Model / tokenizer / CLIP loading and frame sampling with the reference's names and return shapes
(video_chatgpt/eval/model_utils.py:12-52 `load_video`, :55-79 `get_seq_frames`, :82-150 `initialize_model`)."""
from __future__ import annotations

import os

import numpy as np
import torch

from ..constants import DEFAULT_VID_END_TOKEN, DEFAULT_VID_START_TOKEN, DEFAULT_VIDEO_PATCH_TOKEN, NUM_TEMPORAL_TOKENS
from ..utils import disable_torch_init


def get_seq_frames(total_num_frames: int, desired_num_frames: int) -> list:
    """Uniform sampling: index i is the integer midpoint of the rounded ends of segment i, segments being (n-1)/k long;
    rounding is numpy's round-half-to-even (reference eval/model_utils.py:55-79; pinned by tests/golden/seq_frames.json)."""
    seg = float(total_num_frames - 1) / desired_num_frames
    ends = [int(np.round(seg * i)) for i in range(desired_num_frames + 1)]
    return [(ends[i] + ends[i + 1]) // 2 for i in range(desired_num_frames)]


def load_video(vis_path, n_clips=1, num_frm=100, shape=(224, 224)):
    """Decode up to `num_frm` uniformly sampled frames and return PIL images at `shape` (nearest resize, no aspect
    preservation) -- reference eval/model_utils.py:12-52.  Video decoding needs `decord`, which this image does not
    ship; synthetic-clip callers hand uint8 arrays to video_chatgpt_infer directly."""
    try:
        from decord import VideoReader, cpu
    except ImportError as e:
        raise RuntimeError("load_video needs the `decord` package to decode video files") from e
    from PIL import Image
    assert n_clips == 1                                     # the reference supports a single clip only (:30)
    vr = VideoReader(vis_path, ctx=cpu(0))
    total = len(vr)
    k = min(total, num_frm)
    arr = vr.get_batch(get_seq_frames(total, k)).asnumpy()
    th, tw = shape
    if arr.shape[-3] != th or arr.shape[-2] != tw:
        t = torch.from_numpy(arr).permute(0, 3, 1, 2).float()
        t = torch.nn.functional.interpolate(t, size=(th, tw))
        arr = t.permute(0, 2, 3, 1).to(torch.uint8).numpy()
    return [Image.fromarray(arr[j]) for j in range(k)]


def initialize_model(model_name, projection_path=None, torch_dtype=torch.float16):
    """-> (model, vision_tower, tokenizer, image_processor, video_token_len), as eval/model_utils.py:82-150.
    `model_name` is a local checkpoint directory (config.json + weights + tokenizer); its `mm_vision_tower` must
    resolve to a local CLIP directory (no network here)."""
    from transformers import AutoTokenizer, CLIPImageProcessor

    from ..model.video_chatgpt import VideoChatGPTLlamaForCausalLM
    from ..vision_tower import CLIPVisionTower

    disable_torch_init()
    model_name = os.path.expanduser(model_name)
    tokenizer = AutoTokenizer.from_pretrained(model_name)
    model = VideoChatGPTLlamaForCausalLM.from_pretrained(model_name, low_cpu_mem_usage=True, torch_dtype=torch_dtype, use_cache=True)
    image_processor = CLIPImageProcessor.from_pretrained(model.config.mm_vision_tower)

    mm_use_vid_start_end = True
    tokenizer.add_tokens([DEFAULT_VIDEO_PATCH_TOKEN], special_tokens=True)
    if mm_use_vid_start_end:
        tokenizer.add_tokens([DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))

    if projection_path:
        print(f"Loading weights from {projection_path}")
        status = model.load_state_dict(torch.load(projection_path, map_location="cpu"), strict=False)
        if status.unexpected_keys:
            print(f"Unexpected Keys: {status.unexpected_keys}.\nThe Video-ChatGPT weights are not loaded correctly.")
        print(f"Weights loaded from {projection_path}")

    model = model.eval().cuda()
    vision_tower = CLIPVisionTower.from_pretrained(model.config.mm_vision_tower, torch_dtype=torch_dtype, low_cpu_mem_usage=True).cuda().eval()

    vision_config = model.get_model().vision_config
    vision_config.vid_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_VIDEO_PATCH_TOKEN])[0]
    vision_config.use_vid_start_end = mm_use_vid_start_end
    if mm_use_vid_start_end:
        vision_config.vid_start_token, vision_config.vid_end_token = tokenizer.convert_tokens_to_ids(
            [DEFAULT_VID_START_TOKEN, DEFAULT_VID_END_TOKEN])
    num_patches_per_frame = (vision_config.frame_size // vision_config.patch_size) ** 2
    video_token_len = num_patches_per_frame + NUM_TEMPORAL_TOKENS
    return model, vision_tower, tokenizer, image_processor, video_token_len
