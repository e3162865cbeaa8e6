"""Offline CLIP feature extraction with the reference script's functions, CLI and on-disk format
(scripts/save_spatio_temporal_clip_features.py: `load_video` :13-32, `get_seq_frames` :35-43,
`get_spatio_temporal_features` :46-57, `parse_args` :60-71, `main` :74-139).

Per video it writes `{clip_feat_path}/{video_id}.pkl` = pickle(np.float16[100 + P, 1024]) -- the file the reference's
training loader opens (video_chatgpt/train/train.py:404-405).  Device work (preprocess, CLIP tower, pooling) runs in libpgv;
this module is the host loop: directory walk, idempotent skip, decode, periodic flush.

Differences that do not change the bytes written:
  * `--infer_batch` is accepted for drop-in compatibility; the tower processes all sampled frames of a clip in one call
    because its result is bit-identical for every split (tests/test_gpu_vision.py::test_vit_100_frames_properties), and the
    105 MB fp32 D2H copy per clip of the reference's batching loop (:108-121) disappears (features stay on the device until
    the 0.73 MB pooled result).
  * frames may also come from `<name>.npy` files (uint8 [T, H, W, 3]) -- `decord` is not installable offline here.
"""
from __future__ import annotations

import argparse
import os
import pickle

import numpy as np
import torch

from .eval.model_utils import get_seq_frames  # noqa: F401  (re-exported: the reference script defines its own copy)

LLAVA_VERSIONS = {"1.1": ("openai/clip-vit-large-patch14", (224, 224)), "1.5": ("openai/clip-vit-large-patch14-336", (336, 336))}
VIDEO_EXTENSIONS_NPY = (".npy",)


class NativeFrames:
    """Sampled frames at the decoder's resolution (uint8 [k, H, W, 3]) plus the size `load_video` would resize them to.  The resize is
    then done on the device, fused with the CLIP normalisation (pgv_ingest_u8: the same nearest-neighbour index rule, bit-exact), so
    the host never touches the pixels and the upload is the native uint8 frames."""
    __slots__ = ("array", "shape")

    def __init__(self, array: np.ndarray, shape):
        self.array, self.shape = array, (int(shape[0]), int(shape[1]))

    def __len__(self):
        return self.array.shape[0]

    def resized(self) -> np.ndarray:
        """Host twin of the device resize: exactly the reference's statements (eval/model_utils.py:38-43)."""
        return resize_nearest(self.array, self.shape)


class PinnedRing:
    """Page-locked host buffers for the runners' prefetch thread, reused instead of re-allocated (hipHostMalloc of 15 MB costs milliseconds
    per clip).  Two generations alternate: the buffers handed out for group g + 2 are those of group g, whose uploads completed before its
    answers were read back -- which happens before group g + 2 is prepared (parallel.run_sharded prepares exactly one group ahead)."""

    def __init__(self):
        self._gen = [[], []]
        self._turn = 0
        self._next = 0

    def new_group(self):
        self._turn ^= 1
        self._next = 0

    def take(self, shape):
        pool = self._gen[self._turn]
        n = int(np.prod(shape))
        if self._next < len(pool) and pool[self._next].numel() >= n:
            buf = pool[self._next]
        else:
            buf = torch.empty(max(n, 1), dtype=torch.uint8).pin_memory()
            if self._next < len(pool):
                pool[self._next] = buf
            else:
                pool.append(buf)
        self._next += 1
        return buf[:n].view(*shape)


def pin_frames(frames, ring: "PinnedRing | None" = None):
    """Move sampled frames (NativeFrames or a uint8 array) into page-locked host memory so the upload is one asynchronous DMA (called on the
    runners' prefetch thread).  Without a GPU runtime the frames are returned unchanged."""
    if not torch.cuda.is_available():
        return frames
    arr = frames.array if isinstance(frames, NativeFrames) else frames
    if not (isinstance(arr, np.ndarray) and arr.dtype == np.uint8):
        return frames
    pinned = ring.take(arr.shape) if ring is not None else torch.empty(arr.shape, dtype=torch.uint8).pin_memory()
    view = pinned.numpy()
    view[...] = arr
    return NativeFrames(view, frames.shape) if isinstance(frames, NativeFrames) else view


def resize_nearest(arr: np.ndarray, shape) -> np.ndarray:
    h, w = shape
    if arr.shape[-3] == h and arr.shape[-2] == w:
        return arr
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(0, 3, 1, 2).float()
    t = torch.nn.functional.interpolate(t, size=(h, w))
    return t.permute(0, 2, 3, 1).to(torch.uint8).numpy()


IMAGE_EXTENSIONS = (".jpg", ".jpeg", ".png", ".bmp", ".webp")
PIL_ANIMATED_EXTENSIONS = (".gif", ".webp", ".apng")


def sample_frames(vis_path, num_frm=100) -> np.ndarray:
    """Up to `num_frm` uniformly sampled frames (get_seq_frames) at the source's own resolution, uint8 [k, H, W, 3].
    Sources: a video file (through `decord`, like the reference), a `.npy` array of frames, or a DIRECTORY of image files (one frame per
    file, lexicographic order -- what `ffmpeg -i clip.mp4 frames/%06d.jpg` leaves behind), or an animated GIF / WebP / APNG decoded by Pillow:
    the decord-free front ends."""
    if os.path.isdir(vis_path):
        from PIL import Image
        names = sorted(n for n in os.listdir(vis_path) if n.lower().endswith(IMAGE_EXTENSIONS))
        if not names:
            raise ValueError(f"{vis_path}: no image files ({', '.join(IMAGE_EXTENSIONS)})")
        total = len(names)
        k = min(total, num_frm)
        frames = [np.asarray(Image.open(os.path.join(vis_path, names[i])).convert("RGB")) for i in get_seq_frames(total, k)]
        if any(f.shape != frames[0].shape for f in frames):
            raise ValueError(f"{vis_path}: frames of different sizes")
        return np.ascontiguousarray(np.stack(frames))
    if str(vis_path).endswith(VIDEO_EXTENSIONS_NPY):
        src = np.load(vis_path, mmap_mode="r")
        if src.dtype != np.uint8 or src.ndim != 4 or src.shape[-1] != 3:
            raise ValueError(f"{vis_path}: expected uint8 [T, H, W, 3], got {src.dtype} {src.shape}")
        total = src.shape[0]
        k = min(total, num_frm)
        return np.ascontiguousarray(src[get_seq_frames(total, k)])
    try:
        from decord import VideoReader, cpu
    except ImportError as e:
        if str(vis_path).lower().endswith(PIL_ANIMATED_EXTENSIONS):
            # animated GIF / WebP / APNG (the TGIF runner's files) without decord: Pillow decodes them frame by frame, composited to RGB
            from PIL import Image
            with Image.open(vis_path) as im:
                total = getattr(im, "n_frames", 1)
                k = min(total, num_frm)
                frames = []
                for i in get_seq_frames(total, k):            # ascending: a forward-only seek pattern
                    im.seek(int(i))
                    frames.append(np.asarray(im.convert("RGB")))
            return np.ascontiguousarray(np.stack(frames))
        raise RuntimeError("load_video needs the `decord` package to decode video files (or pass .npy frame arrays, a directory of frame "
                           "images, or an animated .gif / .webp / .apng)") from e
    vr = VideoReader(vis_path, ctx=cpu(0))
    total = len(vr)
    k = min(total, num_frm)
    return vr.get_batch(get_seq_frames(total, k)).asnumpy()


DEVICE_RESIZE_MAX_BYTES = 256 << 20     # native-resolution upload cap per clip (100 frames of 720p = 276 MB is already above; 480p = 92 MB)


def load_video(vis_path, num_frm=100, shape=(224, 224), device_resize=False):
    """Up to `num_frm` uniformly sampled frames as a uint8 array [k, h, w, 3]: nearest-neighbour resize to `shape` without aspect
    preservation, exactly the arithmetic of the reference (:13-32; F.interpolate default mode on the float tensor, cast back to
    uint8).  The reference returns PIL images only to feed CLIPImageProcessor; the HIP preprocessing takes the array directly.
    device_resize=True returns the frames un-resized as `NativeFrames`: inference.frames_to_pixels then resizes and normalises them
    in one HIP pass (pgv_ingest_u8)."""
    arr = sample_frames(vis_path, num_frm)
    if device_resize and arr.nbytes <= DEVICE_RESIZE_MAX_BYTES:
        return NativeFrames(arr, shape)
    # large sources (100 frames of 1080p = 620 MB, 4K = 2.5 GB) are resized on the host: the upload then is the 15 MB of crop-sized frames
    # instead of a transient device allocation that scales with the source resolution (next to a 13B model + KV cache that can be an OOM)
    return resize_nearest(arr, shape)


def get_spatio_temporal_features(features, num_temporal_tokens=100):
    """[t, s, c] -> np.float16 [num_temporal_tokens + s, c] (reference :46-57: np.mean over patches, zero padding to 100 rows,
    np.mean over frames, concatenate).  A device tensor is pooled by the HIP kernel (fp32 accumulation, one pass); a numpy array
    takes the reference's numpy arithmetic unchanged."""
    if isinstance(features, np.ndarray):
        t = features.shape[0]
        temporal = np.mean(features, axis=1)
        if num_temporal_tokens - t > 0:
            temporal = np.pad(temporal, ((0, num_temporal_tokens - t), (0, 0)), mode="constant")
        return np.concatenate([temporal, np.mean(features, axis=0)], axis=0)
    from . import _lib
    if features.shape[0] > num_temporal_tokens:
        raise ValueError(f"{features.shape[0]} frames > {num_temporal_tokens} temporal tokens (callers sample at most {num_temporal_tokens})")
    if features.dtype not in (torch.float16, torch.bfloat16):
        features = features.half()
    return _lib.Context.get(features.device).st_pool(features, num_temporal_tokens, torch.float16).cpu().numpy()


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Training")
    parser.add_argument("--llava", required=True, choices=["1.1", "1.5"], help="LLaVA version")
    parser.add_argument("--video_dir_path", required=True, help="Path to read the videos from.")
    parser.add_argument("--clip_feat_path", required=True, help="The output dir to save the features in.")
    parser.add_argument("--infer_batch", required=False, type=int, default=32, help="Number of frames/images to perform batch inference.")
    parser.add_argument("--clip_path", default=None, help="local directory of the CLIP checkpoint (default: the hub id of --llava)")
    return parser.parse_args(argv)


def extract_clip_features(video_path, vision_tower, frame_size):
    """One video -> np.float16 [100 + P, 1024] (the body of the reference's per-video try block, :103-123)."""
    from . import _lib
    frames = sample_frames(video_path)                      # native resolution: resize + normalise happen in one pass on the device
    dev = vision_tower.device
    if frame_size[0] != frame_size[1]:
        raise ValueError(f"square CLIP inputs only (got {frame_size})")
    px = _lib.Context.get(dev).ingest_u8(torch.from_numpy(frames).to(dev), frame_size[0], vision_tower.dtype)
    with torch.no_grad():
        hidden = vision_tower(px, output_hidden_states=True).hidden_states[-2]
    return get_spatio_temporal_features(hidden[:, 1:])


def run(args, vision_tower=None, save_every=512, log=print):
    """The reference's main loop (:74-139).  `vision_tower` may be injected (tests, random-init benchmarks)."""
    os.makedirs(args.clip_feat_path, exist_ok=True)
    hub_id, frame_size = LLAVA_VERSIONS[args.llava]
    if vision_tower is None:
        from .vision_tower import CLIPVisionTower
        vision_tower = CLIPVisionTower.from_pretrained(args.clip_path or hub_id, torch_dtype=torch.float16, low_cpu_mem_usage=True).cuda().eval()
    pending = {}
    counter = 0

    def flush():
        for key, feats in pending.items():
            with open(f"{args.clip_feat_path}/{key}.pkl", "wb") as f:
                pickle.dump(feats, f)
        pending.clear()

    for video_name in sorted(os.listdir(args.video_dir_path)):
        video_path = f"{args.video_dir_path}/{video_name}"
        video_id = video_name.split(".")[0]
        if os.path.exists(f"{args.clip_feat_path}/{video_id}.pkl"):      # already processed
            continue
        try:
            pending[video_id] = extract_clip_features(video_path, vision_tower, frame_size)
            counter += 1
        except Exception as e:                                           # noqa: BLE001 -- the reference prints and continues (:126-127)
            log(f"Can't process {video_path}: {e}")
        if counter % save_every == 0:
            flush()
    flush()
    return counter


def main(argv=None):
    run(parse_args(argv))


if __name__ == "__main__":
    main()
