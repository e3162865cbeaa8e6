"""A stand-in for the `decord` package (absent offline) -- TEST INFRASTRUCTURE (see oracle/__init__.py).

`install()` registers a module named `decord` whose `VideoReader(path, ctx=cpu(0))` serves synthetic clips, so the decord branches of
the reference's `load_video` (video_chatgpt/eval/model_utils.py:12-52, scripts/save_spatio_temporal_clip_features.py:13-32) and of this
package's mirrors execute: `len(vr)`, `vr.get_batch(indices).asnumpy()`.  A path of the form `synth:<frames>x<H>x<W>:<seed>` yields
`numpy.random.default_rng(seed).integers(0, 256, (frames, H, W, 3), uint8)`; a `.npy` path yields the stored array."""
from __future__ import annotations

import sys
import types

import numpy as np


def synth_clip(path: str) -> np.ndarray:
    if path.startswith("synth:"):
        _, dims, seed = path.split(":")
        t, h, w = (int(x) for x in dims.split("x"))
        return np.random.default_rng(int(seed)).integers(0, 256, (t, h, w, 3), dtype=np.uint8)
    return np.load(path)


class _Batch:
    def __init__(self, arr):
        self._arr = arr

    def asnumpy(self):
        return self._arr


class VideoReader:
    def __init__(self, path, ctx=None):
        self._clip = synth_clip(str(path))

    def __len__(self):
        return self._clip.shape[0]

    def get_batch(self, indices):
        return _Batch(np.ascontiguousarray(self._clip[list(indices)]))


def install():
    d = types.ModuleType("decord")
    d.VideoReader = VideoReader
    d.cpu = lambda i: None
    sys.modules["decord"] = d
    return d


def uninstall():
    sys.modules.pop("decord", None)
