"""TGIF-QA runner (video_chatgpt/eval/run_inference_qa_tgif.py:16-114): samples are the rows of a tab-separated ground-truth file
(`gif_name`, `question`, `description`, ...), the clip is `{gif_name}.gif`, sampled at 8 segment-centred frames (:31-50) at the
GIF's native size (the image processor resizes), output = the row as a dict with `pred` added."""
from __future__ import annotations

import json
import os

import numpy as np

from .run_inference_qa_msrvtt import parse_args


def gif_frame_indices(num_frames: int, num_segments: int = 8):
    """start + round(seg * idx) with seg = (n - 1) / segments, start = int(seg / 2)  (:32-38; np.round = round-half-even)."""
    seg_size = float(num_frames - 1) / num_segments
    start = int(seg_size / 2)
    return [start + int(np.round(seg_size * idx)) for idx in range(num_segments)]


def load_video_from_gif(video_path, num_segments=8, shape=None):
    """-> list of PIL RGB frames (native size; `shape` is accepted and ignored exactly like the reference, :31)."""
    from PIL import Image
    gif = Image.open(video_path)
    images_group = []
    for i in gif_frame_indices(gif.n_frames, num_segments):
        gif.seek(i)
        images_group.append(Image.fromarray(np.array(gif.copy().convert("RGB"))))
    return images_group


def run_inference(args, components=None, load_frames=None):
    import pandas as pd
    from . import _sharded

    rank, world, components = _sharded.setup(args, components)
    load_frames = load_frames or load_video_from_gif
    os.makedirs(args.output_dir, exist_ok=True)
    rows = [row.to_dict() for _, row in pd.read_csv(args.gt_file, sep="\t").iterrows()]
    tasks = [{"path": _sharded.first_existing(args.video_dir, r["gif_name"], [".gif", *_sharded.DECORD_FREE_FORMATS]), "name": r["gif_name"], "question": r["question"]} for r in rows]
    preds = _sharded.answer_tasks(args, tasks, components, load_frames, rank, world)
    output_list = []
    for r, p in zip(rows, preds):
        if p is not None:
            output_set = json.loads(json.dumps(r, default=lambda o: o.item() if hasattr(o, "item") else str(o)))   # numpy scalars -> JSON types
            output_set["pred"] = p
            output_list.append(output_set)
    return _sharded.write_output(args, output_list, rank)


if __name__ == "__main__":
    run_inference(parse_args())
