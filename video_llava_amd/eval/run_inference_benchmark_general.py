"""Video-ChatGPT generative benchmark (correctness / detail / context / temporal) runner with the reference's CLI and output
schema (video_chatgpt/eval/run_inference_benchmark_general.py:9-95): samples `{video_name, Q, A, ...}` from one ground-truth
JSON, video file `{video_name}.{mp4,avi,mov,mkv}`, output = every answered sample with a `pred` key added.
The loop itself is `_sharded.answer_tasks` (data-parallel over ranks, batched greedy decode)."""
from __future__ import annotations

import argparse
import json
import os

VIDEO_FORMATS = [".mp4", ".avi", ".mov", ".mkv"]


def parse_args(argv=None):
    from ._sharded import add_runtime_arguments
    parser = argparse.ArgumentParser()
    parser.add_argument("--video_dir", help="Directory containing video files.", required=True)
    parser.add_argument("--gt_file", help="Path to the ground truth file.", required=True)
    parser.add_argument("--output_dir", help="Directory to save the model results JSON.", required=True)
    parser.add_argument("--output_name", help="Name of the file for storing results JSON.", required=True)
    parser.add_argument("--model-name", type=str, required=True)
    parser.add_argument("--conv-mode", type=str, required=False, default="pg-video-llava")
    parser.add_argument("--projection_path", type=str, required=True)
    parser.add_argument("--use_asr", action="store_true", help="Whether to use audio transcripts or not")
    return add_runtime_arguments(parser).parse_args(argv)


def run_inference(args, components=None, load_frames=None, questions=("Q",), pred_keys=("pred",)):
    """`questions` / `pred_keys`: the sample keys to ask and the keys the answers are stored under (the consistency benchmark asks two)."""
    from ..feature_extraction import load_video
    from . import _sharded

    rank, world, components = _sharded.setup(args, components)
    image_processor = components[3]
    frame_size = (image_processor.crop_size["height"], image_processor.crop_size["width"])
    load_frames = load_frames or (lambda path: load_video(path, shape=frame_size, device_resize=True))
    with open(args.gt_file) as f:
        gt_contents = json.load(f)
    os.makedirs(args.output_dir, exist_ok=True)
    tasks = []
    for sample in gt_contents:                                       # :57-69 -- extension order of the reference
        path = _sharded.first_existing(args.video_dir, sample["video_name"], VIDEO_FORMATS + list(_sharded.DECORD_FREE_FORMATS))
        tasks += [{"path": path, "name": sample["video_name"], "question": sample[q]} for q in questions]
    preds = _sharded.answer_tasks(args, tasks, components, load_frames, rank, world)
    output_list = []
    for i, sample in enumerate(gt_contents):
        mine = preds[i * len(questions):(i + 1) * len(questions)]
        if all(p is not None for p in mine):                         # any failure drops the whole sample (one try block, :78-87)
            sample_set = dict(sample)
            sample_set.update(dict(zip(pred_keys, mine)))
            output_list.append(sample_set)
    return _sharded.write_output(args, output_list, rank)


if __name__ == "__main__":
    run_inference(parse_args())
