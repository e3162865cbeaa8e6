"""MSRVTT-QA runner (video_chatgpt/eval/run_inference_qa_msrvtt.py:11-83): samples `{video_id, question, ...}` from one JSON,
video file `video{video_id}.mp4` (:47, :51), output = every answered sample with `pred` added (:63-64)."""
from __future__ import annotations

import argparse
import json
import os


def parse_args(argv=None, mapper=False):
    from ._sharded import add_runtime_arguments
    parser = argparse.ArgumentParser()
    parser.add_argument("--video_dir", help="dir containing video files", required=True)
    parser.add_argument("--gt_file", help="path to gt", required=True)
    if mapper:
        parser.add_argument("--mapper", help="path to mapper", required=True)
    parser.add_argument("--output_dir", help="dir to save model result json", required=True)
    parser.add_argument("--output_name", help="name of the file for storing result json", required=True)
    parser.add_argument("--model-name", type=str, required=True)
    parser.add_argument("--conv-mode", type=str, required=False, default="pg-video-llava")
    parser.add_argument("--projection_path", type=str, required=True)
    parser.add_argument("--use_asr", action="store_true", help="Whether to use audio transcripts or not")
    return add_runtime_arguments(parser).parse_args(argv)


def run_inference(args, components=None, load_frames=None, video_stem=None, extensions=(".mp4",)):
    """`video_stem(sample) -> file stem`; MSRVTT: `video{video_id}`."""
    from ..feature_extraction import load_video
    from . import _sharded

    rank, world, components = _sharded.setup(args, components)
    image_processor = components[3]
    frame_size = (image_processor.crop_size["height"], image_processor.crop_size["width"])
    load_frames = load_frames or (lambda path: load_video(path, shape=frame_size, device_resize=True))
    video_stem = video_stem or (lambda sample: f"video{sample['video_id']}")
    os.makedirs(args.output_dir, exist_ok=True)
    with open(args.gt_file) as f:
        gt_contents = json.load(f)
    tasks = []
    for sample in gt_contents:
        stem = video_stem(sample)
        tasks.append({"path": _sharded.first_existing(args.video_dir, stem, list(extensions) + list(_sharded.DECORD_FREE_FORMATS)), "name": stem, "question": sample["question"]})
    preds = _sharded.answer_tasks(args, tasks, components, load_frames, rank, world)
    output_list = []
    for sample, p in zip(gt_contents, preds):
        if p is not None:
            output_set = dict(sample)
            output_set["pred"] = p
            output_list.append(output_set)
    return _sharded.write_output(args, output_list, rank)


if __name__ == "__main__":
    run_inference(parse_args())
