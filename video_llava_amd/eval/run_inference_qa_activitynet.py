"""ActivityNet-QA inference runner with the reference's CLI and output schema
(video_chatgpt/eval/run_inference_qa_activitynet.py:9-113): for every question find `v_{video_name}.{mp4,avi,mov,mkv}`,
answer it, and write `[{"id", "question", "answer", "pred"}, ...]` to `{output_dir}/{output_name}.json`.

The reference loops over the samples serially on one GPU (:63-104).  Here the loop is `_sharded.answer_tasks`: samples sharded over the
ranks of a `torch.distributed.run` launch, batched per rank, one all-gather of the answer token ids; rank 0 writes the JSON.  A sample
whose video is missing or fails keeps the reference's policy: it is reported and left out of the output list (:103-104).
`--do_sample` restores the reference's temperature-0.2 sampling through `video_chatgpt_infer`, one sample at a time.
"""
from __future__ import annotations

import argparse
import json
import os

VIDEO_FORMATS = [".mp4", ".avi", ".mov", ".mkv"]


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--video_dir", help="Directory containing video files.", required=True)
    parser.add_argument("--gt_file_question", help="Path to the ground truth file containing question.", required=True)
    parser.add_argument("--gt_file_answers", help="Path to the ground truth file containing answers.", required=True)
    parser.add_argument("--output_dir", help="Directory to save the model results JSON.", required=True)
    parser.add_argument("--output_name", help="Name of the file for storing results JSON.", required=True)
    parser.add_argument("--model-name", type=str, required=True)
    parser.add_argument("--conv-mode", type=str, required=False, default="pg-video-llava")
    parser.add_argument("--projection_path", type=str, required=True)
    parser.add_argument("--use_asr", action="store_true", help="Whether to use audio transcripts or not")
    from ._sharded import add_runtime_arguments
    return add_runtime_arguments(parser).parse_args(argv)


def find_video(video_dir, video_name, extra_formats=(".npy", "")):
    """First existing `v_{video_name}{ext}` in the reference's extension order (:70-76); None when absent."""
    for fmt in list(VIDEO_FORMATS) + list(extra_formats):
        path = os.path.join(video_dir, f"v_{video_name}{fmt}")
        if os.path.exists(path):
            return path
    return None


def load_samples(gt_file_question, gt_file_answers):
    """[{video_name, question, id, answer}] -- questions and answers are paired by position like the reference (:63-69)."""
    with open(gt_file_question) as f:
        gt_questions = json.load(f)
    with open(gt_file_answers) as f:
        gt_answers = json.load(f)
    return [{"video_name": q["video_name"], "question": q["question"], "id": q["question_id"], "answer": gt_answers[i]["answer"]}
            for i, q in enumerate(gt_questions)]


def build_output(samples, preds):
    """Reference schema; samples without a prediction (missing video / failure) are left out like the reference's `except`."""
    return [{"id": s["id"], "question": s["question"], "answer": s["answer"], "pred": p} for s, p in zip(samples, preds) if p is not None]


def run_inference(args, components=None, load_frames=None):
    """components = (model, vision_tower, tokenizer, image_processor, video_token_len) may be injected (tests)."""
    from ..feature_extraction import load_video
    from . import _sharded

    rank, world, components = _sharded.setup(args, components)
    image_processor = components[3]
    frame_size = (image_processor.crop_size["height"], image_processor.crop_size["width"])
    load_frames = load_frames or (lambda path: load_video(path, shape=frame_size, device_resize=True))
    samples = load_samples(args.gt_file_question, args.gt_file_answers)
    os.makedirs(args.output_dir, exist_ok=True)
    tasks = [{"path": find_video(args.video_dir, s["video_name"]), "name": s["video_name"], "question": s["question"]} for s in samples]
    preds = _sharded.answer_tasks(args, tasks, components, load_frames, rank, world)
    return _sharded.write_output(args, build_output(samples, preds), rank)


if __name__ == "__main__":
    run_inference(parse_args())
