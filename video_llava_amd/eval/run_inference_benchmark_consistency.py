"""Video-ChatGPT consistency benchmark runner (video_chatgpt/eval/run_inference_benchmark_consistency.py:9-102): like the general
benchmark, but every sample asks `Q1` and `Q2` about the same clip and stores `pred1` / `pred2` (:50-51, :85-91); the clip's
features are computed once for the pair."""
from __future__ import annotations

from .run_inference_benchmark_general import parse_args, run_inference as _run_general


def run_inference(args, components=None, load_frames=None):
    return _run_general(args, components, load_frames, questions=("Q1", "Q2"), pred_keys=("pred1", "pred2"))


if __name__ == "__main__":
    run_inference(parse_args())
