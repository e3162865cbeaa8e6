"""MSVD-QA runner (video_chatgpt/eval/run_inference_qa_msvd.py:11-94): like MSRVTT, but the video file of a sample is looked up
through the `--mapper` text file (`<youtube clip name> vid<N>` per line, :45-51) and has the `.avi` extension (:62)."""
from __future__ import annotations

from .run_inference_qa_msrvtt import parse_args as _parse_args, run_inference as _run_msrvtt


def parse_args(argv=None):
    return _parse_args(argv, mapper=True)


def load_mapper(path):
    """{video_id: clip name} from lines `<name> vid<id>`."""
    mappings = {}
    with open(path) as f:
        for entry in f.read().splitlines():
            if entry.strip():
                mappings[int(entry.split()[1].strip("vid"))] = entry.split()[0]
    return mappings


def eval_model(args, components=None, load_frames=None):
    mappings = load_mapper(args.mapper)
    return _run_msrvtt(args, components, load_frames, video_stem=lambda sample: mappings[sample["video_id"]], extensions=(".avi",))


run_inference = eval_model        # the reference names this entry point eval_model (:26); both names work here


if __name__ == "__main__":
    eval_model(parse_args())
