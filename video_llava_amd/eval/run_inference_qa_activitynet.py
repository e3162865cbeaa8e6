"""ActivityNet-QA inference runner with the reference's CLI and output schema
(video_chatgpt/eval/run_inference_qa_activitynet.py:9-113): for every question find `v_{video_name}.{mp4,avi,mov,mkv}`,
answer it, and write `[{"id", "question", "answer", "pred"}, ...]` to `{output_dir}/{output_name}.json`.

The reference loops over the samples serially on one GPU (:63-104).  Here the samples are sharded over the ranks of a
`torch.distributed.run` launch (one process per GPU), each rank answers its shard in batches (one ViT pass per clip, batched
prefill + KV-cached greedy decode in libpgv), and the only collective is the final all-gather of the answer token ids
(video_llava_amd/parallel.py); rank 0 detokenises and writes the JSON.  A sample whose video is missing or fails keeps the
reference's policy: it is reported and left out of the output list (:103-104).
`--do_sample` restores the reference's temperature-0.2 sampling through `video_chatgpt_infer`, one sample at a time.
"""
from __future__ import annotations

import argparse
import json
import os

import torch

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
    parser.add_argument("--batch", type=int, default=8, help="clips answered together per GPU (greedy decoding)")
    parser.add_argument("--max_new_tokens", type=int, default=1024)
    parser.add_argument("--do_sample", action="store_true", help="reference decoding: temperature-0.2 sampling, one clip at a time")
    return parser.parse_args(argv)


def find_video(video_dir, video_name, extra_formats=(".npy",)):
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
    from .. import parallel
    from ..feature_extraction import load_video
    from ..inference import build_prompt, video_chatgpt_infer, video_features

    rank, world, local = parallel.init_distributed()
    if components is None:
        from .model_utils import initialize_model
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        components = initialize_model(args.model_name, args.projection_path)
    model, vision_tower, tokenizer, image_processor, video_token_len = components
    frame_size = (image_processor.crop_size["height"], image_processor.crop_size["width"])
    load_frames = load_frames or (lambda path: load_video(path, shape=frame_size))
    if args.use_asr:
        raise NotImplementedError("--use_asr needs the WhisperX transcript stack, which is outside this package's hot path")
    samples = load_samples(args.gt_file_question, args.gt_file_answers)
    os.makedirs(args.output_dir, exist_ok=True)
    use_se = model.get_model().vision_config.use_vid_start_end
    stop_strs = {}

    def infer_batch(indices):
        """-> (tokens [n, <= max_new] int32, lengths); a sample whose video is missing raises for the whole group only if every
        sample is missing, otherwise it gets length 0."""
        prompts, feats, keep = [], [], []
        for j, idx in enumerate(indices):
            s = samples[idx]
            path = find_video(args.video_dir, s["video_name"])
            if path is None:
                print(f"Error processing video file '{s['video_name']}': not found")
                continue
            try:
                frames = load_frames(path)
                feats.append(video_features(frames, vision_tower, image_processor))
                prompt, stop = build_prompt(s["question"], args.conv_mode, video_token_len, use_se)
                stop_strs[idx] = stop
                prompts.append(tokenizer([prompt]).input_ids[0])
                keep.append(j)
            except Exception as e:                                   # noqa: BLE001
                print(f"Error processing video file '{s['video_name']}': {e}")
        toks = torch.zeros(len(indices), args.max_new_tokens, dtype=torch.int32)
        lens = [0] * len(indices)
        if keep:
            out = model.generate(prompts, video_spatio_temporal_features=torch.stack(feats), do_sample=False,
                                 max_new_tokens=args.max_new_tokens).cpu()
            eos = model.config.eos_token_id
            for r, j in enumerate(keep):
                new = out[r, len(prompts[r]):].tolist()
                if eos is not None and eos in new:
                    new = new[:new.index(eos)]
                new = new[:args.max_new_tokens]
                toks[j, :len(new)] = torch.tensor(new, dtype=torch.int32)
                lens[j] = max(len(new), 1) if new else 0
        return toks, lens

    if args.do_sample:
        preds = [None] * len(samples)
        for idx in parallel.shard_indices(len(samples), rank, world):
            s = samples[idx]
            path = find_video(args.video_dir, s["video_name"])
            try:
                if path is None:
                    raise FileNotFoundError(s["video_name"])
                preds[idx] = video_chatgpt_infer(load_frames(path), s["question"], args.conv_mode, model, vision_tower, tokenizer,
                                                 image_processor, video_token_len, None, max_new_tokens=args.max_new_tokens)
            except Exception as e:                                   # noqa: BLE001
                print(f"Error processing video file '{s['video_name']}': {e}")
        if world > 1:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, preds)
            preds = [next((g[i] for g in gathered if g[i] is not None), None) for i in range(len(samples))]
    else:
        device = vision_tower.device if hasattr(vision_tower, "device") else torch.device("cpu")
        answers = parallel.run_sharded(len(samples), infer_batch, args.max_new_tokens, rank, world, device, per_gpu_batch=args.batch)
        preds = []
        for idx, ids in enumerate(answers):
            if not ids:
                preds.append(None)
                continue
            text = tokenizer.batch_decode([ids], skip_special_tokens=True)[0].strip()
            stop = stop_strs.get(idx) or build_prompt("", args.conv_mode, 1, use_se)[1]
            preds.append(text.rstrip(stop).strip() if stop else text)
    output_list = build_output(samples, preds)
    if rank == 0:
        with open(os.path.join(args.output_dir, f"{args.output_name}.json"), "w") as f:
            json.dump(output_list, f)
    return output_list


if __name__ == "__main__":
    run_inference(parse_args())
