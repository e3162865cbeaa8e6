"""Shared core of the eval runners: answer a list of (video, question) tasks data-parallel over the ranks of a
`torch.distributed.run` launch and hand every rank the same list of prediction strings.

The reference's runners (video_chatgpt/eval/run_inference_*.py) all share one loop -- load the video, call
`video_chatgpt_infer`, append the sample with its prediction, print and skip on any exception -- and differ only in how a
sample names its video file and which keys the output record carries.  Here that loop is one function: tasks are sharded over
the ranks (one process per GPU), each rank answers its shard in groups of `--batch` clips: ONE ViT pass over all frames of the group
(the launch shape bench.py measures), batched prefill + KV-cached greedy decode in libpgv, while a background thread decodes the next
group's frames into pinned memory; the only collective is the final all-gather of the answer token ids (video_llava_amd/parallel.py).
A task whose video is missing or fails yields None (the reference prints and leaves the sample out).
`do_sample=True` restores the reference's temperature-0.2 sampling through `video_chatgpt_infer`, one task at a time.
"""
from __future__ import annotations

import torch


MAX_BATCH = 16      # pgv_kv_create: the decode GEMVs put the batch on the 16 columns of one MFMA tile


def add_runtime_arguments(parser):
    """Flags this package adds to every runner on top of the reference's."""
    parser.add_argument("--batch", type=int, default=8, choices=range(1, MAX_BATCH + 1), metavar=f"[1-{MAX_BATCH}]",
                        help=f"clips answered together per GPU (greedy decoding; the decode kernels tile at most {MAX_BATCH} sequences)")
    parser.add_argument("--max_new_tokens", type=int, default=1024)
    parser.add_argument("--do_sample", action="store_true", help="reference decoding: temperature-0.2 sampling, one clip at a time")
    return parser


def setup(args, components=None):
    """-> (rank, world, components).  components = (model, vision_tower, tokenizer, image_processor, video_token_len) may be injected."""
    from .. import parallel
    rank, world, local = parallel.init_distributed()
    if components is None:
        from .model_utils import initialize_model
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        components = initialize_model(args.model_name, args.projection_path)
    if getattr(args, "use_asr", False):
        raise NotImplementedError("--use_asr needs the WhisperX transcript stack, which is outside this package's hot path")
    return rank, world, components


def answer_tasks(args, tasks, components, load_frames, rank, world):
    """tasks: [{"path": str | None, "name": str, "question": str}] -> [str | None] in task order, identical on every rank."""
    from .. import parallel
    from ..inference import build_prompt, video_chatgpt_infer, video_features, video_features_batch

    model, vision_tower, tokenizer, image_processor, video_token_len = components
    use_se = model.get_model().vision_config.use_vid_start_end
    stop_strs = {}

    from ..feature_extraction import PinnedRing
    ring = PinnedRing()

    def prepare(indices):
        """HOST half of a group, run one group ahead on the prefetch thread (parallel.run_sharded): decode / sample the frames of the group's
        distinct clips into pinned memory.  -> {path: frames | Exception}"""
        from ..feature_extraction import pin_frames
        ring.new_group()
        clips = {}
        for idx in indices:
            path = tasks[idx]["path"]
            if path is None or path in clips:
                continue
            try:
                clips[path] = pin_frames(load_frames(path), ring)
            except Exception as e:                                     # noqa: BLE001 -- reported per task below, like the reference's except
                clips[path] = e
        return clips

    def infer_batch(indices, clips=None):
        """-> (tokens [n, <= max_new] int32, lengths).  Length encoding: 0 = the task failed (no prediction), k + 1 = an answer of k tokens
        (k = 0 is a legitimate empty answer: the reference writes pred = '' when the first token is EOS).
        DEVICE half of a group: ONE tower pass over the frames of all distinct clips of the group (inference.video_features_batch; bitwise
        equal to per-clip passes), then batched prefill + decode."""
        if clips is None:
            clips = prepare(indices)
        good = [p for p, f in clips.items() if not isinstance(f, Exception)]
        feat_of = {}
        if good:
            try:
                feat_of = dict(zip(good, video_features_batch([clips[p] for p in good], vision_tower, image_processor)))
            except Exception as e:                                     # noqa: BLE001 -- one bad clip must not take the group down
                print(f"batched feature extraction failed ({e}); retrying the {len(good)} clips one by one")
                for p in good:
                    try:
                        feat_of[p] = video_features(clips[p], vision_tower, image_processor)
                    except Exception as e1:                            # noqa: BLE001
                        clips[p] = e1
        prompts, feats, keep = [], [], []
        for j, idx in enumerate(indices):
            t = tasks[idx]
            if t["path"] is None:
                print(f"Error processing video file '{t['name']}': not found")
                continue
            try:
                if t["path"] not in feat_of:
                    raise clips[t["path"]]
                prompt, stop = build_prompt(t["question"], args.conv_mode, video_token_len, use_se)
                ids = tokenizer([prompt]).input_ids[0]
                feats.append(feat_of[t["path"]])
                stop_strs[idx] = stop
                prompts.append(ids)
                keep.append(j)
            except Exception as e:                                     # noqa: BLE001 -- the reference's print-and-continue
                print(f"Error processing video file '{t['name']}': {e}")
        toks = torch.zeros(len(indices), args.max_new_tokens, dtype=torch.int32)
        lens = [0] * len(indices)
        eos = model.config.eos_token_id

        def record(j, row, n_prompt):
            new = row[n_prompt:].tolist()
            if eos is not None and eos in new:
                new = new[:new.index(eos)]
            new = new[:args.max_new_tokens]
            toks[j, :len(new)] = torch.tensor(new, dtype=torch.int32)
            lens[j] = len(new) + 1

        if keep:
            try:
                out = model.generate(prompts, video_spatio_temporal_features=torch.stack(feats), do_sample=False,
                                     max_new_tokens=args.max_new_tokens).cpu()
                for r, j in enumerate(keep):
                    record(j, out[r], len(prompts[r]))
            except Exception as e:                                     # noqa: BLE001
                # one bad sample (e.g. an over-long prompt) must not take the whole group down: the reference loses only that sample
                print(f"batched generation failed ({e}); retrying the {len(keep)} samples one by one")
                for r, j in enumerate(keep):
                    try:
                        out = model.generate([prompts[r]], video_spatio_temporal_features=feats[r][None], do_sample=False,
                                             max_new_tokens=args.max_new_tokens).cpu()
                        record(j, out[0], len(prompts[r]))
                    except Exception as e1:                            # noqa: BLE001
                        print(f"Error processing video file '{tasks[indices[j]]['name']}': {e1}")
        return toks, lens

    if args.do_sample:
        preds = [None] * len(tasks)
        for idx in parallel.shard_indices(len(tasks), rank, world):
            t = tasks[idx]
            try:
                if t["path"] is None:
                    raise FileNotFoundError(t["name"])
                preds[idx] = video_chatgpt_infer(load_frames(t["path"]), t["question"], args.conv_mode, model, vision_tower, tokenizer,
                                                 image_processor, video_token_len, None, max_new_tokens=args.max_new_tokens)
            except Exception as e:                                     # noqa: BLE001
                print(f"Error processing video file '{t['name']}': {e}")
        if world > 1:
            gathered = [None] * world
            torch.distributed.all_gather_object(gathered, preds)
            preds = [next((g[i] for g in gathered if g[i] is not None), None) for i in range(len(tasks))]
        return preds

    device = vision_tower.device if hasattr(vision_tower, "device") else torch.device("cpu")
    answers = parallel.run_sharded(len(tasks), infer_batch, args.max_new_tokens, rank, world, device, per_gpu_batch=args.batch,
                                   length_offset=1, prepare=prepare)
    from ..model.utils import first_stop_length
    preds = []
    for idx, ids in enumerate(answers):
        if ids is None:
            preds.append(None)                                         # failed / missing: left out of the output like the reference's except
            continue
        stop = stop_strs.get(idx) or build_prompt("", args.conv_mode, 1, use_se)[1]
        # The batched greedy loop runs to EOS / max_new_tokens; the reference's per-sample loop additionally stops when the conv mode's stop
        # string shows up in the decoded tail (KeywordsStoppingCriteria).  Cut where that loop would have stopped, then clean like it does.
        n = first_stop_length(ids, tokenizer, [stop]) if stop else None
        if n is not None:
            ids = ids[:n]
        text = tokenizer.batch_decode([ids], skip_special_tokens=True)[0].strip()
        preds.append(text.rstrip(stop).strip() if stop else text)     # the reference's own stop handling (inference.py:123)
    return preds


DECORD_FREE_FORMATS = (".npy", "")      # a uint8 [T,H,W,3] array, or a directory of frame images named like the video (no extension)


def first_existing(video_dir, stem, formats):
    """First existing `{stem}{ext}` in the given extension order; None when absent."""
    import os
    for fmt in formats:
        path = os.path.join(video_dir, f"{stem}{fmt}")
        if os.path.exists(path):
            return path
    return None


def write_output(args, output_list, rank):
    import json
    import os
    if rank == 0:
        with open(os.path.join(args.output_dir, f"{args.output_name}.json"), "w") as f:
            json.dump(output_list, f)
    return output_list
