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


MAX_BATCH = 64      # pgv_kv_create: the decode GEMVs tile the batch over up to 4 MFMA column tiles of 16 sequences (weights streamed once)


def add_runtime_arguments(parser):
    """Flags this package adds to every runner on top of the reference's."""
    parser.add_argument("--batch", type=_batch_arg, default="auto", metavar=f"[auto|1-{MAX_BATCH}]",
                        help=f"clips answered together per GPU (greedy decoding; the decode kernels tile at most {MAX_BATCH} sequences; the weight stream of a "
                             "token step is shared by the whole group, so larger groups raise videos/s: 8.9 / 13.4 / 16.2 / 18.9 at 8 / 16 / 32 / 64 on MI355X). "
                             "auto (default): the largest of 8 / 16 / 32 / 64 whose KV cache + tower workspace fit the GPU's free memory (pick_batch)")
    parser.add_argument("--max_new_tokens", type=int, default=1024)
    parser.add_argument("--do_sample", action="store_true", help="reference decoding: temperature-0.2 sampling, one clip at a time")
    parser.add_argument("--timings", default=None, metavar="OUT.jsonl",
                        help="append one JSON line per answered task (rank-local file `OUT.jsonl.rank<r>` when world > 1): frame load, upload + ingest, "
                             "tower + pool, prefill, decode seconds, generated tokens, group size, feature-cache hit (SURVEY 5: per-clip stage timings)")
    parser.add_argument("--feature-cache", type=int, default=FEATURE_CACHE_CLIPS, metavar="N",
                        help="pooled video features of the last N distinct clips stay on the device and are reused by later questions about the same "
                             "video (ActivityNet-QA asks several per clip; the reference recomputes them, chat.py:137-144); 0 disables")
    return parser


def _batch_arg(text):
    if text == "auto":
        return text
    n = int(text)
    if not 1 <= n <= MAX_BATCH:
        import argparse
        raise argparse.ArgumentTypeError(f"--batch {n} outside [1, {MAX_BATCH}]")
    return n


def batch_bytes(batch: int, model_config, image_size: int, max_new_tokens: int, frames_per_clip: int = 100, prompt_tokens: int | None = None) -> int:
    """Device bytes a group of `batch` clips needs beyond the weights: the decoder's KV cache + decode buffers for `batch` sequences of
    prompt + max_new_tokens positions, and the tower's per-pass workspace for batch x frames_per_clip frames (fp32 residual,
    16-bit operand / qkv / attention / MLP buffers of both lanes)."""
    c = model_config
    patches = (image_size // 14) ** 2
    prompt = prompt_tokens if prompt_tokens is not None else 100 + patches + 128
    seq = min((prompt + max_new_tokens + 63) // 64 * 64, int(getattr(c, "max_position_embeddings", 4096)))
    kv = 2 * c.num_hidden_layers * c.hidden_size * seq * 2 * batch
    logits = batch * (c.vocab_size + 64) * 4 * 3
    rows = batch * frames_per_clip * (patches + 1)
    vit = rows * (1024 * 4 + 1024 * 2 + 3072 * 2 + 1024 * 2 + 4096 * 2 + 16 * 8) + batch * frames_per_clip * image_size * image_size * 3 * (1 + 2)
    prefill = batch * prompt * (c.hidden_size * (4 + 2 + 6 + 2) + c.intermediate_size * 2)
    return int(kv + logits + max(vit, prefill))


def pick_batch(model_config, image_size: int, max_new_tokens: int, free_bytes: int, frames_per_clip: int = 100) -> int:
    """Largest group size in (64, 32, 16, 8) that fits 70 % of the free device memory (a real ActivityNet queue should run in
    the wide-batch regime the decode kernels support, not at a fixed 8); never below 8 -- smaller GPUs fail loudly in the allocation instead."""
    for b in (64, 32, 16):
        if batch_bytes(b, model_config, image_size, max_new_tokens, frames_per_clip) <= 0.7 * free_bytes:
            return b
    return 8


def setup(args, components=None):
    """-> (rank, world, components).  components = (model, vision_tower, tokenizer, image_processor, video_token_len) may be injected."""
    import os
    from .. import parallel
    if components is None and int(os.environ.get("WORLD_SIZE", "1")) > 1 and torch.cuda.is_available():
        # host placement FIRST: threads the process-group backend starts below inherit the mask
        parallel.pin_rank_to_numa_node(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local = parallel.init_distributed()
    if components is None:
        from .model_utils import initialize_model
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        components = initialize_model(args.model_name, args.projection_path)
    if getattr(args, "use_asr", False):
        raise NotImplementedError("--use_asr needs the WhisperX transcript stack, which is outside this package's hot path")
    return rank, world, components


FEATURE_CACHE_CLIPS = 64      # x 0.73 MB ([356, 1024] fp16) at 224 px, 1.4 MB at 336 px


class FeatureCache:
    """LRU of pooled video features keyed by video path.  The prefetch thread only READS (`get`, which also refreshes recency and hands out a
    reference, so an entry evicted before its group runs stays alive for that group); the device thread inserts."""

    def __init__(self, capacity: int):
        import collections
        import threading
        self.capacity, self._d, self._lock = max(int(capacity), 0), collections.OrderedDict(), threading.Lock()
        self.hits = self.misses = 0

    def get(self, path):
        with self._lock:
            f = self._d.get(path)
            if f is not None:
                self._d.move_to_end(path)
                self.hits += 1
            else:
                self.misses += 1
            return f

    def put(self, path, feat):
        if self.capacity <= 0:
            return
        with self._lock:
            self._d[path] = feat
            self._d.move_to_end(path)
            while len(self._d) > self.capacity:
                self._d.popitem(last=False)


class _Cached:
    """Marker the prefetch thread returns for a clip whose pooled features are in the cache (no frames were read)."""

    def __init__(self, feat):
        self.feat = feat


def answer_tasks(args, tasks, components, load_frames, rank, world):
    """tasks: [{"path": str | None, "name": str, "question": str}] -> [str | None] in task order, identical on every rank."""
    import json
    import time
    from .. import parallel
    from ..inference import build_prompt, video_chatgpt_infer_ids, video_features, video_features_batch

    model, vision_tower, tokenizer, image_processor, video_token_len = components
    use_se = model.get_model().vision_config.use_vid_start_end
    stop_strs = {}
    cache = FeatureCache(getattr(args, "feature_cache", FEATURE_CACHE_CLIPS))
    timings_path = getattr(args, "timings", None)
    if timings_path and world > 1:
        timings_path = f"{timings_path}.rank{rank}"
    tfile = open(timings_path, "a") if timings_path else None
    group_no = [0]

    from ..feature_extraction import PinnedRing
    ring = PinnedRing()

    def prepare(indices):
        """HOST half of a group, run one group ahead on the prefetch thread (parallel.run_sharded): decode / sample the frames of the group's
        distinct clips into pinned memory -- except clips whose pooled features are still cached from an earlier group.
        -> {path: frames | _Cached | Exception} (+ "__load_s__": {path: seconds})"""
        from ..feature_extraction import pin_frames
        ring.new_group()
        clips, load_s = {}, {}
        for idx in indices:
            path = tasks[idx]["path"]
            if path is None or path in clips:
                continue
            hit = cache.get(path)
            if hit is not None:
                clips[path] = _Cached(hit)
                continue
            t0 = time.perf_counter()
            try:
                clips[path] = pin_frames(load_frames(path), ring)
            except Exception as e:                                     # noqa: BLE001 -- reported per task below, like the reference's except
                clips[path] = e
            load_s[path] = time.perf_counter() - t0
        clips["__load_s__"] = load_s
        return clips

    def features_for(clips, tm):
        """{path: pooled features} for every usable clip of a group: cached ones as they are, the others through ONE tower pass."""
        feat_of = {p: f.feat for p, f in clips.items() if isinstance(f, _Cached)}
        good = []
        for p, f in clips.items():
            if isinstance(f, (Exception, _Cached)):
                continue
            hit = cache.get(p)               # the prefetch thread runs one group ahead: a clip it loaded may have been pooled by the group in between
            if hit is not None:
                feat_of[p] = hit
                clips[p] = _Cached(hit)
            else:
                good.append(p)
        if good:
            try:
                fresh = dict(zip(good, video_features_batch([clips[p] for p in good], vision_tower, image_processor, timings=tm)))
            except Exception as e:                                     # noqa: BLE001 -- one bad clip must not take the group down
                print(f"batched feature extraction failed ({e}); retrying the {len(good)} clips one by one")
                fresh = {}
                for p in good:
                    try:
                        fresh[p] = video_features(clips[p], vision_tower, image_processor)
                    except Exception as e1:                            # noqa: BLE001
                        clips[p] = e1
            for p, f in fresh.items():
                cache.put(p, f)
            feat_of.update(fresh)
        return feat_of

    def infer_batch(indices, clips=None):
        """-> (tokens [n, <= max_new] int32, lengths).  Length encoding: 0 = the task failed (no prediction), k + 1 = an answer of k tokens
        (k = 0 is a legitimate empty answer: the reference writes pred = '' when the first token is EOS).
        DEVICE half of a group: ONE tower pass over the frames of all distinct uncached clips of the group (inference.video_features_batch;
        bitwise equal to per-clip passes), then batched prefill + decode with the per-chunk stop-string check."""
        if clips is None:
            clips = prepare(indices)
        load_s = clips.pop("__load_s__", {})
        tm_v, tm_g = ({}, {}) if tfile else (None, None)
        feat_of = features_for(clips, tm_v)
        prompts, feats, keep, stops = [], [], [], []
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
                stops.append(stop)
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
            # a stop string that IS the EOS piece (pg-video-llava: "</s>") is already handled on the device by the EOS flag; the host check
            # is for conv modes whose stop is ordinary text (video-chatgpt_v1: "###") -- without it every group decodes max_new_tokens steps
            try:
                out = model.generate(prompts, video_spatio_temporal_features=torch.stack(feats), do_sample=False,
                                     max_new_tokens=args.max_new_tokens, stop_strings=stops, tokenizer=tokenizer, timings=tm_g).cpu()
                for r, j in enumerate(keep):
                    record(j, out[r], len(prompts[r]))
            except Exception as e:                                     # noqa: BLE001
                # one bad sample (e.g. an over-long prompt) must not take the whole group down: the reference loses only that sample.  Out of
                # memory (the KV cache of the group did not fit next to whatever else lives on the device: ADVICE r5) is not a sample's fault:
                # halve the group until it fits, one by one only for everything else
                oom = isinstance(e, (MemoryError, torch.cuda.OutOfMemoryError))
                size = max(1, len(keep) // 2) if oom and len(keep) > 1 else 1
                print(f"batched generation failed ({type(e).__name__}: {e}); retrying the {len(keep)} samples in groups of {size}")
                pending = [list(range(i, min(i + size, len(keep)))) for i in range(0, len(keep), size)]
                while pending:
                    rs = pending.pop(0)
                    try:
                        out = model.generate([prompts[r] for r in rs], video_spatio_temporal_features=torch.stack([feats[r] for r in rs]), do_sample=False,
                                             max_new_tokens=args.max_new_tokens, stop_strings=[stops[r] for r in rs], tokenizer=tokenizer).cpu()
                        for q, r in enumerate(rs):
                            record(keep[r], out[q], len(prompts[r]))
                    except Exception as e1:                            # noqa: BLE001
                        if len(rs) > 1:                                # still too big (or one bad sample inside): split again
                            pending[:0] = [rs[:len(rs) // 2], rs[len(rs) // 2:]]
                        else:
                            print(f"Error processing video file '{tasks[indices[keep[rs[0]]]]['name']}': {e1}")
        if tfile:
            n_fresh = sum(1 for f in clips.values() if not isinstance(f, (Exception, _Cached)))
            for j, idx in enumerate(indices):
                t = tasks[idx]
                hit = isinstance(clips.get(t["path"]), _Cached)
                tfile.write(json.dumps({"task": idx, "video": t["name"], "rank": rank, "group": group_no[0], "group_size": len(indices), "ok": lens[j] > 0,
                                        "tokens": max(lens[j] - 1, 0), "feature_cache_hit": hit, "load_s": load_s.get(t["path"], 0.0),
                                        "upload_ingest_s_group": (tm_v or {}).get("upload_ingest_s", 0.0), "tower_pool_s_group": (tm_v or {}).get("tower_pool_s", 0.0),
                                        "clips_in_tower_pass": n_fresh, "prefill_s_group": (tm_g or {}).get("prefill_s"), "decode_s_group": (tm_g or {}).get("decode_s"),
                                        "decode_steps_group": (tm_g or {}).get("steps")}) + "\n")
            tfile.flush()
        group_no[0] += 1
        return toks, lens

    def infer_sampled(indices, clips=None):
        """--do_sample: the reference's temperature-0.2 sampling with its stopping criterion, one task at a time (video_chatgpt_infer's
        path up to the detokenisation); the ids travel through the same fixed-shape collation as the greedy answers."""
        if clips is None:
            clips = prepare(indices)
        clips.pop("__load_s__", None)
        feat_of = features_for(clips, None)
        toks = torch.zeros(len(indices), args.max_new_tokens, dtype=torch.int32)
        lens = [0] * len(indices)
        for j, idx in enumerate(indices):
            t = tasks[idx]
            try:
                if t["path"] is None:
                    raise FileNotFoundError(t["name"])
                if t["path"] not in feat_of:
                    raise clips[t["path"]]
                new, stop = video_chatgpt_infer_ids(None, t["question"], args.conv_mode, model, vision_tower, tokenizer, image_processor, video_token_len,
                                                    None, max_new_tokens=args.max_new_tokens, features=feat_of[t["path"]])
                new = new[:args.max_new_tokens]
                stop_strs[idx] = stop
                toks[j, :len(new)] = torch.tensor(new, dtype=torch.int32)
                lens[j] = len(new) + 1
            except Exception as e:                                     # noqa: BLE001
                print(f"Error processing video file '{t['name']}': {e}")
        return toks, lens

    device = vision_tower.device if hasattr(vision_tower, "device") else torch.device("cpu")
    import os
    if getattr(args, "batch", "auto") == "auto":
        free_b = torch.cuda.mem_get_info(device)[0] if device.type == "cuda" else 0
        # processes that share this device see the same free pool and would each claim 70 % of it: PGV_DEVICE_SHARERS (or, in a
        # shared-device control-flow run, the world size) divides it
        sharers = int(os.environ.get("PGV_DEVICE_SHARERS", world if os.environ.get("PGV_BENCH_SHARE_DEVICE") else 1))
        args.batch = pick_batch(model.config, vision_tower.config.image_size, args.max_new_tokens, free_b // max(sharers, 1)) if free_b else 8
        if world > 1 and torch.distributed.is_available() and torch.distributed.is_initialized():
            # one group size for the whole job: the smallest any rank can afford (ranks with different free memory would otherwise run
            # different launch shapes -- same answers, since results do not depend on the batch, but skewed finish times)
            t = torch.tensor([args.batch], dtype=torch.int32, device=device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN)
            args.batch = int(t[0])
        if rank == 0:
            print(f"[pgv] --batch auto: {args.batch} clips per group ({free_b / 2 ** 30:.0f} GiB free on {device}, shared by {sharers})", flush=True)
    spill = os.path.join(args.output_dir, args.output_name) if getattr(args, "output_dir", None) and getattr(args, "output_name", None) else None
    if spill:
        os.makedirs(args.output_dir, exist_ok=True)
    try:
        if args.do_sample:
            answers = parallel.run_sharded(len(tasks), infer_sampled, args.max_new_tokens, rank, world, device, per_gpu_batch=1, length_offset=1, prepare=prepare,
                                           spill_path=spill)
        else:
            answers = parallel.run_sharded(len(tasks), infer_batch, args.max_new_tokens, rank, world, device, per_gpu_batch=args.batch,
                                           length_offset=1, prepare=prepare, spill_path=spill)
    finally:
        if tfile:
            tfile.close()
    from ..model.utils import first_stop_length
    preds = []
    for idx, ids in enumerate(answers):
        if ids is None:
            preds.append(None)                                         # failed / missing: left out of the output like the reference's except
            continue
        stop = stop_strs.get(idx) or build_prompt("", args.conv_mode, 1, use_se)[1]
        if args.do_sample:
            # the ids already end where the reference's criterion stopped them; its own post-processing (inference.py:119-123)
            text = tokenizer.batch_decode([ids], skip_special_tokens=True)[0]
            preds.append(text.strip().rstrip(stop).strip())
            continue
        # The batched greedy loop stops a sequence at most one chunk late and cuts it where the reference's per-sample loop would have stopped
        # (KeywordsStoppingCriteria); ids gathered from OTHER ranks went through the same cut.  Re-applying the cut here is idempotent and
        # keeps this function correct for a generate() without stop support; then clean like the reference does.
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
