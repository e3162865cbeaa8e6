"""Measurement library behind bench.py and the GPU tests: the synthetic workload of BASELINE.json's configs as an object (`Workload`: random-init
or caller-supplied tower + decoder, frames resident in HBM, ActivityNet-QA-shaped prompts, `step` = the call bench.py times), the per-family
profiled pass with its rooflines, the guarded side lines, the runner-level measurement and the timing of the one collective.  bench.py keeps
the command line, the rank bring-up, the CPU baseline (the only part that may touch oracle/ or the reference's modules) and the JSON line;
tests import THIS module, not the script (tests/test_gpu_fulldepth.py drives `Workload.step` on seeded weights the CPU oracle also holds).
Nothing here imports oracle/.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_MFMA_TFLOPS = 2500.0     # dense bf16/fp16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E spec, MI355X_MICROARCH.md
VIT_FLOP_PER_FRAME_23L = {224: 155.29e9, 336: 366.0e9}     # 23 layers actually needed; 336 px: 577 tokens/frame


def make_prompts(n, vocab, video_rows, seed):
    """ActivityNet-QA-shaped prompts as token ids: ~70 template tokens + an 8-20 token question + <vid_start> + 356 x
    <vid_patch> + <vid_end> + role tag (no tokenizer files offline, so ids are synthetic)."""
    rng = np.random.default_rng(seed)
    PATCH, START, END = vocab - 3, vocab - 2, vocab - 1
    out = []
    for _ in range(n):
        q = int(rng.integers(8, 21))
        out.append([1] + rng.integers(3, vocab - 3, 70 + q).tolist() + [START] + [PATCH] * video_rows + [END]
                   + rng.integers(3, vocab - 3, 6).tolist())
    return out


class BenchTokenizer:
    """Word-level synthetic tokenizer with the four calls the path makes (no tokenizer files offline): one id per word (two for words longer
    than six characters, about what the LLaMA tokenizer yields on English prompts), the three video tokens at the top of the vocabulary."""

    def __init__(self, vocab):
        self.vocab = vocab
        self.special = {"<vid_patch>": vocab - 3, "<vid_start>": vocab - 2, "<vid_end>": vocab - 1}

    def _encode(self, text):
        import re
        ids = [1]
        for piece in re.findall(r"<vid_patch>|<vid_start>|<vid_end>|[^\s<]+|<", text):
            if piece in self.special:
                ids.append(self.special[piece])
                continue
            h = sum((i + 1) * ord(c) for i, c in enumerate(piece))
            ids.append(3 + h % (self.vocab - 8))
            if len(piece) > 6:
                ids.append(3 + (h * 31 + 7) % (self.vocab - 8))
        return ids

    def __call__(self, x):
        class R:
            pass
        r = R()
        r.input_ids = [self._encode(t) for t in x] if isinstance(x, (list, tuple)) else self._encode(x)
        return r

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(t)) for t in row) for row in ids]


QUESTION_WORDS = ("what is the person in the video doing while the dog runs across the yard and then jumps over a small fence near the house "
                  "before the man picks up the ball and throws it again towards the trees").split()


def runner_measurement(args, model, tower, video_rows, S, rank, world):
    """Runner-level throughput: what a user of video_chatgpt/eval/run_inference_qa_activitynet.py sees.  Synthetic clips are written to local
    disk as `v_<name>.npy` (uint8 [100, S, S, 3]), an ActivityNet-QA-shaped question / answer file pair is generated, and the package's runner
    (`run_inference`: file discovery, frame sampling on a prefetch thread, ONE tower pass per group, batched prefill + decode, stop-string
    handling, detokenisation, JSON dump) answers `--runner-groups` groups of `--clips-per-gpu` clips after one warm-up group."""
    import shutil
    import tempfile
    from video_llava_amd.eval import run_inference_qa_activitynet as qa
    B, T, NEW = args.clips_per_gpu, args.frames, args.new_tokens
    rng = np.random.default_rng(77)
    tmp = tempfile.mkdtemp(prefix="pgv_runner_")
    try:
        vd = os.path.join(tmp, "videos")
        os.makedirs(vd)
        n_files = 2 * B                                         # consecutive groups read different files
        for i in range(n_files):
            np.save(os.path.join(vd, f"v_clip{i:03d}.npy"), rng.integers(0, 256, (T, S, S, 3), dtype=np.uint8))

        class IP:
            crop_size = {"height": S, "width": S}
        tok = BenchTokenizer(model.vocab_size)
        components = (model, tower, tok, IP(), video_rows)

        def run(n_groups, tag):
            n = n_groups * B * world
            qs = [{"video_name": f"clip{(i % n_files):03d}", "question": " ".join(rng.choice(QUESTION_WORDS, int(rng.integers(6, 16)))) + "?",
                   "question_id": f"q{i}"} for i in range(n)]
            with open(os.path.join(tmp, f"q_{tag}.json"), "w") as f:
                json.dump(qs, f)
            with open(os.path.join(tmp, f"a_{tag}.json"), "w") as f:
                json.dump([{"answer": "yes"}] * n, f)
            a = qa.parse_args(["--video_dir", vd, "--gt_file_question", os.path.join(tmp, f"q_{tag}.json"), "--gt_file_answers",
                               os.path.join(tmp, f"a_{tag}.json"), "--output_dir", os.path.join(tmp, "out"), "--output_name", tag, "--model-name", "synthetic",
                               "--projection_path", "synthetic", "--batch", str(B), "--max_new_tokens", str(NEW),
                               "--feature-cache", "0"])            # the files repeat every other group: a cached tower pass would be skipped work
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = qa.run_inference(a, components=components)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert len(out) == n and all(len(o["pred"].split()) == NEW for o in out), "runner dropped samples or cut answers short"
            return n, dt
        run(1, "warm")
        n, dt = run(args.runner_groups, "timed")
        return {"videos_per_sec": n / dt, "seconds": dt, "clips": n, "groups": args.runner_groups, "clips_per_group": B,
                "entry": "video_llava_amd.eval.run_inference_qa_activitynet.run_inference", "frames_source": f"{n_files} .npy files on local disk (page cache)",
                "includes": "file discovery, frame sampling (prefetch thread, pinned), upload, one tower pass per group, prefill, decode, stop handling, "
                            "detokenisation, JSON dump"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# kernel-name prefixes of a family in the rocprofv3 tables; a name in PMC_SECONDARY belongs to a launch of the family that a primary kernel already
# counts (the finish launch of the 8-phase residual producers): its bytes are added, its calls are not
PMC_PREFIX = {"gemm": ("gemm_w4",), "vit_attn": ("vit_attn_kernel",), "llm_prefill_attn": ("prefill_attn_kernel",),
              "decode_gemv": ("gemv_mfma_kernel", "gemv_k8_kernel", "gemv_k8_finish_kernel"), "decode_attn": ("decode_attn_kernel", "decode_attn_split_kernel")}
PMC_SECONDARY = ("gemv_k8_finish_kernel",)


def pmc_traffic(family, tag="traffic"):
    """HBM-side traffic per launch of a kernel family from the committed rocprofv3 PMC passes (scripts/pmc_traffic.sh ->
    scripts/pmc_summary.py -> profiles/*pmc_traffic.json): FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE,
    call-weighted over the family's kernels.  rocprofv3 cannot wrap the process that is being timed, so the counters come from
    a separate run of the same bench command with 9 decode tokens; null when no profile file is present."""
    import glob
    # `tag`: "traffic" = the headline configuration (7B, 16-bit weights); "<llm>_<weights>" for the others (e.g. 13b_fp8), so a side line never
    # borrows the headline's counters
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{tag}.json")))
    if not files or family not in PMC_PREFIX:
        return {"traffic": None}
    d = json.load(open(files[-1]))["kernels"]
    calls = tot = 0.0
    for name, r in d.items():
        if name.startswith(PMC_PREFIX[family]) and r.get("fetch_bytes_corrected_x2") is not None:
            if not name.startswith(PMC_SECONDARY):
                calls += r["calls"]
            tot += r["calls"] * (r["fetch_bytes_corrected_x2"] + (r.get("write_bytes_raw") or 0.0))
    if calls == 0:
        return {"traffic": None}
    return {"traffic": tot / calls, "traffic_unit": "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, L2-miss side; Infinity-Cache hits are counted)",
            "traffic_source": os.path.relpath(files[-1], ROOT)}


def dry_pattern(idx, new_tokens, vocab):
    """Token pattern of clip `idx` in --dry mode (a function of the GLOBAL clip index, so the collation can be checked on every rank)."""
    return [(idx * 7919 + 31 * t + 5) % vocab for t in range(new_tokens)]


def time_collective(parallel, toks, lens, n_global, rank, world, sync, reps=5):
    """The one exchange step of the path, timed alone: `reps` all-gathers of the answer buffer bracketed by barriers (median, ms) and the
    number of ranks whose slots arrived filled."""
    import torch.distributed as dist
    ts, answers = [], None
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        answers = parallel.gather_answers(toks, lens, n_global, rank, world)
        ts.append((time.perf_counter() - t0) * 1e3)
    seen = sum(1 for r in range(world) if all(answers[i] is not None and len(answers[i]) > 0 for i in parallel.shard_indices(n_global, r, world)))
    return {"backend": dist.get_backend() if world > 1 else None, "world_seen": seen, "gather_ms": sorted(ts)[len(ts) // 2],
            "bytes_per_rank": int((toks.numel() + lens.numel()) * 4), "op": "all_gather_into_tensor"}, answers


class Workload:
    """One configuration of the hot path on this rank's GPU: random-init tower + decoder of the named shapes, synthetic frames resident in HBM,
    ActivityNet-QA-shaped prompts.  `a` carries dtype / llm / weights / image / workload / clips_per_gpu / frames / new_tokens.
    `tower` / `model`: prebuilt objects with weights already loaded (tests/test_gpu_fulldepth.py drives `step` -- the call this file times --
    on seeded weights the CPU oracle also holds); by default both are built here with random-init weights."""

    def __init__(self, a, dev, rank, world, tower=None, model=None):
        from video_llava_amd import _lib, parallel
        from video_llava_amd import random_init as ri
        from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
        from video_llava_amd.vision_tower import CLIPVisionTower, CLIPVisionTowerConfig
        self.a, self.dev, self.rank, self.world, self.parallel = a, dev, rank, world, parallel
        self.dtype = dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float16
        self.ctx = _lib.Context.get(dev)
        self.B, self.T, self.NEW, self.S = a.clips_per_gpu, a.frames, a.new_tokens, a.image
        S = self.S
        self.video_rows = 100 + (S // 14) ** 2                      # 356 at 224 px, 676 at 336 px
        self.projector = "linear" if S == 224 else "mlp2x_gelu"    # reference rule: model/video_chatgpt.py:52-55
        self.tower = tower
        if tower is None:
            self.tower = CLIPVisionTower(CLIPVisionTowerConfig(image_size=S), dtype, dev)
            ri.load_streaming(self.tower, ri.iter_clip_tensors(image=S, device=dev, dtype=dtype, seed=1))
        shapes = dict(hidden=4096, inter=11008, layers=32, heads=32) if a.llm == "7b" else dict(hidden=5120, inter=13824, layers=40, heads=40)
        self.vocab = vocab = 32003
        self.model, self.proj = model, None
        if a.workload == "full" and model is None:
            cfg = VideoChatGPTConfig(vocab_size=vocab, hidden_size=shapes["hidden"], intermediate_size=shapes["inter"],
                                     num_hidden_layers=shapes["layers"], num_attention_heads=shapes["heads"], eos_token_id=None,
                                     mm_projector_type=self.projector)
            self.model = VideoChatGPTLlamaForCausalLM(cfg, VisionConfig(frame_size=S), dtype, dev)
            ri.load_streaming(self.model, ri.iter_llama_tensors(vocab=vocab, hidden=shapes["hidden"], inter=shapes["inter"], layers=shapes["layers"],
                                                                projector=self.projector, device=dev, dtype=dtype, seed=2))
            vc = self.model.get_model().vision_config
            vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = vocab - 3, vocab - 2, vocab - 1, True
            if a.weights == "fp8":
                self.model.quantize_weights_fp8()
        elif a.workload != "full":
            from video_llava_amd.model.multimodal_projector.builder import HipLinear
            self.proj = HipLinear(1024, 4096, dtype, dev)
            self.proj.weight.data.normal_(0, 0.02); self.proj.bias.data.normal_(0, 0.02)
        # ---- synthetic inputs resident in HBM ----
        gen = torch.Generator(device=dev).manual_seed(100 + rank)
        self.frames = torch.randint(0, 256, (self.B * self.T, S, S, 3), dtype=torch.uint8, device=dev, generator=gen)
        self.n_global = self.B * world
        prompts_all = make_prompts(self.n_global, vocab, self.video_rows, seed=5)
        self.mine = parallel.shard_indices(self.n_global, rank, world)
        self.prompts = [prompts_all[i] for i in self.mine]
        self.vit_events = []

    def vision(self, frames_u8):
        from video_llava_amd.inference import get_spatio_temporal_features_torch
        B, T = self.B, self.T
        px = self.ctx.preprocess_u8(frames_u8, self.dtype)
        hid = self.tower(px, output_hidden_states=True).hidden_states[-2]
        return torch.stack([get_spatio_temporal_features_torch(hid[b * T:(b + 1) * T, 1:]) for b in range(B)])   # [B, 356, 1024] fp16

    def step(self, new_tokens, collate=True, host_frames=None):
        B, dev, prompts = self.B, self.dev, self.prompts
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                                   # torch's current stream IS the stream libpgv launches on
        pooled = self.vision(self.frames if host_frames is None else host_frames.to(dev, non_blocking=True))
        e1.record()
        self.vit_events.append((e0, e1))
        if self.a.workload == "vision":
            return self.proj(pooled.to(self.dtype))
        out = self.model.generate(prompts, video_spatio_temporal_features=pooled, do_sample=False, max_new_tokens=new_tokens,
                                  eos_token_id=None, chunk=64)
        toks = torch.stack([out[b, len(prompts[b]):len(prompts[b]) + new_tokens] for b in range(B)]).to(torch.int32)
        lens = torch.full((B,), new_tokens, dtype=torch.int32, device=dev)
        if not collate:                               # rank-0-only profiled pass: no collective (the other ranks are not in it)
            return toks
        return self.parallel.gather_answers(toks, lens, self.n_global, self.rank, self.world)

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if self.world > 1:
            tt = torch.tensor([seconds], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            return float(tt[0])
        return seconds

    def timed(self, steps, warmup, host_frames=None):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks.  Returns (seconds, per-rank
        per-step list of seconds is not kept), the ViT-stage ms per step from events on the launch stream."""
        for _ in range(warmup):
            self.step(self.NEW, host_frames=host_frames)
        self.barrier()
        self.vit_events.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(self.NEW, host_frames=host_frames)
        self.barrier()
        own = time.perf_counter() - t0
        elapsed = self.max_over_ranks(own)
        vit_ms = sum(x.elapsed_time(y) for x, y in self.vit_events) / max(len(self.vit_events), 1)      # frames -> pooled features, per step
        return elapsed, vit_ms, own

    def latency_b1(self):
        """BASELINE configs[2] as written: ONE synthetic 100-frame clip, batch 1, frames -> NEW greedy tokens (host-visible latency incl. the final
        D2H) -- and a SECOND chat turn on the same clip (its prompt = first prompt + the answer + a 24-token follow-up question): with the KV prefix
        kept (generate(kv_reuse_key=...) -> pgv_llm_prefill_append) against the reference's behaviour, a full re-prefill of the whole conversation
        (video_chatgpt/chat.py:108-160)."""
        from .inference import get_spatio_temporal_features_torch
        ctx, tower, model, frames, prompts, T, NEW, dev = self.ctx, self.tower, self.model, self.frames, self.prompts, self.T, self.NEW, self.dev

        def features():
            px = ctx.preprocess_u8(frames[:T], self.dtype)
            hid = tower(px, output_hidden_states=True).hidden_states[-2]
            return get_spatio_temporal_features_torch(hid[:, 1:])[None]

        def one_clip():
            return model.generate([prompts[0]], video_spatio_temporal_features=features(), do_sample=False, max_new_tokens=NEW, eos_token_id=None, chunk=64).cpu()
        one_clip()
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter(); one_clip(); ts.append(time.perf_counter() - t1)
        out = {"clips": 1, "frames": T, "new_tokens": NEW, "seconds_median": sorted(ts)[1], "seconds_min": min(ts), "videos_per_sec_batch1": 1.0 / sorted(ts)[1]}
        try:
            pooled = features()
            NEW2 = min(NEW, 64)
            follow = np.random.default_rng(9).integers(3, self.vocab - 3, 24).tolist()

            def turns(reuse):
                key = object() if reuse else None
                tm = {}
                o1 = model.generate([prompts[0]], video_spatio_temporal_features=pooled, do_sample=False, max_new_tokens=NEW2, eos_token_id=None, chunk=64, kv_reuse_key=key)
                p2 = o1[0].tolist() + follow
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                o2 = model.generate([p2], video_spatio_temporal_features=pooled, do_sample=False, max_new_tokens=NEW2, eos_token_id=None, chunk=64, kv_reuse_key=key, timings=tm).cpu()
                return time.perf_counter() - t1, tm, o2
            turns(True); turns(False)                                # warm both shapes (graphs, workspace)
            t_re, tm_re, o_re = turns(True)
            t_full, tm_full, o_full = turns(False)
            out["second_turn"] = {"new_tokens": NEW2, "prompt_tokens": int(o_re.shape[1] - NEW2), "reused_tokens": tm_re.get("reused_tokens"),
                                  "seconds_kv_prefix_kept": t_re, "prefill_s_kv_prefix_kept": tm_re.get("prefill_s"),
                                  "seconds_full_reprefill": t_full, "prefill_s_full_reprefill": tm_full.get("prefill_s"),
                                  # the kept cache entries of the first answer were written by DECODE steps, the re-prefill recomputes them on the
                                  # prefill path: same values up to the 16-bit rounding of either path, so a random-init bf16 model can flip a near-tie
                                  "leading_tokens_equal": int((o_re[0, -NEW2:] != o_full[0, -NEW2:]).int().cumsum(0).eq(0).sum())}
        except Exception as e:                                       # noqa: BLE001 -- a side field
            out["second_turn"] = {"error": f"{type(e).__name__}: {e}"}
        return out

    def token_check(self, tokens=32):
        """Calibration-free cross-check of the decode rooflines: `tokens` greedy steps replayed from the production hipGraphs between two events on
        the launch stream (no per-launch event pairs, nothing subtracted), against the ALGORITHMIC bytes a token step moves -- every decoder matrix
        and lm_head once (16-bit or e4m3 + scales) + the K and V entries of the mean context of the run for every sequence.  tb_s / 8 TB/s is the
        whole-token fraction of the HBM peak; the GEMV family's own fraction cannot exceed what this implies."""
        model, prompts, dev = self.model, self.prompts, self.dev
        cfg = model.config
        H, I, L, V = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, model.vocab_size
        wb = 1.0 if model.is_fp8 else 2.0
        n_w = L * (4.0 * H * H + 3.0 * H * I) + float((V + 15) // 16 * 16) * H
        weight_bytes = n_w * wb + (L * (4.0 * H + 2.0 * I + H) + V) * 4.0 * (1.0 if model.is_fp8 else 0.0)      # + per-row fp32 scales of the fp8 copies
        pooled = self.vision(self.frames)
        kv, nxt, _ = model.prefill(prompts, pooled, max(len(p) for p in prompts) + 3 * tokens + 16)
        model.decode_greedy(kv, nxt, 8)                              # eager warm-up, then graph capture
        model.decode_greedy(kv, nxt, tokens)
        torch.cuda.synchronize(dev)
        ctx0 = [model.ctx.lib.pgv_kv_len(kv, b) for b in range(len(prompts))]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); model.decode_greedy(kv, nxt, tokens); e1.record(); torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        mean_ctx = [c + (tokens + 1) / 2.0 for c in ctx0]           # step i of sequence b reads c + i + 1 cache positions (its own token included)
        kv_bytes = sum(2.0 * 2.0 * c * H * L for c in mean_ctx)      # K and V, 16-bit, all layers
        per_tok = weight_bytes + kv_bytes
        tb_s = per_tok / (ms / tokens * 1e-3) / 1e12
        return {"tokens": tokens, "sequences": len(prompts), "graph_replay_ms_per_token": ms / tokens, "weight_bytes_per_token": weight_bytes,
                "kv_bytes_per_token_at_mean_context": kv_bytes, "mean_context": sum(mean_ctx) / len(mean_ctx), "bytes_per_token": per_tok,
                "tb_s": tb_s, "frac": tb_s * 1e3 / PEAK_HBM_GBS,
                "note": "graph replay between two events, no per-launch instrumentation; bytes = every decoder matrix + lm_head once + K/V of the mean context"}

    def clip_tflops(self, vit_ms):
        return self.B * self.T * VIT_FLOP_PER_FRAME_23L[self.S] / (vit_ms * 1e-3) / 1e12

    def profile_pass(self, ms_per_step):
        """hipEvent pairs around every launch of each kernel family, on the launch stream -> (families, roofline of the dominant one)."""
        a, ctx, dev, NEW = self.a, self.ctx, self.dev, self.NEW
        prof_tokens = min(NEW, 9)
        ctx.prof_enable(True); ctx.prof_reset()
        self.step(prof_tokens, collate=False)
        torch.cuda.synchronize(dev)
        raw = ctx.prof_get()
        ctx.prof_enable(False)
        pair_ms = ctx.prof_calibrate(512)         # an empty hipEvent pair (reported for reference)
        # What the per-launch pairs add, measured on the decode kernel mix itself: the same CAL tokens decoded once with the profiler on (eager,
        # a pair around each of the 7 launches per layer) and once as production runs them (hipGraph replay, two events around all CAL tokens).
        # (sum of the pairs - graph time) / launches is removed from every per-launch average below, so the corrected decode families add up
        # to the decode time of the timed region.  An empty pair costs ~4.7 us here.
        ev_us, cal = 0.5 * pair_ms * 1e3, None
        if a.workload == "full":
            CAL = 8
            model, prompts = self.model, self.prompts
            pooled = self.vision(self.frames)
            kv, nxt, _ = model.prefill(prompts, pooled, max(len(p) for p in prompts) + 4 * CAL + 8)
            model.decode_greedy(kv, nxt, CAL)                                   # warm: first call is eager, second captures the graph
            model.decode_greedy(kv, nxt, CAL)
            torch.cuda.synchronize(dev)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); model.decode_greedy(kv, nxt, CAL); c1.record(); torch.cuda.synchronize(dev)
            graph_ms = c0.elapsed_time(c1)
            ctx.prof_enable(True); ctx.prof_reset()
            model.decode_greedy(kv, nxt, CAL); torch.cuda.synchronize(dev)
            pr = ctx.prof_get(); ctx.prof_enable(False)
            dec = [pr[k] for k in ("decode_gemv", "decode_attn", "decode_small")]
            n_l, paired_ms = sum(d["launches"] for d in dec), sum(d["ms"] for d in dec)
            ev_us = min(max((paired_ms - graph_ms) / n_l * 1e3, 0.0), pair_ms * 1e3)
            cal = {"tokens": CAL, "launches": n_l, "graph_replay_ms": graph_ms, "sum_of_pairs_ms": paired_ms, "empty_pair_us": pair_ms * 1e3}
            del kv
        decode_scale = (NEW - 1) / max(prof_tokens - 1, 1)
        DECODE = ("decode_gemv", "decode_attn", "decode_small")
        if cal:
            # The decode families are taken from the CALIBRATION run: its pairs were measured back to back with the graph replay of the same
            # tokens, so after removing ev_us per launch they add up to the graph time by construction.  (The pairs of the profiled STEP above
            # run right behind the ViT + prefill of that step and came out 5 % slower than both the graph replay and the rocprofv3 trace:
            # decode GEMV 19.2 us against 18.1 -- profiles/r04_b_*.)
            raw = dict(raw)
            for k in DECODE:
                raw[k] = pr[k]
            decode_scale = (NEW - 1) / cal["tokens"]
        fam = {}
        for name, r in raw.items():
            if r["launches"] == 0:
                continue
            net_ms = max(r["ms"] - ev_us * 1e-3 * r["launches"], 0.5 * r["ms"])
            avg_ms = net_ms / r["launches"]
            scale = decode_scale if name in DECODE else 1.0
            e = {"launches_per_step": int(round(r["launches"] * scale)), "avg_us": avg_ms * 1e3, "avg_us_raw": r["ms"] / r["launches"] * 1e3,
                 "ms_per_step_est": net_ms * scale}
            if r["flops"] > 0 and name in ("gemm", "vit_attn", "llm_prefill_attn"):
                e["tflops"] = r["flops"] / (net_ms * 1e-3) / 1e12
            if r["bytes"] > 0:
                e["gbs"] = r["bytes"] / (net_ms * 1e-3) / 1e9
            fam[name] = e
        dom = max((k for k in fam if k != "other"), key=lambda k: fam[k]["ms_per_step_est"])
        d = fam[dom]
        if dom in ("gemm", "vit_attn", "llm_prefill_attn"):
            roofline = {"kernel": dom, "bound": "mfma", "achieved": d["tflops"], "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": d["tflops"] / PEAK_MFMA_TFLOPS, "traffic": None}
        else:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": d["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": d["gbs"] / PEAK_HBM_GBS, "traffic": None}
        roofline["avg_launch_us"] = d["avg_us"]
        roofline["event_overhead_removed_us"] = ev_us
        roofline["event_calibration"] = cal
        roofline["share_of_step"] = d["ms_per_step_est"] / ms_per_step
        roofline["algorithmic_bytes_per_launch"] = raw[dom]["bytes"] / raw[dom]["launches"] if raw[dom]["bytes"] > 0 else None
        tag = "traffic" if (a.llm == "7b" and a.weights == "16bit" and a.image == 224) else (f"{a.llm}_{a.weights}" if a.image == 224 else f"image{a.image}")
        if a.clips_per_gpu != 8:                      # wide batches launch other kernel shapes: their own PMC file or no traffic figure at all
            tag = f"clips{a.clips_per_gpu}" if tag == "traffic" else f"{tag}_clips{a.clips_per_gpu}"
        roofline.update(pmc_traffic(dom, tag))
        if a.workload == "full":
            try:
                roofline["token_check"] = self.token_check()
            except Exception as e:                                   # noqa: BLE001 -- a cross-check never costs the line
                roofline["token_check"] = {"error": f"{type(e).__name__}: {e}"}
        # The MFMA-bound family the north star's 40 % target is about (every nn.Linear of the CLIP tower, the projector and the prefill on the
        # persistent GEMM) gets its own record next to the dominant (HBM-bound) one: same live event pairs, same PMC cross-check.
        self.roofline_mfma = None
        if "gemm" in fam and dom != "gemm":
            g = fam["gemm"]
            self.roofline_mfma = {"kernel": "gemm", "bound": "mfma", "achieved": g["tflops"], "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": g["tflops"] / PEAK_MFMA_TFLOPS, "avg_launch_us": g["avg_us"], "launches_per_step": g["launches_per_step"],
                                  "share_of_step": g["ms_per_step_est"] / ms_per_step,
                                  "algorithmic_flops_per_launch": raw["gemm"]["flops"] / raw["gemm"]["launches"],
                                  "algorithmic_bytes_per_launch": raw["gemm"]["bytes"] / raw["gemm"]["launches"]}
            self.roofline_mfma.update(pmc_traffic("gemm", tag))
        # ... and the HBM-bound decode GEMV family when it is NOT the dominant one (wide batches: vision + prefill dominate the step): the record the
        # wide-batch work of round 6 is judged by
        self.roofline_gemv = None
        if "decode_gemv" in fam and dom != "decode_gemv":
            g = fam["decode_gemv"]
            self.roofline_gemv = {"kernel": "decode_gemv", "bound": "hbm", "achieved": g["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": g["gbs"] / PEAK_HBM_GBS,
                                  "avg_launch_us": g["avg_us"], "launches_per_step": g["launches_per_step"], "share_of_step": g["ms_per_step_est"] / ms_per_step,
                                  "algorithmic_bytes_per_launch": raw["decode_gemv"]["bytes"] / raw["decode_gemv"]["launches"]}
            self.roofline_gemv.update(pmc_traffic("decode_gemv", tag))
        return fam, roofline

    def free(self):
        self.model = self.tower = self.proj = self.frames = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()


def side_line(args, dev, overrides, steps, warmup, with_roofline=False):
    """A guarded side measurement of the default run: the same timed loop on another configuration (own models, freed afterwards)."""
    import copy
    a = copy.copy(args)
    for k, v in overrides.items():
        setattr(a, k, v)
    t0 = time.perf_counter()
    w = Workload(a, dev, 0, 1)
    try:
        t_build = time.perf_counter() - t0
        elapsed, vit_ms, _ = w.timed(steps, warmup)
        ms = elapsed / steps * 1e3
        out = {"value": w.n_global * steps / elapsed, "unit": "videos/sec", "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": a.dtype, "llm": a.llm,
               "llm_weights": a.weights, "image": a.image, "clips_per_gpu_per_step": a.clips_per_gpu, "new_tokens": a.new_tokens,
               "clip_feat_ms_per_step": vit_ms, "clip_feat_tflops": w.clip_tflops(vit_ms), "clip_feat_frac": w.clip_tflops(vit_ms) / PEAK_MFMA_TFLOPS, "model_build_s": t_build}
        if with_roofline:
            fam, roof = w.profile_pass(ms)
            out["roofline"] = roof
            if w.roofline_mfma:
                out["roofline_mfma"] = w.roofline_mfma
            if getattr(w, "roofline_gemv", None):
                out["roofline_gemv"] = w.roofline_gemv
            out["family_avg_us"] = {k: v["avg_us"] for k, v in fam.items()}
            out["family_tflops"] = {k: v["tflops"] for k, v in fam.items() if "tflops" in v}
        return out
    finally:
        w.free()
