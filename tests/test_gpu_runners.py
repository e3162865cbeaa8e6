"""GPU end-to-end: the offline feature-extraction loop (reference scripts/save_spatio_temporal_clip_features.py) and the
ActivityNet-QA runner (reference video_chatgpt/eval/run_inference_qa_activitynet.py) on synthetic .npy clips, checked against
the CPU oracle: pooled features <= 1e-3 (fp16), answers token-exact for greedy decoding."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import llm as ollm
from oracle import synth
from oracle import vision as ovis

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


from helpers import SynthTokenizer as _Tok, make_tower as _tower  # noqa: E402


def test_extraction_loop_writes_reference_format(ctx, tmp_path):
    from video_llava_amd import feature_extraction as fx
    ccfg = synth.CLIP_TINY
    cw = synth.make_clip_weights(ccfg, seed=11)
    tower = _tower(ccfg, cw)
    vd, od = tmp_path / "videos", tmp_path / "feats"
    vd.mkdir()
    clips = {"vid_a": synth.make_frames(9, ccfg.image, seed=1), "vid_b": synth.make_frames(120, 20, seed=2)}     # b: 120 frames, needs sampling + resize
    for k, v in clips.items():
        np.save(vd / f"{k}.npy", v)
    args = fx.parse_args(["--llava", "1.1", "--video_dir_path", str(vd), "--clip_feat_path", str(od)])
    orig = dict(fx.LLAVA_VERSIONS)
    fx.LLAVA_VERSIONS["1.1"] = ("tiny", (ccfg.image, ccfg.image))
    try:
        assert fx.run(args, vision_tower=tower) == 2
        assert fx.run(args, vision_tower=tower) == 0                     # second run: everything already on disk
    finally:
        fx.LLAVA_VERSIONS.update(orig)
    P = (ccfg.image // ccfg.patch) ** 2
    for k in clips:
        got = pickle.load(open(od / f"{k}.pkl", "rb"))
        assert isinstance(got, np.ndarray) and got.dtype == np.float16 and got.shape == (100 + P, 1024)
        frames = fx.load_video(str(vd / f"{k}.npy"), shape=(ccfg.image, ccfg.image))
        ref = ovis.spatio_temporal_pool_numpy(ovis.clip_select_features(ovis.clip_preprocess(frames), cw, ccfg).numpy().astype(np.float16))
        err = np.linalg.norm(got.astype(np.float64) - ref.astype(np.float64)) / np.linalg.norm(ref.astype(np.float64))
        assert err < 1e-3, (k, err)
        T = frames.shape[0]
        assert not got[T:100].any()


def test_qa_runner_end_to_end_matches_oracle(ctx, tmp_path, capsys):
    from video_llava_amd.eval import run_inference_qa_activitynet as qa
    from video_llava_amd.inference import build_prompt
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    ccfg, lcfg = synth.CLIP_TINY, synth.LLAMA_TINY
    cw = synth.make_clip_weights(ccfg, seed=21)
    lw = synth.make_llama_weights(lcfg, seed=22, head_std=0.08)
    tower = _tower(ccfg, cw)
    model = VideoChatGPTLlamaForCausalLM(VideoChatGPTConfig(vocab_size=lcfg.vocab, hidden_size=lcfg.hidden, intermediate_size=lcfg.inter,
                                                            num_hidden_layers=lcfg.layers, num_attention_heads=lcfg.heads, eos_token_id=None,
                                                            max_position_embeddings=2048),
                                         VisionConfig(frame_size=ccfg.image), torch.float16)
    model.load_state_dict(lw)
    vc = model.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1, True
    tok = _Tok(lcfg.vocab)

    class IP:
        crop_size = {"height": ccfg.image, "width": ccfg.image}
    P = (ccfg.image // ccfg.patch) ** 2
    V = 100 + P
    vd = tmp_path / "v"; vd.mkdir()
    names = ["k1", "k2", "k3"]
    for i, n in enumerate(names):
        np.save(vd / f"v_{n}.npy", synth.make_frames(5 + i, ccfg.image, seed=30 + i))
    qs = [{"video_name": n, "question": f"what happens {i}?", "question_id": f"{n}_q"} for i, n in enumerate(names)]
    qs.insert(2, {"video_name": "missing", "question": "gone?", "question_id": "missing_q"})
    ans = [{"answer": f"a{i}"} for i in range(len(qs))]
    (tmp_path / "q.json").write_text(json.dumps(qs)); (tmp_path / "a.json").write_text(json.dumps(ans))
    NEW = 6
    args = qa.parse_args(["--video_dir", str(vd), "--gt_file_question", str(tmp_path / "q.json"), "--gt_file_answers", str(tmp_path / "a.json"),
                          "--output_dir", str(tmp_path / "out"), "--output_name", "preds", "--model-name", "x", "--projection_path", "y",
                          "--batch", "2", "--max_new_tokens", str(NEW)])
    out = qa.run_inference(args, components=(model, tower, tok, IP(), V))
    on_disk = json.load(open(tmp_path / "out" / "preds.json"))
    assert on_disk == out and [o["id"] for o in out] == ["k1_q", "k2_q", "k3_q"]          # the missing video is left out, order kept
    assert all(set(o) == {"id", "question", "answer", "pred"} for o in out)
    # oracle: same frames -> pooled features -> greedy tokens
    for o, n in zip(out, names):
        frames = np.load(vd / f"v_{n}.npy")
        pooled = ovis.spatio_temporal_pool_torch(ovis.clip_select_features(ovis.clip_preprocess(frames), cw, ccfg))
        prompt, _ = build_prompt(o["question"], "pg-video-llava", V, True)
        ids = tok([prompt]).input_ids[0]
        ref, margins = ollm.greedy_generate(lw, lcfg, ids, pooled.float(), lcfg.vocab - 2, lcfg.vocab - 1, lcfg.vocab - 3, NEW, return_margins=True)
        got = [int(t) for t in o["pred"].split()]
        assert min(margins) > 0.5, margins          # seeds (21, 22): every step's oracle margin is 0.9+ -- the comparison is never vacuous
        assert got == ref, (n, got, ref, margins)
    # the runners' DEFAULT group size (`--batch auto`, round 5): picked from the GPU's free memory (64 on an MI355X next to this tiny model), and --
    # batch invariance -- the same predictions as the explicit group size above
    capsys.readouterr()
    args2 = qa.parse_args(["--video_dir", str(vd), "--gt_file_question", str(tmp_path / "q.json"), "--gt_file_answers", str(tmp_path / "a.json"),
                           "--output_dir", str(tmp_path / "out2"), "--output_name", "preds", "--model-name", "x", "--projection_path", "y",
                           "--max_new_tokens", str(NEW)])
    assert args2.batch == "auto"
    out2 = qa.run_inference(args2, components=(model, tower, tok, IP(), V))
    assert "--batch auto: 64 clips per group" in capsys.readouterr().out and args2.batch == 64
    assert out2 == out


def test_consistency_runner_end_to_end_matches_oracle(ctx, tmp_path):
    """run_inference_benchmark_consistency: Q1 and Q2 on the same clip (features computed once per pair), schema {.., pred1, pred2}, a sample
    whose clip is missing is dropped; predictions checked against the oracle's greedy tokens."""
    from video_llava_amd.eval import run_inference_benchmark_consistency as cons
    from video_llava_amd.inference import build_prompt
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    ccfg, lcfg = synth.CLIP_TINY, synth.LLAMA_TINY
    cw = synth.make_clip_weights(ccfg, seed=41)
    lw = synth.make_llama_weights(lcfg, seed=39, head_std=0.08)      # searched: all four answers' margins > 0.35
    tower = _tower(ccfg, cw)
    model = VideoChatGPTLlamaForCausalLM(VideoChatGPTConfig(vocab_size=lcfg.vocab, hidden_size=lcfg.hidden, intermediate_size=lcfg.inter,
                                                            num_hidden_layers=lcfg.layers, num_attention_heads=lcfg.heads, eos_token_id=None,
                                                            max_position_embeddings=2048),
                                         VisionConfig(frame_size=ccfg.image), torch.float16)
    model.load_state_dict(lw)
    vc = model.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1, True
    tok = _Tok(lcfg.vocab)

    class IP:
        crop_size = {"height": ccfg.image, "width": ccfg.image}
    V = 100 + (ccfg.image // ccfg.patch) ** 2
    vd = tmp_path / "v"; vd.mkdir()
    for i, n in enumerate(["c1", "c2"]):
        np.save(vd / f"{n}.npy", synth.make_frames(6 + i, ccfg.image, seed=50 + i))
    gt = [{"video_name": "c1", "Q1": "what is shown?", "Q2": "describe the clip", "A": "x"},
          {"video_name": "nope", "Q1": "a?", "Q2": "b?", "A": "y"},
          {"video_name": "c2", "Q1": "who is there?", "Q2": "who appears?", "A": "z"}]
    (tmp_path / "gt.json").write_text(json.dumps(gt))
    NEW = 5
    args = cons.parse_args(["--video_dir", str(vd), "--gt_file", str(tmp_path / "gt.json"), "--output_dir", str(tmp_path / "out"), "--output_name", "c",
                            "--model-name", "x", "--projection_path", "y", "--batch", "3", "--max_new_tokens", str(NEW)])
    out = cons.run_inference(args, components=(model, tower, tok, IP(), V))
    assert json.load(open(tmp_path / "out" / "c.json")) == out and [o["video_name"] for o in out] == ["c1", "c2"]
    for o in out:
        frames = np.load(vd / f"{o['video_name']}.npy")
        pooled = ovis.spatio_temporal_pool_torch(ovis.clip_select_features(ovis.clip_preprocess(frames), cw, ccfg))
        for qk, pk in (("Q1", "pred1"), ("Q2", "pred2")):
            prompt, _ = build_prompt(o[qk], "pg-video-llava", V, True)
            ref, margins = ollm.greedy_generate(lw, lcfg, tok([prompt]).input_ids[0], pooled.float(), lcfg.vocab - 2, lcfg.vocab - 1, lcfg.vocab - 3, NEW,
                                                return_margins=True)
            got = [int(t) for t in o[pk].split()]
            assert min(margins) > 0.3, margins
            assert got == ref, (o["video_name"], qk, got, ref, margins)


def test_chat_interface_multi_turn_matches_oracle(ctx, tmp_path):
    """VideoChatGPTInterface (reference chat.py): FOUR turns on one clip.  The vision stage runs once per uploaded clip; every turn's
    prompt is the reference's (first turn re-rooted on the conv_mode template, <video> replaced by the placeholder run once) and the
    greedy answer equals the oracle's on the same prompt ids and pooled features.  From the second turn on only the tokens behind the
    prefix the KV cache already holds are prefilled (generate(kv_reuse_key=...) -> pgv_kv_truncate + pgv_llm_prefill_append; the reference
    re-runs the whole conversation, chat.py:108-160): `last_timings["reused_tokens"]` covers at least the whole previous prompt, and a twin
    interface with reuse_kv=False (full re-prefill every turn) gives the same answers."""
    from video_llava_amd.chat import VideoChatGPTInterface
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    from video_llava_amd.video_conversation import conv_templates
    ccfg, lcfg = synth.CLIP_TINY, synth.LLAMA_TINY
    cw = synth.make_clip_weights(ccfg, seed=41)
    lw = synth.make_llama_weights(lcfg, seed=44, head_std=0.08)      # searched: all four turns' margins > 1.1
    tower = _tower(ccfg, cw)
    calls = {"n": 0}
    orig_call = type(tower).__call__

    def counting_call(self, *a, **k):
        calls["n"] += 1
        return orig_call(self, *a, **k)
    type(tower).__call__ = counting_call
    try:
        model = VideoChatGPTLlamaForCausalLM(VideoChatGPTConfig(vocab_size=lcfg.vocab, hidden_size=lcfg.hidden, intermediate_size=lcfg.inter,
                                                                num_hidden_layers=lcfg.layers, num_attention_heads=lcfg.heads, eos_token_id=None,
                                                                max_position_embeddings=4096),
                                             VisionConfig(frame_size=ccfg.image), torch.float16)
        model.load_state_dict(lw)
        vc = model.get_model().vision_config
        vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1, True
        tok = _Tok(lcfg.vocab)

        class IP:
            crop_size = {"height": ccfg.image, "width": ccfg.image}
        P = (ccfg.image // ccfg.patch) ** 2
        V = 100 + P
        NEW = 5
        chat = VideoChatGPTInterface("x", "y", components=(model, tower, tok, IP(), V), max_output_tokens=NEW, do_sample=False)
        frames = synth.make_frames(7, ccfg.image, seed=9)
        np.save(tmp_path / "clip.npy", frames)
        chat.upload_video(str(tmp_path / "clip.npy"))
        pooled = ovis.spatio_temporal_pool_torch(ovis.clip_select_features(ovis.clip_preprocess(frames), cw, ccfg))
        replace = "<vid_start>" + "<vid_patch>" * V + "<vid_end>"
        conv = conv_templates["pg-video-llava"].copy()
        questions = ["what is in the video?", "and then?", "who is there?", "why?"]
        answers, prev_prompt = [], 0
        for turn, q in enumerate(questions):
            chat.add_text(q, str(tmp_path / "clip.npy"))
            got = chat.answer()
            conv.append_message(conv.roles[0], (q + "\n<video>") if turn == 0 else q)
            conv.append_message(conv.roles[1], None)
            prompt = conv.get_prompt().replace("<video>", replace, 1)
            ids = tok([prompt]).input_ids[0]
            ref, margins = ollm.greedy_generate(lw, lcfg, ids, pooled.float(), lcfg.vocab - 2, lcfg.vocab - 1, lcfg.vocab - 3, NEW, return_margins=True)
            assert min(margins) > 0.3, margins
            assert [int(t) for t in got.split()] == ref, (turn, got, ref, margins)
            reused = chat.last_timings["reused_tokens"]
            # turn 0 fills the cache; afterwards at least the previous turn's whole prompt (video run included) is taken from it -- the answer's
            # ids come back re-tokenised from text here, so the common prefix ends where the previous prompt ended
            assert (reused == 0) if turn == 0 else (prev_prompt <= reused < len(ids)), (turn, reused, prev_prompt, len(ids))
            prev_prompt = len(ids)
            conv.messages[-1][-1] = got
            answers.append(got)
        assert calls["n"] == 1, "the CLIP tower must run once per uploaded clip, not once per turn"
        # the same conversation with a full re-prefill every turn (the reference's behaviour): identical answers
        plain = VideoChatGPTInterface("x", "y", components=(model, tower, tok, IP(), V), max_output_tokens=NEW, do_sample=False, reuse_kv=False)
        plain.upload_video(frames)
        for turn, q in enumerate(questions):
            plain.add_text(q, None)
            assert plain.answer() == answers[turn] and plain.last_timings["reused_tokens"] == 0, turn
        # a second clip must not continue the first clip's cache
        chat.clear_history()
        chat.upload_video(synth.make_frames(7, ccfg.image, seed=10))
        chat.add_text(questions[0], None)
        chat.answer()
        assert chat.last_timings["reused_tokens"] == 0
        chat.clear_history()
        assert chat.video_features is None and chat.first_run
    finally:
        type(tower).__call__ = orig_call


def test_video_features_batch_equals_per_clip(ctx):
    """The runners' group path: ONE tower pass over the frames of several clips (ragged frame counts, mixed native resolutions, one clip
    already at crop size as a plain uint8 array) must give, clip by clip, bit-identical pooled features to `video_features` on the clip alone
    (the tower is batch-split invariant; the two-lane pass splits the concatenation at a frame that belongs to neither clip boundary)."""
    from video_llava_amd.feature_extraction import NativeFrames
    from video_llava_amd.inference import video_features, video_features_batch
    ccfg = synth.CLIP_TINY
    tower = _tower(ccfg, synth.make_clip_weights(ccfg, seed=51))
    S = ccfg.image
    rng = np.random.default_rng(52)
    clips = [NativeFrames(rng.integers(0, 256, (23, 40, 52, 3), dtype=np.uint8), (S, S)),
             NativeFrames(rng.integers(0, 256, (9, 40, 52, 3), dtype=np.uint8), (S, S)),       # same native size as the first: one upload, one ingest
             NativeFrames(rng.integers(0, 256, (31, 30, 30, 3), dtype=np.uint8), (S, S)),      # another native size (up-sampled)
             rng.integers(0, 256, (5, S, S, 3), dtype=np.uint8)]                               # already crop-sized: the preprocess-only path

    class IP:
        crop_size = {"height": S, "width": S}
    together = video_features_batch(clips, tower, IP())
    assert len(together) == 4
    for c, got in zip(clips, together):
        alone = video_features(c, tower, IP())
        assert got.shape == alone.shape == (100 + ccfg.patches, 1024) and torch.equal(got, alone)


# --------------------------------------------------------------------------------------------------
# round 4: runner details
# --------------------------------------------------------------------------------------------------
def _tiny_components(seed_c=21, seed_l=22, max_pos=2048):
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    ccfg, lcfg = synth.CLIP_TINY, synth.LLAMA_TINY
    cw = synth.make_clip_weights(ccfg, seed=seed_c)
    lw = synth.make_llama_weights(lcfg, seed=seed_l, head_std=0.08)
    tower = _tower(ccfg, cw)
    model = VideoChatGPTLlamaForCausalLM(VideoChatGPTConfig(vocab_size=lcfg.vocab, hidden_size=lcfg.hidden, intermediate_size=lcfg.inter,
                                                            num_hidden_layers=lcfg.layers, num_attention_heads=lcfg.heads, eos_token_id=None,
                                                            max_position_embeddings=max_pos),
                                         VisionConfig(frame_size=ccfg.image), torch.float16)
    model.load_state_dict(lw)
    vc = model.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1, True

    class IP:
        crop_size = {"height": ccfg.image, "width": ccfg.image}
    return ccfg, lcfg, cw, lw, tower, model, IP(), 100 + (ccfg.image // ccfg.patch) ** 2


class _StopTok(_Tok):
    """The synthetic tokenizer with chosen ids decoding to the text of a stop string ("###", the separator of conv mode `default`, is
    NOT a single-id keyword here, so only the decoded-tail branch of KeywordsStoppingCriteria can see it)."""

    def __init__(self, vocab, stop_ids):
        super().__init__(vocab)
        self.stop_ids = set(stop_ids)

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join("###" if int(t) in self.stop_ids else str(int(t)) for t in row) for row in ids]


def test_batched_runner_stops_on_a_text_stop_string(ctx, tmp_path):
    """conv mode `default` (conv_v1_2: stop string "###", ordinary text, not EOS) with the default-sized token budget: the batched greedy path must stop a group
    within ONE chunk of the last sequence's stop string instead of decoding max_new_tokens steps, and every prediction must equal the
    per-sample path (video_chatgpt_infer with the reference's KeywordsStoppingCriteria, reference inference.py:101-102, model/utils.py:6-26)."""
    from video_llava_amd.eval import run_inference_qa_activitynet as qa
    from video_llava_amd.inference import build_prompt, video_chatgpt_infer
    ccfg, lcfg, cw, lw, tower, model, ip, V = _tiny_components()
    vd = tmp_path / "v"; vd.mkdir()
    names = ["s1", "s2", "s3"]
    frames = {n: synth.make_frames(5 + i, ccfg.image, seed=60 + i) for i, n in enumerate(names)}
    for n in names:
        np.save(vd / f"v_{n}.npy", frames[n])
    qs = [{"video_name": n, "question": f"what happens {i}?", "question_id": f"{n}_q"} for i, n in enumerate(names)]
    (tmp_path / "q.json").write_text(json.dumps(qs)); (tmp_path / "a.json").write_text(json.dumps([{"answer": "a"}] * 3))
    # free-running greedy ids of every sample (plain tokenizer, no stop): choose as "stop ids" the token each sample emits at step 5 / 9 / 40
    plain = _Tok(lcfg.vocab)
    runs = {}
    for n, q in zip(names, qs):
        prompt, _ = build_prompt(q["question"], "default", V, True)
        ids = plain([prompt]).input_ids[0]
        pooled = ovis.spatio_temporal_pool_torch(ovis.clip_select_features(ovis.clip_preprocess(frames[n]), cw, ccfg))
        out = model.generate([ids], video_spatio_temporal_features=pooled.half().to(DEV)[None], do_sample=False, max_new_tokens=64)
        runs[n] = out[0, len(ids):].tolist()
    def fresh_token(seq, upto):        # the latest position <= upto whose token has not occurred before it: the stop can fire no earlier there
        return next(seq[i] for i in range(upto, 0, -1) if seq[i] not in seq[:i])
    stop_ids = {fresh_token(runs["s1"], 5), fresh_token(runs["s2"], 9), fresh_token(runs["s3"], 40)}
    tok = _StopTok(lcfg.vocab, stop_ids)
    # tokens each per-token loop keeps: the criterion's first call only records the start, so the earliest stop is at n = 2
    keep_n = {n: next(i for i, t in enumerate(runs[n]) if i >= 1 and t in stop_ids) + 1 for n in names}
    last_stop = max(keep_n.values())                                                                   # tokens until the LAST sequence has stopped
    assert last_stop <= 41
    steps = {"n": 0}
    orig = type(model).decode_greedy

    def counting(self, kv, first, n, eos_id=-1):
        steps["n"] += n
        return orig(self, kv, first, n, eos_id)
    type(model).decode_greedy = counting
    try:
        args = qa.parse_args(["--video_dir", str(vd), "--gt_file_question", str(tmp_path / "q.json"), "--gt_file_answers", str(tmp_path / "a.json"),
                              "--output_dir", str(tmp_path / "out"), "--output_name", "p", "--model-name", "x", "--projection_path", "y",
                              "--conv-mode", "default", "--batch", "3", "--max_new_tokens", "1024", "--timings", str(tmp_path / "t.jsonl")])
        out = qa.run_inference(args, components=(model, tower, tok, ip, V))
    finally:
        type(model).decode_greedy = orig
    assert [o["id"] for o in out] == ["s1_q", "s2_q", "s3_q"]
    assert steps["n"] + 1 <= last_stop + 32, f"{steps['n']} decode steps for a group whose last stop string ends at token {last_stop}"
    for o, n in zip(out, names):
        ref = video_chatgpt_infer(frames[n], o["question"], "default", model, tower, tok, ip, V, do_sample=False, max_new_tokens=1024)
        assert o["pred"] == ref and "###" not in o["pred"], (n, o["pred"], ref)
        assert len(o["pred"].split()) < 45
    # --timings: one JSON line per task with the stage times of its group
    lines = [json.loads(x) for x in open(tmp_path / "t.jsonl")]
    assert [x["task"] for x in lines] == [0, 1, 2] and all(x["ok"] and x["group_size"] == 3 and not x["feature_cache_hit"] for x in lines)
    assert all(x["tower_pool_s_group"] > 0 and x["prefill_s_group"] > 0 and x["decode_s_group"] > 0 and x["load_s"] > 0 for x in lines)
    assert [x["tokens"] for x in lines] == [keep_n[n] for n in names]


def test_runner_feature_cache_and_sampling_collation(ctx, tmp_path):
    """(a) Several questions about the same videos in different groups: the tower runs once per DISTINCT clip (LRU of pooled features across
    groups), and the predictions equal a run with the cache disabled.  (b) --do_sample goes through the same fixed-shape id collation as the
    greedy path and reproduces video_chatgpt_infer's text for the same generator state."""
    from video_llava_amd.eval import run_inference_qa_activitynet as qa
    from video_llava_amd.inference import video_chatgpt_infer
    ccfg, lcfg, cw, lw, tower, model, ip, V = _tiny_components()
    tok = _Tok(lcfg.vocab)
    vd = tmp_path / "v"; vd.mkdir()
    for i, n in enumerate(["a", "b"]):
        np.save(vd / f"v_{n}.npy", synth.make_frames(6 + i, ccfg.image, seed=70 + i))
    order = ["a", "b", "a", "b", "a", "a"]
    qs = [{"video_name": n, "question": f"question number {i}?", "question_id": f"q{i}"} for i, n in enumerate(order)]
    (tmp_path / "q.json").write_text(json.dumps(qs)); (tmp_path / "a.json").write_text(json.dumps([{"answer": "x"}] * len(qs)))
    base = ["--video_dir", str(vd), "--gt_file_question", str(tmp_path / "q.json"), "--gt_file_answers", str(tmp_path / "a.json"),
            "--output_dir", str(tmp_path / "out"), "--model-name", "x", "--projection_path", "y", "--batch", "2", "--max_new_tokens", "5"]
    frames_seen = {"n": 0, "calls": 0}
    orig_call = type(tower).__call__

    def counting_call(self, px, *a, **k):
        frames_seen["n"] += int(px.shape[0]); frames_seen["calls"] += 1
        return orig_call(self, px, *a, **k)
    type(tower).__call__ = counting_call
    try:
        with_cache = qa.run_inference(qa.parse_args(base + ["--output_name", "c1"]), components=(model, tower, tok, ip, V))
        assert frames_seen == {"n": 6 + 7, "calls": 1}, frames_seen           # group 0 holds both clips; groups 1 and 2 hit the cache
        frames_seen.update(n=0, calls=0)
        without = qa.run_inference(qa.parse_args(base + ["--output_name", "c0", "--feature-cache", "0"]), components=(model, tower, tok, ip, V))
        assert frames_seen == {"n": 3 * 13 - 7, "calls": 3}, frames_seen      # groups (a, b), (a, b), (a, a): one pass each over the distinct clips
    finally:
        type(tower).__call__ = orig_call
    assert with_cache == without and len(with_cache) == 6
    # (b) sampling
    one = [qs[1]]
    (tmp_path / "q1.json").write_text(json.dumps(one)); (tmp_path / "a1.json").write_text(json.dumps([{"answer": "x"}]))
    sargs = qa.parse_args(["--video_dir", str(vd), "--gt_file_question", str(tmp_path / "q1.json"), "--gt_file_answers", str(tmp_path / "a1.json"),
                           "--output_dir", str(tmp_path / "out"), "--output_name", "s", "--model-name", "x", "--projection_path", "y", "--max_new_tokens", "12",
                           "--do_sample"])
    torch.manual_seed(1234)
    got = qa.run_inference(sargs, components=(model, tower, tok, ip, V))
    torch.manual_seed(1234)
    ref = video_chatgpt_infer(np.load(vd / "v_b.npy"), one[0]["question"], "pg-video-llava", model, tower, tok, ip, V, max_new_tokens=12)
    assert len(got) == 1 and got[0]["pred"] == ref and len(ref.split()) == 12


def test_rccl_one_rank_process_group_collates_on_the_device(ctx, tmp_path):
    """the `nccl` (= RCCL) branch of parallel.init_distributed (`device_id=`) and the device-side
    `all_gather_into_tensor` of parallel.gather_answers had never executed anywhere (a gpurun box has one GPU, the CPU tests use gloo).
    A ONE-rank process group is legal: this runs both, for real, on the MI355X -- RCCL initialises, the packed int32 answers are gathered on
    the device and come back intact.  (What still has not run: the same call with 2+ ranks over xGMI.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "nccl1.py"
    script.write_text(f'''
import sys, json, torch
sys.path.insert(0, {root!r})
from video_llava_amd import parallel
rank, world, local = parallel.init_distributed(timeout_s=120)
assert (rank, world, local) == (0, 1, 0) and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
dev = torch.device("cuda", 0)
N, NEW = 5, 6
def infer(group):
    toks = torch.stack([torch.arange(NEW, dtype=torch.int32, device=dev) + 100 * i for i in group])
    return toks, [NEW - (i % 3) + 1 for i in group]
ans = parallel.run_sharded(N, infer, NEW, rank, world, dev, per_gpu_batch=2, length_offset=1)
t = torch.ones(4, device=dev); torch.distributed.all_reduce(t); torch.distributed.barrier()
ident = parallel.collective_identity(dev, rank, world)
ident["transport"] = parallel.rccl_transport()
print("IDENT", json.dumps(ident))
print("ANSWERS", json.dumps(ans), float(t.sum()))
torch.distributed.destroy_process_group()
''')
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", PGV_DIST_FORCE_INIT="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PGV_DIST_BACKEND", None)
    env.pop("NCCL_DEBUG_FILE", None)
    from video_llava_amd import parallel as _par
    _par.rccl_debug_env(env)                    # what bench.py sets before `import torch` for the ranks of an N > 1 run
    assert env["NCCL_DEBUG"] == "INFO" and "%p" in env["NCCL_DEBUG_FILE"]
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ANSWERS" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    line = r.stdout.split("ANSWERS", 1)[1].strip().splitlines()[0]                         # (RCCL prints its version banner to stdout at teardown)
    ans = json.loads(line[:line.rindex("]") + 1])
    assert ans == [[100 * i + t for t in range(6 - (i % 3))] for i in range(5)]
    assert line.endswith("4.0")
    # the identity record bench.py's `collective` carries for N > 1: library, version, device, and the captured RCCL INFO log
    ident = json.loads(r.stdout.split("IDENT", 1)[1].strip().splitlines()[0])
    assert ident["backend"] == "nccl" and ident["ranks_answered"] == 1 and ident["distinct_devices"] == 1
    assert ident["rccl_version"] and int(ident["rccl_version"].split(".")[0]) >= 2
    import re
    assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.0", ident["ranks"][0]["device_bdf"]), ident["ranks"]
    tr = ident["transport"]
    assert tr["captured"] and tr["log_lines"] > 0 and tr["verdict"] == "none", tr      # RCCL wrote its INFO log where we pointed it; one rank has no channels
