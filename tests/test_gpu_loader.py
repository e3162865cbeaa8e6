"""GPU: the loader and the named entry points, executed for real on the synthetic checkpoint tree of
oracle/ckpt.py -- sharded safetensors + tokenizer files + a CLIP directory + an mm_projector.bin written with the trainer's key filter --
and checked against tests/golden/loader.npz, which the REFERENCE's own `initialize_model` + forward produced from the same tree
(oracle/gen_golden.py::gen_loader; margins of every greedy step > 0.19, asserted when the fixture was generated and again by
tests/test_oracle_golden.py).

Covers: `initialize_model` (AutoTokenizer, from_pretrained over shards, add_tokens, resize_token_embeddings 512 -> 515, pgv_llm_load_rows via
mm_projector.bin, unexpected-key reporting, CLIPVisionTower.from_pretrained, vision_config ids, video_token_len), `video_chatgpt_infer`
(greedy: tokens == the reference's; sampling: runs, reproducible from the generator seed), the `use_vid_start_end=False` splice branch and
its two ValueErrors, size-mismatch rejection, decode-graph invalidation after a vocabulary resize.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ckpt, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree(tmp_path_factory, golden_dir):
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    root = tmp_path_factory.mktemp("ckpt")
    info = ckpt.write_checkpoint_tree(str(root), synth.LLAMA_TINY, synth.CLIP_TINY, clip_seed=int(g["clip_seed"]), llm_seed=int(g["llm_seed"]),
                                      head_std=float(g["head_std"]))
    return info, g


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_initialize_model_and_infer_match_reference(ctx, tree, dtype, capsys):
    from PIL import Image
    from video_llava_amd.eval.model_utils import initialize_model
    from video_llava_amd.inference import video_chatgpt_infer
    info, g = tree
    lcfg, ccfg = synth.LLAMA_TINY, synth.CLIP_TINY
    model, vision_tower, tokenizer, image_processor, video_token_len = initialize_model(info["llm"], info["projector"], torch_dtype=dtype)
    out = capsys.readouterr().out
    assert "Loading weights from" in out and "Unexpected Keys" not in out
    # what eval/model_utils.py:114-150 sets up
    assert len(tokenizer) == lcfg.vocab and model.vocab_size == lcfg.vocab and model.config.vocab_size == lcfg.vocab
    assert video_token_len == 100 + ccfg.patches
    vc = model.get_model().vision_config
    assert (vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end) == (lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1, True)
    assert image_processor.crop_size["height"] == ccfg.image
    # the loaded tables: base rows from the shards, all 515 embedding rows from mm_projector.bin, lm_head's three new rows zero
    w = info["weights"]
    head = model.get_weight("lm_head.weight").cpu()
    assert torch.equal(head[:ckpt.BASE_VOCAB], torch.from_numpy(w["lm_head.weight"][:ckpt.BASE_VOCAB]).to(dtype).float())
    assert not head[ckpt.BASE_VOCAB:].any()
    k = "model.layers.1.mlp.down_proj.weight"
    assert torch.equal(model.get_weight(k).cpu(), torch.from_numpy(w[k]).to(dtype).float())

    frames = synth.make_frames(int(g["n_frames"]), ccfg.image, seed=int(g["frame_seed"]))
    question, conv_mode, n_new = str(g["question"]), str(g["conv_mode"]), int(g["n_new"])
    want_ids, want_text = g["tokens"].tolist(), str(g["text"])
    # 1) the reference's calling convention: a list of PIL images through the caller's CLIPImageProcessor
    pil = [Image.fromarray(f) for f in frames]
    text = video_chatgpt_infer(pil, question, conv_mode, model, vision_tower, tokenizer, image_processor, video_token_len, do_sample=False,
                               max_new_tokens=n_new)
    assert text == want_text.strip().rstrip(str(g["stop_str"])).strip()
    # 2) uint8 frames through the fused HIP preprocessing: the same token ids as the reference (token-exact in fp16 AND bf16: the
    #    fixture's smallest margin, 0.219, is > 10x the bf16 logit noise of this model)
    prompt_ids = torch.tensor([g["ids"].tolist()])
    from video_llava_amd.inference import video_features
    feats = video_features(frames, vision_tower, image_processor)
    ref_pooled = torch.from_numpy(g["pooled"]).float()
    rel = float((feats.float().cpu() - ref_pooled).norm() / ref_pooled.norm())
    assert rel < (2e-3 if dtype == torch.float16 else 1e-2), rel          # the fixture itself is the reference's fp16 CPU run
    out_ids = model.generate(prompt_ids, video_spatio_temporal_features=feats[None], do_sample=False, max_new_tokens=n_new)
    assert out_ids[0, :prompt_ids.shape[1]].tolist() == g["ids"].tolist()
    assert out_ids[0, prompt_ids.shape[1]:].tolist() == want_ids
    kv, nxt, logits = model.prefill([g["ids"].tolist()], feats[None], 640, want_logits=True)
    lr = torch.from_numpy(g["prefill_logits"]).float()
    rel = float((logits[0, :512].cpu() - lr).norm() / lr.norm())
    assert rel < (3e-3 if dtype == torch.float16 else 1.5e-2), rel
    assert float(logits[0, 512:].abs().max()) == 0.0                      # zero lm_head rows of the added tokens

    # 3) the reference's DEFAULT decode mode (do_sample=True, temperature=0.2): runs on the device, reproducible from the generator
    gen = torch.Generator(device="cuda").manual_seed(5)
    a = model.generate(prompt_ids, video_spatio_temporal_features=feats[None], do_sample=True, temperature=0.2, max_new_tokens=n_new, generator=gen)
    gen.manual_seed(5)
    b = model.generate(prompt_ids, video_spatio_temporal_features=feats[None], do_sample=True, temperature=0.2, max_new_tokens=n_new, generator=gen, chunk=3)
    assert torch.equal(a, b) and a.shape[1] <= prompt_ids.shape[1] + n_new
    assert isinstance(video_chatgpt_infer(frames, question, conv_mode, model, vision_tower, tokenizer, image_processor, video_token_len,
                                          max_new_tokens=4), str)        # defaults: do_sample=True, temperature=0.2

    # 4) use_vid_start_end = False (model/video_chatgpt.py:147-167): prompt without start/end, tokens == the reference's
    vc.use_vid_start_end = False
    try:
        ids2 = g["ids_nose"].tolist()
        out2 = model.generate(torch.tensor([ids2]), video_spatio_temporal_features=feats[None], do_sample=False, max_new_tokens=n_new)
        assert out2[0, len(ids2):].tolist() == g["tokens_nose"].tolist()
        from video_llava_amd.inference import build_prompt
        p2, _ = build_prompt(question, conv_mode, video_token_len, False)
        assert tokenizer([p2]).input_ids[0] == ids2
        p0 = ids2.index(lcfg.vocab - 3)
        bad = list(ids2); bad[p0 + 3] = 7
        with pytest.raises(ValueError, match="number of video patch tokens should be the same"):
            model.generate(torch.tensor([bad]), video_spatio_temporal_features=feats[None], max_new_tokens=1)
        bad = list(ids2); bad[p0 + 3], bad[p0 + video_token_len] = bad[p0 + video_token_len], bad[p0 + 3]
        with pytest.raises(ValueError, match="video patch tokens should be consecutive"):
            model.generate(torch.tensor([bad]), video_spatio_temporal_features=feats[None], max_new_tokens=1)
    finally:
        vc.use_vid_start_end = True


def test_loader_rejects_wrong_shapes_and_reports_unexpected_keys(ctx, tree, tmp_path, capsys):
    """torch semantics at the loader boundary: a tensor whose shape disagrees with config.json is a size-mismatch RuntimeError (never
    an over-read); keys the model does not have are reported like eval/model_utils.py:125-126 prints them."""
    from safetensors.torch import load_file
    from video_llava_amd import _lib
    from video_llava_amd.eval.model_utils import initialize_model
    from video_llava_amd.model.video_chatgpt import VideoChatGPTLlamaForCausalLM
    from video_llava_amd.vision_tower import CLIPVisionTower
    info, g = tree
    model = VideoChatGPTLlamaForCausalLM.from_pretrained(info["llm"], torch_dtype=torch.float16)
    sd = load_file(os.path.join(info["llm"], "model-00001-of-00002.safetensors"))
    k = "model.layers.0.mlp.down_proj.weight"
    with pytest.raises(RuntimeError, match="size mismatch for model.layers.0.mlp.down_proj.weight"):
        model.load_state_dict({k: sd[k].t().contiguous()}, strict=False)                  # transposed
    with pytest.raises(RuntimeError, match="size mismatch for model.embed_tokens.weight"):
        model.load_state_dict({"model.embed_tokens.weight": torch.zeros(600, 512)}, strict=False)
    # the C ABI re-checks the element count itself
    t = torch.zeros(10, dtype=torch.float16)
    rc = ctx.lib.pgv_llm_load_tensor(model.handle, k.encode(), t.data_ptr(), _lib.PGV_F16, 0, t.numel(), _lib.stream_ptr())
    assert rc == _lib.PGV_EINVAL and b"size mismatch" in ctx.lib.pgv_last_error()
    tower = CLIPVisionTower.from_pretrained(info["clip"], torch_dtype=torch.float16)
    with pytest.raises(RuntimeError, match="size mismatch for vision_model.encoder.layers.0.mlp.fc1.weight"):
        tower.load_state_dict({"vision_model.encoder.layers.0.mlp.fc1.weight": torch.zeros(4, 4)}, strict=False)
    # unexpected keys in the projector file
    proj = torch.load(info["projector"], map_location="cpu")
    proj["model.some_other_adapter.weight"] = torch.zeros(2, 2)
    bad = tmp_path / "mm_projector_extra.bin"
    torch.save(proj, bad)
    capsys.readouterr()
    initialize_model(info["llm"], str(bad))
    out = capsys.readouterr().out
    assert "Unexpected Keys: ['model.some_other_adapter.weight']" in out and "not loaded correctly" in out


def test_vocab_resize_after_generate_recaptures_decode_graph(ctx, tree):
    """A decode hipGraph bakes the vocabulary size into lm_head and the token pick.  generate -> resize_token_embeddings -> load the new
    rows -> generate must equal a model that was resized before its first generate (stale graph after resize)."""
    from safetensors.torch import load_file
    from video_llava_amd.model.video_chatgpt import VideoChatGPTLlamaForCausalLM
    info, g = tree
    lcfg = synth.LLAMA_TINY
    proj = torch.load(info["projector"], map_location="cpu")
    full_head = torch.from_numpy(info["weights"]["lm_head.weight"])

    def fresh():
        m = VideoChatGPTLlamaForCausalLM.from_pretrained(info["llm"], torch_dtype=torch.float16)
        vc = m.get_model().vision_config
        vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1, True
        return m

    def grow(m):
        m.resize_token_embeddings(lcfg.vocab)
        m.load_state_dict(proj, strict=False)
        m.load_state_dict({"lm_head.weight": full_head}, strict=False)        # non-zero rows for the new ids: a stale graph would not see them
    text = [1, 5, 9, 33, 2, 77]
    a = fresh()
    a.generate([text], max_new_tokens=20, eos_token_id=None)                    # captures the decode graph at vocabulary 512
    grow(a)
    ids = g["ids"].tolist()
    feats = torch.from_numpy(g["pooled"])[None].cuda()
    ka, na, la = a.prefill([ids], feats, 640, want_logits=True)
    toks_a = [int(na[0])] + a.decode_greedy(ka, na, 12)[0].tolist()
    na2, la2 = a.decode_step(ka, na, want_logits=True)
    b = fresh()
    grow(b)
    kb, nb, lb = b.prefill([ids], feats, 640, want_logits=True)
    toks_b = [int(nb[0])] + b.decode_greedy(kb, nb, 12)[0].tolist()
    nb2, lb2 = b.decode_step(kb, nb, want_logits=True)
    assert la.shape[1] == lcfg.vocab and torch.equal(la, lb) and toks_a == toks_b and torch.equal(la2, lb2)
    assert float(la2[0, 512:].abs().max()) > 0                                  # the grown rows are live


def test_bin_shards_with_projector_inside_and_projection_path_override(ctx, tmp_path):
    """the format `liuhaotian/llava-v1.5-7b/13b` ship in (reference docs/1-CLI_DEMO.md:27-29, loaded by
    video_chatgpt/eval/model_utils.py:104-105): `pytorch_model-0000{1,2}-of-00002.bin` + `pytorch_model.bin.index.json`, with the LLaVA-1.5
    projector `model.mm_projector.{0,2}.*` (mlp2x_gelu) INSIDE the base shards, and -- order of eval/model_utils.py:104-127 -- a
    `--projection_path` file loaded afterwards that overrides it.  The .bin tree must load through `initialize_model` token-equal to the
    safetensors tree holding the same tensors; with a projection file the file's projector must win; without one the shards' projector is
    live; a stray key of the upstream layout (`model.vision_tower...`) is tolerated like torch's strict=False."""
    import dataclasses
    import json
    import shutil
    from safetensors.torch import load_file
    from oracle import llm as ollm
    from oracle import vision as ovis
    from video_llava_amd.eval.model_utils import initialize_model
    from video_llava_amd.inference import build_prompt, video_features
    lcfg = dataclasses.replace(synth.LLAMA_TINY, projector="mlp2x_gelu")
    ccfg = synth.CLIP_TINY                                                         # 56 px: not 224 -> the projector comes from config.mm_projector_type
    st = ckpt.write_checkpoint_tree(str(tmp_path / "st"), lcfg, ccfg, clip_seed=71, llm_seed=72, head_std=0.08)
    w = st["weights"]
    proj_keys = sorted(k for k in w if k.startswith("model.mm_projector."))
    assert proj_keys == ["model.mm_projector.0.bias", "model.mm_projector.0.weight", "model.mm_projector.2.bias", "model.mm_projector.2.weight"]
    # a DIFFERENT projector for the base shards (what llava-v1.5 itself carries); the projection file holds the trained one (w)
    rng = np.random.default_rng(73)
    p_base = {k: torch.from_numpy((rng.standard_normal(w[k].shape) * 0.02).astype(np.float32)).half() for k in proj_keys}
    # ---- the .bin tree: same config / tokenizer files, shards re-saved with torch.save, projector + a stray upstream key inside shard 2 ----
    bin_dir = tmp_path / "bin" / "llm"
    shutil.copytree(st["llm"], bin_dir)
    weight_map = {}
    for i in (1, 2):
        sd = load_file(str(bin_dir / f"model-0000{i}-of-00002.safetensors"))
        if i == 2:
            sd.update(p_base)
            sd["model.vision_tower.vision_tower.vision_model.embeddings.class_embedding"] = torch.zeros(1024, dtype=torch.float16)
        name = f"pytorch_model-0000{i}-of-00002.bin"
        torch.save(sd, bin_dir / name)
        weight_map.update({k: name for k in sd})
        os.remove(bin_dir / f"model-0000{i}-of-00002.safetensors")
    os.remove(bin_dir / "model.safetensors.index.json")
    (bin_dir / "pytorch_model.bin.index.json").write_text(json.dumps({"metadata": {}, "weight_map": weight_map}))
    assert not [f for f in os.listdir(bin_dir) if f.endswith(".safetensors")]
    # projection files: the trainer's (projector + embedding table), and one with ONLY the base projector (to build the safetensors twin of case 1)
    only_base = tmp_path / "p_base.bin"
    torch.save(p_base, only_base)

    frames = synth.make_frames(7, ccfg.image, seed=74)
    V = 100 + ccfg.patches

    def run(llm_dir, projection):
        model, tower, tok, ip, vtl = initialize_model(str(llm_dir), str(projection) if projection else None)
        assert vtl == V and type(model.get_model().mm_projector).__name__ != "HipLinear"
        prompt, _ = build_prompt("what is shown?", "pg-video-llava", vtl, True)
        ids = tok([prompt]).input_ids[0]
        feats = video_features(frames, tower, ip)
        kv, nxt, lg = model.prefill([ids], feats[None], 1024, want_logits=True)
        out = model.generate(torch.tensor([ids]), video_spatio_temporal_features=feats[None], do_sample=False, max_new_tokens=10, eos_token_id=None)
        return ids, out[0, len(ids):].tolist(), lg[0].float().cpu(), model

    # case 1: no projection path -> the projector inside the .bin shards is live; twin = safetensors tree + a file with only that projector
    ids, t_bin, lg_bin, m_bin = run(bin_dir, None)
    assert torch.equal(m_bin.get_model().mm_projector.state_dict()["0.weight"].cpu().half(), p_base["model.mm_projector.0.weight"])
    _, t_st, lg_st, _ = run(st["llm"], only_base)
    assert t_bin == t_st and torch.equal(lg_bin, lg_st)
    # case 2: --projection_path after the shards -> the file's projector (and embedding table) override the shards'
    _, t_bin2, lg_bin2, m_bin2 = run(bin_dir, st["projector"])
    _, t_st2, lg_st2, _ = run(st["llm"], st["projector"])
    assert t_bin2 == t_st2 and torch.equal(lg_bin2, lg_st2)
    assert torch.equal(m_bin2.get_model().mm_projector.state_dict()["0.weight"].cpu().half(), torch.from_numpy(w["model.mm_projector.0.weight"]).half())
    assert not torch.equal(lg_bin2, lg_bin), "the projection file did not override the projector of the base shards"
    # and the overridden model computes what the fp32 oracle computes with the file's weights (lm_head rows of the 3 added tokens: zero-filled)
    wo = dict(w)
    head = wo["lm_head.weight"].copy(); head[ckpt.BASE_VOCAB:] = 0
    wo["lm_head.weight"] = head
    pooled = ovis.spatio_temporal_pool_torch(ovis.clip_select_features(ovis.clip_preprocess(frames), st["clip_weights"], ccfg)).float()
    with torch.no_grad():
        ref = ollm.LlamaOracle(wo, lcfg).prefill(ids, pooled, lcfg.vocab - 2, lcfg.vocab - 1, lcfg.vocab - 3)[0]
    rel = float((lg_bin2.double() - ref.double()).norm() / ref.double().norm())
    print(f".bin shards + projection override: prefill logits vs fp32 oracle {rel:.3e}")
    assert rel < 3e-3
