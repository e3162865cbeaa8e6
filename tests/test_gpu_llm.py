"""GPU parity: HIP decoder path (splice, RMSNorm, RoPE, KV-cached attention, SwiGLU MLP, lm_head, greedy) vs the
golden fixtures generated from the reference's own forward and vs the CPU oracle.

Bars: greedy decode token-exact (fp16; the fixtures' minimum top-1/top-2 margin is asserted > 0.02 at generation
time); logits within 1e-3 normwise in fp16, 8e-3 in bf16.
"""
import os

import numpy as np
import pytest
import torch

from oracle import llm as ollm
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b) -> float:
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


from helpers import make_model as _model  # noqa: E402


@pytest.mark.parametrize("tag,proj,image", [("lin", "linear", 224), ("mlp", "mlp2x_gelu", 336)])
def test_forward_is_callable_like_the_reference(ctx, golden_dir, tag, proj, image):
    """VERDICT r4 missing #5: `model(input_ids=..., video_spatio_temporal_features=..., use_cache=True)` -- the reference's public forward
    (video_chatgpt/model/video_chatgpt.py:193-251), the way SURVEY.md 8c drives it as an oracle: prefill returns the logits of ALL positions
    + past_key_values; single-token calls with that cache continue.  Checked against the fixture the REFERENCE's forward produced
    (`*_prefill_logits` [S, vocab], `*_step_logits`, `*_tokens`)."""
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    cfg = synth.LlamaCfg(**{**synth.LLAMA_TINY.__dict__, "projector": proj})
    w = synth.make_llama_weights(cfg, seed=int(g[tag + "_weight_seed"]), head_std=float(g["head_std"]))
    m = _model(cfg, w, torch.float16, image)
    ids = g[tag + "_ids"].tolist()
    feats = torch.from_numpy(g[tag + "_feats"]).half()[None]
    toks_ref = g[tag + "_tokens"].tolist()
    out = m(input_ids=torch.tensor([ids]), video_spatio_temporal_features=feats, use_cache=True)
    ref_all = torch.from_numpy(g[tag + "_prefill_logits"])
    assert tuple(out.logits.shape) == (1, len(ids), cfg.vocab) and out.logits.dtype == torch.float32
    assert ref_all.shape[0] == len(ids)
    assert rel(out.logits[0], ref_all) < 1e-3
    per_pos = [rel(out.logits[0, i], ref_all[i]) for i in range(len(ids))]
    assert max(per_pos) < 2e-3, (int(np.argmax(per_pos)), max(per_pos))
    assert out.past_key_values and out.past_key_values.get_seq_length() == len(ids) and out[0] is out.logits
    tok = out.logits[:, -1].argmax(-1, keepdim=True)
    got = [int(tok[0, 0])]
    for i in range(1, len(toks_ref)):
        out = m(input_ids=tok, past_key_values=out.past_key_values, video_spatio_temporal_features=feats, use_cache=True)
        assert tuple(out.logits.shape) == (1, 1, cfg.vocab)
        assert rel(out.logits[0, 0], g[tag + "_step_logits"][i]) < 1e-3, i
        tok = out.logits[:, -1].argmax(-1, keepdim=True)
        got.append(int(tok[0, 0]))
    assert got == toks_ref and out.past_key_values.get_seq_length() == len(ids) + len(toks_ref) - 1
    # a two-sequence batch of equal-length prompts equals the single runs position by position (text-only rows too)
    out2 = m(input_ids=torch.tensor([ids, ids]), video_spatio_temporal_features=torch.cat([feats, feats]))
    assert tuple(out2.logits.shape) == (2, len(ids), cfg.vocab) and torch.equal(out2.logits[0], out2.logits[1])
    assert rel(out2.logits[0], ref_all) < 1e-3
    # what has no counterpart in the eval path fails loudly
    with pytest.raises(NotImplementedError):
        m(input_ids=torch.tensor([ids]), labels=torch.tensor([ids]))
    with pytest.raises(ValueError, match="sequences for a cache"):
        m(input_ids=torch.tensor([ids[:3]]), past_key_values=out2.past_key_values)
    with pytest.raises(RuntimeError, match="stale"):          # `out`'s cache was refilled by the batch-2 prefill above
        m(input_ids=tok, past_key_values=out.past_key_values)
    with pytest.raises(RuntimeError, match="stale"):
        stale = out2.past_key_values
        m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats, max_new_tokens=2, eos_token_id=None)
        m(input_ids=torch.tensor([[5], [6]]), past_key_values=stale)


def test_forward_stale_cache_of_the_same_shape_is_refused(ctx):
    """one KV cache is kept alive per model and a second prefill of the SAME (batch, capacity) refills it in place -- the first
    call's past_key_values then describes a cache that no longer holds its tokens.  The handle is the same object, so only the epoch tells."""
    cfg = synth.LLAMA_TINY
    m = _model(cfg, synth.make_llama_weights(cfg, seed=3, head_std=0.08), torch.float16)
    a = m(input_ids=torch.tensor([[1, 7, 8, 9]]))
    b = m(input_ids=torch.tensor([[1, 20, 21, 22]]))
    assert a.past_key_values.kv is b.past_key_values.kv            # same shape -> same cache, refilled
    with pytest.raises(RuntimeError, match="stale"):
        m(input_ids=torch.tensor([[5]]), past_key_values=a.past_key_values)
    nxt = m(input_ids=torch.tensor([[5]]), past_key_values=b.past_key_values)        # the live one still steps
    ref = m(input_ids=torch.tensor([[1, 20, 21, 22, 5]]))
    assert torch.equal(nxt.logits[0, 0], ref.logits[0, -1]) or rel(nxt.logits[0, 0], ref.logits[0, -1]) < 1e-3
    # capacity follows the prompt (+ 256 positions, rounded to 64), not the whole context window; max_length overrides
    assert ref.past_key_values.max_seq == 320 and m(input_ids=torch.tensor([[1, 2]]), max_length=1000).past_key_values.max_seq == 1024
    small = m(input_ids=torch.tensor([[1, 2, 3]]), max_length=4)
    m(input_ids=torch.tensor([[4]]), past_key_values=small.past_key_values)
    assert small.past_key_values.max_seq == 64                     # rounded up to 64 positions
    with pytest.raises(ValueError, match="exceed the cache"):
        m(input_ids=torch.tensor([list(range(3, 3 + 61))]), past_key_values=small.past_key_values)


@pytest.mark.parametrize("tag,proj,image", [("lin", "linear", 224), ("mlp", "mlp2x_gelu", 336)])
def test_forward_with_cache_appends_many_tokens_bitwise(ctx, golden_dir, tag, proj, image):
    """VERDICT r5 missing #4: `model(input_ids [B, S > 1], past_key_values=...)` (video_chatgpt/model/video_chatgpt.py:193-251 accepts any input_ids
    next to a cache; a later chat turn) -> pgv_llm_prefill_append.  The prompt of the reference-generated fixture cut into 2 and 3 calls at
    several points -- before the video run (the splice then happens inside the APPEND call), inside the text behind it, one token before the
    end -- gives, row for row, BITWISE the logits of the one-call forward (which is checked against the reference's own output), and the
    greedy continuation reproduces the fixture's tokens."""
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    cfg = synth.LlamaCfg(**{**synth.LLAMA_TINY.__dict__, "projector": proj})
    w = synth.make_llama_weights(cfg, seed=int(g[tag + "_weight_seed"]), head_std=float(g["head_std"]))
    m = _model(cfg, w, torch.float16, image)
    ids = g[tag + "_ids"].tolist()
    S = len(ids)
    feats = torch.from_numpy(g[tag + "_feats"]).half()[None]
    toks_ref = g[tag + "_tokens"].tolist()
    full = m(input_ids=torch.tensor([ids]), video_spatio_temporal_features=feats).logits[0].clone()
    assert rel(full, torch.from_numpy(g[tag + "_prefill_logits"])) < 1e-3
    start = ids.index(cfg.vocab - 2)                                # <vid_start>
    end = ids.index(cfg.vocab - 1)                                  # <vid_end>
    for cuts in ((start,), (2,), (end + 1,), (end + 3,), (S - 1,), (2, end + 2), (start, end + 1, S - 2)):
        bounds = [0, *cuts, S]
        out, rows = None, []
        for a, b in zip(bounds[:-1], bounds[1:]):
            out = m(input_ids=torch.tensor([ids[a:b]]), video_spatio_temporal_features=feats, past_key_values=None if out is None else out.past_key_values,
                    max_length=S + 16)
            assert tuple(out.logits.shape) == (1, b - a, cfg.vocab) and out.past_key_values.get_seq_length() == b
            rows.append(out.logits[0].clone())
        for (a, b), r in zip(zip(bounds[:-1], bounds[1:]), rows):
            if b - a == 1 and a > 0:         # ONE token next to a cache is the decode step (input_ids.shape[1] == 1, :103): the GEMV path, same values to 1e-3
                assert rel(r, full[a:b]) < 1e-3, (cuts, a)
            else:
                assert torch.equal(r, full[a:b]), (cuts, a, b, rel(r, full[a:b]))
        tok = out.logits[:, -1].argmax(-1, keepdim=True)
        got = [int(tok[0, 0])]
        for _ in range(1, len(toks_ref)):
            out = m(input_ids=tok, past_key_values=out.past_key_values)
            tok = out.logits[:, -1].argmax(-1, keepdim=True)
            got.append(int(tok[0, 0]))
        assert got == toks_ref, cuts
    # a cut INSIDE the placeholder run cannot be spliced: the reference's own error (model/video_chatgpt.py:128)
    out = m(input_ids=torch.tensor([ids[:start + 3]]), max_length=S + 16)
    with pytest.raises(ValueError, match="video start tokens and video end tokens should be the same"):
        m(input_ids=torch.tensor([ids[start + 3:]]), video_spatio_temporal_features=feats, past_key_values=out.past_key_values)
    # ragged batch through the low-level call: two sequences with different prefixes and different appended lengths == their single runs
    A, Bq = ids, ids[:end + 2] + [7, 8, 9, 10, 11]
    kv, _n, _l = m.prefill([A[:end + 1], Bq[:start]], torch.cat([feats, feats]), S + 32)
    _kv, nxt2, lg2 = m.prefill([A[end + 1:], Bq[start:]], torch.cat([feats, feats]), 0, want_logits=True, append_to=kv)
    assert [m.ctx.lib.pgv_kv_len(kv, b) for b in range(2)] == [len(A), len(Bq)]
    with pytest.raises(RuntimeError, match="holds 2 prefilled"):
        m.prefill([[3]], None, 0, append_to=kv)
    for b, seq in enumerate((A, Bq)):
        _k, n1, l1 = m.prefill([seq], feats, S + 32, want_logits=True)
        assert torch.equal(l1[0], lg2[b]) and int(n1[0]) == int(nxt2[b]), b




@pytest.mark.parametrize("tag,proj,image", [("lin", "linear", 224), ("mlp", "mlp2x_gelu", 336)])
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_tiny_llama_golden(ctx, golden_dir, tag, proj, image, dtype, tol):
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    cfg = synth.LlamaCfg(**{**synth.LLAMA_TINY.__dict__, "projector": proj})
    w = synth.make_llama_weights(cfg, seed=int(g[tag + "_weight_seed"]), head_std=float(g["head_std"]))
    m = _model(cfg, w, dtype, image)
    ids = g[tag + "_ids"].tolist()
    feats = torch.from_numpy(g[tag + "_feats"])
    toks_ref = g[tag + "_tokens"].tolist()
    n = len(toks_ref)
    # prefill logits of the last position (the reference computes all positions; we only need the last)
    kv, nxt, logits = m.prefill([ids], feats.to(dtype), 64, want_logits=True)
    assert rel(logits[0], g[tag + "_prefill_logits"][-1]) < tol
    assert rel(logits[0], g[tag + "_step_logits"][0]) < tol
    # step-by-step decode, logits of every step
    got = [int(nxt[0])]
    for i in range(1, n):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        assert rel(lg[0], g[tag + "_step_logits"][i]) < tol, i
        got.append(int(nxt[0]))
    # token-exact greedy: always in fp16 (fixture margins 0.12 / 0.042 vs ~0.002 of fp16 logit noise); in bf16 (noise ~0.02) for the `lin`
    # fixture, whose smallest margin is 6 sigma -- the `mlp` fixture's 0.042 is not, so only its logits are bounded in bf16
    exact = dtype == torch.float16 or tag == "lin"
    if exact:
        assert got == toks_ref
    # generate(): prompt echoed, same tokens, chunked device-side greedy loop
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats[None].to(dtype), do_sample=False,
                     max_new_tokens=n, chunk=5)
    assert out.shape == (1, len(ids) + n)
    assert out[0, :len(ids)].tolist() == ids
    if exact:
        assert out[0, len(ids):].tolist() == toks_ref


def test_splice_errors_and_text_only(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=int(g["lin_weight_seed"]), head_std=float(g["head_std"]))
    m = _model(cfg, w, torch.float16)
    ids = g["lin_ids"].tolist()
    feats = torch.from_numpy(g["lin_feats"]).half()
    e = ids.index(cfg.vocab - 1)
    bad = list(ids); bad[e] = 7
    with pytest.raises(ValueError, match="number of video start tokens"):
        m.generate(torch.tensor([bad]), video_spatio_temporal_features=feats[None], max_new_tokens=1)
    bad = list(ids); bad[e], bad[e + 1] = bad[e + 1], bad[e]
    with pytest.raises(ValueError, match="video end token should follow"):
        m.generate(torch.tensor([bad]), video_spatio_temporal_features=feats[None], max_new_tokens=1)
    # text-only prompt with features given: embeddings untouched (video_chatgpt.py:113-118) == oracle without video
    text = [1, 5, 9, 33, 2, 77]
    ref = ollm.greedy_generate(w, cfg, text, None, cfg.vocab - 2, cfg.vocab - 1, cfg.vocab - 3, 6)
    out = m.generate(torch.tensor([text]), video_spatio_temporal_features=feats[None], max_new_tokens=6)
    assert out[0, len(text):].tolist() == ref


def test_ragged_batch_matches_single(ctx, golden_dir):
    """Data-parallel unit = one clip; batching clips on a GPU must not change any clip's answer."""
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=8, head_std=0.08)     # seed searched on the CPU: every prompt's 10 margins > 0.04 (asserted below)
    m = _model(cfg, w, torch.float16)
    rng = np.random.default_rng(0)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    V = 20
    prompts, feats = [], []
    for b, extra in enumerate((3, 11, 0, 7, 29)):
        head = rng.integers(1, cfg.vocab - 3, 4 + extra).tolist()
        prompts.append([1] + head + [START] + [PATCH] * V + [END] + rng.integers(1, cfg.vocab - 3, 5).tolist())
        feats.append(torch.from_numpy(rng.standard_normal((V, 1024), dtype=np.float32)))
    feats_t = torch.stack(feats).half()
    n = 10
    singles = [m.generate([p], video_spatio_temporal_features=feats_t[i:i + 1], max_new_tokens=n)[0, len(p):].tolist()
               for i, p in enumerate(prompts)]
    batch = m.generate(prompts, video_spatio_temporal_features=feats_t, max_new_tokens=n)
    for i, p in enumerate(prompts):
        assert batch[i, len(p):len(p) + n].tolist() == singles[i]
        ref, margins = ollm.greedy_generate(w, cfg, p, feats[i], START, END, PATCH, n, return_margins=True)
        assert min(margins) > 0.03, (i, margins)             # ~15x the fp16 logit noise of this model: the comparison below is never vacuous
        assert singles[i] == ref


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 115, 116, 117])
def test_ragged_batch_random_shapes(ctx, seed):
    """Randomised version of the property above: 1..16 sequences (seeds > 100: 17..64, the batches that span several MFMA column tiles),
    prompt lengths from a few tokens to several prefill-attention chunks (64-key chunks, 128-query blocks), so prompts end on both sides of
    every tiling boundary; greedy tokens of the batch == single runs."""
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=3, head_std=0.08)
    m = _model(cfg, w, torch.float16)
    rng = np.random.default_rng(seed)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    V = 20
    B = int(rng.integers(1, 17)) if seed < 100 else int(rng.integers(17, 65))
    prompts, feats = [], []
    for b in range(B):
        head = rng.integers(1, cfg.vocab - 3, int(rng.integers(1, 260))).tolist()
        tail = rng.integers(1, cfg.vocab - 3, int(rng.integers(1, 40))).tolist()
        prompts.append([1] + head + [START] + [PATCH] * V + [END] + tail)
        feats.append(torch.from_numpy(rng.standard_normal((V, 1024), dtype=np.float32)))
    feats_t = torch.stack(feats).half()
    n = 7
    batch = m.generate(prompts, video_spatio_temporal_features=feats_t, max_new_tokens=n)
    for i in rng.permutation(B)[:5]:                      # single-sequence reruns of up to five of them
        p = prompts[i]
        single = m.generate([p], video_spatio_temporal_features=feats_t[i:i + 1], max_new_tokens=n)[0, len(p):].tolist()
        assert batch[i, len(p):len(p) + n].tolist() == single, (seed, B, i, len(p))


@pytest.mark.parametrize("B", [17, 32, 33, 48, 64])
def test_wide_batch_tile_edges(ctx, B):
    """Batch sizes on both sides of every MFMA column-tile boundary (16 / 32 / 48): the kernels run 2 or 4 tiles and write every tile's slice of
    the tile-major side arrays, also for tiles that hold no sequence (B = 33 ran 4 tiles over arrays sized for 3 in a first version and
    faulted).  Greedy tokens of the batch through the graph-replayed device loop == single-sequence runs, and a second generate() on the same
    model with another batch size right after (cache and side arrays re-created) still agrees."""
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=3, head_std=0.08)
    m = _model(cfg, w, torch.float16)
    rng = np.random.default_rng(1000 + B)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    V = 20
    prompts = [[1] + rng.integers(1, cfg.vocab - 3, int(rng.integers(1, 90))).tolist() + [START] + [PATCH] * V + [END] + rng.integers(1, cfg.vocab - 3, 3).tolist()
               for _ in range(B)]
    feats_t = torch.from_numpy(rng.standard_normal((B, V, 1024), dtype=np.float32)).half()
    n = 11
    batch = m.generate(prompts, video_spatio_temporal_features=feats_t, max_new_tokens=n, chunk=4)
    for i in (0, 15, 16, B - 1):
        p = prompts[i]
        single = m.generate([p], video_spatio_temporal_features=feats_t[i:i + 1], max_new_tokens=n)[0, len(p):].tolist()
        assert batch[i, len(p):len(p) + n].tolist() == single, (B, i)
    half = m.generate(prompts[:B // 2], video_spatio_temporal_features=feats_t[:B // 2], max_new_tokens=n)
    for i in (0, B // 2 - 1):
        assert half[i, len(prompts[i]):len(prompts[i]) + n].tolist() == batch[i, len(prompts[i]):len(prompts[i]) + n].tolist()


TWO_LAYER_SEED = {"7b": 11, "13b": 8}      # 13B: margins 0.61 (fp16) / 0.50 (bf16)


@pytest.mark.parametrize("shape", ["7b", "13b"])
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 3e-3), (torch.bfloat16, 2.4e-2)])
def test_7b_shaped_two_layers_vs_oracle(ctx, dtype, tol, shape):
    """BASELINE config 3 shapes (H 4096, I 11008, 32 heads, vocab 32003, 356 video rows, ~450-token prompt) and config 5's 13B shapes
    (H 5120, I 13824, 40 heads: ragged K-splits in every decode GEMV) on 2 layers.
    The decoder's parity bar is token-exact greedy decode.  Logits are additionally bounded at 3e-3 normwise (fp16):
    at width 4096 every 16-bit intermediate (normed x, q/k/v, probabilities, attention out, SwiGLU act, projected
    video rows) contributes ~2.8e-4 rms, ~16 of them over two layers -> ~1.5e-3 measured; the residual stream itself is fp32."""
    cfg = synth.LlamaCfg(layers=2) if shape == "7b" else synth.LlamaCfg(layers=2, hidden=5120, inter=13824, heads=40)
    # checkpoints are 16-bit: both the oracle and the HIP path get the same 16-bit-valued weights.  Weight seeds searched on the CPU so
    # that all six greedy steps have an oracle margin > 0.3 in both dtypes (7B: 0.39 / 0.32; 13B: see TWO_LAYER_SEED) -- asserted below.
    w = synth.quantize_weights(synth.make_llama_weights(cfg, seed=TWO_LAYER_SEED[shape], head_std=0.05), str(dtype).split(".")[1])
    m = _model(cfg, w, dtype)
    rng = np.random.default_rng(1)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    V = 356
    ids = [1] + rng.integers(3, 32000, 70).tolist() + [START] + [PATCH] * V + [END] + rng.integers(3, 32000, 12).tolist()
    feats = torch.from_numpy(rng.standard_normal((V, 1024), dtype=np.float32) * 2).to(dtype).float()
    n = 6
    orc = ollm.LlamaOracle(w, cfg)
    lg_ref = orc.prefill(ids, feats, START, END, PATCH)
    kv, nxt, lg = m.prefill([ids], feats.to(dtype), 512, want_logits=True)
    e = rel(lg[0], lg_ref[0])
    print(f"{shape}-shaped 2-layer prefill logits rel err ({dtype}): {e:.3e}")
    assert e < tol
    toks_ref, margins = ollm.greedy_generate(w, cfg, ids, feats, START, END, PATCH, n, return_margins=True)
    toks = [int(nxt[0])] + m.decode_greedy(kv, nxt, n - 1)[0].tolist()
    print(f"{shape}-shaped 2-layer oracle margins ({dtype}): {[round(x, 3) for x in margins]}")
    assert min(margins) > 0.3, margins                       # > 5 sigma of the bf16 logit noise at this width (0.06), ~70 sigma in fp16
    assert toks == toks_ref, (toks, toks_ref, margins)       # token-exact in fp16 AND bf16


@pytest.mark.parametrize("shape", ["7b", "13b"])
def test_context_horizon_4096_two_layers_vs_oracle(ctx, shape):
    """VERDICT r5 #1b: kMaxPos.  A 2-layer 7B-shaped model (unsplit decode attention) and a 2-layer 13B-shaped one (40 heads: the context-split
    attention with its fence-free merge) decode to the LAST position the cache can hold, 4095 (max_position_embeddings of Vicuna-1.5 / LLaVA-1.5 =
    kMaxPos in csrc/llm.hip; the reference's default run ends at 1475 / 1795): `generate()` free-runs from a 441-token prompt until the cache is
    full (3655 tokens through the hipGraph loop), the same tokens are then fed one by one through decode_step and the logits of the last 64
    positions (4032 .. 4095) are compared with ONE causal fp32 oracle pass over all 4096 tokens (RoPE angles, the key loop and the split merge
    at full length); the stepwise argmax there equals what generate() emitted.  One token more is refused."""
    dtype = torch.float16
    cfg = synth.LlamaCfg(layers=2) if shape == "7b" else synth.LlamaCfg(layers=2, hidden=5120, inter=13824, heads=40)
    w = synth.quantize_weights(synth.make_llama_weights(cfg, seed=TWO_LAYER_SEED[shape], head_std=0.05), "float16")
    m = _model(cfg, w, dtype)
    rng = np.random.default_rng(1)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    V, T, TAIL = 356, 4096, 64
    ids = [1] + rng.integers(3, 32000, 70).tolist() + [START] + [PATCH] * V + [END] + rng.integers(3, 32000, 12).tolist()
    S = len(ids)
    feats = torch.from_numpy(rng.standard_normal((V, 1024), dtype=np.float32) * 2).to(dtype)
    with pytest.raises(ValueError, match="exceeds max_position_embeddings"):
        m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats[None], do_sample=False, max_new_tokens=T - S + 1)
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats[None], do_sample=False, max_new_tokens=T - S, chunk=256)
    gen = out[0, S:].tolist()
    assert len(gen) == T - S and out.shape[1] == T
    # stepwise over the same tokens: gen[i] is fed at position S + i; the step that feeds gen[T - S - 2] writes position 4094 ... the cache
    # holds prompt + gen[:-1] = 4095 entries afterwards; one more step (feeding gen[-1] at position 4095) fills it
    kv, nxt, lg = m.prefill([ids], feats, T, want_logits=True)
    assert int(nxt[0]) == gen[0]
    gen_dev = torch.tensor(gen, dtype=torch.int32, device=DEV)
    L, picks = [], []
    for i in range(T - S):
        want = i >= T - S - TAIL
        nxt, lg = m.decode_step(kv, gen_dev[i:i + 1], want_logits=want)
        if want:
            L.append(lg[0].clone()); picks.append(int(nxt[0]))
    assert m.ctx.lib.pgv_kv_len(kv, 0) == T
    with pytest.raises(ValueError, match="cache holds"):
        m.decode_step(kv, gen_dev[:1])
    L = torch.stack(L).float().cpu()                                # logits after feeding positions 4032 .. 4095
    import time
    t0 = time.time()
    with torch.no_grad():
        lg_ref = ollm.LlamaOracle(w, cfg).prefill(ids + gen, feats.float(), START, END, PATCH, all_logits=TAIL, n_prompt=S)
    errs = [rel(L[i], lg_ref[i]) for i in range(TAIL)]
    top2 = torch.topk(lg_ref, 2, dim=-1)
    margins = (top2.values[:, 0] - top2.values[:, 1]).tolist()
    sigma = (L - lg_ref).std(dim=-1).tolist()
    decisive = 0
    for i in range(TAIL):
        if margins[i] > 6.0 * sigma[i]:
            decisive += 1
            assert picks[i] == int(top2.indices[i, 0]), (i, picks[i], int(top2.indices[i, 0]), margins[i], sigma[i])
        if i + 1 < TAIL:
            assert picks[i] == gen[T - S - TAIL + i + 1], i           # the graph-replayed free run emitted exactly the stepwise argmax
    print(f"{shape}-shaped 2-layer model at cache positions {T - TAIL} .. {T - 1}: logits rel err worst {max(errs):.3e}, median {float(np.median(errs)):.3e}; "
          f"{decisive}/{TAIL} decisive positions, all exact (oracle pass over {T} tokens: {time.time() - t0:.0f}s)")
    assert max(errs) < 4e-3 and decisive >= 40


def test_kv_bounds_and_errors(ctx):
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=3, head_std=0.08)
    m = _model(cfg, w, torch.float16)
    kv, nxt, _ = m.prefill([[1, 2, 3]], None, 64)
    m.decode_greedy(kv, nxt, 61)
    with pytest.raises(ValueError, match="cache holds"):
        m.decode_greedy(kv, nxt, 1)                     # 3 + 61 = 64 tokens: full
    with pytest.raises(ValueError, match="outside the vocabulary"):
        m.prefill([[1, 99999]], None, 64)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("mode,N,K,B", [(0, 1536, 512, 3), (1, 512, 768, 8), (2, 1536, 512, 16), (3, 515, 512, 5), (0, 12288, 4096, 8),
                                        # the wide-batch launch shapes of round 6 against fp32 directly (not only through batch invariance): gate/up with
                                        # three (gate, up) pairs per workgroup (688 pairs: a ragged last workgroup), lm_head with eight row blocks
                                        # (2001 row blocks: seven tiles past the matrix in the last workgroup), at two and four column tiles
                                        (2, 22016, 4096, 64), (2, 22016, 4096, 19), (3, 32003, 4096, 40), (3, 32003, 4096, 12), (0, 12288, 4096, 64),
                                        (1, 4096, 4096, 33),
                                        # ... and the 13B shapes: gate/up with four pairs per workgroup (864 pairs -> 216 workgroups), qkv with four row blocks
                                        (2, 27648, 5120, 64), (2, 27648, 5120, 24), (0, 15360, 5120, 64), (0, 15360, 5120, 17)])
def test_gemv_building_block(ctx, dtype, tol, mode, N, K, B):
    """pgv_gemv on fragment-blocked weights (pgv_pack_blocked) vs torch fp32, every epilogue mode."""
    from video_llava_amd import _lib
    g = torch.Generator().manual_seed(N + K + B)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype)
    x = torch.randn(B, K, generator=g).to(dtype)
    Np = (N + 15) // 16 * 16
    wd = w.to(DEV)
    wb = torch.zeros(Np, K, dtype=dtype, device=DEV)
    _lib.check(ctx.lib.pgv_pack_blocked(ctx.handle, _lib.dtype_code(dtype), wd.data_ptr(), N, K, wb.data_ptr(), _lib.stream_ptr()))
    xd = x.to(DEV)
    y_ref = x.float() @ w.float().t()
    if mode == 0:
        out = torch.empty(B, N, dtype=dtype, device=DEV)
    elif mode == 1:
        r0 = torch.randn(B, N, generator=g)
        out = r0.to(DEV).clone()
        y_ref = r0 + y_ref
    elif mode == 2:
        out = torch.empty(B, N // 2, dtype=dtype, device=DEV)
        yy = y_ref.view(B, N // 64, 2, 32)
        y_ref = (torch.nn.functional.silu(yy[:, :, 0]) * yy[:, :, 1]).reshape(B, N // 2)
    else:
        out = torch.empty(B, N, dtype=torch.float32, device=DEV)
    _lib.check(ctx.lib.pgv_gemv(ctx.handle, _lib.dtype_code(dtype), mode, wb.data_ptr(), xd.data_ptr(), K, out.data_ptr(), out.shape[1],
                                N, K, B, _lib.stream_ptr()))
    assert rel(out, y_ref) < (tol if mode in (0, 2) else 1e-5 + (2e-5 if mode == 3 else 0))


# --------------------------------------------------------------------------------------------------
# fp8 weight path (BASELINE config 5): quantiser == CPU twin bit for bit; fp8 GEMV == 16-bit GEMV on the dequantised matrix bit for
# bit; a quantised model decodes token-exactly against the oracle run on the dequantised weights.
# --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,K", [(512, 512), (1536, 1024), (48, 11008)])
def test_fp8_quantizer_matches_cpu_twin(ctx, dtype, N, K):
    from video_llava_amd import _lib
    g = torch.Generator().manual_seed(N * 3 + K)
    w = (torch.randn(N, K, generator=g) * 0.02 * (1 + 10 * torch.rand(N, 1, generator=g))).to(dtype)     # row scales spread over a decade
    w[3].zero_()                                                                                          # an all-zero row: scale 1
    wd = w.to(DEV)
    wb = torch.zeros(N, K, dtype=dtype, device=DEV)
    _lib.check(ctx.lib.pgv_pack_blocked(ctx.handle, _lib.dtype_code(dtype), wd.data_ptr(), N, K, wb.data_ptr(), _lib.stream_ptr()))
    w8 = torch.zeros(N * K, dtype=torch.uint8, device=DEV)
    sc = torch.zeros(N, dtype=torch.float32, device=DEV)
    _lib.check(ctx.lib.pgv_quantize_fp8_blocked(ctx.handle, _lib.dtype_code(dtype), wb.data_ptr(), w8.data_ptr(), sc.data_ptr(), N, K, _lib.stream_ptr()))
    back = torch.empty(N, K, dtype=torch.float32, device=DEV)
    _lib.check(ctx.lib.pgv_unpack_blocked(ctx.handle, _lib.dtype_code(dtype), wb.data_ptr(), back.data_ptr(), N, K, _lib.stream_ptr()))
    twin = ollm.quantize_e4m3_rows(w.float())
    assert torch.equal(back.cpu(), twin), float((back.cpu() - twin).abs().max())
    s = sc.cpu()
    assert torch.equal(torch.log2(s), torch.log2(s).round()) and float(s[3]) == 1.0                      # powers of two
    amax = w.float().abs().amax(1)
    nz = amax > 0
    assert bool((amax[nz] / s[nz] <= 448).all()) and bool((amax[nz] / s[nz] > 224).all())                # tightest power of two
    # the fp8 codes reproduce the dequantised matrix: byte (row, col) of the documented layout
    codes = w8.cpu().view(N // 16, K // 64, 64, 16)
    r, c = 21 % N, 77 % K
    byte = codes[r // 16, c // 64, ((c % 32) // 8) * 16 + r % 16, ((c // 32) % 2) * 8 + c % 8]
    assert float(byte.view(torch.float8_e4m3fn).float() * s[r]) == float(twin[r, c])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode,N,K,B", [(0, 1536, 512, 3), (1, 512, 768, 8), (2, 1536, 512, 16), (3, 512, 1024, 5), (0, 12288, 4096, 8), (2, 22016, 4096, 8),
                                        # the wide launch shapes with fp8 weights (round 6: weights buffered per 64-column group, x per k-block): gate/up with three
                                        # pairs per workgroup at 2 / 4 column tiles, four pairs (13B), a ragged last workgroup over K = 192 (waves 3 .. 7 hold no
                                        # group at all, every wave's second group lies past the end of K), lm_head with eight row blocks at 1 / 2 / 4 column tiles
                                        (2, 22016, 4096, 64), (2, 22016, 4096, 24), (2, 27648, 5120, 64), (2, 27648, 5120, 30), (2, 19264, 192, 40),
                                        (3, 32000, 4096, 64), (3, 32000, 4096, 20), (3, 32000, 4096, 12), (3, 4000, 512, 40)])
def test_gemv_fp8_bit_equal_to_16bit_on_dequantised(ctx, dtype, mode, N, K, B):
    from video_llava_amd import _lib
    g = torch.Generator().manual_seed(N + K + B + 1)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).to(DEV)
    x = torch.randn(B, K, generator=g).to(dtype).to(DEV)
    dc = _lib.dtype_code(dtype)
    wb = torch.zeros(N, K, dtype=dtype, device=DEV)
    _lib.check(ctx.lib.pgv_pack_blocked(ctx.handle, dc, w.data_ptr(), N, K, wb.data_ptr(), _lib.stream_ptr()))
    w8 = torch.zeros(N * K, dtype=torch.uint8, device=DEV)
    sc = torch.zeros(N, dtype=torch.float32, device=DEV)
    _lib.check(ctx.lib.pgv_quantize_fp8_blocked(ctx.handle, dc, wb.data_ptr(), w8.data_ptr(), sc.data_ptr(), N, K, _lib.stream_ptr()))

    def out_buf():
        if mode == 0:
            return torch.empty(B, N, dtype=dtype, device=DEV)
        if mode == 1:
            return torch.ones(B, N, dtype=torch.float32, device=DEV)
        if mode == 2:
            return torch.empty(B, N // 2, dtype=dtype, device=DEV)
        return torch.empty(B, N, dtype=torch.float32, device=DEV)
    o16, o8 = out_buf(), out_buf()
    _lib.check(ctx.lib.pgv_gemv(ctx.handle, dc, mode, wb.data_ptr(), x.data_ptr(), K, o16.data_ptr(), o16.shape[1], N, K, B, _lib.stream_ptr()))
    _lib.check(ctx.lib.pgv_gemv_fp8(ctx.handle, dc, mode, w8.data_ptr(), sc.data_ptr(), x.data_ptr(), K, o8.data_ptr(), o8.shape[1], N, K, B,
                                    _lib.stream_ptr()))
    assert torch.equal(o16, o8)                                     # same weights, same accumulation order, exact scale


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode,N,K", [(0, 12288, 4096), (0, 1536, 1056), (2, 22016, 4096), (3, 515, 11040), (1, 512, 11008)])
def test_gemv_batch_columns_are_independent(ctx, dtype, mode, N, K):
    """A sequence's GEMV result must not depend on how many other sequences share the launch.  With B <= 8 the kernel fetches the x fragments
    of two k-blocks in one wave-load (lanes of the unused batch columns carry the second block, a DPP move redistributes it); with B > 8 it
    issues the two loads separately.  Rows 0..7 of a B = 8 and of a B = 11 launch on the same inputs must agree bit for bit -- 16-bit and fp8
    weights, K with an odd number of 32-blocks (K = 1056, 11040: the last 64-column group is half empty) included."""
    from video_llava_amd import _lib
    g = torch.Generator().manual_seed(N + K)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dtype).to(DEV)
    x = torch.randn(64, K, generator=g).to(dtype).to(DEV)
    dc = _lib.dtype_code(dtype)
    Np = (N + 15) // 16 * 16
    wb = torch.zeros(Np, K, dtype=dtype, device=DEV)
    _lib.check(ctx.lib.pgv_pack_blocked(ctx.handle, dc, w.data_ptr(), N, K, wb.data_ptr(), _lib.stream_ptr()))
    fp8 = K % 64 == 0 and N % 16 == 0
    w8 = sc = None

    def run(B, quantised):
        if mode == 0:
            out = torch.empty(B, N, dtype=dtype, device=DEV)
        elif mode == 1:
            out = torch.ones(B, N, dtype=torch.float32, device=DEV)
        elif mode == 2:
            out = torch.empty(B, N // 2, dtype=dtype, device=DEV)
        else:
            out = torch.empty(B, N, dtype=torch.float32, device=DEV)
        if quantised:
            _lib.check(ctx.lib.pgv_gemv_fp8(ctx.handle, dc, mode, w8.data_ptr(), sc.data_ptr(), x.data_ptr(), K, out.data_ptr(), out.shape[1], N, K, B,
                                            _lib.stream_ptr()))
        else:
            _lib.check(ctx.lib.pgv_gemv(ctx.handle, dc, mode, wb.data_ptr(), x.data_ptr(), K, out.data_ptr(), out.shape[1], N, K, B, _lib.stream_ptr()))
        return out
    for quantised in ([False, True] if fp8 else [False]):
        if quantised:                                                    # rounds wb to the e4m3 grid in place and emits the codes
            w8 = torch.zeros(N * K, dtype=torch.uint8, device=DEV)
            sc = torch.zeros(N, dtype=torch.float32, device=DEV)
            _lib.check(ctx.lib.pgv_quantize_fp8_blocked(ctx.handle, dc, wb.data_ptr(), w8.data_ptr(), sc.data_ptr(), N, K, _lib.stream_ptr()))
        o8, o11, o1 = run(8, quantised), run(11, quantised), run(1, quantised)
        assert torch.equal(o8, o11[:8]) and torch.equal(o1, o11[:1]), (mode, N, K, quantised)
        # batches beyond one MFMA tile (round 4): 2 / 4 column tiles of 16 sequences per weight fragment -- every tile is a separate pass of
        # the 16-column arithmetic, so the rows shared with a narrower launch must stay bit-identical (17 and 33: a tile holding one sequence)
        o16 = run(16, quantised)
        assert torch.equal(o11, o16[:11])
        for B in (17, 32, 33, 48, 64):
            oB = run(B, quantised)
            assert torch.equal(oB[:16], o16), (mode, N, K, quantised, B)
            if B == 64:
                o64 = oB
        for B in (17, 32, 33, 48):
            assert torch.equal(run(B, quantised), o64[:B]), (mode, N, K, quantised, B)
        if mode in (0, 3) and not quantised:
            ref = x.cpu().float() @ w.cpu().float().t()                 # on the CPU: exact fp32 accumulation
            assert rel(o8, ref[:8]) < (8e-3 if dtype == torch.bfloat16 else 1e-3)
            assert rel(o64, ref) < (8e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_llm_fp8_weights_vs_oracle(ctx, golden_dir, dtype, tol):
    cfg = synth.LlamaCfg(**{**synth.LLAMA_TINY.__dict__, "projector": "linear"})
    w = synth.make_llama_weights(cfg, seed=78, head_std=0.08)   # seed searched on the CPU: margins of all 8 steps > 0.3 in fp16 and bf16
    m = _model(cfg, w, dtype).quantize_weights_fp8()
    assert m.is_fp8
    wq = ollm.quantize_llama_weights_fp8(w, round16=dtype)
    for key in ("model.layers.0.self_attn.k_proj.weight", "model.layers.1.mlp.up_proj.weight", "model.layers.1.mlp.down_proj.weight", "lm_head.weight"):
        assert torch.equal(m.get_weight(key).cpu(), torch.from_numpy(wq[key])), key
    V = 100 + 4
    rng = np.random.default_rng(5)
    feats = torch.from_numpy(rng.standard_normal((V, 1024)).astype(np.float32) * 0.5)
    ids = [1, 17, 9, cfg.vocab - 2] + [cfg.vocab - 3] * V + [cfg.vocab - 1, 44, 8]
    NEW = 8
    ref, margins = ollm.greedy_generate(wq, cfg, ids, feats, cfg.vocab - 2, cfg.vocab - 1, cfg.vocab - 3, NEW, return_margins=True)
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats.to(dtype)[None], do_sample=False, max_new_tokens=NEW)
    got = out[0, len(ids):].tolist()
    assert min(margins) > 0.25, margins                       # > 10x the bf16 logit noise of this model: never vacuous
    assert got == ref, (got, ref, margins)                    # token-exact in fp16 and bf16
    # a second model loaded with the dequantised weights (no fp8) produces the same logits bit for bit: prefill reads the 16-bit
    # copy, decode the fp8 copy
    m2 = _model(cfg, wq, dtype)
    kv, nxt, lg = m.prefill([ids], feats.to(dtype), 192, want_logits=True)
    kv2, nxt2, lg2 = m2.prefill([ids], feats.to(dtype), 192, want_logits=True)
    assert torch.equal(lg, lg2)
    for _ in range(4):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        nxt2, lg2 = m2.decode_step(kv2, nxt2, want_logits=True)
        assert torch.equal(lg, lg2) and torch.equal(nxt, nxt2)


@pytest.mark.parametrize("weights", ["bf16", "fp8"])
def test_full_7b_batch_invariance(ctx, weights):
    """BASELINE size (PG-Video-LLaVA-7B shapes: 32 layers, hidden 4096, 356 video tokens), size-independent property: the data-parallel
    unit is one clip, so a clip's greedy answer and its logits must not depend on which other clips share the GPU batch.  Every kernel on
    the path computes a sequence's rows / columns independently of the others, so the match is bitwise -- with 16-bit and with fp8 weights."""
    from video_llava_amd import random_init as ri
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    vocab = 32003
    cfg = VideoChatGPTConfig(vocab_size=vocab, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, eos_token_id=None)
    m = VideoChatGPTLlamaForCausalLM(cfg, VisionConfig(frame_size=224), torch.bfloat16, torch.device(DEV))
    ri.load_streaming(m, ri.iter_llama_tensors(vocab=vocab, hidden=4096, inter=11008, layers=32, device=DEV, dtype=torch.bfloat16, seed=11))
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = vocab - 3, vocab - 2, vocab - 1, True
    if weights == "fp8":
        m.quantize_weights_fp8()
    rng = np.random.default_rng(1)
    V = 356
    prompts = [[1] + rng.integers(3, vocab - 3, 60 + extra).tolist() + [vocab - 2] + [vocab - 3] * V + [vocab - 1] + rng.integers(3, vocab - 3, 6).tolist()
               for extra in (0, 17, 5)]
    feats = torch.from_numpy(rng.standard_normal((3, V, 1024)).astype(np.float32) * 0.5).to(torch.float16).to(DEV)
    n = 6
    kv, nxt_b, lg_b = m.prefill(prompts, feats, 512, want_logits=True)
    steps_b = [(nxt_b.clone(), lg_b.clone())]
    for _ in range(n - 1):
        nxt_b, lg_b = m.decode_step(kv, nxt_b, want_logits=True)
        steps_b.append((nxt_b.clone(), lg_b.clone()))
    del kv
    for i, p in enumerate(prompts):
        kv1, nxt, lg = m.prefill([p], feats[i:i + 1], 512, want_logits=True)
        for t in range(n):
            assert torch.equal(lg[0], steps_b[t][1][i]), (weights, i, t)
            assert int(nxt[0]) == int(steps_b[t][0][i])
            if t + 1 < n:
                nxt, lg = m.decode_step(kv1, nxt, want_logits=True)
        del kv1
    assert torch.isfinite(steps_b[-1][1]).all()


@pytest.mark.parametrize("B,weights", [(32, "bf16"), (64, "bf16"), (64, "fp8")])
def test_full_7b_wide_batch_invariance(ctx, B, weights):
    """Decode batches beyond 16: 32 and 64 sequences per token step share ONE pass over the weights (2 / 4 MFMA column tiles per
    weight fragment).  At PG-Video-LLaVA-7B size (32 layers), ragged prompts with 356 video rows each: the logits of prefill + 5 decode steps
    of sequences 0, 17, B / 2 + 3 and B - 1 inside the batch are BITWISE those of the same sequence decoded alone, and the device-side greedy
    loop (pgv_llm_decode_greedy, graph-replayed) returns the same tokens as the stepwise path."""
    from video_llava_amd import random_init as ri
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    vocab = 32003
    cfg = VideoChatGPTConfig(vocab_size=vocab, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, eos_token_id=None)
    m = VideoChatGPTLlamaForCausalLM(cfg, VisionConfig(frame_size=224), torch.bfloat16, torch.device(DEV))
    ri.load_streaming(m, ri.iter_llama_tensors(vocab=vocab, hidden=4096, inter=11008, layers=32, device=DEV, dtype=torch.bfloat16, seed=11))
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = vocab - 3, vocab - 2, vocab - 1, True
    if weights == "fp8":
        m.quantize_weights_fp8()
    rng = np.random.default_rng(2)
    V = 356
    prompts = [[1] + rng.integers(3, vocab - 3, 40 + int(rng.integers(0, 30))).tolist() + [vocab - 2] + [vocab - 3] * V + [vocab - 1]
               + rng.integers(3, vocab - 3, 6).tolist() for _ in range(B)]
    feats = torch.from_numpy(rng.standard_normal((B, V, 1024)).astype(np.float32) * 0.5).to(torch.float16).to(DEV)
    n = 6
    kv, nxt_b, lg_b = m.prefill(prompts, feats, 512, want_logits=True)
    first = nxt_b.clone()
    steps_b = [(nxt_b.clone(), lg_b.clone())]
    for _ in range(n - 1):
        nxt_b, lg_b = m.decode_step(kv, nxt_b, want_logits=True)
        steps_b.append((nxt_b.clone(), lg_b.clone()))
    assert torch.isfinite(steps_b[-1][1]).all()
    # the chunked device-side loop on the same batch (eager first step, then hipGraph replays): same tokens as the stepwise path
    kv, nxt2, _ = m.prefill(prompts, feats, 512)
    assert torch.equal(nxt2, first)
    toks = m.decode_greedy(kv, nxt2, n - 1)
    for t in range(1, n):
        assert torch.equal(toks[:, t - 1], steps_b[t][0]), t
    del kv
    for i in (0, 17, B // 2 + 3, B - 1):
        kv1, nxt, lg = m.prefill([prompts[i]], feats[i:i + 1], 512, want_logits=True)
        for t in range(n):
            assert torch.equal(lg[0], steps_b[t][1][i]), (weights, B, i, t)
            assert int(nxt[0]) == int(steps_b[t][0][i])
            if t + 1 < n:
                nxt, lg = m.decode_step(kv1, nxt, want_logits=True)
        del kv1


@pytest.mark.parametrize("B,weights", [(12, "bf16"), (24, "bf16"), (64, "bf16"), (24, "fp8"), (64, "fp8")])
def test_13b_shaped_wide_batch_invariance(ctx, B, weights):
    """The 13B launch shapes of the wide batches (round 6: qkv with four row blocks per workgroup, gate/up with FOUR (gate, up) pairs -- 864 pairs -> 216
    workgroups --, lm_head with eight row blocks, the blocked activation layout at hidden 5120, the 10-row-block 8-phase producers; the same shapes with fp8
    weights) inside the whole decode chain: a 2-layer 13B-shaped bf16 model, ragged prompts with 356 video rows; logits of prefill + 4 decode steps of sequences
    0, B / 2 + 1 and B - 1 inside the batch are BITWISE those of the sequence decoded alone (whose launch shapes are the narrow ones, checked against
    the oracle by test_7b_shaped_two_layers_vs_oracle), and the graph-replayed greedy loop returns the stepwise tokens."""
    from video_llava_amd import random_init as ri
    from video_llava_amd.model.video_chatgpt import VideoChatGPTConfig, VideoChatGPTLlamaForCausalLM, VisionConfig
    vocab = 32003
    cfg = VideoChatGPTConfig(vocab_size=vocab, hidden_size=5120, intermediate_size=13824, num_hidden_layers=2, num_attention_heads=40, eos_token_id=None)
    m = VideoChatGPTLlamaForCausalLM(cfg, VisionConfig(frame_size=224), torch.bfloat16, torch.device(DEV))
    ri.load_streaming(m, ri.iter_llama_tensors(vocab=vocab, hidden=5120, inter=13824, layers=2, device=DEV, dtype=torch.bfloat16, seed=13, head_std=0.05))
    if weights == "fp8":
        m.quantize_weights_fp8()
    vc = m.get_model().vision_config
    vc.vid_patch_token, vc.vid_start_token, vc.vid_end_token, vc.use_vid_start_end = vocab - 3, vocab - 2, vocab - 1, True
    rng = np.random.default_rng(20 + B)
    V = 356
    prompts = [[1] + rng.integers(3, vocab - 3, 30 + int(rng.integers(0, 40))).tolist() + [vocab - 2] + [vocab - 3] * V + [vocab - 1]
               + rng.integers(3, vocab - 3, 5).tolist() for _ in range(B)]
    feats = torch.from_numpy(rng.standard_normal((B, V, 1024)).astype(np.float32) * 0.5).to(torch.float16).to(DEV)
    n = 5
    kv, nxt_b, lg_b = m.prefill(prompts, feats, 512, want_logits=True)
    first = nxt_b.clone()
    steps_b = [(nxt_b.clone(), lg_b.clone())]
    for _ in range(n - 1):
        nxt_b, lg_b = m.decode_step(kv, nxt_b, want_logits=True)
        steps_b.append((nxt_b.clone(), lg_b.clone()))
    assert torch.isfinite(steps_b[-1][1]).all()
    kv, nxt2, _ = m.prefill(prompts, feats, 512)
    assert torch.equal(nxt2, first)
    toks = m.decode_greedy(kv, nxt2, n - 1)
    for t in range(1, n):
        assert torch.equal(toks[:, t - 1], steps_b[t][0]), t
    del kv
    for i in (0, B // 2 + 1, B - 1):
        kv1, nxt, lg = m.prefill([prompts[i]], feats[i:i + 1], 512, want_logits=True)
        for t in range(n):
            assert torch.equal(lg[0], steps_b[t][1][i]), (B, i, t)
            assert int(nxt[0]) == int(steps_b[t][0][i])
            if t + 1 < n:
                nxt, lg = m.decode_step(kv1, nxt, want_logits=True)
        del kv1


def test_greedy_pick_takes_the_first_index_on_ties(ctx):
    """torch.argmax (the reference's greedy pick) returns the first maximal index.  With every lm_head row identical all logits of a step are
    bitwise equal, so prefill, the chunked device-side greedy loop and the stepwise path must all pick token 0 -- through the per-workgroup
    candidates of the lm_head GEMV (2001 workgroups at vocabulary 32003; 33 here) and the final reduction over them."""
    cfg = synth.LLAMA_TINY
    w = dict(synth.make_llama_weights(cfg, seed=5, head_std=0.08))
    w["lm_head.weight"] = np.repeat(w["lm_head.weight"][7:8], cfg.vocab, axis=0)
    m = _model(cfg, w, torch.float16)
    ids = [1, 17, 230, 9, 44]
    kv, nxt, lg = m.prefill([ids, ids[:3]], None, 64, want_logits=True)
    assert bool((lg == lg[:, :1]).all()) and nxt.tolist() == [0, 0]
    assert m.decode_greedy(kv, nxt, 6).tolist() == [[0] * 6, [0] * 6]
    nxt2, lg2 = m.decode_step(kv, nxt, want_logits=True)
    assert nxt2.tolist() == [0, 0] and bool((lg2 == lg2[:, :1]).all())
    # a NaN logit is never picked: poison row 0 of lm_head -> logit 0 is NaN -> the pick moves to index 1
    w["lm_head.weight"] = w["lm_head.weight"].copy()
    w["lm_head.weight"][0, 0] = np.nan
    m2 = _model(cfg, w, torch.float16)
    kv, nxt, lg = m2.prefill([ids], None, 64, want_logits=True)
    assert bool(torch.isnan(lg[0, 0])) and nxt.tolist() == [1]


def test_plain_decode_step_after_an_eos_terminated_run(ctx, golden_dir):
    """A cache whose `done` flag was set by decode_greedy (EOS hit) must still serve plain pgv_llm_decode steps: the single-step entry
    records nothing and ignores `done` (it used to emit eos = -1 as the next token and then read embed[-H]).  The step after the run must
    equal the step of a fresh run that never saw EOS."""
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=int(g["lin_weight_seed"]), head_std=float(g["head_std"]))
    m = _model(cfg, w, torch.float16)
    ids, toks_ref = g["lin_ids"].tolist(), g["lin_tokens"].tolist()
    feats = torch.from_numpy(g["lin_feats"]).half()
    kv, nxt, _ = m.prefill([ids], feats, 64)
    assert int(nxt[0]) == toks_ref[0]
    got = m.decode_greedy(kv, nxt, 4, eos_id=toks_ref[2])[0].tolist()          # EOS = the token of step 2: the run goes sticky from there
    assert got == [toks_ref[1], toks_ref[2], toks_ref[2], toks_ref[2]]
    # the cache now holds ids + toks_ref[0..2] + two sticky EOS tokens at positions the reference would never produce; what matters here:
    # a plain step with an explicit token works, returns a real token id and the logits of exactly that context
    nxt2, lg = m.decode_step(kv, torch.tensor([toks_ref[3]], dtype=torch.int32, device=DEV), want_logits=True)
    assert 0 <= int(nxt2[0]) < cfg.vocab and int(nxt2[0]) == int(lg[0].argmax())
    o = ollm.LlamaOracle(w, cfg)
    o.prefill(ids, feats.float(), cfg.vocab - 2, cfg.vocab - 1, cfg.vocab - 3)
    for t in [toks_ref[0], toks_ref[1], toks_ref[2], toks_ref[2]]:
        o.step(t)
    want = o.step(toks_ref[3])[0]
    assert rel(lg[0], want) < 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
def test_folded_rmsnorm_outlier_channels_and_wide_gammas(ctx, dtype, tol):
    """The RMSNorm folded into the decode GEMVs (and the prefill norm) rounds round16(resid * gamma) BEFORE the normalisation, so its
    overflow / underflow envelope differs from HF's (x * rstd).half() * gamma.  Random-init fixtures have gamma = 1 +- 0.1 and no outliers;
    released LLaMA checkpoints have massive-activation channels (~1e3 x the rest) and gammas spanning 1e-2 .. 10.  This case injects both:
    four embedding columns at 300 - 2000 x the typical magnitude, every norm weight log-uniform in [0.02, 8].  Prefill + 6 decode steps against
    the fp32 oracle; the logits must stay finite and within 2 x the plain-fixture tolerance (fp16: overflow would show up as inf / nan)."""
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=31, head_std=0.08)
    rng = np.random.default_rng(32)
    for k in list(w):
        if k.endswith("layernorm.weight") or k == "model.norm.weight":
            w[k] = np.exp(rng.uniform(np.log(0.02), np.log(8.0), w[k].shape)).astype(np.float32)
    emb = w["model.embed_tokens.weight"]
    for col, scale in zip((5, 100, 101, 400), (300.0, 800.0, 2000.0, 1200.0)):
        emb[:, col] *= scale                                           # residual values up to ~2000 x 0.08 x 3 = 500 in these channels
    m = _model(cfg, w, dtype)
    V = 24
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    ids = [1] + rng.integers(3, cfg.vocab - 3, 9).tolist() + [START] + [PATCH] * V + [END] + rng.integers(3, cfg.vocab - 3, 5).tolist()
    feats = torch.from_numpy(rng.standard_normal((V, 1024), dtype=np.float32))
    o = ollm.LlamaOracle(w, cfg)
    ref = o.prefill(ids, feats, START, END, PATCH)[0]
    kv, nxt, lg = m.prefill([ids], feats.to(dtype), 64, want_logits=True)
    assert bool(torch.isfinite(lg).all())
    worst = rel(lg[0], ref)
    peak = float(torch.as_tensor(w["model.embed_tokens.weight"]).abs().max())
    for _ in range(6):
        tok = int(ref.argmax())
        ref = o.step(tok)[0]
        nxt, lg = m.decode_step(kv, torch.tensor([tok], dtype=torch.int32, device=DEV), want_logits=True)     # teacher-forced with the oracle's token
        assert bool(torch.isfinite(lg).all())
        worst = max(worst, rel(lg[0], ref))
    print(f"folded RMSNorm with outlier channels (|embed| up to {peak:.0f}) and gammas in [0.02, 8], {dtype}: worst logits rel err {worst:.3e}")
    assert worst < tol


@pytest.mark.parametrize("switch", ["PGV_LLM_NORM_FOLD=0", "PGV_DATTN_SPLIT=1", "PGV_DATTN_SPLIT=4", "PGV_DATTN_SPLIT=8"])
def test_decoder_path_switches_match_goldens(ctx, golden_dir, switch):
    """Every decode-path variant must reproduce the same reference-generated goldens as the default (the switches are read once per process,
    so each runs in a child process):
      PGV_LLM_NORM_FOLD=0   a standalone RMSNorm in front of every consumer GEMV (HF's order of operations): the bisect switch for
                            real-checkpoint regressions;
      PGV_DATTN_SPLIT=n     decode attention cut into n context parts per (sequence, head), merged across workgroups without fences
                            (default: 2 for head counts like the tiny model's and 13B's, 1 for 7B)."""
    import subprocess
    import sys
    code = f'''
import sys, os, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from oracle import synth
from helpers import make_model
g = np.load({os.path.join(golden_dir, "llama_tiny.npz")!r})
cfg = synth.LLAMA_TINY
w = synth.make_llama_weights(cfg, seed=int(g["lin_weight_seed"]), head_std=float(g["head_std"]))
def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm())
for dtype, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
    m = make_model(cfg, w, dtype)
    ids, want = g["lin_ids"].tolist(), g["lin_tokens"].tolist()
    feats = torch.from_numpy(g["lin_feats"]).to(dtype)
    kv, nxt, lg = m.prefill([ids], feats, 64, want_logits=True)
    assert rel(lg[0], g["lin_step_logits"][0]) < tol
    got = [int(nxt[0])]
    for i in range(1, len(want)):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        assert rel(lg[0], g["lin_step_logits"][i]) < tol, i
        got.append(int(nxt[0]))
    assert got == want, (got, want)
    out = m.generate(torch.tensor([ids]), video_spatio_temporal_features=feats[None], do_sample=False, max_new_tokens=len(want), chunk=5)
    assert out[0, len(ids):].tolist() == want
print("unfolded ok")
'''
    k, v = switch.split("=")
    env = dict(os.environ, **{k: v})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "unfolded ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])



def test_13b_fp8_down_proj_8_phase_form_is_bitwise_the_16_row_kernel(ctx):
    """Config 5's down_proj (fp8, K = 13 824, 320 row blocks) runs the 8-phase residual producer also at batches <= 16
    (gemv.hip choose_gemv: w8 && K >= 12 288).  The form is a different launch shape of the same arithmetic: prefill + 4 decode steps of a
    13B-shaped 2-layer fp8 model at 1, 3 and 16 sequences must give the same logits BIT FOR BIT with the threshold out of reach (the 16-row
    kernel).  The threshold is a launch-shape switch of the LAB library only (-DPGV_LAB: PGV_GEMV_K8_NARROW_MINK; the release library reads no
    launch-shape switches), so both settings run in children bound to libpgv_lab.so -- the same sources, built here if missing -- and the
    release library (this process) must reproduce the same digest."""
    import hashlib
    import subprocess
    import sys
    from video_llava_amd import build as pgv_build
    lab_lib = pgv_build.build(lab=True)
    code = f'''
import sys, os, hashlib, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from video_llava_amd import _lib
if os.environ.get("PGV_TEST_LAB_LIB"): _lib.LIB_PATH = os.environ["PGV_TEST_LAB_LIB"]
from oracle import synth
from helpers import make_model
cfg = synth.LlamaCfg(layers=2, hidden=5120, inter=13824, heads=40)
w = synth.quantize_weights(synth.make_llama_weights(cfg, seed=8, head_std=0.05), "bfloat16")
m = make_model(cfg, w, torch.bfloat16)
m.quantize_weights_fp8()
rng = np.random.default_rng(3)
PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
V = 40
h = hashlib.sha256()
for B in (1, 3, 16):
    prompts = [[1] + rng.integers(3, 32000, 20 + 3 * b).tolist() + [START] + [PATCH] * V + [END] + rng.integers(3, 32000, 5).tolist() for b in range(B)]
    feats = torch.from_numpy(rng.standard_normal((B, V, 1024)).astype(np.float32)).to(torch.bfloat16).cuda()
    kv, nxt, lg = m.prefill(prompts, feats, 128, want_logits=True)
    h.update(lg.cpu().numpy().tobytes())
    for _ in range(4):
        nxt, lg = m.decode_step(kv, nxt, want_logits=True)
        assert torch.isfinite(lg).all()
        h.update(lg.cpu().numpy().tobytes()); h.update(nxt.cpu().numpy().tobytes())
    kv2, nxt2, _ = m.prefill(prompts, feats, 128)
    h.update(m.decode_greedy(kv2, nxt2, 9).cpu().numpy().tobytes())        # eager step + graph replays
    del kv, kv2
print("DIGEST", h.hexdigest())
'''
    digests = []
    for env in (dict(PGV_TEST_LAB_LIB=lab_lib, PGV_GEMV_K8_NARROW_MINK="12288"), dict(PGV_TEST_LAB_LIB=lab_lib, PGV_GEMV_K8_NARROW_MINK="100000000"), {}):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0 and "DIGEST" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
        digests.append(r.stdout.split("DIGEST")[1].split()[0])
    assert digests[0] == digests[1] == digests[2], digests
