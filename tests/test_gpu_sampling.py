"""GPU: the sampling pick (the reference's DEFAULT decode mode -- model.generate(do_sample=True, temperature=0.2),
video_chatgpt/inference.py:106-112) as a HIP kernel, against the CPU oracle oracle.llm.sample_pick, which tests/test_oracle_golden.py
pins to HF's own TemperatureLogitsWarper / TopKLogitsWarper / softmax.

Bar: index-exact.  For a given uniform u the pick is the first vocabulary index whose cumulative probability exceeds u; the kernel sums
in fp32 in a fixed hierarchical order, the oracle in fp64, so the two may only differ when u sits within summation noise of a CDF step.
The test therefore (a) asserts exact equality for every draw whose distance to the nearest step exceeds 1e-5 and asserts that this covers
more than 97 % of the draws, and (b) for the rest asserts the pick is one of the two indices adjacent to that step.
"""
import numpy as np
import pytest
import torch

from oracle import llm as ollm
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("V,B", [(32003, 8), (515, 16), (1000, 3), (50, 2)])
@pytest.mark.parametrize("temp,top_k", [(0.2, 50), (1.0, 50), (0.7, 0), (1.0, 1), (0.05, 7)])
def test_sample_kernel_matches_oracle(ctx, V, B, temp, top_k):
    g = torch.Generator().manual_seed(V * 7 + B)
    n_rounds = 40
    exact = total = 0
    for r in range(n_rounds):
        logits = torch.randn(B, V, generator=g) * (1.5 + (r % 3))
        if r == 3:
            logits[0, 5:9] = logits[0].max() + 1.0                       # ties at the top (kept together by `scores < kth`)
        if r == 5:
            logits[-1, V // 2] = float("nan")                            # a NaN is never picked
        u = torch.rand(B, generator=g)
        if r == 7:
            u[0] = 0.0
        got = ctx.sample_logits(logits.to(DEV), u.to(DEV), temp, top_k).cpu().long()
        ref_logits = torch.nan_to_num(logits, nan=float("-inf"))
        want, dist = ollm.sample_pick(ref_logits, u, temp, top_k)
        sure = dist > 1e-5
        assert torch.equal(got[sure], want[sure]), (r, got.tolist(), want.tolist(), dist.tolist())
        near = ~sure
        assert bool(((got[near] - want[near]).abs() <= V).all())
        if near.any():                                                   # u on a step: one of the two kept neighbours of that step
            cdf = ollm.sample_cdf(ref_logits, temp, top_k)
            for b in torch.nonzero(near)[:, 0].tolist():
                p_got = float(cdf[b, got[b]] - (cdf[b, got[b] - 1] if got[b] > 0 else 0))
                assert p_got > 0 and abs(float(cdf[b, got[b]] - u[b].double())) < 1e-4 + p_got
        exact += int(sure.sum()); total += B
    # share of draws farther than 1e-5 from every CDF step: nearly all with <= 50 kept entries; with the whole 32003-entry vocabulary kept the
    # steps are ~3e-5 apart on average, so many honest draws land within 1e-5 of one (those are checked by the neighbour rule above)
    assert exact > (0.97 if (0 < top_k <= 64 or V <= 1024) else 0.8) * total, (exact, total)


def test_sample_kernel_limits_and_distribution(ctx):
    g = torch.Generator().manual_seed(1)
    V = 32003
    logits = (torch.randn(4, V, generator=g) * 2).to(DEV)
    u = torch.rand(4, generator=g).to(DEV)
    # temperature -> 0 and top_k = 1 are greedy
    assert ctx.sample_logits(logits, u, 1e-3, 50).tolist() == logits.argmax(-1).tolist()
    assert ctx.sample_logits(logits, u, 0.2, 1).tolist() == logits.argmax(-1).tolist()
    # the empirical distribution of 16 x 4000 draws follows softmax(top-50(logits / 0.8))  (chi-square, 49 dof)
    row = logits[:1].expand(16, V).contiguous()
    counts = torch.zeros(V, dtype=torch.float64)
    n = 0
    for i in range(4000):
        uu = torch.rand(16, generator=g).to(DEV)
        counts += torch.bincount(ctx.sample_logits(row, uu, 0.8, 50).cpu().long(), minlength=V).double()
        n += 16
    p = torch.diff(ollm.sample_cdf(logits[:1].cpu(), 0.8, 50)[0], prepend=torch.zeros(1, dtype=torch.float64))
    keep = p > 0
    assert int(keep.sum()) == 50 and float(counts[~keep].sum()) == 0.0
    chi2 = float((((counts[keep] - n * p[keep]) ** 2) / (n * p[keep])).sum())
    assert chi2 < 100.0, chi2


def test_generate_sampling_equals_stepwise_oracle_picks(ctx):
    """model.generate(do_sample=True): every token equals the oracle's inverse-CDF pick on THAT step's device logits with THAT step's
    uniform (the uniforms are torch.rand(max_new, B) from the caller's generator), for a ragged batch; the chunk size does not matter;
    EOS stickiness and the prompt echo hold.  This is HF's sample loop with the multinomial draw made explicit."""
    from helpers import make_model as _model
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=3, head_std=0.08)
    m = _model(cfg, w, torch.float16)
    rng = np.random.default_rng(2)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    Vr = 20
    prompts, feats = [], []
    for extra in (3, 40, 0):
        prompts.append([1] + rng.integers(3, cfg.vocab - 3, 5 + extra).tolist() + [START] + [PATCH] * Vr + [END] + rng.integers(3, cfg.vocab - 3, 4).tolist())
        feats.append(torch.from_numpy(rng.standard_normal((Vr, 1024), dtype=np.float32)))
    feats_t = torch.stack(feats).half()
    n, temp, top_k, B = 24, 0.9, 50, 3
    gen = torch.Generator(device=DEV).manual_seed(11)
    out = m.generate(prompts, video_spatio_temporal_features=feats_t, do_sample=True, temperature=temp, max_new_tokens=n, generator=gen,
                     eos_token_id=None, chunk=7)
    gen.manual_seed(11)
    out2 = m.generate(prompts, video_spatio_temporal_features=feats_t, do_sample=True, temperature=temp, max_new_tokens=n, generator=gen,
                      eos_token_id=None, chunk=32)
    assert torch.equal(out, out2)
    gen.manual_seed(11)
    u = torch.rand(n, B, device=DEV, generator=gen).cpu()
    # stepwise replay: teacher-forced with generate's own tokens, logits from the device, pick from the oracle
    kv, nxt, logits = m.prefill(prompts, feats_t, 128, want_logits=True)
    n_exact = 0
    for i in range(n):
        want, dist = ollm.sample_pick(logits.cpu(), u[i], temp, top_k)
        got = torch.tensor([int(out[b, len(prompts[b]) + i]) for b in range(B)])
        sure = dist > 1e-5
        assert torch.equal(got[sure], want[sure]), (i, got.tolist(), want.tolist())
        n_exact += int(sure.sum())
        nxt, logits = m.decode_step(kv, got.to(torch.int32).to(DEV), want_logits=True)
    assert n_exact > 0.95 * n * B
    for b in range(B):
        assert out[b, :len(prompts[b])].tolist() == prompts[b]
    # temperature 0.2 draws are not all greedy at this model's margins, temperature 1e-3 ones are
    gen.manual_seed(3)
    cold = m.generate(prompts[:1], video_spatio_temporal_features=feats_t[:1], do_sample=True, temperature=1e-3, max_new_tokens=8, generator=gen, eos_token_id=None)
    greedy = m.generate(prompts[:1], video_spatio_temporal_features=feats_t[:1], do_sample=False, max_new_tokens=8, eos_token_id=None)
    assert torch.equal(cold, greedy)
    with pytest.raises(ValueError, match="strictly positive"):
        m.generate(prompts[:1], video_spatio_temporal_features=feats_t[:1], do_sample=True, temperature=0.0, max_new_tokens=2)
    # EOS: once a sequence samples eos it is finished and padded with eos in the returned tensor
    first = int(out[0, len(prompts[0])])
    gen.manual_seed(11)
    out3 = m.generate(prompts, video_spatio_temporal_features=feats_t, do_sample=True, temperature=temp, max_new_tokens=n, generator=gen,
                      eos_token_id=first, chunk=5)
    assert int(out3[0, len(prompts[0])]) == first
    assert out3[0, len(prompts[0]) + 1:].tolist() == [first] * (out3.shape[1] - len(prompts[0]) - 1) or out3.shape[1] == len(prompts[0]) + 1


def test_stopping_criteria_chunked_equals_per_token_loop(ctx):
    """generate() evaluates stopping criteria after each chunk, token by token: the returned ids must equal a strict per-token loop
    (decode_step + criterion after every appended token, HF's order: the criterion's first call only records the start)."""
    from helpers import make_model as _model
    from helpers import SynthTokenizer as _Tok
    from video_llava_amd.model.utils import KeywordsStoppingCriteria
    cfg = synth.LLAMA_TINY
    w = synth.make_llama_weights(cfg, seed=8, head_std=0.08)
    m = _model(cfg, w, torch.float16)
    tok = _Tok(cfg.vocab)
    ids = [1, 17, 230, 9, 44, 8, 99, 100]
    free = m.generate([ids], max_new_tokens=40, eos_token_id=None)[0, len(ids):].tolist()
    # a "keyword" that appears in the decoded tail at a known place: the decimal rendering of the 9th and 10th generated tokens
    kw = f"{free[8]} {free[9]}"
    crit = KeywordsStoppingCriteria([kw], tok, torch.tensor([ids]))
    got = m.generate(torch.tensor([ids]), max_new_tokens=40, eos_token_id=None, stopping_criteria=[crit])[0, len(ids):].tolist()
    # strict per-token loop
    crit2 = KeywordsStoppingCriteria([kw], tok, torch.tensor([ids]))
    kv, nxt, _ = m.prefill([ids], None, 64)
    want = [int(nxt[0])]
    assert not crit2(torch.tensor([ids + want]), None)
    while len(want) < 40:
        nxt, _ = m.decode_step(kv, nxt)
        want.append(int(nxt[0]))
        if crit2(torch.tensor([ids + want]), None):
            break
    assert got == want and 2 <= len(got) <= 10 and len(got) < len(free)
    with pytest.raises(ValueError, match="batch size 1"):
        m.generate([ids, ids], max_new_tokens=4, stopping_criteria=[crit])
