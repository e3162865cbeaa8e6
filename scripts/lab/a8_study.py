#!/usr/bin/env python3
"""What would fp8 ACTIVATIONS cost in accuracy?  TEST INFRASTRUCTURE / study script (oracle/__init__.py).

BASELINE configs[4] names an "fp8 MFMA weight path"; this repo's config-5 path is weight-only e4m3 (widened in registers, 16-bit MFMA).  A true
fp8 x fp8 MFMA (v_mfma_f32_16x16x32_fp8_fp8) needs the activation operand in e4m3 as well.  This script measures, on the CPU and at full
depth, what that does to the logits: the fp32 oracle of oracle/llm.py on the fp8-dequantised weights of the 13B full-depth case
(oracle/fulldepth.py), once as is and once with the INPUT of every linear layer (q/k/v, o, gate/up, down, lm_head) rounded to e4m3 with a
per-token power-of-two scale chosen from the exact amax (the most favourable activation quantiser a decode GEMV could implement).
Usage: python scripts/lab/a8_study.py [7b|13b] [n_new] [e4m3|hilo]   (the fp8 x fp8 path it priced was removed in round 5, ABI 300)"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fulldepth as fd      # noqa: E402
from oracle import llm as ollm          # noqa: E402


MODE = "e4m3"      # "e4m3": one e4m3 value per activation;  "hilo": activation = hi + lo, both e4m3 (two fp8 MFMAs per fragment)


def q_e4m3_rows(x: torch.Tensor) -> torch.Tensor:
    """Per-row (token) power-of-two scale from the exact amax, e4m3 round-to-nearest-even, dequantised.  MODE "hilo" (VERDICT r3 #3c, the
    one point between "bf16 activations" and "plain e4m3"): x ~ hi + lo with hi = e4m3(x) and lo = e4m3(x - hi), each under its own
    exact-amax power-of-two row scale -- what two v_mfma_f32_16x16x32_fp8_fp8 per weight fragment (W x hi, W x lo, both accumulating
    in fp32) would compute, with no in-register widening of the weights."""
    hi = ollm.quantize_e4m3_rows(x)
    if MODE != "hilo":
        return hi
    return hi + ollm.quantize_e4m3_rows(x - hi)


class A8Oracle(ollm.LlamaOracle):
    """LlamaOracle with e4m3 activations in front of every matmul (monkey-patches torch.Tensor.__matmul__-free: overrides _layer)."""

    def _layer(self, i, x, cos, sin):
        cfg, w, dt = self.cfg, self.w, self.dtype
        q = f"model.layers.{i}."
        S = x.shape[0]
        nh, hd = cfg.heads, cfg.head_dim
        h = q_e4m3_rows(ollm.rms_norm(x, ollm._t(w, q + "input_layernorm.weight", dt), cfg.eps))
        qs = (h @ ollm._t(w, q + "self_attn.q_proj.weight", dt).t()).view(S, nh, hd).transpose(0, 1)
        ks = (h @ ollm._t(w, q + "self_attn.k_proj.weight", dt).t()).view(S, nh, hd).transpose(0, 1)
        vs = (h @ ollm._t(w, q + "self_attn.v_proj.weight", dt).t()).view(S, nh, hd).transpose(0, 1)
        qs, ks = ollm.apply_rope(qs, cos, sin), ollm.apply_rope(ks, cos, sin)
        if len(self.k) <= i:
            self.k.append(ks); self.v.append(vs)
        else:
            self.k[i] = torch.cat([self.k[i], ks], dim=1); self.v[i] = torch.cat([self.v[i], vs], dim=1)
        K, V = self.k[i], self.v[i]
        Skv = K.shape[1]
        sc = (qs @ K.transpose(-1, -2)) * (hd ** -0.5)
        qpos = torch.arange(Skv - S, Skv)[:, None]
        kpos = torch.arange(Skv)[None, :]
        sc = sc.masked_fill(kpos > qpos, float("-inf"))
        o = (torch.softmax(sc, dim=-1) @ V).transpose(0, 1).reshape(S, nh * hd)
        x = x + q_e4m3_rows(o) @ ollm._t(w, q + "self_attn.o_proj.weight", dt).t()
        h = q_e4m3_rows(ollm.rms_norm(x, ollm._t(w, q + "post_attention_layernorm.weight", dt), cfg.eps))
        g = h @ ollm._t(w, q + "mlp.gate_proj.weight", dt).t()
        u = h @ ollm._t(w, q + "mlp.up_proj.weight", dt).t()
        return x + q_e4m3_rows(torch.nn.functional.silu(g) * u) @ ollm._t(w, q + "mlp.down_proj.weight", dt).t()

    def _forward_embeds(self, x, all_logits=False):
        cfg, w, dt = self.cfg, self.w, self.dtype
        S = x.shape[0]
        cos, sin = ollm.rope_cos_sin(torch.arange(self.pos, self.pos + S), cfg.head_dim, cfg.rope_theta, dt)
        for i in range(cfg.layers):
            x = self._layer(i, x, cos, sin)
        self.pos += S
        x = ollm.rms_norm(x, ollm._t(w, "model.norm.weight", dt), cfg.eps)
        if not all_logits:
            x = x[-1:]
        return q_e4m3_rows(x) @ ollm._t(w, "lm_head.weight", dt).t()


def main():
    global MODE
    name = sys.argv[1] if len(sys.argv) > 1 else "13b"
    n_tf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    MODE = sys.argv[3] if len(sys.argv) > 3 else "e4m3"
    c = fd.CASES[name]
    cfg = c["cfg"]
    t0 = time.time()
    w = fd.make_weights(name, "float16")
    for k in list(w):
        if k == "lm_head.weight" or any(f".{n}." in k for n in ollm.FP8_KEYS):
            w[k] = ollm.quantize_e4m3_rows(w[k].float()).half()
    print(f"weights (fp8-dequantised) {time.time() - t0:.0f}s", flush=True)
    ids, feats = fd.make_prompt(cfg, c["prompt_seed"])
    cont = fd.teacher_tokens(cfg, c["prompt_seed"], n_tf)
    PATCH, START, END = cfg.vocab - 3, cfg.vocab - 2, cfg.vocab - 1
    with torch.no_grad():
        t0 = time.time()
        ref = ollm.LlamaOracle(w, cfg).prefill(list(ids) + cont, feats, START, END, PATCH, all_logits=True)[len(ids) - 1:]
        print(f"fp32 oracle {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        a8 = A8Oracle(w, cfg).prefill(list(ids) + cont, feats, START, END, PATCH, all_logits=True)[len(ids) - 1:]
        print(f"e4m3-activation oracle {time.time() - t0:.0f}s", flush=True)
    err = ((a8 - ref).double().norm(dim=-1) / ref.double().norm(dim=-1))
    top2 = torch.topk(ref, 2, dim=-1)
    margins = top2.values[:, 0] - top2.values[:, 1]
    agree = (a8.argmax(-1) == ref.argmax(-1))
    sigma = (a8 - ref).std(dim=-1)
    safe = margins > 6 * sigma
    print(f"{name} fp8 weights, {cfg.layers} layers, {len(cont) + 1} teacher-forced positions: logits error with {MODE} activations "
          f"{float(err[0]):.3e} at the prefill position, median {float(err.median()):.3e}, worst {float(err.max()):.3e}; argmax agreement "
          f"{int(agree.sum())}/{len(agree)}; positions whose fp32 margin exceeds 6 sigma of that noise: {int(safe.sum())} "
          f"(all agree: {bool(agree[safe].all())}); weight-only fp8 on the GPU at the same depth: 7.0e-3 (fp16 activations), 6.7e-2 (bf16)")


if __name__ == "__main__":
    main()
