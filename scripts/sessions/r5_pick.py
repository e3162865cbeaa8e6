#!/usr/bin/env python3
"""Print the few numbers of a bench.py JSON line that an A/B session compares."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:                                   # noqa: BLE001
        print(f"  {path}: unreadable ({e})")
        continue
    fam = d.get("families", {})
    parts = [f"value {d['value']:.3f} videos/s", f"{d['ms_per_step']:.1f} ms/step", f"clip {d.get('clip_feat_ms_per_step', 0):.1f} ms ({d.get('clip_feat_frac_of_mfma_peak', 0):.3f})"]
    for k in ("gemm", "vit_attn", "decode_gemv", "decode_attn"):
        if k in fam:
            x = fam[k]
            parts.append(f"{k} {x['avg_us']:.1f}us" + (f" {x['tflops']:.0f}TF" if "tflops" in x else "") + (f" {x['gbs']:.0f}GB/s" if "gbs" in x and "tflops" not in x else ""))
    if d.get("roofline"):
        parts.append(f"roofline {d['roofline']['kernel']} {d['roofline']['frac']:.3f}")
    print("  " + "; ".join(parts))
