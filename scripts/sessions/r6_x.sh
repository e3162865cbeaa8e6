#!/bin/bash
# Session r6_x: what are the 15 us of an 8-phase producer launch made of?  Lab library, PGV_K8_ABLATE (1 = no x staging, 2 = no MFMA / fragment reads,
# 4 = no weight loads, 8 = no partial-tile stores, 16 = no finish launch), 7B bf16 at 64 clips: graph-replayed ms per token (64 producer launches + 64 finish
# launches per token), so (ms - ms_all_off) / 64 = us per launch pair.
O=gpurun_out/r6_x; mkdir -p $O
python -c "from video_llava_amd import build; build.build(); build.build(lab=True)" > $O/build.log 2>&1
Q="--steps 1 --warmup 0 --new-tokens 40 --no-side --no-latency --no-cpu-baseline --no-runner --clips-per-gpu 64"
for a in 0 1 2 4 8 16 6 7 15 31 3 5; do
  PGV_K8_ABLATE=$a timeout 600 python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so bench.py $Q > $O/abl_$a.json 2> $O/abl_$a.err
done
python - <<'PY'
import json
for a in (0, 1, 2, 4, 8, 16, 3, 5, 6, 7, 15, 31):
    try:
        j = json.loads(open(f"gpurun_out/r6_x/abl_{a}.json").read().strip().splitlines()[-1])
        t = (j["roofline"].get("token_check") or (j.get("roofline_gemv") or {}).get("token_check") or {})
        print(f"abl {a:2d}: token {t.get('graph_replay_ms_per_token', float('nan')):.3f} ms")
    except Exception as e:
        print(a, "unreadable", e)
PY
