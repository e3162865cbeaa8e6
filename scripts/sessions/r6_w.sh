#!/bin/bash
# Session r6_w: the 8-phase residual producers with the weight stream running through the pass boundaries (two register buffers in flight at all times,
# both requested before the first x slice) against the previous commit's library (libpgv_head.so).  Tests of the LLM path first.
O=gpurun_out/r6_w; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_llm.py -q -x > $O/pytest_llm.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest_llm.log | cut -c1-200
Q="--steps 2 --warmup 1 --no-side --no-latency --no-cpu-baseline --no-runner"
run() {  # name, args
  timeout 900 python bench.py $Q $2 > $O/new_$1.json 2> $O/new_$1.err
  timeout 900 python scripts/lab/with_lib.py video_llava_amd/libpgv_head.so bench.py $Q $2 > $O/old_$1.json 2> $O/old_$1.err
}
run cfg5 "--llm 13b --weights fp8 --steps 4"
run c64 "--clips-per-gpu 64"
run c32 "--clips-per-gpu 32"
run c64_fp8 "--clips-per-gpu 64 --weights fp8"
run c64_13b "--clips-per-gpu 64 --llm 13b"
run c8 "--steps 4"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_w/*.json")):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    g = j.get("roofline_gemv") or j["roofline"]
    t = j["roofline"].get("token_check") or {}
    print(f.split("/")[-1], round(j["value"], 3), round(j["ms_per_step"], 1), "gemv us", round(g["avg_launch_us"], 2), "frac", round(g["frac"], 3), "token ms", round(t.get("graph_replay_ms_per_token", 0), 3), round(t.get("frac", 0), 3))
PY
