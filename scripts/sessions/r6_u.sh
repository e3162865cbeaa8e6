#!/bin/bash
# Session r6_u: wide decode batches with fp8 weights -- the wide launch shapes (gate/up three / four pairs, lm_head eight row blocks per workgroup) for
# the fp8 GEMVs (weights buffered per 64-column group, x fragments per k-block).  Tests first, then release A/B against the previous commit's
# library (video_llava_amd/libpgv_head.so, build_variant from HEAD's sources).
O=gpurun_out/r6_u; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_llm.py -q -x --durations=8 > $O/pytest_llm.log 2>&1; echo "tests rc=$?"; tail -12 $O/pytest_llm.log | cut -c1-200
Q="--steps 2 --warmup 1 --no-side --no-latency --no-cpu-baseline --no-runner --weights fp8"
for cfg in "7b 64" "7b 32" "13b 64" "13b 32"; do
  set -- $cfg
  timeout 900 python bench.py $Q --llm $1 --clips-per-gpu $2 > $O/new_$1_$2.json 2> $O/new_$1_$2.err
  timeout 900 python scripts/lab/with_lib.py video_llava_amd/libpgv_head.so bench.py $Q --llm $1 --clips-per-gpu $2 > $O/old_$1_$2.json 2> $O/old_$1_$2.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6_u/*.json")):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    g = j.get("roofline_gemv") or j["roofline"]
    t = j["roofline"].get("token_check") or {}
    print(f.split("/")[-1], round(j["value"], 3), round(j["ms_per_step"], 1), "gemv us", round(g["avg_launch_us"], 2), "frac", round(g["frac"], 3), "token ms", t.get("graph_replay_ms_per_token"), t.get("frac"))
PY
