#!/bin/bash
# Round 3 session l: split-tail decode attention (13B at 8 sequences), unfolded decoder switch: parity + 13B bench lines A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -s -x > $O/pytest.log 2>&1; echo "tests rc=$?"
grep -E "passed|failed|split|Error|assert" $O/pytest.log | tail -8
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner"
timeout 600 python bench.py $S --weights fp8 --llm 13b > $O/bench_13b_fp8_split.json 2> $O/err1
PGV_DATTN_SPLIT_TAIL=0 timeout 600 python bench.py $S --weights fp8 --llm 13b > $O/bench_13b_fp8_nosplit.json 2> $O/err2
timeout 600 python bench.py $S --llm 13b > $O/bench_13b_bf16_split.json 2> $O/err3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3l/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "videos/s %.3f" % d["value"], "ms/step %.1f" % d["ms_per_step"], {k: (round(v["avg_us"], 2), round(v.get("gbs", 0))) for k, v in d["families"].items() if k.startswith("decode")})
    except Exception as e:
        print(f, "ERR", e)
PY
