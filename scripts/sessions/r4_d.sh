#!/bin/bash
# round 4, session d: context-split decode attention -- parity (llm + runner tests run the tiny models through SPLIT=4) and A/B timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_llm.py tests/test_gpu_runners.py -q -x > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.log
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-runner --no-side"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    f=d.get("families",{})
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), "dattn %.2f us" % f.get("decode_attn",{}).get("avg_us",-1), "gemv %.2f us" % f.get("decode_gemv",{}).get("avg_us",-1), "lat_b1", (d.get("latency_b1") or {}).get("seconds_median"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for sp in 1 4 8 2; do
  PGV_DATTN_SPLIT=$sp timeout 300 python bench.py $S --llm 13b --weights fp8 --no-latency > $O/b13_fp8_split$sp.json 2> $O/b13_fp8_split$sp.err; show $O/b13_fp8_split$sp.json
done
for sp in 1 2 4 8; do
  PGV_DATTN_SPLIT=$sp timeout 300 python bench.py $S > $O/b7_split$sp.json 2> $O/b7_split$sp.err; show $O/b7_split$sp.json
done
