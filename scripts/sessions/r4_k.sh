#!/bin/bash
# round 4, session k: fp8 x fp8 MFMA form (hi + lo e4m3 activations) -- parity + A/B bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "hilo or a8 or fp8_mfma or fp8" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -12 $O/pytest.log | cut -c1-220
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-runner --no-side --no-latency"
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    f=d.get("families",{})
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), "gemv %.2f us" % f.get("decode_gemv",{}).get("avg_us",-1), "small %.2f us x %d" % (f.get("decode_small",{}).get("avg_us",-1), f.get("decode_small",{}).get("launches_per_step",0)), "roof", d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
for f in 0 1; do
  PGV_FP8_MFMA=$f timeout 400 python bench.py $S --llm 13b --weights fp8 > $O/b13_a8_$f.json 2> $O/b13_a8_$f.err; show $O/b13_a8_$f.json
  PGV_FP8_MFMA=$f timeout 400 python bench.py $S --weights fp8 > $O/b7_a8_$f.json 2> $O/b7_a8_$f.err; show $O/b7_a8_$f.json
done
