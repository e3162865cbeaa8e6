#!/bin/bash
# round 4, session y: the 8-phase residual producer for fp8 down_proj at batches <= 16 (13B: K = 13 824): parity + A/B on config 5 and on 7B fp8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "8_phase or 13b or fp8" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log | cut -c1-200
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass"
for i in 1 2; do
  PGV_GEMV_K8_NARROW_MINK=100000000 timeout 400 python bench.py $S --llm 13b --weights fp8 > $O/c5_off_$i.json 2> $O/c5_off_$i.err
  timeout 400 python bench.py $S --llm 13b --weights fp8 > $O/c5_on_$i.json 2> $O/c5_on_$i.err
done
PGV_GEMV_K8_NARROW_MINK=8192 timeout 400 python bench.py $S --weights fp8 > $O/b7_on.json 2> $O/b7_on.err
timeout 400 python bench.py $S --weights fp8 > $O/b7_off.json 2> $O/b7_off.err
for f in $O/c5_*.json $O/b7_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "value %.4f ms %.2f clip %.2f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
