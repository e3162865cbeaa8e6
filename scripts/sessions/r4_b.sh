#!/bin/bash
# round 4, session b: runner tests + the refactored bench (default command, as the driver runs it) with its new side lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_runners.py -q -x > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -12 $O/pytest.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -c 400 $O/bench.err; cat $O/bench.time
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.3f ms/step %.1f clip frac %.4f roof %.3f runner %s" % (d["value"], d["ms_per_step"], d["clip_feat_frac_of_mfma_peak"], d["roofline"]["frac"], (d.get("runner") or {}).get("ratio_to_value")))
print("side", json.dumps(d.get("side"))[:1500])
c=d.get("cpu_baseline",{}); print("cpu", {k:c.get(k) for k in ("value","cores","threads_used","decode_step_s_per_layer","decode_step_by_threads_s","prefill_by_threads_s","config1_by_threads_s","pinning","child_wall_s","error")})
PY
