#!/bin/bash
# round 4, session x: decode graphs of 8 steps only (PGV_GRAPH_STEPS_LONG=0) vs an extra 16- / 32-step graph; golden decode tests under the default
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_llm.py -q -x -k "golden or invariance or ragged or sample" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log | cut -c1-200
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass"
for i in 1 2; do for L in 0 16 32; do PGV_GRAPH_STEPS_LONG=$L timeout 300 python bench.py $S > $O/b_${L}_$i.json 2> $O/b_${L}_$i.err; done; done
for f in $O/b_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "value %.4f ms %.2f clip %.2f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
