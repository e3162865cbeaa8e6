#!/bin/bash
# round 5, closing check on the committed tree: the whole GPU suite, smoke, the driver's bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 1500 python -m pytest tests -q -m gpu --durations=10 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -3 $O/pytest.log | cut -c1-200; tail -3 $O/pytest.time
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log | cut -c1-200
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -2 $O/bench.time
python - $O/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("default: value %.3f ms/step %.1f clip frac %.4f roofline %.4f mfma %.4f runner %.3f" % (d["value"], d["ms_per_step"], d["clip_feat_frac_of_mfma_peak"], d["roofline"]["frac"], d["roofline_mfma"]["frac"], d["runner"]["ratio_to_value"]))
for k, v in d.get("side", {}).items():
    print(" side", k, v.get("value"), v.get("ms_per_step"), v.get("clip_feat_frac"), (v.get("roofline") or {}).get("frac"), (v.get("roofline_mfma") or {}).get("frac"), v.get("error"))
PY
