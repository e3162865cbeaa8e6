#!/bin/bash
# Round 3 session g: centred folded LayerNorm (parity incl. mean-dominated rows), lanes 1..4, the driver's bench command with the runner
# measurement and the full-depth CPU baseline.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vision.py tests/test_gpu_runners.py -x -q -s > $O/pytest.log 2>&1; echo "tests rc=$?"
grep "folded LayerNorm\|passed\|failed\|rel err" $O/pytest.log | tail -14
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for ln in 2 1 3 4 2; do
  PGV_VIT_LANES=$ln timeout 120 python bench.py $V > $O/vis_lanes${ln}_$RANDOM.json 2> $O/vis_lanes.err
done
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3g/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              ("lat_b1 %.3f" % d["latency_b1"]["seconds_median"]) if "latency_b1" in d else "", d.get("runner", ""))
        if "cpu_baseline" in d: print("   cpu:", d["cpu_baseline"]["sample"], d["cpu_baseline"].get("decode_step_by_threads_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/bench.err
