#!/bin/bash
# Round 3 session f: two-lane ViT pass (the frames of a pass split over two streams so one lane's tail overlaps the other's head):
# parity (vision + runner tests), vision-only bench with lanes on / off, single-clip latency, full bench.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vision.py tests/test_gpu_runners.py -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"
tail -3 $O/pytest.log
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for rep in 1 2; do
  timeout 120 python bench.py $V > $O/vis_lanes2_$rep.json 2> $O/vis_lanes2_$rep.err
  PGV_VIT_LANES=1 timeout 120 python bench.py $V > $O/vis_lanes1_$rep.json 2> $O/vis_lanes1_$rep.err
done
timeout 120 python bench.py $V --clips-per-gpu 1 > $O/vis_clip1_lanes2.json 2> $O/vis_clip1_lanes2.err
PGV_VIT_LANES=1 timeout 120 python bench.py $V --clips-per-gpu 1 > $O/vis_clip1_lanes1.json 2> $O/vis_clip1_lanes1.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_lanes2.json 2> $O/bench_lanes2.err
PGV_VIT_LANES=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-frames --no-latency --no-profile-pass > $O/bench_lanes1.json 2> $O/bench_lanes1.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "ms/step %.1f" % d["ms_per_step"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"],
              ("lat_b1 %.3f" % d["latency_b1"]["seconds_median"]) if "latency_b1" in d else "")
    except Exception as e:
        print(f, "ERR", e)
PY
