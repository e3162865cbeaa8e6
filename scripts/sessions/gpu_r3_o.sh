#!/bin/bash
# Round 3 session o: band height 8 for the narrow CLIP GEMMs (lab variant) A/B; the runner tests incl. the batch-vs-per-clip equality.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_runners.py -x -q > $O/pytest_runners.log 2>&1; echo "runner tests rc=$?"; tail -2 $O/pytest_runners.log
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for rep in 1 2; do
  timeout 120 python bench.py $V > $O/vis_gm4_$rep.json 2> $O/vis.err
  timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_gm8.so bench.py $V > $O/vis_gm8_$rep.json 2> $O/vis.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3o/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"])
    except Exception as e:
        print(f, "ERR", e)
PY
