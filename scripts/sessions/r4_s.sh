#!/bin/bash
# round 4, session s: second half of the next tile's W(1) issued in the epilogue (EARLY_W) vs in the first K-step's G0 (libpgv_noearly.so)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4s2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_vision.py -q -x > $O/pytest_vision.log 2>&1; echo "vision tests rc=$?"; tail -3 $O/pytest_vision.log | cut -c1-200
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for i in 1 2; do
  timeout 300 python bench.py $V > $O/vis_early_$i.json 2> $O/vis_early_$i.err
  timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_noearly.so bench.py $V > $O/vis_noearly_$i.json 2> $O/vis_noearly_$i.err
done
for f in $O/vis_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "clip ms %.2f frac %.4f" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
