#!/bin/bash
# round 4, session u: W-resident tile order with TWO column groups (qkv: 6 panels per XCD; g2f: fc1 too, 8 panels) vs the four-group order; one lane (ntm = 804)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PGV_VIT_LANES=1
O=gpurun_out/r4u; mkdir -p $O
timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_g2f.so -m pytest tests/test_gpu_vision.py -q -x -k "benched or 800 or tile" > $O/pytest_g2f.log 2>&1; echo "g2f tests rc=$?"; tail -2 $O/pytest_g2f.log | cut -c1-200
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for i in 1 2; do
  timeout 300 python bench.py $V > $O/vis_base_$i.json 2> $O/vis_base_$i.err
  for v in g2 g2f; do timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_$v.so bench.py $V > $O/vis_${v}_$i.json 2> $O/vis_${v}_$i.err; done
done
for f in $O/vis_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "clip ms %.2f frac %.4f" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
