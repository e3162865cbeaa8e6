#!/bin/bash
# Round 3 session h: does a smaller per-pass working set (frames per lane and pass) speed the CLIP GEMMs up?  lab library, 2 lanes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
V="--workload vision --steps 8 --warmup 2 --no-host-frames --no-profile-pass"
for ck in 400 200 100 50 400; do
  PGV_VIT_CHUNK=$ck timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so bench.py $V > $O/vis_chunk${ck}_$RANDOM.json 2> $O/vis.err
done
for ck in 400 100; do
  PGV_VIT_LANES=1 PGV_VIT_CHUNK=$ck timeout 120 python scripts/lab/with_lib.py video_llava_amd/libpgv_lab.so bench.py $V > $O/vis_lane1_chunk${ck}.json 2> $O/vis.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], "videos/s %.2f" % d["value"], "clip_ms %.2f" % d["clip_feat_ms_per_step"], "frac %.4f" % d["clip_feat_frac_of_mfma_peak"])
    except Exception as e:
        print(f, "ERR", e)
PY
