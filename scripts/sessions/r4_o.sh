#!/bin/bash
# round 4, session o: non-temporal K / V staging loads in the ViT attention (variant) on the vision bench; 13B bf16 line after the TL3 fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for i in 1 2; do
  timeout 300 python bench.py $V > $O/vis_base_$i.json 2> $O/vis_base_$i.err
  timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_attnnt.so bench.py $V > $O/vis_attnnt_$i.json 2> $O/vis_attnnt_$i.err
done
for f in $O/vis*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "clip ms %.2f frac %.4f" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --llm 13b > $O/b13_bf16.json 2> $O/b13_bf16.err
python - $O/b13_bf16.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("13B bf16", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
