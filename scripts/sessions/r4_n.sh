#!/bin/bash
# round 4, session n: nt A-operand DMA only in the W-resident (LayerNorm-consumer) GEMMs, vision bench A/B at 8 clips and at 1 clip per step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for i in 1 2; do
  timeout 300 python bench.py $V > $O/vis_base_$i.json 2> $O/vis_base_$i.err
  timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_alnin.so bench.py $V > $O/vis_alnin_$i.json 2> $O/vis_alnin_$i.err
done
timeout 300 python bench.py $V --clips-per-gpu 1 --steps 30 > $O/vis1_base.json 2> $O/vis1_base.err
timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_alnin.so bench.py $V --clips-per-gpu 1 --steps 30 > $O/vis1_alnin.json 2> $O/vis1_alnin.err
for f in $O/vis*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "clip ms %.2f frac %.4f" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
