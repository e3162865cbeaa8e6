#!/bin/bash
# round 4, session m: cache-policy variants of the CLIP GEMMs (nt 16-bit output stores of qkv / fc1; nt A-operand DMA) on the vision bench + A8 test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_llm.py -q -x -k "a8 or fp8_mfma" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log | cut -c1-200
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for i in 1 2; do
  timeout 300 python bench.py $V > $O/vis_base_$i.json 2> $O/vis_base_$i.err
  for v in o16nt ant both; do timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_$v.so bench.py $V > $O/vis_${v}_$i.json 2> $O/vis_${v}_$i.err; done
done
for f in $O/vis_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "clip ms %.2f frac %.4f" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
