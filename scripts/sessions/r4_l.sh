#!/bin/bash
# round 4, session l: A8 parity tests (fixed file) + non-temporal residual read-modify-write in the GEMM epilogue (variant library) A/B on the vision bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -q -x -k "hilo or a8 or fp8_mfma" > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.log | cut -c1-220
V="--workload vision --steps 10 --warmup 3 --no-host-frames --no-profile-pass"
for i in 1 2; do
  timeout 300 python bench.py $V > $O/vis_base_$i.json 2> $O/vis_base_$i.err
  timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_epint.so bench.py $V > $O/vis_epint_$i.json 2> $O/vis_epint_$i.err
done
for f in vis_base_1 vis_epint_1 vis_base_2 vis_epint_2; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[1].split('/')[-1], "clip ms %.2f frac %.4f" % (d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
timeout 300 python scripts/lab/with_lib.py video_llava_amd/libpgv_epint.so -m pytest tests/test_gpu_vision.py -q -x -k "tiny_golden or 100_frames" > $O/pytest_epint.log 2>&1; echo "epint tests rc=$?"; tail -2 $O/pytest_epint.log
