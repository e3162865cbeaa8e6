#!/bin/bash
# Round-2 trip E: LayerNorm folded into the CLIP GEMMs (parity tests of the tower, A/B bench of the vision stage).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vision.py -m gpu -q -s > $O/pytest_vision.log 2>&1; tail -4 $O/pytest_vision.log | cut -c1-300; grep -n "rel err" $O/pytest_vision.log | cut -c1-200
PGV_VIT_LN_FOLD=0 timeout 900 python -m pytest tests/test_gpu_vision.py -m gpu -q -s -k "vit_l14" > $O/pytest_vision_nofold.log 2>&1; grep -n "rel err" $O/pytest_vision_nofold.log | cut -c1-200
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --workload vision"
timeout 600 $B > $O/bench_vision_fold.json 2> $O/bench_vision_fold.err
PGV_VIT_LN_FOLD=0 timeout 600 $B > $O/bench_vision_nofold.json 2> $O/bench_vision_nofold.err
timeout 600 $B > $O/bench_vision_fold2.json 2> $O/bench_vision_fold2.err
for f in fold nofold fold2; do python - $O/bench_vision_$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    f=d.get("families",{})
    print(sys.argv[1], "value %.2f clip_ms %.2f frac %.4f" % (d["value"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), {k:(round(v["avg_us"],1), v["launches_per_step"]) for k,v in f.items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
tail -c 300 $O/bench_vision_fold.err
