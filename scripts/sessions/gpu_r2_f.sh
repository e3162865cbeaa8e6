#!/bin/bash
# Round-2 trip F: software-pipelined ViT attention A/B (PGV_ATTN_SWP) + 8-step decode graphs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2f; mkdir -p $O
PGV_ATTN_SWP=1 timeout 600 python -m pytest tests/test_gpu_vision.py -m gpu -q -x -k "attention or vit_tiny or vit_l14_config1" > $O/pytest_swp.log 2>&1; tail -3 $O/pytest_swp.log | cut -c1-200
for v in 0 1 0 1; do PGV_ATTN_SWP=$v timeout 300 python scripts/microbench.py attn 2>&1 | grep "T=" | sed "s/^/SWP=$v /"; done | tee $O/attn_ab.txt
timeout 600 python -m pytest tests/test_gpu_llm.py -m gpu -q -x > $O/pytest_llm.log 2>&1; tail -2 $O/pytest_llm.log | cut -c1-200
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency"
timeout 600 $B > $O/bench.json 2> $O/bench.err
PGV_ATTN_SWP=1 timeout 600 $B > $O/bench_swp.json 2> $O/bench_swp.err
for f in bench bench_swp; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    f=d.get("families",{})
    print(sys.argv[1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), {k:round(v["avg_us"],1) for k,v in f.items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
