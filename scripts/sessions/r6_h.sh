#!/bin/bash
# Round-6 session H: ViT attention -- row sums by v_dot2c against ones, staging of the second part of K / V behind the first query block
# (336 px): parity tests of the vision side, microbenchmark, vision-only bench lines at both resolutions.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6h}; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1; tail -1 $O/build.log
( time timeout 900 python -m pytest tests/test_gpu_vision.py -q -s -x --durations=5 > $O/pytest.log 2>&1 ) 2> $O/pytest.time; echo "tests rc=$?"; tail -8 $O/pytest.log | cut -c1-250
timeout 300 python scripts/microbench.py attn > $O/micro_attn.txt 2>&1; cat $O/micro_attn.txt | grep -v "^$"
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side"
timeout 600 python bench.py $Q --workload vision --steps 10 --warmup 3 > $O/bench_vision_only.json 2> $O/bench_vision_only.err
timeout 600 python bench.py $Q --image 336 --workload vision --steps 6 --warmup 2 > $O/bench_vision_only_336.json 2> $O/bench_vision_only_336.err
timeout 600 python bench.py $Q --dtype fp16 --workload vision --steps 10 --warmup 3 > $O/bench_vision_only_fp16.json 2> $O/bench_vision_only_fp16.err
for f in bench_vision_only bench_vision_only_336 bench_vision_only_fp16; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), {k: (round(v["avg_us"],1), round(v.get("tflops",0),1)) for k,v in d["families"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
