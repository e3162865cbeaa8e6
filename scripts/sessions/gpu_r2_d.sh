#!/bin/bash
# Round-2 trip D: deferred ssq loads / residual prefetch in the GEMVs, lm_head greedy candidates; decoder tests + full-depth 7B + bench.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_sampling.py tests/test_gpu_loader.py tests/test_gpu_runners.py "tests/test_gpu_fulldepth.py::test_7b_full_depth_fp16_token_exact" -m gpu -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency"
timeout 600 $B > $O/bench.json 2> $O/bench.err
timeout 600 $B --weights fp8 > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err
for f in bench bench_7b_fp8; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    f=d.get("families",{})
    print(sys.argv[1], "value %.3f ms/step %.1f" % (d["value"], d["ms_per_step"]), {k:(round(v["avg_us"],2), round(v.get("gbs",0))) for k,v in f.items() if k.startswith("decode")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace.txt 2>&1
rm -rf $O/prof
head -12 $O/kernel_trace.txt | cut -c1-170
