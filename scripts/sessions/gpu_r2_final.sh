#!/bin/bash
# Round-2 full check: whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel trace of the same command, PMC traffic passes,
# side benches (fp8 7B / 13B, vision-only, 336 px).  Everything lands in gpurun_out/r2final/ (summaries are copied to profiles/ by hand).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2final4; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -s --durations=8 > $O/pytest.log 2>&1
tail -14 $O/pytest.log | cut -c1-200; grep -n "rel err" $O/pytest.log | cut -c1-160 | head -30
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency > $O/bench_under_rocprof.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace.txt 2>&1
python scripts/trace_gaps.py $O/prof/bench_results.db > $O/gaps.txt 2>&1
rm -rf $O/prof
ARGS="--steps 1 --warmup 0 --new-tokens 9 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o pmc -- python bench.py $ARGS > $O/fetch.json 2> $O/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o pmc -- python bench.py $ARGS > $O/write.json 2> $O/write.err
python scripts/pmc_summary.py $O/fetch/pmc_results.db $O/write/pmc_results.db $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
rm -rf $O/fetch $O/write
S="--steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency"
timeout 600 python bench.py $S --weights fp8 > $O/bench_7b_fp8.json 2> $O/bench_7b_fp8.err
timeout 900 python bench.py $S --weights fp8 --llm 13b > $O/bench_13b_fp8.json 2> $O/bench_13b_fp8.err
timeout 900 python bench.py $S --llm 13b > $O/bench_13b_bf16.json 2> $O/bench_13b_bf16.err
timeout 600 python bench.py $S --workload vision > $O/bench_vision_only.json 2> $O/bench_vision_only.err
timeout 600 python bench.py $S --image 336 > $O/bench_image336.json 2> $O/bench_image336.err
timeout 600 python bench.py $S --clips-per-gpu 16 > $O/bench_clips16.json 2> $O/bench_clips16.err
for f in bench bench_7b_fp8 bench_13b_fp8 bench_13b_bf16 bench_vision_only bench_image336 bench_clips16; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], "value %.3f ms/step %.1f clip_ms %.2f frac %.4f" % (d["value"], d["ms_per_step"], d["clip_feat_ms_per_step"], d["clip_feat_frac_of_mfma_peak"]), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -14 $O/kernel_trace.txt | cut -c1-150
