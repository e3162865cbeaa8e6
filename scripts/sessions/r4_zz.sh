#!/bin/bash
# round 4, last session: kernel trace of config 5 (13B fp8) on the final library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4zz; mkdir -p $O
C5="--llm 13b --weights fp8 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass"
timeout 125 rocprofv3 --kernel-trace --stats -d $O/prof5 -o bench -- python bench.py --steps 1 --warmup 1 $C5 > $O/bench_13b_fp8_under_rocprof.json 2> $O/rocprof5.err
python scripts/rocprof_summary.py $O/prof5/bench_results.db > $O/kernel_trace_13b_fp8.txt 2>&1
rm -rf $O/prof5
head -16 $O/kernel_trace_13b_fp8.txt | cut -c1-150
