#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/trace
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/trace/prof -o t -- python bench.py --steps 1 --warmup 0 --new-tokens 17 --no-cpu-baseline --no-profile-pass > gpurun_out/trace/bench.json 2> gpurun_out/trace/err.log
python scripts/rocprof_summary.py gpurun_out/trace/prof/t_results.db > gpurun_out/trace/summary.txt 2>&1
head -16 gpurun_out/trace/summary.txt | cut -c1-160
python scripts/trace_gaps.py gpurun_out/trace/prof/t_results.db > gpurun_out/trace/gaps.txt 2>&1
cat gpurun_out/trace/gaps.txt
