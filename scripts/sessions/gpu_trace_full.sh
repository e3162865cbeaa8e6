#!/bin/bash
# rocprofv3 kernel trace of the default bench command only (the last part of gpu_full.sh).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/full
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/full/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/full/bench_under_rocprof.json 2> gpurun_out/full/rocprof.err
python scripts/rocprof_summary.py gpurun_out/full/prof/bench_results.db > gpurun_out/full/kernel_trace.txt 2>&1
python scripts/trace_gaps.py gpurun_out/full/prof/bench_results.db > gpurun_out/full/gaps.txt 2>&1
rm -rf gpurun_out/full/prof
head -14 gpurun_out/full/kernel_trace.txt | cut -c1-150
