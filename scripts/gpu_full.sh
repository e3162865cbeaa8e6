#!/bin/bash
# Full GPU check: test-suite, bench line, rocprofv3 kernel trace of the same bench command (summary -> gpurun_out/full/).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/full
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1
tail -3 gpurun_out/full/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
tail -c 600 gpurun_out/full/bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/full/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side > gpurun_out/full/bench_under_rocprof.json 2> gpurun_out/full/rocprof.err
python scripts/rocprof_summary.py gpurun_out/full/prof/bench_results.db > gpurun_out/full/kernel_trace.txt 2>&1
python scripts/trace_gaps.py gpurun_out/full/prof/bench_results.db > gpurun_out/full/gaps.txt 2>&1
rm -rf gpurun_out/full/prof      # raw trace (tens of MB) stays on the box: gpurun merges at most 64 MiB back
head -12 gpurun_out/full/kernel_trace.txt | cut -c1-150
