#!/bin/bash
# round 4, session v: where the GPU idles inside a bench step (union of kernel intervals over all streams, gaps by neighbouring kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4v; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass > $O/bench.json 2> $O/rocprof.err
python scripts/trace_idle.py $O/prof/bench_results.db 15 20000 > $O/idle.txt 2>&1
python scripts/trace_idle.py $O/prof/bench_results.db 2 15 > $O/idle_small.txt 2>&1
rm -rf $O/prof
cat $O/idle.txt | cut -c1-130; head -12 $O/idle_small.txt | cut -c1-130
