#!/bin/bash
# round 4, session w: position (token in chunk, kernel in token) of the idle gaps inside the decode graphs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass > $O/bench.json 2> $O/rocprof.err
python scripts/lab/idle_pos.py $O/prof/bench_results.db > $O/idle_pos.txt 2>&1
rm -rf $O/prof
head -130 $O/idle_pos.txt
