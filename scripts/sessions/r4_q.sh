#!/bin/bash
# round 4, session q: kernel trace of the 32-clip line (short decode) -- per-kernel times of the wide GEMVs and the 8-phase producers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 1 --warmup 0 --new-tokens 17 --clips-per-gpu 32 --no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass > $O/bench.json 2> $O/rocprof.err
python scripts/rocprof_summary.py $O/prof/bench_results.db > $O/kernel_trace_clips32.txt 2>&1
rm -rf $O/prof
grep -i "gemv\|decode_attn" $O/kernel_trace_clips32.txt | cut -c1-190
