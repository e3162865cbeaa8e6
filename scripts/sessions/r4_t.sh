#!/bin/bash
# round 4, session t: per-kernel A/B of the direct vs the staged GEMM epilogue (kernel trace of the vision bench under both libraries, same box)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
V="--workload vision --steps 4 --warmup 2 --no-host-frames --no-profile-pass --no-cpu-baseline --no-latency --no-runner --no-side"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p1 -o t -- python bench.py $V > $O/direct.json 2> $O/direct.err
python scripts/rocprof_summary.py $O/p1/t_results.db > $O/trace_direct.txt 2>&1; rm -rf $O/p1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p2 -o t -- python scripts/lab/with_lib.py video_llava_amd/libpgv_staged.so bench.py $V > $O/staged.json 2> $O/staged.err
python scripts/rocprof_summary.py $O/p2/t_results.db > $O/trace_staged.txt 2>&1; rm -rf $O/p2
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p3 -o t -- python bench.py $V > $O/direct2.json 2> $O/direct2.err
python scripts/rocprof_summary.py $O/p3/t_results.db > $O/trace_direct2.txt 2>&1; rm -rf $O/p3
for t in direct staged direct2; do echo "== $t"; head -9 $O/trace_$t.txt | cut -c1-110; done
