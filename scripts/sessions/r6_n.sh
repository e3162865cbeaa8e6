#!/bin/bash
# Round-6 session N: register-buffer depth of the 8-phase producers at 64 clips (variant libraries), kernel trace each
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r6n}; mkdir -p $O
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass"
for v in libpgv k8pu6 k8pu3; do
  lib=video_llava_amd/libpgv_$v.so; [ $v = libpgv ] && lib=video_llava_amd/libpgv.so
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o bench -- python scripts/lab/with_lib.py $lib bench.py --steps 1 --warmup 1 $Q --clips-per-gpu 64 > $O/bench_$v.json 2> $O/rocprof_$v.err
  python scripts/rocprof_summary.py $O/prof_$v/bench_results.db > $O/kernel_trace_$v.txt 2>&1; rm -rf $O/prof_$v
  echo "--- $v"; grep -E "gemv_k8" $O/kernel_trace_$v.txt | cut -c1-150
done
