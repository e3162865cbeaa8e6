#!/bin/bash
# Session r6_y: kernel traces of the wide-batch configurations whose launch shapes changed last (13B bf16 and fp8 weights at 64 clips per step).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6_y; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
Q="--no-cpu-baseline --no-host-frames --no-latency --no-runner --no-side --no-profile-pass"
trace() {   # name, extra args
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o bench -- python bench.py --steps 1 --warmup 1 $Q $2 > $O/bench_under_rocprof_$1.json 2> $O/rocprof_$1.err
  python scripts/rocprof_summary.py $O/prof_$1/bench_results.db > $O/kernel_trace_$1.txt 2>&1
  rm -rf $O/prof_$1
  head -16 $O/kernel_trace_$1.txt | cut -c1-170
}
trace 13b_bf16_clips64 "--llm 13b --clips-per-gpu 64"
trace 7b_fp8_clips64 "--weights fp8 --clips-per-gpu 64"
trace 13b_fp8_clips64 "--llm 13b --weights fp8 --clips-per-gpu 64"
