#!/bin/bash
# round 5, session i: which host calls sit in the GPU's idle stretches of a headline step (HIP API trace + kernel trace, no counters).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/t -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-side --no-runner --no-latency --no-host-frames --no-profile-pass > $O/bench.json 2> $O/rocprof.err
tail -2 $O/rocprof.err | cut -c1-200
find $O/t -name "*.csv" | head; 
python scripts/trace_idle_api.py $O/t 250 > $O/idle_api.txt 2>&1; head -120 $O/idle_api.txt | cut -c1-200
rm -rf $O/t
