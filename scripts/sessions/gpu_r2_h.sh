#!/bin/bash
# Round-2 trip H: decode attention (KV loads first, non-temporal, raw exp2) -- tests + bench + short trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_runners.py -m gpu -q -x > $O/pytest_llm.log 2>&1; tail -2 $O/pytest_llm.log | cut -c1-200
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-profile-pass"
for v in 1 2; do timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('run $v value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee $O/ab.txt
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o t -- python bench.py --steps 1 --warmup 0 --new-tokens 65 --no-cpu-baseline --no-profile-pass --no-host-frames --no-latency > $O/bench_trace.json 2> $O/err.log
python scripts/rocprof_summary.py $O/prof/t_results.db > $O/summary.txt 2>&1
rm -rf $O/prof
head -9 $O/summary.txt | cut -c1-150
