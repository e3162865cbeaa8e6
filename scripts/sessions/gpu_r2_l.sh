#!/bin/bash
# Round-2 trip L: shuffle-free reductions in the GEMV epilogues -- lab chain, LLM tests, bench.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; mkdir -p $O
timeout 200 scripts/lab/gemv_chain.exe | tail -6
timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_runners.py tests/test_gpu_fulldepth.py tests/test_gpu_sampling.py tests/test_gpu_loader.py -m gpu -q -x > $O/pytest_llm.log 2>&1; tail -3 $O/pytest_llm.log | cut -c1-200
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-profile-pass"
for v in "" ""; do timeout 900 $B $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v] value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee $O/ab.txt
