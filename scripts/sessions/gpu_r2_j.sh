#!/bin/bash
# Round-2 trip J: GEMV with the pipelined weight stream as the only path + merged x-fragment loads -- LLM tests, bench bf16 / fp8 / 13B fp8.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH="$GRAFT_REPO_ROOT"
O=gpurun_out/r2j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_llm.py tests/test_gpu_runners.py tests/test_gpu_fulldepth.py tests/test_gpu_sampling.py tests/test_gpu_loader.py -m gpu -q -x > $O/pytest_llm.log 2>&1; tail -3 $O/pytest_llm.log | cut -c1-200
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-profile-pass"
for v in "" "--weights fp8" "--weights fp8 --llm 13b" "--llm 13b"; do timeout 900 $B $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$v] value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee $O/ab.txt
PGV_GEMV_X2=0 timeout 600 $B --weights fp8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[fp8, X2 off] value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
