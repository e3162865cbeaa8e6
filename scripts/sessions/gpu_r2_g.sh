#!/bin/bash
# Round-2 trip G: qkv GEMV with three row blocks per workgroup (PGV_GEMV_TL3 A/B).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llm.py -m gpu -q -x > $O/pytest_llm.log 2>&1; tail -2 $O/pytest_llm.log | cut -c1-200
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-frames --no-latency --no-profile-pass"
for v in 1 0 1 0; do PGV_GEMV_TL3=$v timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('TL3=$v value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee $O/ab.txt
for v in 1 0; do PGV_GEMV_TL3=$v timeout 600 $B --weights fp8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp8 TL3=$v value %.3f ms/step %.1f' % (d['value'], d['ms_per_step']))"; done | tee -a $O/ab.txt
