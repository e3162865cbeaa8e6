#!/bin/bash
# All GPU tests + one bench line (no rocprof).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/tb
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/tb/pytest.log 2>&1
tail -3 gpurun_out/tb/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/tb/bench.json 2> gpurun_out/tb/bench.err
python -c "
import json; d=json.load(open('gpurun_out/tb/bench.json')); print(d['value'], d['ms_per_step'], d['clip_feat_tflops'], d['clip_feat_ms_per_step']); print({k:(round(v['avg_us'],1), round(v['ms_per_step_est'],1)) for k,v in d['families'].items()})"
