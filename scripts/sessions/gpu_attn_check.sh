#!/bin/bash
# ViT attention: parity tests, microbench, and the bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/attn
timeout 600 python -m pytest tests/test_gpu_vision.py -x -q 2>&1 | tail -3
timeout 120 python scripts/microbench.py attn 2>&1 | grep "T="
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/attn/bench.json 2> gpurun_out/attn/bench.err
python -c "
import json; d=json.load(open('gpurun_out/attn/bench.json')); print(d['value'], d['ms_per_step'], d['clip_feat_tflops'], d['clip_feat_ms_per_step'])"
