#!/bin/bash
# Side bench lines: vision-only workload (BASELINE configs[1]) and the 336-px / mlp2x_gelu variant.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/side
timeout 600 python bench.py --steps 3 --warmup 1 --workload vision > gpurun_out/side/vision.json 2> gpurun_out/side/vision.err
timeout 900 python bench.py --steps 2 --warmup 1 --image 336 --no-cpu-baseline > gpurun_out/side/image336.json 2> gpurun_out/side/image336.err
for f in vision image336; do python -c "
import json; d=json.load(open('gpurun_out/side/$f.json')); print('$f', round(d['value'],2), d['unit'], round(d['ms_per_step'],1), 'ms/step; CLIP stage', round(d['clip_feat_tflops'],1), 'TF/s')" || tail -5 gpurun_out/side/$f.err; done
