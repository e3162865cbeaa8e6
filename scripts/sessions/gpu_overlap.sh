#!/bin/bash
# EXPERIMENT: vision stage of batch i+1 on a second (optionally CU-masked) stream beside batch i's decode.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/overlap
for m in ${1:-"" first:64 stride:64 stride:128}; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass --overlap-vision --vision-cu-mask "$m" > gpurun_out/overlap/bench_$m.json 2> gpurun_out/overlap/bench_$m.err
  tail -c 300 gpurun_out/overlap/bench_$m.err | grep -v amdgpu.ids
  python -c "
import json,sys; d=json.load(open('gpurun_out/overlap/bench_$m.json')); print('$m', round(d['ms_per_step'],1), d.get('overlap_vision'))"
done
