#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fp8
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/fp8/b7_16.json 2> gpurun_out/fp8/b7_16.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --weights fp8 > gpurun_out/fp8/b7_fp8.json 2> gpurun_out/fp8/b7_fp8.err
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --llm 13b --weights fp8 > gpurun_out/fp8/b13_fp8.json 2> gpurun_out/fp8/b13_fp8.err
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --llm 13b > gpurun_out/fp8/b13_16.json 2> gpurun_out/fp8/b13_16.err
tail -c 300 gpurun_out/fp8/*.err
