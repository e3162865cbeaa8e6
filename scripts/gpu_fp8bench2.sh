#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fp8
timeout 600 python -m pytest tests/test_gpu_llm.py -x -q > gpurun_out/fp8/pytest.log 2>&1; tail -2 gpurun_out/fp8/pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/fp8/b7_16.json 2> gpurun_out/fp8/b7_16.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --weights fp8 > gpurun_out/fp8/b7_fp8.json 2> gpurun_out/fp8/b7_fp8.err
